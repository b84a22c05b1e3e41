#!/usr/bin/env python3
"""Scratch: GPU encoder against its CPU twin on random structured inputs (sizes, levels, frame sizes, prefixes).
   python tools/fuzz_encode_gpu.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import zko
import zeekstd_amd as zk

def piece(rng, n):
    k = int(rng.integers(0, 7))
    s = int(rng.integers(0, 1 << 30))
    if k == 0: return zko.gen_text(n, s)
    if k == 1: return zko.gen_random(n, s)
    if k == 2: return bytes(n)
    if k == 3: return (zko.gen_random(int(rng.integers(1, 300)), s) * (n // 1 + 1))[:n]
    if k == 4: return np.repeat(np.frombuffer(zko.gen_random(n // 7 + 1, s), np.uint8), 7)[:n].tobytes()
    if k == 5: return zko.gen_chunks(n, s % 1000)
    return (zko.gen_text(max(1, n // 5), s) * 6)[:n]

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    eng = zk.Engine(0)
    bad = 0
    for c in range(cases):
        n = int(rng.choice([0, 1, 5, 100, 4095, 4096, 70000, 300000, 1 << 20, (1 << 21) + 17, 3 << 20]))
        parts, left = [], n
        while left > 0:
            m = int(min(left, rng.integers(1, max(2, n))))
            parts.append(piece(rng, m)); left -= m
        data = b"".join(parts)
        level = int(rng.choice([1, 2, 3, 6, 0, 9]))
        fs = int(rng.choice([1000, 65536, 100003, 300001, 1 << 20, 2 << 20, 5 << 20]))      # (odd sizes: unaligned frames in the dense levels' candidate array)
        prefix = None
        if rng.integers(0, 3) == 0:
            pn = int(rng.choice([1, 3, 10, 5000, 57280, 57284, 200000, 1 << 20]))
            prefix = piece(rng, pn)
            if rng.integers(0, 2) and len(data) > 1000 and pn > 2000:      # make the frame share content with the prefix
                data = prefix[pn // 3:pn // 3 + len(data) // 2] + data[len(data) // 2:]
        cks = bool(rng.integers(0, 2))
        comp, frames = eng.encode_frames(data, fs, level, cks, prefix=prefix)
        pos = dpos = 0
        for ci, (cs, ds) in enumerate(frames):
            f = comp[pos:pos + cs]
            want = zko.frame_encode(data[dpos:dpos + ds], level, cks, prefix=prefix)
            if f != want:
                bad += 1
                print("MISMATCH case", c, "n", n, "level", level, "fs", fs, "prefix", None if prefix is None else len(prefix), "frame", ci, len(f), len(want))
                break
            pos += cs; dpos += ds
        if dpos != len(data) and not bad: print("size mismatch", c); bad += 1
    print("cases", cases, "mismatches", bad)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
