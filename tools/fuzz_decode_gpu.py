#!/usr/bin/env python3
"""Scratch: GPU decoder on archives libzstd makes of random structured inputs (levels, frame sizes, prefixes, batch sizes:
   small batches take the seek path, large ones the batch kernels).   python tools/fuzz_decode_gpu.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import zko, libzstd_ref as Z
import zeekstd_amd as zk
from fuzz_encode_gpu import piece

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    which = "1.5.7" if Z.load("1.5.7") is not None else "system"
    eng = zk.Engine(0)
    bad = 0
    for c in range(cases):
        n = int(rng.choice([0, 1, 100, 4096, 70000, 300000, 1 << 20, (1 << 21) + 17, 6 << 20, 24 << 20]))
        parts, left = [], n
        while left > 0:
            m = int(min(left, rng.integers(1, max(2, n))))
            parts.append(piece(rng, m)); left -= m
        data = b"".join(parts)
        level = int(rng.choice([-3, 1, 1, 3, 3, 5, 9, 19])) if n <= (6 << 20) else int(rng.choice([1, 3]))
        fs = int(rng.choice([1000, 65536, 1 << 20, 2 << 20, 5 << 20])) if n <= (6 << 20) else int(rng.choice([65536, 2 << 20]))
        prefix = None
        if rng.integers(0, 4) == 0 and n:
            prefix = piece(rng, int(rng.choice([10, 5000, 200000, 1 << 20])))
        cks = bool(rng.integers(0, 2))
        comp, frames = Z.encode_seekable_frames(data, fs, level, cks, which, prefix=prefix)
        c_off = np.concatenate([[0], np.cumsum([f[0] for f in frames])]).astype(np.uint64)
        d_off = np.concatenate([[0], np.cumsum([f[1] for f in frames])]).astype(np.uint64)
        out, st = eng.decode_frames(comp + b"\0" * 8, c_off, d_off, verify=True, raise_on_error=False, prefix=prefix)
        if st.any() or bytes(out[:len(data)]) != data:
            bad += 1
            print("MISMATCH case", c, "n", n, "level", level, "fs", fs, "frames", len(frames), "prefix", None if prefix is None else len(prefix), "status", st[:4])
    print("cases", cases, "mismatches", bad, "libzstd", which)
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
