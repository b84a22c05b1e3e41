#!/usr/bin/env python3
"""Scratch: the GPU encoder against its CPU twin over random shapes -- inputs stitched from random pieces (text, random bytes, zeros, short
patterns, records, repeats of an earlier slice), random frame sizes (1 ... 3 MiB), levels, checksums, now and then a prefix.  Per case:
every frame's bytes are the twin's, the oracle and the box's libzstd read the archive back.   python tools/scratch_gpu/enc_fuzz.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import zko
from oracle import libzstd_ref as Z
import zeekstd_amd as zk


def piece(rng, have):
    k = int(rng.integers(0, 7))
    n = int(rng.integers(1, 60000)) if rng.integers(0, 4) else int(rng.integers(1, 400))
    s = int(rng.integers(0, 1 << 30))
    if k == 0: return zko.gen_text(n, s % 50)
    if k == 1: return zko.gen_random(n, s)
    if k == 2: return bytes(n)
    if k == 3:
        pat = zko.gen_random(int(rng.integers(1, 12)), s)
        return (pat * (n // len(pat) + 1))[:n]
    if k == 4:
        const = zko.gen_random(int(rng.integers(3, 20)), s)
        r = zko.gen_random(4 * (n // 8 + 1), s + 1)
        return b"".join(r[4 * i:4 * i + 4] + const for i in range(n // 8 + 1))[:n]
    if k == 5 and len(have) > 100:                      # an earlier slice again (near or far)
        a = int(rng.integers(0, len(have) - 50)); return bytes(have[a:a + n])
    return zko.gen_chunks(max(n, 64), s % 20)[:n]


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    eng = zk.Engine()
    wrong = 0
    nbytes = nframes = npre = 0
    t0 = time.time()
    for c in range(cases):
        data = bytearray()
        for _ in range(int(rng.integers(1, 12))):
            data += piece(rng, data)
        data = bytes(data)
        level = int(rng.choice([-3, 1, 1, 1, 2, 3, 3, 6, 9]))
        cks = bool(rng.integers(0, 2))
        r = int(rng.integers(0, 6))
        fs = [int(rng.integers(1, 300)), int(rng.integers(300, 5000)), 32768, 65536, int(rng.integers(5000, 400000)), 3 << 20][r]
        if fs < 300 and len(data) > 20000: data = data[:20000]
        prefix = None
        if rng.integers(0, 4) == 0:
            prefix = zko.gen_text(int(rng.integers(1, 150000)), 41) if rng.integers(0, 2) else data[:int(rng.integers(1, len(data) + 1))]
        comp, frames = eng.encode_frames(data, fs, level, cks, prefix=prefix)
        nbytes += len(data); nframes += len(frames); npre += prefix is not None
        pos = dpos = 0
        bad = None
        for i, (cs, ds) in enumerate(frames):
            f = comp[pos:pos + cs]
            if f != zko.frame_encode(data[dpos:dpos + ds], level, cks, prefix=prefix): bad = "frame %d differs from the twin" % i; break
            try:
                o, used = zko.frame_decode(f, ds, True, prefix=prefix)
                if used != cs or o != data[dpos:dpos + ds]: bad = "frame %d: the oracle reads other bytes" % i; break
            except zko.OracleError as e:
                bad = "frame %d: the oracle refuses (%s)" % (i, e); break
            pos += cs; dpos += ds
        if bad is None and (pos != len(comp) or dpos != len(data)): bad = "sizes"
        if bad is None and Z.load("system") is not None:
            try:
                if Z.decode_stream(comp, len(data), "system", prefix=prefix, window_log_max=31 if prefix else 0) != data: bad = "libzstd reads other bytes"
            except Exception as e:
                bad = "libzstd: %s" % e
        if bad:
            wrong += 1
            print("WRONG case", c, "len", len(data), "fs", fs, "level", level, "cks", cks, "prefix", None if prefix is None else len(prefix), bad)
    print("cases", cases, "bytes", nbytes, "frames", nframes, "with a prefix", npre, "wrong", wrong, "seconds %.1f" % (time.time() - t0))


if __name__ == "__main__":
    main()
