#!/usr/bin/env python3
"""Scratch: generated frames (tests/helpers/zstd_gen.py) through the kernels, more seeds than the suite has: batches of 1500 frames, whole and damaged
(checksums off; the oracle's verdict per hit frame).   python tools/scratch_gpu/gen_campaign.py [batches] [first seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import offsets_from_frames
from helpers import zstd_gen
from oracle import zko
import zeekstd_amd as zk


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
    eng = zk.Engine()
    wrong = nframes = nrefused = 0
    t0 = time.time()
    for b in range(batches):
        frames, comp, data = [], bytearray(), bytearray()
        kw = dict(max_blocks=10, max_seq=4000, max_lit=100000) if b % 4 == 3 else {}
        for seed in range(seed0 + b * 1500, seed0 + b * 1500 + (300 if kw else 1500)):
            f, out, _ = zstd_gen.generate(seed, zko.xxh64, **kw)
            frames.append((len(f), len(out))); comp += f; data += out
        c, d = offsets_from_frames(frames)
        out, st = eng.decode_frames(bytes(comp) + b"\0" * 8, c, d, verify=True, raise_on_error=False)
        if np.asarray(st).any() or out != bytes(data):
            wrong += 1; print("WRONG batch", b, "whole:", np.flatnonzero(st)[:5])
        sizes, st = eng.frame_content_sizes(bytes(comp), c)
        if np.asarray(st).any() or [int(x) for x in sizes] != [ds for _, ds in frames]:
            wrong += 1; print("WRONG batch", b, "sizes")
        rng = np.random.default_rng(b)
        bad = bytearray(comp); hit = set()
        for _ in range(len(frames) // 2):
            i = int(rng.integers(0, len(bad))); bad[i] ^= 1 << int(rng.integers(0, 8))
            hit.add(int(np.searchsorted(c, i, side="right")) - 1)
        out, st = eng.decode_frames(bytes(bad) + b"\0" * 8, c, d, verify=False, raise_on_error=False)
        for f in range(len(frames)):
            lo, hi = int(d[f]), int(d[f + 1])
            if f not in hit:
                if st[f] != 0 or out[lo:hi] != bytes(data[lo:hi]): wrong += 1; print("WRONG batch", b, "untouched frame", f)
                continue
            try:
                o, used = zko.frame_decode(bytes(bad[int(c[f]):int(c[f + 1])]), hi - lo + 64, False)
                ok = len(o) == hi - lo and used == int(c[f + 1] - c[f])
            except zko.OracleError:
                ok = False
            if ok != (st[f] == 0) or (ok and out[lo:hi] != o): wrong += 1; print("WRONG batch", b, "frame", f, int(st[f]), ok)
            nrefused += not ok
        nframes += len(frames)
    print("frames", nframes, "refused (damaged)", nrefused, "wrong", wrong, "seconds %.1f" % (time.time() - t0))


if __name__ == "__main__":
    main()
