#!/usr/bin/env python3
"""Scratch: the BATCH kernels' verdicts (more than 4096 blocks in a call) on damaged frames, checksums off, against the oracle frame by frame.
An archive of many frames (this engine's encoder, or the reference loop over the box's libzstd at level 1 / 3), a few hundred flipped bits:
   the oracle refuses a frame  <=> the engine reports it;   both accept => the same bytes (of the frame's length);
   an untouched frame is never reported."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import offsets_from_frames
from oracle import zko
from oracle import libzstd_ref as Z
import zeekstd_amd as zk


def run(eng, name, comp, frames, data, flips, seed):
    c, d = offsets_from_frames(frames)
    rng = np.random.default_rng(seed)
    bad = bytearray(comp)
    hit = set()
    for _ in range(flips):
        i = int(rng.integers(0, len(bad)))
        bad[i] ^= 1 << int(rng.integers(0, 8))
        hit.add(int(np.searchsorted(c, i, side="right")) - 1)
    out, st = eng.decode_frames(bytes(bad) + b"\0" * 8, c, d, verify=False, raise_on_error=False)
    st = np.asarray(st)
    wrong = 0
    for f in range(len(frames)):
        lo, hi = int(d[f]), int(d[f + 1])
        if f not in hit:
            if st[f] != 0 or out[lo:hi] != data[lo:hi]:
                wrong += 1; print("UNTOUCHED frame", f, "status", int(st[f]))
            continue
        try:
            o, used = zko.frame_decode(bytes(bad[int(c[f]):int(c[f + 1])]), hi - lo + 64, False)
            ok = len(o) == hi - lo and used == int(c[f + 1] - c[f])
        except zko.OracleError:
            ok = False
        if ok != (st[f] == 0):
            wrong += 1; print("VERDICT frame", f, "engine", int(st[f]), "oracle accepts" if ok else "oracle refuses")
        elif ok and out[lo:hi] != o:
            wrong += 1; print("BYTES frame", f)
    print(name, "frames", len(frames), "hit", len(hit), "reported", int((st != 0).sum()), "wrong", wrong)
    return wrong


def main():
    eng = zk.Engine()
    wrong = 0
    data = zko.gen_chunks(96 << 20, 3)
    for fs, level in ((65536, 1), (32768, 3), (262144, 1)):
        comp, frames = eng.encode_frames(data, fs, level, False)
        for seed in (1, 2):
            wrong += run(eng, "engine-made fs=%d level=%d" % (fs, level), comp, frames, data, 400, seed)
    sub = data[:48 << 20]
    for fs, level in ((65536, 1), (131072, 3), (1 << 20, 1)):
        comp, frames = Z.encode_seekable_frames(sub, fs, level, False, "system")
        for seed in (1, 2):
            wrong += run(eng, "libzstd-made fs=%d level=%d" % (fs, level), comp, frames, sub, 400, seed)
    print("total wrong", wrong)


if __name__ == "__main__":
    main()
