#!/usr/bin/env python3
"""Scratch: do the kernels reach the verdict the same device code reaches on the CPU (tests/sim/zk_sim.cpp)?  Damaged goldens WITHOUT checksum
verification, so a frame is refused only by the format checks: engine status per frame against the simulator's, and the bytes where both accept."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import GOLDENS, sim_decode
import zeekstd_amd as zk

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    eng = zk.Engine()
    small = [g for g in GOLDENS if 0 < g.meta["input_len"] <= 400000]
    same = diff = 0
    for c in range(cases):
        g = small[int(rng.integers(0, len(small)))]
        bad = bytearray(g.comp)
        flips = []
        for _ in range(int(rng.integers(1, 4))):
            i, b = int(rng.integers(0, len(bad))), int(rng.integers(0, 8))
            bad[i] ^= 1 << b; flips.append((i, b))
        co, do = g.offsets()
        out, st = eng.decode_frames(bytes(bad) + b"\0" * 8, co, do, verify=False, raise_on_error=False)
        rc, sout, sst = sim_decode(bytes(bad), g.frames)
        st = np.asarray(st); sst = np.asarray(sst)
        ok = True
        for f in range(len(g.frames)):
            if (st[f] != 0) != (sst[f] != 0): ok = False
            elif st[f] == 0 and out[int(do[f]):int(do[f + 1])] != sout[int(do[f]):int(do[f + 1])]: ok = False
        if ok: same += 1
        else:
            diff += 1
            print("DIFFER", c, g.name, flips, "gpu", [int(x) for x in st[st != sst]][:4], "sim", [int(x) for x in sst[st != sst]][:4])
    print("cases", cases, "same", same, "differ", diff)

if __name__ == "__main__":
    main()
