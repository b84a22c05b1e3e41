#!/usr/bin/env python3
"""Damaged archives through the Level-C shim (the real engine behind it) against the box's libzstd, same call sequence (oracle/libzstd_ref.py
decode_stream_verdict, judge_damaged: the rule a case must meet -- the shim decodes whole frames, libzstd hands out blocks as they come).
tests/test_gpu_levelc.py runs 300 of these; this is the open-ended form.
   python tools/fuzz_levelc_gpu.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import GOLDENS
from oracle import zko
from oracle import libzstd_ref as Z


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ref = "1.5.7" if Z.load("1.5.7") is not None else "system"          # the version the reference pins, where the image has it
    assert Z.load("shim") is not None and Z.load(ref) is not None
    stricter = 0
    small = [g for g in GOLDENS if 0 < g.meta["input_len"] <= 400000]
    tally = {}
    wrong = 0
    for c in range(cases):
        g = small[int(rng.integers(0, len(small)))]
        bad = bytearray(g.comp)
        flips = []
        for _ in range(int(rng.integers(1, 4))):
            i, b = int(rng.integers(0, len(bad))), int(rng.integers(0, 8))
            bad[i] ^= 1 << b; flips.append((i, b))
        a, b = Z.decode_stream_verdict(bytes(bad), ref), Z.decode_stream_verdict(bytes(bad), "shim")

        def format_refuses():                           # the oracle (the format, restated) on every frame of the damaged stream
            pos = 0
            for cs, ds in g.frames:
                try:
                    o, used = zko.frame_decode(bytes(bad[pos:pos + cs]), ds + 64, True)
                    if used != cs: return True
                except zko.OracleError:
                    return True
                pos += cs
            return False
        key = ("end" if a[1] == "end" else "more" if a[1] == "more" else "refused", "end" if b[1] == "end" else "more" if b[1] == "more" else "refused")
        tally[key] = tally.get(key, 0) + 1
        w = Z.judge_damaged(a, b, g.input(), format_refuses)
        stricter += w is None and a[1] == "end" and b[1] != "end"
        if w:
            wrong += 1
            print("WRONG case", c, g.name, flips, w, "| libzstd:", a[1], "| shim:", b[1])
    print("cases", cases, "against libzstd", Z.version(ref), "(libzstd, shim):", sorted(tally.items()), "refused with the format where libzstd is laxer", stricter, "wrong", wrong)
    sys.exit(1 if wrong else 0)


if __name__ == "__main__":
    main()
