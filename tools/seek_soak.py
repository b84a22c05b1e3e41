#!/usr/bin/env python3
"""Many single seeks (the bench's configs[3] protocol) without any verification in the Decoder; every failing seek is reported
   with the runs of bytes that differ.   python tools/seek_soak.py [MiB] [trials] [level] [checksum 0/1] [engine-made 0/1]
   ZEEKSTD_AMD_LIB=<path> picks the library (A/B runs of tools/variants/*.so).
   The loop itself is tests/test_gpu_seek_soak.py::soak (the -m gpu test of the same name runs it on four archives)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import zko
from oracle import libzstd_ref as Z
import zeekstd_amd as zk
from test_gpu_seek_soak import soak, FRAME

if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    cks = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
    own = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
    n = mib << 20
    data = np.frombuffer(zko.gen_chunks(n), np.uint8)
    eng = zk.Engine(0)
    print("library:", zk.LIB_PATH, flush=True)
    if own: comp, frames = eng.encode_frames(data, FRAME, level, cks)
    else: comp, frames = Z.encode_seekable_frames(data.tobytes(), FRAME, level, cks)
    t0 = time.time()
    fails, us = soak(eng, comp, frames, data, trials, 0x5EED0003, keep_bytes=True)
    for i, frame, first, nwrong, runs in fails:
        print(f"  seek {i}: frame {frame}, {nwrong} wrong bytes in runs (offset in the read, length): {runs[:12]}")
    print(f"{'engine' if own else 'libzstd'}-made level {level} checksums {int(cks)}: {trials} seeks, {len(fails)} wrong, "
          f"p50 {np.percentile(np.abs(us), 50):.0f} us, {time.time() - t0:.0f} s", flush=True)
