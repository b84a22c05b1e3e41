#!/usr/bin/env python3
"""Single-seek latency through the zeekstd Decoder API (BASELINE configs[3] protocol at a chosen size)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nseeks = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    fsz = 65536
    cores = os.cpu_count() or 8
    nfr = mib // 2
    data, _, _, _ = bench.build_inputs(0, nfr, 1, True, max(1, min(64, cores - 1)), False, 0)
    from oracle import libzstd_ref as Z
    src = np.ascontiguousarray(data[:nfr * bench.FRAME])
    total = src.size
    offs, lens = bench.seek_protocol(nseeks, total)
    archives = {}
    if Z.load("system") is not None:            # forked workers: before any HIP initialisation
        archives["libzstd_made"] = bench.libzstd_archive_parallel(src, fsz, 1, True, max(1, min(64, cores - 1)))
    import zeekstd_amd as zk
    eng = zk.Engine(0)
    comp, frames = eng.encode_frames(src, fsz, 1, True)
    archives["gpu_made"] = (comp, frames)
    for name, (comp, frames) in archives.items():
        print(name, bench.time_single_seeks(eng, zk, comp, frames, src, offs, lens), flush=True)


if __name__ == "__main__":
    main()
