#!/usr/bin/env python3
"""Scratch (round 6): single seeks into archives of zeekstd's DEFAULT frame size (2 MiB) -- set_offset + set_offset_limit + read through the
Decoder handle (bench.py's seek protocol: random offsets, 1 ... 8192 bytes), the executor per frame against the executor in segments, on the
archive this engine writes and on the one the reference's Encoder writes, with and without checksums; the reference's CPU loop beside.
    python tools/seek2m_probe.py [MiB] [seeks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from oracle import zko


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    n = mib << 20
    src = np.frombuffer(zko.gen_chunks(n), np.uint8)
    offs, lens = bench.seek_protocol(trials, n)
    arch = {}
    for cks in (False, True):
        arch[("libzstd", cks)] = bench.libzstd_archive_parallel(src, bench.FRAME, 1, cks, max(1, min(32, (os.cpu_count() or 8) - 1)))
    import zeekstd_amd as zk
    eng = zk.Engine(0)
    for cks in (False, True):
        arch[("gpu", cks)] = eng.encode_frames(src, bench.FRAME, 1, cks)
    for (who, cks), (comp, frames) in arch.items():
        for name, ch in (("frame", {"exec_seg": 1}), ("by shape", {})):
            eng.set_kernel_choice(reset=0)
            eng.set_kernel_choice(**ch)
            r = bench.time_single_seeks(eng, zk, comp, frames, src, offs, lens)
            g, c = r["gpu_decoder_us"], r.get("cpu_reference_us", {})
            print(f"{who:8s} checksums {int(cks)} executor {name:9s}: GPU p50 {g['p50']:8.1f} us  p95 {g['p95']:8.1f}   CPU reference loop p50 {c.get('p50')}", flush=True)
    eng.set_kernel_choice(reset=0)


if __name__ == "__main__":
    main()
