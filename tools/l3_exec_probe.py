#!/usr/bin/env python3
"""Scratch (round 6): the executor's tile width on archives the reference writes at level 3 (2 MiB window: every far match a line
from beyond the L2).  Fewer, wider workgroups keep fewer frames' histories live: 256 lanes = 1 280 resident frames, 512 = 512, 1024 = 256.
    python tools/l3_exec_probe.py [MiB] [level ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from oracle import zko

CHOICES = [("default", {}), ("t256_ring2", {"exec_lanes": 256, "exec_ring": 1}), ("t256_ring4", {"exec_lanes": 256, "exec_ring": 2}),
           ("t512", {"exec_lanes": 512}), ("t1024", {"exec_lanes": 1024}), ("t128", {"exec_lanes": 128})]     # (round 6 also timed exec_far = the sources of far matches touched ahead: slower, removed -- profiles/r06_l3_far_touch_probe.txt)


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    levels = [int(x) for x in sys.argv[2:]] or [3]
    F = bench.FRAME
    cores = os.cpu_count() or 8
    src = np.frombuffer(zko.gen_chunks(mib << 20), np.uint8)
    arch = {}
    for lvl in levels:                       # forked workers: before any HIP initialisation
        t = time.perf_counter()
        arch[lvl] = bench.libzstd_archive_parallel(src, F, lvl, True, max(1, min(96, cores - 1)))
        print(f"level {lvl}: libzstd wrote it in {time.perf_counter() - t:.1f} s, ratio {src.size / len(arch[lvl][0]):.3f}", flush=True)
    import torch
    import zeekstd_amd as zk
    eng = zk.Engine(0)
    dev = torch.device("cuda:0")
    n = src.size
    d_ref = torch.from_numpy(src.copy()).to(dev)
    for lvl, (comp, frames) in arch.items():
        nf = len(frames)
        c = np.zeros(nf + 1, np.uint64); d = np.zeros(nf + 1, np.uint64)
        c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
        d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
        d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
        d_out = torch.empty(n + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(nf, dtype=torch.int32, device=dev)
        for name, ch in CHOICES:
            eng.set_kernel_choice(reset=0)
            eng.set_kernel_choice(**ch)
            eng.set_profiling(False)
            eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, nf, d_out, n, True, d_st)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3):
                rc = eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, nf, d_out, n, True, d_st)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
            eng.set_profiling(True)
            eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, nf, d_out, n, True, d_st)
            kt = {k.replace("zk_k_", ""): round(v, 2) for k, v in eng.kernel_times().items() if v >= 0.05}
            print(f"level {lvl} {name:12s}: decode {n / 2**30 / dt:6.1f} GiB/s ({dt * 1e3:.2f} ms one batch at a time) rc {rc} ok {bool(torch.equal(d_out[:n], d_ref))} {kt}", flush=True)
        eng.set_kernel_choice(reset=0)


if __name__ == "__main__":
    main()
