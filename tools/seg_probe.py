#!/usr/bin/env python3
"""Scratch (round 6): the executor in segments (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill) against one workgroup per frame, on small
batches of 2 MiB frames (HBM-resident, verified or not): correctness against the input, ms per decode, the executor stage alone.
    python tools/seg_probe.py [--frames 1,5,16,64] [--archive gpu|libzstd] [--level 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from oracle import zko


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="1,5,16,64,256")
    ap.add_argument("--archive", default="gpu")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--seg-kib", default="128")
    args = ap.parse_args()
    counts = [int(x) for x in args.frames.split(",")]
    F = bench.FRAME
    nmax = max(counts)
    src = np.frombuffer(zko.gen_chunks(nmax * F), np.uint8)
    arch = None
    if args.archive == "libzstd":
        arch = bench.libzstd_archive_parallel(src, F, args.level, True, max(1, min(96, (os.cpu_count() or 8) - 1)))
    import torch
    import zeekstd_amd as zk
    eng = zk.Engine(0)
    dev = torch.device("cuda:0")
    if arch is None:
        comp, frames = eng.encode_frames(src.tobytes(), F, args.level, True)
    else:
        comp, frames = arch
    c = np.zeros(len(frames) + 1, np.uint64); d = np.zeros(len(frames) + 1, np.uint64)
    c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
    d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_ref = torch.from_numpy(src.copy()).to(dev)
    for nf in counts:
        n = nf * F
        d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev); d_st = torch.zeros(nf, dtype=torch.int32, device=dev)
        for name, ch in [("frame", {"exec_seg": 1}), ("frame1024", {"exec_seg": 1, "exec_lanes": 1024})] + \
                        [(f"seg{k}", {"exec_seg": 2, "seg_kib": int(k)}) for k in args.seg_kib.split(",")] + \
                        [(f"seg{k}mem", {"exec_seg": 2, "seg_kib": int(k), "seg_fill": 1}) for k in args.seg_kib.split(",")[:1]] + [("seg_t256", {"exec_seg": 2, "exec_lanes": 256})]:
            for verify in (False, True):
                eng.set_kernel_choice(reset=0)
                eng.set_kernel_choice(**ch)
                eng.set_profiling(False)
                d_out.fill_(0x5A)
                rc = eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, nf, d_out, n, verify, d_st)
                ok = bool(torch.equal(d_out[:n], d_ref[:n])) and int(d_st.abs().sum().item()) == 0
                torch.cuda.synchronize(); t = time.perf_counter()
                reps = 20 if nf <= 64 else 5
                for _ in range(reps):
                    eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, nf, d_out, n, verify, d_st)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
                fol = eng.checksums_followed()
                eng.set_profiling(True)
                eng.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, nf, d_out, n, verify, d_st)
                kt = {k.replace("zk_k_", ""): round(v, 3) for k, v in eng.kernel_times().items() if v >= 0.02}
                print(f"{nf:4d} frames {name:10s} verify {int(verify)}: {dt * 1e3:7.3f} ms = {n / 2**30 / dt:6.1f} GiB/s rc {rc} ok {ok} followed {fol} {kt}", flush=True)
    eng.set_kernel_choice(reset=0)


if __name__ == "__main__":
    main()
