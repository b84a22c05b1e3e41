#!/usr/bin/env python3
"""Times the 4 GiB decode of bench.py (configs[2]: 2048 x 2 MiB frames, checksums verified) under a list of kernel choices:
executor residency (4 / 5 workgroups per CU) x where the checksums run (behind the executor: zk_k_xxh64_wide; beside it:
zk_k_xxh64_follow, enqueued behind / in front of the executor).  Per choice: ms per step with two batches in flight and one
batch at a time, frames the checksum waves beside the executor verified, parity of the output with the input.

    python tools/follow_probe.py [--frames 2048] [--steps 8] [--cache DIR] > gpurun_out/follow_probe.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

CHOICES = [
    ("default", {}),
    ("behind_wide", {"xxh64": 2}),
    ("behind_narrow", {"xxh64": 1}),
    ("follow", {"xxh64": 4}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--cache", default=None)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--only", default=None, help="comma-separated names out of CHOICES")
    ap.add_argument("--no-fork", action="store_true", help="generate the input in this process (under rocprofv3)")
    args = ap.parse_args()
    nframes, FRAME = args.frames, bench.FRAME
    workers = 1 if args.no_fork else max(1, min(64, (os.cpu_count() or 8) - 1))
    data, _, _, hashes = bench.build_inputs(0, nframes, args.level, True, workers, False, 0, args.cache)

    import torch
    import zeekstd_amd as zk
    dev = torch.device("cuda", 0)
    eng = zk.Engine(0)
    dsize = nframes * FRAME
    d_src = torch.from_numpy(np.asarray(data)).to(dev)
    cap = int(zk.lib.zk_compress_bound(dsize, FRAME))
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.zeros(nframes, dtype=torch.int32, device=dev)
    d_ds = torch.zeros(nframes, dtype=torch.int32, device=dev)
    nf, csize = eng.encode_frames_dev(d_src, dsize, FRAME, args.level, True, d_comp, cap, d_cs, d_ds)
    cs = d_cs.cpu().numpy().astype(np.uint64)
    c = np.zeros(nframes + 1, np.uint64); c[1:] = np.cumsum(cs)
    d = np.arange(nframes + 1, dtype=np.uint64) * FRAME
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_outs = [torch.zeros(dsize + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_sts = [torch.zeros(nframes, dtype=torch.int32, device=dev) for _ in range(2)]

    def pipelined(k):
        pending, followed = [], []
        for i in range(k):
            if len(pending) == 2:
                assert eng.decode_wait(pending.pop(0)) == 0
                followed.append(eng.checksums_followed())
            pending.append(eng.decode_submit_dev(d_comp, csize, d_c, d_d, 0, nframes, d_outs[i & 1], dsize, True, d_sts[i & 1]))
        for sl in pending:
            assert eng.decode_wait(sl) == 0
            followed.append(eng.checksums_followed())
        return followed

    def one_at_a_time(k):
        followed = []
        for _ in range(k):
            assert eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nframes, d_outs[0], dsize, True, d_sts[0]) == 0
            followed.append(eng.checksums_followed())
        return followed

    out = {"frames": nframes, "steps": args.steps, "compressed_bytes": csize, "choices": {}}
    for name, choice in CHOICES:
        if args.only and name not in args.only.split(","):
            continue
        eng.set_kernel_choice(reset=0)
        eng.set_kernel_choice(**choice)
        for o in d_outs:
            o.zero_()
        pipelined(2); one_at_a_time(1)
        torch.cuda.synchronize()
        t = time.perf_counter(); f2 = pipelined(args.steps); torch.cuda.synchronize(); two = (time.perf_counter() - t) / args.steps
        t = time.perf_counter(); f1 = one_at_a_time(args.steps); torch.cuda.synchronize(); one = (time.perf_counter() - t) / args.steps
        ok = all(torch.equal(o[:dsize], d_src) for o in d_outs) and all(int(s.abs().sum().item()) == 0 for s in d_sts)
        eng.set_profiling(True)
        eng.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nframes, d_outs[0], dsize, True, d_sts[0])
        km = eng.kernel_times()
        eng.set_profiling(False)
        out["choices"][name] = {"choice": choice, "two_in_flight_ms": round(two * 1e3, 3), "GiB_s": round(dsize / two / 2**30, 1),
                                "one_at_a_time_ms": round(one * 1e3, 3), "followed_two_in_flight": f2, "followed_one_at_a_time": f1,
                                "parity": ok, "exec_alone_ms": round(km.get("zk_k_exec", 0.0), 3), "xxh64_alone_ms": round(km.get("zk_k_xxh64", 0.0), 3)}
        print(name, out["choices"][name], file=sys.stderr, flush=True)
    eng.set_kernel_choice(reset=0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
