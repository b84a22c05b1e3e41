import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/zeekstd_amd") else ".")
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
nf = 2048; F = 2 << 20
dev = torch.device("cuda:0"); eng = zk.Engine(0)
data = np.frombuffer(zko.gen_chunks(64 * F), np.uint8)
d_src = torch.from_numpy(np.tile(data, nf // 64)).to(dev)
n = nf * F
cap = int(zk.lib.zk_compress_bound(n, F))
d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
for lvl in (1, 2, 3, 6, 9):
    eng.set_profiling(False)
    eng.encode_frames_dev(d_src, n, F, lvl, True, d_comp, cap)
    torch.cuda.synchronize(); t = time.perf_counter()
    _, cs = eng.encode_frames_dev(d_src, n, F, lvl, True, d_comp, cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    eng.set_profiling(True); eng.encode_frames_dev(d_src, n, F, lvl, True, d_comp, cap)
    print("level", lvl, "ratio", round(n / cs, 3), "encode GiB/s", round(n / 2**30 / dt, 1), {k: round(v, 1) for k, v in eng.kernel_times().items() if "enc" in k})
