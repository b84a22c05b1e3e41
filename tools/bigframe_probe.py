import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
dev = torch.device("cuda:0"); eng = zk.Engine(0)
total = 1 << 30
data = np.frombuffer(zko.gen_chunks(128 << 20), np.uint8)
d_src = torch.from_numpy(np.tile(data, total // len(data))).to(dev)
for fs in (16 << 20, 128 << 20):
    for cks in (True, False):
        cap = int(zk.lib.zk_compress_bound(total, fs))
        d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
        eng.set_profiling(True)
        eng.encode_frames_dev(d_src, total, fs, 1, cks, d_comp, cap)
        eng.encode_frames_dev(d_src, total, fs, 1, cks, d_comp, cap)
        print(fs >> 20, "MiB frames, checksum", cks, {k: round(v, 1) for k, v in eng.kernel_times().items() if "enc" in k})
