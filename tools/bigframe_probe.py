import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import zko, libzstd_ref as Z
import zeekstd_amd as zk
eng = zk.Engine(0)
N = 48 << 20
data = zko.gen_chunks(N, 77)
for name, (comp, frames) in {"gpu": eng.encode_frames(data, N, 1, True), "libzstd_l3": Z.encode_seekable_frames(data, N, 3, True, "system")}.items():
    c = np.array([0, len(comp)], np.uint64); d = np.array([0, N], np.uint64)
    t = time.time(); out, st = eng.decode_frames(comp + b"\0"*8, c, d, verify=True); dt = time.time() - t
    print(name, "frames", frames, "ok", out == data, st, "decode s", round(dt, 3))
    if name == "gpu": assert Z.decode_stream(comp, N, "system") == data
