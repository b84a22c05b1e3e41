import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import numpy as np
import bench
import zeekstd_amd as zk
fsz = 65536
nfr = 32
data, _, _, _ = bench.build_inputs(0, nfr, 1, True, 8, False, 0)
src = np.ascontiguousarray(data[:nfr * bench.FRAME])
offs, lens = bench.seek_protocol(600, src.size)
eng = zk.Engine(0)
comp, frames = eng.encode_frames(src, fsz, 1, True)
for lanes in (0, 256, 512, 1024, 128):
    eng.set_kernel_choice(exec_lanes=lanes)
    r = bench.time_single_seeks(eng, zk, comp, frames, src, offs, lens)
    print("exec_lanes", lanes, r["gpu_decoder_us"], flush=True)
