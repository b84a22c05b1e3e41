"""Scratch: single seeks with the small path's entropy stage split into its two kernels (zk_k_small_huf / zk_k_small_fse) -- which one is the longer?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import zeekstd_amd as zk
fsz = 65536
nfr = 32
data, _, _, _ = bench.build_inputs(0, nfr, 1, True, 8, False, 0)
src = np.ascontiguousarray(data[:nfr * bench.FRAME])
offs, lens = bench.seek_protocol(400, src.size)
eng = zk.Engine(0)
comp, frames = eng.encode_frames(src, fsz, 1, True)
eng.set_kernel_choice(small_path=2)
r = bench.time_single_seeks(eng, zk, comp, frames, src, offs, lens)
print("split", r["gpu_decoder_us"], flush=True)
