#!/usr/bin/env python3
"""Scratch: many single seeks (the bench's configs[3] protocol) on an archive CPU libzstd wrote and on one this engine wrote; on a
   failure the seek is found and repeated.   python tools/seek_soak.py [MiB] [trials] [only]
   The decoder is opened WITHOUT verification (ZK_DEC_NO_VERIFY): with it the Decoder checks a frame that offset_limit cuts against its
   checksum and decodes it once more on a mismatch (host/decoder.cpp), which hides the open defect this tool is for.
   ZK_SEEK_DEBUG=1: the failing seek is read again from the decoder's cache (c_api.cpp)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import zko
from oracle import libzstd_ref as Z
import zeekstd_amd as zk
from zeekstd_amd import api
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
n = mib << 20
data = np.frombuffer(zko.gen_chunks(n), np.uint8)
eng = zk.Engine(0)
offs, lens = bench.seek_protocol(trials, n)
lib = zk.lib
lib.zk_decoder_open_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
lib.zk_decoder_time_seeks.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
lib.zk_decoder_time_seeks.restype = C.c_int
lib.zk_decoder_free.argtypes = [C.c_void_p]
def soak(name, comp, frames):
    st = zk.SeekTable.new()
    for c_, d_ in frames: st.log_frame(c_, d_)
    seekable = comp + st.to_bytes()
    o = api.zk_decode_opts(); h = C.c_void_p()
    o.flags = 16                                             # ZK_DEC_NO_VERIFY
    assert lib.zk_decoder_open_bytes(eng._h, seekable, len(seekable), C.byref(o), C.byref(h)) == 0
    buf = np.zeros(8192 + 64, np.uint8); us = np.zeros(trials, np.float64)
    def run(lo, cnt):
        return lib.zk_decoder_time_seeks(h, offs[lo:].ctypes.data, lens[lo:].ctypes.data, cnt, buf.ctypes.data, buf.size, data.ctypes.data, us.ctypes.data)
    bad = 0
    for lo in range(0, trials, 500):
        rc = run(lo, min(500, trials - lo))
        if rc != 0:
            bad += 1
            for i in range(lo, min(lo + 500, trials)):
                if run(i, 1) != 0:
                    print(name, "seek", i, "off", int(offs[i]), "len", int(lens[i]), "frame", int(offs[i]) >> 16, "fails alone; again:", [run(i, 1) for _ in range(3)])
                    break
            else:
                print(name, "a seek in", lo, "..", lo + 500, "failed in the run (rc", rc, ") but none fails alone")
    print(name, "runs of 500 seeks that failed:", bad, "of", (trials + 499) // 500)
    lib.zk_decoder_free(h)
comp, frames = Z.encode_seekable_frames(data.tobytes(), 65536, 1, True)
soak("libzstd-made", comp, frames)
if len(sys.argv) > 3: sys.exit(0)
comp, frames = eng.encode_frames(data, 65536, 1, True)
soak("engine-made", comp, frames)
