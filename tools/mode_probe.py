#!/usr/bin/env python3
"""Scratch: host-to-host rates of the less travelled modes -- prefix (patch) mode both ways, FrameSizePolicy::Compressed(n)."""
import os, sys, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import zko
import zeekstd_amd as zk
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 512 << 20
eng = zk.Engine(0)
data = np.frombuffer(zko.gen_chunks(n), np.uint8)
prefix = zko.gen_chunks(3 << 20)[2 << 20:]
F = 2 << 20
def offs(frames):
    c = np.zeros(len(frames) + 1, np.uint64); d = np.zeros(len(frames) + 1, np.uint64)
    c[1:] = np.cumsum([f[0] for f in frames]); d[1:] = np.cumsum([f[1] for f in frames])
    return c, d
for pre in (None, prefix):
    eng.encode_frames(data[:64 << 20], F, 1, True, prefix=pre)
    t = time.perf_counter(); comp, frames = eng.encode_frames(data, F, 1, True, prefix=pre); te = time.perf_counter() - t
    c, d = offs(frames)
    comp = comp + b"\0" * 8
    eng.decode_frames(comp, c, d, verify=True, prefix=pre)
    t = time.perf_counter(); out, st = eng.decode_frames(comp, c, d, verify=True, prefix=pre); td = time.perf_counter() - t
    print(f"prefix={'1 MiB' if pre else 'none':6s} ratio {n / len(comp):5.2f}  encode {n / 2**30 / te:6.2f} GiB/s  decode {n / 2**30 / td:6.2f} GiB/s  ok {out == data.tobytes()}", flush=True)
for policy, name in ((zk.FrameSizePolicy.Uncompressed(F), "Uncompressed(2 MiB)"), (zk.FrameSizePolicy.Compressed(1 << 20), "Compressed(1 MiB)"),
                     (zk.FrameSizePolicy.Compressed(64 << 10), "Compressed(64 KiB)")):
    for rep in range(2):
        sink = io.BytesIO()
        enc = zk.EncodeOptions().engine(eng).frame_size_policy(policy).checksum_flag(True).compression_level(1).into_encoder(sink)
        t = time.perf_counter()
        enc.write_all(data)
        enc.finish()
        te = time.perf_counter() - t
    blob = sink.getvalue()
    dec = zk.Decoder(blob)
    t = time.perf_counter(); out = dec.read_to_end(); td = time.perf_counter() - t
    print(f"{name:20s} frames {dec.seek_table().num_frames():6d} ratio {n / len(blob):5.2f}  Encoder {n / 2**30 / te:6.2f} GiB/s  Decoder {n / 2**30 / td:6.2f} GiB/s  ok {out == data.tobytes()}", flush=True)
