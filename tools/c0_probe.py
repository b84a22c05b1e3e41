#!/usr/bin/env python3
"""Scratch (round 6): configs[0]'s Decoder leg (10 192 446 bytes = 5 frames, checksums off, 131 072-byte reads until 0, then reset:
lib/benches/decompress.rs:18-39) with the executor per frame and in segments; where the time goes on the host side.
    python tools/c0_probe.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import zko


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    import zeekstd_amd as zk
    from zeekstd_amd import EncodeOptions, DecodeOptions
    eng = zk.Engine(0)
    n = 10192446
    data = zko.gen_chunks(n)

    class Sink:
        def __init__(self): self.parts = []
        def write(self, b): self.parts.append(bytes(b)); return len(b)
        def flush(self): pass
    sink = Sink()
    enc = EncodeOptions().engine(eng).compression_level(1).into_encoder(sink)
    enc.write_all(data); enc.end_frame(); enc.finish()
    arch = b"".join(sink.parts)
    for name, ch in (("frame", {"exec_seg": 1}), ("by shape", {}), ("segments", {"exec_seg": 2})):
        eng.set_kernel_choice(reset=0)
        eng.set_kernel_choice(**ch)
        dec = DecodeOptions(arch).engine(eng).into_decoder()
        for bufsize in (131072, 16 << 20):
            buf = bytearray(bufsize)
            ts, first_call = [], []
            ok = True
            for r in range(reps):
                parts = []
                t = time.perf_counter()
                k = dec.decompress(buf)
                t1 = time.perf_counter()
                while k:
                    if r == 0:
                        parts.append(bytes(buf[:k]))
                    k = dec.decompress(buf)
                ts.append(time.perf_counter() - t); first_call.append(t1 - t)
                dec.reset()
                if r == 0:
                    ok = b"".join(parts) == data
            print(f"{name:9s} reads of {bufsize:9d}: best {min(ts) * 1e3:6.2f} ms = {n / 2**30 / min(ts):5.2f} GiB/s (first call {min(first_call) * 1e3:5.2f} ms) ok {ok}", flush=True)
    eng.set_kernel_choice(reset=0)


if __name__ == "__main__":
    main()
