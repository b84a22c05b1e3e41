#!/usr/bin/env python3
"""End-to-end (host buffers in, host buffers out) throughput of the zeekstd API on the GPU engine: zk_encoder_compress +
finish into a buffer writer, zk_decoder_decompress into caller memory; pageable and pinned (zk_host_alloc) caller buffers."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    fsz = int(sys.argv[2]) if len(sys.argv) > 2 else bench.FRAME
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cores = os.cpu_count() or 8
    data, _, _, _ = bench.build_inputs(0, nframes, 1, True, max(1, min(64, cores - 1)), False, 0)
    import zeekstd_amd as zk
    from zeekstd_amd import api
    lib = zk.lib
    eng = zk.Engine(0)
    if threads:
        lib.zk_engine_set_host_threads(eng._h, threads)
    n = nframes * bench.FRAME
    src = np.ascontiguousarray(data[:n])

    class BW(C.Structure):
        _fields_ = [("data", C.c_void_p), ("cap", C.c_uint64), ("len", C.c_uint64), ("engine", C.c_void_p)]
    cap = int(lib.zk_compress_bound(n, fsz)) + (nframes * (bench.FRAME // fsz) * 8 + 64)
    lib.zk_buffer_writer_write.restype = C.c_int
    wfn = C.cast(lib.zk_buffer_writer_write, C.c_void_p)

    def encode(src_ptr, out_ptr, label):
        bw = BW(out_ptr, cap, 0, eng._h)
        o = api.zk_encode_opts(0, fsz, 1, 1, 0)
        h = C.c_void_p()
        lib.zk_encoder_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        assert lib.zk_encoder_new(eng._h, C.byref(o), wfn, C.byref(bw), C.byref(h)) == 0
        lib.zk_encoder_compress.restype = C.c_int64
        lib.zk_encoder_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        t = time.perf_counter()
        r = lib.zk_encoder_compress(h, src_ptr, n)
        assert r == n, r
        total = C.c_uint64()
        lib.zk_encoder_finish.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        assert lib.zk_encoder_finish(h, 1, C.byref(total)) == 0
        dt = time.perf_counter() - t
        lib.zk_encoder_free.argtypes = [C.c_void_p]
        lib.zk_encoder_free(h)
        print(f"encode {label}: {n / dt / 2**30:.2f} GiB/s ({dt * 1e3:.1f} ms), {total.value} bytes", flush=True)
        return total.value

    def decode(arch_ptr, arch_len, out_ptr, label):
        o = api.zk_decode_opts()
        h = C.c_void_p()
        lib.zk_decoder_open_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        assert lib.zk_decoder_open_bytes(eng._h, arch_ptr, arch_len, C.byref(o), C.byref(h)) == 0
        lib.zk_decoder_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.zk_decoder_decompress.restype = C.c_int64
        t = time.perf_counter()
        got = 0
        while True:
            r = lib.zk_decoder_decompress(h, out_ptr + got, n - got)
            assert r >= 0, r
            if r == 0:
                break
            got += r
            if got == n:
                break
        dt = time.perf_counter() - t
        lib.zk_decoder_free.argtypes = [C.c_void_p]
        lib.zk_decoder_free(h)
        assert got == n
        print(f"decode {label}: {n / dt / 2**30:.2f} GiB/s ({dt * 1e3:.1f} ms)", flush=True)

    out_pg = np.empty(cap, np.uint8)
    out_pg[:] = 0                                   # touch the pages
    dec_pg = np.empty(n, np.uint8); dec_pg[:] = 0
    for rep in range(3):
        total = encode(src.ctypes.data, out_pg.ctypes.data, "pageable")
    for rep in range(3):
        decode(out_pg.ctypes.data, total, dec_pg.ctypes.data, "pageable")
    assert np.array_equal(dec_pg, src), "round trip mismatch"
    # pinned caller buffers
    lib.zk_host_alloc.restype = C.c_void_p
    t = time.perf_counter()
    p_src = lib.zk_host_alloc(n); p_out = lib.zk_host_alloc(cap); p_dec = lib.zk_host_alloc(n)
    print(f"zk_host_alloc of {(2 * n + cap) >> 20} MiB: {time.perf_counter() - t:.2f} s")
    C.memmove(p_src, src.ctypes.data, n)
    for rep in range(3):
        total = encode(p_src, p_out, "pinned")
    for rep in range(3):
        decode(p_out, total, p_dec, "pinned")
    got = np.ctypeslib.as_array(C.cast(p_dec, C.POINTER(C.c_uint8)), shape=(n,))
    assert np.array_equal(got, src), "pinned round trip mismatch"
    print("ok")


if __name__ == "__main__":
    main()
