"""The unverified seek soak: >= 100 000 single seeks (BASELINE configs[3]'s protocol: set_offset; set_offset_limit; decompress) on
64 KiB frames WRITTEN BY LIBZSTD (one 64 KiB block of ~4 500 sequences with its own tables), levels 1 and 3, with and without
Content_Checksums, through a Decoder opened with ZK_DEC_NO_VERIFY -- no checksum anywhere between the kernels and the
comparison with the archive's bytes, which is how zeekstd's default archives (checksums off, lib/src/encode.rs:163-167) are read.

Why it exists: round 3 ended with one such seek in ~8 000 delivering wrong bytes with status OK (the Huffman companion wave's
exit, DESIGN.md section 8 "the small-batch defect"); every other seek test verifies checksums and so could not see it.
Reference behaviour: ZSTD_decompressStream never returns wrong bytes with status 0 (lib/src/decode.rs:242-256, 402-437)."""
import ctypes as C
import time

import numpy as np
import pytest

from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu

FRAME = 65536


def seek_protocol(trials, total, seed):
    """xorshift64* seeded through splitmix64; off = next() % total, len = 1 + next() % 8192 (bench.seek_protocol)."""
    import bench
    return bench.seek_protocol(trials, total, seed)


def soak(eng, comp, frames, data, trials, seed, verify=False, keep_bytes=False):
    """-> (list of (seek index, frame, first wrong byte of the read, number of wrong bytes[, runs of wrong bytes]), microseconds per
    seek); data: np.uint8 array"""
    import zeekstd_amd as zk
    from zeekstd_amd import api
    lib = zk.lib
    lib.zk_decoder_open_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.zk_decoder_time_seeks.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.zk_decoder_time_seeks.restype = C.c_int
    lib.zk_decoder_free.argtypes = [C.c_void_p]
    st = zk.SeekTable.new()
    for c_, d_ in frames:
        st.log_frame(c_, d_)
    seekable = comp + st.to_bytes()
    o = api.zk_decode_opts()
    h = C.c_void_p()
    o.flags = 0 if verify else 16                            # ZK_DEC_NO_VERIFY
    assert lib.zk_decoder_open_bytes(eng._h, seekable, len(seekable), C.byref(o), C.byref(h)) == 0
    offs, lens = seek_protocol(trials, len(data), seed)
    buf = np.zeros(8192 + 64, np.uint8)
    us = np.zeros(trials, np.float64)
    fails = []
    lo = 0
    while lo < trials:
        cnt = min(4000, trials - lo)
        rc = lib.zk_decoder_time_seeks(h, offs[lo:].ctypes.data, lens[lo:].ctypes.data, cnt, buf.ctypes.data, buf.size,
                                       data.ctypes.data, us[lo:].ctypes.data)
        if rc == 0:
            lo += cnt
            continue
        neg = np.nonzero(us[lo:lo + cnt] < 0)[0]
        assert len(neg) == 1, (rc, neg)                      # a decoder error (not a mismatch) would leave no mark
        i = lo + int(neg[0])
        n = int(lens[i])
        diff = np.nonzero(buf[:n] != data[int(offs[i]):int(offs[i]) + n])[0]
        rec = (i, int(offs[i]) // FRAME, int(diff[0]) if len(diff) else -1, len(diff))
        if keep_bytes:                                       # (start, length) of every run of wrong bytes
            cuts = np.nonzero(np.diff(diff) > 1)[0] if len(diff) else []
            starts = [int(diff[0])] + [int(diff[k + 1]) for k in cuts] if len(diff) else []
            ends = [int(diff[k]) for k in cuts] + [int(diff[-1])] if len(diff) else []
            rec += ([(a, b - a + 1) for a, b in zip(starts, ends)],)
        fails.append(rec)
        lo = i + 1
    lib.zk_decoder_free(h)
    return fails, us


@pytest.fixture(scope="module")
def text():
    return np.frombuffer(zko.gen_chunks(128 << 20, 0x50A4), np.uint8)


@pytest.mark.parametrize("level,checksums,seed", [(1, True, 0x5EED0003), (1, False, 0x5EED1003), (3, True, 0x5EED2003), (3, False, 0x5EED3003)])
def test_unverified_seeks_on_libzstd_made_frames(engine, text, level, checksums, seed):
    comp, frames = Z.encode_seekable_frames(text.tobytes(), FRAME, level, checksums)
    t0 = time.time()
    fails, us = soak(engine, comp, frames, text, 26000, seed)
    assert fails == [], f"level {level} checksums {checksums}: {len(fails)} of 26000 seeks delivered wrong bytes: {fails[:8]}"
    print(f"level {level} checksums {int(checksums)}: 26000 seeks clean, p50 {np.percentile(us, 50):.0f} us, {time.time() - t0:.0f} s")


def test_unverified_seeks_on_engine_made_frames(engine, text):
    """This engine's own 64 KiB frames (sixteen 4 KiB blocks, shared tables: sixty-four Huffman streams per workgroup)."""
    comp, frames = engine.encode_frames(text, FRAME, 1, False)
    fails, us = soak(engine, comp, frames, text, 12000, 0x5EED4003)
    assert fails == [], fails[:8]
