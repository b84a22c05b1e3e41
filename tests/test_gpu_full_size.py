"""BASELINE.json configs[2] and configs[3] at their FULL sizes, as tests (round 3 reached these sizes only inside bench.py).

configs[2]  4 GiB synthetic, 2048 x 2 MiB frames, encode + decode on one MI355X, XXH64 checksums on:
            GPU encode -> a sample of 64 frames is byte-identical to the CPU twin and decodes with the box's libzstd ->
            GPU decode (one batch, then two batches in flight) -> every byte equals the input, every frame's XXH64 equals the
            oracle's hash of the generator's bytes, every Content_Checksum verified on the device.
configs[3]  the same 4 GiB as 65 536 x 64 KiB frames: the random set_offset / set_offset_limit protocol through the Decoder
            (unverified: nothing between the kernels and the comparison), and 8 192 random frames in one submission.
Size-independent properties used where the oracle would take minutes: round trip, checksum of checksums, twin identity of a
sample.  Reference: lib/src/lib.rs:82-134 (cycle), lib/src/decode.rs:402-437 (seeks), lib/benches/decompress.rs:27-41."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu

FRAME = 2 << 20
NFRAMES = 2048


@pytest.fixture(scope="module")
def big():
    """(4 GiB of the 8d generator's text as a uint8 array, XXH64 of every 2 MiB frame by the oracle) -- threads, not forks:
    HIP is already initialised in this process."""
    data = np.empty(NFRAMES * FRAME, np.uint8)
    hashes = np.zeros(NFRAMES, np.uint64)
    per = 16

    def part(i0):
        b = zko.gen_chunks(per * FRAME, 0x2000 + i0)
        data[i0 * FRAME:(i0 + per) * FRAME] = np.frombuffer(b, np.uint8)
        for i in range(per):
            hashes[i0 + i] = zko.xxh64(b[i * FRAME:(i + 1) * FRAME])
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(part, range(0, NFRAMES, per)))
    return data, hashes


def _dev(engine):
    import torch
    return torch, torch.device("cuda", 0)


def test_configs2_encode_decode_4gib(engine, big):
    import zeekstd_amd as zk
    torch, dev = _dev(engine)
    data, hashes = big
    n = data.size
    d_src = torch.from_numpy(data).to(dev)
    cap = int(zk.lib.zk_compress_bound(n, FRAME))
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.zeros(NFRAMES, dtype=torch.int32, device=dev)
    d_ds = torch.zeros(NFRAMES, dtype=torch.int32, device=dev)
    nf, csize = engine.encode_frames_dev(d_src, n, FRAME, 1, True, d_comp, cap, d_cs, d_ds)
    torch.cuda.synchronize()
    assert nf == NFRAMES
    cs = d_cs.cpu().numpy().astype(np.uint64)
    assert (d_ds.cpu().numpy() == FRAME).all() and int(cs.sum()) == csize
    c = np.zeros(NFRAMES + 1, np.uint64); c[1:] = np.cumsum(cs)
    d = np.arange(NFRAMES + 1, dtype=np.uint64) * FRAME
    # a sample of 64 frames: the CPU twin's bytes, and the box's libzstd takes them back to the input
    sample = sorted(set([0, 1, NFRAMES - 1] + list(range(7, NFRAMES, NFRAMES // 61))))[:64]

    def check(f):
        fr = bytes(d_comp[int(c[f]):int(c[f + 1])].cpu().numpy())
        want = data[f * FRAME:(f + 1) * FRAME].tobytes()
        assert fr == zko.frame_encode(want, 1, True), f
        if Z.load("system") is not None:
            assert Z.decode_stream(fr, FRAME, "system") == want, f
        return True
    for f in sample:                                     # (serially: the twin keeps its level settings and code tables in globals)
        check(f)
    # decode: one batch, every checksum verified on the device
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_st = torch.full((NFRAMES,), -1, dtype=torch.int32, device=dev)
    assert engine.decode_frames_dev(d_comp, csize, d_c, d_d, 0, NFRAMES, d_out, n, True, d_st) == 0
    assert int(d_st.abs().sum().item()) == 0
    assert torch.equal(d_out[:n], d_src)
    # a synchronous decode of this size has its checksums computed beside the executor (zk_k_xxh64_follow): all 2048 frames on every
    # box so far; how many is the dispatcher's habit, and what the waves leave is verified behind the executor
    assert 1 <= engine.checksums_followed() <= NFRAMES, engine.checksums_followed()
    d_hash = torch.zeros(NFRAMES, dtype=torch.int64, device=dev)
    engine.xxh64_frames_dev(d_out, d_d, NFRAMES, d_hash)
    assert np.array_equal(d_hash.cpu().numpy().view(np.uint64), hashes)          # the oracle's XXH64 of the generator's bytes
    # two batches in flight (what bench.py times), into fresh buffers
    outs = [torch.zeros(n + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
    sts = [torch.full((NFRAMES,), -1, dtype=torch.int32, device=dev) for _ in range(2)]
    slots = [engine.decode_submit_dev(d_comp, csize, d_c, d_d, 0, NFRAMES, outs[i], n, True, sts[i]) for i in range(2)]
    for s in slots:
        assert engine.decode_wait(s) == 0
    for i in range(2):
        assert int(sts[i].abs().sum().item()) == 0 and torch.equal(outs[i][:n], d_src)
    # a damaged frame in the middle of the full batch is reported, its neighbours are not
    f = 1234
    at = int(c[f + 1]) - 1
    d_comp[at] = d_comp[at] ^ 1
    rc = engine.decode_frames_dev(d_comp, csize, d_c, d_d, 0, NFRAMES, d_out, n, True, d_st)
    st = d_st.cpu().numpy()
    assert rc == -22 and st[f] == 22 and not np.delete(st, f).any()
    d_comp[at] = d_comp[at] ^ 1


def test_configs2_at_the_reference_clis_default_level(engine, big):
    """(round 6) The same 4 GiB at level 3 -- the reference CLI's default (cli/src/args.rs:192), this encoder's dense far history
    (zk_k_enc_dense_part / zk_k_enc_dense_cand, 8 bytes of device scratch per input byte): a sample of frames is byte-identical to the
    CPU twin and decodes with the box's libzstd, the whole archive decodes on the device to the input with every Content_Checksum
    verified, the ratio is the twin's (2.73 on this text), and the dense kernels ran (their time is reported)."""
    import zeekstd_amd as zk
    torch, dev = _dev(engine)
    data, hashes = big
    n = data.size
    d_src = torch.from_numpy(data).to(dev)
    cap = int(zk.lib.zk_compress_bound(n, FRAME))
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.zeros(NFRAMES, dtype=torch.int32, device=dev)
    d_ds = torch.zeros(NFRAMES, dtype=torch.int32, device=dev)
    engine.set_profiling(True)
    nf, csize = engine.encode_frames_dev(d_src, n, FRAME, 3, True, d_comp, cap, d_cs, d_ds)
    times = engine.kernel_times()
    engine.set_profiling(False)
    torch.cuda.synchronize()
    assert nf == NFRAMES and times.get("zk_k_enc_dense_cand", 0) > 0, times
    cs = d_cs.cpu().numpy().astype(np.uint64)
    assert int(cs.sum()) == csize and 2.70 < n / csize < 2.77, n / csize
    c = np.zeros(NFRAMES + 1, np.uint64); c[1:] = np.cumsum(cs)
    d = np.arange(NFRAMES + 1, dtype=np.uint64) * FRAME
    for f in (0, 1, 777, NFRAMES - 1):
        fr = bytes(d_comp[int(c[f]):int(c[f + 1])].cpu().numpy())
        want = data[f * FRAME:(f + 1) * FRAME].tobytes()
        assert fr == zko.frame_encode(want, 3, True), f
        if Z.load("system") is not None:
            assert Z.decode_stream(fr, FRAME, "system") == want, f
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_st = torch.full((NFRAMES,), -1, dtype=torch.int32, device=dev)
    assert engine.decode_frames_dev(d_comp, csize, d_c, d_d, 0, NFRAMES, d_out, n, True, d_st) == 0
    assert int(d_st.abs().sum().item()) == 0 and torch.equal(d_out[:n], d_src)
    d_hash = torch.zeros(NFRAMES, dtype=torch.int64, device=dev)
    engine.xxh64_frames_dev(d_out, d_d, NFRAMES, d_hash)
    assert np.array_equal(d_hash.cpu().numpy().view(np.uint64), hashes)
    # level 2 of the same engine runs without the dense kernels
    engine.set_profiling(True)
    engine.encode_frames_dev(d_src[:64 * FRAME], 64 * FRAME, FRAME, 2, True, d_comp, cap, d_cs, d_ds)
    t2 = engine.kernel_times()
    engine.set_profiling(False)
    assert not t2.get("zk_k_enc_dense_cand", 0), t2


def test_configs3_seeks_over_65536_frames(engine, big):
    import zeekstd_amd as zk
    from zeekstd_amd import api
    import bench
    torch, dev = _dev(engine)
    data, _ = big
    n = data.size
    fs = 65536
    nf = n // fs
    d_src = torch.from_numpy(data).to(dev)
    cap = int(zk.lib.zk_compress_bound(n, fs))
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.zeros(nf, dtype=torch.int32, device=dev)
    d_ds = torch.zeros(nf, dtype=torch.int32, device=dev)
    got_nf, csize = engine.encode_frames_dev(d_src, n, fs, 1, True, d_comp, cap, d_cs, d_ds)
    assert got_nf == nf == 65536
    cs = d_cs.cpu().numpy().astype(np.uint64)
    c = np.zeros(nf + 1, np.uint64); c[1:] = np.cumsum(cs)
    d = np.arange(nf + 1, dtype=np.uint64) * fs
    # 8 192 random frames in ONE submission (zk_decode_frame_list_dev), packed output compared on the device
    rng = np.random.default_rng(0xC3)
    ids = rng.integers(0, nf, 8192).astype(np.uint32)
    ooff = np.arange(len(ids) + 1, dtype=np.uint64) * fs
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_ids = torch.from_numpy(ids.view(np.int32)).to(dev); d_oo = torch.from_numpy(ooff.view(np.int64)).to(dev)
    d_o = torch.zeros(len(ids) * fs + 64, dtype=torch.uint8, device=dev)
    d_s = torch.full((len(ids),), -1, dtype=torch.int32, device=dev)
    assert engine.decode_frame_list_dev(d_comp, csize, d_c, d_d, d_ids, d_oo, len(ids), d_o, len(ids) * fs, True, d_s) == 0
    assert int(d_s.abs().sum().item()) == 0
    want = d_src.view(nf, fs)[torch.from_numpy(ids.astype(np.int64)).to(dev)].reshape(-1)
    assert torch.equal(d_o[:len(ids) * fs], want)
    # the seek protocol as written (10 000 seeks, seed of the bench), single seeks through the Decoder, unverified
    comp = bytes(d_comp[:csize].cpu().numpy())
    st = zk.SeekTable.new()
    for i in range(nf):
        st.log_frame(int(cs[i]), fs)
    seekable = comp + st.to_bytes()
    lib = zk.lib
    lib.zk_decoder_open_bytes.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.zk_decoder_time_seeks.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.zk_decoder_time_seeks.restype = C.c_int
    lib.zk_decoder_free.argtypes = [C.c_void_p]
    o = api.zk_decode_opts(); h = C.c_void_p()
    o.flags = 16
    assert lib.zk_decoder_open_bytes(engine._h, seekable, len(seekable), C.byref(o), C.byref(h)) == 0
    offs, lens = bench.seek_protocol(10000, n)
    buf = np.zeros(8192 + 64, np.uint8); us = np.zeros(len(offs), np.float64)
    rc = lib.zk_decoder_time_seeks(h, offs.ctypes.data, lens.ctypes.data, len(offs), buf.ctypes.data, buf.size, data.ctypes.data, us.ctypes.data)
    lib.zk_decoder_free(h)
    assert rc == 0, (rc, np.nonzero(us < 0)[0][:4])
