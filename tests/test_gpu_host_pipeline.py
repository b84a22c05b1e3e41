"""The host-pointer pipeline (zk_engine_host.hip): pinned staging rings, chunked H2D / decode / D2H overlap, caller-pinned
buffers, callback sources, and the error semantics of the reference's streaming Decoder (lib/src/decode.rs:201-270: every
byte in front of a damaged frame is delivered first).  Parity is against the generator bytes / the system libzstd."""
import ctypes as C
import os
import io

import numpy as np
import pytest

import zeekstd_amd as zk
from zeekstd_amd import DecodeOptions, EncodeOptions, Format, FrameSizePolicy, SeekTable
from conftest import PREFIX_GOLDENS, offsets_from_frames
from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu


def _seekable(comp, frames):
    st = SeekTable.new()
    for c, d in frames:
        st.log_frame(c, d)
    return comp + st.to_bytes()


@pytest.fixture(scope="module")
def big(engine):
    """96 MiB of generator text in 1 MiB frames, encoded by the engine: several pipeline chunks (16 MiB each)."""
    data = zko.gen_chunks(96 << 20, 300)
    comp, frames = engine.encode_frames(np.frombuffer(data, np.uint8), 1 << 20, 1, True)
    return data, comp, frames


def test_multi_chunk_host_roundtrip(engine, big):
    data, comp, frames = big
    assert len(frames) == 96
    c, d = offsets_from_frames(frames)
    out, st = engine.decode_frames(comp + b"\0" * 8, c, d, verify=True)
    assert not st.any() and out == data
    # a frame range in the middle, chunked too
    out, st = engine.decode_frames(comp + b"\0" * 8, c, d, first=17, count=60)
    assert out == data[17 << 20:77 << 20]
    if Z.load("system") is not None:            # libzstd accepts every frame the chunked encode produced
        assert Z.decode_stream(comp, len(data), "system") == data


def test_pinned_caller_buffers(engine, big):
    """zk_host_alloc memory is moved by DMA directly (no staging rings): same bytes."""
    data, comp, frames = big
    c, d = offsets_from_frames(frames)
    lib = zk.lib
    n_c, n_d = len(comp) + 64, len(data)
    p_c, p_d = lib.zk_host_alloc(n_c), lib.zk_host_alloc(n_d)
    assert p_c and p_d
    try:
        C.memmove(p_c, comp, len(comp))
        st = np.zeros(len(frames), np.int32)
        rc = lib.zk_decode_frames(engine._h, p_c, len(comp), c.ctypes.data, d.ctypes.data, 0, len(frames), p_d, n_d, 1, st.ctypes.data)
        assert rc == 0 and not st.any()
        assert C.string_at(p_d, n_d) == data
        # encode from / into pinned memory
        cap = int(lib.zk_compress_bound(n_d, 1 << 20))
        p_o = lib.zk_host_alloc(cap)
        cs = np.zeros(len(frames), np.uint32); ds = np.zeros(len(frames), np.uint32)
        nf = C.c_uint32(); wr = C.c_uint64()
        rc = lib.zk_encode_frames(engine._h, p_d, n_d, 1 << 20, 1, 1, p_o, cap, cs.ctypes.data, ds.ctypes.data, len(frames), C.byref(nf), C.byref(wr))
        assert rc == 0 and nf.value == len(frames)
        assert C.string_at(p_o, wr.value) == comp          # same bytes as from pageable memory
        lib.zk_host_free(p_o)
    finally:
        lib.zk_host_free(p_c); lib.zk_host_free(p_d)


def test_decoder_direct_and_cached_reads(engine, big):
    """One big read goes frame-direct into the caller's buffer; small reads go through the cache; mixed sizes agree."""
    data, comp, frames = big
    seekable = _seekable(comp, frames)
    d = DecodeOptions(seekable).engine(engine).into_decoder()
    out = bytearray(len(data))
    assert d.decompress(out) == len(data) and bytes(out) == data
    assert d.decompress(out) == 0
    # unaligned start + limit: head and tail frames through the cache, the middle direct
    d.set_offset(3 * (1 << 20) + 12345)
    d.set_offset_limit(90 * (1 << 20) + 777)
    n = 90 * (1 << 20) + 777 - (3 * (1 << 20) + 12345)
    got = bytearray()
    buf = bytearray(40 << 20)
    while True:
        k = d.decompress(buf)
        if k == 0:
            break
        got += buf[:k]
    assert len(got) == n and bytes(got) == data[3 * (1 << 20) + 12345:90 * (1 << 20) + 777]
    # streaming reads of an odd size
    d.reset()
    got = bytearray()
    buf = bytearray(1000003)
    while True:
        k = d.decompress(buf)
        if k == 0:
            break
        got += buf[:k]
    assert bytes(got) == data


def test_callback_source(engine, big):
    """A Read + Seek object behind zk_decoder_open_callbacks (the Seekable trait, seekable.rs:16-39)."""
    data, comp, frames = big
    f = io.BytesIO(_seekable(comp, frames))
    d = DecodeOptions(f).engine(engine).into_decoder()
    assert d.seek_table().num_frames() == len(frames)
    d.set_offset(5 << 20)
    d.set_offset_limit((5 << 20) + 70000)
    buf = bytearray(1 << 20)
    assert d.decompress(buf) == 70000 and bytes(buf[:70000]) == data[5 << 20:(5 << 20) + 70000]
    d.reset()
    out = bytearray(len(data))
    assert d.decompress(out) == len(data) and bytes(out) == data


def test_bytes_before_a_damaged_frame_are_delivered(engine):
    """decode.rs:221-267: reads in front of a corrupt frame succeed although the engine decodes ahead (read-ahead never fails
    a read that does not reach the damaged frame); the read that reaches it fails, as the `?` at decode.rs:242-245 does."""
    data = zko.gen_text(40 * 3000, 99)
    comp, frames = engine.encode_frames(np.frombuffer(data, np.uint8), 3000, 1, True)
    c, _ = offsets_from_frames(frames)
    bad = bytearray(comp)
    at = int(c[20]) + (int(c[21]) - int(c[20])) // 2
    bad[at] ^= 0x5A                                        # frame 20 is damaged
    for batch in (1 << 26, 4096):
        for step, expect in ((3000, 60000), (7000, 56000), (1, 60000) if batch == 4096 else (2500, 60000)):
            d = DecodeOptions(_seekable(bytes(bad), frames)).engine(engine).batch_bytes(batch).into_decoder()
            got = bytearray()
            buf = bytearray(step)
            with pytest.raises(zk.Error):
                for _ in range(100000):
                    k = d.decompress(buf)
                    assert k > 0
                    got += buf[:k]
            assert bytes(got) == data[:expect], (batch, step)   # every read that ends in front of frame 20 succeeded
            with pytest.raises(zk.Error):                  # and the retry fails the same way
                d.decompress(buf)
    # one large read spans the damaged frame: it fails like the reference's single call does
    d = DecodeOptions(_seekable(bytes(bad), frames)).engine(engine).batch_bytes(8).into_decoder()
    with pytest.raises(zk.Error):
        d.decompress(bytearray(len(data)))
    # frames behind the damaged one are still reachable
    d.set_offset(21 * 3000)
    out = bytearray(19 * 3000)
    assert d.decompress(out) == len(out) and bytes(out) == data[21 * 3000:]


def test_crafted_seek_table_cannot_force_allocations(engine):
    """A seek table is untrusted: a frame that claims more than 32768 x its compressed size (or > 1 GiB) is refused."""
    data = zko.gen_text(5000, 3)
    comp, frames = engine.encode_frames(np.frombuffer(data, np.uint8), 5000, 1, False)
    for claimed in (0xFFFFFFFF, 0x40000001, len(comp) * 32768 + 1):
        st = SeekTable.new()
        st.log_frame(frames[0][0], claimed)
        d = DecodeOptions(comp).engine(engine).seek_table(st).into_decoder()
        with pytest.raises(zk.Error):
            d.decompress(bytearray(1))


def test_prefix_is_not_cached_by_address(engine):
    """two equal-length prefixes that differ in one byte in the middle, in the SAME buffer, must both decode
    correctly (the staged device copy may not be reused by address + sampled fingerprint)."""
    g = next(x for x in PREFIX_GOLDENS if x.meta["prefix_len"] >= (1 << 20))
    pre = g.prefix()
    data = g.input()
    c, d = g.offsets()
    buf = np.frombuffer(bytearray(pre), np.uint8)          # one host buffer reused for both bases
    comp = np.frombuffer(g.comp + b"\0" * 8, np.uint8)
    out = np.empty(len(data), np.uint8)
    st = np.zeros(len(g.frames), np.int32)

    def run():
        return zk.lib.zk_decode_frames_prefix(engine._h, comp.ctypes.data, len(g.comp), c.ctypes.data, d.ctypes.data, 0, len(g.frames),
                                              buf.ctypes.data, buf.size, out.ctypes.data, out.size, 0, st.ctypes.data)
    assert run() == 0 and out.tobytes() == data
    # find a prefix byte that the frames really copy from, and flip it in place
    ref = out.tobytes()
    for at in range(5000, len(pre), 4099):
        buf[at] ^= 0xFF
        rc = run()
        changed = out.tobytes() != ref
        buf[at] ^= 0xFF
        if changed or rc != 0:
            break
    else:
        pytest.skip("no referenced prefix byte found")
    assert run() == 0 and out.tobytes() == data            # and back again: the restored prefix is used, not the stale copy


def test_offset_arrays_are_validated(engine):
    data = zko.gen_text(6000, 8)
    comp, frames = engine.encode_frames(np.frombuffer(data, np.uint8), 1000, 1, False)
    c, d = offsets_from_frames(frames)
    buf = np.frombuffer(comp + b"\0" * 8, np.uint8)
    out = np.empty(len(data), np.uint8)
    st = np.zeros(len(frames), np.int32)
    lib = zk.lib
    bad = c.copy(); bad[3] = bad[5] + 7                   # non-monotone entry in the middle
    assert lib.zk_decode_frames(engine._h, buf.ctypes.data, len(comp), bad.ctypes.data, d.ctypes.data, 0, len(frames), out.ctypes.data, out.size, 1, st.ctypes.data) == -2003
    bad = c.copy(); bad[-1] = len(comp) + 100             # beyond comp_size
    assert lib.zk_decode_frames(engine._h, buf.ctypes.data, len(comp), bad.ctypes.data, d.ctypes.data, 0, len(frames), out.ctypes.data, out.size, 1, st.ctypes.data) == -2003
    # device-pointer entry points: comp_size / dst_cap are enforced per frame on the device
    import torch
    dev = torch.device("cuda", 0)
    d_comp = torch.from_numpy(buf.copy()).to(dev)
    d_c = torch.from_numpy(c.view(np.int64)).to(dev); d_d = torch.from_numpy(d.view(np.int64)).to(dev)
    d_out = torch.zeros(len(data) + 64, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(len(frames), dtype=torch.int32, device=dev)
    rc = engine.decode_frames_dev(d_comp, int(c[4]) + 3, d_c, d_d, 0, len(frames), d_out, len(data), True, d_st)     # comp_size cuts frame 4
    s = d_st.cpu().numpy()
    assert rc == -72 and not s[:4].any() and (s[4:] == 72).all()
    assert bytes(d_out[:4000].cpu().numpy()) == data[:4000]
    rc = engine.decode_frames_dev(d_comp, len(comp), d_c, d_d, 0, len(frames), d_out, 2500, True, d_st)              # dst_cap cuts frame 2
    s = d_st.cpu().numpy()
    assert rc == -70 and not s[:2].any() and (s[2:] == 70).all()


def test_limit_raised_after_a_cut_frame_verifies_it(engine):
    """a frame cut by offset_limit is not verified (decode.rs:425-427); once the limit is raised past its end the
    rest of it must not be served unverified."""
    data = zko.gen_text(3 * 4000, 5)
    comp, frames = engine.encode_frames(np.frombuffer(data, np.uint8), 4000, 1, True)
    c, _ = offsets_from_frames(frames)
    bad = bytearray(comp)
    bad[int(c[2]) - 1] ^= 0x01                             # frame 1's stored checksum is wrong; its content decodes
    d = DecodeOptions(_seekable(bytes(bad), frames)).engine(engine).into_decoder()
    d.set_offset(4000)
    d.set_offset_limit(4100)
    buf = bytearray(8000)
    assert d.decompress(buf) == 100 and bytes(buf[:100]) == data[4000:4100]
    d.set_offset_limit(12000)
    with pytest.raises(zk.Error):
        d.decompress(buf)


def test_encoder_large_write_direct(engine, big):
    """One large write: whole frames are encoded from the caller's buffer, the last stays open until finish."""
    data, comp, frames = big
    w = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(1 << 20)).checksum_flag(True).compression_level(1).into_encoder(w)
    cut = 50 * (1 << 20) + 4321
    assert enc.compress(data[:cut]) == cut
    assert enc.compress(data[cut:]) == len(data) - cut
    total = enc.finish()
    sink = w.getvalue()
    assert total == len(sink)
    st = SeekTable.from_seekable(bytes(sink))
    assert st.num_frames() == 96 and st.size_decomp() == len(data)
    assert bytes(sink[:len(comp)]) == comp                 # same frames as the batch call
    out = bytearray(len(data))
    assert DecodeOptions(bytes(sink)).engine(engine).into_decoder().decompress(out) == len(data) and bytes(out) == data


def test_encode_sharded_under_nccl_world1(engine, big):
    """parallel.encode_sharded (the function every GPU rank runs) under a real RCCL process group of one rank: encode on the
    device, gather, table appended; the archive decodes to the input."""
    import socket
    import torch
    import torch.distributed as dist
    from zeekstd_amd import parallel
    data, _, _ = big
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        d_src = torch.frombuffer(bytearray(data[:32 << 20]), dtype=torch.uint8).to(dev)
        out, table = parallel.encode_sharded(engine, d_src, 1 << 20, 1, True, root=0)
        torch.cuda.synchronize()
        stream = bytes(out.cpu().numpy())
    finally:
        dist.destroy_process_group()
    assert table.num_frames() == 32 and table.size_decomp() == 32 << 20
    d = DecodeOptions(stream).engine(engine).into_decoder()
    got = bytearray(32 << 20)
    assert d.decompress(got) == len(got) and bytes(got) == data[:32 << 20]


def test_c_abi_gather_world1(engine, big):
    """zk_gather_seekable (the sharded path for hosts without torch) with a one-rank RCCL communicator made through librccl."""
    import torch
    data, comp, frames = big
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        try:
            rccl = C.CDLL("/opt/rocm/lib/librccl.so.1")
        except OSError:
            pytest.skip("librccl.so.1 not loadable")
    uid = (C.c_char * 128)()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0

    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    u = Uid(); C.memmove(C.byref(u), uid, 128)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0
    dev = torch.device("cuda", 0)
    d_pay = torch.frombuffer(bytearray(comp), dtype=torch.uint8).to(dev)
    cs = np.array([f[0] for f in frames], np.uint32); ds = np.array([f[1] for f in frames], np.uint32)
    cap = len(comp) + 8 * len(frames) + 64
    d_out = torch.zeros(cap, dtype=torch.uint8, device=dev)
    nbytes = C.c_uint64(); tab = C.c_void_p()
    lib = zk.lib
    lib.zk_gather_seekable.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.zk_gather_seekable(engine._h, comm, 0, 1, 0, d_pay.data_ptr(), len(comp), cs.ctypes.data, ds.ctypes.data, len(frames), 1,
                                d_out.data_ptr(), cap, C.byref(nbytes), C.byref(tab), None)
    assert rc == 0, zk.lib.zk_engine_last_hip_error(engine._h)
    stream = bytes(d_out[:nbytes.value].cpu().numpy())
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    assert nbytes.value == len(comp) + 8 * len(frames) + 17 and stream[:len(comp)] == comp
    st = SeekTable.from_seekable(stream)
    assert st.num_frames() == len(frames) and st.size_decomp() == len(data)
    lib.zk_seek_table_num_frames.argtypes = [C.c_void_p]
    assert lib.zk_seek_table_num_frames(tab) == len(frames)
    lib.zk_seek_table_free.argtypes = [C.c_void_p]
    lib.zk_seek_table_free(tab)


def test_source_failure_after_the_first_chunk_is_reported_not_unwound(engine, big):
    """a Seekable that fails in the middle (a file error, a host callback returning < 0) used to throw through the
    engine's pipeline with DMA, kernels and worker tasks in flight on that frame.  The failure is parked, the pipeline fails
    the chunk and drains, then the error surfaces -- and the decoder and the engine stay usable."""
    data, comp, frames = big
    arch = _seekable(comp, frames)

    class Flaky(io.BytesIO):
        budget = 3 << 20                                       # payload bytes delivered before read starts to fail
        armed = False

        def read(self, n=-1):
            if self.armed and self.tell() < len(comp):
                if self.budget <= 0:
                    raise OSError("disk on fire")
                self.budget -= n
            return super().read(n)

    f = Flaky(arch)
    d = DecodeOptions(f).engine(engine).batch_bytes(64 << 20).into_decoder()
    f.armed = True
    out = bytearray(len(data))
    with pytest.raises(zk.Error):
        d.decompress(out)
    f.armed = False                                            # the source recovers: the same decoder goes on
    d.reset()
    assert d.decompress(out) == len(data) and bytes(out) == data
    out2, st = engine.decode_frames(comp + b"\0" * 8, *offsets_from_frames(frames), verify=True)
    assert out2 == data and not st.any()


def test_seekable_with_its_own_integrity_field(engine):
    """Seekable::seek_table_integrity is a REQUIRED method of the trait (seekable.rs:33-38): zk_decoder_open_seekable asks the
    source.  Here the source keeps the 9-byte field apart from the stream (its tail is zeroed), which the two-callback form
    -- reading the field at End(-9) like the reference's own impls -- could not serve."""
    data = zko.gen_text(90000, 12)
    comp, frames = engine.encode_frames(np.frombuffer(data, np.uint8), 10000, 1, True)
    arch = bytearray(_seekable(comp, frames))
    field = bytes(arch[-9:])
    arch[-9:] = bytes(9)

    class Apart(io.BytesIO):
        asked = []

        def seek_table_integrity(self, fmt):
            self.asked.append(fmt)
            return field

    f = Apart(bytes(arch))
    d = DecodeOptions(f).engine(engine).into_decoder()
    assert f.asked == [Format.Foot] and d.seek_table().num_frames() == len(frames)
    out = bytearray(len(data))
    assert d.decompress(out) == len(data) and bytes(out) == data
    with pytest.raises(zk.Error):                              # without the hook the zeroed tail is what gets parsed
        DecodeOptions(io.BytesIO(bytes(arch))).engine(engine).into_decoder()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_decode_shard_at_the_c_abi(engine, world):
    """SURVEY 8e, decode side, as a host without torch calls it: every rank takes ONLY the compressed bytes of its contiguous
    frame range (zk_shard_range) and zk_decode_shard turns them into its slice of the output; no collective."""
    import ctypes as C
    from zeekstd_amd import parallel
    data = zko.gen_text(11 * 7000 + 1234, 5)
    comp, frames = engine.encode_frames(np.frombuffer(data, np.uint8), 7000, 1, True)
    st = SeekTable.new()
    for c, d in frames:
        st.log_frame(c, d)
    c_off, d_off = offsets_from_frames(frames)
    got = bytearray()
    for rank in range(world):
        lo, hi = parallel.shard_range(len(frames), rank, world)
        first, count = C.c_uint32(), C.c_uint32()
        assert zk.lib.zk_shard_range(len(frames), rank, world, C.byref(first), C.byref(count)) == 0
        assert (first.value, count.value) == (lo, hi - lo)
        f, n, out = parallel.decode_shard(engine, comp[int(c_off[lo]):int(c_off[hi])], st, rank, world)
        assert (f, n) == (lo, hi - lo) and out == data[int(d_off[lo]):int(d_off[hi])]
        got += out
    assert bytes(got) == data
    # a shard that does not hold its frames is refused, not read past
    lo, hi = parallel.shard_range(len(frames), 0, 2)
    with pytest.raises(zk.Error):
        parallel.decode_shard(engine, comp[:int(c_off[hi]) - 1], st, 0, 2)
