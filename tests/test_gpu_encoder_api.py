"""zeekstd::RawEncoder / Encoder semantics on the GPU engine -- the reference's encode + round-trip tests
restated: lib/src/encode.rs:802-871, lib/src/lib.rs:69-358, README doctests (encode.rs:220-265,
decode.rs:520-542), fuzz/fuzz_targets/roundtrip_basic.rs."""
import io

import numpy as np
import pytest

import zeekstd_amd as zk
from zeekstd_amd import (DecodeOptions, Decoder, EncodeOptions, Encoder, Format, FrameSizePolicy, RawEncoder, SeekFrom,
                         SeekTable)
from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu

INPUT = zko.gen_text(12345, 4242)


def raw_roundtrip(eng, policy=None, checksum=False, scratch=None):
    """lib.rs:82-134: raw API with a small scratch buffer so that every call is partial."""
    opts = EncodeOptions().engine(eng).checksum_flag(checksum)
    if policy:
        opts.frame_size_policy(policy)
    enc = opts.into_raw_encoder()
    buf = bytearray(scratch or max(1, len(INPUT) // 500))
    seekable = bytearray()
    in_prog = 0
    while in_prog < len(INPUT):
        p = enc.compress(INPUT[in_prog:], buf)
        seekable += buf[:p.out_progress()]
        in_prog += p.in_progress()
    while True:
        p = enc.end_frame(buf)
        seekable += buf[:p.out_progress()]
        if p.data_left() == 0:
            break
    st = enc.into_seek_table()
    assert st.size_comp() == len(seekable) and st.size_decomp() == len(INPUT)
    ser = st.into_serializer()
    while True:
        n = ser.write_into(buf)
        if n == 0:
            break
        seekable += buf[:n]
    return bytes(seekable), st


def test_raw_roundtrip_default_policy(engine):
    seekable, st = raw_roundtrip(engine)
    assert st.num_frames() == 1
    d = DecodeOptions(seekable).engine(engine).into_decoder()
    assert d.read_to_end() == INPUT
    if Z.load("system") is not None:       # payload (without the skippable seek table) decodes under the real libzstd
        assert Z.decode_stream(seekable, len(INPUT), "system") == INPUT


@pytest.mark.parametrize("fs", [1, 7, 100, 1000, 1023, 4096])     # proptest lib.rs:315-357 (1..1023)
def test_raw_roundtrip_frame_sizes(engine, fs):
    seekable, st = raw_roundtrip(engine, FrameSizePolicy.Uncompressed(fs), checksum=bool(fs & 1), scratch=97)
    assert st.num_frames() == -(-len(INPUT) // fs)
    assert st.max_frame_size_decomp() == min(fs, len(INPUT))
    d = DecodeOptions(seekable).engine(engine).into_decoder()
    assert d.read_to_end() == INPUT


def test_compress_never_closes_and_compresses_in_one_call(engine):
    # encode.rs:317-353: a call either closes a completed frame (consuming no input) or compresses
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(100)).into_raw_encoder()
    out = bytearray(4096)
    p = enc.compress(INPUT[:250], out)
    assert (p.in_progress(), p.out_progress()) == (100, 0)         # limited to the frame
    p = enc.compress(INPUT[100:250], out)
    assert p.in_progress() == 0 and p.out_progress() > 0           # frame complete: closed, no input consumed
    assert enc.seek_table().num_frames() == 1
    p = enc.compress(INPUT[100:250], out)
    assert p.in_progress() == 100


def test_checksum_flag_sets_header_bit(engine):                    # encode.rs:834-870
    for flag in (False, True):
        seekable, st = raw_roundtrip(engine, FrameSizePolicy.Uncompressed(len(INPUT) // 3), checksum=flag, scratch=5000)
        for i in range(st.num_frames()):
            assert bool(seekable[st.frame_start_comp(i) + 4] & 0x04) == flag


def test_reset_frame_and_seek_table(engine):                       # encode.rs:811-831
    enc = EncodeOptions().engine(engine).into_raw_encoder()
    out = bytearray(1 << 16)
    enc.compress(INPUT[:1000], out)
    enc.reset_frame()                                              # in-flight frame discarded
    enc.compress(INPUT, out)
    a = bytearray()
    while True:
        p = enc.end_frame(out); a += out[:p.out_progress()]
        if p.data_left() == 0:
            break
    assert enc.seek_table().num_frames() == 1 and enc.seek_table().size_decomp() == len(INPUT)
    enc.reset_seek_table()
    assert enc.seek_table().num_frames() == 0
    enc.compress(INPUT, out)
    b = bytearray()
    while True:
        p = enc.end_frame(out); b += out[:p.out_progress()]
        if p.data_left() == 0:
            break
    assert bytes(a) == bytes(b)                                    # reproducible


def test_end_frame_without_input_is_the_golden_empty_frame(engine):
    enc = EncodeOptions().engine(engine).into_raw_encoder()
    out = bytearray(64)
    p = enc.end_frame(out)
    assert bytes(out[:p.out_progress()]) == bytes.fromhex("28b52ffd2000010000") and p.data_left() == 0
    assert enc.seek_table().num_frames() == 1 and enc.seek_table().frame_size_decomp(0) == 0


def test_end_frame_with_tiny_buffers(engine):                      # the epilogue loop, encode.rs:442-464
    enc = EncodeOptions().engine(engine).checksum_flag(True).into_raw_encoder()
    enc.compress(INPUT, bytearray(0))
    out = bytearray(3)
    got = bytearray()
    while True:
        p = enc.end_frame(out)
        got += out[:p.out_progress()]
        if p.data_left() == 0:
            break
        assert p.out_progress() == 3
    assert enc.seek_table().frame_size_comp(0) == len(got)
    assert zko.frame_decode(bytes(got), len(INPUT), True)[0] == INPUT


def test_std_encoder_io_copy_roundtrip(engine):                    # lib.rs:265-287
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(777)).into_encoder(sink)
    for i in range(0, len(INPUT), 1000):
        assert enc.write(INPUT[i:i + 1000]) == len(INPUT[i:i + 1000])
    n = enc.finish()
    seekable = sink.getvalue()
    assert n == len(seekable)                                      # finish() returns the bytes written
    d = DecodeOptions(seekable).engine(engine).into_decoder()
    assert d.read_to_end() == INPUT
    assert d.seek_table().num_frames() == -(-len(INPUT) // 777)


def test_encoder_empty_input_yields_one_empty_frame(engine):       # encode.rs:755-757
    sink = io.BytesIO()
    n = EncodeOptions().engine(engine).into_encoder(sink).finish()
    assert sink.getvalue() == bytes.fromhex("28b52ffd2000010000") + bytes.fromhex("5e2a4d1811000000" "09000000" "00000000" "01000000" "00" "b1ea928f")
    assert n == 9 + 25


def test_encoder_exact_multiple_has_no_trailing_empty_frame(engine):
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(1000)).batch_frames(2).into_encoder(sink)
    enc.write(INPUT[:6000])
    enc.finish()
    st = SeekTable.from_seekable(sink.getvalue())
    assert st.num_frames() == 6 and st.size_decomp() == 6000


def test_encoder_explicit_end_frame_then_finish_adds_empty_frame(engine):
    # the reference bench pattern: write_all + end_frame (benches/compress.rs:42-45); finish() then ends a new, empty frame
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).into_encoder(sink)
    enc.write(INPUT)
    assert enc.end_frame() > 0
    assert enc.seek_table().num_frames() == 1
    enc.finish()
    st = SeekTable.from_seekable(sink.getvalue())
    assert st.num_frames() == 2 and st.frame_size_decomp(1) == 0 and st.frame_size_comp(1) == 9


def test_stand_alone_seek_table_head_and_foot(engine):             # lib.rs:136-200
    for fmt in (Format.Head, Format.Foot):
        sink = io.BytesIO()
        enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(2000)).into_encoder(sink)
        enc.write(INPUT)
        enc.end_frame()
        enc.flush()                                                   # push the 131 591-byte staging buffer out (encode.rs:796-799)
        payload = sink.getvalue()
        st = enc.seek_table()
        table = st.to_bytes(fmt)
        parsed = SeekTable.from_reader(table) if fmt == Format.Head else SeekTable.from_seekable(table)
        assert parsed == st
        d = DecodeOptions(payload).engine(engine).seek_table(parsed).into_decoder()     # payload without any seek table
        assert d.read_to_end() == INPUT


def test_readme_seek_example(engine):                              # decode.rs:533-540: seek(Start(7)) then read == "World!"
    sink = io.BytesIO()
    enc = Encoder(sink, EncodeOptions().engine(engine))
    enc.write(b"Hello, World!")
    enc.finish()
    d = DecodeOptions(sink.getvalue()).engine(engine).into_decoder()
    d.seek(SeekFrom.Start, 7)
    assert d.read(100) == b"World!"


def test_patch_cycle(engine):                                       # lib.rs:202-263 (test_patch_cycle), both directions on the GPU
    old = zko.gen_text(40000, 21)
    new = old[:15000] + b"++inserted++" + old[15000:30000] + zko.gen_text(2000, 22) + old[30000:]
    for policy in (FrameSizePolicy.Uncompressed(3000), FrameSizePolicy.Compressed(700)):
        enc = EncodeOptions().engine(engine).checksum_flag(True).frame_size_policy(policy).into_raw_encoder()
        out = bytearray()
        scratch = bytearray(max(1, len(new) // 500))               # every call is partial (lib.rs:214-216)
        pos = 0
        while pos < len(new):
            p = enc.compress_with_prefix(new[pos:pos + 777], scratch, old)
            out += scratch[:p.out_progress()]
            pos += p.in_progress()
        while True:
            p = enc.end_frame(scratch)
            out += scratch[:p.out_progress()]
            if p.data_left() == 0:
                break
        st = enc.seek_table()
        seekable = bytes(out) + st.to_bytes()
        d = DecodeOptions(seekable).engine(engine).into_decoder()
        got = bytearray()
        buf = bytearray(max(1, len(new) // 300))
        while True:
            n = d.decompress_with_prefix(buf, old)
            if n == 0:
                break
            got += buf[:n]
        assert bytes(got) == new
        if Z.load("system") is not None:                            # and the real libzstd agrees, given the same prefix
            assert Z.decode_stream(bytes(out), len(new), "system", prefix=old) == new
        plain = EncodeOptions().engine(engine).checksum_flag(True).frame_size_policy(policy).into_raw_encoder()
        assert st.num_frames() >= 1


def test_encoder_prefix_switch_takes_effect_at_frame_start(engine):  # encode.rs:334-338: ref_prefix only when frame_d_size == 0
    a, b = zko.gen_text(5000, 31), zko.gen_text(5000, 32)
    data1, data2 = a[1000:3500], b[500:4200]                        # frame 1 (+ part of 2) under prefix a, the rest under b
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).checksum_flag(True).frame_size_policy(FrameSizePolicy.Uncompressed(2000)).into_encoder(sink)
    enc.compress_with_prefix(data1, a)                              # frames [0,2000) and the open frame [2000,2500) begin under a
    enc.compress_with_prefix(data2, b)                              # fills the open frame (still a's), later frames begin under b
    enc.finish()
    seekable = sink.getvalue()
    whole = data1 + data2
    st = zk.SeekTable.from_seekable(seekable)
    nf = st.num_frames()
    assert nf == -(-len(whole) // 2000)
    pos = 0
    for i in range(nf):
        c = st.frame_size_comp(i); dsz = st.frame_size_decomp(i)
        pre = a if i < 2 else b
        out, used = zko.frame_decode(seekable[pos:pos + c], dsz, True, prefix=pre)
        assert used == c and out == whole[i * 2000:i * 2000 + dsz], i
        pos += c


def test_fuzz_roundtrip_basic(engine):                             # fuzz_targets/roundtrip_basic.rs: 100-byte frames
    rng = np.random.default_rng(8)
    for i in range(20):
        n = int(rng.integers(0, 3000))
        data = zko.gen_text(n, 50 + i) if i % 2 else zko.gen_random(n, 50 + i)
        sink = io.BytesIO()
        enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(100)).into_encoder(sink)
        enc.write(data)
        enc.finish()
        assert Decoder(DecodeOptions(sink.getvalue()).engine(engine)).read_to_end() == data


@pytest.mark.parametrize("n", [1, 60, 500, 1023, 4000])       # proptest lib.rs:317-319, 328-330: Compressed(1..1023)
def test_compressed_frame_size_policy_roundtrip(engine, n):
    seekable, st = raw_roundtrip(engine, FrameSizePolicy.Compressed(n), checksum=True, scratch=211)
    assert st.size_decomp() == len(INPUT)
    # every frame but the last reached the compressed-size threshold
    for i in range(st.num_frames() - 1):
        assert st.frame_size_comp(i) >= n
    if n >= 500:
        assert st.num_frames() < len(INPUT) // 100
    d = DecodeOptions(seekable).engine(engine).into_decoder()
    assert d.read_to_end() == INPUT
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Compressed(n)).into_encoder(sink)
    for i in range(0, len(INPUT), 700):
        enc.write(INPUT[i:i + 700])
    total = enc.finish()
    assert total == len(sink.getvalue())
    assert Decoder(DecodeOptions(sink.getvalue()).engine(engine)).read_to_end() == INPUT


def _compressed_policy_check(engine, data, n, writes):
    """Encoder<W> under Compressed(n): every frame but the last ends inside upstream's window n <= c_size < n + 131 591
    (encode.rs:341-347, 537-544: the size is compared after every call, a call emits at most the 131 591-byte buffer)."""
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Compressed(n)).checksum_flag(True).into_encoder(sink)
    for piece in writes(data):
        enc.write_all(piece)
    total = enc.finish()
    blob = sink.getvalue()
    assert total == len(blob)
    dec = Decoder(DecodeOptions(blob).engine(engine))
    st = dec.seek_table()
    assert st.size_decomp() == len(data)
    sizes = [st.frame_size_comp(i) for i in range(st.num_frames())]
    assert all(n <= c < n + 131591 for c in sizes[:-1]), (n, sizes[:8])
    assert dec.read_to_end() == bytes(data)
    return sizes


def test_compressed_policy_one_large_write(engine):
    # a single 96 MiB write must not become a single frame: ~90 frames of ~1 MiB compressed, cut and encoded in batches
    data = zko.gen_chunks(96 << 20)
    sizes = _compressed_policy_check(engine, data, 1 << 20, lambda d: [d])
    assert 30 < len(sizes) < 50


def test_compressed_policy_streamed_in_small_writes(engine):
    data = zko.gen_chunks(40 << 20)
    sizes = _compressed_policy_check(engine, data, 256 << 10, lambda d: (d[i:i + 8192] for i in range(0, len(d), 8192)))
    assert len(sizes) > 40


def test_compressed_policy_when_the_ratio_jumps(engine):
    # text, random bytes, zeros, text: the predicted frame ends are wrong at every change and are cut again (or left to the
    # exact path); zero-filled stretches never reach n and end with the stream
    rng = np.random.default_rng(5)
    text = zko.gen_chunks(24 << 20)
    data = text[:12 << 20] + rng.integers(0, 256, 6 << 20, dtype=np.uint8).tobytes() + bytes(3 << 20) + text[12 << 20:]
    _compressed_policy_check(engine, data, 512 << 10, lambda d: [d[:20 << 20], d[20 << 20:]])


def test_compressed_policy_short_stream_exact_path(engine):
    # too little input for a batch: the frame-by-frame path alone, same window
    data = zko.gen_chunks(5 << 20)
    sizes = _compressed_policy_check(engine, data, 300_000, lambda d: [d])
    assert len(sizes) >= 5


def test_compressed_policy_with_a_prefix_in_bulk(engine):
    """lib.rs:350 (test_patch_cycle under Compressed(n)) at a size where the frames are cut and encoded in batches: every
    batch goes against the same prefix, the frames end inside the window, the Decoder needs the prefix to get them back."""
    old = zko.gen_text(400_000, 31)
    new = zko.gen_chunks(40 << 20, 9)
    n = 256 << 10
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Compressed(n)).checksum_flag(True).into_encoder(sink)
    for i in range(0, len(new), 12 << 20):
        piece = new[i:i + (12 << 20)]
        done = 0
        while done < len(piece):
            done += enc.compress_with_prefix(piece[done:], old)
    enc.finish()
    blob = sink.getvalue()
    dec = Decoder(DecodeOptions(blob).engine(engine))
    st = dec.seek_table()
    sizes = [st.frame_size_comp(i) for i in range(st.num_frames())]
    assert len(sizes) > 40 and all(n <= c < n + 131591 for c in sizes[:-1])
    out = bytearray(len(new))
    got = 0
    while got < len(new):
        k = dec.decompress_with_prefix(memoryview(out)[got:], old)
        assert k > 0
        got += k
    assert bytes(out) == new
    with pytest.raises(zk.Error):                                   # without the prefix the first frame does not decode to anything valid
        Decoder(DecodeOptions(blob).engine(engine)).read_to_end()


def test_large_writes_into_frames_larger_than_the_write(engine):
    """Uncompressed(fs) with fs above 64 MiB fed by 64 MiB+ writes that are smaller than fs -- the large-write
    branch computed how much of the write completes the open frame without clamping it to the write (it read past the
    caller's buffer and its length arithmetic wrapped).  Frames: 160 MiB, 160 MiB, 16 MiB; the output round-trips."""
    fs = 160 << 20
    piece = np.frombuffer(zko.gen_chunks(8 << 20, 3), np.uint8)
    w = np.tile(piece, 12)                                          # 96 MiB per write: >= 64 MiB and < fs
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(fs)).checksum_flag(True).into_encoder(sink)
    total_in = 0
    for k in range(3):
        enc.write_all(w.tobytes())
        total_in += len(w)
    enc.write_all(w[:48 << 20].tobytes())                           # ends exactly ON a frame boundary?  no: 336 MiB = 2 fs + 16 MiB
    total_in += 48 << 20
    n = enc.finish()
    blob = sink.getvalue()
    assert n == len(blob)
    dec = Decoder(DecodeOptions(blob).engine(engine))
    st = dec.seek_table()
    assert [st.frame_size_decomp(i) for i in range(st.num_frames())] == [fs, fs, total_in - 2 * fs]
    out = dec.read_to_end()
    assert len(out) == total_in
    assert out[:len(w)] == w.tobytes() and out[3 * len(w):] == w[:48 << 20].tobytes()
    # a write that ends exactly at the frame's end leaves that frame open (upstream closes it on the next call): no empty tail
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Uncompressed(96 << 20)).into_encoder(sink)
    enc.write_all(w[:32 << 20].tobytes())
    enc.write_all(w[:64 << 20].tobytes())                           # 64 MiB write, fills the 96 MiB frame to the byte
    enc.finish()
    st = Decoder(DecodeOptions(sink.getvalue()).engine(engine)).seek_table()
    assert [st.frame_size_decomp(i) for i in range(st.num_frames())] == [96 << 20]


def test_compressed_policy_when_compressible_input_turns_random(engine):
    """far from n on input that compresses very well the probes of the exact path used to step geometrically (an eighth
    of the frame so far), and input that stopped compressing inside such a step carried the frame past n + 131 591.  The frame is now
    measured piece by piece there (host/encoder.cpp, compress_with_prefix): 80 MiB of zeros -- a few KiB -- then random bytes."""
    rng = np.random.default_rng(11)
    data = bytes(80 << 20) + rng.integers(0, 256, 6 << 20, dtype=np.uint8).tobytes() + bytes(20 << 20) + zko.gen_chunks(4 << 20, 3)
    sizes = _compressed_policy_check(engine, data, 1 << 20, lambda d: (d[i:i + (4 << 20)] for i in range(0, len(d), 4 << 20)))
    assert len(sizes) >= 6
    # the same through the RawEncoder alone (no batches: every frame end is found by the probes)
    from zeekstd_amd import EncodeOptions as EO
    raw = EO().engine(engine).frame_size_policy(FrameSizePolicy.Compressed(1 << 20)).checksum_flag(True).into_raw_encoder()
    out = bytearray(131591)
    blob = bytearray()
    pos, mv = 0, memoryview(data)
    while pos < len(data):
        p = raw.compress(mv[pos:pos + (1 << 20)], out)
        pos += p.in_progress(); blob += out[:p.out_progress()]
    while True:
        e = raw.end_frame(out)
        blob += out[:e.out_progress()]
        if not e.data_left():
            break
    st = raw.seek_table()
    cs = [st.frame_size_comp(i) for i in range(st.num_frames())]
    assert all((1 << 20) <= c < (1 << 20) + 131591 for c in cs[:-1]), cs[:8]
    assert st.size_decomp() == len(data) and st.size_comp() == len(blob)


def test_compressed_policy_on_input_that_barely_has_a_size(engine):
    """under Compressed(n) zero-filled input never lets the speculative path cut a frame (a frame would
    pass MAX_FRAME_SIZE before reaching n); the held bytes used to pile up in pinned memory until finish().  They are now
    drained through the exact path once 64 MiB are held, whose probes step geometrically while far from n."""
    sink = io.BytesIO()
    enc = EncodeOptions().engine(engine).frame_size_policy(FrameSizePolicy.Compressed(1 << 20)).into_encoder(sink)
    z = bytes(48 << 20)
    for _ in range(4):
        enc.write_all(z)                                            # 192 MiB of zeros: a few KiB compressed
    enc.finish()
    blob = sink.getvalue()
    dec = Decoder(DecodeOptions(blob).engine(engine))
    assert dec.seek_table().size_decomp() == 192 << 20 and len(blob) < 1 << 20
    out = dec.read_to_end()
    assert len(out) == 192 << 20 and not any(out[::4099])
