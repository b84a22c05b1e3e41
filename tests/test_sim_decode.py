"""The kernels' per-lane device code (zeekstd_amd/csrc/zk_device.h) executed on the CPU by
tests/sim/zk_sim.cpp, checked against the golden archives and the oracle.  This is what lets the
CPU suite vouch for the decode logic; the -m gpu tests then check the real kernels."""
import numpy as np
import pytest

from conftest import sim_decode
from oracle import zko
from oracle import libzstd_ref as Z


@pytest.mark.parametrize("tile", [(16, 1024), (4, 7), (1, 1)])
def test_sim_goldens(golden, tile):
    if tile == (1, 1) and golden.meta["input_len"] > 50000:
        pytest.skip("slow")
    data = golden.input()
    rc, out, st = sim_decode(golden.comp, golden.frames, B=tile[0], CH=tile[1])
    assert rc == 0 and not st.any()
    assert out == data


def test_sim_frame_subrange():
    from conftest import GOLDENS
    g = next(x for x in GOLDENS if x.name == "text_100B_frames")
    data = g.input()
    _, d = g.offsets()
    rc, out, st = sim_decode(g.comp, g.frames, first=7, count=20)
    assert rc == 0 and out == data[int(d[7]):int(d[27])]


def test_sim_rejects_corruption():
    from conftest import GOLDENS
    g = next(x for x in GOLDENS if x.name == "text_l1_64k")
    rng = np.random.default_rng(5)
    data = g.input()
    bad = ok = 0
    for _ in range(40):
        comp = bytearray(g.comp)
        i = int(rng.integers(0, len(comp)))
        comp[i] ^= 1 << int(rng.integers(0, 8))
        rc, out, st = sim_decode(bytes(comp), g.frames)
        # either the frame is rejected, or (flip in a place the format does not protect without a
        # checksum pass) it decodes to *something* of the right length; never a crash / overrun
        assert len(out) == len(data)
        bad += rc != 0
        ok += rc == 0
    assert bad > 0


def test_sim_bad_magic_and_truncation():
    from conftest import GOLDENS
    g = next(x for x in GOLDENS if x.name == "hello")
    rc, _, st = sim_decode(b"\x00" + g.comp[1:], g.frames)
    assert rc == -10 and st[0] == 10                      # prefix_unknown
    rc, _, st = sim_decode(g.comp, [(g.frames[0][0] - 1, g.frames[0][1])])
    assert rc != 0
    rc, _, st = sim_decode(g.comp, [(g.frames[0][0], g.frames[0][1] + 1)])
    assert rc == -20


@pytest.mark.skipif(Z.load("system") is None, reason="no system libzstd")
@pytest.mark.parametrize("level", [1, 3, 19])
def test_sim_vs_live_libzstd(level):
    data = zko.make_input([["text", 300000, 70 + level], ["rep", "00", 200000], ["random", 20000, 6], ["text", 100000, 71]])
    for fs in (2 << 20, 65536, 700):
        d = data if fs > 700 else data[:20000]
        comp, frames = Z.encode_seekable_frames(d, fs, level, fs != 65536, "system")
        rc, out, st = sim_decode(comp, frames)
        assert rc == 0 and out == d


def test_sim_prefix_goldens(prefix_golden):
    g = prefix_golden
    rc, out, st = sim_decode(g.comp, g.frames, prefix=g.prefix())
    assert rc == 0 and not st.any()
    assert out == g.input()


# ---- zk_k_fse_quad: three lock-stepped lanes per block (zk_seq_walk_quad), fibers on the CPU ----------------------
# quad = 1: 16-bit cells (sym | x), the large layouts; 2: 8-byte cells (ZkCells64), what the small-batch kernels walk
@pytest.mark.parametrize("quad", [1, 2])
def test_sim_quad_goldens(golden, quad):
    rc, out, st = sim_decode(golden.comp, golden.frames, quad=quad)
    assert rc == 0 and not st.any()
    assert out == golden.input()


@pytest.mark.parametrize("quad", [1, 2])
def test_sim_quad_prefix_goldens(prefix_golden, quad):
    g = prefix_golden
    rc, out, st = sim_decode(g.comp, g.frames, prefix=g.prefix(), quad=quad)
    assert rc == 0 and not st.any()
    assert out == g.input()


def test_sim_quad_matches_lane_walk_on_corrupt_input():
    """Both sequence walks must agree frame by frame on damaged archives too: same status words, same bytes where a
    frame still decodes (status 1020 would mean the three lanes of a quad fell out of step)."""
    from conftest import GOLDENS
    rng = np.random.default_rng(23)
    for name in ("text_l1_64k", "mixed_l19", "text_l3"):
        g = next((x for x in GOLDENS if x.name == name), None)
        if g is None:
            continue
        for _ in range(30):
            comp = bytearray(g.comp)
            comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
            rc1, out1, st1 = sim_decode(bytes(comp), g.frames)
            _, d = g.offsets()
            for quad in (1, 2):
                rc2, out2, st2 = sim_decode(bytes(comp), g.frames, quad=quad)
                assert list(st1) == list(st2) and rc1 == rc2
                for f in range(len(g.frames)):
                    if st1[f] == 0:
                        assert out1[int(d[f]):int(d[f + 1])] == out2[int(d[f]):int(d[f + 1])]


@pytest.mark.skipif(Z.load("system") is None, reason="no system libzstd")
@pytest.mark.parametrize("level", [1, 3, 19])
def test_sim_quad_vs_live_libzstd(level):
    data = zko.make_input([["text", 300000, 80 + level], ["rep", "00", 200000], ["random", 20000, 7], ["text", 100000, 81]])
    for fs in (2 << 20, 65536):
        comp, frames = Z.encode_seekable_frames(data, fs, level, fs != 65536, "system")
        for quad in (1, 2):
            rc, out, st = sim_decode(comp, frames, quad=quad)
            assert rc == 0 and out == data


@pytest.mark.skipif(Z.load("system") is None, reason="no system libzstd")
@pytest.mark.parametrize("quad", [False, True, 2])
def test_sim_with_poisoned_tables(quad):
    """Every table and scratch area a block's lane builds and then reads -- Huffman weights, decode table, FSE cells: on the device
    LDS that still holds what the workgroup before left there -- is filled with pseudo-random bytes before each block: code that
    reads what it has not written would show up as wrong bytes or a false corruption verdict (64 KiB frames written by libzstd at
    levels 1 and 3: one large block with its own tables per frame, the shape a single seek decodes)."""
    import ctypes as C
    from conftest import sim_lib
    lib = sim_lib()
    lib.zk_sim_set_poison.argtypes = [C.c_uint64]
    data = zko.gen_chunks(3 << 20, 5)
    try:
        for level in (1, 3):
            comp, frames = Z.encode_seekable_frames(data, 65536, level, True)
            for seed in (0x1234567, 0xDEADBEEF12345):
                lib.zk_sim_set_poison(seed)
                rc, out, st = sim_decode(comp, frames, quad=quad, CH=2048)
                assert rc == 0 and not st.any() and out == data, (level, seed)
    finally:
        lib.zk_sim_set_poison(0)


@pytest.mark.parametrize("quad", [False, True, 2])
def test_sim_handmade_frames(quad):
    """RLE_Mode sequence tables (hand-written frames): one-cell tables, accuracy log 0, through both sequence walks."""
    from conftest import HANDMADE, HANDMADE_BAD, HANDMADE_BAD_CPU
    for name, frame, expect in HANDMADE:
        rc, out, st = sim_decode(frame, [(len(frame), len(expect))], quad=quad)
        assert rc == 0 and not st.any() and out == expect, name
    for name, frame, dsize, code in HANDMADE_BAD + HANDMADE_BAD_CPU:      # libzstd 1.5.7's verdict, code for code
        rc, out, st = sim_decode(frame, [(len(frame), dsize)], quad=quad)
        assert rc == -code and st[0] == code, name


# ---- archives of THIS engine's encoder (its CPU twin writes the same bytes): one set of FSE tables per frame, Repeat_Mode in the later
# blocks, 32 KiB blocks, from level 2 on offsets that reach back through the whole frame -- the shapes the shared-table walks and
# the window check of the executor see on the device, here through the same lane code on the CPU
def _doc_like_small(seed=5):
    rng = np.random.default_rng(seed)
    pool = [zko.gen_text(int(rng.integers(2000, 20000)), 300 + i) for i in range(6)]
    out = bytearray()
    for i in range(14):
        out += zko.gen_text(int(rng.integers(20000, 90000)), 400 + i)
        out += pool[int(rng.integers(0, len(pool)))]
    return bytes(out)


@pytest.mark.parametrize("level", [1, 3, 6])
@pytest.mark.parametrize("quad", [0, 1, 2])
def test_sim_twin_made_frames(level, quad):
    rng = np.random.default_rng(17)
    inputs = {
        "text": zko.gen_chunks(700_000, 3),
        "doc_like": _doc_like_small(),                               # repeats hundreds of KiB back: in-frame far matches from level 2
        "runs": b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400)) for _ in range(3000)),
        "period37": (bytes(range(37)) * 9000)[:300_000],
        "mixed": zko.gen_text(150_000, 9) + rng.integers(0, 256, 40_000, dtype=np.uint8).tobytes() + bytes(70_000) + zko.gen_text(90_000, 10),
    }
    for name, data in inputs.items():
        for fs in (len(data), 65536):
            frames, comp = [], bytearray()
            for o in range(0, len(data), fs):
                fr = zko.frame_encode(data[o:o + fs], level, True)
                frames.append((len(fr), len(data[o:o + fs])))
                comp += fr
            rc, out, st = sim_decode(bytes(comp), frames, quad=quad)
            assert rc == 0 and not st.any(), (name, fs)
            assert out == data, (name, fs)


def test_sim_twin_made_frames_damaged():
    """The three sequence walks and the executor's checks on damaged frames of the engine's own shape (far offsets, a declared
    window over the frame, Repeat_Mode tables): the same verdict per frame from every walk, the same bytes where a frame still
    decodes, and nothing that reads or writes where it should not (the harness runs under the allocator's guard of numpy buffers)."""
    data = _doc_like_small(6)[:600_000]
    fs = 200_000
    frames, comp = [], bytearray()
    for o in range(0, len(data), fs):
        fr = zko.frame_encode(data[o:o + fs], 3, True)
        frames.append((len(fr), len(data[o:o + fs])))
        comp += fr
    _, d = (np.cumsum([0] + [f[0] for f in frames]), np.cumsum([0] + [f[1] for f in frames]))
    rng = np.random.default_rng(29)
    seen_bad = 0
    for _ in range(60):                                                  # (round 5's level-3 bytes: one flip in twelve is structural damage, the rest is wrong bytes)
        bad = bytearray(comp)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        rc0, out0, st0 = sim_decode(bytes(bad), frames)
        seen_bad += int(any(st0))
        for quad in (1, 2):
            rc, out, st = sim_decode(bytes(bad), frames, quad=quad)
            assert list(st) == list(st0) and rc == rc0
            for f in range(len(frames)):
                if st0[f] == 0:
                    assert out[int(d[f]):int(d[f + 1])] == out0[int(d[f]):int(d[f + 1])]
    assert seen_bad >= 3                                             # (the harness does not compare checksums: structural damage only)


def test_sim_under_sanitizers_on_damaged_archives(tmp_path):
    """The decode lane code under AddressSanitizer + UBSan (tests/sim/decode_fuzz.cpp): damaged golden archives -- bit flips, random
    bytes, random and constant runs -- through the three sequence walks, every buffer an exact-size heap allocation (compressed bytes
    + ZK_COMP_PADDING, output bytes): a lane that follows a damaged header, table or offset out of its buffers is a report, where on
    the device it would be a silent read or a fault.  (The suite runs the small archives; the campaign over all 28 is in DESIGN.md.)"""
    import os
    import shutil
    import struct
    import subprocess
    from conftest import GOLDENS, ROOT
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "decfuzz")
    src = os.path.join(ROOT, "tests", "sim", "decode_fuzz.cpp")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", src, "-o", exe],
                        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("no sanitizer runtime for g++ here")
    assert cc.returncode == 0, cc.stderr[-2000:]

    def case(g):
        data = g.input()
        path = str(tmp_path / (g.name + ".bin"))
        with open(path, "wb") as f:
            f.write(struct.pack("<IQQ", len(g.frames), len(g.comp), len(data)))
            for c, d in g.frames:
                f.write(struct.pack("<QQ", c, d))
            f.write(g.comp); f.write(data)
        return path
    first = True
    for name, iters in (("text_100B_frames", 600), ("text_len_div7", 300), ("oneshot_small", 600), ("tiny_frames_10B", 400), ("hello_cks", 300)):
        g = next(x for x in GOLDENS if x.name == name)
        path = case(g)
        for walk in (0, 1, 2):
            r = subprocess.run([exe, path, str(iters if walk == 0 else iters // 6), "5", str(walk)], capture_output=True, text=True, timeout=600)
            if first and r.returncode != 0 and "AddressSanitizer" in r.stderr and "ERROR: AddressSanitizer:" not in r.stderr:
                pytest.skip("the sanitizer runtime cannot start here")             # (an environment without the address-space layout ASan needs)
            first = False
            assert r.returncode == 0, (name, walk, r.stdout[-300:], r.stderr[-3000:])
    # frames made against a prefix (history positions below the frame's first byte: the executor's second address range)
    from conftest import PREFIX_GOLDENS
    for name, iters in (("pfx_small_frames", 400), ("pfx_patch_1m_ldm", 200), ("pfx_tiny", 200)):
        g = next(x for x in PREFIX_GOLDENS if x.name == name)
        path = case(g)
        pre = str(tmp_path / (name + ".pre"))
        open(pre, "wb").write(g.prefix())
        for walk in (0, 2):
            r = subprocess.run([exe, path, str(iters if walk == 0 else iters // 10), "6", str(walk), pre], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (name, walk, r.stdout[-300:], r.stderr[-3000:])
    # frames no encoder here writes (tests/helpers/zstd_gen.py: Treeless and RLE literals, RLE / Repeat sequence tables, FSE-compressed weights, every header
    # form), damaged: the corners of the lane code that libzstd-made archives reach least
    from helpers import zstd_gen
    frames, comp, data = [], bytearray(), bytearray()
    for seed in range(800000, 800040):
        f, out, _ = zstd_gen.generate(seed, zko.xxh64)
        frames.append((len(f), len(out))); comp += f; data += out
    path = str(tmp_path / "generated.bin")
    with open(path, "wb") as fh:
        fh.write(struct.pack("<IQQ", len(frames), len(comp), len(data)))
        for c, d in frames:
            fh.write(struct.pack("<QQ", c, d))
        fh.write(comp); fh.write(data)
    for walk, iters in ((0, 500), (1, 30), (2, 30)):
        r = subprocess.run([exe, path, str(iters), "7", str(walk)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, ("generated", walk, r.stdout[-300:], r.stderr[-3000:])


@pytest.mark.parametrize("quad", [False, True, 2])
def test_sim_block_regenerates_more_than_the_window_allows(quad):
    """Block_Maximum_Size = min(Window_Size, 128 KiB) bounds what a block REGENERATES, not only what it holds (RFC 8878 3.1.1.2.4): a frame
    whose Window_Descriptor is turned down to 1 KiB while its blocks regenerate 32 KiB (offsets of 2: inside any window) is refused -- by the
    oracle, by libzstd's buffer sizing, and by the executor (round 5: it compared against 128 KiB only; found with a flipped descriptor bit
    in a prefix archive, where offsets are not held against the window)."""
    data = b"ab" * 50000
    f = bytearray(zko.frame_encode(data, 1, False))
    assert not f[4] & 0x20                                # not single segment: byte 5 is the Window_Descriptor
    rc, out, st = sim_decode(bytes(f), [(len(f), len(data))], quad=quad)
    assert rc == 0 and out == data
    f[5] = 0x00                                           # 1 KiB
    with pytest.raises(zko.OracleError):
        zko.frame_decode(bytes(f), len(data), False)
    rc, out, st = sim_decode(bytes(f), [(len(f), len(data))], quad=quad)
    assert rc == -20 and st[0] == 20



# ---- the executor in segments (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill): the same lane code, several "workgroups" per frame ------
# seg = (segment bytes, lanes of a fill round, log2 of output bytes per hole-record slot).  The simulation runs a frame's segments LAST
# first over a poisoned output buffer (no segment may need another one's bytes), a tile's waves in reverse order (the order of a tile's
# records is whatever the waves' atomics make it), the fill rounds and their lanes in reverse too, and checks that no record of a tile
# reads what another record of the tile writes.
SEGS = [(131072, 64, 2), (4096, 7, 2), (300, 1024, 2), (65536, 256, 1)]


@pytest.mark.parametrize("seg", SEGS, ids=[f"seg{s[0]}_L{s[1]}" for s in SEGS])
def test_sim_goldens_in_segments(golden, seg):
    rc, out, st = sim_decode(golden.comp, golden.frames, seg=seg)
    assert rc == 0 and not st.any()
    assert out == golden.input()


def test_sim_segments_that_overflow_are_executed_again(golden):
    """A region of one record slot per 2^20 bytes: every segment with a hole overflows, the frame goes through the frame executor."""
    from conftest import sim_lib
    import ctypes as C
    a = (C.c_uint64 * 4)()
    sim_lib().zk_sim_seg_stats(a, 1)
    rc, out, st = sim_decode(golden.comp, golden.frames, seg=(131072, 64, 20))
    assert rc == 0 and not st.any() and out == golden.input()
    sim_lib().zk_sim_seg_stats(a, 1)
    assert a[0] == 0 or a[3] > 0                       # records were only copied where nothing overflowed


@pytest.mark.skipif(Z.load("system") is None, reason="no system libzstd")
@pytest.mark.parametrize("level", [1, 3, 19])
def test_sim_segments_vs_live_libzstd(level):
    """2 MiB frames of 128 KiB blocks: at level 1 two fifths of a segment's bytes are holes (their origin lies before the segment), at
    level 3 nearly nine tenths."""
    from conftest import sim_lib
    import ctypes as C
    data = zko.make_input([["text", 700000, 170 + level], ["rep", "00", 200000], ["random", 20000, 6], ["text", 400000, 171]])
    a = (C.c_uint64 * 4)()
    for fs in (2 << 20, 300000):
        comp, frames = Z.encode_seekable_frames(data, fs, level, True, "system")
        sim_lib().zk_sim_seg_stats(a, 1)
        rc, out, st = sim_decode(comp, frames, seg=(131072, 256, 2))
        assert rc == 0 and out == data
        sim_lib().zk_sim_seg_stats(a, 1)
        assert a[0] > 0 and a[3] == 0                  # holes were left and filled; nothing had to be executed again


def test_sim_segments_match_the_frame_executor_on_corrupt_input():
    from conftest import GOLDENS
    rng = np.random.default_rng(29)
    for name in ("text_l1_64k", "mixed_l19", "text_l3", "mixed"):
        g = next((x for x in GOLDENS if x.name == name), None)
        if g is None:
            continue
        _, d = g.offsets()
        for _ in range(40):
            comp = bytearray(g.comp)
            comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
            rc1, out1, st1 = sim_decode(bytes(comp), g.frames)
            for seg in ((131072, 64, 2), (2048, 16, 2)):
                rc2, out2, st2 = sim_decode(bytes(comp), g.frames, seg=seg)
                assert [bool(x) for x in st1] == [bool(x) for x in st2]
                for f in range(len(g.frames)):
                    if st1[f] == 0:
                        assert out1[int(d[f]):int(d[f + 1])] == out2[int(d[f]):int(d[f + 1])]
