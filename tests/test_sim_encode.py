"""The encoder's match + parse kernel (zeekstd_amd/csrc/zk_enc_match.h -- the source hipcc compiles for gfx950) executed
on the CPU by tests/sim/zk_enc_sim.cpp under a workgroup emulator (a fiber per lane, barriers and wave collectives as
meeting points), compared block by block with the CPU twin oracle/zstd_oracle_enc.c.  What the reference does here is
ZSTD_compressStream2 (lib/src/encode.rs:340-346); compressed bytes are unpinned by it, so the twin is the yardstick and
the -m gpu tests then hold the real kernel to the same bytes."""
import numpy as np
import pytest

from conftest import enc_sim_match
from oracle import zko


def _compare(data, frame_size, level, prefix=None):
    got = enc_sim_match(data, frame_size, level, prefix)
    want = []
    for o in range(0, max(len(data), 1), frame_size):
        want += zko.enc_match_debug(data[o:o + frame_size], level, prefix)
    assert len(got) == len(want)
    for b, ((gs, gl, bsz), (ws, wl)) in enumerate(zip(got, want)):
        assert len(gs) == len(ws), (b, len(gs), len(ws))
        bad = np.nonzero(gs != ws)[0]
        assert bad.size == 0, (b, int(bad[0]), hex(int(gs[bad[0]])), hex(int(ws[bad[0]])))
        assert gl == wl, b
        # the sequences + literals reproduce the block size
        assert sum((int(x) >> 16) & 0xFFFF for x in gs) + len(gl) == bsz                     # match bytes + literals = the block


CASES = {
    "text": [["text", 70000, 11]],
    "mixed": [["text", 20000, 1], ["zeros", 9000], ["random", 7000, 3], ["rep", "616263", 3000], ["text", 30000, 2]],
    "runs": [["rep", "61", 700], ["rep", "62", 13], ["rep", "6364", 900], ["random", 300, 5], ["rep", "00", 5000]],
    "records": [["records", 4000, 7, "00010203040506070809"]],
    "tiny": [["text", 23, 5]],
    "seven": [["text", 7, 5]],
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("level", [1, 3, 6])
def test_sim_match_equals_twin(name, level):
    data = zko.make_input(CASES[name])
    _compare(data, 1 << 21, level)


def test_sim_match_small_frames_and_blocks():
    data = zko.make_input([["text", 50000, 21]])
    _compare(data, 10000, 1)          # 1 KiB .. 4 KiB blocks, partial groups
    _compare(data, 3000, 3)


def test_sim_match_segments():
    """a frame above ZKE_SEGMENT (256 KiB): the second segment starts from the 57280 bytes before it"""
    data = zko.make_input([["text", 300000, 31], ["rep", "6162636465666768", 2000]])
    _compare(data, 1 << 21, 1)


def test_sim_match_prefix():
    prefix = zko.make_input([["text", 70001, 41]])
    data = prefix[1000:30000] + zko.make_input([["text", 9000, 42]])
    _compare(data, 1 << 21, 1, prefix)
    _compare(data[:5000], 1 << 21, 3, prefix[:1003])


def _edited(old, seed, n_edits):
    """old with n_edits changes: replacements, insertions, deletions of up to a KiB"""
    rng = np.random.default_rng(seed)
    pos = sorted(int(x) for x in rng.integers(0, len(old) - 2048, n_edits))
    out, last = bytearray(), 0
    for i, p in enumerate(pos):
        if p < last:
            continue
        out += old[last:p]
        if i % 3 == 0:
            out += zko.gen_text(300 + 40 * i, 900 + i); last = p + 300 + 40 * i
        elif i % 3 == 1:
            out += zko.gen_text(100 + 30 * i, 900 + i); last = p
        else:
            last = p + 150 + 20 * i
    out += old[last:]
    return bytes(out)


@pytest.mark.parametrize("level", [1, 3, 6])
def test_sim_match_long_distance(level):
    """patch mode: a prefix beyond the ring's reach is found through the long-distance table (sampled positions, the tile's
    first hit as an offset for the whole tile, previous offsets compared through memory, joined across tiles)"""
    old = zko.make_input([["text", 200000, 51], ["random", 30000, 52], ["text", 170000, 53]])
    new = _edited(old, 7, 12)
    _compare(new, 1 << 21, level, old)                 # one frame, two segments
    if level == 1:
        _compare(new, 150000, level, old)              # three frames against the same prefix
        _compare(new[:70000], 1 << 21, level, old[:57284])      # a prefix four bytes beyond the ring's reach


@pytest.mark.parametrize("level", [3, 6])
def test_sim_match_far_history_inside_a_frame(level):
    """round 4: from level 2 on a frame longer than the ring's reach finds its own far history through a table over its own bytes
    (repeats 100+ KB apart: a section that comes back edited, a second copy of another) -- one frame of two segments, three frames
    with their own tables, and a last frame too short for any of it"""
    a = zko.make_input([["text", 90000, 61]])
    b = zko.make_input([["text", 70000, 62], ["random", 8000, 63]])
    data = a + b + _edited(a, 9, 6) + zko.make_input([["text", 30000, 64]]) + b[20000:60000] + a[:15000]
    _compare(data, 1 << 21, level)
    if level == 3:
        _compare(data, 150000, level)                      # 150 000 + 150 000 + a tail below the ring's reach
        _compare(data[:200000], 65536, level)              # frames of 64 KiB: in frame, but nothing lies further back than the ring reaches


@pytest.mark.parametrize("level", [2, 3, 9])
def test_sim_match_dense_far_history(level):
    """round 6 (VERDICT r5 "a real upper level"): level 0 / 3 and up look every position up in two tables per matcher segment (first
    occurrence in the segment, last occurrence in the segment before; 2^17 slots, 2^18 from level 9 on) -- the short far matches of
    rare words that libzstd's level 3 (cli/src/args.rs:192) finds in its 2 MiB window.  Three segments of the 8d text: the kernel under
    the emulator equals the twin sequence for sequence, and the dense levels find far matches that level 2 does not."""
    data = zko.gen_chunks(600000)
    _compare(data, 1 << 21, level)
    if level == 3:
        # where a far copy must NOT be taken (the twin's rules: a byte run; a cheap offset a few positions later): records of 4 random + 16
        # constant bytes, runs of ten equal bytes -- cut by tiles, segments and a second frame
        import numpy as np
        rec = zko.make_input([["records", 16000, 7, "000102030405060708090a0b0c0d0e0f"]])
        runs = np.repeat(np.random.default_rng(5).integers(0, 256, 30000, dtype=np.uint8), 10).tobytes()
        _compare(rec + runs + rec[:100000], 500000, level)
    far = lambda lv: sum(int(((sq >> 32) > 57280 + 3).sum()) for sq, _ in zko.enc_match_debug(data, lv))
    if level == 2:
        assert far(2) < 200
    else:
        assert far(level) > 20 * max(far(2), 50), (far(level), far(2))


def test_huffman_build_as_the_kernel_does_it():
    """zk_k_enc_entropy ranks a block's symbols with all lanes of the wave and hands them to zke_huf_lengths sorted, then assigns the
    canonical codes from per-weight counts and ranks: the same lengths, depth and codes as the serial functions (random, skewed,
    tied and degenerate histograms; counts that need the halving rounds to fit 11 bits)."""
    import ctypes as C
    from conftest import enc_sim_lib
    lib = enc_sim_lib()
    lib.zk_enc_sim_huf.restype = C.c_int
    lib.zk_enc_sim_huf.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(5)
    cases = []
    for _ in range(300):
        nsym = int(rng.integers(2, 129))
        kind = int(rng.integers(0, 5))
        if kind == 0: c = rng.integers(0, 1000, nsym)
        elif kind == 1: c = (rng.random(nsym) ** 8 * 100000).astype(np.int64)                 # skewed: deep trees, halving rounds
        elif kind == 2: c = rng.integers(0, 3, nsym)                                           # ties
        elif kind == 3: c = np.where(rng.random(nsym) < 0.1, rng.integers(1, 50, nsym), 0)     # few symbols
        else: c = 2 ** rng.integers(0, 17, nsym)                                               # a Fibonacci-like spread
        c = c.astype(np.uint32); c[-1] = max(int(c[-1]), 1)                                    # the last symbol occurs (nsym = maxsym + 1)
        cases.append(c)
    cases += [np.array([1, 1], np.uint32), np.array([5, 0, 0, 7], np.uint32), np.arange(1, 129, dtype=np.uint32), np.full(128, 9, np.uint32)]
    for c in cases:
        assert lib.zk_enc_sim_huf(c.ctypes.data, len(c)) == 0, c.tolist()


def test_match_kernel_under_sanitizers(tmp_path):
    """The match + parse kernel under the workgroup emulator, built with AddressSanitizer + UBSan (tests/sim/encode_san.cpp): the
    kernel's `__shared__` arrays are real arrays on the CPU, so an index that leaves the ring, the table, `best[]` or a tile's sequence
    area -- on the device silent corruption of whatever lies next to it in LDS -- is a report, as is an access outside the source /
    prefix / output buffers (exact-size heap allocations).  Text with a prefix beyond the ring (the long-distance table), byte runs in
    small frames at level 6, a frame larger than a segment with far history inside it at level 3 (the DENSE instance: its entries per
    position, the over-read into the slack), odd-sized frames at level 9.  (54 further shapes: DESIGN.md 3.)"""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "encsan")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                         os.path.join(ROOT, "tests", "sim", "encode_san.cpp"), "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("no sanitizer runtime for g++ here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    first = True
    for cfg in (("1", "90000", "90000", "0", "200000"), ("6", "100000", "32768", "1", "20000"), ("3", "400000", "400000", "0"),
                ("9", "420001", "300001", "0")):       # (round 6) the dense levels' candidate entries: frames of an odd size, a second frame beyond the ring's reach
        r = subprocess.run([exe, *cfg], capture_output=True, text=True, timeout=900)
        if first and r.returncode != 0 and "AddressSanitizer" in r.stderr and "ERROR: AddressSanitizer:" not in r.stderr:
            pytest.skip("the sanitizer runtime cannot start here")
        first = False
        assert r.returncode == 0, (cfg, r.stdout[-300:], r.stderr[-3000:])
