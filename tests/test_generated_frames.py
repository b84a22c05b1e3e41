"""Frames no encoder here writes (tests/helpers/zstd_gen.py: every choice RFC 8878 leaves open drawn at random -- header forms, block types,
literal modes and header sizes, Treeless literals, the four modes per sequence table with random FSE distributions, repeat codes with and without
Literals_Length 0) through the real libzstd 1.5.7, the oracle and the kernels' lane code on the CPU (tests/sim/zk_sim.cpp, all three sequence
walks).  libzstd decides what a frame means; the generator's own model, the oracle and the lane code have to agree with it.  The kernels themselves:
tests/test_gpu_generated_frames.py.  (The goldens are what libzstd's encoder emits; this is the rest of the format.)"""
import collections

import pytest

from conftest import sim_decode
from helpers import zstd_gen
from oracle import zko
from oracle import libzstd_ref as Z

WANTED = {"raw", "rle", "compressed", "single", "windowed", "fcs0", "fcs1", "fcs2", "fcs3", "checksum", "did_field", "unused_bit", "lit_raw", "lit_rle", "lit_huf", "lit_treeless",
          "lit_hdr1", "lit_hdr2", "lit_hdr3", "huf_streams1", "huf_streams4", "huf_fmt0", "huf_fmt1", "huf_fmt2", "huf_fmt3", "huf_weights_direct", "huf_weights_fse", "fse_lowprob",
          "nseq_form1", "nseq_form2", "off_new", "rep_idx0", "rep_idx1", "rep_idx2", "rep_idx3"} | {"%s_mode%d" % (t, m) for t in ("ll", "of", "ml") for m in range(4)}


def test_generated_frames_mean_what_libzstd_says():
    ref = "1.5.7" if Z.load("1.5.7") is not None else "system"
    if Z.load(ref) is None:
        pytest.skip("no libzstd in the image")
    seen = collections.Counter()
    for seed in range(1500):
        f, out, feats = zstd_gen.generate(seed, zko.xxh64)
        got, state = Z.decode_stream_verdict(f, ref)
        assert state == "end" and got == out, (seed, state, len(got), len(out), sorted(feats))
        o, used = zko.frame_decode(f, len(out) + 64, True)
        assert used == len(f) and o == out, (seed, sorted(feats))
        seen.update(feats)
    assert not WANTED - set(seen), WANTED - set(seen)


@pytest.mark.parametrize("quad", [False, True, 2])
def test_generated_frames_through_the_lane_code(quad):
    for seed in range(600):
        f, out, feats = zstd_gen.generate(seed, zko.xxh64)
        rc, o, st = sim_decode(f, [(len(f), len(out))], quad=quad)
        assert rc == 0 and o == out, (seed, rc, sorted(feats))


def test_generated_frames_through_the_executor_in_segments():
    """several "workgroups" per frame (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill): segments of one block or a few, of a few hundred
    bytes (a segment per block: every offset across a block's first byte is a hole), fill rounds of 3 ... 1024 lanes"""
    for seed in range(600):
        f, out, feats = zstd_gen.generate(seed, zko.xxh64)
        seg = ((131072, 64, 2), (1 + seed % 5000, 3 + seed % 200, 2), (65536, 1024, 1))[seed % 3]
        rc, o, st = sim_decode(f, [(len(f), len(out))], seg=seg)
        assert rc == 0 and o == out, (seed, rc, seg, sorted(feats))


def test_generated_frames_side_by_side_in_one_archive():
    """many of them in one call: the block records of different frames sit side by side, Repeat / Treeless reach back inside their own frame only"""
    frames, comp, data = [], bytearray(), bytearray()
    for seed in range(2000, 2300):
        f, out, _ = zstd_gen.generate(seed, zko.xxh64)
        frames.append((len(f), len(out))); comp += f; data += out
    for quad in (False, 2):
        rc, o, st = sim_decode(bytes(comp), frames, quad=quad)
        assert rc == 0 and not st.any() and o == bytes(data)


def test_generated_frames_with_long_blocks():
    """thousands of sequences per block, literal sections of tens of KiB (4-stream Huffman with the wide header forms), ten blocks per frame"""
    ref = "1.5.7" if Z.load("1.5.7") is not None else "system"
    for seed in range(150):
        f, out, feats = zstd_gen.generate(100000 + seed, zko.xxh64, max_blocks=10, max_seq=4000, max_lit=100000)
        if Z.load(ref) is not None:
            assert Z.decode_stream_verdict(f, ref) == (out, "end"), seed
        o, used = zko.frame_decode(f, len(out) + 64, True)
        assert used == len(f) and o == out, seed
        for quad in (False, True, 2):
            rc, so, st = sim_decode(f, [(len(f), len(out))], quad=quad)
            assert rc == 0 and so == out, (seed, quad)
        rc, so, st = sim_decode(f, [(len(f), len(out))], seg=(131072, 256, 1))
        assert rc == 0 and so == out, seed
        # the executor in segments (several workgroups per frame): segments of 128 KiB and of a few hundred bytes
        for seg in ((131072, 64, 2), (700 + seed % 3000, 16, 2)):
            rc, so, st = sim_decode(f, [(len(f), len(out))], seg=seg)
            assert rc == 0 and so == out, (seed, seg)


def test_generated_frames_against_a_prefix():
    """the same against a raw-content prefix (ZSTD_DCtx_refPrefix): offsets reach across the frame's first byte into it, within the window"""
    ref = "1.5.7" if Z.load("1.5.7") is not None else "system"
    into = 0
    for seed in range(500):
        prefix = zko.gen_text(1 + (seed * 7919) % 90000, seed % 7)
        f, out, feats = zstd_gen.generate(300000 + seed, zko.xxh64, prefix=prefix)
        if Z.load(ref) is not None:
            assert Z.decode_stream(f, -1, ref, prefix=prefix, window_log_max=31) == out, seed
        o, used = zko.frame_decode(f, len(out) + 64, True, prefix=prefix)
        assert used == len(f) and o == out, seed
        rc, so, st = sim_decode(f, [(len(f), len(out))], prefix=prefix, quad=(False, True, 2)[seed % 3])
        assert rc == 0 and so == out, seed
        into += "off_into_prefix" in feats
    assert into > 100


def test_generated_frames_with_more_than_0x7F00_sequences_in_a_block():
    """the three-byte Number_of_Sequences: 32 512 ... 38 000 matches of three or four bytes back to back (the densest a block can be)"""
    ref = "1.5.7" if Z.load("1.5.7") is not None else "system"
    for seed in range(4):
        f, out, feats = zstd_gen.generate(500000 + seed, zko.xxh64, dense=True, max_blocks=8)
        assert "nseq_form3" in feats
        if Z.load(ref) is not None:
            assert Z.decode_stream_verdict(f, ref) == (out, "end"), seed
        o, used = zko.frame_decode(f, len(out) + 64, True)
        assert used == len(f) and o == out, seed
        for quad in (False, True, 2):
            rc, so, st = sim_decode(f, [(len(f), len(out))], quad=quad)
            assert rc == 0 and so == out, (seed, quad)



@pytest.mark.parametrize("quad", [False, 2])
def test_damaged_generated_frames_the_lane_code_against_the_oracle(quad):
    """one to three flipped bits per hit frame, checksums not verified: the lane code refuses exactly the frames the oracle refuses and yields its
    bytes otherwise (tests/test_gpu_generated_frames.py does this with the kernels)"""
    import numpy as np
    frames, comp, data = [], bytearray(), bytearray()
    for seed in range(20000, 20400):
        f, out, _ = zstd_gen.generate(seed, zko.xxh64)
        frames.append((len(f), len(out))); comp += f; data += out
    c = np.concatenate([[0], np.cumsum([f[0] for f in frames])]); d = np.concatenate([[0], np.cumsum([f[1] for f in frames])])
    rng = np.random.default_rng(9)
    bad = bytearray(comp)
    hit = set()
    for _ in range(300):
        i = int(rng.integers(0, len(bad)))
        bad[i] ^= 1 << int(rng.integers(0, 8))
        hit.add(int(np.searchsorted(c, i, side="right")) - 1)
    # checksum verification off: the simulator's checksum pass is separate (sim_decode runs the format checks only)
    rc, out, st = sim_decode(bytes(bad), frames, quad=quad)
    refused = 0
    for f in range(len(frames)):
        lo, hi = int(d[f]), int(d[f + 1])
        if f not in hit:
            assert st[f] == 0 and out[lo:hi] == bytes(data[lo:hi]), f
            continue
        try:
            o, used = zko.frame_decode(bytes(bad[int(c[f]):int(c[f + 1])]), hi - lo + 64, False)
            ok = len(o) == hi - lo and used == int(c[f + 1] - c[f])
        except zko.OracleError:
            ok = False
        assert ok == (st[f] == 0), (f, int(st[f]), ok)
        if ok:
            assert out[lo:hi] == o, f
        refused += not ok
    assert refused > 50
