"""tests/test_generated_frames.py on the device: frames drawn from the whole of RFC 8878 (tests/helpers/zstd_gen.py; libzstd 1.5.7 is the judge of what they mean
in the CPU test, the generator's model -- which agrees with it there -- is the expectation here) through the kernels: one at a time (the small-batch path), a few
thousand side by side in one call (the batch kernels, shared-table sequence kernels included: the frames' blocks use every table mode), frames whose size
nobody tells the engine (zk_frame_content_sizes), the Level-C shim, and damaged copies against the oracle's verdict."""
import numpy as np
import pytest

from conftest import offsets_from_frames
from helpers import zstd_gen
from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu


def archive(seeds, **kw):
    frames, comp, data = [], bytearray(), bytearray()
    for seed in seeds:
        f, out, _ = zstd_gen.generate(seed, zko.xxh64, **kw)
        frames.append((len(f), len(out))); comp += f; data += out
    return bytes(comp), frames, bytes(data)


def test_generated_frames_one_at_a_time(engine):
    for seed in range(400):
        f, out, feats = zstd_gen.generate(seed, zko.xxh64)
        o, st = engine.decode_frames(f + b"\0" * 8, [0, len(f)], [0, len(out)], verify=True, raise_on_error=False)
        assert st[0] == 0 and o == out, (seed, int(st[0]), sorted(feats))
        sizes, st = engine.frame_content_sizes(f, [0, len(f)])
        assert st[0] == 0 and int(sizes[0]) == len(out), (seed, sorted(feats))


CHOICES = {
    "by_batch_shape": {},
    "lane_per_block_predef": dict(fse_own=1, fse_shared=1, exec_lanes=256, xxh64=1, small_path=1),
    "quad_fed": dict(fse_own=2, fse_shared=2, exec_lanes=512, xxh64=2, small_path=1),
    "quad_sets": dict(fse_own=2, fse_shared=3, exec_lanes=128, exec_ring=2, xxh64=2, small_path=1),
}


@pytest.mark.parametrize("mode", list(CHOICES))
def test_generated_frames_in_batches(engine, mode):
    """3000 frames in one call (~10 000 blocks: the batch kernels), by batch shape and with every sequence kernel pinned in turn"""
    comp, frames, data = archive(range(5000, 8000))
    c, d = offsets_from_frames(frames)
    try:
        engine.set_kernel_choice(**CHOICES[mode])
        out, st = engine.decode_frames(comp + b"\0" * 8, c, d, verify=True, raise_on_error=False)
    finally:
        engine.set_kernel_choice(reset=0)
    bad = np.flatnonzero(st)
    assert len(bad) == 0, (bad[:5], st[bad[:5]])
    assert out == data
    sizes, st = engine.frame_content_sizes(comp, c)
    assert not st.any() and [int(x) for x in sizes] == [ds for _, ds in frames]


def test_generated_frames_in_small_groups(engine):
    """groups of 2 ... 64 frames (the small-batch kernels with several frames), any subrange"""
    comp, frames, data = archive(range(9000, 9400))
    c, d = offsets_from_frames(frames)
    rng = np.random.default_rng(5)
    for _ in range(60):
        first = int(rng.integers(0, len(frames) - 64)); count = int(rng.integers(1, 65))
        out, st = engine.decode_frames(comp + b"\0" * 8, c, d, first=first, count=count, verify=True, raise_on_error=False)
        assert not st.any() and out == data[int(d[first]):int(d[first + count])], (first, count)


def test_generated_frames_through_the_shim(engine):
    if Z.load("shim") is None:
        pytest.skip("the shim is not built")
    comp, frames, data = archive(range(12000, 12150))
    assert Z.decode_stream(comp, len(data), "shim") == data


def test_damaged_generated_frames_against_the_oracle(engine):
    """one to three flipped bits in a generated frame, checksums not verified: the engine refuses exactly what the oracle refuses and yields its bytes otherwise"""
    comp, frames, data = archive(range(20000, 21000))
    c, d = offsets_from_frames(frames)
    rng = np.random.default_rng(9)
    bad = bytearray(comp)
    hit = set()
    for _ in range(700):
        i = int(rng.integers(0, len(bad)))
        bad[i] ^= 1 << int(rng.integers(0, 8))
        hit.add(int(np.searchsorted(c, i, side="right")) - 1)
    out, st = engine.decode_frames(bytes(bad) + b"\0" * 8, c, d, verify=False, raise_on_error=False)
    refused = 0
    for f in range(len(frames)):
        lo, hi = int(d[f]), int(d[f + 1])
        if f not in hit:
            assert st[f] == 0 and out[lo:hi] == data[lo:hi], f
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
    assert refused > 100


def test_generated_frames_with_long_blocks(engine):
    """thousands of sequences per block, literal sections of tens of KiB, ten blocks per frame: one at a time and 600 side by side"""
    kw = dict(max_blocks=10, max_seq=4000, max_lit=100000)
    for seed in range(100000, 100080):
        f, out, feats = zstd_gen.generate(seed, zko.xxh64, **kw)
        o, st = engine.decode_frames(f + b"\0" * 8, [0, len(f)], [0, len(out)], verify=True, raise_on_error=False)
        assert st[0] == 0 and o == out, (seed, int(st[0]), sorted(feats))
    comp, frames, data = archive(range(100000, 100600), **kw)
    c, d = offsets_from_frames(frames)
    out, st = engine.decode_frames(comp + b"\0" * 8, c, d, verify=True, raise_on_error=False)
    assert not st.any() and out == data
    sizes, st = engine.frame_content_sizes(comp, c)
    assert not st.any() and [int(x) for x in sizes] == [ds for _, ds in frames]


def test_generated_frames_against_a_prefix(engine):
    """offsets across the frame's first byte into a raw-content prefix: one frame at a time, and 300 frames written against ONE prefix in one call"""
    for seed in range(300):
        prefix = zko.gen_text(1 + (seed * 7919) % 90000, seed % 7)
        f, out, feats = zstd_gen.generate(300000 + seed, zko.xxh64, prefix=prefix)
        o, st = engine.decode_frames(f + b"\0" * 8, [0, len(f)], [0, len(out)], verify=True, raise_on_error=False, prefix=prefix)
        assert st[0] == 0 and o == out, (seed, int(st[0]), sorted(feats))
    prefix = zko.gen_text(70000, 3)
    comp, frames, data = archive(range(310000, 310300), prefix=prefix)
    c, d = offsets_from_frames(frames)
    out, st = engine.decode_frames(comp + b"\0" * 8, c, d, verify=True, raise_on_error=False, prefix=prefix)
    assert not st.any() and out == data


def test_generated_frames_with_more_than_0x7F00_sequences_in_a_block(engine):
    """the three-byte Number_of_Sequences (a block of 32 512+ three-byte matches: a sequence per three bytes is what the small path's record
    scratch is sized for): one at a time, then together with 200 ordinary frames in one call"""
    dense = [zstd_gen.generate(500000 + seed, zko.xxh64, dense=True, max_blocks=8) for seed in range(5)]
    for f, out, feats in dense:
        assert "nseq_form3" in feats
        o, st = engine.decode_frames(f + b"\0" * 8, [0, len(f)], [0, len(out)], verify=True, raise_on_error=False)
        assert st[0] == 0 and o == out
        sizes, st = engine.frame_content_sizes(f, [0, len(f)])
        assert st[0] == 0 and int(sizes[0]) == len(out)
    comp, frames, data = archive(range(600000, 600200))
    comp, data = bytearray(comp), bytearray(data)
    for f, out, _ in dense:
        frames.append((len(f), len(out))); comp += f; data += out
    c, d = offsets_from_frames(frames)
    out, st = engine.decode_frames(bytes(comp) + b"\0" * 8, c, d, verify=True, raise_on_error=False)
    assert not st.any() and out == bytes(data)



def test_generated_frames_behind_a_seek_table_through_the_decoder(engine):
    """a seekable archive of foreign frames (400 generated ones + a Foot seek table): zeekstd's Decoder reads it whole, from any offset, between limits"""
    import zeekstd_amd as zk
    comp, frames, data = archive(range(700000, 700400))
    st = zk.SeekTable.new()
    for cs, ds in frames:
        st.log_frame(cs, ds)
    seekable = comp + st.to_bytes()
    d = zk.DecodeOptions(seekable).engine(engine).into_decoder()
    assert d.read_to_end() == data
    rng = np.random.default_rng(4)
    for _ in range(30):
        lo = int(rng.integers(0, len(data) + 1)); hi = int(rng.integers(lo, len(data) + 1))
        d.set_offset(lo); d.set_offset_limit(hi)
        assert d.read_to_end() == data[lo:hi], (lo, hi)
