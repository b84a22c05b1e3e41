"""The executor in SEGMENTS (several workgroups per frame: zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill) pinned onto everything the
one-workgroup-per-frame executor is tested on: every golden archive (libzstd 1.5.7's bytes), hand-made frames, live archives of the
box's libzstd and this engine's own, whole and damaged -- and the frames it gives up on (more hole records than a segment's region
holds: executed again by zk_k_exec).  What the reference does with the same bytes: lib/src/decode.rs:242-256."""
import numpy as np
import pytest

from conftest import GOLDENS, HANDMADE, HANDMADE_BAD, offsets_from_frames
from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu

# segment size in KiB (0 = 128), tile width of the segment kernel, ring, which fill kernel, and a general-pipeline host path (small_path=1)
VARIANTS = {
    "seg128_t1024_fill_lds": dict(exec_seg=2, exec_lanes=1024, seg_fill=3, small_path=1),
    "seg128_t256_ring2_fill_mem1024": dict(exec_seg=2, exec_lanes=256, exec_ring=1, seg_fill=1, small_path=1),
    "seg32_t256_ring4_fill_mem256": dict(exec_seg=2, seg_kib=32, exec_lanes=256, exec_ring=2, seg_fill=2, small_path=1),
    "seg1_t1024_fill_lds": dict(exec_seg=2, seg_kib=1, exec_lanes=1024, seg_fill=3, small_path=1),
    "seg4_by_shape": dict(exec_seg=2, seg_kib=4, small_path=1),
}


@pytest.fixture(params=list(VARIANTS), ids=list(VARIANTS))
def seg(request, engine):
    engine.set_kernel_choice(reset=0)
    engine.set_kernel_choice(**VARIANTS[request.param])
    yield engine
    engine.set_kernel_choice(reset=0)


def test_goldens_in_segments(seg):
    for g in GOLDENS:
        data = g.input()
        c, d = g.offsets()
        out, st = seg.decode_frames(g.comp + b"\0" * 8, c, d, verify=True)
        assert not st.any(), g.name
        assert out == data, g.name
    g = next(x for x in GOLDENS if x.name == "text_100B_frames")
    c, d = g.offsets()
    out, _ = seg.decode_frames(g.comp + b"\0" * 8, c, d, first=5, count=40)
    pos = int(c[5])
    for i in range(5, 45):
        cs, ds = g.frames[i]
        assert out[int(d[i] - d[5]):int(d[i + 1] - d[5])] == zko.frame_decode(g.comp[pos:pos + cs], ds, True)[0]
        pos += cs


def test_handmade_frames_in_segments(seg):
    for name, frame, want in HANDMADE:
        out, st = seg.decode_frames(frame + b"\0" * 8, [0, len(frame)], [0, len(want)], verify=True)
        assert not st.any() and out == want, name
    for name, frame, size, code in HANDMADE_BAD:
        out, st = seg.decode_frames(frame + b"\0" * 8, [0, len(frame)], [0, size], verify=True, raise_on_error=False)
        assert st[0] != 0, name


@pytest.mark.parametrize("level,fsize", [(1, 2 << 20), (3, 2 << 20), (1, 65536), (19, 1 << 20)])
def test_live_libzstd_archives_in_segments(seg, level, fsize):
    """128 KiB blocks with tables of their own; at level 3 and up nearly every byte of a segment depends on bytes before it (the
    hole records outgrow their region: those frames are executed again, one workgroup per frame)."""
    data = zko.gen_chunks(6 << 20, 31 + level)
    comp, frames = Z.encode_seekable_frames(data, fsize, level, True)
    c, d = offsets_from_frames(frames)
    out, st = seg.decode_frames(comp + b"\0" * 8, c, d, verify=True)
    assert not st.any()
    assert out == data


@pytest.mark.parametrize("level,fsize", [(1, 2 << 20), (3, 700000), (6, 1 << 20), (1, 4096)])
def test_engine_made_archives_in_segments(seg, level, fsize):
    data = zko.gen_chunks(12 << 20 if fsize >= 65536 else 1 << 20, 77 + level)
    comp, frames = seg.encode_frames(data, fsize, level, True)
    c, d = offsets_from_frames(frames)
    out, st = seg.decode_frames(comp + b"\0" * 8, c, d, verify=True)
    assert not st.any()
    assert out == data
    lo, hi = len(frames) // 3, len(frames) // 3 + max(1, len(frames) // 4)
    out, st = seg.decode_frames(comp + b"\0" * 8, c, d, first=lo, count=hi - lo, verify=False)
    assert out == data[int(d[lo]):int(d[hi])]


def test_structured_inputs_in_segments(seg):
    """Runs, short periods (matches that overlap themselves across tiles and segments), raw and RLE blocks between compressed ones."""
    rng = np.random.default_rng(6)
    parts = [b"\0" * 300000, bytes(rng.integers(0, 256, 200000, dtype=np.uint8)), b"abc" * 100000, zko.gen_chunks(400000, 3),
             b"x" * 7 + b"0123456" * 60000, bytes(rng.integers(0, 4, 300000, dtype=np.uint8)), zko.gen_chunks(300000, 4)]
    data = b"".join(parts)
    for level in (1, 3):
        for fsize in (len(data), 1 << 20):
            comp, frames = Z.encode_seekable_frames(data, fsize, level, True)
            c, d = offsets_from_frames(frames)
            out, st = seg.decode_frames(comp + b"\0" * 8, c, d, verify=True)
            assert not st.any() and out == data, (level, fsize)
        comp, frames = seg.encode_frames(data, 1 << 20, level, True)
        c, d = offsets_from_frames(frames)
        out, st = seg.decode_frames(comp + b"\0" * 8, c, d, verify=True)
        assert not st.any() and out == data, level


def test_damaged_archives_reach_the_frame_executors_verdict(engine):
    """Damaged frames WITHOUT checksums: per frame the same status as one workgroup per frame gives, the same bytes where a frame still decodes."""
    g = next(x for x in GOLDENS if x.name == "text_l1_64k")
    data = zko.gen_chunks(3 << 20, 9)
    comp0, frames = Z.encode_seekable_frames(data, 1 << 20, 1, False)
    c, d = offsets_from_frames(frames)
    rng = np.random.default_rng(41)
    differ = 0
    try:
        for trial in range(60):
            comp = bytearray(comp0)
            for _ in range(2):
                comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
            engine.set_kernel_choice(reset=0)
            engine.set_kernel_choice(exec_seg=1, small_path=1)
            out1, st1 = engine.decode_frames(bytes(comp) + b"\0" * 8, c, d, verify=False, raise_on_error=False)
            engine.set_kernel_choice(exec_seg=2, small_path=1, seg_kib=(128, 16)[trial & 1])
            out2, st2 = engine.decode_frames(bytes(comp) + b"\0" * 8, c, d, verify=False, raise_on_error=False)
            assert [bool(x) for x in st1] == [bool(x) for x in st2], trial
            differ += int(any(st1))
            for f in range(len(frames)):
                if st1[f] == 0:
                    assert out1[int(d[f]):int(d[f + 1])] == out2[int(d[f]):int(d[f + 1])], (trial, f)
    finally:
        engine.set_kernel_choice(reset=0)
    assert differ > 0
