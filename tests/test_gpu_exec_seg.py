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


def test_frame_lists_and_device_pointers_in_segments(engine):
    """zk_decode_frame_list_dev (frames by index, any order, repeats, packed output) and zk_decode_frames_dev with the executor in segments:
    long frames of an engine-made and a libzstd-made archive, by shape (<= 32 long frames without checksums to verify) and pinned; and two
    batches in flight (zk_decode_submit_dev) whose contexts each own their segment scratch."""
    import torch
    dev = torch.device("cuda:0")
    data = zko.gen_chunks(24 << 20, 123)
    fs = 2 << 20
    archives = [engine.encode_frames(data, fs, 1, False), Z.encode_seekable_frames(data, fs, 3, False)]
    try:
        for comp, frames in archives:
            c, d = offsets_from_frames(frames)
            d_comp = torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev)
            d_c = torch.from_numpy(c.view(np.int64)).to(dev)
            d_d = torch.from_numpy(d.view(np.int64)).to(dev)
            ids = np.array([7, 0, 11, 3, 3, 9], np.uint32)
            off = np.zeros(len(ids) + 1, np.uint64)
            off[1:] = np.cumsum(d[ids.astype(np.int64) + 1] - d[ids.astype(np.int64)])
            d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
            d_off = torch.from_numpy(off.view(np.int64)).to(dev)
            for choice in ({}, {"exec_seg": 2, "exec_lanes": 256}, {"exec_seg": 2, "seg_kib": 48, "seg_fill": 1}):
                engine.set_kernel_choice(reset=0)
                engine.set_kernel_choice(**choice)
                d_out = torch.full((int(off[-1]) + 64,), 0x77, dtype=torch.uint8, device=dev)
                d_st = torch.full((len(ids),), -1, dtype=torch.int32, device=dev)
                assert engine.decode_frame_list_dev(d_comp, len(comp), d_c, d_d, d_ids, d_off, len(ids), d_out, int(off[-1]), True, d_st) == 0
                assert int(d_st.abs().sum().item()) == 0
                out = bytes(d_out[:int(off[-1])].cpu().numpy())
                for i, f in enumerate(ids):
                    assert out[int(off[i]):int(off[i + 1])] == data[int(d[f]):int(d[f + 1])], (choice, i)
                # a frame range from the middle through the device-pointer call
                n = int(d[9] - d[4])
                d_o2 = torch.full((n + 64,), 0x33, dtype=torch.uint8, device=dev)
                d_s2 = torch.full((5,), -1, dtype=torch.int32, device=dev)
                assert engine.decode_frames_dev(d_comp, len(comp), d_c, d_d, 4, 5, d_o2, n, False, d_s2) == 0
                assert bytes(d_o2[:n].cpu().numpy()) == data[int(d[4]):int(d[9])] and int(d_s2.abs().sum().item()) == 0
            # two batches in flight, both by shape in segments
            engine.set_kernel_choice(reset=0)
            n = int(d[6] - d[0])
            outs = [torch.zeros(n + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
            sts = [torch.zeros(6, dtype=torch.int32, device=dev) for _ in range(2)]
            s0 = engine.decode_submit_dev(d_comp, len(comp), d_c, d_d, 0, 6, outs[0], n, False, sts[0])
            s1 = engine.decode_submit_dev(d_comp, len(comp), d_c, d_d, 6, 6, outs[1], n, False, sts[1])
            assert engine.decode_wait(s0) == 0 and engine.decode_wait(s1) == 0
            assert bytes(outs[0][:n].cpu().numpy()) == data[:n] and bytes(outs[1][:n].cpu().numpy()) == data[n:2 * n]
    finally:
        engine.set_kernel_choice(reset=0)


def test_seeks_into_short_frames_in_segments(engine):
    """The small path's short-frame case (a seek into 64 KiB frames of sixteen 4 KiB blocks: a segment per block, one turn of the fill pass per
    frame) through the Decoder handle: random offsets and lengths against the input, unverified -- and the same reads with the executor
    per frame."""
    from zeekstd_amd import DecodeOptions, SeekTable
    data = zko.gen_chunks(6 << 20, 321)
    comp, frames = engine.encode_frames(data, 65536, 1, False)
    st = SeekTable.new()
    for c_, d_ in frames:
        st.log_frame(c_, d_)
    arch = comp + st.to_bytes()
    rng = np.random.default_rng(77)
    try:
        for choice in ({}, {"exec_seg": 1}, {"exec_seg": 2, "seg_kib": 8}):
            engine.set_kernel_choice(reset=0)
            engine.set_kernel_choice(**choice)
            dec = DecodeOptions(arch).engine(engine).into_decoder()
            buf = bytearray(200000)
            for _ in range(150):
                off = int(rng.integers(0, len(data) - 1))
                ln = int(rng.integers(1, 150000 if rng.integers(0, 4) == 0 else 9000))
                lim = min(len(data), off + ln)
                dec.set_offset(off); dec.set_offset_limit(lim)
                got = bytearray()
                while True:
                    k = dec.decompress(buf)
                    if k == 0:
                        break
                    got += buf[:k]
                assert bytes(got) == data[off:lim], (choice, off, lim)
    finally:
        engine.set_kernel_choice(reset=0)
