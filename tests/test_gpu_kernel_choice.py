"""Every kernel variant of the LARGE-batch decode path under small, exhaustively checked inputs.

A 4 GiB batch selects zk_k_fse_predef_fed / zk_k_exec<256> (rings of 2 T and 4 T records) / zk_k_xxh64_wide by its shape;
nothing a test can afford to check byte for byte against the oracle would ever reach them.  zk_engine_set_kernel_choice
(include/zeekstd_amd.h) pins a variant whatever the batch looks like: each combination below decodes every golden archive
(libzstd 1.5.7's bytes), the prefix goldens, hand-made frames, live archives of the box's libzstd and this engine's own
(shared tables: what the fed / sets kernels are for), checksums verified, output compared with the recipe's input and --
per frame -- with the oracle.  What the reference does with the same bytes: lib/src/decode.rs:242-256."""
import numpy as np
import pytest

from conftest import GOLDENS, HANDMADE, PREFIX_GOLDENS, offsets_from_frames
from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu

# small_path=1: the host-pointer call goes through the general pipeline (the small path has its own entropy kernel)
VARIANTS = {
    "fed_exec256_ring2T_xxhwide": dict(fse_shared=2, exec_lanes=256, exec_ring=1, xxh64=2, small_path=1),
    "fed_exec256_ring4T_xxhwide": dict(fse_shared=2, exec_lanes=256, exec_ring=2, xxh64=2, small_path=1),
    "predef_exec512_xxhnarrow": dict(fse_shared=1, exec_lanes=512, xxh64=1, small_path=1),
    "sets_exec128_xxhwide": dict(fse_shared=3, exec_lanes=128, xxh64=2, small_path=1),
    "fed_quad56_exec1024": dict(fse_shared=2, fse_own=2, exec_lanes=1024, xxh64=2, small_path=1),
    "predef_lane_per_block_exec256": dict(fse_shared=1, fse_own=1, exec_lanes=256, exec_ring=2, xxh64=1, small_path=1),
    "small_path_split_roles": dict(small_path=2),
    # the checksums beside the executor (zk_k_xxh64_follow behind the executor's progress words)
    "fed_exec256_checksums_follow": dict(fse_shared=2, exec_lanes=256, exec_ring=1, xxh64=4, small_path=1),
    "exec256_five_resident_checksums_follow": dict(exec_lanes=256, exec_resident=5, xxh64=4, small_path=1),
    "exec1024_checksums_follow": dict(exec_lanes=1024, xxh64=4, small_path=1),
    # 512-lane tiles (what batches of dense sequence streams get: libzstd's level 3 and up)
    "exec512_wide_checksums": dict(exec_lanes=512, xxh64=2, small_path=1),
    # (r6) the chains' wave fed by producer waves: four frames per workgroup (what large batches take)
    "exec256_checksums_fed4": dict(exec_lanes=256, xxh64=5, small_path=1),
}


@pytest.fixture(params=list(VARIANTS), ids=list(VARIANTS))
def pinned(request, engine):
    engine.set_kernel_choice(reset=0)
    engine.set_kernel_choice(**VARIANTS[request.param])
    yield engine
    engine.set_kernel_choice(reset=0)


def test_unknown_choices_are_refused(engine):
    import zeekstd_amd as zk
    for key, value in [("exec_lanes", 64), ("fse_shared", 4), ("xxh64", 6), ("exec_ring", -1), ("pipe_contexts", 7), ("exec_resident", 3)]:
        with pytest.raises(zk.ZkError):
            engine.set_kernel_choice(**{key: value})
    assert zk.lib.zk_engine_set_kernel_choice(engine._h, 99, 0) != 0
    engine.set_kernel_choice(reset=0)


def test_goldens_under_every_variant(pinned):
    for g in GOLDENS:
        data = g.input()
        c, d = g.offsets()
        out, st = pinned.decode_frames(g.comp + b"\0" * 8, c, d, verify=True)
        assert not st.any(), g.name
        assert out == data, g.name
    # per frame against the oracle (the checker), on the archive with the most frames
    g = next(x for x in GOLDENS if x.name == "text_100B_frames")
    c, d = g.offsets()
    out, _ = pinned.decode_frames(g.comp + b"\0" * 8, c, d)
    pos = 0
    for i, (cs, ds) in enumerate(g.frames):
        assert out[int(d[i]):int(d[i + 1])] == zko.frame_decode(g.comp[pos:pos + cs], ds, True)[0]
        pos += cs


def test_prefix_goldens_under_every_variant(pinned):
    for g in PREFIX_GOLDENS:
        c, d = g.offsets()
        out, st = pinned.decode_frames(g.comp + b"\0" * 8, c, d, verify=True, prefix=g.prefix())
        assert not st.any(), g.name
        assert out == g.input(), g.name


def test_handmade_frames_under_every_variant(pinned):
    for name, frame, want in HANDMADE:
        out, st = pinned.decode_frames(frame + b"\0" * 8, [0, len(frame)], [0, len(want)], verify=True)
        assert not st.any() and out == want, name


@pytest.mark.parametrize("level,fsize", [(1, 65536), (3, 2 << 20), (1, 2 << 20)])
def test_live_libzstd_archives_under_every_variant(pinned, level, fsize):
    """Archives the reference's Encoder loop writes with the box's libzstd (per-block tables, 128 KiB blocks, 64 KiB frames as
    one block): 12 MiB of the 8d text, more blocks than a workgroup of any sequence kernel holds."""
    data = zko.gen_chunks(12 << 20, 11 + level)
    comp, frames = Z.encode_seekable_frames(data, fsize, level, True)
    c, d = offsets_from_frames(frames)
    out, st = pinned.decode_frames(comp + b"\0" * 8, c, d, verify=True)
    assert not st.any()
    assert out == data


@pytest.mark.parametrize("level,fsize", [(1, 2 << 20), (3, 65536), (1, 4096), (6, 1 << 20)])
def test_engine_made_archives_under_every_variant(pinned, level, fsize):
    """What this engine's encoder writes: one set of FSE tables per frame, Repeat_Mode in the later blocks -- the shape the
    shared-table kernels (predef / fed / sets) serve a lane per block."""
    data = zko.gen_chunks(16 << 20 if fsize >= 65536 else 2 << 20, 5 + level)
    comp, frames = pinned.encode_frames(data, fsize, level, True)
    c, d = offsets_from_frames(frames)
    out, st = pinned.decode_frames(comp + b"\0" * 8, c, d, verify=True)
    assert not st.any()
    assert out == data
    # a frame range from the middle, and a damaged checksum that must be reported by whichever checksum kernel is pinned
    lo, hi = len(frames) // 3, len(frames) // 3 + max(1, len(frames) // 4)
    out, st = pinned.decode_frames(comp + b"\0" * 8, c, d, first=lo, count=hi - lo, verify=True)
    assert out == data[int(d[lo]):int(d[hi])]
    bad = bytearray(comp)
    bad[int(c[lo + 1]) - 1] ^= 0x10
    out, st = pinned.decode_frames(bytes(bad) + b"\0" * 8, c, d, first=lo, count=hi - lo, verify=True, raise_on_error=False)
    assert st[0] == 22 and not st[1:].any()


def test_checksums_beside_the_executor(engine):
    """zk_k_xxh64_follow on device buffers that are REUSED for different contents (the hand-off crosses CUs and XCDs: a checksum wave
    that read a stale line would not verify its frame -- the pass behind the executor would, and `checksums_followed` would say so).
    Two archives of the same geometry decoded in turn into one output buffer: every byte, every status, and (nearly) every frame
    verified by the waves beside the executor; then a damaged checksum and a damaged payload, which only the pass behind the executor may report."""
    import torch
    dev = torch.device("cuda", 0)
    fs, nf = 1 << 20, 48
    srcs = [zko.gen_chunks(nf * fs, 0x7100 + i) for i in range(2)]
    arch = []
    for data in srcs:
        comp, frames = engine.encode_frames(data, fs, 1, True)
        c, d = offsets_from_frames(frames)
        arch.append((torch.from_numpy(np.frombuffer(comp + b"\0" * 64, np.uint8).copy()).to(dev), len(comp),
                     torch.from_numpy(c.view(np.int64)).to(dev), torch.from_numpy(d.view(np.int64)).to(dev), c))
    d_src = [torch.from_numpy(np.frombuffer(x, np.uint8).copy()).to(dev) for x in srcs]
    d_out = torch.zeros(nf * fs + 64, dtype=torch.uint8, device=dev)
    d_st = torch.full((nf,), -1, dtype=torch.int32, device=dev)
    engine.set_kernel_choice(reset=0)
    try:
        engine.set_kernel_choice(xxh64=4)
        for turn in range(6):
            k = turn & 1
            d_comp, csize, d_c, d_d, _ = arch[k]
            assert engine.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, nf * fs, True, d_st) == 0
            assert int(d_st.abs().sum().item()) == 0 and torch.equal(d_out[:nf * fs], d_src[k])
            # (how many frames the waves beside the executor get is the dispatcher's habit -- all 48 of them, on every box so far --,
            #  not a promise: what they leave is verified behind the executor.  The test insists on "some")
            assert engine.checksums_followed() >= 1, (turn, engine.checksums_followed())
        engine.set_kernel_choice(xxh64=4)
        d_comp, csize, d_c, d_d, c = arch[0]
        for at, code in ((int(c[8]) - 1, 22), (int(c[20]) - 9, None)):       # a stored checksum; a payload byte in front of it
            d_comp[at] = d_comp[at] ^ 4
            rc = engine.decode_frames_dev(d_comp, csize, d_c, d_d, 0, nf, d_out, nf * fs, True, d_st)
            st = d_st.cpu().numpy()
            f = 7 if code else 19
            assert rc < 0 and st[f] != 0 and not np.delete(st, f).any() and (code is None or st[f] == code)
            assert engine.checksums_followed() <= nf - 1          # never the damaged frame
            d_comp[at] = d_comp[at] ^ 4
    finally:
        engine.set_kernel_choice(reset=0)
