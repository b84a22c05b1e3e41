import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---------------------------------------------------------------- golden fixtures
class Golden:
    def __init__(self, meta, blob):
        self.meta = meta
        self.name = meta["name"]
        self.comp = blob[meta["offset"]:meta["offset"] + meta["length"]]
        self.frames = [tuple(f) for f in meta["frames"]]

    def input(self):
        from oracle import zko
        data = zko.make_input(self.meta["recipe"])
        assert len(data) == self.meta["input_len"]
        assert f"{zko.xxh64(data):016x}" == self.meta["input_xxh64"], "input generator drifted"
        return data

    def offsets(self):
        c = np.zeros(len(self.frames) + 1, np.uint64)
        d = np.zeros(len(self.frames) + 1, np.uint64)
        c[1:] = np.cumsum([f[0] for f in self.frames])
        d[1:] = np.cumsum([f[1] for f in self.frames])
        return c, d


def load_goldens():
    gdir = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gdir, "archives.json")) as f:
        idx = json.load(f)
    with open(os.path.join(gdir, "archives.bin"), "rb") as f:
        blob = f.read()
    return [Golden(m, blob) for m in idx["cases"]]


GOLDENS = load_goldens()


class PrefixGolden(Golden):
    """Archive whose frames were compressed against a raw-content prefix (tools/make_prefix_goldens.py)."""

    def prefix(self):
        from oracle import zko
        pre = zko.make_input(self.meta["prefix_recipe"])
        assert len(pre) == self.meta["prefix_len"] and f"{zko.xxh64(pre):016x}" == self.meta["prefix_xxh64"]
        return pre


def load_prefix_goldens():
    gdir = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gdir, "prefix_archives.json")) as f:
        idx = json.load(f)
    with open(os.path.join(gdir, "prefix_archives.bin"), "rb") as f:
        blob = f.read()
    return [PrefixGolden(m, blob) for m in idx["cases"]]


PREFIX_GOLDENS = load_prefix_goldens()


@pytest.fixture(params=PREFIX_GOLDENS, ids=[g.name for g in PREFIX_GOLDENS])
def prefix_golden(request):
    return request.param


@pytest.fixture(params=GOLDENS, ids=[g.name for g in GOLDENS])
def golden(request):
    return request.param


def offsets_from_frames(frames):
    c = np.zeros(len(frames) + 1, np.uint64)
    d = np.zeros(len(frames) + 1, np.uint64)
    c[1:] = np.cumsum([f[0] for f in frames])
    d[1:] = np.cumsum([f[1] for f in frames])
    return c, d


# ---------------------------------------------------------------- hand-written frames (tools/make_handmade_goldens.py)
def _load_handmade():
    with open(os.path.join(ROOT, "tests", "golden", "handmade.json")) as f:
        return json.load(f)


_HM = _load_handmade()
def _hm_output(v):
    if v.get("output") is not None:
        return bytes.fromhex(v["output"])
    import base64, zlib
    return zlib.decompress(base64.b64decode(v["output_zlib"]))


HANDMADE = [(k, bytes.fromhex(v["frame"]), _hm_output(v)) for k, v in _HM.items() if not v.get("error")]
# frames libzstd 1.5.7 rejects: (name, frame, Frame_Content_Size, ZSTD_ErrorCode)
HANDMADE_BAD = [(k, bytes.fromhex(v["frame"]), int(v["content_size"]), int(v["error_code"])) for k, v in _HM.items()
                if v.get("error") and not v.get("cpu_only")]
# damaged frames checked on the CPU only (oracle + simulation of the device code)
HANDMADE_BAD_CPU = [(k, bytes.fromhex(v["frame"]), int(v["content_size"]), int(v["error_code"])) for k, v in _HM.items()
                    if v.get("error") and v.get("cpu_only")]


# ---------------------------------------------------------------- CPU simulation of the device lane code
_SIM = None


def sim_lib():
    """tests/sim/zk_sim.cpp compiled with g++: the kernels' per-lane code run on the CPU."""
    global _SIM
    if _SIM is None:
        src = os.path.join(ROOT, "tests", "sim", "zk_sim.cpp")
        so = os.path.join(ROOT, "tests", "sim", "libzk_sim.so")
        hdr = os.path.join(ROOT, "zeekstd_amd", "csrc", "zk_device.h")
        if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
        l = C.CDLL(so)
        l.zk_sim_decode_prefix.restype = C.c_int
        l.zk_sim_decode_prefix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_int, C.c_void_p, C.c_uint64]
        l.zk_sim_decode.restype = C.c_int
        l.zk_sim_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int]
        _SIM = l
    return _SIM


def sim_decode(comp, frames, first=0, count=None, B=16, CH=1024, prefix=None, quad=False, seg=None):
    """quad: blocks with their own FSE tables go through zk_seq_walk_quad (three lock-stepped lanes per block, the
    device's zk_k_fse_quad) instead of the lane-per-block walk.
    seg = (segment bytes, lanes of the fill pass, log2 of output bytes per hole-record slot): the executor in segments
    (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill) instead of one workgroup per frame."""
    sim_lib().zk_sim_set_fse_quad(int(quad))                # 2: the small-batch kernels' walk (8-byte cells, ZkCells64)
    sim_lib().zk_sim_set_exec_seg(*(seg if seg else (0, 64, 2)))
    c, d = offsets_from_frames(frames)
    n = len(frames)
    if count is None:
        count = n - first
    out_len = int(d[first + count] - d[first])
    out = np.zeros(out_len + 1, np.uint8)
    st = np.zeros(max(count, 1), np.int32)
    buf = np.frombuffer(bytes(comp) + b"\0" * 8, np.uint8)
    if prefix:
        pre = np.frombuffer(bytes(prefix), np.uint8)
        rc = sim_lib().zk_sim_decode_prefix(buf.ctypes.data, c.ctypes.data, d.ctypes.data, first, count, out.ctypes.data,
                                            st.ctypes.data, B, CH, pre.ctypes.data, len(pre))
    else:
        rc = sim_lib().zk_sim_decode(buf.ctypes.data, c.ctypes.data, d.ctypes.data, first, count, out.ctypes.data,
                                     st.ctypes.data, B, CH)
    return rc, out[:out_len].tobytes(), st[:count]


_ENC_SIM = None


def enc_sim_lib():
    """tests/sim/zk_enc_sim.cpp compiled with g++: the encoder's match + parse kernel (zk_enc_match.h, the source hipcc
    compiles) run on the CPU, a fiber per lane."""
    global _ENC_SIM
    if _ENC_SIM is None:
        sim = os.path.join(ROOT, "tests", "sim")
        csrc = os.path.join(ROOT, "zeekstd_amd", "csrc")
        so = os.path.join(sim, "libzk_enc_sim.so")
        deps = [os.path.join(sim, "zk_enc_sim.cpp"), os.path.join(sim, "hip_wg_emu.h")] + \
               [os.path.join(csrc, h) for h in ("zk_enc_match.h", "zk_enc_plan.h", "zk_enc_device.h", "zk_device.h")]
        if not os.path.exists(so) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(so):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, deps[0]])
        l = C.CDLL(so)
        l.zk_enc_sim_match.restype = C.c_int
        l.zk_enc_sim_match.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64, C.c_uint32] + [C.c_void_p] * 6 + \
                                      [C.c_uint64, C.c_void_p, C.c_uint64]
        _ENC_SIM = l
    return _ENC_SIM


def enc_sim_match(data, frame_size, level, prefix=None):
    """-> list of (sequences as a uint64 array, literal bytes, block size) per block, from the emulated kernel"""
    data = bytes(data)
    n = len(data)
    cap = n // 1024 + 64 * ((n + frame_size - 1) // frame_size) + 64
    nseq, nlit, bsz = (np.zeros(cap, np.uint32) for _ in range(3))
    seq_at, lit_at = (np.zeros(cap, np.uint64) for _ in range(2))
    seqs = np.zeros(n // 4 + 2 * cap + 64, np.uint64)
    lits = np.zeros(n + 64, np.uint8)
    src = np.frombuffer(data + b"\0" * 8, np.uint8)
    pre = np.frombuffer(bytes(prefix), np.uint8) if prefix else None
    nb = enc_sim_lib().zk_enc_sim_match(src.ctypes.data, n, frame_size, level, pre.ctypes.data if prefix else None, len(prefix) if prefix else 0,
                                        cap, nseq.ctypes.data, nlit.ctypes.data, bsz.ctypes.data, seq_at.ctypes.data, lit_at.ctypes.data,
                                        seqs.ctypes.data, len(seqs), lits.ctypes.data, len(lits))
    assert nb >= 0
    return [(seqs[int(seq_at[b]):int(seq_at[b]) + int(nseq[b])].copy(), lits[int(lit_at[b]):int(lit_at[b]) + int(nlit[b])].tobytes(), int(bsz[b]))
            for b in range(nb)]


# ---------------------------------------------------------------- GPU engine
@pytest.fixture(scope="session")
def engine():
    import zeekstd_amd as zk
    e = zk.Engine(0)
    yield e
    e.close()
