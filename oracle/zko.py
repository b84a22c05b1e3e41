"""TEST INFRASTRUCTURE -- ctypes binding of the CPU oracle (oracle/libzko.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package (zeekstd_amd) must never import it.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_SRCS = ("zstd_oracle.c", "zstd_oracle_enc.c", "gen.c", "cpu_baseline.c")


class FrameStats(C.Structure):
    _fields_ = [
        ("n_blocks", C.c_uint32), ("n_raw", C.c_uint32), ("n_rle", C.c_uint32), ("n_comp", C.c_uint32),
        ("lit_raw", C.c_uint32), ("lit_rle", C.c_uint32), ("lit_huf4", C.c_uint32), ("lit_huf1", C.c_uint32),
        ("lit_treeless", C.c_uint32),
        ("mode_count", (C.c_uint32 * 4) * 3),
        ("n_seq", C.c_uint64),
        ("window_size", C.c_uint32), ("has_checksum", C.c_uint32), ("single_segment", C.c_uint32),
        ("fcs", C.c_uint64), ("fcs_present", C.c_uint32), ("max_offset", C.c_uint32),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libzko.so")
    srcs = [os.path.join(_HERE, f) for f in _SRCS]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libzko.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        l = C.CDLL(build())
        l.zko_xxh64.restype = C.c_uint64
        l.zko_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        l.zko_frame_decode.restype = C.c_int64
        l.zko_frame_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.c_int, C.POINTER(FrameStats)]
        l.zko_frame_decode_prefix.restype = C.c_int64
        l.zko_frame_decode_prefix.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.POINTER(C.c_size_t), C.c_int, C.POINTER(FrameStats), C.c_char_p, C.c_size_t]
        l.zko_gen_text.restype = None
        l.zko_gen_text.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        l.zko_gen_chunks.restype = None
        l.zko_gen_chunks.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        l.zko_gen_random.restype = None
        l.zko_gen_random.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        l.zko_gen_vocab.restype = C.c_void_p
        l.zko_gen_vocab.argtypes = [C.c_int, C.POINTER(C.c_int)]
        if hasattr(l, "zko_frame_encode"):
            l.zko_frame_encode.restype = C.c_int64
            l.zko_frame_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        _LIB = l
    return _LIB


def xxh64(data: bytes, seed: int = 0) -> int:
    data = bytes(data)
    return lib().zko_xxh64(data, len(data), seed)


def gen_text(n: int, seed: int) -> bytes:
    buf = C.create_string_buffer(max(n, 1))
    lib().zko_gen_text(buf, n, seed)
    return buf.raw[:n]


def gen_chunks(total: int, k0: int = 0) -> bytes:
    buf = C.create_string_buffer(max(total, 1))
    lib().zko_gen_chunks(buf, total, k0)
    return buf.raw[:total]


def gen_random(n: int, seed: int) -> bytes:
    buf = C.create_string_buffer(max(n, 1))
    lib().zko_gen_random(buf, n, seed)
    return buf.raw[:n]


def make_input(recipe) -> bytes:
    """Deterministic test inputs named by a small recipe (list of parts), shared by the golden
    fixtures (tests/golden/archives.json) and the GPU tests:
      ["text", n, seed] | ["zeros", n] | ["random", n, seed] | ["rep", hexpattern, count] | ["chunks", n, k0]
      ["records", count, seed, hexconst]            count x (4 random bytes + const)
      ["slices", src_len, src_seed, count, seed, minlen, maxlen, hexsep]
                                                    text source, then count x (random slice of it + separator)"""
    import struct
    out = bytearray()
    for part in recipe:
        kind = part[0]
        if kind == "text":
            out += gen_text(part[1], part[2])
        elif kind == "chunks":
            out += gen_chunks(part[1], part[2])
        elif kind == "zeros":
            out += bytes(part[1])
        elif kind == "random":
            out += gen_random(part[1], part[2])
        elif kind == "rep":
            out += bytes.fromhex(part[1]) * part[2]
        elif kind == "records":
            rnd, const = gen_random(4 * part[1], part[2]), bytes.fromhex(part[3])
            for i in range(part[1]):
                out += rnd[4 * i:4 * i + 4] + const
        elif kind == "slices":
            _, n_src, s_src, count, seed, lo, hi, sep = part
            src, sepb = gen_text(n_src, s_src), bytes.fromhex(sep)
            r = struct.unpack(f"<{2 * count}I", gen_random(8 * count, seed))
            out += src
            for i in range(count):
                o = r[2 * i] % (n_src - hi)
                l = lo + r[2 * i + 1] % (hi - lo)
                out += src[o:o + l] + sepb
        else:
            raise ValueError(kind)
    return bytes(out)


def gen_vocab(i: int) -> bytes:
    n = C.c_int()
    p = lib().zko_gen_vocab(i, C.byref(n))
    return C.string_at(p, n.value)


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle zstd error code {code}")
        self.code = code


def frame_decode(src: bytes, dst_cap: int, verify: bool = True, want_stats: bool = False, prefix: bytes = None):
    """Decode ONE frame at the start of src. Returns (decoded bytes, consumed[, stats]).
    prefix: raw-content prefix referenced for the frame (ZSTD_DCtx_refPrefix semantics)."""
    src = bytes(src)
    out = C.create_string_buffer(max(dst_cap, 1))
    used = C.c_size_t()
    st = FrameStats()
    if prefix:
        prefix = bytes(prefix)
        r = lib().zko_frame_decode_prefix(src, len(src), out, dst_cap, C.byref(used), int(verify), C.byref(st), prefix, len(prefix))
    else:
        r = lib().zko_frame_decode(src, len(src), out, dst_cap, C.byref(used), int(verify), C.byref(st))
    if r < 0:
        raise OracleError(-r)
    if want_stats:
        return out.raw[:r], used.value, st
    return out.raw[:r], used.value


def frame_encode(src: bytes, level: int = 1, checksum: bool = False, prefix: bytes = None) -> bytes:
    src = bytes(src)
    cap = len(src) + (len(src) >> 7) + 1024
    out = C.create_string_buffer(cap)
    if prefix:
        prefix = bytes(prefix)
        l = lib()
        l.zko_frame_encode_prefix.restype = C.c_int64
        l.zko_frame_encode_prefix.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
        r = l.zko_frame_encode_prefix(src, len(src), out, cap, level, int(checksum), prefix, len(prefix))
    else:
        r = lib().zko_frame_encode(src, len(src), out, cap, level, int(checksum))
    if r < 0:
        raise OracleError(-r)
    return out.raw[:r]


def enc_match_debug(src: bytes, level: int = 1, prefix: bytes = None):
    """The twin's matcher result for ONE frame: list of (sequences as uint64 array in the GPU packing, literal bytes)
    per block (zko_enc_match_debug)."""
    import numpy as np
    src = bytes(src)
    n = len(src)
    l = lib()
    l.zko_enc_match_debug.restype = C.c_int64
    l.zko_enc_match_debug.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    cap = n // 1024 + 64
    nseq, nlit = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    seqs, lits = np.zeros(n // 3 + 64, np.uint64), np.zeros(n + 64, np.uint8)
    prefix = bytes(prefix) if prefix else None
    nb = l.zko_enc_match_debug(src, n, level, prefix, len(prefix) if prefix else 0, cap, nseq.ctypes.data, nlit.ctypes.data,
                               seqs.ctypes.data, lits.ctypes.data)
    assert nb >= 0
    out, s0, l0 = [], 0, 0
    for b in range(nb):
        out.append((seqs[s0:s0 + int(nseq[b])].copy(), lits[l0:l0 + int(nlit[b])].tobytes()))
        s0 += int(nseq[b]); l0 += int(nlit[b])
    return out
