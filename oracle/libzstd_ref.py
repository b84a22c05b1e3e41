"""TEST INFRASTRUCTURE -- the *real* libzstd, loaded from the image (never shipped),
driven with exactly the call sequence and buffer sizes of the reference:
  RawEncoder::compress_with_prefix hot loop  lib/src/encode.rs:340-346
  RawEncoder::end_frame epilogue loop        lib/src/encode.rs:442-464
  Decoder::decompress_with_prefix hot loop   lib/src/decode.rs:242-256
libzstd 1.5.7 is the version the reference pins (Cargo.lock:1192-1193); the only
1.5.7 build in the image is pillow's bundled one (slow build: oracle only, never timed).
The distro 1.4.8 is the timing baseline (bench.py cpu_baseline) and a second decoder
that must accept GPU-encoded frames.
"""
import ctypes as C
import glob
import os

CSTREAM_OUT = 131591   # ZSTD_CStreamOutSize  (encode.rs:599)
DSTREAM_IN = 131075    # ZSTD_DStreamInSize   (decode.rs:181)
DSTREAM_OUT = 131072   # ZSTD_DStreamOutSize  (decode.rs:184)


class InBuf(C.Structure):
    _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


class OutBuf(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


_CANDIDATES = {
    "1.5.7": sorted(glob.glob("/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd-*.so.1.5.7")),
    "system": ["/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1"],
    # the library UNDER TEST where a test says so: the product's Level-C shim (libzstd's symbols over the GPU engine), driven with the very
    # call sequences below -- the unmodified crate's FFI, exercised end to end (tests/test_gpu_levelc.py)
    "shim": [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zeekstd_amd", "libzstd_zeekstd_amd.so")],
}
_cache = {}


def load(which="system"):
    """which: '1.5.7' (pinned oracle build) or 'system' (fast distro build). Returns None if absent."""
    if which in _cache:
        return _cache[which]
    l = None
    for p in _CANDIDATES[which]:
        try:
            l = C.CDLL(p)
            break
        except OSError:
            continue
    if l is not None:
        l.ZSTD_versionString.restype = C.c_char_p
        l.ZSTD_createCCtx.restype = C.c_void_p
        l.ZSTD_createDCtx.restype = C.c_void_p
        l.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        l.ZSTD_freeDCtx.argtypes = [C.c_void_p]
        l.ZSTD_CCtx_setParameter.restype = C.c_size_t
        l.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.ZSTD_CCtx_reset.restype = C.c_size_t
        l.ZSTD_CCtx_reset.argtypes = [C.c_void_p, C.c_int]
        l.ZSTD_DCtx_reset.restype = C.c_size_t
        l.ZSTD_DCtx_reset.argtypes = [C.c_void_p, C.c_int]
        l.ZSTD_compressStream2.restype = C.c_size_t
        l.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(OutBuf), C.POINTER(InBuf), C.c_int]
        l.ZSTD_decompressStream.restype = C.c_size_t
        l.ZSTD_decompressStream.argtypes = [C.c_void_p, C.POINTER(OutBuf), C.POINTER(InBuf)]
        l.ZSTD_CCtx_refPrefix.restype = C.c_size_t
        l.ZSTD_CCtx_refPrefix.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.ZSTD_DCtx_refPrefix.restype = C.c_size_t
        l.ZSTD_DCtx_refPrefix.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.ZSTD_DCtx_setParameter.restype = C.c_size_t
        l.ZSTD_DCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        l.ZSTD_isError.restype = C.c_uint
        l.ZSTD_isError.argtypes = [C.c_size_t]
        l.ZSTD_getErrorName.restype = C.c_char_p
        l.ZSTD_getErrorName.argtypes = [C.c_size_t]
    _cache[which] = l
    return l


def version(which="system"):
    l = load(which)
    return l.ZSTD_versionString().decode() if l else None


class ZstdError(Exception):
    pass


def _chk(l, r):
    if l.ZSTD_isError(r):
        raise ZstdError(l.ZSTD_getErrorName(r).decode())
    return r


def encode_seekable_frames(data: bytes, frame_size: int, level: int = 1, checksum: bool = False,
                           which: str = "system", prefix: bytes = None, window_log: int = 0, ldm: bool = False):
    """Reference Encoder loop (Uncompressed(frame_size) policy). Returns (payload bytes, [(c,d)...]).

    An empty input yields one empty frame (Encoder::finish always calls end_frame, encode.rs:755-757).
    prefix: referenced at the start of every frame (encode.rs:334-338); window_log: ZSTD_c_windowLog, what the CLI
    sets so that the window covers the prefix (cli/src/compress.rs:31-37)."""
    l = load(which)
    cctx = l.ZSTD_createCCtx()
    _chk(l, l.ZSTD_CCtx_setParameter(cctx, 100, level))
    _chk(l, l.ZSTD_CCtx_setParameter(cctx, 201, int(checksum)))
    if window_log:
        _chk(l, l.ZSTD_CCtx_setParameter(cctx, 101, window_log))
    if ldm:
        _chk(l, l.ZSTD_CCtx_setParameter(cctx, 160, 1))             # ZSTD_c_enableLongDistanceMatching (cli/src/compress.rs:35-36)
    pbuf = C.create_string_buffer(bytes(prefix), len(prefix)) if prefix else None
    data = bytes(data)
    src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
    base = C.addressof(src)
    ob = C.create_string_buffer(CSTREAM_OUT)
    out = bytearray()
    frames = []
    pos = 0
    n = len(data)
    first = True
    while pos < n or first:
        first = False
        d = min(frame_size, n - pos)
        c = 0
        inb = InBuf(base + pos, d, 0)
        if pbuf is not None:                            # encode.rs:334-338
            _chk(l, l.ZSTD_CCtx_refPrefix(cctx, C.addressof(pbuf), len(prefix)))
        while inb.pos < d:                              # encode.rs:340-346
            outb = OutBuf(C.addressof(ob), CSTREAM_OUT, 0)
            _chk(l, l.ZSTD_compressStream2(cctx, C.byref(outb), C.byref(inb), 0))
            out += ob.raw[:outb.pos]
            c += outb.pos
        while True:                                     # encode.rs:442-464
            empty = InBuf(None, 0, 0)
            outb = OutBuf(C.addressof(ob), CSTREAM_OUT, 0)
            r = _chk(l, l.ZSTD_compressStream2(cctx, C.byref(outb), C.byref(empty), 2))
            out += ob.raw[:outb.pos]
            c += outb.pos
            if r == 0:
                break
        _chk(l, l.ZSTD_CCtx_reset(cctx, 1))             # encode.rs:504-506
        frames.append((c, d))
        pos += d
    l.ZSTD_freeCCtx(cctx)
    return bytes(out), frames


def encode_oneshot_frames(data: bytes, frame_size: int, level: int = 3, checksum: bool = False, which: str = "system"):
    """One ZSTD_compress2 call per frame: what tools other than zeekstd write (`zstd`, ZSTD_compress): frames that carry
    Frame_Content_Size and, when small, the Single_Segment flag -- zeekstd's own streaming frames carry neither.
    Returns (payload, [(c, d), ...])."""
    l = load(which)
    l.ZSTD_compress2.restype = C.c_size_t
    l.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    l.ZSTD_compressBound.restype = C.c_size_t
    l.ZSTD_compressBound.argtypes = [C.c_size_t]
    cctx = l.ZSTD_createCCtx()
    _chk(l, l.ZSTD_CCtx_setParameter(cctx, 100, level))
    _chk(l, l.ZSTD_CCtx_setParameter(cctx, 201, int(checksum)))
    data = bytes(data)
    out, frames = bytearray(), []
    for pos in range(0, max(len(data), 1), frame_size):
        chunk = data[pos:pos + frame_size]
        dst = C.create_string_buffer(l.ZSTD_compressBound(len(chunk)) + 64)
        c = _chk(l, l.ZSTD_compress2(cctx, dst, len(dst), chunk, len(chunk)))
        out += dst.raw[:c]
        frames.append((c, len(chunk)))
    l.ZSTD_freeCCtx(cctx)
    return bytes(out), frames


def decode_stream(comp: bytes, expect: int = -1, which: str = "system", prefix: bytes = None, window_log_max: int = 0) -> bytes:
    """Reference Decoder hot loop over a whole payload (concatenated frames; skippable frames skipped).
    prefix: referenced before the first frame and again after every frame end (decode.rs:212-214, 248-255)."""
    l = load(which)
    dctx = l.ZSTD_createDCtx()
    if window_log_max:
        _chk(l, l.ZSTD_DCtx_setParameter(dctx, 100, window_log_max))     # cli/src/decompress.rs:56
    pbuf = C.create_string_buffer(bytes(prefix), len(prefix)) if prefix else None
    if pbuf is not None:
        _chk(l, l.ZSTD_DCtx_refPrefix(dctx, C.addressof(pbuf), len(prefix)))
    comp = bytes(comp)
    src = C.create_string_buffer(comp, len(comp)) if comp else C.create_string_buffer(1)
    base = C.addressof(src)
    ob = C.create_string_buffer(DSTREAM_OUT)
    out = bytearray()
    pos = 0
    try:
        while pos < len(comp):
            take = min(DSTREAM_IN, len(comp) - pos)
            inb = InBuf(base + pos, take, 0)
            while inb.pos < take:                       # decode.rs:242-256
                outb = OutBuf(C.addressof(ob), DSTREAM_OUT, 0)
                r = _chk(l, l.ZSTD_decompressStream(dctx, C.byref(outb), C.byref(inb)))
                out += ob.raw[:outb.pos]
                if r == 0 and pbuf is not None:         # frame end: decode.rs:248-255
                    _chk(l, l.ZSTD_DCtx_reset(dctx, 1))
                    _chk(l, l.ZSTD_DCtx_refPrefix(dctx, C.addressof(pbuf), len(prefix)))
            pos += take
        while True:                                     # drain what is still buffered inside the DCtx
            inb = InBuf(base, 0, 0)
            outb = OutBuf(C.addressof(ob), DSTREAM_OUT, 0)
            _chk(l, l.ZSTD_decompressStream(dctx, C.byref(outb), C.byref(inb)))
            if outb.pos == 0:
                break
            out += ob.raw[:outb.pos]
    finally:
        l.ZSTD_freeDCtx(dctx)
    assert expect < 0 or len(out) == expect, (len(out), expect)
    return bytes(out)


def decode_stream_verdict(comp: bytes, which: str = "system"):
    """The same loop for input that may be DAMAGED: (bytes handed out so far, state) with state "end" (the last call returned 0: the input
    ends on a frame end), "more" (the input ran out inside a frame -- the library still waits) or the error's text."""
    l = load(which)
    dctx = l.ZSTD_createDCtx()
    comp = bytes(comp)
    src = C.create_string_buffer(comp, len(comp)) if comp else C.create_string_buffer(1)
    base = C.addressof(src)
    ob = C.create_string_buffer(DSTREAM_OUT)
    out = bytearray()
    pos, r = 0, 0
    try:
        while pos < len(comp):
            take = min(DSTREAM_IN, len(comp) - pos)
            inb = InBuf(base + pos, take, 0)
            while inb.pos < take:
                outb = OutBuf(C.addressof(ob), DSTREAM_OUT, 0)
                r = _chk(l, l.ZSTD_decompressStream(dctx, C.byref(outb), C.byref(inb)))
                out += ob.raw[:outb.pos]
            pos += take
        while r != 0:                                   # what the context still holds
            inb = InBuf(base, 0, 0)
            outb = OutBuf(C.addressof(ob), DSTREAM_OUT, 0)
            r = _chk(l, l.ZSTD_decompressStream(dctx, C.byref(outb), C.byref(inb)))
            if outb.pos == 0:
                break
            out += ob.raw[:outb.pos]
    except ZstdError as e:
        return bytes(out), str(e)
    finally:
        l.ZSTD_freeDCtx(dctx)
    return bytes(out), "end" if r == 0 else "more"


def judge_damaged(a, b, original: bytes, format_refuses=None):
    """a, b: decode_stream_verdict of the real libzstd and of a library under test that decodes WHOLE frames (the Level-C shim), on the same
    damaged stream of which `original` is the undamaged content.  None when b behaves like a, else what is wrong:
       a reaches the end of the stream -> so does b, with the same bytes -- or b refuses AND format_refuses() says the stream is not valid
                                          zstd: libzstd is laxer than the format in places (1.4.8 does not look at the reserved bits of the
                                          Symbol_Compression_Modes byte, 1.5.6+ does; the double-symbol Huffman decoder of every version lets
                                          a stream end a few bits early: HUF_decodeLastSymbolX2 clamps the bit count of the last symbol);
       a still waits for input         -> b waits too or refuses, and what it handed out is a prefix of a's bytes;
       a refuses                       -> b refuses or still waits, and handed out no byte a did not (or: the damaged bit is one the distro's
                                          1.4.8 is stricter about than the format -- b's bytes are then the archive's own)."""
    (oa, sa), (ob, sb) = a, b
    if sa == "end":
        if sb == "end" and ob == oa:
            return None
        if sb not in ("end", "more") and oa[:len(ob)] == ob and format_refuses is not None and format_refuses():
            return None
        return "libzstd decodes the stream; under test: %s, %d bytes of %d" % (sb, len(ob), len(oa))
    if sb == "end":
        return None if (sa != "more" and ob == original) else "decodes a stream libzstd does not (%s)" % sa
    return None if oa[:len(ob)] == ob else "handed out bytes libzstd did not"
