"""Python mirror of the zeekstd crate API over the C ABI (Level B of include/zeekstd_amd.h).

Same names and behaviour as the Rust reference (paths relative to /root/reference/lib/src):
  SeekTable / Serializer / Format      seek_table.rs
  DecodeOptions / Decoder              decode.rs
  Error kinds                          error.rs
so that the parity tests read like the reference's own tests."""
import ctypes as C
import enum

import numpy as np

from . import _lib
from ._lib import declare, lib
from .engine import Engine

_P = C.c_void_p
_u64p = C.POINTER(C.c_uint64)


class Format(enum.IntEnum):            # seek_table.rs:228-241
    Head = 0
    Foot = 1


class Error(Exception):                # error.rs:4-128
    def __init__(self, code):
        self.code = code
        super().__init__(lib.zk_last_error_message().decode() or _lib.error_name(code))

    def is_offset_out_of_range(self):
        return self.code == -1001

    def is_frame_index_too_large(self):
        return self.code == -1002

    def is_number_conversion_failed(self):
        return self.code == -1003

    def is_io(self):
        return self.code == -1004 or self.code <= -2000

    def is_zstd(self):
        return -1000 < self.code < 0


def _chk(rc):
    if rc < 0:
        raise Error(rc)
    return rc


class zk_decode_opts(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("lower_frame", C.c_uint32), ("upper_frame", C.c_uint32),
                ("offset", C.c_uint64), ("offset_limit", C.c_uint64), ("seek_table", _P), ("batch_bytes", C.c_uint64)]


for _n, _r, _a in [
    ("zk_last_error_message", C.c_char_p, []),
    ("zk_seek_table_new", _P, []), ("zk_seek_table_clone", _P, [_P]), ("zk_seek_table_free", None, [_P]),
    ("zk_seek_table_from_bytes", C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    ("zk_seek_table_from_reader_bytes", C.c_int, [_P, C.c_size_t, C.c_size_t, C.POINTER(_P)]),
    ("zk_seek_table_log_frame", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("zk_seek_table_log_frames", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("zk_seek_table_num_frames", C.c_uint32, [_P]),
    ("zk_seek_table_frame_index_comp", C.c_uint32, [_P, C.c_uint64]),
    ("zk_seek_table_frame_index_decomp", C.c_uint32, [_P, C.c_uint64]),
    ("zk_seek_table_max_frame_size_comp", C.c_uint64, [_P]), ("zk_seek_table_max_frame_size_decomp", C.c_uint64, [_P]),
    ("zk_seek_table_size_comp", C.c_uint64, [_P]), ("zk_seek_table_size_decomp", C.c_uint64, [_P]),
    ("zk_seek_table_equal", C.c_int, [_P, _P]),
    ("zk_seek_table_entries", C.c_size_t, [_P, _P, _P, C.c_size_t]),
    ("zk_seek_table_serializer", _P, [_P, C.c_int]),
    ("zk_serializer_write_into", C.c_size_t, [_P, _P, C.c_size_t]),
    ("zk_serializer_reset", None, [_P]), ("zk_serializer_encoded_len", C.c_size_t, [_P]), ("zk_serializer_free", None, [_P]),
    ("zk_decoder_open_bytes", C.c_int, [_P, _P, C.c_size_t, C.POINTER(zk_decode_opts), C.POINTER(_P)]),
    ("zk_decoder_open_file", C.c_int, [_P, C.c_char_p, C.POINTER(zk_decode_opts), C.POINTER(_P)]),
    ("zk_decoder_open_callbacks", C.c_int, [_P, _P, _P, _P, C.POINTER(zk_decode_opts), C.POINTER(_P)]),
    ("zk_decoder_open_seekable", C.c_int, [_P, _P, _P, _P, _P, C.POINTER(zk_decode_opts), C.POINTER(_P)]),
    ("zk_shard_range", C.c_int, [C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("zk_decode_shard", C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, C.c_int, _P, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint64)]),
    ("zk_decoder_free", None, [_P]),
    ("zk_decoder_decompress", C.c_int64, [_P, _P, C.c_size_t]),
    ("zk_decoder_decompress_with_prefix", C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("zk_decoder_reset", None, [_P]),
    ("zk_decoder_set_lower_frame", C.c_int, [_P, C.c_uint32, _u64p]),
    ("zk_decoder_set_upper_frame", C.c_int, [_P, C.c_uint32, _u64p]),
    ("zk_decoder_set_offset", C.c_int, [_P, C.c_uint64]), ("zk_decoder_set_offset_limit", C.c_int, [_P, C.c_uint64]),
    ("zk_decoder_read_compressed", C.c_uint64, [_P]), ("zk_decoder_offset", C.c_uint64, [_P]),
    ("zk_decoder_offset_limit", C.c_uint64, [_P]), ("zk_decoder_seek_table", _P, [_P]),
    ("zk_decoder_seek", C.c_int, [_P, C.c_int, C.c_int64, _u64p]),
    ("zk_decoder_gpu_submissions", C.c_uint64, [_P]),
]:
    declare(_n, _r, _a)
for _n in ("frame_start_comp", "frame_start_decomp", "frame_end_comp", "frame_end_decomp", "frame_size_comp", "frame_size_decomp"):
    declare("zk_seek_table_" + _n, C.c_int, [_P, C.c_uint32, _u64p])


class Serializer:                      # seek_table.rs:937-1059
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib.zk_serializer_free(self._h)
            self._h = None

    def write_into(self, buf) -> int:
        """buf: a writable buffer (bytearray / numpy uint8). Returns bytes written, 0 == done."""
        n = len(buf)
        arr = (C.c_uint8 * n).from_buffer(buf) if n else None
        return lib.zk_serializer_write_into(self._h, arr, n)

    def reset(self):
        lib.zk_serializer_reset(self._h)

    def encoded_len(self) -> int:
        return lib.zk_serializer_encoded_len(self._h)

    def read(self, n: int = -1) -> bytes:          # impl io::Read
        out = bytearray()
        chunk = bytearray(65536 if n < 0 else n)
        while n < 0 or len(out) < n:
            k = self.write_into(chunk)
            if k == 0:
                break
            out += chunk[:k]
            if n >= 0:
                break
        return bytes(out)


class SeekTable:                       # seek_table.rs:243-935
    def __init__(self, _handle=None):
        self._h = _handle if _handle is not None else lib.zk_seek_table_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib.zk_seek_table_free(self._h)
            self._h = None

    @staticmethod
    def new():
        return SeekTable()

    @staticmethod
    def from_seekable(src: bytes):
        return SeekTable.from_seekable_format(src, Format.Foot)

    @staticmethod
    def from_seekable_format(src: bytes, fmt: Format):
        src = bytes(src)
        h = _P()
        _chk(lib.zk_seek_table_from_bytes(src, len(src), int(fmt), C.byref(h)))
        return SeekTable(h)

    @staticmethod
    def from_reader(data: bytes, max_read: int = 0):
        data = bytes(data)
        h = _P()
        _chk(lib.zk_seek_table_from_reader_bytes(data, len(data), max_read, C.byref(h)))
        return SeekTable(h)

    def clone(self):
        return SeekTable(lib.zk_seek_table_clone(self._h))

    def log_frame(self, c_size: int, d_size: int):
        _chk(lib.zk_seek_table_log_frame(self._h, c_size, d_size))

    def log_frames(self, c_sizes, d_sizes):
        """n frames in one call (zk_seek_table_log_frames): arrays of compressed / decompressed sizes."""
        import numpy as np
        c = np.ascontiguousarray(np.asarray(c_sizes, dtype=np.uint32)); d = np.ascontiguousarray(np.asarray(d_sizes, dtype=np.uint32))
        if c.shape != d.shape or c.ndim != 1:
            raise ValueError("log_frames: two equally long 1-d arrays")
        _chk(lib.zk_seek_table_log_frames(self._h, c.size, c.ctypes.data if c.size else None, d.ctypes.data if d.size else None))

    def num_frames(self) -> int:
        return lib.zk_seek_table_num_frames(self._h)

    def frame_index_comp(self, offset: int) -> int:
        return lib.zk_seek_table_frame_index_comp(self._h, offset)

    def frame_index_decomp(self, offset: int) -> int:
        return lib.zk_seek_table_frame_index_decomp(self._h, offset)

    def _idx(self, fn, index):
        v = C.c_uint64()
        _chk(getattr(lib, "zk_seek_table_" + fn)(self._h, index, C.byref(v)))
        return v.value

    def frame_start_comp(self, i): return self._idx("frame_start_comp", i)
    def frame_start_decomp(self, i): return self._idx("frame_start_decomp", i)
    def frame_end_comp(self, i): return self._idx("frame_end_comp", i)
    def frame_end_decomp(self, i): return self._idx("frame_end_decomp", i)
    def frame_size_comp(self, i): return self._idx("frame_size_comp", i)
    def frame_size_decomp(self, i): return self._idx("frame_size_decomp", i)
    def max_frame_size_comp(self): return lib.zk_seek_table_max_frame_size_comp(self._h)
    def max_frame_size_decomp(self): return lib.zk_seek_table_max_frame_size_decomp(self._h)
    def size_comp(self): return lib.zk_seek_table_size_comp(self._h)
    def size_decomp(self): return lib.zk_seek_table_size_decomp(self._h)

    def __eq__(self, other):
        return isinstance(other, SeekTable) and bool(lib.zk_seek_table_equal(self._h, other._h))

    def into_serializer(self) -> Serializer:
        return self.into_format_serializer(Format.Foot)

    def into_format_serializer(self, fmt: Format) -> Serializer:
        return Serializer(lib.zk_seek_table_serializer(self._h, int(fmt)))

    def offsets(self):
        """(c_off, d_off): the n+1 prefix sums as uint64 arrays (what Engine.decode_frames takes)."""
        n = self.num_frames() + 1
        c = np.zeros(n, np.uint64)
        d = np.zeros(n, np.uint64)
        lib.zk_seek_table_entries(self._h, c.ctypes.data, d.ctypes.data, n)
        return c, d

    def to_bytes(self, fmt: Format = Format.Foot) -> bytes:
        ser = self.into_format_serializer(fmt)
        buf = bytearray(ser.encoded_len())
        assert ser.write_into(buf) == len(buf)
        return bytes(buf)


class SeekFrom(enum.IntEnum):
    Start = 0
    End = 1
    Current = 2


class DecodeOptions:                   # decode.rs:13-114
    def __init__(self, src):
        """src: bytes-like (BytesWrapper) or a filesystem path (the Read+Seek blanket impl)."""
        self._src = src
        self._o = zk_decode_opts()
        self._st = None
        self._engine = None

    def engine(self, e: Engine):       # with_dctx / dctx
        self._engine = e
        return self

    def seek_table(self, st: SeekTable):
        self._st = st
        return self

    def lower_frame(self, i):
        self._o.flags |= 4; self._o.lower_frame = i
        return self

    def upper_frame(self, i):
        self._o.flags |= 8; self._o.upper_frame = i
        return self

    def offset(self, off):
        self._o.flags |= 1; self._o.offset = off
        return self

    def offset_limit(self, lim):
        self._o.flags |= 2; self._o.offset_limit = lim
        return self

    def batch_bytes(self, n):
        self._o.batch_bytes = n
        return self

    def verify_checksums(self, v: bool):
        if not v:
            self._o.flags |= 16
        return self

    def into_decoder(self):
        return Decoder(self)


_SEEK_FN = C.CFUNCTYPE(C.c_int64, _P, C.c_int, C.c_int64)
_READ_FN = C.CFUNCTYPE(C.c_int64, _P, _P, C.c_size_t)
_INTEGRITY_FN = C.CFUNCTYPE(C.c_int, _P, C.c_int, _P)


class Decoder:                         # decode.rs:117-579
    def __init__(self, src_or_opts):
        opts = src_or_opts if isinstance(src_or_opts, DecodeOptions) else DecodeOptions(src_or_opts)
        self._engine = opts._engine
        e = opts._engine._h if opts._engine is not None else None
        if opts._st is not None:
            opts._o.seek_table = opts._st._h
        h = _P()
        if isinstance(opts._src, (str,)):
            rc = lib.zk_decoder_open_file(e, opts._src.encode(), C.byref(opts._o), C.byref(h))
            self._keep = None
        elif hasattr(opts._src, "read") and hasattr(opts._src, "seek"):
            # any Read + Seek object: the blanket `impl<T: Read + Seek> Seekable for T` (seekable.rs:112-138) through the
            # callback source of the C ABI
            f = opts._src

            def _seek(_user, whence, value):
                try:
                    return f.seek(value, 0 if whence == 0 else 2)
                except Exception:
                    return -1

            def _read(_user, buf, n):
                try:
                    b = f.read(n)
                    C.memmove(buf, b, len(b))
                    return len(b)
                except Exception:
                    return -1
            if hasattr(f, "seek_table_integrity"):
                # a source that answers Seekable::seek_table_integrity itself (a required method of the trait, seekable.rs:33-38)
                def _integrity(_user, fmt, out):
                    try:
                        b = bytes(f.seek_table_integrity(Format.Head if fmt == 0 else Format.Foot))
                        if len(b) != 9:
                            return -1
                        C.memmove(out, b, 9)
                        return 0
                    except Exception:
                        return -1
                self._keep = (f, _SEEK_FN(_seek), _READ_FN(_read), _INTEGRITY_FN(_integrity))
                rc = lib.zk_decoder_open_seekable(e, self._keep[1], self._keep[2], self._keep[3], None, C.byref(opts._o), C.byref(h))
            else:
                self._keep = (f, _SEEK_FN(_seek), _READ_FN(_read))
                rc = lib.zk_decoder_open_callbacks(e, self._keep[1], self._keep[2], None, C.byref(opts._o), C.byref(h))
        else:
            self._keep = bytes(opts._src)          # the source must outlive the decoder
            rc = lib.zk_decoder_open_bytes(e, self._keep, len(self._keep), C.byref(opts._o), C.byref(h))
        _chk(rc)
        self._h = h

    @staticmethod
    def new(src):
        return Decoder(src)

    def close(self):
        if getattr(self, "_h", None):
            lib.zk_decoder_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def decompress(self, buf) -> int:
        """Fills the writable buffer `buf`; returns the number of bytes written (0 == end of range)."""
        n = len(buf)
        arr = (C.c_uint8 * n).from_buffer(buf) if n else None
        return _chk(lib.zk_decoder_decompress(self._h, arr, n))

    def decompress_with_prefix(self, buf, prefix) -> int:
        n = len(buf)
        arr = (C.c_uint8 * n).from_buffer(buf) if n else None
        out = C.c_size_t()
        p = self._hold_prefix(prefix)
        _chk(lib.zk_decoder_decompress_with_prefix(self._h, arr, n, p, len(p) if p else 0, C.byref(out)))
        return out.value

    def _hold_prefix(self, prefix):
        """The native Decoder keeps only the address of the prefix (like libzstd, decode.rs:201 lifetime bound): the
        same Python object maps to the same immutable copy for as long as this Decoder lives."""
        if prefix is None or len(prefix) == 0:
            return None
        if getattr(self, "_prefix_src", None) is not prefix:
            self._prefix_src = prefix
            self._prefix_copy = prefix if isinstance(prefix, bytes) else bytes(prefix)
        return self._prefix_copy

    def read(self, n: int) -> bytes:               # impl io::Read
        buf = bytearray(n)
        return bytes(buf[:self.decompress(buf)])

    def read_to_end(self) -> bytes:
        out = bytearray()
        buf = bytearray(1 << 20)
        while True:
            k = self.decompress(buf)
            if k == 0:
                return bytes(out)
            out += buf[:k]

    def reset(self):
        lib.zk_decoder_reset(self._h)

    def set_lower_frame(self, i) -> int:
        v = C.c_uint64()
        _chk(lib.zk_decoder_set_lower_frame(self._h, i, C.byref(v)))
        return v.value

    def set_upper_frame(self, i) -> int:
        v = C.c_uint64()
        _chk(lib.zk_decoder_set_upper_frame(self._h, i, C.byref(v)))
        return v.value

    def set_offset(self, off):
        _chk(lib.zk_decoder_set_offset(self._h, off))

    def set_offset_limit(self, lim):
        _chk(lib.zk_decoder_set_offset_limit(self._h, lim))

    def read_compressed(self): return lib.zk_decoder_read_compressed(self._h)
    def offset(self): return lib.zk_decoder_offset(self._h)
    def offset_limit(self): return lib.zk_decoder_offset_limit(self._h)
    def gpu_submissions(self): return lib.zk_decoder_gpu_submissions(self._h)

    def seek_table(self) -> SeekTable:
        return SeekTable(lib.zk_decoder_seek_table(self._h))

    def seek(self, whence: SeekFrom, n: int) -> int:    # impl io::Seek
        v = C.c_uint64()
        _chk(lib.zk_decoder_seek(self._h, int(whence), n, C.byref(v)))
        return v.value


# ---------------------------------------------------------------- encode side (encode.rs)
class FrameSizePolicy:                 # encode.rs:21-39
    def __init__(self, kind, size):
        self.kind, self.size = kind, size

    @staticmethod
    def Compressed(n):
        return FrameSizePolicy(1, n)

    @staticmethod
    def Uncompressed(n):
        return FrameSizePolicy(0, n)

    @staticmethod
    def default():
        return FrameSizePolicy(0, 0x200000)


class zk_encode_opts(C.Structure):
    _fields_ = [("policy", C.c_uint32), ("frame_size", C.c_uint32), ("level", C.c_int32), ("checksum", C.c_int32),
                ("batch_frames", C.c_uint32)]


_WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
_szp = C.POINTER(C.c_size_t)
for _n, _r, _a in [
    ("zk_raw_encoder_new", C.c_int, [_P, C.POINTER(zk_encode_opts), C.POINTER(_P)]),
    ("zk_raw_encoder_free", None, [_P]),
    ("zk_raw_encoder_compress", C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _szp, _szp]),
    ("zk_raw_encoder_compress_with_prefix", C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, _szp, _szp]),
    ("zk_raw_encoder_end_frame", C.c_int, [_P, _P, C.c_size_t, _szp, _szp]),
    ("zk_raw_encoder_seek_table", _P, [_P]),
    ("zk_raw_encoder_reset_frame", None, [_P]), ("zk_raw_encoder_reset_seek_table", None, [_P]),
    ("zk_encoder_new", C.c_int, [_P, C.POINTER(zk_encode_opts), _WRITE_FN, _P, C.POINTER(_P)]),
    ("zk_encoder_free", None, [_P]),
    ("zk_encoder_compress", C.c_int64, [_P, _P, C.c_size_t]),
    ("zk_encoder_compress_with_prefix", C.c_int64, [_P, _P, C.c_size_t, _P, C.c_size_t]),
    ("zk_encoder_end_frame", C.c_int64, [_P]),
    ("zk_encoder_flush", C.c_int, [_P]),
    ("zk_encoder_finish", C.c_int, [_P, C.c_int, _u64p]),
    ("zk_encoder_written_compressed", C.c_uint64, [_P]),
    ("zk_encoder_seek_table", _P, [_P]),
]:
    declare(_n, _r, _a)


class CompressionProgress:             # encode.rs:43-66
    def __init__(self, i, o):
        self._i, self._o = i, o

    def in_progress(self): return self._i
    def out_progress(self): return self._o


class EpilogueProgress:                # encode.rs:69-92
    def __init__(self, o, left):
        self._o, self._l = o, left

    def out_progress(self): return self._o
    def data_left(self): return self._l


class EncodeOptions:                   # encode.rs:110-207
    def __init__(self):
        self._o = zk_encode_opts(0, 0, 0, 0, 0)
        self._engine = None

    @staticmethod
    def new():
        return EncodeOptions()

    def engine(self, e: Engine):       # with_cctx / cctx
        self._engine = e
        return self

    def frame_size_policy(self, p: FrameSizePolicy):
        self._o.policy, self._o.frame_size = p.kind, p.size
        return self

    def checksum_flag(self, flag: bool):
        self._o.checksum = int(flag)
        return self

    def compression_level(self, level: int):
        self._o.level = level
        return self

    def batch_frames(self, n: int):
        self._o.batch_frames = n
        return self

    def into_raw_encoder(self):
        return RawEncoder(self)

    def into_encoder(self, writer):
        return Encoder(writer, self)


def _inbuf(b):
    b = bytes(b) if not isinstance(b, (bytes, bytearray, memoryview)) else b
    return b, len(b)


class RawEncoder:                      # encode.rs:209-545
    def __init__(self, opts: EncodeOptions = None):
        opts = opts or EncodeOptions()
        self._engine = opts._engine
        h = _P()
        _chk(lib.zk_raw_encoder_new(opts._engine._h if opts._engine else None, C.byref(opts._o), C.byref(h)))
        self._h = h

    @staticmethod
    def new():
        return RawEncoder()

    def __del__(self):
        if getattr(self, "_h", None):
            lib.zk_raw_encoder_free(self._h)
            self._h = None

    def compress(self, inp, out) -> CompressionProgress:
        inp = bytes(inp)
        n = len(out)
        arr = (C.c_uint8 * n).from_buffer(out) if n else None
        i, o = C.c_size_t(), C.c_size_t()
        _chk(lib.zk_raw_encoder_compress(self._h, inp, len(inp), arr, n, C.byref(i), C.byref(o)))
        return CompressionProgress(i.value, o.value)

    def compress_with_prefix(self, inp, out, prefix) -> CompressionProgress:
        inp = bytes(inp)
        n = len(out)
        arr = (C.c_uint8 * n).from_buffer(out) if n else None
        i, o = C.c_size_t(), C.c_size_t()
        p = None
        if prefix is not None and len(prefix):
            held = getattr(self, "_prefixes", None)        # the native side keeps only the address until the frame is encoded
            if held is None:
                held = self._prefixes = {}
            if id(prefix) not in held or held[id(prefix)][0] is not prefix:
                held[id(prefix)] = (prefix, prefix if isinstance(prefix, bytes) else bytes(prefix))
            p = held[id(prefix)][1]
        _chk(lib.zk_raw_encoder_compress_with_prefix(self._h, inp, len(inp), arr, n, p, len(p) if p else 0, C.byref(i), C.byref(o)))
        return CompressionProgress(i.value, o.value)

    def end_frame(self, out) -> EpilogueProgress:
        n = len(out)
        arr = (C.c_uint8 * n).from_buffer(out) if n else None
        o, left = C.c_size_t(), C.c_size_t()
        _chk(lib.zk_raw_encoder_end_frame(self._h, arr, n, C.byref(o), C.byref(left)))
        return EpilogueProgress(o.value, left.value)

    def seek_table(self) -> SeekTable:
        return SeekTable(lib.zk_raw_encoder_seek_table(self._h))

    def into_seek_table(self) -> SeekTable:
        return self.seek_table()

    def reset_frame(self):
        lib.zk_raw_encoder_reset_frame(self._h)

    def reset_seek_table(self):
        lib.zk_raw_encoder_reset_seek_table(self._h)


class Encoder:                         # encode.rs:570-800
    def __init__(self, writer, opts: EncodeOptions = None):
        """writer: anything with .write(bytes) (and optionally .flush()) -- io::Write."""
        opts = opts or EncodeOptions()
        self._engine = opts._engine
        self._writer = writer

        def _cb(_user, data, n):
            try:
                writer.write(C.string_at(data, n))
                return 0
            except Exception:           # surfaces as Error::IO
                return 1
        self._cb = _WRITE_FN(_cb)
        h = _P()
        _chk(lib.zk_encoder_new(opts._engine._h if opts._engine else None, C.byref(opts._o), self._cb, None, C.byref(h)))
        self._h = h

    @staticmethod
    def new(writer):
        return Encoder(writer)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.zk_encoder_free(self._h)
            self._h = None

    def compress(self, buf) -> int:
        buf = bytes(buf)
        return _chk(lib.zk_encoder_compress(self._h, buf, len(buf)))

    def compress_with_prefix(self, buf, prefix) -> int:            # encode.rs:641-665
        buf = bytes(buf)
        p = self._hold_prefix(prefix)
        return _chk(lib.zk_encoder_compress_with_prefix(self._h, buf, len(buf), p, len(p) if p else 0))

    def _hold_prefix(self, prefix):
        """The native Encoder keeps only the address of a prefix until the frames begun under it are submitted: every
        distinct prefix object is copied once and kept alive for the life of this Encoder."""
        if prefix is None or len(prefix) == 0:
            return None
        held = getattr(self, "_prefixes", None)
        if held is None:
            held = self._prefixes = {}
        key = id(prefix)
        if key not in held or held[key][0] is not prefix:
            held[key] = (prefix, prefix if isinstance(prefix, bytes) else bytes(prefix))
        return held[key][1]

    write = compress                   # impl io::Write

    def write_all(self, buf):
        self.compress(buf)

    def end_frame(self) -> int:
        return _chk(lib.zk_encoder_end_frame(self._h))

    def flush(self):
        _chk(lib.zk_encoder_flush(self._h))
        if hasattr(self._writer, "flush"):
            self._writer.flush()

    def finish(self) -> int:
        return self.finish_format(Format.Foot)

    def finish_format(self, fmt: Format) -> int:
        v = C.c_uint64()
        _chk(lib.zk_encoder_finish(self._h, int(fmt), C.byref(v)))
        if hasattr(self._writer, "flush"):
            self._writer.flush()
        return v.value

    def written_compressed(self) -> int:
        return lib.zk_encoder_written_compressed(self._h)

    def seek_table(self) -> SeekTable:
        return SeekTable(lib.zk_encoder_seek_table(self._h))

    into_seek_table = seek_table
