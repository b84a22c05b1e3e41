"""zeekstd_amd -- MI355X-native seekable-zstd engine (drop-in for zeekstd's frame hot path).

Python mirror of the C ABI in include/zeekstd_amd.h.  Requires the in-tree HIP library;
there is no CPU fallback (import fails loudly without libzeekstd_amd.so).
"""
from ._lib import LIB_PATH, error_name, lib  # noqa: F401
from .engine import Engine, ZkError  # noqa: F401
from .api import (CompressionProgress, DecodeOptions, Decoder, EncodeOptions, Encoder, EpilogueProgress, Error, Format,  # noqa: F401,E402
                  FrameSizePolicy, RawEncoder, SeekFrom, SeekTable, Serializer)

SEEKABLE_MAGIC_NUMBER = 0x8F92EAB1      # lib.rs:52-58
SEEKABLE_MAX_FRAMES = 0x08000000
SEEK_TABLE_INTEGRITY_SIZE = 9
SEEKABLE_MAX_FRAME_SIZE = 0x40000000
