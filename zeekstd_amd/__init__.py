"""zeekstd_amd -- MI355X-native seekable-zstd engine (drop-in for zeekstd's frame hot path).

Python mirror of the C ABI in include/zeekstd_amd.h.  Requires the in-tree HIP library;
there is no CPU fallback (import fails loudly without libzeekstd_amd.so).
"""
from ._lib import LIB_PATH, error_name, lib  # noqa: F401
from .engine import Engine, ZkError  # noqa: F401
