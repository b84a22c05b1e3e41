"""ctypes loader of libzeekstd_amd.so (the C ABI of include/zeekstd_amd.h).

There is deliberately no fallback: if the shared library (built in-tree by
zeekstd_amd/csrc/Makefile / __graft_entry__.build) is missing, importing fails loudly.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZEEKSTD_AMD_LIB") or os.path.join(_HERE, "libzeekstd_amd.so")   # the override is for A/B kernel experiments

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C zeekstd_amd/csrc` (or __graft_entry__.build()). "
        "zeekstd_amd has no CPU fallback.")

# One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as
# /opt/rocm's).  If torch is going to be used for device buffers in this process it has to be loaded
# first so that our DT_NEEDED libamdhip64.so.7 binds to that already-loaded runtime instead of pulling
# in a second one (two runtimes => "No HIP GPUs are available" in whichever initialises second).
if "torch" not in sys.modules and not os.environ.get("ZEEKSTD_AMD_NO_TORCH"):
    try:
        import torch  # noqa: F401
    except ImportError:
        pass

lib = C.CDLL(LIB_PATH)

_P = C.c_void_p
_sig = {
    "zk_abi_version": (C.c_int, []),
    "zk_error_name": (C.c_char_p, [C.c_int]),
    "zk_engine_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "zk_engine_destroy": (None, [_P]),
    "zk_engine_last_hip_error": (C.c_char_p, [_P]),
    "zk_engine_device_name": (C.c_char_p, [_P]),
    "zk_engine_set_profiling": (C.c_int, [_P, C.c_int]),
    "zk_engine_set_fse_kernel": (C.c_int, [_P, C.c_int]),
    "zk_engine_set_kernel_choice": (C.c_int, [_P, C.c_int, C.c_int]),
    "zk_engine_checksums_followed": (C.c_uint64, [_P]),
    "zk_engine_kernel_count": (C.c_int, []),
    "zk_engine_kernel_name": (C.c_char_p, [C.c_int]),
    "zk_engine_kernel_times": (C.c_int, [_P, _P, C.c_int]),
    "zk_decode_frames": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_int, _P]),
    "zk_decode_frames_prefix": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, _P, C.c_uint64, C.c_int, _P]),
    "zk_decode_frames_prefix_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, _P, C.c_uint64, C.c_int, _P, _P]),
    "zk_decode_frames_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_int, _P, _P]),
    "zk_decode_submit_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_int, _P, C.POINTER(C.c_int)]),
    "zk_decode_wait": (C.c_int, [_P, C.c_int]),
    "zk_compress_bound": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "zk_encode_frames": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.c_int, _P, C.c_uint64, _P, _P, C.c_uint32,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "zk_encode_frames_prefix": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.c_int, _P, C.c_uint64, _P, C.c_uint64, _P, _P, C.c_uint32,
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "zk_encode_frames_dev": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32, C.c_int, C.c_int, _P, C.c_uint64, _P, _P,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), _P]),
    "zk_decode_frame_list_dev": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, _P, C.c_uint32, _P, C.c_uint64, C.c_int, _P, _P]),
    "zk_xxh64_frames": (C.c_int, [_P, _P, _P, C.c_uint32, _P]),
    "zk_host_alloc": (_P, [C.c_size_t]),
    "zk_host_free": (None, [_P]),
    "zk_engine_set_host_threads": (C.c_int, [_P, C.c_int]),
    "zk_xxh64_frames_dev": (C.c_int, [_P, _P, _P, C.c_uint32, _P, _P]),
    "zk_frame_content_sizes": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "zk_frame_content_sizes_dev": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, _P, _P, _P]),
    "zk_set_collective_library": (C.c_int, [C.c_char_p]),
    "zk_gather_seekable": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_int, _P, C.c_uint64, _P, _P, _P]),
}
for _name, (_res, _args) in _sig.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def declare(name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


def error_name(code: int) -> str:
    return lib.zk_error_name(code).decode()
