"""The C-ABI shared library loads (no GPU needed for that) and exports every symbol that
include/zeekstd_amd.h declares.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
            continue
        txt = open(os.path.join(ROOT, "include", fn)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"\b(zk_[a-z0-9_]+)\s*\(", txt):
            names.append(m.group(1))
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    import zeekstd_amd as zk
    names = declared_functions()
    assert len(names) >= 8
    missing = [n for n in names if not hasattr(zk.lib, n)]
    assert not missing, missing


def test_abi_version_and_error_names():
    import zeekstd_amd as zk
    assert zk.lib.zk_abi_version() == 2
    assert zk.error_name(-20) == "Data corruption detected"      # ZSTD_getErrorName strings (error.rs:68)
    assert zk.error_name(-22) == "Restored data doesn't match checksum"
    assert zk.error_name(-1001) == "offset out of range"          # error.rs:60-71
    assert zk.error_name(-1002) == "frame index too large"


def test_no_cpu_fallback_without_gpu():
    """Product path must fail loudly when no gfx950 device is usable."""
    import zeekstd_amd as zk
    if os.path.exists("/dev/kfd"):
        return
    h = C.c_void_p()
    assert zk.lib.zk_engine_create(0, C.byref(h)) == -2002
    assert not h.value


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may import / link / execute anything under oracle/."""
    pkg = os.path.join(ROOT, "zeekstd_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#\s*include\s*[\"<][^\">]*oracle)|(libzko)|(dlopen\([^)]*oracle)", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert not bad.search(txt), (dp, f)
    so = os.path.join(pkg, "libzeekstd_amd.so")
    import subprocess
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "zko" not in needed and "libzstd" not in needed        # the product links neither the oracle nor libzstd


def test_rust_binding_declares_every_symbol():
    """rust/src/ffi.rs (what a zeekstd maintainer binds; uncompiled here -- no cargo in the image) is generated from the header
    by tools/gen_rust_ffi.py: it is current, and declares every exported zk_* function with the header's number of arguments."""
    import subprocess
    import sys
    import zeekstd_amd as zk
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi as g
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]).returncode == 0, \
        "rust/src/ffi.rs is stale: run tools/gen_rust_ffi.py"
    fns = g.parse_functions(open(os.path.join(ROOT, "include", "zeekstd_amd.h")).read())
    rs = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    exported = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", zk.LIB_PATH], text=True).splitlines()
                if l.split()[-1].startswith("zk_")}
    assert {f[0] for f in fns} == exported                      # header == library
    for name, _ret, params in fns:
        m = re.search(r"pub fn %s\(([^)]*)\)" % name, rs)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(params), name
    # the safe wrappers reach every Level-B entry point
    lib_rs = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    for name in exported:
        if any(name.startswith(p) for p in ("zk_decoder_", "zk_encoder_", "zk_raw_encoder_", "zk_seek_table_", "zk_serializer_")) \
                and name not in ("zk_decoder_open_bytes", "zk_decoder_open_file", "zk_decoder_open_callbacks", "zk_decoder_gpu_submissions", "zk_decoder_time_seeks",
                                 "zk_seek_table_from_reader_bytes", "zk_seek_table_entries", "zk_serializer_reset", "zk_raw_encoder_compress"):
            assert "ffi::" + name in lib_rs, name


def test_python_binding_types_agree_with_the_header():
    """names and arity alone say little.  Every ctypes signature the Python mirror declares (zeekstd_amd/_lib.py,
    api.py -- written by hand) is compared with the C declaration in include/zeekstd_amd.h argument by argument: pointer against
    pointer, integer against integer of the same width and signedness, double against double, and the return type.  (rust/src/ffi.rs
    is generated from the same parse of the header, so a type that is right here is right there.)"""
    import sys
    import zeekstd_amd as zk
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi as g
    fns = g.parse_functions(open(os.path.join(ROOT, "include", "zeekstd_amd.h")).read())
    ints = {"int": ("i", 4), "int32_t": ("i", 4), "uint32_t": ("u", 4), "int64_t": ("i", 8), "uint64_t": ("u", 8), "size_t": ("u", 8), "uint8_t": ("u", 1)}

    def c_kind(t):
        t = t.replace("const", "").strip()
        if "*" in t or t.endswith("_fn"):                       # object pointers, handles, callback typedefs
            return ("p", 8)
        if t in ints:
            return ints[t]
        if t in ("double", "float"):
            return ("f", 8 if t == "double" else 4)
        if t == "void":
            return None
        raise AssertionError("type the test does not know: " + t)

    def py_kind(t):
        if t is None:
            return None
        if issubclass(t, (C._Pointer, C.c_void_p, C.c_char_p)) or hasattr(t, "_flags_") and not hasattr(t, "_type_") or issubclass(t, C._CFuncPtr):
            return ("p", 8)
        code = getattr(t, "_type_", None)
        if code in ("P", "z", "Z"):
            return ("p", 8)
        if code in ("d", "f"):
            return ("f", C.sizeof(t))
        if isinstance(code, str) and code in "bBhHiIlLqQ?":
            return ("i" if code in "bhilq" else "u", C.sizeof(t))
        raise AssertionError("ctypes type the test does not know: %r" % (t,))

    undeclared, checked = [], 0
    for name, ret, params in fns:
        f = getattr(zk.lib, name)
        if f.argtypes is None:
            undeclared.append(name)
            continue
        assert len(f.argtypes) == len(params), name
        for i, ((ctype, pname), at) in enumerate(zip(params, f.argtypes)):
            assert py_kind(at) == c_kind(ctype), (name, i, pname, ctype, at)
        want = c_kind(ret)
        got = py_kind(f.restype)
        # (a C `int` status may be read as c_int only; pointers returned as c_char_p / c_void_p)
        assert got == want, (name, "return", ret, f.restype)
        checked += 1
    assert checked >= len(fns) - 6, undeclared               # a handful of entry points are bound where they are used (bench.py, tests)


def test_header_is_plain_c_and_a_c_client_links():
    """include/zeekstd_amd.h is the boundary a cgo / JNI / Rust-bindgen host compiles: it must be plain C99 (and C++11) on its own,
    and a C translation unit that names every declared function must link against the library (no GPU needed for either)."""
    import shutil
    import subprocess
    import sys
    import tempfile
    import zeekstd_amd as zk
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "zeekstd_amd.h")
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", lang, hdr], capture_output=True, text=True)
        assert r.returncode == 0, (cc, r.stderr[-2000:])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi as g
    fns = g.parse_functions(open(hdr).read())
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "client.c")
        with open(src, "w") as f:
            f.write('#include "zeekstd_amd.h"\n#include <stdio.h>\nint main(void)\n{\n    const void *fn[] = {\n')
            f.write("".join("        (const void *)(size_t)&%s,\n" % name for name, _, _ in fns))
            f.write('    };\n    printf("%d %zu\\n", zk_abi_version(), sizeof fn / sizeof fn[0]);\n    return 0;\n}\n')
        exe = os.path.join(td, "client")
        libdir = os.path.dirname(zk.LIB_PATH)
        r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir, "-l:libzeekstd_amd.so",
                            "-Wl,-rpath," + libdir], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.split() == [str(zk.lib.zk_abi_version()), str(len(fns))], (out.stdout, out.stderr[-500:])


# SURVEY Appendix D ("Public API surface to preserve"): every public item of the reference's crate, by type.  (`with_cctx` / `cctx` /
# `with_dctx` / `dctx` hand a libzstd context in or out; here that is the engine: `engine(&Engine)`.)
APPENDIX_D = {
    "consts": {"SEEKABLE_MAGIC_NUMBER", "SEEKABLE_MAX_FRAMES", "SEEK_TABLE_INTEGRITY_SIZE", "SEEKABLE_MAX_FRAME_SIZE"},
    "types": {"DecodeOptions", "Decoder", "Encoder", "CompressionProgress", "EncodeOptions", "EpilogueProgress", "FrameSizePolicy", "RawEncoder",
              "Error", "Result", "SeekTable", "BytesWrapper", "OffsetFrom", "Seekable", "CompressionLevel", "Format", "Serializer"},
    "CompressionProgress": {"in_progress", "out_progress"},
    "EpilogueProgress": {"out_progress", "data_left"},
    "EncodeOptions": {"new", "try_new", "frame_size_policy", "checksum_flag", "compression_level", "into_raw_encoder", "into_encoder", "engine"},
    "RawEncoder": {"with_opts", "new", "compress_with_prefix", "compress", "end_frame", "seek_table", "into_seek_table", "reset_frame", "reset_seek_table"},
    "Encoder": {"new", "with_opts", "seek_table", "written_compressed", "into_seek_table", "compress_with_prefix", "compress", "end_frame", "finish", "finish_format"},
    "DecodeOptions": {"new", "try_new", "seek_table", "lower_frame", "upper_frame", "offset", "offset_limit", "into_decoder", "engine"},
    "Decoder": {"new", "with_opts", "decompress_with_prefix", "decompress", "reset", "set_lower_frame", "set_upper_frame", "set_offset", "set_offset_limit",
                "read_compressed", "seek_table", "offset", "offset_limit"},
    "SeekTable": {"new", "from_seekable", "from_seekable_format", "from_reader", "log_frame", "num_frames", "frame_index_comp", "frame_index_decomp",
                  "frame_start_comp", "frame_start_decomp", "frame_end_comp", "frame_end_decomp", "frame_size_comp", "frame_size_decomp",
                  "max_frame_size_comp", "max_frame_size_decomp", "size_comp", "size_decomp", "into_serializer", "into_format_serializer"},
    "Serializer": {"write_into", "reset", "encoded_len"},
    "BytesWrapper": {"new"},
    "Error": {"is_number_conversion_failed", "is_offset_out_of_range", "is_frame_index_too_large", "is_io", "is_zstd"},
    "Seekable": {"set_offset", "read", "seek_table_integrity"},
    "traits": {("Write", "Encoder"), ("Read", "Decoder"), ("Seek", "Decoder"), ("Read", "Serializer"), ("Seekable", "BytesWrapper"), ("Display", "Error"),
               ("Default", "SeekTable"), ("Clone", "SeekTable"), ("PartialEq", "SeekTable"), ("Eq", "SeekTable"), ("Debug", "SeekTable"),
               ("Default", "FrameSizePolicy"), ("Default", "Format")},
}


def test_rust_face_has_every_public_item_of_the_reference_crate():
    """VERDICT r4 'next' 7a: the Rust face (rust/src/lib.rs, uncompiled here: no rustc in the image) against SURVEY Appendix D, item by
    item -- not only the FFI names.  A brace-matching reader of the file: `impl` blocks -> the type's public functions, trait impls,
    derives, constants, type names."""
    import re
    src = open(os.path.join(ROOT, "rust", "src", "lib.rs")).read()
    src_nc = re.sub(r"//[^\n]*", "", src)
    consts = set(re.findall(r"pub const (\w+)", src_nc))
    assert APPENDIX_D["consts"] <= consts, APPENDIX_D["consts"] - consts
    types = set(re.findall(r"pub (?:struct|enum|trait|type) (\w+)", src_nc))
    assert APPENDIX_D["types"] <= types, APPENDIX_D["types"] - types
    # impl blocks
    fns, traits = {}, set()
    for m in re.finditer(r"\bimpl\b(?:<[^{]*?>)?\s+(?:(?P<tr>[\w:]+(?:<[^{>]*>)?)\s+for\s+)?(?P<ty>[\w:]+)[^{]*\{", src_nc):
        depth, i = 1, m.end()
        while depth and i < len(src_nc):
            depth += {"{": 1, "}": -1}.get(src_nc[i], 0)
            i += 1
        body = src_nc[m.end():i]
        ty = m.group("ty").split("::")[-1]
        if m.group("tr"):
            traits.add((re.sub(r"<.*", "", m.group("tr")).split("::")[-1], ty))
        else:
            fns.setdefault(ty, set()).update(re.findall(r"pub fn (\w+)", body))
    for m in re.finditer(r"#\[derive\(([^)]*)\)\]\s*pub (?:struct|enum) (\w+)", src_nc):
        for tr in m.group(1).split(","):
            traits.add((tr.strip(), m.group(2)))
    tm = re.search(r"pub trait Seekable \{(.*?)\n\}", src_nc, re.S)
    fns["Seekable"] = set(re.findall(r"fn (\w+)", tm.group(1)))
    for ty, want in APPENDIX_D.items():
        if ty in ("consts", "types", "traits"):
            continue
        assert want <= fns.get(ty, set()), (ty, want - fns.get(ty, set()))
    assert APPENDIX_D["traits"] <= traits, APPENDIX_D["traits"] - traits
    # the blanket impl that makes every Read + Seek a source (seekable.rs:112-138), OffsetFrom -> SeekFrom (:100-109), the integer conversion error
    assert re.search(r"impl<T: Read \+ Seek> Seekable for T", src_nc) and ("From", "SeekFrom") in {(t.split("<")[0], y) for t, y in traits} | traits
    assert "From<core::num::TryFromIntError> for Error" in src_nc and "From<io::Error> for Error" in src_nc
    # every FFI name the face calls exists in the generated bindings (and so, by the tests above, in the header and the library)
    ffi = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    for name in set(re.findall(r"ffi::(zk_\w+)", src_nc)):
        assert f"pub fn {name}(" in ffi, name


def test_level_c_shim_exports_the_symbols_the_crate_binds():
    """the exact subset of libzstd the unmodified crate (through zstd-safe) calls; zeekstd_amd/libzstd_zeekstd_amd.so must export
    every one (tests/test_gpu_levelc.py drives them on the GPU)."""
    so = os.path.join(ROOT, "zeekstd_amd", "libzstd_zeekstd_amd.so")
    assert os.path.exists(so), "make -C zeekstd_amd/csrc"
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    have = {l.split()[-1] for l in out.splitlines() if " T " in l}
    want = {"ZSTD_createCCtx", "ZSTD_freeCCtx", "ZSTD_CCtx_setParameter", "ZSTD_CCtx_refPrefix", "ZSTD_compressStream2", "ZSTD_CCtx_reset",
            "ZSTD_CStreamOutSize", "ZSTD_CStreamInSize", "ZSTD_createDCtx", "ZSTD_freeDCtx", "ZSTD_decompressStream", "ZSTD_DCtx_refPrefix",
            "ZSTD_DCtx_reset", "ZSTD_DCtx_setParameter", "ZSTD_DStreamInSize", "ZSTD_DStreamOutSize", "ZSTD_isError", "ZSTD_getErrorCode",
            "ZSTD_getErrorName", "ZSTD_versionNumber", "ZSTD_versionString"}
    assert want <= have, want - have
    assert not {s for s in have if s.startswith("zk_")}, "the shim re-exports nothing of Level A"


def test_a_collective_library_that_cannot_be_loaded_is_refused_at_once():
    """zk_set_collective_library resolves the five entry points when it is called: a path that does not load, or a library without
    them, is an error HERE -- and leaves the provider in force untouched -- not a generic failure of some later gather."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "zeekstd_amd", "libzeekstd_amd.so"))
    lib.zk_set_collective_library.argtypes = [C.c_char_p]
    lib.zk_set_collective_library.restype = C.c_int
    assert lib.zk_set_collective_library(b"/nonexistent/libnothing.so") != 0
    assert lib.zk_set_collective_library(b"libm.so.6") != 0            # loads, but has no ncclAllGather
    assert lib.zk_set_collective_library(None) == 0                    # back to RCCL (resolved, present or not: a gather says so)
    assert lib.zk_set_collective_library(b"") == 0
