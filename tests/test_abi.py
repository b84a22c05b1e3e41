"""The C-ABI shared library loads (no GPU needed for that) and exports every symbol that
include/zeekstd_amd.h declares.  No compute calls here."""
import ctypes as C
import os
import re

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
