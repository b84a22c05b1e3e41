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
