"""The Level-C shim's stream handling on the CPU (no GPU: the engine is stubbed): zeekstd_amd/csrc/levelc/zstd_shim.cpp follows frame and block
headers through the caller's buffers -- UNTRUSTED bytes -- to find where a frame ends.  tests/sim/shim_fuzz.cpp builds it with AddressSanitizer
+ UBSan and feeds it the goldens' archives whole, cut, bit-flipped and overwritten, in ragged input chunks with tiny output buffers, and random
inputs through ZSTD_compressStream2 the same way.  Checked on every call: positions stay in range and never go back, a call makes progress or
reports, the engine is shown exactly one frame at a time (magic number to last byte), every exact-size buffer is touched inside its bounds.
What the reference guards with cargo-fuzz targets; the real decode behind the shim is tests/test_gpu_levelc.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_stream_handling_under_sanitizers(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path / "shimfuzz")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                         os.path.join(ROOT, "tests", "sim", "shim_fuzz.cpp"), "-o", exe, "-pthread"], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr:
        cc = subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "sim", "shim_fuzz.cpp"), "-o", exe, "-pthread"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    for seed in (1, 2, 3):
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "archives.bin"), "20000", str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "no report" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
