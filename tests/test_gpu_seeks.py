"""BASELINE.json configs[3] as a parity test: 64 KiB frames, random set_offset / set_offset_limit / read-to-exhaustion seeks
(the xorshift64* protocol of SURVEY 8d, bench.seek_protocol) through the zeekstd Decoder API, every read compared with the
generator bytes -- on the archive this engine's encoder writes and on the archive the reference's CPU Encoder writes.
Mirrors lib/src/decode.rs:822-851 (byte ranges) and fuzz/fuzz_targets/roundtrip_seek.rs:7-43."""
import ctypes as C
import subprocess
import sys
import os

import numpy as np
import pytest

import zeekstd_amd as zk
from zeekstd_amd import DecodeOptions, SeekTable
import bench
from oracle import zko
from oracle import libzstd_ref as Z

pytestmark = pytest.mark.gpu
FSZ = 65536
NBYTES = 64 << 20                       # 1024 frames of 64 KiB


@pytest.fixture(scope="module")
def text():
    return zko.gen_chunks(NBYTES, 40)


def _archive(engine, text, who):
    if who == "gpu":
        return engine.encode_frames(np.frombuffer(text, np.uint8), FSZ, 1, True)
    if Z.load("system") is None:
        pytest.skip("needs a libzstd to build the reference-made archive")
    return Z.encode_seekable_frames(text, FSZ, 1, True, "system")


def _seekable(comp, frames):
    st = SeekTable.new()
    for c, d in frames:
        st.log_frame(c, d)
    return comp + st.to_bytes()


@pytest.mark.parametrize("who", ["gpu", "libzstd"])
def test_random_seeks_64k_frames(engine, text, who):
    comp, frames = _archive(engine, text, who)
    assert len(frames) == NBYTES // FSZ
    d = DecodeOptions(_seekable(comp, frames)).engine(engine).into_decoder()
    offs, lens = bench.seek_protocol(400, NBYTES)
    buf = bytearray(8192)
    for o, l in zip(offs.tolist(), lens.tolist()):
        d.set_offset_limit(NBYTES)
        d.set_offset(o)
        d.set_offset_limit(o + l)
        got = bytearray()
        while True:
            k = d.decompress(buf)
            if k == 0:
                break
            got += buf[:k]
        assert bytes(got) == text[o:o + l], (who, o, l)
        assert d.offset() == o + l


@pytest.mark.parametrize("who", ["gpu", "libzstd"])
def test_seek_timing_helper_verifies(engine, text, who):
    """zk_decoder_time_seeks (the bench's one-at-a-time leg) compares every read with the expected bytes."""
    comp, frames = _archive(engine, text, who)
    src = np.frombuffer(text, np.uint8)
    offs, lens = bench.seek_protocol(300, NBYTES)
    r = bench.time_single_seeks(engine, zk, comp, frames, src, offs, lens)
    assert r["gpu_decoder_us"]["p50"] > 0
    bad = src.copy(); bad[int(offs[7])] ^= 1            # a wrong expectation must be noticed
    with pytest.raises(RuntimeError):
        bench.time_single_seeks(engine, zk, comp, frames, bad, offs, lens)


def test_batched_seeks(engine, text):
    comp, frames = _archive(engine, text, "gpu")
    src = np.frombuffer(text, np.uint8)
    offs, _ = bench.seek_protocol(2048, NBYTES)
    r = bench.batched_seeks(engine, comp, frames, src, offs, B=1024, reps=2)
    assert r["batches"] == 2 and r["us_per_seek"] > 0


def test_forward_seek_inside_the_cached_frame_costs_no_submission(engine, text):     # decode.rs:912-939
    comp, frames = _archive(engine, text, "gpu")
    d = DecodeOptions(_seekable(comp, frames)).engine(engine).into_decoder()
    d.set_offset(5 * FSZ + 100); d.set_offset_limit(5 * FSZ + 200)
    buf = bytearray(4096)
    assert d.decompress(buf) == 100
    n = d.gpu_submissions()
    d.set_offset_limit(NBYTES); d.set_offset(5 * FSZ + 3000); d.set_offset_limit(5 * FSZ + 3500)
    assert d.decompress(buf) == 500 and bytes(buf[:500]) == text[5 * FSZ + 3000:5 * FSZ + 3500]
    assert d.gpu_submissions() == n and d.read_compressed() > 0


def test_seek_to_a_neighbouring_frame_in_the_read_ahead_costs_no_submission(engine, text):
    """set_offset to another frame (or backwards) resets the decode state (decode.rs:408-410) -- but frames the
    engine has already decoded ahead stay valid, so a seek into them is served without a new submission.  read_compressed
    still restarts from 0 at the seek and counts the frames delivered, as upstream's re-read would."""
    comp, frames = _archive(engine, text, "gpu")
    d = DecodeOptions(_seekable(comp, frames)).engine(engine).batch_bytes(16 * FSZ).into_decoder()
    d.set_offset(4 * FSZ)
    buf = bytearray(FSZ)
    assert d.decompress(buf) == FSZ                                  # frame 4 alone (too small for the decode-ahead)
    assert d.decompress(bytearray(10)) == 10                         # a read that continues where the cache ends: frames 5 .. 20 are decoded ahead
    n = d.gpu_submissions()
    for off in (9 * FSZ + 5, 6 * FSZ + 77, 12 * FSZ):                 # forwards to another frame, backwards, forwards again
        d.set_offset(off); d.set_offset_limit(off + 300)
        assert d.read_compressed() == 0                              # the state was reset (decode.rs:408-410)
        small = bytearray(1000)
        assert d.decompress(small) == 300 and bytes(small[:300]) == text[off:off + 300]
        assert d.read_compressed() == frames[off // FSZ][0]          # one frame's worth, as a re-read would count
        d.set_offset_limit(NBYTES)
    assert d.gpu_submissions() == n
    d.reset()                                                        # Decoder::reset drops the read-ahead like upstream's state
    d.set_offset(9 * FSZ); d.set_offset_limit(9 * FSZ + 10)
    assert d.decompress(bytearray(64)) == 10 and d.gpu_submissions() == n + 1


def test_cpu_suites_also_pass_on_the_gpu_box():
    """The seek-table golden bytes (tests/test_seek_table.py) and the ABI checks (tests/test_abi.py: every symbol of
    include/zeekstd_amd.h is exported, the product never touches oracle/) are CPU tests; the driver's GPU pass selects
    `-m gpu` only, so they are run from here as well."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "tests/test_seek_table.py", "tests/test_abi.py"],
                       cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_a_frame_with_more_sequences_than_its_output_could_hold(engine):
    """the small path sizes its record scratch from bounds the host knows (sequences <= d / 3); a crafted frame
    declares 30 000 sequences in a dozen bytes (RLE_Mode for all three tables, 0-bit codes: RFC 8878 3.1.1.3.2.1) and claims
    100 bytes of output.  zk_k_small_walk now counts the record slots against what was reserved and hands the frame to the general
    path, which sizes the records from the real totals and gives libzstd's verdict: the frame is damaged.  (Before: the small
    entropy kernel wrote 30 000 records into room for ~8 000.)  lib/src/decode.rs:242-256 is the loop this sits under."""
    n = 30000
    block = bytes([4 << 3]) + b"abcd" + bytes([(n >> 8) + 128, n & 0xFF, 0x54, 0, 0, 0]) + b"\x01"
    frame = bytes.fromhex("28B52FFD") + bytes([0x20, 100]) + ((len(block) << 3) | (2 << 1) | 1).to_bytes(3, "little") + block
    good = engine.encode_frames(np.frombuffer(zko.gen_chunks(3 * FSZ, 3), np.uint8), FSZ, 1, False)
    comp = good[0][:good[1][0][0]] + frame + good[0][good[1][0][0]:]
    frames = [good[1][0], (len(frame), 100)] + list(good[1][1:])
    d = DecodeOptions(_seekable(comp, frames)).engine(engine).into_decoder()
    buf = bytearray(FSZ)
    d.set_offset(FSZ + 10)                                      # a seek into the crafted frame: the small path
    d.set_offset_limit(FSZ + 60)
    with pytest.raises(zk.Error):
        while d.decompress(buf):
            pass
    # the frames around it still decode, through the same decoder
    want = zko.gen_chunks(3 * FSZ, 3)
    d2 = DecodeOptions(_seekable(comp, frames)).engine(engine).into_decoder()
    d2.set_offset(FSZ + 100 + 5)
    d2.set_offset_limit(FSZ + 100 + 4000)
    got = bytearray()
    while True:
        k = d2.decompress(buf)
        if not k:
            break
        got += buf[:k]
    assert bytes(got) == want[FSZ + 5:FSZ + 4000]
    # and the engine's own verdict on the frame alone
    _, st = engine.decode_frames(frame + b"\0" * 8, [0, len(frame)], [0, 100], raise_on_error=False)
    assert st[0] != 0
