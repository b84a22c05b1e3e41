"""The encoder's match + parse kernel (zeekstd_amd/csrc/zk_enc_match.h -- the source hipcc compiles for gfx950) executed
on the CPU by tests/sim/zk_enc_sim.cpp under a workgroup emulator (a fiber per lane, barriers and wave collectives as
meeting points), compared block by block with the CPU twin oracle/zstd_oracle_enc.c.  What the reference does here is
ZSTD_compressStream2 (lib/src/encode.rs:340-346); compressed bytes are unpinned by it, so the twin is the yardstick and
the -m gpu tests then hold the real kernel to the same bytes."""
import numpy as np
import pytest

from conftest import enc_sim_match
from oracle import zko


def _compare(data, frame_size, level, prefix=None):
    got = enc_sim_match(data, frame_size, level, prefix)
    want = []
    for o in range(0, max(len(data), 1), frame_size):
        want += zko.enc_match_debug(data[o:o + frame_size], level, prefix)
    assert len(got) == len(want)
    for b, ((gs, gl, bsz), (ws, wl)) in enumerate(zip(got, want)):
        assert len(gs) == len(ws), (b, len(gs), len(ws))
        bad = np.nonzero(gs != ws)[0]
        assert bad.size == 0, (b, int(bad[0]), hex(int(gs[bad[0]])), hex(int(ws[bad[0]])))
        assert gl == wl, b
        # the sequences + literals reproduce the block size
        assert sum(int(x) & 0xFFFFF for x in gs) + sum((int(x) >> 20) & 0xFFFFF for x in gs) + (len(gl) - sum(int(x) & 0xFFFFF for x in gs)) == bsz


CASES = {
    "text": [["text", 70000, 11]],
    "mixed": [["text", 20000, 1], ["zeros", 9000], ["random", 7000, 3], ["rep", "616263", 3000], ["text", 30000, 2]],
    "runs": [["rep", "61", 700], ["rep", "62", 13], ["rep", "6364", 900], ["random", 300, 5], ["rep", "00", 5000]],
    "records": [["records", 4000, 7, "00010203040506070809"]],
    "tiny": [["text", 23, 5]],
    "seven": [["text", 7, 5]],
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("level", [1, 3, 6])
def test_sim_match_equals_twin(name, level):
    data = zko.make_input(CASES[name])
    _compare(data, 1 << 21, level)


def test_sim_match_small_frames_and_blocks():
    data = zko.make_input([["text", 50000, 21]])
    _compare(data, 10000, 1)          # 1 KiB .. 4 KiB blocks, partial groups
    _compare(data, 3000, 3)


def test_sim_match_segments():
    """a frame above ZKE_SEGMENT (256 KiB): the second segment starts from the 57280 bytes before it"""
    data = zko.make_input([["text", 300000, 31], ["rep", "6162636465666768", 2000]])
    _compare(data, 1 << 21, 1)


def test_sim_match_prefix():
    prefix = zko.make_input([["text", 70001, 41]])
    data = prefix[1000:30000] + zko.make_input([["text", 9000, 42]])
    _compare(data, 1 << 21, 1, prefix)
    _compare(data[:5000], 1 << 21, 3, prefix[:1003])
