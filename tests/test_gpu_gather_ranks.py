"""SURVEY 8e / BASELINE configs[4] on the ONE GPU the box has (VERDICT r4 "next" 6): the real zk_gather_seekable
(csrc/zk_engine_gather.hip: all-gather of sizes, all-gather of padded entries, grouped sends / receives into slices of the root's buffer,
the agreed "does not fit" verdict) between 2, 3 and 8 PROCESSES that share device 0, its five collective entry points provided by a
shared-memory transport (tests/sim/zk_shm_collectives.cpp, named through zk_set_collective_library) instead of RCCL -- which needs a
device per rank and therefore never ran with a peer here.  Ragged shards, a rank without frames, a ragged last frame, a root that is not
rank 0; then the decode side: every rank reads the seek table and ONLY its frame range of the one gathered archive, decodes it with
zk_decode_shard, and the concatenation equals the input (lib/src/seek_table.rs:379-436 reads the table; seekable_format.md:45-157)."""
import os
import subprocess
import sys
import tempfile

import pytest

from oracle import zko
from zeekstd_amd import SeekTable

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "gather_rank.py")
FSZ = 256 << 10


def _run(counts, root=0, root_cap=0):
    world = len(counts)
    work = tempfile.mkdtemp(prefix="zk_gather_")
    name = f"zk_gather_{os.getpid()}_{world}_{root}_{root_cap}"
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), name, work, ",".join(map(str, counts)), str(FSZ), str(root), str(root_cap)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][-2000:])
    rcs = [int(open(os.path.join(work, f"rc_{r}")).read()) for r in range(world)]
    return work, rcs


def _input(counts):
    total = sum(counts)
    data = b"".join(zko.gen_text(FSZ, 0x5EED0002 + k) for k in range(total))
    return data[:len(data) - FSZ // 3] if counts[-1] else data


@pytest.mark.parametrize("counts,root", [([3, 5], 0), ([4, 0, 7], 2), ([2, 3, 1, 0, 4, 2, 5, 3], 0)], ids=["world2", "world3_empty_rank_root2", "world8"])
def test_gather_then_sharded_decode(counts, root):
    work, rcs = _run(counts, root)
    assert rcs == [0] * len(counts)
    want = _input(counts)
    archive = open(os.path.join(work, "archive.zst"), "rb").read()
    total = sum(counts)
    table = SeekTable.from_seekable(archive)
    assert table.num_frames() == total and table.size_decomp() == len(want)
    assert len(archive) == table.size_comp() + 8 * total + 17                    # the stream is the plain concatenation + the Foot table
    got = b"".join(open(os.path.join(work, f"out_{r}"), "rb").read() for r in range(len(counts)))
    assert got == want
    # the archive is an ordinary seekable archive: the oracle reads every frame
    c, d = table.offsets()
    for i in (0, total // 2, total - 1):
        out, used = zko.frame_decode(archive[int(c[i]):int(c[i + 1])], int(d[i + 1] - d[i]), True)
        assert out == want[int(d[i]):int(d[i + 1])]


def test_a_root_without_room_is_everybodys_verdict():
    """the root's capacity travels with the sizes: every rank returns dstSize_tooSmall (-70) before a send or a receive is posted -- nobody hangs"""
    work, rcs = _run([2, 2, 1], 0, root_cap=100000)
    assert rcs == [-70, -70, -70]
