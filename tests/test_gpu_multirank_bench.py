"""bench.py's N > 1 path on the ONE GPU of the box: torch.distributed.run starts 2 and 8 ranks as the driver does, every rank uses device 0
(--one-gpu-transport, a test-only flag), the process group runs over gloo and the gather leg goes through the real zk_gather_seekable
(csrc/zk_engine_gather.hip) with its five collective entry points provided by tests/sim/libzk_shm_collectives.so.  What this executes on
hardware: the per-rank setup, the barriers and the max over ranks, the gather leg under its watchdog, the root's decode of the LAST rank's
frames out of the gathered archive, and rank 0's one JSON line with `rccl_gather` in it -- everything of configs[4]'s control flow but
RCCL's own transport (tests/test_zz_two_rank_rccl.py takes that wherever two devices are visible).  The numbers of these runs mean nothing."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 8])
def test_bench_n_ranks_on_one_gpu(world):
    lib = os.path.join(ROOT, "tests", "sim", "libzk_shm_collectives.so")
    if not os.path.exists(lib):
        pytest.skip("tests/sim/libzk_shm_collectives.so is not built (__graft_entry__.build)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--frames", "64", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-seek", "--no-e2e", "--no-c1", "--one-gpu-transport", lib], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["value"] > 0 and line["scaling"] == "weak" and line["steps"] == 2
    g = line["rccl_gather"]
    assert g and "error" not in g, g
    assert g["frames_on_root"] == 64 * world
    assert g["last_ranks_frames_decoded_from_the_gathered_archive"]["bit_exact"] is True
    assert g["last_ranks_frames_decoded_from_the_gathered_archive"]["frames"] == [64 * world - 4, 64 * world]
    e = g["gather_expectation"]
    assert e["peers"] == world - 1 and e["ms_if_the_receives_overlap"] * (world - 1) == pytest.approx(e["ms_if_they_are_serialised"], rel=0.05, abs=0.02)
    assert g["gather_alone_ms"] > 0 and "TEST" in g["transport"]
    assert line["setup_s"] < 60


def test_bench_gather_only_leg_on_one_gpu():
    lib = os.path.join(ROOT, "tests", "sim", "libzk_shm_collectives.so")
    if not os.path.exists(lib):
        pytest.skip("tests/sim/libzk_shm_collectives.so is not built (__graft_entry__.build)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--frames", "32", "--gather-only", "--one-gpu-transport", lib],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, HIP_VISIBLE_DEVICES="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 3 and line["rccl_gather"]["frames_on_root"] == 96 and "error" not in line["rccl_gather"]
