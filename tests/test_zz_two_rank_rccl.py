"""Last file of the suite on purpose: the one test that needs TWO GPUs.  It has only ever been skipped (the boxes the suite runs on
have one device); on a node with more it is the first multi-rank RCCL execution of bench.py's N > 1 path, and whatever it finds
must not keep the rest of the suite from running under `pytest -x`."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_rccl_run_when_two_gpus_are_visible():
    """The launch contract of bench.py with TWO ranks over RCCL (one process per GPU, torch.distributed.run as the driver starts it),
    a small workload: decode of each rank's own 64 frames + the sharded encode with its gather on rank 0.  Runs wherever two devices
    are visible, so that the first multi-rank RCCL execution of this path is not the driver's 8-GPU bench; skips on a one-GPU box
    (where tests/test_parallel_gloo.py covers the same code over gloo and bench.py --dry-run the contract)."""
    import json
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--frames", "64", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-seek", "--no-e2e", "--no-c1"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    g = line["rccl_gather"]
    assert g and "error" not in g and g["frames_on_root"] == 128
