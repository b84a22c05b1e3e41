#!/usr/bin/env python3
"""Scratch (round 6): bench.py's untimed per-rank host setup at world 8 -- eight processes at once, each generating its 2048 frames (4 GiB)
with cores / 8 - 1 forked workers, as `bench.py --gpus 8` does before any HIP call.  No GPU involved.   python tools/setup_probe.py [world] [frames]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ONE = """
import os, sys, time
sys.path.insert(0, %r)
import bench
rank, world, nframes = %d, %d, %d
cores = os.cpu_count() or 8
workers = max(1, min(64, cores // max(1, world) - 1))
t = time.time()
data, comp, frames, hashes = bench.build_inputs(rank * nframes, nframes, 1, True, workers, False, rank)
print(rank, workers, round(time.time() - t, 2), int(data.size))
"""

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, "-c", ONE % (ROOT, r, world, nframes)], stdout=subprocess.PIPE, text=True) for r in range(world)]
    res = [p.communicate()[0].split() for p in ps]
    ts = [float(r[2]) for r in res]
    print(f"world {world}, {nframes} frames per rank ({nframes * 2} MiB), {os.cpu_count()} host threads: per-rank setup {min(ts):.1f} .. {max(ts):.1f} s "
          f"({res[0][1]} workers per rank), wall {time.time() - t0:.1f} s")
