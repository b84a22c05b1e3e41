"""One rank of tests/test_gpu_gather_ranks.py (TEST HARNESS): encodes its contiguous range of frames on the GPU, takes part in the REAL
zk_gather_seekable (csrc/zk_engine_gather.hip) over the shared-memory transport of tests/sim/zk_shm_collectives.cpp, then -- the decode
side of SURVEY 8e -- reads ONLY its range of the gathered archive and decodes it with zk_decode_shard.
   python gather_rank.py <rank> <world> <shm name> <work dir> <frames per rank, comma separated> <frame size> <root> <root cap or 0>"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                            # device buffers only
import zeekstd_amd as zk
from zeekstd_amd import SeekTable
from oracle import zko                                  # input generation (SURVEY 8d generator)

rank, world, name, work = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
counts = [int(x) for x in sys.argv[5].split(",")]
fsz, root, root_cap = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
path = os.path.join(ROOT, "tests", "sim", "libzk_shm_collectives.so")
tr = C.CDLL(path)
tr.zkshm_comm_create.restype = C.c_void_p
tr.zkshm_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64]
tr.zkshm_barrier.argtypes = [C.c_void_p]
tr.zkshm_comm_destroy.argtypes = [C.c_void_p]
lib = zk.lib
lib.zk_set_collective_library.argtypes = [C.c_char_p]
assert lib.zk_set_collective_library(path.encode()) == 0
comm = tr.zkshm_comm_create(name.encode(), rank, world, 64 << 20)
assert comm
eng = zk.Engine(0)
dev = torch.device("cuda", 0)

# my frames: global frame k = chunk k of the generator (every frame distinct); the last frame of the whole input is ragged
first = sum(counts[:rank])
total_frames = sum(counts)
mine = b"".join(zko.gen_text(fsz, 0x5EED0002 + first + i) for i in range(counts[rank]))
if rank == world - 1 and counts[rank]:
    mine = mine[:len(mine) - fsz // 3]                  # a ragged last frame
if mine:
    comp, frames = eng.encode_frames(np.frombuffer(mine, np.uint8), fsz, 1, True)
else:
    comp, frames = b"", []                              # a rank without frames takes part in every collective all the same
d_pay = torch.frombuffer(bytearray(comp) or bytearray(1), dtype=torch.uint8).to(dev)
cs = np.array([f[0] for f in frames], np.uint32)
ds = np.array([f[1] for f in frames], np.uint32)
cap = root_cap if root_cap else (64 << 20)
d_out = torch.zeros(cap if rank == root else 1, dtype=torch.uint8, device=dev)
nbytes = C.c_uint64()
tab = C.c_void_p()
lib.zk_gather_seekable.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                   C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
rc = lib.zk_gather_seekable(eng._h, comm, rank, world, root, d_pay.data_ptr() if comp else None, len(comp),
                            cs.ctypes.data if len(frames) else None, ds.ctypes.data if len(frames) else None, len(frames), 1,
                            d_out.data_ptr() if rank == root else None, cap if rank == root else 0, C.byref(nbytes), C.byref(tab), None)
with open(os.path.join(work, f"rc_{rank}"), "w") as f:
    f.write(str(rc))
if rc == 0:
    if rank == root:
        stream = bytes(d_out[:nbytes.value].cpu().numpy())
        with open(os.path.join(work, "archive.zst"), "wb") as f:
            f.write(stream)
        lib.zk_seek_table_num_frames.argtypes = [C.c_void_p]
        assert lib.zk_seek_table_num_frames(tab) == total_frames
    assert tr.zkshm_barrier(comm) == 0                  # the archive is on disk
    # ---- decode side: the table every rank can read (8 n + 17 bytes at the end), then nothing but this rank's compressed bytes
    with open(os.path.join(work, "archive.zst"), "rb") as f:
        f.seek(0, 2); size = f.tell()
        tail = 8 * total_frames + 17
        f.seek(size - tail)
        table = SeekTable.from_seekable_format(f.read(tail), zk.Format.Foot)
        assert table.num_frames() == total_frames
        from zeekstd_amd import parallel
        lo, hi = parallel.shard_range(total_frames, rank, world)
        c0 = table.frame_start_comp(lo) if lo < total_frames else 0
        c1 = table.frame_end_comp(hi - 1) if hi > lo else c0
        f.seek(c0)
        shard = f.read(c1 - c0)
    out = parallel.decode_shard(eng, shard, table, rank, world, verify=True)[2] if hi > lo else b""
    with open(os.path.join(work, f"out_{rank}"), "wb") as f:
        f.write(out)
tr.zkshm_barrier(comm)
tr.zkshm_comm_destroy(comm)
eng.close()
