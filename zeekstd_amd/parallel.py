"""Frames sharded over the GPUs of one node (BASELINE.json configs[4]).

Frames share no state, so the path shards with no data-path collective: rank r owns the contiguous frame
range [r*F/W, (r+1)*F/W).  The ONE exchange step is the concatenation of the compressed stream and of the
seek table on a root rank:
   1. all_gather of one int64 per rank (its compressed byte total)  -> every rank knows every offset
   2. payload gather to root: point-to-point isend / irecv straight into root_buf[offset_r : offset_r + c_r]
      (RCCL has no gatherv; on xGMI every peer has its own link into the root, so the W-1 receives run
      concurrently -- SURVEY 5 "distributed communication backend")
   3. all_gather of the fixed-size (c_size, d_size) seek entries (padded to the largest shard)
   4. root appends the serialised seek table (8n + 17 bytes)
torch.distributed is plumbing here (backend "nccl" == RCCL on ROCm, "gloo" for the CPU tests); the codec work
is the engine's.  Everything below is written against tensors so the same code runs on both backends.
"""
import numpy as np
import torch
import torch.distributed as dist

from .api import Format, SeekTable


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous frame range of `rank`: keeps the gathered stream a plain concatenation in rank order."""
    per, extra = divmod(n_frames, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class GatherTooSmall(RuntimeError):
    """The root's destination cannot hold the gathered stream + seek table; raised on EVERY rank (nobody has posted a send)."""


def gather_seekable(payload: torch.Tensor, c_sizes: torch.Tensor, d_sizes: torch.Tensor, root: int = 0, group=None,
                    fmt: Format = Format.Foot, out_cap: int = None):
    """payload: uint8 tensor with this rank's concatenated frames; c_sizes / d_sizes: int32 tensors (one per frame).
    Returns on root (stream tensor incl. the seek table, SeekTable); on the other ranks (None, None).
    out_cap (root only, optional): room the root has for the result.  It travels with the sizes in step 1, so that every
    rank reaches the same verdict before any send / receive is posted (zk_gather_seekable at the C ABI does the same): a
    root that bailed out alone would leave its peers inside their sends until the communicator times out."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = payload.device
    # 1. byte totals + frame counts (+ the root's capacity, -1 = unlimited)
    cap = -1 if out_cap is None or rank != root else int(out_cap)
    mine = torch.tensor([payload.numel(), c_sizes.numel(), cap], dtype=torch.int64, device=dev)
    allv = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    totals = [int(v[0]) for v in allv]
    counts = [int(v[1]) for v in allv]
    offs = np.concatenate([[0], np.cumsum(totals)])
    root_cap = int(allv[root][2])
    need = int(offs[-1]) + 8 * sum(counts) + 17                     # seekable_format.md: the table is 8 n + 17 bytes
    if root_cap >= 0 and need > root_cap:
        raise GatherTooSmall(f"gathered stream + seek table need {need} bytes, the root has {root_cap}")
    # 3. seek entries, padded to the largest shard
    mx = max(max(counts), 1)
    ent = torch.zeros(2 * mx, dtype=torch.int32, device=dev)
    ent[:c_sizes.numel()] = c_sizes.to(torch.int32)
    ent[mx:mx + d_sizes.numel()] = d_sizes.to(torch.int32)
    ents = [torch.zeros(2 * mx, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(ents, ent, group=group)
    # 2. payload gather: every peer sends straight into its slot of the root buffer
    table = None
    out = None
    if rank == root:
        table = SeekTable.new()
        for r in range(world):
            e = (ents[r].cpu().numpy().astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)
            table.log_frames(e[:counts[r]], e[mx:mx + counts[r]])          # a shard's entries in one call (16 384 frames per rank in configs[4])
        tbytes = table.to_bytes(fmt)
        out = torch.empty(int(offs[-1]) + len(tbytes), dtype=torch.uint8, device=dev)
        reqs = []
        for r in range(world):
            dst = out[int(offs[r]):int(offs[r + 1])]
            if r == rank:
                dst.copy_(payload)
            elif totals[r]:
                reqs.append(dist.P2POp(dist.irecv, dst, r, group))
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
        out[int(offs[-1]):] = torch.frombuffer(bytearray(tbytes), dtype=torch.uint8).to(dev)
    elif payload.numel():
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, payload.contiguous(), root, group)]):
            w.wait()
    return out, table


def encode_sharded(engine, d_src: torch.Tensor, frame_size: int, level: int = 1, checksum: bool = False, root: int = 0,
                   group=None, fmt: Format = Format.Foot, out_cap: int = None):
    """Each rank passes ITS shard of the input (whole frames except for the global tail); returns gather_seekable()."""
    from . import lib
    n = d_src.numel()
    cap = int(lib.zk_compress_bound(n, frame_size))
    nf = max(1, -(-n // frame_size))
    dev = d_src.device
    d_comp = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_cs = torch.empty(nf, dtype=torch.int32, device=dev)                    # (the engine writes every entry: no fill kernel of torch's to race with)
    d_ds = torch.empty(nf, dtype=torch.int32, device=dev)
    # The engine works on ITS queue (a non-blocking stream): whatever torch still has in flight on the caller's stream for these buffers --
    # the kernel that produced d_src, a fill of a fresh tensor -- must have finished before the call.  (Found on one GPU shared by eight
    # ranks: a delayed torch.zeros() landed on top of the seek entries the encoder had just written.)
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    nfo, written = engine.encode_frames_dev(d_src, n, frame_size, level, checksum, d_comp, cap, d_cs, d_ds)
    return gather_seekable(d_comp[:written], d_cs[:nfo], d_ds[:nfo], root, group, fmt, out_cap)


def decode_sharded(engine, comp: bytes, table: SeekTable, rank: int, world: int, verify: bool = True):
    """Rank r decodes its contiguous frame range of a seekable archive; the output stays sharded.
    Returns (first_frame, last_frame_exclusive, bytes)."""
    lo, hi = shard_range(table.num_frames(), rank, world)
    c, d = table.offsets()
    out, st = engine.decode_frames(comp, c, d, first=lo, count=hi - lo, verify=verify)
    return lo, hi, out


def decode_shard(engine, comp_shard: bytes, table: SeekTable, rank: int, world: int, verify: bool = True):
    """The same through the C ABI's zk_decode_shard (what a Rust / C++ host calls): comp_shard holds ONLY this rank's
    compressed bytes [c_off[first], c_off[first + count]).  Returns (first_frame, count, bytes)."""
    import ctypes as C
    from . import lib
    from .api import _chk
    lo, hi = shard_range(table.num_frames(), rank, world)
    _, d = table.offsets()
    n = int(d[hi] - d[lo])
    out = np.zeros(max(n, 1), np.uint8)
    src = np.frombuffer(bytes(comp_shard) + b"\0" * 8, np.uint8)
    first, count, written = C.c_uint32(), C.c_uint32(), C.c_uint64()
    _chk(lib.zk_decode_shard(engine._h, src.ctypes.data, len(comp_shard), table._h, rank, world, out.ctypes.data, n, int(verify),
                             C.byref(first), C.byref(count), C.byref(written)))
    assert (first.value, first.value + count.value) == (lo, hi) and written.value == n
    return first.value, count.value, out[:n].tobytes()
