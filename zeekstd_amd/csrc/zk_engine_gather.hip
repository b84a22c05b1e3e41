// zk_engine_gather.hip -- the one exchange step of the sharded path (BASELINE configs[4]) at the C ABI, for hosts
// without torch: every rank has encoded its contiguous range of frames (zk_encode_frames_dev); the compressed streams are
// concatenated in rank order on `root` and the seek table (8n + 17 bytes) is appended.  RCCL over xGMI:
//   1. ncclAllGather of (bytes, frames) per rank            -> every rank knows every offset
//   2. ncclAllGather of the (c_size, d_size) entries, padded to the largest shard
//   3. grouped ncclSend (peers) / ncclRecv (root) of the payload straight into root's buffer at offset_r: every peer has its
//      own xGMI link into the root, so the receives run side by side; no padding, no staging
//   4. root serialises the table behind the last frame
// RCCL is resolved at run time (the library must not depend on it for single-GPU use): symbols already in the process first,
// then librccl.so.1 -- or the library named with zk_set_collective_library (tests: a shared-memory transport between processes that
// share one GPU, tests/sim/zk_shm_collectives.cpp).  The communicator is the caller's (ncclCommInitRank), one per rank as always.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/zeekstd_amd.h"
#include "zk_engine.h"
#include "host/zeekstd.hpp"

#define ZK_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            e->last_err = std::string(#call) + ": " + hipGetErrorString(_e);                         \
            return ZK_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)

namespace {
typedef int (*allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*sendrecv_fn)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*group_fn)(void);
struct Rccl { allgather_fn all_gather = nullptr; sendrecv_fn send = nullptr, recv = nullptr; group_fn group_start = nullptr, group_end = nullptr; bool ok = false; };
// Who provides the five entry points: resolved ONCE per naming, under a lock, into a table that is published whole (a gather on another
// thread sees the old table or the new one, never half of one) -- and eagerly, so that a library that cannot be loaded is an error of
// zk_set_collective_library, not a generic one of some later gather.
std::mutex g_mu;
std::string g_collective_path;            // zk_set_collective_library: which library provides the five entry points ("" = RCCL)
bool g_resolved = false;
Rccl g_table;
Rccl resolve(const std::string &path)
{
    Rccl r;
    void *h = nullptr;
    const bool named = !path.empty();
    // a named library is asked first and alone; otherwise symbols already in the process (a host that linked RCCL), then librccl.so.1
    auto sym = [&](const char *n) { void *p = named ? nullptr : dlsym(RTLD_DEFAULT, n); if (!p && h) p = dlsym(h, n); return p; };
    if (named) h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    else if (!dlsym(RTLD_DEFAULT, "ncclAllGather")) {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    }
    r.all_gather = (allgather_fn)sym("ncclAllGather");
    r.send = (sendrecv_fn)sym("ncclSend");
    r.recv = (sendrecv_fn)sym("ncclRecv");
    r.group_start = (group_fn)sym("ncclGroupStart");
    r.group_end = (group_fn)sym("ncclGroupEnd");
    r.ok = r.all_gather && r.send && r.recv && r.group_start && r.group_end;
    return r;
}
Rccl rccl()                                // a copy: five pointers and a flag
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_resolved) { g_table = resolve(g_collective_path); g_resolved = true; }
    return g_table;
}
enum { kUint8 = 1, kUint32 = 3, kUint64 = 5 };           // ncclDataType_t (rccl.h)
}  // namespace

// Which shared library provides ncclAllGather / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd for zk_gather_seekable: NULL or ""
// = RCCL (the default).  Process-wide; resolved at once (an unloadable library is refused here: ZK_ERR_ARGUMENT, the provider in force stays), safe against gathers on other threads.  Replaces round 4's environment
// variable: the library reads no environment.
extern "C" int zk_set_collective_library(const char *path)
{
    const std::string want = path ? path : "";
    Rccl r = resolve(want);                 // (outside the lock: dlopen may take its time)
    if (!want.empty() && !r.ok) return ZK_ERR_ARGUMENT;      // a named library that cannot be loaded, or lacks one of the five: nothing changes
    std::lock_guard<std::mutex> lk(g_mu);
    g_collective_path = want;
    g_table = r;
    g_resolved = true;                      // (RCCL itself may be absent on a single-GPU host: that stays an error of the gather that needs it)
    return 0;
}

extern "C" int zk_gather_seekable(zk_engine *e, void *nccl_comm, int rank, int world, int root, const void *d_payload, uint64_t payload_bytes,
                                  const uint32_t *c_sizes, const uint32_t *d_sizes, uint32_t n_frames, int format, void *d_out, uint64_t out_cap,
                                  uint64_t *out_bytes, zk_seek_table **table_out, void *stream)
{
    if (out_bytes) *out_bytes = 0;
    if (table_out) *table_out = nullptr;
    if (!e || !nccl_comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || (payload_bytes && !d_payload) ||
        (n_frames && (!c_sizes || !d_sizes))) return ZK_ERR_ARGUMENT;
    const Rccl R = rccl();
    if (!R.ok) { e->last_err = "RCCL (librccl.so.1) could not be loaded"; return ZK_ERR_HIP; }
    ZK_HIP(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    int rc;
    // 1. (bytes, frames, room in the destination) of every rank.  The root's capacity travels with the sizes so that EVERY rank
    // reaches the same verdict on "the gathered stream does not fit" before any send / receive is posted -- a root that
    // returned alone would leave its peers inside ncclSend until the communicator times out.
    if ((rc = zk_devbuf_reserve(e, e->st_misc, (size_t)(world + 1) * 32))) return rc;
    uint64_t *d_mine = (uint64_t *)e->st_misc.p, *d_all = d_mine + 4;
    e->h_words[10] = payload_bytes; e->h_words[11] = n_frames; e->h_words[12] = d_out ? out_cap : 0;
    ZK_HIP(hipMemcpyAsync(d_mine, e->h_words + 10, 24, hipMemcpyHostToDevice, st));
    if (R.all_gather(d_mine, d_all, 3, kUint64, nccl_comm, st) != 0) { e->last_err = "ncclAllGather failed"; return ZK_ERR_HIP; }
    std::vector<uint64_t> all3((size_t)world * 3), all((size_t)world * 2);
    ZK_HIP(hipMemcpyAsync(all3.data(), d_all, all3.size() * 8, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    for (int r = 0; r < world; r++) { all[2 * r] = all3[3 * r]; all[2 * r + 1] = all3[3 * r + 1]; }
    std::vector<uint64_t> offs((size_t)world + 1, 0);
    uint64_t mx = 1, total_frames = 0;
    for (int r = 0; r < world; r++) { offs[r + 1] = offs[r] + all[2 * r]; if (all[2 * r + 1] > mx) mx = all[2 * r + 1]; total_frames += all[2 * r + 1]; }
    if (total_frames > ZK_SEEKABLE_MAX_FRAMES) return ZK_ERR_FRAME_INDEX_TOO_LARGE;
    if (offs[world] + 8 * total_frames + 17 > all3[3 * (size_t)root + 2]) return -(int)ZK_E_DST_TOO_SMALL;      // on every rank alike (8 n + 17: seekable_format.md)
    // 2. seek entries, padded to the largest shard
    std::vector<uint32_t> ent((size_t)mx * 2, 0), ents((size_t)world * mx * 2);
    for (uint32_t i = 0; i < n_frames; i++) { ent[i] = c_sizes[i]; ent[mx + i] = d_sizes[i]; }
    if ((rc = zk_devbuf_reserve(e, e->st_off, (size_t)(world + 1) * mx * 8))) return rc;
    uint32_t *d_ent = (uint32_t *)e->st_off.p, *d_ents = d_ent + mx * 2;
    ZK_HIP(hipMemcpyAsync(d_ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice, st));
    if (R.all_gather(d_ent, d_ents, (size_t)mx * 2, kUint32, nccl_comm, st) != 0) { e->last_err = "ncclAllGather failed"; return ZK_ERR_HIP; }
    if (rank == root) ZK_HIP(hipMemcpyAsync(ents.data(), d_ents, ents.size() * 4, hipMemcpyDeviceToHost, st));
    // 3. the payload, peer to peer into the root's buffer
    zeekstd::SeekTable table;
    std::vector<uint8_t> tbytes;
    if (rank == root) {
        ZK_HIP(hipStreamSynchronize(st));                  // the entries
        for (int r = 0; r < world; r++)
            for (uint64_t i = 0; i < all[2 * r + 1]; i++) table.log_frame(ents[(size_t)r * mx * 2 + i], ents[(size_t)r * mx * 2 + mx + i]);
        zeekstd::Serializer ser = table.into_format_serializer(format == ZK_FORMAT_HEAD ? zeekstd::Format::Head : zeekstd::Format::Foot);
        tbytes.resize(ser.encoded_len());
        size_t w = 0;
        for (;;) { const size_t k = ser.write_into(tbytes.data() + w, tbytes.size() - w); if (!k) break; w += k; }
        if (!d_out || offs[world] + tbytes.size() > out_cap) { e->last_err = "gather: table size disagrees with 8 n + 17"; return -(int)ZK_E_GENERIC; }   // (cannot happen: checked collectively above)
    }
    if (R.group_start() != 0) { e->last_err = "ncclGroupStart failed"; return ZK_ERR_HIP; }
    int bad = 0;
    if (rank == root) {
        for (int r = 0; r < world; r++) {
            if (r == rank || !all[2 * r]) continue;
            bad |= R.recv((uint8_t *)d_out + offs[r], (size_t)all[2 * r], kUint8, r, nccl_comm, st);
        }
    } else if (payload_bytes) bad |= R.send(const_cast<void *>(d_payload), (size_t)payload_bytes, kUint8, root, nccl_comm, st);
    bad |= R.group_end();
    if (bad) { e->last_err = "ncclSend / ncclRecv failed"; return ZK_ERR_HIP; }
    if (rank == root) {
        if (payload_bytes) ZK_HIP(hipMemcpyAsync((uint8_t *)d_out + offs[rank], d_payload, payload_bytes, hipMemcpyDeviceToDevice, st));
        // 4. the table behind the last frame
        ZK_HIP(hipMemcpyAsync((uint8_t *)d_out + offs[world], tbytes.data(), tbytes.size(), hipMemcpyHostToDevice, st));
    }
    ZK_HIP(hipStreamSynchronize(st));
    if (rank == root) {
        if (out_bytes) *out_bytes = offs[world] + tbytes.size();
        if (table_out) *table_out = zk_seek_table_from_cpp(&table);
    }
    return 0;
}
