// zk_engine.h -- engine object shared by the decode / encode halves of the C ABI
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "zk_enc_device.h"
#include "zk_kernels.h"

enum { ZK_K_WALK_COUNT = 0, ZK_K_SCAN, ZK_K_WALK_FILL, ZK_K_HUF, ZK_K_FSE, ZK_K_EXEC, ZK_K_XXH64, ZK_K_STATUS,
       ZK_K_ENC_MATCH, ZK_K_ENC_ENTROPY, ZK_K_ENC_COMPACT, ZK_K_ENC_XXH64, ZK_K_ENC_FSE_BUILD, ZK_K_ENC_DENSE, ZK_NKERNELS };

struct zk_devbuf { void *p = nullptr; size_t cap = 0; };
enum { ZK_MAX_CTX = 6 };
// the checksums of a verified batch beside its executor (zk_k_xxh64_follow) or behind it: zk_follow_wanted (zk_engine.hip);
// ZK_CHOICE_XXH64 = 4 asks for "beside" whatever the batch looks like, 1..3 for one of the passes behind the executor
constexpr uint64_t ZK_FOLLOW_MIN_FRAME_BYTES = 512u << 10;

struct zk_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    char devname[320] = {0};
    std::string last_err;
    uint64_t *h_words = nullptr;            // pinned: small read-backs of the encoder (total size)
    // decode contexts: own queues (st + aux: kernels with no mutual dependency overlap, huf || fse), scratch and pinned
    // read-back words, so that several batches can be in flight -- the tail of one (checksum kernel: a per-frame serial
    // chain) overlaps the head of the next.  Context 0 serves the synchronous entry points (its main queue is the engine's
    // `stream`), 0 and 1 zk_decode_submit_dev, all of them the host-pointer pipeline; created on first use.
    struct DecCtx {
        hipStream_t st = nullptr, aux = nullptr;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_exec = nullptr;
        zk_devbuf infos, bases, words, blocks, seqs, lit, prog;     // prog: the executor's progress words (zk_k_xxh64_follow)
        zk_devbuf seg_tab, seg_cnt, seg_holes, seg_tiles;            // the executor in segments: ZkSeg records, per-frame / per-segment counts, hole records, tile counts
        uint64_t *h_words = nullptr;
        bool ready = false;
    } dctx[ZK_MAX_CTX];
    bool slot_busy[2] = {false, false};
    int next_slot = 0;
    // staging for the host-pointer entry points
    zk_devbuf st_comp, st_off, st_dst, st_misc;
    // staged device copy of a raw-content prefix.  The host-pointer Level-A calls upload it on EVERY call (no caching by
    // address: a caller may reuse one buffer for different bases).  A zeekstd::Decoder / Encoder, whose contract is the
    // reference's borrow ("the prefix stays unchanged while it is referenced", decode.rs:201), keeps its upload across its
    // own calls through zk_engine_stage_prefix: owner + address + length name the staged bytes.
    zk_devbuf st_prefix;
    const void *st_prefix_owner = nullptr, *st_prefix_src = nullptr; uint64_t st_prefix_len = 0;
    struct zk_hostpipe *hp = nullptr;       // host-pointer pipeline (pinned staging, copy queues, worker threads): zk_engine_host.hip
    int host_threads = 0;                   // zk_engine_set_host_threads (0 = default)
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    bool profiling = false;
    ZkKernelChoice choice;           // zk_engine_set_kernel_choice: all zero = by batch shape
    uint64_t followed = 0;           // frames of the last finished decode whose checksums zk_k_xxh64_follow verified (zk_engine_checksums_followed)
    int pipe_contexts = 0;           // host pipeline: decode contexts in flight (0 = default) and chunk size, zk_hostpipe_tune
    uint64_t pipe_chunk_bytes = 0;
    hipEvent_t ev_start[ZK_NKERNELS] = {}, ev_stop[ZK_NKERNELS] = {};
    bool ev_used[ZK_NKERNELS] = {};
    float kernel_ms[ZK_NKERNELS] = {};
    // encode scratch
    zk_devbuf enc_a, enc_b, enc_c, enc_d, enc_e, enc_f;
    void *enc_pin = nullptr; size_t enc_pin_cap = 0;   // pinned host copy of the frame / block lists of the encode in flight
    zk_devbuf enc_hist;                     // prefix mode: [prefix tail | frame] records for the matcher
    zk_devbuf enc_seg;                      // frames above ZKE_SEGMENT: the matcher's segment records
    zk_devbuf enc_dense;                    // dense far history (level 0 / >= 3, frames beyond the ring's reach): a candidate per input byte (ZkEncLdm::dense)
    zk_devbuf enc_ldm;                      // prefix beyond the matcher's ring: the long-distance table (ZkEncLdm)
    ZkEncTables enc_tables;
    bool enc_tables_ready = false;
    // second queue of the encoder: the checksum kernel (one serial chain per frame) runs beside the matcher (one workgroup
    // per CU, half of the CU's wave slots free); created on first use
    hipStream_t enc_aux = nullptr;
    hipEvent_t enc_ev_fork = nullptr, enc_ev_join = nullptr;
};

int zk_devbuf_reserve(zk_engine *e, zk_devbuf &b, size_t bytes);

// ---- decode plumbing shared by zk_engine.hip (device-pointer entry points) and zk_engine_host.hip (host pipeline)
struct zk_dec_ctx {
    int slot; hipStream_t st; hipEvent_t ev_exec;
    zk_devbuf &infos, &bases, &words, &blocks, &seqs, &lit;
    uint64_t *h_words;
};
struct zk_dec_args {
    const void *d_comp; uint64_t comp_size; const void *d_c_off, *d_d_off; uint32_t first, count;
    const uint32_t *ids; const uint64_t *out_off;       // frame list (device arrays, both or neither)
    void *d_dst; uint64_t dst_cap; int verify; void *d_frame_status;
    const void *d_prefix; uint64_t prefix_len;
    bool alone = false;                                 // nothing else of this engine is in flight (the synchronous entry points): zk_k_xxh64_follow
    bool single_queue = false;                          // huf and fse on the context's main queue (the host pipeline overlaps whole chunks instead)
    bool mark_exec = false;                             // record the context's ev_exec behind the executor (output bytes final, checksums pending)
};
bool zk_follow_wanted(const zk_engine *e, uint32_t count, uint64_t out_bytes, bool alone);
int zk_dec_ctx_aux(zk_engine *e, int slot);             // the context's second queue + fork / join events, on first use
int zk_dec_ctx_ready(zk_engine *e, int slot);           // creates the context's queues / events on first use
zk_dec_ctx zk_dec_context(zk_engine *e, int slot, void *stream);
int zk_decode_enqueue(zk_engine *e, zk_dec_ctx &c, const zk_dec_args &a);
int zk_decode_finish(zk_engine *e, zk_dec_ctx &c);
namespace zeekstd { class SeekTable; }
struct zk_seek_table;
zk_seek_table *zk_seek_table_from_cpp(const zeekstd::SeekTable *t);
int zk_hostpipe_create(zk_engine *e);
void zk_hostpipe_destroy(zk_engine *e);
void zk_hostpipe_tune(zk_engine *e);                    // applies zk_engine::pipe_contexts / pipe_chunk_bytes (between calls)
enum { ZK_HW_ENC_TOTAL = 8 };               // index into zk_engine::h_words of the encoder's total-size read-back
struct zk_enc_args {
    const void *d_src; uint64_t n; uint32_t frame_size; int level, checksum;
    const void *d_prefix; uint64_t prefix_len; void *d_dst; uint64_t dst_cap; void *d_c_sizes, *d_d_sizes;
};
int zk_encode_enqueue(zk_engine *e, const zk_enc_args &a, hipStream_t st, uint32_t *nf_out);
int zk_encode_finish(zk_engine *e, hipStream_t st, uint64_t *written_out);

// ---- host pipeline, C++ face (the C ABI's host-pointer functions and the zeekstd:: host classes sit on these)
// Where the compressed bytes of a host decode come from: contiguous memory, or a pull callback that delivers the n bytes
// at payload offset `off` straight into pinned staging (Seekable sources: files, callbacks).  Returns bytes delivered.
struct zk_host_src {
    const uint8_t *mem = nullptr;
    size_t (*read)(void *user, uint64_t off, uint8_t *dst, size_t n) = nullptr;
    void *user = nullptr;
};
// Decode frames [first, first + count) into dst (frame `first` at dst[0]); c_off / d_off are the archive's prefix sums and
// the source is addressed with them.  d_prefix: DEVICE copy of the raw-content prefix (or nullptr).  n_ok (optional):
// number of leading frames that decoded fine (bytes of those frames in dst are valid even when the call fails).
int zk_host_decode(zk_engine *e, const zk_host_src &src, const uint64_t *c_off, const uint64_t *d_off, uint32_t first, uint32_t count,
                   const void *d_prefix, uint64_t prefix_len, uint8_t *dst, uint64_t dst_cap, int verify, int32_t *frame_status,
                   uint32_t *n_ok);
// Encode src[0, n) as frames of frame_size bytes; every finished chunk of frames is handed to `sink` (pinned memory,
// valid during the call): the compressed bytes and the chunk's seek entries.  sink returns 0 to go on.
typedef int (*zk_host_sink)(void *user, const uint8_t *data, uint64_t n, const uint32_t *c_sizes, const uint32_t *d_sizes, uint32_t n_frames);
int zk_host_encode(zk_engine *e, const uint8_t *src, uint64_t n, uint32_t frame_size, int level, int checksum, const void *d_prefix,
                   uint64_t prefix_len, zk_host_sink sink, void *user);
// Upload (or reuse) the device copy of a prefix on behalf of `owner`; returns the device pointer in *d_out.
int zk_engine_stage_prefix(zk_engine *e, const void *owner, const uint8_t *prefix, uint64_t len, bool force, const void **d_out);
// multi-threaded memcpy on the engine's worker threads (large host-side copies of the zeekstd:: classes)
void zk_host_copy(zk_engine *e, void *dst, const void *src, size_t n);

// RAII-free helper: brackets one launch with events when profiling is on
struct zk_kernel_timer {
    zk_engine *e; int k; hipStream_t st;
    zk_kernel_timer(zk_engine *e_, int k_, hipStream_t st_) : e(e_), k(k_), st(st_) {
        if (e->profiling) { (void)hipEventRecord(e->ev_start[k], st); e->ev_used[k] = true; }
    }
    ~zk_kernel_timer() { if (e->profiling) (void)hipEventRecord(e->ev_stop[k], st); }
};
void zk_profile_begin(zk_engine *e);
void zk_profile_collect(zk_engine *e);
