// zk_engine.hip -- the batch engine behind the C ABI of include/zeekstd_amd.h (Level A).
// Owns the device scratch (frame infos, block list, sequence records, literal scratch) and
// sequences the kernel pipeline of zk_decode.hip / zk_encode.hip on one HIP stream.
// No CPU fallback: every entry point needs a live gfx950 device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/zeekstd_amd.h"
#include "zk_engine.h"
#include "zk_kernels.h"

#define ZK_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            e->last_err = std::string(#call) + ": " + hipGetErrorString(_e);                         \
            return ZK_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)

int zk_devbuf_reserve(zk_engine *e, zk_devbuf &b, size_t bytes)
{
    if (bytes <= b.cap) return 0;
    if (b.p) ZK_HIP(hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    ZK_HIP(hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

void zk_profile_begin(zk_engine *e)
{
    if (!e->profiling) return;
    for (int k = 0; k < ZK_NKERNELS; k++) { e->ev_used[k] = false; e->kernel_ms[k] = 0.f; }
}
void zk_profile_collect(zk_engine *e)       // call after the stream has been synchronised
{
    if (!e->profiling) return;
    for (int k = 0; k < ZK_NKERNELS; k++)
        if (e->ev_used[k]) { float ms = 0.f; if (hipEventElapsedTime(&ms, e->ev_start[k], e->ev_stop[k]) == hipSuccess) e->kernel_ms[k] = ms; }
}

extern "C" int zk_engine_set_profiling(zk_engine *e, int on)
{
    if (!e) return ZK_ERR_ARGUMENT;
    ZK_HIP(hipSetDevice(e->device));
    if (on && !e->ev_start[0])
        for (int k = 0; k < ZK_NKERNELS; k++) { ZK_HIP(hipEventCreate(&e->ev_start[k])); ZK_HIP(hipEventCreate(&e->ev_stop[k])); }
    e->profiling = on != 0;
    return 0;
}
extern "C" int zk_engine_set_kernel_choice(zk_engine *e, int what, int value)
{
    if (!e) return ZK_ERR_ARGUMENT;
    ZkKernelChoice &k = e->choice;
    switch (what) {
    case ZK_CHOICE_RESET: k = ZkKernelChoice(); e->pipe_contexts = 0; e->pipe_chunk_bytes = 0; zk_hostpipe_tune(e); return 0;   // every key, the host pipeline's two included
    case ZK_CHOICE_FSE_OWN: if (value < 0 || value > 2) return ZK_ERR_ARGUMENT; k.fse_own = value; return 0;
    case ZK_CHOICE_FSE_SHARED: if (value < 0 || value > 3) return ZK_ERR_ARGUMENT; k.fse_shared = value; return 0;
    case ZK_CHOICE_EXEC_LANES: if (value != 0 && value != 128 && value != 256 && value != 512 && value != 1024) return ZK_ERR_ARGUMENT; k.exec_lanes = value; return 0;
    case ZK_CHOICE_EXEC_RING: if (value < 0 || value > 2) return ZK_ERR_ARGUMENT; k.exec_ring = value; return 0;
    case ZK_CHOICE_XXH64: if (value < 0 || value > 5) return ZK_ERR_ARGUMENT; k.xxh = value; return 0;
    case ZK_CHOICE_EXEC_RESIDENT: if (value != 0 && value != 4 && value != 5) return ZK_ERR_ARGUMENT; k.exec_resident = value; return 0;
    case ZK_CHOICE_SMALL_PATH: if (value < 0 || value > 2) return ZK_ERR_ARGUMENT; k.small_path = value; return 0;
    case ZK_CHOICE_EXEC_SEG: if (value < 0 || value > 2) return ZK_ERR_ARGUMENT; k.exec_seg = value; return 0;
    case ZK_CHOICE_SEG_KIB: if (value < 0 || value > 128) return ZK_ERR_ARGUMENT; k.seg_kib = value; return 0;
    case ZK_CHOICE_SEG_FILL: if (value < 0 || value > 3) return ZK_ERR_ARGUMENT; k.seg_fill = value; return 0;
    case ZK_CHOICE_PIPE_CONTEXTS: if (value < 0 || value > ZK_MAX_CTX) return ZK_ERR_ARGUMENT; e->pipe_contexts = value; zk_hostpipe_tune(e); return 0;
    case ZK_CHOICE_PIPE_CHUNK_MIB: if (value < 0 || value > 4096) return ZK_ERR_ARGUMENT; e->pipe_chunk_bytes = (uint64_t)value << 20; zk_hostpipe_tune(e); return 0;
    default: return ZK_ERR_ARGUMENT;
    }
}
extern "C" int zk_engine_set_fse_kernel(zk_engine *e, int mode) { return zk_engine_set_kernel_choice(e, ZK_CHOICE_FSE_OWN, mode); }
extern "C" uint64_t zk_engine_checksums_followed(const zk_engine *e) { return e ? e->followed : 0; }
extern "C" int zk_engine_kernel_count(void) { return ZK_NKERNELS; }
extern "C" const char *zk_engine_kernel_name(int k)
{
    static const char *names[ZK_NKERNELS] = {"zk_k_walk(count)", "zk_k_scan", "zk_k_walk(fill)", "zk_k_huf", "zk_k_fse", "zk_k_exec",
                                             "zk_k_xxh64", "zk_k_status", "zk_k_enc_match", "zk_k_enc_entropy", "zk_k_enc_compact", "zk_k_enc_xxh64", "zk_k_enc_fse_build", "zk_k_enc_dense_cand"};
    return k >= 0 && k < ZK_NKERNELS ? names[k] : "";
}
extern "C" int zk_engine_kernel_times(const zk_engine *e, float *ms_out, int n)
{
    if (!e || !ms_out) return ZK_ERR_ARGUMENT;
    for (int k = 0; k < n && k < ZK_NKERNELS; k++) ms_out[k] = e->kernel_ms[k];
    return 0;
}

extern "C" int zk_abi_version(void) { return ZK_ABI_VERSION; }

extern "C" const char *zk_error_name(int code)
{
    switch (code) {
    case 0: return "No error detected";
    case -1: return "Error (generic)";
    case -10: return "Unknown frame descriptor";
    case -12: return "Version not supported";
    case -14: return "Unsupported frame parameter";
    case -16: return "Frame requires too much memory for decoding";
    case -20: return "Data corruption detected";
    case -22: return "Restored data doesn't match checksum";
    case -30: return "Dictionary is corrupted";
    case -32: return "Dictionary mismatch";
    case -40: return "Unsupported parameter";
    case -42: return "Parameter is out of bound";
    case -64: return "Allocation error : not enough memory";
    case -70: return "Destination buffer is too small";
    case -72: return "Src size is incorrect";
    case -74: return "Operation on NULL destination buffer";
    case ZK_ERR_OFFSET_OUT_OF_RANGE: return "offset out of range";
    case ZK_ERR_FRAME_INDEX_TOO_LARGE: return "frame index too large";
    case ZK_ERR_NUMBER_CONVERSION: return "number conversion failed";
    case ZK_ERR_IO: return "io error";
    case ZK_ERR_HIP: return "HIP runtime error";
    case ZK_ERR_NO_DEVICE: return "no usable gfx950 device";
    case ZK_ERR_ARGUMENT: return "invalid argument";
    default: return "Unspecified error code";
    }
}

// The runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and a copy queue that lands
// on the hardware queue of a compute queue serialises behind its kernels (the host pipeline then runs at 29 instead of
// 46 GiB/s).  The engine owns up to six queues, so it asks for eight -- effective when this library is loaded before the
// process initialises HIP; an explicit setting of the user wins.
__attribute__((constructor)) static void zk_ask_for_hw_queues(void) { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" int zk_engine_create(int device, zk_engine **out)
{
    if (!out) return ZK_ERR_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ZK_ERR_NO_DEVICE;
    zk_engine *e = new zk_engine();
    e->device = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { delete e; return ZK_ERR_NO_DEVICE; }
    snprintf(e->devname, sizeof e->devname, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete e; return ZK_ERR_NO_DEVICE; }   // kernels are built for gfx950 only
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return ZK_ERR_NO_DEVICE; }
    if (hipHostMalloc((void **)&e->h_words, 16 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) { (void)hipStreamDestroy(e->stream); delete e; return ZK_ERR_HIP; }
    if (zk_dec_ctx_ready(e, 0) != 0 || zk_dec_ctx_ready(e, 1) != 0) { zk_engine_destroy(e); return ZK_ERR_HIP; }
    // the host pipeline's two copy queues are created right behind the two compute queues, before any second queue of a
    // context: streams are dealt onto the hardware queues in creation order
    if (zk_hostpipe_create(e) != 0) { zk_engine_destroy(e); return ZK_ERR_HIP; }
    *out = e;
    return 0;
}

extern "C" void zk_engine_destroy(zk_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    zk_hostpipe_destroy(e);
    if (e->enc_pin) (void)hipHostFree(e->enc_pin);
    for (auto &c : e->dctx) if (c.st) (void)hipStreamSynchronize(c.st);
    zk_devbuf *bufs[] = {&e->st_prefix, &e->st_comp, &e->st_off, &e->st_dst, &e->st_misc,
                         &e->enc_a, &e->enc_b, &e->enc_c, &e->enc_d, &e->enc_e, &e->enc_f, &e->enc_hist, &e->enc_seg, &e->enc_ldm, &e->enc_dense};
    for (zk_devbuf *b : bufs) if (b->p) (void)hipFree(b->p);
    if (e->h_words) (void)hipHostFree(e->h_words);
    for (int k = 0; k < ZK_NKERNELS; k++) { if (e->ev_start[k]) (void)hipEventDestroy(e->ev_start[k]); if (e->ev_stop[k]) (void)hipEventDestroy(e->ev_stop[k]); }
    for (int i = 0; i < ZK_MAX_CTX; i++) {
        zk_engine::DecCtx &c = e->dctx[i];
        for (zk_devbuf *b : {&c.infos, &c.bases, &c.words, &c.blocks, &c.seqs, &c.lit, &c.prog, &c.seg_tab, &c.seg_cnt, &c.seg_holes, &c.seg_tiles}) if (b->p) (void)hipFree(b->p);
        if (c.h_words) (void)hipHostFree(c.h_words);
        for (hipEvent_t ev : {c.ev_fork, c.ev_join, c.ev_exec}) if (ev) (void)hipEventDestroy(ev);
        if (c.aux) (void)hipStreamDestroy(c.aux);
        if (c.st && i != 0) (void)hipStreamDestroy(c.st);     // context 0 runs on the engine's own stream
    }
    if (e->enc_aux) { (void)hipStreamSynchronize(e->enc_aux); (void)hipStreamDestroy(e->enc_aux); }
    for (hipEvent_t ev : {e->enc_ev_fork, e->enc_ev_join}) if (ev) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" const char *zk_engine_last_hip_error(const zk_engine *e) { return e ? e->last_err.c_str() : ""; }
extern "C" const char *zk_engine_device_name(const zk_engine *e) { return e ? e->devname : ""; }

// ---------------------------------------------------------------------------------------------- decode
// ids / out_off (device arrays, both or neither): frame f of the batch is archive frame ids[f]; its bytes go to dst + out_off[f]
// One decode in flight per context: queues, scratch and pinned read-back words.  Context 0 is the engine's own (synchronous
// entry points, optionally on the caller's stream), context 1 exists for zk_decode_submit_dev; the host-pointer pipeline
// (zk_engine_host.hip) alternates between the two.
int zk_dec_ctx_ready(zk_engine *e, int slot)
{
    if (slot < 0 || slot >= ZK_MAX_CTX) return ZK_ERR_ARGUMENT;
    zk_engine::DecCtx &c = e->dctx[slot];
    if (c.ready) return 0;
    if (slot == 0) c.st = e->stream;
    else ZK_HIP(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
    ZK_HIP(hipEventCreateWithFlags(&c.ev_exec, hipEventDisableTiming));
    ZK_HIP(hipHostMalloc((void **)&c.h_words, 16 * sizeof(uint64_t), hipHostMallocDefault));
    c.ready = true;
    return 0;
}
// the second queue of a context (huf || fse of one batch) exists only once a batch asked for it: the host pipeline overlaps
// whole chunks on one queue per context, and every stream the process owns competes for the runtime's few hardware queues
int zk_dec_ctx_aux(zk_engine *e, int slot)
{
    zk_engine::DecCtx &c = e->dctx[slot];
    if (c.aux) return 0;
    ZK_HIP(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
    ZK_HIP(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
    ZK_HIP(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
    return 0;
}
zk_dec_ctx zk_dec_context(zk_engine *e, int slot, void *stream)
{
    zk_engine::DecCtx &c = e->dctx[slot];
    return zk_dec_ctx{slot, slot == 0 && stream ? (hipStream_t)stream : c.st, c.ev_exec, c.infos, c.bases, c.words, c.blocks, c.seqs, c.lit, c.h_words};
}

// Where the checksums of a verified batch run: beside the executor (zk_k_xxh64_follow) or behind it.  Measured on 16 / 128 / 512 /
// 2048 frames of 2 MiB (profiles/r04_follow_by_batch_size.txt; ms one batch at a time | two in flight; behind = the better of
// the two passes behind the executor):   16: 4.27 -> 3.50 | 2.19 -> 1.91     128: 5.20 -> 5.77 | 3.03 -> 3.07
//                                       512: 6.48 -> 6.76 | 5.32 -> 5.13    2048: 16.7 -> 15.7 | 14.56 -> 14.49 (at four executor
// workgroups per CU; five, the in-flight default, and a checksum wave do not fit a SIMD).  A few frames leave most CUs to the checksum
// waves; a batch that fills the device alone trades 2.7 ms of an idle device for 1.7 ms of the executor; in between a frame IS a
// workgroup and the slowest one -- the one that shares its SIMD -- ends the kernel.  Frames of less than 512 KiB are short chains.
bool zk_follow_wanted(const zk_engine *e, uint32_t count, uint64_t out_bytes, bool alone)
{
    if (e->profiling) return false;                         // (per-kernel timing serialises the kernels)
    if (e->choice.xxh) return e->choice.xxh == 4;
    if (out_bytes < (uint64_t)count * ZK_FOLLOW_MIN_FRAME_BYTES) return false;
    if (count <= 64) return true;
    return alone ? count >= 1024 : count < 1024;
}

// Several workgroups per frame (zk_k_exec_seg + zk_k_exec_fill) instead of one (zk_k_exec)?  Never with a prefix (history below the frame's
// first byte is the serial kernel's), never when a frame could have more segments than a grid has rows.
static bool zk_seg_wanted(const zk_engine *e, const zk_dec_args &a, uint32_t count, uint64_t out_bytes, uint64_t max_frame, uint64_t nblocks, bool follow)
{
    if (a.d_prefix || e->choice.exec_seg == 1 || !count || !nblocks) return false;
    const uint32_t seg_bytes = e->choice.seg_kib ? (uint32_t)e->choice.seg_kib << 10 : ZK_SEG_BYTES;
    if (2 * ((max_frame + seg_bytes - 1) / seg_bytes) + 1 > 65535) return false;
    if (e->choice.exec_seg == 2) return true;
    // by batch shape: a handful of long frames, where a frame as ONE workgroup leaves the device idle (2 MiB frames, HBM-resident,
    // unverified, ms: 1 / 5 / 16 / 32 frames 2.58 / 2.60 / 2.62 / 2.65 -> 1.79 / 1.83 / 1.85 / 2.39; 64 frames 2.80 -> 2.88: profiles/r06_seg_probe.txt).
    // Not where the checksums run beside the executor: a frame's four XXH64 chains (1.7-2.3 ms per 2 MiB, whoever runs them) then end
    // the decode, not the executor (verified, 16 frames: 3.52 ms either way).
    // (r6, with a wave per frame behind the progress words -- zk_k_xxh64_follow1 -- verified, ms, frame executor | segments: 1 frame 2.94 | 3.01,
    //  5: 3.05 | 3.14, 16: 3.08 | 3.31, 32: 3.69 | 3.33)
    return (!follow || count > 16) && count <= 32 && out_bytes >= (uint64_t)count * (4u * ZK_SEG_BYTES);
}

// Enqueue the whole decode on the context's queues.  Blocks the host once, for the block / sequence / literal totals
// that size the scratch (40 bytes, after the two cheapest kernels); returns with the rest still running.
int zk_decode_enqueue(zk_engine *e, zk_dec_ctx &c, const zk_dec_args &a)
{
    // history positions are 32-bit words biased by 2^30 (zk_device.h): a frame plus its prefix must fit below that
    if (a.d_prefix && a.prefix_len > ZK_MAX_PREFIX) return -(int)ZK_E_WINDOW_TOO_LARGE;
    hipStream_t st = c.st;
    const uint8_t *comp = (const uint8_t *)a.d_comp;
    const uint64_t *c_off = (const uint64_t *)a.d_c_off, *d_off = (const uint64_t *)a.d_d_off;
    const uint32_t first = a.first, count = a.count;
    int rc;
    if ((rc = zk_devbuf_reserve(e, c.infos, (size_t)count * sizeof(ZkFrameInfo)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.bases, (size_t)count * sizeof(ZkFrameBase)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.words, 16 * sizeof(uint64_t)))) return rc;
    ZkFrameInfo *infos = (ZkFrameInfo *)c.infos.p;
    ZkFrameBase *bases = (ZkFrameBase *)c.bases.p;
    uint64_t *words = (uint64_t *)c.words.p;          // [0..2] totals, [3] first error

    zk_profile_begin(e);
    { zk_kernel_timer t(e, ZK_K_WALK_COUNT, st); zk_launch_walk(st, comp, a.comp_size, c_off, d_off, first, count, a.ids, a.out_off, a.dst_cap, nullptr, nullptr, infos); }
    { zk_kernel_timer t(e, ZK_K_SCAN, st); zk_launch_scan(st, infos, count, bases, words, d_off, first, a.out_off); }
    ZK_HIP(hipMemcpyAsync(c.h_words, words, 9 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    const uint64_t nblocks = c.h_words[0], nseq = c.h_words[1], nlit = c.h_words[2];
    const bool verify = a.verify && c.h_words[8] != 0;    // (a batch without a Content_Checksum: zeekstd's library default, encode.rs:163-167 -- no checksum kernel is launched)
    const uint32_t n_own = (uint32_t)c.h_words[4];        // blocks that need per-block sequence tables
    const bool dense = nseq * 10 > c.h_words[5];          // fewer than 10 output bytes per sequence (zk_launch_exec)
    const uint64_t out_bytes = c.h_words[5], max_frame = c.h_words[7];
    if (nblocks > 0xFFFFFFF0ull) return -(int)ZK_E_GENERIC;
    if ((rc = zk_devbuf_reserve(e, c.blocks, (size_t)(nblocks + 1) * sizeof(ZkBlock)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.seqs, (size_t)(nseq + 1) * sizeof(ZkSeqP)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.lit, (size_t)nlit + 64))) return rc;
    ZkBlock *blocks = (ZkBlock *)c.blocks.p;
    ZkSeqP *seqs = (ZkSeqP *)c.seqs.p;
    uint8_t *lit = (uint8_t *)c.lit.p;

    c.h_words[3] = ~0ull;
    ZK_HIP(hipMemcpyAsync(words + 3, c.h_words + 3, sizeof(uint64_t), hipMemcpyHostToDevice, st));
    // checksums WHILE the executor writes (zk_k_xxh64_follow, zk_decode.hip): the executor publishes a progress word per frame, the
    // checksum waves run on the context's second queue beside it, the ordinary pass behind the executor takes what they left
    // (not in the host pipeline's chunks: they overlap whole chunks on one queue per context, PCIe bounds them, and the extra queues
    //  cost the copy queues 1-2 %: 46.4 -> 45.3 GiB/s end to end)
    const bool follow = verify && !a.single_queue && zk_follow_wanted(e, count, c.h_words[5], a.alone);      // (in segments: behind the fill pass's progress words)
    // The executor in segments (several workgroups per frame; zk_device.h): where a frame is long and the frames alone do not fill the device.
    ZkSegScratch sgs{};
    const bool seg = zk_seg_wanted(e, a, count, out_bytes, max_frame, nblocks, follow);
    if (seg) {
        zk_engine::DecCtx &x = e->dctx[c.slot];
        sgs.seg_bytes = e->choice.seg_kib ? (uint32_t)e->choice.seg_kib << 10 : ZK_SEG_BYTES;
        sgs.max_segs = 2u * (uint32_t)((max_frame + sgs.seg_bytes - 1) / sgs.seg_bytes) + 1u;
        const uint64_t nsg = (uint64_t)count * sgs.max_segs;
        if ((rc = zk_devbuf_reserve(e, x.seg_tab, (size_t)nsg * sizeof(ZkSeg)))) return rc;
        if ((rc = zk_devbuf_reserve(e, x.seg_cnt, (size_t)(nsg + count) * sizeof(uint32_t)))) return rc;
        if ((rc = zk_devbuf_reserve(e, x.seg_holes, (size_t)((out_bytes >> 2) + 16 * nsg + 16) * sizeof(ZkHole)))) return rc;
        if ((rc = zk_devbuf_reserve(e, x.seg_tiles, (size_t)((out_bytes >> 10) + 2 * nblocks + 8 * nsg + 16) * sizeof(uint32_t)))) return rc;
        sgs.segs = (ZkSeg *)x.seg_tab.p; sgs.nsegs = (uint32_t *)x.seg_cnt.p; sgs.segn = sgs.nsegs + count;
        sgs.holes = (ZkHole *)x.seg_holes.p; sgs.tilecnt = (uint32_t *)x.seg_tiles.p;
    }
    ZkKernelChoice kc = e->choice;
    if (follow && !kc.exec_resident) kc.exec_resident = 4;
    uint64_t *prog = nullptr;
    if (follow) {
        zk_engine::DecCtx &x = e->dctx[c.slot];
        if ((rc = zk_devbuf_reserve(e, x.prog, (size_t)count * sizeof(uint64_t)))) return rc;
        if ((rc = zk_dec_ctx_aux(e, c.slot))) return rc;
        prog = (uint64_t *)x.prog.p;
        ZK_HIP(hipMemsetAsync(prog, 0, (size_t)count * sizeof(uint64_t), st));
    }
    ZK_HIP(hipMemsetAsync(words + 6, 0, sizeof(uint64_t), st));
    { zk_kernel_timer t(e, ZK_K_WALK_FILL, st); zk_launch_walk(st, comp, a.comp_size, c_off, d_off, first, count, a.ids, a.out_off, a.dst_cap, bases, blocks, infos); }
    // literals (huf) and sequences (fse) of a block are independent: the two kernels run side by side on two queues;
    // with per-kernel timing on they are serialised instead
    if (e->profiling || a.single_queue) {
        { zk_kernel_timer t(e, ZK_K_HUF, st); zk_launch_huf(st, comp, blocks, (uint32_t)nblocks, lit); }
        { zk_kernel_timer t(e, ZK_K_FSE, st); zk_launch_fse(st, comp, blocks, (uint32_t)nblocks, n_own, seqs, e->choice, count); }
    } else {
        if ((rc = zk_dec_ctx_aux(e, c.slot))) return rc;
        zk_engine::DecCtx &x = e->dctx[c.slot];
        ZK_HIP(hipEventRecord(x.ev_fork, st));
        ZK_HIP(hipStreamWaitEvent(x.aux, x.ev_fork, 0));
        zk_launch_huf(x.aux, comp, blocks, (uint32_t)nblocks, lit);
        ZK_HIP(hipEventRecord(x.ev_join, x.aux));
        zk_launch_fse(st, comp, blocks, (uint32_t)nblocks, n_own, seqs, e->choice, count);
        ZK_HIP(hipStreamWaitEvent(st, x.ev_join, 0));
    }
    const uint64_t *x_off = a.out_off ? a.out_off : d_off;   // packed indexed output: out_off (count + 1 prefix sums) doubles as the d_off of the checksum kernels
    const uint32_t x_first = a.out_off ? 0 : first;
    if (follow) {
        zk_engine::DecCtx &x = e->dctx[c.slot];
        ZK_HIP(hipEventRecord(x.ev_fork, st));               // (in front of the executor: the checksum waves start with it)
        ZK_HIP(hipStreamWaitEvent(x.aux, x.ev_fork, 0));
    }
    if (seg) { zk_kernel_timer t(e, ZK_K_EXEC, st); zk_launch_exec_seg(st, comp, d_off, first, count, a.ids, a.out_off, blocks, bases, infos, seqs, lit, (uint8_t *)a.d_dst, sgs, kc, dense, prog); }
    else { zk_kernel_timer t(e, ZK_K_EXEC, st); zk_launch_exec(st, comp, d_off, first, count, a.ids, a.out_off, blocks, bases, infos, seqs, lit, (uint8_t *)a.d_dst, (const uint8_t *)a.d_prefix, a.d_prefix ? a.prefix_len : 0, kc, dense, prog); }
    if (a.mark_exec) ZK_HIP(hipEventRecord(c.ev_exec, st));
    if (follow) {
        // enqueued BEHIND the executor's launch: were the two queues ever served one after the other, the checksum waves would find
        // finished frames, not wait (ZK_FOLLOW_PATIENCE) for an executor that cannot start
        zk_engine::DecCtx &x = e->dctx[c.slot];
        zk_launch_xxh64_follow(x.aux, (const uint8_t *)a.d_dst, x_off, x_first, count, infos, prog);
        ZK_HIP(hipEventRecord(x.ev_join, x.aux));
        ZK_HIP(hipStreamWaitEvent(st, x.ev_join, 0));
        zk_launch_xxh64(st, (const uint8_t *)a.d_dst, x_off, x_first, count, infos, nullptr, e->choice, prog);
    } else if (verify) { zk_kernel_timer t(e, ZK_K_XXH64, st); zk_launch_xxh64(st, (const uint8_t *)a.d_dst, x_off, x_first, count, infos, nullptr, e->choice); }
    { zk_kernel_timer t(e, ZK_K_STATUS, st); zk_launch_status(st, infos, count, (int32_t *)a.d_frame_status, words + 3, prog, words + 6); }
    ZK_HIP(hipMemcpyAsync(c.h_words + 3, words + 3, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    return 0;
}
int zk_decode_finish(zk_engine *e, zk_dec_ctx &c)
{
    ZK_HIP(hipStreamSynchronize(c.st));
    ZK_HIP(hipGetLastError());
    zk_profile_collect(e);
    e->followed = c.h_words[6] & 0xFFFFFFFFull;          // (the high half counts frames the checksum waves hashed and found different)
    if (c.h_words[3] != ~0ull) return -(int)(uint32_t)(c.h_words[3] & 0xFFFFFFFFu);
    return 0;
}

static int zk_decode_impl(zk_engine *e, const zk_dec_args &a, void *stream)
{
    if (!e || (a.count && (!a.d_comp || !a.d_c_off || !a.d_d_off || !a.d_dst))) return ZK_ERR_ARGUMENT;
    if (a.count == 0) return 0;
    if (e->slot_busy[0]) return ZK_ERR_ARGUMENT;            // a submitted batch still owns context 0: zk_decode_wait first
    ZK_HIP(hipSetDevice(e->device));
    zk_dec_ctx c = zk_dec_context(e, 0, stream);
    zk_dec_args b = a;
    b.alone = !e->slot_busy[1];                             // (a batch submitted on the other context would be its neighbour)
    int rc = zk_decode_enqueue(e, c, b);
    if (rc) return rc;
    return zk_decode_finish(e, c);
}

extern "C" int zk_decode_submit_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off,
                                    const void *d_d_off, uint32_t first, uint32_t count, void *d_dst, uint64_t dst_cap,
                                    int verify, void *d_frame_status, int *slot_out)
{
    if (!e || !slot_out || count == 0 || !d_comp || !d_c_off || !d_d_off || !d_dst) return ZK_ERR_ARGUMENT;
    const int slot = e->next_slot;
    if (e->slot_busy[slot]) return ZK_ERR_ARGUMENT;         // both contexts in flight: zk_decode_wait the older one first
    ZK_HIP(hipSetDevice(e->device));
    zk_dec_ctx c = zk_dec_context(e, slot, nullptr);
    const bool prof = e->profiling;
    e->profiling = false;                                   // per-kernel events belong to the synchronous path
    zk_dec_args a{d_comp, comp_size, d_c_off, d_d_off, first, count, nullptr, nullptr, d_dst, dst_cap, verify, d_frame_status, nullptr, 0};
    int rc = zk_decode_enqueue(e, c, a);
    e->profiling = prof;
    if (rc) { (void)hipStreamSynchronize(c.st); return rc; }
    e->slot_busy[slot] = true;
    e->next_slot = slot ^ 1;
    *slot_out = slot;
    return 0;
}

extern "C" int zk_decode_wait(zk_engine *e, int slot)
{
    if (!e || slot < 0 || slot > 1 || !e->slot_busy[slot]) return ZK_ERR_ARGUMENT;
    ZK_HIP(hipSetDevice(e->device));
    zk_dec_ctx c = zk_dec_context(e, slot, nullptr);
    const bool prof = e->profiling;
    e->profiling = false;
    const int rc = zk_decode_finish(e, c);
    e->profiling = prof;
    e->slot_busy[slot] = false;
    return rc;
}

extern "C" int zk_decode_frames_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off,
                                    const void *d_d_off, uint32_t first, uint32_t count, void *d_dst, uint64_t dst_cap,
                                    int verify, void *d_frame_status, void *stream)
{
    zk_dec_args a{d_comp, comp_size, d_c_off, d_d_off, first, count, nullptr, nullptr, d_dst, dst_cap, verify, d_frame_status, nullptr, 0};
    return zk_decode_impl(e, a, stream);
}

extern "C" int zk_decode_frames_prefix_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off,
                                           const void *d_d_off, uint32_t first, uint32_t count, const void *d_prefix, uint64_t prefix_len,
                                           void *d_dst, uint64_t dst_cap, int verify, void *d_frame_status, void *stream)
{
    zk_dec_args a{d_comp, comp_size, d_c_off, d_d_off, first, count, nullptr, nullptr, d_dst, dst_cap, verify, d_frame_status,
                  prefix_len ? d_prefix : nullptr, prefix_len};
    return zk_decode_impl(e, a, stream);
}

extern "C" int zk_decode_frame_list_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off, const void *d_d_off,
                                        const void *d_ids, const void *d_out_off, uint32_t count, void *d_dst, uint64_t dst_cap,
                                        int verify, void *d_frame_status, void *stream)
{
    if (count && (!d_ids || !d_out_off)) return ZK_ERR_ARGUMENT;
    zk_dec_args a{d_comp, comp_size, d_c_off, d_d_off, 0, count, (const uint32_t *)d_ids, (const uint64_t *)d_out_off, d_dst, dst_cap, verify,
                  d_frame_status, nullptr, 0};
    return zk_decode_impl(e, a, stream);
}

// ---------------------------------------------------------------------------------------------- frame sizes nobody knows yet
// What libzstd's streaming decoder needs no table for -- how many bytes a frame decodes to -- this engine is told by the seek table
// (lib/src/seek_table.rs:750).  A host that holds frames WITHOUT their entries (the Level-C shim under an unmodified zeekstd hands
// over what ZSTD_decompressStream receives, lib/src/decode.rs:243-245) asks first: header walk + sequence walks, no literals, no
// output.  A frame that carries Frame_Content_Size is walked like any other and its blocks' sum is held against the header's field
// (zk_k_frame_sizes: corruption_detected when they differ) -- the sizes reported are always what the blocks regenerate.
// (The host-pointer variant below relies on context 0's queue being the engine's own stream: its uploads and this call's kernels are
//  one queue, in order.)
extern "C" int zk_frame_content_sizes_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off, uint32_t first, uint32_t count,
                                          void *d_sizes, void *d_frame_status, void *stream)
{
    if (!e || (count && (!d_comp || !d_c_off || !d_sizes || !d_frame_status))) return ZK_ERR_ARGUMENT;
    if (count == 0) return 0;
    if (e->slot_busy[0]) return ZK_ERR_ARGUMENT;
    ZK_HIP(hipSetDevice(e->device));
    zk_dec_ctx c = zk_dec_context(e, 0, stream);
    hipStream_t st = c.st;
    const uint8_t *comp = (const uint8_t *)d_comp;
    const uint64_t *c_off = (const uint64_t *)d_c_off;
    int rc;
    if ((rc = zk_devbuf_reserve(e, c.infos, (size_t)count * sizeof(ZkFrameInfo)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.bases, (size_t)count * sizeof(ZkFrameBase)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.words, 16 * sizeof(uint64_t)))) return rc;
    ZkFrameInfo *infos = (ZkFrameInfo *)c.infos.p;
    ZkFrameBase *bases = (ZkFrameBase *)c.bases.p;
    uint64_t *words = (uint64_t *)c.words.p;
    zk_launch_walk(st, comp, comp_size, c_off, nullptr, first, count, nullptr, nullptr, 0, nullptr, nullptr, infos);
    zk_launch_scan(st, infos, count, bases, words, nullptr, first, nullptr);
    ZK_HIP(hipMemcpyAsync(c.h_words, words, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    const uint64_t nblocks = c.h_words[0], nseq = c.h_words[1];
    if (nblocks > 0xFFFFFFF0ull) return -(int)ZK_E_GENERIC;
    if ((rc = zk_devbuf_reserve(e, c.blocks, (size_t)(nblocks + 1) * sizeof(ZkBlock)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.seqs, (size_t)(nseq + 1) * sizeof(ZkSeqP)))) return rc;
    ZkBlock *blocks = (ZkBlock *)c.blocks.p;
    zk_launch_walk(st, comp, comp_size, c_off, nullptr, first, count, nullptr, nullptr, 0, bases, blocks, infos);
    zk_launch_fse(st, comp, blocks, (uint32_t)nblocks, (uint32_t)c.h_words[4], (ZkSeqP *)c.seqs.p, e->choice, count);
    zk_launch_frame_sizes(st, infos, bases, blocks, count, (uint64_t *)d_sizes, (int32_t *)d_frame_status);
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    return 0;
}
// host pointers: comp[c_off[first] .. c_off[first + count]) is uploaded, sizes[count] and frame_status[count] come back
extern "C" int zk_frame_content_sizes(zk_engine *e, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off, uint32_t first, uint32_t count,
                                      uint64_t *sizes, int32_t *frame_status)
{
    if (!e || (count && (!comp || !c_off || !sizes || !frame_status))) return ZK_ERR_ARGUMENT;
    if (count == 0) return 0;
    const uint64_t lo = c_off[first], hi = c_off[first + count];
    if (hi < lo || hi > comp_size) return -(int)ZK_E_SRC_SIZE_WRONG;
    ZK_HIP(hipSetDevice(e->device));
    int rc;
    std::vector<uint64_t> rel(count + 1);
    for (uint32_t i = 0; i <= count; i++) rel[i] = c_off[first + i] - lo;
    if ((rc = zk_devbuf_reserve(e, e->st_comp, (size_t)(hi - lo) + 64))) return rc;
    if ((rc = zk_devbuf_reserve(e, e->st_off, (size_t)(count + 1) * 8))) return rc;
    if ((rc = zk_devbuf_reserve(e, e->st_misc, (size_t)count * 12))) return rc;
    hipStream_t st = e->stream;
    ZK_HIP(hipMemsetAsync((uint8_t *)e->st_comp.p + (hi - lo), 0, 64, st));                  // readable padding behind the last frame
    ZK_HIP(hipMemcpyAsync(e->st_comp.p, comp + lo, hi - lo, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(e->st_off.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, st));
    uint64_t *d_sizes = (uint64_t *)e->st_misc.p;
    int32_t *d_stat = (int32_t *)(d_sizes + count);
    if ((rc = zk_frame_content_sizes_dev(e, e->st_comp.p, hi - lo, e->st_off.p, 0, count, d_sizes, d_stat, nullptr))) return rc;
    ZK_HIP(hipMemcpy(sizes, d_sizes, (size_t)count * 8, hipMemcpyDeviceToHost));
    ZK_HIP(hipMemcpy(frame_status, d_stat, (size_t)count * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------------- XXH64
extern "C" int zk_xxh64_frames_dev(zk_engine *e, const void *d_data, const void *d_off, uint32_t count, void *d_out, void *stream)
{
    if (!e || (count && (!d_off || !d_out))) return ZK_ERR_ARGUMENT;
    if (count == 0) return 0;
    ZK_HIP(hipSetDevice(e->device));
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    zk_launch_xxh64(st, (const uint8_t *)d_data, (const uint64_t *)d_off, 0, count, nullptr, (uint64_t *)d_out, e->choice);
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    return 0;
}

