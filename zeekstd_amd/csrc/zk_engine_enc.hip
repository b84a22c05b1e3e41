// zk_engine_enc.hip -- encode half of the batch engine (Level A of include/zeekstd_amd.h).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/zeekstd_amd.h"
#include "zk_engine.h"
#include "zk_kernels.h"
#include "zk_enc_plan.h"

#define ZK_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            e->last_err = std::string(#call) + ": " + hipGetErrorString(_e);                         \
            return ZK_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)

// ---------------------------------------------------------------- predefined FSE compression tables (host, once)
void zk_build_enc_tables(ZkEncTables *t)
{
    static const int16_t ll[36] = ZK_LL_DEFNORM, of[29] = ZK_OF_DEFNORM, ml[53] = ZK_ML_DEFNORM;
    static const uint32_t llv[36] = ZK_LL_TABLE, mlv[53] = ZK_ML_TABLE;
    memset(t, 0, sizeof *t);
    uint8_t sym[512]; int32_t cumul[66];
    zke_build_ctable(ll, 36, 6, t->ll_state, t->ll_dfs, t->ll_dnb, sym, cumul);
    zke_build_ctable(of, 29, 5, t->of_state, t->of_dfs, t->of_dnb, sym, cumul);
    zke_build_ctable(ml, 53, 6, t->ml_state, t->ml_dfs, t->ml_dnb, sym, cumul);
    for (int i = 0; i < 36; i++) t->ll_val[i] = llv[i];
    for (int i = 0; i < 53; i++) t->ml_val[i] = mlv[i];
    t->al[0] = 6; t->al[1] = 5; t->al[2] = 6;
}

extern "C" uint64_t zk_compress_bound(uint64_t n, uint32_t frame_size)
{
    if (frame_size == 0) return 0;
    const uint64_t nf = n == 0 ? 1 : (n + frame_size - 1) / frame_size;
    const uint64_t blocks_per_frame = ((uint64_t)frame_size + 1023) / 1024 + 8;     // blocks are >= 1 KiB except in tiny frames
    return n + nf * (6 + 4 + 3 * blocks_per_frame) + 16;
}

extern "C" int zk_encode_frames_dev(zk_engine *e, const void *d_src, uint64_t n, uint32_t frame_size, int level, int checksum,
                                    void *d_dst, uint64_t dst_cap, void *d_c_sizes, void *d_d_sizes, uint32_t *n_frames_out,
                                    uint64_t *written_out, void *stream)
{
    return zk_encode_frames_prefix_dev(e, d_src, n, frame_size, level, checksum, nullptr, 0, d_dst, dst_cap, d_c_sizes, d_d_sizes,
                                       n_frames_out, written_out, stream);
}

static int zk_pin_reserve(zk_engine *e, size_t bytes)
{
    if (bytes <= e->enc_pin_cap) return 0;
    if (e->enc_pin) ZK_HIP(hipHostFree(e->enc_pin));
    e->enc_pin = nullptr; e->enc_pin_cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    ZK_HIP(hipHostMalloc(&e->enc_pin, want, hipHostMallocDefault));
    e->enc_pin_cap = want;
    return 0;
}

// Enqueue one encode on `st`.  With dst_cap >= zk_compress_bound(n, frame_size) nothing blocks: the total size arrives in
// the engine's pinned word ZK_HW_ENC_TOTAL once the stream has run (zk_encode_finish); a smaller destination needs the
// total before the frames may be assembled, so the stream is synchronised once in the middle.
// The frame / block lists are built in pinned host memory that stays untouched until the next enqueue: one encode in flight.
int zk_encode_enqueue(zk_engine *e, const zk_enc_args &a, hipStream_t st, uint32_t *nf_out)
{
    const uint64_t n = a.n;
    const uint32_t frame_size = a.frame_size;
    // the matcher sees the last `hist` bytes of the prefix right before every frame
    const uint32_t hist = a.d_prefix ? zke_prefix_hist(a.prefix_len) : 0;
    if (!e || frame_size == 0 || frame_size > ZK_SEEKABLE_MAX_FRAME_SIZE || !a.d_dst || (n && !a.d_src)) return ZK_ERR_ARGUMENT;
    const uint64_t nf64 = n == 0 ? 1 : (n + frame_size - 1) / frame_size;
    if (nf64 > ZK_SEEKABLE_MAX_FRAMES) return ZK_ERR_FRAME_INDEX_TOO_LARGE;
    ZK_HIP(hipSetDevice(e->device));

    // frame / block / segment lists (host arithmetic only: zk_enc_plan.h)
    ZkEncPlan pl;
    if (!zke_plan_count(n, frame_size, hist, &pl)) return -(int)ZK_E_GENERIC;
    const uint32_t nf = pl.nf, nb = pl.nb, nseg = pl.nseg;
    const size_t frames_bytes = ((size_t)nf * sizeof(ZkEncFrame) + 63) & ~(size_t)63;
    const size_t blocks_bytes = ((size_t)(nb + 1) * sizeof(ZkEncBlock) + 63) & ~(size_t)63;
    const size_t doff_bytes = ((size_t)(nf + 1) * 8 + 63) & ~(size_t)63;
    const size_t segs_bytes = ((size_t)(nseg + 1) * sizeof(ZkEncFrame) + 63) & ~(size_t)63;
    int rc;
    if ((rc = zk_pin_reserve(e, frames_bytes + blocks_bytes + doff_bytes + sizeof(ZkEncTables) + 64 + segs_bytes))) return rc;
    ZkEncFrame *frames = (ZkEncFrame *)e->enc_pin;
    ZkEncBlock *blocks = (ZkEncBlock *)((uint8_t *)e->enc_pin + frames_bytes);
    uint64_t *doff = (uint64_t *)((uint8_t *)e->enc_pin + frames_bytes + blocks_bytes);
    ZkEncTables *htab = (ZkEncTables *)((uint8_t *)doff + doff_bytes);
    ZkEncFrame *segs = (ZkEncFrame *)((uint8_t *)e->enc_pin + ((frames_bytes + blocks_bytes + doff_bytes + sizeof(ZkEncTables) + 63) & ~(size_t)63));
    zke_plan_fill(n, frame_size, a.level, a.d_prefix ? a.prefix_len : 0, &pl, frames, blocks, segs, doff);
    const uint64_t seq_total = pl.seq_total, scratch_total = pl.scratch_total;
    if ((rc = zk_devbuf_reserve(e, e->enc_a, frames_bytes + blocks_bytes + 256))) return rc;
    if ((rc = zk_devbuf_reserve(e, e->enc_b, (size_t)(seq_total + 1) * 12 + 64))) return rc;       // packed sequences (u64) + match positions (u32)
    if ((rc = zk_devbuf_reserve(e, e->enc_c, (size_t)n + 64))) return rc;
    if ((rc = zk_devbuf_reserve(e, e->enc_d, (size_t)scratch_total + 64))) return rc;
    if ((rc = zk_devbuf_reserve(e, e->enc_e, (size_t)(nf + 1) * 8 * 4 + 64 + sizeof(ZkEncTables)))) return rc;   // c_size64, out_off, hashes, d_off, predefined tables
    if ((rc = zk_devbuf_reserve(e, e->enc_f, (size_t)nf * sizeof(ZkEncTables) + 64))) return rc;                  // the frames' tables
    ZkEncFrame *dfr = (ZkEncFrame *)e->enc_a.p;
    ZkEncBlock *dbl = (ZkEncBlock *)((uint8_t *)e->enc_a.p + frames_bytes);
    uint64_t *c64 = (uint64_t *)e->enc_e.p, *out_off = c64 + (nf + 1), *hashes = out_off + (nf + 1), *d_doff = hashes + (nf + 1);
    ZkEncTables *dtab = (ZkEncTables *)(d_doff + (nf + 1));
    if (!e->enc_tables_ready) { zk_build_enc_tables(&e->enc_tables); e->enc_tables_ready = true; }
    *htab = e->enc_tables;
    ZK_HIP(hipMemcpyAsync(dfr, frames, frames_bytes + (size_t)nb * sizeof(ZkEncBlock), hipMemcpyHostToDevice, st));   // frames + blocks are contiguous
    ZK_HIP(hipMemcpyAsync(d_doff, doff, (size_t)(nf + 1) * 8, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(dtab, htab, sizeof(ZkEncTables), hipMemcpyHostToDevice, st));
    zk_profile_begin(e);
    const uint8_t *src = (const uint8_t *)a.d_src;
    const uint8_t *msrc = src;                               // what the matcher reads
    if (hist) {
        if ((rc = zk_devbuf_reserve(e, e->enc_hist, (size_t)nf * ((size_t)hist + frame_size) + 64))) return rc;
        zk_launch_enc_stage_hist(st, src, (const uint8_t *)a.d_prefix + (a.prefix_len - hist), dfr, nf, (uint8_t *)e->enc_hist.p);
        msrc = (const uint8_t *)e->enc_hist.p;
    }
    // a prefix the ring cannot hold: its sampled positions enter a table in HBM (rebuilt per call: the bytes are the caller's)
    ZkEncLdm ldm = {nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, nullptr, 0, 0};
    // the matcher's workgroups are launched over the segments (one per <= 256 KiB of a frame)
    if ((rc = zk_devbuf_reserve(e, e->enc_seg, segs_bytes + 64))) return rc;
    ZK_HIP(hipMemcpyAsync(e->enc_seg.p, segs, (size_t)nseg * sizeof(ZkEncFrame), hipMemcpyHostToDevice, st));
    // (a prefix of one to three bytes leaves no history -- hist is a multiple of 4 -- but it is a prefix: the frames' windows were planned without
    //  far history, as the twin does; found by reading, round 6: such a frame would have carried offsets beyond its window)
    const uint64_t plen_eff = a.d_prefix ? a.prefix_len : 0;
    uint64_t dense_slice = 4ull << 30;
    if (const char *v = getenv("ZK_DENSE_SLICE_BYTES")) { const unsigned long long x = strtoull(v, nullptr, 10); if (x) dense_slice = x; }
    size_t dense_span = 0;
    if (!hist && zke_ldm_in_frame(a.level, plen_eff, frame_size < a.n ? frame_size : a.n)) {
        // in-frame far history (level >= 2, no prefix, frames beyond the ring's reach): one table per frame over its own bytes
        ldm.inframe = 1; ldm.frame_size = frame_size; ldm.n_total = a.n; ldm.log = zke_ldm_log(frame_size < a.n ? frame_size : a.n);
        if ((rc = zk_devbuf_reserve(e, e->enc_ldm, (((size_t)nf * sizeof(uint32_t)) << ldm.log) + 64))) return rc;
        if (zk_launch_enc_ldm_build_frames(st, src, ldm, (uint32_t *)e->enc_ldm.p, nf)) { e->last_err = "hipMemsetAsync (in-frame long-distance table)"; return ZK_ERR_HIP; }
        ldm.table = (const uint32_t *)e->enc_ldm.p;
        if (zke_dense_in_frame(a.level, plen_eff, frame_size < a.n ? frame_size : a.n)) {
            // dense far history (level 0 / >= 3): a far candidate per input byte, 4 bytes each (with the sorted positions 8 x the input: HBM is what this device has)
            ldm.dlog = zke_dense_log(a.level);
            // + as much again for the positions sorted by the pass their slot belongs to, and a word per segment and pass (+ 1)
            // -- for a SLICE of whole frames at a time (at most dense_slice bytes of input: 4 GiB unless ZK_DENSE_SLICE_BYTES says otherwise, which
            // the tests use): the dense kernels and the match kernel run slice after slice over the same scratch, so a call of any size takes at
            // most 8 x 4 GiB of it
            dense_span = (size_t)(n < dense_slice ? n : (dense_slice / frame_size ? dense_slice / frame_size : 1) * (uint64_t)frame_size);
            if ((rc = zk_devbuf_reserve(e, e->enc_dense, (2 * (dense_span + ZKE_DENSE_SLACK) + (size_t)nseg * (ZKE_DENSE_PASSES_MAX + 1)) * sizeof(uint32_t) + 64))) return rc;
            ldm.dense = (const uint32_t *)e->enc_dense.p;
        }
    }
    if (hist && a.prefix_len > ZKE_WINDOW) {
        const uint64_t usable = zke_ldm_usable(a.prefix_len);
        ldm.pfx = (const uint8_t *)a.d_prefix; ldm.plen = a.prefix_len; ldm.u0 = a.prefix_len - usable; ldm.log = zke_ldm_log(usable);
        if ((rc = zk_devbuf_reserve(e, e->enc_ldm, (sizeof(uint32_t) << ldm.log) + 64))) return rc;
        if (zk_launch_enc_ldm_build(st, ldm, (uint32_t *)e->enc_ldm.p)) { e->last_err = "hipMemsetAsync (long-distance table)"; return ZK_ERR_HIP; }
        ldm.table = (const uint32_t *)e->enc_ldm.p;
    }
    if (ldm.dense && dense_span < n) {
        // slice after slice (whole frames): the candidate arrays are indexed like the source, so a slice's kernels get them shifted by
        // the slice's first byte; the timer of the match kernel covers the dense kernels of such a call too
        zk_kernel_timer t(e, ZK_K_ENC_MATCH, st);
        uint32_t *dc = (uint32_t *)e->enc_dense.p, *dp = dc + (dense_span + ZKE_DENSE_SLACK), *po = dp + (dense_span + ZKE_DENSE_SLACK);
        const uint32_t fpf = (uint32_t)(dense_span / frame_size), spf = (frame_size + ZKE_SEGMENT - 1) / ZKE_SEGMENT;   // frames per slice, segments per whole frame
        for (uint32_t f0 = 0; f0 < nf; f0 += fpf) {
            const uint32_t f1 = nf - f0 < fpf ? nf : f0 + fpf, s0 = f0 * spf, s1 = f1 == nf ? nseg : f1 * spf;
            const uint64_t lo = (uint64_t)f0 * frame_size;
            ZkEncLdm sl = ldm;
            sl.dense = dc - lo;
            const ZkEncFrame *sg = (const ZkEncFrame *)e->enc_seg.p + s0;
            zk_launch_enc_dense_cand(st, src, sg, s1 - s0, sl, dc - lo, dp - lo, po);
            zk_launch_enc_match(st, msrc, sg, s1 - s0, dbl, (uint64_t *)e->enc_b.p, (uint8_t *)e->enc_c.p, a.level, sl);
        }
    } else {
        if (ldm.dense) { zk_kernel_timer t(e, ZK_K_ENC_DENSE, st); uint32_t *dc = (uint32_t *)e->enc_dense.p, *dp = dc + (dense_span + ZKE_DENSE_SLACK);
                         zk_launch_enc_dense_cand(st, src, (const ZkEncFrame *)e->enc_seg.p, nseg, ldm, dc, dp, dp + (dense_span + ZKE_DENSE_SLACK)); }
        { zk_kernel_timer t(e, ZK_K_ENC_MATCH, st); zk_launch_enc_match(st, msrc, (const ZkEncFrame *)e->enc_seg.p, nseg, dbl, (uint64_t *)e->enc_b.p, (uint8_t *)e->enc_c.p, a.level, ldm); }
    }
    ZkEncTables *ftab = (ZkEncTables *)e->enc_f.p;
    { zk_kernel_timer t(e, ZK_K_ENC_FSE_BUILD, st); zk_launch_enc_fse_build(st, src, dfr, nf, dbl, (uint64_t *)e->enc_b.p, (uint32_t *)((uint64_t *)e->enc_b.p + seq_total + 1), dtab, ftab); }
    bool cks_beside = false;                                 // the checksums run on the second queue beside the entropy stage (which waits on its own chains; beside the matcher they
                                                             // cost it 2 ms of vector issue slots) and are joined before the assembly
    if (a.checksum) {
        if (!e->profiling && !e->enc_aux) {
            if (hipStreamCreateWithFlags(&e->enc_aux, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->enc_ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->enc_ev_join, hipEventDisableTiming) != hipSuccess) { e->enc_aux = nullptr; (void)hipGetLastError(); }
        }
        if (!e->profiling && e->enc_aux) {
            ZK_HIP(hipEventRecord(e->enc_ev_fork, st));
            ZK_HIP(hipStreamWaitEvent(e->enc_aux, e->enc_ev_fork, 0));
            zk_launch_xxh64(e->enc_aux, src, d_doff, 0, nf, nullptr, hashes, e->choice, nullptr, 512, true);
            ZK_HIP(hipEventRecord(e->enc_ev_join, e->enc_aux));
            cks_beside = true;
        } else { zk_kernel_timer t(e, ZK_K_ENC_XXH64, st); zk_launch_xxh64(st, src, d_doff, 0, nf, nullptr, hashes, e->choice, nullptr, 512, true); }
    }
    { zk_kernel_timer t(e, ZK_K_ENC_ENTROPY, st); zk_launch_enc_entropy(st, src, dfr, dbl, nb, (uint64_t *)e->enc_b.p, (uint32_t *)((uint64_t *)e->enc_b.p + seq_total + 1), (const uint8_t *)e->enc_c.p, (uint8_t *)e->enc_d.p, ftab); }
    zk_launch_enc_sizes(st, dfr, nf, dbl, ftab, a.checksum, c64, (uint32_t *)a.d_c_sizes, (uint32_t *)a.d_d_sizes);
    zk_launch_scan64(st, c64, nf, out_off);
    ZK_HIP(hipMemcpyAsync(e->h_words + ZK_HW_ENC_TOTAL, out_off + nf, 8, hipMemcpyDeviceToHost, st));
    if (a.dst_cap < zk_compress_bound(n, frame_size)) {      // the frames may not fit: the total decides before anything is written
        ZK_HIP(hipStreamSynchronize(st));
        if (e->h_words[ZK_HW_ENC_TOTAL] > a.dst_cap) return -(int)ZK_E_DST_TOO_SMALL;
    }
    if (cks_beside) ZK_HIP(hipStreamWaitEvent(st, e->enc_ev_join, 0));
    { zk_kernel_timer t(e, ZK_K_ENC_COMPACT, st); zk_launch_enc_assemble(st, src, dfr, nf, dbl, nb, ftab, (const uint8_t *)e->enc_c.p, (const uint8_t *)e->enc_d.p, out_off, c64, hashes, a.checksum, (uint8_t *)a.d_dst); }
    if (nf_out) *nf_out = nf;
    return 0;
}

int zk_encode_finish(zk_engine *e, hipStream_t st, uint64_t *written_out)
{
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    zk_profile_collect(e);
    if (written_out) *written_out = e->h_words[ZK_HW_ENC_TOTAL];
    return 0;
}

extern "C" int zk_encode_frames_prefix_dev(zk_engine *e, const void *d_src, uint64_t n, uint32_t frame_size, int level, int checksum,
                                           const void *d_prefix, uint64_t prefix_len, void *d_dst, uint64_t dst_cap, void *d_c_sizes,
                                           void *d_d_sizes, uint32_t *n_frames_out, uint64_t *written_out, void *stream)
{
    if (!e) return ZK_ERR_ARGUMENT;
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    zk_enc_args a{d_src, n, frame_size, level, checksum, d_prefix, prefix_len, d_dst, dst_cap, d_c_sizes, d_d_sizes};
    uint32_t nf = 0;
    int rc = zk_encode_enqueue(e, a, st, &nf);
    if (rc) return rc;
    uint64_t total = 0;
    if ((rc = zk_encode_finish(e, st, &total))) return rc;
    if (n_frames_out) *n_frames_out = nf;
    if (written_out) *written_out = total;
    return 0;
}
