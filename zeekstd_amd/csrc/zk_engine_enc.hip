// zk_engine_enc.hip -- encode half of the batch engine (Level A of include/zeekstd_amd.h).
#include <hip/hip_runtime.h>
#include <string.h>
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
    // the matcher sees the last `hist` bytes of the prefix right before every frame (its window is 64 KiB)
    const uint32_t hist = a.d_prefix ? (uint32_t)(a.prefix_len < ZKE_WINDOW ? a.prefix_len : ZKE_WINDOW) : 0;
    if (!e || frame_size == 0 || frame_size > ZK_SEEKABLE_MAX_FRAME_SIZE || !a.d_dst || (n && !a.d_src)) return ZK_ERR_ARGUMENT;
    const uint64_t nf64 = n == 0 ? 1 : (n + frame_size - 1) / frame_size;
    if (nf64 > ZK_SEEKABLE_MAX_FRAMES) return ZK_ERR_FRAME_INDEX_TOO_LARGE;
    const uint32_t nf = (uint32_t)nf64;
    ZK_HIP(hipSetDevice(e->device));

    // frame / block lists (host arithmetic only: frame boundaries are the policy's, encode.rs:528-544)
    uint64_t nb64 = 0;
    for (uint32_t f = 0; f < nf; f++) {
        const uint64_t so = (uint64_t)f * frame_size;
        const uint32_t dsz = (uint32_t)(n - so < frame_size ? n - so : frame_size);
        uint32_t bm = zke_block_max(dsz, hist != 0);
        nb64 += dsz ? (dsz + bm - 1) / bm : 0;
    }
    if (nb64 > 0xFFFFFFF0ull) return -(int)ZK_E_GENERIC;
    const uint32_t nb = (uint32_t)nb64;
    const size_t frames_bytes = ((size_t)nf * sizeof(ZkEncFrame) + 63) & ~(size_t)63;
    const size_t blocks_bytes = ((size_t)(nb + 1) * sizeof(ZkEncBlock) + 63) & ~(size_t)63;
    const size_t doff_bytes = ((size_t)(nf + 1) * 8 + 63) & ~(size_t)63;
    // frames above ZKE_SEGMENT: the matcher runs one workgroup per segment (zk_enc_device.h)
    const bool segmented = frame_size > ZKE_SEGMENT;
    uint64_t nseg64 = 0;
    if (segmented) for (uint32_t f = 0; f < nf; f++) {
        const uint64_t so = (uint64_t)f * frame_size;
        const uint64_t dsz = n - so < frame_size ? n - so : frame_size;
        nseg64 += (dsz + ZKE_SEGMENT - 1) / ZKE_SEGMENT;
    }
    if (nseg64 > 0xFFFFFFF0ull) return -(int)ZK_E_GENERIC;
    const uint32_t nseg = (uint32_t)nseg64;
    const size_t segs_bytes = ((size_t)nseg * sizeof(ZkEncFrame) + 63) & ~(size_t)63;
    int rc;
    if ((rc = zk_pin_reserve(e, frames_bytes + blocks_bytes + doff_bytes + sizeof(ZkEncTables) + 64 + segs_bytes))) return rc;
    ZkEncFrame *frames = (ZkEncFrame *)e->enc_pin;
    ZkEncBlock *blocks = (ZkEncBlock *)((uint8_t *)e->enc_pin + frames_bytes);
    uint64_t *doff = (uint64_t *)((uint8_t *)e->enc_pin + frames_bytes + blocks_bytes);
    ZkEncTables *htab = (ZkEncTables *)((uint8_t *)doff + doff_bytes);
    ZkEncFrame *segs = (ZkEncFrame *)((uint8_t *)e->enc_pin + ((frames_bytes + blocks_bytes + doff_bytes + sizeof(ZkEncTables) + 63) & ~(size_t)63));
    uint64_t seq_total = 0, scratch_total = 0;
    uint32_t bcount = 0;
    for (uint32_t f = 0; f < nf; f++) {
        ZkEncFrame &fr = frames[f];
        fr.src_off = (uint64_t)f * frame_size;
        fr.d_size = (uint32_t)(n - fr.src_off < frame_size ? n - fr.src_off : frame_size);
        uint32_t wlog = 10;
        while ((1u << wlog) < fr.d_size && wlog < 17) wlog++;
        if (hist) wlog = 17;                                 // covers every offset the matcher can produce, into the prefix too
        fr.window_log = wlog;
        fr.block_max = zke_block_max(fr.d_size, hist != 0);
        fr.n_blocks = fr.d_size ? (fr.d_size + fr.block_max - 1) / fr.block_max : 0;
        fr.block_base = bcount;
        fr.hist = fr.d_size ? hist : 0;
        fr.m_off = hist ? (uint64_t)f * ((uint64_t)hist + frame_size) : fr.src_off;
        fr.minmatch = zke_minmatch(a.level); fr.pad = 0;
        for (uint32_t b = 0; b < fr.n_blocks; b++) {
            ZkEncBlock &k = blocks[bcount++];
            memset(&k, 0, sizeof k);
            k.frame = f; k.bs = b * fr.block_max;
            k.bsz = fr.d_size - k.bs < fr.block_max ? fr.d_size - k.bs : fr.block_max;
            k.seq_base = seq_total; seq_total += k.bsz / 4 + 2;
            k.lit_base = fr.src_off + k.bs;
            k.scratch_base = scratch_total;
            const uint32_t q = (k.bsz + 3) / 4;
            scratch_total += (uint64_t)k.bsz * 2 + 4ull * (q + (q >> 1) + 16) + 64;
        }
        doff[f] = fr.src_off;
    }
    doff[nf] = n;
    if (segmented) {
        uint32_t sc = 0;
        for (uint32_t f = 0; f < nf; f++) {
            const ZkEncFrame &fr = frames[f];
            const uint32_t per = ZKE_SEGMENT / fr.block_max;              // blocks per segment (block_max is 32 KiB for frames this large)
            for (uint32_t at = 0; at < fr.d_size; at += ZKE_SEGMENT) {
                ZkEncFrame &sg = segs[sc++];
                sg = fr;
                sg.d_size = fr.d_size - at < ZKE_SEGMENT ? fr.d_size - at : ZKE_SEGMENT;
                sg.n_blocks = (sg.d_size + fr.block_max - 1) / fr.block_max;
                sg.block_base = fr.block_base + (at / ZKE_SEGMENT) * per;
                if (at) { sg.hist = ZKE_WINDOW; sg.m_off = fr.m_off + fr.hist + at - ZKE_WINDOW; }
            }
        }
    }
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
    if (a.checksum) { zk_kernel_timer t(e, ZK_K_ENC_XXH64, st); zk_launch_xxh64(st, src, d_doff, 0, nf, nullptr, hashes); }
    const ZkEncFrame *mfr = dfr;                             // what the matcher's workgroups are launched over
    uint32_t nm = nf;
    if (segmented) {
        if ((rc = zk_devbuf_reserve(e, e->enc_seg, segs_bytes + 64))) return rc;
        ZK_HIP(hipMemcpyAsync(e->enc_seg.p, segs, (size_t)nseg * sizeof(ZkEncFrame), hipMemcpyHostToDevice, st));
        mfr = (const ZkEncFrame *)e->enc_seg.p; nm = nseg;
    }
    { zk_kernel_timer t(e, ZK_K_ENC_MATCH, st); zk_launch_enc_match(st, msrc, mfr, nm, dbl, (uint64_t *)e->enc_b.p, (uint32_t *)((uint64_t *)e->enc_b.p + seq_total + 1), (uint8_t *)e->enc_c.p, zke_hash_log(a.level)); }
    ZkEncTables *ftab = (ZkEncTables *)e->enc_f.p;
    { zk_kernel_timer t(e, ZK_K_ENC_FSE_BUILD, st); zk_launch_enc_fse_build(st, src, dfr, nf, dbl, (const uint64_t *)e->enc_b.p, dtab, ftab); }
    { zk_kernel_timer t(e, ZK_K_ENC_ENTROPY, st); zk_launch_enc_entropy(st, src, dfr, dbl, nb, (uint64_t *)e->enc_b.p, (uint32_t *)((uint64_t *)e->enc_b.p + seq_total + 1), (const uint8_t *)e->enc_c.p, (uint8_t *)e->enc_d.p, ftab); }
    zk_launch_enc_sizes(st, dfr, nf, dbl, ftab, a.checksum, c64, (uint32_t *)a.d_c_sizes, (uint32_t *)a.d_d_sizes);
    zk_launch_scan64(st, c64, nf, out_off);
    ZK_HIP(hipMemcpyAsync(e->h_words + ZK_HW_ENC_TOTAL, out_off + nf, 8, hipMemcpyDeviceToHost, st));
    if (a.dst_cap < zk_compress_bound(n, frame_size)) {      // the frames may not fit: the total decides before anything is written
        ZK_HIP(hipStreamSynchronize(st));
        if (e->h_words[ZK_HW_ENC_TOTAL] > a.dst_cap) return -(int)ZK_E_DST_TOO_SMALL;
    }
    { zk_kernel_timer t(e, ZK_K_ENC_COMPACT, st); zk_launch_enc_assemble(st, src, dfr, nf, dbl, nb, ftab, (const uint8_t *)e->enc_d.p, out_off, c64, hashes, a.checksum, (uint8_t *)a.d_dst); }
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
