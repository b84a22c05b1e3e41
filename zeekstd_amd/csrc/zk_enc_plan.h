// zk_enc_plan.h -- host arithmetic of one encode: where the frames, their blocks and the matcher's segments lie.
// Frame boundaries are the policy's (lib/src/encode.rs:528-544, FrameSizePolicy::Uncompressed: frame i covers input bytes
// [i * L, min((i + 1) * L, n))); blocks and segments are this engine's own cut (zk_enc_device.h).  Used by the engine
// (zk_engine_enc.hip) and by the CPU emulation of the match kernel (tests/sim/zk_enc_sim.cpp).
#pragma once
#include <stdint.h>
#include <string.h>
#include "zk_enc_device.h"

struct ZkEncPlan {
    uint32_t nf, nb, nseg;                  // frames, blocks, matcher segments
    uint32_t hist;                          // bytes of prefix laid out before every frame in the matcher's source (a multiple of 4)
    uint64_t seq_total, scratch_total;      // entries of the sequence buffer, bytes of the entropy stage's scratch
};

// bytes of a prefix the matcher can reach: its window, rounded down to a multiple of 4 (a lane takes four positions out of
// aligned ring words, so every record starts on a word)
inline uint32_t zke_prefix_hist(uint64_t prefix_len) { return (uint32_t)(prefix_len < ZKE_WINDOW ? prefix_len : ZKE_WINDOW) & ~3u; }

// counts only; false: too many blocks / segments for 32-bit indices
inline bool zke_plan_count(uint64_t n, uint32_t frame_size, uint32_t hist, ZkEncPlan *pl)
{
    const uint64_t nf64 = n == 0 ? 1 : (n + frame_size - 1) / frame_size;
    uint64_t nb64 = 0, nseg64 = 0;
    // all frames but the last have the same size
    const uint64_t last = n - (nf64 - 1) * (uint64_t)frame_size;
    for (int w = 0; w < 2; w++) {
        const uint64_t cnt = w == 0 ? nf64 - 1 : 1, dsz = w == 0 ? frame_size : last;
        if (!cnt || !dsz) continue;
        const uint32_t bm = zke_block_max((uint32_t)dsz, hist != 0);
        nb64 += cnt * ((dsz + bm - 1) / bm);
        nseg64 += cnt * ((dsz + ZKE_SEGMENT - 1) / ZKE_SEGMENT);
    }
    if (nb64 > 0xFFFFFFF0ull || nseg64 > 0xFFFFFFF0ull || nf64 > 0xFFFFFFF0ull) return false;
    pl->nf = (uint32_t)nf64; pl->nb = (uint32_t)nb64; pl->nseg = (uint32_t)nseg64; pl->hist = hist;
    pl->seq_total = 0; pl->scratch_total = 0;
    return true;
}

// frames[nf], blocks[nb], segs[nseg] (a segment record is a ZkEncFrame that covers <= ZKE_SEGMENT bytes of its frame:
// d_size / n_blocks / block_base are the segment's, hist / m_off its history), doff[nf + 1] (may be null)
// prefix_len: the whole prefix (0: none); above ZKE_WINDOW the matcher also finds long-distance matches into it
// (ZkEncLdm) and the frames' windows cover prefix + frame
inline void zke_plan_fill(uint64_t n, uint32_t frame_size, int level, uint64_t prefix_len, ZkEncPlan *pl, ZkEncFrame *frames, ZkEncBlock *blocks, ZkEncFrame *segs, uint64_t *doff)
{
    const uint32_t hist = pl->hist;
    uint64_t seq_total = 0, scratch_total = 0;
    uint32_t bcount = 0, sc = 0;
    for (uint32_t f = 0; f < pl->nf; f++) {
        ZkEncFrame &fr = frames[f];
        fr.src_off = (uint64_t)f * frame_size;
        fr.d_size = (uint32_t)(n - fr.src_off < frame_size ? n - fr.src_off : frame_size);
        uint32_t wlog = 10;
        while ((1u << wlog) < fr.d_size && wlog < 17) wlog++;
        if (hist) wlog = 17;                                 // covers every offset the matcher can produce, into the prefix too
        if (hist && prefix_len > ZKE_WINDOW) while ((1ull << wlog) < prefix_len + fr.d_size && wlog < 27) wlog++;   // long-distance offsets: < 2^27
        if (zke_ldm_in_frame(level, prefix_len, fr.d_size)) while ((1ull << wlog) < fr.d_size && wlog < 27) wlog++;     // in-frame far history: the window covers the frame
        fr.window_log = wlog;
        fr.block_max = zke_block_max(fr.d_size, hist != 0);
        fr.n_blocks = fr.d_size ? (fr.d_size + fr.block_max - 1) / fr.block_max : 0;
        fr.block_base = bcount;
        fr.hist = fr.d_size ? hist : 0;
        fr.m_off = hist ? (uint64_t)f * ((uint64_t)hist + frame_size) : fr.src_off;
        fr.minmatch = zke_minmatch2(level, prefix_len); fr.seg_at = 0;
        for (uint32_t b = 0; b < fr.n_blocks; b++) {
            ZkEncBlock &k = blocks[bcount++];
            memset(&k, 0, sizeof k);
            k.frame = f; k.bs = b * fr.block_max;
            k.bsz = fr.d_size - k.bs < fr.block_max ? fr.d_size - k.bs : fr.block_max;
            k.seq_base = seq_total; seq_total += k.bsz / 4 + 2;
            k.lit_base = fr.src_off + k.bs;
            k.scratch_base = scratch_total;
            const uint32_t q = (k.bsz + 3) / 4;
            scratch_total += ZKE_SMALL + 4ull * (q + (q >> 1) + 16) + (uint64_t)k.bsz + 64;   // small parts, 4 literal streams, the sequence bitstream + its slack
        }
        if (doff) doff[f] = fr.src_off;
        const uint32_t per = ZKE_SEGMENT / fr.block_max;      // blocks per segment (frames above ZKE_SEGMENT have 16 or 32 KiB blocks)
        for (uint32_t at = 0; at < fr.d_size; at += ZKE_SEGMENT) {
            ZkEncFrame &sg = segs[sc++];
            sg = fr;
            sg.d_size = fr.d_size - at < ZKE_SEGMENT ? fr.d_size - at : ZKE_SEGMENT;
            sg.n_blocks = (sg.d_size + fr.block_max - 1) / fr.block_max;
            sg.block_base = fr.block_base + (at / ZKE_SEGMENT) * per;
            sg.seg_at = at;
            if (at) { sg.hist = ZKE_WINDOW; sg.m_off = fr.m_off + fr.hist + at - ZKE_WINDOW; }
        }
    }
    if (doff) doff[pl->nf] = n;
    pl->seq_total = seq_total; pl->scratch_total = scratch_total;
}
