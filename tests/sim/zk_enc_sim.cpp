// zk_enc_sim.cpp -- TEST HARNESS (never shipped, never linked into libzeekstd_amd.so).
// Runs the encoder's match + parse kernel (zeekstd_amd/csrc/zk_enc_match.h, the very source hipcc compiles for gfx950)
// on the CPU, one workgroup at a time, under the fiber emulator of hip_wg_emu.h, with the frame / block / segment plan the
// engine uses (zk_enc_plan.h).  The -m "not gpu" suite compares its sequences and literals with the CPU twin
// (oracle/zstd_oracle_enc.c) block by block -- the kernel's logic is checked before a GPU is involved.
#include "hip_wg_emu.h"
#include "../../zeekstd_amd/csrc/zk_enc_match.h"
#include "../../zeekstd_amd/csrc/zk_enc_match2.h"
#include "../../zeekstd_amd/csrc/zk_enc_plan.h"

// Encode plan + match kernel over src[0, n) cut into frames of frame_size bytes; prefix (may be null) is what
// zk_encode_frames_prefix references.  Outputs, per block b (in plan order): blk_nseq[b], blk_nlit[b], blk_bsz[b]; the
// block's sequences packed at seqs + seq_at[b] (ll | ml << 20 | Offset_Value << 40), its literals at lits + lit_at[b].
// Returns the number of blocks, or -1 when an output array is too small.
extern "C" int zk_enc_sim_match(const uint8_t *src, uint64_t n, uint32_t frame_size, int level, const uint8_t *prefix, uint64_t prefix_len,
                                uint32_t blk_cap, uint32_t *blk_nseq, uint32_t *blk_nlit, uint32_t *blk_bsz, uint64_t *seq_at, uint64_t *lit_at,
                                uint64_t *seqs, uint64_t seq_cap, uint8_t *lits, uint64_t lit_cap)
{
    const uint32_t hist = prefix ? zke_prefix_hist(prefix_len) : 0;
    ZkEncPlan pl;
    if (!zke_plan_count(n, frame_size, hist, &pl)) return -1;
    if (pl.nb > blk_cap) return -1;
    std::vector<ZkEncFrame> frames(pl.nf), segs(pl.nseg + 1);
    std::vector<ZkEncBlock> blocks(pl.nb + 1);
    zke_plan_fill(n, frame_size, level, prefix ? prefix_len : 0, &pl, frames.data(), blocks.data(), segs.data(), nullptr);
    if (pl.seq_total > seq_cap || n > lit_cap) return -1;
    // the matcher's source: the frames in place, or [prefix tail | frame] records (zk_k_enc_stage_hist)
    std::vector<uint8_t> stage;
    const uint8_t *msrc = src;
    if (hist) {
        stage.resize((size_t)pl.nf * ((size_t)hist + frame_size) + 64);
        for (uint32_t f = 0; f < pl.nf; f++) {
            memcpy(stage.data() + frames[f].m_off, prefix + (prefix_len - hist), hist);
            memcpy(stage.data() + frames[f].m_off + hist, src + frames[f].src_off, frames[f].d_size);
        }
        msrc = stage.data();
    }
    // long-distance table over the prefix (the engine builds it with zk_k_enc_ldm_build; here sequentially, same rule)
    ZkEncLdm ldm = {nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> table, dense;
    std::vector<uint8_t> pcopy;
    if (!prefix && zke_ldm_in_frame(level, 0, frame_size < n ? frame_size : n)) {
        // in-frame far history: one table per frame over its own bytes (the engine: zk_k_enc_ldm_build_frames; here sequentially, same rule)
        ldm.inframe = 1; ldm.frame_size = frame_size; ldm.n_total = n; ldm.log = zke_ldm_log(frame_size < n ? frame_size : n);
        table.assign((size_t)pl.nf << ldm.log, ZKE_LDM_NONE);
        for (uint32_t f = 0; f < pl.nf; f++) {
            const uint64_t at = (uint64_t)f * frame_size, fsz = n - at < frame_size ? n - at : frame_size;
            const uint32_t log = zke_ldm_log(fsz);
            for (uint64_t i = 0; i + ZKE_LDM_MIN <= fsz; i++) {
                uint32_t w[4]; memcpy(w, src + at + i, 16);
                const uint32_t h = zke_ldm_hash(w[0], w[1], w[2], w[3]);
                if (!zke_ldm_selected(h)) continue;
                uint32_t &slot = table[((size_t)f << ldm.log) + zke_ldm_slot(h, log)];
                if ((uint32_t)i < slot) slot = (uint32_t)i;
            }
        }
        ldm.table = table.data();
        if (zke_dense_in_frame(level, 0, frame_size < n ? frame_size : n)) {
            // dense far history: every position's far candidate out of the first / last occurrence per slot and segment (the engine:
            // zk_k_enc_dense_cand; here sequentially, same rule)
            ldm.dlog = zke_dense_log(level);
            dense.assign((size_t)n + ZKE_DENSE_SLACK, 0);
            const size_t slots = (size_t)1 << ldm.dlog;
            std::vector<uint32_t> first(slots), last(slots), prev(slots);
            auto h5 = [&](const uint8_t *p) { uint32_t lo; memcpy(&lo, p, 4); return zke_hash(lo, p[4], ldm.dlog); };
            for (uint32_t g = 0; g < pl.nseg; g++) {
                const ZkEncFrame &sg = segs[g];
                const uint8_t *frame = src + sg.src_off;
                const uint64_t fsz = n - sg.src_off < frame_size ? n - sg.src_off : frame_size;
                prev = last;                                                   // (a segment behind a frame's first follows it in the list)
                std::fill(first.begin(), first.end(), ZKE_DENSE_NONE); std::fill(last.begin(), last.end(), 0u);
                const uint64_t e1 = (uint64_t)sg.seg_at + sg.d_size;
                for (uint64_t q = sg.seg_at; q < e1 && q + 8 <= fsz; q++) {
                    if (frame[q + 1] == frame[q] && frame[q + 2] == frame[q] && frame[q + 3] == frame[q]) continue;      // a byte run: neither entered nor looked up
                    const uint32_t h = h5(frame + q), r = (uint32_t)(q - sg.seg_at);
                    if (r < first[h]) first[h] = r;
                    if (r + 1 > last[h]) last[h] = r + 1;
                }
                for (uint64_t q = sg.seg_at; q < e1 && q + 8 <= fsz; q++) {
                    const uint32_t h = h5(frame + q), r = (uint32_t)(q - sg.seg_at), m1 = first[h], m2 = sg.seg_at ? prev[h] : 0u;
                    uint32_t d = 0;
                    const uint8_t *b = frame + q;
                    if (b[1] == b[0] && b[2] == b[0] && b[3] == b[0]) d = 0;      // a byte run
                    else if (m1 != ZKE_DENSE_NONE && m1 < r && r - m1 > ZKE_WINDOW) d = r - m1;
                    else if (m2 && r + ZKE_SEGMENT - (m2 - 1) > ZKE_WINDOW) d = r + ZKE_SEGMENT - (m2 - 1);
                    if (!d) continue;
                    uint32_t l = 0;
                    while (l < 16 && q + l < fsz && frame[q + l] == frame[q + l - d]) l++;
                    while (l < 16 && q + l >= fsz && frame[q + l - d] == 0) l++;     // the kernel reads zeros behind the frame's last byte (the match kernel caps the length at its tile's end)
                    uint32_t bk = 0;
                    while (bk < 4 && q - bk > d && frame[q - bk - 1] == frame[q - bk - 1 - d]) bk++;      // not past the frame's first byte
                    if (l >= ZKE_DENSE_MIN) dense[sg.src_off + q] = l | (bk << 5) | (d << 8);
                }
            }
            ldm.dense = dense.data();
        }
    }
    if (prefix && prefix_len > ZKE_WINDOW) {
        const uint64_t usable = zke_ldm_usable(prefix_len);
        ldm.plen = prefix_len; ldm.u0 = prefix_len - usable; ldm.log = zke_ldm_log(usable);
        pcopy.assign(16 + usable + ZKE_LDM_SLACK, 0);                 // what the engine keeps on the device: slack | prefix[u0, plen) | slack
        memcpy(pcopy.data() + 16, prefix + ldm.u0, usable);
        ldm.pfx = pcopy.data() + 16 - ldm.u0;
        table.assign((size_t)1 << ldm.log, ZKE_LDM_NONE);
        for (uint64_t i = 0; i + ZKE_LDM_MIN <= usable; i++) {
            uint32_t w[4]; memcpy(w, pcopy.data() + 16 + i, 16);
            const uint32_t h = zke_ldm_hash(w[0], w[1], w[2], w[3]);
            if (!zke_ldm_selected(h)) continue;
            uint32_t &slot = table[zke_ldm_slot(h, ldm.log)];
            if ((uint32_t)i < slot) slot = (uint32_t)i;
        }
        ldm.table = table.data();
    }
    for (uint32_t s = 0; s < pl.nseg; s++) {
        auto run = [&]() {
            if (ldm.dense) {
                if (zke_step(level) == 1024) zk_k_enc_match<15, 1, 1024, true, true>(msrc, segs.data(), blocks.data(), seqs, lits, ldm);
                else zk_k_enc_match<15, 1, 4096, true, true>(msrc, segs.data(), blocks.data(), seqs, lits, ldm);
            }
            else if (ldm.table) {
                if (zke_fast(level)) zk_k_enc_match<14, 0, 4096, true>(msrc, segs.data(), blocks.data(), seqs, lits, ldm);
                else if (zke_step(level) == 1024) zk_k_enc_match<15, 1, 1024, true>(msrc, segs.data(), blocks.data(), seqs, lits, ldm);
                else zk_k_enc_match<15, 1, 4096, true>(msrc, segs.data(), blocks.data(), seqs, lits, ldm);
            }
            else if (zke_fast(level)) zk_k_enc_match2<14>(msrc, segs.data(), blocks.data(), seqs, lits);
            else if (zke_step(level) == 1024) zk_k_enc_match<15, 1, 1024, false>(msrc, segs.data(), blocks.data(), seqs, lits, ldm);
            else zk_k_enc_match<15, 1, 4096, false>(msrc, segs.data(), blocks.data(), seqs, lits, ldm);
        };
        emu_run_workgroup(ZKE_THREADS, s, run);
    }
    for (uint32_t b = 0; b < pl.nb; b++) {
        blk_nseq[b] = blocks[b].nseq; blk_nlit[b] = blocks[b].nlit; blk_bsz[b] = blocks[b].bsz;
        seq_at[b] = blocks[b].seq_base; lit_at[b] = blocks[b].lit_base;
    }
    return (int)pl.nb;
}

// The entropy stage's Huffman build as the kernel does it (zk_k_enc_entropy): the symbols ranked by (count, symbol) beforehand
// (there: by all lanes of the wave) and handed to zke_huf_lengths, the canonical codes from per-weight counts and ranks (there:
// ballots) -- against the plain serial functions zke_huf_lengths / zke_huf_codes the twin's restatement follows.
// cnt[nsym] (nsym <= 128); returns 0 when lengths, depth and codes agree, a positive code for the first disagreement.
extern "C" int zk_enc_sim_huf(const uint32_t *cnt, int nsym)
{
    uint32_t c0[256] = {0}, c1[256] = {0};
    for (int s = 0; s < nsym; s++) c0[s] = c1[s] = cnt[s];
    static ZkHufBuild h0, h1;
    ZkHufCode a, b;
    memset(&a, 0xEE, sizeof a); memset(&b, 0xEE, sizeof b);                 // stale bytes beyond nsym, as in LDS
    const int d0 = zke_huf_lengths(c0, nsym, &h0, a.len);
    // the ranks: key = count << 8 | symbol, rank = keys below mine among the symbols that occur
    int m = 0;
    for (int s = 0; s < nsym; s++) {
        if (!cnt[s]) continue;
        const uint32_t key = (cnt[s] << 8) | (uint32_t)s;
        int r = 0;
        for (int t = 0; t < nsym; t++) if (cnt[t] && (((cnt[t] << 8) | (uint32_t)t) < key)) r++;
        h1.idx[r] = (uint8_t)s; m++;
    }
    const int d1 = zke_huf_lengths(c1, nsym, &h1, b.len, m);
    if (d0 != d1) return 1;
    if (d0 <= 0) return 0;
    for (int s = 0; s < nsym; s++) if (a.len[s] != b.len[s]) return 2;
    zke_huf_codes(&a, nsym, d0);
    // codes by weight: weight 1 first, symbols ascending; 64 "lanes" hold symbols lane and lane + 64
    uint32_t pos = 0;
    for (int wt = 1; wt <= d1; wt++) {
        int n0 = 0, n1 = 0;
        for (int s = 0; s < 64; s++) { const uint32_t l = s < nsym ? b.len[s] : 0; if (l && d1 + 1 - (int)l == wt) n0++; }
        for (int s = 64; s < 128; s++) { const uint32_t l = s < nsym ? b.len[s] : 0; if (l && d1 + 1 - (int)l == wt) n1++; }
        int r0 = 0, r1 = 0;
        for (int s = 0; s < 128; s++) {
            const uint32_t l = s < nsym ? b.len[s] : 0;
            if (!(l && d1 + 1 - (int)l == wt)) continue;
            if (s < 64) b.code[s] = (uint16_t)((pos >> (wt - 1)) + (uint32_t)r0++);
            else b.code[s] = (uint16_t)((pos >> (wt - 1)) + (uint32_t)n0 + (uint32_t)r1++);
        }
        pos += (uint32_t)(n0 + n1) << (wt - 1);
    }
    for (int s = 0; s < nsym; s++) if (a.len[s] && a.code[s] != b.code[s]) return 3;
    return 0;
}
