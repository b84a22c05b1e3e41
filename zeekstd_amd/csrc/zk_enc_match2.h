// zk_enc_match2.h -- the match + parse kernel of the FAST setting (level <= 1 without a long-distance table: what BASELINE.json's
// configs encode with).  Round 5's rebuild of zk_enc_match.h's kernel for that setting; tests/sim/zk_enc_sim.cpp runs this source on
// the CPU under the workgroup emulator, oracle/zstd_oracle_enc.c (g_stride == 2) is its twin, sequence for sequence.
//
// Replaces the inside of ZSTD_compressStream2 that zeekstd drives frame by frame (lib/src/encode.rs:340-346).
//
// What changed against zk_k_enc_match, and why (VERDICT r4 "next" 1: the kernel is bound by its vector instruction count, 4.7 wave
// instructions per input byte, two thirds of the positions it compares are thrown away by the parse):
//   * candidates are looked up and compared at EVEN positions only -- a lane still owns four bytes of its tile and still inserts
//     all four positions into the table (a candidate may lie anywhere), but it reads two far and two near entries and runs eight
//     16-byte comparisons instead of sixteen;
//   * a match that the parse takes is caught up BACKWARDS by up to four bytes that agree at its offset (one more ring word for the
//     winner of a position): what a match loses by being seen one byte late -- and what the stale table lost all along, a match
//     whose first bytes found no entry: the ratio on the 8d text goes UP, 2.4715 -> 2.4846, with 5-byte table matches;
//   * the parse works on SLOTS of two bytes: a tile is two passes of 64 slots instead of four passes of 64 positions, so the
//     sweeps that cost the same per pass whatever a pass covers (masks, next-candidate table, emission) run half as often.
// Everything else -- ring, 32-bit table keys and the one-ds_min insertion, near candidates, tiles, seams, the stitch one group
// later, the record and literal stores -- is zk_enc_match.h's, whose helpers this file uses.
#pragma once
#include "zk_enc_match.h"

constexpr uint32_t ZKE2_SLOTS = ZKE_TILE / 2;            // candidate slots of a tile (even positions)
constexpr uint32_t ZKE2_BACK = 4;                        // catch-up bytes at most (one ring word in front of the candidate)

#ifndef ZKE_FFBH
// v_ffbh_u32: the number of leading zero bits, 0xFFFFFFFF for 0
__device__ __forceinline__ uint32_t zke_ffbh(uint32_t x) { uint32_t r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x)); return r; }
#define ZKE_FFBH(x) zke_ffbh(x)
#endif

// (Measured and dropped, round 5: gfx950 does unaligned ds_read_b32 / b64 / b128 correctly -- tools/ubench/ldsrd.hip -- and the compiler
//  emits them for packed types, one instruction instead of up to five reads and four v_alignbyte; but the comparisons went from 2.5 k to
//  7.5 k clocks per group and wave with them: an unaligned wide read costs the LDS several passes.  Aligned words + v_alignbyte stay.)
// The 20 bytes x[0 .. 4] are a lane's own bytes xor the bytes some distance before them: the number of zero bytes from byte 0 on and
// from byte 2 on (at most 16 each) = how far the match at that distance goes from the lane's two candidate positions.
__device__ __forceinline__ void zke_runs2(const uint32_t x[5], uint32_t &r0, uint32_t &r2)
{
    const uint32_t g = zke_first16(x[1], x[2], x[3], x[4]) + 4;           // first non-zero byte at or behind byte 4 (20: none)
    const uint32_t f0 = ZKE_FFBL(x[0]) >> 3, f2 = ZKE_FFBL(x[0] & 0xFFFF0000u) >> 3;   // in the first word, at or behind byte 0 / byte 2 (none: huge)
    const uint32_t n0 = f0 < g ? f0 : g, n2 = f2 < g ? f2 : g;
    r0 = n0 < ZKE_PARCAP ? n0 : ZKE_PARCAP;
    r2 = n2 - 2 < ZKE_PARCAP ? n2 - 2 : ZKE_PARCAP;
}

// The walk's inner loop: mark slot f, go to the slot its lane names, until that is no plain slot (64: the pass is done; bit 7: a
// match that needs the long extension).  Sixteen waves of a CU share ONE scalar unit and all of them walk at the same time: a step
// costs 16 x its scalar instructions (measured: 216 clocks per step with the compiler's 14), so the loop is written out -- four
// instructions and the wait state v_readlane needs between a scalar write of its lane select and itself.  f < 64 on entry.
#ifndef ZKE_DBG_SLOW
#define ZKE_DBG_SLOW() do { } while (0)     // experiments: how often the walk leaves its inner loop (tests/sim, -DZKE_DBG_COUNTS)
#define ZKE_DBG_TILE(n) do { } while (0)
#endif
#ifndef ZKE_KEEP
// keeps a value's computation where it is written (the compiler otherwise sinks it behind the condition that selects it: a branch)
#define ZKE_KEEP(x) asm volatile("" : "+v"(x))
#endif
// (Measured and dropped, round 6 -- VERDICT r5 "next" 2a: the walk off the scalar unit.  The chain's pointers are per-lane values, so its
//  powers are too: n2 = n1 o n1, n4, n8 by three rounds of ds_bpermute, beside each the 64-bit SET of slots the jump passes; the scalar
//  loop then takes 8 instructions per EIGHT steps (three v_readlane, two s_or, compare, wait state, branch) instead of 5 per step.  Same
//  sequences (emulator + twin), 81 GPU tests green -- and the same speed: 29.3 / 29.6 ms against 29.6 (tables per pass / for both passes
//  at once); phase clocks: the walk 1 780 / 1 690 clocks per group and wave (2 600 before), the sweep in front of it 1 350 (600): the 14
//  ds_bpermute + ~60 vector instructions of the tables cost what the loop saved.  Every unit is near one instruction per turn (DESIGN
//  5.2e): moving work between them does not shorten the kernel.  tools/gpu_calls/r6o.sh ... r6q.sh, gpurun_out -> profiles/r06_enc_walk8.txt)
#ifndef ZKE_WALK
#define ZKE_WALK(taken, f, nx) asm volatile("1:\n\ts_bitset1_b64 %0, %1\n\ts_nop 0\n\tv_readlane_b32 %1, %2, %1\n\ts_cmp_lt_u32 %1, 64\n\ts_cbranch_scc1 1b" \
                                            : "+s"(taken), "+s"(f) : "v"(nx) : "scc")
#endif

// HLOG: log2 of the table entries (32-bit entries: <= 14)
template <int HLOG>
__global__ __launch_bounds__(ZKE_THREADS) void zk_k_enc_match2(const uint8_t *src, const ZkEncFrame *segs, ZkEncBlock *blocks, uint64_t *seqs, uint8_t *lits)
{
    static_assert(HLOG >= 10 && HLOG <= 14, "32-bit table entries");
    constexpr uint32_t TWORDS = 1u << HLOG, DUMMY = TWORDS;   // table[DUMMY]: where lanes without a position read and write
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    __shared__ uint32_t ring[ZKE_RING_WORDS + 8];
    __shared__ uint32_t table[TWORDS + 1];
    __shared__ uint32_t best[ZKE_GROUP_POS / 2];            // per slot of the group: length (5 bits) | catch-up bytes << 5 | offset << 8; a tile's slice later holds the tile's literal bytes
    __shared__ uint64_t tseq[ZKE_GROUP][ZKE_TSEQ_N];        // ll | ml << 12 in the low half, the offset in the high half
    __shared__ uint32_t tsum[2][ZKE_GROUP], tlast[2][ZKE_GROUP], tfirst[2][ZKE_GROUP], tfml[2][ZKE_GROUP];   // per tile (two groups deep), as in zk_k_enc_match
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const ZkEncFrame fr = segs[blockIdx.x];
    const uint8_t *base = src + fr.m_off;
    const uint32_t hist = fr.hist, fend = hist + fr.d_size, minmatch = fr.minmatch, fend4 = (fend + 3) & ~3u;
    const uint64_t lane_lt = zke_lowmask(lane);
    const uint32_t bias = (0u - hist) & (ZKE_GROUP_POS - 1);

    ZKE_CLK_BEGIN();
    // ---- segment start: empty table, history + the first group (+ lookahead) into the ring, history positions into the table
    for (uint32_t i = tid; i <= TWORDS; i += ZKE_THREADS) table[i] = NONE;
    uint32_t loaded = hist + ZKE_GROUP_POS + 64;             // the ring holds the record up to here (or to its end)
    if (loaded > fend4) loaded = fend4;
    if (fend >= 4) { for (uint32_t q = 4 * tid; q < loaded; q += 4 * ZKE_THREADS) zke_ring_put(ring, q >> 2, zke_src_dword(base, q, fend)); }
    else if (tid == 0) { uint32_t v = 0; for (uint32_t k = 0; k < fend; k++) v |= (uint32_t)base[k] << (8 * k); zke_ring_put(ring, 0, v); }
    __syncthreads();
    uint32_t stepno = 0;                                     // steps of the segment so far (the entries carry it)
    for (uint32_t c0 = 0; c0 < hist; c0 += ZKE_GROUP_POS, stepno++) {       // history in steps of 4096 positions under the step rule
        const uint32_t V0 = c0 + 4 * tid, i0 = (V0 >> 2) & 16383u;
        const uint32_t d0 = ring[i0], d1 = ring[i0 + 1], d2 = ring[i0 + 2];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t v = V0 + k, lo = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)k), hi = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)k);
            const bool ok = v < hist && v + 8 <= fend;
            atomicMin(&table[ok ? zke_hash(lo, hi & 0xFF, HLOG) : DUMMY], ok ? ((0xFFFFu - stepno) << 16) | ((v + bias) & 0xFFFFu) : NONE);
        }
    }
    __syncthreads();

    ZKE_CLK(0);
    uint32_t probe = 0;                                      // offset of the last sequence so far (same value in every lane)
    uint32_t nseq = 0, nlit = 0, pend = 0, prev_off = 0;     // parse state of the block being stitched, identical in every lane
    uint32_t par = 0;                                        // which half of tsum / tlast the group at work writes
    struct { bool valid; uint32_t gs, ge, rel; ZkEncBlock *blk; uint64_t *sq; uint8_t *lt; bool last; } todo = {false, 0, 0, 0, nullptr, nullptr, nullptr, false};
    bool held = false; uint64_t held_e = 0; uint64_t *held_at = nullptr;
    // The stitch of zk_k_enc_match (tiles summarised by tsum / tlast / tfirst / tfml, 16-lane DPP scans, seams, one sequence per lane):
    // unchanged but for where a tile's literal bytes lie (its slice of best[] is ZKE2_SLOTS words).
    auto stitch = [&]() {
        const uint32_t ntiles = (todo.ge - todo.gs + ZKE_TILE - 1) / ZKE_TILE;
        const uint32_t *ts_ = tsum[par ^ 1], *tl_ = tlast[par ^ 1], *tf_ = tfirst[par ^ 1];
        uint32_t my_base, my_lit, my_pend, my_poff, my_cnt, my_nl, my_join, my_more, me;
        bool my_open;
        {
            const uint32_t t = lane & 15;
            // (the loads are unconditional -- all sixteen entries exist -- and the selects vector ones: a load behind a condition is a branch
            //  around it, five scalar instructions each, and this block is the scalar unit's: see the walk)
            const bool in = t < ntiles;
            const uint32_t sv_ = ts_[t], lv_ = tl_[t], fo_ = tf_[t], fm = tfml[par ^ 1][t];
            const uint32_t sv = in ? sv_ : 0, lv = in ? lv_ : 0, fo = in ? fo_ : 0;
            const uint32_t cn = sv & 0xFF, tail = (sv >> 8) & 0xFFF, tnl = sv >> 20;
            uint32_t y = tail | (cn ? 0x80000000u : 0u), z = cn ? lv : 0;
#define ZKE_SCAN_STEP(d) { const uint32_t ys = ZKE_ROW_SHR0(y, d), zs = ZKE_ROW_SHR0(z, d); \
                           y = (y & 0x80000000u) ? y : (y + (ys & 0x7FFFFFFFu)) | (ys & 0x80000000u); z = z ? z : zs; }
            ZKE_SCAN_STEP(1) ZKE_SCAN_STEP(2) ZKE_SCAN_STEP(4) ZKE_SCAN_STEP(8)
#undef ZKE_SCAN_STEP
            const uint32_t ye = ZKE_ROW_SHR0(y, 1), ze = ZKE_ROW_SHR0(z, 1);            // what lies in front of tile t
            const uint32_t pend_t = (ye & 0x80000000u) ? ye & 0x7FFFFFFFu : pend + ye, poff_t = ze ? ze : prev_off;
            const bool join = (fo != 0) & (fo == poff_t) & (pend_t == 0) & (((todo.rel + t * ZKE_TILE) & (ZKE_SEAM - 1)) != 0);
            uint32_t x = (cn - (join ? 1u : 0u)) | (tnl << 16);
#define ZKE_SCAN_STEP(d) { x += ZKE_ROW_SHR0(x, d); }
            ZKE_SCAN_STEP(1) ZKE_SCAN_STEP(2) ZKE_SCAN_STEP(4) ZKE_SCAN_STEP(8)
#undef ZKE_SCAN_STEP
            const uint32_t wm = wave ? wave - 1 : 0, wn = wave < 15 ? wave + 1 : 15;
            const uint32_t xp_ = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)wm), yp_ = (uint32_t)__builtin_amdgcn_readlane((int)y, (int)wm),
                           zp_ = (uint32_t)__builtin_amdgcn_readlane((int)z, (int)wm);
            const uint32_t xp = wave ? xp_ : 0, yp = wave ? yp_ : 0, zp = wave ? zp_ : 0;
            me = (uint32_t)__builtin_amdgcn_readlane((int)sv, (int)wave);
            const uint32_t xt = (uint32_t)__builtin_amdgcn_readlane((int)x, 15), yt = (uint32_t)__builtin_amdgcn_readlane((int)y, 15), zt = (uint32_t)__builtin_amdgcn_readlane((int)z, 15);
            my_join = 0; my_more = 0; my_open = wave == 15;
            uint32_t more0 = 0, whole0 = 0;
            if (__ballot(join)) {                                                   // (most groups have no seam to close)
                uint32_t ev = join ? fm : 0, pw = (join & (cn == 1) & (tail == 0)) ? 1u : 0u;
#define ZKE_SCAN_STEP(d) { const uint32_t es = ZKE_ROW_SHL0(ev, d), ps = ZKE_ROW_SHL(pw, d, 1); ev += pw ? es : 0; pw &= ps; }
                ZKE_SCAN_STEP(1) ZKE_SCAN_STEP(2) ZKE_SCAN_STEP(4) ZKE_SCAN_STEP(8)
#undef ZKE_SCAN_STEP
                my_join = (uint32_t)__builtin_amdgcn_readlane((int)(join ? 1u : 0u), (int)wave);
                const uint32_t ev_n = (uint32_t)__builtin_amdgcn_readlane((int)ev, (int)wn), pw_n = (uint32_t)__builtin_amdgcn_readlane((int)pw, (int)wn);
                my_more = wave < 15 ? ev_n : 0;
                my_open = (wave == 15) | (pw_n != 0);                                       // every tile behind mine is whole
                more0 = (uint32_t)__builtin_amdgcn_readlane((int)ev, 0); whole0 = (uint32_t)__builtin_amdgcn_readlane((int)pw, 0);
            }
            my_open = my_open && !todo.last;
            my_base = nseq + (xp & 0xFFFF); my_lit = nlit + (xp >> 16);
            my_pend = (yp & 0x80000000u) ? yp & 0x7FFFFFFFu : pend + yp;
            my_poff = zp ? zp : prev_off;
            my_cnt = me & 0xFF; my_nl = me >> 20;
            if (held) {                                                             // (one lane of the workgroup at most)
                held_e += (uint64_t)more0 << 16;
                if (todo.last || !whole0) { *held_at = held_e; held = false; }
            }
            nseq += xt & 0xFFFF; nlit += xt >> 16;
            pend = (yt & 0x80000000u) ? yt & 0x7FFFFFFFu : pend + yt;
            if (zt) { prev_off = zt; probe = zt; }
        }
        if (wave < ntiles) {
            if ((lane >= my_join) & (lane < my_cnt)) {                              // <= 64 sequences per tile: one per lane
                const uint64_t e = tseq[wave][lane];
                uint32_t ll = (uint32_t)e & 0xFFF, ml = (uint32_t)(e >> 12) & 0xFFF;
                const uint32_t off = (uint32_t)(e >> 32);
                const uint32_t poff_ = (uint32_t)(tseq[wave][lane ? lane - 1 : 0] >> 32), poff = lane ? poff_ : my_poff;
                if (lane == 0) ll += my_pend;
                const uint32_t code = ((ll != 0) & (off == poff)) ? 1u : off + 3;
                const bool ends_tile = (lane + 1 == my_cnt) & (((me >> 8) & 0xFFF) == 0);
                if (ends_tile) ml += my_more;
                const uint64_t rec = (uint64_t)(ll | (ml << 16)) | ((uint64_t)code << 32);
                uint64_t *at = &todo.sq[my_base + lane - my_join];
                if (ends_tile & my_open) { held = true; held_e = rec; held_at = at; }
#ifndef ZKE_KNOCK_STORES
                else *at = rec;
#endif
            }
            // the tile's literals: four bytes per lane (an unaligned dword store), the last bytes one by one
            const uint32_t *tw4 = &best[wave * ZKE2_SLOTS];
            const uint8_t *tl = (const uint8_t *)tw4;
            uint8_t *o = todo.lt + my_lit;
#ifndef ZKE_KNOCK_STORES
            if (4 * lane + 4 <= my_nl) { const uint32_t w4 = tw4[lane]; memcpy(o + 4 * lane, &w4, 4); }
            else for (uint32_t i = 4 * lane; i < my_nl; i++) o[i] = tl[i];
#endif
        }
        if (todo.last) {                                                            // the block is complete
            if (tid == 0) { todo.blk->nseq = nseq; todo.blk->nlit = nlit; }
            nseq = 0; nlit = 0; pend = 0; prev_off = 0;
        }
        todo.valid = false;
    };
    for (uint32_t bi = 0; bi < fr.n_blocks; bi++) {
        const uint32_t bs = hist + bi * fr.block_max;
        const uint32_t be = bs + fr.block_max < fend ? bs + fr.block_max : fend;
        ZkEncBlock *blk = &blocks[fr.block_base + bi];
        uint64_t *sq = seqs + blk->seq_base;
        uint8_t *lt = lits + blk->lit_base;
        for (uint32_t gs = bs; gs < be; gs += ZKE_GROUP_POS) {
            const uint32_t ge = gs + ZKE_GROUP_POS < be ? gs + ZKE_GROUP_POS : be;
            // the next group's input: requested now, stored into the ring behind this group's first barrier (zk_k_enc_match has the why)
            uint32_t target = ge + ZKE_GROUP_POS + 64;
            if (target > fend4) target = fend4;
            const uint32_t pq = loaded + 4 * tid;
            const uint32_t pq_ = pq < target ? pq : fend4 - 4;
            const uint32_t pv_over = fend >= 4 && pq_ + 4 > fend ? pq_ + 4 - fend : 0;
            const uint8_t *pv_at = fend >= 4 ? base + (pq_ - pv_over) : (const uint8_t *)segs;
            uint32_t pv_raw;
            memcpy(&pv_raw, pv_at, 4);

            // ---- 1 + 2: lookups (two candidate positions) and insertions (all four positions)
            const uint32_t P0 = gs + 4 * tid;                                       // my four bytes: P0 .. P0 + 3, tile = wave; my candidate positions: P0, P0 + 2
            const uint32_t i0 = (P0 >> 2) & 16383u;
            const uint32_t dm1 = ring[(i0 - 1) & 16383u], d0 = ring[i0], d1 = ring[i0 + 1], d2 = ring[i0 + 2], d3 = ring[i0 + 3], d4 = ring[i0 + 4];
            uint32_t wlo[4], whi[4], hsh[4], tix[4], tw[2], e1[2];
            bool tabled[4];
            const uint32_t plim = fend >= 7 && fend - 7 < ge ? fend - 7 : (fend >= 7 ? ge : 0);      // positions below it have their 8 bytes and lie in the group
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wlo[k] = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)k);
                whi[k] = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)k);
                const uint32_t p = P0 + k;
                tabled[k] = p < plim;                                                // p < ge && p + 8 <= fend
                uint32_t h = zke_hash(wlo[k], whi[k] & 0xFF, HLOG);
                ZKE_KEEP(h);                                                         // (computed for every lane: behind the condition it is a branch around five instructions)
                hsh[k] = tabled[k] ? h : NONE;
                tix[k] = tabled[k] ? h : DUMMY;
            }
            tw[0] = table[tix[0]]; tw[1] = table[tix[2]];                            // far candidates: the table as it was before the step
            ZKE_CLK(1);
            ZKE_LDS_BARRIER();
            ZKE_CLK(2);
            if (pq < target) zke_ring_put(ring, (pq >> 2) & 16383u, pv_raw >> (8 * pv_over));
            loaded = target;
            {
                // a position whose hash also belongs to one of the four positions before it cannot win its slot: it stays out of the race
                const uint32_t q0 = ZKE_ROW_SHR(hsh[0], 1, NONE), q1 = ZKE_ROW_SHR(hsh[1], 1, NONE), q2 = ZKE_ROW_SHR(hsh[2], 1, NONE), q3 = ZKE_ROW_SHR(hsh[3], 1, NONE);
                bool go[4];
                go[0] = tabled[0] && hsh[0] != q3 && hsh[0] != q2 && hsh[0] != q1 && hsh[0] != q0;
                go[1] = tabled[1] && hsh[1] != hsh[0] && hsh[1] != q3 && hsh[1] != q2 && hsh[1] != q1;
                go[2] = tabled[2] && hsh[2] != hsh[1] && hsh[2] != hsh[0] && hsh[2] != q3 && hsh[2] != q2;
                go[3] = tabled[3] && hsh[3] != hsh[2] && hsh[3] != hsh[1] && hsh[3] != hsh[0] && hsh[3] != q3;
                const uint32_t khi = (0xFFFFu - stepno) << 16;
#pragma unroll
                for (int k = 0; k < 4; k++) atomicMin(&table[go[k] ? tix[k] : DUMMY], go[k] ? khi | ((P0 + k + bias) & 0xFFFFu) : NONE);
            }
            ZKE_CLK(3);
            ZKE_LDS_BARRIER();
            ZKE_CLK(4);
            e1[0] = table[tix[0]]; e1[1] = table[tix[2]];                            // near candidates: what the step left in my slots
            const uint32_t khi = (0xFFFFu - stepno) << 16;                            // the key half of this step
            stepno++;

            // ---- the group before this one: stitch its tiles, store my tile's sequences and literals
            // (Measured and dropped: half of the waves comparing before they stitch, so that the scalar work of the one and the vector
            //  work of the other overlap -- 30.6 instead of 29.6 ms.  A wave issues an instruction every ~8.5 clocks whatever its kind;
            //  with four waves per SIMD that is about one vector and one scalar instruction per turn already.)
            // (Also measured and dropped: the stitch BETWEEN the two barriers, behind the insertions' LDS atomics -- its phase 3.0 k -> 1.9 k clocks,
            //  the comparisons 3.2 k -> 4.0 k, the kernel 30.1 -> 30.5 ms: the time is the instructions, wherever they stand.)
            if (todo.valid) stitch();
            ZKE_WAVE_SYNC();
            const uint32_t R = probe;
            ZKE_CLK(8);
            const uint32_t ts = gs + wave * ZKE_TILE, te = ts + ZKE_TILE < ge ? ts + ZKE_TILE : ge;      // my tile
            {
                // the 20 bytes at my four bytes against the 20 bytes one byte / R bytes before them, once for both positions
                uint32_t x1[5], xr[5], r1[2] = {0, 0}, rr[2] = {0, 0};
                {
                    const uint32_t rb = P0 - R, ri = (rb >> 2) & 16383u, rs = rb & 3u;  // R <= P0 is tested below; a wrong address reads some ring bytes
                    x1[0] = d0 ^ __builtin_amdgcn_alignbyte(d0, dm1, 3u); x1[1] = d1 ^ __builtin_amdgcn_alignbyte(d1, d0, 3u);
                    const uint32_t r0 = ring[ri], r1 = ring[ri + 1], r2 = ring[ri + 2];
                    xr[0] = d0 ^ __builtin_amdgcn_alignbyte(r1, r0, rs); xr[1] = d1 ^ __builtin_amdgcn_alignbyte(r2, r1, rs);
                    // ... and the previous offset needs four equal bytes as well: the rest of its window only where some lane of the wave has them
                    if (__ballot(xr[0] == 0 || ((xr[0] >> 16) | (xr[1] << 16)) == 0)) {
                        const uint32_t r3 = ring[ri + 3], r4 = ring[ri + 4], r5 = ring[ri + 5];
                        xr[2] = d2 ^ __builtin_amdgcn_alignbyte(r3, r2, rs); xr[3] = d3 ^ __builtin_amdgcn_alignbyte(r4, r3, rs); xr[4] = d4 ^ __builtin_amdgcn_alignbyte(r5, r4, rs);
                        zke_runs2(xr, rr[0], rr[1]);
                    }
                }
                // offset 1 needs four equal bytes to count: x1[0] == 0 at my first position, bytes 2 .. 5 of the window at my second.  On text
                // no lane of the wave has them most of the time -- then the rest of the window and its run lengths are skipped (uniform branch)
                if (__ballot(x1[0] == 0 || ((x1[0] >> 16) | (x1[1] << 16)) == 0)) {
                    x1[2] = d2 ^ __builtin_amdgcn_alignbyte(d2, d1, 3u); x1[3] = d3 ^ __builtin_amdgcn_alignbyte(d3, d2, 3u); x1[4] = d4 ^ __builtin_amdgcn_alignbyte(d4, d3, 3u);
                    zke_runs2(x1, r1[0], r1[1]);
                }
                const bool vr0 = R > 1;
                uint32_t entry[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int k = 2 * j;
                    const uint32_t p = P0 + k;
                    const uint32_t ef = tw[j] & 0xFFFFu, en = e1[j] & 0xFFFFu;
                    const uint32_t df = (p + bias - ef) & 0xFFFFu;                      // far: what the table held before the step
                    const bool vf = tabled[k] && df && df <= p && df <= ZKE_WINDOW;
                    const uint32_t dn = (p + bias - en) & 0xFFFFu;                      // near: an earlier position of this step
                    const bool vn = tabled[k] && dn && (e1[j] & 0xFFFF0000u) == khi;
                    const uint32_t o0 = wlo[k], o1 = whi[k], o2 = j ? __builtin_amdgcn_alignbyte(d3, d2, 2u) : d2, o3 = j ? __builtin_amdgcn_alignbyte(d4, d3, 2u) : d3;
                    const uint32_t lf = zke_common16(ring, vf ? p - df : p, o0, o1, o2, o3);
                    const uint32_t ln = zke_common16(ring, vn ? p - dn : p, o0, o1, o2, o3);
                    const uint32_t n = p < te ? (te - p < ZKE_PARCAP ? te - p : ZKE_PARCAP) : 0;   // a match may not leave the tile
                    const uint32_t cf = lf < n ? lf : n, cn = ln < n ? ln : n, c1 = r1[j] < n ? r1[j] : n, cr = rr[j] < n ? rr[j] : n;
                    const bool okf = vf && cf >= minmatch, okn = vn && cn >= minmatch, ok1 = p >= 1 && c1 >= 4, okr = vr0 && R <= p && cr >= 4;
                    // the longest wins, on ties the later of far, near, offset 1, R: one key per candidate (length | rank | offset), the largest key
                    const uint32_t kf = okf ? (cf << 18) | df : 0u, kn = okn ? (cn << 18) | (1u << 16) | dn : 0u,
                                   k1 = ok1 ? (c1 << 18) | (2u << 16) | 1u : 0u, kr = okr ? ((cr + 1) << 18) | (3u << 16) | R : 0u;    // (the previous offset is cheap to code: it also wins one byte short)
                    uint32_t m = kf > kn ? kf : kn;
                    m = m > k1 ? m : k1; m = m > kr ? m : kr;
                    if (m == kr && okr) m -= 1u << 18;
                    // catch-up: how many of the four bytes in front of the position agree with the four bytes in front of the winner's
                    // source (out of the ring, which holds them while offset + 4 <= ZKE_WINDOW); not past the tile's start, not before
                    // the record's first byte.  The parse cuts it down to the literals the match really has in front of it.
                    const uint32_t off = m & 0xFFFFu, len = m >> 18;
                    const uint32_t cb = p - off - 4, ci = (cb >> 2) & 16383u;
                    const uint32_t b0 = ring[ci], b1 = ring[ci + 1];
                    const uint32_t xb = (j ? __builtin_amdgcn_alignbyte(d0, dm1, 2u) : dm1) ^ __builtin_amdgcn_alignbyte(b1, b0, cb & 3u);   // (the four bytes that end at p) ^ (... at its source)
                    uint32_t bk = ZKE_FFBH(xb);                                          // 8 x (agreeing bytes from p - 1 downwards); 0xFFFFFFFF: all four
                    bk = (bk < 32u ? bk : 32u) >> 3;
                    const uint32_t room = p - ts < p - off ? p - ts : p - off;           // (p >= ts: the lane's bytes lie in its tile)
                    bk = bk < room ? bk : room;
                    if (!len || off + 4 > ZKE_WINDOW) bk = 0;
                    entry[j] = len | (bk << 5) | (off << 8);                            // positions past the tile's end: length 0
                }
                best[2 * tid] = entry[0]; best[2 * tid + 1] = entry[1];
            }
            ZKE_CLK(5);
            // ---- 3b: wave w parses tile w (it wrote that slice of best[] itself: LDS operations of a wave complete in order)
            ZKE_WAVE_SYNC();
            if (ts < ge) {                                                              // uniform per wave
                uint8_t *tl = (uint8_t *)&best[wave * ZKE2_SLOTS];                      // the tile's literal bytes, behind the entries already read
                uint32_t skip = 0, c = 0, nl = 0, aend = 0, lastoff = 0;                // tile-relative bytes: first byte not covered yet; sequences; literals; end / offset of the last match
                // The tile in two passes of 64 slots (a slot = an even position and the byte behind it), in sweeps:
                //   A   every pass: best[] entries, candidate masks, per-lane next-candidate table
                //   B   the walk, pass after pass (the only serial part: where a match ends decides which candidate is next)
                //   C1  every pass: where the taken matches really start (catch-up bytes), the sequences
                //   C2  every pass: the literal bytes, all lanes at once (a byte in front of a taken match may belong to its catch-up,
                //       also when that match is the first one of the NEXT pass: hence behind C1 of both)
                uint32_t pv[2], plen[2], pnx[2], ppe[2], pst[2], first_s[2];
                uint64_t pcand[2], pmore[2], ptaken[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t v = best[wave * ZKE2_SLOTS + 64 * u + lane];         // length 0 past the tile's end
                    uint32_t len = v & 0x1F;
                    // a CHEAP offset at the next slot -- the previous offset R or offset 1 -- wins against a fresh offset here even two bytes
                    // shorter (the twin has the measurements: runs of 10 equal bytes 5.5 -> 8.8, of 100: 30 -> 55; the text unchanged)
                    const uint32_t s1 = 64 * u + lane + 1;
                    const uint32_t v1 = best[wave * ZKE2_SLOTS + (s1 < ZKE2_SLOTS ? s1 : 0)];
                    const uint32_t l1 = s1 < ZKE2_SLOTS ? v1 & 0x1F : 0, o0 = v >> 8, o1 = v1 >> 8;
                    const bool cand = len != 0 && !(l1 && o0 != R && o0 != 1 && (o1 == R || o1 == 1) && l1 + 2 >= len);
                    pcand[u] = __ballot(cand);
                    // A match the comparisons capped (16 bytes) usually goes on with the SAME offset at the slot 16 bytes on (every slot inside a
                    // long match is a candidate of it): then its length is 16 + that slot's, exact when that one is not capped itself.  What
                    // is left -- another winner there, or capped again -- is measured in the walk, byte for byte, and marked here.  (The
                    // walk is the scalar unit's, sixteen waves at a time: every instruction of a step counts, so nothing else stays in it.)
                    const uint32_t s8 = 64 * u + lane + 8;
                    const uint32_t v8 = best[wave * ZKE2_SLOTS + (s8 < ZKE2_SLOTS ? s8 : 0)];
                    const bool cap = len == ZKE_PARCAP && ts + 2 * s8 < te;              // (capped at the tile's end: the length is exact)
                    const bool same = s8 < ZKE2_SLOTS && (v8 >> 8) == (v >> 8);
                    const uint32_t len8 = v8 & 0x1F;
                    if (cap && same) len += len8;
                    const bool more = cap && (!same || len8 == ZKE_PARCAP);
                    pv[u] = v; plen[u] = len;
                    pmore[u] = __ballot(more);                                          // to be measured in the walk
                    // every lane: the first candidate slot at or behind the end of its own match (64: none in this pass); bit 7: THAT
                    // candidate is to be measured (an upper bound of a marked lane's own end is all its entry needs: nobody follows it)
                    const uint32_t el = lane + ((len + 1) >> 1);
                    const uint64_t behind = pcand[u] & (el >= 64 ? 0ull : ~0ull << el);
                    const uint32_t nxs = behind ? (uint32_t)__builtin_ctzll(behind) : 64u;
                    pnx[u] = nxs | (((pmore[u] >> (nxs & 63)) & 1) && nxs < 64 ? 0x80u : 0u);
                }
                ZKE_CLK(9);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t wb = 128 * u;
                    const uint64_t candm = pcand[u], morem = pmore[u];
                    const uint32_t v = pv[u], nx = pnx[u];
                    uint32_t len = plen[u];
                    const uint32_t pre = skip > wb ? (skip - wb + 1) >> 1 : 0;          // slots below it start inside a match of the pass before
                    const uint64_t open = candm & (pre >= 64 ? 0ull : ~0ull << pre);
                    uint32_t f = open ? (uint32_t)__builtin_ctzll(open) : 64u;
                    if (f < 64 && ((morem >> f) & 1)) f |= 0x80u;
                    uint64_t taken = 0;
                    for (;;) {                                                          // uniform: every lane walks the same chain
                        if (f < 64) ZKE_WALK(taken, f, nx);
                        if (f == 64) break;
                        // slot f & 0x7F is to be measured: take it and extend, 64 bytes per step (rare)
                        ZKE_DBG_SLOW();
                        f &= 0x7Fu;
                        taken |= 1ull << f;
                        const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)f) >> 8;
                        uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)f);       // 16 or 32 bytes are known
                        for (;;) {
                            const uint32_t q = ts + wb + 2 * f + L + lane;
                            const bool diff = q >= te || zke_ring1(ring, q) != zke_ring1(ring, q - off);
                            const uint64_t dm = __ballot(diff);
                            if (dm) { L += (uint32_t)__builtin_ctzll(dm); break; }
                            L += 64;
                        }
                        if (lane == f) len = L;
                        const uint32_t e = f + ((L + 1) >> 1);
                        const uint64_t rest = candm & (e >= 64 ? 0ull : ~0ull << e);
                        f = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
                        if (f < 64 && ((morem >> f) & 1)) f |= 0x80u;
                    }
                    if (taken) { const uint32_t lastf = 63u - (uint32_t)__builtin_clzll(taken); skip = wb + 2 * lastf + (uint32_t)__builtin_amdgcn_readlane((int)len, (int)lastf); }
                    ptaken[u] = taken; plen[u] = len;
                }
                ZKE_CLK(10);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    // a taken lane's sequence index = sequences so far + taken lanes below it; its match starts up to `catch-up` bytes
                    // in front of its position, but not before the end of the match before it; its literal length = that start - that end
                    const uint32_t pos = 128 * u + 2 * lane;
                    const uint64_t taken = ptaken[u];
                    const uint32_t v = pv[u], len = plen[u];
                    const uint64_t below = taken & lane_lt;
                    const uint32_t myend = pos + len;
                    const uint32_t prevlane = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                    const uint32_t pe_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(prevlane << 2), (int)myend);
                    const uint32_t pe = below ? pe_ : aend;                             // end of the last match in front of me
                    const bool mine = (taken >> lane) & 1;
                    uint32_t bk = (v >> 5) & 7u;
                    bk = bk < pos - pe ? bk : pos - pe;                                 // (a taken lane: pos >= pe)
                    const uint32_t st = pos - bk;
                    if (mine) tseq[wave][c + (uint32_t)__builtin_popcountll(below)] = (uint64_t)((st - pe) | ((len + bk) << 12)) | ((uint64_t)(v >> 8) << 32);
                    ppe[u] = pe; pst[u] = st;
                    first_s[u] = 0xFFFFu;
                    if (taken) {
                        const uint32_t lastl = 63u - (uint32_t)__builtin_clzll(taken);
                        c += (uint32_t)__builtin_popcountll(taken);
                        aend = 128 * u + 2 * lastl + (uint32_t)__builtin_amdgcn_readlane((int)len, (int)lastl);
                        lastoff = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lastl) >> 8;
                        first_s[u] = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)__builtin_ctzll(taken));
                    }
                }
                ZKE_CLK(11);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    // literal bytes: my two bytes unless a match covers them -- the one in front of me (ends at pe), my own, or the catch-up
                    // of the next taken match (starts at sn: in this pass, else the first one of the next pass)
                    const uint32_t pos = 128 * u + 2 * lane, p = ts + pos;
                    const uint64_t taken = ptaken[u];
                    const uint64_t above = taken & ~lane_lt & ~(1ull << lane);
                    const uint32_t nextlane = above ? (uint32_t)__builtin_ctzll(above) : 0u;
                    const uint32_t sn_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nextlane << 2), (int)pst[u]);
                    const uint32_t sn = above ? sn_ : (u == 0 ? first_s[1] : 0xFFFFu);
                    const bool mine = (taken >> lane) & 1;
                    const bool l0 = p < te && !mine && pos >= ppe[u] && pos < sn;
                    const bool l1 = p + 1 < te && !mine && pos + 1 >= ppe[u] && pos + 1 < sn;
                    const uint64_t m0 = __ballot(l0), m1 = __ballot(l1);
                    const uint32_t at = nl + (uint32_t)__builtin_popcountll(m0 & lane_lt) + (uint32_t)__builtin_popcountll(m1 & lane_lt);
                    const uint32_t two = ((const uint16_t *)ring)[(p & 0xFFFFu) >> 1];  // my two bytes (p is even)
                    if (l0) tl[at] = (uint8_t)two;
                    if (l1) tl[at + (l0 ? 1u : 0u)] = (uint8_t)(two >> 8);
                    nl += (uint32_t)__builtin_popcountll(m0) + (uint32_t)__builtin_popcountll(m1);
                }
                ZKE_CLK(12);
                ZKE_DBG_TILE(c);
                if (lane == 0) {
                    tsum[par][wave] = c | (((te - ts) - aend) << 8) | (nl << 20);
                    tlast[par][wave] = lastoff;
                    const uint64_t e0 = tseq[wave][0];                                  // a first sequence at the tile's first byte: its offset (else 0), its length
                    tfirst[par][wave] = c && ((uint32_t)e0 & 0xFFF) == 0 ? (uint32_t)(e0 >> 32) : 0u;
                    tfml[par][wave] = (uint32_t)(e0 >> 12) & 0xFFF;
                }
            }
            ZKE_CLK(6);
            todo.valid = true; todo.gs = gs; todo.ge = ge; todo.rel = gs - bs; todo.blk = blk; todo.sq = sq; todo.lt = lt; todo.last = ge == be;
            par ^= 1;
        }
    }
    if (todo.valid) { ZKE_LDS_BARRIER(); stitch(); }        // the segment's last group
    ZKE_CLK_END();
}
