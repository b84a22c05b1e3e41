// zk_enc_match.h -- the match + parse kernel of the gfx950 frame encoder (included by zk_encode.hip; tests/sim/zk_enc_sim.cpp
// runs the same source on the CPU under a workgroup emulator).
//
// Replaces the inside of ZSTD_compressStream2 that zeekstd drives frame by frame (lib/src/encode.rs:340-346): for every
// SEGMENT (<= 256 KiB of one frame) one workgroup of 16 waves finds matches and parses them into sequences + literals.
// Decisions are those of the CPU twin oracle/zstd_oracle_enc.c (find_sequences), which the output is compared with
// sequence for sequence.
//
// Everything the matcher touches while it works lives in LDS (HBM-bound byte work; no MFMA):
//   ring   64 KiB  the last 65536 bytes of the segment's record [history | data], loaded ONCE from HBM, coalesced, one dword
//                  per lane and group, a group ahead; position p sits at ring byte p mod 65536
//   table  2^HLOG 16-bit entries = ring indices of earlier positions (32 / 64 KiB)
//   best   16 KiB  per position of the group: length | offset << 8 of its best candidate; a tile's slice later holds the
//                  tile's literal bytes
//   tseq   8.3 KiB per tile: its sequences
// so a candidate comparison, a match extension and a literal gather are LDS reads (32 lanes per clock), and HBM sees the
// input once and the sequences / literals once, in full lines.
//
// A group = 16 tiles x 256 positions.  Lane l of wave w owns positions 4l .. 4l + 3 of tile w (aligned ring words):
//   1. lookup: hash of 5 bytes -> table entry as it was before the step ("far" candidate)            | barrier (LDS only)
//   2. insert: compare-and-swap, the smallest position of the step wins a slot                       | barrier
//   3. second lookup: an earlier position of the same step ("near" candidate); comparisons of far / near / offset 1 /
//      previous offset R out of the ring -> best[]; then the wave parses ITS tile (the same wave wrote its best[] slice:
//      no barrier): 64 positions per pass, candidates as a ballot mask consumed by a scalar loop, sequences and literal
//      bytes emitted by all lanes at once                                                            | barrier
//   4. the prefetched input of the next group enters the ring; the 16 tiles are stitched (every wave runs the little
//      scan over the 16 tile summaries itself) and each wave stores its tile's sequences and literals | barrier
#pragma once
#include <stdint.h>
#include "zk_enc_device.h"

constexpr int ZKE_THREADS = 1024;                      // 16 waves: one per tile of a group
constexpr uint32_t ZKE_TSEQ_N = ZKE_TILE / 4 + 2;      // sequences of a tile (shortest match: 4 bytes)
static_assert(ZKE_THREADS * 4 == (int)ZKE_GROUP_POS && ZKE_THREADS / 64 == (int)ZKE_GROUP, "four positions per lane, one wave per tile");

#ifndef ZKE_LDS_BARRIER
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load / store in flight
#define ZKE_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// 8 bytes of the record at position pos (any alignment) out of the ring
__device__ __forceinline__ uint64_t zke_ring8(const uint32_t *ring, uint32_t pos)
{
    const uint32_t i = (pos >> 2) & 16383u, sh = pos & 3u;
    const uint32_t a = ring[i], b = ring[(i + 1) & 16383u], c = ring[(i + 2) & 16383u];
    return (uint64_t)__builtin_amdgcn_alignbyte(b, a, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(c, b, sh) << 32);
}
__device__ __forceinline__ uint32_t zke_ring1(const uint32_t *ring, uint32_t pos) { return ((const uint8_t *)ring)[pos & 0xFFFFu]; }

// common prefix of the bytes at a and at b (a < b), at most n (>= 1); w = the 8 bytes at b
__device__ __forceinline__ uint32_t zke_mlen(const uint32_t *ring, uint32_t a, uint32_t b, uint64_t w, uint32_t n)
{
    uint64_t x = w ^ zke_ring8(ring, a);
    uint32_t l = 0;
    while (x == 0) {
        l += 8;
        if (l >= n) return n;
        x = zke_ring8(ring, b + l) ^ zke_ring8(ring, a + l);
    }
    l += (uint32_t)__builtin_ctzll(x) >> 3;
    return l < n ? l : n;
}

__device__ __forceinline__ uint32_t zke_table_get(const uint32_t *table, uint32_t h) { return (table[h >> 1] >> (16 * (h & 1))) & 0xFFFFu; }

// Insert position p into slot h.  mode 0: history -- the numerically largest position wins (positions < 65536, the empty
// entry 0 loses).  mode 1: a step [ls, ls + span) -- the SMALLEST position of the step wins, any entry that is not of the
// step (entry - ls mod 2^16 >= span) loses.  Order-free rules: the lanes race with compare-and-swap on the word that
// holds two entries, oracle/zstd_oracle_enc.c applies them sequentially.
template <int MODE>
__device__ __forceinline__ void zke_table_put(uint32_t *table, uint32_t h, uint32_t p, uint32_t ls, uint32_t span)
{
    uint32_t *w = &table[h >> 1];
    const uint32_t sh = 16 * (h & 1), mine = p & 0xFFFFu, b16 = ls & 0xFFFFu, mrel = (mine - b16) & 0xFFFFu;
    uint32_t old = *w;
    for (;;) {
        const uint32_t e = (old >> sh) & 0xFFFFu;
        if (MODE == 0) { if (e >= mine) return; }
        else { const uint32_t cur = (e - b16) & 0xFFFFu; if (cur < span && cur <= mrel) return; }
        const uint32_t seen = atomicCAS(w, old, (old & ~(0xFFFFu << sh)) | (mine << sh));
        if (seen == old) return;
        old = seen;
    }
}

// the source dword at record position q (a multiple of 4); bytes at and past fend read as 0
__device__ __forceinline__ uint32_t zke_src_dword(const uint8_t *base, uint32_t q, uint32_t fend)
{
    if (q + 4 <= fend) { uint32_t v; memcpy(&v, base + q, 4); return v; }
    uint32_t v = 0;
    for (uint32_t k = 0; q + k < fend; k++) v |= (uint32_t)base[q + k] << (8 * k);
    return v;
}

__device__ __forceinline__ uint64_t zke_lowmask(uint32_t n) { return n >= 64 ? ~0ull : (1ull << n) - 1; }     // bits [0, n)

// HLOG: log2 of the table entries; LAZY: a longer match one (or a clearly longer one two) positions later wins;
// STEP: positions per lookup step (4096 = the whole group at once, 1024 = four steps of 256 lanes each)
template <int HLOG, int LAZY, int STEP>
__global__ __launch_bounds__(ZKE_THREADS) void zk_k_enc_match(const uint8_t *src, const ZkEncFrame *segs, ZkEncBlock *blocks, uint64_t *seqs, uint8_t *lits)
{
    static_assert(HLOG >= 10 && HLOG <= 15 && (STEP == 1024 || STEP == (int)ZKE_GROUP_POS), "parameters");
    __shared__ uint32_t ring[ZKE_RING / 4];
    __shared__ uint32_t table[1 << (HLOG - 1)];
    __shared__ uint32_t best[ZKE_GROUP_POS];
    __shared__ uint64_t tseq[ZKE_GROUP][ZKE_TSEQ_N];        // ll | ml << 12 | offset << 24
    __shared__ uint32_t tsum[ZKE_GROUP], tlast[ZKE_GROUP];   // count | trailing literals << 8 | literal bytes << 20;  offset of the tile's last sequence
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ZkEncFrame fr = segs[blockIdx.x];
    const uint8_t *base = src + fr.m_off;
    const uint32_t hist = fr.hist, fend = hist + fr.d_size, minmatch = fr.minmatch;
    const uint64_t lane_lt = zke_lowmask(lane);

    // ---- segment start: empty table, history + the first group (+ lookahead) into the ring, history positions into the table
    for (uint32_t i = tid; i < (1u << (HLOG - 1)); i += ZKE_THREADS) table[i] = 0;
    uint32_t loaded = hist + ZKE_GROUP_POS + 64;             // the ring holds the record up to here (or to its end)
    if (loaded > ((fend + 3) & ~3u)) loaded = (fend + 3) & ~3u;
    for (uint32_t q = 4 * tid; q < loaded; q += 4 * ZKE_THREADS) ring[q >> 2] = zke_src_dword(base, q, fend);
    __syncthreads();
    for (uint32_t v = tid; v < hist; v += ZKE_THREADS)
        if (v + 8 <= fend) { const uint64_t w = zke_ring8(ring, v); zke_table_put<0>(table, zke_hash((uint32_t)w, (uint32_t)(w >> 32) & 0xFF, HLOG), v, 0, 0); }
    __syncthreads();

    uint32_t probe = 0;                                      // offset of the last sequence so far (same value in every lane)
    for (uint32_t bi = 0; bi < fr.n_blocks; bi++) {
        const uint32_t bs = hist + bi * fr.block_max;
        const uint32_t be = bs + fr.block_max < fend ? bs + fr.block_max : fend;
        ZkEncBlock *blk = &blocks[fr.block_base + bi];
        uint64_t *sq = seqs + blk->seq_base;
        uint8_t *lt = lits + blk->lit_base;
        uint32_t nseq = 0, nlit = 0, pend = 0, prev_off = 0;     // block-level parse state, identical in every lane
        for (uint32_t gs = bs; gs < be; gs += ZKE_GROUP_POS) {
            const uint32_t ge = gs + ZKE_GROUP_POS < be ? gs + ZKE_GROUP_POS : be;
            const uint32_t R = probe;
            // the next group's input: requested now, stored into the ring after this group's comparisons (it overwrites the
            // oldest bytes of this group's window)
            uint32_t target = ge + ZKE_GROUP_POS + 64;
            if (target > ((fend + 3) & ~3u)) target = (fend + 3) & ~3u;
            const uint32_t pq = loaded + 4 * tid;
            const uint32_t pv = pq < target ? zke_src_dword(base, pq, fend) : 0;

            // ---- 1 + 2: lookups and insertions, step by step
            const uint32_t P0 = gs + 4 * tid;                                       // my four positions: P0 .. P0 + 3, tile = wave
            const uint32_t i0 = (P0 >> 2) & 16383u;
            const uint32_t dm1 = ring[(i0 - 1) & 16383u], d0 = ring[i0], d1 = ring[(i0 + 1) & 16383u], d2 = ring[(i0 + 2) & 16383u];
            uint32_t wlo[4], whi[4], hsh[4], e0[4], e1[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wlo[k] = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)k);
                whi[k] = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)k);
                const uint32_t p = P0 + k;
                hsh[k] = p < ge && p + 8 <= fend ? zke_hash(wlo[k], whi[k] & 0xFF, HLOG) : 0xFFFFFFFFu;
                e0[k] = 0; e1[k] = 0;
            }
            constexpr uint32_t LPS = STEP / 4;                                      // lanes per step
            const uint32_t mystep = tid / LPS;
#pragma unroll
            for (uint32_t s = 0; s < ZKE_GROUP_POS / STEP; s++) {
                const uint32_t ls = gs + s * STEP, le = ls + STEP < ge ? ls + STEP : ge;
                if (ls < ge) {                                                     // uniform
                    if (mystep == s) {
#pragma unroll
                        for (int k = 0; k < 4; k++) if (hsh[k] != 0xFFFFFFFFu) e0[k] = zke_table_get(table, hsh[k]);
                    }
                    ZKE_LDS_BARRIER();
                    // A position whose hash also belongs to one of the four positions before it cannot win its slot (the smaller
                    // position does): it stays out of the race.  On runs of equal bytes or short periods all lanes of a step
                    // would otherwise fight over a few LDS words.  The table ends the step in the same state.
                    const uint32_t q0 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)hsh[0], 0x111, 0xF, 0xF, false);
                    const uint32_t q1 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)hsh[1], 0x111, 0xF, 0xF, false);
                    const uint32_t q2 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)hsh[2], 0x111, 0xF, 0xF, false);
                    const uint32_t q3 = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)hsh[3], 0x111, 0xF, 0xF, false);
                    if (mystep == s) {
                        const uint32_t span = le - ls;
                        if (hsh[0] != 0xFFFFFFFFu && hsh[0] != q3 && hsh[0] != q2 && hsh[0] != q1 && hsh[0] != q0) zke_table_put<1>(table, hsh[0], P0, ls, span);
                        if (hsh[1] != 0xFFFFFFFFu && hsh[1] != hsh[0] && hsh[1] != q3 && hsh[1] != q2 && hsh[1] != q1) zke_table_put<1>(table, hsh[1], P0 + 1, ls, span);
                        if (hsh[2] != 0xFFFFFFFFu && hsh[2] != hsh[1] && hsh[2] != hsh[0] && hsh[2] != q3 && hsh[2] != q2) zke_table_put<1>(table, hsh[2], P0 + 2, ls, span);
                        if (hsh[3] != 0xFFFFFFFFu && hsh[3] != hsh[2] && hsh[3] != hsh[1] && hsh[3] != hsh[0] && hsh[3] != q3) zke_table_put<1>(table, hsh[3], P0 + 3, ls, span);
                    }
                    ZKE_LDS_BARRIER();
                    if (mystep == s) {
#pragma unroll
                        for (int k = 0; k < 4; k++) if (hsh[k] != 0xFFFFFFFFu) e1[k] = zke_table_get(table, hsh[k]);
                    }
                }
            }

            // ---- 3a: comparisons out of the ring -> best[]
            const uint32_t ts = gs + wave * ZKE_TILE, te = ts + ZKE_TILE < ge ? ts + ZKE_TILE : ge;      // my tile
            {
                const uint32_t ls = gs + mystep * STEP;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t p = P0 + k;
                    if (p < ge) {
                        const uint64_t w = (uint64_t)wlo[k] | ((uint64_t)whi[k] << 32);
                        const uint32_t n = te - p < ZKE_PARCAP ? te - p : ZKE_PARCAP;   // a match may not leave the tile
                        uint32_t bl = 0, bo = 0;
                        if (hsh[k] != 0xFFFFFFFFu) {
                            uint32_t d = (p - e0[k]) & 0xFFFFu;                         // far: what the table held before the step
                            if (d && d <= p && d <= ZKE_WINDOW) { const uint32_t l = zke_mlen(ring, p - d, p, w, n); if (l >= minmatch) { bl = l; bo = d; } }
                            d = (p - e1[k]) & 0xFFFFu;                                  // near: an earlier position of this step
                            if (d && d <= p - ls) { const uint32_t l = zke_mlen(ring, p - d, p, w, n); if (l >= minmatch && l >= bl) { bl = l; bo = d; } }
                        }
                        if (p >= 1) {                                                   // offset 1, out of registers: bytes p - 1 .. p + 6
                            const uint64_t w1 = k == 0 ? ((uint64_t)__builtin_amdgcn_alignbyte(d0, dm1, 3u) | ((uint64_t)__builtin_amdgcn_alignbyte(d1, d0, 3u) << 32))
                                                       : ((uint64_t)wlo[k ? k - 1 : 0] | ((uint64_t)whi[k ? k - 1 : 0] << 32));
                            const uint64_t x = w ^ w1;
                            uint32_t l = x ? (uint32_t)__builtin_ctzll(x) >> 3 : zke_mlen(ring, p - 1, p, w, n);
                            if (l > n) l = n;
                            if (l >= 4 && l >= bl) { bl = l; bo = 1; }
                        }
                        if (R > 1 && R <= p) { const uint32_t l = zke_mlen(ring, p - R, p, w, n); if (l >= 4 && l >= bl) { bl = l; bo = R; } }
                        best[4 * tid + k] = bl | (bo << 8);
                    }
                }
            }
            // ---- 3b: wave w parses tile w (it wrote that slice of best[] itself: LDS operations of a wave complete in order)
            __builtin_amdgcn_wave_barrier();
            if (ts < ge) {                                                              // uniform per wave
                uint8_t *tl = (uint8_t *)&best[wave * ZKE_TILE];                        // the tile's literal bytes, behind the entries already read
                uint32_t skip = 0, c = 0, nl = 0, aend = 0, lastoff = 0;                // tile-relative: first position not covered yet; sequences; literals; end / offset of the last match
                for (uint32_t wb = 0; ts + wb < te; wb += 64) {
                    const uint32_t pos = wb + lane, p = ts + pos;
                    const bool in = p < te;
                    const uint32_t v = in ? best[wave * ZKE_TILE + pos] : 0;
                    uint32_t len = v & 0xFF;
                    bool cand = len != 0;
                    if (LAZY) {
                        const uint32_t i1 = wave * ZKE_TILE + pos + 1, i2 = i1 + 1;
                        const uint32_t l1 = p + 1 < te ? best[i1 < ZKE_GROUP_POS ? i1 : 0] & 0xFF : 0;
                        const uint32_t l2 = p + 2 < te ? best[i2 < ZKE_GROUP_POS ? i2 : 0] & 0xFF : 0;
                        if (l1 > len || l2 > len + 1) cand = false;
                    }
                    const uint64_t inmask = __ballot(in);
                    uint64_t m = __ballot(cand);
                    const uint32_t pre = skip > wb ? skip - wb : 0;                     // positions of this pass a match from the pass before covers
                    uint64_t covered = zke_lowmask(pre), taken = 0;
                    m &= ~covered;
                    while (m) {                                                         // uniform: every lane walks the same mask
                        const uint32_t f = (uint32_t)__builtin_ctzll(m);
                        uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)f);
                        if (L == ZKE_PARCAP) {                                          // capped by the comparisons: extend, 64 bytes per step
                            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)f) >> 8;
                            for (;;) {
                                const uint32_t q = ts + wb + f + L + lane;
                                const bool diff = q >= te || zke_ring1(ring, q) != zke_ring1(ring, q - off);
                                const uint64_t dm = __ballot(diff);
                                if (dm) { L += (uint32_t)__builtin_ctzll(dm); break; }
                                L += 64;
                            }
                            if (lane == f) len = L;
                        }
                        taken |= 1ull << f;
                        const uint32_t e = f + L;
                        covered |= zke_lowmask(e) & ~zke_lowmask(f);
                        skip = wb + e;
                        m = e >= 64 ? 0 : m & ~zke_lowmask(e);
                    }
                    // emission, all lanes at once: a taken lane's sequence index = sequences so far + taken lanes below it; its
                    // literal length = its position - the end of the taken lane before it
                    const uint64_t below = taken & lane_lt;
                    const uint32_t myend = pos + len;
                    const uint32_t prevlane = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                    const uint32_t pe = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(prevlane << 2), (int)myend);
                    if ((taken >> lane) & 1) {
                        const uint32_t prev_end = below ? pe : aend;
                        tseq[wave][c + (uint32_t)__builtin_popcountll(below)] = (uint64_t)(pos - prev_end) | ((uint64_t)len << 12) | ((uint64_t)(v >> 8) << 24);
                    }
                    if (taken) {
                        c += (uint32_t)__builtin_popcountll(taken); aend = skip;
                        lastoff = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(63u - (uint32_t)__builtin_clzll(taken))) >> 8;
                    }
                    const uint64_t litm = inmask & ~covered;
                    if ((litm >> lane) & 1) tl[nl + (uint32_t)__builtin_popcountll(litm & lane_lt)] = (uint8_t)zke_ring1(ring, p);
                    nl += (uint32_t)__builtin_popcountll(litm);
                }
                if (lane == 0) {
                    tsum[wave] = c | (((te - ts) - aend) << 8) | (nl << 20);
                    tlast[wave] = lastoff;
                }
            }
            ZKE_LDS_BARRIER();

            // ---- 4: ring <- next group's input; stitch the tiles; store sequences and literals
            if (pq < target) ring[(pq >> 2) & 16383u] = pv;
            loaded = target;
            const uint32_t ntiles = (ge - gs + ZKE_TILE - 1) / ZKE_TILE;
            uint32_t my_base = 0, my_lit = 0, my_pend = 0, my_poff = 0, my_cnt = 0, my_nl = 0;
            {
                const uint32_t sv = (lane & 15) < ntiles ? tsum[lane & 15] : 0, lv = (lane & 15) < ntiles ? tlast[lane & 15] : 0;
#pragma unroll
                for (uint32_t t = 0; t < ZKE_GROUP; t++) {
                    const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)sv, (int)t), lo = (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)t);
                    if (t < ntiles) {
                        const uint32_t cn = s & 0xFF, tail = (s >> 8) & 0xFFF, tnl = s >> 20;
                        if (t == wave) { my_base = nseq; my_lit = nlit; my_pend = pend; my_poff = prev_off; my_cnt = cn; my_nl = tnl; }
                        if (cn) { pend = tail; prev_off = lo; probe = lo; nseq += cn; }
                        else pend += tail;
                        nlit += tnl;
                    }
                }
            }
            if (wave < ntiles) {
                for (uint32_t j = lane; j < my_cnt; j += 64) {
                    const uint64_t e = tseq[wave][j];
                    uint32_t ll = (uint32_t)e & 0xFFF;
                    const uint32_t ml = (uint32_t)(e >> 12) & 0xFFF, off = (uint32_t)(e >> 24);
                    const uint32_t poff = j ? (uint32_t)(tseq[wave][j - 1] >> 24) : my_poff;
                    if (j == 0) ll += my_pend;
                    const uint32_t code = (ll && off == poff) ? 1u : off + 3;
                    sq[my_base + j] = (uint64_t)ll | ((uint64_t)ml << 20) | ((uint64_t)code << 40);
                }
                const uint8_t *tl = (const uint8_t *)&best[wave * ZKE_TILE];
                for (uint32_t i = lane; i < my_nl; i += 64) lt[my_lit + i] = tl[i];
            }
            ZKE_LDS_BARRIER();                                 // the ring's new bytes are visible; best[] / tseq[] may be reused
        }
        if (tid == 0) { blk->nseq = nseq; blk->nlit = nlit; }
    }
}
