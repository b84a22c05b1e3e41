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
//                  per lane and group, a group ahead; position p sits at ring byte p mod 65536 (+ a 32-byte mirror of the
//                  ring's start behind its end, so the 20 bytes at any position are one straight read)
//   table  2^HLOG 16-bit entries = ring indices of earlier positions (32 / 64 KiB)
//   best   16 KiB  per position of the group: length (5 bits) | offset << 5 of its best candidate; a tile's slice later holds the
//                  tile's literal bytes
//   tseq   8.3 KiB per tile: its sequences
// so a candidate comparison, a match extension and a literal gather are LDS reads, and HBM sees the input once and the
// sequences / literals once.
//
// A group = 16 tiles x 256 positions.  Lane l of wave w owns positions 4l .. 4l + 3 of tile w (aligned ring words):
//   1. lookup: hash of 5 bytes -> table entry as it was before the step ("far" candidate)            | barrier (LDS only)
//   2. insert: compare-and-swap, the smallest position of the step wins a slot                       | barrier
//   3. second lookup: an earlier position of the same step ("near" candidate); far / near / offset 1 / previous offset R
//      are compared over 16 bytes each, branch-free, all ring reads of a lane in flight together -> best[]; then the wave
//      parses ITS tile (the same wave wrote its best[] slice: no barrier): 64 positions per pass, candidates as a ballot mask
//      consumed by a short scalar loop, sequences and literal bytes emitted by all lanes at once      | barrier
//   4. the prefetched input of the next group enters the ring; the 16 tiles are stitched (16-lane DPP scans over the tile
//      summaries, every wave on its own) and each wave stores its tile's sequences and literals        | barrier
// The scalar unit is one per CU and 16 waves share it: the loops are written to keep their scalar instruction count low.
#pragma once
#include <stdint.h>
#include "zk_enc_device.h"

constexpr int ZKE_THREADS = 1024;                      // 16 waves: one per tile of a group
constexpr uint32_t ZKE_TSEQ_N = ZKE_TILE / 4 + 2;      // sequences of a tile (shortest match: 4 bytes)
constexpr uint32_t ZKE_RING_WORDS = ZKE_RING / 4;
static_assert(ZKE_THREADS * 4 == (int)ZKE_GROUP_POS && ZKE_THREADS / 64 == (int)ZKE_GROUP, "four positions per lane, one wave per tile");
static_assert(ZKE_PARCAP == 16 && ZKE_TSEQ_N >= 64, "the comparisons measure 16 bytes; one sequence per lane in the stitch");

#ifndef ZKE_CLK
#define ZKE_CLK(i) do { } while (0)       // experiments: shader-clock totals per phase (zk_encode.hip with -DZKE_CLOCKS, tools/enc_clocks.py)
#define ZKE_CLK_BEGIN() do { } while (0)
#define ZKE_CLK_END() do { } while (0)
#endif
#ifndef ZKE_WAVE_SYNC
// Between a wave's LDS writes and its reads of what OTHER lanes of the same wave wrote: the hardware runs a wave's LDS operations in
// order, so all that is needed is that the compiler keeps the order.  (__builtin_amdgcn_wave_barrier() also made the wave wait for
// every global store it had in flight -- the stitch's -- once per group.)
#define ZKE_WAVE_SYNC() asm volatile("" ::: "memory")
#endif
#ifndef ZKE_LDS_BARRIER
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load / store in flight
#define ZKE_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

__device__ __forceinline__ void zke_ring_put(uint32_t *ring, uint32_t word, uint32_t v)   // word < 16384; the first 8 words are mirrored behind the end
{
    ring[word] = v;
    if (word < 8) ring[word + ZKE_RING_WORDS] = v;
}
// 8 bytes of the record at position pos (any alignment) out of the ring
__device__ __forceinline__ uint64_t zke_ring8(const uint32_t *ring, uint32_t pos)
{
    const uint32_t i = (pos >> 2) & 16383u, sh = pos & 3u;
    const uint32_t a = ring[i], b = ring[i + 1], c = ring[i + 2];
    return (uint64_t)__builtin_amdgcn_alignbyte(b, a, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(c, b, sh) << 32);
}
__device__ __forceinline__ uint32_t zke_ring1(const uint32_t *ring, uint32_t pos) { return ((const uint8_t *)ring)[pos & 0xFFFFu]; }

// v_ffbl_b32 itself: the index of the lowest set bit, 0xFFFFFFFF for 0 (__ffs() - 1 says the same, but the compiler guards the
// zero case with a compare and a select of its own: two more instructions per word, 80 per lane and group in this kernel)
#ifndef ZKE_FFBL
__device__ __forceinline__ uint32_t zke_ffbl(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
#define ZKE_FFBL(x) zke_ffbl(x)
#endif
#ifndef ZKE_FFBH
// v_ffbh_u32: the number of leading zero bits, 0xFFFFFFFF for 0
__device__ __forceinline__ uint32_t zke_ffbh(uint32_t x) { uint32_t r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x)); return r; }
#define ZKE_FFBH(x) zke_ffbh(x)
#endif
// index of the first non-zero byte of the 16 bytes x0 .. x3 (16: none): -1 | 32 stays -1.
__device__ __forceinline__ uint32_t zke_first16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
    const uint32_t f0 = ZKE_FFBL(x0), f1 = ZKE_FFBL(x1) | 32u, f2 = ZKE_FFBL(x2) | 64u, f3 = ZKE_FFBL(x3) | 96u;
    // (a chain, not a tree: two v_min3_u32 instead of four v_min_u32)
    const uint32_t a = f0 < f1 ? f0 : f1, b = a < f2 ? a : f2, c = b < f3 ? b : f3;
    return (c < 128u ? c : 128u) >> 3;
}
// bytes (<= 16) that the 16 bytes at ring position c share with the 16 bytes o0 .. o3
__device__ __forceinline__ uint32_t zke_common16(const uint32_t *ring, uint32_t c, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3)
{
    const uint32_t i = (c >> 2) & 16383u, sh = c & 3u;
    const uint32_t a0 = ring[i], a1 = ring[i + 1], a2 = ring[i + 2], a3 = ring[i + 3], a4 = ring[i + 4];
    return zke_first16(o0 ^ __builtin_amdgcn_alignbyte(a1, a0, sh), o1 ^ __builtin_amdgcn_alignbyte(a2, a1, sh),
                       o2 ^ __builtin_amdgcn_alignbyte(a3, a2, sh), o3 ^ __builtin_amdgcn_alignbyte(a4, a3, sh));
}

// The 20 bytes x[0 .. 4] are a lane's own bytes xor the bytes some distance before them: out[k] = the number of zero bytes from byte
// k on (k = 0 .. 3, at most 16) = how far the match at that distance goes from the lane's position k.
__device__ __forceinline__ void zke_runs4(const uint32_t x[5], uint32_t out[4])
{
    const uint32_t g = zke_first16(x[1], x[2], x[3], x[4]) + 4;           // first non-zero byte at or behind byte 4 (20: none)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t w = k ? x[0] & (0xFFFFFFFFu << (8 * k)) : x[0];
        const uint32_t f = ZKE_FFBL(w) >> 3;                              // first non-zero byte of the first word at or behind byte k (none: huge)
        const uint32_t nz = f < g ? f : g;
        out[k] = nz - k < ZKE_PARCAP ? nz - k : ZKE_PARCAP;
    }
}

// Table forms.
// 16-bit entries, two per word (HLOG 15: what fits beside the ring): the low 16 bits of a position.  History: the numerically
// largest position wins a slot (positions < 65536, the empty entry 0 loses).  A step [ls, ls + span): the SMALLEST position
// of the step wins, any entry that is not of the step (entry - ls mod 2^16 >= span) loses.  The lanes race with
// compare-and-swap on the word that holds two entries.
// 32-bit entries (HLOG <= 14): (0xFFFF - step number) << 16 | (position + bias) mod 2^16, empty = 0xFFFFFFFF: an insertion is
// ONE ds_min_u32 -- the latest step wins, inside a step the smallest position; "of this step" is a test on the upper half.
// bias = -history mod 4096 aligns the steps so that none straddles a multiple of 2^16 (distances are unaffected).
// Both rules are order-free; oracle/zstd_oracle_enc.c applies them sequentially.
__device__ __forceinline__ bool zke_step_keeps(uint32_t word, uint32_t sh, uint32_t mine, uint32_t b16, uint32_t span)   // the entry in `word` beats position `mine`
{
    const uint32_t cur = (((word >> sh) & 0xFFFFu) - b16) & 0xFFFFu;
    return cur < span && cur <= ((mine - b16) & 0xFFFFu);
}
template <int MODE>
__device__ __forceinline__ void zke_table_put(uint32_t *table, uint32_t h, uint32_t p, uint32_t ls, uint32_t span, uint32_t old)
{
    uint32_t *w = &table[h >> 1];
    const uint32_t sh = 16 * (h & 1), mine = p & 0xFFFFu, b16 = ls & 0xFFFFu;
    for (;;) {
        if (MODE == 0) { if (((old >> sh) & 0xFFFFu) >= mine) return; }
        else if (zke_step_keeps(old, sh, mine, b16, span)) return;
        const uint32_t seen = atomicCAS(w, old, (old & ~(0xFFFFu << sh)) | (mine << sh));
        if (seen == old) return;
        old = seen;
    }
}

// The source dword at record position q (a multiple of 4, q < fend, fend >= 4); bytes at and past fend read as 0.  No branch:
// the last, ragged dword is cut out of the four bytes that end at fend.
__device__ __forceinline__ uint32_t zke_src_dword(const uint8_t *base, uint32_t q, uint32_t fend)
{
    const uint32_t over = q + 4 > fend ? q + 4 - fend : 0;        // 0 .. 3 bytes of the dword lie past the end
    uint32_t v;
    memcpy(&v, base + (q - over), 4);
    return v >> (8 * over);
}

__device__ __forceinline__ uint64_t zke_lowmask(uint32_t n) { return n >= 64 ? ~0ull : (1ull << n) - 1; }     // bits [0, n)
// value of lane (l - d) of the row of 16 lanes; `fill` where the row has no such lane
#define ZKE_ROW_SHR(v, d, fill) ((uint32_t)__builtin_amdgcn_update_dpp((int)(fill), (int)(v), 0x110 + (d), 0xF, 0xF, false))
// value of lane (l + d) of the row
// the same with 0 for a lane the row does not have: bound_ctrl supplies it, the destination needs no value of its own first
#define ZKE_ROW_SHR0(v, d) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (d), 0xF, 0xF, true))
#define ZKE_ROW_SHL0(v, d) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x100 + (d), 0xF, 0xF, true))
#define ZKE_ROW_SHL(v, d, fill) ((uint32_t)__builtin_amdgcn_update_dpp((int)(fill), (int)(v), 0x100 + (d), 0xF, 0xF, false))

// own[0..4] ^ the 20 bytes at prefix byte a (long-distance compare of a lane's four positions through HBM).  ok: the 20 bytes
// lie inside the prefix; a lane without reads the prefix's last 20 bytes, its flag drops the result.
__device__ __forceinline__ void zke_far_xor(const ZkEncLdm &ldm, int64_t a, bool ok, const uint32_t own[5], uint32_t x[5])
{
    const uint8_t *s = ldm.pfx + (ok ? a : (int64_t)ldm.plen - 20);
#pragma unroll
    for (int i = 0; i < 5; i++) { uint32_t v; memcpy(&v, s + 4 * i, 4); x[i] = own[i] ^ v; }
}

// HLOG: log2 of the table entries (<= 14: 32-bit entries, 15: 16-bit entries); LAZY: a longer match one (or a clearly longer
// one two) positions later wins; STEP: positions per lookup step (4096 = the whole group at once, 1024 = four steps of 256
// lanes each)
// LDM: long-distance candidates out of a prefix (zk_enc_device.h ZkEncLdm); DENSE: ... and the dense far history of the frame (round 6)
template <int HLOG, int LAZY, int STEP, bool LDM, bool DENSE = false>
__global__ __launch_bounds__(ZKE_THREADS) void zk_k_enc_match(const uint8_t *src, const ZkEncFrame *segs, ZkEncBlock *blocks, uint64_t *seqs, uint8_t *lits, ZkEncLdm ldm)
{
    static_assert(HLOG >= 10 && HLOG <= 15 && (STEP == 1024 || STEP == (int)ZKE_GROUP_POS) && (LDM || !DENSE), "parameters");
    constexpr bool T32 = HLOG <= 14;
    constexpr uint32_t TWORDS = T32 ? (1u << HLOG) : (1u << (HLOG - 1)), DUMMY = TWORDS;   // table[DUMMY]: where lanes without a position read and write
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    __shared__ uint32_t ring[ZKE_RING_WORDS + 8];
    __shared__ uint32_t table[TWORDS + 1];
    __shared__ uint32_t best[ZKE_GROUP_POS];
    __shared__ uint64_t tseq[ZKE_GROUP][ZKE_TSEQ_N];        // ll | ml << 12 in the low half, the offset in the high half
    __shared__ uint32_t bkb[ZKE_GROUP_POS / 4];             // (round 5) per position: how many of the bytes in front of it agree at its best candidate's offset (<= 4): catch-up
    __shared__ uint32_t tsum[2][ZKE_GROUP], tlast[2][ZKE_GROUP], tfirst[2][ZKE_GROUP], tfml[2][ZKE_GROUP];   // per tile (two groups deep): count | trailing literals << 8 | literal bytes << 20;  offset of its last sequence
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const ZkEncFrame fr = segs[blockIdx.x];
    const uint8_t *base = src + fr.m_off;
    const uint32_t hist = fr.hist, fend = hist + fr.d_size, minmatch = fr.minmatch, fend4 = (fend + 3) & ~3u;
    const uint64_t lane_lt = zke_lowmask(lane);
    const uint32_t bias = T32 ? (0u - hist) & (ZKE_GROUP_POS - 1) : 0u;
    // record position p = byte abs0 + p of [prefix | frame]; in-frame far history: of the frame itself, and LD is the frame's view
    // of the table (its own 2^log entries, its bytes as the "prefix", its size as the limit)
    const uint64_t abs0 = !LDM ? 0 : ldm.inframe ? fr.seg_at - hist : ldm.plen + fr.seg_at - hist;
    ZkEncLdm LD = ldm;
    if (LDM && ldm.inframe) {
        const uint64_t fsz = ldm.n_total - fr.src_off < ldm.frame_size ? ldm.n_total - fr.src_off : ldm.frame_size;
        LD.pfx = base - abs0; LD.plen = fsz; LD.u0 = 0; LD.log = zke_ldm_log(fsz);
        LD.table = ldm.table + ((fr.src_off / ldm.frame_size) << ldm.log);
    }
    // dense far history (round 6): per position of my segment its far candidate, length | catch-up << 5 | distance << 8, as zk_k_enc_dense_cand left it
    // (0: none); four positions per lane and group in one 16-byte read
    const uint32_t *dcand = DENSE ? ldm.dense + fr.src_off + fr.seg_at : nullptr;

    ZKE_CLK_BEGIN();
    // ---- segment start: empty table, history + the first group (+ lookahead) into the ring, history positions into the table
    for (uint32_t i = tid; i <= TWORDS; i += ZKE_THREADS) table[i] = T32 ? NONE : 0u;
    uint32_t loaded = hist + ZKE_GROUP_POS + 64;             // the ring holds the record up to here (or to its end)
    if (loaded > fend4) loaded = fend4;
    if (fend >= 4) { for (uint32_t q = 4 * tid; q < loaded; q += 4 * ZKE_THREADS) zke_ring_put(ring, q >> 2, zke_src_dword(base, q, fend)); }
    else if (tid == 0) { uint32_t v = 0; for (uint32_t k = 0; k < fend; k++) v |= (uint32_t)base[k] << (8 * k); zke_ring_put(ring, 0, v); }
    __syncthreads();
    uint32_t stepno = 0;                                     // steps of the segment so far (32-bit entries carry it)
    if (T32) {
        // history in steps of 4096 positions under the step rule; the keys order the steps, so no barrier separates them
        for (uint32_t c0 = 0; c0 < hist; c0 += ZKE_GROUP_POS, stepno++) {
            const uint32_t V0 = c0 + 4 * tid, i0 = (V0 >> 2) & 16383u;
            const uint32_t d0 = ring[i0], d1 = ring[i0 + 1], d2 = ring[i0 + 2];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t v = V0 + k, lo = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)k), hi = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)k);
                const bool ok = v < hist && v + 8 <= fend;
                atomicMin(&table[ok ? zke_hash(lo, hi & 0xFF, HLOG) : DUMMY], ok ? ((0xFFFFu - stepno) << 16) | ((v + bias) & 0xFFFFu) : NONE);
            }
        }
    } else {
        for (uint32_t v = tid; v < hist; v += ZKE_THREADS)
            if (v + 8 <= fend) {
                const uint64_t w = zke_ring8(ring, v);
                const uint32_t h = zke_hash((uint32_t)w, (uint32_t)(w >> 32) & 0xFF, HLOG);
                zke_table_put<0>(table, h, v, 0, 0, table[h >> 1]);
            }
    }
    __syncthreads();

    ZKE_CLK(0);
    uint32_t probe = 0;                                      // offset of the last sequence so far (same value in every lane)
    uint32_t nseq = 0, nlit = 0, pend = 0, prev_off = 0;     // parse state of the block being stitched, identical in every lane
    uint32_t par = 0;                                        // which half of tsum / tlast the group at work writes
    // A group's tiles are stitched and stored one group LATER (by the wave that parsed the tile, before it touches its slices
    // of best[] / tseq[] again): what the stitch needs from the other waves -- their tile summaries -- is then two barriers
    // old, so neither the parse nor the stitch ends in a barrier of its own and a wave that is slow in one group's parse
    // catches up in the next group's lookups.
    struct { bool valid; uint32_t gs, ge, rel; ZkEncBlock *blk; uint64_t *sq; uint8_t *lt; bool last; } todo = {false, 0, 0, 0, nullptr, nullptr, nullptr, false};
    // A sequence whose match ends exactly where its group ends may go on in the next group (see "seams" below): the lane that
    // owns it keeps it back until the stitch that knows
    bool held = false; uint64_t held_e = 0; uint64_t *held_at = nullptr;
    auto stitch = [&]() {
        const uint32_t ntiles = (todo.ge - todo.gs + ZKE_TILE - 1) / ZKE_TILE;
        const uint32_t *ts_ = tsum[par ^ 1], *tl_ = tlast[par ^ 1], *tf_ = tfirst[par ^ 1];
        uint32_t my_base, my_lit, my_pend, my_poff, my_cnt, my_nl, my_join, my_more, me;
        bool my_open;
        {
            // lane t of every row of 16 lanes takes tile t's summary; inclusive scans along the row:
            //   literals pending behind tile t      tail(t) if the tile has sequences, else pending(t - 1) + tail(t): segmented sum, bit 31 = "a tile with sequences is inside"
            //   offset of the last sequence so far  the last non-zero value
            // SEAMS.  A tile's matches stop at its end.  Where the next tile opens with a match of the same offset at its first
            // byte, the two are one sequence (`join`); a tile that is nothing but such a match is `whole` and hands on what
            // follows it.  Not across multiples of ZKE_SEAM inside the block: a match length stays below 2^16.
            //   counts and literal bytes            plain sums (count - join | bytes << 16)
            //   what the tiles from t on add to the sequence in front of t     join(t) ? first_ml(t) + (whole(t) ? more(t + 1) : 0) : 0   (a segmented sum from the right)
            const uint32_t t = lane & 15;
            const uint32_t sv = t < ntiles ? ts_[t] : 0, lv = t < ntiles ? tl_[t] : 0, fo = t < ntiles ? tf_[t] : 0, fm = tfml[par ^ 1][t & 15];
            const uint32_t cn = sv & 0xFF, tail = (sv >> 8) & 0xFFF, tnl = sv >> 20;
            uint32_t y = tail | (cn ? 0x80000000u : 0u), z = cn ? lv : 0;
#define ZKE_SCAN_STEP(d) { const uint32_t ys = ZKE_ROW_SHR0(y, d), zs = ZKE_ROW_SHR0(z, d); \
                           y = (y & 0x80000000u) ? y : (y + (ys & 0x7FFFFFFFu)) | (ys & 0x80000000u); z = z ? z : zs; }
            ZKE_SCAN_STEP(1) ZKE_SCAN_STEP(2) ZKE_SCAN_STEP(4) ZKE_SCAN_STEP(8)
#undef ZKE_SCAN_STEP
            const uint32_t ye = ZKE_ROW_SHR0(y, 1), ze = ZKE_ROW_SHR0(z, 1);            // what lies in front of tile t
            const uint32_t pend_t = (ye & 0x80000000u) ? ye & 0x7FFFFFFFu : pend + ye, poff_t = ze ? ze : prev_off;
            const bool join = fo && fo == poff_t && pend_t == 0 && ((todo.rel + t * ZKE_TILE) & (ZKE_SEAM - 1)) != 0;
            uint32_t x = (cn - (join ? 1u : 0u)) | (tnl << 16);
#define ZKE_SCAN_STEP(d) { x += ZKE_ROW_SHR0(x, d); }
            ZKE_SCAN_STEP(1) ZKE_SCAN_STEP(2) ZKE_SCAN_STEP(4) ZKE_SCAN_STEP(8)
#undef ZKE_SCAN_STEP
            // what is in front of my tile = the scans at lane wave - 1; the group's totals = lane 15
            const uint32_t wm = wave ? wave - 1 : 0, wn = wave < 15 ? wave + 1 : 15;
            const uint32_t xp = wave ? (uint32_t)__builtin_amdgcn_readlane((int)x, (int)wm) : 0, yp = wave ? (uint32_t)__builtin_amdgcn_readlane((int)y, (int)wm) : 0,
                           zp = wave ? (uint32_t)__builtin_amdgcn_readlane((int)z, (int)wm) : 0;
            me = (uint32_t)__builtin_amdgcn_readlane((int)sv, (int)wave);
            const uint32_t xt = (uint32_t)__builtin_amdgcn_readlane((int)x, 15), yt = (uint32_t)__builtin_amdgcn_readlane((int)y, 15), zt = (uint32_t)__builtin_amdgcn_readlane((int)z, 15);
            my_join = 0; my_more = 0; my_open = wave == 15;
            uint32_t more0 = 0, whole0 = 0;
            if (__ballot(join)) {                                                   // (most groups have no seam to close)
                uint32_t ev = join ? fm : 0, pw = join && cn == 1 && tail == 0 ? 1u : 0u;
#define ZKE_SCAN_STEP(d) { const uint32_t es = ZKE_ROW_SHL0(ev, d), ps = ZKE_ROW_SHL(pw, d, 1); ev += pw ? es : 0; pw &= ps; }
                ZKE_SCAN_STEP(1) ZKE_SCAN_STEP(2) ZKE_SCAN_STEP(4) ZKE_SCAN_STEP(8)
#undef ZKE_SCAN_STEP
                my_join = (uint32_t)__builtin_amdgcn_readlane((int)(join ? 1u : 0u), (int)wave);
                my_more = wave < 15 ? (uint32_t)__builtin_amdgcn_readlane((int)ev, (int)wn) : 0;
                my_open = wave == 15 || __builtin_amdgcn_readlane((int)pw, (int)wn) != 0;   // every tile behind mine is whole
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
            if (lane >= my_join && lane < my_cnt) {                                 // <= 64 sequences per tile: one per lane
                const uint64_t e = tseq[wave][lane];
                uint32_t ll = (uint32_t)e & 0xFFF, ml = (uint32_t)(e >> 12) & 0xFFF;
                const uint32_t off = (uint32_t)(e >> 32);
                const uint32_t poff = lane ? (uint32_t)(tseq[wave][lane - 1] >> 32) : my_poff;
                if (lane == 0) ll += my_pend;
                const uint32_t code = (ll && off == poff) ? 1u : off + 3;
                const bool ends_tile = lane + 1 == my_cnt && ((me >> 8) & 0xFFF) == 0;
                if (ends_tile) ml += my_more;
                const uint64_t rec = (uint64_t)(ll | (ml << 16)) | ((uint64_t)code << 32);
                uint64_t *at = &todo.sq[my_base + lane - my_join];
                if (ends_tile && my_open) { held = true; held_e = rec; held_at = at; }
#ifndef ZKE_KNOCK_STORES
                else *at = rec;
#endif
            }
            // the tile's literals: four bytes per lane (an unaligned dword store), the last bytes one by one
            const uint32_t *tw4 = &best[wave * ZKE_TILE];
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
            // the next group's input: requested now, stored into the ring behind this group's first barrier, while only the
            // insertions run (nobody reads the ring then; what it overwrites lies in front of this group's window).  A lane
            // past the target repeats the last dword (no branch, unused).
            uint32_t target = ge + ZKE_GROUP_POS + 64;
            if (target > fend4) target = fend4;
            const uint32_t pq = loaded + 4 * tid;
            // (the dword is only REQUESTED here: what zke_src_dword does with it -- the shift of a ragged last dword -- waits until the
            // ring write, or the wait for the load would sit right here, at the top of every group)
            // No branch around it either (a record of fewer than 4 bytes reads the segment list instead: unused).
            const uint32_t pq_ = pq < target ? pq : fend4 - 4;
            const uint32_t pv_over = fend >= 4 && pq_ + 4 > fend ? pq_ + 4 - fend : 0;
            const uint8_t *pv_at = fend >= 4 ? base + (pq_ - pv_over) : (const uint8_t *)segs;
            uint32_t pv_raw;
            memcpy(&pv_raw, pv_at, 4);

            // ---- 1 + 2: lookups and insertions, step by step
            const uint32_t P0 = gs + 4 * tid;                                       // my four positions: P0 .. P0 + 3, tile = wave
            const uint32_t i0 = (P0 >> 2) & 16383u;
            const uint32_t dm1 = ring[(i0 - 1) & 16383u], d0 = ring[i0], d1 = ring[i0 + 1], d2 = ring[i0 + 2], d3 = ring[i0 + 3], d4 = ring[i0 + 4];
            uint32_t wlo[4], whi[4], hsh[4], tix[4], tw[4], e1[4];
            bool tabled[4];
            uint32_t dcv[4] = {0, 0, 0, 0};
            if (DENSE) memcpy(dcv, dcand + (P0 - hist), 16);                            // (lanes past the segment's end read on into the slack: unused)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wlo[k] = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)k);
                whi[k] = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)k);
                const uint32_t p = P0 + k;
                tabled[k] = p < ge && p + 8 <= fend;
                hsh[k] = tabled[k] ? zke_hash(wlo[k], whi[k] & 0xFF, HLOG) : NONE;
                tix[k] = tabled[k] ? (T32 ? hsh[k] : hsh[k] >> 1) : DUMMY;             // the table word of the position
                tw[k] = 0; e1[k] = 0;
            }
            constexpr uint32_t LPS = STEP / 4;                                      // lanes per step
            const uint32_t mystep = tid / LPS;
#pragma unroll
            for (uint32_t s = 0; s < ZKE_GROUP_POS / STEP; s++) {
                const uint32_t ls = gs + s * STEP, le = ls + STEP < ge ? ls + STEP : ge;
                if (ls < ge) {                                                     // uniform
                    if (STEP == (int)ZKE_GROUP_POS || mystep == s) {
#pragma unroll
                        for (int k = 0; k < 4; k++) tw[k] = table[tix[k]];             // far candidates; with 16-bit entries also what the insertion swaps against
                    }
                    ZKE_CLK(1);
                    ZKE_LDS_BARRIER();
                    ZKE_CLK(2);
                    if (s == 0) { if (pq < target) zke_ring_put(ring, (pq >> 2) & 16383u, pv_raw >> (8 * pv_over)); loaded = target; }
                    // A position whose hash also belongs to one of the four positions before it cannot win its slot (the smaller
                    // position does): it stays out of the race.  On runs of equal bytes or short periods all lanes of a step
                    // would otherwise fight over a few LDS words.  The table ends the step in the same state.
                    const uint32_t q0 = ZKE_ROW_SHR(hsh[0], 1, NONE), q1 = ZKE_ROW_SHR(hsh[1], 1, NONE), q2 = ZKE_ROW_SHR(hsh[2], 1, NONE), q3 = ZKE_ROW_SHR(hsh[3], 1, NONE);
                    bool go[4];
                    go[0] = tabled[0] && hsh[0] != q3 && hsh[0] != q2 && hsh[0] != q1 && hsh[0] != q0;
                    go[1] = tabled[1] && hsh[1] != hsh[0] && hsh[1] != q3 && hsh[1] != q2 && hsh[1] != q1;
                    go[2] = tabled[2] && hsh[2] != hsh[1] && hsh[2] != hsh[0] && hsh[2] != q3 && hsh[2] != q2;
                    go[3] = tabled[3] && hsh[3] != hsh[2] && hsh[3] != hsh[1] && hsh[3] != hsh[0] && hsh[3] != q3;
                    if (T32) {
                        const uint32_t khi = (0xFFFFu - stepno) << 16;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool g = go[k] && (STEP == (int)ZKE_GROUP_POS || mystep == s);
                            atomicMin(&table[g ? tix[k] : DUMMY], g ? khi | ((P0 + k + bias) & 0xFFFFu) : NONE);
                        }
                    } else if (mystep == s) {
                        // the four swaps leave together (against the words read before the barrier); what failed -- a neighbour
                        // was faster on the same word -- goes round again on its own
                        const uint32_t span = le - ls, b16 = ls & 0xFFFFu;
                        uint32_t seen[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t sh = 16 * (hsh[k] & 1), mine = (P0 + k) & 0xFFFFu;
                            go[k] = go[k] && !zke_step_keeps(tw[k], sh, mine, b16, span);
                            seen[k] = tw[k];
                            if (go[k]) seen[k] = atomicCAS(&table[tix[k]], tw[k], (tw[k] & ~(0xFFFFu << sh)) | (mine << sh));
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) if (go[k] && seen[k] != tw[k]) zke_table_put<1>(table, hsh[k], P0 + k, ls, span, seen[k]);
                    }
                    ZKE_CLK(3);
                    ZKE_LDS_BARRIER();
                    ZKE_CLK(4);
                    if (STEP == (int)ZKE_GROUP_POS || mystep == s) {
#pragma unroll
                        for (int k = 0; k < 4; k++) e1[k] = table[tix[k]];             // near candidates: what the step left in my slots
                    }
                    stepno++;
                }
            }

            // ---- the group before this one: stitch its tiles, store my tile's sequences and literals
            if (todo.valid) stitch();
            ZKE_WAVE_SYNC();
            const uint32_t R = probe;
            ZKE_CLK(8);
            // ---- 3a: comparisons out of the ring -> best[].  16 bytes per candidate, no branches: an invalid candidate
            // compares the position with itself and is dropped by its flag, so all ring reads of the lane are in flight together.
            const uint32_t ts = gs + wave * ZKE_TILE, te = ts + ZKE_TILE < ge ? ts + ZKE_TILE : ge;      // my tile
            {
                const uint32_t ls = gs + mystep * STEP;
                const uint32_t khi = (0xFFFFu - (stepno - ZKE_GROUP_POS / STEP + mystep)) << 16;       // the key half of my step (32-bit entries)
                // the 20 bytes of my four positions against the 20 bytes one byte / R bytes before them, once for all four
                const uint32_t own[5] = {d0, d1, d2, d3, d4};
                uint32_t x1[5], xr[5];                                                  // own ^ (own one byte earlier), own ^ (own R bytes earlier)
                const bool farR = LDM && R > ZKE_WINDOW;                                // a previous offset beyond the ring: through memory
                const int64_t ap0 = (int64_t)(abs0 + P0);
                // a long-distance offset is usable at my four positions while the 20 bytes I read at that distance lie inside the
                // table's part of the prefix
                auto far_ok = [&](uint32_t off) { const int64_t a = ap0 - (int64_t)off; return a >= (int64_t)LD.u0 && a + 20 <= (int64_t)LD.plen; };
                const bool okR = farR && far_ok(R);
                // dense far history: what zk_k_enc_dense_cand found for my four positions (requested at the top of the group)
                uint32_t dpk[4] = {0, 0, 0, 0};
                if (DENSE) {
#pragma unroll
                    for (int k = 0; k < 4; k++) dpk[k] = tabled[k] ? dcv[k] : 0u;
                }
                {
                    const uint32_t rb = P0 - R, ri = (rb >> 2) & 16383u, rs = rb & 3u;  // R <= P0 is tested below; a wrong address reads some ring bytes
                    x1[0] = d0 ^ __builtin_amdgcn_alignbyte(d0, dm1, 3u); x1[1] = d1 ^ __builtin_amdgcn_alignbyte(d1, d0, 3u); x1[2] = d2 ^ __builtin_amdgcn_alignbyte(d2, d1, 3u);
                    x1[3] = d3 ^ __builtin_amdgcn_alignbyte(d3, d2, 3u); x1[4] = d4 ^ __builtin_amdgcn_alignbyte(d4, d3, 3u);
                    if (farR) zke_far_xor(LD, ap0 - (int64_t)R, okR, own, xr);
                    else {
                        const uint32_t r0 = ring[ri], r1 = ring[ri + 1], r2 = ring[ri + 2], r3 = ring[ri + 3], r4 = ring[ri + 4], r5 = ring[ri + 5];
                        xr[0] = d0 ^ __builtin_amdgcn_alignbyte(r1, r0, rs); xr[1] = d1 ^ __builtin_amdgcn_alignbyte(r2, r1, rs); xr[2] = d2 ^ __builtin_amdgcn_alignbyte(r3, r2, rs);
                        xr[3] = d3 ^ __builtin_amdgcn_alignbyte(r4, r3, rs); xr[4] = d4 ^ __builtin_amdgcn_alignbyte(r5, r4, rs);
                    }
                }
                uint32_t lf[4], ln[4], df[4], dn[4], wl[4], wo[4];              // ... and every position's winner: length, offset
                bool vf[4], vn[4];
                const bool vr0 = R > 1;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t p = P0 + k;
                    const uint32_t ef = T32 ? tw[k] & 0xFFFFu : (tw[k] >> (16 * (hsh[k] & 1))) & 0xFFFFu;
                    const uint32_t en = T32 ? e1[k] & 0xFFFFu : (e1[k] >> (16 * (hsh[k] & 1))) & 0xFFFFu;
                    df[k] = (p + bias - ef) & 0xFFFFu;                                  // far: what the table held before the step
                    vf[k] = tabled[k] && df[k] && df[k] <= p && df[k] <= ZKE_WINDOW;
                    dn[k] = (p + bias - en) & 0xFFFFu;                                  // near: an earlier position of this step
                    vn[k] = tabled[k] && dn[k] && (T32 ? (e1[k] & 0xFFFF0000u) == khi : dn[k] <= p - ls);
                    const uint32_t o0 = wlo[k], o1 = whi[k], o2 = __builtin_amdgcn_alignbyte(own[3], own[2], (uint32_t)k), o3 = __builtin_amdgcn_alignbyte(own[4], own[3], (uint32_t)k);
                    lf[k] = zke_common16(ring, vf[k] ? p - df[k] : p, o0, o1, o2, o3);
                    ln[k] = zke_common16(ring, vn[k] ? p - dn[k] : p, o0, o1, o2, o3);
                }
                // long-distance candidates: my sampled positions against their table entries; the tile's first hit lends its
                // offset to every position of the tile (x2)
                uint32_t hoff[4] = {0, 0, 0, 0}, x2[5] = {~0u, ~0u, ~0u, ~0u, ~0u}, tfar = 0;
                bool ok2 = false;
                if (LDM) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t p = P0 + k;
                        const uint32_t o0 = wlo[k], o1 = whi[k], o2 = __builtin_amdgcn_alignbyte(own[3], own[2], (uint32_t)k), o3 = __builtin_amdgcn_alignbyte(own[4], own[3], (uint32_t)k);
                        const uint32_t h = zke_ldm_hash(o0, o1, o2, o3);
                        if (zke_ldm_selected(h) && p + ZKE_LDM_MIN <= te && p + ZKE_LDM_MIN <= fend) {
                            const uint32_t e = LD.table[zke_ldm_slot(h, LD.log)];
                            const uint64_t q = LD.u0 + e, ap = abs0 + p;
                            if (e != ZKE_LDM_NONE && q + 16 <= LD.plen && ap - q <= ZKE_LDM_MAX_OFF && (!ldm.inframe || (q < ap && ap - q > ZKE_WINDOW))) {   // in-frame: behind me, beyond the ring
                                uint32_t c[4];
                                memcpy(c, LD.pfx + q, 16);
                                if (c[0] == o0 && c[1] == o1 && c[2] == o2 && c[3] == o3) hoff[k] = (uint32_t)(ap - q);
                            }
                        }
                    }
                    const uint32_t first = hoff[0] ? hoff[0] : hoff[1] ? hoff[1] : hoff[2] ? hoff[2] : hoff[3];
                    const uint64_t hits = __ballot(first != 0);
                    if (hits) tfar = (uint32_t)__builtin_amdgcn_readlane((int)first, (int)__builtin_ctzll(hits));
                    if (tfar && tfar != R) { ok2 = far_ok(tfar); zke_far_xor(LD, ap0 - (int64_t)tfar, ok2, own, x2); }      // uniform per wave
                    else tfar = 0;
                }
                // offset 1 and offset R: the run of equal bytes from each of my four positions out of the 20-byte windows, once per window
                uint32_t r1[4], rr[4];
                zke_runs4(x1, r1); zke_runs4(xr, rr);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t p = P0 + k;
                    const uint32_t n = p < te ? (te - p < ZKE_PARCAP ? te - p : ZKE_PARCAP) : 0;   // a match may not leave the tile
                    const uint32_t cf = lf[k] < n ? lf[k] : n, cn = ln[k] < n ? ln[k] : n, c1 = r1[k] < n ? r1[k] : n, cr = rr[k] < n ? rr[k] : n;
                    const bool okf = vf[k] && cf >= minmatch, okn = vn[k] && cn >= minmatch, ok1 = p >= 1 && c1 >= 4,
                               okr = vr0 && (farR ? okR : R <= p) && cr >= 4;
                    if (!LDM) {
                        // the longest wins, on ties the later of far, near, offset 1, R: one key per candidate (length | rank | offset), the largest key
                        const uint32_t kf = okf ? (cf << 18) | df[k] : 0u, kn = okn ? (cn << 18) | (1u << 16) | dn[k] : 0u,
                                       k1 = ok1 ? (c1 << 18) | (2u << 16) | 1u : 0u, kr = okr ? ((cr + 1) << 18) | (3u << 16) | R : 0u;    // (round 5: the previous offset is cheap to code, it also wins one byte short)
                        uint32_t m = kf > kn ? kf : kn;
                        m = m > k1 ? m : k1; m = m > kr ? m : kr;
                        if (m == kr && okr) m -= 1u << 18;
                        best[4 * tid + k] = (m >> 18) | ((m & 0xFFFFu) << 5);           // positions past the tile's end: length 0
                        wl[k] = m >> 18; wo[k] = m & 0xFFFFu;
                    } else {
                        uint32_t bl = 0, bo = 0;
                        if (okf) { bl = cf; bo = df[k]; }
                        if (okn && cn >= bl) { bl = cn; bo = dn[k]; }
                        const bool fills = !ldm.inframe || bl < ZKE_LDM_FILL;               // in frame a far candidate only fills gaps
                        if (hoff[k] && fills) { bl = ZKE_PARCAP; bo = hoff[k]; }            // (16 equal bytes inside the tile)
                        const uint32_t l2 = zke_first16(__builtin_amdgcn_alignbyte(x2[1], x2[0], (uint32_t)k), __builtin_amdgcn_alignbyte(x2[2], x2[1], (uint32_t)k),
                                                        __builtin_amdgcn_alignbyte(x2[3], x2[2], (uint32_t)k), __builtin_amdgcn_alignbyte(x2[4], x2[3], (uint32_t)k));
                        if (tfar && ok2 && tabled[k] && n == ZKE_PARCAP && l2 == ZKE_PARCAP && (!ldm.inframe || bl < ZKE_LDM_FILL)) { bl = ZKE_PARCAP; bo = tfar; }
                        if (DENSE) {
                            const uint32_t cd = (dpk[k] & 31u) < n ? dpk[k] & 31u : n;
                            if ((dpk[k] >> 8) && cd >= ZKE_DENSE_MIN && cd >= bl + ZKE_DENSE_MARGIN) { bl = cd; bo = dpk[k] >> 8; }
                        }
                        if (ok1 && c1 >= bl) { bl = c1; bo = 1; }
                        if (okr && cr + 1 >= bl) { bl = cr; bo = R; }
                        best[4 * tid + k] = bl | (bo << 5);
                        wl[k] = bl; wo[k] = bo;
                    }
                }
                // (round 5) catch-up: how many of the four bytes in front of each position agree with the four bytes in front of its
                // winner's source -- out of the ring, which holds them while offset + 4 <= ZKE_WINDOW (a source in the prefix or far back in
                // the frame: none); not past the tile's start, not before the record's first byte.  The parse cuts it down to the literals
                // the match really has in front of it (zk_enc_match2.h has the measurements: the stale table finds many a match late).
                uint32_t bk4 = 0;
                if (DENSE) {
                    // which of my four far winners give way to a cheap offset (the previous one, or 1) up to four positions later: the cheap
                    // winners' lengths of my positions and of the next lane's (the tile's positions lie four to a lane along the wave), five bits each
                    uint32_t cp = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) cp |= ((wo[k] == R || wo[k] == 1u) ? wl[k] : 0u) << (5 * k);
                    const uint32_t nx_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 1) & 63u) << 2), (int)cp);
                    const uint64_t both = (uint64_t)cp | ((uint64_t)(lane == 63 ? 0u : nx_) << 20);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t m = 0;
#pragma unroll
                        for (uint32_t j = 1; j <= ZKE_DENSE_AHEAD; j++) { const uint32_t c = (uint32_t)(both >> (5 * (k + j))) & 31u, v = c ? c + j : 0u; m = m > v ? m : v; }
                        if (wl[k] && wo[k] > ZKE_WINDOW && wo[k] != R && m + ZKE_DENSE_BONUS >= wl[k]) bk4 |= 0x80u << (8 * k);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t p = P0 + k, off = wo[k];
                    const uint32_t cb = p - off - 4, ci = (cb >> 2) & 16383u;
                    const uint32_t b0 = ring[ci], b1 = ring[ci + 1];
                    const uint32_t mine4 = k ? __builtin_amdgcn_alignbyte(d0, dm1, (uint32_t)k) : dm1;          // the four bytes that end at p
                    uint32_t bk = ZKE_FFBH(mine4 ^ __builtin_amdgcn_alignbyte(b1, b0, cb & 3u));
                    bk = (bk < 32u ? bk : 32u) >> 3;
                    const uint32_t room = p - ts < p - off ? p - ts : p - off;
                    bk = bk < room ? bk : room;
                    if (!wl[k] || off + 4 > ZKE_WINDOW || p < ts) bk = 0;
                    // (round 6) a winner at the distance of the position's dense candidate: the count its entry carries (through memory, zk_k_enc_dense_cand)
                    if (DENSE && wl[k] && (dpk[k] >> 8) && off == (dpk[k] >> 8) && p >= ts) { bk = (dpk[k] >> 5) & 7u; bk = bk < p - ts ? bk : p - ts; }
                    bk4 |= bk << (8 * k);
                }
                bkb[tid] = bk4;
            }
            ZKE_CLK(5);
            // ---- 3b: wave w parses tile w (it wrote that slice of best[] itself: LDS operations of a wave complete in order)
            ZKE_WAVE_SYNC();
            if (ts < ge) {                                                              // uniform per wave
                uint8_t *tl = (uint8_t *)&best[wave * ZKE_TILE];                        // the tile's literal bytes, behind the entries already read
                uint32_t skip = 0, c = 0, nl = 0, aend = 0, lastoff = 0;                // tile-relative: first position not covered yet; sequences; literals; end / offset of the last match
                // The tile in four passes of 64 positions, in three sweeps, so that what does not depend on the walk -- the LDS reads,
                // the ballots, the emission -- is not strung between the walks of consecutive passes:
                //   A  every pass: best[] entries, candidate masks, per-lane next-candidate table
                //   B  the walk, pass after pass (the only serial part: where a match ends decides which candidate is next)
                //   C  every pass: sequences and literal bytes, all lanes at once
                // (a pass past the tile's end has no candidates and no literals: best[] holds length 0 there)
                uint32_t pv[4], plen[4], pnx[4], pbk[4], ppe[4], pst[4], first_s[5];
                uint64_t pcand[4], pcap[4], ptaken[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t pos = 64 * u + lane, p = ts + pos;
                    const uint32_t v = best[wave * ZKE_TILE + pos];                     // length 0 past the tile's end
                    const uint32_t len = v & 0x1F;
                    bool cand = len != 0;
                    if (LAZY) {
                        const uint32_t n1 = best[wave * ZKE_TILE + pos + 1 < ZKE_GROUP_POS ? wave * ZKE_TILE + pos + 1 : 0], l1 = n1 & 0x1F;
                        const uint32_t l2 = best[wave * ZKE_TILE + pos + 2 < ZKE_GROUP_POS ? wave * ZKE_TILE + pos + 2 : 0] & 0x1F;
                        if ((p + 1 < te && l1 > len) || (p + 2 < te && l2 > len + 1)) cand = false;
                        // (round 5) a CHEAP offset one position later -- the previous offset or offset 1 -- wins against a fresh one even one byte shorter
                        // (the twin has the measurements: runs of 10 bytes 5.5 -> 8.8)
                        if ((v >> 5) != R && (v >> 5) != 1 && p + 1 < te && l1 && ((n1 >> 5) == R || (n1 >> 5) == 1) && l1 + 1 >= len) cand = false;
                        // (round 6) ... and a FAR candidate of the dense tables gives way to a cheap offset up to ZKE_DENSE_AHEAD positions later
                        // unless it is ZKE_DENSE_BONUS bytes longer still (the twin has the cases: records, byte runs cut by a tile's end)
                        // (round 6) ... and a FAR candidate of the dense tables gives way to a cheap offset up to ZKE_DENSE_AHEAD positions later
                        // unless it is ZKE_DENSE_BONUS bytes longer still (the twin has the cases: records, byte runs cut by a tile's end): decided
                        // where the candidates are made (3a), bit 7 of the position's catch-up byte -- four more LDS reads and compares per
                        // position HERE were 7 000 of a tile's 18 000 parse clocks (profiles/r06c_dense_probe.txt)
                        if (DENSE && (((const uint8_t *)bkb)[wave * ZKE_TILE + pos] & 0x80u)) cand = false;
                    }
                    pv[u] = v; plen[u] = len;
                    pbk[u] = ((const uint8_t *)bkb)[wave * ZKE_TILE + pos] & 7u;
                    pcand[u] = __ballot(cand); pcap[u] = __ballot(cand && len == ZKE_PARCAP);
                    // every lane: the first candidate at or behind the end of its own match (64: none in this pass); the walk
                    // below then costs the scalar unit a handful of instructions per match
                    const uint32_t el = lane + len;
                    const uint64_t behind = pcand[u] & (el >= 64 ? 0ull : ~0ull << el);
                    pnx[u] = behind ? (uint32_t)__builtin_ctzll(behind) : 64u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t wb = 64 * u;
                    const uint64_t candm = pcand[u], capped = pcap[u];
                    const uint32_t v = pv[u], nx = pnx[u];
                    uint32_t len = plen[u];
                    const uint32_t pre = skip > wb ? skip - wb : 0;
                    const uint64_t open = candm & (pre >= 64 ? 0ull : ~0ull << pre);
                    uint32_t f = open ? (uint32_t)__builtin_ctzll(open) : 64u, lastf = 64;
                    uint64_t taken = 0;
                    while (f < 64) {                                                    // uniform: every lane walks the same chain
                        taken |= 1ull << f; lastf = f;
                        if ((capped >> f) & 1) {                                        // capped by the comparisons: extend, 64 bytes per step
                            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)f) >> 5;
                            uint32_t L = ZKE_PARCAP;
                            if (LDM && off > ZKE_WINDOW) {                                  // the source lies in the prefix: byte by byte through memory, up to the prefix's end
                                const uint64_t a0 = abs0 + ts + wb + f - off;
                                for (;;) {
                                    const uint32_t q = ts + wb + f + L + lane;
                                    const uint64_t sa = a0 + L + lane;
                                    const bool diff = q >= te || sa >= LD.plen || zke_ring1(ring, q) != LD.pfx[sa < LD.plen ? sa : LD.u0];
                                    const uint64_t dm = __ballot(diff);
                                    if (dm) { L += (uint32_t)__builtin_ctzll(dm); break; }
                                    L += 64;
                                }
                            } else
                            for (;;) {
                                const uint32_t q = ts + wb + f + L + lane;
                                const bool diff = q >= te || zke_ring1(ring, q) != zke_ring1(ring, q - off);
                                const uint64_t dm = __ballot(diff);
                                if (dm) { L += (uint32_t)__builtin_ctzll(dm); break; }
                                L += 64;
                            }
                            if (lane == f) len = L;
                            const uint32_t e = f + L;
                            const uint64_t rest = candm & (e >= 64 ? 0ull : ~0ull << e);
                            f = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
                        } else f = (uint32_t)__builtin_amdgcn_readlane((int)nx, (int)f);
                    }
                    if (taken) skip = wb + lastf + (uint32_t)__builtin_amdgcn_readlane((int)len, (int)lastf);
                    ptaken[u] = taken; plen[u] = len;
                }
                first_s[4] = 0xFFFFu;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    // C1, all lanes at once: a taken lane's sequence index = sequences so far + taken lanes below it; its match starts up to
                    // `catch-up` bytes in front of its position, but not before the end of the match before it (round 5); its literal
                    // length = that start - that end
                    const uint32_t pos = 64 * u + lane;
                    const uint64_t taken = ptaken[u];
                    const uint32_t v = pv[u], len = plen[u];
                    const uint64_t below = taken & lane_lt;
                    const uint32_t myend = pos + len;
                    const uint32_t prevlane = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                    const uint32_t pe_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(prevlane << 2), (int)myend);
                    const uint32_t pe = below ? pe_ : aend;                             // end of the last match in front of me
                    const bool mine = (taken >> lane) & 1;
                    uint32_t bk = pbk[u];
                    bk = bk < pos - pe ? bk : pos - pe;                                 // (a taken lane: pos >= pe)
                    const uint32_t st = pos - bk;
                    if (mine) tseq[wave][c + (uint32_t)__builtin_popcountll(below)] = (uint64_t)((st - pe) | ((len + bk) << 12)) | ((uint64_t)(v >> 5) << 32);
                    ppe[u] = pe; pst[u] = st;
                    first_s[u] = 0xFFFFu;
                    if (taken) {
                        const uint32_t lastl = 63u - (uint32_t)__builtin_clzll(taken);
                        c += (uint32_t)__builtin_popcountll(taken);
                        aend = 64 * u + lastl + (uint32_t)__builtin_amdgcn_readlane((int)len, (int)lastl);
                        lastoff = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lastl) >> 5;
                        first_s[u] = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)__builtin_ctzll(taken));
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    // C2: a lane is a literal unless a match covers it -- the one in front of it (ends at pe), its own, or the catch-up of
                    // the next taken match (starts at sn: in this pass, else the first one of the next pass -- a later one is too far away)
                    const uint32_t pos = 64 * u + lane, p = ts + pos;
                    const uint64_t taken = ptaken[u];
                    const uint64_t above = taken & ~lane_lt & ~(1ull << lane);
                    const uint32_t nextlane = above ? (uint32_t)__builtin_ctzll(above) : 0u;
                    const uint32_t sn_ = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nextlane << 2), (int)pst[u]);
                    const uint32_t sn = above ? sn_ : first_s[u + 1];
                    const bool mine = (taken >> lane) & 1;
                    const uint64_t litm = __ballot(p < te && !mine && pos >= ppe[u] && pos < sn);
                    if ((litm >> lane) & 1) tl[nl + (uint32_t)__builtin_popcountll(litm & lane_lt)] = (uint8_t)zke_ring1(ring, p);
                    nl += (uint32_t)__builtin_popcountll(litm);
                }
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
