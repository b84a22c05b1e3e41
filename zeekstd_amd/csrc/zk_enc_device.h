// zk_enc_device.h -- records and per-lane helpers of the gfx950 frame encoder (zk_encode.hip).
// The decisions taken here (hash, code tables, repeat-offset policy, Huffman lengths) are the ones of
// the CPU twin oracle/zstd_oracle_enc.c, which the GPU output is compared against byte for byte.
#pragma once
#include <stdint.h>
#include "zk_device.h"

constexpr uint32_t ZKE_BLOCK = 131072;
constexpr uint32_t ZKE_HASH_LOG = 14;
constexpr uint32_t ZKE_MINMATCH = 6;
constexpr uint32_t ZKE_WINDOW = 65535;
constexpr uint32_t ZKE_TILE = 256;                // parse tile: matches never cross its end
constexpr uint32_t ZKE_LSTEP = 2;                 // tiles per lookup step (one position per lane: 512 lanes)
constexpr uint32_t ZKE_PARCAP = 64;
constexpr uint32_t ZKE_GROUP = 8;                // tiles parsed side by side, one wave each

// Block size the encoder cuts a frame of d_size bytes into.  Blocks are cut smaller than the format's maximum on purpose:
// a block's sequence bitstream is one serial chain for the decoder, so more, shorter blocks = more parallel chains
// (32 KiB for large frames, >= 16 blocks per small frame, never below 4 KiB: a seek into a 64 KiB frame waits for ONE
// block's chain, and at the idle clocks a sparse stream of seeks runs at that is ~40 us per KiB of block).
ZK_HD uint32_t zke_block_max(uint32_t d_size, bool prefix)
{
    uint32_t wlog = 10;
    while ((1u << wlog) < d_size && wlog < 17) wlog++;
    if (prefix) wlog = 17;
    uint32_t bm = (1u << wlog) < ZKE_BLOCK ? (1u << wlog) : ZKE_BLOCK;
    uint32_t t = 32768;
    while (t > 4096 && (uint64_t)t * 16 > d_size) t >>= 1;
    return t < bm ? t : bm;
}

struct ZkEncFrame {
    uint64_t src_off;           // where the frame's input starts in the source buffer
    uint32_t d_size;            // uncompressed bytes
    uint32_t n_blocks;
    uint32_t block_base;        // first entry of the frame in the block list
    uint32_t block_max;         // Block_Maximum_Size = min(window, 128 KiB)
    uint32_t window_log;
    uint32_t hist;              // bytes of history laid out before the frame in the matcher's source (prefix tail; 0 = none)
    uint64_t m_off;             // where that history starts in the matcher's source (== src_off when hist == 0)
};

struct ZkEncBlock {
    uint64_t seq_base;          // packed sequences (ll | ml << 20 | Offset_Value << 40)
    uint64_t lit_base;          // literal bytes
    uint64_t scratch_base;      // payload + bitstream temporaries
    uint32_t frame, bs, bsz;    // frame index, start inside the frame, size
    uint32_t nseq, nlit;
    uint32_t csize;             // content bytes that follow the 3-byte block header
    uint32_t mode;              // Block_Type: 0 raw, 1 RLE, 2 compressed
    uint8_t rle_byte, pad[3];
};

// predefined-distribution FSE compression tables (built on the host at engine creation)
struct ZkEncTables {
    uint16_t ll_state[64], of_state[32], ml_state[64];
    uint32_t ll_dfs[36], of_dfs[32], ml_dfs[56];
    uint32_t ll_dnb[36], of_dnb[32], ml_dnb[56];
    uint32_t ll_val[36], ml_val[56];            // base | extra bits << 24
};

struct ZkHufCode {              // a block's literal code: kept until the block's streams are written
    uint8_t len[128];
    uint16_t code[128];
};
struct ZkHufBuild {             // scratch of one tree build
    uint32_t w[256];
    int16_t parent[256];
    uint8_t idx[128], depth[256];
};

ZK_HD uint32_t zke_hash5(const uint8_t *p) { return (uint32_t)(((zk_ld64(p) << 24) * 889523592379ull) >> (64 - ZKE_HASH_LOG)); }

ZK_HD uint32_t zke_ll_code(uint32_t ll)
{
    if (ll < 16) return ll;
    if (ll < 64) {                                     // 16,16,17,17,18,18,19,19,20x4,21x4,22x8,23x8,24x16
        if (ll < 24) return 16 + ((ll - 16) >> 1);
        if (ll < 32) return 20 + ((ll - 24) >> 2);
        if (ll < 48) return 22 + ((ll - 32) >> 3);
        return 24;
    }
    return zk_highbit(ll) + 19;
}
ZK_HD uint32_t zke_ml_code(uint32_t mlb)               // mlb = match length - 3
{
    if (mlb < 32) return mlb;
    if (mlb < 128) {                                   // 32,32,33,33,34,34,35,35,36x4,37x4,38x8,39x8,40x16,41x16,42x32
        if (mlb < 40) return 32 + ((mlb - 32) >> 1);
        if (mlb < 48) return 36 + ((mlb - 40) >> 2);
        if (mlb < 64) return 38 + ((mlb - 48) >> 3);
        if (mlb < 96) return 40 + ((mlb - 64) >> 4);
        return 42;
    }
    return zk_highbit(mlb) + 36;
}

// offset -> Offset_Value; rep = the decoder's history as far as the encoder knows it (0 = unknown: every
// block starts unknown because the previous block may still be emitted raw / RLE)
ZK_HD uint32_t zke_off_to_code(uint32_t off, uint32_t ll, uint32_t &r0, uint32_t &r1, uint32_t &r2)
{
    uint32_t code = off + 3;
    if (ll) { if (off == r0) code = 1; else if (off == r1 && r0) code = 2; else if (off == r2 && r0 && r1) code = 3; }
    else { if (off == r1 && r0) code = 1; else if (off == r2 && r0 && r1) code = 2; else if (r0 > 1 && off == r0 - 1 && r1) code = 3; }
    if (code > 3) { r2 = r1; r1 = r0; r0 = off; }
    else {
        const uint32_t idx = code - 1 + (ll == 0);
        if (idx) { const uint32_t v = idx == 3 ? r0 - 1 : idx == 1 ? r1 : r2; if (idx > 1) r2 = r1; r1 = r0; r0 = v; }
    }
    return code;
}

ZK_HD uint32_t zke_cinit(const uint16_t *state, uint32_t dnb, uint32_t dfs)
{
    const uint32_t nb = (dnb + (1u << 15)) >> 16;
    const uint32_t v = (nb << 16) - dnb;
    return state[(v >> nb) + dfs];
}

// Huffman code lengths (<= 11) for symbols 0..nsym-1 (nsym <= 128); returns max length or -1.
// Two-queue Huffman over symbols sorted by (count, symbol); counts are halved (rounding up) until the tree fits.
// Code lengths <= 11 from the symbol counts (two-queue Huffman; counts are halved in place until the tree fits).
// cnt is modified.  Returns the deepest length, or -1 when fewer than two symbols occur.
ZK_HD int zke_huf_lengths(uint32_t *cnt, int nsym, ZkHufBuild *h, uint8_t *len)
{
    for (;;) {
        int m = 0;
        for (int s = 0; s < nsym; s++) if (cnt[s]) h->idx[m++] = (uint8_t)s;
        if (m < 2) return -1;
        for (int i = 1; i < m; i++) {
            int k = h->idx[i], j = i - 1;
            while (j >= 0 && (cnt[h->idx[j]] > cnt[k] || (cnt[h->idx[j]] == cnt[k] && h->idx[j] > k))) { h->idx[j + 1] = h->idx[j]; j--; }
            h->idx[j + 1] = (uint8_t)k;
        }
        int nn = m;
        for (int i = 0; i < m; i++) h->w[i] = cnt[h->idx[i]];
        int a = 0, bq = m;
        while ((m - a) + (nn - bq) > 1) {
            int p[2];
            for (int k = 0; k < 2; k++) { if (a < m && (bq >= nn || h->w[a] <= h->w[bq])) p[k] = a++; else p[k] = bq++; }
            h->w[nn] = h->w[p[0]] + h->w[p[1]];
            h->parent[p[0]] = h->parent[p[1]] = (int16_t)nn;
            nn++;
        }
        int maxd = 0;
        h->depth[nn - 1] = 0;
        for (int i = nn - 2; i >= 0; i--) h->depth[i] = (uint8_t)(h->depth[h->parent[i]] + 1);
        for (int i = 0; i < m; i++) if (h->depth[i] > maxd) maxd = h->depth[i];
        if (maxd <= 11) {
            for (int s = 0; s < nsym; s++) len[s] = 0;
            for (int i = 0; i < m; i++) len[h->idx[i]] = h->depth[i];
            return maxd;
        }
        for (int s = 0; s < nsym; s++) if (cnt[s]) cnt[s] = (cnt[s] + 1) >> 1;
    }
}

// canonical codes exactly as the decoder's table fill assigns them (weight 1 first, symbols ascending)
ZK_HD void zke_huf_codes(ZkHufCode *h, int nsym, int maxbits)
{
    uint32_t pos = 0;
    for (int wt = 1; wt <= maxbits; wt++)
        for (int s = 0; s < nsym; s++)
            if (h->len[s] && maxbits + 1 - h->len[s] == wt) { h->code[s] = (uint16_t)(pos >> (wt - 1)); pos += 1u << (wt - 1); }
}
