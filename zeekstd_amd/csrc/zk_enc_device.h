// zk_enc_device.h -- records and per-lane helpers of the gfx950 frame encoder (zk_encode.hip).
// The decisions taken here (hash, code tables, repeat-offset policy, Huffman lengths) are the ones of
// the CPU twin oracle/zstd_oracle_enc.c, which the GPU output is compared against byte for byte.
#pragma once
#include <stdint.h>
#include "zk_device.h"

constexpr uint32_t ZKE_BLOCK = 131072;
// What ZSTD_c_compressionLevel (encode.rs:170, 281-282) buys in the matcher (zk_enc_match.h; the CPU twin
// oracle/zstd_oracle_enc.c applies the same table):
//   level <= 1 (the "fast" end, what BASELINE.json's configs use): table matches of 6+ bytes, 2^14 table entries, greedy parse
//   level 2..5 and 0 = libzstd's default 3 (cli/src/args.rs:192): 5+ bytes, 2^15 entries, lazy parse (a longer match one or
//     two positions later wins)
//   level >= 6: the same with lookup steps of 1024 positions instead of 4096 (fresher tables: +3 % on source code)
//   (round 6) level 0 and 3 and up, frames beyond the ring's reach: DENSE far history -- every position of a matcher segment in two
//     tables of 2^17 slots (first / last occurrence), from level 9 on 2^18 (ZkEncLdm below; zk_k_enc_dense_part / _cand)
// 8d text (round 6): 2.484 (1) / 2.654 (2) / 2.732 (3) / 2.735 (6) / 2.742 (9); libzstd 1.5.7: 2.50 / 2.79 / 2.85 / 2.88.
ZK_HD bool zke_fast(int level) { return level != 0 && level < 2; }
ZK_HD uint32_t zke_minmatch(int level) { return zke_fast(level) ? 6u : 5u; }
ZK_HD uint32_t zke_hash_log(int level) { return zke_fast(level) ? 14u : 15u; }
ZK_HD uint32_t zke_lazy(int level) { return zke_fast(level) ? 0u : 1u; }
constexpr uint32_t ZKE_TILE = 256;                // parse tile: matches never cross its end; one wave each
constexpr uint32_t ZKE_GROUP = 16;                // tiles per group = waves per workgroup
constexpr uint32_t ZKE_GROUP_POS = ZKE_TILE * ZKE_GROUP;
ZK_HD uint32_t zke_step(int level) { return level >= 6 ? 1024u : ZKE_GROUP_POS; }
constexpr uint32_t ZKE_PARCAP = 16;               // match length measured per position (branch-free, 16 bytes per candidate); the parse extends longer ones
// The matcher keeps the last 64 KiB of its input in an LDS ring: the group it works on, the next group (loaded while this one is
// worked on), 64 bytes of lookahead, and the window behind the group -- the largest offset it produces.
constexpr uint32_t ZKE_RING = 65536;
constexpr uint32_t ZKE_WINDOW = ZKE_RING - 2 * ZKE_GROUP_POS - 64;   // 57280: the next group's bytes enter the ring while a group is still compared
// Round 5: the fast setting without a long-distance table (no prefix beyond the ring's reach) runs zk_enc_match2.h's kernel: candidates at
// even positions only, table matches of 5+ bytes, taken matches caught up backwards by up to 4 bytes (the twin: g_stride == 2).
ZK_HD bool zke_fast2(int level, uint64_t prefix_len) { return zke_fast(level) && prefix_len <= ZKE_WINDOW; }
ZK_HD uint32_t zke_minmatch2(int level, uint64_t prefix_len) { return zke_fast2(level, prefix_len) ? 5u : zke_minmatch(level); }
// The matcher's unit of work is a SEGMENT of a frame, one workgroup each: a 2 MiB frame spreads over 8 CUs, and 2048 such
// frames are 16384 workgroups.  A segment after a frame's first starts with an empty table that receives the positions of the
// ZKE_WINDOW bytes before it (what a prefix does for a frame), counts its positions from that history's start and does not
// look past its own end; a frame of up to ZKE_SEGMENT bytes is one segment.  (A multiple of the blocks every frame this
// large is cut into: 16 KiB up to 512 KiB, 32 KiB above.)
constexpr uint32_t ZKE_SEGMENT = 256u << 10;
constexpr uint32_t ZKE_SEAM = 32768;                // sequences of neighbouring tiles are joined except across multiples of this inside a block (match lengths < 2^16)

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
    uint32_t minmatch;          // zke_minmatch(level)
    uint32_t seg_at;            // a segment record: where the segment starts inside its frame
};

// LONG-DISTANCE MATCHES INTO A PREFIX (patch mode, cli/src/compress.rs:31-37: the reference turns on libzstd's long-distance
// matcher and a window that covers the whole prefix).  The ring reaches ZKE_WINDOW bytes back; a prefix longer than that is
// reached through a table in HBM over its last ZKE_LDM_MAX_OFF bytes: the positions whose 16-byte hash has five leading zeros
// (one in 32, chosen by content, so the same text is sampled at the same places in the old and the new file), first
// occurrence per slot.  A sampled position of the frame whose 16 bytes equal the entry's is a hit; the hit's offset is tried
// at every position of its tile, and as "previous offset" from then on -- compared through HBM (L2), not the ring.
constexpr uint32_t ZKE_LDM_MIN = 16;
constexpr uint32_t ZKE_LDM_MAX_OFF = (1u << 27) - 1;          // offsets fit the 27 bits of a best[] entry; Window_Descriptor <= 2^27
constexpr uint32_t ZKE_LDM_NONE = 0xFFFFFFFFu;
constexpr uint32_t ZKE_LDM_SLACK = 64;                        // readable bytes behind the prefix copy on the device
ZK_HD uint32_t zke_ldm_hash(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    return ((w0 * 0x9E3779B1u) ^ (w1 * 0x85EBCA77u)) + ((w2 * 0xC2B2AE3Du) ^ (w3 * 0x27D4EB2Fu));
}
ZK_HD bool zke_ldm_selected(uint32_t h) { return (h >> 27) == 0; }
ZK_HD uint32_t zke_ldm_slot(uint32_t h, uint32_t log) { return (h >> 2) & ((1u << log) - 1); }
ZK_HD uint64_t zke_ldm_usable(uint64_t plen) { return plen < ZKE_LDM_MAX_OFF ? plen : ZKE_LDM_MAX_OFF; }
ZK_HD uint32_t zke_ldm_log(uint64_t usable) { uint32_t l = 10; while (l < 23 && (1ull << l) < usable / 16) l++; return l; }
struct ZkEncLdm {
    const uint8_t *pfx;         // prefix byte q is pfx[q] for q in [u0, plen) (+ ZKE_LDM_SLACK readable bytes); nullptr: no long-distance matching
    const uint32_t *table;      // 2^log entries: position - u0 of the first sampled occurrence, ZKE_LDM_NONE = empty
    uint64_t plen, u0;
    uint32_t log;
    // IN-FRAME far history (round 4; oracle/zstd_oracle_enc.c ldm_build_frame has the rule): the same machinery over every frame's
    // OWN bytes -- from level 2 on, without a prefix, frames beyond the ring's reach.  Then `table` holds one table of 2^log entries
    // per frame of frame_size bytes of the n_total input bytes (a frame's own table uses the first 2^zke_ldm_log(its size) of them),
    // pfx is unused, and the match kernel makes its frame's view of this struct itself (zk_enc_match.h).
    uint32_t inframe;
    uint32_t frame_size, pad;
    uint64_t n_total;
    // DENSE far history (round 6; oracle/zstd_oracle_enc.c dense_build_frame has the rule and the measurements): per input byte position
    // (indexed like the source buffer, + ZKE_DENSE_SLACK entries; behind it as much again for the positions sorted by slot range) the position's far candidate as zk_k_enc_dense_cand found it -- length
    // (<= 16) | catch-up (<= 4: bytes in front of the position that agree at that distance) << 5 | distance << 8, 0: none -- out of two tables per matcher segment over the 5-byte hash of EVERY position (2^dlog slots: the
    // first occurrence in the segment, the last one in the segment before), which live in LDS only.  nullptr: none (levels 1 and 2,
    // prefixes, frames within the ring's reach).
    const uint32_t *dense;
    uint32_t dlog, pad2;
};
constexpr uint32_t ZKE_DENSE_AHEAD = 4, ZKE_DENSE_BONUS = 3;
constexpr uint32_t ZKE_DENSE_PASSES_MAX = 32;          // passes of the candidate kernel at 2^18 slots (2^13 of both tables in LDS per pass)
constexpr uint32_t ZKE_DENSE_MIN = 6, ZKE_DENSE_MARGIN = 2, ZKE_DENSE_NONE = 0xFFFFFFFFu, ZKE_DENSE_SLACK = 4096 + 16;
ZK_HD uint32_t zke_dense_log(int level) { return level >= 9 ? 18u : 17u; }
constexpr uint32_t ZKE_LDM_FILL = 6;    // in frame a far candidate is taken where the ring's best is shorter than this (the twin has the measurements)
ZK_HD bool zke_ldm_in_frame(int level, uint64_t prefix_len, uint64_t frame_bytes) { return !zke_fast(level) && prefix_len == 0 && frame_bytes > ZKE_WINDOW; }
ZK_HD bool zke_dense_in_frame(int level, uint64_t prefix_len, uint64_t frame_bytes) { return zke_ldm_in_frame(level, prefix_len, frame_bytes) && (level == 0 || level >= 3); }

// A compressed block's payload is put together by zk_k_enc_assemble out of the pieces the entropy stage leaves in the block's scratch:
//   [0, 16)      ZkEncPieces
//   [16, ...)    the literals section's head (header, tree description, jump table -- or the raw / RLE header and RLE byte), then the
//                sequences section's head (Number_of_Sequences, Symbol_Compression_Modes)
//   [ZKE_SMALL + k * scap, ...)   literal stream k (scap = q + q / 2 + 16, q = ceil(nlit / 4)),   [ZKE_SMALL + 4 * scap, ...)  the sequence bitstream
constexpr uint32_t ZKE_SMALL = 128;
struct ZkEncPieces {
    uint16_t z[4];              // bytes of the 4 literal streams (lit_mode 2)
    uint32_t zs;                // bytes of the sequence bitstream
    uint8_t head_lit, head_seq; // bytes of the two heads
    uint8_t lit_mode;           // 0 raw (the literal bytes follow the head: nlit of them out of the literal buffer), 1 RLE, 2 Huffman in 4 streams
    uint8_t pad;
};
static_assert(sizeof(ZkEncPieces) == 16, "one 16-byte load");
struct ZkEncBlock {
    uint64_t seq_base;          // packed sequences (ll | ml << 16 | Offset_Value << 32)
    uint64_t lit_base;          // literal bytes
    uint64_t scratch_base;      // the block's part of the entropy stage's scratch (ZKE_SMALL bytes of small parts, the 4 literal streams, the sequence stream)
    uint32_t frame, bs, bsz;    // frame index, start inside the frame, size
    uint32_t nseq, nlit;
    uint32_t csize;             // content bytes that follow the 3-byte block header (incl. the table descriptions of a defining block)
    uint32_t mode;              // Block_Type: 0 raw, 1 RLE, 2 compressed
    uint32_t modes_off;         // where Symbol_Compression_Modes sits in the payload (nseq > 0)
    uint8_t rle_byte;
    uint8_t is_def;             // this block carries the frame's FSE table descriptions (zk_k_enc_sizes decides)
    uint8_t pad[2];
    uint32_t out_at;            // where the block's 3-byte header goes, from the frame's first byte (zk_k_enc_sizes)
};
static_assert(sizeof(ZkEncBlock) == 64, "one line per block record");

// FSE compression tables of one frame: the predefined distributions (built on the host at engine creation) or -- per
// table -- the frame's own, measured from all its sequences by zk_k_enc_fse_build (accuracy logs 9 / 8 / 9).  One set per
// FRAME: the first compressed block with sequences carries the descriptions (FSE_Compressed_Mode), the later ones say
// Repeat_Mode, so a decoder builds them once per frame and still decodes the frame's blocks independently.
constexpr uint32_t ZKE_FSE_MIN_SEQ = 256;       // frames with fewer sequences keep the predefined tables
constexpr uint32_t ZKE_DESC_CAP = 80;           // bytes of one table description (53 symbols x <= 10 bits + repeats)
// Accuracy logs LL, OF, ML of a frame's own tables: the format's maxima (9 / 8 / 9) for frames with many sequences -- 7 / 7 / 7
// would cost 0.55 % of ratio (2.430 vs 2.443 on the 8d text) and measured no faster in the batch decoder -- and 6 / 6 / 6 for
// frames with fewer than ZKE_FSE_SMALL_SEQ sequences (frames up to ~190 KiB): a seek into such a frame waits for one serial
// table build per block, 512-cell tables made that 235 us at the idle clock (single seek 397 -> 633 us), 64-cell tables cost
// what the predefined ones cost; on 64 KiB frames 2.225 instead of 2.236.
constexpr uint32_t ZKE_FSE_SMALL_SEQ = 16384;
ZK_HD int zke_fse_log(int t, uint32_t nseq_frame) { return nseq_frame < ZKE_FSE_SMALL_SEQ ? 6 : t == 1 ? 8 : 9; }
struct ZkEncTables {
    uint16_t ll_state[512], of_state[256], ml_state[512];
    uint32_t ll_dfs[36], of_dfs[32], ml_dfs[56];
    uint32_t ll_dnb[36], of_dnb[32], ml_dnb[56];
    uint32_t ll_val[36], ml_val[56];            // base | extra bits << 24
    uint32_t al[3];                             // accuracy logs LL, OF, ML
    uint32_t custom;                            // bit t: table t is the frame's own
    uint32_t dlen[3];                           // bytes of the descriptions
    uint8_t desc[3][ZKE_DESC_CAP];
};
ZK_HD uint32_t zke_modes_byte(uint32_t custom, uint32_t mode) { return ((custom & 1u) ? mode << 6 : 0u) | ((custom & 2u) ? mode << 4 : 0u) | ((custom & 4u) ? mode << 2 : 0u); }

// ---- per-frame tables: normalisation, description, compression table (the CPU twin oracle/zstd_oracle_enc.c runs the same rules)
// floor(count * 2^L / total), at least 1 for a symbol that occurs; the most frequent symbol (lowest index on ties) takes
// what is missing; a surplus is taken from the largest entries.  false: fewer than two symbols.
ZK_HD bool zke_fse_normalize(const uint32_t *cnt, int nsym, int L, int16_t *norm)
{
    uint64_t total = 0; int distinct = 0, maxs = 0;
    for (int s = 0; s < nsym; s++) { total += cnt[s]; if (cnt[s]) { distinct++; if (cnt[s] > cnt[maxs]) maxs = s; } }
    if (distinct < 2) return false;
    const uint32_t size = 1u << L;
    uint32_t sum = 0;
    for (int s = 0; s < nsym; s++) {
        uint64_t v = cnt[s] ? ((uint64_t)cnt[s] << L) / total : 0;
        if (cnt[s] && v == 0) v = 1;
        norm[s] = (int16_t)v; sum += (uint32_t)v;
    }
    if (sum < size) norm[maxs] = (int16_t)(norm[maxs] + (size - sum));
    while (sum > size) {
        int big = 0;
        for (int s = 1; s < nsym; s++) if (norm[s] > norm[big]) big = s;
        const uint32_t d = (uint32_t)norm[big] - 1 < sum - size ? (uint32_t)norm[big] - 1 : sum - size;
        if (d == 0) return false;
        norm[big] = (int16_t)(norm[big] - d); sum -= d;
    }
    return true;
}
// FSE table description (RFC 8878 4.1.1): forward bits, LSB first.  Returns bytes written (0: no room).
ZK_HD uint32_t zke_fse_write_ncount(uint8_t *dst, uint32_t cap, const int16_t *norm, int nsym, int L)
{
    uint64_t acc = 0; uint32_t n = 0, pos = 0; bool ovf = false;
#define ZKE_NC_ADD(v, nb) do { acc |= (uint64_t)((uint32_t)(v) & ((1u << (nb)) - 1u)) << n; n += (nb); \
        while (n >= 8) { if (pos < cap) dst[pos] = (uint8_t)acc; else ovf = true; pos++; acc >>= 8; n -= 8; } } while (0)
    ZKE_NC_ADD(L - 5, 4);
    int remaining = (1 << L) + 1, threshold = 1 << L, nbits = L + 1, sym = 0;
    bool prev0 = false;
    while (remaining > 1 && sym < nsym) {
        if (prev0) {
            int start = sym;
            while (sym < nsym && norm[sym] == 0) sym++;
            if (sym == nsym) break;
            while (sym >= start + 3) { start += 3; ZKE_NC_ADD(3, 2); }
            ZKE_NC_ADD(sym - start, 2);
        }
        int count = norm[sym++];
        const int max = 2 * threshold - 1 - remaining;
        remaining -= count < 0 ? -count : count;
        count++;
        if (count >= threshold) count += max;
        ZKE_NC_ADD(count, (uint32_t)(nbits - (count < max ? 1 : 0)));
        prev0 = count == 1;
        while (remaining < threshold) { nbits--; threshold >>= 1; }
    }
    if (n) ZKE_NC_ADD(0, 8 - n);
#undef ZKE_NC_ADD
    return ovf ? 0u : pos;
}
// compression table of a normalised distribution (sym[] = scratch of 1 << al bytes, cumul[] of nsym + 2 ints)
ZK_HD void zke_build_ctable(const int16_t *norm, int nsym, int al, uint16_t *state, uint32_t *dfs, uint32_t *dnb, uint8_t *sym, int32_t *cumul)
{
    const int size = 1 << al, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    int high = size - 1;
    cumul[0] = 0;
    for (int u = 1; u <= nsym; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; sym[high--] = (uint8_t)(u - 1); }
        else cumul[u] = cumul[u - 1] + norm[u - 1];
    }
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) { sym[pos] = (uint8_t)s; do pos = (pos + step) & mask; while (pos > high); }
    for (int u = 0; u < size; u++) { const int s = sym[u]; state[cumul[s]++] = (uint16_t)(size + u); }
    int total = 0;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == 0) { dnb[s] = ((uint32_t)(al + 1) << 16) - (1u << al); dfs[s] = 0; }
        else if (norm[s] == -1 || norm[s] == 1) { dnb[s] = ((uint32_t)al << 16) - (1u << al); dfs[s] = (uint32_t)(total - 1); total++; }
        else {
            const uint32_t mbo = (uint32_t)al - zk_highbit((uint32_t)norm[s] - 1), msp = (uint32_t)norm[s] << mbo;
            dnb[s] = (mbo << 16) - msp; dfs[s] = (uint32_t)(total - norm[s]); total += norm[s];
        }
    }
}

struct ZkHufCode {              // a block's literal code: kept until the block's streams are written
    uint8_t len[128];
    uint16_t code[128];
};
struct ZkHufBuild {             // scratch of one tree build
    uint32_t w[256];
    int16_t parent[256];
    uint8_t idx[128], depth[256];
};

// hash of the 5 bytes lo (4) + b4: two 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 run at full rate, a 64-bit multiply at a quarter)
ZK_HD uint32_t zke_hash(uint32_t lo, uint32_t b4, uint32_t hlog) { const uint32_t a = lo & 0xFFFFFFu, b = (lo >> 24) | (b4 << 8); return (a * 0x9E3779u + b * 0x85EBCBu) >> (32 - hlog); }

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
// sorted >= 0: h->idx already holds that many symbols in the order (count, symbol) -- the entropy kernel ranks them with all the
// lanes of its wave; the insertion sort below is ~ m * m / 4 dependent LDS round trips, most of a build.
ZK_HD int zke_huf_lengths(uint32_t *cnt, int nsym, ZkHufBuild *h, uint8_t *len, int sorted = -1)
{
    for (;; sorted = -1) {
        int m = 0;
        if (sorted >= 0) m = sorted;
        else {
            for (int s = 0; s < nsym; s++) if (cnt[s]) h->idx[m++] = (uint8_t)s;
            for (int i = 1; i < m; i++) {
                int k = h->idx[i], j = i - 1;
                while (j >= 0 && (cnt[h->idx[j]] > cnt[k] || (cnt[h->idx[j]] == cnt[k] && h->idx[j] > k))) { h->idx[j + 1] = h->idx[j]; j--; }
                h->idx[j + 1] = (uint8_t)k;
            }
        }
        if (m < 2) return -1;
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
