/*
 * zstd_oracle_enc.c -- TEST INFRASTRUCTURE: CPU twin of the GPU frame encoder (zeekstd_amd/csrc/zk_encode.hip).
 *
 * What the reference does here is libzstd's ZSTD_compressStream2 (lib/src/encode.rs:340-346, 442-464);
 * compressed payload bytes are unpinned by the reference (libzstd 1.4.8 and 1.5.7 already
 * differ), only validity + round trip.  This file restates, sequentially, the algorithm the HIP kernels
 * run in parallel so that block/sequence/bitstream decisions can be compared kernel-vs-CPU byte for byte:
 *   match finder : positions in tiles of 256, groups of 8 tiles; phase 1: two tiles at a time, every position looks its
 *                  5-byte hash up in a 2^14-entry table (window 64 KiB) as it was before the step, and probes the
 *                  offset of the last match taken before the group; lengths capped at 64; then the step is inserted
 *   parse        : phase 2: every tile on its own, greedy, left to right; matches end at the tile end; stitched
 *   literals     : Huffman (<= 11 bits, direct 4-bit weights) in 4 streams, or raw / RLE
 *   sequences    : FSE with the PREDEFINED LL/OF/ML tables (Symbol_Compression_Modes = 0)
 *   frame        : magic, FHD (checksum bit), Window_Descriptor, <=128 KiB blocks, optional XXH64
 * Frames are checked by the oracle decoder AND by the real libzstd 1.5.7 / 1.4.8 in tests.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64; typedef int64_t i64;
u64 zko_xxh64(const u8 *p, size_t len, u64 seed);

#define ZKE_BLOCK 131072u
#define ZKE_HASH_LOG_MAX 16
#define ZKE_SEGMENT (256u << 10)     /* zk_enc_device.h: bytes of a frame one matcher workgroup takes (a multiple of the 32 KiB blocks) */
static u32 g_hash_log = 14;         /* zke_hash_log(level): 2^14 table entries at level <= 1, 2^15 at 2..5 and 0 (= default 3), 2^16 from 6 on */
static u32 g_minmatch = 6;          /* zke_minmatch(level): 6 for level <= 1 (except 0 = default 3), else 5 */
#define ZKE_WINDOW 57280u          /* 65536 - 2 * 4096 - 64: what the 64 KiB LDS ring of the GPU matcher still holds behind a group (zk_enc_device.h) */


static inline u32 hb32(u32 v) { return 31 - (u32)__builtin_clz(v); }
static inline u64 ld64(const u8 *p) { u64 v; memcpy(&v, p, 8); return v; }

/* ------------------------------------------------------------------ code tables (RFC 8878 3.1.1.3.2.1.1) */
static const u8 LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const u32 LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
static const u8 ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
static const u32 ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
static const short LL_DEF[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const short OF_DEF[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const short ML_DEF[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};

static u32 ll_code(u32 ll) { if (ll < 16) return ll; if (ll < 64) { static const u8 t[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24}; return t[ll]; } return hb32(ll) + 19; }
static u32 ml_code(u32 mlb /* ml - 3 */) { if (mlb < 128) { static const u8 t[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42}; return t[mlb]; } return hb32(mlb) + 36; }

/* ------------------------------------------------------------------ FSE compression tables for the predefined distributions */
typedef struct { int al; u16 state[512]; int dfs[64]; u32 dnb[64]; } fse_ctable;
static fse_ctable CT_LL, CT_OF, CT_ML;
static int g_ct_ready;

static void build_ctable(fse_ctable *ct, const short *norm, int nsym, int al)
{
    int size = 1 << al, mask = size - 1, step = (size >> 1) + (size >> 3) + 3, high = size - 1;
    u8 sym[512]; int cumul[66];
    ct->al = al;
    cumul[0] = 0;
    for (int u = 1; u <= nsym; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; sym[high--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + norm[u - 1];
    }
    int pos = 0;
    for (int s = 0; s < nsym; s++) for (int i = 0; i < norm[s]; i++) { sym[pos] = (u8)s; do pos = (pos + step) & mask; while (pos > high); }
    for (int u = 0; u < size; u++) { int s = sym[u]; ct->state[cumul[s]++] = (u16)(size + u); }
    int total = 0;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == 0) { ct->dnb[s] = ((u32)(al + 1) << 16) - (1u << al); ct->dfs[s] = 0; }
        else if (norm[s] == -1 || norm[s] == 1) { ct->dnb[s] = ((u32)al << 16) - (1u << al); ct->dfs[s] = total - 1; total++; }
        else { u32 mbo = (u32)al - hb32((u32)norm[s] - 1); u32 msp = (u32)norm[s] << mbo; ct->dnb[s] = (mbo << 16) - msp; ct->dfs[s] = total - norm[s]; total += norm[s]; }
    }
}
static void ct_init(void)
{
    if (g_ct_ready) return;
    build_ctable(&CT_LL, LL_DEF, 36, 6); build_ctable(&CT_OF, OF_DEF, 29, 5); build_ctable(&CT_ML, ML_DEF, 53, 6);
    g_ct_ready = 1;
}
/* exported so the GPU engine's host-built tables can be compared with these */
void zko_enc_ctable(int which, u16 *state, int *dfs, u32 *dnb)
{
    ct_init();
    fse_ctable *c = which == 0 ? &CT_LL : which == 1 ? &CT_OF : &CT_ML;
    memcpy(state, c->state, sizeof(u16) << c->al); memcpy(dfs, c->dfs, sizeof c->dfs); memcpy(dnb, c->dnb, sizeof c->dnb);
}

/* ------------------------------------------------------------------ per-frame FSE tables
 * The predefined distributions fit this engine's sequences badly (short matches, short literal runs): tables measured from
 * the frame's own sequences save ~20 % of the sequence section.  One set of tables per FRAME, not per block: the first
 * compressed block with sequences carries the three descriptions (FSE_Compressed_Mode), every later block of the frame
 * says Repeat_Mode -- so the decoder builds them once per frame and its blocks still decode independently.
 * Normalisation (own rule, integer only, the GPU kernel zk_k_enc_fse_build runs the same): floor(count * 2^L / total), at
 * least 1 for a symbol that occurs; the most frequent symbol (lowest index on ties) takes what is missing; a surplus is
 * taken from the largest entries.  Accuracy logs 9 / 8 / 9, or 6 / 6 / 6 for small frames.  A table needs >= 2 symbols. */
#define ZKE_FSE_MIN_SEQ 256u        /* frames with fewer sequences keep the predefined tables */
#define ZKE_FSE_SMALL_SEQ 16384u    /* fewer sequences than this in the frame: accuracy logs 6 / 6 / 6 instead of 9 / 8 / 9 (zk_enc_device.h) */
static int fse_normalize(const u32 *cnt, int nsym, int L, short *norm)
{
    u64 total = 0; int distinct = 0, maxs = 0;
    for (int s = 0; s < nsym; s++) { total += cnt[s]; if (cnt[s]) { distinct++; if (cnt[s] > cnt[maxs]) maxs = s; } }
    if (distinct < 2) return 0;
    const u32 size = 1u << L;
    u32 sum = 0;
    for (int s = 0; s < nsym; s++) {
        u64 v = cnt[s] ? ((u64)cnt[s] << L) / total : 0;
        if (cnt[s] && v == 0) v = 1;
        norm[s] = (short)v; sum += (u32)v;
    }
    if (sum < size) norm[maxs] = (short)(norm[maxs] + (size - sum));
    while (sum > size) {
        int big = 0;
        for (int s = 1; s < nsym; s++) if (norm[s] > norm[big]) big = s;
        u32 d = (u32)norm[big] - 1 < sum - size ? (u32)norm[big] - 1 : sum - size;
        if (d == 0) return 0;
        norm[big] = (short)(norm[big] - d); sum -= d;
    }
    return 1;
}

typedef struct { u8 *p; size_t cap, pos; u64 acc; int n; int ovf; } bitw;
static void bw_init(bitw *b, u8 *p, size_t cap);
static void bw_add(bitw *b, u32 v, int n);
/* FSE table description (RFC 8878 4.1.1), forward bits LSB first; returns bytes */
static size_t fse_write_ncount(u8 *dst, size_t cap, const short *norm, int nsym, int L)
{
    bitw b; bw_init(&b, dst, cap);
    bw_add(&b, (u32)(L - 5), 4);
    int remaining = (1 << L) + 1, threshold = 1 << L, nbits = L + 1, sym = 0, prev0 = 0;
    while (remaining > 1 && sym < nsym) {
        if (prev0) {
            int start = sym;
            while (sym < nsym && norm[sym] == 0) sym++;
            if (sym == nsym) break;
            while (sym >= start + 3) { start += 3; bw_add(&b, 3, 2); }
            bw_add(&b, (u32)(sym - start), 2);
        }
        int count = norm[sym++];
        const int max = 2 * threshold - 1 - remaining;
        remaining -= count < 0 ? -count : count;
        count++;
        if (count >= threshold) count += max;
        bw_add(&b, (u32)count, nbits - (count < max ? 1 : 0));
        prev0 = count == 1;
        while (remaining < threshold) { nbits--; threshold >>= 1; }
    }
    while (b.n > 0) bw_add(&b, 0, 8 - b.n);                          /* pad the last byte */
    return b.ovf ? 0 : b.pos;
}

/* ------------------------------------------------------------------ forward bit writer (LSB first) */
static void bw_init(bitw *b, u8 *p, size_t cap) { b->p = p; b->cap = cap; b->pos = 0; b->acc = 0; b->n = 0; b->ovf = 0; }
static void bw_add(bitw *b, u32 v, int n)
{
    if (!n) return;
    b->acc |= (u64)(v & ((n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1))) << b->n; b->n += n;
    while (b->n >= 8) { if (b->pos < b->cap) b->p[b->pos] = (u8)b->acc; else b->ovf = 1; b->pos++; b->acc >>= 8; b->n -= 8; }
}
static size_t bw_close(bitw *b) { bw_add(b, 1, 1); if (b->n) { if (b->pos < b->cap) b->p[b->pos] = (u8)b->acc; else b->ovf = 1; b->pos++; } return b->ovf ? 0 : b->pos; }

/* ------------------------------------------------------------------ sequences */
typedef struct { u32 ll, ml, offbase; } seq_t;      /* offbase = Offset_Value (>3: offset+3, 1..3: repeat codes) */

/* the three compression tables in force for a frame + what its defining block has to carry */
typedef struct { fse_ctable ll, of, ml; int custom[3]; u8 desc[3][80]; size_t dlen[3]; } frame_tables;

static size_t encode_sequences(const frame_tables *ft, const seq_t *sq, u32 n, u8 *dst, size_t cap)
{
    ct_init();
#define CT_LL ft->ll
#define CT_OF ft->of
#define CT_ML ft->ml
    bitw b; bw_init(&b, dst, cap);
    u32 llc = ll_code(sq[n - 1].ll), mlc = ml_code(sq[n - 1].ml - 3), ofc = hb32(sq[n - 1].offbase);
#define CINIT(ct, s, st) do { u32 nb = ((ct).dnb[s] + (1u << 15)) >> 16; u32 v = (nb << 16) - (ct).dnb[s]; st = (ct).state[(v >> nb) + (u32)(ct).dfs[s]]; } while (0)
#define CENC(ct, s, st) do { u32 nb = (st + (ct).dnb[s]) >> 16; bw_add(&b, st, (int)nb); st = (ct).state[(st >> nb) + (u32)(ct).dfs[s]]; } while (0)
    u32 sm, so, sl;
    CINIT(CT_ML, mlc, sm); CINIT(CT_OF, ofc, so); CINIT(CT_LL, llc, sl);
    bw_add(&b, sq[n - 1].ll - LL_BASE[llc], LL_BITS[llc]);
    bw_add(&b, sq[n - 1].ml - ML_BASE[mlc], ML_BITS[mlc]);
    bw_add(&b, sq[n - 1].offbase - (1u << ofc), (int)ofc);
    for (u32 i = n - 1; i-- > 0;) {
        llc = ll_code(sq[i].ll); mlc = ml_code(sq[i].ml - 3); ofc = hb32(sq[i].offbase);
        CENC(CT_OF, ofc, so); CENC(CT_ML, mlc, sm); CENC(CT_LL, llc, sl);
        bw_add(&b, sq[i].ll - LL_BASE[llc], LL_BITS[llc]);
        bw_add(&b, sq[i].ml - ML_BASE[mlc], ML_BITS[mlc]);
        bw_add(&b, sq[i].offbase - (1u << ofc), (int)ofc);
    }
    bw_add(&b, sm, CT_ML.al); bw_add(&b, so, CT_OF.al); bw_add(&b, sl, CT_LL.al);
    return bw_close(&b);
#undef CT_LL
#undef CT_OF
#undef CT_ML
}

/* tables for a frame from the code histograms of all its sequences */
static void frame_tables_build(frame_tables *ft, const u32 *hll, const u32 *hof, const u32 *hml, u32 nseq_frame)
{
    ct_init();
    ft->ll = CT_LL; ft->of = CT_OF; ft->ml = CT_ML;
    for (int t = 0; t < 3; t++) { ft->custom[t] = 0; ft->dlen[t] = 0; }
    if (nseq_frame < ZKE_FSE_MIN_SEQ) return;
    short norm[64];
    const u32 *h[3] = {hll, hof, hml};
    const int nsym[3] = {36, 32, 53};
    const int small = nseq_frame < ZKE_FSE_SMALL_SEQ;
    const int L[3] = {small ? 6 : 9, small ? 6 : 8, small ? 6 : 9};
    fse_ctable *ct[3] = {&ft->ll, &ft->of, &ft->ml};
    for (int t = 0; t < 3; t++) {
        if (!fse_normalize(h[t], nsym[t], L[t], norm)) continue;
        int last = nsym[t];
        while (last > 0 && norm[last - 1] == 0) last--;
        size_t d = fse_write_ncount(ft->desc[t], sizeof ft->desc[t], norm, last, L[t]);
        if (!d) continue;
        build_ctable(ct[t], norm, nsym[t], L[t]);
        ft->custom[t] = 1; ft->dlen[t] = d;
    }
}

/* ------------------------------------------------------------------ Huffman */
/* code lengths (<= 11) from counts: package by repeated count halving until the tree fits */
static int huf_lengths(const u32 *cnt_in, int nsym, u8 *len)
{
    u32 cnt[256]; int idx[256], m = 0;
    memcpy(cnt, cnt_in, sizeof(u32) * (size_t)nsym);
    for (;;) {
        m = 0;
        for (int s = 0; s < nsym; s++) if (cnt[s]) idx[m++] = s;
        if (m < 2) return -1;
        /* two-queue Huffman over symbols sorted by (count, symbol) */
        for (int i = 1; i < m; i++) { int k = idx[i], j = i - 1; while (j >= 0 && (cnt[idx[j]] > cnt[k] || (cnt[idx[j]] == cnt[k] && idx[j] > k))) { idx[j + 1] = idx[j]; j--; } idx[j + 1] = k; }
        u32 w[512]; int parent[512]; int nn = m;
        for (int i = 0; i < m; i++) w[i] = cnt[idx[i]];
        int a = 0, bq = m;                       /* leaves queue [a, m), internal queue [bq, nn) */
        while ((m - a) + (nn - bq) > 1) {
            int p[2];
            for (int k = 0; k < 2; k++) {
                if (a < m && (bq >= nn || w[a] <= w[bq])) p[k] = a++; else p[k] = bq++;
            }
            w[nn] = w[p[0]] + w[p[1]]; parent[p[0]] = parent[p[1]] = nn; nn++;
        }
        int maxd = 0; u8 depth[512];
        depth[nn - 1] = 0;
        for (int i = nn - 2; i >= 0; i--) { depth[i] = (u8)(depth[parent[i]] + 1); }
        for (int i = 0; i < m; i++) if (depth[i] > maxd) maxd = depth[i];
        if (maxd <= 11) {
            memset(len, 0, (size_t)nsym);
            for (int i = 0; i < m; i++) len[idx[i]] = depth[i];
            return maxd;
        }
        for (int s = 0; s < nsym; s++) if (cnt[s]) cnt[s] = (cnt[s] + 1) >> 1;      /* flatten and retry */
    }
}

/* canonical codes as the decoder builds them: weight = maxbits+1-len; table filled weight 1 first,
 * symbols ascending; code value = (table index) >> (maxbits - len), read MSB first */
static void huf_codes(const u8 *len, int nsym, int maxbits, u16 *code)
{
    u32 pos = 0;
    for (int wt = 1; wt <= maxbits; wt++)
        for (int s = 0; s < nsym; s++) if (len[s] && maxbits + 1 - len[s] == wt) { code[s] = (u16)(pos >> (wt - 1)); pos += 1u << (wt - 1); }
}

static size_t huf_encode_stream(const u8 *lit, size_t n, const u8 *len, const u16 *code, u8 *dst, size_t cap)
{
    bitw b; bw_init(&b, dst, cap);
    for (size_t i = n; i-- > 0;) bw_add(&b, code[lit[i]], len[lit[i]]);      /* last symbol first: decoder reads backward */
    return bw_close(&b);
}

/* literals section. Returns bytes written (0 on failure/overflow). */
static size_t encode_literals(const u8 *lit, size_t n, u8 *dst, size_t cap)
{
    /* RLE / tiny / raw fallbacks */
    int same = n > 0;
    for (size_t i = 1; i < n && same; i++) same = lit[i] == lit[0];
    size_t raw_hdr = n < 32 ? 1 : n < 4096 ? 2 : 3;
    if (same && n > 0) {
        if (cap < raw_hdr + 1) return 0;
        if (raw_hdr == 1) dst[0] = (u8)(1 | (n << 3)); else if (raw_hdr == 2) { dst[0] = (u8)(1 | (1 << 2) | (n << 4)); dst[1] = (u8)(n >> 4); }
        else { dst[0] = (u8)(1 | (3 << 2) | (n << 4)); dst[1] = (u8)(n >> 4); dst[2] = (u8)(n >> 12); }
        dst[raw_hdr] = lit[0];
        return raw_hdr + 1;
    }
    u32 cnt[256] = {0}; int maxsym = 0;
    for (size_t i = 0; i < n; i++) cnt[lit[i]]++;
    for (int s = 0; s < 256; s++) if (cnt[s]) maxsym = s;
    size_t out = 0;
    if (n >= 64 && maxsym < 128) {                                   /* direct 4-bit weights cover symbols 0..127 */
        u8 len[256]; u16 code[256];
        int nsym = maxsym + 1;
        int maxbits = huf_lengths(cnt, nsym, len);
        if (maxbits > 0) {
            huf_codes(len, nsym, maxbits, code);
            u8 tmp[ZKE_BLOCK + 1024];
            size_t hdr = n < 1024 ? 3 : n < 16384 ? 4 : 5;           /* 4 streams always (>= 64 literals) */
            size_t p = hdr;
            /* tree: weights of symbols 0..nsym-2 (the last one is implied) */
            int nw = nsym - 1;
            tmp[p++] = (u8)(127 + nw);
            for (int i = 0; i < nw; i += 2) {
                u8 w0 = len[i] ? (u8)(maxbits + 1 - len[i]) : 0, w1 = (i + 1 < nw && len[i + 1]) ? (u8)(maxbits + 1 - len[i + 1]) : 0;
                tmp[p++] = (u8)((w0 << 4) | w1);
            }
            size_t q = (n + 3) / 4, jt = p; p += 6;
            size_t sizes[4]; int ok = 1;
            for (int k = 0; k < 4 && ok; k++) {
                size_t cnt_k = k < 3 ? q : n - 3 * q;
                sizes[k] = huf_encode_stream(lit + k * q, cnt_k, len, code, tmp + p, sizeof tmp - p);
                if (!sizes[k] || (k < 3 && sizes[k] > 65535)) ok = 0;
                p += sizes[k];
            }
            size_t comp = p - hdr;
            if (ok && comp < n - (n >> 6) && comp < (hdr == 3 ? 1024u : hdr == 4 ? 16384u : 262144u)) {
                for (int k = 0; k < 3; k++) { tmp[jt + 2 * k] = (u8)sizes[k]; tmp[jt + 2 * k + 1] = (u8)(sizes[k] >> 8); }
                u64 h;
                if (hdr == 3) h = 2 | (1 << 2) | ((u64)n << 4) | ((u64)comp << 14);
                else if (hdr == 4) h = 2 | (2 << 2) | ((u64)n << 4) | ((u64)comp << 18);
                else h = 2 | (3 << 2) | ((u64)n << 4) | ((u64)comp << 22);
                for (size_t i = 0; i < hdr; i++) tmp[i] = (u8)(h >> (8 * i));
                if (p > cap) return 0;
                memcpy(dst, tmp, p);
                out = p;
            }
        }
    }
    if (!out) {                                                      /* raw literals */
        if (cap < raw_hdr + n) return 0;
        if (raw_hdr == 1) dst[0] = (u8)(n << 3); else if (raw_hdr == 2) { dst[0] = (u8)((1 << 2) | (n << 4)); dst[1] = (u8)(n >> 4); }
        else { dst[0] = (u8)((3 << 2) | (n << 4)); dst[1] = (u8)(n >> 4); dst[2] = (u8)(n >> 12); }
        memcpy(dst + raw_hdr, lit, n);
        out = raw_hdr + n;
    }
    return out;
}

/* ------------------------------------------------------------------ match finder + parse for one block */
/* The matcher of zk_k_enc_match (zeekstd_amd/csrc/zk_enc_match.h), restated sequentially.  Positions count from the start
 * of the segment's record = [history | data]; the GPU keeps the last 64 KiB of the record in an LDS ring, so
 *   - a table entry is the low 16 bits of a position (= its ring index), a candidate is p - d with d = (p - entry) mod 2^16,
 *     valid when 1 <= d <= ZKE_WINDOW and d <= p (anything else is stale or aliased; the bytes decide as for any collision);
 *   - ZKE_WINDOW = 65536 - 2 * ZKE_GROUP_POS - 64: the ring also holds the group being worked on, the next one and 64 bytes of lookahead.
 * Work proceeds in GROUPS of 16 tiles x 256 positions.  Per group (STEP == group) or per step of ZKE_STEP positions:
 *   1. every position looks its 5-byte hash up in the table as it was before the step          -> "far" candidate
 *   2. the step's positions are inserted, the SMALLEST position of the step wins a slot (order-free rule: the GPU's lanes
 *      race with compare-and-swap); an older entry always loses
 *   3. every position looks its slot up again: an earlier position of the same step, if any      -> "near" candidate
 *      (offsets below the step size, which the stale table of 1. cannot see)
 *   4. two more candidates per position: offset 1 (byte runs) and R = the offset of the last sequence before the group
 *   5. per position the longest candidate wins (capped at 16 bytes and at the tile end), ties in the order far < near <
 *      offset 1 < R; table candidates need minmatch bytes (level), the other two 4.
 * Then every tile is parsed on its own -- greedy or lazy (level), matches never cross the tile end -- and the tiles are
 * stitched: literals left over at a tile's end join the next sequence.  Offset_Value: 1 ("repeat the previous offset") when
 * the offset equals the previous sequence's offset of the same block and the sequence has literals, else offset + 3. */
#define ZKE_PARCAP 16u              /* match length measured per position; longer ones are extended by the parse (64 would gain 0.0 % on the 8d text, 0.06 % on source code) */
#define ZKE_TILE 256u
#define ZKE_GROUP 16u               /* tiles per group (one wave each on the GPU) */
#define ZKE_GROUP_POS (ZKE_TILE * ZKE_GROUP)
typedef struct {
    u16 table[1 << ZKE_HASH_LOG_MAX]; u32 t32[1 << 14]; u32 probe, stepno, bias;
    /* long-distance candidates into a prefix (patch mode; zk_enc_device.h "LDM") */
    const u8 *pfx; u64 plen; u32 *ldm; u32 ldm_log; u64 ldm_u0;             /* the whole prefix, table over prefix[u0, plen) */
    u64 lim; int inframe;                                                   /* round 4: the same machinery over the FRAME's own bytes (pfx = the frame, plen = 0, lim = its size) */
    u64 abs0;                                                               /* stream coordinate of record position 0 (prefix byte i = i, frame byte x = plen + x) */
    u32 *dense; u32 dense_log;                                              /* round 6: per matcher segment of the frame, first and last occurrence per slot of EVERY position (below) */
} enc_state;
static u32 g_lazy = 0;              /* zke_lazy(level) */
/* ROUND 5 -- the fast setting (level <= 1 without a long-distance table; zk_enc_match2.h is its kernel): candidates are
 * looked up, compared and parsed at EVEN positions only (every position is still inserted into the table, so a candidate may lie
 * anywhere), table matches need 5 bytes instead of 6, and a match that is taken is extended BACKWARDS by up to 4 bytes that
 * agree at its offset -- into the literals in front of it, never past the tile's start or the previous match.  What libzstd's
 * ZSTD_fast gets from stepping over positions and "catching up" matches: half of the comparisons, and the ratio on the 8d text
 * goes 2.4715 -> 2.4846 (a match the stale table missed at its first byte is found two bytes later and caught up). */
static u32 g_stride = 1, g_back = 0;
/* Two forms of the table (zk_enc_match.h).  16-bit entries as described above: 2^15 of them are what fits beside the ring
 * at level >= 2.  At level <= 1 the 2^14 entries are 32 bits wide: (0xFFFF - step number) << 16 | position mod 2^16, the
 * steps of a segment counted from 0 (history first, 4096 positions per step); "position" here is position + bias with
 * bias = -history mod 4096, so that no step straddles a multiple of 2^16 and the low halves of a step's keys order like its
 * positions (distances are unaffected) -- an insertion is then ONE atomic minimum
 * (the latest step wins, inside a step the smallest position), the empty entry is 0xFFFFFFFF, "an earlier position of this
 * step" is an exact test on the upper half, and history is inserted step by step under the same rule. */
static u32 g_tab32 = 1;
static u32 g_step = ZKE_GROUP_POS;  /* zke_step(level) */

static inline u32 hash5(const u8 *p, u32 log)           /* 5 bytes; two 24-bit multiplies (full-rate v_mul_u32_u24 / v_mad_u32_u24) */
{
    u32 lo; memcpy(&lo, p, 4);
    const u32 a = lo & 0xFFFFFFu, b = (lo >> 24) | ((u32)p[4] << 8);
    return (u32)(a * 0x9E3779u + b * 0x85EBCBu) >> (32 - log);
}
static inline u32 hashx(const u8 *p) { return hash5(p, g_hash_log); }

static u32 match_len(const u8 *a, const u8 *b, const u8 *end)        /* b > a */
{
    const u8 *s = b;
    while (b + 8 <= end) { u64 x = ld64(a) ^ ld64(b); if (x) return (u32)(b - s) + (u32)(__builtin_ctzll(x) >> 3); a += 8; b += 8; }
    while (b < end && *a == *b) { a++; b++; }
    return (u32)(b - s);
}

/* ---- long-distance matches into a prefix (cli/src/compress.rs:31-37: patch mode sets a window over the reference file and
 * enables libzstd's long-distance matcher).  The ring of the GPU matcher reaches 57 280 bytes back; a prefix longer than that
 * gets a coarse hash table over its last 2^27 - 1 bytes: positions whose 16-byte hash has its top 5 bits clear (1 in 32,
 * content-defined, so the old and the new file sample the same places) enter a table of first occurrences.  A sampled
 * position of the frame looks its slot up and compares 16 bytes against the prefix: all equal = a hit, a candidate of length
 * 16 with the offset "position + distance to the prefix byte".  The offset of a tile's first hit is tried at every position
 * of the tile; once such a match is taken its offset is the previous offset R of the next group, compared through memory
 * instead of the ring.  Tiles that continue a match are joined into one sequence (ZKE_SEAM).  Offsets are limited to
 * 2^27 - 1 (the field best[] has for them).  zeekstd_amd/csrc/zk_enc_device.h "ZkEncLdm" is the GPU side. */
#define ZKE_SEAM 32768u
#define ZKE_LDM_MIN 16u
#define ZKE_LDM_MAX_OFF ((1u << 27) - 1)
static inline u32 ldm_hash(const u8 *p)
{
    u32 w[4]; memcpy(w, p, 16);
    return ((w[0] * 0x9E3779B1u) ^ (w[1] * 0x85EBCA77u)) + ((w[2] * 0xC2B2AE3Du) ^ (w[3] * 0x27D4EB2Fu));
}
static inline int ldm_selected(u32 h) { return (h >> 27) == 0; }
static u32 ldm_log_for(u64 usable) { u32 l = 10; while (l < 23 && (1ull << l) < usable / 16) l++; return l; }
/* the table covers the last ZKE_LDM_MAX_OFF bytes of the prefix (all of a shorter one), whatever the frame's size: one
 * table serves every frame of a stream; a position too far back for the offset field is turned down at the lookup */
static void ldm_build(enc_state *st, const u8 *prefix, u64 plen)
{
    st->ldm = NULL; st->pfx = prefix; st->plen = plen; st->lim = plen; st->inframe = 0;
    if (plen <= ZKE_WINDOW) return;
    const u64 usable = plen < ZKE_LDM_MAX_OFF ? plen : ZKE_LDM_MAX_OFF;
    st->ldm_u0 = plen - usable; st->ldm_log = ldm_log_for(usable);
    st->ldm = malloc(sizeof(u32) << st->ldm_log);
    memset(st->ldm, 0xFF, sizeof(u32) << st->ldm_log);
    for (u64 q = st->ldm_u0; q + ZKE_LDM_MIN <= plen; q++) {
        const u32 h = ldm_hash(prefix + q);
        if (!ldm_selected(h)) continue;
        u32 *slot = &st->ldm[(h >> 2) & ((1u << st->ldm_log) - 1)];
        if ((u32)(q - st->ldm_u0) < *slot) *slot = (u32)(q - st->ldm_u0);         /* the first occurrence keeps the slot */
    }
}
/* IN-FRAME far history (round 4): the ring reaches 57 280 bytes back, libzstd's level 3 -- the reference CLI's default,
 * cli/src/args.rs:192 -- two MiB.  From level 2 on a frame without a prefix that is longer than the ring's reach gets the same
 * table over ITS OWN bytes: every sampled position of the frame enters (first occurrence per slot), a sampled position looks its
 * slot up and takes the entry if it lies more than ZKE_WINDOW bytes BEHIND it -- nearer ones are the ring's business -- and its
 * 16 bytes agree.  Such a candidate only fills gaps: it is taken where the ring's best candidate is shorter than ZKE_LDM_FILL = 6
 * bytes (a far offset costs ~8 more bits; with "wherever the ring has fewer than 16" runs of 10 random bytes came out 2.7 %
 * larger than at level 1, with 6 the documents keep all but 0.6 % of the gain), where a hit into a prefix wins.  The frame then
 * declares a window over its whole size. */
#define ZKE_LDM_FILL 6u             /* in frame a far candidate is taken where the ring's best is shorter than this (zk_enc_device.h) */
static int ldm_wanted_in_frame(int level, u64 plen, size_t n) { const int fast = level != 0 && level < 2; return !fast && plen == 0 && n > ZKE_WINDOW; }
static void ldm_build_frame(enc_state *st, const u8 *frame, size_t n)
{
    st->pfx = frame; st->plen = 0; st->lim = n; st->inframe = 1; st->ldm_u0 = 0; st->ldm_log = ldm_log_for(n);
    st->ldm = malloc(sizeof(u32) << st->ldm_log);
    memset(st->ldm, 0xFF, sizeof(u32) << st->ldm_log);
    for (u64 q = 0; q + ZKE_LDM_MIN <= n; q++) {
        const u32 h = ldm_hash(frame + q);
        if (!ldm_selected(h)) continue;
        u32 *slot = &st->ldm[(h >> 2) & ((1u << st->ldm_log) - 1)];
        if ((u32)q < *slot) *slot = (u32)q;
    }
}

/* ROUND 6 -- DENSE far history (level 0 = default 3, and 3 and up; VERDICT r5 "a real upper level"): the sampled table above finds far
 * copies of 16+ bytes; what libzstd's level 3 (cli/src/args.rs:192) gains over a 57 280-byte window on text are SHORT matches of rare words
 * whose last occurrence lies further back.  Per matcher segment g of the frame (ZKE_SEGMENT bytes, one GPU workgroup) two tables of
 * 2^dense_log slots over the 5-byte hash of EVERY position of the segment: first[g] = its smallest, last[g] = its largest position per slot
 * (order-free rules: the GPU builds them with LDS atomics, zk_k_enc_dense_build).  A position p of segment g looks ONE candidate up:
 * first[g] if that lies more than ZKE_WINDOW bytes behind p (nearer ones are the ring's), else last[g - 1] under the same condition --
 * the nearest far copy the two tables know; older segments are not asked (measured on the 8d text at 2^17 slots: asking all of them
 * 2.7009, the one before 2.7088: a farther offset costs more bits than the match saves).  The candidate is taken if it is at least
 * ZKE_DENSE_MIN bytes long and ZKE_DENSE_MARGIN longer than what the ring and the sampled table found (a far offset costs ~8 more bits).
 * 8d text: level 3 2.654 -> 2.709 (2^17 slots), level 9 and up 2.72 (2^18); with the candidates caught up backwards (find_sequences) 2.732 / 2.744;
 * libzstd 1.5.7: 2.79 / 2.88. */
#define ZKE_DENSE_MIN 6u
#define ZKE_DENSE_MARGIN 2u
#define ZKE_DENSE_NONE 0xFFFFFFFFu
#define ZKE_DENSE_AHEAD 4u
#define ZKE_DENSE_BONUS 3u
static int dense_wanted(int level, u64 plen, size_t n) { return ldm_wanted_in_frame(level, plen, n) && (level == 0 || level >= 3); }
static u32 dense_log_for(int level) { return level >= 9 ? 18 : 17; }
static void dense_build_frame(enc_state *st, const u8 *frame, size_t n, int level)
{
    const u32 log = dense_log_for(level), nseg = (u32)((n + ZKE_SEGMENT - 1) / ZKE_SEGMENT);
    st->dense_log = log;
    st->dense = malloc(((size_t)nseg * 2 * sizeof(u32)) << log);
    for (u32 g = 0; g < nseg; g++) {
        u32 *first = st->dense + ((size_t)(2 * g) << log), *last = first + ((size_t)1 << log);
        memset(first, 0xFF, sizeof(u32) << log); memset(last, 0, sizeof(u32) << log);
        const u64 s0 = (u64)g * ZKE_SEGMENT, e1 = s0 + ZKE_SEGMENT < n ? s0 + ZKE_SEGMENT : n;
        for (u64 q = s0; q < e1 && q + 8 <= n; q++) {
            if (frame[q + 1] == frame[q] && frame[q + 2] == frame[q] && frame[q + 3] == frame[q]) continue;      /* a byte run: neither entered nor looked up (dense_lookup) */
            const u32 h = hash5(frame + q, log), r = (u32)(q - s0);
            if (r < first[h]) first[h] = r;
            if (r + 1 > last[h]) last[h] = r + 1;
        }
    }
}
/* the frame position of the dense candidate of position p (frame coordinate ap); ~0: none */
static u64 dense_lookup(const enc_state *st, const u8 *p, u64 ap)
{
    const u32 log = st->dense_log, h = hash5(p, log);
    /* not at a position whose first four bytes are one byte: that is a byte run, and "a literal, then the run at offset 1" codes it in fewer bits
     * than any far copy and keeps the repeat offset (runs of 10 random bytes: level 3 came out 4.5 % behind level 1 without this) */
    if (p[1] == p[0] && p[2] == p[0] && p[3] == p[0]) return ~0ull;
    const u64 g = ap / ZKE_SEGMENT, s0 = g * ZKE_SEGMENT;
    const u32 *first = st->dense + ((size_t)(2 * g) << log);
    u32 m = first[h];
    if (m != ZKE_DENSE_NONE && s0 + m < ap && ap - (s0 + m) > ZKE_WINDOW) return s0 + m;
    if (g) { m = (first - ((size_t)1 << log))[h]; if (m && ap - (s0 - ZKE_SEGMENT + m - 1) > ZKE_WINDOW) return s0 - ZKE_SEGMENT + m - 1; }
    return ~0ull;
}

/* the prefix position a sampled position p (stream coordinate ap) finds in the table, if at least ZKE_LDM_MIN of the fcap bytes agree; ~0: none */
static u64 ldm_lookup(const enc_state *st, const u8 *p, u64 ap, u32 fcap)
{
    const u32 h = ldm_hash(p);
    if (!ldm_selected(h)) return ~0ull;
    const u32 e = st->ldm[(h >> 2) & ((1u << st->ldm_log) - 1)];
    const u64 q = st->ldm_u0 + e;
    if (e == 0xFFFFFFFFu || q + 16 > st->lim || ap - q > ZKE_LDM_MAX_OFF) return ~0ull;
    if (st->inframe && (q >= ap || ap - q <= ZKE_WINDOW)) return ~0ull;
    return match_len(st->pfx + q, p, p + fcap) >= ZKE_LDM_MIN ? q : ~0ull;
}

/* a long-distance offset is compared at position p while the 20 bytes that p's aligned group of four positions reads at that
 * distance lie inside the table's part of the prefix (the GPU lane loads them as five words, once for its four positions) */
static int far_ok(const enc_state *st, u32 p, u32 off)
{
    const u64 a4 = st->abs0 + (p & ~3u);
    return a4 >= st->ldm_u0 + off && a4 - off + 20 <= st->lim;
}

/* history positions [0, hist) enter an empty table, the largest position wins a slot */
static void table_seed(enc_state *st, const u8 *base, u32 hist, u32 fend)
{
    st->probe = 0; st->stepno = 0; st->bias = (0u - hist) & (ZKE_GROUP_POS - 1);
    if (g_tab32) {
        memset(st->t32, 0xFF, sizeof st->t32);
        for (u32 c0 = 0; c0 < hist; c0 += ZKE_GROUP_POS, st->stepno++)
            for (u32 v = c0; v < c0 + ZKE_GROUP_POS && v < hist; v++) if ((size_t)v + 8 <= fend) {
                u32 *slot = &st->t32[hashx(base + v)];
                const u32 key = ((0xFFFFu - st->stepno) << 16) | ((v + st->bias) & 0xFFFFu);
                if (key < *slot) *slot = key;
            }
        return;
    }
    memset(st->table, 0, sizeof st->table);
    for (u32 v = 0; v < hist; v++) if ((size_t)v + 8 <= fend) st->table[hashx(base + v)] = (u16)v;
}

static inline u32 ts0(u32 p, u32 gs, u32 T) { return gs + ((p - gs) / T) * T; }      /* start of p's tile */
static u32 find_sequences(enc_state *st, const u8 *base, u32 bs, u32 be, u32 fend, seq_t *sq, u8 *lits, u32 *nlit_out)
{
    u32 nseq = 0, nlit = 0, anchor = bs, prev_off = 0;
    static u32 blen[ZKE_GROUP_POS], boff[ZKE_GROUP_POS];
    static u16 e0[ZKE_GROUP_POS];
    static u8 bback[ZKE_GROUP_POS];
    const u32 T = ZKE_TILE;
    for (u32 gs = bs; gs < be; gs += ZKE_GROUP_POS) {
        const u32 ge = gs + ZKE_GROUP_POS < be ? gs + ZKE_GROUP_POS : be;
        const u32 R = st->probe;
        u32 tfar[ZKE_GROUP];                                                             /* per tile: the offset of its first long-distance hit */
        for (u32 t = 0; t < ZKE_GROUP; t++) {
            tfar[t] = 0;
            const u32 ts = gs + t * T, te = ts + T < be ? ts + T : be;
            if (st->ldm) for (u32 p = ts; p < te && !tfar[t]; p++) if (p + ZKE_LDM_MIN <= fend) {
                const u32 fcap = te - p < 16 ? te - p : 16;
                const u64 q = ldm_lookup(st, base + p, st->abs0 + p, fcap);
                if (q != ~0ull) tfar[t] = (u32)(st->abs0 + p - q);
            }
        }
        for (u32 ls = gs; ls < ge; ls += g_step) {
            const u32 le = ls + g_step < ge ? ls + g_step : ge;
            const u32 khi = (0xFFFFu - st->stepno) << 16;
            const u32 bias = g_tab32 ? st->bias : 0;
            for (u32 p = ls; p < le; p++) e0[p - ls] = p + 8 <= fend ? (g_tab32 ? (u16)st->t32[hashx(base + p)] : st->table[hashx(base + p)]) : 0;
            if (g_tab32) {
                for (u32 p = ls; p < le; p++) if (p + 8 <= fend) {
                    u32 *slot = &st->t32[hashx(base + p)];
                    if ((khi | ((p + bias) & 0xFFFFu)) < *slot) *slot = khi | ((p + bias) & 0xFFFFu);
                }
            } else {
                const u32 b16 = ls & 0xFFFFu, span = le - ls;
                for (u32 p = ls; p < le; p++) if (p + 8 <= fend) {
                    u16 *slot = &st->table[hashx(base + p)];
                    const u32 mine = (p - b16) & 0xFFFFu, cur = (*slot - b16) & 0xFFFFu;     /* step-relative */
                    if (cur >= span || cur > mine) *slot = (u16)p;
                }
            }
            st->stepno++;
            for (u32 p = ls; p < le; p++) {
                const u32 ts = gs + ((p - gs) / T) * T, te = ts + T < be ? ts + T : be;
                const u8 *lim = base + te;
                const u8 *cap = base + p + ZKE_PARCAP < lim ? base + p + ZKE_PARCAP : lim;
                u32 bl = 0, bo = 0;
                if (p + 8 <= fend) {
                    u32 d = (p + bias - e0[p - ls]) & 0xFFFFu;                               /* far */
                    if (d && d <= p && d <= ZKE_WINDOW) { const u32 l = match_len(base + p - d, base + p, cap); if (l >= g_minmatch) { bl = l; bo = d; } }
                    int near_ok;                                                             /* near: an earlier position of this step */
                    if (g_tab32) { const u32 e = st->t32[hashx(base + p)]; d = (p + bias - e) & 0xFFFFu; near_ok = (e & 0xFFFF0000u) == khi && d; }
                    else { d = (p - st->table[hashx(base + p)]) & 0xFFFFu; near_ok = d && d <= p - ls; }
                    if (near_ok) { const u32 l = match_len(base + p - d, base + p, cap); if (l >= g_minmatch && l >= bl) { bl = l; bo = d; } }
                }
                const u64 ap = st->abs0 + p;                                              /* my stream coordinate */
                const u32 ncap = (u32)(cap - (base + p)), fcap = ncap < 16 ? ncap : 16;
                /* long-distance candidates out of the prefix: the position's own table entry, and the offset of the tile's first
                 * hit (positions in front of a sampled one, and the tiles of a group behind a change, find the copy that way) */
                if (st->ldm && p + 8 <= fend) {
                    const u64 q = p + ZKE_LDM_MIN <= fend ? ldm_lookup(st, base + p, ap, fcap) : ~0ull;
                    if (q != ~0ull) { const u32 l = match_len(st->pfx + q, base + p, base + p + fcap); if (st->inframe ? bl < ZKE_LDM_FILL : (l > bl || l == ZKE_PARCAP)) { bl = l; bo = (u32)(ap - q); } }
                    const u32 R2 = tfar[(p - gs) / T];
                    if (R2 && R2 != R && far_ok(st, p, R2)) {
                        const u32 l = match_len(st->pfx + (ap - R2), base + p, base + p + fcap);
                        if (l >= ZKE_LDM_MIN && (st->inframe ? bl < ZKE_LDM_FILL : (l > bl || l == ZKE_PARCAP))) { bl = l; bo = R2; }
                    }
                }
                u32 dense_d = 0;                                                            /* distance of the position's entry in the GPU's candidate array (0: none) */
                if (st->dense && p + 8 <= fend) {                                          /* round 6: the nearest far copy the segment tables know */
                    const u64 q = dense_lookup(st, base + p, ap);
                    if (q != ~0ull) {
                        const u32 l = match_len(st->pfx + q, base + p, base + p + fcap);
                        /* (the entry exists if 6 of the position's 16 bytes agree, whatever the tile's end cuts off: the kernel that makes it knows no tiles) */
                        const u64 left = st->lim - ap;
                        if (match_len(st->pfx + q, base + p, base + p + (left < 16 ? left : 16)) >= ZKE_DENSE_MIN) dense_d = (u32)(ap - q);
                        if (l >= ZKE_DENSE_MIN && l >= bl + ZKE_DENSE_MARGIN) { bl = l; bo = (u32)(ap - q); }
                    }
                }
                if (p >= 1) { const u32 l = match_len(base + p - 1, base + p, cap); if (l >= 4 && l >= bl) { bl = l; bo = 1; } }
                if (R > 1 && R <= ZKE_WINDOW) { if (R <= p) { const u32 l = match_len(base + p - R, base + p, cap); if (l >= 4 && l + 1 >= bl) { bl = l; bo = R; } } }   /* (round 5) the previous offset is cheap to code: it also wins one byte short */
                else if (R > ZKE_WINDOW && far_ok(st, p, R)) {        /* a previous offset beyond the ring: through memory */
                    const u32 l = match_len(st->pfx + (ap - R), base + p, base + p + fcap);
                    if (l >= 4 && l + 1 >= bl) { bl = l; bo = R; }
                }
                u32 bb = 0;
                if (g_stride == 2 && (p & 1)) { bl = 0; bo = 0; }
                /* catch-up bytes in front of the position, compared as the GPU lane does it: the four bytes in front of the
                 * candidate out of the ring, which holds them while off + 4 <= ZKE_WINDOW (a source beyond the ring: none) */
                if (bl && bo + 4 <= ZKE_WINDOW) while (bb < g_back && p - bb > ts0(p, gs, T) && p - bb > bo && base[p - bb - 1] == base[p - bb - 1 - bo]) bb++;
                /* (round 6) ... and a winner at the distance of the position's dense candidate is caught up through memory (the entry carries the count:
                 * how many of the four bytes in front of the position agree at that distance, not past the frame's first byte): the far copy of a
                 * word is found at its second or third letter as often as a near one.  8d text, level 3: 2.709 -> 2.732 */
                else if (bl && dense_d && bo == dense_d) while (bb < g_back && p - bb > ts0(p, gs, T) && ap - bb > bo && st->pfx[ap - bb - 1] == st->pfx[ap - bb - 1 - bo]) bb++;
                blen[p - gs] = bl; boff[p - gs] = bo; bback[p - gs] = bb;
            }
        }
        /* per-tile parses + stitching */
        for (u32 ts = gs; ts < ge; ts += T) {
            const u32 te = ts + T < be ? ts + T : be;
            const u8 *lim = base + te;
            u32 p = ts;
            while (p < te) {
                u32 len = blen[p - gs];
                if (len && g_lazy) {          /* a longer match one position later, or a clearly longer one two later, wins */
                    if ((p + 1 < te && blen[p + 1 - gs] > len) || (p + 2 < te && blen[p + 2 - gs] > len + 1)) { p++; continue; }
                }
                /* (round 5) A CHEAP offset at the next candidate position -- the previous offset R (a repeat code) or offset 1 (a byte run) -- wins
                 * against a fresh offset here even when it is a little shorter (one byte per position of delay): a run of twenty equal bytes
                 * otherwise takes "20 bytes, as 480 bytes ago" instead of "a literal, then 19 at offset 1", and records whose random tail
                 * happens to agree with an older record's take 17 bytes at a fresh offset instead of 16 at the old one.  Runs of 10: 5.5 -> 8.8,
                 * of 20: 10 -> 18, of 100: 30 -> 55; the 8d text: unchanged. */
                if (len && (g_lazy || g_stride == 2)) {
                    const u32 q = p + g_stride, o0 = boff[p - gs];
                    const u32 lq = q < te ? blen[q - gs] : 0, oq = q < te ? boff[q - gs] : 0;
                    if (lq && o0 != R && o0 != 1 && (oq == R || oq == 1) && lq + g_stride >= len) { p = q; continue; }
                }
                /* (round 6) ... and a FAR candidate of the dense tables (an offset of ~18 bits, and the repeat offset is gone) gives way to a cheap
                 * offset up to ZKE_DENSE_AHEAD positions later unless it is ZKE_DENSE_BONUS bytes longer still: fixed-size records whose random
                 * fields happen to agree with a far record's (2 literals + 16 bytes at the repeat offset, not 18 bytes from far away: level 3 came
                 * out 10 % behind level 2), byte runs cut by a tile's end */
                if (len && st->dense && boff[p - gs] > ZKE_WINDOW && boff[p - gs] != R) {
                    int yields = 0;
                    for (u32 j = 1; j <= ZKE_DENSE_AHEAD && p + j < te && !yields; j++) {
                        const u32 lq = blen[p + j - gs], oq = boff[p + j - gs];
                        yields = lq && (oq == R || oq == 1) && lq + j + ZKE_DENSE_BONUS >= len;
                    }
                    if (yields) { p++; continue; }
                }
                if (len) {
                    const u32 off = boff[p - gs];
                    if (len == ZKE_PARCAP) {
                        if (off <= ZKE_WINDOW) len += match_len(base + p + len - off, base + p + len, lim);
                        else {                                                            /* byte by byte through memory; stops where the source leaves the prefix */
                            const u64 a0 = st->abs0 + p - off;
                            while (base + p + len < lim && a0 + len < st->lim && st->pfx[a0 + len] == base[p + len]) len++;
                        }
                    }
                    u32 next = p + len;                                                      /* the walk goes on behind the match */
                    {
                        u32 bk = bback[p - gs];
                        if (bk > p - anchor) bk = p - anchor;                                /* (anchor may lie in an earlier tile: bback stops at ts) */
                        p -= bk; len += bk;
                        if (g_stride == 2) next = (next + 1) & ~1u;
                    }
                    const u32 ll = p - anchor;
                    if (p == ts && ll == 0 && off == prev_off && ((ts - bs) & (ZKE_SEAM - 1))) {
                        /* the tile before ended in a match with this offset: one sequence goes on across the seam (a tile's
                         * matches stop at its end; without this a long copy costs a sequence per 256 bytes).  No joining at
                         * multiples of ZKE_SEAM inside the block: a match length stays below 2^16. */
                        sq[nseq - 1].ml += len;
                    } else {
                        memcpy(lits + nlit, base + anchor, ll); nlit += ll;
                        sq[nseq].ll = ll; sq[nseq].ml = len;
                        sq[nseq].offbase = (ll && off == prev_off) ? 1 : off + 3;
                        nseq++;
                    }
                    prev_off = off; st->probe = off;
                    anchor = p + len; p = next;
                } else p += g_stride;
            }
        }
    }
    memcpy(lits + nlit, base + anchor, be - anchor); nlit += be - anchor;
    *nlit_out = nlit;
    return nseq;
}

/* ------------------------------------------------------------------ frame */
/* what a level buys (zk_enc_device.h zke_minmatch / zke_hash_log / zke_lazy / zke_step): level <= 1: matches of 6+ bytes,
 * 2^14 table entries, greedy parse; 2..5 and 0 (= libzstd's default 3): 5+ bytes, 2^15 entries, lazy parse; 6 and up: the
 * same with lookup steps of 1024 positions instead of 4096 (fresher tables) */
static void set_level(int level, size_t plen)
{
    const int fast = level != 0 && level < 2;
    const int fast2 = fast && plen <= ZKE_WINDOW;         /* zke_fast2(): no long-distance table -> the even-position matcher */
    g_stride = fast2 ? 2 : 1; g_back = 4;                  /* every setting catches its matches up (round 5); only the fast one skips the odd positions */
    g_minmatch = fast && !fast2 ? 6 : 5;
    g_hash_log = fast ? 14 : 15;
    g_lazy = fast ? 0 : 1;
    g_tab32 = fast ? 1 : 0;
    g_step = level >= 6 ? 1024 : ZKE_GROUP_POS;
}
static u32 block_max_of(size_t n, u32 hist)
{
    u32 wlog = 10; while ((1u << wlog) < n && wlog < 17) wlog++;
    if (hist) wlog = 17;
    u32 bmax = (1u << wlog) < ZKE_BLOCK ? (1u << wlog) : ZKE_BLOCK;
    u32 t = 32768; while (t > 4096 && (u64)t * 16 > n) t >>= 1;
    return t < bmax ? t : bmax;
}

/* TEST SUPPORT: the matcher's result for one frame, block by block, in the GPU kernel's packing (ll | ml << 16 |
 * Offset_Value << 32), so that tests/sim (the kernel under a CPU emulator) and the GPU can be compared with it directly.
 * seqs / lits are filled contiguously; returns the number of blocks (or -1: blk_cap too small). */
i64 zko_enc_match_debug(const u8 *src, size_t n, int level, const u8 *prefix, size_t plen, u32 blk_cap, u32 *blk_nseq, u32 *blk_nlit, u64 *seqs, u8 *lits)
{
    if (!prefix) plen = 0;
    set_level(level, plen);
    const u32 hist = (u32)(plen < ZKE_WINDOW ? plen : ZKE_WINDOW) & ~3u;
    if (n == 0) return 0;
    const u32 bmax = block_max_of(n, hist);
    const u32 nblk = (u32)((n + bmax - 1) / bmax);
    if (nblk > blk_cap) return -1;
    enc_state *st = calloc(1, sizeof *st);
    u8 *cat = malloc((size_t)hist + n + 8);
    if (hist) memcpy(cat, prefix + plen - hist, hist);
    memcpy(cat + hist, src, n);
    seq_t *sq = malloc(sizeof(seq_t) * (bmax / 3 + 16));
    const u8 *sbase = cat;
    u32 shist = hist, sstart = 0, send = n < ZKE_SEGMENT ? (u32)n : ZKE_SEGMENT;
    table_seed(st, sbase, shist, shist + send);
    if (prefix) ldm_build(st, prefix, plen);
    else if (ldm_wanted_in_frame(level, plen, n)) { ldm_build_frame(st, src, n); if (dense_wanted(level, plen, n)) dense_build_frame(st, src, n, level); }
    st->abs0 = plen + sstart - shist;
    u64 ns = 0, nl = 0;
    for (u32 k = 0, bs = 0; bs < n; bs += bmax, k++) {
        const u32 be = bs + bmax < n ? bs + bmax : (u32)n;
        if (bs == send) {
            sstart = bs; send = (u64)bs + ZKE_SEGMENT < n ? bs + ZKE_SEGMENT : (u32)n;
            shist = ZKE_WINDOW; sbase = cat + hist + sstart - shist;
            table_seed(st, sbase, shist, shist + (send - sstart));
            st->abs0 = plen + sstart - shist;
        }
        u32 nlit = 0;
        const u32 nseq = find_sequences(st, sbase, shist + (bs - sstart), shist + (be - sstart), shist + (send - sstart), sq, lits + nl, &nlit);
        for (u32 i = 0; i < nseq; i++) seqs[ns + i] = (u64)sq[i].ll | ((u64)sq[i].ml << 16) | ((u64)sq[i].offbase << 32);
        blk_nseq[k] = nseq; blk_nlit[k] = nlit; ns += nseq; nl += nlit;
    }
    free(st->ldm); free(st->dense); free(st); free(cat); free(sq);
    return nblk;
}

static size_t nseq_header(u8 *d, u32 n) { if (n < 128) { d[0] = (u8)n; return 1; } if (n < 0x7F00) { d[0] = (u8)((n >> 8) + 128); d[1] = (u8)n; return 2; } d[0] = 255; d[1] = (u8)(n - 0x7F00); d[2] = (u8)((n - 0x7F00) >> 8); return 3; }

i64 zko_frame_encode_prefix(const u8 *src, size_t n, u8 *dst, size_t cap, int level, int checksum, const u8 *prefix, size_t plen);
/* Encode src[0..n) as ONE zstd frame. level is accepted for API symmetry (single strategy). Returns size or <0. */
i64 zko_frame_encode(const u8 *src, size_t n, u8 *dst, size_t cap, int level, int checksum)
{
    return zko_frame_encode_prefix(src, n, dst, cap, level, checksum, NULL, 0);
}

/* The same against a raw-content prefix (ZSTD_CCtx_refPrefix at the frame start, lib/src/encode.rs:334-338): the last
 * min(plen, 64 KiB window) bytes of the prefix are laid out right before the frame, their positions enter the hash
 * table before the first block, and matches may start there (and run on into the frame).  The frame then declares
 * a 128 KiB window, which covers every offset this matcher can produce. */
i64 zko_frame_encode_prefix(const u8 *src, size_t n, u8 *dst, size_t cap, int level, int checksum, const u8 *prefix, size_t plen)
{
    if (!prefix) plen = 0;
    set_level(level, plen);
    const u32 hist = (u32)(plen < ZKE_WINDOW ? plen : ZKE_WINDOW) & ~3u;   /* a multiple of 4: the GPU's lanes take 4 positions from aligned ring words */
    if (n > 0x40000000u) return -72;
    size_t p = 0;
    if (cap < 32) return -70;
    if (n == 0) {                                                    /* exactly what the reference emits for an empty frame */
        static const u8 e[9] = {0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00};
        memcpy(dst, e, 9); p = 9; dst[4] = checksum ? 0x24 : 0x20;
        if (checksum) { u32 h = (u32)zko_xxh64(src, 0, 0); memcpy(dst + p, &h, 4); p += 4; }
        return (i64)p;
    }
    dst[0] = 0x28; dst[1] = 0xB5; dst[2] = 0x2F; dst[3] = 0xFD; dst[4] = checksum ? 0x04 : 0x00;
    /* Window_Descriptor: smallest power of two >= min(n, 64 KiB reach) but at least 1 KiB; blocks need window >= block size */
    u32 wlog = 10; while ((1u << wlog) < n && wlog < 17) wlog++;     /* <= 128 KiB: offsets never exceed 65535 */
    if (hist) wlog = 17;
    /* long-distance matches reach anywhere into the prefix: the window covers prefix + frame, as the reference's patch mode sets
     * it (cli/src/compress.rs:31-37: WindowLog(ilog2(prefix_len) + 1)) -- libzstd's streaming decoder keeps a prefix reachable
     * only while the frame fits its window */
    if (hist && plen > ZKE_WINDOW) while ((1ull << wlog) < plen + n && wlog < 27) wlog++;          /* long-distance offsets stay below 2^27 */
    if (ldm_wanted_in_frame(level, plen, n)) while ((1ull << wlog) < n && wlog < 27) wlog++;       /* in-frame far history: the window covers the frame */
    dst[5] = (u8)((wlog - 10) << 3);
    p = 6;
    enc_state *st = calloc(1, sizeof *st);
    const u8 *msrc = src;                                            /* what the matcher sees: [prefix tail | frame] */
    u8 *cat = NULL;
    if (hist) {
        cat = malloc((size_t)hist + n + 8);
        memcpy(cat, prefix + plen - hist, hist); memcpy(cat + hist, src, n);
        msrc = cat;
    }
    table_seed(st, msrc, hist, (u32)(hist + n < ZKE_SEGMENT + hist ? hist + n : ZKE_SEGMENT + hist));
    if (prefix) ldm_build(st, prefix, plen);
    else if (ldm_wanted_in_frame(level, plen, n)) { ldm_build_frame(st, src, n); if (dense_wanted(level, plen, n)) dense_build_frame(st, src, n, level); }
    st->abs0 = plen - hist;
    i64 rc = 0;
    u32 bmax = (1u << wlog) < ZKE_BLOCK ? (1u << wlog) : ZKE_BLOCK;
    /* blocks are cut smaller than the format's maximum on purpose: a block's sequence bitstream is one serial chain
     * for the decoder, so more, shorter blocks = more parallel chains (32 KiB for large frames, >= 16 blocks per small frame) */
    { u32 t = 32768; while (t > 4096 && (u64)t * 16 > n) t >>= 1; if (t < bmax) bmax = t; }
    const u32 nblk = (u32)((n + bmax - 1) / bmax);
    /* pass 1: sequences + literals of every block; code histograms of the frame */
    seq_t *sq = malloc(sizeof(seq_t) * (n / 3 + 8 * (size_t)nblk + 8));
    u8 *lits = malloc(n + 64), *body = malloc(ZKE_BLOCK * 2);
    u32 *bseq = malloc(sizeof(u32) * (nblk + 1)), *blit = malloc(sizeof(u32) * (nblk + 1)), *bnlit = malloc(sizeof(u32) * (nblk + 1));
    u32 hll[36] = {0}, hof[32] = {0}, hml[53] = {0}, nseq_frame = 0, nlit_frame = 0;
    /* The matcher works on segments of ZKE_SEGMENT bytes (one GPU workgroup each, so that a frame larger than that is not
     * one serial job): a segment after the first starts with an empty table that receives the positions of the ZKE_WINDOW bytes
     * before it -- exactly what a prefix does for a frame -- its positions count from that history's start, and nothing of
     * it looks past its own end.  Frames up to ZKE_SEGMENT bytes are one segment: nothing changes for them. */
    const u8 *sbase = msrc;                                          /* the segment's position 0 */
    u32 shist = hist, sstart = 0, send = n < ZKE_SEGMENT ? (u32)n : ZKE_SEGMENT;
    for (u32 k = 0, bs = 0; bs < n; bs += bmax, k++) {
        u32 be = bs + bmax < n ? bs + bmax : (u32)n;
        u32 nlit = 0;
        if (bs == send) {                                            /* next segment */
            sstart = bs; send = (u64)bs + ZKE_SEGMENT < n ? bs + ZKE_SEGMENT : (u32)n;
            shist = ZKE_WINDOW;
            sbase = msrc + hist + sstart - shist;
            table_seed(st, sbase, shist, shist + (send - sstart));
            st->abs0 = plen + sstart - shist;
        }
        bseq[k] = nseq_frame; blit[k] = nlit_frame;
        u32 nseq = find_sequences(st, sbase, shist + (bs - sstart), shist + (be - sstart), shist + (send - sstart), sq + nseq_frame, lits + nlit_frame, &nlit);
        for (u32 i = 0; i < nseq; i++) { const seq_t *q = &sq[nseq_frame + i]; hll[ll_code(q->ll)]++; hml[ml_code(q->ml - 3)]++; hof[hb32(q->offbase)]++; }
        bnlit[k] = nlit; nseq_frame += nseq; nlit_frame += nlit;
    }
    bseq[nblk] = nseq_frame;
    frame_tables *ft = malloc(sizeof *ft);
    frame_tables_build(ft, hll, hof, hml, nseq_frame);
    const u8 modes_rep = (u8)((ft->custom[0] ? 3 << 6 : 0) | (ft->custom[1] ? 3 << 4 : 0) | (ft->custom[2] ? 3 << 2 : 0));
    const u8 modes_def = (u8)((ft->custom[0] ? 2 << 6 : 0) | (ft->custom[1] ? 2 << 4 : 0) | (ft->custom[2] ? 2 << 2 : 0));
    int defined = 0;                                                 /* the tables have been transmitted */
    /* pass 2: every block with the frame's tables; whether a block is emitted compressed is decided on its Repeat_Mode
     * form; the first one that is (and has sequences) becomes the defining block and grows by the descriptions */
    for (u32 k = 0, bs = 0; bs < n && rc == 0; bs += bmax, k++) {
        u32 be = bs + bmax < n ? bs + bmax : (u32)n;
        u32 bsz = be - bs, last = be == n;
        const u32 nlit = bnlit[k], nseq = bseq[k + 1] - bseq[k];
        size_t b = encode_literals(lits + blit[k], nlit, body, ZKE_BLOCK * 2);
        size_t total = 0, modes_at = 0;
        if (b) {
            b += nseq_header(body + b, nseq);
            if (nseq) {
                modes_at = b;
                body[b++] = modes_rep;
                size_t s = encode_sequences(ft, sq + bseq[k], nseq, body + b, ZKE_BLOCK * 2 - b);
                if (s) total = b + s;
            } else total = b;
        }
        int rle = 1;
        for (u32 i = 1; i < bsz && rle; i++) rle = src[bs + i] == src[bs];
        if (p + 3 + bsz + 8 + 256 > cap) { rc = -70; break; }
        if (rle && bsz > 1) {                                        /* RLE block */
            u32 h = last | (1 << 1) | (bsz << 3); dst[p] = (u8)h; dst[p + 1] = (u8)(h >> 8); dst[p + 2] = (u8)(h >> 16); dst[p + 3] = src[bs]; p += 4;
        } else if (total && total < bsz) {
            const int def = nseq && !defined && modes_rep;
            const size_t extra = def ? ft->dlen[0] + ft->dlen[1] + ft->dlen[2] : 0;
            u32 h = last | (2 << 1) | ((u32)(total + extra) << 3); dst[p] = (u8)h; dst[p + 1] = (u8)(h >> 8); dst[p + 2] = (u8)(h >> 16); p += 3;
            if (def) {
                memcpy(dst + p, body, modes_at); p += modes_at;
                dst[p++] = modes_def;
                for (int t = 0; t < 3; t++) { memcpy(dst + p, ft->desc[t], ft->dlen[t]); p += ft->dlen[t]; }
                memcpy(dst + p, body + modes_at + 1, total - modes_at - 1); p += total - modes_at - 1;
                defined = 1;
            } else { memcpy(dst + p, body, total); p += total; }
        } else {                                                     /* raw block */
            u32 h = last | (bsz << 3); dst[p] = (u8)h; dst[p + 1] = (u8)(h >> 8); dst[p + 2] = (u8)(h >> 16); p += 3;
            memcpy(dst + p, src + bs, bsz); p += bsz;
        }
    }
    free(ft); free(bseq); free(blit); free(bnlit);
    if (rc == 0 && checksum) { if (p + 4 > cap) rc = -70; else { u32 h = (u32)zko_xxh64(src, n, 0); memcpy(dst + p, &h, 4); p += 4; } }
    free(st->ldm); free(st->dense); free(st); free(sq); free(lits); free(body); free(cat);
    return rc ? rc : (i64)p;
}
