/*
 * zstd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the arithmetic that sits under
 * zeekstd's hot path.  In the reference that arithmetic is NOT in the repo: it
 * is libzstd 1.5.7 (C), reached through zstd-safe 7.2.4 / zstd-sys
 * 2.0.16+zstd.1.5.7 (reference Cargo.toml:16, Cargo.lock:1183-1199) from the
 * call sites lib/src/encode.rs:340-346, 442-464 and lib/src/decode.rs:242-256.
 * libzstd's source is absent from /root/reference, so this file restates the
 * *published* Zstandard frame format (RFC 8878; the working notes are
 * SURVEY.md Appendix A) and is pinned against outputs of the real libzstd
 * 1.5.7 binary generated in the build container (tests/golden/, made by
 * tools/make_goldens.py) and against the golden bytes of SURVEY.md Appendix B.
 *
 * Pinned further (round 5) on frames NO libzstd encoder writes: tests/helpers/zstd_gen.py
 * draws valid frames from everything the format leaves open, the real libzstd 1.5.7 decodes
 * them, and this decoder has to agree (tests/test_generated_frames.py).  On DAMAGED input it
 * follows the format where libzstd's decoder is laxer -- a Huffman stream must end exactly
 * on its first bit (RFC 8878 4.2.2), the reserved mode bits must be zero (checked by
 * libzstd only since 1.5.6), a block regenerates at most min(Window_Size, 128 KiB):
 * tests/test_oracle.py::test_where_the_oracle_is_stricter_than_libzstd pins those deltas.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.  The product (zeekstd_amd/csrc) never links it.
 *
 * Contents
 *   zko_xxh64            XXH64 (content checksum, encode.rs:163-167 enables it)
 *   zko_frame_decode     full-format zstd frame decoder (what ZSTD_decompressStream does
 *                        for decode.rs:242-256)
 *   zko_frame_encode     a small valid zstd frame encoder (greedy single-hash LZ +
 *                        Huffman literals + FSE sequences; what ZSTD_compressStream2
 *                        does for encode.rs:340-346/442-464 -- payload bytes are
 *                        unpinned by the reference, only validity/round trip)
 *   zko_frame_info       frame walker: block types/sizes (used by tests to assert feature coverage)
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;

/* ZSTD_ErrorCode values (zstd_errors.h); returned negated. */
enum {
    ZKO_OK = 0,
    ZKO_E_GENERIC = 1,
    ZKO_E_PREFIX_UNKNOWN = 10,
    ZKO_E_FRAMEPARAM_UNSUPPORTED = 14,
    ZKO_E_WINDOW_TOO_LARGE = 16,
    ZKO_E_CORRUPTION = 20,
    ZKO_E_CHECKSUM_WRONG = 22,
    ZKO_E_DICT_WRONG = 32,
    ZKO_E_DST_TOO_SMALL = 70,
    ZKO_E_SRC_SIZE_WRONG = 72,
};
#define ERR(c) (-(i64)(c))

/* ------------------------------------------------------------------ XXH64 */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL

static inline u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
static inline u64 rd64(const u8 *p) { u64 v; memcpy(&v, p, 8); return v; }
static inline u32 rd32(const u8 *p) { u32 v; memcpy(&v, p, 4); return v; }
static inline u32 rd24(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }
static inline u16 rd16(const u8 *p) { u16 v; memcpy(&v, p, 2); return v; }
static inline u64 xxh_round(u64 acc, u64 x) { return rotl64(acc + x * P2, 31) * P1; }

u64 zko_xxh64(const u8 *p, size_t len, u64 seed)
{
    const u8 *end = p + len;
    u64 h;
    if (len >= 32) {
        u64 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const u8 *lim = end - 32;
        do {
            v1 = xxh_round(v1, rd64(p));
            v2 = xxh_round(v2, rd64(p + 8));
            v3 = xxh_round(v3, rd64(p + 16));
            v4 = xxh_round(v4, rd64(p + 24));
            p += 32;
        } while (p <= lim);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = (h ^ xxh_round(0, v1)) * P1 + P4;
        h = (h ^ xxh_round(0, v2)) * P1 + P4;
        h = (h ^ xxh_round(0, v3)) * P1 + P4;
        h = (h ^ xxh_round(0, v4)) * P1 + P4;
    } else {
        h = seed + P5;
    }
    h += (u64)len;
    while (p + 8 <= end) { h ^= xxh_round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (u64)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (u64)(*p) * P5; h = rotl64(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* -------------------------------------------------------- bit readers */
static inline int highbit32(u32 v) { return 31 - __builtin_clz(v); }

/* forward LSB-first reader (FSE table descriptions, A.5) */
typedef struct { const u8 *p; size_t len; size_t bitpos; } fwd_t;
static u32 fwd_peek(const fwd_t *f, int n)
{
    u64 v = 0;
    size_t byte = f->bitpos >> 3;
    for (int i = 0; i < 8 && byte + i < f->len; i++) v |= (u64)f->p[byte + i] << (8 * i);
    return (u32)((v >> (f->bitpos & 7)) & ((1ULL << n) - 1));
}

/* backward reader (A.7).  bitpos = number of unread bits counted from byte 0. */
typedef struct { const u8 *p; size_t len; i64 bitpos; } bwd_t;
static int bwd_init(bwd_t *b, const u8 *p, size_t len)
{
    if (len == 0 || p[len - 1] == 0) return -1;
    b->p = p; b->len = len;
    b->bitpos = (i64)(len - 1) * 8 + highbit32(p[len - 1]);
    return 0;
}
/* next n (<=32) bits, first-consumed bit is the MSB of the result, zero-filled past the start */
static u32 bwd_peek(const bwd_t *b, int n)
{
    if (n == 0) return 0;
    i64 lo = b->bitpos - n;
    int shl = 0;
    if (lo < 0) { shl = (int)(-lo); lo = 0; n -= shl; if (n <= 0) return 0; }
    size_t byte = (size_t)(lo >> 3);
    u64 v = 0;
    if (byte + 8 <= b->len) v = rd64(b->p + byte);
    else for (size_t i = 0; byte + i < b->len; i++) v |= (u64)b->p[byte + i] << (8 * i);
    v = (v >> (lo & 7)) & ((1ULL << n) - 1);
    return (u32)(v << shl);
}
static inline u32 bwd_read(bwd_t *b, int n) { u32 v = bwd_peek(b, n); b->bitpos -= n; return v; }

/* ------------------------------------------------------------ FSE tables */
typedef struct { u8 sym; u8 nb; u16 base; } fse_cell;
typedef struct { int al; fse_cell cell[512]; } fse_table;

/* A.5: read normalised counts.  Returns bytes consumed or <0. */
static i64 fse_read_ncount(const u8 *src, size_t len, int max_sym, int max_al, short *norm, int *nsym_out, int *al_out)
{
    if (len == 0) return ERR(ZKO_E_CORRUPTION);
    fwd_t f = { src, len, 0 };
    int al = (int)fwd_peek(&f, 4) + 5; f.bitpos += 4;
    if (al > max_al) return ERR(ZKO_E_CORRUPTION);
    int remaining = (1 << al) + 1, threshold = 1 << al, nb = al + 1, sym = 0;
    while (remaining > 1 && sym <= max_sym) {
        int max = 2 * threshold - 1 - remaining;
        u32 v = fwd_peek(&f, nb);
        int cnt;
        if ((int)(v & (threshold - 1)) < max) { cnt = (int)(v & (threshold - 1)); f.bitpos += nb - 1; }
        else { cnt = (int)(v & (2 * threshold - 1)); if (cnt >= threshold) cnt -= max; f.bitpos += nb; }
        cnt -= 1;
        remaining -= cnt < 0 ? -cnt : cnt;
        norm[sym++] = (short)cnt;
        if (cnt == 0) {
            for (;;) {
                u32 rep = fwd_peek(&f, 2); f.bitpos += 2;
                for (u32 i = 0; i < rep; i++) { if (sym > max_sym) return ERR(ZKO_E_CORRUPTION); norm[sym++] = 0; }
                if (rep != 3) break;
            }
        }
        while (remaining < threshold) { nb--; threshold >>= 1; }
        if (f.bitpos > len * 8 + 16) return ERR(ZKO_E_CORRUPTION);
    }
    if (remaining != 1) return ERR(ZKO_E_CORRUPTION);
    if (sym > max_sym + 1) return ERR(ZKO_E_CORRUPTION);
    size_t used = (f.bitpos + 7) >> 3;
    if (used > len) return ERR(ZKO_E_CORRUPTION);
    *nsym_out = sym; *al_out = al;
    return (i64)used;
}

/* A.6: build the decode table. */
static int fse_build(fse_table *t, const short *norm, int nsym, int al)
{
    int size = 1 << al, high = size - 1;
    u16 next[64];
    t->al = al;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) { t->cell[high--].sym = (u8)s; next[s] = 1; }
        else next[s] = (u16)norm[s];
    }
    int step = (size >> 1) + (size >> 3) + 3, pos = 0, mask = size - 1;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            t->cell[pos].sym = (u8)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return -1;
    for (int i = 0; i < size; i++) {
        int s = t->cell[i].sym;
        u32 x = next[s]++;
        int nb = al - highbit32(x);
        t->cell[i].nb = (u8)nb;
        t->cell[i].base = (u16)((x << nb) - size);
    }
    return 0;
}
static void fse_build_rle(fse_table *t, int sym) { t->al = 0; t->cell[0].sym = (u8)sym; t->cell[0].nb = 0; t->cell[0].base = 0; }

static const short LL_DEF[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const short OF_DEF[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const short ML_DEF[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};

static const u32 LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
static const u8 LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const u32 ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
static const u8 ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};

/* -------------------------------------------------------- Huffman tables */
typedef struct { u8 sym; u8 nb; } huf_cell;
typedef struct { int maxbits; int valid; huf_cell cell[1 << 11]; } huf_table;

/* A.4: parse tree description, build decode table. Returns bytes consumed or <0. */
static i64 huf_read_table(huf_table *ht, const u8 *src, size_t len)
{
    u8 w[256];
    int n = 0;
    if (len < 1) return ERR(ZKO_E_CORRUPTION);
    int h = src[0];
    size_t used;
    if (h >= 128) {
        n = h - 127;
        used = 1 + (size_t)(n + 1) / 2;
        if (used > len) return ERR(ZKO_E_CORRUPTION);
        for (int i = 0; i < n; i++) { u8 b = src[1 + i / 2]; w[i] = (i & 1) ? (b & 15) : (b >> 4); }
    } else {
        used = 1 + (size_t)h;
        if (used > len || h < 2) return ERR(ZKO_E_CORRUPTION);
        short norm[16]; int nsym, al;
        i64 r = fse_read_ncount(src + 1, (size_t)h, 11, 6, norm, &nsym, &al);   /* weights 0..11 -> 12 symbols */
        if (r < 0) return r;
        fse_table ft;
        if (fse_build(&ft, norm, nsym, al) != 0) return ERR(ZKO_E_CORRUPTION);
        bwd_t b;
        if (bwd_init(&b, src + 1 + r, (size_t)h - (size_t)r) != 0) return ERR(ZKO_E_CORRUPTION);
        u32 s1 = bwd_read(&b, al), s2 = bwd_read(&b, al);
        for (;;) {
            if (n >= 254) return ERR(ZKO_E_CORRUPTION);
            w[n++] = ft.cell[s1].sym;
            if (b.bitpos < (i64)ft.cell[s1].nb) {          /* updating s1 would over-read */
                b.bitpos -= ft.cell[s1].nb;
                if (n >= 255) return ERR(ZKO_E_CORRUPTION);
                w[n++] = ft.cell[s2].sym; break;
            }
            s1 = ft.cell[s1].base + bwd_read(&b, ft.cell[s1].nb);
            if (n >= 254) return ERR(ZKO_E_CORRUPTION);
            w[n++] = ft.cell[s2].sym;
            if (b.bitpos < (i64)ft.cell[s2].nb) {
                b.bitpos -= ft.cell[s2].nb;
                if (n >= 255) return ERR(ZKO_E_CORRUPTION);
                w[n++] = ft.cell[s1].sym; break;
            }
            s2 = ft.cell[s2].base + bwd_read(&b, ft.cell[s2].nb);
        }
    }
    /* implied last weight */
    u32 sum = 0;
    for (int i = 0; i < n; i++) { if (w[i] > 11) return ERR(ZKO_E_CORRUPTION); if (w[i]) sum += 1u << (w[i] - 1); }
    if (sum == 0) return ERR(ZKO_E_CORRUPTION);
    int maxbits = highbit32(sum) + 1;
    if (maxbits > 11) return ERR(ZKO_E_CORRUPTION);
    u32 rest = (1u << maxbits) - sum;
    if (rest & (rest - 1)) return ERR(ZKO_E_CORRUPTION);
    w[n++] = (u8)(highbit32(rest) + 1);
    /* canonical fill: weight 1 first, symbols ascending */
    u32 pos = 0;
    for (int wt = 1; wt <= maxbits; wt++) {
        for (int s = 0; s < n; s++) if (w[s] == wt) {
            u32 cnt = 1u << (wt - 1);
            for (u32 k = 0; k < cnt; k++) { ht->cell[pos + k].sym = (u8)s; ht->cell[pos + k].nb = (u8)(maxbits + 1 - wt); }
            pos += cnt;
        }
    }
    if (pos != (1u << maxbits)) return ERR(ZKO_E_CORRUPTION);
    ht->maxbits = maxbits; ht->valid = 1;
    return (i64)used;
}

static int huf_decode_stream(const huf_table *ht, const u8 *src, size_t len, u8 *dst, size_t n)
{
    bwd_t b;
    if (bwd_init(&b, src, len) != 0) return -1;
    int mb = ht->maxbits;
    for (size_t i = 0; i < n; i++) {
        huf_cell c = ht->cell[bwd_peek(&b, mb)];
        dst[i] = c.sym; b.bitpos -= c.nb;
    }
    return b.bitpos == 0 ? 0 : -1;
}

/* ------------------------------------------------------------ frame decode */
typedef struct {
    huf_table huf;
    fse_table ll, of, ml;
    int ll_valid, of_valid, ml_valid;
    u32 rep[3];
    u8 *lit;              /* 128 KiB + slack literal buffer */
} dstate;

/* per-frame stats for tests (feature coverage assertions) */
typedef struct {
    u32 n_blocks, n_raw, n_rle, n_comp;
    u32 lit_raw, lit_rle, lit_huf4, lit_huf1, lit_treeless;
    u32 mode_count[3][4];          /* [LL,OF,ML][predef,rle,fse,repeat] */
    u64 n_seq;
    u32 window_size, has_checksum, single_segment;
    u64 fcs; u32 fcs_present;
    u32 max_offset;
} zko_frame_stats;

static i64 decode_literals(dstate *st, const u8 *src, size_t len, size_t *lit_size, const u8 **lit_ptr, zko_frame_stats *fs)
{
    if (len < 1) return ERR(ZKO_E_CORRUPTION);
    int type = src[0] & 3, sf = (src[0] >> 2) & 3;
    size_t hdr, regen, comp = 0;
    int streams = 4;
    if (type < 2) {
        if (sf == 0 || sf == 2) { hdr = 1; regen = src[0] >> 3; }
        else if (sf == 1) { hdr = 2; if (len < 2) return ERR(ZKO_E_CORRUPTION); regen = (src[0] >> 4) + ((size_t)src[1] << 4); }
        else { hdr = 3; if (len < 3) return ERR(ZKO_E_CORRUPTION); regen = (src[0] >> 4) + ((size_t)src[1] << 4) + ((size_t)src[2] << 12); }
        if (regen > 131072) return ERR(ZKO_E_CORRUPTION);
        if (type == 0) {
            if (hdr + regen > len) return ERR(ZKO_E_CORRUPTION);
            *lit_ptr = src + hdr; *lit_size = regen; if (fs) fs->lit_raw++;
            return (i64)(hdr + regen);
        }
        if (hdr + 1 > len) return ERR(ZKO_E_CORRUPTION);
        memset(st->lit, src[hdr], regen);
        *lit_ptr = st->lit; *lit_size = regen; if (fs) fs->lit_rle++;
        return (i64)(hdr + 1);
    }
    u64 v;
    if (sf == 0 || sf == 1) { hdr = 3; if (len < 3) return ERR(ZKO_E_CORRUPTION); v = rd24(src); regen = (v >> 4) & 0x3ff; comp = (v >> 14) & 0x3ff; if (sf == 0) streams = 1; }
    else if (sf == 2) { hdr = 4; if (len < 4) return ERR(ZKO_E_CORRUPTION); v = rd32(src); regen = (v >> 4) & 0x3fff; comp = (v >> 18) & 0x3fff; }
    else { hdr = 5; if (len < 5) return ERR(ZKO_E_CORRUPTION); v = (u64)rd32(src) | ((u64)src[4] << 32); regen = (v >> 4) & 0x3ffff; comp = (v >> 22) & 0x3ffff; }
    if (regen > 131072 || hdr + comp > len) return ERR(ZKO_E_CORRUPTION);
    const u8 *p = src + hdr; size_t rem = comp;
    if (type == 2) {
        i64 r = huf_read_table(&st->huf, p, rem);
        if (r < 0) return r;
        p += r; rem -= (size_t)r;
        if (fs) { if (streams == 1) fs->lit_huf1++; else fs->lit_huf4++; }
    } else {
        if (!st->huf.valid) return ERR(ZKO_E_CORRUPTION);
        if (fs) fs->lit_treeless++;
    }
    if (streams == 1) {
        if (huf_decode_stream(&st->huf, p, rem, st->lit, regen) != 0) return ERR(ZKO_E_CORRUPTION);
    } else {
        if (rem < 6) return ERR(ZKO_E_CORRUPTION);
        size_t s1 = rd16(p), s2 = rd16(p + 2), s3 = rd16(p + 4);
        if (6 + s1 + s2 + s3 > rem) return ERR(ZKO_E_CORRUPTION);
        size_t s4 = rem - 6 - s1 - s2 - s3;
        size_t q = (regen + 3) / 4;
        if (3 * q > regen) return ERR(ZKO_E_CORRUPTION);
        const u8 *b = p + 6;
        if (huf_decode_stream(&st->huf, b, s1, st->lit, q) ||
            huf_decode_stream(&st->huf, b + s1, s2, st->lit + q, q) ||
            huf_decode_stream(&st->huf, b + s1 + s2, s3, st->lit + 2 * q, q) ||
            huf_decode_stream(&st->huf, b + s1 + s2 + s3, s4, st->lit + 3 * q, regen - 3 * q))
            return ERR(ZKO_E_CORRUPTION);
    }
    *lit_ptr = st->lit; *lit_size = regen;
    return (i64)(hdr + comp);
}

static i64 setup_seq_table(fse_table *t, int *valid, int mode, const u8 *src, size_t len,
                           const short *def, int def_n, int def_al, int max_sym, int max_al)
{
    if (mode == 0) { if (fse_build(t, def, def_n, def_al)) return ERR(ZKO_E_CORRUPTION); *valid = 1; return 0; }
    if (mode == 1) { if (len < 1 || src[0] > max_sym) return ERR(ZKO_E_CORRUPTION); fse_build_rle(t, src[0]); *valid = 1; return 1; }
    if (mode == 2) {
        short norm[64]; int nsym, al;
        i64 r = fse_read_ncount(src, len, max_sym, max_al, norm, &nsym, &al);
        if (r < 0) return r;
        if (fse_build(t, norm, nsym, al)) return ERR(ZKO_E_CORRUPTION);
        *valid = 1; return r;
    }
    if (!*valid) return ERR(ZKO_E_CORRUPTION);
    return 0;
}

/* decode one compressed block body into dst+pos.  frame_start = dst; prefix (raw-content dictionary,
 * ZSTD_DCtx_refPrefix: lib/src/decode.rs:212-214) sits virtually right before it: a match may start up to plen
 * bytes before the frame.  With a prefix only availability bounds an offset (what libzstd's ZSTD_execSequence
 * checks); without one the frame's window does too. */
static i64 decode_compressed_block(dstate *st, const u8 *src, size_t len, u8 *dst, size_t pos, size_t cap,
                                   u32 window, zko_frame_stats *fs, const u8 *prefix, size_t plen)
{
    const u8 *lit; size_t nlit;
    i64 r = decode_literals(st, src, len, &nlit, &lit, fs);
    if (r < 0) return r;
    const u8 *p = src + r; size_t rem = len - (size_t)r;
    if (rem < 1) return ERR(ZKO_E_CORRUPTION);
    u32 nseq; size_t h;
    if (p[0] < 128) { nseq = p[0]; h = 1; }
    else if (p[0] < 255) { if (rem < 2) return ERR(ZKO_E_CORRUPTION); nseq = ((u32)(p[0] - 128) << 8) + p[1]; h = 2; }
    else { if (rem < 3) return ERR(ZKO_E_CORRUPTION); nseq = (u32)p[1] + ((u32)p[2] << 8) + 0x7F00; h = 3; }
    p += h; rem -= h;
    size_t out = pos;
    if (nseq == 0) {
        if (rem != 0) return ERR(ZKO_E_CORRUPTION);
        if (out + nlit > cap) return ERR(ZKO_E_DST_TOO_SMALL);
        memcpy(dst + out, lit, nlit);
        return (i64)nlit;
    }
    if (rem < 1) return ERR(ZKO_E_CORRUPTION);
    int modes = p[0]; p++; rem--;
    if (modes & 3) return ERR(ZKO_E_CORRUPTION);
    int m_ll = modes >> 6, m_of = (modes >> 4) & 3, m_ml = (modes >> 2) & 3;
    if (fs) { fs->mode_count[0][m_ll]++; fs->mode_count[1][m_of]++; fs->mode_count[2][m_ml]++; fs->n_seq += nseq; }
    r = setup_seq_table(&st->ll, &st->ll_valid, m_ll, p, rem, LL_DEF, 36, 6, 35, 9); if (r < 0) return r; p += r; rem -= (size_t)r;
    r = setup_seq_table(&st->of, &st->of_valid, m_of, p, rem, OF_DEF, 29, 5, 31, 8); if (r < 0) return r; p += r; rem -= (size_t)r;
    r = setup_seq_table(&st->ml, &st->ml_valid, m_ml, p, rem, ML_DEF, 53, 6, 52, 9); if (r < 0) return r; p += r; rem -= (size_t)r;
    bwd_t b;
    if (bwd_init(&b, p, rem) != 0) return ERR(ZKO_E_CORRUPTION);
    u32 sl = bwd_read(&b, st->ll.al), so = bwd_read(&b, st->of.al), sm = bwd_read(&b, st->ml.al);
    size_t lpos = 0;
    for (u32 i = 0; i < nseq; i++) {
        fse_cell cl = st->ll.cell[sl], co = st->of.cell[so], cm = st->ml.cell[sm];
        if (co.sym > 31 || cl.sym > 35 || cm.sym > 52) return ERR(ZKO_E_CORRUPTION);
        u32 ofv = (1u << co.sym) + bwd_read(&b, co.sym);
        u32 ml = ML_BASE[cm.sym] + bwd_read(&b, ML_BITS[cm.sym]);
        u32 ll = LL_BASE[cl.sym] + bwd_read(&b, LL_BITS[cl.sym]);
        u32 off;
        if (ofv > 3) { off = ofv - 3; st->rep[2] = st->rep[1]; st->rep[1] = st->rep[0]; st->rep[0] = off; }
        else {
            u32 idx = ofv - 1 + (ll == 0);
            if (idx == 0) off = st->rep[0];
            else {
                off = idx == 3 ? st->rep[0] - 1 : st->rep[idx];
                if (off == 0) return ERR(ZKO_E_CORRUPTION);
                if (idx > 1) st->rep[2] = st->rep[1];
                st->rep[1] = st->rep[0]; st->rep[0] = off;
            }
        }
        if (i + 1 < nseq) {
            sl = cl.base + bwd_read(&b, cl.nb);
            sm = cm.base + bwd_read(&b, cm.nb);
            so = co.base + bwd_read(&b, co.nb);
        }
        if (b.bitpos < 0) return ERR(ZKO_E_CORRUPTION);
        if (lpos + ll > nlit) return ERR(ZKO_E_CORRUPTION);
        if (out + ll + ml > cap) return ERR(ZKO_E_DST_TOO_SMALL);
        memcpy(dst + out, lit + lpos, ll); out += ll; lpos += ll;
        if (off > out + plen || (plen == 0 && off > window)) return ERR(ZKO_E_CORRUPTION);
        if (fs && off > fs->max_offset) fs->max_offset = off;
        for (u32 k = 0; k < ml; k++) {
            const size_t q = out + k;                       /* source = q - off, possibly inside the prefix */
            dst[q] = q >= off ? dst[q - off] : prefix[plen - (off - q)];
        }
        out += ml;
    }
    if (b.bitpos != 0) return ERR(ZKO_E_CORRUPTION);
    size_t tail = nlit - lpos;
    if (out + tail > cap) return ERR(ZKO_E_DST_TOO_SMALL);
    memcpy(dst + out, lit + lpos, tail); out += tail;
    if (out - pos > 131072) return ERR(ZKO_E_CORRUPTION);
    return (i64)(out - pos);
}

/*
 * Decode ONE frame (zstd or skippable) at src.
 * Returns decompressed size (>=0) or -ZSTD_ErrorCode.  *consumed = bytes of src used.
 * verify != 0: check the content checksum when present.
 */
i64 zko_frame_decode_prefix(const u8 *src, size_t src_size, u8 *dst, size_t dst_cap, size_t *consumed,
                            int verify, zko_frame_stats *fs, const u8 *prefix, size_t plen);
i64 zko_frame_decode(const u8 *src, size_t src_size, u8 *dst, size_t dst_cap, size_t *consumed,
                     int verify, zko_frame_stats *fs)
{
    return zko_frame_decode_prefix(src, src_size, dst, dst_cap, consumed, verify, fs, NULL, 0);
}

/* the same with a raw-content prefix referenced for this frame (ZSTD_DCtx_refPrefix semantics) */
i64 zko_frame_decode_prefix(const u8 *src, size_t src_size, u8 *dst, size_t dst_cap, size_t *consumed,
                            int verify, zko_frame_stats *fs, const u8 *prefix, size_t plen)
{
    if (!prefix) plen = 0;
    if (fs) memset(fs, 0, sizeof *fs);
    if (src_size < 4) return ERR(ZKO_E_SRC_SIZE_WRONG);
    u32 magic = rd32(src);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
        if (src_size < 8) return ERR(ZKO_E_SRC_SIZE_WRONG);
        u64 sz = rd32(src + 4);
        if (8 + sz > src_size) return ERR(ZKO_E_SRC_SIZE_WRONG);
        *consumed = 8 + (size_t)sz; return 0;
    }
    if (magic != 0xFD2FB528u) return ERR(ZKO_E_PREFIX_UNKNOWN);
    if (src_size < 6) return ERR(ZKO_E_SRC_SIZE_WRONG);
    u8 fhd = src[4];
    int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, cks = (fhd >> 2) & 1, did = fhd & 3;
    if (fhd & 0x08) return ERR(ZKO_E_FRAMEPARAM_UNSUPPORTED);
    size_t p = 5;
    u64 window = 0;
    if (!single) {
        u8 wd = src[p++];
        int e = wd >> 3, m = wd & 7;
        if (10 + e > 31) return ERR(ZKO_E_WINDOW_TOO_LARGE);
        window = 1ULL << (10 + e); window += (window >> 3) * (u64)m;
    }
    static const int did_len[4] = {0, 1, 2, 4};
    static const int fcs_len[4] = {0, 2, 4, 8};
    int dl = did_len[did], fl = fcs_len[fcs_flag];
    if (fcs_flag == 0 && single) fl = 1;
    if (p + (size_t)dl + (size_t)fl > src_size) return ERR(ZKO_E_SRC_SIZE_WRONG);
    if (dl) { u32 id = 0; for (int i = 0; i < dl; i++) id |= (u32)src[p + i] << (8 * i); if (id) return ERR(ZKO_E_DICT_WRONG); }
    p += (size_t)dl;
    u64 fcs = 0;
    if (fl == 1) fcs = src[p]; else if (fl == 2) fcs = (u64)rd16(src + p) + 256; else if (fl == 4) fcs = rd32(src + p); else if (fl == 8) fcs = rd64(src + p);
    p += (size_t)fl;
    if (single) window = fcs;
    if (fs) { fs->window_size = (u32)window; fs->has_checksum = (u32)cks; fs->single_segment = (u32)single; fs->fcs = fcs; fs->fcs_present = fl != 0; }
    u32 block_max = window < 131072 ? (u32)window : 131072;

    dstate *st = (dstate *)malloc(sizeof *st);
    if (!st) return ERR(ZKO_E_GENERIC);
    st->lit = (u8 *)malloc(131072 + 32);
    st->huf.valid = 0; st->ll_valid = st->of_valid = st->ml_valid = 0;
    st->rep[0] = 1; st->rep[1] = 4; st->rep[2] = 8;
    size_t out = 0;
    i64 rc = 0;
    for (;;) {
        if (p + 3 > src_size) { rc = ERR(ZKO_E_SRC_SIZE_WRONG); break; }
        u32 bh = rd24(src + p); p += 3;
        int last = bh & 1, type = (bh >> 1) & 3; u32 bsize = bh >> 3;
        if (fs) fs->n_blocks++;
        if (type == 3) { rc = ERR(ZKO_E_CORRUPTION); break; }
        if (type == 0) {
            if (bsize > block_max) { rc = ERR(ZKO_E_CORRUPTION); break; }
            if (p + bsize > src_size) { rc = ERR(ZKO_E_SRC_SIZE_WRONG); break; }
            if (out + bsize > dst_cap) { rc = ERR(ZKO_E_DST_TOO_SMALL); break; }
            memcpy(dst + out, src + p, bsize); out += bsize; p += bsize; if (fs) fs->n_raw++;
        } else if (type == 1) {
            if (bsize > block_max) { rc = ERR(ZKO_E_CORRUPTION); break; }
            if (p + 1 > src_size) { rc = ERR(ZKO_E_SRC_SIZE_WRONG); break; }
            if (out + bsize > dst_cap) { rc = ERR(ZKO_E_DST_TOO_SMALL); break; }
            memset(dst + out, src[p], bsize); out += bsize; p += 1; if (fs) fs->n_rle++;
        } else {
            if (bsize > block_max || bsize < 2) { rc = ERR(ZKO_E_CORRUPTION); break; }
            if (p + bsize > src_size) { rc = ERR(ZKO_E_SRC_SIZE_WRONG); break; }
            i64 r = decode_compressed_block(st, src + p, bsize, dst, out, dst_cap, (u32)(window > 0xFFFFFFFFu ? 0xFFFFFFFFu : window), fs, prefix, plen);
            if (r < 0) { rc = r; break; }
            if ((u64)r > block_max) { rc = ERR(ZKO_E_CORRUPTION); break; }
            out += (size_t)r; p += bsize; if (fs) fs->n_comp++;
        }
        if (last) break;
    }
    free(st->lit); free(st);
    if (rc < 0) return rc;
    if (fl && fcs != out) return ERR(ZKO_E_CORRUPTION);
    if (cks) {
        if (p + 4 > src_size) return ERR(ZKO_E_SRC_SIZE_WRONG);
        if (verify && rd32(src + p) != (u32)zko_xxh64(dst, out, 0)) return ERR(ZKO_E_CHECKSUM_WRONG);
        p += 4;
    }
    *consumed = p;
    return (i64)out;
}
