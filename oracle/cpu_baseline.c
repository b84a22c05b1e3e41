/*
 * cpu_baseline.c -- TEST/BENCH INFRASTRUCTURE (bench.py cpu_baseline leg only).
 *
 * The reference's CPU path timed on the host cores of the GPU box: a C restatement ("port") of the
 * zeekstd loops over the libzstd that is installed on the box (dlopen; the Rust reference cannot be
 * built here: no cargo/rustc, and libzstd 1.5.7 sources are not in the reference tree).
 *   decode: Decoder::decompress_with_prefix, lib/src/decode.rs:201-270, driven like the reference
 *           bench lib/benches/decompress.rs:18-25,35-39 (128 KiB output buffer reused, reset per pass)
 *           with a BytesWrapper source (lib/src/seekable.rs:74-80: memcpy into the 131 075-B in_buf)
 *   encode: Encoder::compress + end_frame, lib/src/encode.rs:311-354,438-472,641-665 driven like
 *           lib/benches/compress.rs:42-62 (131 591-B staging buffer flushed into a Vec)
 * Single thread, like the reference (one Decoder/Encoder = one core).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { const void *src; size_t size; size_t pos; } in_buf_t;
typedef struct { void *dst; size_t size; size_t pos; } out_buf_t;

static struct {
    void *h;
    void *(*createDCtx)(void); size_t (*freeDCtx)(void *);
    void *(*createCCtx)(void); size_t (*freeCCtx)(void *);
    size_t (*decompressStream)(void *, out_buf_t *, in_buf_t *);
    size_t (*compressStream2)(void *, out_buf_t *, in_buf_t *, int);
    size_t (*CCtx_setParameter)(void *, int, int);
    size_t (*CCtx_reset)(void *, int);
    size_t (*DCtx_reset)(void *, int);
    unsigned (*isError)(size_t);
    const char *(*versionString)(void);
} Z;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int zkb_open(const char *path)
{
    if (Z.h) return 0;
    Z.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!Z.h) return -1;
#define SYM(f, n) do { *(void **)(&Z.f) = dlsym(Z.h, n); if (!Z.f) return -2; } while (0)
    SYM(createDCtx, "ZSTD_createDCtx"); SYM(freeDCtx, "ZSTD_freeDCtx");
    SYM(createCCtx, "ZSTD_createCCtx"); SYM(freeCCtx, "ZSTD_freeCCtx");
    SYM(decompressStream, "ZSTD_decompressStream"); SYM(compressStream2, "ZSTD_compressStream2");
    SYM(CCtx_setParameter, "ZSTD_CCtx_setParameter"); SYM(CCtx_reset, "ZSTD_CCtx_reset"); SYM(DCtx_reset, "ZSTD_DCtx_reset");
    SYM(isError, "ZSTD_isError"); SYM(versionString, "ZSTD_versionString");
    return 0;
}
const char *zkb_version(void) { return Z.h ? Z.versionString() : ""; }

/* One full decode pass of a seekable payload (offset 0 .. size_decomp). Returns decoded bytes or -1. */
static int64_t decode_pass(void *dctx, const uint8_t *comp, size_t csize, uint64_t limit, uint8_t *in_buf, uint8_t *out)
{
    const size_t IN = 131075, OUT = 131072;
    size_t src_pos = 0, in_pos = 0, in_lim = 0;
    uint64_t offset = 0;
    while (offset < limit) {                                   /* decode.rs:221 */
        if (in_pos == in_lim) {                                /* decode.rs:222-225 + seekable.rs:74-80 */
            size_t n = csize - src_pos < IN ? csize - src_pos : IN;
            memcpy(in_buf, comp + src_pos, n); src_pos += n; in_lim = n; in_pos = 0;
            if (n == 0) return -1;
        }
        in_buf_t ib = { in_buf + in_pos, in_lim - in_pos, 0 };
        uint64_t rem = limit - offset;
        out_buf_t ob = { out, rem < OUT ? (size_t)rem : OUT, 0 };  /* caller's 128 KiB buffer, decode.rs:232-239 */
        size_t in_len = in_lim - in_pos;
        while (ib.pos < in_len && ob.pos < ob.size) {          /* decode.rs:242-256 */
            size_t r = Z.decompressStream(dctx, &ob, &ib);
            if (Z.isError(r)) return -1;
        }
        in_pos += ib.pos; offset += ob.pos;
    }
    return (int64_t)offset;
}

/* Times `reps` passes; returns best seconds per pass (or <0). sink receives a checksum of the last bytes. */
double zkb_time_decode(const uint8_t *comp, size_t csize, uint64_t dsize, int reps, uint64_t *sink)
{
    void *dctx = Z.createDCtx();
    uint8_t *in_buf = malloc(131075), *out = malloc(131072);
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        double t0 = now();
        int64_t n = decode_pass(dctx, comp, csize, dsize, in_buf, out);
        double t = now() - t0;
        Z.DCtx_reset(dctx, 1);                                 /* Decoder::reset, decode.rs:346-357 */
        if (n != (int64_t)dsize) { best = -1; break; }
        if (t < best) best = t;
        if (sink) *sink += out[0];
    }
    free(in_buf); free(out); Z.freeDCtx(dctx);
    return best;
}

/* Encoder loop, Uncompressed(frame_size) policy; output appended to dst (cap must be enough). Returns c bytes. */
static int64_t encode_pass(void *cctx, const uint8_t *src, size_t n, uint32_t frame_size, uint8_t *stage, uint8_t *dst, size_t cap)
{
    const size_t OUT = 131591;
    size_t pos = 0, w = 0;
    do {
        size_t d = n - pos < frame_size ? n - pos : frame_size;
        in_buf_t ib = { src + pos, d, 0 };
        while (ib.pos < d) {                                   /* encode.rs:340-346 */
            out_buf_t ob = { stage, OUT, 0 };
            size_t r = Z.compressStream2(cctx, &ob, &ib, 0);
            if (Z.isError(r) || w + ob.pos > cap) return -1;
            memcpy(dst + w, stage, ob.pos); w += ob.pos;       /* flush_out_buf -> Vec, encode.rs:779-787 */
        }
        for (;;) {                                             /* encode.rs:442-464 */
            in_buf_t e = { NULL, 0, 0 };
            out_buf_t ob = { stage, OUT, 0 };
            size_t r = Z.compressStream2(cctx, &ob, &e, 2);
            if (Z.isError(r) || w + ob.pos > cap) return -1;
            memcpy(dst + w, stage, ob.pos); w += ob.pos;
            if (r == 0) break;
        }
        Z.CCtx_reset(cctx, 1);                                 /* encode.rs:504-506 */
        pos += d;
    } while (pos < n);
    return (int64_t)w;
}

double zkb_time_encode(const uint8_t *src, size_t n, uint32_t frame_size, int level, int checksum, int reps,
                       uint8_t *dst, size_t cap, int64_t *csize_out)
{
    void *cctx = Z.createCCtx();
    Z.CCtx_setParameter(cctx, 100, level);
    Z.CCtx_setParameter(cctx, 201, checksum);
    uint8_t *stage = malloc(131591);
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        double t0 = now();
        int64_t c = encode_pass(cctx, src, n, frame_size, stage, dst, cap);
        double t = now() - t0;
        if (c < 0) { best = -1; break; }
        if (csize_out) *csize_out = c;
        if (t < best) best = t;
    }
    free(stage); Z.freeCCtx(cctx);
    return best;
}

/* Random-seek latency of the reference CPU path (BASELINE.json configs[3]): per seek, what
 * Decoder::set_offset + set_offset_limit + decompress do (decode.rs:402-437, 201-270): locate the frame by binary
 * search (seek_table.rs:916-934), reset the context, decode from the frame start, discard up to `offset`
 * ("dummy decompression", decode.rs:228-231), stop at the limit.  times_us[i] receives the latency of seek i. */
int zkb_time_seeks(const uint8_t *comp, const uint64_t *c_off, const uint64_t *d_off, uint32_t nframes,
                   const uint64_t *offsets, const uint32_t *lens, uint32_t nseeks, double *times_us, uint8_t *out)
{
    void *dctx = Z.createDCtx();
    uint8_t *dummy = malloc(131072);
    for (uint32_t i = 0; i < nseeks; i++) {
        double t0 = now();
        uint64_t off = offsets[i], limit = off + lens[i];
        uint32_t lo = 0, hi = nframes;                           /* frame_index_decomp */
        while (lo + 1 < hi) { uint32_t mid = lo + (hi - lo) / 2; if (d_off[mid] <= off) lo = mid; else hi = mid; }
        Z.DCtx_reset(dctx, 1);
        uint64_t pos = d_off[lo];
        in_buf_t ib = { comp + c_off[lo], (size_t)(c_off[nframes] - c_off[lo]), 0 };
        size_t got = 0;
        while (pos < limit) {
            out_buf_t ob;
            if (pos < off) { uint64_t n = off - pos; ob.dst = dummy; ob.size = n < 131072 ? (size_t)n : 131072; }
            else { ob.dst = out + got; ob.size = (size_t)(limit - pos); }
            ob.pos = 0;
            size_t r = Z.decompressStream(dctx, &ob, &ib);
            if (Z.isError(r)) { free(dummy); Z.freeDCtx(dctx); return -1; }
            if (pos >= off) got += ob.pos;
            pos += ob.pos;
            if (ob.pos == 0 && ib.pos == ib.size) break;
        }
        times_us[i] = (now() - t0) * 1e6;
    }
    free(dummy); Z.freeDCtx(dctx);
    return 0;
}
