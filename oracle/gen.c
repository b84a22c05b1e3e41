/*
 * gen.c -- TEST INFRASTRUCTURE.  Portable integer-only synthetic text generator of
 * SURVEY.md section 8(d) (stand-in for the reference's assets/dickens.txt, which is
 * absent from the checkout: /root/reference/.MISSING_LARGE_BLOBS:1, used by
 * lib/benches/compress.rs:6).  Level-1 zstd ratio ~2.5 at 2 MiB frames.
 * KATs: vocab[0..5] = isa, lv, lcvk, uenoot, uemu; chunk 0 XXH64 = 0xad0311eaad1ed582.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define V 4096
typedef struct { uint8_t len[V]; char w[V][10]; int ready; } vocab_t;
static vocab_t g_vocab;

static uint64_t seed_state(uint64_t seed)
{
    uint64_t x = seed + 0x9E3779B97F4A7C15ULL, z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return z ? z : 0x9E3779B97F4A7C15ULL;
}
static inline uint64_t next(uint64_t *s)
{
    uint64_t x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static void build_vocab(void)
{
    uint64_t s = seed_state(0x5EED0001ULL);
    for (int i = 0; i < V; i++) {
        int n = 2 + (int)((next(&s) >> 33) % 8);
        g_vocab.len[i] = (uint8_t)n;
        for (int k = 0; k < n; k++) g_vocab.w[i][k] = (char)('a' + (next(&s) >> 33) % 26);
    }
    g_vocab.ready = 1;
}
const char *zko_gen_vocab(int i, int *len) { if (!g_vocab.ready) build_vocab(); *len = g_vocab.len[i]; return g_vocab.w[i]; }

/* gen(n, seed): fill dst[0..n) */
void zko_gen_text(uint8_t *dst, size_t n, uint64_t seed)
{
    if (!g_vocab.ready) build_vocab();
    uint64_t s = seed_state(seed);
    size_t pos = 0;
    char tmp[16];
    while (pos < n) {
        uint64_t a = (next(&s) >> 40) % V, b = (next(&s) >> 40) % V, c = (next(&s) >> 40) % V;
        uint64_t idx = ((a * b / V) * c) / V;
        int l = g_vocab.len[idx];
        memcpy(tmp, g_vocab.w[idx], (size_t)l);
        uint64_t t = (next(&s) >> 40) % 64;
        if (t == 0) { tmp[l++] = '.'; tmp[l++] = '\n'; }
        else if (t < 5) { tmp[l++] = ','; tmp[l++] = ' '; }
        else if (t < 9) { tmp[l++] = '.'; tmp[l++] = ' '; }
        else tmp[l++] = ' ';
        size_t take = (size_t)l < n - pos ? (size_t)l : n - pos;
        memcpy(dst + pos, tmp, take);
        pos += take;
    }
}

/* big inputs: 2 MiB chunk k = gen(2 MiB, 0x5EED0002 + k); fills chunks [k0, k0+count) (last may be short) */
void zko_gen_chunks(uint8_t *dst, size_t total, uint64_t k0)
{
    const size_t C = 2u << 20;
    for (size_t off = 0, k = 0; off < total; off += C, k++) {
        size_t n = total - off < C ? total - off : C;
        zko_gen_text(dst + off, n, 0x5EED0002ULL + k0 + k);
    }
}

/* incompressible bytes: xorshift64* stream (same PRNG as the text generator) */
void zko_gen_random(uint8_t *dst, size_t n, uint64_t seed)
{
    uint64_t s = seed_state(seed);
    size_t i = 0;
    while (i + 8 <= n) { uint64_t v = next(&s); memcpy(dst + i, &v, 8); i += 8; }
    if (i < n) { uint64_t v = next(&s); memcpy(dst + i, &v, n - i); }
}
