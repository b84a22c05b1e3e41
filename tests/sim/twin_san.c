/* The encoder's CPU twin (oracle/zstd_oracle_enc.c) and the oracle's decoder under AddressSanitizer + UBSan: 120 round trips -- levels 1 / 3 / 6,
 * words / byte runs / random bytes / long copies, 0 ... 3.5 MB (far history inside a frame), every third with a prefix -- in exact-size heap
 * buffers.  The twin mirrors the match kernel position for position: an index the twin gets wrong is one the kernel gets wrong.
 *   gcc -O1 -g -w -fsanitize=address,undefined tests/sim/twin_san.c oracle/zstd_oracle.c oracle/zstd_oracle_enc.c -ldl -o /tmp/twin_san */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef long long i64; typedef unsigned char u8;
i64 zko_frame_encode_prefix(const u8 *src, size_t n, u8 *dst, size_t cap, int level, int checksum, const u8 *prefix, size_t plen);
i64 zko_frame_decode_prefix(const u8 *src, size_t src_size, u8 *dst, size_t dst_cap, size_t *consumed, int verify, void *fs, const u8 *prefix, size_t plen);
static unsigned long long s_ = 88172645463325252ull;
static unsigned long long rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }
static void fill(u8 *v, size_t n, int kind)
{
    static const char *words[] = {"the ", "of ", "and ", "compression ", "frame ", "seek ", "table ", "entropy ", "zeekstd ", "window ", "match ", "literal "};
    size_t p = 0;
    while (p < n) {
        if (kind == 0) { const char *w = words[rnd() % 12]; for (; *w && p < n; w++) v[p++] = (u8)*w; if (rnd() % 9 == 0 && p < n) v[p++] = (u8)('a' + rnd() % 26); }
        else if (kind == 1) { const u8 b = (u8)rnd(); size_t r = 1 + rnd() % 300; while (r-- && p < n) v[p++] = b; }
        else if (kind == 3) { size_t r = 1 + rnd() % 5000; size_t back = p ? 1 + rnd() % p : 0; while (r-- && p < n) { v[p] = back ? v[p - back] : (u8)rnd(); p++; } if (p < n) v[p++] = (u8)rnd(); }
        else v[p++] = (u8)rnd();
    }
}
int main(void)
{
    char stats[256];
    int runs = 0;
    for (int level = 1; level <= 6; level += (level == 1 ? 2 : 3))
        for (int kind = 0; kind < 4; kind++)
            for (int k = 0; k < 10; k++) {
                static const size_t sizes[] = {0, 1, 7, 100, 4095, 4097, 65536, 200001, 700000, 3500000};
                const size_t n = sizes[k], plen = (k % 3 == 2) ? (size_t)(rnd() % 300000) : 0;
                u8 *src = malloc(n ? n : 1), *pre = malloc(plen ? plen : 1), *dst = malloc(n + (n >> 7) + 1024), *back = malloc(n ? n : 1);
                fill(src, n, kind); fill(pre, plen, kind);
                for (size_t i = 0; i < plen && i < n; i++) if (rnd() % 40) pre[plen - 1 - i] = src[(n < plen ? n : plen) - 1 - i];
                const i64 c = zko_frame_encode_prefix(src, n, dst, n + (n >> 7) + 1024, level, k & 1, plen ? pre : NULL, plen);
                if (c < 0) { printf("encode failed %lld\n", c); return 1; }
                u8 *exact = malloc((size_t)c ? (size_t)c : 1); memcpy(exact, dst, (size_t)c);
                size_t used = 0;
                const i64 d = zko_frame_decode_prefix(exact, (size_t)c, back, n, &used, 1, stats, plen ? pre : NULL, plen);
                if (d != (i64)n || used != (size_t)c || memcmp(back, src, n)) { printf("round trip differs: level %d kind %d n %zu plen %zu (%lld)\n", level, kind, n, plen, d); return 1; }
                free(src); free(pre); free(dst); free(back); free(exact); runs++;
            }
    printf("%d round trips clean\n", runs);
    return 0;
}
