/* The oracle's frame decoder (oracle/zstd_oracle.c) under ASan + UBSan on damaged frames: the checker itself must not
 * leave its buffers.   gcc -O1 -g -fsanitize=address,undefined tests/sim/oracle_fuzz.c -o /tmp/oracle_fuzz && ASAN_OPTIONS=detect_leaks=0 /tmp/oracle_fuzz <case file> <iters> <seed> */
#include "../../oracle/zstd_oracle.c"
#include <stdio.h>
#include <stdlib.h>
static unsigned long long s_;
static unsigned long long rnd(void) { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    unsigned nf; unsigned long long clen, olen;
    if (!f || fread(&nf, 4, 1, f) != 1 || fread(&clen, 8, 1, f) != 1 || fread(&olen, 8, 1, f) != 1) return 2;
    unsigned long long *cd = malloc(16 * nf);
    if (fread(cd, 16, nf, f) != nf) return 2;
    unsigned char *comp = malloc(clen + 1), *want = malloc(olen + 1);
    if ((clen && fread(comp, 1, clen, f) != clen) || (olen && fread(want, 1, olen, f) != olen)) return 2;
    const unsigned long long iters = strtoull(argv[2], 0, 10);
    s_ = strtoull(argv[3], 0, 10) * 0x9E3779B97F4A7C15ull + 3;
    unsigned long long bad = 0;
    for (unsigned long long it = 0; it <= iters; it++) {
        unsigned long long cpos = 0, dpos = 0;
        for (unsigned i = 0; i < nf; i++) {
            const size_t cs = cd[2 * i], ds = cd[2 * i + 1];
            unsigned char *in = malloc(cs ? cs : 1), *out = malloc(ds ? ds : 1);      /* exact sizes: no padding at all for the oracle */
            memcpy(in, comp + cpos, cs);
            if (it && cs) { const int k = 1 + rnd() % 3; for (int j = 0; j < k; j++) in[rnd() % cs] ^= (unsigned char)(1u << (rnd() % 8)); if (rnd() % 7 == 0) in[rnd() % cs] = (unsigned char)rnd(); }
            size_t used = 0; zko_frame_stats st;
            const i64 r = zko_frame_decode(in, it && rnd() % 5 == 0 && cs ? (size_t)(rnd() % cs) : cs, out, ds, &used, 1, &st);
            if (!it && (r != (i64)ds || memcmp(out, want + dpos, ds))) { fprintf(stderr, "undamaged frame %u does not decode\n", i); return 3; }
            bad += r < 0;
            free(in); free(out);
            cpos += cs; dpos += ds;
        }
    }
    printf("%llu rounds over %u frames, %llu refusals\n", iters, nf, bad);
    return 0;
}
