// The decode lane code of the kernels (zk_device.h, through tests/sim/zk_sim.cpp) under AddressSanitizer + UBSan on DAMAGED archives.
// On the device an out-of-range read of a corrupt frame is silent (or a fault of the whole process); here every buffer the harness
// hands over is a heap allocation of exactly the size the C ABI promises -- compressed bytes + ZK_COMP_PADDING, output bytes -- so a
// lane that follows a damaged header, table or offset out of its buffers is a sanitizer report.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined tests/sim/decode_fuzz.cpp -o /tmp/decfuzz
//   /tmp/decfuzz <case file> <iterations> <seed> <walk: 0 lane per block | 1 quad, 16-bit cells | 2 quad, 8-byte cells> [prefix file]
// case file (written by tests/test_sim_decode.py): u32 nframes, u64 comp_len, u64 out_len, (u64 c, u64 d) x nframes, comp, expected.
// Exit 0: every mutated input produced statuses and bytes without leaving its buffers (and the unmutated one the expected bytes).
#include "zk_sim.cpp"
#include <cstdio>
#include <cstdlib>

static uint64_t f_s;
static uint64_t f_rnd() { f_s ^= f_s << 13; f_s ^= f_s >> 7; f_s ^= f_s << 17; return f_s; }

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    const uint64_t iters = strtoull(argv[2], nullptr, 10);
    f_s = strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ull + 7;
    zk_sim_set_fse_quad(atoi(argv[4]));
    uint32_t nf; uint64_t clen, olen;
    if (fread(&nf, 4, 1, f) != 1 || fread(&clen, 8, 1, f) != 1 || fread(&olen, 8, 1, f) != 1) return 2;
    std::vector<uint64_t> c_off(nf + 1, 0), d_off(nf + 1, 0);
    for (uint32_t i = 0; i < nf; i++) {
        uint64_t cd[2];
        if (fread(cd, 8, 2, f) != 2) return 2;
        c_off[i + 1] = c_off[i] + cd[0]; d_off[i + 1] = d_off[i] + cd[1];
    }
    std::vector<uint8_t> comp(clen), want(olen);
    if ((clen && fread(comp.data(), 1, clen, f) != clen) || (olen && fread(want.data(), 1, olen, f) != olen)) return 2;
    fclose(f);
    std::vector<uint8_t> prefix;                            // a raw-content prefix the frames were made against (ZSTD_DCtx_refPrefix)
    if (argc > 5) {
        FILE *pf = fopen(argv[5], "rb");
        if (!pf) return 2;
        uint8_t tmp[65536];
        for (size_t k; (k = fread(tmp, 1, sizeof tmp, pf)) > 0;) prefix.insert(prefix.end(), tmp, tmp + k);
        fclose(pf);
    }
    uint64_t flagged = 0;
    for (uint64_t it = 0; it <= iters; it++) {
        // exact-size heap buffers, fresh every round (what the previous round left must not help)
        uint8_t *in = (uint8_t *)malloc(clen + 8), *out = (uint8_t *)malloc(olen + 1);
        int32_t *st = (int32_t *)malloc(sizeof(int32_t) * (nf ? nf : 1));
        memcpy(in, comp.data(), clen); memset(in + clen, 0, 8); memset(out, 0xEE, olen + 1);
        if (it && clen) {
            switch (f_rnd() % 4) {
            case 0: in[f_rnd() % clen] ^= (uint8_t)(1u << (f_rnd() % 8)); break;
            case 1: for (int k = 0; k < 3; k++) in[f_rnd() % clen] = (uint8_t)f_rnd(); break;
            case 2: { const uint64_t a = f_rnd() % clen, n = 1 + f_rnd() % 16; for (uint64_t k = a; k < clen && k < a + n; k++) in[k] = (uint8_t)f_rnd(); break; }
            case 3: { const uint64_t a = f_rnd() % clen; memset(in + a, (int)(f_rnd() & 0xFF), (size_t)((clen - a) < 64 ? clen - a : 64)); break; }
            }
        }
        uint8_t *pre = prefix.empty() ? nullptr : (uint8_t *)malloc(prefix.size());
        if (pre) memcpy(pre, prefix.data(), prefix.size());
        const int rc = pre ? zk_sim_decode_prefix(in, c_off.data(), d_off.data(), 0, nf, out, st, 16, 1024, pre, prefix.size())
                           : zk_sim_decode(in, c_off.data(), d_off.data(), 0, nf, out, st, 16, 1024);
        free(pre);
        bool any = rc != 0;
        for (uint32_t i = 0; i < nf; i++) any = any || st[i] != 0;
        if (!it && (any || (olen && memcmp(out, want.data(), olen) != 0))) { fprintf(stderr, "the undamaged archive does not decode\n"); return 3; }
        if (out[olen] != 0xEE) { fprintf(stderr, "a byte behind the output was written\n"); return 4; }
        flagged += any;
        free(in); free(out); free(st);
    }
    printf("%llu damaged inputs, %llu flagged\n", (unsigned long long)iters, (unsigned long long)flagged);
    return 0;
}
