// The encoder's match + parse kernel (zk_enc_match.h, the source hipcc compiles) under the workgroup emulator, built with
// AddressSanitizer + UBSan: `__shared__` arrays are real arrays here, so an index that leaves the ring, the hash table, `best[]` or a
// tile's sequence area -- silent corruption of a neighbouring array in LDS on the device -- is a report; so is an access outside the
// source, prefix or output buffers, which are exact-size heap allocations.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined tests/sim/encode_san.cpp -o /tmp/encsan
//   /tmp/encsan <level> <bytes> <frame size> <kind: 0 words | 1 byte runs | 2 random> [prefix bytes]
#include "zk_enc_sim.cpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
static uint64_t s_ = 88172645463325252ull;
static uint64_t rnd() { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; }
static void fill(std::vector<uint8_t> &v, size_t n, int kind)
{
    static const char *words[] = {"the ", "of ", "and ", "compression ", "frame ", "seek ", "table ", "entropy ", "zeekstd ", "window ", "match ", "literal "};
    size_t p = 0;
    while (p < n) {
        if (kind == 0) { const char *w = words[rnd() % 12]; for (; *w && p < n; w++) v[p++] = (uint8_t)*w; if (rnd() % 9 == 0 && p < n) v[p++] = (uint8_t)('a' + rnd() % 26); }
        else if (kind == 1) { const uint8_t b = (uint8_t)rnd(); size_t r = 1 + rnd() % 300; while (r-- && p < n) v[p++] = b; }
        else v[p++] = (uint8_t)rnd();
    }
}
int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int level = atoi(argv[1]);
    const size_t n = strtoull(argv[2], nullptr, 10);
    const uint32_t fs = (uint32_t)strtoul(argv[3], nullptr, 10);
    const int kind = atoi(argv[4]);
    const size_t plen = argc > 5 ? strtoull(argv[5], nullptr, 10) : 0;
    std::vector<uint8_t> src(n + 8), prefix(plen);
    fill(src, n, kind);
    if (plen) {                                              // an "old version": the input's start with edits, so that the frame finds it
        fill(prefix, plen, kind);
        for (size_t i = 0; i < plen && i < n; i++) if (rnd() % 50) prefix[plen - 1 - i] = src[(n < plen ? n : plen) - 1 - i];
    }
    const size_t cap = n / 1024 + 64 * ((n + fs - 1) / fs) + 64;
    std::vector<uint32_t> nseq(cap), nlit(cap), bsz(cap);
    std::vector<uint64_t> seq_at(cap), lit_at(cap), seqs(n / 4 + 2 * cap + 64);
    std::vector<uint8_t> lits(n + 64);
    const int nb = zk_enc_sim_match(src.data(), n, fs, level, plen ? prefix.data() : nullptr, plen, (uint32_t)cap, nseq.data(), nlit.data(), bsz.data(),
                                    seq_at.data(), lit_at.data(), seqs.data(), seqs.size(), lits.data(), lits.size());
    uint64_t total = 0;
    for (int b = 0; b < nb; b++) total += bsz[b];
    printf("level %d, %zu bytes, frames of %u, kind %d, prefix %zu -> %d blocks, %llu bytes covered\n", level, n, fs, kind, plen, nb, (unsigned long long)total);
    return nb >= 0 && total == n ? 0 : 1;
}
