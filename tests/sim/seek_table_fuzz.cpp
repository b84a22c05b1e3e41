// The seek-table parser and serializer of the host mirror (zeekstd_amd/csrc/host/seek_table.cpp, seekable.cpp: the code that reads
// UNTRUSTED bytes -- the tail or head of any file handed to the Decoder) under AddressSanitizer + UBSan, fed mutated tables:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -I zeekstd_amd/csrc/host tests/sim/seek_table_fuzz.cpp \
//       zeekstd_amd/csrc/host/seek_table.cpp zeekstd_amd/csrc/host/seekable.cpp -o /tmp/stfuzz && /tmp/stfuzz 200000 1
// What the reference guards with cargo-fuzz (fuzz/fuzz_targets) and proptests (seek_table.rs:1227-1266).
// Exit code 0: every input either parsed into a table whose every accessor stays in range and that serialises back to a table equal
// to itself, or was refused with a zeekstd::Error; anything else (a sanitizer report, another exception, a mismatch) is a failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "zeekstd.hpp"

extern "C" const char *zk_error_name(int) { return "engine error"; }     // (the one symbol of the engine the two files refer to)
using namespace zeekstd;

static uint64_t g_s;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }

static std::vector<uint8_t> serialise(const SeekTable &t, Format f)
{
    Serializer ser = t.into_format_serializer(f);
    std::vector<uint8_t> out;
    uint8_t buf[97];
    for (;;) {
        const size_t want = 1 + rnd() % sizeof buf;                           // ragged buffer sizes (seek_table.rs:1256-1260)
        const size_t n = ser.write_into(buf, want);
        if (!n) break;
        out.insert(out.end(), buf, buf + n);
    }
    if (out.size() != ser.encoded_len()) { fprintf(stderr, "encoded_len mismatch\n"); exit(3); }
    return out;
}

static void exercise(const SeekTable &t)
{
    const uint32_t n = t.num_frames();
    uint64_t cs = 0, ds = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (t.frame_start_comp(i) != cs || t.frame_start_decomp(i) != ds) { fprintf(stderr, "prefix sums broken\n"); exit(4); }
        cs += t.frame_size_comp(i); ds += t.frame_size_decomp(i);
        if (t.frame_end_comp(i) != cs || t.frame_end_decomp(i) != ds) { fprintf(stderr, "ends broken\n"); exit(4); }
    }
    if (t.size_comp() != cs || t.size_decomp() != ds) { fprintf(stderr, "totals broken\n"); exit(4); }
    (void)t.max_frame_size_comp(); (void)t.max_frame_size_decomp();
    for (int k = 0; k < 8; k++) {
        const uint64_t oc = cs ? rnd() % (cs + 3) : rnd() % 3, od = ds ? rnd() % (ds + 3) : rnd() % 3;
        const uint32_t ic = t.frame_index_comp(oc), id = t.frame_index_decomp(od);
        // (no frames: upstream computes 0u32 - 1 -- a debug panic, 0xFFFFFFFF in a release build -- and so does the mirror)
        if (n ? (ic >= n || id >= n) : (ic != 0xFFFFFFFFu || id != 0xFFFFFFFFu)) { fprintf(stderr, "frame index out of range\n"); exit(4); }
    }
    try { (void)t.frame_size_comp(n); fprintf(stderr, "index n accepted\n"); exit(4); } catch (const Error &e) { if (!e.is_frame_index_too_large()) exit(4); }
    // a table that parsed serialises to bytes that parse to the same table, in both formats
    for (Format f : {Format::Head, Format::Foot}) {
        const std::vector<uint8_t> b = serialise(t, f);
        BytesWrapper w(b.data(), b.size());
        if (!(SeekTable::from_seekable_format(w, f) == t)) { fprintf(stderr, "round trip differs\n"); exit(5); }
        if (f == Format::Head && !(SeekTable::from_bytes_head(b.data(), b.size()) == t)) { fprintf(stderr, "from_reader differs\n"); exit(5); }
    }
}

int main(int argc, char **argv)
{
    const uint64_t iters = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000;
    g_s = argc > 2 ? strtoull(argv[2], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1 : 0x1234567;
    uint64_t parsed = 0, refused = 0;
    for (uint64_t it = 0; it < iters; it++) {
        SeekTable t;
        const uint32_t n = rnd() % 5 == 0 ? 0 : rnd() % 48;
        for (uint32_t i = 0; i < n; i++) t.log_frame((uint32_t)(rnd() % 5 ? rnd() % 100000 : rnd()), (uint32_t)(rnd() % 5 ? rnd() % 300000 : rnd()));
        const Format f = rnd() & 1 ? Format::Head : Format::Foot;
        std::vector<uint8_t> b = serialise(t, f);
        // a payload in front of a Foot table / behind a Head table, as in a real file
        std::vector<uint8_t> pad(rnd() % 64);
        for (auto &x : pad) x = (uint8_t)rnd();
        if (f == Format::Foot) b.insert(b.begin(), pad.begin(), pad.end()); else b.insert(b.end(), pad.begin(), pad.end());
        switch (rnd() % 8) {
        case 0: break;                                                         // untouched: must parse back
        case 1: for (int k = 1 + rnd() % 4; k > 0 && !b.empty(); k--) b[rnd() % b.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 2: b.resize(b.empty() ? 0 : rnd() % b.size()); break;              // truncated
        case 3: { const size_t at = f == Format::Foot ? b.size() - 9 : 8; if (b.size() >= 17) { const uint32_t v = (uint32_t)rnd(); memcpy(&b[at], &v, 4); } break; }   // frame count
        case 4: { const size_t at = f == Format::Foot ? b.size() - 5 : 12; if (b.size() >= 17) b[at] = (uint8_t)rnd(); break; }                                         // descriptor
        case 5: { if (b.size() >= 8) { const uint32_t v = (uint32_t)rnd(); memcpy(&b[f == Format::Foot ? pad.size() + 4 : 4], &v, 4); } break; }                          // skippable frame size
        case 6: for (auto &x : b) x = (uint8_t)rnd(); break;                    // noise of the same length
        case 7: { std::vector<uint8_t> j(rnd() % 40); for (auto &x : j) x = (uint8_t)rnd(); b.insert(b.begin() + (b.empty() ? 0 : rnd() % b.size()), j.begin(), j.end()); break; }
        }
        for (Format g : {f, f == Format::Head ? Format::Foot : Format::Head}) {
            try {
                BytesWrapper w(b.data(), b.size());
                const SeekTable p = SeekTable::from_seekable_format(w, g);
                parsed++;
                exercise(p);
            } catch (const Error &) { refused++; }
        }
        try { const SeekTable p = SeekTable::from_bytes_head(b.data(), b.size()); parsed++; exercise(p); } catch (const Error &) { refused++; }
    }
    printf("%llu inputs: %llu parsed, %llu refused\n", (unsigned long long)iters, (unsigned long long)parsed, (unsigned long long)refused);
    return parsed && refused ? 0 : 6;
}
