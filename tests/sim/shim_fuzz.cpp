// The Level-C shim's stream handling (zeekstd_amd/csrc/levelc/zstd_shim.cpp: ZSTD_decompressStream / ZSTD_compressStream2 over whole frames) under
// AddressSanitizer + UBSan on the CPU, with the engine STUBBED: the shim reads UNTRUSTED bytes -- it follows frame and block headers through
// the caller's buffers to find where a frame ends -- and that bookkeeping is what this harness feeds damaged streams, ragged input chunks and
// tiny output buffers.  (What the engine makes of the bytes is the GPU suite's business: tests/test_gpu_levelc.py.)
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined tests/sim/shim_fuzz.cpp -o /tmp/shimfuzz -pthread
//   /tmp/shimfuzz <golden blob> <cases> <seed>
// Checked on every call: positions stay inside their buffers and never go back; the shim never takes a byte beyond the end of the frame it is
// on (the stub engine reports which bytes it was shown: exactly one frame, from its magic number to its last byte); a call either makes
// progress, reports an error, or asks for input / room it does not have; every exact-size heap buffer is only touched inside its bounds.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// ---- the engine, stubbed (Level A as the shim uses it)
struct zk_engine { int dummy; };
static zk_engine g_stub;
static uint64_t g_seen_frames, g_seen_bytes;
extern "C" {
int zk_engine_create(int, zk_engine **out) { *out = &g_stub; return 0; }
const char *zk_error_name(int) { return "stub error"; }
uint64_t zk_compress_bound(uint64_t n, uint32_t) { return n + (n >> 6) + 64; }
static uint64_t touch(const uint8_t *p, uint64_t n) { uint64_t s = 0; for (uint64_t i = 0; i < n; i++) s += p[i]; return s; }
// "sizes": a deterministic function of the frame's bytes, <= 200 000 (an exact-size caller buffer must hold it)
int zk_frame_content_sizes(zk_engine *, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off, uint32_t first, uint32_t count, uint64_t *sizes, int32_t *st)
{
    if (first != 0 || count != 1 || c_off[0] != 0 || c_off[1] != comp_size) abort();
    if (comp_size < 9 || comp[0] != 0x28 || comp[1] != 0xB5 || comp[2] != 0x2F || comp[3] != 0xFD) abort();      // the shim hands over one frame, from its magic number on
    sizes[0] = touch(comp, comp_size + 8) % 200001;                                                                 // (+ 8: the readable padding Level A asks for)
    st[0] = (comp[comp_size - 1] & 7) == 7 ? -20 : 0;                                                               // now and then: "damaged"
    return 0;
}
int zk_decode_frames_prefix(zk_engine *, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off, const uint64_t *d_off, uint32_t first, uint32_t count,
                            const uint8_t *prefix, uint64_t plen, uint8_t *dst, uint64_t dst_cap, int, int32_t *st)
{
    if (first != 0 || count != 1 || c_off[1] != comp_size || d_off[1] > dst_cap) abort();
    if (comp[0] != 0x28 || comp[1] != 0xB5 || comp[2] != 0x2F || comp[3] != 0xFD) abort();
    g_seen_frames++; g_seen_bytes += comp_size;
    const uint64_t s = touch(comp, comp_size + 8) + (plen ? touch(prefix, plen) : 0);
    memset(dst, (int)(s & 0xFF), (size_t)d_off[1]);
    st[0] = 0;
    return 0;
}
int zk_encode_frames_prefix(zk_engine *, const uint8_t *src, uint64_t n, uint32_t, int, int checksum, const uint8_t *prefix, uint64_t plen, uint8_t *dst, uint64_t cap,
                            uint32_t *cs, uint32_t *ds, uint32_t, uint32_t *nf, uint64_t *written)
{
    // a stored frame: magic, FHD (single segment, 4-byte FCS), one raw block per 100 000 bytes, optional "checksum"
    (void)touch(src, n); if (plen) (void)touch(prefix, plen);
    uint64_t p = 0;
    const uint8_t hdr[9] = {0x28, 0xB5, 0x2F, 0xFD, (uint8_t)(0xA0 | (checksum ? 4 : 0)), (uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    if (cap < n + 64 + 3 * (n / 100000 + 1)) return -70;
    memcpy(dst, hdr, 9); p = 9;
    uint64_t at = 0;
    do {
        const uint64_t k = n - at < 100000 ? n - at : 100000;
        const uint32_t h = (uint32_t)(at + k == n) | (uint32_t)(k << 3);
        dst[p] = (uint8_t)h; dst[p + 1] = (uint8_t)(h >> 8); dst[p + 2] = (uint8_t)(h >> 16); p += 3;
        if (k) memcpy(dst + p, src + at, k);
        p += k; at += k;
    } while (at < n);
    if (checksum) { memset(dst + p, 0x5A, 4); p += 4; }
    *cs = (uint32_t)p; *ds = (uint32_t)n; *nf = 1; *written = p;
    return 0;
}
}
#include "../../zeekstd_amd/csrc/levelc/zstd_shim.cpp"

static uint64_t g_s;
static uint64_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }

// one stream through ZSTD_decompressStream with ragged chunks and small outputs; returns frames that ended (return value 0)
static int run_decode(const std::vector<uint8_t> &stream, bool expect_clean)
{
    ZSTD_DCtx *d = ZSTD_createDCtx();
    size_t pos = 0;
    int ends = 0, guard = 0;
    bool failed = false;
    while (pos < stream.size() && !failed) {
        const size_t take = 1 + rnd() % (rnd() % 4 ? 97 : 40000);
        const size_t n = take < stream.size() - pos ? take : stream.size() - pos;
        std::vector<uint8_t> chunk(stream.begin() + (long)pos, stream.begin() + (long)(pos + n));     // an exact-size heap copy: a read past it is a report
        ZSTD_inBuffer in{chunk.data(), n, 0};
        int idle = 0;
        while (in.pos < n) {
            const size_t room = rnd() % 5 ? 1 + rnd() % 300 : 131072;
            std::vector<uint8_t> ob(room);
            ZSTD_outBuffer out{ob.data(), room, 0};
            const size_t before = in.pos;
            const size_t r = ZSTD_decompressStream(d, &out, &in);
            if (in.pos > n || in.pos < before || out.pos > room) { fprintf(stderr, "positions out of range\n"); exit(3); }
            if (ZSTD_isError(r)) {
                // a context that has failed stays failed, with the SAME verdict, whatever it is shown next -- until it is reset (the
                // second call used to compute a negative count of bytes to take)
                for (int again = 0; again < 3; again++) {
                    std::vector<uint8_t> more(1 + rnd() % 64, (uint8_t)rnd());
                    ZSTD_inBuffer in2{more.data(), more.size(), 0};
                    uint8_t ob2[16];
                    ZSTD_outBuffer out2{ob2, sizeof ob2, 0};
                    const size_t r2 = ZSTD_decompressStream(d, &out2, &in2);
                    if (r2 != r || in2.pos != 0 || out2.pos != 0) { fprintf(stderr, "a failed context answered %zx after %zx\n", r2, r); exit(14); }
                }
                failed = true; break;
            }
            ends += r == 0;
            if (out.pos == 0 && in.pos == before) { if (++idle > 3) { fprintf(stderr, "no progress\n"); exit(4); } } else idle = 0;
            if (++guard > 4000000) { fprintf(stderr, "runaway\n"); exit(5); }
        }
        pos += in.pos ? in.pos : n;
        if (in.pos < n && !failed) { fprintf(stderr, "input left without an error\n"); exit(6); }
    }
    ZSTD_freeDCtx(d);
    if (expect_clean && failed) { fprintf(stderr, "a clean stream failed\n"); exit(7); }
    return ends;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> blob(1 << 22);
    blob.resize(fread(blob.data(), 1, blob.size(), f));
    fclose(f);
    const int cases = atoi(argv[2]);
    g_s = strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
    // the blob is the goldens' archives back to back (tests/golden/archives.bin): every one a sequence of whole zstd frames
    // 1. compress side: random inputs through compressStream2 in ragged pieces, tiny output buffers; the stored frames then go through the decoder side
    std::vector<uint8_t> made;
    int made_frames = 0;
    for (int c = 0; c < 40; c++) {
        ZSTD_CCtx *cc = ZSTD_createCCtx();
        if (ZSTD_isError(ZSTD_CCtx_setParameter(cc, 201, (int)(rnd() & 1)))) return 8;
        const size_t n = rnd() % 5 ? rnd() % 70000 : 0;
        std::vector<uint8_t> src(n);
        for (auto &b : src) b = (uint8_t)rnd();
        size_t at = 0;
        while (at < n) {
            const size_t want = 1 + rnd() % 9000, k = want < n - at ? want : n - at;
            ZSTD_inBuffer in{src.data() + at, k, 0};
            uint8_t ob[64];
            ZSTD_outBuffer out{ob, sizeof ob, 0};
            if (ZSTD_isError(ZSTD_compressStream2(cc, &out, &in, 0)) || in.pos != k || out.pos != 0) { fprintf(stderr, "e_continue\n"); return 9; }
            at += k;
        }
        for (;;) {
            const size_t room = 1 + rnd() % 700;
            std::vector<uint8_t> ob(room);
            ZSTD_inBuffer in{nullptr, 0, 0};
            ZSTD_outBuffer out{ob.data(), room, 0};
            const size_t r = ZSTD_compressStream2(cc, &out, &in, 2);
            if (ZSTD_isError(r)) { fprintf(stderr, "e_end\n"); return 10; }
            made.insert(made.end(), ob.begin(), ob.begin() + (long)out.pos);
            if (r == 0) break;
        }
        made_frames++;
        ZSTD_freeCCtx(cc);
    }
    g_seen_frames = 0;
    if (run_decode(made, true) != made_frames || g_seen_frames != (uint64_t)made_frames) { fprintf(stderr, "the stored frames did not come back one by one\n"); return 11; }
    // 2. the goldens, whole: every frame is handed to the engine exactly once, never with a byte of its neighbour
    g_seen_frames = 0; g_seen_bytes = 0;
    (void)run_decode(blob, false);                        // (the stub calls one frame in eight damaged: the run may end early)
    // 3. damaged streams: flips, cuts, garbage
    for (int c = 0; c < cases; c++) {
        const size_t a = rnd() % blob.size(), len = 1 + rnd() % (rnd() % 3 ? 3000 : 200000);
        std::vector<uint8_t> s(blob.begin() + (long)a, blob.begin() + (long)(a + len < blob.size() ? a + len : blob.size()));
        // start on a frame where one can be found (else the stream is garbage from its first byte: also a case)
        for (size_t i = 0; i + 4 <= s.size() && i < 5000; i++)
            if (s[i] == 0x28 && s[i + 1] == 0xB5 && s[i + 2] == 0x2F && s[i + 3] == 0xFD) { s.erase(s.begin(), s.begin() + (long)i); break; }
        const int kind = (int)(rnd() % 4);
        if (kind == 1) for (int k = 0; k < 1 + (int)(rnd() % 6); k++) s[rnd() % s.size()] ^= (uint8_t)(1u << (rnd() % 8));
        if (kind == 2 && s.size() > 8) s.resize(1 + rnd() % s.size());
        if (kind == 3) for (size_t i = rnd() % s.size(); i < s.size() && (rnd() % 50); i++) s[i] = (uint8_t)rnd();
        (void)run_decode(s, false);
    }
    // 4. verdicts the header walk owes at once -- before the bytes a damaged header asks for are waited for (libzstd refuses there too)
    {
        auto first_call = [](std::vector<uint8_t> v) {
            ZSTD_DCtx *d = ZSTD_createDCtx();
            uint8_t ob[64];
            ZSTD_inBuffer in{v.data(), v.size(), 0};
            ZSTD_outBuffer out{ob, sizeof ob, 0};
            const size_t r = ZSTD_decompressStream(d, &out, &in);
            ZSTD_freeDCtx(d);
            return r;
        };
        const size_t big = first_call({0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58, 0x08, 0x00, 0x10});          // window 1 MiB; a raw block of 2^17 + 1 bytes
        const size_t okb = first_call({0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58, 0x00, 0x00, 0x10});          // ... of 2^17 bytes: waits for them
        const size_t win = first_call({0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x05, 0x31, 0x00, 0x00});          // single segment, 5 bytes of content; a raw block of 6
        const size_t rsv = first_call({0x28, 0xB5, 0x2F, 0xFD, 0x08, 0x58, 0x01, 0x00, 0x00});          // the reserved bit of the descriptor
        const size_t typ = first_call({0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58, 0x07, 0x00, 0x00});          // block type 3
        {   // ZSTD_d_windowLogMax: a window of 2^27 (+ 1 for a Single_Segment size) is the default limit; the parameter moves it; a parameter reset restores it
            const std::vector<uint8_t> w27 = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x88, 0x09, 0x00, 0x00}, w27m = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x89, 0x09, 0x00, 0x00};   // 2^27; 2^27 * 9 / 8
            auto call = [](ZSTD_DCtx *d, std::vector<uint8_t> v) { uint8_t ob[8]; ZSTD_inBuffer in{v.data(), v.size(), 0}; ZSTD_outBuffer out{ob, sizeof ob, 0}; const size_t r = ZSTD_decompressStream(d, &out, &in); ZSTD_DCtx_reset(d, 1); return r; };
            ZSTD_DCtx *d = ZSTD_createDCtx();
            bool ok = !ZSTD_isError(call(d, w27)) && ZSTD_getErrorCode(call(d, w27m)) == 16;
            ok = ok && !ZSTD_isError(ZSTD_DCtx_setParameter(d, 100, 28)) && !ZSTD_isError(call(d, w27m));
            ok = ok && !ZSTD_isError(ZSTD_DCtx_setParameter(d, 100, 20)) && ZSTD_getErrorCode(call(d, w27)) == 16;
            ok = ok && !ZSTD_isError(ZSTD_DCtx_reset(d, 2)) && !ZSTD_isError(call(d, w27)) && ZSTD_getErrorCode(ZSTD_DCtx_setParameter(d, 100, 32)) == 42;
            ZSTD_freeDCtx(d);
            if (!ok) { fprintf(stderr, "window limit\n"); return 13; }
        }
        if (!ZSTD_isError(big) || ZSTD_getErrorCode(big) != 20 || ZSTD_isError(okb) || okb == 0 || !ZSTD_isError(win) || ZSTD_getErrorCode(win) != 20 ||
            !ZSTD_isError(rsv) || ZSTD_getErrorCode(rsv) != 14 || !ZSTD_isError(typ) || ZSTD_getErrorCode(typ) != 20) { fprintf(stderr, "header verdicts\n"); return 12; }
    }
    // 5. a context that failed works again after a session reset; a skippable frame (the seek table of an archive) is walked over, not kept
    {
        ZSTD_DCtx *d = ZSTD_createDCtx();
        uint8_t ob[64];
        std::vector<uint8_t> bad = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58, 0x07, 0x00, 0x00};
        ZSTD_inBuffer in{bad.data(), bad.size(), 0};
        ZSTD_outBuffer out{ob, sizeof ob, 0};
        if (!ZSTD_isError(ZSTD_decompressStream(d, &out, &in))) { fprintf(stderr, "expected an error\n"); return 15; }
        ZSTD_DCtx_reset(d, 1);
        const size_t payload = 6 * 1000 * 1000 + 123;
        std::vector<uint8_t> skip(8 + payload, 0xAB);
        const uint8_t h[8] = {0x5E, 0x2A, 0x4D, 0x18, (uint8_t)payload, (uint8_t)(payload >> 8), (uint8_t)(payload >> 16), (uint8_t)(payload >> 24)};
        memcpy(skip.data(), h, 8);
        size_t pos = 0, ends = 0, most = 0;
        while (pos < skip.size()) {
            const size_t want = 1 + rnd() % 70000, n = want < skip.size() - pos ? want : skip.size() - pos;
            std::vector<uint8_t> chunk(skip.begin() + (long)pos, skip.begin() + (long)(pos + n));
            ZSTD_inBuffer i2{chunk.data(), n, 0};
            ZSTD_outBuffer o2{ob, sizeof ob, 0};
            const size_t r = ZSTD_decompressStream(d, &o2, &i2);
            if (ZSTD_isError(r) || o2.pos != 0 || i2.pos == 0) { fprintf(stderr, "skippable frame: %zx\n", r); return 16; }
            ends += r == 0;
            most = d->acc.capacity() > most ? d->acc.capacity() : most;
            pos += i2.pos;
        }
        ZSTD_freeDCtx(d);
        if (ends != 1 || most > 4096) { fprintf(stderr, "skippable frame: %zu ends, %zu bytes kept\n", ends, most); return 17; }
    }
    printf("shim fuzz: %d damaged streams, %llu frames shown to the engine, no report\n", cases, (unsigned long long)g_seen_frames);
    return 0;
}
