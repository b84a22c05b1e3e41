// zstd_shim.cpp -- LEVEL C of the boundary: the libzstd symbols the UNMODIFIED zeekstd crate binds through zstd-safe, over
// this engine's batch ABI (include/zeekstd_amd.h, Level A).  Built as zeekstd_amd/libzstd_zeekstd_amd.so; link it in libzstd's place
// (INTEGRATION.md "Level C") and zeekstd's own Rust -- RawEncoder / Encoder / Decoder, untouched -- runs on the GPU.
//
// What the crate calls, and where (all of lib/src):
//   ZSTD_createCCtx / ZSTD_freeCCtx                       encode.rs:130, 137
//   ZSTD_CCtx_setParameter (100 level, 201 checksum)      encode.rs:281-284     (101 windowLog, 160 LDM: cli/src/compress.rs:33-36)
//   ZSTD_CCtx_refPrefix                                   encode.rs:336
//   ZSTD_compressStream2 (e_continue 0 / e_end 2)         encode.rs:341-345, 444-448
//   ZSTD_CCtx_reset (session_only 1)                      encode.rs:504-506
//   ZSTD_CStreamOutSize / ZSTD_CStreamInSize              encode.rs:599, cli/src/compress.rs:60
//   ZSTD_createDCtx / ZSTD_freeDCtx                       decode.rs:31, 38
//   ZSTD_decompressStream                                 decode.rs:243-245
//   ZSTD_DCtx_refPrefix / ZSTD_DCtx_reset                 decode.rs:213, 250-253, 354-356
//   ZSTD_DCtx_setParameter (100 windowLogMax)             cli/src/decompress.rs:56
//   ZSTD_DStreamInSize / ZSTD_DStreamOutSize              decode.rs:181, 184
//   ZSTD_isError / ZSTD_getErrorCode / ZSTD_getErrorName  error.rs:68, 125 (and zstd-safe's result parsing)
//
// The shape does not fit -- libzstd is handed 128 KiB at a time and answers from the same call, this engine works on whole frames --
// so a context BUFFERS a frame: compressStream2(e_continue) takes the input and emits nothing, the first e_end encodes the frame
// (zk_encode_frames, one frame) and the calls from there on hand its bytes out; decompressStream takes input until a frame is
// complete (it follows the block headers: it never takes a byte of the NEXT frame), asks the engine for the frame's size when the
// header does not say (zk_frame_content_sizes -- zeekstd's own frames carry no Frame_Content_Size), decodes it (zk_decode_frames,
// checksum verified) and hands the bytes out.  One engine call per frame: this is the compatibility proof, Level B is the fast path.
// What follows from "e_continue emits nothing": the unmodified crate's FrameSizePolicy::Compressed(n) counts the bytes compressStream2 hands
// out (encode.rs:340-351, 537-541) -- none before e_end -- so under this shim that policy never closes a frame on its own: frames end
// where Uncompressed(n) / end_frame() / SEEKABLE_MAX_FRAME_SIZE end them (and a context buffers up to that: 1 GiB).  Compressed(n) is
// honoured at Levels A / B only (host/encoder.cpp); INTEGRATION.md, Level C.
// Return values follow zstd.h: size_t, an error is (size_t)-ZSTD_ErrorCode; compressStream2(e_end) returns 0 when the frame is out,
// decompressStream returns 0 when a frame is decoded AND flushed (decode.rs:246-255 resets on exactly that).
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <mutex>
#include <new>
#include <vector>
#include "../../../include/zeekstd_amd.h"

extern "C" {
typedef struct { const void *src; size_t size; size_t pos; } ZSTD_inBuffer;
typedef struct { void *dst; size_t size; size_t pos; } ZSTD_outBuffer;
}

namespace {
constexpr size_t ZERR_GENERIC = 1, ZERR_PREFIX_UNKNOWN = 10, ZERR_FRAMEPARAM_UNSUPPORTED = 14, ZERR_WINDOW_TOO_LARGE = 16, ZERR_PARAM_UNSUPPORTED = 40, ZERR_PARAM_OOB = 42, ZERR_STAGE_WRONG = 60,
                 ZERR_MEMORY = 64, ZERR_DST_TOO_SMALL = 70, ZERR_SRC_SIZE_WRONG = 72, ZERR_CORRUPTION = 20, ZERR_MAXCODE = 120;
inline size_t zerr(size_t code) { return (size_t)0 - code; }
inline size_t from_zk(int rc) { return rc == 0 ? 0 : zerr(rc < 0 && (size_t)(-rc) < ZERR_MAXCODE ? (size_t)(-rc) : ZERR_GENERIC); }   // Level A: -(ZSTD_ErrorCode); its own -1000.. codes -> GENERIC
constexpr uint64_t MAX_FRAME = 0x40000000ull;            // SEEKABLE_MAX_FRAME_SIZE, lib/src/lib.rs:56: what one context may buffer

// one engine for every context of the process (device 0), one call at a time: Level-A handles are single-thread-at-a-time, libzstd's
// contexts are independent of each other
std::mutex g_mu;
zk_engine *g_engine = nullptr;
int engine(zk_engine **out)
{
    if (!g_engine) { const int rc = zk_engine_create(0, &g_engine); if (rc) { g_engine = nullptr; return rc; } }
    *out = g_engine;
    return 0;
}
}  // namespace

extern "C" {

struct ZSTD_CCtx_s {
    int level = 3, checksum = 0;                         // ZSTD_CLEVEL_DEFAULT; ZSTD_c_checksumFlag off
    std::vector<uint8_t> in, out;                        // the frame's input so far; its encoded bytes once e_end came
    size_t out_pos = 0;
    bool ended = false;
    const uint8_t *prefix = nullptr; size_t plen = 0;    // ZSTD_CCtx_refPrefix: kept by reference, for the next frame only (zstd.h)
};
struct ZSTD_DCtx_s {
    std::vector<uint8_t> acc, out;                       // the frame's bytes so far (a skippable frame: its 8-byte header at most); its decoded bytes
    size_t out_pos = 0;
    size_t eaten = 0;                                    // bytes of the frame taken from the caller so far (== acc.size() unless the frame is skippable)
    size_t failed = 0;                                   // the error this frame ended with: every call returns it until ZSTD_DCtx_reset (libzstd: a context
                                                         // that has failed is unusable until it is reset)
    size_t next_hdr = 0;                                 // where the next block header starts (0: the frame header is not parsed yet)
    size_t total = 0;                                    // the frame's length once the last block's header was seen
    bool have = false;                                   // `out` holds the frame (its last input byte is still the caller's)
    bool skippable = false, cks = false;
    uint64_t fcs = ~0ull;
    uint32_t block_max = 131072;                         // Block_Maximum_Size of the frame on hand: min(window, 128 KiB) (RFC 8878 3.1.1.2.3)
    int wlog_max = 27;                                   // ZSTD_d_windowLogMax (default ZSTD_WINDOWLOG_LIMIT_DEFAULT): a frame may declare a window of (1 << wlog_max) + 1 at most
    const uint8_t *prefix = nullptr; size_t plen = 0;
};
typedef ZSTD_CCtx_s ZSTD_CCtx;
typedef ZSTD_DCtx_s ZSTD_DCtx;

unsigned ZSTD_versionNumber(void) { return 10507; }      // speaks the API of the version the crate pins (Cargo.lock:1192-1193)
const char *ZSTD_versionString(void) { return "1.5.7"; }
unsigned ZSTD_isError(size_t code) { return code > zerr(ZERR_MAXCODE); }
int ZSTD_getErrorCode(size_t code) { return ZSTD_isError(code) ? (int)(0 - code) : 0; }
const char *ZSTD_getErrorName(size_t code) { return ZSTD_isError(code) ? zk_error_name(-(int)(0 - code)) : "No error detected"; }
const char *ZSTD_getErrorString(int code) { return zk_error_name(-code); }
size_t ZSTD_CStreamInSize(void) { return 131072; }
size_t ZSTD_CStreamOutSize(void) { return 131591; }
size_t ZSTD_DStreamInSize(void) { return 131075; }
size_t ZSTD_DStreamOutSize(void) { return 131072; }
size_t ZSTD_compressBound(size_t n) { return (size_t)zk_compress_bound(n, (uint32_t)(n > MAX_FRAME ? MAX_FRAME : (n ? n : 1))); }

// ---------------------------------------------------------------------------------------------- compression
ZSTD_CCtx *ZSTD_createCCtx(void) { return new (std::nothrow) ZSTD_CCtx_s(); }
size_t ZSTD_freeCCtx(ZSTD_CCtx *c) { delete c; return 0; }
size_t ZSTD_CCtx_setParameter(ZSTD_CCtx *c, int param, int value)
{
    if (!c) return zerr(ZERR_GENERIC);
    if (!c->in.empty() || c->ended) return zerr(ZERR_STAGE_WRONG);          // libzstd: parameters can only be changed between frames
    switch (param) {
    case 100: c->level = value; return 0;                                    // ZSTD_c_compressionLevel
    case 201: c->checksum = value != 0; return 0;                            // ZSTD_c_checksumFlag
    case 200: case 202: return 0;                                            // contentSizeFlag / dictIDFlag: streaming frames carry neither here
    case 101:                                                                // ZSTD_c_windowLog: the engine picks the window (it covers prefix + frame in patch mode, what
        return value == 0 || (value >= 10 && value <= 31) ? 0 : zerr(ZERR_PARAM_OOB);   // cli/src/compress.rs:33-34 asks for)
    case 160: return 0;                                                      // ZSTD_c_enableLongDistanceMatching: a long prefix is reached through the engine's own table
    default: return zerr(ZERR_PARAM_UNSUPPORTED);
    }
}
size_t ZSTD_CCtx_refPrefix(ZSTD_CCtx *c, const void *prefix, size_t len)
{
    if (!c) return zerr(ZERR_GENERIC);
    if (!c->in.empty() || c->ended) return zerr(ZERR_STAGE_WRONG);
    c->prefix = len ? (const uint8_t *)prefix : nullptr; c->plen = prefix ? len : 0;
    return 0;
}
size_t ZSTD_CCtx_reset(ZSTD_CCtx *c, int directive)                          // 1 session_only, 2 parameters, 3 both
{
    if (!c) return zerr(ZERR_GENERIC);
    if (directive == 1 || directive == 3) { c->in.clear(); c->out.clear(); c->out_pos = 0; c->ended = false; c->prefix = nullptr; c->plen = 0; }
    if (directive == 2 || directive == 3) {
        if (directive == 2 && (!c->in.empty() || c->ended)) return zerr(ZERR_STAGE_WRONG);
        c->level = 3; c->checksum = 0;
    }
    return 0;
}
size_t ZSTD_compressStream2(ZSTD_CCtx *c, ZSTD_outBuffer *out, ZSTD_inBuffer *in, int directive)
{
    if (!c || !out || !in || out->pos > out->size || in->pos > in->size || directive < 0 || directive > 2) return zerr(ZERR_GENERIC);
    if (c->ended && c->out_pos == c->out.size()) {                           // the last frame is out and nobody reset the session: the next frame begins
        c->in.clear(); c->out.clear(); c->out_pos = 0; c->ended = false; c->prefix = nullptr; c->plen = 0;
    }
    if (!c->ended) {
        const size_t n = in->size - in->pos;
        if (n) {
            if ((uint64_t)c->in.size() + n > MAX_FRAME) return zerr(ZERR_SRC_SIZE_WRONG);
            try { c->in.insert(c->in.end(), (const uint8_t *)in->src + in->pos, (const uint8_t *)in->src + in->size); } catch (...) { return zerr(ZERR_MEMORY); }
            in->pos = in->size;
        }
        if (directive != 2) return 0;                                        // e_continue / e_flush: nothing is pending that a buffer could take (a flush
                                                                             // cannot emit half a frame here; zeekstd never asks for one)
        // e_end: the frame is complete -> one Level-A call
        const uint64_t d = c->in.size();
        const uint32_t fsz = (uint32_t)(d ? d : 1);
        try { c->out.resize((size_t)zk_compress_bound(d, fsz) + 64); } catch (...) { return zerr(ZERR_MEMORY); }
        uint32_t cs = 0, ds = 0, nf = 0; uint64_t written = 0;
        int rc;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            zk_engine *e;
            if ((rc = engine(&e))) return from_zk(rc);
            static const uint8_t none = 0;
            rc = zk_encode_frames_prefix(e, d ? c->in.data() : &none, d, fsz, c->level, c->checksum, c->prefix, c->plen, c->out.data(), c->out.size(), &cs, &ds, 1, &nf, &written);
        }
        if (rc) return from_zk(rc);
        if (nf != 1 || written != cs) return zerr(ZERR_GENERIC);
        c->out.resize((size_t)written); c->out_pos = 0; c->ended = true;
        std::vector<uint8_t>().swap(c->in);
    } else if (in->size != in->pos) return zerr(ZERR_STAGE_WRONG);           // input while a finished frame is still being handed out
    const size_t room = out->size - out->pos, left = c->out.size() - c->out_pos, k = room < left ? room : left;
    if (k) memcpy((uint8_t *)out->dst + out->pos, c->out.data() + c->out_pos, k);
    out->pos += k; c->out_pos += k;
    return c->out.size() - c->out_pos;                                       // 0 <=> the frame is fully written (encode.rs:456-458)
}

// ---------------------------------------------------------------------------------------------- decompression
ZSTD_DCtx *ZSTD_createDCtx(void) { return new (std::nothrow) ZSTD_DCtx_s(); }
size_t ZSTD_freeDCtx(ZSTD_DCtx *d) { delete d; return 0; }
static void dctx_next_frame(ZSTD_DCtx *d)
{
    d->acc.clear(); d->out.clear(); d->out_pos = 0; d->next_hdr = 0; d->total = 0; d->have = false; d->skippable = false; d->cks = false; d->fcs = ~0ull;
    d->block_max = 131072; d->eaten = 0; d->failed = 0;
}
size_t ZSTD_DCtx_reset(ZSTD_DCtx *d, int directive)
{
    if (!d) return zerr(ZERR_GENERIC);
    if (directive == 1 || directive == 3) { dctx_next_frame(d); d->prefix = nullptr; d->plen = 0; }
    if (directive == 2 || directive == 3) d->wlog_max = 27;                  // ZSTD_reset_parameters
    return 0;
}
size_t ZSTD_DCtx_refPrefix(ZSTD_DCtx *d, const void *prefix, size_t len)
{
    if (!d) return zerr(ZERR_GENERIC);
    if (!d->acc.empty() || d->have) return zerr(ZERR_STAGE_WRONG);
    d->prefix = len ? (const uint8_t *)prefix : nullptr; d->plen = prefix ? len : 0;
    return 0;
}
size_t ZSTD_DCtx_setParameter(ZSTD_DCtx *d, int param, int value)
{
    if (!d) return zerr(ZERR_GENERIC);
    if (param == 100) {                                                      // ZSTD_d_windowLogMax (cli/src/decompress.rs:56 raises it for patches): held against the frame header as libzstd
        if (value != 0 && (value < 10 || value > 31)) return zerr(ZERR_PARAM_OOB);   // does -- the engine itself would take any window up to 2^31
        d->wlog_max = value ? value : 27;
        return 0;
    }
    return zerr(ZERR_PARAM_UNSUPPORTED);
}

// the frame's bytes as the context sees them: what it took so far, then what the caller holds
struct View {
    const uint8_t *a; size_t na; const uint8_t *b; size_t nb;
    size_t size() const { return na + nb; }
    uint8_t operator[](size_t i) const { return i < na ? a[i] : b[i - na]; }
};
// How many bytes of the frame are KNOWN to be needed, given its first v.size() bytes: follows the frame header and then block header
// after block header (RFC 8878 3.1.1).  d->total is set once the last block's header was seen.  Returns 0 for a verdict in *err.
static size_t frame_need(ZSTD_DCtx *d, const View &a, size_t *err)
{
    *err = 0;
    if (d->total) return d->total;
    if (d->next_hdr == 0) {
        if (a.size() < 4) return 4;
        const uint32_t magic = (uint32_t)a[0] | (uint32_t)a[1] << 8 | (uint32_t)a[2] << 16 | (uint32_t)a[3] << 24;
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                         // a skippable frame (the seek table is one): its size field says it all
            if (a.size() < 8) return 8;
            d->skippable = true;
            d->total = 8 + ((size_t)a[4] | (size_t)a[5] << 8 | (size_t)a[6] << 16 | (size_t)a[7] << 24);
            return d->total;
        }
        if (magic != 0xFD2FB528u) { *err = zerr(ZERR_PREFIX_UNKNOWN); return 0; }
        if (a.size() < 5) return 5;
        const uint32_t fhd = a[4], fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did = fhd & 3;
        const size_t dl = did == 3 ? 4 : did, fl = fcs_flag == 0 ? single : (size_t)1 << fcs_flag, hdr = 5 + (single ? 0 : 1) + dl + fl;
        if (a.size() < hdr) return hdr;
        if (fhd & 8) { *err = zerr(ZERR_FRAMEPARAM_UNSUPPORTED); return 0; }   // the reserved bit: refused before a byte of the frame is waited for, as libzstd does
        d->cks = (fhd >> 2) & 1;
        if (fl) {
            uint64_t v = 0;
            for (size_t i = 0; i < fl; i++) v |= (uint64_t)a[hdr - fl + i] << (8 * i);
            d->fcs = fl == 2 ? v + 256 : v;
        }
        // a block header that asks for more than Block_Maximum_Size is damage, known the moment the header is read -- not a reason to
        // wait for that many bytes (a flipped size bit would otherwise swallow the rest of the stream as "input still to come")
        uint64_t window = d->fcs;
        if (!single) { const uint32_t wd = a[5], e = wd >> 3, m = wd & 7; window = (1ull << (10 + e)); window += (window >> 3) * m; }
        d->block_max = window < 131072 ? (uint32_t)window : 131072;
        if (window > (1ull << d->wlog_max) + 1) { *err = zerr(ZERR_WINDOW_TOO_LARGE); return 0; }   // "Frame requires too much memory for decoding" (ZSTD_decompressStream: windowSize > maxWindowSize)
        d->next_hdr = hdr;
    }
    for (;;) {
        if (a.size() < d->next_hdr + 3) return d->next_hdr + 3;
        const uint32_t h = (uint32_t)a[d->next_hdr] | (uint32_t)a[d->next_hdr + 1] << 8 | (uint32_t)a[d->next_hdr + 2] << 16;
        const uint32_t last = h & 1, type = (h >> 1) & 3, size = h >> 3;
        if (type == 3 || size > d->block_max) { *err = zerr(ZERR_CORRUPTION); return 0; }
        d->next_hdr += 3 + (type == 1 ? 1 : size);
        if (last) { d->total = d->next_hdr + (d->cks ? 4 : 0); return d->total; }
    }
}

size_t ZSTD_decompressStream(ZSTD_DCtx *d, ZSTD_outBuffer *out, ZSTD_inBuffer *in)
{
    if (!d || !out || !in || out->pos > out->size || in->pos > in->size) return zerr(ZERR_GENERIC);
    if (d->failed) return d->failed;
#define ZK_FAIL(code) do { d->failed = (code); return d->failed; } while (0)
    if (!d->have) {
        // Take input -- never a byte beyond the frame, and not the frame's LAST byte either until its output is out: zeekstd's loop
        // stops calling once the input it holds is consumed (decode.rs:243 `in_buffer.pos() < in_len`), as libzstd leaves a block's
        // input with the caller while it cannot flush.  The headers are read THROUGH the caller's buffer before anything is taken, so
        // the last byte is known for what it is while it is still the caller's.
        const uint8_t *src = (const uint8_t *)in->src + in->pos;
        const size_t avail = in->size - in->pos;
        size_t err;
        const size_t need = frame_need(d, View{d->acc.data(), d->acc.size(), src, avail}, &err);
        if (err) ZK_FAIL(err);
        // A skippable frame (a Foot seek table can be a GiB) is walked over, not kept: nothing of it is needed once its size field is read
        const bool skip = d->skippable && d->total;
        if (d->total && d->eaten >= d->total) ZK_FAIL(zerr(ZERR_GENERIC));   // (cannot happen: the last byte is only ever taken below)
        const bool whole = d->total && d->eaten + avail >= d->total;        // the frame's last byte is in sight
        const size_t take = whole ? d->total - 1 - d->eaten : avail;        // (eaten < total: take >= 0)
        if (!skip) {
            if ((uint64_t)d->acc.size() + take > MAX_FRAME + (MAX_FRAME >> 7) + 1024) ZK_FAIL(zerr(ZERR_SRC_SIZE_WRONG));
            try { d->acc.insert(d->acc.end(), src, src + take); } catch (...) { ZK_FAIL(zerr(ZERR_MEMORY)); }
        }
        d->eaten += take;
        in->pos += take;
        if (!whole) return need - d->eaten;                                 // a hint, as libzstd gives one
        d->out.clear(); d->out_pos = 0;
        if (!d->skippable) {
            try { d->acc.push_back(src[take]); d->acc.resize(d->total + 8); } catch (...) { ZK_FAIL(zerr(ZERR_MEMORY)); }   // a COPY of the last byte (in->pos stays) + readable padding (ZK_COMP_PADDING)
            uint64_t c_off[2] = {0, d->total}, d_off[2] = {0, 0};
            int32_t st = 0;
            int rc;
            std::lock_guard<std::mutex> lk(g_mu);
            zk_engine *e;
            if ((rc = engine(&e))) ZK_FAIL(from_zk(rc));
            uint64_t dsz = d->fcs;
            if (dsz == ~0ull) {                                              // no Frame_Content_Size (zeekstd's own frames): the engine walks the frame
                if ((rc = zk_frame_content_sizes(e, d->acc.data(), d->total, c_off, 0, 1, &dsz, &st))) ZK_FAIL(from_zk(rc));
                if (st) ZK_FAIL(from_zk(st));
            }
            if (dsz > MAX_FRAME) ZK_FAIL(zerr(ZERR_PARAM_UNSUPPORTED));
            d_off[1] = dsz;
            try { d->out.resize((size_t)dsz + 64); } catch (...) { ZK_FAIL(zerr(ZERR_MEMORY)); }
            rc = zk_decode_frames_prefix(e, d->acc.data(), d->total, c_off, d_off, 0, 1, d->prefix, d->plen, d->out.data(), d->out.size(), 1, &st);
            if (rc) ZK_FAIL(from_zk(st ? st : rc));
            d->out.resize((size_t)dsz);
        }
        d->have = true;
    }
#undef ZK_FAIL
    const size_t room = out->size - out->pos, left = d->out.size() - d->out_pos, k = room < left ? room : left;
    if (k) memcpy((uint8_t *)out->dst + out->pos, d->out.data() + d->out_pos, k);
    out->pos += k; d->out_pos += k;
    if (d->out_pos < d->out.size()) return d->out.size() - d->out_pos;       // more to hand out: the caller comes back with room
    // the frame is out: now its last input byte is taken, and 0 says "frame decoded and flushed" (decode.rs:246-255)
    if (in->pos < in->size) in->pos += 1; else return 1;                     // (the caller presents the byte it was not relieved of)
    dctx_next_frame(d);
    return 0;
}

}  // extern "C"
