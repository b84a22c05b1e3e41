// zk_engine_host.hip -- the host-pointer half of the batch engine: what a zeekstd user actually calls.
//
// The reference streams through 128 KiB buffers with no copy cost (lib/src/encode.rs:779-787, decode.rs:222-225,
// cli/src/compress.rs:60-82).  A GPU engine pays PCIe both ways, so the host-pointer entry points are a pipeline:
//
//   caller memory --(worker threads, parallel memcpy)--> pinned ring --(H2D queue)--> HBM chunk buffers
//        --> decode / encode kernels (the engine's two decode contexts alternate; one encode in flight)
//        --> HBM chunk buffers --(D2H queue)--> pinned ring --(worker threads)--> caller memory / sink
//
// Everything is chunked (decode: ~256 MiB of output per chunk, encode: enough frames to fill the GPU), the copy queues
// run beside the compute queues, and a caller buffer that is itself pinned (zk_host_alloc) is moved by DMA directly,
// without the staging copies.  Small requests (a seek: one frame) skip the worker threads and the rings altogether.
// No CPU fallback anywhere: the threads only move bytes.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <memory>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/zeekstd_amd.h"
#include "zk_engine.h"
#include "zk_kernels.h"

#define ZK_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t _e = (call);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            e->last_err = std::string(#call) + ": " + hipGetErrorString(_e);                         \
            return ZK_ERR_HIP;                                                                       \
        }                                                                                            \
    } while (0)

// ---------------------------------------------------------------------------------------------- worker threads
namespace {

class ZkPool {
public:
    ZkPool(int n, int device) : device_(device)
    {
        for (int i = 0; i < n; i++) th_.emplace_back([this] { run(); });
    }
    ~ZkPool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    void post(std::function<void()> f)
    {
        { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(f)); }
        cv_.notify_one();
    }
    // blocking multi-threaded memcpy; the caller takes a share
    void copy(void *dst, const void *src, size_t n)
    {
        const size_t min_piece = 1u << 20;
        size_t pieces = n / min_piece;
        if (pieces > (size_t)size() + 1) pieces = (size_t)size() + 1;
        if (pieces <= 1) { memcpy(dst, src, n); return; }
        const size_t per = ((n + pieces - 1) / pieces + 4095) & ~(size_t)4095;
        // the latch lives on the heap and is shared with the workers: the last worker may still be inside notify while the
        // caller, woken by the count, has already returned (a latch on this frame would be dead by then)
        struct Latch { std::mutex m; std::condition_variable cv; size_t left; };
        auto latch = std::make_shared<Latch>();
        latch->left = pieces - 1;
        for (size_t k = 1; k < pieces; k++) {
            const size_t at = k * per;
            if (at >= n) { std::lock_guard<std::mutex> g(latch->m); latch->left--; continue; }
            const size_t len = n - at < per ? n - at : per;
            post([=] {
                memcpy((uint8_t *)dst + at, (const uint8_t *)src + at, len);
                std::lock_guard<std::mutex> g(latch->m);
                if (--latch->left == 0) latch->cv.notify_one();
            });
        }
        memcpy(dst, src, per < n ? per : n);
        std::unique_lock<std::mutex> g(latch->m);
        latch->cv.wait(g, [&] { return latch->left == 0; });
    }

private:
    void run()
    {
        (void)hipSetDevice(device_);
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
        }
    }
    int device_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    bool stop_ = false;
};

// A ring of pinned staging pieces.  A piece is busy from acquire() until its consumer calls release() (copy-out done) or
// until the event recorded after its H2D has passed (stage-in): acquire() waits for whichever applies.
struct ZkRing {
    struct Piece { uint8_t *p = nullptr; hipEvent_t ev = nullptr; bool ev_pending = false; bool held = false; };
    std::vector<Piece> pc;
    size_t piece_bytes = 0;
    size_t next = 0;
    std::mutex m;
    std::condition_variable cv;

    int init(size_t pieces, size_t bytes)
    {
        piece_bytes = bytes;
        pc.resize(pieces);
        for (auto &x : pc) {
            if (hipHostMalloc((void **)&x.p, bytes, hipHostMallocDefault) != hipSuccess) return -1;
            if (hipEventCreateWithFlags(&x.ev, hipEventDisableTiming) != hipSuccess) return -1;
        }
        return 0;
    }
    void destroy()
    {
        for (auto &x : pc) { if (x.p) (void)hipHostFree(x.p); if (x.ev) (void)hipEventDestroy(x.ev); }
        pc.clear();
    }
    // next piece in ring order, free of its previous use
    int acquire()
    {
        const int k = (int)next;
        next = (next + 1) % pc.size();
        Piece &x = pc[k];
        { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return !x.held; }); }
        if (x.ev_pending) { (void)hipEventSynchronize(x.ev); x.ev_pending = false; }
        return k;
    }
    void hold(int k) { std::lock_guard<std::mutex> g(m); pc[k].held = true; }
    void release(int k) { { std::lock_guard<std::mutex> g(m); pc[k].held = false; } cv.notify_all(); }
    void wait_all_released()
    {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { for (auto &x : pc) if (x.held) return false; return true; });
    }
};

// Caller memory pinned on the fly.  hipHostRegister costs ~3 ms per 256 MiB on this platform and DMA then runs at the
// full PCIe rate straight from / into the caller's pages, so the staging copy through the pinned rings (and the worker
// threads' memcpy, which a remote NUMA node can slow to a few GB/s per thread) is only the fallback.  The range is cut
// into page-aligned units that worker threads register in order, ahead of the copies that need them; a unit that cannot
// be registered (read-only mapping, already registered by the caller, ...) sends its copies through the rings.
struct ZkRegWindow {
    static constexpr size_t UNIT = 64u << 20;
    uintptr_t base = 0;
    size_t nunits = 0, bytes = 0;
    std::vector<int> state;                      // 0 queued, 1 ok, 2 failed
    std::mutex m;
    std::condition_variable cv;
    size_t done = 0;
    bool active = false;

    void start(ZkPool *pool, const void *p, size_t n)
    {
        if (!n) return;
        const uintptr_t a = (uintptr_t)p & ~(uintptr_t)4095, b = ((uintptr_t)p + n + 4095) & ~(uintptr_t)4095;
        base = a; bytes = b - a;
        nunits = (bytes + UNIT - 1) / UNIT;
        state.assign(nunits, 0);
        active = true;
        // one chain of tasks: unit k + 1 is queued when unit k is done, so registrations run in order, one at a time
        // (they serialise inside the driver anyway) and never occupy more than one worker
        post_unit(pool, 0);
    }
    void post_unit(ZkPool *pool, size_t k)
    {
        pool->post([this, pool, k] {
            const size_t len = bytes - k * UNIT < UNIT ? bytes - k * UNIT : UNIT;
            bool ok = k == 0 || state[k - 1] == 1;      // after a failure the rest is not even tried
            if (ok) ok = hipHostRegister((void *)(base + k * UNIT), len, hipHostRegisterDefault) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            { std::lock_guard<std::mutex> g(m); state[k] = ok ? 1 : 2; done = k + 1; }
            cv.notify_all();
            if (k + 1 < nunits) post_unit(pool, k + 1);
        });
    }
    size_t unit_of(const void *p) const { return ((uintptr_t)p - base) / UNIT; }
    // bytes from p to the end of its unit
    size_t span(const void *p) const { return (size_t)(base + (unit_of(p) + 1) * UNIT - (uintptr_t)p); }
    bool ok(const void *p)
    {
        if (!active) return false;
        const size_t k = unit_of(p);
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return done > k; });
        return state[k] == 1;
    }
    void finish()                                 // every queue that used the window has been synchronised
    {
        if (!active) return;
        { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return done == nunits; }); }
        for (size_t k = 0; k < nunits; k++) if (state[k] == 1) (void)hipHostUnregister((void *)(base + k * UNIT));
        active = false;
    }
};

}  // namespace

struct zk_hostpipe {
    static constexpr int NS = ZK_MAX_CTX + 1;    // decode chunks whose HBM buffers can exist at once (contexts in use + 1)
    int nctx = 2;                                // decode contexts the pipeline rotates through (ZK_CHOICE_PIPE_CONTEXTS)
    uint64_t chunk_target = 0;                   // bytes of output per chunk (0 = by total size; ZK_CHOICE_PIPE_CHUNK_MIB)
    static constexpr size_t PIECE = 32u << 20;   // pinned staging piece
    ZkPool *pool = nullptr;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    ZkRing ring_in, ring_out;                    // allocated on the first large call
    bool rings_ready = false;
    struct Slot {
        zk_devbuf d_in, d_out, d_off, d_st;
        hipEvent_t ev_in = nullptr, ev_dec = nullptr, ev_out = nullptr;
        bool out_pending = false;                // ev_out was recorded for an earlier chunk
    } slot[NS];
    // small pinned areas: offsets / status words of the call in flight, and the staging of small requests
    uint8_t *pin_meta = nullptr; size_t pin_meta_cap = 0;
    uint8_t *pin_small = nullptr; size_t pin_small_cap = 0;
    uint32_t *pin_flag = nullptr; uint32_t small_gen = 0;      // completion word of the small path, written by its last kernel
    // encode: double-buffered HBM chunk buffers
    struct ESlot { zk_devbuf d_src, d_dst, d_sizes; hipEvent_t ev_in = nullptr, ev_enc = nullptr, ev_out = nullptr; bool out_pending = false; } es[2];
};

static int zk_pin_grow(zk_engine *e, uint8_t *&p, size_t &cap, size_t bytes)
{
    if (bytes <= cap) return 0;
    if (p) ZK_HIP(hipHostFree(p));
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    ZK_HIP(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
    cap = want;
    return 0;
}

static int zk_hostpipe_get(zk_engine *e, zk_hostpipe **out)
{
    if (e->hp) { *out = e->hp; return 0; }
    zk_hostpipe *hp = new zk_hostpipe();
    int n = e->host_threads;
    if (n <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        n = (int)(hw / 8);
        if (n < 2) n = 2;
        if (n > 16) n = 16;
    }
    hp->pool = new ZkPool(n, e->device);
    e->hp = hp;
    ZK_HIP(hipStreamCreateWithFlags(&hp->s_h2d, hipStreamNonBlocking));
    ZK_HIP(hipStreamCreateWithFlags(&hp->s_d2h, hipStreamNonBlocking));
    for (auto &s : hp->slot) {
        ZK_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&s.ev_dec, hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    }
    ZK_HIP(hipHostMalloc((void **)&hp->pin_flag, 64, hipHostMallocDefault));
    *hp->pin_flag = 0;
    for (auto &s : hp->es) {
        ZK_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&s.ev_enc, hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    }
    *out = hp;
    return 0;
}

static int zk_hostpipe_rings(zk_engine *e, zk_hostpipe *hp)
{
    if (hp->rings_ready) return 0;
    if (hp->ring_in.init(4, zk_hostpipe::PIECE) != 0 || hp->ring_out.init(16, zk_hostpipe::PIECE) != 0) {
        e->last_err = "hipHostMalloc of the pinned staging rings failed";
        return ZK_ERR_HIP;
    }
    hp->rings_ready = true;
    return 0;
}

int zk_hostpipe_create(zk_engine *e) { zk_hostpipe *hp = nullptr; return zk_hostpipe_get(e, &hp); }
void zk_hostpipe_tune(zk_engine *e)
{
    if (!e->hp) return;
    e->hp->nctx = e->pipe_contexts ? e->pipe_contexts : 2;
    e->hp->chunk_target = e->pipe_chunk_bytes;
}

void zk_hostpipe_destroy(zk_engine *e)
{
    zk_hostpipe *hp = e->hp;
    if (!hp) return;
    if (hp->s_h2d) (void)hipStreamSynchronize(hp->s_h2d);
    if (hp->s_d2h) (void)hipStreamSynchronize(hp->s_d2h);
    delete hp->pool;
    hp->ring_in.destroy(); hp->ring_out.destroy();
    for (auto &s : hp->slot) {
        for (zk_devbuf *b : {&s.d_in, &s.d_out, &s.d_off, &s.d_st}) if (b->p) (void)hipFree(b->p);
        for (hipEvent_t ev : {s.ev_in, s.ev_dec, s.ev_out}) if (ev) (void)hipEventDestroy(ev);
    }
    for (auto &s : hp->es) {
        for (zk_devbuf *b : {&s.d_src, &s.d_dst, &s.d_sizes}) if (b->p) (void)hipFree(b->p);
        for (hipEvent_t ev : {s.ev_in, s.ev_enc, s.ev_out}) if (ev) (void)hipEventDestroy(ev);
    }
    if (hp->pin_meta) (void)hipHostFree(hp->pin_meta);
    if (hp->pin_small) (void)hipHostFree(hp->pin_small);
    if (hp->pin_flag) (void)hipHostFree(hp->pin_flag);
    if (hp->s_h2d) (void)hipStreamDestroy(hp->s_h2d);
    if (hp->s_d2h) (void)hipStreamDestroy(hp->s_d2h);
    delete hp;
    e->hp = nullptr;
}

// true when DMA can address the range directly (hipHostMalloc / hipHostRegister memory, e.g. zk_host_alloc)
static bool zk_is_pinned(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

extern "C" void *zk_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void zk_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int zk_engine_set_host_threads(zk_engine *e, int n)
{
    if (!e || n < 0 || n > 256) return ZK_ERR_ARGUMENT;
    if (e->hp && e->hp->pool && e->hp->pool->size() != n && n > 0) {   // takes effect at once
        delete e->hp->pool;
        e->hp->pool = new ZkPool(n, e->device);
    }
    e->host_threads = n;
    return 0;
}

void zk_host_copy(zk_engine *e, void *dst, const void *src, size_t n)
{
    zk_hostpipe *hp = nullptr;
    if (n < (4u << 20) || zk_hostpipe_get(e, &hp) != 0) { memcpy(dst, src, n); return; }
    hp->pool->copy(dst, src, n);
}

int zk_engine_stage_prefix(zk_engine *e, const void *owner, const uint8_t *prefix, uint64_t len, bool force, const void **d_out)
{
    *d_out = nullptr;
    if (!prefix || !len) return 0;
    if (len > ZK_MAX_PREFIX) return -(int)ZK_E_WINDOW_TOO_LARGE;
    ZK_HIP(hipSetDevice(e->device));
    if (force || !owner || e->st_prefix_owner != owner || e->st_prefix_src != prefix || e->st_prefix_len != len) {
        int rc;
        if ((rc = zk_devbuf_reserve(e, e->st_prefix, (size_t)len + 64))) return rc;
        ZK_HIP(hipMemcpyAsync(e->st_prefix.p, prefix, len, hipMemcpyHostToDevice, e->stream));
        ZK_HIP(hipStreamSynchronize(e->stream));
        e->st_prefix_owner = owner; e->st_prefix_src = prefix; e->st_prefix_len = len;
    }
    *d_out = e->st_prefix.p;
    return 0;
}

// ---------------------------------------------------------------------------------------------- decode
namespace {
struct ZkChunk { uint32_t f0, f1; uint64_t c0, c1, d0, d1; };
}

// fills `n` bytes of the source at payload offset `off` into dst (pinned)
static int zk_src_fill(zk_engine *e, zk_hostpipe *hp, const zk_host_src &src, uint64_t off, uint8_t *dst, size_t n)
{
    if (src.mem) { if (n >= (4u << 20)) hp->pool->copy(dst, src.mem + off, n); else memcpy(dst, src.mem + off, n); return 0; }
    size_t got = 0;
    while (got < n) {
        const size_t k = src.read(src.user, off + got, dst + got, n - got);
        if (k == 0) return -(int)ZK_E_SRC_SIZE_WRONG;         // the source ends inside a frame
        got += k;
    }
    (void)e;
    return 0;
}

// ---------------------------------------------------------------------------------------------- decode: the small path
// A seek (one frame, a few): four or five kernel launches on one queue, no copy command, no read-back in the middle, and
// the host spins on a pinned completion word instead of a stream synchronisation.  See zk_decode.hip (zk_k_small_*).
// *fallback is set when the batch does not fit the small path's fixed scratch (the general path takes it then).
static int zk_decode_small(zk_engine *e, zk_hostpipe *hp, const zk_host_src &src, const uint64_t *c_off, const uint64_t *d_off, uint32_t first,
                           uint32_t count, const void *d_prefix, uint64_t prefix_len, uint8_t *dst, bool dst_pinned, int verify,
                           int32_t *frame_status, uint32_t *n_ok, bool *fallback)
{
    *fallback = false;
    const uint64_t c_lo = c_off[first], c_hi = c_off[first + count], d_lo = d_off[first], d_hi = d_off[first + count];
    const uint64_t csz = c_hi - c_lo, dsz = d_hi - d_lo;
    int rc;
    // pinned staging: offsets | compressed bytes | status words | (output bytes when the caller's buffer is not pinned)
    const size_t offs_bytes = ((size_t)(count + 1) * 16 + 63) & ~(size_t)63;
    const size_t comp_bytes = ((size_t)csz + 15 + 64) & ~(size_t)63;
    const size_t stat_bytes = ((size_t)count * 4 + 63) & ~(size_t)63;
    if ((rc = zk_pin_grow(e, hp->pin_small, hp->pin_small_cap, offs_bytes + comp_bytes + stat_bytes + (dst_pinned ? 0 : (size_t)dsz + 64)))) return rc;
    uint64_t *h_offs = (uint64_t *)hp->pin_small;
    uint8_t *h_comp = hp->pin_small + offs_bytes;
    int32_t *h_status = (int32_t *)(h_comp + comp_bytes);
    uint8_t *h_out = dst_pinned ? dst : (uint8_t *)h_status + stat_bytes;
    for (uint32_t k = 0; k <= count; k++) { h_offs[k] = c_off[first + k] - c_lo; h_offs[count + 1 + k] = d_off[first + k] - d_lo; }
    if ((rc = zk_src_fill(e, hp, src, c_lo, h_comp, (size_t)csz))) return rc;
    memset(h_comp + csz, 0, comp_bytes - (size_t)csz);
    for (uint32_t k = 0; k < count; k++) h_status[k] = -1;

    // scratch from bounds the host knows: every sequence regenerates >= 3 bytes, literals never exceed the output
    zk_dec_ctx c = zk_dec_context(e, 0, nullptr);
    zk_hostpipe::Slot &s = hp->slot[0];
    const uint32_t block_cap = (uint32_t)(1024 + dsz / 1024 + 8ull * count);
    if ((rc = zk_devbuf_reserve(e, c.infos, (size_t)count * sizeof(ZkFrameInfo)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.bases, (size_t)count * sizeof(ZkFrameBase)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.words, 16 * sizeof(uint64_t)))) return rc;
    if ((rc = zk_devbuf_reserve(e, c.blocks, (size_t)(block_cap + 1) * sizeof(ZkBlock)))) return rc;
    const uint64_t seq_cap = dsz / 3 + count + 1 + 8ull * (block_cap + 1);        // record slots: what the walk may hand out (it checks)
    if ((rc = zk_devbuf_reserve(e, c.seqs, (size_t)seq_cap * sizeof(ZkSeqP)))) return rc;     // (+ 7 per block: a block's records start on a 64-byte line)
    if ((rc = zk_devbuf_reserve(e, c.lit, (size_t)dsz + 64))) return rc;
    if ((rc = zk_devbuf_reserve(e, s.d_in, comp_bytes + 64))) return rc;
    if ((rc = zk_devbuf_reserve(e, s.d_out, (size_t)dsz + 64))) return rc;
    if ((rc = zk_devbuf_reserve(e, s.d_off, offs_bytes))) return rc;
    if ((rc = zk_devbuf_reserve(e, s.d_st, stat_bytes))) return rc;
    hipStream_t st = c.st;
    ZkFrameInfo *infos = (ZkFrameInfo *)c.infos.p;
    ZkBlock *blocks = (ZkBlock *)c.blocks.p;
    uint64_t *words = (uint64_t *)c.words.p, *d_offs = (uint64_t *)s.d_off.p;
    const uint8_t *comp = (const uint8_t *)s.d_in.p;
    const uint32_t gen = ++hp->small_gen;
    zk_launch_small_walk(st, h_comp, csz, h_offs, count, dsz, block_cap, seq_cap, (uint8_t *)s.d_in.p, d_offs, infos, (ZkFrameBase *)c.bases.p, blocks, words);
    uint32_t groups = (block_cap + 15) / 16;
    if (groups > 32) groups = 32;
    zk_launch_small_entropy(st, comp, blocks, words, (uint8_t *)c.lit.p, (ZkSeqP *)c.seqs.p, groups, e->choice.small_path == 2);
    // long frames (a handful of 2 MiB ones: configs[0]): their checksum chains -- 2.9 ms per 2 MiB, twice what the executor takes --
    // start with the executor, on the context's second queue (zk_k_xxh64_follow; zk_follow_wanted)
    uint64_t *prog = nullptr;
    if (verify && zk_follow_wanted(e, count, dsz, true)) {
        zk_engine::DecCtx &x = e->dctx[0];
        if ((rc = zk_devbuf_reserve(e, x.prog, (size_t)count * sizeof(uint64_t)))) return rc;
        if ((rc = zk_dec_ctx_aux(e, 0))) return rc;
        prog = (uint64_t *)x.prog.p;
        ZK_HIP(hipMemsetAsync(prog, 0, (size_t)count * sizeof(uint64_t), st));
        ZK_HIP(hipEventRecord(x.ev_fork, st));
        ZK_HIP(hipStreamWaitEvent(x.aux, x.ev_fork, 0));
    }
    // long frames (a read of one or two 2 MiB frames -- zeekstd's default frame size): the executor in segments, several workgroups per
    // frame (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill_lds: 1.37 -> 0.6 ms for a 2 MiB frame); the host knows the frames' sizes here
    // ... and SHORT frames in a handful (a seek into 64 KiB frames: sixteen blocks of 4 KiB as this encoder writes them, executed one
    // after the other at ~7 us each by a frame's workgroup): a segment per block, all at once, and one turn of the fill pass for the frame
    uint64_t max_frame = 0;
    for (uint32_t k = 0; k < count; k++) { const uint64_t d = d_off[first + k + 1] - d_off[first + k]; max_frame = d > max_frame ? d : max_frame; }
    const bool long_frames = dsz >= (uint64_t)count * (4u * ZK_SEG_BYTES);
    const uint32_t seg_bytes = e->choice.seg_kib ? (uint32_t)e->choice.seg_kib << 10 : long_frames ? ZK_SEG_BYTES : 4096u;
    const uint64_t max_segs64 = 2 * ((max_frame + seg_bytes - 1) / seg_bytes) + 1;
    bool short_frames = !long_frames && max_frame >= 32768 && max_frame <= ZK_SEG_BYTES && (uint64_t)count * max_segs64 <= 256;
    if (short_frames && e->choice.exec_seg != 2) {
        // ... if they HAVE blocks to deal out: the reference's own 64 KiB frames are one block (+ an empty last one), executed by one
        // workgroup either way -- the extra launches would only cost them ~25 us.  The host holds the frames' bytes (pinned, just
        // filled): the walk over a handful of block headers is the device's lane code, run here (zk_walk_frame is host + device)
        uint32_t nb = 0;
        for (uint32_t k = 0; k < count && short_frames; k++) {
            ZkFrameInfo fi;
            zk_walk_frame(h_comp, h_offs[k], h_offs[k + 1], h_offs[count + 1 + k + 1] - h_offs[count + 1 + k], k, nullptr, nullptr, fi);
            if (fi.status != ZK_OK) short_frames = false;          // (the device gives the verdict)
            nb += fi.n_blocks;
        }
        short_frames = short_frames && nb >= 6u * count;
    }
    const bool seg = !d_prefix && e->choice.exec_seg != 1 && max_segs64 <= 65535 && (e->choice.exec_seg == 2 || long_frames || short_frames);
    if (seg) {
        zk_engine::DecCtx &x = e->dctx[0];
        ZkSegScratch sgs{};
        sgs.seg_bytes = seg_bytes;
        sgs.max_segs = (uint32_t)max_segs64;
        const uint64_t nsg = (uint64_t)count * sgs.max_segs;
        if ((rc = zk_devbuf_reserve(e, x.seg_tab, (size_t)nsg * sizeof(ZkSeg)))) return rc;
        if ((rc = zk_devbuf_reserve(e, x.seg_cnt, (size_t)(nsg + count) * sizeof(uint32_t)))) return rc;
        if ((rc = zk_devbuf_reserve(e, x.seg_holes, (size_t)((dsz >> 2) + 16 * nsg + 16) * sizeof(ZkHole)))) return rc;
        if ((rc = zk_devbuf_reserve(e, x.seg_tiles, (size_t)((dsz >> 10) + 2 * (uint64_t)block_cap + 8 * nsg + 16) * sizeof(uint32_t)))) return rc;
        sgs.segs = (ZkSeg *)x.seg_tab.p; sgs.nsegs = (uint32_t *)x.seg_cnt.p; sgs.segn = sgs.nsegs + count;
        sgs.holes = (ZkHole *)x.seg_holes.p; sgs.tilecnt = (uint32_t *)x.seg_tiles.p;
        zk_launch_exec_seg(st, comp, d_offs + count + 1, 0, count, nullptr, nullptr, blocks, (const ZkFrameBase *)c.bases.p, infos, (const ZkSeqP *)c.seqs.p,
                           (const uint8_t *)c.lit.p, (uint8_t *)s.d_out.p, sgs, e->choice, false, prog);
    } else
    zk_launch_exec(st, comp, d_offs + count + 1, 0, count, nullptr, nullptr, blocks, (const ZkFrameBase *)c.bases.p, infos, (const ZkSeqP *)c.seqs.p,
                   (const uint8_t *)c.lit.p, (uint8_t *)s.d_out.p, (const uint8_t *)d_prefix, d_prefix ? prefix_len : 0, e->choice, false, prog);
    if (prog) {
        zk_engine::DecCtx &x = e->dctx[0];
        zk_launch_xxh64_follow(x.aux, (const uint8_t *)s.d_out.p, d_offs + count + 1, 0, count, infos, prog);
        ZK_HIP(hipEventRecord(x.ev_join, x.aux));
        ZK_HIP(hipStreamWaitEvent(st, x.ev_join, 0));
        zk_launch_xxh64(st, (const uint8_t *)s.d_out.p, d_offs + count + 1, 0, count, infos, nullptr, e->choice, prog);
    } else if (verify) zk_launch_xxh64(st, (const uint8_t *)s.d_out.p, d_offs + count + 1, 0, count, infos, nullptr, e->choice);
    zk_launch_small_publish(st, infos, d_offs, count, (const uint8_t *)s.d_out.p, dsz ? h_out : nullptr, (int32_t *)s.d_st.p, h_status, words, hp->pin_flag, gen);
    // completion: the last workgroup of the publish kernel writes the generation into pinned memory
    volatile uint32_t *flag = hp->pin_flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *flag != gen; spins++) {
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
            ZK_HIP(hipStreamSynchronize(st));              // something is slow or wrong: let the runtime tell
            ZK_HIP(hipGetLastError());
            if (*flag != gen) { e->last_err = "small decode path: completion word never arrived"; return ZK_ERR_HIP; }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (h_status[0] == (int32_t)0xFFFFFFFF) { *fallback = true; return 0; }
    if (!dst_pinned && dsz) memcpy(dst, h_out, (size_t)dsz);
    uint32_t ok = count;
    for (uint32_t i = 0; i < count; i++) if (h_status[i] != 0) { ok = i; break; }
    if (frame_status) memcpy(frame_status, h_status, (size_t)count * 4);
    if (n_ok) *n_ok = ok;
    return ok == count ? 0 : -(int)h_status[ok];
}

int zk_host_decode(zk_engine *e, const zk_host_src &src, const uint64_t *c_off, const uint64_t *d_off, uint32_t first, uint32_t count,
                   const void *d_prefix, uint64_t prefix_len, uint8_t *dst, uint64_t dst_cap, int verify, int32_t *frame_status,
                   uint32_t *n_ok)
{
    if (n_ok) *n_ok = 0;
    if (!e || (count && (!c_off || !d_off || (!src.mem && !src.read)))) return ZK_ERR_ARGUMENT;
    if (count == 0) return 0;
    if (e->slot_busy[0] || e->slot_busy[1]) return ZK_ERR_ARGUMENT;     // submitted batches own the decode contexts: zk_decode_wait first
    ZK_HIP(hipSetDevice(e->device));
    const uint64_t c_lo = c_off[first], d_lo = d_off[first], d_hi = d_off[first + count];
    // the offset arrays are the caller's: every entry is checked, not only the ends (prefix sums of a SeekTable pass trivially)
    for (uint32_t i = 0; i < count; i++)
        if (c_off[first + i + 1] < c_off[first + i] || d_off[first + i + 1] < d_off[first + i]) return ZK_ERR_ARGUMENT;
    if (d_hi - d_lo > dst_cap) return -(int)ZK_E_DST_TOO_SMALL;
    if (d_hi > d_lo && !dst) return ZK_ERR_ARGUMENT;
    zk_hostpipe *hp = nullptr;
    int rc;
    if ((rc = zk_hostpipe_get(e, &hp))) return rc;

    // ---- chunk list
    const uint64_t total_d = d_hi - d_lo;
    uint64_t target = total_d / 8;
    if (target < (16ull << 20)) target = 16ull << 20;
    if (target > (256ull << 20)) target = 256ull << 20;
    if (hp->chunk_target) target = hp->chunk_target;
    const int nctx = hp->nctx, nslots = nctx + 1;
    for (int k = 0; k < nctx; k++) if ((rc = zk_dec_ctx_ready(e, k))) return rc;
    std::vector<ZkChunk> chunks;
    for (uint32_t f = 0; f < count;) {
        uint32_t g = f + 1;
        while (g < count && d_off[first + g + 1] - d_off[first + f] <= target && g - f < (1u << 20)) g++;
        chunks.push_back({f, g, c_off[first + f], c_off[first + g], d_off[first + f], d_off[first + g]});
        f = g;
    }
    const size_t nchunks = chunks.size();
    const bool src_pinned = src.mem && zk_is_pinned(src.mem + c_lo);
    const bool dst_pinned = total_d == 0 || zk_is_pinned(dst);
    uint64_t max_c = 0;
    for (auto &ck : chunks) if (ck.c1 - ck.c0 > max_c) max_c = ck.c1 - ck.c0;
    // a small request (a seek) is staged through one small pinned buffer by this thread: no rings, no hand-over
    const bool small = nchunks == 1 && total_d <= (4u << 20) && max_c <= (4u << 20);
    if (small && count <= 64 && !e->profiling && e->choice.small_path != 1) {
        bool fallback = false;
        rc = zk_decode_small(e, hp, src, c_off, d_off, first, count, d_prefix, prefix_len, dst, dst_pinned, verify, frame_status, n_ok, &fallback);
        if (!fallback) return rc;
    }
    if (!small && !src.mem) { if ((rc = zk_hostpipe_rings(e, hp))) return rc; }     // a pull source is staged through the ring
    // ---- pinned meta: rebased offsets of every chunk + the status words of every frame (every allocation that can fail comes
    // before the registration windows start: their pool tasks point at this frame's objects)
    const size_t off_bytes = ((size_t)(count + nchunks) * 16 + 63) & ~(size_t)63;
    if ((rc = zk_pin_grow(e, hp->pin_meta, hp->pin_meta_cap, off_bytes + (size_t)count * 4 + 64))) return rc;
    uint64_t *pin_offs = (uint64_t *)hp->pin_meta;
    int32_t *pin_status = (int32_t *)(hp->pin_meta + off_bytes);
    if (small) { if ((rc = zk_pin_grow(e, hp->pin_small, hp->pin_small_cap, (size_t)max_c + (size_t)total_d + 256))) return rc; }
    // caller memory that is not pinned already is pinned on the fly, unit by unit, ahead of the copies (ZkRegWindow)
    ZkRegWindow wsrc, wdst;
    if (!small && src.mem && !src_pinned) wsrc.start(hp->pool, src.mem + c_lo, (size_t)(c_off[first + count] - c_lo));
    if (!small && !dst_pinned) wdst.start(hp->pool, dst, (size_t)total_d);

    int fail = 0;                                           // first pipeline-level failure (HIP error, source ended, ...)
    size_t offs_at = 0;
    std::atomic<int> copy_fail{0};

    // Queues: one upload queue, one download queue, one compute queue per decode context (the contexts' second queues
    // stay unused here: whole chunks overlap instead of huf || fse).  With two contexts that is four streams -- the
    // runtime multiplexes streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES), and a copy queue that shares a
    // hardware queue with a compute queue serialises behind its kernels (measured: 135 instead of 87 ms for 4 GiB).
    auto prep = [&](size_t i) -> int {                      // stage + upload the compressed bytes and offsets of chunk i
        const ZkChunk &ck = chunks[i];
        zk_hostpipe::Slot &s = hp->slot[i % nslots];
        const uint32_t nf = ck.f1 - ck.f0;
        const uint64_t csz = ck.c1 - ck.c0;
        int r;
        if ((r = zk_devbuf_reserve(e, s.d_in, (size_t)csz + 64))) return r;
        if ((r = zk_devbuf_reserve(e, s.d_out, (size_t)(ck.d1 - ck.d0) + 64))) return r;
        if ((r = zk_devbuf_reserve(e, s.d_off, (size_t)(nf + 1) * 16))) return r;
        if ((r = zk_devbuf_reserve(e, s.d_st, (size_t)nf * 4 + 16))) return r;
        // the chunk's HBM buffers were last used by chunk i - nslots: its D2H must have run
        if (s.out_pending) ZK_HIP(hipStreamWaitEvent(hp->s_h2d, s.ev_out, 0));
        uint64_t *o = pin_offs + offs_at;
        offs_at += (size_t)(nf + 1) * 2;
        for (uint32_t k = 0; k <= nf; k++) { o[k] = c_off[first + ck.f0 + k] - ck.c0; o[nf + 1 + k] = d_off[first + ck.f0 + k] - ck.d0; }
        ZK_HIP(hipMemcpyAsync(s.d_off.p, o, (size_t)(nf + 1) * 16, hipMemcpyHostToDevice, hp->s_h2d));
        if (src_pinned) {
            if (csz) ZK_HIP(hipMemcpyAsync(s.d_in.p, src.mem + ck.c0, csz, hipMemcpyHostToDevice, hp->s_h2d));
        } else if (small) {
            if ((r = zk_src_fill(e, hp, src, ck.c0, hp->pin_small, (size_t)csz))) return r;
            if (csz) ZK_HIP(hipMemcpyAsync(s.d_in.p, hp->pin_small, csz, hipMemcpyHostToDevice, hp->s_h2d));
        } else {
            for (uint64_t at = 0; at < csz;) {
                if (src.mem && wsrc.ok(src.mem + ck.c0 + at)) {             // straight from the caller's (now pinned) pages
                    const size_t sp = wsrc.span(src.mem + ck.c0 + at), n = (size_t)(csz - at < sp ? csz - at : sp);
                    ZK_HIP(hipMemcpyAsync((uint8_t *)s.d_in.p + at, src.mem + ck.c0 + at, n, hipMemcpyHostToDevice, hp->s_h2d));
                    at += n;
                    continue;
                }
                if ((r = zk_hostpipe_rings(e, hp))) return r;
                const size_t n = (size_t)(csz - at < zk_hostpipe::PIECE ? csz - at : zk_hostpipe::PIECE);
                const int k = hp->ring_in.acquire();
                ZkRing::Piece &pc = hp->ring_in.pc[k];
                if ((r = zk_src_fill(e, hp, src, ck.c0 + at, pc.p, n))) return r;
                ZK_HIP(hipMemcpyAsync((uint8_t *)s.d_in.p + at, pc.p, n, hipMemcpyHostToDevice, hp->s_h2d));
                ZK_HIP(hipEventRecord(pc.ev, hp->s_h2d));
                pc.ev_pending = true;
                at += n;
            }
        }
        ZK_HIP(hipMemsetAsync((uint8_t *)s.d_in.p + csz, 0, 16, hp->s_h2d));     // the readers may touch ZK_COMP_PADDING bytes past the end
        ZK_HIP(hipEventRecord(s.ev_in, hp->s_h2d));
        return 0;
    };

    auto run = [&](size_t i) -> int {                       // decode chunk i and queue its way back
        const ZkChunk &ck = chunks[i];
        zk_hostpipe::Slot &s = hp->slot[i % nslots];
        const uint32_t nf = ck.f1 - ck.f0;
        const uint64_t dsz = ck.d1 - ck.d0;
        zk_dec_ctx c = zk_dec_context(e, (int)(i % nctx), nullptr);
        ZK_HIP(hipStreamWaitEvent(c.st, s.ev_in, 0));
        if (s.out_pending) ZK_HIP(hipStreamWaitEvent(c.st, s.ev_out, 0));
        const uint64_t *dc = (const uint64_t *)s.d_off.p, *dd = dc + nf + 1;
        zk_dec_args a{s.d_in.p, ck.c1 - ck.c0, dc, dd, 0, nf, nullptr, nullptr, s.d_out.p, dsz, verify, s.d_st.p, d_prefix, d_prefix ? prefix_len : 0};
        a.single_queue = nchunks > 1;                       // whole chunks overlap instead of huf || fse
        a.mark_exec = true;                                 // the bytes travel back while the checksum chains still run; the status words follow them
        const bool prof = e->profiling;
        if (nchunks > 1) e->profiling = false;              // per-kernel events describe one synchronous batch
        int r = zk_decode_enqueue(e, c, a);
        e->profiling = prof;
        if (r) return r;
        ZK_HIP(hipEventRecord(s.ev_dec, c.st));
        ZK_HIP(hipStreamWaitEvent(hp->s_d2h, c.ev_exec, 0));
        uint8_t *out = dst + (ck.d0 - d_lo);
        if (dst_pinned) {
            if (dsz) ZK_HIP(hipMemcpyAsync(out, s.d_out.p, dsz, hipMemcpyDeviceToHost, hp->s_d2h));
        } else if (small) {
            uint8_t *stage = hp->pin_small + ((max_c + 63) & ~(uint64_t)63);
            if (dsz) ZK_HIP(hipMemcpyAsync(stage, s.d_out.p, dsz, hipMemcpyDeviceToHost, hp->s_d2h));
        } else {
            for (uint64_t at = 0; at < dsz;) {
                if (wdst.ok(out + at)) {                                     // straight into the caller's (now pinned) pages
                    const size_t sp = wdst.span(out + at), n = (size_t)(dsz - at < sp ? dsz - at : sp);
                    ZK_HIP(hipMemcpyAsync(out + at, (const uint8_t *)s.d_out.p + at, n, hipMemcpyDeviceToHost, hp->s_d2h));
                    at += n;
                    continue;
                }
                if ((r = zk_hostpipe_rings(e, hp))) return r;
                const size_t n = (size_t)(dsz - at < zk_hostpipe::PIECE ? dsz - at : zk_hostpipe::PIECE);
                const int k = hp->ring_out.acquire();
                ZkRing::Piece &pc = hp->ring_out.pc[k];
                ZK_HIP(hipMemcpyAsync(pc.p, (const uint8_t *)s.d_out.p + at, n, hipMemcpyDeviceToHost, hp->s_d2h));
                ZK_HIP(hipEventRecord(pc.ev, hp->s_d2h));
                hp->ring_out.hold(k);
                uint8_t *to = out + at;
                ZkRing *ring = &hp->ring_out;
                ZkPool *pool = hp->pool;
                const size_t parts = n >= (8u << 20) && pool->size() >= 8 ? 4 : 1;
                // copy-out: the first worker waits for the DMA, then the piece is split among the workers
                pool->post([=, &copy_fail] {
                    if (hipEventSynchronize(ring->pc[k].ev) != hipSuccess) copy_fail.store(1);
                    const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
                    auto left = std::make_shared<std::atomic<int>>((int)parts);
                    for (size_t q = 1; q < parts; q++) {
                        const size_t a0 = q * per;
                        if (a0 >= n) { left->fetch_sub(1); continue; }
                        const size_t len = n - a0 < per ? n - a0 : per;
                        pool->post([=] { memcpy(to + a0, ring->pc[k].p + a0, len); if (left->fetch_sub(1) == 1) ring->release(k); });
                    }
                    memcpy(to, ring->pc[k].p, per < n ? per : n);
                    if (left->fetch_sub(1) == 1) ring->release(k);
                });
                at += n;
            }
        }
        ZK_HIP(hipStreamWaitEvent(hp->s_d2h, s.ev_dec, 0));
        ZK_HIP(hipMemcpyAsync(pin_status + ck.f0, s.d_st.p, (size_t)nf * 4, hipMemcpyDeviceToHost, hp->s_d2h));
        ZK_HIP(hipEventRecord(s.ev_out, hp->s_d2h));
        s.out_pending = true;
        return 0;
    };

    if ((fail = prep(0)) == 0) {
        for (size_t i = 0; i < nchunks; i++) {
            if (i + 1 < nchunks && (fail = prep(i + 1))) break;       // the next chunk's upload is queued before this one's decode blocks the thread
            if ((fail = run(i))) break;
        }
    }
    // ---- drain
    hipError_t s1 = hipStreamSynchronize(hp->s_h2d), s2 = hipSuccess, s3 = hipSuccess;
    for (int k = 0; k < nctx; k++) { const hipError_t r = hipStreamSynchronize(e->dctx[k].st); if (r != hipSuccess) s2 = r; }
    const hipError_t s4 = hipStreamSynchronize(hp->s_d2h);
    for (auto &s : hp->slot) s.out_pending = false;
    if (hp->rings_ready) hp->ring_out.wait_all_released();
    wsrc.finish(); wdst.finish();
    if (fail) return fail;
    if (s1 != hipSuccess || s2 != hipSuccess || s3 != hipSuccess || s4 != hipSuccess || copy_fail.load()) {
        e->last_err = "host decode pipeline: a queue failed"; return ZK_ERR_HIP;
    }
    ZK_HIP(hipGetLastError());
    if (nchunks == 1) zk_profile_collect(e);
    if (small && !dst_pinned && total_d) memcpy(dst, hp->pin_small + ((max_c + 63) & ~(uint64_t)63), (size_t)total_d);
    uint32_t ok = count;
    for (uint32_t i = 0; i < count; i++) if (pin_status[i] != 0) { ok = i; break; }
    if (frame_status) memcpy(frame_status, pin_status, (size_t)count * 4);
    if (n_ok) *n_ok = ok;
    return ok == count ? 0 : -(int)pin_status[ok];
}

extern "C" int zk_decode_frames(zk_engine *e, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off,
                                const uint64_t *d_off, uint32_t first, uint32_t count, uint8_t *dst, uint64_t dst_cap,
                                int verify, int32_t *frame_status)
{
    return zk_decode_frames_prefix(e, comp, comp_size, c_off, d_off, first, count, nullptr, 0, dst, dst_cap, verify, frame_status);
}

extern "C" int zk_decode_frames_prefix(zk_engine *e, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off,
                                       const uint64_t *d_off, uint32_t first, uint32_t count, const uint8_t *prefix, uint64_t prefix_len,
                                       uint8_t *dst, uint64_t dst_cap, int verify, int32_t *frame_status)
{
    if (!e || (count && (!comp || !c_off || !d_off))) return ZK_ERR_ARGUMENT;
    if (count == 0) return 0;
    for (uint32_t i = 0; i <= count; i++) if (c_off[first + i] > comp_size) return ZK_ERR_ARGUMENT;
    const void *d_prefix = nullptr;
    if (!prefix) prefix_len = 0;
    // the prefix is uploaded on every call: nothing is cached by address (a caller may decode against base A, then against
    // an equal-length base B in the same buffer)
    int rc = zk_engine_stage_prefix(e, nullptr, prefix, prefix_len, true, &d_prefix);
    if (rc) return rc;
    zk_host_src src;
    src.mem = comp;
    return zk_host_decode(e, src, c_off, d_off, first, count, d_prefix, prefix_len, dst, dst_cap, verify, frame_status, nullptr);
}

// ---------------------------------------------------------------------------------------------- encode
int zk_host_encode(zk_engine *e, const uint8_t *src, uint64_t n, uint32_t frame_size, int level, int checksum, const void *d_prefix,
                   uint64_t prefix_len, zk_host_sink sink, void *user)
{
    if (!e || frame_size == 0 || frame_size > ZK_SEEKABLE_MAX_FRAME_SIZE || !sink || (n && !src)) return ZK_ERR_ARGUMENT;
    const uint64_t nf64 = n == 0 ? 1 : (n + frame_size - 1) / frame_size;
    if (nf64 > ZK_SEEKABLE_MAX_FRAMES) return ZK_ERR_FRAME_INDEX_TOO_LARGE;
    ZK_HIP(hipSetDevice(e->device));
    zk_hostpipe *hp = nullptr;
    int rc;
    if ((rc = zk_hostpipe_get(e, &hp))) return rc;

    // chunks: at least 256 MiB and 128 frames, at most 2 GiB of input.  (The matcher works on 256 KiB segments, a workgroup
    // each: 256 MiB already are four rounds of the whole chip; round 2's one workgroup per frame needed 768 frames = 1.5 GiB
    // per chunk, which had to be uploaded before the first kernel ran.)
    uint64_t per = 128;
    if (per * frame_size < (256ull << 20)) per = ((256ull << 20) + frame_size - 1) / frame_size;
    if (per * frame_size > (2048ull << 20)) per = (2048ull << 20) / frame_size;
    if (per < 1) per = 1;
    if (nf64 <= per + per / 4) per = nf64;                   // no short tail chunk
    const size_t nchunks = (size_t)((nf64 + per - 1) / per);
    const bool src_pinned = n == 0 || zk_is_pinned(src);
    const bool small = nchunks == 1 && n <= (4u << 20);
    if (!small) { if ((rc = zk_hostpipe_rings(e, hp))) return rc; }                 // the compressed bytes travel back through the ring
    const uint64_t max_in = per * frame_size < n ? per * frame_size : n;
    const uint64_t max_bound = zk_compress_bound(max_in, frame_size);
    if (small) { if ((rc = zk_pin_grow(e, hp->pin_small, hp->pin_small_cap, (size_t)max_in + (size_t)max_bound + 256))) return rc; }
    if ((rc = zk_pin_grow(e, hp->pin_meta, hp->pin_meta_cap, (size_t)per * 8 * 2 + 64))) return rc;     // (c, d) sizes of two chunks
    ZkRegWindow wsrc;                                       // (after every allocation that can fail: its pool tasks point at this frame)
    if (!small && !src_pinned) wsrc.start(hp->pool, src, (size_t)n);
    hipStream_t st = e->stream;

    auto chunk_range = [&](size_t i, uint64_t &f0, uint64_t &nf, uint64_t &b0, uint64_t &bn) {
        f0 = (uint64_t)i * per; nf = nf64 - f0 < per ? nf64 - f0 : per;
        b0 = f0 * frame_size; bn = n - b0 < nf * frame_size ? n - b0 : nf * frame_size;
    };
    auto prep = [&](size_t i) -> int {
        uint64_t f0, nf, b0, bn; chunk_range(i, f0, nf, b0, bn);
        zk_hostpipe::ESlot &s = hp->es[i & 1];
        int r;
        if ((r = zk_devbuf_reserve(e, s.d_src, (size_t)bn + 64))) return r;      // (d_dst / d_sizes of the slot may still be on their way down: enqueue() sizes them)
        if (src_pinned) { if (bn) ZK_HIP(hipMemcpyAsync(s.d_src.p, src + b0, bn, hipMemcpyHostToDevice, hp->s_h2d)); }
        else if (small) { memcpy(hp->pin_small, src + b0, (size_t)bn); if (bn) ZK_HIP(hipMemcpyAsync(s.d_src.p, hp->pin_small, bn, hipMemcpyHostToDevice, hp->s_h2d)); }
        else for (uint64_t at = 0; at < bn;) {
            if (wsrc.ok(src + b0 + at)) {                                   // straight from the caller's (now pinned) pages
                const size_t sp = wsrc.span(src + b0 + at), len = (size_t)(bn - at < sp ? bn - at : sp);
                ZK_HIP(hipMemcpyAsync((uint8_t *)s.d_src.p + at, src + b0 + at, len, hipMemcpyHostToDevice, hp->s_h2d));
                at += len;
                continue;
            }
            const size_t len = (size_t)(bn - at < zk_hostpipe::PIECE ? bn - at : zk_hostpipe::PIECE);
            const int k = hp->ring_in.acquire();
            ZkRing::Piece &pc = hp->ring_in.pc[k];
            hp->pool->copy(pc.p, src + b0 + at, len);
            ZK_HIP(hipMemcpyAsync((uint8_t *)s.d_src.p + at, pc.p, len, hipMemcpyHostToDevice, hp->s_h2d));
            ZK_HIP(hipEventRecord(pc.ev, hp->s_h2d));
            pc.ev_pending = true;
            at += len;
        }
        ZK_HIP(hipEventRecord(s.ev_in, hp->s_h2d));
        return 0;
    };
    auto enqueue = [&](size_t i) -> int {
        uint64_t f0, nf, b0, bn; chunk_range(i, f0, nf, b0, bn);
        zk_hostpipe::ESlot &s = hp->es[i & 1];
        int r0;
        if ((r0 = zk_devbuf_reserve(e, s.d_dst, (size_t)zk_compress_bound(bn, frame_size) + 64))) return r0;      // the chunk two back has been handed to the sink by now
        if ((r0 = zk_devbuf_reserve(e, s.d_sizes, (size_t)nf * 8 + 64))) return r0;
        ZK_HIP(hipStreamWaitEvent(st, s.ev_in, 0));
        if (s.out_pending) ZK_HIP(hipStreamWaitEvent(st, s.ev_out, 0));       // the chunk two back has left d_dst
        uint32_t *dc = (uint32_t *)s.d_sizes.p, *dd = dc + nf;
        zk_enc_args a{s.d_src.p, bn, frame_size, level, checksum, d_prefix, d_prefix ? prefix_len : 0, s.d_dst.p,
                      zk_compress_bound(bn, frame_size), dc, dd};
        uint32_t nfo = 0;
        const bool prof = e->profiling;
        if (nchunks > 1) e->profiling = false;
        int r = zk_encode_enqueue(e, a, st, &nfo);
        e->profiling = prof;
        if (r) return r;
        ZK_HIP(hipEventRecord(s.ev_enc, st));
        return 0;
    };
    // wait for chunk i, bring its bytes and seek entries back and hand them to the sink (in order, on this thread)
    auto finish = [&](size_t i, bool more) -> int {
        uint64_t f0, nf, b0, bn; chunk_range(i, f0, nf, b0, bn);
        zk_hostpipe::ESlot &s = hp->es[i & 1];
        ZK_HIP(hipEventSynchronize(s.ev_enc));
        const uint64_t total = e->h_words[ZK_HW_ENC_TOTAL];
        if (nchunks == 1) zk_profile_collect(e);
        int r = 0;
        if (more && (r = enqueue(i + 1))) return r;          // the GPU goes on with the next chunk while this one travels
        // ... and the chunk after it starts its way up now: its source slot is the one this chunk's kernels have just left
        // (issued behind this chunk's download and sink, its upload used to arrive after the GPU had run dry)
        if (i + 2 < nchunks && (r = prep(i + 2))) return r;
        uint32_t *sizes = (uint32_t *)hp->pin_meta + (i & 1) * (size_t)per * 2;
        ZK_HIP(hipMemcpyAsync(sizes, s.d_sizes.p, (size_t)nf * 8, hipMemcpyDeviceToHost, hp->s_d2h));
        if (small) {
            uint8_t *stage = hp->pin_small + ((max_in + 63) & ~(uint64_t)63);
            ZK_HIP(hipMemcpyAsync(stage, s.d_dst.p, total, hipMemcpyDeviceToHost, hp->s_d2h));
            ZK_HIP(hipStreamSynchronize(hp->s_d2h));
            return sink(user, stage, total, sizes, sizes + nf, (uint32_t)nf) ? ZK_ERR_IO : 0;
        }
        // pieces are requested a few ahead of the one the sink is working on
        std::deque<std::pair<int, size_t>> inflight;
        uint64_t at = 0;
        while (at < total || !inflight.empty()) {
            while (at < total && inflight.size() < 4) {
                const size_t len = (size_t)(total - at < zk_hostpipe::PIECE ? total - at : zk_hostpipe::PIECE);
                const int k = hp->ring_out.acquire();
                ZkRing::Piece &pc = hp->ring_out.pc[k];
                ZK_HIP(hipMemcpyAsync(pc.p, (const uint8_t *)s.d_dst.p + at, len, hipMemcpyDeviceToHost, hp->s_d2h));
                ZK_HIP(hipEventRecord(pc.ev, hp->s_d2h));
                hp->ring_out.hold(k);
                inflight.push_back({k, len});
                at += len;
                if (at >= total) { ZK_HIP(hipEventRecord(s.ev_out, hp->s_d2h)); s.out_pending = true; }
            }
            const auto pr = inflight.front();
            inflight.pop_front();
            hipError_t he = hipEventSynchronize(hp->ring_out.pc[pr.first].ev);
            if (he == hipSuccess && !r && sink(user, hp->ring_out.pc[pr.first].p, pr.second, nullptr, nullptr, 0)) r = ZK_ERR_IO;
            hp->ring_out.release(pr.first);
            if (he != hipSuccess) { e->last_err = "host encode pipeline: D2H failed"; r = ZK_ERR_HIP; }
        }
        if (total == 0) { ZK_HIP(hipEventRecord(s.ev_out, hp->s_d2h)); s.out_pending = true; }
        ZK_HIP(hipStreamSynchronize(hp->s_d2h));             // the sizes
        if (!r && sink(user, nullptr, 0, sizes, sizes + nf, (uint32_t)nf)) r = ZK_ERR_IO;
        return r;
    };

    int fail = prep(0);
    if (!fail) fail = enqueue(0);
    if (!fail && nchunks > 1) fail = prep(1);                // upload of the next chunk runs beside this chunk's kernels
    for (size_t i = 0; i < nchunks && !fail; i++) fail = finish(i, i + 1 < nchunks);
    (void)hipStreamSynchronize(hp->s_h2d); (void)hipStreamSynchronize(st); (void)hipStreamSynchronize(hp->s_d2h);
    wsrc.finish();
    for (auto &s : hp->es) s.out_pending = false;
    if (fail) return fail;
    ZK_HIP(hipGetLastError());
    return 0;
}

extern "C" int zk_encode_frames(zk_engine *e, const uint8_t *src, uint64_t n, uint32_t frame_size, int level, int checksum,
                                uint8_t *dst, uint64_t dst_cap, uint32_t *c_sizes, uint32_t *d_sizes, uint32_t frames_cap,
                                uint32_t *n_frames_out, uint64_t *written_out)
{
    return zk_encode_frames_prefix(e, src, n, frame_size, level, checksum, nullptr, 0, dst, dst_cap, c_sizes, d_sizes, frames_cap,
                                   n_frames_out, written_out);
}

namespace {
struct ZkBufSink {
    zk_engine *e; uint8_t *dst; uint64_t cap, pos; uint32_t *c, *d; uint32_t nf; bool overflow;
};
}
static int zk_buf_sink(void *user, const uint8_t *data, uint64_t n, const uint32_t *c_sizes, const uint32_t *d_sizes, uint32_t n_frames)
{
    ZkBufSink *s = (ZkBufSink *)user;
    if (n) {
        if (s->pos + n > s->cap) { s->overflow = true; return 1; }
        zk_host_copy(s->e, s->dst + s->pos, data, (size_t)n);
        s->pos += n;
    }
    if (n_frames) {
        if (s->c) memcpy(s->c + s->nf, c_sizes, (size_t)n_frames * 4);
        if (s->d) memcpy(s->d + s->nf, d_sizes, (size_t)n_frames * 4);
        s->nf += n_frames;
    }
    return 0;
}

extern "C" int zk_encode_frames_prefix(zk_engine *e, const uint8_t *src, uint64_t n, uint32_t frame_size, int level, int checksum,
                                       const uint8_t *prefix, uint64_t prefix_len, uint8_t *dst, uint64_t dst_cap, uint32_t *c_sizes,
                                       uint32_t *d_sizes, uint32_t frames_cap, uint32_t *n_frames_out, uint64_t *written_out)
{
    if (!e || frame_size == 0 || !dst || (n && !src)) return ZK_ERR_ARGUMENT;
    const uint64_t nf = n == 0 ? 1 : (n + frame_size - 1) / frame_size;
    if (nf > ZK_SEEKABLE_MAX_FRAMES) return ZK_ERR_FRAME_INDEX_TOO_LARGE;
    if ((c_sizes || d_sizes) && frames_cap < nf) return ZK_ERR_ARGUMENT;
    // only the tail the matcher can reach is staged
    const void *d_prefix = nullptr;
    const uint64_t tail = prefix ? zke_ldm_usable(prefix_len) : 0;      // what the matcher can reach: its ring's window, and beyond it the long-distance table's span
    int rc = zk_engine_stage_prefix(e, nullptr, tail ? prefix + (prefix_len - tail) : nullptr, tail, true, &d_prefix);
    if (rc) return rc;
    ZkBufSink s{e, dst, dst_cap, 0, c_sizes, d_sizes, 0, false};
    rc = zk_host_encode(e, src, n, frame_size, level, checksum, d_prefix, tail, zk_buf_sink, &s);
    if (s.overflow) return -(int)ZK_E_DST_TOO_SMALL;
    if (rc) return rc;
    if (n_frames_out) *n_frames_out = s.nf;
    if (written_out) *written_out = s.pos;
    return 0;
}

// ---------------------------------------------------------------------------------------------- XXH64
extern "C" int zk_xxh64_frames(zk_engine *e, const uint8_t *data, const uint64_t *off, uint32_t count, uint64_t *out)
{
    if (!e || (count && (!off || !out))) return ZK_ERR_ARGUMENT;
    if (count == 0) return 0;
    ZK_HIP(hipSetDevice(e->device));
    const uint64_t lo = off[0], hi = off[count];
    if (hi < lo || (hi > lo && !data)) return ZK_ERR_ARGUMENT;
    for (uint32_t i = 0; i < count; i++) if (off[i + 1] < off[i]) return ZK_ERR_ARGUMENT;
    std::vector<uint64_t> offs((size_t)count + 1);
    for (uint32_t i = 0; i <= count; i++) offs[i] = off[i] - lo;
    int rc;
    if ((rc = zk_devbuf_reserve(e, e->st_dst, (size_t)(hi - lo) + 64))) return rc;
    if ((rc = zk_devbuf_reserve(e, e->st_off, offs.size() * 8))) return rc;
    if ((rc = zk_devbuf_reserve(e, e->st_misc, (size_t)count * 8))) return rc;
    hipStream_t st = e->stream;
    if (hi > lo) ZK_HIP(hipMemcpyAsync(e->st_dst.p, data + lo, hi - lo, hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(e->st_off.p, offs.data(), offs.size() * 8, hipMemcpyHostToDevice, st));
    ZK_HIP(hipStreamSynchronize(st));
    rc = zk_xxh64_frames_dev(e, e->st_dst.p, e->st_off.p, count, e->st_misc.p, st);
    if (rc) return rc;
    ZK_HIP(hipMemcpy(out, e->st_misc.p, (size_t)count * 8, hipMemcpyDeviceToHost));
    return 0;
}
