// hip_wg_emu.h -- TEST HARNESS (never shipped): runs ONE workgroup of a HIP kernel on the CPU, a ucontext fiber per lane.
// The kernel source is compiled unchanged by g++; this header supplies what the device compiler would:
//   __shared__             -> static (one workgroup at a time)
//   __syncthreads / LDS barrier -> every live lane of the workgroup parks until all arrived
//   wave collectives (__ballot, readlane, ds_bpermute, update_dpp row_shr:1, wave_barrier) -> the 64 lanes of a wave
//                             deposit their value and park until the wave is complete; they must be called by all 64
//                             lanes (wave-uniform control flow), as the kernels under test do
//   atomicCAS on LDS       -> plain (fibers are cooperative)
// Lanes of a wave do NOT run in lock step here: code that relies on the in-order LDS pipeline of one wave needs a
// __builtin_amdgcn_wave_barrier() (a scheduling fence on the device, a wave meeting point here).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <functional>
#include <vector>

struct EmuDim { uint32_t x, y, z; };
struct EmuWave { uint32_t vals[64], res[64]; uint32_t count; uint64_t gen; };
struct EmuLane { ucontext_t ctx; uint32_t tid; bool done; };
struct EmuWG {
    ucontext_t sched;
    std::vector<EmuLane> lanes;
    std::vector<EmuWave> waves;
    std::vector<char> stacks;
    uint32_t nthreads = 0, cur = 0, block = 0, live = 0;
    uint32_t bar_count = 0; uint64_t bar_gen = 0;
    std::function<void()> body;
};
static EmuWG g_emu;

static inline void emu_yield() { swapcontext(&g_emu.lanes[g_emu.cur].ctx, &g_emu.sched); }
static inline EmuDim emu_tidx() { return EmuDim{g_emu.lanes[g_emu.cur].tid, 0, 0}; }
static inline EmuDim emu_bidx() { return EmuDim{g_emu.block, 0, 0}; }

static inline void emu_barrier()
{
    EmuWG &g = g_emu;
    const uint64_t gen = g.bar_gen;
    if (++g.bar_count == g.live) { g.bar_count = 0; g.bar_gen++; return; }
    while (g.bar_gen == gen) emu_yield();
}
// every lane of the wave deposits v; returns the 64 deposited values
static inline const uint32_t *emu_collect(uint32_t v)
{
    EmuWG &g = g_emu;
    const uint32_t tid = g.lanes[g.cur].tid;
    EmuWave &w = g.waves[tid >> 6];
    w.vals[tid & 63] = v;
    const uint64_t gen = w.gen;
    if (++w.count == 64) { w.count = 0; memcpy(w.res, w.vals, sizeof w.res); w.gen++; }
    else while (w.gen == gen) emu_yield();
    return w.res;
}
static inline uint64_t emu_ballot(bool p)
{
    const uint32_t *r = emu_collect(p ? 1u : 0u);
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= (uint64_t)(r[i] & 1) << i;
    return m;
}
static inline int emu_readlane(int v, int l) { return (int)emu_collect((uint32_t)v)[l & 63]; }
static inline int emu_bpermute(int addr, int v)
{
    const uint32_t mine = ((uint32_t)addr >> 2) & 63;        // captured before parking: res[] is shared
    return (int)emu_collect((uint32_t)v)[mine];
}
static inline int emu_dpp(int old, int src, int ctrl)
{
    const uint32_t lane = g_emu.lanes[g_emu.cur].tid & 63;
    const uint32_t *r = emu_collect((uint32_t)src);
    if (ctrl >= 0x111 && ctrl <= 0x11F) {                    // row_shr:n -- lane i of a row of 16 reads lane i - n of the same row
        const uint32_t n = (uint32_t)ctrl - 0x110;
        return (lane & 15) >= n ? (int)r[lane - n] : old;
    }
    if (ctrl >= 0x101 && ctrl <= 0x10F) {                    // row_shl:n
        const uint32_t n = (uint32_t)ctrl - 0x100;
        return (lane & 15) + n < 16 ? (int)r[lane + n] : old;
    }
    abort();
}
static inline void emu_wave_sync() { (void)emu_collect(0); }
static inline uint32_t emu_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (s & 3))); }
static inline uint32_t emu_atomic_cas(uint32_t *p, uint32_t cmp, uint32_t val) { const uint32_t old = *p; if (old == cmp) *p = val; return old; }

static inline uint32_t emu_atomic_min(uint32_t *p, uint32_t v) { const uint32_t old = *p; if (v < old) *p = v; return old; }

static void emu_entry()
{
    g_emu.body();
    EmuWG &g = g_emu;
    g.lanes[g.cur].done = true;
    g.live--;
    swapcontext(&g.lanes[g.cur].ctx, &g.sched);
}

// run `body` (the kernel call) as workgroup `block` of `nthreads` lanes
static void emu_run_workgroup(uint32_t nthreads, uint32_t block, std::function<void()> body)
{
    EmuWG &g = g_emu;
    constexpr size_t STACK = 128 << 10;
    g.nthreads = nthreads; g.block = block; g.live = nthreads; g.body = body;
    g.bar_count = 0;
    g.lanes.assign(nthreads, EmuLane());
    g.waves.assign((nthreads + 63) / 64, EmuWave());
    if (g.stacks.size() < STACK * nthreads) g.stacks.resize(STACK * nthreads);
    for (uint32_t t = 0; t < nthreads; t++) {
        EmuLane &l = g.lanes[t];
        l.tid = t; l.done = false;
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = g.stacks.data() + STACK * t;
        l.ctx.uc_stack.ss_size = STACK;
        l.ctx.uc_link = &g.sched;
        makecontext(&l.ctx, emu_entry, 0);
    }
    while (g.live) {
        for (uint32_t t = 0; t < nthreads; t++) {
            if (g.lanes[t].done) continue;
            g.cur = t;
            swapcontext(&g.sched, &g.lanes[t].ctx);
        }
    }
}

#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define threadIdx (emu_tidx())
#define blockIdx (emu_bidx())
#define __syncthreads() emu_barrier()
#define ZKE_LDS_BARRIER() emu_barrier()
#define __ballot(p) emu_ballot(p)
#define __builtin_amdgcn_readlane(v, l) emu_readlane((v), (l))
#define __builtin_amdgcn_ds_bpermute(a, v) emu_bpermute((a), (v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_dpp((old), (src), (ctrl))
#define __builtin_amdgcn_alignbyte(hi, lo, s) emu_alignbyte((hi), (lo), (s))
#define __builtin_amdgcn_wave_barrier() emu_wave_sync()
#define ZKE_WAVE_SYNC() emu_wave_sync()
#define __builtin_amdgcn_readfirstlane(v) (v)          /* only used on values that are uniform across the wave */
#define atomicCAS(p, c, v) emu_atomic_cas((p), (c), (v))
#define atomicMin(p, v) emu_atomic_min((p), (v))
#define __ffs(x) __builtin_ffs(x)
#define ZKE_FFBL(x) ((x) ? (uint32_t)__builtin_ctz(x) : 0xFFFFFFFFu)          /* v_ffbl_b32 */
#define ZKE_FFBH(x) ((x) ? (uint32_t)__builtin_clz(x) : 0xFFFFFFFFu)          /* v_ffbh_u32 */
#define ZKE_WALK(taken, f, nx) do { taken |= 1ull << f; f = (uint32_t)emu_readlane((int)(nx), (int)f); } while (f < 64)     /* zk_enc_match2.h: the walk's inner loop */
#define ZKE_KEEP(x) do { } while (0)
