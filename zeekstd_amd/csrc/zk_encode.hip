// zk_encode.hip -- gfx950 kernels of the batched seekable-zstd frame encoder.
//
// Replaces, for N frames at a time, what the reference does through ZSTD_compressStream2
// (lib/src/encode.rs:340-346 hot loop, 442-464 epilogue; libzstd 1.5.7 underneath) plus the seek-table
// bookkeeping of SeekTable::log_frame (lib/src/seek_table.rs:513-525): (c_size, d_size) per frame are
// produced on the device.  Compressed bytes are not pinned by the reference (only validity / round trip);
// the CPU twin of this algorithm is oracle/zstd_oracle_enc.c and the GPU output is byte-identical to it.
//
// Pipeline (one stream, everything resident in HBM):
//   zk_k_xxh64        (checksum_flag) XXH64 of every frame's input
//   zk_k_enc_match    (zk_enc_match.h) one workgroup (16 waves) per segment of <= 256 KiB of a frame; the segment's last
//                     64 KiB in an LDS ring, hash table, candidates and parse in LDS -> packed sequences + literals per block.
//                     Prefix mode: the matcher reads [prefix tail | frame] records (zk_k_enc_stage_hist)
//   zk_k_enc_entropy  one workgroup per 16 blocks: literal histograms, Huffman lengths (<= 11 bits) while the other
//                     waves precompute every sequence's codes and extra bits, then 64 literal streams (wave 0) and 16
//                     FSE sequence bitstreams with the predefined tables (wave 1) side by side -- branch-free bit
//                     writers --, block payload assembled in scratch one wave per block, raw / RLE fallbacks decided
//   zk_k_enc_sizes    per frame: compressed size = header + blocks (+ checksum)   -> seek-table entries
//   zk_k_scan64       exclusive scan of the frame sizes -> where each frame lands in the output stream
//   zk_k_enc_assemble one workgroup per frame: frame header, block headers, payload copies into the
//                     final contiguous stream
// HBM-bound integer/byte work; no MFMA.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "zk_device.h"
#include "zk_enc_device.h"
#include "zk_kernels.h"

// ------------------------------------------------------------------------------------------------ match + parse
#ifdef ZKE_CLOCKS
// experiments: shader-clock totals per phase of the match kernel, summed over lane 0 of every wave (tools/enc_clocks.py)
__device__ unsigned long long zke_dbg_clk[16];
extern "C" void zk_debug_enc_clocks(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zke_dbg_clk), z, sizeof z); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(zke_dbg_clk), 16 * sizeof(unsigned long long));
}
#define ZKE_CLK_BEGIN() unsigned long long clk_[16] = {0}, t_ = clock64()
#define ZKE_CLK(i) do { const unsigned long long now_ = clock64(); clk_[i] += now_ - t_; t_ = now_; } while (0)
#define ZKE_CLK_END() do { if ((threadIdx.x & 63) == 0) for (int i = 0; i < 16; i++) atomicAdd(&zke_dbg_clk[i], clk_[i]); } while (0)
#endif
#include "zk_enc_match.h"
#include "zk_enc_match2.h"
#ifdef ZKE_ENT_CLOCKS
// experiments: the same for the entropy kernel (per wave: slot = 4 * phase + wave; tools/ent_clocks.py)
__device__ unsigned long long zke_dbg_clk[48];
__device__ unsigned int zke_dbg_cu[8 * 16 * 16];      // workgroups resident per (XCC, SE, CU)
extern "C" void zk_debug_enc_clocks(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[48] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zke_dbg_clk), z, sizeof z); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(zke_dbg_clk), 48 * sizeof(unsigned long long));
}
#define ZKE_ECLK_BEGIN() unsigned long long eclk_[8] = {0}, et_ = clock64(), ec0_ = et_, ew0_ = wall_clock64(); uint32_t ecu_ = 0; \
    if (threadIdx.x == 0) { uint32_t hw_, xc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc_)); \
        ecu_ = ((xc_ & 7) << 8) | (((hw_ >> 13) & 7) << 4) | ((hw_ >> 8) & 15); const uint32_t was_ = atomicAdd(&zke_dbg_cu[ecu_], 1u); atomicAdd(&zke_dbg_clk[40 + (was_ < 7 ? was_ : 7)], 1ull); }
#define ZKE_ECLK(i) do { const unsigned long long now_ = clock64(); eclk_[i] += now_ - et_; et_ = now_; } while (0)
#define ZKE_ECLK_END() do { if ((threadIdx.x & 63) == 0) for (int i = 0; i < 8; i++) atomicAdd(&zke_dbg_clk[5 * i + (threadIdx.x >> 6)], eclk_[i]); if (threadIdx.x == 0) { atomicSub(&zke_dbg_cu[ecu_], 1u); atomicAdd(&zke_dbg_clk[35], clock64() - ec0_); atomicAdd(&zke_dbg_clk[36], wall_clock64() - ew0_); } } while (0)
#else
#define ZKE_ECLK_BEGIN() do { } while (0)
#define ZKE_ECLK(i) do { } while (0)
#define ZKE_ECLK_END() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------ entropy stage
// LSB-first bit writer of one lane.  (Until round 3 it stored its accumulator's 8 bytes straight to HBM at every flush: 16 lanes
// storing 8 bytes each at 16 places twice per sequence -- and 64 lanes once per 2-3 literals -- were what the CU's write path choked on.)
// The stream goes through LDS: the lane's bits go to its own staging area in whole 32-bit words (the accumulator's two words are
// stored at every flush(); the low one moves on once it is full), and drain() carries the complete 16-byte units to HBM.
// Why: 16 lanes storing 8 bytes each at 16 places twice per sequence were 32 write requests per step and workgroup -- the CU's
// write path was what every wave of the kernel waited for (without those stores: sequence writers 2.04 -> 1.32 M clocks, the
// histogram passes 0.56 -> 0.40, the literal writers 0.55 -> 0.36).  A drain is one request per 16 bytes.
// (ds_write_b64 at an address that is not a multiple of 8 does not write what it should -- tools/ubench/ldsun.hip -- hence words.)
// st: LDS byte address of the lane's staging area (a multiple of 16; ZKE_STAGE bytes for the sequence writers).
// LDS accesses by byte address (an integer made into an LDS pointer directly: through a generic pointer every access would start
// with the null check of the address space cast)
template <typename T> __device__ __forceinline__ T zke_lds_ld_at(uint32_t a) { return *(const volatile ZK_LDS_AS T *)a; }
template <typename T> __device__ __forceinline__ void zke_lds_st_at(uint32_t a, T v) { *(volatile ZK_LDS_AS T *)a = v; }
constexpr uint32_t ZKE_LIT_STAGE = 192;          // the literal writers': <= 4 rounds x 33 bytes + 15 + 8, + the 7 odd symbols' <= 10 before the first round
constexpr uint32_t ZKE_STAGE = 112;              // the sequence emitter's: <= 2 rounds x 42 bytes + 15 left by the last drain + the accumulator's 8; 28 words: the 16 lanes start in 16 different banks
typedef uint32_t zke_u32x4 __attribute__((ext_vector_type(4)));
struct ZkeBitsL {
    uint8_t *g; uint32_t cap, lim, done, st, wa, n, ovf; uint64_t acc;      // done: bytes in HBM; wa: LDS byte address of the accumulator's low word
    // the stream may have cap bytes; lim >= cap bytes at g may be written
    __device__ __forceinline__ void init(uint8_t *g_, uint32_t cap_, uint32_t lim_, uint32_t st_) { g = g_; cap = cap_; lim = lim_; done = 0; st = wa = st_; acc = 0; n = 0; ovf = 0; }
    // v: nb <= 32 bits (nothing above them); < 32 bits are waiting, left by flush()
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb) { acc |= (uint64_t)v << n; n += nb; }
    __device__ __forceinline__ void flush()
    {
        uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
        zke_lds_st_at<uint32_t>(wa, lo); zke_lds_st_at<uint32_t>((wa + 4), hi);
        const bool full = n >= 32;               // the low word is complete: the high one takes its place (two selects: a 64-bit shift costs more)
        lo = full ? hi : lo; hi = full ? 0u : hi;
        acc = lo | (uint64_t)hi << 32;
        wa += full ? 4u : 0u; n &= 31;
    }
    // whole 16-byte units to HBM, the rest moves to the front of the staging area
    __device__ __forceinline__ void drain()
    {
        const uint32_t units = (wa - st) >> 4;
        for (uint32_t u = 0; __ballot(u < units); u++)
            if (u < units) {
                const zke_u32x4 v = zke_lds_ld_at<zke_u32x4>((st + 16 * u));
                if (done + 16 * u + 16 <= lim) zk_glb_st<zke_u32x4>(g + done + 16 * u, v); else ovf = 1;
            }
        const zke_u32x4 a = zke_lds_ld_at<zke_u32x4>((st + 16 * units)), b = zke_lds_ld_at<zke_u32x4>((st + 16 * units + 16));
        zke_lds_st_at<zke_u32x4>(st, a); zke_lds_st_at<zke_u32x4>((st + 16), b);
        done += 16 * units; wa -= 16 * units;
    }
    // end mark, the last bytes; returns the stream's size (0: it did not fit into cap bytes)
    __device__ __forceinline__ uint32_t close()
    {
        flush(); put(1, 1); flush();
        drain();
        const uint32_t bytes = (wa - st) + ((n + 7) >> 3);                 // < 16 + 4
        const zke_u32x4 a = zke_lds_ld_at<zke_u32x4>(st), b = zke_lds_ld_at<zke_u32x4>((st + 16));
        if (done + 32 <= lim) { zk_glb_st<zke_u32x4>(g + done, a); zk_glb_st<zke_u32x4>(g + done + 16, b); }
        else for (uint32_t t = 0; t < bytes; t++) { if (done + t < lim) g[done + t] = zke_lds_ld_at<uint8_t>(st + t); else ovf = 1; }   // the region ends here: byte by byte
        const uint32_t total = done + bytes;
        return ovf || total > cap ? 0u : total;
    }
};

// ------------------------------------------------------------------------------------------------ the frame's FSE tables
// One workgroup per frame: code histograms of all its sequences (LDS atomics), then one lane per table normalises,
// writes the description and builds the compression table into the frame's ZkEncTables (HBM).  A table with fewer than
// two symbols, or a frame with fewer than ZKE_FSE_MIN_SEQ sequences, keeps the predefined one.
// It also rewrites every sequence into what its serial bit writer (zk_k_enc_entropy) needs -- everything that does not depend on
// the FSE states -- so that the writers can start with the kernel:
//   seqs[i]  <- extra bits of LL | ML | OF back to back (<= 16 + 15 + 26 with long-distance offsets), their count << 58
//   mpos[i]  <- LL code | ML code << 8 | OF code << 16
__global__ __launch_bounds__(1024) void zk_k_enc_fse_build(const ZkEncFrame *frames, const ZkEncBlock *blocks, uint64_t *seqs, uint32_t *mpos,
                                                          const ZkEncTables *predef, ZkEncTables *ftab, uint32_t min_seq)
{
    __shared__ uint32_t s_llv[36], s_mlv[56];              // base | extra bits << 24 (the same in every frame)
    __shared__ uint32_t h[3][64];
    __shared__ int16_t norm[3][64];
    __shared__ uint8_t sym[3][512];
    __shared__ int32_t cumul[3][66];
    __shared__ uint32_t s_nseq;
    const uint32_t tid = threadIdx.x;
    const ZkEncFrame fr = frames[blockIdx.x];
    ZkEncTables *T = &ftab[blockIdx.x];
    if (tid < 192) (&h[0][0])[tid] = 0;
    if (tid == 0) s_nseq = 0;
    if (tid < 36) s_llv[tid] = predef->ll_val[tid];
    if (tid >= 64 && tid < 64 + 56) s_mlv[tid - 64] = predef->ml_val[tid - 64];
    __syncthreads();
    uint32_t mine = 0;
    auto one = [&](uint64_t e, uint64_t *sq, uint32_t *cw, uint32_t k) {
        const uint32_t ll = (uint32_t)e & 0xFFFF, ml = (uint32_t)(e >> 16) & 0xFFFF, ob = (uint32_t)(e >> 32);
        const uint32_t llc = zke_ll_code(ll), mlc = zke_ml_code(ml - 3), ofc = zk_highbit(ob);
        atomicAdd(&h[0][llc], 1u); atomicAdd(&h[1][ofc], 1u); atomicAdd(&h[2][mlc], 1u);
        const uint32_t lv = s_llv[llc], mv = s_mlv[mlc];
        const uint32_t ln = lv >> 24, mn = mv >> 24;
        const uint64_t x = (uint64_t)(ll - (lv & 0xFFFFFF)) | ((uint64_t)(ml - (mv & 0xFFFFFF)) << ln) | ((uint64_t)(ob - (1u << ofc)) << (ln + mn));
        sq[k] = x | ((uint64_t)(ln + mn + ofc) << 58);
        cw[k] = llc | (mlc << 8) | (ofc << 16);
        mine++;
    };
    // A lane takes sequences tid, tid + 1024, tid + 2048, tid + 3072 of every block (a 32 KiB block of text has ~2300).  The next
    // block's four records are requested BEFORE this block's stores: loads and stores share one in-order counter, and a load issued
    // behind a store waits for the store's acknowledgement (64 blocks per frame, one such wait each, were most of this kernel).
    auto first4 = [&](uint32_t b, uint64_t e4[4]) {
        const ZkEncBlock &blk = blocks[fr.block_base + (b < fr.n_blocks ? b : 0)];
        const uint64_t *sq = seqs + blk.seq_base;
        const uint32_t nseq = b < fr.n_blocks ? blk.nseq : 0;
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = tid + u * 1024u; e4[u] = sq[k < nseq ? k : 0]; }
    };
    uint64_t e4[4];
    first4(0, e4);
    for (uint32_t b = 0; b < fr.n_blocks; b++) {
        const ZkEncBlock &blk = blocks[fr.block_base + b];
        uint64_t *sq = seqs + blk.seq_base;
        uint32_t *cw = mpos + blk.seq_base;
        const uint32_t nseq = blk.nseq;
        uint64_t n4[4];
        first4(b + 1, n4);
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = tid + u * 1024u; if (k < nseq) one(e4[u], sq, cw, k); }
        for (uint32_t k = tid + 4096; k < nseq; k += 1024) one(sq[k], sq, cw, k);          // (more than 4096 sequences in a block: rare)
#pragma unroll
        for (int u = 0; u < 4; u++) e4[u] = n4[u];
    }
    if (mine) atomicAdd(&s_nseq, mine);
    // the predefined set is the starting point (value tables, and whatever stays predefined)
    for (uint32_t i = tid; i < sizeof(ZkEncTables) / 4; i += 1024) ((uint32_t *)T)[i] = ((const uint32_t *)predef)[i];
    __syncthreads();
    if (tid < 3 && s_nseq >= min_seq) {
        const int t = (int)tid;
        const int nsym = t == 0 ? 36 : t == 1 ? 32 : 53, L = zke_fse_log(t, s_nseq);
        if (zke_fse_normalize(h[t], nsym, L, norm[t])) {
            int last = nsym;
            while (last > 0 && norm[t][last - 1] == 0) last--;
            const uint32_t d = zke_fse_write_ncount(T->desc[t], ZKE_DESC_CAP, norm[t], last, L);
            if (d) {
                uint16_t *st = t == 0 ? T->ll_state : t == 1 ? T->of_state : T->ml_state;
                uint32_t *dfs = t == 0 ? T->ll_dfs : t == 1 ? T->of_dfs : T->ml_dfs;
                uint32_t *dnb = t == 0 ? T->ll_dnb : t == 1 ? T->of_dnb : T->ml_dnb;
                zke_build_ctable(norm[t], nsym, L, st, dfs, dnb, sym[t], cumul[t]);
                T->al[t] = (uint32_t)L; T->dlen[t] = d;
                atomicOr(&T->custom, 1u << t);
            }
        }
    }
}

#ifndef ZKE_ENT_WAVES
#define ZKE_ENT_WAVES 4
#endif
// waves of a workgroup: 0 = literal counts, Huffman builds, literal writers; 1 = sequence chains; then the sequence emitter and the
// helpers that count literals beside wave 0 (4 waves: 2 = emitter, 3 = helper; 5 waves: 2, 3 = helpers, 4 = emitter -- measured:
// five-wave workgroups live shorter, 1.42 instead of ~1.45 M clocks, but fewer of them are on the machine at a time, 2.3 per CU
// instead of 2.8)
constexpr int ZKE_ENT_THREADS = 64 * ZKE_ENT_WAVES;
constexpr uint32_t ZKE_ENT_EMITTER = ZKE_ENT_WAVES == 5 ? 4 : 2, ZKE_ENT_HELPERS = ZKE_ENT_WAVES == 5 ? 2 : 1;
constexpr int ZKE_ENT_BLOCKS = 16;                       // blocks per workgroup: lanes = blocks for the serial bit writers

// Memory -> LDS without a register in between (global_load_lds_dwordx4): every active lane's 16 bytes at g land at LDS byte address
// lds + 16 * lane.  The compiler does not know of it -- neither of the LDS write nor of the entry in the wave's vmcnt queue: who
// reads the bytes waits with ZKE_VM_WAIT(n), n = the vector memory operations issued after this one that may still be under way
// (the queue completes in order; the compiler's own waits only ever get longer by the entries it does not know of).
__device__ __forceinline__ void zke_dma16(const void *g, uint32_t lds)
{
    uint32_t keep;
    lds = __builtin_amdgcn_readfirstlane(lds);          // the same in every lane; the instruction wants it in a scalar register
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %1, off\n\ts_nop 1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
#define ZKE_VM_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

constexpr uint32_t ZKE_SEQ_SLOT = 3 * 16 * ZKE_ENT_BLOCKS;       // bytes of one round in the sequence ring: 2 + 2 rewritten sequences and 4 code words of 16 lanes
constexpr uint32_t ZKE_SEQ_SLOTS = 6;                            // rounds between the fetch of a round and the emitter's last look at it (3 ahead of the chain, <= 2 behind)

// The sequence bitstreams of 16 blocks are written by TWO waves, lane j of each for block j:
//   the chain wave walks the three FSE states (the one thing that is serial: state -> bit count -> next state, one dependent LDS
//     lookup per state and step) and leaves every step's state bits -- one group of <= 26 -- in LDS;
//   the emitter, a round (4 sequences) or more behind, puts the groups and the sequences' extra bits into the stream (ZkeBitsL).
// Both are bound by the instructions they issue; as one wave a step took 76 of them.
// The rounds are fetched three ahead, straight into LDS (zke_dma16; registers would have to be moved from round to round, and a move
// waits for its load): 16 + 16 bytes of the rewritten sequences (the emitter's) and 16 of code words (the chain's).
// Rounds are numbered through the workgroup's life (r, the same in both waves): the blocks of the workgroup's first frame are
// written first, the others (their tables are read out of HBM) after them.
struct ZkeSeqShared {
    zke_u32x4 ring[ZKE_SEQ_SLOTS][3][ZKE_ENT_BLOCKS];    // [slot][sequences i-3, i-2 | i-1, i | codes][lane]
    zke_u32x4 g[ZKE_SEQ_SLOTS][ZKE_ENT_BLOCKS];          // a round's four groups: bits | count << 26
    uint32_t chain_round, emit_round;                     // rounds written by the chain wave / read by the emitter
};
__device__ __forceinline__ void zke_wait_round(uint32_t counter_at, int32_t need)
{
    while ((int32_t)(zke_lds_ld_at<uint32_t>(counter_at) - (uint32_t)need) < 0) __builtin_amdgcn_s_sleep(1);
}
// full rounds of a block with nseq sequences: the last one is the initial state, the lowest <= 3 are left to the tail round
__device__ __forceinline__ uint32_t zke_seq_rounds(uint32_t nseq) { return nseq >= 5 ? (nseq - 1) / 4 : 0; }

// r0: the phase's first round; rmax: the phase's full rounds (the longest lane's); the tail round is r0 + rmax.
template <typename TT>
__device__ __forceinline__ void zke_seq_chain(const TT &T, const ZkEncBlock &blk, const uint64_t *seqs, const uint32_t *mpos, uint32_t sh, uint32_t lane,
                                              uint32_t r0, uint32_t rmax)
{
    const uint32_t nseq = blk.nseq;
    const uint64_t *sq = seqs + blk.seq_base;
    const uint32_t *cw = mpos + blk.seq_base;
    const uint32_t ring = sh + (uint32_t)offsetof(ZkeSeqShared, ring), gat = sh + (uint32_t)offsetof(ZkeSeqShared, g) + 16 * lane;
    const uint32_t emit_at = sh + (uint32_t)offsetof(ZkeSeqShared, emit_round), chain_at = sh + (uint32_t)offsetof(ZkeSeqShared, chain_round);
    int32_t i = (int32_t)nseq - 2;
    // round r = sequences i - 4 (r - r0) ... - 3 (while that is >= 0); fetched from the lowest of them on, from the array's start
    // for the rounds a lane does not have (it never reads those).  The slot was round r - 8's: the emitter must be past that
    auto fetch = [&](int32_t top, uint32_t r) {
        const int32_t lo = top >= 3 ? top - 3 : 0;
        const uint32_t at = ring + (r % ZKE_SEQ_SLOTS) * ZKE_SEQ_SLOT;
        zke_dma16(sq + lo, at); zke_dma16(sq + lo + 2, at + 16 * ZKE_ENT_BLOCKS); zke_dma16(cw + lo, at + 32 * ZKE_ENT_BLOCKS);
    };
    zke_wait_round(emit_at, (int32_t)r0 + 2 - (int32_t)ZKE_SEQ_SLOTS + 1);
    fetch(i, r0); fetch(i - 4, r0 + 1); fetch(i - 8, r0 + 2);
    const uint32_t y0 = cw[0], y1 = cw[nseq > 1 ? 1 : 0], y2 = cw[nseq > 2 ? 2 : 0];     // the tail's codes
    uint32_t sl, sm, so;
    {
        const uint32_t c = cw[nseq - 1], llc = c & 0xFF, mlc = (c >> 8) & 0xFF, ofc = c >> 16;
        sm = zke_cinit(T.ml_state, T.ml_dnb[mlc], T.ml_dfs[mlc]);
        so = zke_cinit(T.of_state, T.of_dnb[ofc], T.of_dfs[ofc]);
        sl = zke_cinit(T.ll_state, T.ll_dnb[llc], T.ll_dfs[llc]);
    }
    asm volatile("" :: "v"(y0), "v"(y1), "v"(y2));
    ZKE_VM_WAIT(0);                                                               // the first three rounds are in the ring
    // what a step needs of the tables besides the states' own cells: found by the codes alone, so fetched for four steps at once
    struct Pre { uint32_t odn, odf, mdn, mdf, ldn, ldf; };
    auto pre = [&](uint32_t c) {
        const uint32_t llc = c & 0xFF, mlc = (c >> 8) & 0xFF, ofc = c >> 16;
        return Pre{T.of_dnb[ofc], T.of_dfs[ofc], T.ml_dnb[mlc], T.ml_dfs[mlc], T.ll_dnb[llc], T.ll_dfs[llc]};
    };
    // one step: the three states' bits as one group (OF, ML, LL from the low end) | their count << 26
    auto states = [&](const Pre &p) {
        const uint32_t no = (so + p.odn) >> 16, nm = (sm + p.mdn) >> 16, nl = (sl + p.ldn) >> 16;
        uint32_t g = so & ((1u << no) - 1u);
        g |= (sm & ((1u << nm) - 1u)) << no;
        g |= (sl & ((1u << nl) - 1u)) << (no + nm);
        so = T.of_state[(so >> no) + p.odf]; sm = T.ml_state[(sm >> nm) + p.mdf]; sl = T.ll_state[(sl >> nl) + p.ldf];
        return g | (no + nm + nl) << 26;
    };
    uint32_t r = r0;
    while (i >= 3) {
        // the fetch below overwrites round r - 5's slot; round r was fetched three rounds ago and only the fetches of rounds r + 1 and
        // r + 2 are behind it in the queue (every round issues its 3 as long as one lane is in the loop; this wave stores nothing)
        zke_wait_round(emit_at, (int32_t)r + 3 - (int32_t)ZKE_SEQ_SLOTS + 1);
        ZKE_VM_WAIT(6);
        const uint32_t slot = (r % ZKE_SEQ_SLOTS);
        const zke_u32x4 c4 = zke_lds_ld_at<zke_u32x4>(ring + slot * ZKE_SEQ_SLOT + 32 * ZKE_ENT_BLOCKS + 16 * lane);
        fetch(i - 12, r + 3);
        const Pre p0 = pre(c4.w), p1 = pre(c4.z), p2 = pre(c4.y), p3 = pre(c4.x);
        zke_u32x4 g;
        g.x = states(p0); g.y = states(p1); g.z = states(p2); g.w = states(p3);
        zke_lds_st_at<zke_u32x4>(gat + slot * (16 * ZKE_ENT_BLOCKS), g);
        zke_lds_st_at<uint32_t>(chain_at, r + 1);
        i -= 4; r++;
    }
    // the tail round: the lowest <= 3 sequences, then the final states
    r = r0 + rmax;
    zke_wait_round(emit_at, (int32_t)r - (int32_t)ZKE_SEQ_SLOTS + 1);
    zke_u32x4 g; g.x = g.y = g.z = 0;
    if (i >= 2) g.x = states(pre(y2));
    if (i >= 1) g.y = states(pre(y1));
    if (i >= 0) g.z = states(pre(y0));
    { const uint32_t am = T.al[2], ao = T.al[1], al = T.al[0];
      g.w = (sm & ((1u << am) - 1u)) | (so & ((1u << ao) - 1u)) << am | (sl & ((1u << al) - 1u)) << (am + ao) | (am + ao + al) << 26; }
    zke_lds_st_at<zke_u32x4>(gat + (r % ZKE_SEQ_SLOTS) * (16 * ZKE_ENT_BLOCKS), g);
    zke_lds_st_at<uint32_t>(chain_at, r + 1);
}

// returns the stream's size (0: it does not fit into the block's bytes)
__device__ __forceinline__ uint32_t zke_seq_emit(const ZkEncBlock &blk, const uint64_t *seqs, uint8_t *scratch, uint32_t sh, uint32_t stage, uint32_t lane,
                                                 uint32_t r0, uint32_t rmax)
{
    const uint32_t nseq = blk.nseq, q = (blk.nlit + 3) / 4, scap = q + (q >> 1) + 16;
    const uint64_t *sq = seqs + blk.seq_base;
    ZkeBitsL b; b.init(scratch + blk.scratch_base + ZKE_SMALL + 4 * scap, blk.bsz, blk.bsz + 64, stage + ZKE_STAGE * lane);
    const uint32_t ring = sh + (uint32_t)offsetof(ZkeSeqShared, ring) + 16 * lane, gat = sh + (uint32_t)offsetof(ZkeSeqShared, g) + 16 * lane;
    const uint32_t emit_at = sh + (uint32_t)offsetof(ZkeSeqShared, emit_round), chain_at = sh + (uint32_t)offsetof(ZkeSeqShared, chain_round);
    int32_t i = (int32_t)nseq - 2;
    const uint64_t x0 = sq[0], x1 = sq[nseq > 1 ? 1 : 0], x2 = sq[nseq > 2 ? 2 : 0];     // the tail's sequences
    // the extra bits of a sequence, in two parts of <= 32 (the second one is empty unless an offset beyond 2^16 meets long literal
    // runs / matches)
    auto extras = [&](uint64_t x) {
        const uint32_t cnt = (uint32_t)(x >> 58), c0 = cnt > 32 ? 32 : cnt;
        b.put((uint32_t)x, c0); b.flush(); b.put((uint32_t)(x >> 32) & 0x3FFFFFFu, cnt - c0); b.flush();
    };
    auto group = [&](uint32_t g) { b.put(g & 0x3FFFFFFu, g >> 26); b.flush(); };
    extras(sq[nseq - 1]);
    uint32_t r = r0, since = 0;
    while (i >= 3) {
        zke_wait_round(chain_at, (int32_t)r + 1);
        const uint32_t slot = (r % ZKE_SEQ_SLOTS);
        const zke_u32x4 lo2 = zke_lds_ld_at<zke_u32x4>(ring + slot * ZKE_SEQ_SLOT), hi2 = zke_lds_ld_at<zke_u32x4>(ring + slot * ZKE_SEQ_SLOT + 16 * ZKE_ENT_BLOCKS);
        const zke_u32x4 g = zke_lds_ld_at<zke_u32x4>(gat + slot * (16 * ZKE_ENT_BLOCKS));
        zke_lds_st_at<uint32_t>(emit_at, r + 1);                                 // behind the reads in the LDS queue
        const uint32_t most = max(max(hi2.w, hi2.y), max(lo2.w, lo2.y)) >> 26;     // extra bits of the round's longest
        if (__ballot(most > 32)) {
            group(g.x); extras(hi2.z | (uint64_t)hi2.w << 32); group(g.y); extras(hi2.x | (uint64_t)hi2.y << 32);
            group(g.z); extras(lo2.z | (uint64_t)lo2.w << 32); group(g.w); extras(lo2.x | (uint64_t)lo2.y << 32);
        } else {                                                                  // one basic block
            group(g.x); b.put(hi2.z, hi2.w >> 26); b.flush(); group(g.y); b.put(hi2.x, hi2.y >> 26); b.flush();
            group(g.z); b.put(lo2.z, lo2.w >> 26); b.flush(); group(g.w); b.put(lo2.x, lo2.y >> 26); b.flush();
        }
        if (++since == 2) { b.drain(); since = 0; }                               // <= 2 rounds x 4 x 83 bits since the last one
        i -= 4; r++;
    }
    b.drain();
    r = r0 + rmax;
    zke_wait_round(chain_at, (int32_t)r + 1);
    const zke_u32x4 g = zke_lds_ld_at<zke_u32x4>(gat + (r % ZKE_SEQ_SLOTS) * (16 * ZKE_ENT_BLOCKS));
    zke_lds_st_at<uint32_t>(emit_at, r + 1);
    if (i >= 2) { group(g.x); extras(x2); }
    if (i >= 1) { group(g.y); extras(x1); }
    if (i >= 0) { group(g.z); extras(x0); }
    b.put(g.w & 0x3FFFFFFu, g.w >> 26);
    return b.close();
}

// One workgroup handles 16 consecutive blocks so that the serial bit writers fill their waves with REAL work:
// wave 1 lanes 0-15 = the 16 sequence bitstreams -- the longest chain of the workgroup, so it starts with the kernel (its input
// was rewritten by zk_k_enc_fse_build); meanwhile waves 0, 2, 3 detect RLE blocks and count literals, wave 0 builds the 16 Huffman
// codes (lane j builds block j's) and then writes the 16 x 4 literal streams; all waves copy the payloads together at the end.
// (A wave with fewer than 16 active lanes runs ~3x slower on gfx950, tools/ubench/lat3.hip.)
#ifndef ZKE_HUF_BUILDS
#define ZKE_HUF_BUILDS 8
#endif
// The workgroup's LDS.  It is passed as DYNAMIC shared memory: knowing its size the compiler concludes that four waves fit per SIMD
// and rounds the kernel's register count up to the least that allows no more (97 instead of the 91 it uses) -- five-wave workgroups
// with 104 allocated registers go two to a CU (tools/ubench/occ.hip), with 96 three.
struct ZkeEntShared {
    uint32_t cnt[ZKE_ENT_BLOCKS][256];                    // literal histograms; later the literal writers' staging areas
    ZkeSeqShared seq;                                      // the sequence waves' rounds (zke_seq_chain / zke_seq_emit)
    zke_u32x4 seq_stage[ZKE_ENT_BLOCKS * ZKE_STAGE / 16];  // the emitter's output on its way to HBM (ZkeBitsL)
    ZkEncTables T;                                         // the FSE compression tables of the frame of the workgroup's first block
    ZkHufCode hw[ZKE_ENT_BLOCKS];
    ZkHufBuild hbuild[ZKE_HUF_BUILDS];                     // trees built side by side (a build needs 1.9 KiB of LDS, the codes 384 B): 16 / NHB rounds
    uint32_t sizes[ZKE_ENT_BLOCKS][5];                     // 4 literal streams + sequence bitstream
    uint32_t lit_mode[ZKE_ENT_BLOCKS], maxbits[ZKE_ENT_BLOCKS], tree[ZKE_ENT_BLOCKS], diff[ZKE_ENT_BLOCKS], mode[ZKE_ENT_BLOCKS];
    uint32_t hist_done;                                    // waves 2, 3 -> wave 0: my histogram passes are done
};
static_assert(64 * ZKE_LIT_STAGE <= sizeof(uint32_t) * ZKE_ENT_BLOCKS * 256, "the literal writers' staging areas take the histograms' place");
static_assert(sizeof(ZkeEntShared) <= 51200, "three workgroups per CU and room for the checksum kernel's eight waves beside them: 160 KiB in units of 1280 bytes");
extern __shared__ __attribute__((aligned(16))) uint8_t zke_ent_lds[];
__global__ __launch_bounds__(ZKE_ENT_THREADS) void zk_k_enc_entropy(const uint8_t *src, const ZkEncFrame *frames, ZkEncBlock *blocks,
                                                                   uint32_t nblocks, uint64_t *seqs, uint32_t *mpos, const uint8_t *lits,
                                                                   uint8_t *scratch, const ZkEncTables *ftab)
{
    ZkeEntShared &S = *reinterpret_cast<ZkeEntShared *>(zke_ent_lds);
    uint32_t (&cnt)[ZKE_ENT_BLOCKS][256] = S.cnt;
    ZkEncTables &T = S.T;
    ZkHufCode (&hw)[ZKE_ENT_BLOCKS] = S.hw;
    constexpr uint32_t NHB = ZKE_HUF_BUILDS;
    ZkHufBuild (&hbuild)[NHB] = S.hbuild;
    uint32_t (&s_sizes)[ZKE_ENT_BLOCKS][5] = S.sizes;
    uint32_t (&s_lit_mode)[ZKE_ENT_BLOCKS] = S.lit_mode, (&s_maxbits)[ZKE_ENT_BLOCKS] = S.maxbits, (&s_tree)[ZKE_ENT_BLOCKS] = S.tree, (&s_diff)[ZKE_ENT_BLOCKS] = S.diff, (&s_mode)[ZKE_ENT_BLOCKS] = S.mode;
    uint32_t &s_hist_done = S.hist_done;
    ZkeSeqShared &s_seq = S.seq;
    zke_u32x4 (&s_seq_stage)[ZKE_ENT_BLOCKS * ZKE_STAGE / 16] = S.seq_stage;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // roles: see ZKE_ENT_WAVES.  (Rotating the roles over the waves from
    // workgroup to workgroup -- so that the chains of the workgroups sharing a CU would not meet on one SIMD -- changed nothing: 11.1 / 11.2 / 11.6 ms.)
    const uint32_t role = wave;
    const uint32_t b0 = blockIdx.x * ZKE_ENT_BLOCKS;
    const uint32_t nb = nblocks - b0 < (uint32_t)ZKE_ENT_BLOCKS ? nblocks - b0 : (uint32_t)ZKE_ENT_BLOCKS;

    ZKE_ECLK_BEGIN();
    for (uint32_t i = tid; i < ZKE_ENT_BLOCKS * 256; i += ZKE_ENT_THREADS) (&cnt[0][0])[i] = 0;
    const uint32_t frame_a = blocks[b0].frame;
    for (uint32_t i = tid; i < sizeof(ZkEncTables) / 4; i += ZKE_ENT_THREADS) ((uint32_t *)&T)[i] = ((const uint32_t *)&ftab[frame_a])[i];
    if (tid < ZKE_ENT_BLOCKS) s_diff[tid] = 0;
    if (tid == 0) { s_hist_done = 0; s_seq.chain_round = 0; s_seq.emit_round = 0; }
    __syncthreads();
    if (role == 1 || role == ZKE_ENT_EMITTER) {
        // the sequence bitstreams (wave 1: the FSE chains, wave 4: the emitter; lane j of both = block j) -- the longest chain of the
        // workgroup, so it starts with the kernel: the input was rewritten by zk_k_enc_fse_build, nothing of what the other waves
        // prepare is needed.  The tables come out of LDS when the block belongs to the workgroup's first frame (the rule: 16 blocks
        // of one frame per workgroup), out of HBM for the blocks of another frame in a mixed workgroup: two phases
        const uint32_t j = lane;
        const bool have = lane < (uint32_t)ZKE_ENT_BLOCKS && j < nb && blocks[b0 + (j < nb ? j : 0)].nseq != 0;
        const bool own = have && blocks[b0 + j].frame == frame_a, other = have && !own;
        const uint32_t myr = have ? zke_seq_rounds(blocks[b0 + j].nseq) : 0;
        uint32_t r_own = own ? myr : 0, r_other = other ? myr : 0;
#pragma unroll
        for (int m = 1; m < ZKE_ENT_BLOCKS; m <<= 1) { r_own = max(r_own, (uint32_t)__shfl_xor((int)r_own, m, 64)); r_other = max(r_other, (uint32_t)__shfl_xor((int)r_other, m, 64)); }
        const uint32_t r1 = __ballot(own) ? r_own + 1 : 0;                        // the second phase's first round
        const uint32_t sh = (uint32_t)(uintptr_t)&s_seq;
        uint32_t sz = 0;
        if (role == 1) {
            if (own) zke_seq_chain(T, blocks[b0 + j], seqs, mpos, sh, lane, 0, r_own);
            if (other) zke_seq_chain(ftab[blocks[b0 + j].frame], blocks[b0 + j], seqs, mpos, sh, lane, r1, r_other);
        } else {
            if (own) sz = zke_seq_emit(blocks[b0 + j], seqs, scratch, sh, (uint32_t)(uintptr_t)s_seq_stage, lane, 0, r_own);
            if (other) sz = zke_seq_emit(blocks[b0 + j], seqs, scratch, sh, (uint32_t)(uintptr_t)s_seq_stage, lane, r1, r_other);
            if (lane < (uint32_t)ZKE_ENT_BLOCKS) s_sizes[lane][4] = sz;
        }
        ZKE_ECLK(4);
    } else {
        constexpr uint32_t HT = (1 + ZKE_ENT_HELPERS) * 64;     // the lanes of wave 0 and the helpers
        const uint32_t ht = (role ? role - (3 - ZKE_ENT_HELPERS) : 0) * 64 + lane;  // my index among them
        // raw block all one byte?  literal histogram -- waves 0, 2 and 3, block after block
        for (uint32_t j = 0; j < nb; j++) {
            const ZkEncBlock &blk = blocks[b0 + j];
            const uint8_t *raw = src + frames[blk.frame].src_off + blk.bs;
            const uint8_t *lt = lits + blk.lit_base;
            // 8 bytes per load, four loads in flight per lane (a byte-at-a-time loop is one L1 round trip per byte)
            const uint64_t first8 = raw[0] * 0x0101010101010101ull;
            uint64_t dx = 0;
            const uint32_t bw = blk.bsz >> 3, lw = blk.nlit >> 3;
            for (uint32_t i = ht; i < bw; i += 4 * HT) {
                uint64_t w[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t k = i + u * HT; w[u] = zk_ld64(raw + 8 * (k < bw ? k : i)); }
#pragma unroll
                for (int u = 0; u < 4; u++) dx |= w[u] ^ first8;
            }
            for (uint32_t i = 8 * bw + ht; i < blk.bsz; i += HT) dx |= raw[i] ^ raw[0];
            if (dx) s_diff[j] = 1;
            for (uint32_t i = ht; i < lw; i += 4 * HT) {
                uint64_t w[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint32_t k = i + u * HT; w[u] = zk_ld64(lt + 8 * (k < lw ? k : i)); }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i + u * HT < lw) {
#pragma unroll
                        for (int t = 0; t < 8; t++) atomicAdd(&cnt[j][(w[u] >> (8 * t)) & 0xFF], 1u);
                    }
            }
            for (uint32_t i = 8 * lw + ht; i < blk.nlit; i += HT) atomicAdd(&cnt[j][lt[i]], 1u);
        }

        ZKE_ECLK(0);
        // the Huffman builds (wave 0) need every wave's counts: waves 2 and 3 sign off (a wave's LDS operations complete in
        // order, so their counts are in place when the flag moves), wave 0 waits for both
        if (role >= 2) { if (lane == 0) atomicAdd(&s_hist_done, 1u); }
        else {
            while (__hip_atomic_load(&s_hist_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ZKE_ENT_HELPERS) __builtin_amdgcn_s_sleep(2);
            ZKE_ECLK(1);
            // literal mode + Huffman code, NHB blocks at a time (a build's scratch is 1.9 KiB)
            for (uint32_t round = 0; round < ZKE_ENT_BLOCKS / NHB; round++) {
                // (A) all lanes, block after block: the literal mode (1 = RLE, 2 = Huffman in 4 streams, 0 = raw; oracle encode_literals) and,
                // for a block that gets a code, its symbols in the order (count, symbol): a lane ranks symbols lane and lane + 64 against all
                for (uint32_t jj = 0; jj < NHB; jj++) {
                    const uint32_t j = jj + NHB * round;
                    if (j >= nb) break;
                    const uint32_t c0 = cnt[j][lane], c1 = cnt[j][lane + 64], c2 = cnt[j][lane + 128], c3 = cnt[j][lane + 192];
                    const uint64_t m0 = __ballot(c0 != 0), m1 = __ballot(c1 != 0), m2 = __ballot(c2 != 0), m3 = __ballot(c3 != 0);
                    const uint32_t distinct = (uint32_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                    const uint32_t maxsym = m3 ? 255u - (uint32_t)__builtin_clzll(m3) : m2 ? 191u - (uint32_t)__builtin_clzll(m2) : m1 ? 127u - (uint32_t)__builtin_clzll(m1)
                                                                                        : m0 ? 63u - (uint32_t)__builtin_clzll(m0) : 0u;
                    const uint32_t nlit = blocks[b0 + j].nlit;
                    uint32_t mode = 0;
                    if (nlit > 0 && distinct == 1) mode = 1;
                    else if (nlit >= 64 && maxsym < 128) {
                        mode = 3;                                                     // to be built
                        const uint32_t key0 = c0 ? (c0 << 8) | lane : 0xFFFFFFFFu, key1 = c1 ? (c1 << 8) | (lane + 64) : 0xFFFFFFFFu;
                        uint32_t r0 = 0, r1 = 0;
#pragma unroll
                        for (int t = 0; t < 64; t++) {
                            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key0, t), k1 = (uint32_t)__builtin_amdgcn_readlane((int)key1, t);
                            r0 += (k0 < key0) + (k1 < key0); r1 += (k0 < key1) + (k1 < key1);
                        }
                        if (c0) hbuild[jj].idx[r0] = (uint8_t)lane;
                        if (c1) hbuild[jj].idx[r1] = (uint8_t)(lane + 64);
                    }
                    if (lane == 0) { s_lit_mode[j] = mode; s_tree[j] = maxsym; s_maxbits[j] = distinct; }
                }
                // (B) one lane per block: the tree and the code lengths.  Block j on lanes j % NHB and j % NHB + NHB (the second lane
                // shadows the first with identical LDS writes: >= 16 active lanes, see zk_decode.hip)
                if (lane < 2 * NHB) {
                    const uint32_t j = (lane & (NHB - 1)) + NHB * round;
                    if (j < nb && s_lit_mode[j] == 3) {
                        const int mb = zke_huf_lengths(cnt[j], (int)s_tree[j] + 1, &hbuild[lane & (NHB - 1)], hw[j].len, (int)s_maxbits[j]);
                        s_lit_mode[j] = mb > 0 ? 2u : 0u; s_maxbits[j] = mb > 0 ? (uint32_t)mb : 0u;
                    }
                }
                // (C) all lanes again: the canonical codes exactly as the decoder's table fill assigns them -- weight 1 first, symbols
                // ascending (zke_huf_codes, here with ballots instead of 11 passes over the symbols)
                for (uint32_t jj = 0; jj < NHB; jj++) {
                    const uint32_t j = jj + NHB * round;
                    if (j >= nb) break;
                    if (s_lit_mode[j] != 2) continue;
                    const uint32_t mb = s_maxbits[j];
                    const uint32_t nsym = s_tree[j] + 1;                              // the lengths beyond the block's last symbol were never written
                    const uint32_t l0 = lane < nsym ? hw[j].len[lane] : 0u, l1 = lane + 64 < nsym ? hw[j].len[lane + 64] : 0u;
                    const uint32_t w0 = l0 ? mb + 1 - l0 : 0, w1 = l1 ? mb + 1 - l1 : 0;
                    const uint64_t below = (1ull << lane) - 1ull;
                    uint32_t pos = 0;
                    for (uint32_t wt = 1; wt <= mb; wt++) {
                        const uint64_t b0m = __ballot(w0 == wt), b1m = __ballot(w1 == wt);
                        const uint32_t n0 = (uint32_t)__popcll(b0m), n1 = (uint32_t)__popcll(b1m);
                        if (w0 == wt) hw[j].code[lane] = (uint16_t)((pos >> (wt - 1)) + (uint32_t)__popcll(b0m & below));
                        if (w1 == wt) hw[j].code[lane + 64] = (uint16_t)((pos >> (wt - 1)) + n0 + (uint32_t)__popcll(b1m & below));
                        pos += (n0 + n1) << (wt - 1);
                    }
                }
            }

            ZKE_ECLK(3);
            // serial bit writers of the literals: lane = (block, stream)
            {
                const uint32_t j = lane >> 2, k = lane & 3;
                if (j < nb && s_lit_mode[j] == 2) {
                    const ZkEncBlock &blk = blocks[b0 + j];
                    const uint32_t nlit = blk.nlit, q = (nlit + 3) / 4, scap = q + (q >> 1) + 16;
                    const uint32_t n_k = k < 3 ? q : nlit - 3 * q;
                    const uint8_t *sp = lits + blk.lit_base + k * q;
                    // last symbol first; the stream is read 8 bytes at a time, three words ahead (unconditional loads: a clamped address
                    // reads the stream's first bytes again).  The bits go through the lane's staging area in LDS (the histograms' space:
                    // the codes are built) and reach HBM 16 bytes at a time, every fourth round
                    ZkeBitsL b; b.init(scratch + blk.scratch_base + ZKE_SMALL + k * scap, scap - 8, scap, (uint32_t)(uintptr_t)&cnt[0][0] + ZKE_LIT_STAGE * lane);
                    const ZkHufCode &h = hw[j];
                    uint32_t i = n_k;
                    auto two = [&](uint32_t s0, uint32_t s1) {                                      // <= 22 bits
                        const uint32_t l0 = h.len[s0];
                        b.put(h.code[s0] | (uint32_t)h.code[s1] << l0, l0 + h.len[s1]); b.flush();
                    };
                    for (uint32_t r = n_k & 7; r; r--) { const uint32_t sy = sp[--i]; b.put(h.code[sy], h.len[sy]); b.flush(); }
                    auto ldw = [&](uint32_t at) { return zk_ld64(sp + (at >= 8 ? at - 8 : 0)); };       // symbols [at - 8, at)
                    auto word = [&](uint64_t w) {
#pragma unroll
                        for (int t = 7; t >= 0; t -= 2) two((uint32_t)(w >> (8 * t)) & 0xFF, (uint32_t)(w >> (8 * t - 8)) & 0xFF);
                    };
                    uint64_t w0 = ldw(i), w1 = ldw(i >= 8 ? i - 8 : 0), w2 = ldw(i >= 16 ? i - 16 : 0);
                    uint32_t rounds = 0;
                    while (i >= 24) {
                        const uint64_t n0 = ldw(i - 24), n1 = ldw(i >= 32 ? i - 32 : 0), n2 = ldw(i >= 40 ? i - 40 : 0);
                        word(w0); word(w1); word(w2);
                        if ((++rounds & 3) == 0) b.drain();                                         // <= 4 x 33 bytes since the last one
                        w0 = n0; w1 = n1; w2 = n2;
                        i -= 24;
                    }
                    b.drain();
                    if (i >= 8) word(w0);
                    if (i >= 16) word(w1);
                    s_sizes[j][k] = b.close();
                }

            }
            ZKE_ECLK(4);
        }
    }
    __syncthreads();
    ZKE_ECLK(5);
    // layout of every block payload (lane j of wave 0)
    if (tid < ZKE_ENT_BLOCKS && tid < nb) {
        const uint32_t j = tid;
        ZkEncBlock *blk = &blocks[b0 + j];
        const uint32_t nlit = blk->nlit, nseq = blk->nseq, bsz = blk->bsz;
        const uint32_t hdr = nlit < 1024 ? 3 : nlit < 16384 ? 4 : 5;
        uint32_t mode = 2, total = 0;
        const uint32_t raw_hdr = nlit < 32 ? 1 : nlit < 4096 ? 2 : 3;
        uint32_t lm = s_lit_mode[j], lit_sz = 0;
        if (lm == 2) {
            const uint32_t tree = 1 + (s_tree[j] + 1) / 2;         // weights of symbols 0..maxsym-1, two per byte
            const uint32_t *z = s_sizes[j];
            const uint32_t comp = tree + 6 + z[0] + z[1] + z[2] + z[3];
            const bool ok = z[0] && z[1] && z[2] && z[3] && z[0] < 65536 && z[1] < 65536 && z[2] < 65536;
            const uint32_t lim = hdr == 3 ? 1024u : hdr == 4 ? 16384u : 262144u;
            if (ok && comp < nlit - (nlit >> 6) && comp < lim) lit_sz = hdr + comp; else lm = 0;
        }
        if (lm == 1) lit_sz = raw_hdr + 1;
        if (lm == 0) lit_sz = raw_hdr + nlit;
        const uint32_t nh = nseq < 128 ? 1 : nseq < 0x7F00 ? 2 : 3;
        total = lit_sz + nh + (nseq ? 1 + s_sizes[j][4] : 0);
        if (nseq && s_sizes[j][4] == 0) total = 0xFFFFFFFFu;
        if (!s_diff[j] && bsz > 1) mode = 1;
        else if (total >= bsz) mode = 0;
        s_lit_mode[j] = lm; s_mode[j] = mode;
        blk->mode = mode;
        blk->csize = mode == 2 ? total : mode == 1 ? 1 : bsz;
        blk->rle_byte = src[frames[blk->frame].src_off + blk->bs];
    }
    __syncthreads();
    // the small parts of every compressed block (lane j of wave 0): the piece table and the two heads.  The streams stay where their
    // writers left them: zk_k_enc_assemble puts the payload together at its place in the output
    if (tid < ZKE_ENT_BLOCKS && tid < nb && s_mode[tid] == 2) {
        const uint32_t j = tid;
        const ZkEncBlock &blk = blocks[b0 + j];
        const uint32_t nlit = blk.nlit, nseq = blk.nseq;
        uint8_t *small = scratch + blk.scratch_base, *head = small + sizeof(ZkEncPieces);
        const uint8_t *lt = lits + blk.lit_base;
        const uint32_t lm = s_lit_mode[j];
        const uint32_t raw_hdr = nlit < 32 ? 1 : nlit < 4096 ? 2 : 3;
        const uint32_t *z = s_sizes[j];
        ZkEncPieces pc;
        pc.z[0] = pc.z[1] = pc.z[2] = pc.z[3] = 0; pc.zs = nseq ? z[4] : 0; pc.lit_mode = (uint8_t)lm; pc.pad = 0;
        uint32_t w = 0, lit_bytes = 0;
        if (lm == 2) {
            const uint32_t hdr = nlit < 1024 ? 3 : nlit < 16384 ? 4 : 5, tree = 1 + (s_tree[j] + 1) / 2, mb = s_maxbits[j];
            const uint32_t comp = tree + 6 + z[0] + z[1] + z[2] + z[3];
            const uint64_t h = hdr == 3 ? (2ull | (1 << 2) | ((uint64_t)nlit << 4) | ((uint64_t)comp << 14))
                             : hdr == 4 ? (2ull | (2 << 2) | ((uint64_t)nlit << 4) | ((uint64_t)comp << 18))
                                        : (2ull | (3 << 2) | ((uint64_t)nlit << 4) | ((uint64_t)comp << 22));
            for (uint32_t i = 0; i < hdr; i++) head[w++] = (uint8_t)(h >> (8 * i));
            const uint32_t nw = s_tree[j];                         // number of explicit weights (symbols 0..maxsym-1)
            head[w++] = (uint8_t)(127 + nw);
            for (uint32_t i = 0; i < nw; i += 2) {
                const uint32_t w0 = hw[j].len[i] ? mb + 1 - hw[j].len[i] : 0;
                const uint32_t w1 = (i + 1 < nw && hw[j].len[i + 1]) ? mb + 1 - hw[j].len[i + 1] : 0;
                head[w++] = (uint8_t)((w0 << 4) | w1);
            }
            for (int k = 0; k < 3; k++) { head[w++] = (uint8_t)z[k]; head[w++] = (uint8_t)(z[k] >> 8); }
            for (int k = 0; k < 4; k++) pc.z[k] = (uint16_t)z[k];
            lit_bytes = z[0] + z[1] + z[2] + z[3];
        } else {
            const uint32_t t = lm;                                 // 0 raw, 1 rle
            if (raw_hdr == 1) head[0] = (uint8_t)(t | (nlit << 3));
            else if (raw_hdr == 2) { head[0] = (uint8_t)(t | (1 << 2) | (nlit << 4)); head[1] = (uint8_t)(nlit >> 4); }
            else { head[0] = (uint8_t)(t | (3 << 2) | (nlit << 4)); head[1] = (uint8_t)(nlit >> 4); head[2] = (uint8_t)(nlit >> 12); }
            w = raw_hdr;
            if (lm == 1) head[w++] = lt[0]; else lit_bytes = nlit;
        }
        pc.head_lit = (uint8_t)w;
        uint32_t v = w;
        if (nseq < 128) head[v++] = (uint8_t)nseq;
        else if (nseq < 0x7F00) { head[v++] = (uint8_t)((nseq >> 8) + 128); head[v++] = (uint8_t)nseq; }
        else { head[v++] = 255; head[v++] = (uint8_t)(nseq - 0x7F00); head[v++] = (uint8_t)((nseq - 0x7F00) >> 8); }
        if (nseq) {
            // Symbol_Compression_Modes in its Repeat_Mode form (the frame's own tables: 3, predefined ones: 0); the block
            // that turns out to be the frame's first compressed one gets FSE_Compressed_Mode + the descriptions at assembly
            blocks[b0 + j].modes_off = w + lit_bytes + (v - w);                       // where the byte sits in the payload
            head[v++] = (uint8_t)zke_modes_byte(ftab[blk.frame].custom, 3);
        }
        pc.head_seq = (uint8_t)(v - w);
        memcpy(small, &pc, sizeof pc);
    }
    ZKE_ECLK(6);
    ZKE_ECLK_END();
}

// ------------------------------------------------------------------------------------------------ frame sizes, scan, assemble
__global__ __launch_bounds__(64) void zk_k_enc_sizes(const ZkEncFrame *frames, uint32_t nframes, ZkEncBlock *blocks, const ZkEncTables *ftab, int checksum,
                                                     uint64_t *c_size64, uint32_t *c_sizes, uint32_t *d_sizes)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    const ZkEncFrame fr = frames[f];
    uint64_t c;
    if (fr.d_size == 0) c = 9;                                      // 28 B5 2F FD 20 00 01 00 00
    else {
        c = 6;
        // the frame's first compressed block with sequences defines its tables: it grows by the descriptions
        bool defined = ftab[f].custom == 0;
        for (uint32_t b = 0; b < fr.n_blocks; b++) {
            ZkEncBlock &blk = blocks[fr.block_base + b];
            if (!defined && blk.mode == 2 && blk.nseq) {
                blk.is_def = 1;
                blk.csize += ftab[f].dlen[0] + ftab[f].dlen[1] + ftab[f].dlen[2];
                defined = true;
            }
            blk.out_at = (uint32_t)c;
            c += 3 + blk.csize;
        }
    }
    if (checksum) c += 4;
    c_size64[f] = c;
    if (c_sizes) c_sizes[f] = (uint32_t)c;
    if (d_sizes) d_sizes[f] = fr.d_size;
}

// exclusive scan of n u64 values (one workgroup); out[n] = total
__global__ __launch_bounds__(1024) void zk_k_scan64(const uint64_t *in, uint32_t n, uint64_t *out)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + tid;
        const uint64_t v = i < n ? in[i] : 0;
        uint64_t x = v;
        for (int d = 1; d < 64; d <<= 1) { uint64_t y = __shfl_up(x, d, 64); if ((int)lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint64_t pre = carry;
        for (uint32_t w = 0; w < wave; w++) pre += wsum[w];
        if (i < n) out[i] = pre + x - v;
        __syncthreads();
        if (tid == 1023) carry = pre + x;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

// Workgroups [0, nframes): what belongs to a frame as a whole (magic + header, the empty frame's bytes, the checksum).
// Workgroups [nframes, nframes + nblocks): one block each -- its header and payload go to out_off[frame] + out_at (the
// prefix sums of zk_k_enc_sizes), so a frame of 128 MiB is copied by 4096 workgroups, not by one.
// n bytes by the 256 lanes of a workgroup: 16 bytes per lane and step (any alignment), four steps in flight, the last bytes one by one
__device__ __forceinline__ void zke_copy_wg(uint8_t *dst, const uint8_t *from, uint32_t n, uint32_t tid)
{
    const uint32_t nq = n >> 4;
    for (uint32_t b0 = 0; b0 < nq; b0 += 4 * 256) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = b0 + u * 256 + tid; memcpy(&v[u], from + 16 * (size_t)(i < nq ? i : 0), 16); }
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = b0 + u * 256 + tid; if (i < nq) memcpy(dst + 16 * (size_t)i, &v[u], 16); }
    }
    const uint32_t t = 16 * nq + tid;
    if (t < n) dst[t] = from[t];
}
__global__ __launch_bounds__(256) void zk_k_enc_assemble(const uint8_t *src, const ZkEncFrame *frames, uint32_t nframes, const ZkEncBlock *blocks, const ZkEncTables *ftab,
                                                         const uint8_t *lits, const uint8_t *scratch, const uint64_t *out_off, const uint64_t *c_size64, const uint64_t *hashes,
                                                         int checksum, uint8_t *dst)
{
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x < nframes) {
        if (tid) return;
        const uint32_t f = blockIdx.x;
        const ZkEncFrame fr = frames[f];
        uint8_t *o = dst + out_off[f];
        if (fr.d_size == 0) { const uint8_t e[9] = {0x28, 0xB5, 0x2F, 0xFD, (uint8_t)(checksum ? 0x24 : 0x20), 0x00, 0x01, 0x00, 0x00}; for (int i = 0; i < 9; i++) o[i] = e[i]; }
        else { o[0] = 0x28; o[1] = 0xB5; o[2] = 0x2F; o[3] = 0xFD; o[4] = checksum ? 0x04 : 0x00; o[5] = (uint8_t)((fr.window_log - 10) << 3); }
        if (checksum) {
            const uint64_t p = c_size64[f] - 4;
            const uint32_t h = (uint32_t)hashes[f];
            o[p] = (uint8_t)h; o[p + 1] = (uint8_t)(h >> 8); o[p + 2] = (uint8_t)(h >> 16); o[p + 3] = (uint8_t)(h >> 24);
        }
        return;
    }
    const uint32_t gb = blockIdx.x - nframes;
    const ZkEncBlock &blk = blocks[gb];
    const ZkEncFrame fr = frames[blk.frame];
    uint8_t *o = dst + out_off[blk.frame];
    uint64_t p = blk.out_at;
    const uint32_t last = gb + 1 == fr.block_base + fr.n_blocks;
    const uint32_t field = blk.mode == 2 ? blk.csize : blk.bsz;     // Block_Size: regenerated size for raw / RLE
    if (tid == 0) { const uint32_t h = last | (blk.mode << 1) | (field << 3); o[p] = (uint8_t)h; o[p + 1] = (uint8_t)(h >> 8); o[p + 2] = (uint8_t)(h >> 16); }
    p += 3;
    if (blk.mode == 1) { if (tid == 0) o[p] = blk.rle_byte; }
    else if (blk.mode == 0) zke_copy_wg(o + p, src + fr.src_off + blk.bs, blk.bsz, tid);
    else {
        // the payload out of its pieces (ZkEncPieces): head of the literals section, the 4 literal streams (or the raw literal bytes), head of
        // the sequences section -- in the frame's defining block with the modes byte in its defining form and the table descriptions
        // behind it -- and the sequence bitstream
        const uint8_t *small = scratch + blk.scratch_base;
        ZkEncPieces pc; memcpy(&pc, small, sizeof pc);
        const uint32_t q = (blk.nlit + 3) / 4, scap = q + (q >> 1) + 16;
        const uint8_t *head = small + sizeof(ZkEncPieces), *stemp = small + ZKE_SMALL, *qtemp = stemp + 4 * scap;
        if (tid < pc.head_lit) o[p + tid] = head[tid];
        p += pc.head_lit;
        // the five streams at once: a lane's loads of all of them are under way before its first store
        const uint8_t *from[5]; uint32_t len[5]; uint64_t to[5];
        uint64_t w = p;
        for (int k = 0; k < 4; k++) { from[k] = pc.lit_mode == 2 ? stemp + k * scap : lits + blk.lit_base; len[k] = pc.lit_mode == 2 ? pc.z[k] : (k == 0 && pc.lit_mode == 0 ? blk.nlit : 0); to[k] = w; w += len[k]; }
        const ZkEncTables &ft = ftab[blk.frame];
        if (blk.nseq && blk.is_def) {
            const uint32_t nh = pc.head_seq - 1u;
            if (tid < nh) o[w + tid] = head[pc.head_lit + tid];
            if (tid == 0) o[w + nh] = (uint8_t)zke_modes_byte(ft.custom, 2);
            w += pc.head_seq;
            for (int t = 0; t < 3; t++) { for (uint32_t i = tid; i < ft.dlen[t]; i += 256) o[w + i] = ft.desc[t][i]; w += ft.dlen[t]; }
        } else {
            if (tid < pc.head_seq) o[w + tid] = head[pc.head_lit + tid];
            w += pc.head_seq;
        }
        from[4] = qtemp; len[4] = pc.zs; to[4] = w;
        uint32_t most = 0;
        for (int k = 0; k < 5; k++) most = len[k] > most ? len[k] : most;
        for (uint32_t b0 = 0; b0 < (most >> 4); b0 += 256) {
            uint4 v[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { const uint32_t i = b0 + tid; memcpy(&v[k], from[k] + 16 * (size_t)(i < (len[k] >> 4) ? i : 0), 16); }
#pragma unroll
            for (int k = 0; k < 5; k++) { const uint32_t i = b0 + tid; if (i < (len[k] >> 4)) memcpy(o + to[k] + 16 * (size_t)i, &v[k], 16); }
        }
#pragma unroll
        for (int k = 0; k < 5; k++) { const uint32_t t = (len[k] & ~15u) + (tid & 15); if ((tid >> 4) == (uint32_t)k && t < len[k]) o[to[k] + t] = from[k][t]; }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
// [prefix tail | frame] records for the matcher (prefix mode only; not on the hot path)
__global__ __launch_bounds__(256) void zk_k_enc_stage_hist(const uint8_t *src, const uint8_t *prefix_tail, const ZkEncFrame *frames, uint8_t *stage)
{
    const ZkEncFrame fr = frames[blockIdx.x];
    uint8_t *rec = stage + fr.m_off;
    for (uint32_t i = threadIdx.x; i < fr.hist; i += 256) rec[i] = prefix_tail[i];
    const uint8_t *from = src + fr.src_off;
    for (uint32_t i = threadIdx.x; i < fr.d_size; i += 256) rec[fr.hist + i] = from[i];
}
void zk_launch_enc_stage_hist(hipStream_t st, const uint8_t *src, const uint8_t *prefix_tail, const ZkEncFrame *frames, uint32_t nframes, uint8_t *stage)
{
    hipLaunchKernelGGL(zk_k_enc_stage_hist, dim3(nframes), dim3(256), 0, st, src, prefix_tail, frames, stage);
}
// The long-distance table over prefix[u0, plen): every sampled position (zke_ldm_selected) enters, the smallest position keeps
// a slot.  Four positions per lane and pass out of 20 bytes; the table was filled with ZKE_LDM_NONE.
__global__ __launch_bounds__(256) void zk_k_enc_ldm_build(ZkEncLdm ldm, uint32_t *table)
{
    const uint64_t n = ldm.plen - ldm.u0;
    const uint8_t *s = ldm.pfx + ldm.u0;
    for (uint64_t i = 4ull * ((uint64_t)blockIdx.x * 256 + threadIdx.x); i + ZKE_LDM_MIN <= n; i += 4ull * gridDim.x * 256) {
        uint32_t w[5];
        memcpy(w, s + i, 16);
        if (i + 20 <= n) memcpy(&w[4], s + i + 16, 4);
        else { w[4] = 0; for (uint64_t j = i + 16; j < n; j++) w[4] |= (uint32_t)s[j] << (8 * (j - i - 16)); }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k + ZKE_LDM_MIN > n) break;
            const uint32_t h = zke_ldm_hash(__builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)k), __builtin_amdgcn_alignbyte(w[2], w[1], (uint32_t)k),
                                            __builtin_amdgcn_alignbyte(w[3], w[2], (uint32_t)k), __builtin_amdgcn_alignbyte(w[4], w[3], (uint32_t)k));
            if (zke_ldm_selected(h)) atomicMin(&table[zke_ldm_slot(h, ldm.log)], (uint32_t)(i + k));
        }
    }
}
// The same for the frames' own tables (in-frame far history): blockIdx.y = frame; frame f's table = table + (f << ldm.log), of which
// its own 2^zke_ldm_log(size) entries are used.
__global__ __launch_bounds__(256) void zk_k_enc_ldm_build_frames(const uint8_t *src, ZkEncLdm ldm, uint32_t *table, uint32_t f0)
{
    const uint64_t f = (uint64_t)f0 + blockIdx.y, at = f * ldm.frame_size;
    const uint64_t n = ldm.n_total - at < ldm.frame_size ? ldm.n_total - at : ldm.frame_size;
    const uint32_t log = zke_ldm_log(n);
    const uint8_t *s = src + at;
    uint32_t *t = table + (f << ldm.log);
    for (uint64_t i = 4ull * ((uint64_t)blockIdx.x * 256 + threadIdx.x); i + ZKE_LDM_MIN <= n; i += 4ull * gridDim.x * 256) {
        uint32_t w[5];
        for (int k = 0; k < 5; k++) { w[k] = 0; const uint64_t q = i + 4 * k; if (q + 4 <= n) memcpy(&w[k], s + q, 4); else for (uint64_t b = q; b < n; b++) w[k] |= (uint32_t)s[b] << (8 * (b - q)); }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k + ZKE_LDM_MIN > n) break;
            const uint32_t h = zke_ldm_hash(__builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)k), __builtin_amdgcn_alignbyte(w[2], w[1], (uint32_t)k),
                                            __builtin_amdgcn_alignbyte(w[3], w[2], (uint32_t)k), __builtin_amdgcn_alignbyte(w[4], w[3], (uint32_t)k));
            if (zke_ldm_selected(h)) atomicMin(&t[zke_ldm_slot(h, log)], (uint32_t)(i + k));
        }
    }
}
// DENSE far history (round 6; ZkEncLdm in zk_enc_device.h, the rule: oracle/zstd_oracle_enc.c dense_build_frame / dense_lookup): two
// kernels, a workgroup per matcher segment each, leave every position's far candidate (length | catch-up << 5 | distance << 8) in
// `cand`, where the match kernel reads it in whole lines.  The two tables a position is looked up in -- smallest position per slot of
// its own segment, largest position + 1 per slot of the segment before, over the 5-byte hash of EVERY position -- exist in LDS only,
// 2^13 slots of both per pass (64 KiB: two workgroups per CU).
//   zk_k_enc_dense_part  hashes every position ONCE and sorts it by the pass its slot belongs to: per segment and pass a list of
//                        `position in the segment << 13 | slot in the pass` in `part` (count, prefix, place: every wave owns a piece of
//                        every list, so the only atomics are a wave's own LDS cursors)
//   zk_k_enc_dense_cand  per pass: the list of the segment before and the own list into the tables (every lane has an entry), the own
//                        list again for the lookups; what a pass finds is appended to the list of the position's 8192-position chunk,
//                        which lives where the chunk's entries of `cand` will be; a last sweep per chunk measures the candidates
//                        (16 bytes + the 4 bytes in front, through L2, all lanes at work) and writes the entries in whole lines.
// How it got here (profiles/r06c_dense_probe.txt, ms per 4 GiB): tables in HBM read by the match kernel 212 (two random 4-byte reads
// per input byte = a line from HBM each); one kernel that hashed both segments in every pass and filtered by slot range 107-134 (~650
// vector instructions per position, one lane in eight at work behind the filter); the partition 75; two workgroups per CU 63.  HBM-bound byte
// work: 4 bytes of `part` and 4 of `cand` per input byte.
constexpr uint32_t ZKD_PLOG = 13, ZKD_PSLOTS = 1u << ZKD_PLOG, ZKD_NBMAX = ZKE_DENSE_PASSES_MAX, ZKD_U = 8;
template <bool PLACE>
__device__ __forceinline__ void zkd_sweep(const uint8_t *frame, uint32_t s0, uint32_t e1, uint32_t fsz, uint32_t dlog, uint32_t tid, uint32_t *wcnt, uint32_t *lst)
{
    // (eight steps' input requested before the first is hashed: four waves per SIMD do not cover a trip to L2 per step)
    for (uint32_t qb = s0 + 4 * tid; qb < e1; qb += 4096 * ZKD_U) {
        uint32_t w0[ZKD_U], w1[ZKD_U];
#pragma unroll
        for (uint32_t u = 0; u < ZKD_U; u++) {
            const uint32_t q0 = qb + 4096 * u, at = q0 < e1 && q0 + 8 <= fsz ? q0 : 0;      // (none of a step's four positions has its eight bytes: the frame's first bytes, unused)
            memcpy(&w0[u], frame + at, 4); memcpy(&w1[u], frame + at + 4, 4);
        }
#pragma unroll
        for (uint32_t u = 0; u < ZKD_U; u++) {
            const uint32_t q0 = qb + 4096 * u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t q = q0 + k;
                const uint32_t lo = __builtin_amdgcn_alignbyte(w1[u], w0[u], (uint32_t)k);
                const uint32_t h = zke_hash(lo, (w1[u] >> (8 * k)) & 0xFFu, dlog);
                // a position whose four bytes are one byte stays out: a byte run, offset 1 codes it better (the twin has the numbers)
                if (q < e1 && q + 8 <= fsz && lo != (lo & 0xFFu) * 0x01010101u) {
                    const uint32_t i = atomicAdd(&wcnt[h >> ZKD_PLOG], 1u);
                    if (PLACE) lst[i] = ((q - s0) << ZKD_PLOG) | (h & (ZKD_PSLOTS - 1));
                }
            }
        }
    }
}
__global__ __launch_bounds__(1024) void zk_k_enc_dense_part(const uint8_t *src, const ZkEncFrame *segs, ZkEncLdm ldm, uint32_t *part, uint32_t *poff)
{
    // per wave and pass: its count, then its cursor.  (A counter per LANE and pass -- no two lanes on one LDS word -- measured slower, 74.9 ->
    // 98.7 ms: a wave's 64 stores then go to 64 different places of the lists instead of a handful of dense streams)
    __shared__ uint32_t cnt[16][ZKD_NBMAX];
    __shared__ uint32_t total[ZKD_NBMAX], start[ZKD_NBMAX + 1];
    const ZkEncFrame sg = segs[blockIdx.x];
    const uint8_t *frame = src + sg.src_off;
    const uint64_t left = ldm.n_total - sg.src_off;
    const uint32_t fsz = (uint32_t)(left < ldm.frame_size ? left : ldm.frame_size);
    const uint32_t s0 = sg.seg_at, e1 = s0 + sg.d_size, tid = threadIdx.x, dlog = ldm.dlog, wave = tid >> 6, nb = 1u << (dlog - ZKD_PLOG);
    uint32_t *lst = part + sg.src_off + s0;                         // the segment's lists, pass after pass
    static_assert(16 * ZKD_NBMAX <= 1024, "one counter per lane");
    if (tid < 16 * ZKD_NBMAX) (&cnt[0][0])[tid] = 0;
    __syncthreads();
    zkd_sweep<false>(frame, s0, e1, fsz, dlog, tid, cnt[wave], lst);
    __syncthreads();
    if (tid < nb) { uint32_t t = 0; for (uint32_t w = 0; w < 16; w++) t += cnt[w][tid]; total[tid] = t; }
    __syncthreads();
    if (tid == 0) { uint32_t run = 0; for (uint32_t b = 0; b < nb; b++) { start[b] = run; run += total[b]; } start[nb] = run; }
    __syncthreads();
    if (tid < nb) { uint32_t run = start[tid]; for (uint32_t w = 0; w < 16; w++) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; } }
    if (tid <= nb) poff[(size_t)blockIdx.x * (ZKD_NBMAX + 1) + tid] = start[tid];
    __syncthreads();
    zkd_sweep<true>(frame, s0, e1, fsz, dlog, tid, cnt[wave], lst);
}
__global__ __launch_bounds__(1024) void zk_k_enc_dense_cand(const uint8_t *src, const ZkEncFrame *segs, ZkEncLdm ldm, const uint32_t *part, const uint32_t *poff, uint32_t *cand)
{
    constexpr uint32_t PLOG = ZKD_PLOG, PSLOTS = ZKD_PSLOTS;
#ifndef ZKD_E
#define ZKD_E 16
#endif
    constexpr uint32_t E = ZKD_E;                                   // list entries requested per lane before the first is used (4 | 8 | 16: 50.1 | 47.9 | 47.1 ms per 4 GiB)
    constexpr uint32_t CLOG = 13, CHUNK = 1u << CLOG, NCHUNK = ZKE_SEGMENT / CHUNK;     // found candidates are listed per chunk of 8192 positions
    static_assert(ZKE_SEGMENT + ZKE_SEGMENT - ZKE_WINDOW <= (1u << (32 - CLOG)), "a list entry: position inside the chunk | (distance - ZKE_WINDOW - 1) << 13");
    static_assert(ZKE_SEGMENT <= (1u << (32 - PLOG)), "an entry of `part`: position inside the segment << 13 | slot");
    __shared__ uint32_t first[PSLOTS], last[PSLOTS];
    __shared__ uint32_t count[NCHUNK];
    // Which segment: workgroups go to the eight XCDs in turn, each with an L2 of its own; workgroup b takes segment (b % 8) * (n / 8) + b / 8, so
    // that an XCD works through CONSECUTIVE segments -- the candidates of a segment lie in its own and the segment before's bytes, which the
    // same L2 has just seen -- instead of every eighth one of all frames (62.4 -> see profiles/r06c_dense_probe.txt)
    const uint32_t nwg = gridDim.x, seg = (nwg & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3);
    const ZkEncFrame sg = segs[seg];
    const uint8_t *frame = src + sg.src_off;
    const uint64_t left = ldm.n_total - sg.src_off;
    const uint32_t fsz = (uint32_t)(left < ldm.frame_size ? left : ldm.frame_size);
    const uint32_t s0 = sg.seg_at, e1 = s0 + sg.d_size, tid = threadIdx.x, dlog = ldm.dlog;
    uint32_t *out = cand + sg.src_off;                              // indexed by the position inside the frame
    const uint32_t *olist = part + sg.src_off + s0, *plist = olist - ZKE_SEGMENT;           // the segment before mine is a whole one, and the one before me in the list
    const uint32_t *oo = poff + (size_t)seg * (ZKD_NBMAX + 1), *po = oo - (ZKD_NBMAX + 1);
    if (tid < NCHUNK) count[tid] = 0;
    ZKE_CLK_BEGIN();
    for (uint32_t pass = 0; pass < (1u << (dlog - PLOG)); pass++) {
        for (uint32_t i = tid; i < PSLOTS; i += 1024) { first[i] = ZKE_DENSE_NONE; last[i] = 0; }
        __syncthreads();
        // (E entries requested per lane before the first is used: a pass's lists are ~16 entries per lane, so usually all of them)
        if (s0) {
            const uint32_t *l = plist + po[pass], n = po[pass + 1] - po[pass];
            for (uint32_t i0 = tid; i0 < n; i0 += 1024 * E) {
                uint32_t e[E];
#pragma unroll
                for (uint32_t u = 0; u < E; u++) e[u] = l[i0 + 1024 * u < n ? i0 + 1024 * u : i0];
#pragma unroll
                for (uint32_t u = 0; u < E; u++) if (i0 + 1024 * u < n) atomicMax(&last[e[u] & (PSLOTS - 1)], (e[u] >> PLOG) + 1);
            }
        }
        const uint32_t *l = olist + oo[pass], n = oo[pass + 1] - oo[pass];
        for (uint32_t i0 = tid; i0 < n; i0 += 1024 * E) {
            uint32_t e[E];
#pragma unroll
            for (uint32_t u = 0; u < E; u++) e[u] = l[i0 + 1024 * u < n ? i0 + 1024 * u : i0];
#pragma unroll
            for (uint32_t u = 0; u < E; u++) if (i0 + 1024 * u < n) atomicMin(&first[e[u] & (PSLOTS - 1)], e[u] >> PLOG);
        }
        __syncthreads();
        ZKE_CLK(13);
        // What a pass finds is NOT stored by position -- a line of `cand` would be written an eighth at a time, pass after pass -- but appended
        // to the list of the position's chunk, which lives where the chunk's entries of `cand` will be (a chunk has at most as many candidates
        // as positions): whole lines, filled front to back.
        for (uint32_t i0 = tid; i0 < n; i0 += 1024 * E) {
            uint32_t e[E];
#pragma unroll
            for (uint32_t u = 0; u < E; u++) e[u] = l[i0 + 1024 * u < n ? i0 + 1024 * u : i0];
#pragma unroll
            for (uint32_t u = 0; u < E; u++) {
                if (i0 + 1024 * u >= n) continue;
                const uint32_t r = e[u] >> PLOG, m1 = first[e[u] & (PSLOTS - 1)], m2 = last[e[u] & (PSLOTS - 1)];
                uint32_t d = 0;
                if (m1 < r && r - m1 > ZKE_WINDOW) d = r - m1;                             // (the slot holds a position: mine at the latest)
                else if (m2 && r + ZKE_SEGMENT - (m2 - 1) > ZKE_WINDOW) d = r + ZKE_SEGMENT - (m2 - 1);
                const uint32_t chunk = r >> CLOG;
                if (d) out[s0 + (chunk << CLOG) + atomicAdd(&count[chunk], 1u)] = (r & (CHUNK - 1)) | ((d - ZKE_WINDOW - 1) << CLOG);
            }
        }
        __syncthreads();
        ZKE_CLK(14);
    }
    // Chunk by chunk: the list into LDS by position, then every position's candidate measured -- 16 bytes at the position against 16
    // bytes `distance` before it, every lane at work -- and the chunk's entries of `cand` written in whole lines over its list.
    uint32_t *dist = first;
    for (uint32_t c = 0; (c << CLOG) < sg.d_size; c++) {
        const uint32_t cb = s0 + (c << CLOG);
        for (uint32_t i = tid; i < CHUNK; i += 1024) dist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < count[c]; i += 1024) { const uint32_t e = out[cb + i]; dist[e & (CHUNK - 1)] = (e >> CLOG) + ZKE_WINDOW + 1; }
        __syncthreads();
        for (uint32_t q0 = cb + 4 * tid; q0 < cb + CHUNK && q0 < e1; q0 += 4096) {
            uint32_t d[4];
#pragma unroll
            for (int k = 0; k < 4; k++) d[k] = dist[q0 - cb + k];
            uint32_t own[5] = {0, 0, 0, 0, 0};
            if (q0 + 20 <= fsz) memcpy(own, frame + q0, 20);
            else for (uint32_t b = q0; b < fsz; b++) own[(b - q0) >> 2] |= (uint32_t)frame[b] << (8 * ((b - q0) & 3));     // the frame's last bytes: zeros behind them
            // ... and the four bytes in front of both (the catch-up count of the entry: how many of them agree, not past the frame's first byte)
            uint32_t before = 0;                                              // bytes q0 - 4 .. q0 - 1 (zeros in front of the frame)
            if (q0 >= 4) memcpy(&before, frame + q0 - 4, 4);
            uint32_t cc[4][5];                                                // the candidate's bytes e - 4 .. e + 15
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t e = d[k] ? q0 + k - d[k] : 4u;                     // (no candidate: the frame's first bytes, unused)
                if (e >= 4) memcpy(cc[k], frame + e - 4, 20);
                else { cc[k][0] = 0; for (uint32_t b = 0; b < e; b++) cc[k][0] |= (uint32_t)frame[b] << (8 * (4 - e + b)); memcpy(&cc[k][1], frame + e, 16); }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t l = zke_first16(cc[k][1] ^ __builtin_amdgcn_alignbyte(own[1], own[0], (uint32_t)k), cc[k][2] ^ __builtin_amdgcn_alignbyte(own[2], own[1], (uint32_t)k),
                                               cc[k][3] ^ __builtin_amdgcn_alignbyte(own[3], own[2], (uint32_t)k), cc[k][4] ^ __builtin_amdgcn_alignbyte(own[4], own[3], (uint32_t)k));
                const uint32_t mine4 = k ? __builtin_amdgcn_alignbyte(own[0], before, (uint32_t)k) : before;      // the four bytes that end at q0 + k
                uint32_t bk = ZKE_FFBH(mine4 ^ cc[k][0]);
                bk = (bk < 32u ? bk : 32u) >> 3;
                const uint32_t e = q0 + k - d[k];
                bk = bk < e ? bk : e;
                d[k] = d[k] && l >= ZKE_DENSE_MIN ? l | (bk << 5) | (d[k] << 8) : 0u;
            }
            if (q0 + 4 <= e1) memcpy(out + q0, d, 16);
            else for (uint32_t k = 0; q0 + k < e1; k++) out[q0 + k] = d[k];
        }
        __syncthreads();
    }
    ZKE_CLK(15);
    ZKE_CLK_END();
}
void zk_launch_enc_dense_cand(hipStream_t st, const uint8_t *src, const ZkEncFrame *segs, uint32_t nsegs, const ZkEncLdm &ldm, uint32_t *cand, uint32_t *part, uint32_t *poff)
{
    if (!nsegs) return;
    hipLaunchKernelGGL(zk_k_enc_dense_part, dim3(nsegs), dim3(1024), 0, st, src, segs, ldm, part, poff);
    hipLaunchKernelGGL(zk_k_enc_dense_cand, dim3(nsegs), dim3(1024), 0, st, src, segs, ldm, (const uint32_t *)part, (const uint32_t *)poff, cand);
}
// (both return the HIP verdict of clearing the table: a table that was not cleared holds an earlier call's positions, and the matcher
//  would follow them -- ADVICE r4 / r5)
int zk_launch_enc_ldm_build_frames(hipStream_t st, const uint8_t *src, const ZkEncLdm &ldm, uint32_t *table, uint32_t nframes)
{
    if (hipMemsetAsync(table, 0xFF, ((size_t)nframes * sizeof(uint32_t)) << ldm.log, st) != hipSuccess) return -1;
    const uint64_t wgs = ((uint64_t)ldm.frame_size + 4095) / 4096;
    // a grid's y dimension ends at 65535 (4 GiB of 64 KiB frames are 65536 of them): frames in launches of at most that many
    for (uint32_t f0 = 0; f0 < nframes; f0 += 65535u) {
        const uint32_t cnt = nframes - f0 < 65535u ? nframes - f0 : 65535u;
        hipLaunchKernelGGL(zk_k_enc_ldm_build_frames, dim3((uint32_t)(wgs < 1024 ? wgs : 1024), cnt), dim3(256), 0, st, src, ldm, table, f0);
    }
    return 0;
}
int zk_launch_enc_ldm_build(hipStream_t st, const ZkEncLdm &ldm, uint32_t *table)
{
    if (hipMemsetAsync(table, 0xFF, sizeof(uint32_t) << ldm.log, st) != hipSuccess) return -1;
    const uint64_t n = ldm.plen - ldm.u0, wgs = (n + 1023) / 1024;
    hipLaunchKernelGGL(zk_k_enc_ldm_build, dim3((uint32_t)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, st, ldm, table);
    return 0;
}
void zk_launch_enc_match(hipStream_t st, const uint8_t *src, const ZkEncFrame *segs, uint32_t nsegs, ZkEncBlock *blocks, uint64_t *seqs, uint8_t *lits, int level, const ZkEncLdm &ldm)
{
    if (!nsegs) return;
    // one instance per setting of zk_enc_device.h (zke_hash_log / zke_lazy / zke_step), with and without long-distance matching
#define ZKE_GO(H, L, S, D) hipLaunchKernelGGL((zk_k_enc_match<H, L, S, D>), dim3(nsegs), dim3(ZKE_THREADS), 0, st, src, segs, blocks, seqs, lits, ldm)
    if (ldm.dense) {                                            // (level 0 / >= 3 in frame: never the fast setting)
        if (zke_step(level) == 1024) hipLaunchKernelGGL((zk_k_enc_match<15, 1, 1024, true, true>), dim3(nsegs), dim3(ZKE_THREADS), 0, st, src, segs, blocks, seqs, lits, ldm);
        else hipLaunchKernelGGL((zk_k_enc_match<15, 1, 4096, true, true>), dim3(nsegs), dim3(ZKE_THREADS), 0, st, src, segs, blocks, seqs, lits, ldm);
    }
    else if (ldm.table) {
        if (zke_fast(level)) ZKE_GO(14, 0, 4096, true);
        else if (zke_step(level) == 1024) ZKE_GO(15, 1, 1024, true);
        else ZKE_GO(15, 1, 4096, true);
    }
    else if (zke_fast(level)) hipLaunchKernelGGL((zk_k_enc_match2<14>), dim3(nsegs), dim3(ZKE_THREADS), 0, st, src, segs, blocks, seqs, lits);   // zke_fast2(): the plan set minmatch 5
    else if (zke_step(level) == 1024) ZKE_GO(15, 1, 1024, false);
    else ZKE_GO(15, 1, 4096, false);
#undef ZKE_GO
}
void zk_launch_enc_fse_build(hipStream_t st, const uint8_t *src, const ZkEncFrame *frames, uint32_t nframes, const ZkEncBlock *blocks, uint64_t *seqs, uint32_t *mpos,
                             const ZkEncTables *predef, ZkEncTables *ftab)
{
    (void)src;
    hipLaunchKernelGGL(zk_k_enc_fse_build, dim3(nframes), dim3(1024), 0, st, frames, blocks, seqs, mpos, predef, ftab, ZKE_FSE_MIN_SEQ);
}
void zk_launch_enc_entropy(hipStream_t st, const uint8_t *src, const ZkEncFrame *frames, ZkEncBlock *blocks, uint32_t nblocks,
                           uint64_t *seqs, uint32_t *mpos, const uint8_t *lits, uint8_t *scratch, const ZkEncTables *ftab)
{
    if (!nblocks) return;
    hipLaunchKernelGGL(zk_k_enc_entropy, dim3((nblocks + ZKE_ENT_BLOCKS - 1) / ZKE_ENT_BLOCKS), dim3(ZKE_ENT_THREADS), sizeof(ZkeEntShared), st, src, frames, blocks, nblocks, seqs, mpos, lits, scratch, ftab);
}
void zk_launch_enc_sizes(hipStream_t st, const ZkEncFrame *frames, uint32_t nframes, ZkEncBlock *blocks, const ZkEncTables *ftab, int checksum,
                         uint64_t *c_size64, uint32_t *c_sizes, uint32_t *d_sizes)
{
    hipLaunchKernelGGL(zk_k_enc_sizes, dim3((nframes + 63) / 64), dim3(64), 0, st, frames, nframes, blocks, ftab, checksum, c_size64, c_sizes, d_sizes);
}
void zk_launch_scan64(hipStream_t st, const uint64_t *in, uint32_t n, uint64_t *out)
{
    hipLaunchKernelGGL(zk_k_scan64, dim3(1), dim3(1024), 0, st, in, n, out);
}
void zk_launch_enc_assemble(hipStream_t st, const uint8_t *src, const ZkEncFrame *frames, uint32_t nframes, const ZkEncBlock *blocks, uint32_t nblocks, const ZkEncTables *ftab,
                            const uint8_t *lits, const uint8_t *scratch, const uint64_t *out_off, const uint64_t *c_size64, const uint64_t *hashes, int checksum, uint8_t *dst)
{
    hipLaunchKernelGGL(zk_k_enc_assemble, dim3(nframes + nblocks), dim3(256), 0, st, src, frames, nframes, blocks, ftab, lits, scratch, out_off, c_size64, hashes, checksum, dst);
}
