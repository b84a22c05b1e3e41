// zk_decode.hip -- gfx950 kernels of the batched seekable-zstd frame decoder.
//
// Replaces, for N frames at a time, what the reference does one 128 KiB step at a time
// through ZSTD_decompressStream (lib/src/decode.rs:242-256; libzstd 1.5.7 underneath).
//
// Pipeline (data resident in HBM; huf and fse run side by side on two queues, the rest in order):
//   zk_k_walk (count)  one lane per frame: frame header + block chain -> ZkFrameInfo
//   zk_k_scan          exclusive prefix sums over frames -> ZkFrameBase, totals
//   zk_k_walk (fill)   one lane per frame -> ZkBlock[] (block list, table inheritance)
//   zk_k_huf           wave 0: one lane per Huffman stream (4 per block), tables in an LDS pool sized by tree depth;
//                      wave 1: companion lanes (touch the stream ahead, store the decoded packs) -> literal scratch
//   zk_k_fse_predef    blocks with predefined tables: one lane per block, 64 blocks per wave in lock step, shared
//                      tables, cooperative 64-B record stores -> ZkSeq[] (+ out_size, symbolic reps)
//   zk_k_fse_quad      blocks with their own tables: FSE tables in LDS, a quad of lanes per block (one lane per state
//                      machine, DPP exchanges), a toucher wave for the bitstreams -> ZkSeq[]
//   zk_k_fse           the same with one lane per block (the form the CPU simulation of tests/sim mirrors; selectable)
//   zk_k_exec          one workgroup per frame: byte-parallel sequence execution through a per-byte source map,
//                      coalesced 16 B stores; optional raw-content prefix before the frame
//   zk_k_xxh64         one wave per frame (4 accumulator chains), verifies Content_Checksum
//
// No MFMA anywhere: this is byte-serial entropy decoding; the roofline is HBM (c_i + d_i bytes per frame).
#include <hip/hip_runtime.h>
#include "zk_device.h"
#include "zk_kernels.h"

// ------------------------------------------------------------------------------------------------ walk
// ids != nullptr: frame f of the batch is frame ids[f] of the archive (random-access batches: many seeks per submission)
// comp_size / dst_cap (the sizes the caller vouches for): a frame whose compressed range leaves the buffer, or whose output
// range leaves the destination, is flagged (srcSize_wrong / dstSize_tooSmall) and contributes no work -- offset tables
// that do not come from a SeekTable (arbitrary caller arrays) cannot drive reads or writes out of bounds.
__global__ __launch_bounds__(64) void zk_k_walk(const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off, const uint64_t *d_off,
                                                uint32_t first, uint32_t count, const uint32_t *ids, const uint64_t *out_off, uint64_t dst_cap,
                                                const ZkFrameBase *bases, ZkBlock *blocks, ZkFrameInfo *infos)
{
    uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= count) return;
    const uint32_t id = ids ? ids[f] : first + f;
    uint64_t cb = c_off[id], ce = c_off[id + 1];
    // d_off == nullptr (zk_frame_content_sizes): nobody knows the frames' sizes yet, nothing is written
    uint64_t dsz = d_off ? d_off[id + 1] - d_off[id] : ZK_SIZE_UNKNOWN;
    ZkFrameInfo fi;
    if (!bases && !d_off) {
        if (ce < cb || ce > comp_size) { fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; fi.status = ZK_E_SRC_SIZE_WRONG; fi.checksum_flag = 0; fi.checksum = 0; fi.window = 0; fi.n_own_tables = 0; fi.fcs = ZK_SIZE_UNKNOWN; }
        else zk_walk_frame(comp, cb, ce, dsz, f, nullptr, nullptr, fi);
        if (fi.status != ZK_OK) { fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; }
        infos[f] = fi;
    } else if (!bases) {                            // pass 1: count
        const uint64_t o = out_off ? out_off[f] : d_off[id] - d_off[first];
        if (ce < cb || ce > comp_size || d_off[id + 1] < d_off[id]) {
            fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; fi.status = ZK_E_SRC_SIZE_WRONG;
            fi.checksum_flag = 0; fi.checksum = 0; fi.window = 0; fi.n_own_tables = 0; fi.fcs = ZK_SIZE_UNKNOWN;
            infos[f] = fi;
            return;
        }
        zk_walk_frame(comp, cb, ce, dsz, f, nullptr, nullptr, fi);
        if (fi.status == ZK_OK && (d_off[id] < d_off[first] && !out_off)) fi.status = ZK_E_SRC_SIZE_WRONG;
        if (fi.status == ZK_OK && (o > dst_cap || dsz > dst_cap - o)) fi.status = ZK_E_DST_TOO_SMALL;
        if (dsz > ZK_MAX_FRAME && fi.status == ZK_OK) fi.status = ZK_E_FRAMEPARAM_UNSUPPORTED;
        if (fi.status != ZK_OK) { fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; }   // contributes no work
        infos[f] = fi;
    } else if (infos[f].status == ZK_OK) {          // pass 2: fill the block list
        zk_walk_frame(comp, cb, ce, dsz, f, &bases[f], blocks, fi);
    }
}

// exclusive scan of (n_blocks, n_seq, lit_bytes) over frames + the count of blocks with their own sequence tables;
// one workgroup.  totals: [0..2] the three sums, [4] that count ([3] is the first-error word of zk_k_status)
__global__ __launch_bounds__(1024) void zk_k_scan(const ZkFrameInfo *infos, uint32_t count, ZkFrameBase *bases, uint64_t *totals, const uint64_t *d_off, uint32_t first, const uint64_t *out_off)
{
    // [5]: output bytes of the batch (what the frames claim): the host sizes nothing from it, it only tells dense sequence streams from sparse ones
    // [7]: the largest frame (what it claims to decode to): how many segments a frame can have (zk_k_exec_seg's grid)
    if (threadIdx.x == 0) totals[5] = out_off ? out_off[count] : d_off ? d_off[first + count] - d_off[first] : 0;
    __shared__ uint64_t wsum[16][4];
    __shared__ uint64_t carry[4];
    __shared__ unsigned long long s_maxd;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 4) carry[tid] = 0;
    if (tid == 0) s_maxd = 0;
    __syncthreads();
    // [8]: frames that carry a Content_Checksum (none: no checksum kernel has anything to do)
    __shared__ uint32_t s_ncks;
    if (tid == 0) s_ncks = 0;
    __syncthreads();
    {
        unsigned long long mx = 0;
        uint32_t ncks = 0;
        for (uint32_t f = tid; f < count; f += 1024) {
            const unsigned long long dsz = out_off ? out_off[f + 1] - out_off[f] : d_off ? d_off[first + f + 1] - d_off[first + f] : 0;
            mx = dsz > mx ? dsz : mx;
            ncks += infos[f].status == ZK_OK && infos[f].checksum_flag ? 1u : 0u;
        }
        if (mx) atomicMax(&s_maxd, mx);
        if (ncks) atomicAdd(&s_ncks, ncks);
    }
    for (uint32_t base = 0; base < count; base += 1024) {
        uint32_t f = base + tid;
        uint64_t v[4] = {0, 0, 0, 0};
        if (f < count) { v[0] = infos[f].n_blocks; v[1] = infos[f].n_seq; v[2] = infos[f].lit_bytes; v[3] = infos[f].n_own_tables; }
        uint64_t inc[4];
        for (int k = 0; k < 4; k++) {
            uint64_t x = v[k];
            for (int d = 1; d < 64; d <<= 1) { uint64_t y = __shfl_up(x, d, 64); if ((int)lane >= d) x += y; }
            inc[k] = x;
            if (lane == 63) wsum[wave][k] = x;
        }
        __syncthreads();
        uint64_t pre[4];
        for (int k = 0; k < 4; k++) {
            uint64_t s = carry[k];
            for (uint32_t w = 0; w < wave; w++) s += wsum[w][k];
            pre[k] = s;
        }
        if (f < count) {
            bases[f].block_base = pre[0] + inc[0] - v[0];
            bases[f].seq_base = pre[1] + inc[1] - v[1];
            bases[f].lit_base = pre[2] + inc[2] - v[2];
        }
        __syncthreads();
        if (tid == 1023) for (int k = 0; k < 4; k++) carry[k] = pre[k] + inc[k];
        __syncthreads();
    }
    if (tid < 3) totals[tid] = carry[tid];
    if (tid == 3) totals[4] = carry[3];
    if (tid == 4) totals[7] = s_maxd;
    if (tid == 5) totals[8] = s_ncks;
}

// ------------------------------------------------------------------------------------------------ Huffman literals
// 64 lanes = 16 blocks x 4 streams.  The decode tables live in a 16 KiB LDS pool that is carved up by each tree's
// depth (2^maxbits u16 cells): sixteen 9-bit tables fit at once, deeper trees take the blocks in several passes.
// A workgroup therefore needs ~21 KiB of LDS and 7 of them share a CU; the chain per symbol
// (shift -> LDS cell -> shift) is latency bound, so lanes in flight per CU is what sets the speed.
constexpr int ZK_HUF_BLOCKS = 16;
constexpr uint32_t ZK_HUF_POOL = 8192;           // u16 cells
constexpr int32_t ZK_HUF_AHEAD = 192;            // bytes the companion wave stays ahead of a stream's read position
static_assert(ZK_HUF_POOL >= 2048, "a maximum-depth table must fit");
static_assert(ZK_HUF_BLOCKS * sizeof(ZkHufTmp) <= ZK_HUF_POOL * sizeof(uint16_t), "parse scratch aliases the pool");

// Companion of one decoding lane (ZkHufMail): stores the lane's packs to HBM and touches the 128-B lines of the
// lane's (backward read) stream ZK_HUF_AHEAD bytes before the decoder gets there, so that the decoder's in-order
// load queue only ever sees L2 hits.  Every wave of 64 streams crosses ~6 new lines per 8 symbols; without this
// each crossing stalls the whole wave for an HBM miss.  The touches are fire-and-forget loads (inline asm into a
// register nobody reads), so this wave never waits on memory and keeps up with 64 decoders.
__device__ void zk_huf_companion(const uint8_t *base, uint32_t len, uint8_t *dst, volatile ZkHufMail *mail, uint32_t l,
                                 const volatile uint32_t *done)
{
    const uintptr_t lo = (uintptr_t)base;
    uintptr_t line = ((uintptr_t)base + len) & ~(uintptr_t)127;      // lowest line touched so far (the decoder's init loads cover it)
    uint32_t stored = 0, sink = 0;
    // bursts start on 32-byte boundaries of the literal scratch: the first one only reaches the boundary (0..3 packs).  A burst that
    // straddles one is two partly written sectors for the L2 (round 3: 2.96 GiB written for 1.45 GB of literals)
    const uint32_t lead = (uint32_t)((0 - (uintptr_t)dst) & 31) >> 3;
    // The loop leaves as ONE wave, by a wave-uniform verdict: a pass in which `done` was read as set BEFORE every lane read its
    // state and no lane found anything left to do.  (Round 3's form -- `if (idle) { if (fin) break; sleep; }` per lane -- was
    // compiled into "idle lanes are parked until every lane of the wave is idle, then ALL of them leave if the LAST read of
    // `done` was set": a lane parked with 1..3 packs waiting for the decoder's end left on a `done` it had never read its state
    // behind whenever a neighbour was busy in the pass where `done` flipped, and its last packs never reached HBM -- stale
    // literal bytes, status OK.  DESIGN.md section 8, "the small-batch defect".)
    for (;;) {
        const uint32_t fin = __builtin_amdgcn_readfirstlane(zk_lds_ld<uint32_t>(done));     // read BEFORE the states: a pack published before `done` is seen
        const uint32_t st = zk_lds_ld<uint32_t>(&mail->state[l]);
        const uint32_t written = st & 0x3fffu;
        bool busy = false;
        const uint32_t avail = (written - stored) & 0x3fffu;
        const uint32_t want_nb = stored < lead ? lead - stored : ZK_HUF_BURST;
        if (avail >= want_nb || (fin && avail)) {                    // whole bursts while the decoder runs, the rest at its end
            const uint32_t nb = avail < want_nb ? avail : want_nb;
            for (uint32_t k = 0; k < nb; k++) {
                const uint64_t pack = zk_lds_ld<uint64_t>(&mail->pack[(stored + k) % ZK_HUF_RING][l]);
                memcpy(dst + (size_t)(stored + k) * 8, &pack, 8);
            }
            stored += nb;
            zk_lds_st<uint32_t>(&mail->consumed[l], stored);
            busy = true;
        }
        const int32_t want = (int32_t)(st >> 14) - 64 - ZK_HUF_AHEAD;
        const uintptr_t wp = (lo + (uintptr_t)(want < 0 ? 0 : want)) & ~(uintptr_t)127;
        if (line > wp && line > lo) {
            line -= 128;
            const uint8_t *a = reinterpret_cast<const uint8_t *>(line < lo ? lo : line);
            asm volatile("global_load_ubyte %0, %1, off" : "+v"(sink) : "v"(a) : "memory");
            busy = true;
        }
        if (__ballot(busy) == 0) {                                   // (the lanes of this call: the wave's active ones)
            if (fin) break;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" :: "v"(sink));
}

// The body of the kernel for block group `group` (16 blocks); also called in a loop by zk_k_small_entropy.
__device__ __forceinline__ void zk_huf_group(uint32_t group, const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, uint8_t *lit_scratch)
{
    __shared__ __attribute__((aligned(16))) uint16_t pool[ZK_HUF_POOL];
    __shared__ ZkHufHdr hdr[ZK_HUF_BLOCKS];
    __shared__ uint32_t s_maxbits[ZK_HUF_BLOCKS], s_desc[ZK_HUF_BLOCKS], s_n[ZK_HUF_BLOCKS];
    __shared__ ZkHufMail mail;
    __shared__ uint32_t s_done;
    // wave 0 decodes (lane = block slot x 4 streams), wave 1 mirrors it lane for lane and touches cache lines ahead.
    // A workgroup with fewer than 4 blocks left keeps >= 16 lanes active: the extra lanes shadow valid streams
    // (< 16 active lanes run ~3x slower on gfx950, tools/ubench/lat3.hip); shadows never store to HBM
    const uint32_t t = threadIdx.x & 63;
    const bool decoder = threadIdx.x < 64, companion = threadIdx.x >= 64 && threadIdx.x < 128;     // (further waves of a caller's workgroup only pass the barriers)
    const uint32_t wb = group * ZK_HUF_BLOCKS;
    const uint32_t nvalid = nblocks - wb < (uint32_t)ZK_HUF_BLOCKS ? nblocks - wb : (uint32_t)ZK_HUF_BLOCKS;
    const bool real = t < 4 * nvalid;
    const uint32_t lane = real ? t : (t < 16 ? t % (4 * nvalid) : t);
    const uint32_t slot = lane >> 2, stream = lane & 3;
    const uint32_t bi = wb + slot;
    bool active = false;
    ZkBlock b;
    if (bi < nblocks) {
        b = blocks[bi];
        active = b.type == 2 && b.lit_type >= 2 && b.status == ZK_OK;
    }
    if (decoder && stream == 0) {
        uint32_t r = 0, mb = 0, n = 0;
        if (active) {
            const ZkBlock &def = blocks[b.huf_def];
            r = zk_huf_read_weights(comp + def.src + def.lit_off, def.lit_comp, &hdr[slot],
                                    reinterpret_cast<ZkHufTmp *>(pool) + slot, &n, &mb);
        }
        s_desc[slot] = r; s_maxbits[slot] = mb; s_n[slot] = n;
    }
    __syncthreads();
    // pool layout: blocks in slot order, a new pass whenever the next table does not fit
    uint32_t my_pass = 0, my_at = 0, npass;
    {
        uint32_t pass = 0, acc = 0;
        for (uint32_t j = 0; j < (uint32_t)ZK_HUF_BLOCKS; j++) {
            const uint32_t sz = s_desc[j] ? 1u << s_maxbits[j] : 0u;
            if (acc + sz > ZK_HUF_POOL) { pass++; acc = 0; }
            if (j == slot) { my_pass = pass; my_at = acc; }
            acc += sz;
        }
        npass = pass + 1;
    }
    const uint32_t desc = s_desc[slot], mb = s_maxbits[slot];
    bool ok = desc != 0;
    uint16_t *tab = pool + my_at;
    for (uint32_t p = 0; p < npass; p++) {
        const bool mine = active && ok && my_pass == p;
        if (decoder && mine && stream == 0) zk_huf_fill_table(tab, &hdr[slot], s_n[slot], mb);
        // this lane's stream
        const uint8_t *sbase = nullptr;
        uint8_t *sdst = nullptr;
        uint32_t slen = 0, sn = 0;
        bool have = false;
        if (mine) {
            const uint8_t *pay = comp + b.src + b.lit_off;
            uint32_t size = b.lit_comp;
            if (b.lit_type == 2) { pay += desc; size -= desc; }      // own tree description precedes the streams
            uint8_t *dst = lit_scratch + b.lit_base;
            const uint32_t regen = b.lit_regen;
            if (b.lit_streams == 1) {
                if (stream == 0) { sbase = pay; slen = size; sdst = dst; sn = regen; have = true; }
            } else if (size < 6) {
                ok = false;
            } else {
                uint32_t s1 = zk_rd16(pay), s2 = zk_rd16(pay + 2), s3 = zk_rd16(pay + 4);
                uint32_t q = (regen + 3) / 4;
                if (6 + s1 + s2 + s3 > size || 3 * q > regen) ok = false;
                else {
                    uint32_t s4 = size - 6 - s1 - s2 - s3;
                    uint32_t start = 6 + (stream > 0 ? s1 : 0) + (stream > 1 ? s2 : 0) + (stream > 2 ? s3 : 0);
                    slen = stream == 0 ? s1 : stream == 1 ? s2 : stream == 2 ? s3 : s4;
                    sn = stream == 3 ? regen - 3 * q : q;
                    sbase = pay + start; sdst = dst + stream * q; have = true;
                }
            }
        }
        if (decoder) { mail.state[t] = zk_huf_mail_state(0, (int32_t)slen - 40); mail.consumed[t] = 0; }
        if (threadIdx.x == 0) s_done = 0;
        __syncthreads();
        if (decoder) {
            if (have) ok = zk_huf_decode_stream(tab, mb, sbase, slen, sdst, sn, real, &mail, t);
            zk_lds_st<uint32_t>(&s_done, 1u);
        } else if (companion && have && real) {
            zk_huf_companion(sbase, slen, sdst + ((0 - (uintptr_t)sdst) & 7), &mail, t, &s_done);     // packs start at the 8-byte aligned output position
        }
        __syncthreads();
    }
    if (decoder && active && !ok && real) blocks[bi].status = ZK_E_CORRUPTION;
}
__global__ __launch_bounds__(128) void zk_k_huf(const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, uint8_t *lit_scratch)
{
    zk_huf_group(blockIdx.x, comp, blocks, nblocks, lit_scratch);
}

// ------------------------------------------------------------------------------------------------ FSE sequences
// One lane per block, the block's LL / ML (2^9) and OF (2^8) tables + build scratch / record ring in LDS.  A block's
// sequences are one serial chain (cells -> bit counts -> next states) of ~0.5 us per sequence, so the kernel's speed
// is blocks in flight x instructions per sequence, and the table footprint sets the former:
//   ZkCells32, 5.25 KiB per block: 28 blocks per CU on 4 waves (one per SIMD) -- fewest instructions per sequence;
//   ZkCells16, 2.75 KiB per block: 56 blocks per CU on 8 waves -- ~18 more instructions per sequence and the value
//              table's bit count on the chain (+50% per lane), twice the lanes: wins once the blocks no longer fit in
//              one round of the 32-bit layout (measured: 4 GiB of libzstd frames 24.0 -> 20.8 ms, 256 MiB 4.6 -> 6.9).
template <typename CP, int ZK_FSE_BLOCKS, int ZK_FSE_WAVES>
__global__ __launch_bounds__(64 * ZK_FSE_WAVES) void zk_k_fse(const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, ZkSeqP *seqs)
{
    constexpr int ZK_FSE_PER_WAVE = ZK_FSE_BLOCKS / ZK_FSE_WAVES;
    __shared__ ZkSeqTablesT<CP> T[ZK_FSE_BLOCKS];
    __shared__ uint32_t llv[36], mlv[53];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const uint32_t ll_init[36] = ZK_LL_TABLE;
        const uint32_t ml_init[53] = ZK_ML_TABLE;
        if (tid < 36) llv[tid] = ll_init[tid];
        if (tid < 53) mlv[tid] = ml_init[tid];
    }
    __syncthreads();
    // gfx950 runs a wave with fewer than 16 active lanes ~3x slower per dependent instruction (measured:
    // tools/ubench/lat3.hip), so lanes 7..20 shadow lanes 0..6: same block, same LDS tables (broadcast
    // reads), identical LDS writes, no HBM writes.
    if (lane >= 3 * ZK_FSE_PER_WAVE) return;
    const uint32_t wbase = blockIdx.x * ZK_FSE_BLOCKS + wave * ZK_FSE_PER_WAVE;        // first block of this wave
    if (wbase >= nblocks) return;
    const uint32_t nvalid = nblocks - wbase < (uint32_t)ZK_FSE_PER_WAVE ? nblocks - wbase : (uint32_t)ZK_FSE_PER_WAVE;
    const bool real = lane < nvalid;
    const uint32_t slot = wave * ZK_FSE_PER_WAVE + lane % nvalid;                      // shadows replicate valid blocks only
    const uint32_t bi = blockIdx.x * ZK_FSE_BLOCKS + slot;
    ZkBlock b = blocks[bi];
    if (b.type != 2 || b.nseq == 0 || b.status != ZK_OK || b.pad) return;      // pad: done by a shared-table kernel.  (All-predefined blocks that kernel left behind -- a workgroup whose reference
                                                                               // had own tables too few neighbours share -- are taken here like any other)
    zk_decode_sequences<ZkRevU, CP>(comp, blocks, b, &T[slot], seqs + b.seq_base, llv, mlv, real);
    if (!real) return;
    ZkBlock *o = &blocks[bi];
    o->out_size = b.out_size;
    o->rep_out[0] = b.rep_out[0]; o->rep_out[1] = b.rep_out[1]; o->rep_out[2] = b.rep_out[2];
    if (b.status != ZK_OK) o->status = b.status;      // (never OK over the literal kernel's verdict: the two run side by side)
}

// ---- three lanes per block ------------------------------------------------------------------------------------
// zk_k_fse spends a full wave instruction per step of each of a block's three state machines although only 7 lanes of
// the wave hold a block: the kernel is VALU bound with 11 % of the lanes doing anything.  Here a block takes a QUAD of
// lanes: lane 0 walks the literal-length table, lane 1 the offset table, lane 2 the match-length table (lane 3 is
// idle).  Each lane decodes one cell, the three (value bits, state bits) pairs are exchanged inside the quad with DPP
// quad_perm moves (no LDS, no latency to speak of), every lane shifts the shared 64-bit window to its own two fields,
// and the three values are exchanged again so that all lanes keep the repeat-offset history and the running sums in
// step.  The step costs ~110 wave instructions instead of ~250.  Same LDS tables, same records, same results.
// The walk itself is zk_seq_walk_quad in zk_device.h (tests/sim runs it on the CPU, three lock-stepped fibers per block).
struct ZkQuadDpp {
    static __device__ __forceinline__ uint32_t bcast(uint32_t v, int k)         // value of quad lane k in every lane of the quad
    {
        return k == 0 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true)
             : k == 1 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, true)
             : k == 2 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, true)
                      : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, true);
    }
};

// Round 4: the walk is two waves.  A lone wave issues in order, so every instruction of a step -- repeat offsets, running sums,
// the record, its store -- used to sit on the chain's clock (~110 instructions, ~1000 clocks per sequence).  Now the WALKER wave
// (a quad per block, as before) only advances the three states and leaves each sequence's three values in an LDS ring; the
// FINISHER wave beside it (same lane layout: quad q = block q of the walker wave, lane j = the j-th sequence of a round of four)
// turns them into records.  Hand-over per walker wave: the walker publishes "sequences < prod are in the ring" in front of every
// fourth sequence and ZK_QUAD_DONE at its end (LDS accesses of a wave are ordered: the ring writes are in front of it); the
// finisher publishes "sequences < cons are read" as soon as a round's values are in its registers, and the walker does not write
// sequence i before cons >= i + 4 - ZK_QUAD_RING.  Both loops are wave-uniform (one counter per wave, a polling loop nobody
// leaves alone -- see zk_huf_companion for what a per-lane exit cost).
constexpr uint32_t ZK_QUAD_RING = 8;                     // sequences per block in the ring (16 B each: the 128 B that held 16 records)
constexpr uint32_t ZK_QUAD_DONE = 0x7FFFFFFFu;
static_assert(ZK_QUAD_RING * 16 == sizeof(((ZkSeqTablesX16 *)nullptr)->ring), "the hand-over ring takes the record ring's place");
struct ZkQuadOut {                                       // OUT of zk_seq_walk_quad on the device
    uint32_t ring;                                       // LDS byte address of the lane's word of entry 0
    volatile uint32_t *prod, *cons, *pos_pub;
    bool lead;                                           // the LL lane publishes for the toucher
    __device__ __forceinline__ void gate(uint32_t i, uint32_t stream_pos)
    {
        zk_lds_st<uint32_t>(prod, i);
        if (lead) zk_lds_st<uint32_t>(pos_pub, stream_pos);
        if (i + ZK_QUAD_ROUND > ZK_QUAD_RING)
            while ((uint32_t)__builtin_amdgcn_readfirstlane(zk_lds_ld<uint32_t>(cons)) + ZK_QUAD_RING < i + ZK_QUAD_ROUND) __builtin_amdgcn_s_sleep(1);
    }
    __device__ __forceinline__ void put(uint32_t i0, uint32_t k, uint32_t v)     // sequence i0 + k; i0 a multiple of the round, k a constant
    {
        zk_lds_st_at<uint32_t>(ring + (i0 & (ZK_QUAD_RING - ZK_QUAD_ROUND)) * 16u + k * 16u, v);
    }
};

// One more wave per workgroup keeps the bitstreams warm: a wave waits for its loads in order, and with 14 streams per
// wave some lane crosses into a new cache line at almost every step -- without help every other step of the whole wave
// waits for an L2 round trip.  The walkers publish their stream position once per round; the toucher wave (its
// own load counter, results never used) requests the two lines below it.
// STAGE (small batches): the quad copies its block's bitstream (up to ZK_FSE_STAGE bytes) into LDS before the walk.
constexpr uint32_t ZK_FSE_STAGE = 3072;
template <typename CP, int ZK_FSE_BLOCKS, int ZK_FSE_WAVES, bool STAGE = false>
__device__ __forceinline__ void zk_fse_quad_group(uint32_t group, const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, ZkSeqP *seqs, uint32_t all_blocks)
{
    constexpr int PER_WAVE = ZK_FSE_BLOCKS / ZK_FSE_WAVES;
    static_assert(PER_WAVE * 4 <= 64 && ZK_FSE_BLOCKS <= 64, "a quad of lanes per block; a toucher lane per block");
    __shared__ ZkSeqTablesT<CP> T[ZK_FSE_BLOCKS];
    __shared__ __attribute__((aligned(16))) uint8_t s_bits[STAGE ? ZK_FSE_BLOCKS : 1][STAGE ? ZK_FSE_STAGE + 16 : 16];
    __shared__ uint32_t llv[36], mlv[53], ofv[32];
    __shared__ uint32_t s_pos[ZK_FSE_BLOCKS], s_wbad[ZK_FSE_BLOCKS], s_live;
    __shared__ uint32_t s_prod[ZK_FSE_WAVES], s_cons[ZK_FSE_WAVES], s_nloop[ZK_FSE_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const uint32_t ll_init[36] = ZK_LL_TABLE;
        const uint32_t ml_init[53] = ZK_ML_TABLE;
        if (tid < 36) llv[tid] = CP::kFat ? ll_init[tid] : ll_init[tid] & 0xFFFFFFu;      // baselines only: the walker's cells (sym | x) keep no extra-bit count, and
        if (tid < 53) mlv[tid] = CP::kFat ? ml_init[tid] : ml_init[tid] & 0xFFFFFFu;      // the walk forms the counts by arithmetic (zk_seq_walk_quad); 8-byte cells: built from both
        if (tid < 32) ofv[tid] = 1u << tid;
        if (tid < (uint32_t)ZK_FSE_BLOCKS) { s_pos[tid] = 0; s_wbad[tid] = 0; }
        if (tid < (uint32_t)ZK_FSE_WAVES) { s_prod[tid] = 0; s_cons[tid] = 0; s_nloop[tid] = 0; }
        if (tid == 0) s_live = 0;
    }
    __syncthreads();
    const bool toucher = wave == 2u * ZK_FSE_WAVES, finisher = !toucher && wave >= (uint32_t)ZK_FSE_WAVES;
    const uint32_t w = finisher ? wave - ZK_FSE_WAVES : wave;          // the walker wave this lane walks / finishes for
    const uint32_t t = lane & 3;
    const uint32_t slot = toucher ? lane : w * PER_WAVE + (lane >> 2);
    const uint32_t bi = group * ZK_FSE_BLOCKS + slot;
    ZkBlock b;
    bool valid = toucher ? lane < (uint32_t)ZK_FSE_BLOCKS : (lane < 4 * PER_WAVE && (finisher || t != 3));
    valid = valid && bi < nblocks;
    if (valid) {
        b = blocks[bi];
        valid = b.type == 2 && b.nseq != 0 && b.status == ZK_OK && b.pad == 0 && (b.seq_modes != 0 || all_blocks);     // pad: done by the shared-table kernel; all-predefined blocks are its job too, unless the batch is small
    }
    if (valid && !toucher && !finisher && t == ZK_TAB_LL) { atomicAdd(&s_live, 1u); atomicMax(&s_nloop[w], b.nseq); }
    __syncthreads();
    if (toucher) {
        if (!valid) return;
        uint32_t last = 0, sink = 0;
        while (zk_lds_ld<uint32_t>(&s_live)) {
            const uint32_t p = zk_lds_ld<uint32_t>(&s_pos[slot]);
            if (p != last) {
                last = p;
                const uint32_t lo = b.src + b.seq_off;                  // nothing of the stream lies below the sequence header
                const uint8_t *a0 = comp + (p > lo + 128 ? p - 128 : lo), *a1 = comp + (p > lo + 256 ? p - 256 : lo);
                asm volatile("global_load_ubyte %0, %1, off" : "+v"(sink) : "v"(a0) : "memory");
                asm volatile("global_load_ubyte %0, %1, off" : "+v"(sink) : "v"(a1) : "memory");
            } else __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(sink));
        return;
    }
    ZkSeqTablesT<CP> *Tb = &T[slot];
    if (finisher) {
        // ---- the records: lane (quad q, j) = sequence i0 + j of the walker wave's block q
        const uint32_t nloop = zk_lds_ld<uint32_t>(&s_nloop[w]);        // the walker wave's longest block (wave-uniform)
        const uint32_t nseq = valid ? b.nseq : 0u;
        ZkSeqCarry cy;
        zk_seq_carry_init(cy);
        const uint32_t ring = zk_lds_addr(Tb->ring);
        ZkSeqP *out = seqs + (valid ? b.seq_base : 0);
        for (uint32_t i0 = 0; i0 < nloop; i0 += ZK_QUAD_ROUND) {
            const uint32_t need = i0 + ZK_QUAD_ROUND < nloop ? i0 + ZK_QUAD_ROUND : nloop;
            for (;;) {                                                   // (wave-uniform: one counter, read by every lane alike)
                const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane(zk_lds_ld<uint32_t>(&s_prod[w]));
                if (p >= need) break;                                    // ZK_QUAD_DONE included
                __builtin_amdgcn_s_sleep(1);
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = zk_lds_ld_at<u32x4>(ring + ((i0 + t) & (ZK_QUAD_RING - 1)) * 16u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the values are in registers: the walker may have the entries back
            zk_lds_st<uint32_t>(&s_cons[w], i0 + ZK_QUAD_ROUND);
            const uint32_t nvalid = nseq > i0 ? (nseq - i0 < ZK_QUAD_ROUND ? nseq - i0 : ZK_QUAD_ROUND) : 0u;
            const ZkSeqP rec = zk_seq_finish_quad<ZkQuadDpp>(t, nvalid, v.x, v.y, v.z, cy, b.lit_regen);
            if (t < nvalid) out[i0 + t] = rec;
        }
        if (!valid || t != 0) return;
        for (;;) {                                                       // the walker's verdict is written in front of its DONE
            if ((uint32_t)zk_lds_ld<uint32_t>(&s_prod[w]) == ZK_QUAD_DONE) break;
            __builtin_amdgcn_s_sleep(1);
        }
        zk_seq_finish_block(b, cy, zk_lds_ld<uint32_t>(&s_wbad[slot]));
        ZkBlock *o = &blocks[bi];
        o->out_size = b.out_size;
        o->rep_out[0] = b.rep_out[0]; o->rep_out[1] = b.rep_out[1]; o->rep_out[2] = b.rep_out[2];
        if (b.status != ZK_OK) o->status = b.status;      // (never OK over the literal kernel's verdict: the two run side by side)
        return;
    }
    // ---- the walker wave.  The three lanes build the block's tables together (identical LDS writes: more active lanes, see above)
    uint32_t wbad = 0;
    if (valid) {
        uint32_t al[3], own = 0;
        bool ok = true;
        for (int u = 0; u < 3; u++) {
            const uint32_t m = (b.seq_modes >> (6 - 2 * u)) & 3;
            const ZkBlock &def = m == 3 ? blocks[b.tab_def[u]] : b;
            const int32_t r = zk_seq_table_setup<CP>(comp, def, u, Tb, &al[u], llv, mlv);
            if (r < 0) { ok = false; break; }
            if (m != 3) own += (uint32_t)r;
        }
        if (!ok) wbad = 1;
        else {
            const uint8_t *bits = nullptr;
            if (STAGE) {
                const uint32_t bs = b.seq_off + 1 + own;
                if (bs < b.bsize && b.bsize - bs <= ZK_FSE_STAGE) {
                    const uint32_t len = b.bsize - bs;
                    const uint8_t *g = comp + b.src + bs;
                    for (uint32_t o = t * 8; o < len + 8; o += 24) { const uint64_t wd = zk_ld64(g + o); memcpy(&s_bits[slot][o], &wd, 8); }
                    bits = s_bits[slot];
                }
            }
            __builtin_amdgcn_wave_barrier();
#ifndef ZK_QUAD_RD
#define ZK_QUAD_RD ZkRevU      // one unaligned 8-byte load per step.  (ZkRevA -- aligned words, three ahead, no wait for the step's own load -- is slower here: 12.9 vs 11.4 ms, its bookkeeping costs more than the load's latency)
#endif
            ZkQuadOut qo;
            qo.ring = zk_lds_addr(Tb->ring) + 4u * t;
            qo.prod = &s_prod[w]; qo.cons = &s_cons[w]; qo.pos_pub = &s_pos[slot]; qo.lead = t == ZK_TAB_LL;
            wbad = zk_seq_walk_quad<ZK_QUAD_RD, CP, ZkQuadDpp>(comp, b, b.seq_off + 1 + own, t,
                                     t == ZK_TAB_LL ? Tb->ll : t == ZK_TAB_OF ? Tb->of : Tb->ml,
                                     t == ZK_TAB_LL ? llv : t == ZK_TAB_OF ? ofv : mlv, al, qo, bits);
        }
        if (t == ZK_TAB_LL) { zk_lds_st<uint32_t>(&s_wbad[slot], wbad); atomicSub(&s_live, 1u); }
    }
    // every lane of the wave is back here (the blocks' loops have different lengths): the ring holds every sequence.  The ballot is
    // a convergent operation: the store that depends on it cannot be moved into one of the paths above (lanes without a block
    // would otherwise be free to publish DONE while their neighbours still walk)
    if (__ballot(1) != 0) zk_lds_st<uint32_t>(&s_prod[w], ZK_QUAD_DONE);
}
template <typename CP, int ZK_FSE_BLOCKS, int ZK_FSE_WAVES>
__global__ __launch_bounds__(64 * (2 * ZK_FSE_WAVES + 1)) void zk_k_fse_quad(const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, ZkSeqP *seqs, uint32_t all_blocks)
{
    zk_fse_quad_group<CP, ZK_FSE_BLOCKS, ZK_FSE_WAVES>(blockIdx.x, comp, blocks, nblocks, seqs, all_blocks);
}

// Blocks that SHARE their three tables need no per-block copy of them: one copy per workgroup, one block per LANE,
// full waves.  That is the case for Predefined_Mode (libzstd's small blocks) and for what this engine's own encoder
// writes: one set of FSE tables per frame -- the frame's first compressed block defines them, every later block says
// Repeat_Mode (zk_encode.hip, zk_k_enc_fse_build).  A workgroup takes 64 consecutive blocks; the first of them that has
// sequences sets the reference (per table: predefined, or the index of the defining block), the lanes whose blocks use
// exactly those tables take part, the others are left to zk_k_fse_quad (which skips what is marked done).  All 64 lanes
// walk in lock step so that the 8-B records can be stored cooperatively (ZkCoopFlush: fewer, fuller L2 write requests,
// which were 40% of the kernel's time).  A reference with own tables that fewer than ZK_FSEP_MIN_SHARE lanes share is not
// worth a lock-stepped wave (archives written by libzstd: every block its own tables): the workgroup leaves at once.
constexpr int ZK_FSEP_LANES = 64;                        // blocks per workgroup (one wave)
constexpr int ZK_FSEP_RING = 8;                          // records per lane between two cooperative flushes (8 x 8 B = one 64-B burst; 128-B bursts measured slower)
constexpr uint32_t ZK_FSEP_MIN_SHARE = 8;
constexpr uint32_t ZK_KEY_PREDEF = 0xFFFFFFFEu;
struct ZkFseShare { uint32_t al[3]; int32_t own[3]; uint32_t ok; };

// Wave-wide (64 lanes, all of them call it; deterministic, so every wave of a workgroup reaches the same verdict on its own):
// picks the reference and tells every lane whether its block takes part.  go: the workgroup has something to share.
struct ZkFsePick { uint32_t k0, k1, k2, r0, r1, r2; bool match, go; };
__device__ __forceinline__ ZkFsePick zk_fse_share_pick(uint32_t nblocks, uint32_t bi, const ZkBlock &b)
{
    ZkFsePick p;
    bool cand = bi < nblocks;
    p.k0 = p.k1 = p.k2 = 0;                                // (scalars, no arrays: nothing here may end up in scratch memory)
    if (cand) {
        cand = b.type == 2 && b.nseq != 0 && b.status == ZK_OK && b.pad == 0;
        const uint32_t m0 = (b.seq_modes >> 6) & 3, m1 = (b.seq_modes >> 4) & 3, m2 = (b.seq_modes >> 2) & 3;
        if (m0 == 1 || m1 == 1 || m2 == 1) cand = false;                 // RLE_Mode tables are per block: zk_k_fse_quad
        p.k0 = m0 == 0 ? ZK_KEY_PREDEF : m0 == 2 ? bi : b.tab_def[0];
        p.k1 = m1 == 0 ? ZK_KEY_PREDEF : m1 == 2 ? bi : b.tab_def[1];
        p.k2 = m2 == 0 ? ZK_KEY_PREDEF : m2 == 2 ? bi : b.tab_def[2];
    }
    const uint64_t cm = __ballot(cand);
    const int ref = cm ? __builtin_ctzll(cm) : 0;
    p.r0 = __shfl(p.k0, ref, 64); p.r1 = __shfl(p.k1, ref, 64); p.r2 = __shfl(p.k2, ref, 64);
    p.match = cand && p.k0 == p.r0 && p.k1 == p.r1 && p.k2 == p.r2;
    const bool predef = p.r0 == ZK_KEY_PREDEF && p.r1 == ZK_KEY_PREDEF && p.r2 == ZK_KEY_PREDEF;
    p.go = cm != 0 && (predef || (uint32_t)__popcll(__ballot(p.match)) >= ZK_FSEP_MIN_SHARE);
    return p;
}
// the reference's three tables (called by 16 lanes redundantly: identical LDS writes, >= 16 active lanes)
template <typename CP>
__device__ __forceinline__ void zk_fse_share_build(const uint8_t *comp, const ZkBlock *blocks, const ZkFsePick &p, ZkSeqTablesT<CP> *T, ZkFseShare *sh,
                                                   const uint32_t *llv, const uint32_t *mlv)
{
    ZkBlock fake;
    fake.seq_modes = 0; fake.seq_off = 0; fake.bsize = 0; fake.src = 0;
    uint32_t ok = 1;
#define ZK_SHARE_TABLE(t, rk) do { uint32_t a = 0; \
        const int32_t r = (rk) == ZK_KEY_PREDEF ? zk_seq_table_setup<CP>(comp, fake, t, T, &a, llv, mlv) \
                                                : zk_seq_table_setup<CP>(comp, blocks[rk], t, T, &a, llv, mlv); \
        sh->al[t] = a; sh->own[t] = r; \
        if (r < 0) ok = 0;                              /* a damaged description: zk_k_fse_quad reports it block by block */ \
    } while (0)
    ZK_SHARE_TABLE(0, p.r0); ZK_SHARE_TABLE(1, p.r1); ZK_SHARE_TABLE(2, p.r2);
#undef ZK_SHARE_TABLE
    sh->ok = ok;
}
// where the lane's bitstream starts: the defining block carries the descriptions itself
__device__ __forceinline__ uint32_t zk_fse_share_offset(const ZkFsePick &p, uint32_t bi, const ZkBlock &b, const ZkFseShare *sh)
{
    uint32_t off = b.seq_off + 1;
    if (p.k0 == bi) off += (uint32_t)sh->own[0];
    if (p.k1 == bi) off += (uint32_t)sh->own[1];
    if (p.k2 == bi) off += (uint32_t)sh->own[2];
    return off;
}

template <typename RD>
__global__ __launch_bounds__(ZK_FSEP_LANES) void zk_k_fse_predef(const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, ZkSeqP *seqs)
{
    __shared__ ZkSeqTables T;                              // shared, read-only after the build
    __shared__ __attribute__((aligned(16))) ZkSeqP ring[ZK_FSEP_LANES][ZK_FSEP_RING];
    __shared__ ZkCoopFlush coop;
    __shared__ ZkFseShare share;
    __shared__ uint32_t llv[36], mlv[53];
    const uint32_t tid = threadIdx.x;
    {
        const uint32_t ll_init[36] = ZK_LL_TABLE;
        const uint32_t ml_init[53] = ZK_ML_TABLE;
        if (tid < 36) llv[tid] = ll_init[tid];
        if (tid < 53) mlv[tid] = ml_init[tid];
        if (tid == 0) { coop.ring = &ring[0][0]; coop.seqs = seqs; coop.nloop = 0; share.ok = 0; }
    }
    __syncthreads();
    const uint32_t bi = blockIdx.x * ZK_FSEP_LANES + tid;
    ZkBlock b;
    if (bi < nblocks) b = blocks[bi];
    const ZkFsePick pk = zk_fse_share_pick(nblocks, bi, b);
    if (!pk.go) return;                                    // nothing shared here (wave-uniform)
    if (tid < 16) zk_fse_share_build<ZkCells32>(comp, blocks, pk, &T, &share, llv, mlv);
    bool active = pk.match;
    coop.base[tid] = active ? b.seq_base : 0;
    coop.nseq[tid] = active ? b.nseq : 0;
    if (active) atomicMax(&coop.nloop, b.nseq);
    __syncthreads();
    if (!share.ok) return;
    // every lane of the wave runs the walk in lock step (inactive ones only help storing the others' records)
    zk_seq_walk<ZK_FSEP_RING, RD>(comp, b, active ? zk_fse_share_offset(pk, bi, b, &share) : 0, T.ll, T.of, T.ml, share.al, ring[tid], seqs, llv, mlv, true, &coop, active, tid);
    if (!active) return;
    ZkBlock *o = &blocks[bi];
    o->out_size = b.out_size;
    o->rep_out[0] = b.rep_out[0]; o->rep_out[1] = b.rep_out[1]; o->rep_out[2] = b.rep_out[2];
    if (b.status != ZK_OK) o->status = b.status;      // (never OK over the literal kernel's verdict: the two run side by side)
    o->pad = 1;                                            // done: zk_k_fse_quad skips it
}

// The same with a feeder wave: wave 0 walks (reader ZkRevL: stream words out of an LDS ring, no global load in the
// walking wave), wave 1 feeds the ring lane for lane.  Used when the device is full of walkers (zk_launch_fse).
// (Measured and dropped, round 6: TWO walker waves of 32 lanes each -- blocks 0..31 and 32..63 of the workgroup -- instead of one of 64, so
//  that four half-busy walker waves interleave on a SIMD where two full ones leave its vector unit 46 % busy (profiles/r06_pmc_entropy.txt);
//  a wave with 32 lanes at work takes two passes of the SIMD per instruction.  Six waves per SIMD need 80 registers (85 without the
//  limit; with it 36 bytes of scratch outside the walk): bit-exact, 3.13 -> 4.8 ms.  profiles/r06_fse_half_waves_probe.txt)
__global__ __launch_bounds__(2 * ZK_FSEP_LANES) void zk_k_fse_predef_fed(const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, ZkSeqP *seqs)
{
    __shared__ ZkSeqTablesT<ZkCells64> T;                  // 64-bit cells: the value baseline rides along (shared tables, LDS is free)
    __shared__ __attribute__((aligned(16))) ZkSeqP ring[ZK_FSEP_LANES][ZK_FSEP_RING];
    __shared__ ZkCoopFlush coop;
    __shared__ ZkRevLShared feed;
    __shared__ ZkFseShare share;
    __shared__ uint32_t llv[36], mlv[53], s_done;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const bool walker = tid < 64;
    {
        const uint32_t ll_init[36] = ZK_LL_TABLE;
        const uint32_t ml_init[53] = ZK_ML_TABLE;
        if (tid < 36) llv[tid] = ll_init[tid];
        if (tid < 53) mlv[tid] = ml_init[tid];
        if (walker) { feed.filled[lane] = 0; feed.taken[lane] = 0; }
        if (tid == 0) { coop.ring = &ring[0][0]; coop.seqs = seqs; coop.nloop = 0; s_done = 0; share.ok = 0; }
    }
    __syncthreads();
    const uint32_t bi = blockIdx.x * ZK_FSEP_LANES + lane;
    ZkBlock b;
    if (bi < nblocks) b = blocks[bi];
    const ZkFsePick pk = zk_fse_share_pick(nblocks, bi, b);          // both waves, same verdict
    if (!pk.go) return;
    if (tid < 16) zk_fse_share_build<ZkCells64>(comp, blocks, pk, &T, &share, llv, mlv);
    const bool active = pk.match;
    if (walker) {
        coop.base[lane] = active ? b.seq_base : 0;
        coop.nseq[lane] = active ? b.nseq : 0;
        if (active) atomicMax(&coop.nloop, b.nseq);
    }
    __syncthreads();
    if (!share.ok) return;
    const uint32_t bs_off = active ? zk_fse_share_offset(pk, bi, b, &share) : 0;
    if (walker) {
        zk_seq_walk<ZK_FSEP_RING, ZkRevL, ZkCells64>(comp, b, bs_off, T.ll, T.of, T.ml, share.al, ring[lane], seqs, llv, mlv, true, &coop, active, lane, &feed);
        zk_lds_st<uint32_t>(&s_done, 1u);
        if (!active) return;
        ZkBlock *o = &blocks[bi];
        o->out_size = b.out_size;
        o->rep_out[0] = b.rep_out[0]; o->rep_out[1] = b.rep_out[1]; o->rep_out[2] = b.rep_out[2];
        if (b.status != ZK_OK) o->status = b.status;      // (never OK over the literal kernel's verdict: the two run side by side)
        o->pad = 1;
    } else if (active && bs_off < b.bsize) {
        // feeder: aligned words of the lane's bitstream, last word first (ZkRevL::word_count / W(j))
        const uint8_t *base = comp + b.src + bs_off;
        const uint32_t len = b.bsize - bs_off, nwords = ZkRevL::word_count(base, len);
        const uint8_t *ptr = reinterpret_cast<const uint8_t *>((((uintptr_t)base + len) + 7) & ~(uintptr_t)7) - 8;
        uint32_t f = 0;
        while (f < nwords && !zk_lds_ld<uint32_t>(&s_done)) {
            if (f - zk_lds_ld<uint32_t>(&feed.taken[lane]) < ZK_REVL_RING) {
                const uint64_t w = *reinterpret_cast<const uint64_t *>(ptr);
                zk_lds_st<uint64_t>(&feed.ring[f % ZK_REVL_RING][lane], w);
                f++; ptr -= 8;
                zk_lds_st<uint32_t>(&feed.filled[lane], f);
            } else __builtin_amdgcn_s_sleep(1);
        }
    }
}

// Several table sets per workgroup.  The kernels above serve ONE set per 64 consecutive blocks, which is a whole frame only
// when a frame has >= 64 blocks (2 MiB at this encoder's 32 KiB blocks): with 1 MiB frames half of every workgroup's
// blocks, with 64 KiB frames (16 blocks of 4 KiB) three quarters, fell through to the per-block kernel (1 GiB of 1 MiB
// frames: 6.3 ms instead of 2.2).  Here up to ZK_FSEP_SETS sets live side by side in LDS -- set s is built by lanes
// 16 s .. 16 s + 15 while the others build theirs -- and a lane walks with the tables of its own set (the table base is a
// lane value either way: same instructions in the walk).
constexpr int ZK_FSEP_SETS = 4;
struct ZkFsePickSets { uint32_t k0, k1, k2; int32_t set; uint32_t nsets; };
__device__ __forceinline__ ZkFsePickSets zk_fse_share_pick_sets(uint32_t nblocks, uint32_t bi, const ZkBlock &b)
{
    ZkFsePickSets p;
    bool cand = bi < nblocks;
    p.k0 = p.k1 = p.k2 = 0; p.set = -1; p.nsets = 0;
    if (cand) {
        cand = b.type == 2 && b.nseq != 0 && b.status == ZK_OK && b.pad == 0;
        const uint32_t m0 = (b.seq_modes >> 6) & 3, m1 = (b.seq_modes >> 4) & 3, m2 = (b.seq_modes >> 2) & 3;
        if (m0 == 1 || m1 == 1 || m2 == 1) cand = false;                 // RLE_Mode tables are per block: zk_k_fse_quad
        p.k0 = m0 == 0 ? ZK_KEY_PREDEF : m0 == 2 ? bi : b.tab_def[0];
        p.k1 = m1 == 0 ? ZK_KEY_PREDEF : m1 == 2 ? bi : b.tab_def[1];
        p.k2 = m2 == 0 ? ZK_KEY_PREDEF : m2 == 2 ? bi : b.tab_def[2];
    }
    uint64_t rem = __ballot(cand);                          // lanes not yet assigned to a set (or found to have too few partners)
    for (int it = 0; it < 2 * ZK_FSEP_SETS && rem && p.nsets < (uint32_t)ZK_FSEP_SETS; it++) {
        const int ref = __builtin_ctzll(rem);
        const uint32_t r0 = __shfl(p.k0, ref, 64), r1 = __shfl(p.k1, ref, 64), r2 = __shfl(p.k2, ref, 64);
        const bool m = cand && p.k0 == r0 && p.k1 == r1 && p.k2 == r2;
        const uint64_t mm = __ballot(m);
        const bool predef = r0 == ZK_KEY_PREDEF && r1 == ZK_KEY_PREDEF && r2 == ZK_KEY_PREDEF;
        if (predef || (uint32_t)__popcll(mm) >= ZK_FSEP_MIN_SHARE) { if (m) p.set = (int32_t)p.nsets; p.nsets++; }
        rem &= ~mm;
    }
    return p;
}
__global__ __launch_bounds__(ZK_FSEP_LANES) void zk_k_fse_sets(const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, ZkSeqP *seqs)
{
    __shared__ ZkSeqTables T[ZK_FSEP_SETS];                // read-only after the build
    __shared__ __attribute__((aligned(16))) ZkSeqP ring[ZK_FSEP_LANES][ZK_FSEP_RING];
    __shared__ ZkCoopFlush coop;
    __shared__ ZkFseShare share[ZK_FSEP_SETS];
    __shared__ uint32_t llv[36], mlv[53];
    const uint32_t tid = threadIdx.x;
    {
        const uint32_t ll_init[36] = ZK_LL_TABLE;
        const uint32_t ml_init[53] = ZK_ML_TABLE;
        if (tid < 36) llv[tid] = ll_init[tid];
        if (tid < 53) mlv[tid] = ml_init[tid];
        if (tid < (uint32_t)ZK_FSEP_SETS) share[tid].ok = 0;
        if (tid == 0) { coop.ring = &ring[0][0]; coop.seqs = seqs; coop.nloop = 0; }
    }
    __syncthreads();
    const uint32_t bi = blockIdx.x * ZK_FSEP_LANES + tid;
    ZkBlock b;
    if (bi < nblocks) b = blocks[bi];
    const ZkFsePickSets pk = zk_fse_share_pick_sets(nblocks, bi, b);
    if (!pk.nsets) return;                                  // nothing shared here (wave-uniform)
    for (int s = 0; s < ZK_FSEP_SETS; s++) {
        const uint64_t mm = __ballot(pk.set == s);
        if (!mm) continue;
        const int ref = __builtin_ctzll(mm);
        ZkFsePick one;
        one.k0 = one.k1 = one.k2 = 0; one.match = false; one.go = true;
        one.r0 = __shfl(pk.k0, ref, 64); one.r1 = __shfl(pk.k1, ref, 64); one.r2 = __shfl(pk.k2, ref, 64);
        if ((int)(tid >> 4) == s) zk_fse_share_build<ZkCells32>(comp, blocks, one, &T[s], &share[s], llv, mlv);
    }
    __syncthreads();
    const uint32_t ms = pk.set < 0 ? 0u : (uint32_t)pk.set;
    const bool active = pk.set >= 0 && share[ms].ok;       // a damaged description: zk_k_fse_quad reports it block by block
    coop.base[tid] = active ? b.seq_base : 0;
    coop.nseq[tid] = active ? b.nseq : 0;
    if (active) atomicMax(&coop.nloop, b.nseq);
    if (!__syncthreads_or(active)) return;
    uint32_t bs_off = 0;
    if (active) {
        bs_off = b.seq_off + 1;
        if (pk.k0 == bi) bs_off += (uint32_t)share[ms].own[0];
        if (pk.k1 == bi) bs_off += (uint32_t)share[ms].own[1];
        if (pk.k2 == bi) bs_off += (uint32_t)share[ms].own[2];
    }
    // every lane of the wave runs the walk in lock step (inactive ones only help storing the others' records)
    zk_seq_walk<ZK_FSEP_RING, ZkRevU>(comp, b, bs_off, T[ms].ll, T[ms].of, T[ms].ml, share[ms].al, ring[tid], seqs, llv, mlv, true, &coop, active, tid);
    if (!active) return;
    ZkBlock *o = &blocks[bi];
    o->out_size = b.out_size;
    o->rep_out[0] = b.rep_out[0]; o->rep_out[1] = b.rep_out[1]; o->rep_out[2] = b.rep_out[2];
    if (b.status != ZK_OK) o->status = b.status;      // (never OK over the literal kernel's verdict: the two run side by side)
    o->pad = 1;                                            // done: zk_k_fse_quad skips it
}

// ------------------------------------------------------------------------------------------------ sequence execution
// One workgroup (T lanes) per frame.  The output of a compressed block is produced in tiles of T x 16 B.  Per tile:
//   1. the sequences overlapping the tile are staged in LDS (16 B records, offsets resolved, validated);
//   2. lane-per-SEQUENCE: the sequence's staged index is written to every 16-B slot whose first byte it covers
//      (usually 0 or 1 slots; sequences that start more than ZK_EXEC_LONG slots are marked by all lanes);
//   3. lane-per-SLOT: walk the staged sequences over the slot's 16 bytes -> 16 source words (literal index or
//      history position) in registers, mirrored in an LDS map;
//   4. every lane follows in-tile sources to their origin through the map (all 16 chains advance together, the
//      LDS reads of one pass are in flight at once), issues the 16 byte gathers back to back (literal buffer /
//      history already in HBM-L2) and does one coalesced 16 B store;
//   5. a workgroup barrier orders the tiles.
constexpr int ZK_EXEC_B = (int)ZK_EXEC_SLOT;

#ifdef ZK_EXEC_CLOCKS
// experiments: shader-clock totals per phase of the tile loop, summed over lane 0 of every workgroup (tools/exec_clocks.py)
__device__ unsigned long long zk_dbg_clk[8];
extern "C" void zk_debug_clocks(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(zk_dbg_clk), z, sizeof z); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(zk_dbg_clk), 8 * sizeof(unsigned long long));
}
#define ZK_CLK(i) do { const unsigned long long now_ = clock64(); clk_[i] += now_ - t_; t_ = now_; } while (0)
#else
#define ZK_CLK(i) do { } while (0)
#endif
// workgroup barrier that orders LDS traffic only (no wait for global loads / stores in flight)
#ifdef ZK_EXEC_FULL_BARRIERS                         // experiment: every barrier of the tile loop waits for the stores too
#define ZK_LDS_BARRIER() __syncthreads()
#else
#define ZK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
// A value every lane of the wave holds alike, said so to the compiler: it lives in a scalar register from here on.  (What the executor
// reads from the frame's and the block's descriptors arrives through vector loads -- the descriptors are written by other kernels,
// nothing tells the compiler they are read-only -- and stayed in vector registers for the whole kernel: ~30 of them.)
__device__ __forceinline__ uint32_t zk_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t zk_uni(uint64_t v) { return (uint64_t)zk_uni((uint32_t)v) | ((uint64_t)zk_uni((uint32_t)(v >> 32)) << 32); }
// (A POINTER is made uniform through its offset -- base + zk_uni(offset), the base a kernel argument: a pointer rebuilt from an integer is a
// generic one, and the compiler emits FLAT loads and stores for it instead of global ones.)

// PROGRESS WORDS (zk_k_xxh64_follow): with `progress` set, lane 0 of a frame's workgroup publishes how many bytes of the frame are
// complete -- whenever a block ends behind another 2^ZK_PUB_LOG bytes, and at the frame's end -- so that the checksum of a frame can
// be computed WHILE the frame is being written, by a wave of another kernel on another queue.
//   The XCDs' L2s are not coherent with each other (MI355X_MICROARCH.md): bytes that have to be seen anywhere on the device need an
// agent-scope release, `buffer_wbl2 sc1`, the write-back of the whole L2's dirty lines.  Built and measured (16 of them per frame:
// the executor 10.7 -> 14.0 ms in a kernel trace, everything the checksums gained and more).  ONE L2, though, is coherent by itself: a
// store that has been acknowledged is in the XCD's L2, and a load from any CU of that XCD that misses its L1 finds it there.  So the
// word carries the XCD the workgroup runs on (HW_REG_XCC_ID), the executor pays a wait for its own stores (behind the block's
// barrier every wave has had that wait) and one 8-byte store, and a checksum wave follows exactly the frames whose executors turn out
// to share ITS XCD -- the dispatcher's observed habit (workgroup b on XCD b % 8, which the checksum kernel's frame order is made for)
// decides how many frames that is, never whether a byte is right: a frame from another XCD is left to the pass behind the executor.
constexpr int ZK_PUB_LOG = 17;
// the word: [31:0] bytes complete, [35:32] XCD + 1 (a word that is 0 says nothing yet), [60] the frame has failed,
// [61] / [62] set by the checksum wave at its end: verified / hashed and found different
constexpr uint64_t ZK_PROG_ABORT = 1ull << 60, ZK_PROG_VERIFIED = 1ull << 61, ZK_PROG_MISMATCH = 1ull << 62;
__device__ __forceinline__ uint32_t zk_xcc_id()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}
__device__ __forceinline__ void zk_publish(uint64_t *word, uint32_t bytes, uint64_t flags = 0)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(word, flags | ((uint64_t)(zk_xcc_id() + 1) << 32) | bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// THE CHASE: a source word that points into the tile [ts, te) is replaced by the word of the byte it points at, until none does.
// Counted on the bench's text (tools/chase_stats.py, the lane code on the CPU): 81.6 % of the bytes need no step, 18.1 % one, 0.3 % two --
// but a WAVE runs as many passes over all 16 words of all its lanes as its deepest chain, plus the one that finds nothing to do: 2.76 on
// average.  So the pass that finds nothing is not a pass: the least of the lane's sixteen (word - first in-tile word) says whether
// any word still points into the tile (a subtraction and a third of a min3 per word, one vote per wave), and the passes themselves keep no
// "again".  The first in-tile word in ONE scalar register (through zk_uni: the compiler otherwise subtracts ts and the bias separately).
// Passes after the first take the lane's words in four groups of four: a group in which no lane of the wave has anything left (the rule:
// 0.3 % of the bytes need a second step) is skipped by a scalar branch -- the least distance of each group is what the test computed anyway.
#ifndef ZK_EXEC_CHASE_GROUPS
#define ZK_EXEC_CHASE_GROUPS 1
#endif
#define ZK_EXEC_CHASE(sw, srcmap, ts, te) do {                                                                         \
        const uint32_t mbase_ = zk_uni(ZK_SRC_BIAS + (ts)), span_ = (te) - (ts);                                        \
        _Pragma("unroll") for (int k = 0; k < ZK_EXEC_B; k++) { const uint32_t d_ = sw[k] - mbase_; if (d_ < span_) sw[k] = srcmap[zk_exec_map_index(d_)]; }  \
        for (;;) {                                                                                                      \
            uint32_t lg_[ZK_EXEC_B / 4];                                                                                \
            _Pragma("unroll") for (int g = 0; g < ZK_EXEC_B / 4; g++) {                                                 \
                const uint32_t d0_ = sw[4 * g] - mbase_, d1_ = sw[4 * g + 1] - mbase_, d2_ = sw[4 * g + 2] - mbase_, d3_ = sw[4 * g + 3] - mbase_;      \
                const uint32_t m_ = d0_ < d1_ ? d0_ : d1_, n_ = d2_ < d3_ ? d2_ : d3_; lg_[g] = m_ < n_ ? m_ : n_;      \
            }                                                                                                           \
            uint32_t least_ = lg_[0];                                                                                   \
            _Pragma("unroll") for (int g = 1; g < ZK_EXEC_B / 4; g++) least_ = lg_[g] < least_ ? lg_[g] : least_;      \
            if (!__any(least_ < span_)) break;                                                                          \
            _Pragma("unroll") for (int g = 0; g < ZK_EXEC_B / 4; g++) {                                                 \
                if (ZK_EXEC_CHASE_GROUPS && !__any(lg_[g] < span_)) continue;                                           \
                _Pragma("unroll") for (int k = 4 * g; k < 4 * g + 4; k++) { const uint32_t d_ = sw[k] - mbase_; if (d_ < span_) sw[k] = srcmap[zk_exec_map_index(d_)]; } \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
// REDO: only the frames zk_k_exec_seg gave up on (ZK_E_SEG_OVERFLOW: more hole records than their region holds) are executed, from scratch
// (Measured and dropped, round 6: touching the sources of FAR matches ahead of time.  A record that was settled into the ring -- one to two
//  tiles before its bytes are gathered -- with a match more than 24 KiB back had the line of its source requested by a load nobody waits
//  for (global_load_lds_dword into a scratch row of LDS, issued behind the tile's store: the next wait for memory is a tile later).  On
//  4 GiB of the reference's level-3 frames (window = the frame, 28 % of the offsets beyond 64 KiB): 512-lane tiles 15.87 -> 16.61 ms,
//  256-lane tiles with a ring of 4 T 17.06 -> 17.78 ms; level 1: 9.39 -> 9.55 / 9.46 -> 9.65 ms.  Slower everywhere: the executor does
//  not sit waiting for those lines -- five workgroups per CU cover each other's misses -- it is the NUMBER of lines that costs
//  (profiles/r05_ref3_pmc_fetch_write.txt: 23 GiB fetched for 4 GiB of output), and a touch that is evicted before its gather fetches
//  the line twice.  What did help there: fewer frames resident, zk_launch_exec.  profiles/r06_l3_far_touch_probe.txt.
//  A first form of the touch -- a byte load into a "sink" register -- faulted: the compiler reuses a register it does not know a load
//  is still going to write.)
#ifndef ZK_EXEC_WIDE
#define ZK_EXEC_WIDE 1
#endif
#ifndef ZK_EXEC_TOUCH
#define ZK_EXEC_TOUCH 1
#endif
#ifdef ZK_EXEC_NO_WPE
#define ZK_EXEC_WPE(T)
#else
#define ZK_EXEC_WPE(T) __attribute__((amdgpu_waves_per_eu(T == 256 ? 5 : 1)))
#endif
template <int T, bool PFX, int CAPX = 2, bool REDO = false>
__global__ __launch_bounds__(T) ZK_EXEC_WPE(T) void zk_k_exec(const uint8_t *__restrict__ comp, const uint64_t *__restrict__ d_off, uint32_t first,
                                               const uint32_t *__restrict__ ids, const uint64_t *__restrict__ out_off,
                                               const ZkBlock *__restrict__ blocks, const ZkFrameBase *__restrict__ bases,
                                               ZkFrameInfo *__restrict__ infos, const ZkSeqP *__restrict__ seqs,
                                               const uint8_t *__restrict__ lit_scratch, uint8_t *__restrict__ dst,
                                               const uint8_t *__restrict__ prefix, uint64_t plen, uint64_t *__restrict__ progress)
{
    // The staged sequences live in an LDS RING of CAP records, slot = block sequence index & (CAP - 1).  A tile retires the
    // jn sequences it has consumed and exactly as many new ones are fetched for the tiles to come -- requested right after
    // the marking pass, written into the retired slots once the slot pass no longer reads them: the fetch (an HBM round
    // trip that used to open every tile, in front of a barrier, for twice the records a tile needs) is off the critical
    // path and every record is read once.
    constexpr int CAP = CAPX * T;                // 2 T records; 4 T where sequences are dense (zk_launch_exec)
    constexpr uint32_t M = CAP - 1;
    constexpr int NPF = CAP / T;                 // records a lane may have to fetch per tile
    __shared__ __attribute__((aligned(16))) ZkSeq S[CAP];
    __shared__ __attribute__((aligned(16))) uint32_t srcmap[T * ZK_EXEC_B];
    __shared__ uint32_t slot_seq[T];
    __shared__ uint32_t longlist[CAP + 1];
    __shared__ uint32_t s_slow[(T + 31) / 32];   // the tile's slots that hold bytes of a match overlapping its own output (zk_exec_mark_runs)
    __shared__ uint32_t s_jn, s_nlong;
    __shared__ uint32_t s_bad[2];                // a lane found a bad record in a tile of this parity (the tile loop's barriers order LDS only)
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    ZkFrameInfo fi = infos[f];
    fi.status = zk_uni(fi.status); fi.n_blocks = zk_uni(fi.n_blocks); fi.window = zk_uni(fi.window);
    if (REDO ? fi.status != ZK_E_SEG_OVERFLOW : fi.status != ZK_OK) {
        if (progress && tid == 0) zk_publish(progress + f, 0, ZK_PROG_ABORT);
        return;
    }
    const uint32_t id = zk_uni(ids ? ids[f] : first + f);
    const uint64_t d_size = zk_uni(d_off[id + 1] - d_off[id]);
    uint32_t published = 0;                                  // the last published count >> ZK_PUB_LOG
    if (progress && tid == 0) zk_publish(progress + f, 0);   // (which XCD this is: a checksum wave elsewhere stops waiting for the frame)
    uint8_t *out = dst + zk_uni((uint64_t)(out_off ? out_off[f] : d_off[id] - d_off[first]));     // indexed batches are packed in list order
    const ZkBlock *fb = blocks + zk_uni((uint64_t)bases[f].block_base);
    uint64_t pos = 0;
    uint32_t rep[3] = {1, 4, 8};
    uint32_t err = ZK_OK;
    const uint32_t block_max = fi.window < ZK_BLOCK_MAX ? fi.window : ZK_BLOCK_MAX;
#ifdef ZK_EXEC_CLOCKS
    unsigned long long clk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_ = clock64();
#endif

    for (uint32_t bk = 0; bk < fi.n_blocks && err == ZK_OK; bk++) {
        const ZkBlock &b = fb[bk];
        const uint32_t b_status = zk_uni(b.status), b_out_size = zk_uni(b.out_size), b_type = zk_uni((uint32_t)b.type);
        if (b_status != ZK_OK) { err = b_status; break; }
        if (pos + b_out_size > d_size) { err = ZK_E_CORRUPTION; break; }
        if (b_out_size > block_max) { err = ZK_E_CORRUPTION; break; }       // Block_Maximum_Size = min(Window_Size, 128 KiB) also bounds what a compressed block regenerates (RFC 8878 3.1.1.2.4; raw / RLE: zk_walk_frame)
        uint8_t *bout = out + pos;
        if (b_type <= 1) {
            // raw / RLE block: bytes up to the first 16-byte boundary of the output, 16-byte stores (the source of a raw
            // block is read with whatever alignment it has), bytes behind the last boundary
            const uint8_t *s = comp + zk_uni(b.src);
            const uint32_t n = zk_uni(b.bsize);
            const uint32_t head0 = (uint32_t)((0 - (uintptr_t)bout) & 15), head = head0 < n ? head0 : n;
            const uint32_t n16 = (n - head) >> 4;
            if (b_type == 0) {
                if (tid < head) bout[tid] = s[tid];
                for (uint32_t i = tid; i < n16; i += T) {
                    uint4 v;
                    __builtin_memcpy(&v, s + head + (i << 4), 16);
                    *reinterpret_cast<uint4 *>(bout + head + (i << 4)) = v;
                }
                for (uint32_t i = head + (n16 << 4) + tid; i < n; i += T) bout[i] = s[i];
            } else {
                const uint32_t v1 = s[0], v4 = v1 * 0x01010101u;
                if (tid < head) bout[tid] = (uint8_t)v1;
                for (uint32_t i = tid; i < n16; i += T) *reinterpret_cast<uint4 *>(bout + head + (i << 4)) = make_uint4(v4, v4, v4, v4);
                for (uint32_t i = head + (n16 << 4) + tid; i < n; i += T) bout[i] = (uint8_t)v1;
            }
        } else {
            const ZkSeqP *sq = seqs + zk_uni(b.seq_base);
            const uint32_t b_lit_type = zk_uni((uint32_t)b.lit_type);
            const uint8_t *lit = b_lit_type >= 2 ? lit_scratch + zk_uni(b.lit_base) : comp + zk_uni(b.src) + zk_uni(b.lit_off);
            const uint32_t lit_mask = b_lit_type == 1 ? 0u : 0x7fffffffu;       // RLE literals: every index reads byte 0
            // the records the NEXT block starts with, touched during this block's last tile (ZK_EXEC_TOUCH_NEXT below)
            const bool nx_any = ZK_EXEC_TOUCH && bk + 1 < fi.n_blocks && zk_uni((uint32_t)fb[bk + 1].type) == 2u;
            const ZkSeqP *nx_sq = seqs + (nx_any ? zk_uni(fb[bk + 1].seq_base) : 0);
            const uint32_t nx_lines = nx_any ? (zk_uni(fb[bk + 1].nseq) + 7u) / 8u : 0u;     // 64-byte lines of its records
            const uint64_t lit_w = (uint64_t)(uintptr_t)lit - ZK_SRC_LIT, his_w = (uint64_t)(uintptr_t)bout - ZK_SRC_BIAS;    // + a source word = the byte's address
            const uint32_t nseq = zk_uni(b.nseq), out_size = b_out_size, lit_regen = zk_uni(b.lit_regen);
            // record idx of the block (idx == nseq: the trailing-literals pseudo sequence) -> staged form, offsets resolved
            // and validated.  Without a prefix an offset is bounded by the bytes produced so far and by the frame's window;
            // with one (ZSTD_DCtx_refPrefix: the prefix sits right before the frame) only by availability, which is all
            // libzstd's ZSTD_execSequence checks.
            auto fetch = [&](uint32_t idx, ZkSeqP &p0, ZkSeqP &p1) {
                p1 = idx < nseq ? sq[idx] : 0;
                p0 = idx && idx < nseq ? sq[idx - 1] : 0;
            };
            auto settle = [&](uint32_t idx, ZkSeqP p0, ZkSeqP p1, int &bad) {
                uint4 r;
                if (idx < nseq) {
                    const ZkSeq q = zk_seq_unpack(p0, p1, idx == 0);
                    const uint32_t off = zk_rep_resolve(q.off, rep);
                    const uint32_t mstart = q.out_end - q.ml;
                    if (PFX ? (off == 0 || pos + mstart + plen < off) : (off == 0 || pos + mstart < off || off > fi.window)) bad = 1;
                    if (off >= ZK_SRC_BIAS || q.ml > q.out_end) bad = 1;     // source words carry positions down to -2^30 only
                    r = make_uint4(q.out_end, q.ml, off, q.lit_end);
                } else r = make_uint4(out_size, 0, 1, lit_regen);
                reinterpret_cast<uint4 *>(S)[idx & M] = r;
            };
            int bad = 0;
            uint32_t ja = 0, ts = 0, prev_end = 0;           // prev_end: out_end of sequence ja - 1
            uint32_t staged_end = nseq + 1 < (uint32_t)CAP ? nseq + 1 : (uint32_t)CAP;      // records [ja, staged_end) are in the ring
            for (uint32_t idx = tid; idx < staged_end; idx += T) { ZkSeqP p0, p1; fetch(idx, p0, p1); settle(idx, p0, p1, bad); }
            if (tid == 0) { s_jn = staged_end; s_nlong = 0; s_bad[0] = 0; s_bad[1] = 0; }        // s_jn starts at "every staged sequence ends inside the tile"
            if (__syncthreads_or(bad)) { err = ZK_E_CORRUPTION; }
            ZK_CLK(0);
            // Barriers of the tile loop.  __syncthreads() makes a wave wait for everything it has in flight -- also for the
            // acknowledgement of the tile's stores, about a microsecond away, at the barrier right behind them.  Only the gathers of
            // the NEXT tile need those bytes (other waves' stores included), three barriers later: that one barrier stays a full one,
            // the others order LDS only, and the stores settle while the next tile is staged, marked and mapped.
            uint32_t tpar = 0;
            while (err == ZK_OK && ts < out_size) {
                const uint32_t nl = staged_end - ja;
                const uint32_t cap_end = zk_uni(S[(staged_end - 1) & M].out_end);
                const uint32_t te = ts + T * ZK_EXEC_B < cap_end ? ts + T * ZK_EXEC_B : cap_end;
                // 1a. the map starts empty: the lane of a sequence leaves, at the bytes where its two runs start, the word that
                //     run adds to a position (zk_exec_mark_runs); the slot pass below only carries them forward
#pragma unroll
                for (int k = 0; k < ZK_EXEC_B; k += 4) *reinterpret_cast<uint4 *>(&srcmap[zk_exec_map_index(tid * ZK_EXEC_B + k)]) = make_uint4(0, 0, 0, 0);
                if (tid < (T + 31) / 32) s_slow[tid] = 0;
                ZK_LDS_BARRIER();
                if (tid == 0) s_bad[tpar ^ 1] = 0;                               // the flag of the tile before: every wave has read it (in front of this barrier); the next tile sets it
                // 2. lane per sequence: mark the slots it starts; the first sequence that outlives the tile sets jn
                // (nl <= CAP = NPF x T: the loop written out, every turn's LDS reads in front -- as a loop each turn waited for its own)
                ZkSeq mes[NPF]; uint32_t starts_[NPF], poffs_[NPF];
#pragma unroll
                for (int u = 0; u < NPF; u++) {
                    const uint32_t i = tid + (uint32_t)u * T, idx = ja + i;
                    mes[u] = S[idx & M];
                    const ZkSeq &pv = S[(idx - 1) & M];
                    starts_[u] = i ? pv.out_end : prev_end; poffs_[u] = pv.off;   // (the offset of a sequence that has left the ring is never needed: zk_exec_mark_runs)
                }
#pragma unroll
                for (int u = 0; u < NPF; u++) {
                    const uint32_t i = tid + (uint32_t)u * T;
                    if (i >= nl) break;
                    const uint32_t idx = ja + i;
                    const ZkSeq me = mes[u];
                    const uint32_t end = me.out_end;
                    const uint32_t start = starts_[u], prev_off = poffs_[u];
                    const uint32_t lo = start > ts ? start : ts, hi = end < te ? end : te;
                    if (lo < hi) {
                        uint32_t s0, n;
                        zk_exec_slot_span(ts, lo, hi, s0, n);
                        if (n > ZK_EXEC_LONG) longlist[atomicAdd(&s_nlong, 1u)] = idx;
                        else {                                   // (written out: as a loop it was two loops, pairs and a remainder, with the mask juggling of both)
#pragma unroll
                            for (uint32_t k = 0; k < ZK_EXEC_LONG; k++) if (k < n) slot_seq[s0 + k] = idx;
                        }
                        zk_exec_mark_runs(me, prev_off, start, ts, te, srcmap, [&](uint32_t w, uint32_t bits) { atomicOr(&s_slow[w], bits); });
                    }
                    if (end > te && start <= te) s_jn = i;
                }
                ZK_CLK(1);
                ZK_LDS_BARRIER();
                ZK_CLK(2);
                const uint32_t nlong = zk_uni(s_nlong), jn = zk_uni(s_jn);
                // the records that take the retired slots: requested now, needed two barriers from here
                const uint32_t fetch_end = staged_end + jn < nseq + 1 ? staged_end + jn : nseq + 1;
                ZkSeqP pf0[NPF], pf1[NPF];
#pragma unroll
                for (int u = 0; u < NPF; u++) {
                    const uint32_t idx = staged_end + tid + (uint32_t)u * T;
                    pf0[u] = 0; pf1[u] = 0;
                    if (idx < fetch_end) fetch(idx, pf0[u], pf1[u]);
                }
                // A block opens with two round trips to memory in a row -- its descriptor, then the first CAP records -- before its first
                // tile can be marked (phase clocks: 15 % of a wave's time).  In the block's LAST tile the lines of the next block's first
                // records are asked for (one 4-byte load per line, nobody needs the value: the tile's full barrier waits for it with
                // everything else), so that the staging finds them in the L2.
                uint32_t touched = 0;
                if (ZK_EXEC_TOUCH && te >= out_size && tid < nx_lines && tid < (uint32_t)(CAP / 8)) touched = *reinterpret_cast<const uint32_t *>(nx_sq + 8u * tid);
                const uint32_t next_prev_end = jn ? zk_uni(S[(ja + jn - 1) & M].out_end) : prev_end;
                // (the same for the LITERALS of the next tile -- a byte per 64-byte line, 2 KiB from where this tile's sequences end -- measured:
                //  6.72 -> 6.91 ms; their lines are not what the gathers wait for, and the touch is one more load per tile.  r6at.sh)
                for (uint32_t k = 0; k < nlong; k++) {       // sequences spanning many slots: all lanes
                    const uint32_t idx = longlist[k];
                    const uint32_t end = S[idx & M].out_end;
                    const uint32_t start = idx != ja ? S[(idx - 1) & M].out_end : prev_end;
                    const uint32_t lo = start > ts ? start : ts, hi = end < te ? end : te;
                    uint32_t s0, n;
                    zk_exec_slot_span(ts, lo, hi, s0, n);
                    for (uint32_t j = tid; j < n; j += T) slot_seq[s0 + j] = idx;
                }
                if (nlong) ZK_LDS_BARRIER();
                // 3. lane per slot: source words of its 16 bytes
                const uint32_t q0 = ts + tid * ZK_EXEC_B;
                const uint32_t nb = q0 >= te ? 0u : te - q0 < (uint32_t)ZK_EXEC_B ? te - q0 : (uint32_t)ZK_EXEC_B;
                uint32_t sw[ZK_EXEC_B];
                if (nb) {
                    {
                        uint32_t mk[ZK_EXEC_B];
#pragma unroll
                        for (int k = 0; k < ZK_EXEC_B; k += 4) {
                            const uint4 v = *reinterpret_cast<const uint4 *>(&srcmap[zk_exec_map_index(tid * ZK_EXEC_B + k)]);
                            mk[k] = v.x; mk[k + 1] = v.y; mk[k + 2] = v.z; mk[k + 3] = v.w;
                        }
                        // (the slot's first mark is never read; said to be, so that the reads stay four of 16 bytes: narrowed to the fifteen words
                        //  that are, they became eight ds_read2_b32 at a lane stride of 64 bytes -- two banks for 64 lanes -- and the executor took
                        //  9.4 instead of 7.4 ms)
                        asm volatile("" :: "v"(mk[0]));
                        zk_exec_slot_words_marked(S, slot_seq[tid], (s_slow[tid >> 5] >> (tid & 31u)) & 1u, q0, nb, mk, sw, M);
                    }
#pragma unroll
                    for (int k = 0; k < ZK_EXEC_B; k += 4)
                        *reinterpret_cast<uint4 *>(&srcmap[zk_exec_map_index(tid * ZK_EXEC_B + k)]) = make_uint4(sw[k], sw[k + 1], sw[k + 2], sw[k + 3]);
                }
                ZK_CLK(3);
                __syncthreads();                                                 // the full one: every wave's stores of the tile before are in memory
                if (ZK_EXEC_TOUCH) asm volatile("" :: "v"(touched));             // (the touch is complete here; its value goes nowhere)
                ZK_CLK(2);
                if (tid == 0) { s_jn = fetch_end - (ja + jn); s_nlong = 0; }      // the next tile's marking pass starts from these (every lane has read this tile's)
                // the slot pass is done with the retired records: the fetched ones move in
#pragma unroll
                for (int u = 0; u < NPF; u++) {
                    const uint32_t idx = staged_end + tid + (uint32_t)u * T;
                    if (idx < fetch_end) settle(idx, pf0[u], pf1[u], bad);
                }
                ZK_CLK(6);
                // 4. origins (in-tile history words are exactly [BIAS + ts, BIAS + te)), gathers, one coalesced store
                if (nb) {
                    ZK_EXEC_CHASE(sw, srcmap, ts, te);
                    ZK_CLK(4);
                    // A byte's address = its source word + one of two bases (the words' tags folded into them): a compare, two selects
                    // and a 64-bit add, and a GLOBAL load -- the bases went through zk_uni (an integer), which left the compiler
                    // with generic pointers: flat loads, and twelve instructions per byte.  Words past the tile's end are ZK_SRC_LIT
                    // (zk_exec_slot_words*): literal 0, always readable.  RLE literals: every literal word becomes literal 0.
                    if (lit_mask == 0) {
#pragma unroll
                        for (int k = 0; k < ZK_EXEC_B; k++) sw[k] = (sw[k] & ZK_SRC_LIT) ? ZK_SRC_LIT : sw[k];
                    }
                    uint32_t ov[ZK_EXEC_B / 4];                       // the slot's sixteen bytes
                    typedef const __attribute__((address_space(1))) uint8_t *gbyte_t;
                    if (!PFX && ZK_EXEC_WIDE && pos + te + 3 <= d_size) {
                        // FOUR BYTES PER LOOKUP where a run allows it.  The L1 takes one lookup per LANE of a byte load whatever the
                        // addresses (TCP_TOTAL_CACHE_ACCESSES: 4.29 G for 4 GiB of output, one per cycle and CU for the whole kernel: the
                        // unit that is full, profiles/r06_exec_gather_probe.txt).  Of a lane's four groups of four bytes about half lie
                        // inside one run -- four consecutive source words: one unaligned 4-byte load at the first word's address brings
                        // all four; the other lanes keep its low byte and load three more bytes, with the first group of lanes masked
                        // off.  The load reads up to three bytes past a source: literals have that slack (ZK_COMP_PADDING, the scratch's
                        // own), history has it unless the frame ends within three bytes of the tile (then: the loop below).
                        uint32_t lone = 0;                            // bit g: group g is not one run
                        uint32_t b1[ZK_EXEC_B / 4], b2[ZK_EXEC_B / 4], b3[ZK_EXEC_B / 4];
#pragma unroll
                        for (int g = 0; g < ZK_EXEC_B / 4; g++) {
                            const uint32_t s0 = sw[4 * g];
                            const bool run = ((sw[4 * g + 1] - s0) == 1u) & ((sw[4 * g + 2] - s0) == 2u) & ((sw[4 * g + 3] - s0) == 3u);
                            uint32_t v;
                            __builtin_memcpy(&v, (const void *)(gbyte_t)(((int32_t)s0 < 0 ? lit_w : his_w) + s0), 4);
                            ov[g] = v; b1[g] = 0; b2[g] = 0; b3[g] = 0;
                            if (!run) {
                                lone |= 1u << g;
                                const uint32_t s1 = sw[4 * g + 1], s2 = sw[4 * g + 2], s3 = sw[4 * g + 3];
                                b1[g] = *(gbyte_t)(((int32_t)s1 < 0 ? lit_w : his_w) + s1);
                                b2[g] = *(gbyte_t)(((int32_t)s2 < 0 ? lit_w : his_w) + s2);
                                b3[g] = *(gbyte_t)(((int32_t)s3 < 0 ? lit_w : his_w) + s3);
                            }
                        }
                        // (byte permutes, not shifts: a shift of "the loaded byte, or 0 where the loads were skipped" is moved INTO the masked
                        //  region by the compiler, with a wait for the load in front of it -- four round trips to memory in a row)
#pragma unroll
                        for (int g = 0; g < ZK_EXEC_B / 4; g++) {
                            const uint32_t lo = __builtin_amdgcn_perm(b1[g], ov[g], 0x0c0c0400u), hi = __builtin_amdgcn_perm(b3[g], b2[g], 0x04000c0cu);
                            ov[g] = (lone >> g) & 1u ? lo | hi : ov[g];
                        }
                    } else {
                        uint32_t ob[ZK_EXEC_B];
#pragma unroll
                        for (int k = 0; k < ZK_EXEC_B; k++) {
                            const uint32_t s = sw[k];
                            uint64_t a = ((int32_t)s < 0 ? lit_w : his_w) + s;
                            if (PFX && (int32_t)s >= 0) {        // a position before the frame's first byte lies in the prefix
                                const int64_t rel = (int64_t)pos + (int32_t)(s - ZK_SRC_BIAS);
                                if (rel < 0) a = (uint64_t)(uintptr_t)prefix + plen + rel;
                            }
                            ob[k] = *(gbyte_t)a;
                        }
#pragma unroll
                        for (int g = 0; g < ZK_EXEC_B / 4; g++) ov[g] = ob[4 * g] | (ob[4 * g + 1] << 8) | (ob[4 * g + 2] << 16) | (ob[4 * g + 3] << 24);
                    }
                    uint8_t *w = bout + q0;
#ifdef ZK_EXEC_CLOCKS
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ZK_CLK(7);
#endif
                    if (nb == ZK_EXEC_B && (((uintptr_t)w) & 15) == 0) *reinterpret_cast<uint4 *>(w) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
                    else {
#pragma unroll
                        for (int k = 0; k < ZK_EXEC_B; k++) if ((uint32_t)k < nb) w[k] = (uint8_t)(ov[k >> 2] >> (8 * (k & 3)));
                    }
                }
                prev_end = next_prev_end;
                ja += jn; staged_end = fetch_end; ts = te;
                ZK_CLK(5);
                // the ring complete, the map free again (the tile's bytes: see the barrier in front of the gathers)
                if (bad) s_bad[tpar] = 1;
                ZK_LDS_BARRIER();
                if (s_bad[tpar]) { err = ZK_E_CORRUPTION; break; }
                tpar ^= 1;
                ZK_CLK(2);
            }
            if (err == ZK_OK) {
                uint32_t r0 = zk_rep_resolve(zk_uni(b.rep_out[0]), rep), r1 = zk_rep_resolve(zk_uni(b.rep_out[1]), rep), r2 = zk_rep_resolve(zk_uni(b.rep_out[2]), rep);
                rep[0] = r0; rep[1] = r1; rep[2] = r2;
            }
        }
        pos += b_out_size;
        //  __syncthreads() orders what the waves of THIS workgroup see (one L1, processed in order: the compiler emits no
        // vmcnt wait for it); the checksum wave of another kernel reads through the XCD's L2, where a store is only once it has been
        // acknowledged -- so with progress words every wave waits out its own stores before the barrier in front of the publish, once
        // per block.  (Otherwise bytes of lane 0's wave alone were known to have landed, and a decode repeated into the same buffer
        // could be "verified" on the bytes of the decode before.)
        if (progress) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // block bytes visible before the next block reads history
        if (progress && ((uint32_t)pos >> ZK_PUB_LOG) != published) {
            published = (uint32_t)pos >> ZK_PUB_LOG;
            if (tid == 0) zk_publish(progress + f, (uint32_t)pos);
        }
    }
    if (err == ZK_OK && pos != d_size) err = ZK_E_CORRUPTION;
    if (tid == 0 && (err != ZK_OK || REDO)) infos[f].status = err;
    if (progress && tid == 0) {                              // (a frame that ends well: behind the last block's barrier)
        if (err == ZK_OK) zk_publish(progress + f, (uint32_t)d_size);
        else zk_publish(progress + f, 0, ZK_PROG_ABORT);
    }
#ifdef ZK_EXEC_CLOCKS
    ZK_CLK(0);
    if ((tid & 63) == 0) for (int i = 0; i < 8; i++) atomicAdd(&zk_dbg_clk[i], clk_[i]);
#endif
}

// ------------------------------------------------------------------------------------------------ sequence execution in segments
// (zk_device.h, "sequence execution in SEGMENTS", has the scheme.)  Three kernels behind the entropy stage:
//   zk_k_seg_prep   one wave per frame: the frame's blocks cut into segments (each block's first byte and repeat offsets are a
//                   serial walk over ~16-64 block descriptors), the frame-level checks of zk_k_exec;
//   zk_k_exec_seg   one workgroup per SEGMENT: zk_k_exec's tile loop; bytes that depend on anything before the segment are left
//                   as hole records (per tile: how many), everything else is final;
//   zk_k_exec_fill  one workgroup per frame: the hole records tile by tile (a tile's holes copy from before the tile: independent).
// A segment whose records do not fit its region marks its frame ZK_E_SEG_OVERFLOW; zk_k_exec<REDO> executes those frames again.
__global__ __launch_bounds__(64) void zk_k_seg_prep(const ZkBlock *blocks, const ZkFrameBase *bases, ZkFrameInfo *infos, const uint64_t *d_off, uint32_t first,
                                                    const uint32_t *ids, uint32_t seg_bytes, ZkSeg *segs, uint32_t *nsegs, uint32_t max_segs)
{
    const uint32_t f = blockIdx.x, lane = threadIdx.x;
    const ZkFrameInfo fi = infos[f];
    if (fi.status != ZK_OK) { if (lane == 0) nsegs[f] = 0; return; }
    const uint32_t id = ids ? ids[f] : first + f;
    const uint64_t d_size = d_off[id + 1] - d_off[id];
    const uint32_t block_max = fi.window < ZK_BLOCK_MAX ? fi.window : ZK_BLOCK_MAX;
    const ZkBlock *fb = blocks + bases[f].block_base;
    ZkSeg *fs = segs + (size_t)f * max_segs;
    ZkSegWalk w;
    zk_seg_walk_init(w);
    for (uint32_t base = 0; base < fi.n_blocks; base += 64) {
        const uint32_t bk = base + lane;
        uint32_t st = 0, out = 0, r0 = 0, r1 = 0, r2 = 0;
        if (bk < fi.n_blocks) { const ZkBlock &b = fb[bk]; st = b.status; out = b.out_size; r0 = b.rep_out[0]; r1 = b.rep_out[1]; r2 = b.rep_out[2]; }
        const uint32_t n = fi.n_blocks - base < 64u ? fi.n_blocks - base : 64u;
        for (uint32_t i = 0; i < n; i++) {                   // every lane runs the same walk (the values come out of lane i's registers); all write the same records
            const uint32_t rp[3] = {(uint32_t)__shfl(r0, (int)i, 64), (uint32_t)__shfl(r1, (int)i, 64), (uint32_t)__shfl(r2, (int)i, 64)};
            zk_seg_step(w, base + i, (uint32_t)__shfl(st, (int)i, 64), (uint32_t)__shfl(out, (int)i, 64), rp, d_size, block_max, seg_bytes, fs, max_segs);
        }
    }
    zk_seg_walk_end(w, d_size, fs, max_segs);
    if (lane == 0) {
        nsegs[f] = w.err == ZK_OK ? w.nsegs : 0u;
        if (w.err != ZK_OK) infos[f].status = w.err;
    }
}

// where a segment's tile counts live (u32 entries) and how many it may leave: at most one per 4 KiB of output plus one per 512
// sequences... bounded by out / 1024 + 2 per block (a tile ends early only when its ring of >= 512 records covers less than the tile)
__device__ __forceinline__ uint64_t zk_seg_tile_region(uint64_t frame_off, uint32_t pos, uint64_t first_block, uint64_t seg_no) { return ((frame_off + pos) >> 10) + 2u * first_block + seg_no * 8u; }
__device__ __forceinline__ uint32_t zk_seg_tile_cap(uint32_t out, uint32_t nb) { return (out >> 10) + 2u * nb + 8u; }

template <int T, int CAPX>
__global__ __launch_bounds__(T) void zk_k_exec_seg(const uint8_t *comp, const uint64_t *d_off, uint32_t first, const uint32_t *ids, const uint64_t *out_off,
                                                   const ZkBlock *blocks, const ZkFrameBase *bases, ZkFrameInfo *infos, const ZkSeqP *seqs,
                                                   const uint8_t *lit_scratch, uint8_t *dst, const ZkSeg *segs, const uint32_t *nsegs, uint32_t max_segs,
                                                   ZkHole *holes, uint32_t *tilecnt, uint32_t *segn)
{
    constexpr int CAP = CAPX * T;
    constexpr uint32_t M = CAP - 1;
    constexpr int NPF = CAP / T;
    __shared__ __attribute__((aligned(16))) ZkSeq S[CAP];
    __shared__ __attribute__((aligned(16))) uint32_t srcmap[T * ZK_EXEC_B];
    __shared__ __attribute__((aligned(16))) uint32_t taint[ZK_SEG_BYTES / 32 + 4];      // one bit per byte of the segment: the byte is a hole
    __shared__ uint32_t slot_seq[T];
    __shared__ uint32_t longlist[CAP + 1];
    __shared__ uint32_t s_slow[(T + 31) / 32];   // the tile's slots that hold bytes of a match overlapping its own output (zk_exec_mark_runs)
    __shared__ uint32_t s_jn, s_nlong, s_nrec;
    __shared__ uint32_t s_bad[2];
    const uint32_t f = blockIdx.x, sgi = blockIdx.y, tid = threadIdx.x;
    ZkFrameInfo fi = infos[f];
    fi.status = zk_uni(fi.status); fi.window = zk_uni(fi.window);
    if (fi.status != ZK_OK) return;                          // (refused by the walk or by zk_k_seg_prep; or another segment has just found an error)
    if (sgi >= zk_uni(nsegs[f])) return;
    const uint64_t seg_no = (uint64_t)f * max_segs + sgi;
    const ZkSeg &sg = segs[seg_no];
    const uint32_t seg_pos = zk_uni(sg.pos), seg_out = zk_uni(sg.out), seg_nb = zk_uni(sg.nb), seg_b0 = zk_uni(sg.b0);
    const uint32_t id = zk_uni(ids ? ids[f] : first + f);
    const uint64_t frame_off = zk_uni(out_off ? out_off[f] : d_off[id] - d_off[first]);
    uint8_t *out = dst + frame_off;
    const uint64_t block_base = zk_uni(bases[f].block_base);
    const ZkBlock *fb = blocks + (block_base + seg_b0);
    ZkHole *rec = holes + zk_uni((uint64_t)zk_seg_region(frame_off, seg_pos, seg_no));
    const uint32_t rec_cap = zk_seg_region_cap(seg_out);
    uint32_t *tc = tilecnt + zk_uni((uint64_t)zk_seg_tile_region(frame_off, seg_pos, block_base + seg_b0, seg_no));
    const uint32_t tc_cap = zk_seg_tile_cap(seg_out, seg_nb);
    uint32_t nrec = 0, ntile = 0;                            // records / tile counts left so far (uniform)
    bool overflow = false;
    for (uint32_t i = tid; i < (seg_out >> 5) + 2u; i += T) taint[i] = 0;
    uint64_t pos = seg_pos;
    uint32_t rep[3] = {zk_uni(sg.rep[0]), zk_uni(sg.rep[1]), zk_uni(sg.rep[2])};
    uint32_t err = ZK_OK;
    __syncthreads();

    for (uint32_t bk = 0; bk < seg_nb && err == ZK_OK; bk++) {
        const ZkBlock &b = fb[bk];
        const uint32_t b_out_size = zk_uni(b.out_size), b_type = zk_uni((uint32_t)b.type);      // (status and sizes: zk_k_seg_prep has checked them)
        uint8_t *bout = out + pos;
        const uint32_t seg_done = (uint32_t)(pos - seg_pos);  // bytes of the segment in front of this block
        if (b_type <= 1) {
            const uint8_t *s = comp + zk_uni(b.src);
            const uint32_t n = zk_uni(b.bsize);
            const uint32_t head0 = (uint32_t)((0 - (uintptr_t)bout) & 15), head = head0 < n ? head0 : n;
            const uint32_t n16 = (n - head) >> 4;
            if (b_type == 0) {
                if (tid < head) bout[tid] = s[tid];
                for (uint32_t i = tid; i < n16; i += T) {
                    uint4 v;
                    __builtin_memcpy(&v, s + head + (i << 4), 16);
                    *reinterpret_cast<uint4 *>(bout + head + (i << 4)) = v;
                }
                for (uint32_t i = head + (n16 << 4) + tid; i < n; i += T) bout[i] = s[i];
            } else {
                const uint32_t v1 = s[0], v4 = v1 * 0x01010101u;
                if (tid < head) bout[tid] = (uint8_t)v1;
                for (uint32_t i = tid; i < n16; i += T) *reinterpret_cast<uint4 *>(bout + head + (i << 4)) = make_uint4(v4, v4, v4, v4);
                for (uint32_t i = head + (n16 << 4) + tid; i < n; i += T) bout[i] = (uint8_t)v1;
            }
        } else {
            const ZkSeqP *sq = seqs + zk_uni(b.seq_base);
            const uint32_t b_lit_type = zk_uni((uint32_t)b.lit_type);
            const uint8_t *lit = b_lit_type >= 2 ? lit_scratch + zk_uni(b.lit_base) : comp + zk_uni(b.src) + zk_uni(b.lit_off);
            const uint32_t lit_mask = b_lit_type == 1 ? 0u : 0x7fffffffu;
            const uint64_t lit_w = (uint64_t)(uintptr_t)lit - ZK_SRC_LIT, his_w = (uint64_t)(uintptr_t)bout - ZK_SRC_BIAS;
            const uint32_t nseq = zk_uni(b.nseq), out_size = b_out_size, lit_regen = zk_uni(b.lit_regen);
            const int32_t seg_lo = -(int32_t)seg_done;       // block-relative position of the segment's first byte
            auto fetch = [&](uint32_t idx, ZkSeqP &p0, ZkSeqP &p1) {
                p1 = idx < nseq ? sq[idx] : 0;
                p0 = idx && idx < nseq ? sq[idx - 1] : 0;
            };
            auto settle = [&](uint32_t idx, ZkSeqP p0, ZkSeqP p1, int &bad) {
                uint4 r;
                if (idx < nseq) {
                    const ZkSeq q = zk_seq_unpack(p0, p1, idx == 0);
                    const uint32_t off = zk_rep_resolve(q.off, rep);
                    const uint32_t mstart = q.out_end - q.ml;
                    if (off == 0 || pos + mstart < off || off > fi.window) bad = 1;
                    if (off >= ZK_SRC_BIAS || q.ml > q.out_end) bad = 1;
                    r = make_uint4(q.out_end, q.ml, off, q.lit_end);
                } else r = make_uint4(out_size, 0, 1, lit_regen);
                reinterpret_cast<uint4 *>(S)[idx & M] = r;
            };
            int bad = 0;
            uint32_t ja = 0, ts = 0, prev_end = 0;
            uint32_t staged_end = nseq + 1 < (uint32_t)CAP ? nseq + 1 : (uint32_t)CAP;
            for (uint32_t idx = tid; idx < staged_end; idx += T) { ZkSeqP p0, p1; fetch(idx, p0, p1); settle(idx, p0, p1, bad); }
            if (tid == 0) { s_jn = staged_end; s_nlong = 0; s_bad[0] = 0; s_bad[1] = 0; s_nrec = 0; }
            if (__syncthreads_or(bad)) { err = ZK_E_CORRUPTION; }
            uint32_t tpar = 0;
            while (err == ZK_OK && ts < out_size) {
                const uint32_t nl = staged_end - ja;
                const uint32_t cap_end = zk_uni(S[(staged_end - 1) & M].out_end);
                const uint32_t te = ts + T * ZK_EXEC_B < cap_end ? ts + T * ZK_EXEC_B : cap_end;
#pragma unroll
                for (int k = 0; k < ZK_EXEC_B; k += 4) *reinterpret_cast<uint4 *>(&srcmap[zk_exec_map_index(tid * ZK_EXEC_B + k)]) = make_uint4(0, 0, 0, 0);
                if (tid < (T + 31) / 32) s_slow[tid] = 0;
                ZK_LDS_BARRIER();
                if (tid == 0) { s_bad[tpar ^ 1] = 0; s_nrec = 0; }                // (the tile before: every wave has read both in front of this barrier)
                for (uint32_t i = tid; i < nl; i += T) {
                    const uint32_t idx = ja + i;
                    const ZkSeq me = S[idx & M];
                    const uint32_t end = me.out_end;
                    uint32_t start = prev_end, prev_off = 1;                      // (the offset of a sequence that has left the ring is never needed: zk_exec_mark_runs)
                    if (i) { const ZkSeq &pv = S[(idx - 1) & M]; start = pv.out_end; prev_off = pv.off; }
                    const uint32_t lo = start > ts ? start : ts, hi = end < te ? end : te;
                    if (lo < hi) {
                        uint32_t s0, n;
                        zk_exec_slot_span(ts, lo, hi, s0, n);
                        if (n > ZK_EXEC_LONG) longlist[atomicAdd(&s_nlong, 1u)] = idx;
                        else {                                   // (written out: as a loop it was two loops, pairs and a remainder, with the mask juggling of both)
#pragma unroll
                            for (uint32_t k = 0; k < ZK_EXEC_LONG; k++) if (k < n) slot_seq[s0 + k] = idx;
                        }
                        zk_exec_mark_runs(me, prev_off, start, ts, te, srcmap, [&](uint32_t w, uint32_t bits) { atomicOr(&s_slow[w], bits); });
                    }
                    if (end > te && start <= te) s_jn = i;
                }
                ZK_LDS_BARRIER();
                const uint32_t nlong = zk_uni(s_nlong), jn = zk_uni(s_jn);
                const uint32_t fetch_end = staged_end + jn < nseq + 1 ? staged_end + jn : nseq + 1;
                ZkSeqP pf0[NPF], pf1[NPF];
#pragma unroll
                for (int u = 0; u < NPF; u++) {
                    const uint32_t idx = staged_end + tid + (uint32_t)u * T;
                    pf0[u] = 0; pf1[u] = 0;
                    if (idx < fetch_end) fetch(idx, pf0[u], pf1[u]);
                }
                const uint32_t next_prev_end = jn ? zk_uni(S[(ja + jn - 1) & M].out_end) : prev_end;
                for (uint32_t k = 0; k < nlong; k++) {
                    const uint32_t idx = longlist[k];
                    const uint32_t end = S[idx & M].out_end;
                    const uint32_t start = idx != ja ? S[(idx - 1) & M].out_end : prev_end;
                    const uint32_t lo = start > ts ? start : ts, hi = end < te ? end : te;
                    uint32_t s0, n;
                    zk_exec_slot_span(ts, lo, hi, s0, n);
                    for (uint32_t j = tid; j < n; j += T) slot_seq[s0 + j] = idx;
                }
                if (nlong) ZK_LDS_BARRIER();
                const uint32_t q0 = ts + tid * ZK_EXEC_B;
                const uint32_t nb = q0 >= te ? 0u : te - q0 < (uint32_t)ZK_EXEC_B ? te - q0 : (uint32_t)ZK_EXEC_B;
                uint32_t sw[ZK_EXEC_B];
                if (nb) {
                    {
                        uint32_t mk[ZK_EXEC_B];
#pragma unroll
                        for (int k = 0; k < ZK_EXEC_B; k += 4) {
                            const uint4 v = *reinterpret_cast<const uint4 *>(&srcmap[zk_exec_map_index(tid * ZK_EXEC_B + k)]);
                            mk[k] = v.x; mk[k + 1] = v.y; mk[k + 2] = v.z; mk[k + 3] = v.w;
                        }
                        // (the slot's first mark is never read; said to be, so that the reads stay four of 16 bytes: narrowed to the fifteen words
                        //  that are, they became eight ds_read2_b32 at a lane stride of 64 bytes -- two banks for 64 lanes -- and the executor took
                        //  9.4 instead of 7.4 ms)
                        asm volatile("" :: "v"(mk[0]));
                        zk_exec_slot_words_marked(S, slot_seq[tid], (s_slow[tid >> 5] >> (tid & 31u)) & 1u, q0, nb, mk, sw, M);
                    }
#pragma unroll
                    for (int k = 0; k < ZK_EXEC_B; k += 4)
                        *reinterpret_cast<uint4 *>(&srcmap[zk_exec_map_index(tid * ZK_EXEC_B + k)]) = make_uint4(sw[k], sw[k + 1], sw[k + 2], sw[k + 3]);
                }
                __syncthreads();                                                 // the full one: every wave's stores of the tile before are in memory
                if (tid == 0) { s_jn = fetch_end - (ja + jn); s_nlong = 0; }
#pragma unroll
                for (int u = 0; u < NPF; u++) {
                    const uint32_t idx = staged_end + tid + (uint32_t)u * T;
                    if (idx < fetch_end) settle(idx, pf0[u], pf1[u], bad);
                }
                // 4. origins, then: which bytes are holes (their origin lies before the segment, or at a hole of an earlier tile), gathers for
                //    the others, one coalesced store (a hole's byte of the store is whatever: zk_k_exec_fill writes it), the holes' runs as records
                uint32_t hm = 0, starts = 0, len[ZK_EXEC_B];
                if (nb) {
                    ZK_EXEC_CHASE(sw, srcmap, ts, te);
                    hm = zk_seg_slot_holes(sw, nb, seg_lo, [&](uint32_t p) { return (bool)((taint[p >> 5] >> (p & 31u)) & 1u); });
                    starts = zk_seg_slot_runs(sw, hm, len);
                    uint32_t ob[ZK_EXEC_B];
                    // (the gathers as in zk_k_exec: a word + one of two bases; a hole reads literal 0 like the words past the tile's end)
#pragma unroll
                    for (int k = 0; k < ZK_EXEC_B; k++) {
                        uint32_t s = sw[k];
                        if (lit_mask == 0) s = (s & ZK_SRC_LIT) ? ZK_SRC_LIT : s;
                        s = ((hm >> k) & 1u) ? ZK_SRC_LIT : s;
                        ob[k] = *(const __attribute__((address_space(1))) uint8_t *)(((int32_t)s < 0 ? lit_w : his_w) + s);
                    }
                    uint8_t *w = bout + q0;
                    if (nb == ZK_EXEC_B && (((uintptr_t)w) & 15) == 0) {
                        uint4 v;
                        v.x = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
                        v.y = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
                        v.z = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
                        v.w = ob[12] | (ob[13] << 8) | (ob[14] << 16) | (ob[15] << 24);
                        *reinterpret_cast<uint4 *>(w) = v;
                    } else {
#pragma unroll
                        for (int k = 0; k < ZK_EXEC_B; k++) if ((uint32_t)k < nb) w[k] = (uint8_t)ob[k];
                    }
                    if (hm) {                                                    // the slot's taint bits (readers: later tiles, behind this tile's last barrier)
                        const uint32_t p0 = seg_done + q0, sh = p0 & 31u;
                        atomicOr(&taint[p0 >> 5], hm << sh);
                        if (sh > 16u) atomicOr(&taint[(p0 >> 5) + 1u], hm >> (32u - sh));
                    }
                }
                // the tile's records: a wave reserves its lanes' records with ONE LDS atomic (the order of a tile's records is whatever the
                // waves make it: zk_k_exec_fill copies a tile's records in any order)
                {
                    const uint32_t cnt = (uint32_t)__popc(starts);
                    uint32_t inc = cnt;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if ((int)(tid & 63) >= d) inc += y; }
                    const uint32_t wtot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    if (wtot) {
                        uint32_t wbase = 0;
                        if ((tid & 63) == 0) wbase = atomicAdd(&s_nrec, wtot);
                        wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
                        uint32_t at = nrec + wbase + inc - cnt;
                        const uint32_t dpos = (uint32_t)pos + q0;                  // frame-relative position of the slot's first byte
#pragma unroll
                        for (int k = 0; k < ZK_EXEC_B; k++)
                            if ((starts >> k) & 1u) {
                                if (at < rec_cap) rec[at] = zk_hole_pack(dpos + (uint32_t)k, len[k], q0 + (uint32_t)k + ZK_SRC_BIAS - sw[k]);
                                at++;
                            }
                    }
                }
                prev_end = next_prev_end;
                ja += jn; staged_end = fetch_end; ts = te;
                if (bad) s_bad[tpar] = 1;
                ZK_LDS_BARRIER();
                if (s_bad[tpar]) { err = ZK_E_CORRUPTION; break; }
                {
                    const uint32_t tn = zk_uni(s_nrec);
                    if (tn) {
                        if (nrec + tn > rec_cap || ntile >= tc_cap) overflow = true;
                        else if (tid == 0) tc[ntile] = tn;
                        nrec += tn; ntile++;
                    }
                }
                tpar ^= 1;
            }
            if (err == ZK_OK) {
                uint32_t r0 = zk_rep_resolve(zk_uni(b.rep_out[0]), rep), r1 = zk_rep_resolve(zk_uni(b.rep_out[1]), rep), r2 = zk_rep_resolve(zk_uni(b.rep_out[2]), rep);
                rep[0] = r0; rep[1] = r1; rep[2] = r2;
            }
        }
        pos += b_out_size;
        __syncthreads();                      // block bytes visible before the next block reads history
    }
    if (tid == 0) {
        segn[seg_no] = overflow ? 0u : ntile;
        if (err != ZK_OK) atomicCAS(&infos[f].status, (uint32_t)ZK_OK, err);
        else if (overflow) atomicCAS(&infos[f].status, (uint32_t)ZK_OK, ZK_E_SEG_OVERFLOW);
    }
}

// pass 2: one workgroup of L lanes per frame.  Segment after segment, tile after tile: a tile's records are independent of each other
// (their sources lie before the tile) -- one round of loads, one of stores, and the stores are waited out before the next tile reads.
template <int L>
__global__ __launch_bounds__(L) void zk_k_exec_fill(const uint64_t *d_off, uint32_t first, const uint32_t *ids, const uint64_t *out_off, const ZkFrameBase *bases,
                                                    const ZkFrameInfo *infos, uint8_t *dst, const ZkSeg *segs, const uint32_t *nsegs, uint32_t max_segs,
                                                    const ZkHole *holes, const uint32_t *tilecnt, const uint32_t *segn, uint64_t *progress)
{
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    if (zk_uni(infos[f].status) != ZK_OK) {
        if (progress && tid == 0) zk_publish(progress + f, 0, ZK_PROG_ABORT);
        return;
    }
    const uint32_t id = zk_uni(ids ? ids[f] : first + f);
    const uint64_t frame_off = zk_uni(out_off ? out_off[f] : d_off[id] - d_off[first]);
    uint8_t *out = dst + frame_off;
    const uint64_t block_base = zk_uni(bases[f].block_base);
    const uint32_t ns = zk_uni(nsegs[f]);
    if (progress && tid == 0) zk_publish(progress + f, 0);   // (which XCD this is)
    for (uint32_t j = 0; j < ns; j++) {
        const uint64_t seg_no = (uint64_t)f * max_segs + j;
        const ZkSeg &sg = segs[seg_no];
        const uint32_t seg_pos = zk_uni(sg.pos);
        const uint32_t nt = zk_uni(segn[seg_no]);
        const ZkHole *rec = holes + zk_uni((uint64_t)zk_seg_region(frame_off, seg_pos, seg_no));
        const uint32_t *tc = tilecnt + zk_uni((uint64_t)zk_seg_tile_region(frame_off, seg_pos, block_base + zk_uni(sg.b0), seg_no));
        if (progress && j && tid == 0) zk_publish(progress + f, seg_pos);        // everything in front of this segment is final (every wave has waited out its stores at the tile barriers)
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t cnt = zk_uni(tc[t]);
            for (uint32_t i = tid; i < cnt; i += L) {
                const ZkHole r = rec[i];
                uint8_t *d = out + zk_hole_dst(r);
                const uint8_t *s = d - zk_hole_off(r);
                const uint32_t n = zk_hole_len(r);
                uint32_t v[ZK_EXEC_B];
#pragma unroll
                for (int k = 0; k < ZK_EXEC_B; k++) v[k] = (uint32_t)k < n ? s[k] : 0u;
#pragma unroll
                for (int k = 0; k < ZK_EXEC_B; k++) if ((uint32_t)k < n) d[k] = (uint8_t)v[k];
            }
            rec += cnt;
            __syncthreads();                  // (waits out the stores: the next tile's sources may be these bytes)
        }
    }
    if (progress && tid == 0) zk_publish(progress + f, (uint32_t)(d_off[id + 1] - d_off[id]));
}

// The same with the SEGMENT in LDS (a handful of long frames: a round of zk_k_exec_fill is a trip to memory and back, ~1.4-3 us, and a
// 2 MiB frame has 128 ... 512 of them).  A hole's source is either final (before the segment: memory) or a hole of an earlier tile of
// the segment.  Per segment:
//   0  the segment as pass 1 left it: memory -> a 128 KiB image in LDS (16 bytes per lane and step);
//   A  its records: requested at once, kept in registers (the common shape: at most ZK_FILL_NT tiles of at most 2 L records);
//   B  every byte whose source lies before the segment: memory -> image, no order among them;
//   C  tile after tile, the bytes whose source is a hole of the segment: image -> image behind an LDS barrier (no memory on the path);
//   E  the image -> memory.
// A record moves as two 8-byte reads and its bytes as byte writes (a run is 1 ... 16 bytes at any alignment; the image's neighbours
// belong to other lanes).  Other shapes take the tile loop with their records from memory.  One workgroup of L = 1024 lanes per frame.
// (Measured, round 6: a 2 MiB frame = 16 turns of ~26 us = 0.42 ms.  A turn rebuilt as TWO trips to memory -- every segment's place and
//  tile counts read ahead, the image by LDS DMA in the same trip as the records, the stores of the turn before waited out by the loads
//  behind them -- took the same 0.42 ms (tools/gpu_calls/r6v.sh): the turn is what its 16 waves issue for ~11 000 runs of ~5 bytes, byte
//  writes each, not what it waits for.  The simpler form stays.)
constexpr int ZK_FILL_NT = 12;
constexpr uint32_t ZK_FILL_IMG = ZK_SEG_BYTES + 48;                              // up to 15 bytes of alignment in front, 16-byte reads behind
// bytes [lo, hi) of the 16-byte value (w0, w1) -> img[at + lo .. at + hi)
__device__ __forceinline__ void zk_img_put(uint8_t *img, uint32_t at, uint32_t lo, uint32_t hi, uint64_t w0, uint64_t w1)
{
#pragma unroll
    for (int k = 0; k < 8; k++) if ((uint32_t)k >= lo && (uint32_t)k < hi) img[at + k] = (uint8_t)(w0 >> (8 * k));
    if (__ballot(hi > 8u) == 0) return;                                          // (runs are ~5 bytes on average: the upper half is rare)
#pragma unroll
    for (int k = 8; k < 16; k++) if ((uint32_t)k >= lo && (uint32_t)k < hi) img[at + k] = (uint8_t)(w1 >> (8 * (k - 8)));
}
// B: the part of a record whose source lies before the segment.  flimit: the frame's size (a 16-byte read must not leave the frame)
__device__ __forceinline__ void zk_fill_from_memory(ZkHole r, uint32_t seg_pos, uint32_t ibase, uint32_t flimit, const uint8_t *out, uint8_t *img)
{
    const uint32_t d = zk_hole_dst(r), sp = d - zk_hole_off(r), n = zk_hole_len(r);
    const bool mine = r != 0 && sp < seg_pos;
    if (__ballot(mine) == 0) return;
    uint64_t w0 = 0, w1 = 0;
    if (mine) {
        if (sp + 16u <= flimit) { w0 = zk_ld64(out + sp); w1 = zk_ld64(out + sp + 8); }
        else for (uint32_t k = 0; k < n; k++) { const uint64_t b = out[sp + k]; if (k < 8) w0 |= b << (8 * k); else w1 |= b << (8 * (k - 8)); }
    }
    const uint32_t hi = mine ? (seg_pos - sp < n ? seg_pos - sp : n) : 0u;
    zk_img_put(img, d - ibase, 0u, hi, w0, w1);
}
// C: the part whose source is a hole of the segment (an earlier tile: in the image by now)
__device__ __forceinline__ void zk_fill_from_image(ZkHole r, uint32_t seg_pos, uint32_t ibase, uint8_t *img)
{
    const uint32_t d = zk_hole_dst(r), sp = d - zk_hole_off(r), n = zk_hole_len(r);
    const bool mine = r != 0 && sp + n > seg_pos;
    if (__ballot(mine) == 0) return;
    const uint32_t lo = mine ? (sp < seg_pos ? seg_pos - sp : 0u) : 16u;
    uint64_t w0 = 0, w1 = 0;
    if (mine) {                                                                  // (bytes below lo are not used: read from where the segment starts)
        const uint32_t a = sp + lo - ibase;
        uint64_t x0, x1;
        __builtin_memcpy(&x0, img + a, 8); __builtin_memcpy(&x1, img + a + 8, 8);
        // the value as if read from sp: shifted up by lo bytes
        if (lo == 0) { w0 = x0; w1 = x1; }
        else if (lo < 8) { w0 = x0 << (8 * lo); w1 = (x1 << (8 * lo)) | (x0 >> (64 - 8 * lo)); }
        else { w0 = 0; w1 = x0 << (8 * (lo - 8)); }
    }
    zk_img_put(img, d - ibase, lo, mine ? n : 0u, w0, w1);
}
// A TURN of the kernel = as many consecutive segments as fit the image (one 128 KiB segment of a long frame; all sixteen 4 KiB blocks of a
// 64 KiB frame: a seek): their bytes loaded once, their tiles in order -- segment after segment -- image -> image, the bytes stored once.
// The frame's first ZK_FILL_META segment descriptors and tile counts are read ahead into LDS (a turn of several segments would
// otherwise pay a trip to memory per segment before it can ask for that segment's records).
constexpr uint32_t ZK_FILL_META = 64, ZK_FILL_TC = 16;
template <int L>
__global__ __launch_bounds__(L) void zk_k_exec_fill_lds(const uint64_t *d_off, uint32_t first, const uint32_t *ids, const uint64_t *out_off, const ZkFrameBase *bases,
                                                        const ZkFrameInfo *infos, uint8_t *dst, const ZkSeg *segs, const uint32_t *nsegs, uint32_t max_segs,
                                                        const ZkHole *holes, const uint32_t *tilecnt, const uint32_t *segn, uint64_t *progress)
{
    __shared__ __attribute__((aligned(16))) uint8_t img[ZK_FILL_IMG];            // frame byte p of the turn at [p - ibase]
    __shared__ uint32_t s_cnt[L];
    __shared__ uint32_t s_meta[ZK_FILL_META][4];                                 // pos, out, first block, tiles with records
    __shared__ uint32_t s_tc[ZK_FILL_META][ZK_FILL_TC];
    static_assert(L >= (int)(ZK_FILL_META * ZK_FILL_TC), "one lane per tile count read ahead");
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    if (zk_uni(infos[f].status) != ZK_OK) {
        if (progress && tid == 0) zk_publish(progress + f, 0, ZK_PROG_ABORT);
        return;
    }
    if (progress && tid == 0) zk_publish(progress + f, 0);   // (which XCD this is)
    uint32_t final_to = 0;                                   // bytes in front of this position are final once the stores in flight have landed
    const uint32_t id = zk_uni(ids ? ids[f] : first + f);
    const uint64_t frame_off = zk_uni(out_off ? out_off[f] : d_off[id] - d_off[first]);
    const uint32_t flimit = (uint32_t)zk_uni(d_off[id + 1] - d_off[id]);
    uint8_t *out = dst + frame_off;
    const uint64_t block_base = zk_uni(bases[f].block_base);
    const uint32_t ns = zk_uni(nsegs[f]);
    const uint64_t seg0 = (uint64_t)f * max_segs;
    {
        const uint32_t nm = ns < ZK_FILL_META ? ns : ZK_FILL_META;
        if (tid < nm) {
            const ZkSeg sg = segs[seg0 + tid];
            s_meta[tid][0] = sg.pos; s_meta[tid][1] = sg.out; s_meta[tid][2] = sg.b0; s_meta[tid][3] = segn[seg0 + tid];
        }
        __syncthreads();
        const uint32_t j = tid / ZK_FILL_TC, t = tid % ZK_FILL_TC;
        if (j < nm && t < s_meta[j][3]) s_tc[j][t] = tilecnt[zk_seg_tile_region(frame_off, s_meta[j][0], block_base + s_meta[j][2], seg0 + j) + t];
        __syncthreads();
    }
    auto seg_pos_of = [&](uint32_t j) { return j < ZK_FILL_META ? zk_uni(s_meta[j][0]) : zk_uni(segs[seg0 + j].pos); };
    auto seg_out_of = [&](uint32_t j) { return j < ZK_FILL_META ? zk_uni(s_meta[j][1]) : zk_uni(segs[seg0 + j].out); };
    auto seg_b0_of = [&](uint32_t j) { return j < ZK_FILL_META ? zk_uni(s_meta[j][2]) : zk_uni(segs[seg0 + j].b0); };
    auto seg_nt_of = [&](uint32_t j) { return j < ZK_FILL_META ? zk_uni(s_meta[j][3]) : zk_uni(segn[seg0 + j]); };
    for (uint32_t j0 = 0; j0 < ns;) {
        // the turn: segments j0 .. j1 - 1
        const uint32_t tpos = seg_pos_of(j0);
        uint32_t j1 = j0, tout = 0, steps = 0;
        while (j1 < ns && (j1 == j0 || tout + seg_out_of(j1) <= ZK_SEG_BYTES)) { tout += seg_out_of(j1); steps += seg_nt_of(j1); j1++; }
        if (!steps) { final_to = tpos + tout; j0 = j1; continue; }               // (no holes: pass 1 left these segments complete)
        const uint32_t shift = (uint32_t)((uintptr_t)(out + tpos) & 15u), ibase = tpos - shift, total = shift + tout;
        const uint8_t *A = out + tpos - shift;                                   // 16-byte aligned; image byte i <-> A[i]
        uint8_t *Aw = out + tpos - shift;
        __syncthreads();                                                         // (the turn before: its image stored and the stores waited out, s_cnt read)
        if (progress && final_to && tid == 0) zk_publish(progress + f, final_to);
        // 0: whole 16-byte units inside the turn's bytes; the ragged ends byte by byte (bytes outside are not this workgroup's)
        for (uint32_t u = tid; u * 16u < total; u += L) {
            const uint32_t i0 = u * 16u;
            if (i0 >= shift && i0 + 16u <= total) *reinterpret_cast<uint4 *>(img + i0) = *reinterpret_cast<const uint4 *>(A + i0);
            else for (uint32_t k = 0; k < 16u; k++) if (i0 + k >= shift && i0 + k < total) img[i0 + k] = A[i0 + k];
        }
        const uint32_t nt0 = seg_nt_of(j0);
        bool fast = j1 == j0 + 1 && j0 < ZK_FILL_META && nt0 <= (uint32_t)ZK_FILL_NT;       // one long segment: its records in registers
        if (fast) {
            const ZkHole *rec = holes + zk_uni((uint64_t)zk_seg_region(frame_off, tpos, seg0 + j0));
            ZkHole r0[ZK_FILL_NT], r1[ZK_FILL_NT];
            uint32_t base = 0;
#pragma unroll
            for (int t = 0; t < ZK_FILL_NT; t++) {                               // A
                const uint32_t cnt = (uint32_t)t < nt0 ? zk_uni(s_tc[j0][t]) : 0u;
                fast = fast && cnt <= 2u * L;
                r0[t] = tid < cnt ? rec[base + tid] : 0;
                r1[t] = tid + L < cnt ? rec[base + tid + L] : 0;
                base += cnt;
            }
            __syncthreads();                                                     // (the image is loaded)
            if (fast) {
#pragma unroll
                for (int t = 0; t < ZK_FILL_NT; t++) {                           // B
                    zk_fill_from_memory(r0[t], tpos, ibase, flimit, out, img);
                    zk_fill_from_memory(r1[t], tpos, ibase, flimit, out, img);
                }
                ZK_LDS_BARRIER();
#pragma unroll
                for (int t = 0; t < ZK_FILL_NT; t++) {                           // C
                    if ((uint32_t)t < nt0) {
                        zk_fill_from_image(r0[t], tpos, ibase, img);
                        zk_fill_from_image(r1[t], tpos, ibase, img);
                        ZK_LDS_BARRIER();
                    }
                }
            }
        } else __syncthreads();                                                  // (the image is loaded)
        if (!fast) {
            // any other shape (several short segments: a seek; more tiles or records than the registers hold): segment after segment, tile after
            // tile, a tile's first L records requested a tile ahead (the counts are known: read ahead, or staged L at a time)
            for (uint32_t j = j0; j < j1; j++) {
                const uint32_t nt = seg_nt_of(j);
                if (!nt) continue;
                const uint32_t spos = seg_pos_of(j);
                const ZkHole *rec = holes + zk_uni((uint64_t)zk_seg_region(frame_off, spos, seg0 + j));
                const uint32_t *tc = tilecnt + zk_uni((uint64_t)zk_seg_tile_region(frame_off, spos, block_base + seg_b0_of(j), seg0 + j));
                const bool ahead = j < ZK_FILL_META && nt <= ZK_FILL_TC;
                uint32_t base = 0;
                for (uint32_t t0 = 0; t0 < nt; t0 += L) {
                    if (!ahead) { __syncthreads(); s_cnt[tid] = t0 + tid < nt ? tc[t0 + tid] : 0u; __syncthreads(); }
                    const uint32_t nchunk = nt - t0 < (uint32_t)L ? nt - t0 : (uint32_t)L;
                    uint32_t cnt = ahead ? zk_uni(s_tc[j][0]) : zk_uni(s_cnt[0]);
                    ZkHole nxt = tid < cnt ? rec[base + tid] : 0;
                    for (uint32_t t = 0; t < nchunk; t++) {
                        const ZkHole cur = nxt;
                        const uint32_t ncnt = t + 1 < nchunk ? (ahead ? zk_uni(s_tc[j][t + 1]) : zk_uni(s_cnt[t + 1])) : 0u;
                        nxt = tid < ncnt ? rec[base + cnt + tid] : 0;
                        for (uint32_t i0 = 0; i0 < cnt; i0 += L) {               // (all lanes: the helpers vote)
                            const ZkHole r = i0 ? (i0 + tid < cnt ? rec[base + i0 + tid] : 0) : cur;
                            zk_fill_from_memory(r, tpos, ibase, flimit, out, img);
                            zk_fill_from_image(r, tpos, ibase, img);
                        }
                        base += cnt; cnt = ncnt;
                        ZK_LDS_BARRIER();
                    }
                }
            }
        }
        // E
        for (uint32_t u = tid; u * 16u < total; u += L) {
            const uint32_t i0 = u * 16u;
            if (i0 >= shift && i0 + 16u <= total) *reinterpret_cast<uint4 *>(Aw + i0) = *reinterpret_cast<const uint4 *>(img + i0);
            else for (uint32_t k = 0; k < 16u; k++) if (i0 + k >= shift && i0 + k < total) Aw[i0 + k] = img[i0 + k];
        }
        final_to = tpos + tout;
        j0 = j1;
    }
    if (progress) {
        __syncthreads();
        if (tid == 0) zk_publish(progress + f, flimit);
    }
}

// ------------------------------------------------------------------------------------------------ XXH64
// hashes != nullptr -> store the 64-bit hash; infos != nullptr -> verify Content_Checksum.
// rotl by 1 ... 31 as two v_alignbit_b32 (the compiler's form: a 64-bit shift, a 32-bit one and an or -- three instructions of a chain on
// which every instruction is ~8 clocks of a lone wave, the 64-bit shift more)
__device__ __forceinline__ uint64_t zk_rotl64(uint64_t x, int r)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    return ((uint64_t)__builtin_amdgcn_alignbit(hi, lo, 32u - (uint32_t)r) << 32) | __builtin_amdgcn_alignbit(lo, hi, 32u - (uint32_t)r);
}
constexpr uint64_t XP1 = 0x9E3779B185EBCA87ull, XP2 = 0xC2B2AE3D27D4EB4Full, XP3 = 0x165667B19E3779F9ull,
                   XP4 = 0x85EBCA77C2B2AE63ull, XP5 = 0x27D4EB2F165667C5ull;
__device__ __forceinline__ uint64_t zk_xround(uint64_t acc, uint64_t x) { return zk_rotl64(acc + x * XP2, 31) * XP1; }

// One wave per frame.  All 64 lanes stream the frame in 1 KiB chunks (coalesced 16 B loads, one chunk
// in flight ahead) and pre-multiply every word by P2; the products go through LDS to lanes 0-3, which
// run the four serial accumulator chains acc = rotl(acc + x*P2, 31) * P1.
__global__ __launch_bounds__(64) void zk_k_xxh64(const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                                                 ZkFrameInfo *infos, uint64_t *hashes)
{
    __shared__ uint64_t prod[128];
    const uint32_t lane = threadIdx.x, f = blockIdx.x;
    if (infos && !(infos[f].status == ZK_OK && infos[f].checksum_flag)) return;
    const uint8_t *p = data + (d_off[first + f] - d_off[first]);
    const uint64_t len = d_off[first + f + 1] - d_off[first + f];
    const uint64_t nchunks = len >> 10;
    // lanes 4..15 shadow lanes 0..3: a wave with < 16 active lanes runs ~3x slower on gfx950 (tools/ubench/lat3.hip)
    const uint32_t kl = lane & 3;
    uint64_t acc = kl == 0 ? XP1 + XP2 : kl == 1 ? XP2 : kl == 2 ? 0 : 0 - XP1;
    const bool aligned = (((uintptr_t)p) & 15) == 0;
    uint64_t a = 0, b = 0;
    auto ldpair = [&](uint64_t c) {
        const uint8_t *q = p + (c << 10) + lane * 16;
        if (aligned) { uint4 v = *reinterpret_cast<const uint4 *>(q); a = v.x | ((uint64_t)v.y << 32); b = v.z | ((uint64_t)v.w << 32); }
        else { a = zk_ld64(q); b = zk_ld64(q + 8); }
    };
    if (nchunks) ldpair(0);
    for (uint64_t c = 0; c < nchunks; c++) {
        prod[2 * lane] = a * XP2; prod[2 * lane + 1] = b * XP2;
        if (c + 1 < nchunks) ldpair(c + 1);
        __syncthreads();
        if (lane < 16) {
#pragma unroll 8
            for (int r = 0; r < 32; r++) acc = zk_rotl64(acc + prod[4 * r + kl], 31) * XP1;
        }
        __syncthreads();
    }
    // remaining whole stripes (< 32 of them), straight from memory
    const uint64_t nstripes = len >> 5;
    if (lane < 16) for (uint64_t i = nchunks << 5; i < nstripes; i++) acc = zk_xround(acc, zk_ld64(p + (i << 5) + 8 * kl));
    uint64_t v1 = __shfl(acc, 0, 64), v2 = __shfl(acc, 1, 64), v3 = __shfl(acc, 2, 64), v4 = __shfl(acc, 3, 64);
    if (lane != 0) return;
    uint64_t h;
    if (len >= 32) {
        h = zk_rotl64(v1, 1) + zk_rotl64(v2, 7) + zk_rotl64(v3, 12) + zk_rotl64(v4, 18);
        h = (h ^ zk_xround(0, v1)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v2)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v3)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v4)) * XP1 + XP4;
    } else h = XP5;
    h += len;
    const uint8_t *t = p + (nstripes << 5), *end = p + len;
    while (t + 8 <= end) { h ^= zk_xround(0, zk_ld64(t)); h = zk_rotl64(h, 27) * XP1 + XP4; t += 8; }
    if (t + 4 <= end) { h ^= (uint64_t)zk_rd32(t) * XP1; h = zk_rotl64(h, 23) * XP2 + XP3; t += 4; }
    while (t < end) { h ^= (uint64_t)(*t) * XP5; h = zk_rotl64(h, 11) * XP1; t++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    if (hashes) hashes[f] = h;
    if (infos && (uint32_t)h != infos[f].checksum) infos[f].status = ZK_E_CHECKSUM_WRONG;
}

// The same for a large batch: SIXTEEN frames per wave, lane = (frame, accumulator).  A frame's four chains are a latency floor
// (65 536 dependent rounds per 2 MiB, ~110 clocks each) whatever runs them; one wave per frame spends a whole wave's instruction
// slots on 4 (16) lanes -- 1.7 G wave instructions per 4 GiB, a twelfth of what the encoder's matcher issues, and whichever kernel
// runs beside the checksums pays for them (the entropy stage: 6.3 -> 7.6 ms).  Here the same chains cost a sixteenth of that:
// 128 waves for 2048 frames, a lane loading its own 8 bytes of every stripe, 24 stripes per batch, two batches under way.
constexpr int ZK_XXW = 24;
__global__ __launch_bounds__(64) void zk_k_xxh64_wide(const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                                                      ZkFrameInfo *infos, uint64_t *hashes, const uint64_t *skip)
{
    const uint32_t lane = threadIdx.x, f = blockIdx.x * 16 + (lane >> 2), kl = lane & 3;
    bool live = f < count;
    if (live && infos && !(infos[f].status == ZK_OK && infos[f].checksum_flag)) live = false;
    if (live && skip && (skip[f] & ZK_PROG_VERIFIED)) live = false;       // zk_k_xxh64_follow has verified this frame
    const uint32_t fa = live ? f : 0;                       // (a lane without a frame reads frame 0 and keeps nothing)
    const uint8_t *p = data + (d_off[first + fa] - d_off[first]);
    const uint64_t len = d_off[first + fa + 1] - d_off[first + fa];
    const uint64_t nstripes = live ? len >> 5 : 0;
    uint64_t acc = kl == 0 ? XP1 + XP2 : kl == 1 ? XP2 : kl == 2 ? 0 : 0 - XP1;
    // the stripes every live frame of the wave has, in whole double batches: no predicates
    uint64_t least = live ? nstripes : ~0ull;
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { const uint64_t o = __shfl_xor(least, m, 64); least = o < least ? o : least; }
    if (least == ~0ull) return;                             // no frame to hash in this wave
    const uint64_t common = least - least % (2 * ZK_XXW);
    const uint8_t *q = p + 8 * kl;
    uint64_t wa[ZK_XXW], wb[ZK_XXW];
    if (common) {
#pragma unroll
        for (int u = 0; u < ZK_XXW; u++) wa[u] = zk_ld64(q + 32 * (uint64_t)u);
        for (uint64_t s = 0; s < common; s += 2 * ZK_XXW) {
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) wb[u] = zk_ld64(q + 32 * (s + ZK_XXW + u));
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) acc = zk_xround(acc, wa[u]);
            const uint64_t nx = s + 2 * ZK_XXW < common ? s + 2 * ZK_XXW : 0;     // (the last round reads the frame's first batch again)
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) wa[u] = zk_ld64(q + 32 * (nx + u));
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) acc = zk_xround(acc, wb[u]);
        }
    }
    // what is left of the longer frames, stripe by stripe
    for (uint64_t i = common; i < nstripes; i++) acc = zk_xround(acc, zk_ld64(q + (i << 5)));
    const uint32_t base = lane & ~3u;
    const uint64_t v1 = __shfl(acc, base, 64), v2 = __shfl(acc, base + 1, 64), v3 = __shfl(acc, base + 2, 64), v4 = __shfl(acc, base + 3, 64);
    if (kl != 0 || !live) return;
    uint64_t h;
    if (len >= 32) {
        h = zk_rotl64(v1, 1) + zk_rotl64(v2, 7) + zk_rotl64(v3, 12) + zk_rotl64(v4, 18);
        h = (h ^ zk_xround(0, v1)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v2)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v3)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v4)) * XP1 + XP4;
    } else h = XP5;
    h += len;
    const uint8_t *t = p + (nstripes << 5), *end = p + len;
    while (t + 8 <= end) { h ^= zk_xround(0, zk_ld64(t)); h = zk_rotl64(h, 27) * XP1 + XP4; t += 8; }
    if (t + 4 <= end) { h ^= (uint64_t)zk_rd32(t) * XP1; h = zk_rotl64(h, 23) * XP2 + XP3; t += 4; }
    while (t < end) { h ^= (uint64_t)(*t) * XP5; h = zk_rotl64(h, 11) * XP1; t++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    if (hashes) hashes[f] = h;
    if (infos && (uint32_t)h != infos[f].checksum) infos[f].status = ZK_E_CHECKSUM_WRONG;
}

// A WAVE THAT ONLY RUNS THE CHAINS, FED BY ANOTHER (r6).  zk_k_xxh64_wide's wave does everything for its 64 chains: per round the load,
// the product x * P2 (four instructions) and the chain's own six -- of one wave, all 64 lanes at work: an instruction is four passes of the
// SIMD, a 32-bit multiply sixteen clocks: ~100 clocks per round, 2.7 ms per 2 MiB frame, during which the device is all but idle (128
// waves) and the decode waits (the step's stages run one after the other).  Here wave 0 runs the chains and nothing else -- a product out
// of LDS, mad + two mul_lo + add3, two alignbit -- in SIXTEEN lanes (NF = 4 frames per workgroup): one pass of the SIMD per instruction
// instead of four.  Wave 1 brings the products a chunk of 32 stripes ahead: 16-byte loads (lane l: words 2l and 2l + 1 of the frame's
// next KiB), D chunks of every frame under way, two products per lane, one 16-byte LDS write -- rows of NF x 32 bytes + 32 bytes of
// padding, so that eight lanes' writes (four rows) fall into eight groups of banks.  One barrier per chunk: behind barrier k the chains
// take buffer k & 1 while the producer fills the other one.
//   Measured (2048 x 2 MiB, tools/gpu_calls/r6am.sh ... r6an.sh): wide 2.72 ms (2.52 with the rotations as two alignbit), 64 chains fed by
// two waves 2.96 (the chains' instructions still take four passes), 16 chains fed by one wave with two chunks under way 2.36 (the chains
// wait for memory at the barrier), with six: 1.65 ms.
constexpr int ZK_XF_CHUNK = 32;
template <int NF>                                            // frames per workgroup: 4 (16 chains: one of the SIMD's four passes per instruction)
__global__ __launch_bounds__(64 * (1 + (NF >= 8 ? NF / 8 : 1))) void zk_k_xxh64_fed(const uint8_t *__restrict__ data, const uint64_t *__restrict__ d_off, uint32_t first, uint32_t count,
                                                      ZkFrameInfo *infos, uint64_t *hashes, const uint64_t *skip)
{
    constexpr int PF = NF >= 8 ? 8 : NF;                     // frames per producer wave
    constexpr int ZK_XF_ROW = NF * 32 + 32;
    __shared__ __attribute__((aligned(16))) uint8_t ring[2][ZK_XF_CHUNK * ZK_XF_ROW];
    const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t f = blockIdx.x * NF + (lane >> 2), kl = lane & 3;
    bool live = lane < 4 * NF && f < count;
    if (live && infos && !(infos[f].status == ZK_OK && infos[f].checksum_flag)) live = false;
    if (live && skip && (skip[f] & ZK_PROG_VERIFIED)) live = false;       // zk_k_xxh64_follow has verified this frame
    const uint32_t fa = live ? f : 0;                       // (a lane without a frame reads frame 0 and keeps nothing)
    const uint8_t *p = data + (d_off[first + fa] - d_off[first]);
    const uint64_t len = d_off[first + fa + 1] - d_off[first + fa];
    const uint64_t nstripes = live ? len >> 5 : 0;
    uint64_t least = live ? nstripes : ~0ull;
    bool ragged = (((uintptr_t)p) & 15) != 0;
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { const uint64_t o = __shfl_xor(least, m, 64); least = o < least ? o : least; }
    if (least == ~0ull) return;                             // no frame to hash in this workgroup (every wave finds the same)
    const bool any_ragged = __any(ragged);
    const uint64_t nchunks = least / ZK_XF_CHUNK;            // the chunks every live frame of the group has
    if (wave != 0) {
        // producers: frames PF (wave - 1) ... of the group; the frame's pointer sits in lane 4 j of the consumer's layout
        const uint32_t j0 = PF * (wave - 1);
        const uint8_t *pj[PF];
#pragma unroll
        for (int j = 0; j < PF; j++) {
            const uint64_t a = (uint64_t)(uintptr_t)p;
            pj[j] = (const uint8_t *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(a >> 32), 4 * (int)(j0 + j)) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a, 4 * (int)(j0 + j)));
        }
        // D chunks of every frame under way (a chunk of the chains is ~0.8 us, a load from memory two or three times that; D = 2 left the
        // chains waiting at the barrier: 2.36 ms per 2 MiB frame with four frames per workgroup)
        constexpr int D = NF == 4 ? 6 : 2;
        uint64_t wa[D][PF], wb[D][PF];
        auto load = [&](uint64_t c, uint64_t *xa, uint64_t *xb) {
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const uint8_t *q = pj[j] + (c << 10) + lane * 16;
                if (!any_ragged) { const uint4 v = *reinterpret_cast<const uint4 *>(q); xa[j] = v.x | ((uint64_t)v.y << 32); xb[j] = v.z | ((uint64_t)v.w << 32); }
                else { xa[j] = zk_ld64(q); xb[j] = zk_ld64(q + 8); }
            }
        };
        auto fill = [&](uint32_t buf, const uint64_t *xa, const uint64_t *xb) {
            uint8_t *row = ring[buf] + (lane >> 1) * ZK_XF_ROW + j0 * 32 + (lane & 1) * 16;
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const uint64_t pa = xa[j] * XP2, pb = xb[j] * XP2;
                *reinterpret_cast<uint4 *>(row + j * 32) = make_uint4((uint32_t)pa, (uint32_t)(pa >> 32), (uint32_t)pb, (uint32_t)(pb >> 32));
            }
        };
        // chunk c lives in registers [c % D]; before barrier k the buffer k & 1 holds chunk k, behind it chunk k + 1 is written
#pragma unroll
        for (int d = 0; d < D; d++) if ((uint64_t)d < nchunks) load(d, wa[d], wb[d]);
        if (nchunks) fill(0, wa[0], wb[0]);
        for (uint64_t k0 = 0; k0 < nchunks; k0 += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const uint64_t k = k0 + d;
                if (k < nchunks) {
                    __syncthreads();
                    if (k + D < nchunks) load(k + D, wa[d], wb[d]);              // (chunk k's registers are free: it went into the ring before this barrier)
                    if (k + 1 < nchunks) fill((uint32_t)(k + 1) & 1u, wa[(d + 1) % D], wb[(d + 1) % D]);
                }
            }
        }
        return;
    }
    // the chains
    uint64_t acc = kl == 0 ? XP1 + XP2 : kl == 1 ? XP2 : kl == 2 ? 0 : 0 - XP1;
    for (uint64_t k = 0; k < nchunks; k++) {
        __syncthreads();
        if (lane < 4 * NF) {                                 // (NF = 4: sixteen lanes at work -- an instruction of the chain takes one pass of the SIMD, not four)
            const uint8_t *row = ring[k & 1] + lane * 8;
#pragma unroll 8
            for (int r = 0; r < ZK_XF_CHUNK; r++) acc = zk_rotl64(acc + *reinterpret_cast<const uint64_t *>(row + r * ZK_XF_ROW), 31) * XP1;
        }
    }
    // what is left of the longer frames, stripe by stripe
    const uint8_t *q = p + 8 * kl;
    for (uint64_t i = nchunks * ZK_XF_CHUNK; i < nstripes; i++) acc = zk_xround(acc, zk_ld64(q + (i << 5)));
    const uint32_t base = lane & ~3u;
    const uint64_t v1 = __shfl(acc, base, 64), v2 = __shfl(acc, base + 1, 64), v3 = __shfl(acc, base + 2, 64), v4 = __shfl(acc, base + 3, 64);
    if (kl != 0 || !live) return;
    uint64_t h;
    if (len >= 32) {
        h = zk_rotl64(v1, 1) + zk_rotl64(v2, 7) + zk_rotl64(v3, 12) + zk_rotl64(v4, 18);
        h = (h ^ zk_xround(0, v1)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v2)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v3)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v4)) * XP1 + XP4;
    } else h = XP5;
    h += len;
    const uint8_t *t = p + (nstripes << 5), *end = p + len;
    while (t + 8 <= end) { h ^= zk_xround(0, zk_ld64(t)); h = zk_rotl64(h, 27) * XP1 + XP4; t += 8; }
    if (t + 4 <= end) { h ^= (uint64_t)zk_rd32(t) * XP1; h = zk_rotl64(h, 23) * XP2 + XP3; t += 4; }
    while (t < end) { h ^= (uint64_t)(*t) * XP5; h = zk_rotl64(h, 11) * XP1; t++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    if (hashes) hashes[f] = h;
    if (infos && (uint32_t)h != infos[f].checksum) infos[f].status = ZK_E_CHECKSUM_WRONG;
}

// The same in 64 registers (12 stripes per batch under waves_per_eu(8); zk_k_xxh64_wide takes 78): what is left on a SIMD beside four
// executor waves of 112, so that the checksums of one batch can run while the executor of the next batch in flight holds the CUs --
// with 78 the checksum waves of batch A wait until executor workgroups of batch B retire.  (A template only for this kernel: the
// same body as a template under zk_k_xxh64_wide made the compiler take 122 registers there.)
template <int ZK_XXW>
__device__ __forceinline__ void zk_xxh64_lean_body(const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                                                    ZkFrameInfo *infos, uint64_t *hashes)
{
    const uint32_t lane = threadIdx.x, f = blockIdx.x * 16 + (lane >> 2), kl = lane & 3;
    bool live = f < count;
    if (live && infos && !(infos[f].status == ZK_OK && infos[f].checksum_flag)) live = false;
    const uint32_t fa = live ? f : 0;                       // (a lane without a frame reads frame 0 and keeps nothing)
    const uint8_t *p = data + (d_off[first + fa] - d_off[first]);
    const uint64_t len = d_off[first + fa + 1] - d_off[first + fa];
    const uint64_t nstripes = live ? len >> 5 : 0;
    uint64_t acc = kl == 0 ? XP1 + XP2 : kl == 1 ? XP2 : kl == 2 ? 0 : 0 - XP1;
    // the stripes every live frame of the wave has, in whole double batches: no predicates
    uint64_t least = live ? nstripes : ~0ull;
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) { const uint64_t o = __shfl_xor(least, m, 64); least = o < least ? o : least; }
    if (least == ~0ull) return;                             // no frame to hash in this wave
    const uint64_t common = least - least % (2 * ZK_XXW);
    const uint8_t *q = p + 8 * kl;
    uint64_t wa[ZK_XXW], wb[ZK_XXW];
    if (common) {
#pragma unroll
        for (int u = 0; u < ZK_XXW; u++) wa[u] = zk_ld64(q + 32 * (uint64_t)u);
        for (uint64_t s = 0; s < common; s += 2 * ZK_XXW) {
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) wb[u] = zk_ld64(q + 32 * (s + ZK_XXW + u));
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) acc = zk_xround(acc, wa[u]);
            const uint64_t nx = s + 2 * ZK_XXW < common ? s + 2 * ZK_XXW : 0;     // (the last round reads the frame's first batch again)
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) wa[u] = zk_ld64(q + 32 * (nx + u));
#pragma unroll
            for (int u = 0; u < ZK_XXW; u++) acc = zk_xround(acc, wb[u]);
        }
    }
    // what is left of the longer frames, stripe by stripe
    for (uint64_t i = common; i < nstripes; i++) acc = zk_xround(acc, zk_ld64(q + (i << 5)));
    const uint32_t base = lane & ~3u;
    const uint64_t v1 = __shfl(acc, base, 64), v2 = __shfl(acc, base + 1, 64), v3 = __shfl(acc, base + 2, 64), v4 = __shfl(acc, base + 3, 64);
    if (kl != 0 || !live) return;
    uint64_t h;
    if (len >= 32) {
        h = zk_rotl64(v1, 1) + zk_rotl64(v2, 7) + zk_rotl64(v3, 12) + zk_rotl64(v4, 18);
        h = (h ^ zk_xround(0, v1)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v2)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v3)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v4)) * XP1 + XP4;
    } else h = XP5;
    h += len;
    const uint8_t *t = p + (nstripes << 5), *end = p + len;
    while (t + 8 <= end) { h ^= zk_xround(0, zk_ld64(t)); h = zk_rotl64(h, 27) * XP1 + XP4; t += 8; }
    if (t + 4 <= end) { h ^= (uint64_t)zk_rd32(t) * XP1; h = zk_rotl64(h, 23) * XP2 + XP3; t += 4; }
    while (t < end) { h ^= (uint64_t)(*t) * XP5; h = zk_rotl64(h, 11) * XP1; t++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    if (hashes) hashes[f] = h;
    if (infos && (uint32_t)h != infos[f].checksum) infos[f].status = ZK_E_CHECKSUM_WRONG;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void zk_k_xxh64_lean(const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                                                                                                  ZkFrameInfo *infos, uint64_t *hashes)
{
    zk_xxh64_lean_body<12>(data, d_off, first, count, infos, hashes);
}

// The checksums of a large batch WHILE the executor writes it (zk_engine.hip launches this kernel on the context's second queue
// beside zk_k_exec).  A frame's four chains are 65 536 dependent rounds per 2 MiB whoever runs them (2.75 ms behind an 8 ms
// executor: a fifth of the step); but the first byte of a frame is final milliseconds before its last one.  The layout of
// zk_k_xxh64_lean -- sixteen frames per wave, 64 registers: a wave that fits beside four executor waves of 96 on a SIMD -- with the
// sixteen frames advancing together behind the slowest of their executors' progress words (zk_publish above: relaxed agent-scope
// polls, one invalidation of this CU's L1 per step forward, plain loads behind it).  Wave g takes the frames g % 8 + 8 k of its
// group of 128: the frames whose executors the dispatcher puts on XCD g % 8, where it puts this wave too -- if it does (see above).
//   Nothing depends on this kernel: a frame it verifies is marked ZK_PROG_VERIFIED in its progress word; every other frame --
// its executor ran on another XCD, never came (the two kernels were serialised: a profiler, one hardware queue), the wave gave up
// after ZK_FOLLOW_PATIENCE without any news, the hash differs from the frame's Content_Checksum for whatever reason -- is left to
// the pass behind the executor (zk_k_xxh64_wide with `skip`), which alone reports ZK_E_CHECKSUM_WRONG.  All loops are left by the
// whole wave on a wave-uniform verdict (the trap of the per-lane exit from a polling loop).
#ifndef ZK_FOLLOW_W
#define ZK_FOLLOW_W 7
#endif
constexpr uint64_t ZK_FOLLOW_PATIENCE = 5000000;             // wall_clock64 ticks (100 MHz): 50 ms without any progress
template <int W>
__device__ __forceinline__ void zk_xxh64_follow_body(const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                                                      const ZkFrameInfo *infos, uint64_t *progress)
{
    const uint32_t lane = threadIdx.x, g = blockIdx.x, kl = lane & 3;
    const uint32_t here = zk_xcc_id() + 1;
    uint64_t t_news = wall_clock64();
    // Which frames run on this XCD?  The dispatcher deals a kernel's workgroups round the XCDs, b -> (b + r) % 8, with a start r that
    // differs from launch to launch (tools/ubench/xcc_map.hip): the first executor whose word appears tells r, and of the 128 frames
    // of this wave's group the class c = (this XCD - r) % 8 is the one to follow.  (A habit, not a promise: every frame's own word is
    // checked again below.)
    uint32_t rot;
    for (;;) {
        uint32_t cand = 0xFFFFFFFFu;
        if (lane < count) {
            const uint64_t w = __hip_atomic_load(progress + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w != 0) cand = (((uint32_t)(w >> 32) & 15u) - 1u - lane) & 7u;
        }
        const unsigned long long m = __ballot(cand != 0xFFFFFFFFu);
        if (m) { rot = (uint32_t)__builtin_amdgcn_readlane(cand, __builtin_ctzll(m)); break; }
        const uint32_t waited = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wall_clock64() - t_news > ZK_FOLLOW_PATIENCE));
        if (waited) return;
        __builtin_amdgcn_s_sleep(20);
    }
    const uint32_t f = (g >> 3) * 128 + ((here - 1u - rot) & 7u) + 8 * (lane >> 2);
    bool live = f < count;
    if (live && !(infos[f].status == ZK_OK && infos[f].checksum_flag)) live = false;
    const uint32_t fa = live ? f : 0;                       // (a lane without a frame reads frame 0 and keeps nothing)
    const uint8_t *p = data + (d_off[first + fa] - d_off[first]);
    const uint64_t len = d_off[first + fa + 1] - d_off[first + fa];
    const uint32_t nstripes = live ? (uint32_t)(len >> 5) : 0;
    uint64_t acc = kl == 0 ? XP1 + XP2 : kl == 1 ? XP2 : kl == 2 ? 0 : 0 - XP1;
    auto wave_min = [](uint32_t v) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const uint32_t o = __shfl_xor(v, m, 64); v = o < v ? o : v; }
        return (uint32_t)__builtin_amdgcn_readfirstlane(v);
    };
    // the frame's word: bytes complete (0 before its executor has started), or 0xFFFFFFFF for "not mine (any more)": failed, or
    // written through another XCD's L2
    auto news = [&]() {
        uint32_t have = 0xFFFFFFFFu;
        if (live) {
            const uint64_t w = __hip_atomic_load(progress + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w & ZK_PROG_ABORT) || (w != 0 && ((uint32_t)(w >> 32) & 15u) != here)) live = false;
            else have = (uint32_t)w;
        }
        return have;
    };
    uint32_t s = 0, common = 0;
    {
        const uint32_t least = wave_min(live ? nstripes : 0xFFFFFFFFu);
        if (least == 0xFFFFFFFFu) return;                   // no frame to hash in this wave
        common = least - least % (2 * W);                   // the stripes every frame has, in whole double batches (a frame that drops out later changes nothing)
    }
    const uint8_t *q = p + 8 * kl;
    uint64_t wa[W], wb[W];
    // A: the sixteen frames together, as far as the slowest executor has come
    while (s < common) {
        const uint32_t have = news();
        if (__ballot(live) == 0) return;
        uint32_t upto = wave_min(have == 0xFFFFFFFFu ? have : have >> 5);
        upto = upto < common ? upto - upto % (2 * W) : common;
        if (upto <= s) {
            const uint32_t waited = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wall_clock64() - t_news > ZK_FOLLOW_PATIENCE));
            if (waited) return;                             // (the whole wave: `waited` is scalar)
            __builtin_amdgcn_s_sleep(100);
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (s_waitcnt vmcnt(0); buffer_inv sc1: this CU's L1 may hold the buffer's bytes of an earlier decode)
        t_news = wall_clock64();
#pragma unroll
        for (int u = 0; u < W; u++) wa[u] = zk_ld64(q + 32 * (uint64_t)(s + u));
        for (uint32_t x = s; x < upto; x += 2 * W) {
#pragma unroll
            for (int u = 0; u < W; u++) wb[u] = zk_ld64(q + 32 * (uint64_t)(x + W + u));
#pragma unroll
            for (int u = 0; u < W; u++) acc = zk_xround(acc, wa[u]);
            const uint32_t nx = x + 2 * W < upto ? x + 2 * W : s;       // (the last round reads the step's first batch again)
#pragma unroll
            for (int u = 0; u < W; u++) wa[u] = zk_ld64(q + 32 * (uint64_t)(nx + u));
#pragma unroll
            for (int u = 0; u < W; u++) acc = zk_xround(acc, wb[u]);
        }
        s = upto;
    }
    // B: what is left of every frame once ALL of it is there
    for (;;) {
        const uint32_t have = news();
        const bool pending = live && have < (uint32_t)len;
        if (__ballot(pending) == 0) break;
        const uint32_t waited = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wall_clock64() - t_news > ZK_FOLLOW_PATIENCE));
        if (waited) return;
        __builtin_amdgcn_s_sleep(100);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t i = common; i < nstripes; i++) acc = zk_xround(acc, zk_ld64(q + ((uint64_t)i << 5)));
    const uint32_t base = lane & ~3u;
    const uint64_t v1 = __shfl(acc, base, 64), v2 = __shfl(acc, base + 1, 64), v3 = __shfl(acc, base + 2, 64), v4 = __shfl(acc, base + 3, 64);
    if (kl != 0 || !live) return;
    uint64_t h;
    if (len >= 32) {
        h = zk_rotl64(v1, 1) + zk_rotl64(v2, 7) + zk_rotl64(v3, 12) + zk_rotl64(v4, 18);
        h = (h ^ zk_xround(0, v1)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v2)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v3)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v4)) * XP1 + XP4;
    } else h = XP5;
    h += len;
    const uint8_t *t = p + ((uint64_t)nstripes << 5), *end = p + len;
    while (t + 8 <= end) { h ^= zk_xround(0, zk_ld64(t)); h = zk_rotl64(h, 27) * XP1 + XP4; t += 8; }
    if (t + 4 <= end) { h ^= (uint64_t)zk_rd32(t) * XP1; h = zk_rotl64(h, 23) * XP2 + XP3; t += 4; }
    while (t < end) { h ^= (uint64_t)(*t) * XP5; h = zk_rotl64(h, 11) * XP1; t++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    progress[f] |= (uint32_t)h == infos[f].checksum ? ZK_PROG_VERIFIED : ZK_PROG_MISMATCH;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void zk_k_xxh64_follow(const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                                                                                                    const ZkFrameInfo *infos, uint64_t *progress)
{
    zk_xxh64_follow_body<ZK_FOLLOW_W>(data, d_off, first, count, infos, progress);
}

// The same for a HANDFUL of frames (r6): a wave per frame -- zk_k_xxh64's layout, the fastest a frame's four chains run on this device
// (1.7 ms per 2 MiB; sixteen frames to a wave: 2.25 ms) -- behind the frame's progress word.  Eight waves per group of eight frames;
// wave g takes, of its group, the frame whose executor shares ITS XCD (the word says where it runs; two waves of a group on one XCD hash
// the same frame twice and leave another to the pass behind the executor, which takes every frame that is not marked verified).
__global__ __launch_bounds__(64) void zk_k_xxh64_follow1(const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                                                         const ZkFrameInfo *infos, uint64_t *progress)
{
    __shared__ uint64_t prod[128];
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t here = zk_xcc_id() + 1;
    uint64_t t_news = wall_clock64();
    const uint32_t g0 = g & ~7u;
    uint32_t f = 0xFFFFFFFFu;
    for (;;) {                                               // whose executor runs here?  Once every word of the group has spoken:
        bool mine = false, pending = false;
        if (lane < 8 && g0 + lane < count) {
            const uint64_t w = __hip_atomic_load(progress + g0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w == 0) pending = true;
            else mine = !(w & ZK_PROG_ABORT) && ((uint32_t)(w >> 32) & 15u) == here;
        }
        const unsigned long long mm = __ballot(mine), pm = __ballot(pending);
        if (!pm) {
            if (!mm) return;                                 // none of the group's frames runs on this XCD
            // the frames that run here, in order; this wave takes number (its place in the group) mod (how many): one each when the
            // dispatcher deals waves and workgroups round the XCDs alike
            unsigned long long m2 = mm;
            for (uint32_t k = (g & 7u) % (uint32_t)__popcll(mm); k; k--) m2 &= m2 - 1;
            f = g0 + (uint32_t)__builtin_ctzll(m2);
            break;
        }
        const uint32_t waited = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wall_clock64() - t_news > ZK_FOLLOW_PATIENCE));
        if (waited) return;
        __builtin_amdgcn_s_sleep(20);
    }
    if (!(infos[f].status == ZK_OK && infos[f].checksum_flag)) return;
    const uint8_t *p = data + (d_off[first + f] - d_off[first]);
    const uint64_t len = d_off[first + f + 1] - d_off[first + f];
    const uint64_t nchunks = len >> 10;
    const uint32_t kl = lane & 3;
    uint64_t acc = kl == 0 ? XP1 + XP2 : kl == 1 ? XP2 : kl == 2 ? 0 : 0 - XP1;
    const bool aligned = (((uintptr_t)p) & 15) == 0;
    uint64_t a = 0, b = 0;
    auto ldpair = [&](uint64_t c) {
        const uint8_t *q = p + (c << 10) + lane * 16;
        if (aligned) { uint4 v = *reinterpret_cast<const uint4 *>(q); a = v.x | ((uint64_t)v.y << 32); b = v.z | ((uint64_t)v.w << 32); }
        else { a = zk_ld64(q); b = zk_ld64(q + 8); }
    };
    uint64_t have = 0;                                       // bytes of the frame known to be final
    // wait until `need` bytes are final; false: the frame is not (any more) this wave's -- it failed, ran elsewhere, or nothing was heard for too long
    auto wait_for = [&](uint64_t need) -> bool {
        while (have < need) {
            const uint64_t w = zk_uni((uint64_t)__hip_atomic_load(progress + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));     // one verdict per pass for the whole wave
            if ((w & ZK_PROG_ABORT) || ((uint32_t)(w >> 32) & 15u) != here) return false;
            const uint64_t got = (uint32_t)w;
            if (got > have) { have = got; t_news = wall_clock64(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); continue; }     // (this CU's L1 may hold the buffer's bytes of an earlier decode)
            if (wall_clock64() - t_news > ZK_FOLLOW_PATIENCE) return false;
            __builtin_amdgcn_s_sleep(50);
        }
        return true;
    };
    // (the word is wave-uniform, so is every verdict: the loops below are left by the whole wave)
    if (nchunks) { if (!wait_for(1024)) return; ldpair(0); }
    for (uint64_t c = 0; c < nchunks; c++) {
        prod[2 * lane] = a * XP2; prod[2 * lane + 1] = b * XP2;
        if (c + 1 < nchunks) { if (!wait_for((c + 2) << 10)) return; ldpair(c + 1); }
        __syncthreads();
        if (lane < 16) {
#pragma unroll 8
            for (int r = 0; r < 32; r++) acc = zk_rotl64(acc + prod[4 * r + kl], 31) * XP1;
        }
        __syncthreads();
    }
    if (!wait_for(len)) return;
    const uint64_t nstripes = len >> 5;
    if (lane < 16) for (uint64_t i = nchunks << 5; i < nstripes; i++) acc = zk_xround(acc, zk_ld64(p + (i << 5) + 8 * kl));
    uint64_t v1 = __shfl(acc, 0, 64), v2 = __shfl(acc, 1, 64), v3 = __shfl(acc, 2, 64), v4 = __shfl(acc, 3, 64);
    if (lane != 0) return;
    uint64_t h;
    if (len >= 32) {
        h = zk_rotl64(v1, 1) + zk_rotl64(v2, 7) + zk_rotl64(v3, 12) + zk_rotl64(v4, 18);
        h = (h ^ zk_xround(0, v1)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v2)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v3)) * XP1 + XP4;
        h = (h ^ zk_xround(0, v4)) * XP1 + XP4;
    } else h = XP5;
    h += len;
    const uint8_t *t = p + (nstripes << 5), *end = p + len;
    while (t + 8 <= end) { h ^= zk_xround(0, zk_ld64(t)); h = zk_rotl64(h, 27) * XP1 + XP4; t += 8; }
    if (t + 4 <= end) { h ^= (uint64_t)zk_rd32(t) * XP1; h = zk_rotl64(h, 23) * XP2 + XP3; t += 4; }
    while (t < end) { h ^= (uint64_t)(*t) * XP5; h = zk_rotl64(h, 11) * XP1; t++; }
    h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
    atomicOr((unsigned long long *)(progress + f), (uint32_t)h == infos[f].checksum ? ZK_PROG_VERIFIED : ZK_PROG_MISMATCH);
}

// per-frame status words + first failing frame ((frame << 32) | code, min over frames); with `followed`: the number of frames whose
// checksums zk_k_xxh64_follow verified
// zk_frame_content_sizes: a frame's decompressed size = the regenerated sizes of its blocks, known once the sequence walks have run
// (a compressed block's size is its literals + its match lengths: zk_k_fse_* leave it in ZkBlock::out_size).  One lane per frame.
__global__ __launch_bounds__(256) void zk_k_frame_sizes(const ZkFrameInfo *infos, const ZkFrameBase *bases, const ZkBlock *blocks, uint32_t count, uint64_t *sizes, int32_t *status_out)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= count) return;
    uint32_t st = infos[f].status;
    uint64_t sum = 0;
    if (st == ZK_OK) {
        const ZkBlock *b = blocks + bases[f].block_base;
        for (uint32_t i = 0; i < infos[f].n_blocks; i++) {
            if (b[i].status != ZK_OK && st == ZK_OK) st = b[i].status;
            if (b[i].out_size > (infos[f].window < ZK_BLOCK_MAX ? infos[f].window : ZK_BLOCK_MAX) && st == ZK_OK) st = ZK_E_CORRUPTION;    // as the executor
            sum += b[i].out_size;
        }
        if (st == ZK_OK && sum > ZK_MAX_FRAME) st = ZK_E_FRAMEPARAM_UNSUPPORTED;
        // a header that carries Frame_Content_Size is held to it: what the blocks regenerate IS the frame's size (libzstd: "corrupted block
        // detected" when the two differ at the frame's end) -- the answer never comes from the header alone
        if (st == ZK_OK && infos[f].fcs != ZK_SIZE_UNKNOWN && infos[f].fcs != sum) st = ZK_E_CORRUPTION;
    }
    sizes[f] = st == ZK_OK ? sum : 0;
    status_out[f] = -(int32_t)st;
}

__global__ __launch_bounds__(256) void zk_k_status(const ZkFrameInfo *infos, uint32_t count, int32_t *status_out, unsigned long long *first_err,
                                                   const uint64_t *progress, unsigned long long *followed)
{
    uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= count) return;
    uint32_t st = infos[f].status;
    if (status_out) status_out[f] = (int32_t)st;
    if (st != ZK_OK) atomicMin(first_err, ((unsigned long long)f << 32) | st);
    if (progress) {
        const uint64_t w = progress[f];
        const unsigned long long m = __ballot((w & ZK_PROG_VERIFIED) != 0), m2 = __ballot((w & ZK_PROG_MISMATCH) != 0);
        if ((threadIdx.x & 63) == (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x & 63) && (m | m2))
            atomicAdd(followed, (unsigned long long)__popcll(m) + ((unsigned long long)__popcll(m2) << 32));
    }
}


// ------------------------------------------------------------------------------------------------ small batches (a seek)
// A handful of frames is all latency: launches, read-backs and copy commands cost more than the decoding.  The small path
// therefore runs without a host round trip and without copy commands:
//   zk_k_small_walk     one workgroup: pulls the staged compressed bytes + offsets out of PINNED HOST memory into HBM scratch
//                       (the upload), walks the frames (count, scan, fill in one kernel) and leaves the totals in HBM;
//   zk_k_small_entropy  Huffman groups and sequence groups (quads) in ONE launch, the group count read from HBM;
//   zk_k_exec           as always; zk_k_xxh64 only for frames that are verified;
//   zk_k_small_publish  writes the frames' bytes and status words into PINNED HOST memory (the download) and, last of all,
//                       a completion word the host spins on.
// Scratch is sized from bounds the host knows (sequences <= d / 3, literals <= d); the block list has a cap and a batch
// that exceeds it reports ZK_SMALL_OVERFLOW, upon which the host takes the general path.
constexpr uint32_t ZK_SMALL_OVERFLOW = 0xFFFFFFFFu;
constexpr int ZK_SMALL_FSE_BLOCKS = 8;
__global__ __launch_bounds__(256) void zk_k_small_walk(const uint8_t *h_comp, uint64_t comp_bytes, const uint64_t *h_offs, uint32_t count,
                                                       uint64_t dst_cap, uint32_t block_cap, uint64_t seq_cap, uint8_t *d_comp, uint64_t *d_offs,
                                                       ZkFrameInfo *infos, ZkFrameBase *bases, ZkBlock *blocks, uint64_t *words)
{
    __shared__ uint32_t s_tot[4];
    // (r4) the bytes also stay in LDS when they fit (a seek: one or two frames of a few tens of KiB): the two header walks below are
    // chains of dependent byte reads -- ~4 per block -- and cost an L2 round trip each out of HBM scratch, an LDS one out of here
    __shared__ uint4 s_stage[3072];                          // 48 KiB
    const uint32_t tid = threadIdx.x;
    // upload: 16 bytes per lane and step, straight over PCIe (h_comp is 16-byte aligned, padded to a multiple of 16)
    const uint64_t n16 = (comp_bytes + 15) >> 4;
    const bool staged = n16 <= 3072;
    // (r5) eight loads per lane in flight before the first store: the loop used to be load -> wait -> store, one PCIe round trip (2-3 us) per
    // 4 KiB of compressed bytes -- seven in a row for a 64 KiB frame, a third of this kernel's 50 us
    for (uint64_t base = 0; base < n16; base += 8 * 256) {
        const uint4 *h16 = reinterpret_cast<const uint4 *>(h_comp);
        uint4 *d16 = reinterpret_cast<uint4 *>(d_comp);
        const uint64_t i0 = base + tid;
        // (eight named values, not an array: indexed it went to scratch memory)
#define ZK_UP_LD(k) const uint4 v##k = h16[i0 + (k) * 256 < n16 ? i0 + (k) * 256 : 0];
#define ZK_UP_ST(k) if (i0 + (k) * 256 < n16) { d16[i0 + (k) * 256] = v##k; if (staged) s_stage[i0 + (k) * 256] = v##k; }
        ZK_UP_LD(0) ZK_UP_LD(1) ZK_UP_LD(2) ZK_UP_LD(3) ZK_UP_LD(4) ZK_UP_LD(5) ZK_UP_LD(6) ZK_UP_LD(7)
        ZK_UP_ST(0) ZK_UP_ST(1) ZK_UP_ST(2) ZK_UP_ST(3) ZK_UP_ST(4) ZK_UP_ST(5) ZK_UP_ST(6) ZK_UP_ST(7)
#undef ZK_UP_LD
#undef ZK_UP_ST
    }
    const uint8_t *wcomp = staged ? reinterpret_cast<const uint8_t *>(s_stage) : d_comp;
    if (tid < 2) reinterpret_cast<uint64_t *>(d_comp + (n16 << 4))[tid] = 0;                 // readable padding behind the last frame
    for (uint32_t i = tid; i < 2 * (count + 1); i += 256) d_offs[i] = h_offs[i];
    __syncthreads();
    const uint64_t *c_off = d_offs, *d_off = d_offs + count + 1;
    ZkFrameInfo fi;
    fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; fi.status = ZK_OK; fi.checksum_flag = 0; fi.checksum = 0; fi.window = 0; fi.n_own_tables = 0; fi.fcs = ZK_SIZE_UNKNOWN;
    uint64_t cb = 0, ce = 0, dsz = 0;
    if (tid < count) {
        cb = c_off[tid]; ce = c_off[tid + 1]; dsz = d_off[tid + 1] - d_off[tid];
        if (ce < cb || ce > comp_bytes || d_off[tid + 1] < d_off[tid]) fi.status = ZK_E_SRC_SIZE_WRONG;
        else {
            zk_walk_frame(wcomp, cb, ce, dsz, tid, nullptr, nullptr, fi);
            if (fi.status == ZK_OK && (d_off[tid] > dst_cap || dsz > dst_cap - d_off[tid])) fi.status = ZK_E_DST_TOO_SMALL;
            if (dsz > ZK_MAX_FRAME && fi.status == ZK_OK) fi.status = ZK_E_FRAMEPARAM_UNSUPPORTED;
        }
        if (fi.status != ZK_OK) { fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; fi.n_own_tables = 0; }
    }
    if (tid < 64) {                                          // count <= 64: one wave scans
        uint32_t v[4] = {fi.n_blocks, fi.n_seq, fi.lit_bytes, fi.n_own_tables}, inc[4];
        for (int k = 0; k < 4; k++) {
            uint32_t x = v[k];
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if ((int)tid >= d) x += y; }
            inc[k] = x;
            if (tid == 63) s_tot[k] = x;
        }
        if (tid < count) {
            ZkFrameBase fb;
            fb.block_base = inc[0] - v[0]; fb.seq_base = inc[1] - v[1]; fb.lit_base = inc[2] - v[2];
            bases[tid] = fb;
        }
    }
    __syncthreads();
    // the scratch was sized from bounds the host knows; a crafted frame can declare more sequences than its output could hold
    // (RLE tables, 0-bit codes: ~98 k sequences in a few bytes) -- more record slots than the host reserved is an overflow too, and
    // the general path, which sizes the records from the real totals, gives the verdict
    const bool overflow = s_tot[0] > block_cap || s_tot[1] > seq_cap;
    if (tid < count) {
        if (overflow) { fi.status = ZK_E_GENERIC; fi.n_blocks = 0; }
        infos[tid] = fi;
        if (!overflow && fi.status == ZK_OK) { ZkFrameInfo f2; zk_walk_frame(wcomp, cb, ce, dsz, tid, &bases[tid], blocks, f2); }
    }
    if (tid == 0) {
        words[0] = overflow ? 0 : s_tot[0]; words[1] = overflow ? 0 : s_tot[1]; words[2] = overflow ? 0 : s_tot[2];
        words[3] = ~0ull; words[4] = s_tot[3]; words[5] = overflow ? 1 : 0; words[6] = 0;
    }
}

// Huffman groups (role 0) and sequence groups (role 1) of a small batch in one launch; the number of blocks comes from HBM.
__global__ __launch_bounds__(192) void zk_k_small_entropy(const uint8_t *comp, ZkBlock *blocks, const uint64_t *words, uint8_t *lit, ZkSeqP *seqs)
{
    const uint32_t nblocks = (uint32_t)words[0];
    const uint32_t role = blockIdx.x & 1, stride = gridDim.x >> 1;
    // (sequence groups: 8 blocks, 8-byte cells -- a seek's chain is ~63 instead of ~80 instructions per sequence, and 10 KiB of
    //  tables per block do not matter where one or two blocks are all there is)
    const uint32_t per = role == 0 ? 16u : (uint32_t)ZK_SMALL_FSE_BLOCKS;
    for (uint32_t g = blockIdx.x >> 1; g * per < nblocks; g += stride) {
        if (role == 0) zk_huf_group(g, comp, blocks, nblocks, lit);
        else zk_fse_quad_group<ZkCells64, ZK_SMALL_FSE_BLOCKS, 1, true>(g, comp, blocks, nblocks, seqs, 1u);
        __syncthreads();                                     // the group's LDS state is re-initialised by the next one
    }
}

// the two roles as kernels of their own (diagnosis: ZK_SMALL_SPLIT=1 shows them separately in a kernel trace)
__global__ __launch_bounds__(128) void zk_k_small_huf(const uint8_t *comp, ZkBlock *blocks, const uint64_t *words, uint8_t *lit)
{
    const uint32_t nblocks = (uint32_t)words[0];
    for (uint32_t g = blockIdx.x; g * 16 < nblocks; g += gridDim.x) { zk_huf_group(g, comp, blocks, nblocks, lit); __syncthreads(); }
}
__global__ __launch_bounds__(192) void zk_k_small_fse(const uint8_t *comp, ZkBlock *blocks, const uint64_t *words, ZkSeqP *seqs)
{
    const uint32_t nblocks = (uint32_t)words[0];
    for (uint32_t g = blockIdx.x; g * ZK_SMALL_FSE_BLOCKS < nblocks; g += gridDim.x) { zk_fse_quad_group<ZkCells64, ZK_SMALL_FSE_BLOCKS, 1, true>(g, comp, blocks, nblocks, seqs, 1u); __syncthreads(); }
}

// One workgroup per frame: the download.  h_out may be null (the caller wants the bytes in HBM only).
__global__ __launch_bounds__(256) void zk_k_small_publish(const ZkFrameInfo *infos, const uint64_t *d_offs, uint32_t count, const uint8_t *dst,
                                                          uint8_t *h_out, int32_t *d_status, int32_t *h_status, uint64_t *words,
                                                          volatile uint32_t *h_flag, uint32_t gen)
{
    const uint32_t f = blockIdx.x, tid = threadIdx.x;
    const uint64_t *d_off = d_offs + count + 1;
    const uint32_t st = words[5] ? ZK_SMALL_OVERFLOW : infos[f].status;
    if (h_out && (st == ZK_OK || st == ZK_E_CHECKSUM_WRONG)) {      // a frame that fails its checksum is complete: a reader that
        const uint64_t lo = d_off[f], hi = d_off[f + 1];          // offset_limit cut short ignores that verdict (decode.rs:425-427)
        const uint8_t *s = dst + lo;
        uint8_t *o = h_out + lo;
        // 16-byte pieces where both sides allow it, bytes at the ragged ends
        const uint64_t head = (0 - (uintptr_t)o) & 15, n = hi - lo;
        if ((((uintptr_t)s + head) & 15) == 0 && n >= 64) {
            for (uint64_t i = tid; i < head; i += 256) o[i] = s[i];
            const uint64_t n16 = (n - head) >> 4;
            for (uint64_t i = tid; i < n16; i += 256) reinterpret_cast<uint4 *>(o + head)[i] = reinterpret_cast<const uint4 *>(s + head)[i];
            for (uint64_t i = head + (n16 << 4) + tid; i < n; i += 256) o[i] = s[i];
        } else for (uint64_t i = tid; i < n; i += 256) o[i] = s[i];
    }
    if (tid == 0) {
        if (d_status) d_status[f] = (int32_t)st;
        h_status[f] = (int32_t)st;
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned long long done = atomicAdd((unsigned long long *)&words[6], 1ull) + 1;
        if (done == count) { __threadfence_system(); *h_flag = gen; }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void zk_launch_status(hipStream_t st, const ZkFrameInfo *infos, uint32_t count, int32_t *status_out, uint64_t *first_err, const uint64_t *progress, uint64_t *followed)
{
    hipLaunchKernelGGL(zk_k_status, dim3((count + 255) / 256), dim3(256), 0, st, infos, count, status_out, (unsigned long long *)first_err, progress, (unsigned long long *)followed);
}
void zk_launch_walk(hipStream_t st, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off, const uint64_t *d_off, uint32_t first,
                    uint32_t count, const uint32_t *ids, const uint64_t *out_off, uint64_t dst_cap, const ZkFrameBase *bases, ZkBlock *blocks, ZkFrameInfo *infos)
{
    hipLaunchKernelGGL(zk_k_walk, dim3((count + 63) / 64), dim3(64), 0, st, comp, comp_size, c_off, d_off, first, count, ids, out_off, dst_cap, bases, blocks, infos);
}
void zk_launch_frame_sizes(hipStream_t st, const ZkFrameInfo *infos, const ZkFrameBase *bases, const ZkBlock *blocks, uint32_t count, uint64_t *sizes, int32_t *status_out)
{
    hipLaunchKernelGGL(zk_k_frame_sizes, dim3((count + 255) / 256), dim3(256), 0, st, infos, bases, blocks, count, sizes, status_out);
}
void zk_launch_scan(hipStream_t st, const ZkFrameInfo *infos, uint32_t count, ZkFrameBase *bases, uint64_t *totals, const uint64_t *d_off, uint32_t first, const uint64_t *out_off)
{
    hipLaunchKernelGGL(zk_k_scan, dim3(1), dim3(1024), 0, st, infos, count, bases, totals, d_off, first, out_off);
}
void zk_launch_huf(hipStream_t st, const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, uint8_t *lit)
{
    if (!nblocks) return;
    hipLaunchKernelGGL(zk_k_huf, dim3((nblocks + ZK_HUF_BLOCKS - 1) / ZK_HUF_BLOCKS), dim3(128), 0, st, comp, blocks, nblocks, lit);
}
void zk_launch_fse(hipStream_t st, const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, uint32_t n_own_tables, ZkSeqP *seqs, const ZkKernelChoice &k, uint32_t frames)
{
    if (!nblocks) return;
    const int own_kernel = k.fse_own;
    // reader choice (zk_device.h): with >= 6 workgroups per CU the memory pipeline is the limit (aligned words, each
    // loaded once); below that a lane's instruction count is (one unaligned load per sequence).  Measured crossover
    // on 32 KiB blocks between 1024 and 2048 frames of 2 MiB.
    const uint32_t wgs = (nblocks + ZK_FSEP_LANES - 1) / ZK_FSEP_LANES;
    // a small batch (a seek, a handful of frames) is all chain latency: every block, predefined tables or not, takes a
    // quad of lanes (each block builds its own copy of the tables: microseconds)
    if (own_kernel == 0 && k.fse_shared == 0 && nblocks <= 16u * 256u) {
        // ... and while 8-block workgroups (8-byte cells: ~63 instead of ~80 instructions per sequence of a chain, 10 KiB of tables per
        // block, one workgroup per CU) hold every block in one round, those (what the small path's kernel runs: zk_k_small_entropy)
        if (nblocks <= 8u * 256u) hipLaunchKernelGGL((zk_k_fse_quad<ZkCells64, 8, 1>), dim3((nblocks + 7) / 8), dim3(192), 0, st, comp, blocks, nblocks, seqs, 1u);
        else hipLaunchKernelGGL((zk_k_fse_quad<ZkCellsX16, 16, 1>), dim3((nblocks + 15) / 16), dim3(192), 0, st, comp, blocks, nblocks, seqs, 1u);
        return;
    }
    // blocks that share their tables with their neighbours (predefined, or one set per frame): one lane each
    // frames of fewer than 64 blocks: a workgroup's 64 blocks span several frames = several table sets
    const int shared = k.fse_shared ? k.fse_shared : (frames && (uint64_t)nblocks < 64ull * frames) ? 3 : wgs >= 6 * 256 ? 2 : 1;
    if (shared == 3) hipLaunchKernelGGL(zk_k_fse_sets, dim3(wgs), dim3(ZK_FSEP_LANES), 0, st, comp, blocks, nblocks, seqs);
    else if (shared == 2) hipLaunchKernelGGL(zk_k_fse_predef_fed, dim3(wgs), dim3(2 * ZK_FSEP_LANES), 0, st, comp, blocks, nblocks, seqs);
    else hipLaunchKernelGGL(zk_k_fse_predef<ZkRevU>, dim3(wgs), dim3(ZK_FSEP_LANES), 0, st, comp, blocks, nblocks, seqs);
    // blocks with their own tables (every block is visited, the others return at once): a quad of lanes per block.
    // While everything fits in one round, small workgroups (16 blocks, one walking wave + the toucher: 45 KiB of LDS,
    // three per CU) spread the blocks over the CUs and a block's chain latency is all that counts (measured, 2048
    // blocks of ~10 k sequences: 4.13 ms; 56-block workgroups 4.41; one lane per block 4.57).  Beyond that, 56 blocks
    // per CU on four waves (32768 blocks: 20.6 ms with one lane per block, 13.2 with quads).
    // own_kernel: zk_engine_set_fse_kernel (1 = the lane-per-block kernel, 2 = quads in the large layout).
    if (!n_own_tables) return;              // no block defines a table: everything was shared (predefined)
    if (own_kernel == 1) { hipLaunchKernelGGL((zk_k_fse<ZkCells16, 56, 8>), dim3((nblocks + 55) / 56), dim3(512), 0, st, comp, blocks, nblocks, seqs); return; }
    // (every block the shared-table kernel has not marked done is taken, predefined tables or not.  n_own_tables counts the
    // DEFINING blocks: an archive of this engine's encoder has one per frame, its blocks were all shared, and the pass that
    // finds nothing left must not wait for a whole CU's LDS -- the small layout fits beside whatever else is resident)
    if (own_kernel != 2 && n_own_tables <= 48u * 256u)
        hipLaunchKernelGGL((zk_k_fse_quad<ZkCellsX16, 16, 1>), dim3((nblocks + 15) / 16), dim3(192), 0, st, comp, blocks, nblocks, seqs, 1u);
    else
        hipLaunchKernelGGL((zk_k_fse_quad<ZkCellsX16, 56, 4>), dim3((nblocks + 55) / 56), dim3(576), 0, st, comp, blocks, nblocks, seqs, 1u);
}
void zk_launch_exec(hipStream_t st, const uint8_t *comp, const uint64_t *d_off, uint32_t first, uint32_t count,
                    const uint32_t *ids, const uint64_t *out_off, const ZkBlock *blocks, const ZkFrameBase *bases, ZkFrameInfo *infos, const ZkSeqP *seqs,
                    const uint8_t *lit, uint8_t *dst, const uint8_t *prefix, uint64_t plen, const ZkKernelChoice &k, bool dense, uint64_t *progress)
{
    // one workgroup per frame: the tile width trades bytes in flight per frame against workgroups per CU
    // (measured on 2 MiB frames: 2048 frames -> 256 lanes.  128 lanes run the kernel alone in 8.9 instead of 9.5 ms -- eight
    // 2-wave workgroups per CU hold all 2048 frames in one round -- but leave no room for the neighbouring batch: with two
    // batches in flight the step is 18.8 ms against 17.2.  1024 frames: 256 lanes 4.95 ms, 128 lanes 6.9; 512 -> 512, 128 -> 1024)
#define ZK_EXEC_LAUNCH(TT, PP, CC) hipLaunchKernelGGL((zk_k_exec<TT, PP, CC>), dim3(count), dim3(TT), (TT) == 256 ? pad : 0, st, comp, d_off, first, ids, out_off, blocks, bases, infos, seqs, lit, dst, prefix, plen, progress)
    // ... and what a frame's matches reach: dense sequence streams (fewer than 10 output bytes per sequence: libzstd from level 3 up, whose
    // window is the whole 2 MiB frame -- every far match a line from beyond the L2) run better with FEWER frames resident: 512-lane tiles
    // hold 512 frames' histories live instead of 1 280 (4 GiB of the reference's level-3 frames: executor 17.0 -> 15.6 ms, the step with
    // two batches in flight 30.6 -> 29.5 ms; level 1, whose matches stay near: 8.4 -> 9.3 ms -- hence only there; tools/gpu_calls/r6a.sh, r6n.sh)
    const int lanes = k.exec_lanes ? k.exec_lanes : count >= 1024 ? (dense ? 512 : 256) : count >= 256 ? 512 : 1024;
    // 93 registers and 31 000 bytes of LDS: five 256-lane workgroups per CU.  Four of them leave 128 registers on every SIMD (room for
    // checksum waves, zk_k_xxh64_follow); a launch that asks for 2.5 KiB more LDS than the kernel uses gets four
    const uint32_t pad = k.exec_resident == 4 ? 2560u : 0u;
    if (prefix && plen) {
        if (lanes <= 256) ZK_EXEC_LAUNCH(256, true, 2); else if (lanes == 512) ZK_EXEC_LAUNCH(512, true, 2); else ZK_EXEC_LAUNCH(1024, true, 2);
        return;
    }
    // fewer than 10 output bytes per sequence (libzstd from level 3 up: ~8): a ring of 2 T records ends most 4 KiB tiles
    // early; 4 T keep them whole (executor -8.5 % there, +2 % on sparser streams, hence the switch)
    const bool ring4 = lanes == 256 && (k.exec_ring ? k.exec_ring == 2 : (count >= 1024 && dense));
    if (ring4) ZK_EXEC_LAUNCH(256, false, 4);
    else if (lanes == 128) ZK_EXEC_LAUNCH(128, false, 2);
    else if (lanes == 256) ZK_EXEC_LAUNCH(256, false, 2);
    else if (lanes == 512) ZK_EXEC_LAUNCH(512, false, 2);
    else ZK_EXEC_LAUNCH(1024, false, 2);
#undef ZK_EXEC_LAUNCH
}
void zk_launch_exec_seg(hipStream_t st, const uint8_t *comp, const uint64_t *d_off, uint32_t first, uint32_t count,
                        const uint32_t *ids, const uint64_t *out_off, const ZkBlock *blocks, const ZkFrameBase *bases, ZkFrameInfo *infos, const ZkSeqP *seqs,
                        const uint8_t *lit, uint8_t *dst, const ZkSegScratch &sg, const ZkKernelChoice &k, bool dense, uint64_t *progress)
{
    hipLaunchKernelGGL(zk_k_seg_prep, dim3(count), dim3(64), 0, st, blocks, bases, infos, d_off, first, ids, sg.seg_bytes, sg.segs, sg.nsegs, sg.max_segs);
    // a segment is a workgroup: wide tiles while the segments alone do not fill the device (a lone frame of 2 MiB: 16 segments), the
    // executor's 256 lanes beyond that
    const uint64_t wgs = (uint64_t)count * sg.max_segs;
    const int lanes = k.exec_lanes == 256 || k.exec_lanes == 1024 ? k.exec_lanes : wgs >= 1024 ? 256 : 1024;
    const dim3 grid(count, sg.max_segs);
#define ZK_SEG_LAUNCH(TT, CC) hipLaunchKernelGGL((zk_k_exec_seg<TT, CC>), grid, dim3(TT), 0, st, comp, d_off, first, ids, out_off, blocks, bases, infos, seqs, lit, dst, \
                                                 sg.segs, sg.nsegs, sg.max_segs, sg.holes, sg.tilecnt, sg.segn)
    if (lanes == 1024) ZK_SEG_LAUNCH(1024, 2);
    else if (k.exec_ring ? k.exec_ring == 2 : dense) ZK_SEG_LAUNCH(256, 4);
    else ZK_SEG_LAUNCH(256, 2);
#undef ZK_SEG_LAUNCH
    if (k.seg_fill ? k.seg_fill == 2 : count >= 512) hipLaunchKernelGGL((zk_k_exec_fill<256>), dim3(count), dim3(256), 0, st, d_off, first, ids, out_off, bases, infos, dst, sg.segs, sg.nsegs, sg.max_segs, sg.holes, sg.tilecnt, sg.segn, progress);
    else if (k.seg_fill == 1) hipLaunchKernelGGL((zk_k_exec_fill<1024>), dim3(count), dim3(1024), 0, st, d_off, first, ids, out_off, bases, infos, dst, sg.segs, sg.nsegs, sg.max_segs, sg.holes, sg.tilecnt, sg.segn, progress);
    else hipLaunchKernelGGL((zk_k_exec_fill_lds<1024>), dim3(count), dim3(1024), 0, st, d_off, first, ids, out_off, bases, infos, dst, sg.segs, sg.nsegs, sg.max_segs, sg.holes, sg.tilecnt, sg.segn, progress);
    // frames a segment gave up on (more hole records than its region holds): executed again, a workgroup per frame
    hipLaunchKernelGGL((zk_k_exec<256, false, 2, true>), dim3(count), dim3(256), 0, st, comp, d_off, first, ids, out_off, blocks, bases, infos, seqs, lit, dst,
                       (const uint8_t *)nullptr, (uint64_t)0, (uint64_t *)nullptr);
}
void zk_launch_xxh64(hipStream_t st, const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                     ZkFrameInfo *infos, uint64_t *hashes, const ZkKernelChoice &k, const uint64_t *skip, uint32_t wide_from, bool beside)
{
    // a large batch: sixteen frames per wave (the chains' latency is the same, the instruction slots a sixteenth).  From how many frames:
    // the decoder's pass runs alone on the device or beside a neighbour's entropy stage -- 512 frames of 2 MiB: a wave per frame 5.3 / 6.5 ms
    // per step against 5.9 / 7.6 (two batches in flight / one at a time), hence 1024; the encoder's runs beside its matcher, which pays for
    // every instruction slot the checksums take -- measured from 512 frames in round 3, and left there
    // (r6) the decoder's large batches: FOUR frames per workgroup, sixteen chains in one wave fed by another (zk_k_xxh64_fed<4>: 1.65 instead
    // of 2.7 ms per 2 MiB frame, three times the instructions); the encoder's pass, beside its entropy stage and shorter than it either way,
    // keeps sixteen frames per wave; 2 / 3 pin the earlier forms for the tests and the probes
    if (!skip && k.xxh == 3) hipLaunchKernelGGL(zk_k_xxh64_lean, dim3((count + 15) / 16), dim3(64), 0, st, data, d_off, first, count, infos, hashes);
    else if (k.xxh == 2 || (beside && k.xxh == 0 && count >= wide_from)) hipLaunchKernelGGL(zk_k_xxh64_wide, dim3((count + 15) / 16), dim3(64), 0, st, data, d_off, first, count, infos, hashes, skip);
    else if (skip || k.xxh >= 4 || (k.xxh == 0 && count >= wide_from)) hipLaunchKernelGGL(zk_k_xxh64_fed<4>, dim3((count + 3) / 4), dim3(128), 0, st, data, d_off, first, count, infos, hashes, skip);
    else hipLaunchKernelGGL(zk_k_xxh64, dim3(count), dim3(64), 0, st, data, d_off, first, count, infos, hashes);
}
void zk_launch_xxh64_follow(hipStream_t st, const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count, const ZkFrameInfo *infos, uint64_t *progress)
{
    // a handful of frames: a wave per frame (a frame's chains run 1.7 ms per 2 MiB there, 2.25 ms sixteen frames to a wave)
    if (count <= 32) { hipLaunchKernelGGL(zk_k_xxh64_follow1, dim3(8 * ((count + 7) / 8)), dim3(64), 0, st, data, d_off, first, count, infos, progress); return; }
    // eight waves per 128 frames: each takes the frames of one XCD
    hipLaunchKernelGGL(zk_k_xxh64_follow, dim3(8 * ((count + 127) / 128)), dim3(64), 0, st, data, d_off, first, count, infos, progress);
}

void zk_launch_small_walk(hipStream_t st, const uint8_t *h_comp, uint64_t comp_bytes, const uint64_t *h_offs, uint32_t count, uint64_t dst_cap,
                          uint32_t block_cap, uint64_t seq_cap, uint8_t *d_comp, uint64_t *d_offs, ZkFrameInfo *infos, ZkFrameBase *bases, ZkBlock *blocks, uint64_t *words)
{
    hipLaunchKernelGGL(zk_k_small_walk, dim3(1), dim3(256), 0, st, h_comp, comp_bytes, h_offs, count, dst_cap, block_cap, seq_cap, d_comp, d_offs, infos, bases, blocks, words);
}
void zk_launch_small_entropy(hipStream_t st, const uint8_t *comp, ZkBlock *blocks, const uint64_t *words, uint8_t *lit, ZkSeqP *seqs, uint32_t groups, bool split)
{
    if (split) {                                             // (diagnosis: the two roles show separately in a kernel trace)
        hipLaunchKernelGGL(zk_k_small_huf, dim3(groups), dim3(128), 0, st, comp, blocks, words, lit);
        hipLaunchKernelGGL(zk_k_small_fse, dim3(groups), dim3(192), 0, st, comp, blocks, words, seqs);
        return;
    }
    hipLaunchKernelGGL(zk_k_small_entropy, dim3(2 * groups), dim3(192), 0, st, comp, blocks, words, lit, seqs);
}
void zk_launch_small_publish(hipStream_t st, const ZkFrameInfo *infos, const uint64_t *d_offs, uint32_t count, const uint8_t *dst, uint8_t *h_out,
                             int32_t *d_status, int32_t *h_status, uint64_t *words, uint32_t *h_flag, uint32_t gen)
{
    hipLaunchKernelGGL(zk_k_small_publish, dim3(count), dim3(256), 0, st, infos, d_offs, count, dst, h_out, d_status, h_status, words, h_flag, gen);
}
