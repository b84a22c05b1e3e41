// zk_sim.cpp -- TEST HARNESS (never shipped, never linked into libzeekstd_amd.so).
// Runs the per-lane device code of zeekstd_amd/csrc/zk_device.h on the CPU, lane after lane,
// with the same orchestration as the kernels in zk_decode.hip (walk -> scan -> walk -> huf ->
// fse -> exec in chunks/tiles).  Lets the CPU test suite check the kernel logic against the
// oracle without a GPU.  Tile writes are committed only after every "lane" of the tile ran, so
// an illegal in-tile history read would surface as a mismatch.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <vector>
#include "../../zeekstd_amd/csrc/zk_device.h"

// ---- a quad of lanes on the CPU: fibers in lock step ---------------------------------------------------------------
// zk_seq_walk_quad / zk_seq_finish_quad (zk_device.h) are written for lanes that meet in every XCH::bcast (a DPP move on the
// device).  Here each lane is a ucontext fiber; bcast parks the lane's value and yields, the scheduler resumes the lanes round
// robin, so every lane reaches exchange r before any lane leaves it (values are double buffered by exchange parity).
// The device runs the two halves as two waves with an 8-entry LDS ring between them (zk_decode.hip, zk_fse_quad_group); here the
// walk of a block runs to its end first (three lanes, values into an array), then the finisher (four lanes, rounds of four).
struct ZkQuadSim {
    ucontext_t main_ctx, ctx[4];
    std::vector<char> stack[4];
    uint32_t buf[2][4], round[4];
    int cur, nlanes;
    bool done[4];
    // arguments of the walk
    const uint8_t *comp; ZkBlock b; uint32_t bs_off; ZkSeqTablesX16 *T; ZkSeqTablesT<ZkCells64> *T64; const uint32_t *al; ZkSeqP *seqs;
    std::vector<uint32_t> vals;                  // [sequence][lane]
    uint32_t walk_bad[3], gates;
    ZkSeqCarry carry[4];
};
static ZkQuadSim *g_quad;
static uint32_t g_ofv[32], g_llb[36], g_mlb[53];      // the walker's baseline tables (zk_fse_quad_group builds the same in LDS)
static int g_fse_quad = 0;
extern "C" void zk_sim_set_fse_quad(int on) { g_fse_quad = on; }
// segmented execution (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill): seg_bytes == 0 -> one workgroup per frame (zk_k_exec)
static uint32_t g_seg_bytes = 0, g_fill_lanes = 64, g_seg_cap_shift = 2;
static uint64_t g_seg_stats[4];                 // holes records, hole bytes, fill steps, frames that overflowed (executed again)
extern "C" void zk_sim_set_exec_seg(uint32_t seg_bytes, uint32_t fill_lanes, uint32_t cap_shift) { g_seg_bytes = seg_bytes; g_fill_lanes = fill_lanes ? fill_lanes : 64; g_seg_cap_shift = cap_shift; }
static int g_chase_stats_on = 0;
static uint64_t g_slot_stats[2];
static uint64_t g_huf_depth[16];
extern "C" void zk_sim_huf_depth(uint64_t *out) { for (int i = 0; i < 16; i++) { out[i] = g_huf_depth[i]; g_huf_depth[i] = 0; } }
extern "C" void zk_sim_slot_stats(uint64_t *out) { out[0] = g_slot_stats[0]; out[1] = g_slot_stats[1]; g_slot_stats[0] = g_slot_stats[1] = 0; }
static uint64_t g_chase_stats[32];       // [0..15]: waves by their deepest chain, [16..31]: bytes by the depth of their chain
extern "C" void zk_sim_chase_stats(uint64_t *out, int on) { for (int i = 0; i < 32; i++) { out[i] = g_chase_stats[i]; g_chase_stats[i] = 0; } g_chase_stats_on = on; }
extern "C" void zk_sim_seg_stats(uint64_t *out, int reset) { for (int i = 0; i < 4; i++) { out[i] = g_seg_stats[i]; if (reset) g_seg_stats[i] = 0; } }
// poison: the tables and scratch a block's lane builds and reads (on the device: LDS that holds whatever the workgroup before left
// there) are filled with pseudo-random bytes before every block -- code that reads what it has not written shows up as a mismatch
static uint64_t g_poison = 0;
extern "C" void zk_sim_set_poison(uint64_t seed) { g_poison = seed; }
static void zk_sim_poison(void *p, size_t n)
{
    if (!g_poison) return;
    uint8_t *b = (uint8_t *)p;
    for (size_t i = 0; i < n; i++) { g_poison = g_poison * 6364136223846793005ull + 1442695040888963407ull; b[i] = (uint8_t)(g_poison >> 56); }
}
struct ZkQuadFibers {
    static uint32_t bcast(uint32_t v, int k)
    {
        ZkQuadSim *q = g_quad;
        const int me = q->cur;
        const uint32_t r = q->round[me]++;
        q->buf[r & 1][me] = v;
        swapcontext(&q->ctx[me], &q->main_ctx);
        return q->buf[r & 1][k];
    }
};
struct ZkQuadSimOut {                            // OUT of zk_seq_walk_quad: the lane's column of the value array
    ZkQuadSim *q; int t;
    void gate(uint32_t i, uint32_t) { if (i % ZK_QUAD_ROUND != 0) q->gates |= 0x80000000u; q->gates++; }
    void put(uint32_t i0, uint32_t k, uint32_t v) { if (i0 % ZK_QUAD_ROUND != 0 || k >= ZK_QUAD_ROUND) q->gates |= 0x80000000u; q->vals[(size_t)(i0 + k) * 4 + (size_t)t] = v; }
};

static const uint32_t LLV[36] = ZK_LL_TABLE;
static const uint32_t MLV[53] = ZK_ML_TABLE;

static const uint32_t LLV_[36] = ZK_LL_TABLE;
static const uint32_t MLV_[53] = ZK_ML_TABLE;
static void zk_quad_walk_lane_main()
{
    ZkQuadSim *q = g_quad;
    const int t = q->cur;
    ZkQuadSimOut out{q, t};
    if (q->T64)                                  // the small-batch kernels: 8-byte cells, no value tables
        q->walk_bad[t] = zk_seq_walk_quad<ZkRevU, ZkCells64, ZkQuadFibers>(q->comp, q->b, q->bs_off, (uint32_t)t,
                                                      t == ZK_TAB_LL ? q->T64->ll : t == ZK_TAB_OF ? q->T64->of : q->T64->ml, nullptr, q->al, out);
    else
        q->walk_bad[t] = zk_seq_walk_quad<ZkRevU, ZkCellsX16, ZkQuadFibers>(q->comp, q->b, q->bs_off, (uint32_t)t,
                                                      t == ZK_TAB_LL ? q->T->ll : t == ZK_TAB_OF ? q->T->of : q->T->ml,
                                                      t == ZK_TAB_LL ? g_llb : t == ZK_TAB_OF ? g_ofv : g_mlb, q->al, out);
    q->done[t] = true;
}
static void zk_quad_finish_lane_main()
{
    ZkQuadSim *q = g_quad;
    const int j = q->cur;
    const uint32_t nseq = q->b.nseq;
    ZkSeqCarry &c = q->carry[j];
    zk_seq_carry_init(c);
    for (uint32_t i0 = 0; i0 < nseq; i0 += ZK_QUAD_ROUND) {
        const uint32_t nvalid = nseq - i0 < ZK_QUAD_ROUND ? nseq - i0 : ZK_QUAD_ROUND;
        const bool have = (uint32_t)j < nvalid;
        const uint32_t *v = &q->vals[(size_t)(i0 + (have ? j : 0)) * 4];
        // a lane without a sequence brings garbage, as a stale ring entry would
        const ZkSeqP rec = zk_seq_finish_quad<ZkQuadFibers>((uint32_t)j, nvalid, have ? v[0] : 0xDEADBEEFu, have ? v[1] : 0xDEADBEEFu, have ? v[2] : 0xDEADBEEFu, c, q->b.lit_regen);
        if (have) q->seqs[i0 + j] = rec;
    }
    q->done[j] = true;
}
static void zk_quad_run(ZkQuadSim &q, int nlanes, void (*lane_main)())
{
    q.nlanes = nlanes;
    for (int l = 0; l < nlanes; l++) {
        q.round[l] = 0; q.done[l] = false;
        q.stack[l].resize(256 << 10);
        getcontext(&q.ctx[l]);
        q.ctx[l].uc_stack.ss_sp = q.stack[l].data();
        q.ctx[l].uc_stack.ss_size = q.stack[l].size();
        q.ctx[l].uc_link = &q.main_ctx;
        makecontext(&q.ctx[l], lane_main, 0);
    }
    for (;;) {
        bool any = false;
        for (int l = 0; l < nlanes; l++)
            if (!q.done[l]) { any = true; q.cur = l; swapcontext(&q.main_ctx, &q.ctx[l]); }
        if (!any) break;
    }
}
// the block's sequences through the quad walk (tables already built in T); false: the lanes disagree
static bool zk_quad_walk_sim(const uint8_t *comp, ZkBlock &b, uint32_t bs_off, ZkSeqTablesX16 *T, const uint32_t *al, ZkSeqP *seqs, ZkSeqTablesT<ZkCells64> *T64 = nullptr)
{
    ZkQuadSim q;
    q.T64 = T64;
    g_quad = &q;
    for (uint32_t k = 0; k < 32; k++) g_ofv[k] = 1u << k;
    for (uint32_t k = 0; k < 36; k++) g_llb[k] = LLV_[k] & 0xFFFFFFu;
    for (uint32_t k = 0; k < 53; k++) g_mlb[k] = MLV_[k] & 0xFFFFFFu;
    q.comp = comp; q.b = b; q.bs_off = bs_off; q.T = T; q.al = al; q.seqs = seqs; q.gates = 0;
    q.vals.assign((size_t)b.nseq * 4 + 4, 0xA5A5A5A5u);
    zk_quad_run(q, 3, zk_quad_walk_lane_main);
    bool same = q.walk_bad[0] == q.walk_bad[1] && q.walk_bad[1] == q.walk_bad[2] && q.round[0] == q.round[1] && q.round[1] == q.round[2];
    same = same && (q.walk_bad[0] || q.gates == 3 * ((b.nseq + ZK_QUAD_ROUND - 1) / ZK_QUAD_ROUND));       // a gate in front of every round, by every lane
    zk_quad_run(q, 4, zk_quad_finish_lane_main);
    for (int l = 1; l < 4; l++)
        same = same && memcmp(&q.carry[l], &q.carry[0], sizeof(ZkSeqCarry)) == 0 && q.round[l] == q.round[0];
    g_quad = nullptr;
    zk_seq_finish_block(b, q.carry[0], q.walk_bad[0]);
    return same;
}

extern "C" int zk_sim_decode_prefix(const uint8_t *comp, const uint64_t *c_off, const uint64_t *d_off, uint32_t first,
                                    uint32_t count, uint8_t *dst, int32_t *status, int exec_b, int exec_chunk,
                                    const uint8_t *prefix, uint64_t plen);
extern "C" int zk_sim_decode(const uint8_t *comp, const uint64_t *c_off, const uint64_t *d_off, uint32_t first,
                             uint32_t count, uint8_t *dst, int32_t *status, int exec_b, int exec_chunk)
{
    return zk_sim_decode_prefix(comp, c_off, d_off, first, count, dst, status, exec_b, exec_chunk, nullptr, 0);
}

extern "C" int zk_sim_decode_prefix(const uint8_t *comp, const uint64_t *c_off, const uint64_t *d_off, uint32_t first,
                                    uint32_t count, uint8_t *dst, int32_t *status, int exec_b, int exec_chunk,
                                    const uint8_t *prefix, uint64_t plen)
{
    if (!prefix) plen = 0;
    const bool PFX = plen != 0;
    std::vector<ZkFrameInfo> infos(count);
    std::vector<ZkFrameBase> bases(count);
    uint64_t nb = 0, ns = 0, nl = 0;
    for (uint32_t f = 0; f < count; f++) {
        ZkFrameInfo fi;
        uint64_t dsz = d_off[first + f + 1] - d_off[first + f];
        zk_walk_frame(comp, c_off[first + f], c_off[first + f + 1], dsz, f, nullptr, nullptr, fi);
        if (dsz > ZK_MAX_FRAME && fi.status == ZK_OK) fi.status = ZK_E_FRAMEPARAM_UNSUPPORTED;
        if (fi.status != ZK_OK) { fi.n_blocks = 0; fi.n_seq = 0; fi.lit_bytes = 0; }
        infos[f] = fi;
        bases[f].block_base = nb; bases[f].seq_base = ns; bases[f].lit_base = nl;
        nb += fi.n_blocks; ns += fi.n_seq; nl += fi.lit_bytes;
    }
    std::vector<ZkBlock> blocks(nb + 1);
    std::vector<ZkSeqP> seqs(ns + 1);
    std::vector<uint8_t> lit(nl + 64);
    for (uint32_t f = 0; f < count; f++) {
        if (infos[f].status != ZK_OK) continue;
        ZkFrameInfo fi;
        zk_walk_frame(comp, c_off[first + f], c_off[first + f + 1], d_off[first + f + 1] - d_off[first + f], f, &bases[f], blocks.data(), fi);
    }
    // huf: one "lane" per stream
    std::vector<uint16_t> tab(2048);
    ZkHufHdr hd;
    ZkHufTmp tmp;
    for (uint64_t bi = 0; bi < nb; bi++) {
        ZkBlock &b = blocks[bi];
        if (!(b.type == 2 && b.lit_type >= 2 && b.status == ZK_OK)) continue;
        const ZkBlock &def = blocks[b.huf_def];
        uint32_t mb = 0, nsym = 0;
        zk_sim_poison(tab.data(), tab.size() * 2); zk_sim_poison(&hd, sizeof hd); zk_sim_poison(&tmp, sizeof tmp);
        uint32_t r = zk_huf_read_weights(comp + def.src + def.lit_off, def.lit_comp, &hd, &tmp, &nsym, &mb);
        if (g_chase_stats_on && r) g_huf_depth[mb < 16 ? mb : 15]++;
        bool ok = r != 0;
        if (ok) {
            zk_huf_fill_table(tab.data(), &hd, nsym, mb);
            const uint8_t *pay = comp + b.src + b.lit_off;
            uint32_t size = b.lit_comp;
            if (b.lit_type == 2) { pay += r; size -= r; }
            uint8_t *d = lit.data() + b.lit_base;
            uint32_t regen = b.lit_regen;
            if (b.lit_streams == 1) ok = zk_huf_decode_stream(tab.data(), mb, pay, size, d, regen);
            else if (size < 6) ok = false;
            else {
                uint32_t s1 = zk_rd16(pay), s2 = zk_rd16(pay + 2), s3 = zk_rd16(pay + 4), q = (regen + 3) / 4;
                if (6 + s1 + s2 + s3 > size || 3 * q > regen) ok = false;
                else {
                    uint32_t s4 = size - 6 - s1 - s2 - s3;
                    for (uint32_t stream = 0; stream < 4 && ok; stream++) {
                        uint32_t start = 6 + (stream > 0 ? s1 : 0) + (stream > 1 ? s2 : 0) + (stream > 2 ? s3 : 0);
                        uint32_t len = stream == 0 ? s1 : stream == 1 ? s2 : stream == 2 ? s3 : s4;
                        uint32_t n = stream == 3 ? regen - 3 * q : q;
                        ok = zk_huf_decode_stream(tab.data(), mb, pay + start, len, d + stream * q, n);
                    }
                }
            }
        }
        if (!ok) b.status = ZK_E_CORRUPTION;
    }
    // fse: one "lane" per block
    ZkSeqTables *T = new ZkSeqTables;
    ZkSeqTables16 *T16 = new ZkSeqTables16;
    ZkSeqTablesX16 *TX16 = new ZkSeqTablesX16;
    ZkSeqTablesT<ZkCells64> *T64 = new ZkSeqTablesT<ZkCells64>;
    for (uint64_t bi = 0; bi < nb; bi++) {
        ZkBlock b = blocks[bi];
        if (b.type != 2 || b.nseq == 0 || b.status != ZK_OK) continue;
        zk_sim_poison(T, sizeof *T); zk_sim_poison(T16, sizeof *T16); zk_sim_poison(TX16, sizeof *TX16); zk_sim_poison(T64, sizeof *T64);
        // like the device: all-predefined blocks go through the aligned-word reader, the rest through the unaligned one
        // (with the quad flag every block takes the quad walk, as small batches do on the device)
        if (b.seq_modes == 0 && !g_fse_quad) zk_decode_sequences<ZkRevA, ZkCells32>(comp, blocks.data(), b, T, seqs.data() + b.seq_base, LLV, MLV);
        else if (!g_fse_quad) zk_decode_sequences<ZkRevU, ZkCells16>(comp, blocks.data(), b, T16, seqs.data() + b.seq_base, LLV, MLV);     // zk_k_fse: compact cells
        else {                                                      // zk_k_fse_quad: the kernel's table setup, then three lock-stepped lanes
            uint32_t al[3], own = 0;
            bool ok = true;
            for (int u = 0; u < 3 && ok; u++) {
                const uint32_t m = (b.seq_modes >> (6 - 2 * u)) & 3;
                const ZkBlock &def = m == 3 ? blocks[b.tab_def[u]] : b;
                const int32_t r = g_fse_quad == 2 ? zk_seq_table_setup<ZkCells64>(comp, def, u, T64, &al[u], LLV, MLV)
                                                  : zk_seq_table_setup<ZkCellsX16>(comp, def, u, TX16, &al[u], LLV, MLV);
                if (r < 0) ok = false;
                else if (m != 3) own += (uint32_t)r;
            }
            if (!ok) b.status = ZK_E_CORRUPTION;
            else if (!zk_quad_walk_sim(comp, b, b.seq_off + 1 + own, TX16, al, seqs.data() + b.seq_base, g_fse_quad == 2 ? T64 : nullptr)) b.status = ZK_E_CORRUPTION + 1000;   // lanes out of step: a bug, not an input error
        }
        blocks[bi].out_size = b.out_size;
        for (int k = 0; k < 3; k++) blocks[bi].rep_out[k] = b.rep_out[k];
        blocks[bi].status = b.status;
    }
    delete T;
    delete T16;
    delete TX16;
    delete T64;
    // exec: one "workgroup" per frame; tiles of THREADS x B bytes, slot marking + per-byte source map (zk_exec_slot_span / zk_exec_slot_words)
    // CAPS: sequences staged at once.  Like the kernel the staged records live in a ring (slot = block sequence index & mask):
    // a tile retires the sequences it consumed and as many new records move into their slots.
    const uint32_t THREADS = 256, B = (uint32_t)exec_b, CAPS = (uint32_t)exec_chunk;
    uint32_t RING = 2;
    while (RING < CAPS) RING <<= 1;
    const uint32_t M = RING - 1;
    std::vector<ZkSeq> st(RING);
    std::vector<uint32_t> srcmap(THREADS * B), slot_seq(THREADS * B / ZK_EXEC_SLOT + 1);
    std::vector<uint8_t> tile(THREADS * B);
    std::vector<uint32_t> slow(THREADS * B / ZK_EXEC_SLOT / 32 + 1);        // the tile's slots for the general walk (zk_exec_mark_runs)
    // SEGMENT MODE (sg != nullptr): the bytes whose origin lies before the segment, or at a tainted byte of it, are not written
    // (poisoned here); their runs go to the hole list, their taint bits are set once the tile is done
    struct SegCtx { int32_t seg_lo; uint32_t seg_done; std::vector<uint32_t> *taint; std::vector<ZkHole> *holes; std::vector<uint32_t> *tiles; uint32_t cap; bool overflow; };
    // one compressed block: bout = its first output byte, pos = that byte's place in the frame
    auto exec_block = [&](const ZkBlock &b, const ZkFrameInfo &fi, uint8_t *out, uint64_t pos, const uint32_t rep[3], SegCtx *sg) -> uint32_t {
        uint32_t err = ZK_OK;
        uint8_t *bout = out + pos;
        const ZkSeqP *sq = seqs.data() + b.seq_base;
        const uint8_t *l = b.lit_type >= 2 ? lit.data() + b.lit_base : comp + b.src + b.lit_off;
        const uint32_t lit_mask = b.lit_type == 1 ? 0u : 0x7fffffffu;
        const uint32_t nseq = b.nseq, out_size = b.out_size;
        int bad = 0;
        auto stage = [&](uint32_t from, uint32_t to) {                          // records [from, to) -> ring, as the kernel's fetch + settle
            for (uint32_t idx = from; idx < to; idx++) {
                ZkSeq r;
                if (idx < nseq) {
                    const ZkSeq q = zk_seq_unpack(idx ? sq[idx - 1] : 0, sq[idx], idx == 0);
                    const uint32_t off = zk_rep_resolve(q.off, rep);
                    const uint32_t mstart = q.out_end - q.ml;
                    if (PFX ? (off == 0 || pos + mstart + plen < off) : (off == 0 || pos + mstart < off || off > fi.window)) bad = 1;
                    if (off >= ZK_SRC_BIAS || q.ml > q.out_end) bad = 1;
                    r.out_end = q.out_end; r.ml = q.ml; r.off = off; r.lit_end = q.lit_end;
                } else { r.out_end = out_size; r.ml = 0; r.off = 1; r.lit_end = b.lit_regen; }
                st[idx & M] = r;
            }
        };
        uint32_t ja = 0, ts = 0, prev_end = 0;
        uint32_t staged_end = nseq + 1 < CAPS ? nseq + 1 : CAPS;
        stage(0, staged_end);
        if (bad) err = ZK_E_CORRUPTION;
        while (err == ZK_OK && ts < out_size) {
            const uint32_t nl = staged_end - ja;
            const uint32_t cap_end = st[(staged_end - 1) & M].out_end;
            const uint32_t te = ts + THREADS * B < cap_end ? ts + THREADS * B : cap_end;
            uint32_t jn = nl;
            std::fill(srcmap.begin(), srcmap.end(), 0u);                        // the map starts empty (run markers)
            std::fill(slot_seq.begin(), slot_seq.end(), 0xFFFFFFFFu);
            std::fill(slow.begin(), slow.end(), 0u);
            for (uint32_t i = 0; i < nl; i++) {                                  // "lane per sequence"
                const uint32_t idx = ja + i;
                const uint32_t end = st[idx & M].out_end;
                const uint32_t start = i ? st[(idx - 1) & M].out_end : prev_end;
                const uint32_t prev_off = i ? st[(idx - 1) & M].off : 0xDEADBEEFu;   // (a sequence that has left the ring: its offset is never needed, and the kernel does not have it)
                const uint32_t lo = start > ts ? start : ts, hi = end < te ? end : te;
                if (lo < hi) {
                    uint32_t s0, n;
                    zk_exec_slot_span(ts, lo, hi, s0, n);
                    for (uint32_t k = 0; k < n; k++) slot_seq[s0 + k] = idx;
                    zk_exec_mark_runs(st[idx & M], prev_off, start, ts, te, srcmap.data(), [&](uint32_t w, uint32_t bits) { slow[w] |= bits; });
                }
                if (end > te && start <= te) jn = i;
            }
            for (uint32_t q0 = ts; q0 < te; q0 += ZK_EXEC_SLOT) {                // "lane per slot"
                const uint32_t nb = te - q0 < ZK_EXEC_SLOT ? te - q0 : ZK_EXEC_SLOT;
                uint32_t sw[ZK_EXEC_SLOT];
                uint32_t mk[ZK_EXEC_SLOT];
                for (uint32_t k = 0; k < ZK_EXEC_SLOT; k++) mk[k] = q0 - ts + k < srcmap.size() ? srcmap[zk_exec_map_index(q0 - ts + k)] : 0u;
                const uint32_t slot = (q0 - ts) / ZK_EXEC_SLOT;
                if (g_chase_stats_on) { g_slot_stats[0]++; g_slot_stats[1] += (slow[slot >> 5] >> (slot & 31u)) & 1u; }
                zk_exec_slot_words_marked(st.data(), slot_seq[slot], (slow[slot >> 5] >> (slot & 31u)) & 1u, q0, nb, mk, sw, M);
                for (uint32_t k = 0; k < nb; k++) srcmap[zk_exec_map_index(q0 - ts + k)] = sw[k];
            }
            const uint32_t mbase = ZK_SRC_BIAS + ts, span = te - ts;
            if (!sg) {
                if (g_chase_stats_on) {                                          // statistics (tools/chase_stats.py): per "wave" of 64 slots the deepest chain, per byte its depth
                    for (uint32_t w0 = ts; w0 < te; w0 += 64 * ZK_EXEC_SLOT) {
                        uint32_t deepest = 0;
                        for (uint32_t q = w0; q < te && q < w0 + 64 * ZK_EXEC_SLOT; q++) {
                            uint32_t s = srcmap[zk_exec_map_index(q - ts)], dep = 0;
                            while (s - mbase < span) { s = srcmap[zk_exec_map_index(s - mbase)]; dep++; }
                            g_chase_stats[16 + (dep < 15 ? dep : 15)]++;
                            deepest = dep > deepest ? dep : deepest;
                        }
                        g_chase_stats[deepest < 15 ? deepest : 15]++;
                    }
                }
                for (uint32_t q = ts; q < te; q++) {                             // origins + gathers
                    uint32_t s = srcmap[zk_exec_map_index(q - ts)];
                    while (s - mbase < span) s = srcmap[zk_exec_map_index(s - mbase)];
                    if (s & ZK_SRC_LIT) tile[q - ts] = l[(s & lit_mask)];
                    else {
                        const int64_t rel = (int64_t)pos + (int32_t)(s - ZK_SRC_BIAS);
                        tile[q - ts] = PFX && rel < 0 ? prefix[(int64_t)plen + rel] : out[rel];
                    }
                }
            } else {
                // lane per slot again, the waves of the workgroup in REVERSE order (the order of a tile's records is whatever the waves'
                // atomics make it: nothing may depend on it)
                const uint32_t nslots = (span + ZK_EXEC_SLOT - 1) / ZK_EXEC_SLOT;
                std::vector<std::pair<uint32_t, uint32_t>> taint_or;             // (segment-relative first byte, 16 hole bits): set behind the tile
                const size_t nrec0 = sg->holes->size();
                for (int32_t wv = (int32_t)((nslots + 63) / 64) - 1; wv >= 0; wv--)
                    for (uint32_t sl = (uint32_t)wv * 64; sl < nslots && sl < (uint32_t)(wv + 1) * 64; sl++) {
                        const uint32_t q0 = ts + sl * ZK_EXEC_SLOT;
                        const uint32_t nb = te - q0 < ZK_EXEC_SLOT ? te - q0 : ZK_EXEC_SLOT;
                        uint32_t sw[ZK_EXEC_SLOT], len[ZK_EXEC_SLOT];
                        for (uint32_t k = 0; k < ZK_EXEC_SLOT; k++) {
                            uint32_t s = k < nb ? srcmap[zk_exec_map_index(q0 - ts + k)] : ZK_SRC_LIT;
                            while (s - mbase < span) s = srcmap[zk_exec_map_index(s - mbase)];
                            sw[k] = s;
                        }
                        const std::vector<uint32_t> &tb = *sg->taint;
                        const uint32_t hm = zk_seg_slot_holes(sw, nb, sg->seg_lo, [&](uint32_t p) { return (bool)((tb[p >> 5] >> (p & 31)) & 1u); });
                        const uint32_t starts = zk_seg_slot_runs(sw, hm, len);
                        for (uint32_t k = 0; k < nb; k++) {
                            const uint32_t s = sw[k];
                            if ((hm >> k) & 1u) tile[q0 - ts + k] = 0xEE;            // (the kernel leaves the byte alone)
                            else if (s & ZK_SRC_LIT) tile[q0 - ts + k] = l[(s & lit_mask)];
                            else tile[q0 - ts + k] = out[(int64_t)pos + (int32_t)(s - ZK_SRC_BIAS)];
                            if ((starts >> k) & 1u) {
                                if (sg->holes->size() >= sg->cap) sg->overflow = true;
                                else sg->holes->push_back(zk_hole_pack((uint32_t)pos + q0 + k, len[k], q0 + k + ZK_SRC_BIAS - s));
                            }
                        }
                        if (hm) taint_or.push_back({sg->seg_done + q0, hm});
                    }
                for (auto &t : taint_or)
                    for (uint32_t k = 0; k < ZK_EXEC_SLOT; k++)
                        if ((t.second >> k) & 1u) { const uint32_t p = t.first + k; (*sg->taint)[p >> 5] |= 1u << (p & 31); }
                if (sg->holes->size() > nrec0) sg->tiles->push_back((uint32_t)(sg->holes->size() - nrec0));      // the tile's count (zk_k_exec_fill: one round per tile)
            }
            memcpy(bout + ts, tile.data(), te - ts);                             // commit the tile after all lanes ran
            const uint32_t next_prev_end = jn ? st[(ja + jn - 1) & M].out_end : prev_end;
            const uint32_t fetch_end = staged_end + jn < nseq + 1 ? staged_end + jn : nseq + 1;
            stage(staged_end, fetch_end);                                        // the retired slots take the next records
            if (bad) { err = ZK_E_CORRUPTION; break; }
            prev_end = next_prev_end;
            ja += jn; staged_end = fetch_end; ts = te;
        }
        return err;
    };
    int first_err = 0;
    const bool seg_mode = g_seg_bytes != 0 && !PFX;
    for (uint32_t f = 0; f < count; f++) {
        const ZkFrameInfo fi = infos[f];
        uint32_t err = fi.status;
        const uint64_t d_size = d_off[first + f + 1] - d_off[first + f];
        uint8_t *out = dst + (d_off[first + f] - d_off[first]);
        const ZkBlock *fb = blocks.data() + bases[f].block_base;
        const uint32_t block_max = fi.window < ZK_BLOCK_MAX ? fi.window : ZK_BLOCK_MAX;
        bool redo = false;
        if (err == ZK_OK && seg_mode) {
            // ---- zk_k_seg_prep: the frame cut into segments
            const uint32_t max_segs = 2 * (uint32_t)((d_size + g_seg_bytes - 1) / g_seg_bytes) + 1;
            std::vector<ZkSeg> segs(max_segs);
            ZkSegWalk w;
            zk_seg_walk_init(w);
            for (uint32_t bk = 0; bk < fi.n_blocks; bk++) zk_seg_step(w, bk, fb[bk].status, fb[bk].out_size, fb[bk].rep_out, d_size, block_max, g_seg_bytes, segs.data(), max_segs);
            zk_seg_walk_end(w, d_size, segs.data(), max_segs);
            err = w.err;
            // ---- zk_k_exec_seg: every segment on its own (here: LAST segment first -- no segment may need another one's bytes)
            std::vector<std::vector<ZkHole>> holes(w.nsegs);
            std::vector<std::vector<uint32_t>> tiles(w.nsegs);
            if (err == ZK_OK) memset(out, 0xDD, d_size);                         // whatever the buffer held: nothing may be read before it is written
            for (int32_t j = (int32_t)w.nsegs - 1; j >= 0 && err == ZK_OK; j--) {
                const ZkSeg &s = segs[j];
                std::vector<uint32_t> taint((s.out >> 5) + 2, 0u);
                SegCtx sg{0, 0, &taint, &holes[j], &tiles[j], ((s.out >> g_seg_cap_shift) + 8u), false};
                uint64_t pos = s.pos;
                uint32_t rep[3] = {s.rep[0], s.rep[1], s.rep[2]};
                uint32_t serr = ZK_OK;
                for (uint32_t bk = s.b0; bk < s.b0 + s.nb && serr == ZK_OK; bk++) {
                    const ZkBlock &b = fb[bk];
                    sg.seg_done = (uint32_t)(pos - s.pos); sg.seg_lo = -(int32_t)sg.seg_done;
                    if (b.type == 0) memcpy(out + pos, comp + b.src, b.bsize);
                    else if (b.type == 1) memset(out + pos, comp[b.src], b.bsize);
                    else {
                        serr = exec_block(b, fi, out, pos, rep, &sg);
                        if (serr == ZK_OK) {
                            uint32_t r0 = zk_rep_resolve(b.rep_out[0], rep), r1 = zk_rep_resolve(b.rep_out[1], rep), r2 = zk_rep_resolve(b.rep_out[2], rep);
                            rep[0] = r0; rep[1] = r1; rep[2] = r2;
                        }
                    }
                    pos += b.out_size;
                }
                if (serr != ZK_OK) err = serr;
                else if (sg.overflow) redo = true;
            }
            // ---- zk_k_exec_fill: segment after segment, tile after tile; a tile's records in rounds of L lanes (every lane loads, then
            //      every lane stores).  What makes a tile one round: no record's source overlaps a destination of the tile (checked here)
            if (err == ZK_OK && !redo) {
                const uint32_t L = g_fill_lanes;
                for (uint32_t j = 0; j < w.nsegs && err == ZK_OK; j++) {
                    const std::vector<ZkHole> &h = holes[j];
                    g_seg_stats[0] += h.size();
                    size_t i = 0;
                    for (uint32_t cnt : tiles[j]) {
                        uint32_t dmin = 0xFFFFFFFFu;
                        for (size_t k = 0; k < cnt; k++) dmin = zk_hole_dst(h[i + k]) < dmin ? zk_hole_dst(h[i + k]) : dmin;
                        for (size_t k = 0; k < cnt; k++) if (!zk_fill_ready(h[i + k], dmin)) err = ZK_E_GENERIC + 2000;   // a bug, not an input error
                        // the rounds run in REVERSE order, lanes too: nothing may depend on the order inside a tile
                        for (size_t r0 = (cnt + L - 1) / L; r0-- > 0;) {
                            const size_t lo = r0 * L, n = cnt - lo < L ? cnt - lo : L;
                            std::vector<uint8_t> got(n * 16);
                            for (size_t k = 0; k < n; k++)
                                for (uint32_t x = 0; x < zk_hole_len(h[i + lo + k]); x++) got[k * 16 + x] = out[zk_hole_dst(h[i + lo + k]) - zk_hole_off(h[i + lo + k]) + x];
                            for (size_t k = n; k-- > 0;)
                                for (uint32_t x = 0; x < zk_hole_len(h[i + lo + k]); x++) { out[zk_hole_dst(h[i + lo + k]) + x] = got[k * 16 + x]; g_seg_stats[1]++; }
                        }
                        i += cnt;
                        g_seg_stats[2]++;
                    }
                    if (i != h.size()) err = ZK_E_GENERIC + 2001;
                }
            }
            if (redo) g_seg_stats[3]++;
        }
        if (err == ZK_OK && (!seg_mode || redo)) {
            uint64_t pos = 0;
            uint32_t rep[3] = {1, 4, 8};
            for (uint32_t bk = 0; bk < fi.n_blocks && err == ZK_OK; bk++) {
                const ZkBlock &b = fb[bk];
                if (b.status != ZK_OK) { err = b.status; break; }
                if (pos + b.out_size > d_size) { err = ZK_E_CORRUPTION; break; }
                if (b.out_size > block_max) { err = ZK_E_CORRUPTION; break; }
                uint8_t *bout = out + pos;
                if (b.type == 0) memcpy(bout, comp + b.src, b.bsize);
                else if (b.type == 1) memset(bout, comp[b.src], b.bsize);
                else {
                    err = exec_block(b, fi, out, pos, rep, nullptr);
                    if (err == ZK_OK) {
                        uint32_t r0 = zk_rep_resolve(b.rep_out[0], rep), r1 = zk_rep_resolve(b.rep_out[1], rep), r2 = zk_rep_resolve(b.rep_out[2], rep);
                        rep[0] = r0; rep[1] = r1; rep[2] = r2;
                    }
                }
                pos += b.out_size;
            }
            if (err == ZK_OK && pos != d_size) err = ZK_E_CORRUPTION;
        }
        if (status) status[f] = (int32_t)err;
        if (err && !first_err) first_err = -(int)err;
    }
    return first_err;
}
