// zk_sim.cpp -- TEST HARNESS (never shipped, never linked into libzeekstd_amd.so).
// Runs the per-lane device code of zeekstd_amd/csrc/zk_device.h on the CPU, lane after lane,
// with the same orchestration as the kernels in zk_decode.hip (walk -> scan -> walk -> huf ->
// fse -> exec in chunks/tiles).  Lets the CPU test suite check the kernel logic against the
// oracle without a GPU.  Tile writes are committed only after every "lane" of the tile ran, so
// an illegal in-tile history read would surface as a mismatch.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../zeekstd_amd/csrc/zk_device.h"

static const uint32_t LLV[36] = ZK_LL_TABLE;
static const uint32_t MLV[53] = ZK_ML_TABLE;

extern "C" int zk_sim_decode(const uint8_t *comp, const uint64_t *c_off, const uint64_t *d_off, uint32_t first,
                             uint32_t count, uint8_t *dst, int32_t *status, int exec_b, int exec_chunk)
{
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
    std::vector<ZkSeq> seqs(ns + 1);
    std::vector<uint8_t> lit(nl + 64);
    for (uint32_t f = 0; f < count; f++) {
        if (infos[f].status != ZK_OK) continue;
        ZkFrameInfo fi;
        zk_walk_frame(comp, c_off[first + f], c_off[first + f + 1], d_off[first + f + 1] - d_off[first + f], f, &bases[f], blocks.data(), fi);
    }
    // huf: one "lane" per stream
    std::vector<uint16_t> tab(2048);
    ZkHufScratch sc;
    for (uint64_t bi = 0; bi < nb; bi++) {
        ZkBlock &b = blocks[bi];
        if (!(b.type == 2 && b.lit_type >= 2 && b.status == ZK_OK)) continue;
        const ZkBlock &def = blocks[b.huf_def];
        uint32_t mb = 0;
        uint32_t r = zk_huf_build(comp + def.src + def.lit_off, def.lit_comp, tab.data(), &sc, &mb);
        bool ok = r != 0;
        if (ok) {
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
    for (uint64_t bi = 0; bi < nb; bi++) {
        ZkBlock b = blocks[bi];
        if (b.type != 2 || b.nseq == 0 || b.status != ZK_OK) continue;
        zk_decode_sequences(comp, blocks.data(), b, T, seqs.data() + b.seq_base, LLV, MLV);
        blocks[bi].out_size = b.out_size;
        for (int k = 0; k < 3; k++) blocks[bi].rep_out[k] = b.rep_out[k];
        blocks[bi].status = b.status;
    }
    delete T;
    // exec: one "workgroup" per frame
    const uint32_t THREADS = 256, B = (uint32_t)exec_b, CH = (uint32_t)exec_chunk;
    std::vector<uint32_t> oe(CH + 1), mlv(CH + 1), ofv(CH + 1), le(CH + 1);
    std::vector<uint8_t> tile(THREADS * B);
    int first_err = 0;
    for (uint32_t f = 0; f < count; f++) {
        const ZkFrameInfo fi = infos[f];
        uint32_t err = fi.status;
        if (err == ZK_OK) {
            const uint64_t d_size = d_off[first + f + 1] - d_off[first + f];
            uint8_t *out = dst + (d_off[first + f] - d_off[first]);
            const ZkBlock *fb = blocks.data() + bases[f].block_base;
            uint64_t pos = 0;
            uint32_t rep[3] = {1, 4, 8};
            for (uint32_t bk = 0; bk < fi.n_blocks && err == ZK_OK; bk++) {
                const ZkBlock &b = fb[bk];
                if (b.status != ZK_OK) { err = b.status; break; }
                if (pos + b.out_size > d_size) { err = ZK_E_CORRUPTION; break; }
                uint8_t *bout = out + pos;
                if (b.type == 0) memcpy(bout, comp + b.src, b.bsize);
                else if (b.type == 1) memset(bout, comp[b.src], b.bsize);
                else {
                    const ZkSeq *sq = seqs.data() + b.seq_base;
                    const uint8_t *l = b.lit_type >= 2 ? lit.data() + b.lit_base : comp + b.src + b.lit_off;
                    const uint32_t lit_stride = b.lit_type == 1 ? 0u : 1u;
                    uint32_t s0 = 0, cpos = 0;
                    for (;;) {
                        const uint32_t nsq = b.nseq - s0 < CH ? b.nseq - s0 : CH;
                        const bool lastchunk = s0 + nsq == b.nseq;
                        int bad = 0;
                        for (uint32_t i = 0; i < nsq; i++) {
                            ZkSeq s = sq[s0 + i];
                            uint32_t off = zk_rep_resolve(s.off, rep);
                            oe[i] = s.out_end; mlv[i] = s.ml; ofv[i] = off; le[i] = s.lit_end;
                            uint32_t mstart = s.out_end - s.ml;
                            if (off == 0 || pos + mstart < off || off > fi.window) bad = 1;
                        }
                        if (lastchunk) { oe[nsq] = b.out_size; mlv[nsq] = 0; ofv[nsq] = 0; le[nsq] = b.lit_regen; }
                        if (bad) { err = ZK_E_CORRUPTION; break; }
                        const uint32_t nent = nsq + (lastchunk ? 1u : 0u);
                        const uint32_t cend = lastchunk ? b.out_size : oe[nsq - 1];
                        for (uint32_t ts = cpos; ts < cend; ts += THREADS * B) {
                            for (uint32_t tid = 0; tid < THREADS; tid++) {
                                const uint32_t q0 = ts + tid * B;
                                if (q0 >= cend) break;
                                const uint32_t n = cend - q0 < B ? cend - q0 : B;
                                uint32_t j = zk_seq_find(oe.data(), 0, nent, q0);
                                for (uint32_t k = 0; k < n; k++) {
                                    const uint32_t q = q0 + k;
                                    while (oe[j] <= q) j++;
                                    uint64_t src = zk_resolve_byte(oe.data(), mlv.data(), ofv.data(), le.data(), j, q, (int32_t)ts);
                                    if (src & ZK_SRC_HIST) tile[tid * B + k] = bout[(int64_t)(int32_t)((uint32_t)src - 0x40000000u)];
                                    else tile[tid * B + k] = l[(uint32_t)src * lit_stride];
                                }
                            }
                            uint32_t tn = cend - ts < THREADS * B ? cend - ts : THREADS * B;
                            memcpy(bout + ts, tile.data(), tn);        // commit the tile after all lanes ran
                        }
                        cpos = cend; s0 += nsq;
                        if (lastchunk) break;
                    }
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
