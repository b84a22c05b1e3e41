// zk_kernels.h -- host-callable launchers of the gfx950 kernels (zk_decode.hip, zk_encode.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "zk_device.h"

// Which kernel variant a launcher picks.  0 everywhere = by batch shape (what production runs); the other values pin one
// variant whatever the batch looks like, so that tests/ can put every kernel of the large-batch path under a small, exhaustively
// checked input (zk_engine_set_kernel_choice, include/zeekstd_amd.h) and tools/ can time one against the other.
struct ZkKernelChoice {
    int fse_own = 0;        // blocks with tables of their own: 1 zk_k_fse (a lane per block), 2 zk_k_fse_quad in the 56-block layout
    int fse_shared = 0;     // blocks that share tables: 1 zk_k_fse_predef, 2 zk_k_fse_predef_fed, 3 zk_k_fse_sets
    int exec_lanes = 0;     // zk_k_exec tile: 128 / 256 / 512 / 1024 lanes
    int exec_ring = 0;      // 256-lane tiles: 1 a ring of 2 T records, 2 of 4 T
    int xxh = 0;            // 1 zk_k_xxh64 (a wave per frame), 2 zk_k_xxh64_wide (sixteen frames per wave), 3 zk_k_xxh64_lean (the same in 64 registers),
                            // 4 zk_k_xxh64_follow beside the executor (zk_engine.hip)
    int exec_resident = 0;  // zk_k_exec<256>: workgroups per CU (4 / 5) through LDS the launch asks for and does not use
    int small_path = 0;     // host-pointer decode of <= 64 frames: 1 the general pipeline instead, 2 the small path's entropy roles as two kernels
    int exec_seg = 0;       // the executor in segments (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill): 1 never, 2 always (without a prefix); 0 by batch shape
    int seg_kib = 0;        // ... output KiB per segment (1..128; 0 = 128)
    int seg_fill = 0;       // ... the fill pass: 1 zk_k_exec_fill<1024> (rounds through memory), 2 zk_k_exec_fill<256>, 3 zk_k_exec_fill_lds (the segment's holes in LDS); 0 by batch size
};
// scratch of the segmented executor (zk_engine.hip sizes it; zk_device.h: zk_seg_region)
struct ZkSegScratch { ZkSeg *segs; uint32_t *nsegs, *segn; ZkHole *holes; uint32_t *tilecnt; uint32_t max_segs, seg_bytes; };

void zk_launch_walk(hipStream_t st, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off, const uint64_t *d_off, uint32_t first,
                    uint32_t count, const uint32_t *ids, const uint64_t *out_off, uint64_t dst_cap, const ZkFrameBase *bases, ZkBlock *blocks, ZkFrameInfo *infos);
void zk_launch_frame_sizes(hipStream_t st, const ZkFrameInfo *infos, const ZkFrameBase *bases, const ZkBlock *blocks, uint32_t count, uint64_t *sizes, int32_t *status_out);
void zk_launch_scan(hipStream_t st, const ZkFrameInfo *infos, uint32_t count, ZkFrameBase *bases, uint64_t *totals, const uint64_t *d_off, uint32_t first, const uint64_t *out_off);
void zk_launch_huf(hipStream_t st, const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, uint8_t *lit);
void zk_launch_fse(hipStream_t st, const uint8_t *comp, ZkBlock *blocks, uint32_t nblocks, uint32_t n_own_tables, ZkSeqP *seqs, const ZkKernelChoice &k, uint32_t frames = 0);
void zk_launch_exec(hipStream_t st, const uint8_t *comp, const uint64_t *d_off, uint32_t first, uint32_t count,
                    const uint32_t *ids, const uint64_t *out_off, const ZkBlock *blocks, const ZkFrameBase *bases, ZkFrameInfo *infos, const ZkSeqP *seqs,
                    const uint8_t *lit, uint8_t *dst, const uint8_t *prefix, uint64_t plen, const ZkKernelChoice &k, bool dense = false,
                    uint64_t *progress = nullptr);      // progress: one word per frame for zk_launch_xxh64_follow (zk_decode.hip: zk_publish)
void zk_launch_exec_seg(hipStream_t st, const uint8_t *comp, const uint64_t *d_off, uint32_t first, uint32_t count,
                        const uint32_t *ids, const uint64_t *out_off, const ZkBlock *blocks, const ZkFrameBase *bases, ZkFrameInfo *infos, const ZkSeqP *seqs,
                        const uint8_t *lit, uint8_t *dst, const ZkSegScratch &sg, const ZkKernelChoice &k, bool dense, uint64_t *progress = nullptr);
void zk_launch_xxh64(hipStream_t st, const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count,
                     ZkFrameInfo *infos, uint64_t *hashes, const ZkKernelChoice &k, const uint64_t *skip = nullptr,
                     uint32_t wide_from = 1024,          // frames from which several share a workgroup (the encoder passes 512: zk_decode.hip)
                     bool beside = false);               // the pass runs beside a kernel that pays for its instruction slots (the encoder's): sixteen frames per wave
// the checksums beside the executor that publishes `progress` (launched on another queue); frames it verifies are marked in `progress`,
// zk_launch_xxh64(..., skip = progress) behind the executor takes the rest
void zk_launch_xxh64_follow(hipStream_t st, const uint8_t *data, const uint64_t *d_off, uint32_t first, uint32_t count, const ZkFrameInfo *infos, uint64_t *progress);
void zk_launch_status(hipStream_t st, const ZkFrameInfo *infos, uint32_t count, int32_t *status_out, uint64_t *first_err,
                      const uint64_t *progress = nullptr, uint64_t *followed = nullptr);

// small batches (a seek): no host round trip, no copy commands -- see zk_decode.hip
void zk_launch_small_walk(hipStream_t st, const uint8_t *h_comp, uint64_t comp_bytes, const uint64_t *h_offs, uint32_t count, uint64_t dst_cap,
                          uint32_t block_cap, uint64_t seq_cap, uint8_t *d_comp, uint64_t *d_offs, ZkFrameInfo *infos, ZkFrameBase *bases, ZkBlock *blocks, uint64_t *words);
void zk_launch_small_entropy(hipStream_t st, const uint8_t *comp, ZkBlock *blocks, const uint64_t *words, uint8_t *lit, ZkSeqP *seqs, uint32_t groups, bool split);
void zk_launch_small_publish(hipStream_t st, const ZkFrameInfo *infos, const uint64_t *d_offs, uint32_t count, const uint8_t *dst, uint8_t *h_out,
                             int32_t *d_status, int32_t *h_status, uint64_t *words, uint32_t *h_flag, uint32_t gen);

// ---- encoder (zk_encode.hip)
#include "zk_enc_device.h"
void zk_launch_enc_stage_hist(hipStream_t st, const uint8_t *src, const uint8_t *prefix_tail, const ZkEncFrame *frames, uint32_t nframes, uint8_t *stage);
void zk_launch_enc_match(hipStream_t st, const uint8_t *src, const ZkEncFrame *segs, uint32_t nsegs, ZkEncBlock *blocks, uint64_t *seqs, uint8_t *lits, int level, const ZkEncLdm &ldm);
int zk_launch_enc_ldm_build(hipStream_t st, const ZkEncLdm &ldm, uint32_t *table);                        // 0, or -1: the table could not be cleared
// cand, part: an entry per input byte + ZKE_DENSE_SLACK; poff: (ZKE_DENSE_PASSES_MAX + 1) words per segment; every entry that is read is written first: no clear
void zk_launch_enc_dense_cand(hipStream_t st, const uint8_t *src, const ZkEncFrame *segs, uint32_t nsegs, const ZkEncLdm &ldm, uint32_t *cand, uint32_t *part, uint32_t *poff);
int zk_launch_enc_ldm_build_frames(hipStream_t st, const uint8_t *src, const ZkEncLdm &ldm, uint32_t *table, uint32_t nframes);
void zk_launch_enc_fse_build(hipStream_t st, const uint8_t *src, const ZkEncFrame *frames, uint32_t nframes, const ZkEncBlock *blocks, uint64_t *seqs, uint32_t *mpos,
                             const ZkEncTables *predef, ZkEncTables *ftab);
void zk_launch_enc_entropy(hipStream_t st, const uint8_t *src, const ZkEncFrame *frames, ZkEncBlock *blocks, uint32_t nblocks,
                           uint64_t *seqs, uint32_t *mpos, const uint8_t *lits, uint8_t *scratch, const ZkEncTables *ftab);
void zk_launch_enc_sizes(hipStream_t st, const ZkEncFrame *frames, uint32_t nframes, ZkEncBlock *blocks, const ZkEncTables *ftab, int checksum,
                         uint64_t *c_size64, uint32_t *c_sizes, uint32_t *d_sizes);
void zk_launch_scan64(hipStream_t st, const uint64_t *in, uint32_t n, uint64_t *out);
void zk_launch_enc_assemble(hipStream_t st, const uint8_t *src, const ZkEncFrame *frames, uint32_t nframes, const ZkEncBlock *blocks, uint32_t nblocks, const ZkEncTables *ftab,
                            const uint8_t *lits, const uint8_t *scratch, const uint64_t *out_off, const uint64_t *c_size64, const uint64_t *hashes, int checksum, uint8_t *dst);
