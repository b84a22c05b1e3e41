/*
 * zeekstd_amd.h -- C ABI of the MI355X-native seekable-zstd engine (libzeekstd_amd.so).
 *
 * This is the FFI a Rust/C/C++/Python host binds instead of the libzstd symbols that
 * zeekstd reaches through zstd-safe (reference boundary: SURVEY.md 8b; call sites
 * lib/src/encode.rs:281-284,336,341-345,444-448,504-506,599 and
 * lib/src/decode.rs:181,184,213,243-245,250-253,354-356).  libzstd's streaming ABI moves
 * <= 128 KiB per call through one context; a GPU wants whole frames, many at a time, so the
 * boundary is re-cut one level up: "N frames in, N frames out" (Level A below), and the
 * zeekstd types (EncodeOptions / RawEncoder / Encoder / DecodeOptions / Decoder / SeekTable /
 * Serializer / Seekable / BytesWrapper) are restated on top of it (Level B, zk_* handle API
 * further down and the C++ classes in zeekstd_amd/csrc/host/zeekstd.hpp).
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; all functions are thread-compatible:
 *     one handle may be used by one thread at a time (libzstd CCtx/DCtx rule).
 *   - return value: 0 on success, negative on error.  -(ZSTD_ErrorCode) for codec errors
 *     (e.g. -20 corruption_detected, -22 checksum_wrong, -10 prefix_unknown, -70
 *     dstSize_tooSmall, -72 srcSize_wrong) exactly the codes zeekstd wraps in Kind::Zstd
 *     (lib/src/error.rs:40-45); zeekstd's own kinds map to the ZK_ERR_* values below.
 *   - "_dev" entry points take DEVICE pointers (HBM resident) and a hipStream_t passed as
 *     void*; the plain ones take HOST pointers and stage through the engine's buffers.
 *     The engine launches on its OWN queues (non-blocking streams; the stream argument, where
 *     there is one, replaces the main one): work the caller still has in flight on other
 *     streams for a buffer it passes -- the kernel that produces the input, the fill of a fresh
 *     output -- must have completed before the call, as for any two unordered HIP streams.
 *   - there is no CPU fallback: if no gfx950 device is usable, zk_engine_create fails.
 */
#ifndef ZEEKSTD_AMD_H
#define ZEEKSTD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_ABI_VERSION 2

/* zeekstd error kinds (lib/src/error.rs:101-113) that are not ZSTD_ErrorCode values */
#define ZK_ERR_OFFSET_OUT_OF_RANGE (-1001)
#define ZK_ERR_FRAME_INDEX_TOO_LARGE (-1002)
#define ZK_ERR_NUMBER_CONVERSION (-1003)
#define ZK_ERR_IO (-1004)
/* engine-level failures (no reference counterpart) */
#define ZK_ERR_HIP (-2001)       /* a HIP runtime call failed; see zk_engine_last_hip_error */
#define ZK_ERR_NO_DEVICE (-2002) /* no usable gfx950 device / kernels not loadable */
#define ZK_ERR_ARGUMENT (-2003)

/* format constants (lib/src/lib.rs:52-62) */
#define ZK_SEEKABLE_MAGIC_NUMBER 0x8F92EAB1u
#define ZK_SEEKABLE_MAX_FRAMES 0x08000000u
#define ZK_SEEK_TABLE_INTEGRITY_SIZE 9
#define ZK_SEEKABLE_MAX_FRAME_SIZE 0x40000000u
#define ZK_SKIPPABLE_HEADER_SIZE 8

typedef struct zk_engine zk_engine;

int zk_abi_version(void);
/* Human readable name of an error code returned by any function here
 * (ZSTD_getErrorName strings for codec errors: lib/src/error.rs:68). */
const char *zk_error_name(int code);

/* ---------------------------------------------------------------- Level A: batch engine */
/* Owns one GPU (device ordinal), its scratch buffers and a private stream.
 * Replaces CCtx::create/DCtx::create (encode.rs:130, decode.rs:31). */
int zk_engine_create(int device, zk_engine **out);
void zk_engine_destroy(zk_engine *e);
const char *zk_engine_last_hip_error(const zk_engine *e);
/* name of the device the engine runs on, e.g. "gfx950..." */
const char *zk_engine_device_name(const zk_engine *e);

/* Per-kernel timing for roofline reports: when on, every kernel launch of the next decode/encode call is
 * bracketed by HIP events on the launch stream; zk_engine_kernel_times returns the last call's
 * durations in ms, indexed like zk_engine_kernel_name (0 <= k < zk_engine_kernel_count()). */
int zk_engine_set_profiling(zk_engine *e, int on);
/* Kernel selection of the sequence decoder for blocks that carry their own FSE tables (archives written by libzstd):
 * 0 = by batch size (default), 1 = one lane per block (zk_k_fse), 2 = a quad of lanes per block (zk_k_fse_quad).
 * Results are identical; the tests use it to run both kernels on small inputs. */
int zk_engine_set_fse_kernel(zk_engine *e, int mode);                /* = zk_engine_set_kernel_choice(e, ZK_CHOICE_FSE_OWN, mode) */
/* Pins one kernel variant whatever the batch looks like (value 0 = by batch shape, the default and what production runs).
 * Every variant computes the same bytes: the point is that tests/ can put the kernels of the large-batch path (which a 4 GiB
 * batch selects) under small, exhaustively checked inputs, and that tools/ can time one variant against another.  Replaces the
 * environment switches of round 3.  ZK_ERR_ARGUMENT for an unknown key or value. */
enum {
    ZK_CHOICE_RESET = 0,          /* every key back to 0 (value ignored) */
    ZK_CHOICE_FSE_OWN = 1,        /* blocks with own FSE tables: 1 zk_k_fse (a lane per block), 2 zk_k_fse_quad in its 56-block layout */
    ZK_CHOICE_FSE_SHARED = 2,     /* blocks that share tables: 1 zk_k_fse_predef, 2 zk_k_fse_predef_fed, 3 zk_k_fse_sets (any of them
                                   * also switches the small-batch shortcut "every block a quad" off) */
    ZK_CHOICE_EXEC_LANES = 3,     /* zk_k_exec tile: 128 / 256 / 512 / 1024 lanes */
    ZK_CHOICE_EXEC_RING = 4,      /* 256-lane tiles: 1 = a ring of 2 T records, 2 = 4 T */
    ZK_CHOICE_XXH64 = 5,          /* 1 zk_k_xxh64 (a wave per frame), 2 zk_k_xxh64_wide (sixteen frames per wave), 3 zk_k_xxh64_lean (the same in 64
                                   * registers), 5 zk_k_xxh64_fed<4> (four frames per workgroup: sixteen chains in one wave, their products
                                   * brought by another; large batches take it by themselves) -- behind the executor;
                                   * 4 zk_k_xxh64_follow: BESIDE the executor, behind its progress words, on the decode's second queue
                                   * (zk_engine_checksums_followed says how many frames it verified; what it leaves is checked behind the
                                   * executor as with 5).  By batch shape: 4 for a synchronous decode of >= 512 frames */
    ZK_CHOICE_SMALL_PATH = 6,     /* host-pointer decode of <= 64 frames: 1 = through the general pipeline, 2 = the small path with its
                                   * entropy roles as two kernels */
    ZK_CHOICE_PIPE_CONTEXTS = 7,  /* host pipeline: decode contexts it rotates through (1..6; 0 = 2) */
    ZK_CHOICE_PIPE_CHUNK_MIB = 8, /* host pipeline: output MiB per chunk (0 = by total size) */
    ZK_CHOICE_EXEC_RESIDENT = 9,  /* zk_k_exec with 256-lane tiles: at most this many workgroups per CU (4 or 5; the launch asks for LDS it does not use) */
    ZK_CHOICE_EXEC_SEG = 10,      /* the executor in segments, several workgroups per frame (zk_k_seg_prep / zk_k_exec_seg / zk_k_exec_fill): 1 never, 2 always
                                     (decodes without a prefix); 0 = by batch shape */
    ZK_CHOICE_SEG_KIB = 11,       /* ... output KiB per segment (1..128; 0 = 128) */
    ZK_CHOICE_SEG_FILL = 12       /* ... its fill pass: 1 zk_k_exec_fill<1024> (rounds through memory), 2 zk_k_exec_fill<256>, 3 zk_k_exec_fill_lds (holes in LDS); 0 by batch size */
};
int zk_engine_set_kernel_choice(zk_engine *e, int what, int value);
/* Frames of the last finished device-pointer decode (zk_decode_frames_dev and its siblings, zk_decode_wait) whose Content_Checksum was
 * verified by zk_k_xxh64_follow, beside the executor; the others were verified behind it.  A diagnostic: results do not depend on it. */
uint64_t zk_engine_checksums_followed(const zk_engine *e);
int zk_engine_kernel_count(void);
const char *zk_engine_kernel_name(int k);
int zk_engine_kernel_times(const zk_engine *e, float *ms_out, int n);

/*
 * Decode frames [first, first+count) of a seekable payload.
 *   comp      compressed payload; frame i occupies comp[c_off[i], c_off[i+1])
 *   c_off     n+1 prefix sums of compressed sizes   (SeekTable entries, seek_table.rs:97-131)
 *   d_off     n+1 prefix sums of decompressed sizes
 *   dst       receives the decompressed bytes of the range, frame `first` at dst[0];
 *             dst_cap >= d_off[first+count] - d_off[first]
 *   verify    != 0: verify Content_Checksum (XXH64 low 32 bits) of frames that carry one
 *   frame_status  optional, count entries: 0 or the ZSTD_ErrorCode of that frame
 * Replaces the ZSTD_decompressStream loop of decode.rs:242-256 for whole frames.
 * Returns 0, or -(code) of the first failing frame.
 */
int zk_decode_frames(zk_engine *e, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off,
                     const uint64_t *d_off, uint32_t first, uint32_t count, uint8_t *dst, uint64_t dst_cap,
                     int verify, int32_t *frame_status);
/* Same with every buffer resident in HBM (c_off/d_off/frame_status are device pointers too).
 * stream: hipStream_t (NULL = the engine's own stream).  Synchronises the stream before returning. */
#define ZK_COMP_PADDING 8 /* device-resident compressed buffers must be readable this many bytes past comp_size */
int zk_decode_frames_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off,
                         const void *d_d_off, uint32_t first, uint32_t count, void *d_dst, uint64_t dst_cap,
                         int verify, void *d_frame_status, void *stream);

/* Random-access batch (many seeks per submission; engine-specific, no reference counterpart -- the reference serves one
 * seek at a time, lib/src/decode.rs:402-437): decode archive frames d_ids[0..count) (uint32, any order, repeats allowed) of a
 * device-resident archive; frame d_ids[i] lands at d_dst + d_out_off[i], d_out_off being the count + 1 prefix sums of the
 * selected frames' decompressed sizes (uint64).  d_c_off / d_d_off are the archive's full n+1 prefix arrays. */
/* Decode with a raw-content prefix (patch mode): what the reference does with ZSTD_DCtx_refPrefix before the first
 * frame and again after every frame end (lib/src/decode.rs:212-214, 248-255) -- every frame of the batch sees the
 * same prefix right before its first byte and a match may start up to prefix_len bytes before the frame.  With a
 * prefix an offset is bounded by availability only (libzstd's ZSTD_execSequence), not by the frame's window.
 * prefix_len may not exceed 2^30 - 2^27 bytes (-16 window_too_large otherwise); prefix_len == 0 is the plain call. */
int zk_decode_frames_prefix_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off,
                                const void *d_d_off, uint32_t first, uint32_t count, const void *d_prefix, uint64_t prefix_len,
                                void *d_dst, uint64_t dst_cap, int verify, void *d_frame_status, void *stream);
int zk_decode_frames_prefix(zk_engine *e, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off,
                            const uint64_t *d_off, uint32_t first, uint32_t count, const uint8_t *prefix, uint64_t prefix_len,
                            uint8_t *dst, uint64_t dst_cap, int verify, int32_t *frame_status);

int zk_decode_frame_list_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off, const void *d_d_off,
                             const void *d_ids, const void *d_out_off, uint32_t count, void *d_dst, uint64_t dst_cap,
                             int verify, void *d_frame_status, void *stream);

/* The decompressed sizes of frames whose seek entries the caller does not hold: header walk + sequence walks on the device, no output
 * (a frame that carries Frame_Content_Size is walked all the same and held to it: corruption_detected when its blocks make another
 * size).  What libzstd's streaming decoder needs no table for
 * (lib/src/decode.rs:243-245: ZSTD_decompressStream is handed bytes, not entries); the Level-C shim asks here before
 * it decodes.  sizes[count] (uint64), frame_status[count] (0 or -ZSTD_ErrorCode).  _dev: device pointers, c_off relative to d_comp. */
int zk_frame_content_sizes(zk_engine *e, const uint8_t *comp, uint64_t comp_size, const uint64_t *c_off, uint32_t first, uint32_t count,
                           uint64_t *sizes, int32_t *frame_status);
int zk_frame_content_sizes_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off, uint32_t first, uint32_t count,
                               void *d_sizes, void *d_frame_status, void *stream);

/* Two batches in flight.  zk_decode_submit_dev enqueues exactly what zk_decode_frames_dev runs, on one of the
 * engine's two decode contexts (own HIP queues and scratch), and returns while the kernels are still running;
 * *slot_out names the context.  zk_decode_wait(slot) blocks until that batch is complete and returns its status (0,
 * or the first failing frame's code like zk_decode_frames_dev); d_frame_status is valid after the wait.  At most
 * two batches may be outstanding (a third submit returns ZK_ERR_ARGUMENT until the older one was waited for) and
 * all buffers of a batch must stay untouched until its wait.  The gain: the last stage of a batch (per-frame XXH64,
 * a serial chain that leaves the GPU mostly idle) overlaps the first stages of the next.  No reference counterpart:
 * zeekstd's Decoder is synchronous (lib/src/decode.rs:201-270); this is the batch engine's pipelining. */
int zk_decode_submit_dev(zk_engine *e, const void *d_comp, uint64_t comp_size, const void *d_c_off,
                         const void *d_d_off, uint32_t first, uint32_t count, void *d_dst, uint64_t dst_cap,
                         int verify, void *d_frame_status, int *slot_out);
int zk_decode_wait(zk_engine *e, int slot);

/*
 * Encode src[0, n) as ceil(n / frame_size) independent zstd frames (FrameSizePolicy::Uncompressed(frame_size),
 * lib/src/encode.rs:21-39, 528-544; n == 0 yields one empty frame like Encoder::finish, encode.rs:755-757),
 * concatenated in dst, and report the seek-table entries: c_sizes[i] / d_sizes[i] are what
 * SeekTable::log_frame receives (seek_table.rs:513-525).  checksum != 0 appends the XXH64 Content_Checksum
 * (ZSTD_c_checksumFlag, encode.rs:283-284).  level is ZSTD_c_compressionLevel (encode.rs:281-282): one strategy
 * (hash matching in a 57 280-byte window -- from level 2 on plus sampled far matches anywhere behind it in the frame, the frame
 * then declares a window over its size --, Huffman literals, FSE tables measured per frame) with five settings
 * (zk_enc_device.h: zke_fast / zke_minmatch / zke_hash_log / zke_lazy / zke_step / zke_dense_in_frame / zke_dense_log) --
 * level <= 1 (negative levels included): table matches of 6+ bytes, 2^14 table entries, greedy parse; level 2: 5+ bytes,
 * 2^15 entries, lazy parse; levels 3..5 and 0 (= libzstd's default 3, the reference CLI's default): the same plus DENSE far
 * history (round 6: every position of the frame looked up in two tables per 256 KiB, 2^17 slots -- the short far matches
 * libzstd's level 3 finds in its 2 MiB window; 8 bytes of device scratch per input byte, at most 32 GiB); level 6..8: the same with the table
 * refreshed every 1024 instead of 4096 positions; level >= 9: 2^18 slots.  Ratio on the survey's text 2.485 / 2.655 / 2.732 /
 * 2.736 / 2.743 (libzstd 1.5.7: 2.50 at level 1, 2.79 at 3, 2.88 at 9), encode 100 / 47 / 24 / 20 / 20 GiB/s HBM to HBM.
 * Replaces the ZSTD_compressStream2 loops of encode.rs:340-346, 442-464.
 * dst_cap >= zk_compress_bound(n, frame_size) always suffices; otherwise -70 (dstSize_tooSmall) may come back.
 */
uint64_t zk_compress_bound(uint64_t n, uint32_t frame_size);
int zk_encode_frames(zk_engine *e, const uint8_t *src, uint64_t n, uint32_t frame_size, int level, int checksum,
                     uint8_t *dst, uint64_t dst_cap, uint32_t *c_sizes, uint32_t *d_sizes, uint32_t frames_cap,
                     uint32_t *n_frames, uint64_t *written);
/* Device-resident variant: d_src / d_dst / d_c_sizes / d_d_sizes (uint32 arrays, may be NULL) live in HBM. */
int zk_encode_frames_dev(zk_engine *e, const void *d_src, uint64_t n, uint32_t frame_size, int level, int checksum,
                         void *d_dst, uint64_t dst_cap, void *d_c_sizes, void *d_d_sizes, uint32_t *n_frames,
                         uint64_t *written, void *stream);

/* Encode against a raw-content prefix: ZSTD_CCtx_refPrefix at the start of every frame (lib/src/encode.rs:334-338).
 * The matcher's ring reaches the last 57280 bytes of the prefix; a longer prefix is also reached through a long-distance
 * table over its last 2^27 - 1 bytes (what the reference CLI asks of libzstd for --patch-from, cli/src/compress.rs:31-37:
 * long-distance matching + a window over the old file).  Frames made this way declare a window that covers prefix + frame
 * (128 KiB .. 128 MiB) and need the same prefix to decode (zk_decode_frames_prefix, or libzstd with ZSTD_DCtx_refPrefix).
 * prefix_len == 0 is the plain call. */
int zk_encode_frames_prefix_dev(zk_engine *e, const void *d_src, uint64_t n, uint32_t frame_size, int level, int checksum,
                                const void *d_prefix, uint64_t prefix_len, void *d_dst, uint64_t dst_cap, void *d_c_sizes,
                                void *d_d_sizes, uint32_t *n_frames_out, uint64_t *written_out, void *stream);
int zk_encode_frames_prefix(zk_engine *e, const uint8_t *src, uint64_t n, uint32_t frame_size, int level, int checksum,
                            const uint8_t *prefix, uint64_t prefix_len, uint8_t *dst, uint64_t dst_cap, uint32_t *c_sizes,
                            uint32_t *d_sizes, uint32_t frames_cap, uint32_t *n_frames_out, uint64_t *written_out);

/* Host memory the host-pointer entry points above move by DMA directly (pinned; hipHostMalloc underneath).  Buffers from
 * anywhere else work too: they are staged through the engine's own pinned rings by worker threads, chunk by chunk, beside
 * the kernels (the reference streams through 128 KiB buffers at no copy cost, lib/src/encode.rs:779-787,
 * decode.rs:222-225; on a GPU the PCIe legs are part of the path and are overlapped instead). */
void *zk_host_alloc(size_t bytes);
void zk_host_free(void *p);
/* worker threads of the host pipeline (parallel staging copies); 0 = default (hardware threads / 8, clamped to 2..16) */
int zk_engine_set_host_threads(zk_engine *e, int n);

/* Sharded encode, the one exchange step (BASELINE.json configs[4]): frames are independent, so rank r of `world`
 * encodes its contiguous frame range on its own GPU (zk_encode_frames_dev) and this call concatenates the ranks' compressed
 * streams in rank order into d_out on `root` and appends the seek table -- an archive zeekstd's Decoder reads
 * (seek_table.rs:379-436).  nccl_comm: the caller's RCCL communicator (ncclComm_t, one per rank); RCCL is resolved at run time.
 * Steps: all-gather of (bytes, frames), all-gather of the padded (c_size, d_size) entries, grouped ncclSend / ncclRecv of the
 * payloads straight into d_out + offset_r over xGMI, table serialised by the root.  d_payload / d_out: device memory;
 * c_sizes / d_sizes: this rank's n_frames entries on the host (what zk_encode_frames_dev reported).  On the root *out_bytes
 * receives stream + table bytes and *table_out (optional) the gathered SeekTable; the other ranks pass d_out = NULL. */
typedef struct zk_seek_table zk_seek_table;
/* The shared library that provides ncclAllGather / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd for zk_gather_seekable; NULL or ""
 * = RCCL (symbols already in the process, else librccl.so.1).  Process-wide; call it before the first gather.  (The library reads no
 * environment variable.)  The tests name a shared-memory transport between processes that share one GPU here. */
int zk_set_collective_library(const char *path);
int zk_gather_seekable(zk_engine *e, void *nccl_comm, int rank, int world, int root, const void *d_payload, uint64_t payload_bytes,
                       const uint32_t *c_sizes, const uint32_t *d_sizes, uint32_t n_frames, int format, void *d_out, uint64_t out_cap,
                       uint64_t *out_bytes, zk_seek_table **table_out, void *stream);

/* Sharded decode (the decode side): rank r of `world` owns the contiguous frame range zk_shard_range gives it,
 * reads / receives the compressed bytes [c_off[first], c_off[first + count]) of the archive -- nothing else -- and decodes
 * them into its own buffer; the output stays sharded, there is no collective.  zk_decode_shard does the arithmetic:
 * comp_shard points at exactly those bytes (frame `first` at comp_shard[0]), table is the archive's seek table
 * (zk_seek_table_from_bytes / _from_reader on the tail every rank can read: 8 n + 17 bytes), dst receives the frames'
 * decompressed bytes, *first / *count / *written report the range and its size.  Returns 0 or -(code) of the first
 * failing frame, like zk_decode_frames. */
int zk_shard_range(uint32_t n_frames, int rank, int world, uint32_t *first, uint32_t *count);
int zk_decode_shard(zk_engine *e, const uint8_t *comp_shard, uint64_t shard_bytes, const zk_seek_table *table, int rank, int world,
                    uint8_t *dst, uint64_t dst_cap, int verify, uint32_t *first, uint32_t *count, uint64_t *written);

/* XXH64(seed 0) of count byte ranges data[off[i], off[i+1]) -> out[i].  (The checksum libzstd
 * computes when ZSTD_c_checksumFlag is set: encode.rs:163-167, 283-284.) */
int zk_xxh64_frames(zk_engine *e, const uint8_t *data, const uint64_t *off, uint32_t count, uint64_t *out);
int zk_xxh64_frames_dev(zk_engine *e, const void *d_data, const void *d_off, uint32_t count, void *d_out, void *stream);

/* ================================================================ Level B: the zeekstd API as C handles
 * One-to-one with the crate's public items; C++ callers can use the classes in
 * zeekstd_amd/csrc/host/zeekstd.hpp directly.  Functions that can fail return 0 / a negative code and
 * deliver values through out-pointers; zk_last_error_message() holds the Display text of the last error
 * on this thread (lib/src/error.rs:60-71). */
const char *zk_last_error_message(void);

#define ZK_FORMAT_HEAD 0 /* seek_table::Format (lib/src/seek_table.rs:228-241) */
#define ZK_FORMAT_FOOT 1

/* ---- SeekTable (lib/src/seek_table.rs:243-935) */
typedef struct zk_serializer zk_serializer;
zk_seek_table *zk_seek_table_new(void);                                                        /* SeekTable::new :287 */
zk_seek_table *zk_seek_table_clone(const zk_seek_table *t);
void zk_seek_table_free(zk_seek_table *t);
/* from_seekable_format over a BytesWrapper (seek_table.rs:379-436, seekable.rs:55-97) */
int zk_seek_table_from_bytes(const uint8_t *src, size_t len, int format, zk_seek_table **out);
/* from_reader (Head format, :461-493); max_read > 0 makes the reader return at most that many bytes per read */
int zk_seek_table_from_reader_bytes(const uint8_t *p, size_t len, size_t max_read, zk_seek_table **out);
int zk_seek_table_log_frame(zk_seek_table *t, uint32_t c_size, uint32_t d_size);               /* :513-525 */
/* the same for n frames in one call (the entries zk_encode_frames* returns, or a whole gathered shard: one call instead of one per
 * frame from a host language).  Stops at the first frame that cannot be logged (:513-525's errors); the frames before it stay. */
int zk_seek_table_log_frames(zk_seek_table *t, uint32_t n, const uint32_t *c_sizes, const uint32_t *d_sizes);
uint32_t zk_seek_table_num_frames(const zk_seek_table *t);                                     /* :540 */
uint32_t zk_seek_table_frame_index_comp(const zk_seek_table *t, uint64_t offset);              /* :560 */
uint32_t zk_seek_table_frame_index_decomp(const zk_seek_table *t, uint64_t offset);            /* :579 */
int zk_seek_table_frame_start_comp(const zk_seek_table *t, uint32_t index, uint64_t *out);     /* :604 */
int zk_seek_table_frame_start_decomp(const zk_seek_table *t, uint32_t index, uint64_t *out);   /* :633 */
int zk_seek_table_frame_end_comp(const zk_seek_table *t, uint32_t index, uint64_t *out);       /* :662 */
int zk_seek_table_frame_end_decomp(const zk_seek_table *t, uint32_t index, uint64_t *out);     /* :691 */
int zk_seek_table_frame_size_comp(const zk_seek_table *t, uint32_t index, uint64_t *out);      /* :720 */
int zk_seek_table_frame_size_decomp(const zk_seek_table *t, uint32_t index, uint64_t *out);    /* :750 */
uint64_t zk_seek_table_max_frame_size_comp(const zk_seek_table *t);                            /* :774 */
uint64_t zk_seek_table_max_frame_size_decomp(const zk_seek_table *t);                          /* :799 */
uint64_t zk_seek_table_size_comp(const zk_seek_table *t);                                      /* :827 */
uint64_t zk_seek_table_size_decomp(const zk_seek_table *t);                                    /* :853 */
int zk_seek_table_equal(const zk_seek_table *a, const zk_seek_table *b);                       /* PartialEq :266 */
/* the n+1 prefix sums (what zk_decode_frames takes); returns n+1 */
size_t zk_seek_table_entries(const zk_seek_table *t, uint64_t *c_off, uint64_t *d_off, size_t cap);
/* ---- Serializer (seek_table.rs:937-1059): resumable at byte granularity, 0 == done */
zk_serializer *zk_seek_table_serializer(const zk_seek_table *t, int format);                   /* into_format_serializer :907 */
size_t zk_serializer_write_into(zk_serializer *s, uint8_t *buf, size_t len);                   /* :967-1005 */
void zk_serializer_reset(zk_serializer *s);                                                    /* :1034 */
size_t zk_serializer_encoded_len(const zk_serializer *s);                                      /* :1042 */
void zk_serializer_free(zk_serializer *s);

/* ---- DecodeOptions / Decoder (lib/src/decode.rs) */
typedef struct zk_decoder zk_decoder;
#define ZK_DEC_HAS_OFFSET 1u
#define ZK_DEC_HAS_OFFSET_LIMIT 2u
#define ZK_DEC_HAS_LOWER_FRAME 4u
#define ZK_DEC_HAS_UPPER_FRAME 8u
#define ZK_DEC_NO_VERIFY 16u
typedef struct zk_decode_opts {          /* DecodeOptions builder fields, decode.rs:13-114 */
    uint32_t flags;
    uint32_t lower_frame, upper_frame;   /* :73, :81 (upper is inclusive) */
    uint64_t offset, offset_limit;       /* :90, :99 */
    const zk_seek_table *seek_table;     /* :65; NULL = parse it from the source's tail */
    uint64_t batch_bytes;                /* engine-specific: decode-ahead per GPU submission (0 = default 64 MiB) */
} zk_decode_opts;
#define ZK_SEEK_START 0
#define ZK_SEEK_END 1
#define ZK_SEEK_CURRENT 2
/* e == NULL: the decoder creates (and owns) an engine on device 0, like DecodeOptions::new creating a DCtx (:30).
 * The byte source is borrowed and must outlive the decoder (BytesWrapper<'a>). */
int zk_decoder_open_bytes(zk_engine *e, const uint8_t *src, size_t len, const zk_decode_opts *o, zk_decoder **out);
int zk_decoder_open_file(zk_engine *e, const char *path, const zk_decode_opts *o, zk_decoder **out);   /* Read+Seek source, seekable.rs:112-138 */
/* Any `impl Seekable` of the host (lib/src/seekable.rs:16-39).  set_offset: whence 0 = OffsetFrom::Start(value), 1 =
 * OffsetFrom::End(value); returns the new position counted from the start, or a negative value on failure.  read: bytes
 * delivered (0 = end of source), or negative on failure.  Failures surface as ZK_ERR_IO.  The engine pulls compressed bytes
 * through `read` straight into its pinned staging buffers; calls come from the thread that calls the decoder.
 * seek_table_integrity is a REQUIRED method of the trait (seekable.rs:33-38; each impl supplies it, :84-96, :126-137):
 * zk_decoder_open_seekable takes it as a third callback -- format 0 = Format::Head, 1 = Format::Foot; it fills the 9-byte
 * integrity field and returns 0, or a negative value on failure -- so a source that keeps the field elsewhere plugs in.
 * zk_decoder_open_callbacks is the two-callback form: the field is then read the way both of the reference's own impls
 * read it (seek to SKIPPABLE_HEADER_SIZE, or to End(-9), and read 9 bytes). */
typedef int64_t (*zk_seek_fn)(void *user, int whence, int64_t value);
typedef int64_t (*zk_read_fn)(void *user, uint8_t *buf, size_t len);
typedef int (*zk_integrity_fn)(void *user, int format, uint8_t out[9]);
int zk_decoder_open_callbacks(zk_engine *e, zk_seek_fn set_offset, zk_read_fn read, void *user, const zk_decode_opts *o, zk_decoder **out);
int zk_decoder_open_seekable(zk_engine *e, zk_seek_fn set_offset, zk_read_fn read, zk_integrity_fn seek_table_integrity, void *user,
                             const zk_decode_opts *o, zk_decoder **out);
void zk_decoder_free(zk_decoder *d);
int64_t zk_decoder_decompress(zk_decoder *d, uint8_t *buf, size_t len);                        /* :314; bytes written or <0 */
int zk_decoder_decompress_with_prefix(zk_decoder *d, uint8_t *buf, size_t len, const uint8_t *prefix, size_t plen, size_t *out); /* :201 */
void zk_decoder_reset(zk_decoder *d);                                                          /* :346 */
int zk_decoder_set_lower_frame(zk_decoder *d, uint32_t index, uint64_t *out);                  /* :367 */
int zk_decoder_set_upper_frame(zk_decoder *d, uint32_t index, uint64_t *out);                  /* :383 */
int zk_decoder_set_offset(zk_decoder *d, uint64_t offset);                                     /* :402 */
int zk_decoder_set_offset_limit(zk_decoder *d, uint64_t limit);                                /* :432 */
uint64_t zk_decoder_read_compressed(const zk_decoder *d);                                      /* :448 */
uint64_t zk_decoder_offset(const zk_decoder *d);                                               /* :458 */
uint64_t zk_decoder_offset_limit(const zk_decoder *d);                                         /* :463 */
zk_seek_table *zk_decoder_seek_table(const zk_decoder *d);                                     /* :453 (a copy; free it) */
int zk_decoder_seek(zk_decoder *d, int whence, int64_t n, uint64_t *out);                      /* io::Seek :545-579 */
uint64_t zk_decoder_gpu_submissions(const zk_decoder *d);                                      /* engine-specific counter */
/* Measurement helper (BASELINE.json configs[3]): n seeks one at a time -- set_offset(offs[i]);
 * set_offset_limit(offs[i] + lens[i]); decompress to exhaustion (decode.rs:402-437, 201-270) -- each timed on its own with
 * the monotonic clock, microseconds into us_out[i].  expect (optional): the archive's uncompressed bytes; every read is
 * compared with expect + offs[i] outside the timed span.  Returns 0, a decoder error, or ZK_ERR_ARGUMENT on a mismatch: the
 * loop stops there, us_out[i] of that seek is negated and buf holds what it delivered. */
int zk_decoder_time_seeks(zk_decoder *d, const uint64_t *offs, const uint32_t *lens, uint32_t n, uint8_t *buf, size_t buf_len,
                          const uint8_t *expect, double *us_out);

/* ---- EncodeOptions / RawEncoder / Encoder (lib/src/encode.rs) */
typedef struct zk_raw_encoder zk_raw_encoder;
typedef struct zk_encoder zk_encoder;
#define ZK_POLICY_UNCOMPRESSED 0 /* FrameSizePolicy::Uncompressed(n), encode.rs:21-39 (default, n = 0x200000) */
#define ZK_POLICY_COMPRESSED 1   /* FrameSizePolicy::Compressed(n): frame closes once its encoding reaches n bytes (probed speculatively) */
typedef struct zk_encode_opts {  /* EncodeOptions builder fields, encode.rs:110-207 */
    uint32_t policy, frame_size; /* frame_size 0 = default policy */
    int32_t level;               /* compression_level :170 (default 0) */
    int32_t checksum;            /* checksum_flag :164 (default false) */
    uint32_t batch_frames;       /* engine-specific: frames per GPU submission gathered by Encoder (0 = 64) */
} zk_encode_opts;
/* e == NULL: the encoder creates (and owns) an engine on device 0, like EncodeOptions::new creating a CCtx (:129) */
int zk_raw_encoder_new(zk_engine *e, const zk_encode_opts *o, zk_raw_encoder **out);                 /* with_opts :280 */
void zk_raw_encoder_free(zk_raw_encoder *r);
int zk_raw_encoder_compress(zk_raw_encoder *r, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                            size_t *in_progress, size_t *out_progress);                                /* :398 */
int zk_raw_encoder_compress_with_prefix(zk_raw_encoder *r, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                                        const uint8_t *prefix, size_t plen, size_t *in_progress, size_t *out_progress); /* :311 */
int zk_raw_encoder_end_frame(zk_raw_encoder *r, uint8_t *out, size_t out_len, size_t *out_progress, size_t *data_left); /* :438 */
zk_seek_table *zk_raw_encoder_seek_table(const zk_raw_encoder *r);                                    /* :487 (a copy) */
void zk_raw_encoder_reset_frame(zk_raw_encoder *r);                                                   /* :501 */
void zk_raw_encoder_reset_seek_table(zk_raw_encoder *r);                                              /* :524 */
/* Encoder<W>: W is a write callback (return 0 on success) = io::Write::write_all */
typedef int (*zk_write_fn)(void *user, const uint8_t *data, size_t len);
int zk_encoder_new(zk_engine *e, const zk_encode_opts *o, zk_write_fn write, void *user, zk_encoder **out);   /* with_opts :596 */
/* A ready-made W for Encoder<W>: appends into caller memory like `Vec<u8>` does for the reference's benches
 * (lib/benches/compress.rs:47-51).  zk_buffer_writer_write is a zk_write_fn, `user` = the zk_buffer_writer.  With
 * `engine` set, large pieces are copied by the engine's worker threads.  A write past cap fails (-> ZK_ERR_IO). */
typedef struct zk_buffer_writer { uint8_t *data; uint64_t cap, len; zk_engine *engine; } zk_buffer_writer;
int zk_buffer_writer_write(void *user, const uint8_t *data, size_t len);
void zk_encoder_free(zk_encoder *e);
int64_t zk_encoder_compress(zk_encoder *e, const uint8_t *buf, size_t len);                           /* :692 / io::Write::write :791 */
int64_t zk_encoder_compress_with_prefix(zk_encoder *e, const uint8_t *buf, size_t len, const uint8_t *prefix, size_t plen); /* :641; the prefix buffer must stay valid and unchanged until the frames begun under it are out (end_frame / flush / finish) */
int64_t zk_encoder_end_frame(zk_encoder *e);                                                          /* :704 */
int zk_encoder_flush(zk_encoder *e);                                                                  /* io::Write::flush :796 */
int zk_encoder_finish(zk_encoder *e, int format, uint64_t *total);                                    /* finish_format :755 */
uint64_t zk_encoder_written_compressed(const zk_encoder *e);                                          /* :615 */
zk_seek_table *zk_encoder_seek_table(const zk_encoder *e);                                            /* :610 (a copy) */

#ifdef __cplusplus
}
#endif
#endif /* ZEEKSTD_AMD_H */
