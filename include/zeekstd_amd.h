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
 *   - there is no CPU fallback: if no gfx950 device is usable, zk_engine_create fails.
 */
#ifndef ZEEKSTD_AMD_H
#define ZEEKSTD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_ABI_VERSION 1

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

/* XXH64(seed 0) of count byte ranges data[off[i], off[i+1]) -> out[i].  (The checksum libzstd
 * computes when ZSTD_c_checksumFlag is set: encode.rs:163-167, 283-284.) */
int zk_xxh64_frames(zk_engine *e, const uint8_t *data, const uint64_t *off, uint32_t count, uint64_t *out);
int zk_xxh64_frames_dev(zk_engine *e, const void *d_data, const void *d_off, uint32_t count, void *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ZEEKSTD_AMD_H */
