// decoder.cpp -- zeekstd::Decoder on the batch engine.  Mirrors /root/reference/lib/src/decode.rs.
//
// The reference decodes 128 KiB at a time through ZSTD_decompressStream and throws away the bytes in
// front of `offset` ("dummy decompression", decode.rs:228-231).  Here the unit is the frame: the frames
// that cover the requested range are decoded by ONE engine submission into a host cache and reads are
// served from it.  Observable behaviour kept: lazy positioning at the frame of `offset`,
// nothing returned at or beyond offset_limit, offset advances by the bytes returned, a forward seek
// inside the cached frame does not touch the decode state (decode.rs:407-410, test :912-939), a frame
// cut short by offset_limit does not fail on its checksum (doc :425-427).
#include <exception>
#include <string.h>
#include <algorithm>
#include "../../../include/zeekstd_amd.h"
#include "zeekstd.hpp"
#include "../zk_engine.h"

namespace zeekstd {

Decoder DecodeOptions::into_decoder() { return Decoder(std::move(*this)); }

Decoder::Decoder(std::shared_ptr<Seekable> src) : Decoder(DecodeOptions(std::move(src))) {}

Decoder::Decoder(DecodeOptions &&opts)                                        // with_opts, decode.rs:152-187
{
    src_ = std::move(opts.src_);
    seek_table_ = opts.seek_table_ ? std::move(*opts.seek_table_) : SeekTable::from_seekable(*src_);
    uint64_t offset = opts.lower_frame_ ? seek_table_.frame_start_decomp(*opts.lower_frame_) : opts.offset_.value_or(0);
    check_offset(offset);
    uint64_t limit = opts.upper_frame_ ? seek_table_.frame_end_decomp(*opts.upper_frame_)
                                       : opts.offset_limit_.value_or(seek_table_.size_decomp());
    check_offset(limit);
    offset_ = offset; offset_limit_ = limit;
    batch_bytes_ = opts.batch_bytes_; verify_ = opts.verify_;
    if (opts.engine_) engine_ = opts.engine_;
    else {                                                                    // DCtx::create(), decode.rs:31
        int rc = zk_engine_create(0, &engine_);
        if (rc != 0) throw Error::from_engine_code(rc);
        owns_engine_ = true;
    }
}

Decoder::~Decoder()
{
    if (cache_) zk_host_free(cache_);
    if (owns_engine_ && engine_) zk_engine_destroy(engine_);
}

Decoder::Decoder(Decoder &&o) noexcept
    : engine_(o.engine_), owns_engine_(o.owns_engine_), seek_table_(std::move(o.seek_table_)), src_(std::move(o.src_)),
      offset_(o.offset_), offset_limit_(o.offset_limit_), read_compressed_(o.read_compressed_), batch_bytes_(o.batch_bytes_),
      verify_(o.verify_), cache_(o.cache_), cache_cap_(o.cache_cap_), cache_first_(o.cache_first_),
      cache_count_(o.cache_count_), cache_d_start_(o.cache_d_start_), cache_d_end_(o.cache_d_end_),
      cache_unverified_end_(o.cache_unverified_end_), submissions_(o.submissions_)
{
    cache_prefix_ = o.cache_prefix_; cache_prefix_len_ = o.cache_prefix_len_;
    o.engine_ = nullptr; o.owns_engine_ = false; o.cache_ = nullptr; o.cache_cap_ = 0;
}

void Decoder::check_offset(uint64_t offset) const                             // decode.rs:439-445
{
    if (offset > seek_table_.size_decomp()) throw Error::offset_out_of_range();
}

// The source as the engine's host pipeline sees it: contiguous memory when the Seekable is a byte slice, otherwise a pull
// callback that reads straight into the pipeline's pinned staging (set_offset + read, decode.rs:208-209, 222-225).
// A Seekable may throw (a file error, a host callback that fails): the exception must not unwind through the engine's
// pipeline, which has DMA, kernels and worker tasks in flight on its frame.  It is parked here, the pull reports "no bytes"
// so that the pipeline fails the chunk and drains in its normal way, and decode_range rethrows it afterwards.
struct SeekablePull { Seekable *src; std::exception_ptr error; };
static size_t seekable_pull(void *user, uint64_t off, uint8_t *dst, size_t n)
{
    SeekablePull *sp = (SeekablePull *)user;
    if (sp->error) return 0;
    try {
        sp->src->set_offset(OffsetFrom::Start(off));
        return sp->src->read(dst, n);
    } catch (...) { sp->error = std::current_exception(); return 0; }
}

// A seek table is untrusted input: entries that cannot describe a zstd frame are refused before anything is sized from
// them.  Frames are at most SEEKABLE_MAX_FRAME_SIZE (lib.rs:58) and no zstd frame expands more than 32768 : 1 (a 4-byte
// RLE block regenerates at most 128 KiB), so a tiny crafted table cannot force a multi-GiB allocation.
void Decoder::check_frames(uint32_t first, uint32_t count) const
{
    const auto &E = seek_table_.entries();
    for (uint32_t i = first; i < first + count; i++) {
        const uint64_t c = E[i + 1].c_offset - E[i].c_offset, d = E[i + 1].d_offset - E[i].d_offset;
        if (d > SEEKABLE_MAX_FRAME_SIZE) throw Error::zstd(14 /* frameParameter_unsupported */);
        if (d > c * 32768ull) throw Error::zstd(20 /* corruption_detected */);
    }
}

// Decode frames [first, first + count) into dst with one engine call (chunked and pipelined inside the engine).  Returns
// the number of leading frames that are good; `err` is the status of the first bad one (0 = all fine).  A frame cut short
// by offset_limit does not fail on its checksum (decode.rs:425-427): it counts as good and is remembered as unverified.
uint32_t Decoder::decode_range(uint32_t first, uint32_t count, uint8_t *dst, uint64_t dst_cap, const uint8_t *prefix, size_t prefix_len,
                               uint32_t *err)
{
    check_frames(first, count);
    const auto &E = seek_table_.entries();
    std::vector<uint64_t> c(count + 1), d(count + 1);
    for (uint32_t i = 0; i <= count; i++) { c[i] = E[first + i].c_offset; d[i] = E[first + i].d_offset; }
    zk_host_src hs;
    size_t mem_len = 0;
    const uint8_t *mem = src_->contiguous(&mem_len);
    if (mem) { if (c[count] > mem_len) throw Error::zstd(72 /* srcSize_wrong */); hs.mem = mem; }
    SeekablePull pull{src_.get(), nullptr};
    if (!mem) { hs.read = seekable_pull; hs.user = &pull; }
    const void *d_prefix = nullptr;
    int rc = zk_engine_stage_prefix(engine_, this, prefix, prefix ? prefix_len : 0, prefix_dirty_, &d_prefix);
    if (rc != 0) throw Error::from_engine_code(rc, zk_engine_last_hip_error(engine_));
    prefix_dirty_ = false;
    std::vector<int32_t> status(count);
    // a single frame that offset_limit cuts short is not verified at all: the checksum kernel is a serial chain per frame
    const bool cut_tail = E[first + count].d_offset > offset_limit_;
    const int verify = verify_ && !(count == 1 && cut_tail) ? 1 : 0;
    uint32_t n_ok = 0;
    rc = zk_host_decode(engine_, hs, c.data(), d.data(), 0, count, d_prefix, prefix ? prefix_len : 0, dst, dst_cap, verify, status.data(), &n_ok);
    submissions_++;
    if (pull.error) std::rethrow_exception(pull.error);                        // the source's own failure, after the pipeline has drained
    *err = 0;
    if (rc <= -1000) throw Error::from_engine_code(rc, zk_engine_last_hip_error(engine_));
    if (count == 1 && cut_tail && verify_ && n_ok == 1) unverified_end_tmp_ = E[first + 1].d_offset;     // not checked at all
    else unverified_end_tmp_ = 0;
    if (rc != 0) {
        for (uint32_t i = n_ok; i < count; i++) {
            if (status[i] == 0) { n_ok = i + 1; continue; }
            if (status[i] == 22 && i == count - 1 && cut_tail) { unverified_end_tmp_ = E[first + i + 1].d_offset; n_ok = i + 1; continue; }
            *err = (uint32_t)status[i];
            break;
        }
    }
    count_frames(first, first + n_ok);
    if (n_ok < count && *err == 0) *err = 1;
    return n_ok;
}

// Decode the frames that cover [offset_, want_end) into the (pinned) cache.  Frames up to request_end are what the caller
// asked for: a damaged one among them fails the call, as it does upstream (the `?` in decode.rs:242-245).  Frames beyond
// it are read-ahead: a damaged one there is left for the read that reaches it, the frames in front of it stay readable
// (the reference delivers every byte up to the corrupt frame, decode.rs:221-267).
void Decoder::fill_cache(uint64_t want_end, uint64_t request_end, const uint8_t *prefix, size_t prefix_len)
{
    const uint32_t first = seek_table_.frame_index_decomp(offset_);          // decode.rs:207
    uint32_t last = seek_table_.frame_index_decomp(want_end - 1);
    if (last < first) last = first;
    const auto &E = seek_table_.entries();
    const uint64_t d_lo = E[first].d_offset, d_hi = E[last + 1].d_offset;
    const uint32_t count = last - first + 1;
    check_frames(first, count);
    if ((size_t)(d_hi - d_lo) + 64 > cache_cap_) {
        if (cache_) zk_host_free(cache_);
        cache_ = nullptr; cache_cap_ = 0;
        const size_t want = (size_t)(d_hi - d_lo) + (size_t)(d_hi - d_lo) / 4 + 4096;
        cache_ = (uint8_t *)zk_host_alloc(want);         // pinned: the engine's D2H lands here directly
        if (!cache_) throw std::bad_alloc();
        cache_cap_ = want;
    }
    cache_count_ = 0; cache_d_start_ = cache_d_end_ = 0;
    uint32_t err = 0;
    const uint32_t n_ok = decode_range(first, count, cache_, d_hi - d_lo, prefix, prefix_len, &err);
    cache_prefix_ = prefix; cache_prefix_len_ = prefix ? prefix_len : 0;
    cache_first_ = first; cache_count_ = n_ok; cache_d_start_ = d_lo; cache_d_end_ = E[first + n_ok].d_offset;
    cache_unverified_end_ = unverified_end_tmp_;
    if (n_ok < count && E[first + n_ok].d_offset < request_end) throw Error::zstd(err);
}

size_t Decoder::decompress_with_prefix(uint8_t *buf, size_t len, const uint8_t *prefix, size_t prefix_len)
{
    if (!prefix) prefix_len = 0;
    // frames are decoded ahead of the reads; the ones in the cache were decoded with the prefix of the call that
    // filled it.  A different prefix (address or length: like libzstd, only the reference is kept) drops them.  The
    // reference applies a new prefix at the next frame start (decode.rs:248-255); here a switch in the middle of a
    // frame re-decodes that frame with the new one.
    if (prefix != cache_prefix_ || prefix_len != cache_prefix_len_) {
        cache_count_ = 0; cache_d_start_ = cache_d_end_ = 0;
        cache_prefix_ = prefix; cache_prefix_len_ = prefix_len;
        prefix_dirty_ = true;
    }
    const auto &E = seek_table_.entries();
    size_t progress = 0;
    while (offset_ < offset_limit_ && progress < len) {                      // decode.rs:221
        if (cache_count_ && offset_ >= cache_d_start_ && offset_ < cache_d_end_) {
            // a frame whose checksum went unchecked because offset_limit cut it short, and the limit has been raised past
            // its end since: the reference reaches the frame end now and verifies -- so the frame is decoded again
            if (cache_unverified_end_ && offset_limit_ >= cache_unverified_end_ && verify_) { cache_count_ = 0; continue; }
            size_t n = (size_t)std::min<uint64_t>({(uint64_t)(len - progress), offset_limit_ - offset_, cache_d_end_ - offset_});
            // frames kept across a seek count as read when they are delivered (upstream reads them again after its reset)
            count_frames(seek_table_.frame_index_decomp(offset_), seek_table_.frame_index_decomp(offset_ + n - 1) + 1);
            if (n >= (8u << 20)) zk_host_copy(engine_, buf + progress, cache_ + (offset_ - cache_d_start_), n);
            else memcpy(buf + progress, cache_ + (offset_ - cache_d_start_), n);
            offset_ += n; progress += n;                                      // decode.rs:263-266
            continue;
        }
        const uint64_t want = std::min<uint64_t>(offset_limit_, offset_ + (len - progress));
        const uint32_t first = seek_table_.frame_index_decomp(offset_);
        // whole frames inside the request go straight into the caller's buffer (no cache, no second copy): the engine
        // pipelines chunks of them through pinned staging, or moves them by DMA when buf is pinned (zk_host_alloc)
        if (offset_ == E[first].d_offset && first < seek_table_.num_frames() && E[first + 1].d_offset <= want &&
            want - offset_ >= batch_bytes_ / 8) {
            uint32_t last = seek_table_.frame_index_decomp(want - 1);
            if (E[last + 1].d_offset > want) last--;                          // the frame that `want` cuts is served through the cache
            const uint32_t count = last - first + 1;
            uint32_t err = 0;
            const uint32_t n_ok = decode_range(first, count, buf + progress, E[last + 1].d_offset - offset_, prefix, prefix_len, &err);
            if (n_ok < count) throw Error::zstd(err);                          // a damaged frame inside the request fails the call (decode.rs:242-245)
            const size_t n = (size_t)(E[first + n_ok].d_offset - offset_);
            offset_ += n; progress += n;
            continue;
        }
        uint64_t upto = want;
        // streaming reads (this call continues where the cache ends): decode ahead, a batch at a time
        // (a read at the very start of the archive with nothing decoded yet is how a streaming reader opens -- upstream's Decoder
        //  after new() / reset(), its bench protocol lib/benches/decompress.rs:27-39 --: it pays for one batch, not for a lone
        //  first frame and then a batch; a seek sets an offset first)
        const bool sequential = cache_count_ ? offset_ == cache_d_end_ : offset_ == 0;
        if (sequential || len - progress >= batch_bytes_)
            upto = std::min<uint64_t>(offset_limit_, std::max<uint64_t>(want, offset_ + batch_bytes_));
        fill_cache(upto, want, prefix, prefix_len);
    }
    return progress;
}

void Decoder::reset()                                                         // decode.rs:346-350
{
    reset_dctx();
    offset_ = 0;
    offset_limit_ = seek_table_.size_decomp();
}

void Decoder::reset_dctx(bool keep_cache)                                     // decode.rs:352-357
{
    read_compressed_ = 0;
    counted_lo_ = counted_hi_ = 0;
    // The frames decoded ahead stay valid whatever the decode state (the source is borrowed unchanged, decode.rs:201): a seek
    // into them is served from the cache instead of a new submission.  Decoder::reset() drops them like upstream's state.
    if (!keep_cache) { cache_count_ = 0; cache_d_start_ = cache_d_end_ = 0; cache_unverified_end_ = 0; }
    prefix_dirty_ = true;                       // a reset decoder references its prefix anew (decode.rs:212-214)
}

// read_compressed (decode.rs:448): compressed bytes of the frames [first, end) are added unless the range counted since the
// last reset already holds them
void Decoder::count_frames(uint32_t first, uint32_t end)
{
    const auto &E = seek_table_.entries();
    if (end > seek_table_.num_frames()) end = seek_table_.num_frames();
    for (uint32_t f = first; f < end; f++) {
        if (f >= counted_lo_ && f < counted_hi_) continue;
        read_compressed_ += E[f + 1].c_offset - E[f].c_offset;
        if (counted_lo_ == counted_hi_) { counted_lo_ = f; counted_hi_ = f + 1; }
        else if (f == counted_hi_) counted_hi_++;
        else if (f + 1 == counted_lo_) counted_lo_--;
        else { counted_lo_ = f; counted_hi_ = f + 1; }
    }
}

uint64_t Decoder::set_lower_frame(uint32_t index)                             // decode.rs:367-372
{
    uint64_t off = seek_table_.frame_start_decomp(index);
    set_offset(off);
    return off;
}

uint64_t Decoder::set_upper_frame(uint32_t index)                             // decode.rs:383-388
{
    uint64_t off = seek_table_.frame_end_decomp(index);
    set_offset_limit(off);
    return off;
}

void Decoder::set_offset(uint64_t offset)                                     // decode.rs:402-414
{
    check_offset(offset);
    const uint32_t current = seek_table_.frame_index_decomp(offset_);
    const uint32_t target = seek_table_.frame_index_decomp(offset);
    // only reset if we cannot continue from the previous decompression
    if (current != target || offset < offset_) reset_dctx(cache_count_ && offset >= cache_d_start_ && offset < cache_d_end_);
    offset_ = offset;
}

void Decoder::set_offset_limit(uint64_t limit)                                // decode.rs:432-437
{
    check_offset(limit);
    offset_limit_ = limit;
}

uint64_t Decoder::seek(SeekFrom from, int64_t n)                              // decode.rs:545-579
{
    uint64_t off;
    if (from == SeekFrom::Start) off = (uint64_t)n;
    else if (from == SeekFrom::End) {
        if (n > 0) throw Error::offset_out_of_range();
        uint64_t size = seek_table_.size_decomp();
        if ((uint64_t)(-n) > size) throw Error::offset_out_of_range();        // checked_add_signed
        off = size - (uint64_t)(-n);
    } else {
        if (n < 0) { if ((uint64_t)(-n) > offset_) throw Error::offset_out_of_range(); off = offset_ - (uint64_t)(-n); }
        else { if ((uint64_t)n > UINT64_MAX - offset_) throw Error::offset_out_of_range(); off = offset_ + (uint64_t)n; }
    }
    set_offset(off);
    return off;
}

}  // namespace zeekstd
