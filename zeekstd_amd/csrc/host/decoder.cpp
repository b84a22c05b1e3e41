// decoder.cpp -- zeekstd::Decoder on the batch engine.  Mirrors /root/reference/lib/src/decode.rs.
//
// The reference decodes 128 KiB at a time through ZSTD_decompressStream and throws away the bytes in
// front of `offset` ("dummy decompression", decode.rs:228-231).  Here the unit is the frame: the frames
// that cover the requested range are decoded by ONE engine submission into a host cache and reads are
// served from it.  Observable behaviour kept: lazy positioning at the frame of `offset`,
// nothing returned at or beyond offset_limit, offset advances by the bytes returned, a forward seek
// inside the cached frame does not touch the decode state (decode.rs:407-410, test :912-939), a frame
// cut short by offset_limit does not fail on its checksum (doc :425-427).
#include <string.h>
#include <algorithm>
#include "../../../include/zeekstd_amd.h"
#include "zeekstd.hpp"

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

Decoder::~Decoder() { if (owns_engine_ && engine_) zk_engine_destroy(engine_); }

Decoder::Decoder(Decoder &&o) noexcept
    : engine_(o.engine_), owns_engine_(o.owns_engine_), seek_table_(std::move(o.seek_table_)), src_(std::move(o.src_)),
      offset_(o.offset_), offset_limit_(o.offset_limit_), read_compressed_(o.read_compressed_), batch_bytes_(o.batch_bytes_),
      verify_(o.verify_), cache_(std::move(o.cache_)), comp_buf_(std::move(o.comp_buf_)), cache_first_(o.cache_first_),
      cache_count_(o.cache_count_), cache_d_start_(o.cache_d_start_), cache_d_end_(o.cache_d_end_), last_end_(o.last_end_),
      submissions_(o.submissions_)
{
    cache_prefix_ = o.cache_prefix_; cache_prefix_len_ = o.cache_prefix_len_;
    o.engine_ = nullptr; o.owns_engine_ = false;
}

void Decoder::check_offset(uint64_t offset) const                             // decode.rs:439-445
{
    if (offset > seek_table_.size_decomp()) throw Error::offset_out_of_range();
}

// Decode the frames that cover [offset_, want_end) into the host cache with one engine submission.
void Decoder::fill_cache(uint64_t want_end, const uint8_t *prefix, size_t prefix_len)
{
    const uint32_t first = seek_table_.frame_index_decomp(offset_);          // decode.rs:207
    uint32_t last = seek_table_.frame_index_decomp(want_end - 1);
    if (last < first) last = first;
    const auto &E = seek_table_.entries();
    const uint64_t c_lo = E[first].c_offset, c_hi = E[last + 1].c_offset;
    const uint64_t d_lo = E[first].d_offset, d_hi = E[last + 1].d_offset;
    const uint32_t count = last - first + 1;
    comp_buf_.resize((size_t)(c_hi - c_lo) + ZK_COMP_PADDING);
    src_->set_offset(OffsetFrom::Start(c_lo));                                // decode.rs:208-209
    size_t got = 0;
    while (got < c_hi - c_lo) {                                               // refill loop, decode.rs:222-225
        size_t n = src_->read(comp_buf_.data() + got, (size_t)(c_hi - c_lo) - got);
        if (n == 0) throw Error::zstd(72 /* srcSize_wrong: the source ends inside a frame */);
        got += n;
    }
    std::vector<uint64_t> c(count + 1), d(count + 1);
    for (uint32_t i = 0; i <= count; i++) { c[i] = E[first + i].c_offset - c_lo; d[i] = E[first + i].d_offset - d_lo; }
    cache_.resize((size_t)(d_hi - d_lo) + 1);
    std::vector<int32_t> status(count);
    // every frame of the submission sees the prefix right before its first byte: ref_prefix before the first frame
    // and again after each frame end, decode.rs:212-214, 248-255
    int rc = zk_decode_frames_prefix(engine_, comp_buf_.data(), c_hi - c_lo, c.data(), d.data(), 0, count, prefix, prefix ? prefix_len : 0,
                                     cache_.data(), d_hi - d_lo, verify_ ? 1 : 0, status.data());
    cache_prefix_ = prefix; cache_prefix_len_ = prefix ? prefix_len : 0;
    submissions_++;
    if (rc != 0) {
        if (rc <= -1000) throw Error::from_engine_code(rc, zk_engine_last_hip_error(engine_));
        for (uint32_t i = 0; i < count; i++) {
            if (status[i] == 0) continue;
            // the reference never verifies the checksum of a frame that offset_limit cuts short (decode.rs:425-427)
            const bool cut = E[first + i + 1].d_offset > offset_limit_;
            if (status[i] == 22 && cut) continue;
            throw Error::zstd((uint32_t)status[i]);
        }
    }
    cache_first_ = first; cache_count_ = count; cache_d_start_ = d_lo; cache_d_end_ = d_hi;
    read_compressed_ += c_hi - c_lo;
}

size_t Decoder::decompress_with_prefix(uint8_t *buf, size_t len, const uint8_t *prefix, size_t prefix_len)
{
    if (!prefix) prefix_len = 0;
    if (read_compressed_ == 0) { cache_count_ = 0; cache_d_start_ = cache_d_end_ = 0; }   // fresh decode state, decode.rs:206-218
    // frames are decoded ahead of the reads; the ones in the cache were decoded with the prefix of the call that
    // filled it.  A different prefix (address or length: like libzstd, only the reference is kept) drops them.  The
    // reference applies a new prefix at the next frame start (decode.rs:248-255); here a switch in the middle of a
    // frame re-decodes that frame with the new one.
    if (cache_count_ && (prefix != cache_prefix_ || prefix_len != cache_prefix_len_)) { cache_count_ = 0; cache_d_start_ = cache_d_end_ = 0; }
    size_t progress = 0;
    while (offset_ < offset_limit_ && progress < len) {                      // decode.rs:221
        if (!(cache_count_ && offset_ >= cache_d_start_ && offset_ < cache_d_end_)) {
            uint64_t want = std::min<uint64_t>(offset_limit_, offset_ + (len - progress));
            // streaming reads (this call continues where the cache ends): decode ahead, a batch at a time
            const bool sequential = cache_count_ && offset_ == cache_d_end_;
            if (sequential || len - progress >= batch_bytes_)
                want = std::min<uint64_t>(offset_limit_, std::max<uint64_t>(want, offset_ + batch_bytes_));
            fill_cache(want, prefix, prefix_len);
        }
        size_t n = (size_t)std::min<uint64_t>({(uint64_t)(len - progress), offset_limit_ - offset_, cache_d_end_ - offset_});
        memcpy(buf + progress, cache_.data() + (offset_ - cache_d_start_), n);
        offset_ += n; progress += n;                                          // decode.rs:263-266
    }
    return progress;
}

void Decoder::reset()                                                         // decode.rs:346-350
{
    reset_dctx();
    offset_ = 0;
    offset_limit_ = seek_table_.size_decomp();
}

void Decoder::reset_dctx()                                                    // decode.rs:352-357
{
    read_compressed_ = 0;
    cache_count_ = 0; cache_d_start_ = cache_d_end_ = 0;
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
    if (current != target || offset < offset_) reset_dctx();
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
