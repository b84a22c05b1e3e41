// encoder.cpp -- zeekstd::RawEncoder / Encoder on the batch engine.  Mirrors /root/reference/lib/src/encode.rs.
//
// libzstd streams: every compress() call may emit output.  Here a frame is encoded by ONE engine
// submission once it is complete, so RawEncoder::compress() only accepts input (out_progress = 0) and the
// compressed bytes of a frame appear through end_frame() -- or through the compress() call that finds the
// frame complete, exactly the call that runs the end_frame loop upstream (encode.rs:317-327).  Frame
// boundaries, in/out accounting, the seek table and the "close OR compress, never both" rule are kept.
// Encoder<W> additionally gathers `batch_frames` complete frames per submission (EncodeOptions::batch_frames).
#include <string.h>
#include <algorithm>
#include "../../../include/zeekstd_amd.h"
#include "zeekstd.hpp"
#include "../zk_engine.h"

namespace zeekstd {

static const uint32_t MAX_FRAME_SIZE = (uint32_t)SEEKABLE_MAX_FRAME_SIZE;      // encode.rs:14

RawEncoder EncodeOptions::into_raw_encoder() { return RawEncoder(std::move(*this)); }
Encoder EncodeOptions::into_encoder(std::shared_ptr<Writer> writer) { return Encoder(std::move(writer), std::move(*this)); }

RawEncoder::RawEncoder() : RawEncoder(EncodeOptions()) {}

RawEncoder::RawEncoder(EncodeOptions &&opts)                                   // with_opts, encode.rs:280-293
    : policy_(opts.policy_), checksum_(opts.checksum_), level_(opts.level_)
{
    if (opts.engine_) engine_ = opts.engine_;
    else {                                                                     // CCtx::create(), encode.rs:130
        int rc = zk_engine_create(0, &engine_);
        if (rc != 0) throw Error::from_engine_code(rc);
        owns_engine_ = true;
    }
}

RawEncoder::~RawEncoder() { if (owns_engine_ && engine_) zk_engine_destroy(engine_); }

RawEncoder::RawEncoder(RawEncoder &&o) noexcept
    : engine_(o.engine_), owns_engine_(o.owns_engine_), policy_(o.policy_), checksum_(o.checksum_), level_(o.level_),
      frame_c_size_(o.frame_c_size_), frame_d_size_(o.frame_d_size_), seek_table_(std::move(o.seek_table_)),
      frame_in_(std::move(o.frame_in_)), pending_(std::move(o.pending_)), pending_pos_(o.pending_pos_), encoded_(o.encoded_),
      next_probe_(o.next_probe_), ratio_(o.ratio_), coarse_(o.coarse_), est_c_(o.est_c_), piece_start_(o.piece_start_),
      frame_prefix_(o.frame_prefix_), frame_prefix_len_(o.frame_prefix_len_)
{
    o.engine_ = nullptr; o.owns_engine_ = false;
}

size_t RawEncoder::remaining_frame_size() const                                // encode.rs:528-535
{
    if (policy_.kind == FrameSizePolicy::Kind::Compressed) return MAX_FRAME_SIZE - frame_d_size_;
    return std::min(MAX_FRAME_SIZE, policy_.size) - frame_d_size_;
}

// Compressed(n): upstream compares n with the bytes libzstd has emitted so far for the frame, after every call, and a call
// emits at most the caller's output buffer (encode.rs:341-347): its frames end with n <= c_size < n + out.len() (131 591
// for Encoder<W>).  A frame is encoded whole here, so (1) a call takes no more input than the caller's output buffer could
// hold, capped at 131 591 bytes -- a larger step could not overshoot less than upstream, and one huge write must not become
// one huge frame -- and (2) the compressed size of the frame-so-far is obtained by encoding it speculatively: first where
// the last frame's ratio says n will be reached, then as far ahead as the measured ratio says is still short of n (steps of
// at least 32 KiB, never more than the missing bytes plus the window).  The encoding is kept when it reaches n, otherwise
// dropped.  Bulk data does not come this way: Encoder cuts and encodes whole batches of frames (speculate_compressed).
bool RawEncoder::is_frame_complete() const                                     // encode.rs:537-544
{
    if (policy_.kind == FrameSizePolicy::Kind::Compressed)
        return (encoded_ && policy_.size <= pending_.size()) || MAX_FRAME_SIZE <= frame_d_size_;
    return std::min(MAX_FRAME_SIZE, policy_.size) <= frame_d_size_;
}

void RawEncoder::encode_pending()
{
    if (encoded_) return;
    const size_t n = frame_in_.size();
    pending_.resize((size_t)zk_compress_bound(n, n ? (uint32_t)n : 1));
    uint32_t c = 0, d = 0, nf = 0;
    uint64_t written = 0;
    int rc = zk_encode_frames_prefix(engine_, frame_in_.data(), n, n ? (uint32_t)n : 1, level_, checksum_ ? 1 : 0, frame_prefix_,
                                     frame_prefix_ ? frame_prefix_len_ : 0, pending_.data(), pending_.size(), &c, &d, 1, &nf, &written);
    if (rc != 0) throw Error::from_engine_code(rc, zk_engine_last_hip_error(engine_));
    pending_.resize((size_t)written);
    pending_pos_ = 0;
    encoded_ = true;
}

size_t RawEncoder::encode_piece(size_t lo, size_t hi)
{
    const size_t n = hi - lo;
    if (!n) return 0;
    piece_out_.resize((size_t)zk_compress_bound(n, (uint32_t)n));
    uint32_t c = 0, d = 0, nf = 0;
    uint64_t written = 0;
    int rc = zk_encode_frames_prefix(engine_, frame_in_.data() + lo, n, (uint32_t)n, level_, checksum_ ? 1 : 0, nullptr, 0, piece_out_.data(),
                                     piece_out_.size(), &c, &d, 1, &nf, &written);
    if (rc != 0) throw Error::from_engine_code(rc, zk_engine_last_hip_error(engine_));
    return (size_t)written;
}

CompressionProgress RawEncoder::compress_with_prefix(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                                                     const uint8_t *prefix, size_t prefix_len)
{
    if (is_frame_complete()) {                                                 // encode.rs:317-327
        size_t out_progress = 0;
        while (out_progress < out_len) {
            EpilogueProgress p = end_frame(out + out_progress, out_len - out_progress);
            out_progress += p.out_progress();
            if (p.data_left() == 0) break;
        }
        return {0, out_progress};
    }
    size_t limit = std::min(in_len, remaining_frame_size());                   // encode.rs:329
    const bool by_output = policy_.kind == FrameSizePolicy::Kind::Compressed;
    const size_t slack = 131072 - 4096;
    if (by_output) {
        limit = std::min(limit, std::min<size_t>(out_len, 131591));
        // a call never takes input past the next probe: the frame-so-far is then measured exactly where the schedule says --
        // taking a whole 131 591-byte call beyond it let incompressible input carry a frame past n + 131 591
        const size_t want0 = policy_.size;
        if (next_probe_ == 0) next_probe_ = std::max<size_t>(1, std::min<size_t>(ratio_ > 0 ? (size_t)(0.97 * ratio_ * (double)want0) : want0, want0 + slack));
        if (next_probe_ > frame_in_.size()) limit = std::min(limit, next_probe_ - frame_in_.size());
    }
    // the prefix of the call that starts a frame is the frame's prefix (ref_prefix only if frame_d_size == 0,
    // encode.rs:334-338); like libzstd only the reference is kept until the frame is encoded
    if (frame_d_size_ == 0) { frame_prefix_ = prefix; frame_prefix_len_ = prefix ? prefix_len : 0; }
    frame_in_.insert(frame_in_.end(), in, in + limit);
    frame_d_size_ += (uint32_t)limit;                                          // encode.rs:350
    if (by_output && limit) {
        const size_t want = policy_.size;
        // (b more compressed bytes need at least ~b more input bytes: a probe never lies further ahead than that plus the
        // window, so even input that stops compressing cannot carry the frame past n + 131 591)
        if (frame_in_.size() >= next_probe_) {
            const size_t d = frame_in_.size();
            bool exact = true;
            if (coarse_) {
                // Far from n on input that compresses very well (d >> c), a probe -- a full encode of the frame so far -- every
                // (n - c) + slack bytes is quadratic work (192 MiB of zeros under Compressed(1 MiB): 207 s), and a geometric step
                // instead (round 3: 7 s) gives up the window when the input stops compressing inside a step.  So, while
                // less than half of n is reached, only the bytes since the last probe are encoded, as a frame of their own: what
                // they add to the frame is at most that (they lose their history, nothing else), the sum stays an upper bound of
                // the frame's size, every piece is O(its length), and the steps keep the cap that keeps the window.
                // ("upper bound" is a heuristic -- the FSE tables are fitted per frame, so pieces of heterogeneous input can
                // come out a little smaller apart than together; coarse mode therefore ends at HALF of n by the bound, and the frame's
                // real size decides from there on.)
                est_c_ += encode_piece(piece_start_, d);
                piece_start_ = d;
                if (2 * est_c_ < want) {
                    next_probe_ = d + (want - est_c_) + slack;
                    exact = false;
                } else coarse_ = false;                                          // (near n by the bound: time for the frame's real size)
            }
            if (exact) {
                encoded_ = false;
                encode_pending();                                              // speculative: how large is the frame so far?
                const size_t c = pending_.size();
                if (c < want) {
                    encoded_ = false;
                    const double r = std::max(1.0, (double)d / (double)std::max<size_t>(c, 1));
                    const size_t by_ratio = std::max<size_t>(32768, (size_t)(0.9 * r * (double)(want - c)));
                    const size_t step = std::min<size_t>(by_ratio, (want - c) + slack);
                    // whole-frame probes every `step` bytes from here would re-encode more than eight times what they add
                    if (2 * c < want && d / 8 > step) { coarse_ = true; est_c_ = c; piece_start_ = d; }
                    next_probe_ = d + step;
                } else ratio_ = (double)d / (double)c;
            }
        }
    }
    return {limit, 0};
}

EpilogueProgress RawEncoder::end_frame(uint8_t *out, size_t out_len)           // encode.rs:438-472
{
    if (encoded_ && pending_pos_ == 0 && policy_.kind == FrameSizePolicy::Kind::Compressed && !is_frame_complete()) encoded_ = false;
    encode_pending();
    const size_t n = std::min(out_len, pending_.size() - pending_pos_);
    memcpy(out, pending_.data() + pending_pos_, n);
    pending_pos_ += n;
    frame_c_size_ += (uint32_t)n;
    const size_t left = pending_.size() - pending_pos_;
    if (left) return {n, left};                                                // more buffer space is required
    seek_table_.log_frame(frame_c_size_, frame_d_size_);                       // encode.rs:466-467
    reset_frame();
    return {n, 0};
}

void RawEncoder::reset_frame()                                                 // encode.rs:501-507
{
    frame_c_size_ = 0; frame_d_size_ = 0;
    frame_in_.clear(); pending_.clear(); pending_pos_ = 0; encoded_ = false; next_probe_ = 0;
    coarse_ = false; est_c_ = 0; piece_start_ = 0;
    frame_prefix_ = nullptr; frame_prefix_len_ = 0;
}

// ---------------------------------------------------------------- Encoder<W>
Encoder::Encoder(std::shared_ptr<Writer> writer) : Encoder(std::move(writer), EncodeOptions()) {}

Encoder::Encoder(std::shared_ptr<Writer> writer, EncodeOptions &&opts)         // with_opts, encode.rs:596-606
    : raw_(EncodeOptions(opts)), writer_(std::move(writer)), out_buf_(131591 /* ZSTD_CStreamOutSize, encode.rs:599 */),
      batch_frames_(opts.batch_frames_)
{
}

Encoder::~Encoder() { if (batch_in_) zk_host_free(batch_in_); }

Encoder::Encoder(Encoder &&o) noexcept
    : raw_(std::move(o.raw_)), writer_(std::move(o.writer_)), out_buf_(std::move(o.out_buf_)), out_buf_pos_(o.out_buf_pos_),
      written_compressed_(o.written_compressed_), batch_frames_(o.batch_frames_), since_end_(o.since_end_), batch_in_(o.batch_in_),
      batch_len_(o.batch_len_), batch_cap_(o.batch_cap_), batch_prefix_(o.batch_prefix_), batch_prefix_len_(o.batch_prefix_len_)
{
    o.batch_in_ = nullptr; o.batch_len_ = o.batch_cap_ = 0;
}

void Encoder::emit(const uint8_t *p, size_t n)                                 // through the staging buffer, like encode.rs:641-665
{
    while (n) {
        // a large piece with nothing staged goes to the writer as it is (same byte stream, fewer write_all calls)
        if (out_buf_pos_ == 0 && n >= out_buf_.size()) {
            writer_->write_all(p, n);
            written_compressed_ += n;
            return;
        }
        const size_t k = std::min(n, out_buf_.size() - out_buf_pos_);
        memcpy(out_buf_.data() + out_buf_pos_, p, k);
        out_buf_pos_ += k; p += k; n -= k;
        flush_out_buf(false);
    }
}

void Encoder::flush_out_buf(bool force)                                        // encode.rs:779-787
{
    if (out_buf_pos_ == out_buf_.size() || force) {
        writer_->write_all(out_buf_.data(), out_buf_pos_);
        written_compressed_ += out_buf_pos_;
        out_buf_pos_ = 0;
    }
}

void Encoder::batch_append(const uint8_t *p, size_t n)
{
    if (batch_len_ + n > batch_cap_) {
        size_t want = std::max<size_t>(batch_len_ + n, batch_cap_ * 2);
        want = std::max<size_t>(want, 1u << 20);
        uint8_t *q = (uint8_t *)zk_host_alloc(want);     // pinned: the engine uploads it by DMA without staging
        if (!q) throw std::bad_alloc();
        if (batch_len_) memcpy(q, batch_in_, batch_len_);
        if (batch_in_) zk_host_free(batch_in_);
        batch_in_ = q; batch_cap_ = want;
    }
    if (n >= (8u << 20)) zk_host_copy(raw_.engine_, batch_in_ + batch_len_, p, n);
    else memcpy(batch_in_ + batch_len_, p, n);
    batch_len_ += n;
}

// what the engine's host pipeline hands back per chunk: compressed bytes (in order), then the chunk's seek entries
int Encoder::sink(void *user, const uint8_t *data, uint64_t n, const uint32_t *c_sizes, const uint32_t *d_sizes, uint32_t n_frames)
{
    Encoder *self = (Encoder *)user;
    try {
        if (n) self->emit(data, (size_t)n);
        for (uint32_t i = 0; i < n_frames; i++) self->raw_.seek_table_.log_frame(c_sizes[i], d_sizes[i]);
    } catch (...) { self->sink_error_ = std::current_exception(); return 1; }
    return 0;
}

// Encode src[0, take) (whole frames of fs bytes, the last one possibly short) with one pipelined engine call.
void Encoder::encode_span(const uint8_t *src, size_t take)
{
    const uint32_t fs = std::min(MAX_FRAME_SIZE, raw_.policy_.size);
    const void *d_prefix = nullptr;
    const uint64_t tail = batch_prefix_ ? zke_ldm_usable(batch_prefix_len_) : 0;    // what the matcher can reach (ring window + long-distance table)
    int rc = zk_engine_stage_prefix(raw_.engine_, this, tail ? batch_prefix_ + (batch_prefix_len_ - tail) : nullptr, tail, prefix_dirty_, &d_prefix);
    if (rc != 0) throw Error::from_engine_code(rc, zk_engine_last_hip_error(raw_.engine_));
    prefix_dirty_ = false;
    sink_error_ = nullptr;
    rc = zk_host_encode(raw_.engine_, src, take, fs, raw_.level_, raw_.checksum_ ? 1 : 0, d_prefix, tail, &Encoder::sink, this);
    if (sink_error_) std::rethrow_exception(sink_error_);
    if (rc != 0) throw Error::from_engine_code(rc, zk_engine_last_hip_error(raw_.engine_));
}

// Encode the complete frames gathered so far (and the partial tail frame if asked) with one submission.
void Encoder::submit_batch(bool include_partial)
{
    const uint32_t fs = std::min(MAX_FRAME_SIZE, raw_.policy_.size);
    // without include_partial the trailing frame stays behind, full or not: it is still open upstream
    size_t take = batch_len_ == 0 ? 0 : ((batch_len_ - 1) / fs) * (size_t)fs;
    if (include_partial) take = batch_len_;
    if (take == 0 && !include_partial) return;
    encode_span(batch_in_, take);
    if (take < batch_len_) memmove(batch_in_, batch_in_ + take, batch_len_ - take);
    batch_len_ -= take;
}

// Compressed(n) on a large write.  Where a frame ends depends on its compressed size, which upstream learns call by call; a
// GPU learns it by encoding.  So the ends are PREDICTED from the running ratio -- every frame is given the input that should
// compress to the middle of upstream's window [n, n + 131 591) -- a whole batch of such frames is encoded with one engine
// call, and the frames are then accepted in order for as long as their compressed size lies inside that window.  The first
// one that does not corrects the ratio and everything from it on is cut again; a position that fails three times is left to
// the exact frame-by-frame path (RawEncoder).  Returns the bytes consumed = whole accepted frames; the rest of the write,
// at least two frames' worth, stays with the caller's loop (the last frame is still open upstream).
size_t Encoder::speculate_compressed(const uint8_t *buf, size_t len, const uint8_t *prefix, size_t prefix_len)
{
    const uint64_t want = raw_.policy_.size, window = out_buf_.size();
    double ratio = raw_.ratio_ > 0 ? raw_.ratio_ : 2.0;
    size_t pos = 0;
    int failures = 0;
    while (failures < 3) {
        const uint64_t s64 = (uint64_t)(ratio * (double)(want + window / 2));
        if (s64 == 0 || s64 >= MAX_FRAME_SIZE) break;                          // one frame per MAX_FRAME_SIZE: the exact path knows that rule
        const size_t s = (size_t)s64;
        if (len - pos < 3 * s) break;
        size_t k = (len - pos) / s - 2;                                        // the tail (two frames' worth or more) is not speculated on
        if (raw_.ratio_ <= 0 && pos == 0) k = std::min<size_t>(k, 4);          // no ratio yet: a small first batch measures it
        k = std::min<size_t>(k, 8192);
        spec_out_.resize((size_t)zk_compress_bound((uint64_t)k * s, (uint32_t)s));
        spec_c_.resize(k); spec_d_.resize(k);
        uint32_t nf = 0;
        uint64_t written = 0;
        const int rc = zk_encode_frames_prefix(raw_.engine_, buf + pos, (uint64_t)k * s, (uint32_t)s, raw_.level_, raw_.checksum_ ? 1 : 0,
                                               prefix, prefix ? prefix_len : 0, spec_out_.data(), spec_out_.size(), spec_c_.data(), spec_d_.data(),
                                               (uint32_t)k, &nf, &written);
        if (rc != 0) throw Error::from_engine_code(rc, zk_engine_last_hip_error(raw_.engine_));
        size_t at = 0, good = 0;
        uint64_t sum_c = 0;
        while (good < nf && spec_c_[good] >= want && spec_c_[good] < want + window) { sum_c += spec_c_[good]; at += spec_c_[good]; good++; }
        if (good) {
            emit(spec_out_.data(), at);
            for (size_t i = 0; i < good; i++) raw_.seek_table_.log_frame(spec_c_[i], spec_d_[i]);
            pos += good * s;
            ratio = (double)(good * s) / (double)sum_c;
            raw_.ratio_ = ratio;
            failures = 0;
        }
        if (good < nf) {                                                       // the frame at pos missed the window: its own ratio cuts it again
            ratio = (double)s / (double)std::max<uint32_t>(spec_c_[good], 1);
            failures++;
        }
    }
    return pos;
}

// Compressed(n): p[0, n) through the policy in order -- batches of predicted frames while no frame is open and plenty of
// input is left, upstream's own loop (encode.rs:641-665) around the exact path for the rest.  The last frame stays open.
void Encoder::process_compressed(const uint8_t *p, size_t n, const uint8_t *prefix, size_t prefix_len, bool speculate)
{
    size_t done = 0;
    while (done < n) {
        if (speculate && raw_.frame_d_size_ == 0 && !raw_.encoded_ && n - done >= (8u << 20)) {
            const size_t took = speculate_compressed(p + done, n - done, prefix, prefix_len);
            done += took;
            if (took) continue;
        }
        CompressionProgress q = raw_.compress_with_prefix(p + done, n - done, out_buf_.data() + out_buf_pos_, out_buf_.size() - out_buf_pos_,
                                                          prefix, prefix_len);
        out_buf_pos_ += q.out_progress();
        flush_out_buf(false);
        done += q.in_progress();
    }
}

size_t Encoder::compress_with_prefix(const uint8_t *buf, size_t len, const uint8_t *prefix, size_t prefix_len)   // encode.rs:641-665
{
    if (!prefix) prefix_len = 0;
    if (raw_.policy_.kind == FrameSizePolicy::Kind::Compressed) {               // frame ends depend on output
        since_end_ += len;
        const bool same_prefix = prefix == batch_prefix_ && prefix_len == batch_prefix_len_;
        // (frames of tens of MiB compressed are not worth gathering three of: the exact path takes them as they come)
        if (raw_.frame_d_size_ || raw_.encoded_ || (batch_len_ && !same_prefix) || raw_.policy_.size > (16u << 20)) {
            // a frame is open in the exact path, or the prefix changes under gathered bytes (upstream: it takes effect at the
            // next frame start, encode.rs:334-338): everything goes through in order, nothing is held back
            const size_t held = batch_len_;
            if (held) { process_compressed(batch_in_, held, batch_prefix_, batch_prefix_len_, true); batch_len_ = 0; }
            batch_prefix_ = prefix; batch_prefix_len_ = prefix_len;
            process_compressed(buf, len, prefix, prefix_len, true);
            return len;
        }
        batch_prefix_ = prefix; batch_prefix_len_ = prefix_len;
        size_t took = 0;
        if (batch_len_ == 0 && len >= (32u << 20)) took = speculate_compressed(buf, len, prefix, prefix_len);     // where it lies
        batch_append(buf + took, len - took);
        if (batch_len_ >= (16u << 20)) {
            const size_t t = speculate_compressed(batch_in_, batch_len_, prefix, prefix_len);
            if (t) { memmove(batch_in_, batch_in_ + t, batch_len_ - t); batch_len_ -= t; }
            else if (batch_len_ >= (64u << 20)) {
                // speculation takes nothing (input so compressible that a frame would pass MAX_FRAME_SIZE, or a position that
                // keeps failing): the exact path drains what is held -- the pinned batch stays bounded -- and keeps the frame it
                // opens, so the following writes go straight through it
                const size_t held = batch_len_;
                process_compressed(batch_in_, held, prefix, prefix_len, true);
                batch_len_ = 0;
            }
        }
        return len;
    }
    const uint32_t fs = std::min(MAX_FRAME_SIZE, raw_.policy_.size);
    const size_t total_len = len;
    // A batch is encoded against ONE prefix.  Upstream a new prefix takes effect at the next frame start
    // (encode.rs:334-338); the frames gathered so far began under the old one, so they are cut and submitted first:
    // complete frames as they are, and the open frame is filled up to its boundary from this call's bytes.
    if (batch_len_ && (prefix != batch_prefix_ || prefix_len != batch_prefix_len_)) {
        const size_t open = batch_len_ % fs;
        const size_t fill = open ? std::min<size_t>(len, fs - open) : 0;
        batch_append(buf, fill);
        since_end_ += fill;
        if (fill == len && batch_len_ % fs) return len;                         // still inside the old frame: nothing switches yet
        submit_batch(true);                                                     // every gathered frame is complete here
        buf += fill; len -= fill;
    }
    if (batch_len_ == 0 && (prefix != batch_prefix_ || prefix_len != batch_prefix_len_)) {
        batch_prefix_ = prefix; batch_prefix_len_ = prefix_len; prefix_dirty_ = true;
    }
    since_end_ += len;
    // A large write is encoded where it lies: the open frame is completed from it, then every whole frame but the last goes
    // to the engine straight from the caller's buffer (chunked and overlapped with the PCIe legs inside the engine); only the
    // last frame -- still open upstream until the next call (encode.rs:317-327) -- is copied into the batch buffer.
    if (len >= (size_t)fs * 4 || len >= (64u << 20)) {
        if (batch_len_) {
            // (with frames larger than the write -- fs > len is possible here from 64 MiB on -- the open frame may take all of it)
            const size_t fill = std::min<size_t>(len, (fs - batch_len_ % fs) % fs);
            batch_append(buf, fill);
            buf += fill; len -= fill;
            if (len == 0) {                                                     // the write ended inside (or exactly at the end of) the open frame
                if (batch_len_ > (size_t)fs * batch_frames_) submit_batch(false);
                return total_len;
            }
            submit_batch(true);                                                 // every gathered frame is complete and more input follows
        }
        const size_t take = ((len - 1) / fs) * (size_t)fs;
        if (take) encode_span(buf, take);
        batch_append(buf + take, len - take);
        return total_len;
    }
    batch_append(buf, len);
    // the last full frame stays open (upstream closes it on the NEXT call), so it is held back
    if (batch_len_ > (size_t)fs * batch_frames_) submit_batch(false);
    return total_len;
}

size_t Encoder::end_frame()                                                    // encode.rs:704-717
{
    // Upstream a frame stays open -- even when it is full -- until the next compress() call or end_frame():
    // ending a frame that has received no byte since the last end_frame yields an EMPTY frame, anything
    // else is exactly the frames already cut every frame_size bytes plus the partial tail.
    const uint64_t before = written_compressed_ + out_buf_pos_;
    if (raw_.policy_.kind == FrameSizePolicy::Kind::Compressed) {               // encode.rs:704-717 verbatim, after the gathered bytes
        if (batch_len_) { const size_t held = batch_len_; process_compressed(batch_in_, held, batch_prefix_, batch_prefix_len_, true); batch_len_ = 0; }
        for (;;) {
            EpilogueProgress p = raw_.end_frame(out_buf_.data() + out_buf_pos_, out_buf_.size() - out_buf_pos_);
            out_buf_pos_ += p.out_progress();
            flush_out_buf(false);
            if (p.data_left() == 0) break;
        }
    } else if (since_end_ == 0 || batch_len_) submit_batch(true);
    since_end_ = 0;
    return (size_t)(written_compressed_ + out_buf_pos_ - before);
}

uint64_t Encoder::finish_format(Format format)                                 // encode.rs:755-775
{
    end_frame();                                                               // always (encode.rs:756): an empty stream yields one empty frame
    Serializer ser = raw_.seek_table_.into_format_serializer(format);
    for (;;) {
        size_t n = ser.write_into(out_buf_.data() + out_buf_pos_, out_buf_.size() - out_buf_pos_);
        if (n == 0) {
            writer_->write_all(out_buf_.data(), out_buf_pos_);
            written_compressed_ += out_buf_pos_;
            out_buf_pos_ = 0;
            writer_->flush();
            return written_compressed_;
        }
        out_buf_pos_ += n;
        if (out_buf_pos_ == out_buf_.size()) {
            writer_->write_all(out_buf_.data(), out_buf_pos_);
            written_compressed_ += out_buf_pos_;
            out_buf_pos_ = 0;
        }
    }
}

void Encoder::flush()                                                          // io::Write::flush, encode.rs:796-799
{
    if (raw_.policy_.kind == FrameSizePolicy::Kind::Compressed) {
        if (batch_len_ && !raw_.frame_d_size_ && !raw_.encoded_) {
            const size_t t = speculate_compressed(batch_in_, batch_len_, batch_prefix_, batch_prefix_len_);
            if (t) { memmove(batch_in_, batch_in_ + t, batch_len_ - t); batch_len_ -= t; }
        }
    } else submit_batch(false);                                                // complete frames are pushed out
    flush_out_buf(true);
    writer_->flush();
}

}  // namespace zeekstd
