// zeekstd.hpp -- host-side mirror of the zeekstd crate's public API (Level B of include/zeekstd_amd.h).
//
// Same type names, argument meaning and error behaviour as the Rust reference, restated in C++ on top
// of the batch engine (Level A).  Each declaration cites the reference item it mirrors
// (paths relative to /root/reference/lib/src).  Rust `Result<T>` becomes "returns T or throws
// zeekstd::Error"; the C ABI (zk_* handle functions) converts exceptions back into error codes.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <array>
#include <exception>
#include <memory>
#include <optional>
#include <string>
#include <vector>

struct zk_engine;

namespace zeekstd {

// ---------------------------------------------------------------- constants (lib.rs:52-62)
constexpr uint32_t SEEKABLE_MAGIC_NUMBER = 0x8F92EAB1u;
constexpr uint32_t SEEKABLE_MAX_FRAMES = 0x08000000u;
constexpr size_t SEEK_TABLE_INTEGRITY_SIZE = 9;
constexpr size_t SEEKABLE_MAX_FRAME_SIZE = 0x40000000u;
constexpr size_t SKIPPABLE_HEADER_SIZE = 8;
constexpr uint32_t SKIPPABLE_MAGIC_NUMBER = 0x184D2A50u | 0xE;      // seek_table.rs:89
using CompressionLevel = int32_t;                                   // lib.rs:49 (zstd_safe::CompressionLevel)

// ---------------------------------------------------------------- Error (error.rs:4-128)
class Error : public std::exception {
public:
    enum class Kind { NumberConversionFailed, OffsetOutOfRange, FrameIndexTooLarge, IO, Zstd };
    static Error offset_out_of_range() { return Error(Kind::OffsetOutOfRange, 0, ""); }          // error.rs:18-22
    static Error frame_index_too_large() { return Error(Kind::FrameIndexTooLarge, 0, ""); }       // error.rs:29-33
    static Error number_conversion_failed(const std::string &what) { return Error(Kind::NumberConversionFailed, 0, what); }
    static Error io(const std::string &what) { return Error(Kind::IO, 0, what); }
    // ZSTD_ErrorCode -> wrapped "0 - code" like error.rs:40-45
    static Error zstd(uint32_t zstd_error_code) { return Error(Kind::Zstd, (size_t)0 - (size_t)zstd_error_code, ""); }
    static Error from_engine_code(int rc, const std::string &detail = "");   // negative code of the C ABI

    bool is_number_conversion_failed() const { return kind_ == Kind::NumberConversionFailed; }    // error.rs:14
    bool is_offset_out_of_range() const { return kind_ == Kind::OffsetOutOfRange; }                // error.rs:25
    bool is_frame_index_too_large() const { return kind_ == Kind::FrameIndexTooLarge; }            // error.rs:36
    bool is_io() const { return kind_ == Kind::IO; }                                               // error.rs:50
    bool is_zstd() const { return kind_ == Kind::Zstd; }                                           // error.rs:55
    Kind kind() const { return kind_; }
    size_t raw_zstd_code() const { return code_; }                   // the wrapped size_t, as zstd-safe's ErrorCode
    int abi_code() const;                                            // the negative int used by the C ABI
    const char *what() const noexcept override { return msg_.c_str(); }   // Display, error.rs:60-71

private:
    Error(Kind k, size_t code, const std::string &detail);
    Kind kind_;
    size_t code_;
    std::string msg_;
};

// ---------------------------------------------------------------- Format / OffsetFrom / Seekable
enum class Format { Head, Foot };                                   // seek_table.rs:228-241 (default Foot)

struct OffsetFrom {                                                 // seekable.rs:8-13
    enum class From { Start, End } from;
    int64_t value;                                                  // Start: u64 offset; End: i64 delta
    static OffsetFrom Start(uint64_t v) { return {From::Start, (int64_t)v}; }
    static OffsetFrom End(int64_t v) { return {From::End, v}; }
};

class Seekable {                                                    // seekable.rs:16-39
public:
    virtual ~Seekable() = default;
    virtual uint64_t set_offset(OffsetFrom offset) = 0;
    virtual size_t read(uint8_t *buf, size_t len) = 0;
    virtual std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> seek_table_integrity(Format format) = 0;
    // engine-specific: a source that is one contiguous byte range says so, and the host pipeline reads it in place
    virtual const uint8_t *contiguous(size_t *len) const { (void)len; return nullptr; }
};

class BytesWrapper : public Seekable {                              // seekable.rs:43-97
public:
    BytesWrapper(const uint8_t *src, size_t len) : src_(src), len_(len), pos_(0) {}
    uint64_t set_offset(OffsetFrom offset) override;
    size_t read(uint8_t *buf, size_t len) override;
    std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> seek_table_integrity(Format format) override;
    const uint8_t *contiguous(size_t *len) const override { *len = len_; return src_; }

private:
    const uint8_t *src_;
    size_t len_, pos_;
};

// A Seekable made of three callbacks: what a host in another language plugs its own `impl Seekable` into
// (seekable.rs:16-39; zk_decoder_open_callbacks of the C ABI).  whence: 0 = from the start, 1 = from the end.
class CallbackSeekable : public Seekable {
public:
    typedef int64_t (*set_offset_fn)(void *user, int whence, int64_t value);      // new position from the start, or < 0
    typedef int64_t (*read_fn)(void *user, uint8_t *buf, size_t len);             // bytes read (0 = end), or < 0
    typedef int (*integrity_fn)(void *user, int format, uint8_t out[SEEK_TABLE_INTEGRITY_SIZE]);   // 0, or < 0 on failure
    CallbackSeekable(set_offset_fn so, read_fn rd, void *user, integrity_fn ig = nullptr) : so_(so), rd_(rd), ig_(ig), user_(user) {}
    uint64_t set_offset(OffsetFrom offset) override;
    size_t read(uint8_t *buf, size_t len) override;
    std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> seek_table_integrity(Format format) override;

private:
    set_offset_fn so_;
    read_fn rd_;
    integrity_fn ig_;
    void *user_;
};

// The blanket `impl<T: Read + Seek> Seekable for T` (seekable.rs:112-138), for stdio files.
class FileSeekable : public Seekable {
public:
    explicit FileSeekable(FILE *f, bool owns = false) : f_(f), owns_(owns) {}
    ~FileSeekable() override { if (owns_ && f_) fclose(f_); }
    uint64_t set_offset(OffsetFrom offset) override;
    size_t read(uint8_t *buf, size_t len) override;
    std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> seek_table_integrity(Format format) override;

private:
    FILE *f_;
    bool owns_;
};

// ---------------------------------------------------------------- SeekTable (seek_table.rs:243-935)
class Serializer;

class SeekTable {
public:
    SeekTable();                                                                     // new, :287
    static SeekTable from_seekable(Seekable &src) { return from_seekable_format(src, Format::Foot); }   // :338
    static SeekTable from_seekable_format(Seekable &src, Format format);             // :379-436
    // from_reader (:461-493): Head format only, any forward-only byte source
    struct Reader { virtual ~Reader() = default; virtual size_t read(uint8_t *buf, size_t len) = 0; };
    static SeekTable from_reader(Reader &reader);
    static SeekTable from_bytes_head(const uint8_t *p, size_t len);                  // convenience over from_reader

    void log_frame(uint32_t c_size, uint32_t d_size);                                // :513-525
    uint32_t num_frames() const { return (uint32_t)(entries_.size() - 1); }          // :540
    uint32_t frame_index_comp(uint64_t offset) const;                                // :560
    uint32_t frame_index_decomp(uint64_t offset) const;                              // :579
    uint64_t frame_start_comp(uint32_t index) const;                                 // :604
    uint64_t frame_start_decomp(uint32_t index) const;                               // :633
    uint64_t frame_end_comp(uint32_t index) const;                                   // :662
    uint64_t frame_end_decomp(uint32_t index) const;                                 // :691
    uint64_t frame_size_comp(uint32_t index) const;                                  // :720
    uint64_t frame_size_decomp(uint32_t index) const;                                // :750
    uint64_t max_frame_size_comp() const;                                            // :774
    uint64_t max_frame_size_decomp() const;                                          // :799
    uint64_t size_comp() const { return entries_.back().c_offset; }                  // :827
    uint64_t size_decomp() const { return entries_.back().d_offset; }                // :853
    Serializer into_serializer() const;                                              // :883
    Serializer into_format_serializer(Format format) const;                          // :907
    bool operator==(const SeekTable &o) const;                                       // derive(PartialEq), :266

    struct Entry { uint64_t c_offset, d_offset; };
    const std::vector<Entry> &entries() const { return entries_; }                   // n+1 prefix sums

private:
    uint32_t frame_index_at(uint64_t offset, bool comp) const;                       // :916-934
    std::vector<Entry> entries_;
};

class Serializer {                                                                   // seek_table.rs:937-1059
public:
    size_t write_into(uint8_t *buf, size_t len);                                     // :967-1005 (0 == done)
    void reset() { write_pos_ = 0; }                                                 // :1034
    size_t encoded_len() const { return SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE + frames_.size() * 8; }   // :1042
    size_t read(uint8_t *buf, size_t len) { return write_into(buf, len); }           // impl io::Read, :1055-1059

private:
    friend class SeekTable;
    struct Frame { uint32_t c_size, d_size; };
    std::vector<Frame> frames_;
    uint8_t byte_at(size_t pos) const;         // the table as a function of the byte position
    size_t write_pos_ = 0;
    Format format_ = Format::Foot;
};

// ---------------------------------------------------------------- decode side (decode.rs)
class Decoder;

class DecodeOptions {                                                                // decode.rs:13-114
public:
    explicit DecodeOptions(std::shared_ptr<Seekable> src) : src_(std::move(src)) {}  // new, :30
    DecodeOptions &engine(zk_engine *e) { engine_ = e; return *this; }               // with_dctx/dctx (:43,:56): inject the context
    DecodeOptions &seek_table(SeekTable t) { seek_table_ = std::move(t); return *this; }   // :65
    DecodeOptions &lower_frame(uint32_t i) { lower_frame_ = i; return *this; }       // :73
    DecodeOptions &upper_frame(uint32_t i) { upper_frame_ = i; return *this; }       // :81
    DecodeOptions &offset(uint64_t o) { offset_ = o; return *this; }                 // :90
    DecodeOptions &offset_limit(uint64_t l) { offset_limit_ = l; return *this; }     // :99
    // engine-specific knobs (no reference counterpart): how much to decode per GPU submission
    DecodeOptions &batch_bytes(uint64_t b) { batch_bytes_ = b; return *this; }
    DecodeOptions &verify_checksums(bool v) { verify_ = v; return *this; }
    Decoder into_decoder();                                                          // :111

private:
    friend class Decoder;
    std::shared_ptr<Seekable> src_;
    zk_engine *engine_ = nullptr;
    std::optional<SeekTable> seek_table_;
    std::optional<uint32_t> lower_frame_, upper_frame_;
    std::optional<uint64_t> offset_, offset_limit_;
    uint64_t batch_bytes_ = 64ull << 20;
    bool verify_ = true;
};

class Decoder {                                                                      // decode.rs:117-466
public:
    explicit Decoder(std::shared_ptr<Seekable> src);                                 // new, :143
    explicit Decoder(DecodeOptions &&opts);                                          // with_opts, :152-187
    ~Decoder();
    Decoder(Decoder &&) noexcept;
    Decoder(const Decoder &) = delete;

    // :201-270; prefix (patch mode): every frame sees the prefix right before its first byte (:212-214, 248-255);
    // like libzstd only the reference is kept -- the buffer must stay unchanged while it is in use
    // Error::zstd(parameter_unsupported)
    size_t decompress_with_prefix(uint8_t *buf, size_t len, const uint8_t *prefix, size_t prefix_len);
    size_t decompress(uint8_t *buf, size_t len) { return decompress_with_prefix(buf, len, nullptr, 0); }   // :314
    void reset();                                                                    // :346-350
    uint64_t set_lower_frame(uint32_t index);                                        // :367
    uint64_t set_upper_frame(uint32_t index);                                        // :383
    void set_offset(uint64_t offset);                                                // :402-414
    void set_offset_limit(uint64_t limit);                                           // :432-437
    uint64_t read_compressed() const { return read_compressed_; }                    // :448
    const SeekTable &seek_table() const { return seek_table_; }                      // :453
    uint64_t offset() const { return offset_; }                                      // :458
    uint64_t offset_limit() const { return offset_limit_; }                          // :463
    size_t read(uint8_t *buf, size_t len) { return decompress(buf, len); }           // impl io::Read, :510-514
    enum class SeekFrom { Start, End, Current };
    uint64_t seek(SeekFrom from, int64_t n);                                         // impl io::Seek, :545-579
    uint64_t gpu_submissions() const { return submissions_; }                        // observability (engine-specific)

private:
    void check_offset(uint64_t offset) const;                                        // :439-445
    void reset_dctx(bool keep_cache = false);
    void count_frames(uint32_t first, uint32_t end);                                                               // :352-357
    void fill_cache(uint64_t want_end, uint64_t request_end, const uint8_t *prefix, size_t prefix_len);
    void check_frames(uint32_t first, uint32_t count) const;
    uint32_t decode_range(uint32_t first, uint32_t count, uint8_t *dst, uint64_t dst_cap, const uint8_t *prefix, size_t prefix_len, uint32_t *err);
    const uint8_t *cache_prefix_ = nullptr; size_t cache_prefix_len_ = 0;           // the prefix the cached frames were decoded with
    zk_engine *engine_ = nullptr;
    bool owns_engine_ = false;
    SeekTable seek_table_;
    std::shared_ptr<Seekable> src_;
    uint64_t offset_ = 0, offset_limit_ = 0, read_compressed_ = 0;
    uint32_t counted_lo_ = 0, counted_hi_ = 0;                  // frames whose compressed bytes read_compressed_ holds since the last reset
    uint64_t batch_bytes_ = 64ull << 20;
    bool verify_ = true;
    // decoded frames [cache_first_, cache_first_ + cache_count_) live in cache_
    // (pinned host memory: the engine's D2H lands in it without staging)
    uint8_t *cache_ = nullptr; size_t cache_cap_ = 0;
    uint32_t cache_first_ = 0, cache_count_ = 0;
    uint64_t cache_d_start_ = 0, cache_d_end_ = 0;
    uint64_t cache_unverified_end_ = 0;        // end offset of a cached frame whose checksum went unchecked (cut by offset_limit), or 0
    uint64_t unverified_end_tmp_ = 0;
    bool prefix_dirty_ = true;                 // the engine's staged copy of the prefix must be refreshed
    uint64_t submissions_ = 0;
};

// ---------------------------------------------------------------- encode side (encode.rs)
struct FrameSizePolicy {                                                             // encode.rs:21-39
    enum class Kind { Compressed, Uncompressed } kind = Kind::Uncompressed;
    uint32_t size = 0x200000;
    static FrameSizePolicy Compressed(uint32_t n) { return {Kind::Compressed, n}; }
    static FrameSizePolicy Uncompressed(uint32_t n) { return {Kind::Uncompressed, n}; }
};

struct CompressionProgress {                                                         // encode.rs:43-66
    size_t in_progress_, out_progress_;
    size_t in_progress() const { return in_progress_; }
    size_t out_progress() const { return out_progress_; }
};

struct EpilogueProgress {                                                            // encode.rs:69-92
    size_t out_progress_, data_left_;
    size_t out_progress() const { return out_progress_; }
    size_t data_left() const { return data_left_; }
};

class RawEncoder;
class Writer {                                                                       // std::io::Write as Encoder<W> uses it
public:
    virtual ~Writer() = default;
    virtual void write_all(const uint8_t *p, size_t n) = 0;
    virtual void flush() {}
};
class VecWriter : public Writer {
public:
    std::vector<uint8_t> data;
    void write_all(const uint8_t *p, size_t n) override { data.insert(data.end(), p, p + n); }
};
class Encoder;

class EncodeOptions {                                                                // encode.rs:110-207
public:
    EncodeOptions() = default;                                                       // new, :129
    EncodeOptions &engine(zk_engine *e) { engine_ = e; return *this; }               // with_cctx/cctx (:142,:152)
    EncodeOptions &frame_size_policy(FrameSizePolicy p) { policy_ = p; return *this; }   // :158
    EncodeOptions &checksum_flag(bool f) { checksum_ = f; return *this; }            // :164
    EncodeOptions &compression_level(CompressionLevel l) { level_ = l; return *this; }   // :170
    // engine-specific: frames collected per GPU submission by Encoder (RawEncoder is always 1)
    EncodeOptions &batch_frames(uint32_t n) { batch_frames_ = n ? n : 1; return *this; }
    RawEncoder into_raw_encoder();                                                   // :180
    Encoder into_encoder(std::shared_ptr<Writer> writer);                            // :204

private:
    friend class RawEncoder;
    friend class Encoder;
    zk_engine *engine_ = nullptr;
    FrameSizePolicy policy_;
    bool checksum_ = false;                                                          // :146 default false
    CompressionLevel level_ = 0;                                                     // :147 default 0 (== 3)
    uint32_t batch_frames_ = 64;
};

class RawEncoder {                                                                   // encode.rs:209-545
public:
    RawEncoder();                                                                    // new, :365
    explicit RawEncoder(EncodeOptions &&opts);                                       // with_opts, :280-293
    ~RawEncoder();
    RawEncoder(RawEncoder &&) noexcept;
    RawEncoder(const RawEncoder &) = delete;

    CompressionProgress compress_with_prefix(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len,
                                             const uint8_t *prefix, size_t prefix_len);   // :311-354
    CompressionProgress compress(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
    { return compress_with_prefix(in, in_len, out, out_len, nullptr, 0); }           // :398
    EpilogueProgress end_frame(uint8_t *out, size_t out_len);                        // :438-472
    const SeekTable &seek_table() const { return seek_table_; }                      // :487
    SeekTable into_seek_table() { return std::move(seek_table_); }                   // :492
    void reset_frame();                                                              // :501-507
    void reset_seek_table() { seek_table_ = SeekTable(); }                           // :524

private:
    friend class Encoder;
    size_t remaining_frame_size() const;                                             // :528-535
    bool is_frame_complete() const;                                                  // :537-544
    void encode_pending();
    size_t encode_piece(size_t lo, size_t hi);       // compressed size of frame_in_[lo, hi) as a frame of its own
    zk_engine *engine_ = nullptr;
    bool owns_engine_ = false;
    FrameSizePolicy policy_;
    bool checksum_ = false;
    CompressionLevel level_ = 0;
    uint32_t frame_c_size_ = 0, frame_d_size_ = 0;
    SeekTable seek_table_;
    std::vector<uint8_t> frame_in_;           // uncompressed bytes of the frame in progress
    std::vector<uint8_t> pending_;            // encoded frame being drained by end_frame
    size_t pending_pos_ = 0;
    bool encoded_ = false;
    size_t next_probe_ = 0;                   // Compressed(n): buffered size at which the next speculative encode happens
    double ratio_ = 0;                        // Compressed(n): uncompressed / compressed of the last frame closed (0: none yet)
    // Compressed(n), far from n on input that compresses very well: the frame is measured PIECE BY PIECE (each piece as a frame of
    // its own: an upper bound of what it adds) instead of whole again at every probe -- see compress_with_prefix
    bool coarse_ = false;
    size_t est_c_ = 0, piece_start_ = 0;      // the bound so far; where the next piece starts
    std::vector<uint8_t> piece_out_;
    const uint8_t *frame_prefix_ = nullptr;   // prefix referenced when the frame in progress began (encode.rs:334-338)
    size_t frame_prefix_len_ = 0;
};

class Encoder {                                                                      // encode.rs:570-800
public:
    explicit Encoder(std::shared_ptr<Writer> writer);                                // new, :587
    Encoder(std::shared_ptr<Writer> writer, EncodeOptions &&opts);                   // with_opts, :596
    ~Encoder();
    Encoder(Encoder &&) noexcept;
    Encoder(const Encoder &) = delete;
    const SeekTable &seek_table() const { return raw_.seek_table_; }                 // :610
    uint64_t written_compressed() const { return written_compressed_; }              // :615
    SeekTable into_seek_table() { return raw_.into_seek_table(); }                   // :620
    size_t compress_with_prefix(const uint8_t *buf, size_t len, const uint8_t *prefix, size_t prefix_len);   // :641-665
    size_t compress(const uint8_t *buf, size_t len) { return compress_with_prefix(buf, len, nullptr, 0); }   // :692
    size_t end_frame();                                                              // :704-717
    uint64_t finish() { return finish_format(Format::Foot); }                        // :743
    uint64_t finish_format(Format format);                                           // :755-775
    size_t write(const uint8_t *buf, size_t len) { return compress(buf, len); }      // impl io::Write, :791-794
    void flush();                                                                    // :796-799

private:
    void submit_batch(bool include_partial);
    size_t speculate_compressed(const uint8_t *buf, size_t len, const uint8_t *prefix, size_t prefix_len);
    void process_compressed(const uint8_t *p, size_t n, const uint8_t *prefix, size_t prefix_len, bool speculate);
    void encode_span(const uint8_t *src, size_t take);
    void batch_append(const uint8_t *p, size_t n);
    static int sink(void *user, const uint8_t *data, uint64_t n, const uint32_t *c_sizes, const uint32_t *d_sizes, uint32_t n_frames);
    void emit(const uint8_t *p, size_t n);
    void flush_out_buf(bool force);                                                  // :779-787
    RawEncoder raw_;
    std::shared_ptr<Writer> writer_;
    std::vector<uint8_t> out_buf_;            // CCtx::out_size() = 131 591-byte staging buffer, :599
    size_t out_buf_pos_ = 0;
    uint64_t written_compressed_ = 0;
    uint32_t batch_frames_ = 64;
    uint64_t since_end_ = 0;                  // bytes accepted since the last end_frame
    uint8_t *batch_in_ = nullptr;             // whole frames (+ the partial one at the tail) awaiting submission (pinned)
    size_t batch_len_ = 0, batch_cap_ = 0;
    const uint8_t *batch_prefix_ = nullptr;   // the prefix the frames in batch_in_ began with
    size_t batch_prefix_len_ = 0;
    bool prefix_dirty_ = true;                // the engine's staged copy of the prefix must be refreshed
    std::exception_ptr sink_error_;           // a writer failure inside the engine's sink callback
    std::vector<uint8_t> spec_out_;           // Compressed(n): the frames of one speculative batch
    std::vector<uint32_t> spec_c_, spec_d_;
};

}  // namespace zeekstd
