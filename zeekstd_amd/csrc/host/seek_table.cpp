// seek_table.cpp -- Zstandard Seekable Format seek table: in-memory index, serializer, parser.  Host-side integer code.
// The wire format is the spec's (/root/reference/seekable_format.md:45-157) and the observable behaviour -- errors and their
// order, partial reads and writes at any byte, legacy 12-byte entries -- is that of /root/reference/lib/src/seek_table.rs
// (cited per function); the structure is this file's own: the parser is a byte-indexed state machine fed pieces of any
// size, the serializer computes the table as a function of the byte position.
#include <string.h>
#include <algorithm>
#include "zeekstd.hpp"

namespace zeekstd {

namespace {
constexpr uint32_t ZSTD_error_prefix_unknown = 10, ZSTD_error_corruption_detected = 20;

inline uint32_t le32(const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }

// What the 9-byte integrity field says about the table (seekable_format.md:116-157; checks and their order as in
// seek_table.rs:144-172: magic, reserved descriptor bits, frame count).
struct TableShape {
    uint32_t frames = 0;
    uint32_t entry_bytes = 8;            // 12 with the legacy per-frame checksum (descriptor bit 7): carried, never verified
    uint64_t total = 0;                  // bytes of the whole skippable frame
    explicit TableShape(const uint8_t integrity[SEEK_TABLE_INTEGRITY_SIZE])
    {
        if (le32(integrity + 5) != SEEKABLE_MAGIC_NUMBER) throw Error::zstd(ZSTD_error_prefix_unknown);
        const uint8_t descriptor = integrity[4];
        if (descriptor & 0x7C) throw Error::zstd(ZSTD_error_corruption_detected);
        frames = le32(integrity);
        if (frames > SEEKABLE_MAX_FRAMES) throw Error::frame_index_too_large();
        entry_bytes = (descriptor & 0x80) ? 12 : 8;
        total = (uint64_t)frames * entry_bytes + SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE;
    }
    void check_header(const uint8_t header[SKIPPABLE_HEADER_SIZE]) const       // seek_table.rs:174-184
    {
        if (le32(header) != SKIPPABLE_MAGIC_NUMBER) throw Error::zstd(ZSTD_error_prefix_unknown);
        if ((uint64_t)le32(header + 4) + SKIPPABLE_HEADER_SIZE != total) throw Error::zstd(ZSTD_error_corruption_detected);
    }
};

// The entries of a table as a byte-indexed state machine: bytes arrive in pieces of any size (a partial entry needs no
// buffer shuffling), `at` is the index inside the entries region, and every completed little-endian word goes to the
// running sum its place in the entry names (0: compressed size, 1: decompressed size, 2: legacy checksum -- dropped).
class EntryScanner {
public:
    EntryScanner(uint32_t frames, uint32_t entry_bytes) : frames_(frames), entry_bytes_(entry_bytes)
    {
        out_.reserve((size_t)frames + 1);
        out_.push_back({0, 0});
    }
    uint64_t missing() const { return (uint64_t)frames_ * entry_bytes_ - at_; }
    void feed(const uint8_t *p, size_t n)
    {
        for (size_t i = 0; i < n; i++) {
            const uint32_t in_entry = (uint32_t)(at_ % entry_bytes_);
            word_ |= (uint32_t)p[i] << (8 * (in_entry & 3));
            at_++;
            if ((in_entry & 3) != 3) continue;
            if (in_entry == 3) c_ += word_;
            else if (in_entry == 7) d_ += word_;
            word_ = 0;
            if (in_entry + 1 == entry_bytes_) out_.push_back({c_, d_});
        }
    }
    std::vector<SeekTable::Entry> finish()                                     // seek_table.rs:218-224
    {
        if (out_.size() != (size_t)frames_ + 1) throw Error::zstd(ZSTD_error_corruption_detected);
        return std::move(out_);
    }

private:
    uint32_t frames_, entry_bytes_, word_ = 0;
    uint64_t at_ = 0, c_ = 0, d_ = 0;
    std::vector<SeekTable::Entry> out_;
};

// pulls exactly n bytes through `read` into dst (or discards them when dst is null); false when the source ends first
template <typename ReadFn>
bool pull(ReadFn &&read, uint8_t *dst, size_t n, uint8_t *scratch, size_t scratch_len)
{
    size_t got = 0;
    while (got < n) {
        const size_t k = dst ? read(dst + got, n - got) : read(scratch, std::min(scratch_len, n - got));
        if (k == 0) return false;
        got += k;
    }
    return true;
}
template <typename ReadFn>
std::vector<SeekTable::Entry> scan_entries(ReadFn &&read, const TableShape &shape)
{
    EntryScanner scan(shape.frames, shape.entry_bytes);
    std::vector<uint8_t> buf((size_t)std::min<uint64_t>(8192, std::max<uint64_t>(scan.missing(), 1)));
    while (scan.missing()) {
        // short reads are fine (a regression fixed upstream: CHANGELOG_LIB.md:14-15); a source that ends early is corruption
        const size_t k = read(buf.data(), (size_t)std::min<uint64_t>(buf.size(), scan.missing()));
        if (k == 0) throw Error::zstd(ZSTD_error_corruption_detected);
        scan.feed(buf.data(), k);
    }
    return scan.finish();
}
}  // namespace

// ---------------------------------------------------------------- SeekTable
SeekTable::SeekTable() { entries_.push_back({0, 0}); }

bool SeekTable::operator==(const SeekTable &o) const
{
    if (entries_.size() != o.entries_.size()) return false;
    for (size_t i = 0; i < entries_.size(); i++)
        if (entries_[i].c_offset != o.entries_[i].c_offset || entries_[i].d_offset != o.entries_[i].d_offset) return false;
    return true;
}

SeekTable SeekTable::from_seekable_format(Seekable &src, Format format)      // seek_table.rs:379-436
{
    const auto integrity = src.seek_table_integrity(format);
    const TableShape shape(integrity.data());
    if (format == Format::Head) src.set_offset(OffsetFrom::Start(0));
    else src.set_offset(OffsetFrom::End(-(int64_t)shape.total));
    auto read = [&](uint8_t *p, size_t n) { return src.read(p, n); };
    uint8_t header[SKIPPABLE_HEADER_SIZE], skip[SEEK_TABLE_INTEGRITY_SIZE];
    // (the reference reads the header into the start of its buffer on every iteration, :391-399: a source that returns
    //  fewer than 8 bytes at a time would tear it; the bytes already read are kept here -- identical for any other source)
    if (!pull(read, header, sizeof header, nullptr, 0)) throw Error::zstd(ZSTD_error_corruption_detected);
    shape.check_header(header);
    if (format == Format::Head && !pull(read, nullptr, sizeof skip, skip, sizeof skip)) throw Error::zstd(ZSTD_error_corruption_detected);
    SeekTable t;
    t.entries_ = scan_entries(read, shape);
    return t;
}

SeekTable SeekTable::from_reader(Reader &reader)                              // seek_table.rs:461-493 (Head format)
{
    auto read = [&](uint8_t *p, size_t n) { return reader.read(p, n); };
    uint8_t head[SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE];
    if (!pull(read, head, sizeof head, nullptr, 0)) throw Error::io("failed to fill whole buffer");      // read_exact
    const TableShape shape(head + SKIPPABLE_HEADER_SIZE);
    shape.check_header(head);
    SeekTable t;
    t.entries_ = scan_entries(read, shape);
    return t;
}

SeekTable SeekTable::from_bytes_head(const uint8_t *p, size_t len)
{
    struct R : Reader {
        const uint8_t *p; size_t len, pos = 0;
        size_t read(uint8_t *buf, size_t n) override { n = std::min(n, len - pos); memcpy(buf, p + pos, n); pos += n; return n; }
    } r;
    r.p = p; r.len = len;
    return from_reader(r);
}

void SeekTable::log_frame(uint32_t c_size, uint32_t d_size)                   // seek_table.rs:513-525
{
    if (num_frames() >= SEEKABLE_MAX_FRAMES) throw Error::frame_index_too_large();
    const Entry &last = entries_[num_frames()];
    entries_.push_back({last.c_offset + c_size, last.d_offset + d_size});
}

uint32_t SeekTable::frame_index_at(uint64_t offset, bool comp) const           // seek_table.rs:916-934
{
    auto at = [&](uint32_t i) { return comp ? entries_[i].c_offset : entries_[i].d_offset; };
    // (a zero-frame table computes 0u32 - 1 upstream, :917-918 -- a debug-build panic; unreachable through Encoder.
    //  Here it wraps to 0xFFFFFFFF like a release build.)
    if (offset >= at(num_frames())) return num_frames() - 1;
    uint32_t low = 0, high = num_frames();
    while (low + 1 < high) {
        uint32_t mid = low + (high - low) / 2;
        if (at(mid) <= offset) low = mid; else high = mid;
    }
    return low;
}
uint32_t SeekTable::frame_index_comp(uint64_t offset) const { return frame_index_at(offset, true); }
uint32_t SeekTable::frame_index_decomp(uint64_t offset) const { return frame_index_at(offset, false); }

#define ZK_CHECK_INDEX(i) do { if ((i) >= num_frames()) throw Error::frame_index_too_large(); } while (0)
uint64_t SeekTable::frame_start_comp(uint32_t i) const { ZK_CHECK_INDEX(i); return entries_[i].c_offset; }
uint64_t SeekTable::frame_start_decomp(uint32_t i) const { ZK_CHECK_INDEX(i); return entries_[i].d_offset; }
uint64_t SeekTable::frame_end_comp(uint32_t i) const { ZK_CHECK_INDEX(i); return entries_[i + 1].c_offset; }
uint64_t SeekTable::frame_end_decomp(uint32_t i) const { ZK_CHECK_INDEX(i); return entries_[i + 1].d_offset; }
uint64_t SeekTable::frame_size_comp(uint32_t i) const { ZK_CHECK_INDEX(i); return entries_[i + 1].c_offset - entries_[i].c_offset; }
uint64_t SeekTable::frame_size_decomp(uint32_t i) const { ZK_CHECK_INDEX(i); return entries_[i + 1].d_offset - entries_[i].d_offset; }
uint64_t SeekTable::max_frame_size_comp() const
{
    uint64_t m = 0;
    for (uint32_t i = 0; i < num_frames(); i++) m = std::max(m, entries_[i + 1].c_offset - entries_[i].c_offset);
    return m;
}
uint64_t SeekTable::max_frame_size_decomp() const
{
    uint64_t m = 0;
    for (uint32_t i = 0; i < num_frames(); i++) m = std::max(m, entries_[i + 1].d_offset - entries_[i].d_offset);
    return m;
}

Serializer SeekTable::into_serializer() const { return into_format_serializer(Format::Foot); }
Serializer SeekTable::into_format_serializer(Format format) const             // seek_table.rs:907-914 + into_frames :113-121
{
    Serializer s;
    s.format_ = format;
    s.frames_.reserve(num_frames());
    for (uint32_t i = 0; i < num_frames(); i++)
        s.frames_.push_back({(uint32_t)(entries_[i + 1].c_offset - entries_[i].c_offset),
                             (uint32_t)(entries_[i + 1].d_offset - entries_[i].d_offset)});
    return s;
}

// ---------------------------------------------------------------- Serializer (seek_table.rs:955-1059)
// The table is a function of the byte position: header words, the 9-byte integrity field (in front of the entries in the
// Head format, behind them in the Foot format) and 8 bytes per frame.  write_into fills the caller's buffer from the
// position it stopped at -- resumable at any byte -- and the only state is that position.
uint8_t Serializer::byte_at(size_t pos) const
{
    auto le = [](uint32_t v, size_t k) { return (uint8_t)(v >> (8 * k)); };
    const size_t n = frames_.size();
    if (pos < 4) return le(SKIPPABLE_MAGIC_NUMBER, pos);
    if (pos < 8) return le((uint32_t)(encoded_len() - SKIPPABLE_HEADER_SIZE), pos - 4);
    const size_t integrity_at = format_ == Format::Head ? SKIPPABLE_HEADER_SIZE : SKIPPABLE_HEADER_SIZE + 8 * n;
    if (pos >= integrity_at && pos < integrity_at + SEEK_TABLE_INTEGRITY_SIZE) {
        const size_t k = pos - integrity_at;                          // Number_Of_Frames, Seek_Table_Descriptor (0), Seekable_Magic_Number
        return k < 4 ? le((uint32_t)n, k) : k == 4 ? 0 : le(SEEKABLE_MAGIC_NUMBER, k - 5);
    }
    const size_t e = pos - SKIPPABLE_HEADER_SIZE - (format_ == Format::Head ? SEEK_TABLE_INTEGRITY_SIZE : 0);
    const Frame &f = frames_[e / 8];
    return le((e & 4) ? f.d_size : f.c_size, e & 3);
}

size_t Serializer::write_into(uint8_t *buf, size_t len)
{
    const size_t total = encoded_len();
    const size_t n = std::min(len, total - std::min(total, write_pos_));
    const size_t entries_at = SKIPPABLE_HEADER_SIZE + (format_ == Format::Head ? SEEK_TABLE_INTEGRITY_SIZE : 0), entries_end = entries_at + 8 * frames_.size();
    size_t i = 0;
    while (i < n) {
        const size_t pos = write_pos_ + i;
        if (pos >= entries_at && (pos - entries_at) % 8 == 0 && pos + 8 <= entries_end && i + 8 <= n) {      // whole entries: eight bytes at a time
            const Frame &f = frames_[(pos - entries_at) / 8];
            const uint8_t e[8] = {(uint8_t)f.c_size, (uint8_t)(f.c_size >> 8), (uint8_t)(f.c_size >> 16), (uint8_t)(f.c_size >> 24),
                                  (uint8_t)f.d_size, (uint8_t)(f.d_size >> 8), (uint8_t)(f.d_size >> 16), (uint8_t)(f.d_size >> 24)};
            memcpy(buf + i, e, 8);
            i += 8;
        } else { buf[i] = byte_at(pos); i++; }
    }
    write_pos_ += n;
    return n;
}

}  // namespace zeekstd
