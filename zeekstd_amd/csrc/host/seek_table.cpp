// seek_table.cpp -- Zstandard Seekable Format seek table: in-memory index, resumable serializer,
// streaming parser.  Host-side integer code; mirrors /root/reference/lib/src/seek_table.rs
// (spec: /root/reference/seekable_format.md:45-157) behaviour for behaviour, including its quirks.
#include <string.h>
#include <algorithm>
#include "zeekstd.hpp"

namespace zeekstd {

namespace {
constexpr size_t SIZE_PER_FRAME = 8;                                  // seek_table.rs:87
constexpr uint32_t ZSTD_error_prefix_unknown = 10, ZSTD_error_corruption_detected = 20;

inline uint32_t read_le32(const uint8_t *b, size_t off)              // macro read_le32!, seek_table.rs:14-21
{
    return (uint32_t)b[off] | ((uint32_t)b[off + 1] << 8) | ((uint32_t)b[off + 2] << 16) | ((uint32_t)b[off + 3] << 24);
}
}  // namespace

// ---------------------------------------------------------------- Parser (seek_table.rs:134-225)
class SeekTableParser {
public:
    size_t num_frames = 0, size_per_frame = 8, seek_table_size = 0;
    std::vector<SeekTable::Entry> entries;
    uint64_t c_offset = 0, d_offset = 0;

    static SeekTableParser from_bytes(const uint8_t *buf)            // 9-byte integrity field, :144-172
    {
        if (read_le32(buf, 5) != SEEKABLE_MAGIC_NUMBER) throw Error::zstd(ZSTD_error_prefix_unknown);
        if (((buf[4] >> 2) & 0x1f) > 0) throw Error::zstd(ZSTD_error_corruption_detected);   // reserved descriptor bits
        const bool with_checksum = (buf[4] & (1 << 7)) > 0;
        const uint32_t n = read_le32(buf, 0);
        if (n > SEEKABLE_MAX_FRAMES) throw Error::frame_index_too_large();
        SeekTableParser p;
        p.num_frames = n;
        p.size_per_frame = with_checksum ? 12 : 8;
        p.seek_table_size = p.num_frames * p.size_per_frame + SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE;
        p.entries.reserve(std::max<size_t>(p.num_frames, 1) + 1);
        return p;
    }
    void verify_skippable_header(const uint8_t *buf) const           // :174-184
    {
        if (read_le32(buf, 0) != SKIPPABLE_MAGIC_NUMBER) throw Error::zstd(ZSTD_error_prefix_unknown);
        const size_t size = read_le32(buf, 4);
        if (size + SKIPPABLE_HEADER_SIZE != seek_table_size) throw Error::zstd(ZSTD_error_corruption_detected);
    }
    size_t parse_entries(const uint8_t *buf, size_t len)             // :186-209
    {
        size_t pos = 0;
        while (entries.size() < num_frames) {
            if (pos + size_per_frame > len) return pos;
            log_entry();
            c_offset += read_le32(buf, pos);
            d_offset += read_le32(buf, pos + 4);
            pos += size_per_frame;                                    // a legacy per-frame checksum is skipped, not verified
        }
        log_entry();                                                  // final entry: end of the last frame
        return pos;
    }
    void log_entry() { entries.push_back({c_offset, d_offset}); }
    void verify() const                                              // :218-224
    {
        if (entries.size() != num_frames + 1) throw Error::zstd(ZSTD_error_corruption_detected);
    }
    SeekTable into_table()
    {
        SeekTable t;
        t.entries_ = std::move(entries);
        return t;
    }
};

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
    auto integrity = src.seek_table_integrity(format);
    SeekTableParser parser = SeekTableParser::from_bytes(integrity.data());
    if (format == Format::Head) src.set_offset(OffsetFrom::Start(0));
    else src.set_offset(OffsetFrom::End(-(int64_t)parser.seek_table_size));

    const size_t len = std::min<size_t>(8192, parser.seek_table_size);
    std::vector<uint8_t> buf(len);
    size_t read = 0;
    while (read < SKIPPABLE_HEADER_SIZE) {
        // NB the reference reads into the start of buf on every iteration (seek_table.rs:391-399); with a
        // source that returns fewer than 8 bytes at a time the header check would see a torn header.
        // Reading at buf+read keeps the bytes it already has; identical for any source that delivers >= 8 bytes.
        size_t n = src.read(buf.data() + read, buf.size() - read);
        if (n == 0) throw Error::zstd(ZSTD_error_corruption_detected);
        read += n;
    }
    parser.verify_skippable_header(buf.data());

    size_t buf_start = SKIPPABLE_HEADER_SIZE;
    if (format == Format::Head) buf_start += SEEK_TABLE_INTEGRITY_SIZE;
    size_t remaining = parser.seek_table_size - SKIPPABLE_HEADER_SIZE - SEEK_TABLE_INTEGRITY_SIZE;
    size_t buf_end = read;
    if (buf_start > buf_end) {                                        // Head format with a short first read
        while (buf_end < buf_start) {
            size_t n = src.read(buf.data() + buf_end, buf.size() - buf_end);
            if (n == 0) throw Error::zstd(ZSTD_error_corruption_detected);
            buf_end += n;
        }
    }
    for (;;) {
        size_t n = parser.parse_entries(buf.data() + buf_start, buf_end - buf_start);
        remaining -= n;
        if (remaining == 0) break;
        // move the unparsed tail (a partial entry) to the front, then read more
        size_t offset = buf_end - (buf_start + n);
        memmove(buf.data(), buf.data() + buf_start + n, offset);
        size_t m = src.read(buf.data() + offset, buf.size() - offset);
        if (remaining > 0 && m == 0) throw Error::zstd(ZSTD_error_corruption_detected);
        buf_start = 0;
        buf_end = offset + m;
    }
    parser.verify();
    return parser.into_table();
}

SeekTable SeekTable::from_reader(Reader &reader)                              // seek_table.rs:461-493 (Head format)
{
    uint8_t head[SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE];
    size_t got = 0;
    while (got < sizeof head) {                                       // read_exact
        size_t n = reader.read(head + got, sizeof head - got);
        if (n == 0) throw Error::io("failed to fill whole buffer");
        got += n;
    }
    SeekTableParser parser = SeekTableParser::from_bytes(head + SKIPPABLE_HEADER_SIZE);
    parser.verify_skippable_header(head);
    size_t remaining = parser.seek_table_size - SKIPPABLE_HEADER_SIZE - SEEK_TABLE_INTEGRITY_SIZE;
    std::vector<uint8_t> buf(std::min<size_t>(8192, remaining));
    size_t have = 0;                                                  // bytes valid at the start of buf
    size_t unread = remaining;                                        // bytes of the table not yet pulled from the reader
    for (;;) {
        if (unread > 0) {
            // short reads are fine (regression fixed upstream: CHANGELOG_LIB.md:14-15)
            size_t want = std::min(buf.size() - have, unread);
            size_t n = reader.read(buf.data() + have, want);
            if (n == 0) throw Error::zstd(ZSTD_error_corruption_detected);
            have += n; unread -= n;
        }
        size_t n = parser.parse_entries(buf.data(), have);
        remaining -= n;
        if (remaining == 0) break;
        memmove(buf.data(), buf.data() + n, have - n);
        have -= n;
    }
    parser.verify();
    return parser.into_table();
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

// ---------------------------------------------------------------- Serializer (seek_table.rs:23-84, 955-1059)
// Resumable at byte granularity: write_pos_ counts the bytes of the table emitted so far; every field
// emits only its not-yet-written bytes and returns as soon as the caller's buffer is full.
size_t Serializer::write_into(uint8_t *buf, size_t len)
{
    size_t buf_pos = 0;
    bool full = false;
    auto write_le32 = [&](uint32_t value, size_t offset) {            // macro write_le32!, :23-43
        if (full) return;
        if (write_pos_ < offset + 4) {
            size_t n = std::min(len - buf_pos, offset + 4 - write_pos_);
            size_t val_offset = write_pos_ - offset;
            uint8_t le[4] = {(uint8_t)value, (uint8_t)(value >> 8), (uint8_t)(value >> 16), (uint8_t)(value >> 24)};
            memcpy(buf + buf_pos, le + val_offset, n);
            buf_pos += n; write_pos_ += n;
            if (buf_pos == len) full = true;
        }
    };
    auto write_integrity = [&](uint32_t num_frames, size_t offset) {  // macro write_integrity!, :67-84
        write_le32(num_frames, offset);
        if (full) return;
        if (write_pos_ < offset + 5) {                                // Seek_Table_Descriptor, always 0
            // (upstream writes this byte unconditionally; it cannot overflow there because write_le32
            //  returns when the buffer is full -- same here through `full`)
            buf[buf_pos] = 0; buf_pos += 1; write_pos_ += 1;
            if (buf_pos == len) { full = true; }
        }
        // upstream does not test "buffer full" after the descriptor byte: the next write_le32 sees a
        // zero-length remainder, copies nothing and returns buf_pos -- equivalent
        write_le32(SEEKABLE_MAGIC_NUMBER, offset + 5);
    };
    if (len == 0) return 0;
    write_le32(SKIPPABLE_MAGIC_NUMBER, 0);
    write_le32((uint32_t)(encoded_len() - SKIPPABLE_HEADER_SIZE), 4);
    if (format_ == Format::Head) write_integrity((uint32_t)frames_.size(), SKIPPABLE_HEADER_SIZE);
    while (!full && frame_index_ < frames_.size()) {
        size_t offset = SKIPPABLE_HEADER_SIZE + SIZE_PER_FRAME * frame_index_;
        if (format_ == Format::Head) offset += SEEK_TABLE_INTEGRITY_SIZE;
        write_le32(frames_[frame_index_].c_size, offset);
        write_le32(frames_[frame_index_].d_size, offset + 4);
        if (!full || write_pos_ >= offset + 8) {
            // the frame is complete once its 8 bytes are out (upstream advances frame_index only when both
            // writes went through without an early return; a full buffer exactly at the frame end is picked
            // up on the next call by the write_pos test)
            if (write_pos_ >= offset + 8) frame_index_ += 1;
        }
    }
    if (!full && format_ == Format::Foot) {
        size_t offset = SKIPPABLE_HEADER_SIZE + SIZE_PER_FRAME * frames_.size();
        write_integrity((uint32_t)frames_.size(), offset);
    }
    return buf_pos;
}

}  // namespace zeekstd
