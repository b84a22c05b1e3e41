// seekable.cpp -- Error + the Seekable sources (mirrors /root/reference/lib/src/error.rs, seekable.rs)
#include <errno.h>
#include <string.h>
#include "../../../include/zeekstd_amd.h"
#include "zeekstd.hpp"

namespace zeekstd {

// ---------------------------------------------------------------- Error
Error::Error(Kind k, size_t code, const std::string &detail) : kind_(k), code_(code)
{
    switch (k) {                                                      // Display impl, error.rs:60-71
    case Kind::NumberConversionFailed: msg_ = "number conversion failed: " + detail; break;
    case Kind::OffsetOutOfRange: msg_ = "offset out of range"; break;
    case Kind::FrameIndexTooLarge: msg_ = "frame index too large"; break;
    case Kind::IO: msg_ = "io error: " + detail; break;
    case Kind::Zstd: msg_ = zk_error_name(-(int)(uint32_t)((size_t)0 - code)); if (!detail.empty()) msg_ += " (" + detail + ")"; break;
    }
}

int Error::abi_code() const
{
    switch (kind_) {
    case Kind::NumberConversionFailed: return ZK_ERR_NUMBER_CONVERSION;
    case Kind::OffsetOutOfRange: return ZK_ERR_OFFSET_OUT_OF_RANGE;
    case Kind::FrameIndexTooLarge: return ZK_ERR_FRAME_INDEX_TOO_LARGE;
    case Kind::IO: return ZK_ERR_IO;
    case Kind::Zstd: return -(int)(uint32_t)((size_t)0 - code_);
    }
    return -1;
}

Error Error::from_engine_code(int rc, const std::string &detail)
{
    if (rc == ZK_ERR_OFFSET_OUT_OF_RANGE) return offset_out_of_range();
    if (rc == ZK_ERR_FRAME_INDEX_TOO_LARGE) return frame_index_too_large();
    if (rc == ZK_ERR_NUMBER_CONVERSION) return number_conversion_failed(detail);
    if (rc <= -1000) return io(std::string(zk_error_name(rc)) + (detail.empty() ? "" : ": " + detail));   // HIP / device failures surface as IO
    return Error(Kind::Zstd, (size_t)0 - (size_t)(uint32_t)(-rc), detail);
}

// ---------------------------------------------------------------- BytesWrapper (seekable.rs:55-97)
uint64_t BytesWrapper::set_offset(OffsetFrom offset)
{
    size_t pos;
    if (offset.from == OffsetFrom::From::Start) {
        pos = (size_t)(uint64_t)offset.value;
    } else {
        int64_t d = offset.value;                                     // len.checked_add_signed(d)
        if (d < 0) { if ((uint64_t)(-d) > len_) throw Error::offset_out_of_range(); pos = len_ - (size_t)(-d); }
        else { if ((uint64_t)d > SIZE_MAX - len_) throw Error::offset_out_of_range(); pos = len_ + (size_t)d; }
    }
    if (pos > len_) throw Error::offset_out_of_range();
    pos_ = pos;
    return pos;
}

size_t BytesWrapper::read(uint8_t *buf, size_t len)
{
    size_t n = len < len_ - pos_ ? len : len_ - pos_;
    memcpy(buf, src_ + pos_, n);
    pos_ += n;
    return n;
}

std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> BytesWrapper::seek_table_integrity(Format format)
{
    size_t off;
    if (format == Format::Head) {
        if (len_ < SKIPPABLE_HEADER_SIZE + SEEK_TABLE_INTEGRITY_SIZE) throw Error::offset_out_of_range();
        off = SKIPPABLE_HEADER_SIZE;
    } else {
        if (len_ < SEEK_TABLE_INTEGRITY_SIZE) throw Error::offset_out_of_range();
        off = len_ - SEEK_TABLE_INTEGRITY_SIZE;                       // last 9 bytes
    }
    std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> a;
    memcpy(a.data(), src_ + off, SEEK_TABLE_INTEGRITY_SIZE);
    return a;
}

// ---------------------------------------------------------------- CallbackSeekable (the trait itself, seekable.rs:16-39)
uint64_t CallbackSeekable::set_offset(OffsetFrom offset)
{
    const int64_t r = so_(user_, offset.from == OffsetFrom::From::Start ? 0 : 1, offset.value);
    if (r < 0) throw Error::io("set_offset callback failed");
    return (uint64_t)r;
}

size_t CallbackSeekable::read(uint8_t *buf, size_t len)
{
    const int64_t r = rd_(user_, buf, len);
    if (r < 0 || (uint64_t)r > len) throw Error::io("read callback failed");
    return (size_t)r;
}

// Seekable::seek_table_integrity is a REQUIRED method of the trait (seekable.rs:33-38): a host source may keep its
// integrity field anywhere, so its own callback is asked first.  Without one the field is read the way both of the
// reference's impls read it (seekable.rs:84-96, 126-137): position at the field, read_exact 9 bytes.
std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> CallbackSeekable::seek_table_integrity(Format format)
{
    if (ig_) {
        std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> a;
        if (ig_(user_, format == Format::Head ? 0 : 1, a.data()) < 0) throw Error::io("seek_table_integrity callback failed");
        return a;
    }
    if (format == Format::Head) set_offset(OffsetFrom::Start(SKIPPABLE_HEADER_SIZE));
    else set_offset(OffsetFrom::End(-(int64_t)SEEK_TABLE_INTEGRITY_SIZE));
    std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> a;
    size_t got = 0;
    while (got < a.size()) {
        size_t n = read(a.data() + got, a.size() - got);
        if (n == 0) throw Error::io("failed to fill whole buffer");
        got += n;
    }
    return a;
}

// ---------------------------------------------------------------- FileSeekable (seekable.rs:112-138)
uint64_t FileSeekable::set_offset(OffsetFrom offset)
{
    int whence = offset.from == OffsetFrom::From::Start ? SEEK_SET : SEEK_END;
    if (fseeko(f_, (off_t)offset.value, whence) != 0) throw Error::io(strerror(errno));
    return (uint64_t)ftello(f_);
}

size_t FileSeekable::read(uint8_t *buf, size_t len)
{
    size_t n = fread(buf, 1, len, f_);
    if (n == 0 && ferror(f_)) throw Error::io(strerror(errno));
    return n;
}

std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> FileSeekable::seek_table_integrity(Format format)
{
    if (format == Format::Head) set_offset(OffsetFrom::Start(SKIPPABLE_HEADER_SIZE));
    else set_offset(OffsetFrom::End(-(int64_t)SEEK_TABLE_INTEGRITY_SIZE));
    std::array<uint8_t, SEEK_TABLE_INTEGRITY_SIZE> a;
    size_t got = 0;
    while (got < a.size()) {                                          // read_exact
        size_t n = read(a.data() + got, a.size() - got);
        if (n == 0) throw Error::io("failed to fill whole buffer");
        got += n;
    }
    return a;
}

}  // namespace zeekstd
