"""TEST INFRASTRUCTURE -- numpy/pure-Python restatement of the seek-table wire format
(/root/reference/seekable_format.md:45-157; lib/src/seek_table.rs:967-1005 serializer,
:134-225/:379-436 parser).  Checker for zeekstd_amd's C++ SeekTable; never used by the product.
Pinned by the golden bytes of SURVEY.md Appendix B (doctests seek_table.rs:872-880, 895-904)."""
import struct

import numpy as np

SKIPPABLE_MAGIC = 0x184D2A5E
SEEKABLE_MAGIC = 0x8F92EAB1
MAX_FRAMES = 0x08000000


class SeekTableError(Exception):
    def __init__(self, kind):
        super().__init__(kind)
        self.kind = kind


def serialize(frames, fmt="foot", with_checksum=False, checksums=None):
    """frames: [(c_size, d_size)].  with_checksum writes the legacy 12-byte entries + descriptor bit 7."""
    n = len(frames)
    per = 12 if with_checksum else 8
    entries = b"".join(struct.pack("<II", c, d) + (struct.pack("<I", (checksums or [0] * n)[i]) if with_checksum else b"")
                       for i, (c, d) in enumerate(frames))
    integrity = struct.pack("<IBI", n, 0x80 if with_checksum else 0, SEEKABLE_MAGIC)
    head = struct.pack("<II", SKIPPABLE_MAGIC, per * n + 9)
    return head + (integrity + entries if fmt == "head" else entries + integrity)


def parse(buf, fmt="foot"):
    """Returns (c_off, d_off) prefix-sum arrays (n+1 each)."""
    buf = bytes(buf)
    if fmt == "foot":
        if len(buf) < 9:
            raise SeekTableError("offset_out_of_range")
        integ = buf[-9:]
    else:
        if len(buf) < 17:
            raise SeekTableError("offset_out_of_range")
        integ = buf[8:17]
    n, desc, magic = struct.unpack("<IBI", integ)
    if magic != SEEKABLE_MAGIC:
        raise SeekTableError("prefix_unknown")
    if (desc >> 2) & 0x1F:
        raise SeekTableError("corruption_detected")
    if n > MAX_FRAMES:
        raise SeekTableError("frame_index_too_large")
    per = 12 if desc & 0x80 else 8
    size = per * n + 17
    if len(buf) < size:
        raise SeekTableError("offset_out_of_range")
    tab = buf[len(buf) - size:] if fmt == "foot" else buf[:size]
    m, fsz = struct.unpack("<II", tab[:8])
    if m != SKIPPABLE_MAGIC:
        raise SeekTableError("prefix_unknown")
    if fsz + 8 != size:
        raise SeekTableError("corruption_detected")
    body = tab[8:8 + per * n] if fmt == "foot" else tab[17:17 + per * n]
    a = np.frombuffer(body, dtype="<u4").reshape(n, per // 4) if n else np.zeros((0, 2), np.uint32)
    c = np.zeros(n + 1, np.uint64)
    d = np.zeros(n + 1, np.uint64)
    c[1:] = np.cumsum(a[:, 0].astype(np.uint64))
    d[1:] = np.cumsum(a[:, 1].astype(np.uint64))
    return c, d


def frame_index(off_arr, offset):
    """seek_table.rs:916-934 semantics: i with off[i] <= offset < off[i+1]; offset >= total -> n-1."""
    n = len(off_arr) - 1
    if offset >= int(off_arr[n]):
        return n - 1
    lo, hi = 0, n
    while lo + 1 < hi:
        mid = (lo + hi) // 2
        if int(off_arr[mid]) <= offset:
            lo = mid
        else:
            hi = mid
    return lo
