#!/usr/bin/env python3
"""Mint tests/golden/handmade.json: frames written by hand for corners of the format that libzstd's encoder never
produced for the archives of tools/make_goldens.py, each accepted by the real libzstd **1.5.7** (and the distro build)
before it is stored -- the decoder of the pinned library is the judge of what the bytes mean, as for every other golden.

  rle_seq_tables   Sequences_Section with RLE_Mode for all three tables (Symbol_Compression_Modes 0x54): the tables have
                   one cell and accuracy log 0, so the bitstream carries no state bits at all (RFC 8878 3.1.1.3.2.1)
  rle_ll_ml_predef_of  RLE_Mode for LL / ML next to Predefined_Mode offsets (modes 0x44)
  rep_across_blocks_ll0  the repeat-offset history crosses a block boundary and is used with Literals_Length 0 and
                   Offset_Value 3 ("Repeated_Offset1 - 1 byte", RFC 8878 3.1.1.5)
  rep0_minus_1_is_zero  the same with Repeated_Offset1 == 1: the offset would be 0 -- libzstd rejects the frame
                   (corruption_detected) and so must every decoder here            [stored with "error": true]

Run in the BUILD container only.  Output: tests/golden/handmade.json ({name: {frame: hex, output: hex, note}}).
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import libzstd_ref as Z


def frame(content_size, *blocks, fhd_extra=0):
    """blocks: bytes = payload of a Compressed_Block, ("raw", bytes) = Raw_Block.  Single_Segment frame."""
    if content_size < 256:
        out = bytes.fromhex("28B52FFD") + bytes([0x20 | fhd_extra, content_size])          # 1-byte Frame_Content_Size
    else:
        out = bytes.fromhex("28B52FFD") + bytes([0xA0 | fhd_extra]) + content_size.to_bytes(4, "little")     # 4-byte
    for i, p in enumerate(blocks):
        last = 1 if i + 1 == len(blocks) else 0
        if isinstance(p, tuple):
            out += ((len(p[1]) << 3) | (0 << 1) | last).to_bytes(3, "little") + p[1]
        else:
            out += ((len(p) << 3) | (2 << 1) | last).to_bytes(3, "little") + p
    return out


def raw_literals(b):
    assert len(b) < 32
    return bytes([len(b) << 3]) + b


CASES = {}
# three sequences, each: 2 literals, then a match of length 4 at "repeat offset 1" (initially 1) -> the last literal four times
CASES["rle_seq_tables"] = dict(
    frame=frame(18, raw_literals(b"abcdef") + bytes([3, 0x54, 2, 0, 1]) + b"\x01"),
    output=b"abbbbb" + b"cddddd" + b"efffff",
    note="LL code 2, OF code 0, ML code 1, all RLE_Mode; bitstream = the padding marker only")
# LL / ML RLE, offsets through the predefined table (accuracy log 5): two sequences with Offset_Value 1 (code 0, state 0
# of the predefined OF table decodes symbol 0) -- initial OF state 0 (5 bits), update after sequence 1: code 0 at state 0
# has nbBits 5 and baseline 0 -> 5 bits 00000 select state 0 again
def backward_bitstream(fields):
    """Bytes of a sequence bitstream whose reader meets `fields` [(value, width), ...] in this order: the first field
    sits right below the padding marker, the stream is little endian with the marker in its last byte."""
    acc, n = 1, 0
    for v, w in fields:
        acc = (acc << w) | v
        n += w
    return acc.to_bytes((n + 1 + 7) // 8, "little")


CASES["rle_ll_ml_predef_of"] = dict(
    frame=frame(12, raw_literals(b"wxyz") + bytes([2, 0x44, 2, 1]) + backward_bitstream([(0, 5), (0, 5)])),
    output=b"wxxxxx" + b"yzzzzz",
    note="LL code 2 / ML code 1 RLE_Mode, offsets Predefined_Mode: initial OF state 0 (5 bits), one 5-bit state update")

# block 1: "abcd", match 4 @ offset 3 (Offset_Value 6 = code 2 + extra 0b10)          -> abcdbcdb, history {3, 1, 4}
# block 2: no literals, Literals_Length 0, match 3, Offset_Value 3 (code 1 + extra 1) -> offset = 3 - 1 = 2 -> dbd
CASES["rep_across_blocks_ll0"] = dict(
    frame=frame(11, raw_literals(b"abcd") + bytes([1, 0x54, 4, 2, 1]) + backward_bitstream([(2, 2)]),
                raw_literals(b"") + bytes([1, 0x54, 0, 1, 0]) + backward_bitstream([(1, 1)])),
    output=b"abcdbcdb" + b"dbd",
    note="two blocks, one RLE_Mode sequence each; block 2 uses Repeated_Offset1 - 1 with Literals_Length 0")
# block 1: match 4 @ offset 1 (Offset_Value 4 = code 2 + extra 0b00) -> history {1, 1, 4}; block 2 as above -> offset 0
CASES["rep0_minus_1_is_zero"] = dict(
    frame=frame(11, raw_literals(b"abcd") + bytes([1, 0x54, 4, 2, 1]) + backward_bitstream([(0, 2)]),
                raw_literals(b"") + bytes([1, 0x54, 0, 1, 0]) + backward_bitstream([(1, 1)])),
    output=None,
    note="Repeated_Offset1 - 1 == 0: corruption_detected (20) in libzstd 1.5.7")

# a compressed block without sequences (Number_of_Sequences 0: the block is its literals)
CASES["block_without_sequences"] = dict(
    frame=frame(11, raw_literals(b"only") + bytes([0]), raw_literals(b"literal") + bytes([0])),
    output=b"onlyliteral",
    note="two Compressed_Blocks with Number_of_Sequences == 0")
# 0x7F00 sequences in one block (the 3-byte Number_of_Sequences form), none of them with a single bit in the bitstream:
# Literals_Length 0, Match_Length 3, Offset_Value 1 == "Repeated_Offset2" after a literal-less sequence, so the history
# rotates between 4 and 1.  Whatever libzstd 1.5.7 makes of it is the expected output.
N_TINY = 0x7F00
CASES["many_tiny_sequences"] = dict(
    frame=frame(4 + 3 * N_TINY, ("raw", b"wxyz"),
                raw_literals(b"") + bytes([255]) + (N_TINY - 0x7F00).to_bytes(2, "little") + bytes([0x54, 0, 0, 0]) + b"\x01"),
    output="libzstd",
    note="Raw_Block 'wxyz', then 32512 RLE_Mode sequences (ll 0, ml 3, Offset_Value 1) and a 1-byte bitstream")
# Reserved_bit of the Frame_Header_Descriptor set
CASES["reserved_bit"] = dict(
    frame=frame(18, raw_literals(b"abcdef") + bytes([3, 0x54, 2, 0, 1]) + b"\x01", fhd_extra=0x08),
    output=None, error_code=14, error_match="nsupported",
    note="Frame_Header_Descriptor with the Reserved_bit set: frameParameter_unsupported (14)")

out = {}
for name, c in CASES.items():
    if c["output"] == "libzstd":
        n = int.from_bytes(c["frame"][5:9], "little")
        c["output"] = Z.decode_stream(c["frame"], n, "1.5.7")
        assert len(c["output"]) == n
    if c["output"] is None:
        # the pinned library decides: 1.5.7 makes the offset invalid on purpose ("temp -= !temp") and fails in
        # ZSTD_execSequence; libzstd 1.4.8 still forced such an offset to 1 and went on
        assert Z.load("1.5.7") is not None, "libzstd 1.5.7 (pillow bundled) not found in this image"
        try:
            Z.decode_stream(c["frame"], c["frame"][5], "1.5.7")
        except Exception as ex:
            assert c.get("error_match", "orrupt") in str(ex), (name, ex)
        else:
            raise AssertionError((name, "libzstd 1.5.7 accepted the frame"))
        out[name] = {"frame": c["frame"].hex(), "error": True, "error_code": c.get("error_code", 20),
                     "content_size": c["frame"][5], "note": c["note"],
                     "rejected_by": "libzstd 1.5.7" + ("; 1.4.8 forced the offset to 1 instead" if name == "rep0_minus_1_is_zero" else "")}
        continue
    for which in ("1.5.7", "system"):
        if Z.load(which) is None:
            assert which != "1.5.7", "libzstd 1.5.7 (pillow bundled) not found in this image"
            continue
        got = Z.decode_stream(c["frame"], len(c["output"]), which)
        assert got == c["output"], (name, which, got)
    out[name] = {"frame": c["frame"].hex(), "output": c["output"].hex() if len(c["output"]) < 4096 else None,
                 "output_zlib": None if len(c["output"]) < 4096 else __import__("base64").b64encode(__import__("zlib").compress(c["output"], 9)).decode(),
                 "note": c["note"],
                 "accepted_by": "libzstd 1.5.7 and the distro libzstd of the build container (ZSTD_decompressStream)"}
# ---- damaged variants of rle_seq_tables: each must be corruption_detected (20) in libzstd 1.5.7 (one-shot ZSTD_decompress,
#      code read with ZSTD_getErrorCode).  Checked on the CPU only (oracle + simulation of the device code): "cpu_only".
import ctypes as C
_l = Z.load("1.5.7")
_l.ZSTD_decompress.restype = C.c_size_t
_l.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
_l.ZSTD_getErrorCode.restype = C.c_int
_l.ZSTD_getErrorCode.argtypes = [C.c_size_t]


def verdict(fr):
    buf = C.create_string_buffer(1 << 16)
    r = _l.ZSTD_decompress(buf, len(buf), fr, len(fr))
    return _l.ZSTD_getErrorCode(r) if _l.ZSTD_isError(r) else 0


SEQ3 = bytes([3, 0x54, 2, 0, 1]) + b"\x01"
LIT6 = raw_literals(b"abcdef")
NEG = {
    "fcs_too_big": frame(19, LIT6 + SEQ3),
    "fcs_too_small": frame(17, LIT6 + SEQ3),
    "literals_exhausted": frame(18, raw_literals(b"abc") + SEQ3),
    "bitstream_not_consumed": frame(18, LIT6 + bytes([3, 0x54, 2, 0, 1]) + b"\x00\x01"),
    "bitstream_zero_last_byte": frame(18, LIT6 + bytes([3, 0x54, 2, 0, 1]) + b"\x00"),
    "rle_symbol_out_of_range": frame(18, LIT6 + bytes([3, 0x54, 36, 0, 1]) + b"\x01"),
    "modes_reserved_bits": frame(18, LIT6 + bytes([3, 0x55, 2, 0, 1]) + b"\x01"),
}
_f = bytearray(frame(18, LIT6 + SEQ3)); _f[6] |= 0x06
NEG["block_type_reserved"] = bytes(_f)
for name, fr in NEG.items():
    assert verdict(fr) == 20, (name, verdict(fr))
    out[name] = {"frame": fr.hex(), "error": True, "error_code": 20, "content_size": fr[5], "cpu_only": True,
                 "note": "damaged variant of rle_seq_tables", "rejected_by": "libzstd 1.5.7 (corruption_detected)"}
with open(os.path.join(ROOT, "tests", "golden", "handmade.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(out), "frames")
