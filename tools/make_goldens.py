#!/usr/bin/env python3
"""Mint the golden fixtures under tests/golden/ (run in the BUILD container only).

The compressed payloads are produced by the real libzstd **1.5.7** (the version the reference pins,
Cargo.lock:1192-1193) driven with the reference's exact call sequence and buffer sizes
(oracle/libzstd_ref.py: lib/src/encode.rs:340-346, 442-464).  Inputs are NOT stored: each fixture
names a deterministic recipe (oracle/zko.make_input) plus the XXH64 of the input, so a fixture is
"input recipe -> expected compressed frames (c,d sizes) -> expected decoded bytes".

Output: tests/golden/archives.json (index) + tests/golden/archives.bin (concatenated payloads).
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zko, libzstd_ref as Z

assert Z.version("1.5.7") == "1.5.7", "libzstd 1.5.7 (pillow bundled) not found in this image"

K = 1024
CASES = [
    # name, recipe, frame_size, level, checksum
    ("empty", [], 2 << 20, 1, False),
    ("empty_cks", [], 2 << 20, 1, True),
    ("hello", [["rep", b"Hello, World!".hex(), 1]], 2 << 20, 1, False),
    ("hello_cks", [["rep", b"Hello, World!".hex(), 1]], 2 << 20, 1, True),
    ("hello_default_level", [["rep", b"Hello, World!".hex(), 1]], 2 << 20, 0, False),
    ("text_l1_64k", [["text", 200 * K, 11]], 64 * K, 1, True),
    ("text_l1_one_frame", [["text", 300 * K, 12]], 2 << 20, 1, True),          # 3 blocks: huf + treeless
    ("text_l3", [["text", 200 * K, 13]], 2 << 20, 3, False),
    ("text_l5", [["text", 150 * K, 14]], 2 << 20, 5, True),
    ("text_l9", [["text", 150 * K, 15]], 2 << 20, 9, False),
    ("text_l19", [["text", 150 * K, 16]], 2 << 20, 19, True),
    ("text_neg5", [["text", 150 * K, 17]], 2 << 20, -5, False),
    ("text_100B_frames", [["text", 6000, 18]], 100, 1, False),                   # raw literals, huf_1s, predefined/rle modes
    ("text_1000B_frames", [["text", 20000, 19]], 1000, 3, True),
    ("text_len_div7", [["text", 7 * 3001, 20]], 3001, 1, True),
    ("zeros", [["zeros", 300 * K]], 2 << 20, 1, True),                           # RLE blocks + 1 sequence
    ("random", [["random", 200 * K, 21]], 2 << 20, 1, True),                     # raw blocks
    ("mixed", [["random", 30 * K, 22], ["text", 120 * K, 23], ["zeros", 40 * K], ["rep", "616263", 20000],
               ["text", 50 * K, 23], ["rep", "6162", 3000], ["rep", "78", 70000]], 2 << 20, 1, True),
    ("mixed_l19", [["random", 10 * K, 24], ["text", 100 * K, 25], ["rep", "0102030405", 9000], ["text", 60 * K, 25]],
     2 << 20, 19, False),
    ("long_offsets_l19", [["text", 180 * K, 26], ["random", 150 * K, 27], ["text", 180 * K, 26]], 2 << 20, 19, True),
    ("tiny_frames_10B", [["text", 500, 28]], 10, 1, False),
    ("records_of_rle", [["records", 6000, 99, b"0123456789abcdefghij".hex()]], 2 << 20, 1, True),   # OF table in RLE mode
    ("slices_l19", [["slices", 100000, 77, 9000, 5, 8, 40, "ff"]], 2 << 20, 19, True),              # RLE literals, Repeat_Mode x3
    ("one_byte", [["rep", "41", 1]], 2 << 20, 1, True),
]
# frames written by ONE ZSTD_compress2 call each (what `zstd` / ZSTD_compress write): Frame_Content_Size in the header and,
# for small frames, Single_Segment -- zeekstd's own streaming frames carry neither, other writers of seekable archives do
ONESHOT = [
    ("oneshot_small", [["text", 300, 31]], 100, 3, False),                        # single-segment, 1-byte FCS
    ("oneshot_text_l3", [["text", 150 * K, 32]], 50 * K, 3, True),                # 2-byte FCS (+256), three seek entries
    ("oneshot_l19_4byte_fcs", [["text", 90 * K, 33], ["rep", "6162636465", 4000]], 2 << 20, 19, True),
    ("oneshot_empty", [], 2 << 20, 3, False),
]

blob = bytearray()
index = []
for name, recipe, fs, level, cks in CASES + ONESHOT:
    data = zko.make_input(recipe)
    if name.startswith("oneshot"):
        comp, frames = Z.encode_oneshot_frames(data, fs, level, cks, "1.5.7")
    else:
        comp, frames = Z.encode_seekable_frames(data, fs, level, cks, "1.5.7")
    # self-check with BOTH real decoders before committing
    assert Z.decode_stream(comp, len(data), "1.5.7") == data
    assert Z.decode_stream(comp, len(data), "system") == data
    index.append({"name": name, "recipe": recipe, "frame_size": fs, "level": level, "checksum": cks,
                  "input_len": len(data), "input_xxh64": f"{zko.xxh64(data):016x}",
                  "frames": frames, "offset": len(blob), "length": len(comp),
                  "comp_xxh64": f"{zko.xxh64(comp):016x}"})
    blob += comp
    print(f"{name:24s} in={len(data):8d} comp={len(comp):8d} frames={len(frames)}")

os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
with open(os.path.join(ROOT, "tests", "golden", "archives.bin"), "wb") as f:
    f.write(blob)
with open(os.path.join(ROOT, "tests", "golden", "archives.json"), "w") as f:
    json.dump({"libzstd": "1.5.7", "generator": "tools/make_goldens.py", "cases": index}, f, indent=1)
print("total payload bytes:", len(blob))
