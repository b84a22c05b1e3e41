#!/usr/bin/env python3
"""Mint the prefix / patch-mode golden fixtures under tests/golden/ (run in the BUILD container only).

Same scheme as tools/make_goldens.py: payloads come from the real libzstd **1.5.7** driven with the reference's call
sequence -- here with ZSTD_CCtx_refPrefix at the start of every frame (lib/src/encode.rs:334-338) and, for the
patch-shaped cases, the window / long-distance-matching parameters zeekstd's CLI sets (cli/src/compress.rs:31-37).
Inputs and prefixes are named by deterministic recipes (oracle/zko.make_input), not stored.

Output: tests/golden/prefix_archives.json + tests/golden/prefix_archives.bin.
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zko, libzstd_ref as Z

assert Z.version("1.5.7") == "1.5.7", "libzstd 1.5.7 (pillow bundled) not found in this image"
K = 1024
CASES = [
    # name, prefix recipe, data recipe, frame_size, level, checksum, window_log, ldm
    ("pfx_text_l1", [["text", 100 * K, 41]], [["text", 40 * K, 41], ["text", 60 * K, 42], ["text", 30 * K, 41]], 32 * K, 1, True, 0, False),
    ("pfx_text_l3_big_prefix", [["text", 300 * K, 43]], [["text", 150 * K, 43], ["text", 50 * K, 44]], 64 * K, 3, False, 0, False),
    ("pfx_l19_one_frame", [["text", 64 * K, 45]], [["text", 30 * K, 46], ["text", 64 * K, 45], ["text", 100 * K, 46]], 2 << 20, 19, True, 0, False),
    ("pfx_patch_1m_ldm", [["text", 1024 * K, 47]],
     [["text", 300 * K, 47], ["rep", "5041544348", 40], ["text", 700 * K, 47]], 256 * K, 1, True, 21, True),   # new = old with an insertion
    ("pfx_tiny", [["rep", "68656c6c6f20776f726c6421", 1]], [["rep", "68656c6c6f20776f726c6421", 9]], 2 << 20, 1, True, 0, False),
    ("pfx_random_data", [["text", 50 * K, 48]], [["random", 40 * K, 49], ["text", 20 * K, 48]], 16 * K, 1, True, 0, False),
    ("pfx_small_frames", [["text", 20 * K, 50]], [["text", 12 * K, 50]], 1000, 1, False, 0, False),
]


def encode(data, fs, level, cks, prefix, wlog, ldm):
    if not ldm:
        return Z.encode_seekable_frames(data, fs, level, cks, "1.5.7", prefix=prefix, window_log=wlog)
    l = Z.load("1.5.7")
    orig = l.ZSTD_CCtx_setParameter

    class Patched:                                   # one more parameter right after the context is configured
        def __call__(self, cctx, k, v):
            r = orig(cctx, k, v)
            if k == 101:
                Z._chk(l, orig(cctx, 160, 1))        # ZSTD_c_enableLongDistanceMatching, cli/src/compress.rs:36
            return r
    l.ZSTD_CCtx_setParameter = Patched()
    try:
        return Z.encode_seekable_frames(data, fs, level, cks, "1.5.7", prefix=prefix, window_log=wlog)
    finally:
        l.ZSTD_CCtx_setParameter = orig


blob = bytearray()
index = []
for name, pre_recipe, recipe, fs, level, cks, wlog, ldm in CASES:
    prefix = zko.make_input(pre_recipe)
    data = zko.make_input(recipe)
    comp, frames = encode(data, fs, level, cks, prefix, wlog, ldm)
    plain, _ = Z.encode_seekable_frames(data, fs, level, cks, "1.5.7")
    for which in ("1.5.7", "system"):                # self-check with BOTH real decoders before committing
        assert Z.decode_stream(comp, len(data), which, prefix=prefix, window_log_max=wlog) == data
    index.append({"name": name, "prefix_recipe": pre_recipe, "recipe": recipe, "frame_size": fs, "level": level, "checksum": cks,
                  "window_log": wlog, "ldm": ldm, "prefix_len": len(prefix), "prefix_xxh64": f"{zko.xxh64(prefix):016x}",
                  "input_len": len(data), "input_xxh64": f"{zko.xxh64(data):016x}",
                  "frames": frames, "offset": len(blob), "length": len(comp), "comp_xxh64": f"{zko.xxh64(comp):016x}",
                  "length_without_prefix": len(plain)})
    blob += comp
    print(f"{name:24s} prefix={len(prefix):8d} in={len(data):8d} comp={len(comp):8d} (plain {len(plain)}) frames={len(frames)}")

with open(os.path.join(ROOT, "tests", "golden", "prefix_archives.bin"), "wb") as f:
    f.write(blob)
with open(os.path.join(ROOT, "tests", "golden", "prefix_archives.json"), "w") as f:
    json.dump({"libzstd": "1.5.7", "generator": "tools/make_prefix_goldens.py", "cases": index}, f, indent=1)
print("total payload bytes:", len(blob))
