#!/usr/bin/env python3
"""Scratch: more verdicts on damaged input, checksums off.
 (a) prefix archives: the engine with the prefix against the oracle with the prefix, frame by frame;
 (b) zk_frame_content_sizes on damaged goldens (frames nobody holds sizes for): where the oracle accepts a frame the engine's size is the
     oracle's length and status 0; a decode with the sizes the engine reported then reaches the oracle's verdict and bytes;
 (c) the Level-B Decoder over a damaged archive (host buffers): bytes in front of the first damaged frame are readable and are the input's,
     the read that reaches the damaged frame fails."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import GOLDENS, PREFIX_GOLDENS, offsets_from_frames
from oracle import zko
import zeekstd_amd as zk


def oracle_frames(bad, c, want, prefix=None):
    """[(ok, bytes)] per frame: the oracle on each frame's bytes; ok also needs the frame to end where the table says"""
    res = []
    for f in range(len(c) - 1):
        try:
            o, used = zko.frame_decode(bytes(bad[int(c[f]):int(c[f + 1])]), (want[f] if want is not None else 1 << 21) + 64, False, prefix=prefix)
            res.append((used == int(c[f + 1] - c[f]) and (want is None or len(o) == want[f]), o))
        except zko.OracleError:
            res.append((False, b""))
    return res


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    eng = zk.Engine()
    wrong = 0
    # (a)
    for case in range(cases):
        g = PREFIX_GOLDENS[int(rng.integers(0, len(PREFIX_GOLDENS)))]
        bad = bytearray(g.comp)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        c, d = g.offsets()
        out, st = eng.decode_frames(bytes(bad) + b"\0" * 8, c, d, verify=False, raise_on_error=False, prefix=g.prefix())
        ref = oracle_frames(bad, c, [int(d[f + 1] - d[f]) for f in range(len(c) - 1)], g.prefix())
        for f, (ok, o) in enumerate(ref):
            if ok != (st[f] == 0) or (ok and out[int(d[f]):int(d[f + 1])] != o):
                wrong += 1; print("WRONG (a)", case, g.name, f, int(st[f]), ok); break
    print("(a) prefix archives:", cases, "cases, wrong so far", wrong)
    # (b)
    small = [g for g in GOLDENS if 0 < g.meta["input_len"] <= 400000]
    nacc = 0
    for case in range(cases):
        g = small[int(rng.integers(0, len(small)))]
        bad = bytearray(g.comp)
        for _ in range(int(rng.integers(1, 3))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        c, d = g.offsets()
        sizes, st = eng.frame_content_sizes(bytes(bad), c)
        ref = oracle_frames(bad, c, None)
        msg = None
        for f, (ok, o) in enumerate(ref):
            if ok and (st[f] != 0 or int(sizes[f]) != len(o)): msg = "frame %d: oracle %d bytes, engine status %d size %d" % (f, len(o), int(st[f]), int(sizes[f])); break
        if msg is None:
            # decode with the engine's own sizes (frames it refused: size 0)
            sz = [int(sizes[f]) if st[f] == 0 else 0 for f in range(len(c) - 1)]
            dd = np.concatenate([[0], np.cumsum(sz)]).astype(np.uint64)
            out, st2 = eng.decode_frames(bytes(bad) + b"\0" * 8, c, dd, verify=False, raise_on_error=False)
            for f, (ok, o) in enumerate(ref):
                if st[f] != 0: continue
                if ok != (st2[f] == 0) or (ok and out[int(dd[f]):int(dd[f + 1])] != o): msg = "frame %d: decode with reported size: status %d, oracle %s" % (f, int(st2[f]), ok); break
                nacc += ok
        if msg:
            wrong += 1; print("WRONG (b)", case, g.name, msg)
    print("(b) sizes of damaged frames:", cases, "cases,", nacc, "frames accepted by both, wrong so far", wrong)
    # (c)
    multi = [g for g in small if len(g.frames) >= 3]
    nread = nacc2 = 0
    for case in range(cases // 3):
        g = multi[int(rng.integers(0, len(multi)))]
        c, d = g.offsets()
        bad = bytearray(g.comp)
        f0 = int(rng.integers(0, len(g.frames)))
        lo, hi = int(c[f0]), int(c[f0 + 1])
        for _ in range(3):
            bad[int(rng.integers(lo, hi))] ^= 1 << int(rng.integers(0, 8))
        try:
            o, used = zko.frame_decode(bytes(bad[lo:hi]), int(d[f0 + 1] - d[f0]) + 64, True)
            accepted = used == hi - lo and len(o) == int(d[f0 + 1] - d[f0])
        except zko.OracleError:
            accepted = False
        table = zk.SeekTable.new()
        for cs, ds in g.frames: table.log_frame(cs, ds)
        dec = zk.DecodeOptions(bytes(bad)).seek_table(table).into_decoder()
        got = bytearray(); failed = False
        try:
            while True:
                buf = dec.read(int(rng.integers(1, 50000)))
                if not buf: break
                got += buf
        except (zk.Error, zk.ZkError):
            failed = True
        data = g.input()
        if accepted:                                     # a flip the format does not notice: the frame's bytes are the oracle's
            want = data[:int(d[f0])] + o + data[int(d[f0 + 1]):]
            if failed or bytes(got) != want:
                wrong += 1; print("WRONG (c)", case, g.name, "frame", f0, "is valid zstd (other bytes) but the Decoder", "failed" if failed else "read other bytes")
        elif not failed:
            wrong += 1; print("WRONG (c)", case, g.name, "damaged frame", f0, "was read through:", len(got), "bytes")
        elif bytes(got) != data[:len(got)] or len(got) > int(d[f0]):
            wrong += 1; print("WRONG (c)", case, g.name, "bytes in front of the damage differ / too many:", len(got), int(d[f0]))
        nacc2 += accepted
        nread += 1
    print("(c) Decoder over damaged archives:", nread, "cases,", nacc2, "of them still valid zstd, wrong so far", wrong)
    print("total wrong", wrong)


if __name__ == "__main__":
    main()
