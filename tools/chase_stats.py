#!/usr/bin/env python3
"""How deep the executor's in-tile chains are (the chase loop of zk_k_exec runs deepest-chain-of-the-wave + 1 times over all 16 bytes of
every lane): the lane code on the CPU (tests/sim/zk_sim.cpp) over archives of the bench's text, this encoder's (level 1) and libzstd's."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import sim_decode, sim_lib
from oracle import zko
from oracle import libzstd_ref as Z

F = 2 << 20
data = zko.gen_chunks(2 * F)
def stats(comp, frames, lanes):
    buf = (C.c_uint64 * 32)()
    sim_lib().zk_sim_chase_stats(buf, 1)
    rc, out, st = sim_decode(comp, frames, CH=2 * lanes)
    assert rc == 0 and out == data
    sim_lib().zk_sim_chase_stats(buf, 0)
    hd = (C.c_uint64 * 16)(); sim_lib().zk_sim_huf_depth(hd)
    print("  Huffman trees by depth (a table of 2^depth cells; zk_k_huf's pool holds 8192 cells for 16 blocks): " + " ".join("%d:%d" % (i, hd[i]) for i in range(16) if hd[i]))
    sl = (C.c_uint64 * 2)(); sim_lib().zk_sim_slot_stats(sl)
    print("  slots that take the general walk (a match overlapping its own output): %d of %d = %.2f%%" % (sl[1], sl[0], 100.0 * sl[1] / max(1, sl[0])))
    v = np.array(buf[:], np.float64)
    w, b = v[:16], v[16:]
    print("  waves by deepest chain: " + " ".join("%d:%.1f%%" % (i, 100 * w[i] / w.sum()) for i in range(16) if w[i]), " mean %.2f -> %.2f passes of the loop" % ((w * np.arange(16)).sum() / w.sum(), (w * np.arange(16)).sum() / w.sum() + 1))
    print("  bytes by chain depth:   " + " ".join("%d:%.1f%%" % (i, 100 * b[i] / b.sum()) for i in range(16) if b[i]))
for name, mk in (("this encoder, level 1", lambda x: zko.frame_encode(x, 1, True)), ("libzstd level 1", 1), ("libzstd level 3", 3)):
    if isinstance(mk, int):
        ref = "1.5.7" if Z.load("1.5.7") is not None else "system"
        payload, frames = Z.encode_seekable_frames(data, F, mk, True, ref)
        fr, at = [], 0
        for c, d in frames: fr.append(payload[at:at + c]); at += c
    else:
        fr = [mk(data[i:i + F]) for i in range(0, len(data), F)]
    print(name, "(tiles of 256 lanes x 16 bytes)")
    stats(b"".join(fr), [(len(f), F) for f in fr], 256)
