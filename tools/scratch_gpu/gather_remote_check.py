"""Scratch: the root-side check of bench.py's gather leg (decode the last four frames out of a gathered archive) on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import zko
import zeekstd_amd as zk
from zeekstd_amd import SeekTable, Format
FRAME = 2 << 20
dev = torch.device("cuda", 0)
eng = zk.Engine(0)
nf = 8
data = zko.gen_chunks(nf * FRAME, 0)
comp, frames = eng.encode_frames(np.frombuffer(data, np.uint8), FRAME, 1, True)
table = SeekTable.new()
table.log_frames(np.array([f[0] for f in frames], np.uint32), np.array([f[1] for f in frames], np.uint32))
out = torch.frombuffer(bytearray(comp + table.to_bytes(Format.Foot)), dtype=torch.uint8).to(dev)
nf_all = table.num_frames()
first_f = nf_all - 4
tc, td = table.offsets()
c0, c1 = int(tc[first_f]), int(tc[nf_all])
rel_c = torch.from_numpy((tc[first_f:] - tc[first_f]).astype(np.int64)).to(dev)
rel_d = torch.from_numpy((td[first_f:] - td[first_f]).astype(np.int64)).to(dev)
piece = torch.empty(c1 - c0 + 64, dtype=torch.uint8, device=dev)
piece[:c1 - c0] = out[c0:c1]
piece[c1 - c0:] = 0
o4 = torch.empty(4 * FRAME + 64, dtype=torch.uint8, device=dev)
s4 = torch.zeros(4, dtype=torch.int32, device=dev)
eng.decode_frames_dev(piece, c1 - c0, rel_c, rel_d, 0, 4, o4, 4 * FRAME, True, s4)
want4 = zko.gen_chunks(4 * FRAME, first_f)
print("REMOTE CHECK bit_exact", bytes(o4[:4 * FRAME].cpu().numpy()) == want4 and int(s4.abs().sum().item()) == 0)
