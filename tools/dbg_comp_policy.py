import io, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zeekstd_amd as zk
from zeekstd_amd import EncodeOptions, DecodeOptions, Decoder, FrameSizePolicy
from oracle import zko
eng = zk.Engine(0)
rng = np.random.default_rng(5)
text = zko.gen_chunks(24 << 20)
data = text[:12 << 20] + rng.integers(0, 256, 6 << 20, dtype=np.uint8).tobytes() + bytes(3 << 20) + text[12 << 20:]
n = 512 << 10
sink = io.BytesIO()
enc = EncodeOptions().engine(eng).frame_size_policy(FrameSizePolicy.Compressed(n)).checksum_flag(True).into_encoder(sink)
for piece in [data[:20 << 20], data[20 << 20:]]:
    enc.write_all(piece)
enc.finish()
dec = Decoder(DecodeOptions(sink.getvalue()).engine(eng))
st = dec.seek_table()
for i in range(st.num_frames()):
    c, d = st.frame_size_comp(i), st.frame_size_decomp(i)
    flag = "" if n <= c < n + 131591 or i == st.num_frames() - 1 else "  <-- outside"
    print(i, c, d, st.frame_start_decomp(i) >> 20, flag)
