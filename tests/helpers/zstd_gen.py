"""TEST INFRASTRUCTURE -- a generator of VALID zstd frames that no encoder here would write: every choice the format leaves open (RFC 8878) is
drawn at random, so that the decoders meet corners the libzstd-made goldens only reach by chance:

  frame header   Single_Segment or a Window_Descriptor (exponent + mantissa), Frame_Content_Size of 0 / 1 / 2 / 4 / 8 bytes, checksum or not,
                 a Dictionary_ID field of any width holding 0, the Unused_Bit
  blocks         Raw, RLE, Compressed in any order; empty blocks; a last block of any type
  literals       Raw / RLE (1-, 2-, 3-byte headers), Huffman with 1 or 4 streams (a random complete code of depth <= 11; weights direct or FSE-compressed),
                 Treeless (the table of an earlier block)
  sequences      0 ... a few hundred per block; Predefined / RLE / FSE_Compressed (random normalised counts, "less than 1" probabilities,
                 zero runs, any legal accuracy log) / Repeat mode per table, independently; 1-, 2-, 3-byte Number_of_Sequences
  offsets        new offsets up to the window, the three repeat codes with and without Literals_Length 0 (incl. "Repeated_Offset1 - 1")

It writes the bytes AND runs the sequences on a model of its own; what the frame MEANS is still decided by the real libzstd 1.5.7 in the tests that
use it (tests/test_generated_frames.py: libzstd, the oracle, the lane code on the CPU; tests/test_gpu_generated_frames.py: the kernels) -- the
model's output only has to agree with it, which checks the generator.  Nothing of the product imports this."""
import random
import struct

LL_BASE = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
LL_BITS = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
ML_BASE = list(range(3, 35)) + [35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539]
ML_BITS = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
LL_DEF = [4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1]
OF_DEF = [1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1]
ML_DEF = [1, 4, 3, 2, 2, 2, 2, 2, 2] + [1] * 37 + [-1] * 7
assert len(LL_BASE) == len(LL_BITS) == len(LL_DEF) == 36 and len(ML_BASE) == len(ML_BITS) == len(ML_DEF) == 53 and len(OF_DEF) == 29


class BackBits:
    """A backward bitstream (RFC 8878 4.1): fields are added in the order the READER meets them."""

    def __init__(self):
        self.acc, self.n = 1, 0

    def add(self, v, w):
        assert 0 <= v < (1 << w) or (w == 0 and v == 0)
        self.acc = (self.acc << w) | v
        self.n += w

    def bytes(self):
        return self.acc.to_bytes((self.n + 1 + 7) // 8, "little")


class FwdBits:
    def __init__(self):
        self.acc, self.n = 0, 0

    def add(self, v, w):
        assert 0 <= v < (1 << w)
        self.acc |= v << self.n
        self.n += w

    def bytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def fse_cells(norm, al):
    """The decoding table of RFC 8878 4.1.1 as [(symbol, number of bits, baseline)]"""
    size, high = 1 << al, (1 << al) - 1
    sym = [0] * size
    nxt = []
    for s, c in enumerate(norm):
        if c == -1:
            sym[high] = s; high -= 1; nxt.append(1)
        else:
            nxt.append(c)
    step, pos, mask = (size >> 1) + (size >> 3) + 3, 0, size - 1
    for s, c in enumerate(norm):
        for _ in range(max(c, 0)):
            sym[pos] = s
            pos = (pos + step) & mask
            while pos > high:
                pos = (pos + step) & mask
    assert pos == 0
    cells = []
    for i in range(size):
        s = sym[i]
        x = nxt[s]; nxt[s] += 1
        nb = al - (x.bit_length() - 1)
        cells.append((s, nb, (x << nb) - size))
    return cells


def write_ncount(norm, al):
    """FSE table description (RFC 8878 4.1.1): the mirror of the oracle's fse_read_ncount"""
    f = FwdBits()
    f.add(al - 5, 4)
    remaining, threshold, nb = (1 << al) + 1, 1 << al, al + 1
    s = 0
    while remaining > 1:
        cnt = norm[s]; s += 1
        v, mx = cnt + 1, 2 * threshold - 1 - remaining
        if v < mx: f.add(v, nb - 1)
        elif v < threshold: f.add(v, nb)
        else: f.add(v + mx, nb)
        remaining -= abs(cnt)
        if cnt == 0:
            z = 0
            while s + z < len(norm) and norm[s + z] == 0: z += 1
            s += z
            while z >= 3: f.add(3, 2); z -= 3
            f.add(z, 2)
        while remaining < threshold:
            nb -= 1; threshold >>= 1
    assert remaining == 1 and all(c == 0 for c in norm[s:])
    return f.bytes()


def random_norm(rng, used, nsym, al_lo, al_hi):
    """normalised counts over symbols 0..nsym-1 with every symbol of `used` present: (norm, accuracy log)"""
    used = sorted(set(used) | ({rng.randrange(nsym)} if rng.random() < 0.5 else set()))
    lo = al_lo
    while (1 << lo) < len(used): lo += 1
    al = rng.randint(lo, al_hi)
    size = 1 << al
    norm = [0] * nsym
    left = size
    for s in used:
        if rng.random() < 0.3: norm[s] = -1
        else: norm[s] = 1
        left -= 1
    while left:                                              # the rest to random symbols (not the "less than 1" ones), in lumps
        s = rng.choice(used)
        if norm[s] == -1:
            if all(norm[u] == -1 for u in used): norm[s] = 1
            continue
        k = rng.randint(1, max(1, left // 2)) if rng.random() < 0.5 else 1
        norm[s] += k; left -= k
    while norm and norm[-1] == 0: norm.pop()
    return norm, al


def code_of(v, base, bits):
    c = len(base) - 1
    while base[c] > v: c -= 1
    assert v - base[c] < (1 << bits[c])
    return c, v - base[c], bits[c]


class HufCode:
    """a random complete prefix code of depth <= 11 over `symbols` (>= 2 of them; the largest symbol's weight is the implied one)"""

    def __init__(self, rng, symbols):
        symbols = sorted(symbols)
        assert len(symbols) >= 2
        depths = [1, 1]
        while len(depths) < len(symbols):
            cand = [i for i, d in enumerate(depths) if d < 11]
            i = rng.choice(cand)
            d = depths.pop(i) + 1
            depths += [d, d]
        rng.shuffle(depths)
        self.maxbits = max(depths)
        self.weights = [0] * (symbols[-1] + 1)
        for s, d in zip(symbols, depths): self.weights[s] = self.maxbits + 1 - d
        pos, self.code = 0, {}
        for wt in range(1, self.maxbits + 1):
            for s, w in enumerate(self.weights):
                if w == wt:
                    nb = self.maxbits + 1 - wt
                    self.code[s] = (pos >> (self.maxbits - nb), nb)
                    pos += 1 << (wt - 1)
        assert pos == 1 << self.maxbits

    def description(self, rng):
        """-> (bytes, "direct" | "fse") or None: four bits per weight, or the weights as an FSE stream on two interleaved states (RFC 8878 4.2.1.2)"""
        w = self.weights[:-1]
        if len(w) <= 128 and (len(w) < 2 or rng.random() < 0.5):
            out = bytearray([127 + len(w)])
            for i in range(0, len(w), 2):
                out.append((w[i] << 4) | (w[i + 1] if i + 1 < len(w) else 0))
            return bytes(out), "direct"
        for _ in range(8):
            norm, al = random_norm(rng, w, 12, 5, 6)
            cells = fse_cells(norm, al)
            # state A decodes w[0], w[2], ..., state B w[1], w[3], ...; the reader stops when the state that gave w[n - 2] cannot be
            # updated any more (it would read past the start) -- so that state's last cell has to ask for at least one bit
            n = len(w)
            chains = []
            for start in (0, 1):
                idx = list(range(start, n, 2))
                last = [k for k, c in enumerate(cells) if c[0] == w[idx[-1]] and (c[1] > 0 or idx[-1] != n - 2)]
                if not last: break
                states, bits = [rng.choice(last)], []
                for i in reversed(idx[:-1]):
                    nxt = states[0]
                    k = next(k for k, c in enumerate(cells) if c[0] == w[i] and c[2] <= nxt < c[2] + (1 << c[1]))
                    bits.insert(0, (nxt - cells[k][2], cells[k][1])); states.insert(0, k)
                chains.append((states, bits))
            if len(chains) < 2: continue
            b = BackBits()
            b.add(chains[0][0][0], al); b.add(chains[1][0][0], al)
            for k in range(n - 2):
                b.add(*chains[k & 1][1][k >> 1])
            body = write_ncount(norm, al) + b.bytes()
            if 2 <= len(body) < 128: return bytes([len(body)]) + body, "fse"
        return None

    def stream(self, data):
        b = BackBits()
        for x in data:                                      # symbols are decoded first to last, each from the top of what is left
            c, nb = self.code[x]
            b.add(c, nb)
        return b.bytes()


class FrameGen:
    def __init__(self, seed, max_blocks=6, max_seq=300, max_lit=3000, prefix=b"", dense=False):
        self.rng = random.Random(seed)
        self.want_dense = dense                              # one block of the frame with more than 0x7F00 sequences
        self.prefix = bytes(prefix)                          # a raw-content prefix the frame is written against (ZSTD_CCtx_refPrefix): offsets reach into it
        self.max_blocks, self.max_seq, self.max_lit = max_blocks, max_seq, max_lit
        self.features = set()

    # ---- literals
    def literals_section(self, lits):
        rng = self.rng
        n = len(lits)
        alphabet = sorted(set(lits))
        kinds = ["raw"]
        if n and len(alphabet) == 1: kinds += ["rle"] * 3
        if n >= 2 and len(alphabet) >= 2: kinds += ["huf"] * 4
        if n >= 1 and self.huf is not None and all(x in self.huf.code for x in alphabet): kinds += ["treeless"] * 8
        kind = rng.choice(kinds)
        self.features.add("lit_" + kind)
        if kind in ("raw", "rle"):
            t = 0 if kind == "raw" else 1
            fmts = [f for f, cap in ((1, 1 << 12), (3, 1 << 20)) if n < cap] + ([0, 0] if n < 32 else [])
            f = rng.choice(fmts)
            if f == 0: hdr = bytes([t | (n << 3)])          # Size_Format ?0: one bit, the size's five bits start at bit 3
            elif f == 1: hdr = struct.pack("<H", t | (1 << 2) | (n << 4))
            else: hdr = (t | (3 << 2) | (n << 4)).to_bytes(3, "little")
            self.features.add("lit_hdr%d" % len(hdr))
            return hdr + (bytes(lits) if kind == "raw" else bytes(lits[:1]))
        if kind == "huf":
            extra = set(rng.sample(range(0, 256), rng.randint(0, 3))) if rng.random() < 0.3 else set()
            huf = HufCode(rng, set(alphabet) | extra)
            d = huf.description(rng)
            if d is None: return None
            self.huf, desc = huf, d[0]
            self.features.add("huf_weights_" + d[1])
        else:
            desc = b""
        one = n < 1024 and (n < 8 or rng.random() < 0.4)
        if one:
            payload = desc + self.huf.stream(lits)
            streams_fmt = [0]
        else:
            q = (n + 3) // 4
            parts = [self.huf.stream(lits[i * q:(i + 1) * q]) for i in range(3)] + [self.huf.stream(lits[3 * q:])]
            if any(len(p) > 0xFFFF for p in parts[:3]): return None
            payload = desc + struct.pack("<HHH", *(len(p) for p in parts[:3])) + b"".join(parts)
            streams_fmt = [1, 2, 3]
        c = len(payload)
        fmts = [f for f in streams_fmt if (f <= 1 and n < 1024 and c < 1024) or (f == 2 and n < 16384 and c < 16384) or (f == 3 and n < (1 << 18) and c < (1 << 18))]
        if not fmts: return None
        f = rng.choice(fmts)
        t = 2 if kind == "huf" else 3
        self.features.add("huf_streams%d" % (1 if one else 4)); self.features.add("huf_fmt%d" % f)
        if f <= 1: hdr = (t | (f << 2) | (n << 4) | (c << 14)).to_bytes(3, "little")
        elif f == 2: hdr = (t | (2 << 2) | (n << 4) | (c << 18)).to_bytes(4, "little")
        else: hdr = (t | (3 << 2) | (n << 4) | (c << 22)).to_bytes(5, "little")
        return hdr + payload

    # ---- sequences
    def table_for(self, which, codes, nsym, al_lo, al_hi, default, default_al):
        """-> (mode, description bytes, cells, accuracy log) for the table `which` given the codes this block uses"""
        rng = self.rng
        modes = []
        if all(c < len(default) and default[c] != 0 for c in codes): modes += [0, 0]
        if len(set(codes)) == 1: modes += [1, 1]
        modes += [2, 2]
        prev = self.tables.get(which)
        if prev is not None and all(any(cell[0] == c for cell in prev[0]) for c in set(codes)): modes += [3, 3]
        mode = rng.choice(modes)
        self.features.add("%s_mode%d" % (which, mode))
        if mode == 0: cells, al, desc = fse_cells(default, default_al), default_al, b""
        elif mode == 1: cells, al, desc = [(codes[0], 0, 0)], 0, bytes([codes[0]])
        elif mode == 2:
            norm, al = random_norm(rng, codes, nsym, al_lo, al_hi)
            if -1 in norm: self.features.add("fse_lowprob")
            cells, desc = fse_cells(norm, al), write_ncount(norm, al)
        else: cells, al = prev; desc = b""
        self.tables[which] = (cells, al)
        return mode, desc, cells, al

    def sequences_section(self, seqs):
        """seqs: [(ll, ml, offset_value)] -> bytes"""
        rng = self.rng
        n = len(seqs)
        if n == 0: return b"\x00"
        form = 3 if n >= 0x7F00 else 2 if n >= 128 else rng.choice([1, 1, 1, 2])       # (the two-byte form may carry a small count)
        if form == 1: hdr = bytes([n])
        elif form == 2: hdr = bytes([128 + (n >> 8), n & 255])
        else: hdr = bytes([255]) + struct.pack("<H", n - 0x7F00)
        self.features.add("nseq_form%d" % form)
        ll = [code_of(s[0], LL_BASE, LL_BITS) for s in seqs]
        ml = [code_of(s[1], ML_BASE, ML_BITS) for s in seqs]
        of = []
        for s in seqs:
            c = s[2].bit_length() - 1
            of.append((c, s[2] - (1 << c), c))
        m_ll, d_ll, c_ll, a_ll = self.table_for("ll", [x[0] for x in ll], 36, 5, 9, LL_DEF, 6)
        m_of, d_of, c_of, a_of = self.table_for("of", [x[0] for x in of], 32, 5, 8, OF_DEF, 5)
        m_ml, d_ml, c_ml, a_ml = self.table_for("ml", [x[0] for x in ml], 53, 5, 9, ML_DEF, 6)

        def chain(cells, codes):
            """states[i] of a decoder that sees codes[i] in state i, and the bits that take it from i to i + 1"""
            by_sym = {}
            for k, cell in enumerate(cells): by_sym.setdefault(cell[0], []).append(k)
            states, bits = [rng.choice(by_sym[codes[-1]])], []
            for c in reversed(codes[:-1]):
                nxt = states[-1]
                k = next(k for k in by_sym[c] if cells[k][2] <= nxt < cells[k][2] + (1 << cells[k][1]))
                bits.append((nxt - cells[k][2], cells[k][1]))
                states.append(k)
            states.reverse(); bits.reverse()
            return states, bits
        s_ll, b_ll = chain(c_ll, [x[0] for x in ll])
        s_of, b_of = chain(c_of, [x[0] for x in of])
        s_ml, b_ml = chain(c_ml, [x[0] for x in ml])
        b = BackBits()
        b.add(s_ll[0], a_ll); b.add(s_of[0], a_of); b.add(s_ml[0], a_ml)
        for i in range(n):
            b.add(of[i][1], of[i][2]); b.add(ml[i][1], ml[i][2]); b.add(ll[i][1], ll[i][2])
            if i + 1 < n:
                b.add(*b_ll[i]); b.add(*b_ml[i]); b.add(*b_of[i])
        return hdr + bytes([(m_ll << 6) | (m_of << 4) | (m_ml << 2)]) + d_ll + d_of + d_ml + b.bytes()

    # ---- a compressed block: sequences on the model, then the two sections
    def compressed_block(self, out, rep, window, block_max):
        rng = self.rng
        nseq = 0 if rng.random() < 0.15 else rng.randint(1, rng.choice([3, 20, self.max_seq]))
        style = rng.random()
        dense = self.dense and block_max == 1 << 17 and len(out) >= 8
        if dense: nseq = rng.randint(0x7F00, 0x7F00 + 6000)     # the three-byte Number_of_Sequences: matches of 3 ... 4 bytes back to back
        self.dense = self.dense and not dense
        if self.alpha is not None and rng.random() < 0.4: alpha = bytes(rng.sample(list(self.alpha), rng.randint(1, len(self.alpha))))   # (a block the table of the one before fits: Treeless)
        else: alpha = bytes(rng.sample(range(0, rng.choice([129, 256])), rng.randint(1, rng.choice([2, 6, 40, 100])))) if rng.random() < 0.8 else bytes(range(256))
        self.alpha = alpha
        lits, seqs = bytearray(), []
        produced = 0
        start = len(out)
        for _ in range(nseq):
            ll = 0 if rng.random() < 0.3 else rng.choice([rng.randint(1, 8), rng.randint(1, 40), rng.randint(1, 2000 if style < 0.1 else 60)])
            ml = rng.choice([3, rng.randint(3, 12), rng.randint(3, 130), rng.randint(3, 3000 if style < 0.1 else 200)])
            if dense: ll, ml = (1 if rng.random() < 0.02 else 0), (3 if rng.random() < 0.9 else 4)
            if len(lits) + ll > self.max_lit or produced + ll + ml > block_max: break
            here = len(self.prefix) + len(out) + ll           # bytes available behind the literals
            reach = min(here, window)                         # (an offset beyond the window into a prefix: libzstd takes it until its ring wraps -- not valid zstd, not drawn)
            r = rng.random()
            ofv = None
            if r < 0.45:                                      # a repeat code
                code = rng.randint(1, 3)
                idx = code - 1 + (1 if ll == 0 else 0)
                off = rep[0] if idx == 0 else (rep[0] - 1 if idx == 3 else rep[idx])
                if 0 < off <= reach: ofv = code
            if ofv is None:
                if here == 0: continue
                hi = reach
                off = rng.choice([1, rng.randint(1, min(hi, 16)), rng.randint(1, hi), hi, max(1, min(hi, len(out) + ll + rng.randint(0, 40)))])   # (the last: around the frame's first byte, where a prefix begins)
                ofv = off + 3
            if ofv > 3: rep = [ofv - 3, rep[0], rep[1]]; self.features.add("off_new")
            else:
                idx = ofv - 1 + (1 if ll == 0 else 0)
                self.features.add("rep_idx%d" % idx)
                if idx:
                    off = rep[0] - 1 if idx == 3 else rep[idx]
                    rep = [off, rep[0], rep[1]] if idx > 1 else [off, rep[0], rep[2]]
            new = bytes(rng.choice(alpha) for _ in range(ll))
            lits += new; out += new
            self.max_off = max(self.max_off, off)
            if off > len(out): self.features.add("off_into_prefix")
            for _ in range(ml):
                i = len(out) - off
                out.append(out[i] if i >= 0 else self.prefix[len(self.prefix) + i])
            produced += ll + ml
            seqs.append((ll, ml, ofv))
        tail = rng.randint(0, min(self.max_lit - len(lits), block_max - produced, rng.choice([0, 5, 300, 3000])))
        new = bytes(rng.choice(alpha) for _ in range(tail))
        lits += new; out += new
        lit = self.literals_section(lits)
        if lit is None: return None, rep
        body = lit + self.sequences_section(seqs)
        if len(body) > block_max or len(body) >= (1 << 21): return None, rep
        return body, rep

    def frame(self):
        """-> (frame bytes, decoded bytes, features)"""
        rng = self.rng
        while True:
            self.huf, self.tables, self.alpha = None, {}, None
            self.features = set()
            self.max_off = 0
            self.dense = self.want_dense
            exp, mant = rng.choice([0, 0, 1, 3, 7, rng.randint(0, 10)]), rng.randint(0, 7)
            window = (1 << (10 + exp)) + ((1 << (10 + exp)) >> 3) * mant
            block_max = min(window, 1 << 17)
            out, rep = bytearray(), [1, 4, 8]
            blocks = []
            ok = True
            for _ in range(rng.randint(1, self.max_blocks)):
                r = rng.random()
                if r < 0.15:
                    n = rng.choice([0, rng.randint(0, 40), rng.randint(0, min(block_max, 5000))])
                    data = bytes(rng.randrange(256) for _ in range(n))
                    blocks.append((0, n, data)); out += data; self.features.add("raw")
                elif r < 0.3:
                    n = rng.choice([0, 1, rng.randint(0, 300), rng.randint(0, block_max)])
                    v = rng.randrange(256)
                    blocks.append((1, n, bytes([v]))); out += bytes([v]) * n; self.features.add("rle")
                else:
                    snap = (len(out), list(rep), self.huf, dict(self.tables))
                    body, rep2 = self.compressed_block(out, rep, window, block_max)
                    if body is None:
                        del out[snap[0]:]; rep, self.huf, self.tables = snap[1], snap[2], snap[3]
                        continue
                    rep = rep2
                    blocks.append((2, len(body), body)); self.features.add("compressed")
            if not blocks or self.dense: continue               # (a dense block was asked for and did not come about: window too small, ...)
            total = len(out)
            # Single_Segment: the window IS the content size, and Block_Maximum_Size follows it -- also for what a block holds
            single = total <= window and rng.random() < 0.4 and all(n <= min(total, 1 << 17) for t, n, _ in blocks if t == 2) and self.max_off <= total
            cks = rng.random() < 0.5
            if single:
                flags = [f for f, lo, hi in ((0, 0, 255), (1, 256, 65791), (2, 0, (1 << 32) - 1), (3, 0, (1 << 64) - 1)) if lo <= total <= hi]
                fcs_flag = rng.choice(flags)
            else:
                fcs_flag = rng.choice([0, 0] + [f for f, lo, hi in ((1, 256, 65791), (2, 0, (1 << 32) - 1), (3, 0, (1 << 64) - 1)) if lo <= total <= hi])
            did = rng.choice([0, 0, 0, 1, 2, 3])              # a Dictionary_ID field of 1 / 2 / 4 bytes holding 0 = "no dictionary" (RFC 8878 3.1.1.1.3)
            unused = 0x10 if rng.random() < 0.2 else 0         # Unused_Bit: a decoder shall not interpret it
            fhd = (fcs_flag << 6) | (0x20 if single else 0) | unused | (4 if cks else 0) | did
            hdr = bytes.fromhex("28B52FFD") + bytes([fhd]) + (b"" if single else bytes([(exp << 3) | mant])) + bytes((0, 1, 2, 4)[did])
            if did: self.features.add("did_field")
            if unused: self.features.add("unused_bit")
            if single and fcs_flag == 0: hdr += bytes([total])
            elif fcs_flag == 1: hdr += struct.pack("<H", total - 256)
            elif fcs_flag == 2: hdr += struct.pack("<I", total)
            elif fcs_flag == 3: hdr += struct.pack("<Q", total)
            self.features.add("single" if single else "windowed"); self.features.add("fcs%d" % fcs_flag)
            body = bytearray()
            for i, (t, n, payload) in enumerate(blocks):
                body += ((n << 3) | (t << 1) | (1 if i + 1 == len(blocks) else 0)).to_bytes(3, "little") + payload
            return bytes(hdr + body), bytes(out), cks, set(self.features)


def generate(seed, xxh64=None, **kw):
    """-> (frame bytes, decoded bytes, features).  xxh64: the caller's hash function (this file has none) -- without one no frame gets a checksum"""
    f, out, cks, feats = FrameGen(seed, **kw).frame()
    if cks and xxh64 is not None:
        f += (xxh64(out) & 0xFFFFFFFF).to_bytes(4, "little")
    elif cks:
        f = f[:4] + bytes([f[4] & ~4]) + f[5:]
    return f, out, feats | ({"checksum"} if cks and xxh64 is not None else set())
