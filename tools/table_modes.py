#!/usr/bin/env python3
"""Scratch (round 6): which sequence-table modes libzstd's blocks use on the 8d text (2 MiB frames, levels 1 and 3) and the accuracy log of the
first table a block defines -- what zk_k_fse_quad's LDS budget per block depends on.  CPU only.   python tools/table_modes.py"""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import zko
from oracle import libzstd_ref as Z
import collections
def parse(comp, frames):
    cnt = collections.Counter(); pos = 0; nb = 0; als = collections.Counter()
    for cs, ds in frames:
        f = comp[pos:pos+cs]; pos += cs
        fhd = f[4]; single = (fhd>>5)&1; did = fhd&3; fcsf = fhd>>6
        p = 5 + (0 if single else 1) + (4 if did==3 else did) + (single if fcsf==0 else 1<<fcsf)
        while True:
            h = f[p] | f[p+1]<<8 | f[p+2]<<16; p += 3
            last, typ, size = h&1, (h>>1)&3, h>>3
            if typ == 2:
                c = f[p:p+size]
                lt = c[0]&3; sf = (c[0]>>2)&3
                if lt < 2:
                    hl = 1 if sf in (0,2) else (2 if sf==1 else 3)
                    regen = (c[0]>>3) if sf in (0,2) else ((c[0]>>4)|(c[1]<<4) if sf==1 else (c[0]>>4)|(c[1]<<4)|(c[2]<<12))
                    lsz = hl + (regen if lt==0 else 1)
                else:
                    if sf in (0,1): hl=3; comp_sz = ((c[1]>>6)|(c[2]<<2)) & 0x3ff
                    elif sf==2: hl=4; comp_sz = ((c[2]>>2)|(c[3]<<6)) & 0x3fff
                    else: hl=5; comp_sz = ((c[2]>>6)|(c[3]<<2)|(c[4]<<10)) & 0x3ffff
                    lsz = hl + comp_sz
                q = lsz; b0 = c[q]
                if b0 == 0: nseq=0; q+=1
                elif b0 < 128: nseq=b0; q+=1
                elif b0 < 255: nseq=((b0-128)<<8)+c[q+1]; q+=2
                else: nseq=c[q+1]+(c[q+2]<<8)+0x7f00; q+=3
                if nseq:
                    m = c[q]; q += 1
                    modes = ((m>>6)&3, (m>>4)&3, (m>>2)&3)
                    cnt[modes] += 1; nb += 1
                    # accuracy logs of FSE_Compressed tables (first 4 bits + 5)
                    for t, md in enumerate(modes):
                        if md == 2:
                            al = (c[q] & 15) + 5; als[(t, al)] += 1
                            break   # only the first table's header is at q (others follow after variable length)
            p += size if typ != 1 else 1
            if last: break
    return cnt, nb, als
for level in (1, 3):
    data = zko.gen_chunks(8 << 20, 11 + level)
    comp, frames = Z.encode_seekable_frames(data, 2 << 20, level, True)
    cnt, nb, als = parse(comp, frames)
    print("level", level, "blocks", nb, {k: v for k, v in cnt.most_common(8)}, dict(als))
