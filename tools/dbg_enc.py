import sys; sys.path.insert(0,'.')
import numpy as np
import zeekstd_amd as zk
from oracle import zko
eng=zk.Engine(0)
def parse_block(f, p):
    bh=int.from_bytes(f[p:p+3],'little'); typ=(bh>>1)&3; bs=bh>>3
    b=f[p+3:p+3+bs]
    info={"type":typ,"bsize":bs}
    if typ==2:
        t=b[0]&3; sf=(b[0]>>2)&3
        if t<2:
            hdr=1 if sf in(0,2) else 2 if sf==1 else 3
            regen=(b[0]>>3) if hdr==1 else ((b[0]>>4)+(b[1]<<4)) if hdr==2 else ((b[0]>>4)+(b[1]<<4)+(b[2]<<12)); ls=hdr+(regen if t==0 else 1)
        else:
            hdr=3 if sf<2 else 4 if sf==2 else 5
            v=int.from_bytes(b[:hdr],'little'); bits=10 if sf<2 else 14 if sf==2 else 18
            regen=(v>>4)&((1<<bits)-1); comp=(v>>(4+bits))&((1<<bits)-1); ls=hdr+comp
            info["tree_hdr"]=b[hdr]; info["jump"]=[int.from_bytes(b[hdr+1+ (b[hdr]-127+1)//2+2*k: hdr+1+(b[hdr]-127+1)//2+2*k+2],'little') for k in range(3)]
        info.update(lit_type=t, regen=regen, lit_section=ls)
        q=b[ls:]; n=q[0]; nh=1
        if n>=128: n=((q[0]-128)<<8)+q[1]; nh=2
        info.update(nseq=n, seq_bytes=bs-ls-nh)
    return info, p+3+(bs if typ!=1 else 1)
for n in (3000, 20000, 131072, 300000):
    data=zko.gen_text(n, 77)
    g,_=eng.encode_frames(data, 2<<20, 1, False)
    o=zko.frame_encode(data,1,False)
    print("n",n,"gpu",len(g),"oracle",len(o),"equal",g==o)
    if g!=o:
        pg=po=6
        for b in range(4):
            ig,pg=parse_block(g,pg); io,po=parse_block(o,po)
            print(" blk",b,"gpu",ig); print("       ora",io)
            if pg>=len(g) or po>=len(o): break
        break
