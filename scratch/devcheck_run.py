import ctypes as C, sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import refslice as rs
from scratch.cmp_chain import static_clip
lib = C.CDLL('/root/repo/oracle/libdevcheck.so')
base = dict(me_method=1, n_refs=3, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
for size,qp,subme,ekw,clipf in [((208,144),26,7,dict(trellis=1,psy_rd=1.0),rs.clip),((208,144),18,7,dict(trellis=2,psy_rd=1.0,aq_mode=1),rs.clip),
                          ((200,120),34,6,dict(),static_clip),((208,144),26,5,dict(),rs.clip),((96,80),10,7,dict(trellis=2),rs.clip),((96,80),44,7,dict(trellis=1,psy_rd=0.5),rs.clip)]:
    p = rs.make_params(size[0],size[1],4,qp=qp,subme=subme,**base)
    y,u,v = clipf(size[0],size[1],4)
    b = rs.run2(lib,"x264o_encode_chain2",p,rs.make_ext(**ekw),y,u,v)
    print(size,qp,subme,ekw,"calls",lib.x264o_devcheck_calls(),"bad",lib.x264o_devcheck_bad(), flush=True)
