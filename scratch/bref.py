import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import refslice as R
W,H,F=160,128,8
y,u,v=R.clip(W,H,F)
p=R.make_params(W,H,F,qp=28,me_method=R.ME_HEX,subme=int(sys.argv[1]) if len(sys.argv)>1 else 7,n_refs=2,inter=0x113,intra=0x3,transform8x8=1,cabac=1,mixed_refs=1,deblock=1,keyint=0)
e=R.make_ext(trellis=1,psy_rd=1.0,aq_mode=1,bframes=2,weightb=1,direct_pred=int(sys.argv[2]) if len(sys.argv)>2 else 1)
a=R.run_reference2(p,e,y,u,v)
print(a['frame_info']); print(a['frame_info2']); print(a['payload_len'])
for f in range(F):
    print(f, np.bincount(a['mb_type'][f].astype(np.int64), minlength=19))
