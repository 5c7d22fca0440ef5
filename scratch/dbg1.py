import ctypes as C, os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import refslice as rs
ora = C.CDLL('/root/repo/oracle/liboracle.so')
base = dict(me_method=1, n_refs=3, inter=0x13, intra=0x3, transform8x8=1, mixed_refs=1, cabac=1, deblock=1)
p = rs.make_params(208,144,1,qp=26,subme=7,**base)
y,u,v = rs.clip(208,144,1)
ekw=dict(trellis=1,psy_rd=1.0)
a = rs.run_reference2(p, rs.make_ext(**ekw), y,u,v)
b = rs.run2(ora, "x264o_encode_chain2", p, rs.make_ext(**ekw), y,u,v)
mb=0
print("type",a["mb_type"][0,mb],b["mb_type"][0,mb],"t8",a["t8"][0,mb],b["t8"][0,mb], "cbp", hex(a["cbp"][0,mb]), hex(b["cbp"][0,mb]))
print("nnz a",a["nnz"][0,mb]); print("nnz b",b["nnz"][0,mb])
la=a["luma"][0,mb].reshape(4,64); lb=b["luma"][0,mb].reshape(4,64)
for i in range(4):
    print(i, "a", la[i][:32]); print(i, "b", lb[i][:32])
print("i4mode", a["i4mode"][0,mb], b["i4mode"][0,mb])
