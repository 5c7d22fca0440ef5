import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
a=s.index("template <int WPE, bool LL = false>")
b=s.index("// b_fast_intra's raster-order term, settled once the frame is complete")
k=s[a:b]
k=k.replace("a.lambda","Q.lambda")
k=re.sub(r"refs\.ref_cost\[([^\]]+)\]", r"(Q.lambda * refs.ref_bits[\1])", k)
k=k.replace("(signed char)a.qp;","(signed char)Q.qp;")
s=s[:a]+k+s[b:]
s=s.replace("        t.ref_cost[i] = a.lambda * bits;","        t.ref_bits[i] = bits;")
open(p,'w').write(s)
print("ok")
