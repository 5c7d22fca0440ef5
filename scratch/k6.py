import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep("__device__ __forceinline__ void sw_encode_i4x4(SwLds &s, const SwArgs &a, int idx, int &cbp_luma, int lane)",
    "__device__ __forceinline__ void sw_encode_i4x4(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int idx, int &cbp_luma, int lane)")
rep("    const int q = quant_one(v, s.qmf[0][l16], s.qbias[0][l16]);\n    const int nz = (__ballot(q != 0) & 0xffffull) != 0;",
    '''    int q;
    if (tq.on) {                                      // x264_quant_4x4_trellis( .., DCT_LUMA_4x4, 1, idx ), macroblock.c:134
        if (lane < 16) s.coef[idx][l16] = (i16)v;
        WAVE_SYNC();
        if (lane == 0) td_trellis_quant(tq.r->ts, &s.coef[idx][0], s.qmf[0], tq.r->unq4[0], d_w4z, d_zz4, tq.r->cabac, 2, d_trellis_lambda2[1][Q.qp], 0, 0, 16);
        WAVE_SYNC();
        q = s.coef[idx][l16];
    } else
        q = quant_one(v, s.qmf[0][l16], s.qbias[0][l16]);
    const int nz = (__ballot(q != 0) & 0xffffull) != 0;''')
rep("        int d = dequant_one(q, s.qdq[0][l16], a.qp / 6 - 4);","        int d = dequant_one(q, s.qdq[0][l16], Q.qp / 6 - 4);")
open(p,'w').write(s)
print("ok")
