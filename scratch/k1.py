import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)

rep("    int ref_cost[SW_MAX_REFS], poc_delta[SW_MAX_REFS], l0_inv_ref_poc[SW_MAX_REFS];",
    "    int ref_bits[SW_MAX_REFS], poc_delta[SW_MAX_REFS], l0_inv_ref_poc[SW_MAX_REFS];   // REF_COST = lambda * ref_bits (bs_size_te, R/encoder/analyse.c:195-197)")
rep("""struct SwLds {""","""// the macroblock's QP and what follows from it (x264_mb_analyse_init, R/encoder/analyse.c:227-230): one set per slice at constant
// QP, per macroblock with adaptive quantisation
struct SwQp { int qp, qpc, lambda, lambda2, skip_thresh; };

struct SwLds {""")
# helper signatures
rep("__device__ __forceinline__ void sw_luma4x4_fwd(SwLds &s, const SwArgs &a, int cat, bool dc_out, int lane, int *nr_acc4 = nullptr, int nr_on = 0)",
    "__device__ __forceinline__ void sw_luma4x4_fwd(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int cat, bool dc_out, int lane, int *nr_acc4 = nullptr, int nr_on = 0)")
rep("        int nz = 0, bits = a.qp / 6 - 4;\n#pragma unroll\n        for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }\n        SCAN4_FRAME(lv, c);\n        u32 nzm, big;\n        LEVEL_MASKS(lv, nzm, big);\n#pragma unroll\n        for (int i = 0; i < 16; i++) { s.lv_y[16 * lane + i] = lv[i]; s.coef[lane][i] = (i16)dequant_one(c[i], dq[i], bits); }",
    "        int nz = 0, bits = Q.qp / 6 - 4;\n        if (tq.on) {\n#pragma unroll\n            for (int i = 0; i < 16; i++) c[i] = s.coef[lane][i];          // quantised by the trellis below\n#pragma unroll\n            for (int i = 0; i < 16; i++) nz |= c[i];\n        } else {\n#pragma unroll\n            for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }\n        }\n        SCAN4_FRAME(lv, c);\n        u32 nzm, big;\n        LEVEL_MASKS(lv, nzm, big);\n#pragma unroll\n        for (int i = 0; i < 16; i++) { s.lv_y[16 * lane + i] = lv[i]; s.coef[lane][i] = (i16)dequant_one(c[i], dq[i], bits); }")
open(p,'w').write(s)
print("ok")
