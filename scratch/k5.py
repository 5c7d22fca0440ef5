import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep("__device__ __forceinline__ void sw_luma8x8_fwd(SwLds &s, int cat, int mask, int lane, int *nr_acc8 = nullptr, int nr_on = 0)",
    "__device__ __forceinline__ void sw_luma8x8_fwd(SwLds &s, const SwQp &Q, SwTq tq, int cat, int mask, int lane, int *nr_acc8 = nullptr, int nr_on = 0)")
rep('''            int q = quant_one(coef[64 * j + lane], mfl, bsl);
            coef[64 * j + lane] = (i16)q;
            nzmask[j] = __ballot(q != 0);
        }
    WAVE_SYNC();''','''            if (!tq.on) {
                int q = quant_one(coef[64 * j + lane], mfl, bsl);
                coef[64 * j + lane] = (i16)q;
                nzmask[j] = __ballot(q != 0);
            }
        }
    WAVE_SYNC();
    if (tq.on) {                                      // x264_quant_8x8_trellis (R/encoder/rdo.c:652-660), one lane
        if (lane == 0)
            for (int j = 0; j < 4; j++)
                if ((mask >> j) & 1)
                    td_trellis_quant(tq.r->ts, coef + 64 * j, s.q8mf[cat], tq.r->unq8[cat], SwW8(), c_scan8[0], tq.r->cabac, 5, d_trellis_lambda2[cat == 0][Q.qp], 0, 0, 64);
        WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((mask >> j) & 1) nzmask[j] = __ballot(coef[64 * j + lane] != 0);
    }''')
rep("__device__ __forceinline__ int sw_encode_inter_luma8(SwLds &s, const SwArgs &a, int lane, int *nr_acc8 = nullptr, int nr_on = 0)\n{\n    sw_luma8x8_fwd(s, 1, 0xf, lane, nr_acc8, nr_on);",
    "__device__ __forceinline__ int sw_encode_inter_luma8(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int lane, int *nr_acc8 = nullptr, int nr_on = 0)\n{\n    sw_luma8x8_fwd(s, Q, tq, 1, 0xf, lane, nr_acc8, nr_on);\n    const int b_decimate = a.dct_decimate && !tq.on;           // \"8x8 trellis is inherently optimal decimation\", macroblock.c:630")
rep('''        if (v >> 8) {
            if (a.dct_decimate) { dec_mb += v & 255; if ((v & 255) >= 4) cbp |= 1 << i; }
            else cbp |= 1 << i;
        }
    }
    if (a.dct_decimate && dec_mb < 6) cbp = 0;''','''        if (v >> 8) {
            if (b_decimate) { dec_mb += v & 255; if ((v & 255) >= 4) cbp |= 1 << i; }
            else cbp |= 1 << i;
        }
    }
    if (b_decimate && dec_mb < 6) cbp = 0;''')
rep("    sw_luma8x8_add(s, 1, a.qp, cbp, lane);\n    return cbp;","    sw_luma8x8_add(s, 1, Q.qp, cbp, lane);\n    return cbp;")
rep("__device__ __forceinline__ void sw_encode_i8x8(SwLds &s, const SwArgs &a, int idx, int &cbp_luma, int lane)\n{\n    if (a.lossless) { sw_ll_i8x8(s, idx, cbp_luma, lane); return; }\n    sw_luma8x8_fwd(s, 0, 1 << idx, lane);",
    "__device__ __forceinline__ void sw_encode_i8x8(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int idx, int &cbp_luma, int lane)\n{\n    if (a.lossless) { sw_ll_i8x8(s, idx, cbp_luma, lane); return; }\n    sw_luma8x8_fwd(s, Q, tq, 0, 1 << idx, lane);")
rep("    if (nz) { cbp_luma |= 1 << idx; sw_luma8x8_add(s, 0, a.qp, 1 << idx, lane); }","    if (nz) { cbp_luma |= 1 << idx; sw_luma8x8_add(s, 0, Q.qp, 1 << idx, lane); }")
open(p,'w').write(s)
print("ok")
