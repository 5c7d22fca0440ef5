import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep("static __device__ const int d_w4z[16] = {800, 320, 320, 800, 128, 800, 320, 128, 128, 320, 320, 800, 128, 320, 320, 128};",
    "static __device__ const int d_w4z[16] = {800, 320, 320, 800, 128, 800, 320, 320, 320, 320, 128, 800, 128, 320, 320, 128};")
a=s.index("__device__ __forceinline__ void sw_luma4x4_fwd(")
b=s.index("__device__ __forceinline__ void sw_luma4x4_add(")
new='''__device__ __forceinline__ void sw_luma4x4_fwd(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int cat, bool dc_out, int lane, int *nr_acc4 = nullptr, int nr_on = 0)
{
    i16 c[16], lv[16];
    if (lane < 16) {
        int bx, by, r[16];
        sw_blk_xy(lane, bx, by);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                r[4 * j + i] = (int)s.fe[(by + j) * 16 + bx + i] - (int)s.fd[FDY + (by + j) * FD + bx + i];
        fwd4x4(c, r);
        if (nr_acc4 && nr_on) {
            // --nr: every coefficient but the first of every block, and the sum of magnitudes per coefficient index over the 16
            // blocks (lane i keeps index i's running sum for the whole row; added to the chain's totals at the end of the row)
#pragma unroll
            for (int i = 1; i < 16; i++) {
                int la;
                c[i] = (i16)sw_denoise(c[i], s.nr_off4[i], la);
                const int t = row_sum16(la);
                if (lane == i) *nr_acc4 += t;
            }
        }
        if (dc_out) { s.dc16[(by >> 2) * 4 + (bx >> 2)] = c[0]; c[0] = 0; }
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) s.coef[lane][i] = c[i];
        }
    }
    if (tq.on) {
        // x264_quant_4x4_trellis (R/encoder/rdo.c:641-650): a serial dynamic programme per block, walked by one lane
        WAVE_SYNC();
        if (lane == 0)
            for (int b = 0; b < 16; b++)
                td_trellis_quant(tq.r->ts, &s.coef[b][0], s.qmf[cat], tq.r->unq4[cat], d_w4z, d_zz4, tq.r->cabac, dc_out ? 1 : 2,
                                 d_trellis_lambda2[cat == 0][Q.qp], dc_out ? 1 : 0, 0, 16);
        WAVE_SYNC();
    }
    if (lane < 16) {
        const u16 *mf = s.qmf[cat], *bs = s.qbias[cat];
        const int *dq = s.qdq[cat];
        int nz = 0, bits = Q.qp / 6 - 4;
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) { c[i] = s.coef[lane][i]; nz |= c[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }
        }
        SCAN4_FRAME(lv, c);
        u32 nzm, big;
        LEVEL_MASKS(lv, nzm, big);
#pragma unroll
        for (int i = 0; i < 16; i++) { s.lv_y[16 * lane + i] = lv[i]; s.coef[lane][i] = (i16)dequant_one(c[i], dq[i], bits); }
        s.score[lane] = (nz ? (dc_out ? decimate_masks(nzm >> 1, big >> 1) : decimate_masks(nzm, big)) : 0) | ((nz != 0) << 8);
    }
    WAVE_SYNC();
}
'''
s=s[:a]+new+s[b:]

# inter luma 4x4
rep("__device__ __forceinline__ int sw_encode_inter_luma(SwLds &s, const SwArgs &a, int lane, int *nr_acc4 = nullptr, int nr_on = 0)\n{\n    if (a.lossless) return sw_ll_luma16(s, false, lane);\n    sw_luma4x4_fwd(s, a, 1, false, lane, nr_acc4, nr_on);",
    "__device__ __forceinline__ int sw_encode_inter_luma(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int lane, int *nr_acc4 = nullptr, int nr_on = 0)\n{\n    if (a.lossless) return sw_ll_luma16(s, false, lane);\n    sw_luma4x4_fwd(s, a, Q, tq, 1, false, lane, nr_acc4, nr_on);")
# i16x16
rep("__device__ __forceinline__ int sw_encode_i16x16(SwLds &s, const SwArgs &a, int lane)\n{\n    if (a.lossless) return sw_ll_luma16(s, true, lane);\n    sw_luma4x4_fwd(s, a, 0, true, lane);",
    "__device__ __forceinline__ int sw_encode_i16x16(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int lane)\n{\n    if (a.lossless) return sw_ll_luma16(s, true, lane);\n    sw_luma4x4_fwd(s, a, Q, tq, 0, true, lane);")
rep("""        const int mf = (int)s.qmf[0][0] >> 1, bias = (int)s.qbias[0][0] << 1;
        int nz = 0;
        for (int i = 0; i < 16; i++) { int q = quant_one(d[i], mf, bias); d[i] = (i16)q; nz |= q; }
        s.nnz[24] = (u8)(nz != 0);""","""        const int mf = (int)s.qmf[0][0] >> 1, bias = (int)s.qbias[0][0] << 1;
        int nz = 0;
        if (tq.on) {                                   // x264_quant_dc_trellis( .., DCT_LUMA_DC, 1 ), macroblock.c:247-248
#pragma unroll
            for (int i = 0; i < 16; i++) s.dc16[i] = d[i];
            nz = td_trellis_quant(tq.r->ts, &s.dc16[0], s.qmf[0], tq.r->unq4[0], d_w4z, d_zz4, tq.r->cabac, 0, d_trellis_lambda2[1][Q.qp], 0, 1, 16);
#pragma unroll
            for (int i = 0; i < 16; i++) d[i] = s.dc16[i];
        } else
            for (int i = 0; i < 16; i++) { int q = quant_one(d[i], mf, bias); d[i] = (i16)q; nz |= q; }
        s.nnz[24] = (u8)(nz != 0);""")
rep("            const int m = s.qdq[0][0], bits = a.qp / 6 - 6;","            const int m = s.qdq[0][0], bits = Q.qp / 6 - 6;")
open(p,'w').write(s)
print("ok")
