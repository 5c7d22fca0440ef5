import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep("__device__ __forceinline__ int sw_encode_chroma(SwLds &s, const SwArgs &a, int b_inter, int lane)",
    "__device__ __forceinline__ int sw_encode_chroma(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int b_inter, int lane)")
rep('''    if (lane < 8) {
        int ch = lane >> 2, i4 = lane & 3, bx = (i4 & 1) * 4, by = (i4 >> 1) * 4, r[16];
        const u8 *fe = s.fe + 256 + 64 * ch, *pr = s.fd + (ch ? FDV : FDU);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                r[4 * j + i] = (int)fe[(by + j) * 8 + bx + i] - (int)pr[(by + j) * FD + bx + i];
        i16 c[16], lv[16];
        fwd4x4(c, r);
        s.cdc[lane] = c[0];
        c[0] = 0;                                     // dct2x2dc takes the DCs out (macroblock.c:73-85)
        const int *dq = s.qdq[cat];
        int nz = 0, bits = a.qpc / 6 - 4;
#pragma unroll
        for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }
        SCAN4_FRAME(lv, c);''','''    i16 c[16], lv[16];
    if (lane < 8) {
        int ch = lane >> 2, i4 = lane & 3, bx = (i4 & 1) * 4, by = (i4 >> 1) * 4, r[16];
        const u8 *fe = s.fe + 256 + 64 * ch, *pr = s.fd + (ch ? FDV : FDU);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                r[4 * j + i] = (int)fe[(by + j) * 8 + bx + i] - (int)pr[(by + j) * FD + bx + i];
        fwd4x4(c, r);
        s.cdc[lane] = c[0];
        c[0] = 0;                                     // dct2x2dc takes the DCs out (macroblock.c:73-85)
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) s.ccoef[lane][i] = c[i];
        }
    }
    if (tq.on) {                                      // x264_quant_4x4_trellis( .., DCT_CHROMA_AC, !b_inter, 0 ), macroblock.c:310-311
        WAVE_SYNC();
        if (lane == 0)
            for (int b = 0; b < 8; b++)
                td_trellis_quant(tq.r->ts, &s.ccoef[b][0], s.qmf[cat], tq.r->unq4[cat], d_w4z, d_zz4, tq.r->cabac, 4, d_trellis_lambda2[!b_inter][Q.qpc], 1, 0, 16);
        WAVE_SYNC();
    }
    if (lane < 8) {
        const int *dq = s.qdq[cat];
        int nz = 0, bits = Q.qpc / 6 - 4;
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) { c[i] = s.ccoef[lane][i]; nz |= c[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }
        }
        SCAN4_FRAME(lv, c);''')
rep('''    if (lane < 2) {
        const int ch = lane;
        int b0 = s.cdc[4 * ch], b1 = s.cdc[4 * ch + 1], b2 = s.cdc[4 * ch + 2], b3 = s.cdc[4 * ch + 3];
        int a0 = b0 + b1, a1 = b2 + b3, a2 = b0 - b1, a3 = b2 - b3;
        i16 d2[4] = {(i16)(a0 + a1), (i16)(a0 - a1), (i16)(a2 + a3), (i16)(a2 - a3)};   // [0][0] [0][1] [1][0] [1][1]
        int nz_dc = 0;
        for (int i = 0; i < 4; i++) { int q = quant_one(d2[i], (int)mf[0] >> 1, (int)bs[0] << 1); d2[i] = (i16)q; nz_dc |= q; }''','''    i16 d2[4] = {0, 0, 0, 0};                          // [0][0] [0][1] [1][0] [1][1]
    if (lane < 2) {
        const int ch = lane;
        int b0 = s.cdc[4 * ch], b1 = s.cdc[4 * ch + 1], b2 = s.cdc[4 * ch + 2], b3 = s.cdc[4 * ch + 3];
        int a0 = b0 + b1, a1 = b2 + b3, a2 = b0 - b1, a3 = b2 - b3;
        d2[0] = (i16)(a0 + a1); d2[1] = (i16)(a0 - a1); d2[2] = (i16)(a2 + a3); d2[3] = (i16)(a2 - a3);
        if (tq.on) { s.cdcout[4 * ch] = d2[0]; s.cdcout[4 * ch + 1] = d2[1]; s.cdcout[4 * ch + 2] = d2[2]; s.cdcout[4 * ch + 3] = d2[3]; }
    }
    if (tq.on) {                                      // x264_quant_dc_trellis( .., DCT_CHROMA_DC, !b_inter ), macroblock.c:325-326
        WAVE_SYNC();
        if (lane == 0)
            for (int ch = 0; ch < 2; ch++)
                td_trellis_quant(tq.r->ts, &s.cdcout[4 * ch], s.qmf[cat], tq.r->unq4[cat], d_w4z, d_zz2, tq.r->cabac, 3, d_trellis_lambda2[!b_inter][Q.qpc], 0, 1, 4);
        WAVE_SYNC();
    }
    if (lane < 2) {
        const int ch = lane;
        int nz_dc = 0;
        if (tq.on) { for (int i = 0; i < 4; i++) { d2[i] = s.cdcout[4 * ch + i]; nz_dc |= d2[i]; } }
        else for (int i = 0; i < 4; i++) { int q = quant_one(d2[i], (int)mf[0] >> 1, (int)bs[0] << 1); d2[i] = (i16)q; nz_dc |= q; }''')
rep("        int dmf = s.qdq[cat][0], qbits = a.qpc / 6 - 5;","        int dmf = s.qdq[cat][0], qbits = Q.qpc / 6 - 5;")
# probe_pskip
rep("__device__ __forceinline__ int sw_probe_pskip(SwLds &s, const SwRefs &refs, const SwArgs &a, int pmx, int pmy, int mbx, int mby,",
    "__device__ __forceinline__ int sw_probe_pskip(SwLds &s, const SwRefs &refs, const SwArgs &a, const SwQp &Q, int pmx, int pmy, int mbx, int mby,")
rep("        if (c_ssd[ch] < a.chroma_skip_thresh) continue;","        if (c_ssd[ch] < Q.skip_thresh) continue;")
open(p,'w').write(s)
print("ok")
