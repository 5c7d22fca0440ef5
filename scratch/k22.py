import sys
p='/root/repo/x264_vs2008_amd/csrc/slice_kernel.h'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:120]); sys.exit(1)
    s=s.replace(a,b)
TW="tq.r->tw, (u32 *)s.patch"
rep('''        // x264_quant_4x4_trellis (R/encoder/rdo.c:641-650): a serial dynamic programme per block, walked by one lane
        WAVE_SYNC();
        if (lane == 0)
            for (int b = 0; b < 16; b++)
                td_trellis_quant(tq.r->ts, &s.coef[b][0], s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz4, tq.r->cabac, dc_out ? 1 : 2,
                                 d_trellis_lambda2[cat == 0][Q.qp], dc_out ? 1 : 0, 0, 16);
        WAVE_SYNC();''','''        // x264_quant_4x4_trellis (R/encoder/rdo.c:641-650): four blocks at a time, sixteen lanes each (trellis_wave.h)
        WAVE_SYNC();
#pragma nounroll
        for (int it = 0; it < 4; it++)
            td_trellis_wave(%s, &s.coef[4 * it + (lane >> 4)][0], true, s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz4, tq.r->cabac, dc_out ? 1 : 2,
                            d_trellis_lambda2[cat == 0][Q.qp], dc_out ? 1 : 0, 0, 16, lane);
        WAVE_SYNC();''' % TW)
# I16 DC
rep('''    sw_luma4x4_fwd(s, a, Q, tq, 0, true, lane);
    if (lane == 0) {
        const int b_decimate = a.dct_decimate && a.slice_type == 0;
        int score = b_decimate ? 0 : 9, cbp = 0;''','''    sw_luma4x4_fwd(s, a, Q, tq, 0, true, lane);
    i16 d[16], t[16];
    int nz = 0, cbp = 0;
    if (lane == 0) {
        const int b_decimate = a.dct_decimate && a.slice_type == 0;
        int score = b_decimate ? 0 : 9;''')
rep('''        // dct4x4dc (R/common/dct.c:39-71), quant_4x4_dc, scan, idct4x4dc, dequant_4x4_dc (quant.c:151-178)
        i16 d[16], t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = s.dc16[i];''','''        // dct4x4dc (R/common/dct.c:39-71), quant_4x4_dc, scan, idct4x4dc, dequant_4x4_dc (quant.c:151-178)
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = s.dc16[i];''')
rep('''        const int mf = (int)s.qmf[0][0] >> 1, bias = (int)s.qbias[0][0] << 1;
        int nz = 0;
        if (tq.on) {                                   // x264_quant_dc_trellis( .., DCT_LUMA_DC, 1 ), macroblock.c:247-248
#pragma unroll
            for (int i = 0; i < 16; i++) s.dc16[i] = d[i];
            nz = td_trellis_quant(tq.r->ts, &s.dc16[0], s.qmf[0], tq.r->unq4[0], tq.r->w4z, tq.r->zz4, tq.r->cabac, 0, d_trellis_lambda2[1][Q.qp], 0, 1, 16);
#pragma unroll
            for (int i = 0; i < 16; i++) d[i] = s.dc16[i];
        } else
            for (int i = 0; i < 16; i++) { int q = quant_one(d[i], mf, bias); d[i] = (i16)q; nz |= q; }''','''        const int mf = (int)s.qmf[0][0] >> 1, bias = (int)s.qbias[0][0] << 1;
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) s.dc16[i] = d[i];
        } else
            for (int i = 0; i < 16; i++) { int q = quant_one(d[i], mf, bias); d[i] = (i16)q; nz |= q; }
    }
    if (tq.on) {                                       // x264_quant_dc_trellis( .., DCT_LUMA_DC, 1 ), macroblock.c:247-248
        WAVE_SYNC();
        td_trellis_wave(%s, &s.dc16[0], lane < 16, s.qmf[0], tq.r->unq4[0], tq.r->w4z, tq.r->zz4, tq.r->cabac, 0, d_trellis_lambda2[1][Q.qp], 0, 1, 16, lane);
        WAVE_SYNC();
    }
    if (lane == 0) {
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) { d[i] = s.dc16[i]; nz |= d[i]; }
        }''' % TW)
# chroma AC
rep('''        WAVE_SYNC();
        if (lane == 0)
            for (int b = 0; b < 8; b++)
                td_trellis_quant(tq.r->ts, &s.ccoef[b][0], s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz4, tq.r->cabac, 4, d_trellis_lambda2[!b_inter][Q.qpc], 1, 0, 16);
        WAVE_SYNC();''','''        WAVE_SYNC();
#pragma nounroll
        for (int it = 0; it < 2; it++)
            td_trellis_wave(%s, &s.ccoef[4 * it + (lane >> 4)][0], true, s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz4, tq.r->cabac, 4, d_trellis_lambda2[!b_inter][Q.qpc], 1, 0, 16, lane);
        WAVE_SYNC();''' % TW)
# chroma DC
rep('''        WAVE_SYNC();
        if (lane == 0)
            for (int ch = 0; ch < 2; ch++)
                td_trellis_quant(tq.r->ts, &s.cdcout[4 * ch], s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz2, tq.r->cabac, 3, d_trellis_lambda2[!b_inter][Q.qpc], 0, 1, 4);
        WAVE_SYNC();''','''        WAVE_SYNC();
        td_trellis_wave(%s, &s.cdcout[4 * ((lane >> 4) & 1)], lane < 32, s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz2, tq.r->cabac, 3, d_trellis_lambda2[!b_inter][Q.qpc], 0, 1, 4, lane);
        WAVE_SYNC();''' % TW)
# luma 8x8
rep('''    if (tq.on) {                                      // x264_quant_8x8_trellis (R/encoder/rdo.c:652-660), one lane
        if (lane == 0)
            for (int j = 0; j < 4; j++)
                if ((mask >> j) & 1)
                    td_trellis_quant(tq.r->ts, coef + 64 * j, s.q8mf[cat], tq.r->unq8[cat], tq.r->w8z, tq.r->zz8, tq.r->cabac, 5, d_trellis_lambda2[cat == 0][Q.qp], 0, 0, 64);
        WAVE_SYNC();''','''    if (tq.on) {                                      // x264_quant_8x8_trellis (R/encoder/rdo.c:652-660): one block at a time (its level lists fill the scratch area), sixteen lanes
#pragma nounroll
        for (int j = 0; j < 4; j++)
            if ((mask >> j) & 1)
                td_trellis_wave(%s, coef + 64 * j, lane < 16, s.q8mf[cat], tq.r->unq8[cat], tq.r->w8z, tq.r->zz8, tq.r->cabac, 5, d_trellis_lambda2[cat == 0][Q.qp], 0, 0, 64, lane);
        WAVE_SYNC();''' % TW)
# i4x4
rep('''        if (lane == 0) td_trellis_quant(tq.r->ts, &s.coef[idx][0], s.qmf[0], tq.r->unq4[0], tq.r->w4z, tq.r->zz4, tq.r->cabac, 2, d_trellis_lambda2[1][Q.qp], 0, 0, 16);''',
    '''        td_trellis_wave(%s, &s.coef[idx][0], lane < 16, s.qmf[0], tq.r->unq4[0], tq.r->w4z, tq.r->zz4, tq.r->cabac, 2, d_trellis_lambda2[1][Q.qp], 0, 0, 16, lane);''' % TW)
rep("TrellisScratch ts;","TdWave tw;")
rep('#include "trellis_dev.h"','#include "trellis_wave.h"')
open(p,'w').write(s)
print('ok')
