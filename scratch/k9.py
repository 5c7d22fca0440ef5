import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep("    int aq, qp_min, qp_max;\n    float f_qpm;","    int aq, qp_min, qp_max, chroma_qp_offset;\n    float f_qpm;")
rep('''        if (mbx > 0 && mby > 0) { nb |= NB_TOPLEFT; type_topleft = UNI(a.mb_type[mb - a.mb_w - 1]); }

        int type = T_I_16x16,''','''        if (mbx > 0 && mby > 0) { nb |= NB_TOPLEFT; type_topleft = UNI(a.mb_type[mb - a.mb_w - 1]); }
        int cbp_top = -1, cpm_top = 0, t8_top = 0;
        if constexpr (RD) {
            // ---- x264_ratecontrol_qp + x264_adaptive_quant (R/encoder/analyse.c:2162-2164, ratecontrol.c:257-265) ----
            int qp = a.qp;
            if (rd.aq) {
                const float off = __builtin_bit_cast(float, UNI(__builtin_bit_cast(int, rd.aq_offset[cb + mb])));
                qp = clip3((int)((double)(rd.f_qpm + off) + .5), rd.qp_min, rd.qp_max);
                if (iabs(qp - last_qp) == 1) qp = last_qp;
            }
            if (qp != Q.qp) {
                Q.qp = qp; Q.qpc = d_chroma_qp[clip3(qp + rd.chroma_qp_offset, 0, 51)];
                Q.lambda = d_lambda_tab[qp]; Q.lambda2 = d_lambda2_tab[qp]; Q.skip_thresh = (d_lambda2_tab[Q.qpc] + 32) >> 6;
                cost_g = rd.cost_mv_all + (size_t)qp * (2 * a.cost_center + 1) + a.cost_center;
                WAVE_SYNC();
                load_qp_tables(lane);
                WAVE_SYNC();
            }
            // ---- what the entropy coder reads of the neighbours (R/common/macroblock.c:896-1010,1129-1160) ----
            if (lane < 48) { sr.cmvd[lane][0] = 0; sr.cmvd[lane][1] = 0; }
            WAVE_SYNC();
            if (nb & NB_TOP) {
                const int top = mb - a.mb_w;
                const u8 *nz = (a.nnz + 27 * cb) + (size_t)top * 27;
                cbp_top = UNI((a.cbp + cb)[top]); t8_top = UNI((a.t8 + cb)[top]);
                { const int ct = UNI((a.chroma_mode + cb)[top]); cpm_top = type_top == T_I_PCM ? 0 : sw_fix8c(ct); }
                if (lane < 4) sr.nz_t[lane] = nz[lane == 0 ? 10 : lane == 1 ? 11 : lane == 2 ? 14 : 15];
                else if (lane < 8) sr.nz_tc[(lane - 4) >> 1][lane & 1] = nz[16 + 4 * ((lane - 4) >> 1) + 2 + (lane & 1)];
                else if (lane < 12) {
                    const i16 *mvd = rd.mvd + ((cb + top) * 16 + 12 + (lane - 8)) * 2;
                    sr.cmvd[4 + lane - 8][0] = mvd[0]; sr.cmvd[4 + lane - 8][1] = mvd[1];
                }
            } else if (lane < 4) sr.nz_t[lane] = 0x80;
            else if (lane < 8) sr.nz_tc[(lane - 4) >> 1][lane & 1] = 0x80;
            if (nb & NB_LEFT) {
                if (lane >= 16 && lane < 20) sr.nz_l[lane - 16] = sr.left_nz[lane - 16];
                else if (lane >= 20 && lane < 24) sr.nz_lc[(lane - 20) >> 1][lane & 1] = sr.left_nz[4 + lane - 20];
                else if (lane >= 24 && lane < 28) { sr.cmvd[11 + 8 * (lane - 24)][0] = sr.left_mvd[lane - 24][0]; sr.cmvd[11 + 8 * (lane - 24)][1] = sr.left_mvd[lane - 24][1]; }
            } else if (lane >= 16 && lane < 20) sr.nz_l[lane - 16] = 0x80;
            else if (lane >= 20 && lane < 24) sr.nz_lc[(lane - 20) >> 1][lane & 1] = 0x80;
            WAVE_SYNC();
        }

        int type = T_I_16x16,''')
open(p,'w').write(s)
print("ok")
