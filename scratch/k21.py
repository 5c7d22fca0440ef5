import sys
p='/root/repo/x264_vs2008_amd/csrc/slice_kernel.h'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:120]); sys.exit(1)
    s=s.replace(a,b)
# A: the I flow of the raster variant moves into the candidate loop of the P path
a=s.index("        if (!is_p) {\n          if constexpr (RD) {")
b=s.index("          } else {\n            analyse_intra(MX_COST_MAX);\n            type = T_I_16x16;")
s=s[:a]+"        if (!RD && !is_p) {\n          {\n"+s[b+len("          } else {\n"):]
rep('''        } else {
            // ---- motion neighbours: what cache_load puts around the block (R/common/macroblock.c:1040-1128) ----''','''        } else {
            // (The raster variant sends an I slice's macroblocks down this path too, its motion parts skipped: the candidate loop at the
            // end -- and with it the encoder, the distortion and the bit counter -- then exists ONCE in the kernel.  Two call sites of
            // the encoder made the compiler keep it as a function, and every variable it shares with the rest in scratch memory.)
            // ---- motion neighbours: what cache_load puts around the block (R/common/macroblock.c:1040-1128) ----''')
rep("            if (nb & NB_TOP) { const int o = mb - a.mb_w; rb = UNI(a.ref[o * 4 + 2]);","            if (is_p && (nb & NB_TOP)) { const int o = mb - a.mb_w; rb = UNI(a.ref[o * 4 + 2]);")
rep("            if (nb & NB_TOPRIGHT) { const int o = mb - a.mb_w + 1; rc = UNI(a.ref[o * 4 + 2]);","            if (!is_p) {}\n            else if (nb & NB_TOPRIGHT) { const int o = mb - a.mb_w + 1; rc = UNI(a.ref[o * 4 + 2]);")
rep("            if (RD || (a.flags_inter & 0x10)) {\n                // the full motion cache","            if (is_p && (RD || (a.flags_inter & 0x10))) {\n                // the full motion cache")
rep('''                if (lane < 48) { sr.cref[lane] = (signed char)cref_v; sr.cmv[lane][0] = (i16)cmvx_v; sr.cmv[lane][1] = (i16)cmvy_v; }
                WAVE_SYNC();''','''                if (is_p) {
                    if (lane < 48) { sr.cref[lane] = (signed char)cref_v; sr.cmv[lane][0] = (i16)cmvx_v; sr.cmv[lane][1] = (i16)cmvy_v; }
                    WAVE_SYNC();
                }''')
rep("            if (a.fast_pskip) {\n                if (a.subme >= 3) try_pskip = 1;","            if (is_p && a.fast_pskip) {\n                if (a.subme >= 3) try_pskip = 1;")
rep("                for (int r = 0; r < a.n_refs; r++) {\n                    int mvpx, mvpy;\n                    predict16(r, mvpx, mvpy);","                for (int r = 0; r < (is_p ? a.n_refs : 0); r++) {\n                    int mvpx, mvpy;\n                    predict16(r, mvpx, mvpy);")
# C/D: the loop
rep('''                        for (int step = 0; step < 11; step++) {
                            if (step == 0) {
                                if (!mbrd) continue;
                                cache_fenc_satd();
                                if (!(me16r == 0 && me16x == pskx && me16y == psky)) continue;
                                type = T_P_L0; part = 16;
                            } else if (step == 1) {
                                if (rd_skip) { step = 9; continue; }
                                type = T_P_L0;
                                search_partitions();
                                if (!mbrd) refine_winner();
                                WAVE_SYNC();
                                if (part == 13) sub_t_mb = sub_t;
                                PROF(2);
                                LAUNDER();
                                final_type = type; final_part = part;
                                if (a.chroma_me) {
                                    analyse_chroma();
                                    analyse_intra(i_cost - satd_chroma);
                                    satd_i16 += satd_chroma; satd_i8 += satd_chroma; satd_i4 += satd_chroma;
                                } else
                                    analyse_intra(i_cost);
                                satd_inter = i_cost; satd_intra = min(min(satd_i16, satd_i8), satd_i4);
                                if (!mbrd) { step = 9; continue; }
                                rd_isat = min(satd_inter, satd_intra); rd_thresh = rd_isat * 5 / 4;
                                type = T_P_L0;
                                continue;
                            } else if (step == 2) {''','''                        for (int step = 0; step < 11; step++) {
                            bool fin = false;
                            if (step == 0) {
                                if (!mbrd) continue;
                                cache_fenc_satd();
                                if (!is_p || !(me16r == 0 && me16x == pskx && me16y == psky)) continue;
                                type = T_P_L0; part = 16;
                            } else if (step == 1) {
                                if (rd_skip) { step = 9; continue; }
                                int intra_thresh = MX_COST_MAX;              // an I slice: x264_mb_analyse_intra(h, &analysis, COST_MAX), analyse.c:2175
                                if (is_p) {
                                    type = T_P_L0;
                                    search_partitions();
                                    if (!mbrd) refine_winner();
                                    WAVE_SYNC();
                                    if (part == 13) sub_t_mb = sub_t;
                                    PROF(2);
                                    LAUNDER();
                                    final_type = type; final_part = part;
                                    intra_thresh = i_cost;
                                    if (a.chroma_me) { analyse_chroma(); intra_thresh = i_cost - satd_chroma; }
                                }
                                analyse_intra(intra_thresh);
                                if (is_p && a.chroma_me) { satd_i16 += satd_chroma; satd_i8 += satd_chroma; satd_i4 += satd_chroma; }
                                satd_inter = i_cost; satd_intra = min(min(satd_i16, satd_i8), satd_i4);
                                if (!mbrd) { step = 9; continue; }
                                rd_isat = min(satd_inter, satd_intra); rd_thresh = rd_isat * 5 / 4;
                                type = T_P_L0;
                                if (!is_p) step = 6;                         // an I slice: straight to x264_intra_rd (:2177)
                                continue;
                            } else if (step == 2) {''')
rep('''                            } else if (step == 7) {                                                // x264_intra_rd, :845-874
                                if (!(satd_i16 <= satd_inter * 5 / 4)) { satd_i16 = MX_COST_MAX; continue; }
                                type = T_I_16x16;
                            } else if (step == 8) {
                                if (!(satd_i4 <= satd_inter * 5 / 4 && satd_i4 < MX_COST_MAX)) { satd_i4 = MX_COST_MAX; continue; }
                                type = T_I_4x4;
                            } else if (step == 9) {
                                if (!(satd_i8 <= satd_inter * 5 / 4 && satd_i8 < MX_COST_MAX)) { satd_i8 = MX_COST_MAX; continue; }
                                type = T_I_8x8;
                            } else {
                                if (rd_skip) type = T_P_SKIP;
                                else {''','''                            } else if (step == 7) {                                                // x264_intra_rd, :845-874 (threshold COST_MAX in an I slice)
                                if (!(satd_i16 <= (is_p ? satd_inter * 5 / 4 : MX_COST_MAX))) { satd_i16 = MX_COST_MAX; continue; }
                                type = T_I_16x16;
                            } else if (step == 8) {
                                if (!(satd_i4 <= (is_p ? satd_inter * 5 / 4 : MX_COST_MAX) && satd_i4 < MX_COST_MAX)) { satd_i4 = MX_COST_MAX; continue; }
                                type = T_I_4x4;
                            } else if (step == 9) {
                                if (!(satd_i8 <= (is_p ? satd_inter * 5 / 4 : MX_COST_MAX) && satd_i8 < MX_COST_MAX)) { satd_i8 = MX_COST_MAX; continue; }
                                type = T_I_8x8;
                            } else {
                                fin = true;
                                if (!is_p) {                                 // analyse.c:2179-2184: 16x16, then 4x4, then 8x8, then PCM on strict improvement
                                    type = T_I_16x16;
                                    int ic = satd_i16;
                                    if (satd_i4 < ic) { ic = satd_i4; type = T_I_4x4; }
                                    if (satd_i8 < ic) { ic = satd_i8; type = T_I_8x8; }
                                    if (satd_pcm < ic) type = T_I_PCM;
                                } else if (rd_skip) type = T_P_SKIP;
                                else {''')
rep('''                                    stat_inter = i_cost;
                                    if (mbrd && !IS_INTRA_T(type)) update_cache_p();              // x264_analyse_update_cache, :2763
                                }
                                tq.on = rd.trellis != 0;                                          // :2768-2773
                                if (rd.trellis == 1 || a.nr) skip_intra = 0;
                                PROF(6);
                                if (type != T_I_PCM) encode_mb(1);
                                encoded = true;
                                break;
                            }
                            if (!IS_INTRA_T(type)) update_cache_p();
                            const int c = rd_cost_mb();
                            if (step == 0)''','''                                    stat_inter = i_cost;
                                }
                                tq.on = rd.trellis != 0;                                          // :2768-2773
                                if (rd.trellis == 1 || a.nr) skip_intra = 0;
                            }
                            // x264_analyse_update_cache (:2763 for the final type), then the encoder: the trial of x264_rd_cost_mb
                            // (R/encoder/rdo.c:139-171) or the real thing
                            if ((!fin || mbrd) && !IS_INTRA_T(type)) update_cache_p();
                            const int t8_bak = t8;
                            PROF(6);
                            if (!(fin && type == T_I_PCM)) encode_mb(fin ? 1 : 0);
                            if (fin) { encoded = true; break; }
                            PROF(0);
                            // distortion, and the syntax priced against a copy of the live contexts.  Like the reference this leaves `type`
                            // as the encode left it (P_SKIP when nothing was left to code on the skip vector).
                            int c = ssd_mb();
                            if (type == T_P_SKIP) c += (Q.lambda2 + 128) >> 8;
                            else {
                                syn_prepare();
                                for (int k = lane; k < 460; k += 64) sr.cabac_tmp[k] = sr.cabac[k];
                                const MbSynDev y0 = make_syn();
                                WAVE_SYNC();
                                if (lane == 0) {
                                    DCabac tcb = {0, 0x1FE, -1, 0, nullptr, 0};
                                    MbSynDev y = y0;
                                    cw_macroblock(tcb, sr.cabac_tmp, 1, y, s.fe, 0);
                                    sr.tmp_i[0] = tcb.f8;
                                }
                                WAVE_SYNC();
                                const int f8 = UNI(sr.tmp_i[0]);
                                c += (int)(((unsigned long long)(u32)f8 * (u32)Q.lambda2 + 32768) >> 16);
                            }
                            t8 = t8_bak;
                            PROF(7);
                            if (step == 0)''')
# the rd_cost_mb lambda goes
a=s.index("        // x264_rd_cost_mb (R/encoder/rdo.c:139-171): trial encode, distortion, the syntax priced against a copy of the live contexts.\n")
b=s.index("        (void)cache_fenc_satd; (void)rd_cost_mb;")
s=s[:a]+"        (void)cache_fenc_satd; (void)ssd_mb;"+s[b+len("        (void)cache_fenc_satd; (void)rd_cost_mb;"):]
open(p,'w').write(s)
print("ok")
