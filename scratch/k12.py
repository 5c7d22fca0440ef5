import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
a=s.index("        // ---- x264_analyse_update_cache + x264_macroblock_encode ----\n        int cbp_luma = 0, cbp_chroma = 0;")
b=s.index("        const int intra = IS_INTRA_T(type);\n        if (cbp_luma == 0 && type != T_I_8x8) t8 = 0;")
tail=s[a:b]
new_lambda='''        // ---- x264_macroblock_encode (R/encoder/macroblock.c:475-790) of the macroblock as type / part / t8 / the intra modes / s.mv4 /
        // s.ref8 describe it now.  The final encode, and with the RD levels every trial encode of x264_rd_cost_mb (final_pass = 0).
        int cbp_luma = 0, cbp_chroma = 0;
        auto encode_mb = [&](int final_pass) {
            cbp_luma = 0; cbp_chroma = 0;
            if (lane < 32) s.nnz[lane] = 0;
            WAVE_SYNC();
            if (type == T_P_SKIP) {
                mvx = pskx; mvy = psky; ref = 0;
                if (lane < 16) { s.mv4[lane][0] = (i16)pskx; s.mv4[lane][1] = (i16)psky; }
                if (lane < 4) s.ref8[lane] = 0;
                WAVE_SYNC();
                if (!skip_mc) {
                    const int vx = clip3(mvx, 4 * (-16 * mbx - 24), 4 * (16 * (a.mb_w - mbx - 1) + 24));
                    const int vy = clip3(mvy, 4 * (-16 * mby - 24), 4 * (16 * (a.mb_h - mby - 1) + 24));
                    sw_mc16(s, refs, a, 0, vx, vy, oy, oc, by_, bc_, lane, true);
                    WAVE_SYNC();
                }
                return;
            }
            if (type == T_I_16x16) {
                t8 = 0;
                analyse_chroma();
                sw_pred16(s, pred16, lane, a.lossless);
                cbp_luma = sw_encode_i16x16(s, a, Q, tq, lane);
                sw_pred8c(s, predc, lane, a.lossless);
                cbp_chroma = sw_encode_chroma(s, a, Q, tq, 0, lane);
            } else if (type == T_I_8x8 || type == T_I_4x4) {
                // x264_analyse_update_cache: the winner's modes into the cache; then macroblock.c:527-590.  With i_skip_intra the
                // analysis already encoded all blocks but the last: take its state and finish; without it (trellis 1, --nr,
                // lossless) every block is predicted and coded again.
                const bool i8 = type == T_I_8x8;
                if (lane < 16) s.i4c[sw_scan8(lane)] = i8 ? s.pred8[lane >> 2] : s.pred4[lane];
                analyse_chroma();
                if (skip_intra) {
                    *(u32 *)(s.fd + FDY + (lane >> 2) * FD + (lane & 3) * 4) = *(const u32 *)((i8 ? s.i8_fdec : s.i4_fdec) + lane * 4);
                    if (lane < 16) s.nnz[lane] = i8 ? s.i8_nnz[lane] : s.i4_nnz[lane];
                    cbp_luma = i8 ? i8_cbp : i4_cbp;
                    if constexpr (RD) {                  // "In RD mode, restore the now-overwritten DCT data", macroblock.c:543
                        if (skip_intra == 2) for (int k = lane; k < 256; k += 64) { if (i8) s.lv_y8[k] = sr.i8_dct[k]; else s.lv_y[k] = sr.i4_dct[k]; }
                    }
                }
                WAVE_SYNC();
                if (i8) {
                    t8 = 1;
                    for (int idx = skip_intra ? 3 : 0; idx < 4; idx++) {
                        const int bx = 8 * (idx & 1), by = 8 * (idx >> 1);
                        const int mode = __builtin_amdgcn_readfirstlane((int)s.pred8[idx]), nb8 = sw_nb8(idx, nb);
                        // x264_pred_i4x4_neighbors (R/common/macroblock.h:40-54)
                        const int need = mode == 0 || mode == 10 ? NB_TOP : mode == 1 || mode == 8 || mode == 9 ? NB_LEFT : mode == 2 ? NB_LEFT | NB_TOP
                                       : mode == 3 || mode == 7 ? NB_TOP | NB_TOPRIGHT : mode == 11 ? 0 : NB_LEFT | NB_TOPLEFT | NB_TOP;
                        if (lane == 0) pred8_filter(s.edge8, s.fd + FDY + by * FD + bx, FD, nb8, need);
                        WAVE_SYNC();
                        const int v = a.lossless && mode < 2 ? sw_ll_px(s, 0, mode, bx + (lane & 7), by + (lane >> 3)) : pred8_px(mode, s.edge8, lane & 7, lane >> 3);
                        WAVE_SYNC();
                        s.fd[FDY + (by + (lane >> 3)) * FD + bx + (lane & 7)] = (u8)v;
                        WAVE_SYNC();
                        sw_encode_i8x8(s, a, Q, tq, idx, cbp_luma, lane);
                    }
                } else {
                    t8 = 0;
                    for (int idx = skip_intra ? 15 : 0; idx < 16; idx++) {
                        int bx, by;
                        sw_blk_xy(idx, bx, by);
                        u8 *dst = s.fd + FDY + by * FD + bx;
                        const int mode = __builtin_amdgcn_readfirstlane((int)s.pred4[idx]);
                        if ((sw_nb4(idx, nb) & (NB_TOPRIGHT | NB_TOP)) == NB_TOP && lane < 4) dst[4 - FD + lane] = dst[3 - FD];
                        WAVE_SYNC();
                        if (lane < 13) pred4_edges(s.e4, dst, FD, lane);
                        WAVE_SYNC();
                        if (lane < 16) dst[(lane >> 2) * FD + (lane & 3)] = (u8)(a.lossless && mode < 2 ? sw_ll_px(s, 0, mode, bx + (lane & 3), by + (lane >> 2))
                                                                                                       : pred4_px(mode, s.e4, lane & 3, lane >> 2));
                        WAVE_SYNC();
                        sw_encode_i4x4(s, a, Q, tq, idx, cbp_luma, lane);
                    }
                }
                sw_pred8c(s, predc, lane, a.lossless);
                cbp_chroma = sw_encode_chroma(s, a, Q, tq, 0, lane);
            } else {
                sw_mc_parts(s, refs, a, oy, oc, by_, bc_, lane);
                WAVE_SYNC();
                // x264_mb_transform_8x8_allowed: a P_8x8 macroblock only with four 8x8 sub-partitions
                if (!mbrd && a.transform8x8 && !a.lossless && (type != T_P_8x8 || __ballot(lane < 4 && sub_t_mb != 3) == 0)) {
                    // x264_mb_analyse_transform (R/encoder/analyse.c:2109-2126): SA8D against SATD of the 16x16 prediction error
                    int raw = 0;
                    if (lane < 32) {
                        const int blk = lane >> 3, r = lane & 7;
                        raw = sw_sa8d_rows(s.fe + ((blk >> 1) * 8 + r) * 16 + (blk & 1) * 8, s.fd + FDY + ((blk >> 1) * 8 + r) * FD + (blk & 1) * 8, lane);
                    }
                    const int c8 = (__builtin_amdgcn_readlane(raw, 0) + __builtin_amdgcn_readlane(raw, 8) + __builtin_amdgcn_readlane(raw, 16)
                                    + __builtin_amdgcn_readlane(raw, 24) + 2) >> 2;
                    const int c4 = sw_cmp_luma16(s, 1, lane);
                    t8 = c8 < c4;
                }
                const int nr_on = a.nr && final_pass;        // h->mb.b_noise_reduction is off while analysing (analyse.c:237,2769)
                if (nr_on) { if (t8) nr_n8 += 4; else nr_n4 += 16; }
                cbp_luma = t8 ? sw_encode_inter_luma8(s, a, Q, tq, lane, &nr_acc8, nr_on) : sw_encode_inter_luma(s, a, Q, tq, lane, &nr_acc4, nr_on);   // never a conditional pointer: that pins the counter in scratch memory
                cbp_chroma = sw_encode_chroma(s, a, Q, tq, 1, lane);
                if (type == T_P_L0 && part == 16 && !(cbp_luma | cbp_chroma) && mvx == pskx && mvy == psky && ref == 0) type = T_P_SKIP;
            }
        };
'''
# the tail becomes a call in the non-RD variant
s=s[:a]+"        // ---- x264_analyse_update_cache + x264_macroblock_encode ----\n        if constexpr (!RD) encode_mb(1);\n"+s[b:]
# place lambda before the I/P branch: find "        if (!is_p) {\n            analyse_intra(MX_COST_MAX);"
k="        if (!is_p) {\n            analyse_intra(MX_COST_MAX);"
assert s.count(k)==1
s=s.replace(k,new_lambda+"\n"+k)
open(p,'w').write(s)
print("ok")
