import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)

# split P_SKIP encode out of encode_mb
rep('''        auto encode_mb = [&](int final_pass) {
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
            }''','''        bool encoded = false;               // (RD) the final encode has run inside the candidate loop
        auto encode_pskip = [&]() {         // x264_macroblock_encode_pskip, macroblock.c:378-402
            cbp_luma = 0; cbp_chroma = 0;
            if (lane < 32) s.nnz[lane] = 0;
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
        };
        auto encode_mb = [&](int final_pass) {
            if (type == T_P_SKIP) { encode_pskip(); return; }
            cbp_luma = 0; cbp_chroma = 0;
            if (lane < 32) s.nnz[lane] = 0;
            WAVE_SYNC();''')
rep("        if constexpr (!RD) encode_mb(1);\n","        if constexpr (!RD) encode_mb(1);\n        else if (!encoded) encode_pskip();                     // the fast / early P_SKIP exits of the analysis\n")
# I flow: mark encoded
rep("                    if (type != T_I_PCM) encode_mb(1);\n                    break;","                    if (type != T_I_PCM) encode_mb(1);\n                    encoded = true;\n                    break;")

pflow='''// ---- the raster variant's P macroblock (analyse.c:2228-2405): the rest of the analysis, the RD candidates of
                        // x264_mb_analyse_p_rd / x264_mb_analyse_transform_rd / x264_intra_rd, and the final encode, through ONE copy of
                        // x264_rd_cost_mb: step 0 the early 16x16 trial (:1134-1143), 1 the analysis, 2-5 p_rd, 6 the transform, 7-9 intra, 10 final.
                        const int me16x = mvx, me16y = mvy, me16r = ref;
                        int rd16 = MX_COST_MAX, satd_inter = 0, satd_intra = 0, final_type = T_P_L0, final_part = 16, rd_thresh = 0, rd_isat = 0;
                        bool rd_skip = false;
                        // x264_analyse_update_cache for a P candidate (analyse.c:2803-2846): type / part -> s.mv4 / s.ref8 (and the 16x16 scalars)
                        auto update_cache_p = [&]() {
                            if (type == T_P_SKIP) return;                        // encode_pskip sets the skip vector itself
                            const int bx4 = lane & 3, by4 = (lane >> 2) & 3, bx8 = lane & 1, by8 = (lane >> 1) & 1;
                            const int slot = part == 14 ? 4 + (by4 >> 1) : part == 15 ? 6 + (bx4 >> 1) : (by4 >> 1) * 2 + (bx4 >> 1);
                            const int slot8 = part == 14 ? 4 + by8 : part == 15 ? 6 + bx8 : by8 * 2 + bx8;
                            int vx = __shfl(pme_v, slot * 8 + 0, 64), vy = __shfl(pme_v, slot * 8 + 1, 64), vr = __shfl(pme_v, slot8 * 8 + 4, 64);
                            if (part == 16) { vx = me16x; vy = me16y; vr = me16r; }
                            if (lane < 16) { s.mv4[lane][0] = (i16)vx; s.mv4[lane][1] = (i16)vy; }
                            if (lane < 4) s.ref8[lane] = (signed char)vr;
                            mvx = me16x; mvy = me16y; ref = me16r;
                            WAVE_SYNC();
                        };
#pragma nounroll
                        for (int step = 0; step < 11; step++) {
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
                            } else if (step == 2) {
                                if (!(rd16 == MX_COST_MAX && best <= rd_isat * 3 / 2)) continue;
                                part = 16;
                            } else if (step == 3) {
                                if (!(c16x8 <= rd_thresh)) { c16x8 = MX_COST_MAX; continue; }
                                part = 14;
                            } else if (step == 4) {
                                if (!(c8x16 <= rd_thresh)) { c8x16 = MX_COST_MAX; continue; }
                                part = 15;
                            } else if (step == 5) {
                                if (!(c8x8 <= rd_thresh)) { c8x8 = MX_COST_MAX; continue; }
                                type = T_P_8x8; part = 13;
                            } else if (step == 6) {
                                final_type = T_P_L0; final_part = 16; i_cost = rd16;
                                if (c16x8 < i_cost) { i_cost = c16x8; final_part = 14; }
                                if (c8x16 < i_cost) { i_cost = c8x16; final_part = 15; }
                                if (c8x8 < i_cost) { i_cost = c8x8; final_part = 13; final_type = T_P_8x8; }
                                type = final_type; part = final_part;
                                if (!(i_cost < MX_COST_MAX) || !a.transform8x8) continue;        // x264_mb_analyse_transform_rd, :2127-2150
                                t8 = !t8;
                            } else if (step == 7) {                                                // x264_intra_rd, :845-874
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
                                else {
                                    // analyse.c:2391-2404: best intra type (16x16, then 8x8, then 4x4, then PCM on strict improvement) against inter
                                    int itype = T_I_16x16, icost = satd_i16;
                                    if (satd_i8 < icost) { icost = satd_i8; itype = T_I_8x8; }
                                    if (satd_i4 < icost) { icost = satd_i4; itype = T_I_4x4; }
                                    if (satd_pcm < icost) { icost = satd_pcm; itype = T_I_PCM; }
                                    type = final_type; part = final_part;
                                    if (icost < i_cost) { i_cost = icost; type = itype; }
                                    if (icost == MX_COST_MAX) icost = i_cost * satd_intra / satd_inter + 1;
                                    stat_intra = icost; analysed = 1;
                                    stat_inter = i_cost;
                                    if (mbrd && !IS_INTRA_T(type)) update_cache_p();              // x264_analyse_update_cache, :2763
                                }
                                tq.on = rd.trellis != 0;                                          // :2768-2773
                                if (rd.trellis == 1 || a.nr) skip_intra = 0;
                                if (type != T_I_PCM) encode_mb(1);
                                encoded = true;
                                break;
                            }
                            if (!IS_INTRA_T(type)) update_cache_p();
                            const int c = rd_cost_mb();
                            if (step == 0) { rd16 = c; if (type == T_P_SKIP) rd_skip = true; }
                            else if (step == 2) rd16 = c;
                            else if (step == 3) c16x8 = c;
                            else if (step == 4) c8x16 = c;
                            else if (step == 5) c8x8 = c;
                            else if (step == 6) {
                                if (i_cost >= c) {
                                    if (i_cost > 0) satd_inter = (int)((long long)satd_inter * c / i_cost);
                                    if (satd_inter == 0) satd_inter = 1;
                                    i_cost = c;
                                } else
                                    t8 = !t8;
                            } else if (step == 7) satd_i16 = c;
                            else if (step == 8) satd_i4 = c;
                            else satd_i9_PLACEHOLDER;
                        }'''
pflow=pflow.replace("else satd_i9_PLACEHOLDER;","else satd_i8 = c;")
rep("                        RD_P_FLOW\n", "                        "+pflow+"\n")
open(p,'w').write(s)
print("ok")
