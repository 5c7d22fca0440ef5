import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep('''        int stat_intra = 0, stat_inter = 0, analysed = 0;
''','''        int stat_intra = 0, stat_inter = 0, analysed = 0;
        // x264_mb_analyse_init (R/encoder/analyse.c:235-252): h->mb.b_trellis while analysing, i_skip_intra
        const int mbrd = RD ? rd.mbrd : 0;
        SwTq tq = {RD && rd.trellis > 1 && mbrd, &sr};
        int skip_intra = a.lossless ? 0 : mbrd ? 2 : (RD ? (!rd.trellis && !a.nr) : 1);
        (void)skip_intra;
''')
rep('''            if (IS_INTRA_T(left_type) || IS_INTRA_T(type_top) || IS_INTRA_T(type_topleft) || IS_INTRA_T(type_topright)) return 0;
            if (a.l0_type && IS_INTRA_T(UNI(a.l0_type[mb]))) return 0;''','''            if (IS_INTRA_T(left_type) || IS_INTRA_T(type_top) || IS_INTRA_T(type_topleft) || IS_INTRA_T(type_topright)) return 0;
            if (a.l0_type && IS_INTRA_T(UNI(a.l0_type[mb]))) return 0;
            if constexpr (RD) return mb < 3 * intra_before ? 0 : 1;        // raster order: every earlier macroblock is done''')
rep('''                const int thresh = min(satd_inter, satd_i16);
                int cost = 0, idx, acbp = 0;''','''                const int thresh = mbrd ? MX_COST_MAX : min(satd_inter, satd_i16);
                int cost = 0, idx, acbp = 0;''')
rep("                    sw_encode_i8x8(s, a, idx, acbp, lane);\n                }\n                if (idx == 3) {\n                    satd_i8 = cost; i8_cbp = acbp;",
    "                    sw_encode_i8x8(s, a, Q, tq, idx, acbp, lane);\n                }\n                if (idx == 3) {\n                    satd_i8 = cost; i8_cbp = acbp;\n                    if constexpr (RD) { if (skip_intra == 2) for (int k = lane; k < 256; k += 64) sr.i8_dct[k] = s.lv_y8[k]; }")
rep("                if (min(cost, satd_i16) > satd_inter * 5 / 4) return;","                if (min(cost, satd_i16) > satd_inter * (5 + !!mbrd) / 4) return;")
rep('''                const int thresh = min(min(satd_inter, satd_i16), satd_i8);
                int cost = Q.lambda * 24, idx, acbp = 0;''','''                int thresh = min(min(satd_inter, satd_i16), satd_i8);
                if (mbrd) thresh = thresh * (10 - fast_intra_now(0)) / 8;
                int cost = Q.lambda * 24, idx, acbp = 0;''')
rep("                    sw_encode_i4x4(s, a, idx, acbp, lane);\n                }\n                if (idx == 15) {\n                    satd_i4 = cost; i4_cbp = acbp;",
    "                    sw_encode_i4x4(s, a, Q, tq, idx, acbp, lane);\n                }\n                if (idx == 15) {\n                    satd_i4 = cost; i4_cbp = acbp;\n                    if constexpr (RD) { if (skip_intra == 2) for (int k = lane; k < 256; k += 64) sr.i4_dct[k] = s.lv_y[k]; }")
rep("b_skip = sw_probe_pskip(s, refs, a, pskx, psky,","b_skip = sw_probe_pskip(s, refs, a, Q, pskx, psky,")
rep("if (sw_probe_pskip(s, refs, a, pskx, psky,","if (sw_probe_pskip(s, refs, a, Q, pskx, psky,")
open(p,'w').write(s)
print("ok")
