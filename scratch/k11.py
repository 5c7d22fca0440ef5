import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
# motion cache always loaded in the raster variant
rep('''            int sub_mx = 0, sub_my = 0, sub_cost = 0, sub_px = 0, sub_py = 0, sub_t = 3;      // sub-8x8 records (lanes 0..31) and chosen type (lanes 0..3)
            if (a.flags_inter & 0x10) {''','''            int sub_mx = 0, sub_my = 0, sub_cost = 0, sub_px = 0, sub_py = 0, sub_t = 3;      // sub-8x8 records (lanes 0..31) and chosen type (lanes 0..3)
            if (RD || (a.flags_inter & 0x10)) {''')
rep('''            // x264_mb_predict_mv_16x16, :90-128
            auto predict16 = [&](int i_ref, int &px, int &py) {''','''            if constexpr (RD) {     // the neighbours' part of the motion cache, for the entropy coder's x264_mb_predict_mv / ref contexts
                if (lane < 48) { sr.cref[lane] = (signed char)cref_v; sr.cmv[lane][0] = (i16)cmvx_v; sr.cmv[lane][1] = (i16)cmvy_v; }
                WAVE_SYNC();
            }
            // x264_mb_predict_mv_16x16, :90-128
            auto predict16 = [&](int i_ref, int &px, int &py) {''')
rep('''                    int i_cost = best;
                    part = 16;                                       // D_16x16
                    if (a.flags_inter & 0x10) {''','''                    int i_cost = best;
                    int c8x8 = MX_COST_MAX, c16x8 = MX_COST_MAX, c8x16 = MX_COST_MAX;   // a->l0.i_cost8x8 / i_cost16x8 / i_cost8x16
                    part = 16;                                       // D_16x16
                    if (a.flags_inter & 0x10) {''')
rep('''                                if (sum < i_cost) { i_cost = sum; type = T_P_L0; part = dir ? 15 : 14; }
                            }
                    }''','''                                if (dir) c8x16 = sum; else c16x8 = sum;
                                if (sum < i_cost) { i_cost = sum; type = T_P_L0; part = dir ? 15 : 14; }
                            }
                        c8x8 = cost8x8;
                    }''')
open(p,'w').write(s)
print("ok")
