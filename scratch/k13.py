import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
# 1. partitions -> lambda
rep('''                    int i_cost = best;
                    int c8x8 = MX_COST_MAX, c16x8 = MX_COST_MAX, c8x16 = MX_COST_MAX;   // a->l0.i_cost8x8 / i_cost16x8 / i_cost8x16
                    part = 16;                                       // D_16x16
                    if (a.flags_inter & 0x10) {''','''                    int i_cost = best;
                    int c8x8 = MX_COST_MAX, c16x8 = MX_COST_MAX, c8x16 = MX_COST_MAX;   // a->l0.i_cost8x8 / i_cost16x8 / i_cost8x16
                    auto search_partitions = [&]() {
                    part = 16;                                       // D_16x16
                    if (a.flags_inter & 0x10) {''')
rep('''                        c8x8 = cost8x8;
                    }
                    // x264_me_refine_qpel on the winning partition (analyse.c:2289-2352); the reference cost leaves every block's sum (me.c:639-640)
                    if (part == 16) {''','''                        c8x8 = cost8x8;
                    }
                    };
                    // x264_me_refine_qpel on the winning partition (analyse.c:2289-2352); the reference cost leaves every block's sum (me.c:639-640)
                    auto refine_winner = [&]() {
                    if (part == 16) {''')
rep('''                    WAVE_SYNC();
                    if (part == 13) sub_t_mb = sub_t;
                    PROF(2);
                    LAUNDER();
                    if (a.chroma_me) {''','''                    };
                    if constexpr (!RD) {
                    search_partitions();
                    refine_winner();
                    WAVE_SYNC();
                    if (part == 13) sub_t_mb = sub_t;
                    PROF(2);
                    LAUNDER();
                    if (a.chroma_me) {''')
rep('''                    stat_intra = icost; analysed = 1;
                    stat_inter = i_cost;
                }
            }
        }
        (void)analysed;''','''                    stat_intra = icost; analysed = 1;
                    stat_inter = i_cost;
                    } else {
                        RD_P_FLOW
                    }
                }
            }
        }
        (void)analysed;''')
open(p,'w').write(s)
print("ok")
