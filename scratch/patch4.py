import re,sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
a=s.index("static void analyse_mb(ssl *S, smb *m)")
b=s.index("/* x264_analyse_update_cache + x264_mb_analyse_transform (non-RD)")
f=s[a:b]
def rep(x,y,cnt=1):
    global f
    n=f.count(x)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",x[:90]); sys.exit(1)
    f=f.replace(x,y)
# remove local decls
rep("""            struct { int mvx, mvy, cost, cost_mv, ref, ref_cost; i16 mvp[2]; } me8[4], me16x8[2], me8x16[2];
            int cost8x8 = S_COST_MAX, cost16x8 = S_COST_MAX, cost8x16 = S_COST_MAX, part = S_D_16x16;
            sub_me me4[4][4], me84[4][2], me48[4][2];
            int sub[4] = {S_D_L0_8x8, S_D_L0_8x8, S_D_L0_8x8, S_D_L0_8x8};
""","""            int cost8x8 = S_COST_MAX, cost16x8 = S_COST_MAX, cost8x16 = S_COST_MAX, part = S_D_16x16;
            for (int i = 0; i < 4; i++) A->sub[i] = S_D_L0_8x8;
            A->me16.mvx = bmx; A->me16.mvy = bmy; A->me16.cost = best; A->me16.ref = bref; A->me16.ref_cost = S->ref_cost[bref];
            A->me16.mvp[0] = bmvp[0]; A->me16.mvp[1] = bmvp[1];
            A->rd16 = S_COST_MAX;
            if (S->mbrd) {                                       /* :1134-1143 */
                cache_fenc_satd(S, m);
                if (bref == 0 && bmx == m->pskip_mv[0] && bmy == m->pskip_mv[1]) {
                    m->partition = S_D_16x16;
                    update_cache(S, m, A);
                    A->rd16 = rd_cost_mb(S, m, S->lambda2);
                    if (m->type == S_P_SKIP) return;              /* :2230: the trial encode found nothing to code on the skip vector */
                }
            }
""")
# rename candidate arrays
f=re.sub(r'(?<![\w>.])me8\[', 'A->me8[', f)
f=re.sub(r'(?<![\w>.])me16x8\[', 'A->me16x8[', f)
f=re.sub(r'(?<![\w>.])me8x16\[', 'A->me8x16[', f)
f=re.sub(r'(?<![\w>.])me4\[', 'A->me4[', f)
f=re.sub(r'(?<![\w>.])me84\[', 'A->me84[', f)
f=re.sub(r'(?<![\w>.])me48\[', 'A->me48[', f)
f=re.sub(r'(?<![\w>.])sub\[', 'A->sub[', f)
f=f.replace("typeof(A->me16x8[0]) *d","pme *d").replace("typeof(A->me8[0]) *d","pme *d")
f=f.replace("static void analyse_mb(ssl *S, smb *m)\n{","static void analyse_mb(ssl *S, smb *m, panalysis *A)\n{")
s=s[:a]+f+s[b:]
open(p,'w').write(s)
print('ok')
