import sys
p='/root/repo/x264_vs2008_amd/csrc/slice_kernel.h'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt): print("MISMATCH",n,a[:110]); sys.exit(1)
    s=s.replace(a,b)
rep('''enum { T_I_4x4 = 0, T_I_8x8 = 1, T_I_16x16 = 2, T_I_PCM = 3, T_P_L0 = 4, T_P_8x8 = 5, T_P_SKIP = 6 };''','''enum { T_I_4x4 = 0, T_I_8x8 = 1, T_I_16x16 = 2, T_I_PCM = 3, T_P_L0 = 4, T_P_8x8 = 5, T_P_SKIP = 6,
       T_B_DIRECT = 7, T_B_L0_L0 = 8, T_B_L1_L1 = 12, T_B_BI_BI = 16, T_B_8x8 = 17, T_B_SKIP = 18 };
#define IS_SKIP_T(t) ((t) == T_P_SKIP || (t) == T_B_SKIP)''')
rep('''    int ref_bits[SW_MAX_REFS], poc_delta[SW_MAX_REFS], l0_inv_ref_poc[SW_MAX_REFS];   // REF_COST = lambda * ref_bits (bs_size_te, R/encoder/analyse.c:195-197)
};''','''    int ref_bits[SW_MAX_REFS], poc_delta[SW_MAX_REFS], l0_inv_ref_poc[SW_MAX_REFS];   // REF_COST = lambda * ref_bits (bs_size_te, R/encoder/analyse.c:195-197)
    // B slices: the list-1 picture (x264 core 66 without b-pyramid has one) and h->mb.bipred_weight[list-0 reference][0]
    const u8 *y1[4], *u1, *v1;
    int biw[SW_MAX_REFS];
};''')
rep('''    i16 *mvd;                       // h->mb.mvd[0]: [batch][n_mb][16][2]
};''','''    i16 *mvd;                       // h->mb.mvd[0]: [batch][n_mb][16][2]
    // B slices: list 1 of the per-macroblock state, h->mb.skipbp, and the co-located picture's arrays (direct prediction)
    i16 *mv1, *mvr1, *mvd1;
    signed char *ref1;
    u8 *skipbp;
    const signed char *col_type, *col_ref;
    const i16 *col_mv;
};
// what a B slice adds to the wavefront's LDS: list 1 of the motion caches, the direct prediction, the analysis records
struct SwLdsB {
    signed char cref1[48], cskip[48];
    i16 cmv1[48][2], cmvd1[48][2];
    signed char dref[2][4], sub[4];     // h->mb.cache.direct_ref; h->mb.i_sub_partition
    i16 dmv[2][16][2];                  // h->mb.cache.direct_mv (the 16 blocks in raster order)
    i16 mv4_1[16][2];
    signed char ref8_1[4];
    i16 left_mv4_1[4][2], left_mvd1[4][2], left_mvr1[2];
    signed char left_r8_1[2];
    u8 left_skipbp;
    int me[2][9][6];                    // x264_me_t records of a->l0 / a->l1: [list][me16x16, me8x8 x 4, me16x8 x 2, me8x16 x 2][mv x, y, cost, cost_mv, mvp x, y]
    int cost8direct[4];
    u8 visited[512];                    // x264_me_refine_bidir's visited[8][8][8]
};
struct SwLdsBNone { int unused; };
// x264_me_refine_bidir's 32 candidate offsets per pass in evaluation order (CHECK_BIDIR8 / CHECK_BIDIR2, R/encoder/me.c:893-909):
// (m0x, m0y, m1x, m1y) offsets + 1 in four 2-bit fields
static __device__ const u8 c_bidir_dirs[32] = {BIDIR_DIRS};''')
rep('''template <int WPE, bool LL = false, bool RD = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_slice_sweep(SwArgs a, SwRefs refs, SwRd rd)
{''','''template <int WPE, bool LL = false, bool RD = false, bool BS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_slice_sweep(SwArgs a, SwRefs refs, SwRd rd)
{
    static_assert(!BS || RD, "B slices run in the raster variant");''')
rep('''    SwLdsRd &sr = *(SwLdsRd *)&sr_;                     // only touched when RD''','''    SwLdsRd &sr = *(SwLdsRd *)&sr_;                     // only touched when RD
    __shared__ typename std::conditional<BS, SwLdsB, SwLdsBNone>::type sb_;
    SwLdsB &sb = *(SwLdsB *)&sb_;                       // only touched when BS''')
rep('''        if (is_p)
            for (int k = lane; k < 2 * MX_COST_LDS + 1; k += 64) s.costl[k] = cost_g[k - MX_COST_LDS];''','''        if (is_p || BS)
            for (int k = lane; k < 2 * MX_COST_LDS + 1; k += 64) s.costl[k] = cost_g[k - MX_COST_LDS];''')
rep('''            if (!is_p || mb <= 4) return 0;
            if (IS_INTRA_T(left_type) || IS_INTRA_T(type_top) || IS_INTRA_T(type_topleft) || IS_INTRA_T(type_topright)) return 0;
            if (a.l0_type && IS_INTRA_T(UNI(a.l0_type[mb]))) return 0;''','''            if ((!is_p && !BS) || mb <= 4) return 0;
            if (IS_INTRA_T(left_type) || IS_INTRA_T(type_top) || IS_INTRA_T(type_topleft) || IS_INTRA_T(type_topright)) return 0;
            if (is_p && a.l0_type && IS_INTRA_T(UNI(a.l0_type[mb]))) return 0;      // only in a P slice (analyse.c:357)''')
rep('''                    if (c < satd_i16) { satd_i16 = c; pred16 = m; }
                }
            }
            if (!(a.flags_intra & 3)) return;''','''                    if (c < satd_i16) { satd_i16 = c; pred16 = m; }
                }
            }
            if constexpr (BS) satd_i16 += Q.lambda * 9;                  // i_mb_b_cost_table[I_16x16], analyse.c:659-661
            if (!(a.flags_intra & 3)) return;''')
rep('''                int cost = 0, idx, acbp = 0;
                for (idx = 0;; idx++) {
                    const int bx = 8 * (idx & 1), by = 8 * (idx >> 1), pm = sw_pred_i4mode(s, 4 * idx), nb8 = sw_nb8(idx, nb);''','''                int cost = BS ? Q.lambda * 9 : 0, idx, acbp = 0;            // i_mb_b_cost_table[I_8x8], :676-677
                for (idx = 0;; idx++) {
                    const int bx = 8 * (idx & 1), by = 8 * (idx >> 1), pm = sw_pred_i4mode(s, 4 * idx), nb8 = sw_nb8(idx, nb);''')
rep('''                int cost = Q.lambda * 24, idx, acbp = 0;''','''                int cost = Q.lambda * (BS ? 24 + 9 : 24), idx, acbp = 0;    // + i_mb_b_cost_table[I_4x4] in a B slice, :770-771''')
# encode_mb
rep('''        auto encode_mb = [&](int final_pass) {
            if (type == T_P_SKIP) { encode_pskip(); return; }
            cbp_luma = 0; cbp_chroma = 0;
            if (lane < 32) s.nnz[lane] = 0;
            WAVE_SYNC();''','''        auto encode_mb = [&](int final_pass) {
            if (type == T_P_SKIP) { encode_pskip(); return; }
            cbp_luma = 0; cbp_chroma = 0;
            if (lane < 32) s.nnz[lane] = 0;
            WAVE_SYNC();
            if (BS && type == T_B_SKIP) return;              // x264_macroblock_encode_skip: the prediction (made by the caller) is the reconstruction''')
rep('''            } else {
                sw_mc_parts(s, refs, a, oy, oc, by_, bc_, lane);
                WAVE_SYNC();''','''            } else {
                if constexpr (!BS) sw_mc_parts(s, refs, a, oy, oc, by_, bc_, lane);    // (B slice: the caller has run the bi-predictive motion compensation)
                WAVE_SYNC();''')
rep('''                if (type == T_P_L0 && part == 16 && !(cbp_luma | cbp_chroma) && mvx == pskx && mvy == psky && ref == 0) type = T_P_SKIP;''','''                if (type == T_P_L0 && part == 16 && !(cbp_luma | cbp_chroma) && mvx == pskx && mvy == psky && ref == 0) type = T_P_SKIP;
                if (BS && type == T_B_DIRECT && !(cbp_luma | cbp_chroma)) type = T_B_SKIP;       // macroblock.c:784-788''')
# make_syn
rep('''            y.t8_allowed = a.transform8x8 && (type == T_P_L0 || (type == T_P_8x8 && __ballot(lane < 4 && sub_t_mb != 3) == 0));''','''            y.t8_allowed = a.transform8x8 && (type == T_P_L0 || (type == T_P_8x8 && __ballot(lane < 4 && sub_t_mb != 3) == 0));
            if constexpr (BS) y.t8_allowed = a.transform8x8 && type >= T_B_DIRECT && type <= T_B_8x8;''')
rep('''            y.sub = sr.sub; y.i4c = s.i4c; y.cref = sr.cref; y.cmv = sr.cmv; y.cmvd = sr.cmvd;''','''            y.sub = sr.sub; y.i4c = s.i4c; y.cref = sr.cref; y.cmv = sr.cmv; y.cmvd = sr.cmvd;
            y.n_ref1 = 0; y.cref1 = nullptr; y.cskip = nullptr; y.cmv1 = nullptr; y.cmvd1 = nullptr;
            if constexpr (BS) { y.n_ref1 = 1; y.sub = sb.sub; y.cref1 = sb.cref1; y.cskip = sb.cskip; y.cmv1 = sb.cmv1; y.cmvd1 = sb.cmvd1; }''')
# the flow
rep('''        if (!RD && !is_p) {
          {
            analyse_intra(MX_COST_MAX);''','''        if constexpr (BS) {
#include "slice_b_flow.h"
        } else
        if (!RD && !is_p) {
          {
            analyse_intra(MX_COST_MAX);''')
# tail: writer
rep('''                    if (type == T_P_SKIP) cw_mb_skip(cab, sr.cabac, left_type, type_top, 1);
                    else {
                        if (is_p) cw_mb_skip(cab, sr.cabac, left_type, type_top, 0);''','''                    if (IS_SKIP_T(type)) cw_mb_skip(cab, sr.cabac, left_type, type_top, 1, a.slice_type);
                    else {
                        if (is_p || BS) cw_mb_skip(cab, sr.cabac, left_type, type_top, 0, a.slice_type);''')
rep('''                if (type != T_P_SKIP) mb_qp = UNI(sr.tmp_i[1]);''','''                if (!IS_SKIP_T(type)) mb_qp = UNI(sr.tmp_i[1]);''')
rep('''            const signed char rv = (signed char)(is_p ? (intra ? -1 : s.ref8[lane]) : -1);''','''            const signed char rv = (signed char)(is_p || BS ? (intra ? -1 : s.ref8[lane]) : -1);''')
rep('''        if (lane < 27) (a.nnz + 27 * cb)[(size_t)mb * 27 + lane] = type == T_P_SKIP ? (u8)0 : s.nnz[lane];
        if (lane < 4) (a.sub_partition + 4 * cb)[(size_t)mb * 4 + lane] = (signed char)(type == T_P_8x8 ? sub_t_mb : 0);''','''        if (lane < 27) (a.nnz + 27 * cb)[(size_t)mb * 27 + lane] = IS_SKIP_T(type) ? (u8)0 : s.nnz[lane];
        if (lane < 4) (a.sub_partition + 4 * cb)[(size_t)mb * 4 + lane] = (signed char)(type == T_P_8x8 ? sub_t_mb : BS && type == T_B_8x8 ? (int)sb.sub[lane] : 0);
        if constexpr (BS) {     // list 1 of x264_macroblock_cache_save, h->mb.skipbp, and what the next macroblock sees to its left
            if (lane < 16) {
                const i16 vx = (i16)(intra ? 0 : sb.mv4_1[lane][0]), vy = (i16)(intra ? 0 : sb.mv4_1[lane][1]);
                (rd.mv1 + 32 * cb)[((size_t)mb * 16 + lane) * 2] = vx; (rd.mv1 + 32 * cb)[((size_t)mb * 16 + lane) * 2 + 1] = vy;
                if ((lane & 3) == 3) { sb.left_mv4_1[lane >> 2][0] = vx; sb.left_mv4_1[lane >> 2][1] = vy; }
            }
            if (lane < 4) {
                const signed char rv1 = (signed char)(intra ? -1 : sb.ref8_1[lane]);
                (rd.ref1 + 4 * cb)[(size_t)mb * 4 + lane] = rv1;
                if (lane & 1) sb.left_r8_1[lane >> 1] = rv1;
            }
            if (lane == 0) {
                const int sbp = type == T_B_SKIP || type == T_B_DIRECT ? 0xf
                              : type == T_B_8x8 ? (sb.sub[0] == 12) | (sb.sub[1] == 12) << 1 | (sb.sub[2] == 12) << 2 | (sb.sub[3] == 12) << 3 : 0;
                (rd.skipbp + cb)[mb] = (u8)sbp; sb.left_skipbp = (u8)sbp;
            }
        }''')
rep('''            (a.partition + cb)[mb] = (signed char)(intra || type == T_P_SKIP ? 16 : part);''','''            (a.partition + cb)[mb] = (signed char)(intra || IS_SKIP_T(type) || type == T_B_DIRECT ? 16 : part);''')
rep('''            (a.cbp + cb)[mb] = (i16)(type == T_P_SKIP ? 0 : type == T_I_PCM ? 0x72f : (cbp_dc << 8) | (cbp_chroma << 4) | cbp_luma);''','''            (a.cbp + cb)[mb] = (i16)(IS_SKIP_T(type) ? 0 : type == T_I_PCM ? 0x72f : (cbp_dc << 8) | (cbp_chroma << 4) | cbp_luma);''')
rep('''            const bool coded = type != T_P_SKIP && type != T_I_PCM;''','''            const bool coded = !IS_SKIP_T(type) && type != T_I_PCM;''')
rep('''            cbp_store = type == T_P_SKIP ? 0 : type == T_I_PCM ? 0x72f : (UNI(cbp_dc) << 8) | (cbp_chroma << 4) | cbp_luma;
            const bool keep = !intra && type != T_P_SKIP;
            if (lane < 16) {
                const int k = 12 + (lane & 3) + 8 * (lane >> 2);
                i16 *mvd = rd.mvd + ((cb + mb) * 16 + lane) * 2;
                mvd[0] = keep ? sr.cmvd[k][0] : (i16)0; mvd[1] = keep ? sr.cmvd[k][1] : (i16)0;
                if ((lane & 3) == 3) { sr.left_mvd[lane >> 2][0] = mvd[0]; sr.left_mvd[lane >> 2][1] = mvd[1]; }
            } else if (lane < 24) {
                const int j = lane - 16;
                const int idx = j < 4 ? (j == 0 ? 5 : j == 1 ? 7 : j == 2 ? 13 : 15) : 16 + 4 * ((j - 4) >> 1) + 1 + 2 * (j & 1);
                sr.left_nz[j] = type == T_P_SKIP ? (u8)0 : s.nnz[idx];
            }''','''            cbp_store = IS_SKIP_T(type) ? 0 : type == T_I_PCM ? 0x72f : (UNI(cbp_dc) << 8) | (cbp_chroma << 4) | cbp_luma;
            const bool keep = !intra && !IS_SKIP_T(type) && type != T_B_DIRECT;
            if (lane < 16) {
                const int k = 12 + (lane & 3) + 8 * (lane >> 2);
                i16 *mvd = rd.mvd + ((cb + mb) * 16 + lane) * 2;
                mvd[0] = keep ? sr.cmvd[k][0] : (i16)0; mvd[1] = keep ? sr.cmvd[k][1] : (i16)0;
                if ((lane & 3) == 3) { sr.left_mvd[lane >> 2][0] = mvd[0]; sr.left_mvd[lane >> 2][1] = mvd[1]; }
                if constexpr (BS) {
                    i16 *mvd1 = rd.mvd1 + ((cb + mb) * 16 + lane) * 2;
                    mvd1[0] = keep ? sb.cmvd1[k][0] : (i16)0; mvd1[1] = keep ? sb.cmvd1[k][1] : (i16)0;
                    if ((lane & 3) == 3) { sb.left_mvd1[lane >> 2][0] = mvd1[0]; sb.left_mvd1[lane >> 2][1] = mvd1[1]; }
                }
            } else if (lane < 24) {
                const int j = lane - 16;
                const int idx = j < 4 ? (j == 0 ? 5 : j == 1 ? 7 : j == 2 ? 13 : 15) : 16 + 4 * ((j - 4) >> 1) + 1 + 2 * (j & 1);
                sr.left_nz[j] = IS_SKIP_T(type) ? (u8)0 : s.nnz[idx];
            }''')
rep('''        left_ref = is_p ? (intra ? -1 : UNI(s.ref8[1])) : -1;''','''        left_ref = is_p || BS ? (intra ? -1 : UNI(s.ref8[1])) : -1;''')
# i16 decimate in B
rep('''        const int b_decimate = a.dct_decimate && a.slice_type == 0;''','''        const int b_decimate = a.slice_type == 1 || (a.dct_decimate && a.slice_type == 0);     // macroblock.c:193''')
# bidir dirs
dirs=[(0,0,0,1),(0,0,0,-1),(0,0,1,0),(0,0,-1,0),(0,1,0,0),(0,-1,0,0),(1,0,0,0),(-1,0,0,0),
 (0,0,1,1),(0,0,-1,-1),(0,1,1,0),(0,-1,-1,0),(1,1,0,0),(-1,-1,0,0),(1,0,0,1),(-1,0,0,-1),
 (0,1,0,1),(0,-1,0,-1),(1,0,1,0),(-1,0,-1,0),
 (0,0,-1,1),(0,0,1,-1),(0,-1,1,0),(0,1,-1,0),(-1,1,0,0),(1,-1,0,0),(1,0,0,-1),(-1,0,0,1),
 (0,-1,0,1),(0,1,0,-1),(-1,0,1,0),(1,0,-1,0)]
assert len(dirs)==32
vals=[ (a+1)|((b+1)<<2)|((c+1)<<4)|((d+1)<<6) for a,b,c,d in dirs]
rep("{BIDIR_DIRS}","{"+", ".join(str(v) for v in vals)+"}")
open(p,'w').write(s)
print("ok")
