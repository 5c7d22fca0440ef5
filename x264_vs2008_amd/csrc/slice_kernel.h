#pragma once
// frame_slice.hip -- FRAME LEVEL, part 7: the reference's per-macroblock hot loop itself.
//
// What x264_slice_write does for every macroblock of a slice (R/encoder/encoder.c:1141-1291):
//   x264_macroblock_cache_load   R/common/macroblock.c:872-1187   neighbour state, predictors
//   x264_macroblock_analyse      R/encoder/analyse.c:2156-2774    I and P slices, no RD (subme <= 5)
//   x264_macroblock_encode       R/encoder/macroblock.c:475-790
//   x264_macroblock_cache_save   R/common/macroblock.c:1208-1372
// in ONE launch per frame for a whole batch of independent GOP chains.
//
// Schedule.  A macroblock needs its left, top-left, top and top-right neighbours finished
// (reconstructed pixels for intra prediction, vectors / references / types for the predictors), so
// a frame is a 2:1 wavefront.  One wavefront (= one workgroup) owns one macroblock ROW of one
// chain and walks it left to right; row r may start macroblock x once row r-1 has published x+2
// finished macroblocks (release store / acquire load on a per-row progress word in HBM).  The left
// neighbour is the wave's own previous iteration: its state stays in LDS / registers.  Workgroup
// ids are laid out so that all rows of a chain land on the same XCD (id % 8 == chain % 8): the
// cross-row traffic stays inside one L2.  Every wait is bounded: a wave that spins too long raises
// the abort flag and every wave leaves, so the grid always drains.  FORWARD PROGRESS ASSUMPTION: the grid is oversubscribed
// (more row waves than wave slots), so a waiting wave relies on the waves of the rows above being resident or dispatched before it:
// workgroups are dispatched in increasing blockIdx order on this hardware (ids are laid out row-major for exactly that reason).
// HIP does not promise the order; if it ever changed, waves would exhaust their spin budget and the launch would be reported as
// aborted (never silently wrong): the abort count is sticky per context (x264hip_slice_sweep_status).
//
// Inside a macroblock all 64 lanes work on the same block and every decision is wave-uniform scalar state: 4 luma pixels
// per lane for prediction and SAD, one lane per 8x4 block for SATD, one lane per coefficient for the 4x4 transform, a lane
// per column / row for the 8x8 one; the motion search scores one candidate per lane group and ranks a trip's candidates
// with one packed key (me_exact.h); intra 4x4 / 8x8 modes are read out of a per-block table through a compile-time LUT
// (intra_pred.h); the motion cache and the partition candidates live in lane-indexed registers.
// Built: I_16x16 / I_8x8 / I_4x4 + chroma modes, P_SKIP (fast and early), P 16x16 / 16x8 / 8x16 / 8x8 over several
// references with or without mixed references (DIA / HEX / UMH, subme 0..5, chroma ME), 4x4 / 8x8 transform choice, CQP.
// The lane id is laundered once per macroblock and at phase boundaries (LAUNDER): otherwise every lane-derived address of the
// 60k-instruction body is hoisted out of the macroblock loop and spilled.
#include <cstring>
#include "me_exact.h"
#include "intra_pred.h"
#include "frame_internal.h"
#include "trellis_wave.h"

using namespace x264hip;

#define SW_MAX_REFS 8
#define SW_SPIN_LIMIT (1 << 21)
// Most bytes one macroblock's CABAC syntax can take, proven rather than observed: 384 coefficients x (significance + last flag + 14 prefix
// bins, each at most -log2(0.01875) < 6 bits when it is the least probable symbol of the most skewed state, + 31 bypass bins of the
// Exp-Golomb suffix of a 16-bit level + the sign) = 384 x 128 bits = 6144 bytes, + under 400 bytes of header, modes, vector differences
// of 16 blocks x 2 lists; I_PCM is 384.  The sweep stops (abort flag) before a macroblock that might not fit.
#define SW_MB_BYTES_MAX 8192
#define FD 32                      // FDEC_STRIDE
#define FDY (2 * FD)               // fdec_buf layout, R/common/macroblock.c:721-737
#define FDU (19 * FD)
#define FDV (19 * FD + 16)

enum { T_I_4x4 = 0, T_I_8x8 = 1, T_I_16x16 = 2, T_I_PCM = 3, T_P_L0 = 4, T_P_8x8 = 5, T_P_SKIP = 6,
       T_B_DIRECT = 7, T_B_L0_L0 = 8, T_B_L1_L1 = 12, T_B_BI_BI = 16, T_B_8x8 = 17, T_B_SKIP = 18 };
#define IS_SKIP_T(t) ((t) == T_P_SKIP || (t) == T_B_SKIP)
enum { NB_LEFT = 1, NB_TOP = 2, NB_TOPRIGHT = 4, NB_TOPLEFT = 8 };
#define IS_INTRA_T(t) ((t) >= 0 && (t) <= T_I_PCM)

struct SwRefs {
    const u8 *y[SW_MAX_REFS][4];
    const u8 *u[SW_MAX_REFS], *v[SW_MAX_REFS];
    // everything indexed by a run-time reference number lives here, in the argument the kernel never writes: SwArgs is adjusted per
    // chain at the top of the kernel, and a modified argument struct with a dynamically indexed member is kept in scratch memory whole
    int ref_bits[SW_MAX_REFS], poc_delta[SW_MAX_REFS], l0_inv_ref_poc[SW_MAX_REFS];   // REF_COST = lambda * ref_bits (bs_size_te, R/encoder/analyse.c:195-197)
    // B slices: the list-1 picture (x264 core 66 without b-pyramid has one) and h->mb.bipred_weight[list-0 reference][0]
    const u8 *y1[4], *u1, *v1;
    int biw[SW_MAX_REFS];
    int dsf[SW_MAX_REFS], map_col[SW_MAX_REFS];     // temporal direct: h->mb.dist_scale_factor[i][0], h->mb.map_col_to_list0[i]
};
struct SwArgs {
    int mb_w, mb_h, sy, sc, batch, batch_pad;
    int chain;                  // the chain-table launch (template argument CH): the batch element this table entry codes
    size_t bs_y, bs_c;
    int slice_type, qp, qpc, lambda, chroma_skip_thresh, n_refs;
    int l0_n_ref0;
    int me_method, me_range, subme, chroma_me, fast_pskip, dct_decimate, cabac, mv_range;
    int flags_inter, mixed_refs; // X264_ANALYSE_PSUB16x16 (0x10) / PSUB8x8 (0x20) of param.analyse.inter; param.analyse.b_mixed_references
    int flags_intra;            // X264_ANALYSE_I4x4 | I8x8 bits that apply to this slice type (param.analyse.intra / .inter)
    int transform8x8;
    const u16 *q4mf, *q4bias, *q8mf, *q8bias;
    const int *dq4, *dq8;
    const i16 *cost_mv;
    int cost_center;
    const i16 *lowres0, *lowres1;   // fenc->lowres_mvs of list 0 / 1 towards reference 0: [batch][n_mb][2], or NULL (x264hip_slice_params.lowres_mv)
    const u8 *fy, *fu, *fv;
    u8 *dy, *du, *dv;
    const signed char *l0_type, *l0_ref;
    const i16 *l0_mv;
    signed char *mb_type, *partition, *sub_partition, *ref, *i4mode, *i16mode, *chroma_mode, *qp_out, *t8;
    i16 *mv, *mvr, *cbp;
    u8 *nnz;
    i16 *luma, *luma_dc, *chroma_dc, *chroma_ac;
    int *cost_intra, *cost_inter, *cost_alt;
    int *progress, *abort_flag;
    int *abort_total;           // per context, never reset: waves that gave up waiting, over all launches (x264hip_slice_sweep_status)
    int spin_limit;             // polls before a waiting wave gives up (SW_SPIN_LIMIT; X264HIP_SPIN_LIMIT overrides it for the abort-path test)
    long long *prof;            // optional [batch][mb_h][8] accumulated wall-clock ticks per phase (developer aid)
    int nr;                     // param.analyse.i_noise_reduction != 0
    int lossless;               // h->mb.b_lossless
    u32 *nr_sum, *nr_count;     // [batch][2][64], [batch][2]
    const u16 *nr_offset;       // [batch][2][64]
};

// the macroblock's QP and what follows from it (x264_mb_analyse_init, R/encoder/analyse.c:227-230): one set per slice at constant
// QP, per macroblock with adaptive quantisation
struct SwQp { int qp, qpc, lambda, lambda2, skip_thresh; };

struct SwLds {
    __attribute__((aligned(16))) u8 fe[384];   // source: Y 16x16 | U 8x8 | V 8x8
    u8 fd[27 * FD];             // prediction / reconstruction with its borders, fdec_buf layout
    i16 coef[16][16];           // dequantised luma coefficients
    i16 ccoef[8][16];           // dequantised chroma AC
    int score[16], cscore[8];
    i16 cdc[8], cdcout[8], dc16[16];
    int keep8, cmode[2], nzdc16;
    i16 lv_y[256], lv_dc[16], lv_cdc[8], lv_cac[128];
    u8 nnz[32];
    i16 mvc[9][2];              // x264_mb_predict_mv_ref16x16's list: direct, lookahead, four neighbours, three temporal
    i16 left_mvr[SW_MAX_REFS][2];
    // intra 4x4 / 8x8 analysis: prediction-mode cache in x264_scan8 layout, edge arrays, and what the reference keeps
    // when i_skip_intra is set (the partly encoded macroblock of the analysis is the final one, macroblock.c:527-577)
    signed char i4c[48];
    u8 e4[16], edge8[40];
    __attribute__((aligned(4))) u8 pt4[48];   // the current 4x4 / 8x8 block's prediction table (intra_pred.h: RAW | F1 | F2 | DC..)
    __attribute__((aligned(4))) u8 pt8[80];
    u32 p4lut[48], p8lut[192];  // c_plut4 / c_plut8
    u16 nr_off4[16], nr_off8[64];   // h->nr_offset[0] / [1] of this chain (--nr)
    __attribute__((aligned(16))) u8 patch[MX_PATCH_BYTES];   // the motion search's staged sub-pel neighbourhood (me_exact.h)
    u8 i4_fdec[256], i8_fdec[256], i4_nnz[16], i8_nnz[16];
    i16 lv_y8[256];             // levels of the 8x8 transform (h->dct.luma8x8), separate from the 4x4 ones like the reference's
    i16 t8[256];                // 8x8 transform: intermediate between the two 1-D passes
    signed char left_i4[4];     // the left macroblock's modes of blocks 5, 7, 13, 15
    signed char pred4[16], pred8[4];
    // P partitions: the motion cache (h->mb.cache.ref / mv, x264_scan8 layout), a->l0.mvc, candidate records, final vectors
    i16 l0mvc[SW_MAX_REFS][5][2];
    i16 mv4[16][2];
    signed char ref8[4];
    i16 left_mv4[4][2];         // the left macroblock's vectors of blocks 3, 7, 11, 15 and references of its 8x8 blocks 1, 3
    signed char left_r8[2];
    u16 q8mf[2][64], q8bias[2][64];
    int q8dq[2][64];
    // this frame's quantiser rows (cat 0 intra Y, 1 inter Y at qp; 2 intra C, 3 inter C at the chroma qp) and the centre of p_cost_mv
    u16 qmf[4][16], qbias[4][16];
    int qdq[4][16];
    i16 costl[2 * MX_COST_LDS + 2];
};

// ---- round 2: what the raster-order variant of the sweep (RD levels, trellis, adaptive quantisation, the entropy coder) adds ----
struct SwRd {                       // kernel argument
    int on;                         // this launch is the raster variant
    int mbrd, trellis, psy_rd;      // a->i_mbrd, param.analyse.i_trellis, h->mb.i_psy_rd
    int write, cabac_init_idc, i_frame, i_frame_stride;
    int aq, qp_min, qp_max, chroma_qp_offset;
    float f_qpm;
    const float *aq_offset;         // [batch][n_mb]
    const i16 *cost_mv_all;         // [52][2 * cost_center + 1]: p_cost_mv of every QP
    const int *unq4, *unq8;         // h->unquant4_mf [4][52][16], h->unquant8_mf [2][52][64]
    u8 *payload; int payload_cap; int *payload_len, *mb_bits;
    i16 *mvd;                       // h->mb.mvd[0]: [batch][n_mb][16][2]
    // B slices: list 1 of the per-macroblock state, h->mb.skipbp, and the co-located picture's arrays (direct prediction)
    i16 *mv1, *mvr1, *mvd1;
    signed char *ref1;
    u8 *skipbp;
    const signed char *col_type, *col_ref;
    const i16 *col_mv;
    int direct_temporal;            // !sh.b_direct_spatial_mv_pred
    int *direct_score;              // --direct auto: [batch][2] h->stat.frame.i_direct_score ({temporal, spatial}); NULL: off
    i16 *stale;                     // [batch][8]: the cache entry of block 12 that survives macroblocks and frames (x264hip_slice_rd.stale)
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
    i16 stale[6];                       // SwRd::stale while the slice is coded: {ref, mv x, mv y} of cache entry 30, list 0 | list 1
    int me[2][9][6];                    // x264_me_t records of a->l0 / a->l1: [list][me16x16, me8x8 x 4, me16x8 x 2, me8x16 x 2][mv x, y, cost, cost_mv, mvp x, y]
    int cost8direct[4];
};
struct SwLdsRdB;
// x264_me_refine_bidir's 32 candidate offsets per pass in evaluation order (CHECK_BIDIR8 / CHECK_BIDIR2, R/encoder/me.c:893-909):
// (m0x, m0y, m1x, m1y) offsets + 1 in four 2-bit fields
static __device__ const u8 c_bidir_dirs[32] = {149, 21, 101, 69, 89, 81, 86, 84, 165, 5, 105, 65, 90, 80, 150, 20, 153, 17, 102, 68, 133, 37, 97, 73, 88, 82, 22, 148, 145, 25, 100, 70};
struct SwLdsRd {
    u8 cabac[460], cabac_tmp[460];  // h->cabac.state and the RD trial's copy (COPY_CABAC, R/encoder/rdo.c:62)
    // what the entropy coder reads beyond SwLds (MbSynDev below points into both)
    signed char cref[48], sub[4];
    i16 cmv[48][2], cmvd[48][2];
    u8 nz_l[4], nz_t[4], nz_lc[2][2], nz_tc[2][2];
    i16 i4_dct[256], i8_dct[256];   // h->mb.pic.i4x4_dct_buf / i8x8_dct_buf (i_skip_intra == 2)
    int fenc_satd[16], fenc_sa8d[4];   // h->mb.pic.fenc_satd / fenc_sa8d (psy-RD)
    int unq4[4][16], unq8[2][64];   // unquant rows of the current QPs
    i16 left_mvd[4][2];             // the left macroblock's mvd of blocks 3, 7, 11, 15
    u8 left_nz[8];                  // its non_zero_count of blocks 5 7 13 15 | U 1 3 | V 1 3
    u8 zz2[4], zz4[16], zz8[64];    // scan position -> raster index; the trellis weights in scan order (x264_dct4/8_weight2_zigzag[0])
    int w4z[16], w8z[64];
    u8 zero16[16];                  // sixteen zeros (SATD / SA8D of the source against nothing)
    int tmp_i[4];                   // lane 0 -> wave: bit count / QP after the writer
    TdWave tw;
};
struct SwLdsNone { int unused; };
struct SwLdsRdB { SwLdsRd r; SwLdsB b; };
// what the RD refinement (subme 8+, template argument RF) adds: the analysis' per-mode costs (a->i_satd_i16x16_dir[mode],
// i_satd_i8x8chroma_dir[list position], i_satd_i8x8_dir[mode][block]) and the pixels x264_intra_rd_refine keeps of its best trial
struct SwLdsRf { int i16dir[8], cdir[4], i8dir[12][4]; u8 pels[16]; };
struct SwLdsRdF { SwLdsRd r; SwLdsRf f; };
// the record cabac_dev.h's writer walks (same member names as MbSyn): scalars in registers, arrays where the kernel keeps them in LDS
struct MbSynDev {
    int slice_type, type, partition, i16mode, chroma_mode, cbp_luma, cbp_chroma, t8, qp, n_ref, pps_t8, t8_allowed;
    int type_left, type_top, cbp_left, cbp_top, cpm_left, cpm_top, nb_t8, last_qp, last_dqp, prev_coded;
    signed char *sub, *i4c, *cref;
    i16 (*cmv)[2], (*cmvd)[2];
    int n_ref1;                          // B slices: list 1 of the caches, the skip flags of direct blocks
    signed char *cref1, *cskip;
    i16 (*cmv1)[2], (*cmvd1)[2];
    u8 *nnz, *nz_l, *nz_t;
    u8 (*nz_lc)[2], (*nz_tc)[2];
    i16 (*lv4)[16], (*lv8)[64], *lv_dc, (*lv_cdc)[4], (*lv_cac)[16];
};
// trellis context handed to the quantising helpers: on = 0 -> plain dead-zone quantisation
struct SwTq { int on; SwLdsRd *r; };
// x264_dct4_weight2_zigzag[0] / x264_dct8_weight2_zigzag[0] (R/common/dct.c:476-483) and x264_zigzag_scan4[0]
static __device__ const int d_w4z[16] = {800, 320, 320, 800, 128, 800, 320, 320, 320, 320, 128, 800, 128, 320, 320, 128};
static __device__ const u8 d_zz4[16] = {0, 4, 1, 2, 5, 8, 12, 9, 6, 3, 7, 10, 13, 14, 11, 15};
static __device__ const u8 d_zz2[4] = {0, 1, 2, 3};
static __device__ const u16 d_w8k[6] = {256, 201, 656, 227, 410, 363};
static __device__ const u8 d_w8cls[16] = {0, 3, 4, 3, 3, 1, 5, 1, 4, 5, 2, 5, 3, 1, 5, 1};
__device__ __forceinline__ int sw_w8z(int pos) { const int r = c_scan8[0][pos]; return d_w8k[d_w8cls[((r >> 1) & 12) | (r & 3)]]; }
struct SwW8 { __device__ __forceinline__ int operator[](int pos) const { return sw_w8z(pos); } };

// lanes exchange data through LDS only: order LDS traffic (lgkmcnt) and leave global loads / stores in flight
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

__device__ __forceinline__ void sw_blk_xy(int k, int &x, int &y)
{
    x = ((k >> 2) & 1) * 8 + (k & 1) * 4;
    y = (k >> 3) * 8 + ((k >> 1) & 1) * 4;
}
__device__ __forceinline__ int sw_decimate(const i16 *lv, int n)
{   // R/common/quant.c:213-239 on scanned levels lv[0..n)
    int i = n - 1, score = 0;
    while (i >= 0 && lv[i] == 0) i--;
    while (i >= 0) {
        if ((unsigned)(lv[i--] + 1) > 2u) return 9;
        int run = 0;
        while (i >= 0 && lv[i] == 0) { i--; run++; }
        score += c_decimate4[run];
    }
    return score;
}
__device__ __forceinline__ int sw_ue_size(int v) { return v == 0 ? 1 : v < 3 ? 3 : 5; }   // bs_size_ue for 0..6
__device__ __forceinline__ int sw_median(int a, int b, int c) { int mx = max(a, b), mn = min(a, b); return c > mx ? mx : c < mn ? mn : c; }

// ---- motion compensation of one 16x16 vector into s.fd (x264_mb_mc_0xywh, R/common/macroblock.c:462-476)
__device__ __forceinline__ void sw_mc16(SwLds &s, const SwRefs &refs, const SwArgs &a, int ri, int mvx, int mvy, ptrdiff_t oy, ptrdiff_t oc,
                                        size_t by, size_t bc, int lane, bool do_chroma)
{
    {
        const int r = lane >> 2, x = (lane & 3) * 4;
        const int qx = mvx & 3, qy = mvy & 3, idx = qy * 4 + qx;
        const ptrdiff_t base = oy + (ptrdiff_t)((mvy >> 2) + r) * a.sy + (mvx >> 2) + x + (ptrdiff_t)by;
        const u8 *pa = refs.y[ri][c_qpel_a[idx]] + base + (qy == 3) * a.sy;
        const u8 *pb = refs.y[ri][c_qpel_b[idx]] + base + (qx == 3);
#pragma unroll
        for (int i = 0; i < 4; i++)
            s.fd[FDY + r * FD + x + i] = (idx & 5) ? (u8)(((int)pa[i] + (int)pb[i] + 1) >> 1) : pa[i];
    }
    if (do_chroma) {
        const int cx = lane & 7, cy = lane >> 3;
        const int dx = mvx & 7, dyy = mvy & 7;
        const int ca = (8 - dx) * (8 - dyy), cb = dx * (8 - dyy), cc = (8 - dx) * dyy, cd = dx * dyy;
        const ptrdiff_t cbase = oc + (ptrdiff_t)((mvy >> 3) + cy) * a.sc + (mvx >> 3) + cx + (ptrdiff_t)bc;
        const u8 *pu = refs.u[ri] + cbase, *pv = refs.v[ri] + cbase;
        s.fd[FDU + cy * FD + cx] = (u8)((ca * pu[0] + cb * pu[1] + cc * pu[a.sc] + cd * pu[a.sc + 1] + 32) >> 6);
        s.fd[FDV + cy * FD + cx] = (u8)((ca * pv[0] + cb * pv[1] + cc * pv[a.sc] + cd * pv[a.sc + 1] + 32) >> 6);
    }
}

// x264_mb_mc for any P partition: every pixel with the vector of its 4x4 block and the reference of its 8x8 (s.mv4 / s.ref8)
// clip: x264_mb_mc_0xywh's clip of the vector to h->mb.mv_min / mv_max (R/common/macroblock.c:465-466); the analysis never leaves a vector
// outside them, the candidates of the RD refinement (subme 8+) may sit a quarter sample or two beyond
__device__ __forceinline__ void sw_mc_parts(SwLds &s, const SwRefs &refs, const SwArgs &a, ptrdiff_t oy, ptrdiff_t oc, size_t by, size_t bc, int lane,
                                            bool clip = false, int mbx = 0, int mby = 0)
{
    const int lox = 4 * (-16 * mbx - 24), hix = 4 * (16 * (a.mb_w - mbx - 1) + 24), loy = 4 * (-16 * mby - 24), hiy = 4 * (16 * (a.mb_h - mby - 1) + 24);
    {
        const int r = lane >> 2, x = (lane & 3) * 4, blk = (r >> 2) * 4 + (x >> 2);
        int mvx = s.mv4[blk][0], mvy = s.mv4[blk][1];
        const int ri = s.ref8[(r >> 3) * 2 + (x >> 3)];
        if (clip) { mvx = clip3(mvx, lox, hix); mvy = clip3(mvy, loy, hiy); }
        const int qx = mvx & 3, qy = mvy & 3, idx = qy * 4 + qx;
        const ptrdiff_t base = oy + (ptrdiff_t)((mvy >> 2) + r) * a.sy + (mvx >> 2) + x + (ptrdiff_t)by;
        const u8 *pa = refs.y[ri][c_qpel_a[idx]] + base + (qy == 3) * a.sy;
        const u8 *pb = refs.y[ri][c_qpel_b[idx]] + base + (qx == 3);
#pragma unroll
        for (int i = 0; i < 4; i++)
            s.fd[FDY + r * FD + x + i] = (idx & 5) ? (u8)(((int)pa[i] + (int)pb[i] + 1) >> 1) : pa[i];
    }
    {
        const int cx = lane & 7, cy = lane >> 3, blk = (cy >> 1) * 4 + (cx >> 1);
        int mvx = s.mv4[blk][0], mvy = s.mv4[blk][1];
        const int ri = s.ref8[(cy >> 2) * 2 + (cx >> 2)];
        if (clip) { mvx = clip3(mvx, lox, hix); mvy = clip3(mvy, loy, hiy); }
        const int dx = mvx & 7, dyy = mvy & 7;
        const int ca = (8 - dx) * (8 - dyy), cb = dx * (8 - dyy), cc = (8 - dx) * dyy, cd = dx * dyy;
        const ptrdiff_t cbase = oc + (ptrdiff_t)((mvy >> 3) + cy) * a.sc + (mvx >> 3) + cx + (ptrdiff_t)bc;
        const u8 *pu = refs.u[ri] + cbase, *pv = refs.v[ri] + cbase;
        s.fd[FDU + cy * FD + cx] = (u8)((ca * pu[0] + cb * pu[1] + cc * pu[a.sc] + cd * pu[a.sc + 1] + 32) >> 6);
        s.fd[FDV + cy * FD + cx] = (u8)((ca * pv[0] + cb * pv[1] + cc * pv[a.sc] + cd * pv[a.sc + 1] + 32) >> 6);
    }
}

// ---- block costs between s.fe and s.fd ---------------------------------------------------------
// one row of an 8x4 block per lane (lanes of one block are l, l^1, l^2, l^3); returns the block SATD
__device__ __forceinline__ int sw_satd_row8(const u8 *f, const u8 *p, int lane)
{
    int d[8];
#pragma unroll
    for (int x = 0; x < 8; x++) d[x] = (int)f[x] - (int)p[x];
    u32 p0 = (u32)d[0] + ((u32)d[4] << 16), p1 = (u32)d[1] + ((u32)d[5] << 16);
    u32 p2 = (u32)d[2] + ((u32)d[6] << 16), p3 = (u32)d[3] + ((u32)d[7] << 16);
    u32 t0, t1, t2, t3;
    wht4(t0, t1, t2, t3, p0, p1, p2, p3);
    return satd_rows4(t0, t1, t2, t3, lane);
}
// mbcmp[PIXEL_16x16](fdec luma, fenc luma): SATD above subme 1, else SAD (R/encoder/encoder.c:608-618)
__device__ __forceinline__ int sw_cmp_luma16(const SwLds &s, int satd, int lane)
{
    int v = 0;
    if (satd) {
        if (lane < 32) {
            const int blk = lane >> 2, r = lane & 3, bx = (blk & 1) * 8, y = (blk >> 1) * 4 + r;
            v = sw_satd_row8(s.fe + y * 16 + bx, s.fd + FDY + y * FD + bx, lane);
            if (r) v = 0;
        }
    } else {
        const int r = lane >> 2, x = (lane & 3) * 4;
        v = (int)sad4(*(const u32 *)(s.fe + r * 16 + x), *(const u32 *)(s.fd + FDY + r * FD + x), 0);
    }
    return wave_sum(v);
}
// mbcmp[PIXEL_8x8] of both chroma planes, summed
__device__ __forceinline__ int sw_cmp_chroma(const SwLds &s, int satd, int lane)
{
    int v = 0;
    if (satd) {
        if (lane < 16) {
            const int pl = lane >> 3, blk = (lane >> 2) & 1, r = lane & 3, y = blk * 4 + r;
            v = sw_satd_row8(s.fe + 256 + 64 * pl + y * 8, s.fd + (pl ? FDV : FDU) + y * FD, lane);
            if (r) v = 0;
        }
    } else {
        const int x = lane & 7, y = lane >> 3;
        v = iabs((int)s.fe[256 + y * 8 + x] - (int)s.fd[FDU + y * FD + x]) + iabs((int)s.fe[320 + y * 8 + x] - (int)s.fd[FDV + y * FD + x]);
    }
    return wave_sum(v);
}

// ---- intra prediction into s.fd ----------------------------------------------------------------
// x264_predict_16x16_* (R/common/predict.c:52-170): the sums the DC and plane modes need are reduced
// across the wave once per call instead of per pixel
// x264_predict_lossless_* (R/encoder/macroblock.c:405-470): vertical / horizontal prediction take the SOURCE one row up / one column
// left.  Everything coded so far reconstructs to its source, so outside the macroblock that is the neighbour row / column
// already in s.fd and inside it the macroblock's own source in s.fe.  (x, y) relative to the macroblock; plane 0 luma, 1 U, 2 V.
__device__ __forceinline__ int sw_ll_px(const SwLds &s, int plane, int horiz, int x, int y)
{
    const int st = plane ? 8 : 16;
    const u8 *fe = s.fe + (plane == 0 ? 0 : plane == 1 ? 256 : 320), *fd = s.fd + (plane == 0 ? FDY : plane == 1 ? FDU : FDV);
    if (horiz) return x == 0 ? fd[y * FD - 1] : fe[y * st + x - 1];
    return y == 0 ? fd[x - FD] : fe[(y - 1) * st + x];
}
__device__ __forceinline__ void sw_pred16(SwLds &s, int mode, int lane, int ll = 0)
{
    const int r = lane >> 2, x = (lane & 3) * 4;
    const u8 *top = s.fd + FDY - FD, *left = s.fd + FDY - 1;
    int v[4];
    if (mode == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = top[x + i];
    } else if (mode == 1) {
        v[0] = v[1] = v[2] = v[3] = left[r * FD];
    } else if (mode == 3) {
        int h = 0, w = 0;
        if (lane < 8) { h = (lane + 1) * ((int)top[8 + lane] - (int)top[6 - lane]); w = (lane + 1) * ((int)left[(8 + lane) * FD] - (int)left[(6 - lane) * FD]); }
        const int H = wave_sum(h), V = wave_sum(w);
        const int a = 16 * ((int)left[15 * FD] + (int)top[15]), b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
        const int i00 = a - 7 * b - 7 * c + 16 + c * r;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = clip_u8((i00 + b * (x + i)) >> 5);
    } else {
        int dc = 128;
        if (mode != 6) {
            const int t = wave_sum(lane < 16 ? (int)top[lane] : 0), l = wave_sum(lane < 16 ? (int)left[lane * FD] : 0);
            dc = mode == 2 ? (t + l + 16) >> 5 : mode == 4 ? (l + 8) >> 4 : (t + 8) >> 4;
        }
        v[0] = v[1] = v[2] = v[3] = dc;
    }
    if (ll && mode < 2) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = sw_ll_px(s, 0, mode, x + i, r);
    }
    WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < 4; i++) s.fd[FDY + r * FD + x + i] = (u8)v[i];
    WAVE_SYNC();
}
__device__ __forceinline__ void sw_pred8c(SwLds &s, int mode, int lane, int ll = 0)
{
    const int x = lane & 7, y = lane >> 3;
    int pu = pred_px(1, mode, s.fd + FDU, FD, x, y), pv = pred_px(1, mode, s.fd + FDV, FD, x, y);
    if (ll && (mode == 1 || mode == 2)) { pu = sw_ll_px(s, 1, mode == 1, x, y); pv = sw_ll_px(s, 2, mode == 1, x, y); }
    WAVE_SYNC();
    s.fd[FDU + y * FD + x] = (u8)pu; s.fd[FDV + y * FD + x] = (u8)pv;
    WAVE_SYNC();
}
// predict_16x16_mode_available / predict_8x8chroma_mode_available, R/encoder/analyse.c:374-433
// the list is a packed word, one nibble per mode in the reference's order (an array indexed in a loop would live in scratch memory)
__device__ __forceinline__ u32 sw_modes16(int nb, int &n)
{
    if (nb & NB_TOPLEFT) { n = 4; return 0x3210; }
    if (nb & NB_LEFT) { n = 2; return 0x14; }
    if (nb & NB_TOP) { n = 2; return 0x05; }
    n = 1; return 6;
}
__device__ __forceinline__ u32 sw_modes8c(int nb, int &n)
{
    if (nb & NB_TOPLEFT) { n = 4; return 0x3012; }
    if (nb & NB_LEFT) { n = 2; return 0x14; }
    if (nb & NB_TOP) { n = 2; return 0x25; }
    n = 1; return 6;
}
__device__ __forceinline__ int sw_fix16(int m) { return m < 4 ? m : 2; }      // x264_mb_pred_mode16x16_fix
__device__ __forceinline__ int sw_fix8c(int m) { return m < 4 ? m : 0; }      // x264_mb_pred_mode8x8c_fix

// ---- encode pieces (R/encoder/macroblock.c:116-363, 596-768; no trellis, not lossless) ------------
// luma 4x4 transform + quant + scan + dequant of the 16 blocks, lanes 0-15.  cat: 0 intra, 1 inter.
// i16 mode takes the DC out first (s.dc16 in raster order) and scores with decimate_score15.
// x264_denoise_dct (R/common/quant.c:180-192) on coefficient v with offset off: returns the new coefficient, la = |v|
__device__ __forceinline__ int sw_denoise(int v, int off, int &la)
{
    const int sign = v >> 15;
    int level = (v + sign) ^ sign;
    la = level;
    level -= off;
    return level < 0 ? 0 : (level ^ sign) - sign;
}
// mask8: the 8x8 blocks to do (x264_macroblock_encode_p8x8 codes one); lanes of other blocks leave everything of theirs alone
__device__ __forceinline__ void sw_luma4x4_fwd(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int cat, bool dc_out, int lane, int *nr_acc4 = nullptr, int nr_on = 0,
                                               int mask8 = 0xf)
{
    i16 c[16], lv[16];
    const bool mine = lane < 16 && ((mask8 >> (lane >> 2)) & 1);
    if (mine) {
        int bx, by, r[16];
        sw_blk_xy(lane, bx, by);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                r[4 * j + i] = (int)s.fe[(by + j) * 16 + bx + i] - (int)s.fd[FDY + (by + j) * FD + bx + i];
        fwd4x4(c, r);
        if (nr_acc4 && nr_on) {
            // --nr: every coefficient but the first of every block, and the sum of magnitudes per coefficient index over the 16
            // blocks (lane i keeps index i's running sum for the whole row; added to the chain's totals at the end of the row)
#pragma unroll
            for (int i = 1; i < 16; i++) {
                int la;
                c[i] = (i16)sw_denoise(c[i], s.nr_off4[i], la);
                const int t = row_sum16(la);
                if (lane == i) *nr_acc4 += t;
            }
        }
        if (dc_out) { s.dc16[(by >> 2) * 4 + (bx >> 2)] = c[0]; c[0] = 0; }
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) s.coef[lane][i] = c[i];
        }
    }
    if (tq.on) {
        // x264_quant_4x4_trellis (R/encoder/rdo.c:641-650): four blocks at a time, sixteen lanes each (trellis_wave.h)
        WAVE_SYNC();
#pragma nounroll
        for (int it = 0; it < 4; it++)
            if ((mask8 >> it) & 1)
            td_trellis_wave(tq.r->tw, (u32 *)s.patch, &s.coef[4 * it + (lane >> 4)][0], true, s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz4, tq.r->cabac, dc_out ? 1 : 2,
                            d_trellis_lambda2[cat == 0][Q.qp], dc_out ? 1 : 0, 0, 16, lane);
        WAVE_SYNC();
    }
    if (mine) {
        const u16 *mf = s.qmf[cat], *bs = s.qbias[cat];
        const int *dq = s.qdq[cat];
        int nz = 0, bits = Q.qp / 6 - 4;
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) { c[i] = s.coef[lane][i]; nz |= c[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }
        }
        SCAN4_FRAME(lv, c);
        u32 nzm, big;
        LEVEL_MASKS(lv, nzm, big);
#pragma unroll
        for (int i = 0; i < 16; i++) { s.lv_y[16 * lane + i] = lv[i]; s.coef[lane][i] = (i16)dequant_one(c[i], dq[i], bits); }
        s.score[lane] = (nz ? (dc_out ? decimate_masks(nzm >> 1, big >> 1) : decimate_masks(nzm, big)) : 0) | ((nz != 0) << 8);
    }
    WAVE_SYNC();
}
__device__ __forceinline__ void sw_luma4x4_add(SwLds &s, int lane, int keep8)
{
    if (lane < 16 && ((keep8 >> (lane >> 2)) & 1)) {
        int bx, by, res[16];
        sw_blk_xy(lane, bx, by);
        inv4x4(res, s.coef[lane]);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                u8 *p = s.fd + FDY + (by + j) * FD + bx + i;
                *p = (u8)clip_u8((int)*p + res[4 * j + i]);
            }
    }
    WAVE_SYNC();
}
__device__ __forceinline__ void sw_ll_i4x4(SwLds &s, int idx, int &cbp_luma, int lane);
__device__ __forceinline__ void sw_ll_i8x8(SwLds &s, int idx, int &cbp_luma, int lane);
__device__ __forceinline__ int sw_ll_luma16(SwLds &s, bool dc_out, int lane);
__device__ __forceinline__ int sw_ll_chroma(SwLds &s, int lane);
// x264_macroblock_encode's inter 4x4-transform branch; returns cbp_luma, fills s.nnz[0..15]
__device__ __forceinline__ int sw_encode_inter_luma(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int lane, int *nr_acc4 = nullptr, int nr_on = 0)
{
    if (a.lossless) return sw_ll_luma16(s, false, lane);
    sw_luma4x4_fwd(s, a, Q, tq, 1, false, lane, nr_acc4, nr_on);
    if (lane == 0) {
        int cbp = 0, dec_mb = 0;
        for (int i8 = 0; i8 < 4; i8++) {
            int dec8 = 0, any = 0;
            for (int i4 = 0; i4 < 4; i4++) {
                int v = s.score[4 * i8 + i4];
                s.nnz[4 * i8 + i4] = (u8)(v >> 8);
                if (v >> 8) { any = 1; if (a.dct_decimate && dec8 < 6) dec8 += v & 255; }
            }
            dec_mb += dec8;
            if (a.dct_decimate) {
                if (dec8 < 4) s.nnz[4 * i8] = s.nnz[4 * i8 + 1] = s.nnz[4 * i8 + 2] = s.nnz[4 * i8 + 3] = 0;
                else cbp |= 1 << i8;
            } else if (any) cbp |= 1 << i8;
        }
        if (a.dct_decimate && dec_mb < 6) { cbp = 0; for (int i = 0; i < 16; i++) s.nnz[i] = 0; }
        s.keep8 = cbp;
    }
    WAVE_SYNC();
    const int keep = __builtin_amdgcn_readfirstlane(s.keep8);
    sw_luma4x4_add(s, lane, keep);
    return keep;
}
// x264_mb_encode_i16x16 (prediction already in s.fd); returns cbp_luma, fills s.nnz[0..15], s.nnz[24]
__device__ __forceinline__ int sw_encode_i16x16(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int lane, bool b_slice = false)
{
    if (a.lossless) return sw_ll_luma16(s, true, lane);
    sw_luma4x4_fwd(s, a, Q, tq, 0, true, lane);
    i16 d[16], t[16];
    int nz = 0, cbp = 0;
    if (lane == 0) {
        const int b_decimate = b_slice || (a.dct_decimate && a.slice_type == 0);     // macroblock.c:193: B always, P with dct_decimate
        int score = b_decimate ? 0 : 9;
        for (int i = 0; i < 16; i++) {
            int v = s.score[i];
            s.nnz[i] = (u8)(v >> 8);
            if (v >> 8) { if (score < 6) score += v & 255; cbp = 0xf; }
        }
        if (score < 6) { cbp = 0; for (int i = 0; i < 16; i++) s.nnz[i] = 0; }
        // dct4x4dc (R/common/dct.c:39-71), quant_4x4_dc, scan, idct4x4dc, dequant_4x4_dc (quant.c:151-178)
#pragma unroll
        for (int i = 0; i < 16; i++) d[i] = s.dc16[i];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int p = d[4 * r] + d[4 * r + 1], q = d[4 * r] - d[4 * r + 1], u = d[4 * r + 2] + d[4 * r + 3], w = d[4 * r + 2] - d[4 * r + 3];
            t[r] = (i16)(p + u); t[4 + r] = (i16)(p - u); t[8 + r] = (i16)(q - w); t[12 + r] = (i16)(q + w);
        }
        for (int r = 0; r < 4; r++) {
            int p = t[4 * r] + t[4 * r + 1], q = t[4 * r] - t[4 * r + 1], u = t[4 * r + 2] + t[4 * r + 3], w = t[4 * r + 2] - t[4 * r + 3];
            d[4 * r] = (i16)((p + u + 1) >> 1); d[4 * r + 1] = (i16)((p - u + 1) >> 1);
            d[4 * r + 2] = (i16)((q - w + 1) >> 1); d[4 * r + 3] = (i16)((q + w + 1) >> 1);
        }
        const int mf = (int)s.qmf[0][0] >> 1, bias = (int)s.qbias[0][0] << 1;
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) s.dc16[i] = d[i];
        } else
            for (int i = 0; i < 16; i++) { int q = quant_one(d[i], mf, bias); d[i] = (i16)q; nz |= q; }
    }
    if (tq.on) {                                       // x264_quant_dc_trellis( .., DCT_LUMA_DC, 1 ), macroblock.c:247-248
        WAVE_SYNC();
        td_trellis_wave(tq.r->tw, (u32 *)s.patch, &s.dc16[0], lane < 16, s.qmf[0], tq.r->unq4[0], tq.r->w4z, tq.r->zz4, tq.r->cabac, 0, d_trellis_lambda2[1][Q.qp], 0, 1, 16, lane);
        WAVE_SYNC();
    }
    if (lane == 0) {
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) { d[i] = s.dc16[i]; nz |= d[i]; }
        }
        s.nnz[24] = (u8)(nz != 0);
        if (nz) {
            { i16 lvd[16]; SCAN4_FRAME(lvd, d);
#pragma unroll
              for (int i = 0; i < 16; i++) s.lv_dc[i] = lvd[i]; }
            for (int r = 0; r < 4; r++) {
                int p = d[4 * r] + d[4 * r + 1], q = d[4 * r] - d[4 * r + 1], u = d[4 * r + 2] + d[4 * r + 3], w = d[4 * r + 2] - d[4 * r + 3];
                t[r] = (i16)(p + u); t[4 + r] = (i16)(p - u); t[8 + r] = (i16)(q - w); t[12 + r] = (i16)(q + w);
            }
            for (int r = 0; r < 4; r++) {
                int p = t[4 * r] + t[4 * r + 1], q = t[4 * r] - t[4 * r + 1], u = t[4 * r + 2] + t[4 * r + 3], w = t[4 * r + 2] - t[4 * r + 3];
                d[4 * r] = (i16)(p + u); d[4 * r + 1] = (i16)(p - u); d[4 * r + 2] = (i16)(q - w); d[4 * r + 3] = (i16)(q + w);
            }
            const int m = s.qdq[0][0], bits = Q.qp / 6 - 6;
            for (int i = 0; i < 16; i++) s.dc16[i] = (i16)dequant_one(d[i], m, bits);
        }
        s.keep8 = cbp; s.nzdc16 = nz != 0;
    }
    WAVE_SYNC();
    const int keep = __builtin_amdgcn_readfirstlane(s.keep8), nzdc = __builtin_amdgcn_readfirstlane(s.nzdc16);
    if (keep) {
        if (lane < 16) {
            int bx, by;
            sw_blk_xy(lane, bx, by);
            if (nzdc) s.coef[lane][0] = s.dc16[(by >> 2) * 4 + (bx >> 2)];
        }
        WAVE_SYNC();
        sw_luma4x4_add(s, lane, 0xf);
    } else if (nzdc) {
        // add16x16_idct_dc, R/common/dct.c:369-382
        const int r = lane >> 2, x = (lane & 3) * 4;
        const int dc = (int)(i16)((s.dc16[(r >> 2) * 4 + (x >> 2)] + 32) >> 6);
#pragma unroll
        for (int i = 0; i < 4; i++) { u8 *p = s.fd + FDY + r * FD + x + i; *p = (u8)clip_u8((int)*p + dc); }
        WAVE_SYNC();
    }
    return keep;
}
// x264_mb_encode_8x8_chroma; returns cbp_chroma, fills s.nnz[16..23], s.nnz[25..26]
__device__ __forceinline__ int sw_encode_chroma(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int b_inter, int lane)
{
    if (a.lossless) return sw_ll_chroma(s, lane);
    const int cat = 2 + b_inter, b_decimate = b_inter && a.dct_decimate;
    const u16 *mf = s.qmf[cat], *bs = s.qbias[cat];
    i16 c[16], lv[16];
    if (lane < 8) {
        int ch = lane >> 2, i4 = lane & 3, bx = (i4 & 1) * 4, by = (i4 >> 1) * 4, r[16];
        const u8 *fe = s.fe + 256 + 64 * ch, *pr = s.fd + (ch ? FDV : FDU);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                r[4 * j + i] = (int)fe[(by + j) * 8 + bx + i] - (int)pr[(by + j) * FD + bx + i];
        fwd4x4(c, r);
        s.cdc[lane] = c[0];
        c[0] = 0;                                     // dct2x2dc takes the DCs out (macroblock.c:73-85)
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) s.ccoef[lane][i] = c[i];
        }
    }
    if (tq.on) {                                      // x264_quant_4x4_trellis( .., DCT_CHROMA_AC, !b_inter, 0 ), macroblock.c:310-311
        WAVE_SYNC();
#pragma nounroll
        for (int it = 0; it < 2; it++)
            td_trellis_wave(tq.r->tw, (u32 *)s.patch, &s.ccoef[4 * it + (lane >> 4)][0], true, s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz4, tq.r->cabac, 4, d_trellis_lambda2[!b_inter][Q.qpc], 1, 0, 16, lane);
        WAVE_SYNC();
    }
    if (lane < 8) {
        const int *dq = s.qdq[cat];
        int nz = 0, bits = Q.qpc / 6 - 4;
        if (tq.on) {
#pragma unroll
            for (int i = 0; i < 16; i++) { c[i] = s.ccoef[lane][i]; nz |= c[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }
        }
        SCAN4_FRAME(lv, c);
        u32 nzm, big;
        LEVEL_MASKS(lv, nzm, big);
#pragma unroll
        for (int i = 0; i < 16; i++) { s.lv_cac[16 * lane + i] = lv[i]; s.ccoef[lane][i] = nz ? (i16)dequant_one(c[i], dq[i], bits) : (i16)0; }
        s.cscore[lane] = (nz ? decimate_masks(nzm >> 1, big >> 1) : 0) | ((nz != 0) << 8);
    }
    WAVE_SYNC();
    i16 d2[4] = {0, 0, 0, 0};                          // [0][0] [0][1] [1][0] [1][1]
    if (lane < 2) {
        const int ch = lane;
        int b0 = s.cdc[4 * ch], b1 = s.cdc[4 * ch + 1], b2 = s.cdc[4 * ch + 2], b3 = s.cdc[4 * ch + 3];
        int a0 = b0 + b1, a1 = b2 + b3, a2 = b0 - b1, a3 = b2 - b3;
        d2[0] = (i16)(a0 + a1); d2[1] = (i16)(a0 - a1); d2[2] = (i16)(a2 + a3); d2[3] = (i16)(a2 - a3);
        if (tq.on) { s.cdcout[4 * ch] = d2[0]; s.cdcout[4 * ch + 1] = d2[1]; s.cdcout[4 * ch + 2] = d2[2]; s.cdcout[4 * ch + 3] = d2[3]; }
    }
    if (tq.on) {                                      // x264_quant_dc_trellis( .., DCT_CHROMA_DC, !b_inter ), macroblock.c:325-326
        WAVE_SYNC();
        td_trellis_wave(tq.r->tw, (u32 *)s.patch, &s.cdcout[4 * ((lane >> 4) & 1)], lane < 32, s.qmf[cat], tq.r->unq4[cat], tq.r->w4z, tq.r->zz2, tq.r->cabac, 3, d_trellis_lambda2[!b_inter][Q.qpc], 0, 1, 4, lane);
        WAVE_SYNC();
    }
    if (lane < 2) {
        const int ch = lane;
        int nz_dc = 0;
        if (tq.on) { for (int i = 0; i < 4; i++) { d2[i] = s.cdcout[4 * ch + i]; nz_dc |= d2[i]; } }
        else for (int i = 0; i < 4; i++) { int q = quant_one(d2[i], (int)mf[0] >> 1, (int)bs[0] << 1); d2[i] = (i16)q; nz_dc |= q; }
        int score = 0, nz_ac = 0;
        u8 nzf[4];
        for (int i = 0; i < 4; i++) { int v = s.cscore[4 * ch + i]; nzf[i] = (u8)(v >> 8); if (v >> 8) { nz_ac = 1; if (b_decimate) score += v & 255; } }
        int e0 = d2[0] + d2[1], e1 = d2[2] + d2[3], e2 = d2[0] - d2[1], e3 = d2[2] - d2[3];
        int dmf = s.qdq[cat][0], qbits = Q.qpc / 6 - 5;
        if (qbits > 0) { dmf <<= qbits; qbits = 0; }
        int mode;
        if ((b_decimate && score < 7) || !nz_ac) { nzf[0] = nzf[1] = nzf[2] = nzf[3] = 0; mode = nz_dc ? 1 : 0; }
        else mode = 2;
        const bool put = nz_dc != 0;
        s.lv_cdc[4 * ch] = put ? d2[0] : (i16)0; s.lv_cdc[4 * ch + 1] = put ? d2[2] : (i16)0;
        s.lv_cdc[4 * ch + 2] = put ? d2[1] : (i16)0; s.lv_cdc[4 * ch + 3] = put ? d2[3] : (i16)0;
        s.cdcout[4 * ch + 0] = (i16)((e0 + e1) * dmf >> -qbits); s.cdcout[4 * ch + 1] = (i16)((e0 - e1) * dmf >> -qbits);
        s.cdcout[4 * ch + 2] = (i16)((e2 + e3) * dmf >> -qbits); s.cdcout[4 * ch + 3] = (i16)((e2 - e3) * dmf >> -qbits);
        if (!nz_dc) s.cdcout[4 * ch] = s.cdcout[4 * ch + 1] = s.cdcout[4 * ch + 2] = s.cdcout[4 * ch + 3] = 0;
        s.cmode[ch] = mode | (nz_dc ? 16 : 0);
        for (int i = 0; i < 4; i++) s.nnz[16 + 4 * ch + i] = nzf[i];
        s.nnz[25 + ch] = (u8)(nz_dc != 0);
    }
    WAVE_SYNC();
    if (lane < 8) {
        int ch = lane >> 2, i4 = lane & 3, bx = (i4 & 1) * 4, by = (i4 >> 1) * 4, mode = s.cmode[ch] & 15;
        u8 *pr = s.fd + (ch ? FDV : FDU);
        if (mode == 2) {
            int res[16];
            s.ccoef[lane][0] = s.cdcout[lane];
            inv4x4(res, s.ccoef[lane]);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) { u8 *p = pr + (by + j) * FD + bx + i; *p = (u8)clip_u8((int)*p + res[4 * j + i]); }
        } else if (mode == 1) {
            int dc = (int)(i16)((s.cdcout[lane] + 32) >> 6);
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) { u8 *p = pr + (by + j) * FD + bx + i; *p = (u8)clip_u8((int)*p + dc); }
        }
    }
    WAVE_SYNC();
    const int m0 = __builtin_amdgcn_readfirstlane(s.cmode[0]), m1 = __builtin_amdgcn_readfirstlane(s.cmode[1]);
    return ((m0 & 15) == 2 || (m1 & 15) == 2) ? 2 : (((m0 | m1) & 16) ? 1 : 0);
}
// x264_macroblock_probe_skip, P path (R/encoder/macroblock.c:797-883); leaves the P-skip prediction in s.fd
__device__ __forceinline__ int sw_probe_pskip(SwLds &s, const SwRefs &refs, const SwArgs &a, const SwQp &Q, int pmx, int pmy, int mbx, int mby,
                                              ptrdiff_t oy, ptrdiff_t oc, size_t by_, size_t bc_, int lane, bool b_bidir = false)
{
    if (!b_bidir) {                 // x264_macroblock_probe_bskip: the (direct) prediction is in fdec already
        const int vx = clip3(pmx, 4 * (-16 * mbx - 24), 4 * (16 * (a.mb_w - mbx - 1) + 24));
        const int vy = clip3(pmy, 4 * (-16 * mby - 24), 4 * (16 * (a.mb_h - mby - 1) + 24));
        sw_mc16(s, refs, a, 0, vx, vy, oy, oc, by_, bc_, lane, true);
        WAVE_SYNC();
    }
    int score = 0, dc = 0, ssd = 0;
    if (lane < 24) {
        const bool luma = lane < 16;
        int bx, by, r[16];
        const u8 *fe, *pr; int st;
        if (luma) { sw_blk_xy(lane, bx, by); fe = s.fe; pr = s.fd + FDY; st = 16; }
        else { int l = lane - 16, i4 = l & 3; bx = (i4 & 1) * 4; by = (i4 >> 1) * 4; fe = s.fe + 256 + 64 * (l >> 2); pr = s.fd + ((l >> 2) ? FDV : FDU); st = 8; }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r[4 * j + i] = (int)fe[(by + j) * st + bx + i] - (int)pr[(by + j) * FD + bx + i];
                ssd += r[4 * j + i] * r[4 * j + i];
            }
        i16 c[16], lv[16];
        fwd4x4(c, r);
        const int cat = luma ? 1 : 3;
        const u16 *mf = s.qmf[cat], *bs = s.qbias[cat];
        if (!luma) { dc = c[0]; c[0] = 0; }
        int nz = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { int qq = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)qq; nz |= qq; }
        SCAN4_FRAME(lv, c);
        u32 nzm, big;
        LEVEL_MASKS(lv, nzm, big);
        if (nz) score = luma ? decimate_masks(nzm, big) : decimate_masks(nzm >> 1, big >> 1);
    }
    int luma_sum = 0, c_sum[2] = {0, 0}, c_ssd[2] = {0, 0}, c_dc[2][4];
#pragma unroll
    for (int k = 0; k < 16; k++) luma_sum += __builtin_amdgcn_readlane(score, k);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c_sum[k >> 2] += __builtin_amdgcn_readlane(score, 16 + k);
        c_ssd[k >> 2] += __builtin_amdgcn_readlane(ssd, 16 + k);
        c_dc[k >> 2][k & 3] = __builtin_amdgcn_readlane(dc, 16 + k);
    }
    int ok = luma_sum < 6;
    const u16 *mf = s.qmf[3], *bs = s.qbias[3];
    for (int ch = 0; ch < 2 && ok; ch++) {
        if (c_ssd[ch] < Q.skip_thresh) continue;
        int b0 = c_dc[ch][0], b1 = c_dc[ch][1], b2 = c_dc[ch][2], b3 = c_dc[ch][3];
        int a0 = b0 + b1, a1 = b2 + b3, a2 = b0 - b1, a3 = b2 - b3;
        int d2[4] = {(i16)(a0 + a1), (i16)(a0 - a1), (i16)(a2 + a3), (i16)(a2 - a3)};
        int nzdc = 0;
        for (int i = 0; i < 4; i++) nzdc |= quant_one(d2[i], (int)mf[0] >> 1, (int)bs[0] << 1);
        if (nzdc || c_sum[ch] >= 7) ok = 0;
    }
    return ok;
}

// ---- 8x8 transform path (R/common/dct.c:238-349, quant 8x8, scan, decimate_score64) --------------
// forward transform + quant + scan of the 8x8 luma blocks in `mask`, all at once: lane = (block b, column / row k)
// for the two 1-D passes (32 lanes), then 64 lanes x one coefficient per block.  Leaves the quantised
// coefficients (transposed storage) in s.coef[4*b..][..] = [4][64], levels in s.lv_y8, per block
// s.score[b] = decimate_score64 | nz << 8.  cat: 0 intra, 1 inter.
__device__ __forceinline__ void sw_luma8x8_fwd(SwLds &s, const SwQp &Q, SwTq tq, int cat, int mask, int lane, int *nr_acc8 = nullptr, int nr_on = 0)
{
    i16 *tmp = s.t8, *coef = &s.coef[0][0];
    const int b = lane >> 3, k8 = lane & 7;
    const bool on = lane < 32 && ((mask >> b) & 1);
    if (on) {
        const u8 *p1 = s.fe + (b >> 1) * 8 * 16 + (b & 1) * 8 + k8, *p2 = s.fd + FDY + (b >> 1) * 8 * FD + (b & 1) * 8 + k8;
        int v[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (int)p1[k * 16] - (int)p2[k * FD];
        fwd8_1d(o, v);                                 // column k8
#pragma unroll
        for (int k = 0; k < 8; k++) tmp[64 * b + k * 8 + k8] = (i16)o[k];
    }
    WAVE_SYNC();
    if (on) {
        int v[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = tmp[64 * b + k8 * 8 + k];
        fwd8_1d(o, v);                                 // row k8, stored transposed (dct.c:278-283)
#pragma unroll
        for (int k = 0; k < 8; k++) coef[64 * b + k * 8 + k8] = (i16)o[k];
    }
    WAVE_SYNC();
    const int mfl = s.q8mf[cat][lane], bsl = s.q8bias[cat][lane];
    unsigned long long nzmask[4] = {0, 0, 0, 0}, bigmask[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++)
        if ((mask >> j) & 1) {
            if (nr_acc8 && nr_on && lane) {      // --nr: lane = coefficient index, the first one is left alone
                int la;
                coef[64 * j + lane] = (i16)sw_denoise(coef[64 * j + lane], s.nr_off8[lane], la);
                *nr_acc8 += la;
            }
            if (!tq.on) {
                int q = quant_one(coef[64 * j + lane], mfl, bsl);
                coef[64 * j + lane] = (i16)q;
                nzmask[j] = __ballot(q != 0);
            }
        }
    WAVE_SYNC();
    if (tq.on) {                                      // x264_quant_8x8_trellis (R/encoder/rdo.c:652-660): one block at a time (its level lists fill the scratch area), sixteen lanes
#pragma nounroll
        for (int j = 0; j < 4; j++)
            if ((mask >> j) & 1)
                td_trellis_wave(tq.r->tw, (u32 *)s.patch, coef + 64 * j, lane < 16, s.q8mf[cat], tq.r->unq8[cat], tq.r->w8z, tq.r->zz8, tq.r->cabac, 5, d_trellis_lambda2[cat == 0][Q.qp], 0, 0, 64, lane);
        WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((mask >> j) & 1) nzmask[j] = __ballot(coef[64 * j + lane] != 0);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if ((mask >> j) & 1) {
            int lvv = nzmask[j] ? (int)coef[64 * j + c_scan8[0][lane]] : 0;
            s.lv_y8[64 * j + lane] = (i16)lvv;
            nzmask[j] = __ballot(lvv != 0);
            bigmask[j] = __ballot((unsigned)(lvv + 1) > 2u);
        }
    if (lane < 4 && ((mask >> lane) & 1)) {
        unsigned long long m = lane == 0 ? nzmask[0] : lane == 1 ? nzmask[1] : lane == 2 ? nzmask[2] : nzmask[3];
        unsigned long long bg = lane == 0 ? bigmask[0] : lane == 1 ? bigmask[1] : lane == 2 ? bigmask[2] : bigmask[3];
        int sc = 0;
        if (bg) sc = 9;
        else {
            int idx = m ? 63 - __clzll(m) : -1;
            while (idx >= 0) {
                unsigned long long below = idx ? (m & ((1ull << idx) - 1)) : 0ull;
                int prev = below ? 63 - __clzll(below) : -1;
                sc += c_decimate8[idx - prev - 1];
                idx = prev;
            }
        }
        s.score[lane] = sc | ((m != 0) << 8);
    }
    WAVE_SYNC();
}
// dequant + inverse 8x8 + add for the blocks in `keep`
__device__ __forceinline__ void sw_luma8x8_add(SwLds &s, int cat, int qp, int keep, int lane)
{
    i16 *coef = &s.coef[0][0];
    const int b = lane >> 3, k8 = lane & 7, bits = qp / 6 - 6, dql = s.q8dq[cat][lane];
#pragma unroll
    for (int j = 0; j < 4; j++)
        if ((keep >> j) & 1) {
            int v = dequant_one(coef[64 * j + lane], dql, bits);
            if (lane == 0) v = (int)(i16)(v + 32);           // rounding term, dct.c:326
            coef[64 * j + lane] = (i16)v;
        }
    WAVE_SYNC();
    const bool on = lane < 32 && ((keep >> b) & 1);
    if (on) {
        int v[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = coef[64 * b + k * 8 + k8];
        inv8_1d(o, v);
#pragma unroll
        for (int k = 0; k < 8; k++) coef[64 * b + k * 8 + k8] = (i16)o[k];
    }
    WAVE_SYNC();
    if (on) {
        u8 *dst = s.fd + FDY + (b >> 1) * 8 * FD + (b & 1) * 8;
        int v[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = coef[64 * b + k8 * 8 + k];
        inv8_1d(o, v);
#pragma unroll
        for (int k = 0; k < 8; k++) { u8 *p = dst + k8 + k * FD; *p = (u8)clip_u8((int)*p + (o[k] >> 6)); }
    }
    WAVE_SYNC();
}
// inter, 8x8 transform (R/encoder/macroblock.c:627-669); returns cbp_luma, fills s.nnz[0..15]
__device__ __forceinline__ int sw_encode_inter_luma8(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int lane, int *nr_acc8 = nullptr, int nr_on = 0)
{
    sw_luma8x8_fwd(s, Q, tq, 1, 0xf, lane, nr_acc8, nr_on);
    const int b_decimate = a.dct_decimate && !tq.on;           // "8x8 trellis is inherently optimal decimation", macroblock.c:630
    int cbp = 0, dec_mb = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = __builtin_amdgcn_readfirstlane(s.score[i]);
        if (v >> 8) {
            if (b_decimate) { dec_mb += v & 255; if ((v & 255) >= 4) cbp |= 1 << i; }
            else cbp |= 1 << i;
        }
    }
    if (b_decimate && dec_mb < 6) cbp = 0;
    if (lane < 16) s.nnz[lane] = (u8)((cbp >> (lane >> 2)) & 1);
    WAVE_SYNC();
    sw_luma8x8_add(s, 1, Q.qp, cbp, lane);
    return cbp;
}
// x264_mb_encode_i8x8 for block idx (prediction already in s.fd)
__device__ __forceinline__ void sw_encode_i8x8(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int idx, int &cbp_luma, int lane)
{
    if (a.lossless) { sw_ll_i8x8(s, idx, cbp_luma, lane); return; }
    sw_luma8x8_fwd(s, Q, tq, 0, 1 << idx, lane);
    const int nz = (__builtin_amdgcn_readfirstlane(s.score[idx]) >> 8) & 1;
    if (lane < 4) s.nnz[4 * idx + lane] = (u8)nz;
    WAVE_SYNC();
    if (nz) { cbp_luma |= 1 << idx; sw_luma8x8_add(s, 0, Q.qp, 1 << idx, lane); }
}
// x264_mb_encode_i4x4 for block idx (prediction already in s.fd): one lane per coefficient (the four 16-lane rows of the
// wave run the same block; only the first stores).  The 1-D transforms work on the four values of a quad (DPP
// broadcasts); between the passes the 4x4 is transposed with one ds_bpermute, so after the forward pair lane i holds
// dct[i] in the reference's (transposed) storage order and the quantiser rows / zigzag are indexed by the lane.
__device__ __forceinline__ void sw_quad4(int v, int &v0, int &v1, int &v2, int &v3)
{
    v0 = __builtin_amdgcn_update_dpp(0, v, 0x00, 0xf, 0xf, false); v1 = __builtin_amdgcn_update_dpp(0, v, 0x55, 0xf, 0xf, false);
    v2 = __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xf, 0xf, false); v3 = __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xf, 0xf, false);
}
__device__ __forceinline__ int sw_fwd4_quad(int v, int k)        // dct.c:122-145, one 1-D pass: output k of this quad's four inputs
{
    int v0, v1, v2, v3;
    sw_quad4(v, v0, v1, v2, v3);
    const int s03 = v0 + v3, s12 = v1 + v2, d03 = v0 - v3, d12 = v1 - v2;
    return (int)(i16)(k == 0 ? s03 + s12 : k == 1 ? 2 * d03 + d12 : k == 2 ? s03 - s12 : d03 - 2 * d12);
}
__device__ __forceinline__ int sw_inv4_quad(int v, int k, int last)   // dct.c:174-206
{
    int d0, d1, d2, d3;
    sw_quad4(v, d0, d1, d2, d3);
    const int e = d0 + d2, f = d0 - d2, g = d1 + (d3 >> 1), h = (d1 >> 1) - d3;
    const int r = k == 0 ? e + g : k == 1 ? f + h : k == 2 ? f - h : e - g;
    return (int)(i16)(last ? (r + 32) >> 6 : r);
}
__device__ __forceinline__ void sw_encode_i4x4(SwLds &s, const SwArgs &a, const SwQp &Q, SwTq tq, int idx, int &cbp_luma, int lane)
{
    if (a.lossless) { sw_ll_i4x4(s, idx, cbp_luma, lane); return; }
    int bx, by;
    sw_blk_xy(idx, bx, by);
    const int l16 = lane & 15, x = l16 & 3, y = l16 >> 2, tl = (lane & 48) | (x << 2) | y;
    int v = (int)s.fe[(by + y) * 16 + bx + x] - (int)s.fd[FDY + (by + y) * FD + bx + x];
    v = sw_fwd4_quad(v, x);
    v = __shfl(v, tl, 64);
    v = sw_fwd4_quad(v, x);                                        // dct[l16]
    int q;
    if (tq.on) {                                      // x264_quant_4x4_trellis( .., DCT_LUMA_4x4, 1, idx ), macroblock.c:134
        if (lane < 16) s.coef[idx][l16] = (i16)v;
        WAVE_SYNC();
        td_trellis_wave(tq.r->tw, (u32 *)s.patch, &s.coef[idx][0], lane < 16, s.qmf[0], tq.r->unq4[0], tq.r->w4z, tq.r->zz4, tq.r->cabac, 2, d_trellis_lambda2[1][Q.qp], 0, 0, 16, lane);
        WAVE_SYNC();
        q = s.coef[idx][l16];
    } else
        q = quant_one(v, s.qmf[0][l16], s.qbias[0][l16]);
    const int nz = (__ballot(q != 0) & 0xffffull) != 0;
    if (lane == 0) s.nnz[idx] = (u8)nz;
    if (nz) {
        if (lane < 16) s.lv_y[16 * idx + (int)((0xFDC6EB75A8419320ull >> (4 * l16)) & 15)] = (i16)q;      // zigzag position of dct[l16]
        int d = dequant_one(q, s.qdq[0][l16], Q.qp / 6 - 4);
        d = __shfl(d, tl, 64);                                     // lane (c = l16 >> 2, p = l16 & 3) holds dct[4p + c]
        d = sw_inv4_quad(d, x, 0);                                 // ... now mid[l16]
        d = __shfl(d, tl, 64);
        d = sw_inv4_quad(d, x, 1);                                 // ... now the residual of pixel (row x, column y)
        if (lane < 16) { u8 *p = s.fd + FDY + (by + x) * FD + bx + y; *p = (u8)clip_u8((int)*p + d); }
        cbp_luma |= 1 << (idx >> 2);
    }
    WAVE_SYNC();
}

// ---- intra 4x4 / 8x8 analysis helpers ------------------------------------------------------------
// i_neighbour4 / i_neighbour8 (R/common/macroblock.c:733-743, 1172-1186)
// ---- lossless (R/encoder/macroblock.c:123-130,160-167,196-213,288-303,602-626; zigzag_sub_*, R/common/dct.c:564-606) ----
// The levels are the prediction error itself in zigzag order and the reconstruction is the source.  Lane = zigzag position.
#define SW_ZZ4(p) ((int)((0xFBEDA7369C852140ull >> (4 * (p))) & 15))      /* position -> 4 * x + y */
__device__ __forceinline__ void sw_ll_i4x4(SwLds &s, int idx, int &cbp_luma, int lane)
{
    int bx, by;
    sw_blk_xy(idx, bx, by);
    const int c = SW_ZZ4(lane & 15), o_e = (by + (c & 3)) * 16 + bx + (c >> 2), o_d = FDY + (by + (c & 3)) * FD + bx + (c >> 2);
    const int v = (int)s.fe[o_e] - (int)s.fd[o_d];
    const int nz = (__ballot(v != 0) & 0xffffull) != 0;
    if (lane < 16) { s.lv_y[16 * idx + lane] = (i16)v; s.fd[o_d] = s.fe[o_e]; }
    if (lane == 0) s.nnz[idx] = (u8)nz;
    cbp_luma |= nz << (idx >> 2);
    WAVE_SYNC();
}
__device__ __forceinline__ void sw_ll_i8x8(SwLds &s, int idx, int &cbp_luma, int lane)
{
    const int c = c_scan8[0][lane], bx = 8 * (idx & 1), by = 8 * (idx >> 1);
    const int o_e = (by + (c & 7)) * 16 + bx + (c >> 3), o_d = FDY + (by + (c & 7)) * FD + bx + (c >> 3);
    const int v = (int)s.fe[o_e] - (int)s.fd[o_d];
    const int nz = __ballot(v != 0) != 0;
    s.lv_y8[64 * idx + lane] = (i16)v; s.fd[o_d] = s.fe[o_e];
    if (lane < 4) s.nnz[4 * idx + lane] = (u8)nz;
    cbp_luma |= nz << idx;
    WAVE_SYNC();
}
// the 16 luma 4x4 blocks (inter, or I_16x16 with dc_out: the first level of every block goes to the DC block); returns cbp_luma
__device__ __forceinline__ int sw_ll_luma16(SwLds &s, bool dc_out, int lane)
{
    int cbp = 0, dc_any = 0;
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int blk = 4 * pass + (lane >> 4), p = lane & 15, c = SW_ZZ4(p);
        int bx, by;
        sw_blk_xy(blk, bx, by);
        const int o_e = (by + (c & 3)) * 16 + bx + (c >> 2), o_d = FDY + (by + (c & 3)) * FD + bx + (c >> 2);
        int v = (int)s.fe[o_e] - (int)s.fd[o_d];
        s.fd[o_d] = s.fe[o_e];
        if (dc_out) {
            dc_any |= __ballot(p == 0 && v != 0) != 0;
            if (p == 0) { s.dc16[(bx >> 2) * 4 + (by >> 2)] = (i16)v; v = 0; }
        }
        s.lv_y[16 * blk + p] = (i16)v;
        const unsigned long long m = __ballot(v != 0);
        if (lane < 4) s.nnz[4 * pass + lane] = (u8)(((m >> (16 * lane)) & 0xffffull) != 0);
        if (m) cbp |= dc_out ? 0xf : 1 << pass;
    }
    if (dc_out) {
        WAVE_SYNC();
        if (lane < 16) s.lv_dc[lane] = s.dc16[SW_ZZ4(lane)];
        if (lane == 0) s.nnz[24] = (u8)dc_any;
    }
    WAVE_SYNC();
    return cbp;
}
// both chroma planes; returns cbp_chroma
__device__ __forceinline__ int sw_ll_chroma(SwLds &s, int lane)
{
    int ac = 0, dc = 0;
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        const int blk = lane >> 4, p = lane & 15, c = SW_ZZ4(p), bx = (blk & 1) * 4, by = (blk >> 1) * 4;
        const int o_e = 256 + 64 * ch + (by + (c & 3)) * 8 + bx + (c >> 2), o_d = (ch ? FDV : FDU) + (by + (c & 3)) * FD + bx + (c >> 2);
        int v = (int)s.fe[o_e] - (int)s.fd[o_d];
        s.fd[o_d] = s.fe[o_e];
        const unsigned long long md = __ballot(p == 0 && v != 0);
        if (p == 0) { s.lv_cdc[4 * ch + blk] = (i16)v; v = 0; }
        s.lv_cac[(4 * ch + blk) * 16 + p] = (i16)v;
        const unsigned long long m = __ballot(v != 0);
        if (lane < 4) s.nnz[16 + 4 * ch + lane] = (u8)(((m >> (16 * lane)) & 0xffffull) != 0);
        if (lane == 0) s.nnz[25 + ch] = (u8)(md != 0);
        ac |= m != 0; dc |= md != 0;
    }
    WAVE_SYNC();
    return ac ? 2 : dc ? 1 : 0;
}
__device__ __forceinline__ int sw_nb4(int idx, int nb)
{
    const int all = NB_LEFT | NB_TOP | NB_TOPLEFT | NB_TOPRIGHT;
    switch (idx) {
    case 0: return (nb & (NB_TOP | NB_LEFT | NB_TOPLEFT)) | ((nb & NB_TOP) ? NB_TOPRIGHT : 0);
    case 1: case 4: return NB_LEFT | ((nb & NB_TOP) ? (NB_TOP | NB_TOPLEFT | NB_TOPRIGHT) : 0);
    case 2: case 8: case 10: return NB_TOP | NB_TOPRIGHT | ((nb & NB_LEFT) ? (NB_LEFT | NB_TOPLEFT) : 0);
    case 5: return NB_LEFT | (nb & NB_TOPRIGHT) | ((nb & NB_TOP) ? NB_TOP | NB_TOPLEFT : 0);
    case 6: case 9: case 12: case 14: return all;
    default: return NB_LEFT | NB_TOP | NB_TOPLEFT;          // 3 7 11 13 15
    }
}
__device__ __forceinline__ int sw_nb8(int idx, int nb)
{
    switch (idx) {
    case 0: return (nb & (NB_TOP | NB_LEFT | NB_TOPLEFT)) | ((nb & NB_TOP) ? NB_TOPRIGHT : 0);
    case 1: return NB_LEFT | (nb & NB_TOPRIGHT) | ((nb & NB_TOP) ? NB_TOP | NB_TOPLEFT : 0);
    case 2: return NB_TOP | NB_TOPRIGHT | ((nb & NB_LEFT) ? (NB_LEFT | NB_TOPLEFT) : 0);
    default: return NB_LEFT | NB_TOP | NB_TOPLEFT;
    }
}
// predict_4x4_mode_available (R/encoder/analyse.c:435-471) as a nibble list: mode i = (list >> 4i) & 15
__device__ __forceinline__ unsigned long long sw_modes4(int nb, int &n)
{
    if ((nb & NB_LEFT) && (nb & NB_TOP)) {
        if (nb & NB_TOPLEFT) { n = 9; return 0x876543012ull; }
        n = 6; return 0x873012ull;
    }
    if (nb & NB_LEFT) { n = 3; return 0x819ull; }
    if (nb & NB_TOP) { n = 4; return 0x730Aull; }
    n = 1; return 0xBull;
}
__device__ __forceinline__ int sw_scan8(int i) { int x, y; sw_blk_xy(i, x, y); return 4 + 1 * 8 + (x >> 2) + 8 * (y >> 2); }
__device__ __forceinline__ int sw_fix4(int m) { return m < 0 ? -1 : m < 9 ? m : 2; }      // x264_mb_pred_mode4x4_fix
// x264_mb_predict_intra4x4_mode (R/common/macroblock.h:423-434)
__device__ __forceinline__ int sw_pred_i4mode(const SwLds &s, int idx)
{
    const int ma = sw_fix4(__builtin_amdgcn_readfirstlane((int)s.i4c[sw_scan8(idx) - 1]));
    const int mb = sw_fix4(__builtin_amdgcn_readfirstlane((int)s.i4c[sw_scan8(idx) - 8]));
    const int m = ma < mb ? ma : mb;
    return m < 0 ? 2 : m;
}
// SATD / SAD of a 4x4 block from one row of differences per lane (rows of a block in lanes l, l^1, l^2, l^3); pixel.c:187-212
__device__ __forceinline__ int sw_cost4x4_rows(int d0, int d1, int d2, int d3, int satd, int lane)
{
    if (!satd) return quad_sum4(iabs(d0) + iabs(d1) + iabs(d2) + iabs(d3));
    const u32 e0 = (u32)(d0 + d1) + ((u32)(d0 - d1) << 16), e1 = (u32)(d2 + d3) + ((u32)(d2 - d3) << 16);
    u32 c[2] = {e0 + e1, e0 - e1}, acc = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        u32 v = c[k], o = (u32)dpp_mov<DPP_XOR1>((int)v);
        v = (lane & 1) ? o - v : v + o;
        o = (u32)dpp_mov<DPP_XOR2>((int)v);
        v = (lane & 2) ? o - v : v + o;
        acc += lanes_abs(v);
    }
    acc = (u32)quad_sum4((int)acc);
    return (int)(((acc & 0xffffu) + (acc >> 16)) >> 1);
}
// unnormalised 8x8 Hadamard SATD of a block from one row per lane (rows in 8 consecutive lanes); pixel.c:256-289
__device__ __forceinline__ int sw_sa8d_rows(const u8 *f, const u8 *p, int lane)
{
    u32 e[4], t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int d0 = (int)f[2 * k] - (int)p[2 * k], d1 = (int)f[2 * k + 1] - (int)p[2 * k + 1];
        e[k] = (u32)(d0 + d1) + ((u32)(d0 - d1) << 16);
    }
    wht4(t[0], t[1], t[2], t[3], e[0], e[1], e[2], e[3]);
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 v = t[k], o = (u32)dpp_mov<DPP_XOR1>((int)v);
        v = (lane & 1) ? o - v : v + o;
        o = (u32)dpp_mov<DPP_XOR2>((int)v);
        v = (lane & 2) ? o - v : v + o;
        o = (u32)__shfl_xor((int)v, 4, 64);
        v = (lane & 4) ? o - v : v + o;
        acc += lanes_abs(v);
    }
    return half_sum8((int)((acc & 0xffffu) + (acc >> 16)));
}

// the 4x4 block's prediction table (intra_pred.h) from the pixels around dst (stride FD): 13 edge samples, their two- and
// three-tap filtered forms, the three DC values and 128
__device__ __forceinline__ void sw_pred4_table(SwLds &s, const u8 *dst, int lane)
{
    const int k = lane < 13 ? lane : 12;
    const int e = (int)(k < 4 ? dst[-1 + (3 - k) * FD] : k == 4 ? dst[-1 - FD] : dst[(k - 5) - FD]);
    int prev = dpp_mov<0x111>(e), next = dpp_mov<0x101>(e);      // row_shr:1 / row_shl:1: lanes k - 1 / k + 1
    if (lane == 0) prev = e;
    if (lane >= 12) next = e;
    if (lane < 13) { s.pt4[lane] = (u8)e; s.pt4[13 + lane] = (u8)((e + next + 1) >> 1); s.pt4[26 + lane] = (u8)((prev + 2 * e + next + 2) >> 2); }
    int dv = 0;
    if (lane >= 16 && lane < 20) dv = dst[(lane - 16) - FD];
    else if (lane >= 20 && lane < 24) dv = dst[-1 + (lane - 20) * FD];
    dv = quad_sum4(dv);
    const int t = __builtin_amdgcn_readlane(dv, 16), l = __builtin_amdgcn_readlane(dv, 20);
    if (lane >= 24 && lane < 28) s.pt4[39 + lane - 24] = (u8)(lane == 24 ? (t + l + 4) >> 3 : lane == 25 ? (l + 2) >> 2 : lane == 26 ? (t + 2) >> 2 : 128);
}
// x264_predict_8x8_filter with every filter on (R/common/predict.c:499-540): one lane per edge entry 7..32
__device__ __forceinline__ void sw_pred8_filter_all(u8 *edge, const u8 *src, int neigh, int lane)
{
#define PX(xx, yy) ((int)src[(xx) + (yy) * FD])
    const int have_tl = neigh & NB_TOPLEFT, have_tr = neigh & NB_TOPRIGHT;
    if (lane < 26) {
        const int i = 7 + lane;
        int av, bv, cv;
        if (i < 15) {                                    // left y = 14 - i
            const int y = 14 - i;
            av = y == 0 ? (have_tl ? PX(-1, -1) : PX(-1, 0)) : PX(-1, y - 1); bv = PX(-1, y); cv = y == 7 ? PX(-1, 7) : PX(-1, y + 1);
        } else if (i == 15) { av = PX(0, -1); bv = PX(-1, -1); cv = PX(-1, 0); }
        else if (i < 24) {                               // top x = i - 16
            const int x = i - 16;
            av = x == 0 ? (have_tl ? PX(-1, -1) : PX(0, -1)) : PX(x - 1, -1); bv = PX(x, -1); cv = x == 7 ? (have_tr ? PX(8, -1) : PX(7, -1)) : PX(x + 1, -1);
        } else if (have_tr) {                            // top right x = 8 .. 15, edge[32] = edge[31]
            const int x = i < 32 ? i - 16 : 15;
            av = PX(x - 1, -1); bv = PX(x, -1); cv = x == 15 ? PX(15, -1) : PX(x + 1, -1);
        } else av = bv = cv = PX(7, -1);
        edge[i] = (u8)((av + 2 * bv + cv + 2) >> 2);
    }
#undef PX
}
// the 8x8 block's prediction table from the filtered edge array (e[j] = edge[7 + j], j = 0..24)
__device__ __forceinline__ void sw_pred8_table(SwLds &s, int lane)
{
    if (lane < 25) {
        const int e = s.edge8[7 + lane], prev = lane == 0 ? e : (int)s.edge8[6 + lane], next = lane == 24 ? e : (int)s.edge8[8 + lane];
        s.pt8[lane] = (u8)e; s.pt8[25 + lane] = (u8)((e + next + 1) >> 1); s.pt8[50 + lane] = (u8)((prev + 2 * e + next + 2) >> 2);
    }
    int dv = 0;
    if (lane >= 32 && lane < 40) dv = s.edge8[16 + lane - 32];
    else if (lane >= 40 && lane < 48) dv = s.edge8[7 + lane - 40];
    dv = half_sum8(dv);
    const int t = __builtin_amdgcn_readlane(dv, 32), l = __builtin_amdgcn_readlane(dv, 40);
    if (lane >= 48 && lane < 52) s.pt8[75 + lane - 48] = (u8)(lane == 48 ? (l + t + 8) >> 4 : lane == 49 ? (l + 4) >> 3 : lane == 50 ? (t + 4) >> 3 : 128);
}
// unnormalised 8x8 Hadamard SATD from one row of differences per lane (rows in 8 consecutive lanes); pixel.c:256-289
__device__ __forceinline__ int sw_sa8d_rows_d(const int d[8], int lane)
{
    u32 e[4], t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = (u32)(d[2 * k] + d[2 * k + 1]) + ((u32)(d[2 * k] - d[2 * k + 1]) << 16);
    wht4(t[0], t[1], t[2], t[3], e[0], e[1], e[2], e[3]);
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 v = t[k], o = (u32)dpp_mov<DPP_XOR1>((int)v);
        v = (lane & 1) ? o - v : v + o;
        o = (u32)dpp_mov<DPP_XOR2>((int)v);
        v = (lane & 2) ? o - v : v + o;
        const u32 up = (u32)dpp_mov<0x104>((int)v), dn = (u32)dpp_mov<0x114>((int)v);     // row_shl:4 / row_shr:4: lanes + 4 / - 4 (both run in all lanes)
        o = (lane & 4) ? dn : up;                                                          // = lane ^ 4
        v = (lane & 4) ? o - v : v + o;
        acc += lanes_abs(v);
    }
    return half_sum8((int)((acc & 0xffffu) + (acc >> 16)));
}

// x264_mb_analyse_inter_p4x4_chroma (R/encoder/analyse.c:1373-1405): mc_chroma of the 8x8 block's sub-partitions into one 4x4 per
// plane, then mbcmp 4x4 against the source.  Lane = plane * 4 + row (lanes 8.. repeat the work); every pixel takes the vector
// of the sub-block that covers it (sub = 0 four 2x2, 1 two 4x2, 2 two 2x4), read from the lane-indexed records at rec0 + k.
__device__ __forceinline__ int sw_sub_chroma(const SwLds &s, const SwRefs &refs, const SwArgs &a, int r, int i8, int sub, int rec0, int sub_mx, int sub_my,
                                             int satd, ptrdiff_t oc, size_t bc, int lane)
{
    const int pl = (lane >> 2) & 1, y = lane & 3;
    const u8 *plane = (pl ? refs.v[r] : refs.u[r]) + bc + oc + 4 * (i8 & 1) + (ptrdiff_t)(4 * (i8 >> 1)) * a.sc;
    const u8 *src = s.fe + 256 + 64 * pl + (4 * (i8 >> 1) + y) * 8 + 4 * (i8 & 1);
    int d[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int k = sub == 0 ? (y >> 1) * 2 + (x >> 1) : sub == 1 ? (y >> 1) : (x >> 1);
        const int mvx = __shfl(sub_mx, rec0 + k, 64), mvy = __shfl(sub_my, rec0 + k, 64);
        const int dx = mvx & 7, dy = mvy & 7;
        const u8 *p = plane + (ptrdiff_t)(y + (mvy >> 3)) * a.sc + x + (mvx >> 3);
        const int v = ((8 - dx) * (8 - dy) * p[0] + dx * (8 - dy) * p[1] + (8 - dx) * dy * p[a.sc] + dx * dy * p[a.sc + 1] + 32) >> 6;
        d[x] = (int)src[x] - v;
    }
    const int c4 = sw_cost4x4_rows(d[0], d[1], d[2], d[3], satd, lane);
    return __builtin_amdgcn_readlane(c4, 0) + __builtin_amdgcn_readlane(c4, 4);
}

__device__ __forceinline__ int sw_load_acq(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

// WPE = waves per SIMD the register allocation is held to.  A row wave spends most of its time waiting
// on dependent LDS / L2 round trips, so throughput comes from other chains' waves filling those gaps:
// fewer registers per wave (some spilled) and more waves resident beats one fat wave per SIMD.
// LL: lossless, as a compile-time constant (its branches cost the usual path nothing)
// RD: the raster-order variant.  With the RD levels (the trial encodes are priced against the live CABAC contexts), trellis
// (same) or adaptive quantisation (a macroblock's QP follows from the previous one's, R/encoder/ratecontrol.c:263-264) a slice is
// one serial chain of macroblocks; one wavefront then owns a whole frame of one chain and walks it in raster order, rows and all,
// and the entropy coder (cabac_dev.h) runs inside the loop exactly where x264_slice_write has it.  Throughput comes from the
// number of frames in flight (grid = batch), not from a wavefront schedule inside the frame.
static __device__ const u8 d_lambda_tab[52] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
                                               6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};
static __device__ const int d_lambda2_tab[52] = {14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1161, 1462, 1843, 2322, 2925,
    3686, 4644, 5851, 7372, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628, 117964,
    148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436};
static __device__ const u8 d_chroma_qp[52] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                              29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

// RF: the I / P kernel with the RD refinement of subme 8-9 (x264_me_refine_qpel_rd, x264_intra_rd_refine: slice_refine.h)
// CH: the chain-table launch.  Chains that no longer move in lock step (adaptive B placement, per-chain QPs) each code their own
// frame type from their own pictures: block i takes ALL three argument structures from tab[i], built by the host exactly as for a
// launch of its own (x264hip_slice_sweep_chains), and codes batch element tab[i].a.chain.  The table is read through the constant
// address space, like the kernel-argument segment it replaces, so the same loads can be re-issued instead of kept in registers.
struct SwDesc { SwArgs a; SwRefs t; SwRd r; };
template <int WPE, bool LL = false, bool RD = false, bool BS = false, bool TD = false, bool RF = false, bool CH = false>       // TD: the extended B kernel (temporal direct prediction, lookahead candidates)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_slice_sweep(SwArgs a, SwRefs refs_k, SwRd rd_k, const SwDesc *tab)
{
    static_assert(!CH || RD, "the chain table belongs to the raster variant");
    SwRd rd_l;
    if constexpr (CH) {
        typedef const __attribute__((address_space(4))) SwDesc *desc_p;
        const desc_p d = (desc_p)(uintptr_t)tab + blockIdx.x;
        typedef const __attribute__((address_space(4))) u32 *word_p;
        static_assert(sizeof(SwArgs) % 4 == 0 && sizeof(SwRd) % 4 == 0 && alignof(SwDesc) >= 4, "copied by dwords");
        const word_p wa = (word_p)&d->a, wr = (word_p)&d->r;
#pragma unroll
        for (unsigned i = 0; i < sizeof(SwArgs) / 4; i++) ((u32 *)&a)[i] = wa[i];
#pragma unroll
        for (unsigned i = 0; i < sizeof(SwRd) / 4; i++) ((u32 *)&rd_l)[i] = wr[i];
    }
    const SwRd &rd = CH ? rd_l : rd_k;
    const SwRefs &refs = CH ? tab[blockIdx.x].t : refs_k;
    // A step's I / P chains and B chains run side by side, and an I / P chain is the longer of the two (more references, more candidates):
    // where a SIMD holds one of each, the I / P wavefront issues first, so that both kernels end at about the same time instead of
    // the B chains' wave slots idling while the step waits for its P chains.
    if constexpr (CH && !BS) __builtin_amdgcn_s_setprio(3);
    static_assert(!BS || RD, "B slices run in the raster variant");
    static_assert(!TD || BS, "temporal direct prediction is a B-slice matter");
    static_assert(!RF || (RD && !BS), "the RD refinement is built for the raster variant's I / P kernel");
#undef IS_SKIP_T
#define IS_SKIP_T(t) (BS ? ((t) == T_P_SKIP || (t) == T_B_SKIP) : (t) == T_P_SKIP)      /* BS is a template constant: the other kernels keep their single compare */
    __builtin_assume(a.lossless == (int)LL);           // the host launches the matching variant; do not write to `a` (a modified
                                                        // kernel argument is copied to scratch memory whole)
    __shared__ SwLds s;
    __shared__ typename std::conditional<BS, SwLdsRdB, typename std::conditional<RF, SwLdsRdF, typename std::conditional<RD, SwLdsRd, SwLdsNone>::type>::type>::type sr_;
    SwLdsRd &sr = *(SwLdsRd *)&sr_;                     // only touched when RD
    SwLdsB &sb = *(SwLdsB *)((char *)&sr_ + sizeof(SwLdsRd));    // only touched when BS (then sr_ is an SwLdsRdB)
    SwLdsRf &sf = *(SwLdsRf *)((char *)&sr_ + sizeof(SwLdsRd));  // only touched when RF (then sr_ is an SwLdsRdF)
    (void)sf;
    const int lane_id = threadIdx.x, lane = lane_id;
    const int bz = CH ? a.chain : RD ? (int)blockIdx.x : (int)(blockIdx.x % a.batch_pad), mby0 = RD ? 0 : (int)(blockIdx.x / a.batch_pad);
    if (bz >= a.batch) return;
    const size_t nmb = (size_t)a.mb_w * a.mb_h, cb = nmb * bz, by_ = a.bs_y * bz, bc_ = a.bs_c * bz;
    // batch element
    a.fy += by_; a.fu += bc_; a.fv += bc_; a.dy += by_; a.du += bc_; a.dv += bc_;
    // only what every macroblock reads is adjusted here; the arrays that are written once per macroblock are addressed as base + cb
    // at the store (a base straight from the kernel arguments can be re-loaded; an adjusted one occupies two SGPRs for the whole body)
    a.mb_type += nmb * bz; a.ref += 4 * nmb * bz; a.i4mode += 16 * nmb * bz;
    a.mv += 32 * nmb * bz; a.mvr += 2 * SW_MAX_REFS * nmb * bz;
    if (a.l0_type) { a.l0_type += nmb * bz; a.l0_ref += 4 * nmb * bz; a.l0_mv += 32 * nmb * bz; }
    int *prog = a.progress + (size_t)bz * a.mb_h;

    const int satd = a.subme > 1 && !a.lossless, is_p = a.slice_type == 0;
    const MeOpts mo = {a.me_method, a.me_range, a.subme, a.chroma_me, a.lossless};
    SwQp Q = {a.qp, a.qpc, a.lambda, d_lambda2_tab[a.qp], a.chroma_skip_thresh};
    const i16 *cost_g = a.cost_mv + a.cost_center;     // p_cost_mv of the current QP, centred
    // tables of the current QP: into LDS (once per slice; again whenever adaptive quantisation changes the macroblock's QP)
    auto load_qp_tables = [&](int lane) {
        const int cat = lane >> 4, i = lane & 15, q = cat < 2 ? Q.qp : Q.qpc;
        s.qmf[cat][i] = a.q4mf[(cat * 52 + q) * 16 + i]; s.qbias[cat][i] = a.q4bias[(cat * 52 + q) * 16 + i];
        s.qdq[cat][i] = a.dq4[cat * 96 + (q % 6) * 16 + i];
        if (is_p || BS)
            for (int k = lane; k < 2 * MX_COST_LDS + 1; k += 64) s.costl[k] = cost_g[k - MX_COST_LDS];
        if (a.transform8x8)
            for (int c8 = 0; c8 < 2; c8++) {
                s.q8mf[c8][lane] = a.q8mf[(c8 * 52 + Q.qp) * 64 + lane]; s.q8bias[c8][lane] = a.q8bias[(c8 * 52 + Q.qp) * 64 + lane];
                s.q8dq[c8][lane] = a.dq8[c8 * 384 + (Q.qp % 6) * 64 + lane];
            }
        if constexpr (RD) {
            if (rd.trellis) {
                sr.unq4[cat][i] = rd.unq4[(cat * 52 + q) * 16 + i];
                if (a.transform8x8) for (int c8 = 0; c8 < 2; c8++) sr.unq8[c8][lane] = rd.unq8[(c8 * 52 + Q.qp) * 64 + lane];
            }
        }
    };
    load_qp_tables(lane);
    {
        if (a.nr) { s.nr_off8[lane] = a.nr_offset[(size_t)bz * 128 + 64 + lane]; if (lane < 16) s.nr_off4[lane] = a.nr_offset[(size_t)bz * 128 + lane]; }
        if (lane < 48) s.p4lut[lane] = ((const u32 *)&c_plut4)[lane];
        for (int k = lane; k < 192; k += 64) s.p8lut[k] = ((const u32 *)&c_plut8)[k];
    }
    // the entropy coder of this chain's slice (x264_slice_write, R/encoder/encoder.c:1155-1165)
    DCabac cab = {0, 0x1FE, -1, 0, nullptr, 0};
    // h->mb.cache.ref / mv [list][x264_scan8[12]] as the previous macroblock (or frame) left it: x264_macroblock_cache_load never rewrites
    // the cache's inner entries (SwRd::stale; only a B macroblock whose temporal direct prediction fails ever looks at it)
    int st0r = 0, st0x = 0, st0y = 0, st1r = 0, st1x = 0, st1y = 0;
    if constexpr (RD) {
        if (rd.stale) {
            const i16 *sp = rd.stale + (size_t)bz * 8;
            st0r = __builtin_amdgcn_readfirstlane(sp[0]); st0x = __builtin_amdgcn_readfirstlane(sp[1]); st0y = __builtin_amdgcn_readfirstlane(sp[2]);
            st1r = __builtin_amdgcn_readfirstlane(sp[3]); st1x = __builtin_amdgcn_readfirstlane(sp[4]); st1y = __builtin_amdgcn_readfirstlane(sp[5]);
        }
        if constexpr (TD) {         // the B flow keeps it in LDS (slice_b_flow.h)
            if (lane == 30) { sb.stale[0] = (i16)st0r; sb.stale[1] = (i16)st0x; sb.stale[2] = (i16)st0y; sb.stale[3] = (i16)st1r; sb.stale[4] = (i16)st1x; sb.stale[5] = (i16)st1y; }
        }
    }
    u8 *payload0 = nullptr;
    int last_qp = a.qp, last_dqp = 0, prev_coded = 0, intra_before = 0;      // h->mb.i_last_qp / i_last_dqp; the previous macroblock "has coefficients"
    int dscore0 = 0, dscore1 = 0;                                           // h->stat.frame.i_direct_score[temporal / spatial] (--direct auto)
    (void)dscore0; (void)dscore1;
    if constexpr (RD) {
        if (rd.write) {
            payload0 = rd.payload + (size_t)bz * rd.payload_cap + 64;
            cab.p = payload0;
            for (int k = lane; k < 460; k += 64) sr.cabac[k] = (u8)cd_context_init_one(k, a.slice_type, a.qp, rd.cabac_init_idc);
        }
        if (lane < 16) { sr.zero16[lane] = 0; sr.zz4[lane] = d_zz4[lane]; sr.w4z[lane] = d_w4z[lane]; }
        if (lane < 4) sr.zz2[lane] = (u8)lane;
        sr.zz8[lane] = c_scan8[0][lane]; sr.w8z[lane] = sw_w8z(lane);
        cd_load_tables(lane);
    }
    WAVE_SYNC();

    int nr_acc4 = 0, nr_acc8 = 0, nr_n4 = 0, nr_n8 = 0;      // --nr: this row's additions to nr_residual_sum (lane = coefficient index) / nr_count
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ptime = a.prof ? (long long)wall_clock64() : 0;
#define PROF(k_) do { if (a.prof) { long long now_ = (long long)wall_clock64(); pacc[k_] += now_ - ptime; ptime = now_; } } while (0)
  for (int mby = mby0; mby < (RD ? a.mb_h : mby0 + 1); mby++) {
    if constexpr (RD) {
        // this wave's own stores of the row above (pixels, types, vectors ...) must be what its loads see
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // the left neighbour = this wave's previous macroblock
    int left_type = -1, left_ref = -2, left_mvx = 0, left_mvy = 0, row_intra = 0;
    int left_cbp = -1, left_cpm = 0, left_t8 = 0;          // (RD) h->mb.cbp / chroma_pred_mode / mb_transform_size of the left macroblock
    u32 pre_y;
    u8 pre_u, pre_v;
    {
        const ptrdiff_t oy0 = (ptrdiff_t)16 * mby * a.sy, oc0 = (ptrdiff_t)8 * mby * a.sc;
        pre_y = *(const u32 *)(a.fy + oy0 + (ptrdiff_t)(lane >> 2) * a.sy + (lane & 3) * 4);
        pre_u = a.fu[oc0 + (ptrdiff_t)(lane >> 3) * a.sc + (lane & 7)];
        pre_v = a.fv[oc0 + (ptrdiff_t)(lane >> 3) * a.sc + (lane & 7)];
    }

    for (int mbx = 0; mbx < a.mb_w; mbx++) {
        // The lane id is laundered once per macroblock: otherwise every lane-derived address and index of the body is hoisted
        // out of this loop, and, being live across all of it, spilled to scratch at the top and reloaded at its use (measured:
        // ~160 scratch stores per macroblock).  Recomputing them from the lane id costs a few VALU operations each.
        int lane = lane_id;
#define LAUNDER() asm volatile("" : "+v"(lane))
        LAUNDER();
        const int mb = mby * a.mb_w + mbx;
        // ---- wait for the row above: left-top, top and top-right neighbours finished ----
        if (!RD && mby > 0) {
            const int need = min(mbx + 2, a.mb_w);
            int spins = 0;
            // Poll with relaxed loads: an acquire load invalidates this CU's vector L1 on every poll, for every wave
            // resident on it (measured: +20 % frames/s).  One acquire fence once the count has been seen.
            for (;;) {
                int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog + mby - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((v & 0xffff) >= need) break;
                if (spins < 4) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(100);
                int ab = (spins & 15) == 15 ? __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0;
                if (ab || ++spins > a.spin_limit) {
                    if (lane == 0) { __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (!ab) atomicAdd(a.abort_total, 1); }
                    return;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        PROF(0);
        const ptrdiff_t oy = (ptrdiff_t)16 * mby * a.sy + 16 * mbx, oc = (ptrdiff_t)8 * mby * a.sc + 8 * mbx;
        // ---- x264_macroblock_cache_load: pixels ----
        if (mbx > 0) {      // copy_column8: the column to the left is the previous reconstruction's last column
            if (lane < 16) s.fd[FDY + lane * FD - 1] = s.fd[FDY + lane * FD + 15];
            else if (lane < 24) s.fd[FDU + (lane - 16) * FD - 1] = s.fd[FDU + (lane - 16) * FD + 7];
            else if (lane < 32) s.fd[FDV + (lane - 24) * FD - 1] = s.fd[FDV + (lane - 24) * FD + 7];
        }
        {   // source pixels: fetched one macroblock ahead (they depend on nothing), parked in registers meanwhile
            const int r = lane >> 2, x = (lane & 3) * 4;
            *(u32 *)(s.fe + r * 16 + x) = pre_y;
            s.fe[256 + lane] = pre_u;
            s.fe[320 + lane] = pre_v;
        }
        if (mby > 0) {      // the row above, still unfiltered: x = -1 .. w*3/2-1
            if (lane < 25) s.fd[FDY - FD - 1 + lane] = a.dy[oy - a.sy - 1 + lane];
            else if (lane >= 32 && lane < 45) s.fd[FDU - FD - 1 + (lane - 32)] = a.du[oc - a.sc - 1 + (lane - 32)];
            else if (lane >= 48 && lane < 61) s.fd[FDV - FD - 1 + (lane - 48)] = a.dv[oc - a.sc - 1 + (lane - 48)];
        }
        if (mbx + 1 < a.mb_w) {     // issued after the loads above so that waiting for those leaves these in flight
            const int r = lane >> 2, x = (lane & 3) * 4, cx = lane & 7, cy = lane >> 3;
            pre_y = *(const u32 *)(a.fy + oy + 16 + (ptrdiff_t)r * a.sy + x);
            pre_u = a.fu[oc + 8 + (ptrdiff_t)cy * a.sc + cx];
            pre_v = a.fv[oc + 8 + (ptrdiff_t)cy * a.sc + cx];
        }
        WAVE_SYNC();
        PROF(1);
        LAUNDER();
        // ---- neighbour availability and types ----
        int nb = 0, type_top = -1, type_topleft = -1, type_topright = -1;
#define UNI(x_) __builtin_amdgcn_readfirstlane((int)(x_))      /* a wave-uniform load: keep the value in a scalar register */
        if (mby > 0) { nb |= NB_TOP; type_top = UNI(a.mb_type[mb - a.mb_w]); }
        if (mbx > 0) nb |= NB_LEFT;
        if (mbx < a.mb_w - 1 && mby > 0) { nb |= NB_TOPRIGHT; type_topright = UNI(a.mb_type[mb - a.mb_w + 1]); }
        if (mbx > 0 && mby > 0) { nb |= NB_TOPLEFT; type_topleft = UNI(a.mb_type[mb - a.mb_w - 1]); }
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

        int type = T_I_16x16, mvx = 0, mvy = 0, ref = 0, skip_mc = 0, pred16 = 0, predc = 0, part = 16;
        int sub_t_mb = 3;                    // lanes 0..3: h->mb.i_sub_partition[] (D_L0_4x4 0, 8x4 1, 4x8 2, 8x8 3)
        int satd_i16 = MX_COST_MAX, satd_chroma = MX_COST_MAX, pskx = 0, psky = 0;
        int satd_i8 = MX_COST_MAX, satd_i4 = MX_COST_MAX, i8_cbp = 0, i4_cbp = 0, t8 = 0, fi_open = 0, stat_alt = -1;
        if (a.flags_intra & 3) {
            // intra4x4_pred_mode cache (R/common/macroblock.c:907-980): -1 where there is no neighbour; the frame array holds
            // I_PRED_4x4_DC for every macroblock that is not I_4x4 / I_8x8
            if (lane < 48) s.i4c[lane] = -1;
            WAVE_SYNC();
            if ((nb & NB_TOP) && lane < 4)
                s.i4c[4 + lane] = a.i4mode[(size_t)(mb - a.mb_w) * 16 + (lane == 0 ? 10 : lane == 1 ? 11 : lane == 2 ? 14 : 15)];
            if ((nb & NB_LEFT) && lane >= 8 && lane < 12) s.i4c[11 + 8 * (lane - 8)] = s.left_i4[lane - 8];
            WAVE_SYNC();
        }
        int stat_intra = 0, stat_inter = 0, analysed = 0;
        // x264_mb_analyse_init (R/encoder/analyse.c:235-252): h->mb.b_trellis while analysing, i_skip_intra
        const int mbrd = RD ? rd.mbrd : 0;
        SwTq tq = {RD && rd.trellis > 1 && mbrd, &sr};
        int skip_intra = a.lossless ? 0 : mbrd ? 2 : (RD ? (!rd.trellis && !a.nr) : 1);
        (void)skip_intra;

        // x264_mb_analyse_intra_chroma, R/encoder/analyse.c:539-610
        auto analyse_chroma = [&]() {
            if (satd_chroma < MX_COST_MAX) return;
            int n;
            const u32 list = sw_modes8c(nb, n);
            for (int i = 0; i < n; i++) {
                const int m = (int)((list >> (4 * i)) & 15);
                sw_pred8c(s, m, lane, a.lossless);
                int c = sw_cmp_chroma(s, satd, lane) + Q.lambda * sw_ue_size(sw_fix8c(m));
                if constexpr (RF) { if (lane == 0) sf.cdir[i] = c; }
                if (c < satd_chroma) { satd_chroma = c; predc = m; }
            }
        };
        // a->b_fast_intra (R/encoder/analyse.c:345-362), evaluated only when its value matters.  Its last term counts
        // the intra macroblocks BEFORE this one in raster order, some of which (to the right in the rows above) may
        // not be coded yet: bound the count from what the rows above have published, and wait only while the
        // bounds leave the answer open (the rows above never wait for this one, so this terminates).
        // wait = 0: answer 0 / 1, or 2 when the bounds do not decide it yet; wait = 1: poll until they do
        auto fast_intra_now = [&](int wait) -> int {
            if ((!is_p && !BS) || mb <= 4) return 0;
            if (IS_INTRA_T(left_type) || IS_INTRA_T(type_top) || IS_INTRA_T(type_topleft) || IS_INTRA_T(type_topright)) return 0;
            if ((!BS || is_p) && a.l0_type && IS_INTRA_T(UNI(a.l0_type[mb]))) return 0;      // only in a P slice (analyse.c:357)
            if constexpr (RD) return mb < 3 * intra_before ? 0 : 1;        // raster order: every earlier macroblock is done
            for (int spins = 0;; spins++) {
                int known = row_intra, pending = 0;
                for (int r0 = 0; r0 < mby; r0 += 64) {
                    const int r = r0 + lane;
                    const int v = r < mby ? __hip_atomic_load(prog + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    known += wave_sum(r < mby ? v >> 16 : 0);
                    pending += wave_sum(r < mby ? a.mb_w - (v & 0xffff) : 0);
                }
                if (mb < 3 * known) return 0;
                if (mb >= 3 * (known + pending)) return 1;
                if (!wait) return 2;
                __builtin_amdgcn_s_sleep(100);
                if (spins > a.spin_limit) { if (lane == 0) { __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(a.abort_total, 1); } return 0; }
            }
        };
        // x264_mb_analyse_intra, R/encoder/analyse.c:612-843
        auto analyse_intra = [&](int satd_inter) {
            LAUNDER();
            {
                int n;
                const u32 list = sw_modes16(nb, n);
                for (int i = 0; i < n; i++) {
                    const int m = (int)((list >> (4 * i)) & 15);
                    sw_pred16(s, m, lane, a.lossless);
                    int c = sw_cmp_luma16(s, satd, lane) + Q.lambda * sw_ue_size(sw_fix16(m));
                    if constexpr (RF) { if (lane == 0) sf.i16dir[m] = c; }
                    if (c < satd_i16) { satd_i16 = c; pred16 = m; }
                }
            }
            if constexpr (BS) satd_i16 += Q.lambda * 9;                  // i_mb_b_cost_table[I_16x16], analyse.c:659-661
            if (!(a.flags_intra & 3)) return;
            if (satd_i16 > 2 * satd_inter) {
                // b_fast_intra would end the analysis here.  If the raster-order count behind it is not decidable yet, go on
                // as if it were 0: the extra analysis only matters if 8x8 / 4x4 then beat the inter cost, which the caller
                // checks (and only then waits for the exact answer); the statistics term is settled after the frame.
                const int fi = fast_intra_now(0);
                if (fi == 1) return;
                fi_open = fi == 2;
            }
            if (a.flags_intra & 2) {                                   // X264_ANALYSE_I8x8
                const int thresh = mbrd ? MX_COST_MAX : min(satd_inter, satd_i16);
                int cost = BS ? Q.lambda * 9 : 0, idx, acbp = 0;            // i_mb_b_cost_table[I_8x8], :676-677
                for (idx = 0;; idx++) {
                    const int bx = 8 * (idx & 1), by = 8 * (idx >> 1), pm = sw_pred_i4mode(s, 4 * idx), nb8 = sw_nb8(idx, nb);
                    int n;
                    const unsigned long long list = sw_modes4(nb8, n);
                    sw_pred8_filter_all(s.edge8, s.fd + FDY + by * FD + bx, nb8, lane);
                    WAVE_SYNC();
                    sw_pred8_table(s, lane);
                    WAVE_SYNC();
                    u32 kb = 0xffffffffu;
#pragma unroll
                    for (int pass = 0; pass < 2; pass++) {
                        const int g = (lane >> 3) + 8 * pass, r = lane & 7;
                        u32 key = 0xffffffffu;
                        if (g < n) {
                            const int mode = (int)((list >> (4 * g)) & 15);
                            const u32 o0 = s.p8lut[(mode * 8 + r) * 2], o1 = s.p8lut[(mode * 8 + r) * 2 + 1];
                            const u32 f0 = *(const u32 *)(s.fe + (by + r) * 16 + bx), f1 = *(const u32 *)(s.fe + (by + r) * 16 + bx + 4);
                            int d[8];
#pragma unroll
                            for (int x = 0; x < 4; x++) {
                                d[x] = (int)((f0 >> (8 * x)) & 255) - (int)s.pt8[(o0 >> (8 * x)) & 255];
                                d[4 + x] = (int)((f1 >> (8 * x)) & 255) - (int)s.pt8[(o1 >> (8 * x)) & 255];
                            }
                            if (a.lossless && mode < 2) {
#pragma unroll
                                for (int x = 0; x < 8; x++) d[x] = (int)s.fe[(by + r) * 16 + bx + x] - sw_ll_px(s, 0, mode, bx + x, by + r);
                            }
                            int c;
                            if (satd) c = (sw_sa8d_rows_d(d, lane) + 2) >> 2;
                            else {
                                int sd = 0;
#pragma unroll
                                for (int x = 0; x < 8; x++) sd += iabs(d[x]);
                                c = half_sum8(sd);
                            }
                            key = ((u32)(c + Q.lambda * (pm == sw_fix4(mode) ? 1 : 4)) << 4) | (u32)g;
                            if constexpr (RF) { if (r == 0) sf.i8dir[mode][idx] = (int)(key >> 4); }
                        }
                        // the reference's in-order strict '<' over the modes = the smallest (cost, slot) key
#pragma unroll
                        for (int k = 0; k < 8; k++) { const u32 t = (u32)__builtin_amdgcn_readlane((int)key, 8 * k); kb = t < kb ? t : kb; }
                    }
                    const int best = (int)(kb >> 4), bmode = (int)((list >> (4 * (kb & 15))) & 15);
                    cost += best;
                    if (lane == 0) s.pred8[idx] = (signed char)bmode;
                    if (idx == 3 || cost > thresh) break;
                    {
                        int v = s.pt8[(s.p8lut[(bmode * 8 + (lane >> 3)) * 2 + ((lane >> 2) & 1)] >> (8 * (lane & 3))) & 255];
                        if (a.lossless && bmode < 2) v = sw_ll_px(s, 0, bmode, bx + (lane & 7), by + (lane >> 3));
                        WAVE_SYNC();
                        s.fd[FDY + (by + (lane >> 3)) * FD + bx + (lane & 7)] = (u8)v;
                        if (lane < 4) s.i4c[sw_scan8(4 * idx) + (lane & 1) + 8 * (lane >> 1)] = (signed char)bmode;
                        WAVE_SYNC();
                    }
                    sw_encode_i8x8(s, a, Q, tq, idx, acbp, lane);
                }
                if (idx == 3) {
                    satd_i8 = cost; i8_cbp = acbp;
                    if constexpr (RD) { if (skip_intra == 2) for (int k = lane; k < 256; k += 64) sr.i8_dct[k] = s.lv_y8[k]; }
                    *(u32 *)(s.i8_fdec + lane * 4) = *(const u32 *)(s.fd + FDY + (lane >> 2) * FD + (lane & 3) * 4);
                    if (lane < 16) s.i8_nnz[lane] = s.nnz[lane];
                    WAVE_SYNC();
                } else {
                    satd_i8 = MX_COST_MAX;
                    cost = (cost * (idx == 0 ? 1024 : idx == 1 ? 512 : 341)) >> 8;
                }
                if (min(cost, satd_i16) > satd_inter * (5 + !!mbrd) / 4) return;
            }
            if (a.flags_intra & 1) {                                   // X264_ANALYSE_I4x4
                int thresh = min(min(satd_inter, satd_i16), satd_i8);
                if (mbrd) thresh = thresh * (10 - fast_intra_now(0)) / 8;
                int cost = Q.lambda * (BS ? 24 + 9 : 24), idx, acbp = 0;    // + i_mb_b_cost_table[I_4x4] in a B slice, :770-771
                for (idx = 0;; idx++) {
                    int bx, by, n;
                    sw_blk_xy(idx, bx, by);
                    const int pm = sw_pred_i4mode(s, idx), nb4 = sw_nb4(idx, nb);
                    const unsigned long long list = sw_modes4(nb4, n);
                    u8 *dst = s.fd + FDY + by * FD + bx;
                    if ((nb4 & (NB_TOPRIGHT | NB_TOP)) == NB_TOP && lane < 4) dst[4 - FD + lane] = dst[3 - FD];    // emulate missing topright samples
                    WAVE_SYNC();
                    sw_pred4_table(s, dst, lane);
                    WAVE_SYNC();
                    u32 key = 0xffffffffu;
                    {
                        const int g = lane >> 2, r = lane & 3;
                        if (g < n) {
                            const int mode = (int)((list >> (4 * g)) & 15);
                            const u32 off = s.p4lut[mode * 4 + r], fw = *(const u32 *)(s.fe + (by + r) * 16 + bx);
                            int d0 = (int)(fw & 255) - (int)s.pt4[off & 255], d1 = (int)((fw >> 8) & 255) - (int)s.pt4[(off >> 8) & 255];
                            int d2 = (int)((fw >> 16) & 255) - (int)s.pt4[(off >> 16) & 255], d3 = (int)(fw >> 24) - (int)s.pt4[off >> 24];
                            if (a.lossless && mode < 2) {
                                d0 = (int)(fw & 255) - sw_ll_px(s, 0, mode, bx, by + r); d1 = (int)((fw >> 8) & 255) - sw_ll_px(s, 0, mode, bx + 1, by + r);
                                d2 = (int)((fw >> 16) & 255) - sw_ll_px(s, 0, mode, bx + 2, by + r); d3 = (int)(fw >> 24) - sw_ll_px(s, 0, mode, bx + 3, by + r);
                            }
                            const int c = sw_cost4x4_rows(d0, d1, d2, d3, satd, lane);
                            key = ((u32)(c + Q.lambda * (pm == sw_fix4(mode) ? 1 : 4)) << 4) | (u32)g;
                        }
                    }
                    u32 kb = (u32)__builtin_amdgcn_readlane((int)key, 0);
#pragma unroll
                    for (int k = 1; k < 9; k++) { const u32 t = (u32)__builtin_amdgcn_readlane((int)key, 4 * k); kb = t < kb ? t : kb; }
                    const int best = (int)(kb >> 4), bmode = (int)((list >> (4 * (kb & 15))) & 15);
                    cost += best;
                    if (lane == 0) s.pred4[idx] = (signed char)bmode;
                    if (cost > thresh || idx == 15) break;
                    if (lane < 16) dst[(lane >> 2) * FD + (lane & 3)] = a.lossless && bmode < 2 ? (u8)sw_ll_px(s, 0, bmode, bx + (lane & 3), by + (lane >> 2))
                                                                         : s.pt4[(s.p4lut[bmode * 4 + (lane >> 2)] >> (8 * (lane & 3))) & 255];
                    if (lane == 0) s.i4c[sw_scan8(idx)] = (signed char)bmode;
                    WAVE_SYNC();
                    sw_encode_i4x4(s, a, Q, tq, idx, acbp, lane);
                }
                if (idx == 15) {
                    satd_i4 = cost; i4_cbp = acbp;
                    if constexpr (RD) { if (skip_intra == 2) for (int k = lane; k < 256; k += 64) sr.i4_dct[k] = s.lv_y[k]; }
                    *(u32 *)(s.i4_fdec + lane * 4) = *(const u32 *)(s.fd + FDY + (lane >> 2) * FD + (lane & 3) * 4);
                    if (lane < 16) s.i4_nnz[lane] = s.nnz[lane];
                    WAVE_SYNC();
                } else
                    satd_i4 = MX_COST_MAX;
            }
        };

        // ---- x264_macroblock_encode (R/encoder/macroblock.c:475-790) of the macroblock as type / part / t8 / the intra modes / s.mv4 /
        // s.ref8 describe it now.  The final encode, and with the RD levels every trial encode of x264_rd_cost_mb (final_pass = 0).
        int cbp_luma = 0, cbp_chroma = 0;
        bool encoded = false;               // (RD) the final encode has run inside the candidate loop
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
            WAVE_SYNC();
            if (BS && type == T_B_SKIP) return;              // x264_macroblock_encode_skip: the prediction (made by the caller) is the reconstruction
            if (type == T_I_16x16) {
                t8 = 0;
                analyse_chroma();
                sw_pred16(s, pred16, lane, a.lossless);
                cbp_luma = sw_encode_i16x16(s, a, Q, tq, lane, BS);
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
                if constexpr (!BS) sw_mc_parts(s, refs, a, oy, oc, by_, bc_, lane, RF, mbx, mby);    // (B slice: the caller has run the bi-predictive motion compensation)
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
                if (BS && type == T_B_DIRECT && !(cbp_luma | cbp_chroma)) type = T_B_SKIP;       // macroblock.c:784-788
            }
        };

        // ---- the RD levels: x264_mb_cache_fenc_satd, ssd_mb, x264_macroblock_size_cabac, x264_rd_cost_mb ----
        int fenc_satd_sum = 0, fenc_sa8d_sum = 0;
        auto cache_fenc_satd = [&]() {     // R/encoder/analyse.c:509-537 (the 16x16 sums; sub-partition RD is not built)
            if (!rd.psy_rd) return;
            int v4 = 0, v8 = 0;
            if (lane < 16) {
                const u8 *fe = s.fe + (lane >> 2) * 64 + (lane & 3) * 4;
                int sad = 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) sad += fe[j * 16 + i];
                v4 = satd_4x4(sr.zero16, 0, fe, 16) - (sad >> 1);
            } else if (lane < 20) {
                const int b = lane - 16;
                const u8 *fe = s.fe + (b >> 1) * 128 + (b & 1) * 8;
                int sad = 0;
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int i = 0; i < 8; i++) sad += fe[j * 16 + i];
                v8 = ((sa8d_8x8_raw(sr.zero16, 0, fe, 16) + 2) >> 2) - (sad >> 2);
            }
            fenc_satd_sum = wave_sum(v4); fenc_sa8d_sum = wave_sum(v8);
            if constexpr (RF) {         // h->mb.pic.fenc_satd[y][x] / fenc_sa8d[y][x]: the partial RD costs sum them over their blocks (sum_satd / sum_sa8d, rdo.c:66-91)
                if (lane < 16) sr.fenc_satd[lane] = v4; else if (lane < 20) sr.fenc_sa8d[lane - 16] = v8;
                WAVE_SYNC();
            }
        };
        auto ssd_mb = [&]() -> int {       // ssd_mb / ssd_plane, R/encoder/rdo.c:106-137
            int acc = 0;
            {
                const int r = lane >> 2, x = (lane & 3) * 4, cx = lane & 7, cy = lane >> 3;
#pragma unroll
                for (int i = 0; i < 4; i++) { const int d = (int)s.fe[r * 16 + x + i] - (int)s.fd[FDY + r * FD + x + i]; acc += d * d; }
                const int du = (int)s.fe[256 + cy * 8 + cx] - (int)s.fd[FDU + cy * FD + cx], dv = (int)s.fe[320 + cy * 8 + cx] - (int)s.fd[FDV + cy * FD + cx];
                acc += du * du + dv * dv;
            }
            int ssd = wave_sum(acc);
            if (rd.psy_rd) {
                unsigned long long h = 0;
                if (lane < 4) h = hadamard_ac_8x8(s.fd + FDY + (lane >> 1) * 8 * FD + (lane & 1) * 8, FD);
                const u32 lo = (u32)h, hi = (u32)(h >> 32);
                unsigned long long sum = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) sum += ((unsigned long long)(u32)__builtin_amdgcn_readlane((int)hi, k) << 32) + (u32)__builtin_amdgcn_readlane((int)lo, k);
                const int s4 = (int)((u32)sum >> 1), s8 = (int)(sum >> 34);
                const int satd = (iabs(s4 - fenc_satd_sum) + iabs(s8 - fenc_sa8d_sum)) >> 1;
                ssd += (satd * rd.psy_rd * Q.lambda + 128) >> 8;
            }
            return ssd;
        };
        // what the entropy coder reads of this macroblock: the interior of the motion cache from s.mv4 / s.ref8 (all lanes) ...
        auto syn_prepare = [&]() {
            if (is_p && lane < 16) {
                const int k = 12 + (lane & 3) + 8 * (lane >> 2);
                sr.cref[k] = s.ref8[(lane >> 3) * 2 + ((lane & 3) >> 1)]; sr.cmv[k][0] = s.mv4[lane][0]; sr.cmv[k][1] = s.mv4[lane][1];
            }
            if (lane < 4) sr.sub[lane] = (signed char)sub_t_mb;
            WAVE_SYNC();
        };
        // ... and the record the writer walks (scalars: wave-uniform registers)
        auto make_syn = [&]() -> MbSynDev {
            MbSynDev y;
            y.slice_type = a.slice_type; y.type = type; y.partition = part; y.i16mode = pred16; y.chroma_mode = predc;
            y.cbp_luma = cbp_luma; y.cbp_chroma = cbp_chroma; y.t8 = t8; y.qp = Q.qp; y.n_ref = a.n_refs; y.pps_t8 = a.transform8x8;
            y.t8_allowed = a.transform8x8 && (type == T_P_L0 || (type == T_P_8x8 && __ballot(lane < 4 && sub_t_mb != 3) == 0));
            if constexpr (BS) y.t8_allowed = a.transform8x8 && type >= T_B_DIRECT && type <= T_B_8x8;
            // h->mb.type[] holds x264_mb_type_fix'ed types (I_8x8 is stored as I_4x4, R/common/macroblock.c:1209,1226)
            y.type_left = left_type == T_I_8x8 ? T_I_4x4 : left_type; y.type_top = type_top == T_I_8x8 ? T_I_4x4 : type_top; y.cbp_left = left_cbp; y.cbp_top = cbp_top; y.cpm_left = left_cpm; y.cpm_top = cpm_top;
            y.nb_t8 = (left_type >= 0 && left_t8) + (type_top >= 0 && t8_top);
            y.last_qp = last_qp; y.last_dqp = last_dqp; y.prev_coded = prev_coded;
            y.sub = sr.sub; y.i4c = s.i4c; y.cref = sr.cref; y.cmv = sr.cmv; y.cmvd = sr.cmvd;
            y.n_ref1 = 0; y.cref1 = nullptr; y.cskip = nullptr; y.cmv1 = nullptr; y.cmvd1 = nullptr;
            if constexpr (BS) { y.n_ref1 = 1; y.sub = sb.sub; y.cref1 = sb.cref1; y.cskip = sb.cskip; y.cmv1 = sb.cmv1; y.cmvd1 = sb.cmvd1; }
            y.nnz = s.nnz; y.nz_l = sr.nz_l; y.nz_t = sr.nz_t; y.nz_lc = sr.nz_lc; y.nz_tc = sr.nz_tc;
            y.lv4 = (i16 (*)[16])s.lv_y; y.lv8 = (i16 (*)[64])s.lv_y8; y.lv_dc = s.lv_dc; y.lv_cdc = (i16 (*)[4])s.lv_cdc; y.lv_cac = (i16 (*)[16])s.lv_cac;
            return y;
        };
        (void)cache_fenc_satd; (void)ssd_mb;
        // a->i_satd_pcm, analyse.c:246
        const int satd_pcm = RD && !rd.psy_rd && mbrd ? (int)(((unsigned long long)(386 * 8) * (u32)Q.lambda2 + 128) >> 8) : MX_COST_MAX;

        if constexpr (BS) {
#include "slice_b_flow.h"
        } else
        if (!RD && !is_p) {
          {
            analyse_intra(MX_COST_MAX);
            type = T_I_16x16;
            int i_cost = satd_i16;
            if (satd_i4 < i_cost) { i_cost = satd_i4; type = T_I_4x4; }
            if (satd_i8 < i_cost) { i_cost = satd_i8; type = T_I_8x8; }
          }
        } else {
            // (The raster variant sends an I slice's macroblocks down this path too, its motion parts skipped: the candidate loop at the
            // end -- and with it the encoder, the distortion and the bit counter -- then exists ONCE in the kernel.  Two call sites of
            // the encoder made the compiler keep it as a function, and every variable it shares with the rest in scratch memory.)
            // ---- motion neighbours: what cache_load puts around the block (R/common/macroblock.c:1040-1128) ----
            int ra = left_ref, ax = left_mvx, ay = left_mvy;                 // A
            int rb = -2, bx = 0, byv = 0, rc = -2, cx = 0, cy = 0;            // B, C (or D)
            if (is_p && (nb & NB_TOP)) { const int o = mb - a.mb_w; rb = UNI(a.ref[o * 4 + 2]); bx = UNI(a.mv[(o * 16 + 12) * 2]); byv = UNI(a.mv[(o * 16 + 12) * 2 + 1]); }
            if (!is_p) {}
            else if (nb & NB_TOPRIGHT) { const int o = mb - a.mb_w + 1; rc = UNI(a.ref[o * 4 + 2]); cx = UNI(a.mv[(o * 16 + 12) * 2]); cy = UNI(a.mv[(o * 16 + 12) * 2 + 1]); }
            else if (nb & NB_TOPLEFT) { const int o = mb - a.mb_w - 1; rc = UNI(a.ref[o * 4 + 3]); cx = UNI(a.mv[(o * 16 + 15) * 2]); cy = UNI(a.mv[(o * 16 + 15) * 2 + 1]); }
            // The motion cache (h->mb.cache.ref[0] / mv[0], x264_scan8 layout) and the partition analysis' candidate records live in
            // the register file as lane-indexed arrays: entry k = lane k of a VGPR, read with v_readlane (uniform index), written
            // by the lane itself or with v_writelane -- no LDS round trip, no barrier.
            int cref_v = -2, cmvx_v = 0, cmvy_v = 0, pme_v = 0;
            int sub_mx = 0, sub_my = 0, sub_cost = 0, sub_px = 0, sub_py = 0, sub_t = 3;      // sub-8x8 records (lanes 0..31) and chosen type (lanes 0..3)
            if (is_p && (RD || (a.flags_inter & 0x10))) {
                // the full motion cache for x264_mb_predict_mv on partitions: -2 = not available, neighbours as cache_load leaves them
                if ((nb & NB_TOP) && lane >= 4 && lane < 8) {
                    const int o = mb - a.mb_w, k = lane - 4;
                    cref_v = a.ref[o * 4 + 2 + (k >> 1)]; cmvx_v = a.mv[(o * 16 + 12 + k) * 2]; cmvy_v = a.mv[(o * 16 + 12 + k) * 2 + 1];
                }
                if ((nb & NB_TOPLEFT) && lane == 3) {
                    const int o = mb - a.mb_w - 1;
                    cref_v = a.ref[o * 4 + 3]; cmvx_v = a.mv[(o * 16 + 15) * 2]; cmvy_v = a.mv[(o * 16 + 15) * 2 + 1];
                }
                if ((nb & NB_TOPRIGHT) && lane == 8) {
                    const int o = mb - a.mb_w + 1;
                    cref_v = a.ref[o * 4 + 2]; cmvx_v = a.mv[(o * 16 + 12) * 2]; cmvy_v = a.mv[(o * 16 + 12) * 2 + 1];
                }
                if ((nb & NB_LEFT) && lane >= 11 && lane < 36 && ((lane - 11) & 7) == 0) {
                    const int i = (lane - 11) >> 3;
                    cref_v = s.left_r8[i >> 1]; cmvx_v = s.left_mv4[i][0]; cmvy_v = s.left_mv4[i][1];
                }
            }
            if constexpr (RD) { if (is_p && lane == 30) { cref_v = st0r; cmvx_v = st0x; cmvy_v = st0y; } }      // the entry cache_load does not rewrite
            if constexpr (RD) {     // the neighbours' part of the motion cache, for the entropy coder's x264_mb_predict_mv / ref contexts
                if (is_p) {
                    if (lane < 48) { sr.cref[lane] = (signed char)cref_v; sr.cmv[lane][0] = (i16)cmvx_v; sr.cmv[lane][1] = (i16)cmvy_v; }
                    WAVE_SYNC();
                }
            }
            // x264_mb_predict_mv_16x16, :90-128
            auto predict16 = [&](int i_ref, int &px, int &py) {
                const int cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
                if (cnt > 1) { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
                else if (cnt == 1) { if (ra == i_ref) { px = ax; py = ay; } else if (rb == i_ref) { px = bx; py = byv; } else { px = cx; py = cy; } }
                else if (rb == -2 && rc == -2 && ra != -2) { px = ax; py = ay; }
                else { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
            };
            // x264_mb_predict_mv_pskip, :131-149
            if (ra == -2 || rb == -2 || !(ra | ax | ay) || !(rb | bx | byv)) { pskx = 0; psky = 0; }
            else predict16(0, pskx, psky);

            int b_skip = 0, try_pskip = 0;
            if (is_p && a.fast_pskip) {
                if (a.subme >= 3) try_pskip = 1;
                else if (left_type == T_P_SKIP || type_top == T_P_SKIP || type_topleft == T_P_SKIP || type_topright == T_P_SKIP) {
                    b_skip = sw_probe_pskip(s, refs, a, Q, pskx, psky, mbx, mby, oy, oc, by_, bc_, lane);
                    skip_mc = b_skip;
                }
            }
            if (b_skip) type = T_P_SKIP;
            else {
                // ---- x264_mb_analyse_inter_p16x16, R/encoder/analyse.c:1077-1143 ----
                const MeLimits L = me_limits(mbx, mby, a.mb_w, a.mb_h, a.mv_range);
                MxCtx c;
                c.fe = (MX_LDS(u32))s.fe; c.fe_u = (MX_LDS(u8))(s.fe + 256); c.fe_v = (MX_LDS(u8))(s.fe + 320); c.sy = a.sy; c.sc = a.sc; c.lane = lane; c.set_block(16, 16, 0, 0);
                c.cost_g = (MX_GLB(i16))cost_g;           // (the current macroblock's QP: with adaptive quantisation not the slice's)
                c.cost_l = (MX_LDS(i16))s.costl; c.has_cost_l = true; c.patch = (MX_LDS(u8))s.patch; c.has_patch = true; c.patch_on = false;
                int thresh = 0x7fffffff, best = 0x7fffffff, bmvpx = 0, bmvpy = 0;
                bool early_skip = false;
                for (int r = 0; r < (is_p ? a.n_refs : 0); r++) {
                    int mvpx, mvpy;
                    predict16(r, mvpx, mvpy);
                    // x264_mb_predict_mv_ref16x16, R/common/macroblock.c:376-437
                    int n_mvc = 0;
                    {
                        const i16 *mvr = a.mvr + (size_t)r * nmb * 2;
                        const int top = mb - a.mb_w;
                        WAVE_SYNC();                                   // the previous reference's candidates have been read
                        // every lane stores the same values: the list is wave-uniform
#define SETC(vx_, vy_) do { s.mvc[n_mvc][0] = (i16)(vx_); s.mvc[n_mvc][1] = (i16)(vy_); n_mvc++; } while (0)
                        if (r == 0 && a.lowres0) {       // the lookahead's vector, twice (R/common/macroblock.c:393-398); 0x7fff in the chain's first component: none
                            const i16 *lw = a.lowres0 + 2 * cb;      // (re-derived from the argument where it is used: nothing to keep live across the macroblock)
                            if (UNI(lw[0]) != 0x7fff) SETC((u16)(UNI(lw[2 * mb]) << 1), (u16)(UNI(lw[2 * mb + 1]) << 1));
                        }
                        if ((nb & NB_LEFT) && left_type != T_P_SKIP) SETC(s.left_mvr[r][0], s.left_mvr[r][1]);
                        if (nb & NB_TOP) {
                            if (type_top != T_P_SKIP) SETC(mvr[2 * top], mvr[2 * top + 1]);
                            if ((nb & NB_TOPLEFT) && type_topleft != T_P_SKIP) SETC(mvr[2 * (top - 1)], mvr[2 * (top - 1) + 1]);
                            if (mbx < a.mb_w - 1 && type_topright != T_P_SKIP) SETC(mvr[2 * (top + 1)], mvr[2 * (top + 1) + 1]);
                        }
                        if (a.l0_n_ref0 > 0)
                            for (int k = 0; k < 3; k++) {
                                const int dx = k == 1, dy = k == 2;
                                if ((dx && mbx >= a.mb_w - 1) || (dy && mby >= a.mb_h - 1)) continue;
                                const int o = mb + dx + dy * a.mb_w, ref_col = a.l0_ref[o * 4];
                                if (ref_col >= 0) {
                                    const int scale = refs.poc_delta[r] * refs.l0_inv_ref_poc[ref_col];
                                    SETC((a.l0_mv[o * 32] * scale + 128) >> 8, (a.l0_mv[o * 32 + 1] * scale + 128) >> 8);
                                }
                            }
#undef SETC
                        WAVE_SYNC();
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) c.pl[k] = (MX_GLB(u8))(refs.y[r][k] + by_ + oy);
                    c.cu = (MX_GLB(u8))(refs.u[r] + bc_ + oc); c.cv = (MX_GLB(u8))(refs.v[r] + bc_ + oc);
                    c.mvpx = mvpx; c.mvpy = mvpy;
                    thresh -= (Q.lambda * refs.ref_bits[r]);
                    int smx, smy, cost_mv;
                    LAUNDER(); c.lane = lane;
                    int cost = me_search_ref16(c, L, mo, &s.mvc[0][0], n_mvc, &thresh, smx, smy, cost_mv);   // with one reference the threshold never bites (it starts at COST_MAX); a conditional
                                                                                        // pointer would pin it in scratch memory
                    if (r == 0 && try_pskip && cost - cost_mv < 300 * Q.lambda && iabs(smx - pskx) + iabs(smy - psky) <= 1) {
                        if (sw_probe_pskip(s, refs, a, Q, pskx, psky, mbx, mby, oy, oc, by_, bc_, lane)) { early_skip = true; break; }
                    }
                    cost += (Q.lambda * refs.ref_bits[r]);
                    thresh += (Q.lambda * refs.ref_bits[r]);
                    if (cost < best) { best = cost; mvx = smx; mvy = smy; ref = r; bmvpx = mvpx; bmvpy = mvpy; }
                    if (lane == 0) {
                        a.mvr[((size_t)r * nmb + mb) * 2] = (i16)smx; a.mvr[((size_t)r * nmb + mb) * 2 + 1] = (i16)smy;
                        s.left_mvr[r][0] = (i16)smx; s.left_mvr[r][1] = (i16)smy;
                        s.l0mvc[r][0][0] = (i16)smx; s.l0mvc[r][0][1] = (i16)smy;          // a->l0.mvc[i_ref][0]
                    }
                }
                if (early_skip) { type = T_P_SKIP; skip_mc = 1; }
                else {
                    type = T_P_L0;
                    // point the search context at a block of reference r (LOAD_HPELS, analyse.c:1065-1072)
                    auto aim = [&](int r, int w, int h, int bx, int by) {
#pragma unroll
                        for (int k = 0; k < 4; k++) c.pl[k] = (MX_GLB(u8))(refs.y[r][k] + by_ + oy + (ptrdiff_t)by * a.sy + bx);
                        c.cu = (MX_GLB(u8))(refs.u[r] + bc_ + oc + (ptrdiff_t)(by >> 1) * a.sc + (bx >> 1)); c.cv = (MX_GLB(u8))(refs.v[r] + bc_ + oc + (ptrdiff_t)(by >> 1) * a.sc + (bx >> 1));
                        c.set_block(w, h, bx, by);
                    };
                    // candidate records of the partition analysis (x264_me_t's mv / cost / cost_mv / i_ref / i_ref_cost / mvp):
                    // slots 0-3 me8x8, 4-5 me16x8, 6-7 me8x16.  Wave-uniform values, parked in LDS because they are indexed.
                    auto pme_put = [&](int slot, int vx, int vy, int cost, int cost_mv, int r, int ref_cost, int px, int py) {
                        const int f = lane - slot * 8;                          // this lane's field of that record, if 0..7
                        pme_v = f == 0 ? vx : f == 1 ? vy : f == 2 ? cost : f == 3 ? cost_mv : f == 4 ? r : f == 5 ? ref_cost : f == 6 ? px : f == 7 ? py : pme_v;
                    };
                    auto pme = [&](int slot, int f) -> int { return __builtin_amdgcn_readlane(pme_v, slot * 8 + f); };
                    // x264_macroblock_cache_ref / _mv on a run of 4x4 blocks of the motion cache
                    auto cache_set = [&](int x, int y, int w, int h, int r, int vx, int vy, int set_mv) {
                        const int k = lane - 12, i = k & 7, j = k >> 3;          // cache entry 12 + i + 8 j = 4x4 block (i, j)
                        if (k >= 0 && i < 4 && j < 4 && i >= x && i < x + w && j >= y && j < y + h) {
                            cref_v = r;
                            if (set_mv) { cmvx_v = vx; cmvy_v = vy; }
                        }
                    };
                    // x264_mb_predict_mv (R/common/macroblock.c:28-88) from the cache; cur_part = h->mb.i_partition
                    auto predict_blk = [&](int cur_part, int idx, int width, int &px, int &py) {
                        const int i8 = sw_scan8(idx), i_ref = __builtin_amdgcn_readlane(cref_v, i8);
                        int ra = __builtin_amdgcn_readlane(cref_v, i8 - 1), rb = __builtin_amdgcn_readlane(cref_v, i8 - 8), rc = __builtin_amdgcn_readlane(cref_v, i8 - 8 + width), kc = i8 - 8 + width;
                        if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { kc = i8 - 8 - 1; rc = __builtin_amdgcn_readlane(cref_v, kc); }
                        const int ax = __builtin_amdgcn_readlane(cmvx_v, i8 - 1), ay = __builtin_amdgcn_readlane(cmvy_v, i8 - 1), bx = __builtin_amdgcn_readlane(cmvx_v, i8 - 8), byv = __builtin_amdgcn_readlane(cmvy_v, i8 - 8);
                        const int cx = __builtin_amdgcn_readlane(cmvx_v, kc), cy = __builtin_amdgcn_readlane(cmvy_v, kc);
                        if (cur_part == 14) {                       // D_16x8
                            if (idx == 0 && rb == i_ref) { px = bx; py = byv; return; }
                            if (idx != 0 && ra == i_ref) { px = ax; py = ay; return; }
                        } else if (cur_part == 15) {                // D_8x16
                            if (idx == 0 && ra == i_ref) { px = ax; py = ay; return; }
                            if (idx != 0 && rc == i_ref) { px = cx; py = cy; return; }
                        }
                        const int cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
                        if (cnt > 1) { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
                        else if (cnt == 1) { if (ra == i_ref) { px = ax; py = ay; } else if (rb == i_ref) { px = bx; py = byv; } else { px = cx; py = cy; } }
                        else if (rb == -2 && rc == -2 && ra != -2) { px = ax; py = ay; }
                        else { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
                    };
                    int i_cost = best;
                    int c8x8 = MX_COST_MAX, c16x8 = MX_COST_MAX, c8x16 = MX_COST_MAX;   // a->l0.i_cost8x8 / i_cost16x8 / i_cost8x16
                    auto search_partitions = [&]() {
                    part = 16;                                       // D_16x16
                    if (a.flags_inter & 0x10) {
                        // ---- X264_ANALYSE_PSUB16x16: p8x8, then p16x8 / p8x16 (R/encoder/analyse.c:2222-2265) ----
                        cache_set(0, 0, 4, 4, ref, 0, 0, 0);
                        int cost8x8;
                        if (a.mixed_refs) {                          // x264_mb_analyse_inter_p8x8_mixed_ref, :1146-1219
                            int maxref = a.n_refs - 1;
                            const int tt = type_top == T_I_8x8 ? 0 : type_top, tl = left_type == T_I_8x8 ? 0 : left_type;   // as cache_save stores them
                            if (maxref > 0 && ref == 0 && tt && tl) {
                                maxref = 0;
                                maxref = max(maxref, __builtin_amdgcn_readlane(cref_v, 3)); maxref = max(maxref, __builtin_amdgcn_readlane(cref_v, 4)); maxref = max(maxref, __builtin_amdgcn_readlane(cref_v, 6));
                                maxref = max(maxref, __builtin_amdgcn_readlane(cref_v, 8)); maxref = max(maxref, __builtin_amdgcn_readlane(cref_v, 11)); maxref = max(maxref, __builtin_amdgcn_readlane(cref_v, 27));
                            }
                            for (int i = 0; i < 4; i++) {
                                int bcost = 0x7fffffff, bvx = 0, bvy = 0, bcm = 0, br = 0, bpx = 0, bpy = 0;
                                for (int r = 0; r <= maxref; r++) {
                                    cache_set(2 * (i & 1), 2 * (i >> 1), 2, 2, r, 0, 0, 0);
                                    int px, py, vx, vy, cm;
                                    predict_blk(13, 4 * i, 2, px, py);
                                    aim(r, 8, 8, 8 * (i & 1), 8 * (i >> 1));
                                    c.mvpx = px; c.mvpy = py;
                                    LAUNDER(); c.lane = lane;
                                    int cost = me_search_ref16(c, L, mo, &s.l0mvc[r][0][0], i + 1, nullptr, vx, vy, cm) + (Q.lambda * refs.ref_bits[r]);
                                    if (lane == 0) { s.l0mvc[r][i + 1][0] = (i16)vx; s.l0mvc[r][i + 1][1] = (i16)vy; }
                                    WAVE_SYNC();
                                    if (cost < bcost) { bcost = cost; bvx = vx; bvy = vy; bcm = cm; br = r; bpx = px; bpy = py; }
                                }
                                cache_set(2 * (i & 1), 2 * (i >> 1), 2, 2, br, bvx, bvy, 1);
                                pme_put(i, bvx, bvy, bcost + Q.lambda, bcm, br, (Q.lambda * refs.ref_bits[br]), bpx, bpy);      // + lambda * i_sub_mb_p_cost_table[D_L0_8x8]
                            }
                            cost8x8 = pme(0, 2) + pme(1, 2) + pme(2, 2) + pme(3, 2);
                            if (!a.cabac && !(pme(0, 4) | pme(1, 4) | pme(2, 4) | pme(3, 4))) cost8x8 -= (Q.lambda * refs.ref_bits[0]) * 4;
                        } else {                                     // x264_mb_analyse_inter_p8x8, :1221-1272
                            const int r = ref, ref_cost = a.cabac || r ? (Q.lambda * refs.ref_bits[r]) : 0;
                            if (lane == 0) { s.l0mvc[r][0][0] = (i16)mvx; s.l0mvc[r][0][1] = (i16)mvy; }
                            WAVE_SYNC();
                            for (int i = 0; i < 4; i++) {
                                int px, py, vx, vy, cm;
                                predict_blk(13, 4 * i, 2, px, py);
                                aim(r, 8, 8, 8 * (i & 1), 8 * (i >> 1));
                                c.mvpx = px; c.mvpy = py;
                                LAUNDER(); c.lane = lane;
                                const int cost = me_search_ref16(c, L, mo, &s.l0mvc[r][0][0], i + 1, nullptr, vx, vy, cm);
                                cache_set(2 * (i & 1), 2 * (i >> 1), 2, 2, r, vx, vy, 1);
                                if (lane == 0) { s.l0mvc[r][i + 1][0] = (i16)vx; s.l0mvc[r][i + 1][1] = (i16)vy; }
                                pme_put(i, vx, vy, cost + ref_cost + Q.lambda, cm, r, ref_cost, px, py);
                            }
                            cost8x8 = pme(0, 2) + pme(1, 2) + pme(2, 2) + pme(3, 2);
                            if (a.cabac) cost8x8 -= ref_cost;
                        }
                        if (cost8x8 < best) { type = T_P_8x8; part = 13; i_cost = cost8x8; }
                        if ((a.flags_inter & 0x20) && type == T_P_8x8) {
                            // ---- X264_ANALYSE_PSUB8x8 (R/encoder/analyse.c:2252-2277): p4x4, and only if that beats the 8x8 block, p8x4 and p4x8
                            // (:1407-1519).  Records (mv, cost, mvp) of me4x4[i][k] / me8x4[i][k] / me4x8[i][k] sit in lanes 4i+k / 16+2i+k / 24+2i+k.
                            MeOpts mo_sub = mo;
                            mo_sub.chroma_me = 0;                        // b_chroma_me && i_pixel <= PIXEL_8x8, me.c:654
                            for (int i = 0; i < 4; i++) {
                                const int r = pme(i, 4), x0 = 2 * (i & 1), y0 = 2 * (i >> 1);
                                int c8 = 0, subt = 3;
                                for (int t = 0; t < 3; t++) {
                                    const int sw = t == 1 ? 2 : 1, sh = t == 2 ? 2 : 1, sn = t == 0 ? 4 : 2, rec0 = t == 0 ? 4 * i : t == 1 ? 16 + 2 * i : 24 + 2 * i;
                                    const int cvx = t == 0 ? pme(i, 0) : __builtin_amdgcn_readlane(sub_mx, 4 * i), cvy = t == 0 ? pme(i, 1) : __builtin_amdgcn_readlane(sub_my, 4 * i);
                                    int sum = 0;
                                    for (int k = 0; k < sn; k++) {
                                        const int x4 = x0 + (t == 0 ? (k & 1) : t == 2 ? k : 0), y4 = y0 + (t == 0 ? (k >> 1) : t == 1 ? k : 0);
                                        const int idx = 4 * i + (y4 - y0) * 2 + (x4 - x0);
                                        int px, py, vx, vy, cm;
                                        predict_blk(13, idx, sw, px, py);
                                        aim(r, 4 * sw, 4 * sh, 4 * x4, 4 * y4);
                                        c.mvpx = px; c.mvpy = py;
                                        WAVE_SYNC();
                                        if (lane < 2) s.mvc[0][lane] = (i16)(lane ? cvy : cvx);
                                        WAVE_SYNC();
                                        LAUNDER(); c.lane = lane;
                                        const int cost = me_search_ref16(c, L, mo_sub, &s.mvc[0][0], k == 0 ? 1 : 0, nullptr, vx, vy, cm);
                                        if (lane == rec0 + k) { sub_mx = vx; sub_my = vy; sub_cost = cost; sub_px = px; sub_py = py; }
                                        cache_set(x4, y4, sw, sh, r, vx, vy, 1);
                                        sum += cost;
                                    }
                                    int cst = sum + (Q.lambda * refs.ref_bits[r]) + Q.lambda * (t == 0 ? 5 : 3);          // i_sub_mb_p_cost_table
                                    if (a.chroma_me && a.subme >= 5) cst += sw_sub_chroma(s, refs, a, r, i, t, rec0, sub_mx, sub_my, satd, oc, bc_, lane);
                                    if (t == 0) {
                                        if (!(cst < pme(i, 2))) break;
                                        c8 = cst; subt = 0;
                                    } else if (cst < c8) { c8 = cst; subt = t; }
                                }
                                if (subt != 3) i_cost += c8 - pme(i, 2);
                                // x264_mb_cache_mv_p8x8
                                if (subt == 3) cache_set(x0, y0, 2, 2, r, pme(i, 0), pme(i, 1), 1);
                                else {
                                    const int sw = subt == 1 ? 2 : 1, sh = subt == 2 ? 2 : 1, sn = subt == 0 ? 4 : 2, rec0 = subt == 0 ? 4 * i : subt == 1 ? 16 + 2 * i : 24 + 2 * i;
                                    for (int k = 0; k < sn; k++)
                                        cache_set(x0 + (subt == 0 ? (k & 1) : subt == 2 ? k : 0), y0 + (subt == 0 ? (k >> 1) : subt == 1 ? k : 0), sw, sh, r,
                                                  __builtin_amdgcn_readlane(sub_mx, rec0 + k), __builtin_amdgcn_readlane(sub_my, rec0 + k), 1);
                                }
                                if (lane == i) sub_t = subt;
                            }
                            cost8x8 = i_cost;
                        }
                        const int thresh16x8 = pme(1, 3) + pme(2, 3);
                        if (cost8x8 < best + thresh16x8)
                            for (int dir = 0; dir < 2; dir++) {      // 0: x264_mb_analyse_inter_p16x8 (:1274), 1: _p8x16 (:1324)
                                int sum = 0;
                                for (int i = 0; i < 2; i++) {
                                    const int ra = dir ? pme(i, 4) : pme(2 * i, 4), rb = dir ? pme(i + 2, 4) : pme(2 * i + 1, 4), nr = ra == rb ? 1 : 2;
                                    int bcost = 0x7fffffff, bvx = 0, bvy = 0, bcm = 0, br = 0, bpx = 0, bpy = 0;
                                    for (int j = 0; j < nr; j++) {
                                        const int r = j ? rb : ra, k1 = dir ? i + 1 : 2 * i + 1, k2 = dir ? i + 3 : 2 * i + 2;
                                        WAVE_SYNC();
                                        if (lane < 6) {
                                            const int k = lane >> 1 == 0 ? 0 : lane >> 1 == 1 ? k1 : k2;
                                            s.mvc[lane >> 1][lane & 1] = s.l0mvc[r][k][lane & 1];
                                        }
                                        if (dir) cache_set(2 * i, 0, 2, 4, r, 0, 0, 0); else cache_set(0, 2 * i, 4, 2, r, 0, 0, 0);
                                        int px, py, vx, vy, cm;
                                        predict_blk(dir ? 15 : 14, dir ? 4 * i : 8 * i, dir ? 2 : 4, px, py);
                                        aim(r, dir ? 8 : 16, dir ? 16 : 8, dir ? 8 * i : 0, dir ? 0 : 8 * i);
                                        c.mvpx = px; c.mvpy = py;
                                        LAUNDER(); c.lane = lane;
                                        const int cost = me_search_ref16(c, L, mo, &s.mvc[0][0], 3, nullptr, vx, vy, cm) + (Q.lambda * refs.ref_bits[r]);
                                        if (cost < bcost) { bcost = cost; bvx = vx; bvy = vy; bcm = cm; br = r; bpx = px; bpy = py; }
                                    }
                                    if (dir) cache_set(2 * i, 0, 2, 4, br, bvx, bvy, 1); else cache_set(0, 2 * i, 4, 2, br, bvx, bvy, 1);
                                    pme_put(4 + 2 * dir + i, bvx, bvy, bcost, bcm, br, (Q.lambda * refs.ref_bits[br]), bpx, bpy);
                                    sum += bcost;
                                }
                                if (dir) c8x16 = sum; else c16x8 = sum;
                                if (sum < i_cost) { i_cost = sum; type = T_P_L0; part = dir ? 15 : 14; }
                            }
                        c8x8 = cost8x8;
                    }
                    };
                    // x264_me_refine_qpel on the winning partition (analyse.c:2289-2352); the reference cost leaves every block's sum (me.c:639-640)
                    auto refine_winner = [&]() {
                    if (part == 16) {
                        aim(ref, 16, 16, 0, 0);
                        c.mvpx = bmvpx; c.mvpy = bmvpy;
                        best -= (Q.lambda * refs.ref_bits[ref]);
                        LAUNDER(); c.lane = lane;
                        best = me_refine_qpel16(c, L, mo, best, mvx, mvy);
                        i_cost = best;
                        if (lane < 16) { s.mv4[lane][0] = (i16)mvx; s.mv4[lane][1] = (i16)mvy; }
                        if (lane < 4) s.ref8[lane] = (signed char)ref;
                    } else {
                        i_cost = 0;
                        const int nblk = part == 13 ? 4 : 2, slot0 = part == 13 ? 0 : part == 14 ? 4 : 6;
                        for (int i = 0; i < nblk; i++) {
                            // an 8x8 block of a P_8x8 macroblock refines its sub-partitions (analyse.c:2317-2352): no reference cost in their
                            // sums and no chroma (me.c:639, :654)
                            const int subt = part == 13 ? __builtin_amdgcn_readlane(sub_t, i) : 3, nj = subt == 3 ? 1 : subt == 0 ? 4 : 2;
                            for (int k = 0; k < nj; k++) {
                                int bx = part == 13 ? 8 * (i & 1) : part == 15 ? 8 * i : 0, by = part == 13 ? 8 * (i >> 1) : part == 14 ? 8 * i : 0;
                                int w = part == 14 ? 16 : 8, h = part == 15 ? 16 : 8;
                                const int r = pme(slot0 + i, 4);
                                int vx = pme(slot0 + i, 0), vy = pme(slot0 + i, 1), cin = pme(slot0 + i, 2) - pme(slot0 + i, 5);
                                MeOpts mo_r = mo;
                                c.mvpx = pme(slot0 + i, 6); c.mvpy = pme(slot0 + i, 7);
                                if (subt != 3) {
                                    const int rec = (subt == 0 ? 4 * i : subt == 1 ? 16 + 2 * i : 24 + 2 * i) + k;
                                    bx += 4 * (subt == 0 ? (k & 1) : subt == 2 ? k : 0); by += 4 * (subt == 0 ? (k >> 1) : subt == 1 ? k : 0);
                                    w = subt == 1 ? 8 : 4; h = subt == 2 ? 8 : 4;
                                    vx = __builtin_amdgcn_readlane(sub_mx, rec); vy = __builtin_amdgcn_readlane(sub_my, rec); cin = __builtin_amdgcn_readlane(sub_cost, rec);
                                    c.mvpx = __builtin_amdgcn_readlane(sub_px, rec); c.mvpy = __builtin_amdgcn_readlane(sub_py, rec);
                                    mo_r.chroma_me = 0;
                                }
                                aim(r, w, h, bx, by);
                                LAUNDER(); c.lane = lane;
                                i_cost += me_refine_qpel16(c, L, mo_r, cin, vx, vy);
                                WAVE_SYNC();
                                if (lane < 16) {
                                    const int x4 = (lane & 3) * 4, y4 = (lane >> 2) * 4;
                                    if (x4 >= bx && x4 < bx + w && y4 >= by && y4 < by + h) { s.mv4[lane][0] = (i16)vx; s.mv4[lane][1] = (i16)vy; }
                                }
                                if (lane < 4) {
                                    const int x8 = (lane & 1) * 8, y8 = (lane >> 1) * 8;
                                    if (x8 >= (bx & ~7) && x8 < (bx & ~7) + (w < 8 ? 8 : w) && y8 >= (by & ~7) && y8 < (by & ~7) + (h < 8 ? 8 : h)) s.ref8[lane] = (signed char)r;
                                }
                            }
                        }
                    }
                    };
                    if constexpr (!RD) {
                    search_partitions();
                    refine_winner();
                    WAVE_SYNC();
                    if (part == 13) sub_t_mb = sub_t;
                    PROF(2);
                    LAUNDER();
                    if (a.chroma_me) {
                        analyse_chroma();
                        analyse_intra(i_cost - satd_chroma);
                        satd_i16 += satd_chroma; satd_i8 += satd_chroma; satd_i4 += satd_chroma;
                    } else
                        analyse_intra(i_cost);
                    if (fi_open) {
                        if (min(satd_i8, satd_i4) < i_cost) {        // the answer decides the macroblock type: it must be exact
                            if (fast_intra_now(1)) satd_i8 = satd_i4 = MX_COST_MAX;
                            fi_open = 0;
                        } else
                            stat_alt = satd_i16;                     // i_intra_cost if b_fast_intra turns out to be 1
                    }
                    // analyse.c:2372-2400: best intra type (16x16, then 8x8, then 4x4 on strict improvement) against inter
                    int itype = T_I_16x16, icost = satd_i16;
                    if (satd_i8 < icost) { icost = satd_i8; itype = T_I_8x8; }
                    if (satd_i4 < icost) { icost = satd_i4; itype = T_I_4x4; }
                    if (icost < i_cost) { i_cost = icost; type = itype; }
                    stat_intra = icost; analysed = 1;
                    stat_inter = i_cost;
                    } else {
                        // ---- the raster variant's P macroblock (analyse.c:2228-2405): the rest of the analysis, the RD candidates of
                        // x264_mb_analyse_p_rd / x264_mb_analyse_transform_rd / x264_intra_rd, and the final encode, through ONE copy of
                        // x264_rd_cost_mb: step 0 the early 16x16 trial (:1134-1143), 1 the analysis, 2-5 p_rd, 6 the transform, 7-9 intra, 10 final.
                        int me16x = mvx, me16y = mvy;                    // (the RD refinement moves the 16x16 vector)
                        const int me16r = ref;
                        int i8_cbp_rd = 0;                               // a->i_cbp_i8x8_luma (x264_intra_rd, analyse.c:869)
                        // the RD refinement's state (slice_refine.h): what is being refined, the best cost so far, and the candidate generator of
                        // x264_me_refine_qpel_rd
                        int rf_kind = 0, rf_i = 0, rf_n = 0, rf_old16 = 0, rf_best16 = 0, rf_thr = 0;
                        u32 rf_list = 0;
                        unsigned long long rf_best = 0;
                        int q_st = 0, q_j = 0, q_it = 0, q_dir = -2, q_odir = 0, q_tag = 0, q_after_pm = 0;
                        int q_bmx = 0, q_bmy = 0, q_omx = 0, q_omy = 0, q_pmx = 0, q_pmy = 0, q_m0x = 0, q_m0y = 0, q_mvpx = 0, q_mvpy = 0, q_cx = 0, q_cy = 0;
                        int q_pix = 0, q_bx = 0, q_by = 0, q_w = 16, q_h = 16, q_slot = -1, q_ref = 0, q_i4 = 0, q_satds = 0;
                        u32 q_bsatd = 0;
                        (void)rf_kind; (void)rf_i; (void)rf_n; (void)rf_old16; (void)rf_best16; (void)rf_thr; (void)rf_list; (void)rf_best; (void)i8_cbp_rd;
                        (void)q_st; (void)q_j; (void)q_it; (void)q_dir; (void)q_odir; (void)q_tag; (void)q_after_pm; (void)q_bmx; (void)q_bmy; (void)q_omx; (void)q_omy;
                        (void)q_pmx; (void)q_pmy; (void)q_m0x; (void)q_m0y; (void)q_mvpx; (void)q_mvpy; (void)q_cx; (void)q_cy; (void)q_pix; (void)q_bx; (void)q_by;
                        (void)q_w; (void)q_h; (void)q_slot; (void)q_ref; (void)q_i4; (void)q_satds; (void)q_bsatd;
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
                            {   // the motion cache's copy of block 12 (raster block 10, 8x8 block 3) follows the candidate
                                const int nx = __shfl(vx, 10, 64), ny = __shfl(vy, 10, 64), nr = __shfl(vr, 3, 64);
                                if (lane == 30) { cref_v = nr; cmvx_v = nx; cmvy_v = ny; }
                            }
                            mvx = me16x; mvy = me16y; ref = me16r;
                            WAVE_SYNC();
                        };
#pragma nounroll
                        for (int step = 0; step < (RF ? 13 : 11); step++) {      // RF: 10 decides, 11 refines (once per full-macroblock candidate), 12 is the final encode
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
                            } else if (step == 7) {                                                // x264_intra_rd, :845-874 (threshold COST_MAX in an I slice)
                                if (!(satd_i16 <= (is_p ? satd_inter * 5 / 4 : MX_COST_MAX))) { satd_i16 = MX_COST_MAX; continue; }
                                type = T_I_16x16;
                            } else if (step == 8) {
                                if (!(satd_i4 <= (is_p ? satd_inter * 5 / 4 : MX_COST_MAX) && satd_i4 < MX_COST_MAX)) { satd_i4 = MX_COST_MAX; continue; }
                                type = T_I_4x4;
                            } else if (step == 9) {
                                if (!(satd_i8 <= (is_p ? satd_inter * 5 / 4 : MX_COST_MAX) && satd_i8 < MX_COST_MAX)) { satd_i8 = MX_COST_MAX; continue; }
                                type = T_I_8x8;
                            } else if (!RF || step == 10) {
                                fin = !RF;
                                if (!is_p) {                                 // analyse.c:2179-2184: 16x16, then 4x4, then 8x8, then PCM on strict improvement
                                    type = T_I_16x16;
                                    int ic = satd_i16;
                                    if (satd_i4 < ic) { ic = satd_i4; type = T_I_4x4; }
                                    if (satd_i8 < ic) { ic = satd_i8; type = T_I_8x8; }
                                    if (satd_pcm < ic) type = T_I_PCM;
                                } else if (rd_skip) type = T_P_SKIP;
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
                                }
                                if constexpr (RF) {
                                    // analyse.c:2184-2185 (I), :2406-2464 (P): with a->i_mbrd >= 2 the winner's intra modes / vectors are refined by RD
                                    rf_kind = 9;
                                    if (mbrd >= 2 && type != T_I_PCM && !rd_skip) {
                                        if (IS_INTRA_T(type)) {
                                            skip_intra = 0;                  // x264_intra_rd_refine's first statement
                                            rf_kind = 2;
                                            if (type == T_I_16x16) {
                                                rf_kind = 1; rf_list = sw_modes16(nb, rf_n); rf_i = 0; rf_old16 = rf_best16 = pred16;
                                                rf_thr = UNI(sf.i16dir[pred16]) * 9 / 8; rf_best = (unsigned long long)(u32)satd_i16;
                                            }
                                        } else { rf_kind = 3; rf_i = 0; q_st = -1; }
                                    }
                                    continue;
                                } else {
                                    tq.on = rd.trellis != 0;                                      // :2768-2773
                                    if (rd.trellis == 1 || a.nr) skip_intra = 0;
                                }
                            } else if (step == 11) {
                                if constexpr (RF) {
#include "slice_refine.h"
                                }
                            } else {
                                fin = true;
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
                            else if (step == 9) { satd_i8 = c; i8_cbp_rd = cbp_luma; }
                            else if constexpr (RF) {                     // step 11: the full-macroblock candidate the refinement asked for
                                if (rf_kind == 1) { if ((unsigned long long)(u32)c < rf_best) { rf_best = (unsigned long long)(u32)c; rf_best16 = pred16; } }
                                else {
                                    type = T_P_L0;                       // x264_rd_cost_part( .., PIXEL_16x16 ) restores h->mb.i_type (rdo.c:209-213)
                                    if ((unsigned long long)(u32)c < rf_best) { rf_best = (unsigned long long)(u32)c; q_bmx = q_cx; q_bmy = q_cy; if (q_tag != -3) q_dir = q_tag; }
                                }
                                step = 10;                               // back into the refinement
                            }
                        }
                    }
                }
            }
            if constexpr (RD) {     // what the next macroblock finds in the cache's entry of block 12 (oracle/slice_oracle.c: stale_ref)
                if (is_p && rd.stale) {
                    if (type == T_P_SKIP) { st0r = 0; st0x = pskx; st0y = psky; }
                    else if (!IS_INTRA_T(type) && encoded) { st0r = UNI(s.ref8[3]); st0x = UNI(s.mv4[10][0]); st0y = UNI(s.mv4[10][1]); }
                    else { st0r = __builtin_amdgcn_readlane(cref_v, 30); st0x = __builtin_amdgcn_readlane(cmvx_v, 30); st0y = __builtin_amdgcn_readlane(cmvy_v, 30); }
                }
            }
        }
        (void)analysed;
        if constexpr (!RD) PROF(6);
        LAUNDER();

        // ---- x264_analyse_update_cache + x264_macroblock_encode ----
        if constexpr (!RD) encode_mb(1);
        else if (!encoded) encode_pskip();                     // the fast / early P_SKIP exits of the analysis
        const int intra = IS_INTRA_T(type);
        int mb_qp = Q.qp, cbp_store = 0;
        if constexpr (RD) {
            PROF(3);
            if (type == T_I_PCM) {          // the samples themselves are sent: the reconstruction is the source (R/encoder/cabac.c:801-818)
                *(u32 *)(s.fd + FDY + (lane >> 2) * FD + (lane & 3) * 4) = *(const u32 *)(s.fe + (lane >> 2) * 16 + (lane & 3) * 4);
                s.fd[FDU + (lane >> 3) * FD + (lane & 7)] = s.fe[256 + lane]; s.fd[FDV + (lane >> 3) * FD + (lane & 7)] = s.fe[320 + lane];
                cbp_luma = 0xf; cbp_chroma = 2; t8 = 0;
                WAVE_SYNC();
            }
            // ---- the entropy coder, where x264_slice_write has it (R/encoder/encoder.c:1192-1205) ----
            if (rd.write) {
                syn_prepare();
                const MbSynDev y0 = make_syn();
                if (lane == 0) {
                    if (mb > 0) cd_encode_terminal(cab);
                    if (IS_SKIP_T(type)) cw_mb_skip(cab, sr.cabac, left_type, type_top, 1, a.slice_type);
                    else {
                        if (is_p || BS) cw_mb_skip(cab, sr.cabac, left_type, type_top, 0, a.slice_type);
                        MbSynDev y = y0;
                        cw_macroblock(cab, sr.cabac, 0, y, s.fe, rd.i_frame + bz * rd.i_frame_stride);
                        sr.tmp_i[1] = y.qp;
                    }
                    const int pos = cd_pos(cab, payload0);
                    if (rd.mb_bits) rd.mb_bits[cb + mb] = pos;
                    sr.tmp_i[2] = (pos >> 3) + SW_MB_BYTES_MAX + 64 > rd.payload_cap;      // the next macroblock (and the flush) may not fit
                }
                WAVE_SYNC();
                if (UNI(sr.tmp_i[2])) {          // out of payload space: never write past the chain's buffer; the frame is reported aborted
                    if (lane == 0) { __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicAdd(a.abort_total, 1); }
                    return;
                }
                if (!IS_SKIP_T(type)) mb_qp = UNI(sr.tmp_i[1]);
            } else if (!a.cabac && type == T_I_16x16 && !(cbp_luma | cbp_chroma) && !UNI((int)s.nnz[24])) {
                // a CAVLC slice is written after the sweep (x264hip_cavlc_write_frame), but cavlc_qp_delta's side effect belongs here: an I_16x16
                // macroblock without any coefficient takes the previous QP (R/encoder/cavlc.c:205-211), which the next macroblock's QP rule reads
                mb_qp = last_qp;
            }
            // x264_macroblock_cache_save's QP rules (R/common/macroblock.c:1244-1272): a macroblock without coefficients has no QP of its own
            if (type == T_I_PCM) { mb_qp = 0; last_dqp = 0; if (lane < 27) s.nnz[lane] = 16; WAVE_SYNC(); }
            else {
                if (type != T_I_16x16 && cbp_luma == 0 && cbp_chroma == 0) mb_qp = last_qp;
                last_dqp = mb_qp - last_qp; last_qp = mb_qp;
            }
        }
        if (cbp_luma == 0 && type != T_I_8x8) t8 = 0;           // x264_macroblock_cache_save, R/common/macroblock.c:1273-1275
        PROF(RD ? 5 : 3);
        LAUNDER();

        // ---- x264_macroblock_cache_save: reconstruction, per-macroblock state, levels ----
        {
            const int r = lane >> 2, x = (lane & 3) * 4;
            *(u32 *)(a.dy + oy + (ptrdiff_t)r * a.sy + x) = *(const u32 *)(s.fd + FDY + r * FD + x);
            if (lane < 32) {
                const int chn = lane >> 4, l = lane & 15, cr = l >> 1, cx4 = (l & 1) * 4;
                *(u32 *)((chn ? a.dv : a.du) + oc + (ptrdiff_t)cr * a.sc + cx4) = *(const u32 *)(s.fd + (chn ? FDV : FDU) + cr * FD + cx4);
            }
        }
        if (lane < 16) {
            a.mv[((size_t)mb * 16 + lane) * 2] = (i16)(intra ? 0 : s.mv4[lane][0]);
            a.mv[((size_t)mb * 16 + lane) * 2 + 1] = (i16)(intra ? 0 : s.mv4[lane][1]);
            if ((lane & 3) == 3) { s.left_mv4[lane >> 2][0] = (i16)(intra ? 0 : s.mv4[lane][0]); s.left_mv4[lane >> 2][1] = (i16)(intra ? 0 : s.mv4[lane][1]); }
            const bool i48 = type == T_I_4x4 || type == T_I_8x8;
            a.i4mode[(size_t)mb * 16 + lane] = i48 ? s.i4c[sw_scan8(lane)] : (signed char)2;
            if (lane == 5 || lane == 7 || lane == 13 || lane == 15)       // what the next macroblock sees to its left
                s.left_i4[lane == 5 ? 0 : lane == 7 ? 1 : lane == 13 ? 2 : 3] = i48 ? s.i4c[sw_scan8(lane)] : (signed char)2;
        }
        if (lane < 4) {
            const signed char rv = (signed char)(is_p || BS ? (intra ? -1 : s.ref8[lane]) : -1);
            a.ref[(size_t)mb * 4 + lane] = rv;
            if (lane & 1) s.left_r8[lane >> 1] = rv;
        }
        if (lane < 27) (a.nnz + 27 * cb)[(size_t)mb * 27 + lane] = IS_SKIP_T(type) ? (u8)0 : s.nnz[lane];
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
        }
        if (lane == 0) {
            const int cbp_dc = a.cabac ? (s.nnz[24] | s.nnz[25] << 1 | s.nnz[26] << 2) : 0;
            a.mb_type[mb] = (signed char)type;
            (a.partition + cb)[mb] = (signed char)(intra || IS_SKIP_T(type) || (BS && type == T_B_DIRECT) ? 16 : part);
            (a.i16mode + cb)[mb] = (signed char)(type == T_I_16x16 ? pred16 : 0);
            (a.chroma_mode + cb)[mb] = (signed char)(intra ? predc : 0);
            (a.qp_out + cb)[mb] = (signed char)mb_qp;
            (a.t8 + cb)[mb] = (signed char)t8;
            (a.cbp + cb)[mb] = (i16)(IS_SKIP_T(type) ? 0 : type == T_I_PCM ? 0x72f : (cbp_dc << 8) | (cbp_chroma << 4) | cbp_luma);
            (a.cost_intra + cb)[mb] = stat_intra; (a.cost_inter + cb)[mb] = stat_inter; (a.cost_alt + cb)[mb] = stat_alt;
        }
        if (a.luma) {   // coefficient levels, masked by what the entropy coder reads (cbp, then nnz); a state without level arrays (the payload is the product) skips them
            const bool coded = !IS_SKIP_T(type) && type != T_I_PCM;
            i16 *ly = (a.luma + 256 * cb) + (size_t)mb * 256, *cac = (a.chroma_ac + 128 * cb) + (size_t)mb * 128;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = lane + 64 * k, blk = i >> 4;
                ly[i] = (coded && ((cbp_luma >> (blk >> 2)) & 1) && s.nnz[blk]) ? (t8 ? s.lv_y8[i] : s.lv_y[i]) : (i16)0;
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int i = lane + 64 * k, blk = i >> 4;
                cac[i] = (coded && cbp_chroma == 2 && s.nnz[16 + blk]) ? s.lv_cac[i] : (i16)0;
            }
            if (lane < 16) (a.luma_dc + 16 * cb)[(size_t)mb * 16 + lane] = (coded && type == T_I_16x16 && s.nnz[24]) ? s.lv_dc[lane] : (i16)0;
            if (lane < 8) (a.chroma_dc + 8 * cb)[(size_t)mb * 8 + lane] = (coded && cbp_chroma && s.nnz[25 + (lane >> 2)]) ? s.lv_cdc[lane] : (i16)0;
        }
        if constexpr (RD) {     // what the next macroblock's entropy coding reads of this one (kept in LDS / registers), and mvd for the row below
            const int cbp_dc = s.nnz[24] | s.nnz[25] << 1 | s.nnz[26] << 2;
            cbp_store = IS_SKIP_T(type) ? 0 : type == T_I_PCM ? 0x72f : (UNI(cbp_dc) << 8) | (cbp_chroma << 4) | cbp_luma;
            const bool keep = !intra && !IS_SKIP_T(type) && !(BS && type == T_B_DIRECT);
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
            }
            left_cbp = cbp_store; left_cpm = intra && type != T_I_PCM ? sw_fix8c(predc) : 0; left_t8 = t8;
            prev_coded = type == T_I_16x16 || (cbp_store & 0x3f);
            intra_before += intra;
            WAVE_SYNC();
        }
        left_type = type;
        left_ref = is_p || BS ? (intra ? -1 : UNI(s.ref8[1])) : -1; left_mvx = intra ? 0 : UNI(s.mv4[3][0]); left_mvy = intra ? 0 : UNI(s.mv4[3][1]);
        PROF(4);
        LAUNDER();
        if constexpr (!RD) {
        // ---- publish: everything this macroblock wrote is visible before the count moves ----
        __threadfence();
        __builtin_amdgcn_wave_barrier();
        row_intra += intra;
        if (lane == 0) __hip_atomic_store(prog + mby, (mbx + 1) | (row_intra << 16), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        PROF(5);
    }
    if (a.prof && lane < 8) {
        long long v = lane == 0 ? pacc[0] : lane == 1 ? pacc[1] : lane == 2 ? pacc[2] : lane == 3 ? pacc[3] : lane == 4 ? pacc[4] : lane == 5 ? pacc[5] : lane == 6 ? pacc[6] : pacc[7];
        a.prof[((size_t)bz * a.mb_h + mby) * 8 + lane] = v;
    }
  }   // rows
    if constexpr (RD) {     // x264_slice_write's end (R/encoder/encoder.c:1269-1273)
        if (rd.write && lane == 0) { cd_encode_flush(cab, rd.i_frame + bz * rd.i_frame_stride); rd.payload_len[bz] = (int)(cab.p - payload0); }
        if constexpr (TD) if (rd.direct_score && lane == 0) { rd.direct_score[2 * bz] = dscore0; rd.direct_score[2 * bz + 1] = dscore1; }
        if constexpr (TD) {
            if (rd.stale && lane == 30) { i16 *sp = rd.stale + (size_t)bz * 8; for (int k = 0; k < 6; k++) sp[k] = sb.stale[k]; }
        } else if (!BS && rd.stale && is_p && lane == 0) {            // (an I slice never touches the motion cache)
            i16 *sp = rd.stale + (size_t)bz * 8;
            sp[0] = (i16)st0r; sp[1] = (i16)st0x; sp[2] = (i16)st0y;
        }
    }
    if (a.nr) {
        if (lane >= 1 && lane < 16 && nr_acc4) atomicAdd(a.nr_sum + (size_t)bz * 128 + lane, (u32)nr_acc4);
        if (lane >= 1 && nr_acc8) atomicAdd(a.nr_sum + (size_t)bz * 128 + 64 + lane, (u32)nr_acc8);
        if (lane == 0 && (nr_n4 | nr_n8)) { atomicAdd(a.nr_count + (size_t)bz * 2, (u32)nr_n4); atomicAdd(a.nr_count + (size_t)bz * 2 + 1, (u32)nr_n8); }
    }
#undef PROF
#undef LAUNDER
}

