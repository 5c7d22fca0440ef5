/* slice_oracle.c -- TEST INFRASTRUCTURE; textually included at the end of frame_oracle.c (it shares
 * that file's table instances and motion-search helpers).
 *
 * CPU restatement of the reference's per-macroblock hot loop over a chain of frames: what
 * x264_slice_write does for every macroblock (R/encoder/encoder.c:1141-1291) --
 *   x264_macroblock_cache_load   R/common/macroblock.c:872-1187  (neighbour state, predictors)
 *   x264_macroblock_analyse      R/encoder/analyse.c:2156-2774   (I and P slices, no RD)
 *   x264_macroblock_encode       R/encoder/macroblock.c:475-790
 *   x264_macroblock_cache_save   R/common/macroblock.c:1208-1372
 * followed per frame by the loop filter, border expansion and half-pel planes (the twins above).
 * Supported: every I, P and B macroblock type x264 core 66 produces (I_16x16 / I_4x4 / I_8x8 / I_PCM, P_SKIP, P 16x16 .. 4x4 over
 * several references, B_SKIP / B_DIRECT (spatial and temporal) / B 16x16, 16x8, 8x16, 8x8 with list 0, list 1 and bi-prediction,
 * weighted bi-prediction), the 8x8 transform, DIA / HEX / UMH / ESA, subme 0..9 (mode-decision RD: cabac_oracle.c / rd_oracle.c /
 * b_oracle.c; RD refinement of vectors and intra modes, sub-8x8 partitions under the RD levels: refine_oracle.c), trellis 1 and 2, psy-rd,
 * adaptive quantisation, --nr, lossless, the CABAC writer.  Not yet: psy-trellis, CAVLC with the RD levels, B pyramid, adaptive B placement.  Pinned against the reference's own functions by
 * tests/test_oracle_slice.py (oracle/ref_slice.c runs them for the same inputs).               */
#include <math.h>
#include <stdio.h>

typedef struct {
    int width, height, n_frames, qp;
    int me_method, me_range, subme, n_refs;
    int inter, intra;
    int transform8x8, fast_pskip, dct_decimate, chroma_me, cabac, mixed_refs;
    int deblock, alpha_c0, beta, chroma_qp_offset, keyint;
    int noise_reduction;
    int mv_range;
    int cqm_preset;                          /* 0 flat, 1 jvt */
} slice_params;

typedef struct {
    int8_t *mb_type, *partition, *sub_partition;
    i16 *mv;
    int8_t *ref;
    i16 *mvr;
    u8 *nnz;
    int8_t *i4mode, *i16mode, *chroma_mode, *qp;
    i16 *cbp;
    int8_t *t8;
    i16 *luma, *luma_dc, *chroma_dc, *chroma_ac;
    u8 *rec_y, *rec_u, *rec_v, *fin_y, *fin_u, *fin_v;
    int32_t *frame_info;
    int64_t *stat;
} slice_out;

/* round 2: the twin of refslice_ext / refslice_out2 (oracle/ref_slice.c) */
typedef struct {
    int trellis;
    float psy_rd, psy_trellis;
    int aq_mode; float aq_strength;
    int write, payload_cap, cabac_init_idc;
    int bframes, weightb, direct_pred;       /* B slices: param.i_bframe (fixed pattern, no pyramid), b_weighted_bipred, i_direct_mv_pred (1 spatial, 2 temporal) */
    const i16 *lowres_mv;                    /* the lookahead's vectors [frame in coding order][list][n_mb][2] or NULL (oracle/ref_slice.c: refslice_ext) */
} slice_ext;
typedef struct {
    u8 *payload;
    int32_t *payload_len, *mb_bits;
    float *qp_offset;
    i16 *mv1; int8_t *ref1;                  /* list 1 */
    int32_t *frame_info2;                    /* [F][4]: display index, i_ref1, kept as reference, 0 */
} slice_out2;

/* mb types / partitions / slice types with the reference's numbering (R/common/macroblock.h:55-102, R/common/common.h:128-134) */
enum { S_I_4x4 = 0, S_I_8x8 = 1, S_I_16x16 = 2, S_I_PCM = 3, S_P_L0 = 4, S_P_8x8 = 5, S_P_SKIP = 6,
       S_B_DIRECT = 7, S_B_L0_L0 = 8, S_B_L0_L1 = 9, S_B_L0_BI = 10, S_B_L1_L0 = 11, S_B_L1_L1 = 12, S_B_L1_BI = 13,
       S_B_BI_L0 = 14, S_B_BI_L1 = 15, S_B_BI_BI = 16, S_B_8x8 = 17, S_B_SKIP = 18 };
enum { S_D_L0_4x4 = 0, S_D_L0_8x4 = 1, S_D_L0_4x8 = 2, S_D_L0_8x8 = 3, S_D_L1_8x8 = 7, S_D_BI_8x8 = 11, S_D_DIRECT_8x8 = 12,
       S_D_8x8 = 13, S_D_16x8 = 14, S_D_8x16 = 15, S_D_16x16 = 16 };
enum { S_SLICE_P = 0, S_SLICE_B = 1, S_SLICE_I = 2 };
#define S_IS_SKIP(t) ((t) == S_P_SKIP || (t) == S_B_SKIP)
#define S_IS_DIRECT(t) ((t) == S_B_DIRECT)
enum { NB_LEFT = 1, NB_TOP = 2, NB_TOPRIGHT = 4, NB_TOPLEFT = 8 };
#define S_COST_MAX (1 << 28)
#define S_IS_INTRA(t) ((t) >= 0 && (t) <= S_I_PCM)

static const int s_lambda_tab[52] = {    /* R/encoder/analyse.c:140-149 */
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
    6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};
static const int s_lambda2_tab[52] = {   /* :151-159 */
    14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1161, 1462, 1843, 2322, 2925,
    3686, 4644, 5851, 7372, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628, 117964,
    148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436};
static const u8 s_chroma_qp[52] = {      /* i_chroma_qp_table, R/common/macroblock.h */
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
static const u8 s_z2r[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};   /* block_idx_xy_1d */
static const int8_t s_fix4[13] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 2, 2, 2};            /* x264_mb_pred_mode4x4_fix, index mode+1 */
static const u8 s_fix16[7] = {0, 1, 2, 3, 2, 2, 2}, s_fix8c[7] = {0, 1, 2, 3, 0, 0, 0};
static const u8 s_pred4_nb[12] = {NB_TOP, NB_LEFT, NB_LEFT | NB_TOP, NB_TOP | NB_TOPRIGHT, NB_LEFT | NB_TOPLEFT | NB_TOP,
                                  NB_LEFT | NB_TOPLEFT | NB_TOP, NB_LEFT | NB_TOPLEFT | NB_TOP, NB_TOP | NB_TOPRIGHT, NB_LEFT,
                                  NB_LEFT, NB_TOP, 0};
static int s_ue_size(int v) { int n = 0; v++; while (v >> (n + 1)) n++; return 2 * n + 1; }   /* bs_size_ue */
static int s_te_size(int x, int v) { return x == 1 ? 1 : x > 1 ? s_ue_size(v) : 0; }          /* bs_size_te */

static x264hip_predict_t s_p16[7], s_p8c[7], s_p4[12];
static x264hip_predict8x8_t s_p8[12];
static x264hip_predict_8x8_filter_t s_p8filter;
void x264o_cqm(int preset, int cat, int qp, int is8x8, u16 *mf, u16 *bias, int *dequant);
void x264o_cqm_unquant(int preset, int cat, int qp, int is8x8, int *unq);
void x264o_predict_16x16_init(x264hip_predict_t pf[7]);
void x264o_predict_4x4_init(x264hip_predict_t pf[12]);
#ifdef X264O_USE_REF
void x264_predict_16x16_init(int, x264hip_predict_t pf[7]);
void x264_predict_4x4_init(int, x264hip_predict_t pf[12]);
#endif
static i16 *s_cost_mv[52];
static void s_setup(void)
{
    static int done;
    init();
    if (done) return;
#ifdef X264O_USE_REF
    x264_predict_16x16_init(0, s_p16); x264_predict_8x8c_init(0, s_p8c); x264_predict_4x4_init(0, s_p4);
    x264_predict_8x8_init(0, s_p8, &s_p8filter);
#else
    x264o_predict_16x16_init(s_p16); x264o_predict_8x8c_init(s_p8c); x264o_predict_4x4_init(s_p4);
    x264o_predict_8x8_init(s_p8, &s_p8filter);
#endif
    done = 1;
}
/* p_cost_mv, x264_mb_analyse_load_costs (R/encoder/analyse.c:182-198); log2f is the reference's macro (:40) */
static const i16 *s_load_cost_mv(int qp)
{
    if (!s_cost_mv[qp]) {
        i16 *t = malloc((4 * 4 * 2048 + 1) * sizeof(i16));
        t += 2 * 4 * 2048;
        for (int i = 0; i <= 2 * 4 * 2048; i++)
            t[-i] = t[i] = (i16)(s_lambda_tab[qp] * (((float)log((double)(i + 1))) / (log((double)2)) * 2 + 0.718f + !!i) + .5f);
        s_cost_mv[qp] = t;
    }
    return s_cost_mv[qp];
}

/* the table of one QP from -span (test access: tests/test_cpu_abi_and_shard.py compares the product's C table with it) */
const i16 *x264o_cost_mv_row(int qp) { return s_load_cost_mv(qp) - 2 * 4 * 2048; }

typedef struct {
    u8 *alloc[4];
    u8 *plane[3], *filt[4];
    int8_t *mb_type;          /* as stored by cache_save (x264_mb_type_fix applied: I_8x8 -> I_4x4) */
    i16 *mv;                  /* [n][16][2], 4x4 blocks in raster order inside the macroblock */
    int8_t *ref;              /* [n][4] */
    i16 *mv1; int8_t *ref1;   /* list 1 (B slices) */
    int poc, n_ref0, ref_poc[16], inv_ref_poc[16];
    int kept;                 /* b_kept_as_ref */
} sframe;

typedef struct {                             /* x264_cabac_t, R/common/cabac.h:27-46 */
    int low, range, queue, outstanding;
    u8 *p, *start, *end;
    int f8;                                  /* f8_bits_encoded */
    int i_frame;                             /* frames coded before this one (x264_cabac_encode_flush's padding bit) */
    u8 state[460];
} o_cabac;

typedef struct {
    const slice_params *p;
    int mb_w, mb_h, n, sy, sc, w16, h16;
    sframe *fenc, *fdec, *fref[16];
    int n_ref, slice_type, qp, qpc, lambda, lambda2;
    const i16 *cost_mv;
    int ref_cost[16];
    u8 *nnz; int8_t *i4mode, *t8; i16 *mvr;
    int intra_count;
    int64_t stat_intra, stat_inter, stat_n;
    u16 mf4[4][16], b4[4][16], mf8[2][64], b8[2][64];
    int dq4[4][6][16], dq8[2][6][64];
    slice_out *o;
    int f;
    /* --nr: h->nr_residual_sum / nr_count / nr_offset, [0] 4x4, [1] 8x8 (R/common/common.h:308-310) */
    int lossless;                            /* h->mb.b_lossless: constant QP 0 (R/encoder/encoder.c:401-421) */
    int b_nr;                                /* h->mb.b_noise_reduction: 0 while a macroblock is analysed -- the RD levels' trial encodes neither denoise nor
                                                count (analyse.c:237) -- param.analyse.i_noise_reduction != 0 for the final encode (:2769) */
    /* round 2: RD levels, trellis, the entropy coder, per-macroblock QP */
    const slice_ext *e;
    const i16 *lowres_mv[2];                 /* this frame's fenc->lowres_mvs of list 0 / 1 towards reference 0, or NULL (none / marked 0x7fff) */
    slice_out2 *o2;
    int mbrd, psy_rd, trellis, b_trellis;    /* a->i_mbrd, h->mb.i_psy_rd, param i_trellis, h->mb.b_trellis (what the encode functions obey now) */
    int chroma_qp_offset;                    /* after x264_validate_parameters' psy adjustment */
    int frame_qp, qp_min, qp_max;            /* rc->qpm of the frame; param.rc.i_qp_min / max */
    float f_qpm, *aq_offset;                 /* rc->f_qpm; fenc->f_qp_offset[n] (adaptive quantisation) */
    int last_qp, last_dqp, prev_mb;          /* h->mb.i_last_qp / i_last_dqp / i_mb_prev_xy */
    i16 *cbp;                                /* h->mb.cbp[n] */
    int8_t *chroma_pm;                       /* h->mb.chroma_pred_mode[n] ("fixed" numbering, DC for anything not intra) */
    i16 *mvd;                                /* h->mb.mvd[0]: [n][16][2] */
    int8_t *qp_mb;                           /* h->mb.qp[n] */
    int unq4[4][16], unq8[2][64];            /* h->unquant4_mf / unquant8_mf at the current QPs */
    u8 zz4[16], zz8[64]; int w4z[16], w8z[64];   /* x264_zigzag_scan4/8[0], x264_dct4/8_weight2_zigzag[0] */
    u8 *bsbuf; int i_skip;
    o_cabac cb;                              /* h->cabac */
    uint32_t nr_sum[2][64], nr_count[2];
    uint16_t nr_offset[2][64];
    /* B slices */
    sframe *fref1[2]; int n_ref1;            /* h->fref1 / h->i_ref1 */
    i16 *mvr1, *mvd1;                        /* h->mb.mvr[1][0], h->mb.mvd[1] */
    u8 *skipbp;                              /* h->mb.skipbp */
    int direct_spatial;                      /* sh.b_direct_spatial_mv_pred */
    int bipred_weight[16][2], dist_scale[16][2];   /* h->mb.bipred_weight / dist_scale_factor (x264_macroblock_bipred_init) */
    int8_t map_col_store[18];                /* h->mb.map_col_to_list0 with its -1 / -2 entries */
    int ref_cost1[2];                        /* a->p_cost_ref1 */
    /* h->mb.cache.ref / mv are never cleared between macroblocks: block 12's entry of either list, as the previous macroblock (of
     * whatever frame) left it, is what x264_mb_predict_mv_ref16x16 reads as its "direct" candidate when temporal direct prediction
     * gave up before writing it */
    int8_t stale_ref[2]; i16 stale_mv[2][2];
    /* ... and so are the macroblock's own entries of h->mb.cache.non_zero_count and h->mb.cache.mvd: x264_macroblock_cache_load only
     * rewrites the neighbours' entries.  A full trial encode writes all of its own before anything reads them, but the PARTIAL trials of
     * the RD refinement / sub-8x8 RD (x264_rd_cost_part: one 8x8 block encoded and priced) read the blocks beside theirs -- whatever the
     * last trial, or the previous macroblock (of whatever frame), left there (the reference's own FIXME, analyse.c:1976-1977) */
    u8 carry_nnz[27]; i16 carry_mvd[2][16][2];
} ssl;

typedef struct {
    int mbx, mby, mb, nb, nb4[16], nb8[4];
    int type_left, type_top, type_topleft, type_topright;
    u8 fenc[24 * FENC], fdec[27 * FDEC];
    u8 *fe[3], *fd[3];
    int8_t i4c[48];                      /* intra4x4_pred_mode cache, x264_scan8 layout */
    u8 nnz[27];
    i16 luma4[16][16], luma8[4][64], dc16[16], cdc[2][4], cac[8][16];
    int cbp_luma, cbp_chroma, type, t8, i16mode, chroma_mode, skip_mc;
    int mvx, mvy, ref;                   /* the 16x16 vector */
    int partition;                       /* D_16x16 / D_16x8 / D_8x16 / D_8x8 */
    i16 mv4[16][2];                      /* final vectors per 4x4 block (raster inside the macroblock) and references per 8x8 */
    int8_t ref8[4], sub[4];
    int8_t cref[48];                     /* h->mb.cache.ref / mv for list 0, x264_scan8 layout */
    i16 cmv[48][2];
    i16 l0mvc[16][5][2];                 /* a->l0.mvc[ref][0 = 16x16, 1..4 = 8x8 blocks] */
    i16 pskip_mv[2];
    /* analysis */
    int satd_i16, satd_i8, satd_i4, satd_chroma, fast_intra, pred16, pred8[4], pred4[16], predc;
    int satd_i16_dir[7], satd_c_dir[4], satd_i8_dir[12][4];   /* a->i_satd_i16x16_dir[mode], i_satd_i8x8chroma_dir[list position], i_satd_i8x8_dir[mode][block] */
    int cbp_i8_rd;                       /* a->i_cbp_i8x8_luma (x264_intra_rd, analyse.c:869) */
    u8 i4_fdec[256], i8_fdec[256], i4_nnz[16], i8_nnz[16];
    int i4_cbp, i8_cbp;
    /* round 2 */
    int qp;                              /* h->mb.i_qp */
    int skip_intra;                      /* h->mb.i_skip_intra */
    i16 i4_dct[16][16], i8_dct[4][64];   /* h->mb.pic.i4x4_dct_buf / i8x8_dct_buf (i_skip_intra == 2) */
    int cbp_left, cbp_top, cpm_left, cpm_top, nb_t8;   /* cache.i_cbp_left / top (-1: none), neighbours' chroma modes, i_neighbour_transform_size */
    u8 nz_l[4], nz_t[4], nz_lc[2][2], nz_tc[2][2];      /* neighbours' non_zero_count next to this macroblock (0x80: none) */
    i16 cmvd[48][2];                     /* h->mb.cache.mvd[0], x264_scan8 layout */
    int fenc_satd[4][4], fenc_sa8d[2][2], fenc_satd_sum, fenc_sa8d_sum;   /* h->mb.pic.fenc_satd ... (psy-RD) */
    /* B slices: list 1 of the caches, the skip flags of direct blocks, the direct prediction, the final list-1 vectors */
    int8_t cref1[48]; i16 cmv1[48][2], cmvd1[48][2];
    int8_t cskip[48];                    /* h->mb.cache.skip */
    int8_t direct_ref[2][4]; i16 direct_mv[2][16][2];   /* h->mb.cache.direct_ref / direct_mv (the 16 blocks in raster order) */
    i16 mv4_1[16][2]; int8_t ref8_1[4];
} smb;
#define CREF(m_, l_) ((l_) ? (m_)->cref1 : (m_)->cref)
#define CMV(m_, l_) ((l_) ? (m_)->cmv1 : (m_)->cmv)
#define CMVD(m_, l_) ((l_) ? (m_)->cmvd1 : (m_)->cmvd)

/* what x264_mb_analysis_t keeps of the P analysis (R/encoder/analyse.c:42-137) */
typedef struct { int mvx, mvy, cost, cost_mv, ref, ref_cost; i16 mvp[2]; } pme;
typedef struct { int mvx, mvy, cost; i16 mvp[2]; } sub_me;
struct banalysis;
typedef struct {
    pme me16, me8[4], me16x8[2], me8x16[2];
    sub_me me4[4][4], me84[4][2], me48[4][2];
    int sub[4];
    int cost_sub[4][3];                  /* a->l0.i_cost4x4 / i_cost8x4 / i_cost4x8 of every 8x8 block (COST_MAX: not searched) */
    int cost8x8, cost16x8, cost8x16, rd16;
    struct banalysis *B;                 /* the B-slice half of x264_mb_analysis_t (b_oracle.c) */
} panalysis;

static int s_scan8(int i)
{   /* x264_scan8, R/common/common.h:196-238 */
    if (i < 16) return 4 + 1 * 8 + (blk_x[i] >> 2) + 8 * (blk_y[i] >> 2);
    if (i < 20) return 1 + 1 * 8 + ((i - 16) & 1) + 8 * ((i - 16) >> 1);
    if (i < 24) return 1 + 4 * 8 + ((i - 20) & 1) + 8 * ((i - 20) >> 1);
    return 4 + 5 * 8 + (i - 24);
}

static sframe *sframe_new(const ssl *S)
{
    sframe *f = calloc(1, sizeof(*f));
    size_t ly = (size_t)S->sy * (S->h16 + 2 * 32 + 2), lc = (size_t)S->sc * (S->h16 / 2 + 2 * 32 + 2);
    f->alloc[0] = calloc(4, ly); f->alloc[1] = calloc(1, lc); f->alloc[2] = calloc(1, lc);
    for (int i = 0; i < 4; i++) f->filt[i] = f->alloc[0] + i * ly + 32 * S->sy + 32;
    f->plane[0] = f->filt[0];
    f->plane[1] = f->alloc[1] + 16 * S->sc + 16; f->plane[2] = f->alloc[2] + 16 * S->sc + 16;
    f->mb_type = calloc(S->n, 1); f->mv = calloc(S->n * 32, sizeof(i16)); f->ref = calloc(S->n * 4, 1);
    f->mv1 = calloc(S->n * 32, sizeof(i16)); f->ref1 = calloc(S->n * 4, 1);
    return f;
}
static void sframe_free(sframe *f)
{
    if (!f) return;
    for (int i = 0; i < 3; i++) free(f->alloc[i]);
    free(f->mb_type); free(f->mv); free(f->ref); free(f->mv1); free(f->ref1); free(f);
}

/* ------------------------------------------------------------------ neighbour motion state
 * what x264_macroblock_cache_load puts in h->mb.cache.ref / mv around the macroblock
 * (R/common/macroblock.c:1040-1128): -2 = not available, -1 = intra.                       */
static int nb_ref_mv(const ssl *S, const smb *m, int which, i16 mv[2])
{
    int mbx = m->mbx, mby = m->mby, blk;
    mv[0] = mv[1] = 0;
    switch (which) {
    case 0: if (!(m->nb & NB_LEFT)) return -2; mbx--; blk = 3; break;                 /* A: left of block 0 */
    case 1: if (!(m->nb & NB_TOP)) return -2; mby--; blk = 12; break;                 /* B: above block 0 */
    case 2: if (!(m->nb & NB_TOPRIGHT)) return -2; mbx++; mby--; blk = 12; break;     /* C: above-right of the macroblock */
    default: if (!(m->nb & NB_TOPLEFT)) return -2; mbx--; mby--; blk = 15; break;     /* D: above-left */
    }
    int o = mby * S->mb_w + mbx;
    mv[0] = S->fdec->mv[(o * 16 + blk) * 2]; mv[1] = S->fdec->mv[(o * 16 + blk) * 2 + 1];
    return S->fdec->ref[o * 4 + (blk >> 3) * 2 + ((blk & 3) >> 1)];
}
static int s_median(int a, int b, int c) { int mx = a > b ? a : b, mn = a < b ? a : b; return c > mx ? mx : c < mn ? mn : c; }

/* x264_mb_predict_mv_16x16, R/common/macroblock.c:90-128 */
static void predict_mv_16x16(const ssl *S, const smb *m, int i_ref, i16 mvp[2])
{
    i16 a[2], b[2], c[2];
    int ra = nb_ref_mv(S, m, 0, a), rb = nb_ref_mv(S, m, 1, b), rc = nb_ref_mv(S, m, 2, c), cnt;
    if (rc == -2) rc = nb_ref_mv(S, m, 3, c);
    cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
    if (cnt > 1) { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
    else if (cnt == 1) { const i16 *s = ra == i_ref ? a : rb == i_ref ? b : c; mvp[0] = s[0]; mvp[1] = s[1]; }
    else if (rb == -2 && rc == -2 && ra != -2) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
}
/* x264_mb_predict_mv_pskip, :131-149 */
static void predict_mv_pskip(const ssl *S, const smb *m, i16 mv[2])
{
    i16 a[2], b[2];
    int ra = nb_ref_mv(S, m, 0, a), rb = nb_ref_mv(S, m, 1, b);
    if (ra == -2 || rb == -2 || !(ra | a[0] | a[1]) || !(rb | b[0] | b[1])) mv[0] = mv[1] = 0;
    else predict_mv_16x16(S, m, 0, mv);
}
/* x264_mb_predict_mv_ref16x16, :376-437 (P slice; no lookahead vectors in this configuration) */
static int predict_mv_ref16x16(const ssl *S, const smb *m, int i_ref, i16 mvc[8][2])
{
    const i16 *mvr = S->mvr + (size_t)i_ref * S->n * 2;
    const int8_t *type = S->fdec->mb_type;
    int i = 0, top = m->mb - S->mb_w;
    if (i_ref == 0 && S->lowres_mv[0]) {                   /* the lookahead's vector, twice (half resolution -> full), :393-398 */
        mvc[i][0] = (i16)(u16)(S->lowres_mv[0][2 * m->mb] << 1); mvc[i][1] = (i16)(u16)(S->lowres_mv[0][2 * m->mb + 1] << 1); i++;
    }
#define SET(o) do { mvc[i][0] = mvr[2 * (o)]; mvc[i][1] = mvr[2 * (o) + 1]; i++; } while (0)
    if ((m->nb & NB_LEFT) && type[m->mb - 1] != S_P_SKIP) SET(m->mb - 1);
    if (m->nb & NB_TOP) {
        if (type[top] != S_P_SKIP) SET(top);
        if ((m->nb & NB_TOPLEFT) && type[top - 1] != S_P_SKIP) SET(top - 1);
        if (m->mbx < S->mb_w - 1 && type[top + 1] != S_P_SKIP) SET(top + 1);
    }
#undef SET
    if (S->fref[0]->n_ref0 > 0) {
        const sframe *l0 = S->fref[0];
        for (int k = 0; k < 3; k++) {
            int dx = k == 1, dy = k == 2;
            if ((dx && m->mbx >= S->mb_w - 1) || (dy && m->mby >= S->mb_h - 1)) continue;
            int o = m->mb + dx + dy * S->mb_w, ref_col = l0->ref[o * 4];
            if (ref_col >= 0) {
                int scale = (S->fdec->poc - S->fdec->ref_poc[i_ref]) * l0->inv_ref_poc[ref_col];
                mvc[i][0] = (i16)((l0->mv[o * 32] * scale + 128) >> 8);
                mvc[i][1] = (i16)((l0->mv[o * 32 + 1] * scale + 128) >> 8);
                i++;
            }
        }
    }
    return i;
}

/* ------------------------------------------------------------------ encode pieces
 * x264_mb_encode_i4x4 / _i8x8 / _i16x16 / _8x8_chroma, R/encoder/macroblock.c:116-363 (no trellis, not lossless) */
/* ---- lossless (qp 0): predictive lossless intra prediction and the zigzag "transforms" (R/encoder/macroblock.c:405-470, R/common/dct.c:564-606) ---- */
static const u8 s_z2r_yx[16] = {0, 4, 1, 5, 8, 12, 9, 13, 2, 6, 3, 7, 10, 14, 11, 15};  /* block_idx_yx_1d */
static int any_nz(const i16 *l, int n) { for (int i = 0; i < n; i++) if (l[i]) return 1; return 0; }
/* the source plane (with its borders) at pixel (x, y) of this macroblock: h->mb.pic.p_fenc_plane */
static const u8 *ll_src(const ssl *S, const smb *m, int pl, int x, int y)
{
    const int w = pl ? 8 : 16, st = pl ? S->sc : S->sy;
    return S->fenc->plane[pl] + (w * m->mby + y) * st + w * m->mbx + x;
}
static void ll_copy(u8 *dst, const u8 *src, int stride, int w, int h) { for (int y = 0; y < h; y++) memcpy(dst + y * FDEC, src + y * stride, w); }
/* x264_predict_lossless_*: vertical / horizontal prediction copy the source shifted by one row / column, the other modes predict as usual */
static void pred_16x16(const ssl *S, smb *m, int mode)
{
    if (S->lossless && mode == 0) ll_copy(m->fd[0], ll_src(S, m, 0, 0, -1), S->sy, 16, 16);
    else if (S->lossless && mode == 1) ll_copy(m->fd[0], ll_src(S, m, 0, -1, 0), S->sy, 16, 16);
    else s_p16[mode](m->fd[0]);
}
static void pred_chroma(const ssl *S, smb *m, int mode)     /* slots: DC 0, H 1, V 2, P 3 */
{
    for (int pl = 1; pl < 3; pl++) {
        if (S->lossless && mode == 2) ll_copy(m->fd[pl], ll_src(S, m, pl, 0, -1), S->sc, 8, 8);
        else if (S->lossless && mode == 1) ll_copy(m->fd[pl], ll_src(S, m, pl, -1, 0), S->sc, 8, 8);
        else s_p8c[mode](m->fd[pl]);
    }
}
static void pred_8x8(const ssl *S, smb *m, int idx, int mode, u8 *edge)
{
    u8 *dst = m->fd[0] + 8 * (idx & 1) + 8 * (idx >> 1) * FDEC;
    if (S->lossless && mode == 0) ll_copy(dst, ll_src(S, m, 0, 8 * (idx & 1), 8 * (idx >> 1) - 1), S->sy, 8, 8);
    else if (S->lossless && mode == 1) ll_copy(dst, ll_src(S, m, 0, 8 * (idx & 1) - 1, 8 * (idx >> 1)), S->sy, 8, 8);
    else s_p8[mode](dst, edge);
}
static void pred_4x4(const ssl *S, smb *m, int idx, int mode)
{
    u8 *dst = m->fd[0] + blk_x[idx] + blk_y[idx] * FDEC;
    if (S->lossless && mode == 0) ll_copy(dst, ll_src(S, m, 0, blk_x[idx], blk_y[idx] - 1), S->sy, 4, 4);
    else if (S->lossless && mode == 1) ll_copy(dst, ll_src(S, m, 0, blk_x[idx] - 1, blk_y[idx]), S->sy, 4, 4);
    else s_p4[mode](dst);
}

/* x264_quant_4x4 / x264_quant_8x8 (R/encoder/macroblock.c:87-103) and the DC calls: plain dead-zone quantisation or trellis */
static int trellis_quant(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                         int cat, int lambda2, int b_ac, int dc, int n_coef);
static const int s_trellis_lambda2[2][52];
static int q4(const ssl *S, i16 d[4][4], int qcat, int ctxcat, int b_intra, int qp)
{
    if (S->b_trellis)
        return trellis_quant(S, &d[0][0], S->mf4[qcat], S->unq4[qcat], S->w4z, S->zz4, ctxcat, s_trellis_lambda2[b_intra][qp],
                             ctxcat == 1 || ctxcat == 4, 0, 16);
    return quantf.quant_4x4(d, (u16 *)S->mf4[qcat], (u16 *)S->b4[qcat]);
}
static int q8(const ssl *S, i16 d[8][8], int qcat, int b_intra, int qp)
{
    if (S->b_trellis)
        return trellis_quant(S, &d[0][0], S->mf8[qcat], S->unq8[qcat], S->w8z, S->zz8, 5, s_trellis_lambda2[b_intra][qp], 0, 0, 64);
    return quantf.quant_8x8(d, (u16 *)S->mf8[qcat], (u16 *)S->b8[qcat]);
}
static int qdc(const ssl *S, i16 *d, int qcat, int ctxcat, int b_intra, int qp)
{   /* x264_quant_dc_trellis (rdo.c:632-639) or quant_4x4_dc / quant_2x2_dc */
    static const u8 zz2[4] = {0, 1, 2, 3};
    if (S->b_trellis)
        return trellis_quant(S, d, S->mf4[qcat], S->unq4[qcat], 0, ctxcat == 3 ? zz2 : S->zz4, ctxcat, s_trellis_lambda2[b_intra][qp], 0, 1, ctxcat == 3 ? 4 : 16);
    if (ctxcat == 3) return quantf.quant_2x2_dc((i16 (*)[2])d, S->mf4[qcat][0] >> 1, S->b4[qcat][0] << 1);
    return quantf.quant_4x4_dc((i16 (*)[4])d, S->mf4[qcat][0] >> 1, S->b4[qcat][0] << 1);
}

static void enc_i4x4(ssl *S, smb *m, int idx)
{
    if (S->lossless) {                                   /* macroblock.c:123-130 */
        u8 *src = m->fe[0] + blk_x[idx] + blk_y[idx] * FENC, *dst = m->fd[0] + blk_x[idx] + blk_y[idx] * FDEC;
        zigf[0].sub_4x4(m->luma4[idx], src, dst);
        int nz = any_nz(m->luma4[idx], 16);
        m->nnz[idx] = nz;
        m->cbp_luma |= nz << (idx >> 2);
        return;
    }
    i16 d[4][4];
    u8 *src = m->fe[0] + blk_x[idx] + blk_y[idx] * FENC, *dst = m->fd[0] + blk_x[idx] + blk_y[idx] * FDEC;
    dctf.sub4x4_dct(d, src, dst);
    int nz = q4(S, d, 0, 2, 1, S->qp);
    m->nnz[idx] = nz;
    if (nz) {
        m->cbp_luma |= 1 << (idx >> 2);
        zigf[0].scan_4x4(m->luma4[idx], d);
        quantf.dequant_4x4(d, (int (*)[4][4])S->dq4[0], S->qp);
        dctf.add4x4_idct(dst, d);
    }
}
static void enc_i8x8(ssl *S, smb *m, int idx)
{
    i16 d[8][8];
    int x = 8 * (idx & 1), y = 8 * (idx >> 1);
    u8 *src = m->fe[0] + x + y * FENC, *dst = m->fd[0] + x + y * FDEC;
    if (S->lossless) {                                   /* macroblock.c:160-167 */
        zigf[0].sub_8x8(m->luma8[idx], src, dst);
        int nz = any_nz(m->luma8[idx], 64);
        for (int k = 0; k < 4; k++) m->nnz[4 * idx + k] = nz;
        m->cbp_luma |= nz << idx;
        return;
    }
    dctf.sub8x8_dct8(d, src, dst);
    int nz = q8(S, d, 0, 1, S->qp);
    if (nz) {
        m->cbp_luma |= 1 << idx;
        zigf[0].scan_8x8(m->luma8[idx], d);
        quantf.dequant_8x8(d, (int (*)[8][8])S->dq8[0], S->qp);
        dctf.add8x8_idct8(dst, d);
    }
    for (int k = 0; k < 4; k++) m->nnz[4 * idx + k] = !!nz;
}
static void enc_i16x16(ssl *S, smb *m)
{
    i16 d[16][4][4], dc[4][4];
    int b_decimate = S->slice_type == S_SLICE_B || (S->p->dct_decimate && S->slice_type == S_SLICE_P), score = b_decimate ? 0 : 9, nz;   /* macroblock.c:193 */
    if (S->lossless) {                                   /* macroblock.c:196-213 */
        for (int i = 0; i < 16; i++) {
            zigf[0].sub_4x4(m->luma4[i], m->fe[0] + blk_x[i] + blk_y[i] * FENC, m->fd[0] + blk_x[i] + blk_y[i] * FDEC);
            dc[0][s_z2r_yx[i]] = m->luma4[i][0];
            m->luma4[i][0] = 0;
            nz = any_nz(m->luma4[i], 16);
            m->nnz[i] = nz;
            m->cbp_luma |= nz;
        }
        m->cbp_luma *= 0xf;
        m->nnz[24] = any_nz(&dc[0][0], 16);
        zigf[0].scan_4x4(m->dc16, dc);
        return;
    }
    dctf.sub16x16_dct(d, m->fe[0], m->fd[0]);
    for (int i = 0; i < 16; i++) {
        dc[0][s_z2r[i]] = d[i][0][0];
        d[i][0][0] = 0;
        nz = q4(S, d[i], 0, 1, 1, S->qp);
        m->nnz[i] = nz;
        if (nz) {
            zigf[0].scan_4x4(m->luma4[i], d[i]);
            quantf.dequant_4x4(d[i], (int (*)[4][4])S->dq4[0], S->qp);
            if (score < 6) score += quantf.decimate_score15(m->luma4[i]);
            m->cbp_luma = 0xf;
        }
    }
    if (score < 6) { m->cbp_luma = 0; memset(m->nnz, 0, 16); }
    dctf.dct4x4dc(dc);
    nz = qdc(S, &dc[0][0], 0, 0, 1, S->qp);
    m->nnz[24] = nz;
    if (nz) {
        zigf[0].scan_4x4(m->dc16, dc);
        dctf.idct4x4dc(dc);
        quantf.dequant_4x4_dc(dc, (int (*)[4][4])S->dq4[0], S->qp);
        if (m->cbp_luma)
            for (int i = 0; i < 16; i++) d[i][0][0] = dc[0][s_z2r[i]];
    }
    if (m->cbp_luma) dctf.add16x16_idct(m->fd[0], d);
    else if (nz) dctf.add16x16_idct_dc(m->fd[0], dc);
}
static void enc_chroma(ssl *S, smb *m, int b_inter)
{
    int cat = 2 + b_inter, qpc = S->qpc, b_decimate = b_inter && (S->slice_type == S_SLICE_B || S->p->dct_decimate);   /* macroblock.c:275 */
    int (*dq)[4][4] = (int (*)[4][4])S->dq4[cat];
    m->cbp_chroma = 0;
    for (int ch = 0; ch < 2; ch++) {
        u8 *ps = m->fe[1 + ch], *pd = m->fd[1 + ch];
        i16 d4[4][4][4], d2[2][2];
        int score = 0, nz_ac = 0;
        if (S->lossless) {                               /* macroblock.c:288-303 */
            for (int i = 0; i < 4; i++) {
                i16 *lv = m->cac[4 * ch + i];
                zigf[0].sub_4x4(lv, ps + 4 * (i & 1) + 4 * (i >> 1) * FENC, pd + 4 * (i & 1) + 4 * (i >> 1) * FDEC);
                m->cdc[ch][i] = lv[0];
                lv[0] = 0;
                int nz = any_nz(lv, 16);
                m->nnz[16 + 4 * ch + i] = nz;
                m->cbp_chroma |= nz;
            }
            m->nnz[25 + ch] = any_nz(m->cdc[ch], 4);
            continue;
        }
        dctf.sub8x8_dct(d4, ps, pd);
        {   /* dct2x2dc, :73-85 */
            int a = d4[0][0][0] + d4[1][0][0], b = d4[2][0][0] + d4[3][0][0];
            int c = d4[0][0][0] - d4[1][0][0], d = d4[2][0][0] - d4[3][0][0];
            d2[0][0] = a + b; d2[1][0] = c + d; d2[0][1] = a - b; d2[1][1] = c - d;
            d4[0][0][0] = d4[1][0][0] = d4[2][0][0] = d4[3][0][0] = 0;
        }
        for (int i = 0; i < 4; i++) {
            int nz = q4(S, d4[i], cat, 4, !b_inter, qpc);
            m->nnz[16 + 4 * ch + i] = nz;
            if (nz) {
                nz_ac = 1;
                zigf[0].scan_4x4(m->cac[4 * ch + i], d4[i]);
                quantf.dequant_4x4(d4[i], dq, qpc);
                if (b_decimate) score += quantf.decimate_score15(m->cac[4 * ch + i]);
            }
        }
        int nz_dc = qdc(S, &d2[0][0], cat, 3, !b_inter, qpc);
        m->nnz[25 + ch] = nz_dc;
        /* IDCT_DEQUANT_START, :40-51 */
        int e0 = d2[0][0] + d2[0][1], e1 = d2[1][0] + d2[1][1], e2 = d2[0][0] - d2[0][1], e3 = d2[1][0] - d2[1][1];
        int dmf = dq[qpc % 6][0][0], qbits = qpc / 6 - 5;
        if (qbits > 0) { dmf <<= qbits; qbits = 0; }
        if ((b_decimate && score < 7) || !nz_ac) {
            memset(m->nnz + 16 + 4 * ch, 0, 4);
            if (!nz_dc) continue;
            m->cdc[ch][0] = d2[0][0]; m->cdc[ch][1] = d2[1][0]; m->cdc[ch][2] = d2[0][1]; m->cdc[ch][3] = d2[1][1];
            i16 dd[2][2];
            dd[0][0] = (e0 + e1) * dmf >> -qbits; dd[0][1] = (e0 - e1) * dmf >> -qbits;
            dd[1][0] = (e2 + e3) * dmf >> -qbits; dd[1][1] = (e2 - e3) * dmf >> -qbits;
            dctf.add8x8_idct_dc(pd, dd);
        } else {
            m->cbp_chroma = 1;
            if (nz_dc) {
                m->cdc[ch][0] = d2[0][0]; m->cdc[ch][1] = d2[1][0]; m->cdc[ch][2] = d2[0][1]; m->cdc[ch][3] = d2[1][1];
                d4[0][0][0] = (e0 + e1) * dmf >> -qbits; d4[1][0][0] = (e0 - e1) * dmf >> -qbits;
                d4[2][0][0] = (e2 + e3) * dmf >> -qbits; d4[3][0][0] = (e2 - e3) * dmf >> -qbits;
            }
            dctf.add8x8_idct(pd, d4);
        }
    }
    if (m->cbp_chroma) m->cbp_chroma = 2;
    else if (m->nnz[25] | m->nnz[26]) m->cbp_chroma = 1;
}
/* inter luma, R/encoder/macroblock.c:596-768 */
static void enc_inter_luma(ssl *S, smb *m)
{
    int b_decimate = S->slice_type == S_SLICE_B || S->p->dct_decimate, decimate_mb = 0;   /* macroblock.c:479 */
    if (S->lossless) {                                   /* macroblock.c:602-626 (the 8x8 transform is never chosen for inter here, analyse.c:2111) */
        for (int i = 0; i < 16; i++) {
            zigf[0].sub_4x4(m->luma4[i], m->fe[0] + blk_x[i] + blk_y[i] * FENC, m->fd[0] + blk_x[i] + blk_y[i] * FDEC);
            int nz = any_nz(m->luma4[i], 16);
            m->nnz[i] = nz;
            m->cbp_luma |= nz << (i >> 2);
        }
        return;
    }
    if (m->t8) {
        i16 d8[4][8][8];
        b_decimate &= !S->b_trellis;                     /* "8x8 trellis is inherently optimal decimation", macroblock.c:630 */
        dctf.sub16x16_dct8(d8, m->fe[0], m->fd[0]);
        if (S->b_nr && !S->lossless) S->nr_count[1] += 4;
        for (int idx = 0; idx < 4; idx++) {
            if (S->b_nr && !S->lossless) quantf.denoise_dct(&d8[idx][0][0], S->nr_sum[1], S->nr_offset[1], 64);   /* macroblock.c:636 */
            int nz = q8(S, d8[idx], 1, 0, S->qp);
            if (nz) {
                zigf[0].scan_8x8(m->luma8[idx], d8[idx]);
                if (b_decimate) {
                    int s = quantf.decimate_score64(m->luma8[idx]);
                    decimate_mb += s;
                    if (s >= 4) m->cbp_luma |= 1 << idx;
                } else
                    m->cbp_luma |= 1 << idx;
            }
        }
        if (decimate_mb < 6 && b_decimate) { m->cbp_luma = 0; memset(m->nnz, 0, 16); }
        else
            for (int idx = 0; idx < 4; idx++) {
                int on = m->cbp_luma >> idx & 1;
                if (on) {
                    quantf.dequant_8x8(d8[idx], (int (*)[8][8])S->dq8[1], S->qp);
                    dctf.add8x8_idct8(m->fd[0] + (idx & 1) * 8 + (idx >> 1) * 8 * FDEC, d8[idx]);
                }
                for (int k = 0; k < 4; k++) m->nnz[4 * idx + k] = on;
            }
    } else {
        i16 d4[16][4][4];
        dctf.sub16x16_dct(d4, m->fe[0], m->fd[0]);
        if (S->b_nr && !S->lossless) S->nr_count[0] += 16;
        for (int i8 = 0; i8 < 4; i8++) {
            int dec8 = 0, cbp = 0;
            for (int i4 = 0; i4 < 4; i4++) {
                if (S->b_nr && !S->lossless) quantf.denoise_dct(&d4[4 * i8 + i4][0][0], S->nr_sum[0], S->nr_offset[0], 16);   /* macroblock.c:694 */
                int idx = 4 * i8 + i4, nz = q4(S, d4[idx], 1, 2, 0, S->qp);
                m->nnz[idx] = nz;
                if (nz) {
                    zigf[0].scan_4x4(m->luma4[idx], d4[idx]);
                    quantf.dequant_4x4(d4[idx], (int (*)[4][4])S->dq4[1], S->qp);
                    if (b_decimate && dec8 < 6) dec8 += quantf.decimate_score16(m->luma4[idx]);
                    cbp = 1;
                }
            }
            decimate_mb += dec8;
            if (b_decimate) {
                if (dec8 < 4) memset(m->nnz + 4 * i8, 0, 4);
                else m->cbp_luma |= 1 << i8;
            } else if (cbp) {
                dctf.add8x8_idct(m->fd[0] + (i8 & 1) * 8 + (i8 >> 1) * 8 * FDEC, &d4[4 * i8]);
                m->cbp_luma |= 1 << i8;
            }
        }
        if (b_decimate) {
            if (decimate_mb < 6) { m->cbp_luma = 0; memset(m->nnz, 0, 16); }
            else
                for (int i8 = 0; i8 < 4; i8++)
                    if (m->cbp_luma >> i8 & 1) dctf.add8x8_idct(m->fd[0] + (i8 & 1) * 8 + (i8 >> 1) * 8 * FDEC, &d4[4 * i8]);
        }
    }
}
/* x264_mb_mc for one 16x16 vector (R/common/macroblock.c:462-476), into fdec */
static void mc_16x16(const ssl *S, smb *m, int ref, int mvx, int mvy)
{
    const sframe *r = S->fref[ref];
    int oy = 16 * m->mby * S->sy + 16 * m->mbx, oc = 8 * m->mby * S->sc + 8 * m->mbx;
    u8 *src4[4] = {r->filt[0] + oy, r->filt[1] + oy, r->filt[2] + oy, r->filt[3] + oy};
    mcf.mc_luma(m->fd[0], FDEC, src4, S->sy, mvx, mvy, 16, 16);
    mcf.mc_chroma(m->fd[1], FDEC, r->plane[1] + oc, S->sc, mvx, mvy, 8, 8);
    mcf.mc_chroma(m->fd[2], FDEC, r->plane[2] + oc, S->sc, mvx, mvy, 8, 8);
}
static void mv_clip_frame(const ssl *S, const smb *m, int *mvx, int *mvy)
{   /* h->mb.mv_min / mv_max, R/encoder/analyse.c:258-259,290-291 */
    *mvx = clip3i(*mvx, 4 * (-16 * m->mbx - 24), 4 * (16 * (S->mb_w - m->mbx - 1) + 24));
    *mvy = clip3i(*mvy, 4 * (-16 * m->mby - 24), 4 * (16 * (S->mb_h - m->mby - 1) + 24));
}
/* x264_mb_mc for any of the partitions above: per 4x4 block, its vector and its 8x8's reference (R/common/macroblock.c:462-546) */
static void mc_parts(const ssl *S, smb *m)
{
    for (int by = 0; by < 4; by++)
        for (int bx = 0; bx < 4; bx++) {
            const sframe *r = S->fref[m->ref8[(by >> 1) * 2 + (bx >> 1)]];
            int v[2] = {m->mv4[by * 4 + bx][0], m->mv4[by * 4 + bx][1]};
            int o = (16 * m->mby + 4 * by) * S->sy + 16 * m->mbx + 4 * bx, oc = (8 * m->mby + 2 * by) * S->sc + 8 * m->mbx + 2 * bx;
            u8 *src4[4] = {r->filt[0] + o, r->filt[1] + o, r->filt[2] + o, r->filt[3] + o};
            mv_clip_frame(S, m, &v[0], &v[1]);                 /* x264_mb_mc_0xywh clips to h->mb.mv_min / mv_max (a vector the RD refinement tries may sit just outside) */
            mcf.mc_luma(m->fd[0] + 4 * by * FDEC + 4 * bx, FDEC, src4, S->sy, v[0], v[1], 4, 4);
            mcf.mc_chroma(m->fd[1] + 2 * by * FDEC + 2 * bx, FDEC, r->plane[1] + oc, S->sc, v[0], v[1], 2, 2);
            mcf.mc_chroma(m->fd[2] + 2 * by * FDEC + 2 * bx, FDEC, r->plane[2] + oc, S->sc, v[0], v[1], 2, 2);
        }
}
/* x264_macroblock_probe_skip, P path (R/encoder/macroblock.c:797-883) */
static int probe_pskip(ssl *S, smb *m)
{
    int mvx = m->pskip_mv[0], mvy = m->pskip_mv[1], dec = 0;
    const sframe *r = S->fref[0];
    int oy = 16 * m->mby * S->sy + 16 * m->mbx, oc = 8 * m->mby * S->sc + 8 * m->mbx;
    i16 d4[4][4][4], d2[2][2], scan[16];
    mv_clip_frame(S, m, &mvx, &mvy);
    u8 *src4[4] = {r->filt[0] + oy, r->filt[1] + oy, r->filt[2] + oy, r->filt[3] + oy};
    mcf.mc_luma(m->fd[0], FDEC, src4, S->sy, mvx, mvy, 16, 16);
    for (int i8 = 0; i8 < 4; i8++) {
        dctf.sub8x8_dct(d4, m->fe[0] + (i8 & 1) * 8 + (i8 >> 1) * 8 * FENC, m->fd[0] + (i8 & 1) * 8 + (i8 >> 1) * 8 * FDEC);
        for (int i4 = 0; i4 < 4; i4++) {
            if (!quantf.quant_4x4(d4[i4], S->mf4[1], S->b4[1])) continue;
            zigf[0].scan_4x4(scan, d4[i4]);
            dec += quantf.decimate_score16(scan);
            if (dec >= 6) return 0;
        }
    }
    int thresh = (s_lambda2_tab[S->qpc] + 32) >> 6;
    for (int ch = 0; ch < 2; ch++) {
        mcf.mc_chroma(m->fd[1 + ch], FDEC, r->plane[1 + ch] + oc, S->sc, mvx, mvy, 8, 8);
        if (pixf.ssd[X264HIP_PIXEL_8x8](m->fd[1 + ch], FDEC, m->fe[1 + ch], FENC) < thresh) continue;
        dctf.sub8x8_dct(d4, m->fe[1 + ch], m->fd[1 + ch]);
        int a = d4[0][0][0] + d4[1][0][0], b = d4[2][0][0] + d4[3][0][0];
        int c = d4[0][0][0] - d4[1][0][0], d = d4[2][0][0] - d4[3][0][0];
        d2[0][0] = a + b; d2[1][0] = c + d; d2[0][1] = a - b; d2[1][1] = c - d;
        d4[0][0][0] = d4[1][0][0] = d4[2][0][0] = d4[3][0][0] = 0;
        if (quantf.quant_2x2_dc(d2, S->mf4[3][0] >> 1, S->b4[3][0] << 1)) return 0;
        dec = 0;
        for (int i4 = 0; i4 < 4; i4++) {
            if (!quantf.quant_4x4(d4[i4], S->mf4[3], S->b4[3])) continue;
            zigf[0].scan_4x4(scan, d4[i4]);
            dec += quantf.decimate_score15(scan);
            if (dec >= 7) return 0;
        }
    }
    m->skip_mc = 1;
    return 1;
}

/* ------------------------------------------------------------------ intra analysis
 * predict_*_mode_available, R/encoder/analyse.c:374-471 */
static int modes_16x16(int nb, int *mode)
{
    if (nb & NB_TOPLEFT) { mode[0] = 0; mode[1] = 1; mode[2] = 2; mode[3] = 3; return 4; }
    if (nb & NB_LEFT) { mode[0] = 4; mode[1] = 1; return 2; }
    if (nb & NB_TOP) { mode[0] = 5; mode[1] = 0; return 2; }
    mode[0] = 6; return 1;
}
static int modes_chroma(int nb, int *mode)
{
    if (nb & NB_TOPLEFT) { mode[0] = 2; mode[1] = 1; mode[2] = 0; mode[3] = 3; return 4; }
    if (nb & NB_LEFT) { mode[0] = 4; mode[1] = 1; return 2; }
    if (nb & NB_TOP) { mode[0] = 5; mode[1] = 2; return 2; }
    mode[0] = 6; return 1;
}
static int modes_4x4(int nb, int *mode)
{
    int n = 0;
    if ((nb & NB_LEFT) && (nb & NB_TOP)) {
        mode[n++] = 2; mode[n++] = 1; mode[n++] = 0; mode[n++] = 3;
        if (nb & NB_TOPLEFT) { mode[n++] = 4; mode[n++] = 5; mode[n++] = 6; }
        mode[n++] = 7; mode[n++] = 8;
    } else if (nb & NB_LEFT) { mode[n++] = 9; mode[n++] = 1; mode[n++] = 8; }
    else if (nb & NB_TOP) { mode[n++] = 10; mode[n++] = 0; mode[n++] = 3; mode[n++] = 7; }
    else mode[n++] = 11;
    return n;
}
static int pred_intra4x4_mode(const smb *m, int idx)
{   /* x264_mb_predict_intra4x4_mode, R/common/macroblock.h:423-434 */
    int ma = s_fix4[m->i4c[s_scan8(idx) - 1] + 1], mb = s_fix4[m->i4c[s_scan8(idx) - 8] + 1], v = ma < mb ? ma : mb;
    return v < 0 ? 2 : v;
}
/* x264_mb_analyse_intra_chroma, R/encoder/analyse.c:539-610 */
static void analyse_intra_chroma(ssl *S, smb *m)
{
    int mode[4], n, satd = S->p->subme > 1 && !S->lossless;
    if (m->satd_chroma < S_COST_MAX) return;
    n = modes_chroma(m->nb, mode);
    for (int i = 0; i < n; i++) {
        pred_chroma(S, m, mode[i]);
        int c = (satd ? pixf.satd : pixf.sad)[X264HIP_PIXEL_8x8](m->fd[1], FDEC, m->fe[1], FENC)
              + (satd ? pixf.satd : pixf.sad)[X264HIP_PIXEL_8x8](m->fd[2], FDEC, m->fe[2], FENC)
              + S->lambda * s_ue_size(s_fix8c[mode[i]]);
        m->satd_c_dir[i] = c;
        if (c < m->satd_chroma) { m->satd_chroma = c; m->predc = mode[i]; }
    }
    m->chroma_mode = m->predc;
}
/* x264_mb_analyse_intra, :612-843 */
static void analyse_intra(ssl *S, smb *m, int satd_inter)
{
    const int flags = S->slice_type == S_SLICE_I ? S->p->intra : S->p->inter, satd = S->p->subme > 1 && !S->lossless;
    x264hip_pixel_cmp_t *cmp = satd ? pixf.satd : pixf.sad;
    int mode[9], n;
    n = modes_16x16(m->nb, mode);
    for (int i = 0; i < n; i++) {
        pred_16x16(S, m, mode[i]);
        int c = cmp[X264HIP_PIXEL_16x16](m->fd[0], FDEC, m->fe[0], FENC) + S->lambda * s_ue_size(s_fix16[mode[i]]);
        if (c < m->satd_i16) { m->satd_i16 = c; m->pred16 = mode[i]; }
        m->satd_i16_dir[mode[i]] = c;
    }
    if (S->slice_type == S_SLICE_B) m->satd_i16 += S->lambda * 9;      /* i_mb_b_cost_table[I_16x16], analyse.c:659-661 */
    if (m->fast_intra && m->satd_i16 > 2 * satd_inter) return;

    if (flags & 2) {                                           /* X264_ANALYSE_I8x8 */
        u8 edge[40];
        x264hip_pixel_cmp_t sa8d = satd ? pixf.sa8d[X264HIP_PIXEL_8x8] : pixf.sad[X264HIP_PIXEL_8x8];
        int thresh = S->mbrd ? S_COST_MAX : satd_inter < m->satd_i16 ? satd_inter : m->satd_i16, cost = 0, idx;
        if (S->slice_type == S_SLICE_B) cost += S->lambda * 9;         /* i_mb_b_cost_table[I_8x8], :676-677 */
        m->cbp_luma = 0;
        for (idx = 0;; idx++) {
            int x = idx & 1, y = idx >> 1, best = S_COST_MAX, pm = pred_intra4x4_mode(m, 4 * idx);
            u8 *src = m->fe[0] + 8 * x + 8 * y * FENC, *dst = m->fd[0] + 8 * x + 8 * y * FDEC;
            n = modes_4x4(m->nb8[idx], mode);
            s_p8filter(dst, edge, m->nb8[idx], 0xf);
            for (int i = 0; i < n; i++) {
                pred_8x8(S, m, idx, mode[i], edge);
                int c = sa8d(dst, FDEC, src, FENC) + S->lambda * (pm == s_fix4[mode[i] + 1] ? 1 : 4);
                if (c < best) { best = c; m->pred8[idx] = mode[i]; }
                m->satd_i8_dir[mode[i]][idx] = c;
            }
            cost += best;
            if (idx == 3 || cost > thresh) break;
            pred_8x8(S, m, idx, m->pred8[idx], edge);
            enc_i8x8(S, m, idx);
            for (int k = 0; k < 4; k++) m->i4c[s_scan8(4 * idx) + (k & 1) + 8 * (k >> 1)] = m->pred8[idx];
        }
        if (idx == 3) {
            m->satd_i8 = cost;
            if (m->skip_intra) {
                for (int r = 0; r < 16; r++) memcpy(m->i8_fdec + 16 * r, m->fd[0] + r * FDEC, 16);
                memcpy(m->i8_nnz, m->nnz, 16); m->i8_cbp = m->cbp_luma;
                if (m->skip_intra == 2) memcpy(m->i8_dct, m->luma8, sizeof(m->i8_dct));
            }
        } else {
            static const u16 div8[3] = {1024, 512, 341};
            m->satd_i8 = S_COST_MAX;
            cost = (cost * div8[idx]) >> 8;
        }
        if ((cost < m->satd_i16 ? cost : m->satd_i16) > satd_inter * (5 + !!S->mbrd) / 4) return;
    }
    if (flags & 1) {                                           /* X264_ANALYSE_I4x4 */
        int thresh = satd_inter < m->satd_i16 ? satd_inter : m->satd_i16, cost = S->lambda * 24, idx;
        if (m->satd_i8 < thresh) thresh = m->satd_i8;
        if (S->mbrd) thresh = thresh * (10 - m->fast_intra) / 8;
        if (S->slice_type == S_SLICE_B) cost += S->lambda * 9;         /* i_mb_b_cost_table[I_4x4], :770-771 */
        m->cbp_luma = 0;
        for (idx = 0;; idx++) {
            u8 *src = m->fe[0] + blk_x[idx] + blk_y[idx] * FENC, *dst = m->fd[0] + blk_x[idx] + blk_y[idx] * FDEC;
            int best = S_COST_MAX, pm = pred_intra4x4_mode(m, idx);
            n = modes_4x4(m->nb4[idx], mode);
            if ((m->nb4[idx] & (NB_TOPRIGHT | NB_TOP)) == NB_TOP) memset(dst + 4 - FDEC, dst[3 - FDEC], 4);
            for (int i = 0; i < n; i++) {
                pred_4x4(S, m, idx, mode[i]);
                int c = cmp[X264HIP_PIXEL_4x4](dst, FDEC, src, FENC) + S->lambda * (pm == s_fix4[mode[i] + 1] ? 1 : 4);
                if (c < best) { best = c; m->pred4[idx] = mode[i]; }
            }
            cost += best;
            if (cost > thresh || idx == 15) break;
            pred_4x4(S, m, idx, m->pred4[idx]);
            enc_i4x4(S, m, idx);
            m->i4c[s_scan8(idx)] = m->pred4[idx];
        }
        if (idx == 15) {
            m->satd_i4 = cost;
            if (m->skip_intra) {
                for (int r = 0; r < 16; r++) memcpy(m->i4_fdec + 16 * r, m->fd[0] + r * FDEC, 16);
                memcpy(m->i4_nnz, m->nnz, 16); m->i4_cbp = m->cbp_luma;
                if (m->skip_intra == 2) memcpy(m->i4_dct, m->luma4, sizeof(m->i4_dct));
            }
        } else
            m->satd_i4 = S_COST_MAX;
    }
}

/* ------------------------------------------------------------------ one macroblock */
static void load_mb(ssl *S, smb *m, int mbx, int mby)
{
    const u8 *src[3] = {S->fenc->plane[0], S->fenc->plane[1], S->fenc->plane[2]};
    memset(m, 0, sizeof(*m));
    m->mbx = mbx; m->mby = mby; m->mb = mby * S->mb_w + mbx;
    m->fe[0] = m->fenc; m->fe[1] = m->fenc + 16 * FENC; m->fe[2] = m->fenc + 16 * FENC + 8;
    m->fd[0] = m->fdec + 2 * FDEC; m->fd[1] = m->fdec + 19 * FDEC; m->fd[2] = m->fdec + 19 * FDEC + 16;
    m->type_left = m->type_top = m->type_topleft = m->type_topright = -1;
    if (mby > 0) { m->nb |= NB_TOP; m->type_top = S->fdec->mb_type[m->mb - S->mb_w]; }
    if (mbx > 0) { m->nb |= NB_LEFT; m->type_left = S->fdec->mb_type[m->mb - 1]; }
    if (mbx < S->mb_w - 1 && mby > 0) { m->nb |= NB_TOPRIGHT; m->type_topright = S->fdec->mb_type[m->mb - S->mb_w + 1]; }
    if (mbx > 0 && mby > 0) { m->nb |= NB_TOPLEFT; m->type_topleft = S->fdec->mb_type[m->mb - S->mb_w - 1]; }
    for (int pl = 0; pl < 3; pl++) {
        int w = pl ? 8 : 16, st = pl ? S->sc : S->sy, o = w * mby * st + w * mbx;
        const u8 *rec = S->fdec->plane[pl] + o;
        for (int y = 0; y < w; y++) memcpy(m->fe[pl] + y * FENC, src[pl] + o + y * st, w);
        /* reconstructed neighbours, still unfiltered: the row above (x = -1 .. w*3/2-1) and the column to the left */
        if (mby > 0) memcpy(m->fd[pl] - FDEC - 1, rec - st - 1, w * 3 / 2 + 1);
        if (mbx > 0) for (int y = 0; y < w; y++) m->fd[pl][y * FDEC - 1] = rec[y * st - 1];
    }
    /* intra4x4_pred_mode cache: -1 where there is no neighbour, DC for anything that is not I_4x4 / I_8x8 */
    memset(m->i4c, -1, sizeof(m->i4c));
    if (m->nb & NB_TOP) {
        const int8_t *t = S->i4mode + (m->mb - S->mb_w) * 16;
        m->i4c[s_scan8(0) - 8] = t[10]; m->i4c[s_scan8(1) - 8] = t[11]; m->i4c[s_scan8(4) - 8] = t[14]; m->i4c[s_scan8(5) - 8] = t[15];
    }
    if (m->nb & NB_LEFT) {
        const int8_t *l = S->i4mode + (m->mb - 1) * 16;
        m->i4c[s_scan8(0) - 1] = l[5]; m->i4c[s_scan8(2) - 1] = l[7]; m->i4c[s_scan8(8) - 1] = l[13]; m->i4c[s_scan8(10) - 1] = l[15];
    }
    /* i_neighbour4 / i_neighbour8, R/common/macroblock.c:733-743,1172-1186 */
    int nb = m->nb, all = NB_LEFT | NB_TOP | NB_TOPLEFT | NB_TOPRIGHT;
    m->nb4[0] = m->nb8[0] = (nb & (NB_TOP | NB_LEFT | NB_TOPLEFT)) | ((nb & NB_TOP) ? NB_TOPRIGHT : 0);
    m->nb4[4] = m->nb4[1] = NB_LEFT | ((nb & NB_TOP) ? (NB_TOP | NB_TOPLEFT | NB_TOPRIGHT) : 0);
    m->nb4[2] = m->nb4[8] = m->nb4[10] = m->nb8[2] = NB_TOP | NB_TOPRIGHT | ((nb & NB_LEFT) ? (NB_LEFT | NB_TOPLEFT) : 0);
    m->nb4[5] = m->nb8[1] = NB_LEFT | (nb & NB_TOPRIGHT) | ((nb & NB_TOP) ? NB_TOP | NB_TOPLEFT : 0);
    m->nb4[6] = m->nb4[9] = m->nb4[12] = m->nb4[14] = all;
    m->nb4[3] = m->nb4[7] = m->nb4[11] = m->nb4[13] = m->nb4[15] = m->nb8[3] = NB_LEFT | NB_TOP | NB_TOPLEFT;
    m->satd_i16 = m->satd_i8 = m->satd_i4 = m->satd_chroma = S_COST_MAX;
    if (S->slice_type == S_SLICE_P) predict_mv_pskip(S, m, m->pskip_mv);
    for (int list = 0; list < (S->slice_type == S_SLICE_B ? 2 : S->slice_type == S_SLICE_P ? 1 : 0); list++) {
        /* h->mb.cache.ref / mv around the macroblock (R/common/macroblock.c:1040-1128): -2 = not available */
        int8_t *cref = CREF(m, list);
        i16 (*cmv)[2] = CMV(m, list);
        memset(cref, -2, 48); memset(cmv, 0, sizeof(m->cmv));
        const i16 *fmv = list ? S->fdec->mv1 : S->fdec->mv;
        const int8_t *fref = list ? S->fdec->ref1 : S->fdec->ref;
#define NBSET(k_, o_, blk_) do { cref[k_] = fref[(o_) * 4 + ((blk_) >> 3) * 2 + (((blk_) & 3) >> 1)]; \
                                 cmv[k_][0] = fmv[((o_) * 16 + (blk_)) * 2]; cmv[k_][1] = fmv[((o_) * 16 + (blk_)) * 2 + 1]; } while (0)
        if (m->nb & NB_TOPLEFT) NBSET(3, m->mb - S->mb_w - 1, 15);
        if (m->nb & NB_TOP) for (int i = 0; i < 4; i++) NBSET(4 + i, m->mb - S->mb_w, 12 + i);
        if (m->nb & NB_TOPRIGHT) NBSET(8, m->mb - S->mb_w + 1, 12);
        if (m->nb & NB_LEFT) for (int i = 0; i < 4; i++) NBSET(11 + 8 * i, m->mb - 1, 3 + 4 * i);
#undef NBSET
        cref[s_scan8(12)] = S->stale_ref[list]; cmv[s_scan8(12)][0] = S->stale_mv[list][0]; cmv[s_scan8(12)][1] = S->stale_mv[list][1];
    }
    m->partition = S_D_16x16;
    /* what the entropy coder reads of the neighbours (R/common/macroblock.c:896-1010,1129-1160) */
    m->cbp_left = m->cbp_top = -1; m->cpm_left = m->cpm_top = 0; m->nb_t8 = 0;
    memset(m->nz_l, 0x80, 4); memset(m->nz_t, 0x80, 4); memset(m->nz_lc, 0x80, 4); memset(m->nz_tc, 0x80, 4);
    memset(m->cmvd, 0, sizeof(m->cmvd));
    memcpy(m->nnz, S->carry_nnz, 27);                     /* the macroblock's own cache entries: as the previous macroblock left them (see carry_nnz) */
    for (int i = 0; i < 16; i++) { const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2); m->cmvd[k][0] = S->carry_mvd[0][i][0]; m->cmvd[k][1] = S->carry_mvd[0][i][1]; }
    if (S->cbp) {
        if (m->nb & NB_TOP) {
            const int t = m->mb - S->mb_w;
            const u8 *nz = S->nnz + t * 27;
            m->cbp_top = S->cbp[t]; m->cpm_top = S->chroma_pm[t]; m->nb_t8 += S->t8[t];
            m->nz_t[0] = nz[10]; m->nz_t[1] = nz[11]; m->nz_t[2] = nz[14]; m->nz_t[3] = nz[15];
            for (int ch = 0; ch < 2; ch++) { m->nz_tc[ch][0] = nz[16 + 4 * ch + 2]; m->nz_tc[ch][1] = nz[16 + 4 * ch + 3]; }
            for (int i = 0; i < 4; i++) { m->cmvd[4 + i][0] = S->mvd[(t * 16 + 12 + i) * 2]; m->cmvd[4 + i][1] = S->mvd[(t * 16 + 12 + i) * 2 + 1]; }
        }
        if (m->nb & NB_LEFT) {
            const int l = m->mb - 1;
            const u8 *nz = S->nnz + l * 27;
            m->cbp_left = S->cbp[l]; m->cpm_left = S->chroma_pm[l]; m->nb_t8 += S->t8[l];
            m->nz_l[0] = nz[5]; m->nz_l[1] = nz[7]; m->nz_l[2] = nz[13]; m->nz_l[3] = nz[15];
            for (int ch = 0; ch < 2; ch++) { m->nz_lc[ch][0] = nz[16 + 4 * ch + 1]; m->nz_lc[ch][1] = nz[16 + 4 * ch + 3]; }
            for (int i = 0; i < 4; i++) { m->cmvd[11 + 8 * i][0] = S->mvd[(l * 16 + 3 + 4 * i) * 2]; m->cmvd[11 + 8 * i][1] = S->mvd[(l * 16 + 3 + 4 * i) * 2 + 1]; }
        }
        if (S->slice_type == S_SLICE_B) {                /* list 1 of the mvd cache and the skip flags of direct blocks, macroblock.c:1129-1160 */
            memset(m->cmvd1, 0, sizeof(m->cmvd1)); memset(m->cskip, 0, sizeof(m->cskip));
            for (int i = 0; i < 16; i++) { const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2); m->cmvd1[k][0] = S->carry_mvd[1][i][0]; m->cmvd1[k][1] = S->carry_mvd[1][i][1]; }
            if (m->nb & NB_TOP) {
                const int t = m->mb - S->mb_w, sb = S->skipbp[t];
                for (int i = 0; i < 4; i++) { m->cmvd1[4 + i][0] = S->mvd1[(t * 16 + 12 + i) * 2]; m->cmvd1[4 + i][1] = S->mvd1[(t * 16 + 12 + i) * 2 + 1]; }
                m->cskip[s_scan8(0) - 8] = sb & 4; m->cskip[s_scan8(4) - 8] = sb & 8;
            }
            if (m->nb & NB_LEFT) {
                const int l = m->mb - 1, sb = S->skipbp[l];
                for (int i = 0; i < 4; i++) { m->cmvd1[11 + 8 * i][0] = S->mvd1[(l * 16 + 3 + 4 * i) * 2]; m->cmvd1[11 + 8 * i][1] = S->mvd1[(l * 16 + 3 + 4 * i) * 2 + 1]; }
                m->cskip[s_scan8(0) - 1] = sb & 2; m->cskip[s_scan8(8) - 1] = sb & 8;
            }
        }
    }
}

static void set_me_ctx_blk_l(const ssl *S, const smb *m, int list, int ref, const i16 mvp[2], me_ctx *c, int pix, int bx, int by)
{
    static const u8 bw[7] = {16, 16, 8, 8, 8, 4, 4}, bh[7] = {16, 8, 16, 8, 4, 8, 4};
    int oy = (16 * m->mby + by) * S->sy + 16 * m->mbx + bx, oc = (8 * m->mby + by / 2) * S->sc + 8 * m->mbx + bx / 2, sp[4], fp[4];
    const sframe *r = list ? S->fref1[ref] : S->fref[ref];
    mv_limits(S->mb_w, S->mb_h, m->mbx, m->mby, S->p->mv_range > 0 ? S->p->mv_range : 512, sp, fp);
    c->fenc = S->fenc->plane[0] + oy; c->fenc_u = S->fenc->plane[1] + oc; c->fenc_v = S->fenc->plane[2] + oc;
    c->sy = S->sy; c->sc = S->sc;
    for (int k = 0; k < 4; k++) c->fref[k] = r->filt[k] + oy;
    c->fref[4] = r->plane[1] + oc; c->fref[5] = r->plane[2] + oc;
    c->cmx = S->cost_mv - mvp[0]; c->cmy = S->cost_mv - mvp[1];
    c->fmin[0] = fp[0]; c->fmax[0] = fp[1]; c->fmin[1] = fp[2]; c->fmax[1] = fp[3];
    c->smin[0] = sp[0]; c->smax[0] = sp[1]; c->smin[1] = sp[2]; c->smax[1] = sp[3];
    c->pix = pix; c->bw = bw[pix]; c->bh = bh[pix];
}
static void set_me_ctx_blk(const ssl *S, const smb *m, int ref, const i16 mvp[2], me_ctx *c, int pix, int bx, int by) { set_me_ctx_blk_l(S, m, 0, ref, mvp, c, pix, bx, by); }
static void set_me_ctx(const ssl *S, const smb *m, int ref, const i16 mvp[2], me_ctx *c) { set_me_ctx_blk(S, m, ref, mvp, c, X264HIP_PIXEL_16x16, 0, 0); }
/* h->mb.cache.ref / mv helpers: x264_macroblock_cache_ref / _mv (R/common/macroblock.h) on a w x h run of 4x4 blocks */
static void cache_set_l(smb *m, int list, int x, int y, int w, int h, int ref, int mvx, int mvy, int set_mv)
{
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            int k = 4 + 1 * 8 + x + i + 8 * (y + j);
            CREF(m, list)[k] = (int8_t)ref;
            if (set_mv) { CMV(m, list)[k][0] = (i16)mvx; CMV(m, list)[k][1] = (i16)mvy; }
        }
}
static void cache_set(smb *m, int x, int y, int w, int h, int ref, int mvx, int mvy, int set_mv) { cache_set_l(m, 0, x, y, w, h, ref, mvx, mvy, set_mv); }
/* x264_mb_predict_mv, R/common/macroblock.c:28-88 */
static void predict_mv_blk_l(const smb *m, int list, int idx, int width, i16 mvp[2])
{
    const int8_t *cref = CREF(m, list);
    const i16 (*cmv)[2] = CMV(m, list);
    const int i8 = s_scan8(idx), i_ref = cref[i8];
    int ra = cref[i8 - 1], rb = cref[i8 - 8], rc = cref[i8 - 8 + width], cnt;
    const i16 *a = cmv[i8 - 1], *b = cmv[i8 - 8], *c = cmv[i8 - 8 + width];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { rc = cref[i8 - 8 - 1]; c = cmv[i8 - 8 - 1]; }
    if (m->partition == S_D_16x8) {
        if (idx == 0 && rb == i_ref) { mvp[0] = b[0]; mvp[1] = b[1]; return; }
        if (idx != 0 && ra == i_ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
    } else if (m->partition == S_D_8x16) {
        if (idx == 0 && ra == i_ref) { mvp[0] = a[0]; mvp[1] = a[1]; return; }
        if (idx != 0 && rc == i_ref) { mvp[0] = c[0]; mvp[1] = c[1]; return; }
    }
    cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
    if (cnt > 1) { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
    else if (cnt == 1) { const i16 *s = ra == i_ref ? a : rb == i_ref ? b : c; mvp[0] = s[0]; mvp[1] = s[1]; }
    else if (rb == -2 && rc == -2 && ra != -2) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
}
static void predict_mv_blk(const smb *m, int idx, int width, i16 mvp[2]) { predict_mv_blk_l(m, 0, idx, width, mvp); }
/* x264_mb_transform_8x8_allowed (R/common/macroblock.h:452-466): large P partitions, P_8x8 only with four 8x8 sub-partitions */
static int s_t8_allowed(const ssl *S, const smb *m)
{
    if (!S->p->transform8x8) return 0;
    if (m->type == S_P_L0 || (m->type >= S_B_DIRECT && m->type <= S_B_8x8)) return 1;   /* every B type but B_SKIP (direct_8x8_inference is on) */
    return m->type == S_P_8x8 && m->sub[0] == S_D_L0_8x8 && m->sub[1] == S_D_L0_8x8 && m->sub[2] == S_D_L0_8x8 && m->sub[3] == S_D_L0_8x8;
}
#include "cabac_oracle.c"
#ifdef X264O_DEVCHECK
/* Cross-check of the PRODUCT's scalar device code, built for the host (oracle/devcheck.cpp): every call of the twin's CABAC
 * writer / bit counter and trellis quantiser is replayed through x264_vs2008_amd/csrc/cabac_dev.h / trellis_dev.h on a copy
 * of the same state and compared.  g_devcheck_bad counts differences (tests/test_devhost.py expects 0).                    */
#include "../x264_vs2008_amd/csrc/mbsyn.h"
void devhost_cw_macroblock(DCabac *cb, uint8_t *st, int rd, MbSyn *m, const uint8_t *fe, int i_frame);
void devhost_mb_skip(DCabac *cb, uint8_t *st, int type_left, int type_top, int b_skip);
void devhost_terminal(DCabac *cb);
void devhost_flush(DCabac *cb, int i_frame);
void devhost_context_init(uint8_t *st, int slice_type, int qp, int model);
int devhost_trellis(int16_t *dct, const uint16_t *mf, const int *unq, const int *weight, const uint8_t *zz, const uint8_t *st,
                    int cat, int lambda2, int b_ac, int dc, int n_coef);
int g_devcheck_bad, g_devcheck_calls;
int x264o_devcheck_bad(void) { return g_devcheck_bad; }
int x264o_devcheck_calls(void) { return g_devcheck_calls; }
static void mbsyn_fill(const ssl *S, const smb *m, MbSyn *y)
{
    memset(y, 0, sizeof(*y));
    y->slice_type = S->slice_type; y->type = m->type; y->partition = m->partition;
    y->i16mode = m->i16mode; y->chroma_mode = m->chroma_mode; y->cbp_luma = m->cbp_luma; y->cbp_chroma = m->cbp_chroma; y->t8 = m->t8; y->qp = m->qp;
    y->n_ref = S->n_ref; y->n_ref1 = S->n_ref1; y->pps_t8 = S->p->transform8x8; y->t8_allowed = s_t8_allowed(S, m);
    y->type_left = m->type_left; y->type_top = m->type_top; y->cbp_left = m->cbp_left; y->cbp_top = m->cbp_top;
    y->cpm_left = m->cpm_left; y->cpm_top = m->cpm_top; y->nb_t8 = m->nb_t8;
    y->last_qp = S->last_qp; y->last_dqp = S->last_dqp;
    y->prev_coded = m->mb > 0 && (S->fdec->mb_type[S->prev_mb] == S_I_16x16 || (S->cbp[S->prev_mb] & 0x3f));
    memcpy(y->sub, m->sub, 4); memcpy(y->i4c, m->i4c, 48); memcpy(y->cref, m->cref, 48);
    memcpy(y->cmv, m->cmv, sizeof(y->cmv)); memcpy(y->cmvd, m->cmvd, sizeof(y->cmvd));
    memcpy(y->cref1, m->cref1, 48); memcpy(y->cmv1, m->cmv1, sizeof(y->cmv1)); memcpy(y->cmvd1, m->cmvd1, sizeof(y->cmvd1)); memcpy(y->cskip, m->cskip, 48);
    memcpy(y->nnz, m->nnz, 27);
    memcpy(y->nz_l, m->nz_l, 4); memcpy(y->nz_t, m->nz_t, 4); memcpy(y->nz_lc, m->nz_lc, 4); memcpy(y->nz_tc, m->nz_tc, 4);
    memcpy(y->lv4, m->luma4, sizeof(y->lv4)); memcpy(y->lv8, m->luma8, sizeof(y->lv8)); memcpy(y->lv_dc, m->dc16, sizeof(y->lv_dc));
    memcpy(y->lv_cdc, m->cdc, sizeof(y->lv_cdc)); memcpy(y->lv_cac, m->cac, sizeof(y->lv_cac));
}
static void cw_macroblock_chk(ssl *S, o_cabac *cb, int rd, smb *m)
{
    MbSyn y;
    DCabac d = {cb->low, cb->range, cb->queue, cb->outstanding, 0, cb->f8};
    u8 st[460], out[64 + 1024], fe[384];
    memcpy(st, cb->state, 460);
    memset(out, 0, sizeof(out));
    d.p = out + 64;
    mbsyn_fill(S, m, &y);
    memcpy(fe, m->fe[0], 256);
    for (int pl = 1; pl < 3; pl++) for (int i = 0; i < 8; i++) memcpy(fe + 256 + 64 * (pl - 1) + 8 * i, m->fe[pl] + i * FENC, 8);
    u8 *p0 = cb->p;
    cw_macroblock(S, cb, rd, m);
    devhost_cw_macroblock(&d, st, rd, &y, fe, cb->i_frame);
    g_devcheck_calls++;
    int bad = memcmp(st, cb->state, 460) != 0 || y.qp != m->qp || memcmp(y.cmvd, m->cmvd, sizeof(y.cmvd)) != 0
           || (S->slice_type == S_SLICE_B && memcmp(y.cmvd1, m->cmvd1, sizeof(y.cmvd1)) != 0);
    if (rd) bad |= d.f8 != cb->f8;
    else {
        const int n = (int)(cb->p - p0);
        bad |= d.low != cb->low || d.range != cb->range || d.queue != cb->queue || d.outstanding != cb->outstanding || (int)(d.p - (out + 64)) != n
            || memcmp(out + 64, p0, n > 0 ? n : 0) != 0;
    }
    if (bad) { if (!g_devcheck_bad) fprintf(stderr, "devcheck: cw_macroblock differs (rd %d, frame %d, mb %d, type %d)\n", rd, S->f, m->mb, m->type); g_devcheck_bad++; }
}
#define cw_macroblock cw_macroblock_chk
#endif
#include "rd_oracle.c"

/* x264_me_refine_qpel -> refine_subpel(.., b_refine_qpel = 1), R/encoder/me.c:634-778, 16x16 */
static int refine_qpel16(const ssl *S, const me_ctx *c, int cost, int *pmx, int *pmy, const i16 mvp[2])
{
    int subme = S->p->subme, hpel = me_subpel_iters[subme][0], qpel = me_subpel_iters[subme][1];
    int satd = subme > 1 && !S->lossless, chroma_me = S->p->chroma_me && S->slice_type == S_SLICE_P && subme >= 5 && c->pix <= X264HIP_PIXEL_8x8;   /* b_chroma_me (P slices only, analyse.c:234) && i_pixel <= PIXEL_8x8, me.c:654 */
    int bx = *pmx, by = *pmy, bc = cost, i, cst;
    if (hpel && subme < 3) {
        int mx = clip3i(mvp[0], c->smin[0], c->smax[0]), my = clip3i(mvp[1], c->smin[1], c->smax[1]);
        if ((mx - bx) | (my - by)) { cst = me_qpel_cmp(c, mx, my, 0); if (cst < bc) { bc = cst; bx = mx; by = my; } }
    }
    for (i = hpel; i > 0; i--) {
        int ox = bx, oy = by, c0 = me_qpel_cmp(c, ox, oy - 2, 0), c1 = me_qpel_cmp(c, ox, oy + 2, 0);
        int c2 = me_qpel_cmp(c, ox - 2, oy, 0), c3 = me_qpel_cmp(c, ox + 2, oy, 0);
        if (c0 < bc) { bc = c0; by = oy - 2; }
        if (c1 < bc) { bc = c1; by = oy + 2; }
        if (c2 < bc) { bc = c2; bx = ox - 2; by = oy; }
        if (c3 < bc) { bc = c3; bx = ox + 2; by = oy; }
        if (bx == ox && by == oy) break;
    }
    for (i = qpel; i > 0; i--) {
        static const int dq[4][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}};
        int ox = bx, oy = by;
        for (int d = 0; d < 4; d++) {
            cst = me_satd_chroma(c, ox + dq[d][0], oy + dq[d][1], bc, chroma_me, satd);
            if (cst < bc) { bc = cst; bx = ox + dq[d][0]; by = oy + dq[d][1]; }
        }
        if (bx == ox && by == oy) break;
    }
    if (by > c->smax[1]) {
        by = c->smax[1]; bc = S_COST_MAX;
        cst = me_satd_chroma(c, bx, by, bc, chroma_me, satd);
        if (cst < bc) bc = cst;
    }
    *pmx = bx; *pmy = by;
    return bc;
}

/* x264_mb_analyse_inter_p4x4_chroma, R/encoder/analyse.c:1373-1405: mc_chroma of every sub-block into a 4x4, then mbcmp 4x4 on U and V */
static int p4x4_chroma(const ssl *S, const smb *m, int ref, int i8, int sub, const sub_me *me)
{
    u8 pix1[16 * 8], *pix2 = pix1 + 8;
    const sframe *r = S->fref[ref];
    const int oc = (8 * m->mby + 4 * (i8 >> 1)) * S->sc + 8 * m->mbx + 4 * (i8 & 1), oe = 4 * (i8 >> 1) * FENC + 4 * (i8 & 1);
    const int n = sub == S_D_L0_4x4 ? 4 : 2, w = sub == S_D_L0_8x4 ? 4 : 2, h = sub == S_D_L0_4x8 ? 4 : 2;
    for (int k = 0; k < n; k++) {
        const int x = sub == S_D_L0_4x4 ? 2 * (k & 1) : sub == S_D_L0_4x8 ? 2 * k : 0, y = sub == S_D_L0_4x4 ? 2 * (k >> 1) : sub == S_D_L0_8x4 ? 2 * k : 0;
        mcf.mc_chroma(&pix1[x + y * 16], 16, r->plane[1] + oc + x + y * S->sc, S->sc, me[k].mvx, me[k].mvy, w, h);
        mcf.mc_chroma(&pix2[x + y * 16], 16, r->plane[2] + oc + x + y * S->sc, S->sc, me[k].mvx, me[k].mvy, w, h);
    }
    const int satd = S->p->subme > 1 && !S->lossless;
    return (satd ? pixf.satd : pixf.sad)[6](m->fe[1] + oe, FENC, pix1, 16) + (satd ? pixf.satd : pixf.sad)[6](m->fe[2] + oe, FENC, pix2, 16);
}

/* x264_noise_reduction_update, R/encoder/macroblock.c:890-911 (weights: x264_dct4_weight2_tab / x264_dct8_weight2_tab, dct.h:56-83) */
static void nr_update(ssl *S)
{
    static const uint16_t w4[3] = {800, 320, 128}, w8[6] = {256, 201, 656, 227, 410, 363};
    static const u8 k4[16] = {0, 1, 0, 1, 1, 2, 1, 2, 0, 1, 0, 1, 1, 2, 1, 2};
    static const u8 k8[32] = {0, 3, 4, 3, 0, 3, 4, 3, 3, 1, 5, 1, 3, 1, 5, 1, 4, 5, 2, 5, 4, 5, 2, 5, 3, 1, 5, 1, 3, 1, 5, 1};
    for (int cat = 0; cat < 2; cat++) {
        const int size = cat ? 64 : 16;
        if (S->nr_count[cat] > (cat ? (1u << 16) : (1u << 18))) {
            for (int i = 0; i < size; i++) S->nr_sum[cat][i] >>= 1;
            S->nr_count[cat] >>= 1;
        }
        for (int i = 0; i < size; i++) {
            const unsigned long long w = cat ? w8[k8[i & 31]] : w4[k4[i]];
            S->nr_offset[cat][i] = (uint16_t)(((unsigned long long)S->p->noise_reduction * S->nr_count[cat] + S->nr_sum[cat][i] / 2)
                                         / ((unsigned long long)S->nr_sum[cat][i] * w / 256 + 1));
        }
    }
}

/* x264_analyse_update_cache, R/encoder/analyse.c:2777-2846 (I and P types): the candidate `m->type / m->partition` names becomes
 * the macroblock's vectors, references (h->mb.cache and what cache_save will store) or intra modes.                              */
struct banalysis;
static void update_cache_b(ssl *S, smb *m, struct banalysis *B);
static void fill_part(smb *m, int x, int y, int w, int h, int ref, int mvx, int mvy)
{
    cache_set(m, x, y, w, h, ref, mvx, mvy, 1);
    for (int j = y; j < y + h; j++)
        for (int i = x; i < x + w; i++) { m->mv4[j * 4 + i][0] = (i16)mvx; m->mv4[j * 4 + i][1] = (i16)mvy; m->ref8[(j >> 1) * 2 + (i >> 1)] = (int8_t)ref; }
}
static void update_cache(ssl *S, smb *m, const panalysis *A)
{
    switch (m->type) {
    case S_I_4x4:
        for (int i = 0; i < 16; i++) m->i4c[s_scan8(i)] = m->pred4[i];
        analyse_intra_chroma(S, m);
        break;
    case S_I_8x8:
        for (int i = 0; i < 16; i++) m->i4c[s_scan8(i)] = m->pred8[i >> 2];
        analyse_intra_chroma(S, m);
        break;
    case S_I_16x16:
        m->i16mode = m->pred16;
        analyse_intra_chroma(S, m);
        break;
    case S_P_L0:
        if (m->partition == S_D_16x16) fill_part(m, 0, 0, 4, 4, A->me16.ref, A->me16.mvx, A->me16.mvy);
        else if (m->partition == S_D_16x8)
            for (int i = 0; i < 2; i++) fill_part(m, 0, 2 * i, 4, 2, A->me16x8[i].ref, A->me16x8[i].mvx, A->me16x8[i].mvy);
        else
            for (int i = 0; i < 2; i++) fill_part(m, 2 * i, 0, 2, 4, A->me8x16[i].ref, A->me8x16[i].mvx, A->me8x16[i].mvy);
        break;
    case S_P_8x8:
        for (int i = 0; i < 4; i++) {                        /* x264_mb_cache_mv_p8x8, :1058-1075 */
            const int x0 = 2 * (i & 1), y0 = 2 * (i >> 1), r = A->me8[i].ref, t = A->sub[i];
            m->sub[i] = (int8_t)t;
            if (t == S_D_L0_8x8) fill_part(m, x0, y0, 2, 2, r, A->me8[i].mvx, A->me8[i].mvy);
            else if (t == S_D_L0_8x4) for (int k = 0; k < 2; k++) fill_part(m, x0, y0 + k, 2, 1, r, A->me84[i][k].mvx, A->me84[i][k].mvy);
            else if (t == S_D_L0_4x8) for (int k = 0; k < 2; k++) fill_part(m, x0 + k, y0, 1, 2, r, A->me48[i][k].mvx, A->me48[i][k].mvy);
            else for (int k = 0; k < 4; k++) fill_part(m, x0 + (k & 1), y0 + (k >> 1), 1, 1, r, A->me4[i][k].mvx, A->me4[i][k].mvy);
        }
        break;
    case S_P_SKIP:
        m->partition = S_D_16x16;
        fill_part(m, 0, 0, 4, 4, 0, m->pskip_mv[0], m->pskip_mv[1]);
        m->mvx = m->pskip_mv[0]; m->mvy = m->pskip_mv[1]; m->ref = 0;
        break;
    case S_I_PCM:
        break;
    default:
        update_cache_b(S, m, A->B);
        break;
    }
}
/* x264_mb_analyse_p_rd, :1935-2005 */
static uint64_t rd_cost_part(ssl *S, smb *m, int lambda2, int i4, int pix);
static void cache_mv_p8x8(smb *m, const panalysis *A, int i)
{   /* x264_mb_cache_mv_p8x8, :1058-1075, with the sub-partition type in m->sub[i] */
    const int x0 = 2 * (i & 1), y0 = 2 * (i >> 1), r = A->me8[i].ref, t = m->sub[i];
    if (t == S_D_L0_8x8) fill_part(m, x0, y0, 2, 2, r, A->me8[i].mvx, A->me8[i].mvy);
    else if (t == S_D_L0_8x4) for (int k = 0; k < 2; k++) fill_part(m, x0, y0 + k, 2, 1, r, A->me84[i][k].mvx, A->me84[i][k].mvy);
    else if (t == S_D_L0_4x8) for (int k = 0; k < 2; k++) fill_part(m, x0 + k, y0, 1, 2, r, A->me48[i][k].mvx, A->me48[i][k].mvy);
    else for (int k = 0; k < 4; k++) fill_part(m, x0 + (k & 1), y0 + (k >> 1), 1, 1, r, A->me4[i][k].mvx, A->me4[i][k].mvy);
}
static void analyse_p_rd(ssl *S, smb *m, panalysis *A, int i_satd)
{
    const int thresh = i_satd * 5 / 4;
    m->type = S_P_L0;
    if (A->rd16 == S_COST_MAX && A->me16.cost <= i_satd * 3 / 2) {
        m->partition = S_D_16x16;
        update_cache(S, m, A);
        A->rd16 = rd_cost_mb(S, m, S->lambda2);
    }
    A->me16.cost = A->rd16;
    if (A->cost16x8 <= thresh) { m->partition = S_D_16x8; update_cache(S, m, A); A->cost16x8 = rd_cost_mb(S, m, S->lambda2); }
    else A->cost16x8 = S_COST_MAX;
    if (A->cost8x16 <= thresh) { m->partition = S_D_8x16; update_cache(S, m, A); A->cost8x16 = rd_cost_mb(S, m, S->lambda2); }
    else A->cost8x16 = S_COST_MAX;
    if (A->cost8x8 <= thresh) {
        m->type = S_P_8x8; m->partition = S_D_8x8;
        if (S->p->inter & 0x20) {                            /* X264_ANALYSE_PSUB8x8: every 8x8 block's sub-partition by RD, :1968-1996 */
            for (int i = 0; i < 4; i++) cache_set(m, 2 * (i & 1), 2 * (i >> 1), 2, 2, A->me8[i].ref, 0, 0, 0);
            for (int i = 0; i < 4; i++) {
                const int costs[4] = {A->cost_sub[i][0], A->cost_sub[i][1], A->cost_sub[i][2], A->me8[i].cost};
                int mn = costs[0] < costs[1] ? costs[0] : costs[1], btype = S_D_L0_8x8;
                if (costs[2] < mn) mn = costs[2];
                if (costs[3] < mn) mn = costs[3];
                const int th = mn * 5 / 4;
                uint64_t bcost = (uint64_t)1 << 60;
                for (int subtype = S_D_L0_4x4; subtype <= S_D_L0_8x8; subtype++) {
                    if (costs[subtype] > th || (subtype == S_D_L0_8x8 && bcost == (uint64_t)1 << 60)) continue;
                    m->sub[i] = (int8_t)subtype;
                    cache_mv_p8x8(m, A, i);
                    const uint64_t c = rd_cost_part(S, m, S->lambda2, i << 2, X264HIP_PIXEL_8x8);
                    if (c < bcost) { bcost = c; btype = subtype; }
                }
                m->sub[i] = (int8_t)btype; A->sub[i] = btype;
                cache_mv_p8x8(m, A, i);
            }
        } else
            update_cache(S, m, A);
        A->cost8x8 = rd_cost_mb(S, m, S->lambda2);
    } else A->cost8x8 = S_COST_MAX;
}
/* x264_intra_rd, :845-874 */
static void intra_rd(ssl *S, smb *m, const panalysis *A, int thresh)
{
    if (m->satd_i16 <= thresh) { m->type = S_I_16x16; update_cache(S, m, A); m->satd_i16 = rd_cost_mb(S, m, S->lambda2); }
    else m->satd_i16 = S_COST_MAX;
    if (m->satd_i4 <= thresh && m->satd_i4 < S_COST_MAX) { m->type = S_I_4x4; update_cache(S, m, A); m->satd_i4 = rd_cost_mb(S, m, S->lambda2); }
    else m->satd_i4 = S_COST_MAX;
    if (m->satd_i8 <= thresh && m->satd_i8 < S_COST_MAX) { m->type = S_I_8x8; update_cache(S, m, A); m->satd_i8 = rd_cost_mb(S, m, S->lambda2); m->cbp_i8_rd = m->cbp_luma; }
    else m->satd_i8 = S_COST_MAX;
}
/* x264_mb_analyse_transform_rd, :2127-2150 */
static void transform_rd(ssl *S, smb *m, const panalysis *A, int *i_satd, int *i_rd)
{
    if (!s_t8_allowed(S, m) || !S->p->transform8x8) return;
    update_cache(S, m, A);
    m->t8 = !m->t8;
    const int rd8 = rd_cost_mb(S, m, S->lambda2);
    if (*i_rd >= rd8) {
        if (*i_rd > 0) *i_satd = (int)((int64_t)*i_satd * rd8 / *i_rd);
        if (*i_satd == 0) *i_satd = 1;
        *i_rd = rd8;
    } else
        m->t8 = !m->t8;
}

#include "b_oracle.c"
#include "refine_oracle.c"
static void analyse_mb(ssl *S, smb *m, panalysis *A)
{
    const slice_params *p = S->p;
    int i_cost = S_COST_MAX;
    m->skip_mc = 0; m->t8 = 0;
    /* x264_mb_analyse_init's fast-intra decision, R/encoder/analyse.c:345-362 */
    if (S->slice_type != S_SLICE_I && m->mb > 4) {
        int likely = S_IS_INTRA(m->type_left) || S_IS_INTRA(m->type_top) || S_IS_INTRA(m->type_topleft) || S_IS_INTRA(m->type_topright)
                  || (S->slice_type == S_SLICE_P && S_IS_INTRA(S->fref[0]->mb_type[m->mb])) || m->mb < 3 * S->intra_count;
        m->fast_intra = !likely;
    }
    const int satd_pcm = !S->psy_rd && S->mbrd ? (int)(((uint64_t)(386 * 8) * S->lambda2 + 128) >> 8) : S_COST_MAX;   /* a->i_satd_pcm, :246 */
    if (S->slice_type == S_SLICE_I) {
        if (S->mbrd) cache_fenc_satd(S, m);
        analyse_intra(S, m, S_COST_MAX);
        if (S->mbrd) intra_rd(S, m, A, S_COST_MAX);
        i_cost = m->satd_i16; m->type = S_I_16x16;
        if (m->satd_i4 < i_cost) { i_cost = m->satd_i4; m->type = S_I_4x4; }
        if (m->satd_i8 < i_cost) { i_cost = m->satd_i8; m->type = S_I_8x8; }
        if (satd_pcm < i_cost) m->type = S_I_PCM;
        else if (S->mbrd >= 2) intra_rd_refine(S, m);            /* :2184-2185 */
    } else if (S->slice_type == S_SLICE_B) {
        analyse_b(S, m, A, satd_pcm);
    } else {
        int b_skip = 0, try_pskip = 0;
        if (p->fast_pskip && !S->lossless) {
            if (p->subme >= 3) try_pskip = 1;
            else if (m->type_left == S_P_SKIP || m->type_top == S_P_SKIP || m->type_topleft == S_P_SKIP || m->type_topright == S_P_SKIP)
                b_skip = probe_pskip(S, m);
        }
        if (b_skip) m->type = S_P_SKIP;
        else {
            /* x264_mb_analyse_inter_p16x16, :1077-1143 */
            int thresh = 0x7fffffff, best = 0x7fffffff, bmx = 0, bmy = 0, bref = 0, chroma_me = p->chroma_me && p->subme >= 5;
            i16 bmvp[2] = {0, 0};
            for (int r = 0; r < S->n_ref; r++) {
                i16 mvp[2], mvc[8][2];
                me_ctx c;
                int mvx, mvy, cost, cost_mv = 0, n_mvc;
                predict_mv_16x16(S, m, r, mvp);
                n_mvc = predict_mv_ref16x16(S, m, r, mvc);
                set_me_ctx(S, m, r, mvp, &c);
                thresh -= S->ref_cost[r];
                cost = me_search16(&c, mvp, (const i16 (*)[2])mvc, n_mvc, p->me_method, p->me_range, p->subme, chroma_me,
                                   S->n_ref > 1 ? &thresh : 0, &mvx, &mvy, &cost_mv);
                if (r == 0 && try_pskip && cost - cost_mv < 300 * S->lambda
                    && abs(mvx - m->pskip_mv[0]) + abs(mvy - m->pskip_mv[1]) <= 1 && probe_pskip(S, m)) {
                    m->type = S_P_SKIP;
                    return;
                }
                cost += S->ref_cost[r];
                thresh += S->ref_cost[r];
                if (cost < best) { best = cost; bmx = mvx; bmy = mvy; bref = r; bmvp[0] = mvp[0]; bmvp[1] = mvp[1]; }
                S->mvr[((size_t)r * S->n + m->mb) * 2] = mvx; S->mvr[((size_t)r * S->n + m->mb) * 2 + 1] = mvy;
                m->l0mvc[r][0][0] = mvx; m->l0mvc[r][0][1] = mvy;
            }
            m->type = S_P_L0;
            cache_set(m, 0, 0, 4, 4, bref, 0, 0, 0);
            /* ---- sub-16x16 partitions (X264_ANALYSE_PSUB16x16), R/encoder/analyse.c:2222-2265 ---- */
            int cost8x8 = S_COST_MAX, cost16x8 = S_COST_MAX, cost8x16 = S_COST_MAX, part = S_D_16x16;
            for (int i = 0; i < 4; i++) { A->sub[i] = S_D_L0_8x8; A->cost_sub[i][0] = A->cost_sub[i][1] = A->cost_sub[i][2] = S_COST_MAX; }
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
            i_cost = best;
            if (p->inter & 0x10) {
                m->partition = S_D_8x8;
                if (p->mixed_refs) {                                 /* x264_mb_analyse_inter_p8x8_mixed_ref, :1146-1219 */
                    int maxref = S->n_ref - 1;
                    if (maxref > 0 && bref == 0 && m->type_top && m->type_left) {
                        static const int8_t look[6] = {3, 4, 6, 8, 11, 27};
                        maxref = 0;
                        for (int k = 0; k < 6; k++) if (m->cref[look[k]] > maxref) maxref = m->cref[look[k]];
                    }
                    for (int i = 0; i < 4; i++) {
                        A->me8[i].cost = 0x7fffffff;
                        for (int r = 0; r <= maxref; r++) {
                            me_ctx c; i16 mvp[2]; int mx, my, cmv = 0, cost;
                            cache_set(m, 2 * (i & 1), 2 * (i >> 1), 2, 2, r, 0, 0, 0);
                            predict_mv_blk(m, 4 * i, 2, mvp);
                            set_me_ctx_blk(S, m, r, mvp, &c, X264HIP_PIXEL_8x8, 8 * (i & 1), 8 * (i >> 1));
                            cost = me_search16(&c, mvp, (const i16 (*)[2])m->l0mvc[r], i + 1, p->me_method, p->me_range, p->subme, chroma_me, 0, &mx, &my, &cmv);
                            cost += S->ref_cost[r];
                            m->l0mvc[r][i + 1][0] = mx; m->l0mvc[r][i + 1][1] = my;
                            if (cost < A->me8[i].cost) { A->me8[i].cost = cost; A->me8[i].mvx = mx; A->me8[i].mvy = my; A->me8[i].cost_mv = cmv; A->me8[i].ref = r;
                                                      A->me8[i].ref_cost = S->ref_cost[r]; A->me8[i].mvp[0] = mvp[0]; A->me8[i].mvp[1] = mvp[1]; }
                        }
                        cache_set(m, 2 * (i & 1), 2 * (i >> 1), 2, 2, A->me8[i].ref, A->me8[i].mvx, A->me8[i].mvy, 1);
                        A->me8[i].cost += S->lambda * 1;                /* i_sub_mb_p_cost_table[D_L0_8x8] */
                    }
                    cost8x8 = A->me8[0].cost + A->me8[1].cost + A->me8[2].cost + A->me8[3].cost;
                    if (!p->cabac && !(A->me8[0].ref | A->me8[1].ref | A->me8[2].ref | A->me8[3].ref)) cost8x8 -= S->ref_cost[0] * 4;
                } else {                                             /* x264_mb_analyse_inter_p8x8, :1221-1272 */
                    const int r = bref, ref_cost = p->cabac || r ? S->ref_cost[r] : 0;
                    int n_mvc = 1;
                    m->l0mvc[r][0][0] = bmx; m->l0mvc[r][0][1] = bmy;
                    for (int i = 0; i < 4; i++) {
                        me_ctx c; i16 mvp[2]; int mx, my, cmv = 0, cost;
                        predict_mv_blk(m, 4 * i, 2, mvp);
                        set_me_ctx_blk(S, m, r, mvp, &c, X264HIP_PIXEL_8x8, 8 * (i & 1), 8 * (i >> 1));
                        cost = me_search16(&c, mvp, (const i16 (*)[2])m->l0mvc[r], n_mvc, p->me_method, p->me_range, p->subme, chroma_me, 0, &mx, &my, &cmv);
                        cache_set(m, 2 * (i & 1), 2 * (i >> 1), 2, 2, r, mx, my, 1);
                        m->l0mvc[r][n_mvc][0] = mx; m->l0mvc[r][n_mvc][1] = my; n_mvc++;
                        A->me8[i].cost = cost + ref_cost + S->lambda * 1; A->me8[i].mvx = mx; A->me8[i].mvy = my; A->me8[i].cost_mv = cmv; A->me8[i].ref = r;
                        A->me8[i].ref_cost = ref_cost; A->me8[i].mvp[0] = mvp[0]; A->me8[i].mvp[1] = mvp[1];
                    }
                    cost8x8 = A->me8[0].cost + A->me8[1].cost + A->me8[2].cost + A->me8[3].cost;
                    if (p->cabac) cost8x8 -= ref_cost;
                }
                if (cost8x8 < best) { m->type = S_P_8x8; part = S_D_8x8; i_cost = cost8x8; }
                if ((p->inter & 0x20) && m->type == S_P_8x8) {       /* X264_ANALYSE_PSUB8x8, :2252-2277 with _p4x4 / _p8x4 / _p4x8 (:1407-1519) */
                    static const int subw[3] = {1, 2, 1}, subh[3] = {1, 1, 2}, subn[3] = {4, 2, 2}, subpix[3] = {6, 4, 5}, subbits[3] = {5, 3, 3};
                    m->partition = S_D_8x8;
                    for (int i = 0; i < 4; i++) {
                        const int r = A->me8[i].ref, x0 = 2 * (i & 1), y0 = 2 * (i >> 1);
                        int c8 = 0, costs[3];
                        for (int t = 0; t < 3; t++) {                /* D_L0_4x4, then (only if that beats the 8x8) D_L0_8x4 and D_L0_4x8 */
                            sub_me *me = t == 0 ? A->me4[i] : t == 1 ? A->me84[i] : A->me48[i];
                            int sum = 0;
                            for (int k = 0; k < subn[t]; k++) {
                                const int x4 = x0 + (t == 0 ? (k & 1) : t == 2 ? k : 0), y4 = y0 + (t == 0 ? (k >> 1) : t == 1 ? k : 0), idx = 4 * i + (y4 - y0) * 2 + (x4 - x0);
                                me_ctx c; i16 mvp[2], mvc1[1][2]; int mx, my, cmv = 0;
                                mvc1[0][0] = t == 0 ? A->me8[i].mvx : A->me4[i][0].mvx; mvc1[0][1] = t == 0 ? A->me8[i].mvy : A->me4[i][0].mvy;
                                predict_mv_blk(m, idx, subw[t], mvp);
                                set_me_ctx_blk(S, m, r, mvp, &c, subpix[t], 4 * x4, 4 * y4);
                                me[k].cost = me_search16(&c, mvp, (const i16 (*)[2])mvc1, k == 0, p->me_method, p->me_range, p->subme, 0, 0, &mx, &my, &cmv);
                                me[k].mvx = mx; me[k].mvy = my; me[k].mvp[0] = mvp[0]; me[k].mvp[1] = mvp[1];
                                cache_set(m, x4, y4, subw[t], subh[t], r, mx, my, 1);
                                sum += me[k].cost;
                            }
                            costs[t] = sum + S->ref_cost[r] + S->lambda * subbits[t];
                            if (p->chroma_me && p->subme >= 5) costs[t] += p4x4_chroma(S, m, r, i, t, me);
                            A->cost_sub[i][t] = costs[t];
                            if (t == 0) {
                                if (!(costs[0] < A->me8[i].cost)) break;
                                c8 = costs[0]; A->sub[i] = S_D_L0_4x4;
                            } else if (costs[t] < c8) { c8 = costs[t]; A->sub[i] = t; }
                        }
                        if (A->sub[i] != S_D_L0_8x8) i_cost += c8 - A->me8[i].cost;
                        /* x264_mb_cache_mv_p8x8 */
                        if (A->sub[i] == S_D_L0_8x8) cache_set(m, x0, y0, 2, 2, r, A->me8[i].mvx, A->me8[i].mvy, 1);
                        else {
                            const int t = A->sub[i];
                            const sub_me *me = t == 0 ? A->me4[i] : t == 1 ? A->me84[i] : A->me48[i];
                            for (int k = 0; k < subn[t]; k++)
                                cache_set(m, x0 + (t == 0 ? (k & 1) : t == 2 ? k : 0), y0 + (t == 0 ? (k >> 1) : t == 1 ? k : 0), subw[t], subh[t], r, me[k].mvx, me[k].mvy, 1);
                        }
                    }
                    cost8x8 = i_cost;
                }
                const int thresh16x8 = A->me8[1].cost_mv + A->me8[2].cost_mv;
                if (cost8x8 < best + thresh16x8) {
                    for (int dir = 0; dir < 2; dir++) {              /* 0: x264_mb_analyse_inter_p16x8 (:1274), 1: _p8x16 (:1324) */
                        int sum = 0;
                        m->partition = dir ? S_D_8x16 : S_D_16x8;
                        for (int i = 0; i < 2; i++) {
                            const int ra = dir ? A->me8[i].ref : A->me8[2 * i].ref, rb = dir ? A->me8[i + 2].ref : A->me8[2 * i + 1].ref;
                            const int rr[2] = {ra, rb}, nr = ra == rb ? 1 : 2;
                            int bcost = 0x7fffffff, bx = 0, by = 0, br = 0, bcm = 0; i16 bp[2] = {0, 0};
                            for (int j = 0; j < nr; j++) {
                                const int r = rr[j];
                                me_ctx c; i16 mvp[2], mvc3[3][2]; int mx, my, cmv = 0, cost;
                                const int k1 = dir ? i + 1 : 2 * i + 1, k2 = dir ? i + 3 : 2 * i + 2;
                                mvc3[0][0] = m->l0mvc[r][0][0]; mvc3[0][1] = m->l0mvc[r][0][1];
                                mvc3[1][0] = m->l0mvc[r][k1][0]; mvc3[1][1] = m->l0mvc[r][k1][1];
                                mvc3[2][0] = m->l0mvc[r][k2][0]; mvc3[2][1] = m->l0mvc[r][k2][1];
                                if (dir) cache_set(m, 2 * i, 0, 2, 4, r, 0, 0, 0); else cache_set(m, 0, 2 * i, 4, 2, r, 0, 0, 0);
                                predict_mv_blk(m, dir ? 4 * i : 8 * i, dir ? 2 : 4, mvp);
                                set_me_ctx_blk(S, m, r, mvp, &c, dir ? X264HIP_PIXEL_8x16 : X264HIP_PIXEL_16x8, dir ? 8 * i : 0, dir ? 0 : 8 * i);
                                cost = me_search16(&c, mvp, (const i16 (*)[2])mvc3, 3, p->me_method, p->me_range, p->subme, chroma_me, 0, &mx, &my, &cmv);
                                cost += S->ref_cost[r];
                                if (cost < bcost) { bcost = cost; bx = mx; by = my; br = r; bcm = cmv; bp[0] = mvp[0]; bp[1] = mvp[1]; }
                            }
                            if (dir) cache_set(m, 2 * i, 0, 2, 4, br, bx, by, 1); else cache_set(m, 0, 2 * i, 4, 2, br, bx, by, 1);
                            pme *d = dir ? &A->me8x16[i] : &A->me16x8[i];
                            d->cost = bcost; d->mvx = bx; d->mvy = by; d->ref = br; d->cost_mv = bcm; d->ref_cost = S->ref_cost[br]; d->mvp[0] = bp[0]; d->mvp[1] = bp[1];
                            sum += bcost;
                        }
                        if (dir) cost8x16 = sum; else cost16x8 = sum;
                        if (sum < i_cost) { i_cost = sum; m->type = S_P_L0; part = dir ? S_D_8x16 : S_D_16x8; }
                    }
                }
            }
            m->partition = part;
            A->cost8x8 = cost8x8; A->cost16x8 = cost16x8; A->cost8x16 = cost8x16;
            /* x264_me_refine_qpel on the winning partition (:2289-2352; the reference cost leaves every block's sum, me.c:639-640);
             * with the RD levels the vectors stay as the searches left them ("refine later", :2296-2299) */
            if (S->mbrd) {
            } else if (part == S_D_16x16) {
                me_ctx c;
                set_me_ctx(S, m, bref, bmvp, &c);
                best -= S->ref_cost[bref];
                best = refine_qpel16(S, &c, best, &bmx, &bmy, bmvp);
                i_cost = best;
                for (int i = 0; i < 16; i++) { m->mv4[i][0] = bmx; m->mv4[i][1] = bmy; }
                for (int i = 0; i < 4; i++) m->ref8[i] = bref;
            } else {
                i_cost = 0;
                for (int i = 0; i < (part == S_D_8x8 ? 4 : 2); i++) {
                    if (part == S_D_8x8 && A->sub[i] != S_D_L0_8x8) {   /* the sub-8x8 blocks: no reference cost in their sums, no chroma (me.c:639, :654) */
                        static const int subw[3] = {1, 2, 1}, subh[3] = {1, 1, 2}, subn[3] = {4, 2, 2}, subpix[3] = {6, 4, 5};
                        const int t = A->sub[i], x0 = 2 * (i & 1), y0 = 2 * (i >> 1);
                        sub_me *me = t == 0 ? A->me4[i] : t == 1 ? A->me84[i] : A->me48[i];
                        for (int k = 0; k < subn[t]; k++) {
                            const int x4 = x0 + (t == 0 ? (k & 1) : t == 2 ? k : 0), y4 = y0 + (t == 0 ? (k >> 1) : t == 1 ? k : 0);
                            me_ctx c;
                            set_me_ctx_blk(S, m, A->me8[i].ref, me[k].mvp, &c, subpix[t], 4 * x4, 4 * y4);
                            me[k].cost = refine_qpel16(S, &c, me[k].cost, &me[k].mvx, &me[k].mvy, me[k].mvp);
                            i_cost += me[k].cost;
                            for (int y = 0; y < subh[t]; y++)
                                for (int x = 0; x < subw[t]; x++) { m->mv4[(y4 + y) * 4 + x4 + x][0] = me[k].mvx; m->mv4[(y4 + y) * 4 + x4 + x][1] = me[k].mvy; }
                        }
                        m->ref8[i] = A->me8[i].ref;
                        continue;
                    }
                    pme *d = part == S_D_8x8 ? &A->me8[i] : part == S_D_16x8 ? &A->me16x8[i] : &A->me8x16[i];
                    const int pix = part == S_D_8x8 ? X264HIP_PIXEL_8x8 : part == S_D_16x8 ? X264HIP_PIXEL_16x8 : X264HIP_PIXEL_8x16;
                    const int bx = part == S_D_8x8 ? 8 * (i & 1) : part == S_D_8x16 ? 8 * i : 0, by = part == S_D_8x8 ? 8 * (i >> 1) : part == S_D_16x8 ? 8 * i : 0;
                    const int w4 = part == S_D_16x8 ? 4 : 2, h4 = part == S_D_8x16 ? 4 : 2;
                    me_ctx c;
                    set_me_ctx_blk(S, m, d->ref, d->mvp, &c, pix, bx, by);
                    d->cost = refine_qpel16(S, &c, d->cost - d->ref_cost, &d->mvx, &d->mvy, d->mvp);
                    i_cost += d->cost;
                    for (int y = 0; y < h4; y++)
                        for (int x = 0; x < w4; x++) { m->mv4[(by / 4 + y) * 4 + bx / 4 + x][0] = d->mvx; m->mv4[(by / 4 + y) * 4 + bx / 4 + x][1] = d->mvy; }
                    for (int y = 0; y < h4 / 2; y++)
                        for (int x = 0; x < w4 / 2; x++) m->ref8[(by / 8 + y) * 2 + bx / 8 + x] = d->ref;
                }
            }
            m->mvx = bmx; m->mvy = bmy; m->ref = bref;
            for (int i = 0; i < 4; i++) m->sub[i] = part == S_D_8x8 ? A->sub[i] : S_D_L0_8x8;
            (void)cost16x8; (void)cost8x16;
            if (chroma_me) {
                analyse_intra_chroma(S, m);
                analyse_intra(S, m, i_cost - m->satd_chroma);
                m->satd_i16 += m->satd_chroma; m->satd_i8 += m->satd_chroma; m->satd_i4 += m->satd_chroma;
            } else
                analyse_intra(S, m, i_cost);
            int satd_inter = i_cost, satd_intra = m->satd_i16 < m->satd_i8 ? m->satd_i16 : m->satd_i8;
            if (m->satd_i4 < satd_intra) satd_intra = m->satd_i4;
            if (S->mbrd) {                                       /* :2375-2389 */
                analyse_p_rd(S, m, A, satd_inter < satd_intra ? satd_inter : satd_intra);
                m->type = S_P_L0; part = S_D_16x16; i_cost = A->me16.cost;
                if (A->cost16x8 < i_cost) { i_cost = A->cost16x8; part = S_D_16x8; }
                if (A->cost8x16 < i_cost) { i_cost = A->cost8x16; part = S_D_8x16; }
                if (A->cost8x8 < i_cost) { i_cost = A->cost8x8; part = S_D_8x8; m->type = S_P_8x8; }
                m->partition = part;
                if (i_cost < S_COST_MAX) transform_rd(S, m, A, &satd_inter, &i_cost);
                const int keep = m->type == S_P_SKIP ? (part == S_D_8x8 ? S_P_8x8 : S_P_L0) : m->type;   /* i_type is a local of the reference: the trial's P_SKIP does not stick */
                intra_rd(S, m, A, satd_inter * 5 / 4);
                m->type = keep;
            }
            int itype = S_I_16x16, icost = m->satd_i16;
            if (m->satd_i8 < icost) { icost = m->satd_i8; itype = S_I_8x8; }
            if (m->satd_i4 < icost) { icost = m->satd_i4; itype = S_I_4x4; }
            if (satd_pcm < icost) { icost = satd_pcm; itype = S_I_PCM; }
            if (icost < i_cost) { i_cost = icost; m->type = itype; }
            if (icost == S_COST_MAX) icost = i_cost * satd_intra / satd_inter + 1;
            S->stat_intra += icost; S->stat_inter += i_cost; S->stat_n++;
            if (S->mbrd >= 2 && m->type != S_I_PCM) refine_p_rd(S, m, A);      /* :2406-2464 */
        }
    }
}

/* x264_analyse_update_cache + x264_mb_analyse_transform (non-RD), R/encoder/analyse.c:2109-2126,2777-2826 */
static void update_mb(ssl *S, smb *m)
{
    switch (m->type) {
    case S_I_4x4:
        for (int i = 0; i < 16; i++) m->i4c[s_scan8(i)] = m->pred4[i];
        analyse_intra_chroma(S, m);
        break;
    case S_I_8x8:
        for (int i = 0; i < 16; i++) m->i4c[s_scan8(i)] = m->pred8[i >> 2];
        analyse_intra_chroma(S, m);
        break;
    case S_I_16x16:
        m->i16mode = m->pred16;
        analyse_intra_chroma(S, m);
        break;
    case S_P_SKIP:
        m->mvx = m->pskip_mv[0]; m->mvy = m->pskip_mv[1]; m->ref = 0; m->partition = S_D_16x16;
        for (int i = 0; i < 16; i++) { m->mv4[i][0] = m->pskip_mv[0]; m->mv4[i][1] = m->pskip_mv[1]; }
        memset(m->ref8, 0, 4);
        break;
    default:
        break;
    }
    /* x264_mb_transform_8x8_allowed (R/common/macroblock.h): a P_8x8 macroblock only with four 8x8 sub-partitions */
    if ((m->type == S_P_L0 || (m->type == S_P_8x8 && m->sub[0] == S_D_L0_8x8 && m->sub[1] == S_D_L0_8x8 && m->sub[2] == S_D_L0_8x8 && m->sub[3] == S_D_L0_8x8))
        && S->p->transform8x8 && !S->lossless) {
        mc_parts(S, m);
        int c8 = pixf.sa8d[X264HIP_PIXEL_16x16](m->fe[0], FENC, m->fd[0], FDEC);
        int c4 = pixf.satd[X264HIP_PIXEL_16x16](m->fe[0], FENC, m->fd[0], FDEC);
        m->t8 = c8 < c4;
        m->skip_mc = 1;
    }
}

/* x264_macroblock_encode, R/encoder/macroblock.c:475-790 */
static void mc_b(const ssl *S, smb *m);
static void encode_mb(ssl *S, smb *m)
{
    m->cbp_luma = 0; m->nnz[24] = 0;
    if (m->type == S_I_PCM) return;          /* the reference runs its inter branch on nothing here (no list is active); the writer stores the source */
    if (m->type == S_P_SKIP) {
        if (!m->skip_mc) {
            int mvx = m->mv4[0][0], mvy = m->mv4[0][1];          /* h->mb.cache.mv[0][x264_scan8[0]], macroblock.c:380-383 */
            mv_clip_frame(S, m, &mvx, &mvy);
            mc_16x16(S, m, 0, mvx, mvy);
        }
        m->cbp_luma = m->cbp_chroma = 0;
        memset(m->nnz, 0, sizeof(m->nnz));
        return;
    }
    if (m->type == S_B_SKIP) {                   /* macroblock.c:508-515 */
        if (!m->skip_mc) mc_b(S, m);
        m->cbp_luma = m->cbp_chroma = 0;
        memset(m->nnz, 0, sizeof(m->nnz));
        return;
    }
    if (m->type == S_I_16x16) {
        m->t8 = 0;
        pred_16x16(S, m, m->i16mode);
        enc_i16x16(S, m);
    } else if (!m->skip_intra && (m->type == S_I_8x8 || m->type == S_I_4x4)) {
        /* i_skip_intra = 0 (lossless, analyse.c:250; trellis 1 or --nr, :2772): nothing of the analysis' trial encode is kept, every block is predicted and coded again */
        u8 edge[40];
        m->t8 = m->type == S_I_8x8;
        if (m->type == S_I_8x8)
            for (int i = 0; i < 4; i++) {
                const int mode = m->i4c[s_scan8(4 * i)];
                s_p8filter(m->fd[0] + 8 * (i & 1) + 8 * (i >> 1) * FDEC, edge, m->nb8[i], s_pred4_nb[mode]);
                pred_8x8(S, m, i, mode, edge);
                enc_i8x8(S, m, i);
            }
        else
            for (int i = 0; i < 16; i++) {
                u8 *dst = m->fd[0] + blk_x[i] + blk_y[i] * FDEC;
                if ((m->nb4[i] & (NB_TOPRIGHT | NB_TOP)) == NB_TOP) memset(dst + 4 - FDEC, dst[3 - FDEC], 4);
                pred_4x4(S, m, i, m->i4c[s_scan8(i)]);
                enc_i4x4(S, m, i);
            }
    } else if (m->type == S_I_8x8) {
        u8 edge[40];
        m->t8 = 1;
        for (int r = 0; r < 16; r++) memcpy(m->fd[0] + r * FDEC, m->i8_fdec + 16 * r, 16);
        memcpy(m->nnz, m->i8_nnz, 16); m->cbp_luma = m->i8_cbp;
        if (m->skip_intra == 2) memcpy(m->luma8, m->i8_dct, sizeof(m->i8_dct));   /* "In RD mode, restore the now-overwritten DCT data", macroblock.c:543 */
        {
            u8 *dst = m->fd[0] + 8 + 8 * FDEC;
            int mode = m->i4c[s_scan8(12)];
            s_p8filter(dst, edge, m->nb8[3], s_pred4_nb[mode]);
            s_p8[mode](dst, edge);
            enc_i8x8(S, m, 3);
        }
    } else if (m->type == S_I_4x4) {
        m->t8 = 0;
        for (int r = 0; r < 16; r++) memcpy(m->fd[0] + r * FDEC, m->i4_fdec + 16 * r, 16);
        memcpy(m->nnz, m->i4_nnz, 16); m->cbp_luma = m->i4_cbp;
        if (m->skip_intra == 2) memcpy(m->luma4, m->i4_dct, sizeof(m->i4_dct));
        {
            u8 *dst = m->fd[0] + 12 + 12 * FDEC;
            if ((m->nb4[15] & (NB_TOPRIGHT | NB_TOP)) == NB_TOP) memset(dst + 4 - FDEC, dst[3 - FDEC], 4);
            s_p4[m->i4c[s_scan8(15)]](dst);
            enc_i4x4(S, m, 15);
        }
    } else {
        if (!m->skip_mc) { if (m->type >= S_B_DIRECT) mc_b(S, m); else mc_parts(S, m); }
        enc_inter_luma(S, m);
    }
    if (S_IS_INTRA(m->type)) pred_chroma(S, m, m->chroma_mode);
    enc_chroma(S, m, !S_IS_INTRA(m->type));
    if (m->type == S_P_L0 && m->partition == S_D_16x16 && !(m->cbp_luma | m->cbp_chroma) && m->mv4[0][0] == m->pskip_mv[0]
        && m->mv4[0][1] == m->pskip_mv[1] && m->ref8[0] == 0)
        m->type = S_P_SKIP;
    if (m->type == S_B_DIRECT && !(m->cbp_luma | m->cbp_chroma)) m->type = S_B_SKIP;   /* macroblock.c:784-788 */
}

/* x264_macroblock_cache_save (+ the copy-out the golden harness compares) */
static void save_mb(ssl *S, smb *m)
{
    slice_out *o = S->o;
    size_t M = (size_t)S->f * S->n + m->mb;
    int intra = S_IS_INTRA(m->type), cbp_dc = S->p->cabac ? (m->nnz[24] | m->nnz[25] << 1 | m->nnz[26] << 2) : 0;
    if (m->type == S_I_PCM) {                            /* R/common/macroblock.c:1245-1255 */
        m->qp = 0; S->last_dqp = 0; m->cbp_chroma = 2; m->cbp_luma = 0xf; m->t8 = 0; cbp_dc = 7;
        memset(m->nnz, 16, 27);                           /* the harness reports 16 for every entry of an I_PCM macroblock */
    } else {                                             /* :1268-1272: a macroblock without coefficients has no QP of its own */
        if (m->type != S_I_16x16 && m->cbp_luma == 0 && m->cbp_chroma == 0) m->qp = S->last_qp;
        S->last_dqp = m->qp - S->last_qp;
        S->last_qp = m->qp;
    }
    S->prev_mb = m->mb;
    if (S->cbp) {
        S->cbp[m->mb] = (i16)(S_IS_SKIP(m->type) ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma);
        S->chroma_pm[m->mb] = (int8_t)(intra && m->type != S_I_PCM ? s_fix8c[m->chroma_mode] : 0);
        S->qp_mb[m->mb] = (int8_t)m->qp;
        for (int i = 0; i < 16; i++) {
            const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2), keep = !intra && !S_IS_SKIP(m->type) && !S_IS_DIRECT(m->type);
            S->mvd[(m->mb * 16 + i) * 2] = keep ? m->cmvd[k][0] : 0; S->mvd[(m->mb * 16 + i) * 2 + 1] = keep ? m->cmvd[k][1] : 0;
            if (S->slice_type == S_SLICE_B) { S->mvd1[(m->mb * 16 + i) * 2] = keep ? m->cmvd1[k][0] : 0; S->mvd1[(m->mb * 16 + i) * 2 + 1] = keep ? m->cmvd1[k][1] : 0; }
        }
        if (S->slice_type == S_SLICE_B)                  /* macroblock.c:1354-1368 */
            S->skipbp[m->mb] = m->type == S_B_SKIP || m->type == S_B_DIRECT ? 0xf
                             : m->type == S_B_8x8 ? (m->sub[0] == S_D_DIRECT_8x8) | (m->sub[1] == S_D_DIRECT_8x8) << 1 | (m->sub[2] == S_D_DIRECT_8x8) << 2 | (m->sub[3] == S_D_DIRECT_8x8) << 3 : 0;
    }
    for (int pl = 0; pl < 3; pl++) {
        int w = pl ? 8 : 16, st = pl ? S->sc : S->sy;
        u8 *rec = S->fdec->plane[pl] + w * m->mby * st + w * m->mbx;
        u8 *out = (pl == 0 ? o->rec_y : pl == 1 ? o->rec_u : o->rec_v) + ((size_t)S->f * w * S->mb_h + w * m->mby) * w * S->mb_w + w * m->mbx;
        for (int y = 0; y < w; y++) { memcpy(rec + y * st, m->fd[pl] + y * FDEC, w); memcpy(out + (size_t)y * w * S->mb_w, m->fd[pl] + y * FDEC, w); }
    }
    S->fdec->mb_type[m->mb] = m->type == S_I_8x8 ? S_I_4x4 : m->type;
    if (m->type == S_I_4x4 || m->type == S_I_8x8) for (int i = 0; i < 16; i++) S->i4mode[m->mb * 16 + i] = m->i4c[s_scan8(i)];
    else memset(S->i4mode + m->mb * 16, 2, 16);
    if (m->cbp_luma == 0 && m->type != S_I_8x8) m->t8 = 0;
    S->t8[m->mb] = m->t8;
    memcpy(S->nnz + m->mb * 27, m->nnz, 27);
    for (int i = 0; i < 16; i++) {
        S->fdec->mv[(m->mb * 16 + i) * 2] = intra ? 0 : m->mv4[i][0];
        S->fdec->mv[(m->mb * 16 + i) * 2 + 1] = intra ? 0 : m->mv4[i][1];
    }
    for (int i = 0; i < 4; i++) S->fdec->ref[m->mb * 4 + i] = intra ? -1 : m->ref8[i];
    if (S->slice_type == S_SLICE_B) {
        for (int i = 0; i < 16; i++) {
            S->fdec->mv1[(m->mb * 16 + i) * 2] = intra ? 0 : m->mv4_1[i][0];
            S->fdec->mv1[(m->mb * 16 + i) * 2 + 1] = intra ? 0 : m->mv4_1[i][1];
        }
        for (int i = 0; i < 4; i++) S->fdec->ref1[m->mb * 4 + i] = intra ? -1 : m->ref8_1[i];
    }
    if (intra) S->intra_count++;

    o->mb_type[M] = m->type; o->partition[M] = intra || S_IS_SKIP(m->type) || m->type == S_B_DIRECT ? S_D_16x16 : m->partition;
    for (int i = 0; i < 4; i++) o->sub_partition[M * 4 + i] = m->type == S_P_8x8 || m->type == S_B_8x8 ? m->sub[i] : 0;
    memcpy(o->nnz + M * 27, m->nnz, 27);
    o->qp[M] = m->qp;
    o->cbp[M] = S_IS_SKIP(m->type) ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma;
    o->t8[M] = m->t8;
    o->i16mode[M] = m->type == S_I_16x16 ? m->i16mode : 0;
    o->chroma_mode[M] = intra ? m->chroma_mode : 0;
    memcpy(o->i4mode + M * 16, S->i4mode + m->mb * 16, 16);
    if (S->o2 && S->o2->mv1) {
        if (S->slice_type == S_SLICE_B) { memcpy(S->o2->mv1 + M * 32, S->fdec->mv1 + m->mb * 32, 64); memcpy(S->o2->ref1 + M * 4, S->fdec->ref1 + m->mb * 4, 4); }
        else { memset(S->o2->mv1 + M * 32, 0, 64); memset(S->o2->ref1 + M * 4, -1, 4); }
    }
    if (S->slice_type != S_SLICE_I) {
        memcpy(o->mv + M * 32, S->fdec->mv + m->mb * 32, 64);
        memcpy(o->ref + M * 4, S->fdec->ref + m->mb * 4, 4);
        for (int r = 0; r < S->n_ref; r++) {
            o->mvr[(((size_t)S->f * S->p->n_refs + r) * S->n + m->mb) * 2] = S->mvr[((size_t)r * S->n + m->mb) * 2];
            o->mvr[(((size_t)S->f * S->p->n_refs + r) * S->n + m->mb) * 2 + 1] = S->mvr[((size_t)r * S->n + m->mb) * 2 + 1];
        }
    } else
        memset(o->ref + M * 4, -1, 4);
    i16 *ly = o->luma + M * 256, *ldc = o->luma_dc + M * 16, *cdc = o->chroma_dc + M * 8, *cac = o->chroma_ac + M * 128;
    memset(ly, 0, 512); memset(ldc, 0, 32); memset(cdc, 0, 16); memset(cac, 0, 256);
    if (!S_IS_SKIP(m->type) && m->type != S_I_PCM) {
        if (m->type == S_I_16x16 && m->nnz[24]) memcpy(ldc, m->dc16, 32);
        if (m->t8) {
            for (int i = 0; i < 4; i++) if ((m->cbp_luma >> i & 1) && m->nnz[4 * i]) memcpy(ly + 64 * i, m->luma8[i], 128);
        } else
            for (int i = 0; i < 16; i++) if ((m->cbp_luma >> (i >> 2) & 1) && m->nnz[i]) memcpy(ly + 16 * i, m->luma4[i], 32);
        if (m->cbp_chroma) for (int i = 0; i < 2; i++) if (m->nnz[25 + i]) memcpy(cdc + 4 * i, m->cdc[i], 8);
        if (m->cbp_chroma == 2) for (int i = 0; i < 8; i++) if (m->nnz[16 + i]) memcpy(cac + 16 * i, m->cac[i], 32);
    }
}

/* ------------------------------------------------------------------ the chain */
/* per-QP tables of the current macroblock: h->quant4_mf[..][qp] etc., the lambdas and the mv / reference cost tables
 * (x264_mb_analyse_init + x264_mb_analyse_load_costs, R/encoder/analyse.c:220-232,182-218) */
static void set_mb_qp(ssl *S, smb *m, int qp)
{
    const slice_params *p = S->p;
    if (m) m->qp = qp;
    if (qp == S->qp && S->cost_mv) return;
    S->qp = qp;
    S->qpc = s_chroma_qp[clip3i(qp + (S->lossless ? 0 : S->chroma_qp_offset), 0, 51)];
    S->lambda = s_lambda_tab[qp]; S->lambda2 = s_lambda2_tab[qp];
    S->cost_mv = s_load_cost_mv(qp);
    for (int i = 0; i < 16; i++) S->ref_cost[i] = S->lambda * s_te_size(clip3i((S->n_ref <= 0 ? 1 : S->n_ref) - 1, 0, 2), i);
    for (int cat = 0; cat < 4; cat++) {
        x264o_cqm(p->cqm_preset, cat, cat < 2 ? S->qp : S->qpc, 0, S->mf4[cat], S->b4[cat], &S->dq4[cat][0][0]);
        x264o_cqm_unquant(p->cqm_preset, cat, cat < 2 ? S->qp : S->qpc, 0, S->unq4[cat]);
    }
    for (int cat = 0; cat < 2; cat++) {
        x264o_cqm(p->cqm_preset, cat, S->qp, 1, S->mf8[cat], S->b8[cat], &S->dq8[cat][0][0]);
        x264o_cqm_unquant(p->cqm_preset, cat, S->qp, 1, S->unq8[cat]);
    }
}
/* x264_adaptive_quant_frame, R/encoder/ratecontrol.c:231-249 (float, compiled like the reference: -ffp-contract=off) */
static void aq_frame(ssl *S)
{
    static const float log2_lut[128] = {
        0.00000, 0.01123, 0.02237, 0.03342, 0.04439, 0.05528, 0.06609, 0.07682, 0.08746, 0.09803, 0.10852, 0.11894, 0.12928, 0.13955, 0.14975, 0.15987,
        0.16993, 0.17991, 0.18982, 0.19967, 0.20945, 0.21917, 0.22882, 0.23840, 0.24793, 0.25739, 0.26679, 0.27612, 0.28540, 0.29462, 0.30378, 0.31288,
        0.32193, 0.33092, 0.33985, 0.34873, 0.35755, 0.36632, 0.37504, 0.38370, 0.39232, 0.40088, 0.40939, 0.41785, 0.42626, 0.43463, 0.44294, 0.45121,
        0.45943, 0.46761, 0.47573, 0.48382, 0.49185, 0.49985, 0.50779, 0.51570, 0.52356, 0.53138, 0.53916, 0.54689, 0.55459, 0.56224, 0.56986, 0.57743,
        0.58496, 0.59246, 0.59991, 0.60733, 0.61471, 0.62205, 0.62936, 0.63662, 0.64386, 0.65105, 0.65821, 0.66534, 0.67243, 0.67948, 0.68650, 0.69349,
        0.70044, 0.70736, 0.71425, 0.72110, 0.72792, 0.73471, 0.74147, 0.74819, 0.75489, 0.76155, 0.76818, 0.77479, 0.78136, 0.78790, 0.79442, 0.80090,
        0.80735, 0.81378, 0.82018, 0.82655, 0.83289, 0.83920, 0.84549, 0.85175, 0.85798, 0.86419, 0.87036, 0.87652, 0.88264, 0.88874, 0.89482, 0.90087,
        0.90689, 0.91289, 0.91886, 0.92481, 0.93074, 0.93664, 0.94251, 0.94837, 0.95420, 0.96000, 0.96578, 0.97154, 0.97728, 0.98299, 0.98868, 0.99435};
    const float strength = S->e->aq_strength * 1.0397;
    for (int mby = 0; mby < S->mb_h; mby++)
        for (int mbx = 0; mbx < S->mb_w; mbx++) {
            uint32_t energy = pixf.var[X264HIP_PIXEL_16x16](S->fenc->plane[0] + 16 * (mbx + mby * S->sy), S->sy)
                            + pixf.var[X264HIP_PIXEL_8x8](S->fenc->plane[1] + 8 * (mbx + mby * S->sc), S->sc)
                            + pixf.var[X264HIP_PIXEL_8x8](S->fenc->plane[2] + 8 * (mbx + mby * S->sc), S->sc);
            if (energy < 1) energy = 1;
            const int lz = __builtin_clz(energy);
            S->aq_offset[mbx + mby * S->mb_w] = strength * (log2_lut[(energy << lz >> 24) & 0x7f] - lz + 16.573f);
        }
}

static int s_encode_chain(const slice_params *p, const slice_ext *e, const u8 *src_y, const u8 *src_u, const u8 *src_v, slice_out *o, slice_out2 *o2)
{
    ssl S;
    const int b_write = e && e->write;
    sframe *refs[16] = {0};
    int n_avail = 0, last_idr = 0, cw = p->width / 2, chh = p->height / 2;
    s_setup();
    if (p->subme > 9 || p->me_method > 3 || (p->me_method == 3 && p->subme < 1)) return -3;   /* ESA at subme 0: the reference never fills the integral plane */
    if (p->subme > 5 && (!b_write || !p->cabac || p->qp == 0)) return -3;  /* RD levels: CABAC with the writer in the loop; not yet CAVLC / lossless */
    if (b_write && !p->cabac) return -3;
    if (e && e->psy_trellis != 0) return -3;
    const int nb = e ? clip3i(e->bframes, 0, 16) : 0;
    if (nb && (!b_write || p->qp == 0 || p->noise_reduction || (e->direct_pred != 1 && e->direct_pred != 2) || p->subme < 1)) return -3;   /* B slices: CABAC with the writer in the loop */
    memset(&S, 0, sizeof(S));
    S.p = p; S.o = o; S.e = e; S.o2 = o2;
    S.chroma_qp_offset = p->chroma_qp_offset;
    S.qp_min = p->cqm_preset ? 6 : 0; S.qp_max = 51;
    if (e) {                                 /* x264_validate_parameters, R/encoder/encoder.c:493-522 */
        const float psy = p->subme < 6 ? 0 : e->psy_rd < 0 ? 0 : e->psy_rd > 10 ? 10 : e->psy_rd;
        S.trellis = p->cabac ? clip3i(e->trellis, 0, 2) : 0;
        S.psy_rd = (int)(psy * (1 << 8) + .5);
        if (S.psy_rd) S.chroma_qp_offset -= psy < 0.25 ? 1 : 2;
        S.chroma_qp_offset = clip3i(S.chroma_qp_offset, -12, 12);
    }
    S.lossless = p->qp == 0;                 /* constant QP 0 = lossless (x264_validate_parameters) */
    g_me_lossless = S.lossless;
    S.mb_w = (p->width + 15) / 16; S.mb_h = (p->height + 15) / 16; S.n = S.mb_w * S.mb_h;
    S.w16 = 16 * S.mb_w; S.h16 = 16 * S.mb_h;
    S.sy = (S.w16 + 64 + 15) & ~15; S.sc = ((S.sy >> 1) + 15) & ~15;
    S.nnz = calloc(S.n, 27); S.i4mode = calloc(S.n, 16); S.t8 = calloc(S.n, 1);
    S.mvr = calloc((size_t)p->n_refs * S.n * 2, sizeof(i16));
    S.mvr1 = calloc((size_t)S.n * 2, sizeof(i16)); S.mvd1 = calloc((size_t)S.n * 32, sizeof(i16)); S.skipbp = calloc(S.n, 1);
    S.fenc = sframe_new(&S);
    if (e) {
        S.cbp = calloc(S.n, sizeof(i16)); S.chroma_pm = calloc(S.n, 1); S.mvd = calloc((size_t)S.n * 32, sizeof(i16)); S.qp_mb = calloc(S.n, 1);
        S.aq_offset = calloc(S.n, sizeof(float));
        if (b_write) S.bsbuf = malloc(64 + (size_t)e->payload_cap + 4096);
        {   /* scan position -> raster index, from the scan functions themselves; the trellis weights in scan order (R/common/dct.c:476-483) */
            static const u16 w4[3] = {800, 320, 128}, w8[6] = {256, 201, 656, 227, 410, 363};
            static const u8 k8[16] = {0, 3, 4, 3, 3, 1, 5, 1, 4, 5, 2, 5, 3, 1, 5, 1};
            i16 d4[4][4], l4[16], d8[8][8], l8[64];
            for (int i = 0; i < 16; i++) (&d4[0][0])[i] = (i16)i;
            zigf[0].scan_4x4(l4, d4);
            for (int i = 0; i < 16; i++) { S.zz4[i] = (u8)l4[i]; S.w4z[i] = w4[(l4[i] & 1) + ((l4[i] >> 2) & 1)]; }
            for (int i = 0; i < 64; i++) (&d8[0][0])[i] = (i16)i;
            zigf[0].scan_8x8(l8, d8);
            for (int i = 0; i < 64; i++) { S.zz8[i] = (u8)l8[i]; S.w8z[i] = w8[k8[((l8[i] >> 1) & 12) | (l8[i] & 3)]]; }
        }
    }
    /* coding order with a fixed pattern of nb disposable B frames (x264_slicetype_decide without b-adapt, then the reordering of
     * x264_encoder_encode, R/encoder/encoder.c:1390-1460): an anchor every nb + 1 frames after an IDR, the last frame before the next
     * IDR / the end of the clip is an anchor too, and every anchor is coded before the B frames it closes */
    int *order = malloc(sizeof(int) * (p->n_frames + 1)), *ftype = malloc(sizeof(int) * (p->n_frames + 1)), n_order = 0;
    for (int t = 0; t < p->n_frames;) {
        if (p->keyint > 0 ? t % p->keyint == 0 : t == 0) { order[n_order] = t; ftype[n_order++] = S_SLICE_I; t++; continue; }
        int lim = p->keyint > 0 ? (t / p->keyint + 1) * p->keyint : p->n_frames;
        if (lim > p->n_frames) lim = p->n_frames;
        const int anchor = t + nb < lim - 1 ? t + nb : lim - 1;
        order[n_order] = anchor; ftype[n_order++] = S_SLICE_P;
        for (int b = t; b < anchor; b++) { order[n_order] = b; ftype[n_order++] = S_SLICE_B; }
        t = anchor + 1;
    }
    const int dpb = p->n_refs > (nb ? 2 : 1) ? p->n_refs : (nb ? 2 : 1);   /* sps->vui.i_max_dec_frame_buffering, R/encoder/set.c:196-200 */
    for (int f = 0; f < p->n_frames; f++) {
        const int disp = order[f], idr = ftype[f] == S_SLICE_I, is_b = ftype[f] == S_SLICE_B;
        size_t F = f, D = disp;
        if (idr) { for (int i = 0; i < n_avail; i++) sframe_free(refs[i]); n_avail = 0; last_idr = disp; }
        for (int y = 0; y < p->height; y++) memcpy(S.fenc->plane[0] + y * S.sy, src_y + (D * p->height + y) * p->width, p->width);
        for (int y = 0; y < chh; y++) {
            memcpy(S.fenc->plane[1] + y * S.sc, src_u + (D * chh + y) * cw, cw);
            memcpy(S.fenc->plane[2] + y * S.sc, src_v + (D * chh + y) * cw, cw);
        }
        x264o_plane_pad_mod16(S.fenc->plane[0], S.sy, p->width, p->height, S.w16, S.h16);
        x264o_plane_pad_mod16(S.fenc->plane[1], S.sc, cw, chh, S.w16 / 2, S.h16 / 2);
        x264o_plane_pad_mod16(S.fenc->plane[2], S.sc, cw, chh, S.w16 / 2, S.h16 / 2);
        S.f = f;
        S.fdec = sframe_new(&S);
        S.fdec->poc = 2 * (disp - last_idr); S.fdec->kept = !is_b;
        S.lowres_mv[0] = S.lowres_mv[1] = NULL;
        if (e && e->lowres_mv && !idr) {
            const i16 *lm = e->lowres_mv + (size_t)F * 2 * S.n * 2;
            if (lm[0] != 0x7fff) S.lowres_mv[0] = lm;
            if (is_b && lm[2 * S.n] != 0x7fff) S.lowres_mv[1] = lm + 2 * S.n;
        }
        /* x264_reference_build_list, R/encoder/encoder.c:911-981: list 0 = earlier pictures, nearest first; list 1 = later pictures, nearest first */
        S.n_ref = S.n_ref1 = 0;
        for (int i = 0; i < n_avail; i++) {
            if (refs[i]->poc < S.fdec->poc) S.fref[S.n_ref++] = refs[i];
            else if (refs[i]->poc > S.fdec->poc && S.n_ref1 < 2) S.fref1[S.n_ref1++] = refs[i];
        }
        for (int i = 0; i < S.n_ref; i++)
            for (int k = i + 1; k < S.n_ref; k++)
                if (S.fref[k]->poc > S.fref[i]->poc) { sframe *t_ = S.fref[i]; S.fref[i] = S.fref[k]; S.fref[k] = t_; }
        if (S.n_ref1 == 2 && S.fref1[1]->poc < S.fref1[0]->poc) { sframe *t_ = S.fref1[0]; S.fref1[0] = S.fref1[1]; S.fref1[1] = t_; }
        if (S.n_ref1 > (nb ? 1 : 0)) S.n_ref1 = nb ? 1 : 0;     /* h->frames.i_max_ref1 */
        if (S.n_ref > p->n_refs) S.n_ref = p->n_refs;
        S.slice_type = ftype[f];
        S.direct_spatial = !e || e->direct_pred != 2;
        /* CQP: x264_ratecontrol_new / _start, R/encoder/ratecontrol.c:370-373,845-853 (ip_factor 1.4, pb_factor 1.3) */
        S.frame_qp = idr ? clip3i((int)(p->qp - 6.0 * log(1.4f) / log(2.0) + 0.5), 0, 51)
                   : is_b ? clip3i((int)(p->qp + 6.0 * log(1.3f) / log(2.0) + 0.5), 0, 51) : p->qp;
        S.mbrd = (p->subme - is_b >= 6) + (p->subme - is_b >= 8);   /* analyse.c:222-225: one level less in B slices */
        S.f_qpm = (float)S.frame_qp;                      /* rc->f_qpm = q, ratecontrol.c:868 (constant QP: an integer) */
        S.cost_mv = 0;
        set_mb_qp(&S, 0, S.frame_qp);
        const int b_aq = e && e->aq_mode > 0 && e->aq_strength != 0;
        if (b_aq) aq_frame(&S);
        /* x264_macroblock_slice_init, R/common/macroblock.c:771-808 */
        S.fdec->n_ref0 = S.n_ref;
        for (int i = 0; i < S.n_ref; i++) {
            int delta = S.fdec->poc - S.fref[i]->poc;
            S.fdec->ref_poc[i] = S.fref[i]->poc;
            S.fdec->inv_ref_poc[i] = (256 + delta / 2) / delta;
        }
        if (is_b) b_slice_init(&S, e);
        S.intra_count = 0; S.stat_intra = S.stat_inter = S.stat_n = 0;
        S.last_qp = S.frame_qp; S.last_dqp = 0; S.i_skip = 0;
        if (b_write) {                                    /* x264_slice_write, R/encoder/encoder.c:1155-1165 */
            memset(S.bsbuf, 0, 64 + (size_t)e->payload_cap + 4096);
            cb_context_init(&S.cb, S.slice_type, S.frame_qp, clip3i(e->cabac_init_idc, 0, 2));
            cb_encode_init(&S.cb, S.bsbuf + 64, S.bsbuf + 64 + e->payload_cap + 4096);
            S.cb.i_frame = f;
        }
        o->frame_info[4 * F] = S.slice_type; o->frame_info[4 * F + 1] = S.frame_qp; o->frame_info[4 * F + 2] = S.n_ref;
        o->frame_info[4 * F + 3] = S.fdec->poc;
        if (o2 && o2->frame_info2) { o2->frame_info2[4 * F] = disp; o2->frame_info2[4 * F + 1] = S.n_ref1; o2->frame_info2[4 * F + 2] = !is_b; o2->frame_info2[4 * F + 3] = 0; }
        for (int mb = 0; mb < S.n; mb++) {
            smb m;
            panalysis A;
            load_mb(&S, &m, mb % S.mb_w, mb / S.mb_w);
            /* x264_ratecontrol_qp + x264_adaptive_quant, R/encoder/analyse.c:2162-2164, ratecontrol.c:257-265 */
            int qp = S.frame_qp;
            if (b_aq) {
                qp = clip3i((int)(S.f_qpm + S.aq_offset[mb] + .5), S.qp_min, S.qp_max);
                if (abs(qp - S.last_qp) == 1) qp = S.last_qp;
            }
            set_mb_qp(&S, &m, qp);
            /* x264_mb_analyse_init, analyse.c:235-252 */
            S.b_trellis = S.trellis > 1 && S.mbrd;
            S.b_nr = 0;
            m.skip_intra = S.lossless ? 0 : S.mbrd ? 2 : !S.trellis && !p->noise_reduction;
            struct banalysis BA;
            memset(&A, 0, sizeof(A)); memset(&BA, 0, sizeof(BA));
            A.B = &BA;
            analyse_mb(&S, &m, &A);
            if (is_b) { update_cache(&S, &m, &A); if (!S.mbrd) analyse_transform_b(&S, &m); b_final_vectors(&m); }   /* :2763-2766 */
            else if (S.mbrd) update_cache(&S, &m, &A);     /* :2763 */
            else update_mb(&S, &m);
            S.b_trellis = S.trellis;                       /* :2768-2773 */
            S.b_nr = p->noise_reduction != 0;
            if (S.b_trellis == 1 || p->noise_reduction) m.skip_intra = 0;
            encode_mb(&S, &m);
            if (b_write) {                                 /* encoder.c:1192-1205 */
                if (mb > 0) cb_encode_terminal(&S.cb);
                if (S_IS_SKIP(m.type)) cw_mb_skip(&S, &S.cb, &m, 1);
                else {
                    if (S.slice_type != S_SLICE_I) cw_mb_skip(&S, &S.cb, &m, 0);
                    if (!S_IS_INTRA(m.type) && !is_b) for (int i = 0; i < 16; i++) {   /* the cache as x264_analyse_update_cache leaves it */
                        const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2);
                        m.cmv[k][0] = m.mv4[i][0]; m.cmv[k][1] = m.mv4[i][1]; m.cref[k] = m.ref8[(i >> 3) * 2 + ((i & 3) >> 1)];
                    }
                    cw_macroblock(&S, &S.cb, 0, &m);
                }
                o2->mb_bits[F * S.n + mb] = cb_pos(&S.cb);
                if (o2->mb_bits[F * S.n + mb] / 8 + 2048 > e->payload_cap) return -5;
            }
            memcpy(S.carry_nnz, m.nnz, 27);                 /* (before save_mb reports an I_PCM macroblock's counts as 16: the cache itself keeps the last trial's) */
            save_mb(&S, &m);
            for (int i = 0; i < 16; i++) {
                const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2);
                S.carry_mvd[0][i][0] = m.cmvd[k][0]; S.carry_mvd[0][i][1] = m.cmvd[k][1];
                if (is_b) { S.carry_mvd[1][i][0] = m.cmvd1[k][0]; S.carry_mvd[1][i][1] = m.cmvd1[k][1]; }
            }
            for (int l = 0; l < (is_b ? 2 : idr ? 0 : 1); l++) {      /* the cache entry the next macroblock inherits (see stale_ref) */
                const int k = s_scan8(12), inter = !S_IS_INTRA(m.type) && !is_b;
                S.stale_ref[l] = inter ? m.ref8[3] : CREF(&m, l)[k];
                S.stale_mv[l][0] = inter ? m.mv4[10][0] : CMV(&m, l)[k][0]; S.stale_mv[l][1] = inter ? m.mv4[10][1] : CMV(&m, l)[k][1];
            }
            if (o2) o2->qp_offset[F * S.n + mb] = b_aq ? S.aq_offset[mb] : 0;
        }
        if (b_write) {                                     /* encoder.c:1269-1273 */
            cb_encode_flush(&S.cb, f);
            const int len = (int)(S.cb.p - (S.bsbuf + 64));
            if (len > e->payload_cap) return -5;
            o2->payload_len[F] = len;
            memcpy(o2->payload + F * e->payload_cap, S.bsbuf + 64, len);
        }
        if (p->noise_reduction) nr_update(&S);
        o->stat[4 * F] = S.stat_intra; o->stat[4 * F + 1] = S.stat_inter; o->stat[4 * F + 2] = S.stat_n; o->stat[4 * F + 3] = 0;
        /* x264_fdec_filter_row over the finished frame: loop filter, borders, half-pel planes -- nothing of it for a disposable B frame (encoder.c:986-1024) */
        if (is_b) {
            for (int y = 0; y < S.h16; y++) memcpy(o->fin_y + (F * S.h16 + y) * S.w16, S.fdec->plane[0] + y * S.sy, S.w16);
            for (int y = 0; y < S.h16 / 2; y++) {
                memcpy(o->fin_u + (F * S.h16 / 2 + y) * (S.w16 / 2), S.fdec->plane[1] + y * S.sc, S.w16 / 2);
                memcpy(o->fin_v + (F * S.h16 / 2 + y) * (S.w16 / 2), S.fdec->plane[2] + y * S.sc, S.w16 / 2);
            }
            sframe_free(S.fdec);
            continue;
        }
        if (p->deblock) {
            u8 *t = malloc(S.n), *q = malloc(S.n), *t8 = malloc(S.n), *nz = malloc(S.n * 26);
            for (int mb = 0; mb < S.n; mb++) {
                int ty = o->mb_type[F * S.n + mb];
                t[mb] = S_IS_INTRA(ty) ? 1 : ty == S_P_SKIP ? 2 : ty == S_P_8x8 && (p->inter & 0x20) ? 3 : 0;
                q[mb] = (u8)o->qp[F * S.n + mb]; t8[mb] = (u8)S.t8[mb];
                memcpy(nz + mb * 26, S.nnz + mb * 27, 24); nz[mb * 26 + 24] = S.nnz[mb * 27 + 25]; nz[mb * 26 + 25] = S.nnz[mb * 27 + 26];
            }
            x264o_frame_deblock(S.fdec->plane[0], S.fdec->plane[1], S.fdec->plane[2], S.mb_w, S.mb_h, S.sy, S.sc, t, q, nz, t8,
                                S.fdec->mv, S.fdec->ref, p->alpha_c0, p->beta, S.chroma_qp_offset);
            free(t); free(q); free(t8); free(nz);
        }
        x264o_plane_expand_border(S.fdec->plane[0], S.sy, S.w16, S.h16, 32, 32);
        x264o_plane_expand_border(S.fdec->plane[1], S.sc, S.w16 / 2, S.h16 / 2, 16, 16);
        x264o_plane_expand_border(S.fdec->plane[2], S.sc, S.w16 / 2, S.h16 / 2, 16, 16);
        if (p->subme) x264o_frame_hpel(S.fdec->filt[0], S.fdec->filt[1], S.fdec->filt[2], S.fdec->filt[3], S.sy, S.w16, S.h16, S.mb_h);
        else for (int k = 1; k < 4; k++) S.fdec->filt[k] = S.fdec->filt[0];
        for (int y = 0; y < S.h16; y++) memcpy(o->fin_y + (F * S.h16 + y) * S.w16, S.fdec->plane[0] + y * S.sy, S.w16);
        for (int y = 0; y < S.h16 / 2; y++) {
            memcpy(o->fin_u + (F * S.h16 / 2 + y) * (S.w16 / 2), S.fdec->plane[1] + y * S.sc, S.w16 / 2);
            memcpy(o->fin_v + (F * S.h16 / 2 + y) * (S.w16 / 2), S.fdec->plane[2] + y * S.sc, S.w16 / 2);
        }
        for (int i = n_avail; i > 0; i--) refs[i] = refs[i - 1];
        refs[0] = S.fdec; n_avail++;
        if (n_avail > dpb) sframe_free(refs[--n_avail]);        /* x264_reference_update, encoder.c:1060-1093 */
    }
    free(order); free(ftype);
    for (int i = 0; i < n_avail; i++) sframe_free(refs[i]);
    sframe_free(S.fenc);
    free(S.nnz); free(S.i4mode); free(S.t8); free(S.mvr); free(S.mvr1); free(S.mvd1); free(S.skipbp);
    free(S.cbp); free(S.chroma_pm); free(S.mvd); free(S.qp_mb); free(S.aq_offset); free(S.bsbuf);
    return 0;
}

int x264o_encode_chain(const slice_params *p, const u8 *src_y, const u8 *src_u, const u8 *src_v, slice_out *o)
{
    return s_encode_chain(p, 0, src_y, src_u, src_v, o, 0);
}
int x264o_encode_chain2(const slice_params *p, const slice_ext *e, const u8 *src_y, const u8 *src_u, const u8 *src_v, slice_out *o, slice_out2 *o2)
{
    return s_encode_chain(p, e, src_y, src_u, src_v, o, o2);
}
