import sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:110]); sys.exit(1)
    s=s.replace(a,b)
# structs
rep('''    int write, payload_cap, cabac_init_idc;
} slice_ext;
typedef struct {
    u8 *payload;
    int32_t *payload_len, *mb_bits;
    float *qp_offset;
} slice_out2;''','''    int write, payload_cap, cabac_init_idc;
    int bframes, weightb, direct_pred;       /* B slices: param.i_bframe (fixed pattern, no pyramid), b_weighted_bipred, i_direct_mv_pred (1 spatial, 2 temporal) */
} slice_ext;
typedef struct {
    u8 *payload;
    int32_t *payload_len, *mb_bits;
    float *qp_offset;
    i16 *mv1; int8_t *ref1;                  /* list 1 */
    int32_t *frame_info2;                    /* [F][4]: display index, i_ref1, kept as reference, 0 */
} slice_out2;''')
rep('''enum { S_I_4x4 = 0, S_I_8x8 = 1, S_I_16x16 = 2, S_I_PCM = 3, S_P_L0 = 4, S_P_8x8 = 5, S_P_SKIP = 6 };
enum { S_D_L0_4x4 = 0, S_D_L0_8x4 = 1, S_D_L0_4x8 = 2, S_D_L0_8x8 = 3, S_D_8x8 = 13, S_D_16x8 = 14, S_D_8x16 = 15, S_D_16x16 = 16 };
enum { S_SLICE_P = 0, S_SLICE_I = 2 };''','''enum { S_I_4x4 = 0, S_I_8x8 = 1, S_I_16x16 = 2, S_I_PCM = 3, S_P_L0 = 4, S_P_8x8 = 5, S_P_SKIP = 6,
       S_B_DIRECT = 7, S_B_L0_L0 = 8, S_B_L0_L1 = 9, S_B_L0_BI = 10, S_B_L1_L0 = 11, S_B_L1_L1 = 12, S_B_L1_BI = 13,
       S_B_BI_L0 = 14, S_B_BI_L1 = 15, S_B_BI_BI = 16, S_B_8x8 = 17, S_B_SKIP = 18 };
enum { S_D_L0_4x4 = 0, S_D_L0_8x4 = 1, S_D_L0_4x8 = 2, S_D_L0_8x8 = 3, S_D_L1_8x8 = 7, S_D_BI_8x8 = 11, S_D_DIRECT_8x8 = 12,
       S_D_8x8 = 13, S_D_16x8 = 14, S_D_8x16 = 15, S_D_16x16 = 16 };
enum { S_SLICE_P = 0, S_SLICE_B = 1, S_SLICE_I = 2 };
#define S_IS_SKIP(t) ((t) == S_P_SKIP || (t) == S_B_SKIP)
#define S_IS_DIRECT(t) ((t) == S_B_DIRECT)''')
rep('''    int8_t *ref;              /* [n][4] */
    int poc, n_ref0, ref_poc[16], inv_ref_poc[16];
} sframe;''','''    int8_t *ref;              /* [n][4] */
    i16 *mv1; int8_t *ref1;   /* list 1 (B slices) */
    int poc, n_ref0, ref_poc[16], inv_ref_poc[16];
    int kept;                 /* b_kept_as_ref */
} sframe;''')
rep('''    uint32_t nr_sum[2][64], nr_count[2];
    uint16_t nr_offset[2][64];
} ssl;''','''    uint32_t nr_sum[2][64], nr_count[2];
    uint16_t nr_offset[2][64];
    /* B slices */
    sframe *fref1[2]; int n_ref1;            /* h->fref1 / h->i_ref1 */
    i16 *mvr1, *mvd1;                        /* h->mb.mvr[1][0], h->mb.mvd[1] */
    u8 *skipbp;                              /* h->mb.skipbp */
    int direct_spatial;                      /* sh.b_direct_spatial_mv_pred */
    int bipred_weight[16][2], dist_scale[16][2];   /* h->mb.bipred_weight / dist_scale_factor (x264_macroblock_bipred_init) */
    int8_t map_col_store[18];                /* h->mb.map_col_to_list0 with its -1 / -2 entries */
    int ref_cost1[2];                        /* a->p_cost_ref1 */
} ssl;''')
rep('''    int fenc_satd[4][4], fenc_sa8d[2][2], fenc_satd_sum, fenc_sa8d_sum;   /* h->mb.pic.fenc_satd ... (psy-RD) */
} smb;''','''    int fenc_satd[4][4], fenc_sa8d[2][2], fenc_satd_sum, fenc_sa8d_sum;   /* h->mb.pic.fenc_satd ... (psy-RD) */
    /* B slices: list 1 of the caches, the skip flags of direct blocks, the direct prediction, the final list-1 vectors */
    int8_t cref1[48]; i16 cmv1[48][2], cmvd1[48][2];
    int8_t cskip[48];                    /* h->mb.cache.skip */
    int8_t direct_ref[2][4]; i16 direct_mv[2][16][2];   /* h->mb.cache.direct_ref / direct_mv (the 16 blocks in raster order) */
    i16 mv4_1[16][2]; int8_t ref8_1[4];
} smb;
#define CREF(m_, l_) ((l_) ? (m_)->cref1 : (m_)->cref)
#define CMV(m_, l_) ((l_) ? (m_)->cmv1 : (m_)->cmv)
#define CMVD(m_, l_) ((l_) ? (m_)->cmvd1 : (m_)->cmvd)''')
rep('''typedef struct {
    pme me16, me8[4], me16x8[2], me8x16[2];
    sub_me me4[4][4], me84[4][2], me48[4][2];
    int sub[4];
    int cost8x8, cost16x8, cost8x16, rd16;
} panalysis;''','''struct banalysis;
typedef struct {
    pme me16, me8[4], me16x8[2], me8x16[2];
    sub_me me4[4][4], me84[4][2], me48[4][2];
    int sub[4];
    int cost8x8, cost16x8, cost8x16, rd16;
    struct banalysis *B;                 /* the B-slice half of x264_mb_analysis_t (b_oracle.c) */
} panalysis;''')
rep('''    f->mb_type = calloc(S->n, 1); f->mv = calloc(S->n * 32, sizeof(i16)); f->ref = calloc(S->n * 4, 1);
    return f;''','''    f->mb_type = calloc(S->n, 1); f->mv = calloc(S->n * 32, sizeof(i16)); f->ref = calloc(S->n * 4, 1);
    f->mv1 = calloc(S->n * 32, sizeof(i16)); f->ref1 = calloc(S->n * 4, 1);
    return f;''')
rep('''    free(f->mb_type); free(f->mv); free(f->ref); free(f);''','''    free(f->mb_type); free(f->mv); free(f->ref); free(f->mv1); free(f->ref1); free(f);''')
# decimation in B slices
rep('''    int b_decimate = S->p->dct_decimate && S->slice_type == S_SLICE_P, score = b_decimate ? 0 : 9, nz;''',
    '''    int b_decimate = S->slice_type == S_SLICE_B || (S->p->dct_decimate && S->slice_type == S_SLICE_P), score = b_decimate ? 0 : 9, nz;   /* macroblock.c:193 */''')
rep('''    int cat = 2 + b_inter, qpc = S->qpc, b_decimate = b_inter && S->p->dct_decimate;''','''    int cat = 2 + b_inter, qpc = S->qpc, b_decimate = b_inter && (S->slice_type == S_SLICE_B || S->p->dct_decimate);   /* macroblock.c:275 */''')
rep('''    int b_decimate = S->p->dct_decimate, decimate_mb = 0;''','''    int b_decimate = S->slice_type == S_SLICE_B || S->p->dct_decimate, decimate_mb = 0;   /* macroblock.c:479 */''')
# predict_mv_blk: list-generic
rep('''static void cache_set(smb *m, int x, int y, int w, int h, int ref, int mvx, int mvy, int set_mv)
{
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            int k = 4 + 1 * 8 + x + i + 8 * (y + j);
            m->cref[k] = (int8_t)ref;
            if (set_mv) { m->cmv[k][0] = (i16)mvx; m->cmv[k][1] = (i16)mvy; }
        }
}''','''static void cache_set_l(smb *m, int list, int x, int y, int w, int h, int ref, int mvx, int mvy, int set_mv)
{
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            int k = 4 + 1 * 8 + x + i + 8 * (y + j);
            CREF(m, list)[k] = (int8_t)ref;
            if (set_mv) { CMV(m, list)[k][0] = (i16)mvx; CMV(m, list)[k][1] = (i16)mvy; }
        }
}
static void cache_set(smb *m, int x, int y, int w, int h, int ref, int mvx, int mvy, int set_mv) { cache_set_l(m, 0, x, y, w, h, ref, mvx, mvy, set_mv); }''')
rep('''static void predict_mv_blk(const smb *m, int idx, int width, i16 mvp[2])
{
    const int i8 = s_scan8(idx), i_ref = m->cref[i8];
    int ra = m->cref[i8 - 1], rb = m->cref[i8 - 8], rc = m->cref[i8 - 8 + width], cnt;
    const i16 *a = m->cmv[i8 - 1], *b = m->cmv[i8 - 8], *c = m->cmv[i8 - 8 + width];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { rc = m->cref[i8 - 8 - 1]; c = m->cmv[i8 - 8 - 1]; }''','''static void predict_mv_blk_l(const smb *m, int list, int idx, int width, i16 mvp[2])
{
    const int8_t *cref = CREF(m, list);
    const i16 (*cmv)[2] = CMV(m, list);
    const int i8 = s_scan8(idx), i_ref = cref[i8];
    int ra = cref[i8 - 1], rb = cref[i8 - 8], rc = cref[i8 - 8 + width], cnt;
    const i16 *a = cmv[i8 - 1], *b = cmv[i8 - 8], *c = cmv[i8 - 8 + width];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { rc = cref[i8 - 8 - 1]; c = cmv[i8 - 8 - 1]; }''')
rep('''    else { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
}
/* x264_mb_transform_8x8_allowed''','''    else { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
}
static void predict_mv_blk(const smb *m, int idx, int width, i16 mvp[2]) { predict_mv_blk_l(m, 0, idx, width, mvp); }
/* x264_mb_transform_8x8_allowed''')
rep('''    if (!S->p->transform8x8) return 0;
    if (m->type == S_P_L0) return 1;''','''    if (!S->p->transform8x8) return 0;
    if (m->type == S_P_L0 || (m->type >= S_B_DIRECT && m->type <= S_B_8x8)) return 1;   /* every B type but B_SKIP (direct_8x8_inference is on) */''')
# me ctx: list-generic
rep('''static void set_me_ctx_blk(const ssl *S, const smb *m, int ref, const i16 mvp[2], me_ctx *c, int pix, int bx, int by)
{
    static const u8 bw[7] = {16, 16, 8, 8, 8, 4, 4}, bh[7] = {16, 8, 16, 8, 4, 8, 4};
    int oy = (16 * m->mby + by) * S->sy + 16 * m->mbx + bx, oc = (8 * m->mby + by / 2) * S->sc + 8 * m->mbx + bx / 2, sp[4], fp[4];
    const sframe *r = S->fref[ref];''','''static void set_me_ctx_blk_l(const ssl *S, const smb *m, int list, int ref, const i16 mvp[2], me_ctx *c, int pix, int bx, int by)
{
    static const u8 bw[7] = {16, 16, 8, 8, 8, 4, 4}, bh[7] = {16, 8, 16, 8, 4, 8, 4};
    int oy = (16 * m->mby + by) * S->sy + 16 * m->mbx + bx, oc = (8 * m->mby + by / 2) * S->sc + 8 * m->mbx + bx / 2, sp[4], fp[4];
    const sframe *r = list ? S->fref1[ref] : S->fref[ref];''')
rep('''static void set_me_ctx(const ssl *S, const smb *m, int ref, const i16 mvp[2], me_ctx *c) { set_me_ctx_blk(S, m, ref, mvp, c, X264HIP_PIXEL_16x16, 0, 0); }''',
'''static void set_me_ctx_blk(const ssl *S, const smb *m, int ref, const i16 mvp[2], me_ctx *c, int pix, int bx, int by) { set_me_ctx_blk_l(S, m, 0, ref, mvp, c, pix, bx, by); }
static void set_me_ctx(const ssl *S, const smb *m, int ref, const i16 mvp[2], me_ctx *c) { set_me_ctx_blk(S, m, ref, mvp, c, X264HIP_PIXEL_16x16, 0, 0); }''')
# refine_qpel chroma only in P slices
rep('''chroma_me = S->p->chroma_me && subme >= 5 && c->pix <= X264HIP_PIXEL_8x8;   /* b_chroma_me && i_pixel <= PIXEL_8x8, me.c:654 */''',
    '''chroma_me = S->p->chroma_me && S->slice_type == S_SLICE_P && subme >= 5 && c->pix <= X264HIP_PIXEL_8x8;   /* b_chroma_me (P slices only, analyse.c:234) && i_pixel <= PIXEL_8x8, me.c:654 */''')
open(p,'w').write(s)
print("ok")
