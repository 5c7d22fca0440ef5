/* refine_oracle.c -- TEST INFRASTRUCTURE; textually included by slice_oracle.c after b_oracle.c.
 *
 * CPU restatement of the reference's RD refinement (subme 8-9, a->i_mbrd >= 2) and of the partial RD costs it is built on:
 *   x264_macroblock_encode_p8x8 / _p4x4            R/encoder/macroblock.c:917-1077   (RD-only partial encodes)
 *   x264_rd_cost_part / _subpart / _i8x8 / _i4x4 / _i8x8_chroma     R/encoder/rdo.c:173-315
 *   x264_me_refine_qpel_rd                         R/encoder/me.c:961-1047
 *   x264_me_refine_bidir_rd                        R/encoder/me.c:780-933 (the rd = 1 form of x264_me_refine_bidir)
 *   x264_intra_rd_refine                           R/encoder/analyse.c:876-1056
 *   the sub-8x8 branch of x264_mb_analyse_p_rd     R/encoder/analyse.c:1968-1996
 * and the call sites in x264_macroblock_analyse (analyse.c:2406-2464 for P, :2702-2758 for B).
 * State that is "left over" matters here and is kept exactly as the reference leaves it: a partial encode overwrites only its own
 * blocks' levels / non_zero_count / cbp bits, and the bit counters read whatever the previous trial left in the blocks around it
 * (the reference's own FIXME, analyse.c:1976-1977).                                                                            */

static const u8 r_pix_w4[7] = {4, 4, 2, 2, 2, 1, 1}, r_pix_h4[7] = {4, 2, 4, 2, 1, 2, 1};      /* x264_pixel_size in 4x4 blocks */

/* x264_mb_mc_8x8 (R/common/macroblock.c:561-600): the 8x8 block's prediction from the motion caches, by its sub-partition type */
static void mc_8x8(const ssl *S, smb *m, int i8)
{
    const int x = 2 * (i8 & 1), y = 2 * (i8 >> 1);
    switch (m->sub[i8]) {
    case S_D_L0_8x8: mc_b_part(S, m, x, y, 2, 2, 1); break;
    case S_D_L0_8x4: mc_b_part(S, m, x, y, 2, 1, 1); mc_b_part(S, m, x, y + 1, 2, 1, 1); break;
    case S_D_L0_4x8: mc_b_part(S, m, x, y, 1, 2, 1); mc_b_part(S, m, x + 1, y, 1, 2, 1); break;
    case S_D_L0_4x4: mc_b_part(S, m, x, y, 1, 1, 1); mc_b_part(S, m, x + 1, y, 1, 1, 1); mc_b_part(S, m, x, y + 1, 1, 1, 1); mc_b_part(S, m, x + 1, y + 1, 1, 1, 1); break;
    case S_D_L1_8x8: mc_b_part(S, m, x, y, 2, 2, 2); break;
    case S_D_BI_8x8: mc_b_part(S, m, x, y, 2, 2, 3); break;
    case S_D_DIRECT_8x8: mc_b_direct8x8(S, m, x, y); break;
    default: break;
    }
}
static void store_8x8_nnz(smb *m, int i8, int nz) { for (int k = 0; k < 4; k++) m->nnz[4 * i8 + k] = (u8)nz; }

/* x264_macroblock_encode_p8x8, macroblock.c:917-1042 (not lossless: the RD levels are refused there) */
static void encode_p8x8(ssl *S, smb *m, int i8)
{
    const int x = 8 * (i8 & 1), y = 8 * (i8 >> 1);
    const int b_decimate = S->slice_type == S_SLICE_B || S->p->dct_decimate;
    u8 *fe = m->fe[0] + x + y * FENC, *fd = m->fd[0] + x + y * FDEC;
    int nnz8x8 = 0;
    mc_8x8(S, m, i8);
    if (m->t8) {
        i16 d8[8][8];
        dctf.sub8x8_dct8(d8, fe, fd);
        nnz8x8 = q8(S, d8, 1, 0, S->qp);
        if (nnz8x8) {
            zigf[0].scan_8x8(m->luma8[i8], d8);
            if (b_decimate && !S->b_trellis) nnz8x8 = 4 <= quantf.decimate_score64(m->luma8[i8]);
            if (nnz8x8) {
                quantf.dequant_8x8(d8, (int (*)[8][8])S->dq8[1], S->qp);
                dctf.add8x8_idct8(fd, d8);
                store_8x8_nnz(m, i8, 1);
            } else store_8x8_nnz(m, i8, 0);
        } else store_8x8_nnz(m, i8, 0);
    } else {
        i16 d4[4][4][4];
        int dec = 0;
        dctf.sub8x8_dct(d4, fe, fd);
        for (int i4 = 0; i4 < 4; i4++) {
            const int nz = q4(S, d4[i4], 1, 2, 0, S->qp);
            m->nnz[4 * i8 + i4] = (u8)nz;
            if (nz) {
                zigf[0].scan_4x4(m->luma4[4 * i8 + i4], d4[i4]);
                quantf.dequant_4x4(d4[i4], (int (*)[4][4])S->dq4[1], S->qp);
                if (b_decimate) dec += quantf.decimate_score16(m->luma4[4 * i8 + i4]);
                nnz8x8 = 1;
            }
        }
        if (b_decimate && dec < 4) nnz8x8 = 0;
        if (nnz8x8) dctf.add8x8_idct(fd, d4);
        else store_8x8_nnz(m, i8, 0);
    }
    for (int ch = 0; ch < 2; ch++) {
        i16 d[4][4];
        u8 *ce = m->fe[1 + ch] + (i8 & 1) * 4 + (i8 >> 1) * 4 * FENC, *cd = m->fd[1 + ch] + (i8 & 1) * 4 + (i8 >> 1) * 4 * FDEC;
        dctf.sub4x4_dct(d, ce, cd);
        d[0][0] = 0;
        const int nz = q4(S, d, 3, 4, 0, S->qpc);
        m->nnz[16 + i8 + 4 * ch] = (u8)nz;
        if (nz) {
            zigf[0].scan_4x4(m->cac[4 * ch + i8], d);
            quantf.dequant_4x4(d, (int (*)[4][4])S->dq4[3], S->qpc);
            dctf.add4x4_idct(cd, d);
        }
    }
    m->cbp_luma &= ~(1 << i8);
    m->cbp_luma |= nnz8x8 << i8;
    m->cbp_chroma = 2;
}
/* x264_macroblock_encode_p4x4, macroblock.c:1047-1077: luma only, list 0 */
static void encode_p4x4(ssl *S, smb *m, int i4)
{
    u8 *fe = m->fe[0] + blk_x[i4] + blk_y[i4] * FENC, *fd = m->fd[0] + blk_x[i4] + blk_y[i4] * FDEC;
    const int k = s_scan8(i4), ref = m->cref[k];
    const sframe *r = S->fref[ref];
    int mvx = m->cmv[k][0], mvy = m->cmv[k][1];
    const int o = (16 * m->mby + blk_y[i4]) * S->sy + 16 * m->mbx + blk_x[i4];
    u8 *src4[4] = {r->filt[0] + o, r->filt[1] + o, r->filt[2] + o, r->filt[3] + o};
    i16 d[4][4];
    mv_clip_frame(S, m, &mvx, &mvy);
    mcf.mc_luma(fd, FDEC, src4, S->sy, mvx, mvy, 4, 4);
    dctf.sub4x4_dct(d, fe, fd);
    const int nz = q4(S, d, 1, 2, 0, S->qp);
    m->nnz[i4] = (u8)nz;
    if (nz) {
        zigf[0].scan_4x4(m->luma4[i4], d);
        quantf.dequant_4x4(d, (int (*)[4][4])S->dq4[1], S->qp);
        dctf.add4x4_idct(fd, d);
    }
}

/* the partition RD costs carry 8 more bits than x264_rd_cost_mb's (rdo.c:171) */
static uint64_t rd_bits8(const o_cabac *tmp, int lambda2) { return ((uint64_t)tmp->f8 * lambda2 + 128) >> 8; }
static void rd_tmp_init(const ssl *S, o_cabac *tmp) { tmp->f8 = 0; memcpy(tmp->state, S->cb.state, 460); }      /* COPY_CABAC */

#ifdef X264O_DEVCHECK
/* the product's device text of the partial writers (cabac_dev.h built for the host, oracle/devcheck.cpp) replayed on a copy of the state */
void devhost_cw_part(DCabac *cb, uint8_t *st, MbSyn *m, int kind, int a, int b);
static void part_chk_begin(const ssl *S, const smb *m, MbSyn *y, u8 *st) { mbsyn_fill(S, m, y); memcpy(st, S->cb.state, 460); }
static void part_chk_end(const smb *m, const o_cabac *tmp, MbSyn *y, u8 *st, int kind, int a, int b)
{
    DCabac d = {0, 0x1FE, -1, 0, 0, 0};
    devhost_cw_part(&d, st, y, kind, a, b);
    g_devcheck_calls++;
    if (d.f8 != tmp->f8 || memcmp(st, tmp->state, 460) || memcmp(y->cmvd, m->cmvd, sizeof(y->cmvd)) || memcmp(y->cmvd1, m->cmvd1, sizeof(y->cmvd1))) {
        if (!g_devcheck_bad) fprintf(stderr, "devcheck: partial writer %d (%d, %d) differs: f8 %d vs %d\n", kind, a, b, d.f8, tmp->f8);
        g_devcheck_bad++;
    }
}
#define PART_CHK_BEGIN MbSyn y_; u8 st_[460]; part_chk_begin(S, m, &y_, st_)
#define PART_CHK_END(kind_, a_, b_) part_chk_end(m, &tmp, &y_, st_, kind_, a_, b_)
#else
#define PART_CHK_BEGIN do {} while (0)
#define PART_CHK_END(kind_, a_, b_) do {} while (0)
#endif
static uint64_t rd_cost_subpart(ssl *S, smb *m, int lambda2, int i4, int pix)
{   /* rdo.c:173-200 */
    o_cabac tmp;
    encode_p4x4(S, m, i4);
    if (pix == X264HIP_PIXEL_8x4) encode_p4x4(S, m, i4 + 1);
    if (pix == X264HIP_PIXEL_4x8) encode_p4x4(S, m, i4 + 2);
    const uint64_t ssd = (uint64_t)ssd_plane(S, m, pix, 0, blk_x[i4], blk_y[i4]);
    rd_tmp_init(S, &tmp);
    PART_CHK_BEGIN;
    cw_subpartition_size(&tmp, m, i4, pix);
    PART_CHK_END(1, i4, pix);
    return (ssd << 8) + rd_bits8(&tmp, lambda2);
}
static uint64_t rd_cost_part(ssl *S, smb *m, int lambda2, int i4, int pix)
{   /* rdo.c:202-242 */
    int i8 = i4 >> 2;
    o_cabac tmp;
    if (pix == X264HIP_PIXEL_16x16) {
        const int type_bak = m->type, c = rd_cost_mb(S, m, lambda2);
        m->type = type_bak;
        return (uint64_t)c;
    }
    if (pix > X264HIP_PIXEL_8x8) return rd_cost_subpart(S, m, lambda2, i4, pix);
    m->cbp_luma = 0;
    encode_p8x8(S, m, i8);
    if (pix == X264HIP_PIXEL_16x8) encode_p8x8(S, m, i8 + 1);
    if (pix == X264HIP_PIXEL_8x16) encode_p8x8(S, m, i8 + 2);
    const uint64_t ssd = (uint64_t)ssd_plane(S, m, pix, 0, (i8 & 1) * 8, (i8 >> 1) * 8)
                       + (uint64_t)ssd_plane(S, m, pix + 3, 1, (i8 & 1) * 4, (i8 >> 1) * 4) + (uint64_t)ssd_plane(S, m, pix + 3, 2, (i8 & 1) * 4, (i8 >> 1) * 4);
    rd_tmp_init(S, &tmp);
    PART_CHK_BEGIN;
    cw_partition_size(S, &tmp, m, i8, pix);
    PART_CHK_END(0, i8, pix);
    return (ssd << 8) + rd_bits8(&tmp, lambda2);
}
static uint64_t rd_cost_i8x8(ssl *S, smb *m, int lambda2, int i8, int mode)
{   /* rdo.c:244-266 */
    o_cabac tmp;
    m->cbp_luma &= ~(1 << i8);
    m->t8 = 1;
    enc_i8x8(S, m, i8);
    const uint64_t ssd = (uint64_t)ssd_plane(S, m, X264HIP_PIXEL_8x8, 0, (i8 & 1) * 8, (i8 >> 1) * 8);
    rd_tmp_init(S, &tmp);
    PART_CHK_BEGIN;
    cw_partition_i8x8_size(&tmp, m, i8, mode);
    PART_CHK_END(2, i8, mode);
    return (ssd << 8) + rd_bits8(&tmp, lambda2);
}
static uint64_t rd_cost_i4x4(ssl *S, smb *m, int lambda2, int i4, int mode)
{   /* rdo.c:268-288 */
    o_cabac tmp;
    enc_i4x4(S, m, i4);
    const uint64_t ssd = (uint64_t)ssd_plane(S, m, X264HIP_PIXEL_4x4, 0, blk_x[i4], blk_y[i4]);
    rd_tmp_init(S, &tmp);
    PART_CHK_BEGIN;
    cw_partition_i4x4_size(&tmp, m, i4, mode);
    PART_CHK_END(3, i4, mode);
    return (ssd << 8) + rd_bits8(&tmp, lambda2);
}
static uint64_t rd_cost_i8x8_chroma(ssl *S, smb *m, int lambda2, int mode, int b_dct)
{   /* rdo.c:290-315 */
    o_cabac tmp;
    if (b_dct) enc_chroma(S, m, 0);
    const uint64_t ssd = (uint64_t)ssd_plane(S, m, X264HIP_PIXEL_8x8, 1, 0, 0) + (uint64_t)ssd_plane(S, m, X264HIP_PIXEL_8x8, 2, 0, 0);
    m->chroma_mode = mode;
    rd_tmp_init(S, &tmp);
    PART_CHK_BEGIN;
    cw_i8x8_chroma_size(&tmp, m);
    PART_CHK_END(4, 0, 0);
    return (ssd << 8) + rd_bits8(&tmp, lambda2);
}

/* ------------------------------------------------------------------ x264_me_refine_qpel_rd, me.c:961-1047
 * one block's vector (list `list`, block i4, size pix) by rate-distortion: the SATD of a candidate decides whether its RD cost
 * is worth computing; the candidates go into the motion cache (the two entries the partial encoders read) for the trial. */
#define R_COST_MAX64 ((uint64_t)1 << 60)
static void refine_qpel_rd(ssl *S, smb *m, int list, int pix, int i4, int *pmvx, int *pmvy, int *pcost, i16 mvp[2])
{
    static const u8 offs[7] = {0, 2, 16, 0, 1, 8, 0};                  /* pixel_mv_offs in cache entries */
    static const int8_t hex2[8][2] = {{-1, -2}, {-2, 0}, {-1, 2}, {1, 2}, {2, 0}, {1, -2}, {-1, -2}, {-2, 0}};
    static const int8_t mod6m1[8] = {5, 0, 1, 2, 3, 4, 5, 0};
    static const int8_t square1[8][2] = {{0, -1}, {0, 1}, {-1, 0}, {1, 0}, {-1, -1}, {1, 1}, {-1, 1}, {1, -1}};
    const int bw = r_pix_w4[pix], bh = r_pix_h4[pix], k0 = s_scan8(i4), k1 = k0 + offs[pix];
    i16 (*cmv)[2] = CMV(m, list);
    const int ref = CREF(m, list)[k0];
    uint64_t bcost = pix == X264HIP_PIXEL_16x16 ? (uint64_t)(unsigned)*pcost : R_COST_MAX64;
    int bmx = *pmvx, bmy = *pmvy, omx, omy, pmx, pmy, dir = -2, satds[8];
    const int m_mvx = bmx, m_mvy = bmy;
    unsigned bsatd;
    me_ctx c;
    if (pix != X264HIP_PIXEL_16x16 && i4 != 0) predict_mv_blk_l(m, list, i4, bw, mvp);
    pmx = mvp[0]; pmy = mvp[1];
    set_me_ctx_blk_l(S, m, list, ref, mvp, &c, pix, blk_x[i4], blk_y[i4]);
#define SATD_AT(mx_, my_, dst_, avoid_) do { \
        if (!(avoid_) || !((mx_) == pmx && (my_) == pmy)) { dst_ = me_qpel_cmp(&c, mx_, my_, 1); if ((unsigned)(dst_) < bsatd) bsatd = (unsigned)(dst_); } \
        else dst_ = S_COST_MAX; } while (0)
#define RD_AT(mx_, my_, satd_, do_dir_, mdir_) do { \
        if ((unsigned)(satd_) <= bsatd * 17 / 16) { \
            cmv[k0][0] = cmv[k1][0] = (i16)(mx_); cmv[k0][1] = cmv[k1][1] = (i16)(my_); \
            if (pix == X264HIP_PIXEL_16x16 && S->slice_type != S_SLICE_B) fill_part(m, 0, 0, 4, 4, ref, mx_, my_);      /* (the twin's P encoder reads mv4 / ref8) */ \
            const uint64_t cost_ = rd_cost_part(S, m, S->lambda2, i4, pix); \
            if (cost_ < bcost) { bcost = cost_; bmx = (mx_); bmy = (my_); if (do_dir_) dir = (mdir_); } } } while (0)
    {   /* COST_MV_SATD( bmx, bmy, bsatd, 0 ): the first value of bsatd */
        bsatd = (unsigned)me_qpel_cmp(&c, bmx, bmy, 1);
    }
    RD_AT(bmx, bmy, 0, 0, 0);
    /* check the predicted mv */
    if ((bmx != pmx || bmy != pmy) && pmx >= c.smin[0] && pmx <= c.smax[0] && pmy >= c.smin[1] && pmy <= c.smax[1]) {
        int satd;
        SATD_AT(pmx, pmy, satd, 0);
        RD_AT(pmx, pmy, satd, 0, 0);
        /* "if pmv is chosen, set the MV to avoid checking to bmv instead" */
        if (bmx == pmx && bmy == pmy) { pmx = m_mvx; pmy = m_mvy; }
    }
    /* subpel hex search */
    dir = -2; omx = bmx; omy = bmy;
    for (int j = 0; j < 6; j++) SATD_AT(omx + hex2[j + 1][0], omy + hex2[j + 1][1], satds[j], 1);
    for (int j = 0; j < 6; j++) RD_AT(omx + hex2[j + 1][0], omy + hex2[j + 1][1], satds[j], 1, j);
    if (dir != -2)
        for (int i = 1; i < 10; i++) {
            const int odir = mod6m1[dir + 1];
            if (bmy > c.smax[1] - 2 || bmy < c.smin[1] - 2) break;
            dir = -2; omx = bmx; omy = bmy;
            for (int j = 0; j < 3; j++) SATD_AT(omx + hex2[odir + j][0], omy + hex2[odir + j][1], satds[j], 1);
            for (int j = 0; j < 3; j++) RD_AT(omx + hex2[odir + j][0], omy + hex2[odir + j][1], satds[j], 1, odir - 1 + j);
            if (dir == -2) break;
        }
    /* square refine */
    omx = bmx; omy = bmy;
    for (int i = 0; i < 8; i++) SATD_AT(omx + square1[i][0], omy + square1[i][1], satds[i], 1);
    for (int i = 0; i < 8; i++) RD_AT(omx + square1[i][0], omy + square1[i][1], satds[i], 0, 0);
#undef SATD_AT
#undef RD_AT
    bmy = clip3i(bmy, c.smin[1], c.smax[1]);
    *pcost = (int)bcost; *pmvx = bmx; *pmvy = bmy;
    cache_mv_l(m, list, blk_x[i4] >> 2, blk_y[i4] >> 2, bw, bh, bmx, bmy);
    for (int j = 0; j < bh; j++)
        for (int i = 0; i < bw; i++) {
            const int k = 4 + 1 * 8 + (blk_x[i4] >> 2) + i + 8 * ((blk_y[i4] >> 2) + j);
            CMVD(m, list)[k][0] = (i16)(bmx - mvp[0]); CMVD(m, list)[k][1] = (i16)(bmy - mvp[1]);
        }
}

/* ------------------------------------------------------------------ x264_me_refine_bidir (me.c:780-928), both forms: rd = 0 is
 * x264_me_refine_bidir_satd, rd = 1 x264_me_refine_bidir_rd.  NOTE the reference hands i8 to x264_rd_cost_part where that function
 * expects a 4x4 index (me.c:819 `x264_rd_cost_part( h, i_lambda2, i8, m0->i_pixel )`, rdo.c:205 `int i8 = i4 >> 2`): whatever the
 * partition, the trial encodes and prices the macroblock's FIRST 8x8 block (pair).  Kept, because the bytes depend on it. */
static void refine_bidir_any(ssl *S, smb *m, const struct banalysis *B, pme *m0, pme *m1, int pix, int bx, int by, int i8, int rd)
{
    static const int8_t dirs[32][4] = {
        {0, 0, 0, 1}, {0, 0, 0, -1}, {0, 0, 1, 0}, {0, 0, -1, 0}, {0, 1, 0, 0}, {0, -1, 0, 0}, {1, 0, 0, 0}, {-1, 0, 0, 0},
        {0, 0, 1, 1}, {0, 0, -1, -1}, {0, 1, 1, 0}, {0, -1, -1, 0}, {1, 1, 0, 0}, {-1, -1, 0, 0}, {1, 0, 0, 1}, {-1, 0, 0, -1},
        {0, 1, 0, 1}, {0, -1, 0, -1}, {1, 0, 1, 0}, {-1, 0, -1, 0},
        {0, 0, -1, 1}, {0, 0, 1, -1}, {0, -1, 1, 0}, {0, 1, -1, 0}, {-1, 1, 0, 0}, {1, -1, 0, 0}, {1, 0, 0, -1}, {-1, 0, 0, 1},
        {0, -1, 0, 1}, {0, 1, 0, -1}, {-1, 0, 1, 0}, {1, 0, -1, 0}};
    static const u8 bws[4] = {16, 16, 8, 8}, bhs[4] = {16, 8, 16, 8}, offs[4] = {0, 2, 16, 0};
    const int bw = bws[pix], bh = bhs[pix], weight = S->bipred_weight[B->l[0].i_ref][B->l[1].i_ref];
    const int k0 = s_scan8(4 * i8), k1 = k0 + offs[pix];
    me_ctx c0, c1;
    set_me_ctx_blk_l(S, m, 0, B->l[0].i_ref, m0->mvp, &c0, pix, bx, by);
    set_me_ctx_blk_l(S, m, 1, B->l[1].i_ref, m1->mvp, &c1, pix, bx, by);
    /* the cost tables are centred on the predictors clipped to the horizontal range in both components, as the reference does */
    const i16 *cm0x = S->cost_mv - clip3i(m0->mvp[0], c0.smin[0], c0.smax[0]), *cm0y = S->cost_mv - clip3i(m0->mvp[1], c0.smin[0], c0.smax[0]);
    const i16 *cm1x = S->cost_mv - clip3i(m1->mvp[0], c0.smin[0], c0.smax[0]), *cm1y = S->cost_mv - clip3i(m1->mvp[1], c0.smin[0], c0.smax[0]);
    u8 pix0[9][16 * 16], pix1[9][16 * 16], pixb[16 * 16], *src0[9], *src1[9], visited[8][8][8];
    int stride0[9], stride1[9];
    int bm0x = m0->mvx, bm0y = m0->mvy, bm1x = m1->mvx, bm1y = m1->mvy, om0x = bm0x, om0y = bm0y, om1x = bm1x, om1y = bm1y, bcost = S_COST_MAX;
    uint64_t bcostrd = R_COST_MAX64;
    if (bm0y > c0.smax[1] - 8 || bm1y > c0.smax[1] - 8) return;
    memset(visited, 0, sizeof(visited));
#define BIME_CACHE(dx, dy) do { const int i_ = 4 + 3 * (dx) + (dy); stride0[i_] = bw; stride1[i_] = bw; \
        src0[i_] = mcf.get_ref(pix0[i_], &stride0[i_], (u8 **)c0.fref, S->sy, om0x + (dx), om0y + (dy), bw, bh); \
        src1[i_] = mcf.get_ref(pix1[i_], &stride1[i_], (u8 **)c1.fref, S->sy, om1x + (dx), om1y + (dy), bw, bh); } while (0)
#define CHECK_BIDIR(a, b, c, d) do { const int x0 = om0x + (a), y0 = om0y + (b), x1 = om1x + (c), y1 = om1y + (d); \
        if (pass == 0 || !(visited[x0 & 7][y0 & 7][x1 & 7] & (1 << (y1 & 7)))) { \
            const int i0 = 4 + 3 * (a) + (b), i1 = 4 + 3 * (c) + (d); \
            visited[x0 & 7][y0 & 7][x1 & 7] |= (u8)(1 << (y1 & 7)); \
            mcf.avg[pix](pixb, bw, src0[i0], stride0[i0], src1[i1], stride1[i1], weight); \
            const int cost = b_mbcmp(S, pix, m->fe[0] + bx + by * FENC, FENC, pixb, bw) + cm0x[x0] + cm0y[y0] + cm1x[x1] + cm1y[y1]; \
            if (rd) { \
                if ((int64_t)cost < (int64_t)bcost * 17 / 16) { \
                    if (cost < bcost) bcost = cost; \
                    m->cmv[k0][0] = m->cmv[k1][0] = (i16)x0; m->cmv[k0][1] = m->cmv[k1][1] = (i16)y0; \
                    m->cmv1[k0][0] = m->cmv1[k1][0] = (i16)x1; m->cmv1[k0][1] = m->cmv1[k1][1] = (i16)y1; \
                    const uint64_t costrd = rd_cost_part(S, m, S->lambda2, i8, pix); \
                    if (costrd < bcostrd) { bcostrd = costrd; bm0x = x0; bm0y = y0; bm1x = x1; bm1y = y1; } } \
            } else if (cost < bcost) { bcost = cost; bm0x = x0; bm0y = y0; bm1x = x1; bm1y = y1; } } } while (0)
    int pass = 0;
    BIME_CACHE(0, 0);
    CHECK_BIDIR(0, 0, 0, 0);
    for (pass = 0; pass < 8; pass++) {
        BIME_CACHE(1, 0); BIME_CACHE(-1, 0); BIME_CACHE(0, 1); BIME_CACHE(0, -1);
        BIME_CACHE(1, 1); BIME_CACHE(-1, -1); BIME_CACHE(1, -1); BIME_CACHE(-1, 1);
        for (int k = 0; k < 32; k++) CHECK_BIDIR(dirs[k][0], dirs[k][1], dirs[k][2], dirs[k][3]);
        if (om0x == bm0x && om0y == bm0y && om1x == bm1x && om1y == bm1y) break;
        om0x = bm0x; om0y = bm0y; om1x = bm1x; om1y = bm1y;
        BIME_CACHE(0, 0);
    }
#undef BIME_CACHE
#undef CHECK_BIDIR
    m0->mvx = bm0x; m0->mvy = bm0y; m1->mvx = bm1x; m1->mvy = bm1y;
}

/* ------------------------------------------------------------------ x264_intra_rd_refine, analyse.c:876-1056 */
static void intra_rd_refine(ssl *S, smb *m)
{
    int mode[9], n;
    uint64_t best;
    m->skip_intra = 0;
    if (m->type == S_I_16x16) {
        const int old = m->pred16, thresh = m->satd_i16_dir[old] * 9 / 8;
        best = (uint64_t)(unsigned)m->satd_i16;
        n = modes_16x16(m->nb, mode);
        for (int i = 0; i < n; i++) {
            if (mode[i] == old || m->satd_i16_dir[mode[i]] > thresh) continue;
            m->i16mode = mode[i];
            const uint64_t c = (uint64_t)(unsigned)rd_cost_mb(S, m, S->lambda2);
            if (c < best) { best = c; m->pred16 = mode[i]; }
        }
    }
    /* RD selection for chroma prediction */
    n = modes_chroma(m->nb, mode);
    if (n > 1) {
        const int thresh = m->satd_chroma * 5 / 4;
        int j = 0;
        for (int i = 0; i < n; i++)
            if (m->satd_c_dir[i] < thresh && mode[i] != m->predc) mode[j++] = mode[i];
        n = j;
        if (n > 0) {
            int cbp_best = m->cbp_chroma;
            const int lam = s_lambda2_tab[S->qpc];
            /* "the previous thing encoded was x264_intra_rd(), so the pixels and coefs for the current chroma mode are still around" */
            best = rd_cost_i8x8_chroma(S, m, lam, m->predc, 0);
            for (int i = 0; i < n; i++) {
                pred_chroma(S, m, mode[i]);
                /* "if we've already found a mode that needs no residual, then probably any mode with a residual will be worse" */
                const uint64_t c = rd_cost_i8x8_chroma(S, m, lam, mode[i], m->cbp_chroma != 0);
                if (c < best) { best = c; m->predc = mode[i]; cbp_best = m->cbp_chroma; }
            }
            m->chroma_mode = m->predc;
            m->cbp_chroma = cbp_best;
        }
    }
    if (m->type == S_I_4x4) {
        for (int idx = 0; idx < 16; idx++) {
            u8 *dst = m->fd[0] + blk_x[idx] + blk_y[idx] * FDEC, pels[16];
            int nnz = 0;
            best = R_COST_MAX64;
            n = modes_4x4(m->nb4[idx], mode);
            if ((m->nb4[idx] & (NB_TOPRIGHT | NB_TOP)) == NB_TOP) memset(dst + 4 - FDEC, dst[3 - FDEC], 4);
            memset(pels, 0, sizeof(pels));
            for (int i = 0; i < n; i++) {
                pred_4x4(S, m, idx, mode[i]);
                const uint64_t c = rd_cost_i4x4(S, m, S->lambda2, idx, mode[i]);
                if (best > c) {
                    m->pred4[idx] = mode[i]; best = c;
                    for (int r = 0; r < 4; r++) memcpy(pels + 4 * r, dst + r * FDEC, 4);
                    nnz = m->nnz[idx];
                }
            }
            for (int r = 0; r < 4; r++) memcpy(dst + r * FDEC, pels + 4 * r, 4);
            m->nnz[idx] = (u8)nnz;
            m->i4c[s_scan8(idx)] = (int8_t)m->pred4[idx];
        }
    } else if (m->type == S_I_8x8) {
        u8 edge[40];
        for (int idx = 0; idx < 4; idx++) {
            const int x = idx & 1, y = idx >> 1, thresh = m->satd_i8_dir[m->pred8[idx]][idx] * 11 / 8;
            u8 *dst = m->fd[0] + 8 * x + 8 * y * FDEC, pels_h[8], pels_v[7], nz[4] = {0, 0, 0, 0};
            int cbp_new = 0;
            best = R_COST_MAX64;
            memset(pels_h, 0, 8); memset(pels_v, 0, 7);
            n = modes_4x4(m->nb8[idx], mode);
            s_p8filter(dst, edge, m->nb8[idx], 0xf);
            for (int i = 0; i < n; i++) {
                if (m->satd_i8_dir[mode[i]][idx] > thresh) continue;
                pred_8x8(S, m, idx, mode[i], edge);
                m->cbp_luma = m->cbp_i8_rd;
                const uint64_t c = rd_cost_i8x8(S, m, S->lambda2, idx, mode[i]);
                if (best > c) {
                    m->pred8[idx] = mode[i]; cbp_new = m->cbp_luma; best = c;
                    memcpy(pels_h, dst + 7 * FDEC, 8);
                    if (!(idx & 1)) for (int j = 0; j < 7; j++) pels_v[j] = dst[7 + j * FDEC];
                    memcpy(nz, m->nnz + 4 * idx, 4);
                }
            }
            m->cbp_i8_rd = cbp_new;
            memcpy(dst + 7 * FDEC, pels_h, 8);
            if (!(idx & 1)) for (int j = 0; j < 7; j++) dst[7 + j * FDEC] = pels_v[j];
            memcpy(m->nnz + 4 * idx, nz, 4);
            for (int k = 0; k < 4; k++) m->i4c[s_scan8(4 * idx) + (k & 1) + 8 * (k >> 1)] = (int8_t)m->pred8[idx];
        }
    }
}

/* ------------------------------------------------------------------ the P call site, analyse.c:2406-2464 */
static void refine_p_rd(ssl *S, smb *m, panalysis *A)
{
    if (S_IS_INTRA(m->type)) { intra_rd_refine(S, m); return; }
    if (m->partition == S_D_16x16) {
        cache_set(m, 0, 0, 4, 4, A->me16.ref, 0, 0, 0);
        refine_qpel_rd(S, m, 0, X264HIP_PIXEL_16x16, 0, &A->me16.mvx, &A->me16.mvy, &A->me16.cost, A->me16.mvp);
    } else if (m->partition == S_D_16x8) {
        for (int i = 0; i < 4; i++) m->sub[i] = S_D_L0_8x8;
        cache_set(m, 0, 0, 4, 2, A->me16x8[0].ref, 0, 0, 0); cache_set(m, 0, 2, 4, 2, A->me16x8[1].ref, 0, 0, 0);
        for (int i = 0; i < 2; i++) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_16x8, 8 * i, &A->me16x8[i].mvx, &A->me16x8[i].mvy, &A->me16x8[i].cost, A->me16x8[i].mvp);
    } else if (m->partition == S_D_8x16) {
        for (int i = 0; i < 4; i++) m->sub[i] = S_D_L0_8x8;
        cache_set(m, 0, 0, 2, 4, A->me8x16[0].ref, 0, 0, 0); cache_set(m, 2, 0, 2, 4, A->me8x16[1].ref, 0, 0, 0);
        for (int i = 0; i < 2; i++) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_8x16, 4 * i, &A->me8x16[i].mvx, &A->me8x16[i].mvy, &A->me8x16[i].cost, A->me8x16[i].mvp);
    } else if (m->partition == S_D_8x8) {
        update_cache(S, m, A);
        for (int i = 0; i < 4; i++) {
            if (m->sub[i] == S_D_L0_8x8) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_8x8, 4 * i, &A->me8[i].mvx, &A->me8[i].mvy, &A->me8[i].cost, A->me8[i].mvp);
            else if (m->sub[i] == S_D_L0_8x4)
                for (int k = 0; k < 2; k++) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_8x4, 4 * i + 2 * k, &A->me84[i][k].mvx, &A->me84[i][k].mvy, &A->me84[i][k].cost, A->me84[i][k].mvp);
            else if (m->sub[i] == S_D_L0_4x8)
                for (int k = 0; k < 2; k++) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_4x8, 4 * i + k, &A->me48[i][k].mvx, &A->me48[i][k].mvy, &A->me48[i][k].cost, A->me48[i][k].mvp);
            else
                for (int k = 0; k < 4; k++) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_4x4, 4 * i + k, &A->me4[i][k].mvx, &A->me4[i][k].mvy, &A->me4[i][k].cost, A->me4[i][k].mvp);
        }
    }
}
/* ... and the B one, :2707-2758 (after x264_refine_bidir) */
static void refine_b_rd(ssl *S, smb *m, panalysis *A)
{
    struct banalysis *B = A->B;
    if (!(m->type > S_B_DIRECT && m->type < S_B_SKIP)) return;
    update_cache(S, m, A);
    if (m->partition == S_D_16x16) {
        if (m->type == S_B_L0_L0) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_16x16, 0, &B->l[0].me16.mvx, &B->l[0].me16.mvy, &B->l[0].me16.cost, B->l[0].me16.mvp);
        else if (m->type == S_B_L1_L1) refine_qpel_rd(S, m, 1, X264HIP_PIXEL_16x16, 0, &B->l[1].me16.mvx, &B->l[1].me16.mvy, &B->l[1].me16.cost, B->l[1].me16.mvp);
        else if (m->type == S_B_BI_BI) refine_bidir_any(S, m, B, &B->l[0].me16, &B->l[1].me16, X264HIP_PIXEL_16x16, 0, 0, 0, 1);
    } else if (m->partition == S_D_16x8) {
        for (int i = 0; i < 2; i++) {
            m->sub[2 * i] = m->sub[2 * i + 1] = (int8_t)B->part16x8[i];
            if (B->part16x8[i] == S_D_L0_8x8) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_16x8, 8 * i, &B->l[0].me16x8[i].mvx, &B->l[0].me16x8[i].mvy, &B->l[0].me16x8[i].cost, B->l[0].me16x8[i].mvp);
            else if (B->part16x8[i] == S_D_L1_8x8) refine_qpel_rd(S, m, 1, X264HIP_PIXEL_16x8, 8 * i, &B->l[1].me16x8[i].mvx, &B->l[1].me16x8[i].mvy, &B->l[1].me16x8[i].cost, B->l[1].me16x8[i].mvp);
            else if (B->part16x8[i] == S_D_BI_8x8) refine_bidir_any(S, m, B, &B->l[0].me16x8[i], &B->l[1].me16x8[i], X264HIP_PIXEL_16x8, 0, 8 * i, 2 * i, 1);
        }
    } else if (m->partition == S_D_8x16) {
        for (int i = 0; i < 2; i++) {
            m->sub[i] = m->sub[i + 2] = (int8_t)B->part8x16[i];
            if (B->part8x16[i] == S_D_L0_8x8) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_8x16, 4 * i, &B->l[0].me8x16[i].mvx, &B->l[0].me8x16[i].mvy, &B->l[0].me8x16[i].cost, B->l[0].me8x16[i].mvp);
            else if (B->part8x16[i] == S_D_L1_8x8) refine_qpel_rd(S, m, 1, X264HIP_PIXEL_8x16, 4 * i, &B->l[1].me8x16[i].mvx, &B->l[1].me8x16[i].mvy, &B->l[1].me8x16[i].cost, B->l[1].me8x16[i].mvp);
            else if (B->part8x16[i] == S_D_BI_8x8) refine_bidir_any(S, m, B, &B->l[0].me8x16[i], &B->l[1].me8x16[i], X264HIP_PIXEL_8x16, 8 * i, 0, i, 1);
        }
    } else if (m->partition == S_D_8x8) {
        for (int i = 0; i < 4; i++) {
            if (m->sub[i] == S_D_L0_8x8) refine_qpel_rd(S, m, 0, X264HIP_PIXEL_8x8, 4 * i, &B->l[0].me8[i].mvx, &B->l[0].me8[i].mvy, &B->l[0].me8[i].cost, B->l[0].me8[i].mvp);
            else if (m->sub[i] == S_D_L1_8x8) refine_qpel_rd(S, m, 1, X264HIP_PIXEL_8x8, 4 * i, &B->l[1].me8[i].mvx, &B->l[1].me8[i].mvy, &B->l[1].me8[i].cost, B->l[1].me8[i].mvp);
            else if (m->sub[i] == S_D_BI_8x8) refine_bidir_any(S, m, B, &B->l[0].me8[i], &B->l[1].me8[i], X264HIP_PIXEL_8x8, 8 * (i & 1), 8 * (i >> 1), i, 1);
        }
    }
}
