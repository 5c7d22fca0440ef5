/* b_oracle.c -- TEST INFRASTRUCTURE; textually included by slice_oracle.c before analyse_mb.
 *
 * CPU restatement of the B-slice half of the reference's per-macroblock loop:
 *   x264_macroblock_bipred_init, map_col_to_list0            R/common/macroblock.c:1374-1408, 787-805
 *   x264_mb_predict_mv_direct16x16 (spatial / temporal)      R/common/macroblock.c:155-343
 *   x264_mb_load_mv_direct8x8                                 :345-358
 *   x264_mb_predict_mv_ref16x16 (both lists)                  :361-437
 *   x264_mb_mc for the B types (x264_mb_mc_0/1/01xywh ...)    :462-648
 *   x264_macroblock_probe_bskip                               R/encoder/macroblock.c:797-883
 *   x264_mb_analyse_inter_direct / _b16x16 / _b8x8 / _b16x8 / _b8x16, x264_mb_analyse_b_rd,
 *   x264_refine_bidir, the B branch of x264_macroblock_analyse and of x264_analyse_update_cache
 *                                                             R/encoder/analyse.c:1521-1933, 2005-2107, 2467-2733, 2848-2914
 *   x264_me_refine_bidir_satd                                 R/encoder/me.c:790-928
 * x264 core 66 uses no B partition smaller than 8x8; the B frames of a chain are disposable (no b-pyramid): one list-1 picture. */

typedef struct { pme me16, me8[4], me16x8[2], me8x16[2]; int i_ref, rd16; } blist;
static void intra_rd_refine(ssl *S, smb *m);                       /* refine_oracle.c */
static void refine_b_rd(ssl *S, smb *m, panalysis *A);
struct banalysis {
    blist l[2];
    int direct_available;
    int cost16bi, cost16direct, cost8direct[4], cost8bi, cost16x8bi, cost8x16bi;
    int rd16bi, rd16direct, rd8bi, rd16x8bi, rd8x16bi;
    int part16x8[2], part8x16[2], type16x8, type8x16;    /* a->i_mb_partition16x8 / 8x16 (D_L0_8x8 / D_L1_8x8 / D_BI_8x8), a->i_mb_type16x8 / 8x16 */
};
static const u8 s_mb_b_cost[19] = {9, 9, 9, 9, 0, 0, 0, 1, 3, 7, 7, 7, 3, 7, 7, 7, 5, 9, 0};      /* i_mb_b_cost_table, analyse.c:162-164 */
static const u8 s_mb_b16x8_cost[17] = {0, 0, 0, 0, 0, 0, 0, 0, 5, 7, 7, 7, 5, 7, 9, 9, 9};        /* i_mb_b16x8_cost_table */
static const u8 s_sub_mb_b_cost[13] = {7, 5, 5, 3, 7, 5, 7, 3, 7, 7, 7, 5, 1};                    /* i_sub_mb_b_cost_table */
#define SCAN8_0 (4 + 1 * 8)

/* x264_macroblock_slice_init's B part and x264_macroblock_bipred_init */
static void b_slice_init(ssl *S, const slice_ext *e)
{
    int8_t *map = S->map_col_store + 2;
    map[-1] = -1; map[-2] = -2;
    for (int i = 0; i < S->fref1[0]->n_ref0; i++) {
        const int poc = S->fref1[0]->ref_poc[i];
        map[i] = -2;
        for (int j = 0; j < S->n_ref; j++) if (S->fref[j]->poc == poc) { map[i] = (int8_t)j; break; }
    }
    for (int i0 = 0; i0 < S->n_ref; i0++) {
        const int poc0 = S->fref[i0]->poc;
        for (int i1 = 0; i1 < S->n_ref1; i1++) {
            const int td = clip3i(S->fref1[i1]->poc - poc0, -128, 127);
            int dsf;
            if (td == 0) dsf = 256;
            else {
                const int tb = clip3i(S->fdec->poc - poc0, -128, 127), tx = (16384 + (abs(td) >> 1)) / td;
                dsf = clip3i((tb * tx + 32) >> 6, -1024, 1023);
            }
            S->dist_scale[i0][i1] = dsf;
            dsf >>= 2;
            S->bipred_weight[i0][i1] = e->weightb && dsf >= -64 && dsf <= 128 ? 64 - dsf : 32;
        }
    }
}
static int b_ref_cost(const ssl *S, int list, int ref)     /* REF_COST, analyse.c:200-216 */
{
    if (!list) return S->ref_cost[ref];
    return S->lambda * s_te_size(clip3i((S->n_ref1 <= 0 ? 1 : S->n_ref1) - 1, 0, 2), ref);
}

static void cache_mv_l(smb *m, int list, int x, int y, int w, int h, int mvx, int mvy)
{
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) { const int k = SCAN8_0 + x + i + 8 * (y + j); CMV(m, list)[k][0] = (i16)mvx; CMV(m, list)[k][1] = (i16)mvy; }
}
static void cache_mvd_l(smb *m, int list, int x, int y, int w, int h)    /* x264_macroblock_cache_mvd(.., 0) */
{
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) { const int k = SCAN8_0 + x + i + 8 * (y + j); CMVD(m, list)[k][0] = CMVD(m, list)[k][1] = 0; }
}
static void cache_skip(smb *m, int x, int y, int w, int h, int v)
{
    for (int j = 0; j < h; j++) for (int i = 0; i < w; i++) m->cskip[SCAN8_0 + x + i + 8 * (y + j)] = (int8_t)v;
}
/* x264_mb_predict_mv_16x16 from the cache, either list (macroblock.c:90-128) */
static void predict_mv_16x16_c(const smb *m, int list, int i_ref, i16 mvp[2])
{
    const int8_t *cref = CREF(m, list);
    const i16 (*cmv)[2] = CMV(m, list);
    int ra = cref[SCAN8_0 - 1], rb = cref[SCAN8_0 - 8], rc = cref[SCAN8_0 - 8 + 4], cnt;
    const i16 *a = cmv[SCAN8_0 - 1], *b = cmv[SCAN8_0 - 8], *c = cmv[SCAN8_0 - 8 + 4];
    if (rc == -2) { rc = cref[SCAN8_0 - 8 - 1]; c = cmv[SCAN8_0 - 8 - 1]; }
    cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
    if (cnt > 1) { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
    else if (cnt == 1) { const i16 *s = ra == i_ref ? a : rb == i_ref ? b : c; mvp[0] = s[0]; mvp[1] = s[1]; }
    else if (rb == -2 && rc == -2 && ra != -2) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = s_median(a[0], b[0], c[0]); mvp[1] = s_median(a[1], b[1], c[1]); }
}

/* ------------------------------------------------------------------ direct prediction */
static int b_direct_temporal(ssl *S, smb *m)
{
    const sframe *col = S->fref1[0];
    const int8_t *map = S->map_col_store + 2;
    cache_set_l(m, 1, 0, 0, 4, 4, 0, 0, 0, 0);
    if (S_IS_INTRA(col->mb_type[m->mb])) {
        cache_set_l(m, 0, 0, 0, 4, 4, 0, 0, 0, 1);
        cache_mv_l(m, 1, 0, 0, 4, 4, 0, 0);
        return 1;
    }
    for (int i8 = 0; i8 < 4; i8++) {
        const int x8 = i8 & 1, y8 = i8 >> 1, i_ref = map[col->ref[m->mb * 4 + i8]];
        if (i_ref < 0) return 0;                         /* the co-located reference is not in list 0 */
        const int dsf = S->dist_scale[i_ref][0];
        const i16 *mv_col = col->mv + (m->mb * 16 + 3 * x8 + 3 * y8 * 4) * 2;
        const int l0x = (dsf * mv_col[0] + 128) >> 8, l0y = (dsf * mv_col[1] + 128) >> 8;
        cache_set_l(m, 0, 2 * x8, 2 * y8, 2, 2, i_ref, (i16)l0x, (i16)l0y, 1);
        cache_mv_l(m, 1, 2 * x8, 2 * y8, 2, 2, (i16)(l0x - mv_col[0]), (i16)(l0y - mv_col[1]));
    }
    return 1;
}
static int b_direct_spatial(ssl *S, smb *m)
{
    const sframe *col = S->fref1[0];
    int ref[2];
    i16 mv[2][2];
    for (int l = 0; l < 2; l++) {
        const int8_t *cref = CREF(m, l);
        int ra = cref[SCAN8_0 - 1], rb = cref[SCAN8_0 - 8], rc = cref[SCAN8_0 - 8 + 4];
        if (rc == -2) rc = cref[SCAN8_0 - 8 - 1];
        ref[l] = ra;
        if (ref[l] < 0 || (rb < ref[l] && rb >= 0)) ref[l] = rb;
        if (ref[l] < 0 || (rc < ref[l] && rc >= 0)) ref[l] = rc;
        if (ref[l] < 0) ref[l] = -1;
    }
    if (ref[0] < 0 && ref[1] < 0) {
        cache_set_l(m, 0, 0, 0, 4, 4, 0, 0, 0, 1); cache_set_l(m, 1, 0, 0, 4, 4, 0, 0, 0, 1);
        return 1;
    }
    for (int l = 0; l < 2; l++) {
        if (ref[l] >= 0) predict_mv_16x16_c(m, l, ref[l], mv[l]);
        else mv[l][0] = mv[l][1] = 0;
    }
    cache_set_l(m, 0, 0, 0, 4, 4, ref[0], mv[0][0], mv[0][1], 1);
    cache_set_l(m, 1, 0, 0, 4, 4, ref[1], mv[1][0], mv[1][1], 1);
    if (S_IS_INTRA(col->mb_type[m->mb]) || (ref[0] && ref[1])) return 1;
    for (int i8 = 0; i8 < 4; i8++) {                       /* col_zero_flag */
        const int x8 = i8 & 1, y8 = i8 >> 1, r0 = col->ref[m->mb * 4 + i8], r1 = col->ref1[m->mb * 4 + i8];
        if (r0 == 0 || (r0 < 0 && r1 == 0)) {
            const i16 *mvcol = (r0 == 0 ? col->mv : col->mv1) + (m->mb * 16 + 3 * x8 + 3 * y8 * 4) * 2;
            if (abs(mvcol[0]) <= 1 && abs(mvcol[1]) <= 1) {
                if (ref[0] == 0) cache_mv_l(m, 0, 2 * x8, 2 * y8, 2, 2, 0, 0);
                if (ref[1] == 0) cache_mv_l(m, 1, 2 * x8, 2 * y8, 2, 2, 0, 0);
            }
        }
    }
    return 1;
}
static int b_predict_direct(ssl *S, smb *m)               /* x264_mb_predict_mv_direct16x16(h, NULL) */
{
    const int ok = S->direct_spatial ? b_direct_spatial(S, m) : b_direct_temporal(S, m);
    if (ok)
        for (int l = 0; l < 2; l++) {
            for (int i = 0; i < 4; i++) m->direct_ref[l][i] = CREF(m, l)[s_scan8(4 * i)];
            for (int i = 0; i < 16; i++) { const int k = SCAN8_0 + (i & 3) + 8 * (i >> 2); m->direct_mv[l][i][0] = CMV(m, l)[k][0]; m->direct_mv[l][i][1] = CMV(m, l)[k][1]; }
        }
    return ok;
}
static void b_load_mv_direct8x8(smb *m, int idx)
{
    const int x = 2 * (idx & 1), y = 2 * (idx >> 1);
    for (int l = 0; l < 2; l++) {
        cache_set_l(m, l, x, y, 2, 2, m->direct_ref[l][idx], 0, 0, 0);
        for (int j = 0; j < 2; j++)
            for (int i = 0; i < 2; i++) {
                const int k = SCAN8_0 + x + i + 8 * (y + j), b = (y + j) * 4 + x + i;
                CMV(m, l)[k][0] = m->direct_mv[l][b][0]; CMV(m, l)[k][1] = m->direct_mv[l][b][1];
            }
    }
}

/* x264_mb_predict_mv_ref16x16 for either list of a B slice */
static int b_predict_mv_ref16x16(const ssl *S, const smb *m, int list, int i_ref, i16 mvc[9][2])
{
    const i16 *mvr = list ? S->mvr1 : S->mvr + (size_t)i_ref * S->n * 2;
    const int8_t *type = S->fdec->mb_type;
    int i = 0, top = m->mb - S->mb_w;
    if (CREF(m, list)[s_scan8(12)] == i_ref) { mvc[i][0] = CMV(m, list)[s_scan8(12)][0]; mvc[i][1] = CMV(m, list)[s_scan8(12)][1]; i++; }   /* b_direct */
    if (i_ref == 0 && S->lowres_mv[list]) {                /* the lookahead's vector, :393-398 */
        mvc[i][0] = (i16)(u16)(S->lowres_mv[list][2 * m->mb] << 1); mvc[i][1] = (i16)(u16)(S->lowres_mv[list][2 * m->mb + 1] << 1); i++;
    }
#define SET(o) do { mvc[i][0] = mvr[2 * (o)]; mvc[i][1] = mvr[2 * (o) + 1]; i++; } while (0)
    if ((m->nb & NB_LEFT) && !S_IS_SKIP(type[m->mb - 1])) SET(m->mb - 1);
    if (m->nb & NB_TOP) {
        if (!S_IS_SKIP(type[top])) SET(top);
        if ((m->nb & NB_TOPLEFT) && !S_IS_SKIP(type[top - 1])) SET(top - 1);
        if (m->mbx < S->mb_w - 1 && !S_IS_SKIP(type[top + 1])) SET(top + 1);
    }
#undef SET
    if (S->fref[0]->n_ref0 > 0) {                          /* temporal predictors: always from list 0's first picture, scaled with list 0's POC distance */
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

/* ------------------------------------------------------------------ motion compensation of the B types */
static u8 *b_get_ref(const ssl *S, const smb *m, int list, int ref, int bx, int by, int mvx, int mvy, int w, int h, u8 *buf, int *stride)
{
    const sframe *r = list ? S->fref1[ref] : S->fref[ref];
    const int o = (16 * m->mby + by) * S->sy + 16 * m->mbx + bx;
    u8 *src4[4] = {r->filt[0] + o, r->filt[1] + o, r->filt[2] + o, r->filt[3] + o};
    return mcf.get_ref(buf, stride, src4, S->sy, mvx, mvy, w, h);
}
static int b_size2pixel(int w4, int h4) { return w4 == 4 ? (h4 == 4 ? 0 : 1) : (h4 == 4 ? 2 : 3); }   /* x264_size2pixel for 16x16 / 16x8 / 8x16 / 8x8 */
/* x264_mb_mc_0xywh / _1xywh / _01xywh on a w x h run of 4x4 blocks at (x, y); lists: 1 = list 0, 2 = list 1, 3 = both */
static void mc_b_part(const ssl *S, smb *m, int x, int y, int w, int h, int lists)
{
    const int k = SCAN8_0 + x + 8 * y, oc = (8 * m->mby + 2 * y) * S->sc + 8 * m->mbx + 2 * x;
    u8 *dst[3] = {m->fd[0] + 4 * y * FDEC + 4 * x, m->fd[1] + 2 * y * FDEC + 2 * x, m->fd[2] + 2 * y * FDEC + 2 * x};
    if (lists != 3) {
        const int l = lists == 2, ref = CREF(m, l)[k];
        const sframe *r = l ? S->fref1[ref] : S->fref[ref];
        int mvx = CMV(m, l)[k][0], mvy = CMV(m, l)[k][1];
        const int o = (16 * m->mby + 4 * y) * S->sy + 16 * m->mbx + 4 * x;
        u8 *src4[4] = {r->filt[0] + o, r->filt[1] + o, r->filt[2] + o, r->filt[3] + o};
        mv_clip_frame(S, m, &mvx, &mvy);
        mcf.mc_luma(dst[0], FDEC, src4, S->sy, mvx, mvy, 4 * w, 4 * h);
        mcf.mc_chroma(dst[1], FDEC, r->plane[1] + oc, S->sc, mvx, mvy, 2 * w, 2 * h);
        mcf.mc_chroma(dst[2], FDEC, r->plane[2] + oc, S->sc, mvx, mvy, 2 * w, 2 * h);
        return;
    }
    const int ref0 = m->cref[k], ref1 = m->cref1[k], weight = S->bipred_weight[ref0][ref1], mode = b_size2pixel(w, h);
    int mvx0 = m->cmv[k][0], mvy0 = m->cmv[k][1], mvx1 = m->cmv1[k][0], mvy1 = m->cmv1[k][1], s0 = 16, s1 = 16;
    u8 tmp0[16 * 16], tmp1[16 * 16];
    mv_clip_frame(S, m, &mvx0, &mvy0); mv_clip_frame(S, m, &mvx1, &mvy1);
    u8 *src0 = b_get_ref(S, m, 0, ref0, 4 * x, 4 * y, mvx0, mvy0, 4 * w, 4 * h, tmp0, &s0);
    u8 *src1 = b_get_ref(S, m, 1, ref1, 4 * x, 4 * y, mvx1, mvy1, 4 * w, 4 * h, tmp1, &s1);
    mcf.avg[mode](dst[0], FDEC, src0, s0, src1, s1, weight);
    for (int pl = 1; pl < 3; pl++) {
        mcf.mc_chroma(tmp0, 16, S->fref[ref0]->plane[pl] + oc, S->sc, mvx0, mvy0, 2 * w, 2 * h);
        mcf.mc_chroma(tmp1, 16, S->fref1[ref1]->plane[pl] + oc, S->sc, mvx1, mvy1, 2 * w, 2 * h);
        mcf.avg[mode + 3](dst[pl], FDEC, tmp0, 16, tmp1, 16, weight);
    }
}
static void mc_b_direct8x8(const ssl *S, smb *m, int x, int y)
{
    const int k = SCAN8_0 + x + 8 * y;
    mc_b_part(S, m, x, y, 2, 2, m->cref[k] >= 0 ? (m->cref1[k] >= 0 ? 3 : 1) : 2);
}
static void mc_b(const ssl *S, smb *m)                    /* x264_mb_mc, the B types */
{
    if (m->type == S_B_8x8) {
        for (int i = 0; i < 4; i++) {
            const int x = 2 * (i & 1), y = 2 * (i >> 1), t = m->sub[i];
            if (t == S_D_DIRECT_8x8) mc_b_direct8x8(S, m, x, y);
            else mc_b_part(S, m, x, y, 2, 2, t == S_D_L0_8x8 ? 1 : t == S_D_L1_8x8 ? 2 : 3);
        }
    } else if (m->type == S_B_SKIP || m->type == S_B_DIRECT) {
        mc_b_direct8x8(S, m, 0, 0); mc_b_direct8x8(S, m, 2, 0); mc_b_direct8x8(S, m, 0, 2); mc_b_direct8x8(S, m, 2, 2);
    } else {
        const int n = m->partition == S_D_16x16 ? 1 : 2;
        for (int i = 0; i < n; i++) {
            const int lists = b_type_uses(m->type, 0, i) | b_type_uses(m->type, 1, i) << 1;
            if (!lists) continue;
            if (m->partition == S_D_16x16) mc_b_part(S, m, 0, 0, 4, 4, lists);
            else if (m->partition == S_D_16x8) mc_b_part(S, m, 0, 2 * i, 4, 2, lists);
            else mc_b_part(S, m, 2 * i, 0, 2, 4, lists);
        }
    }
}

/* x264_macroblock_probe_bskip: the prediction is in fdec already */
static int probe_bskip(ssl *S, smb *m)
{
    int dec = 0;
    i16 d4[4][4][4], d2[2][2], scan[16];
    for (int i8 = 0; i8 < 4; i8++) {
        dctf.sub8x8_dct(d4, m->fe[0] + (i8 & 1) * 8 + (i8 >> 1) * 8 * FENC, m->fd[0] + (i8 & 1) * 8 + (i8 >> 1) * 8 * FDEC);
        for (int i4 = 0; i4 < 4; i4++) {
            if (!quantf.quant_4x4(d4[i4], S->mf4[1], S->b4[1])) continue;
            zigf[0].scan_4x4(scan, d4[i4]);
            dec += quantf.decimate_score16(scan);
            if (dec >= 6) return 0;
        }
    }
    const int thresh = (s_lambda2_tab[S->qpc] + 32) >> 6;
    for (int ch = 0; ch < 2; ch++) {
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

/* ------------------------------------------------------------------ x264_analyse_update_cache, the B types */
static void b_cache_mv_bi(smb *m, int x, int y, int w, int h, const struct banalysis *B, const pme *me0, const pme *me1, int part, int b_mvd)
{   /* CACHE_MV_BI, analyse.c:1672-1698 */
    const pme *me[2] = {me0, me1};
    for (int l = 0; l < 2; l++) {
        if (b_sub_uses(part, l)) cache_set_l(m, l, x, y, w, h, B->l[l].i_ref, me[l]->mvx, me[l]->mvy, 1);
        else {
            cache_set_l(m, l, x, y, w, h, -1, 0, 0, 1);
            if (b_mvd) cache_mvd_l(m, l, x, y, w, h);
        }
    }
}
static void b_cache_mv_b8x8(smb *m, const struct banalysis *B, int i, int b_mvd)
{   /* x264_mb_cache_mv_b8x8, :1700-1718 */
    const int x = 2 * (i & 1), y = 2 * (i >> 1);
    if (m->sub[i] == S_D_DIRECT_8x8) {
        b_load_mv_direct8x8(m, i);
        if (b_mvd) { cache_mvd_l(m, 0, x, y, 2, 2); cache_mvd_l(m, 1, x, y, 2, 2); cache_skip(m, x, y, 2, 2, 1); }
    } else
        b_cache_mv_bi(m, x, y, 2, 2, B, &B->l[0].me8[i], &B->l[1].me8[i], m->sub[i], b_mvd);
}
static void update_cache_b(ssl *S, smb *m, struct banalysis *B)
{
    (void)S;
    switch (m->type) {
    case S_B_SKIP: case S_B_DIRECT:
        for (int i = 0; i < 4; i++) b_load_mv_direct8x8(m, i);
        break;
    case S_B_8x8:
        for (int i = 0; i < 4; i++) b_cache_mv_b8x8(m, B, i, 1);
        break;
    default:
        if (m->partition == S_D_16x16) {
            const int part = m->type == S_B_L0_L0 ? S_D_L0_8x8 : m->type == S_B_L1_L1 ? S_D_L1_8x8 : S_D_BI_8x8;
            b_cache_mv_bi(m, 0, 0, 4, 4, B, &B->l[0].me16, &B->l[1].me16, part, 1);
        } else if (m->partition == S_D_16x8)
            for (int i = 0; i < 2; i++) b_cache_mv_bi(m, 0, 2 * i, 4, 2, B, &B->l[0].me16x8[i], &B->l[1].me16x8[i], B->part16x8[i], 1);
        else
            for (int i = 0; i < 2; i++) b_cache_mv_bi(m, 2 * i, 0, 2, 4, B, &B->l[0].me8x16[i], &B->l[1].me8x16[i], B->part8x16[i], 1);
        break;
    }
}
/* what x264_macroblock_cache_save will store: the 16 blocks of both lists of the cache */
static void b_final_vectors(smb *m)
{
    if (S_IS_INTRA(m->type)) return;
    for (int i = 0; i < 16; i++) {
        const int k = SCAN8_0 + (i & 3) + 8 * (i >> 2);
        m->mv4[i][0] = m->cmv[k][0]; m->mv4[i][1] = m->cmv[k][1]; m->mv4_1[i][0] = m->cmv1[k][0]; m->mv4_1[i][1] = m->cmv1[k][1];
    }
    for (int i = 0; i < 4; i++) { m->ref8[i] = m->cref[s_scan8(4 * i)]; m->ref8_1[i] = m->cref1[s_scan8(4 * i)]; }
}
/* x264_mb_analyse_transform (no RD), analyse.c:2109-2126 */
static void analyse_transform_b(ssl *S, smb *m)
{
    if (!s_t8_allowed(S, m)) return;
    mc_b(S, m);
    const int c8 = pixf.sa8d[X264HIP_PIXEL_16x16](m->fe[0], FENC, m->fd[0], FDEC);
    const int c4 = pixf.satd[X264HIP_PIXEL_16x16](m->fe[0], FENC, m->fd[0], FDEC);
    m->t8 = c8 < c4;
    m->skip_mc = 1;
}

/* ------------------------------------------------------------------ the searches */
static int b_mbcmp(const ssl *S, int pix, const u8 *a, int sa, const u8 *b, int sb)
{
    return (S->p->subme > 1 ? pixf.satd : pixf.sad)[pix]((u8 *)a, sa, (u8 *)b, sb);
}
static void b_search(const ssl *S, const smb *m, int list, int ref, const i16 mvp[2], const i16 (*mvc)[2], int n_mvc, int pix, int bx, int by,
                     int *thresh, pme *out)
{
    me_ctx c;
    int mx, my, cmv = 0;
    set_me_ctx_blk_l(S, m, list, ref, mvp, &c, pix, bx, by);
    out->cost = me_search16(&c, mvp, mvc, n_mvc, S->p->me_method, S->p->me_range, S->p->subme, 0, thresh, &mx, &my, &cmv);
    out->mvx = mx; out->mvy = my; out->cost_mv = cmv; out->ref = ref; out->ref_cost = 0; out->mvp[0] = mvp[0]; out->mvp[1] = mvp[1];
}
static void b_refine_qpel(const ssl *S, const smb *m, int list, int pix, int bx, int by, pme *me)    /* x264_me_refine_qpel: no reference cost to take off in a B slice */
{
    me_ctx c;
    set_me_ctx_blk_l(S, m, list, me->ref, me->mvp, &c, pix, bx, by);
    me->cost = refine_qpel16(S, &c, me->cost, &me->mvx, &me->mvy, me->mvp);
}
/* the weighted average of the two predictions of one block against the source */
static int b_bi_cost(const ssl *S, const smb *m, const struct banalysis *B, const pme *me0, const pme *me1, int pix, int bx, int by, int w, int h)
{
    u8 pix0[16 * 16], pix1[16 * 16];
    int s0 = w, s1 = w;
    u8 *src0 = b_get_ref(S, m, 0, B->l[0].i_ref, bx, by, me0->mvx, me0->mvy, w, h, pix0, &s0);
    u8 *src1 = b_get_ref(S, m, 1, B->l[1].i_ref, bx, by, me1->mvx, me1->mvy, w, h, pix1, &s1);
    mcf.avg[pix](pix0, w, src0, s0, src1, s1, S->bipred_weight[B->l[0].i_ref][B->l[1].i_ref]);
    return b_mbcmp(S, pix, m->fe[0] + bx + by * FENC, FENC, pix0, w);
}

static void analyse_inter_direct(ssl *S, smb *m, struct banalysis *B)
{
    B->cost16direct = S->lambda * s_mb_b_cost[S_B_DIRECT];
    for (int i = 0; i < 4; i++) {
        const int x = (i & 1) * 8, y = (i >> 1) * 8;
        B->cost8direct[i] = b_mbcmp(S, X264HIP_PIXEL_8x8, m->fe[0] + x + y * FENC, FENC, m->fd[0] + x + y * FDEC, FDEC);
        B->cost16direct += B->cost8direct[i];
        B->cost8direct[i] += S->lambda * s_sub_mb_b_cost[S_D_DIRECT_8x8];
    }
}
static void analyse_b16x16(ssl *S, smb *m, struct banalysis *B)
{
    for (int l = 0; l < 2; l++) {
        const int n = l ? S->n_ref1 : S->n_ref;
        int thresh = 0x7fffffff;
        B->l[l].me16.cost = 0x7fffffff;
        for (int r = 0; r < n; r++) {
            i16 mvp[2], mvc[9][2];
            pme t;
            predict_mv_16x16_c(m, l, r, mvp);
            const int n_mvc = b_predict_mv_ref16x16(S, m, l, r, mvc);
            b_search(S, m, l, r, mvp, (const i16 (*)[2])mvc, n_mvc, X264HIP_PIXEL_16x16, 0, 0, n > 1 ? &thresh : 0, &t);
            t.cost += b_ref_cost(S, l, r);
            if (t.cost < B->l[l].me16.cost) { B->l[l].i_ref = r; B->l[l].me16 = t; }
            i16 *mvr = (l ? S->mvr1 : S->mvr + (size_t)r * S->n * 2) + m->mb * 2;
            mvr[0] = (i16)t.mvx; mvr[1] = (i16)t.mvy;
        }
        B->l[l].me16.cost -= b_ref_cost(S, l, B->l[l].i_ref);
    }
    cache_set_l(m, 0, 0, 0, 4, 4, B->l[0].i_ref, 0, 0, 0);
    cache_set_l(m, 1, 0, 0, 4, 4, B->l[1].i_ref, 0, 0, 0);
    B->cost16bi = b_bi_cost(S, m, B, &B->l[0].me16, &B->l[1].me16, X264HIP_PIXEL_16x16, 0, 0, 16, 16)
                + b_ref_cost(S, 0, B->l[0].i_ref) + b_ref_cost(S, 1, B->l[1].i_ref) + B->l[0].me16.cost_mv + B->l[1].me16.cost_mv;
    B->cost16bi += S->lambda * s_mb_b_cost[S_B_BI_BI];
    B->l[0].me16.cost += S->lambda * s_mb_b_cost[S_B_L0_L0];
    B->l[1].me16.cost += S->lambda * s_mb_b_cost[S_B_L1_L1];
}
static void analyse_b8x8(ssl *S, smb *m, struct banalysis *B)
{
    m->partition = S_D_8x8;
    B->cost8bi = 0;
    for (int i = 0; i < 4; i++) {
        const int x8 = i & 1, y8 = i >> 1;
        int part_cost, part_cost_bi = 0;
        for (int l = 0; l < 2; l++) {
            blist *lX = &B->l[l];
            i16 mvp[2], mvc[1][2] = {{(i16)lX->me16.mvx, (i16)lX->me16.mvy}};
            predict_mv_blk_l(m, l, 4 * i, 2, mvp);
            b_search(S, m, l, lX->i_ref, mvp, (const i16 (*)[2])mvc, 1, X264HIP_PIXEL_8x8, 8 * x8, 8 * y8, 0, &lX->me8[i]);
            cache_mv_l(m, l, 2 * x8, 2 * y8, 2, 2, lX->me8[i].mvx, lX->me8[i].mvy);
            part_cost_bi += lX->me8[i].cost_mv;
        }
        part_cost_bi += b_bi_cost(S, m, B, &B->l[0].me8[i], &B->l[1].me8[i], X264HIP_PIXEL_8x8, 8 * x8, 8 * y8, 8, 8) + S->lambda * s_sub_mb_b_cost[S_D_BI_8x8];
        B->l[0].me8[i].cost += S->lambda * s_sub_mb_b_cost[S_D_L0_8x8];
        B->l[1].me8[i].cost += S->lambda * s_sub_mb_b_cost[S_D_L1_8x8];
        part_cost = B->l[0].me8[i].cost; m->sub[i] = S_D_L0_8x8;
        if (B->l[1].me8[i].cost < part_cost) { part_cost = B->l[1].me8[i].cost; m->sub[i] = S_D_L1_8x8; }
        if (part_cost_bi < part_cost) { part_cost = part_cost_bi; m->sub[i] = S_D_BI_8x8; }
        if (B->cost8direct[i] < part_cost) { part_cost = B->cost8direct[i]; m->sub[i] = S_D_DIRECT_8x8; }
        B->cost8bi += part_cost;
        b_cache_mv_b8x8(m, B, i, 0);
    }
    B->cost8bi += S->lambda * s_mb_b_cost[S_B_8x8];
}
/* x264_mb_analyse_inter_b16x8 (dir 0) / _b8x16 (dir 1) */
static void analyse_b16x8(ssl *S, smb *m, struct banalysis *B, int dir)
{
    const int pix = dir ? X264HIP_PIXEL_8x16 : X264HIP_PIXEL_16x8, w = dir ? 8 : 16, h = dir ? 16 : 8;
    int total = 0, *parts = dir ? B->part8x16 : B->part16x8;
    m->partition = dir ? S_D_8x16 : S_D_16x8;
    for (int i = 0; i < 2; i++) {
        const int bx = dir ? 8 * i : 0, by = dir ? 0 : 8 * i;
        int part_cost, part_cost_bi = 0;
        pme *me[2];
        for (int l = 0; l < 2; l++) {
            blist *lX = &B->l[l];
            const pme *a = dir ? &lX->me8[i] : &lX->me8[2 * i], *b = dir ? &lX->me8[i + 2] : &lX->me8[2 * i + 1];
            i16 mvp[2], mvc[2][2] = {{(i16)a->mvx, (i16)a->mvy}, {(i16)b->mvx, (i16)b->mvy}};
            me[l] = dir ? &lX->me8x16[i] : &lX->me16x8[i];
            predict_mv_blk_l(m, l, dir ? 4 * i : 8 * i, 2, mvp);          /* width 2 for both shapes, as the reference has it */
            b_search(S, m, l, lX->i_ref, mvp, (const i16 (*)[2])mvc, 2, pix, bx, by, 0, me[l]);
            part_cost_bi += me[l]->cost_mv;
        }
        part_cost_bi += b_bi_cost(S, m, B, me[0], me[1], pix, bx, by, w, h);
        part_cost = me[0]->cost; parts[i] = S_D_L0_8x8;
        if (me[1]->cost < part_cost) { part_cost = me[1]->cost; parts[i] = S_D_L1_8x8; }
        if (part_cost_bi + S->lambda * 1 < part_cost) { part_cost = part_cost_bi; parts[i] = S_D_BI_8x8; }
        total += part_cost;
        if (dir) b_cache_mv_bi(m, 2 * i, 0, 2, 4, B, me[0], me[1], parts[i], 0);
        else b_cache_mv_bi(m, 0, 2 * i, 4, 2, B, me[0], me[1], parts[i], 0);
    }
    const int type = S_B_L0_L0 + (parts[0] >> 2) * 3 + (parts[1] >> 2);
    total += S->lambda * s_mb_b16x8_cost[type];
    if (dir) { B->type8x16 = type; B->cost8x16bi = total; } else { B->type16x8 = type; B->cost16x8bi = total; }
}

/* x264_mb_analyse_b_rd, analyse.c:2007-2076 */
static void analyse_b_rd(ssl *S, smb *m, panalysis *A, int i_satd_inter)
{
    struct banalysis *B = A->B;
    const int thresh = i_satd_inter * (17 + (!!S->psy_rd)) / 16;
    if (B->direct_available && B->rd16direct == S_COST_MAX) {
        m->type = S_B_DIRECT;
        m->skip_mc = 1;                                   /* "Assumes direct/skip MC is still in fdec" */
        update_cache(S, m, A);
        B->rd16direct = rd_cost_mb(S, m, S->lambda2);
        m->skip_mc = 0;
    }
    m->partition = S_D_16x16;
    if (B->l[0].me16.cost <= thresh && B->l[0].rd16 == S_COST_MAX) { m->type = S_B_L0_L0; update_cache(S, m, A); B->l[0].rd16 = rd_cost_mb(S, m, S->lambda2); }
    if (B->l[1].me16.cost <= thresh && B->l[1].rd16 == S_COST_MAX) { m->type = S_B_L1_L1; update_cache(S, m, A); B->l[1].rd16 = rd_cost_mb(S, m, S->lambda2); }
    if (B->cost16bi <= thresh && B->rd16bi == S_COST_MAX) { m->type = S_B_BI_BI; update_cache(S, m, A); B->rd16bi = rd_cost_mb(S, m, S->lambda2); }
    if (B->cost8bi <= thresh && B->rd8bi == S_COST_MAX) {
        m->type = S_B_8x8; m->partition = S_D_8x8;
        update_cache(S, m, A);
        B->rd8bi = rd_cost_mb(S, m, S->lambda2);
        cache_skip(m, 0, 0, 4, 4, 0);
    }
    if (B->cost16x8bi <= thresh && B->rd16x8bi == S_COST_MAX) { m->type = B->type16x8; m->partition = S_D_16x8; update_cache(S, m, A); B->rd16x8bi = rd_cost_mb(S, m, S->lambda2); }
    if (B->cost8x16bi <= thresh && B->rd8x16bi == S_COST_MAX) { m->type = B->type8x16; m->partition = S_D_8x16; update_cache(S, m, A); B->rd8x16bi = rd_cost_mb(S, m, S->lambda2); }
}

/* x264_me_refine_bidir_satd, me.c:790-928: both vectors of a bi-predicted block together, +-1 quarter sample in at most two components per step */
static void refine_bidir_satd(const ssl *S, const smb *m, const struct banalysis *B, pme *m0, pme *m1, int pix, int bx, int by)
{
    static const int8_t dirs[44][4] = {
        {0, 0, 0, 1}, {0, 0, 0, -1}, {0, 0, 1, 0}, {0, 0, -1, 0}, {0, 1, 0, 0}, {0, -1, 0, 0}, {1, 0, 0, 0}, {-1, 0, 0, 0},
        {0, 0, 1, 1}, {0, 0, -1, -1}, {0, 1, 1, 0}, {0, -1, -1, 0}, {1, 1, 0, 0}, {-1, -1, 0, 0}, {1, 0, 0, 1}, {-1, 0, 0, -1},
        {0, 1, 0, 1}, {0, -1, 0, -1}, {1, 0, 1, 0}, {-1, 0, -1, 0},
        {0, 0, -1, 1}, {0, 0, 1, -1}, {0, -1, 1, 0}, {0, 1, -1, 0}, {-1, 1, 0, 0}, {1, -1, 0, 0}, {1, 0, 0, -1}, {-1, 0, 0, 1},
        {0, -1, 0, 1}, {0, 1, 0, -1}, {-1, 0, 1, 0}, {1, 0, -1, 0}};
    static const u8 bws[4] = {16, 16, 8, 8}, bhs[4] = {16, 8, 16, 8};
    const int bw = bws[pix], bh = bhs[pix], weight = S->bipred_weight[B->l[0].i_ref][B->l[1].i_ref];
    me_ctx c0, c1;
    set_me_ctx_blk_l(S, m, 0, B->l[0].i_ref, m0->mvp, &c0, pix, bx, by);
    set_me_ctx_blk_l(S, m, 1, B->l[1].i_ref, m1->mvp, &c1, pix, bx, by);
    /* the cost tables are centred on the predictors clipped to the horizontal range in both components, as the reference does */
    const i16 *cm0x = S->cost_mv - clip3i(m0->mvp[0], c0.smin[0], c0.smax[0]), *cm0y = S->cost_mv - clip3i(m0->mvp[1], c0.smin[0], c0.smax[0]);
    const i16 *cm1x = S->cost_mv - clip3i(m1->mvp[0], c0.smin[0], c0.smax[0]), *cm1y = S->cost_mv - clip3i(m1->mvp[1], c0.smin[0], c0.smax[0]);
    u8 pix0[9][16 * 16], pix1[9][16 * 16], pixb[16 * 16], *src0[9], *src1[9], visited[8][8][8];
    int stride0[9], stride1[9];
    int bm0x = m0->mvx, bm0y = m0->mvy, bm1x = m1->mvx, bm1y = m1->mvy, om0x = bm0x, om0y = bm0y, om1x = bm1x, om1y = bm1y, bcost = S_COST_MAX;
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
            if (cost < bcost) { bcost = cost; bm0x = x0; bm0y = y0; bm1x = x1; bm1y = y1; } } } while (0)
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
/* x264_refine_bidir, analyse.c:2078-2107 */
static void refine_bidir(ssl *S, smb *m, struct banalysis *B)
{
    if (S_IS_INTRA(m->type)) return;
    switch (m->partition) {
    case S_D_16x16:
        if (m->type == S_B_BI_BI) refine_bidir_satd(S, m, B, &B->l[0].me16, &B->l[1].me16, X264HIP_PIXEL_16x16, 0, 0);
        break;
    case S_D_16x8:
        for (int i = 0; i < 2; i++) if (B->part16x8[i] == S_D_BI_8x8) refine_bidir_satd(S, m, B, &B->l[0].me16x8[i], &B->l[1].me16x8[i], X264HIP_PIXEL_16x8, 0, 8 * i);
        break;
    case S_D_8x16:
        for (int i = 0; i < 2; i++) if (B->part8x16[i] == S_D_BI_8x8) refine_bidir_satd(S, m, B, &B->l[0].me8x16[i], &B->l[1].me8x16[i], X264HIP_PIXEL_8x16, 8 * i, 0);
        break;
    case S_D_8x8:
        for (int i = 0; i < 4; i++) if (m->sub[i] == S_D_BI_8x8) refine_bidir_satd(S, m, B, &B->l[0].me8[i], &B->l[1].me8[i], X264HIP_PIXEL_8x8, 8 * (i & 1), 8 * (i >> 1));
        break;
    default:
        break;
    }
}

/* ------------------------------------------------------------------ the B branch of x264_macroblock_analyse, analyse.c:2467-2733 */
static void analyse_b(ssl *S, smb *m, panalysis *A, int satd_pcm)
{
    struct banalysis *B = A->B;
    const slice_params *p = S->p;
    int i_bskip_cost = S_COST_MAX, b_skip = 0, i_cost, i_type, i_partition, i_satd_inter = 0;
    for (int l = 0; l < 2; l++) { B->l[l].me16.cost = S_COST_MAX; B->l[l].rd16 = S_COST_MAX; }
    for (int i = 0; i < 4; i++) B->cost8direct[i] = S_COST_MAX;
    B->rd16bi = B->rd16direct = B->rd8bi = B->rd16x8bi = B->rd8x16bi = S_COST_MAX;
    B->cost16bi = B->cost16direct = B->cost8bi = B->cost16x8bi = B->cost8x16bi = S_COST_MAX;
    if (S->mbrd) cache_fenc_satd(S, m);
    m->type = S_B_SKIP;
    B->direct_available = b_predict_direct(S, m);
    if (B->direct_available) {
        mc_b(S, m);
        if (S->mbrd) {
            i_bskip_cost = ssd_mb(S, m);
            b_skip = m->skip_mc = i_bskip_cost <= ((6 * S->lambda2 + 128) >> 8);   /* "6 = minimum cavlc cost of a non-skipped MB" */
        } else
            b_skip = probe_bskip(S, m);
    }
    if (b_skip) return;
    m->skip_mc = 0;
    if (B->direct_available) analyse_inter_direct(S, m, B);
    analyse_b16x16(S, m, B);
    i_type = S_B_L0_L0; i_partition = S_D_16x16; i_cost = B->l[0].me16.cost;
    if (B->l[1].me16.cost < i_cost) { i_cost = B->l[1].me16.cost; i_type = S_B_L1_L1; }
    if (B->cost16bi < i_cost) { i_cost = B->cost16bi; i_type = S_B_BI_BI; }
    if (B->cost16direct < i_cost) { i_cost = B->cost16direct; i_type = S_B_DIRECT; }
    if (S->mbrd && B->cost16direct <= i_cost * 33 / 32) {
        analyse_b_rd(S, m, A, i_cost);
        if (i_bskip_cost < B->rd16direct && i_bskip_cost < B->rd16bi && i_bskip_cost < B->l[0].rd16 && i_bskip_cost < B->l[1].rd16) {
            m->type = S_B_SKIP;
            update_cache(S, m, A);
            return;
        }
    }
    if (p->inter & 0x100) {                                /* X264_ANALYSE_BSUB16x16 */
        analyse_b8x8(S, m, B);
        if (B->cost8bi < i_cost) {
            i_type = S_B_8x8; i_partition = S_D_8x8; i_cost = B->cost8bi;
            if (m->sub[0] == m->sub[1] || m->sub[2] == m->sub[3]) {
                analyse_b16x8(S, m, B, 0);
                if (B->cost16x8bi < i_cost) { i_cost = B->cost16x8bi; i_type = B->type16x8; i_partition = S_D_16x8; }
            }
            if (m->sub[0] == m->sub[2] || m->sub[1] == m->sub[3]) {
                analyse_b16x8(S, m, B, 1);
                if (B->cost8x16bi < i_cost) { i_cost = B->cost8x16bi; i_type = B->type8x16; i_partition = S_D_8x16; }
            }
        }
    }
    if (S->mbrd) {
        /* refine later */
    } else if (i_partition == S_D_16x16) {                 /* :2586-2608 */
        B->l[0].me16.cost -= S->lambda * s_mb_b_cost[S_B_L0_L0];
        B->l[1].me16.cost -= S->lambda * s_mb_b_cost[S_B_L1_L1];
        if (i_type == S_B_L0_L0) { b_refine_qpel(S, m, 0, X264HIP_PIXEL_16x16, 0, 0, &B->l[0].me16); i_cost = B->l[0].me16.cost + S->lambda * s_mb_b_cost[S_B_L0_L0]; }
        else if (i_type == S_B_L1_L1) { b_refine_qpel(S, m, 1, X264HIP_PIXEL_16x16, 0, 0, &B->l[1].me16); i_cost = B->l[1].me16.cost + S->lambda * s_mb_b_cost[S_B_L1_L1]; }
        else if (i_type == S_B_BI_BI) { b_refine_qpel(S, m, 0, X264HIP_PIXEL_16x16, 0, 0, &B->l[0].me16); b_refine_qpel(S, m, 1, X264HIP_PIXEL_16x16, 0, 0, &B->l[1].me16); }
    } else if (i_partition == S_D_16x8) {
        for (int i = 0; i < 2; i++) {
            if (B->part16x8[i] != S_D_L1_8x8) b_refine_qpel(S, m, 0, X264HIP_PIXEL_16x8, 0, 8 * i, &B->l[0].me16x8[i]);
            if (B->part16x8[i] != S_D_L0_8x8) b_refine_qpel(S, m, 1, X264HIP_PIXEL_16x8, 0, 8 * i, &B->l[1].me16x8[i]);
        }
    } else if (i_partition == S_D_8x16) {
        for (int i = 0; i < 2; i++) {
            if (B->part8x16[i] != S_D_L1_8x8) b_refine_qpel(S, m, 0, X264HIP_PIXEL_8x16, 8 * i, 0, &B->l[0].me8x16[i]);
            if (B->part8x16[i] != S_D_L0_8x8) b_refine_qpel(S, m, 1, X264HIP_PIXEL_8x16, 8 * i, 0, &B->l[1].me8x16[i]);
        }
    } else {
        for (int i = 0; i < 4; i++) {
            const int t = m->sub[i], b_bidir = t == S_D_BI_8x8;
            if (t == S_D_DIRECT_8x8) continue;
            for (int l = 0; l < 2; l++)
                if (b_sub_uses(t, l)) {
                    pme *me = &B->l[l].me8[i];
                    const int old = me->cost, type_cost = S->lambda * s_sub_mb_b_cost[l ? S_D_L1_8x8 : S_D_L0_8x8];
                    me->cost -= type_cost;
                    b_refine_qpel(S, m, l, X264HIP_PIXEL_8x8, 8 * (i & 1), 8 * (i >> 1), me);
                    if (!b_bidir) B->cost8bi += me->cost + type_cost - old;
                }
        }
    }
    if (S->mbrd) {                                         /* :2657-2675 */
        i_satd_inter = i_cost;
        analyse_b_rd(S, m, A, i_satd_inter);
        i_type = S_B_SKIP; i_cost = i_bskip_cost; i_partition = S_D_16x16;
        if (B->l[0].rd16 < i_cost) { i_cost = B->l[0].rd16; i_type = S_B_L0_L0; }
        if (B->l[1].rd16 < i_cost) { i_cost = B->l[1].rd16; i_type = S_B_L1_L1; }
        if (B->rd16bi < i_cost) { i_cost = B->rd16bi; i_type = S_B_BI_BI; }
        if (B->rd16direct < i_cost) { i_cost = B->rd16direct; i_type = S_B_DIRECT; }
        if (B->rd16x8bi < i_cost) { i_cost = B->rd16x8bi; i_type = B->type16x8; i_partition = S_D_16x8; }
        if (B->rd8x16bi < i_cost) { i_cost = B->rd8x16bi; i_type = B->type8x16; i_partition = S_D_8x16; }
        if (B->rd8bi < i_cost) { i_cost = B->rd8bi; i_type = S_B_8x8; i_partition = S_D_8x8; }
        m->type = i_type; m->partition = i_partition;
    }
    analyse_intra(S, m, i_satd_inter);                     /* without the RD levels the reference passes 0 here: only I_16x16 gets a cost */
    if (S->mbrd) {
        transform_rd(S, m, A, &i_satd_inter, &i_cost);
        intra_rd(S, m, A, i_satd_inter * 17 / 16);
    }
    if (m->satd_i16 < i_cost) { i_cost = m->satd_i16; i_type = S_I_16x16; }
    if (m->satd_i8 < i_cost) { i_cost = m->satd_i8; i_type = S_I_8x8; }
    if (m->satd_i4 < i_cost) { i_cost = m->satd_i4; i_type = S_I_4x4; }
    if (satd_pcm < i_cost) { i_cost = satd_pcm; i_type = S_I_PCM; }
    m->type = i_type; m->partition = i_partition;
    if (S->mbrd >= 2 && S_IS_INTRA(i_type) && i_type != S_I_PCM) intra_rd_refine(S, m);     /* :2702-2703 */
    if (p->subme >= 5) refine_bidir(S, m, B);
    if (S->mbrd >= 2) refine_b_rd(S, m, A);                                                    /* :2707-2758 */
}
