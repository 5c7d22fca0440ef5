/* rd_oracle.c -- TEST INFRASTRUCTURE; textually included by slice_oracle.c after cabac_oracle.c.
 *
 * CPU restatement of the reference's rate-distortion glue (R/encoder/rdo.c):
 *   psy-RD distortion: cached source complexity and ssd_plane   analyse.c:509-537, rdo.c:66-137
 *   x264_rd_cost_mb                                              rdo.c:139-171
 *   trellis quantisation against the live CABAC contexts        rdo.c:320-660
 * The callers (x264_mb_analyse_p_rd, x264_intra_rd, x264_mb_analyse_transform_rd) sit in
 * slice_oracle.c's analyse_mb.                                                                      */

/* ---- psy-RD: x264_mb_cache_fenc_satd, R/encoder/analyse.c:509-537 ---- */
static void cache_fenc_satd(const ssl *S, smb *m)
{
    static u8 zero[16];
    if (!S->psy_rd) return;
    m->fenc_satd_sum = m->fenc_sa8d_sum = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            u8 *fe = m->fe[0] + 4 * x + 4 * y * FENC;
            m->fenc_satd[y][x] = pixf.satd[X264HIP_PIXEL_4x4](zero, 0, fe, FENC) - (pixf.sad[X264HIP_PIXEL_4x4](zero, 0, fe, FENC) >> 1);
            m->fenc_satd_sum += m->fenc_satd[y][x];
        }
    for (int y = 0; y < 2; y++)
        for (int x = 0; x < 2; x++) {
            u8 *fe = m->fe[0] + 8 * x + 8 * y * FENC;
            m->fenc_sa8d[y][x] = pixf.sa8d[X264HIP_PIXEL_8x8](zero, 0, fe, FENC) - (pixf.sad[X264HIP_PIXEL_8x8](zero, 0, fe, FENC) >> 2);
            m->fenc_sa8d_sum += m->fenc_sa8d[y][x];
        }
}
static const u8 s_pix_w[7] = {16, 16, 8, 8, 8, 4, 4}, s_pix_h[7] = {16, 8, 16, 8, 4, 8, 4};
/* ssd_plane, rdo.c:106-130 */
static int ssd_plane(const ssl *S, const smb *m, int size, int p, int x, int y)
{
    static u8 zero[16];
    int satd = 0;
    u8 *fd = m->fd[p] + x + y * FDEC, *fe = m->fe[p] + x + y * FENC;
    if (p == 0 && S->psy_rd) {
        int s4 = 0, s8 = 0;
        if (size == X264HIP_PIXEL_16x16) { s4 = m->fenc_satd_sum; s8 = m->fenc_sa8d_sum; }
        else {
            for (int j = y >> 2; j < (y >> 2) + (s_pix_h[size] >> 2); j++)
                for (int i = x >> 2; i < (x >> 2) + (s_pix_w[size] >> 2); i++) s4 += m->fenc_satd[j][i];
            for (int j = y >> 3; j < (y >> 3) + (s_pix_h[size] >> 3); j++)
                for (int i = x >> 3; i < (x >> 3) + (s_pix_w[size] >> 3); i++) s8 += m->fenc_sa8d[j][i];
        }
        if (size <= X264HIP_PIXEL_8x8) {
            uint64_t acs = pixf.hadamard_ac[size](fd, FDEC);
            satd = abs((int32_t)acs - s4) + abs((int32_t)(acs >> 32) - s8);
            satd >>= 1;
        } else {
            int dc = pixf.sad[size](fd, FDEC, zero, 0) >> 1;
            satd = abs(pixf.satd[size](fd, FDEC, zero, 0) - dc - s4);
        }
        satd = (satd * S->psy_rd * s_lambda_tab[m->qp] + 128) >> 8;
    }
    return pixf.ssd[size](fe, FENC, fd, FDEC) + satd;
}
static int ssd_mb(const ssl *S, const smb *m)
{
    return ssd_plane(S, m, X264HIP_PIXEL_16x16, 0, 0, 0) + ssd_plane(S, m, X264HIP_PIXEL_8x8, 1, 0, 0) + ssd_plane(S, m, X264HIP_PIXEL_8x8, 2, 0, 0);
}

/* x264_rd_cost_mb, rdo.c:139-171 (CABAC).  Like the reference it leaves m->type as the encode left it
 * (a P 16x16 without coefficients on the skip vector has become P_SKIP).                              */
static void encode_mb(ssl *S, smb *m);
static int rd_cost_mb(ssl *S, smb *m, int lambda2)
{
    const int t8_bak = m->t8;
    int bits;
    encode_mb(S, m);
    const int ssd = ssd_mb(S, m);
    if (S_IS_SKIP(m->type)) bits = (1 * lambda2 + 128) >> 8;
    else {
        o_cabac tmp;
        tmp.f8 = 0;
        memcpy(tmp.state, S->cb.state, 460);
        cw_macroblock(S, &tmp, 1, m);
        bits = (int)(((uint64_t)tmp.f8 * lambda2 + 32768) >> 16);
    }
    m->t8 = t8_bak;
    return ssd + bits;
}

/* ---- trellis quantisation, rdo.c:320-660 ---- */
static const int s_trellis_lambda2[2][52] = {        /* lambda2_tab, rdo.c:362-383: [0] inter, [1] intra */
    {46, 58, 73, 92, 117, 147, 185, 233, 294, 370, 466, 587, 740, 932, 1174, 1480, 1864, 2349, 2959, 3728, 4697, 5918, 7457, 9395,
     11837, 14914, 18790, 23674, 29828, 37581, 47349, 59656, 75163, 94699, 119313, 150326, 189399, 238627, 300652, 378798,
     477255, 601304, 757596, 954511, 1202608, 1515192, 1909022, 2405217, 3030384, 3818045, 4810435, 6060769},
    {27, 34, 43, 54, 68, 86, 108, 136, 172, 216, 273, 343, 433, 545, 687, 865, 1090, 1374, 1731, 2180, 2747, 3461, 4361, 5494,
     6922, 8721, 10988, 13844, 17442, 21976, 27688, 34885, 43953, 55377, 69771, 87906, 110755, 139543, 175813, 221511,
     279087, 351627, 443023, 558174, 703255, 886046, 1116348, 1406511, 1772093, 2232697, 2813022, 3544186}};
#define TRELLIS_INF ((int64_t)1 << 50)
typedef struct { int64_t score; int lv; u8 st[10]; } tnode;

/* quant_trellis_cabac, rdo.c:411-628.  dct is in raster order, zz maps scan position -> raster index;
 * dc: the block is a DC block (one multiplier, weight 256); b_ac: scan position 0 is not part of it.     */
static int trellis_quant_twin(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                              int cat, int lambda2, int b_ac, int dc, int n_coef);
static int trellis_quant(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                         int cat, int lambda2, int b_ac, int dc, int n_coef)
{
#ifdef X264O_DEVCHECK
    i16 copy[64];
    memcpy(copy, dct, n_coef * sizeof(i16));
    const int r2 = devhost_trellis(copy, mf, unq, weight, zz, S->cb.state, cat, lambda2, b_ac, dc, n_coef);
    const int r1 = trellis_quant_twin(S, dct, mf, unq, weight, zz, cat, lambda2, b_ac, dc, n_coef);
    g_devcheck_calls++;
    if (r1 != r2 || memcmp(copy, dct, n_coef * sizeof(i16))) { if (!g_devcheck_bad) fprintf(stderr, "devcheck: trellis differs (cat %d, n %d)\n", cat, n_coef); g_devcheck_bad++; }
    return r1;
#else
    return trellis_quant_twin(S, dct, mf, unq, weight, zz, cat, lambda2, b_ac, dc, n_coef);
#endif
}
static int trellis_quant_twin(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                              int cat, int lambda2, int b_ac, int dc, int n_coef)
{
    int abs_c[64], sgn[64];
    tnode nodes[2][8], *cur = nodes[0], *prev = nodes[1];
    u8 st_sig[64], st_last[64];
    u16 lvl_abs[64 * 8 * 2], lvl_next[64 * 8 * 2];
    int n_lvl = 1, i, j;
    const int f = 1 << 15;
    const u8 *cs = S->cb.state;

    for (i = n_coef - 1; i >= b_ac; i--)
        if ((unsigned)(dct[zz[i]] * (dc ? mf[0] >> 1 : mf[zz[i]]) + f - 1) >= 2u * f) break;
    if (i < b_ac) { memset(dct, 0, n_coef * sizeof(*dct)); return 0; }
    const int last_nnz = i;
    for (j = 0; j < n_coef; j++) sgn[j] = 1;         /* positions past the last candidate keep level 0 */
    for (; i >= b_ac; i--) { int c = dct[zz[i]]; abs_c[i] = abs(c); sgn[i] = c < 0 ? -1 : 1; }

    for (j = 1; j < 8; j++) cur[j].score = TRELLIS_INF;
    cur[0].score = 0; cur[0].lv = 0;
    lvl_abs[0] = 0; lvl_next[0] = 0;

    if (n_coef == 64)
        for (i = 0; i < 63; i++) { st_sig[i] = cs[cw_sig_off[cat] + cw_sig8[i]]; st_last[i] = cs[cw_last_off[cat] + cw_last8[i]]; }
    else {
        const int k = (!dc || cat != 3) ? 15 : 3;
        memcpy(st_sig, cs + cw_sig_off[cat], k); memcpy(st_last, cs + cw_last_off[cat], k);
    }
    memcpy(cur[0].st, cs + cw_level_off[cat], 10);

    for (i = last_nnz; i >= b_ac; i--) {
        const int coef = abs_c[i], q = (f + coef * (dc ? mf[0] >> 1 : mf[zz[i]])) >> 16;
        int cost_sig[2], cost_last[2];
        if (q == 0) {                                    /* only the "not significant" flag to pay, for every live node but 0 */
            const uint32_t c0 = (uint32_t)((uint64_t)o_cabac_entropy[st_sig[i]][0] * lambda2 >> 4);
            for (j = 1; j < 8; j++)
                if (cur[j].score != TRELLIS_INF) {
                    lvl_abs[n_lvl] = 0; lvl_next[n_lvl] = (u16)cur[j].lv; cur[j].lv = n_lvl++;
                    cur[j].score += c0;
                }
            continue;
        }
        { tnode *t = cur; cur = prev; prev = t; }
        for (j = 0; j < 8; j++) cur[j].score = TRELLIS_INF;
        if (i < n_coef - 1) {
            cost_sig[0] = o_cabac_entropy[st_sig[i]][0]; cost_sig[1] = o_cabac_entropy[st_sig[i]][1];
            cost_last[0] = o_cabac_entropy[st_last[i]][0]; cost_last[1] = o_cabac_entropy[st_last[i]][1];
        } else
            cost_sig[0] = cost_sig[1] = cost_last[0] = cost_last[1] = 0;

        for (int lvl = q; lvl >= q - 1; lvl--) {
            const int unq_lvl = ((dc ? unq[0] << 1 : unq[zz[i]]) * lvl + 128) >> 8, d = coef - unq_lvl;
            const int64_t ssd = (int64_t)d * d * (dc ? 256 : weight[i]);
            for (j = 0; j < 8; j++) {
                int node = j;
                if (prev[j].score == TRELLIS_INF) continue;
                tnode n = prev[j];
                if (lvl || node) {
                    unsigned bits = cost_sig[lvl != 0];
                    if (lvl) {
                        const int prefix = lvl - 1 < 14 ? lvl - 1 : 14;
                        u8 *c1 = &n.st[cw_lvl1_ctx[node]];
                        bits += cost_last[node == 0];
                        bits += o_cabac_entropy[*c1][prefix > 0]; *c1 = o_cabac_transition[*c1][prefix > 0];
                        if (prefix > 0) {
                            bits += cb_unary(&n.st[cw_lvlgt1_ctx[node]], prefix);
                            if (lvl >= 15) bits += s_ue_size(lvl - 15) << 8;
                            node = cw_node_next[1][node];
                        } else {
                            bits += 256;
                            node = cw_node_next[0][node];
                        }
                    }
                    n.score += (int64_t)((uint64_t)bits * lambda2 >> 4);
                }
                n.score += ssd;
                if (n.score < cur[node].score) {
                    lvl_abs[n_lvl] = (u16)lvl; lvl_next[n_lvl] = (u16)n.lv; n.lv = n_lvl++;
                    cur[node] = n;
                }
            }
        }
    }
    const tnode *b = &cur[0];
    for (j = 1; j < 8; j++) if (cur[j].score < b->score) b = &cur[j];
    int nz = 0;
    j = b->lv;
    for (i = b_ac; i < n_coef; i++) {
        dct[zz[i]] = (i16)(lvl_abs[j] * sgn[i]);
        nz |= lvl_abs[j];
        j = lvl_next[j];
    }
    return !!nz;
}
