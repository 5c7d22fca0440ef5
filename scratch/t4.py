import sys
p='/root/repo/oracle/cabac_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:110]); sys.exit(1)
    s=s.replace(a,b)
rep('''static void cw_mb_type(const ssl *S, o_cabac *cb, int rd, const smb *m)
{   /* x264_cabac_mb_type, :64-196 (I and P slices) */
    if (S->slice_type == S_SLICE_I) {
        int ctx = (m->type_left >= 0 && m->type_left != S_I_4x4) + (m->type_top >= 0 && m->type_top != S_I_4x4);
        cw_mb_type_intra(cb, rd, m, m->type, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7);
    } else if (m->type == S_P_L0) {''','''static void cw_mb_type(const ssl *S, o_cabac *cb, int rd, const smb *m)
{   /* x264_cabac_mb_type, :64-196 */
    if (S->slice_type == S_SLICE_I) {
        int ctx = (m->type_left >= 0 && m->type_left != S_I_4x4) + (m->type_top >= 0 && m->type_top != S_I_4x4);
        cw_mb_type_intra(cb, rd, m, m->type, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7);
    } else if (S->slice_type == S_SLICE_B) {             /* :126-190 */
        const int ctx = (m->type_left >= 0 && m->type_left != S_B_SKIP && m->type_left != S_B_DIRECT)
                      + (m->type_top >= 0 && m->type_top != S_B_SKIP && m->type_top != S_B_DIRECT);
        if (m->type == S_B_DIRECT) cbd_noup(cb, rd, 27 + ctx, 0);
        else if (m->type == S_B_8x8) {
            cbd_noup(cb, rd, 27 + ctx, 1); cbd_noup(cb, rd, 27 + 3, 1); cbd_noup(cb, rd, 27 + 4, 1);
            cbd(cb, rd, 27 + 5, 1); cbd(cb, rd, 27 + 5, 1); cbd_noup(cb, rd, 27 + 5, 1);
        } else if (S_IS_INTRA(m->type)) {
            cbd_noup(cb, rd, 27 + ctx, 1); cbd_noup(cb, rd, 27 + 3, 1); cbd_noup(cb, rd, 27 + 4, 1);
            cbd(cb, rd, 27 + 5, 1); cbd(cb, rd, 27 + 5, 0); cbd(cb, rd, 27 + 5, 1);
            cw_mb_type_intra(cb, rd, m, m->type, 32 + 0, 32 + 1, 32 + 2, 32 + 2, 32 + 3, 32 + 3);
        } else {
            /* the bin strings of table 9-37 for the 16x8 / 8x16 / 16x16 forms of the nine list combinations */
            static const u8 len[9 * 3] = {6, 6, 3, 6, 6, 0, 7, 7, 0, 6, 6, 0, 6, 6, 3, 7, 7, 0, 7, 7, 0, 7, 7, 0, 7, 7, 6};
            static const u8 bits[9 * 3][7] = {
                {1, 1, 0, 0, 0, 1}, {1, 1, 0, 0, 1, 0}, {1, 0, 0},
                {1, 1, 0, 1, 0, 1}, {1, 1, 0, 1, 1, 0}, {0},
                {1, 1, 1, 0, 0, 0, 0}, {1, 1, 1, 0, 0, 0, 1}, {0},
                {1, 1, 0, 1, 1, 1}, {1, 1, 1, 1, 1, 0}, {0},
                {1, 1, 0, 0, 1, 1}, {1, 1, 0, 1, 0, 0}, {1, 0, 1},
                {1, 1, 1, 0, 0, 1, 0}, {1, 1, 1, 0, 0, 1, 1}, {0},
                {1, 1, 1, 0, 1, 0, 0}, {1, 1, 1, 0, 1, 0, 1}, {0},
                {1, 1, 1, 0, 1, 1, 0}, {1, 1, 1, 0, 1, 1, 1}, {0},
                {1, 1, 1, 1, 0, 0, 0}, {1, 1, 1, 1, 0, 0, 1}, {1, 1, 0, 0, 0, 0}};
            const int idx = (m->type - S_B_L0_L0) * 3 + (m->partition - S_D_16x8);
            cbd_noup(cb, rd, 27 + ctx, bits[idx][0]);
            cbd_noup(cb, rd, 27 + 3, bits[idx][1]);
            cbd(cb, rd, 27 + 5 - bits[idx][1], bits[idx][2]);
            for (int i = 3; i < len[idx]; i++) cbd(cb, rd, 27 + 5, bits[idx][i]);
        }
    } else if (m->type == S_P_L0) {''')
rep('''static void cw_mb_skip(const ssl *S, o_cabac *cb, const smb *m, int b_skip)
{   /* x264_cabac_mb_skip, :300-306 */
    int ctx = (m->type_left >= 0 && m->type_left != S_P_SKIP) + (m->type_top >= 0 && m->type_top != S_P_SKIP) + 11;
    (void)S;
    cb_encode_decision(cb, ctx, b_skip);
}''','''static void cw_mb_skip(const ssl *S, o_cabac *cb, const smb *m, int b_skip)
{   /* x264_cabac_mb_skip, :300-306 */
    int ctx = (m->type_left >= 0 && !S_IS_SKIP(m->type_left)) + (m->type_top >= 0 && !S_IS_SKIP(m->type_top)) + (S->slice_type == S_SLICE_P ? 11 : 24);
    cb_encode_decision(cb, ctx, b_skip);
}
static void cw_sub_b_partition(o_cabac *cb, int rd, int sub)
{   /* x264_cabac_mb_sub_b_partition, :332-367 */
    static const u8 part_bits[12][7] = {{6, 1, 1, 1, 0, 1, 1}, {5, 1, 1, 0, 0, 1}, {5, 1, 1, 0, 1, 0}, {3, 1, 0, 0}, {5, 1, 1, 1, 1, 0}, {5, 1, 1, 0, 1, 1},
                                        {6, 1, 1, 1, 0, 0, 0}, {3, 1, 0, 1}, {5, 1, 1, 1, 1, 1}, {6, 1, 1, 1, 0, 0, 1}, {6, 1, 1, 1, 0, 1, 0}, {5, 1, 1, 0, 0, 0}};
    if (sub == S_D_DIRECT_8x8) { cbd(cb, rd, 36, 0); return; }
    const int len = part_bits[sub][0];
    cbd(cb, rd, 36, part_bits[sub][1]);
    cbd(cb, rd, 37, part_bits[sub][2]);
    if (len == 3) cbd(cb, rd, 39, part_bits[sub][3]);
    else {
        cbd(cb, rd, 38, part_bits[sub][3]); cbd(cb, rd, 39, part_bits[sub][4]); cbd(cb, rd, 39, part_bits[sub][5]);
        if (len == 6) cbd(cb, rd, 39, part_bits[sub][6]);
    }
}''')
rep('''static void cw_ref(o_cabac *cb, int rd, const smb *m, int idx)
{   /* x264_cabac_mb_ref, :375-395 (list 0; no skip flags in a P slice) */
    const int i8 = s_scan8(idx), refa = m->cref[i8 - 1], refb = m->cref[i8 - 8];
    int ref = m->cref[i8], ctx = (refa > 0) + 2 * (refb > 0);''','''static void cw_ref_l(o_cabac *cb, int rd, const smb *m, int list, int idx)
{   /* x264_cabac_mb_ref, :375-395 (h->mb.cache.skip is all zero in a P slice) */
    const int8_t *cref = CREF(m, list);
    const int i8 = s_scan8(idx), refa = cref[i8 - 1], refb = cref[i8 - 8];
    int ref = cref[i8], ctx = (refa > 0 && !m->cskip[i8 - 1]) + 2 * (refb > 0 && !m->cskip[i8 - 8]);''')
rep('''    cbd(cb, rd, 54 + ctx, 0);
}
static void cw_mvd_cpn(o_cabac *cb, int rd, const smb *m, int idx, int l, int mvd)
{   /* x264_cabac_mb_mvd_cpn, :397-445 */
    static const u8 ctxes[9] = {0, 3, 4, 5, 6, 6, 6, 6, 6};
    const int i8 = s_scan8(idx), amvd = abs(m->cmvd[i8 - 1][l]) + abs(m->cmvd[i8 - 8][l]), a = abs(mvd), base = l ? 47 : 40;''','''    cbd(cb, rd, 54 + ctx, 0);
}
static void cw_ref(o_cabac *cb, int rd, const smb *m, int idx) { cw_ref_l(cb, rd, m, 0, idx); }
static void cw_mvd_cpn(o_cabac *cb, int rd, const smb *m, int list, int idx, int l, int mvd)
{   /* x264_cabac_mb_mvd_cpn, :397-445 */
    static const u8 ctxes[9] = {0, 3, 4, 5, 6, 6, 6, 6, 6};
    const i16 (*cmvd)[2] = CMVD(m, list);
    const int i8 = s_scan8(idx), amvd = abs(cmvd[i8 - 1][l]) + abs(cmvd[i8 - 8][l]), a = abs(mvd), base = l ? 47 : 40;''')
rep('''static void cw_mvd(o_cabac *cb, int rd, smb *m, int idx, int width, int height)
{   /* x264_cabac_mb_mvd, :447-463: vector minus its prediction, and the difference goes into the mvd cache */
    i16 mvp[2];
    const int i8 = s_scan8(idx);
    predict_mv_blk(m, idx, width, mvp);
    int dx = m->cmv[i8][0] - mvp[0], dy = m->cmv[i8][1] - mvp[1];
    cw_mvd_cpn(cb, rd, m, idx, 0, dx);
    cw_mvd_cpn(cb, rd, m, idx, 1, dy);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) { m->cmvd[i8 + x + 8 * y][0] = (i16)dx; m->cmvd[i8 + x + 8 * y][1] = (i16)dy; }
}''','''static void cw_mvd_l(o_cabac *cb, int rd, smb *m, int list, int idx, int width, int height)
{   /* x264_cabac_mb_mvd, :447-463: vector minus its prediction, and the difference goes into the mvd cache */
    i16 mvp[2];
    const int i8 = s_scan8(idx);
    predict_mv_blk_l(m, list, idx, width, mvp);
    int dx = CMV(m, list)[i8][0] - mvp[0], dy = CMV(m, list)[i8][1] - mvp[1];
    cw_mvd_cpn(cb, rd, m, list, idx, 0, dx);
    cw_mvd_cpn(cb, rd, m, list, idx, 1, dy);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) { CMVD(m, list)[i8 + x + 8 * y][0] = (i16)dx; CMVD(m, list)[i8 + x + 8 * y][1] = (i16)dy; }
}
static void cw_mvd(o_cabac *cb, int rd, smb *m, int idx, int width, int height) { cw_mvd_l(cb, rd, m, 0, idx, width, height); }
/* x264_mb_type_list_table (R/common/macroblock.h:94-106) for the B types with explicit lists: does partition `part` use list `list`? */
static int b_type_uses(int type, int list, int part)
{
    static const u8 t[9][2][2] = {{{1, 1}, {0, 0}}, {{1, 0}, {0, 1}}, {{1, 1}, {0, 1}}, {{0, 1}, {1, 0}}, {{0, 0}, {1, 1}}, {{0, 1}, {1, 1}},
                                  {{1, 1}, {1, 0}}, {{1, 0}, {1, 1}}, {{1, 1}, {1, 1}}};
    return t[type - S_B_L0_L0][list][part];
}
/* x264_mb_partition_listX_table (:140-156) for the 8x8 sub-partitions */
static int b_sub_uses(int sub, int list) { return sub == S_D_DIRECT_8x8 ? 0 : list ? sub >= 4 && sub <= 11 : sub <= 3 || (sub >= 8 && sub <= 11); }''')
rep('''    } else if (type == S_P_8x8) {
        for (int i = 0; i < 4; i++) cw_sub_p_partition(cb, rd, m->sub[i]);
        if (S->n_ref > 1) for (int i = 0; i < 4; i++) cw_ref(cb, rd, m, 4 * i);
        for (int i = 0; i < 4; i++) cw_mb8x8_mvd(cb, rd, m, i);
    }''','''    } else if (type == S_P_8x8) {
        for (int i = 0; i < 4; i++) cw_sub_p_partition(cb, rd, m->sub[i]);
        if (S->n_ref > 1) for (int i = 0; i < 4; i++) cw_ref(cb, rd, m, 4 * i);
        for (int i = 0; i < 4; i++) cw_mb8x8_mvd(cb, rd, m, i);
    } else if (type == S_B_8x8) {                        /* :894-916 (x264 uses no sub-8x8 B partitions) */
        for (int i = 0; i < 4; i++) cw_sub_b_partition(cb, rd, m->sub[i]);
        for (int list = 0; list < 2; list++) {
            if ((list ? S->n_ref1 : S->n_ref) == 1) continue;
            for (int i = 0; i < 4; i++) if (b_sub_uses(m->sub[i], list)) cw_ref_l(cb, rd, m, list, 4 * i);
        }
        for (int list = 0; list < 2; list++)
            for (int i = 0; i < 4; i++) if (b_sub_uses(m->sub[i], list)) cw_mvd_l(cb, rd, m, list, 4 * i, 2, 2);
    } else if (type != S_B_DIRECT) {                     /* :917-962: the B types with explicit lists */
        const int n = m->partition == S_D_16x16 ? 1 : 2, step = m->partition == S_D_16x8 ? 8 : 4;
        const int w = m->partition == S_D_8x16 ? 2 : 4, h = m->partition == S_D_16x8 ? 2 : 4;
        for (int list = 0; list < 2; list++)
            if ((list ? S->n_ref1 : S->n_ref) > 1)
                for (int i = 0; i < n; i++) if (b_type_uses(type, list, i)) cw_ref_l(cb, rd, m, list, step * i);
        for (int list = 0; list < 2; list++)
            for (int i = 0; i < n; i++) if (b_type_uses(type, list, i)) cw_mvd_l(cb, rd, m, list, step * i, w, h);
    }''')
open(p,'w').write(s)
print("ok")
