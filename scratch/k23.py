import sys
def patch(p, pairs):
    s=open(p).read()
    for a,b in pairs:
        if s.count(a)!=1: print("MISMATCH",p,s.count(a),a[:100]); sys.exit(1)
        s=s.replace(a,b)
    open(p,'w').write(s)
C='/root/repo/x264_vs2008_amd/csrc/'
patch(C+'mbsyn.h',[('''    int slice_type;             // 0 P, 2 I''','''    int slice_type;             // 0 P, 1 B, 2 I'''),
('''    int n_ref;                  // h->mb.pic.i_fref[0]''','''    int n_ref, n_ref1;          // h->mb.pic.i_fref[0] / [1]'''),
('''    int16_t cmv[48][2], cmvd[48][2];   // h->mb.cache.mv[0] / mvd[0]''','''    int16_t cmv[48][2], cmvd[48][2];   // h->mb.cache.mv[0] / mvd[0]
    signed char cref1[48];      // list 1 of the same (B slices)
    int16_t cmv1[48][2], cmvd1[48][2];
    signed char cskip[48];      // h->mb.cache.skip: direct blocks, whose references do not count in a reference index's context''')])
patch(C+'cabac_dev.h',[
('''enum { CD_I_4x4 = 0, CD_I_8x8 = 1, CD_I_16x16 = 2, CD_I_PCM = 3, CD_P_L0 = 4, CD_P_8x8 = 5, CD_P_SKIP = 6 };
enum { CD_D_L0_4x4 = 0, CD_D_L0_8x4 = 1, CD_D_L0_4x8 = 2, CD_D_L0_8x8 = 3, CD_D_8x8 = 13, CD_D_16x8 = 14, CD_D_8x16 = 15, CD_D_16x16 = 16 };''',
'''enum { CD_I_4x4 = 0, CD_I_8x8 = 1, CD_I_16x16 = 2, CD_I_PCM = 3, CD_P_L0 = 4, CD_P_8x8 = 5, CD_P_SKIP = 6,
       CD_B_DIRECT = 7, CD_B_L0_L0 = 8, CD_B_8x8 = 17, CD_B_SKIP = 18 };
enum { CD_D_L0_4x4 = 0, CD_D_L0_8x4 = 1, CD_D_L0_4x8 = 2, CD_D_L0_8x8 = 3, CD_D_L1_8x8 = 7, CD_D_BI_8x8 = 11, CD_D_DIRECT_8x8 = 12,
       CD_D_8x8 = 13, CD_D_16x8 = 14, CD_D_8x16 = 15, CD_D_16x16 = 16 };
#define CD_IS_SKIP(t_) ((t_) == CD_P_SKIP || (t_) == CD_B_SKIP)
// the motion caches of either list
#define CD_CREF(m_, l_) ((l_) ? (m_).cref1 : (m_).cref)
#define CD_CMV(m_, l_) ((l_) ? (m_).cmv1 : (m_).cmv)
#define CD_CMVD(m_, l_) ((l_) ? (m_).cmvd1 : (m_).cmvd)
// x264_mb_type_list_table (R/common/macroblock.h:94-106): does partition `part` of B type `t` (B_L0_L0 .. B_BI_BI) use list `l`?
// rows: L0L0 L0L1 L0BI L1L0 L1L1 L1BI BIL0 BIL1 BIBI; four bits each: l0p0 l0p1 l1p0 l1p1
#define CD_B_USES(t_, l_, part_) ((int)((0xfdebe7c6a593ull >> (4 * ((t_) - CD_B_L0_L0) + 2 * (l_) + (part_))) & 1))
// x264_mb_partition_listX_table for the 8x8 sub-partitions (:140-156)
#define CD_SUB_USES(s_, l_) ((s_) == CD_D_DIRECT_8x8 ? 0 : (l_) ? ((s_) >= 4 && (s_) <= 11) : ((s_) <= 3 || ((s_) >= 8 && (s_) <= 11)))'''),
# predict_mv: list-generic
('''template <class MS> CD_FN void cd_predict_mv(const MS &m, int idx, int width, int &px, int &py)
{
    const int i8 = cd_scan8(idx), i_ref = m.cref[i8];
    int ra = m.cref[i8 - 1], rb = m.cref[i8 - 8], kc = i8 - 8 + width, rc = m.cref[kc];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { kc = i8 - 8 - 1; rc = m.cref[kc]; }
    const int ax = m.cmv[i8 - 1][0], ay = m.cmv[i8 - 1][1], bx = m.cmv[i8 - 8][0], by = m.cmv[i8 - 8][1], cx = m.cmv[kc][0], cy = m.cmv[kc][1];''',
'''template <class MS> CD_FN void cd_predict_mv(const MS &m, int list, int idx, int width, int &px, int &py)
{
    const int i8 = cd_scan8(idx), i_ref = CD_CREF(m, list)[i8];
    int ra = CD_CREF(m, list)[i8 - 1], rb = CD_CREF(m, list)[i8 - 8], kc = i8 - 8 + width, rc = CD_CREF(m, list)[kc];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { kc = i8 - 8 - 1; rc = CD_CREF(m, list)[kc]; }
    const int ax = CD_CMV(m, list)[i8 - 1][0], ay = CD_CMV(m, list)[i8 - 1][1], bx = CD_CMV(m, list)[i8 - 8][0], by = CD_CMV(m, list)[i8 - 8][1],
              cx = CD_CMV(m, list)[kc][0], cy = CD_CMV(m, list)[kc][1];'''),
# mb_type B
('''{   // x264_cabac_mb_type, :64-196 (I and P slices)
    if (m.slice_type == 2) {
        const int ctx = (m.type_left >= 0 && m.type_left != CD_I_4x4) + (m.type_top >= 0 && m.type_top != CD_I_4x4);
        cw_mb_type_intra(cb, st, rd, m, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7, i_frame);
    } else if (m.type == CD_P_L0) {''','''{   // x264_cabac_mb_type, :64-196
    if (m.slice_type == 2) {
        const int ctx = (m.type_left >= 0 && m.type_left != CD_I_4x4) + (m.type_top >= 0 && m.type_top != CD_I_4x4);
        cw_mb_type_intra(cb, st, rd, m, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7, i_frame);
    } else if (m.slice_type == 1) {                       // :126-190
        const int ctx = (m.type_left >= 0 && m.type_left != CD_B_SKIP && m.type_left != CD_B_DIRECT)
                      + (m.type_top >= 0 && m.type_top != CD_B_SKIP && m.type_top != CD_B_DIRECT);
        if (m.type == CD_B_DIRECT) cdd_noup(cb, st, rd, 27 + ctx, 0);
        else if (m.type == CD_B_8x8) {
            cdd_noup(cb, st, rd, 27 + ctx, 1); cdd_noup(cb, st, rd, 27 + 3, 1); cdd_noup(cb, st, rd, 27 + 4, 1);
            cdd(cb, st, rd, 27 + 5, 1); cdd(cb, st, rd, 27 + 5, 1); cdd_noup(cb, st, rd, 27 + 5, 1);
        } else if (m.type <= CD_I_PCM) {
            cdd_noup(cb, st, rd, 27 + ctx, 1); cdd_noup(cb, st, rd, 27 + 3, 1); cdd_noup(cb, st, rd, 27 + 4, 1);
            cdd(cb, st, rd, 27 + 5, 1); cdd(cb, st, rd, 27 + 5, 0); cdd(cb, st, rd, 27 + 5, 1);
            cw_mb_type_intra(cb, st, rd, m, 32 + 0, 32 + 1, 32 + 2, 32 + 2, 32 + 3, 32 + 3, i_frame);
        } else {
            // the bin strings of the 16x8 / 8x16 / 16x16 forms of the nine list combinations: length << 8 | bits, first bin in bit 0
            const int idx = (m.type - CD_B_L0_L0) * 3 + (m.partition - CD_D_16x8);
            const u32 code = d_cw_b_bins[idx];
            const int len = (int)(code >> 8), b1 = (int)((code >> 1) & 1);
            cdd_noup(cb, st, rd, 27 + ctx, (int)(code & 1));
            cdd_noup(cb, st, rd, 27 + 3, b1);
            cdd(cb, st, rd, 27 + 5 - b1, (int)((code >> 2) & 1));
            for (int i = 3; i < len; i++) cdd(cb, st, rd, 27 + 5, (int)((code >> i) & 1));
        }
    } else if (m.type == CD_P_L0) {'''),
('''// x264_cabac_mb_skip, :300-306 (P slices)
template <class ST> CD_FN void cw_mb_skip(DCabac &cb, ST st, int type_left, int type_top, int b_skip)
{
    const int ctx = (type_left >= 0 && type_left != CD_P_SKIP) + (type_top >= 0 && type_top != CD_P_SKIP) + 11;
    cd_encode_decision(cb, st, ctx, b_skip);
}''','''// x264_cabac_mb_skip, :300-306
template <class ST> CD_FN void cw_mb_skip(DCabac &cb, ST st, int type_left, int type_top, int b_skip, int slice_type = 0)
{
    const int ctx = (type_left >= 0 && !CD_IS_SKIP(type_left)) + (type_top >= 0 && !CD_IS_SKIP(type_top)) + (slice_type == 0 ? 11 : 24);
    cd_encode_decision(cb, st, ctx, b_skip);
}
template <class ST> CD_FN void cw_sub_b_partition(DCabac &cb, ST st, int rd, int sub)
{   // x264_cabac_mb_sub_b_partition, :332-367 (only the 8x8 shapes: x264 uses no smaller B partition)
    if (sub == CD_D_DIRECT_8x8) { cdd(cb, st, rd, 36, 0); return; }
    cdd(cb, st, rd, 36, 1);
    if (sub == CD_D_BI_8x8) { cdd(cb, st, rd, 37, 1); cdd(cb, st, rd, 38, 0); cdd(cb, st, rd, 39, 0); cdd(cb, st, rd, 39, 0); }
    else { cdd(cb, st, rd, 37, 0); cdd(cb, st, rd, 39, sub == CD_D_L1_8x8); }
}'''),
('''template <class ST, class MS> CD_FN void cw_ref(DCabac &cb, ST st, int rd, const MS &m, int idx)
{   // x264_cabac_mb_ref, :375-395 (list 0, P slice)
    const int i8 = cd_scan8(idx), refa = m.cref[i8 - 1], refb = m.cref[i8 - 8];
    int ref = m.cref[i8], ctx = (refa > 0) + 2 * (refb > 0);''','''template <class ST, class MS> CD_FN void cw_ref(DCabac &cb, ST st, int rd, const MS &m, int idx, int list = 0)
{   // x264_cabac_mb_ref, :375-395 (h->mb.cache.skip is all zero in a P slice)
    const int i8 = cd_scan8(idx), refa = CD_CREF(m, list)[i8 - 1], refb = CD_CREF(m, list)[i8 - 8];
    int ref = CD_CREF(m, list)[i8], ctx = (refa > 0) + 2 * (refb > 0);
    if (m.slice_type == 1) ctx = (refa > 0 && !m.cskip[i8 - 1]) + 2 * (refb > 0 && !m.cskip[i8 - 8]);'''),
('''template <class ST, class MS> CD_FN void cw_mvd_cpn(DCabac &cb, ST st, int rd, const MS &m, int idx, int l, int mvd)
{   // x264_cabac_mb_mvd_cpn, :397-445
    const int i8 = cd_scan8(idx), amvd = cd_abs(m.cmvd[i8 - 1][l]) + cd_abs(m.cmvd[i8 - 8][l]), a = cd_abs(mvd), base = l ? 47 : 40;''',
'''template <class ST, class MS> CD_FN void cw_mvd_cpn(DCabac &cb, ST st, int rd, const MS &m, int list, int idx, int l, int mvd)
{   // x264_cabac_mb_mvd_cpn, :397-445
    const int i8 = cd_scan8(idx), amvd = cd_abs(CD_CMVD(m, list)[i8 - 1][l]) + cd_abs(CD_CMVD(m, list)[i8 - 8][l]), a = cd_abs(mvd), base = l ? 47 : 40;'''),
('''template <class ST, class MS> CD_FN void cw_mvd(DCabac &cb, ST st, int rd, MS &m, int idx, int width, int height)
{   // x264_cabac_mb_mvd, :447-463
    int px, py;
    const int i8 = cd_scan8(idx);
    cd_predict_mv(m, idx, width, px, py);
    const int dx = m.cmv[i8][0] - px, dy = m.cmv[i8][1] - py;
    cw_mvd_cpn(cb, st, rd, m, idx, 0, dx);
    cw_mvd_cpn(cb, st, rd, m, idx, 1, dy);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) { m.cmvd[i8 + x + 8 * y][0] = (i16)dx; m.cmvd[i8 + x + 8 * y][1] = (i16)dy; }
}''','''template <class ST, class MS> CD_FN void cw_mvd(DCabac &cb, ST st, int rd, MS &m, int idx, int width, int height, int list = 0)
{   // x264_cabac_mb_mvd, :447-463
    int px, py;
    const int i8 = cd_scan8(idx);
    cd_predict_mv(m, list, idx, width, px, py);
    const int dx = CD_CMV(m, list)[i8][0] - px, dy = CD_CMV(m, list)[i8][1] - py;
    cw_mvd_cpn(cb, st, rd, m, list, idx, 0, dx);
    cw_mvd_cpn(cb, st, rd, m, list, idx, 1, dy);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) { CD_CMVD(m, list)[i8 + x + 8 * y][0] = (i16)dx; CD_CMVD(m, list)[i8 + x + 8 * y][1] = (i16)dy; }
}'''),
('''    } else if (type == CD_P_8x8) {
        for (int i = 0; i < 4; i++) cw_sub_p_partition(cb, st, rd, m.sub[i]);
        if (m.n_ref > 1) for (int i = 0; i < 4; i++) cw_ref(cb, st, rd, m, 4 * i);
        for (int i = 0; i < 4; i++) cw_mb8x8_mvd(cb, st, rd, m, i);
    }''','''    } else if (type == CD_P_8x8) {
        for (int i = 0; i < 4; i++) cw_sub_p_partition(cb, st, rd, m.sub[i]);
        if (m.n_ref > 1) for (int i = 0; i < 4; i++) cw_ref(cb, st, rd, m, 4 * i);
        for (int i = 0; i < 4; i++) cw_mb8x8_mvd(cb, st, rd, m, i);
    } else if (type == CD_B_8x8) {                           // :894-916
        for (int i = 0; i < 4; i++) cw_sub_b_partition(cb, st, rd, m.sub[i]);
        for (int l = 0; l < 2; l++) {
            if ((l ? m.n_ref1 : m.n_ref) == 1) continue;
            for (int i = 0; i < 4; i++) if (CD_SUB_USES(m.sub[i], l)) cw_ref(cb, st, rd, m, 4 * i, l);
        }
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < 4; i++) if (CD_SUB_USES(m.sub[i], l)) cw_mvd(cb, st, rd, m, 4 * i, 2, 2, l);
    } else if (type > CD_B_DIRECT && type < CD_B_8x8) {      // :917-962: the B types with explicit lists
        const int n = m.partition == CD_D_16x16 ? 1 : 2, step = m.partition == CD_D_16x8 ? 8 : 4;
        const int w = m.partition == CD_D_8x16 ? 2 : 4, h = m.partition == CD_D_16x8 ? 2 : 4;
        for (int l = 0; l < 2; l++)
            if ((l ? m.n_ref1 : m.n_ref) > 1)
                for (int i = 0; i < n; i++) if (CD_B_USES(type, l, i)) cw_ref(cb, st, rd, m, step * i, l);
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < n; i++) if (CD_B_USES(type, l, i)) cw_mvd(cb, st, rd, m, step * i, w, h, l);
    }'''),
])
print('ok')
