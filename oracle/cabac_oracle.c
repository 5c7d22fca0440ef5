/* cabac_oracle.c -- TEST INFRASTRUCTURE; textually included by slice_oracle.c (it uses that file's
 * ssl / smb state).
 *
 * CPU restatement of the reference's CABAC side of the per-macroblock loop:
 *   the arithmetic coder                       R/common/cabac.c:787-927
 *   the bit-counting variant used by the RD    R/common/cabac.h:74-101, R/encoder/rdo.c:49-63
 *   levels (8.8 fixed-point bits, no output)
 *   x264_macroblock_write_cabac (I and P)      R/encoder/cabac.c:32-1022
 *   the RD-only partial writers                R/encoder/cabac.c:1024-1126
 * One body serves both uses: `rd` selects bit counting, where the reference's RDO_SKIP_BS build of the
 * same file differs only in (a) contexts that are not updated (`_noup`), (b) bypass bins counted as 256,
 * (c) the order in which a residual block's flags and levels are visited (cabac.c:679-763) -- which
 * matters for 8x8 blocks, whose significance contexts are shared between positions -- and (d) the
 * I_16x16-without-coefficients QP side effect of mb_qp_delta being switched off (cabac.c:274).
 * Tables: oracle/cabac_tables.h (generated, data only).                                              */
#include "cabac_tables.h"


/* x264_cabac_context_init, R/common/cabac.c:787-805 */
static void cb_context_init(o_cabac *cb, int slice_type, int qp, int model)
{
    const int8_t (*t)[2] = o_cabac_init_mn[slice_type == S_SLICE_I ? 0 : 1 + model];
    for (int i = 0; i < 460; i++) cb->state[i] = (u8)clip3i(((t[i][0] * qp) >> 4) + t[i][1], 1, 126);
}
/* x264_cabac_encode_init, :807-816 */
static void cb_encode_init(o_cabac *cb, u8 *data, u8 *end)
{
    cb->low = 0; cb->range = 0x1FE; cb->queue = -1; cb->outstanding = 0;
    cb->start = cb->p = data; cb->end = end; cb->f8 = 0;
}
static int cb_pos(const o_cabac *cb) { return (int)(cb->p - cb->start + cb->outstanding) * 8 + cb->queue; }
/* x264_cabac_putbyte, :818-850 */
static void cb_putbyte(o_cabac *cb)
{
    if (cb->queue < 8) return;
    int out = cb->low >> (cb->queue + 2);
    cb->low &= (4 << cb->queue) - 1;
    cb->queue -= 8;
    if ((out & 0xff) == 0xff) { cb->outstanding++; return; }
    int carry = out >> 8;
    cb->p[-1] += carry;                      /* may touch the byte before the payload: the slice header's last byte */
    for (; cb->outstanding > 0; cb->outstanding--) *cb->p++ = (u8)(carry - 1);
    *cb->p++ = (u8)out;
}
static void cb_renorm(o_cabac *cb)
{
    int shift = o_cabac_renorm_shift[cb->range >> 3];
    cb->range <<= shift; cb->low <<= shift; cb->queue += shift;
    cb_putbyte(cb);
}
/* x264_cabac_encode_decision_c, :861-873 */
static void cb_encode_decision(o_cabac *cb, int ctx, int b)
{
    int st = cb->state[ctx], lps = o_cabac_range_lps[st][(cb->range >> 6) & 3];
    cb->range -= lps;
    if (b != (st >> 6)) { cb->low += cb->range; cb->range = lps; }
    cb->state[ctx] = o_cabac_transition[st][b];
    cb_renorm(cb);
}
static void cb_encode_bypass(o_cabac *cb, int b)
{
    cb->low <<= 1; cb->low += -b & cb->range; cb->queue += 1;
    cb_putbyte(cb);
}
/* x264_cabac_encode_ue_bypass, :883-900 */
static void cb_encode_ue_bypass(o_cabac *cb, int exp_bits, int val)
{
    int k, i;
    uint32_t x;
    for (k = exp_bits; val >= (1 << k); k++) val -= 1 << k;
    x = (((1u << (k - exp_bits)) - 1) << (k + 1)) + val;
    k = 2 * k + 1 - exp_bits;
    i = ((k - 1) & 7) + 1;
    do {
        k -= i;
        cb->low <<= i; cb->low += ((x >> k) & 0xff) * cb->range; cb->queue += i;
        cb_putbyte(cb);
        i = 8;
    } while (k > 0);
}
static void cb_encode_terminal(o_cabac *cb) { cb->range -= 2; cb_renorm(cb); }
/* x264_cabac_encode_flush, :908-927 (i_frame: frames coded before this one) */
static void cb_encode_flush(o_cabac *cb, int i_frame)
{
    cb->low += cb->range - 2; cb->low |= 1; cb->low <<= 9; cb->queue += 9;
    cb_putbyte(cb); cb_putbyte(cb);
    cb->low <<= 8 - cb->queue;
    cb->low |= (0x35a4e4f5 >> (i_frame & 31) & 1) << 10;
    cb->queue = 8;
    cb_putbyte(cb);
    for (; cb->outstanding > 0; cb->outstanding--) *cb->p++ = 0xff;
}

/* the four primitives, writing or counting */
static inline void cbd(o_cabac *cb, int rd, int ctx, int b)
{
    if (!rd) { cb_encode_decision(cb, ctx, b); return; }
    int st = cb->state[ctx];
    cb->state[ctx] = o_cabac_transition[st][b];
    cb->f8 += o_cabac_entropy[st][b];
}
static inline void cbd_noup(o_cabac *cb, int rd, int ctx, int b)
{
    if (!rd) cb_encode_decision(cb, ctx, b);
    else cb->f8 += o_cabac_entropy[cb->state[ctx]][b];
}
static inline void cbb(o_cabac *cb, int rd, int b) { if (!rd) cb_encode_bypass(cb, b); else cb->f8 += 256; }
static inline void cb_ue(o_cabac *cb, int rd, int e, int v)
{
    if (!rd) cb_encode_ue_bypass(cb, e, v);
    else cb->f8 += (s_ue_size(v + (1 << e) - 1) - e) << 8;           /* rdo.c:57 */
}
/* cabac_size_unary / cabac_transition_unary, cabac_size_5ones (x264_rdo_init, rdo.c:326-358), evaluated on the fly */
static int cb_unary(u8 *state, int prefix)
{
    int bits = 0, st = *state;
    for (int i = 1; i < prefix; i++) { bits += o_cabac_entropy[st][1]; st = o_cabac_transition[st][1]; }
    if (prefix > 0 && prefix < 14) { bits += o_cabac_entropy[st][0]; st = o_cabac_transition[st][0]; }
    *state = (u8)st;
    return bits + 256;
}

/* ------------------------------------------------------------------ macroblock syntax elements */
static void cw_mb_type_intra(o_cabac *cb, int rd, const smb *m, int type, int c0, int c1, int c2, int c3, int c4, int c5)
{   /* x264_cabac_mb_type_intra, R/encoder/cabac.c:32-62 */
    if (type == S_I_4x4 || type == S_I_8x8) cbd_noup(cb, rd, c0, 0);
    else if (type == S_I_PCM) { cbd_noup(cb, rd, c0, 1); if (!rd) cb_encode_flush(cb, cb->i_frame); }
    else {
        int pred = s_fix16[m->i16mode];
        cbd_noup(cb, rd, c0, 1);
        if (!rd) cb_encode_terminal(cb); else cb->f8 += o_cabac_entropy[cb->state[276]][0];
        cbd_noup(cb, rd, c1, !!m->cbp_luma);
        if (m->cbp_chroma == 0) cbd_noup(cb, rd, c2, 0);
        else { cbd(cb, rd, c2, 1); cbd_noup(cb, rd, c3, m->cbp_chroma != 1); }
        cbd(cb, rd, c4, pred >> 1);
        cbd_noup(cb, rd, c5, pred & 1);
    }
}
static void cw_mb_type(const ssl *S, o_cabac *cb, int rd, const smb *m)
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
    } else if (m->type == S_P_L0) {
        cbd_noup(cb, rd, 14, 0);
        if (m->partition == S_D_16x16) { cbd_noup(cb, rd, 15, 0); cbd_noup(cb, rd, 16, 0); }
        else { cbd_noup(cb, rd, 15, 1); cbd_noup(cb, rd, 17, m->partition == S_D_16x8); }
    } else if (m->type == S_P_8x8) {
        cbd_noup(cb, rd, 14, 0); cbd_noup(cb, rd, 15, 0); cbd_noup(cb, rd, 16, 1);
    } else {
        cbd_noup(cb, rd, 14, 1);
        cw_mb_type_intra(cb, rd, m, m->type, 17 + 0, 17 + 1, 17 + 2, 17 + 2, 17 + 3, 17 + 3);
    }
}
static void cw_intra4x4_pred_mode(o_cabac *cb, int rd, int pred, int mode)
{   /* :198-211 */
    if (pred == mode) { cbd(cb, rd, 68, 1); return; }
    cbd(cb, rd, 68, 0);
    if (mode > pred) mode--;
    cbd(cb, rd, 69, mode & 1); cbd(cb, rd, 69, (mode >> 1) & 1); cbd(cb, rd, 69, (mode >> 2) & 1);
}
static void cw_chroma_pred_mode(o_cabac *cb, int rd, const smb *m)
{   /* :213-231; the neighbours' modes are stored already "fixed", DC (0) for anything not intra */
    int mode = s_fix8c[m->chroma_mode], ctx = (m->cpm_left != 0) + (m->cpm_top != 0);
    cbd_noup(cb, rd, 64 + ctx, mode > 0);
    if (mode > 0) {
        cbd(cb, rd, 64 + 3, mode > 1);
        if (mode > 1) cbd_noup(cb, rd, 64 + 3, mode > 2);
    }
}
static void cw_cbp_luma(o_cabac *cb, int rd, const smb *m)
{   /* :233-242 */
    int cbp = m->cbp_luma, l = m->cbp_left, t = m->cbp_top;
    cbd(cb, rd, 76 - ((l >> 1) & 1) - ((t >> 1) & 2), cbp & 1);
    cbd(cb, rd, 76 - ((cbp >> 0) & 1) - ((t >> 2) & 2), (cbp >> 1) & 1);
    cbd(cb, rd, 76 - ((l >> 3) & 1) - ((cbp << 1) & 2), (cbp >> 2) & 1);
    cbd_noup(cb, rd, 76 - ((cbp >> 2) & 1) - ((cbp >> 0) & 2), (cbp >> 3) & 1);
}
static void cw_cbp_chroma(o_cabac *cb, int rd, const smb *m)
{   /* :244-263 */
    int a = m->cbp_left & 0x30, b = m->cbp_top & 0x30, ctx = 0;
    if (a && m->cbp_left != -1) ctx++;
    if (b && m->cbp_top != -1) ctx += 2;
    if (m->cbp_chroma == 0) { cbd_noup(cb, rd, 77 + ctx, 0); return; }
    cbd_noup(cb, rd, 77 + ctx, 1);
    ctx = 4 + (a == 0x20) + 2 * (b == 0x20);
    cbd_noup(cb, rd, 77 + ctx, m->cbp_chroma > 1);
}
static void cw_qp_delta(ssl *S, o_cabac *cb, int rd, smb *m)
{   /* :265-297 */
    int dqp = m->qp - S->last_qp, ctx;
    if (m->type == S_I_16x16 && !(m->cbp_luma | m->cbp_chroma | m->nnz[24] | m->nnz[25] | m->nnz[26])) {   /* !h->mb.cbp[mb] */
        if (!rd) m->qp = S->last_qp;
        dqp = 0;
    }
    ctx = S->last_dqp && (S->fdec->mb_type[S->prev_mb] == S_I_16x16 || (S->cbp[S->prev_mb] & 0x3f));
    if (dqp) {
        int val = dqp <= 0 ? -2 * dqp : 2 * dqp - 1;
        if (val >= 51 && val != 52) val = 103 - val;
        while (val--) { cbd(cb, rd, 60 + ctx, 1); ctx = 2 + (ctx >> 1); }
    }
    cbd_noup(cb, rd, 60 + ctx, 0);
}
static void cw_mb_skip(const ssl *S, o_cabac *cb, const smb *m, int b_skip)
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
}
static void cw_sub_p_partition(o_cabac *cb, int rd, int sub)
{   /* :309-330 */
    if (sub == S_D_L0_8x8) { cbd(cb, rd, 21, 1); return; }
    cbd(cb, rd, 21, 0);
    if (sub == S_D_L0_8x4) { cbd(cb, rd, 22, 0); return; }
    cbd(cb, rd, 22, 1);
    cbd(cb, rd, 23, sub == S_D_L0_4x8);
}
static void cw_ref_l(o_cabac *cb, int rd, const smb *m, int list, int idx)
{   /* x264_cabac_mb_ref, :375-395 (h->mb.cache.skip is all zero in a P slice) */
    const int8_t *cref = CREF(m, list);
    const int i8 = s_scan8(idx), refa = cref[i8 - 1], refb = cref[i8 - 8];
    int ref = cref[i8], ctx = (refa > 0 && !m->cskip[i8 - 1]) + 2 * (refb > 0 && !m->cskip[i8 - 8]);
    for (; ref > 0; ref--) { cbd(cb, rd, 54 + ctx, 1); ctx = (ctx >> 2) + 4; }
    cbd(cb, rd, 54 + ctx, 0);
}
static void cw_ref(o_cabac *cb, int rd, const smb *m, int idx) { cw_ref_l(cb, rd, m, 0, idx); }
static void cw_mvd_cpn(o_cabac *cb, int rd, const smb *m, int list, int idx, int l, int mvd)
{   /* x264_cabac_mb_mvd_cpn, :397-445 */
    static const u8 ctxes[9] = {0, 3, 4, 5, 6, 6, 6, 6, 6};
    const i16 (*cmvd)[2] = CMVD(m, list);
    const int i8 = s_scan8(idx), amvd = abs(cmvd[i8 - 1][l]) + abs(cmvd[i8 - 8][l]), a = abs(mvd), base = l ? 47 : 40;
    const int ctx = (amvd > 2) + (amvd > 32);
    if (a == 0) { cbd(cb, rd, base + ctx, 0); return; }
    cbd(cb, rd, base + ctx, 1);
    if (a < 9) {
        if (rd && a > 4) {
            for (int i = 1; i < 4; i++) cbd(cb, rd, base + ctxes[i], 1);
            cb->f8 += cb_unary(&cb->state[base + 6], a - 3);
        } else {
            for (int i = 1; i < a; i++) cbd(cb, rd, base + ctxes[i], 1);
            cbd(cb, rd, base + ctxes[a], 0);
            cbb(cb, rd, mvd < 0);
        }
    } else if (rd) {
        for (int i = 1; i < 4; i++) cbd(cb, rd, base + ctxes[i], 1);
        for (int i = 0; i < 5; i++) cbd(cb, rd, base + 6, 1);                  /* cabac_size_5ones */
        cb->f8 += 256;
        cb_ue(cb, rd, 3, a - 9);
    } else {
        for (int i = 1; i < 9; i++) cbd(cb, rd, base + ctxes[i], 1);
        cb_ue(cb, rd, 3, a - 9);
        cbb(cb, rd, mvd < 0);
    }
}
static void cw_mvd_l(o_cabac *cb, int rd, smb *m, int list, int idx, int width, int height)
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
static int b_sub_uses(int sub, int list) { return sub == S_D_DIRECT_8x8 ? 0 : list ? sub >= 4 && sub <= 11 : sub <= 3 || (sub >= 8 && sub <= 11); }
static void cw_mb8x8_mvd(o_cabac *cb, int rd, smb *m, int i)
{   /* x264_cabac_mb8x8_mvd, :465-498 (list 0) */
    switch (m->sub[i]) {
    case S_D_L0_8x8: cw_mvd(cb, rd, m, 4 * i, 2, 2); break;
    case S_D_L0_8x4: cw_mvd(cb, rd, m, 4 * i, 2, 1); cw_mvd(cb, rd, m, 4 * i + 2, 2, 1); break;
    case S_D_L0_4x8: cw_mvd(cb, rd, m, 4 * i, 1, 2); cw_mvd(cb, rd, m, 4 * i + 1, 1, 2); break;
    default: for (int k = 0; k < 4; k++) cw_mvd(cb, rd, m, 4 * i + k, 1, 1); break;
    }
}

/* neighbouring non_zero_count as h->mb.cache holds it: 0x80 where there is no neighbour (R/common/macroblock.c:917-985) */
static int nz_left(const smb *m, int idx)
{
    if (idx < 16) return blk_x[idx] ? m->nnz[idx - (idx & 1 ? 1 : 3)] : m->nz_l[blk_y[idx] >> 2];
    const int k = (idx - 16) & 3, ch = (idx - 16) >> 2;
    return (k & 1) ? m->nnz[idx - 1] : m->nz_lc[ch][k >> 1];
}
static int nz_top(const smb *m, int idx)
{
    if (idx < 16) return blk_y[idx] ? m->nnz[idx - (idx & 2 ? 2 : 6)] : m->nz_t[blk_x[idx] >> 2];
    const int k = (idx - 16) & 3, ch = (idx - 16) >> 2;
    return (k & 2) ? m->nnz[idx - 2] : m->nz_tc[ch][k & 1];
}
static int cw_cbf_ctx(const smb *m, int cat, int idx)
{   /* x264_cabac_mb_cbf_ctxidxinc, :508-538 */
    const int intra = S_IS_INTRA(m->type);
    int a, b;
    switch (cat) {
    case 1: case 2: case 4:
        a = nz_left(m, idx) & (0x7f + (intra << 7)); b = nz_top(m, idx) & (0x7f + (intra << 7));
        return 4 * cat + 2 * !!b + !!a;
    case 0:
        return 4 * cat + 2 * ((m->cbp_top >> 8) & 1) + ((m->cbp_left >> 8) & 1);
    default:
        idx -= 25;
        a = m->cbp_left != -1 ? (m->cbp_left >> (9 + idx)) & 1 : intra;
        b = m->cbp_top != -1 ? (m->cbp_top >> (9 + idx)) & 1 : intra;
        return 4 * cat + 2 * b + a;
    }
}

static const u16 cw_sig_off[6] = {105, 120, 134, 149, 152, 402}, cw_last_off[6] = {166, 181, 195, 210, 213, 417};
static const u16 cw_level_off[6] = {227, 237, 247, 257, 266, 426};
static const u8 cw_sig8[63] = {                                         /* significant_coeff_flag_offset_8x8[0], ITU-T H.264 table 9-43 (frame) */
    0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7,
    7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};
static const u8 cw_last8[63] = {
    0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2,
    3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
/* node -> context of "level is 1" / "level > 1", and the node after either (:570-581) */
static const u8 cw_lvl1_ctx[8] = {1, 2, 3, 4, 0, 0, 0, 0}, cw_lvlgt1_ctx[8] = {5, 5, 5, 5, 6, 7, 8, 9};
static const u8 cw_node_next[2][8] = {{1, 2, 3, 3, 4, 5, 6, 7}, {4, 4, 4, 4, 5, 6, 7, 7}};

static int coeff_last_n(const i16 *l, int n) { int i = n - 1; while (i >= 0 && !l[i]) i--; return i; }

/* block_residual_write_cabac: the writer's order (:584-674) when !rd, the RD order (:683-763) when rd.
 * cat: 0 luma DC, 1 luma AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8 (no coded-block flag).       */
static void cw_residual(o_cabac *cb, int rd, const smb *m, int cat, int idx, const i16 *l, int count)
{
    const int c_sig = cw_sig_off[cat], c_last = cw_last_off[cat], c_lvl = cw_level_off[cat], b8 = cat == 5;
    if (!b8) {
        int ctx = 85 + cw_cbf_ctx(m, cat, idx);
        if (!m->nnz[idx]) { cbd(cb, rd, ctx, 0); return; }
        cbd(cb, rd, ctx, 1);
    }
    const int last = coeff_last_n(l, count);
    int node = 0;
    if (!rd) {
        int am1[64], sign[64], n = 0, i, sigmap = last + 1 < count - 1 ? last + 1 : count - 1;
        for (i = 0; i < sigmap; i++) {
            const int cs = c_sig + (b8 ? cw_sig8[i] : i), cl = c_last + (b8 ? cw_last8[i] : i);
            if (l[i]) {
                am1[n] = abs(l[i]) - 1; sign[n] = l[i] < 0; n++;
                cb_encode_decision(cb, cs, 1);
                cb_encode_decision(cb, cl, i == last);
            } else
                cb_encode_decision(cb, cs, 0);
        }
        if (i == last) { am1[n] = abs(l[i]) - 1; sign[n] = l[i] < 0; n++; }
        do {
            n--;
            const int prefix = am1[n] < 14 ? am1[n] : 14;
            int ctx = cw_lvl1_ctx[node] + c_lvl;
            if (prefix) {
                cb_encode_decision(cb, ctx, 1);
                ctx = cw_lvlgt1_ctx[node] + c_lvl;
                for (i = 0; i < prefix - 1; i++) cb_encode_decision(cb, ctx, 1);
                if (prefix < 14) cb_encode_decision(cb, ctx, 0);
                else cb_encode_ue_bypass(cb, 0, am1[n] - 14);
                node = cw_node_next[1][node];
            } else {
                cb_encode_decision(cb, ctx, 0);
                node = cw_node_next[0][node];
            }
            cb_encode_bypass(cb, sign[n]);
        } while (n > 0);
        return;
    }
    /* bit counting: the last coefficient first, then backwards, flags and level of a position together */
    for (int i = last; i >= 0; i--) {
        const int cs = c_sig + (b8 ? cw_sig8[i] : i), cl = c_last + (b8 ? cw_last8[i] : i);
        if (i == last) {
            if (last != count - 1) { cbd(cb, 1, cs, 1); cbd(cb, 1, cl, 1); }
        } else if (l[i]) { cbd(cb, 1, cs, 1); cbd(cb, 1, cl, 0); }
        else { cbd(cb, 1, cs, 0); continue; }
        const int am1 = abs(l[i]) - 1, prefix = am1 < 14 ? am1 : 14;
        int ctx = cw_lvl1_ctx[node] + c_lvl;
        if (prefix) {
            cbd(cb, 1, ctx, 1);
            ctx = cw_lvlgt1_ctx[node] + c_lvl;
            cb->f8 += cb_unary(&cb->state[ctx], prefix);
            if (prefix >= 14) cb_ue(cb, 1, 0, am1 - 14);
            node = cw_node_next[1][node];
        } else {
            cbd(cb, 1, ctx, 0);
            node = cw_node_next[0][node];
            cb->f8 += 256;
        }
    }
}

/* x264_macroblock_write_cabac, :781-1022 (I and P slices).  Writes (rd = 0) or counts (rd = 1, then it is
 * x264_macroblock_size_cabac).  I_PCM only when writing (:796-822).                                         */
static void cw_macroblock(ssl *S, o_cabac *cb, int rd, smb *m)
{
    const int type = m->type;
    cw_mb_type(S, cb, rd, m);
    if (!rd && type == S_I_PCM) {
        memcpy(cb->p, m->fe[0], 256); cb->p += 256;          /* FENC stride is 16: the luma block is contiguous */
        for (int pl = 1; pl < 3; pl++) { for (int i = 0; i < 8; i++) memcpy(cb->p + i * 8, m->fe[pl] + i * FENC, 8); cb->p += 64; }
        cb->low = 0; cb->range = 0x1FE; cb->queue = -1; cb->outstanding = 0;
        for (int y = 0; y < 16; y++) memcpy(m->fd[0] + y * FDEC, m->fe[0] + y * FENC, 16);
        for (int pl = 1; pl < 3; pl++) for (int y = 0; y < 8; y++) memcpy(m->fd[pl] + y * FDEC, m->fe[pl] + y * FENC, 8);
        return;
    }
    if (S_IS_INTRA(type)) {
        if (S->p->transform8x8 && type != S_I_16x16) cbd_noup(cb, rd, 399 + m->nb_t8, m->t8);
        if (type != S_I_16x16)
            for (int i = 0; i < 16; i += type == S_I_8x8 ? 4 : 1)
                cw_intra4x4_pred_mode(cb, rd, pred_intra4x4_mode(m, i), s_fix4[m->i4c[s_scan8(i)] + 1]);
        cw_chroma_pred_mode(cb, rd, m);
    } else if (type == S_P_L0) {
        const int multi = S->n_ref > 1;
        if (m->partition == S_D_16x16) {
            if (multi) cw_ref(cb, rd, m, 0);
            cw_mvd(cb, rd, m, 0, 4, 4);
        } else if (m->partition == S_D_16x8) {
            if (multi) { cw_ref(cb, rd, m, 0); cw_ref(cb, rd, m, 8); }
            cw_mvd(cb, rd, m, 0, 4, 2); cw_mvd(cb, rd, m, 8, 4, 2);
        } else {
            if (multi) { cw_ref(cb, rd, m, 0); cw_ref(cb, rd, m, 4); }
            cw_mvd(cb, rd, m, 0, 2, 4); cw_mvd(cb, rd, m, 4, 2, 4);
        }
    } else if (type == S_P_8x8) {
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
    }
    if (type != S_I_16x16) { cw_cbp_luma(cb, rd, m); cw_cbp_chroma(cb, rd, m); }
    if (S->p->transform8x8 && m->cbp_luma && s_t8_allowed(S, m)) cbd_noup(cb, rd, 399 + m->nb_t8, m->t8);
    if (m->cbp_luma > 0 || m->cbp_chroma > 0 || type == S_I_16x16) {
        cw_qp_delta(S, cb, rd, m);
        if (type == S_I_16x16) {
            cw_residual(cb, rd, m, 0, 24, m->dc16, 16);
            if (m->cbp_luma) for (int i = 0; i < 16; i++) cw_residual(cb, rd, m, 1, i, m->luma4[i] + 1, 15);
        } else if (m->t8) {
            for (int i = 0; i < 4; i++) if (m->cbp_luma & (1 << i)) cw_residual(cb, rd, m, 5, 4 * i, m->luma8[i], 64);
        } else
            for (int i = 0; i < 16; i++) if (m->cbp_luma & (1 << (i >> 2))) cw_residual(cb, rd, m, 2, i, m->luma4[i], 16);
        if (m->cbp_chroma & 3) { cw_residual(cb, rd, m, 3, 25, m->cdc[0], 4); cw_residual(cb, rd, m, 3, 26, m->cdc[1], 4); }
        if (m->cbp_chroma & 2) for (int i = 16; i < 24; i++) cw_residual(cb, rd, m, 4, i, m->cac[i - 16] + 1, 15);
    }
}

/* ------------------------------------------------------------------ the RD-only partial writers, R/encoder/cabac.c:1024-1126
 * (bit counting only: "doesn't write cbp or chroma dc, doesn't write ref or subpartition") */
static void cw_partition_size(const ssl *S, o_cabac *cb, smb *m, int i8, int pix)
{   /* x264_partition_size_cabac, :1032-1081 */
    static const u8 pix_h[7] = {16, 8, 16, 8, 4, 8, 4};
    const int b_8x16 = m->partition == S_D_8x16;
    (void)S;
    if (m->type == S_P_8x8) cw_mb8x8_mvd(cb, 1, m, i8);
    else if (m->type == S_P_L0) cw_mvd(cb, 1, m, 4 * i8, 4 >> b_8x16, 2 << b_8x16);
    else if (m->type > S_B_DIRECT && m->type < S_B_8x8) {
        if (b_type_uses(m->type, 0, !!i8)) cw_mvd_l(cb, 1, m, 0, 4 * i8, 4 >> b_8x16, 2 << b_8x16);
        if (b_type_uses(m->type, 1, !!i8)) cw_mvd_l(cb, 1, m, 1, 4 * i8, 4 >> b_8x16, 2 << b_8x16);
    } else if (m->type == S_B_8x8) {
        for (int l = 0; l < 2; l++) if (b_sub_uses(m->sub[i8], l)) cw_mvd_l(cb, 1, m, l, 4 * i8, 2, 2);      /* x264_cabac_mb8x8_mvd: no B partition below 8x8 */
    } else
        return;
    for (int j = pix < X264HIP_PIXEL_8x8; j >= 0; j--) {
        if (m->cbp_luma & (1 << i8)) {
            if (m->t8) cw_residual(cb, 1, m, 5, 4 * i8, m->luma8[i8], 64);
            else for (int i4 = 0; i4 < 4; i4++) cw_residual(cb, 1, m, 2, i4 + 4 * i8, m->luma4[i4 + 4 * i8], 16);
        }
        cw_residual(cb, 1, m, 4, 16 + i8, m->cac[i8] + 1, 15);
        cw_residual(cb, 1, m, 4, 20 + i8, m->cac[4 + i8] + 1, 15);
        i8 += pix_h[pix] >> 3;
    }
}
static void cw_subpartition_size(o_cabac *cb, smb *m, int i4, int pix)
{   /* x264_subpartition_size_cabac, :1083-1095 */
    const int b_8x4 = pix == X264HIP_PIXEL_8x4;
    cw_residual(cb, 1, m, 2, i4, m->luma4[i4], 16);
    if (pix == X264HIP_PIXEL_4x4) cw_mvd(cb, 1, m, i4, 1, 1);
    else {
        cw_mvd(cb, 1, m, i4, 1 + b_8x4, 2 - b_8x4);
        cw_residual(cb, 1, m, 2, i4 + 2 - b_8x4, m->luma4[i4 + 2 - b_8x4], 16);
    }
}
static void cw_partition_i8x8_size(o_cabac *cb, smb *m, int i8, int mode)
{   /* x264_partition_i8x8_size_cabac, :1097-1105 */
    cw_intra4x4_pred_mode(cb, 1, pred_intra4x4_mode(m, 4 * i8), s_fix4[mode + 1]);
    cw_cbp_luma(cb, 1, m);
    if (m->cbp_luma & (1 << i8)) cw_residual(cb, 1, m, 5, 4 * i8, m->luma8[i8], 64);
}
static void cw_partition_i4x4_size(o_cabac *cb, smb *m, int i4, int mode)
{   /* x264_partition_i4x4_size_cabac, :1107-1113 */
    cw_intra4x4_pred_mode(cb, 1, pred_intra4x4_mode(m, i4), s_fix4[mode + 1]);
    cw_residual(cb, 1, m, 2, i4, m->luma4[i4], 16);
}
static void cw_i8x8_chroma_size(o_cabac *cb, smb *m)
{   /* x264_i8x8_chroma_size_cabac, :1115-1131 */
    cw_chroma_pred_mode(cb, 1, m);
    cw_cbp_chroma(cb, 1, m);
    if (m->cbp_chroma > 0) {
        cw_residual(cb, 1, m, 3, 25, m->cdc[0], 4); cw_residual(cb, 1, m, 3, 26, m->cdc[1], 4);
        if (m->cbp_chroma == 2) for (int i = 16; i < 24; i++) cw_residual(cb, 1, m, 4, i, m->cac[i - 16] + 1, 15);
    }
}
