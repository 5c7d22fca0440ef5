// cabac_dev.h -- the CABAC side of the per-macroblock loop for the slice kernel's RD levels and its slice payload:
//   the arithmetic coder                         R/common/cabac.c:787-927
//   bit counting for the RD levels               R/common/cabac.h:74-101, R/encoder/rdo.c:49-63 (8.8 fixed-point bits, no output)
//   x264_macroblock_write_cabac (I and P)        R/encoder/cabac.c:32-1022
// The entropy coder is a serial machine: one lane runs it while the other 63 wait, on state that lives in LDS (the 460
// context states, the macroblock's syntax record `MbSyn`) and in registers (low / range / queue).  The kernel gathers what
// the writer reads into an MbSyn with all lanes, then lane 0 walks it.  `rd` selects bit counting, where the reference's
// RDO_SKIP_BS build of the same source differs in: contexts that are not updated (_noup), bypass bins counted as 256, the
// order in which a residual block's flags and levels are visited (cabac.c:679-763; it matters for 8x8 blocks, whose
// significance contexts are shared between positions) and mb_qp_delta's side effect on the QP being switched off.
//
// The file is plain C++ so that the same text also compiles for the host (-DX264HIP_HOST_TEST): tests/ drive it there
// against the CPU restatement on every macroblock of a chain (no GPU needed), which is how it was brought up.
#pragma once
#include <stdint.h>
#ifdef X264HIP_HOST_TEST
#include <string.h>
#include <stdlib.h>
#define CD_FN static inline
#define __device__
typedef uint8_t u8; typedef int16_t i16; typedef uint16_t u16; typedef uint32_t u32;
#else
#define CD_FN __device__ __forceinline__
#endif
#include "cabac_tables.h"

#include "mbsyn.h"

// The coder's tables are read once or twice per bin by a single lane: from global memory that is two dependent ~500-cycle round trips
// per bin.  On the device they therefore live in LDS (cd_load_tables fills them once per kernel): entropy and next state of
// (state, bin) packed in one word, the four LPS ranges of a state in another.
#ifdef X264HIP_HOST_TEST
#define CD_ENT(s_, b_) ((int)d_cabac_entropy[s_][b_])
#define CD_TRANS(s_, b_) ((int)d_cabac_transition[s_][b_])
#define CD_LPS(s_, i_) ((int)d_cabac_range_lps[s_][i_])
#define CD_SIG8(i_) ((int)d_cw_sig8[i_])
#define CD_LAST8(i_) ((int)d_cw_last8[i_])
#else
static __shared__ u32 cd_lds_et[256];        // [state * 2 + bin]: entropy | next state << 16
static __shared__ u32 cd_lds_lps[128];       // [state]: range_lps[0..3], one byte each
static __shared__ u8 cd_lds_sig8[64], cd_lds_last8[64];
#define CD_ENT(s_, b_) ((int)(cd_lds_et[(s_) * 2 + (b_)] & 0xffffu))
#define CD_TRANS(s_, b_) ((int)(cd_lds_et[(s_) * 2 + (b_)] >> 16))
#define CD_LPS(s_, i_) ((int)((cd_lds_lps[s_] >> (8 * (i_))) & 255u))
#define CD_SIG8(i_) ((int)cd_lds_sig8[i_])
#define CD_LAST8(i_) ((int)cd_lds_last8[i_])
#endif

enum { CD_I_4x4 = 0, CD_I_8x8 = 1, CD_I_16x16 = 2, CD_I_PCM = 3, CD_P_L0 = 4, CD_P_8x8 = 5, CD_P_SKIP = 6,
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
#define CD_B_USES(t_, l_, part_) ((int)((0xfd7ec6b93ull >> (4 * ((t_) - CD_B_L0_L0) + 2 * (l_) + (part_))) & 1))
// x264_cabac_mb_type's bin strings of the 9 list combinations x {16x8, 8x16, 16x16} (R/encoder/cabac.c:150-176): length << 8 | bins, first bin in bit 0
static __device__ const u16 d_cw_b_bins[27] = {0x623, 0x613, 0x301, 0x62b, 0x61b, 0x0, 0x707, 0x747, 0x0, 0x63b, 0x61f, 0x0, 0x633, 0x60b, 0x305,
                                               0x727, 0x767, 0x0, 0x717, 0x757, 0x0, 0x737, 0x777, 0x0, 0x70f, 0x74f, 0x603};
// x264_mb_partition_listX_table for the 8x8 sub-partitions (:140-156)
#define CD_SUB_USES(s_, l_) ((s_) == CD_D_DIRECT_8x8 ? 0 : (l_) ? ((s_) >= 4 && (s_) <= 11) : ((s_) <= 3 || ((s_) >= 8 && (s_) <= 11)))

CD_FN int cd_clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
CD_FN int cd_abs(int v) { return v < 0 ? -v : v; }
CD_FN int cd_ue_size(unsigned v) { int n = 0; v++; while (v >> (n + 1)) n++; return 2 * n + 1; }      // bs_size_ue_big
CD_FN int cd_scan8(int i)
{   // x264_scan8, R/common/common.h:196-238
    if (i < 16) return 4 + 1 * 8 + ((i & 1) | ((i >> 1) & 2)) + 8 * (((i >> 1) & 1) | ((i >> 2) & 2));
    if (i < 20) return 1 + 1 * 8 + ((i - 16) & 1) + 8 * ((i - 16) >> 1);
    if (i < 24) return 1 + 4 * 8 + ((i - 20) & 1) + 8 * ((i - 20) >> 1);
    return 4 + 5 * 8 + (i - 24);
}

// x264_cabac_context_init, R/common/cabac.c:787-805: state i of the 460 (call with i = lane, lane + 64, ...)
CD_FN int cd_context_init_one(int i, int slice_type, int qp, int model)
{
    const int t = slice_type == 2 ? 0 : 1 + model;
    return cd_clip3(((d_cabac_init_mn[t][i][0] * qp) >> 4) + d_cabac_init_mn[t][i][1], 1, 126);
}
CD_FN void cd_encode_init(DCabac &cb, u8 *p) { cb.low = 0; cb.range = 0x1FE; cb.queue = -1; cb.outstanding = 0; cb.p = p; cb.f8 = 0; }
CD_FN int cd_pos(const DCabac &cb, const u8 *start) { return (int)(cb.p - start + cb.outstanding) * 8 + cb.queue; }
CD_FN void cd_putbyte(DCabac &cb)
{   // x264_cabac_putbyte, :818-850
    if (cb.queue < 8) return;
    const int out = cb.low >> (cb.queue + 2);
    cb.low &= (4 << cb.queue) - 1;
    cb.queue -= 8;
    if ((out & 0xff) == 0xff) { cb.outstanding++; return; }
    const int carry = out >> 8;
    if (carry) cb.p[-1] += (u8)carry;        // never reaches before the payload: that would be a probability above 1 (cabac.c:832-837)
    for (; cb.outstanding > 0; cb.outstanding--) *cb.p++ = (u8)(carry - 1);
    *cb.p++ = (u8)out;
}
CD_FN void cd_renorm(DCabac &cb)
{
    const int shift = __builtin_clz((unsigned)(((cb.range >> 3) << 1) | 1)) - 25;     // x264_cabac_renorm_shift[range >> 3], as arithmetic
    cb.range <<= shift; cb.low <<= shift; cb.queue += shift;
    cd_putbyte(cb);
}
template <class ST> CD_FN void cd_encode_decision(DCabac &cb, ST st, int ctx, int b)
{   // x264_cabac_encode_decision_c, :861-873
    const int s = st[ctx], lps = CD_LPS(s, (cb.range >> 6) & 3);
    cb.range -= lps;
    if (b != (s >> 6)) { cb.low += cb.range; cb.range = lps; }
    st[ctx] = CD_TRANS(s, b);
    cd_renorm(cb);
}
CD_FN void cd_encode_bypass(DCabac &cb, int b) { cb.low <<= 1; cb.low += -b & cb.range; cb.queue += 1; cd_putbyte(cb); }
CD_FN void cd_encode_ue_bypass(DCabac &cb, int exp_bits, int val)
{   // :883-900
    int k, i;
    for (k = exp_bits; val >= (1 << k); k++) val -= 1 << k;
    const u32 x = (((1u << (k - exp_bits)) - 1) << (k + 1)) + (u32)val;
    k = 2 * k + 1 - exp_bits;
    i = ((k - 1) & 7) + 1;
    do {
        k -= i;
        cb.low <<= i; cb.low += (int)((x >> k) & 0xff) * cb.range; cb.queue += i;
        cd_putbyte(cb);
        i = 8;
    } while (k > 0);
}
CD_FN void cd_encode_terminal(DCabac &cb) { cb.range -= 2; cd_renorm(cb); }
CD_FN void cd_encode_flush(DCabac &cb, int i_frame)
{   // x264_cabac_encode_flush, :908-927 (i_frame: frames coded before this one)
    cb.low += cb.range - 2; cb.low |= 1; cb.low <<= 9; cb.queue += 9;
    cd_putbyte(cb); cd_putbyte(cb);
    cb.low <<= 8 - cb.queue;
    cb.low |= (0x35a4e4f5 >> (i_frame & 31) & 1) << 10;
    cb.queue = 8;
    cd_putbyte(cb);
    for (; cb.outstanding > 0; cb.outstanding--) *cb.p++ = 0xff;
}

// writing (rd = 0) or counting (rd = 1)
template <class ST> CD_FN void cdd(DCabac &cb, ST st, int rd, int ctx, int b)
{
    if (!rd) { cd_encode_decision(cb, st, ctx, b); return; }
    const int s = st[ctx];
    st[ctx] = CD_TRANS(s, b);
    cb.f8 += CD_ENT(s, b);
}
template <class ST> CD_FN void cdd_noup(DCabac &cb, ST st, int rd, int ctx, int b)
{
    if (!rd) cd_encode_decision(cb, st, ctx, b);
    else cb.f8 += CD_ENT(st[ctx], b);
}
CD_FN void cdb(DCabac &cb, int rd, int b) { if (!rd) cd_encode_bypass(cb, b); else cb.f8 += 256; }
CD_FN void cd_ue(DCabac &cb, int rd, int e, int v)
{
    if (!rd) cd_encode_ue_bypass(cb, e, v);
    else cb.f8 += (cd_ue_size((unsigned)(v + (1 << e) - 1)) - e) << 8;              // rdo.c:57
}
// cabac_size_unary / cabac_transition_unary (x264_rdo_init, rdo.c:326-345) evaluated on the fly: prefix - 1 ones, a zero unless
// the prefix is 14, and the bypass sign
template <class ST> CD_FN int cd_unary(ST st, int ctx, int prefix)
{
    int bits = 0, s = st[ctx];
    for (int i = 1; i < prefix; i++) { bits += CD_ENT(s, 1); s = CD_TRANS(s, 1); }
    if (prefix > 0 && prefix < 14) { bits += CD_ENT(s, 0); s = CD_TRANS(s, 0); }
    st[ctx] = (u8)s;
    return bits + 256;
}

CD_FN int cd_median(int a, int b, int c) { const int mx = a > b ? a : b, mn = a < b ? a : b; return c > mx ? mx : c < mn ? mn : c; }
// x264_mb_predict_mv (R/common/macroblock.c:28-88) on the syntax record's motion cache
template <class MS> CD_FN void cd_predict_mv(const MS &m, int list, int idx, int width, int &px, int &py)
{
    const int i8 = cd_scan8(idx), i_ref = CD_CREF(m, list)[i8];
    int ra = CD_CREF(m, list)[i8 - 1], rb = CD_CREF(m, list)[i8 - 8], kc = i8 - 8 + width, rc = CD_CREF(m, list)[kc];
    if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { kc = i8 - 8 - 1; rc = CD_CREF(m, list)[kc]; }
    const int ax = CD_CMV(m, list)[i8 - 1][0], ay = CD_CMV(m, list)[i8 - 1][1], bx = CD_CMV(m, list)[i8 - 8][0], by = CD_CMV(m, list)[i8 - 8][1],
              cx = CD_CMV(m, list)[kc][0], cy = CD_CMV(m, list)[kc][1];
    if (m.partition == CD_D_16x8) {
        if (idx == 0 && rb == i_ref) { px = bx; py = by; return; }
        if (idx != 0 && ra == i_ref) { px = ax; py = ay; return; }
    } else if (m.partition == CD_D_8x16) {
        if (idx == 0 && ra == i_ref) { px = ax; py = ay; return; }
        if (idx != 0 && rc == i_ref) { px = cx; py = cy; return; }
    }
    const int cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
    if (cnt > 1) { px = cd_median(ax, bx, cx); py = cd_median(ay, by, cy); }
    else if (cnt == 1) { if (ra == i_ref) { px = ax; py = ay; } else if (rb == i_ref) { px = bx; py = by; } else { px = cx; py = cy; } }
    else if (rb == -2 && rc == -2 && ra != -2) { px = ax; py = ay; }
    else { px = cd_median(ax, bx, cx); py = cd_median(ay, by, cy); }
}
CD_FN int cd_fix4(int m) { return m < 0 ? -1 : m < 9 ? m : 2; }          // x264_mb_pred_mode4x4_fix
template <class MS> CD_FN int cd_pred_i4mode(const MS &m, int idx)
{   // x264_mb_predict_intra4x4_mode, R/common/macroblock.h:423-434
    const int ma = cd_fix4(m.i4c[cd_scan8(idx) - 1]), mb = cd_fix4(m.i4c[cd_scan8(idx) - 8]), v = ma < mb ? ma : mb;
    return v < 0 ? 2 : v;
}

template <class ST, class MS> CD_FN void cw_mb_type_intra(DCabac &cb, ST st, int rd, const MS &m, int c0, int c1, int c2, int c3, int c4, int c5, int i_frame)
{   // x264_cabac_mb_type_intra, R/encoder/cabac.c:32-62
    if (m.type == CD_I_4x4 || m.type == CD_I_8x8) cdd_noup(cb, st, rd, c0, 0);
    else if (m.type == CD_I_PCM) { cdd_noup(cb, st, rd, c0, 1); if (!rd) cd_encode_flush(cb, i_frame); }
    else {
        const int pred = m.i16mode < 4 ? m.i16mode : 2;          // x264_mb_pred_mode16x16_fix
        cdd_noup(cb, st, rd, c0, 1);
        if (!rd) cd_encode_terminal(cb); else cb.f8 += CD_ENT(st[276], 0);
        cdd_noup(cb, st, rd, c1, !!m.cbp_luma);
        if (m.cbp_chroma == 0) cdd_noup(cb, st, rd, c2, 0);
        else { cdd(cb, st, rd, c2, 1); cdd_noup(cb, st, rd, c3, m.cbp_chroma != 1); }
        cdd(cb, st, rd, c4, pred >> 1);
        cdd_noup(cb, st, rd, c5, pred & 1);
    }
}
template <class ST, class MS> CD_FN void cw_mb_type(DCabac &cb, ST st, int rd, const MS &m, int i_frame)
{   // x264_cabac_mb_type, :64-196
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
    } else if (m.type == CD_P_L0) {
        cdd_noup(cb, st, rd, 14, 0);
        if (m.partition == CD_D_16x16) { cdd_noup(cb, st, rd, 15, 0); cdd_noup(cb, st, rd, 16, 0); }
        else { cdd_noup(cb, st, rd, 15, 1); cdd_noup(cb, st, rd, 17, m.partition == CD_D_16x8); }
    } else if (m.type == CD_P_8x8) {
        cdd_noup(cb, st, rd, 14, 0); cdd_noup(cb, st, rd, 15, 0); cdd_noup(cb, st, rd, 16, 1);
    } else {
        cdd_noup(cb, st, rd, 14, 1);
        cw_mb_type_intra(cb, st, rd, m, 17 + 0, 17 + 1, 17 + 2, 17 + 2, 17 + 3, 17 + 3, i_frame);
    }
}
template <class ST> CD_FN void cw_intra4x4_pred_mode(DCabac &cb, ST st, int rd, int pred, int mode)
{   // :198-211
    if (pred == mode) { cdd(cb, st, rd, 68, 1); return; }
    cdd(cb, st, rd, 68, 0);
    if (mode > pred) mode--;
    cdd(cb, st, rd, 69, mode & 1); cdd(cb, st, rd, 69, (mode >> 1) & 1); cdd(cb, st, rd, 69, (mode >> 2) & 1);
}
template <class ST, class MS> CD_FN void cw_chroma_pred_mode(DCabac &cb, ST st, int rd, const MS &m)
{   // :213-231
    const int mode = m.chroma_mode < 4 ? m.chroma_mode : 0, ctx = (m.cpm_left != 0) + (m.cpm_top != 0);     // x264_mb_pred_mode8x8c_fix
    cdd_noup(cb, st, rd, 64 + ctx, mode > 0);
    if (mode > 0) {
        cdd(cb, st, rd, 64 + 3, mode > 1);
        if (mode > 1) cdd_noup(cb, st, rd, 64 + 3, mode > 2);
    }
}
template <class ST, class MS> CD_FN void cw_cbp_luma(DCabac &cb, ST st, int rd, const MS &m)
{   // x264_cabac_mb_cbp_luma, :233-242
    const int cbp = m.cbp_luma, l = m.cbp_left, t = m.cbp_top;
    cdd(cb, st, rd, 76 - ((l >> 1) & 1) - ((t >> 1) & 2), cbp & 1);
    cdd(cb, st, rd, 76 - ((cbp >> 0) & 1) - ((t >> 2) & 2), (cbp >> 1) & 1);
    cdd(cb, st, rd, 76 - ((l >> 3) & 1) - ((cbp << 1) & 2), (cbp >> 2) & 1);
    cdd_noup(cb, st, rd, 76 - ((cbp >> 2) & 1) - ((cbp >> 0) & 2), (cbp >> 3) & 1);
}
template <class ST, class MS> CD_FN void cw_cbp_chroma(DCabac &cb, ST st, int rd, const MS &m);
template <class ST, class MS> CD_FN void cw_cbp(DCabac &cb, ST st, int rd, const MS &m)
{   // x264_cabac_mb_cbp_luma + _chroma, :233-263
    cw_cbp_luma(cb, st, rd, m);
    cw_cbp_chroma(cb, st, rd, m);
}
template <class ST, class MS> CD_FN void cw_cbp_chroma(DCabac &cb, ST st, int rd, const MS &m)
{   // x264_cabac_mb_cbp_chroma, :244-263
    const int l = m.cbp_left, t = m.cbp_top;
    const int a = l & 0x30, b = t & 0x30;
    int ctx = 0;
    if (a && l != -1) ctx++;
    if (b && t != -1) ctx += 2;
    if (m.cbp_chroma == 0) { cdd_noup(cb, st, rd, 77 + ctx, 0); return; }
    cdd_noup(cb, st, rd, 77 + ctx, 1);
    ctx = 4 + (a == 0x20) + 2 * (b == 0x20);
    cdd_noup(cb, st, rd, 77 + ctx, m.cbp_chroma > 1);
}
template <class ST, class MS> CD_FN void cw_qp_delta(DCabac &cb, ST st, int rd, MS &m)
{   // :265-297
    int dqp = m.qp - m.last_qp, ctx;
    if (m.type == CD_I_16x16 && !(m.cbp_luma | m.cbp_chroma | m.nnz[24] | m.nnz[25] | m.nnz[26])) {
        if (!rd) m.qp = m.last_qp;
        dqp = 0;
    }
    ctx = m.last_dqp && m.prev_coded;
    if (dqp) {
        int val = dqp <= 0 ? -2 * dqp : 2 * dqp - 1;
        if (val >= 51 && val != 52) val = 103 - val;
        while (val--) { cdd(cb, st, rd, 60 + ctx, 1); ctx = 2 + (ctx >> 1); }
    }
    cdd_noup(cb, st, rd, 60 + ctx, 0);
}
// x264_cabac_mb_skip, :300-306
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
}
template <class ST> CD_FN void cw_sub_p_partition(DCabac &cb, ST st, int rd, int sub)
{   // :309-330
    if (sub == CD_D_L0_8x8) { cdd(cb, st, rd, 21, 1); return; }
    cdd(cb, st, rd, 21, 0);
    if (sub == CD_D_L0_8x4) { cdd(cb, st, rd, 22, 0); return; }
    cdd(cb, st, rd, 22, 1);
    cdd(cb, st, rd, 23, sub == CD_D_L0_4x8);
}
template <class ST, class MS> CD_FN void cw_ref(DCabac &cb, ST st, int rd, const MS &m, int idx, int list = 0)
{   // x264_cabac_mb_ref, :375-395 (h->mb.cache.skip is all zero in a P slice)
    const int i8 = cd_scan8(idx), refa = CD_CREF(m, list)[i8 - 1], refb = CD_CREF(m, list)[i8 - 8];
    int ref = CD_CREF(m, list)[i8], ctx = (refa > 0) + 2 * (refb > 0);
    if (m.slice_type == 1) ctx = (refa > 0 && !m.cskip[i8 - 1]) + 2 * (refb > 0 && !m.cskip[i8 - 8]);
    for (; ref > 0; ref--) { cdd(cb, st, rd, 54 + ctx, 1); ctx = (ctx >> 2) + 4; }
    cdd(cb, st, rd, 54 + ctx, 0);
}
template <class ST, class MS> CD_FN void cw_mvd_cpn(DCabac &cb, ST st, int rd, const MS &m, int list, int idx, int l, int mvd)
{   // x264_cabac_mb_mvd_cpn, :397-445
    const int i8 = cd_scan8(idx), amvd = cd_abs(CD_CMVD(m, list)[i8 - 1][l]) + cd_abs(CD_CMVD(m, list)[i8 - 8][l]), a = cd_abs(mvd), base = l ? 47 : 40;
    const int ctx = (amvd > 2) + (amvd > 32);
#define CD_MVCTX(i_) ((i_) < 4 ? (i_) + 2 : 6)                             /* ctxes[] = {0,3,4,5,6,6,6,6,6} for i >= 1 */
    if (a == 0) { cdd(cb, st, rd, base + ctx, 0); return; }
    cdd(cb, st, rd, base + ctx, 1);
    if (a < 9) {
        if (rd && a > 4) {
            for (int i = 1; i < 4; i++) cdd(cb, st, rd, base + CD_MVCTX(i), 1);
            cb.f8 += cd_unary(st, base + 6, a - 3);
        } else {
            for (int i = 1; i < a; i++) cdd(cb, st, rd, base + CD_MVCTX(i), 1);
            cdd(cb, st, rd, base + CD_MVCTX(a), 0);
            cdb(cb, rd, mvd < 0);
        }
    } else if (rd) {
        for (int i = 1; i < 4; i++) cdd(cb, st, rd, base + CD_MVCTX(i), 1);
        for (int i = 0; i < 5; i++) cdd(cb, st, rd, base + 6, 1);               // cabac_size_5ones
        cb.f8 += 256;
        cd_ue(cb, rd, 3, a - 9);
    } else {
        for (int i = 1; i < 9; i++) cdd(cb, st, rd, base + CD_MVCTX(i), 1);
        cd_ue(cb, rd, 3, a - 9);
        cdb(cb, rd, mvd < 0);
    }
#undef CD_MVCTX
}
template <class ST, class MS> CD_FN void cw_mvd(DCabac &cb, ST st, int rd, MS &m, int idx, int width, int height, int list = 0)
{   // x264_cabac_mb_mvd, :447-463
    int px, py;
    const int i8 = cd_scan8(idx);
    cd_predict_mv(m, list, idx, width, px, py);
    const int dx = CD_CMV(m, list)[i8][0] - px, dy = CD_CMV(m, list)[i8][1] - py;
    cw_mvd_cpn(cb, st, rd, m, list, idx, 0, dx);
    cw_mvd_cpn(cb, st, rd, m, list, idx, 1, dy);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) { CD_CMVD(m, list)[i8 + x + 8 * y][0] = (i16)dx; CD_CMVD(m, list)[i8 + x + 8 * y][1] = (i16)dy; }
}
template <class ST, class MS> CD_FN void cw_mb8x8_mvd(DCabac &cb, ST st, int rd, MS &m, int i)
{   // :465-498 (list 0)
    const int sub = m.sub[i];
    if (sub == CD_D_L0_8x8) cw_mvd(cb, st, rd, m, 4 * i, 2, 2);
    else if (sub == CD_D_L0_8x4) { cw_mvd(cb, st, rd, m, 4 * i, 2, 1); cw_mvd(cb, st, rd, m, 4 * i + 2, 2, 1); }
    else if (sub == CD_D_L0_4x8) { cw_mvd(cb, st, rd, m, 4 * i, 1, 2); cw_mvd(cb, st, rd, m, 4 * i + 1, 1, 2); }
    else for (int k = 0; k < 4; k++) cw_mvd(cb, st, rd, m, 4 * i + k, 1, 1);
}

template <class MS> CD_FN int cd_nz_left(const MS &m, int idx)
{
    if (idx < 16) return ((idx & 1) | (idx & 4)) ? m.nnz[idx - ((idx & 1) ? 1 : 3)] : m.nz_l[((idx >> 1) & 1) | ((idx >> 2) & 2)];
    const int k = (idx - 16) & 3, ch = (idx - 16) >> 2;
    return (k & 1) ? m.nnz[idx - 1] : m.nz_lc[ch][k >> 1];
}
template <class MS> CD_FN int cd_nz_top(const MS &m, int idx)
{
    if (idx < 16) return ((idx & 2) | (idx & 8)) ? m.nnz[idx - ((idx & 2) ? 2 : 6)] : m.nz_t[(idx & 1) | ((idx >> 1) & 2)];
    const int k = (idx - 16) & 3, ch = (idx - 16) >> 2;
    return (k & 2) ? m.nnz[idx - 2] : m.nz_tc[ch][k & 1];
}
template <class MS> CD_FN int cw_cbf_ctx(const MS &m, int cat, int idx)
{   // x264_cabac_mb_cbf_ctxidxinc, :508-538
    const int intra = m.type <= CD_I_PCM;
    int a, b;
    if (cat == 1 || cat == 2 || cat == 4) {
        a = cd_nz_left(m, idx) & (0x7f + (intra << 7)); b = cd_nz_top(m, idx) & (0x7f + (intra << 7));
        return 4 * cat + 2 * !!b + !!a;
    }
    if (cat == 0) return 4 * cat + 2 * ((m.cbp_top >> 8) & 1) + ((m.cbp_left >> 8) & 1);
    idx -= 25;
    a = m.cbp_left != -1 ? (m.cbp_left >> (9 + idx)) & 1 : intra;
    b = m.cbp_top != -1 ? (m.cbp_top >> (9 + idx)) & 1 : intra;
    return 4 * cat + 2 * b + a;
}

// context offsets by block category (0 luma DC, 1 luma AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8), frame macroblocks
#define CD_SIG_OFF(c_) ((c_) == 0 ? 105 : (c_) == 1 ? 120 : (c_) == 2 ? 134 : (c_) == 3 ? 149 : (c_) == 4 ? 152 : 402)
#define CD_LAST_OFF(c_) ((c_) == 0 ? 166 : (c_) == 1 ? 181 : (c_) == 2 ? 195 : (c_) == 3 ? 210 : (c_) == 4 ? 213 : 417)
#define CD_LEVEL_OFF(c_) ((c_) == 0 ? 227 : (c_) == 1 ? 237 : (c_) == 2 ? 247 : (c_) == 3 ? 257 : (c_) == 4 ? 266 : 426)
static __device__ const u8 d_cw_sig8[63] = {     // significant_coeff_flag_offset_8x8[0], ITU-T H.264 table 9-43 (frame)
    0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7,
    7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};
static __device__ const u8 d_cw_last8[63] = {
    0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2,
    3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
#ifndef X264HIP_HOST_TEST
// all 64 lanes, once per kernel (followed by a barrier of the caller's)
CD_FN void cd_load_tables(int lane)
{
    for (int k = lane; k < 256; k += 64) cd_lds_et[k] = (u32)d_cabac_entropy[k >> 1][k & 1] | ((u32)d_cabac_transition[k >> 1][k & 1] << 16);
    for (int k = lane; k < 128; k += 64)
        cd_lds_lps[k] = (u32)d_cabac_range_lps[k][0] | ((u32)d_cabac_range_lps[k][1] << 8) | ((u32)d_cabac_range_lps[k][2] << 16) | ((u32)d_cabac_range_lps[k][3] << 24);
    if (lane < 63) { cd_lds_sig8[lane] = d_cw_sig8[lane]; cd_lds_last8[lane] = d_cw_last8[lane]; }
}
#endif
// node -> context of "level is 1" / "level > 1" and the node after either (cabac.c:570-581), as nibble lists
#define CD_LVL1_CTX(n_) ((int)((0x00004321u >> (4 * (n_))) & 15))
#define CD_LVLGT1_CTX(n_) ((int)((0x98765555u >> (4 * (n_))) & 15))
#define CD_NODE_NEXT0(n_) ((int)((0x76543321u >> (4 * (n_))) & 15))
#define CD_NODE_NEXT1(n_) ((int)((0x77654444u >> (4 * (n_))) & 15))

template <class LV> CD_FN int cd_coeff_last(LV l, int n) { int i = n - 1; while (i >= 0 && !l[i]) i--; return i; }

// block_residual_write_cabac: the writer's order (:584-674) when !rd, the RD order (:683-763) when rd
template <class ST, class MS, class LV> CD_FN void cw_residual(DCabac &cb, ST st, int rd, const MS &m, int cat, int idx, LV l, int count)
{
    const int c_sig = CD_SIG_OFF(cat), c_last = CD_LAST_OFF(cat), c_lvl = CD_LEVEL_OFF(cat), b8 = cat == 5;
    if (!b8) {
        const int ctx = 85 + cw_cbf_ctx(m, cat, idx);
        if (!m.nnz[idx]) { cdd(cb, st, rd, ctx, 0); return; }
        cdd(cb, st, rd, ctx, 1);
    }
    const int last = cd_coeff_last(l, count);
    int node = 0;
    if (!rd) {
        // significance map forwards, then the levels backwards (they are re-read from l instead of being parked in an array)
        const int sigmap = last + 1 < count - 1 ? last + 1 : count - 1;
        for (int i = 0; i < sigmap; i++) {
            const int cs = c_sig + (b8 ? CD_SIG8(i) : i), cl = c_last + (b8 ? CD_LAST8(i) : i);
            if (l[i]) { cd_encode_decision(cb, st, cs, 1); cd_encode_decision(cb, st, cl, i == last); }
            else cd_encode_decision(cb, st, cs, 0);
        }
        for (int i = last; i >= 0; i--) {
            const int v = l[i];
            if (!v) continue;
            const int am1 = cd_abs(v) - 1, prefix = am1 < 14 ? am1 : 14;
            int ctx = CD_LVL1_CTX(node) + c_lvl;
            if (prefix) {
                cd_encode_decision(cb, st, ctx, 1);
                ctx = CD_LVLGT1_CTX(node) + c_lvl;
                for (int k = 0; k < prefix - 1; k++) cd_encode_decision(cb, st, ctx, 1);
                if (prefix < 14) cd_encode_decision(cb, st, ctx, 0);
                else cd_encode_ue_bypass(cb, 0, am1 - 14);
                node = CD_NODE_NEXT1(node);
            } else {
                cd_encode_decision(cb, st, ctx, 0);
                node = CD_NODE_NEXT0(node);
            }
            cd_encode_bypass(cb, v < 0);
        }
        return;
    }
    for (int i = last; i >= 0; i--) {
        const int cs = c_sig + (b8 ? CD_SIG8(i < 63 ? i : 62) : i), cl = c_last + (b8 ? CD_LAST8(i < 63 ? i : 62) : i);
        const int v = l[i];
        if (i == last) {
            if (last != count - 1) { cdd(cb, st, 1, cs, 1); cdd(cb, st, 1, cl, 1); }
        } else if (v) { cdd(cb, st, 1, cs, 1); cdd(cb, st, 1, cl, 0); }
        else { cdd(cb, st, 1, cs, 0); continue; }
        const int am1 = cd_abs(v) - 1, prefix = am1 < 14 ? am1 : 14;
        int ctx = CD_LVL1_CTX(node) + c_lvl;
        if (prefix) {
            cdd(cb, st, 1, ctx, 1);
            ctx = CD_LVLGT1_CTX(node) + c_lvl;
            cb.f8 += cd_unary(st, ctx, prefix);
            if (prefix >= 14) cd_ue(cb, 1, 0, am1 - 14);
            node = CD_NODE_NEXT1(node);
        } else {
            cdd(cb, st, 1, ctx, 0);
            node = CD_NODE_NEXT0(node);
            cb.f8 += 256;
        }
    }
}

// x264_macroblock_write_cabac, :781-1022 (I and P slices): writes (rd = 0) or counts (rd = 1: x264_macroblock_size_cabac).
// fe: the source macroblock (Y 16x16, U 8x8, V 8x8 contiguous) for I_PCM, which is only ever written.  The caller copies the
// source into the reconstruction for I_PCM (cabac.c:815-818).
template <class ST, class MS, class FE> CD_FN void cw_macroblock(DCabac &cb, ST st, int rd, MS &m, FE fe, int i_frame)
{
    const int type = m.type;
    cw_mb_type(cb, st, rd, m, i_frame);
    if (!rd && type == CD_I_PCM) {
        for (int i = 0; i < 384; i++) *cb.p++ = fe[i];
        cb.low = 0; cb.range = 0x1FE; cb.queue = -1; cb.outstanding = 0;
        return;
    }
    if (type <= CD_I_PCM) {
        if (m.pps_t8 && type != CD_I_16x16) cdd_noup(cb, st, rd, 399 + m.nb_t8, m.t8);
        if (type != CD_I_16x16)
            for (int i = 0; i < 16; i += type == CD_I_8x8 ? 4 : 1)
                cw_intra4x4_pred_mode(cb, st, rd, cd_pred_i4mode(m, i), cd_fix4(m.i4c[cd_scan8(i)]));
        cw_chroma_pred_mode(cb, st, rd, m);
    } else if (type == CD_P_L0) {
        const int multi = m.n_ref > 1;
        if (m.partition == CD_D_16x16) {
            if (multi) cw_ref(cb, st, rd, m, 0);
            cw_mvd(cb, st, rd, m, 0, 4, 4);
        } else if (m.partition == CD_D_16x8) {
            if (multi) { cw_ref(cb, st, rd, m, 0); cw_ref(cb, st, rd, m, 8); }
            cw_mvd(cb, st, rd, m, 0, 4, 2); cw_mvd(cb, st, rd, m, 8, 4, 2);
        } else {
            if (multi) { cw_ref(cb, st, rd, m, 0); cw_ref(cb, st, rd, m, 4); }
            cw_mvd(cb, st, rd, m, 0, 2, 4); cw_mvd(cb, st, rd, m, 4, 2, 4);
        }
    } else if (type == CD_P_8x8) {
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
    }
    if (type != CD_I_16x16) cw_cbp(cb, st, rd, m);
    if (m.t8_allowed && m.cbp_luma) cdd_noup(cb, st, rd, 399 + m.nb_t8, m.t8);
    if (m.cbp_luma > 0 || m.cbp_chroma > 0 || type == CD_I_16x16) {
        cw_qp_delta(cb, st, rd, m);
        if (type == CD_I_16x16) {
            cw_residual(cb, st, rd, m, 0, 24, &m.lv_dc[0], 16);
            if (m.cbp_luma) for (int i = 0; i < 16; i++) cw_residual(cb, st, rd, m, 1, i, &m.lv4[i][1], 15);
        } else if (m.t8) {
            for (int i = 0; i < 4; i++) if (m.cbp_luma & (1 << i)) cw_residual(cb, st, rd, m, 5, 4 * i, &m.lv8[i][0], 64);
        } else
            for (int i = 0; i < 16; i++) if (m.cbp_luma & (1 << (i >> 2))) cw_residual(cb, st, rd, m, 2, i, &m.lv4[i][0], 16);
        if (m.cbp_chroma & 3) { cw_residual(cb, st, rd, m, 3, 25, &m.lv_cdc[0][0], 4); cw_residual(cb, st, rd, m, 3, 26, &m.lv_cdc[1][0], 4); }
        if (m.cbp_chroma & 2) for (int i = 16; i < 24; i++) cw_residual(cb, st, rd, m, 4, i, &m.lv_cac[i - 16][1], 15);
    }
}

// ---- the RD-only partial writers, R/encoder/cabac.c:1024-1131 (RDO_SKIP_BS build): they only count, on a copy of the contexts.
// "doesn't write cbp or chroma dc, doesn't write ref or subpartition".  pix: 1 16x8, 2 8x16, 3 8x8 (4 8x4, 5 4x8, 6 4x4 for sub-partitions)
template <class ST, class MS> CD_FN void cw_partition_size(DCabac &cb, ST st, MS &m, int i8, int pix)
{   // x264_partition_size_cabac, :1032-1081
    const int b_8x16 = m.partition == CD_D_8x16;
    if (m.type == CD_P_8x8) cw_mb8x8_mvd(cb, st, 1, m, i8);
    else if (m.type == CD_P_L0) cw_mvd(cb, st, 1, m, 4 * i8, 4 >> b_8x16, 2 << b_8x16);
    else if (m.type > CD_B_DIRECT && m.type < CD_B_8x8) {
        if (CD_B_USES(m.type, 0, !!i8)) cw_mvd(cb, st, 1, m, 4 * i8, 4 >> b_8x16, 2 << b_8x16, 0);
        if (CD_B_USES(m.type, 1, !!i8)) cw_mvd(cb, st, 1, m, 4 * i8, 4 >> b_8x16, 2 << b_8x16, 1);
    } else if (m.type == CD_B_8x8) {
        for (int l = 0; l < 2; l++) if (CD_SUB_USES(m.sub[i8], l)) cw_mvd(cb, st, 1, m, 4 * i8, 2, 2, l);
    } else
        return;
    for (int j = pix < 3; j >= 0; j--) {
        if (m.cbp_luma & (1 << i8)) {
            if (m.t8) cw_residual(cb, st, 1, m, 5, 4 * i8, &m.lv8[i8][0], 64);
            else for (int i4 = 0; i4 < 4; i4++) cw_residual(cb, st, 1, m, 2, i4 + 4 * i8, &m.lv4[i4 + 4 * i8][0], 16);
        }
        cw_residual(cb, st, 1, m, 4, 16 + i8, &m.lv_cac[i8][1], 15);
        cw_residual(cb, st, 1, m, 4, 20 + i8, &m.lv_cac[4 + i8][1], 15);
        i8 += pix == 2 ? 2 : 1;                       // x264_pixel_size[pix].h >> 3
    }
}
template <class ST, class MS> CD_FN void cw_subpartition_size(DCabac &cb, ST st, MS &m, int i4, int pix)
{   // x264_subpartition_size_cabac, :1083-1095
    const int b_8x4 = pix == 4;
    cw_residual(cb, st, 1, m, 2, i4, &m.lv4[i4][0], 16);
    if (pix == 6) cw_mvd(cb, st, 1, m, i4, 1, 1);
    else {
        cw_mvd(cb, st, 1, m, i4, 1 + b_8x4, 2 - b_8x4);
        cw_residual(cb, st, 1, m, 2, i4 + 2 - b_8x4, &m.lv4[i4 + 2 - b_8x4][0], 16);
    }
}
template <class ST, class MS> CD_FN void cw_partition_i8x8_size(DCabac &cb, ST st, const MS &m, int i8, int mode)
{   // x264_partition_i8x8_size_cabac, :1097-1105
    cw_intra4x4_pred_mode(cb, st, 1, cd_pred_i4mode(m, 4 * i8), cd_fix4(mode));
    cw_cbp_luma(cb, st, 1, m);
    if (m.cbp_luma & (1 << i8)) cw_residual(cb, st, 1, m, 5, 4 * i8, &m.lv8[i8][0], 64);
}
template <class ST, class MS> CD_FN void cw_partition_i4x4_size(DCabac &cb, ST st, const MS &m, int i4, int mode)
{   // x264_partition_i4x4_size_cabac, :1107-1113
    cw_intra4x4_pred_mode(cb, st, 1, cd_pred_i4mode(m, i4), cd_fix4(mode));
    cw_residual(cb, st, 1, m, 2, i4, &m.lv4[i4][0], 16);
}
template <class ST, class MS> CD_FN void cw_i8x8_chroma_size(DCabac &cb, ST st, const MS &m)
{   // x264_i8x8_chroma_size_cabac, :1115-1131
    cw_chroma_pred_mode(cb, st, 1, m);
    cw_cbp_chroma(cb, st, 1, m);
    if (m.cbp_chroma > 0) {
        cw_residual(cb, st, 1, m, 3, 25, &m.lv_cdc[0][0], 4); cw_residual(cb, st, 1, m, 3, 26, &m.lv_cdc[1][0], 4);
        if (m.cbp_chroma == 2) for (int i = 16; i < 24; i++) cw_residual(cb, st, 1, m, 4, i, &m.lv_cac[i - 16][1], 15);
    }
}
