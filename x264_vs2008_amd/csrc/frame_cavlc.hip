// frame_cavlc.hip -- FRAME LEVEL, part 7: the CAVLC writer (x264_macroblock_write_cavlc with block_residual_write_cavlc and the skip
// runs of x264_slice_write, R/encoder/cavlc.c:60-620, R/encoder/encoder.c:1200-1280) for I and P slices, as a pass over the state a sweep
// left: macroblock types, partitions, references, vectors, prediction modes, cbp and the coefficient levels.
//
// CAVLC has no adaptive state, so nothing forces it into the macroblock loop: the RD levels do not run with it (R/encoder/rdo.c prices
// CAVLC bits with a counting twin of this writer; refused, as before) and a `--no-cabac` slice is the wavefront variant's -- constant QP,
// decisions and levels in the x264hip_mb_state.  What the writer needs beyond the state it derives as the reference's cache does:
// the predicted vector of every partition (x264_mb_predict_mv on a scan8-shaped cache of the neighbours' vectors and references), the
// predicted intra mode (min of the left and top block's, DC when either is unavailable), and nC from the left and top block's
// coefficient counts -- the TOTALS block_residual_write_cavlc stores back (an encode leaves only non-zero flags there), kept here for
// the row above and the macroblock to the left.
// Mapping: one wavefront per chain, lane 0 writes (a slice's bit string is serial; the chains are the parallelism, and the pass costs a
// few percent of the sweep it follows).  Payload layout as the CABAC writer's: X264HIP_PAYLOAD_LEAD bytes of each chain's slot, then
// slice_data() from bit 0, rbsp trailing bits included.
#include "device_prims.h"
#include "frame_internal.h"
#include "x264hip_lookahead.h"
#include "cavlc_tables.h"

using namespace x264hip;

#define CV_MAX_W 512
enum { CV_I_4x4 = 0, CV_I_8x8 = 1, CV_I_16x16 = 2, CV_I_PCM = 3, CV_P_L0 = 4, CV_P_8x8 = 5, CV_P_SKIP = 6 };      // R/common/macroblock.h:78-102
enum { CV_D_L0_4x4 = 0, CV_D_L0_8x4 = 1, CV_D_L0_4x8 = 2, CV_D_L0_8x8 = 3, CV_D_8x8 = 13, CV_D_16x8 = 14, CV_D_8x16 = 15, CV_D_16x16 = 16 };   // :55-76

struct CvBs { u8 *p; unsigned long long acc; int n; };                 // bits not yet stored, MSB first
__device__ __forceinline__ void cv_put(CvBs &b, int n, u32 v)
{
    b.acc = (b.acc << n) | (v & ((n >= 32) ? 0xffffffffu : ((1u << n) - 1u)));
    b.n += n;
    while (b.n >= 8) { *b.p++ = (u8)(b.acc >> (b.n - 8)); b.n -= 8; }
}
__device__ __forceinline__ void cv_ue(CvBs &b, u32 v)
{   // bs_write_ue_big: Exp-Golomb
    v += 1;
    const int len = 32 - __clz((int)v);
    if (len > 1) cv_put(b, len - 1, 0);
    cv_put(b, len, v);
}
__device__ __forceinline__ void cv_se(CvBs &b, int v) { cv_ue(b, v <= 0 ? (u32)(-2 * v) : (u32)(2 * v - 1)); }
__device__ __forceinline__ void cv_te(CvBs &b, int x, int v) { if (x == 1) cv_put(b, 1, v ^ 1); else cv_ue(b, (u32)v); }
__device__ __forceinline__ void cv_vlc(CvBs &b, unsigned short e) { cv_put(b, e & 0xff, (u32)(e >> 8)); }

// x264_scan8 for the 24 blocks (R/common/macroblock.h:204-230)
__device__ __forceinline__ int cv_scan8(int i)
{
    if (i < 16) { const int x = (i & 1) + ((i >> 2) & 1) * 2, y = ((i >> 1) & 1) + (i >> 3) * 2; return 4 + 8 + x + 8 * y; }
    const int c = i - 16, ch = c >> 2, k = c & 3;
    return (ch ? 1 + 8 * 4 : 1 + 8 * 1) + (k & 1) + 8 * (k >> 1);
}

struct CvArgs {
    const signed char *mb_type, *partition, *sub_partition, *ref, *i4mode, *i16mode, *chroma_mode, *t8;
    const i16 *mv, *cbp, *luma, *luma_dc, *chroma_dc, *chroma_ac;
    const u8 *nnz;
    const signed char *qp; int slice_qp;
    u8 *payload; int payload_cap; int *payload_len, *mb_bits; int *abort_flag;
    int mb_w, mb_h, slice_type, n_ref0, psub8x8, t8_mode, profile_high;
};

// one residual block: block_residual_write_cavlc.  l: the block's coefficients in scan order (count of them), nC already predicted.
__device__ int cv_residual(CvBs &b, const i16 *l, int count, int nc_class, bool chroma_dc, int profile_high)
{
    int last = count - 1;
    while (last >= 0 && l[last] == 0) last--;
    if (last < 0) { cv_vlc(b, c_cv_coeff0[nc_class]); return 0; }
    __shared__ int level[16], run[16];                                       // lane 0 is the only one here: its work arrays live in LDS, not scratch
    int total = 0, i_last = last;
    do {
        int r = 0;
        level[total] = l[i_last];
        while (--i_last >= 0 && l[i_last] == 0) r++;
        run[total++] = r;
    } while (i_last >= 0);
    int total_zero = last + 1 - total;
    int trailing = 0;
    while (trailing < 3 && trailing < total && (level[trailing] == 1 || level[trailing] == -1)) trailing++;
    u32 sign = 0;
    for (int i = 0; i < trailing; i++) sign = (sign << 1) | (level[i] < 0);
    cv_vlc(b, c_cv_coeff[nc_class * 64 + total * 4 + trailing - 4]);
    int suffix = total > 10 && trailing < 3;
    if (trailing > 0) cv_put(b, trailing, sign);
    for (int i = trailing; i < total; i++) {
        int val = level[i];
        if (i == trailing && trailing < 3) val -= (val >> 31) | 1;        // the first level after fewer than three trailing ones cannot be +-1
        // x264_level_token[suffix][val] (R/common/vlc.c:874-915) / block_residual_write_cavlc_escape beyond the table
        const int orig = level[i];
        const int mask = val >> 31, abs_level = (val ^ mask) - mask;
        int code = abs_level * 2 - mask - 2;
        const bool in_table = (unsigned)(orig + 64) < 128u && (unsigned)(val + 64) < 128u;
        if (in_table) {
            if ((code >> suffix) < 14) cv_put(b, (code >> suffix) + 1 + suffix, (1u << suffix) + (code & ((1 << suffix) - 1)));
            else if (suffix == 0 && code < 30) cv_put(b, 19, (1u << 4) + (code - 14));
            else if (suffix > 0 && (code >> suffix) == 14) cv_put(b, 15 + suffix, (1u << suffix) + (code & ((1 << suffix) - 1)));
            else { code -= 15 << suffix; if (suffix == 0) code -= 15; cv_put(b, 28, (1u << 12) + code); }
        } else {
            int prefix = 15;
            if ((code >> suffix) < 15) cv_put(b, (code >> suffix) + 1 + suffix, (1u << suffix) + (code & ((1 << suffix) - 1)));
            else {
                code -= 15 << suffix;
                if (suffix == 0) code -= 15;
                if (code >= 1 << 12) {
                    if (profile_high) while (code > 1 << (prefix - 3)) { code -= 1 << (prefix - 3); prefix++; }
                    else code = (1 << 12) - 2 + (code & 1);
                }
                cv_put(b, prefix + 1, 1);
                cv_put(b, prefix - 3, code & ((1 << (prefix - 3)) - 1));
            }
        }
        // i_next: by the ORIGINAL level (x264_level_token[..][val_original].i_next; the escape computes it from the adjusted one)
        const int a2 = in_table ? (orig < 0 ? -orig : orig) : abs_level;
        if (suffix == 0) suffix++;
        if (a2 > (3 << (suffix - 1)) && suffix < 6) suffix++;
    }
    if (total < count) cv_vlc(b, chroma_dc ? c_cv_total_zeros_dc[(total - 1) * 4 + total_zero] : c_cv_total_zeros[(total - 1) * 16 + total_zero]);
    for (int i = 0; i < total - 1 && total_zero > 0; i++) {
        const int zl = total_zero - 1 < 6 ? total_zero - 1 : 6;
        cv_vlc(b, c_cv_run_before[zl * 16 + run[i]]);
        total_zero -= run[i];
    }
    return total;
}

__global__ __launch_bounds__(64) void k_cavlc_write(CvArgs a)
{
    // the row above and the left macroblock: coefficient totals (4 luma + 2 Cb + 2 Cr per side), kept by this writer
    __shared__ u8 s_top_nnz[CV_MAX_W][8];
    __shared__ u8 s_left_nnz[8];
    if (threadIdx.x != 0) return;
    const int bz = blockIdx.x, n = a.mb_w * a.mb_h;
    const size_t cb = (size_t)n * bz;
    u8 *out = a.payload + (size_t)bz * a.payload_cap + 64;
    CvBs b = {out, 0ull, 0};
    const u8 *limit = out + a.payload_cap - 64 - 1024;
    int skip_run = 0, last_qp = a.slice_qp;                                  // h->mb.i_last_qp (x264_slice_write starts it at the slice's QP)
    const bool is_p = a.slice_type == 0;
    for (int mb = 0; mb < n; mb++) {
        const int mbx = mb % a.mb_w, mby = mb / a.mb_w;
        const size_t M = cb + mb;
        const int type = a.mb_type[M];
        u8 *tn = s_top_nnz[mbx];
        if (b.p > limit || type == CV_I_PCM || type > CV_P_SKIP) { atomicAdd(a.abort_flag, 1); a.payload_len[bz] = 0; return; }
        if (type == CV_P_SKIP) {
            skip_run++;
            last_qp = a.qp[M];                                                 // x264_macroblock_cache_save: every macroblock leaves its QP (a skipped one: the previous)
            for (int k = 0; k < 8; k++) { tn[k] = 0; s_left_nnz[k] = 0; }
            if (a.mb_bits) a.mb_bits[M] = (int)((b.p - out) * 8 + b.n);
            continue;
        }
        if (is_p) { cv_ue(b, (u32)skip_run); skip_run = 0; }
        const int off = is_p ? 5 : 0;
        const int cbp = a.cbp[M], cbp_luma = cbp & 15, cbp_chroma = (cbp >> 4) & 3, t8 = a.t8[M];
        const bool has_left = mbx > 0, has_top = mby > 0;
        // ---- type, prediction, vectors ----
        if (type == CV_I_4x4 || type == CV_I_8x8) {
            cv_ue(b, (u32)off);
            if (a.t8_mode) cv_put(b, 1, (u32)(type == CV_I_8x8));
            const signed char *mine = a.i4mode + M * 16;
            for (int i = 0; i < 16; i += (type == CV_I_8x8 ? 4 : 1)) {
                // x264_mb_predict_intra4x4_mode: the block to the left / above; a neighbour macroblock that is not I_4x4 / I_8x8 counts as DC, none as -1
                const int s8 = cv_scan8(i), x = (s8 & 7) - 4, y = (s8 >> 3) - 1;
                int ma, mbm;
                auto nb_mode = [&](size_t Mn, int bx, int by) -> int {
                    const int tn_ = a.mb_type[Mn];
                    if (tn_ != CV_I_4x4 && tn_ != CV_I_8x8) return 2;
                    const int bi = (bx & 1) + ((by & 1) << 1) + ((bx >> 1) << 2) + ((by >> 1) << 3);
                    return a.i4mode[Mn * 16 + bi];
                };
                if (x > 0) ma = mine[(x - 1 & 1) + ((y & 1) << 1) + (((x - 1) >> 1) << 2) + ((y >> 1) << 3)];
                else ma = has_left ? nb_mode(M - 1, 3, y) : -1;
                if (y > 0) mbm = mine[(x & 1) + (((y - 1) & 1) << 1) + ((x >> 1) << 2) + (((y - 1) >> 1) << 3)];
                else mbm = has_top ? nb_mode(M - a.mb_w, x, 3) : -1;
                const int fa = ma < 0 ? -1 : ma < 9 ? ma : 2, fb = mbm < 0 ? -1 : mbm < 9 ? mbm : 2;      // x264_mb_pred_mode4x4_fix
                int pred = fa < fb ? fa : fb;
                if (pred < 0) pred = 2;
                const int m0 = mine[i], mode = m0 < 9 ? m0 : 2;
                if (pred == mode) cv_put(b, 1, 1);
                else cv_put(b, 4, (u32)(mode - (mode > pred)));
            }
            const int cm = a.chroma_mode[M];
            cv_ue(b, (u32)(cm < 4 ? cm : 0));                                  // x264_mb_pred_mode8x8c_fix
        } else if (type == CV_I_16x16) {
            const int m16 = a.i16mode[M];
            cv_ue(b, (u32)(off + 1 + (m16 < 4 ? m16 : 2) + cbp_chroma * 4 + (cbp_luma == 0 ? 0 : 12)));
            const int cm = a.chroma_mode[M];
            cv_ue(b, (u32)(cm < 4 ? cm : 0));
        } else {
            // P_L0 / P_8x8: the motion cache of x264_macroblock_cache_load (scan8 layout, 5 rows x 8), then x264_mb_predict_mv per partition
            __shared__ signed char cref[40];
            __shared__ int cmvx[40], cmvy[40];
            for (int k = 0; k < 40; k++) { cref[k] = -2; cmvx[k] = cmvy[k] = 0; }
            auto load_nb = [&](size_t Mn, int pos, int bx, int by) {          // neighbour macroblock's 4x4 block (bx, by) -> cache position
                const int tn_ = a.mb_type[Mn];
                if (tn_ < CV_P_L0) { cref[pos] = -1; return; }                // intra: reference -1, vector 0
                cref[pos] = a.ref[Mn * 4 + (bx >> 1) + (by >> 1) * 2];
                cmvx[pos] = a.mv[(Mn * 16 + by * 4 + bx) * 2]; cmvy[pos] = a.mv[(Mn * 16 + by * 4 + bx) * 2 + 1];
            };
            if (has_top) for (int x = 0; x < 4; x++) load_nb(M - a.mb_w, 4 + x, x, 3);
            if (has_top && has_left) load_nb(M - a.mb_w - 1, 3, 3, 3);
            if (has_top && mbx < a.mb_w - 1) load_nb(M - a.mb_w + 1, 8, 0, 3);
            if (has_left) for (int y = 0; y < 4; y++) load_nb(M - 1, 11 + 8 * y, 3, y);
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) {
                    const int pos = 12 + x + 8 * y;
                    cref[pos] = a.ref[M * 4 + (x >> 1) + (y >> 1) * 2];
                    cmvx[pos] = a.mv[(M * 16 + y * 4 + x) * 2]; cmvy[pos] = a.mv[(M * 16 + y * 4 + x) * 2 + 1];
                }
            // what the decoder has not reached when it predicts: the positions right of blocks 5, 7 and 13 (R/common/macroblock.c:1050-1052)
            cref[cv_scan8(5) + 1] = cref[cv_scan8(7) + 1] = cref[cv_scan8(13) + 1] = -2;
            auto predict = [&](int idx, int width, int &px, int &py) {        // x264_mb_predict_mv, R/common/macroblock.c:31-118 (list 0)
                const int i8 = cv_scan8(idx), i_ref = cref[i8];
                int ra = cref[i8 - 1], ax = cmvx[i8 - 1], ay = cmvy[i8 - 1];
                int rb = cref[i8 - 8], bx_ = cmvx[i8 - 8], by_ = cmvy[i8 - 8];
                int rc = cref[i8 - 8 + width], cx = cmvx[i8 - 8 + width], cy = cmvy[i8 - 8 + width];
                if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { rc = cref[i8 - 8 - 1]; cx = cmvx[i8 - 8 - 1]; cy = cmvy[i8 - 8 - 1]; }
                const int part = a.partition[M];
                if (part == CV_D_16x8) {
                    if (idx == 0) { if (rb == i_ref) { px = bx_; py = by_; return; } }
                    else if (ra == i_ref) { px = ax; py = ay; return; }
                } else if (part == CV_D_8x16) {
                    if (idx == 0) { if (ra == i_ref) { px = ax; py = ay; return; } }
                    else if (rc == i_ref) { px = cx; py = cy; return; }
                }
                const int cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
                auto med = [](int p, int q, int r) { const int mx = p > q ? p : q, mn = p > q ? q : p; return r > mx ? mx : r < mn ? mn : r; };
                if (cnt > 1) { px = med(ax, bx_, cx); py = med(ay, by_, cy); }
                else if (cnt == 1) { if (ra == i_ref) { px = ax; py = ay; } else if (rb == i_ref) { px = bx_; py = by_; } else { px = cx; py = cy; } }
                else if (rb == -2 && rc == -2 && ra != -2) { px = ax; py = ay; }
                else { px = med(ax, bx_, cx); py = med(ay, by_, cy); }
            };
            auto mvd = [&](int idx, int width) {
                int px, py;
                predict(idx, width, px, py);
                const int i8 = cv_scan8(idx);
                cv_se(b, cmvx[i8] - px); cv_se(b, cmvy[i8] - py);
            };
            if (type == CV_P_L0) {
                const int part = a.partition[M];
                if (part == CV_D_16x16) {
                    cv_ue(b, 0);
                    if (a.n_ref0 > 1) cv_te(b, a.n_ref0 - 1, cref[cv_scan8(0)]);
                    mvd(0, 4);
                } else if (part == CV_D_16x8) {
                    cv_ue(b, 1);
                    if (a.n_ref0 > 1) { cv_te(b, a.n_ref0 - 1, cref[cv_scan8(0)]); cv_te(b, a.n_ref0 - 1, cref[cv_scan8(8)]); }
                    mvd(0, 4); mvd(8, 4);
                } else {
                    cv_ue(b, 2);
                    if (a.n_ref0 > 1) { cv_te(b, a.n_ref0 - 1, cref[cv_scan8(0)]); cv_te(b, a.n_ref0 - 1, cref[cv_scan8(4)]); }
                    mvd(0, 2); mvd(4, 2);
                }
            } else {
                const bool all0 = (cref[cv_scan8(0)] | cref[cv_scan8(4)] | cref[cv_scan8(8)] | cref[cv_scan8(12)]) == 0;
                cv_ue(b, all0 ? 4u : 3u);
                const signed char *sub = a.sub_partition + M * 4;
                if (a.psub8x8) { for (int i = 0; i < 4; i++) { const int sp = sub[i]; cv_ue(b, sp == CV_D_L0_8x8 ? 0u : sp == CV_D_L0_8x4 ? 1u : sp == CV_D_L0_4x8 ? 2u : 3u); } }
                else cv_put(b, 4, 0xf);
                if (!all0 && a.n_ref0 > 1) for (int i = 0; i < 4; i++) cv_te(b, a.n_ref0 - 1, cref[cv_scan8(4 * i)]);
                for (int i = 0; i < 4; i++) {
                    const int sp = sub[i];
                    if (sp == CV_D_L0_8x8) mvd(4 * i, 2);
                    else if (sp == CV_D_L0_8x4) { mvd(4 * i, 2); mvd(4 * i + 2, 2); }
                    else if (sp == CV_D_L0_4x8) { mvd(4 * i, 1); mvd(4 * i + 1, 1); }
                    else { mvd(4 * i, 1); mvd(4 * i + 1, 1); mvd(4 * i + 2, 1); mvd(4 * i + 3, 1); }
                }
            }
        }
        // ---- coded block pattern, transform size ----
        if (type == CV_I_4x4 || type == CV_I_8x8) cv_ue(b, c_cv_cbp_intra[(cbp_chroma << 4) | cbp_luma]);
        else if (type != CV_I_16x16) cv_ue(b, c_cv_cbp_inter[(cbp_chroma << 4) | cbp_luma]);
        if (a.t8_mode && cbp_luma) {                                           // x264_mb_transform_8x8_allowed
            bool allowed = type == CV_P_L0;
            if (type == CV_P_8x8) { const signed char *sub = a.sub_partition + M * 4; allowed = sub[0] == CV_D_L0_8x8 && sub[1] == CV_D_L0_8x8 && sub[2] == CV_D_L0_8x8 && sub[3] == CV_D_L0_8x8; }
            if (allowed) cv_put(b, 1, (u32)t8);
        }
        // ---- residual ----
        // the nnz cache of this macroblock in scan8 layout: neighbours' totals (0x80: none), own totals as they are written
        __shared__ u8 cn[48];
        for (int k = 0; k < 48; k++) cn[k] = 0;
        for (int k = 0; k < 4; k++) { cn[4 + k] = has_top ? tn[k] : 0x80; cn[11 + 8 * k] = has_left ? s_left_nnz[k] : 0x80; }
        for (int k = 0; k < 2; k++) {
            cn[1 + k] = has_top ? tn[4 + k] : 0x80; cn[8 + 8 * k] = has_left ? s_left_nnz[4 + k] : 0x80;         // Cb: scan8[16] = 1 + 8 * 1
            cn[1 + 8 * 3 + k] = has_top ? tn[6 + k] : 0x80; cn[8 * 4 + 8 * k] = has_left ? s_left_nnz[6 + k] : 0x80;   // Cr: scan8[20] = 1 + 8 * 4
        }
        auto nc_of = [&](int idx) -> int {
            const int s8 = cv_scan8(idx);
            int r = cn[s8 - 1] + cn[s8 - 8];
            if (r < 0x80) r = (r + 1) >> 1;
            r &= 0x7f;
            return r < 2 ? 0 : r < 4 ? 1 : r < 8 ? 2 : 3;
        };
        const bool coded = type == CV_I_16x16 || cbp_luma || cbp_chroma;
        if (coded) {                                                           // cavlc_qp_delta: the state's QP already has the rules applied (an empty I_16x16 took the previous one)
            int dqp = a.qp[M] - last_qp;
            if (dqp < -26) dqp += 52; else if (dqp > 25) dqp -= 52;
            cv_se(b, dqp);
        }
        last_qp = a.qp[M];
        const i16 *ly = a.luma + M * 256;
        if (type == CV_I_16x16) {
            cv_residual(b, a.luma_dc + M * 16, 16, nc_of(0), false, a.profile_high);
            if (cbp_luma)
                for (int i = 0; i < 16; i++) cn[cv_scan8(i)] = (u8)cv_residual(b, ly + 16 * i + 1, 15, nc_of(i), false, a.profile_high);
        } else if (cbp_luma | cbp_chroma) {
            for (int i8 = 0; i8 < 4; i8++) {
                if (!(cbp_luma >> i8 & 1)) continue;
                for (int i4 = 0; i4 < 4; i4++) {
                    const int i = 4 * i8 + i4;
                    __shared__ i16 blk[16];
                    if (t8) for (int j = 0; j < 16; j++) blk[j] = ly[64 * i8 + i4 + 4 * j];        // zigzag_interleave_8x8_cavlc
                    else for (int j = 0; j < 16; j++) blk[j] = ly[16 * i + j];
                    cn[cv_scan8(i)] = (u8)cv_residual(b, blk, 16, nc_of(i), false, a.profile_high);
                }
            }
        }
        if (cbp_chroma) {
            cv_residual(b, a.chroma_dc + M * 8, 4, 4, true, a.profile_high);
            cv_residual(b, a.chroma_dc + M * 8 + 4, 4, 4, true, a.profile_high);
            if (cbp_chroma & 2)
                for (int i = 16; i < 24; i++) cn[cv_scan8(i)] = (u8)cv_residual(b, a.chroma_ac + M * 128 + 16 * (i - 16) + 1, 15, nc_of(i), false, a.profile_high);
        }
        // what the neighbours to come read: this macroblock's bottom row and right column
        for (int k = 0; k < 4; k++) { tn[k] = cn[12 + 8 * 3 + k]; s_left_nnz[k] = cn[12 + 3 + 8 * k]; }
        for (int k = 0; k < 2; k++) {
            tn[4 + k] = cn[1 + 8 * 2 + k]; s_left_nnz[4 + k] = cn[2 + 8 * 1 + 8 * k];
            tn[6 + k] = cn[1 + 8 * 5 + k]; s_left_nnz[6 + k] = cn[2 + 8 * 4 + 8 * k];
        }
        if (a.mb_bits) a.mb_bits[M] = (int)((b.p - out) * 8 + b.n);
    }
    if (is_p && skip_run > 0) cv_ue(b, (u32)skip_run);
    cv_put(b, 1, 1);                                                           // bs_rbsp_trailing
    if (b.n) cv_put(b, 8 - b.n, 0);
    a.payload_len[bz] = (int)(b.p - out);
}

extern "C" int x264hip_cavlc_write_frame(x264hip_frame_ctx *c, const x264hip_mb_state *st, const x264hip_cavlc_params *p)
{
    if (!st || !p || !st->luma || !p->payload || !p->payload_len) { set_error("cavlc_write_frame: needs a state with coefficient levels and payload buffers"); return -1; }
    if (p->slice_type != 0 && p->slice_type != 2) { set_error("cavlc_write_frame: I and P slices (B slices are the raster variant's, which codes CABAC)"); return -1; }
    if (c->d.mb_w > CV_MAX_W) { set_error("cavlc_write_frame: %d macroblocks per row, at most %d", c->d.mb_w, CV_MAX_W); return -1; }
    if (p->payload_cap < 4096) { set_error("cavlc_write_frame: payload_cap"); return -1; }
    CvArgs a;
    a.mb_type = (const signed char *)st->mb_type; a.partition = (const signed char *)st->partition; a.sub_partition = (const signed char *)st->sub_partition;
    a.ref = (const signed char *)st->ref; a.i4mode = (const signed char *)st->i4mode; a.i16mode = (const signed char *)st->i16mode;
    a.chroma_mode = (const signed char *)st->chroma_mode; a.t8 = (const signed char *)st->t8;
    a.qp = (const signed char *)st->qp; a.slice_qp = p->slice_qp;
    a.mv = st->mv; a.cbp = st->cbp; a.luma = st->luma; a.luma_dc = st->luma_dc; a.chroma_dc = st->chroma_dc; a.chroma_ac = st->chroma_ac; a.nnz = st->nnz;
    a.payload = p->payload; a.payload_cap = p->payload_cap; a.payload_len = p->payload_len; a.mb_bits = p->mb_bits;
    a.abort_flag = st->progress + (size_t)c->d.mb_h * c->batch;
    a.mb_w = c->d.mb_w; a.mb_h = c->d.mb_h; a.slice_type = p->slice_type; a.n_ref0 = p->n_ref0; a.psub8x8 = (p->analyse_inter & 0x20) != 0;
    a.t8_mode = p->transform8x8 != 0; a.profile_high = p->transform8x8 != 0 || p->cqm_custom != 0;
    hipLaunchKernelGGL(k_cavlc_write, dim3((unsigned)c->batch), dim3(64), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
