// intra_pred.h -- per-pixel intra prediction (16x16, 8x8 chroma, 4x4 and the directional modes of
// 8x8), shared by the table-level kernel (l1_tables.hip) and the macroblock sweep (frame_slice.hip).
#pragma once
#include "device_prims.h"

// Directional intra prediction of pixel (x,y) of an NxN block from an edge
// array e[] with e[n-1-k] = left k, e[n] = top-left, e[n+1+k] = top k
// (H.264 8.3.1.2 / 8.3.2.2; R/common/predict.c:398-497, :618-751).
template <typename E>   // const int * (private array) or const u8 * (edge bytes in LDS)
__device__ int dir_pred_px(int n, int mode, E e, int x, int y)
{
#define EL(k) ((int)e[n - 1 - (k)])
#define ET(k) ((int)e[n + 1 + (k)])
#define EZ(k) ((int)e[n + (k)])
#define F1(a, b) (((a) + (b) + 1) >> 1)
#define F2(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)
    switch (mode) {
    case 3:
        if (x == n - 1 && y == n - 1) return F2(ET(2 * n - 2), ET(2 * n - 1), ET(2 * n - 1));
        return F2(ET(x + y), ET(x + y + 1), ET(x + y + 2));
    case 4:
        return F2(EZ(x - y - 1), EZ(x - y), EZ(x - y + 1));
    case 5: {
        int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0) return (z & 1) ? F2(EZ(i - 1), EZ(i), EZ(i + 1)) : F1(EZ(i), EZ(i + 1));
        if (z == -1) return F2(EL(0), EZ(0), ET(0));
        return F2(EL(y - 2 * x - 1), EL(y - 2 * x - 2), EL(y - 2 * x - 3));
    }
    case 6: {
        int z = 2 * y - x, i = y - (x >> 1);
        if (z >= 0) return (z & 1) ? F2(EZ(-i + 1), EZ(-i), EZ(-i - 1)) : F1(EZ(-i), EZ(-i - 1));
        if (z == -1) return F2(EL(0), EZ(0), ET(0));
        return F2(ET(x - 2 * y - 1), ET(x - 2 * y - 2), ET(x - 2 * y - 3));
    }
    case 7: {
        int i = x + (y >> 1);
        return (y & 1) ? F2(ET(i), ET(i + 1), ET(i + 2)) : F1(ET(i), ET(i + 1));
    }
    default: {
        int z = x + 2 * y, last = 2 * n - 3, i = y + (x >> 1);
        if (z > last) return EL(n - 1);
        if (z == last) return F2(EL(n - 2), EL(n - 1), EL(n - 1));
        return (z & 1) ? F2(EL(i), EL(i + 1), EL(i + 2)) : F1(EL(i), EL(i + 1));
    }
    }
#undef EL
#undef ET
#undef EZ
}

// prediction of pixel (x,y) for the table families 16x16 (fam 0), 8x8 chroma
// (fam 1), 4x4 (fam 2).  s points at the block inside a local buffer of
// stride ls that holds row -1 and column -1.  Mode numbers are the table
// slots (R/common/predict.h:31-93).
__device__ int pred_px(int fam, int mode, const u8 *s, int ls, int x, int y)
{
#define PX(xx, yy) ((int)s[(xx) + (yy) * ls])
    const int n = fam == 0 ? 16 : fam == 1 ? 8 : 4;
    if (fam == 0 || fam == 1) {
        // slot order differs: 16x16 = V,H,DC,P,DCL,DCT,DC128; 8x8c = DC,H,V,P,DCL,DCT,DC128
        int kind = mode;   // canonical: 0 V 1 H 2 DC 3 P 4 DCL 5 DCT 6 128
        if (fam == 1) kind = mode == 0 ? 2 : mode == 2 ? 0 : mode;
        if (kind == 0) return PX(x, -1);
        if (kind == 1) return PX(-1, y);
        if (kind == 6) return 128;
        if (kind == 3) {
            int half = n / 2, H = 0, V = 0;
            for (int i = 1; i <= half; i++) {
                H += i * (PX(half - 1 + i, -1) - PX(half - 1 - i, -1));
                V += i * (PX(-1, half - 1 + i) - PX(-1, half - 1 - i));
            }
            int coef = fam == 0 ? 5 : 17, sh = fam == 0 ? 6 : 5;
            int a = 16 * (PX(-1, n - 1) + PX(n - 1, -1));
            int b = (coef * H + (1 << (sh - 1))) >> sh, c = (coef * V + (1 << (sh - 1))) >> sh;
            return clip_u8((a - (half - 1) * (b + c) + 16 + b * x + c * y) >> 5);
        }
        if (fam == 0) {
            int t = 0, l = 0;
            for (int i = 0; i < 16; i++) { t += PX(i, -1); l += PX(-1, i); }
            if (kind == 2) return (t + l + 16) >> 5;
            if (kind == 4) return (l + 8) >> 4;
            return (t + 8) >> 4;
        }
        // chroma: per-quadrant DC rules (R/common/predict.c:176-262)
        int qx = x >> 2, qy = y >> 2, t = 0, l = 0;
        for (int i = 0; i < 4; i++) { t += PX(4 * qx + i, -1); l += PX(-1, 4 * qy + i); }
        if (kind == 4) return (l + 2) >> 2;
        if (kind == 5) return (t + 2) >> 2;
        if (qx == qy) return (t + l + 4) >> 3;
        return qx ? (t + 2) >> 2 : (l + 2) >> 2;
    }
    // 4x4: V,H,DC,DDL,DDR,VR,HD,VL,HU,DCL,DCT,DC128
    if (mode == 0) return PX(x, -1);
    if (mode == 1) return PX(-1, y);
    if (mode == 11) return 128;
    if (mode == 2 || mode == 9 || mode == 10) {
        int t = 0, l = 0;
        for (int i = 0; i < 4; i++) { t += PX(i, -1); l += PX(-1, i); }
        if (mode == 2) return (t + l + 4) >> 3;
        return mode == 9 ? (l + 2) >> 2 : (t + 2) >> 2;
    }
    int e[13];
    for (int k = 0; k < 4; k++) e[3 - k] = PX(-1, k);
    e[4] = PX(-1, -1);
    for (int k = 0; k < 8; k++) e[5 + k] = PX(k, -1);
    return dir_pred_px(4, mode, e, x, y);
#undef PX
}

// ---- the same predictors from an edge array in LDS (no private arrays) ------------------------
// 4x4: e4[0..3] = left 3..0, e4[4] = top-left, e4[5..12] = top 0..7 (dir_pred_px's layout for n = 4)
__device__ __forceinline__ void pred4_edges(u8 *e4, const u8 *s, int ls, int k /* 0..12: one entry per caller */)
{
    e4[k] = k < 4 ? s[-1 + (3 - k) * ls] : k == 4 ? s[-1 - ls] : s[(k - 5) - ls];
}
__device__ __forceinline__ int pred4_px(int mode, const u8 *e4, int x, int y)
{   // table slots: V,H,DC,DDL,DDR,VR,HD,VL,HU,DC_LEFT,DC_TOP,DC_128 (R/common/predict.h:60-76)
    if (mode == 0) return e4[5 + x];
    if (mode == 1) return e4[3 - y];
    if (mode == 11) return 128;
    if (mode == 2 || mode == 9 || mode == 10) {
        const int t = e4[5] + e4[6] + e4[7] + e4[8], l = e4[0] + e4[1] + e4[2] + e4[3];
        return mode == 2 ? (t + l + 4) >> 3 : mode == 9 ? (l + 2) >> 2 : (t + 2) >> 2;
    }
    return dir_pred_px(4, mode, e4, x, y);
}
// 8x8: edge[] as x264_predict_8x8_filter leaves it (R/common/predict.c:499-540): edge[14-k] = left k,
// edge[15] = top-left, edge[16+k] = top k (k < 16)
__device__ __forceinline__ int pred8_px(int mode, const u8 *edge, int x, int y)
{
    if (mode == 0) return edge[16 + x];
    if (mode == 1) return edge[14 - y];
    if (mode == 11) return 128;
    if (mode == 2 || mode == 9 || mode == 10) {
        int l = 0, t = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { l += edge[7 + i]; t += edge[16 + i]; }
        return mode == 2 ? (l + t + 8) >> 4 : mode == 9 ? (l + 4) >> 3 : (t + 4) >> 3;
    }
    return dir_pred_px(8, mode, edge + 7, x, y);
}
// x264_predict_8x8_filter for the block at s (row -1 / column -1 reachable, stride ls); one lane runs it
__device__ __forceinline__ void pred8_filter(u8 *edge, const u8 *s, int ls, int neigh, int filt)
{
#define PX(xx, yy) ((int)s[(xx) + (yy) * ls])
    const int have_tl = neigh & 8;       // MB_TOPLEFT
    if (filt & 1) {                      // MB_LEFT
        edge[15] = (u8)((PX(0, -1) + 2 * PX(-1, -1) + PX(-1, 0) + 2) >> 2);
        edge[14] = (u8)(((have_tl ? PX(-1, -1) : PX(-1, 0)) + 2 * PX(-1, 0) + PX(-1, 1) + 2) >> 2);
        for (int y = 1; y < 7; y++) edge[14 - y] = (u8)((PX(-1, y - 1) + 2 * PX(-1, y) + PX(-1, y + 1) + 2) >> 2);
        edge[7] = (u8)((PX(-1, 6) + 3 * PX(-1, 7) + 2) >> 2);
    }
    if (filt & 2) {                      // MB_TOP
        const int have_tr = neigh & 4;   // MB_TOPRIGHT
        edge[16] = (u8)(((have_tl ? PX(-1, -1) : PX(0, -1)) + 2 * PX(0, -1) + PX(1, -1) + 2) >> 2);
        for (int x = 1; x < 7; x++) edge[16 + x] = (u8)((PX(x - 1, -1) + 2 * PX(x, -1) + PX(x + 1, -1) + 2) >> 2);
        edge[23] = (u8)((PX(6, -1) + 2 * PX(7, -1) + (have_tr ? PX(8, -1) : PX(7, -1)) + 2) >> 2);
        if (filt & 4) {
            if (have_tr) {
                for (int x = 8; x < 15; x++) edge[16 + x] = (u8)((PX(x - 1, -1) + 2 * PX(x, -1) + PX(x + 1, -1) + 2) >> 2);
                edge[31] = edge[32] = (u8)((PX(14, -1) + 3 * PX(15, -1) + 2) >> 2);
            } else
                for (int i = 24; i < 33; i++) edge[i] = (u8)PX(7, -1);
        }
    }
#undef PX
}

// ---- table-driven 4x4 / 8x8 prediction ---------------------------------------------------------
// Every pixel of every 4x4 / 8x8 mode is one entry of a small per-block table built from the edge array
// e[0..3n] (e[n-1-k] = left k, e[n] = top-left, e[n+1+k] = top k):
//   RAW[j] = e[j];  F1[j] = (e[j] + e[j+1] + 1) >> 1;  F2[j] = (e[j-1] + 2 e[j] + e[j+1] + 2) >> 2 (ends repeated),
// followed by DC, DC_LEFT, DC_TOP and 128.  Which entry pixel (x, y) of a mode reads is a compile-time
// function (pt_off below restates dir_pred_px's index arithmetic), tabulated one byte per pixel, so a lane
// predicts a row with one LUT read and n byte reads whatever its mode -- no divergent mode switch.
constexpr int pt_entries(int n) { return 3 * n + 1; }
constexpr int pt_size(int n) { return 3 * pt_entries(n) + 4; }
constexpr int pt_off(int n, int mode, int x, int y)
{
    const int E = pt_entries(n), F1 = E, F2 = 2 * E, DC = 3 * E;
    switch (mode) {
    case 0: return n + 1 + x;
    case 1: return n - 1 - y;
    case 2: return DC;
    case 9: return DC + 1;
    case 10: return DC + 2;
    case 11: return DC + 3;
    case 3: return F2 + n + 2 + x + y;
    case 4: return F2 + n + x - y;
    case 5: {
        const int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0) return (z & 1) ? F2 + n + i : F1 + n + i;
        if (z == -1) return F2 + n;
        return F2 + n + 1 - y + 2 * x;
    }
    case 6: {
        const int z = 2 * y - x, i = y - (x >> 1);
        if (z >= 0) return (z & 1) ? F2 + n - i : F1 + n - i - 1;
        if (z == -1) return F2 + n;
        return F2 + n - 1 + x - 2 * y;
    }
    case 7: {
        const int i = x + (y >> 1);
        return (y & 1) ? F2 + n + 2 + i : F1 + n + 1 + i;
    }
    default: {
        const int z = x + 2 * y, last = 2 * n - 3, i = y + (x >> 1);
        if (z > last) return 0;
        if (z == last) return F2;
        return (z & 1) ? F2 + n - 2 - i : F1 + n - 2 - i;
    }
    }
}
struct PLut4 { u32 v[12][4]; };          // [mode][row]: four byte offsets
struct PLut8 { u32 v[12][8][2]; };       // [mode][row][half]: eight byte offsets
constexpr PLut4 make_plut4()
{
    PLut4 t{};
    for (int m = 0; m < 12; m++)
        for (int y = 0; y < 4; y++) {
            u32 w = 0;
            for (int x = 0; x < 4; x++) w |= (u32)pt_off(4, m, x, y) << (8 * x);
            t.v[m][y] = w;
        }
    return t;
}
constexpr PLut8 make_plut8()
{
    PLut8 t{};
    for (int m = 0; m < 12; m++)
        for (int y = 0; y < 8; y++)
            for (int h = 0; h < 2; h++) {
                u32 w = 0;
                for (int x = 0; x < 4; x++) w |= (u32)pt_off(8, m, 4 * h + x, y) << (8 * x);
                t.v[m][y][h] = w;
            }
    return t;
}
static __constant__ PLut4 c_plut4 = make_plut4();
static __constant__ PLut8 c_plut8 = make_plut8();
