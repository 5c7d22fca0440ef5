// device_prims.h -- exact integer primitives shared by every gfx950 kernel of
// the x264 hot path.  Written for wave64; no CUDA idioms, no portability layer.
//
// Arithmetic contracts restate x264 core 66 (R/ = x264-snapshot-20090216-2245/):
// every helper cites the reference lines whose results it must reproduce bit
// for bit.  32-bit two's-complement int with arithmetic >> throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t  u8;
typedef int16_t  i16;
typedef uint16_t u16;
typedef uint32_t u32;

#define X264HIP_WAVE 64

__device__ __forceinline__ int clip_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// ---- wave64 reductions on the DPP cross-lane path (VALU latency, no LDS round trip per step) ----
// Call these with every lane of the groups involved active.
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
#define DPP_XOR1 0xB1          /* quad_perm [1,0,3,2]: lane ^ 1 */
#define DPP_XOR2 0x4E          /* quad_perm [2,3,0,1]: lane ^ 2 */
// every lane gets the sum of its aligned group of 4 / 8 / 16 lanes
__device__ __forceinline__ int quad_sum4(int v) { v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); return v; }
__device__ __forceinline__ int half_sum8(int v) { v += dpp_mov<0x141>(v); return quad_sum4(v); }                       // row_half_mirror first
__device__ __forceinline__ int row_sum16(int v) { v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v); return v; }  // row_ror 8,4,2,1
// sum over the whole wave, returned as a scalar (uniform) value
__device__ __forceinline__ int wave_sum(int v)
{
    v = row_sum16(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (u32)__shfl_xor((int)v, m, 64);
    return v;
}
// sum over aligned groups of `g` lanes (g = power of two <= 64)
__device__ __forceinline__ int group_sum(int v, int g)
{
    for (int m = g >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ---- XCD-aware block order -----------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Per-macroblock kernels re-read their neighbours' pixels (search
// windows, filter context), so each XCD gets one contiguous run of logical
// block ids = one horizontal band of the frame, instead of every 8th group of
// macroblocks.  A bijection for any grid size; affects speed / HBM traffic only.
__device__ __forceinline__ int xcd_band_order(int bid, int nb)
{
    int q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
    return x * q + (x < r ? x : r) + i;
}

// ---- 4 packed bytes at a time ----------------------------------------------
// sum |a_i - b_i| over the 4 bytes of two dwords: one v_sad_u8.
__device__ __forceinline__ u32 sad4(u32 a, u32 b, u32 acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }

// unaligned 4-byte load from global/LDS memory assembled from bytes where
// alignment is unknown (reference planes are read at arbitrary x).
__device__ __forceinline__ u32 load4u(const u8 *p)
{
    return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
}

// ---- two 16-bit lanes in one dword (R/common/pixel.c:164-181) ---------------
// A word holds lo + (hi << 16) modulo 2^32, exactly the reference's layout, so
// each 32-bit VALU op advances two Hadamard lanes.
__device__ __forceinline__ u32 lanes_abs(u32 v)
{
    u32 m = ((v >> 15) & 0x10001u) * 0xffffu;
    return (v + m) ^ m;
}
// 4-point butterfly with the reference's output order (pixel.c:164-173)
__device__ __forceinline__ void wht4(u32 &o0, u32 &o1, u32 &o2, u32 &o3, u32 i0, u32 i1, u32 i2, u32 i3)
{
    u32 p = i0 + i1, q = i0 - i1, r = i2 + i3, s = i2 - i3;
    o0 = p + r; o2 = p - r; o1 = q + s; o3 = q - s;
}

// SATD of one 8x4 block (two 4x4 side by side), halved once; pixel.c:214-233.
__device__ __forceinline__ int satd_8x4(const u8 *a, int sa, const u8 *b, int sb)
{
    u32 t[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        u32 d[4];
#pragma unroll
        for (int x = 0; x < 4; x++)
            d[x] = (u32)((int)a[y * sa + x] - (int)b[y * sb + x]) + ((u32)((int)a[y * sa + x + 4] - (int)b[y * sb + x + 4]) << 16);
        wht4(t[y][0], t[y][1], t[y][2], t[y][3], d[0], d[1], d[2], d[3]);
    }
    u32 acc = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        u32 v0, v1, v2, v3;
        wht4(v0, v1, v2, v3, t[0][x], t[1][x], t[2][x], t[3][x]);
        acc += lanes_abs(v0) + lanes_abs(v1) + lanes_abs(v2) + lanes_abs(v3);
    }
    return (int)(((acc & 0xffffu) + (acc >> 16)) >> 1);
}
// SATD of one 4x4 block; pixel.c:187-212.
__device__ __forceinline__ int satd_4x4(const u8 *a, int sa, const u8 *b, int sb)
{
    u32 c0[4], c1[4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        u32 d0 = (int)a[y * sa] - (int)b[y * sb], d1 = (int)a[y * sa + 1] - (int)b[y * sb + 1];
        u32 d2 = (int)a[y * sa + 2] - (int)b[y * sb + 2], d3 = (int)a[y * sa + 3] - (int)b[y * sb + 3];
        u32 e0 = (d0 + d1) + ((d0 - d1) << 16), e1 = (d2 + d3) + ((d2 - d3) << 16);
        c0[y] = e0 + e1; c1[y] = e0 - e1;
    }
    u32 t0, t1, t2, t3, m;
    int total = 0;
    wht4(t0, t1, t2, t3, c0[0], c0[1], c0[2], c0[3]);
    m = lanes_abs(t0) + lanes_abs(t1) + lanes_abs(t2) + lanes_abs(t3);
    total += (int)((m & 0xffffu) + (m >> 16));
    wht4(t0, t1, t2, t3, c1[0], c1[1], c1[2], c1[3]);
    m = lanes_abs(t0) + lanes_abs(t1) + lanes_abs(t2) + lanes_abs(t3);
    total += (int)((m & 0xffffu) + (m >> 16));
    return total >> 1;
}
// unnormalised 8x8 Hadamard SATD; pixel.c:256-289.
__device__ __forceinline__ int sa8d_8x8_raw(const u8 *a, int sa, const u8 *b, int sb)
{
    u32 t[8][4];
#pragma unroll
    for (int y = 0; y < 8; y++) {
        u32 e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 d0 = (int)a[y * sa + 2 * k] - (int)b[y * sb + 2 * k];
            u32 d1 = (int)a[y * sa + 2 * k + 1] - (int)b[y * sb + 2 * k + 1];
            e[k] = (d0 + d1) + ((d0 - d1) << 16);
        }
        wht4(t[y][0], t[y][1], t[y][2], t[y][3], e[0], e[1], e[2], e[3]);
    }
    int total = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        u32 u0, u1, u2, u3, v0, v1, v2, v3;
        wht4(u0, u1, u2, u3, t[0][x], t[1][x], t[2][x], t[3][x]);
        wht4(v0, v1, v2, v3, t[4][x], t[5][x], t[6][x], t[7][x]);
        u32 m = lanes_abs(u0 + v0) + lanes_abs(u0 - v0) + lanes_abs(u1 + v1) + lanes_abs(u1 - v1)
              + lanes_abs(u2 + v2) + lanes_abs(u2 - v2) + lanes_abs(u3 + v3) + lanes_abs(u3 - v3);
        total += (int)((m & 0xffffu) + (m >> 16));
    }
    return total;
}
// AC energies of the 4x4 and 8x8 Hadamards of one 8x8 source block:
// returns (sum8 << 32) + sum4, un-normalised; pixel.c:306-344.
__device__ __forceinline__ unsigned long long hadamard_ac_8x8(const u8 *p, int stride)
{
    u32 w[32];
#pragma unroll
    for (int y = 0; y < 8; y++) {
        const u8 *r = p + y * stride;
        int g = (y & 3) + (y & 4) * 4;
        u32 e0 = (u32)(r[0] + r[1]) + ((u32)(r[0] - r[1]) << 16);
        u32 e1 = (u32)(r[2] + r[3]) + ((u32)(r[2] - r[3]) << 16);
        u32 e2 = (u32)(r[4] + r[5]) + ((u32)(r[4] - r[5]) << 16);
        u32 e3 = (u32)(r[6] + r[7]) + ((u32)(r[6] - r[7]) << 16);
        w[g] = e0 + e1; w[g + 4] = e0 - e1; w[g + 8] = e2 + e3; w[g + 12] = e2 - e3;
    }
    u32 acc4 = 0, acc8 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        u32 a0, a1, a2, a3;
        wht4(a0, a1, a2, a3, w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        w[4 * k] = a0; w[4 * k + 1] = a1; w[4 * k + 2] = a2; w[4 * k + 3] = a3;
        acc4 += lanes_abs(a0) + lanes_abs(a1) + lanes_abs(a2) + lanes_abs(a3);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        u32 a0, a1, a2, a3;
        wht4(a0, a1, a2, a3, w[k], w[8 + k], w[16 + k], w[24 + k]);
        acc8 += lanes_abs(a0) + lanes_abs(a1) + lanes_abs(a2) + lanes_abs(a3);
    }
    u32 dc = (w[0] + w[8] + w[16] + w[24]) & 0xffffu;
    int s4 = (int)((acc4 & 0xffffu) + (acc4 >> 16) - dc);
    int s8 = (int)((acc8 & 0xffffu) + (acc8 >> 16) - dc);
    return ((unsigned long long)(u32)s8 << 32) + (unsigned long long)(long long)s4;
}

// ---- transforms (R/common/dct.c) ------------------------------------------------
// forward 4x4 core transform of a residual held row-major in r[16]; the
// intermediate narrows to int16 as the reference's does (dct.c:122-155).
__device__ __forceinline__ void fwd4x4(i16 *out, const int *r)
{
    i16 mid[16];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        int a = r[4 * y] + r[4 * y + 3], b = r[4 * y + 1] + r[4 * y + 2];
        int c = r[4 * y] - r[4 * y + 3], d = r[4 * y + 1] - r[4 * y + 2];
        mid[y] = (i16)(a + b); mid[4 + y] = (i16)(2 * c + d); mid[8 + y] = (i16)(a - b); mid[12 + y] = (i16)(c - 2 * d);
    }
#pragma unroll
    for (int y = 0; y < 4; y++) {
        int a = mid[4 * y] + mid[4 * y + 3], b = mid[4 * y + 1] + mid[4 * y + 2];
        int c = mid[4 * y] - mid[4 * y + 3], d = mid[4 * y + 1] - mid[4 * y + 2];
        out[4 * y] = (i16)(a + b); out[4 * y + 1] = (i16)(2 * c + d); out[4 * y + 2] = (i16)(a - b); out[4 * y + 3] = (i16)(c - 2 * d);
    }
}
// inverse 4x4: residual (already (x+32)>>6) into res[16] row-major (dct.c:174-216)
__device__ __forceinline__ void inv4x4(int *res, const i16 *dct)
{
    i16 mid[16];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int e = dct[c] + dct[8 + c], f = dct[c] - dct[8 + c];
        int g = dct[4 + c] + (dct[12 + c] >> 1), h = (dct[4 + c] >> 1) - dct[12 + c];
        mid[4 * c] = (i16)(e + g); mid[4 * c + 1] = (i16)(f + h); mid[4 * c + 2] = (i16)(f - h); mid[4 * c + 3] = (i16)(e - g);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int e = mid[c] + mid[8 + c], f = mid[c] - mid[8 + c];
        int g = mid[4 + c] + (mid[12 + c] >> 1), h = (mid[4 + c] >> 1) - mid[12 + c];
        res[c] = (i16)((e + g + 32) >> 6); res[4 + c] = (i16)((f + h + 32) >> 6);
        res[8 + c] = (i16)((f - h + 32) >> 6); res[12 + c] = (i16)((e - g + 32) >> 6);
    }
}
// 8-point lifting steps (dct.c:238-261 and :295-321)
__device__ __forceinline__ void fwd8_1d(int *o, const int *s)
{
    int p07 = s[0] + s[7], p16 = s[1] + s[6], p25 = s[2] + s[5], p34 = s[3] + s[4];
    int m07 = s[0] - s[7], m16 = s[1] - s[6], m25 = s[2] - s[5], m34 = s[3] - s[4];
    int a0 = p07 + p34, a1 = p16 + p25, a2 = p07 - p34, a3 = p16 - p25;
    int a4 = m16 + m25 + (m07 + (m07 >> 1));
    int a5 = m07 - m34 - (m25 + (m25 >> 1));
    int a6 = m07 + m34 - (m16 + (m16 >> 1));
    int a7 = m16 - m25 + (m34 + (m34 >> 1));
    o[0] = a0 + a1;        o[1] = a4 + (a7 >> 2);
    o[2] = a2 + (a3 >> 1); o[3] = a5 + (a6 >> 2);
    o[4] = a0 - a1;        o[5] = a6 - (a5 >> 2);
    o[6] = (a2 >> 1) - a3; o[7] = (a4 >> 2) - a7;
}
__device__ __forceinline__ void inv8_1d(int *o, const int *s)
{
    int a0 = s[0] + s[4], a2 = s[0] - s[4];
    int a4 = (s[2] >> 1) - s[6], a6 = (s[6] >> 1) + s[2];
    int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    int a1 = -s[3] + s[5] - s[7] - (s[7] >> 1);
    int a3 =  s[1] + s[7] - s[3] - (s[3] >> 1);
    int a5 = -s[1] + s[7] + s[5] + (s[5] >> 1);
    int a7 =  s[3] + s[5] + s[1] + (s[1] >> 1);
    int b1 = (a7 >> 2) + a1, b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    o[0] = b0 + b7; o[1] = b2 + b5; o[2] = b4 + b3; o[3] = b6 + b1;
    o[4] = b6 - b1; o[5] = b4 - b3; o[6] = b2 - b5; o[7] = b0 - b7;
}

// ---- quantisation (R/common/quant.c:33-40, :76-178) -------------------------------
__device__ __forceinline__ int quant_one(int v, int mf, int bias)
{
    if (v > 0) v = (int)((u32)(bias + v) * (u32)mf) >> 16;
    else       v = -((int)((u32)(bias - v) * (u32)mf) >> 16);
    return (int)(i16)v;
}
__device__ __forceinline__ int dequant_one(int v, int m, int bits)
{
    if (bits >= 0) return (int)(i16)((v * m) << bits);
    return (int)(i16)((v * m + (1 << (-bits - 1))) >> -bits);
}

// ---- filters (R/common/mc.c) -----------------------------------------------------
__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return a + f - 5 * (b + e) + 20 * (c + d); }
__device__ __forceinline__ int avg4r(int a, int b, int c, int d) { return (((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1; }

// scan orders: i-th scanned coefficient = coef[scan[i]] in the reference's
// (transposed) storage; [0] frame, [1] field.  R/common/dct.c:488-562.
static __constant__ u8 c_scan4[2][16] = {
    { 0, 4, 1, 2, 5, 8, 12, 9, 6, 3, 7, 10, 13, 14, 11, 15 },
    { 0, 1, 4, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 } };
static __constant__ u8 c_scan8[2][64] = {
    { 0, 8, 1, 2, 9, 16, 24, 17, 10, 3, 4, 11, 18, 25, 32, 40,
      33, 26, 19, 12, 5, 6, 13, 20, 27, 34, 41, 48, 56, 49, 42, 35,
      28, 21, 14, 7, 15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30,
      23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63 },
    { 0, 1, 2, 8, 9, 3, 4, 10, 16, 11, 5, 6, 7, 12, 17, 24,
      18, 13, 14, 15, 19, 25, 32, 26, 20, 21, 22, 23, 27, 33, 40, 34,
      28, 29, 30, 31, 35, 41, 48, 42, 36, 37, 38, 39, 43, 49, 50, 44,
      45, 46, 47, 51, 56, 57, 52, 53, 54, 55, 58, 59, 60, 61, 62, 63 } };
// qpel -> which half-pel planes to blend (R/common/mc.c:157-158)
static __constant__ u8 c_qpel_a[16] = {0,1,1,1, 0,1,1,1, 2,3,3,3, 0,1,1,1};
static __constant__ u8 c_qpel_b[16] = {0,0,0,0, 2,2,3,2, 2,2,3,2, 2,2,3,2};
// JVT-B118 run scores (R/common/quant.c:203-211)
static __constant__ u8 c_decimate4[16] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static __constant__ u8 c_decimate8[64] = {3,3,3,3,2,2,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,
    0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};

// ---- register-only scan + decimate (no indexed private arrays, so nothing spills to scratch) -------
// frame zigzag of a 4x4 block held in c[16] (reference's transposed storage), R/common/dct.c:488-500
#define SCAN4_FRAME(lv, c) do { \
    lv[0] = c[0]; lv[1] = c[4]; lv[2] = c[1]; lv[3] = c[2]; lv[4] = c[5]; lv[5] = c[8]; lv[6] = c[12]; lv[7] = c[9]; \
    lv[8] = c[6]; lv[9] = c[3]; lv[10] = c[7]; lv[11] = c[10]; lv[12] = c[13]; lv[13] = c[14]; lv[14] = c[11]; lv[15] = c[15]; } while (0)
// JVT-B118 score (R/common/quant.c:203-239) from bit masks of the scanned levels: bit i of nzm = level i
// non-zero, of big = |level i| > 1.  Pass the masks shifted right by one for decimate_score15.
__device__ __forceinline__ int decimate_masks(u32 nzm, u32 big)
{
    if (big) return 9;
    int score = 0, prev = 32;
    // walk the set bits from the highest: run = zeros between consecutive non-zero levels
    while (nzm) {
        const int idx = 31 - __clz(nzm);
        nzm &= ~(1u << idx);
        const int below = nzm ? 31 - __clz(nzm) : -1, run = idx - below - 1;
        score += (run < 1) + (run < 3) + (run < 6);
        prev = idx;
    }
    (void)prev;
    return score;
}
#define LEVEL_MASKS(lv, nzm, big) do { nzm = 0; big = 0; _Pragma("unroll") for (int i_ = 0; i_ < 16; i_++) { \
    nzm |= (u32)(lv[i_] != 0) << i_; big |= (u32)((unsigned)(lv[i_] + 1) > 2u) << i_; } } while (0)

