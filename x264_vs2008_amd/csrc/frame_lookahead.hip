// frame_lookahead.hip -- FRAME LEVEL, part 5: the intra half of the lookahead's
// per-macroblock cost (x264_slicetype_mb_cost, R/encoder/slicetype.c:186-245):
// for every 8x8 block of the half-resolution luma plane, predict it from its
// SOURCE neighbours (not a reconstruction, so every block is independent) with
// the four 8x8 chroma-style predictors (DC, H, V, plane) and the six
// directional 8x8 luma predictors (DDL..HU) on the low-passed edge
// (x264_predict_8x8_filter with all neighbours), score each with SATD 8x8
// (mbcmp at subme > 1), keep the minimum and add the intra penalty (5).
// The result is frame->i_intra_cost[mb].
//
// Mapping: three lowres blocks per wavefront; lane = (block, mode 0..9, upper /
// lower 8x4 half).  Each lane builds its 32 predicted pixels in registers from
// the 17 + 8 neighbours, takes the two-lane SWAR Hadamard of the difference,
// the halves are added with one shuffle and a lane per block takes the minimum.
#include "device_prims.h"
#include "frame_internal.h"

using namespace x264hip;

#define LA_WAVES 4
#define LA_MB_PER_WAVE 3

__device__ __forceinline__ int la_f2(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }
__device__ __forceinline__ int la_f1(int a, int b) { return (a + b + 1) >> 1; }

// directional 8x8 predictor for pixel (x,y) from the filtered edge e[]:
// e[7-k] = left k, e[8] = top-left, e[9+k] = top k (k < 16).  H.264 8.3.2.2 /
// R/common/predict.c:618-751; modes 3 DDL, 4 DDR, 5 VR, 6 HD, 7 VL, 8 HU.
__device__ int la_dir8(int mode, const int *e, int x, int y)
{
#define EL(k) e[7 - (k)]
#define ET(k) e[9 + (k)]
#define EZ(k) e[8 + (k)]
    switch (mode) {
    case 3:
        if (x == 7 && y == 7) return la_f2(ET(14), ET(15), ET(15));
        return la_f2(ET(x + y), ET(x + y + 1), ET(x + y + 2));
    case 4:
        return la_f2(EZ(x - y - 1), EZ(x - y), EZ(x - y + 1));
    case 5: {
        int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0) return (z & 1) ? la_f2(EZ(i - 1), EZ(i), EZ(i + 1)) : la_f1(EZ(i), EZ(i + 1));
        if (z == -1) return la_f2(EL(0), EZ(0), ET(0));
        return la_f2(EL(y - 2 * x - 1), EL(y - 2 * x - 2), EL(y - 2 * x - 3));
    }
    case 6: {
        int z = 2 * y - x, i = y - (x >> 1);
        if (z >= 0) return (z & 1) ? la_f2(EZ(-i + 1), EZ(-i), EZ(-i - 1)) : la_f1(EZ(-i), EZ(-i - 1));
        if (z == -1) return la_f2(EL(0), EZ(0), ET(0));
        return la_f2(ET(x - 2 * y - 1), ET(x - 2 * y - 2), ET(x - 2 * y - 3));
    }
    case 7: {
        int i = x + (y >> 1);
        return (y & 1) ? la_f2(ET(i), ET(i + 1), ET(i + 2)) : la_f1(ET(i), ET(i + 1));
    }
    default: {
        int z = x + 2 * y, i = y + (x >> 1);
        if (z > 13) return EL(7);
        if (z == 13) return la_f2(EL(6), EL(7), EL(7));
        return (z & 1) ? la_f2(EL(i), EL(i + 1), EL(i + 2)) : la_f1(EL(i), EL(i + 1));
    }
    }
#undef EL
#undef ET
#undef EZ
}

__global__ __launch_bounds__(64 * LA_WAVES) void k_lookahead_intra(const u8 *__restrict__ low, size_t bs, int stride, int mb_w, int mb_count,
                                                                    int *__restrict__ out)
{
    __shared__ int s_cost[LA_WAVES][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = lane / 20, sub = lane % 20;              // 3 blocks x 20 lanes; lanes 60-63 idle
    const int mb = (xcd_band_order(blockIdx.x, gridDim.x) * LA_WAVES + wave) * LA_MB_PER_WAVE + slot;
    low += bs * blockIdx.y; out += (size_t)mb_count * blockIdx.y;
    const bool live = slot < LA_MB_PER_WAVE && mb < mb_count;
    int cost = 0x7fffffff;
    if (live) {
        const int mode = sub >> 1, half = sub & 1;            // modes 0-3: 8x8c DC,H,V,P; 4-9: 8x8 DDL..HU
        const u8 *src = low + (ptrdiff_t)(8 * (mb / mb_w)) * stride + 8 * (mb % mb_w);
        int top[17], left[8];                                 // top[0] = top-left, top[1+k] = pixel k of the row above
#pragma unroll
        for (int k = 0; k < 17; k++) top[k] = src[-stride - 1 + k];
#pragma unroll
        for (int k = 0; k < 8; k++) left[k] = src[(ptrdiff_t)k * stride - 1];
        int e[25];
        if (mode >= 4) {
            // x264_predict_8x8_filter(ALL_NEIGHBORS, ALL_NEIGHBORS), R/common/predict.c:499-540
            e[8] = (top[1] + 2 * top[0] + left[0] + 2) >> 2;
            e[7] = (top[0] + 2 * left[0] + left[1] + 2) >> 2;
#pragma unroll
            for (int k = 1; k < 7; k++) e[7 - k] = la_f2(left[k - 1], left[k], left[k + 1]);
            e[0] = (left[6] + 3 * left[7] + 2) >> 2;
            e[9] = (top[0] + 2 * top[1] + top[2] + 2) >> 2;
#pragma unroll
            for (int k = 1; k < 15; k++) e[9 + k] = la_f2(top[k], top[k + 1], top[k + 2]);
            e[24] = (top[15] + 3 * top[16] + 2) >> 2;
        }
        int H = 0, V = 0, b = 0, c = 0, i00 = 0, dcq[4] = {0, 0, 0, 0};
        if (mode == 3) {                                      // plane, R/common/predict.c:311-346
#pragma unroll
            for (int i = 0; i < 4; i++) { H += (i + 1) * (top[5 + i] - top[3 - i]); V += (i + 1) * ((i + 4 < 8 ? left[i + 4] : 0) - (2 - i >= 0 ? left[2 - i] : top[0])); }
            int a = 16 * (left[7] + top[8]);
            b = (17 * H + 16) >> 5; c = (17 * V + 16) >> 5;
            i00 = a - 3 * b - 3 * c + 16;
        } else if (mode == 0) {                               // 4-quadrant DC, predict.c:229-262
            int s0 = top[1] + top[2] + top[3] + top[4], s1 = top[5] + top[6] + top[7] + top[8];
            int s2 = left[0] + left[1] + left[2] + left[3], s3 = left[4] + left[5] + left[6] + left[7];
            dcq[0] = (s0 + s2 + 4) >> 3; dcq[1] = (s1 + 2) >> 2; dcq[2] = (s3 + 2) >> 2; dcq[3] = (s1 + s3 + 4) >> 3;
        }
        int d[4][8];
#pragma unroll
        for (int yy = 0; yy < 4; yy++) {
            const int y = 4 * half + yy;
#pragma unroll
            for (int x = 0; x < 8; x++) {
                int p;
                if (mode == 0) p = dcq[(y >> 2) * 2 + (x >> 2)];
                else if (mode == 1) p = left[y];
                else if (mode == 2) p = top[1 + x];
                else if (mode == 3) p = clip_u8((i00 + b * x + c * y) >> 5);
                else p = la_dir8(mode - 1, e, x, y);
                d[yy][x] = p - (int)src[(ptrdiff_t)y * stride + x];     // satd(pred, fenc)
            }
        }
        // SATD 8x4 of the difference (two 4x4 Hadamards, halved once)
        u32 t[4][4], acc = 0;
#pragma unroll
        for (int yy = 0; yy < 4; yy++) {
            u32 p0 = (u32)d[yy][0] + ((u32)d[yy][4] << 16), p1 = (u32)d[yy][1] + ((u32)d[yy][5] << 16);
            u32 p2 = (u32)d[yy][2] + ((u32)d[yy][6] << 16), p3 = (u32)d[yy][3] + ((u32)d[yy][7] << 16);
            wht4(t[yy][0], t[yy][1], t[yy][2], t[yy][3], p0, p1, p2, p3);
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            u32 v0, v1, v2, v3;
            wht4(v0, v1, v2, v3, t[0][x], t[1][x], t[2][x], t[3][x]);
            acc += lanes_abs(v0) + lanes_abs(v1) + lanes_abs(v2) + lanes_abs(v3);
        }
        cost = (int)(((acc & 0xffffu) + (acc >> 16)) >> 1);
    }
    // upper + lower half (neighbouring lanes: sub is even/odd within a 20-lane slot, 20 is even so pairs never straddle slots)
    int other = __shfl_xor(cost, 1, 64);
    int total = live ? cost + other : 0x7fffffff;
    s_cost[wave][lane] = total;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (live && sub == 0) {
        int best = 0x7fffffff;
#pragma unroll
        for (int m = 0; m < 10; m++) { int v = s_cost[wave][slot * 20 + 2 * m]; best = v < best ? v : best; }
        out[mb] = best + 5;                                    // intra_penalty, slicetype.c:196
    }
}

extern "C" int x264hip_lookahead_intra_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, int32_t *out_cost_dev)
{
    int n = c->d.mb_w * c->d.mb_h;
    int per_block = LA_WAVES * LA_MB_PER_WAVE;
    hipLaunchKernelGGL(k_lookahead_intra, dim3((n + per_block - 1) / per_block, c->batch), dim3(64 * LA_WAVES), 0, c->stream,
                       pic->lowres[0], c->bs_l, pic->stride_lowres, c->d.mb_w, n, out_cost_dev);
    HIPCHK(hipGetLastError());
    return 0;
}
