// me_exact.h -- the reference's motion search for one block on one wavefront, shared by the
// per-frame search kernel (frame_me_exact.hip) and the macroblock sweep (frame_slice.hip).
//
// x264_me_search_ref + refine_subpel (R/encoder/me.c:156-778) and x264_me_refine_qpel (:634-644).
// The walk is data dependent, so the wave is organised around "trips": one trip scores every candidate
// the reference's control flow would look at before its next decision, one candidate per lane group:
//   * full-pel trips score 4 candidates (16 lanes each, one picture row per lane: four v_sad_u8 on dwords
//     re-aligned with v_alignbyte) or 8 candidates (8 lanes each, two rows per lane);
//   * sub-pel trips score 4 quarter-pel candidates, luma AND both chroma planes at once: one lane = one
//     8x4 block of one candidate (SATD on the reference's two-lanes-per-dword layout, halved per block as
//     pixel.c:214-253 does);
//   * every lane group derives ITS candidate from the trip's centre with a few VALU ops (offset tables are
//     nibble-packed immediates), looks the mv cost up itself (one LDS read per lane) and adds it to the
//     block sum that the DPP reduction leaves in all of its lanes; the candidates are then ranked by ONE
//     key = cost << 3 | group: the minimum over the group leaders (v_readlane + s_min) is exactly the
//     reference's sequence of strict '<' tests in evaluation order.  COST_MV_SATD's "add chroma only while
//     still below the best" rule decides the same way on the finished sums, since the partial sums only grow;
//   * the walk's state (best vector, cost, direction, ranges) lives in scalar registers: all pointers carry
//     their address space and everything taken from the caller is declared wave-uniform up front.
// Reference pixels are read straight from HBM/L2 (the walk may start anywhere inside the mv limits);
// the source block and, in the sweep, the centre of the mv-cost table live in LDS.
#pragma once
#include "device_prims.h"

#define MX_COST_MAX (1 << 28)
#define MX_COST_LDS 256           // half-width of the LDS copy of p_cost_mv (quarter-pels = 64 pixels either side of the predictor; beyond it the table in HBM)

// R/encoder/me.c:34-50: subpel_iterations
static __constant__ int c_subpel_iters[10][4] = {{0,0,0,0},{1,1,0,0},{0,1,1,0},{0,2,1,0},{0,2,1,1},{0,2,1,2},{0,0,2,2},{0,0,2,2},{0,0,4,10},{0,0,4,10}};

// Pointers carry their address space: reference planes and the full mv-cost table are global memory, the source block and
// the table's centre are LDS.  (A generic pointer makes every access a flat load, which the compiler must treat as
// lane-divergent -- private memory is reachable through it -- so all control flow that depends on a loaded value turns
// into exec-mask code; with typed pointers the walk's state stays in scalar registers.)
#define MX_GLB(T_) const __attribute__((address_space(1))) T_ *
#define MX_LDS(T_) const __attribute__((address_space(3))) T_ *
#define MX_UNI(v_) __builtin_amdgcn_readfirstlane(v_)

// eight signed nibbles in a dword; element i (lane-varying or scalar)
#define MX_NIB8(a, b, c_, d, e, f, g, h) ((u32)(((a) & 15) | (((b) & 15) << 4) | (((c_) & 15) << 8) | (((d) & 15) << 12) | (((e) & 15) << 16) | \
                                                (((f) & 15) << 20) | (((g) & 15) << 24) | (((u32)(h) & 15u) << 28)))
__device__ __forceinline__ int mx_nib(u32 pk, int i) { return ((int)(pk << (28 - 4 * i))) >> 28; }
// me.c:45-50: hex2 (radius-2 hexagon with repeats) x / y, indexed dir + 1; mod6m1[dir + 1] = (dir + 6) % 6
#define MX_HEX2_DX MX_NIB8(-1, -2, -1, 1, 2, 1, -1, -2)
#define MX_HEX2_DY MX_NIB8(-2, 0, 2, 2, 0, -2, -2, 0)
// the small diamond (0,-1) (0,1) (-1,0) (1,0)
#define MX_DIA_DX MX_NIB8(0, 0, -1, 1, 0, 0, 0, 0)
#define MX_DIA_DY MX_NIB8(-1, 1, 0, 0, 0, 0, 0, 0)

// 16 / 8(+1) consecutive bytes at an arbitrary address as dwords (aligned loads + v_alignbyte)
__device__ __forceinline__ void load16u(MX_GLB(u8) p, u32 o[4])
{
    const uintptr_t a = (uintptr_t)p;
    const u32 s = (u32)(a & 3);
    MX_GLB(u32) q = (MX_GLB(u32))(a - s);
    u32 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
    o[0] = __builtin_amdgcn_alignbyte(w1, w0, s); o[1] = __builtin_amdgcn_alignbyte(w2, w1, s);
    o[2] = __builtin_amdgcn_alignbyte(w3, w2, s); o[3] = __builtin_amdgcn_alignbyte(w4, w3, s);
}
__device__ __forceinline__ void load9u(MX_GLB(u8) p, u32 &o0, u32 &o1, u32 &o2)   // bytes 0..7 in o0,o1; byte 8 in the low byte of o2
{
    const uintptr_t a = (uintptr_t)p;
    const u32 s = (u32)(a & 3);
    MX_GLB(u32) q = (MX_GLB(u32))(a - s);
    u32 w0 = q[0], w1 = q[1], w2 = q[2];
    o0 = __builtin_amdgcn_alignbyte(w1, w0, s); o1 = __builtin_amdgcn_alignbyte(w2, w1, s); o2 = __builtin_amdgcn_alignbyte(0u, w2, s);
}
// rounded byte-wise average of two dwords: (a + b + 1) >> 1 per byte, no carries across bytes
__device__ __forceinline__ u32 avg4(u32 a, u32 b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
__device__ __forceinline__ int byte_of(u32 w, int k) { return (int)((w >> (8 * k)) & 255u); }

struct MxCtx {
    MX_LDS(u32) fe;           // LDS: 16 rows x 4 dwords
    MX_LDS(u8) fe_u; MX_LDS(u8) fe_v;    // LDS: 8x8 each
    MX_GLB(u8) pl[4];         // four half-pel planes at the macroblock origin
    // a lane-dependent plane: selects, NOT pl[k] -- a dynamically indexed member forces the whole context into scratch memory
    // (every field access a scratch load, every search a 200-byte-per-lane scratch copy)
    __device__ __forceinline__ MX_GLB(u8) plane(int k) const { return k == 0 ? pl[0] : k == 1 ? pl[1] : k == 2 ? pl[2] : pl[3]; }
    MX_GLB(u8) cu; MX_GLB(u8) cv;        // chroma planes at the macroblock origin
    MX_GLB(i16) cost_g;       // p_cost_mv, centred (global memory)
    MX_LDS(i16) cost_l;       // LDS copy of cost_g[-MX_COST_LDS .. MX_COST_LDS] (only read when has_cost_l)
    bool has_cost_l;
    // optional LDS staging of the sub-pel neighbourhood (mx_load_patch): the four half-pel planes and both chroma planes
    // around the refinement's start, so that its rounds read LDS instead of issuing dozens of scattered global loads each
    MX_LDS(u8) patch;
    bool has_patch, patch_on;
    int px0, py0, cx0, cy0;   // top-left sample of the luma / chroma patch relative to the block's position
    int mvpx, mvpy;           // the predictor the costs are relative to
    int sy, sc, lane;
    // the block searched: 16x16, 16x8, 8x16 or 8x8 at (bx, by) inside the macroblock.  pl / cu / cv point at the block
    // (chroma at bx/2, by/2); fe_off = dword offset of the block in fe (by*4 + bx/4), cfe_off = byte offset in fe_u / fe_v
    int bw, bh, fe_off, cfe_off;
    __device__ __forceinline__ void set_block(int w, int h, int bx, int by) { bw = w; bh = h; fe_off = by * 4 + (bx >> 2); cfe_off = (by >> 1) * 8 + (bx >> 1); }
    __device__ __forceinline__ int cost1(int d) const
    {
        // d is wave-uniform: keep the looked-up cost in a scalar register
        if (has_cost_l && (unsigned)(d + MX_COST_LDS) <= 2u * MX_COST_LDS) return MX_UNI((int)cost_l[d + MX_COST_LDS]);
        return MX_UNI((int)cost_g[d]);
    }
    __device__ __forceinline__ int cost(int mx, int my) const { return cost1(mx - mvpx) + cost1(my - mvpy); }   // p_cost_mvx[mx] + p_cost_mvy[my]
    // the same for a lane-varying quarter-pel vector: one LDS read per component unless some lane is outside the copy
    __device__ __forceinline__ int lane_cost(int qx, int qy) const
    {
        const int dx = qx - mvpx, dy = qy - mvpy;
        const bool in = has_cost_l && (unsigned)(dx + MX_COST_LDS) <= 2u * MX_COST_LDS && (unsigned)(dy + MX_COST_LDS) <= 2u * MX_COST_LDS;
        if (__ballot(!in) == 0) return (int)cost_l[dx + MX_COST_LDS] + (int)cost_l[dy + MX_COST_LDS];
        return (int)cost_g[dx] + (int)cost_g[dy];
    }
};

// ---- LDS staging of the sub-pel neighbourhood ----------------------------------------------------
#define MX_PS 28                       // luma patch row stride: bw + 5 samples wanted, dword reads may run 3 further
#define MX_PROWS 21                    // bh + 5 rows
#define MX_PPL (MX_PS * MX_PROWS)      // one luma plane
#define MX_CS 16                       // chroma patch: bw/2 + 3 samples, bh/2 + 3 rows
#define MX_CROWS 11
#define MX_PATCH_BYTES (4 * MX_PPL + 2 * MX_CS * MX_CROWS + 16)
// 16 / 8(+1) consecutive bytes at an arbitrary LDS address (aligned dword reads + v_alignbyte)
__device__ __forceinline__ void lds16u(MX_LDS(u8) p, u32 o[4])
{
    const u32 a = (u32)(uintptr_t)p, s = a & 3;
    MX_LDS(u32) q = (MX_LDS(u32))(uintptr_t)(a - s);
    const u32 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
    o[0] = __builtin_amdgcn_alignbyte(w1, w0, s); o[1] = __builtin_amdgcn_alignbyte(w2, w1, s);
    o[2] = __builtin_amdgcn_alignbyte(w3, w2, s); o[3] = __builtin_amdgcn_alignbyte(w4, w3, s);
}
__device__ __forceinline__ void lds9u(MX_LDS(u8) p, u32 &o0, u32 &o1, u32 &o2)
{
    const u32 a = (u32)(uintptr_t)p, s = a & 3;
    MX_LDS(u32) q = (MX_LDS(u32))(uintptr_t)(a - s);
    const u32 w0 = q[0], w1 = q[1], w2 = q[2];
    o0 = __builtin_amdgcn_alignbyte(w1, w0, s); o1 = __builtin_amdgcn_alignbyte(w2, w1, s); o2 = __builtin_amdgcn_alignbyte(0u, w2, s);
}
// Stage what refine_subpel can reach from its start (qx, qy): +-6 quarter-pels, i.e. full-pel columns (qx >> 2) - 2 ..
// (qx >> 2) + bw + 2 of each of the four planes (bw + 5 samples, bh + 5 rows) and the chroma samples (qx >> 3) - 1 ..
// + bw/2 + 1 (bw/2 + 3 samples, bh/2 + 3 rows).  Lane = one row: 84 luma rows in two passes, the 22 chroma rows ride in
// the second.  Only the samples the rounds can read are loaded (the same ones the global path would touch).
__device__ __forceinline__ void mx_load_patch(MxCtx &c, int qx, int qy)
{
    typedef __attribute__((address_space(3))) u32 *lds_w;
    c.px0 = (qx >> 2) - 2; c.py0 = (qy >> 2) - 2; c.cx0 = (qx >> 3) - 1; c.cy0 = (qy >> 3) - 1;
    const int prow = c.bh + 5, crow = (c.bh >> 1) + 3, nl = 4 * prow;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int i = c.lane + 64 * pass;
        if (i < nl) {
            const int k = i / prow, r = i - k * prow;
            MX_GLB(u8) src = c.plane(k) + (ptrdiff_t)(c.py0 + r) * c.sy + c.px0;
            const uintptr_t a = (uintptr_t)src;
            const u32 sft = (u32)(a & 3);
            MX_GLB(u32) q = (MX_GLB(u32))(a - sft);
            lds_w d = (lds_w)(uintptr_t)(u32)(uintptr_t)(c.patch + k * MX_PPL + r * MX_PS);
            if (c.bw == 16) {
                const u32 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4], w5 = q[5], w6 = q[6];
                d[0] = __builtin_amdgcn_alignbyte(w1, w0, sft); d[1] = __builtin_amdgcn_alignbyte(w2, w1, sft); d[2] = __builtin_amdgcn_alignbyte(w3, w2, sft);
                d[3] = __builtin_amdgcn_alignbyte(w4, w3, sft); d[4] = __builtin_amdgcn_alignbyte(w5, w4, sft); d[5] = __builtin_amdgcn_alignbyte(w6, w5, sft);
            } else {
                const u32 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
                d[0] = __builtin_amdgcn_alignbyte(w1, w0, sft); d[1] = __builtin_amdgcn_alignbyte(w2, w1, sft);
                d[2] = __builtin_amdgcn_alignbyte(w3, w2, sft); d[3] = __builtin_amdgcn_alignbyte(w4, w3, sft);
            }
        } else if (i - nl < 2 * crow) {
            const int j = i - nl, k = j >= crow, r = j - k * crow;
            MX_GLB(u8) src = (k ? c.cv : c.cu) + (ptrdiff_t)(c.cy0 + r) * c.sc + c.cx0;
            const uintptr_t a = (uintptr_t)src;
            const u32 sft = (u32)(a & 3);
            MX_GLB(u32) q = (MX_GLB(u32))(a - sft);
            lds_w d = (lds_w)(uintptr_t)(u32)(uintptr_t)(c.patch + 4 * MX_PPL + (k * MX_CROWS + r) * MX_CS);
            const u32 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];
            d[0] = __builtin_amdgcn_alignbyte(w1, w0, sft); d[1] = __builtin_amdgcn_alignbyte(w2, w1, sft); d[2] = __builtin_amdgcn_alignbyte(w3, w2, sft);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    c.patch_on = true;
}
// true when every lane's quarter-pel candidate reads inside the staged neighbourhood
__device__ __forceinline__ bool mx_in_patch(const MxCtx &c, int mx, int my)
{
    if (!c.has_patch || !c.patch_on) return false;
    const int fx = (mx >> 2) - c.px0, fy = (my >> 2) - c.py0, gx = (mx >> 3) - c.cx0, gy = (my >> 3) - c.cy0;
    const bool in = fx >= 0 && fx <= 4 && fy >= 0 && fy <= 4 && gx >= 0 && gx <= 2 && gy >= 0 && gy <= 2;
    return __ballot(!in) == 0;
}

// ---- block sums for this lane group's candidate; the result is present in every lane of the group ----
// SAD of the full-pel candidate (mx, my): 16 lanes per candidate, one picture row per lane
__device__ __forceinline__ int sad_fpel16_lane(const MxCtx &c, int mx, int my)
{
    const int row = c.lane & 15;
    int v = 0;
    if (row < c.bh) {
        MX_GLB(u8) p = c.pl[0] + (ptrdiff_t)(my + row) * c.sy + mx;
        MX_LDS(u32) f = c.fe + c.fe_off + 4 * row;
        u32 s;
        if (c.bw == 16) {
            u32 r[4];
            load16u(p, r);
            s = sad4(r[0], f[0], 0); s = sad4(r[1], f[1], s); s = sad4(r[2], f[2], s); s = sad4(r[3], f[3], s);
        } else {
            u32 r0, r1, t;
            load9u(p, r0, r1, t);
            s = sad4(r0, f[0], 0);
            if (c.bw == 8) s = sad4(r1, f[1], s);
        }
        v = (int)s;
    }
    return row_sum16(v);
}
// the same with 8 lanes per candidate, two picture rows per lane
__device__ __forceinline__ int sad_fpel8_lane(const MxCtx &c, int mx, int my)
{
    const int r = c.lane & 7;
    u32 s = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int row = r + 8 * half;
        if (row < c.bh) {
            MX_GLB(u8) p = c.pl[0] + (ptrdiff_t)(my + row) * c.sy + mx;
            MX_LDS(u32) f = c.fe + c.fe_off + 4 * row;
            if (c.bw == 16) {
                u32 a[4];
                load16u(p, a);
                s = sad4(a[0], f[0], s); s = sad4(a[1], f[1], s); s = sad4(a[2], f[2], s); s = sad4(a[3], f[3], s);
            } else {
                u32 a0, a1, t;
                load9u(p, a0, a1, t);
                s = sad4(a0, f[0], s);
                if (c.bw == 8) s = sad4(a1, f[1], s);
            }
        }
    }
    return half_sum8((int)s);
}
// SAD of the quarter-pel candidate (mx, my) through get_ref's blend (mc.c:181-202): 16 lanes per candidate
__device__ __forceinline__ int sad_qpel16_lane(const MxCtx &c, int mx, int my)
{
    const int row = c.lane & 15;
    const int fx = mx & 3, fy = my & 3, idx = fy * 4 + fx;
    const ptrdiff_t base = (ptrdiff_t)((my >> 2) + row) * c.sy + (mx >> 2);
    int v = 0;
    if (mx_in_patch(c, mx, my)) {
        if (row < c.bh) {
            const int o = ((my >> 2) - c.py0 + row) * MX_PS + (mx >> 2) - c.px0;
            MX_LDS(u8) pa = c.patch + c_qpel_a[idx] * MX_PPL + o + (fy == 3) * MX_PS; MX_LDS(u8) pb = c.patch + c_qpel_b[idx] * MX_PPL + o + (fx == 3);
            MX_LDS(u32) f = c.fe + c.fe_off + 4 * row;
            u32 s;
            if (c.bw == 16) {
                u32 a[4];
                lds16u(pa, a);
                if (idx & 5) {
                    u32 b[4];
                    lds16u(pb, b);
#pragma unroll
                    for (int k = 0; k < 4; k++) a[k] = avg4(a[k], b[k]);
                }
                s = sad4(a[0], f[0], 0); s = sad4(a[1], f[1], s); s = sad4(a[2], f[2], s); s = sad4(a[3], f[3], s);
            } else {
                u32 a0, a1, t;
                lds9u(pa, a0, a1, t);
                if (idx & 5) { u32 b0, b1; lds9u(pb, b0, b1, t); a0 = avg4(a0, b0); a1 = avg4(a1, b1); }
                s = sad4(a0, f[0], 0); if (c.bw == 8) s = sad4(a1, f[1], s);
            }
            v = (int)s;
        }
        return row_sum16(v);
    }
    if (row < c.bh) {
        MX_GLB(u8) pa = c.plane(c_qpel_a[idx]) + base + (fy == 3) * c.sy; MX_GLB(u8) pb = c.plane(c_qpel_b[idx]) + base + (fx == 3);
        MX_LDS(u32) f = c.fe + c.fe_off + 4 * row;
        u32 s;
        if (c.bw == 16) {
            u32 a[4];
            load16u(pa, a);
            if (idx & 5) {
                u32 b[4];
                load16u(pb, b);
#pragma unroll
                for (int k = 0; k < 4; k++) a[k] = avg4(a[k], b[k]);
            }
            s = sad4(a[0], f[0], 0); s = sad4(a[1], f[1], s); s = sad4(a[2], f[2], s); s = sad4(a[3], f[3], s);
        } else {
            u32 a0, a1, t;
            load9u(pa, a0, a1, t);
            if (idx & 5) { u32 b0, b1; load9u(pb, b0, b1, t); a0 = avg4(a0, b0); a1 = avg4(a1, b1); }
            s = sad4(a0, f[0], 0); if (c.bw == 8) s = sad4(a1, f[1], s);
        }
        v = (int)s;
    }
    return row_sum16(v);
}
// the same with 8 lanes per candidate, two picture rows per lane (global memory only: the predictor stage, whose
// candidates lie anywhere inside the mv limits)
__device__ __forceinline__ int sad_qpel8_lane(const MxCtx &c, int mx, int my)
{
    const int r = c.lane & 7;
    const int fx = mx & 3, fy = my & 3, idx = fy * 4 + fx;
    u32 s = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int row = r + 8 * half;
        if (row < c.bh) {
            const ptrdiff_t base = (ptrdiff_t)((my >> 2) + row) * c.sy + (mx >> 2);
            MX_GLB(u8) pa = c.plane(c_qpel_a[idx]) + base + (fy == 3) * c.sy; MX_GLB(u8) pb = c.plane(c_qpel_b[idx]) + base + (fx == 3);
            MX_LDS(u32) f = c.fe + c.fe_off + 4 * row;
            if (c.bw == 16) {
                u32 a[4];
                load16u(pa, a);
                if (idx & 5) {
                    u32 b[4];
                    load16u(pb, b);
#pragma unroll
                    for (int k = 0; k < 4; k++) a[k] = avg4(a[k], b[k]);
                }
                s = sad4(a[0], f[0], s); s = sad4(a[1], f[1], s); s = sad4(a[2], f[2], s); s = sad4(a[3], f[3], s);
            } else {
                u32 a0, a1, t;
                load9u(pa, a0, a1, t);
                if (idx & 5) { u32 b0, b1; load9u(pb, b0, b1, t); a0 = avg4(a0, b0); a1 = avg4(a1, b1); }
                s = sad4(a0, f[0], s); if (c.bw == 8) s = sad4(a1, f[1], s);
            }
        }
    }
    return half_sum8((int)s);
}
// vertical half of the 8x4 SATD when the four rows of a block sit in lanes l, l^1, l^2, l^3
__device__ __forceinline__ int satd_rows4(u32 t0, u32 t1, u32 t2, u32 t3, int lane)
{
    u32 t[4] = {t0, t1, t2, t3}, acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 v = t[k], o = (u32)dpp_mov<DPP_XOR1>((int)v);
        v = (lane & 1) ? o - v : v + o;                 // rows (0,1) and (2,3): sum / difference
        o = (u32)dpp_mov<DPP_XOR2>((int)v);
        v = (lane & 2) ? o - v : v + o;                 // second butterfly level
        acc += lanes_abs(v);
    }
    acc = (u32)quad_sum4((int)acc);
    return (int)(((acc & 0xffffu) + (acc >> 16)) >> 1);
}

// cost of one 8x4 block whose four rows are given as source / prediction dword pairs:
// SATD (x264_pixel_satd_8x4, pixel.c:214-233) or SAD
__device__ __forceinline__ int blk8x4_cost(const u32 f[4][2], const u32 p[4][2], int satd)
{
    if (!satd) {
        u32 s = 0;
#pragma unroll
        for (int y = 0; y < 4; y++) { s = sad4(f[y][0], p[y][0], s); s = sad4(f[y][1], p[y][1], s); }
        return (int)s;
    }
    u32 t[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        u32 d[4];
#pragma unroll
        for (int x = 0; x < 4; x++)
            d[x] = (u32)(byte_of(f[y][0], x) - byte_of(p[y][0], x)) + ((u32)(byte_of(f[y][1], x) - byte_of(p[y][1], x)) << 16);
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

// Pixel x of a row's left half and pixel x of its right half side by side in 16-bit lanes (x264's SATD layout, pixel.c:175-181):
// pair x of the eight pixels held in (w0 = px 0..3, w1 = px 4..7) = w0.byte[x] | w1.byte[x] << 16, one v_perm_b32
__device__ __forceinline__ u32 mx_pair(u32 w1, u32 w0, int x) { return __builtin_amdgcn_perm(w1, w0, 0x0c000c00u | (u32)x | ((u32)(4 + x) << 16)); }
// SATD of an 8x4 block from the packed differences d[y][x] = F - P (plain 32-bit subtraction of the pairs), pixel.c:214-233
__device__ __forceinline__ int satd8x4_packed(const u32 d[4][4])
{
    u32 t[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) wht4(t[y][0], t[y][1], t[y][2], t[y][3], d[y][0], d[y][1], d[y][2], d[y][3]);
    u32 acc = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        u32 v0, v1, v2, v3;
        wht4(v0, v1, v2, v3, t[0][x], t[1][x], t[2][x], t[3][x]);
        acc += lanes_abs(v0) + lanes_abs(v1) + lanes_abs(v2) + lanes_abs(v3);
    }
    return (int)(((acc & 0xffffu) + (acc >> 16)) >> 1);
}
typedef unsigned short mx_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mx_u16x2 mx_as_pk(u32 v) { return __builtin_bit_cast(mx_u16x2, v); }
__device__ __forceinline__ u32 mx_as_u32(mx_u16x2 v) { return __builtin_bit_cast(u32, v); }

// COST_MV_SATD's sum (me.c:654-677) for this lane group's quarter-pel candidate, 16 lanes per candidate:
// mbcmp_unaligned[block] of the get_ref prediction + (chroma) mbcmp[chroma block] of mc_chroma for U and V.
// Lane j of the group = one unit: 0-7 luma 8x4 blocks, 8-9 U, 10-11 V, each halved on its own as the
// reference's composites do (pixel.c:235-253); the units add up to the whole sum.
// With SATD the pixels go straight into the packed pair layout (v_perm_b32), mc_chroma's bilinear blend runs on
// pairs (v_pk_mul / v_pk_mad: every term stays below 2^16) and a difference is one 32-bit subtraction.
__device__ __forceinline__ int subpel_sum16_lane(const MxCtx &c, int mx, int my, int satd, int chroma)
{
    const int j = c.lane & 15;
    int v = 0;
    const int nbx = c.bw == 16 ? 2 : 1, n_luma = nbx * (c.bh >> 2), n_cunits = c.bh >> 3;   // 8x4 (4x4 when bw = 4) luma blocks; 4-row chroma units per plane
    const u32 lkeep = c.bw == 4 ? 0x0000ffffu : 0xffffffffu;                          // a 4-wide block: the right half of the unit contributes nothing
    const bool staged = mx_in_patch(c, mx, my);
    u32 d[4][4];                 // packed differences of this lane's 8x4 unit (SATD): luma and chroma lanes share one transform below
    bool have_d = false;
    if (j < n_luma) {
        const int bx = (nbx == 2 ? (j & 1) : 0) * 8, by = (nbx == 2 ? (j >> 1) : j) * 4;
        const int fx = mx & 3, fy = my & 3, idx = fy * 4 + fx;
        const ptrdiff_t base = (ptrdiff_t)((my >> 2) + by) * c.sy + (mx >> 2) + bx;
        MX_GLB(u8) pa = c.plane(c_qpel_a[idx]) + base + (fy == 3) * c.sy; MX_GLB(u8) pb = c.plane(c_qpel_b[idx]) + base + (fx == 3);
        u32 f[4][2], p[4][2];
        if (staged) {
            const int o = ((my >> 2) - c.py0 + by) * MX_PS + (mx >> 2) - c.px0 + bx;
            MX_LDS(u8) la = c.patch + c_qpel_a[idx] * MX_PPL + o + (fy == 3) * MX_PS; MX_LDS(u8) lb = c.patch + c_qpel_b[idx] * MX_PPL + o + (fx == 3);
#pragma unroll
            for (int y = 0; y < 4; y++) {
                u32 t;
                lds9u(la + y * MX_PS, p[y][0], p[y][1], t);
                if (idx & 5) {
                    u32 b0, b1;
                    lds9u(lb + y * MX_PS, b0, b1, t);
                    p[y][0] = avg4(p[y][0], b0); p[y][1] = avg4(p[y][1], b1);
                }
            }
        } else {
#pragma unroll
            for (int y = 0; y < 4; y++) {
                u32 t;
                load9u(pa + (ptrdiff_t)y * c.sy, p[y][0], p[y][1], t);
                if (idx & 5) {
                    u32 b0, b1;
                    load9u(pb + (ptrdiff_t)y * c.sy, b0, b1, t);
                    p[y][0] = avg4(p[y][0], b0); p[y][1] = avg4(p[y][1], b1);
                }
            }
        }
#pragma unroll
        for (int y = 0; y < 4; y++) { f[y][0] = c.fe[c.fe_off + (by + y) * 4 + (bx >> 2)]; f[y][1] = c.fe[c.fe_off + (by + y) * 4 + (bx >> 2) + 1]; }
        if (satd) {
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) d[y][x] = (mx_pair(f[y][1], f[y][0], x) & lkeep) - (mx_pair(p[y][1], p[y][0], x) & lkeep);
            have_d = true;
        } else {
            if (c.bw == 4) {
#pragma unroll
                for (int y = 0; y < 4; y++) { f[y][1] = 0; p[y][1] = 0; }
            }
            v = blk8x4_cost(f, p, 0);
        }
    } else if (chroma && j >= 8 && j < 12 && ((j - 8) & 1) < n_cunits) {
        // mbcmp[i_pixel + 3]: 8x8 / 8x4 chroma blocks are 8x4 units, 4x8 / 4x4 ones are 4x4 units
        const int by = ((j - 8) & 1) * 4, wide = c.bw == 16;
        MX_GLB(u8) plane = j < 10 ? c.cu : c.cv; MX_LDS(u8) fe = (j < 10 ? c.fe_u : c.fe_v) + c.cfe_off;
        const int dx = mx & 7, dy = my & 7;
        const int ca = (8 - dx) * (8 - dy), cb = dx * (8 - dy), cc = (8 - dx) * dy, cd = dx * dy;
        MX_GLB(u8) s = plane + (ptrdiff_t)((my >> 3) + by) * c.sc + (mx >> 3);
        u32 r0[5], r1[5], r2[5];
        if (staged) {
            MX_LDS(u8) ls = c.patch + 4 * MX_PPL + ((j >= 10) * MX_CROWS + (my >> 3) - c.cy0 + by) * MX_CS + (mx >> 3) - c.cx0;
#pragma unroll
            for (int y = 0; y < 5; y++) lds9u(ls + y * MX_CS, r0[y], r1[y], r2[y]);
        } else {
#pragma unroll
            for (int y = 0; y < 5; y++) load9u(s + (ptrdiff_t)y * c.sc, r0[y], r1[y], r2[y]);
        }
        if (satd) {
            // pair x of a row = (px x, px x + 4); the row moved one sample right is pairs 1..3 and (px 4, px 8)
            u32 a[5][5];
#pragma unroll
            for (int y = 0; y < 5; y++) {
#pragma unroll
                for (int x = 0; x < 4; x++) a[y][x] = mx_pair(r1[y], r0[y], x);
                a[y][4] = mx_pair(r2[y], r1[y], 0);
            }
            const mx_u16x2 ka = {(unsigned short)ca, (unsigned short)ca}, kb = {(unsigned short)cb, (unsigned short)cb};
            const mx_u16x2 kc = {(unsigned short)cc, (unsigned short)cc}, kd = {(unsigned short)cd, (unsigned short)cd}, k32 = {32, 32};
            const u32 keep = wide ? 0xffffffffu : 0x0000ffffu;      // a 4-wide unit: the right half contributes nothing
#pragma unroll
            for (int y = 0; y < 4; y++) {
                MX_LDS(u32) fr = (MX_LDS(u32))(fe + (by + y) * 8);
                const u32 f0 = fr[0], f1 = fr[1];
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const mx_u16x2 pv = (mx_as_pk(a[y][x]) * ka + mx_as_pk(a[y][x + 1]) * kb + mx_as_pk(a[y + 1][x]) * kc + mx_as_pk(a[y + 1][x + 1]) * kd + k32) >> 6;
                    d[y][x] = (mx_pair(f1, f0, x) & keep) - (mx_as_u32(pv) & keep);
                }
            }
            have_d = true;
        } else {
            u32 f[4][2], p[4][2];
#pragma unroll
            for (int y = 0; y < 4; y++) {
                u32 w0 = 0, w1 = 0;
#pragma unroll
                for (int x = 0; x < 8; x++) {
                    const int a0 = x < 4 ? byte_of(r0[y], x) : byte_of(r1[y], x - 4), a1 = x < 3 ? byte_of(r0[y], x + 1) : x < 7 ? byte_of(r1[y], x - 3) : byte_of(r2[y], 0);
                    const int b0 = x < 4 ? byte_of(r0[y + 1], x) : byte_of(r1[y + 1], x - 4), b1 = x < 3 ? byte_of(r0[y + 1], x + 1) : x < 7 ? byte_of(r1[y + 1], x - 3) : byte_of(r2[y + 1], 0);
                    const u32 px = (u32)((ca * a0 + cb * a1 + cc * b0 + cd * b1 + 32) >> 6);
                    if (x < 4) w0 |= px << (8 * x); else w1 |= px << (8 * (x - 4));
                }
                MX_LDS(u32) fr = (MX_LDS(u32))(fe + (by + y) * 8);
                p[y][0] = w0; f[y][0] = fr[0];
                p[y][1] = wide ? w1 : 0u; f[y][1] = wide ? fr[1] : 0u;
            }
            v = blk8x4_cost(f, p, 0);
        }
    }
    if (have_d) v = satd8x4_packed(d);
    return row_sum16(v);
}

// The reference's sequence of "if (cost < bcost) take it" tests over the candidates of one trip, in candidate order:
// key = cost << 3 | group; the smallest key over the group leaders is the first candidate with the lowest cost.
// SH = log2(lanes per group).  Groups with ok == false take no part.  Costs stay below 2^28.
template <int SH> __device__ __forceinline__ u32 mx_best_key(int cost, bool ok, int lane)
{
    const u32 key = ok ? ((u32)cost << 3) | (u32)(lane >> SH) : 0xffffffffu;
    u32 k = (u32)__builtin_amdgcn_readlane((int)key, 0);
#pragma unroll
    for (int g = 1; g < (64 >> SH); g++) { const u32 t = (u32)__builtin_amdgcn_readlane((int)key, g << SH); k = t < k ? t : k; }
    return k;
}
// if the trip's best beats bcost: take its cost and the (x, y) its lane group holds
#define MX_TAKE(SH_, key_, bcost_, x_, y_, bx_, by_) do { if (((key_) >> 3) < (u32)(bcost_)) { (bcost_) = (int)((key_) >> 3); \
        const int l_ = (int)((key_) & 7u) << (SH_); (bx_) = __builtin_amdgcn_readlane((x_), l_); (by_) = __builtin_amdgcn_readlane((y_), l_); } } while (0)

// mv limits of one macroblock, R/encoder/analyse.c:258-298 (one thread, frame coding)
struct MeLimits { int smin0, smax0, smin1, smax1, fmin0, fmax0, fmin1, fmax1; };
__device__ __forceinline__ MeLimits me_limits(int mbx, int mby, int mb_w, int mb_h, int mv_range)
{
    MeLimits L;
    const int fr = 4 * mv_range, lo = 4 * (-512 + 8) > -fr ? 4 * (-512 + 8) : -fr;
    L.smin0 = clip3(4 * (-16 * mbx - 24), -fr, fr - 1); L.smax0 = clip3(4 * (16 * (mb_w - mbx - 1) + 24), -fr, fr - 1);
    L.smin1 = clip3(4 * (-16 * mby - 24), lo, fr);      L.smax1 = clip3(4 * (16 * (mb_h - mby - 1) + 24), -fr, fr - 1);
    L.fmin0 = (L.smin0 >> 2) + 5; L.fmax0 = (L.smax0 >> 2) - 5; L.fmin1 = (L.smin1 >> 2) + 5; L.fmax1 = (L.smax1 >> 2) - 5;
    return L;
}
struct MeOpts { int method, me_range, subme, chroma_me, sad_only; };   // sad_only: lossless (mbcmp = SAD at every subme, R/encoder/encoder.c:594-604)

// Everything the search takes from its caller is the same in all lanes except MxCtx::lane; say so (v_readfirstlane), so
// that the walk's comparisons and branches are scalar.
template <class P> __device__ __forceinline__ P mx_uni_ptr(P p)
{
    const uintptr_t a = (uintptr_t)p;
    const u32 lo = (u32)MX_UNI((int)(u32)a), hi = (u32)MX_UNI((int)(u32)(a >> 32));
    return (P)(((uintptr_t)hi << 32) | lo);
}
__device__ __forceinline__ MxCtx mx_uniform(const MxCtx &i)
{
    MxCtx c = i;
#pragma unroll
    for (int k = 0; k < 4; k++) c.pl[k] = mx_uni_ptr(i.pl[k]);
    c.cu = mx_uni_ptr(i.cu); c.cv = mx_uni_ptr(i.cv); c.cost_g = mx_uni_ptr(i.cost_g);
    c.mvpx = MX_UNI(i.mvpx); c.mvpy = MX_UNI(i.mvpy); c.sy = MX_UNI(i.sy); c.sc = MX_UNI(i.sc);
    c.bw = MX_UNI(i.bw); c.bh = MX_UNI(i.bh); c.fe_off = MX_UNI(i.fe_off); c.cfe_off = MX_UNI(i.cfe_off);
    return c;
}
__device__ __forceinline__ MeLimits mx_uniform(const MeLimits &i)
{
    MeLimits L;
    L.smin0 = MX_UNI(i.smin0); L.smax0 = MX_UNI(i.smax0); L.smin1 = MX_UNI(i.smin1); L.smax1 = MX_UNI(i.smax1);
    L.fmin0 = MX_UNI(i.fmin0); L.fmax0 = MX_UNI(i.fmax0); L.fmin1 = MX_UNI(i.fmin1); L.fmax1 = MX_UNI(i.fmax1);
    return L;
}
__device__ __forceinline__ MeOpts mx_uniform(const MeOpts &i)
{
    MeOpts o;
    o.method = MX_UNI(i.method); o.me_range = MX_UNI(i.me_range); o.subme = MX_UNI(i.subme); o.chroma_me = MX_UNI(i.chroma_me); o.sad_only = MX_UNI(i.sad_only);
    return o;
}

// X264_ME_UMH (me.c:306-447): the offsets of its fixed candidate groups, in the reference's evaluation order
static __constant__ signed char c_umh_tab[40][2] = {
    {0,-1},{0,1},{-1,0},{1,0},                                                   //  0: DIA1
    {0,-2},{-1,-1},{1,-1},{-2,0},{2,0},{-1,1},{1,1},{0,2},                       //  4: early-termination ring, me.c:340-343
    {-1,-2},{1,-2},{-2,-1},{2,-1},{-2,1},{2,1},{-1,2},{1,2},                     // 12: second ring, me.c:353-356
    {-2,-2},{-2,2},{2,-2},{2,2},                                                 // 20: 5x5 corners, me.c:401
    {-4,2},{-4,1},{-4,0},{-4,-1},{-4,-2},{4,-2},{4,-1},{4,0},{4,1},{4,2},{2,3},{0,4},{-2,3},{-2,-3},{0,-4},{2,-3}};   // 24: hex4, me.c:409-414
static __constant__ int c_umh_range_mul[4][4] = {{3, 3, 4, 4}, {3, 4, 4, 4}, {4, 4, 4, 5}, {4, 4, 5, 6}};
// One candidate stream of the UMH search around (omx, omy): CROSS(start, x_max, y_max) -- horizontal arm, then vertical arm,
// each "+i then -i" for i = start, start+2, .. with the reference's one-sided range tests -- followed by `rings` scaled
// copies of a c_umh_tab group (grid = the hexagon grid, whose points are range-tested).  Candidate n of the stream is a pure
// function of n, so every 8-lane group computes its own; a trip scores eight.
__device__ __forceinline__ void umh_stream(const MxCtx &c, const MeLimits &L, int omx, int omy, int start, int x_max, int y_max,
                                           int toff, int tlog, int rings, bool grid, int &bcost, int &bmx, int &bmy)
{
    const int nh = x_max > start ? ((x_max - start + 1) >> 1) * 2 : 0, nv = y_max > start ? ((y_max - start + 1) >> 1) * 2 : 0;
    const int total = nh + nv + (rings << tlog);
    const int g = c.lane >> 3;
    for (int base = 0; base < total; base += 8) {
        const int n = base + g;
        int dx = 0, dy = 0;
        bool ok = n < total;
        if (n < nh) {
            const int i = start + (n >> 1) * 2;
            if (n & 1) { dx = -i; ok = ok && omx - i >= L.fmin0; } else { dx = i; ok = ok && omx + i <= L.fmax0; }
        } else if (n < nh + nv) {
            const int m = n - nh, i = start + (m >> 1) * 2;
            if (m & 1) { dy = -i; ok = ok && omy - i >= L.fmin1; } else { dy = i; ok = ok && omy + i <= L.fmax1; }
        } else if (ok) {
            const int t = n - nh - nv, ring = (t >> tlog) + 1, e = toff + (t & ((1 << tlog) - 1));
            dx = (int)c_umh_tab[e][0] * ring; dy = (int)c_umh_tab[e][1] * ring;
            if (grid) ok = omx + dx >= L.fmin0 && omx + dx <= L.fmax0 && omy + dy >= L.fmin1 && omy + dy <= L.fmax1;
        }
        const int x = ok ? omx + dx : omx, y = ok ? omy + dy : omy;
        const int cost = sad_fpel8_lane(c, x, y) + c.lane_cost(x << 2, y << 2);
        const u32 key = mx_best_key<3>(cost, ok, c.lane);
        MX_TAKE(3, key, bcost, x, y, bmx, bmy);
    }
}

// x264_me_search_ref for one block.  c.mvpx / c.mvpy = the predictor (m->mvp).
// Returns m->cost (without the reference cost); thresh = p_halfpel_thresh or nullptr.
__device__ __forceinline__ int me_search_ref16(const MxCtx &c_in, const MeLimits &L_in, const MeOpts &o_in, const i16 *mvc, int n_mvc,
                                               int *thresh, int &out_mvx, int &out_mvy, int &out_cost_mv)
{
    MxCtx c = mx_uniform(c_in);
    c.patch_on = false;
    const MeLimits L = mx_uniform(L_in);
    const MeOpts o = mx_uniform(o_in);
    n_mvc = MX_UNI(n_mvc);
    const int lane = c.lane, g16 = lane >> 4, g8 = lane >> 3;
    const int satd = o.subme > 1 && !o.sad_only, mvpx = c.mvpx, mvpy = c.mvpy;
    int bmx = clip3(mvpx, L.fmin0 * 4, L.fmax0 * 4), bmy = clip3(mvpy, L.fmin1 * 4, L.fmax1 * 4);
    const int pmx = (bmx + 2) >> 2, pmy = (bmy + 2) >> 2;
    int bcost = MX_COST_MAX, bpx = 0, bpy = 0, bpcost = MX_COST_MAX;
#define INRANGE(x_, y_) ((x_) >= L.fmin0 && (x_) <= L.fmax0 && (y_) >= L.fmin1 && (y_) <= L.fmax1)
    if (o.subme >= 3) {
        // me.c:188-210: the predictor and every distinct non-zero candidate at quarter-pel precision (SAD)
        const int px = bmx, py = bmy;
        for (int base = 0; base < 1 + n_mvc; base += 8) {      // eight per trip: one trip for all but the longest lists
            const int k = base + g8;
            int x = px, y = py;
            bool ok = k == 0;
            if (k >= 1 && k <= n_mvc) {
                const int vx = (int)mvc[2 * (k - 1)], vy = (int)mvc[2 * (k - 1) + 1];
                if ((vx | vy) && (vx != (int)(i16)px || vy != (int)(i16)py)) {
                    ok = true; x = clip3(vx, L.fmin0 * 4, L.fmax0 * 4); y = clip3(vy, L.fmin1 * 4, L.fmax1 * 4);
                }
            }
            const int cost = sad_qpel8_lane(c, x, y) + c.lane_cost(x, y);
            const u32 key = mx_best_key<3>(cost, ok, lane);
            MX_TAKE(3, key, bpcost, x, y, bpx, bpy);
        }
        bmx = (bpx + 2) >> 2; bmy = (bpy + 2) >> 2;
        // COST_MV(bmx, bmy); COST_MV(0, 0)
        const int x = g16 == 0 ? bmx : 0, y = g16 == 0 ? bmy : 0;
        const int cost = sad_fpel16_lane(c, x, y) + c.lane_cost(x << 2, y << 2);
        const u32 key = mx_best_key<4>(cost, g16 < 2, lane);
        MX_TAKE(4, key, bcost, x, y, bmx, bmy);
    } else {
        // me.c:211-229: full-pel predictor (its mv cost taken out again), rounded candidates that differ from the running
        // best, then (0,0).  A candidate equal to the running best cannot win (same SAD, mv cost >= 0), so skipping it or
        // not gives the same walk; the predictor is first and always taken, at its SAD alone.
        const int total = n_mvc + 2;
        for (int base = 0; base < total; base += 4) {
            const int k = base + g16;
            int x = pmx, y = pmy;
            bool ok = k == 0;
            if (k >= 1 && k <= n_mvc) {
                const int ux = ((int)mvc[2 * (k - 1)] + 2) >> 2, uy = ((int)mvc[2 * (k - 1) + 1] + 2) >> 2;
                if (ux | uy) { ok = true; x = clip3(ux, L.fmin0, L.fmax0); y = clip3(uy, L.fmin1, L.fmax1); }
            } else if (k == n_mvc + 1) { ok = true; x = 0; y = 0; }
            const int cost = sad_fpel16_lane(c, x, y) + (k == 0 ? 0 : c.lane_cost(x << 2, y << 2));
            const u32 key = mx_best_key<4>(cost, ok, lane);
            MX_TAKE(4, key, bcost, x, y, bmx, bmy);
        }
    }
    bool do_hex = o.method == 1;
    int hex_range = o.me_range;
    if (o.method == 0) {
        // diamond, me.c:233-244
        int i = 0;
        do {
            const int ox = bmx, oyy = bmy;
            const int x = ox + mx_nib(MX_DIA_DX, g16), y = oyy + mx_nib(MX_DIA_DY, g16);
            const int cost = sad_fpel16_lane(c, x, y) + c.lane_cost(x << 2, y << 2);
            const u32 key = mx_best_key<4>(cost, true, lane);
            MX_TAKE(4, key, bcost, x, y, bmx, bmy);
            if (bmx == ox && bmy == oyy) break;
            if (!INRANGE(bmx, bmy)) break;
        } while (++i < o.me_range);
    } else if (o.method == 3) {
        // X264_ME_ESA, me.c:449-600.  The reference drops, row by row, every position whose ADS bound (sum of |differences of block
        // sums| + mv cost, which never exceeds SAD + mv cost) is not below the best cost so far; such a position could not have passed
        // COST_MV's strict '<' either, so its walk equals the plain raster scan of its own "#if 0" branch (pinned: scratch/cmp_esa.py) --
        // over min_x .. min_x + width - 1 with the width rounded up to a multiple of 4 (:456), which can stop one column short of max_x
        // or run up to three past it.  Eight positions of a row per trip; a row whose vertical mv cost alone reaches the best is skipped.
        const int min_x = max(bmx - o.me_range, L.fmin0), min_y = max(bmy - o.me_range, L.fmin1);
        const int max_x = min(bmx + o.me_range, L.fmax0), max_y = min(bmy + o.me_range, L.fmax1);
        const int width = (max_x - min_x + 3) & ~3;
        for (int my = min_y; my <= max_y; my++) {
            if (bcost <= c.cost1((my << 2) - mvpy)) continue;
            for (int x0 = 0; x0 < width; x0 += 8) {
                const int n = x0 + g8;
                const bool ok = n < width;
                const int x = min_x + (ok ? n : 0), y = my;
                const int cost = sad_fpel8_lane(c, x, y) + c.lane_cost(x << 2, y << 2);
                const u32 key = mx_best_key<3>(cost, ok, lane);
                MX_TAKE(3, key, bcost, x, y, bmx, bmy);
            }
        }
    } else if (o.method == 2) {
        // uneven-cross multi-hexagon, me.c:306-447.  A small state machine so that the candidate scorer is instantiated once.
        const int shift = (c.bw == 8) + (c.bh == 8) + 2 * ((c.bw == 4) + (c.bh == 4));   // x264_pixel_size_shift of the block: 0 1 1 2 3 3 4
        const bool tiny = c.bw == 4 && c.bh == 4;                         // "if(i_pixel == PIXEL_4x4) goto me_hex2", me.c:323
#define SAD_THRESH(v_) (bcost < ((v_) >> shift))
        const int ucost1 = bcost;
        int ucost2 = 0, cross_start = 1, omx = pmx, omy = pmy, ph = 0, et_range = 0;
        while (ph < 7) {
            int start = 0, x_max = 0, y_max = 0, toff = 0, tlog = 2, rings = 1;
            bool grid = false, run = true;
            switch (ph) {
            case 0: omx = pmx; omy = pmy; break;                            // DIA1 around the rounded predictor
            case 1: omx = 0; omy = 0; run = (pmx | pmy) != 0; break;        // ... and around (0,0)
            case 2: run = (bmx | bmy) && ((bmx - pmx) | (bmy - pmy)); omx = bmx; omy = bmy; break;
            case 3: toff = 4; tlog = 3; break;                              // early-termination ring
            case 4: et_range = (hex_range >> 1) | 1; start = 3; x_max = y_max = et_range; toff = 12; tlog = 3; break;
            case 5: start = cross_start; x_max = hex_range; y_max = hex_range >> 1; toff = 20; break;
            default: omx = bmx; omy = bmy; toff = 24; tlog = 4; rings = hex_range / 4 > 1 ? hex_range / 4 : 1; grid = true; break;
            }
            if (run) umh_stream(c, L, omx, omy, start, x_max, y_max, toff, tlog, rings, grid, bcost, bmx, bmy);
            switch (ph) {
            case 0: ph = 1; break;
            case 1: ucost2 = bcost; ph = tiny ? 9 : 2; break;
            case 2:
                if (bcost == ucost2) cross_start = 3;
                omx = bmx; omy = bmy;
                ph = (bcost == ucost2 && SAD_THRESH(2000)) ? 3 : 5;
                break;
            case 3:
                if (bcost == ucost1 && SAD_THRESH(500)) ph = 8;
                else ph = bcost == ucost2 ? 4 : 5;
                break;
            case 4:
                if (bcost == ucost2) ph = 8;
                else { cross_start = et_range + 2; ph = 5; }
                break;
            case 5: ph = 6; break;
            default: ph = 7; break;
            }
            if (ph == 5 && n_mvc) {
                // adaptive search range from the spread of the candidates, me.c:363-397
                int mvd, denom = 1;
                const int whole = c.bw == 16 && c.bh == 16;
#define MVC_(k_, d_) MX_UNI((int)mvc[2 * (k_) + (d_)])
                const int d0 = abs(mvpx - MVC_(0, 0)) + abs(mvpy - MVC_(0, 1));
                if (n_mvc == 1) mvd = whole ? 25 : d0;
                else {
                    denom = n_mvc - 1; mvd = 0;
                    if (!whole) { mvd = d0; denom++; }
                    for (int j = 0; j < n_mvc - 1; j++) mvd += abs(MVC_(j, 0) - MVC_(j + 1, 0)) + abs(MVC_(j, 1) - MVC_(j + 1, 1));
                }
#undef MVC_
                const int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
                const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
                hex_range = hex_range * c_umh_range_mul[mvd_ctx][sad_ctx] / 4;
            }
        }
#undef SAD_THRESH
        if ((ph == 7 && bmy <= L.fmax1) || ph == 9) do_hex = true;
    }
    if (do_hex) {
        // hexagon, me.c:246-305.  The first ring in one trip: (-2,0) (-1,2) (1,2) (2,0) (1,-2) (-1,-2) = hex2[1..6]
        int dir = -2;
        {
            const int hi = (g8 < 6 ? g8 : 0) + 1;                                                    // groups 6, 7 repeat 0 and are not ranked
            const int x = bmx + mx_nib(MX_HEX2_DX, hi), y = bmy + mx_nib(MX_HEX2_DY, hi);
            const int cost = sad_fpel8_lane(c, x, y) + c.lane_cost(x << 2, y << 2);
            const u32 key = mx_best_key<3>(cost, g8 < 6, lane);
            if ((key >> 3) < (u32)bcost) { bcost = (int)(key >> 3); dir = (int)(key & 7u); }
        }
        if (dir != -2) {
            bmx += mx_nib(MX_HEX2_DX, dir + 1); bmy += mx_nib(MX_HEX2_DY, dir + 1);
            for (int i = 1; i < hex_range / 2 && INRANGE(bmx, bmy); i++) {
                const int odir = dir + 1 >= 7 ? dir - 6 : dir + 1 <= 0 ? dir + 6 : dir;                 // mod6m1[dir + 1]
                const int idx = odir + (g16 < 3 ? g16 : 0);
                const int x = bmx + mx_nib(MX_HEX2_DX, idx), y = bmy + mx_nib(MX_HEX2_DY, idx);
                const int cost = sad_fpel16_lane(c, x, y) + c.lane_cost(x << 2, y << 2);
                const u32 key = mx_best_key<4>(cost, g16 < 3, lane);
                if ((key >> 3) >= (u32)bcost) break;
                bcost = (int)(key >> 3); dir = odir - 1 + (int)(key & 7u);
                bmx += mx_nib(MX_HEX2_DX, dir + 1); bmy += mx_nib(MX_HEX2_DY, dir + 1);
            }
        }
        // square refine: (0,-1) (0,1) (-1,0) (1,0) (-1,-1) (-1,1) (1,-1) (1,1) around the hexagon's best, me.c:300-304
        const int x = bmx + mx_nib(MX_NIB8(0, 0, -1, 1, -1, -1, 1, 1), g8), y = bmy + mx_nib(MX_NIB8(-1, 1, 0, 0, -1, 1, -1, 1), g8);
        const int cost = sad_fpel8_lane(c, x, y) + c.lane_cost(x << 2, y << 2);
        const u32 key = mx_best_key<3>(cost, true, lane);
        MX_TAKE(3, key, bcost, x, y, bmx, bmy);
    }
    int mvx, mvy, mcost;
    if (bpcost < bcost) { mvx = bpx; mvy = bpy; mcost = bpcost; }
    else { mvx = bmx << 2; mvy = bmy << 2; mcost = bcost; }
    out_cost_mv = c.cost(mvx, mvy);                                          // m->cost_mv, me.c:615
    if (bmx == pmx && bmy == pmy && o.subme < 3) mcost += out_cost_mv;
    if (o.subme >= 2) {
        // refine_subpel(.., b_refine_qpel = 0), me.c:680-778
        const int hpel = c_subpel_iters[o.subme][2], qpel = c_subpel_iters[o.subme][3];
        int bx = mvx, by = mvy, bc = mcost, odir = -1, bdir;
        bool early = false;
        if (c.has_patch) mx_load_patch(c, bx, by);
        if (hpel && o.subme < 3) {
            const int mx = clip3(mvpx, L.smin0, L.smax0), my = clip3(mvpy, L.smin1, L.smax1);
            if ((mx - bx) | (my - by)) {
                const int cost = __builtin_amdgcn_readlane(sad_qpel16_lane(c, mx, my), 0) + c.cost(mx, my);
                if (cost < bc) { bc = cost; bx = mx; by = my; }
            }
        }
        for (int i = hpel; i > 0; i--) {
            const int ox = bx, oyy = by;
            const int x = ox + 2 * mx_nib(MX_DIA_DX, g16), y = oyy + 2 * mx_nib(MX_DIA_DY, g16);
            const int cost = sad_qpel16_lane(c, x, y) + c.lane_cost(x, y);
            const u32 key = mx_best_key<4>(cost, true, lane);
            MX_TAKE(4, key, bc, x, y, bx, by);
            if (bx == ox && by == oyy) break;
        }
        if (by > L.smax1) by = L.smax1;
        bc = __builtin_amdgcn_readlane(subpel_sum16_lane(c, bx, by, satd, o.chroma_me), 0) + c.cost(bx, by);
        if (thresh) {
            const int th = MX_UNI(*thresh);
            if (((bc * 7) >> 3) > th) early = true;
            else if (bc < th) *thresh = bc;
        }
        if (!early) {
            bdir = -1;
            for (int i = qpel; i > 0; i--) {
                const int ox = bx, oyy = by;
                odir = bdir;
                const int x = ox + mx_nib(MX_DIA_DX, g16), y = oyy + mx_nib(MX_DIA_DY, g16);
                const int cost = subpel_sum16_lane(c, x, y, satd, o.chroma_me) + c.lane_cost(x, y);
                const u32 key = mx_best_key<4>(cost, (g16 ^ 1) != odir, lane);       // the direction just come from is not tried again
                if ((key >> 3) < (u32)bc) { bdir = (int)(key & 7u); MX_TAKE(4, key, bc, x, y, bx, by); }
                if (bx == ox && by == oyy) break;
            }
            if (by > L.smax1) {
                by = L.smax1;
                bc = __builtin_amdgcn_readlane(subpel_sum16_lane(c, bx, by, satd, o.chroma_me), 0) + c.cost(bx, by);
            }
            out_cost_mv = c.cost(bx, by);                                    // me.c:777
        }
        mvx = bx; mvy = by; mcost = bc;
    } else if (mvy > L.smax1) mvy = L.smax1;
#undef INRANGE
    out_mvx = mvx; out_mvy = mvy;
    return mcost;
}

// x264_me_refine_qpel -> refine_subpel(.., b_refine_qpel = 1) for one block (me.c:634-644, 680-778):
// half-pel diamond with SAD, quarter-pel diamond with mbcmp (all four directions every round).
// cost comes in without the reference cost; returns the refined cost.
__device__ __forceinline__ int me_refine_qpel16(const MxCtx &c_in, const MeLimits &L_in, const MeOpts &o_in, int cost_in, int &mvx, int &mvy)
{
    MxCtx c = mx_uniform(c_in);
    c.patch_on = false;
    const MeLimits L = mx_uniform(L_in);
    const MeOpts o = mx_uniform(o_in);
    const int lane = c.lane, g16 = lane >> 4;
    const int hpel = c_subpel_iters[o.subme][0], qpel = c_subpel_iters[o.subme][1], satd = o.subme > 1 && !o.sad_only;
    int bx = MX_UNI(mvx), by = MX_UNI(mvy), bc = MX_UNI(cost_in);
    if (c.has_patch) mx_load_patch(c, bx, by);
    if (hpel && o.subme < 3) {
        const int mx = clip3(c.mvpx, L.smin0, L.smax0), my = clip3(c.mvpy, L.smin1, L.smax1);
        if ((mx - bx) | (my - by)) {
            const int cst = __builtin_amdgcn_readlane(sad_qpel16_lane(c, mx, my), 0) + c.cost(mx, my);
            if (cst < bc) { bc = cst; bx = mx; by = my; }
        }
    }
    for (int i = hpel; i > 0; i--) {
        const int ox = bx, oyy = by;
        const int x = ox + 2 * mx_nib(MX_DIA_DX, g16), y = oyy + 2 * mx_nib(MX_DIA_DY, g16);
        const int cost = sad_qpel16_lane(c, x, y) + c.lane_cost(x, y);
        const u32 key = mx_best_key<4>(cost, true, lane);
        MX_TAKE(4, key, bc, x, y, bx, by);
        if (bx == ox && by == oyy) break;
    }
    for (int i = qpel; i > 0; i--) {
        const int ox = bx, oyy = by;
        const int x = ox + mx_nib(MX_DIA_DX, g16), y = oyy + mx_nib(MX_DIA_DY, g16);
        const int cost = subpel_sum16_lane(c, x, y, satd, o.chroma_me) + c.lane_cost(x, y);
        const u32 key = mx_best_key<4>(cost, true, lane);
        MX_TAKE(4, key, bc, x, y, bx, by);
        if (bx == ox && by == oyy) break;
    }
    if (by > L.smax1) {
        by = L.smax1;
        bc = __builtin_amdgcn_readlane(subpel_sum16_lane(c, bx, by, satd, o.chroma_me), 0) + c.cost(bx, by);
    }
    mvx = bx; mvy = by;
    return bc;
}
