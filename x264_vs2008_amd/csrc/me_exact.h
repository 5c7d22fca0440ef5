// me_exact.h -- the reference's motion search for one 16x16 block on one wavefront, shared by the
// per-frame search kernel (frame_me_exact.hip) and the macroblock sweep (frame_slice.hip).
//
// x264_me_search_ref + refine_subpel (R/encoder/me.c:156-778) and x264_me_refine_qpel (:634-644).
// The walk is data dependent and every step costs one round trip to L2/HBM for reference pixels, so
// the code is organised to make each round trip score as many candidates as the reference's own
// control flow allows, and replays the reference's sequential comparisons on the scores afterwards:
//   * predictor candidates are scored four at a time (16 lanes per candidate, one picture row per
//     lane = four v_sad_u8 on dwords re-aligned with v_alignbyte), then compared in list order;
//   * the hexagon's first ring and the square refine score 6 / 8 candidates per trip (8 lanes each);
//   * sub-pel rounds score up to four candidates, luma AND both chroma planes, in one trip: one lane
//     = one 8x4 block of one candidate (SATD on the reference's two-lanes-per-dword layout, the
//     per-block halving kept per block as pixel.c:214-253 does); the reference's "add chroma only
//     while still below the best" rule is applied to the finished sums, which gives the same result;
//   * all lanes carry the same scalar state (best vector, cost, direction): control flow is uniform.
// Reference pixels are read straight from HBM/L2 (the walk may start anywhere inside the mv limits);
// the source block and, in the sweep, the centre of the mv-cost table live in LDS.
#pragma once
#include "device_prims.h"

#define MX_COST_MAX (1 << 28)
#define MX_COST_LDS 1024          // half-width of the LDS copy of p_cost_mv (quarter-pels)

// R/encoder/me.c:34-50: subpel_iterations, hex2 (radius-2 hexagon with repeats), mod6m1
static __constant__ int c_subpel_iters[10][4] = {{0,0,0,0},{1,1,0,0},{0,1,1,0},{0,2,1,0},{0,2,1,1},{0,2,1,2},{0,0,2,2},{0,0,2,2},{0,0,4,10},{0,0,4,10}};
static __constant__ int c_hex2[8][2] = {{-1,-2},{-2,0},{-1,2},{1,2},{2,0},{1,-2},{-1,-2},{-2,0}};
static __constant__ int c_mod6m1[8] = {5,0,1,2,3,4,5,0};

// 16 / 8(+1) consecutive bytes at an arbitrary address as dwords (aligned loads + v_alignbyte)
__device__ __forceinline__ void load16u(const u8 *p, u32 o[4])
{
    const uintptr_t a = (uintptr_t)p;
    const u32 s = (u32)(a & 3);
    const u32 *q = (const u32 *)(a - s);
    u32 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
    o[0] = __builtin_amdgcn_alignbyte(w1, w0, s); o[1] = __builtin_amdgcn_alignbyte(w2, w1, s);
    o[2] = __builtin_amdgcn_alignbyte(w3, w2, s); o[3] = __builtin_amdgcn_alignbyte(w4, w3, s);
}
__device__ __forceinline__ void load9u(const u8 *p, u32 &o0, u32 &o1, u32 &o2)   // bytes 0..7 in o0,o1; byte 8 in the low byte of o2
{
    const uintptr_t a = (uintptr_t)p;
    const u32 s = (u32)(a & 3);
    const u32 *q = (const u32 *)(a - s);
    u32 w0 = q[0], w1 = q[1], w2 = q[2];
    o0 = __builtin_amdgcn_alignbyte(w1, w0, s); o1 = __builtin_amdgcn_alignbyte(w2, w1, s); o2 = __builtin_amdgcn_alignbyte(0u, w2, s);
}
// rounded byte-wise average of two dwords: (a + b + 1) >> 1 per byte, no carries across bytes
__device__ __forceinline__ u32 avg4(u32 a, u32 b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }
__device__ __forceinline__ int byte_of(u32 w, int k) { return (int)((w >> (8 * k)) & 255u); }

struct MxCtx {
    const u32 *fe;            // LDS: 16 rows x 4 dwords
    const u8 *fe_u, *fe_v;    // LDS: 8x8 each
    const u8 *pl[4];          // four half-pel planes at the macroblock origin
    const u8 *cu, *cv;        // chroma planes at the macroblock origin
    const i16 *cost_g;        // p_cost_mv, centred (global memory)
    const i16 *cost_l;        // LDS copy of cost_g[-MX_COST_LDS .. MX_COST_LDS], or nullptr
    int mvpx, mvpy;           // the predictor the costs are relative to
    int sy, sc, lane;
    // the block searched: 16x16, 16x8, 8x16 or 8x8 at (bx, by) inside the macroblock.  pl / cu / cv point at the block
    // (chroma at bx/2, by/2); fe_off = dword offset of the block in fe (by*4 + bx/4), cfe_off = byte offset in fe_u / fe_v
    int bw, bh, fe_off, cfe_off;
    __device__ __forceinline__ void set_block(int w, int h, int bx, int by) { bw = w; bh = h; fe_off = by * 4 + (bx >> 2); cfe_off = (by >> 1) * 8 + (bx >> 1); }
    __device__ __forceinline__ int cost1(int d) const
    {
        // d is wave-uniform: keep the looked-up cost in a scalar register
        return __builtin_amdgcn_readfirstlane((cost_l && (unsigned)(d + MX_COST_LDS) <= 2u * MX_COST_LDS) ? (int)cost_l[d + MX_COST_LDS] : (int)cost_g[d]);
    }
    __device__ __forceinline__ int cost(int mx, int my) const { return cost1(mx - mvpx) + cost1(my - mvpy); }   // p_cost_mvx[mx] + p_cost_mvy[my]
};

#define MX_PICK4(g_, v_) ((g_) == 0 ? (v_)[0] : (g_) == 1 ? (v_)[1] : (g_) == 2 ? (v_)[2] : (v_)[3])
// SAD 16x16 of up to four full-pel candidates (fx[k], fy[k]); result for candidate k in out[k]
__device__ __forceinline__ void sad_fpel4(const MxCtx &c, const int fx[4], const int fy[4], int out[4])
{
    const int g = c.lane >> 4, row = c.lane & 15;
    const int mx = MX_PICK4(g, fx), my = MX_PICK4(g, fy);
    int v = 0;
    if (row < c.bh) {
        const u8 *p = c.pl[0] + (ptrdiff_t)(my + row) * c.sy + mx;
        const u32 *f = c.fe + c.fe_off + 4 * row;
        u32 s;
        if (c.bw == 16) {
            u32 r[4];
            load16u(p, r);
            s = sad4(r[0], f[0], 0); s = sad4(r[1], f[1], s); s = sad4(r[2], f[2], s); s = sad4(r[3], f[3], s);
        } else {
            u32 r0, r1, t;
            load9u(p, r0, r1, t);
            s = sad4(r0, f[0], 0); s = sad4(r1, f[1], s);
        }
        v = (int)s;
    }
    v = row_sum16(v);
    out[0] = __builtin_amdgcn_readlane(v, 0); out[1] = __builtin_amdgcn_readlane(v, 16); out[2] = __builtin_amdgcn_readlane(v, 32); out[3] = __builtin_amdgcn_readlane(v, 48);
}
// the same for eight candidates: 8 lanes each, two picture rows per lane
__device__ __forceinline__ void sad_fpel8_at(const MxCtx &c, int mx, int my, int out[8])     // (mx, my): this lane group's candidate
{
    const int r = c.lane & 7;
    u32 s = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int row = r + 8 * half;
        if (row < c.bh) {
            const u8 *p = c.pl[0] + (ptrdiff_t)(my + row) * c.sy + mx;
            const u32 *f = c.fe + c.fe_off + 4 * row;
            if (c.bw == 16) {
                u32 a[4];
                load16u(p, a);
                s = sad4(a[0], f[0], s); s = sad4(a[1], f[1], s); s = sad4(a[2], f[2], s); s = sad4(a[3], f[3], s);
            } else {
                u32 a0, a1, t;
                load9u(p, a0, a1, t);
                s = sad4(a0, f[0], s); s = sad4(a1, f[1], s);
            }
        }
    }
    int v = (int)s;
    v = half_sum8(v);
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = __builtin_amdgcn_readlane(v, 8 * k);
}
__device__ __forceinline__ void sad_fpel8(const MxCtx &c, const int fx[8], const int fy[8], int out[8])
{
    const int g = c.lane >> 3;
    int mx = fx[0], my = fy[0];
#pragma unroll
    for (int k = 1; k < 8; k++) if (g == k) { mx = fx[k]; my = fy[k]; }
    sad_fpel8_at(c, mx, my, out);
}
// SAD 16x16 of up to four quarter-pel candidates through get_ref's blend (mc.c:181-202)
__device__ __forceinline__ void sad_qpel4(const MxCtx &c, const int qx[4], const int qy[4], int out[4])
{
    const int g = c.lane >> 4, row = c.lane & 15;
    const int mx = MX_PICK4(g, qx), my = MX_PICK4(g, qy);
    const int fx = mx & 3, fy = my & 3, idx = fy * 4 + fx;
    const ptrdiff_t base = (ptrdiff_t)((my >> 2) + row) * c.sy + (mx >> 2);
    int v = 0;
    if (row < c.bh) {
        const u8 *pa = c.pl[c_qpel_a[idx]] + base + (fy == 3) * c.sy, *pb = c.pl[c_qpel_b[idx]] + base + (fx == 3);
        const u32 *f = c.fe + c.fe_off + 4 * row;
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
            s = sad4(a0, f[0], 0); s = sad4(a1, f[1], s);
        }
        v = (int)s;
    }
    v = row_sum16(v);
    out[0] = __builtin_amdgcn_readlane(v, 0); out[1] = __builtin_amdgcn_readlane(v, 16); out[2] = __builtin_amdgcn_readlane(v, 32); out[3] = __builtin_amdgcn_readlane(v, 48);
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

// COST_MV_SATD's three sums (me.c:654-677) for up to four quarter-pel candidates in one trip:
// outL = mbcmp_unaligned[16x16] of the get_ref prediction, outU / outV = mbcmp[8x8] of mc_chroma.
// Lane = candidate (lane >> 4) x block: 0-7 luma 8x4 blocks, 8-9 U, 10-11 V.
__device__ __forceinline__ void me_subpel_costs4(const MxCtx &c, const int qx[4], const int qy[4], int satd, int chroma,
                                                 int outL[4], int outU[4], int outV[4])
{
    const int g = c.lane >> 4, j = c.lane & 15;
    const int mx = MX_PICK4(g, qx), my = MX_PICK4(g, qy);
    int v = 0;
    const int nbx = c.bw >> 3, n_luma = nbx * (c.bh >> 2), n_cunits = c.bh >> 3;      // 8x4 luma blocks; 4-row chroma units per plane
    if (j < n_luma) {
        const int bx = (nbx == 2 ? (j & 1) : 0) * 8, by = (nbx == 2 ? (j >> 1) : j) * 4;
        const int fx = mx & 3, fy = my & 3, idx = fy * 4 + fx;
        const ptrdiff_t base = (ptrdiff_t)((my >> 2) + by) * c.sy + (mx >> 2) + bx;
        const u8 *pa = c.pl[c_qpel_a[idx]] + base + (fy == 3) * c.sy, *pb = c.pl[c_qpel_b[idx]] + base + (fx == 3);
        u32 f[4][2], p[4][2];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            u32 t;
            load9u(pa + (ptrdiff_t)y * c.sy, p[y][0], p[y][1], t);
            if (idx & 5) {
                u32 b0, b1;
                load9u(pb + (ptrdiff_t)y * c.sy, b0, b1, t);
                p[y][0] = avg4(p[y][0], b0); p[y][1] = avg4(p[y][1], b1);
            }
            f[y][0] = c.fe[c.fe_off + (by + y) * 4 + (bx >> 2)]; f[y][1] = c.fe[c.fe_off + (by + y) * 4 + (bx >> 2) + 1];
        }
        v = blk8x4_cost(f, p, satd);
    } else if (chroma && j >= 8 && j < 12 && ((j - 8) & 1) < n_cunits) {
        // mbcmp[i_pixel + 3]: 8x8 / 8x4 chroma blocks are 8x4 units, 4x8 / 4x4 ones are 4x4 units (each halved on its own, pixel.c:235-253)
        const int by = ((j - 8) & 1) * 4, wide = c.bw == 16;
        const u8 *plane = j < 10 ? c.cu : c.cv, *fe = (j < 10 ? c.fe_u : c.fe_v) + c.cfe_off;
        const int dx = mx & 7, dy = my & 7;
        const int ca = (8 - dx) * (8 - dy), cb = dx * (8 - dy), cc = (8 - dx) * dy, cd = dx * dy;
        const u8 *s = plane + (ptrdiff_t)((my >> 3) + by) * c.sc + (mx >> 3);
        u32 r0[5], r1[5], r2[5];
#pragma unroll
        for (int y = 0; y < 5; y++) load9u(s + (ptrdiff_t)y * c.sc, r0[y], r1[y], r2[y]);
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
            const u32 *fr = (const u32 *)(fe + (by + y) * 8);
            p[y][0] = w0; f[y][0] = fr[0];
            p[y][1] = wide ? w1 : 0u; f[y][1] = wide ? fr[1] : 0u;      // a 4-wide unit: the right half contributes nothing
        }
        v = blk8x4_cost(f, p, satd);
    }
    const int s1 = v + dpp_mov<DPP_XOR1>(v);         // pairs: the two 8x4 halves of a chroma plane
    const int s4 = half_sum8(v);                     // the eight luma blocks
#pragma unroll
    for (int k = 0; k < 4; k++) { outL[k] = __builtin_amdgcn_readlane(s4, 16 * k); outU[k] = __builtin_amdgcn_readlane(s1, 16 * k + 8); outV[k] = __builtin_amdgcn_readlane(s1, 16 * k + 10); }
}
// COST_MV_SATD's running rule: chroma is added only while the sum is still below the best
__device__ __forceinline__ int me_satd_total(const MxCtx &c, int chroma, int L, int U, int V, int mx, int my, int limit)
{
    int cost = L + c.cost(mx, my);
    if (chroma && cost < limit) { cost += U; if (cost < limit) cost += V; }
    return cost;
}

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
struct MeOpts { int method, me_range, subme, chroma_me; };

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
// function of n, so every 8-lane group computes its own; a trip scores eight, and the reference's in-order strict '<'
// comparisons are replayed on the scores.
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
        int er[8];
        sad_fpel8_at(c, x, y, er);
        const unsigned long long okm = __ballot(ok);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((okm >> (8 * k)) & 1) {
                const int xs = __builtin_amdgcn_readlane(x, 8 * k), ys = __builtin_amdgcn_readlane(y, 8 * k);
                const int cost = er[k] + c.cost(xs << 2, ys << 2);
                if (cost < bcost) { bcost = cost; bmx = xs; bmy = ys; }
            }
    }
}

// x264_me_search_ref for PIXEL_16x16.  c.mvpx / c.mvpy = the predictor (m->mvp).
// Returns m->cost (without the reference cost); thresh = p_halfpel_thresh or nullptr.
__device__ int me_search_ref16(const MxCtx &c, const MeLimits &L, const MeOpts &o, const i16 *mvc, int n_mvc,
                               int *thresh, int &out_mvx, int &out_mvy, int &out_cost_mv)
{
    const int satd = o.subme > 1, mvpx = c.mvpx, mvpy = c.mvpy;
    int bmx = clip3(mvpx, L.fmin0 * 4, L.fmax0 * 4), bmy = clip3(mvpy, L.fmin1 * 4, L.fmax1 * 4);
    const int pmx = (bmx + 2) >> 2, pmy = (bmy + 2) >> 2;
    int bcost = MX_COST_MAX, bpx = 0, bpy = 0, bpcost = MX_COST_MAX;
    int cx[4], cy[4], res[4];
#define INRANGE(x_, y_) ((x_) >= L.fmin0 && (x_) <= L.fmax0 && (y_) >= L.fmin1 && (y_) <= L.fmax1)
    if (o.subme >= 3) {
        // me.c:188-210: the predictor and every distinct non-zero candidate at quarter-pel precision (SAD)
        const int px = bmx, py = bmy;
        for (int base = 0; base < 1 + n_mvc; base += 4) {
            bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = base + j;
                ok[j] = false; cx[j] = px; cy[j] = py;
                if (k == 0) ok[j] = true;
                else if (k <= n_mvc) {
                    const int vx = __builtin_amdgcn_readfirstlane((int)mvc[2 * (k - 1)]), vy = __builtin_amdgcn_readfirstlane((int)mvc[2 * (k - 1) + 1]);
                    if ((vx | vy) && (vx != (int)(i16)px || vy != (int)(i16)py)) {
                        ok[j] = true; cx[j] = clip3(vx, L.fmin0 * 4, L.fmax0 * 4); cy[j] = clip3(vy, L.fmin1 * 4, L.fmax1 * 4);
                    }
                }
            }
            sad_qpel4(c, cx, cy, res);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ok[j]) { const int cost = res[j] + c.cost(cx[j], cy[j]); if (cost < bpcost) { bpcost = cost; bpx = cx[j]; bpy = cy[j]; } }
        }
        bmx = (bpx + 2) >> 2; bmy = (bpy + 2) >> 2;
        // COST_MV(bmx, bmy); COST_MV(0, 0)
        cx[0] = bmx; cy[0] = bmy; cx[1] = cx[2] = cx[3] = 0; cy[1] = cy[2] = cy[3] = 0;
        sad_fpel4(c, cx, cy, res);
        { const int tx = bmx, ty = bmy, c0 = res[0] + c.cost(tx << 2, ty << 2); if (c0 < bcost) { bcost = c0; bmx = tx; bmy = ty; } }
        { const int c1 = res[1] + c.cost(0, 0); if (c1 < bcost) { bcost = c1; bmx = 0; bmy = 0; } }
    } else {
        // me.c:211-229: full-pel predictor (its mv cost taken out again), rounded candidates, then (0,0)
        const int total = n_mvc + 2;
        for (int base = 0; base < total; base += 4) {
            int kind[4];                             // 0 none, 1 predictor, 2 candidate, 3 zero
            int ux[4], uy[4];                        // unclipped candidate (the reference tests these against the running best)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = base + j;
                kind[j] = 0; cx[j] = pmx; cy[j] = pmy; ux[j] = uy[j] = 0;
                if (k == 0) kind[j] = 1;
                else if (k <= n_mvc) {
                    ux[j] = (__builtin_amdgcn_readfirstlane((int)mvc[2 * (k - 1)]) + 2) >> 2; uy[j] = (__builtin_amdgcn_readfirstlane((int)mvc[2 * (k - 1) + 1]) + 2) >> 2;
                    if (ux[j] | uy[j]) { kind[j] = 2; cx[j] = clip3(ux[j], L.fmin0, L.fmax0); cy[j] = clip3(uy[j], L.fmin1, L.fmax1); }
                } else if (k == n_mvc + 1) { kind[j] = 3; cx[j] = 0; cy[j] = 0; }
            }
            sad_fpel4(c, cx, cy, res);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (kind[j] == 0) continue;
                if (kind[j] == 2 && !((ux[j] - bmx) | (uy[j] - bmy))) continue;
                const int cost = res[j] + c.cost(cx[j] << 2, cy[j] << 2);
                if (cost < bcost) { bcost = cost; bmx = cx[j]; bmy = cy[j]; }
                if (kind[j] == 1) bcost -= c.cost(pmx << 2, pmy << 2);
            }
        }
    }
    // four candidates around (ox, oy) in the given order, strict '<' updates (COST_MV_X4)
#define X4(ox_, oy_, ax, ay, bx_, by_, cx_, cy_, dx_, dy_) do { \
    cx[0] = (ox_) + (ax); cy[0] = (oy_) + (ay); cx[1] = (ox_) + (bx_); cy[1] = (oy_) + (by_); \
    cx[2] = (ox_) + (cx_); cy[2] = (oy_) + (cy_); cx[3] = (ox_) + (dx_); cy[3] = (oy_) + (dy_); \
    sad_fpel4(c, cx, cy, res); \
    for (int k_ = 0; k_ < 4; k_++) { int cost_ = res[k_] + c.cost(cx[k_] << 2, cy[k_] << 2); \
        if (cost_ < bcost) { bcost = cost_; bmx = cx[k_]; bmy = cy[k_]; } } } while (0)
    if (o.method == 0) {
        int i = 0;
        do {
            const int ox = bmx, oyy = bmy;
            X4(ox, oyy, 0, -1, 0, 1, -1, 0, 1, 0);
            if (bmx == ox && bmy == oyy) break;
            if (!INRANGE(bmx, bmy)) break;
        } while (++i < o.me_range);
    }
    bool do_hex = o.method == 1;
    int hex_range = o.me_range;
    if (o.method == 2) {
        // uneven-cross multi-hexagon, me.c:306-447.  A small state machine so that the candidate scorer is instantiated once.
        const int shift = (c.bw == 8) + (c.bh == 8);                      // x264_pixel_size_shift of the block
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
            case 1: ucost2 = bcost; ph = 2; break;
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
#define MVC_(k_, d_) __builtin_amdgcn_readfirstlane((int)mvc[2 * (k_) + (d_)])
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
        if (ph == 7 && bmy <= L.fmax1) do_hex = true;
    }
    if (do_hex) {
        int dir = -2, ex[8], ey[8], er[8];
        // the first ring in one trip: (-2,0) (-1,2) (1,2) (2,0) (1,-2) (-1,-2), me.c:254-262
        ex[0] = bmx - 2; ey[0] = bmy; ex[1] = bmx - 1; ey[1] = bmy + 2; ex[2] = bmx + 1; ey[2] = bmy + 2; ex[3] = bmx + 2; ey[3] = bmy;
        ex[4] = bmx + 1; ey[4] = bmy - 2; ex[5] = bmx - 1; ey[5] = bmy - 2; ex[6] = ex[7] = bmx; ey[6] = ey[7] = bmy;
        sad_fpel8(c, ex, ey, er);
#pragma unroll
        for (int k = 0; k < 6; k++) { const int cost = er[k] + c.cost(ex[k] << 2, ey[k] << 2); if (cost < bcost) { bcost = cost; dir = k; } }
        if (dir != -2) {
            bmx += c_hex2[dir + 1][0]; bmy += c_hex2[dir + 1][1];
            for (int i = 1; i < hex_range / 2 && INRANGE(bmx, bmy); i++) {
                const int odir = c_mod6m1[dir + 1];
#pragma unroll
                for (int k = 0; k < 3; k++) { cx[k] = bmx + c_hex2[odir + k][0]; cy[k] = bmy + c_hex2[odir + k][1]; }
                cx[3] = cx[0]; cy[3] = cy[0];
                sad_fpel4(c, cx, cy, res);
                dir = -2;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int cost = res[k] + c.cost(cx[k] << 2, cy[k] << 2);
                    if (cost < bcost) { bcost = cost; dir = odir - 1 + k; }
                }
                if (dir == -2) break;
                bmx += c_hex2[dir + 1][0]; bmy += c_hex2[dir + 1][1];
            }
        }
        // square refine: (0,-1) (0,1) (-1,0) (1,0) (-1,-1) (-1,1) (1,-1) (1,1) around the hexagon's best, me.c:300-304
        const int ox = bmx, oyy = bmy;
        ex[0] = ox; ey[0] = oyy - 1; ex[1] = ox; ey[1] = oyy + 1; ex[2] = ox - 1; ey[2] = oyy; ex[3] = ox + 1; ey[3] = oyy;
        ex[4] = ox - 1; ey[4] = oyy - 1; ex[5] = ox - 1; ey[5] = oyy + 1; ex[6] = ox + 1; ey[6] = oyy - 1; ex[7] = ox + 1; ey[7] = oyy + 1;
        sad_fpel8(c, ex, ey, er);
#pragma unroll
        for (int k = 0; k < 8; k++) { const int cost = er[k] + c.cost(ex[k] << 2, ey[k] << 2); if (cost < bcost) { bcost = cost; bmx = ex[k]; bmy = ey[k]; } }
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
        int cl[4], cu[4], cv[4];
        bool early = false;
        if (hpel && o.subme < 3) {
            int mx = clip3(mvpx, L.smin0, L.smax0), my = clip3(mvpy, L.smin1, L.smax1);
            if ((mx - bx) | (my - by)) {
                cx[0] = cx[1] = cx[2] = cx[3] = mx; cy[0] = cy[1] = cy[2] = cy[3] = my;
                sad_qpel4(c, cx, cy, res);
                int cost = res[0] + c.cost(mx, my);
                if (cost < bc) { bc = cost; bx = mx; by = my; }
            }
        }
        for (int i = hpel; i > 0; i--) {
            const int ox = bx, oyy = by;
            cx[0] = ox; cy[0] = oyy - 2; cx[1] = ox; cy[1] = oyy + 2; cx[2] = ox - 2; cy[2] = oyy; cx[3] = ox + 2; cy[3] = oyy;
            sad_qpel4(c, cx, cy, res);
            int c0 = res[0] + c.cost(ox, oyy - 2), c1 = res[1] + c.cost(ox, oyy + 2);
            int c2 = res[2] + c.cost(ox - 2, oyy), c3 = res[3] + c.cost(ox + 2, oyy);
            if (c0 < bc) { bc = c0; by = oyy - 2; }
            if (c1 < bc) { bc = c1; by = oyy + 2; }
            if (c2 < bc) { bc = c2; bx = ox - 2; by = oyy; }
            if (c3 < bc) { bc = c3; bx = ox + 2; by = oyy; }
            if (bx == ox && by == oyy) break;
        }
        if (by > L.smax1) by = L.smax1;
        bc = MX_COST_MAX;
        cx[0] = cx[1] = cx[2] = cx[3] = bx; cy[0] = cy[1] = cy[2] = cy[3] = by;
        me_subpel_costs4(c, cx, cy, satd, o.chroma_me, cl, cu, cv);
        { int cost = me_satd_total(c, o.chroma_me, cl[0], cu[0], cv[0], bx, by, bc); if (cost < bc) bc = cost; }
        if (thresh) {
            if (((bc * 7) >> 3) > *thresh) early = true;
            else if (bc < *thresh) *thresh = bc;
        }
        if (!early) {
            bdir = -1;
            for (int i = qpel; i > 0; i--) {
                const int ox = bx, oyy = by;
                odir = bdir;
                cx[0] = ox; cy[0] = oyy - 1; cx[1] = ox; cy[1] = oyy + 1; cx[2] = ox - 1; cy[2] = oyy; cx[3] = ox + 1; cy[3] = oyy;
                me_subpel_costs4(c, cx, cy, satd, o.chroma_me, cl, cu, cv);
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    if ((d ^ 1) == odir) continue;
                    const int cost = me_satd_total(c, o.chroma_me, cl[d], cu[d], cv[d], cx[d], cy[d], bc);
                    if (cost < bc) { bc = cost; bx = cx[d]; by = cy[d]; bdir = d; }
                }
                if (bx == ox && by == oyy) break;
            }
            if (by > L.smax1) {
                by = L.smax1; bc = MX_COST_MAX;
                cx[0] = cx[1] = cx[2] = cx[3] = bx; cy[0] = cy[1] = cy[2] = cy[3] = by;
                me_subpel_costs4(c, cx, cy, satd, o.chroma_me, cl, cu, cv);
                int cost = me_satd_total(c, o.chroma_me, cl[0], cu[0], cv[0], bx, by, bc); if (cost < bc) bc = cost;
            }
            out_cost_mv = c.cost(bx, by);                                    // me.c:777
        }
        mvx = bx; mvy = by; mcost = bc;
    } else if (mvy > L.smax1) mvy = L.smax1;
#undef X4
#undef INRANGE
    out_mvx = mvx; out_mvy = mvy;
    return mcost;
}

// x264_me_refine_qpel -> refine_subpel(.., b_refine_qpel = 1) for PIXEL_16x16 (me.c:634-644, 680-778):
// half-pel diamond with SAD, quarter-pel diamond with mbcmp (all four directions every round).
// cost comes in without the reference cost; returns the refined cost.
__device__ int me_refine_qpel16(const MxCtx &c, const MeLimits &L, const MeOpts &o, int cost, int &mvx, int &mvy)
{
    const int hpel = c_subpel_iters[o.subme][0], qpel = c_subpel_iters[o.subme][1], satd = o.subme > 1;
    int bx = mvx, by = mvy, bc = cost, cx[4], cy[4], res[4], cl[4], cu[4], cv[4];
    if (hpel && o.subme < 3) {
        int mx = clip3(c.mvpx, L.smin0, L.smax0), my = clip3(c.mvpy, L.smin1, L.smax1);
        if ((mx - bx) | (my - by)) {
            cx[0] = cx[1] = cx[2] = cx[3] = mx; cy[0] = cy[1] = cy[2] = cy[3] = my;
            sad_qpel4(c, cx, cy, res);
            int cst = res[0] + c.cost(mx, my);
            if (cst < bc) { bc = cst; bx = mx; by = my; }
        }
    }
    for (int i = hpel; i > 0; i--) {
        const int ox = bx, oyy = by;
        cx[0] = ox; cy[0] = oyy - 2; cx[1] = ox; cy[1] = oyy + 2; cx[2] = ox - 2; cy[2] = oyy; cx[3] = ox + 2; cy[3] = oyy;
        sad_qpel4(c, cx, cy, res);
        int c0 = res[0] + c.cost(ox, oyy - 2), c1 = res[1] + c.cost(ox, oyy + 2);
        int c2 = res[2] + c.cost(ox - 2, oyy), c3 = res[3] + c.cost(ox + 2, oyy);
        if (c0 < bc) { bc = c0; by = oyy - 2; }
        if (c1 < bc) { bc = c1; by = oyy + 2; }
        if (c2 < bc) { bc = c2; bx = ox - 2; by = oyy; }
        if (c3 < bc) { bc = c3; bx = ox + 2; by = oyy; }
        if (bx == ox && by == oyy) break;
    }
    for (int i = qpel; i > 0; i--) {
        const int ox = bx, oyy = by;
        cx[0] = ox; cy[0] = oyy - 1; cx[1] = ox; cy[1] = oyy + 1; cx[2] = ox - 1; cy[2] = oyy; cx[3] = ox + 1; cy[3] = oyy;
        me_subpel_costs4(c, cx, cy, satd, o.chroma_me, cl, cu, cv);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int cst = me_satd_total(c, o.chroma_me, cl[d], cu[d], cv[d], cx[d], cy[d], bc);
            if (cst < bc) { bc = cst; bx = cx[d]; by = cy[d]; }
        }
        if (bx == ox && by == oyy) break;
    }
    if (by > L.smax1) {
        by = L.smax1; bc = MX_COST_MAX;
        cx[0] = cx[1] = cx[2] = cx[3] = bx; cy[0] = cy[1] = cy[2] = cy[3] = by;
        me_subpel_costs4(c, cx, cy, satd, o.chroma_me, cl, cu, cv);
        int cst = me_satd_total(c, o.chroma_me, cl[0], cu[0], cv[0], bx, by, bc); if (cst < bc) bc = cst;
    }
    mvx = bx; mvy = by;
    return bc;
}
