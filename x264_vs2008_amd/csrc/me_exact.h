// me_exact.h -- the reference's motion search for one 16x16 block on one wavefront, shared by the
// per-frame search kernel (frame_me_exact.hip) and the macroblock sweep (frame_slice.hip).
//
// x264_me_search_ref + refine_subpel (R/encoder/me.c:156-778) and x264_me_refine_qpel (:634-644),
// run exactly in the reference's candidate order:
//   * up to four candidates are scored at once (COST_MV_X4 / _X3_DIR): 16 lanes per candidate,
//     one picture row per lane = four v_sad_u8 on dwords re-aligned with v_alignbyte, reduced
//     with four shuffles inside the 16-lane group;
//   * sub-pel candidates blend two of the four half-pel planes on the fly (get_ref, mc.c:181-202);
//   * SATD uses 32 lanes = 8 (8x4 blocks) x 4 rows: horizontal butterflies in registers on the
//     reference's two-lanes-per-dword layout, vertical butterflies with two shuffles, the
//     per-block halving kept per block as the reference does (pixel.c:214-253);
//   * all lanes carry the same scalar state (best vector, cost, direction), so control flow is
//     wave-uniform.
// Reference pixels are read straight from HBM/L2 (the walk may start anywhere inside the mv
// limits, so no window is staged); the source block lives in LDS.
#pragma once
#include "device_prims.h"

#define MX_COST_MAX (1 << 28)

// 16 consecutive bytes at an arbitrary address as four dwords (aligned loads + v_alignbyte)
__device__ __forceinline__ void load16u(const u8 *p, u32 o[4])
{
    const uintptr_t a = (uintptr_t)p;
    const u32 s = (u32)(a & 3);
    const u32 *q = (const u32 *)(a - s);
    u32 w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
    o[0] = __builtin_amdgcn_alignbyte(w1, w0, s); o[1] = __builtin_amdgcn_alignbyte(w2, w1, s);
    o[2] = __builtin_amdgcn_alignbyte(w3, w2, s); o[3] = __builtin_amdgcn_alignbyte(w4, w3, s);
}
// rounded byte-wise average of two dwords: (a + b + 1) >> 1 per byte, no carries across bytes
__device__ __forceinline__ u32 avg4(u32 a, u32 b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }

struct MxCtx {
    const u32 *fe;            // LDS: 16 rows x 4 dwords
    const u8 *fe_u, *fe_v;    // LDS: 8x8 each
    const u8 *pl[4];          // four half-pel planes at the macroblock origin
    const u8 *cu, *cv;        // chroma planes at the macroblock origin
    const i16 *cmx, *cmy;     // cost tables offset by the predictor
    int sy, sc, lane;
};

// SAD 16x16 of up to four full-pel candidates (fx[k], fy[k]); result for candidate k in out[k]
__device__ __forceinline__ void sad_fpel4(const MxCtx &c, const int fx[4], const int fy[4], int out[4])
{
    const int g = c.lane >> 4, row = c.lane & 15;
    const int mx = g == 0 ? fx[0] : g == 1 ? fx[1] : g == 2 ? fx[2] : fx[3];
    const int my = g == 0 ? fy[0] : g == 1 ? fy[1] : g == 2 ? fy[2] : fy[3];
    u32 r[4];
    load16u(c.pl[0] + (ptrdiff_t)(my + row) * c.sy + mx, r);
    const u32 *f = c.fe + 4 * row;
    u32 s = sad4(r[0], f[0], 0); s = sad4(r[1], f[1], s); s = sad4(r[2], f[2], s); s = sad4(r[3], f[3], s);
    int v = (int)s;
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    out[0] = __shfl(v, 0, 64); out[1] = __shfl(v, 16, 64); out[2] = __shfl(v, 32, 64); out[3] = __shfl(v, 48, 64);
}
// SAD 16x16 of up to four quarter-pel candidates through get_ref's blend
__device__ __forceinline__ void sad_qpel4(const MxCtx &c, const int qx[4], const int qy[4], int out[4])
{
    const int g = c.lane >> 4, row = c.lane & 15;
    const int mx = g == 0 ? qx[0] : g == 1 ? qx[1] : g == 2 ? qx[2] : qx[3];
    const int my = g == 0 ? qy[0] : g == 1 ? qy[1] : g == 2 ? qy[2] : qy[3];
    const int fx = mx & 3, fy = my & 3, idx = fy * 4 + fx;
    const ptrdiff_t base = (ptrdiff_t)((my >> 2) + row) * c.sy + (mx >> 2);
    u32 a[4];
    load16u(c.pl[c_qpel_a[idx]] + base + (fy == 3) * c.sy, a);
    if (idx & 5) {
        u32 b[4];
        load16u(c.pl[c_qpel_b[idx]] + base + (fx == 3), b);
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = avg4(a[k], b[k]);
    }
    const u32 *f = c.fe + 4 * row;
    u32 s = sad4(a[0], f[0], 0); s = sad4(a[1], f[1], s); s = sad4(a[2], f[2], s); s = sad4(a[3], f[3], s);
    int v = (int)s;
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    out[0] = __shfl(v, 0, 64); out[1] = __shfl(v, 16, 64); out[2] = __shfl(v, 32, 64); out[3] = __shfl(v, 48, 64);
}
// vertical half of the 8x4 SATD: lanes l, l^1, l^2 hold rows of one block; t[] = this row's
// horizontally transformed packed words.  Returns the block's SATD (same value in its 4 lanes).
__device__ __forceinline__ int satd_rows4(u32 t0, u32 t1, u32 t2, u32 t3, int lane)
{
    u32 t[4] = {t0, t1, t2, t3}, acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 v = t[k], o = (u32)__shfl_xor((int)v, 1, 64);
        v = (lane & 1) ? o - v : v + o;                 // rows (0,1) and (2,3): sum / difference
        o = (u32)__shfl_xor((int)v, 2, 64);
        v = (lane & 2) ? o - v : v + o;                 // second butterfly level
        acc += lanes_abs(v);
    }
    acc += (u32)__shfl_xor((int)acc, 1, 64);
    acc += (u32)__shfl_xor((int)acc, 2, 64);
    return (int)(((acc & 0xffffu) + (acc >> 16)) >> 1);
}
// SATD 16x16 of one quarter-pel candidate (mbcmp_unaligned at subme > 1)
__device__ __forceinline__ int satd_qpel(const MxCtx &c, int mx, int my)
{
    int blk_satd = 0;
    if (c.lane < 32) {
        const int blk = c.lane >> 2, r = c.lane & 3, bx = (blk & 1) * 8, y = (blk >> 1) * 4 + r;
        const int fx = mx & 3, fy = my & 3, idx = fy * 4 + fx;
        const ptrdiff_t base = (ptrdiff_t)((my >> 2) + y) * c.sy + (mx >> 2) + bx;
        const u8 *pa = c.pl[c_qpel_a[idx]] + base + (fy == 3) * c.sy, *pb = c.pl[c_qpel_b[idx]] + base + (fx == 3);
        const u8 *f = (const u8 *)c.fe + y * 16 + bx;
        int d[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
            int p = (idx & 5) ? ((int)pa[x] + (int)pb[x] + 1) >> 1 : (int)pa[x];
            d[x] = (int)f[x] - p;
        }
        u32 p0 = (u32)d[0] + ((u32)d[4] << 16), p1 = (u32)d[1] + ((u32)d[5] << 16);
        u32 p2 = (u32)d[2] + ((u32)d[6] << 16), p3 = (u32)d[3] + ((u32)d[7] << 16);
        u32 t0, t1, t2, t3;
        wht4(t0, t1, t2, t3, p0, p1, p2, p3);
        blk_satd = satd_rows4(t0, t1, t2, t3, c.lane);
        if (r != 0) blk_satd = 0;
    } else {
        // lanes 32-63 still take part in the shuffles of satd_rows4's callers below
    }
    return wave_sum(blk_satd);
}
// SATD 8x8 of a chroma plane predicted with mc_chroma at qpel vector (mx,my) (mbcmp[PIXEL_8x8])
__device__ __forceinline__ int satd_chroma(const MxCtx &c, const u8 *plane, const u8 *fe, int mx, int my)
{
    int blk_satd = 0;
    if (c.lane < 8) {
        const int blk = c.lane >> 2, r = c.lane & 3, y = blk * 4 + r;
        const int dx = mx & 7, dy = my & 7;
        const int ca = (8 - dx) * (8 - dy), cb = dx * (8 - dy), cc = (8 - dx) * dy, cd = dx * dy;
        const u8 *s = plane + (ptrdiff_t)((my >> 3) + y) * c.sc + (mx >> 3);
        int d[8];
#pragma unroll
        for (int x = 0; x < 8; x++) {
            int p = (ca * s[x] + cb * s[x + 1] + cc * s[c.sc + x] + cd * s[c.sc + x + 1] + 32) >> 6;
            d[x] = (int)fe[y * 8 + x] - p;
        }
        u32 p0 = (u32)d[0] + ((u32)d[4] << 16), p1 = (u32)d[1] + ((u32)d[5] << 16);
        u32 p2 = (u32)d[2] + ((u32)d[6] << 16), p3 = (u32)d[3] + ((u32)d[7] << 16);
        u32 t0, t1, t2, t3;
        wht4(t0, t1, t2, t3, p0, p1, p2, p3);
        blk_satd = satd_rows4(t0, t1, t2, t3, c.lane);
        if (r != 0) blk_satd = 0;
    }
    return wave_sum(blk_satd);
}
// SAD 8x8 of a chroma plane predicted with mc_chroma (mbcmp[PIXEL_8x8] at subme <= 1)
__device__ __forceinline__ int sad_chroma(const MxCtx &c, const u8 *plane, const u8 *fe, int mx, int my)
{
    const int x = c.lane & 7, y = c.lane >> 3;
    const int dx = mx & 7, dy = my & 7;
    const int ca = (8 - dx) * (8 - dy), cb = dx * (8 - dy), cc = (8 - dx) * dy, cd = dx * dy;
    const u8 *s = plane + (ptrdiff_t)((my >> 3) + y) * c.sc + (mx >> 3) + x;
    int p = (ca * s[0] + cb * s[1] + cc * s[c.sc] + cd * s[c.sc + 1] + 32) >> 6;
    return wave_sum(iabs((int)fe[y * 8 + x] - p));
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

// COST_MV_SATD (me.c:654-677): luma through mbcmp_unaligned (SATD above subme 1), then the chroma
// planes while the sum is still below the best
__device__ __forceinline__ int me_cost_satd(const MxCtx &c, int satd, int chroma_me, int mx, int my, int limit)
{
    int cost;
    if (satd) cost = satd_qpel(c, mx, my);
    else { int qx[4] = {mx, mx, mx, mx}, qy[4] = {my, my, my, my}, res[4]; sad_qpel4(c, qx, qy, res); cost = res[0]; }
    cost += c.cmx[mx] + c.cmy[my];
    if (chroma_me && cost < limit) {
        cost += satd ? satd_chroma(c, c.cu, c.fe_u, mx, my) : sad_chroma(c, c.cu, c.fe_u, mx, my);
        if (cost < limit) cost += satd ? satd_chroma(c, c.cv, c.fe_v, mx, my) : sad_chroma(c, c.cv, c.fe_v, mx, my);
    }
    return cost;
}

// x264_me_search_ref for PIXEL_16x16.  c.cmx / c.cmy must already be offset by the predictor.
// Returns m->cost (without the reference cost); thresh = p_halfpel_thresh or nullptr.
__device__ int me_search_ref16(const MxCtx &c, const MeLimits &L, const MeOpts &o, int mvpx, int mvpy, const i16 *mvc, int n_mvc,
                               int *thresh, int &out_mvx, int &out_mvy, int &out_cost_mv)
{
    const int satd = o.subme > 1;
    int bmx = clip3(mvpx, L.fmin0 * 4, L.fmax0 * 4), bmy = clip3(mvpy, L.fmin1 * 4, L.fmax1 * 4);
    const int pmx = (bmx + 2) >> 2, pmy = (bmy + 2) >> 2;
    int bcost = MX_COST_MAX, bpx = 0, bpy = 0, bpcost = MX_COST_MAX;
    int cx[4], cy[4], res[4];
#define FPEL1(mx_, my_) do { cx[0] = cx[1] = cx[2] = cx[3] = (mx_); cy[0] = cy[1] = cy[2] = cy[3] = (my_); sad_fpel4(c, cx, cy, res); \
    int cost_ = res[0] + c.cmx[(mx_) << 2] + c.cmy[(my_) << 2]; if (cost_ < bcost) { bcost = cost_; bmx = (mx_); bmy = (my_); } } while (0)
#define INRANGE(x_, y_) ((x_) >= L.fmin0 && (x_) <= L.fmax0 && (y_) >= L.fmin1 && (y_) <= L.fmax1)
    if (o.subme >= 3) {
        const int px = bmx, py = bmy;
        cx[0] = cx[1] = cx[2] = cx[3] = px; cy[0] = cy[1] = cy[2] = cy[3] = py;
        sad_qpel4(c, cx, cy, res);
        { int cost = res[0] + c.cmx[px] + c.cmy[py]; if (cost < bpcost) { bpcost = cost; bpx = px; bpy = py; } }
        for (int i = 0; i < n_mvc; i++) {
            int vx = mvc[2 * i], vy = mvc[2 * i + 1];
            if ((vx | vy) && (vx != (int)(i16)px || vy != (int)(i16)py)) {
                int mx = clip3(vx, L.fmin0 * 4, L.fmax0 * 4), my = clip3(vy, L.fmin1 * 4, L.fmax1 * 4);
                cx[0] = cx[1] = cx[2] = cx[3] = mx; cy[0] = cy[1] = cy[2] = cy[3] = my;
                sad_qpel4(c, cx, cy, res);
                int cost = res[0] + c.cmx[mx] + c.cmy[my];
                if (cost < bpcost) { bpcost = cost; bpx = mx; bpy = my; }
            }
        }
        bmx = (bpx + 2) >> 2; bmy = (bpy + 2) >> 2;
        { int tx = bmx, ty = bmy; FPEL1(tx, ty); }
    } else {
        FPEL1(pmx, pmy);
        bcost -= c.cmx[pmx << 2] + c.cmy[pmy << 2];
        for (int i = 0; i < n_mvc; i++) {
            int mx = (mvc[2 * i] + 2) >> 2, my = (mvc[2 * i + 1] + 2) >> 2;
            if ((mx | my) && ((mx - bmx) | (my - bmy))) {
                mx = clip3(mx, L.fmin0, L.fmax0); my = clip3(my, L.fmin1, L.fmax1);
                FPEL1(mx, my);
            }
        }
    }
    FPEL1(0, 0);
    // four candidates around (ox, oy) in the given order, strict '<' updates (COST_MV_X4)
#define X4(ox_, oy_, ax, ay, bx_, by_, cx_, cy_, dx_, dy_) do { \
    cx[0] = (ox_) + (ax); cy[0] = (oy_) + (ay); cx[1] = (ox_) + (bx_); cy[1] = (oy_) + (by_); \
    cx[2] = (ox_) + (cx_); cy[2] = (oy_) + (cy_); cx[3] = (ox_) + (dx_); cy[3] = (oy_) + (dy_); \
    sad_fpel4(c, cx, cy, res); \
    for (int k_ = 0; k_ < 4; k_++) { int cost_ = res[k_] + c.cmx[cx[k_] << 2] + c.cmy[cy[k_] << 2]; \
        if (cost_ < bcost) { bcost = cost_; bmx = cx[k_]; bmy = cy[k_]; } } } while (0)
    if (o.method == 0) {
        int i = 0;
        do {
            const int ox = bmx, oyy = bmy;
            X4(ox, oyy, 0, -1, 0, 1, -1, 0, 1, 0);
            if (bmx == ox && bmy == oyy) break;
            if (!INRANGE(bmx, bmy)) break;
        } while (++i < o.me_range);
    } else {
        const int hex2[8][2] = {{-1,-2},{-2,0},{-1,2},{1,2},{2,0},{1,-2},{-1,-2},{-2,0}};
        const int mod6m1[8] = {5,0,1,2,3,4,5,0};
        int dir = -2, costs[6];
        cx[0] = bmx - 2; cy[0] = bmy; cx[1] = bmx - 1; cy[1] = bmy + 2; cx[2] = bmx + 1; cy[2] = bmy + 2; cx[3] = bmx + 2; cy[3] = bmy;
        sad_fpel4(c, cx, cy, res);
        for (int k = 0; k < 4; k++) costs[k] = res[k] + c.cmx[cx[k] << 2] + c.cmy[cy[k] << 2];
        cx[0] = bmx + 1; cy[0] = bmy - 2; cx[1] = bmx - 1; cy[1] = bmy - 2; cx[2] = cx[0]; cy[2] = cy[0]; cx[3] = cx[0]; cy[3] = cy[0];
        sad_fpel4(c, cx, cy, res);
        for (int k = 0; k < 2; k++) costs[4 + k] = res[k] + c.cmx[cx[k] << 2] + c.cmy[cy[k] << 2];
        for (int k = 0; k < 6; k++) if (costs[k] < bcost) { bcost = costs[k]; dir = k; }
        if (dir != -2) {
            bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            for (int i = 1; i < o.me_range / 2 && INRANGE(bmx, bmy); i++) {
                const int odir = mod6m1[dir + 1];
                for (int k = 0; k < 3; k++) { cx[k] = bmx + hex2[odir + k][0]; cy[k] = bmy + hex2[odir + k][1]; }
                cx[3] = cx[0]; cy[3] = cy[0];
                sad_fpel4(c, cx, cy, res);
                dir = -2;
                for (int k = 0; k < 3; k++) {
                    int cost = res[k] + c.cmx[cx[k] << 2] + c.cmy[cy[k] << 2];
                    if (cost < bcost) { bcost = cost; dir = odir - 1 + k; }
                }
                if (dir == -2) break;
                bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            }
        }
        const int ox = bmx, oyy = bmy;
        X4(ox, oyy, 0, -1, 0, 1, -1, 0, 1, 0);
        X4(ox, oyy, -1, -1, -1, 1, 1, -1, 1, 1);
    }
    int mvx, mvy, mcost;
    if (bpcost < bcost) { mvx = bpx; mvy = bpy; mcost = bpcost; }
    else { mvx = bmx << 2; mvy = bmy << 2; mcost = bcost; }
    out_cost_mv = c.cmx[mvx] + c.cmy[mvy];                                   // m->cost_mv, me.c:615
    if (bmx == pmx && bmy == pmy && o.subme < 3) mcost += out_cost_mv;
    if (o.subme >= 2) {
        const int sub_iters[10][2] = {{0,0},{0,0},{1,0},{1,0},{1,1},{1,2},{2,2},{2,2},{4,10},{4,10}};   // subpel_iterations[][2..3]
        const int hpel = sub_iters[o.subme][0], qpel = sub_iters[o.subme][1];
        int bx = mvx, by = mvy, bc = mcost, odir = -1, bdir;
        bool early = false;
        if (hpel && o.subme < 3) {
            int mx = clip3(mvpx, L.smin0, L.smax0), my = clip3(mvpy, L.smin1, L.smax1);
            if ((mx - bx) | (my - by)) {
                cx[0] = cx[1] = cx[2] = cx[3] = mx; cy[0] = cy[1] = cy[2] = cy[3] = my;
                sad_qpel4(c, cx, cy, res);
                int cost = res[0] + c.cmx[mx] + c.cmy[my];
                if (cost < bc) { bc = cost; bx = mx; by = my; }
            }
        }
        for (int i = hpel; i > 0; i--) {
            const int ox = bx, oyy = by;
            cx[0] = ox; cy[0] = oyy - 2; cx[1] = ox; cy[1] = oyy + 2; cx[2] = ox - 2; cy[2] = oyy; cx[3] = ox + 2; cy[3] = oyy;
            sad_qpel4(c, cx, cy, res);
            int c0 = res[0] + c.cmx[ox] + c.cmy[oyy - 2], c1 = res[1] + c.cmx[ox] + c.cmy[oyy + 2];
            int c2 = res[2] + c.cmx[ox - 2] + c.cmy[oyy], c3 = res[3] + c.cmx[ox + 2] + c.cmy[oyy];
            if (c0 < bc) { bc = c0; by = oyy - 2; }
            if (c1 < bc) { bc = c1; by = oyy + 2; }
            if (c2 < bc) { bc = c2; bx = ox - 2; by = oyy; }
            if (c3 < bc) { bc = c3; bx = ox + 2; by = oyy; }
            if (bx == ox && by == oyy) break;
        }
        if (by > L.smax1) by = L.smax1;
        bc = MX_COST_MAX;
        { int cost = me_cost_satd(c, satd, o.chroma_me, bx, by, bc); if (cost < bc) bc = cost; }
        if (thresh) {
            if (((bc * 7) >> 3) > *thresh) early = true;
            else if (bc < *thresh) *thresh = bc;
        }
        if (!early) {
            bdir = -1;
            for (int i = qpel; i > 0; i--) {
                const int ox = bx, oyy = by;
                odir = bdir;
#pragma unroll 1
                for (int d = 0; d < 4; d++) {
                    if ((d ^ 1) == odir) continue;
                    const int mx = ox + (d == 2 ? -1 : d == 3 ? 1 : 0), my = oyy + (d == 0 ? -1 : d == 1 ? 1 : 0);
                    int cost = me_cost_satd(c, satd, o.chroma_me, mx, my, bc);
                    if (cost < bc) { bc = cost; bx = mx; by = my; bdir = d; }
                }
                if (bx == ox && by == oyy) break;
            }
            if (by > L.smax1) {
                by = L.smax1; bc = MX_COST_MAX;
                int cost = me_cost_satd(c, satd, o.chroma_me, bx, by, bc); if (cost < bc) bc = cost;
            }
            out_cost_mv = c.cmx[bx] + c.cmy[by];                             // me.c:777
        }
        mvx = bx; mvy = by; mcost = bc;
    } else if (mvy > L.smax1) mvy = L.smax1;
#undef X4
#undef FPEL1
#undef INRANGE
    out_mvx = mvx; out_mvy = mvy;
    return mcost;
}

// x264_me_refine_qpel -> refine_subpel(.., b_refine_qpel = 1) for PIXEL_16x16 (me.c:634-644, 680-778):
// half-pel diamond with SAD, quarter-pel diamond with mbcmp (all four directions every round).
// cost comes in without the reference cost; returns the refined cost.
__device__ int me_refine_qpel16(const MxCtx &c, const MeLimits &L, const MeOpts &o, int mvpx, int mvpy, int cost, int &mvx, int &mvy)
{
    const int it[10][2] = {{0,0},{1,1},{0,1},{0,2},{0,2},{0,2},{0,0},{0,0},{0,0},{0,0}};               // subpel_iterations[][0..1]
    const int hpel = it[o.subme][0], qpel = it[o.subme][1], satd = o.subme > 1;
    int bx = mvx, by = mvy, bc = cost, cx[4], cy[4], res[4];
    if (hpel && o.subme < 3) {
        int mx = clip3(mvpx, L.smin0, L.smax0), my = clip3(mvpy, L.smin1, L.smax1);
        if ((mx - bx) | (my - by)) {
            cx[0] = cx[1] = cx[2] = cx[3] = mx; cy[0] = cy[1] = cy[2] = cy[3] = my;
            sad_qpel4(c, cx, cy, res);
            int cst = res[0] + c.cmx[mx] + c.cmy[my];
            if (cst < bc) { bc = cst; bx = mx; by = my; }
        }
    }
    for (int i = hpel; i > 0; i--) {
        const int ox = bx, oyy = by;
        cx[0] = ox; cy[0] = oyy - 2; cx[1] = ox; cy[1] = oyy + 2; cx[2] = ox - 2; cy[2] = oyy; cx[3] = ox + 2; cy[3] = oyy;
        sad_qpel4(c, cx, cy, res);
        int c0 = res[0] + c.cmx[ox] + c.cmy[oyy - 2], c1 = res[1] + c.cmx[ox] + c.cmy[oyy + 2];
        int c2 = res[2] + c.cmx[ox - 2] + c.cmy[oyy], c3 = res[3] + c.cmx[ox + 2] + c.cmy[oyy];
        if (c0 < bc) { bc = c0; by = oyy - 2; }
        if (c1 < bc) { bc = c1; by = oyy + 2; }
        if (c2 < bc) { bc = c2; bx = ox - 2; by = oyy; }
        if (c3 < bc) { bc = c3; bx = ox + 2; by = oyy; }
        if (bx == ox && by == oyy) break;
    }
    for (int i = qpel; i > 0; i--) {
        const int ox = bx, oyy = by;
#pragma unroll 1
        for (int d = 0; d < 4; d++) {
            const int mx = ox + (d == 2 ? -1 : d == 3 ? 1 : 0), my = oyy + (d == 0 ? -1 : d == 1 ? 1 : 0);
            int cst = me_cost_satd(c, satd, o.chroma_me, mx, my, bc);
            if (cst < bc) { bc = cst; bx = mx; by = my; }
        }
        if (bx == ox && by == oyy) break;
    }
    if (by > L.smax1) {
        by = L.smax1; bc = MX_COST_MAX;
        int cst = me_cost_satd(c, satd, o.chroma_me, bx, by, bc); if (cst < bc) bc = cst;
    }
    mvx = bx; mvy = by;
    return bc;
}
