// frame_me.hip -- FRAME LEVEL, part 2: motion estimation, one wavefront per
// macroblock, every macroblock of a frame per launch.
//
// Full-pel stage (x264hip_me_fullpel_frame): the arithmetic of
// x264_me_search_ref's candidate test (COST_MV, R/encoder/me.c:54-62:
// cost = fpelcmp(fenc, ref + mv) + p_cost_mvx[mx<<2] + p_cost_mvy[my<<2])
// evaluated for EVERY full-pel vector of a +-range window (the set the
// reference's ESA scan visits, me.c:449-560) for the nine partitions
// 16x16, 16x8 x2, 8x16 x2, 8x8 x4 of the macroblock at once, with the
// reference's motion-vector limits (mv_min_fpel / mv_max_fpel,
// R/encoder/analyse.c:258-298).  Ties resolve to the first vector in (my, mx)
// raster order.  Optionally the raw 16x16 SAD surface is written out so a
// host-side replica of the DIA/HEX/UMH walks can look candidates up instead
// of computing them.
//
// gfx950 mapping: the (16+2R)^2 reference window and the 16x16 source block
// are staged once in LDS per wavefront with aligned dword loads; a lane owns
// four horizontally adjacent candidates of one window row, so one
// v_qsad_pk_u16_u8 (4 SADs of 4 bytes at byte offsets 0..3, packed 16-bit
// accumulate) advances four candidates by four pixels.  Quadrant sums give
// all nine partitions; the argmin is a wave64 shuffle reduction over packed
// (cost << 12 | raster index) keys.  No MFMA: these are byte absolute
// differences, not contractions.
//
// Sub-pel stage (x264hip_me_subpel_frame): exhaustive 3x3 half-pel then 3x3
// quarter-pel refinement of the 16x16 vector scored with SATD + mv cost
// through the reference's qpel blend of the four half-pel planes (get_ref,
// R/common/mc.c:181-202; mbcmp = SATD, R/encoder/encoder.c:608-618).
#include <cstdlib>
#include "device_prims.h"
#include "frame_internal.h"

using namespace x264hip;

#define ME_MAX_RANGE 24
#define ME_WAVES 4

struct MeGeom {
    int mb_w, mb_h, stride, width16, lines16, range, mv_range, cost_center;
    size_t bs;          // bytes between batch elements of a luma-sized plane
};

// mv_min_fpel / mv_max_fpel of R/encoder/analyse.c:258-298 (single slice thread)
__device__ __forceinline__ void mv_limits_fpel(const MeGeom &g, int mbx, int mby, int &x0, int &x1, int &y0, int &y1)
{
    int fr = 4 * g.mv_range;
    int minx = clip3(4 * (-16 * mbx - 24), -fr, fr - 1), maxx = clip3(4 * (16 * (g.mb_w - mbx - 1) + 24), -fr, fr - 1);
    int lo = 4 * (-512 + 8) > -fr ? 4 * (-512 + 8) : -fr;
    int miny = clip3(4 * (-16 * mby - 24), lo, fr), maxy = clip3(4 * (16 * (g.mb_h - mby - 1) + 24), -fr, fr - 1);
    x0 = (minx >> 2) + 5; x1 = (maxx >> 2) - 5; y0 = (miny >> 2) + 5; y1 = (maxy >> 2) - 5;
}
__device__ __forceinline__ void mv_limits_spel(const MeGeom &g, int mbx, int mby, int &x0, int &x1, int &y0, int &y1)
{
    int fr = 4 * g.mv_range;
    x0 = clip3(4 * (-16 * mbx - 24), -fr, fr - 1); x1 = clip3(4 * (16 * (g.mb_w - mbx - 1) + 24), -fr, fr - 1);
    int lo = 4 * (-512 + 8) > -fr ? 4 * (-512 + 8) : -fr;
    y0 = clip3(4 * (-16 * mby - 24), lo, fr); y1 = clip3(4 * (16 * (g.mb_h - mby - 1) + 24), -fr, fr - 1);
}

template <int R>
__global__ __launch_bounds__(64 * ME_WAVES) void k_me_fullpel(const u8 *__restrict__ fenc, const u8 *__restrict__ ref, MeGeom g,
                                                              const u16 *__restrict__ cost_mv, const i16 *__restrict__ centers,
                                                              const i16 *__restrict__ mvp_in, i16 *__restrict__ out_mv,
                                                              int *__restrict__ out_cost, u16 *__restrict__ surface)
{
    constexpr int N = 2 * R + 1;             // candidates per axis
    constexpr int G = (N + 3 + 3) / 4;       // 4-candidate groups per row (window origin aligned down to 4 bytes)
    constexpr int WROWS = 16 + 2 * R;
    constexpr int WSD = G + 5;               // dwords per window row: 4*(G-1) + 20 bytes, +1 dword slack
    __shared__ u32 s_win[ME_WAVES][WROWS * WSD];
    __shared__ u32 s_fenc[ME_WAVES][64];
    __shared__ u16 s_cost[ME_WAVES][2][4 * G];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mb = xcd_band_order(blockIdx.x, gridDim.x) * ME_WAVES + wave;
    if (mb >= g.mb_w * g.mb_h) return;       // whole wave exits together; no block-wide barrier below
    {   // batch element: shift every per-frame pointer once
        const size_t bz = blockIdx.y, nmb = (size_t)g.mb_w * g.mb_h;
        fenc += g.bs * bz; ref += g.bs * bz;
        if (centers) centers += 2 * nmb * bz;
        if (mvp_in) mvp_in += 2 * nmb * bz;
        out_mv += 18 * nmb * bz; out_cost += 9 * nmb * bz;
        if (surface) surface += (size_t)N * N * nmb * bz;
    }
    const int mbx = mb % g.mb_w, mby = mb / g.mb_w;
    const int px = mbx * 16, py = mby * 16;
    int cx = 0, cy = 0, pvx = 0, pvy = 0;
    if (centers) { cx = centers[2 * mb]; cy = centers[2 * mb + 1]; }
    if (mvp_in) { pvx = mvp_in[2 * mb]; pvy = mvp_in[2 * mb + 1]; }
    int lx0, lx1, ly0, ly1;
    mv_limits_fpel(g, mbx, mby, lx0, lx1, ly0, ly1);

    // ---- stage the source block and the window ----
    u32 *win = s_win[wave];
    {
        int r = lane >> 2, q = lane & 3;
        s_fenc[wave][lane] = *(const u32 *)(fenc + (ptrdiff_t)(py + r) * g.stride + px + 4 * q);
    }
    const int wx = px + cx - R;                       // first candidate column (absolute pixel x)
    const int ox = wx & ~3, shift = wx - ox;          // aligned window origin
    const int wy = py + cy - R;
    for (int i = lane; i < WROWS * WSD; i += 64) {
        int r = i / WSD, d = i % WSD;
        int y = wy + r, x = ox + 4 * d;
        // stay inside the padded plane; clamped dwords only feed vectors outside the mv limits
        y = y < -PADV ? -PADV : (y > g.lines16 + PADV - 1 ? g.lines16 + PADV - 1 : y);
        x = x < -PADH ? -PADH : (x > g.width16 + PADH - 4 ? g.width16 + PADH - 4 : x);
        win[i] = *(const u32 *)(ref + (ptrdiff_t)y * g.stride + x);
    }
    // mv cost of every candidate column / row (qpel delta to the predictor)
    for (int i = lane; i < 2 * 4 * G; i += 64) {
        int axis = i / (4 * G), p = i % (4 * G);
        int mvq = axis == 0 ? 4 * (cx - R + p - shift) - pvx : 4 * (cy - R + p) - pvy;
        int idx = g.cost_center + mvq;
        idx = idx < 0 ? 0 : (idx > 2 * g.cost_center ? 2 * g.cost_center : idx);
        s_cost[wave][axis][p] = cost_mv[idx];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);   // LDS writes of this wave visible to its own later reads

    u32 best[9];
#pragma unroll
    for (int k = 0; k < 9; k++) best[k] = 0xffffffffu;

    const u32 *fe = s_fenc[wave];
    for (int task = lane; task < N * G; task += 64) {
        const int j = task / G, gq = task % G;       // window row of the candidate, 4-candidate group
        unsigned long long acc[4] = {0, 0, 0, 0};    // quadrants TL, TR, BL, BR; 4 x u16 each
        const u32 *wr = win + j * WSD + gq;
#pragma unroll
        for (int row = 0; row < 16; row++) {
            const u32 *w = wr + row * WSD;
            u32 r0 = w[0], r1 = w[1], r2 = w[2], r3 = w[3], r4 = w[4];
            const u32 *f = fe + 4 * row;
            const int qb = row < 8 ? 0 : 2;
            acc[qb]     = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)r1 << 32) | r0, f[0], acc[qb]);
            acc[qb]     = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)r2 << 32) | r1, f[1], acc[qb]);
            acc[qb + 1] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)r3 << 32) | r2, f[2], acc[qb + 1]);
            acc[qb + 1] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)r4 << 32) | r3, f[3], acc[qb + 1]);
        }
        const int my = cy - R + j;
        const int cyc = s_cost[wave][1][j];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int p = 4 * gq + c;                      // byte position in the aligned window
            int dxi = p - shift;                     // 0..2R for real candidates
            int mx = cx - R + dxi;
            bool ok = dxi >= 0 && dxi < N && mx >= lx0 && mx <= lx1 && my >= ly0 && my <= ly1;
            u32 s0 = (u32)(acc[0] >> (16 * c)) & 0xffffu, s1 = (u32)(acc[1] >> (16 * c)) & 0xffffu;
            u32 s2 = (u32)(acc[2] >> (16 * c)) & 0xffffu, s3 = (u32)(acc[3] >> (16 * c)) & 0xffffu;
            if (surface && dxi >= 0 && dxi < N)
                surface[(size_t)mb * N * N + j * N + dxi] = (u16)(s0 + s1 + s2 + s3);
            if (!ok) continue;
            u32 mvc = (u32)s_cost[wave][0][p] + (u32)cyc;
            u32 idx = (u32)(j * N + dxi);
            u32 part[9] = {s0 + s1 + s2 + s3, s0 + s1, s2 + s3, s0 + s2, s1 + s3, s0, s1, s2, s3};
#pragma unroll
            for (int k = 0; k < 9; k++) {
                u32 key = ((part[k] + mvc) << 12) | idx;
                best[k] = key < best[k] ? key : best[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        u32 v = best[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            u32 o = (u32)__shfl_xor((int)v, m, 64);
            v = o < v ? o : v;
        }
        best[k] = v;
    }
    if (lane < 9) {
        u32 v = best[0];
#pragma unroll
        for (int k = 1; k < 9; k++) v = lane == k ? best[k] : v;
        int idx = (int)(v & 0xfffu), cost = (int)(v >> 12);
        int mvx = cx - R + idx % N, mvy = cy - R + idx / N;
        if (v == 0xffffffffu) { mvx = 0; mvy = 0; cost = 0x7fffffff; }   // no admissible vector
        out_mv[((size_t)mb * 9 + lane) * 2] = (i16)mvx;
        out_mv[((size_t)mb * 9 + lane) * 2 + 1] = (i16)mvy;
        out_cost[(size_t)mb * 9 + lane] = cost;
    }
}

// ---------------------------------------------------------------- sub-pel
// one wavefront per macroblock; lane = candidate (8 neighbours) x 8x4 block (8)
__device__ __forceinline__ int satd_8x4_d(const int d[4][8])
{
    u32 t[4][4];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        u32 p0 = (u32)d[y][0] + ((u32)d[y][4] << 16), p1 = (u32)d[y][1] + ((u32)d[y][5] << 16);
        u32 p2 = (u32)d[y][2] + ((u32)d[y][6] << 16), p3 = (u32)d[y][3] + ((u32)d[y][7] << 16);
        wht4(t[y][0], t[y][1], t[y][2], t[y][3], p0, p1, p2, p3);
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

#define SP_W 20                       // window stride (18 used) per plane
__global__ __launch_bounds__(64 * ME_WAVES) void k_me_subpel(const u8 *__restrict__ fenc, const u8 *__restrict__ p0, const u8 *__restrict__ p1,
                                                             const u8 *__restrict__ p2, const u8 *__restrict__ p3, MeGeom g,
                                                             const u16 *__restrict__ cost_mv, const i16 *__restrict__ mvp_in,
                                                             const i16 *__restrict__ mv_fullpel, i16 *__restrict__ out_mv, int *__restrict__ out_cost)
{
    __shared__ u8 s_ref[ME_WAVES][4][18 * SP_W];
    __shared__ u8 s_fe[ME_WAVES][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mb = xcd_band_order(blockIdx.x, gridDim.x) * ME_WAVES + wave;
    if (mb >= g.mb_w * g.mb_h) return;
    {
        const size_t bz = blockIdx.y, nmb = (size_t)g.mb_w * g.mb_h;
        fenc += g.bs * bz; p0 += g.bs * bz; p1 += g.bs * bz; p2 += g.bs * bz; p3 += g.bs * bz;
        if (mvp_in) mvp_in += 2 * nmb * bz;
        mv_fullpel += 18 * nmb * bz; out_mv += 2 * nmb * bz; out_cost += nmb * bz;
    }
    const int mbx = mb % g.mb_w, mby = mb / g.mb_w, px = mbx * 16, py = mby * 16;
    int pvx = 0, pvy = 0;
    if (mvp_in) { pvx = mvp_in[2 * mb]; pvy = mvp_in[2 * mb + 1]; }
    const int fx = mv_fullpel[(size_t)mb * 18], fy = mv_fullpel[(size_t)mb * 18 + 1];
    int sx0, sx1, sy0, sy1;
    mv_limits_spel(g, mbx, mby, sx0, sx1, sy0, sy1);

    ((u32 *)s_fe[wave])[lane] = *(const u32 *)(fenc + (ptrdiff_t)(py + (lane >> 2)) * g.stride + px + 4 * (lane & 3));
    // 18x18 window of each half-pel plane around the full-pel block (origin -1,-1)
    for (int i = lane; i < 4 * 18 * 18; i += 64) {
        int pl = i / 324, r = (i % 324) / 18, c = i % 18;
        const u8 *src = pl == 0 ? p0 : pl == 1 ? p1 : pl == 2 ? p2 : p3;
        s_ref[wave][pl][r * SP_W + c] = src[(ptrdiff_t)(py + fy - 1 + r) * g.stride + px + fx - 1 + c];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);

    const u8 *fe = s_fe[wave];
    const int blk = lane & 7, bx = (blk & 1) * 8, by = (blk >> 1) * 4;
    int bmx = 4 * fx, bmy = 4 * fy, bcost = 0;
    // SATD + mv cost of qpel vector (mx,my) for this lane's 8x4 block, summed over the 8 lanes of a candidate
    auto score = [&](int mx, int my) -> int {
        int qx = mx & 3, qy = my & 3, idx = qy * 4 + qx;
        int ix = (mx >> 2) - fx + 1, iy = (my >> 2) - fy + 1;       // position inside the 18x18 window
        const u8 *a = s_ref[wave][c_qpel_a[idx]] + (iy + (qy == 3)) * SP_W + ix;
        const u8 *b = s_ref[wave][c_qpel_b[idx]] + iy * SP_W + ix + (qx == 3);
        int d[4][8];
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int x = 0; x < 8; x++) {
                int o = (by + y) * SP_W + bx + x;
                int pr = (idx & 5) ? ((int)a[o] + (int)b[o] + 1) >> 1 : (int)a[o];
                d[y][x] = (int)fe[(by + y) * 16 + bx + x] - pr;
            }
        int s = group_sum(satd_8x4_d(d), 8);
        int ci = g.cost_center + (mx - pvx), cj = g.cost_center + (my - pvy);
        ci = ci < 0 ? 0 : (ci > 2 * g.cost_center ? 2 * g.cost_center : ci);
        cj = cj < 0 ? 0 : (cj > 2 * g.cost_center ? 2 * g.cost_center : cj);
        return s + (int)cost_mv[ci] + (int)cost_mv[cj];
    };
#pragma unroll 1
    for (int step = 2; step >= 1; step--) {          // half-pel, then quarter-pel
        // centre first (all lanes compute it redundantly in groups of 8), then the 8 neighbours in raster order
        int cc = score(bmx, bmy);
        if (step == 2) bcost = cc;                   // stage-B centre equals stage-A winner's cost
        int n = lane >> 3;                           // neighbour 0..7 -> raster slot skipping the centre
        int slot = n < 4 ? n : n + 1;
        int mx = bmx + step * (slot % 3 - 1), my = bmy + step * (slot / 3 - 1);
        bool ok = mx >= sx0 && mx <= sx1 && my >= sy0 && my <= sy1;
        // out-of-range candidates still run the arithmetic on a safe vector so every lane reaches the shuffles
        int sc = score(ok ? mx : bmx, ok ? my : bmy);
        u32 key = ok ? (((u32)sc << 4) | (u32)slot) : 0xffffffffu;
        u32 ckey = ((u32)cc << 4) | 4u;
        key = key < ckey ? key : ckey;
#pragma unroll
        for (int m = 32; m >= 8; m >>= 1) {
            u32 o = (u32)__shfl_xor((int)key, m, 64);
            key = o < key ? o : key;
        }
        int win = (int)(key & 15u);
        bcost = (int)(key >> 4);
        bmx += step * (win % 3 - 1); bmy += step * (win / 3 - 1);
    }
    if (lane == 0) {
        out_mv[2 * mb] = (i16)bmx; out_mv[2 * mb + 1] = (i16)bmy;
        out_cost[mb] = bcost;
    }
}

// -------------------------------------------------------------------- host
static MeGeom make_geom(const x264hip_frame_ctx *c, const x264hip_me_params *p)
{
    MeGeom g;
    g.mb_w = c->d.mb_w; g.mb_h = c->d.mb_h; g.stride = c->d.stride_y;
    g.width16 = c->width16; g.lines16 = c->lines16;
    g.range = p->range; g.mv_range = p->mv_range > 0 ? p->mv_range : 512;
    g.cost_center = p->cost_mv_range;
    g.bs = c->bs_y;
    return g;
}

extern "C" int x264hip_me_fullpel_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                                        const x264hip_me_params *p, int16_t *out_mv_dev, int32_t *out_cost_dev)
{
    if (!p->cost_mv || p->cost_mv_range < 4 * (p->range + 8)) { set_error("me: cost_mv table missing or too short"); return -1; }
    MeGeom g = make_geom(c, p);
    int n = g.mb_w * g.mb_h;
    dim3 grid((n + ME_WAVES - 1) / ME_WAVES, c->batch), block(64 * ME_WAVES);
#define LAUNCH_ME(RR) hipLaunchKernelGGL(k_me_fullpel<RR>, grid, block, 0, c->stream, fenc->plane[0], ref->plane[0], g, \
        p->cost_mv, p->centers, p->mvp, out_mv_dev, out_cost_dev, p->sad_surface)
    switch (p->range) {
    case 8: LAUNCH_ME(8); break;
    case 16: LAUNCH_ME(16); break;
    case 24: LAUNCH_ME(24); break;
    default: set_error("me: range %d not built (8, 16, 24)", p->range); return -1;
    }
#undef LAUNCH_ME
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int x264hip_me_subpel_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                                       const x264hip_me_params *p, const int16_t *mv_fullpel_dev, int16_t *out_mv_qpel_dev, int32_t *out_cost_dev)
{
    if (!p->cost_mv) { set_error("me: cost_mv table missing"); return -1; }
    MeGeom g = make_geom(c, p);
    int n = g.mb_w * g.mb_h;
    hipLaunchKernelGGL(k_me_subpel, dim3((n + ME_WAVES - 1) / ME_WAVES, c->batch), dim3(64 * ME_WAVES), 0, c->stream, fenc->plane[0],
                       ref->filtered[0], ref->filtered[1], ref->filtered[2], ref->filtered[3], g, p->cost_mv, p->mvp,
                       mv_fullpel_dev, out_mv_qpel_dev, out_cost_dev);
    HIPCHK(hipGetLastError());
    return 0;
}
