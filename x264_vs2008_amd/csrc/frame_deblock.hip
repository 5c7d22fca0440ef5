// frame_deblock.hip -- FRAME LEVEL, part 4: the in-loop deblocking filter of a
// whole progressive P frame (x264_frame_deblock_row for every row,
// R/common/frame.c:621-792; edge kernels :420-586; tables :377-417).
//
// Exactness constraint: H.264 filters macroblocks in raster order, vertical
// edges then horizontal edges of each, and every edge reads pixels its
// predecessors already modified.  Macroblock (x, y) therefore needs (x-1, y)
// and (x+1, y-1) finished, nothing else: all macroblocks with equal x + 2y
// are independent.  The frame is swept as 2:1 anti-diagonals, one launch per
// diagonal (mb_w + 2*mb_h - 2 launches), one wavefront per macroblock, so the
// kernel boundary is the only synchronisation and results cannot depend on
// dispatch order.  Tiles of concurrently filtered macroblocks never overlap
// (their column ranges are >= 12 pixels apart).
//
// Inside a wavefront: the 20x20 luma and two 12x12 chroma tiles (4 pixels of
// left / top context) live in LDS; lane = picture line across the edge (16
// luma + 8 U + 8 V lines), edges of one direction are walked left-to-right /
// top-to-bottom in registers, exactly the order of the reference.
#include "device_prims.h"
#include "frame_internal.h"

using namespace x264hip;

static __constant__ u8 c_alpha[76] = {
    0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,25,28,32,36,40,45,50,56,63,71,
    80,90,101,113,127,144,162,182,203,226,255,255, 255,255,255,255,255,255,255,255,255,255,255,255};
static __constant__ u8 c_beta[76] = {
    0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,8,8,9,9,10,10,11,11,12,12,
    13,13,14,14,15,15,16,16,17,17,18,18, 18,18,18,18,18,18,18,18,18,18,18,18};
static __constant__ signed char c_tc0[76][4] = {
    {-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},
    {-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},
    {-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},{-1,0,0,0},
    {-1,0,0,1},{-1,0,0,1},{-1,0,0,1},{-1,0,0,1},{-1,0,1,1},{-1,0,1,1},{-1,1,1,1},{-1,1,1,1},{-1,1,1,1},{-1,1,1,1},{-1,1,1,2},{-1,1,1,2},
    {-1,1,1,2},{-1,1,1,2},{-1,1,2,3},{-1,1,2,3},{-1,2,2,3},{-1,2,2,4},{-1,2,3,4},{-1,2,3,4},{-1,3,3,5},{-1,3,4,6},{-1,3,4,6},{-1,4,5,7},
    {-1,4,5,8},{-1,4,6,9},{-1,5,7,10},{-1,6,8,11},{-1,6,8,13},{-1,7,10,14},{-1,8,11,16},{-1,9,12,18},{-1,10,13,20},{-1,11,15,23},{-1,13,17,25},
    {-1,13,17,25},{-1,13,17,25},{-1,13,17,25},{-1,13,17,25},{-1,13,17,25},{-1,13,17,25},
    {-1,13,17,25},{-1,13,17,25},{-1,13,17,25},{-1,13,17,25},{-1,13,17,25},{-1,13,17,25}};
static __constant__ u8 c_chroma_qp[76] = {
    0,0,0,0,0,0,0,0,0,0,0,0, 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,
    29,30,31,32,32,33,34,34,35,35,36,36,37,37,37,38,38,38,39,39,39,39, 39,39,39,39,39,39,39,39,39,39,39,39};

struct DbGeom {
    int mb_w, mb_h, sy, sc, a_off, b_off, cqp_off, diag, y_min, count;
    size_t bs_y, bs_c;  // bytes between batch elements
    int layout;         // 0: compact type codes (0 inter 1 intra 2 skip 3 P_8x8 with sub-8x8 on), nnz[26]; 1: x264hip_mb_state (reference numbering, nnz[27])
    int sub8x8;         // X264_ANALYSE_PSUB8x8 is on
};

__device__ __forceinline__ int z_of(int x, int y) { return (y >> 1) * 8 + (x >> 1) * 4 + (y & 1) * 2 + (x & 1); }

// one line across an edge; v[0..7] = p3 p2 p1 p0 q0 q1 q2 q3.  kind: 0 luma bS<4,
// 1 chroma bS<4 (tc0 already +1), 2 luma strong, 3 chroma strong.  R/common/frame.c:420-586
__device__ __forceinline__ void db_line(int kind, int *v, int alpha, int beta, int tc0)
{
    int p2 = v[1], p1 = v[2], p0 = v[3], q0 = v[4], q1 = v[5], q2 = v[6];
    if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    if (kind == 0) {
        int tc = tc0;
        if (iabs(p2 - p0) < beta) { v[2] = p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0); tc++; }
        if (iabs(q2 - q0) < beta) { v[5] = q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0); tc++; }
        int d = clip3((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
        v[3] = clip_u8(p0 + d); v[4] = clip_u8(q0 - d);
    } else if (kind == 1) {
        int d = clip3((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc0, tc0);
        v[3] = clip_u8(p0 + d); v[4] = clip_u8(q0 - d);
    } else if (kind == 2) {
        if (iabs(p0 - q0) < (alpha >> 2) + 2) {
            if (iabs(p2 - p0) < beta) {
                int p3 = v[0];
                v[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
                v[2] = (p2 + p1 + p0 + q0 + 2) >> 2;
                v[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
            } else
                v[3] = (2 * p1 + p0 + q1 + 2) >> 2;
            if (iabs(q2 - q0) < beta) {
                int q3 = v[7];
                v[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
                v[5] = (p0 + q0 + q1 + q2 + 2) >> 2;
                v[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
            } else
                v[4] = (2 * q1 + q0 + p1 + 2) >> 2;
        } else {
            v[3] = (2 * p1 + p0 + q1 + 2) >> 2;
            v[4] = (2 * q1 + q0 + p1 + 2) >> 2;
        }
    } else {
        v[3] = (2 * p1 + p0 + q1 + 2) >> 2;
        v[4] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
}

#define DB_WAVES 4
#define LT 24        // luma tile stride (20 used)
#define CT 12        // chroma tile stride

struct DbLds {
    u8 y[20 * LT];
    u8 c[2][12 * CT];
    u8 bs[2][4][4];      // [dir][edge][segment]; 4 = strong (intra macroblock edge), 255 = edge not filtered
    u8 qpe[2][2];        // [dir][0: edge 0, 1: inner] averaged luma qp
    u8 qpc[2][2];        // same for chroma
};

__global__ __launch_bounds__(64 * DB_WAVES) void k_deblock_diag(u8 *__restrict__ py, u8 *__restrict__ pu, u8 *__restrict__ pv, DbGeom g,
                                                                const u8 *__restrict__ mb_type, const u8 *__restrict__ qp,
                                                                const u8 *__restrict__ nnz, const u8 *__restrict__ t8x8,
                                                                const i16 *__restrict__ mv, const signed char *__restrict__ ref, const int *__restrict__ elems)
{
    __shared__ DbLds s_all[DB_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * DB_WAVES + wave;
    if (k >= g.count) return;
    {   // batch element
        const size_t bz = elems ? elems[blockIdx.y] : blockIdx.y, nmb = (size_t)g.mb_w * g.mb_h;
        py += g.bs_y * bz; pu += g.bs_c * bz; pv += g.bs_c * bz;
        mb_type += nmb * bz; qp += nmb * bz; nnz += (size_t)(g.layout ? 27 : 26) * nmb * bz; t8x8 += nmb * bz; mv += 32 * nmb * bz; ref += 4 * nmb * bz;
    }
    DbLds &s = s_all[wave];
    const int mby = g.y_min + k, mbx = g.diag - 2 * mby, mb = mby * g.mb_w + mbx;
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); \
                         __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
    const int ns = g.layout ? 27 : 26;
#define DB_TYPE(t_) (g.layout ? ((t_) <= 3 ? 1 : (t_) == 6 ? 2 : 0) : (int)(t_) == 3 ? 0 : (int)(t_))
    const int q = qp[mb], t8 = t8x8[mb], type = DB_TYPE(mb_type[mb]);
    // no_sub8x8 (frame.c:645): 0 only for a P_8x8 macroblock while sub-8x8 partitions are analysed
    const int no_sub = g.layout ? !(g.sub8x8 && mb_type[mb] == 5) : mb_type[mb] != 3;
    const int qp_thresh = 15 - (g.a_off < g.b_off ? g.a_off : g.b_off) - (g.cqp_off > 0 ? g.cqp_off : 0);
    const int edge_end = (type == 2 || q <= qp_thresh) ? 1 : 4;

    // ---- boundary strengths: lane = dir*16 + edge*4 + segment (DEBLOCK_STRENGTH, frame.c:697-742) ----
    if (lane < 32) {
        int dir = lane >> 4, edge = (lane >> 2) & 3, i = lane & 3;
        bool at_border = dir ? mby == 0 : mbx == 0;
        bool active = edge == 0 ? !at_border : (edge < edge_end && (!t8 || !(edge & 1)));
        int bs = 255;
        if (active) {
            int mbn = edge ? mb : (dir ? mb - g.mb_w : mb - 1);
            bool intra = type == 1 || DB_TYPE(mb_type[mbn]) == 1;
            if (intra) bs = edge == 0 ? 4 : 3;
            else {
                // segments 0..i are needed for the "copy the previous segment" rule: recompute them serially
                int prev = 0;
                bs = 0;
                for (int j = 0; j <= i; j++) {
                    int x = dir == 0 ? edge : j, y = dir == 0 ? j : edge;
                    int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
                    int b = 0;
                    if (nnz[mb * ns + z_of(x, y)] || nnz[mbn * ns + z_of(xn, yn)]) b = 2;
                    else if (!(edge & no_sub)) {
                        if ((j & no_sub) && prev != 2) b = prev;
                        else {
                            const i16 *mp = mv + ((size_t)mb * 16 + x + 4 * y) * 2, *mq = mv + ((size_t)mbn * 16 + xn + 4 * yn) * 2;
                            int rp = ref[mb * 4 + (x >> 1) + (y >> 1) * 2], rq = ref[mbn * 4 + (xn >> 1) + (yn >> 1) * 2];
                            if (rp != rq || iabs(mp[0] - mq[0]) >= 4 || iabs(mp[1] - mq[1]) >= 4) b = 1;
                        }
                    }
                    prev = b;
                    bs = b;
                }
            }
        }
        s.bs[dir][edge][i] = (u8)bs;
        if (edge < 2 && i == 0) {
            bool nb = edge == 0 && !at_border;
            int qn = nb ? qp[dir ? mb - g.mb_w : mb - 1] : q;
            const u8 *cqt = c_chroma_qp + 12 + g.cqp_off;
            s.qpe[dir][edge] = (u8)((q + qn + 1) >> 1);
            s.qpc[dir][edge] = (u8)((cqt[q] + cqt[qn] + 1) >> 1);
        }
    }
    // ---- tiles: 4 pixels of left / top context ----
    u8 *ty = py + (ptrdiff_t)(16 * mby - 4) * g.sy + 16 * mbx - 4;
    for (int i = lane; i < 20 * 5; i += 64) {
        int r = i / 5, d = i % 5;
        *(u32 *)(s.y + r * LT + 4 * d) = *(const u32 *)(ty + (ptrdiff_t)r * g.sy + 4 * d);
    }
    u8 *tc[2] = {pu + (ptrdiff_t)(8 * mby - 4) * g.sc + 8 * mbx - 4, pv + (ptrdiff_t)(8 * mby - 4) * g.sc + 8 * mbx - 4};
    for (int i = lane; i < 2 * 12 * 3; i += 64) {
        int pl = i / 36, r = (i % 36) / 3, d = i % 3;
        *(u32 *)(s.c[pl] + r * CT + 4 * d) = *(const u32 *)(tc[pl] + (ptrdiff_t)r * g.sc + 4 * d);
    }
    WAVE_SYNC();

    // ---- filter: dir 0 = vertical edges (lines are rows), dir 1 = horizontal edges (lines are columns) ----
    for (int dir = 0; dir < 2; dir++) {
        if (lane < 32) {
            const bool luma = lane < 16;
            const int pl = (lane >> 3) & 1;                   // chroma plane for lanes 16..31
            const int line = luma ? lane : (lane & 7);
            u8 *base = luma ? s.y + 4 * LT + 4 : s.c[pl] + 4 * CT + 4;     // pixel (0,0) of the macroblock
            const int ts = luma ? LT : CT;
            const int along = dir == 0 ? ts : 1, across = dir == 0 ? 1 : ts;
            for (int edge = 0; edge < 4; edge++) {
                if (!luma && (edge & 1)) continue;
                int seg = luma ? line >> 2 : line >> 1;
                int bs = s.bs[dir][edge][seg];
                if (bs == 255) continue;
                int qe = luma ? s.qpe[dir][edge ? 1 : 0] : s.qpc[dir][edge ? 1 : 0];
                int ia = qe + g.a_off, alpha = c_alpha[ia + 12], beta = c_beta[qe + g.b_off + 12];
                if (!alpha || !beta) continue;
                // an edge whose four strengths are all zero is not filtered at all (frame.c:763,777)
                u32 all = *(const u32 *)s.bs[dir][edge];
                if (all == 0) continue;
                int tc0 = bs == 4 ? 0 : c_tc0[ia + 12][bs] + (luma ? 0 : 1);
                if (bs != 4 && (luma ? tc0 < 0 : tc0 <= 0)) continue;
                u8 *p = base + line * along + (luma ? 4 : 2) * edge * across;
                int v[8];
                const int half = luma ? 4 : 2;
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = (i - 4 >= -half && i - 4 < half) ? p[(i - 4) * across] : 0;
                db_line(bs == 4 ? (luma ? 2 : 3) : (luma ? 0 : 1), v, alpha, beta, tc0);
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (i - 4 >= -half + (luma ? 1 : 1) - 1 && i - 4 < half) p[(i - 4) * across] = (u8)v[i];
            }
        }
        WAVE_SYNC();
    }

    // ---- write the tiles back (rows / columns -4.. are unchanged context unless an edge touched them) ----
    for (int i = lane; i < 20 * 5; i += 64) {
        int r = i / 5, d = i % 5;
        *(u32 *)(ty + (ptrdiff_t)r * g.sy + 4 * d) = *(const u32 *)(s.y + r * LT + 4 * d);
    }
    for (int i = lane; i < 2 * 12 * 3; i += 64) {
        int pl = i / 36, r = (i % 36) / 3, d = i % 3;
        *(u32 *)(tc[pl] + (ptrdiff_t)r * g.sc + 4 * d) = *(const u32 *)(s.c[pl] + r * CT + 4 * d);
    }
#undef WAVE_SYNC
}

extern "C" int x264hip_deblock_frame(x264hip_frame_ctx *c, x264hip_picture *recon, const x264hip_deblock_params *p)
{
    DbGeom g;
    g.mb_w = c->d.mb_w; g.mb_h = c->d.mb_h; g.sy = c->d.stride_y; g.sc = c->d.stride_c;
    g.bs_y = c->bs_y; g.bs_c = c->bs_c;
    g.a_off = p->alpha_c0_offset; g.b_off = p->beta_offset; g.cqp_off = p->chroma_qp_offset; g.layout = p->state_layout ? 1 : 0; g.sub8x8 = p->sub8x8 != 0;
    if (g.a_off < -12 || g.a_off > 12 || g.b_off < -12 || g.b_off > 12 || g.cqp_off < -12 || g.cqp_off > 12) {
        set_error("deblock: offsets out of range");
        return -1;
    }
    const int last = g.mb_w - 1 + 2 * (g.mb_h - 1);
    for (int t = 0; t <= last; t++) {
        int y_min = t - (g.mb_w - 1) > 0 ? (t - (g.mb_w - 1) + 1) / 2 : 0;
        int y_max = t / 2 < g.mb_h - 1 ? t / 2 : g.mb_h - 1;
        if (y_max < y_min) continue;
        g.diag = t; g.y_min = y_min; g.count = y_max - y_min + 1;
        hipLaunchKernelGGL(k_deblock_diag, dim3((g.count + DB_WAVES - 1) / DB_WAVES, c->elems ? c->n_elems : c->batch), dim3(64 * DB_WAVES), 0, c->stream,
                           recon->plane[0], recon->plane[1], recon->plane[2], g, p->mb_type, p->qp, p->nnz, p->transform8x8,
                           p->mv, (const signed char *)p->ref, c->elems);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
