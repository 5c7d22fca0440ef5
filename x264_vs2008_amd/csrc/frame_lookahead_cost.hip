// frame_lookahead_cost.hip -- FRAME LEVEL, part 6: the lookahead's per-frame cost, x264_slicetype_frame_cost's uncached branch with
// x264_slicetype_mb_cost inside it (R/encoder/slicetype.c:43-253, 256-345), one TASK = (frame b scored against p0 and p1) per wavefront.
//
// Why one wavefront per task: the macroblocks of a task form a chain -- they are visited in reverse raster order so that the vectors of
// the right and the three lower neighbours are the candidates of the next search -- and a GPU full of GOP chains has thousands of
// independent tasks at a time, so parallelism comes from the tasks.  Inside the macroblock the wave is used as the main encode's search
// uses it (me_exact.h: x264_me_search_ref + refine_subpel as "trips" of 4 / 8 candidates, one per lane group), on an 8x8 block of the
// half-resolution planes at subme 4 with min(HEX, me); the bidirectional tries put one pixel on each lane (get_ref's blend, the
// (weighted) average and the 8x8 SATD as four cross-lane butterfly stages).  The intra half of the cost is read from
// x264hip_lookahead_intra_frame's output (every block independent there).
//
// Neighbour vectors travel through LDS (two rows of the task), never through global memory written by this launch; the arrays
// lowres_mvs / lowres_mv_costs in HBM are outputs here and inputs of later tasks and of the main encode (x264_mb_predict_mv_ref16x16).
#include "device_prims.h"
#include "frame_internal.h"
#include "me_exact.h"

using namespace x264hip;

#define LK_MAX_W 512                    // macroblocks per row the LDS rows hold (8192 luma samples)
#define LK_COST_MAX (1 << 28)

struct LookTaskDev {
    const u8 *pl[3][4];                 // lowres luma + H, V, HV of frame b, p0, p1 at this chain's picture origin
    i16 *mv[2];                         // frames[b]->lowres_mvs[l][dist - 1] of this chain, [n][2]
    int *mcost[2];                      // frames[b]->lowres_mv_costs[l][dist - 1], [n]
    const i16 *mvr;                     // frames[p1]->lowres_mvs[0][p1 - p0 - 1] (b < p1)
    const int *intra;                   // frames[b]->i_intra_cost, [n]
    int d0, d1;                         // b - p0, p1 - b
    int do_search[2];
};

#define LK_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); \
                       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

// hpel_ref0 / hpel_ref1 of get_ref (R/common/mc.c:176-177), two bits per quarter-pel phase
#define LK_HREF0 0x54FE5454u            // {0,1,1,1,0,1,1,1,2,3,3,3,0,1,1,1}
#define LK_HREF1 0xBABABA00u            // {0,0,0,0,2,2,3,2,2,2,3,2,2,2,3,2}

// get_ref's sample (x, y) of the 8x8 block at quarter-pel vector (mvx, mvy): mc.c:181-202
__device__ __forceinline__ int lk_ref_px(const u8 *p0, const u8 *p1, const u8 *p2, const u8 *p3, int stride, ptrdiff_t off, int mvx, int mvy)
{
    const int qi = ((mvy & 3) << 2) + (mvx & 3);
    const ptrdiff_t o = off + (ptrdiff_t)(mvy >> 2) * stride + (mvx >> 2);
    const int k0 = (LK_HREF0 >> (2 * qi)) & 3, k1 = (LK_HREF1 >> (2 * qi)) & 3;
    const u8 *a = k0 == 0 ? p0 : k0 == 1 ? p1 : k0 == 2 ? p2 : p3;
    int v = a[o + ((mvy & 3) == 3 ? stride : 0)];
    if (qi & 5) {
        const u8 *b = k1 == 0 ? p0 : k1 == 1 ? p1 : k1 == 2 ? p2 : p3;
        v = (v + (int)b[o + ((mvx & 3) == 3)] + 1) >> 1;
    }
    return v;
}
// x264_pixel_satd_8x8 of a difference block, one sample per lane (lane = 8 * y + x): two 8x4 halves, each the sum of its two 4x4
// Hadamards halved once (R/common/pixel.c:211-253)
__device__ __forceinline__ int lk_satd8x8(int d, int lane)
{
    int t = dpp_mov<DPP_XOR1>(d); d = (lane & 1) ? t - d : d + t;
    t = dpp_mov<DPP_XOR2>(d); d = (lane & 2) ? t - d : d + t;
    t = __shfl_xor(d, 8, 64); d = (lane & 8) ? t - d : d + t;
    t = __shfl_xor(d, 16, 64); d = (lane & 16) ? t - d : d + t;
    int a = row_sum16(iabs(d));
    const int top = __builtin_amdgcn_readlane(a, 0) + __builtin_amdgcn_readlane(a, 16);
    const int bot = __builtin_amdgcn_readlane(a, 32) + __builtin_amdgcn_readlane(a, 48);
    return (top >> 1) + (bot >> 1);
}

__global__ __launch_bounds__(64) void k_look_cost(const LookTaskDev *__restrict__ tasks, int mb_w, int mb_h, int stride, int method, int me_range,
                                                  int weighted_bipred, int bframe_bias, const i16 *__restrict__ cost_g, int *__restrict__ out)
{
    __shared__ u32 s_fe[16 * 4];                              // the source block in the macroblock layout the search reads (16-byte rows)
    __shared__ i16 s_costl[2 * MX_COST_LDS + 2];
    __shared__ u32 s_row[2][2][LK_MAX_W];                     // [list][row parity][mb x]: the vectors of this row and the one below
    __shared__ i16 s_mvc[8];
    const int lane = threadIdx.x;
    const LookTaskDev T = tasks[blockIdx.x];
    const int d0 = T.d0, d1 = T.d1, b_bidir = d1 > 0, intra_only = d0 == 0 && d1 == 0;
    int dist_scale_factor = 128;
    if (d0 + d1 != 0) dist_scale_factor = ((d0 << 8) + ((d0 + d1) >> 1)) / (d0 + d1);
    const int bipred_weight = weighted_bipred ? 64 - (dist_scale_factor >> 2) : 32;
    for (int i = lane; i < 2 * MX_COST_LDS + 1; i += 64) s_costl[i] = cost_g[i - MX_COST_LDS];
    for (int i = lane; i < 2 * 2 * LK_MAX_W; i += 64) (&s_row[0][0][0])[i] = 0;
    LK_SYNC();
    int score = 0, intra_mbs = 0, cost00 = 0;
    const int px = lane & 7, py = lane >> 3;
    for (int my = mb_h - 2; my > 0; my--)
        for (int mx = mb_w - 2; mx > 0; mx--) {
            const int xy = mx + my * mb_w;
            const ptrdiff_t off = 8 * ((ptrdiff_t)mx + (ptrdiff_t)my * stride);
            int bcost = LK_COST_MAX;
            if (!intra_only) {
                LK_SYNC();                                    // the previous block's readers are done with s_fe / s_mvc
                if (lane < 16) s_fe[(lane >> 1) * 4 + (lane & 1)] = *(const u32 *)(T.pl[0][0] + off + (ptrdiff_t)(lane >> 1) * stride + 4 * (lane & 1));
                LK_SYNC();
                const int fpx = (int)((const u8 *)s_fe)[py * 16 + px];
                MeLimits L;
                L.fmin0 = -8 * mx - 4; L.fmax0 = 8 * (mb_w - mx - 1) + 4; L.fmin1 = -8 * my - 4; L.fmax1 = 8 * (mb_h - my - 1) + 4;
                L.smin0 = 4 * (L.fmin0 - 8); L.smax0 = 4 * (L.fmax0 + 8); L.smin1 = 4 * (L.fmin1 - 8); L.smax1 = 4 * (L.fmax1 + 8);
                const ptrdiff_t poff = off + (ptrdiff_t)py * stride + px;
#define LK_TRY_BIDIR(ax_, ay_, bx_, by_, penalty_) do { \
                    const int r0_ = lk_ref_px(T.pl[1][0], T.pl[1][1], T.pl[1][2], T.pl[1][3], stride, poff, (ax_), (ay_)); \
                    const int r1_ = lk_ref_px(T.pl[2][0], T.pl[2][1], T.pl[2][2], T.pl[2][3], stride, poff, (bx_), (by_)); \
                    const int av_ = bipred_weight == 32 ? (r0_ + r1_ + 1) >> 1 : clip_u8((r0_ * bipred_weight + r1_ * (64 - bipred_weight) + 32) >> 6); \
                    const int c_ = (penalty_) + lk_satd8x8(fpx - av_, lane); \
                    if (bcost > c_) bcost = c_; } while (0)
                if (b_bidir) {
                    const int rx = MX_UNI((int)T.mvr[2 * xy]), ry = MX_UNI((int)T.mvr[2 * xy + 1]);
                    int ax = (rx * dist_scale_factor + 128) >> 8, ay = (ry * dist_scale_factor + 128) >> 8;
                    int bx = ax - rx, by = ay - ry;
                    ax = clip3(ax, L.smin0, L.smax0); ay = clip3(ay, L.smin1, L.smax1);
                    bx = clip3(bx, L.smin0, L.smax0); by = clip3(by, L.smin1, L.smax1);
                    LK_TRY_BIDIR(ax, ay, bx, by, 0);
                    if (ax | ay | bx | by) LK_TRY_BIDIR(0, 0, 0, 0, 0);
                }
                int mvx[2] = {0, 0}, mvy[2] = {0, 0};
                for (int l = 0; l < 1 + b_bidir; l++) {
                    int cost, vx, vy;
                    // (selects, not T.x[l]: a dynamically indexed member would put the whole task record into private memory)
                    i16 *const mv_l = l ? T.mv[1] : T.mv[0];
                    int *const mcost_l = l ? T.mcost[1] : T.mcost[0];
                    if (l ? T.do_search[1] : T.do_search[0]) {
                        // reverse-order predictors, slicetype.c:151-163: right, below, below-left, below-right (zero where absent)
                        const u32 *rc = s_row[l][my & 1], *rb = s_row[l][(my + 1) & 1];     // (LDS: indexing is free)
                        u32 cand[4] = {0, 0, 0, 0};
                        int n_mvc = 0;
                        if (mx < mb_w - 1) cand[n_mvc++] = rc[mx + 1];
                        if (my < mb_h - 1) {
                            cand[n_mvc++] = rb[mx];
                            if (mx > 0) cand[n_mvc++] = rb[mx - 1];
                            if (mx < mb_w - 1) cand[n_mvc++] = rb[mx + 1];
                        }
                        int cx[4], cy[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) { cx[k] = MX_UNI((int)(i16)(cand[k] & 0xffff)); cy[k] = MX_UNI((int)(i16)(cand[k] >> 16)); }
                        const int mvpx = max(min(cx[0], cx[1]), min(max(cx[0], cx[1]), cx[2]));     // x264_median_mv of the first three
                        const int mvpy = max(min(cy[0], cy[1]), min(max(cy[0], cy[1]), cy[2]));
                        if (lane < 4) { s_mvc[2 * lane] = (i16)cx[lane == 0 ? 0 : lane == 1 ? 1 : lane == 2 ? 2 : 3]; s_mvc[2 * lane + 1] = (i16)cy[lane == 0 ? 0 : lane == 1 ? 1 : lane == 2 ? 2 : 3]; }
                        LK_SYNC();
                        MxCtx c;
                        c.fe = (MX_LDS(u32))s_fe; c.fe_u = (MX_LDS(u8))s_fe; c.fe_v = (MX_LDS(u8))s_fe;
                        c.pl[0] = (MX_GLB(u8))((l ? T.pl[2][0] : T.pl[1][0]) + off); c.pl[1] = (MX_GLB(u8))((l ? T.pl[2][1] : T.pl[1][1]) + off);
                        c.pl[2] = (MX_GLB(u8))((l ? T.pl[2][2] : T.pl[1][2]) + off); c.pl[3] = (MX_GLB(u8))((l ? T.pl[2][3] : T.pl[1][3]) + off);
                        c.cu = c.pl[0]; c.cv = c.pl[0];
                        c.cost_g = (MX_GLB(i16))cost_g; c.cost_l = (MX_LDS(i16))s_costl; c.has_cost_l = true;
                        c.patch = (MX_LDS(u8))s_fe; c.has_patch = false; c.patch_on = false;
                        c.px0 = c.py0 = c.cx0 = c.cy0 = 0;
                        c.mvpx = mvpx; c.mvpy = mvpy; c.sy = stride; c.sc = stride; c.lane = lane;
                        c.set_block(8, 8, 0, 0);
                        MeOpts o;
                        o.method = method; o.me_range = me_range; o.subme = 4; o.chroma_me = 0; o.sad_only = 0;
                        int cmv;
                        cost = me_search_ref16(c, L, o, s_mvc, n_mvc, nullptr, vx, vy, cmv);
                        cost -= 2;                            // remove mvcost from skip mbs
                        if (vx | vy) cost += 5;
                        LK_SYNC();                            // s_mvc read; the row entry below is this block's own
                        if (lane == 0) {
                            s_row[l][my & 1][mx] = (u32)(u16)vx | ((u32)(u16)vy << 16);
                            *(u32 *)(mv_l + 2 * xy) = (u32)(u16)vx | ((u32)(u16)vy << 16);
                            mcost_l[xy] = cost;
                        }
                    } else {
                        vx = MX_UNI((int)mv_l[2 * xy]); vy = MX_UNI((int)mv_l[2 * xy + 1]); cost = MX_UNI(mcost_l[xy]);
                    }
                    if (l) { mvx[1] = vx; mvy[1] = vy; } else { mvx[0] = vx; mvy[0] = vy; }
                    bcost = min(bcost, cost);
                }
                if (b_bidir && (mvx[0] | mvy[0] | mvx[1] | mvy[1])) LK_TRY_BIDIR(mvx[0], mvy[0], mvx[1], mvy[1], 5);
            }
            if (!b_bidir) {                                   // no intra blocks in B frames
                const int icost = MX_UNI(T.intra[xy]);
                const int b_intra = icost < bcost;
                if (b_intra) bcost = icost;
                intra_mbs += b_intra; cost00 += icost;
            }
            score += bcost;
        }
    if (d1 != 0) score = score * 100 / (120 + bframe_bias);
    if (lane == 0) { out[4 * blockIdx.x] = score; out[4 * blockIdx.x + 1] = intra_mbs; out[4 * blockIdx.x + 2] = cost00; out[4 * blockIdx.x + 3] = 0; }
}

// Host side: resolve the tasks' slots to device pointers, stage them, launch.
extern "C" int x264hip_lookahead_cost_frames(x264hip_frame_ctx *c, const x264hip_look_slot *slots, int n_slots, const x264hip_look_task *tasks,
                                             int n_tasks, const x264hip_look_params *p, void *staging_host, void *tasks_dev, int32_t *out_dev)
{
    const int mb_w = c->d.mb_w, mb_h = c->d.mb_h, n = mb_w * mb_h;
    if (n_tasks <= 0) return 0;
    if (mb_w <= 2 || mb_h <= 2) { set_error("lookahead_cost_frames: frames of at most two macroblock rows / columns are scored edge and all (slicetype.c:292-297): not built"); return -1; }
    if (mb_w > LK_MAX_W) { set_error("lookahead_cost_frames: %d macroblocks per row, at most %d", mb_w, LK_MAX_W); return -1; }
    if (p->subme_param < 2 || p->lossless) { set_error("lookahead_cost_frames: mbcmp is SAD (subme < 2 or lossless); only the SATD lookahead is built"); return -1; }
    if (p->bframes < 0 || p->bframes > 16 || !staging_host || !tasks_dev || !out_dev || !p->cost_mv) { set_error("lookahead_cost_frames: bad arguments"); return -1; }
    LookTaskDev *st = (LookTaskDev *)staging_host;
    const int nd = p->bframes + 1;
    for (int i = 0; i < n_tasks; i++) {
        const x264hip_look_task &t = tasks[i];
        if (t.slot_b < 0 || t.slot_b >= n_slots || t.slot_p0 < 0 || t.slot_p0 >= n_slots || t.slot_p1 < 0 || t.slot_p1 >= n_slots || t.chain < 0 || t.chain >= c->batch ||
            t.d0 < 0 || t.d0 > nd || t.d1 < 0 || t.d1 > nd) { set_error("lookahead_cost_frames: task %d out of range", i); return -1; }
        const x264hip_look_slot *sl[3] = {&slots[t.slot_b], &slots[t.slot_p0], &slots[t.slot_p1]};
        LookTaskDev &d = st[i];
        for (int f = 0; f < 3; f++)
            for (int k = 0; k < 4; k++) d.pl[f][k] = sl[f]->pic->lowres[k] + c->bs_l * t.chain;
        // [batch][2][bframes + 1][n]
        const size_t per_chain = (size_t)2 * nd * n;
        d.mv[0] = t.d0 ? sl[0]->mv + ((size_t)t.chain * per_chain + (size_t)(0 * nd + t.d0 - 1) * n) * 2 : nullptr;
        d.mv[1] = t.d1 ? sl[0]->mv + ((size_t)t.chain * per_chain + (size_t)(1 * nd + t.d1 - 1) * n) * 2 : nullptr;
        d.mcost[0] = t.d0 ? sl[0]->mv_cost + (size_t)t.chain * per_chain + (size_t)(0 * nd + t.d0 - 1) * n : nullptr;
        d.mcost[1] = t.d1 ? sl[0]->mv_cost + (size_t)t.chain * per_chain + (size_t)(1 * nd + t.d1 - 1) * n : nullptr;
        d.mvr = t.d1 ? sl[2]->mv + ((size_t)t.chain * per_chain + (size_t)(0 * nd + t.d0 + t.d1 - 1) * n) * 2 : nullptr;
        if (t.d1 && t.d0 + t.d1 > nd) { set_error("lookahead_cost_frames: task %d spans %d frames, more than bframes + 1", i, t.d0 + t.d1); return -1; }
        d.intra = sl[0]->intra_cost + (size_t)t.chain * n;
        d.d0 = t.d0; d.d1 = t.d1; d.do_search[0] = t.do_search[0]; d.do_search[1] = t.do_search[1];
    }
    // tasks_dev == staging_host: the kernel reads the records in place, from pinned host memory (168 bytes per task, once) -- for callers that
    // keep the device full, where even a small upload's copy kernel would wait for a wave slot
    if (tasks_dev != staging_host) HIPCHK(hipMemcpyAsync(tasks_dev, st, sizeof(LookTaskDev) * (size_t)n_tasks, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_look_cost, dim3(n_tasks), dim3(64), 0, c->stream, (const LookTaskDev *)tasks_dev, mb_w, mb_h, slots[0].pic->stride_lowres,
                       p->me_method < 1 ? p->me_method : 1, p->me_range, p->weighted_bipred, p->bframe_bias, p->cost_mv + p->cost_mv_range, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" size_t x264hip_lookahead_task_bytes(void) { return sizeof(LookTaskDev); }
