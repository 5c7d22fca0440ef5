// frame_me_exact.hip -- FRAME LEVEL, part 6: the reference's own motion search,
// x264_me_search_ref + refine_subpel for the 16x16 block (R/encoder/me.c:156-778), for every
// macroblock of a frame and every reference in one launch, inside the loop of
// x264_mb_analyse_inter_p16x16 (R/encoder/analyse.c:1077-1127): predictor tests (both the
// subme < 3 and >= 3 forms), DIA or HEX walk with square refine, half-pel and quarter-pel diamond
// refinement with SAD / SATD and chroma ME, the half-pel early-termination threshold carried from
// one reference to the next, and the first-minimum choice of the best reference.  Predictors
// (mvp, mvc) are inputs: deriving them from neighbours is the caller's serial part.
//
// The walk is data dependent, so it cannot be precomputed; it is run, one wavefront per
// macroblock, exactly in the reference's candidate order:
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
#include "me_exact.h"
#include "frame_internal.h"

using namespace x264hip;

#define MX_WAVES 4
#define MX_MAX_REFS 8

struct MxRefs {
    const u8 *y[MX_MAX_REFS][4];
    const u8 *u[MX_MAX_REFS], *v[MX_MAX_REFS];
};
struct MxGeom {
    int mb_w, mb_h, sy, sc, n_refs, method, me_range, subme, chroma_me, mv_range, cost_center;
    size_t bs_y, bs_c;
    int ref_cost[MX_MAX_REFS];
};

__global__ __launch_bounds__(64 * MX_WAVES) void k_me_search16(const u8 *__restrict__ fy, const u8 *__restrict__ fu, const u8 *__restrict__ fv,
                                                               MxRefs refs, MxGeom g, const i16 *__restrict__ cost_mv,
                                                               const i16 *__restrict__ mvp_in, const i16 *__restrict__ mvc_in,
                                                               const u8 *__restrict__ n_mvc_in, i16 *__restrict__ out_mv,
                                                               int *__restrict__ out_cost, int *__restrict__ best_out)
{
    __shared__ u32 s_fe[MX_WAVES][64];
    __shared__ __attribute__((aligned(16))) u8 s_fc[MX_WAVES][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mb = xcd_band_order(blockIdx.x, gridDim.x) * MX_WAVES + wave;
    if (mb >= g.mb_w * g.mb_h) return;
    const size_t bz = blockIdx.y, nmb = (size_t)g.mb_w * g.mb_h;
    fy += g.bs_y * bz; fu += g.bs_c * bz; fv += g.bs_c * bz;
    mvp_in += 2 * g.n_refs * nmb * bz; mvc_in += 16 * g.n_refs * nmb * bz; n_mvc_in += g.n_refs * nmb * bz;
    out_mv += 2 * g.n_refs * nmb * bz; out_cost += g.n_refs * nmb * bz; best_out += 4 * nmb * bz;
    const int mbx = mb % g.mb_w, mby = mb / g.mb_w;
    const ptrdiff_t oy = (ptrdiff_t)16 * mby * g.sy + 16 * mbx, oc = (ptrdiff_t)8 * mby * g.sc + 8 * mbx;
    s_fe[wave][lane] = *(const u32 *)(fy + oy + (ptrdiff_t)(lane >> 2) * g.sy + 4 * (lane & 3));
    s_fc[wave][lane] = fu[oc + (ptrdiff_t)(lane >> 3) * g.sc + (lane & 7)];
    s_fc[wave][64 + lane] = fv[oc + (ptrdiff_t)(lane >> 3) * g.sc + (lane & 7)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const MeLimits L = me_limits(mbx, mby, g.mb_w, g.mb_h, g.mv_range);
    const MeOpts o = {g.method, g.me_range, g.subme, g.chroma_me, 0};
    int thresh = 0x7fffffff, bestc = 0x7fffffff, best_r = 0, best_x = 0, best_y = 0;

    for (int r = 0; r < g.n_refs; r++) {     // the loop of x264_mb_analyse_inter_p16x16, R/encoder/analyse.c:1090-1127
        MxCtx c;
        c.fe = (MX_LDS(u32))s_fe[wave]; c.fe_u = (MX_LDS(u8))s_fc[wave]; c.fe_v = (MX_LDS(u8))(s_fc[wave] + 64); c.sy = g.sy; c.sc = g.sc; c.lane = lane; c.set_block(16, 16, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) c.pl[k] = (MX_GLB(u8))(refs.y[r][k] + g.bs_y * bz + oy);
        c.cu = (MX_GLB(u8))(refs.u[r] + g.bs_c * bz + oc); c.cv = (MX_GLB(u8))(refs.v[r] + g.bs_c * bz + oc);
        const i16 *mvp = mvp_in + ((size_t)mb * g.n_refs + r) * 2;
        const int mvpx = mvp[0], mvpy = mvp[1];
        c.cost_g = (MX_GLB(i16))(cost_mv + g.cost_center); c.cost_l = (MX_LDS(i16))s_fe[wave]; c.has_cost_l = false; c.patch = (MX_LDS(u8))s_fe[wave]; c.has_patch = false; c.patch_on = false; c.mvpx = mvpx; c.mvpy = mvpy;
        thresh -= g.ref_cost[r];
        int mvx, mvy, cost_mv_out;
        int mcost = me_search_ref16(c, L, o, mvc_in + ((size_t)mb * g.n_refs + r) * 16, n_mvc_in[(size_t)mb * g.n_refs + r],
                                    g.n_refs > 1 ? &thresh : nullptr, mvx, mvy, cost_mv_out);
        mcost += g.ref_cost[r];
        thresh += g.ref_cost[r];
        if (lane == 0) {
            out_mv[((size_t)mb * g.n_refs + r) * 2] = (i16)mvx; out_mv[((size_t)mb * g.n_refs + r) * 2 + 1] = (i16)mvy;
            out_cost[(size_t)mb * g.n_refs + r] = mcost;
        }
        if (mcost < bestc) { bestc = mcost; best_r = r; best_x = mvx; best_y = mvy; }
    }
    if (lane == 0) {
        best_out[4 * (size_t)mb] = best_r; best_out[4 * (size_t)mb + 1] = best_x; best_out[4 * (size_t)mb + 2] = best_y;
        best_out[4 * (size_t)mb + 3] = bestc;
    }
}


extern "C" int x264hip_me_search16_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *const *refs,
                                         int n_refs, const x264hip_me16_params *p, int16_t *out_mv_dev, int32_t *out_cost_dev,
                                         int32_t *best_dev)
{
    if (n_refs < 1 || n_refs > MX_MAX_REFS) { set_error("me_search16: %d references (1..%d)", n_refs, MX_MAX_REFS); return -1; }
    if (p->me_method < 0 || p->me_method > 3 || (p->me_method == 3 && p->subme < 1)) { set_error("me_search16: method %d not built (0 DIA, 1 HEX, 2 UMH, 3 ESA with subme >= 1)", p->me_method); return -1; }
    if (p->subme < 0 || p->subme > 9 || !p->cost_mv || !p->mvp || !p->mvc || !p->n_mvc) { set_error("me_search16: bad parameters"); return -1; }
    MxRefs t;
    for (int i = 0; i < MX_MAX_REFS; i++) {
        const x264hip_picture *r = refs[i < n_refs ? i : 0];
        for (int k = 0; k < 4; k++) t.y[i][k] = r->filtered[k];
        t.u[i] = r->plane[1]; t.v[i] = r->plane[2];
    }
    MxGeom g;
    g.mb_w = c->d.mb_w; g.mb_h = c->d.mb_h; g.sy = c->d.stride_y; g.sc = c->d.stride_c; g.n_refs = n_refs;
    g.method = p->me_method; g.me_range = p->me_range; g.subme = p->subme; g.chroma_me = p->chroma_me;
    g.mv_range = p->mv_range > 0 ? p->mv_range : 512; g.cost_center = p->cost_mv_range;
    g.bs_y = c->bs_y; g.bs_c = c->bs_c;
    for (int i = 0; i < MX_MAX_REFS; i++) g.ref_cost[i] = i < n_refs ? p->ref_cost[i] : 0;
    int n = g.mb_w * g.mb_h;
    hipLaunchKernelGGL(k_me_search16, dim3((n + MX_WAVES - 1) / MX_WAVES, c->batch), dim3(64 * MX_WAVES), 0, c->stream,
                       fenc->plane[0], fenc->plane[1], fenc->plane[2], t, g, p->cost_mv, p->mvp, p->mvc, p->n_mvc,
                       out_mv_dev, out_cost_dev, best_dev);
    HIPCHK(hipGetLastError());
    return 0;
}
