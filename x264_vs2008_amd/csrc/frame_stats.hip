// frame_stats.hip -- h->stat.frame's terms that x264_encoder_encode reads after a P slice (the post-encode scene cut, R/encoder/encoder.c:1603-1699)
// as one reduction per chain over the state the sweep left, and the decision itself in host C with the reference's float expression.
#include "frame_internal.h"
#include "../../include/x264hip_stream.h"

using x264hip::set_error;

// one block per chain: macroblock counts by type (I_16x16 + I_8x8 + I_4x4, P_L0 + P_8x8, P_SKIP), the analysed macroblocks and their cost sums
__global__ __launch_bounds__(256) void k_frame_stats(const signed char *mb_type, const int *cost_intra, const int *cost_inter, int n_mb, x264hip_frame_stat *out)
{
    __shared__ unsigned long long s_intra, s_inter;
    __shared__ int s_cnt[4];
    const int bz = blockIdx.x, t = threadIdx.x;
    if (t == 0) { s_intra = 0; s_inter = 0; s_cnt[0] = s_cnt[1] = s_cnt[2] = s_cnt[3] = 0; }
    __syncthreads();
    unsigned long long ci = 0, cp = 0;
    int n_i = 0, n_p = 0, n_s = 0, n_a = 0;
    const size_t base = (size_t)bz * n_mb;
    for (int mb = t; mb < n_mb; mb += 256) {
        const int ty = mb_type[base + mb], a = cost_intra[base + mb], b = cost_inter[base + mb];
        n_i += ty >= 0 && ty <= 2; n_p += ty == 4 || ty == 5; n_s += ty == 6;
        if (a | b) { n_a++; ci += (unsigned long long)(unsigned)a; cp += (unsigned long long)(unsigned)b; }      // analysed macroblocks carry a positive intra cost (analyse.c:2392-2404)
    }
    atomicAdd(&s_intra, ci); atomicAdd(&s_inter, cp);
    atomicAdd(&s_cnt[0], n_i); atomicAdd(&s_cnt[1], n_p); atomicAdd(&s_cnt[2], n_s); atomicAdd(&s_cnt[3], n_a);
    __syncthreads();
    if (t == 0) {
        x264hip_frame_stat r;
        r.intra_cost = (int64_t)s_intra; r.inter_cost = (int64_t)s_inter;
        r.mb_i = s_cnt[0]; r.mb_p = s_cnt[1]; r.mb_skip = s_cnt[2]; r.mbs_analysed = s_cnt[3];
        out[bz] = r;
    }
}

extern "C" int x264hip_frame_stats(x264hip_frame_ctx *c, const x264hip_mb_state *st, x264hip_frame_stat *out_dev)
{
    if (!c || !st || !out_dev || !st->mb_type || !st->cost_intra || !st->cost_inter) { set_error("frame_stats: bad argument"); return -1; }
    hipLaunchKernelGGL(k_frame_stats, dim3(c->batch), dim3(256), 0, c->stream, (const signed char *)st->mb_type, (const int *)st->cost_intra,
                       (const int *)st->cost_inter, c->d.mb_w * c->d.mb_h, out_dev);
    if (hipGetLastError() != hipSuccess) { set_error("frame_stats: launch failed"); return -1; }
    return 0;
}

// encoder.c:1603-1644: 1 if x264_encoder_encode would code this P frame again as I / IDR (or re-type the B frames before it)
extern "C" int x264hip_scenecut_post(const x264hip_frame_stat *s, int i_mb, int i_gop_size, int scenecut_threshold, int keyint_min, int keyint_max)
{
    if (scenecut_threshold < 0) return 0;
    int64_t i_inter_cost = s->inter_cost, i_intra_cost = s->intra_cost;
    float f_bias;
    const float f_thresh_max = (float)(scenecut_threshold / 100.0);
    float f_thresh_min = f_thresh_max * keyint_min / (keyint_max * 4);
    if (keyint_min == keyint_max) f_thresh_min = f_thresh_max;
    if (s->mbs_analysed > 0) i_intra_cost = i_intra_cost * i_mb / s->mbs_analysed;
    if (i_gop_size < keyint_min / 4) f_bias = f_thresh_min / 4;
    else if (i_gop_size <= keyint_min) f_bias = f_thresh_min * i_gop_size / keyint_min;
    else f_bias = f_thresh_min + (f_thresh_max - f_thresh_min) * (i_gop_size - keyint_min) / (keyint_max - keyint_min);
    f_bias = (float)((double)f_bias < 1.0 ? (double)f_bias : 1.0);
    return s->mbs_analysed > 0 && (double)i_inter_cost >= (1.0 - (double)f_bias) * (double)i_intra_cost;
}
