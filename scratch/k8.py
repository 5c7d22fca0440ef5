import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)

rep('''template <int WPE, bool LL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_slice_sweep(SwArgs a, SwRefs refs)
{
    __builtin_assume(a.lossless == (int)LL);           // the host launches the matching variant; do not write to `a` (a modified
                                                        // kernel argument is copied to scratch memory whole)
    __shared__ SwLds s;
    const int lane_id = threadIdx.x, lane = lane_id;
    const int bz = blockIdx.x % a.batch_pad, mby = blockIdx.x / a.batch_pad;
    if (bz >= a.batch) return;''','''// RD: the raster-order variant.  With the RD levels (the trial encodes are priced against the live CABAC contexts), trellis
// (same) or adaptive quantisation (a macroblock's QP follows from the previous one's, R/encoder/ratecontrol.c:263-264) a slice is
// one serial chain of macroblocks; one wavefront then owns a whole frame of one chain and walks it in raster order, rows and all,
// and the entropy coder (cabac_dev.h) runs inside the loop exactly where x264_slice_write has it.  Throughput comes from the
// number of frames in flight (grid = batch), not from a wavefront schedule inside the frame.
static __device__ const u8 d_lambda_tab[52] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
                                               6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};
static __device__ const int d_lambda2_tab[52] = {14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1161, 1462, 1843, 2322, 2925,
    3686, 4644, 5851, 7372, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628, 117964,
    148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436};
static __device__ const u8 d_chroma_qp[52] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                                              29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

template <int WPE, bool LL = false, bool RD = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_slice_sweep(SwArgs a, SwRefs refs, SwRd rd)
{
    __builtin_assume(a.lossless == (int)LL);           // the host launches the matching variant; do not write to `a` (a modified
                                                        // kernel argument is copied to scratch memory whole)
    __shared__ SwLds s;
    __shared__ typename std::conditional<RD, SwLdsRd, SwLdsNone>::type sr_;
    SwLdsRd &sr = *(SwLdsRd *)&sr_;                     // only touched when RD
    const int lane_id = threadIdx.x, lane = lane_id;
    const int bz = RD ? (int)blockIdx.x : (int)(blockIdx.x % a.batch_pad), mby0 = RD ? 0 : (int)(blockIdx.x / a.batch_pad);
    if (bz >= a.batch) return;''')

rep('''    const MeOpts mo = {a.me_method, a.me_range, a.subme, a.chroma_me, a.lossless};
    {   // tables that every macroblock of the row reads: into LDS once
        const int cat = lane >> 4, i = lane & 15, q = cat < 2 ? a.qp : a.qpc;
        s.qmf[cat][i] = a.q4mf[(cat * 52 + q) * 16 + i]; s.qbias[cat][i] = a.q4bias[(cat * 52 + q) * 16 + i];
        s.qdq[cat][i] = a.dq4[cat * 96 + (q % 6) * 16 + i];
        if (is_p)
            for (int k = lane; k < 2 * MX_COST_LDS + 1; k += 64) s.costl[k] = a.cost_mv[a.cost_center - MX_COST_LDS + k];
        if (a.nr) { s.nr_off8[lane] = a.nr_offset[(size_t)bz * 128 + 64 + lane]; if (lane < 16) s.nr_off4[lane] = a.nr_offset[(size_t)bz * 128 + lane]; }
        if (lane < 48) s.p4lut[lane] = ((const u32 *)&c_plut4)[lane];
        for (int k = lane; k < 192; k += 64) s.p8lut[k] = ((const u32 *)&c_plut8)[k];
        if (a.transform8x8)
            for (int c8 = 0; c8 < 2; c8++) {
                s.q8mf[c8][lane] = a.q8mf[(c8 * 52 + a.qp) * 64 + lane]; s.q8bias[c8][lane] = a.q8bias[(c8 * 52 + a.qp) * 64 + lane];
                s.q8dq[c8][lane] = a.dq8[c8 * 384 + (a.qp % 6) * 64 + lane];
            }
    }
    WAVE_SYNC();
''','''    const MeOpts mo = {a.me_method, a.me_range, a.subme, a.chroma_me, a.lossless};
    SwQp Q = {a.qp, a.qpc, a.lambda, d_lambda2_tab[a.qp], a.chroma_skip_thresh};
    const i16 *cost_g = a.cost_mv + a.cost_center;     // p_cost_mv of the current QP, centred
    // tables of the current QP: into LDS (once per slice; again whenever adaptive quantisation changes the macroblock's QP)
    auto load_qp_tables = [&](int lane) {
        const int cat = lane >> 4, i = lane & 15, q = cat < 2 ? Q.qp : Q.qpc;
        s.qmf[cat][i] = a.q4mf[(cat * 52 + q) * 16 + i]; s.qbias[cat][i] = a.q4bias[(cat * 52 + q) * 16 + i];
        s.qdq[cat][i] = a.dq4[cat * 96 + (q % 6) * 16 + i];
        if (is_p)
            for (int k = lane; k < 2 * MX_COST_LDS + 1; k += 64) s.costl[k] = cost_g[k - MX_COST_LDS];
        if (a.transform8x8)
            for (int c8 = 0; c8 < 2; c8++) {
                s.q8mf[c8][lane] = a.q8mf[(c8 * 52 + Q.qp) * 64 + lane]; s.q8bias[c8][lane] = a.q8bias[(c8 * 52 + Q.qp) * 64 + lane];
                s.q8dq[c8][lane] = a.dq8[c8 * 384 + (Q.qp % 6) * 64 + lane];
            }
        if constexpr (RD) {
            if (rd.trellis) {
                sr.unq4[cat][i] = rd.unq4[(cat * 52 + q) * 16 + i];
                if (a.transform8x8) for (int c8 = 0; c8 < 2; c8++) sr.unq8[c8][lane] = rd.unq8[(c8 * 52 + Q.qp) * 64 + lane];
            }
        }
    };
    load_qp_tables(lane);
    {
        if (a.nr) { s.nr_off8[lane] = a.nr_offset[(size_t)bz * 128 + 64 + lane]; if (lane < 16) s.nr_off4[lane] = a.nr_offset[(size_t)bz * 128 + lane]; }
        if (lane < 48) s.p4lut[lane] = ((const u32 *)&c_plut4)[lane];
        for (int k = lane; k < 192; k += 64) s.p8lut[k] = ((const u32 *)&c_plut8)[k];
    }
    // the entropy coder of this chain's slice (x264_slice_write, R/encoder/encoder.c:1155-1165)
    DCabac cb = {0, 0x1FE, -1, 0, nullptr, 0};
    u8 *payload0 = nullptr;
    int last_qp = a.qp, last_dqp = 0, prev_coded = 0, intra_before = 0;      // h->mb.i_last_qp / i_last_dqp; the previous macroblock "has coefficients"
    if constexpr (RD) {
        if (rd.write) {
            payload0 = rd.payload + (size_t)bz * rd.payload_cap + 64;
            cb.p = payload0;
            for (int k = lane; k < 460; k += 64) sr.cabac[k] = (u8)cd_context_init_one(k, a.slice_type, a.qp, rd.cabac_init_idc);
        }
    }
    WAVE_SYNC();
''')

rep('''    // the left neighbour = this wave's previous macroblock
    int left_type = -1, left_ref = -2, left_mvx = 0, left_mvy = 0, row_intra = 0;
    u32 pre_y;
    u8 pre_u, pre_v;
    {''','''  for (int mby = mby0; mby < (RD ? a.mb_h : mby0 + 1); mby++) {
    if constexpr (RD) {
        // this wave's own stores of the row above (pixels, types, vectors ...) must be what its loads see
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // the left neighbour = this wave's previous macroblock
    int left_type = -1, left_ref = -2, left_mvx = 0, left_mvy = 0, row_intra = 0;
    int left_cbp = -1, left_cpm = 0, left_t8 = 0;          // (RD) h->mb.cbp / chroma_pred_mode / mb_transform_size of the left macroblock
    u32 pre_y;
    u8 pre_u, pre_v;
    {''')

rep('''        // ---- wait for the row above: left-top, top and top-right neighbours finished ----
        if (mby > 0) {''','''        // ---- wait for the row above: left-top, top and top-right neighbours finished ----
        if (!RD && mby > 0) {''')
open(p,'w').write(s)
print("ok")
