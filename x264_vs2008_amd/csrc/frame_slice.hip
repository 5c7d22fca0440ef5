// frame_slice.hip -- host side of the macroblock sweep (x264hip_slice_sweep_frame ...) and the wavefront-schedule kernel variants.
// The kernel itself lives in slice_kernel.h; its raster-order variant is instantiated in frame_slice_rd.hip (a translation unit of
// its own so that the two compile side by side).
#include <mutex>
#include <unordered_map>
#include <vector>
#include "slice_kernel.h"
#include "x264hip_lookahead.h"

void x264hip_launch_slice_rd(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream);
void x264hip_launch_slice_b(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream);
void x264hip_launch_slice_bt(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream);
void x264hip_launch_slice_rf(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream);
void x264hip_launch_slice_rd_ch(const SwDesc *tab, int n, hipStream_t stream);
void x264hip_launch_slice_bt_ch(const SwDesc *tab, int n, hipStream_t stream);
void x264hip_launch_slice_rf_ch(const SwDesc *tab, int n, hipStream_t stream);

// b_fast_intra's raster-order term, settled once the frame is complete: macroblocks whose analysis went on without
// knowing it (it could not change their type) recorded the statistics term for the other answer in cost_alt.
// One wave per chain, 64 macroblocks per step, running count of intra macroblocks via ballot.
__global__ __launch_bounds__(64) void k_resolve_fast_intra(const signed char *mb_type, int *cost_intra, const int *cost_alt, int n_mb)
{
    const int lane = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * n_mb;
    int before = 0;                                    // intra macroblocks before this group of 64
    for (int m0 = 0; m0 < n_mb; m0 += 64) {
        const int mb = m0 + lane;
        const int t = mb < n_mb ? (int)mb_type[base + mb] : 99;
        const unsigned long long im = __ballot(t >= 0 && t <= 3);
        const int count = before + __popcll(im & ((1ull << lane) - 1));
        if (mb < n_mb) {
            const int alt = cost_alt[base + mb];
            if (alt >= 0 && !(mb < 3 * count)) cost_intra[base + mb] = alt;     // b_fast_intra was 1
        }
        before += __popcll(im);
    }
}


// ------------------------------------------------------------------ host
static const int k_lambda_tab[52] = {    // x264_lambda_tab, R/encoder/analyse.c:140-149
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6,
    6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};
static const int k_lambda2_tab[52] = {   // x264_lambda2_tab, :151-159
    14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1161, 1462, 1843, 2322, 2925,
    3686, 4644, 5851, 7372, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628, 117964,
    148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436};
static const uint8_t k_chroma_qp[52] = {  // i_chroma_qp_table, R/common/macroblock.h
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};

extern "C" void x264hip_mb_state_free(x264hip_frame_ctx *c, x264hip_mb_state *st);
extern "C" int x264hip_mb_state_alloc_ex(x264hip_frame_ctx *c, x264hip_mb_state *st, int flags);
extern "C" int x264hip_mb_state_alloc(x264hip_frame_ctx *c, x264hip_mb_state *st) { return x264hip_mb_state_alloc_ex(c, st, 0); }
extern "C" int x264hip_mb_state_alloc_ex(x264hip_frame_ctx *c, x264hip_mb_state *st, int flags)
{
    const size_t n = (size_t)c->d.mb_w * c->d.mb_h * c->batch;
    const bool no_levels = flags & X264HIP_STATE_NO_LEVELS;     // 816 of a macroblock's 1184 bytes: nobody reads them when the sweep writes the payload itself
    memset(st, 0, sizeof(*st));
    struct { void **p; size_t bytes; } items[] = {
        {(void **)&st->mb_type, n}, {(void **)&st->partition, n}, {(void **)&st->sub_partition, 4 * n}, {(void **)&st->ref, 4 * n}, {(void **)&st->i4mode, 16 * n},
        {(void **)&st->i16mode, n}, {(void **)&st->chroma_mode, n}, {(void **)&st->qp, n}, {(void **)&st->t8, n},
        {(void **)&st->mv, 64 * n}, {(void **)&st->mvr, 4 * SW_MAX_REFS * n}, {(void **)&st->cbp, 2 * n}, {(void **)&st->nnz, 27 * n},
        {(void **)&st->luma, 512 * n}, {(void **)&st->luma_dc, 32 * n}, {(void **)&st->chroma_dc, 16 * n}, {(void **)&st->chroma_ac, 256 * n},
        {(void **)&st->cost_intra, 4 * n}, {(void **)&st->cost_inter, 4 * n}, {(void **)&st->cost_intra_alt, 4 * n},
        {(void **)&st->progress, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1)}, {(void **)&st->mvd, 64 * n},
        {(void **)&st->mv1, 64 * n}, {(void **)&st->ref1, 4 * n}, {(void **)&st->mvr1, 4 * n}, {(void **)&st->mvd1, 64 * n}, {(void **)&st->skipbp, n}};
    for (auto &it : items) {
        if (no_levels && (it.p == (void **)&st->luma || it.p == (void **)&st->luma_dc || it.p == (void **)&st->chroma_dc || it.p == (void **)&st->chroma_ac)) continue;
        if (hipMalloc(it.p, it.bytes) != hipSuccess || zero_async(*it.p, it.bytes, c->stream) != 0) {
            set_error("mb_state_alloc: %zu bytes", it.bytes);
            x264hip_mb_state_free(c, st);              // what was allocated so far
            return -1;
        }
    }
    return 0;
}
extern "C" void x264hip_mb_state_free(x264hip_frame_ctx *c, x264hip_mb_state *st)
{
    (void)c;
    void *ps[] = {st->mb_type, st->partition, st->sub_partition, st->ref, st->i4mode, st->i16mode, st->chroma_mode, st->qp, st->t8, st->mv, st->mvr, st->cbp,
                  st->nnz, st->luma, st->luma_dc, st->chroma_dc, st->chroma_ac, st->cost_intra, st->cost_inter, st->cost_intra_alt, st->progress, st->mvd,
                  st->mv1, st->ref1, st->mvr1, st->mvd1, st->skipbp};
    for (void *p : ps) if (p) (void)hipFree(p);
    memset(st, 0, sizeof(*st));
}

// x264_noise_reduction_update, R/encoder/macroblock.c:890-911: one block per chain, thread = cat * 64 + coefficient
static __constant__ u16 c_nr_w4[16] = {800, 320, 800, 320, 320, 128, 320, 128, 800, 320, 800, 320, 320, 128, 320, 128};        // x264_dct4_weight2_tab
static __constant__ u16 c_nr_w8k[6] = {256, 201, 656, 227, 410, 363};                                                     // x264_dct8_weight2_tab's W(i)
static __constant__ u8 c_nr_k8[32] = {0, 3, 4, 3, 0, 3, 4, 3, 3, 1, 5, 1, 3, 1, 5, 1, 4, 5, 2, 5, 4, 5, 2, 5, 3, 1, 5, 1, 3, 1, 5, 1};
__global__ __launch_bounds__(128) void k_nr_update(u32 *sum, u32 *count, u16 *offset, int strength)
{
    const int cat = threadIdx.x >> 6, i = threadIdx.x & 63, size = cat ? 64 : 16;
    sum += (size_t)blockIdx.x * 128; count += (size_t)blockIdx.x * 2; offset += (size_t)blockIdx.x * 128;
    u32 cnt = count[cat], sv = i < size ? sum[cat * 64 + i] : 0;
    const bool halve = cnt > (cat ? (1u << 16) : (1u << 18));
    __syncthreads();                                   // everyone has read the count before it is rewritten
    if (halve) { sv >>= 1; cnt >>= 1; if (i < size) sum[cat * 64 + i] = sv; if (i == 0) count[cat] = cnt; }
    if (i < size) {
        const unsigned long long w = cat ? c_nr_w8k[c_nr_k8[i & 31]] : c_nr_w4[i];
        offset[cat * 64 + i] = (u16)(((unsigned long long)strength * cnt + sv / 2) / ((unsigned long long)sv * w / 256 + 1));
    }
}
extern "C" int x264hip_nr_state_alloc(x264hip_frame_ctx *c, x264hip_nr_state *nr)
{
    memset(nr, 0, sizeof(*nr));
    const size_t B = (size_t)c->batch;
    HIPCHK(hipMalloc((void **)&nr->sum, B * 128 * 4)); HIPCHK(hipMalloc((void **)&nr->count, B * 2 * 4)); HIPCHK(hipMalloc((void **)&nr->offset, B * 128 * 2));
    HIPCHK(hipMemsetAsync(nr->sum, 0, B * 128 * 4, c->stream)); HIPCHK(hipMemsetAsync(nr->count, 0, B * 2 * 4, c->stream));
    HIPCHK(hipMemsetAsync(nr->offset, 0, B * 128 * 2, c->stream));
    return 0;
}
extern "C" void x264hip_nr_state_free(x264hip_frame_ctx *c, x264hip_nr_state *nr)
{
    (void)c;
    if (nr->sum) (void)hipFree(nr->sum);
    if (nr->count) (void)hipFree(nr->count);
    if (nr->offset) (void)hipFree(nr->offset);
    memset(nr, 0, sizeof(*nr));
}
extern "C" int x264hip_noise_reduction_update(x264hip_frame_ctx *c, const x264hip_nr_state *nr, int noise_reduction)
{
    if (!nr || !nr->sum || !nr->count || !nr->offset) { set_error("noise_reduction_update: no state"); return -1; }
    hipLaunchKernelGGL(k_nr_update, dim3((unsigned)c->batch), dim3(128), 0, c->stream, nr->sum, nr->count, nr->offset, noise_reduction);
    HIPCHK(hipGetLastError());
    return 0;
}

struct ChainAux { hipStream_t stream = nullptr; hipEvent_t ready = nullptr, done = nullptr; };
// The three argument structures of one sweep launch from the ABI's description of it, and which kernel codes it
enum { SW_KIND_PLAIN = 0, SW_KIND_RD, SW_KIND_RF, SW_KIND_B, SW_KIND_BT };
static void sweep_note_frame(const x264hip_slice_params *p, int n_refs, x264hip_mb_state *out);
static int sweep_build(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *const *refs, int n_refs,
                       x264hip_picture *recon, const x264hip_slice_params *p, const x264hip_mb_state *l0,
                       x264hip_mb_state *out, SwArgs &a, SwRefs &t, SwRd &r, int &kind)
{
    const bool is_b = p->slice_type == 1, is_p = p->slice_type == 0 || is_b;     // is_p: "has list 0" in what follows
    if (p->slice_type != 0 && p->slice_type != 1 && p->slice_type != 2) { set_error("slice_sweep: slice type %d (0 P, 1 B, 2 I)", p->slice_type); return -1; }
    if (is_p && (n_refs < 1 || n_refs > SW_MAX_REFS)) { set_error("slice_sweep: %d references (1..%d)", n_refs, SW_MAX_REFS); return -1; }
    const x264hip_slice_b *pb = is_b ? p->b : nullptr;
    if (is_b) {
        if (!pb || !pb->fref1 || !pb->l1_state || !p->rd) { set_error("slice_sweep: a B slice needs x264hip_slice_params.b (list 1) and .rd (the raster variant)"); return -1; }
        if ((!pb->direct_spatial || pb->direct_score) && !p->rd->stale) { set_error("slice_sweep: temporal direct prediction needs x264hip_slice_rd.stale (in every sweep of the chain)"); return -1; }
        if (p->subme < 2 || p->subme > 8 || !p->rd->write || !p->cabac) { set_error("slice_sweep: B slices are built for subme 2..8 with the CABAC writer in the loop (subme 9 refines a B macroblock's vectors by RD: x264_me_refine_bidir_rd, not built)"); return -1; }
        if (p->noise_reduction || p->lossless) { set_error("slice_sweep: B slices with --nr / lossless are not built"); return -1; }
        if (!out->mv1 || !pb->l1_state->mb_type) { set_error("slice_sweep: mb_state without list-1 arrays"); return -1; }
    }
    if (p->qp < 0 || p->qp > 51) { set_error("slice_sweep: qp out of range"); return -1; }
    if (is_p && p->subme >= 1) {        /* a source-only picture (x264hip_picture_alloc_source) has no half-pel planes: not a reference */
        for (int i = 0; i < n_refs; i++)
            if (!refs[i] || !refs[i]->filtered[1] || !refs[i]->filtered[2] || !refs[i]->filtered[3]) { set_error("slice_sweep: reference %d has no half-pel planes (a source-only picture?)", i); return -1; }
        if (is_b && (!pb->fref1->filtered[1] || !pb->fref1->filtered[2] || !pb->fref1->filtered[3])) { set_error("slice_sweep: the list-1 reference has no half-pel planes (a source-only picture?)"); return -1; }
    }
    const x264hip_slice_rd *prd = p->rd;
    const int mbrd = (p->subme - is_b >= 6) + (p->subme - is_b >= 8);       /* one level less in a B slice, R/encoder/analyse.c:222-225 */
    if (p->subme < 0 || p->subme > 9) { set_error("slice_sweep: subme %d", p->subme); return -1; }
    if (mbrd && (!prd || !prd->write || !p->cabac)) { set_error("slice_sweep: subme %d prices its trial encodes against the live CABAC contexts: it needs x264hip_slice_params.rd with write = 1 and cabac = 1", p->subme); return -1; }
    if (prd) {
        if (prd->write && !p->cabac) { set_error("slice_sweep: the in-loop entropy coder is CABAC only"); return -1; }
        if (prd->write && (!prd->payload || !prd->payload_len || prd->payload_cap < SW_MB_BYTES_MAX + 128)) { set_error("slice_sweep: payload buffers missing or smaller than one macroblock's worst case (%d bytes)", SW_MB_BYTES_MAX + 128); return -1; }
        if (prd->trellis && (!prd->write || !prd->unquant4_mf || (p->transform8x8 && !prd->unquant8_mf))) { set_error("slice_sweep: trellis needs write = 1 and the unquant tables"); return -1; }
        if (prd->trellis < 0 || prd->trellis > 2) { set_error("slice_sweep: trellis %d", prd->trellis); return -1; }
        if (prd->aq_offset && !prd->cost_mv_all) { set_error("slice_sweep: adaptive quantisation needs cost_mv_all"); return -1; }
        if (p->lossless) { set_error("slice_sweep: lossless is not built in the raster variant"); return -1; }
        if (mbrd && (p->analyse_inter & 0x20)) { set_error("slice_sweep: sub-8x8 partitions with the RD levels not built (their partial bit counts read cache entries the previous macroblock left)"); return -1; }
        if (!out->mvd) { set_error("slice_sweep: mb_state without mvd"); return -1; }
    }
    if (!out->luma && !(prd && prd->write)) { set_error("slice_sweep: an mb_state without level arrays (X264HIP_STATE_NO_LEVELS) needs the entropy coder in the loop (rd.write)"); return -1; }
    if (p->me_method < 0 || p->me_method > 3) { set_error("slice_sweep: me method %d not built (0 DIA, 1 HEX, 2 UMH, 3 ESA)", p->me_method); return -1; }
    if (p->me_method == 3 && p->subme < 1) { set_error("slice_sweep: ESA at subme 0 is undefined in the reference (it never fills the integral plane there, encoder.c:1009 / mc.c:431)"); return -1; }
    if (p->transform8x8 && (!p->quant8_mf || !p->quant8_bias || !p->dequant8_mf)) { set_error("slice_sweep: 8x8 quantiser tables missing"); return -1; }
    if (is_p && !p->cost_mv) { set_error("slice_sweep: cost_mv missing"); return -1; }
    if (c->d.mb_w > 0xffff) { set_error("slice_sweep: frame too wide"); return -1; }
    memset(&a, 0, sizeof(a)); memset(&t, 0, sizeof(t));
    a.mb_w = c->d.mb_w; a.mb_h = c->d.mb_h; a.sy = c->d.stride_y; a.sc = c->d.stride_c; a.batch = c->batch; a.batch_pad = (c->batch + 7) & ~7;
    a.bs_y = c->bs_y; a.bs_c = c->bs_c;
    a.slice_type = p->slice_type; a.qp = p->qp;
    int qc = p->qp + p->chroma_qp_offset; qc = qc < 0 ? 0 : qc > 51 ? 51 : qc;
    a.qpc = k_chroma_qp[qc];
    a.lambda = k_lambda_tab[p->qp];
    a.chroma_skip_thresh = (k_lambda2_tab[a.qpc] + 32) >> 6;
    a.n_refs = is_p ? n_refs : 0;
    for (int i = 0; i < SW_MAX_REFS; i++) {
        int x = (n_refs <= 0 ? 1 : n_refs) - 1; x = x > 2 ? 2 : x;
        // REF_COST: lambda * bs_size_te(x, i), R/encoder/analyse.c:195-197
        int bits = x == 1 ? 1 : x > 1 ? (i == 0 ? 1 : i < 3 ? 3 : i < 7 ? 5 : 7) : 0;
        t.ref_bits[i] = bits;
        t.poc_delta[i] = i < n_refs ? p->poc - p->ref_poc[i] : 0;
        t.l0_inv_ref_poc[i] = l0 ? l0->inv_ref_poc[i] : 0;
    }
    a.l0_n_ref0 = l0 ? l0->n_ref0 : 0;
    a.me_method = p->me_method; a.me_range = p->me_range; a.subme = p->subme;
    a.chroma_me = p->chroma_me && is_p && !is_b && p->subme >= 5;   // h->mb.b_chroma_me, analyse.c:234-235
    a.fast_pskip = p->fast_pskip; a.cabac = p->cabac; a.mv_range = p->mv_range > 0 ? p->mv_range : 512;
    a.dct_decimate = p->dct_decimate || is_b;     // B slices always decimate (R/encoder/macroblock.c:193,275,479)
    a.q4mf = p->quant4_mf; a.q4bias = p->quant4_bias; a.dq4 = p->dequant4_mf;
    a.q8mf = p->quant8_mf; a.q8bias = p->quant8_bias; a.dq8 = p->dequant8_mf;
    a.transform8x8 = p->transform8x8 != 0;
    a.flags_inter = is_p ? (p->analyse_inter & (is_b ? 0x100 : 0x30)) : 0; a.mixed_refs = p->mixed_refs != 0;     // B: X264_ANALYSE_BSUB16x16
    // x264_mb_analyse_intra takes its flags from param.analyse.intra in I slices and from .inter in P slices (analyse.c:614);
    // i8x8 needs the 8x8 transform (x264_validate_parameters, R/encoder/encoder.c:487-491)
    a.flags_intra = (is_p ? p->analyse_inter : p->analyse_intra) & (a.transform8x8 ? 3 : 1);
    a.cost_mv = p->cost_mv; a.cost_center = p->cost_mv_range;
    a.lowres0 = is_p ? p->lowres_mv : nullptr; a.lowres1 = is_b ? pb->lowres_mv1 : nullptr;
    a.fy = fenc->plane[0]; a.fu = fenc->plane[1]; a.fv = fenc->plane[2];
    a.dy = recon->plane[0]; a.du = recon->plane[1]; a.dv = recon->plane[2];
    if (l0) { a.l0_type = (const signed char *)l0->mb_type; a.l0_ref = (const signed char *)l0->ref; a.l0_mv = l0->mv; }
    a.mb_type = (signed char *)out->mb_type; a.partition = (signed char *)out->partition; a.sub_partition = (signed char *)out->sub_partition;
    a.ref = (signed char *)out->ref;
    a.i4mode = (signed char *)out->i4mode; a.i16mode = (signed char *)out->i16mode; a.chroma_mode = (signed char *)out->chroma_mode;
    a.qp_out = (signed char *)out->qp; a.t8 = (signed char *)out->t8; a.mv = out->mv; a.mvr = out->mvr; a.cbp = out->cbp; a.nnz = out->nnz;
    a.luma = out->luma; a.luma_dc = out->luma_dc; a.chroma_dc = out->chroma_dc; a.chroma_ac = out->chroma_ac;
    a.cost_intra = out->cost_intra; a.cost_inter = out->cost_inter; a.cost_alt = out->cost_intra_alt;
    a.progress = out->progress; a.abort_flag = out->progress + (size_t)c->d.mb_h * c->batch;
    a.abort_total = (int *)((char *)c->ssd_dev + 24 * (size_t)c->batch + 32);
    { const char *e = getenv("X264HIP_SPIN_LIMIT"); a.spin_limit = e && atoi(e) > 0 ? atoi(e) : SW_SPIN_LIMIT; }
    a.prof = (long long *)p->profile;
    a.nr = p->noise_reduction != 0;
    a.lossless = p->lossless != 0;
    if (a.lossless && (p->qp != 0 || a.nr || p->fast_pskip)) { set_error("slice_sweep: lossless needs qp 0, no fast_pskip, no noise reduction (x264_validate_parameters)"); return -1; }
    if (a.nr && (!p->nr || !p->nr->sum || !p->nr->count || !p->nr->offset)) { set_error("slice_sweep: noise_reduction without an x264hip_nr_state"); return -1; }
    a.nr_sum = a.nr ? p->nr->sum : nullptr; a.nr_count = a.nr ? p->nr->count : nullptr; a.nr_offset = a.nr ? p->nr->offset : nullptr;
    for (int i = 0; i < SW_MAX_REFS; i++) {
        const x264hip_picture *r = (is_p && n_refs > 0) ? refs[i < n_refs ? i : 0] : fenc;
        for (int k = 0; k < 4; k++) t.y[i][k] = r->filtered[k];
        t.u[i] = r->plane[1]; t.v[i] = r->plane[2];
    }
    if (is_b) {                                                      // x264_macroblock_bipred_init, R/common/macroblock.c:1374-1408
        for (int k = 0; k < 4; k++) t.y1[k] = pb->fref1->filtered[k];
        t.u1 = pb->fref1->plane[1]; t.v1 = pb->fref1->plane[2];
        for (int i = 0; i < SW_MAX_REFS; i++) {
            const int poc0 = p->ref_poc[i < n_refs ? i : 0];
            int td = pb->ref1_poc - poc0; td = td < -128 ? -128 : td > 127 ? 127 : td;
            int dsf = 256;
            if (td) {
                int tb = p->poc - poc0; tb = tb < -128 ? -128 : tb > 127 ? 127 : tb;
                const int tx = (16384 + (abs(td) >> 1)) / td;
                dsf = (tb * tx + 32) >> 6; dsf = dsf < -1024 ? -1024 : dsf > 1023 ? 1023 : dsf;
            }
            t.dsf[i] = dsf;                                            // h->mb.dist_scale_factor[i][0]
            dsf >>= 2;
            t.biw[i] = pb->weightb && dsf >= -64 && dsf <= 128 ? 64 - dsf : 32;
        }
        // h->mb.map_col_to_list0 (x264_macroblock_slice_init, R/common/macroblock.c:790-804): the co-located picture's list 0 by POC in ours
        for (int i = 0; i < SW_MAX_REFS; i++) {
            t.map_col[i] = -2;
            if (i < pb->l1_state->n_ref0)
                for (int j = 0; j < n_refs; j++) if (p->ref_poc[j] == pb->l1_state->ref_poc[i]) { t.map_col[i] = j; break; }
        }
    } else {
        for (int k = 0; k < 4; k++) t.y1[k] = t.y[0][k];
        t.u1 = t.u[0]; t.v1 = t.v[0];
        for (int i = 0; i < SW_MAX_REFS; i++) { t.biw[i] = 32; t.dsf[i] = 256; t.map_col[i] = -2; }
    }
    memset(&r, 0, sizeof(r));
    kind = SW_KIND_PLAIN;
    if (prd) {
        r.on = 1; r.mbrd = mbrd; r.trellis = p->cabac ? prd->trellis : 0; r.psy_rd = mbrd ? prd->psy_rd : 0;
        r.write = prd->write; r.cabac_init_idc = prd->cabac_init_idc; r.i_frame = prd->i_frame; r.i_frame_stride = prd->i_frame_stride;
        r.aq = prd->aq_offset != nullptr; r.qp_min = prd->qp_min; r.qp_max = prd->qp_max; r.chroma_qp_offset = p->chroma_qp_offset;
        r.f_qpm = prd->f_qpm; r.aq_offset = prd->aq_offset; r.cost_mv_all = prd->cost_mv_all;
        r.unq4 = prd->unquant4_mf; r.unq8 = prd->unquant8_mf;
        r.payload = prd->payload; r.payload_cap = prd->payload_cap; r.payload_len = prd->payload_len; r.mb_bits = prd->mb_bits;
        r.mvd = out->mvd;
        r.stale = prd->stale;
        if (is_b) {
            r.direct_temporal = !pb->direct_spatial;
            r.direct_score = pb->direct_score;
            r.mv1 = out->mv1; r.ref1 = (signed char *)out->ref1; r.mvr1 = out->mvr1; r.mvd1 = out->mvd1; r.skipbp = out->skipbp;
            r.col_type = (const signed char *)pb->l1_state->mb_type; r.col_ref = (const signed char *)pb->l1_state->ref; r.col_mv = pb->l1_state->mv;
            // the extended B kernel (temporal direct prediction, the lookahead's candidates) only where it is needed: the plain one is 6-8 % faster
            kind = r.direct_temporal || r.direct_score || a.lowres0 || a.lowres1 ? SW_KIND_BT : SW_KIND_B;
        } else
            kind = mbrd >= 2 ? SW_KIND_RF : SW_KIND_RD;     // subme 8-9: the I / P kernel with the RD refinement (slice_refine.h)
    }
    return 0;
}

// the frame-level scalars later frames read from this one (x264_macroblock_slice_init, R/common/macroblock.c:771-808)
static void sweep_note_frame(const x264hip_slice_params *p, int n_refs, x264hip_mb_state *out)
{
    const bool is_p = p->slice_type == 0 || p->slice_type == 1;
    out->poc = p->poc; out->n_ref0 = is_p ? n_refs : 0;
    for (int i = 0; i < 8; i++) out->ref_poc[i] = i < n_refs ? p->ref_poc[i] : 0;     /* (a B frame's state is never read by later frames) */
    for (int i = 0; i < SW_MAX_REFS; i++) {
        int delta = i < n_refs && is_p ? p->poc - p->ref_poc[i] : 0;
        out->inv_ref_poc[i] = delta ? (256 + delta / 2) / delta : 0;
    }
}

extern "C" int x264hip_slice_sweep_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *const *refs, int n_refs,
                                         x264hip_picture *recon, const x264hip_slice_params *p, const x264hip_mb_state *l0,
                                         x264hip_mb_state *out)
{
    SwArgs a; SwRefs t; SwRd r;
    int kind;
    if (sweep_build(c, fenc, refs, n_refs, recon, p, l0, out, a, t, r, kind)) return -1;
    const bool is_p = p->slice_type == 0;
    HIPCHK(hipMemsetAsync(out->progress, 0, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1), c->stream));
    static int wpe = 0;
    if (!wpe) {                                                      // developer knob: X264HIP_SWEEP_WPE = 1..3
        const char *e = getenv("X264HIP_SWEEP_WPE");
        wpe = e ? atoi(e) : 3;                                       // 3 waves/SIMD (168 VGPRs, 12 waves per CU with 13 KB of LDS each): measured best
        if (wpe < 1 || wpe > 3) wpe = 3;
    }
    switch (kind) {
    case SW_KIND_BT: x264hip_launch_slice_bt(a, t, r, c->stream); break;
    case SW_KIND_B: x264hip_launch_slice_b(a, t, r, c->stream); break;
    case SW_KIND_RF: x264hip_launch_slice_rf(a, t, r, c->stream); break;
    case SW_KIND_RD: x264hip_launch_slice_rd(a, t, r, c->stream); break;
    default: {
        const dim3 grid((unsigned)(a.batch_pad * a.mb_h)), block(64);
        switch (a.lossless ? 0 : wpe) {
        case 0: hipLaunchKernelGGL((k_slice_sweep<2, true>), grid, block, 0, c->stream, a, t, r, nullptr); break;
        case 1: hipLaunchKernelGGL(k_slice_sweep<1>, grid, block, 0, c->stream, a, t, r, nullptr); break;
        case 3: hipLaunchKernelGGL(k_slice_sweep<3>, grid, block, 0, c->stream, a, t, r, nullptr); break;
        default: hipLaunchKernelGGL(k_slice_sweep<2>, grid, block, 0, c->stream, a, t, r, nullptr); break;
        }
    }
    }
    if (kind == SW_KIND_PLAIN && is_p && a.flags_intra)
        hipLaunchKernelGGL(k_resolve_fast_intra, dim3(c->batch), dim3(64), 0, c->stream, (const signed char *)out->mb_type, out->cost_intra,
                           (const int *)out->cost_intra_alt, c->d.mb_w * c->d.mb_h);
    HIPCHK(hipGetLastError());
    sweep_note_frame(p, n_refs, out);
    return 0;
}

// The chain-table launch: every entry is a sweep of ONE chain (batch element) with its own pictures, states and slice parameters.
static int sweep_chains(x264hip_frame_ctx *c, x264hip_chain_sweep *e, int n, void *staging_host, void *table_dev, void *ev_ip, void *ev_b, bool join);
static std::mutex g_aux_mu;
static std::unordered_map<x264hip_frame_ctx *, ChainAux> g_aux_of;
// the stream the B kernel of this context's chain-table launches runs on (default: one the library creates): e.g. one restricted to
// a part of the device (x264hip_stream_create_cu_range) while the context's own stream has the rest
extern "C" int x264hip_frame_ctx_set_b_stream(x264hip_frame_ctx *c, void *hip_stream)
{
    std::lock_guard<std::mutex> g(g_aux_mu);
    ChainAux &ax = g_aux_of[c];
    if (!ax.ready) {
        HIPCHK(hipEventCreateWithFlags(&ax.ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ax.done, hipEventDisableTiming));
    }
    ax.stream = (hipStream_t)hip_stream;
    return 0;
}
extern "C" int x264hip_slice_sweep_chains(x264hip_frame_ctx *c, x264hip_chain_sweep *e, int n, void *staging_host, void *table_dev)
{
    return sweep_chains(c, e, n, staging_host, table_dev, nullptr, nullptr, true);
}
// The same with the two kinds' completion visible to the caller: ev_ip is recorded behind the I / P kernels on the context's stream, ev_b
// behind the B kernel on its own stream, and the context's stream does NOT wait for the B kernel (a disposable B frame has no end-of-frame
// work: what follows on the context's stream -- the filters of the kept frames -- concerns other chains).  For a scheduler that hands
// every chain its next frame as soon as its own kernel is done.
extern "C" int x264hip_slice_sweep_chains_events(x264hip_frame_ctx *c, x264hip_chain_sweep *e, int n, void *staging_host, void *table_dev, void *ev_ip, void *ev_b)
{
    if (!ev_ip || !ev_b) { set_error("slice_sweep_chains_events: events missing"); return -1; }
    return sweep_chains(c, e, n, staging_host, table_dev, ev_ip, ev_b, false);
}
static int sweep_chains(x264hip_frame_ctx *c, x264hip_chain_sweep *e, int n, void *staging_host, void *table_dev, void *ev_ip, void *ev_b, bool join)
{
    if (n <= 0) return 0;
    if (!staging_host || !table_dev) { set_error("slice_sweep_chains: staging / table buffers missing"); return -1; }
    SwDesc *st = (SwDesc *)staging_host;
    // entries sorted by kernel: [RD | RF | BT]; a first pass builds, a second places
    int cnt[5] = {0, 0, 0, 0, 0};
    static thread_local std::vector<SwDesc> tmp;
    static thread_local std::vector<int> kinds;
    tmp.resize((size_t)n); kinds.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        x264hip_chain_sweep &s = e[i];
        if (s.chain < 0 || s.chain >= c->batch) { set_error("slice_sweep_chains: entry %d: chain %d", i, s.chain); return -1; }
        if (!s.params || !s.params->rd || !s.params->rd->write) { set_error("slice_sweep_chains: entry %d: the chain table belongs to the raster variant with the entropy coder in the loop (params.rd, write = 1)", i); return -1; }
        int kind;
        if (sweep_build(c, s.fenc, s.refs, s.n_refs, s.recon, s.params, s.l0, s.out, tmp[i].a, tmp[i].t, tmp[i].r, kind)) return -1;
        if (kind == SW_KIND_B) kind = SW_KIND_BT;                       // one B kernel in the table launches
        tmp[i].a.chain = s.chain;
        kinds[i] = kind; cnt[kind]++;
        sweep_note_frame(s.params, s.n_refs, s.out);
    }
    int base[5], at[5];
    base[SW_KIND_RD] = 0; base[SW_KIND_RF] = cnt[SW_KIND_RD]; base[SW_KIND_BT] = base[SW_KIND_RF] + cnt[SW_KIND_RF];
    for (int k = 0; k < 5; k++) at[k] = base[k];
    for (int i = 0; i < n; i++) st[at[kinds[i]]++] = tmp[i];
    HIPCHK(hipMemcpyAsync(table_dev, st, sizeof(SwDesc) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    const SwDesc *tab = (const SwDesc *)table_dev;
    // The I / P chains and the B chains of a step are different chains: their kernels run side by side, the B kernel on a stream of its
    // own between two events (behind the table's upload, ahead of whatever follows on the context's stream).
    const bool two = cnt[SW_KIND_BT] && (cnt[SW_KIND_RD] || cnt[SW_KIND_RF] || !join);
    ChainAux *ax = nullptr;
    if (two) {
        std::lock_guard<std::mutex> g(g_aux_mu);
        ax = &g_aux_of[c];
        if (!ax->stream) HIPCHK(hipStreamCreateWithFlags(&ax->stream, hipStreamNonBlocking));
        if (!ax->ready) {
            HIPCHK(hipEventCreateWithFlags(&ax->ready, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ax->done, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ax->ready, c->stream));
        HIPCHK(hipStreamWaitEvent(ax->stream, ax->ready, 0));
    }
    // the I / P kernel first: its wavefronts -- the step's long ones -- are dealt evenly over the SIMDs before the B kernel's fill the rest
    if (cnt[SW_KIND_RD]) x264hip_launch_slice_rd_ch(tab + base[SW_KIND_RD], cnt[SW_KIND_RD], c->stream);
    if (cnt[SW_KIND_RF]) x264hip_launch_slice_rf_ch(tab + base[SW_KIND_RF], cnt[SW_KIND_RF], c->stream);
    if (cnt[SW_KIND_BT]) x264hip_launch_slice_bt_ch(tab + base[SW_KIND_BT], cnt[SW_KIND_BT], two ? ax->stream : c->stream);
    if (two && join) {
        HIPCHK(hipEventRecord(ax->done, ax->stream));
        HIPCHK(hipStreamWaitEvent(c->stream, ax->done, 0));
    }
    if (ev_ip) HIPCHK(hipEventRecord((hipEvent_t)ev_ip, c->stream));
    if (ev_b) HIPCHK(hipEventRecord((hipEvent_t)ev_b, two ? ax->stream : c->stream));
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" size_t x264hip_chain_sweep_bytes(void) { return sizeof(SwDesc); }
// the abort flag of a state (what x264hip_slice_sweep_frame clears before each launch; a chain-table caller clears it once per state it writes)
extern "C" int x264hip_mb_state_clear_progress(x264hip_frame_ctx *c, x264hip_mb_state *st)
{
    HIPCHK(hipMemsetAsync(st->progress, 0, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1), c->stream));
    return 0;
}

extern "C" int x264hip_slice_sweep_status(x264hip_frame_ctx *c, const x264hip_mb_state *st)
{
    int flag = 0, total = 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(&flag, st->progress + (size_t)c->d.mb_h * c->batch, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&total, (char *)c->ssd_dev + 24 * (size_t)c->batch + 32, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) { set_error("slice_sweep: a wavefront gave up -- waiting for its neighbours, or out of payload space (aborted frame)"); return -1; }
    // sticky: an aborted frame may have been used as a reference since, and its own flag is cleared when its state is reused
    if (total) { set_error("slice_sweep: %d wavefront(s) of an EARLIER frame of this context gave up -- waiting, or out of payload space (aborted frame): everything coded since is invalid", total); return -1; }
    return 0;
}
