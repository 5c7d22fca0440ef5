/* oracle/look_oracle.c -- TEST INFRASTRUCTURE (part of liboracle.so, included by frame_oracle.c).
 *
 * CPU restatement of the lookahead's per-frame cost, x264_slicetype_frame_cost's uncached branch with x264_slicetype_mb_cost inside it
 * (R/encoder/slicetype.c:43-253, 256-345), as ONE task: frame b scored against p0 (list 0) and p1 (list 1) on the half-resolution planes,
 * macroblocks in reverse raster order so that the right / lower neighbours' vectors are the search's candidates.  The memoisation around it
 * (i_cost_est, the "searched" markers in lowres_mvs[..][0][0], b_intra_calculated) belongs to the caller: the library's host C
 * (x264_vs2008_amd/csrc/lookahead_host.hip) keeps it, tests drive that code with this function as the cost provider and compare frame types,
 * QPs and vectors with the reference's own queue (oracle/ref_slice.c refslice_encode_stream).
 *
 * Only frames wider and taller than two macroblocks (the reference scores the interior only then, slicetype.c:322-336), no VBV row sums. */

typedef struct {
    int mb_w, mb_h, stride;            /* lowres stride (all twelve planes share it) */
    int p0, p1, b;                     /* positions in the lookahead's frame list; p0 == p1 == b: intra only */
    int do_search[2];                  /* list l's vectors are not there yet: search and store them */
    int me_method, me_range;           /* X264_MIN(X264_ME_HEX, param me), param me_range */
    int weighted_bipred;               /* param.analyse.b_weighted_bipred */
    int bframe_bias;                   /* param.i_bframe_bias (B score * 100 / (120 + bias)) */
} x264o_look_task;

/* planes: [3][4] = frame b, p0, p1: lowres luma + H, V, HV at the picture origin.  mv / cost: this frame's lowres_mvs[l][dist - 1] and
 * lowres_mv_costs[l][dist - 1], [n][2] / [n]; mvr: frames[p1]->lowres_mvs[0][p1 - p0 - 1] (bidirectional only).  intra_cost [n]:
 * x264o_frame_lookahead_intra's output for frame b.  out: {i_score (before the intra penalty of the caller), i_intra_mbs, i_cost_est[0][0]} */
void x264o_look_frame_cost(const x264o_look_task *t, u8 *const *planes, i16 *mv0, int32_t *cost0, i16 *mv1, int32_t *cost1, const i16 *mvr,
                           const int32_t *intra_cost, const i16 *cost_mv /* centred table of lambda 1 */, int32_t *out)
{
    init();
    const int mb_w = t->mb_w, mb_h = t->mb_h, st = t->stride, b_bidir = t->b < t->p1, intra_only = !t->p0 && !t->p1 && !t->b;
    int dist_scale_factor = 128, score = 0, intra_mbs = 0, cost00 = 0;
    if (t->p1 != t->p0) dist_scale_factor = (((t->b - t->p0) << 8) + ((t->p1 - t->p0) >> 1)) / (t->p1 - t->p0);
    const int bipred_weight = t->weighted_bipred ? 64 - (dist_scale_factor >> 2) : 32;
    i16 *mvs[2] = {mv0, mv1};
    int32_t *costs[2] = {cost0, cost1};
    for (int my = mb_h - 2; my > 0; my--)
        for (int mx = mb_w - 2; mx > 0; mx--) {
            const int xy = mx + my * mb_w, off = 8 * (mx + my * st);
            int bcost = ME_COST_MAX;
            if (!intra_only) {
                me_ctx c;
                c.fenc = planes[0] + off; c.fenc_u = c.fenc_v = 0; c.sy = st; c.sc = 0;
                c.pix = X264HIP_PIXEL_8x8; c.bw = c.bh = 8;
                c.fmin[0] = -8 * mx - 4; c.fmax[0] = 8 * (mb_w - mx - 1) + 4;
                c.fmin[1] = -8 * my - 4; c.fmax[1] = 8 * (mb_h - my - 1) + 4;
                for (int k = 0; k < 2; k++) { c.smin[k] = 4 * (c.fmin[k] - 8); c.smax[k] = 4 * (c.fmax[k] + 8); }
                u8 *ref[2][6] = {{0}};
                for (int l = 0; l < 2; l++) for (int k = 0; k < 4; k++) ref[l][k] = planes[4 * (1 + l) + k] + off;
                int mcost[2] = {0, 0}, mvl[2][2] = {{0, 0}, {0, 0}};
#define LOOK_TRY_BIDIR(a0, a1, penalty) do { \
                    u8 pix1[8 * 16], pix2[8 * 16], avg[8 * 16]; int s1 = 16, s2 = 16; \
                    u8 *src1 = mcf.get_ref(pix1, &s1, ref[0], st, (a0)[0], (a0)[1], 8, 8); \
                    u8 *src2 = mcf.get_ref(pix2, &s2, ref[1], st, (a1)[0], (a1)[1], 8, 8); \
                    mcf.avg[X264HIP_PIXEL_8x8](avg, 16, src1, s1, src2, s2, bipred_weight); \
                    int cst = (penalty) + pixf.satd[X264HIP_PIXEL_8x8](c.fenc, st, avg, 16); \
                    if (bcost > cst) bcost = cst; } while (0)
                if (b_bidir) {
                    const i16 *r = mvr + 2 * xy;
                    int dmv[2][2], zero[2] = {0, 0};
                    dmv[0][0] = (r[0] * dist_scale_factor + 128) >> 8; dmv[0][1] = (r[1] * dist_scale_factor + 128) >> 8;
                    dmv[1][0] = dmv[0][0] - r[0]; dmv[1][1] = dmv[0][1] - r[1];
                    for (int k = 0; k < 2; k++) { dmv[k][0] = clip3i(dmv[k][0], c.smin[0], c.smax[0]); dmv[k][1] = clip3i(dmv[k][1], c.smin[1], c.smax[1]); }
                    LOOK_TRY_BIDIR(dmv[0], dmv[1], 0);
                    if (dmv[0][0] | dmv[0][1] | dmv[1][0] | dmv[1][1]) LOOK_TRY_BIDIR(zero, zero, 0);
                }
                for (int l = 0; l < 1 + b_bidir; l++) {
                    if (t->do_search[l]) {
                        i16 mvc[4][2] = {{0}}, mvp[2];
                        int n_mvc = 0;
                        const i16 *fm = mvs[l] + 2 * xy;
#define LOOK_MVC(d) do { mvc[n_mvc][0] = fm[2 * (d)]; mvc[n_mvc][1] = fm[2 * (d) + 1]; n_mvc++; } while (0)
                        if (mx < mb_w - 1) LOOK_MVC(1);
                        if (my < mb_h - 1) {
                            LOOK_MVC(mb_w);
                            if (mx > 0) LOOK_MVC(mb_w - 1);
                            if (mx < mb_w - 1) LOOK_MVC(mb_w + 1);
                        }
                        for (int k = 0; k < 2; k++) {                       /* x264_median_mv of the first three (zero where absent) */
                            int a = mvc[0][k], bb = mvc[1][k], cc = mvc[2][k];
                            int mn = a < bb ? a : bb, mxv = a < bb ? bb : a;
                            mvp[k] = (i16)(cc < mn ? mn : cc > mxv ? mxv : cc);
                        }
                        for (int k = 0; k < 6; k++) c.fref[k] = ref[l][k];
                        c.cmx = cost_mv - mvp[0]; c.cmy = cost_mv - mvp[1];
                        int mxo, myo, cmv;
                        int cost = me_search16(&c, mvp, (const i16 (*)[2])mvc, n_mvc, t->me_method, t->me_range, 4, 0, 0, &mxo, &myo, &cmv);
                        cost -= 2;                                          /* remove mvcost from skip mbs */
                        if (mxo | myo) cost += 5;
                        mvs[l][2 * xy] = (i16)mxo; mvs[l][2 * xy + 1] = (i16)myo; costs[l][xy] = cost;
                    }
                    mvl[l][0] = mvs[l][2 * xy]; mvl[l][1] = mvs[l][2 * xy + 1]; mcost[l] = costs[l][xy];
                    if (mcost[l] < bcost) bcost = mcost[l];
                }
                if (b_bidir && (mvl[0][0] | mvl[0][1] | mvl[1][0] | mvl[1][1])) LOOK_TRY_BIDIR(mvl[0], mvl[1], 5);
            }
            if (!b_bidir) {                                                /* no intra blocks in B frames */
                int icost = intra_cost[xy], b_intra = icost < bcost;
                if (b_intra) bcost = icost;
                intra_mbs += b_intra; cost00 += icost;
            }
            score += bcost;
        }
    if (t->b != t->p1) score = score * 100 / (120 + t->bframe_bias);
    out[0] = score; out[1] = intra_mbs; out[2] = cost00;
}
