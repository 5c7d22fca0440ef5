/* frame_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU twin of the frame level of libx264hip.so: every x264hip_*_frame entry
 * has a counterpart here that is nothing but the reference's own per-frame
 * driver logic restated around the oracle's table entries (x264_oracle.c,
 * which is pinned to the reference's C build by golden vectors).  Host
 * buffers use the same padded layout as the device planes (pointers address
 * pixel (0,0); PADH = PADV = 32 luma, 16 chroma).
 *
 * R/ = x264-snapshot-20090216-2245/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/x264hip_tables.h"

typedef uint8_t u8;
typedef int16_t i16;
typedef uint16_t u16;

void x264o_pixel_init(x264hip_pixel_function_t *);
void x264o_dct_init(x264hip_dct_function_t *);
void x264o_zigzag_init(x264hip_zigzag_function_t *, int);
void x264o_quant_init(x264hip_quant_function_t *);
void x264o_mc_init(x264hip_mc_functions_t *);
void x264o_deblock_init(x264hip_deblock_function_t *);
int64_t x264o_pixel_ssd_wxh(u8 *, int, u8 *, int, int, int);

static x264hip_pixel_function_t pixf;
static x264hip_dct_function_t dctf;
static x264hip_zigzag_function_t zigf[2];
static x264hip_quant_function_t quantf;
static x264hip_mc_functions_t mcf;
static x264hip_deblock_function_t dbf;
static int ready;
#ifdef X264O_USE_REF
/* Second build of this file (oracle/_ref/libframe_ref.so, `make ref`): the
 * same per-frame loops, but every table entry is the REFERENCE's own C
 * function from _ref/libx264ref.so.  bench.py times this one as
 * cpu_baseline.kind = "reference"; tests use it to check the twin's loops. */
void x264_pixel_init(int, x264hip_pixel_function_t *);
void x264_dct_init(int, x264hip_dct_function_t *);
void x264_zigzag_init(int, x264hip_zigzag_function_t *, int);
void x264_quant_init(void *, int, x264hip_quant_function_t *);
void x264_mc_init(int, x264hip_mc_functions_t *);
void x264_deblock_init(int, x264hip_deblock_function_t *);
int64_t x264_pixel_ssd_wxh(x264hip_pixel_function_t *, u8 *, int, u8 *, int, int, int);
#define x264o_pixel_ssd_wxh(a, sa, b, sb, w, h) x264_pixel_ssd_wxh(&pixf, a, sa, b, sb, w, h)
#define x264o_plane_expand_border x264r_plane_expand_border
#define x264o_plane_pad_mod16 x264r_plane_pad_mod16
#define x264o_frame_hpel x264r_frame_hpel
#define x264o_frame_lowres x264r_frame_lowres
#define x264o_frame_aq_var x264r_frame_aq_var
#define x264o_frame_ssd x264r_frame_ssd
#define x264o_frame_me_fullpel x264r_frame_me_fullpel
#define x264o_frame_me_subpel x264r_frame_me_subpel
#define x264o_frame_inter_residual x264r_frame_inter_residual
#define x264o_frame_deblock x264r_frame_deblock
#define x264o_frame_lookahead_intra x264r_frame_lookahead_intra
#define x264o_frame_inter_residual_mp x264r_frame_inter_residual_mp
#define x264o_frame_probe_skip x264r_frame_probe_skip
#define x264o_frame_me_search16 x264r_frame_me_search16
#endif
static void init(void)
{
    if (ready) return;
#ifdef X264O_USE_REF
    x264_pixel_init(0, &pixf); x264_dct_init(0, &dctf);
    x264_zigzag_init(0, &zigf[0], 0); x264_zigzag_init(0, &zigf[1], 1);
    x264_quant_init(0, 0, &quantf); x264_mc_init(0, &mcf); x264_deblock_init(0, &dbf);
#else
    x264o_pixel_init(&pixf); x264o_dct_init(&dctf);
    x264o_zigzag_init(&zigf[0], 0); x264o_zigzag_init(&zigf[1], 1);
    x264o_quant_init(&quantf); x264o_mc_init(&mcf); x264o_deblock_init(&dbf);
#endif
    ready = 1;
}

/* plane_expand_border, R/common/frame.c:218-240 */
void x264o_plane_expand_border(u8 *pix, int stride, int width, int height, int padh, int padv)
{
    for (int y = 0; y < height; y++) {
        memset(pix + y * stride - padh, pix[y * stride], padh);
        memset(pix + y * stride + width, pix[y * stride + width - 1], padh);
    }
    for (int y = 0; y < padv; y++)
        memcpy(pix - padh + (-y - 1) * stride, pix - padh, width + 2 * padh);
    for (int y = 0; y < padv; y++)
        memcpy(pix - padh + (height + y) * stride, pix - padh + (height - 1) * stride, width + 2 * padh);
}

/* x264_frame_expand_border_mod16, R/common/frame.c:303-334 */
void x264o_plane_pad_mod16(u8 *p, int stride, int w, int h, int w16, int h16)
{
    if (w16 > w)
        for (int y = 0; y < h; y++) memset(p + y * stride + w, p[y * stride + w - 1], w16 - w);
    for (int y = h; y < h16; y++) memcpy(p + y * stride, p + (h - 1) * stride, w16);
}

/* x264_frame_filter band by band + x264_frame_expand_border_filtered,
 * R/common/mc.c:404-426 and R/common/frame.c:272-296 (frame mode). */
void x264o_frame_hpel(u8 *plane, u8 *fh, u8 *fv, u8 *fc, int stride, int width16, int lines16, int mb_h)
{
    init();
    i16 *buf = malloc((width16 + 64) * sizeof(i16));
    /* x264_fdec_filter_row calls x264_frame_filter(min_y, b_end) for min_y =
     * 0 .. mb_h-1 (R/encoder/encoder.c:983-1024); each call filters the band
     * of rows [16*min_y - 8, (b_end ? lines : 16*min_y) + 8), 8 columns
     * beyond each side (R/common/mc.c:408-425). */
    for (int min_y = 0; min_y < mb_h; min_y++) {
        int b_end = min_y == mb_h - 1;
        int start = min_y * 16 - 8;
        int height = (b_end ? lines16 : min_y * 16) + 8;
        int offs = start * stride - 8;
        mcf.hpel_filter(fh + offs, fv + offs, fc + offs, plane + offs, stride, width16 + 16, height - start, buf);
    }
    free(buf);
    for (int i = 0; i < 3; i++) {
        u8 *p = (i == 0 ? fh : i == 1 ? fv : fc) - 8 * stride - 4;
        x264o_plane_expand_border(p, stride, width16 + 8, lines16 + 16, 32 - 4, 32 - 8);
    }
}

/* x264_frame_init_lowres, R/common/mc.c:306-331 */
void x264o_frame_lowres(u8 *plane, int stride, int width16, int lines16, u8 *l0, u8 *lh, u8 *lv, u8 *lc,
                        int stride_lowres, int width_lowres, int lines_lowres)
{
    init();
    for (int y = 0; y < lines16; y++) plane[width16 + y * stride] = plane[width16 - 1 + y * stride];
    memcpy(plane + stride * lines16, plane + stride * (lines16 - 1), width16);
    mcf.frame_init_lowres_core(plane, l0, lh, lv, lc, stride, stride_lowres, width_lowres, lines_lowres);
    u8 *pl[4] = {l0, lh, lv, lc};
    for (int i = 0; i < 4; i++)
        x264o_plane_expand_border(pl[i], stride_lowres, stride_lowres - 64, lines_lowres, 32, 32);
}

/* ac_energy_mb for every macroblock, R/encoder/ratecontrol.c:171-195 */
void x264o_frame_aq_var(u8 *py, u8 *pu, u8 *pv, int sy, int sc, int mb_w, int mb_h, int32_t *out)
{
    init();
    for (int my = 0; my < mb_h; my++)
        for (int mx = 0; mx < mb_w; mx++) {
            unsigned v = pixf.var[X264HIP_PIXEL_16x16](py + 16 * (mx + my * sy), sy);
            v += pixf.var[X264HIP_PIXEL_8x8](pu + 8 * (mx + my * sc), sc);
            v += pixf.var[X264HIP_PIXEL_8x8](pv + 8 * (mx + my * sc), sc);
            out[mx + my * mb_w] = v ? v : 1;
        }
}

int64_t x264o_frame_ssd(u8 *a, int sa, u8 *b, int sb, int w, int h) { return x264o_pixel_ssd_wxh(a, sa, b, sb, w, h); }

/* ---------------------------------------------------------------- lookahead
 * intra half of x264_slicetype_mb_cost, R/encoder/slicetype.c:186-245 */
#ifdef X264O_USE_REF
void x264_predict_8x8c_init(int, x264hip_predict_t pf[7]);
void x264_predict_8x8_init(int, x264hip_predict8x8_t pf[12], x264hip_predict_8x8_filter_t *);
#else
void x264o_predict_8x8c_init(x264hip_predict_t pf[7]);
void x264o_predict_8x8_init(x264hip_predict8x8_t pf[12], x264hip_predict_8x8_filter_t *);
#endif
void x264o_frame_lookahead_intra(u8 *low, int stride, int mb_w, int mb_h, int32_t *out)
{
    init();
    x264hip_predict_t p8c[7];
    x264hip_predict8x8_t p8[12];
    x264hip_predict_8x8_filter_t filt;
#ifdef X264O_USE_REF
    x264_predict_8x8c_init(0, p8c); x264_predict_8x8_init(0, p8, &filt);
#else
    x264o_predict_8x8c_init(p8c); x264o_predict_8x8_init(p8, &filt);
#endif
    for (int my = 0; my < mb_h; my++)
        for (int mx = 0; mx < mb_w; mx++) {
            u8 fenc[8 * 16], pix1[9 * 32], edge[33];
            u8 *src = low + 8 * (mx + my * stride) - 1;
            u8 *pix = &pix1[8 + 32 - 1];
            for (int y = 0; y < 8; y++) memcpy(fenc + y * 16, src + 1 + y * stride, 8);
            memcpy(pix - 32, src - stride, 17);
            for (int i = 0; i < 8; i++) pix[i * 32] = src[i * stride];
            pix++;
            int best = 0x7fffffff;
            for (int i = 0; i < 4; i++) {
                p8c[i](pix);
                int s = pixf.satd[X264HIP_PIXEL_8x8](pix, 32, fenc, 16);
                if (s < best) best = s;
            }
            filt(pix, edge, 0xf, 0xf);
            for (int i = 3; i < 9; i++) {
                p8[i](pix, edge);
                int s = pixf.satd[X264HIP_PIXEL_8x8](pix, 32, fenc, 16);
                if (s < best) best = s;
            }
            out[mx + my * mb_w] = best + 5;
        }
}

/* ------------------------------------------------------------------ motion
 * mv_min_fpel / mv_max_fpel and the spel limits, R/encoder/analyse.c:258-298 */
static int clip3i(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static void mv_limits(int mb_w, int mb_h, int mbx, int mby, int mv_range, int lim_spel[4], int lim_fpel[4])
{
    int fr = 4 * mv_range;
    int lo = 4 * (-512 + 8) > -fr ? 4 * (-512 + 8) : -fr;
    lim_spel[0] = clip3i(4 * (-16 * mbx - 24), -fr, fr - 1);
    lim_spel[1] = clip3i(4 * (16 * (mb_w - mbx - 1) + 24), -fr, fr - 1);
    lim_spel[2] = clip3i(4 * (-16 * mby - 24), lo, fr);
    lim_spel[3] = clip3i(4 * (16 * (mb_h - mby - 1) + 24), -fr, fr - 1);
    lim_fpel[0] = (lim_spel[0] >> 2) + 5; lim_fpel[1] = (lim_spel[1] >> 2) - 5;
    lim_fpel[2] = (lim_spel[2] >> 2) + 5; lim_fpel[3] = (lim_spel[3] >> 2) - 5;
}
static int mvcost(const u16 *cost_mv, int center, int d)
{
    int i = center + d;
    return cost_mv[i < 0 ? 0 : i > 2 * center ? 2 * center : i];
}

/* Every full-pel vector of the window, COST_MV of R/encoder/me.c:54-62, nine
 * partitions; first minimum in (my, mx) raster order wins. */
void x264o_frame_me_fullpel(u8 *fenc, u8 *ref, int mb_w, int mb_h, int stride, int range, int mv_range,
                            const u16 *cost_mv, int cost_center, const i16 *centers, const i16 *mvp,
                            i16 *out_mv, int32_t *out_cost, u16 *surface, u8 *surface_valid)
{
    init();
    static const int part_pix[9] = {X264HIP_PIXEL_16x16, X264HIP_PIXEL_16x8, X264HIP_PIXEL_16x8, X264HIP_PIXEL_8x16,
                                    X264HIP_PIXEL_8x16, X264HIP_PIXEL_8x8, X264HIP_PIXEL_8x8, X264HIP_PIXEL_8x8, X264HIP_PIXEL_8x8};
    static const int part_x[9] = {0, 0, 0, 0, 8, 0, 8, 0, 8}, part_y[9] = {0, 0, 8, 0, 0, 0, 0, 8, 8};
    int n = 2 * range + 1;
    for (int mb = 0; mb < mb_w * mb_h; mb++) {
        int mbx = mb % mb_w, mby = mb / mb_w, sp[4], fp[4];
        int cx = centers ? centers[2 * mb] : 0, cy = centers ? centers[2 * mb + 1] : 0;
        int pvx = mvp ? mvp[2 * mb] : 0, pvy = mvp ? mvp[2 * mb + 1] : 0;
        mv_limits(mb_w, mb_h, mbx, mby, mv_range, sp, fp);
        u8 *src = fenc + (16 * mby) * stride + 16 * mbx;
        int best[9], bx[9], by[9];
        for (int k = 0; k < 9; k++) { best[k] = 0x7fffffff; bx[k] = by[k] = 0; }
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) {
                int mx = cx - range + i, my = cy - range + j;
                int ok = mx >= fp[0] && mx <= fp[1] && my >= fp[2] && my <= fp[3];
                if (surface_valid) surface_valid[(size_t)mb * n * n + j * n + i] = (u8)ok;
                if (!ok) continue;
                u8 *r = ref + (16 * mby + my) * stride + 16 * mbx + mx;
                int mvc = mvcost(cost_mv, cost_center, 4 * mx - pvx) + mvcost(cost_mv, cost_center, 4 * my - pvy);
                for (int k = 0; k < 9; k++) {
                    int o = part_y[k] * stride + part_x[k];
                    int sad = pixf.sad[part_pix[k]](src + o, stride, r + o, stride);
                    if (k == 0 && surface) surface[(size_t)mb * n * n + j * n + i] = (u16)sad;
                    if (sad + mvc < best[k]) { best[k] = sad + mvc; bx[k] = mx; by[k] = my; }
                }
            }
        for (int k = 0; k < 9; k++) {
            out_mv[(mb * 9 + k) * 2] = bx[k]; out_mv[(mb * 9 + k) * 2 + 1] = by[k];
            out_cost[mb * 9 + k] = best[k];
        }
    }
}

/* 3x3 half-pel then 3x3 quarter-pel refinement of the 16x16 vector with
 * SATD (mbcmp) through get_ref (R/common/mc.c:181-202); raster order, first
 * minimum wins, vectors outside the spel limits skipped. */
void x264o_frame_me_subpel(u8 *fenc, u8 *p0, u8 *p1, u8 *p2, u8 *p3, int mb_w, int mb_h, int stride, int mv_range,
                           const u16 *cost_mv, int cost_center, const i16 *mvp, const i16 *mv_fullpel,
                           i16 *out_mv, int32_t *out_cost)
{
    init();
    u8 tmp[16 * 16];
    for (int mb = 0; mb < mb_w * mb_h; mb++) {
        int mbx = mb % mb_w, mby = mb / mb_w, sp[4], fp[4];
        int pvx = mvp ? mvp[2 * mb] : 0, pvy = mvp ? mvp[2 * mb + 1] : 0;
        mv_limits(mb_w, mb_h, mbx, mby, mv_range, sp, fp);
        int off = 16 * mby * stride + 16 * mbx;
        u8 *src[4] = {p0 + off, p1 + off, p2 + off, p3 + off};
        int bmx = 4 * mv_fullpel[mb * 18], bmy = 4 * mv_fullpel[mb * 18 + 1], bcost = 0;
        for (int step = 2; step >= 1; step--) {
            int best = 0x7fffffff, wx = bmx, wy = bmy;
            for (int s = 0; s < 9; s++) {
                int mx = bmx + step * (s % 3 - 1), my = bmy + step * (s / 3 - 1);
                if (s != 4 && !(mx >= sp[0] && mx <= sp[1] && my >= sp[2] && my <= sp[3])) continue;
                int rs = 16;
                u8 *r = mcf.get_ref(tmp, &rs, src, stride, mx, my, 16, 16);
                int c = pixf.satd[X264HIP_PIXEL_16x16](fenc + off, stride, r, rs)
                      + mvcost(cost_mv, cost_center, mx - pvx) + mvcost(cost_mv, cost_center, my - pvy);
                if (c < best) { best = c; wx = mx; wy = my; }
            }
            bmx = wx; bmy = wy; bcost = best;
        }
        out_mv[2 * mb] = bmx; out_mv[2 * mb + 1] = bmy; out_cost[mb] = bcost;
    }
}

/* ------------------------------------------------------- x264_me_search_ref
 * The reference's own search for a 16x16 block (R/encoder/me.c:156-778) run for every
 * macroblock and every reference with caller-supplied predictors, in the loop of
 * x264_mb_analyse_inter_p16x16 (R/encoder/analyse.c:1077-1127): predictor tests, DIA or
 * HEX walk + square refine, sub-pel refinement (refine_subpel, subpel_iterations[subme][2..3]),
 * chroma ME, the half-pel early-termination threshold carried from reference to reference, and
 * the choice of the best reference (cost + ref cost, first minimum).
 * method: 0 = DIA, 1 = HEX (X264_ME_DIA / X264_ME_HEX).  cost_mv: int16 table, cost of qpel
 * delta d at cost_mv[cost_center + d].  mvp: [mb][n_refs][2]; mvc: [mb][n_refs][8][2] with
 * n_mvc: [mb][n_refs]; ref_cost: [n_refs].
 * out_mv: [mb][n_refs][2] qpel, out_cost: [mb][n_refs] (ref cost included), best: [mb][4] =
 * {ref, mvx, mvy, cost}. */
typedef struct {
    u8 *fenc, *fenc_u, *fenc_v;       /* source block pointers (plane stride) */
    u8 *fref[6];                      /* 4 luma planes + U + V at the block position */
    int sy, sc;
    const i16 *cmx, *cmy;             /* cost tables already offset by the predictor */
    int fmin[2], fmax[2], smin[2], smax[2];
    int pix, bw, bh;                  /* block: X264HIP_PIXEL_16x16 / 16x8 / 8x16 / 8x8 and its size (pointers already offset to it) */
} me_ctx;
static const int me_subpel_iters[10][4] = {{0,0,0,0},{1,1,0,0},{0,1,1,0},{0,2,1,0},{0,2,1,1},{0,2,1,2},{0,0,2,2},{0,0,2,2},{0,0,4,10},{0,0,4,10}};
static const int me_hex2[8][2] = {{-1,-2},{-2,0},{-1,2},{1,2},{2,0},{1,-2},{-1,-2},{-2,0}};
static const int me_mod6m1[8] = {5,0,1,2,3,4,5,0};
#define ME_COST_MAX (1 << 28)

static int me_fpel_cost(const me_ctx *c, int mx, int my)
{   /* COST_MV's cost, me.c:54-62 */
    return pixf.sad[c->pix](c->fenc, c->sy, c->fref[0] + my * c->sy + mx, c->sy) + c->cmx[mx << 2] + c->cmy[my << 2];
}
static int me_qpel_cmp(const me_ctx *c, int mx, int my, int satd)
{   /* get_ref + fpelcmp / mbcmp_unaligned, me.c:64-71,644-652 */
    u8 pix[16 * 16];
    int stride = 16;
    u8 *src = mcf.get_ref(pix, &stride, (u8 **)c->fref, c->sy, mx, my, c->bw, c->bh);
    return (satd ? pixf.satd : pixf.sad)[c->pix](c->fenc, c->sy, src, stride) + c->cmx[mx] + c->cmy[my];
}
static int me_satd_chroma(const me_ctx *c, int mx, int my, int bcost, int chroma_me, int satd)
{   /* COST_MV_SATD's cost with the chroma terms, me.c:654-677 */
    int cost = me_qpel_cmp(c, mx, my, satd);
    if (chroma_me && cost < bcost) {
        u8 pix[8 * 8];
        mcf.mc_chroma(pix, 8, c->fref[4], c->sc, mx, my, c->bw / 2, c->bh / 2);
        cost += (satd ? pixf.satd : pixf.sad)[c->pix + 3](c->fenc_u, c->sc, pix, 8);
        if (cost < bcost) {
            mcf.mc_chroma(pix, 8, c->fref[5], c->sc, mx, my, c->bw / 2, c->bh / 2);
            cost += (satd ? pixf.satd : pixf.sad)[c->pix + 3](c->fenc_v, c->sc, pix, 8);
        }
    }
    return cost;
}

static int g_me_lossless;             /* set by the slice twin for a lossless chain: every mbcmp is SAD */
/* returns cost; *pmvx,*pmvy the vector; *thresh the half-pel threshold (NULL = none) */
static int me_search16(const me_ctx *c, const i16 mvp[2], const i16 (*mvc)[2], int n_mvc, int method, int me_range, int subme,
                       int chroma_me, int *thresh, int *pmvx, int *pmvy, int *pcost_mv)
{
    const int satd = subme > 1 && !g_me_lossless;   /* mbcmp = SATD above subme 1 unless lossless; fpelcmp stays SAD (encoder.c:608-618) */
    int bmx = clip3i(mvp[0], c->fmin[0] * 4, c->fmax[0] * 4), bmy = clip3i(mvp[1], c->fmin[1] * 4, c->fmax[1] * 4);
    int pmx = (bmx + 2) >> 2, pmy = (bmy + 2) >> 2;
    int bcost = ME_COST_MAX, bpx = 0, bpy = 0, bpcost = ME_COST_MAX, cost, i;
#define TRY(mx, my) do { cost = me_fpel_cost(c, mx, my); if (cost < bcost) { bcost = cost; bmx = mx; bmy = my; } } while (0)
#define INRANGE(x, y) ((x) >= c->fmin[0] && (x) <= c->fmax[0] && (y) >= c->fmin[1] && (y) <= c->fmax[1])
    if (subme >= 3) {
        int px = bmx, py = bmy;
        cost = me_qpel_cmp(c, px, py, 0);
        if (cost < bpcost) { bpcost = cost; bpx = px; bpy = py; }
        for (i = 0; i < n_mvc; i++)
            if ((mvc[i][0] | mvc[i][1]) && (mvc[i][0] != (i16)px || mvc[i][1] != (i16)py)) {
                int mx = clip3i(mvc[i][0], c->fmin[0] * 4, c->fmax[0] * 4), my = clip3i(mvc[i][1], c->fmin[1] * 4, c->fmax[1] * 4);
                cost = me_qpel_cmp(c, mx, my, 0);
                if (cost < bpcost) { bpcost = cost; bpx = mx; bpy = my; }
            }
        bmx = (bpx + 2) >> 2; bmy = (bpy + 2) >> 2;
        { int tx = bmx, ty = bmy; TRY(tx, ty); }
    } else {
        TRY(pmx, pmy);
        bcost -= c->cmx[pmx << 2] + c->cmy[pmy << 2];
        for (i = 0; i < n_mvc; i++) {
            int mx = (mvc[i][0] + 2) >> 2, my = (mvc[i][1] + 2) >> 2;
            if ((mx | my) && ((mx - bmx) | (my - bmy))) {
                mx = clip3i(mx, c->fmin[0], c->fmax[0]); my = clip3i(my, c->fmin[1], c->fmax[1]);
                TRY(mx, my);
            }
        }
    }
    TRY(0, 0);
    if (method == 0) {                                           /* diamond, me.c:233-244 */
        i = 0;
        do {
            int ox = bmx, oy = bmy;
            TRY(ox, oy - 1); TRY(ox, oy + 1); TRY(ox - 1, oy); TRY(ox + 1, oy);
            if (bmx == ox && bmy == oy) break;
            if (!INRANGE(bmx, bmy)) break;
        } while (++i < me_range);
    }
    if (method == 3) {
        /* X264_ME_ESA, me.c:449-600.  The reference drops, row by row, every position whose ADS bound (sum of |differences of block
         * sums| + mv cost <= SAD + mv cost) is not below the best cost so far; such a position could not have passed COST_MV's strict
         * '<' either, so the walk equals the plain raster scan of its own "#if 0" branch -- over min_x .. min_x + width - 1 with the
         * width rounded up to a multiple of 4 (:456), which can stop one column short of max_x or run up to three past it. */
        const int min_x = bmx - me_range > c->fmin[0] ? bmx - me_range : c->fmin[0], min_y = bmy - me_range > c->fmin[1] ? bmy - me_range : c->fmin[1];
        const int max_x = bmx + me_range < c->fmax[0] ? bmx + me_range : c->fmax[0], max_y = bmy + me_range < c->fmax[1] ? bmy + me_range : c->fmax[1];
        const int width = (max_x - min_x + 3) & ~3;
        for (int my = min_y; my <= max_y; my++)
            for (int mx = min_x; mx < min_x + width; mx++) TRY(mx, my);
    }
    int do_hex = method == 1, hex_range = me_range;
    if (method == 2) {                                           /* uneven-cross multi-hexagon, me.c:306-447 */
        static const int size_shift[7] = {0, 1, 1, 2, 3, 3, 4};
        static const int range_mul[4][4] = {{3, 3, 4, 4}, {3, 4, 4, 4}, {4, 4, 4, 5}, {4, 4, 5, 6}};
        static const int hex4[16][2] = {{-4, 2}, {-4, 1}, {-4, 0}, {-4, -1}, {-4, -2}, {4, -2}, {4, -1}, {4, 0}, {4, 1}, {4, 2},
                                        {2, 3}, {0, 4}, {-2, 3}, {-2, -3}, {0, -4}, {2, -3}};
        int ucost1, ucost2, cross_start = 1, omx, omy, done = 0, j;
#define SAD_THRESH(v) (bcost < ((v) >> size_shift[c->pix]))
#define X4(ax, ay, bx_, by_, cx_, cy_, dx_, dy_) do { TRY(omx + (ax), omy + (ay)); TRY(omx + (bx_), omy + (by_)); TRY(omx + (cx_), omy + (cy_)); TRY(omx + (dx_), omy + (dy_)); } while (0)
#define DIA1(mx_, my_) do { omx = (mx_); omy = (my_); X4(0, -1, 0, 1, -1, 0, 1, 0); } while (0)
#define CROSS(start, x_max, y_max) do { \
        i = (start); \
        if ((x_max) <= (c->fmax[0] - omx < omx - c->fmin[0] ? c->fmax[0] - omx : omx - c->fmin[0])) \
            for (; i < (x_max) - 2; i += 4) X4(i, 0, -i, 0, i + 2, 0, -i - 2, 0); \
        for (; i < (x_max); i += 2) { if (omx + i <= c->fmax[0]) TRY(omx + i, omy); if (omx - i >= c->fmin[0]) TRY(omx - i, omy); } \
        i = (start); \
        if ((y_max) <= (c->fmax[1] - omy < omy - c->fmin[1] ? c->fmax[1] - omy : omy - c->fmin[1])) \
            for (; i < (y_max) - 2; i += 4) X4(0, i, 0, -i, 0, i + 2, 0, -i - 2); \
        for (; i < (y_max); i += 2) { if (omy + i <= c->fmax[1]) TRY(omx, omy + i); if (omy - i >= c->fmin[1]) TRY(omx, omy - i); } } while (0)
        ucost1 = bcost;
        DIA1(pmx, pmy);
        if (pmx | pmy) DIA1(0, 0);
        if (c->pix == X264HIP_PIXEL_4x4) { do_hex = 1; goto umh_end; }   /* "if(i_pixel == PIXEL_4x4) goto me_hex2", me.c:323 */
        ucost2 = bcost;
        if ((bmx | bmy) && ((bmx - pmx) | (bmy - pmy))) { int tx = bmx, ty = bmy; DIA1(tx, ty); }
        if (bcost == ucost2) cross_start = 3;
        omx = bmx; omy = bmy;
        if (bcost == ucost2 && SAD_THRESH(2000)) {               /* early termination */
            X4(0, -2, -1, -1, 1, -1, -2, 0);
            X4(2, 0, -1, 1, 1, 1, 0, 2);
            if (bcost == ucost1 && SAD_THRESH(500)) done = 1;
            else if (bcost == ucost2) {
                int range = (hex_range >> 1) | 1;
                CROSS(3, range, range);
                X4(-1, -2, 1, -2, -2, -1, 2, -1);
                X4(-2, 1, 2, 1, -1, 2, 1, 2);
                if (bcost == ucost2) done = 1;
                else cross_start = range + 2;
            }
        }
        if (!done) {
            if (n_mvc) {                                         /* adaptive search range */
                int mvd, denom = 1, sad_ctx, mvd_ctx;
                if (n_mvc == 1) mvd = c->pix == X264HIP_PIXEL_16x16 ? 25 : abs(mvp[0] - mvc[0][0]) + abs(mvp[1] - mvc[0][1]);
                else {
                    denom = n_mvc - 1; mvd = 0;
                    if (c->pix != X264HIP_PIXEL_16x16) { mvd = abs(mvp[0] - mvc[0][0]) + abs(mvp[1] - mvc[0][1]); denom++; }
                    for (j = 0; j < n_mvc - 1; j++) mvd += abs(mvc[j][0] - mvc[j + 1][0]) + abs(mvc[j][1] - mvc[j + 1][1]);   /* x264_predictor_difference */
                }
                sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
                mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
                hex_range = hex_range * range_mul[mvd_ctx][sad_ctx] / 4;
            }
            CROSS(cross_start, hex_range, hex_range / 2);
            X4(-2, -2, -2, 2, 2, -2, 2, 2);
            omx = bmx; omy = bmy;                                /* hexagon grid */
            i = 1;
            do {
                int lim = c->fmax[0] - omx;
                if (omx - c->fmin[0] < lim) lim = omx - c->fmin[0];
                if (c->fmax[1] - omy < lim) lim = c->fmax[1] - omy;
                if (omy - c->fmin[1] < lim) lim = omy - c->fmin[1];
                if (4 * i > lim) {
                    for (j = 0; j < 16; j++) { int mx = omx + hex4[j][0] * i, my = omy + hex4[j][1] * i; if (INRANGE(mx, my)) TRY(mx, my); }
                } else
                    for (j = 0; j < 16; j++) TRY(omx + hex4[j][0] * i, omy + hex4[j][1] * i);
            } while (++i <= hex_range / 4);
            if (bmy <= c->fmax[1]) do_hex = 1;
        }
umh_end:;
#undef SAD_THRESH
#undef X4
#undef DIA1
#undef CROSS
    }
    if (do_hex) {                                                /* hexagon, me.c:246-305 */
        const int me_range_h = hex_range;
        int dir = -2, costs[6], ox, oy;
        static const int first[6][2] = {{-2,0},{-1,2},{1,2},{2,0},{1,-2},{-1,-2}};
        for (i = 0; i < 6; i++) costs[i] = me_fpel_cost(c, bmx + first[i][0], bmy + first[i][1]);
        for (i = 0; i < 6; i++) if (costs[i] < bcost) { bcost = costs[i]; dir = i; }
        if (dir != -2) {
            bmx += me_hex2[dir + 1][0]; bmy += me_hex2[dir + 1][1];
            for (i = 1; i < me_range_h / 2 && INRANGE(bmx, bmy); i++) {
                int odir = me_mod6m1[dir + 1], k;
                for (k = 0; k < 3; k++) costs[k] = me_fpel_cost(c, bmx + me_hex2[odir + k][0], bmy + me_hex2[odir + k][1]);
                dir = -2;
                for (k = 0; k < 3; k++) if (costs[k] < bcost) { bcost = costs[k]; dir = odir - 1 + k; }
                if (dir == -2) break;
                bmx += me_hex2[dir + 1][0]; bmy += me_hex2[dir + 1][1];
            }
        }
        ox = bmx; oy = bmy;                                      /* square refine */
        TRY(ox, oy - 1); TRY(ox, oy + 1); TRY(ox - 1, oy); TRY(ox + 1, oy);
        TRY(ox - 1, oy - 1); TRY(ox - 1, oy + 1); TRY(ox + 1, oy - 1); TRY(ox + 1, oy + 1);
    }
    int mvx, mvy, mcost;
    if (bpcost < bcost) { mvx = bpx; mvy = bpy; mcost = bpcost; }
    else { mvx = bmx << 2; mvy = bmy << 2; mcost = bcost; }
    if (pcost_mv) *pcost_mv = c->cmx[mvx] + c->cmy[mvy];          /* m->cost_mv, me.c:615 */
    if (bmx == pmx && bmy == pmy && subme < 3) mcost += c->cmx[mvx] + c->cmy[mvy];
    if (subme >= 2) {                                            /* refine_subpel(.., b_refine_qpel = 0), me.c:680-778 */
        int hpel = me_subpel_iters[subme][2], qpel = me_subpel_iters[subme][3];
        int bx = mvx, by = mvy, bc = mcost, odir = -1, bdir;
        if (hpel && subme < 3) {
            int mx = clip3i(mvp[0], c->smin[0], c->smax[0]), my = clip3i(mvp[1], c->smin[1], c->smax[1]);
            if ((mx - bx) | (my - by)) { cost = me_qpel_cmp(c, mx, my, 0); if (cost < bc) { bc = cost; bx = mx; by = my; } }
        }
        for (i = hpel; i > 0; i--) {
            int ox = bx, oy = by, c0 = me_qpel_cmp(c, ox, oy - 2, 0), c1 = me_qpel_cmp(c, ox, oy + 2, 0);
            int c2 = me_qpel_cmp(c, ox - 2, oy, 0), c3 = me_qpel_cmp(c, ox + 2, oy, 0);
            if (c0 < bc) { bc = c0; by = oy - 2; }
            if (c1 < bc) { bc = c1; by = oy + 2; }
            if (c2 < bc) { bc = c2; bx = ox - 2; by = oy; }
            if (c3 < bc) { bc = c3; bx = ox + 2; by = oy; }
            if (bx == ox && by == oy) break;
        }
        if (by > c->smax[1]) by = c->smax[1];
        bc = ME_COST_MAX;
        cost = me_satd_chroma(c, bx, by, bc, chroma_me, satd);
        if (cost < bc) bc = cost;
        if (thresh) {
            if (((bc * 7) >> 3) > *thresh) { *pmvx = bx; *pmvy = by; return bc; }
            if (bc < *thresh) *thresh = bc;
        }
        bdir = -1;
        for (i = qpel; i > 0; i--) {
            static const int dq[4][2] = {{0,-1},{0,1},{-1,0},{1,0}};
            int ox = bx, oy = by, d;
            odir = bdir;
            for (d = 0; d < 4; d++)
                if ((d ^ 1) != odir) {
                    cost = me_satd_chroma(c, ox + dq[d][0], oy + dq[d][1], bc, chroma_me, satd);
                    if (cost < bc) { bc = cost; bx = ox + dq[d][0]; by = oy + dq[d][1]; bdir = d; }
                }
            if (bx == ox && by == oy) break;
        }
        if (by > c->smax[1]) {
            by = c->smax[1]; bc = ME_COST_MAX;
            cost = me_satd_chroma(c, bx, by, bc, chroma_me, satd);
            if (cost < bc) bc = cost;
        }
        mvx = bx; mvy = by; mcost = bc;
        if (pcost_mv) *pcost_mv = c->cmx[bx] + c->cmy[by];       /* me.c:777 */
    } else if (mvy > c->smax[1]) mvy = c->smax[1];
#undef TRY
#undef INRANGE
    *pmvx = mvx; *pmvy = mvy;
    return mcost;
}

void x264o_frame_me_search16(u8 *fy, u8 *fu, u8 *fv, u8 *const *refs /* [n][6] */, int n_refs, int mb_w, int mb_h, int sy, int sc,
                             int method, int me_range, int subme, int chroma_me, int mv_range, const i16 *cost_mv, int cost_center,
                             const i16 *mvp, const i16 *mvc, const u8 *n_mvc, const int32_t *ref_cost,
                             i16 *out_mv, int32_t *out_cost, int32_t *best)
{
    init();
    for (int mb = 0; mb < mb_w * mb_h; mb++) {
        int mbx = mb % mb_w, mby = mb / mb_w, sp[4], fp[4];
        int oy = 16 * mby * sy + 16 * mbx, oc = 8 * mby * sc + 8 * mbx;
        int thresh = 0x7fffffff, bestc = 0x7fffffff;
        mv_limits(mb_w, mb_h, mbx, mby, mv_range, sp, fp);
        for (int r = 0; r < n_refs; r++) {
            me_ctx c;
            const i16 *p = mvp + ((size_t)mb * n_refs + r) * 2;
            c.fenc = fy + oy; c.fenc_u = fu + oc; c.fenc_v = fv + oc; c.sy = sy; c.sc = sc; c.pix = X264HIP_PIXEL_16x16; c.bw = c.bh = 16;
            for (int k = 0; k < 4; k++) c.fref[k] = refs[6 * r + k] + oy;
            c.fref[4] = refs[6 * r + 4] + oc; c.fref[5] = refs[6 * r + 5] + oc;
            c.cmx = cost_mv + cost_center - p[0]; c.cmy = cost_mv + cost_center - p[1];
            c.fmin[0] = fp[0]; c.fmax[0] = fp[1]; c.fmin[1] = fp[2]; c.fmax[1] = fp[3];
            c.smin[0] = sp[0]; c.smax[0] = sp[1]; c.smin[1] = sp[2]; c.smax[1] = sp[3];
            int mvx, mvy, cost;
            thresh -= ref_cost[r];
            cost = me_search16(&c, p, (const i16 (*)[2])(mvc + ((size_t)mb * n_refs + r) * 16), n_mvc[mb * n_refs + r], method, me_range,
                               subme, chroma_me, n_refs > 1 ? &thresh : 0, &mvx, &mvy, 0);
            cost += ref_cost[r];
            thresh += ref_cost[r];
            out_mv[((size_t)mb * n_refs + r) * 2] = mvx; out_mv[((size_t)mb * n_refs + r) * 2 + 1] = mvy;
            out_cost[(size_t)mb * n_refs + r] = cost;
            if (cost < bestc) { bestc = cost; best[4 * mb] = r; best[4 * mb + 1] = mvx; best[4 * mb + 2] = mvy; best[4 * mb + 3] = cost; }
        }
    }
}

/* ---------------------------------------------------------------- residual
 * x264_macroblock_encode, inter 16x16 branch (R/encoder/macroblock.c:596-768)
 * with b_dct_decimate on, no trellis / noise reduction / lossless, followed by
 * x264_mb_encode_8x8_chroma(b_inter = 1) (:272-363).  The macroblock is moved
 * through fenc_buf / fdec_buf-shaped scratch (strides 16 / 32) exactly as
 * x264_macroblock_cache_load does, so the table entries see the strides they
 * were written for. */
#define FENC 16
#define FDEC 32
static const u8 blk_x[16] = {0, 4, 0, 4, 8, 12, 8, 12, 0, 4, 0, 4, 8, 12, 8, 12};
static const u8 blk_y[16] = {0, 0, 4, 4, 0, 0, 4, 4, 8, 8, 12, 12, 8, 8, 12, 12};

/* general form: any P partition (one vector per 4x4 block, raster order) and several references.
 * refs[i] = {full, H, V, HV, U, V} planes of reference i; ref8 = per-8x8 index or NULL (all 0);
 * mv_per_mb = 1 (one 16x16 vector) or 16. */
void x264o_frame_inter_residual_mp(u8 *fy, u8 *fu, u8 *fv, u8 *const *refs /* [n][6] */, int n_refs,
                                   u8 *dy, u8 *du, u8 *dv,
                                   int mb_w, int mb_h, int sy, int sc, int qp, int qpc, int transform8x8, int interlaced,
                                   const u16 *q4mf, const u16 *q4bias, const u16 *q8mf, const u16 *q8bias,
                                   const int32_t *dq4, const int32_t *dq8, const i16 *mv, int mv_per_mb, const int8_t *ref8,
                                   i16 *levels_y, i16 *levels_c, i16 *dc_c, int32_t *cbp_out, u8 *nnz_out);

void x264o_frame_inter_residual(u8 *fy, u8 *fu, u8 *fv,                 /* source planes */
                                u8 *r0, u8 *r1, u8 *r2, u8 *r3, u8 *ru, u8 *rv,   /* reference: 4 luma planes + chroma */
                                u8 *dy, u8 *du, u8 *dv,                 /* reconstruction planes */
                                int mb_w, int mb_h, int sy, int sc, int qp, int qpc, int transform8x8, int interlaced,
                                const u16 *q4mf, const u16 *q4bias, const u16 *q8mf, const u16 *q8bias,
                                const int32_t *dq4, const int32_t *dq8, const i16 *mv,
                                i16 *levels_y, i16 *levels_c, i16 *dc_c, int32_t *cbp_out, u8 *nnz_out)
{
    u8 *one[6] = {r0, r1, r2, r3, ru, rv};
    x264o_frame_inter_residual_mp(fy, fu, fv, one, 1, dy, du, dv, mb_w, mb_h, sy, sc, qp, qpc, transform8x8, interlaced,
                                  q4mf, q4bias, q8mf, q8bias, dq4, dq8, mv, 1, 0, levels_y, levels_c, dc_c, cbp_out, nnz_out);
}

/* x264_mb_mc for a P macroblock given per-4x4 vectors and per-8x8 reference indices
 * (R/common/macroblock.c:462-546: every partition is mc_luma + mc_chroma with its own vector) */
static void mb_mc(u8 *fd_y, u8 *fd_u, u8 *fd_v, u8 *const *refs, int oy, int ocs, int sy, int sc,
                  const i16 *mv, int mv_per_mb, const int8_t *ref8)
{
    for (int by = 0; by < 4; by++)
        for (int bx = 0; bx < 4; bx++) {
            const i16 *m = mv_per_mb == 1 ? mv : mv + (bx + 4 * by) * 2;
            u8 *const *r = refs + 6 * (ref8 ? ref8[(by >> 1) * 2 + (bx >> 1)] : 0);
            int o = oy + 4 * by * sy + 4 * bx, oc = ocs + 2 * by * sc + 2 * bx;
            u8 *src4[4] = {r[0] + o, r[1] + o, r[2] + o, r[3] + o};
            mcf.mc_luma(fd_y + 4 * by * FDEC + 4 * bx, FDEC, src4, sy, m[0], m[1], 4, 4);
            mcf.mc_chroma(fd_u + 2 * by * FDEC + 2 * bx, FDEC, r[4] + oc, sc, m[0], m[1], 2, 2);
            mcf.mc_chroma(fd_v + 2 * by * FDEC + 2 * bx, FDEC, r[5] + oc, sc, m[0], m[1], 2, 2);
        }
}

void x264o_frame_inter_residual_mp(u8 *fy, u8 *fu, u8 *fv, u8 *const *refs, int n_refs,
                                   u8 *dy, u8 *du, u8 *dv,
                                   int mb_w, int mb_h, int sy, int sc, int qp, int qpc, int transform8x8, int interlaced,
                                   const u16 *q4mf, const u16 *q4bias, const u16 *q8mf, const u16 *q8bias,
                                   const int32_t *dq4, const int32_t *dq8, const i16 *mv, int mv_per_mb, const int8_t *ref8,
                                   i16 *levels_y, i16 *levels_c, i16 *dc_c, int32_t *cbp_out, u8 *nnz_out)
{
    init();
    (void)n_refs;
    u8 fenc[24 * FENC], fdec[27 * FDEC];
    u8 *fe_y = fenc, *fe_u = fenc + 16 * FENC, *fe_v = fenc + 16 * FENC + 8;
    u8 *fd_y = fdec + 2 * FDEC, *fd_u = fdec + 19 * FDEC, *fd_v = fdec + 19 * FDEC + 16;
    u16 *mf4y = (u16 *)q4mf + (1 * 52 + qp) * 16, *b4y = (u16 *)q4bias + (1 * 52 + qp) * 16;
    u16 *mf4c = (u16 *)q4mf + (3 * 52 + qpc) * 16, *b4c = (u16 *)q4bias + (3 * 52 + qpc) * 16;
    u16 *mf8y = (u16 *)q8mf + (1 * 52 + qp) * 64, *b8y = (u16 *)q8bias + (1 * 52 + qp) * 64;
    int (*dq4y)[4][4] = (int (*)[4][4])(dq4 + 1 * 96), (*dq4c)[4][4] = (int (*)[4][4])(dq4 + 3 * 96);
    int (*dq8y)[8][8] = (int (*)[8][8])(dq8 + 1 * 384);
    x264hip_zigzag_function_t *zz = &zigf[!!interlaced];
    for (int mb = 0; mb < mb_w * mb_h; mb++) {
        int mbx = mb % mb_w, mby = mb / mb_w;
        int oy = 16 * mby * sy + 16 * mbx, ocs = 8 * mby * sc + 8 * mbx;
        i16 *ly = levels_y + mb * 256, *lc = levels_c + mb * 128, *ldc = dc_c + mb * 8;
        u8 *nnz = nnz_out + mb * 26;
        memset(ly, 0, 512); memset(lc, 0, 256); memset(ldc, 0, 16); memset(nnz, 0, 26);
        for (int y = 0; y < 16; y++) memcpy(fe_y + y * FENC, fy + oy + y * sy, 16);
        for (int y = 0; y < 8; y++) { memcpy(fe_u + y * FENC, fu + ocs + y * sc, 8); memcpy(fe_v + y * FENC, fv + ocs + y * sc, 8); }
        mb_mc(fd_y, fd_u, fd_v, refs, oy, ocs, sy, sc, mv + 2 * mv_per_mb * mb, mv_per_mb, ref8 ? ref8 + 4 * mb : 0);
        int cbp_luma = 0, decimate_mb = 0;
        if (transform8x8) {
            i16 dct8[4][8][8];
            dctf.sub16x16_dct8(dct8, fe_y, fd_y);
            for (int idx = 0; idx < 4; idx++) {
                int nz = quantf.quant_8x8(dct8[idx], mf8y, b8y);
                if (nz) {
                    zz->scan_8x8(ly + 64 * idx, dct8[idx]);
                    int s = quantf.decimate_score64(ly + 64 * idx);
                    decimate_mb += s;
                    if (s >= 4) cbp_luma |= 1 << idx;
                }
            }
            if (decimate_mb < 6) cbp_luma = 0;
            else
                for (int idx = 0; idx < 4; idx++)
                    if (cbp_luma & (1 << idx)) {
                        quantf.dequant_8x8(dct8[idx], dq8y, qp);
                        dctf.add8x8_idct8(fd_y + (idx & 1) * 8 + (idx >> 1) * 8 * FDEC, dct8[idx]);
                        for (int k = 0; k < 4; k++) nnz[4 * idx + k] = 1;
                    }
        } else {
            i16 dct4[16][4][4];
            dctf.sub16x16_dct(dct4, fe_y, fd_y);
            for (int i8 = 0; i8 < 4; i8++) {
                int dec8 = 0;
                for (int i4 = 0; i4 < 4; i4++) {
                    int idx = 4 * i8 + i4;
                    int nz = quantf.quant_4x4(dct4[idx], mf4y, b4y);
                    nnz[idx] = nz;
                    if (nz) {
                        zz->scan_4x4(ly + 16 * idx, dct4[idx]);
                        quantf.dequant_4x4(dct4[idx], dq4y, qp);
                        if (dec8 < 6) dec8 += quantf.decimate_score16(ly + 16 * idx);
                    }
                }
                decimate_mb += dec8;
                if (dec8 < 4) nnz[4 * i8] = nnz[4 * i8 + 1] = nnz[4 * i8 + 2] = nnz[4 * i8 + 3] = 0;
                else cbp_luma |= 1 << i8;
            }
            if (decimate_mb < 6) { cbp_luma = 0; memset(nnz, 0, 16); }
            else
                for (int i8 = 0; i8 < 4; i8++)
                    if (cbp_luma & (1 << i8))
                        dctf.add8x8_idct(fd_y + (i8 & 1) * 8 + (i8 >> 1) * 8 * FDEC, &dct4[4 * i8]);
        }
        /* chroma, R/encoder/macroblock.c:272-363 */
        int cbp_chroma = 0;
        for (int ch = 0; ch < 2; ch++) {
            u8 *ps = ch ? fe_v : fe_u, *pd = ch ? fd_v : fd_u;
            i16 d4[4][4][4], d2[2][2];
            int score = 0, nz_ac = 0;
            dctf.sub8x8_dct(d4, ps, pd);
            {   /* dct2x2dc, :73-85 */
                int a = d4[0][0][0] + d4[1][0][0], b = d4[2][0][0] + d4[3][0][0];
                int c = d4[0][0][0] - d4[1][0][0], d = d4[2][0][0] - d4[3][0][0];
                d2[0][0] = a + b; d2[1][0] = c + d; d2[0][1] = a - b; d2[1][1] = c - d;
                d4[0][0][0] = d4[1][0][0] = d4[2][0][0] = d4[3][0][0] = 0;
            }
            for (int i = 0; i < 4; i++) {
                int nz = quantf.quant_4x4(d4[i], mf4c, b4c);
                nnz[16 + 4 * ch + i] = nz;
                if (nz) {
                    nz_ac = 1;
                    zz->scan_4x4(lc + (4 * ch + i) * 16, d4[i]);
                    quantf.dequant_4x4(d4[i], dq4c, qpc);
                    score += quantf.decimate_score15(lc + (4 * ch + i) * 16);
                }
            }
            int nz_dc = quantf.quant_2x2_dc(d2, mf4c[0] >> 1, b4c[0] << 1);
            nnz[24 + ch] = nz_dc;
            /* IDCT_DEQUANT_START, :40-51 */
            int e0 = d2[0][0] + d2[0][1], e1 = d2[1][0] + d2[1][1], e2 = d2[0][0] - d2[0][1], e3 = d2[1][0] - d2[1][1];
            int dmf = dq4c[qpc % 6][0][0], qbits = qpc / 6 - 5;
            if (qbits > 0) { dmf <<= qbits; qbits = 0; }
            if (score < 7 || !nz_ac) {
                nnz[16 + 4 * ch] = nnz[17 + 4 * ch] = nnz[18 + 4 * ch] = nnz[19 + 4 * ch] = 0;
                if (!nz_dc) continue;
                ldc[4 * ch] = d2[0][0]; ldc[4 * ch + 1] = d2[1][0]; ldc[4 * ch + 2] = d2[0][1]; ldc[4 * ch + 3] = d2[1][1];
                i16 dd[2][2];
                dd[0][0] = (e0 + e1) * dmf >> -qbits; dd[0][1] = (e0 - e1) * dmf >> -qbits;
                dd[1][0] = (e2 + e3) * dmf >> -qbits; dd[1][1] = (e2 - e3) * dmf >> -qbits;
                dctf.add8x8_idct_dc(pd, dd);
            } else {
                cbp_chroma = 1;
                if (nz_dc) {
                    ldc[4 * ch] = d2[0][0]; ldc[4 * ch + 1] = d2[1][0]; ldc[4 * ch + 2] = d2[0][1]; ldc[4 * ch + 3] = d2[1][1];
                    d4[0][0][0] = (e0 + e1) * dmf >> -qbits; d4[1][0][0] = (e0 - e1) * dmf >> -qbits;
                    d4[2][0][0] = (e2 + e3) * dmf >> -qbits; d4[3][0][0] = (e2 - e3) * dmf >> -qbits;
                }
                dctf.add8x8_idct(pd, d4);
            }
        }
        if (cbp_chroma) cbp_chroma = 2;
        else if (nnz[24] | nnz[25]) cbp_chroma = 1;
        cbp_out[mb] = cbp_luma | (cbp_chroma << 4);
        (void)blk_x; (void)blk_y;
        for (int y = 0; y < 16; y++) memcpy(dy + oy + y * sy, fd_y + y * FDEC, 16);
        for (int y = 0; y < 8; y++) { memcpy(du + ocs + y * sc, fd_u + y * FDEC, 8); memcpy(dv + ocs + y * sc, fd_v + y * FDEC, 8); }
    }
}

/* x264_macroblock_probe_skip (P path), R/encoder/macroblock.c:797-883, for every macroblock */
void x264o_frame_probe_skip(u8 *fy, u8 *fu, u8 *fv, u8 *r0, u8 *r1, u8 *r2, u8 *r3, u8 *ru, u8 *rv,
                            int mb_w, int mb_h, int sy, int sc, int qp, int qpc, int lambda2_chroma, int interlaced,
                            const u16 *q4mf, const u16 *q4bias, const i16 *pskip_mv, u8 *skip_out)
{
    init();
    u8 fenc[24 * FENC], fdec[27 * FDEC];
    u8 *fe_y = fenc, *fe_c[2] = {fenc + 16 * FENC, fenc + 16 * FENC + 8};
    u8 *fd_y = fdec + 2 * FDEC, *fd_c[2] = {fdec + 19 * FDEC, fdec + 19 * FDEC + 16};
    u16 *mf4y = (u16 *)q4mf + (1 * 52 + qp) * 16, *b4y = (u16 *)q4bias + (1 * 52 + qp) * 16;
    u16 *mf4c = (u16 *)q4mf + (3 * 52 + qpc) * 16, *b4c = (u16 *)q4bias + (3 * 52 + qpc) * 16;
    x264hip_zigzag_function_t *zz = &zigf[!!interlaced];
    int thresh = (lambda2_chroma + 32) >> 6;
    for (int mb = 0; mb < mb_w * mb_h; mb++) {
        int mbx = mb % mb_w, mby = mb / mb_w;
        int oy = 16 * mby * sy + 16 * mbx, ocs = 8 * mby * sc + 8 * mbx;
        int mvx = clip3i(pskip_mv[2 * mb], 4 * (-16 * mbx - 24), 4 * (16 * (mb_w - mbx - 1) + 24));
        int mvy = clip3i(pskip_mv[2 * mb + 1], 4 * (-16 * mby - 24), 4 * (16 * (mb_h - mby - 1) + 24));
        u8 *pc[2] = {fu, fv}, *rc[2] = {ru, rv};
        for (int y = 0; y < 16; y++) memcpy(fe_y + y * FENC, fy + oy + y * sy, 16);
        for (int ch = 0; ch < 2; ch++)
            for (int y = 0; y < 8; y++) memcpy(fe_c[ch] + y * FENC, pc[ch] + ocs + y * sc, 8);
        u8 *src4[4] = {r0 + oy, r1 + oy, r2 + oy, r3 + oy};
        mcf.mc_luma(fd_y, FDEC, src4, sy, mvx, mvy, 16, 16);
        int ok = 1, dec = 0;
        i16 d4[4][4][4], d2[2][2], scan[16];
        for (int i8 = 0; i8 < 4 && ok; i8++) {
            dctf.sub8x8_dct(d4, fe_y + (i8 & 1) * 8 + (i8 >> 1) * 8 * FENC, fd_y + (i8 & 1) * 8 + (i8 >> 1) * 8 * FDEC);
            for (int i4 = 0; i4 < 4; i4++) {
                if (!quantf.quant_4x4(d4[i4], mf4y, b4y)) continue;
                zz->scan_4x4(scan, d4[i4]);
                dec += quantf.decimate_score16(scan);
                if (dec >= 6) { ok = 0; break; }
            }
        }
        for (int ch = 0; ch < 2 && ok; ch++) {
            mcf.mc_chroma(fd_c[ch], FDEC, rc[ch] + ocs, sc, mvx, mvy, 8, 8);
            if (pixf.ssd[X264HIP_PIXEL_8x8](fd_c[ch], FDEC, fe_c[ch], FENC) < thresh) continue;
            dctf.sub8x8_dct(d4, fe_c[ch], fd_c[ch]);
            int a = d4[0][0][0] + d4[1][0][0], b = d4[2][0][0] + d4[3][0][0];
            int c = d4[0][0][0] - d4[1][0][0], d = d4[2][0][0] - d4[3][0][0];
            d2[0][0] = a + b; d2[1][0] = c + d; d2[0][1] = a - b; d2[1][1] = c - d;
            d4[0][0][0] = d4[1][0][0] = d4[2][0][0] = d4[3][0][0] = 0;
            if (quantf.quant_2x2_dc(d2, mf4c[0] >> 1, b4c[0] << 1)) { ok = 0; break; }
            dec = 0;
            for (int i4 = 0; i4 < 4; i4++) {
                if (!quantf.quant_4x4(d4[i4], mf4c, b4c)) continue;
                zz->scan_4x4(scan, d4[i4]);
                dec += quantf.decimate_score15(scan);
                if (dec >= 7) { ok = 0; break; }
            }
        }
        skip_out[mb] = (u8)ok;
    }
}

/* ----------------------------------------------------------------- deblock
 * x264_frame_deblock_row for every row of a progressive P frame
 * (R/common/frame.c:621-792), macroblocks in raster order, vertical edges
 * then horizontal edges of each.  Tables: the standard's alpha / beta / tc0
 * (R/common/frame.c:377-417).
 *   mb_type: 0 inter, 1 intra, 2 P_SKIP; nnz: [mb][26] in block z-order (as
 *   x264hip_inter_residual_frame writes it); mv: [mb][16][2] qpel, 4x4
 *   blocks in raster order; ref: [mb][4] per 8x8. */
static const u8 db_alpha[52 + 24] = {
    0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,4,4,5,6,7,8,9,10,12,13,15,17,20,22,25,28,32,36,40,45,50,56,63,71,
    80,90,101,113,127,144,162,182,203,226,255,255, 255,255,255,255,255,255,255,255,255,255,255,255};
static const u8 db_beta[52 + 24] = {
    0,0,0,0,0,0,0,0,0,0,0,0, 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,2,2,3,3,3,3,4,4,4,6,6,7,7,8,8,9,9,10,10,11,11,12,12,
    13,13,14,14,15,15,16,16,17,17,18,18, 18,18,18,18,18,18,18,18,18,18,18,18};
static const int8_t db_tc0[52 + 24][3] = {   /* bS 1..3; bS 0 is -1 */
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
    {0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},{1,1,1},{1,1,1},{1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},
    {2,2,3},{2,2,4},{2,3,4},{2,3,4},{3,3,5},{3,4,6},{3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},
    {9,12,18},{10,13,20},{11,15,23},{13,17,25},
    {13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25},{13,17,25}};
static const u8 db_chroma_qp[52 + 24] = {
    0,0,0,0,0,0,0,0,0,0,0,0, 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,
    29,30,31,32,32,33,34,34,35,35,36,36,37,37,37,38,38,38,39,39,39,39, 39,39,39,39,39,39,39,39,39,39,39,39};
static int z_of(int x, int y) { return (y >> 1) * 8 + (x >> 1) * 4 + (y & 1) * 2 + (x & 1); }

static void db_edge(u8 *p1, u8 *p2, int stride, const u8 bS[4], int qp, int chroma, int a_off, int b_off,
                    x264hip_deblock_inter_t f)
{   /* deblock_edge, R/common/frame.c:588-606 */
    int ia = qp + a_off, alpha = db_alpha[ia + 12], beta = db_beta[qp + b_off + 12];
    int8_t tc[4];
    if (!alpha || !beta) return;
    for (int i = 0; i < 4; i++) tc[i] = (bS[i] ? db_tc0[ia + 12][bS[i] - 1] : -1) + chroma;
    f(p1, stride, alpha, beta, tc);
    if (chroma) f(p2, stride, alpha, beta, tc);
}
static void db_edge_intra(u8 *p1, u8 *p2, int stride, int qp, int chroma, int a_off, int b_off, x264hip_deblock_intra_t f)
{   /* deblock_edge_intra, :608-619 */
    int alpha = db_alpha[qp + a_off + 12], beta = db_beta[qp + b_off + 12];
    if (!alpha || !beta) return;
    f(p1, stride, alpha, beta);
    if (chroma) f(p2, stride, alpha, beta);
}

/* mb_type: 0 inter, 1 intra, 2 P_SKIP, 3 P_8x8 while X264_ANALYSE_PSUB8x8 is on (frame.c:645: only then no_sub8x8 = 0, every
 * 4-pixel edge segment of that macroblock compares vectors; otherwise only the 8x8 grid does and an odd segment copies its neighbour) */
void x264o_frame_deblock(u8 *py, u8 *pu, u8 *pv, int mb_w, int mb_h, int sy, int sc,
                         const u8 *mb_type, const u8 *qp, const u8 *nnz, const u8 *t8x8, const i16 *mv, const int8_t *ref,
                         int a_off, int b_off, int cqp_off)
{
    init();
    int qp_thresh = 15 - (a_off < b_off ? a_off : b_off) - (cqp_off > 0 ? cqp_off : 0);
    const u8 *cqt = db_chroma_qp + 12 + cqp_off;
    for (int mby = 0; mby < mb_h; mby++)
        for (int mbx = 0; mbx < mb_w; mbx++) {
            int mb = mby * mb_w + mbx, t8 = t8x8[mb], q = qp[mb], no_sub8x8 = mb_type[mb] != 3;
            int edge_end = (mb_type[mb] == 2 || q <= qp_thresh) ? 1 : 4;
            u8 *piy = py + 16 * mby * sy + 16 * mbx, *piu = pu + 8 * mby * sc + 8 * mbx, *piv = pv + 8 * mby * sc + 8 * mbx;
            for (int dir = 0; dir < 2; dir++) {
                /* edge 0 (shared with the left / top macroblock) is filtered whenever that
                 * neighbour exists; inner edges 1+t8, ... only below edge_end (:744-780) */
                int at_border = dir ? mby == 0 : mbx == 0;
                for (int edge = at_border ? 1 + t8 : 0; edge == 0 || edge < edge_end; edge = edge ? edge + t8 + 1 : t8 + 1) {
                    int mbn = edge ? mb : (dir ? mb - mb_w : mb - 1), qn = qp[mbn];
                    u8 bS[4] = {0, 0, 0, 0};
                    int intra_edge = edge == 0 && (mb_type[mb] == 1 || mb_type[mbn] == 1);
                    u8 *ly = dir ? piy + 4 * edge * sy : piy + 4 * edge;
                    u8 *lu = dir ? piu + 2 * edge * sc : piu + 2 * edge, *lv = dir ? piv + 2 * edge * sc : piv + 2 * edge;
                    int qpa = (q + qn + 1) >> 1, qpc = (cqt[q] + cqt[qn] + 1) >> 1;
                    if (intra_edge) {
                        db_edge_intra(ly, 0, sy, qpa, 0, a_off, b_off, dir ? dbf.deblock_v_luma_intra : dbf.deblock_h_luma_intra);
                        if (!(edge & 1))
                            db_edge_intra(lu, lv, sc, qpc, 1, a_off, b_off, dir ? dbf.deblock_v_chroma_intra : dbf.deblock_h_chroma_intra);
                    } else {
                        /* DEBLOCK_STRENGTH, :697-742 (P slice, 16x16 / 8x8 partitions: no_sub8x8 = 1) */
                        if (mb_type[mb] == 1 || mb_type[mbn] == 1) bS[0] = bS[1] = bS[2] = bS[3] = 3;
                        else
                            for (int i = 0; i < 4; i++) {
                                int x = dir == 0 ? edge : i, y = dir == 0 ? i : edge;
                                int xn = dir == 0 ? (x - 1) & 3 : x, yn = dir == 0 ? y : (y - 1) & 3;
                                if (nnz[mb * 26 + z_of(x, y)] || nnz[mbn * 26 + z_of(xn, yn)]) bS[i] = 2;
                                else if (!(edge & no_sub8x8)) {
                                    if ((i & no_sub8x8) && bS[i - 1] != 2) bS[i] = bS[i - 1];
                                    else {
                                        const i16 *mp = mv + (mb * 16 + x + 4 * y) * 2, *mq = mv + (mbn * 16 + xn + 4 * yn) * 2;
                                        int rp = ref[mb * 4 + (x >> 1) + (y >> 1) * 2], rq = ref[mbn * 4 + (xn >> 1) + (yn >> 1) * 2];
                                        if (rp != rq || abs(mp[0] - mq[0]) >= 4 || abs(mp[1] - mq[1]) >= 4) bS[i] = 1;
                                    }
                                }
                            }
                        if (bS[0] | bS[1] | bS[2] | bS[3]) {
                            db_edge(ly, 0, sy, bS, qpa, 0, a_off, b_off, dir ? dbf.deblock_v_luma : dbf.deblock_h_luma);
                            if (!(edge & 1))
                                db_edge(lu, lv, sc, bS, qpc, 1, a_off, b_off, dir ? dbf.deblock_v_chroma : dbf.deblock_h_chroma);
                        }
                    }
                }
            }
        }
}

#ifndef X264O_USE_REF
#include "look_oracle.c"      /* the lookahead's per-frame cost (x264_slicetype_frame_cost) */
#include "slice_oracle.c"     /* the per-macroblock sweep twin; shares the helpers above */
#endif
