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
static void init(void)
{
    if (ready) return;
    x264o_pixel_init(&pixf); x264o_dct_init(&dctf);
    x264o_zigzag_init(&zigf[0], 0); x264o_zigzag_init(&zigf[1], 1);
    x264o_quant_init(&quantf); x264o_mc_init(&mcf); x264o_deblock_init(&dbf);
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
