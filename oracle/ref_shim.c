/* ref_shim.c -- TEST INFRASTRUCTURE.  Compiled only into
 * oracle/_ref/libx264ref.so, against the reference's own headers where they
 * lie (-I$(REF)).  It contains no reference code: it only builds the x264_t
 * state the reference's x264_cqm_init() (R/common/set.c:68-168) expects and
 * hands the resulting quantiser tables back as flat arrays, so that
 * oracle/gen_golden.py can store them as golden vectors.
 */
#include "common/common.h"

static x264_t *g_h;
static const uint8_t flat16[64] = {
    16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,
    16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16 };

/* deadzones as x264_param_default sets them (R/common/common.c:129-130) */
int refshim_cqm_flat_init(void)
{
    int i;
    if (g_h)
        return 0;
    g_h = calloc(1, sizeof(x264_t));
    if (!g_h)
        return -1;
    g_h->pps = &g_h->pps_array[0];
    for (i = 0; i < 6; i++)
        g_h->pps->scaling_list[i] = flat16;
    g_h->param.analyse.i_luma_deadzone[0] = 21;
    g_h->param.analyse.i_luma_deadzone[1] = 11;
    g_h->param.analyse.b_transform_8x8 = 1;
    g_h->param.rc.i_qp_min = 0;
    return x264_cqm_init(g_h);
}
/* x264_cqm_init for a preset (0 flat, 1 jvt) into a fresh context that the getters below then read */
static x264_t *g_cq;
int refshim_cqm_init_preset(int preset)
{
    int i;
    g_cq = calloc(1, sizeof(x264_t));
    if (!g_cq) return -1;
    g_cq->pps = &g_cq->pps_array[0];
    for (i = 0; i < 6; i++) g_cq->pps->scaling_list[i] = preset ? x264_cqm_jvt[i] : flat16;
    g_cq->param.analyse.i_luma_deadzone[0] = 21; g_cq->param.analyse.i_luma_deadzone[1] = 11;
    g_cq->param.analyse.b_transform_8x8 = 1;
    g_cq->param.rc.i_qp_min = 51;                     /* do not refuse the low qps whose multipliers overflow 16 bits with jvt */
    return x264_cqm_init(g_cq);
}
const uint16_t *refshim_cq_quant4_mf(int cat, int qp)   { return g_cq->quant4_mf[cat][qp]; }
const uint16_t *refshim_cq_quant4_bias(int cat, int qp) { return g_cq->quant4_bias[cat][qp]; }
const uint16_t *refshim_cq_quant8_mf(int cat, int qp)   { return g_cq->quant8_mf[cat][qp]; }
const uint16_t *refshim_cq_quant8_bias(int cat, int qp) { return g_cq->quant8_bias[cat][qp]; }
const int *refshim_cq_dequant4_mf(int cat) { return &g_cq->dequant4_mf[cat][0][0][0]; }
const int *refshim_cq_dequant8_mf(int cat) { return &g_cq->dequant8_mf[cat][0][0][0]; }
/* cat: 0 intra-Y 1 inter-Y 2 intra-C 3 inter-C (4x4); 0 intra-Y 1 inter-Y (8x8) */
const uint16_t *refshim_quant4_mf(int cat, int qp)   { return g_h->quant4_mf[cat][qp]; }
const uint16_t *refshim_quant4_bias(int cat, int qp) { return g_h->quant4_bias[cat][qp]; }
const uint16_t *refshim_quant8_mf(int cat, int qp)   { return g_h->quant8_mf[cat][qp]; }
const uint16_t *refshim_quant8_bias(int cat, int qp) { return g_h->quant8_bias[cat][qp]; }
const int *refshim_dequant4_mf(int cat) { return &g_h->dequant4_mf[cat][0][0][0]; }
const int *refshim_dequant8_mf(int cat) { return &g_h->dequant8_mf[cat][0][0][0]; }

/* ---------------------------------------------------------------------------
 * The reference's own per-frame drivers run on caller-provided planes:
 * x264_frame_expand_border_mod16 / x264_frame_init_lowres /
 * x264_frame_expand_border / x264_frame_filter /
 * x264_frame_expand_border_filtered / x264_frame_deblock_row
 * (R/common/frame.c, R/common/mc.c).  The shim only builds the x264_t and
 * x264_frame_t those functions read; every pixel is produced by reference code.
 * Planes are given by the address of pixel (0,0) inside padded images laid out
 * as x264_frame_new lays them out (strides / padding computed by the caller with
 * the same formulas, R/common/frame.c:29-152).                                */
typedef struct {
    int width, height;            /* param.i_width / i_height */
    int stride_y, stride_c, stride_lowres;
    uint8_t *plane[3], *filtered[4], *lowres[4];
} refshim_frame;

static x264_t *mk_h(const refshim_frame *f, x264_frame_t *fr)
{
    x264_t *h = calloc(1, sizeof(x264_t));
    int i, j;
    h->sps = &h->sps_array[0];
    h->pps = &h->pps_array[0];
    h->param.i_width = f->width; h->param.i_height = f->height;
    h->sps->i_mb_width = (f->width + 15) / 16; h->sps->i_mb_height = (f->height + 15) / 16;
    h->mb.i_mb_stride = h->sps->i_mb_width;
    h->mb.i_mb_count = h->sps->i_mb_width * h->sps->i_mb_height;
    x264_mc_init(0, &h->mc);
    x264_deblock_init(0, &h->loopf);
    h->scratch_buffer = malloc((f->stride_y + 64) * sizeof(int16_t));
    memset(fr, 0, sizeof(*fr));
    fr->i_plane = 3;
    for (i = 0; i < 3; i++) {
        fr->plane[i] = f->plane[i];
        fr->i_stride[i] = i ? f->stride_c : f->stride_y;
        fr->i_width[i] = (16 * h->sps->i_mb_width) >> !!i;
        fr->i_lines[i] = (16 * h->sps->i_mb_height) >> !!i;
    }
    for (i = 0; i < 4; i++) { fr->filtered[i] = f->filtered[i]; fr->lowres[i] = f->lowres[i]; }
    fr->i_stride_lowres = f->stride_lowres;
    fr->i_width_lowres = fr->i_width[0] / 2;
    fr->i_lines_lowres = fr->i_lines[0] / 2;
    /* arrays x264_frame_init_lowres resets (R/common/mc.c:322-330) with i_bframe = 0 */
    for (i = 0; i < 2; i++)
        for (j = 0; j < 2; j++)
            fr->i_row_satds[i][j] = calloc(h->sps->i_mb_height + 1, sizeof(int));
    fr->lowres_mvs[0][0] = calloc(h->mb.i_mb_count + 1, 2 * sizeof(int16_t));
    h->fdec = fr;
    return h;
}
static void rm_h(x264_t *h, x264_frame_t *fr)
{
    int i, j;
    for (i = 0; i < 2; i++) for (j = 0; j < 2; j++) free(fr->i_row_satds[i][j]);
    free(fr->lowres_mvs[0][0]);
    free(h->scratch_buffer);
    free(h);
}

/* what x264_encoder_encode does to an input picture after the copy (encoder.c:1411-1418) */
void refshim_source_prepare(const refshim_frame *f, int do_lowres)
{
    x264_frame_t fr;
    x264_t *h = mk_h(f, &fr);
    x264_frame_expand_border_mod16(h, &fr);
    if (do_lowres)
        x264_frame_init_lowres(h, &fr);
    rm_h(h, &fr);
}

/* what x264_fdec_filter_row does to a reconstructed frame, row by row in the
 * reference's own order (encoder.c:983-1024): [deblock] -> expand border ->
 * half-pel filter -> expand filtered border.  mb_type uses this repository's
 * codes (0 inter 16x16-style, 1 intra, 2 P_SKIP, 3 inter with four 8x8
 * vectors); nnz is [mb][26] in block z-order, mv [mb][16][2] raster per MB,
 * ref [mb][4]; they are converted here to the reference's frame-level layouts
 * (x264_macroblock_cache_save, R/common/macroblock.c:1208-1372).              */
void refshim_fdec_filter(const refshim_frame *f, int do_deblock, const uint8_t *mb_type, const uint8_t *qp,
                         const uint8_t *nnz26, const uint8_t *t8x8, const int16_t *mv16, const int8_t *ref4,
                         int a_off, int b_off, int cqp_off, int do_hpel)
{
    static const uint8_t z2r[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};  /* block idx -> x+4y (block_idx_xy_1d) */
    x264_frame_t fr;
    x264_t *h = mk_h(f, &fr);
    int mb_w = h->sps->i_mb_width, mb_h = h->sps->i_mb_height, n = mb_w * mb_h, mb, i, mb_y;
    if (do_deblock) {
        h->mb.type = calloc(n, sizeof(int8_t));
        h->mb.qp = calloc(n, sizeof(int8_t));
        h->mb.non_zero_count = calloc(n, 24);
        h->mb.mb_transform_size = calloc(n, sizeof(int8_t));
        h->mb.mv[0] = calloc(n * 16, 2 * sizeof(int16_t));
        h->mb.ref[0] = calloc(n * 4, sizeof(int8_t));
        h->sh.i_alpha_c0_offset = a_off; h->sh.i_beta_offset = b_off;
        h->sh.i_type = SLICE_TYPE_P;
        h->chroma_qp_table = i_chroma_qp_table + 12 + cqp_off;
        h->param.analyse.i_chroma_qp_offset = cqp_off;
        h->pps->b_cabac = 1;
        h->pps->b_transform_8x8_mode = 1;
        for (mb = 0; mb < n; mb++) {
            int mx = mb % mb_w, my = mb / mb_w;
            h->mb.type[mb] = mb_type[mb] == 1 ? I_16x16 : mb_type[mb] == 2 ? P_SKIP : mb_type[mb] == 3 ? P_8x8 : P_L0;
            h->mb.qp[mb] = qp[mb];
            h->mb.mb_transform_size[mb] = t8x8[mb];
            for (i = 0; i < 16; i++) h->mb.non_zero_count[mb][z2r[i]] = nnz26[mb * 26 + i];
            for (i = 0; i < 8; i++) h->mb.non_zero_count[mb][16 + i] = nnz26[mb * 26 + 16 + i];
            for (i = 0; i < 16; i++) {
                int x = i & 3, y = i >> 2, o = 4 * (4 * mb_w) * my + 4 * mx + x + y * (4 * mb_w);
                h->mb.mv[0][o][0] = mv16[(mb * 16 + i) * 2]; h->mb.mv[0][o][1] = mv16[(mb * 16 + i) * 2 + 1];
            }
            for (i = 0; i < 4; i++)
                h->mb.ref[0][2 * (2 * mb_w) * my + 2 * mx + (i & 1) + (i >> 1) * (2 * mb_w)] = ref4[mb * 4 + i];
        }
    }
    for (mb_y = 1; mb_y <= mb_h; mb_y++) {
        int b_end = mb_y == mb_h, min_y = mb_y - 1;
        if (do_deblock)
            x264_frame_deblock_row(h, min_y);
        x264_frame_expand_border(h, &fr, min_y, b_end);
        if (do_hpel) {
            x264_frame_filter(h, &fr, min_y, b_end);
            x264_frame_expand_border_filtered(h, &fr, min_y, b_end);
        }
    }
    if (do_deblock) {
        free(h->mb.type); free(h->mb.qp); free(h->mb.non_zero_count); free(h->mb.mb_transform_size);
        free(h->mb.mv[0]); free(h->mb.ref[0]);
    }
    rm_h(h, &fr);
}

/* ---------------------------------------------------------------------------
 * The reference's own x264_macroblock_encode / x264_macroblock_probe_skip
 * (R/encoder/macroblock.c:475-883) run macroblock by macroblock over a frame.
 * Each macroblock is presented as P_8x8 with four D_L0_4x4 sub-partitions, so
 * x264_mb_mc motion-compensates every 4x4 block with its own vector and the
 * reference index of its 8x8 (R/common/macroblock.c:489-546): the general P
 * case.  The shim fills the fields of x264_t those functions read (what
 * x264_macroblock_cache_load would have filled) and copies the results out.   */
#define SENTINEL 0x7f7f
typedef struct {
    uint8_t *y[4], *u, *v;         /* full, H, V, HV planes + chroma of one reference (pixel 0,0) */
} refshim_ref;

static x264_t *enc_h(int width, int height, int stride_y, int stride_c, int qp, int qpc, int t8, int interlaced)
{
    x264_t *h;
    refshim_cqm_flat_init();
    h = g_h;
    h->sps = &h->sps_array[0];
    h->param.i_width = width; h->param.i_height = height;
    h->sps->i_mb_width = (width + 15) / 16; h->sps->i_mb_height = (height + 15) / 16;
    h->mb.i_mb_stride = h->sps->i_mb_width;
    h->mb.i_mb_count = h->sps->i_mb_width * h->sps->i_mb_height;
    x264_pixel_init(0, &h->pixf); x264_dct_init(0, &h->dctf); x264_zigzag_init(0, &h->zigzagf, interlaced);
    x264_quant_init(h, 0, &h->quantf); x264_mc_init(0, &h->mc);
    x264_predict_16x16_init(0, h->predict_16x16); x264_predict_8x8c_init(0, h->predict_8x8c);
    x264_predict_8x8_init(0, h->predict_8x8, &h->predict_8x8_filter); x264_predict_4x4_init(0, h->predict_4x4);
    h->param.analyse.b_dct_decimate = 1;
    h->param.b_cabac = 1;
    h->sh.i_type = SLICE_TYPE_P;
    h->mb.b_trellis = 0; h->mb.b_noise_reduction = 0; h->mb.b_lossless = 0; h->mb.b_interlaced = 0;
    h->mb.i_qp = qp; h->mb.i_chroma_qp = qpc; h->mb.b_transform_8x8 = t8;
    h->mb.pic.i_stride[0] = stride_y; h->mb.pic.i_stride[1] = h->mb.pic.i_stride[2] = stride_c;
    h->mb.pic.p_fenc[0] = h->mb.pic.fenc_buf;
    h->mb.pic.p_fenc[1] = h->mb.pic.fenc_buf + 16 * FENC_STRIDE;
    h->mb.pic.p_fenc[2] = h->mb.pic.fenc_buf + 16 * FENC_STRIDE + 8;
    h->mb.pic.p_fdec[0] = h->mb.pic.fdec_buf + 2 * FDEC_STRIDE;
    h->mb.pic.p_fdec[1] = h->mb.pic.fdec_buf + 19 * FDEC_STRIDE;
    h->mb.pic.p_fdec[2] = h->mb.pic.fdec_buf + 19 * FDEC_STRIDE + 16;
    if (!h->mb.cbp) h->mb.cbp = calloc(1 << 20, sizeof(int16_t));
    return h;
}
static void enc_load_mb(x264_t *h, int mb, uint8_t *fy, uint8_t *fu, uint8_t *fv, const refshim_ref *refs, int n_refs)
{
    int mbx = mb % h->sps->i_mb_width, mby = mb / h->sps->i_mb_width, y, r, k;
    int sy = h->mb.pic.i_stride[0], sc = h->mb.pic.i_stride[1];
    int oy = 16 * mby * sy + 16 * mbx, oc = 8 * mby * sc + 8 * mbx;
    h->mb.i_mb_x = mbx; h->mb.i_mb_y = mby; h->mb.i_mb_xy = mb;
    h->mb.mv_min[0] = 4 * (-16 * mbx - 24); h->mb.mv_max[0] = 4 * (16 * (h->sps->i_mb_width - mbx - 1) + 24);
    h->mb.mv_min[1] = 4 * (-16 * mby - 24); h->mb.mv_max[1] = 4 * (16 * (h->sps->i_mb_height - mby - 1) + 24);
    for (y = 0; y < 16; y++) memcpy(h->mb.pic.p_fenc[0] + y * FENC_STRIDE, fy + oy + y * sy, 16);
    for (y = 0; y < 8; y++) {
        memcpy(h->mb.pic.p_fenc[1] + y * FENC_STRIDE, fu + oc + y * sc, 8);
        memcpy(h->mb.pic.p_fenc[2] + y * FENC_STRIDE, fv + oc + y * sc, 8);
    }
    for (r = 0; r < n_refs; r++) {
        for (k = 0; k < 4; k++) h->mb.pic.p_fref[0][r][k] = refs[r].y[k] + oy;
        h->mb.pic.p_fref[0][r][4] = refs[r].u + oc;
        h->mb.pic.p_fref[0][r][5] = refs[r].v + oc;
    }
}

void refshim_inter_encode_frame(uint8_t *fy, uint8_t *fu, uint8_t *fv, const refshim_ref *refs, int n_refs,
                                uint8_t *dy, uint8_t *du, uint8_t *dv, int width, int height, int stride_y, int stride_c,
                                int qp, int qpc, int t8, int interlaced, const int16_t *mv16, const int8_t *ref4,
                                int16_t *levels_y, int16_t *levels_c, int16_t *dc_c, int32_t *cbp_out, uint8_t *nnz_out)
{
    x264_t *h = enc_h(width, height, stride_y, stride_c, qp, qpc, t8, interlaced);
    int n = h->mb.i_mb_count, mb, i, ch, y;
    for (mb = 0; mb < n; mb++) {
        int mbx = mb % h->sps->i_mb_width, mby = mb / h->sps->i_mb_width;
        int oy = 16 * mby * stride_y + 16 * mbx, oc = 8 * mby * stride_c + 8 * mbx;
        int16_t *ly = levels_y + mb * 256, *lc = levels_c + mb * 128, *ldc = dc_c + mb * 8, *p;
        enc_load_mb(h, mb, fy, fu, fv, refs, n_refs);
        h->mb.i_type = P_8x8; h->mb.i_partition = D_8x8; h->mb.b_skip_mc = 0;
        h->mb.b_transform_8x8 = t8;
        for (i = 0; i < 4; i++) h->mb.i_sub_partition[i] = D_L0_4x4;
        for (i = 0; i < 16; i++) {
            int x = i & 3, yy = i >> 2, c8 = x264_scan8[0] + x + 8 * yy;
            h->mb.cache.mv[0][c8][0] = mv16[(mb * 16 + i) * 2]; h->mb.cache.mv[0][c8][1] = mv16[(mb * 16 + i) * 2 + 1];
            h->mb.cache.ref[0][c8] = ref4 ? ref4[mb * 4 + (yy >> 1) * 2 + (x >> 1)] : 0;
        }
        for (p = &h->dct.luma4x4[0][0], i = 0; i < 24 * 16; i++) p[i] = SENTINEL;
        for (p = &h->dct.luma8x8[0][0], i = 0; i < 4 * 64; i++) p[i] = SENTINEL;
        for (p = &h->dct.chroma_dc[0][0], i = 0; i < 8; i++) p[i] = SENTINEL;
        x264_macroblock_encode(h);
        /* scanned levels; blocks the encoder never wrote (they quantised to nothing) read as zero */
        if (t8)
            for (i = 0; i < 4; i++) {
                int w = h->dct.luma8x8[i][0] != SENTINEL || h->dct.luma8x8[i][63] != SENTINEL;
                for (y = 0; y < 64; y++) ly[64 * i + y] = w ? h->dct.luma8x8[i][y] : 0;
            }
        else
            for (i = 0; i < 16; i++) {
                int w = h->dct.luma4x4[i][0] != SENTINEL || h->dct.luma4x4[i][15] != SENTINEL;
                for (y = 0; y < 16; y++) ly[16 * i + y] = w ? h->dct.luma4x4[i][y] : 0;
            }
        for (i = 0; i < 8; i++) {
            int w = h->dct.luma4x4[16 + i][0] != SENTINEL || h->dct.luma4x4[16 + i][15] != SENTINEL;
            for (y = 0; y < 16; y++) lc[16 * i + y] = w ? h->dct.luma4x4[16 + i][y] : 0;
        }
        for (ch = 0; ch < 2; ch++)
            for (i = 0; i < 4; i++) ldc[4 * ch + i] = h->dct.chroma_dc[ch][i] == SENTINEL ? 0 : h->dct.chroma_dc[ch][i];
        for (i = 0; i < 24; i++) nnz_out[mb * 26 + i] = h->mb.cache.non_zero_count[x264_scan8[i]];
        nnz_out[mb * 26 + 24] = h->mb.cache.non_zero_count[x264_scan8[25]];
        nnz_out[mb * 26 + 25] = h->mb.cache.non_zero_count[x264_scan8[26]];
        cbp_out[mb] = h->mb.i_cbp_luma | (h->mb.i_cbp_chroma << 4);
        for (y = 0; y < 16; y++) memcpy(dy + oy + y * stride_y, h->mb.pic.p_fdec[0] + y * FDEC_STRIDE, 16);
        for (y = 0; y < 8; y++) {
            memcpy(du + oc + y * stride_c, h->mb.pic.p_fdec[1] + y * FDEC_STRIDE, 8);
            memcpy(dv + oc + y * stride_c, h->mb.pic.p_fdec[2] + y * FDEC_STRIDE, 8);
        }
    }
}

void refshim_probe_skip_frame(uint8_t *fy, uint8_t *fu, uint8_t *fv, const refshim_ref *ref, int width, int height,
                              int stride_y, int stride_c, int qp, int qpc, int interlaced, const int16_t *pskip_mv,
                              uint8_t *skip_out)
{
    x264_t *h = enc_h(width, height, stride_y, stride_c, qp, qpc, 0, interlaced);
    int n = h->mb.i_mb_count, mb;
    for (mb = 0; mb < n; mb++) {
        enc_load_mb(h, mb, fy, fu, fv, ref, 1);
        h->mb.cache.pskip_mv[0] = pskip_mv[2 * mb]; h->mb.cache.pskip_mv[1] = pskip_mv[2 * mb + 1];
        skip_out[mb] = (uint8_t)x264_macroblock_probe_skip(h, 0);
    }
}
/* ---------------------------------------------------------------------------
 * The reference's own x264_me_search_ref (R/encoder/me.c:156-631, with its
 * refine_subpel) for the 16x16 block of every macroblock and every reference,
 * in the loop of x264_mb_analyse_inter_p16x16 (R/encoder/analyse.c:1077-1127):
 * the half-pel threshold is carried from reference to reference and the best
 * reference is the first minimum of cost + ref cost.  Predictors (mvp, mvc)
 * are inputs.  mbcmp / fpelcmp are selected as mbcmp_init does
 * (R/encoder/encoder.c:608-618; that function is static there).              */
#include <limits.h>
#include "encoder/me.h"
void refshim_me_search16_frame(uint8_t *fy, uint8_t *fu, uint8_t *fv, const refshim_ref *refs, int n_refs, int width, int height,
                               int stride_y, int stride_c, int method, int me_range, int subme, int chroma_me, int mv_range,
                               int16_t *cost_mv_center, const int16_t *mvp, const int16_t *mvc, const uint8_t *n_mvc,
                               const int32_t *ref_cost, int16_t *out_mv, int32_t *out_cost, int32_t *best)
{
    x264_t *h = enc_h(width, height, stride_y, stride_c, 26, 26, 0, 0);
    int n = h->mb.i_mb_count, mb, r, satd = subme > 1;
    int mb_w = h->sps->i_mb_width, mb_h = h->sps->i_mb_height;
    memcpy(h->pixf.mbcmp, satd ? h->pixf.satd : h->pixf.sad_aligned, sizeof(h->pixf.mbcmp));
    memcpy(h->pixf.mbcmp_unaligned, satd ? h->pixf.satd : h->pixf.sad, sizeof(h->pixf.mbcmp_unaligned));
    memcpy(h->pixf.fpelcmp, h->pixf.sad, sizeof(h->pixf.fpelcmp));
    memcpy(h->pixf.fpelcmp_x3, h->pixf.sad_x3, sizeof(h->pixf.fpelcmp_x3));
    memcpy(h->pixf.fpelcmp_x4, h->pixf.sad_x4, sizeof(h->pixf.fpelcmp_x4));
    h->mb.i_me_method = method ? X264_ME_HEX : X264_ME_DIA;
    h->mb.i_subpel_refine = subme;
    h->mb.b_chroma_me = chroma_me;
    h->param.analyse.i_me_range = me_range;
    for (mb = 0; mb < n; mb++) {
        int mbx = mb % mb_w, mby = mb / mb_w;
        int fr = 4 * mv_range, lo = 4 * (-512 + 8) > -fr ? 4 * (-512 + 8) : -fr;
        int thresh = INT_MAX, bestc = INT_MAX;
        enc_load_mb(h, mb, fy, fu, fv, refs, n_refs);
        /* x264_mb_analyse_init, R/encoder/analyse.c:258-298 */
        h->mb.mv_min_spel[0] = x264_clip3(h->mb.mv_min[0], -fr, fr - 1);
        h->mb.mv_max_spel[0] = x264_clip3(h->mb.mv_max[0], -fr, fr - 1);
        h->mb.mv_min_spel[1] = x264_clip3(h->mb.mv_min[1], lo, fr);
        h->mb.mv_max_spel[1] = x264_clip3(h->mb.mv_max[1], -fr, fr - 1);
        h->mb.mv_min_fpel[0] = (h->mb.mv_min_spel[0] >> 2) + 5; h->mb.mv_max_fpel[0] = (h->mb.mv_max_spel[0] >> 2) - 5;
        h->mb.mv_min_fpel[1] = (h->mb.mv_min_spel[1] >> 2) + 5; h->mb.mv_max_fpel[1] = (h->mb.mv_max_spel[1] >> 2) - 5;
        for (r = 0; r < n_refs; r++) {
            x264_me_t m;
            DECLARE_ALIGNED_4(int16_t cand[8][2]);
            int k, nc = n_mvc[mb * n_refs + r];
            memset(&m, 0, sizeof(m));
            m.i_pixel = PIXEL_16x16;
            m.p_cost_mv = cost_mv_center;
            m.i_ref_cost = ref_cost[r]; m.i_ref = r;
            for (k = 0; k < 3; k++) m.p_fenc[k] = h->mb.pic.p_fenc[k];
            for (k = 0; k < 6; k++) m.p_fref[k] = h->mb.pic.p_fref[0][r][k];
            m.i_stride[0] = stride_y; m.i_stride[1] = stride_c;
            m.mvp[0] = mvp[(mb * n_refs + r) * 2]; m.mvp[1] = mvp[(mb * n_refs + r) * 2 + 1];
            for (k = 0; k < 8; k++) { cand[k][0] = mvc[((mb * n_refs + r) * 8 + k) * 2]; cand[k][1] = mvc[((mb * n_refs + r) * 8 + k) * 2 + 1]; }
            thresh -= ref_cost[r];
            x264_me_search_ref(h, &m, cand, nc, n_refs > 1 ? &thresh : NULL);
            m.cost += ref_cost[r];
            thresh += ref_cost[r];
            out_mv[(mb * n_refs + r) * 2] = m.mv[0]; out_mv[(mb * n_refs + r) * 2 + 1] = m.mv[1];
            out_cost[mb * n_refs + r] = m.cost;
            if (m.cost < bestc) { bestc = m.cost; best[4 * mb] = r; best[4 * mb + 1] = m.mv[0]; best[4 * mb + 2] = m.mv[1]; best[4 * mb + 3] = m.cost; }
        }
    }
}

extern const int x264_lambda2_tab[52];   /* R/encoder/analyse.c:151-159 */
int refshim_lambda2(int qp) { return x264_lambda2_tab[qp]; }

/* round 2: the reference's trellis quantiser (R/encoder/rdo.c:632-660) on one block, against context states the caller supplies.
 * kind 0: x264_quant_4x4_trellis, 1: x264_quant_8x8_trellis, 2: x264_quant_dc_trellis.  refshim_cqm_init_preset first.        */
int x264_quant_dc_trellis(x264_t *h, int16_t *dct, int i_quant_cat, int i_qp, int i_ctxBlockCat, int b_intra);
int x264_quant_4x4_trellis(x264_t *h, int16_t dct[4][4], int i_quant_cat, int i_qp, int i_ctxBlockCat, int b_intra, int idx);
int x264_quant_8x8_trellis(x264_t *h, int16_t dct[8][8], int i_quant_cat, int i_qp, int b_intra, int idx);
void x264_rdo_init(void);
int refshim_trellis(int16_t *dct, int kind, int qcat, int ctxcat, int qp, int b_intra, const uint8_t *state)
{
    static int init;
    if (!init) { x264_rdo_init(); x264_dct_init_weights(); init = 1; }
    memcpy(g_cq->cabac.state, state, 460);
    if (kind == 0) return x264_quant_4x4_trellis(g_cq, (int16_t (*)[4])dct, qcat, qp, ctxcat, b_intra, 0);
    if (kind == 1) return x264_quant_8x8_trellis(g_cq, (int16_t (*)[8])dct, qcat, qp, b_intra, 0);
    return x264_quant_dc_trellis(g_cq, dct, qcat, qp, ctxcat, b_intra);
}
