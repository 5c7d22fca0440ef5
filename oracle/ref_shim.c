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
