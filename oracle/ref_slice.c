/* ref_slice.c -- TEST INFRASTRUCTURE.  Compiled only into oracle/_ref/libx264ref.so against the
 * reference's headers where they lie.  It contains no reference code.
 *
 * Runs the reference's OWN per-macroblock hot loop over a chain of frames (I then P, every frame
 * kept as reference): x264_macroblock_cache_load -> x264_macroblock_analyse ->
 * x264_macroblock_encode -> x264_macroblock_cache_save, with the row-delayed
 * x264_frame_deblock_row / x264_frame_expand_border / x264_frame_filter, i.e. what
 * x264_slice_write + x264_fdec_filter_row do (R/encoder/encoder.c:983-1056,1141-1291; both are
 * static in encoder.c, which cannot be built here, so their dozen lines of sequencing are restated
 * below).  refslice_encode_chain (round 1) does not call the entropy writer: without RD (subme < 6)
 * nothing it computes feeds back into analysis.  refslice_encode_chain2 (round 2) is the same loop with
 * the writer in it, exactly as x264_slice_write sequences it (R/encoder/encoder.c:1155-1165,1192-1219,
 * 1269-1280): x264_cabac_context_init / x264_cabac_encode_init per slice, the terminal bit, the skip
 * flag and x264_macroblock_write_cabac (or the skip run and x264_macroblock_write_cavlc) per
 * macroblock, x264_cabac_encode_flush at the end -- so h->cabac evolves as in the encoder and the RD
 * levels (subme >= 6), trellis, psy-rd and adaptive quantisation (x264_adaptive_quant_frame) see the
 * state they see there.  It also returns each slice's payload bytes (slice_data(), i.e. everything
 * after the slice header).  Frames come from x264_frame_new, tables from the x264_*_init functions,
 * the QP from x264_ratecontrol_new/start (CQP).  Every decision, every pixel and every payload byte is
 * produced by reference code; this file only sequences calls and copies results out.               */
#include "common/common.h"
#include "encoder/ratecontrol.h"
#include "encoder/analyse.h"
#include "encoder/macroblock.h"

typedef struct {
    int width, height, n_frames, qp;
    int me_method, me_range, subme, n_refs;
    int inter, intra;                        /* X264_ANALYSE_* */
    int transform8x8, fast_pskip, dct_decimate, chroma_me, cabac, mixed_refs;
    int deblock, alpha_c0, beta, chroma_qp_offset, keyint;
    int noise_reduction;                     /* param.analyse.i_noise_reduction */
    int mv_range;                            /* param.analyse.i_mv_range (0 = 512) */
    int cqm_preset;                          /* param.i_cqm_preset: 0 X264_CQM_FLAT, 1 X264_CQM_JVT */
} refslice_params;

/* round 2: what refslice_encode_chain2 takes on top of refslice_params */
typedef struct {
    int trellis;                             /* param.analyse.i_trellis 0..2 */
    float psy_rd, psy_trellis;               /* param.analyse.f_psy_rd / f_psy_trellis (x264_validate_parameters' effects are applied below) */
    int aq_mode; float aq_strength;          /* param.rc.i_aq_mode / f_aq_strength */
    int write;                               /* 1: call the entropy writer after every macroblock (required for subme >= 6) */
    int payload_cap;                         /* bytes per frame available in refslice_out2.payload */
    int cabac_init_idc;                      /* param.i_cabac_init_idc */
    /* B slices: a fixed pattern of `bframes` non-reference B frames between anchors (what x264_slicetype_decide produces with
     * --b-adapt 0 and no --b-pyramid); the clip stays in display order, the chain is coded in coding order */
    int bframes;                             /* param.i_bframe */
    int weightb;                             /* param.analyse.b_weighted_bipred */
    int direct_pred;                         /* param.analyse.i_direct_mv_pred: 1 spatial, 2 temporal, 3 auto (the reference only: the twin and the product refuse it) */
    /* the lookahead's motion vectors (fenc->lowres_mvs, h->frames.b_have_lowres): x264_mb_predict_mv_ref16x16 offers twice the vector of
     * the macroblock's half-resolution block as a candidate of every 16x16 search on reference 0 (R/common/macroblock.c:393-398).  The
     * caller supplies them, [frame in coding order][list][n_mb][2] int16; a frame / list whose first component is 0x7fff has none
     * (x264_frame_init_lowres' marker, R/common/mc.c:330).  NULL: b_have_lowres = 0 as before */
    const int16_t *lowres_mv;
    /* round 3: the real lookahead and rate control in front of the loop (refslice_encode_stream): x264_slicetype_decide (b-adapt, pre-scenecut)
     * places the frame types, x264_ratecontrol_start gives every frame its QP (CRF), the lookahead's own vectors feed the 16x16 searches */
    int b_adapt;                             /* param.i_bframe_adaptive: 0 none, 1 fast, 2 trellis */
    int pre_scenecut, scenecut_threshold;    /* param.b_pre_scenecut, i_scenecut_threshold (-1: off) */
    int keyint_min;                          /* param.i_keyint_min (0: keyint_max / 10 ... as x264_validate_parameters leaves the default 25) */
    float crf;                               /* param.rc.f_rf_constant; < 0: constant QP */
    int bframe_bias;                         /* param.i_bframe_bias */
} refslice_ext;

typedef struct {
    uint8_t *payload;                        /* [F][payload_cap]: slice_data() of every frame */
    int32_t *payload_len;                    /* [F] */
    int32_t *mb_bits;                        /* [F][n]: bits written once this macroblock is out (x264_cabac_pos / bs_pos) */
    float *qp_offset;                        /* [F][n]: fenc->f_qp_offset (0 without AQ) */
    int16_t *mv1;                            /* [F][n][16][2]: list 1 (B slices) */
    int8_t *ref1;                            /* [F][n][4] */
    int32_t *frame_info2;                    /* [F][4]: display index, i_ref1, kept as reference, 0 */
    /* the lookahead's results per coded frame (refslice_encode_stream; NULL otherwise) */
    float *rc_info;                          /* [F][4]: rc->f_qpm (the frame's QP before rounding), fdec->f_qp_avg_rc, fdec->i_satd (x264_rc_analyse_slice), 0 */
    int16_t *look_mv;                        /* [F][2][n][2]: the lowres vectors offered to the 16x16 searches of this frame (list 0 / 1 towards reference 0; 0x7fff first = none) */
    int32_t *look_cost;                      /* [F][8]: fenc->i_cost_est[b - p0][p1 - b] of the frame as coded, i_cost_est_aq, i_intra_mbs[b - p0], i_cost_est[0][0], p0, p1, b, 0 */
} refslice_out2;

typedef struct {
    int8_t *mb_type, *partition, *sub_partition;     /* [F][n], [F][n], [F][n][4] */
    int16_t *mv;                                      /* [F][n][16][2] */
    int8_t *ref;                                      /* [F][n][4] */
    int16_t *mvr;                                     /* [F][n_refs][n][2] */
    uint8_t *nnz;                                     /* [F][n][27] in x264_scan8 index order */
    int8_t *i4mode, *i16mode, *chroma_mode, *qp;      /* [F][n][16], [F][n] ... */
    int16_t *cbp;                                     /* [F][n] */
    int8_t *t8;                                       /* [F][n] */
    int16_t *luma, *luma_dc, *chroma_dc, *chroma_ac;  /* [F][n][256], [16], [8], [128] */
    uint8_t *rec_y, *rec_u, *rec_v;                   /* [F][16 mb_h][16 mb_w] ... before deblocking */
    uint8_t *fin_y, *fin_u, *fin_v;                   /* after x264_fdec_filter_row */
    int32_t *frame_info;                              /* [F][4] slice type, qp, i_ref0, poc */
    int64_t *stat;                                    /* [F][4] i_intra_cost, i_inter_cost, i_mbs_analysed, 0 */
} refslice_out;

static void sel_cmp(x264_t *h)        /* what mbcmp_init selects (R/encoder/encoder.c:608-618) */
{
    int satd = !h->mb.b_lossless && h->param.analyse.i_subpel_refine > 1;
    memcpy(h->pixf.mbcmp, satd ? h->pixf.satd : h->pixf.sad_aligned, sizeof(h->pixf.mbcmp));
    memcpy(h->pixf.mbcmp_unaligned, satd ? h->pixf.satd : h->pixf.sad, sizeof(h->pixf.mbcmp_unaligned));
    h->pixf.intra_mbcmp_x3_16x16 = satd ? h->pixf.intra_satd_x3_16x16 : h->pixf.intra_sad_x3_16x16;
    memcpy(h->pixf.fpelcmp, h->pixf.sad, sizeof(h->pixf.fpelcmp));
    memcpy(h->pixf.fpelcmp_x3, h->pixf.sad_x3, sizeof(h->pixf.fpelcmp_x3));
    memcpy(h->pixf.fpelcmp_x4, h->pixf.sad_x4, sizeof(h->pixf.fpelcmp_x4));
}

static void filter_row(x264_t *h, int mb_y)        /* x264_fdec_filter_row's sequencing, no mbaff, one thread */
{
    int b_end = mb_y == h->sps->i_mb_height, min_y = mb_y - 1, i;
    if (min_y < 0)
        return;
    if (!b_end)
        for (i = 0; i < 3; i++)
            memcpy(h->mb.intra_border_backup[0][i], h->fdec->plane[i] + ((mb_y * 16 >> !!i) - 1) * h->fdec->i_stride[i],
                   h->sps->i_mb_width * 16 >> !!i);
    if (!h->fdec->b_kept_as_ref)                     /* a disposable B frame is neither filtered nor interpolated (encoder.c:986-991,1016) */
        return;
    if (!h->sh.i_disable_deblocking_filter_idc)
        x264_frame_deblock_row(h, min_y);
    x264_frame_expand_border(h, h->fdec, min_y, b_end);
    if (h->param.analyse.i_subpel_refine) {
        x264_frame_filter(h, h->fdec, min_y, b_end);
        x264_frame_expand_border_filtered(h, h->fdec, min_y, b_end);
    }
}

static const uint8_t flat16[64] = {
    16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,
    16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16 };

#ifdef REFSLICE_TRACE
static int refslice_trace_frame, refslice_trace_mb;
static x264_t *refslice_trace_h;
#endif
typedef struct {
    x264_t *h;
    uint8_t *bsbuf;
    int b_write, mb_w, mb_h, n, cw, ch, stream;
    int defer_end, pending_len;              /* run_stream without --pre-scenecut: x264_encoder_frame_end's part waits for the post-encode scene cut's verdict */
} rctx;

/* the encoder as x264_encoder_open leaves it for this path (R/encoder/encoder.c:628-760) */
static int setup_encoder(rctx *c, const refslice_params *p, const refslice_ext *e, int stream)
{
    x264_t *h = calloc(1, sizeof(x264_t));
    uint8_t *bsbuf = NULL;
    const int b_write = e && e->write;
    int i, k, mb_w, mb_h, n;
    c->h = h; c->stream = stream;

    x264_param_default(&h->param);
    h->param.i_width = p->width; h->param.i_height = p->height;
    h->param.i_threads = 1; h->param.b_cabac = p->cabac; h->param.i_frame_reference = p->n_refs; h->param.i_bframe = 0;
    h->param.i_log_level = X264_LOG_NONE;
    h->param.analyse.inter = p->inter; h->param.analyse.intra = p->intra;
    h->param.analyse.i_me_method = p->me_method; h->param.analyse.i_me_range = p->me_range;
    h->param.analyse.i_mv_range = p->mv_range > 0 ? p->mv_range : 512; h->param.analyse.i_subpel_refine = p->subme;
    h->param.analyse.b_chroma_me = p->chroma_me; h->param.analyse.b_mixed_references = p->mixed_refs;
    h->param.analyse.b_fast_pskip = p->fast_pskip; h->param.analyse.b_dct_decimate = p->dct_decimate;
    h->param.analyse.b_transform_8x8 = p->transform8x8; h->param.analyse.i_trellis = 0;
    h->param.analyse.i_noise_reduction = p->noise_reduction; h->param.analyse.f_psy_rd = 0; h->param.analyse.f_psy_trellis = 0;
    h->param.analyse.i_chroma_qp_offset = p->chroma_qp_offset;
    h->param.rc.i_rc_method = X264_RC_CQP; h->param.rc.i_qp_constant = p->qp; h->param.rc.i_aq_mode = 0;
    h->param.i_keyint_max = p->keyint > 0 ? p->keyint : 1 << 30;
    if (stream) {                                        /* x264_validate_parameters, R/encoder/encoder.c:455-470 */
        h->param.i_bframe_adaptive = x264_clip3(e->b_adapt, X264_B_ADAPT_NONE, X264_B_ADAPT_TRELLIS);
        h->param.i_bframe_bias = x264_clip3(e->bframe_bias, -90, 100);
        h->param.b_pre_scenecut = e->pre_scenecut; h->param.i_scenecut_threshold = e->scenecut_threshold;
        h->param.i_keyint_min = e->keyint_min > 0 ? e->keyint_min : h->param.i_keyint_max / 10;
        h->param.i_keyint_min = x264_clip3(h->param.i_keyint_min, 1, h->param.i_keyint_max / 2 + 1);
        if (e->crf >= 0) { h->param.rc.i_rc_method = X264_RC_CRF; h->param.rc.f_rf_constant = e->crf; }
    }
    if (e) {                                             /* x264_validate_parameters, R/encoder/encoder.c:493-522 */
        h->param.analyse.i_trellis = p->cabac ? x264_clip3(e->trellis, 0, 2) : 0;
        h->param.analyse.f_psy_rd = p->subme < 6 ? 0 : x264_clip3f(e->psy_rd, 0, 10);
        h->param.analyse.f_psy_trellis = h->param.analyse.i_trellis ? x264_clip3f(e->psy_trellis, 0, 10) : 0;
        h->mb.i_psy_rd = FIX8(h->param.analyse.f_psy_rd);
        if (h->mb.i_psy_rd) h->param.analyse.i_chroma_qp_offset -= h->param.analyse.f_psy_rd < 0.25 ? 1 : 2;
        h->mb.i_psy_trellis = FIX8(h->param.analyse.f_psy_trellis / 4);
        if (h->mb.i_psy_trellis) h->param.analyse.i_chroma_qp_offset -= h->param.analyse.f_psy_trellis < 0.25 ? 1 : 2;
        h->param.analyse.i_chroma_qp_offset = x264_clip3(h->param.analyse.i_chroma_qp_offset, -12, 12);
        h->param.rc.f_aq_strength = x264_clip3f(e->aq_strength, 0, 3);
        h->param.rc.i_aq_mode = h->param.rc.f_aq_strength == 0 ? 0 : x264_clip3(e->aq_mode, 0, 1);
        h->param.i_cabac_init_idc = x264_clip3(e->cabac_init_idc, 0, 2);
        h->param.i_bframe = x264_clip3(e->bframes, 0, X264_BFRAME_MAX);
        h->param.analyse.b_weighted_bipred = e->weightb && h->param.i_bframe > 0;
        h->param.analyse.i_direct_mv_pred = e->direct_pred ? e->direct_pred : X264_DIRECT_PRED_SPATIAL;
        if (!p->subme && h->param.analyse.i_direct_mv_pred > X264_DIRECT_PRED_SPATIAL) h->param.analyse.i_direct_mv_pred = X264_DIRECT_PRED_SPATIAL;
        h->mb.b_direct_auto_write = h->param.analyse.i_direct_mv_pred == X264_DIRECT_PRED_AUTO && h->param.i_bframe;      /* encoder.c:460-462 (no 2-pass here) */
        if (p->subme >= 6 && !b_write) return -4;        /* the RD levels read the live entropy-coder state */
        x264_rdo_init();                                 /* R/encoder/encoder.c:728 */
    }
    h->param.rc.i_qp_min = p->cqm_preset ? 6 : 0; h->param.rc.i_qp_max = 51;   /* jvt: qp < 6 overflows the 16-bit multipliers (x264_cqm_init refuses) */
    if (p->qp == 0) {                                    /* x264_validate_parameters, R/encoder/encoder.c:401-421: lossless */
        h->mb.b_lossless = 1;
        h->param.rc.f_ip_factor = 1; h->param.rc.f_pb_factor = 1;
        h->param.analyse.i_chroma_qp_offset = 0; h->param.analyse.b_fast_pskip = 0; h->param.analyse.i_noise_reduction = 0;
        if (!h->param.b_cabac) h->param.analyse.b_transform_8x8 = 0;
    }
    h->thread[0] = h;
    h->sps = &h->sps_array[0]; h->pps = &h->pps_array[0];
    mb_w = h->sps->i_mb_width = (p->width + 15) / 16; mb_h = h->sps->i_mb_height = (p->height + 15) / 16;
    h->sps->b_frame_mbs_only = 1;
    h->sps->b_direct8x8_inference = 1;                   /* x264_sps_init, R/encoder/set.c:136 */
    n = h->mb.i_mb_count = mb_w * mb_h;
    h->pps->b_cabac = p->cabac; h->pps->b_transform_8x8_mode = h->param.analyse.b_transform_8x8;
    for (i = 0; i < 6; i++) h->pps->scaling_list[i] = p->cqm_preset ? x264_cqm_jvt[i] : flat16;   /* x264_pps_init, R/encoder/set.c */
    h->param.i_cqm_preset = p->cqm_preset;
    h->chroma_qp_table = i_chroma_qp_table + 12 + h->param.analyse.i_chroma_qp_offset;
    if (x264_cqm_init(h) < 0) return -1;
    x264_pixel_init(0, &h->pixf); x264_dct_init(0, &h->dctf); x264_zigzag_init(0, &h->zigzagf, 0);
    x264_quant_init(h, 0, &h->quantf); x264_mc_init(0, &h->mc); x264_deblock_init(0, &h->loopf);
    x264_dct_init_weights(); x264_init_vlc_tables();                                             /* R/encoder/encoder.c:736-743 */
    x264_predict_16x16_init(0, h->predict_16x16); x264_predict_8x8c_init(0, h->predict_8x8c);
    x264_predict_8x8_init(0, h->predict_8x8, &h->predict_8x8_filter); x264_predict_4x4_init(0, h->predict_4x4);
    sel_cmp(h);
    h->frames.b_have_lowres = stream;
    h->frames.b_have_sub8x8_esa = !!(h->param.analyse.inter & X264_ANALYSE_PSUB8x8);   /* encoder.c:717: the 4x4 integral plane of ESA */
    h->fenc = x264_frame_new(h);
    h->fdec = x264_frame_new(h);
    if (x264_macroblock_cache_init(h) < 0 || x264_ratecontrol_new(h) < 0) return -2;
    if (e && e->lowres_mv) {                              /* what x264_frame_new would have added with b_have_lowres (R/common/frame.c:80-96,140-142) */
        h->frames.b_have_lowres = 1;
        for (i = 0; i < 2; i++)
            for (k = 0; k <= h->param.i_bframe + 1 && k <= X264_BFRAME_MAX; k++) {
                h->fenc->lowres_mvs[i][k] = x264_malloc(2 * n * sizeof(int16_t));
                h->fenc->lowres_mvs[i][k][0][0] = 0x7fff;
            }
        if (h->param.rc.i_aq_mode) h->fenc->i_inv_qscale_factor = x264_malloc(n * sizeof(uint16_t));     /* written by x264_adaptive_quant_frame */
    }
    if (b_write) bsbuf = malloc(64 + (size_t)e->payload_cap + 4096);
    c->defer_end = 0; c->pending_len = 0;
    c->bsbuf = bsbuf; c->b_write = b_write; c->mb_w = mb_w; c->mb_h = mb_h; c->n = n; c->cw = p->width / 2; c->ch = p->height / 2;
    return 0;
}

/* x264_frame_copy_picture + x264_frame_expand_border_mod16 (R/encoder/encoder.c:1406-1413) */
static void load_picture(rctx *c, const refslice_params *p, x264_frame_t *fr, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v, size_t D)
{
    int y;
    for (y = 0; y < p->height; y++)
        memcpy(fr->plane[0] + y * fr->i_stride[0], src_y + (D * p->height + y) * p->width, p->width);
    for (y = 0; y < c->ch; y++) {
        memcpy(fr->plane[1] + y * fr->i_stride[1], src_u + (D * c->ch + y) * c->cw, c->cw);
        memcpy(fr->plane[2] + y * fr->i_stride[2], src_v + (D * c->ch + y) * c->cw, c->cw);
    }
    x264_frame_expand_border_mod16(c->h, fr);
}

/* one slice: x264_ratecontrol_start, x264_slice_init's fields, x264_slice_write's loop, the frame end (h->fenc, h->fdec, the lists are set) */
/* x264_encoder_frame_end's part of a coded frame (R/encoder/encoder.c:1722-1790): the rate control's state after the frame, --nr's tables, --direct auto's
 * running scores.  Not reached for a P frame the post-encode scene cut gives up (encoder.c:1645-1699 jumps back to do_encode before it). */
static void frame_end(rctx *c, refslice_out2 *o2, size_t F)
{
    x264_t *h = c->h;
    int i;
    if (c->b_write && c->stream) {
        x264_ratecontrol_end(h, 8 * c->pending_len);
        h->stat.i_slice_size[h->sh.i_type] += c->pending_len + 5;      /* NALU_OVERHEAD, encoder.c:44 */
        if (o2->rc_info) {
            o2->rc_info[4 * F] = h->sh.i_qp; o2->rc_info[4 * F + 1] = h->fdec->f_qp_avg_rc;
            o2->rc_info[4 * F + 2] = h->fdec->i_satd; o2->rc_info[4 * F + 3] = h->fdec->f_qp_avg_aq;
        }
    }
    x264_noise_reduction_update(h);                          /* encoder.c:1755 */
    if (h->sh.i_type == SLICE_TYPE_B && h->mb.b_direct_auto_write) {     /* encoder.c:1777-1790 */
        if (h->stat.i_direct_score[0] + h->stat.i_direct_score[1] > h->mb.i_mb_count)
            for (i = 0; i < 2; i++) h->stat.i_direct_score[i] = h->stat.i_direct_score[i] * 9 / 10;
        for (i = 0; i < 2; i++) h->stat.i_direct_score[i] += h->stat.frame.i_direct_score[i];
    }
}

static int code_frame(rctx *c, const refslice_params *p, const refslice_ext *e, refslice_out *o, refslice_out2 *o2, size_t F, int disp)
{
    x264_t *h = c->h;
    uint8_t *bsbuf = c->bsbuf;
    const int b_write = c->b_write, mb_w = c->mb_w, mb_h = c->mb_h, n = c->n, stream = c->stream;
    const int is_b = IS_X264_TYPE_B(h->fenc->i_type), idr = IS_X264_TYPE_I(h->fenc->i_type);
    int i, k, y;
    {
        memset(&h->sh, 0, sizeof(h->sh));
        h->sh.i_type = idr ? SLICE_TYPE_I : is_b ? SLICE_TYPE_B : SLICE_TYPE_P;
        h->sh.b_direct_spatial_mv_pred = h->param.analyse.i_direct_mv_pred == X264_DIRECT_PRED_SPATIAL;   /* x264_slice_header_init, encoder.c:113-119 */
        if (h->mb.b_direct_auto_write) h->sh.b_direct_spatial_mv_pred = h->stat.i_direct_score[1] > h->stat.i_direct_score[0];   /* --direct auto: the running scores decide */
        h->sh.i_first_mb = 0; h->sh.i_last_mb = n;
        h->sh.i_num_ref_idx_l0_active = h->i_ref0 <= 0 ? 1 : h->i_ref0;
        h->sh.i_num_ref_idx_l1_active = h->i_ref1 <= 0 ? 1 : h->i_ref1;
        h->sh.i_disable_deblocking_filter_idc = !p->deblock;
        h->sh.i_alpha_c0_offset = p->alpha_c0; h->sh.i_beta_offset = p->beta;
        h->sh.i_cabac_init_idc = h->param.i_cabac_init_idc;
        x264_ratecontrol_start(h, 0);
        h->sh.i_qp = x264_ratecontrol_qp(h);
        if (stream && o2->look_mv) {                     /* what x264_mb_analyse_inter_p16x16 / _b16x16 will be offered (R/encoder/analyse.c:1085-1100) */
            int16_t *lm = o2->look_mv + (size_t)F * 2 * n * 2;
            for (i = 0; i < 2 * n * 2; i++) lm[i] = i & 1 ? 0 : 0x7fff;
            if (h->i_ref0 > 0) memcpy(lm, h->fenc->lowres_mvs[0][h->fenc->i_frame - h->fref0[0]->i_frame - 1], 2 * n * sizeof(int16_t));
            if (h->i_ref1 > 0) memcpy(lm + 2 * n, h->fenc->lowres_mvs[1][h->fref1[0]->i_frame - h->fenc->i_frame - 1], 2 * n * sizeof(int16_t));
        }
        if (!stream && e && e->lowres_mv) {              /* this frame's lookahead vectors, filed under the distance to the reference they point at */
            const int16_t *lm = e->lowres_mv + (size_t)F * 2 * n * 2;
            if (h->i_ref0 > 0) memcpy(h->fenc->lowres_mvs[0][h->fenc->i_frame - h->fref0[0]->i_frame - 1], lm, 2 * n * sizeof(int16_t));
            if (h->i_ref1 > 0) memcpy(h->fenc->lowres_mvs[1][h->fref1[0]->i_frame - h->fenc->i_frame - 1], lm + 2 * n, 2 * n * sizeof(int16_t));
        }
        if (is_b) x264_macroblock_bipred_init(h);            /* encoder.c:1534-1535 */
        x264_macroblock_slice_init(h);
        /* PINNED (found with `make -C oracle msan`): an I slice never writes fdec->ref[0] / mv[0] (x264_macroblock_cache_save, R/common/macroblock.c:1296),
         * but for an I picture that is not an IDR x264_macroblock_slice_init leaves fdec->i_ref[0] = h->i_ref0 > 0, so the next P picture's
         * x264_mb_predict_mv_ref16x16 takes "temporal predictors" from those arrays (R/common/macroblock.c:420-441) -- whatever the frame
         * structure held before, malloc's leftovers on its first use: the reference's output then depends on what the process did earlier.  Here an I
         * picture has no motion to offer: ref = -1, which is also what the product's I slices leave in their state. */
        if (h->sh.i_type == SLICE_TYPE_I) {
            memset(h->fdec->ref[0], -1, 4 * (size_t)n * sizeof(int8_t));
            memset(h->fdec->mv[0], 0, 2 * 16 * (size_t)n * sizeof(int16_t));
        }
        memset(&h->stat.frame, 0, sizeof(h->stat.frame));
        int i_skip = 0;
        if (b_write) {                                    /* x264_slice_write after the header, encoder.c:1155-1165 */
            memset(bsbuf, 0, 64 + (size_t)e->payload_cap + 4096);
            bs_init(&h->out.bs, bsbuf + 64, e->payload_cap + 4096);
            if (h->param.b_cabac) {
                x264_cabac_context_init(&h->cabac, h->sh.i_type, h->sh.i_qp, h->sh.i_cabac_init_idc);
                x264_cabac_encode_init(&h->cabac, h->out.bs.p, h->out.bs.p_end);
            }
        }
        h->mb.i_last_qp = h->sh.i_qp; h->mb.i_last_dqp = 0;
        o->frame_info[4 * F] = h->sh.i_type; o->frame_info[4 * F + 1] = h->sh.i_qp;
        o->frame_info[4 * F + 2] = h->i_ref0; o->frame_info[4 * F + 3] = h->fdec->i_poc;
        if (o2 && o2->frame_info2) { o2->frame_info2[4 * F] = disp; o2->frame_info2[4 * F + 1] = h->i_ref1; o2->frame_info2[4 * F + 2] = !is_b; o2->frame_info2[4 * F + 3] = 0; }

        for (int mb = 0; mb < n; mb++) {
            int mx = mb % mb_w, my = mb / mb_w;
            size_t M = F * n + mb;
#ifdef REFSLICE_TRACE
            refslice_trace_frame = (int)F; refslice_trace_mb = mb; refslice_trace_h = h;      /* oracle/msan_main.c: where a sanitizer report happened */
#endif
            int16_t *ly = o->luma + M * 256, *ldc = o->luma_dc + M * 16, *cdc = o->chroma_dc + M * 8, *cac = o->chroma_ac + M * 128;
            uint8_t *nz = o->nnz + M * 27;
            if (mx == 0) filter_row(h, my);
            x264_macroblock_cache_load(h, mx, my);
            x264_macroblock_analyse(h);
            x264_macroblock_encode(h);
            /* reconstruction before deblocking */
            for (y = 0; y < 16; y++)
                memcpy(o->rec_y + (F * 16 * mb_h + 16 * my + y) * 16 * mb_w + 16 * mx, h->mb.pic.p_fdec[0] + y * FDEC_STRIDE, 16);
            for (y = 0; y < 8; y++) {
                memcpy(o->rec_u + (F * 8 * mb_h + 8 * my + y) * 8 * mb_w + 8 * mx, h->mb.pic.p_fdec[1] + y * FDEC_STRIDE, 8);
                memcpy(o->rec_v + (F * 8 * mb_h + 8 * my + y) * 8 * mb_w + 8 * mx, h->mb.pic.p_fdec[2] + y * FDEC_STRIDE, 8);
            }
            if (b_write) {                                /* encoder.c:1192-1219 */
                if (h->param.b_cabac) {
                    if (mb > 0) x264_cabac_encode_terminal(&h->cabac);
                    if (IS_SKIP(h->mb.i_type)) x264_cabac_mb_skip(h, 1);
                    else {
                        if (h->sh.i_type != SLICE_TYPE_I) x264_cabac_mb_skip(h, 0);
                        x264_macroblock_write_cabac(h, &h->cabac);
                    }
                    o2->mb_bits[M] = x264_cabac_pos(&h->cabac);
                } else {
                    if (IS_SKIP(h->mb.i_type)) i_skip++;
                    else {
                        if (h->sh.i_type != SLICE_TYPE_I) { bs_write_ue(&h->out.bs, i_skip); i_skip = 0; }
                        x264_macroblock_write_cavlc(h, &h->out.bs);
                    }
                    o2->mb_bits[M] = bs_pos(&h->out.bs);
                }
                if (o2->mb_bits[M] / 8 + 2048 > e->payload_cap) return -5;
            }
            x264_macroblock_cache_save(h);
            if (stream) x264_ratecontrol_mb(h, mb ? o2->mb_bits[M] - o2->mb_bits[M - 1] : o2->mb_bits[M]);   /* encoder.c:1240 */
            h->stat.frame.i_mb_count[h->mb.i_type]++;
            if (o2) o2->qp_offset[M] = h->param.rc.i_aq_mode ? h->fenc->f_qp_offset[mb] : 0;

            o->mb_type[M] = h->mb.i_type;
            o->partition[M] = IS_INTRA(h->mb.i_type) || IS_SKIP(h->mb.i_type) || h->mb.i_type == B_DIRECT ? D_16x16 : h->mb.i_partition;
            for (i = 0; i < 4; i++) o->sub_partition[M * 4 + i] = h->mb.i_type == P_8x8 || h->mb.i_type == B_8x8 ? h->mb.i_sub_partition[i] : 0;
            for (i = 0; i < 27; i++) nz[i] = h->mb.cache.non_zero_count[x264_scan8[i]];
            if (h->mb.i_type == I_PCM) memset(nz, 16, 27);
            o->qp[M] = h->mb.qp[mb]; o->cbp[M] = h->mb.cbp[mb]; o->t8[M] = h->mb.mb_transform_size[mb];
            o->i16mode[M] = h->mb.i_type == I_16x16 ? h->mb.i_intra16x16_pred_mode : 0;
            o->chroma_mode[M] = IS_INTRA(h->mb.i_type) ? h->mb.i_chroma_pred_mode : 0;
            for (i = 0; i < 16; i++)
                o->i4mode[M * 16 + i] = h->mb.i_type == I_4x4 || h->mb.i_type == I_8x8 ? h->mb.cache.intra4x4_pred_mode[x264_scan8[i]] : I_PRED_4x4_DC;
            if (h->sh.i_type != SLICE_TYPE_I) {
                for (i = 0; i < 16; i++) {
                    int o4 = h->mb.i_b4_xy + (i & 3) + (i >> 2) * h->mb.i_b4_stride;
                    o->mv[(M * 16 + i) * 2] = h->mb.mv[0][o4][0]; o->mv[(M * 16 + i) * 2 + 1] = h->mb.mv[0][o4][1];
                }
                for (i = 0; i < 4; i++) o->ref[M * 4 + i] = h->mb.ref[0][h->mb.i_b8_xy + (i & 1) + (i >> 1) * h->mb.i_b8_stride];
                if (o2 && o2->mv1) {
                    const int l1 = h->sh.i_type == SLICE_TYPE_B;
                    for (i = 0; i < 16; i++) {
                        int o4 = h->mb.i_b4_xy + (i & 3) + (i >> 2) * h->mb.i_b4_stride;
                        o2->mv1[(M * 16 + i) * 2] = l1 ? h->mb.mv[1][o4][0] : 0; o2->mv1[(M * 16 + i) * 2 + 1] = l1 ? h->mb.mv[1][o4][1] : 0;
                    }
                    for (i = 0; i < 4; i++) o2->ref1[M * 4 + i] = l1 ? h->mb.ref[1][h->mb.i_b8_xy + (i & 1) + (i >> 1) * h->mb.i_b8_stride] : -1;
                }
                for (k = 0; k < h->i_ref0; k++) {
                    o->mvr[((F * p->n_refs + k) * n + mb) * 2] = h->mb.mvr[0][k][mb][0];
                    o->mvr[((F * p->n_refs + k) * n + mb) * 2 + 1] = h->mb.mvr[0][k][mb][1];
                }
            } else {
                for (i = 0; i < 4; i++) o->ref[M * 4 + i] = -1;
                if (o2 && o2->ref1) for (i = 0; i < 4; i++) o2->ref1[M * 4 + i] = -1;
            }
            /* coefficient levels, masked by what the entropy coder would read (cbp, then nnz) */
            memset(ly, 0, 512); memset(ldc, 0, 32); memset(cdc, 0, 16); memset(cac, 0, 256);
            if (!IS_SKIP(h->mb.i_type) && h->mb.i_type != I_PCM) {
                if (h->mb.i_type == I_16x16 && nz[24]) memcpy(ldc, h->dct.luma16x16_dc, 32);
                if (h->mb.b_transform_8x8) {
                    for (i = 0; i < 4; i++)
                        if ((h->mb.i_cbp_luma >> i & 1) && nz[4 * i]) memcpy(ly + 64 * i, h->dct.luma8x8[i], 128);
                } else
                    for (i = 0; i < 16; i++)
                        if ((h->mb.i_cbp_luma >> (i >> 2) & 1) && nz[i]) memcpy(ly + 16 * i, h->dct.luma4x4[i], 32);
                if (h->mb.i_cbp_chroma)
                    for (i = 0; i < 2; i++) if (nz[25 + i]) memcpy(cdc + 4 * i, h->dct.chroma_dc[i], 8);
                if (h->mb.i_cbp_chroma == 2)
                    for (i = 0; i < 8; i++) if (nz[16 + i]) memcpy(cac + 16 * i, h->dct.luma4x4[16 + i], 32);
            }
        }
        if (b_write) {                                    /* encoder.c:1269-1280 */
            if (h->param.b_cabac) { x264_cabac_encode_flush(h, &h->cabac); h->out.bs.p = h->cabac.p; }
            else { if (i_skip > 0) bs_write_ue(&h->out.bs, i_skip); bs_rbsp_trailing(&h->out.bs); }
            int len = (int)(h->out.bs.p - (bsbuf + 64));
            if (len > e->payload_cap) return -5;
            o2->payload_len[F] = len;
            memcpy(o2->payload + F * e->payload_cap, bsbuf + 64, len);
            c->pending_len = len;
        }
        filter_row(h, mb_h);
        if (!c->defer_end) frame_end(c, o2, F);
        if (o2 && o2->frame_info2) o2->frame_info2[4 * F + 3] = h->sh.i_type == SLICE_TYPE_B ? h->sh.b_direct_spatial_mv_pred : 0;
        o->stat[4 * F] = h->stat.frame.i_intra_cost; o->stat[4 * F + 1] = h->stat.frame.i_inter_cost;
        o->stat[4 * F + 2] = h->stat.frame.i_mbs_analysed; o->stat[4 * F + 3] = 0;
        for (y = 0; y < 16 * mb_h; y++)
            memcpy(o->fin_y + (F * 16 * mb_h + y) * 16 * mb_w, h->fdec->plane[0] + y * h->fdec->i_stride[0], 16 * mb_w);
        for (y = 0; y < 8 * mb_h; y++) {
            memcpy(o->fin_u + (F * 8 * mb_h + y) * 8 * mb_w, h->fdec->plane[1] + y * h->fdec->i_stride[1], 8 * mb_w);
            memcpy(o->fin_v + (F * 8 * mb_h + y) * 8 * mb_w, h->fdec->plane[2] + y * h->fdec->i_stride[2], 8 * mb_w);
        }
    }
    return 0;
}

static int run_chain(const refslice_params *p, const refslice_ext *e, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v,
                     refslice_out *o, refslice_out2 *o2)
{
    rctx cx, *c = &cx;
    int rc_ = setup_encoder(c, p, e, 0);
    if (rc_) return rc_;
    x264_t *h = c->h;
    x264_frame_t *refs[16] = {0};
    int n_avail = 0, f, i, k, last_idr = 0;
    const int n = c->n;
    (void)n;

    /* coding order (what x264_slicetype_decide + the frame reordering of x264_encoder_encode give for a fixed B pattern):
     * anchors every bframes + 1 frames from the last IDR, the last frame before an IDR / the end of the clip is an anchor too;
     * each anchor is coded before the B frames that precede it in display order */
    const int nb = e ? h->param.i_bframe : 0, dpb = X264_MAX(p->n_refs, nb ? 2 : 1);   /* sps->vui.i_max_dec_frame_buffering, set.c */
    int *order = malloc(sizeof(int) * p->n_frames), *ftype = malloc(sizeof(int) * p->n_frames), n_order = 0;
    for (int t = 0; t < p->n_frames;) {
        int is_idr = p->keyint > 0 ? t % p->keyint == 0 : t == 0;
        if (is_idr) { order[n_order] = t; ftype[n_order++] = X264_TYPE_IDR; t++; continue; }
        int next_idr = p->keyint > 0 ? (t / p->keyint + 1) * p->keyint : p->n_frames, lim = X264_MIN(next_idr, p->n_frames);
        int anchor = X264_MIN(t + nb, lim - 1);
        order[n_order] = anchor; ftype[n_order++] = X264_TYPE_P;
        for (int b = t; b < anchor; b++) { order[n_order] = b; ftype[n_order++] = X264_TYPE_B; }
        t = anchor + 1;
    }
    for (f = 0; f < p->n_frames; f++) {
        const int disp = order[f], is_b = ftype[f] == X264_TYPE_B;
        int idr = ftype[f] == X264_TYPE_IDR;
        size_t F = f, D = disp;
        if (idr) {
            for (i = 0; i < n_avail; i++) x264_frame_delete(refs[i]);
            n_avail = 0; last_idr = disp;
        }
        load_picture(c, p, h->fenc, src_y, src_u, src_v, D);
        h->fenc->i_frame = disp; h->fenc->i_poc = 2 * (disp - last_idr);
        h->fenc->i_type = ftype[f];
        h->fdec->i_frame = disp; h->fdec->i_poc = h->fenc->i_poc; h->fdec->i_type = h->fenc->i_type;
        h->fenc->b_kept_as_ref = h->fdec->b_kept_as_ref = !is_b;
        h->i_frame = f;                                   /* frames coded so far (x264_reference_update, encoder.c:1063) */
        if (h->param.rc.i_aq_mode) x264_adaptive_quant_frame(h, h->fenc);   /* encoder.c:1421 */
        /* x264_reference_build_list, R/encoder/encoder.c:911-981: by POC, list 0 downwards from the frame, list 1 upwards */
        h->i_ref0 = h->i_ref1 = 0;
        for (i = 0; i < n_avail; i++) {
            if (refs[i]->i_poc < h->fdec->i_poc) h->fref0[h->i_ref0++] = refs[i];
            else if (refs[i]->i_poc > h->fdec->i_poc) h->fref1[h->i_ref1++] = refs[i];
        }
        for (i = 0; i < h->i_ref0; i++)
            for (k = i + 1; k < h->i_ref0; k++)
                if (h->fref0[k]->i_poc > h->fref0[i]->i_poc) { x264_frame_t *t_ = h->fref0[i]; h->fref0[i] = h->fref0[k]; h->fref0[k] = t_; }
        for (i = 0; i < h->i_ref1; i++)
            for (k = i + 1; k < h->i_ref1; k++)
                if (h->fref1[k]->i_poc < h->fref1[i]->i_poc) { x264_frame_t *t_ = h->fref1[i]; h->fref1[i] = h->fref1[k]; h->fref1[k] = t_; }
        h->i_ref1 = X264_MIN(h->i_ref1, nb ? 1 : 0);                /* h->frames.i_max_ref1 = sps->vui.i_num_reorder_frames */
        h->i_ref0 = X264_MIN(h->i_ref0, p->n_refs);
        h->mb.pic.i_fref[0] = h->i_ref0; h->mb.pic.i_fref[1] = h->i_ref1;
        if ((rc_ = code_frame(c, p, e, o, o2, F, disp)) != 0) return rc_;
        /* x264_reference_update, encoder.c:1060-1093: a disposable frame is dropped, a kept one pushes the oldest out of the DPB */
        if (!is_b) {
            if (n_avail == 16) x264_frame_delete(refs[--n_avail]);
            for (i = n_avail; i > 0; i--) refs[i] = refs[i - 1];
            refs[0] = h->fdec; n_avail++;
            if (n_avail > dpb) x264_frame_delete(refs[--n_avail]);
            h->fdec = x264_frame_new(h);
        }
    }
    for (i = 0; i < n_avail; i++) x264_frame_delete(refs[i]);
    x264_frame_delete(h->fdec); x264_frame_delete(h->fenc);
    x264_ratecontrol_delete(h);
    x264_macroblock_cache_end(h);
    x264_cqm_delete(h);
    free(c->bsbuf); free(order); free(ftype);
    free(h);
    return 0;
}

int refslice_encode_chain(const refslice_params *p, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v,
                          refslice_out *o)
{
    return run_chain(p, NULL, src_y, src_u, src_v, o, NULL);
}

int refslice_encode_chain2(const refslice_params *p, const refslice_ext *e, const uint8_t *src_y, const uint8_t *src_u,
                           const uint8_t *src_v, refslice_out *o, refslice_out2 *o2)
{
    return run_chain(p, e, src_y, src_u, src_v, o, o2);
}

/* x264_encoder_encode's frame queue with the real lookahead and rate control in front of the slice loop (R/encoder/encoder.c:1340-1600, one thread):
 * every picture enters frames.next with its half-resolution planes, x264_slicetype_decide types the head of the queue once the delay is filled,
 * the typed mini-GOP moves to frames.current anchor first, x264_ratecontrol_start prices the frame (CRF reads x264_rc_analyse_slice), and
 * x264_reference_update hands the source's lowres planes to the reconstructed frame.  Without --pre-scenecut the post-encode scene cut is here too
 * (encoder.c:1603-1699): a given-up P picture is coded again as I / IDR, or the B picture before it becomes the P and the queues are rearranged;
 * look_cost[..][7] carries the frame_num each coded picture's slice header had. */
static int run_stream(const refslice_params *p, const refslice_ext *e, const uint8_t *src_y, const uint8_t *src_u, const uint8_t *src_v,
                      refslice_out *o, refslice_out2 *o2)
{
    rctx cx, *c = &cx;
    int rc_, i, fed = 0, coded = 0, i_frame_num = 0, giveups = 0;
    if (!e || !e->write || !o2) return -4;
    if ((rc_ = setup_encoder(c, p, e, 1)) != 0) return rc_;
    c->defer_end = 1;                                            /* frame_end() is called below, after the post-encode scene cut had its say */
    x264_t *h = c->h;
    x264_frame_delete(h->fenc); x264_frame_delete(h->fdec);      /* made before b_have_lowres was known to x264_frame_new */
    h->frames.i_delay = h->param.i_bframe_adaptive == X264_B_ADAPT_TRELLIS ? X264_MAX(h->param.i_bframe, 3) * 4 : h->param.i_bframe;
    h->frames.i_max_ref0 = h->param.i_frame_reference;
    h->frames.i_max_ref1 = h->param.i_bframe ? 1 : 0;            /* sps->vui.i_num_reorder_frames, R/encoder/set.c:176 */
    h->frames.i_max_dpb = X264_MIN(16, X264_MAX(h->param.i_frame_reference, 1 + h->frames.i_max_ref1));
    h->frames.i_last_idr = -h->param.i_keyint_max;
    h->frames.i_input = 0; h->frames.last_nonb = NULL;
    h->fenc = NULL;
    h->fdec = x264_frame_pop_unused(h);
    h->i_frame = 0;

    while (coded < p->n_frames) {
        /* x264_reference_update, encoder.c:1060-1093 */
        if (h->fdec->i_frame >= 0) h->i_frame++;
        if (h->fdec->b_kept_as_ref) {
            for (i = 0; i < 4; i++) {
                XCHG(uint8_t *, h->fdec->lowres[i], h->fenc->lowres[i]);
                XCHG(uint8_t *, h->fdec->buffer_lowres[i], h->fenc->buffer_lowres[i]);
            }
            if (h->sh.i_type != SLICE_TYPE_B) h->frames.last_nonb = h->fdec;
            x264_frame_push(h->frames.reference, h->fdec);
            if (h->frames.reference[h->frames.i_max_dpb]) x264_frame_push_unused(h, x264_frame_shift(h->frames.reference));
            h->fdec = x264_frame_pop_unused(h);
        }
        if (h->fenc) { x264_frame_push_unused(h, h->fenc); h->fenc = NULL; }     /* x264_encoder_frame_end, encoder.c:1722 */
        h->fdec->i_frame = -1; h->fdec->b_kept_as_ref = 0;                       /* a recycled frame is not "coded" until it is */

        if (fed < p->n_frames) {
            x264_frame_t *fr = x264_frame_pop_unused(h);
            load_picture(c, p, fr, src_y, src_u, src_v, fed);
            fr->i_frame = h->frames.i_input++; fr->i_type = X264_TYPE_AUTO; fr->i_qpplus1 = 0;
            fed++;
            x264_frame_push(h->frames.next, fr);
            /* Two places x264_frame_init_lowres reads but nobody writes: the luma sample below-right of the picture (it copies the last row
             * without its duplicated last column, R/common/mc.c:316-317) and, when the lowres width is not a multiple of 16, the columns
             * between it and stride - 64, which x264_frame_expand_border_lowres takes for picture content (R/common/frame.c:297-302).  In
             * the encoder they hold what malloc / an earlier reconstruction left; fresh pages give 0, and so does this harness, every time. */
            memset(fr->buffer_lowres[0], 0, 4 * (size_t)fr->i_stride_lowres * (fr->i_lines[0] / 2 + 2 * PADV));
            fr->plane[0][fr->i_stride[0] * fr->i_lines[0] + fr->i_width[0]] = 0;
            x264_frame_init_lowres(h, fr);
            if (h->param.rc.i_aq_mode) x264_adaptive_quant_frame(h, fr);
            if (h->frames.i_input <= h->frames.i_delay) continue;                /* encoder.c:1425: the B buffer is filling */
        }
        if (h->frames.current[0] == NULL) {
            int bframes = 0;
            if (h->frames.next[0] == NULL) break;
            x264_slicetype_decide(h);
            while (IS_X264_TYPE_B(h->frames.next[bframes]->i_type)) bframes++;
            x264_frame_push(h->frames.current, x264_frame_shift(&h->frames.next[bframes]));
            while (bframes--) x264_frame_push(h->frames.current, x264_frame_shift(h->frames.next));
        }
        h->fenc = x264_frame_shift(h->frames.current);
do_encode:
        if (h->fenc->i_type == X264_TYPE_IDR) {
            h->frames.i_last_idr = h->fenc->i_frame;
            while (h->frames.reference[0]) x264_frame_push_unused(h, x264_frame_pop(h->frames.reference));   /* x264_reference_reset */
        }
        const int is_b = IS_X264_TYPE_B(h->fenc->i_type);
        h->fdec->i_poc = h->fenc->i_poc = 2 * (h->fenc->i_frame - h->frames.i_last_idr);
        h->fdec->i_type = h->fenc->i_type; h->fdec->i_frame = h->fenc->i_frame;
        h->fenc->b_kept_as_ref = h->fdec->b_kept_as_ref = !is_b && h->param.i_keyint_max > 1;
        /* x264_reference_build_list, encoder.c:911-981 */
        h->i_ref0 = h->i_ref1 = 0;
        for (i = 0; h->frames.reference[i]; i++) {
            if (h->frames.reference[i]->i_poc < h->fdec->i_poc) h->fref0[h->i_ref0++] = h->frames.reference[i];
            else if (h->frames.reference[i]->i_poc > h->fdec->i_poc) h->fref1[h->i_ref1++] = h->frames.reference[i];
        }
        for (i = 0; i < h->i_ref0; i++)
            for (int k = i + 1; k < h->i_ref0; k++)
                if (h->fref0[k]->i_poc > h->fref0[i]->i_poc) XCHG(x264_frame_t *, h->fref0[i], h->fref0[k]);
        for (i = 0; i < h->i_ref1; i++)
            for (int k = i + 1; k < h->i_ref1; k++)
                if (h->fref1[k]->i_poc < h->fref1[i]->i_poc) XCHG(x264_frame_t *, h->fref1[i], h->fref1[k]);
        h->i_ref1 = X264_MIN(h->i_ref1, h->frames.i_max_ref1);
        h->i_ref0 = X264_MIN(h->i_ref0, h->frames.i_max_ref0);
        h->mb.pic.i_fref[0] = h->i_ref0; h->mb.pic.i_fref[1] = h->i_ref1;
        /* x264_rc_analyse_slice reads frames.current behind the P frame; code_frame calls x264_ratecontrol_start */
        if ((rc_ = code_frame(c, p, e, o, o2, coded, h->fenc->i_frame)) != 0) return rc_;
        /* x264_encoder_encode, encoder.c:1538-1541: the slice header carries h->i_frame_num, a kept picture advances it (never wrapped, never reset at an
         * IDR -- except by the scene cut below) */
        const int frame_num_used = i_frame_num;
        if (h->fenc->b_kept_as_ref) i_frame_num++;
        /* the post-encode scene cut, encoder.c:1603-1699: a P picture whose inter cost is no better than its intra cost is given up */
        if (h->sh.i_type == SLICE_TYPE_P && h->param.i_scenecut_threshold >= 0 && !h->param.b_pre_scenecut) {
            int64_t i_inter_cost = h->stat.frame.i_inter_cost, i_intra_cost = h->stat.frame.i_intra_cost;
            float f_bias;
            int i_gop_size = h->fenc->i_frame - h->frames.i_last_idr;
            float f_thresh_max = h->param.i_scenecut_threshold / 100.0;
            float f_thresh_min = f_thresh_max * h->param.i_keyint_min / (h->param.i_keyint_max * 4);
            if (h->param.i_keyint_min == h->param.i_keyint_max) f_thresh_min = f_thresh_max;
            if (h->stat.frame.i_mbs_analysed > 0) i_intra_cost = i_intra_cost * c->n / h->stat.frame.i_mbs_analysed;
            if (i_gop_size < h->param.i_keyint_min / 4) f_bias = f_thresh_min / 4;
            else if (i_gop_size <= h->param.i_keyint_min) f_bias = f_thresh_min * i_gop_size / h->param.i_keyint_min;
            else f_bias = f_thresh_min + (f_thresh_max - f_thresh_min) * (i_gop_size - h->param.i_keyint_min) / (h->param.i_keyint_max - h->param.i_keyint_min);
            f_bias = X264_MIN(f_bias, 1.0);
            if (h->stat.frame.i_mbs_analysed > 0 && i_inter_cost >= (1.0 - f_bias) * i_intra_cost) {
                int b;
                giveups++;
                i_frame_num--;
                for (b = 0; h->frames.current[b] && IS_X264_TYPE_B(h->frames.current[b]->i_type); b++);
                if (b > 0) {
                    if (h->param.i_bframe_adaptive || b > 1) h->fenc->i_type = X264_TYPE_AUTO;
                    x264_frame_sort_pts(h->frames.current);
                    x264_frame_unshift(h->frames.next, h->fenc);
                    h->fenc = h->frames.current[b - 1];
                    h->frames.current[b - 1] = NULL;
                    h->fenc->i_type = X264_TYPE_P;
                    x264_frame_sort_dts(h->frames.current);
                } else if (i_gop_size >= h->param.i_keyint_min) {
                    i_frame_num = 0;
                    h->fenc->i_type = X264_TYPE_IDR;
                    h->fenc->i_poc = 0;
                    while (h->frames.current[0]) x264_frame_push(h->frames.next, x264_frame_shift(h->frames.current));
                    x264_frame_sort_pts(h->frames.next);
                } else
                    h->fenc->i_type = X264_TYPE_I;
                goto do_encode;
            }
        }
        frame_end(c, o2, coded);
        o->stat[4 * coded + 3] = giveups; giveups = 0;          /* attempts the post-encode scene cut gave up before this picture was coded for good */
        if (o2->look_cost) {
            int32_t *lc = o2->look_cost + 8 * (size_t)coded;
            int d0 = h->i_ref0 && !IS_X264_TYPE_I(h->fenc->i_type) ? h->fenc->i_frame - h->fref0[0]->i_frame : 0;
            int d1 = is_b ? h->fref1[0]->i_frame - h->fenc->i_frame : 0;
            lc[0] = is_b ? 0 : h->fenc->i_cost_est[d0][d1]; lc[1] = is_b ? 0 : h->fenc->i_cost_est_aq[d0][d1];
            lc[2] = is_b ? 0 : h->fenc->i_intra_mbs[d0]; lc[3] = h->fenc->i_cost_est[0][0]; lc[4] = d0; lc[5] = d1; lc[6] = h->fenc->i_type; lc[7] = frame_num_used;
        }
        coded++;
    }
    /* the frames are left to the process: the harness is one call per chain in a test, and x264_frame_delete of a frame whose lowres planes were
     * exchanged is the encoder's own business at x264_encoder_close */
    x264_ratecontrol_delete(h);
    x264_macroblock_cache_end(h);
    x264_cqm_delete(h);
    free(c->bsbuf);
    free(h);
    return coded == p->n_frames ? 0 : -7;
}

int refslice_encode_stream(const refslice_params *p, const refslice_ext *e, const uint8_t *src_y, const uint8_t *src_u,
                           const uint8_t *src_v, refslice_out *o, refslice_out2 *o2)
{
    return run_stream(p, e, src_y, src_u, src_v, o, o2);
}
