/* x264hip.h -- C ABI of the MI355X (gfx950) back-end for x264 core 66's
 * per-macroblock analyse/encode hot path.
 *
 * R/ = x264-snapshot-20090216-2245/ of chinaxuyongtao/x264-vs2008.
 *
 * Two levels, both plain C (pointers + sizes, no C++ or torch types):
 *
 *  1. TABLE LEVEL -- x264_<family>_init_hip() fill the reference's own six
 *     function-pointer tables (include/x264hip_tables.h) with entries that
 *     take the reference's exact arguments (host pointers) and run the
 *     arithmetic on the GPU.  This is the hook shape the reference already
 *     has for its SIMD back-ends (x264_pixel_altivec_init, R/common/pixel.c:
 *     781-786; x264_mc_init_mmx, R/common/mc.c:395-397) and is what
 *     x264_encoder_open calls at R/encoder/encoder.c:730-745.  One call = one
 *     launch + one sync: exact, but latency-bound; it exists for drop-in
 *     parity, not for speed.
 *
 *  2. FRAME LEVEL -- x264hip_frame_* / x264hip_*_frame keep whole planes
 *     resident in HBM and process every macroblock of a frame per launch
 *     (one wavefront per macroblock for motion search).  These are what the
 *     callers named below invoke when the HIP flag is set; see INTEGRATION.md.
 *
 * Error convention: table entries cannot fail (the reference's signatures
 * have no error channel); everything fallible happens in x264hip_init(),
 * which returns 0 or a negative code and leaves x264hip_last_error() set.
 * The *_init_hip() functions return -1 and leave the table untouched if the
 * library is not initialised.  Frame-level calls return 0 / negative.
 *
 * Threading: table entries use a per-thread stream + pinned staging arena
 * (the reference calls them concurrently from its frame threads without
 * locks, R/common/common.h:50).  Frame-level calls are stream-ordered on the
 * stream of the x264hip_frame_ctx they are given.
 */
#ifndef X264HIP_H
#define X264HIP_H

#include "x264hip_tables.h"

#ifdef __cplusplus
extern "C" {
#endif

/* flag bit proposed for x264_cpu_names[] / param.cpu: next free after
 * X264_CPU_LZCNT 0x010000 (R/x264.h:65) */
#define X264_CPU_HIP 0x020000

typedef struct {
    int    device;        /* HIP device ordinal */
    size_t arena_bytes;   /* per-thread pinned staging arena for table calls; 0 = default (4 MiB) */
} x264hip_cfg;

int         x264hip_init(const x264hip_cfg *cfg);   /* 0 ok; <0: no device / alloc failure */
void        x264hip_shutdown(void);
const char *x264hip_last_error(void);
int         x264hip_device_count(void);
/* plain device memory for hosts without a HIP binding of their own (tests, bench.py) */
void *x264hip_malloc(size_t bytes);          /* zero-filled; NULL on failure */
void  x264hip_free(void *dev);
int   x264hip_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int   x264hip_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int   x264hip_device_synchronize(void);
/* pinned host memory, asynchronous copies and streams for hosts without a HIP binding: frame ingest / payload egress overlapped with the
 * kernels (the staging of R/muxers.c:63-130 read_frame_yuv / write_nalu, with the DMA engines reading / writing the buffers directly) */
void *x264hip_host_alloc(size_t bytes);
void  x264hip_host_free(void *host);
int   x264hip_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, void *hip_stream);
int   x264hip_memcpy_h2d_async(void *dst_dev, const void *src_host, size_t bytes, void *hip_stream);
int   x264hip_mem_info(size_t *free_bytes, size_t *total_bytes);      /* hipMemGetInfo of the library's device */
void *x264hip_stream_create(void);
void  x264hip_stream_destroy(void *hip_stream);
int   x264hip_stream_synchronize(void *hip_stream);
/* HIP events on a named stream (timing of kernels on the stream they run on) */
void *x264hip_event_create(void);
void  x264hip_event_destroy(void *ev);
int   x264hip_event_record(void *ev, void *hip_stream);
float x264hip_event_elapsed_ms(void *start, void *stop);   /* waits for stop; <0 on error */
int   x264hip_stream_wait_event(void *hip_stream, void *ev);   /* work enqueued on hip_stream from now on runs after what ev recorded */

/* ---- table level: replaces x264_*_init(cpu, ...) of R/encoder/encoder.c:730-745 */
int x264_pixel_init_hip(x264hip_pixel_function_t *pixf);            /* R/common/pixel.c:565 */
int x264_dct_init_hip(x264hip_dct_function_t *dctf);                /* R/common/dct.c:388 */
int x264_zigzag_init_hip(x264hip_zigzag_function_t *pf, int b_interlaced); /* R/common/dct.c:626 */
int x264_quant_init_hip(x264hip_quant_function_t *pf);              /* R/common/quant.c:303 */
int x264_mc_init_hip(x264hip_mc_functions_t *pf);                   /* R/common/mc.c:359 */
int x264_predict_16x16_init_hip(x264hip_predict_t pf[7]);           /* R/common/predict.c:753 */
int x264_predict_8x8c_init_hip(x264hip_predict_t pf[7]);            /* R/common/predict.c:774 */
int x264_predict_4x4_init_hip(x264hip_predict_t pf[12]);            /* R/common/predict.c:818 */
int x264_predict_8x8_init_hip(x264hip_predict8x8_t pf[12], x264hip_predict_8x8_filter_t *filter); /* predict.c:795 */
int x264_deblock_init_hip(x264hip_deblock_function_t *pf);          /* R/common/frame.c:835 */

/* ---- frame level -------------------------------------------------------------
 * Plane layout follows x264_frame_new (R/common/frame.c:29-152): every plane
 * is padded by PADH = PADV = 32 pixels (R/common/frame.h:27-29), luma stride
 * = ALIGN(width + 64, 16), chroma stride = luma stride / 2; the four luma
 * planes of a reference (full, H, V, HV half-pel) are separate allocations
 * here.  All pointers handed back are DEVICE pointers to the first visible
 * pixel (not to the padding), exactly like x264_frame_t.plane[].          */
typedef struct x264hip_frame_ctx x264hip_frame_ctx;

typedef struct {
    int width, height;        /* visible luma size; coded size is rounded up to 16 */
    int mb_w, mb_h;           /* filled by x264hip_frame_ctx_new */
    int stride_y, stride_c;   /* filled by x264hip_frame_ctx_new */
    int lines_y, lines_c;     /* coded lines */
    int batch;                /* in: independent frames processed per launch (one per GOP chain); 0 = 1 */
} x264hip_frame_dims;

/* one picture resident in HBM: source (fenc) or reconstruction (fdec/ref) */
typedef struct {
    uint8_t *plane[3];        /* Y, U, V (device) */
    uint8_t *filtered[4];     /* [0] = plane[0], [1..3] = H, V, HV half-pel planes (device) */
    uint8_t *lowres[4];       /* half-resolution luma + its H,V,HV planes (device) */
    uint16_t *integral;       /* ESA integral image or NULL */
    int      stride_lowres, width_lowres, lines_lowres;
} x264hip_picture;

x264hip_frame_ctx *x264hip_frame_ctx_new(x264hip_frame_dims *dims, void *hip_stream /* NULL = own stream */);
void  x264hip_frame_ctx_delete(x264hip_frame_ctx *c);
void *x264hip_frame_ctx_stream(x264hip_frame_ctx *c);
int   x264hip_picture_alloc(x264hip_frame_ctx *c, x264hip_picture *pic);
/* a source frame: Y, U, V only (no half-pel / lowres planes); copy_element: one batch element's planes from picture to picture */
int   x264hip_picture_alloc_source(x264hip_frame_ctx *c, x264hip_picture *pic);
int   x264hip_picture_copy_element(x264hip_frame_ctx *c, x264hip_picture *dst, int dst_b, const x264hip_picture *src, int src_b);
void  x264hip_picture_free(x264hip_frame_ctx *c, x264hip_picture *pic);
int   x264hip_sync(x264hip_frame_ctx *c);
/* Batching: a context created with dims.batch = B holds B independent frames per picture (one per
 * GOP chain); every *_frame call below processes all B in the same launches (blockIdx.z).  Each
 * plane pointer of a picture addresses element 0; element b lies b * (padded plane size rounded to
 * 256 B) further.  Per-macroblock device arrays are B consecutive [n_mb][...] blocks.  upload /
 * download / x264hip_ssd_frame address the element chosen here (default 0).                      */
int   x264hip_frame_ctx_select(x264hip_frame_ctx *c, int batch_index);

/* x264_frame_copy_picture + border pad (R/common/frame.c:185-216, encoder.c:1406-1411):
 * host I420 -> device planes, then edges replicated into the padding.      */
int x264hip_picture_upload(x264hip_frame_ctx *c, x264hip_picture *pic,
                           const uint8_t *y, int sy, const uint8_t *u, int su, const uint8_t *v, int sv);
/* the same without the final synchronisation: y / u / v pinned (x264hip_host_alloc) and left alone until the stream has passed this
 * point; hip_stream NULL = the context's stream */
int x264hip_picture_upload_async(x264hip_frame_ctx *c, x264hip_picture *pic, const uint8_t *y, int sy, const uint8_t *u, int su,
                                 const uint8_t *v, int sv, void *hip_stream);
/* synthetic source: batch element b becomes frame t0 + b * t_stride of the integer-only test clip (the generator of
 * x264_vs2008_amd/synth.py, bit for bit), padded to the coded size; asynchronous on the context's stream.  Benchmarks take their input
 * from here, so every (chain, frame) is a picture of its own and no upload is involved. */
int x264hip_picture_synth(x264hip_frame_ctx *c, x264hip_picture *pic, int t0, int t_stride);
int x264hip_picture_download(x264hip_frame_ctx *c, const x264hip_picture *pic, int plane_id /*0..2 Y,U,V; 3..5 H,V,HV; 6..9 lowres*/,
                             uint8_t *dst, int dst_stride, int with_padding);
/* x264_frame_expand_border / _filtered / _lowres (R/common/frame.c:218-334) */
int x264hip_expand_border(x264hip_frame_ctx *c, x264hip_picture *pic, int which /*0 planes, 1 filtered, 2 lowres*/);
/* x264_frame_filter: H, V, HV half-pel planes of the whole frame (R/common/mc.c:404-426) */
int x264hip_hpel_filter_frame(x264hip_frame_ctx *c, x264hip_picture *pic);
/* x264_frame_init_lowres (R/common/mc.c:306-357) */
int x264hip_lowres_init_frame(x264hip_frame_ctx *c, x264hip_picture *pic);
/* Lookahead intra cost of every macroblock (the intra half of x264_slicetype_mb_cost,
 * R/encoder/slicetype.c:186-245): on the half-resolution plane produced by
 * x264hip_lowres_init_frame, each 8x8 block is predicted from its source neighbours with the
 * four 8x8c predictors and the six directional 8x8 predictors on the filtered edge, scored with
 * SATD 8x8; out[mb] = min + 5 = frame->i_intra_cost[mb].                                      */
int x264hip_lookahead_intra_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, int32_t *out_cost_dev);
/* AQ energy: pixf.var of Y 16x16 and U,V 8x8 per macroblock
 * (x264_adaptive_quant_frame's ac_energy_mb, R/encoder/ratecontrol.c:171-195).
 * out[mb] = var16(Y) + var8(U) + var8(V)                                    */
/* Host tables with floating-point arithmetic in the reference, built in C with its expression (no GPU involved):
 * p_cost_mv for one lambda (R/encoder/analyse.c:182-198), out[2 * span + 1] centred at span; h->unquant4_mf / unquant8_mf
 * (R/common/set.c:146,158) for every QP from the unshifted multipliers quant_mf6 [n_cat][6][n] -> out [n_cat][52][n]. */
void x264hip_cost_mv_table(int lambda, int span, int16_t *out);
/* x264_cqm_init (R/common/set.c:68-168) in the library's host C: every table the sweep's x264hip_slice_params / x264hip_slice_rd name,
 * from the PPS's six scaling lists (raster order; NULL or a NULL entry = flat 16) and param.analyse.i_luma_deadzone ({inter, intra};
 * NULL = {21, 11}).  -1 + error string on "Quantization overflow" at a QP >= qp_min. */
typedef struct {
    uint16_t quant4_mf[4][52][16], quant4_bias[4][52][16], quant8_mf[2][52][64], quant8_bias[2][52][64];
    int32_t  dequant4_mf[4][6][16], dequant8_mf[2][6][64];
    int32_t  unquant4_mf[4][52][16], unquant8_mf[2][52][64];
} x264hip_cqm_tables;
int x264hip_cqm_init(const uint8_t *const scaling_list[6], const int luma_deadzone[2], int qp_min, x264hip_cqm_tables *out);
void x264hip_unquant_table(const int32_t *quant_mf6, int n_cat, int n, int32_t *out);
/* x264_nal_encode (R/common/common.c:656): start code (b_annexb) + NAL header + payload with emulation prevention; returns the size.
 * Host side, no device needed; dst must hold 5 + len * 3 / 2 bytes. */
int x264hip_nal_encode(uint8_t *dst, int b_annexb, int i_ref_idc, int i_type, const uint8_t *payload, int len);
int x264hip_aq_var_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, int32_t *out_dev);
/* x264_adaptive_quant_frame (R/encoder/ratecontrol.c:231-249): fenc->f_qp_offset[batch][n_mb] (float, device) from the energies
 * above; aq_strength = param.rc.f_aq_strength.  energy_dev: scratch [batch][n_mb] int32. */
int x264hip_adaptive_quant_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, float aq_strength, int32_t *energy_dev, float *offset_dev);
/* x264_pixel_ssd_wxh over the three planes (PSNR, R/encoder/encoder.c:1034-1045) */
int x264hip_ssd_frame(x264hip_frame_ctx *c, const x264hip_picture *a, const x264hip_picture *b, int64_t ssd_host[3]);
/* same, stream-ordered with the three sums left in device memory (no host sync) */
int x264hip_ssd_frame_async(x264hip_frame_ctx *c, const x264hip_picture *a, const x264hip_picture *b, uint64_t *ssd_dev3);

/* Motion search, one wavefront per macroblock (x264_me_search_ref's full-pel
 * stage as an exhaustive +-range window, R/encoder/me.c:156-631 with the ESA
 * cost rule :449-470; cost = SAD + p_cost_mv[mx - mvp.x] + p_cost_mv[my - mvp.y],
 * COST_MV R/encoder/me.c:54-62).  For every macroblock and every partition
 * of {16x16, 16x8 x2, 8x16 x2, 8x8 x4} it returns the best full-pel vector
 * and its cost; ties resolve to the first candidate in raster order of
 * (my, mx), which is the order of the reference's ESA scan (me.c:480-560).
 *   cost_mv : device table of uint16 costs indexed by qpel delta + 2*cost_mv_range
 *   centers : per-MB search centre in full pels (int16 x,y) or NULL for (0,0)
 *   mvp     : per-MB predictor in qpel (int16 x,y) or NULL for (0,0)
 *   out_mv  : [mb][9] int16 x,y (full-pel, absolute)   out_cost : [mb][9] int32 */
typedef struct {
    int range;                 /* full-pel search range, <= 24 */
    const uint16_t *cost_mv;   /* device; centre entry at cost_mv[cost_mv_range] */
    int cost_mv_range;         /* qpel span on each side */
    const int16_t *centers;    /* device or NULL */
    const int16_t *mvp;        /* device or NULL */
    uint16_t *sad_surface;     /* device or NULL: [mb][(2*range+1)^2] raw 16x16 SADs in (my,mx) raster
                                  order; entries of vectors outside the mv limits are unspecified */
    int mv_range;              /* param.analyse.i_mv_range in pixels (level limit); 0 = 512 */
} x264hip_me_params;
int x264hip_me_fullpel_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                             const x264hip_me_params *p, int16_t *out_mv_dev, int32_t *out_cost_dev);

/* Sub-pel refinement of the 16x16 vector (refine_subpel, R/encoder/me.c:680-778,
 * exhaustive form): half-pel then quarter-pel 3x3 neighbourhoods scored with
 * SATD (mbcmp) + mv cost through the qpel blend of the four planes
 * (get_ref, R/common/mc.c:181-202).  in/out mv in qpel.                      */
int x264hip_me_subpel_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                            const x264hip_me_params *p, const int16_t *mv_fullpel_dev /* [mb][9][2], uses entry 0 */,
                            int16_t *out_mv_qpel_dev /* [mb][2] */, int32_t *out_cost_dev /* [mb] */);

/* The reference's own search, exactly: x264_me_search_ref + refine_subpel for the 16x16 block
 * (R/encoder/me.c:156-778) for every macroblock and every reference, inside the loop of
 * x264_mb_analyse_inter_p16x16 (R/encoder/analyse.c:1077-1127): predictor tests, DIA / HEX / UMH walk,
 * square refine, half-pel + quarter-pel diamonds (SAD / SATD per subme, chroma ME), the half-pel
 * early-termination threshold carried across references, best reference = first minimum of
 * cost + ref_cost.  Predictors are inputs (the caller derives them from neighbours / lookahead).
 *   cost_mv  : device int16 table, cost of qpel delta d at cost_mv[cost_mv_range + d] (p_cost_mv)
 *   mvp      : [mb][n_refs][2] qpel;  mvc : [mb][n_refs][8][2];  n_mvc : [mb][n_refs] (0..8)
 *   out_mv   : [mb][n_refs][2] qpel;  out_cost : [mb][n_refs] (ref cost included)
 *   best     : [mb][4] = {ref, mvx, mvy, cost}                                             */
typedef struct {
    int me_method;             /* 0 = X264_ME_DIA, 1 = X264_ME_HEX, 2 = X264_ME_UMH, 3 = X264_ME_ESA (subme >= 1); TESA is refused */
    int me_range, subme, chroma_me;
    int mv_range;              /* pixels; 0 = 512 */
    const int16_t *cost_mv;    /* device */
    int cost_mv_range;
    const int16_t *mvp, *mvc;  /* device */
    const uint8_t *n_mvc;      /* device */
    int ref_cost[8];           /* REF_COST(0, i) = lambda * bs_size_te(n_refs - 1, i) */
} x264hip_me16_params;
int x264hip_me_search16_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *const *refs,
                              int n_refs, const x264hip_me16_params *p, int16_t *out_mv_dev, int32_t *out_cost_dev,
                              int32_t *best_dev);

/* ---- the per-macroblock hot loop itself ------------------------------------------------------
 * One launch per frame runs, for every macroblock of every chain of the batch, what
 * x264_slice_write does (R/encoder/encoder.c:1171-1222): x264_macroblock_cache_load
 * (R/common/macroblock.c:872), x264_macroblock_analyse (R/encoder/analyse.c:2156),
 * x264_macroblock_encode (R/encoder/macroblock.c:475), x264_macroblock_cache_save
 * (R/common/macroblock.c:1208), in the wavefront order their neighbour dependences allow.  The
 * entropy coder is not part of it: it consumes x264hip_mb_state on the host.
 *
 * x264hip_mb_state = what the reference keeps per frame in x264_frame_t / h->mb (mb_type, mv, ref,
 * non_zero_count, intra modes, qp, cbp ...) plus h->dct for every macroblock; every array is
 * [batch][n_mb][...] in HBM.  Numbering follows the reference: mb types R/common/macroblock.h:78-102
 * (I_4x4 0, I_8x8 1, I_16x16 2, I_PCM 3, P_L0 4, P_8x8 5, P_SKIP 6), partitions :55-76, intra modes
 * R/common/predict.h:31-107, nnz index = block index of x264_scan8 (0-15 luma, 16-23 chroma AC,
 * 24 luma DC, 25/26 chroma DC).  Levels are in scan order and zero wherever cbp / nnz say "not coded". */
typedef struct {
    int8_t  *mb_type, *partition, *sub_partition;   /* sub_partition [mb][4]: D_L0_4x4 0, 8x4 1, 4x8 2, 8x8 3 of each 8x8 block of a P_8x8 macroblock */
    int8_t  *ref;          /* [n][4] per 8x8; -1 intra */
    int8_t  *i4mode;       /* [n][16] intra 4x4 / 8x8 modes by block index; I_PRED_4x4_DC elsewhere */
    int8_t  *i16mode, *chroma_mode, *qp, *t8;
    int16_t *mv;           /* [n][16][2] quarter-pel, 4x4 blocks in raster order inside the macroblock */
    int16_t *mvr;          /* [8][n][2] 16x16 search result per reference (h->mb.mvr) */
    int16_t *cbp;          /* h->mb.cbp: dc << 8 | chroma << 4 | luma */
    uint8_t *nnz;          /* [n][27] */
    int16_t *luma, *luma_dc, *chroma_dc, *chroma_ac;   /* [n][256] [n][16] [n][2][4] [n][8][16] */
    int32_t *cost_intra, *cost_inter;                  /* per macroblock terms of h->stat.frame.i_intra_cost / i_inter_cost */
    int32_t *cost_intra_alt;                           /* scratch of the sweep: b_fast_intra's raster-order term, settled after the frame */
    int32_t *progress;     /* [batch][mb_h] + abort flag: the sweep's row counters */
    int poc, n_ref0, inv_ref_poc[8];                   /* x264_frame_t.i_poc / i_ref[0] / inv_ref_poc, filled by the sweep */
    int16_t *mvd;          /* [n][16][2] h->mb.mvd[0] (CABAC contexts of the row below; raster variant only) */
    /* B slices: list 1 of the same (h->mb.mv[1] / ref[1] / mvr[1][0] / mvd[1]) and h->mb.skipbp */
    int16_t *mv1;          /* [n][16][2] */
    int8_t  *ref1;         /* [n][4] */
    int16_t *mvr1;         /* [n][2] */
    int16_t *mvd1;         /* [n][16][2] */
    uint8_t *skipbp;       /* [n] */
    int ref_poc[8];        /* fdec->ref_poc[0][]: the POCs of the list-0 pictures this frame was coded from, filled by the sweep (a later B
                              frame's temporal direct prediction maps its co-located references through them, h->mb.map_col_to_list0) */
} x264hip_mb_state;

/* ---- round 2: the raster-order variant of the sweep ------------------------------------------------
 * With the RD levels (subme >= 6: every trial encode is priced against the LIVE CABAC contexts, R/encoder/rdo.c:62,139-171),
 * trellis quantisation (rdo.c:475-493) or adaptive quantisation (a macroblock's QP follows from the previous one's,
 * R/encoder/ratecontrol.c:263-264) a slice is one serial chain of macroblocks.  When x264hip_slice_params.rd is set, one
 * wavefront owns a whole frame of one chain of the batch and walks it in raster order, and the entropy coder runs inside the
 * loop exactly where x264_slice_write has it (R/encoder/encoder.c:1155-1165,1192-1205,1269-1273): the launch also returns every
 * chain's slice_data() bytes.  Throughput then comes from the number of chains in flight (the batch), not from a wavefront
 * schedule inside the frame.  I, P and B slices, CABAC; refused with an error string: sub-8x8 partitions together with the RD levels,
 * psy-trellis, subme >= 8, CAVLC together with the writer.                                                                */
typedef struct x264hip_slice_rd {
    int trellis;                   /* param.analyse.i_trellis 0..2 */
    int psy_rd;                    /* h->mb.i_psy_rd = FIX8(param.analyse.f_psy_rd) (0 below subme 6); the caller lowers chroma_qp_offset
                                      as x264_validate_parameters does (R/encoder/encoder.c:509-514) */
    int write;                     /* 1: x264_macroblock_write_cabac after every macroblock (required for subme >= 6 / trellis) */
    int cabac_init_idc;            /* param.i_cabac_init_idc */
    int i_frame;                   /* frames coded before this one (x264_cabac_encode_flush's padding bit, R/common/cabac.c:918) */
    int qp_min, qp_max;            /* param.rc.i_qp_min / i_qp_max (adaptive quantisation clips to them) */
    float f_qpm;                   /* rc->f_qpm: the frame's QP before the per-macroblock offset */
    const float *aq_offset;        /* device [batch][n_mb]: fenc->f_qp_offset (x264_adaptive_quant_frame), or NULL = no AQ */
    const int16_t *cost_mv_all;    /* device [52][2 * cost_mv_range + 1]: p_cost_mv of every QP (needed with aq_offset) */
    const int32_t *unquant4_mf;    /* device [4][52][16]  h->unquant4_mf (trellis) */
    const int32_t *unquant8_mf;    /* device [2][52][64]  h->unquant8_mf */
    uint8_t *payload;              /* device [batch][payload_cap]: every chain's slice_data() starts 64 bytes into its slot */
    int payload_cap;
    int32_t *payload_len;          /* device [batch] */
    int32_t *mb_bits;              /* optional device [batch][n_mb]: x264_cabac_pos after every macroblock */
    int16_t *stale;                /* device [batch][8] (zero before a chain's first frame), or NULL: the h->mb.cache.ref / mv entry of block 12 of
                                      both lists {ref0, mvx0, mvy0, ref1, mvx1, mvy1}.  x264_macroblock_cache_load never rewrites the cache's inner
                                      entries, so this one survives from macroblock to macroblock and from frame to frame, and a B macroblock whose
                                      TEMPORAL direct prediction fails (a co-located reference outside list 0) offers it to its 16x16 searches as the
                                      "direct" candidate (x264_mb_predict_mv_ref16x16, R/common/macroblock.c:376-386).  Every sweep of a chain that
                                      uses temporal direct prediction reads and updates it; without it temporal direct is refused */
    int i_frame_stride;            /* chain b of the batch has coded i_frame + b * i_frame_stride frames before this one: the chains of a
                                      launch may be the closed GOPs of ONE stream (GOP g starts keyint frames after GOP g - 1), see
                                      x264_vs2008_amd/shard.py.  0: every chain counts alike */
} x264hip_slice_rd;
#define X264HIP_PAYLOAD_LEAD 64

struct x264hip_slice_rd;
struct x264hip_slice_b;
typedef struct {
    int slice_type;                    /* 0 = SLICE_TYPE_P, 1 = SLICE_TYPE_B (needs .b and .rd), 2 = SLICE_TYPE_I (R/common/common.h:128-134) */
    int qp, chroma_qp_offset;
    int me_method, me_range, subme, chroma_me, mv_range;       /* param.analyse.* */
    int fast_pskip, dct_decimate, cabac, transform8x8;
    int analyse_inter, analyse_intra;  /* X264_ANALYSE_* masks (R/x264.h:190-199) */
    const uint16_t *quant4_mf, *quant4_bias, *quant8_mf, *quant8_bias;   /* device, as x264hip_residual_params */
    const int32_t *dequant4_mf, *dequant8_mf;
    const int16_t *cost_mv;            /* device: p_cost_mv for this qp's lambda, centred at cost_mv_range */
    int cost_mv_range;
    int poc, ref_poc[8];               /* fdec->i_poc and fdec->ref_poc[0][] */
    int mixed_refs;                    /* param.analyse.b_mixed_references (p8x8 blocks search every reference) */
    int64_t *profile;                  /* NULL, or device [batch][mb_h][8]: 100 MHz ticks each row wave spent
                                          0 waiting 1 loading 2 inter search 3 encode 4 stores 5 publish 6 intra analysis */
    /* param.analyse.i_noise_reduction (R/encoder/macroblock.c:632-637,682-695): when non-zero, inter luma coefficients go through
     * x264_denoise_dct with nr->offset before quantisation and their magnitudes / block counts are added to nr->sum / nr->count */
    int noise_reduction;
    const struct x264hip_nr_state *nr;
    /* h->mb.b_lossless (constant QP 0, R/encoder/encoder.c:401-421): predictive lossless intra prediction, the prediction error
     * itself in zigzag order as levels, SAD for every comparison, no transform-size analysis.  The caller applies the rest of
     * x264_validate_parameters: qp 0 for every slice, chroma_qp_offset 0, fast_pskip 0, noise_reduction 0, 8x8dct only with CABAC */
    int lossless;
    const struct x264hip_slice_rd *rd;   /* NULL: the wavefront schedule of round 1; set: the raster-order variant (below) */
    /* fenc->lowres_mvs[0][fenc->i_frame - fref0[0]->i_frame - 1] of every chain (h->frames.b_have_lowres): device [batch][n_mb][2]
     * int16, the lookahead's half-resolution vectors towards reference 0, which x264_mb_predict_mv_ref16x16 offers (doubled) to the
     * 16x16 search on reference 0 (R/common/macroblock.c:393-398).  NULL, or 0x7fff in a chain's first component: none */
    const int16_t *lowres_mv;
    const struct x264hip_slice_b *b;     /* slice_type 1 (B): list 1 and what direct prediction reads (below); needs rd */
} x264hip_slice_params;

/* A B slice (slice_type = 1; the raster variant with the entropy coder: rd set, write = 1, subme 2..7).  x264 core 66 without
 * b-pyramid has one list-1 picture and its B frames are disposable (never references).  refs / n_refs of the call are list 0
 * (x264_reference_build_list, R/encoder/encoder.c:911-981: earlier pictures, nearest first); l0 = refs[0]'s state as always. */
typedef struct x264hip_slice_b {
    const x264hip_picture *fref1;        /* h->fref1[0]: the next anchor, reconstructed, borders expanded, half-pel planes built */
    const x264hip_mb_state *l1_state;    /* the state it was coded with: mb_type / ref / mv of the co-located macroblocks (x264_mb_predict_mv_direct16x16) */
    int ref1_poc;                        /* h->fref1[0]->i_poc (x264_macroblock_bipred_init, R/common/macroblock.c:1374-1408) */
    int weightb;                         /* param.analyse.b_weighted_bipred */
    const int16_t *lowres_mv1;           /* fenc->lowres_mvs[1][fref1[0]->i_frame - fenc->i_frame - 1], as x264hip_slice_params.lowres_mv, or NULL */
    int direct_spatial;                  /* sh.b_direct_spatial_mv_pred; 0 = temporal (R/common/macroblock.c:155-224): needs rd.stale in every sweep of the chain */
    int32_t *direct_score;               /* --direct auto (h->mb.b_direct_auto_write): device [batch][2].  The sweep then predicts BOTH direct modes in every
                                          * macroblock and leaves h->stat.frame.i_direct_score[0 temporal, 1 spatial] here (R/encoder/analyse.c:2476-2496); the host
                                          * keeps the running scores that pick direct_spatial of the next B frame (encoder.c:113-118,1777-1790).  NULL: off.  Needs rd.stale. */
} x264hip_slice_b;

/* h->nr_residual_sum / nr_count / nr_offset of every chain of the batch (R/common/common.h:308-310), device memory:
 * sum [batch][2][64] uint32 (cat 0: 4x4, first 16 used; cat 1: 8x8), count [batch][2] uint32, offset [batch][2][64] uint16 */
typedef struct x264hip_nr_state {
    uint32_t *sum, *count;
    uint16_t *offset;
} x264hip_nr_state;
int  x264hip_nr_state_alloc(x264hip_frame_ctx *c, x264hip_nr_state *nr);
void x264hip_nr_state_free(x264hip_frame_ctx *c, x264hip_nr_state *nr);
/* x264_noise_reduction_update (R/encoder/macroblock.c:890-911) for every chain: call once after each frame's sweep */
int  x264hip_noise_reduction_update(x264hip_frame_ctx *c, const x264hip_nr_state *nr, int noise_reduction);

int  x264hip_mb_state_alloc(x264hip_frame_ctx *c, x264hip_mb_state *st);
#define X264HIP_STATE_NO_LEVELS 1   /* luma / luma_dc / chroma_dc / chroma_ac stay NULL: for sweeps that write the payload themselves (rd.write) */
int  x264hip_mb_state_alloc_ex(x264hip_frame_ctx *c, x264hip_mb_state *st, int flags);
void x264hip_mb_state_free(x264hip_frame_ctx *c, x264hip_mb_state *st);
/* refs: list0, most recent first (reconstructed, borders expanded, half-pel planes built);
 * l0 = the state refs[0] was coded with (temporal predictors, fast-intra test) or NULL;
 * recon: receives the unfiltered reconstruction (deblock / expand / hpel are separate calls). */
int  x264hip_slice_sweep_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *const *refs,
                               int n_refs, x264hip_picture *recon, const x264hip_slice_params *p,
                               const x264hip_mb_state *l0, x264hip_mb_state *out);
/* synchronises; -1 if a wavefront gave up waiting for its neighbours (the frame is then invalid) */
int  x264hip_slice_sweep_status(x264hip_frame_ctx *c, const x264hip_mb_state *st);

/* Inter residual pipeline for every macroblock (x264_macroblock_encode's
 * inter branch, R/encoder/macroblock.c:596-768, without trellis/denoise):
 * x264_mb_mc 16x16 (mc_luma + mc_chroma) -> sub16x16_dct(8) -> quant ->
 * zigzag scan -> decimate -> dequant -> add idct into the reconstruction;
 * chroma: sub8x8_dct, 2x2 DC, quant, decimate (<7), dequant, idct.
 *   mv_qpel : [mb][2]; qp / qp_chroma : per-frame luma and chroma QP
 *   (h->mb.i_chroma_qp); transform8x8 : 0 = 4x4, 1 = 8x8 luma transform;
 *   b_interlaced selects the field scan tables.
 *   levels_y : [mb][16][16] (4x4) or [mb][4][64] (8x8) scanned levels, zero
 *              for blocks that quantise to nothing;
 *   levels_c : [mb][2][4][16] scanned AC (index 0 is the removed DC = 0);
 *   dc_c    : [mb][2][4] quantised 2x2 chroma DC in zigzag_scan_2x2_dc order
 *   cbp     : [mb] luma cbp (bits 0-3) | chroma cbp (0,1,2) << 4
 *   nnz     : [mb][26] non-zero flags: 16 Y 4x4 blocks (x264 z-order), 4 U,
 *              4 V AC blocks, U DC, V DC -- after decimation, as stored in
 *              h->mb.cache.non_zero_count                                   */
typedef struct {
    int qp, qp_chroma, transform8x8, b_interlaced;
    const uint16_t *quant4_mf, *quant4_bias;   /* device [4][52][16] */
    const uint16_t *quant8_mf, *quant8_bias;   /* device [2][52][64] */
    const int32_t  *dequant4_mf;               /* device [4][6][16]  */
    const int32_t  *dequant8_mf;               /* device [2][6][64]  */
    /* optional (NULL = skip): the frame's mv / ref arrays as x264_macroblock_cache_save
     * leaves them (R/common/macroblock.c:1264-1295): [mb][16][2] qpel, [mb][4] ref idx */
    int16_t *mv4x4_out;
    int8_t  *ref_out;
} x264hip_residual_params;
int x264hip_inter_residual_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                                 x264hip_picture *recon, const x264hip_residual_params *p,
                                 const int16_t *mv_qpel_dev, int16_t *levels_y_dev, int16_t *levels_c_dev,
                                 int16_t *dc_c_dev, int32_t *cbp_dev, uint8_t *nnz_dev);

/* General P form of the same pipeline: any partition shape and several references.
 * x264_mb_mc (R/common/macroblock.c:462-546) is mc_luma + mc_chroma per partition with its own
 * vector and reference, which per pixel depends only on the 4x4 block it lies in, so the
 * partitioning is given as one vector per 4x4 block (raster order inside the macroblock, the
 * layout of x264_frame_t.mv) and one reference index per 8x8 (x264_frame_t.ref); refs[] lists
 * up to 8 reference pictures (list0 order).  mv4x4 : [mb][16][2], ref8x8 : [mb][4] or NULL (= 0).
 * Must not alias params->mv4x4_out / ref_out.                                                */
int x264hip_inter_residual_frame_mp(x264hip_frame_ctx *c, const x264hip_picture *fenc,
                                    const x264hip_picture *const *refs, int n_refs, x264hip_picture *recon,
                                    const x264hip_residual_params *p, const int16_t *mv4x4_dev, const int8_t *ref8x8_dev,
                                    int16_t *levels_y_dev, int16_t *levels_c_dev, int16_t *dc_c_dev, int32_t *cbp_dev,
                                    uint8_t *nnz_dev);

/* x264_macroblock_probe_skip for every macroblock (R/encoder/macroblock.c:797-883, P path):
 * skip_out[mb] = 1 iff the macroblock predicted with its P-skip vector (clipped to mv_min/mv_max)
 * quantises to nothing: summed luma decimate scores < 6 and, per chroma plane whose SSD reaches
 * (lambda2_chroma + 32) >> 6, a zero 2x2 DC and summed AC scores < 7.  lambda2_chroma =
 * x264_lambda2_tab[qp_chroma] (R/encoder/analyse.c:151-159).  pskip_mv : [mb][2] qpel.       */
int x264hip_probe_skip_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                             const x264hip_residual_params *p, int lambda2_chroma, const int16_t *pskip_mv_dev,
                             uint8_t *skip_out_dev);

/* x264_frame_deblock_row for every row of a progressive P frame
 * (R/common/frame.c:621-792): bS from intra flags / nnz / mv+ref differences
 * (no_sub8x8 partitions), alpha/beta/tc0 from the per-MB qp and the slice
 * offsets, P_SKIP and low-qp macroblocks filter only their outer edge.
 * Macroblocks are filtered in the standard's order (raster; vertical then
 * horizontal edges): the frame is swept in 2:1 anti-diagonals, one launch per
 * diagonal, because MB (x,y) needs (x-1,y) and (x+1,y-1) finished.
 *   mb_type : [mb] u8, 0 inter, 1 intra, 2 P_SKIP, 3 P_8x8 with sub-8x8 analysis on ; qp : [mb] u8 ;
 *   nnz : [mb][26] as x264hip_inter_residual_frame writes it ;
 *   mv : [mb][16][2] int16 qpel per 4x4 block in raster order ;
 *   ref : [mb][4] int8 per 8x8 ; transform8x8 : [mb] u8 (all device) */
typedef struct {
    const uint8_t *mb_type, *qp, *nnz, *transform8x8;
    const int16_t *mv;
    const int8_t  *ref;
    int alpha_c0_offset, beta_offset, chroma_qp_offset;
    int state_layout;      /* 0: compact codes above, nnz [mb][26]; 1: the arrays of an x264hip_mb_state (reference type numbers, nnz [mb][27]) */
    int sub8x8;            /* param.analyse.inter & X264_ANALYSE_PSUB8x8: P_8x8 macroblocks (layout 1: type 5; layout 0: code 3) compare vectors
                            * on every 4-pixel edge segment (no_sub8x8 = 0, R/common/frame.c:645) */
} x264hip_deblock_params;
int x264hip_deblock_frame(x264hip_frame_ctx *c, x264hip_picture *recon, const x264hip_deblock_params *p);

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Lookahead and rate control of ONE GOP chain: the host half.  x264_encoder_encode's frame queue (R/encoder/encoder.c:1390-1470:
 * frames.next / frames.current, the B-frame delay, x264_reference_update's last_nonb and DPB), x264_slicetype_decide with
 * x264_slicetype_analyse (b-adapt 1 and 2, the pre-encode scene cut; R/encoder/slicetype.c:359-636), x264_rc_analyse_slice (:638-680)
 * and the rate control of constant QP and CRF (x264_ratecontrol_new / _start / _mb / _end, rate_estimate_qscale, get_qscale,
 * accum_p_qp_update; R/encoder/ratecontrol.c:268-420, 776-870, 1077-1160, 1168-1195, 1396-1615), in the library's host C with the
 * reference's float / double types expression by expression.  No device involved: the per-frame costs the decisions read
 * (x264_slicetype_frame_cost) are REQUESTED -- x264hip_lookahead_get returns X264HIP_LOOK_NEED with the tasks whose results it
 * lacks; the caller computes them (x264hip_lookahead_cost_frames on the GPU for all its chains at once), hands the results back with
 * x264hip_lookahead_set_cost and calls get again.  A cost is a pure function of (b, p0, p1) and the pictures, so the order in which
 * they are computed does not change any of them; get restarts its decision from the unchanged queue every time.
 * Not here: 2-pass, ABR, VBV, zones, B-pyramid.  With param.b_pre_scenecut = 0 and a threshold >= 0 the queue decides without scene cuts, as the
 * reference's does; the look x264_encoder_encode then takes at every coded P frame is the caller's (x264hip_stream.h: x264hip_frame_stats +
 * x264hip_scenecut_post), and what follows a hit -- the picture coded again as I / IDR, or the B picture before it as the P, frames put back into
 * the queue -- is x264hip_lookahead_scenecut below. */
typedef struct x264hip_lookahead x264hip_lookahead;
typedef struct {
    int mb_w, mb_h;
    int bframes, b_adapt, bframe_bias;           /* param.i_bframe, i_bframe_adaptive (0 none, 1 fast, 2 trellis), i_bframe_bias */
    int keyint_max, keyint_min;                  /* as x264_validate_parameters leaves them (min: clip(min ? min : max / 10 ..., 1, max / 2 + 1)) */
    int scenecut_threshold, pre_scenecut;        /* param.i_scenecut_threshold (< 0: off), b_pre_scenecut */
    int rc_method;                               /* 0: X264_RC_CQP, 1: X264_RC_CRF */
    int qp_constant;                             /* param.rc.i_qp_constant */
    float rf_constant, ip_factor, pb_factor, qcompress;   /* param.rc.f_rf_constant, f_ip_factor (1.4), f_pb_factor (1.3), f_qcompress (0.6) */
    int qp_min, qp_max, qp_step;                 /* param.rc.i_qp_min (10), i_qp_max (51), i_qp_step (4) */
} x264hip_lookahead_params;
/* one x264_slicetype_frame_cost to compute: frame numbers in input order (p0 == p1 == b: the intra cost of b alone) */
typedef struct {
    int b, p0, p1;
    int do_search[2];        /* frames[b]->lowres_mvs[l][dist - 1] carries the "not searched" marker: search list l and keep the vectors */
    int speculative;         /* 1: not asked for yet, but independent of everything else in this batch and likely to be (b-adapt 1's next costs) */
} x264hip_look_need;
/* the frame x264_encoder_encode would code now */
typedef struct {
    int frame;               /* input number (fenc->i_frame) */
    int type;                /* X264_TYPE_IDR 1, I 2, P 3, B 5 (R/x264.h:116-121) */
    int poc, kept_as_ref;
    int qp;                  /* rc->qp = h->sh.i_qp */
    float f_qpm;             /* rc->f_qpm: the QP before rounding, what x264_adaptive_quant adds its offset to */
    int ref0_frame, ref1_frame;          /* fref0[0]->i_frame, fref1[0]->i_frame or -1 */
    int lowres_l0, lowres_l1;            /* 1: fenc->lowres_mvs[0][frame - ref0_frame - 1] / [1][ref1_frame - frame - 1] was searched -- the vectors
                                          * x264_mb_predict_mv_ref16x16 offers the 16x16 search (R/common/macroblock.c:393-398) */
    int i_satd;              /* fdec->i_satd (x264_rc_analyse_slice) or 0 */
    int frame_num_reset;     /* 1: a scene-cut IDR (x264hip_lookahead_scenecut): x264_encoder_encode restarts h->i_frame_num at 0 for it (encoder.c:1682);
                              * the IDRs of --keyint do not (this version never resets the counter elsewhere) */
} x264hip_look_frame;
enum { X264HIP_LOOK_NONE = 0, X264HIP_LOOK_FRAME = 1, X264HIP_LOOK_NEED = 2, X264HIP_LOOK_END = 3 };
x264hip_lookahead *x264hip_lookahead_new(const x264hip_lookahead_params *p);
void x264hip_lookahead_delete(x264hip_lookahead *la);
/* a picture enters frames.next (encoder.c:1404-1421); returns its input number.  Its lowres planes and intra costs are the caller's
 * (x264hip_lowres_init_frame, x264hip_lookahead_intra_frame). */
int x264hip_lookahead_put(x264hip_lookahead *la);
/* the rest of x264_encoder_encode up to the slice: NONE = the B buffer is still filling (call put again); FRAME = *out is to be coded
 * now; NEED = n_need tasks in need[] first (at most max_need, >= 1); END = flushing and nothing left.  flushing: no more input. */
int x264hip_lookahead_get(x264hip_lookahead *la, int flushing, x264hip_look_frame *out, x264hip_look_need *need, int max_need, int *n_need);
/* a computed task: frame->i_cost_est[b - p0][p1 - b] = score, i_intra_mbs[b - p0], i_cost_est[0][0] (the latter two when b == p1).
 * speculative (as the need said): kept aside until the decision asks for it -- a cost the reference never computes must not make its
 * vectors visible to the main encode (lowres_l0 / lowres_l1 of x264hip_look_frame). */
void x264hip_lookahead_set_cost(x264hip_lookahead *la, int b, int p0, int p1, int score, int intra_mbs, int cost00, int speculative);
/* x264_ratecontrol_end + the next call's x264_reference_update for the frame get returned */
void x264hip_lookahead_end(x264hip_lookahead *la);
/* The post-encode scene cut (param.b_pre_scenecut = 0; R/encoder/encoder.c:1603-1699), INSTEAD of x264hip_lookahead_end, when the caller found the P
 * picture it just coded no better than an intra picture (x264hip_stream.h: x264hip_frame_stats + x264hip_scenecut_post): the attempt is given up
 * (its reconstruction, its payload, its place in the DPB: the caller's to discard) and the next x264hip_lookahead_get hands out what is coded instead
 * -- the same picture as I / IDR, or, with B pictures waiting before it, the last of them as the P.  Returns 1 (same picture), 2 (another), -1 (no P
 * picture in flight).  The rate control keeps what x264_ratecontrol_start did for the given-up attempt, as the reference's does. */
int x264hip_lookahead_scenecut(x264hip_lookahead *la);
/* Running x264hip_lookahead_end / _put / _get for the NEXT frame ahead of that verdict (beside the sweep whose P picture is being judged) is possible with
 * a copy of the state taken before them: x264hip_lookahead_save (0, or -1 if more pictures are queued than the copy holds: do not run ahead then) into
 * x264hip_lookahead_state_bytes() bytes, and, if the verdict is "give up", x264hip_lookahead_restore -- back to the frame in flight, the pictures that came
 * in meanwhile queued again -- followed by x264hip_lookahead_scenecut.  Costs computed for the abandoned decisions are dropped with them (they would be
 * computed again, with the same result, if asked for). */
size_t x264hip_lookahead_state_bytes(void);
int x264hip_lookahead_save(const x264hip_lookahead *la, void *buf);
void x264hip_lookahead_restore(x264hip_lookahead *la, const void *buf);
/* frames the caller may drop now: every input number < the returned one is neither queued, nor last_nonb, nor a reference */
int x264hip_lookahead_oldest_live(const x264hip_lookahead *la);

/* The device half: x264_slicetype_frame_cost (with x264_slicetype_mb_cost, R/encoder/slicetype.c:43-345) for many chains at once, one
 * task per wavefront.  A slot is one input frame of every chain of the context's batch: its picture (lowres planes from
 * x264hip_lowres_init_frame), its intra costs (x264hip_lookahead_intra_frame) and its vectors and their costs for every (list, distance)
 * -- fenc->lowres_mvs / lowres_mv_costs.  mv: int16 [batch][2][bframes + 1][n_mb][2], mv_cost: int32 [batch][2][bframes + 1][n_mb]; zeroed
 * once by the caller (the frame's edge macroblocks are never searched and are read as zero vectors).  A task names its three frames by
 * slot, its chain, the distances d0 = b - p0 and d1 = p1 - b (both 0: intra only) and which lists to search (x264hip_look_need);
 * out_dev[task] = {i_cost_est, i_intra_mbs, i_cost_est[0][0], 0}.  Two tasks of one call must not search the same (chain, frame, list,
 * distance), and a bidirectional task needs frames[p1]'s list-0 vectors over d0 + d1 from an EARLIER call (x264hip_lookahead_get asks
 * in that order).  Asynchronous on the context's stream; staging_host (pinned) and tasks_dev hold n_tasks * x264hip_lookahead_task_bytes()
 * and must not be reused before the stream has passed the call.  Refused: frames of <= 2 macroblock rows / columns, subme < 2, lossless. */
typedef struct {
    const x264hip_picture *pic;
    const int32_t *intra_cost;   /* device [batch][n_mb] */
    int16_t *mv;                 /* device */
    int32_t *mv_cost;            /* device */
} x264hip_look_slot;
typedef struct {
    int chain, slot_b, slot_p0, slot_p1, d0, d1;
    int do_search[2];
} x264hip_look_task;
typedef struct {
    int me_method, me_range;     /* param.analyse.i_me_method (the lookahead uses min(HEX, it)), i_me_range */
    int weighted_bipred;         /* param.analyse.b_weighted_bipred */
    int bframes, bframe_bias;    /* param.i_bframe (the arrays' second dimension is bframes + 1), i_bframe_bias */
    int subme_param, lossless;   /* param.analyse.i_subpel_refine and qp == 0: what mbcmp is (encoder.c:608-618) */
    const int16_t *cost_mv;      /* device: p_cost_mv of lambda 1 (x264_lambda_tab[12], slicetype.c:35), centre at cost_mv[cost_mv_range] */
    int cost_mv_range;
} x264hip_look_params;
int x264hip_lookahead_cost_frames(x264hip_frame_ctx *c, const x264hip_look_slot *slots, int n_slots, const x264hip_look_task *tasks,
                                  int n_tasks, const x264hip_look_params *p, void *staging_host, void *tasks_dev, int32_t *out_dev);
size_t x264hip_lookahead_task_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* X264HIP_H */
