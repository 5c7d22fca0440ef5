/* x264hip_tables.h -- the six DSP function-pointer tables of x264 core 66,
 * restated field-for-field so that a table filled by this library can be
 * struct-copied over the one inside x264_t (R/common/common.h:618-630).
 *
 * R/ = x264-snapshot-20090216-2245/ of chinaxuyongtao/x264-vs2008.
 * Every struct below cites the reference typedef whose memory layout it
 * must match.  The layout IS the ABI: field order, array lengths and
 * argument lists may not change.
 */
#ifndef X264HIP_TABLES_H
#define X264HIP_TABLES_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scratch strides of the per-macroblock buffers (R/common/common.h:464-465). */
#define X264HIP_FENC_STRIDE 16
#define X264HIP_FDEC_STRIDE 32

/* Block-size enum (R/common/pixel.h:30-42). */
enum {
    X264HIP_PIXEL_16x16 = 0,
    X264HIP_PIXEL_16x8  = 1,
    X264HIP_PIXEL_8x16  = 2,
    X264HIP_PIXEL_8x8   = 3,
    X264HIP_PIXEL_8x4   = 4,
    X264HIP_PIXEL_4x8   = 5,
    X264HIP_PIXEL_4x4   = 6,
    X264HIP_PIXEL_4x2   = 7,
    X264HIP_PIXEL_2x4   = 8,
    X264HIP_PIXEL_2x2   = 9
};

/* ---- pixel metrics: R/common/pixel.h:26-28 and :63-103 ---------------- */
typedef int  (*x264hip_pixel_cmp_t)   (uint8_t *, int, uint8_t *, int);
typedef void (*x264hip_pixel_cmp_x3_t)(uint8_t *, uint8_t *, uint8_t *, uint8_t *, int, int[3]);
typedef void (*x264hip_pixel_cmp_x4_t)(uint8_t *, uint8_t *, uint8_t *, uint8_t *, uint8_t *, int, int[4]);

typedef struct {
    x264hip_pixel_cmp_t    sad[7];
    x264hip_pixel_cmp_t    ssd[7];
    x264hip_pixel_cmp_t    satd[7];
    x264hip_pixel_cmp_t    ssim[7];
    x264hip_pixel_cmp_t    sa8d[4];
    x264hip_pixel_cmp_t    mbcmp[7];
    x264hip_pixel_cmp_t    mbcmp_unaligned[7];
    x264hip_pixel_cmp_t    fpelcmp[7];
    x264hip_pixel_cmp_x3_t fpelcmp_x3[7];
    x264hip_pixel_cmp_x4_t fpelcmp_x4[7];
    x264hip_pixel_cmp_t    sad_aligned[7];

    int      (*var[4])(uint8_t *pix, int stride);
    uint64_t (*hadamard_ac[4])(uint8_t *pix, int stride);

    void  (*ssim_4x4x2_core)(const uint8_t *pix1, int stride1,
                             const uint8_t *pix2, int stride2, int sums[2][4]);
    float (*ssim_end4)(int sum0[5][4], int sum1[5][4], int width);

    x264hip_pixel_cmp_x3_t sad_x3[7];
    x264hip_pixel_cmp_x4_t sad_x4[7];
    x264hip_pixel_cmp_x3_t satd_x3[7];
    x264hip_pixel_cmp_x4_t satd_x4[7];

    int (*ads[7])(int enc_dc[4], uint16_t *sums, int delta,
                  uint16_t *cost_mvx, int16_t *mvs, int width, int thresh);

    /* fused predict(V,H,DC)+cost; NULL in the reference's C build
     * (callers test for NULL, R/encoder/analyse.c:559,621,673,765). */
    void (*intra_mbcmp_x3_16x16)(uint8_t *fenc, uint8_t *fdec, int res[3]);
    void (*intra_satd_x3_16x16) (uint8_t *fenc, uint8_t *fdec, int res[3]);
    void (*intra_sad_x3_16x16)  (uint8_t *fenc, uint8_t *fdec, int res[3]);
    void (*intra_satd_x3_8x8c)  (uint8_t *fenc, uint8_t *fdec, int res[3]);
    void (*intra_satd_x3_4x4)   (uint8_t *fenc, uint8_t *fdec, int res[3]);
    void (*intra_sa8d_x3_8x8)   (uint8_t *fenc, uint8_t edge[33], int res[3]);
} x264hip_pixel_function_t;

/* ---- transforms: R/common/dct.h:89-124 -------------------------------- */
typedef struct {
    void (*sub4x4_dct)      (int16_t dct[4][4], uint8_t *pix1, uint8_t *pix2);
    void (*add4x4_idct)     (uint8_t *p_dst, int16_t dct[4][4]);
    void (*sub8x8_dct)      (int16_t dct[4][4][4], uint8_t *pix1, uint8_t *pix2);
    void (*add8x8_idct)     (uint8_t *p_dst, int16_t dct[4][4][4]);
    void (*add8x8_idct_dc)  (uint8_t *p_dst, int16_t dct[2][2]);
    void (*sub16x16_dct)    (int16_t dct[16][4][4], uint8_t *pix1, uint8_t *pix2);
    void (*add16x16_idct)   (uint8_t *p_dst, int16_t dct[16][4][4]);
    void (*add16x16_idct_dc)(uint8_t *p_dst, int16_t dct[4][4]);
    void (*sub8x8_dct8)     (int16_t dct[8][8], uint8_t *pix1, uint8_t *pix2);
    void (*add8x8_idct8)    (uint8_t *p_dst, int16_t dct[8][8]);
    void (*sub16x16_dct8)   (int16_t dct[4][8][8], uint8_t *pix1, uint8_t *pix2);
    void (*add16x16_idct8)  (uint8_t *p_dst, int16_t dct[4][8][8]);
    void (*dct4x4dc)        (int16_t d[4][4]);
    void (*idct4x4dc)       (int16_t d[4][4]);
} x264hip_dct_function_t;

typedef struct {
    void (*scan_8x8)(int16_t level[64], int16_t dct[8][8]);
    void (*scan_4x4)(int16_t level[16], int16_t dct[4][4]);
    void (*sub_8x8) (int16_t level[64], const uint8_t *p_src, uint8_t *p_dst);
    void (*sub_4x4) (int16_t level[16], const uint8_t *p_src, uint8_t *p_dst);
    void (*interleave_8x8_cavlc)(int16_t *dst, int16_t *src, uint8_t *nnz);
} x264hip_zigzag_function_t;

/* ---- quantisation: R/common/quant.h:26-44, run/level R/common/bs.h:53-58 */
typedef struct {
    int     last;
    int16_t level[16];
    uint8_t run[16];
} x264hip_run_level_t;

/* order of the coeff_last[] / coeff_level_run[] slots
 * (R/common/macroblock.h block_idx enum: DCT_LUMA_DC..DCT_LUMA_8x8). */
enum {
    X264HIP_DCT_LUMA_DC   = 0,
    X264HIP_DCT_LUMA_AC   = 1,
    X264HIP_DCT_LUMA_4x4  = 2,
    X264HIP_DCT_CHROMA_DC = 3,
    X264HIP_DCT_CHROMA_AC = 4,
    X264HIP_DCT_LUMA_8x8  = 5
};

typedef struct {
    int  (*quant_8x8)   (int16_t dct[8][8], uint16_t mf[64], uint16_t bias[64]);
    int  (*quant_4x4)   (int16_t dct[4][4], uint16_t mf[16], uint16_t bias[16]);
    int  (*quant_4x4_dc)(int16_t dct[4][4], int mf, int bias);
    int  (*quant_2x2_dc)(int16_t dct[2][2], int mf, int bias);
    void (*dequant_8x8)   (int16_t dct[8][8], int dequant_mf[6][8][8], int i_qp);
    void (*dequant_4x4)   (int16_t dct[4][4], int dequant_mf[6][4][4], int i_qp);
    void (*dequant_4x4_dc)(int16_t dct[4][4], int dequant_mf[6][4][4], int i_qp);
    void (*denoise_dct)(int16_t *dct, uint32_t *sum, uint16_t *offset, int size);
    int  (*decimate_score15)(int16_t *dct);
    int  (*decimate_score16)(int16_t *dct);
    int  (*decimate_score64)(int16_t *dct);
    int  (*coeff_last[6])(int16_t *dct);
    int  (*coeff_level_run[5])(int16_t *dct, x264hip_run_level_t *runlevel);
} x264hip_quant_function_t;

/* ---- motion compensation + frame filters: R/common/mc.h:31-77 --------- */
typedef struct {
    void     (*mc_luma)(uint8_t *dst, int i_dst, uint8_t **src, int i_src,
                        int mvx, int mvy, int i_width, int i_height);
    uint8_t *(*get_ref)(uint8_t *dst, int *i_dst, uint8_t **src, int i_src,
                        int mvx, int mvy, int i_width, int i_height);
    void     (*mc_chroma)(uint8_t *dst, int i_dst, uint8_t *src, int i_src,
                          int mvx, int mvy, int i_width, int i_height);
    void (*avg[10])(uint8_t *dst, int, uint8_t *src1, int, uint8_t *src2, int, int i_weight);
    void (*copy[7])(uint8_t *dst, int, uint8_t *src, int, int i_height);
    void (*copy_16x16_unaligned)(uint8_t *dst, int, uint8_t *src, int, int i_height);
    void (*plane_copy)(uint8_t *dst, int i_dst, uint8_t *src, int i_src, int w, int h);
    void (*hpel_filter)(uint8_t *dsth, uint8_t *dstv, uint8_t *dstc, uint8_t *src,
                        int i_stride, int i_width, int i_height, int16_t *buf);
    void (*prefetch_fenc)(uint8_t *pix_y, int stride_y, uint8_t *pix_uv, int stride_uv, int mb_x);
    void (*prefetch_ref)(uint8_t *pix, int stride, int parity);
    void *(*memcpy_aligned)(void *dst, const void *src, size_t n);
    void (*memzero_aligned)(void *dst, int n);
    void (*integral_init4h)(uint16_t *sum, uint8_t *pix, int stride);
    void (*integral_init8h)(uint16_t *sum, uint8_t *pix, int stride);
    void (*integral_init4v)(uint16_t *sum8, uint16_t *sum4, int stride);
    void (*integral_init8v)(uint16_t *sum8, int stride);
    void (*frame_init_lowres_core)(uint8_t *src0, uint8_t *dst0, uint8_t *dsth,
                                   uint8_t *dstv, uint8_t *dstc,
                                   int src_stride, int dst_stride, int width, int height);
} x264hip_mc_functions_t;

/* ---- intra prediction: R/common/predict.h:27-29, enums :31-107 -------- */
typedef void (*x264hip_predict_t)(uint8_t *src);
typedef void (*x264hip_predict8x8_t)(uint8_t *src, uint8_t edge[33]);
typedef void (*x264hip_predict_8x8_filter_t)(uint8_t *src, uint8_t edge[33],
                                             int i_neighbor, int i_filters);

/* neighbour-availability bits (R/common/macroblock.h:28-36) */
enum {
    X264HIP_MB_LEFT     = 0x01,
    X264HIP_MB_TOP      = 0x02,
    X264HIP_MB_TOPRIGHT = 0x04,
    X264HIP_MB_TOPLEFT  = 0x08
};

/* ---- deblocking: R/common/frame.h:94-108 ------------------------------ */
typedef void (*x264hip_deblock_inter_t)(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0);
typedef void (*x264hip_deblock_intra_t)(uint8_t *pix, int stride, int alpha, int beta);
typedef struct {
    x264hip_deblock_inter_t deblock_v_luma;
    x264hip_deblock_inter_t deblock_h_luma;
    x264hip_deblock_inter_t deblock_v_chroma;
    x264hip_deblock_inter_t deblock_h_chroma;
    x264hip_deblock_intra_t deblock_v_luma_intra;
    x264hip_deblock_intra_t deblock_h_luma_intra;
    x264hip_deblock_intra_t deblock_v_chroma_intra;
    x264hip_deblock_intra_t deblock_h_chroma_intra;
} x264hip_deblock_function_t;

#ifdef __cplusplus
}
#endif
#endif /* X264HIP_TABLES_H */
