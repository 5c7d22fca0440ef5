/* x264hip_stream.h -- the bytes around slice_data(): what x264_encoder_encode puts in front of and around the payloads the sweep returns,
 * so that a host that is NOT the reference's encoder.c (x264_vs2008_amd/mux.py, a C consumer) can emit the complete Annex B stream.
 * Host C of the library, no device call inside.  A maintainer of the reference keeps encoder.c's own writers and needs none of this.
 *
 *   x264hip_validate_parameters   x264_validate_parameters (R/encoder/encoder.c:335-606), the part that reaches the stream: clamps, the level
 *                                 picked from frame size / DPB / macroblock rate (x264_validate_levels, R/encoder/set.c:538-577), mv_range,
 *                                 the psy-RD shift of the chroma QP offset; then x264_sps_init / x264_pps_init's derived values (set.c:77-212,367-431)
 *   x264hip_param2string          x264_param2string( p, 0 ) (R/common/common.c:816-909)
 *   x264hip_sps_write / _pps_write / _sei_version_write      x264_sps_write, x264_pps_write, x264_sei_version_write (set.c:215-365,433-506), as RBSP bytes
 *   x264hip_slice_nal             x264_slice_header_write (encoder.c:168-299) + bs_align_1 and the CABAC bytes, or the CAVLC bits spliced on
 *                                 behind the header's last bit + bs_rbsp_trailing (encoder.c:1151-1282), through x264_nal_encode
 *
 * Pinning: x264hip_param2string against the reference's own x264_param2string (oracle/_ref); the writers cannot be compared live (R/encoder/set.c and
 * encoder.c need the configure-generated config.h: not buildable here) and are pinned end to end by the md5 of the reference CLI's whole .264 that
 * SURVEY.md 8(c) records for BASELINE's configurations (tests/test_gpu_mux.py).
 */
#ifndef X264HIP_STREAM_H
#define X264HIP_STREAM_H
#include <stdint.h>
#include "x264hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* x264_param_t's fields that reach the stream (names follow R/x264.h:152-295).  Fill like x264_param_default + the command line, then call
 * x264hip_validate_parameters once: it edits the fields as x264_encoder_open does and fills the derived ones (d_*). */
typedef struct x264hip_encoder_params {
    int width, height, fps_num, fps_den;
    int level_idc;                         /* -1: chosen like x264_validate_parameters does */
    int threads;                           /* 1 (printed in the SEI) */
    int frame_reference, keyint_max, keyint_min, scenecut_threshold, pre_scenecut;
    int bframe, bframe_adaptive, bframe_bias, bframe_pyramid;
    int deblocking_filter, deblocking_filter_alphac0, deblocking_filter_beta;
    int cabac, cabac_init_idc, interlaced, cqm_preset;
    unsigned intra, inter;                 /* param.analyse.intra / .inter */
    int transform_8x8, weighted_bipred, direct_mv_pred, chroma_qp_offset;
    int me_method, me_range, mv_range, subpel_refine, chroma_me, mixed_references, trellis, fast_pskip, dct_decimate, noise_reduction;
    float psy_rd, psy_trellis;
    int luma_deadzone[2];
    int rc_method;                         /* X264_RC_CQP 0, X264_RC_CRF 1 (ABR / VBV / 2-pass: not built) */
    int qp_constant, qp_min, qp_max, qp_step;
    float rf_constant, ip_factor, pb_factor, qcompress;
    int aq_mode;
    float aq_strength;
    const uint8_t *scaling_list[6];        /* cqm_preset != 0: the PPS's lists (4iy 4ic 4py 4pc 8iy 8py), zigzag order is applied here */
    /* derived by x264hip_validate_parameters */
    int d_valid, d_lossless, d_profile_idc, d_num_ref_frames, d_num_reorder_frames, d_log2_max_frame_num, d_log2_max_poc_lsb,
        d_mb_width, d_mb_height, d_pic_init_qp, d_log2_max_mv_length, d_psy_rd_fix8;
} x264hip_encoder_params;

void x264hip_encoder_params_default(x264hip_encoder_params *p);            /* x264_param_default's values for the fields above */
int x264hip_validate_parameters(x264hip_encoder_params *p);                /* 0, or -1 + x264hip_last_error() */
int x264hip_param2string(const x264hip_encoder_params *p, char *dst, int cap);      /* length, or -1 */
/* RBSP bytes (no NAL header, no emulation prevention: x264hip_nal_encode adds both); return the length or -1 */
int x264hip_sps_write(const x264hip_encoder_params *p, uint8_t *dst, int cap);
int x264hip_pps_write(const x264hip_encoder_params *p, uint8_t *dst, int cap);
int x264hip_sei_version_write(const x264hip_encoder_params *p, uint8_t *dst, int cap);

typedef struct x264hip_slice_header {
    int nal_type;              /* NAL_SLICE 1, NAL_SLICE_IDR 5 */
    int nal_ref_idc;           /* 3 IDR, 2 I / P, 0 disposable B */
    int slice_type;            /* 0 P, 1 B, 2 I (x264hip_slice_params.slice_type) */
    int frame_num, idr_pic_id /* -1 unless IDR */, poc, qp;
    int n_ref0, n_ref1;        /* h->i_ref0 / i_ref1 */
    int direct_spatial;        /* B: sh.b_direct_spatial_mv_pred */
    int ref_frame_num[16];     /* P: frame_num of list 0's pictures, for x264_reference_build_list's reorder check (encoder.c:961-972) */
} x264hip_slice_header;

/* One slice NAL, Annex B start code included: header + payload (CABAC: the bytes of x264hip_slice_rd.payload; CAVLC: the bits
 * x264hip_cavlc_write_frame wrote, trailing bits included -- they are re-aligned behind the header).  dst: payload_len * 3 / 2 + 64 bytes.
 * Returns the NAL's length or -1. */
int x264hip_slice_nal(const x264hip_encoder_params *p, const x264hip_slice_header *sh, const uint8_t *payload, int payload_len,
                      uint8_t *dst, int cap);

/* h->stat.frame's terms the post-encode scene cut of x264_encoder_encode reads after a P slice (R/encoder/encoder.c:1603-1644), per chain, from
 * the state the sweep left: out_dev [batch] records on the device (stream-ordered).  x264hip_scenecut_post is the decision (host C, the reference's
 * float expression): 1 = the reference gives this attempt up and codes again -- the picture as I / IDR, or the B picture before it as the P: the host
 * discards the attempt (reconstruction, payload, its place in the DPB), calls x264hip_lookahead_scenecut (x264hip.h) instead of x264hip_lookahead_end and
 * codes what x264hip_lookahead_get hands out next (x264_vs2008_amd/stream.py: StreamEncoder.step does it inside the step, for the chains concerned). */
typedef struct x264hip_frame_stat { int64_t intra_cost, inter_cost; int32_t mbs_analysed, mb_i, mb_p, mb_skip; } x264hip_frame_stat;
int x264hip_frame_stats(x264hip_frame_ctx *c, const x264hip_mb_state *st, x264hip_frame_stat *out_dev);
int x264hip_scenecut_post(const x264hip_frame_stat *s, int i_mb, int i_gop_size, int scenecut_threshold, int keyint_min, int keyint_max);

#ifdef __cplusplus
}
#endif
#endif
