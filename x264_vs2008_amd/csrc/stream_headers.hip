// stream_headers.hip -- host C only: the parameter sets, the version SEI and the slice headers around the payloads (include/x264hip_stream.h).
// The reference writes these in R/encoder/set.c and R/encoder/encoder.c; a host that is not the reference (x264_vs2008_amd/mux.py) needs
// them to turn the sweep's slice_data() bytes into the Annex B stream x264's CLI writes.  Nothing here touches the device.
#include <cstdio>
#include <cstring>
#include <cmath>
#include "internal.h"
#include "../../include/x264hip_stream.h"

using x264hip::set_error;

namespace {

// MSB-first bit writer over a zeroed byte buffer (bs_write / bs_write_ue / bs_write_se / bs_rbsp_trailing, R/common/bs.h)
struct Bits {
    uint8_t *p; int cap; long pos; bool over;
    Bits(uint8_t *d, int c) : p(d), cap(c), pos(0), over(false) { if (c > 0) memset(d, 0, (size_t)c); }
    void put1(unsigned b) { if ((pos >> 3) >= cap) { over = true; return; } if (b & 1) p[pos >> 3] |= (uint8_t)(0x80 >> (pos & 7)); pos++; }
    void put(int n, uint32_t v) { for (int i = n - 1; i >= 0; i--) put1((v >> i) & 1); }
    void ue(uint32_t v) { int n = 0; const uint32_t t = v + 1; while ((t >> (n + 1)) != 0) n++; put(n, 0); put(n + 1, t); }
    void se(int v) { ue(v <= 0 ? (uint32_t)(-2 * v) : (uint32_t)(2 * v - 1)); }
    void trailing() { put1(1); while (pos & 7) put1(0); }
    void align1() { while (pos & 7) put1(1); }
    int bytes() const { return (int)((pos + 7) >> 3); }
};

inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
inline float clip3f(float v, float lo, float hi) { return v < lo ? lo : v > hi ? hi : v; }

// x264_levels (R/encoder/set.c:508-527): level_idc, MaxMBPS, MaxFS, MaxDpb bytes, mv range -- ITU-T H.264 table A-1
struct Level { int idc, mbps, frame_size, dpb, mv_range; };
const Level k_levels[] = {
    {10, 1485, 99, 152064, 64},       {11, 3000, 396, 345600, 128},      {12, 6000, 396, 912384, 128},      {13, 11880, 396, 912384, 128},
    {20, 11880, 396, 912384, 128},    {21, 19800, 792, 1824768, 256},    {22, 20250, 1620, 3110400, 256},   {30, 40500, 1620, 3110400, 256},
    {31, 108000, 3600, 6912000, 512}, {32, 216000, 5120, 7864320, 512},  {40, 245760, 8192, 12582912, 512}, {41, 245760, 8192, 12582912, 512},
    {42, 522240, 8704, 13369344, 512}, {50, 589824, 22080, 42393600, 512}, {51, 983040, 36864, 70778880, 512}, {0, 0, 0, 0, 0}};

enum { PROFILE_BASELINE = 66, PROFILE_MAIN = 77, PROFILE_HIGH = 100, PROFILE_HIGH444_PREDICTIVE = 244 };
enum { RC_CQP = 0, RC_CRF = 1 };

// x264_sps_init's values that do not depend on the level (set.c:77-212)
void sps_derive(x264hip_encoder_params *p)
{
    const bool bypass = p->rc_method == RC_CQP && p->qp_constant == 0;
    p->d_profile_idc = bypass ? PROFILE_HIGH444_PREDICTIVE : (p->transform_8x8 || p->cqm_preset != 0) ? PROFILE_HIGH
                       : (p->cabac || p->bframe > 0) ? PROFILE_MAIN : PROFILE_BASELINE;
    p->d_log2_max_frame_num = 4;
    while ((1 << p->d_log2_max_frame_num) <= p->keyint_max) p->d_log2_max_frame_num++;
    p->d_log2_max_frame_num++;
    p->d_log2_max_poc_lsb = p->d_log2_max_frame_num + 1;
    p->d_mb_width = (p->width + 15) / 16;
    p->d_mb_height = (p->height + 15) / 16;
    p->d_num_reorder_frames = p->bframe_pyramid ? 2 : p->bframe ? 1 : 0;
    const int r = p->frame_reference > 1 + p->d_num_reorder_frames ? p->frame_reference : 1 + p->d_num_reorder_frames;
    p->d_num_ref_frames = r < 16 ? r : 16;
}

// x264_validate_levels (set.c:538-577) for the level `l`: nonzero if a limit is exceeded
int level_exceeded(const x264hip_encoder_params *p, const Level *l)
{
    const int mbs = p->d_mb_width * p->d_mb_height;
    const int dpb = mbs * 384 * p->d_num_ref_frames;
    int ret = 0;
    if (l->frame_size < mbs || l->frame_size * 8 < p->d_mb_width * p->d_mb_width || l->frame_size * 8 < p->d_mb_height * p->d_mb_height) ret = 1;
    if (dpb > l->dpb) ret = 1;
    if (p->mv_range > l->mv_range) ret = 1;
    if (p->fps_den > 0 && (int)((int64_t)mbs * p->fps_num / p->fps_den) > l->mbps) ret = 1;
    return ret;                    // VBV limits: no VBV here (0 passes); interlaced: refused before
}

}  // namespace

extern "C" void x264hip_encoder_params_default(x264hip_encoder_params *p)
{   // x264_param_default, R/common/common.c:43-149
    memset(p, 0, sizeof(*p));
    p->fps_num = 25; p->fps_den = 1; p->level_idc = -1; p->threads = 1;
    p->frame_reference = 1; p->keyint_max = 250; p->keyint_min = 25; p->scenecut_threshold = 40; p->bframe_adaptive = 1;
    p->deblocking_filter = 1; p->cabac = 1;
    p->rc_method = RC_CRF; p->qp_constant = 26; p->qp_min = 10; p->qp_max = 51; p->qp_step = 4; p->ip_factor = 1.4f; p->pb_factor = 1.3f;
    p->aq_mode = 1; p->aq_strength = 1.0f; p->qcompress = 0.6f;
    p->intra = 0x3; p->inter = 0x113; p->direct_mv_pred = 1; p->me_method = 1; p->psy_rd = 1.0f; p->me_range = 16; p->subpel_refine = 6;
    p->chroma_me = 1; p->mv_range = -1; p->fast_pskip = 1; p->dct_decimate = 1; p->luma_deadzone[0] = 21; p->luma_deadzone[1] = 11;
}

extern "C" int x264hip_validate_parameters(x264hip_encoder_params *p)
{   // x264_validate_parameters, R/encoder/encoder.c:335-606, in its order
    p->d_valid = 0;
    if (p->width <= 0 || p->height <= 0) { set_error("invalid width x height (%dx%d)", p->width, p->height); return -1; }
    if (p->width % 2 || p->height % 2) { set_error("width or height not divisible by 2 (%dx%d)", p->width, p->height); return -1; }
    if (p->threads != 1) { set_error("threads = %d: the stream writer follows --threads 1 (more threads change the stream: forced pre-scenecut, mv range per thread)", p->threads); return -1; }
    if (p->interlaced) { set_error("interlaced: not built"); return -1; }
    if (p->rc_method != RC_CQP && p->rc_method != RC_CRF) { set_error("rate control %d: constant QP (0) and CRF (1) are built", p->rc_method); return -1; }
    p->rf_constant = clip3f(p->rf_constant, 0, 51);
    p->qp_constant = clip3(p->qp_constant, 0, 51);
    if (p->rc_method == RC_CRF) p->qp_constant = (int)p->rf_constant;
    p->d_lossless = 0;
    if (p->qp_constant == 0) {
        p->d_lossless = 1;
        p->cqm_preset = 0; p->rc_method = RC_CQP; p->ip_factor = 1; p->pb_factor = 1; p->chroma_qp_offset = 0; p->trellis = 0; p->fast_pskip = 0;
        p->noise_reduction = 0; p->psy_rd = 0; p->bframe = 0;
        if (!p->cabac) p->transform_8x8 = 0;
    }
    if (p->rc_method == RC_CQP) {
        const float qp_p = (float)p->qp_constant;
        const float qp_i = (float)(qp_p - 6 * log((double)p->ip_factor) / log((double)2));
        const float qp_b = (float)(qp_p + 6 * log((double)p->pb_factor) / log((double)2));
        const float lo = qp_p < qp_i ? (qp_p < qp_b ? qp_p : qp_b) : (qp_i < qp_b ? qp_i : qp_b);
        const float hi = qp_p > qp_i ? (qp_p > qp_b ? qp_p : qp_b) : (qp_i > qp_b ? qp_i : qp_b);
        p->qp_min = clip3((int)lo, 0, 51);
        p->qp_max = clip3((int)(hi + .999), 0, 51);
        p->aq_mode = 0;
    }
    p->qp_max = clip3(p->qp_max, 0, 51);
    p->qp_min = clip3(p->qp_min, 0, p->qp_max);
    p->frame_reference = clip3(p->frame_reference, 1, 16);
    if (p->keyint_max <= 0) p->keyint_max = 1;
    p->keyint_min = clip3(p->keyint_min, 1, p->keyint_max / 2 + 1);
    if (!p->subpel_refine && p->direct_mv_pred > 1) p->direct_mv_pred = 1;
    p->bframe = clip3(p->bframe, 0, 16);
    p->bframe_bias = clip3(p->bframe_bias, -90, 100);
    p->bframe_pyramid = p->bframe_pyramid && p->bframe > 1;
    if (!p->bframe) p->bframe_adaptive = 0;
    p->weighted_bipred = p->weighted_bipred && p->bframe > 0;
    if (p->scenecut_threshold < 0) p->pre_scenecut = 0;
    p->deblocking_filter_alphac0 = clip3(p->deblocking_filter_alphac0, -6, 6);
    p->deblocking_filter_beta = clip3(p->deblocking_filter_beta, -6, 6);
    p->luma_deadzone[0] = clip3(p->luma_deadzone[0], 0, 32);
    p->luma_deadzone[1] = clip3(p->luma_deadzone[1], 0, 32);
    p->cabac_init_idc = clip3(p->cabac_init_idc, 0, 2);
    if (p->cqm_preset < 0 || p->cqm_preset > 2) p->cqm_preset = 0;
    if (p->me_method < 0 || p->me_method > 4) p->me_method = 1;
    if (p->me_range < 4) p->me_range = 4;
    if (p->me_range > 16 && p->me_method <= 1) p->me_range = 16;
    if (p->me_method == 4 && (p->d_lossless || p->subpel_refine <= 1)) p->me_method = 3;
    p->subpel_refine = clip3(p->subpel_refine, 0, 9);
    p->mixed_references = p->mixed_references && p->frame_reference > 1;
    p->inter &= 0x10 | 0x20 | 0x100 | 0x1 | 0x2;
    p->intra &= 0x1 | 0x2;
    if (!(p->inter & 0x10)) p->inter &= ~0x20u;
    if (!p->transform_8x8) { p->inter &= ~0x2u; p->intra &= ~0x2u; }
    p->chroma_qp_offset = clip3(p->chroma_qp_offset, -12, 12);
    if (!p->cabac) p->trellis = 0;
    p->trellis = clip3(p->trellis, 0, 2);
    if (!p->trellis) p->psy_trellis = 0;
    p->psy_rd = clip3f(p->psy_rd, 0, 10);
    p->psy_trellis = clip3f(p->psy_trellis, 0, 10);
    if (p->subpel_refine < 6) p->psy_rd = 0;
    p->d_psy_rd_fix8 = (int)(p->psy_rd * (1 << 8) + .5);
    if (p->d_psy_rd_fix8) p->chroma_qp_offset -= p->psy_rd < 0.25 ? 1 : 2;
    if ((int)(p->psy_trellis / 4 * (1 << 8) + .5)) p->chroma_qp_offset -= p->psy_trellis < 0.25 ? 1 : 2;
    p->chroma_qp_offset = clip3(p->chroma_qp_offset, -12, 12);
    p->aq_mode = clip3(p->aq_mode, 0, 1);
    p->aq_strength = clip3f(p->aq_strength, 0, 3);
    if (p->aq_strength == 0) p->aq_mode = 0;
    p->noise_reduction = clip3(p->noise_reduction, 0, 1 << 16);
    if (p->bframe_pyramid) { set_error("b-pyramid: not built"); return -1; }
    if (p->cqm_preset == 2) { set_error("custom quantiser matrices in the PPS: not built in the stream writer (flat and jvt are)"); return -1; }
    sps_derive(p);
    const Level *l = k_levels;
    if (p->level_idc < 0) {
        do p->level_idc = l->idc; while (l[1].idc && level_exceeded(p, l) && l++);
    } else {
        while (l->idc && l->idc != p->level_idc) l++;
        if (!l->idc) { set_error("invalid level_idc: %d", p->level_idc); return -1; }
    }
    if (p->mv_range <= 0) p->mv_range = l->mv_range;
    else p->mv_range = clip3(p->mv_range, 32, 512);
    p->cabac = !!p->cabac; p->deblocking_filter = !!p->deblocking_filter; p->transform_8x8 = !!p->transform_8x8; p->chroma_me = !!p->chroma_me;
    p->fast_pskip = !!p->fast_pskip;
    // x264_encoder_open: x264_reduce_fraction on the frame rate, then x264_sps_init / x264_pps_init (encoder.c:683-694)
    if (p->fps_num && p->fps_den) {
        int a = p->fps_num, b = p->fps_den, c = a % b;
        while (c) { a = b; b = c; c = a % b; }
        p->fps_num /= b; p->fps_den /= b;
    }
    sps_derive(p);
    int v = p->mv_range * 4 - 1, n = 0;               // (int)(log(v) / log(2)) + 1 for an odd v: its bit length
    while (v) { n++; v >>= 1; }
    p->d_log2_max_mv_length = n;
    p->d_pic_init_qp = p->qp_constant;                // ABR would be 26 (set.c:384)
    p->d_valid = 1;
    return 0;
}

extern "C" int x264hip_param2string(const x264hip_encoder_params *p, char *dst, int cap)
{   // x264_param2string( p, 0 ), R/common/common.c:816-909
    static const char *const me_names[] = {"dia", "hex", "umh", "esa", "tesa"};
    char buf[1200], *s = buf;
    s += sprintf(s, "cabac=%d", p->cabac);
    s += sprintf(s, " ref=%d", p->frame_reference);
    s += sprintf(s, " deblock=%d:%d:%d", p->deblocking_filter, p->deblocking_filter_alphac0, p->deblocking_filter_beta);
    s += sprintf(s, " analyse=%#x:%#x", p->intra, p->inter);
    s += sprintf(s, " me=%s", me_names[clip3(p->me_method, 0, 4)]);
    s += sprintf(s, " subme=%d", p->subpel_refine);
    s += sprintf(s, " psy_rd=%.1f:%.1f", p->psy_rd, p->psy_trellis);
    s += sprintf(s, " mixed_ref=%d", p->mixed_references);
    s += sprintf(s, " me_range=%d", p->me_range);
    s += sprintf(s, " chroma_me=%d", p->chroma_me);
    s += sprintf(s, " trellis=%d", p->trellis);
    s += sprintf(s, " 8x8dct=%d", p->transform_8x8);
    s += sprintf(s, " cqm=%d", p->cqm_preset);
    s += sprintf(s, " deadzone=%d,%d", p->luma_deadzone[0], p->luma_deadzone[1]);
    s += sprintf(s, " chroma_qp_offset=%d", p->chroma_qp_offset);
    s += sprintf(s, " threads=%d", p->threads);
    s += sprintf(s, " nr=%d", p->noise_reduction);
    s += sprintf(s, " decimate=%d", p->dct_decimate);
    s += sprintf(s, " mbaff=%d", p->interlaced);
    s += sprintf(s, " bframes=%d", p->bframe);
    if (p->bframe)
        s += sprintf(s, " b_pyramid=%d b_adapt=%d b_bias=%d direct=%d wpredb=%d", p->bframe_pyramid, p->bframe_adaptive, p->bframe_bias,
                     p->direct_mv_pred, p->weighted_bipred);
    s += sprintf(s, " keyint=%d keyint_min=%d scenecut=%d%s", p->keyint_max, p->keyint_min, p->scenecut_threshold, p->pre_scenecut ? "(pre)" : "");
    s += sprintf(s, " rc=%s", p->rc_method == RC_CRF ? "crf" : "cqp");
    if (p->rc_method == RC_CRF) {
        s += sprintf(s, " crf=%.1f", p->rf_constant);
        s += sprintf(s, " qcomp=%.2f qpmin=%d qpmax=%d qpstep=%d", p->qcompress, p->qp_min, p->qp_max, p->qp_step);
    } else
        s += sprintf(s, " qp=%d", p->qp_constant);
    if (!(p->rc_method == RC_CQP && p->qp_constant == 0)) {
        s += sprintf(s, " ip_ratio=%.2f", p->ip_factor);
        if (p->bframe) s += sprintf(s, " pb_ratio=%.2f", p->pb_factor);
        s += sprintf(s, " aq=%d", p->aq_mode);
        if (p->aq_mode) s += sprintf(s, ":%.2f", p->aq_strength);
    }
    const int n = (int)(s - buf);
    if (n + 1 > cap) { set_error("param2string: %d bytes needed", n + 1); return -1; }
    memcpy(dst, buf, (size_t)n + 1);
    return n;
}

#define NEED_VALID(p_, what_) do { if (!(p_)->d_valid) { set_error(what_ ": call x264hip_validate_parameters first"); return -1; } } while (0)

extern "C" int x264hip_sps_write(const x264hip_encoder_params *p, uint8_t *dst, int cap)
{   // x264_sps_write, R/encoder/set.c:215-365, with x264_sps_init's constants (:77-212) in place
    NEED_VALID(p, "sps_write");
    Bits s(dst, cap);
    const int prof = p->d_profile_idc;
    s.put(8, (uint32_t)prof);
    s.put1(prof == PROFILE_BASELINE);                       // constraint_set0
    s.put1(prof <= PROFILE_MAIN);                           // constraint_set1
    s.put1(0);                                              // constraint_set2: never set
    s.put(5, 0);
    s.put(8, (uint32_t)p->level_idc);
    s.ue(0);                                                // sps id
    if (prof >= PROFILE_HIGH) {
        s.ue(1); s.ue(0); s.ue(0);                          // 4:2:0, 8 bits luma, 8 bits chroma
        s.put1(prof == PROFILE_HIGH444_PREDICTIVE);         // qpprime_y_zero_transform_bypass
        s.put1(0);                                          // seq_scaling_matrix_present_flag
    }
    s.ue((uint32_t)(p->d_log2_max_frame_num - 4));
    s.ue(0);                                                // poc type 0
    s.ue((uint32_t)(p->d_log2_max_poc_lsb - 4));
    s.ue((uint32_t)p->d_num_ref_frames);
    s.put1(0);                                              // gaps_in_frame_num_value_allowed
    s.ue((uint32_t)(p->d_mb_width - 1));
    s.ue((uint32_t)(p->d_mb_height - 1));
    s.put1(1);                                              // frame_mbs_only
    s.put1(1);                                              // direct8x8_inference
    const int crop_r = p->d_mb_width * 16 - p->width, crop_b = p->d_mb_height * 16 - p->height;
    s.put1(crop_r || crop_b);
    if (crop_r || crop_b) { s.ue(0); s.ue((uint32_t)(crop_r / 2)); s.ue(0); s.ue((uint32_t)(crop_b / 2)); }
    s.put1(1);                                              // vui
    s.put1(0);                                              // aspect_ratio_info_present (no SAR here)
    s.put1(0);                                              // overscan_info_present
    s.put1(0);                                              // video_signal_type_present (all "undef")
    s.put1(0);                                              // chroma_loc_info_present
    const int timing = p->fps_num > 0 && p->fps_den > 0;
    s.put1((unsigned)timing);
    if (timing) { s.put(32, (uint32_t)p->fps_den); s.put(32, (uint32_t)p->fps_num * 2u); s.put1(1); }
    s.put1(0); s.put1(0); s.put1(0);                        // nal hrd, vcl hrd, pic_struct
    s.put1(1);                                              // bitstream_restriction
    s.put1(1);                                              // motion_vectors_over_pic_boundaries
    s.ue(0); s.ue(0);                                       // max_bytes_per_pic_denom, max_bits_per_mb_denom
    s.ue((uint32_t)p->d_log2_max_mv_length); s.ue((uint32_t)p->d_log2_max_mv_length);
    s.ue((uint32_t)p->d_num_reorder_frames);
    s.ue((uint32_t)p->d_num_ref_frames);                    // max_dec_frame_buffering
    s.trailing();
    if (s.over) { set_error("sps_write: buffer too small"); return -1; }
    return s.bytes();
}

extern "C" int x264hip_pps_write(const x264hip_encoder_params *p, uint8_t *dst, int cap)
{   // x264_pps_write, set.c:433-474, with x264_pps_init's constants (:367-431)
    NEED_VALID(p, "pps_write");
    Bits s(dst, cap);
    s.ue(0); s.ue(0);                                       // pps id, sps id
    s.put1((unsigned)p->cabac);
    s.put1(0);                                              // pic_order_present
    s.ue(0);                                                // one slice group
    s.ue(0); s.ue(0);                                       // num_ref_idx_l0 / l1 default active - 1
    s.put1(0);                                              // weighted_pred
    s.put(2, p->weighted_bipred ? 2u : 0u);
    s.se(p->d_pic_init_qp - 26);
    s.se(0);                                                // pic_init_qs
    s.se(p->chroma_qp_offset);
    s.put1(1);                                              // deblocking_filter_control_present
    s.put1(0);                                              // constrained_intra_pred
    s.put1(0);                                              // redundant_pic_cnt_present
    if (p->transform_8x8 || p->cqm_preset != 0) {
        s.put1((unsigned)p->transform_8x8);
        s.put1(p->cqm_preset != 0);
        if (p->cqm_preset != 0) {
            // the jvt preset: every list equals its fall-back (scaling_list_write, set.c:42-75), one zero flag each; Cr follows Cb
            s.put1(0); s.put1(0); s.put1(0); s.put1(0); s.put1(0); s.put1(0);
            if (p->transform_8x8) { s.put1(0); s.put1(0); }
        }
        s.se(p->chroma_qp_offset);                          // second_chroma_qp_index_offset
    }
    s.trailing();
    if (s.over) { set_error("pps_write: buffer too small"); return -1; }
    return s.bytes();
}

extern "C" int x264hip_sei_version_write(const x264hip_encoder_params *p, uint8_t *dst, int cap)
{   // x264_sei_version_write, set.c:476-506; X264_VERSION is "" for a source tree that is not a git checkout (R/version.sh)
    NEED_VALID(p, "sei_version_write");
    static const uint8_t uuid[16] = {0xdc, 0x45, 0xe9, 0xbd, 0xe6, 0xd9, 0x48, 0xb7, 0x96, 0x2c, 0xd8, 0x20, 0xd9, 0x23, 0xee, 0xef};
    char opts[1200], version[1500];
    if (x264hip_param2string(p, opts, (int)sizeof(opts)) < 0) return -1;
    snprintf(version, sizeof(version), "x264 - core %d%s - H.264/MPEG-4 AVC codec - Copyleft 2003-2008 - http://www.videolan.org/x264.html - options: %s",
             66, "", opts);
    const int length = (int)strlen(version) + 1 + 16;
    Bits s(dst, cap);
    s.put(8, 5);                                            // user_data_unregistered
    int i;
    for (i = 0; i <= length - 255; i += 255) s.put(8, 255);
    s.put(8, (uint32_t)(length - i));
    for (i = 0; i < 16; i++) s.put(8, uuid[i]);
    for (i = 0; i < length - 16; i++) s.put(8, (uint8_t)version[i]);
    s.trailing();
    if (s.over) { set_error("sei_version_write: buffer too small"); return -1; }
    return s.bytes();
}

extern "C" int x264hip_slice_nal(const x264hip_encoder_params *p, const x264hip_slice_header *sh, const uint8_t *payload, int payload_len,
                                 uint8_t *dst, int cap)
{
    NEED_VALID(p, "slice_nal");
    if (payload_len < 0 || (payload_len && !payload)) { set_error("slice_nal: no payload"); return -1; }
    if (cap < payload_len + payload_len / 2 + 64) { set_error("slice_nal: dst needs payload_len * 3 / 2 + 64 bytes"); return -1; }
    const int is_idr = sh->nal_type == 5, st = sh->slice_type;
    if (st < 0 || st > 2 || (sh->nal_type != 1 && sh->nal_type != 5)) { set_error("slice_nal: slice type %d / nal type %d", st, sh->nal_type); return -1; }
    uint8_t *rbsp = (uint8_t *)malloc((size_t)payload_len + 64);
    if (!rbsp) { set_error("slice_nal: out of memory"); return -1; }
    Bits s(rbsp, payload_len + 64);
    // ---- x264_slice_header_write (encoder.c:168-299) on x264_slice_header_init / x264_slice_init's values (:81-155, :1102-1140) ----
    s.ue(0);                                                // first_mb
    s.ue((uint32_t)(st + 5));                               // "same type things"
    s.ue(0);                                                // pps id
    s.put(p->d_log2_max_frame_num, (uint32_t)sh->frame_num & ((1u << p->d_log2_max_frame_num) - 1));
    if (is_idr) s.ue((uint32_t)sh->idr_pic_id);
    s.put(p->d_log2_max_poc_lsb, (uint32_t)sh->poc & ((1u << p->d_log2_max_poc_lsb) - 1));
    if (st == 1) s.put1((unsigned)(sh->direct_spatial != 0));
    const int n0 = sh->n_ref0 <= 0 ? 1 : sh->n_ref0, n1 = sh->n_ref1 <= 0 ? 1 : sh->n_ref1;
    if (st != 2) {
        s.put1(1);                                          // num_ref_idx_override: "always set the real higher num of ref frame used"
        s.ue((uint32_t)(n0 - 1));
        if (st == 1) s.ue((uint32_t)(n1 - 1));
    }
    if (st != 2) {
        int reorder = 0;                                    // x264_reference_build_list's check (encoder.c:961-972): P slices only
        if (st == 0) for (int i = 0; i < sh->n_ref0 - 1; i++) if (sh->ref_frame_num[i] < sh->ref_frame_num[i + 1]) { reorder = 1; break; }
        s.put1((unsigned)reorder);
        if (reorder) {
            int pred = sh->frame_num;
            for (int i = 0; i < n0; i++) {
                const int diff = sh->ref_frame_num[i] - pred;
                s.ue(diff > 0); s.ue((uint32_t)((diff < 0 ? -diff : diff) - 1));
                pred = sh->ref_frame_num[i];
            }
            s.ue(3);
        }
    }
    if (st == 1) s.put1(0);                                 // list 1 is never reordered
    if (sh->nal_ref_idc != 0) {
        if (is_idr) { s.put1(0); s.put1(0); }               // no_output_of_prior_pics, long_term_reference
        else s.put1(0);                                     // adaptive_ref_pic_marking_mode
    }
    if (p->cabac && st != 2) s.ue((uint32_t)p->cabac_init_idc);
    s.se(sh->qp - p->d_pic_init_qp);
    {   // deblocking_filter_control_present is always 1; "if effective qp <= 15, deblocking would have no effect anyway"
        const int mn = p->deblocking_filter_alphac0 < p->deblocking_filter_beta ? p->deblocking_filter_alphac0 : p->deblocking_filter_beta;
        const int on = p->deblocking_filter && (p->aq_mode /* h->mb.b_variable_qp */ || 15 < sh->qp + 2 * mn);
        s.ue(on ? 0u : 1u);
        if (on) { s.se(p->deblocking_filter_alphac0); s.se(p->deblocking_filter_beta); }
    }
    // ---- slice_data(), x264_slice_write (encoder.c:1151-1282) ----
    if (p->cabac) {
        s.align1();
        memcpy(rbsp + (s.pos >> 3), payload, (size_t)payload_len);
        s.pos += (long)payload_len * 8;
    } else {
        // the CAVLC pass wrote slice_data() from bit 0 of its buffer and closed it with rbsp_trailing_bits: here the bits go on behind the
        // header's last bit, as in the reference's one bitstream, and the trailing bits are written again
        int nbits = 0;
        if (payload_len > 0) {
            const uint8_t last = payload[payload_len - 1];
            if (!last) { free(rbsp); set_error("slice_nal: the CAVLC payload does not end with its stop bit"); return -1; }
            int tz = 0;
            while (!((last >> tz) & 1)) tz++;
            nbits = 8 * (payload_len - 1) + 7 - tz;
        }
        const int sh_bits = (int)(s.pos & 7);
        if (!sh_bits) { memcpy(rbsp + (s.pos >> 3), payload, (size_t)((nbits + 7) >> 3)); }
        else {
            uint8_t *d = rbsp + (s.pos >> 3);
            for (int i = 0; i < (nbits + 7) >> 3; i++) { d[i] |= (uint8_t)(payload[i] >> sh_bits); d[i + 1] = (uint8_t)(payload[i] << (8 - sh_bits)); }
        }
        s.pos += nbits;
        {   // clear what the copy left behind the last data bit (the old stop bit and padding), then close
            const long end = s.pos;
            uint8_t *d = rbsp + (end >> 3);
            if (end & 7) { *d &= (uint8_t)(0xff00 >> (end & 7)); d[1] = 0; } else { d[0] = 0; }
        }
        s.trailing();
    }
    if (s.over) { free(rbsp); set_error("slice_nal: header does not fit"); return -1; }
    const int n = x264hip_nal_encode(dst, 1, sh->nal_ref_idc, sh->nal_type, rbsp, s.bytes());
    free(rbsp);
    return n;
}
