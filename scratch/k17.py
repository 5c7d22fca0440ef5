import sys,re
p='/root/repo/include/x264hip.h'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep('''    int poc, n_ref0, inv_ref_poc[8];                   /* x264_frame_t.i_poc / i_ref[0] / inv_ref_poc, filled by the sweep */
} x264hip_mb_state;''','''    int poc, n_ref0, inv_ref_poc[8];                   /* x264_frame_t.i_poc / i_ref[0] / inv_ref_poc, filled by the sweep */
    int16_t *mvd;          /* [n][16][2] h->mb.mvd[0] (CABAC contexts of the row below; raster variant only) */
} x264hip_mb_state;

/* ---- round 2: the raster-order variant of the sweep ------------------------------------------------
 * With the RD levels (subme >= 6: every trial encode is priced against the LIVE CABAC contexts, R/encoder/rdo.c:62,139-171),
 * trellis quantisation (rdo.c:475-493) or adaptive quantisation (a macroblock's QP follows from the previous one's,
 * R/encoder/ratecontrol.c:263-264) a slice is one serial chain of macroblocks.  When x264hip_slice_params.rd is set, one
 * wavefront owns a whole frame of one chain of the batch and walks it in raster order, and the entropy coder runs inside the
 * loop exactly where x264_slice_write has it (R/encoder/encoder.c:1155-1165,1192-1205,1269-1273): the launch also returns every
 * chain's slice_data() bytes.  Throughput then comes from the number of chains in flight (the batch), not from a wavefront
 * schedule inside the frame.  I and P slices, CABAC; sub-8x8 partitions, psy-trellis and subme >= 8 are refused.        */
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
} x264hip_slice_rd;
#define X264HIP_PAYLOAD_LEAD 64''')
rep('''    int lossless;
} x264hip_slice_params;''','''    int lossless;
    const struct x264hip_slice_rd *rd;   /* NULL: the wavefront schedule of round 1; set: the raster-order variant (below) */
} x264hip_slice_params;''')
rep('''typedef struct {
    int slice_type;                    /* 0 = SLICE_TYPE_P, 2 = SLICE_TYPE_I (R/common/common.h:128-134) */''','''struct x264hip_slice_rd;
typedef struct {
    int slice_type;                    /* 0 = SLICE_TYPE_P, 2 = SLICE_TYPE_I (R/common/common.h:128-134) */''')
open(p,'w').write(s)

p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
rep('''        {(void **)&st->progress, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1)}};''','''        {(void **)&st->progress, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1)}, {(void **)&st->mvd, 64 * n}};''')
rep('''                  st->nnz, st->luma, st->luma_dc, st->chroma_dc, st->chroma_ac, st->cost_intra, st->cost_inter, st->cost_intra_alt, st->progress};''',
    '''                  st->nnz, st->luma, st->luma_dc, st->chroma_dc, st->chroma_ac, st->cost_intra, st->cost_inter, st->cost_intra_alt, st->progress, st->mvd};''')
rep('''    if (p->subme < 0 || p->subme > 5) { set_error("slice_sweep: subme %d needs RD, not built", p->subme); return -1; }''',
    '''    const x264hip_slice_rd *prd = p->rd;
    const int mbrd = (p->subme >= 6) + (p->subme >= 8);
    if (p->subme < 0 || p->subme > 7) { set_error("slice_sweep: subme %d (RD refinement of vectors and intra modes) not built", p->subme); return -1; }
    if (mbrd && (!prd || !prd->write || !p->cabac)) { set_error("slice_sweep: subme %d prices its trial encodes against the live CABAC contexts: it needs x264hip_slice_params.rd with write = 1 and cabac = 1", p->subme); return -1; }
    if (prd) {
        if (prd->write && !p->cabac) { set_error("slice_sweep: the in-loop entropy coder is CABAC only"); return -1; }
        if (prd->write && (!prd->payload || !prd->payload_len || prd->payload_cap < 4096)) { set_error("slice_sweep: payload buffers missing"); return -1; }
        if (prd->trellis && (!prd->write || !prd->unquant4_mf || (p->transform8x8 && !prd->unquant8_mf))) { set_error("slice_sweep: trellis needs write = 1 and the unquant tables"); return -1; }
        if (prd->trellis < 0 || prd->trellis > 2) { set_error("slice_sweep: trellis %d", prd->trellis); return -1; }
        if (prd->aq_offset && !prd->cost_mv_all) { set_error("slice_sweep: adaptive quantisation needs cost_mv_all"); return -1; }
        if (p->lossless) { set_error("slice_sweep: lossless is not built in the raster variant"); return -1; }
        if (mbrd && (p->analyse_inter & 0x20)) { set_error("slice_sweep: sub-8x8 partitions with the RD levels (x264_rd_cost_part) not built"); return -1; }
        if (!out->mvd) { set_error("slice_sweep: mb_state without mvd"); return -1; }
    }''')
rep('''    const dim3 grid((unsigned)(a.batch_pad * a.mb_h)), block(64);
    switch (a.lossless ? 0 : wpe) {
    case 0: hipLaunchKernelGGL((k_slice_sweep<2, true>), grid, block, 0, c->stream, a, t); break;
    case 1: hipLaunchKernelGGL(k_slice_sweep<1>, grid, block, 0, c->stream, a, t); break;
    case 3: hipLaunchKernelGGL(k_slice_sweep<3>, grid, block, 0, c->stream, a, t); break;
    default: hipLaunchKernelGGL(k_slice_sweep<2>, grid, block, 0, c->stream, a, t); break;
    }
    if (is_p && a.flags_intra)''','''    SwRd r;
    memset(&r, 0, sizeof(r));
    if (prd) {
        r.on = 1; r.mbrd = mbrd; r.trellis = p->cabac ? prd->trellis : 0; r.psy_rd = mbrd ? prd->psy_rd : 0;
        r.write = prd->write; r.cabac_init_idc = prd->cabac_init_idc; r.i_frame = prd->i_frame;
        r.aq = prd->aq_offset != nullptr; r.qp_min = prd->qp_min; r.qp_max = prd->qp_max; r.chroma_qp_offset = p->chroma_qp_offset;
        r.f_qpm = prd->f_qpm; r.aq_offset = prd->aq_offset; r.cost_mv_all = prd->cost_mv_all;
        r.unq4 = prd->unquant4_mf; r.unq8 = prd->unquant8_mf;
        r.payload = prd->payload; r.payload_cap = prd->payload_cap; r.payload_len = prd->payload_len; r.mb_bits = prd->mb_bits;
        r.mvd = out->mvd;
        hipLaunchKernelGGL((k_slice_sweep<2, false, true>), dim3((unsigned)a.batch), dim3(64), 0, c->stream, a, t, r);
    } else {
    const dim3 grid((unsigned)(a.batch_pad * a.mb_h)), block(64);
    switch (a.lossless ? 0 : wpe) {
    case 0: hipLaunchKernelGGL((k_slice_sweep<2, true>), grid, block, 0, c->stream, a, t, r); break;
    case 1: hipLaunchKernelGGL(k_slice_sweep<1>, grid, block, 0, c->stream, a, t, r); break;
    case 3: hipLaunchKernelGGL(k_slice_sweep<3>, grid, block, 0, c->stream, a, t, r); break;
    default: hipLaunchKernelGGL(k_slice_sweep<2>, grid, block, 0, c->stream, a, t, r); break;
    }
    }
    if (!prd && is_p && a.flags_intra)''')
open(p,'w').write(s)
print("ok")
