import sys
def patch(p, pairs):
    s=open(p).read()
    for a,b in pairs:
        if s.count(a)!=1: print("MISMATCH",p,s.count(a),a[:100]); sys.exit(1)
        s=s.replace(a,b)
    open(p,'w').write(s)
R='/root/repo/'
patch(R+'include/x264hip.h',[
('''    int16_t *mvd;          /* [n][16][2] h->mb.mvd[0] (CABAC contexts of the row below; raster variant only) */
} x264hip_mb_state;''','''    int16_t *mvd;          /* [n][16][2] h->mb.mvd[0] (CABAC contexts of the row below; raster variant only) */
    /* B slices: list 1 of the same (h->mb.mv[1] / ref[1] / mvr[1][0] / mvd[1]) and h->mb.skipbp */
    int16_t *mv1;          /* [n][16][2] */
    int8_t  *ref1;         /* [n][4] */
    int16_t *mvr1;         /* [n][2] */
    int16_t *mvd1;         /* [n][16][2] */
    uint8_t *skipbp;       /* [n] */
} x264hip_mb_state;'''),
('''struct x264hip_slice_rd;
''','''struct x264hip_slice_rd;
struct x264hip_slice_b;
'''),
('''    const struct x264hip_slice_rd *rd;   /* NULL: the wavefront schedule of round 1; set: the raster-order variant (below) */
} x264hip_slice_params;''','''    const struct x264hip_slice_rd *rd;   /* NULL: the wavefront schedule of round 1; set: the raster-order variant (below) */
    const struct x264hip_slice_b *b;     /* slice_type 1 (B): list 1 and what direct prediction reads (below); needs rd */
} x264hip_slice_params;

/* A B slice (slice_type = 1; the raster variant with the entropy coder: rd set, write = 1, subme 7).  x264 core 66 without
 * b-pyramid has one list-1 picture and its B frames are disposable (never references).  refs / n_refs of the call are list 0
 * (x264_reference_build_list, R/encoder/encoder.c:911-981: earlier pictures, nearest first); l0 = refs[0]'s state as always. */
typedef struct x264hip_slice_b {
    const x264hip_picture *fref1;        /* h->fref1[0]: the next anchor, reconstructed, borders expanded, half-pel planes built */
    const x264hip_mb_state *l1_state;    /* the state it was coded with: mb_type / ref / mv of the co-located macroblocks (x264_mb_predict_mv_direct16x16) */
    int ref1_poc;                        /* h->fref1[0]->i_poc (x264_macroblock_bipred_init, R/common/macroblock.c:1374-1408) */
    int weightb;                         /* param.analyse.b_weighted_bipred */
    int direct_spatial;                  /* sh.b_direct_spatial_mv_pred; 0 (temporal) is refused for now */
} x264hip_slice_b;''')])
patch(R+'x264_vs2008_amd/slice.py',[
('''               [("progress", C.c_void_p), ("poc", C.c_int), ("n_ref0", C.c_int), ("inv_ref_poc", C.c_int * 8), ("mvd", C.c_void_p)]''',
'''               [("progress", C.c_void_p), ("poc", C.c_int), ("n_ref0", C.c_int), ("inv_ref_poc", C.c_int * 8), ("mvd", C.c_void_p),
                ("mv1", C.c_void_p), ("ref1", C.c_void_p), ("mvr1", C.c_void_p), ("mvd1", C.c_void_p), ("skipbp", C.c_void_p)]


class SliceB(C.Structure):
    """x264hip_slice_b: list 1 of a B slice and what direct prediction reads."""
    _fields_ = [("fref1", C.c_void_p), ("l1_state", C.c_void_p), ("ref1_poc", C.c_int), ("weightb", C.c_int), ("direct_spatial", C.c_int)]'''),
('''                ("rd", C.c_void_p)]


class NrState''','''                ("rd", C.c_void_p), ("b", C.c_void_p)]


class NrState''')])
patch(R+'x264_vs2008_amd/csrc/frame_slice.hip',[
('''        {(void **)&st->progress, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1)}, {(void **)&st->mvd, 64 * n}};''','''        {(void **)&st->progress, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1)}, {(void **)&st->mvd, 64 * n},
        {(void **)&st->mv1, 64 * n}, {(void **)&st->ref1, 4 * n}, {(void **)&st->mvr1, 4 * n}, {(void **)&st->mvd1, 64 * n}, {(void **)&st->skipbp, n}};'''),
('''st->nnz, st->luma, st->luma_dc, st->chroma_dc, st->chroma_ac, st->cost_intra, st->cost_inter, st->cost_intra_alt, st->progress, st->mvd};''',
'''st->nnz, st->luma, st->luma_dc, st->chroma_dc, st->chroma_ac, st->cost_intra, st->cost_inter, st->cost_intra_alt, st->progress, st->mvd,
                  st->mv1, st->ref1, st->mvr1, st->mvd1, st->skipbp};'''),
('''    const bool is_p = p->slice_type == 0;
    if (p->slice_type != 0 && p->slice_type != 2) { set_error("slice_sweep: slice type %d not built (0 P, 2 I)", p->slice_type); return -1; }
    if (is_p && (n_refs < 1 || n_refs > SW_MAX_REFS)) { set_error("slice_sweep: %d references (1..%d)", n_refs, SW_MAX_REFS); return -1; }''',
'''    const bool is_b = p->slice_type == 1, is_p = p->slice_type == 0 || is_b;     // is_p: "has list 0" in what follows
    if (p->slice_type != 0 && p->slice_type != 1 && p->slice_type != 2) { set_error("slice_sweep: slice type %d (0 P, 1 B, 2 I)", p->slice_type); return -1; }
    if (is_p && (n_refs < 1 || n_refs > SW_MAX_REFS)) { set_error("slice_sweep: %d references (1..%d)", n_refs, SW_MAX_REFS); return -1; }
    const x264hip_slice_b *pb = is_b ? p->b : nullptr;
    if (is_b) {
        if (!pb || !pb->fref1 || !pb->l1_state || !p->rd) { set_error("slice_sweep: a B slice needs x264hip_slice_params.b (list 1) and .rd (the raster variant)"); return -1; }
        if (!pb->direct_spatial) { set_error("slice_sweep: temporal direct prediction is not built in the kernel yet (spatial is)"); return -1; }
        if (p->subme != 7 || !p->rd->write || !p->cabac) { set_error("slice_sweep: B slices are built for subme 7 (mode-decision RD) with the CABAC writer in the loop"); return -1; }
        if (p->noise_reduction || p->lossless) { set_error("slice_sweep: B slices with --nr / lossless are not built"); return -1; }
        if (!out->mv1 || !pb->l1_state->mb_type) { set_error("slice_sweep: mb_state without list-1 arrays"); return -1; }
    }'''),
('''    const int mbrd = (p->subme >= 6) + (p->subme >= 8);''','''    const int mbrd = (p->subme - is_b >= 6) + (p->subme - is_b >= 8);       /* one level less in a B slice, R/encoder/analyse.c:222-225 */'''),
('''    a.chroma_me = p->chroma_me && is_p && p->subme >= 5;            // h->mb.b_chroma_me, analyse.c:234-235
    a.fast_pskip = p->fast_pskip; a.dct_decimate = p->dct_decimate;''','''    a.chroma_me = p->chroma_me && is_p && !is_b && p->subme >= 5;   // h->mb.b_chroma_me, analyse.c:234-235
    a.fast_pskip = p->fast_pskip; a.dct_decimate = p->dct_decimate || is_b;     // B slices always decimate (R/encoder/macroblock.c:193,275,479)'''),
('''    a.flags_inter = is_p ? (p->analyse_inter & 0x30) : 0; a.mixed_refs = p->mixed_refs != 0;''','''    a.flags_inter = is_p ? (p->analyse_inter & (is_b ? 0x100 : 0x30)) : 0; a.mixed_refs = p->mixed_refs != 0;     // B: X264_ANALYSE_BSUB16x16'''),
('''    HIPCHK(hipMemsetAsync(out->progress, 0, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1), c->stream));''','''    if (is_b) {                                                      // x264_macroblock_bipred_init, R/common/macroblock.c:1374-1408
        for (int k = 0; k < 4; k++) t.y1[k] = pb->fref1->filtered[k];
        t.u1 = pb->fref1->plane[1]; t.v1 = pb->fref1->plane[2];
        for (int i = 0; i < SW_MAX_REFS; i++) {
            const int poc0 = p->ref_poc[i < n_refs ? i : 0];
            int td = pb->ref1_poc - poc0; td = td < -128 ? -128 : td > 127 ? 127 : td;
            int dsf = 256;
            if (td) {
                int tb = p->poc - poc0; tb = tb < -128 ? -128 : tb > 127 ? 127 : tb;
                const int tx = (16384 + (abs(td) >> 1)) / td;
                dsf = (tb * tx + 32) >> 6; dsf = dsf < -1024 ? -1024 : dsf > 1023 ? 1023 : dsf;
            }
            dsf >>= 2;
            t.biw[i] = pb->weightb && dsf >= -64 && dsf <= 128 ? 64 - dsf : 32;
        }
    } else {
        for (int k = 0; k < 4; k++) t.y1[k] = t.y[0][k];
        t.u1 = t.u[0]; t.v1 = t.v[0];
        for (int i = 0; i < SW_MAX_REFS; i++) t.biw[i] = 32;
    }
    HIPCHK(hipMemsetAsync(out->progress, 0, sizeof(int) * ((size_t)c->d.mb_h * c->batch + 1), c->stream));'''),
('''        r.mvd = out->mvd;
        x264hip_launch_slice_rd(a, t, r, c->stream);''','''        r.mvd = out->mvd;
        if (is_b) {
            r.mv1 = out->mv1; r.ref1 = (signed char *)out->ref1; r.mvr1 = out->mvr1; r.mvd1 = out->mvd1; r.skipbp = out->skipbp;
            r.col_type = (const signed char *)pb->l1_state->mb_type; r.col_ref = (const signed char *)pb->l1_state->ref; r.col_mv = pb->l1_state->mv;
            x264hip_launch_slice_b(a, t, r, c->stream);
        } else
        x264hip_launch_slice_rd(a, t, r, c->stream);'''),
('''    out->poc = p->poc; out->n_ref0 = is_p ? n_refs : 0;''','''    out->poc = p->poc; out->n_ref0 = is_p ? n_refs : 0;     /* (a B frame's state is never read by later frames) */'''),
])
print('ok')
