import sys
p='/root/repo/oracle/ref_slice.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep('''    int cabac_init_idc;                      /* param.i_cabac_init_idc */
} refslice_ext;''','''    int cabac_init_idc;                      /* param.i_cabac_init_idc */
    /* B slices: a fixed pattern of `bframes` non-reference B frames between anchors (what x264_slicetype_decide produces with
     * --b-adapt 0 and no --b-pyramid); the clip stays in display order, the chain is coded in coding order */
    int bframes;                             /* param.i_bframe */
    int weightb;                             /* param.analyse.b_weighted_bipred */
    int direct_pred;                         /* param.analyse.i_direct_mv_pred: 1 spatial, 2 temporal */
} refslice_ext;''')
rep('''    float *qp_offset;                        /* [F][n]: fenc->f_qp_offset (0 without AQ) */
} refslice_out2;''','''    float *qp_offset;                        /* [F][n]: fenc->f_qp_offset (0 without AQ) */
    int16_t *mv1;                            /* [F][n][16][2]: list 1 (B slices) */
    int8_t *ref1;                            /* [F][n][4] */
    int32_t *frame_info2;                    /* [F][4]: display index, i_ref1, kept as reference, 0 */
} refslice_out2;''')
# filter_row: only frames kept as reference are filtered (x264_fdec_filter_row: b_deblock &= b_hpel)
rep('''    if (!h->sh.i_disable_deblocking_filter_idc)
        x264_frame_deblock_row(h, min_y);
    x264_frame_expand_border(h, h->fdec, min_y, b_end);
    if (h->param.analyse.i_subpel_refine) {
        x264_frame_filter(h, h->fdec, min_y, b_end);
        x264_frame_expand_border_filtered(h, h->fdec, min_y, b_end);
    }''','''    if (!h->fdec->b_kept_as_ref)                     /* a disposable B frame is neither filtered nor interpolated (encoder.c:986-991,1016) */
        return;
    if (!h->sh.i_disable_deblocking_filter_idc)
        x264_frame_deblock_row(h, min_y);
    x264_frame_expand_border(h, h->fdec, min_y, b_end);
    if (h->param.analyse.i_subpel_refine) {
        x264_frame_filter(h, h->fdec, min_y, b_end);
        x264_frame_expand_border_filtered(h, h->fdec, min_y, b_end);
    }''')
rep('''        h->param.i_cabac_init_idc = x264_clip3(e->cabac_init_idc, 0, 2);''','''        h->param.i_cabac_init_idc = x264_clip3(e->cabac_init_idc, 0, 2);
        h->param.i_bframe = x264_clip3(e->bframes, 0, X264_BFRAME_MAX);
        h->param.analyse.b_weighted_bipred = e->weightb && h->param.i_bframe > 0;
        h->param.analyse.i_direct_mv_pred = e->direct_pred ? e->direct_pred : X264_DIRECT_PRED_SPATIAL;
        if (!p->subme && h->param.analyse.i_direct_mv_pred > X264_DIRECT_PRED_SPATIAL) h->param.analyse.i_direct_mv_pred = X264_DIRECT_PRED_SPATIAL;''')
rep('''    h->sps->b_frame_mbs_only = 1;''','''    h->sps->b_frame_mbs_only = 1;
    h->sps->b_direct8x8_inference = 1;                   /* x264_sps_init, R/encoder/set.c:136 */''')

# the frame loop: coding order
rep('''    for (f = 0; f < p->n_frames; f++) {
        int idr = p->keyint > 0 ? f % p->keyint == 0 : f == 0;
        size_t F = f;
        if (idr) {
            for (i = 0; i < n_avail; i++) x264_frame_delete(refs[i]);
            n_avail = 0; last_idr = f;
        }''','''    /* coding order (what x264_slicetype_decide + the frame reordering of x264_encoder_encode give for a fixed B pattern):
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
        }''')
rep('''            memcpy(h->fenc->plane[0] + y * h->fenc->i_stride[0], src_y + (F * p->height + y) * p->width, p->width);''',
    '''            memcpy(h->fenc->plane[0] + y * h->fenc->i_stride[0], src_y + (D * p->height + y) * p->width, p->width);''')
rep('''            memcpy(h->fenc->plane[1] + y * h->fenc->i_stride[1], src_u + (F * ch + y) * cw, cw);
            memcpy(h->fenc->plane[2] + y * h->fenc->i_stride[2], src_v + (F * ch + y) * cw, cw);''','''            memcpy(h->fenc->plane[1] + y * h->fenc->i_stride[1], src_u + (D * ch + y) * cw, cw);
            memcpy(h->fenc->plane[2] + y * h->fenc->i_stride[2], src_v + (D * ch + y) * cw, cw);''')
rep('''        h->fenc->i_frame = f; h->fenc->i_poc = 2 * (f - last_idr);
        h->fenc->i_type = idr ? X264_TYPE_IDR : X264_TYPE_P;
        h->fdec->i_frame = f; h->fdec->i_poc = h->fenc->i_poc; h->fdec->i_type = h->fenc->i_type; h->fdec->b_kept_as_ref = 1;''',
    '''        h->fenc->i_frame = disp; h->fenc->i_poc = 2 * (disp - last_idr);
        h->fenc->i_type = ftype[f];
        h->fdec->i_frame = disp; h->fdec->i_poc = h->fenc->i_poc; h->fdec->i_type = h->fenc->i_type;
        h->fenc->b_kept_as_ref = h->fdec->b_kept_as_ref = !is_b;''')
rep('''        h->i_ref0 = n_avail < p->n_refs ? n_avail : p->n_refs;
        for (i = 0; i < h->i_ref0; i++) h->fref0[i] = refs[i];
        h->i_ref1 = 0;
        h->mb.pic.i_fref[0] = h->i_ref0; h->mb.pic.i_fref[1] = 0;
        memset(&h->sh, 0, sizeof(h->sh));
        h->sh.i_type = idr ? SLICE_TYPE_I : SLICE_TYPE_P;''','''        /* x264_reference_build_list, R/encoder/encoder.c:911-981: by POC, list 0 downwards from the frame, list 1 upwards */
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
        memset(&h->sh, 0, sizeof(h->sh));
        h->sh.i_type = idr ? SLICE_TYPE_I : is_b ? SLICE_TYPE_B : SLICE_TYPE_P;
        h->sh.b_direct_spatial_mv_pred = h->param.analyse.i_direct_mv_pred == X264_DIRECT_PRED_SPATIAL;   /* x264_slice_header_init, encoder.c:116-122 */''')
rep('''        h->sh.i_num_ref_idx_l1_active = 1;''','''        h->sh.i_num_ref_idx_l1_active = h->i_ref1 <= 0 ? 1 : h->i_ref1;''')
rep('''        h->sh.i_qp = x264_ratecontrol_qp(h);
        x264_macroblock_slice_init(h);''','''        h->sh.i_qp = x264_ratecontrol_qp(h);
        if (is_b) x264_macroblock_bipred_init(h);            /* encoder.c:1534-1535 */
        x264_macroblock_slice_init(h);''')
rep('''        o->frame_info[4 * F + 2] = h->i_ref0; o->frame_info[4 * F + 3] = h->fdec->i_poc;''','''        o->frame_info[4 * F + 2] = h->i_ref0; o->frame_info[4 * F + 3] = h->fdec->i_poc;
        if (o2 && o2->frame_info2) { o2->frame_info2[4 * F] = disp; o2->frame_info2[4 * F + 1] = h->i_ref1; o2->frame_info2[4 * F + 2] = !is_b; o2->frame_info2[4 * F + 3] = 0; }''')
open(p,'w').write(s)
print('ok')
