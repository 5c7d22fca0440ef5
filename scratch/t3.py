import sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:110]); sys.exit(1)
    s=s.replace(a,b)
# save_mb
rep('''        S->cbp[m->mb] = (i16)(m->type == S_P_SKIP ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma);''',
    '''        S->cbp[m->mb] = (i16)(S_IS_SKIP(m->type) ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma);''')
rep('''            const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2), keep = !intra && m->type != S_P_SKIP;
            S->mvd[(m->mb * 16 + i) * 2] = keep ? m->cmvd[k][0] : 0; S->mvd[(m->mb * 16 + i) * 2 + 1] = keep ? m->cmvd[k][1] : 0;
        }''','''            const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2), keep = !intra && !S_IS_SKIP(m->type) && !S_IS_DIRECT(m->type);
            S->mvd[(m->mb * 16 + i) * 2] = keep ? m->cmvd[k][0] : 0; S->mvd[(m->mb * 16 + i) * 2 + 1] = keep ? m->cmvd[k][1] : 0;
            if (S->slice_type == S_SLICE_B) { S->mvd1[(m->mb * 16 + i) * 2] = keep ? m->cmvd1[k][0] : 0; S->mvd1[(m->mb * 16 + i) * 2 + 1] = keep ? m->cmvd1[k][1] : 0; }
        }
        if (S->slice_type == S_SLICE_B)                  /* macroblock.c:1354-1368 */
            S->skipbp[m->mb] = m->type == S_B_SKIP || m->type == S_B_DIRECT ? 0xf
                             : m->type == S_B_8x8 ? (m->sub[0] == S_D_DIRECT_8x8) | (m->sub[1] == S_D_DIRECT_8x8) << 1 | (m->sub[2] == S_D_DIRECT_8x8) << 2 | (m->sub[3] == S_D_DIRECT_8x8) << 3 : 0;''')
rep('''    for (int i = 0; i < 4; i++) S->fdec->ref[m->mb * 4 + i] = intra ? -1 : m->ref8[i];
    if (intra) S->intra_count++;''','''    for (int i = 0; i < 4; i++) S->fdec->ref[m->mb * 4 + i] = intra ? -1 : m->ref8[i];
    if (S->slice_type == S_SLICE_B) {
        for (int i = 0; i < 16; i++) {
            S->fdec->mv1[(m->mb * 16 + i) * 2] = intra ? 0 : m->mv4_1[i][0];
            S->fdec->mv1[(m->mb * 16 + i) * 2 + 1] = intra ? 0 : m->mv4_1[i][1];
        }
        for (int i = 0; i < 4; i++) S->fdec->ref1[m->mb * 4 + i] = intra ? -1 : m->ref8_1[i];
    }
    if (intra) S->intra_count++;''')
rep('''    o->mb_type[M] = m->type; o->partition[M] = intra || m->type == S_P_SKIP ? S_D_16x16 : m->partition;
    for (int i = 0; i < 4; i++) o->sub_partition[M * 4 + i] = m->type == S_P_8x8 ? m->sub[i] : 0;''','''    o->mb_type[M] = m->type; o->partition[M] = intra || S_IS_SKIP(m->type) || m->type == S_B_DIRECT ? S_D_16x16 : m->partition;
    for (int i = 0; i < 4; i++) o->sub_partition[M * 4 + i] = m->type == S_P_8x8 || m->type == S_B_8x8 ? m->sub[i] : 0;''')
rep('''    o->cbp[M] = m->type == S_P_SKIP ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma;''','''    o->cbp[M] = S_IS_SKIP(m->type) ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma;''')
rep('''    if (S->slice_type == S_SLICE_P) {
        memcpy(o->mv + M * 32, S->fdec->mv + m->mb * 32, 64);
        memcpy(o->ref + M * 4, S->fdec->ref + m->mb * 4, 4);''','''    if (S->o2 && S->o2->mv1) {
        if (S->slice_type == S_SLICE_B) { memcpy(S->o2->mv1 + M * 32, S->fdec->mv1 + m->mb * 32, 64); memcpy(S->o2->ref1 + M * 4, S->fdec->ref1 + m->mb * 4, 4); }
        else { memset(S->o2->mv1 + M * 32, 0, 64); memset(S->o2->ref1 + M * 4, -1, 4); }
    }
    if (S->slice_type != S_SLICE_I) {
        memcpy(o->mv + M * 32, S->fdec->mv + m->mb * 32, 64);
        memcpy(o->ref + M * 4, S->fdec->ref + m->mb * 4, 4);''')
rep('''    if (m->type != S_P_SKIP && m->type != S_I_PCM) {
        if (m->type == S_I_16x16 && m->nnz[24]) memcpy(ldc, m->dc16, 32);''','''    if (!S_IS_SKIP(m->type) && m->type != S_I_PCM) {
        if (m->type == S_I_16x16 && m->nnz[24]) memcpy(ldc, m->dc16, 32);''')
# the chain
rep('''    if (e && e->psy_trellis != 0) return -3;
    memset(&S, 0, sizeof(S));''','''    if (e && e->psy_trellis != 0) return -3;
    const int nb = e ? clip3i(e->bframes, 0, 16) : 0;
    if (nb && (!b_write || p->qp == 0 || p->noise_reduction || (e->direct_pred != 1 && e->direct_pred != 2) || p->subme < 1)) return -3;   /* B slices: CABAC with the writer in the loop */
    memset(&S, 0, sizeof(S));''')
rep('''    S.mvr = calloc((size_t)p->n_refs * S.n * 2, sizeof(i16));
    S.fenc = sframe_new(&S);''','''    S.mvr = calloc((size_t)p->n_refs * S.n * 2, sizeof(i16));
    S.mvr1 = calloc((size_t)S.n * 2, sizeof(i16)); S.mvd1 = calloc((size_t)S.n * 32, sizeof(i16)); S.skipbp = calloc(S.n, 1);
    S.fenc = sframe_new(&S);''')
rep('''    for (int f = 0; f < p->n_frames; f++) {
        int idr = p->keyint > 0 ? f % p->keyint == 0 : f == 0;
        size_t F = f;
        if (idr) { for (int i = 0; i < n_avail; i++) sframe_free(refs[i]); n_avail = 0; last_idr = f; }
        for (int y = 0; y < p->height; y++) memcpy(S.fenc->plane[0] + y * S.sy, src_y + (F * p->height + y) * p->width, p->width);
        for (int y = 0; y < chh; y++) {
            memcpy(S.fenc->plane[1] + y * S.sc, src_u + (F * chh + y) * cw, cw);
            memcpy(S.fenc->plane[2] + y * S.sc, src_v + (F * chh + y) * cw, cw);
        }''','''    /* coding order with a fixed pattern of nb disposable B frames (x264_slicetype_decide without b-adapt, then the reordering of
     * x264_encoder_encode, R/encoder/encoder.c:1390-1460): an anchor every nb + 1 frames after an IDR, the last frame before the next
     * IDR / the end of the clip is an anchor too, and every anchor is coded before the B frames it closes */
    int *order = malloc(sizeof(int) * (p->n_frames + 1)), *ftype = malloc(sizeof(int) * (p->n_frames + 1)), n_order = 0;
    for (int t = 0; t < p->n_frames;) {
        if (p->keyint > 0 ? t % p->keyint == 0 : t == 0) { order[n_order] = t; ftype[n_order++] = S_SLICE_I; t++; continue; }
        int lim = p->keyint > 0 ? (t / p->keyint + 1) * p->keyint : p->n_frames;
        if (lim > p->n_frames) lim = p->n_frames;
        const int anchor = t + nb < lim - 1 ? t + nb : lim - 1;
        order[n_order] = anchor; ftype[n_order++] = S_SLICE_P;
        for (int b = t; b < anchor; b++) { order[n_order] = b; ftype[n_order++] = S_SLICE_B; }
        t = anchor + 1;
    }
    const int dpb = p->n_refs > (nb ? 2 : 1) ? p->n_refs : (nb ? 2 : 1);   /* sps->vui.i_max_dec_frame_buffering, R/encoder/set.c:196-200 */
    for (int f = 0; f < p->n_frames; f++) {
        const int disp = order[f], idr = ftype[f] == S_SLICE_I, is_b = ftype[f] == S_SLICE_B;
        size_t F = f, D = disp;
        if (idr) { for (int i = 0; i < n_avail; i++) sframe_free(refs[i]); n_avail = 0; last_idr = disp; }
        for (int y = 0; y < p->height; y++) memcpy(S.fenc->plane[0] + y * S.sy, src_y + (D * p->height + y) * p->width, p->width);
        for (int y = 0; y < chh; y++) {
            memcpy(S.fenc->plane[1] + y * S.sc, src_u + (D * chh + y) * cw, cw);
            memcpy(S.fenc->plane[2] + y * S.sc, src_v + (D * chh + y) * cw, cw);
        }''')
rep('''        S.fdec->poc = 2 * (f - last_idr);
        S.n_ref = n_avail < p->n_refs ? n_avail : p->n_refs;
        for (int i = 0; i < S.n_ref; i++) S.fref[i] = refs[i];
        S.slice_type = idr ? S_SLICE_I : S_SLICE_P;
        /* CQP: x264_ratecontrol_new / _start, R/encoder/ratecontrol.c:370-373,845-853 (ip_factor 1.4) */
        S.frame_qp = idr ? clip3i((int)(p->qp - 6.0 * log(1.4f) / log(2.0) + 0.5), 0, 51) : p->qp;''','''        S.fdec->poc = 2 * (disp - last_idr); S.fdec->kept = !is_b;
        /* x264_reference_build_list, R/encoder/encoder.c:911-981: list 0 = earlier pictures, nearest first; list 1 = later pictures, nearest first */
        S.n_ref = S.n_ref1 = 0;
        for (int i = 0; i < n_avail; i++) {
            if (refs[i]->poc < S.fdec->poc) S.fref[S.n_ref++] = refs[i];
            else if (refs[i]->poc > S.fdec->poc && S.n_ref1 < 2) S.fref1[S.n_ref1++] = refs[i];
        }
        for (int i = 0; i < S.n_ref; i++)
            for (int k = i + 1; k < S.n_ref; k++)
                if (S.fref[k]->poc > S.fref[i]->poc) { sframe *t_ = S.fref[i]; S.fref[i] = S.fref[k]; S.fref[k] = t_; }
        if (S.n_ref1 == 2 && S.fref1[1]->poc < S.fref1[0]->poc) { sframe *t_ = S.fref1[0]; S.fref1[0] = S.fref1[1]; S.fref1[1] = t_; }
        if (S.n_ref1 > (nb ? 1 : 0)) S.n_ref1 = nb ? 1 : 0;     /* h->frames.i_max_ref1 */
        if (S.n_ref > p->n_refs) S.n_ref = p->n_refs;
        S.slice_type = ftype[f];
        S.direct_spatial = !e || e->direct_pred != 2;
        /* CQP: x264_ratecontrol_new / _start, R/encoder/ratecontrol.c:370-373,845-853 (ip_factor 1.4, pb_factor 1.3) */
        S.frame_qp = idr ? clip3i((int)(p->qp - 6.0 * log(1.4f) / log(2.0) + 0.5), 0, 51)
                   : is_b ? clip3i((int)(p->qp + 6.0 * log(1.3f) / log(2.0) + 0.5), 0, 51) : p->qp;
        S.mbrd = (p->subme - is_b >= 6) + (p->subme - is_b >= 8);   /* analyse.c:222-225: one level less in B slices */''')
rep('''        S.fdec->n_ref0 = S.n_ref;
        for (int i = 0; i < S.n_ref; i++) {
            int delta = S.fdec->poc - S.fref[i]->poc;
            S.fdec->ref_poc[i] = S.fref[i]->poc;
            S.fdec->inv_ref_poc[i] = (256 + delta / 2) / delta;
        }''','''        S.fdec->n_ref0 = S.n_ref;
        for (int i = 0; i < S.n_ref; i++) {
            int delta = S.fdec->poc - S.fref[i]->poc;
            S.fdec->ref_poc[i] = S.fref[i]->poc;
            S.fdec->inv_ref_poc[i] = (256 + delta / 2) / delta;
        }
        if (is_b) b_slice_init(&S, e);''')
rep('''        o->frame_info[4 * F + 3] = S.fdec->poc;
        for (int mb = 0; mb < S.n; mb++) {''','''        o->frame_info[4 * F + 3] = S.fdec->poc;
        if (o2 && o2->frame_info2) { o2->frame_info2[4 * F] = disp; o2->frame_info2[4 * F + 1] = S.n_ref1; o2->frame_info2[4 * F + 2] = !is_b; o2->frame_info2[4 * F + 3] = 0; }
        for (int mb = 0; mb < S.n; mb++) {''')
rep('''            memset(&A, 0, sizeof(A));
            analyse_mb(&S, &m, &A);
            if (S.mbrd) update_cache(&S, &m, &A);          /* :2763 */
            else update_mb(&S, &m);''','''            struct banalysis BA;
            memset(&A, 0, sizeof(A)); memset(&BA, 0, sizeof(BA));
            A.B = &BA;
            analyse_mb(&S, &m, &A);
            if (is_b) { update_cache(&S, &m, &A); if (!S.mbrd) analyse_transform_b(&S, &m); b_final_vectors(&m); }   /* :2763-2766 */
            else if (S.mbrd) update_cache(&S, &m, &A);     /* :2763 */
            else update_mb(&S, &m);''')
rep('''                if (m.type == S_P_SKIP) cw_mb_skip(&S, &S.cb, &m, 1);
                else {
                    if (S.slice_type != S_SLICE_I) cw_mb_skip(&S, &S.cb, &m, 0);
                    if (!S_IS_INTRA(m.type)) for (int i = 0; i < 16; i++) {   /* the cache as x264_analyse_update_cache leaves it */''','''                if (S_IS_SKIP(m.type)) cw_mb_skip(&S, &S.cb, &m, 1);
                else {
                    if (S.slice_type != S_SLICE_I) cw_mb_skip(&S, &S.cb, &m, 0);
                    if (!S_IS_INTRA(m.type) && !is_b) for (int i = 0; i < 16; i++) {   /* the cache as x264_analyse_update_cache leaves it */''')
rep('''        /* x264_fdec_filter_row over the finished frame: loop filter, borders, half-pel planes */
        if (p->deblock) {''','''        /* x264_fdec_filter_row over the finished frame: loop filter, borders, half-pel planes -- nothing of it for a disposable B frame (encoder.c:986-1024) */
        if (is_b) {
            for (int y = 0; y < S.h16; y++) memcpy(o->fin_y + (F * S.h16 + y) * S.w16, S.fdec->plane[0] + y * S.sy, S.w16);
            for (int y = 0; y < S.h16 / 2; y++) {
                memcpy(o->fin_u + (F * S.h16 / 2 + y) * (S.w16 / 2), S.fdec->plane[1] + y * S.sc, S.w16 / 2);
                memcpy(o->fin_v + (F * S.h16 / 2 + y) * (S.w16 / 2), S.fdec->plane[2] + y * S.sc, S.w16 / 2);
            }
            sframe_free(S.fdec);
            continue;
        }
        if (p->deblock) {''')
rep('''        refs[0] = S.fdec; n_avail++;
        if (n_avail > p->n_refs) sframe_free(refs[--n_avail]);
    }''','''        refs[0] = S.fdec; n_avail++;
        if (n_avail > dpb) sframe_free(refs[--n_avail]);        /* x264_reference_update, encoder.c:1060-1093 */
    }
    free(order); free(ftype);''')
rep('''    free(S.nnz); free(S.i4mode); free(S.t8); free(S.mvr);''','''    free(S.nnz); free(S.i4mode); free(S.t8); free(S.mvr); free(S.mvr1); free(S.mvd1); free(S.skipbp);''')
rep('''    S.mbrd = (p->subme >= 6) + (p->subme >= 8);
''','')
open(p,'w').write(s)
print("ok")
