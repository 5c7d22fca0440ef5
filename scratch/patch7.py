import re,sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:90]); sys.exit(1)
    s=s.replace(a,b)

# ---- save_mb
rep("""    int intra = S_IS_INTRA(m->type), cbp_dc = S->p->cabac ? (m->nnz[24] | m->nnz[25] << 1 | m->nnz[26] << 2) : 0;""",
"""    int intra = S_IS_INTRA(m->type), cbp_dc = S->p->cabac ? (m->nnz[24] | m->nnz[25] << 1 | m->nnz[26] << 2) : 0;
    if (m->type == S_I_PCM) {                            /* R/common/macroblock.c:1245-1255 */
        m->qp = 0; S->last_dqp = 0; m->cbp_chroma = 2; m->cbp_luma = 0xf; m->t8 = 0; cbp_dc = 7;
        memset(m->nnz, 16, 24); m->nnz[24] = m->nnz[25] = m->nnz[26] = 1;
    } else {                                             /* :1268-1272: a macroblock without coefficients has no QP of its own */
        if (m->type != S_I_16x16 && m->cbp_luma == 0 && m->cbp_chroma == 0) m->qp = S->last_qp;
        S->last_dqp = m->qp - S->last_qp;
        S->last_qp = m->qp;
    }
    S->prev_mb = m->mb;
    if (S->cbp) {
        S->cbp[m->mb] = (i16)(m->type == S_P_SKIP ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma);
        S->chroma_pm[m->mb] = (int8_t)(intra && m->type != S_I_PCM ? s_fix8c[m->chroma_mode] : 0);
        S->qp_mb[m->mb] = (int8_t)m->qp;
        for (int i = 0; i < 16; i++) {
            const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2), keep = !intra && m->type != S_P_SKIP;
            S->mvd[(m->mb * 16 + i) * 2] = keep ? m->cmvd[k][0] : 0; S->mvd[(m->mb * 16 + i) * 2 + 1] = keep ? m->cmvd[k][1] : 0;
        }
    }""")
rep("""    o->qp[M] = S->qp;
    o->cbp[M] = m->type == S_P_SKIP ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma;""",
"""    o->qp[M] = m->qp;
    o->cbp[M] = m->type == S_P_SKIP ? 0 : (cbp_dc << 8) | (m->cbp_chroma << 4) | m->cbp_luma;""")
rep("""    if (m->type != S_P_SKIP) {
        if (m->type == S_I_16x16 && m->nnz[24]) memcpy(ldc, m->dc16, 32);""","""    if (m->type != S_P_SKIP && m->type != S_I_PCM) {
        if (m->type == S_I_16x16 && m->nnz[24]) memcpy(ldc, m->dc16, 32);""")

# ---- chain
rep("""int x264o_encode_chain(const slice_params *p, const u8 *src_y, const u8 *src_u, const u8 *src_v, slice_out *o)
{
    ssl S;""","""/* per-QP tables of the current macroblock: h->quant4_mf[..][qp] etc., the lambdas and the mv / reference cost tables
 * (x264_mb_analyse_init + x264_mb_analyse_load_costs, R/encoder/analyse.c:220-232,182-218) */
static void set_mb_qp(ssl *S, smb *m, int qp)
{
    const slice_params *p = S->p;
    if (m) m->qp = qp;
    if (qp == S->qp && S->cost_mv) return;
    S->qp = qp;
    S->qpc = s_chroma_qp[clip3i(qp + (S->lossless ? 0 : S->chroma_qp_offset), 0, 51)];
    S->lambda = s_lambda_tab[qp]; S->lambda2 = s_lambda2_tab[qp];
    S->cost_mv = s_load_cost_mv(qp);
    for (int i = 0; i < 16; i++) S->ref_cost[i] = S->lambda * s_te_size(clip3i((S->n_ref <= 0 ? 1 : S->n_ref) - 1, 0, 2), i);
    for (int cat = 0; cat < 4; cat++) {
        x264o_cqm(p->cqm_preset, cat, cat < 2 ? S->qp : S->qpc, 0, S->mf4[cat], S->b4[cat], &S->dq4[cat][0][0]);
        x264o_cqm_unquant(p->cqm_preset, cat, cat < 2 ? S->qp : S->qpc, 0, S->unq4[cat]);
    }
    for (int cat = 0; cat < 2; cat++) {
        x264o_cqm(p->cqm_preset, cat, S->qp, 1, S->mf8[cat], S->b8[cat], &S->dq8[cat][0][0]);
        x264o_cqm_unquant(p->cqm_preset, cat, S->qp, 1, S->unq8[cat]);
    }
}
/* x264_adaptive_quant_frame, R/encoder/ratecontrol.c:231-249 (float, compiled like the reference: -ffp-contract=off) */
static void aq_frame(ssl *S)
{
    static const float log2_lut[128] = {
        0.00000, 0.01123, 0.02237, 0.03342, 0.04439, 0.05528, 0.06609, 0.07682, 0.08746, 0.09803, 0.10852, 0.11894, 0.12928, 0.13955, 0.14975, 0.15987,
        0.16993, 0.17991, 0.18982, 0.19967, 0.20945, 0.21917, 0.22882, 0.23840, 0.24793, 0.25739, 0.26679, 0.27612, 0.28540, 0.29462, 0.30378, 0.31288,
        0.32193, 0.33092, 0.33985, 0.34873, 0.35755, 0.36632, 0.37504, 0.38370, 0.39232, 0.40088, 0.40939, 0.41785, 0.42626, 0.43463, 0.44294, 0.45121,
        0.45943, 0.46761, 0.47573, 0.48382, 0.49185, 0.49985, 0.50779, 0.51570, 0.52356, 0.53138, 0.53916, 0.54689, 0.55459, 0.56224, 0.56986, 0.57743,
        0.58496, 0.59246, 0.59991, 0.60733, 0.61471, 0.62205, 0.62936, 0.63662, 0.64386, 0.65105, 0.65821, 0.66534, 0.67243, 0.67948, 0.68650, 0.69349,
        0.70044, 0.70736, 0.71425, 0.72110, 0.72792, 0.73471, 0.74147, 0.74819, 0.75489, 0.76155, 0.76818, 0.77479, 0.78136, 0.78790, 0.79442, 0.80090,
        0.80735, 0.81378, 0.82018, 0.82655, 0.83289, 0.83920, 0.84549, 0.85175, 0.85798, 0.86419, 0.87036, 0.87652, 0.88264, 0.88874, 0.89482, 0.90087,
        0.90689, 0.91289, 0.91886, 0.92481, 0.93074, 0.93664, 0.94251, 0.94837, 0.95420, 0.96000, 0.96578, 0.97154, 0.97728, 0.98299, 0.98868, 0.99435};
    const float strength = S->e->aq_strength * 1.0397;
    for (int mby = 0; mby < S->mb_h; mby++)
        for (int mbx = 0; mbx < S->mb_w; mbx++) {
            uint32_t energy = pixf.var[X264HIP_PIXEL_16x16](S->fenc->plane[0] + 16 * (mbx + mby * S->sy), S->sy)
                            + pixf.var[X264HIP_PIXEL_8x8](S->fenc->plane[1] + 8 * (mbx + mby * S->sc), S->sc)
                            + pixf.var[X264HIP_PIXEL_8x8](S->fenc->plane[2] + 8 * (mbx + mby * S->sc), S->sc);
            if (energy < 1) energy = 1;
            const int lz = __builtin_clz(energy);
            S->aq_offset[mbx + mby * S->mb_w] = strength * (log2_lut[(energy << lz >> 24) & 0x7f] - lz + 16.573f);
        }
}

static int s_encode_chain(const slice_params *p, const slice_ext *e, const u8 *src_y, const u8 *src_u, const u8 *src_v, slice_out *o, slice_out2 *o2)
{
    ssl S;
    const int b_write = e && e->write;""")
rep("""    if (p->subme > 5 || p->me_method > 3 || (p->me_method == 3 && p->subme < 1)) return -3;   /* ESA at subme 0: the reference never fills the integral plane */
    memset(&S, 0, sizeof(S));
    S.p = p; S.o = o;""","""    if (p->subme > 7 || p->me_method > 3 || (p->me_method == 3 && p->subme < 1)) return -3;   /* ESA at subme 0: the reference never fills the integral plane */
    if (p->subme > 5 && (!b_write || !p->cabac || (p->inter & 0x20) || p->qp == 0)) return -3;  /* RD levels: CABAC with the writer in the loop; not yet sub-8x8 / CAVLC / lossless */
    if (b_write && !p->cabac) return -3;
    if (e && e->psy_trellis != 0) return -3;
    memset(&S, 0, sizeof(S));
    S.p = p; S.o = o; S.e = e; S.o2 = o2;
    S.chroma_qp_offset = p->chroma_qp_offset;
    S.qp_min = p->cqm_preset ? 6 : 0; S.qp_max = 51;
    if (e) {                                 /* x264_validate_parameters, R/encoder/encoder.c:493-522 */
        const float psy = p->subme < 6 ? 0 : e->psy_rd < 0 ? 0 : e->psy_rd > 10 ? 10 : e->psy_rd;
        S.trellis = p->cabac ? clip3i(e->trellis, 0, 2) : 0;
        S.psy_rd = (int)(psy * (1 << 8) + .5);
        if (S.psy_rd) S.chroma_qp_offset -= psy < 0.25 ? 1 : 2;
        S.chroma_qp_offset = clip3i(S.chroma_qp_offset, -12, 12);
    }
    S.mbrd = (p->subme >= 6) + (p->subme >= 8);""")
rep("""    S.mvr = calloc((size_t)p->n_refs * S.n * 2, sizeof(i16));
    S.fenc = sframe_new(&S);""","""    S.mvr = calloc((size_t)p->n_refs * S.n * 2, sizeof(i16));
    S.fenc = sframe_new(&S);
    if (e) {
        S.cbp = calloc(S.n, sizeof(i16)); S.chroma_pm = calloc(S.n, 1); S.mvd = calloc((size_t)S.n * 32, sizeof(i16)); S.qp_mb = calloc(S.n, 1);
        S.aq_offset = calloc(S.n, sizeof(float));
        if (b_write) S.bsbuf = malloc(64 + (size_t)e->payload_cap + 4096);
        {   /* scan position -> raster index, from the scan functions themselves; the trellis weights in scan order (R/common/dct.c:476-483) */
            static const u16 w4[3] = {800, 320, 128}, w8[6] = {256, 201, 656, 227, 410, 363};
            static const u8 k8[16] = {0, 3, 4, 3, 3, 1, 5, 1, 4, 5, 2, 5, 3, 1, 5, 1};
            i16 d4[4][4], l4[16], d8[8][8], l8[64];
            for (int i = 0; i < 16; i++) d4[0][i] = (i16)i;
            zigf[0].scan_4x4(l4, d4);
            for (int i = 0; i < 16; i++) { S.zz4[i] = (u8)l4[i]; S.w4z[i] = w4[(l4[i] & 1) + ((l4[i] >> 2) & 1)]; }
            for (int i = 0; i < 64; i++) d8[0][i] = (i16)i;
            zigf[0].scan_8x8(l8, d8);
            for (int i = 0; i < 64; i++) { S.zz8[i] = (u8)l8[i]; S.w8z[i] = w8[k8[((l8[i] >> 1) & 12) | (l8[i] & 3)]]; }
        }
    }""")
rep("""        S.qp = idr ? clip3i((int)(p->qp - 6.0 * log(1.4f) / log(2.0) + 0.5), 0, 51) : p->qp;
        S.qpc = s_chroma_qp[clip3i(S.qp + (S.lossless ? 0 : p->chroma_qp_offset), 0, 51)];
        S.lambda = s_lambda_tab[S.qp]; S.lambda2 = s_lambda2_tab[S.qp];
        S.cost_mv = s_load_cost_mv(S.qp);
        for (int i = 0; i < 16; i++) S.ref_cost[i] = S.lambda * s_te_size(clip3i((S.n_ref <= 0 ? 1 : S.n_ref) - 1, 0, 2), i);
        for (int cat = 0; cat < 4; cat++) x264o_cqm(p->cqm_preset, cat, cat < 2 ? S.qp : S.qpc, 0, S.mf4[cat], S.b4[cat], &S.dq4[cat][0][0]);
        for (int cat = 0; cat < 2; cat++) x264o_cqm(p->cqm_preset, cat, S.qp, 1, S.mf8[cat], S.b8[cat], &S.dq8[cat][0][0]);""",
"""        S.frame_qp = idr ? clip3i((int)(p->qp - 6.0 * log(1.4f) / log(2.0) + 0.5), 0, 51) : p->qp;
        S.f_qpm = (float)S.frame_qp;                      /* rc->f_qpm = q, ratecontrol.c:868 (constant QP: an integer) */
        S.cost_mv = 0;
        set_mb_qp(&S, 0, S.frame_qp);
        const int b_aq = e && e->aq_mode > 0 && e->aq_strength != 0;
        if (b_aq) aq_frame(&S);""")
rep("""        S.intra_count = 0; S.stat_intra = S.stat_inter = S.stat_n = 0;""","""        S.intra_count = 0; S.stat_intra = S.stat_inter = S.stat_n = 0;
        S.last_qp = S.frame_qp; S.last_dqp = 0; S.i_skip = 0;
        if (b_write) {                                    /* x264_slice_write, R/encoder/encoder.c:1155-1165 */
            memset(S.bsbuf, 0, 64 + (size_t)e->payload_cap + 4096);
            cb_context_init(&S.cb, S.slice_type, S.frame_qp, clip3i(e->cabac_init_idc, 0, 2));
            cb_encode_init(&S.cb, S.bsbuf + 64, S.bsbuf + 64 + e->payload_cap + 4096);
            S.cb.i_frame = f;
        }""")
rep("""            smb m;
            load_mb(&S, &m, mb % S.mb_w, mb / S.mb_w);
            analyse_mb(&S, &m);
            update_mb(&S, &m);
            encode_mb(&S, &m);
            save_mb(&S, &m);
        }""","""            smb m;
            panalysis A;
            load_mb(&S, &m, mb % S.mb_w, mb / S.mb_w);
            /* x264_ratecontrol_qp + x264_adaptive_quant, R/encoder/analyse.c:2162-2164, ratecontrol.c:257-265 */
            int qp = S.frame_qp;
            if (b_aq) {
                qp = clip3i((int)(S.f_qpm + S.aq_offset[mb] + .5), S.qp_min, S.qp_max);
                if (abs(qp - S.last_qp) == 1) qp = S.last_qp;
            }
            set_mb_qp(&S, &m, qp);
            /* x264_mb_analyse_init, analyse.c:235-252 */
            S.b_trellis = S.trellis > 1 && S.mbrd;
            m.skip_intra = S.lossless ? 0 : S.mbrd ? 2 : !S.trellis && !p->noise_reduction;
            memset(&A, 0, sizeof(A));
            analyse_mb(&S, &m, &A);
            if (S.mbrd) update_cache(&S, &m, &A);          /* :2763 */
            else update_mb(&S, &m);
            S.b_trellis = S.trellis;                       /* :2768-2773 */
            if (S.b_trellis == 1 || p->noise_reduction) m.skip_intra = 0;
            encode_mb(&S, &m);
            if (b_write) {                                 /* encoder.c:1192-1205 */
                if (mb > 0) cb_encode_terminal(&S.cb);
                if (m.type == S_P_SKIP) cw_mb_skip(&S, &S.cb, &m, 1);
                else {
                    if (S.slice_type != S_SLICE_I) cw_mb_skip(&S, &S.cb, &m, 0);
                    if (!S_IS_INTRA(m.type)) for (int i = 0; i < 16; i++) {   /* the cache as x264_analyse_update_cache leaves it */
                        const int k = 4 + 1 * 8 + (i & 3) + 8 * (i >> 2);
                        m.cmv[k][0] = m.mv4[i][0]; m.cmv[k][1] = m.mv4[i][1]; m.cref[k] = m.ref8[(i >> 3) * 2 + ((i & 3) >> 1)];
                    }
                    cw_macroblock(&S, &S.cb, 0, &m);
                }
                o2->mb_bits[F * S.n + mb] = cb_pos(&S.cb);
                if (o2->mb_bits[F * S.n + mb] / 8 + 2048 > e->payload_cap) return -5;
            }
            save_mb(&S, &m);
            if (o2) o2->qp_offset[F * S.n + mb] = b_aq ? S.aq_offset[mb] : 0;
        }
        if (b_write) {                                     /* encoder.c:1269-1273 */
            cb_encode_flush(&S.cb, f);
            const int len = (int)(S.cb.p - (S.bsbuf + 64));
            if (len > e->payload_cap) return -5;
            o2->payload_len[F] = len;
            memcpy(o2->payload + F * e->payload_cap, S.bsbuf + 64, len);
        }""")
rep("""    sframe_free(S.fenc);
    free(S.nnz); free(S.i4mode); free(S.t8); free(S.mvr);
    return 0;
}""","""    sframe_free(S.fenc);
    free(S.nnz); free(S.i4mode); free(S.t8); free(S.mvr);
    free(S.cbp); free(S.chroma_pm); free(S.mvd); free(S.qp_mb); free(S.aq_offset); free(S.bsbuf);
    return 0;
}

int x264o_encode_chain(const slice_params *p, const u8 *src_y, const u8 *src_u, const u8 *src_v, slice_out *o)
{
    return s_encode_chain(p, 0, src_y, src_u, src_v, o, 0);
}
int x264o_encode_chain2(const slice_params *p, const slice_ext *e, const u8 *src_y, const u8 *src_u, const u8 *src_v, slice_out *o, slice_out2 *o2)
{
    return s_encode_chain(p, e, src_y, src_u, src_v, o, o2);
}""")
# qp used in chain frame_info
rep("""o->frame_info[4 * F + 1] = S.qp;""","""o->frame_info[4 * F + 1] = S.frame_qp;""")
open(p,'w').write(s)
print('ok')
