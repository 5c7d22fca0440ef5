import re,sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:90]); sys.exit(1)
    s=s.replace(a,b)

# encode_mb P_SKIP uses the cached vector
rep("""        if (!m->skip_mc) {
            int mvx = m->mvx, mvy = m->mvy;
            mv_clip_frame(S, m, &mvx, &mvy);""","""        if (!m->skip_mc) {
            int mvx = m->mv4[0][0], mvy = m->mv4[0][1];          /* h->mb.cache.mv[0][x264_scan8[0]], macroblock.c:380-383 */
            mv_clip_frame(S, m, &mvx, &mvy);""")

# load_mb: CABAC neighbour state
rep("""    m->partition = S_D_16x16;
}
""","""    m->partition = S_D_16x16;
    /* what the entropy coder reads of the neighbours (R/common/macroblock.c:896-1010,1129-1160) */
    m->cbp_left = m->cbp_top = -1; m->cpm_left = m->cpm_top = 0; m->nb_t8 = 0;
    memset(m->nz_l, 0x80, 4); memset(m->nz_t, 0x80, 4); memset(m->nz_lc, 0x80, 4); memset(m->nz_tc, 0x80, 4);
    memset(m->cmvd, 0, sizeof(m->cmvd));
    if (S->cbp) {
        if (m->nb & NB_TOP) {
            const int t = m->mb - S->mb_w;
            const u8 *nz = S->nnz + t * 27;
            m->cbp_top = S->cbp[t]; m->cpm_top = S->chroma_pm[t]; m->nb_t8 += S->t8[t];
            m->nz_t[0] = nz[10]; m->nz_t[1] = nz[11]; m->nz_t[2] = nz[14]; m->nz_t[3] = nz[15];
            for (int ch = 0; ch < 2; ch++) { m->nz_tc[ch][0] = nz[16 + 4 * ch + 2]; m->nz_tc[ch][1] = nz[16 + 4 * ch + 3]; }
            for (int i = 0; i < 4; i++) { m->cmvd[4 + i][0] = S->mvd[(t * 16 + 12 + i) * 2]; m->cmvd[4 + i][1] = S->mvd[(t * 16 + 12 + i) * 2 + 1]; }
        }
        if (m->nb & NB_LEFT) {
            const int l = m->mb - 1;
            const u8 *nz = S->nnz + l * 27;
            m->cbp_left = S->cbp[l]; m->cpm_left = S->chroma_pm[l]; m->nb_t8 += S->t8[l];
            m->nz_l[0] = nz[5]; m->nz_l[1] = nz[7]; m->nz_l[2] = nz[13]; m->nz_l[3] = nz[15];
            for (int ch = 0; ch < 2; ch++) { m->nz_lc[ch][0] = nz[16 + 4 * ch + 1]; m->nz_lc[ch][1] = nz[16 + 4 * ch + 3]; }
            for (int i = 0; i < 4; i++) { m->cmvd[11 + 8 * i][0] = S->mvd[(l * 16 + 3 + 4 * i) * 2]; m->cmvd[11 + 8 * i][1] = S->mvd[(l * 16 + 3 + 4 * i) * 2 + 1]; }
        }
    }
}
""")
open(p,'w').write(s)
print('ok')
