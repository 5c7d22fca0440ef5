import sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:110]); sys.exit(1)
    s=s.replace(a,b)
rep('''    if (S->slice_type == S_SLICE_P) {
        predict_mv_pskip(S, m, m->pskip_mv);
        /* h->mb.cache.ref / mv around the macroblock (R/common/macroblock.c:1040-1128): -2 = not available */
        memset(m->cref, -2, sizeof(m->cref)); memset(m->cmv, 0, sizeof(m->cmv));
        const i16 *fmv = S->fdec->mv;
        const int8_t *fref = S->fdec->ref;
#define NBSET(k_, o_, blk_) do { m->cref[k_] = fref[(o_) * 4 + ((blk_) >> 3) * 2 + (((blk_) & 3) >> 1)]; \\
                                 m->cmv[k_][0] = fmv[((o_) * 16 + (blk_)) * 2]; m->cmv[k_][1] = fmv[((o_) * 16 + (blk_)) * 2 + 1]; } while (0)
        if (m->nb & NB_TOPLEFT) NBSET(3, m->mb - S->mb_w - 1, 15);
        if (m->nb & NB_TOP) for (int i = 0; i < 4; i++) NBSET(4 + i, m->mb - S->mb_w, 12 + i);
        if (m->nb & NB_TOPRIGHT) NBSET(8, m->mb - S->mb_w + 1, 12);
        if (m->nb & NB_LEFT) for (int i = 0; i < 4; i++) NBSET(11 + 8 * i, m->mb - 1, 3 + 4 * i);
#undef NBSET
    }''','''    if (S->slice_type == S_SLICE_P) predict_mv_pskip(S, m, m->pskip_mv);
    for (int list = 0; list < (S->slice_type == S_SLICE_B ? 2 : S->slice_type == S_SLICE_P ? 1 : 0); list++) {
        /* h->mb.cache.ref / mv around the macroblock (R/common/macroblock.c:1040-1128): -2 = not available */
        int8_t *cref = CREF(m, list);
        i16 (*cmv)[2] = CMV(m, list);
        memset(cref, -2, 48); memset(cmv, 0, sizeof(m->cmv));
        const i16 *fmv = list ? S->fdec->mv1 : S->fdec->mv;
        const int8_t *fref = list ? S->fdec->ref1 : S->fdec->ref;
#define NBSET(k_, o_, blk_) do { cref[k_] = fref[(o_) * 4 + ((blk_) >> 3) * 2 + (((blk_) & 3) >> 1)]; \\
                                 cmv[k_][0] = fmv[((o_) * 16 + (blk_)) * 2]; cmv[k_][1] = fmv[((o_) * 16 + (blk_)) * 2 + 1]; } while (0)
        if (m->nb & NB_TOPLEFT) NBSET(3, m->mb - S->mb_w - 1, 15);
        if (m->nb & NB_TOP) for (int i = 0; i < 4; i++) NBSET(4 + i, m->mb - S->mb_w, 12 + i);
        if (m->nb & NB_TOPRIGHT) NBSET(8, m->mb - S->mb_w + 1, 12);
        if (m->nb & NB_LEFT) for (int i = 0; i < 4; i++) NBSET(11 + 8 * i, m->mb - 1, 3 + 4 * i);
#undef NBSET
    }''')
rep('''            for (int i = 0; i < 4; i++) { m->cmvd[11 + 8 * i][0] = S->mvd[(l * 16 + 3 + 4 * i) * 2]; m->cmvd[11 + 8 * i][1] = S->mvd[(l * 16 + 3 + 4 * i) * 2 + 1]; }
        }
    }''','''            for (int i = 0; i < 4; i++) { m->cmvd[11 + 8 * i][0] = S->mvd[(l * 16 + 3 + 4 * i) * 2]; m->cmvd[11 + 8 * i][1] = S->mvd[(l * 16 + 3 + 4 * i) * 2 + 1]; }
        }
        if (S->slice_type == S_SLICE_B) {                /* list 1 of the mvd cache and the skip flags of direct blocks, macroblock.c:1129-1160 */
            memset(m->cmvd1, 0, sizeof(m->cmvd1)); memset(m->cskip, 0, sizeof(m->cskip));
            if (m->nb & NB_TOP) {
                const int t = m->mb - S->mb_w, sb = S->skipbp[t];
                for (int i = 0; i < 4; i++) { m->cmvd1[4 + i][0] = S->mvd1[(t * 16 + 12 + i) * 2]; m->cmvd1[4 + i][1] = S->mvd1[(t * 16 + 12 + i) * 2 + 1]; }
                m->cskip[s_scan8(0) - 8] = sb & 4; m->cskip[s_scan8(4) - 8] = sb & 8;
            }
            if (m->nb & NB_LEFT) {
                const int l = m->mb - 1, sb = S->skipbp[l];
                for (int i = 0; i < 4; i++) { m->cmvd1[11 + 8 * i][0] = S->mvd1[(l * 16 + 3 + 4 * i) * 2]; m->cmvd1[11 + 8 * i][1] = S->mvd1[(l * 16 + 3 + 4 * i) * 2 + 1]; }
                m->cskip[s_scan8(0) - 1] = sb & 2; m->cskip[s_scan8(8) - 1] = sb & 8;
            }
        }
    }''')
# analyse_intra: cavlc mb type prefix in B slices
rep('''        if (c < m->satd_i16) { m->satd_i16 = c; m->pred16 = mode[i]; }
    }
    if (m->fast_intra && m->satd_i16 > 2 * satd_inter) return;''','''        if (c < m->satd_i16) { m->satd_i16 = c; m->pred16 = mode[i]; }
    }
    if (S->slice_type == S_SLICE_B) m->satd_i16 += S->lambda * 9;      /* i_mb_b_cost_table[I_16x16], analyse.c:659-661 */
    if (m->fast_intra && m->satd_i16 > 2 * satd_inter) return;''')
rep('''        int thresh = S->mbrd ? S_COST_MAX : satd_inter < m->satd_i16 ? satd_inter : m->satd_i16, cost = 0, idx;
        m->cbp_luma = 0;''','''        int thresh = S->mbrd ? S_COST_MAX : satd_inter < m->satd_i16 ? satd_inter : m->satd_i16, cost = 0, idx;
        if (S->slice_type == S_SLICE_B) cost += S->lambda * 9;         /* i_mb_b_cost_table[I_8x8], :676-677 */
        m->cbp_luma = 0;''')
rep('''        if (S->mbrd) thresh = thresh * (10 - m->fast_intra) / 8;
        m->cbp_luma = 0;''','''        if (S->mbrd) thresh = thresh * (10 - m->fast_intra) / 8;
        if (S->slice_type == S_SLICE_B) cost += S->lambda * 9;         /* i_mb_b_cost_table[I_4x4], :770-771 */
        m->cbp_luma = 0;''')
# fast intra for B
rep('''    if (S->slice_type == S_SLICE_P && m->mb > 4) {
        int likely = S_IS_INTRA(m->type_left) || S_IS_INTRA(m->type_top) || S_IS_INTRA(m->type_topleft) || S_IS_INTRA(m->type_topright)
                  || S_IS_INTRA(S->fref[0]->mb_type[m->mb]) || m->mb < 3 * S->intra_count;''','''    if (S->slice_type != S_SLICE_I && m->mb > 4) {
        int likely = S_IS_INTRA(m->type_left) || S_IS_INTRA(m->type_top) || S_IS_INTRA(m->type_topleft) || S_IS_INTRA(m->type_topright)
                  || (S->slice_type == S_SLICE_P && S_IS_INTRA(S->fref[0]->mb_type[m->mb])) || m->mb < 3 * S->intra_count;''')
rep('''        if (satd_pcm < i_cost) m->type = S_I_PCM;
    } else {
        int b_skip = 0, try_pskip = 0;''','''        if (satd_pcm < i_cost) m->type = S_I_PCM;
    } else if (S->slice_type == S_SLICE_B) {
        analyse_b(S, m, A, satd_pcm);
    } else {
        int b_skip = 0, try_pskip = 0;''')
# update_cache default
rep('''        m->mvx = m->pskip_mv[0]; m->mvy = m->pskip_mv[1]; m->ref = 0;
        break;
    default:
        break;
    }
}
/* x264_mb_analyse_p_rd''','''        m->mvx = m->pskip_mv[0]; m->mvy = m->pskip_mv[1]; m->ref = 0;
        break;
    case S_I_PCM:
        break;
    default:
        update_cache_b(S, m, A->B);
        break;
    }
}
/* x264_mb_analyse_p_rd''')
rep('''static void analyse_mb(ssl *S, smb *m, panalysis *A)
{''','''#include "b_oracle.c"
static void analyse_mb(ssl *S, smb *m, panalysis *A)
{''')
rep('''static void fill_part(smb *m, int x, int y, int w, int h, int ref, int mvx, int mvy)
{''','''struct banalysis;
static void update_cache_b(ssl *S, smb *m, struct banalysis *B);
static void fill_part(smb *m, int x, int y, int w, int h, int ref, int mvx, int mvy)
{''')
# encode_mb
rep('''    if (m->type == S_I_16x16) {
        m->t8 = 0;
        pred_16x16(S, m, m->i16mode);
        enc_i16x16(S, m);''','''    if (m->type == S_B_SKIP) {                   /* macroblock.c:508-515 */
        if (!m->skip_mc) mc_b(S, m);
        m->cbp_luma = m->cbp_chroma = 0;
        memset(m->nnz, 0, sizeof(m->nnz));
        return;
    }
    if (m->type == S_I_16x16) {
        m->t8 = 0;
        pred_16x16(S, m, m->i16mode);
        enc_i16x16(S, m);''')
rep('''    } else {
        if (!m->skip_mc) mc_parts(S, m);
        enc_inter_luma(S, m);
    }''','''    } else {
        if (!m->skip_mc) { if (m->type >= S_B_DIRECT) mc_b(S, m); else mc_parts(S, m); }
        enc_inter_luma(S, m);
    }''')
rep('''        && m->mv4[0][1] == m->pskip_mv[1] && m->ref8[0] == 0)
        m->type = S_P_SKIP;
}''','''        && m->mv4[0][1] == m->pskip_mv[1] && m->ref8[0] == 0)
        m->type = S_P_SKIP;
    if (m->type == S_B_DIRECT && !(m->cbp_luma | m->cbp_chroma)) m->type = S_B_SKIP;   /* macroblock.c:784-788 */
}''')
rep('''static void encode_mb(ssl *S, smb *m)
{
    m->cbp_luma = 0; m->nnz[24] = 0;''','''static void mc_b(const ssl *S, smb *m);
static void encode_mb(ssl *S, smb *m)
{
    m->cbp_luma = 0; m->nnz[24] = 0;''')
open(p,'w').write(s)
print("ok")
