import re,sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:90]); sys.exit(1)
    s=s.replace(a,b)

# new helper functions before analyse_mb
rep("""static void analyse_mb(ssl *S, smb *m, panalysis *A)
{""","""/* x264_analyse_update_cache, R/encoder/analyse.c:2777-2846 (I and P types): the candidate `m->type / m->partition` names becomes
 * the macroblock's vectors, references (h->mb.cache and what cache_save will store) or intra modes.                              */
static void fill_part(smb *m, int x, int y, int w, int h, int ref, int mvx, int mvy)
{
    cache_set(m, x, y, w, h, ref, mvx, mvy, 1);
    for (int j = y; j < y + h; j++)
        for (int i = x; i < x + w; i++) { m->mv4[j * 4 + i][0] = (i16)mvx; m->mv4[j * 4 + i][1] = (i16)mvy; m->ref8[(j >> 1) * 2 + (i >> 1)] = (int8_t)ref; }
}
static void update_cache(ssl *S, smb *m, const panalysis *A)
{
    switch (m->type) {
    case S_I_4x4:
        for (int i = 0; i < 16; i++) m->i4c[s_scan8(i)] = m->pred4[i];
        analyse_intra_chroma(S, m);
        break;
    case S_I_8x8:
        for (int i = 0; i < 16; i++) m->i4c[s_scan8(i)] = m->pred8[i >> 2];
        analyse_intra_chroma(S, m);
        break;
    case S_I_16x16:
        m->i16mode = m->pred16;
        analyse_intra_chroma(S, m);
        break;
    case S_P_L0:
        if (m->partition == S_D_16x16) fill_part(m, 0, 0, 4, 4, A->me16.ref, A->me16.mvx, A->me16.mvy);
        else if (m->partition == S_D_16x8)
            for (int i = 0; i < 2; i++) fill_part(m, 0, 2 * i, 4, 2, A->me16x8[i].ref, A->me16x8[i].mvx, A->me16x8[i].mvy);
        else
            for (int i = 0; i < 2; i++) fill_part(m, 2 * i, 0, 2, 4, A->me8x16[i].ref, A->me8x16[i].mvx, A->me8x16[i].mvy);
        break;
    case S_P_8x8:
        for (int i = 0; i < 4; i++) {                        /* x264_mb_cache_mv_p8x8, :1058-1075 */
            const int x0 = 2 * (i & 1), y0 = 2 * (i >> 1), r = A->me8[i].ref, t = A->sub[i];
            m->sub[i] = (int8_t)t;
            if (t == S_D_L0_8x8) fill_part(m, x0, y0, 2, 2, r, A->me8[i].mvx, A->me8[i].mvy);
            else if (t == S_D_L0_8x4) for (int k = 0; k < 2; k++) fill_part(m, x0, y0 + k, 2, 1, r, A->me84[i][k].mvx, A->me84[i][k].mvy);
            else if (t == S_D_L0_4x8) for (int k = 0; k < 2; k++) fill_part(m, x0 + k, y0, 1, 2, r, A->me48[i][k].mvx, A->me48[i][k].mvy);
            else for (int k = 0; k < 4; k++) fill_part(m, x0 + (k & 1), y0 + (k >> 1), 1, 1, r, A->me4[i][k].mvx, A->me4[i][k].mvy);
        }
        break;
    case S_P_SKIP:
        m->partition = S_D_16x16;
        fill_part(m, 0, 0, 4, 4, 0, m->pskip_mv[0], m->pskip_mv[1]);
        m->mvx = m->pskip_mv[0]; m->mvy = m->pskip_mv[1]; m->ref = 0;
        break;
    default:
        break;
    }
}
/* x264_mb_analyse_p_rd, :1935-2005 (sub-8x8 partitions are refused with the RD levels for now) */
static void analyse_p_rd(ssl *S, smb *m, panalysis *A, int i_satd)
{
    const int thresh = i_satd * 5 / 4;
    m->type = S_P_L0;
    if (A->rd16 == S_COST_MAX && A->me16.cost <= i_satd * 3 / 2) {
        m->partition = S_D_16x16;
        update_cache(S, m, A);
        A->rd16 = rd_cost_mb(S, m, S->lambda2);
    }
    A->me16.cost = A->rd16;
    if (A->cost16x8 <= thresh) { m->partition = S_D_16x8; update_cache(S, m, A); A->cost16x8 = rd_cost_mb(S, m, S->lambda2); }
    else A->cost16x8 = S_COST_MAX;
    if (A->cost8x16 <= thresh) { m->partition = S_D_8x16; update_cache(S, m, A); A->cost8x16 = rd_cost_mb(S, m, S->lambda2); }
    else A->cost8x16 = S_COST_MAX;
    if (A->cost8x8 <= thresh) {
        m->type = S_P_8x8; m->partition = S_D_8x8;
        update_cache(S, m, A);
        A->cost8x8 = rd_cost_mb(S, m, S->lambda2);
    } else A->cost8x8 = S_COST_MAX;
}
/* x264_intra_rd, :845-874 */
static void intra_rd(ssl *S, smb *m, const panalysis *A, int thresh)
{
    if (m->satd_i16 <= thresh) { m->type = S_I_16x16; update_cache(S, m, A); m->satd_i16 = rd_cost_mb(S, m, S->lambda2); }
    else m->satd_i16 = S_COST_MAX;
    if (m->satd_i4 <= thresh && m->satd_i4 < S_COST_MAX) { m->type = S_I_4x4; update_cache(S, m, A); m->satd_i4 = rd_cost_mb(S, m, S->lambda2); }
    else m->satd_i4 = S_COST_MAX;
    if (m->satd_i8 <= thresh && m->satd_i8 < S_COST_MAX) { m->type = S_I_8x8; update_cache(S, m, A); m->satd_i8 = rd_cost_mb(S, m, S->lambda2); }
    else m->satd_i8 = S_COST_MAX;
}
/* x264_mb_analyse_transform_rd, :2127-2150 */
static void transform_rd(ssl *S, smb *m, const panalysis *A, int *i_satd, int *i_rd)
{
    if (!s_t8_allowed(S, m) || !S->p->transform8x8) return;
    update_cache(S, m, A);
    m->t8 = !m->t8;
    const int rd8 = rd_cost_mb(S, m, S->lambda2);
    if (*i_rd >= rd8) {
        if (*i_rd > 0) *i_satd = (int)((int64_t)*i_satd * rd8 / *i_rd);
        if (*i_satd == 0) *i_satd = 1;
        *i_rd = rd8;
    } else
        m->t8 = !m->t8;
}

static void analyse_mb(ssl *S, smb *m, panalysis *A)
{""")

# I-slice branch
rep("""    if (S->slice_type == S_SLICE_I) {
        analyse_intra(S, m, S_COST_MAX);
        i_cost = m->satd_i16; m->type = S_I_16x16;
        if (m->satd_i4 < i_cost) { i_cost = m->satd_i4; m->type = S_I_4x4; }
        if (m->satd_i8 < i_cost) { i_cost = m->satd_i8; m->type = S_I_8x8; }
    } else {""","""    const int satd_pcm = !S->psy_rd && S->mbrd ? (int)(((uint64_t)(386 * 8) * S->lambda2 + 128) >> 8) : S_COST_MAX;   /* a->i_satd_pcm, :246 */
    if (S->slice_type == S_SLICE_I) {
        if (S->mbrd) cache_fenc_satd(S, m);
        analyse_intra(S, m, S_COST_MAX);
        if (S->mbrd) intra_rd(S, m, A, S_COST_MAX);
        i_cost = m->satd_i16; m->type = S_I_16x16;
        if (m->satd_i4 < i_cost) { i_cost = m->satd_i4; m->type = S_I_4x4; }
        if (m->satd_i8 < i_cost) { i_cost = m->satd_i8; m->type = S_I_8x8; }
        if (satd_pcm < i_cost) m->type = S_I_PCM;
    } else {""")

# refinement only without mbrd
rep("""            m->partition = part;
            /* x264_me_refine_qpel on the winning partition (:2289-2352; the reference cost leaves every block's sum, me.c:639-640) */
            if (part == S_D_16x16) {""","""            m->partition = part;
            A->cost8x8 = cost8x8; A->cost16x8 = cost16x8; A->cost8x16 = cost8x16;
            /* x264_me_refine_qpel on the winning partition (:2289-2352; the reference cost leaves every block's sum, me.c:639-640);
             * with the RD levels the vectors stay as the searches left them ("refine later", :2296-2299) */
            if (S->mbrd) {
            } else if (part == S_D_16x16) {""")
rep("""            int satd_inter = i_cost, satd_intra = m->satd_i16 < m->satd_i8 ? m->satd_i16 : m->satd_i8;
            if (m->satd_i4 < satd_intra) satd_intra = m->satd_i4;
            int itype = S_I_16x16, icost = m->satd_i16;
            if (m->satd_i8 < icost) { icost = m->satd_i8; itype = S_I_8x8; }
            if (m->satd_i4 < icost) { icost = m->satd_i4; itype = S_I_4x4; }
            if (icost < i_cost) { i_cost = icost; m->type = itype; }""","""            int satd_inter = i_cost, satd_intra = m->satd_i16 < m->satd_i8 ? m->satd_i16 : m->satd_i8;
            if (m->satd_i4 < satd_intra) satd_intra = m->satd_i4;
            if (S->mbrd) {                                       /* :2375-2389 */
                analyse_p_rd(S, m, A, satd_inter < satd_intra ? satd_inter : satd_intra);
                m->type = S_P_L0; part = S_D_16x16; i_cost = A->me16.cost;
                if (A->cost16x8 < i_cost) { i_cost = A->cost16x8; part = S_D_16x8; }
                if (A->cost8x16 < i_cost) { i_cost = A->cost8x16; part = S_D_8x16; }
                if (A->cost8x8 < i_cost) { i_cost = A->cost8x8; part = S_D_8x8; m->type = S_P_8x8; }
                m->partition = part;
                if (i_cost < S_COST_MAX) transform_rd(S, m, A, &satd_inter, &i_cost);
                const int keep = m->type == S_P_SKIP ? (part == S_D_8x8 ? S_P_8x8 : S_P_L0) : m->type;   /* i_type is a local of the reference: the trial's P_SKIP does not stick */
                intra_rd(S, m, A, satd_inter * 5 / 4);
                m->type = keep;
            }
            int itype = S_I_16x16, icost = m->satd_i16;
            if (m->satd_i8 < icost) { icost = m->satd_i8; itype = S_I_8x8; }
            if (m->satd_i4 < icost) { icost = m->satd_i4; itype = S_I_4x4; }
            if (satd_pcm < icost) { icost = satd_pcm; itype = S_I_PCM; }
            if (icost < i_cost) { i_cost = icost; m->type = itype; }""")
open(p,'w').write(s)
print('ok')
