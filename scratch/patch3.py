import re,sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:80]); sys.exit(1)
    s=s.replace(a,b)

# analyse_intra: i8x8 threshold
rep("""        int thresh = satd_inter < m->satd_i16 ? satd_inter : m->satd_i16, cost = 0, idx;
        m->cbp_luma = 0;
        for (idx = 0;; idx++) {
            int x = idx & 1, y = idx >> 1, best = S_COST_MAX, pm = pred_intra4x4_mode(m, 4 * idx);""",
"""        int thresh = S->mbrd ? S_COST_MAX : satd_inter < m->satd_i16 ? satd_inter : m->satd_i16, cost = 0, idx;
        m->cbp_luma = 0;
        for (idx = 0;; idx++) {
            int x = idx & 1, y = idx >> 1, best = S_COST_MAX, pm = pred_intra4x4_mode(m, 4 * idx);""")
rep("""            m->satd_i8 = cost;
            for (int r = 0; r < 16; r++) memcpy(m->i8_fdec + 16 * r, m->fd[0] + r * FDEC, 16);
            memcpy(m->i8_nnz, m->nnz, 16); m->i8_cbp = m->cbp_luma;""","""            m->satd_i8 = cost;
            if (m->skip_intra) {
                for (int r = 0; r < 16; r++) memcpy(m->i8_fdec + 16 * r, m->fd[0] + r * FDEC, 16);
                memcpy(m->i8_nnz, m->nnz, 16); m->i8_cbp = m->cbp_luma;
                if (m->skip_intra == 2) memcpy(m->i8_dct, m->luma8, sizeof(m->i8_dct));
            }""")
rep("""        if ((cost < m->satd_i16 ? cost : m->satd_i16) > satd_inter * 5 / 4) return;""",
"""        if ((cost < m->satd_i16 ? cost : m->satd_i16) > satd_inter * (5 + !!S->mbrd) / 4) return;""")
rep("""        if (m->satd_i8 < thresh) thresh = m->satd_i8;
        m->cbp_luma = 0;""","""        if (m->satd_i8 < thresh) thresh = m->satd_i8;
        if (S->mbrd) thresh = thresh * (10 - m->fast_intra) / 8;
        m->cbp_luma = 0;""")
rep("""            m->satd_i4 = cost;
            for (int r = 0; r < 16; r++) memcpy(m->i4_fdec + 16 * r, m->fd[0] + r * FDEC, 16);
            memcpy(m->i4_nnz, m->nnz, 16); m->i4_cbp = m->cbp_luma;""","""            m->satd_i4 = cost;
            if (m->skip_intra) {
                for (int r = 0; r < 16; r++) memcpy(m->i4_fdec + 16 * r, m->fd[0] + r * FDEC, 16);
                memcpy(m->i4_nnz, m->nnz, 16); m->i4_cbp = m->cbp_luma;
                if (m->skip_intra == 2) memcpy(m->i4_dct, m->luma4, sizeof(m->i4_dct));
            }""")

# encode_mb: skip_intra
rep("""    } else if (S->lossless && (m->type == S_I_8x8 || m->type == S_I_4x4)) {
        /* i_skip_intra = 0 (analyse.c:250): nothing of the analysis' trial encode is kept, every block is predicted and coded again */""",
"""    } else if (!m->skip_intra && (m->type == S_I_8x8 || m->type == S_I_4x4)) {
        /* i_skip_intra = 0 (lossless, analyse.c:250; trellis 1 or --nr, :2772): nothing of the analysis' trial encode is kept, every block is predicted and coded again */""")
rep("""        for (int r = 0; r < 16; r++) memcpy(m->fd[0] + r * FDEC, m->i8_fdec + 16 * r, 16);
        memcpy(m->nnz, m->i8_nnz, 16); m->cbp_luma = m->i8_cbp;""","""        for (int r = 0; r < 16; r++) memcpy(m->fd[0] + r * FDEC, m->i8_fdec + 16 * r, 16);
        memcpy(m->nnz, m->i8_nnz, 16); m->cbp_luma = m->i8_cbp;
        if (m->skip_intra == 2) memcpy(m->luma8, m->i8_dct, sizeof(m->i8_dct));   /* "In RD mode, restore the now-overwritten DCT data", macroblock.c:543 */""")
rep("""        for (int r = 0; r < 16; r++) memcpy(m->fd[0] + r * FDEC, m->i4_fdec + 16 * r, 16);
        memcpy(m->nnz, m->i4_nnz, 16); m->cbp_luma = m->i4_cbp;""","""        for (int r = 0; r < 16; r++) memcpy(m->fd[0] + r * FDEC, m->i4_fdec + 16 * r, 16);
        memcpy(m->nnz, m->i4_nnz, 16); m->cbp_luma = m->i4_cbp;
        if (m->skip_intra == 2) memcpy(m->luma4, m->i4_dct, sizeof(m->i4_dct));""")
open(p,'w').write(s)
print('ok')
