import re,sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:80]); sys.exit(1)
    s=s.replace(a,b)

# o_cabac typedef before ssl
rep("""typedef struct {
    const slice_params *p;
    int mb_w, mb_h, n, sy, sc, w16, h16;""","""typedef struct {                             /* x264_cabac_t, R/common/cabac.h:27-46 */
    int low, range, queue, outstanding;
    u8 *p, *start, *end;
    int f8;                                  /* f8_bits_encoded */
    int i_frame;                             /* frames coded before this one (x264_cabac_encode_flush's padding bit) */
    u8 state[460];
} o_cabac;

typedef struct {
    const slice_params *p;
    int mb_w, mb_h, n, sy, sc, w16, h16;""")
rep("""    u8 *bsbuf; int i_skip;""","""    u8 *bsbuf; int i_skip;
    o_cabac cb;                              /* h->cabac */""")

# quant dispatch helpers before enc_i4x4
rep("""static void enc_i4x4(ssl *S, smb *m, int idx)
{""","""/* x264_quant_4x4 / x264_quant_8x8 (R/encoder/macroblock.c:87-103) and the DC calls: plain dead-zone quantisation or trellis */
static int trellis_quant(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                         int cat, int lambda2, int b_ac, int dc, int n_coef);
static const int s_trellis_lambda2[2][52];
static int q4(const ssl *S, i16 d[4][4], int qcat, int ctxcat, int b_intra, int qp)
{
    if (S->b_trellis)
        return trellis_quant(S, &d[0][0], S->mf4[qcat], S->unq4[qcat], S->w4z, S->zz4, ctxcat, s_trellis_lambda2[b_intra][qp],
                             ctxcat == 1 || ctxcat == 4, 0, 16);
    return quantf.quant_4x4(d, (u16 *)S->mf4[qcat], (u16 *)S->b4[qcat]);
}
static int q8(const ssl *S, i16 d[8][8], int qcat, int b_intra, int qp)
{
    if (S->b_trellis)
        return trellis_quant(S, &d[0][0], S->mf8[qcat], S->unq8[qcat], S->w8z, S->zz8, 5, s_trellis_lambda2[b_intra][qp], 0, 0, 64);
    return quantf.quant_8x8(d, (u16 *)S->mf8[qcat], (u16 *)S->b8[qcat]);
}
static int qdc(const ssl *S, i16 *d, int qcat, int ctxcat, int b_intra, int qp)
{   /* x264_quant_dc_trellis (rdo.c:632-639) or quant_4x4_dc / quant_2x2_dc */
    static const u8 zz2[4] = {0, 1, 2, 3};
    if (S->b_trellis)
        return trellis_quant(S, d, S->mf4[qcat], S->unq4[qcat], 0, ctxcat == 3 ? zz2 : S->zz4, ctxcat, s_trellis_lambda2[b_intra][qp], 0, 1, ctxcat == 3 ? 4 : 16);
    if (ctxcat == 3) return quantf.quant_2x2_dc((i16 (*)[2])d, S->mf4[qcat][0] >> 1, S->b4[qcat][0] << 1);
    return quantf.quant_4x4_dc((i16 (*)[4])d, S->mf4[qcat][0] >> 1, S->b4[qcat][0] << 1);
}

static void enc_i4x4(ssl *S, smb *m, int idx)
{""")
rep("""    dctf.sub4x4_dct(d, src, dst);
    int nz = quantf.quant_4x4(d, S->mf4[0], S->b4[0]);
    m->nnz[idx] = nz;""","""    dctf.sub4x4_dct(d, src, dst);
    int nz = q4(S, d, 0, 2, 1, S->qp);
    m->nnz[idx] = nz;""")
rep("""    dctf.sub8x8_dct8(d, src, dst);
    int nz = quantf.quant_8x8(d, S->mf8[0], S->b8[0]);""","""    dctf.sub8x8_dct8(d, src, dst);
    int nz = q8(S, d, 0, 1, S->qp);""")
rep("""        nz = quantf.quant_4x4(d[i], S->mf4[0], S->b4[0]);
        m->nnz[i] = nz;""","""        nz = q4(S, d[i], 0, 1, 1, S->qp);
        m->nnz[i] = nz;""")
rep("""    nz = quantf.quant_4x4_dc(dc, S->mf4[0][0] >> 1, S->b4[0][0] << 1);
    m->nnz[24] = nz;""","""    nz = qdc(S, &dc[0][0], 0, 0, 1, S->qp);
    m->nnz[24] = nz;""")
rep("""            int nz = quantf.quant_4x4(d4[i], S->mf4[cat], S->b4[cat]);
            m->nnz[16 + 4 * ch + i] = nz;""","""            int nz = q4(S, d4[i], cat, 4, !b_inter, qpc);
            m->nnz[16 + 4 * ch + i] = nz;""")
rep("""        int nz_dc = quantf.quant_2x2_dc(d2, S->mf4[cat][0] >> 1, S->b4[cat][0] << 1);""","""        int nz_dc = qdc(S, &d2[0][0], cat, 3, !b_inter, qpc);""")
rep("""        i16 d8[4][8][8];
        dctf.sub16x16_dct8(d8, m->fe[0], m->fd[0]);""","""        i16 d8[4][8][8];
        b_decimate &= !S->b_trellis;                     /* "8x8 trellis is inherently optimal decimation", macroblock.c:630 */
        dctf.sub16x16_dct8(d8, m->fe[0], m->fd[0]);""")
rep("""            int nz = quantf.quant_8x8(d8[idx], S->mf8[1], S->b8[1]);""","""            int nz = q8(S, d8[idx], 1, 0, S->qp);""")
rep("""                int idx = 4 * i8 + i4, nz = quantf.quant_4x4(d4[idx], S->mf4[1], S->b4[1]);""","""                int idx = 4 * i8 + i4, nz = q4(S, d4[idx], 1, 2, 0, S->qp);""")
open(p,'w').write(s)

p='/root/repo/oracle/cabac_oracle.c'
s=open(p).read()
a=s.index("typedef struct {\n    int low, range")
b=s.index("} o_cabac;")+len("} o_cabac;\n")
s=s[:a]+s[b:]
open(p,'w').write(s)
p='/root/repo/oracle/rd_oracle.c'
s=open(p).read()
s=s.replace("static const int s_trellis_lambda2[2][52] = {  ","static const int s_trellis_lambda2[2][52] = {  ")
open(p,'w').write(s)
print('ok')
