import sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:90]); sys.exit(1)
    s=s.replace(a,b)
rep('''#include "cabac_oracle.c"
#include "rd_oracle.c"
''','''#include "cabac_oracle.c"
#ifdef X264O_DEVCHECK
/* Cross-check of the PRODUCT's scalar device code, built for the host (oracle/devcheck.cpp): every call of the twin's CABAC
 * writer / bit counter and trellis quantiser is replayed through x264_vs2008_amd/csrc/cabac_dev.h / trellis_dev.h on a copy
 * of the same state and compared.  g_devcheck_bad counts differences (tests/test_devhost.py expects 0).                    */
#include "../x264_vs2008_amd/csrc/mbsyn.h"
void devhost_cw_macroblock(DCabac *cb, uint8_t *st, int rd, MbSyn *m, const uint8_t *fe, int i_frame);
void devhost_mb_skip(DCabac *cb, uint8_t *st, int type_left, int type_top, int b_skip);
void devhost_terminal(DCabac *cb);
void devhost_flush(DCabac *cb, int i_frame);
void devhost_context_init(uint8_t *st, int slice_type, int qp, int model);
int devhost_trellis(int16_t *dct, const uint16_t *mf, const int *unq, const int *weight, const uint8_t *zz, const uint8_t *st,
                    int cat, int lambda2, int b_ac, int dc, int n_coef);
int g_devcheck_bad, g_devcheck_calls;
int x264o_devcheck_bad(void) { return g_devcheck_bad; }
int x264o_devcheck_calls(void) { return g_devcheck_calls; }
static void mbsyn_fill(const ssl *S, const smb *m, MbSyn *y)
{
    memset(y, 0, sizeof(*y));
    y->slice_type = S->slice_type; y->type = m->type; y->partition = m->partition;
    y->i16mode = m->i16mode; y->chroma_mode = m->chroma_mode; y->cbp_luma = m->cbp_luma; y->cbp_chroma = m->cbp_chroma; y->t8 = m->t8; y->qp = m->qp;
    y->n_ref = S->n_ref; y->pps_t8 = S->p->transform8x8; y->t8_allowed = s_t8_allowed(S, m);
    y->type_left = m->type_left; y->type_top = m->type_top; y->cbp_left = m->cbp_left; y->cbp_top = m->cbp_top;
    y->cpm_left = m->cpm_left; y->cpm_top = m->cpm_top; y->nb_t8 = m->nb_t8;
    y->last_qp = S->last_qp; y->last_dqp = S->last_dqp;
    y->prev_coded = m->mb > 0 && (S->fdec->mb_type[S->prev_mb] == S_I_16x16 || (S->cbp[S->prev_mb] & 0x3f));
    memcpy(y->sub, m->sub, 4); memcpy(y->i4c, m->i4c, 48); memcpy(y->cref, m->cref, 48);
    memcpy(y->cmv, m->cmv, sizeof(y->cmv)); memcpy(y->cmvd, m->cmvd, sizeof(y->cmvd));
    memcpy(y->nnz, m->nnz, 27);
    memcpy(y->nz_l, m->nz_l, 4); memcpy(y->nz_t, m->nz_t, 4); memcpy(y->nz_lc, m->nz_lc, 4); memcpy(y->nz_tc, m->nz_tc, 4);
    memcpy(y->lv4, m->luma4, sizeof(y->lv4)); memcpy(y->lv8, m->luma8, sizeof(y->lv8)); memcpy(y->lv_dc, m->dc16, sizeof(y->lv_dc));
    memcpy(y->lv_cdc, m->cdc, sizeof(y->lv_cdc)); memcpy(y->lv_cac, m->cac, sizeof(y->lv_cac));
}
static void cw_macroblock_chk(ssl *S, o_cabac *cb, int rd, smb *m)
{
    MbSyn y;
    DCabac d = {cb->low, cb->range, cb->queue, cb->outstanding, 0, cb->f8};
    u8 st[460], out[64 + 1024], fe[384];
    memcpy(st, cb->state, 460);
    memset(out, 0, sizeof(out));
    d.p = out + 64;
    mbsyn_fill(S, m, &y);
    memcpy(fe, m->fe[0], 256);
    for (int pl = 1; pl < 3; pl++) for (int i = 0; i < 8; i++) memcpy(fe + 256 + 64 * (pl - 1) + 8 * i, m->fe[pl] + i * FENC, 8);
    u8 *p0 = cb->p;
    cw_macroblock(S, cb, rd, m);
    devhost_cw_macroblock(&d, st, rd, &y, fe, cb->i_frame);
    g_devcheck_calls++;
    int bad = memcmp(st, cb->state, 460) != 0 || y.qp != m->qp || memcmp(y.cmvd, m->cmvd, sizeof(y.cmvd)) != 0;
    if (rd) bad |= d.f8 != cb->f8;
    else {
        const int n = (int)(cb->p - p0);
        bad |= d.low != cb->low || d.range != cb->range || d.queue != cb->queue || d.outstanding != cb->outstanding || (int)(d.p - (out + 64)) != n
            || memcmp(out + 64, p0, n > 0 ? n : 0) != 0;
    }
    if (bad) { if (!g_devcheck_bad) fprintf(stderr, "devcheck: cw_macroblock differs (rd %d, frame %d, mb %d, type %d)\\n", rd, S->f, m->mb, m->type); g_devcheck_bad++; }
}
#define cw_macroblock cw_macroblock_chk
#endif
#include "rd_oracle.c"
''')
open(p,'w').write(s)

p='/root/repo/oracle/rd_oracle.c'
s=open(p).read()
a='''static int trellis_quant(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                         int cat, int lambda2, int b_ac, int dc, int n_coef)
{'''
assert s.count(a)==1
s=s.replace(a,'''static int trellis_quant_twin(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                              int cat, int lambda2, int b_ac, int dc, int n_coef);
static int trellis_quant(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                         int cat, int lambda2, int b_ac, int dc, int n_coef)
{
#ifdef X264O_DEVCHECK
    i16 copy[64];
    memcpy(copy, dct, n_coef * sizeof(i16));
    const int r2 = devhost_trellis(copy, mf, unq, weight, zz, S->cb.state, cat, lambda2, b_ac, dc, n_coef);
    const int r1 = trellis_quant_twin(S, dct, mf, unq, weight, zz, cat, lambda2, b_ac, dc, n_coef);
    g_devcheck_calls++;
    if (r1 != r2 || memcmp(copy, dct, n_coef * sizeof(i16))) { if (!g_devcheck_bad) fprintf(stderr, "devcheck: trellis differs (cat %d, n %d)\\n", cat, n_coef); g_devcheck_bad++; }
    return r1;
#else
    return trellis_quant_twin(S, dct, mf, unq, weight, zz, cat, lambda2, b_ac, dc, n_coef);
#endif
}
static int trellis_quant_twin(const ssl *S, i16 *dct, const u16 *mf, const int *unq, const int *weight, const u8 *zz,
                              int cat, int lambda2, int b_ac, int dc, int n_coef)
{''')
open(p,'w').write(s)

p='/root/repo/oracle/Makefile'
s=open(p).read()
s=s.replace('''ref: _ref/libx264ref.so _ref/libframe_ref.so''','''# the twin with every CABAC / trellis call replayed through the PRODUCT's scalar device code compiled for the host
# (x264_vs2008_amd/csrc/cabac_dev.h, trellis_dev.h via devcheck.cpp): tests/test_devhost.py
devcheck: libdevcheck.so
libdevcheck.so: $(OSRC) slice_oracle.c cabac_oracle.c rd_oracle.c devcheck.cpp ../x264_vs2008_amd/csrc/cabac_dev.h ../x264_vs2008_amd/csrc/trellis_dev.h ../x264_vs2008_amd/csrc/mbsyn.h
	g++ -O2 -fPIC -std=c++17 -Wall -Wno-unused-function -c devcheck.cpp -o devcheck.o
	$(CC) $(OFLAGS) -DX264O_DEVCHECK -shared -o $@ $(OSRC) devcheck.o -lm -lstdc++

ref: _ref/libx264ref.so _ref/libframe_ref.so''')
s=s.replace("liboracle.so: $(OSRC) slice_oracle.c ../include/x264hip_tables.h","liboracle.so: $(OSRC) slice_oracle.c cabac_oracle.c rd_oracle.c cabac_tables.h ../include/x264hip_tables.h")
s=s.replace(".PHONY: all ref clean",".PHONY: all ref clean devcheck")
open(p,'w').write(s)
print('ok')
