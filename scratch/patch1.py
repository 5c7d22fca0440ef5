import re,sys
p='/root/repo/oracle/slice_oracle.c'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:80]); sys.exit(1)
    s=s.replace(a,b)

# ---- (1) ext params / outputs
rep("""/* mb types / partitions / slice types""","""/* round 2: the twin of refslice_ext / refslice_out2 (oracle/ref_slice.c) */
typedef struct {
    int trellis;
    float psy_rd, psy_trellis;
    int aq_mode; float aq_strength;
    int write, payload_cap, cabac_init_idc;
} slice_ext;
typedef struct {
    u8 *payload;
    int32_t *payload_len, *mb_bits;
    float *qp_offset;
} slice_out2;

/* mb types / partitions / slice types""")

# ---- (2) ssl
rep("""    int lossless;                            /* h->mb.b_lossless: constant QP 0 (R/encoder/encoder.c:401-421) */""",
"""    int lossless;                            /* h->mb.b_lossless: constant QP 0 (R/encoder/encoder.c:401-421) */
    /* round 2: RD levels, trellis, the entropy coder, per-macroblock QP */
    const slice_ext *e;
    slice_out2 *o2;
    int mbrd, psy_rd, trellis, b_trellis;    /* a->i_mbrd, h->mb.i_psy_rd, param i_trellis, h->mb.b_trellis (what the encode functions obey now) */
    int chroma_qp_offset;                    /* after x264_validate_parameters' psy adjustment */
    int frame_qp, qp_min, qp_max;            /* rc->qpm of the frame; param.rc.i_qp_min / max */
    float f_qpm, *aq_offset;                 /* rc->f_qpm; fenc->f_qp_offset[n] (adaptive quantisation) */
    int last_qp, last_dqp, prev_mb;          /* h->mb.i_last_qp / i_last_dqp / i_mb_prev_xy */
    i16 *cbp;                                /* h->mb.cbp[n] */
    int8_t *chroma_pm;                       /* h->mb.chroma_pred_mode[n] ("fixed" numbering, DC for anything not intra) */
    i16 *mvd;                                /* h->mb.mvd[0]: [n][16][2] */
    int8_t *qp_mb;                           /* h->mb.qp[n] */
    int unq4[4][16], unq8[2][64];            /* h->unquant4_mf / unquant8_mf at the current QPs */
    u8 zz4[16], zz8[64]; int w4z[16], w8z[64];   /* x264_zigzag_scan4/8[0], x264_dct4/8_weight2_zigzag[0] */
    u8 *bsbuf; int i_skip;""")

# ---- (3) smb
rep("""    u8 i4_fdec[256], i8_fdec[256], i4_nnz[16], i8_nnz[16];
    int i4_cbp, i8_cbp;
} smb;""","""    u8 i4_fdec[256], i8_fdec[256], i4_nnz[16], i8_nnz[16];
    int i4_cbp, i8_cbp;
    /* round 2 */
    int qp;                              /* h->mb.i_qp */
    int skip_intra;                      /* h->mb.i_skip_intra */
    i16 i4_dct[16][16], i8_dct[4][64];   /* h->mb.pic.i4x4_dct_buf / i8x8_dct_buf (i_skip_intra == 2) */
    int cbp_left, cbp_top, cpm_left, cpm_top, nb_t8;   /* cache.i_cbp_left / top (-1: none), neighbours' chroma modes, i_neighbour_transform_size */
    u8 nz_l[4], nz_t[4], nz_lc[2][2], nz_tc[2][2];      /* neighbours' non_zero_count next to this macroblock (0x80: none) */
    i16 cmvd[48][2];                     /* h->mb.cache.mvd[0], x264_scan8 layout */
    int fenc_satd[4][4], fenc_sa8d[2][2], fenc_satd_sum, fenc_sa8d_sum;   /* h->mb.pic.fenc_satd ... (psy-RD) */
} smb;

/* what x264_mb_analysis_t keeps of the P analysis (R/encoder/analyse.c:42-137) */
typedef struct { int mvx, mvy, cost, cost_mv, ref, ref_cost; i16 mvp[2]; } pme;
typedef struct { int mvx, mvy, cost; i16 mvp[2]; } sub_me;
typedef struct {
    pme me16, me8[4], me16x8[2], me8x16[2];
    sub_me me4[4][4], me84[4][2], me48[4][2];
    int sub[4];
    int cost8x8, cost16x8, cost8x16, rd16;
} panalysis;""")
rep("""typedef struct { int mvx, mvy, cost; i16 mvp[2]; } sub_me;
static int p4x4_chroma""","""static int p4x4_chroma""")

# ---- (4) includes before refine_qpel16
rep("""/* x264_me_refine_qpel -> refine_subpel(.., b_refine_qpel = 1), R/encoder/me.c:634-778, 16x16 */""",
"""/* x264_mb_transform_8x8_allowed (R/common/macroblock.h:452-466): large P partitions, P_8x8 only with four 8x8 sub-partitions */
static int s_t8_allowed(const ssl *S, const smb *m)
{
    if (!S->p->transform8x8) return 0;
    if (m->type == S_P_L0) return 1;
    return m->type == S_P_8x8 && m->sub[0] == S_D_L0_8x8 && m->sub[1] == S_D_L0_8x8 && m->sub[2] == S_D_L0_8x8 && m->sub[3] == S_D_L0_8x8;
}
#include "cabac_oracle.c"
#include "rd_oracle.c"

/* x264_me_refine_qpel -> refine_subpel(.., b_refine_qpel = 1), R/encoder/me.c:634-778, 16x16 */""")
open(p,'w').write(s)
print("ok")
