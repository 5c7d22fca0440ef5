import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)

rep('''#include "frame_internal.h"
''','''#include "frame_internal.h"
#include "trellis_dev.h"
''')
rep('''// lanes exchange data through LDS only''','''// ---- round 2: what the raster-order variant of the sweep (RD levels, trellis, adaptive quantisation, the entropy coder) adds ----
struct SwRd {                       // kernel argument
    int on;                         // this launch is the raster variant
    int mbrd, trellis, psy_rd;      // a->i_mbrd, param.analyse.i_trellis, h->mb.i_psy_rd
    int write, cabac_init_idc, i_frame;
    int aq, qp_min, qp_max;
    float f_qpm;
    const float *aq_offset;         // [batch][n_mb]
    const i16 *cost_mv_all;         // [52][2 * cost_center + 1]: p_cost_mv of every QP
    const int *unq4, *unq8;         // h->unquant4_mf [4][52][16], h->unquant8_mf [2][52][64]
    u8 *payload; int payload_cap; int *payload_len, *mb_bits;
    i16 *mvd;                       // h->mb.mvd[0]: [batch][n_mb][16][2]
};
struct SwLdsRd {
    u8 cabac[460], cabac_tmp[460];  // h->cabac.state and the RD trial's copy (COPY_CABAC, R/encoder/rdo.c:62)
    // what the entropy coder reads beyond SwLds (MbSynDev below points into both)
    signed char cref[48], sub[4];
    i16 cmv[48][2], cmvd[48][2];
    u8 nz_l[4], nz_t[4], nz_lc[2][2], nz_tc[2][2];
    i16 i4_dct[256], i8_dct[256];   // h->mb.pic.i4x4_dct_buf / i8x8_dct_buf (i_skip_intra == 2)
    int fenc_satd[16], fenc_sa8d[4];   // h->mb.pic.fenc_satd / fenc_sa8d (psy-RD)
    int unq4[4][16], unq8[2][64];   // unquant rows of the current QPs
    i16 left_mvd[4][2];             // the left macroblock's mvd of blocks 3, 7, 11, 15
    u8 left_nz[8];                  // its non_zero_count of blocks 5 7 13 15 | U 1 3 | V 1 3
    TrellisScratch ts;
};
struct SwLdsNone { int unused; };
// trellis context handed to the quantising helpers: on = 0 -> plain dead-zone quantisation
struct SwTq { int on; SwLdsRd *r; };
// x264_dct4_weight2_zigzag[0] / x264_dct8_weight2_zigzag[0] (R/common/dct.c:476-483) and x264_zigzag_scan4[0]
static __device__ const int d_w4z[16] = {800, 320, 320, 800, 128, 800, 320, 128, 128, 320, 320, 800, 128, 320, 320, 128};
static __device__ const u8 d_zz4[16] = {0, 4, 1, 2, 5, 8, 12, 9, 6, 3, 7, 10, 13, 14, 11, 15};
static __device__ const u8 d_zz2[4] = {0, 1, 2, 3};
static __device__ const u16 d_w8k[6] = {256, 201, 656, 227, 410, 363};
static __device__ const u8 d_w8cls[16] = {0, 3, 4, 3, 3, 1, 5, 1, 4, 5, 2, 5, 3, 1, 5, 1};
__device__ __forceinline__ int sw_w8z(int pos) { const int r = c_scan8[0][pos]; return d_w8k[d_w8cls[((r >> 1) & 12) | (r & 3)]]; }
struct SwW8 { __device__ __forceinline__ int operator[](int pos) const { return sw_w8z(pos); } };

// lanes exchange data through LDS only''')
open(p,'w').write(s)
print("ok")
