// mbsyn.h -- the two plain records the CABAC code of the slice kernel works on (valid C and C++: the CPU-side tests fill them too)
#pragma once
#include <stdint.h>

typedef struct DCabac {         // x264_cabac_t's coder registers (R/common/cabac.h:27-46); the context states are passed separately
    int low, range, queue, outstanding;
    uint8_t *p;                 // next output byte (global memory)
    int f8;                     // f8_bits_encoded (bit counting only)
} DCabac;

// everything x264_macroblock_write_cabac reads of the macroblock and its neighbours (h->mb, h->mb.cache, h->dct)
typedef struct MbSyn {
    int slice_type;             // 0 P, 1 B, 2 I
    int type, partition;        // T_* / D_* with the reference's numbering
    int i16mode, chroma_mode, cbp_luma, cbp_chroma, t8, qp;
    int n_ref, n_ref1;          // h->mb.pic.i_fref[0] / [1]
    int pps_t8, t8_allowed;     // pps->b_transform_8x8_mode, x264_mb_transform_8x8_allowed
    int type_left, type_top;    // -1: not available
    int cbp_left, cbp_top;      // h->mb.cache.i_cbp_left / top, -1: not available
    int cpm_left, cpm_top;      // neighbours' chroma_pred_mode ("fixed", 0 for anything not intra)
    int nb_t8;                  // h->mb.cache.i_neighbour_transform_size
    int last_qp, last_dqp, prev_coded;   // h->mb.i_last_qp / i_last_dqp; type[prev] == I_16x16 || cbp[prev] & 0x3f
    signed char sub[4];         // h->mb.i_sub_partition
    signed char i4c[48];        // intra4x4_pred_mode cache, x264_scan8 layout
    signed char cref[48];       // h->mb.cache.ref[0]
    int16_t cmv[48][2], cmvd[48][2];   // h->mb.cache.mv[0] / mvd[0]
    signed char cref1[48];      // list 1 of the same (B slices)
    int16_t cmv1[48][2], cmvd1[48][2];
    signed char cskip[48];      // h->mb.cache.skip: direct blocks, whose references do not count in a reference index's context
    uint8_t nnz[28];            // this macroblock's non_zero_count: 0..15 luma, 16..23 chroma AC, 24 luma DC, 25 / 26 chroma DC
    uint8_t nz_l[4], nz_t[4], nz_lc[2][2], nz_tc[2][2];   // the neighbours' counts next to it, 0x80: none
    int16_t lv4[16][16], lv8[4][64], lv_dc[16], lv_cdc[2][4], lv_cac[8][16];   // h->dct
} MbSyn;
