// devcheck.cpp -- TEST INFRASTRUCTURE.  Host build of the scalar (one-lane) device code of the slice kernel -- the CABAC
// writer / bit counter of x264_vs2008_amd/csrc/cabac_dev.h and the trellis quantiser of trellis_dev.h -- so that the CPU twin
// (oracle/slice_oracle.c built with -DX264O_DEVCHECK) can replay every call it makes through the product's own text and
// compare: same context states, same bit counts, same bytes.  No GPU involved; the lane-parallel parts are tested on the GPU.
#define X264HIP_HOST_TEST 1
#include "../x264_vs2008_amd/csrc/cabac_dev.h"
#include "../x264_vs2008_amd/csrc/trellis_dev.h"

extern "C" void devhost_context_init(uint8_t *st, int slice_type, int qp, int model)
{
    for (int i = 0; i < 460; i++) st[i] = (uint8_t)cd_context_init_one(i, slice_type, qp, model);
}
extern "C" void devhost_cw_macroblock(DCabac *cb, uint8_t *st, int rd, MbSyn *m, const uint8_t *fe, int i_frame) { cw_macroblock(*cb, st, rd, *m, fe, i_frame); }
extern "C" void devhost_mb_skip(DCabac *cb, uint8_t *st, int type_left, int type_top, int b_skip) { cw_mb_skip(*cb, st, type_left, type_top, b_skip); }
extern "C" void devhost_terminal(DCabac *cb) { cd_encode_terminal(*cb); }
extern "C" void devhost_flush(DCabac *cb, int i_frame) { cd_encode_flush(*cb, i_frame); }
extern "C" int devhost_trellis(int16_t *dct, const uint16_t *mf, const int *unq, const int *weight, const uint8_t *zz, const uint8_t *st,
                               int cat, int lambda2, int b_ac, int dc, int n_coef)
{
    TrellisScratch ts;
    return td_trellis_quant(ts, dct, mf, unq, weight, zz, st, cat, lambda2, b_ac, dc, n_coef);
}

// the RD-only partial writers: kind 0 partition_size(i8, pix), 1 subpartition_size(i4, pix), 2 partition_i8x8_size(i8, mode),
// 3 partition_i4x4_size(i4, mode), 4 i8x8_chroma_size
extern "C" void devhost_cw_part(DCabac *cb, uint8_t *st, MbSyn *m, int kind, int a, int b)
{
    if (kind == 0) cw_partition_size(*cb, st, *m, a, b);
    else if (kind == 1) cw_subpartition_size(*cb, st, *m, a, b);
    else if (kind == 2) cw_partition_i8x8_size(*cb, st, *m, a, b);
    else if (kind == 3) cw_partition_i4x4_size(*cb, st, *m, a, b);
    else cw_i8x8_chroma_size(*cb, st, *m);
}
