/* ref_shim.c -- TEST INFRASTRUCTURE.  Compiled only into
 * oracle/_ref/libx264ref.so, against the reference's own headers where they
 * lie (-I$(REF)).  It contains no reference code: it only builds the x264_t
 * state the reference's x264_cqm_init() (R/common/set.c:68-168) expects and
 * hands the resulting quantiser tables back as flat arrays, so that
 * oracle/gen_golden.py can store them as golden vectors.
 */
#include "common/common.h"

static x264_t *g_h;
static const uint8_t flat16[64] = {
    16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,
    16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16,16 };

/* deadzones as x264_param_default sets them (R/common/common.c:129-130) */
int refshim_cqm_flat_init(void)
{
    int i;
    if (g_h)
        return 0;
    g_h = calloc(1, sizeof(x264_t));
    if (!g_h)
        return -1;
    g_h->pps = &g_h->pps_array[0];
    for (i = 0; i < 6; i++)
        g_h->pps->scaling_list[i] = flat16;
    g_h->param.analyse.i_luma_deadzone[0] = 21;
    g_h->param.analyse.i_luma_deadzone[1] = 11;
    g_h->param.analyse.b_transform_8x8 = 1;
    g_h->param.rc.i_qp_min = 0;
    return x264_cqm_init(g_h);
}
/* cat: 0 intra-Y 1 inter-Y 2 intra-C 3 inter-C (4x4); 0 intra-Y 1 inter-Y (8x8) */
const uint16_t *refshim_quant4_mf(int cat, int qp)   { return g_h->quant4_mf[cat][qp]; }
const uint16_t *refshim_quant4_bias(int cat, int qp) { return g_h->quant4_bias[cat][qp]; }
const uint16_t *refshim_quant8_mf(int cat, int qp)   { return g_h->quant8_mf[cat][qp]; }
const uint16_t *refshim_quant8_bias(int cat, int qp) { return g_h->quant8_bias[cat][qp]; }
const int *refshim_dequant4_mf(int cat) { return &g_h->dequant4_mf[cat][0][0][0]; }
const int *refshim_dequant8_mf(int cat) { return &g_h->dequant8_mf[cat][0][0][0]; }
