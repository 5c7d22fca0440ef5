import sys,re
def patch(path, pairs):
    s=open(path).read()
    for a,b in pairs:
        n=s.count(a)
        if n!=1:
            print("MISMATCH",n,path,a[:80]); sys.exit(1)
        s=s.replace(a,b)
    open(path,'w').write(s)

patch('/root/repo/x264_vs2008_amd/csrc/runtime.hip', [
('''    (void)hipMemset(p, 0, bytes);
    return p;
}''','''    (void)hipMemset(p, 0, bytes);
    (void)hipDeviceSynchronize();      // the frame contexts' streams are non-blocking: nothing may touch the buffer before the clear has landed
    return p;
}

// ---- host tables whose arithmetic is floating point in the reference: built here, in C, with the reference's expression and the
// build's -ffp-contract=off (a NumPy twin of this lives in x264_vs2008_amd/frame.py only as a cross-check) ----
// p_cost_mv (x264_mb_analyse_load_costs, R/encoder/analyse.c:182-198; its log2f is the macro of analyse.c:40): out[span + i] =
// out[span - i] = (int16)(lambda * (log2f(i + 1) * 2 + 0.718f + !!i) + .5f), i = 0 .. span
extern "C" void x264hip_cost_mv_table(int lambda, int span, int16_t *out)
{
    for (int i = 0; i <= span; i++)
        out[span - i] = out[span + i] = (int16_t)(lambda * (((float)log((double)(i + 1))) / (log((double)2)) * 2 + 0.718f + !!i) + .5f);
}
// h->unquant4_mf / unquant8_mf (x264_cqm_init, R/common/set.c:146,158) from the quantiser multipliers BEFORE their qp/6 shift:
// quant_mf6 [n_cat][6][n] (= DIV(def_quant * 16, scaling_list)) -> out [n_cat][52][n]
extern "C" void x264hip_unquant_table(const int32_t *quant_mf6, int n_cat, int n, int32_t *out)
{
    for (int c = 0; c < n_cat; c++)
        for (int q = 0; q < 52; q++)
            for (int i = 0; i < n; i++)
                out[((size_t)c * 52 + q) * n + i] = (int32_t)((1ULL << (q / 6 + (n == 64 ? 16 : 15) + 8)) / (uint64_t)quant_mf6[((size_t)c * 6 + q % 6) * n + i]);
}''')])
s=open('/root/repo/x264_vs2008_amd/csrc/runtime.hip').read()
if '#include <math.h>' not in s and '#include <cmath>' not in s:
    s=s.replace('#include "internal.h"','#include "internal.h"\n#include <math.h>',1)
    open('/root/repo/x264_vs2008_amd/csrc/runtime.hip','w').write(s)
patch('/root/repo/include/x264hip.h', [('''int x264hip_aq_var_frame(''','''/* Host tables with floating-point arithmetic in the reference, built in C with its expression (no GPU involved):
 * p_cost_mv for one lambda (R/encoder/analyse.c:182-198), out[2 * span + 1] centred at span; h->unquant4_mf / unquant8_mf
 * (R/common/set.c:146,158) for every QP from the unshifted multipliers quant_mf6 [n_cat][6][n] -> out [n_cat][52][n]. */
void x264hip_cost_mv_table(int lambda, int span, int16_t *out);
void x264hip_unquant_table(const int32_t *quant_mf6, int n_cat, int n, int32_t *out);
int x264hip_aq_var_frame(''')])
print("ok")
