// frame_slice_bt.hip -- the EXTENDED B-slice instantiation of the raster sweep (slice_b_flow.h under `TD`): temporal direct prediction
// (x264_mb_predict_mv_direct16x16's temporal branch, its failure path, and the motion-cache entry that survives from macroblock to
// macroblock, x264hip_slice_rd.stale) and the lookahead's candidates of the 16x16 searches (x264hip_slice_params.lowres_mv /
// x264hip_slice_b.lowres_mv1).  A kernel of its own so that the plain one -- spatial direct prediction, no lookahead: the bench's --
// keeps its registers: with either addition inside it the B launches ran 6-8 % slower.
#include "slice_kernel.h"

void x264hip_launch_slice_bt(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream)
{
    hipLaunchKernelGGL((k_slice_sweep<2, false, true, true, true>), dim3((unsigned)a.batch), dim3(64), 0, stream, a, t, r, nullptr);
}
