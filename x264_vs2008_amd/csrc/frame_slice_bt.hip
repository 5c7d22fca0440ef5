// frame_slice_bt.hip -- the B-slice instantiation of the raster sweep with TEMPORAL direct prediction (slice_b_flow.h under
// `if constexpr (TD)`: x264_mb_predict_mv_direct16x16's temporal branch, its failure path, and the motion-cache entry that survives
// from macroblock to macroblock, x264hip_slice_rd.stale).  A kernel of its own so that the spatial one -- the medium preset's, the
// bench's -- keeps its registers: with the branch inside it the B launches ran 8 % slower.
#include "slice_kernel.h"

void x264hip_launch_slice_bt(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream)
{
    hipLaunchKernelGGL((k_slice_sweep<2, false, true, true, true>), dim3((unsigned)a.batch), dim3(64), 0, stream, a, t, r);
}
