// frame_slice_b.hip -- the B-slice instantiation of the raster sweep (slice_kernel.h with slice_b_flow.h): one wavefront per chain,
// list 0 and list 1, direct prediction, bi-prediction, the CABAC B syntax.  A separate kernel so that the I / P one keeps its
// registers and LDS.
#include "slice_kernel.h"

void x264hip_launch_slice_b(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream)
{
    hipLaunchKernelGGL((k_slice_sweep<2, false, true, true>), dim3((unsigned)a.batch), dim3(64), 0, stream, a, t, r, nullptr);
}
