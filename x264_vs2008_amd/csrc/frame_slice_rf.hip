// frame_slice_rf.hip -- the I / P instantiation of the raster sweep with the RD refinement of subme 8-9 (slice_kernel.h, template
// argument RF; slice_refine.h): x264_me_refine_qpel_rd on every partition of the winning type, x264_intra_rd_refine on an intra
// winner, on top of everything the subme 6-7 kernel does.  A kernel of its own: the refinement's code and state cost the common
// (subme <= 7) kernel nothing.
#include "slice_kernel.h"

void x264hip_launch_slice_rf(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream)
{
    hipLaunchKernelGGL((k_slice_sweep<2, false, true, false, false, true>), dim3((unsigned)a.batch), dim3(64), 0, stream, a, t, r, nullptr);
}
