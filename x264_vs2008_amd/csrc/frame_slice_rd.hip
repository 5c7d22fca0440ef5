// frame_slice_rd.hip -- the raster-order variant of the macroblock sweep (slice_kernel.h, template argument RD): one wavefront per
// chain walks the whole frame, with the RD levels, trellis, adaptive quantisation and the CABAC coder inside the loop.
#include "slice_kernel.h"

void x264hip_launch_slice_rd(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream)
{
    hipLaunchKernelGGL((k_slice_sweep<2, false, true>), dim3((unsigned)a.batch), dim3(64), 0, stream, a, t, r, nullptr);
}
