// frame_slice_rd.hip -- the raster-order variant of the macroblock sweep (slice_kernel.h, template argument RD): one wavefront per
// chain walks the whole frame, with the RD levels, trellis, adaptive quantisation and the CABAC coder inside the loop.
#include <cstdlib>
#include "slice_kernel.h"

void x264hip_launch_slice_rd(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream)
{
    const size_t lds_bytes = sw_lds_bytes<true, false>();
    static int wpe = 0;
    if (!wpe) { const char *e = getenv("X264HIP_RASTER_WPE"); wpe = e && atoi(e) == 2 ? 2 : 3; }     // developer knob: registers per chain (2: up to 256, 3: 168)
    if (wpe == 2) hipLaunchKernelGGL((k_slice_sweep<2, false, true>), dim3((unsigned)a.batch), dim3(64), lds_bytes, stream, a, t, r);
    else hipLaunchKernelGGL((k_slice_sweep<3, false, true>), dim3((unsigned)a.batch), dim3(64), lds_bytes, stream, a, t, r);
}
int x264hip_occupancy_slice_rd(void)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_slice_sweep<3, false, true>, 64, sw_lds_bytes<true, false>()) != hipSuccess) return -1;
    return n;
}
