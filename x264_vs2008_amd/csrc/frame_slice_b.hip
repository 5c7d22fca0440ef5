// frame_slice_b.hip -- the B-slice instantiation of the raster sweep (slice_kernel.h with slice_b_flow.h): one wavefront per chain,
// list 0 and list 1, direct prediction, bi-prediction, the CABAC B syntax.  A separate kernel so that the I / P one keeps its
// registers and LDS.
#include <cstdlib>
#include "slice_kernel.h"

void x264hip_launch_slice_b(const SwArgs &a, const SwRefs &t, const SwRd &r, hipStream_t stream)
{
    const size_t lds_bytes = sw_lds_bytes<true, true>();
    static int wpe = 0;
    if (!wpe) { const char *e = getenv("X264HIP_RASTER_WPE"); wpe = e && atoi(e) == 2 ? 2 : 3; }     // developer knob: registers per chain (2: up to 256, 3: 168)
    if (wpe == 2) hipLaunchKernelGGL((k_slice_sweep<2, false, true, true>), dim3((unsigned)a.batch), dim3(64), lds_bytes, stream, a, t, r);
    else hipLaunchKernelGGL((k_slice_sweep<3, false, true, true>), dim3((unsigned)a.batch), dim3(64), lds_bytes, stream, a, t, r);
}
int x264hip_occupancy_slice_b(void)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_slice_sweep<3, false, true, true>, 64, sw_lds_bytes<true, true>()) != hipSuccess) return -1;
    return n;
}
