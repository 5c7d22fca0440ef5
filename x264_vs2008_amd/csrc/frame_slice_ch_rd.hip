// frame_slice_ch_rd.hip -- the chain-table launch (slice_kernel.h, template argument CH) of the raster sweep's I / P kernel: every
// block takes its arguments from its own table entry, so the chains of one launch may code frames of different QPs, references and
// pictures (x264hip_slice_sweep_chains).  A kernel of its own: the lock-step launches keep their argument passing untouched.
#include "slice_kernel.h"

void x264hip_launch_slice_rd_ch(const SwDesc *tab, int n, hipStream_t stream)
{
    SwArgs a; SwRefs t; SwRd r;
    memset(&a, 0, sizeof(a)); memset(&t, 0, sizeof(t)); memset(&r, 0, sizeof(r));
    hipLaunchKernelGGL((k_slice_sweep<2, false, true, false, false, false, true>), dim3((unsigned)n), dim3(64), 0, stream, a, t, r, tab);
}
