// frame_slice_ch_bt.hip -- the chain-table launch (slice_kernel.h, template argument CH) of the extended B kernel (temporal direct
// prediction, the lookahead's candidates): see frame_slice_ch_rd.hip.
#include "slice_kernel.h"

void x264hip_launch_slice_bt_ch(const SwDesc *tab, int n, hipStream_t stream)
{
    SwArgs a; SwRefs t; SwRd r;
    memset(&a, 0, sizeof(a)); memset(&t, 0, sizeof(t)); memset(&r, 0, sizeof(r));
    hipLaunchKernelGGL((k_slice_sweep<2, false, true, true, true, false, true>), dim3((unsigned)n), dim3(64), 0, stream, a, t, r, tab);
}
