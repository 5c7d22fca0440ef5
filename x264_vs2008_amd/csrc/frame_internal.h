// frame_internal.h -- state of the frame level (planes resident in HBM)
#pragma once
#include "internal.h"

#define PADH 32   // R/common/frame.h:27-29
#define PADV 32

struct x264hip_frame_ctx {
    x264hip_frame_dims d;
    hipStream_t stream;
    bool own_stream;
    int width16, lines16;          // coded luma size
    unsigned long long *ssd_dev;   // [3] accumulators for x264hip_ssd_frame
    int *diag_dev;                 // scratch
};

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
