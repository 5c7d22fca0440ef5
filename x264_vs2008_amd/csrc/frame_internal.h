// frame_internal.h -- state of the frame level (planes resident in HBM)
#pragma once
#include "internal.h"

#define PADH 32   // R/common/frame.h:27-29
#define PADV 32

// A context describes `batch` independent frames of one size (one per GOP
// chain) that every frame-level call processes in the same launches
// (blockIdx.z = batch element).  Each plane of a picture is `batch`
// consecutive padded images `bs_*` bytes apart; per-macroblock arrays are
// `batch` consecutive [n_mb][...] blocks.
struct x264hip_frame_ctx {
    x264hip_frame_dims d;
    hipStream_t stream;
    bool own_stream;
    int width16, lines16;          // coded luma size
    int batch, sel;                // batch size; element addressed by upload / download
    size_t bs_y, bs_c, bs_l;       // bytes between batch elements: luma-sized, chroma-sized, lowres planes
    int stride_l, width_l, lines_l;
    unsigned long long *ssd_dev;   // [batch][3] accumulators for x264hip_ssd_frame
    // x264hip_frame_ctx_elements: the batch elements the end-of-frame calls (deblock, border expansion, half-pel filter) touch;
    // NULL = all.  Chains that code different kinds of frames keep some of a picture's elements as they are (device list).
    const int *elems; int n_elems;
};

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }
