/* x264hip_lookahead.h -- the lookahead's device-side helpers that came after include/x264hip.h's "Lookahead and rate control" section
 * (x264hip_lookahead_* host state machine, x264hip_lookahead_cost_frames): storage of a lookahead slot and the per-chain indirections
 * the main encode needs once chains stop moving in lock step (adaptive B placement, CRF).  C ABI, plain pointers and sizes. */
#ifndef X264HIP_LOOKAHEAD_H
#define X264HIP_LOOKAHEAD_H
#include "x264hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A lookahead slot's picture: Y, U, V and the four half-resolution planes, no half-pel planes (an input frame waiting in frames.next:
 * x264_frame_new with b_have_lowres, R/common/frame.c:80-96).  Freed with x264hip_picture_free. */
int x264hip_picture_alloc_lookahead(x264hip_frame_ctx *c, x264hip_picture *pic);

#ifdef __cplusplus
}
#endif
#endif
