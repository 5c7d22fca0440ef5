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


/* The macroblock sweep for chains that no longer move in lock step.  x264hip_slice_sweep_frame codes the same kind of frame in every
 * batch element; once x264_slicetype_decide places B frames per chain and the rate control prices every frame on its own, chain A
 * codes a P frame from references {9, 8} at QP 24 while chain B codes a B frame between 8 and 12 at QP 27.  Here every entry is ONE
 * chain's sweep, described exactly like a call of x264hip_slice_sweep_frame -- its own source picture, references, reconstruction,
 * states and slice parameters (QP, POCs, lowres vectors, list 1) -- and the chain (batch element) it applies to: the pictures and states
 * are the batch-wide ones, the entry touches element `chain` of each.  One launch per kernel kind (I / P, I / P with the subme 8-9
 * refinement, B), each block taking its arguments from its entry (csrc/slice_kernel.h, template argument CH).  The raster variant with
 * the entropy coder in the loop only (params->rd with write = 1).  Every entry's `out` state gets the frame-level scalars later frames
 * read (poc, ref_poc ...): give each chain its own COPY of the x264hip_mb_state structure (same device arrays, its own scalars).  The
 * abort flag of every distinct state written must be cleared before the call (x264hip_mb_state_clear_progress).  staging_host (pinned)
 * and table_dev: n * x264hip_chain_sweep_bytes() each, left alone until the stream has passed the call. */
typedef struct {
    int chain;
    const x264hip_picture *fenc;
    const x264hip_picture *const *refs;
    int n_refs;
    x264hip_picture *recon;
    const x264hip_slice_params *params;
    const x264hip_mb_state *l0;
    x264hip_mb_state *out;
} x264hip_chain_sweep;
int x264hip_slice_sweep_chains(x264hip_frame_ctx *c, x264hip_chain_sweep *entries, int n, void *staging_host, void *table_dev);
size_t x264hip_chain_sweep_bytes(void);
int x264hip_mb_state_clear_progress(x264hip_frame_ctx *c, x264hip_mb_state *st);
/* x264hip_slice_sweep_chains with the completion of its two kernel kinds visible to the caller: ev_ip (x264hip_event_create) is recorded
 * behind the I / P kernels on the context's stream, ev_b behind the B kernel on the library's second stream of this context, and the
 * context's stream does not wait for the B kernel.  A scheduler polls the events (x264hip_event_query) and hands each chain its next
 * frame as soon as its own kernel is done instead of when the step's slowest chain is (x264_vs2008_amd/stream.py: AsyncStreamEncoder). */
int x264hip_slice_sweep_chains_events(x264hip_frame_ctx *c, x264hip_chain_sweep *entries, int n, void *staging_host, void *table_dev, void *ev_ip, void *ev_b);
int x264hip_event_query(void *ev);       /* 1 finished, 0 not yet, < 0 error */
/* a stream of the device's greatest priority (x264hip_stream_destroy frees it): its kernels' wavefronts are dispatched first when a slot
 * frees -- the lookahead's short kernels beside sweeps that keep the device full */
void *x264hip_stream_create_high_priority(void);
/* a stream whose kernels run on compute units [first, first + n) of 256 only; x264hip_frame_ctx_set_b_stream: the stream the B kernel of
 * this context's chain-table launches runs on (default: one the library creates) -- a caller may give the step's I / P chains and its B chains
 * disjoint parts of the device (x264hip_frame_ctx_new takes the context's own stream) */
void *x264hip_stream_create_cu_range(int first_cu, int n_cus);
int x264hip_frame_ctx_set_b_stream(x264hip_frame_ctx *c, void *hip_stream);
/* The batch elements the end-of-frame calls of this context touch from now on -- x264hip_deblock_frame, x264hip_expand_border,
 * x264hip_hpel_filter_frame: a device list of n element indices, or NULL = all.  With chains out of lock step a pool picture holds,
 * per element, either the frame that chain has just coded into it (to be filtered and kept as a reference) or an older reference of
 * another chain that must stay as it is. */
int x264hip_frame_ctx_elements(x264hip_frame_ctx *c, const int *elems_dev, int n);


/* The CAVLC writer: x264_macroblock_write_cavlc + x264_slice_write's skip runs (R/encoder/cavlc.c:60-620, R/encoder/encoder.c:1200-1280) for
 * every chain's I or P slice, as a pass over the state x264hip_slice_sweep_frame left (a `--no-cabac` slice below the RD levels: the wavefront
 * variant at constant QP, or the raster variant without its writer when adaptive quantisation gives every macroblock its QP; decisions and
 * coefficient levels in the x264hip_mb_state -- allocate it WITH level arrays).  payload: device
 * [batch][payload_cap] bytes, chain b's slice_data() starts X264HIP_PAYLOAD_LEAD (64) bytes into its slot, from bit 0, rbsp trailing bits
 * included; payload_len: device [batch] int32; mb_bits (optional): device [batch][n_mb], bit position after every macroblock.
 * Asynchronous on the context's stream.  I_PCM macroblocks and a slot too small end with x264hip_slice_sweep_status reporting an abort. */
typedef struct {
    int slice_type;            /* 0 P, 2 I */
    int n_ref0;                /* h->mb.pic.i_fref[0]: references of list 0 (the number te() is written against) */
    int analyse_inter;         /* param.analyse.inter (X264_ANALYSE_PSUB8x8 decides how sub-partition types are written) */
    int transform8x8;          /* pps->b_transform_8x8_mode */
    int cqm_custom;            /* param.i_cqm_preset != FLAT (High profile: level escapes beyond prefix 15) */
    uint8_t *payload; int payload_cap; int32_t *payload_len; int32_t *mb_bits;
    int slice_qp;              /* h->sh.i_qp: what mb_qp_delta of the first coded macroblock is relative to (per-macroblock QPs: the state's qp array) */
} x264hip_cavlc_params;
int x264hip_cavlc_write_frame(x264hip_frame_ctx *c, const x264hip_mb_state *st, const x264hip_cavlc_params *p);

#ifdef __cplusplus
}
#endif
#endif
