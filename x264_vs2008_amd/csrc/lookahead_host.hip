// lookahead_host.hip -- HOST ONLY: the frame queue, slice-type decision and rate control of one GOP chain (include/x264hip.h,
// "Lookahead and rate control of ONE GOP chain").  Nothing here touches the device; the file is a .hip only so that it is built with
// the rest of the library (-ffp-contract=off: the rate control's float / double expressions must round like the reference's C).
//
//   x264_encoder_encode's queue            R/encoder/encoder.c:1390-1470, x264_reference_update :1058-1094, x264_reference_build_list :911-981
//   x264_slicetype_decide / _analyse       R/encoder/slicetype.c:476-636, scenecut :437-474, the trellis paths :359-435
//   x264_rc_analyse_slice                  R/encoder/slicetype.c:638-680
//   x264_ratecontrol_new / _start / _end   R/encoder/ratecontrol.c:268-420, 792-870, 1077-1160; rate_estimate_qscale :1396-1615
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "x264hip.h"

namespace {

enum { T_AUTO = 0, T_IDR = 1, T_I = 2, T_P = 3, T_BREF = 4, T_B = 5 };      // R/x264.h:116-121
enum { ST_P = 0, ST_B = 1, ST_I = 2 };                                     // R/common/common.h:107-113 (slice_type_e)
#define IS_TYPE_I(t) ((t) == T_I || (t) == T_IDR)
#define IS_TYPE_B(t) ((t) == T_B || (t) == T_BREF)
#define BF_MAX 16                                                          // X264_BFRAME_MAX
#define MAX_LENGTH (BF_MAX * 4)
#define RING 1024
#define COST_MAX (1 << 28)

struct LF {                                   // the fields of x264_frame_t this path reads (R/common/frame.h:28-100)
    int frame, type, poc, kept_as_ref;
    int cost_est[BF_MAX + 2][BF_MAX + 2];
    int intra_mbs[BF_MAX + 2];
    bool searched[2][BF_MAX + 1];             // !(lowres_mvs[l][d][0][0] == 0x7FFF)
    // results of speculative tasks: the reference has not asked for them (yet), so they must not show -- a searched list is offered to the
    // main encode's 16x16 search.  They move to the fields above when the decision does ask.
    int spec_cost[BF_MAX + 2][BF_MAX + 2], spec_intra_mbs[BF_MAX + 2], spec_cost00;
    bool dev_searched[2][BF_MAX + 1];         // the caller's arrays hold the vectors (whether or not the reference would have them)
    float f_qp_avg_rc;
    int i_satd;
};

struct RC {                                   // x264_ratecontrol_t, the types of R/encoder/ratecontrol.c:63-131
    int b_abr;
    double fps, bitrate;
    int nmb;
    int qp_constant[5];
    int qp;
    float f_qpm, qpa_rc;
    int last_satd;
    double last_rceq, cplxr_sum, wanted_bits_window, cbr_decay, short_term_cplxsum, short_term_cplxcount, rate_factor_constant, ip_offset, pb_offset;
    double last_qscale, last_qscale_for[5];
    int last_non_b_pict_type;
    double accum_p_qp, accum_p_norm, last_accum_p_norm, lmin[5], lmax[5], lstep;
    int bframes;
};

}  // namespace

struct x264hip_lookahead {
    x264hip_lookahead_params p;
    LF ring[RING];
    LF *next[MAX_LENGTH + 8], *current[MAX_LENGTH + 8], *reference[16 + 2];
    LF *last_nonb, *fenc, *fref0[16 + 2], *fref1[16 + 2];
    int n_ref0, n_ref1;
    int i_input, i_last_idr, i_frame, i_delay, i_max_dpb, i_max_ref1;
    int slice_type;
    bool started, miss;
    bool setup;                                // la->fenc's type-dependent state (last IDR, reference lists, POC) is in place
    int frame_num_reset;                       // the frame handed out is a scene-cut IDR: x264_encoder_encode restarts h->i_frame_num for it (encoder.c:1682)
    x264hip_look_need needs[16];
    int n_needs;
    RC rc;
};

namespace {

typedef x264hip_lookahead LA;

inline double qp2qscale(double qp) { return 0.85 * pow(2.0, (qp - 12.0) / 6.0); }              // ratecontrol.c:146-153
inline double qscale2qp(double qscale) { return 12.0 + 6.0 * log(qscale / 0.85) / log(2.0); }
inline double clip3f(double v, double lo, double hi) { return v < lo ? lo : v > hi ? hi : v; }
inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

int list_len(LF **l) { int n = 0; while (l[n]) n++; return n; }
void list_push(LF **l, LF *f) { int n = list_len(l); l[n] = f; l[n + 1] = nullptr; }           // x264_frame_push
LF *list_shift(LF **l) { LF *f = l[0]; int i; for (i = 0; l[i]; i++) l[i] = l[i + 1]; return f; }   // x264_frame_shift
LF *list_pop(LF **l) { int n = list_len(l); LF *f = l[n - 1]; l[n - 1] = nullptr; return f; }

int num_mbs(const LA *la)
{   // NUM_MBS, slicetype.c:251-254
    return la->p.mb_w > 2 && la->p.mb_h > 2 ? (la->p.mb_w - 2) * (la->p.mb_h - 2) : la->p.mb_w * la->p.mb_h;
}

void add_need(LA *la, LF **frames, int p0, int p1, int b, int speculative)
{
    const int fb = frames[b]->frame, f0 = fb - (b - p0), f1 = fb + (p1 - b);
    for (int i = 0; i < la->n_needs; i++)
        if (la->needs[i].b == fb && la->needs[i].p0 == f0 && la->needs[i].p1 == f1) return;
    if (la->n_needs == 16) return;
    x264hip_look_need *n = &la->needs[la->n_needs];
    n->b = fb; n->p0 = f0; n->p1 = f1;
    n->do_search[0] = b != p0 && !frames[b]->dev_searched[0][b - p0 - 1];                        // slicetype.c:281-284
    n->do_search[1] = b != p1 && !frames[b]->dev_searched[1][p1 - b - 1];
    n->speculative = speculative;
    if (speculative)                                                                             // two tasks of one batch never search the same vectors
        for (int i = 0; i < la->n_needs; i++)
            if (la->needs[i].b == fb && ((n->do_search[0] && la->needs[i].do_search[0] && la->needs[i].p0 == f0) ||
                                         (n->do_search[1] && la->needs[i].do_search[1] && la->needs[i].p1 == f1))) return;
    la->n_needs++;
}

// x264_slicetype_frame_cost's memoised face (slicetype.c:256-345): the cached score or a request for it
int frame_cost(LA *la, LF **frames, int p0, int p1, int b, int b_intra_penalty)
{
    if (la->miss) return 0;
    int score = frames[b]->cost_est[b - p0][p1 - b];
    if (score < 0 && frames[b]->spec_cost[b - p0][p1 - b] >= 0) {                                // computed ahead of the question: it shows from now on
        LF *f = frames[b];
        score = f->cost_est[b - p0][p1 - b] = f->spec_cost[b - p0][p1 - b];
        if (b == p1) { f->intra_mbs[b - p0] = f->spec_intra_mbs[b - p0]; f->cost_est[0][0] = f->spec_cost00; }
        if (b != p0) f->searched[0][b - p0 - 1] = true;
        if (b != p1) f->searched[1][p1 - b - 1] = true;
    }
    if (score < 0) { add_need(la, frames, p0, p1, b, 0); la->miss = true; return 0; }
    if (b_intra_penalty) {
        const int nmb = num_mbs(la);
        score += score * frames[b]->intra_mbs[b - p0] / (nmb * 8);
    }
    return score;
}

int path_cost(LA *la, LF **frames, char *path, int threshold)
{   // x264_slicetype_path_cost, slicetype.c:359-391
    int loc = 1, cost = 0, cur_p = 0;
    path--;
    while (path[loc]) {
        int next_p = loc, next_b;
        while (path[next_p] && path[next_p] != 'P') next_p++;
        if (path[next_p] != 'P') return cost;
        cost += frame_cost(la, frames, cur_p, next_p, next_p, 0);
        if (la->miss) return cost;
        if (cost > threshold) break;
        for (next_b = loc; next_b < next_p && cost < threshold; next_b++) {
            cost += frame_cost(la, frames, cur_p, next_p, next_b, 0);
            if (la->miss) return cost;
        }
        loc = next_p + 1;
        cur_p = next_p;
    }
    return cost;
}

void slicetype_path(LA *la, LF **frames, int length, int max_bframes, char (*best_paths)[MAX_LENGTH])
{   // x264_slicetype_path, slicetype.c:395-426
    char paths[BF_MAX + 2][MAX_LENGTH];
    memset(paths, 0, sizeof(paths));
    int num_paths = max_bframes + 1 < length ? max_bframes + 1 : length;
    int best_cost = COST_MAX, best_path_index = 0;
    length = length < MAX_LENGTH ? length : MAX_LENGTH;
    for (int suffix_size = 0; suffix_size < num_paths; suffix_size++) {
        memcpy(paths[suffix_size], best_paths[length - (suffix_size + 1)], length - (suffix_size + 1));
        for (int loc = 0; loc < suffix_size; loc++) strcat(paths[suffix_size], "B");
        strcat(paths[suffix_size], "P");
    }
    for (int path = 0; path < num_paths; path++) {
        int cost = path_cost(la, frames, paths[path], best_cost);
        if (la->miss) return;
        if (cost < best_cost) { best_cost = cost; best_path_index = path; }
    }
    memcpy(best_paths[length], paths[best_path_index], length);
}

int path_search(LA *la, LF **frames, int length, int bframes)
{   // x264_slicetype_path_search, slicetype.c:428-435
    static thread_local char best_paths[MAX_LENGTH][MAX_LENGTH];
    memset(best_paths, 0, sizeof(best_paths));
    best_paths[1][0] = 'P';
    for (int n = 2; n < length - 1; n++) {
        slicetype_path(la, frames, n, bframes, best_paths);
        if (la->miss) return 0;
    }
    return (int)strspn(best_paths[length - 2], "B");
}

int scenecut(const LA *la, const LF *frame, int pdist)
{   // slicetype.c:437-474
    int icost = frame->cost_est[0][0];
    int pcost = frame->cost_est[pdist][0];
    float f_bias;
    int i_gop_size = frame->frame - la->i_last_idr;
    float f_thresh_max = la->p.scenecut_threshold / 100.0;
    float f_thresh_min = f_thresh_max * la->p.keyint_min / (la->p.keyint_max * 4);
    if (la->p.keyint_min == la->p.keyint_max) f_thresh_min = f_thresh_max;
    if (i_gop_size < la->p.keyint_min / 4)
        f_bias = f_thresh_min / 4;
    else if (i_gop_size <= la->p.keyint_min)
        f_bias = f_thresh_min * i_gop_size / la->p.keyint_min;
    else
        f_bias = f_thresh_min + (f_thresh_max - f_thresh_min) * (i_gop_size - la->p.keyint_min) / (la->p.keyint_max - la->p.keyint_min);
    return pcost >= (1.0 - f_bias) * icost;
}

void slicetype_analyse(LA *la)
{   // x264_slicetype_analyse, slicetype.c:476-575
    LF *frames[BF_MAX * 4 + 3] = {nullptr};
    int num_frames, keyint_limit, j;
    const int i_mb_count = num_mbs(la);
    int cost1p0, cost2p0, cost1b1, cost2p1, idr_frame_type;

    if (!la->last_nonb) return;
    frames[0] = la->last_nonb;
    for (j = 0; la->next[j] && la->next[j]->type == T_AUTO; j++) frames[j + 1] = la->next[j];
    keyint_limit = la->p.keyint_max - frames[0]->frame + la->i_last_idr - 1;
    num_frames = j < keyint_limit ? j : keyint_limit;
    if (num_frames == 0) return;

    idr_frame_type = frames[1]->frame - la->i_last_idr >= la->p.keyint_min ? T_IDR : T_I;

    if (num_frames == 1) {
no_b_frames:
        if (la->miss) return;
        frames[1]->type = T_P;
        if (la->p.pre_scenecut) {
            frame_cost(la, frames, 0, 1, 1, 0);
            if (la->miss) return;
            if (scenecut(la, frames[1], 1)) frames[1]->type = idr_frame_type;
        }
        return;
    }

    if (la->p.b_adapt == 2) {
        int num_bframes;
        int max_bframes = num_frames - 1 < la->p.bframes ? num_frames - 1 : la->p.bframes;
        if (la->p.pre_scenecut) {
            frame_cost(la, frames, 0, 1, 1, 0);
            if (la->miss) return;
            if (scenecut(la, frames[1], 1)) { frames[1]->type = idr_frame_type; return; }
        }
        num_bframes = path_search(la, frames, num_frames, max_bframes);
        if (la->miss) return;
        for (j = 1; j < num_bframes + 1; j++) {
            if (la->p.pre_scenecut) {
                // scenecut() reads i_cost_est[j + 1][0] of frames[j + 1] whether or not the paths priced it: -1 then, as in the reference
                if (scenecut(la, frames[j + 1], j + 1)) { frames[j]->type = T_P; frames[j + 1]->type = idr_frame_type; return; }
            }
            frames[j]->type = T_B;
        }
        frames[num_bframes + 1]->type = T_P;
    } else {
        // what this decision can still ask for, P-type tasks only: independent of each other, so a caller may compute them in one batch
        const int jmax = la->p.bframes < num_frames - 1 ? la->p.bframes : num_frames - 1;
        cost2p1 = frame_cost(la, frames, 0, 2, 2, 1);
        if (la->miss) goto speculate;
        if (frames[2]->intra_mbs[2] > i_mb_count / 2) goto no_b_frames;

        cost1b1 = frame_cost(la, frames, 0, 2, 1, 0);
        cost1p0 = frame_cost(la, frames, 0, 1, 1, 0);
        cost2p0 = frame_cost(la, frames, 1, 2, 2, 0);
        if (la->miss) goto speculate;

        if (cost1p0 + cost2p0 < cost1b1 + cost2p1) goto no_b_frames;

        frames[1]->type = T_B;
        for (j = 2; j <= jmax; j++) {
            int pthresh = 300 - (50 - la->p.bframe_bias) * (j - 1);                               // INTER_THRESH, P_SENS_BIAS
            if (pthresh < 300 / 10) pthresh = 300 / 10;
            int pcost = frame_cost(la, frames, 0, j + 1, j + 1, 1);
            if (la->miss) goto speculate;
            if (pcost > pthresh * i_mb_count || frames[j + 1]->intra_mbs[j + 1] > i_mb_count / 3) { frames[j]->type = T_P; break; }
            else frames[j]->type = T_B;
        }
        return;
speculate:
#define UNKNOWN(f_, d_) ((f_)->cost_est[d_][0] < 0 && (f_)->spec_cost[d_][0] < 0)
        if (UNKNOWN(frames[2], 2)) add_need(la, frames, 0, 2, 2, 1);
        if (UNKNOWN(frames[1], 1)) add_need(la, frames, 0, 1, 1, 1);
        if (UNKNOWN(frames[2], 1)) add_need(la, frames, 1, 2, 2, 1);
        for (j = 2; j <= jmax; j++)
            if (UNKNOWN(frames[j + 1], j + 1)) add_need(la, frames, 0, j + 1, j + 1, 1);
#undef UNKNOWN
    }
}

void slicetype_decide(LA *la)
{   // x264_slicetype_decide, slicetype.c:577-636
    if (!la->next[0]) return;
    if ((la->p.bframes && la->p.b_adapt) || la->p.pre_scenecut) slicetype_analyse(la);
    if (la->miss) return;
    for (int bframes = 0;; bframes++) {
        LF *frm = la->next[bframes];
        if (frm->frame - la->i_last_idr >= la->p.keyint_max) {
            if (frm->type == T_AUTO) frm->type = T_IDR;
        }
        if (frm->type == T_IDR) {
            if (bframes > 0) { bframes--; la->next[bframes]->type = T_P; }
        }
        if (bframes == la->p.bframes || la->next[bframes + 1] == nullptr) {
            if (frm->type == T_AUTO || IS_TYPE_B(frm->type)) frm->type = T_P;
        }
        if (frm->type == T_AUTO) frm->type = T_B;
        else if (!IS_TYPE_B(frm->type)) break;
    }
}

// ---- rate control ----------------------------------------------------------------------------------------------------------------
void rc_new(LA *la)
{   // x264_ratecontrol_new, ratecontrol.c:268-377
    RC *rc = &la->rc;
    const x264hip_lookahead_params *p = &la->p;
    memset(rc, 0, sizeof(*rc));
    rc->b_abr = p->rc_method != 0;
    rc->fps = 25.0;
    rc->bitrate = 0 * 1000.;
    rc->nmb = p->mb_w * p->mb_h;
    rc->last_non_b_pict_type = -1;
    rc->cbr_decay = 1.0;
#define ABR_INIT_QP (p->rc_method == 1 ? p->rf_constant : 24)
    if (rc->b_abr) {
        rc->accum_p_norm = .01;
        rc->accum_p_qp = ABR_INIT_QP * rc->accum_p_norm;
        rc->cplxr_sum = .01 * pow(7.0e5, (double)p->qcompress) * pow((double)rc->nmb, 0.5);
        rc->wanted_bits_window = 1.0 * rc->bitrate / rc->fps;
        rc->last_non_b_pict_type = ST_I;
    }
    if (p->rc_method == 1) {
        double base_cplx = rc->nmb * (p->bframes ? 120 : 80);
        rc->rate_factor_constant = pow(base_cplx, (double)(1 - p->qcompress)) / qp2qscale(p->rf_constant);
    }
    rc->ip_offset = 6.0 * log((double)p->ip_factor) / log(2.0);
    rc->pb_offset = 6.0 * log((double)p->pb_factor) / log(2.0);
    rc->qp_constant[ST_P] = p->qp_constant;
    rc->qp_constant[ST_I] = clip3((int)(p->qp_constant - rc->ip_offset + 0.5), 0, 51);
    rc->qp_constant[ST_B] = clip3((int)(p->qp_constant + rc->pb_offset + 0.5), 0, 51);
    rc->lstep = pow(2.0, p->qp_step / 6.0);
    rc->last_qscale = qp2qscale(26);
    for (int i = 0; i < 5; i++) {
        rc->last_qscale_for[i] = qp2qscale(ABR_INIT_QP);
        rc->lmin[i] = qp2qscale(p->qp_min);
        rc->lmax[i] = qp2qscale(p->qp_max);
    }
}

double get_qscale(LA *la, int tex_bits, int mv_bits, float blurred_complexity, double rate_factor)
{   // ratecontrol.c:1168-1195 (no zones)
    RC *rcc = &la->rc;
    double q = pow((double)blurred_complexity, (double)(1 - la->p.qcompress));     // C: pow(double(float), double(float))
    if (!isfinite(q) || tex_bits + mv_bits == 0)
        q = rcc->last_qscale;
    else {
        rcc->last_rceq = q;
        q /= rate_factor;
        rcc->last_qscale = q;
    }
    return q;
}

float rate_estimate_qscale(LA *la, int satd)
{   // ratecontrol.c:1396-1615, the B branch and the 1-pass branch with CRF; no VBV: clip_qscale is the clip to lmin / lmax
    float q;
    RC *rcc = &la->rc;
    const int pict_type = la->slice_type;
    const x264hip_lookahead_params *p = &la->p;
    if (pict_type == ST_B) {
        const LF *r0 = la->fref0[0], *r1 = la->fref1[0];
        int i0 = IS_TYPE_I(r0->type), i1 = IS_TYPE_I(r1->type);
        int dt0 = abs(la->fenc->poc - r0->poc), dt1 = abs(la->fenc->poc - r1->poc);
        float q0 = r0->f_qp_avg_rc, q1 = r1->f_qp_avg_rc;
        if (i0 && i1) q = (q0 + q1) / 2 + rcc->ip_offset;
        else if (i0) q = q1;
        else if (i1) q = q0;
        else q = (q0 * dt1 + q1 * dt0) / (dt0 + dt1);
        if (la->fenc->kept_as_ref) q += rcc->pb_offset / 2;
        else q += rcc->pb_offset;
        rcc->last_satd = 0;
        return qp2qscale(q);
    } else {
        rcc->last_satd = satd;
        rcc->short_term_cplxsum *= 0.5;
        rcc->short_term_cplxcount *= 0.5;
        rcc->short_term_cplxsum += rcc->last_satd;
        rcc->short_term_cplxcount++;
        const int tex_bits = rcc->last_satd;
        const float blurred_complexity = rcc->short_term_cplxsum / rcc->short_term_cplxcount;
        q = get_qscale(la, tex_bits, 0, blurred_complexity, rcc->rate_factor_constant);
        if (pict_type == ST_I && p->keyint_max > 1 && rcc->last_non_b_pict_type != ST_I) {
            q = qp2qscale(rcc->accum_p_qp / rcc->accum_p_norm);
            q /= fabs((double)p->ip_factor);
        } else if (la->i_frame > 0) {
            double lmin = rcc->last_qscale_for[pict_type] / rcc->lstep;
            double lmax = rcc->last_qscale_for[pict_type] * rcc->lstep;
            q = clip3f(q, lmin, lmax);
        } else {
            q = qp2qscale(ABR_INIT_QP) / fabs((double)p->ip_factor);
        }
        {
            double lmin = rcc->lmin[pict_type], lmax = rcc->lmax[pict_type];
            double qq = q;                                                                       // clip_qscale(h, pict_type, q): q is widened to double
            if (lmin == lmax) qq = lmin; else qq = clip3f(qq, lmin, lmax);
            q = qq;
        }
        rcc->last_qscale_for[pict_type] = rcc->last_qscale = q;
        if (la->fenc->frame == 0) rcc->last_qscale_for[ST_P] = q;
        return q;
    }
}

void rc_start(LA *la, int satd)
{   // x264_ratecontrol_start, ratecontrol.c:792-870
    RC *rc = &la->rc;
    float q;
    if (la->slice_type != ST_B) {
        rc->bframes = 0;
        while (la->current[rc->bframes] && IS_TYPE_B(la->current[rc->bframes]->type)) rc->bframes++;
    }
    if (rc->b_abr)
        q = qscale2qp(rate_estimate_qscale(la, satd));
    else {
        if (la->slice_type == ST_B && la->fenc->kept_as_ref) q = (rc->qp_constant[ST_B] + rc->qp_constant[ST_P]) / 2;
        else q = rc->qp_constant[la->slice_type];
    }
    rc->qpa_rc = 0;
    rc->qp = clip3((int)(q + 0.5), 0, 51);
    la->fenc->f_qp_avg_rc = rc->qp;
    rc->f_qpm = q;
    if (la->slice_type != ST_B) rc->last_non_b_pict_type = la->slice_type;
}

void rc_end(LA *la)
{   // x264_ratecontrol_mb's running sum and x264_ratecontrol_end, ratecontrol.c:930, 1077-1135.  What depends on the frame's size in bits
    // (cplxr_sum, the size predictors) feeds ABR and VBV only and is not kept.
    RC *rc = &la->rc;
    rc->qpa_rc = 0;
    for (int i = 0; i < rc->nmb; i++) rc->qpa_rc += rc->f_qpm;
    la->fenc->f_qp_avg_rc = rc->qpa_rc /= rc->nmb;
    if (rc->b_abr) {
        const float qp = rc->qpa_rc;                                                             // accum_p_qp_update( h, rc->qpa_rc ), :776-786
        rc->accum_p_qp *= .95;
        rc->accum_p_norm *= .95;
        rc->accum_p_norm += 1;
        if (la->slice_type == ST_I) rc->accum_p_qp += qp + rc->ip_offset;
        else rc->accum_p_qp += qp;
    }
}

}  // namespace

extern "C" x264hip_lookahead *x264hip_lookahead_new(const x264hip_lookahead_params *p)
{
    if (!p || p->bframes < 0 || p->bframes > BF_MAX || p->mb_w <= 0 || p->mb_h <= 0 || p->keyint_max < 1) return nullptr;
    // pre_scenecut = 0 with a threshold >= 0: the queue then decides without scene cuts, as the reference's does, and x264_encoder_encode looks at
    // every P frame after coding it (encoder.c:1603-1699).  That look is the caller's (x264hip_frame_stats + x264hip_scenecut_post); what follows
    // a hit is x264hip_lookahead_scenecut.
    LA *la = (LA *)calloc(1, sizeof(LA));
    if (!la) return nullptr;
    la->p = *p;
    la->i_delay = p->b_adapt == 2 ? (p->bframes > 3 ? p->bframes : 3) * 4 : p->bframes;            // encoder.c:703-706, one thread
    la->i_max_ref1 = p->bframes ? 1 : 0;                                                         // sps->vui.i_num_reorder_frames, set.c:176
    la->i_max_dpb = 16;                                                                          // only the nearest of each list is read here
    la->i_last_idr = -p->keyint_max;
    rc_new(la);
    return la;
}

extern "C" void x264hip_lookahead_delete(x264hip_lookahead *la) { free(la); }

extern "C" int x264hip_lookahead_put(x264hip_lookahead *la)
{
    LF *f = &la->ring[la->i_input % RING];
    memset(f, 0, sizeof(*f));
    f->frame = la->i_input++;
    f->type = T_AUTO;
    for (int i = 0; i < BF_MAX + 2; i++) for (int k = 0; k < BF_MAX + 2; k++) f->cost_est[i][k] = f->spec_cost[i][k] = -1;   // x264_frame_init_lowres, mc.c:323-330
    list_push(la->next, f);
    return f->frame;
}

extern "C" int x264hip_lookahead_get(x264hip_lookahead *la, int flushing, x264hip_look_frame *out, x264hip_look_need *need, int max_need, int *n_need)
{
    la->n_needs = 0; la->miss = false;
    if (n_need) *n_need = 0;
    if (la->started) return -1;
    if (!la->fenc) {
        if (!flushing && la->i_input <= la->i_delay) return X264HIP_LOOK_NONE;                   // encoder.c:1423-1430
        if (!la->current[0]) {
            int bframes = 0, types[MAX_LENGTH + 8], n = 0;
            if (!la->next[0]) return X264HIP_LOOK_END;
            for (n = 0; la->next[n]; n++) types[n] = la->next[n]->type;
            slicetype_decide(la);
            if (la->miss) {
                for (int i = 0; i < n; i++) la->next[i]->type = types[i];
                goto needs;
            }
            while (IS_TYPE_B(la->next[bframes]->type)) bframes++;                                // encoder.c:1444-1458
            list_push(la->current, list_shift(&la->next[bframes]));
            while (bframes--) list_push(la->current, list_shift(la->next));
        }
        la->fenc = list_shift(la->current);
        la->setup = false; la->frame_num_reset = 0;
    }
    if (!la->setup) {                                                                            // do_encode:, encoder.c:1471-1530
        LF *f = la->fenc;
        la->setup = true;
        if (f->type == T_IDR) {
            la->i_last_idr = f->frame;
            while (la->reference[0]) list_pop(la->reference);                                    // x264_reference_reset
        }
        la->slice_type = IS_TYPE_I(f->type) ? ST_I : f->type == T_P ? ST_P : ST_B;
        f->poc = 2 * (f->frame - la->i_last_idr);
        f->kept_as_ref = !IS_TYPE_B(f->type) && la->p.keyint_max > 1;
        la->n_ref0 = la->n_ref1 = 0;                                                             // x264_reference_build_list
        for (int i = 0; la->reference[i]; i++) {
            if (la->reference[i]->poc < f->poc) la->fref0[la->n_ref0++] = la->reference[i];
            else if (la->reference[i]->poc > f->poc) la->fref1[la->n_ref1++] = la->reference[i];
        }
        for (int i = 0; i < la->n_ref0; i++)
            for (int k = i + 1; k < la->n_ref0; k++)
                if (la->fref0[k]->poc > la->fref0[i]->poc) { LF *t = la->fref0[i]; la->fref0[i] = la->fref0[k]; la->fref0[k] = t; }
        for (int i = 0; i < la->n_ref1; i++)
            for (int k = i + 1; k < la->n_ref1; k++)
                if (la->fref1[k]->poc < la->fref1[i]->poc) { LF *t = la->fref1[i]; la->fref1[i] = la->fref1[k]; la->fref1[k] = t; }
        if (la->n_ref1 > la->i_max_ref1) la->n_ref1 = la->i_max_ref1;
    }
    {
        LF *f = la->fenc;
        int satd = 0;
        if (la->rc.b_abr && la->slice_type != ST_B) {                                            // x264_rc_analyse_slice, slicetype.c:638-680
            LF *frames[BF_MAX * 4 + 2] = {nullptr};
            int p0 = 0, p1, b;
            if (la->slice_type == ST_I) p1 = b = 0;
            else {
                p1 = 0;
                while (la->current[p1] && IS_TYPE_B(la->current[p1]->type)) p1++;
                p1++;
                b = p1;
            }
            frames[p0] = la->n_ref0 ? la->fref0[0] : nullptr;
            frames[b] = f;
            satd = frame_cost(la, frames, p0, p1, b, 0);
            if (la->miss) goto needs;
            f->i_satd = satd;
        }
        rc_start(la, satd);
        if (out) {
            out->frame = f->frame; out->type = f->type; out->poc = f->poc; out->kept_as_ref = f->kept_as_ref;
            out->qp = la->rc.qp; out->f_qpm = la->rc.f_qpm;
            const bool inter = la->slice_type != ST_I;
            out->ref0_frame = inter && la->n_ref0 ? la->fref0[0]->frame : -1;
            out->ref1_frame = la->slice_type == ST_B && la->n_ref1 ? la->fref1[0]->frame : -1;
            out->lowres_l0 = out->ref0_frame >= 0 && f->frame - out->ref0_frame - 1 <= BF_MAX && f->searched[0][f->frame - out->ref0_frame - 1];
            out->lowres_l1 = out->ref1_frame >= 0 && out->ref1_frame - f->frame - 1 <= BF_MAX && f->searched[1][out->ref1_frame - f->frame - 1];
            out->i_satd = f->i_satd;
            out->frame_num_reset = la->frame_num_reset;
        }
        la->started = true;
        return X264HIP_LOOK_FRAME;
    }
needs:
    {
        // the asked-for task first, the speculative ones after it
        int n = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int i = 0; i < la->n_needs && n < max_need; i++)
                if (la->needs[i].speculative == pass) need[n++] = la->needs[i];
        if (n_need) *n_need = n;
        return X264HIP_LOOK_NEED;
    }
}

extern "C" void x264hip_lookahead_set_cost(x264hip_lookahead *la, int b, int p0, int p1, int score, int intra_mbs, int cost00, int speculative)
{
    if (b < 0 || b >= la->i_input || la->i_input - b > RING || b - p0 < 0 || b - p0 > BF_MAX + 1 || p1 - b < 0 || p1 - b > BF_MAX + 1) return;
    LF *f = &la->ring[b % RING];
    if (b != p0) f->dev_searched[0][b - p0 - 1] = true;
    if (b != p1) f->dev_searched[1][p1 - b - 1] = true;
    if (speculative) {
        f->spec_cost[b - p0][p1 - b] = score;
        if (b == p1) { f->spec_intra_mbs[b - p0] = intra_mbs; f->spec_cost00 = cost00; }
        return;
    }
    f->cost_est[b - p0][p1 - b] = score;
    if (b == p1) { f->intra_mbs[b - p0] = intra_mbs; f->cost_est[0][0] = cost00; }
    if (b != p0) f->searched[0][b - p0 - 1] = true;
    if (b != p1) f->searched[1][p1 - b - 1] = true;
}

extern "C" void x264hip_lookahead_end(x264hip_lookahead *la)
{
    if (!la->started) return;
    rc_end(la);
    LF *f = la->fenc;
    la->i_frame++;                                                                               // x264_reference_update, encoder.c:1062-1093
    if (f->kept_as_ref) {
        if (la->slice_type != ST_B) la->last_nonb = f;
        list_push(la->reference, f);
        if (la->reference[la->i_max_dpb]) list_shift(la->reference);
    }
    la->fenc = nullptr;
    la->started = false;
    la->setup = false;
}

// x264_frame_sort (R/common/frame.c:957-975): by input number, or by type then input number
static void list_sort(LF **l, int b_dts)
{
    bool ok;
    if (!l[0]) return;
    do {
        ok = true;
        for (int i = 0; l[i + 1]; i++) {
            const int dtype = l[i]->type - l[i + 1]->type, dtime = l[i]->frame - l[i + 1]->frame;
            if (b_dts ? dtype > 0 || (dtype == 0 && dtime > 0) : dtime > 0) { LF *t = l[i]; l[i] = l[i + 1]; l[i + 1] = t; ok = false; }
        }
    } while (!ok);
}

// The post-encode scene cut found the P picture just coded no better than an intra picture (x264hip_scenecut_post; R/encoder/encoder.c:1645-1699): the
// attempt is given up -- no x264_ratecontrol_end, no x264_reference_update for it -- and the next x264hip_lookahead_get hands out what is coded
// instead: the same picture as I / IDR, or, when B pictures wait before it, the last of them as the P (the given-up picture goes back to the head of
// the undecided queue).  Returns 1 (same picture again), 2 (another picture), -1 if no P picture is being coded.
extern "C" int x264hip_lookahead_scenecut(x264hip_lookahead *la)
{
    if (!la->started || !la->fenc || la->slice_type != ST_P) return -1;
    LF *f = la->fenc;
    const int gop = f->frame - la->i_last_idr;
    int b = 0, ret = 1;
    while (la->current[b] && IS_TYPE_B(la->current[b]->type)) b++;
    la->frame_num_reset = 0;
    if (b > 0) {
        if (la->p.b_adapt || b > 1) f->type = T_AUTO;
        list_sort(la->current, 0);
        int n = list_len(la->next);                                                              // x264_frame_unshift
        la->next[n + 1] = nullptr;
        for (int i = n; i > 0; i--) la->next[i] = la->next[i - 1];
        la->next[0] = f;
        la->fenc = la->current[b - 1];
        la->current[b - 1] = nullptr;
        la->fenc->type = T_P;
        list_sort(la->current, 1);
        ret = 2;
    } else if (gop >= la->p.keyint_min) {
        f->type = T_IDR;
        f->poc = 0;
        while (la->current[0]) list_push(la->next, list_shift(la->current));
        list_sort(la->next, 0);
        la->frame_num_reset = 1;
    } else
        f->type = T_I;
    la->started = false;
    la->setup = false;
    return ret;
}

// A copy of everything x264hip_lookahead_end / _put / _get change, so that a caller may run them AHEAD of the post-encode scene cut's verdict (beside
// the sweep whose P picture is being judged) and come back if the verdict is "give up": the queues (pointers into the ring), the frame in flight and
// its lists, the rate control, and of every queued picture the fields a decision rewrites.  Costs that arrived meanwhile stay (they are facts).
namespace {
enum { SAVED_LF = 96 };
struct Saved {
    LF *next[MAX_LENGTH + 8], *current[MAX_LENGTH + 8], *reference[16 + 2];
    LF *last_nonb, *fenc, *fref0[16 + 2], *fref1[16 + 2];
    int n_ref0, n_ref1, i_input, i_last_idr, i_frame, slice_type, frame_num_reset;
    bool started, setup;
    RC rc;
    int n_lf;
    struct { LF *f; LF copy; } lf[SAVED_LF];             // every picture a queue, the DPB or the frame in flight names: whole records (types, costs, what is "searched")
};
}
extern "C" size_t x264hip_lookahead_state_bytes(void) { return sizeof(Saved); }
extern "C" int x264hip_lookahead_save(const x264hip_lookahead *la, void *buf)
{
    Saved *s = (Saved *)buf;
    memcpy(s->next, la->next, sizeof(s->next)); memcpy(s->current, la->current, sizeof(s->current)); memcpy(s->reference, la->reference, sizeof(s->reference));
    s->last_nonb = la->last_nonb; s->fenc = la->fenc; memcpy(s->fref0, la->fref0, sizeof(s->fref0)); memcpy(s->fref1, la->fref1, sizeof(s->fref1));
    s->n_ref0 = la->n_ref0; s->n_ref1 = la->n_ref1; s->i_input = la->i_input; s->i_last_idr = la->i_last_idr; s->i_frame = la->i_frame;
    s->slice_type = la->slice_type; s->frame_num_reset = la->frame_num_reset; s->started = la->started; s->setup = la->setup; s->rc = la->rc;
    s->n_lf = 0;
    bool full = false;
    auto keep = [&](LF *f) {
        if (!f) return;
        for (int i = 0; i < s->n_lf; i++) if (s->lf[i].f == f) return;
        if (s->n_lf >= SAVED_LF) { full = true; return; }
        s->lf[s->n_lf].f = f; s->lf[s->n_lf].copy = *f; s->n_lf++;
    };
    for (int i = 0; la->next[i]; i++) keep(la->next[i]);
    for (int i = 0; la->current[i]; i++) keep(la->current[i]);
    for (int i = 0; la->reference[i]; i++) keep(la->reference[i]);
    keep(la->fenc); keep(la->last_nonb);
    return full ? -1 : 0;
}
// back to the saved state; pictures that came in since (x264hip_lookahead_put) are queued again as put left them, in input order
extern "C" void x264hip_lookahead_restore(x264hip_lookahead *la, const void *buf)
{
    const Saved *s = (const Saved *)buf;
    const int i_input_now = la->i_input;
    memcpy(la->next, s->next, sizeof(s->next)); memcpy(la->current, s->current, sizeof(s->current)); memcpy(la->reference, s->reference, sizeof(s->reference));
    la->last_nonb = s->last_nonb; la->fenc = s->fenc; memcpy(la->fref0, s->fref0, sizeof(s->fref0)); memcpy(la->fref1, s->fref1, sizeof(s->fref1));
    la->n_ref0 = s->n_ref0; la->n_ref1 = s->n_ref1; la->i_last_idr = s->i_last_idr; la->i_frame = s->i_frame;
    la->slice_type = s->slice_type; la->frame_num_reset = s->frame_num_reset; la->started = s->started; la->setup = s->setup; la->rc = s->rc;
    for (int i = 0; i < s->n_lf; i++) *s->lf[i].f = s->lf[i].copy;
    la->i_input = s->i_input;
    for (int n = s->i_input; n < i_input_now; n++) x264hip_lookahead_put(la);
    la->n_needs = 0; la->miss = false;
}

extern "C" int x264hip_lookahead_oldest_live(const x264hip_lookahead *la)
{
    int m = la->i_input;
    for (int i = 0; la->next[i]; i++) if (la->next[i]->frame < m) m = la->next[i]->frame;
    for (int i = 0; la->current[i]; i++) if (la->current[i]->frame < m) m = la->current[i]->frame;
    if (la->fenc && la->fenc->frame < m) m = la->fenc->frame;
    if (la->last_nonb && la->last_nonb->frame < m) m = la->last_nonb->frame;
    if (la->fenc) {
        if (la->n_ref0 && la->fref0[0]->frame < m) m = la->fref0[0]->frame;
        if (la->n_ref1 && la->fref1[0]->frame < m) m = la->fref1[0]->frame;
    }
    return m;
}
