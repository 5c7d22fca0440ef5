// l1_tables.hip -- TABLE LEVEL of the C ABI: the reference's six DSP tables
// filled with entries that keep the reference's exact signatures (host
// pointers in, host pointers out) and do the arithmetic on the GPU.
//
// Mechanics of one call: the wrapper gathers the operand footprint into the
// calling thread's pinned, device-mapped arena, launches k_l1 on the thread's
// stream (one workgroup; the kernel reads/writes the arena directly over the
// host link, no staging copies), waits, and scatters the documented outputs
// back.  Exact and re-entrant, but one launch + sync per call: this level
// exists so the library is a drop-in behind x264_*_init (R/encoder/encoder.c:
// 730-745) and so parity can be tested entry by entry like R/tools/checkasm.c;
// throughput comes from the frame level (frame_*.hip).
//
// Entries with no arithmetic (copy[], plane_copy, memcpy/memzero, prefetch,
// and the pointer-returning branch of get_ref) stay host-side memory moves,
// as they are in every back-end of the reference.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "device_prims.h"
#include "intra_pred.h"
#include "internal.h"

using namespace x264hip;

#define FENC X264HIP_FENC_STRIDE
#define FDEC X264HIP_FDEC_STRIDE

enum {
    OP_CMP = 1, OP_VAR, OP_HAC, OP_SSIM_CORE, OP_SSIM_END4, OP_ADS,
    OP_SUB_DCT4, OP_SUB_DCT8, OP_ADD_IDCT4, OP_ADD_IDCT8, OP_ADD_DC, OP_DC4X4,
    OP_SCAN, OP_ZSUB, OP_INTERLEAVE,
    OP_QUANT, OP_DEQUANT, OP_DENOISE, OP_DECIMATE, OP_COEFF_LAST, OP_LEVEL_RUN,
    OP_AVG, OP_MC_CHROMA, OP_HPEL, OP_INTEGRAL_H, OP_INTEGRAL_4V, OP_INTEGRAL_8V, OP_LOWRES,
    OP_PRED, OP_PRED8, OP_PRED8_FILTER, OP_DEBLOCK
};
enum { CMP_SAD = 0, CMP_SSD, CMP_SATD, CMP_SA8D };

struct L1Args {
    int op;
    int p[12];
    u32 off[10];
};

// ------------------------------------------------------------------ device
// z-order position of 4x4 block k inside a 16x16 (also valid for 8x8: k < 4)
__device__ __forceinline__ void blk4_xy(int k, int &x, int &y)
{
    x = ((k >> 2) & 1) * 8 + (k & 1) * 4;
    y = (k >> 3) * 8 + ((k >> 1) & 1) * 4;
}

// one line of a deblocking edge; v[0..7] = p3 p2 p1 p0 q0 q1 q2 q3
// (luma) or v[2..5] = p1 p0 q0 q1 (chroma).  R/common/frame.c:420-586.
__device__ void deblock_line(int kind, int *v, int alpha, int beta, int tc0)
{
    int p2 = v[1], p1 = v[2], p0 = v[3], q0 = v[4], q1 = v[5], q2 = v[6];
    if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    if (kind == 0) {            // luma, bS < 4
        int tc = tc0;
        if (iabs(p2 - p0) < beta) { v[2] = p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0); tc++; }
        if (iabs(q2 - q0) < beta) { v[5] = q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0); tc++; }
        int d = clip3((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
        v[3] = clip_u8(p0 + d); v[4] = clip_u8(q0 - d);
    } else if (kind == 1) {     // chroma, bS < 4 (tc0 already includes +1)
        int d = clip3((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc0, tc0);
        v[3] = clip_u8(p0 + d); v[4] = clip_u8(q0 - d);
    } else if (kind == 2) {     // luma intra
        if (iabs(p0 - q0) < (alpha >> 2) + 2) {
            if (iabs(p2 - p0) < beta) {
                int p3 = v[0];
                v[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
                v[2] = (p2 + p1 + p0 + q0 + 2) >> 2;
                v[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
            } else
                v[3] = (2 * p1 + p0 + q1 + 2) >> 2;
            if (iabs(q2 - q0) < beta) {
                int q3 = v[7];
                v[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
                v[5] = (p0 + q0 + q1 + q2 + 2) >> 2;
                v[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
            } else
                v[4] = (2 * q1 + q0 + p1 + 2) >> 2;
        } else {
            v[3] = (2 * p1 + p0 + q1 + 2) >> 2;
            v[4] = (2 * q1 + q0 + p1 + 2) >> 2;
        }
    } else {                    // chroma intra
        v[3] = (2 * p1 + p0 + q1 + 2) >> 2;
        v[4] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
}

__global__ __launch_bounds__(256) void k_l1(L1Args a, u8 *arena)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool w0 = tid < 64;
    __shared__ int s_tmp[64];
    switch (a.op) {
    case OP_CMP: {
        if (!w0) break;
        const int kind = a.p[0], w = a.p[1], h = a.p[2], nref = a.p[3], sa = a.p[4];
        const u8 *fenc = arena + a.off[0];
        int *out = (int *)(arena + a.off[9]);
        for (int r = 0; r < nref; r++) {
            const u8 *ref = arena + a.off[1 + r];
            int part = 0;
            if (kind == CMP_SAD || kind == CMP_SSD) {
                int gpr = w >> 2, ng = gpr * h;
                if (lane < ng) {
                    int y = lane / gpr, x = (lane % gpr) * 4;
                    if (kind == CMP_SAD)
                        part = (int)sad4(load4u(fenc + y * sa + x), load4u(ref + y * w + x), 0);
                    else
                        for (int i = 0; i < 4; i++) { int d = (int)fenc[y * sa + x + i] - (int)ref[y * w + x + i]; part += d * d; }
                }
            } else if (kind == CMP_SATD) {
                if (w == 4) {
                    if (lane < (h >> 2)) part = satd_4x4(fenc + lane * 4 * sa, sa, ref + lane * 4 * w, w);
                } else {
                    int bx = w >> 3, nb = bx * (h >> 2);
                    if (lane < nb) {
                        int x = (lane % bx) * 8, y = (lane / bx) * 4;
                        part = satd_8x4(fenc + y * sa + x, sa, ref + y * w + x, w);
                    }
                }
            } else {
                int bx = w >> 3, nb = bx * (h >> 3);
                if (lane < nb) {
                    int x = (lane % bx) * 8, y = (lane / bx) * 8;
                    part = sa8d_8x8_raw(fenc + y * sa + x, sa, ref + y * w + x, w);
                }
            }
            int tot = wave_sum(part);
            if (kind == CMP_SA8D) tot = (tot + 2) >> 2;
            if (lane == 0) out[r] = tot;
        }
        break;
    }
    case OP_VAR: {
        if (!w0) break;
        const int n = a.p[0], shift = a.p[1];
        const u8 *p = arena + a.off[0];
        u32 sum = 0, sqr = 0;
        if (lane < n)
            for (int x = 0; x < n; x++) { u32 v = p[lane * n + x]; sum += v; sqr += v * v; }
        sum = wave_sum_u32(sum); sqr = wave_sum_u32(sqr);
        if (lane == 0) *(int *)(arena + a.off[9]) = (int)(sqr - (sum * sum >> shift));
        break;
    }
    case OP_HAC: {
        if (!w0) break;
        const int w = a.p[0], h = a.p[1];
        const u8 *p = arena + a.off[0];
        int bx = w >> 3, nb = bx * (h >> 3);
        unsigned long long v = 0;
        if (lane < nb) v = hadamard_ac_8x8(p + (lane / bx) * 8 * w + (lane % bx) * 8, w);
        for (int m = 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0)
            *(unsigned long long *)(arena + a.off[9]) = ((v >> 34) << 32) + (unsigned long long)((u32)v >> 1);
        break;
    }
    case OP_SSIM_CORE: {
        if (tid < 2) {
            const u8 *p1 = arena + a.off[0] + 4 * tid, *p2 = arena + a.off[1] + 4 * tid;
            u32 s1 = 0, s2 = 0, ss = 0, s12 = 0;
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++) {
                    int u = p1[y * 8 + x], v = p2[y * 8 + x];
                    s1 += u; s2 += v; ss += u * u + v * v; s12 += u * v;
                }
            int *o = (int *)(arena + a.off[9]) + 4 * tid;
            o[0] = s1; o[1] = s2; o[2] = ss; o[3] = s12;
        }
        break;
    }
    case OP_SSIM_END4: {
        // float; summed strictly left to right as R/common/pixel.c:458-468 does
        if (tid == 0) {
            const int *s0 = (const int *)(arena + a.off[0]), *s1 = (const int *)(arena + a.off[1]);
            const int c1 = 416, c2 = 235963;   // (int)(.01*.01*255*255*64+.5), (int)(.03*.03*255*255*64*63+.5)
            float acc = 0.0f;
            for (int i = 0; i < a.p[0]; i++) {
                int t[4];
                for (int k = 0; k < 4; k++) t[k] = s0[4 * i + k] + s0[4 * i + 4 + k] + s1[4 * i + k] + s1[4 * i + 4 + k];
                int vars = t[2] * 64 - t[0] * t[0] - t[1] * t[1];
                int covar = t[3] * 64 - t[0] * t[1];
                float num = __fmul_rn((float)(2 * t[0] * t[1] + c1), (float)(2 * covar + c2));
                float den = __fmul_rn((float)(t[0] * t[0] + t[1] * t[1] + c1), (float)(vars + c2));
                acc = __fadd_rn(acc, __fdiv_rn(num, den));
            }
            *(float *)(arena + a.off[9]) = acc;
        }
        break;
    }
    case OP_ADS: {
        if (!w0) break;
        const int n = a.p[0], delta = a.p[1], width = a.p[2], thresh = a.p[3];
        const int *dc = (const int *)(arena + a.off[0]);
        const u16 *sums = (const u16 *)(arena + a.off[1]);
        const u16 *cost = (const u16 *)(arena + a.off[2]);
        i16 *mvs = (i16 *)(arena + a.off[8]);
        int found = 0;
        for (int base = 0; base < width; base += 64) {
            int i = base + lane;
            bool keep = false;
            if (i < width) {
                int v = iabs(dc[0] - (int)sums[i]) + (int)cost[i];
                if (n == 2) v += iabs(dc[1] - (int)sums[i + delta]);
                if (n == 4) v += iabs(dc[1] - (int)sums[i + 8]) + iabs(dc[2] - (int)sums[i + delta]) + iabs(dc[3] - (int)sums[i + delta + 8]);
                keep = v < thresh;
            }
            unsigned long long m = __ballot(keep);
            if (keep) mvs[found + __popcll(m & ((1ull << lane) - 1))] = (i16)i;
            found += __popcll(m);
        }
        if (lane == 0) *(int *)(arena + a.off[9]) = found;
        break;
    }
    case OP_SUB_DCT4: {
        const int nblk = a.p[0], n = a.p[1];     // n = block edge of the whole call (4, 8, 16)
        if (tid < nblk) {
            const u8 *p1 = arena + a.off[0], *p2 = arena + a.off[1];
            int x, y, r[16];
            blk4_xy(tid, x, y);
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++)
                    r[4 * j + i] = (int)p1[(y + j) * n + x + i] - (int)p2[(y + j) * n + x + i];
            fwd4x4((i16 *)(arena + a.off[9]) + 16 * tid, r);
        }
        break;
    }
    case OP_SUB_DCT8: {
        const int nblk = a.p[0], n = a.p[1];
        if (tid < nblk) {
            const u8 *p1 = arena + a.off[0] + (tid >> 1) * 8 * n + (tid & 1) * 8;
            const u8 *p2 = arena + a.off[1] + (tid >> 1) * 8 * n + (tid & 1) * 8;
            i16 *out = (i16 *)(arena + a.off[9]) + 64 * tid;
            i16 t[64];
            int s[8], o[8];
            for (int c = 0; c < 8; c++) {           // columns first, narrowed to int16 (dct.c:270-276)
                for (int k = 0; k < 8; k++) s[k] = (int)p1[k * n + c] - (int)p2[k * n + c];
                fwd8_1d(o, s);
                for (int k = 0; k < 8; k++) t[k * 8 + c] = (i16)o[k];
            }
            for (int r = 0; r < 8; r++) {           // rows, stored transposed (dct.c:278-283)
                for (int k = 0; k < 8; k++) s[k] = t[r * 8 + k];
                fwd8_1d(o, s);
                for (int k = 0; k < 8; k++) out[k * 8 + r] = (i16)o[k];
            }
        }
        break;
    }
    case OP_ADD_IDCT4: {
        const int nblk = a.p[0], n = a.p[1];
        if (tid < nblk) {
            u8 *dst = arena + a.off[0];
            int x, y, res[16];
            blk4_xy(tid, x, y);
            inv4x4(res, (const i16 *)(arena + a.off[1]) + 16 * tid);
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) {
                    u8 *d = dst + (y + j) * n + x + i;
                    *d = (u8)clip_u8((int)*d + res[4 * j + i]);
                }
        }
        break;
    }
    case OP_ADD_IDCT8: {
        const int nblk = a.p[0], n = a.p[1];
        if (tid < nblk) {
            u8 *dst = arena + a.off[0] + (tid >> 1) * 8 * n + (tid & 1) * 8;
            i16 *d = (i16 *)(arena + a.off[1]) + 64 * tid;
            int s[8], o[8];
            d[0] = (i16)(d[0] + 32);                 // rounding term, dct.c:326
            for (int c = 0; c < 8; c++) {
                for (int k = 0; k < 8; k++) s[k] = d[k * 8 + c];
                inv8_1d(o, s);
                for (int k = 0; k < 8; k++) d[k * 8 + c] = (i16)o[k];
            }
            for (int r = 0; r < 8; r++) {
                for (int k = 0; k < 8; k++) s[k] = d[r * 8 + k];
                inv8_1d(o, s);
                for (int k = 0; k < 8; k++) {
                    u8 *px = dst + r + k * n;
                    *px = (u8)clip_u8((int)*px + (o[k] >> 6));
                }
            }
        }
        break;
    }
    case OP_ADD_DC: {
        const int nx = a.p[0], nblk = a.p[1], n = a.p[2];
        if (tid < nblk) {
            u8 *dst = arena + a.off[0] + (tid / nx) * 4 * n + (tid % nx) * 4;
            int dc = (int)(i16)((((const i16 *)(arena + a.off[1]))[tid] + 32) >> 6);
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) dst[j * n + i] = (u8)clip_u8((int)dst[j * n + i] + dc);
        }
        break;
    }
    case OP_DC4X4: {
        // luma-DC Hadamard, int16 intermediate (dct.c:39-105); 4 lanes = 4 rows per pass
        i16 *d = (i16 *)(arena + a.off[0]);
        const int rnd = a.p[0];
        if (tid < 4) {
            int r = tid;
            int x = d[4 * r] + d[4 * r + 1], y = d[4 * r] - d[4 * r + 1], z = d[4 * r + 2] + d[4 * r + 3], u = d[4 * r + 2] - d[4 * r + 3];
            s_tmp[0 + r] = (i16)(x + z); s_tmp[4 + r] = (i16)(x - z); s_tmp[8 + r] = (i16)(y - u); s_tmp[12 + r] = (i16)(y + u);
        }
        __syncthreads();
        if (tid < 4) {
            int r = tid;
            int x = s_tmp[4 * r] + s_tmp[4 * r + 1], y = s_tmp[4 * r] - s_tmp[4 * r + 1];
            int z = s_tmp[4 * r + 2] + s_tmp[4 * r + 3], u = s_tmp[4 * r + 2] - s_tmp[4 * r + 3];
            d[4 * r] = (i16)((x + z + rnd) >> rnd); d[4 * r + 1] = (i16)((x - z + rnd) >> rnd);
            d[4 * r + 2] = (i16)((y - u + rnd) >> rnd); d[4 * r + 3] = (i16)((y + u + rnd) >> rnd);
        }
        break;
    }
    case OP_SCAN: {
        const int n = a.p[0], field = a.p[1];
        if (tid < n) {
            const i16 *src = (const i16 *)(arena + a.off[0]);
            ((i16 *)(arena + a.off[9]))[tid] = src[n == 64 ? c_scan8[field][tid] : c_scan4[field][tid]];
        }
        break;
    }
    case OP_ZSUB: {
        // residual in scan order, then source copied over the prediction (dct.c:564-606)
        const int n = a.p[0], field = a.p[1], nn = n * n;
        const u8 *src = arena + a.off[0];
        u8 *dst = arena + a.off[1];
        int lv = 0, cp = 0;
        if (tid < nn) {
            int k = n == 8 ? c_scan8[field][tid] : c_scan4[field][tid];
            int x = n == 8 ? k >> 3 : k >> 2, y = k & (n - 1);
            lv = (int)src[y * n + x] - (int)dst[y * n + x];
            cp = src[tid];
        }
        __syncthreads();
        if (tid < nn) { ((i16 *)(arena + a.off[9]))[tid] = (i16)lv; dst[tid] = (u8)cp; }
        break;
    }
    case OP_INTERLEAVE: {
        if (!w0) break;
        const i16 *src = (const i16 *)(arena + a.off[0]);
        int g = lane >> 4, j = lane & 15;
        i16 v = src[g + 4 * j];
        ((i16 *)(arena + a.off[9]))[16 * g + j] = v;
        unsigned long long m = __ballot(v != 0);
        if (j == 0) (arena + a.off[8])[g] = (u8)(((m >> (16 * g)) & 0xffffull) != 0);
        break;
    }
    case OP_QUANT: {
        if (!w0) break;
        const int n = a.p[0], scalar = a.p[1];
        i16 *d = (i16 *)(arena + a.off[0]);
        int q = 0;
        if (lane < n) {
            int mf = scalar ? a.p[2] : (int)((const u16 *)(arena + a.off[1]))[lane];
            int bs = scalar ? a.p[3] : (int)((const u16 *)(arena + a.off[2]))[lane];
            q = quant_one(d[lane], mf, bs);
            d[lane] = (i16)q;
        }
        unsigned long long m = __ballot(q != 0);
        if (lane == 0) *(int *)(arena + a.off[9]) = m != 0;
        break;
    }
    case OP_DEQUANT: {
        const int n = a.p[0], base = a.p[1], qp = a.p[2], dcmode = a.p[3];
        if (tid < n) {
            i16 *d = (i16 *)(arena + a.off[0]);
            const int *mf = (const int *)(arena + a.off[1]);
            int bits = qp / 6 - base;
            if (!dcmode) d[tid] = (i16)dequant_one(d[tid], mf[tid], bits);
            else if (bits >= 0) d[tid] = (i16)((int)d[tid] * (mf[0] << bits));
            else d[tid] = (i16)(((int)d[tid] * mf[0] + (1 << (-bits - 1))) >> -bits);
        }
        break;
    }
    case OP_DENOISE: {
        const int size = a.p[0];
        if (tid >= 1 && tid < size) {
            i16 *d = (i16 *)(arena + a.off[0]);
            u32 *sum = (u32 *)(arena + a.off[1]);
            const u16 *off = (const u16 *)(arena + a.off[2]);
            int v = d[tid], neg = v >> 15, mag = (v + neg) ^ neg;
            sum[tid] += (u32)mag;
            mag -= (int)off[tid];
            d[tid] = mag < 0 ? (i16)0 : (i16)((mag ^ neg) - neg);
        }
        break;
    }
    case OP_DECIMATE: {
        // JVT-B118 score from the run lengths between non-zero levels (quant.c:213-239):
        // lanes vote which coefficients are non-zero / large, lane 0 walks the bit mask.
        if (!w0) break;
        const int n = a.p[0];
        const i16 *d = (const i16 *)(arena + a.off[0]);
        int v = lane < n ? (int)d[lane] : 0;
        unsigned long long nzm = __ballot(v != 0), big = __ballot((unsigned)(v + 1) > 2u);
        if (lane == 0) {
            int score = 0;
            if (big) score = 9;
            else {
                int idx = nzm ? 63 - __clzll(nzm) : -1;
                while (idx >= 0) {
                    unsigned long long below = idx ? (nzm & ((1ull << idx) - 1)) : 0ull;
                    int prev = below ? 63 - __clzll(below) : -1;
                    int run = idx - prev - 1;
                    score += n == 64 ? c_decimate8[run] : c_decimate4[run];
                    idx = prev;
                }
            }
            *(int *)(arena + a.off[9]) = score;
        }
        break;
    }
    case OP_COEFF_LAST: {
        if (!w0) break;
        const i16 *d = (const i16 *)(arena + a.off[0]);
        unsigned long long m = __ballot(lane < a.p[0] && d[lane] != 0);
        if (lane == 0) *(int *)(arena + a.off[9]) = m ? 63 - __clzll(m) : -1;
        break;
    }
    case OP_LEVEL_RUN: {
        // run/level pairs from last to first non-zero (quant.c:282-296); n <= 16
        if (!w0) break;
        const i16 *d = (const i16 *)(arena + a.off[0]);
        x264hip_run_level_t *rl = (x264hip_run_level_t *)(arena + a.off[8]);
        unsigned long long m = __ballot(lane < a.p[0] && d[lane] != 0);
        int total = __popcll(m);
        int last = m ? 63 - __clzll(m) : -1;
        if (lane < a.p[0] && d[lane] != 0) {
            int rank = __popcll(m >> lane) - 1;          // 0 for the last non-zero
            unsigned long long below = lane ? (m & ((1ull << lane) - 1)) : 0ull;
            int prev = below ? 63 - __clzll(below) : -1;
            rl->level[rank] = d[lane];
            rl->run[rank] = (u8)(lane - prev - 1);
        }
        if (lane == 0) {
            rl->last = last;      // callers only pass blocks that hold a coefficient
            *(int *)(arena + a.off[9]) = total;
        }
        break;
    }
    case OP_AVG: {
        const int w = a.p[0], h = a.p[1], wt = a.p[2];
        const u8 *s1 = arena + a.off[0], *s2 = arena + a.off[1];
        u8 *dst = arena + a.off[9];
        for (int i = tid; i < w * h; i += 256)
            dst[i] = wt == 32 ? (u8)(((int)s1[i] + (int)s2[i] + 1) >> 1)
                              : (u8)clip_u8(((int)s1[i] * wt + (int)s2[i] * (64 - wt) + 32) >> 6);
        break;
    }
    case OP_MC_CHROMA: {
        const int w = a.p[0], h = a.p[1], dx = a.p[2], dy = a.p[3], ls = w + 1;
        const u8 *s = arena + a.off[0];
        u8 *dst = arena + a.off[9];
        int ca = (8 - dx) * (8 - dy), cb = dx * (8 - dy), cc = (8 - dx) * dy, cd = dx * dy;
        for (int i = tid; i < w * h; i += 256) {
            int x = i % w, y = i / w;
            dst[i] = (u8)((ca * s[y * ls + x] + cb * s[y * ls + x + 1] + cc * s[(y + 1) * ls + x] + cd * s[(y + 1) * ls + x + 1] + 32) >> 6);
        }
        break;
    }
    case OP_HPEL: {
        // local source: rows -2..height+2, columns -2..width+2, stride ls
        const int width = a.p[0], height = a.p[1], ls = width + 5;
        const u8 *src = arena + a.off[0] + 2 * ls + 2;         // -> sample (0,0)
        i16 *vraw = (i16 *)(arena + a.off[1]);                 // [height][ls] raw vertical taps
        u8 *dv = arena + a.off[7], *dh = arena + a.off[8], *dc = arena + a.off[9];
        for (int i = tid; i < height * ls; i += 256) {
            int y = i / ls, x = i % ls - 2;
            const u8 *p = src + y * ls + x;
            int v = tap6(p[-2 * ls], p[-ls], p[0], p[ls], p[2 * ls], p[3 * ls]);
            dv[i] = (u8)clip_u8((v + 16) >> 5);
            vraw[i] = (i16)v;
        }
        __syncthreads();
        for (int i = tid; i < height * width; i += 256) {
            int y = i / width, x = i % width;
            const i16 *b = vraw + y * ls + x + 2;
            dc[i] = (u8)clip_u8((tap6(b[-2], b[-1], b[0], b[1], b[2], b[3]) + 512) >> 10);
            const u8 *p = src + y * ls + x;
            dh[i] = (u8)clip_u8((tap6(p[-2], p[-1], p[0], p[1], p[2], p[3]) + 16) >> 5);
        }
        break;
    }
    case OP_INTEGRAL_H: {
        const int n = a.p[0], cnt = a.p[1];
        const u8 *pix = arena + a.off[0];
        const u16 *prev = (const u16 *)(arena + a.off[1]);
        u16 *out = (u16 *)(arena + a.off[9]);
        for (int x = tid; x < cnt; x += 256) {
            int v = 0;
            for (int k = 0; k < n; k++) v += pix[x + k];
            out[x] = (u16)(v + prev[x]);
        }
        break;
    }
    case OP_INTEGRAL_4V: {
        const int cnt = a.p[0];
        const u16 *r0 = (const u16 *)(arena + a.off[0]), *r4 = (const u16 *)(arena + a.off[1]), *r8 = (const u16 *)(arena + a.off[2]);
        u16 *o4 = (u16 *)(arena + a.off[8]), *o8 = (u16 *)(arena + a.off[9]);
        for (int x = tid; x < cnt; x += 256) {
            o4[x] = (u16)(r4[x] - r0[x]);
            o8[x] = (u16)(r8[x] + r8[x + 4] - r0[x] - r0[x + 4]);
        }
        break;
    }
    case OP_INTEGRAL_8V: {
        const int cnt = a.p[0];
        const u16 *r0 = (const u16 *)(arena + a.off[0]), *r8 = (const u16 *)(arena + a.off[1]);
        u16 *o = (u16 *)(arena + a.off[9]);
        for (int x = tid; x < cnt; x += 256) o[x] = (u16)(r8[x] - r0[x]);
        break;
    }
    case OP_LOWRES: {
        const int w = a.p[0], h = a.p[1], ls = 2 * w + 1;
        const u8 *s = arena + a.off[0];
        u8 *d0 = arena + a.off[6], *dh = arena + a.off[7], *dv = arena + a.off[8], *dc = arena + a.off[9];
        for (int i = tid; i < w * h; i += 256) {
            int x = i % w, y = i / w;
            const u8 *r0 = s + 2 * y * ls + 2 * x, *r1 = r0 + ls, *r2 = r1 + ls;
            d0[i] = (u8)avg4r(r0[0], r1[0], r0[1], r1[1]);
            dh[i] = (u8)avg4r(r0[1], r1[1], r0[2], r1[2]);
            dv[i] = (u8)avg4r(r1[0], r2[0], r1[1], r2[1]);
            dc[i] = (u8)avg4r(r1[1], r2[1], r1[2], r2[2]);
        }
        break;
    }
    case OP_PRED: {
        const int fam = a.p[0], mode = a.p[1], ls = a.p[2];
        const int n = fam == 0 ? 16 : fam == 1 ? 8 : 4;
        u8 *blk = arena + a.off[0] + ls + 1;
        int v = 0;
        if (tid < n * n) v = pred_px(fam, mode, blk, ls, tid % n, tid / n);
        __syncthreads();
        if (tid < n * n) (arena + a.off[9])[tid] = (u8)v;
        break;
    }
    case OP_PRED8: {
        const int mode = a.p[0];
        const u8 *edge = arena + a.off[0];
        if (tid < 64) {
            int x = tid & 7, y = tid >> 3, v;
            if (mode == 0) v = edge[16 + x];
            else if (mode == 1) v = edge[14 - y];
            else if (mode == 11) v = 128;
            else if (mode == 2 || mode == 9 || mode == 10) {
                int l = 0, t = 0;
                for (int i = 0; i < 8; i++) { l += edge[7 + i]; t += edge[16 + i]; }
                v = mode == 2 ? (l + t + 8) >> 4 : mode == 9 ? (l + 4) >> 3 : (t + 4) >> 3;
            } else {
                int e[25];
                for (int i = 0; i < 25; i++) e[i] = edge[7 + i];
                v = dir_pred_px(8, mode, e, x, y);
            }
            (arena + a.off[9])[tid] = (u8)v;
        }
        break;
    }
    case OP_PRED8_FILTER: {
        // R/common/predict.c:499-540; local buffer stride 17: row -1 then rows 0..7, col -1 first
        if (tid == 0) {
            const int neigh = a.p[0], filt = a.p[1], ls = 17;
            const u8 *s = arena + a.off[0] + ls + 1;
            u8 *edge = arena + a.off[9];
#define PX(xx, yy) ((int)s[(xx) + (yy) * ls])
            int have_tl = neigh & X264HIP_MB_TOPLEFT;
            if (filt & X264HIP_MB_LEFT) {
                edge[15] = (u8)((PX(0, -1) + 2 * PX(-1, -1) + PX(-1, 0) + 2) >> 2);
                edge[14] = (u8)(((have_tl ? PX(-1, -1) : PX(-1, 0)) + 2 * PX(-1, 0) + PX(-1, 1) + 2) >> 2);
                for (int y = 1; y < 7; y++) edge[14 - y] = (u8)((PX(-1, y - 1) + 2 * PX(-1, y) + PX(-1, y + 1) + 2) >> 2);
                edge[7] = (u8)((PX(-1, 6) + 3 * PX(-1, 7) + 2) >> 2);
            }
            if (filt & X264HIP_MB_TOP) {
                int have_tr = neigh & X264HIP_MB_TOPRIGHT;
                edge[16] = (u8)(((have_tl ? PX(-1, -1) : PX(0, -1)) + 2 * PX(0, -1) + PX(1, -1) + 2) >> 2);
                for (int x = 1; x < 7; x++) edge[16 + x] = (u8)((PX(x - 1, -1) + 2 * PX(x, -1) + PX(x + 1, -1) + 2) >> 2);
                edge[23] = (u8)((PX(6, -1) + 2 * PX(7, -1) + (have_tr ? PX(8, -1) : PX(7, -1)) + 2) >> 2);
                if (filt & X264HIP_MB_TOPRIGHT) {
                    if (have_tr) {
                        for (int x = 8; x < 15; x++) edge[16 + x] = (u8)((PX(x - 1, -1) + 2 * PX(x, -1) + PX(x + 1, -1) + 2) >> 2);
                        edge[31] = edge[32] = (u8)((PX(14, -1) + 3 * PX(15, -1) + 2) >> 2);
                    } else
                        for (int i = 24; i < 33; i++) edge[i] = (u8)PX(7, -1);
                }
            }
#undef PX
        }
        break;
    }
    case OP_DEBLOCK: {
        // lines gathered by the host as [line][8] = p3..q3 (luma) / [line][8] with p1..q1 at 2..5 (chroma)
        const int kind = a.p[0], nlines = a.p[1], alpha = a.p[2], beta = a.p[3];
        u8 *lines = arena + a.off[0];
        if (tid < nlines) {
            int g = kind == 0 ? tid >> 2 : tid >> 1;
            int tc0 = (kind == 0 || kind == 1) ? a.p[4 + g] : 0;
            bool skip = (kind == 0 && tc0 < 0) || (kind == 1 && tc0 <= 0);
            if (!skip) {
                int v[8];
                for (int i = 0; i < 8; i++) v[i] = lines[tid * 8 + i];
                deblock_line(kind, v, alpha, beta, tc0);
                for (int i = 0; i < 8; i++) lines[tid * 8 + i] = (u8)v[i];
            }
        }
        break;
    }
    default:
        break;
    }
}

// -------------------------------------------------------------------- host
namespace {

struct Call {
    ThreadCtx *t;
    L1Args a;
    explicit Call(int op) : t(thread_ctx())
    {
        memset(&a, 0, sizeof(a));
        a.op = op;
        t->top = 0;
    }
    u32 alloc(size_t n)
    {
        size_t o = (t->top + 15) & ~(size_t)15;
        if (o + n > t->cap) {
            fprintf(stderr, "x264hip: staging arena too small (%zu + %zu > %zu); raise x264hip_cfg.arena_bytes\n", o, n, t->cap);
            abort();
        }
        t->top = o + n;
        return (u32)o;
    }
    u8 *at(u32 off) { return t->host + off; }
    u32 in(const void *src, size_t n)
    {
        u32 o = alloc(n);
        memcpy(at(o), src, n);
        return o;
    }
    u32 zero(size_t n)
    {
        u32 o = alloc(n);
        memset(at(o), 0, n);
        return o;
    }
    // gather a w x h block into a compact (stride = w) image
    u32 in2d(const u8 *src, int stride, int w, int h)
    {
        u32 o = alloc((size_t)w * h);
        for (int y = 0; y < h; y++) memcpy(at(o) + (size_t)y * w, src + (ptrdiff_t)y * stride, w);
        return o;
    }
    void out2d(u32 o, u8 *dst, int stride, int w, int h)
    {
        for (int y = 0; y < h; y++) memcpy(dst + (ptrdiff_t)y * stride, at(o) + (size_t)y * w, w);
    }
    void run()
    {
        hipLaunchKernelGGL(k_l1, dim3(1), dim3(256), 0, t->stream, a, t->dev);
        hipError_t e = hipStreamSynchronize(t->stream);
        if (e != hipSuccess) {
            fprintf(stderr, "x264hip: table kernel failed: %s\n", hipGetErrorString(e));
            abort();
        }
    }
};

const int kW[10] = {16, 16, 8, 8, 8, 4, 4, 4, 2, 2};
const int kH[10] = {16, 8, 16, 8, 4, 8, 4, 2, 4, 2};

// ---- pixel ---------------------------------------------------------------
int cmp_n(int kind, int w, int h, u8 *fenc, int s1, u8 *const *refs, int s2, int nref, int *scores)
{
    Call c(OP_CMP);
    c.a.p[0] = kind; c.a.p[1] = w; c.a.p[2] = h; c.a.p[3] = nref; c.a.p[4] = w;
    c.a.off[0] = c.in2d(fenc, s1, w, h);
    for (int r = 0; r < nref; r++) c.a.off[1 + r] = c.in2d(refs[r], s2, w, h);
    c.a.off[9] = c.zero(16);
    c.run();
    const int *o = (const int *)c.at(c.a.off[9]);
    for (int r = 0; r < nref; r++) scores[r] = o[r];
    return o[0];
}
template <int KIND, int W, int H> int t_cmp(u8 *a, int sa, u8 *b, int sb)
{
    int sc[4];
    u8 *refs[1] = {b};
    return cmp_n(KIND, W, H, a, sa, refs, sb, 1, sc);
}
template <int KIND, int W, int H> void t_cmp_x3(u8 *f, u8 *p0, u8 *p1, u8 *p2, int s, int sc[3])
{
    u8 *refs[3] = {p0, p1, p2};
    int tmp[4];
    cmp_n(KIND, W, H, f, FENC, refs, s, 3, tmp);
    sc[0] = tmp[0]; sc[1] = tmp[1]; sc[2] = tmp[2];
}
template <int KIND, int W, int H> void t_cmp_x4(u8 *f, u8 *p0, u8 *p1, u8 *p2, u8 *p3, int s, int sc[4])
{
    u8 *refs[4] = {p0, p1, p2, p3};
    cmp_n(KIND, W, H, f, FENC, refs, s, 4, sc);
}
template <int N, int SHIFT> int t_var(u8 *p, int s)
{
    Call c(OP_VAR);
    c.a.p[0] = N; c.a.p[1] = SHIFT;
    c.a.off[0] = c.in2d(p, s, N, N);
    c.a.off[9] = c.zero(16);
    c.run();
    return *(const int *)c.at(c.a.off[9]);
}
template <int W, int H> uint64_t t_hac(u8 *p, int s)
{
    Call c(OP_HAC);
    c.a.p[0] = W; c.a.p[1] = H;
    c.a.off[0] = c.in2d(p, s, W, H);
    c.a.off[9] = c.zero(16);
    c.run();
    return *(const uint64_t *)c.at(c.a.off[9]);
}
void t_ssim_core(const u8 *p1, int s1, const u8 *p2, int s2, int sums[2][4])
{
    Call c(OP_SSIM_CORE);
    c.a.off[0] = c.in2d(p1, s1, 8, 4);
    c.a.off[1] = c.in2d(p2, s2, 8, 4);
    c.a.off[9] = c.zero(32);
    c.run();
    memcpy(sums, c.at(c.a.off[9]), 32);
}
float t_ssim_end4(int sum0[5][4], int sum1[5][4], int width)
{
    Call c(OP_SSIM_END4);
    c.a.p[0] = width;
    c.a.off[0] = c.in(sum0, (size_t)(width + 1) * 16);
    c.a.off[1] = c.in(sum1, (size_t)(width + 1) * 16);
    c.a.off[9] = c.zero(16);
    c.run();
    return *(const float *)c.at(c.a.off[9]);
}
template <int N> int t_ads(int dc[4], u16 *sums, int delta, u16 *cost, i16 *mvs, int width, int thresh)
{
    Call c(OP_ADS);
    c.a.p[0] = N; c.a.p[1] = delta; c.a.p[2] = width; c.a.p[3] = thresh;
    c.a.off[0] = c.in(dc, 16);
    size_t span = (size_t)width + (N == 1 ? 0 : delta) + (N == 4 ? 8 : 0);
    c.a.off[1] = c.in(sums, span * 2);
    c.a.off[2] = c.in(cost, (size_t)width * 2);
    c.a.off[8] = c.zero((size_t)width * 2);
    c.a.off[9] = c.zero(16);
    c.run();
    int n = *(const int *)c.at(c.a.off[9]);
    memcpy(mvs, c.at(c.a.off[8]), (size_t)n * 2);
    return n;
}

// ---- dct -------------------------------------------------------------------
template <int N, int DCT8> void t_sub_dct(i16 *dct, u8 *p1, u8 *p2)
{
    Call c(DCT8 ? OP_SUB_DCT8 : OP_SUB_DCT4);
    c.a.p[0] = DCT8 ? (N / 8) * (N / 8) : (N / 4) * (N / 4);
    c.a.p[1] = N;
    c.a.off[0] = c.in2d(p1, FENC, N, N);
    c.a.off[1] = c.in2d(p2, FDEC, N, N);
    c.a.off[9] = c.zero((size_t)N * N * 2);
    c.run();
    memcpy(dct, c.at(c.a.off[9]), (size_t)N * N * 2);
}
template <int N, int DCT8> void t_add_idct(u8 *dst, i16 *dct)
{
    Call c(DCT8 ? OP_ADD_IDCT8 : OP_ADD_IDCT4);
    c.a.p[0] = DCT8 ? (N / 8) * (N / 8) : (N / 4) * (N / 4);
    c.a.p[1] = N;
    c.a.off[0] = c.in2d(dst, FDEC, N, N);
    c.a.off[1] = c.in(dct, (size_t)N * N * 2);
    c.run();
    c.out2d(c.a.off[0], dst, FDEC, N, N);
    if (DCT8) memcpy(dct, c.at(c.a.off[1]), (size_t)N * N * 2);   // the reference's idct8 leaves its scratch in dct[]
}
template <int N> void t_add_dc(u8 *dst, i16 *dct)
{
    Call c(OP_ADD_DC);
    c.a.p[0] = N / 4; c.a.p[1] = (N / 4) * (N / 4); c.a.p[2] = N;
    c.a.off[0] = c.in2d(dst, FDEC, N, N);
    c.a.off[1] = c.in(dct, (size_t)(N / 4) * (N / 4) * 2);
    c.run();
    c.out2d(c.a.off[0], dst, FDEC, N, N);
}
template <int RND> void t_dc4x4(i16 *d)
{
    Call c(OP_DC4X4);
    c.a.p[0] = RND;
    c.a.off[0] = c.in(d, 32);
    c.run();
    memcpy(d, c.at(c.a.off[0]), 32);
}
template <int N, int FIELD> void t_scan(i16 *level, i16 *dct)
{
    Call c(OP_SCAN);
    c.a.p[0] = N; c.a.p[1] = FIELD;
    c.a.off[0] = c.in(dct, N * 2);
    c.a.off[9] = c.zero(N * 2);
    c.run();
    memcpy(level, c.at(c.a.off[9]), N * 2);
}
template <int N, int FIELD> void t_zsub(i16 *level, const u8 *src, u8 *dst)
{
    Call c(OP_ZSUB);
    c.a.p[0] = N; c.a.p[1] = FIELD;
    c.a.off[0] = c.in2d(src, FENC, N, N);
    c.a.off[1] = c.in2d(dst, FDEC, N, N);
    c.a.off[9] = c.zero(N * N * 2);
    c.run();
    memcpy(level, c.at(c.a.off[9]), N * N * 2);
    c.out2d(c.a.off[1], dst, FDEC, N, N);
}
void t_interleave(i16 *dst, i16 *src, u8 *nnz)
{
    Call c(OP_INTERLEAVE);
    c.a.off[0] = c.in(src, 128);
    c.a.off[8] = c.zero(16);
    c.a.off[9] = c.zero(128);
    c.run();
    memcpy(dst, c.at(c.a.off[9]), 128);
    const u8 *z = c.at(c.a.off[8]);
    nnz[0] = z[0]; nnz[1] = z[1]; nnz[8] = z[2]; nnz[9] = z[3];
}

// ---- quant -----------------------------------------------------------------
int quant_call(i16 *d, int n, const u16 *mf, const u16 *bias, int smf, int sbias)
{
    Call c(OP_QUANT);
    c.a.p[0] = n; c.a.p[1] = mf == nullptr; c.a.p[2] = smf; c.a.p[3] = sbias;
    c.a.off[0] = c.in(d, (size_t)n * 2);
    if (mf) { c.a.off[1] = c.in(mf, (size_t)n * 2); c.a.off[2] = c.in(bias, (size_t)n * 2); }
    c.a.off[9] = c.zero(16);
    c.run();
    memcpy(d, c.at(c.a.off[0]), (size_t)n * 2);
    return *(const int *)c.at(c.a.off[9]);
}
int t_quant_8x8(i16 d[8][8], u16 mf[64], u16 bias[64]) { return quant_call(&d[0][0], 64, mf, bias, 0, 0); }
int t_quant_4x4(i16 d[4][4], u16 mf[16], u16 bias[16]) { return quant_call(&d[0][0], 16, mf, bias, 0, 0); }
int t_quant_4x4_dc(i16 d[4][4], int mf, int bias) { return quant_call(&d[0][0], 16, nullptr, nullptr, mf, bias); }
int t_quant_2x2_dc(i16 d[2][2], int mf, int bias) { return quant_call(&d[0][0], 4, nullptr, nullptr, mf, bias); }
void dequant_call(i16 *d, int n, const int *mf_row, int base, int qp, int dcmode)
{
    Call c(OP_DEQUANT);
    c.a.p[0] = n; c.a.p[1] = base; c.a.p[2] = qp; c.a.p[3] = dcmode;
    c.a.off[0] = c.in(d, (size_t)n * 2);
    c.a.off[1] = c.in(mf_row, (size_t)(dcmode ? 1 : n) * 4);
    c.run();
    memcpy(d, c.at(c.a.off[0]), (size_t)n * 2);
}
void t_dequant_8x8(i16 d[8][8], int mf[6][8][8], int qp) { dequant_call(&d[0][0], 64, &mf[qp % 6][0][0], 6, qp, 0); }
void t_dequant_4x4(i16 d[4][4], int mf[6][4][4], int qp) { dequant_call(&d[0][0], 16, &mf[qp % 6][0][0], 4, qp, 0); }
void t_dequant_4x4_dc(i16 d[4][4], int mf[6][4][4], int qp) { dequant_call(&d[0][0], 16, &mf[qp % 6][0][0], 6, qp, 1); }
void t_denoise(i16 *d, u32 *sum, u16 *offset, int size)
{
    Call c(OP_DENOISE);
    c.a.p[0] = size;
    c.a.off[0] = c.in(d, (size_t)size * 2);
    c.a.off[1] = c.in(sum, (size_t)size * 4);
    c.a.off[2] = c.in(offset, (size_t)size * 2);
    c.run();
    memcpy(d, c.at(c.a.off[0]), (size_t)size * 2);
    memcpy(sum, c.at(c.a.off[1]), (size_t)size * 4);
}
template <int N, int SKIP> int t_decimate(i16 *d)
{
    Call c(OP_DECIMATE);
    c.a.p[0] = N;
    c.a.off[0] = c.in(d + SKIP, (size_t)N * 2);
    c.a.off[9] = c.zero(16);
    c.run();
    return *(const int *)c.at(c.a.off[9]);
}
template <int N> int t_coeff_last(i16 *d)
{
    Call c(OP_COEFF_LAST);
    c.a.p[0] = N;
    c.a.off[0] = c.in(d, (size_t)N * 2);
    c.a.off[9] = c.zero(16);
    c.run();
    return *(const int *)c.at(c.a.off[9]);
}
template <int N> int t_level_run(i16 *d, x264hip_run_level_t *rl)
{
    Call c(OP_LEVEL_RUN);
    c.a.p[0] = N;
    c.a.off[0] = c.in(d, (size_t)N * 2);
    c.a.off[8] = c.in(rl, sizeof(*rl));
    c.a.off[9] = c.zero(16);
    c.run();
    memcpy(rl, c.at(c.a.off[8]), sizeof(*rl));
    return *(const int *)c.at(c.a.off[9]);
}

// ---- mc --------------------------------------------------------------------
void avg_call(u8 *dst, int sd, const u8 *s1, int i1, const u8 *s2, int i2, int w, int h, int wt)
{
    Call c(OP_AVG);
    c.a.p[0] = w; c.a.p[1] = h; c.a.p[2] = wt;
    c.a.off[0] = c.in2d(s1, i1, w, h);
    c.a.off[1] = c.in2d(s2, i2, w, h);
    c.a.off[9] = c.zero((size_t)w * h);
    c.run();
    c.out2d(c.a.off[9], dst, sd, w, h);
}
template <int I> void t_avg(u8 *dst, int sd, u8 *s1, int i1, u8 *s2, int i2, int wt)
{
    avg_call(dst, sd, s1, i1, s2, i2, kW[I], kH[I], wt);
}
const u8 h_qpel_a[16] = {0,1,1,1, 0,1,1,1, 2,3,3,3, 0,1,1,1};
const u8 h_qpel_b[16] = {0,0,0,0, 2,2,3,2, 2,2,3,2, 2,2,3,2};
// qpel addressing of the four half-pel planes (R/common/mc.c:160-202).  Which
// plane(s) and which offsets is pointer arithmetic and stays on the host; the
// rounded average is the arithmetic and goes to the GPU.
void t_mc_luma(u8 *dst, int sd, u8 **src, int ss, int mvx, int mvy, int w, int h)
{
    int fx = mvx & 3, fy = mvy & 3, idx = fy * 4 + fx;
    ptrdiff_t base = (ptrdiff_t)(mvy >> 2) * ss + (mvx >> 2);
    const u8 *a = src[h_qpel_a[idx]] + base + (fy == 3) * ss;
    if (idx & 5) {
        const u8 *b = src[h_qpel_b[idx]] + base + (fx == 3);
        avg_call(dst, sd, a, ss, b, ss, w, h, 32);
    } else
        for (int y = 0; y < h; y++) memcpy(dst + (ptrdiff_t)y * sd, a + (ptrdiff_t)y * ss, w);
}
u8 *t_get_ref(u8 *dst, int *sd, u8 **src, int ss, int mvx, int mvy, int w, int h)
{
    int fx = mvx & 3, fy = mvy & 3, idx = fy * 4 + fx;
    ptrdiff_t base = (ptrdiff_t)(mvy >> 2) * ss + (mvx >> 2);
    u8 *a = src[h_qpel_a[idx]] + base + (fy == 3) * ss;
    if (idx & 5) {
        const u8 *b = src[h_qpel_b[idx]] + base + (fx == 3);
        avg_call(dst, *sd, a, ss, b, ss, w, h, 32);
        return dst;
    }
    *sd = ss;
    return a;
}
void t_mc_chroma(u8 *dst, int sd, u8 *src, int ss, int mvx, int mvy, int w, int h)
{
    Call c(OP_MC_CHROMA);
    c.a.p[0] = w; c.a.p[1] = h; c.a.p[2] = mvx & 7; c.a.p[3] = mvy & 7;
    c.a.off[0] = c.in2d(src + (ptrdiff_t)(mvy >> 3) * ss + (mvx >> 3), ss, w + 1, h + 1);
    c.a.off[9] = c.zero((size_t)w * h);
    c.run();
    c.out2d(c.a.off[9], dst, sd, w, h);
}
template <int W> void t_copy(u8 *dst, int sd, u8 *src, int ss, int h)
{
    for (int y = 0; y < h; y++) memcpy(dst + (ptrdiff_t)y * sd, src + (ptrdiff_t)y * ss, W);
}
void t_plane_copy(u8 *dst, int sd, u8 *src, int ss, int w, int h)
{
    for (int y = 0; y < h; y++) memcpy(dst + (ptrdiff_t)y * sd, src + (ptrdiff_t)y * ss, w);
}
void t_hpel_filter(u8 *dh, u8 *dv, u8 *dc, u8 *src, int stride, int width, int height, i16 *buf)
{
    (void)buf;   // caller's scratch; the kernel keeps its own int16 row store
    Call c(OP_HPEL);
    int ls = width + 5;
    c.a.p[0] = width; c.a.p[1] = height;
    c.a.off[0] = c.in2d(src - 2 * (ptrdiff_t)stride - 2, stride, ls, height + 5);
    c.a.off[1] = c.zero((size_t)height * ls * 2);
    c.a.off[7] = c.zero((size_t)height * ls);
    c.a.off[8] = c.zero((size_t)height * width);
    c.a.off[9] = c.zero((size_t)height * width);
    c.run();
    c.out2d(c.a.off[7], dv - 2, stride, ls, height);      // dstv is written for x in [-2, width+3), mc.c:140-145
    c.out2d(c.a.off[8], dh, stride, width, height);
    c.out2d(c.a.off[9], dc, stride, width, height);
}
template <int N> void t_integral_h(u16 *sum, u8 *pix, int stride)
{
    Call c(OP_INTEGRAL_H);
    int cnt = stride - N;
    c.a.p[0] = N; c.a.p[1] = cnt;
    c.a.off[0] = c.in(pix, stride);
    c.a.off[1] = c.in(sum - stride, (size_t)cnt * 2);
    c.a.off[9] = c.zero((size_t)cnt * 2);
    c.run();
    memcpy(sum, c.at(c.a.off[9]), (size_t)cnt * 2);
}
void t_integral_4v(u16 *sum8, u16 *sum4, int stride)
{
    Call c(OP_INTEGRAL_4V);
    int cnt = stride - 8;
    c.a.p[0] = cnt;
    c.a.off[0] = c.in(sum8, (size_t)(cnt + 4) * 2);
    c.a.off[1] = c.in(sum8 + 4 * stride, (size_t)cnt * 2);
    c.a.off[2] = c.in(sum8 + 8 * stride, (size_t)(cnt + 4) * 2);
    c.a.off[8] = c.zero((size_t)cnt * 2);
    c.a.off[9] = c.zero((size_t)cnt * 2);
    c.run();
    memcpy(sum4, c.at(c.a.off[8]), (size_t)cnt * 2);
    memcpy(sum8, c.at(c.a.off[9]), (size_t)cnt * 2);
}
void t_integral_8v(u16 *sum8, int stride)
{
    Call c(OP_INTEGRAL_8V);
    int cnt = stride - 8;
    c.a.p[0] = cnt;
    c.a.off[0] = c.in(sum8, (size_t)cnt * 2);
    c.a.off[1] = c.in(sum8 + 8 * stride, (size_t)cnt * 2);
    c.a.off[9] = c.zero((size_t)cnt * 2);
    c.run();
    memcpy(sum8, c.at(c.a.off[9]), (size_t)cnt * 2);
}
void t_lowres(u8 *src, u8 *d0, u8 *dh, u8 *dv, u8 *dc, int ss, int ds, int w, int h)
{
    Call c(OP_LOWRES);
    c.a.p[0] = w; c.a.p[1] = h;
    c.a.off[0] = c.in2d(src, ss, 2 * w + 1, 2 * h + 1);
    for (int k = 6; k < 10; k++) c.a.off[k] = c.zero((size_t)w * h);
    c.run();
    c.out2d(c.a.off[6], d0, ds, w, h); c.out2d(c.a.off[7], dh, ds, w, h);
    c.out2d(c.a.off[8], dv, ds, w, h); c.out2d(c.a.off[9], dc, ds, w, h);
}
void t_prefetch_fenc(u8 *, int, u8 *, int, int) {}
void t_prefetch_ref(u8 *, int, int) {}
void t_memzero(void *d, int n) { memset(d, 0, n); }

// ---- predict ---------------------------------------------------------------
// Neighbour needs per table slot: bit0 left, bit1 top, bit2 top-right, bit3 top-left.
const u8 need16[7] = {2, 1, 3, 11, 1, 2, 0};
const u8 need8c[7] = {3, 1, 2, 11, 1, 2, 0};
const u8 need4[12] = {2, 1, 3, 6, 11, 11, 11, 6, 1, 1, 2, 0};
void pred_call(int fam, int mode, u8 *src, int need)
{
    const int n = fam == 0 ? 16 : fam == 1 ? 8 : 4;
    const int ls = 2 * n + 1;
    Call c(OP_PRED);
    c.a.p[0] = fam; c.a.p[1] = mode; c.a.p[2] = ls;
    c.a.off[0] = c.zero((size_t)(n + 1) * ls);
    u8 *loc = c.at(c.a.off[0]) + ls + 1;
    if (need & 2) memcpy(loc - ls, src - FDEC, n);
    if (need & 4) memcpy(loc - ls + n, src - FDEC + n, n);
    if (need & 8) loc[-ls - 1] = src[-FDEC - 1];
    if (need & 1) for (int y = 0; y < n; y++) loc[y * ls - 1] = src[y * FDEC - 1];
    c.a.off[9] = c.zero((size_t)n * n);
    c.run();
    c.out2d(c.a.off[9], src, FDEC, n, n);
}
template <int M> void t_pred16(u8 *s) { pred_call(0, M, s, need16[M]); }
template <int M> void t_pred8c(u8 *s) { pred_call(1, M, s, need8c[M]); }
template <int M> void t_pred4(u8 *s)  { pred_call(2, M, s, need4[M]); }
template <int M> void t_pred8(u8 *src, u8 edge[33])
{
    Call c(OP_PRED8);
    c.a.p[0] = M;
    c.a.off[0] = c.in(edge, 33);
    c.a.off[9] = c.zero(64);
    c.run();
    c.out2d(c.a.off[9], src, FDEC, 8, 8);
}
void t_pred8_filter(u8 *src, u8 edge[33], int i_neighbor, int i_filters)
{
    Call c(OP_PRED8_FILTER);
    const int ls = 17;
    c.a.p[0] = i_neighbor; c.a.p[1] = i_filters;
    c.a.off[0] = c.zero(9 * ls);
    u8 *loc = c.at(c.a.off[0]) + ls + 1;
    // the reference reads exactly these neighbours for the given flags (predict.c:499-540)
    if (i_filters & (X264HIP_MB_LEFT | X264HIP_MB_TOP)) {
        loc[-ls - 1] = src[-FDEC - 1];
        memcpy(loc - ls, src - FDEC, 8);
    }
    if (i_filters & X264HIP_MB_LEFT)
        for (int y = 0; y < 8; y++) loc[y * ls - 1] = src[y * FDEC - 1];
    if ((i_filters & X264HIP_MB_TOP) && (i_neighbor & X264HIP_MB_TOPRIGHT))
        memcpy(loc - ls + 8, src - FDEC + 8, 8);
    c.a.off[9] = c.in(edge, 33);
    c.run();
    memcpy(edge, c.at(c.a.off[9]), 33);
}

// ---- deblock -----------------------------------------------------------------
// kind: 0 luma 1 chroma 2 luma-intra 3 chroma-intra; dir: 0 = v (edge between
// rows, lines run along x), 1 = h (edge between columns, lines run along y).
void deblock_call(int kind, int dir, u8 *pix, int stride, int alpha, int beta, const int8_t *tc0)
{
    const int luma = kind == 0 || kind == 2;
    const int nlines = luma ? 16 : 8, half = luma ? 4 : 2;
    Call c(OP_DEBLOCK);
    c.a.p[0] = kind; c.a.p[1] = nlines; c.a.p[2] = alpha; c.a.p[3] = beta;
    if (tc0) for (int i = 0; i < 4; i++) c.a.p[4 + i] = tc0[i];
    c.a.off[0] = c.zero((size_t)nlines * 8);
    u8 *lines = c.at(c.a.off[0]);
    const ptrdiff_t xs = dir == 0 ? stride : 1, ys = dir == 0 ? 1 : stride;
    for (int l = 0; l < nlines; l++)
        for (int k = -half; k < half; k++) lines[l * 8 + 4 + k] = pix[l * ys + k * xs];
    c.run();
    for (int l = 0; l < nlines; l++)
        for (int k = -half; k < half; k++) pix[l * ys + k * xs] = lines[l * 8 + 4 + k];
}
void t_db_v_luma(u8 *p, int s, int a, int b, int8_t *t)   { deblock_call(0, 0, p, s, a, b, t); }
void t_db_h_luma(u8 *p, int s, int a, int b, int8_t *t)   { deblock_call(0, 1, p, s, a, b, t); }
void t_db_v_chroma(u8 *p, int s, int a, int b, int8_t *t) { deblock_call(1, 0, p, s, a, b, t); }
void t_db_h_chroma(u8 *p, int s, int a, int b, int8_t *t) { deblock_call(1, 1, p, s, a, b, t); }
void t_db_v_luma_i(u8 *p, int s, int a, int b)   { deblock_call(2, 0, p, s, a, b, nullptr); }
void t_db_h_luma_i(u8 *p, int s, int a, int b)   { deblock_call(2, 1, p, s, a, b, nullptr); }
void t_db_v_chroma_i(u8 *p, int s, int a, int b) { deblock_call(3, 0, p, s, a, b, nullptr); }
void t_db_h_chroma_i(u8 *p, int s, int a, int b) { deblock_call(3, 1, p, s, a, b, nullptr); }

}  // namespace

// ------------------------------------------------------------- table fillers
#define FILL7(dst, FN, KIND) do { dst[0] = FN<KIND,16,16>; dst[1] = FN<KIND,16,8>; dst[2] = FN<KIND,8,16>; \
    dst[3] = FN<KIND,8,8>; dst[4] = FN<KIND,8,4>; dst[5] = FN<KIND,4,8>; dst[6] = FN<KIND,4,4>; } while (0)

extern "C" int x264_pixel_init_hip(x264hip_pixel_function_t *pf)
{
    if (!initialised()) { set_error("x264_pixel_init_hip: call x264hip_init first"); return -1; }
    memset(pf, 0, sizeof(*pf));
    FILL7(pf->sad, t_cmp, CMP_SAD); FILL7(pf->sad_aligned, t_cmp, CMP_SAD);
    FILL7(pf->ssd, t_cmp, CMP_SSD); FILL7(pf->satd, t_cmp, CMP_SATD);
    FILL7(pf->sad_x3, t_cmp_x3, CMP_SAD); FILL7(pf->sad_x4, t_cmp_x4, CMP_SAD);
    FILL7(pf->satd_x3, t_cmp_x3, CMP_SATD); FILL7(pf->satd_x4, t_cmp_x4, CMP_SATD);
    pf->sa8d[X264HIP_PIXEL_16x16] = t_cmp<CMP_SA8D, 16, 16>;
    pf->sa8d[X264HIP_PIXEL_8x8] = t_cmp<CMP_SA8D, 8, 8>;
    pf->var[X264HIP_PIXEL_16x16] = t_var<16, 8>;
    pf->var[X264HIP_PIXEL_8x8] = t_var<8, 6>;
    pf->hadamard_ac[0] = t_hac<16, 16>; pf->hadamard_ac[1] = t_hac<16, 8>;
    pf->hadamard_ac[2] = t_hac<8, 16>;  pf->hadamard_ac[3] = t_hac<8, 8>;
    pf->ssim_4x4x2_core = t_ssim_core;
    pf->ssim_end4 = t_ssim_end4;
    pf->ads[X264HIP_PIXEL_16x16] = t_ads<4>; pf->ads[X264HIP_PIXEL_16x8] = t_ads<2>; pf->ads[X264HIP_PIXEL_8x8] = t_ads<1>;
    return 0;
}
extern "C" int x264_dct_init_hip(x264hip_dct_function_t *f)
{
    if (!initialised()) { set_error("x264_dct_init_hip: call x264hip_init first"); return -1; }
    typedef void (*sub_t)(i16 *, u8 *, u8 *);
    typedef void (*add_t)(u8 *, i16 *);
#define S(fn) ((sub_t)fn)
#define A(fn) ((add_t)fn)
    f->sub4x4_dct = (decltype(f->sub4x4_dct))S((t_sub_dct<4, 0>));
    f->add4x4_idct = (decltype(f->add4x4_idct))A((t_add_idct<4, 0>));
    f->sub8x8_dct = (decltype(f->sub8x8_dct))S((t_sub_dct<8, 0>));
    f->add8x8_idct = (decltype(f->add8x8_idct))A((t_add_idct<8, 0>));
    f->add8x8_idct_dc = (decltype(f->add8x8_idct_dc))A((t_add_dc<8>));
    f->sub16x16_dct = (decltype(f->sub16x16_dct))S((t_sub_dct<16, 0>));
    f->add16x16_idct = (decltype(f->add16x16_idct))A((t_add_idct<16, 0>));
    f->add16x16_idct_dc = (decltype(f->add16x16_idct_dc))A((t_add_dc<16>));
    f->sub8x8_dct8 = (decltype(f->sub8x8_dct8))S((t_sub_dct<8, 1>));
    f->add8x8_idct8 = (decltype(f->add8x8_idct8))A((t_add_idct<8, 1>));
    f->sub16x16_dct8 = (decltype(f->sub16x16_dct8))S((t_sub_dct<16, 1>));
    f->add16x16_idct8 = (decltype(f->add16x16_idct8))A((t_add_idct<16, 1>));
    f->dct4x4dc = (decltype(f->dct4x4dc))((void (*)(i16 *))t_dc4x4<1>);
    f->idct4x4dc = (decltype(f->idct4x4dc))((void (*)(i16 *))t_dc4x4<0>);
#undef S
#undef A
    return 0;
}
extern "C" int x264_zigzag_init_hip(x264hip_zigzag_function_t *f, int b_interlaced)
{
    if (!initialised()) { set_error("x264_zigzag_init_hip: call x264hip_init first"); return -1; }
    typedef void (*scan_t)(i16 *, i16 *);
    if (b_interlaced) {
        f->scan_8x8 = (decltype(f->scan_8x8))((scan_t)t_scan<64, 1>);
        f->scan_4x4 = (decltype(f->scan_4x4))((scan_t)t_scan<16, 1>);
        f->sub_8x8 = t_zsub<8, 1>; f->sub_4x4 = t_zsub<4, 1>;
    } else {
        f->scan_8x8 = (decltype(f->scan_8x8))((scan_t)t_scan<64, 0>);
        f->scan_4x4 = (decltype(f->scan_4x4))((scan_t)t_scan<16, 0>);
        f->sub_8x8 = t_zsub<8, 0>; f->sub_4x4 = t_zsub<4, 0>;
    }
    f->interleave_8x8_cavlc = t_interleave;
    return 0;
}
extern "C" int x264_quant_init_hip(x264hip_quant_function_t *f)
{
    if (!initialised()) { set_error("x264_quant_init_hip: call x264hip_init first"); return -1; }
    f->quant_8x8 = t_quant_8x8; f->quant_4x4 = t_quant_4x4;
    f->quant_4x4_dc = t_quant_4x4_dc; f->quant_2x2_dc = t_quant_2x2_dc;
    f->dequant_8x8 = t_dequant_8x8; f->dequant_4x4 = t_dequant_4x4; f->dequant_4x4_dc = t_dequant_4x4_dc;
    f->denoise_dct = t_denoise;
    f->decimate_score15 = t_decimate<15, 1>; f->decimate_score16 = t_decimate<16, 0>; f->decimate_score64 = t_decimate<64, 0>;
    f->coeff_last[X264HIP_DCT_CHROMA_DC] = t_coeff_last<4>;  f->coeff_last[X264HIP_DCT_LUMA_AC] = t_coeff_last<15>;
    f->coeff_last[X264HIP_DCT_LUMA_4x4] = t_coeff_last<16>;  f->coeff_last[X264HIP_DCT_LUMA_8x8] = t_coeff_last<64>;
    f->coeff_last[X264HIP_DCT_LUMA_DC] = t_coeff_last<16>;   f->coeff_last[X264HIP_DCT_CHROMA_AC] = t_coeff_last<15>;
    f->coeff_level_run[X264HIP_DCT_CHROMA_DC] = t_level_run<4>;
    f->coeff_level_run[X264HIP_DCT_LUMA_AC] = t_level_run<15>;
    f->coeff_level_run[X264HIP_DCT_LUMA_4x4] = t_level_run<16>;
    f->coeff_level_run[X264HIP_DCT_LUMA_DC] = t_level_run<16>;
    f->coeff_level_run[X264HIP_DCT_CHROMA_AC] = t_level_run<15>;
    return 0;
}
extern "C" int x264_mc_init_hip(x264hip_mc_functions_t *f)
{
    if (!initialised()) { set_error("x264_mc_init_hip: call x264hip_init first"); return -1; }
    memset(f, 0, sizeof(*f));
    f->mc_luma = t_mc_luma; f->get_ref = t_get_ref; f->mc_chroma = t_mc_chroma;
    f->avg[0] = t_avg<0>; f->avg[1] = t_avg<1>; f->avg[2] = t_avg<2>; f->avg[3] = t_avg<3>; f->avg[4] = t_avg<4>;
    f->avg[5] = t_avg<5>; f->avg[6] = t_avg<6>; f->avg[7] = t_avg<7>; f->avg[8] = t_avg<8>; f->avg[9] = t_avg<9>;
    f->copy_16x16_unaligned = t_copy<16>;
    f->copy[X264HIP_PIXEL_16x16] = t_copy<16>; f->copy[X264HIP_PIXEL_8x8] = t_copy<8>; f->copy[X264HIP_PIXEL_4x4] = t_copy<4>;
    f->plane_copy = t_plane_copy;
    f->hpel_filter = t_hpel_filter;
    f->prefetch_fenc = t_prefetch_fenc; f->prefetch_ref = t_prefetch_ref;
    f->memcpy_aligned = memcpy; f->memzero_aligned = t_memzero;
    f->integral_init4h = t_integral_h<4>; f->integral_init8h = t_integral_h<8>;
    f->integral_init4v = t_integral_4v; f->integral_init8v = t_integral_8v;
    f->frame_init_lowres_core = t_lowres;
    return 0;
}
extern "C" int x264_predict_16x16_init_hip(x264hip_predict_t pf[7])
{
    if (!initialised()) { set_error("x264_predict_16x16_init_hip: call x264hip_init first"); return -1; }
    pf[0] = t_pred16<0>; pf[1] = t_pred16<1>; pf[2] = t_pred16<2>; pf[3] = t_pred16<3>;
    pf[4] = t_pred16<4>; pf[5] = t_pred16<5>; pf[6] = t_pred16<6>;
    return 0;
}
extern "C" int x264_predict_8x8c_init_hip(x264hip_predict_t pf[7])
{
    if (!initialised()) { set_error("x264_predict_8x8c_init_hip: call x264hip_init first"); return -1; }
    pf[0] = t_pred8c<0>; pf[1] = t_pred8c<1>; pf[2] = t_pred8c<2>; pf[3] = t_pred8c<3>;
    pf[4] = t_pred8c<4>; pf[5] = t_pred8c<5>; pf[6] = t_pred8c<6>;
    return 0;
}
extern "C" int x264_predict_4x4_init_hip(x264hip_predict_t pf[12])
{
    if (!initialised()) { set_error("x264_predict_4x4_init_hip: call x264hip_init first"); return -1; }
    pf[0] = t_pred4<0>; pf[1] = t_pred4<1>; pf[2] = t_pred4<2>; pf[3] = t_pred4<3>; pf[4] = t_pred4<4>; pf[5] = t_pred4<5>;
    pf[6] = t_pred4<6>; pf[7] = t_pred4<7>; pf[8] = t_pred4<8>; pf[9] = t_pred4<9>; pf[10] = t_pred4<10>; pf[11] = t_pred4<11>;
    return 0;
}
extern "C" int x264_predict_8x8_init_hip(x264hip_predict8x8_t pf[12], x264hip_predict_8x8_filter_t *filter)
{
    if (!initialised()) { set_error("x264_predict_8x8_init_hip: call x264hip_init first"); return -1; }
    pf[0] = t_pred8<0>; pf[1] = t_pred8<1>; pf[2] = t_pred8<2>; pf[3] = t_pred8<3>; pf[4] = t_pred8<4>; pf[5] = t_pred8<5>;
    pf[6] = t_pred8<6>; pf[7] = t_pred8<7>; pf[8] = t_pred8<8>; pf[9] = t_pred8<9>; pf[10] = t_pred8<10>; pf[11] = t_pred8<11>;
    *filter = t_pred8_filter;
    return 0;
}
extern "C" int x264_deblock_init_hip(x264hip_deblock_function_t *f)
{
    if (!initialised()) { set_error("x264_deblock_init_hip: call x264hip_init first"); return -1; }
    f->deblock_v_luma = t_db_v_luma; f->deblock_h_luma = t_db_h_luma;
    f->deblock_v_chroma = t_db_v_chroma; f->deblock_h_chroma = t_db_h_chroma;
    f->deblock_v_luma_intra = t_db_v_luma_i; f->deblock_h_luma_intra = t_db_h_luma_i;
    f->deblock_v_chroma_intra = t_db_v_chroma_i; f->deblock_h_chroma_intra = t_db_h_chroma_i;
    return 0;
}
