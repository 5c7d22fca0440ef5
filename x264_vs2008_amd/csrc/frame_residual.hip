// frame_residual.hip -- FRAME LEVEL, part 3: the inter residual pipeline of
// every macroblock of a frame in one launch, one wavefront per macroblock:
//
//   x264_mb_mc (16x16: mc_luma through the qpel blend of the four half-pel
//   planes + bilinear 1/8-pel mc_chroma)  ->  sub16x16_dct / sub16x16_dct8
//   -> quant -> zigzag scan -> decimate (JVT-B118 scores: 8x8 < 4, MB < 6,
//   chroma < 7) -> dequant -> inverse transform added to the prediction
//   -> reconstruction planes, scanned levels, cbp, nnz flags.
//
// Restates x264_macroblock_encode's inter branch (R/encoder/macroblock.c:
// 596-768) and x264_mb_encode_8x8_chroma (:272-363) with b_dct_decimate on
// and trellis / noise reduction / lossless off.  Block-level steps are the
// table entries of device_prims.h; the decision logic (which blocks survive
// decimation) is a few dozen scalar operations done by one lane per
// macroblock from scores the other lanes left in LDS.
//
// Lane roles inside a wavefront: 64 lanes = 256 luma prediction pixels / 4;
// lanes 0-15 own one luma 4x4 block each (lanes 0-3 one 8x8 block each in
// 8x8-transform mode), lanes 0-7 own the chroma 4x4 blocks (U: 0-3, V: 4-7).
#include "device_prims.h"
#include "frame_internal.h"

using namespace x264hip;

#define RS_WAVES 4

struct ResGeom {
    int mb_w, mb_h, sy, sc, qp, qpc, field;
    size_t bs_y, bs_c;  // bytes between batch elements
    int mv_per_mb;      // 1: one 16x16 vector per macroblock; 16: one per 4x4 block (raster), any P partition
};

// reference pictures a macroblock may point at (list0), passed by value
#define RS_MAX_REFS 8
struct RefTab {
    const u8 *y[RS_MAX_REFS][4];   // full, H, V, HV planes
    const u8 *u[RS_MAX_REFS], *v[RS_MAX_REFS];
};

struct ResLds {
    u8  fe[384];          // source: Y 16x16, U 8x8, V 8x8
    u8  pr[384];          // prediction, then reconstruction
    i16 coef[16][16];     // dequantised luma coefficients (4x4 mode: [blk][16]; 8x8 mode: [4][64] flat)
    i16 t8[256];          // 8x8 mode: intermediate of the forward transform between its two passes
    i16 ccoef[8][16];     // dequantised chroma AC (+ DC once decided)
    int score[16];        // luma decimate scores / nz flags packed: score | nz << 8
    int cscore[8];
    i16 cdc[8];           // chroma DCs before the 2x2 transform
    int keep8;            // luma cbp after decimation
    int cmode[2];         // chroma per channel: 0 pred only, 1 DC only, 2 full
    i16 cdcout[8];        // chroma per-block DC to add (mode 1) or to put at coef[0] (mode 2)
};

__device__ __forceinline__ void blk_xy(int k, int &x, int &y)
{
    x = ((k >> 2) & 1) * 8 + (k & 1) * 4;
    y = (k >> 3) * 8 + ((k >> 1) & 1) * 4;
}
__device__ __forceinline__ int decimate_scan(const i16 *lv, int n, const u8 *tab)
{   // R/common/quant.c:213-239 on already scanned levels lv[0..n)
    int i = n - 1, score = 0;
    while (i >= 0 && lv[i] == 0) i--;
    while (i >= 0) {
        if ((unsigned)(lv[i--] + 1) > 2u) return 9;
        int run = 0;
        while (i >= 0 && lv[i] == 0) { i--; run++; }
        score += tab[run];
    }
    return score;
}

// Motion-compensated prediction of one macroblock into s.pr (Y 16x16, U 8x8, V 8x8).
// Lane = 4 luma pixels of one row (all inside one 4x4 block) + one U and one V pixel.
// mc_luma: R/common/mc.c:160-179; mc_chroma: mc.c:205-236.
__device__ __forceinline__ void predict_mb(ResLds &s, const RefTab &refs, const ResGeom &g, const i16 *mv, const signed char *ref8,
                                           int mb, int lane, ptrdiff_t oy, ptrdiff_t oc, size_t bz)
{
    {
        int r = lane >> 2, x = (lane & 3) * 4;
        int blk = (r >> 2) * 4 + (x >> 2);
        const i16 *m = g.mv_per_mb == 1 ? mv + 2 * (size_t)mb : mv + ((size_t)mb * 16 + blk) * 2;
        int mvx = m[0], mvy = m[1];
        int ri = ref8 ? ref8[(size_t)mb * 4 + (r >> 3) * 2 + (x >> 3)] : 0;
        int qx = mvx & 3, qy = mvy & 3, idx = qy * 4 + qx;
        ptrdiff_t base = oy + (ptrdiff_t)((mvy >> 2) + r) * g.sy + (mvx >> 2) + x + (ptrdiff_t)(g.bs_y * bz);
        const u8 *pa = refs.y[ri][c_qpel_a[idx]] + base + (qy == 3) * g.sy;
        const u8 *pb = refs.y[ri][c_qpel_b[idx]] + base + (qx == 3);
#pragma unroll
        for (int i = 0; i < 4; i++)
            s.pr[r * 16 + x + i] = (idx & 5) ? (u8)(((int)pa[i] + (int)pb[i] + 1) >> 1) : pa[i];
    }
    {
        int cx = lane & 7, cy = lane >> 3;
        int blk = (cy >> 1) * 4 + (cx >> 1);
        const i16 *m = g.mv_per_mb == 1 ? mv + 2 * (size_t)mb : mv + ((size_t)mb * 16 + blk) * 2;
        int mvx = m[0], mvy = m[1];
        int ri = ref8 ? ref8[(size_t)mb * 4 + (cy >> 2) * 2 + (cx >> 2)] : 0;
        int dx = mvx & 7, dyy = mvy & 7;
        int ca = (8 - dx) * (8 - dyy), cb = dx * (8 - dyy), cc = (8 - dx) * dyy, cd = dx * dyy;
        ptrdiff_t cbase = oc + (ptrdiff_t)((mvy >> 3) + cy) * g.sc + (mvx >> 3) + cx + (ptrdiff_t)(g.bs_c * bz);
        const u8 *pu = refs.u[ri] + cbase, *pv = refs.v[ri] + cbase;
        s.pr[256 + lane] = (u8)((ca * pu[0] + cb * pu[1] + cc * pu[g.sc] + cd * pu[g.sc + 1] + 32) >> 6);
        s.pr[320 + lane] = (u8)((ca * pv[0] + cb * pv[1] + cc * pv[g.sc] + cd * pv[g.sc + 1] + 32) >> 6);
    }
}

template <int DCT8>
__global__ __launch_bounds__(64 * RS_WAVES) void k_inter_residual(
    const u8 *__restrict__ fy, const u8 *__restrict__ fu, const u8 *__restrict__ fv, RefTab refs,
    u8 *__restrict__ dy, u8 *__restrict__ du, u8 *__restrict__ dv, ResGeom g,
    const u16 *__restrict__ q4mf, const u16 *__restrict__ q4bias, const u16 *__restrict__ q8mf, const u16 *__restrict__ q8bias,
    const int *__restrict__ dq4, const int *__restrict__ dq8, const i16 *__restrict__ mv, const signed char *__restrict__ ref8,
    i16 *__restrict__ levels_y, i16 *__restrict__ levels_c, i16 *__restrict__ dc_c, int *__restrict__ cbp_out, u8 *__restrict__ nnz_out,
    i16 *__restrict__ mv4x4_out, signed char *__restrict__ ref_out)
{
    __shared__ ResLds s_all[RS_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mb = xcd_band_order(blockIdx.x, gridDim.x) * RS_WAVES + wave;
    if (mb >= g.mb_w * g.mb_h) return;
    const size_t bz = blockIdx.y;
    {   // batch element
        const size_t nmb = (size_t)g.mb_w * g.mb_h;
        fy += g.bs_y * bz; dy += g.bs_y * bz;
        fu += g.bs_c * bz; fv += g.bs_c * bz; du += g.bs_c * bz; dv += g.bs_c * bz;
        mv += 2 * g.mv_per_mb * nmb * bz; levels_y += 256 * nmb * bz; levels_c += 128 * nmb * bz; dc_c += 8 * nmb * bz;
        cbp_out += nmb * bz; nnz_out += 26 * nmb * bz;
        if (ref8) ref8 += 4 * nmb * bz;
        if (mv4x4_out) mv4x4_out += 32 * nmb * bz;
        if (ref_out) ref_out += 4 * nmb * bz;
    }
    ResLds &s = s_all[wave];
    const int mbx = mb % g.mb_w, mby = mb / g.mb_w;
    const ptrdiff_t oy = (ptrdiff_t)16 * mby * g.sy + 16 * mbx, oc = (ptrdiff_t)8 * mby * g.sc + 8 * mbx;
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); \
                         __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

    // ---- source + prediction into LDS ----
    // x264_mb_mc (R/common/macroblock.c:462-546): every partition is mc_luma / mc_chroma with its own
    // vector and reference; per pixel that depends only on the 4x4 block the pixel lies in.
    predict_mb(s, refs, g, mv, ref8, mb, lane, oy, oc, bz);
    {
        int r = lane >> 2, x = (lane & 3) * 4, cx = lane & 7, cy = lane >> 3;
        *(u32 *)(s.fe + r * 16 + x) = *(const u32 *)(fy + oy + (ptrdiff_t)r * g.sy + x);
        s.fe[256 + lane] = fu[oc + (ptrdiff_t)cy * g.sc + cx];
        s.fe[320 + lane] = fv[oc + (ptrdiff_t)cy * g.sc + cx];
    }
    WAVE_SYNC();

    i16 *ly = levels_y + (size_t)mb * 256;
    // ---- luma transform + quant + scan + dequant + scores ----
    if (!DCT8) {
        if (lane < 16) {
            int bx, by, r[16];
            blk_xy(lane, bx, by);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    r[4 * j + i] = (int)s.fe[(by + j) * 16 + bx + i] - (int)s.pr[(by + j) * 16 + bx + i];
            i16 c[16];
            fwd4x4(c, r);
            const u16 *mf = q4mf + (1 * 52 + g.qp) * 16, *bs = q4bias + (1 * 52 + g.qp) * 16;
            const int *dq = dq4 + 1 * 96 + (g.qp % 6) * 16;
            int nz = 0, bits = g.qp / 6 - 4;
            i16 lv[16];
#pragma unroll
            for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }
#pragma unroll
            for (int i = 0; i < 16; i++) lv[i] = nz ? c[c_scan4[g.field][i]] : (i16)0;
#pragma unroll
            for (int i = 0; i < 16; i++) { ly[16 * lane + i] = lv[i]; s.coef[lane][i] = (i16)dequant_one(c[i], dq[i], bits); }
            s.score[lane] = (nz ? decimate_scan(lv, 16, c_decimate4) : 0) | ((nz != 0) << 8);
        }
    } else {
        // 8x8 transform spread over the wavefront: lane = (block b, column / row k) for the two
        // 1-D passes (32 lanes), then all 64 lanes quantise / scan 4 coefficients each.
        i16 *tmp = s.t8;                               // [4][64] intermediate, int16 as in the reference (dct.c:266-276)
        const int b = lane >> 3, k8 = lane & 7;
        if (lane < 32) {
            const u8 *p1 = s.fe + (b >> 1) * 8 * 16 + (b & 1) * 8 + k8, *p2 = s.pr + (b >> 1) * 8 * 16 + (b & 1) * 8 + k8;
            int a[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = (int)p1[k * 16] - (int)p2[k * 16];
            fwd8_1d(o, a);                             // column k8
#pragma unroll
            for (int k = 0; k < 8; k++) tmp[64 * b + k * 8 + k8] = (i16)o[k];
        }
        WAVE_SYNC();
        i16 *coef = &s.coef[0][0];
        if (lane < 32) {
            int a[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = tmp[64 * b + k8 * 8 + k];
            fwd8_1d(o, a);                             // row k8, stored transposed (dct.c:278-283)
#pragma unroll
            for (int k = 0; k < 8; k++) coef[64 * b + k * 8 + k8] = (i16)o[k];
        }
        WAVE_SYNC();
        // quantise: lane handles coefficients lane, lane+64, ... i.e. block j, position lane
        const u16 *mf = q8mf + (1 * 52 + g.qp) * 64, *bs = q8bias + (1 * 52 + g.qp) * 64;
        const int mfl = mf[lane], bsl = bs[lane];
        unsigned long long nzmask[4], bigmask[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int q = quant_one(coef[64 * j + lane], mfl, bsl);
            coef[64 * j + lane] = (i16)q;
            nzmask[j] = __ballot(q != 0);
        }
        WAVE_SYNC();
        // scan order: level i of block j = coef[scan[i]]; masks of the scanned levels for the score
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int lvv = nzmask[j] ? (int)coef[64 * j + c_scan8[g.field][lane]] : 0;
            ly[64 * j + lane] = (i16)lvv;
            nzmask[j] = __ballot(lvv != 0);
            bigmask[j] = __ballot((unsigned)(lvv + 1) > 2u);
        }
        if (lane < 4) {
            // decimate_score64 from the bit mask of non-zero scanned levels (quant.c:213-239)
            unsigned long long m = lane == 0 ? nzmask[0] : lane == 1 ? nzmask[1] : lane == 2 ? nzmask[2] : nzmask[3];
            unsigned long long bg = lane == 0 ? bigmask[0] : lane == 1 ? bigmask[1] : lane == 2 ? bigmask[2] : bigmask[3];
            int sc = 0;
            if (bg) sc = 9;
            else {
                int idx = m ? 63 - __clzll(m) : -1;
                while (idx >= 0) {
                    unsigned long long below = idx ? (m & ((1ull << idx) - 1)) : 0ull;
                    int prev = below ? 63 - __clzll(below) : -1;
                    sc += c_decimate8[idx - prev - 1];
                    idx = prev;
                }
            }
            s.score[lane] = sc | ((m != 0) << 8);
        }
    }
    WAVE_SYNC();

    // ---- decimation decisions (macroblock.c:627-742), one lane ----
    u8 *nnz = nnz_out + (size_t)mb * 26;
    if (lane == 0) {
        int cbp = 0, dec_mb = 0;
        u8 nz16[16];
        if (!DCT8) {
            for (int i8 = 0; i8 < 4; i8++) {
                int dec8 = 0;
                for (int i4 = 0; i4 < 4; i4++) {
                    int v = s.score[4 * i8 + i4];
                    nz16[4 * i8 + i4] = (u8)(v >> 8);
                    if ((v >> 8) && dec8 < 6) dec8 += v & 255;
                }
                dec_mb += dec8;
                if (dec8 < 4) nz16[4 * i8] = nz16[4 * i8 + 1] = nz16[4 * i8 + 2] = nz16[4 * i8 + 3] = 0;
                else cbp |= 1 << i8;
            }
            if (dec_mb < 6) { cbp = 0; for (int i = 0; i < 16; i++) nz16[i] = 0; }
        } else {
            for (int i = 0; i < 4; i++) {
                int v = s.score[i];
                if (v >> 8) { dec_mb += v & 255; if ((v & 255) >= 4) cbp |= 1 << i; }
            }
            if (dec_mb < 6) cbp = 0;
            for (int i = 0; i < 16; i++) nz16[i] = (u8)((cbp >> (i >> 2)) & 1);
        }
        s.keep8 = cbp;
        for (int i = 0; i < 16; i++) nnz[i] = nz16[i];
    }
    WAVE_SYNC();

    // ---- luma reconstruction ----
    if (!DCT8) {
        if (lane < 16 && ((s.keep8 >> (lane >> 2)) & 1)) {
            int bx, by, res[16];
            blk_xy(lane, bx, by);
            inv4x4(res, s.coef[lane]);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    u8 *p = s.pr + (by + j) * 16 + bx + i;
                    *p = (u8)clip_u8((int)*p + res[4 * j + i]);
                }
        }
    } else {
        // dequant (64 lanes x 4 coefficients), then the two inverse 1-D passes on 32 lanes;
        // blocks dropped by decimation skip the arithmetic but every lane reaches the syncs
        i16 *coef = &s.coef[0][0];
        const int keep = s.keep8, b = lane >> 3, k8 = lane & 7;
        const int *dq = dq8 + 1 * 384 + (g.qp % 6) * 64;
        const int bits = g.qp / 6 - 6, dql = dq[lane];
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((keep >> j) & 1) {
                int v = dequant_one(coef[64 * j + lane], dql, bits);
                if (lane == 0) v = (int)(i16)(v + 32);           // rounding term, dct.c:326
                coef[64 * j + lane] = (i16)v;
            }
        WAVE_SYNC();
        if (lane < 32 && ((keep >> b) & 1)) {
            int a[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = coef[64 * b + k * 8 + k8];
            inv8_1d(o, a);                             // column k8, in place, narrowed to int16 (dct.c:328-333)
#pragma unroll
            for (int k = 0; k < 8; k++) coef[64 * b + k * 8 + k8] = (i16)o[k];
        }
        WAVE_SYNC();
        if (lane < 32 && ((keep >> b) & 1)) {
            u8 *dst = s.pr + (b >> 1) * 8 * 16 + (b & 1) * 8;
            int a[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = coef[64 * b + k8 * 8 + k];
            inv8_1d(o, a);                             // row k8 -> picture column k8 (dct.c:335-340)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                u8 *p = dst + k8 + k * 16;
                *p = (u8)clip_u8((int)*p + (o[k] >> 6));
            }
        }
    }

    // ---- chroma blocks: transform, AC quant, scan, dequant, scores ----
    i16 *lc = levels_c + (size_t)mb * 128;
    if (lane < 8) {
        int ch = lane >> 2, i4 = lane & 3, bx = (i4 & 1) * 4, by = (i4 >> 1) * 4, r[16];
        const u8 *fe = s.fe + 256 + 64 * ch, *pr = s.pr + 256 + 64 * ch;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                r[4 * j + i] = (int)fe[(by + j) * 8 + bx + i] - (int)pr[(by + j) * 8 + bx + i];
        i16 c[16], lv[16];
        fwd4x4(c, r);
        s.cdc[lane] = c[0];
        c[0] = 0;                                     // dct2x2dc takes the DCs out (macroblock.c:73-85)
        const u16 *mf = q4mf + (3 * 52 + g.qpc) * 16, *bs = q4bias + (3 * 52 + g.qpc) * 16;
        const int *dq = dq4 + 3 * 96 + (g.qpc % 6) * 16;
        int nz = 0, bits = g.qpc / 6 - 4;
#pragma unroll
        for (int i = 0; i < 16; i++) { int q = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)q; nz |= q; }
#pragma unroll
        for (int i = 0; i < 16; i++) lv[i] = nz ? c[c_scan4[g.field][i]] : (i16)0;
#pragma unroll
        for (int i = 0; i < 16; i++) { lc[16 * lane + i] = lv[i]; s.ccoef[lane][i] = nz ? (i16)dequant_one(c[i], dq[i], bits) : (i16)0; }
        s.cscore[lane] = (nz ? decimate_scan(lv + 1, 15, c_decimate4) : 0) | ((nz != 0) << 8);
    }
    WAVE_SYNC();
    // ---- chroma DC + decisions, one lane per channel (macroblock.c:320-356) ----
    if (lane < 2) {
        int ch = lane;
        int b0 = s.cdc[4 * ch], b1 = s.cdc[4 * ch + 1], b2 = s.cdc[4 * ch + 2], b3 = s.cdc[4 * ch + 3];
        int a0 = b0 + b1, a1 = b2 + b3, a2 = b0 - b1, a3 = b2 - b3;
        i16 d2[4] = {(i16)(a0 + a1), (i16)(a0 - a1), (i16)(a2 + a3), (i16)(a2 - a3)};   // [0][0] [0][1] [1][0] [1][1]
        const u16 *mf = q4mf + (3 * 52 + g.qpc) * 16, *bs = q4bias + (3 * 52 + g.qpc) * 16;
        int nz_dc = 0;
        for (int i = 0; i < 4; i++) { int q = quant_one(d2[i], (int)mf[0] >> 1, (int)bs[0] << 1); d2[i] = (i16)q; nz_dc |= q; }
        int score = 0, nz_ac = 0;
        u8 nzf[4];
        for (int i = 0; i < 4; i++) { int v = s.cscore[4 * ch + i]; nzf[i] = (u8)(v >> 8); if (v >> 8) { nz_ac = 1; score += v & 255; } }
        int e0 = d2[0] + d2[1], e1 = d2[2] + d2[3], e2 = d2[0] - d2[1], e3 = d2[2] - d2[3];
        int dmf = dq4[3 * 96 + (g.qpc % 6) * 16], qbits = g.qpc / 6 - 5;
        if (qbits > 0) { dmf <<= qbits; qbits = 0; }
        int mode;
        if (score < 7 || !nz_ac) { nzf[0] = nzf[1] = nzf[2] = nzf[3] = 0; mode = nz_dc ? 1 : 0; }
        else mode = 2;
        i16 *ldc = dc_c + (size_t)mb * 8 + 4 * ch;
        bool put = nz_dc != 0;
        ldc[0] = put ? d2[0] : (i16)0; ldc[1] = put ? d2[2] : (i16)0; ldc[2] = put ? d2[1] : (i16)0; ldc[3] = put ? d2[3] : (i16)0;
        s.cdcout[4 * ch + 0] = (i16)((e0 + e1) * dmf >> -qbits); s.cdcout[4 * ch + 1] = (i16)((e0 - e1) * dmf >> -qbits);
        s.cdcout[4 * ch + 2] = (i16)((e2 + e3) * dmf >> -qbits); s.cdcout[4 * ch + 3] = (i16)((e2 - e3) * dmf >> -qbits);
        if (!nz_dc) s.cdcout[4 * ch] = s.cdcout[4 * ch + 1] = s.cdcout[4 * ch + 2] = s.cdcout[4 * ch + 3] = 0;
        s.cmode[ch] = mode | (nz_dc ? 16 : 0);
        for (int i = 0; i < 4; i++) nnz[16 + 4 * ch + i] = nzf[i];
        nnz[24 + ch] = (u8)(nz_dc != 0);
    }
    WAVE_SYNC();
    // ---- chroma reconstruction ----
    if (lane < 8) {
        int ch = lane >> 2, i4 = lane & 3, bx = (i4 & 1) * 4, by = (i4 >> 1) * 4, mode = s.cmode[ch] & 15;
        u8 *pr = s.pr + 256 + 64 * ch;
        if (mode == 2) {
            int res[16];
            s.ccoef[lane][0] = s.cdcout[lane];
            inv4x4(res, s.ccoef[lane]);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    u8 *p = pr + (by + j) * 8 + bx + i;
                    *p = (u8)clip_u8((int)*p + res[4 * j + i]);
                }
        } else if (mode == 1) {
            int dc = (int)(i16)((s.cdcout[lane] + 32) >> 6);
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) {
                    u8 *p = pr + (by + j) * 8 + bx + i;
                    *p = (u8)clip_u8((int)*p + dc);
                }
        }
    }
    if (lane == 0) {
        int m0 = s.cmode[0], m1 = s.cmode[1];
        int cc = ((m0 & 15) == 2 || (m1 & 15) == 2) ? 2 : (((m0 | m1) & 16) ? 1 : 0);
        cbp_out[mb] = s.keep8 | (cc << 4);
    }
    WAVE_SYNC();
    // ---- per-frame mv / ref arrays as x264_macroblock_cache_save leaves them (R/common/macroblock.c:1264-1295) ----
    if (mv4x4_out && lane < 16) {
        const i16 *m = g.mv_per_mb == 1 ? mv + 2 * (size_t)mb : mv + ((size_t)mb * 16 + lane) * 2;
        mv4x4_out[((size_t)mb * 16 + lane) * 2] = m[0];
        mv4x4_out[((size_t)mb * 16 + lane) * 2 + 1] = m[1];
    }
    if (ref_out && lane < 4) ref_out[(size_t)mb * 4 + lane] = ref8 ? ref8[(size_t)mb * 4 + lane] : (signed char)0;
    // ---- write the reconstruction ----
    {
        int r = lane >> 2, x = (lane & 3) * 4;
        *(u32 *)(dy + oy + (ptrdiff_t)r * g.sy + x) = *(const u32 *)(s.pr + r * 16 + x);
        if (lane < 32) {
            int chn = lane >> 4, l = lane & 15, cr = l >> 1, cx4 = (l & 1) * 4;
            u8 *d = (chn ? dv : du) + oc + (ptrdiff_t)cr * g.sc + cx4;
            *(u32 *)d = *(const u32 *)(s.pr + 256 + 64 * chn + cr * 8 + cx4);
        }
    }
#undef WAVE_SYNC
}

// ---------------------------------------------------------------- probe skip
// x264_macroblock_probe_skip (R/encoder/macroblock.c:797-883, P path): would the macroblock,
// predicted with the P-skip vector, quantise to nothing?  Luma: 16 4x4 blocks, summed
// decimate_score16 < 6; chroma per plane: if SSD >= (lambda2[qpc]+32)>>6, the 2x2 DC must
// quantise to zero and the summed decimate_score15 stay < 7.
__global__ __launch_bounds__(64 * RS_WAVES) void k_probe_skip(
    const u8 *__restrict__ fy, const u8 *__restrict__ fu, const u8 *__restrict__ fv, RefTab refs, ResGeom g, int chroma_thresh,
    const u16 *__restrict__ q4mf, const u16 *__restrict__ q4bias, const i16 *__restrict__ mv, u8 *__restrict__ skip_out)
{
    __shared__ ResLds s_all[RS_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int mb = xcd_band_order(blockIdx.x, gridDim.x) * RS_WAVES + wave;
    if (mb >= g.mb_w * g.mb_h) return;
    const size_t bz = blockIdx.y, nmb = (size_t)g.mb_w * g.mb_h;
    fy += g.bs_y * bz; fu += g.bs_c * bz; fv += g.bs_c * bz; mv += 2 * nmb * bz; skip_out += nmb * bz;
    ResLds &s = s_all[wave];
    const int mbx = mb % g.mb_w, mby = mb / g.mb_w;
    const ptrdiff_t oy = (ptrdiff_t)16 * mby * g.sy + 16 * mbx, oc = (ptrdiff_t)8 * mby * g.sc + 8 * mbx;
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); \
                         __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
    // the P-skip vector is clipped to mv_min / mv_max first (macroblock.c:813-814; analyse.c:266-267,296-297)
    if (lane == 0) {
        int vx = clip3(mv[2 * mb], 4 * (-16 * mbx - 24), 4 * (16 * (g.mb_w - mbx - 1) + 24));
        int vy = clip3(mv[2 * mb + 1], 4 * (-16 * mby - 24), 4 * (16 * (g.mb_h - mby - 1) + 24));
        s.cdc[0] = (i16)vx; s.cdc[1] = (i16)vy;
    }
    WAVE_SYNC();
    ResGeom g1 = g; g1.mv_per_mb = 1;
    predict_mb(s, refs, g1, s.cdc - 2 * (ptrdiff_t)mb, nullptr, mb, lane, oy, oc, bz);
    {
        int r = lane >> 2, x = (lane & 3) * 4, cx = lane & 7, cy = lane >> 3;
        *(u32 *)(s.fe + r * 16 + x) = *(const u32 *)(fy + oy + (ptrdiff_t)r * g.sy + x);
        s.fe[256 + lane] = fu[oc + (ptrdiff_t)cy * g.sc + cx];
        s.fe[320 + lane] = fv[oc + (ptrdiff_t)cy * g.sc + cx];
    }
    WAVE_SYNC();
    // lanes 0-15 luma 4x4 blocks, lanes 16-23 chroma 4x4 blocks, lanes 24-25 chroma SSD halves... keep it simple:
    int score = 0, dc = 0, ssd = 0;
    if (lane < 24) {
        const bool luma = lane < 16;
        int bx, by, r[16];
        const u8 *fe, *pr; int st;
        if (luma) { blk_xy(lane, bx, by); fe = s.fe; pr = s.pr; st = 16; }
        else { int l = lane - 16, i4 = l & 3; bx = (i4 & 1) * 4; by = (i4 >> 1) * 4; fe = s.fe + 256 + 64 * (l >> 2); pr = s.pr + 256 + 64 * (l >> 2); st = 8; }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                r[4 * j + i] = (int)fe[(by + j) * st + bx + i] - (int)pr[(by + j) * st + bx + i];
                ssd += r[4 * j + i] * r[4 * j + i];
            }
        i16 c[16], lv[16];
        fwd4x4(c, r);
        const int cat = luma ? 1 : 3, q = luma ? g.qp : g.qpc;
        const u16 *mf = q4mf + (cat * 52 + q) * 16, *bs = q4bias + (cat * 52 + q) * 16;
        if (!luma) { dc = c[0]; c[0] = 0; }
        int nz = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { int qq = quant_one(c[i], mf[i], bs[i]); c[i] = (i16)qq; nz |= qq; }
#pragma unroll
        for (int i = 0; i < 16; i++) lv[i] = c[c_scan4[g.field][i]];
        if (nz) score = luma ? decimate_scan(lv, 16, c_decimate4) : decimate_scan(lv + 1, 15, c_decimate4);
    }
    // gather through shuffles: lane 0 decides
    int luma_sum = 0, c_sum[2] = {0, 0}, c_ssd[2] = {0, 0}, c_dc[2][4];
#pragma unroll
    for (int k = 0; k < 16; k++) luma_sum += __shfl(score, k, 64);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        c_sum[k >> 2] += __shfl(score, 16 + k, 64);
        c_ssd[k >> 2] += __shfl(ssd, 16 + k, 64);
        c_dc[k >> 2][k & 3] = __shfl(dc, 16 + k, 64);
    }
    if (lane == 0) {
        int ok = luma_sum < 6;
        const u16 *mf = q4mf + (3 * 52 + g.qpc) * 16, *bs = q4bias + (3 * 52 + g.qpc) * 16;
        for (int ch = 0; ch < 2 && ok; ch++) {
            if (c_ssd[ch] < chroma_thresh) continue;
            int b0 = c_dc[ch][0], b1 = c_dc[ch][1], b2 = c_dc[ch][2], b3 = c_dc[ch][3];
            int a0 = b0 + b1, a1 = b2 + b3, a2 = b0 - b1, a3 = b2 - b3;
            int d2[4] = {(i16)(a0 + a1), (i16)(a0 - a1), (i16)(a2 + a3), (i16)(a2 - a3)};
            int nzdc = 0;
            for (int i = 0; i < 4; i++) nzdc |= quant_one(d2[i], (int)mf[0] >> 1, (int)bs[0] << 1);
            if (nzdc || c_sum[ch] >= 7) ok = 0;
        }
        skip_out[mb] = (u8)ok;
    }
#undef WAVE_SYNC
}

static void fill_reftab(RefTab &t, const x264hip_picture *const *refs, int n)
{
    for (int i = 0; i < RS_MAX_REFS; i++) {
        const x264hip_picture *r = refs[i < n ? i : 0];
        for (int k = 0; k < 4; k++) t.y[i][k] = r->filtered[k];
        t.u[i] = r->plane[1]; t.v[i] = r->plane[2];
    }
}

static int launch_residual(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *const *refs, int n_refs,
                           x264hip_picture *recon, const x264hip_residual_params *p, const int16_t *mv_dev, int mv_per_mb,
                           const int8_t *ref8_dev, int16_t *levels_y_dev, int16_t *levels_c_dev, int16_t *dc_c_dev,
                           int32_t *cbp_dev, uint8_t *nnz_dev)
{
    if (p->qp < 0 || p->qp > 51 || p->qp_chroma < 0 || p->qp_chroma > 51) { set_error("residual: qp out of range"); return -1; }
    if (n_refs < 1 || n_refs > RS_MAX_REFS) { set_error("residual: %d references (1..%d)", n_refs, RS_MAX_REFS); return -1; }
    ResGeom g = {c->d.mb_w, c->d.mb_h, c->d.stride_y, c->d.stride_c, p->qp, p->qp_chroma, !!p->b_interlaced, c->bs_y, c->bs_c, mv_per_mb};
    RefTab t;
    fill_reftab(t, refs, n_refs);
    int n = g.mb_w * g.mb_h;
    dim3 grid((n + RS_WAVES - 1) / RS_WAVES, c->batch), block(64 * RS_WAVES);
#define ARGS fenc->plane[0], fenc->plane[1], fenc->plane[2], t, recon->plane[0], recon->plane[1], recon->plane[2], g, p->quant4_mf, \
        p->quant4_bias, p->quant8_mf, p->quant8_bias, p->dequant4_mf, p->dequant8_mf, mv_dev, (const signed char *)ref8_dev, \
        levels_y_dev, levels_c_dev, dc_c_dev, cbp_dev, nnz_dev, p->mv4x4_out, (signed char *)p->ref_out
    if (p->transform8x8) hipLaunchKernelGGL(k_inter_residual<1>, grid, block, 0, c->stream, ARGS);
    else hipLaunchKernelGGL(k_inter_residual<0>, grid, block, 0, c->stream, ARGS);
#undef ARGS
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int x264hip_inter_residual_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                                            x264hip_picture *recon, const x264hip_residual_params *p,
                                            const int16_t *mv_qpel_dev, int16_t *levels_y_dev, int16_t *levels_c_dev,
                                            int16_t *dc_c_dev, int32_t *cbp_dev, uint8_t *nnz_dev)
{
    return launch_residual(c, fenc, &ref, 1, recon, p, mv_qpel_dev, 1, nullptr, levels_y_dev, levels_c_dev, dc_c_dev, cbp_dev, nnz_dev);
}

extern "C" int x264hip_inter_residual_frame_mp(x264hip_frame_ctx *c, const x264hip_picture *fenc,
                                               const x264hip_picture *const *refs, int n_refs, x264hip_picture *recon,
                                               const x264hip_residual_params *p, const int16_t *mv4x4_dev, const int8_t *ref8x8_dev,
                                               int16_t *levels_y_dev, int16_t *levels_c_dev, int16_t *dc_c_dev, int32_t *cbp_dev,
                                               uint8_t *nnz_dev)
{
    if (p->transform8x8 && mv4x4_dev == nullptr) { set_error("residual_mp: mv4x4 missing"); return -1; }
    return launch_residual(c, fenc, refs, n_refs, recon, p, mv4x4_dev, 16, ref8x8_dev, levels_y_dev, levels_c_dev, dc_c_dev, cbp_dev, nnz_dev);
}

extern "C" int x264hip_probe_skip_frame(x264hip_frame_ctx *c, const x264hip_picture *fenc, const x264hip_picture *ref,
                                        const x264hip_residual_params *p, int lambda2_chroma, const int16_t *pskip_mv_dev,
                                        uint8_t *skip_out_dev)
{
    if (p->qp < 0 || p->qp > 51 || p->qp_chroma < 0 || p->qp_chroma > 51) { set_error("probe_skip: qp out of range"); return -1; }
    ResGeom g = {c->d.mb_w, c->d.mb_h, c->d.stride_y, c->d.stride_c, p->qp, p->qp_chroma, !!p->b_interlaced, c->bs_y, c->bs_c, 1};
    RefTab t;
    fill_reftab(t, &ref, 1);
    int n = g.mb_w * g.mb_h;
    hipLaunchKernelGGL(k_probe_skip, dim3((n + RS_WAVES - 1) / RS_WAVES, c->batch), dim3(64 * RS_WAVES), 0, c->stream,
                       fenc->plane[0], fenc->plane[1], fenc->plane[2], t, g, (lambda2_chroma + 32) >> 6, p->quant4_mf, p->quant4_bias,
                       pskip_mv_dev, skip_out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}
