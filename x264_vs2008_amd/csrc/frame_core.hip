// frame_core.hip -- FRAME LEVEL, part 1: pictures resident in HBM and the
// dependency-free whole-plane kernels (border expansion, half-pel planes,
// half-resolution planes, AQ energy, SSD).  These are the operations the
// reference runs per frame / per macroblock row behind table pointers
// (R/common/mc.c:404-463, :306-357; R/common/frame.c:218-334;
// R/encoder/ratecontrol.c:171-195; R/encoder/encoder.c:1034-1045), so they
// drop in with no control-flow change in the caller.
//
// Memory layout (x264_frame_new, R/common/frame.c:29-152): every plane keeps
// PADH/PADV = 32 (luma) or 16 (chroma) pixels of padding around the coded
// picture; stride_y = ALIGN(16*mb_w + 64, 16).  Pointers in x264hip_picture
// address pixel (0,0).  All kernels are HBM-bound byte streams: coalesced
// dword / 16-byte accesses, LDS only where a tile is reused (the 6-tap filter).
#include <cstdlib>
#include <cstring>
#include "device_prims.h"
#include "frame_internal.h"
#include "x264hip_lookahead.h"

using namespace x264hip;

// ------------------------------------------------------------------ kernels
// x264_frame_expand_border_mod16 (R/common/frame.c:303-334): replicate the
// last visible column / row into the coded area.
__global__ void k_pad_mod16(u8 *p, int stride, int w, int h, int w16, int h16)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w16 || y >= h16) return;
    if (x < w && y < h) return;
    int sx = x < w ? x : w - 1, sy = y < h ? y : h - 1;
    p[y * stride + x] = p[sy * stride + sx];
}

// The integer-only synthetic I420 source of SURVEY.md 8(d) (x264_vs2008_amd/synth.py is the same generator on the host): a smooth
// translating texture, a moving 64x64 box and +-3 LSB hash noise, uint32 wrap-around arithmetic.  Frame index of batch element b is
// t0 + b * t_stride, so every (chain, display index) of a benchmark run is a picture of its own.  One dword (4 pixels) per thread;
// columns / rows beyond the visible picture replicate its last column / row (x264_frame_expand_border_mod16).
__device__ __forceinline__ int synth_tri(int v) { const int a = v & 255; return a < 128 ? a - 64 : 191 - a; }
__device__ __forceinline__ int synth_noise(u32 x, u32 y, u32 t, u32 p, int amp)
{
    u32 h = x * 73856093u ^ y * 19349663u ^ t * 83492791u ^ p * 2654435761u ^ 1234u;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (int)(h % (u32)(2 * amp + 1)) - amp;
}
__global__ __launch_bounds__(256) void k_synth_plane(u8 *pix, size_t bs, int stride, int plane, int w, int h, int w16, int h16, int full_w, int full_h,
                                                     int t0, int t_stride)
{
    const int xq = (blockIdx.x * blockDim.x + threadIdx.x) * 4, yy = blockIdx.y;
    if (xq >= w16) return;
    const int t = t0 + (int)blockIdx.z * t_stride, y = yy < h ? yy : h - 1;
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = xq + i < w ? xq + i : w - 1;
        int v;
        if (plane == 0) {
            v = 128 + (synth_tri((x + 3 * t) * 4) >> 1) + (synth_tri((y - 2 * t) * 6) >> 2) + (synth_tri((x + y + 5 * t) * 9) >> 3);
            const int bx = (40 + 7 * t) % (full_w - 64), by = (30 + 3 * t) % (full_h - 64);
            if (x >= bx && x < bx + 64 && y >= by && y < by + 64) v += synth_tri(x * 16) >> 1;
            v += synth_noise((u32)x, (u32)y, (u32)t, 0, 3);
        } else if (plane == 1) v = 128 + (synth_tri((x + 2 * t) * 3) >> 2) + synth_noise((u32)x, (u32)y, (u32)t, 1, 1);
        else v = 128 + (synth_tri((y - t) * 5) >> 2) + synth_noise((u32)x, (u32)y, (u32)t, 2, 1);
        out |= (u32)clip_u8(v) << (8 * i);
    }
    *(u32 *)(pix + bs * blockIdx.z + (size_t)yy * stride + xq) = out;
}

// plane_expand_border (R/common/frame.c:218-240) for the whole plane at once:
// every padding byte takes the nearest interior pixel (left/right bands first,
// then whole rows copied up/down, which is the same thing).
// Grid: y over [-padv, height+padv), x in dwords over [-padh, width+padh).
__global__ void k_expand_border(u8 *pix, size_t bs, int stride, int width, int height, int padh, int padv, const int *elems)
{
    pix += bs * (elems ? elems[blockIdx.z] : blockIdx.z);
    int xq = blockIdx.x * blockDim.x + threadIdx.x;      // dword index from -padh
    int y = (int)blockIdx.y - padv;
    int x = xq * 4 - padh;
    if (x >= width + padh) return;
    bool row_inside = y >= 0 && y < height;
    if (row_inside && x >= 0 && x + 3 < width) return;   // interior dword: untouched
    int sy = y < 0 ? 0 : (y >= height ? height - 1 : y);
    const u8 *srow = pix + (ptrdiff_t)sy * stride;
    u8 *drow = pix + (ptrdiff_t)y * stride;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int xx = x + i;
        if (row_inside && xx >= 0 && xx < width) continue;
        int sx = xx < 0 ? 0 : (xx >= width ? width - 1 : xx);
        drow[xx] = srow[sx];
    }
}

// Half-pel planes of the whole frame (hpel_filter, R/common/mc.c:133-155,
// driven over the region x264_frame_filter covers, mc.c:404-426):
//   outputs for x in [x_lo, x_lo + nx), y in [y_lo, y_lo + ny)
// One workgroup = 64 x 16 output pixels.  LDS: source tile 21 rows x 72 B and
// the raw vertical taps (int16) 16 rows x 72, so the HV plane reuses them.
#define HP_TW 64
#define HP_TH 16
__global__ __launch_bounds__(256) void k_hpel(const u8 *__restrict__ src, u8 *__restrict__ dh, u8 *__restrict__ dv,
                                              u8 *__restrict__ dc, size_t bs, int stride, int x_lo, int y_lo, int nx, int ny, const int *elems)
{
    { const size_t be = elems ? elems[blockIdx.z] : blockIdx.z; src += bs * be; dh += bs * be; dv += bs * be; dc += bs * be; }
    __shared__ u32 s_src[21 * 18];          // 21 rows x 72 bytes, column 0 = x0 - 4
    __shared__ i16 s_v[HP_TH * 72];         // raw vertical 6-tap, column 0 = x0 - 4
    const int tid = threadIdx.x;
    const int x0 = x_lo + blockIdx.x * HP_TW, y0 = y_lo + blockIdx.y * HP_TH;
    // x0 - 4 is 4-byte aligned because x_lo is a multiple of 4 and planes are 16-B aligned
    for (int i = tid; i < 21 * 18; i += 256) {
        int r = i / 18, c = i % 18;
        s_src[i] = *(const u32 *)(src + (ptrdiff_t)(y0 - 2 + r) * stride + (x0 - 4) + 4 * c);
    }
    __syncthreads();
    const u8 *t = (const u8 *)s_src;
    // vertical taps for tile columns 0..71 (= x0-4 .. x0+67), rows 0..15
    for (int i = tid; i < HP_TH * 72; i += 256) {
        int r = i / 72, c = i % 72;
        const u8 *p = t + (r + 2) * 72 + c;
        s_v[i] = (i16)tap6(p[-2 * 72], p[-72], p[0], p[72], p[2 * 72], p[3 * 72]);
    }
    __syncthreads();
    const int r = tid >> 4, c4 = (tid & 15) * 4;      // 4 consecutive outputs per thread
    const int y = y0 + r, x = x0 + c4;
    if (y >= y_lo + ny || x >= x_lo + nx) return;
    u32 oh = 0, ov = 0, oc = 0;
    // the 4 outputs of this thread need source bytes and raw vertical taps of
    // tile columns c4+2 .. c4+10: fetch them once into registers
    int sp[9], vp[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        sp[k] = t[(r + 2) * 72 + c4 + 2 + k];
        vp[k] = s_v[r * 72 + c4 + 2 + k];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int hh = clip_u8((tap6(sp[i], sp[i + 1], sp[i + 2], sp[i + 3], sp[i + 4], sp[i + 5]) + 16) >> 5);
        int vv = clip_u8((vp[i + 2] + 16) >> 5);
        int cc = clip_u8((tap6(vp[i], vp[i + 1], vp[i + 2], vp[i + 3], vp[i + 4], vp[i + 5]) + 512) >> 10);
        oh |= (u32)hh << (8 * i); ov |= (u32)vv << (8 * i); oc |= (u32)cc << (8 * i);
    }
    ptrdiff_t o = (ptrdiff_t)y * stride + x;
    if (x + 3 < x_lo + nx) {
        *(u32 *)(dh + o) = oh; *(u32 *)(dv + o) = ov; *(u32 *)(dc + o) = oc;
    } else {
        for (int i = 0; x + i < x_lo + nx; i++) {
            dh[o + i] = (u8)(oh >> (8 * i)); dv[o + i] = (u8)(ov >> (8 * i)); dc[o + i] = (u8)(oc >> (8 * i));
        }
    }
}

// frame_init_lowres_core (R/common/mc.c:333-357): four half-resolution planes.
// Each thread makes 4 consecutive pixels of each plane from three source rows.
__global__ __launch_bounds__(256) void k_lowres(const u8 *__restrict__ src, u8 *__restrict__ d0, u8 *__restrict__ dh,
                                                u8 *__restrict__ dv, u8 *__restrict__ dc, size_t bs_src, size_t bs_dst, int ss, int ds, int w, int h)
{
    src += bs_src * blockIdx.z; d0 += bs_dst * blockIdx.z; dh += bs_dst * blockIdx.z; dv += bs_dst * blockIdx.z; dc += bs_dst * blockIdx.z;
    int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const u8 *r0 = src + (ptrdiff_t)2 * y * ss + 2 * x, *r1 = r0 + ss, *r2 = r1 + ss;
    u8 a[3][12];
    // 2x is 8-byte aligned: two dwords + one more for the +2 taps
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const u8 *r = k == 0 ? r0 : k == 1 ? r1 : r2;
        u32 w0 = *(const u32 *)r, w1 = *(const u32 *)(r + 4), w2 = *(const u32 *)(r + 8);
#pragma unroll
        for (int i = 0; i < 4; i++) { a[k][i] = (u8)(w0 >> (8 * i)); a[k][4 + i] = (u8)(w1 >> (8 * i)); a[k][8 + i] = (u8)(w2 >> (8 * i)); }
    }
    u32 o0 = 0, oh = 0, ov = 0, oc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int k = 2 * i;
        o0 |= (u32)avg4r(a[0][k], a[1][k], a[0][k + 1], a[1][k + 1]) << (8 * i);
        oh |= (u32)avg4r(a[0][k + 1], a[1][k + 1], a[0][k + 2], a[1][k + 2]) << (8 * i);
        ov |= (u32)avg4r(a[1][k], a[2][k], a[1][k + 1], a[2][k + 1]) << (8 * i);
        oc |= (u32)avg4r(a[1][k + 1], a[2][k + 1], a[1][k + 2], a[2][k + 2]) << (8 * i);
    }
    ptrdiff_t o = (ptrdiff_t)y * ds + x;
    if (x + 3 < w) {
        *(u32 *)(d0 + o) = o0; *(u32 *)(dh + o) = oh; *(u32 *)(dv + o) = ov; *(u32 *)(dc + o) = oc;
    } else
        for (int i = 0; x + i < w; i++) {
            d0[o + i] = (u8)(o0 >> (8 * i)); dh[o + i] = (u8)(oh >> (8 * i));
            dv[o + i] = (u8)(ov >> (8 * i)); dc[o + i] = (u8)(oc >> (8 * i));
        }
}
// x264_frame_init_lowres's edge duplication (R/common/mc.c:314-317): column
// `width` := column width-1 for rows < height, then row `height` := row height-1.
__global__ void k_dup_edge(u8 *p, size_t bs, int stride, int width, int height)
{
    p += bs * blockIdx.z;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < height) p[(ptrdiff_t)i * stride + width] = p[(ptrdiff_t)i * stride + width - 1];
}
__global__ void k_dup_row(u8 *p, size_t bs, int stride, int width, int height)
{
    p += bs * blockIdx.z;
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < width) p[(ptrdiff_t)height * stride + x] = p[(ptrdiff_t)(height - 1) * stride + x];
    // The sample below-right of the picture feeds the last pixel of the HV lowres plane, and the reference never writes it (it copies
    // `width` samples of the last row, not width + 1): there it holds what malloc or an earlier reconstruction's border left.  0 here,
    // the value of a fresh page -- what oracle/ref_slice.c pins the reference to.
    if (x == 0) p[(ptrdiff_t)height * stride + width] = 0;
}

// AQ energy per macroblock: var16x16(Y) + var8x8(U) + var8x8(V)
// (ac_energy_mb, R/encoder/ratecontrol.c:171-195; pixel var R/common/pixel.c:142-161).
// One wavefront per macroblock: lane = (row, 4-pixel group).
__global__ __launch_bounds__(256) void k_aq_var(const u8 *__restrict__ py, const u8 *__restrict__ pu, const u8 *__restrict__ pv,
                                                size_t bs_y, size_t bs_c, int sy, int sc, int mb_w, int mb_count, int *__restrict__ out)
{
    py += bs_y * blockIdx.z; pu += bs_c * blockIdx.z; pv += bs_c * blockIdx.z; out += (size_t)mb_count * blockIdx.z;
    int mb = xcd_band_order(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (mb >= mb_count) return;
    int mx = mb % mb_w, my = mb / mb_w;
    u32 w = *(const u32 *)(py + (ptrdiff_t)(my * 16 + (lane >> 2)) * sy + mx * 16 + (lane & 3) * 4);
    u32 sum = (w & 255) + ((w >> 8) & 255) + ((w >> 16) & 255) + (w >> 24);
    u32 sqr = (w & 255) * (w & 255) + ((w >> 8) & 255) * ((w >> 8) & 255) + ((w >> 16) & 255) * ((w >> 16) & 255) + (w >> 24) * (w >> 24);
    sum = wave_sum_u32(sum); sqr = wave_sum_u32(sqr);
    u32 vy = sqr - (sum * sum >> 8);
    // chroma: lanes 0-15 -> U (row = lane>>1, half = lane&1), lanes 16-31 -> V
    u32 cs = 0, cq = 0;
    if (lane < 32) {
        const u8 *pc = (lane < 16 ? pu : pv) + (ptrdiff_t)(my * 8 + ((lane & 15) >> 1)) * sc + mx * 8 + (lane & 1) * 4;
        u32 c = *(const u32 *)pc;
        cs = (c & 255) + ((c >> 8) & 255) + ((c >> 16) & 255) + (c >> 24);
        cq = (c & 255) * (c & 255) + ((c >> 8) & 255) * ((c >> 8) & 255) + ((c >> 16) & 255) * ((c >> 16) & 255) + (c >> 24) * (c >> 24);
    }
    u32 us = (u32)group_sum((int)cs, 16), uq = (u32)group_sum((int)cq, 16);
    u32 vu = uq - (us * us >> 6);                       // lanes 0-15 hold U, 16-31 hold V
    u32 vv = (u32)__shfl((int)vu, 16, 64);
    u32 e = vy + vu + vv;
    if (lane == 0) out[mb] = (int)(e ? e : 1u);         // X264_MAX(var, 1), ratecontrol.c:190
}

// sum of squared differences of two planes (x264_pixel_ssd_wxh, R/common/pixel.c:98-136)
__global__ __launch_bounds__(256) void k_ssd(const u8 *__restrict__ a, int sa, const u8 *__restrict__ b, int sb,
                                             size_t bs, int w, int h, unsigned long long *acc)
{
    a += bs * blockIdx.z; b += bs * blockIdx.z; acc += 3 * blockIdx.z;
    unsigned long long part = 0;
    int nq = (w + 3) >> 2;
    for (int y = blockIdx.y; y < h; y += gridDim.y)
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
            int x = q * 4;
            const u8 *pa = a + (ptrdiff_t)y * sa + x, *pb = b + (ptrdiff_t)y * sb + x;
            for (int i = 0; i < 4 && x + i < w; i++) { int d = (int)pa[i] - (int)pb[i]; part += (unsigned)(d * d); }
        }
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if ((threadIdx.x & 63) == 0 && part) atomicAdd(acc, part);
}

// --------------------------------------------------------------------- host
static int alloc_plane(u8 **out, int stride, int lines, int padh, int padv, int batch, size_t bs, hipStream_t s)
{
    u8 *base = nullptr;
    HIPCHK(hipMalloc((void **)&base, bs * batch + 256));
    if (zero_async(base, bs * batch + 256, s)) return -1;
    *out = base + (size_t)stride * padv + padh;
    return 0;
}
static void free_plane(u8 *p, int stride, int padh, int padv)
{
    if (p) (void)hipFree(p - (size_t)stride * padv - padh);
}

extern "C" x264hip_frame_ctx *x264hip_frame_ctx_new(x264hip_frame_dims *d, void *hip_stream)
{
    if (!initialised()) { set_error("x264hip_frame_ctx_new: call x264hip_init first"); return nullptr; }
    if (d->width < 16 || d->height < 16 || d->width > 16384 || d->height > 16384) {
        set_error("unsupported frame size %dx%d", d->width, d->height);
        return nullptr;
    }
    if (d->batch < 0 || d->batch > 4096) { set_error("unsupported batch %d", d->batch); return nullptr; }
    x264hip_frame_ctx *c = (x264hip_frame_ctx *)calloc(1, sizeof(*c));
    if (!c) return nullptr;
    if (d->batch == 0) d->batch = 1;
    d->mb_w = (d->width + 15) / 16; d->mb_h = (d->height + 15) / 16;
    d->stride_y = align_up(d->mb_w * 16 + 2 * PADH, 16);
    d->stride_c = align_up(d->stride_y >> 1, 16);
    d->lines_y = d->mb_h * 16; d->lines_c = d->lines_y / 2;
    c->d = *d;
    c->batch = d->batch; c->sel = 0;
    c->width16 = d->mb_w * 16; c->lines16 = d->lines_y;
    c->width_l = c->width16 / 2; c->lines_l = d->lines_y / 2; c->stride_l = align_up(c->width_l + 2 * PADH, 16);
    // +stride slack: tile loaders may read a few dwords past the last padded row's end
    c->bs_y = align_up_sz((size_t)d->stride_y * (d->lines_y + 2 * PADV + 1), 256);
    c->bs_c = align_up_sz((size_t)d->stride_c * (d->lines_c + PADV + 1), 256);
    c->bs_l = align_up_sz((size_t)c->stride_l * (c->lines_l + 2 * PADV + 1), 256);
    if (hipSetDevice(device_id()) != hipSuccess) { free(c); return nullptr; }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { free(c); return nullptr; }
    else c->own_stream = true;
    if (hipMalloc((void **)&c->ssd_dev, 24 * (size_t)c->batch + 64) != hipSuccess) { free(c); return nullptr; }
    (void)hipMemset(c->ssd_dev, 0, 24 * (size_t)c->batch + 64);   // the tail holds the context's sticky sweep-abort counter (frame_slice.hip)
    return c;
}
extern "C" void x264hip_frame_ctx_delete(x264hip_frame_ctx *c)
{
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    if (c->ssd_dev) (void)hipFree(c->ssd_dev);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    free(c);
}
extern "C" void *x264hip_frame_ctx_stream(x264hip_frame_ctx *c) { return (void *)c->stream; }
extern "C" int x264hip_sync(x264hip_frame_ctx *c) { HIPCHK(hipStreamSynchronize(c->stream)); return 0; }
extern "C" int x264hip_frame_ctx_elements(x264hip_frame_ctx *c, const int *elems_dev, int n)
{
    if (elems_dev && (n <= 0 || n > c->batch)) { set_error("frame_ctx_elements: %d elements of a batch of %d", n, c->batch); return -1; }
    c->elems = elems_dev; c->n_elems = elems_dev ? n : 0;
    return 0;
}
extern "C" int x264hip_frame_ctx_select(x264hip_frame_ctx *c, int batch_index)
{
    if (batch_index < 0 || batch_index >= c->batch) { set_error("batch index %d out of range", batch_index); return -1; }
    c->sel = batch_index;
    return 0;
}

extern "C" int x264hip_picture_alloc(x264hip_frame_ctx *c, x264hip_picture *pic)
{
    memset(pic, 0, sizeof(*pic));
    const x264hip_frame_dims &d = c->d;
    if (alloc_plane(&pic->plane[0], d.stride_y, d.lines_y, PADH, PADV, c->batch, c->bs_y, c->stream)) return -1;
    for (int i = 1; i < 3; i++)
        if (alloc_plane(&pic->plane[i], d.stride_c, d.lines_c, PADH / 2, PADV / 2, c->batch, c->bs_c, c->stream)) return -1;
    pic->filtered[0] = pic->plane[0];
    for (int i = 1; i < 4; i++)
        if (alloc_plane(&pic->filtered[i], d.stride_y, d.lines_y, PADH, PADV, c->batch, c->bs_y, c->stream)) return -1;
    pic->width_lowres = c->width_l; pic->lines_lowres = c->lines_l; pic->stride_lowres = c->stride_l;
    for (int i = 0; i < 4; i++)
        if (alloc_plane(&pic->lowres[i], c->stride_l, c->lines_l, PADH, PADV, c->batch, c->bs_l, c->stream)) return -1;
    return 0;
}
// A source picture: Y, U, V only (x264_frame_t of an input frame needs no half-pel planes; 3.5 MB per 1080p chain instead of 12.7)
extern "C" int x264hip_picture_alloc_source(x264hip_frame_ctx *c, x264hip_picture *pic)
{
    memset(pic, 0, sizeof(*pic));
    const x264hip_frame_dims &d = c->d;
    if (alloc_plane(&pic->plane[0], d.stride_y, d.lines_y, PADH, PADV, c->batch, c->bs_y, c->stream)) return -1;
    for (int i = 1; i < 3; i++)
        if (alloc_plane(&pic->plane[i], d.stride_c, d.lines_c, PADH / 2, PADV / 2, c->batch, c->bs_c, c->stream)) return -1;
    pic->filtered[0] = pic->plane[0];
    return 0;
}
// A lookahead slot's picture: the source planes and the half-resolution planes (include/x264hip_lookahead.h)
extern "C" int x264hip_picture_alloc_lookahead(x264hip_frame_ctx *c, x264hip_picture *pic)
{
    if (x264hip_picture_alloc_source(c, pic)) return -1;
    pic->width_lowres = c->width_l; pic->lines_lowres = c->lines_l; pic->stride_lowres = c->stride_l;
    for (int i = 0; i < 4; i++)
        if (alloc_plane(&pic->lowres[i], c->stride_l, c->lines_l, PADH, PADV, c->batch, c->bs_l, c->stream)) return -1;
    return 0;
}
// batch element src_b of `src` -> element dst_b of `dst` (Y, U, V with their padding), on the context's stream
extern "C" int x264hip_picture_copy_element(x264hip_frame_ctx *c, x264hip_picture *dst, int dst_b, const x264hip_picture *src, int src_b)
{
    const x264hip_frame_dims &d = c->d;
    if (dst_b < 0 || dst_b >= c->batch || src_b < 0 || src_b >= c->batch) { set_error("picture_copy_element: batch index"); return -1; }
    for (int i = 0; i < 3; i++) {
        const size_t bs = i ? c->bs_c : c->bs_y, off = i ? (size_t)d.stride_c * (PADV / 2) + PADH / 2 : (size_t)d.stride_y * PADV + PADH;
        HIPCHK(hipMemcpyAsync(dst->plane[i] - off + bs * dst_b, src->plane[i] - off + bs * src_b, bs, hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}
extern "C" void x264hip_picture_free(x264hip_frame_ctx *c, x264hip_picture *pic)
{
    const x264hip_frame_dims &d = c->d;
    (void)hipStreamSynchronize(c->stream);
    free_plane(pic->plane[0], d.stride_y, PADH, PADV);
    for (int i = 1; i < 3; i++) free_plane(pic->plane[i], d.stride_c, PADH / 2, PADV / 2);
    for (int i = 1; i < 4; i++) free_plane(pic->filtered[i], d.stride_y, PADH, PADV);
    for (int i = 0; i < 4; i++) free_plane(pic->lowres[i], pic->stride_lowres ? pic->stride_lowres : 1, PADH, PADV);
    memset(pic, 0, sizeof(*pic));
}

extern "C" int x264hip_picture_upload(x264hip_frame_ctx *c, x264hip_picture *pic, const uint8_t *y, int sy,
                                      const uint8_t *u, int su, const uint8_t *v, int sv)
{
    const x264hip_frame_dims &d = c->d;
    const u8 *src[3] = {y, u, v};
    const int ss[3] = {sy, su, sv};
    for (int i = 0; i < 3; i++) {
        int w = d.width >> !!i, h = d.height >> !!i, st = i ? d.stride_c : d.stride_y;
        u8 *dst = pic->plane[i] + (i ? c->bs_c : c->bs_y) * c->sel;
        HIPCHK(hipMemcpy2DAsync(dst, st, src[i], ss[i], w, h, hipMemcpyHostToDevice, c->stream));
        // (a copy from pageable memory may leave hipErrorInvalidValue as the thread's "last error" although it returns success -- the
        // runtime's own look-up of the host pointer; seen on ROCm 7.2 -- so the launch check below must not read what the copy left)
        (void)hipGetLastError();
        int w16 = c->width16 >> !!i, h16 = c->lines16 >> !!i;
        if (w16 != w || h16 != h) {
            hipLaunchKernelGGL(k_pad_mod16, dim3((w16 + 255) / 256, h16), dim3(256), 0, c->stream, dst, st, w, h, w16, h16);
            HIPCHK(hipGetLastError());
        }
    }
    // the caller owns y/u/v and may release them on return (they are usually
    // pageable): do not leave a DMA reading them in flight
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Synthetic source (k_synth_plane): element b of `pic` becomes frame t0 + b * t_stride of the SURVEY 8(d) clip, padded to the coded size.
// Asynchronous on the context's stream.  Needs width, height >= 65 (the moving box).
extern "C" int x264hip_picture_synth(x264hip_frame_ctx *c, x264hip_picture *pic, int t0, int t_stride)
{
    const x264hip_frame_dims &d = c->d;
    if (d.width < 65 || d.height < 65 || t0 < 0 || t_stride < 0) { set_error("picture_synth: needs a picture larger than 64x64 and non-negative frame numbers"); return -1; }
    for (int i = 0; i < 3; i++) {
        const int w = d.width >> !!i, h = d.height >> !!i, st = i ? d.stride_c : d.stride_y, w16 = c->width16 >> !!i, h16 = c->lines16 >> !!i;
        hipLaunchKernelGGL(k_synth_plane, dim3((w16 / 4 + 255) / 256, h16, c->batch), dim3(256), 0, c->stream, pic->plane[i], i ? c->bs_c : c->bs_y, st, i, w, h, w16, h16,
                           d.width, d.height, t0, t_stride);
    }
    HIPCHK(hipGetLastError());
    return 0;
}
// host I420 -> device planes like x264hip_picture_upload, but asynchronous: the caller's buffers must be pinned (x264hip_host_alloc) and stay
// untouched until the context's stream has passed this point (x264hip_sync / an event): frame ingest overlapped with the sweep of the frame before
extern "C" int x264hip_picture_upload_async(x264hip_frame_ctx *c, x264hip_picture *pic, const uint8_t *y, int sy,
                                            const uint8_t *u, int su, const uint8_t *v, int sv, void *hip_stream)
{
    const x264hip_frame_dims &d = c->d;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const u8 *src[3] = {y, u, v};
    const int ss[3] = {sy, su, sv};
    for (int i = 0; i < 3; i++) {
        int w = d.width >> !!i, h = d.height >> !!i, st = i ? d.stride_c : d.stride_y;
        u8 *dst = pic->plane[i] + (i ? c->bs_c : c->bs_y) * c->sel;
        HIPCHK(hipMemcpy2DAsync(dst, st, src[i], ss[i], w, h, hipMemcpyHostToDevice, s));
        int w16 = c->width16 >> !!i, h16 = c->lines16 >> !!i;
        if (w16 != w || h16 != h)
            hipLaunchKernelGGL(k_pad_mod16, dim3((w16 + 255) / 256, h16), dim3(256), 0, s, dst, st, w, h, w16, h16);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int plane_geometry(const x264hip_frame_ctx *c, const x264hip_picture *pic, int id, u8 **p, int *stride, int *w, int *h, int *padh, int *padv)
{
    const x264hip_frame_dims &d = c->d;
    if (id == 0)      { *p = pic->plane[0] + c->bs_y * c->sel; *stride = d.stride_y; *w = c->width16; *h = c->lines16; *padh = PADH; *padv = PADV; }
    else if (id < 3)  { *p = pic->plane[id] + c->bs_c * c->sel; *stride = d.stride_c; *w = c->width16 / 2; *h = c->lines16 / 2; *padh = PADH / 2; *padv = PADV / 2; }
    else if (id < 6)  { *p = pic->filtered[id - 2] + c->bs_y * c->sel; *stride = d.stride_y; *w = c->width16; *h = c->lines16; *padh = PADH; *padv = PADV; }
    else if (id < 10) { *p = pic->lowres[id - 6] + c->bs_l * c->sel; *stride = pic->stride_lowres; *w = pic->width_lowres; *h = pic->lines_lowres; *padh = PADH; *padv = PADV; }
    else return -1;
    return 0;
}

extern "C" int x264hip_picture_download(x264hip_frame_ctx *c, const x264hip_picture *pic, int plane_id,
                                        uint8_t *dst, int dst_stride, int with_padding)
{
    u8 *p; int stride, w, h, padh, padv;
    if (plane_geometry(c, pic, plane_id, &p, &stride, &w, &h, &padh, &padv)) { set_error("bad plane id %d", plane_id); return -1; }
    if (with_padding) {
        // the stored row is `stride` wide; hand back the whole padded image
        p -= (size_t)stride * padv + padh; w = stride; h += 2 * padv;
    }
    HIPCHK(hipMemcpy2DAsync(dst, dst_stride, p, stride, w, h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

static void launch_expand(const x264hip_frame_ctx *c, u8 *pix, size_t bs, int stride, int width, int height, int padh, int padv)
{
    int nq = (width + 2 * padh + 3) / 4;
    hipLaunchKernelGGL(k_expand_border, dim3((nq + 255) / 256, height + 2 * padv, c->elems ? c->n_elems : c->batch), dim3(256), 0, c->stream, pix, bs, stride, width, height, padh, padv, c->elems);
}

extern "C" int x264hip_expand_border(x264hip_frame_ctx *c, x264hip_picture *pic, int which)
{
    const x264hip_frame_dims &d = c->d;
    if (which == 0) {           // x264_frame_expand_border, R/common/frame.c:242-270
        launch_expand(c, pic->plane[0], c->bs_y, d.stride_y, c->width16, c->lines16, PADH, PADV);
        for (int i = 1; i < 3; i++) launch_expand(c, pic->plane[i], c->bs_c, d.stride_c, c->width16 / 2, c->lines16 / 2, PADH / 2, PADV / 2);
    } else if (which == 1) {    // x264_frame_expand_border_filtered, frame.c:272-296: image = cols [-4,w+4) rows [-8,h+8)
        for (int i = 1; i < 4; i++)
            launch_expand(c, pic->filtered[i] - 8 * (ptrdiff_t)d.stride_y - 4, c->bs_y, d.stride_y, c->width16 + 8, c->lines16 + 16, PADH - 4, PADV - 8);
    } else if (which == 2) {    // x264_frame_expand_border_lowres, frame.c:298-301 (width = stride - 2*PADH)
        for (int i = 0; i < 4; i++)
            launch_expand(c, pic->lowres[i], c->bs_l, pic->stride_lowres, pic->stride_lowres - 2 * PADH, pic->lines_lowres, PADH, PADV);
    } else { set_error("bad border kind %d", which); return -1; }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int x264hip_hpel_filter_frame(x264hip_frame_ctx *c, x264hip_picture *pic)
{
    const x264hip_frame_dims &d = c->d;
    // region kept by the reference after border expansion: x in [-4, w+4), y in [-8, h+8)
    int nx = c->width16 + 8, ny = c->lines16 + 16;
    dim3 grid((nx + HP_TW - 1) / HP_TW, (ny + HP_TH - 1) / HP_TH, c->elems ? c->n_elems : c->batch);
    hipLaunchKernelGGL(k_hpel, grid, dim3(256), 0, c->stream, pic->plane[0], pic->filtered[1], pic->filtered[2], pic->filtered[3],
                       c->bs_y, d.stride_y, -4, -8, nx, ny, c->elems);
    HIPCHK(hipGetLastError());
    return x264hip_expand_border(c, pic, 1);
}

extern "C" int x264hip_lowres_init_frame(x264hip_frame_ctx *c, x264hip_picture *pic)
{
    const x264hip_frame_dims &d = c->d;
    hipLaunchKernelGGL(k_dup_edge, dim3((c->lines16 + 255) / 256, 1, c->batch), dim3(256), 0, c->stream, pic->plane[0], c->bs_y, d.stride_y, c->width16, c->lines16);
    hipLaunchKernelGGL(k_dup_row, dim3((c->width16 + 255) / 256, 1, c->batch), dim3(256), 0, c->stream, pic->plane[0], c->bs_y, d.stride_y, c->width16, c->lines16);
    int w = pic->width_lowres, h = pic->lines_lowres;
    hipLaunchKernelGGL(k_lowres, dim3(((w + 3) / 4 + 255) / 256, h, c->batch), dim3(256), 0, c->stream, pic->plane[0], pic->lowres[0], pic->lowres[1],
                       pic->lowres[2], pic->lowres[3], c->bs_y, c->bs_l, d.stride_y, pic->stride_lowres, w, h);
    HIPCHK(hipGetLastError());
    return x264hip_expand_border(c, pic, 2);
}

extern "C" int x264hip_aq_var_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, int32_t *out_dev)
{
    const x264hip_frame_dims &d = c->d;
    int n = d.mb_w * d.mb_h;
    hipLaunchKernelGGL(k_aq_var, dim3((n + 3) / 4, 1, c->batch), dim3(256), 0, c->stream, pic->plane[0], pic->plane[1], pic->plane[2],
                       c->bs_y, c->bs_c, d.stride_y, d.stride_c, d.mb_w, n, out_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

// x264_adaptive_quant_frame (R/encoder/ratecontrol.c:231-249): fenc->f_qp_offset of every macroblock from its AC energy.  The
// float arithmetic is the reference's, operation for operation (fp32 subtract / add / multiply are exactly rounded on both sides;
// the library is built with -ffp-contract=off).
static __device__ const float d_log2_lut[128] = {
    0.00000, 0.01123, 0.02237, 0.03342, 0.04439, 0.05528, 0.06609, 0.07682, 0.08746, 0.09803, 0.10852, 0.11894, 0.12928, 0.13955, 0.14975, 0.15987,
    0.16993, 0.17991, 0.18982, 0.19967, 0.20945, 0.21917, 0.22882, 0.23840, 0.24793, 0.25739, 0.26679, 0.27612, 0.28540, 0.29462, 0.30378, 0.31288,
    0.32193, 0.33092, 0.33985, 0.34873, 0.35755, 0.36632, 0.37504, 0.38370, 0.39232, 0.40088, 0.40939, 0.41785, 0.42626, 0.43463, 0.44294, 0.45121,
    0.45943, 0.46761, 0.47573, 0.48382, 0.49185, 0.49985, 0.50779, 0.51570, 0.52356, 0.53138, 0.53916, 0.54689, 0.55459, 0.56224, 0.56986, 0.57743,
    0.58496, 0.59246, 0.59991, 0.60733, 0.61471, 0.62205, 0.62936, 0.63662, 0.64386, 0.65105, 0.65821, 0.66534, 0.67243, 0.67948, 0.68650, 0.69349,
    0.70044, 0.70736, 0.71425, 0.72110, 0.72792, 0.73471, 0.74147, 0.74819, 0.75489, 0.76155, 0.76818, 0.77479, 0.78136, 0.78790, 0.79442, 0.80090,
    0.80735, 0.81378, 0.82018, 0.82655, 0.83289, 0.83920, 0.84549, 0.85175, 0.85798, 0.86419, 0.87036, 0.87652, 0.88264, 0.88874, 0.89482, 0.90087,
    0.90689, 0.91289, 0.91886, 0.92481, 0.93074, 0.93664, 0.94251, 0.94837, 0.95420, 0.96000, 0.96578, 0.97154, 0.97728, 0.98299, 0.98868, 0.99435};
__global__ __launch_bounds__(256) void k_aq_offset(const int *__restrict__ energy, float *__restrict__ out, int n, float strength)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 e = (u32)energy[i];
    const int lz = __clz((int)e);
    out[i] = strength * (d_log2_lut[(e << lz >> 24) & 0x7f] - lz + 16.573f);
}
extern "C" int x264hip_adaptive_quant_frame(x264hip_frame_ctx *c, const x264hip_picture *pic, float aq_strength, int32_t *energy_dev, float *offset_dev)
{
    if (x264hip_aq_var_frame(c, pic, energy_dev)) return -1;
    const int n = c->d.mb_w * c->d.mb_h * c->batch;
    const float strength = aq_strength * 1.0397;                  // ratecontrol.c:235 (float * double, rounded to float)
    hipLaunchKernelGGL(k_aq_offset, dim3((n + 255) / 256), dim3(256), 0, c->stream, energy_dev, offset_dev, n, strength);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int x264hip_ssd_frame_async(x264hip_frame_ctx *c, const x264hip_picture *a, const x264hip_picture *b, uint64_t *ssd_dev)
{
    const x264hip_frame_dims &d = c->d;
    HIPCHK(hipMemsetAsync(ssd_dev, 0, 24 * (size_t)c->batch, c->stream));
    for (int i = 0; i < 3; i++) {
        int w = d.width >> !!i, h = d.height >> !!i, st = i ? d.stride_c : d.stride_y;
        hipLaunchKernelGGL(k_ssd, dim3(4, h < 64 ? h : 64, c->batch), dim3(256), 0, c->stream, a->plane[i], st, b->plane[i], st,
                           i ? c->bs_c : c->bs_y, w, h, (unsigned long long *)ssd_dev + i);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// blocking form for the selected batch element
extern "C" int x264hip_ssd_frame(x264hip_frame_ctx *c, const x264hip_picture *a, const x264hip_picture *b, int64_t ssd_host[3])
{
    if (x264hip_ssd_frame_async(c, a, b, (uint64_t *)c->ssd_dev)) return -1;
    HIPCHK(hipMemcpyAsync(ssd_host, c->ssd_dev + 3 * c->sel, 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
