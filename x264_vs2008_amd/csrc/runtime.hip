// runtime.hip -- library lifecycle: device selection, error string, per-thread
// stream + pinned staging arena for the table-level entries.
//
// Everything that can fail happens here, at x264hip_init() time, mirroring the
// reference's rule that table entries have no error channel and only
// x264_encoder_open may fail (R/encoder/encoder.c:634-645).
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "internal.h"
#include <math.h>

namespace x264hip {

static std::mutex g_mu;
static char g_err[512] = "";
static std::atomic<bool> g_init{false};
static int g_device = 0;
static size_t g_arena = 4u << 20;

void set_error(const char *fmt, ...)
{
    std::lock_guard<std::mutex> lk(g_mu);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
bool initialised() { return g_init.load(); }
int device_id() { return g_device; }
size_t arena_bytes() { return g_arena; }

ThreadCtx::~ThreadCtx()
{
    // Process teardown order between thread_local destructors and the HIP
    // runtime is unspecified; leak rather than call into a dead runtime.
}

ThreadCtx *thread_ctx()
{
    static thread_local ThreadCtx ctx;
    if (ctx.ok) return &ctx;
    if (!initialised()) {
        fprintf(stderr, "x264hip: table entry called before x264hip_init() succeeded\n");
        abort();
    }
    hipError_t e = hipSetDevice(g_device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx.host, g_arena, hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx.dev, ctx.host, 0);
    if (e != hipSuccess) {
        // No CPU fallback by design: a table entry that cannot reach the GPU
        // must not silently compute something else.
        fprintf(stderr, "x264hip: cannot create per-thread HIP context: %s\n", hipGetErrorString(e));
        abort();
    }
    ctx.cap = g_arena;
    ctx.ok = true;
    return &ctx;
}

int zero_async(void *dev, size_t bytes, hipStream_t s)
{
    const size_t piece = (size_t)1 << 30;
    for (size_t o = 0; o < bytes; o += piece)
        HIPCHK(hipMemsetAsync((char *)dev + o, 0, bytes - o < piece ? bytes - o : piece, s));
    return 0;
}

}  // namespace x264hip

using namespace x264hip;

extern "C" int x264hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int x264hip_init(const x264hip_cfg *cfg)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return -1;
    }
    int dev = cfg ? cfg->device : 0;
    if (dev < 0 || dev >= n) {
        set_error("device %d out of range (0..%d)", dev, n - 1);
        return -2;
    }
    HIPCHK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (prop.warpSize != 64) {
        set_error("device %d has wavefront size %d; this library is written for wave64 gfx950", dev, prop.warpSize);
        return -3;
    }
    g_device = dev;
    if (cfg && cfg->arena_bytes) g_arena = cfg->arena_bytes;
    g_init.store(true);
    set_error("");
    return 0;
}

extern "C" void x264hip_shutdown(void)
{
    if (!g_init.load()) return;
    (void)hipDeviceSynchronize();
    g_init.store(false);
}

extern "C" const char *x264hip_last_error(void) { return g_err; }

// ---- plain device-memory helpers for hosts that have no HIP binding of their own ----
extern "C" void *x264hip_malloc(size_t bytes)
{
    void *p = nullptr;
    if (!initialised() || hipSetDevice(g_device) != hipSuccess) return nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { set_error("hipMalloc(%zu) failed", bytes); return nullptr; }
    (void)zero_async(p, bytes, nullptr);
    (void)hipDeviceSynchronize();      // the frame contexts' streams are non-blocking: nothing may touch the buffer before the clear has landed
    return p;
}

// x264_nal_encode (R/common/common.c:656-693): Annex B start code (always the long one) or nothing, the NAL header byte, and the payload
// with an emulation-prevention 0x03 before every byte <= 3 that follows two zero bytes.  Host code, like the reference's: the slice
// payloads come back from the device once per frame (x264hip_slice_rd.payload).  dst needs 5 + len * 3 / 2 bytes at worst.
extern "C" int x264hip_nal_encode(uint8_t *dst, int b_annexb, int i_ref_idc, int i_type, const uint8_t *payload, int len)
{
    uint8_t *d = dst;
    int zeros = 0;
    if (b_annexb) { *d++ = 0; *d++ = 0; *d++ = 0; *d++ = 1; }
    *d++ = (uint8_t)((i_ref_idc << 5) | i_type);
    for (int i = 0; i < len; i++) {
        const uint8_t v = payload[i];
        if (zeros == 2 && v <= 3) { *d++ = 3; zeros = 0; }
        zeros = v ? 0 : zeros + 1;
        *d++ = v;
    }
    return (int)(d - dst);
}

// ---- host tables whose arithmetic is floating point in the reference: built here, in C, with the reference's expression and the
// build's -ffp-contract=off (a NumPy twin of this lives in x264_vs2008_amd/frame.py only as a cross-check) ----
// p_cost_mv (x264_mb_analyse_load_costs, R/encoder/analyse.c:182-198; its log2f is the macro of analyse.c:40): out[span + i] =
// out[span - i] = (int16)(lambda * (log2f(i + 1) * 2 + 0.718f + !!i) + .5f), i = 0 .. span
extern "C" void x264hip_cost_mv_table(int lambda, int span, int16_t *out)
{
    for (int i = 0; i <= span; i++)
        out[span - i] = out[span + i] = (int16_t)(lambda * (((float)log((double)(i + 1))) / (log((double)2)) * 2 + 0.718f + !!i) + .5f);
}
// h->unquant4_mf / unquant8_mf (x264_cqm_init, R/common/set.c:146,158) from the quantiser multipliers BEFORE their qp/6 shift:
// quant_mf6 [n_cat][6][n] (= DIV(def_quant * 16, scaling_list)) -> out [n_cat][52][n]
extern "C" void x264hip_unquant_table(const int32_t *quant_mf6, int n_cat, int n, int32_t *out)
{
    for (int c = 0; c < n_cat; c++)
        for (int q = 0; q < 52; q++)
            for (int i = 0; i < n; i++)
                out[((size_t)c * 52 + q) * n + i] = (int32_t)((1ULL << (q / 6 + (n == 64 ? 16 : 15) + 8)) / (uint64_t)quant_mf6[((size_t)c * 6 + q % 6) * n + i]);
}

// x264_cqm_init (R/common/set.c:68-168): the quantiser / dequantiser / unquant tables of every QP from the six scaling lists of the PPS
// (4x4 intra Y, inter Y, intra C, inter C; 8x8 intra Y, inter Y -- raster order, NULL = flat 16) and the two luma dead zones
// (param.analyse.i_luma_deadzone: {inter, intra}, default {21, 11}).  Integer arithmetic only.  Returns 0, or -1 with the error string
// set when a multiplier overflows 16 bits at a QP >= qp_min ("Quantization overflow", set.c:160-166): such a matrix set needs a larger qp_min.
extern "C" int x264hip_cqm_init(const uint8_t *const scaling_list[6], const int luma_deadzone[2], int qp_min, x264hip_cqm_tables *t)
{
    static const int dequant4_scale[6][3] = {{10, 13, 16}, {11, 14, 18}, {13, 16, 20}, {14, 18, 23}, {16, 20, 25}, {18, 23, 29}};
    static const int quant4_scale[6][3] = {{13107, 8066, 5243}, {11916, 7490, 4660}, {10082, 6554, 4194}, {9362, 5825, 3647}, {8192, 5243, 3355}, {7282, 4559, 2893}};
    static const int quant8_scan[16] = {0, 3, 4, 3, 3, 1, 5, 1, 4, 5, 2, 5, 3, 1, 5, 1};
    static const int dequant8_scale[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                                             {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
    static const int quant8_scale[6][6] = {{13107, 11428, 20972, 12222, 16777, 15481}, {11916, 10826, 19174, 11058, 14980, 14290},
                                           {10082, 8943, 15978, 9675, 12710, 11985}, {9362, 8228, 14913, 8931, 11984, 11259},
                                           {8192, 7346, 13159, 7740, 10486, 9777}, {7282, 6428, 11570, 6830, 9118, 8640}};
    const int dz_inter = luma_deadzone ? luma_deadzone[0] : 21, dz_intra = luma_deadzone ? luma_deadzone[1] : 11;
    const int deadzone[4] = {32 - dz_intra, 32 - dz_inter, 32 - 11, 32 - 21};
    auto sl = [&](int l, int i) -> int { return scaling_list && scaling_list[l] ? scaling_list[l][i] : 16; };
    auto div_round = [](int n, int d) { return (n + (d >> 1)) / d; };
    auto shift_round = [](int x, int sh) { return sh < 0 ? x << -sh : sh == 0 ? x : (x + (1 << (sh - 1))) >> sh; };
    static int q4[4][6][16], q8[2][6][64];
    int max_qp_err = -1;
    for (int l = 0; l < 6; l++)
        for (int i = 0; i < (l < 4 ? 16 : 64); i++)
            if (sl(l, i) < 1) { set_error("cqm_init: scaling list %d has a zero entry", l); return -1; }
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    for (int q = 0; q < 6; q++) {
        for (int l = 0; l < 4; l++)
            for (int i = 0; i < 16; i++) {
                const int j = (i & 1) + ((i >> 2) & 1);
                t->dequant4_mf[l][q][i] = dequant4_scale[q][j] * sl(l, i);
                q4[l][q][i] = div_round(quant4_scale[q][j] * 16, sl(l, i));
            }
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < 64; i++) {
                const int j = quant8_scan[((i >> 1) & 12) | (i & 3)];
                t->dequant8_mf[l][q][i] = dequant8_scale[q][j] * sl(4 + l, i);
                q8[l][q][i] = div_round(quant8_scale[q][j] * 16, sl(4 + l, i));
            }
    }
    for (int q = 0; q < 52; q++) {
        for (int l = 0; l < 4; l++)
            for (int i = 0; i < 16; i++) {
                const int j = shift_round(q4[l][q % 6][i], q / 6 - 1), b = div_round(deadzone[l] << 10, j), m = (1 << 15) / j;
                t->unquant4_mf[l][q][i] = (int32_t)((1ULL << (q / 6 + 15 + 8)) / (uint64_t)q4[l][q % 6][i]);
                t->quant4_mf[l][q][i] = (uint16_t)j;
                t->quant4_bias[l][q][i] = (uint16_t)(b < m ? b : m);
                if (j > 0xffff && q > max_qp_err) max_qp_err = q;
            }
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < 64; i++) {
                const int j = shift_round(q8[l][q % 6][i], q / 6), b = div_round(deadzone[l] << 10, j), m = (1 << 15) / j;
                t->unquant8_mf[l][q][i] = (int32_t)((1ULL << (q / 6 + 16 + 8)) / (uint64_t)q8[l][q % 6][i]);
                t->quant8_mf[l][q][i] = (uint16_t)j;
                t->quant8_bias[l][q][i] = (uint16_t)(b < m ? b : m);
                if (j > 0xffff && q > max_qp_err) max_qp_err = q;
            }
    }
    if (max_qp_err >= qp_min) { set_error("cqm_init: quantisation overflow -- these matrices need QP >= %d, qp_min is %d", max_qp_err + 1, qp_min); return -1; }
    return 0;
}
extern "C" void x264hip_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int x264hip_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes)
{
    HIPCHK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int x264hip_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes)
{
    HIPCHK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}
// pinned host memory and asynchronous copies on a caller-named stream: frame ingest / payload egress that overlaps the kernels
// (the staging x264's muxers do with malloc'ed buffers, R/muxers.c:63-130; here the DMA engines read / write them directly)
extern "C" void *x264hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (!initialised() || hipSetDevice(g_device) != hipSuccess) return nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}
extern "C" void x264hip_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int x264hip_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t bytes, void *hip_stream)
{
    HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    return 0;
}
extern "C" int x264hip_memcpy_h2d_async(void *dst_dev, const void *src_host, size_t bytes, void *hip_stream)
{
    HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
    return 0;
}
// A stream whose wavefronts are dispatched ahead of those of ordinary streams when a slot frees (hipStreamCreateWithPriority, the device's
// greatest priority): for short kernels that must get through while long ones keep the device full (include/x264hip_lookahead.h)
extern "C" void *x264hip_stream_create_high_priority(void)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return nullptr;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) != hipSuccess) return nullptr;
    return (void *)s;
}
// A stream whose kernels run on compute units [first, first + n) only (hipExtStreamCreateWithCUMask; bit i of the mask = compute unit i
// in the runtime's numbering): to give two kinds of long kernels their own parts of the device (include/x264hip_lookahead.h)
extern "C" void *x264hip_stream_create_cu_range(int first, int n)
{
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (first < 0 || n <= 0 || first + n > 256) return nullptr;
    for (int i = first; i < first + n; i++) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (void *)s;
}
// 1: everything the event recorded has finished, 0: not yet, < 0: error (include/x264hip_lookahead.h)
extern "C" int x264hip_event_query(void *ev)
{
    hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }      // "not ready" must not stay behind as the thread's last error
    set_error("hipEventQuery: %s", hipGetErrorString(e));
    return -1;
}
extern "C" int x264hip_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    HIPCHK(hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}
extern "C" void *x264hip_stream_create(void)
{
    hipStream_t s = nullptr;
    if (!initialised() || hipSetDevice(g_device) != hipSuccess) return nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return (void *)s;
}
extern "C" void x264hip_stream_destroy(void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
extern "C" int x264hip_stream_synchronize(void *s)
{
    HIPCHK(hipStreamSynchronize((hipStream_t)s));
    return 0;
}
extern "C" int x264hip_device_synchronize(void)
{
    HIPCHK(hipDeviceSynchronize());
    return 0;
}

// ---- HIP events on a caller-named stream (bench.py times kernels on the stream they run on) ----
extern "C" void *x264hip_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void *)e;
}
extern "C" void x264hip_event_destroy(void *e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
extern "C" int x264hip_event_record(void *e, void *stream)
{
    HIPCHK(hipEventRecord((hipEvent_t)e, (hipStream_t)stream));
    return 0;
}
// everything enqueued on `stream` after this call runs after the work the event recorded (another stream's): how the B frames of a
// mini-GOP, each on a stream of its own, are ordered behind the anchor they predict from (x264_vs2008_amd/slice.py: lanes)
extern "C" int x264hip_stream_wait_event(void *stream, void *e)
{
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e, 0));
    return 0;
}
extern "C" float x264hip_event_elapsed_ms(void *start, void *stop)
{
    float ms = -1.0f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.0f;
    return ms;
}
