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
    (void)hipMemset(p, 0, bytes);
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
