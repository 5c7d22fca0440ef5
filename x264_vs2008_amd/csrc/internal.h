// internal.h -- host-side state shared by the translation units of libx264hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/x264hip.h"

namespace x264hip {

void set_error(const char *fmt, ...);
bool initialised();
int  device_id();
size_t arena_bytes();

#define HIPCHK(expr)                                                                 \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            ::x264hip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                               \
        }                                                                            \
    } while (0)

// Per-thread staging for table-level calls: a pinned, device-mapped arena the
// kernel reads operands from and writes results to (no explicit copies), and
// a private stream so concurrent frame threads never serialise on each other.
struct ThreadCtx {
    hipStream_t stream = nullptr;
    uint8_t *host = nullptr;     // pinned host view
    uint8_t *dev = nullptr;      // device view of the same memory
    size_t cap = 0, top = 0;
    bool ok = false;
    ~ThreadCtx();
};
ThreadCtx *thread_ctx();         // lazily created; aborts loudly if HIP is unusable

// device fill with zeros in pieces of at most 1 GiB (one fill kernel over several GiB is what rocprofv3 --pmc of ROCm 7.2 was seen to die in)
int zero_async(void *dev, size_t bytes, hipStream_t s);

}  // namespace x264hip
