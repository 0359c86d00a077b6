// cm2_common.h -- shared helpers for the gfx950 kernels behind include/cosmomap2.h
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/cosmomap2.h"

namespace cm2 {

void set_error(const char *fmt, ...);
// cm2_set_exact_order: 1 = per-pixel sums in the reference's pure serial order everywhere, 0 = the
// default fixed regroupings of very long runs, -1 = not set (CM2_PT_ORDER / CM2_WEIGHTS_ORDER decide)
int exact_order_setting();

// host <-> device copies through a page-locked staging buffer of the calling thread; both return when
// the copy is complete.  Never pass the caller's pageable memory to hipMemcpy (cm2_core.hip says why).
hipError_t upload(void *d_dst, const void *h_src, size_t bytes, hipStream_t stream);
hipError_t download(void *h_dst, const void *d_src, size_t bytes, hipStream_t stream);
hipError_t read_back(void *dst, const void *d_src, size_t bytes, hipStream_t stream);

// device memory of the library (cm2_core.hip): hipMalloc / hipFree semantics, freed blocks cached
hipError_t dev_malloc_bytes(void **p, size_t bytes);
hipError_t dev_free(void *p);
template <typename T>
static inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc_bytes(reinterpret_cast<void **>(p), bytes); }

// Status codes of the C ABI (include/cosmomap2.h): CM2_ERR_HIP a HIP runtime call failed, CM2_ERR_ARGUMENT a
// CM2_CHECK on arguments / object state failed, CM2_ERR_OUT_OF_MEMORY a device (or page-locked host) allocation
// failed after the library had released its own cached blocks.  An out-of-memory failure also clears the
// runtime's sticky last error, so that the next CM2_LAUNCH_OK() of this thread does not report it again.
#define CM2_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            cm2::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),     \
                           __FILE__, __LINE__);                                        \
            if (e__ == hipErrorOutOfMemory) {                                          \
                (void)hipGetLastError();                                               \
                return CM2_ERR_OUT_OF_MEMORY;                                          \
            }                                                                          \
            return CM2_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define CM2_CHECK(cond, ...)                                                           \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            cm2::set_error(__VA_ARGS__);                                               \
            return CM2_ERR_ARGUMENT;                                                   \
        }                                                                              \
    } while (0)

// launch-error check after a kernel launch (does not synchronise)
#define CM2_LAUNCH_OK() CM2_HIP(hipGetLastError())

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // default workgroup: 4 waves
constexpr int kNumCU = 256;        // MI355X
constexpr uint32_t kInvalidSample = 0xFFFFFFFFu;

// memory-bound grid: enough workgroups to fill 256 CUs x 8, grid-stride the rest
static inline int grid_for(int64_t n, int block = kBlock, int max_blocks = kNumCU * 8)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: granted[] (one
// static array per kernel instance at the call site) remembers what each device was given, so the
// runtime is asked once per device and again only when a launch needs more.
// Application calls may come from several host threads (include/cosmomap2.h): the table is read and
// written under one lock per translation unit, so a smaller request can never overwrite a larger grant.
static inline hipError_t ensure_dynamic_lds(const void *func, size_t bytes, size_t (&granted)[64])
{
    static std::mutex mu;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> hold(mu);
    if (dev >= 0 && dev < 64 && granted[dev] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) granted[dev] = bytes;
    return e;
}

// Scratch allocation of a setup routine: freed on every exit path (the CM2_HIP / CM2_CHECK
// macros return early on failure).  keep() hands the buffer over to a longer-lived owner.
template <typename T>
struct DevTemp {
    T *p = nullptr;
    DevTemp() = default;
    DevTemp(const DevTemp &) = delete;
    DevTemp &operator=(const DevTemp &) = delete;
    ~DevTemp() { if (p) (void)dev_free(p); }
    hipError_t alloc(size_t count) { return dev_malloc(&p, sizeof(T) * (count ? count : 1)); }
    T *keep() { T *q = p; p = nullptr; return q; }
    void release() { if (p) (void)dev_free(p); p = nullptr; }
    operator T *() const { return p; }
};

// full-wave sum via DPP-friendly shuffles (64 lanes), result valid in lane 0
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum for 256-thread workgroups; result valid in thread 0.
// fixed order: lanes tree-reduced per wave, then waves 0..3 added in order.
__device__ __forceinline__ double block_sum_256(double v, double *lds4)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds4[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = ((lds4[0] + lds4[1]) + lds4[2]) + lds4[3];
    __syncthreads();
    return r;
}

}  // namespace cm2
