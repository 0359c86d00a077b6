// cm2_core.hip -- error reporting and device queries of the C ABI
#include "cm2_common.h"

#include <cstdarg>
#include <cstdlib>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace cm2 {
static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- host <-> device copies of the library -------------------------------------------------------
// Every copy between host memory and the device goes through a page-locked staging buffer of the
// calling thread (1 MB, in pieces), never straight from / into the caller's pageable memory.  Why: the
// HIP runtime PINS pageable host memory for copies of 1 MB and more (GPU_PINNED_MIN_XFER_SIZE); when the
// host later frees that memory (a NumPy array of noise bands, a std::vector of plan offsets), the
// kernel driver's MMU notifier evicts the process's GPU queues and restores them some 10-30 ms later
// -- a kernel launched in that moment completes 11-36 ms late (the "stall in cm2_bd_det_mask" of
// rounds 3-4: profiles/r04_stall_probe.md; with pinning switched off in the runtime it never occurs).
// Both calls return when the copy is complete (the host buffer may be reused / is filled).
namespace {
constexpr size_t kStage = (size_t)1 << 20;
// one page-locked block per host thread, returned when the thread ends (the main thread's block is left to
// the process teardown: hipHostFree at static-destruction time races the runtime's own shutdown)
struct StageBlock {
    void *p = nullptr;
    bool main_thread = false;
    ~StageBlock()
    {
        if (p && !main_thread) (void)hipHostFree(p);
    }
};
void *stage_buffer()
{
    static const std::thread::id first = std::this_thread::get_id();      // whoever copies first: in practice main
    static thread_local StageBlock blk;
    if (!blk.p) {
        if (hipHostMalloc(&blk.p, kStage, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            blk.p = nullptr;
        }
        blk.main_thread = (std::this_thread::get_id() == first);
    }
    return blk.p;
}
}  // namespace

hipError_t upload(void *d_dst, const void *h_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    void *st = stage_buffer();
    if (!st) return hipErrorOutOfMemory;
    for (size_t o = 0; o < bytes; o += kStage) {
        const size_t n = bytes - o < kStage ? bytes - o : kStage;
        memcpy(st, static_cast<const char *>(h_src) + o, n);
        hipError_t e = hipMemcpyAsync(static_cast<char *>(d_dst) + o, st, n, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t download(void *h_dst, const void *d_src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    void *st = stage_buffer();
    if (!st) return hipErrorOutOfMemory;
    for (size_t o = 0; o < bytes; o += kStage) {
        const size_t n = bytes - o < kStage ? bytes - o : kStage;
        hipError_t e = hipMemcpyAsync(st, static_cast<const char *>(d_src) + o, n, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        memcpy(static_cast<char *>(h_dst) + o, st, n);
    }
    return hipSuccess;
}

hipError_t read_back(void *dst, const void *d_src, size_t bytes, hipStream_t stream)
{
    return download(dst, d_src, bytes, stream);
}

// ---- device memory of the library ---------------------------------------------------------------
// Plans are built from many large temporaries (hundreds of MB each) and hold GBs of lists.  The
// driver's hipMalloc / hipFree cost 0.1 .. 0.6 ms a call at these sizes and, when an allocation
// follows a large release closely (a second plan built after the first was destroyed), one
// hipMalloc was measured at 156 ms (profiles/r03_setup_slow_calls.txt).  Freed blocks are
// therefore kept per device and handed out again: a request takes the smallest cached block of at
// least its size and at most 25 % (+ 1 MB) more.  dev_free waits for the device like hipFree does,
// so a block is never reused while a kernel that was given it may still run.  The cache is capped
// (CM2_DEVICE_CACHE_MB, default: an eighth of the device memory; 0 = no caching): beyond the cap the largest cached blocks go
// back to the driver.  cm2_release_cached_memory() returns everything.
namespace {
struct DevCache {
    std::mutex lock;
    std::unordered_map<void *, std::pair<int, size_t>> live;            // block -> (device, bytes)
    std::multimap<std::pair<int, size_t>, void *> cached;               // (device, bytes) -> block
    size_t cached_bytes = 0, live_bytes = 0, cap = 0;
    int64_t hits = 0, misses = 0;
    bool cap_read = false;
};
DevCache &dev_cache()
{
    static DevCache *c = new DevCache();      // never destroyed: frees at process exit race the runtime's teardown
    return *c;
}
size_t round_size(size_t bytes)
{
    if (bytes == 0) bytes = 1;
    const size_t q = bytes < ((size_t)1 << 20) ? 512 : ((size_t)1 << 16);
    return (bytes + q - 1) / q * q;
}
}  // namespace

hipError_t dev_malloc_bytes(void **p, size_t bytes)
{
    DevCache &c = dev_cache();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t want = round_size(bytes);
    {
        std::lock_guard<std::mutex> hold(c.lock);
        if (!c.cap_read) {
            // default cap: an eighth of the device's memory (36 GB on an MI355X, 4 GB on a 32 GB card) --
            // torch's allocator and other ranks share the device and know nothing of this pool
            const char *env = getenv("CM2_DEVICE_CACHE_MB");
            if (env) {
                c.cap = (size_t)atoll(env) << 20;
            } else {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
                    (void)hipGetLastError();
                    total_b = (size_t)32 << 30;
                }
                c.cap = total_b / 8;
            }
            c.cap_read = true;
        }
        auto it = c.cached.lower_bound({dev, want});
        if (it != c.cached.end() && it->first.first == dev &&
            it->first.second <= want + want / 4 + ((size_t)1 << 20)) {
            *p = it->second;
            c.live[*p] = it->first;
            c.cached_bytes -= it->first.second;
            c.live_bytes += it->first.second;
            c.cached.erase(it);
            ++c.hits;
            return hipSuccess;
        }
    }
    e = hipMalloc(p, want);
    if (e != hipSuccess) {                    // out of memory: give the cache back and try once more
        (void)hipGetLastError();
        cm2_release_cached_memory();
        e = hipMalloc(p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();          // the failure is reported through e: do not leave it as the
            return e;                         // runtime's last error for the next CM2_LAUNCH_OK() to find
        }
    }
    std::lock_guard<std::mutex> hold(c.lock);
    c.live[*p] = {dev, want};
    c.live_bytes += want;
    ++c.misses;
    return hipSuccess;
}

hipError_t dev_free(void *p)
{
    if (!p) return hipSuccess;
    DevCache &c = dev_cache();
    std::pair<int, size_t> key;
    {
        std::lock_guard<std::mutex> hold(c.lock);
        auto it = c.live.find(p);
        if (it == c.live.end()) return hipFree(p);       // not ours (never happens inside the library)
        key = it->second;
        c.live.erase(it);
        c.live_bytes -= key.second;
        if (c.cap == 0) key.second = 0;                  // caching switched off
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != key.first) key.second = 0;   // another device is current: no
                                                                                // cheap way to wait for the block's
    if (key.second == 0) return hipFree(p);
    // what hipFree guarantees: no work that may touch the block is still running
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    std::vector<void *> evict;
    {
        std::lock_guard<std::mutex> hold(c.lock);
        c.cached.insert({key, p});
        c.cached_bytes += key.second;
        while (c.cached_bytes > c.cap && !c.cached.empty()) {
            auto big = std::prev(c.cached.end());        // (largest block of the highest device id)
            evict.push_back(big->second);
            c.cached_bytes -= big->first.second;
            c.cached.erase(big);
        }
    }
    for (void *q : evict) (void)hipFree(q);
    return hipSuccess;
}
}  // namespace cm2

extern "C" int cm2_release_cached_memory(void)
{
    cm2::DevCache &c = cm2::dev_cache();
    std::vector<void *> all;
    {
        std::lock_guard<std::mutex> hold(c.lock);
        for (auto &kv : c.cached) all.push_back(kv.second);
        c.cached.clear();
        c.cached_bytes = 0;
    }
    // every block goes back to the driver; the first failure is reported afterwards
    hipError_t first = hipSuccess;
    for (void *q : all) {
        const hipError_t e = hipFree(q);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    if (first != hipSuccess) {
        (void)hipGetLastError();
        cm2::set_error("cm2_release_cached_memory: hipFree failed: %s", hipGetErrorString(first));
        return 1;
    }
    return 0;
}

extern "C" int cm2_device_memory_info(int64_t *h_info)
{
    CM2_CHECK(h_info != nullptr, "cm2_device_memory_info: h_info is NULL");
    cm2::DevCache &c = cm2::dev_cache();
    std::lock_guard<std::mutex> hold(c.lock);
    h_info[0] = (int64_t)c.live_bytes;
    h_info[1] = (int64_t)c.cached_bytes;
    h_info[2] = c.hits;
    h_info[3] = c.misses;
    return 0;
}

// process-wide summation-order switch (cm2_set_exact_order): -1 = not set (environment decides)
namespace cm2 {
static std::atomic<int> g_exact_order{-1};
int exact_order_setting() { return g_exact_order.load(); }
}  // namespace cm2

extern "C" int cm2_set_exact_order(int on)
{
    cm2::g_exact_order.store(on < 0 ? -1 : (on ? 1 : 0));
    return 0;
}

extern "C" const char *cm2_last_error(void) { return cm2::g_err; }

extern "C" int cm2_abi_version(void) { return CM2_ABI_VERSION; }

extern "C" int cm2_device_info(int device, char *h_name, int *h_num_cu, double *h_hbm_gib)
{
    hipDeviceProp_t prop;
    CM2_HIP(hipGetDeviceProperties(&prop, device));
    if (h_name) {
        strncpy(h_name, prop.gcnArchName, 255);
        h_name[255] = 0;
    }
    if (h_num_cu) *h_num_cu = prop.multiProcessorCount;
    if (h_hbm_gib) *h_hbm_gib = (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0);
    return 0;
}
