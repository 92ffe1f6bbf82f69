// Device memory for the setup routines: a block cache in front of hipMalloc / hipFree.
// hipFree waits for the device and costs ~140 us for any block of a megabyte or more on this stack (hipMalloc: ~10 us --
// tools/malloc_probe.py); a preconditioner setup frees dozens of temporaries and, when it replaces a preconditioner, the
// arrays of the old one: 5-7 ms of a 9 ms IC(0) setup at 1M rows.  Freed blocks are therefore kept and handed out again:
//   * inside a SetupScope (every ABI entry point that builds something on a stream opens one) a freed block is reusable by
//     the SAME scope at once -- everything the scope enqueues goes to its one stream, so the reuse is ordered behind the
//     last use -- and moves to the process-wide pool when the scope ends, after the stream has been waited for;
//   * outside a scope a free waits for the device (what hipFree does) before the block goes to the pool.
// So a block in the pool is never in use by the device, and any stream may take it.  Blocks are kept per DEVICE (a process
// may hold handles on several GPUs): a block serves requests made with its own device current, and a free that happens
// with another device current waits for the block's device.  The pool is bounded (DPCG_CACHE_MB, default 1024, over all
// devices; 0 switches the cache off); dpcg_release_cached_memory() returns it to the driver.
#include <map>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

#include "dpcg_host.h"

namespace dpcg {

namespace {
std::shared_mutex &capture_lock() {
    static std::shared_mutex *m = new std::shared_mutex();
    return *m;
}
}  // namespace

namespace {
// The lock is not recursive: a device-wide wait or a hipFree issued by the thread that holds a CaptureGuard would wait for itself.
// Nothing inside a capture allocates or frees today; should that change, the wait is refused (it would void the capture anyway)
// and the free is deferred to the end of the capture instead of deadlocking.
thread_local int tl_capture_depth = 0;
thread_local std::vector<void *> *tl_deferred_frees = nullptr;
}  // namespace

CaptureGuard::CaptureGuard() {
    if (tl_capture_depth++ == 0) capture_lock().lock_shared();
}
CaptureGuard::~CaptureGuard() {
    if (--tl_capture_depth != 0) return;
    capture_lock().unlock_shared();
    if (tl_deferred_frees) {
        std::vector<void *> *list = tl_deferred_frees;
        tl_deferred_frees = nullptr;
        for (void *p : *list) (void)device_free(p);
        delete list;
    }
}

hipError_t device_wide_wait() {
    if (tl_capture_depth > 0) return hipErrorStreamCaptureUnsupported;
    std::unique_lock<std::shared_mutex> lock(capture_lock());
    return hipDeviceSynchronize();
}

hipError_t device_free(void *p) {
    if (tl_capture_depth > 0) {
        if (!tl_deferred_frees) tl_deferred_frees = new std::vector<void *>();
        tl_deferred_frees->push_back(p);
        return hipSuccess;
    }
    std::unique_lock<std::shared_mutex> lock(capture_lock());
    return hipFree(p);
}

namespace {
struct Block {
    size_t size;
    int device;
};
struct Pool {
    std::mutex mu;
    std::unordered_map<void *, Block> info;              // every block handed out by cached_alloc and not yet returned to the driver
    std::map<int, std::multimap<size_t, void *>> free_blocks;   // per device: idle, not in use by the device
    size_t free_bytes = 0;                               // over all devices
    size_t cap = 0;
    Pool() {
        const char *e = getenv("DPCG_CACHE_MB");
        cap = (size_t)(e ? atoll(e) : 1024) << 20;
    }
};
Pool &pool() {
    static Pool *p = new Pool();                         // (never destroyed: blocks may be freed during process teardown)
    return *p;
}
thread_local SetupScope *tl_scope = nullptr;

size_t rounded(size_t bytes) {
    if (bytes == 0) bytes = 1;
    const size_t q = bytes >= ((size_t)1 << 20) ? ((size_t)64 << 10) : 512;
    return (bytes + q - 1) / q * q;
}
// a cached block serves a request when it is large enough and wastes at most a quarter
bool take_from(std::multimap<size_t, void *> &m, size_t want, void **out, size_t *got) {
    auto it = m.lower_bound(want);
    if (it == m.end() || it->first > want + want / 4 + 4096) return false;
    *out = it->second;
    *got = it->first;
    m.erase(it);
    return true;
}
int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
void pool_insert(void *p, size_t size, int device) {
    Pool &P = pool();
    std::vector<void *> evict;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        P.free_blocks[device].emplace(size, p);
        P.free_bytes += size;
        while (P.free_bytes > P.cap) {                   // over the bound: the largest blocks go back to the driver
            std::multimap<size_t, void *> *largest = nullptr;
            for (auto &kv : P.free_blocks)
                if (!kv.second.empty() && (!largest || std::prev(kv.second.end())->first > std::prev(largest->end())->first))
                    largest = &kv.second;
            if (!largest) break;
            auto it = std::prev(largest->end());
            P.free_bytes -= it->first;
            P.info.erase(it->second);
            evict.push_back(it->second);
            largest->erase(it);
        }
    }
    for (void *q : evict) (void)device_free(q);          // (hipFree finds the block's device itself)
}
}  // namespace

void release_cached_memory() {
    Pool &P = pool();
    std::vector<void *> all;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        for (auto &dev : P.free_blocks)
            for (auto &kv : dev.second) {
                P.info.erase(kv.second);
                all.push_back(kv.second);
            }
        P.free_blocks.clear();
        P.free_bytes = 0;
    }
    for (void *q : all) (void)device_free(q);
}

size_t cached_memory_bytes() {
    Pool &P = pool();
    std::lock_guard<std::mutex> lock(P.mu);
    return P.free_bytes;
}

hipError_t cached_alloc(void **out, size_t bytes) {
    Pool &P = pool();
    *out = nullptr;
    if (P.cap == 0) return hipMalloc(out, bytes ? bytes : 1);
    const size_t want = rounded(bytes);
    const int device = current_device();                 // a block serves requests on the device it was allocated on only
    size_t got = 0;
    if (tl_scope && tl_scope->device == device && take_from(tl_scope->idle, want, out, &got)) return hipSuccess;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto dev = P.free_blocks.find(device);
        if (dev != P.free_blocks.end() && take_from(dev->second, want, out, &got)) {
            P.free_bytes -= got;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, want);
    if (e == hipErrorOutOfMemory) {                      // give the cache back and try once more
        (void)hipGetLastError();
        release_cached_memory();
        e = hipMalloc(out, want);
    }
    if (e != hipSuccess) {
        *out = nullptr;
        return e;
    }
    std::lock_guard<std::mutex> lock(P.mu);
    P.info[*out] = Block{want, device};
    return hipSuccess;
}

void cached_free(void *p) {
    if (!p) return;
    Pool &P = pool();
    size_t size = 0;
    int device = 0;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.info.find(p);
        if (it != P.info.end()) {
            size = it->second.size;
            device = it->second.device;
            if (P.cap == 0 || size > P.cap / 4) {        // too large to keep
                P.info.erase(it);
                size = 0;
            }
        }
    }
    if (size == 0) {                                     // not ours to keep (or the cache is off)
        (void)device_free(p);
        return;
    }
    if (tl_scope && tl_scope->device == device) {
        tl_scope->idle.emplace(size, p);
        return;
    }
    // what hipFree would have waited for: the device the block lives on
    const int here = current_device();
    if (here != device) (void)hipSetDevice(device);
    (void)device_wide_wait();
    if (here != device) (void)hipSetDevice(here);
    pool_insert(p, size, device);
}

SetupScope::SetupScope(hipStream_t s, bool wait_for_device) : stream(s) {
    if (tl_scope) return;                                // nested: the outer scope (same thread, same call) keeps the blocks
    owner = true;
    device = current_device();                           // (the stream's device: the caller made it current)
    if (wait_for_device) (void)device_wide_wait();
    tl_scope = this;
}

SetupScope::~SetupScope() {
    if (!owner) return;
    tl_scope = nullptr;
    // everything the scope enqueued has run when this returns: the blocks are idle, and the call's results are complete
    if (hipStreamSynchronize(stream) != hipSuccess) (void)device_wide_wait();
    for (auto &kv : idle) pool_insert(kv.second, kv.first, device);
    idle.clear();
}

}  // namespace dpcg

extern "C" int dpcg_release_cached_memory(void) {
    dpcg::release_cached_memory();
    return DPCG_OK;
}
