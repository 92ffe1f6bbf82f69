// rocPRIM-backed device-wide primitives (see dpcg_prims.h).  The one translation unit that instantiates rocPRIM
// templates, so the rest of the library compiles in seconds.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>

#include "dpcg_internal.h"
#include "dpcg_prims.h"

namespace dpcg {

namespace {
struct Scratch {
    void *p = nullptr;
    ~Scratch() {
        if (p) cached_free(p);
    }
    int reserve(size_t bytes) {
        if (cached_alloc(&p, bytes ? bytes : 1) != hipSuccess) {
            p = nullptr;
            set_error("device primitive: scratch allocation failed");
            return DPCG_ERR_NOMEM;
        }
        return DPCG_OK;
    }
};

#define PRIM_HIP(call)                                                   \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) return hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

template <typename K>
int sort_pairs(const K *keys_in, K *keys_out, const int32_t *vals_in, int32_t *vals_out, int64_t count, int end_bit,
               hipStream_t s) {
    if (count <= 0) return DPCG_OK;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > (int)(8 * sizeof(K))) end_bit = (int)(8 * sizeof(K));
    size_t bytes = 0;
    PRIM_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)count, 0u,
                                       (unsigned)end_bit, s));
    Scratch tmp;
    if (int st = tmp.reserve(bytes); st < 0) return st;
    PRIM_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)count, 0u,
                                       (unsigned)end_bit, s));
    PRIM_HIP(hipStreamSynchronize(s));   // the scratch buffer is released on return
    return DPCG_OK;
}
}  // namespace

int sort_pairs_u64_i32(const uint64_t *keys_in, uint64_t *keys_out, const int32_t *vals_in, int32_t *vals_out,
                       int64_t count, int end_bit, hipStream_t s) {
    return sort_pairs<uint64_t>(keys_in, keys_out, vals_in, vals_out, count, end_bit, s);
}

int sort_pairs_u32_i32(const uint32_t *keys_in, uint32_t *keys_out, const int32_t *vals_in, int32_t *vals_out,
                       int64_t count, int end_bit, hipStream_t s) {
    return sort_pairs<uint32_t>(keys_in, keys_out, vals_in, vals_out, count, end_bit, s);
}

int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t count, hipStream_t s) {
    if (count <= 0) return DPCG_OK;
    size_t bytes = 0;
    PRIM_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (int32_t)0, (size_t)count, rocprim::plus<int32_t>(), s));
    Scratch tmp;
    if (int st = tmp.reserve(bytes); st < 0) return st;
    PRIM_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, (int32_t)0, (size_t)count, rocprim::plus<int32_t>(), s));
    PRIM_HIP(hipStreamSynchronize(s));
    return DPCG_OK;
}

// The same scan with the caller's workspace (scan_workspace_bytes(count) bytes): no allocation, no synchronisation.
size_t scan_workspace_bytes(int64_t count) {
    size_t bytes = 0;
    if (count <= 0) return 0;
    if (rocprim::exclusive_scan(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)count,
                                rocprim::plus<int32_t>(), nullptr) != hipSuccess)
        return 0;
    return bytes;
}

int exclusive_scan_i32_ws(const int32_t *in, int32_t *out, int64_t count, void *workspace, size_t workspace_bytes, hipStream_t s) {
    if (count <= 0) return DPCG_OK;
    size_t bytes = workspace_bytes;
    PRIM_HIP(rocprim::exclusive_scan(workspace, bytes, in, out, (int32_t)0, (size_t)count, rocprim::plus<int32_t>(), s));
    return DPCG_OK;
}

int inclusive_scan_i32(const int32_t *in, int32_t *out, int64_t count, hipStream_t s) {
    if (count <= 0) return DPCG_OK;
    size_t bytes = 0;
    PRIM_HIP(rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)count, rocprim::plus<int32_t>(), s));
    Scratch tmp;
    if (int st = tmp.reserve(bytes); st < 0) return st;
    PRIM_HIP(rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)count, rocprim::plus<int32_t>(), s));
    PRIM_HIP(hipStreamSynchronize(s));
    return DPCG_OK;
}

int reduce_max_i32(const int32_t *in, int32_t *out_dev, int64_t count, hipStream_t s) {
    if (count <= 0) return DPCG_OK;
    size_t bytes = 0;
    PRIM_HIP(rocprim::reduce(nullptr, bytes, in, out_dev, (int32_t)(-2147483647 - 1), (size_t)count,
                             rocprim::maximum<int32_t>(), s));
    Scratch tmp;
    if (int st = tmp.reserve(bytes); st < 0) return st;
    PRIM_HIP(rocprim::reduce(tmp.p, bytes, in, out_dev, (int32_t)(-2147483647 - 1), (size_t)count,
                             rocprim::maximum<int32_t>(), s));
    PRIM_HIP(hipStreamSynchronize(s));
    return DPCG_OK;
}

}  // namespace dpcg
