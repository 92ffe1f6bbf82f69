// COO -> CSR on the device: the on-disk formats of the reference are coordinate triplets
// (scipy COO npz, generate_data.py:109; OpenFOAM `i,j,value` dump, pEqn.H:98-108; StAn npz, data_set.py:186-188).
// Sort by (row, col) with a stable radix sort, add up duplicates in storage order, emit int32 CSR.
#include <algorithm>

#include "dpcg_internal.h"
#include "dpcg_prims.h"

using namespace dpcg;

namespace {

__global__ void k_make_keys(int64_t nnz, const int32_t *__restrict__ rows, const int32_t *__restrict__ cols,
                            uint64_t *__restrict__ keys, int32_t *__restrict__ perm, int64_t n, int *bad) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const int32_t r = rows[k], c = cols[k];
        if (r < 0 || r >= n || c < 0 || c >= n) atomicExch(bad, 1);
        keys[k] = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
        perm[k] = (int32_t)k;
    }
}

__global__ void k_flag_heads(int64_t nnz, const uint64_t *__restrict__ keys, int32_t *__restrict__ head) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride)
        head[k] = (k == 0 || keys[k] != keys[k - 1]) ? 1 : 0;
}

// one thread per run head: add the run's values in storage order, write the unique entry, count its row
__global__ void k_emit_unique(int64_t nnz, const uint64_t *__restrict__ keys, const int32_t *__restrict__ perm,
                              const int32_t *__restrict__ head, const int32_t *__restrict__ pos,
                              const double *__restrict__ vals, int32_t *__restrict__ col_out,
                              double *__restrict__ val_out, int32_t *__restrict__ row_count, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        if (!head[k]) continue;
        const uint64_t key = keys[k];
        double s = vals[perm[k]];
        for (int64_t j = k + 1; j < nnz && keys[j] == key; ++j) s += vals[perm[j]];
        const int32_t p = pos[k];
        col_out[p] = (int32_t)(key & 0xffffffffu);
        val_out[p] = s;
        // an out-of-range row (flagged in `bad` by k_make_keys; the call fails after this kernel) must not be counted:
        // row_count has n + 1 entries.  Negative rows are huge as unsigned.
        const uint32_t row = (uint32_t)(key >> 32);
        if (row < (uint32_t)n) atomicAdd(&row_count[row + 1], 1);
    }
}

template <typename T>
int alloc(T **p, int64_t count) {
    *p = nullptr;
    if (cached_alloc((void **)p, (size_t)(count > 0 ? count : 1) * sizeof(T)) != hipSuccess) {
        set_error("dpcg_coo_to_csr: hipMalloc failed");
        return DPCG_ERR_NOMEM;
    }
    return DPCG_OK;
}

}  // namespace

extern "C" int dpcg_coo_to_csr(int64_t n, int64_t nnz, const int32_t *rows, const int32_t *cols, const double *vals,
                               int32_t *rowptr, int32_t *col_out, double *val_out, int64_t *nnz_out,
                               dpcg_stream_t stream) {
    if (n <= 0 || nnz < 0 || !rowptr || !nnz_out || (nnz > 0 && (!rows || !cols || !vals || !col_out || !val_out))) {
        set_error("dpcg_coo_to_csr: bad arguments");
        return DPCG_ERR_INVALID;
    }
    if (nnz > 2147483647LL || n > 2147483646LL) {
        set_error("dpcg_coo_to_csr: int32 limits exceeded");
        return DPCG_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s);
    uint64_t *keys = nullptr, *keys_sorted = nullptr;
    int32_t *perm = nullptr, *perm_sorted = nullptr, *head = nullptr, *pos = nullptr, *row_count = nullptr;
    int *bad = nullptr;
    int st = DPCG_OK;
    auto cleanup = [&]() {
        for (void *p : {(void *)keys, (void *)keys_sorted, (void *)perm, (void *)perm_sorted, (void *)head, (void *)pos,
                        (void *)row_count, (void *)bad})
            if (p) cached_free(p);
    };
#define COO_TRY(expr)            \
    do {                         \
        st = (expr);             \
        if (st < 0) {            \
            cleanup();           \
            return st;           \
        }                        \
    } while (0)
#define COO_HIP(call)                                                       \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            cleanup();                                                      \
            return ::dpcg::hip_fail(e_, #call, __FILE__, __LINE__);         \
        }                                                                   \
    } while (0)
    COO_TRY(alloc(&keys, nnz));
    COO_TRY(alloc(&keys_sorted, nnz));
    COO_TRY(alloc(&perm, nnz));
    COO_TRY(alloc(&perm_sorted, nnz));
    COO_TRY(alloc(&head, nnz));
    COO_TRY(alloc(&pos, nnz));
    COO_TRY(alloc(&row_count, n + 1));
    COO_TRY(alloc(&bad, 1));
    COO_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));
    COO_HIP(hipMemsetAsync(row_count, 0, (size_t)(n + 1) * sizeof(int32_t), s));
    int64_t unique = 0;
    if (nnz > 0) {
        const int grid = (int)std::min<int64_t>((nnz + 255) / 256, 4096);
        hipLaunchKernelGGL(k_make_keys, dim3(grid), dim3(256), 0, s, nnz, rows, cols, keys, perm, n, bad);
        // stable sort by (row, col): duplicates keep their storage order, so their sum is reproducible
        COO_TRY(sort_pairs_u64_i32(keys, keys_sorted, perm, perm_sorted, nnz, 32 + bits_for((uint64_t)n), s));
        hipLaunchKernelGGL(k_flag_heads, dim3(grid), dim3(256), 0, s, nnz, keys_sorted, head);
        COO_TRY(exclusive_scan_i32(head, pos, nnz, s));
        hipLaunchKernelGGL(k_emit_unique, dim3(grid), dim3(256), 0, s, nnz, keys_sorted, perm_sorted, head, pos, vals,
                           col_out, val_out, row_count, n);
        int32_t last_pos = 0, last_head = 0;
        int h_bad = 0;
        COO_HIP(hipMemcpyAsync(&last_pos, pos + nnz - 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        COO_HIP(hipMemcpyAsync(&last_head, head + nnz - 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        COO_HIP(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
        COO_HIP(hipStreamSynchronize(s));
        if (h_bad) {
            cleanup();
            set_error("dpcg_coo_to_csr: index out of range");
            return DPCG_ERR_INVALID;
        }
        unique = (int64_t)last_pos + last_head;
    }
    // rowptr = inclusive scan of the per-row counts (row_count[r+1] holds row r)
    COO_TRY(inclusive_scan_i32(row_count, rowptr, n + 1, s));
    *nnz_out = unique;
    cleanup();
    COO_HIP(hipGetLastError());
    return DPCG_OK;
#undef COO_TRY
#undef COO_HIP
}
