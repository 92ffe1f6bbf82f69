// Setup and utility kernels: plan analysis, on-device Poisson generators, the batched COO SpMV of utils.py:15-43.
// Compiled with -ffp-contract=off so that a*b+c is two roundings, as in the CPU reference path (scipy/ATen CSR row
// sums, unfused torch mul+add at cg.py:79-83); in-order sums then reproduce the oracle bit for bit.
#include "dpcg_device.h"

namespace dpcg {

// ------------------------------------------------------------------------------------------------
// Setup helpers
// ------------------------------------------------------------------------------------------------
// max over row-blocks of the non-zeros in `rows_per_block` consecutive rows (stream-kernel test).
__global__ __launch_bounds__(kBlock) void k_block_nnz_max(int64_t n, const int32_t *__restrict__ rowptr,
                                                          int rows_per_block, int *out_max) {
    const int64_t nrb = (n + rows_per_block - 1) / rows_per_block;
    int m = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t rb = (int64_t)blockIdx.x * kBlock + threadIdx.x; rb < nrb; rb += stride) {
        const int64_t r0 = rb * rows_per_block;
        const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
        const int c = rowptr[r1] - rowptr[r0];
        m = c > m ? c : m;
    }
    atomicMax(out_max, m);
}

void launch_block_nnz_max(const CsrDev &A, int rows_per_block, int *out_max_dev, hipStream_t s) {
    const int64_t nrb = (A.n + rows_per_block - 1) / rows_per_block;
    int64_t g = (nrb + kBlock - 1) / kBlock;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_block_nnz_max, dim3((int)g), dim3(kBlock), 0, s, A.n, A.rowptr, rows_per_block, out_max_dev);
}

// Closed-form 5-point / 7-point Poisson CSR (kron(I,T)+kron(T,I)[+...], T = tridiag(-1,2,-1)),
// one thread per row, columns ascending.  Generated in HBM so the 256^3 systems (1.4 GB each) never
// cross PCIe.
template <typename VT>
__global__ __launch_bounds__(kBlock) void k_gen_poisson(int dim, int64_t n, int32_t *__restrict__ rowptr,
                                                        int32_t *__restrict__ col, VT *__restrict__ val) {
    const int64_t n2 = n * n;
    const int64_t N = dim == 2 ? n2 : n2 * n;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= N; i += stride) {
        // non-zeros before row i: full stencil minus the neighbours cut off by each face
        int64_t before;
        if (dim == 2) {
            const int64_t lo_y = i < n ? i : n;                         // rows with iy == 0
            const int64_t hi_y = i > n * (n - 1) ? i - n * (n - 1) : 0; // rows with iy == n-1
            before = 5 * i - lo_y - hi_y - (i + n - 1) / n - i / n;
        } else {
            const int64_t lo_z = i < n2 ? i : n2;
            const int64_t hi_z = i > n2 * (n - 1) ? i - n2 * (n - 1) : 0;
            const int64_t planes = i / n2, rem = i % n2;
            const int64_t lo_y = planes * n + (rem < n ? rem : n);
            const int64_t hi_y = planes * n + (rem > n * (n - 1) ? rem - n * (n - 1) : 0);
            before = 7 * i - lo_z - hi_z - lo_y - hi_y - (i + n - 1) / n - i / n;
        }
        rowptr[i] = (int32_t)before;
        if (i == N) continue;
        int64_t k = before;
        const int64_t ix = i % n, iy = (i / n) % n, iz = i / n2;
        if (dim == 3 && iz > 0) { col[k] = (int32_t)(i - n2); val[k++] = (VT)-1; }
        if (iy > 0) { col[k] = (int32_t)(i - n); val[k++] = (VT)-1; }
        if (ix > 0) { col[k] = (int32_t)(i - 1); val[k++] = (VT)-1; }
        col[k] = (int32_t)i;
        val[k++] = (VT)(dim == 2 ? 4 : 6);
        if (ix < n - 1) { col[k] = (int32_t)(i + 1); val[k++] = (VT)-1; }
        if (iy < n - 1) { col[k] = (int32_t)(i + n); val[k++] = (VT)-1; }
        if (dim == 3 && iz < n - 1) { col[k] = (int32_t)(i + n2); val[k++] = (VT)-1; }
    }
}

void launch_gen_poisson(int dim, int64_t n, int32_t *rowptr, int32_t *col, void *val, int val_dtype, hipStream_t s) {
    const int64_t N = dim == 2 ? n * n : n * n * n;
    int64_t g = (N + 1 + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    if (val_dtype == DPCG_F32)
        hipLaunchKernelGGL(k_gen_poisson<float>, dim3((int)g), dim3(kBlock), 0, s, dim, n, rowptr, col, (float *)val);
    else
        hipLaunchKernelGGL(k_gen_poisson<double>, dim3((int)g), dim3(kBlock), 0, s, dim, n, rowptr, col,
                           (double *)val);
}

// ------------------------------------------------------------------------------------------------
// The box's own streaming ceiling: R read units of 16 bytes per lane for every 16 bytes written -- the
// reads of one output element CONTIGUOUS in memory, like the val[] stream of an SpMV (R = 1: copy, 2: triad, 11: the
// read:write ratio of a 7-point CSR SpMV -- 90 B read, 8 B written per row) -- or only reduced (W = false: read-only).
// Contiguous slab per workgroup, slabs laid out XCD by XCD like the SpMV's row blocks, U output elements in flight per
// lane; NT: non-temporal loads and stores.  Shapes and grid chosen with tools/stream_lab.
// ------------------------------------------------------------------------------------------------
template <int R, int U, bool W, bool NT>
__global__ __launch_bounds__(kBlock) void k_stream_bench(int64_t n2, const double2 *__restrict__ in, double2 *__restrict__ out,
                                                         double *__restrict__ part) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int v = virtual_block();
    const int64_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)v * per, hi = lo + per < n2 ? lo + per : n2;
    double acc = 0.0;
    for (int64_t i0 = lo; i0 < hi; i0 += kBlock * U) {
        d2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d2 sum = {0.0, 0.0};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const d2 *p = reinterpret_cast<const d2 *>(in) + ((i0 + u * kBlock) * R + (int64_t)r * kBlock + threadIdx.x);
                sum += NT ? __builtin_nontemporal_load(p) : *p;
            }
            a[u] = sum;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * kBlock + threadIdx.x;
            if (W) {
                if (i < hi) {
                    d2 *q = reinterpret_cast<d2 *>(out) + i;
                    if (NT) __builtin_nontemporal_store(a[u], q);
                    else *q = a[u];
                }
            } else {
                acc += a[u].x + a[u].y;
            }
        }
    }
    if (!W && acc == 12345.678) part[blockIdx.x] = acc;      // keeps the loads alive; never true for the zero-filled input
}

// The same streams WALKED TOGETHER: workgroup b takes elements b, b + G, b + 2 G, ... (G = gridDim.x; with G = n2 / 256 every
// workgroup moves one 4-KiB piece: the "float4 copy" shape MI355X_MICROARCH.md quotes 6.29 TB/s for), so that at any instant the
// chip touches one contiguous window of each stream instead of G distant slabs.  Measured on this pool (tools/stream_lab,
// profiles/r04_stream_lab.txt): copy 5.3 (slabs) -> 6.3 TB/s (6.7 non-temporal), 11 : 1 5.3 -> 5.6 (6.0 non-temporal).
template <int R, bool W, bool NT>
__global__ __launch_bounds__(kBlock) void k_stream_walk(int64_t n2, const double2 *__restrict__ in, double2 *__restrict__ out,
                                                        double *__restrict__ part) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int64_t span = (int64_t)gridDim.x * kBlock;
    double acc = 0.0;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < n2; i0 += span) {
        const int64_t i = i0 + threadIdx.x;
        d2 sum = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const d2 *p = reinterpret_cast<const d2 *>(in) + (i0 * R + (int64_t)r * kBlock + threadIdx.x);
            sum += NT ? __builtin_nontemporal_load(p) : *p;
        }
        if (W) {
            if (i < n2) {
                d2 *q = reinterpret_cast<d2 *>(out) + i;
                if (NT) __builtin_nontemporal_store(sum, q);
                else *q = sum;
            }
        } else {
            acc += sum.x + sum.y;
        }
    }
    if (!W && acc == 12345.678) part[blockIdx.x] = acc;
}

// out_bytes: size of the written stream (rounded down to 16 B; the input is n_read times as long, plus one slab of slack);
// returns the bytes one launch moves (reads + writes), or -1 for an unsupported n_read
// `in_flight`: output elements in flight per lane (2 suits streams from HBM, 1 footprints inside the Infinity Cache -- tools/stream_lab)
int64_t launch_stream_bench(int n_read, bool write, bool nt, int64_t out_bytes, const double *in, double *out, double *part,
                            int grid, hipStream_t s, int in_flight) {
    const int64_t n2 = out_bytes / 16;
    const double2 *i2 = reinterpret_cast<const double2 *>(in);
    double2 *o2 = reinterpret_cast<double2 *>(out);
    if (in_flight == 0) {                 // the streams walked together (k_stream_walk); `grid` as given
#define DPCG_WALK_CASE(RV)                                                                                                      \
    case RV:                                                                                                                    \
        if (write && nt) hipLaunchKernelGGL((k_stream_walk<RV, true, true>), dim3(grid), dim3(kBlock), 0, s, n2, i2, o2, part);   \
        else if (write) hipLaunchKernelGGL((k_stream_walk<RV, true, false>), dim3(grid), dim3(kBlock), 0, s, n2, i2, o2, part);   \
        else if (nt) hipLaunchKernelGGL((k_stream_walk<RV, false, true>), dim3(grid), dim3(kBlock), 0, s, n2, i2, o2, part);      \
        else hipLaunchKernelGGL((k_stream_walk<RV, false, false>), dim3(grid), dim3(kBlock), 0, s, n2, i2, o2, part);             \
        break
        switch (n_read) {
            DPCG_WALK_CASE(1);
            DPCG_WALK_CASE(2);
            DPCG_WALK_CASE(4);
            DPCG_WALK_CASE(11);
            default: return -1;
        }
#undef DPCG_WALK_CASE
        return n2 * 16 * (n_read + (write ? 1 : 0));
    }
#define DPCG_STREAM_LAUNCH(RV, WV, NTV)                                                                                     \
    do {                                                                                                                    \
        if (in_flight == 1)                                                                                                 \
            hipLaunchKernelGGL((k_stream_bench<RV, 1, WV, NTV>), dim3(grid), dim3(kBlock), 0, s, n2, i2, o2, part);         \
        else                                                                                                                \
            hipLaunchKernelGGL((k_stream_bench<RV, 2, WV, NTV>), dim3(grid), dim3(kBlock), 0, s, n2, i2, o2, part);         \
    } while (0)
#define DPCG_STREAM_CASE(RV)                                             \
    case RV:                                                             \
        if (write && nt) DPCG_STREAM_LAUNCH(RV, true, true);             \
        else if (write) DPCG_STREAM_LAUNCH(RV, true, false);             \
        else if (nt) DPCG_STREAM_LAUNCH(RV, false, true);                \
        else DPCG_STREAM_LAUNCH(RV, false, false);                       \
        break
    switch (n_read) {
        DPCG_STREAM_CASE(1);
        DPCG_STREAM_CASE(2);
        DPCG_STREAM_CASE(4);
        DPCG_STREAM_CASE(11);
        default: return -1;
    }
#undef DPCG_STREAM_CASE
#undef DPCG_STREAM_LAUNCH
    return n2 * 16 * (n_read + (write ? 1 : 0));
}

// sparse_matvec_mul (utils.py:26-41): out[b, row] += feature * vec[b, col] over COO triples.
// One lane per triple, fp32 atomics on the output -- the same scatter-add the reference's own
// CUDA path performs (torch scatter_reduce on a GPU tensor is an atomicAdd).
__global__ __launch_bounds__(kBlock) void k_batched_coo_spmv(int64_t nnz, const int32_t *__restrict__ idx,
                                                             const float *__restrict__ feat, int batch, int64_t dof,
                                                             const float *__restrict__ vec, float *__restrict__ out,
                                                             int transpose) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride) {
        const int b = idx[3 * k];
        const int r = idx[3 * k + (transpose ? 2 : 1)];                 // utils.py:27
        const int c = idx[3 * k + (transpose ? 1 : 2)];                 // utils.py:28
        if (b < 0 || b >= batch || r < 0 || r >= dof || c < 0 || c >= dof) continue;
        atomicAdd(out + (int64_t)b * dof + r, feat[k] * vec[(int64_t)b * dof + c]);   // utils.py:32,36-41
    }
}

// Per-triple products out[k] = a[b, row_k] * c[b, col_k]: the gradient of sparse_matvec_mul with respect to the
// matrix entries (d/d feature_k of sum_b <g_b, A_b v_b> = g[b,row_k] * v[b,col_k]); needed to train through
// `frobenius_loss` (metrics.py:28-29).
__global__ __launch_bounds__(kBlock) void k_batched_coo_edge(int64_t nnz, const int32_t *__restrict__ idx, int batch,
                                                             int64_t dof, const float *__restrict__ a,
                                                             const float *__restrict__ c, float *__restrict__ out,
                                                             int transpose) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride) {
        const int b = idx[3 * k];
        const int r = idx[3 * k + (transpose ? 2 : 1)];
        const int cc = idx[3 * k + (transpose ? 1 : 2)];
        const bool ok = b >= 0 && b < batch && r >= 0 && r < dof && cc >= 0 && cc < dof;
        out[k] = ok ? a[(int64_t)b * dof + r] * c[(int64_t)b * dof + cc] : 0.0f;
    }
}

// The same two operators with `ncols` right-hand sides per sample (row-major panels B[b, row, c]): what the sparse form of
// `inverse_loss` (metrics.py:34-55) is made of -- L^T (A[:, J]) and L (.) on a panel of columns J instead of dense N x N
// products.  One lane per (triple, column): consecutive lanes take consecutive columns of a panel row (coalesced).
__global__ __launch_bounds__(kBlock) void k_batched_coo_spmm(int64_t nnz, const int32_t *__restrict__ idx,
                                                             const float *__restrict__ feat, int batch, int64_t dof,
                                                             int ncols, const float *__restrict__ B,
                                                             float *__restrict__ out, int transpose) {
    const int64_t total = nnz * ncols;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const int64_t k = t / ncols;
        const int c = (int)(t - k * ncols);
        const int b = idx[3 * k];
        const int r = idx[3 * k + (transpose ? 2 : 1)];
        const int cc = idx[3 * k + (transpose ? 1 : 2)];
        if (b < 0 || b >= batch || r < 0 || r >= dof || cc < 0 || cc >= dof) continue;
        const float f = feat[k];
        if (f == 0.0f) continue;                        // explicit zeros (the masked upper triangle of the network output)
        atomicAdd(out + ((int64_t)b * dof + r) * ncols + c, f * B[((int64_t)b * dof + cc) * ncols + c]);
    }
}

// out[k] = sum_c G[b, row_k, c] * B[b, col_k, c]: the gradient of the panel product with respect to entry k.
// One wave per triple, lanes over the columns, fixed-order wave sum.
__global__ __launch_bounds__(kBlock) void k_batched_coo_sddmm(int64_t nnz, const int32_t *__restrict__ idx, int batch,
                                                              int64_t dof, int ncols, const float *__restrict__ G,
                                                              const float *__restrict__ B, float *__restrict__ out,
                                                              int transpose) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t k = wave; k < nnz; k += nwaves) {
        const int b = idx[3 * k];
        const int r = idx[3 * k + (transpose ? 2 : 1)];
        const int cc = idx[3 * k + (transpose ? 1 : 2)];
        float acc = 0.0f;
        if (b >= 0 && b < batch && r >= 0 && r < dof && cc >= 0 && cc < dof) {
            const float *g = G + ((int64_t)b * dof + r) * ncols, *v = B + ((int64_t)b * dof + cc) * ncols;
            for (int c = lane; c < ncols; c += 64) acc += g[c] * v[c];
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) out[k] = acc;
    }
}

void launch_batched_coo_spmm(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof, int ncols,
                             const float *B, float *out, int transpose, hipStream_t s) {
    int64_t g = (nnz * ncols + kBlock - 1) / kBlock;
    if (g > 16384) g = 16384;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_batched_coo_spmm, dim3((int)g), dim3(kBlock), 0, s, nnz, indices, features, batch, dof, ncols, B,
                       out, transpose);
}

void launch_batched_coo_sddmm(int64_t nnz, const int32_t *indices, int batch, int64_t dof, int ncols, const float *G,
                              const float *B, float *out, int transpose, hipStream_t s) {
    int64_t g = (nnz * 64 + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_batched_coo_sddmm, dim3((int)g), dim3(kBlock), 0, s, nnz, indices, batch, dof, ncols, G, B, out,
                       transpose);
}

void launch_batched_coo_edge(int64_t nnz, const int32_t *indices, int batch, int64_t dof, const float *a, const float *c,
                             float *out, int transpose, hipStream_t s) {
    int64_t g = (nnz + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_batched_coo_edge, dim3((int)g), dim3(kBlock), 0, s, nnz, indices, batch, dof, a, c, out,
                       transpose);
}

void launch_batched_coo_spmv(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof,
                             const float *vectors, float *out, int transpose, hipStream_t s) {
    int64_t g = (nnz + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_batched_coo_spmv, dim3((int)g), dim3(kBlock), 0, s, nnz, indices, features, batch, dof,
                       vectors, out, transpose);
}

}  // namespace dpcg
