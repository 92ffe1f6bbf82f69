// spmv_lab: development harness for the CSR-stream SpMV kernel variants (not part of the product).
// Builds Poisson systems in HBM, runs every variant interleaved in one process (rounds x variants),
// checks each output bit-for-bit against variant 0 and prints median/min microseconds and GB/s.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/spmv_lab.hip -o tools/spmv_lab && tools/spmv_lab
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

constexpr int kBlock = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}
__device__ __forceinline__ int virtual_block() {
    const int G = gridDim.x, b = blockIdx.x;
    return (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
}

template <typename VT>
__global__ void k_gen_poisson(int dim, int64_t n, int32_t *rowptr, int32_t *col, VT *val) {
    const int64_t n2 = n * n;
    const int64_t N = dim == 2 ? n2 : n2 * n;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= N; i += stride) {
        int64_t before;
        if (dim == 2) {
            const int64_t lo_y = i < n ? i : n;
            const int64_t hi_y = i > n * (n - 1) ? i - n * (n - 1) : 0;
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

// ---------------------------------------------------------------------------------------------
// V0: the shipped kernel (round-1 first version)
// ---------------------------------------------------------------------------------------------
template <int CAP, int ABL = 0>
__global__ __launch_bounds__(kBlock) void spmv_v0(int64_t n, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  double *__restrict__ part) {
    __shared__ double prod[CAP];
    __shared__ double sh[4];
    const int t = threadIdx.x, G = gridDim.x, v = virtual_block();
    const int rb_lo = (int)(((int64_t)v * nrb) / G), rb_hi = (int)(((int64_t)(v + 1) * nrb) / G);
    double acc = 0.0;
    for (int rb = rb_lo; rb < rb_hi; ++rb) {
        const int64_t r0 = (int64_t)rb * 256, row = r0 + t;
        const int64_t rlast = (r0 + 256 < n) ? r0 + 256 : n;
        const int base = rowptr[r0];
        const int cnt = rowptr[rlast] - base;
        int rs = 0, re = 0;
        if (row < n) { rs = rowptr[row] - base; re = rowptr[row + 1] - base; }
        const int32_t *__restrict__ cb = col + base;
        const double *__restrict__ vb = val + base;
        for (int k0 = t; k0 < cnt; k0 += 4 * kBlock) {
            int c[4]; double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = k0 + u * kBlock; const int kk = k < cnt ? k : cnt - 1; c[u] = cb[kk]; a[u] = vb[kk]; }
            double xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = (ABL == 1 || ABL == 3) ? (double)c[u] : x[c[u]];
            if (ABL >= 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += a[u] * xv[u];
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int k = k0 + u * kBlock; if (k < cnt) prod[k] = a[u] * xv[u]; }
            }
        }
        if (ABL >= 2) {
            if (row < n) y[row] = acc + (double)(rs + re);
            continue;
        }
        __syncthreads();
        if (row < n) {
            double s = 0.0;
            for (int k = rs; k < re; ++k) s += prod[k];
            y[row] = s;
            acc += s * x[row];
        }
        __syncthreads();
    }
    const double tot = block_sum(acc, sh);
    if (t == 0) part[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------
// V1: software-pipelined: (col,val) of the NEXT row-block are fetched into registers while the
// current one is gathered/reduced; LDS row sums read up to 8 products in one batch; optional
// non-temporal matrix stream.  U = loads per thread per row-block (U*256 >= max nnz per block).
// ---------------------------------------------------------------------------------------------
template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_v1(int64_t n, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  double *__restrict__ part) {
    __shared__ double prod[U * kBlock];
    __shared__ double sh[4];
    const int t = threadIdx.x, G = gridDim.x, v = virtual_block();
    const int rb_lo = (int)(((int64_t)v * nrb) / G), rb_hi = (int)(((int64_t)(v + 1) * nrb) / G);
    double acc = 0.0;
    int c[U];
    double a[U];
    int base = 0, cnt = 0, rs = 0, re = 0;
    auto fetch = [&](int rb) {
        const int64_t r0 = (int64_t)rb * 256, row = r0 + t;
        const int64_t rlast = (r0 + 256 < n) ? r0 + 256 : n;
        base = rowptr[r0];
        cnt = rowptr[rlast] - base;
        rs = re = 0;
        if (row < n) { rs = rowptr[row] - base; re = rowptr[row + 1] - base; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = t + u * kBlock;
            const int kk = k < cnt ? k : (cnt > 0 ? cnt - 1 : 0);
            if (NT) {
                c[u] = __builtin_nontemporal_load(col + base + kk);
                a[u] = __builtin_nontemporal_load(val + base + kk);
            } else {
                c[u] = col[base + kk];
                a[u] = val[base + kk];
            }
        }
    };
    if (rb_lo < rb_hi) fetch(rb_lo);
    for (int rb = rb_lo; rb < rb_hi; ++rb) {
        const int64_t row = (int64_t)rb * 256 + t;
        const int cnt_cur = cnt, rs_cur = rs, re_cur = re;
        double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
        double pr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pr[u] = a[u] * xv[u];
        if (rb + 1 < rb_hi) fetch(rb + 1);   // next block's matrix stream is in flight from here on
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = t + u * kBlock;
            if (k < cnt_cur) prod[k] = pr[u];
        }
        __syncthreads();
        if (row < n) {
            const int len = re_cur - rs_cur;
            double s = 0.0;
            if (len <= 8) {
                double q[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] = prod[rs_cur + (j < len ? j : 0)];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (j < len) s += q[j];
            } else {
                for (int k = rs_cur; k < re_cur; ++k) s += prod[k];
            }
            y[row] = s;
            acc += s * x[row];
        }
        __syncthreads();
    }
    const double tot = block_sum(acc, sh);
    if (t == 0) part[blockIdx.x] = tot;
}


// ---------------------------------------------------------------------------------------------
// V3: 16-byte matrix stream.  A thread loads PAIRS of consecutive non-zeros (double2 val, int2 col)
// starting at an even element index, so every val load is a 16-B aligned dwordx4 and every col load
// an 8-B aligned dwordx2.  BS threads own BS rows per row-block; CAP = 2*U*BS products in LDS.
// ---------------------------------------------------------------------------------------------
template <int BS, int U>
__global__ __launch_bounds__(BS) void spmv_v3(int64_t n, int64_t nnz_total, const int32_t *__restrict__ rowptr,
                                              const int32_t *__restrict__ col, const double *__restrict__ val,
                                              const double *__restrict__ x, double *__restrict__ y, int nrb,
                                              double *__restrict__ part) {
    constexpr int CAP = 2 * U * BS;
    __shared__ double prod[CAP + 2];
    __shared__ double sh[BS / 64];
    const int t = threadIdx.x, G = gridDim.x, v = virtual_block();
    const int rb_lo = (int)(((int64_t)v * nrb) / G), rb_hi = (int)(((int64_t)(v + 1) * nrb) / G);
    double acc = 0.0;
    for (int rb = rb_lo; rb < rb_hi; ++rb) {
        const int64_t r0 = (int64_t)rb * BS, row = r0 + t;
        const int64_t rlast = (r0 + BS < n) ? r0 + BS : n;
        const int base = rowptr[r0];
        const int cnt = rowptr[rlast] - base;
        int rs = 0, re = 0;
        if (row < n) { rs = rowptr[row] - base; re = rowptr[row + 1] - base; }
        const int start = base & ~1;            // even element index: 16-B aligned val, 8-B aligned col
        const int lead = base - start;          // 0 or 1 element belonging to the previous block
        const int npairs = (cnt + lead + 1) >> 1;
        int2 c[U];
        double2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int j = t + u * BS;
            j = j < npairs ? j : (npairs > 0 ? npairs - 1 : 0);
            const int64_t e0 = (int64_t)start + 2 * j;
            if (e0 + 1 < nnz_total) {
                c[u] = *reinterpret_cast<const int2 *>(col + e0);
                a[u] = *reinterpret_cast<const double2 *>(val + e0);
            } else {                            // the very last element of the matrix at an even index
                const int64_t e = e0 < nnz_total ? e0 : nnz_total - 1;
                c[u] = make_int2(col[e], col[e]);
                a[u] = make_double2(val[e], 0.0);
            }
        }
        double2 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xv[u].x = x[c[u].x]; xv[u].y = x[c[u].y]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = t + u * BS;
            const int k = 2 * j - lead;         // product index of the pair's first element
            if (j < npairs) {
                if (k >= 0) prod[k] = a[u].x * xv[u].x;
                if (k + 1 < cnt) prod[k + 1] = a[u].y * xv[u].y;
            }
        }
        __syncthreads();
        if (row < n) {
            double s = 0.0;
            for (int k = rs; k < re; ++k) s += prod[k];
            y[row] = s;
            acc += s * x[row];
        }
        __syncthreads();
    }
    // block sum over BS/64 waves
    acc = wave_sum(acc);
    __syncthreads();
    if ((t & 63) == 0) sh[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int w = 0; w < BS / 64; ++w) tot += sh[w];
        part[blockIdx.x] = tot;
    }
}

// ---------------------------------------------------------------------------------------------
// V4: wave-granular CSR-stream.  Each wave64 owns 64 consecutive rows and a private 512-product LDS
// slice: no workgroup barrier in the loop (LDS operations of one wave execute in order), waves
// drift apart freely, so one wave's gather latency is covered by its neighbours' streaming.
// ---------------------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(kBlock) void spmv_v4(int64_t n, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int ngroups,
                                                  double *__restrict__ part) {
    __shared__ double prod_all[4][U * 64];
    __shared__ double sh[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    double *prod = prod_all[w];
    const int G = gridDim.x, v = virtual_block();
    const int64_t gw = (int64_t)v * 4 + w, nw = (int64_t)G * 4;     // contiguous groups per wave, XCD-contiguous
    const int g_lo = (int)((gw * ngroups) / nw), g_hi = (int)(((gw + 1) * ngroups) / nw);
    double acc = 0.0;
    for (int g = g_lo; g < g_hi; ++g) {
        const int64_t r0 = (int64_t)g * 64, row = r0 + lane;
        const int64_t rlast = (r0 + 64 < n) ? r0 + 64 : n;
        const int base = __builtin_amdgcn_readfirstlane(rowptr[r0]);
        const int cnt = __builtin_amdgcn_readfirstlane(rowptr[rlast]) - base;
        int rs = 0, re = 0;
        if (row < n) { rs = rowptr[row] - base; re = rowptr[row + 1] - base; }
        int c[U]; double a[U];
        const int last = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = lane + u * 64;
            const int kk = k < cnt ? k : last;
            c[u] = col[base + kk];
            a[u] = val[base + kk];
        }
        double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = lane + u * 64;
            if (k < cnt) prod[k] = a[u] * xv[u];
        }
        __builtin_amdgcn_wave_barrier();
        if (row < n) {
            double s = 0.0;
            for (int k = rs; k < re; ++k) s += prod[k];
            y[row] = s;
            acc += s * x[row];
        }
        __builtin_amdgcn_wave_barrier();
    }
    const double tot = block_sum(acc, sh);
    if (t == 0) part[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------
// V2: thread-per-row, no LDS (the naive scalar CSR kernel) -- a reference point.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void spmv_v2(int64_t n, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, int nrb,
                                                  double *__restrict__ part) {
    __shared__ double sh[4];
    const int t = threadIdx.x, G = gridDim.x, v = virtual_block();
    const int rb_lo = (int)(((int64_t)v * nrb) / G), rb_hi = (int)(((int64_t)(v + 1) * nrb) / G);
    double acc = 0.0;
    for (int rb = rb_lo; rb < rb_hi; ++rb) {
        const int64_t row = (int64_t)rb * 256 + t;
        if (row < n) {
            double s = 0.0;
            for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) s += val[k] * x[col[k]];
            y[row] = s;
            acc += s * x[row];
        }
    }
    const double tot = block_sum(acc, sh);
    if (t == 0) part[blockIdx.x] = tot;
}

// pure streaming reference: reads the same arrays with 16-byte loads and writes y (bandwidth ceiling
// for this byte mix, results meaningless)
__global__ __launch_bounds__(kBlock) void stream_ref(int64_t n, int64_t nnz, const int32_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ col, const double *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ y) {
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x, stride = (int64_t)gridDim.x * kBlock;
    double s = 0.0;
    const double2 *v2 = (const double2 *)val;
    for (int64_t i = tid; i < nnz / 2; i += stride) { double2 q = v2[i]; s += q.x + q.y; }
    const int4 *c4 = (const int4 *)col;
    for (int64_t i = tid; i < nnz / 4; i += stride) { int4 q = c4[i]; s += (double)(q.x ^ q.y ^ q.z ^ q.w); }
    const int4 *r4 = (const int4 *)rowptr;
    for (int64_t i = tid; i < n / 4; i += stride) { int4 q = r4[i]; s += (double)(q.x ^ q.w); }
    const double2 *x2 = (const double2 *)x;
    double2 *y2 = (double2 *)y;
    for (int64_t i = tid; i < n / 2; i += stride) { double2 q = x2[i]; y2[i] = make_double2(q.x + s, q.y); }
}

struct Variant {
    const char *name;
    void (*launch)(int grid, int64_t n, const int32_t *, const int32_t *, const double *, const double *, double *, int,
                   double *, hipStream_t);
    int grid_mult;  // blocks per CU
    int rows;       // rows per row-block
};

#define LAUNCHER(NAME, KERNEL)                                                                                       \
    static void NAME(int grid, int64_t n, const int32_t *rp, const int32_t *ci, const double *v, const double *x,   \
                     double *y, int nrb, double *part, hipStream_t s) {                                              \
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(kBlock), 0, s, n, rp, ci, v, x, y, nrb, part);                   \
    }
LAUNCHER(l_v0, (spmv_v0<2048>))
LAUNCHER(l_v0_a1, (spmv_v0<2048, 1>))
LAUNCHER(l_v0_a2, (spmv_v0<2048, 2>))
LAUNCHER(l_v0_a3, (spmv_v0<2048, 3>))
LAUNCHER(l_v1_8, (spmv_v1<8, false>))
LAUNCHER(l_v1_8nt, (spmv_v1<8, true>))
LAUNCHER(l_v2, spmv_v2)
LAUNCHER(l_v4, (spmv_v4<8>))
static int64_t g_nnz = 0;
static void l_v3_256(int grid, int64_t n, const int32_t *rp, const int32_t *ci, const double *v, const double *x,
                     double *y, int nrb, double *part, hipStream_t s) {
    hipLaunchKernelGGL((spmv_v3<256, 4>), dim3(grid), dim3(256), 0, s, n, g_nnz, rp, ci, v, x, y, nrb, part);
}
static void l_v3_512(int grid, int64_t n, const int32_t *rp, const int32_t *ci, const double *v, const double *x,
                     double *y, int nrb, double *part, hipStream_t s) {
    hipLaunchKernelGGL((spmv_v3<512, 4>), dim3(grid), dim3(512), 0, s, n, g_nnz, rp, ci, v, x, y, nrb, part);
}

int main(int argc, char **argv) {
    int rounds = argc > 1 ? atoi(argv[1]) : 15;
    struct Case { int dim; int64_t n; } cases[] = {{3, 100}, {2, 1024}, {3, 256}};
    std::vector<Variant> variants = {
        {"v0 shipped, 4 blk/CU", l_v0, 4, 256}, {"v0 shipped, 6 blk/CU", l_v0, 6, 256}, {"v0 shipped, 8 blk/CU", l_v0, 8, 256},
        {"v4 wave-stream, 4 blk/CU", l_v4, 4, 64}, {"v4 wave-stream, 6 blk/CU", l_v4, 6, 64}, {"v4 wave-stream, 8 blk/CU", l_v4, 8, 64},
    };

    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto cs : cases) {
        const int64_t n2 = cs.n * cs.n, N = cs.dim == 2 ? n2 : n2 * cs.n;
        const int64_t nnz = cs.dim == 2 ? 5 * n2 - 4 * cs.n : 7 * n2 * cs.n - 6 * n2;
        int32_t *rp, *ci;
        double *val, *x, *y, *y0, *part;
        CK(hipMalloc(&rp, (N + 1) * 4));
        CK(hipMalloc(&ci, nnz * 4));
        CK(hipMalloc(&val, nnz * 8));
        CK(hipMalloc(&x, N * 8));
        CK(hipMalloc(&y, N * 8));
        CK(hipMalloc(&y0, N * 8));
        CK(hipMalloc(&part, 4096 * 8));
        hipLaunchKernelGGL(k_gen_poisson<double>, dim3(4096), dim3(kBlock), 0, s, cs.dim, cs.n, rp, ci, val);
        std::vector<double> hx(N);
        srand(1);
        for (auto &q : hx) q = (double)rand() / RAND_MAX - 0.5;
        CK(hipMemcpy(x, hx.data(), N * 8, hipMemcpyHostToDevice));
        g_nnz = nnz;
        const double bytes = (double)nnz * 12 + (N + 1) * 4.0 + 16.0 * N;
        printf("== poisson%dd n=%ld N=%ld nnz=%ld  algorithmic bytes %.1f MB\n", cs.dim, (long)cs.n, (long)N, (long)nnz,
               bytes / 1e6);
        const int reps = N > 4000000 ? 10 : 50;
        std::vector<std::vector<float>> times(variants.size() + 1);
        std::vector<double> hy0(N), hy(N);
        for (int r = 0; r < rounds; ++r) {
            for (size_t vi = 0; vi <= variants.size(); ++vi) {
                int grid;
                int nrb = 0;
                if (vi < variants.size()) {
                    nrb = (int)((N + variants[vi].rows - 1) / variants[vi].rows);
                    grid = std::min(nrb, 256 * variants[vi].grid_mult);
                    if (grid > 8) grid -= grid % 8;
                } else grid = 2048;
                auto go = [&]() {
                    if (vi < variants.size()) variants[vi].launch(grid, N, rp, ci, val, x, y, nrb, part, s);
                    else hipLaunchKernelGGL(stream_ref, dim3(grid), dim3(kBlock), 0, s, N, nnz, rp, ci, val, x, y);
                };
                go();
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) go();
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                times[vi].push_back(ms * 1e3f / reps);
                if (r == 0 && vi < variants.size()) {
                    CK(hipMemcpy(hy.data(), y, N * 8, hipMemcpyDeviceToHost));
                    if (vi == 0) hy0 = hy;
                    else if (memcmp(hy.data(), hy0.data(), N * 8) != 0 && strncmp(variants[vi].name, "ABL", 3)) printf("   !! %s differs from v0\n", variants[vi].name);
                }
            }
        }
        for (size_t vi = 0; vi <= variants.size(); ++vi) {
            auto &tv = times[vi];
            std::sort(tv.begin(), tv.end());
            const float med = tv[tv.size() / 2], mn = tv[0];
            printf("  %-32s median %8.2f us  min %8.2f us   %7.1f GB/s (median)  %5.1f%% of 8 TB/s\n",
                   vi < variants.size() ? variants[vi].name : "stream_ref (16B loads, same bytes)", med, mn,
                   bytes / med / 1e3, bytes / med / 1e3 / 80.0);
        }
        hipFree(rp); hipFree(ci); hipFree(val); hipFree(x); hipFree(y); hipFree(y0); hipFree(part);
    }
    return 0;
}
