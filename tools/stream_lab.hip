// stream_lab: what does THIS box's HBM deliver to plain streaming kernels?  Copy / read-mostly / read-only sweeps over
// grid size, loads in flight per lane, interleaved vs contiguous read streams and non-temporal accesses, on buffers far
// beyond the 256 MiB Infinity Cache.  The best figures are the ceiling `bench.py` reports (dpcg_stream_bench uses the
// winning shape).  Development tool:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_lab.hip -o tools/stream_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int vblock() {
    const int G = gridDim.x, b = blockIdx.x;
    return (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
}

// R read units of 16 B per lane for every 16 B written (W) -- the reads of one output element are CONTIGUOUS in memory
// (one input stream R times as long as the output, like the val[] stream of an SpMV), U output elements in flight per lane.
template <int R, int U, bool W, bool NT>
__global__ __launch_bounds__(256) void k_ratio(int64_t n2, const d2 *__restrict__ in, d2 *__restrict__ out, double *part) {
    const int v = vblock();
    const int64_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)v * per, hi = lo + per < n2 ? lo + per : n2;
    double acc = 0.0;
    for (int64_t i0 = lo; i0 < hi; i0 += 256 * U) {
        d2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = i0 + u * 256 + threadIdx.x;
            i = i < hi ? i : hi - 1;
            d2 s = {0.0, 0.0};
            // block-contiguous: the WG's 256 lanes read R consecutive 4-KiB pieces
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const d2 *p = in + ((i0 + u * 256) * R + (int64_t)r * 256 + threadIdx.x);
                const d2 t = NT ? __builtin_nontemporal_load(p) : *p;
                s += t;
            }
            a[u] = s;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * 256 + threadIdx.x;
            if (W) {
                if (i < hi) {
                    if (NT) __builtin_nontemporal_store(a[u], out + i);
                    else out[i] = a[u];
                }
            } else {
                acc += a[u].x + a[u].y;
            }
        }
    }
    if (!W && acc == 12345.678) part[blockIdx.x] = acc;
}

// R separate input streams (interleaved), as dpcg_stream_bench's first version
template <int R, int U, bool W>
__global__ __launch_bounds__(256) void k_streams(int64_t n2, const d2 *__restrict__ in, d2 *__restrict__ out, double *part) {
    const int v = vblock();
    const int64_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)v * per, hi = lo + per < n2 ? lo + per : n2;
    double acc = 0.0;
    for (int64_t i0 = lo; i0 < hi; i0 += 256 * U) {
        d2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = i0 + u * 256 + threadIdx.x;
            i = i < hi ? i : hi - 1;
            d2 s = {0.0, 0.0};
#pragma unroll
            for (int r = 0; r < R; ++r) s += in[(int64_t)r * n2 + i];
            a[u] = s;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * 256 + threadIdx.x;
            if (W) { if (i < hi) out[i] = a[u]; }
            else acc += a[u].x + a[u].y;
        }
    }
    if (!W && acc == 12345.678) part[blockIdx.x] = acc;
}

// ---- the guide's shape (MI355X_MICROARCH.md: "6.29 TB/s measured (float4 copy)"): 16 B per lane, no slabs --------------------
// flat: one element per thread, one workgroup per 4 KiB (grid = n / BS).  stride: a persistent grid whose workgroups walk the
// buffer TOGETHER (element i of pass p at p * G * BS * U + ...): at any instant the chip reads one contiguous window instead
// of G distant slabs -- fewer DRAM pages open at once.
template <int BS, bool NT>
__global__ __launch_bounds__(BS) void k_copy_flat(int64_t n2, const d2 *__restrict__ in, d2 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (i < n2) {
        const d2 t = NT ? __builtin_nontemporal_load(in + i) : in[i];
        if (NT) __builtin_nontemporal_store(t, out + i);
        else out[i] = t;
    }
}
template <int BS, int U, int R, bool W, bool NT>
__global__ __launch_bounds__(BS) void k_copy_stride(int64_t n2, const d2 *__restrict__ in, d2 *__restrict__ out, double *part) {
    const int64_t span = (int64_t)gridDim.x * BS * U;
    double acc = 0.0;
    for (int64_t i0 = (int64_t)blockIdx.x * BS * U; i0 < n2; i0 += span) {
        d2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = i0 + u * BS + threadIdx.x;
            i = i < n2 ? i : n2 - 1;
            d2 sacc = {0.0, 0.0};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const d2 *p = in + ((i0 + u * BS) * R + (int64_t)r * BS + threadIdx.x);
                sacc += NT ? __builtin_nontemporal_load(p) : *p;
            }
            a[u] = sacc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * BS + threadIdx.x;
            if (W) {
                if (i < n2) {
                    if (NT) __builtin_nontemporal_store(a[u], out + i);
                    else out[i] = a[u];
                }
            } else {
                acc += a[u].x + a[u].y;
            }
        }
    }
    if (!W && acc == 12345.678) part[blockIdx.x] = acc;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int64_t out_bytes = 128ll << 20;                 // 128 MiB written; reads R times that (R = 11: 1.4 GiB)
    const int64_t n2 = out_bytes / 16;
    d2 *in, *out; double *part;
    CK(hipMalloc(&in, (size_t)out_bytes * 12)); CK(hipMalloc(&out, (size_t)out_bytes * 8)); CK(hipMalloc(&part, 65536 * 8));
    CK(hipMemset(in, 0, (size_t)out_bytes * 12)); CK(hipMemset(out, 0, (size_t)out_bytes * 8));
    auto timeit = [&](const char *name, double bytes, auto launch) {
        std::vector<float> t;
        for (int rd = 0; rd < 5; ++rd) {
            launch();
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 5);
        }
        std::sort(t.begin(), t.end());
        printf("%-58s median %8.1f us  %7.1f GB/s (best %7.1f)\n", name, t[2] * 1e3, bytes / t[2] / 1e6, bytes / t[0] / 1e6);
        CK(hipGetLastError());
    };
    char name[128];
#define RATIO(R, U, W, NT, G, N2)                                                                                         \
    do {                                                                                                                  \
        snprintf(name, sizeof name, "ratio R=%d U=%d write=%d nt=%d grid=%d out=%ld MiB", R, U, W, NT, G, (long)((N2) * 16 >> 20)); \
        timeit(name, (double)(N2) * 16 * (R + (W ? 1 : 0)),                                                                \
               [&] { hipLaunchKernelGGL((k_ratio<R, U, W, NT>), dim3(G), dim3(256), 0, s, (int64_t)(N2), in, out, part); }); \
    } while (0)
#define STREAMS(R, U, W, G, N2)                                                                                           \
    do {                                                                                                                  \
        snprintf(name, sizeof name, "streams R=%d U=%d write=%d grid=%d each=%ld MiB", R, U, W, G, (long)((N2) * 16 >> 20)); \
        timeit(name, (double)(N2) * 16 * (R + (W ? 1 : 0)),                                                                \
               [&] { hipLaunchKernelGGL((k_streams<R, U, W>), dim3(G), dim3(256), 0, s, (int64_t)(N2), in, out, part); }); \
    } while (0)
    // ---- the guide's float4-copy shapes first (1 GiB in, 1 GiB out) ----
    {
        const int64_t N2 = n2 * 8;
#define FLAT(BS, NT)                                                                                                       \
    do {                                                                                                                   \
        snprintf(name, sizeof name, "flat copy BS=%d nt=%d (one 16 B element per thread)", BS, NT);                         \
        timeit(name, (double)N2 * 32, [&] { hipLaunchKernelGGL((k_copy_flat<BS, NT>), dim3((unsigned)((N2 + BS - 1) / BS)), dim3(BS), 0, s, N2, in, out); }); \
    } while (0)
#define STRIDE(BS, U, R, W, NT, G, NN)                                                                                     \
    do {                                                                                                                   \
        snprintf(name, sizeof name, "grid-stride R=%d write=%d BS=%d U=%d nt=%d grid=%d", R, W, BS, U, NT, G);              \
        timeit(name, (double)(NN) * 16 * (R + (W ? 1 : 0)),                                                                 \
               [&] { hipLaunchKernelGGL((k_copy_stride<BS, U, R, W, NT>), dim3(G), dim3(BS), 0, s, (int64_t)(NN), in, out, part); }); \
    } while (0)
        FLAT(256, false); FLAT(512, false); FLAT(1024, false); FLAT(256, true);
        for (int g : {1024, 2048, 4096, 8192}) { STRIDE(256, 1, 1, true, false, g, N2); STRIDE(256, 2, 1, true, false, g, N2); STRIDE(256, 4, 1, true, false, g, N2); }
        STRIDE(512, 2, 1, true, false, 1024, N2); STRIDE(1024, 1, 1, true, false, 512, N2); STRIDE(1024, 2, 1, true, false, 1024, N2);
        STRIDE(256, 2, 1, true, true, 2048, N2); STRIDE(256, 4, 1, true, true, 4096, N2);
        snprintf(name, sizeof name, "hipMemcpyAsync device-to-device");
        timeit(name, (double)N2 * 32, [&] { CK(hipMemcpyAsync(out, in, (size_t)N2 * 16, hipMemcpyDeviceToDevice, s)); });
        // the SpMV's ratio, walked together instead of in slabs
        for (int g : {1536, 2048, 4096}) { STRIDE(256, 1, 11, true, false, g, n2); STRIDE(256, 2, 11, true, false, g, n2); }
        STRIDE(256, 1, 11, true, true, 2048, n2); STRIDE(256, 2, 11, true, true, 2048, n2);
        for (int g : {2048, 4096}) { STRIDE(256, 2, 2, true, false, g, n2 * 4); STRIDE(256, 4, 4, false, false, g, n2 * 3); }
    }
    // copy (1 GiB in, 1 GiB out)
    for (int g : {1024, 2048, 4096, 8192}) { RATIO(1, 2, true, false, g, n2 * 8); RATIO(1, 4, true, false, g, n2 * 8); }
    RATIO(1, 4, true, true, 2048, n2 * 8);
    RATIO(1, 8, true, false, 2048, n2 * 8);
    // triad-like 2:1
    for (int g : {2048, 4096}) { RATIO(2, 2, true, false, g, n2 * 4); RATIO(2, 4, true, false, g, n2 * 4); }
    STREAMS(2, 2, true, 2048, n2 * 4);
    // SpMV-like 11:1 (1.4 GiB read, 128 MiB written)
    for (int g : {1536, 2048, 4096}) { RATIO(11, 1, true, false, g, n2); RATIO(11, 2, true, false, g, n2); }
    RATIO(11, 1, true, true, 2048, n2);
    STREAMS(11, 1, true, 2048, n2);
    STREAMS(11, 2, true, 2048, n2);
    // read only
    for (int g : {2048, 4096}) { RATIO(4, 2, false, false, g, n2 * 3); RATIO(4, 4, false, false, g, n2 * 3); }
    RATIO(4, 2, false, true, 2048, n2 * 3);
    STREAMS(3, 2, false, 2048, n2 * 4);
    // cache-resident footprints (the 1M-DoF systems: ~100-150 MB per kernel, inside the 256 MiB Infinity Cache)
    printf("-- Infinity-Cache-resident footprints --\n");
    for (int g : {1536, 2048, 4096}) { RATIO(11, 1, true, false, g, n2 / 16); RATIO(11, 2, true, false, g, n2 / 16); }   // 88 MiB in, 8 MiB out
    RATIO(11, 2, true, true, 2048, n2 / 16);
    for (int g : {512, 1024, 2048}) { RATIO(2, 2, true, false, g, n2 / 16); RATIO(2, 4, true, false, g, n2 / 16); }     // K2/K3-like: 16 + 8 MiB
    for (int g : {512, 1024, 2048}) { RATIO(4, 2, false, false, g, n2 / 16); }                                          // read only 32 MiB
    RATIO(1, 4, true, false, 2048, n2 / 2);                                                                            // copy 64 + 64 MiB
    return 0;
}
