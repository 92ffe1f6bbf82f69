// wg_lab: are 256 workgroups of 1024 threads resident together?  Each workgroup runs `steps` barrier-separated
// steps of pure LDS/ALU work (no global memory) -- time for G workgroups vs one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
template <int MODE>
__global__ __launch_bounds__(1024) void k(int steps, double *out, const double *gin, int stride) {
    extern __shared__ double lds[];
    double acc = threadIdx.x;
    lds[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        double v = lds[(threadIdx.x * 7 + s) & 1023];
        if (MODE == 1) v += gin[(size_t)blockIdx.x * stride + ((threadIdx.x + s * 1024) & 8191)];   // 64 KB per WG from global
        acc = acc * 0.999 + v;
        __syncthreads();
        lds[threadIdx.x] = acc;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
int main() {
    double *out, *gin; CK(hipMalloc(&out, 4096 * 8)); CK(hipMalloc(&gin, (size_t)256 * (1 << 20) * 8 / 8 + (1<<20)));
    CK(hipMemset(gin, 0, (size_t)256 * (1 << 20) + (1<<20)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int lds : {8192, 24576, 65536})
            for (int G : {1, 32, 64, 128, 256}) {
                for (int stride : {131072, 131072 + 520}) {   // doubles between workgroups' global slabs
                    if (mode == 0 && stride != 131072) continue;
                    auto go = [&]() { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(1024), lds, 0, 2000, out, gin, stride); else hipLaunchKernelGGL(k<1>, dim3(G), dim3(1024), lds, 0, 2000, out, gin, stride); };
                    if (lds > 49152) { CK(hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); CK(hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); }
                    go(); CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("mode %d (%s) lds %6d stride %7d G %3d: %.3f ms -> %.2f us/step\n", mode, mode ? "global+lds" : "lds only", lds, stride, G, ms, ms * 1e3 / 2000);
                }
            }
    return 0;
}
