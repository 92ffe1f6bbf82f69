// Hop latency of a single-word (8-byte, value = flag) hand-off between workgroups on MI355X, the primitive a sync-free
// triangular solve is made of.  A chain: workgroup w waits for slot[w-1] != pending, adds 1, stores slot[w].  The time of
// the whole chain / hops = latency per dependent hop.  Variants: store flavour (agent-scope sc1 write-through vs plain
// write-through-to-L2) x placement (neighbours on the same XCD vs on different XCDs), consumer always polls with sc1 loads.
//   hipcc --offload-arch=gfx950 -O3 tools/hop_lab.hip -o tools/hop_lab && tools/hop_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

constexpr unsigned long long kPending = 0x7ff8dead0badbeefULL;

// ranks: logical position of this workgroup in the chain.  mode 0: position = blockIdx (neighbours on different XCDs);
// mode 1: only workgroups of XCD `want` take part, positions by a ticket (neighbours on the same XCD).
template <bool PLAIN_STORE>
__global__ void k_chain(double *slot, unsigned *ticket, int hops, int same_xcd, unsigned *xcc_seen, int rounds) {
    __shared__ unsigned s_pos;
    __shared__ int s_take;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        int take = 1;
        if (same_xcd) {
            // the XCD of the first workgroup to arrive owns the chain
            unsigned owner = atomicCAS(xcc_seen, 0xffffffffu, x);
            if (owner == 0xffffffffu) owner = x;
            take = owner == x;
        }
        s_take = take;
        s_pos = take ? atomicAdd(ticket, 1u) : 0u;
    }
    __syncthreads();
    if (!s_take) return;
    if (threadIdx.x != 0) return;
    // this workgroup serves chain positions pos, pos + P, ... where P = number of participants (unknown: use tickets again)
    unsigned pos = s_pos;
    while ((int)pos < hops * rounds) {
        double v = 0.0;
        if (pos > 0) {
            for (unsigned spin = 0;; ++spin) {
                v = __hip_atomic_load(slot + pos - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned long long)__double_as_longlong(v) != kPending) break;
                if (spin > (1u << 14)) { v = -1e9; break; }   // bounded: a broken hand-off shows as a wrong end value
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (PLAIN_STORE) __hip_atomic_store(slot + pos, v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(slot + pos, v + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pos = atomicAdd(ticket, 1u);
    }
}

__global__ void k_fill(double *slot, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slot[i] = __longlong_as_double((long long)kPending);
}

int main() {
    const int hops = 512, rounds = 1;
    setvbuf(stdout, nullptr, _IONBF, 0);
    double *slot;
    unsigned *ctl;
    hipMalloc(&slot, hops * sizeof(double));
    hipMalloc(&ctl, 2 * sizeof(unsigned));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int same = 0; same < 2; ++same)
        for (int plain = 0; plain < 2; ++plain)
            for (int grid : {8, 64, 256}) {
                if (!same && plain) continue;   // a plain store never leaves the producer's XCD L2 before the kernel ends
                float best = 1e9f;
                double last = 0;
                for (int rep = 0; rep < 5; ++rep) {
                    hipLaunchKernelGGL(k_fill, dim3((hops + 255) / 256), dim3(256), 0, 0, slot, hops);
                    unsigned init[2] = {0u, 0xffffffffu};
                    hipMemcpy(ctl, init, sizeof(init), hipMemcpyHostToDevice);
                    hipEventRecord(e0);
                    if (plain) hipLaunchKernelGGL(k_chain<true>, dim3(grid), dim3(64), 0, 0, slot, ctl, hops, same, ctl + 1, rounds);
                    else hipLaunchKernelGGL(k_chain<false>, dim3(grid), dim3(64), 0, 0, slot, ctl, hops, same, ctl + 1, rounds);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms = 0;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                    hipMemcpy(&last, slot + hops - 1, sizeof(double), hipMemcpyDeviceToHost);
                }
                printf("placement %-9s store %-5s grid %3d: %7.3f us/hop  (chain end value %.0f, expect %d)\n",
                       same ? "same-XCD" : "any-XCD", plain ? "plain" : "sc1", grid, best * 1e3 / hops, last, hops);
            }
    return 0;
}
