// Device-side helpers shared by the kernel files of libdpcg.so (gfx950, wave64): deterministic workgroup
// reductions, the workgroup -> work mapping, and the control heads that let the PCG iteration run without a host
// round trip.  Include from .hip files only.
#pragma once

#include "dpcg_internal.h"

namespace dpcg {

// ------------------------------------------------------------------------------------------------
// Deterministic reductions: wave64 shuffle tree -> 4 wave sums in LDS -> fixed-order add.
// ------------------------------------------------------------------------------------------------
// Sum over the 256 threads of the workgroup; every thread gets the result.  sh: 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    __syncthreads();  // sh may still be read from a previous use
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// Two sums at once (one pair of barriers).  sh: 8 doubles of LDS.
__device__ __forceinline__ void block_sum2(double &a, double &b, double *sh) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) {
        sh[threadIdx.x >> 6] = a;
        sh[4 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    b = ((sh[4] + sh[5]) + sh[6]) + sh[7];
}

// Every workgroup re-reduces the <= kMaxGrid partials a previous kernel wrote: same order in every
// workgroup, so all of them hold bit-identical scalars without any inter-workgroup hand-off.
__device__ __forceinline__ double reduce_partials(const double *__restrict__ part, int n_part, double *sh) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += kBlock) s += part[i];
    return block_sum(s, sh);
}

// The same reduction with the loads issued EARLY: a kernel whose first action depends on the reduced value (alpha, beta)
// requests its partials together with its first vector loads and its control words, and only then looks at them -- one
// exposed memory round trip at the head of the kernel instead of three (control word, partials, first data).  SLOTS *
// kBlock >= n_part; absent slots hold +0.0, which leaves the sum's bits unchanged (x + 0.0 == x), so the result is
// bit-identical to reduce_partials.
template <int SLOTS>
struct EarlyPartials {
    double v[SLOTS];
    // unconditional loads (index clamped to the last partial; n_part >= 1): a predicated load would become a branch
    // with its own wait
    __device__ __forceinline__ void request(const double *__restrict__ part, int n_part) {
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int i = (int)threadIdx.x + u * kBlock;
            v[u] = part[i < n_part ? i : n_part - 1];
        }
    }
    // Call in front of the branch on the control word: makes the values "used" there, so the compiler cannot sink the
    // loads behind the branch (it waits for them here -- together with the first vector loads, which are older).
    __device__ __forceinline__ void land() {
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) asm volatile("" : "+v"(v[u]));
    }
    __device__ __forceinline__ double reduce(int n_part, double *sh) const {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) s += ((int)threadIdx.x + u * kBlock < n_part) ? v[u] : 0.0;
        return block_sum(s, sh);
    }
};
// pins a scalar the head has loaded in front of the `done` branch (the compiler would otherwise sink the load behind it)
__device__ __forceinline__ void pin_scalar(double &x) { asm volatile("" : "+s"(x)); }
constexpr int kSpmvPartSlots = kMaxSpmvGrid / kBlock;   // 8
constexpr int kVecPartSlots = kMaxGrid / kBlock;        // 2

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous slab of the
// work so the x-vector halo of neighbouring row-blocks is shared in that XCD's 4 MiB L2.
// Placement is a speed matter only; any mapping gives the same result.
__device__ __forceinline__ int virtual_block() {
    const int G = gridDim.x, b = blockIdx.x;
    return (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
}


// [lo,hi) = floor(v*total/G), floor((v+1)*total/G) for G = gridDim.x <= 2048, without 64-bit division:
// total = q*G + r  =>  floor(v*total/G) = v*q + floor(v*r/G), and v*r < G*G fits 32 bits.
__device__ __forceinline__ void split_range(int total, int v, int &lo, int &hi) {
    const unsigned G = gridDim.x;
    const unsigned q = (unsigned)total / G, r = (unsigned)total - q * G;
    lo = (int)((unsigned)v * q + ((unsigned)v * r) / G);
    hi = (int)((unsigned)(v + 1) * q + ((unsigned)(v + 1) * r) / G);
}

// ------------------------------------------------------------------------------------------------
// Control flow of the iteration without a host round trip (cg.py:70-71):
//   * the stopping test for iterate k+1 is evaluated by workgroup 0 of K3 (the last kernel of
//     update k+1), which writes done / k / history / rz_next into the device-resident Scalars;
//   * K1 (the SpMV, head of the next update) only reads the `done` word an EARLIER kernel wrote and
//     workgroup 0 rotates rz_next -> rz; when `done` is set it latches `done_seen` for the K3 of its update.
//   * K2 and the preconditioner kernels read `done` (written by the K3 of an earlier update); K3 reads `done_seen`.
// No kernel reads a scalar that the same kernel writes.  Once `done` is set every later kernel of the replayed
// graph returns at once.
// ------------------------------------------------------------------------------------------------
struct IterCtlDev {
    Scalars *scal;
};

__device__ __forceinline__ bool iteration_head(const IterCtlDev &c) {
    Scalars *sc = c.scal;
    if (sc->done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->done_seen = 1;    // read by the K3 of this update (a later kernel)
        return false;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->rz = sc->rz_next;     // <r,z> of the current iterate, cg.py:76
    return true;
}

// The test of cg.py:71 on the iterate that update k has just produced; one thread of one workgroup.
__device__ __forceinline__ void record_and_test(Scalars *sc, double rr, double rz_next, double *hist, int hist_cap,
                                                int k) {
    const double res = rr / sc->bb;                                     // cg.py:15-17
    const bool conv = (res < sc->rtol_sq) || (rr < sc->atol_sq);        // cg.py:71
    if (k < hist_cap) hist[k] = res;                                    // cg.py:67,88
    sc->res = res;
    sc->k = k;
    sc->rz_next = rz_next;
    int done = 0;
    if (conv) { done = 1; sc->status = DPCG_OK; }
    else if (!(res == res)) { done = 1; sc->status = DPCG_BREAKDOWN; }
    if (done) sc->done = 1;
    if (sc->progress)   // one posted 8-byte write to pinned host memory per update
        __hip_atomic_store(sc->progress, ((unsigned long long)k << 1) | (unsigned long long)done, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// Two-kernel iteration.  CG has two global reductions per update (<p,Ap> and <r,z>), so two kernels is the floor:
//   KA (an SpMV kernel with FUSE): every workgroup re-reduces the partials of <r,z> and <r,r> that KB (or the
//      initial state) left, runs the stopping test of cg.py:71 on the current iterate k, forms
//      beta = <r,z>_k / <r,z>_{k-1} (cg.py:82), then  p_k = z + beta p_{k-1} (cg.py:83) and the deferred
//      x += alpha_{k-1} p_{k-1} (cg.py:79) for its own rows, and q = A p_k with the partials of <p_k,q>.  The
//      columns it gathers are recomputed as z[c] + beta p_{k-1}[c]: the same expression, hence the same bits,
//      as the stored p_k.  p is double-buffered (P[k & 1]) because neighbours still read p_{k-1}.
//   KB (k_update_r<PRE, true>): alpha, r, z, partials as before; workgroup 0 also advances k and rz_prev.
// x lags one update behind; k_final_fused applies the last one.  Before the first update rz_prev = +inf and
// alpha = 0, P[1] = 0, so update 0 degenerates to p_0 = z_0, x unchanged.  Scalars obey the same rule as in the
// three-kernel form: nobody reads a word that the same kernel writes -- KA's head included (see fused_head).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool fused_head(Scalars *sc, const FuseArgs &f, double *sh, double &alpha, double &beta) {
    // KA never reads `done`: workgroup 0 of THIS launch may be setting it while a late workgroup starts, and the waves of
    // that workgroup would not agree on what they saw -- a wave that left alone would leave its slots of the block reduction
    // below unwritten, the others would read stale LDS there, miss the stop and apply the deferred x += alpha p a second time
    // (seen with several solves sharing the chip: the last increment doubled in 64-row runs).  Every workgroup derives the
    // stop from the partials instead, which no kernel changes once the solve has stopped (KB and the preconditioner kernels
    // are no-ops then), so the no-op launches that follow convergence decide the same way; the one case the partials cannot
    // show -- the INITIAL iterate already passed the test (k = 0) -- travels in `done_seen`, written by k_fused_init only.
    if (sc->done_seen) return false;
    const int k = sc->k;                                   // updates completed (written by KB / the initial state)
    double rz = 0.0, rr = 0.0;
    for (int i = threadIdx.x; i < f.n_part; i += kBlock) rz += f.part_rz[i];
    for (int i = threadIdx.x; i < f.n_part_rr; i += kBlock) rr += f.part_rr[i];
    block_sum2(rz, rr, sh);                                // the arithmetic of reduce_partials, twice
    alpha = sc->alpha;
    beta = rz / sc->rz_prev;                               // cg.py:82
    bool stop = false;
    if (k > 0) {                                           // iterate 0 was tested by k_finalize_init (cg.py:66)
        const double res = rr / sc->bb;                    // cg.py:15-17
        const bool conv = (res < sc->rtol_sq) || (rr < sc->atol_sq);   // cg.py:71
        stop = conv || !(res == res);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (k < f.hist_cap) f.hist[k] = res;           // cg.py:88
            sc->res = res;
            if (stop) {
                sc->status = conv ? DPCG_OK : DPCG_BREAKDOWN;
                sc->done = 1;                              // read by KB and the preconditioner kernels (later launches)
                if (sc->progress)
                    __hip_atomic_store(sc->progress, ((unsigned long long)k << 1) | 1ull, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (stop) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->rz = rz;  // <r,z> of the current iterate, read by KB (cg.py:76)
    return true;
}

}  // namespace dpcg
