// Whole-solve kernel for SMALL systems: one 1024-thread workgroup runs the complete PCG (cg.py:58-90) of one
// system in a single launch; a batch is one launch with one workgroup per system (one system per CU).
//
// Why: the reference's real matrices have 2.4k-5.5k rows (params.yaml mesh_cells 2).
// At that size a PCG update is three ~3 us launches of almost empty kernels; here an update is a handful of
// workgroup barriers.  Vectors a thread owns (x, r, z, its part of p and A p) live in registers, the vectors other
// threads gather from (p for A p; r and L^T r for the multiply-type preconditioners) live in LDS, the matrix streams
// from L2.  Row sums run in column order with one rounding per operation, exactly as in the large-system kernels and
// the CPU path; reductions use a fixed shuffle/LDS tree, so results are bitwise reproducible.
#include "dpcg_internal.h"

namespace dpcg {

constexpr int kSmallThreads = 1024;

// Two sums over the 1024 threads at once; every thread gets both.  `red` = 2 x 32 doubles of LDS, `phase`
// alternates between two halves so one barrier per reduction suffices.
template <int NT>
__device__ __forceinline__ void small_reduce2(double &a, double &b, double *red, int &phase) {
    constexpr int NW = NT / 64;
    a = wave_sum(a);   // DPP tree, result in lane 63
    b = wave_sum(b);
    double *slot = red + (phase & 1) * 32;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) {
        slot[w] = a;
        slot[16 + w] = b;
    }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        sa += slot[i];
        sb += slot[16 + i];
    }
    a = sa;
    b = sb;
    ++phase;
}

// The matrices of a small system are re-laid out once as slab-ELL: rows in slabs of 1024 (one per thread),
// entry j of row i at [(slab*W + j)*1024 + i%1024].  Lanes of a wave then read consecutive addresses for a fixed j
// (coalesced 512-B loads; thread-per-row CSR reads touch 20 lines per load and thrash the 32 KiB L1), and the
// per-row order of the entries -- hence the rounding of the row sum -- is exactly the CSR order.
__global__ void k_build_ell(int n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                            const double *__restrict__ v, int W, int32_t *__restrict__ ell_col,
                            double *__restrict__ ell_val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int slab = i / kSmallThreads, lane = i % kSmallThreads;
    const int s = rp[i], len = rp[i + 1] - s;
    for (int j = 0; j < W; ++j) {
        const size_t o = ((size_t)slab * W + j) * kSmallThreads + lane;
        ell_col[o] = j < len ? ci[s + j] : i;
        ell_val[o] = j < len ? v[s + j] : 0.0;
    }
}

__global__ void k_max_row_len(int n, const int32_t *__restrict__ rp, int *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(out, rp[i + 1] - rp[i]);
}

void launch_max_row_len(int n, const int32_t *rp, int *out_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_max_row_len, dim3((n + 255) / 256), dim3(256), 0, s, n, rp, out_dev);
}
void launch_build_ell(int n, const int32_t *rp, const int32_t *ci, const double *v, int W, int32_t *ell_col,
                      double *ell_val, hipStream_t s) {
    hipLaunchKernelGGL(k_build_ell, dim3((n + 255) / 256), dim3(256), 0, s, n, rp, ci, v, W, ell_col, ell_val);
}

// q[k] = (row t + 1024 k) . xs with xs in LDS.  Entry index j is the OUTER loop so that the loads of all six
// rows of a thread are in flight together (memory-level parallelism); each row's sum still runs in column order.
// ELL offset of entry j of row i (slabs of 1024 rows)
__device__ __forceinline__ size_t ell_off(const SmallEll &E, int i, int j) {
    return ((size_t)(i / kSmallThreads) * E.W + j) * kSmallThreads + (i % kSmallThreads);
}

template <int NT, int RR>
__device__ __forceinline__ void small_spmv(const SmallEll &E, const int (&len)[RR], const double *xs,
                                           double (&q)[RR]) {
    // (an opaque copy of the thread index per call: the ELL offsets of the rows are loop-invariant, and hoisted out of the update loop
    // for three matrices they are registers a 1024-thread variant does not have -- 28 spill reloads per update of z = L (L^T r))
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    int lmax = 0;
#pragma unroll
    for (int k = 0; k < RR; ++k) {
        q[k] = 0.0;
        lmax = len[k] > lmax ? len[k] : lmax;
    }
#pragma unroll 2
    for (int j = 0; j < lmax; ++j) {
        int cc[RR];
        double vv[RR];
#pragma unroll
        for (int k = 0; k < RR; ++k) {
            const size_t o = ell_off(E, t + k * NT, j < len[k] ? j : 0);
            cc[k] = len[k] > 0 ? E.col[o] : 0;
            vv[k] = len[k] > 0 ? E.val[o] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < RR; ++k)
            if (j < len[k]) q[k] += vv[k] * xs[cc[k]];
    }
}

// The matrix of a really small system (<= RR rows per thread, <= WR entries per row) fits the register file:
// it is read ONCE before the loop; an update then touches no global memory at all except the history word.
template <int NT, int RR, int WR>
struct RegMatrix {
    static constexpr int WP = WR > 0 ? (WR + 1) / 2 : 1;
    double v[RR][WR > 0 ? WR : 1];
    unsigned c2[RR][WP];      // two 16-bit column indices per register (n <= 6144 < 65536)
    __device__ __forceinline__ void load(const SmallEll &E, const int (&len)[RR]) {
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < RR; ++k) {
#pragma unroll
            for (int j = 0; j < WP; ++j) c2[k][j] = 0u;
#pragma unroll
            for (int j = 0; j < WR; ++j) {
                // entries beyond the row's length become 0.0 * x[col of entry 0]: the sum below then runs
                // branch-free over all WR slots (adding +-0.0 changes nothing)
                const size_t o = len[k] > 0 ? ell_off(E, t + k * NT, j < len[k] ? j : 0) : 0;
                v[k][j] = (len[k] > 0 && j < len[k]) ? E.val[o] : 0.0;
                const unsigned c = len[k] > 0 ? (unsigned)E.col[o] : 0u;
                c2[k][j / 2] |= (j & 1) ? (c << 16) : c;
            }
        }
    }
    __device__ __forceinline__ void spmv(const int (&len)[RR], const double *xs, double (&q)[RR]) const {
#pragma unroll
        for (int k = 0; k < RR; ++k) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < WR; ++j)
                s += v[k][j] * xs[(j & 1) ? (c2[k][j / 2] >> 16) : (c2[k][j / 2] & 0xffffu)];
            q[k] = s;
        }
        (void)len;
    }
};

template <int NT, int RR>
__device__ __forceinline__ void small_rows(int n, const int32_t *__restrict__ rp, int (&len)[RR]) {
#pragma unroll
    for (int k = 0; k < RR; ++k) {
        const int i = (int)threadIdx.x + k * NT;
        len[k] = i < n ? rp[i + 1] - rp[i] : 0;
    }
}

// PRE: DPCG_PRECOND_NONE / JACOBI / CSR / LLT_MULTIPLY (compile-time, so each variant carries only its state)
// NT = threads per workgroup, RR = rows per thread this variant is compiled for, WR = register width of the matrix
// (0: stream it from L2).  The register-matrix variants run 768 threads: 12 waves = 3 per SIMD leave 168 VGPRs per
// thread, enough for the matrix without a single spill (a spill reload is a dependent ~1 us round trip per update).
template <int PRE, int NT, int RR, int WR>
__global__ __launch_bounds__(NT) void k_pcg_small(const SmallDesc *__restrict__ descs) {
    const SmallDesc d = descs[blockIdx.x];
    if (d.precond != PRE || d.variant != (RR * 16 + WR)) return;   // a mixed batch is launched once per variant present
    constexpr int kSmallRows = RR;
    constexpr int kSmallThreads = NT;   // row i of this thread: t + k * NT (shadows the slab width on purpose)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = d.n;
    double *ps = lds;                 // p, gathered by A p
    double *w1 = ps + n;              // r, gathered by M r / L^T r (multiply-type preconditioners)
    double *w2 = w1 + n;              // L^T r, gathered by L (L^T r)
    double *red = lds + (size_t)d.lds_vectors * n;   // 64 doubles
    int phase = 0;
    const int t = threadIdx.x;
    constexpr bool kFusedZ = PRE == DPCG_PRECOND_NONE || PRE == DPCG_PRECOND_JACOBI;   // z = r or dinv*r on the fly
    // The register-matrix variants keep x and dinv in LDS (the register file is for the matrix); row i of this
    // thread is slot t + 1024 k of those LDS vectors, touched by this thread only (no barrier needed).
    constexpr bool kLdsXD = WR > 0;
    double *xl = lds + (size_t)n;          // x      (kLdsXD only)
    double *dl = lds + 2 * (size_t)n;      // dinv   (kLdsXD only)
    double x[kLdsXD ? 1 : kSmallRows], dinv[kLdsXD ? 1 : kSmallRows];
    double r[kSmallRows], p[kSmallRows], q[kSmallRows];
    double z[kFusedZ ? 1 : kSmallRows];
    auto getx = [&](int k) -> double {
        if (kLdsXD) return t + k * NT < n ? xl[t + k * NT] : 0.0;
        return x[kLdsXD ? 0 : k];
    };
    auto setx = [&](int k, double v) {
        if (kLdsXD) { if (t + k * NT < n) xl[t + k * NT] = v; }
        else x[kLdsXD ? 0 : k] = v;
    };
    auto getd = [&](int k) -> double {
        if (kLdsXD) return t + k * NT < n ? dl[t + k * NT] : 0.0;
        return dinv[kLdsXD ? 0 : k];
    };
    auto zk = [&](int k) -> double {
        if (PRE == DPCG_PRECOND_NONE) return r[k];
        if (PRE == DPCG_PRECOND_JACOBI) return getd(k) * r[k];
        return z[kFusedZ ? 0 : k];
    };
    int len[kSmallRows], mlen[kSmallRows], tlen[kSmallRows];
    small_rows<NT>(n, d.rp, len);           // the row lengths never change: read them once
    RegMatrix<NT, RR, WR> regA;
    if (WR > 0) regA.load(d.ell_a, len);
    if (PRE == DPCG_PRECOND_CSR || PRE == DPCG_PRECOND_LLT_MULTIPLY) small_rows<NT>(n, d.m_rp, mlen);
    if (PRE == DPCG_PRECOND_LLT_MULTIPLY) small_rows<NT>(n, d.t_rp, tlen);
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) {
        const int i = t + k * kSmallThreads;
        const double dv = (PRE == DPCG_PRECOND_JACOBI && i < n) ? d.dinv[i] : 0.0;
        if (kLdsXD) { if (i < n) dl[i] = dv; } else dinv[kLdsXD ? 0 : k] = dv;
    }

    // z = M r for this thread's rows (cg.py:61,81)
    auto apply_precond = [&]() {
        if (kFusedZ) {
            // nothing to store: zk(k) recomputes r or dinv*r where it is used
        } else {
#pragma unroll
            for (int k = 0; k < kSmallRows; ++k) {
                const int i = t + k * kSmallThreads;
                if (i < n) w1[i] = r[k];
            }
            __syncthreads();
            if (PRE == DPCG_PRECOND_CSR) {
                small_spmv<NT>(d.ell_m, mlen, w1, q);                                  // z = M r (q is free here)
            } else {                                                              // z = L (L^T r)
                double tmp[kSmallRows];
                small_spmv<NT>(d.ell_t, tlen, w1, tmp);
#pragma unroll
                for (int k = 0; k < kSmallRows; ++k) {
                    const int i = t + k * kSmallThreads;
                    if (i < n) w2[i] = tmp[k];
                }
                __syncthreads();
                small_spmv<NT>(d.ell_m, mlen, w2, q);
            }
#pragma unroll
            for (int k = 0; k < kSmallRows; ++k) z[kFusedZ ? 0 : k] = q[k];
            __syncthreads();  // w1/w2 are rewritten by the next apply
        }
    };

    // ---- start of the solve (cg.py:58-67) ----
    double a_bb = 0.0, a_dummy = 0.0;
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) {
        const int i = t + k * kSmallThreads;
        const double bi = i < n ? d.b[i] : 0.0;
        setx(k, (i < n && d.x0) ? d.x0[i] : 0.0);                                  // cg.py:58
        r[k] = bi;
        a_bb += bi * bi;
    }
    if (d.x0) {                                                                    // r = b - A x0 (cg.py:60)
#pragma unroll
        for (int k = 0; k < kSmallRows; ++k) {
            const int i = t + k * kSmallThreads;
            if (i < n) ps[i] = getx(k);
        }
        __syncthreads();
        if (WR > 0) regA.spmv(len, ps, q);
        else small_spmv<NT>(d.ell_a, len, ps, q);
#pragma unroll
        for (int k = 0; k < kSmallRows; ++k) r[k] = r[k] - q[k];
        __syncthreads();
    }
    apply_precond();                                                               // cg.py:61
    double a_rz = 0.0, a_t = 0.0;
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) {
        const int i = t + k * kSmallThreads;
        const double zi = zk(k);
        p[k] = zi;                                                                 // cg.py:62
        if (i < n) ps[i] = zi;
        a_rz += r[k] * zi;
        a_t += d.init_check_r ? r[k] * r[k] : zi * zi;                             // cg.py:66 tests zk
    }
    small_reduce2<NT>(a_bb, a_dummy, red, phase);
    small_reduce2<NT>(a_rz, a_t, red, phase);                                          // also orders the ps writes
    const double bb = a_bb;
    double rz = a_rz;
    double res = a_t / bb;
    bool conv = (res < d.rtol_sq) || (a_t < d.atol_sq);
    if (t == 0 && d.hist_cap > 0) d.hist[0] = res;
    int it = 0, status = DPCG_MAX_ITER;
    if (conv) status = DPCG_OK;
    else if (!(res == res)) status = DPCG_BREAKDOWN;

    // ---- the loop (cg.py:70-87) ----
    while (status == DPCG_MAX_ITER && it < d.max_iter) {
        if (WR > 0) regA.spmv(len, ps, q);                                         // cg.py:75
        else small_spmv<NT>(d.ell_a, len, ps, q);
        double a_pq = 0.0, a_z = 0.0;
#pragma unroll
        for (int k = 0; k < kSmallRows; ++k) a_pq += p[k] * q[k];
        small_reduce2<NT>(a_pq, a_z, red, phase);
        const double alpha = rz / a_pq;                                            // cg.py:78
#pragma unroll
        for (int k = 0; k < kSmallRows; ++k) {
            setx(k, getx(k) + alpha * p[k]);                                       // cg.py:79
            r[k] = r[k] - alpha * q[k];                                            // cg.py:80
        }
        apply_precond();                                                           // cg.py:81
        double a_rzn = 0.0, a_rr = 0.0;
#pragma unroll
        for (int k = 0; k < kSmallRows; ++k) {
            a_rzn += r[k] * zk(k);
            a_rr += r[k] * r[k];
        }
        small_reduce2<NT>(a_rzn, a_rr, red, phase);   // every wave is past its reads of ps here
        const double beta = a_rzn / rz;                                            // cg.py:82
        rz = a_rzn;
#pragma unroll
        for (int k = 0; k < kSmallRows; ++k) {
            const int i = t + k * kSmallThreads;
            p[k] = zk(k) + beta * p[k];                                            // cg.py:83
            if (i < n) ps[i] = p[k];
        }
        __syncthreads();
        ++it;
        res = a_rr / bb;                                                           // cg.py:86
        if (t == 0 && it < d.hist_cap) d.hist[it] = res;
        conv = (res < d.rtol_sq) || (a_rr < d.atol_sq);                            // cg.py:71
        if (conv) status = DPCG_OK;
        else if (!(res == res)) status = DPCG_BREAKDOWN;
    }
#pragma unroll
    for (int k = 0; k < kSmallRows; ++k) {
        const int i = t + k * kSmallThreads;
        if (i < n && d.x) d.x[i] = getx(k);
    }
    if (t == 0) {
        d.out->k = it;
        d.out->res = res;
        d.out->status = status;
        d.out->done = 1;
        d.out->rz = rz;
        d.out->bb = bb;
    }
}

template <int PRE, int NT, int RR, int WR>
static int launch_one(const SmallDesc *descs_dev, int count, int lds_bytes, hipStream_t s) {
    static int attr_set_for = 0;
    if (lds_bytes > attr_set_for) {
        if (hipFuncSetAttribute((const void *)k_pcg_small<PRE, NT, RR, WR>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes) != hipSuccess)
            return DPCG_ERR_HIP;
        attr_set_for = lds_bytes;
    }
    hipLaunchKernelGGL((k_pcg_small<PRE, NT, RR, WR>), dim3(count), dim3(NT), (size_t)lds_bytes, s, descs_dev);
    return DPCG_OK;
}

// Register-matrix variants (768 threads) exist for M in {I, Jacobi}: 4 rows/thread x <= 7 entries covers n <= 3072 --
// the reference's res-2 meshes (2424 rows) -- and 6 rows/thread x <= 5 entries covers 5-point systems up to 4608 rows.
constexpr int kRegThreads = 768;
int small_variant(int n, int max_row_len, int precond) {
    if (precond == DPCG_PRECOND_NONE || precond == DPCG_PRECOND_JACOBI) {
        const int rows = (n + kRegThreads - 1) / kRegThreads;
        if (rows <= 4 && max_row_len <= 7) return 4 * 16 + 7;
        if (rows <= 6 && max_row_len <= 5) return 6 * 16 + 5;
    }
    return (kSmallMaxN / 1024) * 16 + 0;
}

// kinds_mask: bit p set = some system of the batch uses preconditioner kind p; variants_mask: bit 0 = streamed
// matrix, bit 1 = (4,7) registers, bit 2 = (6,5) registers.  One launch per combination present; workgroups of
// another combination return at once.
int launch_pcg_small(const SmallDesc *descs_dev, int count, int lds_bytes, int kinds_mask, int variants_mask,
                     hipStream_t s) {
    int st = DPCG_OK;
    constexpr int R6 = kSmallMaxN / 1024;
#define DPCG_SMALL(PREV)                                                                                              \
    if (st >= 0 && (kinds_mask & (1 << PREV)) && (variants_mask & 1)) st = launch_one<PREV, 1024, R6, 0>(descs_dev, count, lds_bytes, s)
    DPCG_SMALL(DPCG_PRECOND_NONE);
    DPCG_SMALL(DPCG_PRECOND_JACOBI);
    DPCG_SMALL(DPCG_PRECOND_CSR);
    DPCG_SMALL(DPCG_PRECOND_LLT_MULTIPLY);
#undef DPCG_SMALL
#define DPCG_SMALL_REG(PREV, RRV, WRV, BIT)                                                                           \
    if (st >= 0 && (kinds_mask & (1 << PREV)) && (variants_mask & BIT)) st = launch_one<PREV, kRegThreads, RRV, WRV>(descs_dev, count, lds_bytes, s)
    DPCG_SMALL_REG(DPCG_PRECOND_NONE, 4, 7, 2);
    DPCG_SMALL_REG(DPCG_PRECOND_JACOBI, 4, 7, 2);
    DPCG_SMALL_REG(DPCG_PRECOND_NONE, 6, 5, 4);
    DPCG_SMALL_REG(DPCG_PRECOND_JACOBI, 6, 5, 4);
#undef DPCG_SMALL_REG
    return st;
}

}  // namespace dpcg
