// Vector kernels of the PCG iteration (cg.py:58-90) and of the start / end of a solve.
// Compiled with -ffp-contract=off so that a*b+c is two roundings, as in the CPU reference path (scipy/ATen CSR row
// sums, unfused torch mul+add at cg.py:79-83); in-order sums then reproduce the oracle bit for bit.
#include "dpcg_device.h"

namespace dpcg {

// 16-byte non-temporal accesses (the builtin wants a native vector type)
typedef double nt_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load2(const double2 *p) {
    const nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_d2 *>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void nt_store2(double2 v, double2 *p) {
    nt_d2 w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<nt_d2 *>(p));
}

// ------------------------------------------------------------------------------------------------
// Vector updates of the iteration (cg.py:78-83) in two passes of 40 bytes per row each:
//   K2 (k_update_r):  alpha;  r -= alpha q;  z = dinv r;  partials <r,z>, <r,r>     reads q,r,dinv  writes r,z
//   K3 (k_update_xp): beta;   x += alpha p;  p = z + beta p                         reads z,p,x     writes x,p
// p is read once for both of its uses.  All vectors are handle-owned (256-B aligned): 16-byte
// accesses, loads of the next pair issued before the current pair is consumed, and the first loads
// issued before the partial reduction so that its latency is hidden.
// ------------------------------------------------------------------------------------------------
// PRE: 0 = M = I (z aliases r, not stored), 1 = Jacobi fused, 2 = generic M (z computed later), 3 = as 2, and the first
// level of the lower triangular solve that follows rides along (colour sweeps: its rows have no dependencies, y = r / d):
// `dinv` then holds that level's DIAGONAL by row (RideArgs), and y goes to the solve's own numbering, out[pos[row]].
struct RideArgs {
    const int32_t *pos;     // row -> level-order position of the factor (Levels::lm_pos)
    double *out;            // the lower solve's solution vector by position (Levels::lm_out)
    int count;              // rows of the first level (positions 0 .. count)
};
// NT (systems whose vectors stream from HBM: 256^3): q -- read once here -- is loaded and r stored non-temporally, so that the kernel
// does not leave an L2 full of dirty lines behind (their write-back is what the next launch waits for: 6 us at every K3 -> K1
// boundary of the 256^3 loop) and the streams do not evict what IS reused.
template <int PRE, bool F2 = false, bool NT = false>   // F2: KB of the two-kernel iteration (workgroup 0 also advances k and rz_prev)
__global__ __launch_bounds__(kBlock) void k_update_r(int64_t n, Scalars *__restrict__ sc,
                                                     const double *__restrict__ part_pq, int n_part_pq,
                                                     const double *__restrict__ q, double *__restrict__ r,
                                                     const double *__restrict__ dinv, double *__restrict__ z,
                                                     double *__restrict__ part_rz, double *__restrict__ part_rr,
                                                     int store_z, RideArgs ride) {
    __shared__ double sh[8];
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double2 *__restrict__ q2 = reinterpret_cast<const double2 *>(q);
    const double2 *__restrict__ d2 = reinterpret_cast<const double2 *>(dinv);
    double2 *__restrict__ r2 = reinterpret_cast<double2 *>(r);
    double2 *__restrict__ z2 = reinterpret_cast<double2 *>(z);
    const int2 *__restrict__ pos2 = reinterpret_cast<const int2 *>(ride.pos);
    double2 qa = make_double2(0, 0), ra = qa, da = qa;
    int2 pa = make_int2(0, 0);
    bool have = i < n2;
    if (have) {
        qa = NT ? nt_load2(q2 + i) : q2[i];
        ra = r2[i];
        if (PRE == 1 || PRE == 3) da = d2[i];
        if (PRE == 3) pa = pos2[i];
    }
    // Everything the head needs is requested before anything is looked at: the SpMV's partials of <p,Ap>, the `done`
    // word and <r,z> travel together with the first vector loads, so the head exposes ONE memory round trip; a finished
    // solve has merely loaded a few values for nothing.
    EarlyPartials<kSpmvPartSlots> ep;
    ep.request(part_pq, n_part_pq);
    const int done = sc->done;
    double rz_cur = sc->rz;
    pin_scalar(rz_cur);
    ep.land();
    if (done) return;
    const double pq = ep.reduce(n_part_pq, sh);
    const double alpha = rz_cur / pq;                                   // cg.py:78
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->alpha = alpha;                                              // read by K3 / KA (a later kernel)
        if (F2) {
            const int k1 = sc->k + 1;                                   // this update is complete once KB has run
            sc->rz_prev = rz_cur;                                       // read by the next KA only
            sc->k = k1;
            if (sc->progress)
                __hip_atomic_store(sc->progress, (unsigned long long)k1 << 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    double a_rz = 0.0, a_rr = 0.0;
    while (have) {
        const int64_t cur = i;
        const double2 qc = qa, rc = ra, dc = da;
        const int2 pc = pa;
        i += stride;
        have = i < n2;
        if (have) {
            qa = NT ? nt_load2(q2 + i) : q2[i];
            ra = r2[i];
            if (PRE == 1 || PRE == 3) da = d2[i];
            if (PRE == 3) pa = pos2[i];
        }
        double2 rn;
        rn.x = rc.x - alpha * qc.x;                                     // cg.py:80
        rn.y = rc.y - alpha * qc.y;
        if (NT) nt_store2(rn, r2 + cur);
        else r2[cur] = rn;
        a_rr += rn.x * rn.x;                                            // cg.py:86
        a_rr += rn.y * rn.y;
        if (PRE == 3) {                                                 // first level of L y = r (cg.py:81)
            if (pc.x < ride.count) ride.out[pc.x] = rn.x / dc.x;
            if (pc.y < ride.count) ride.out[pc.y] = rn.y / dc.y;
        }
        if (PRE == 1) {
            double2 zn;
            zn.x = dc.x * rn.x;                                         // cg.py:81 (M = diag(1/a_ii))
            zn.y = dc.y * rn.y;
            if (store_z) z2[cur] = zn;
            a_rz += rn.x * zn.x;                                        // cg.py:82
            a_rz += rn.y * zn.y;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {               // odd tail element
        const int64_t e = n - 1;
        const double rn = r[e] - alpha * q[e];
        r[e] = rn;
        a_rr += rn * rn;
        if (PRE == 3) {
            const int pe = ride.pos[e];
            if (pe < ride.count) ride.out[pe] = rn / dinv[e];
        }
        if (PRE == 1) {
            const double zn = dinv[e] * rn;
            z[e] = zn;
            a_rz += rn * zn;
        }
    }
    block_sum2(a_rr, a_rz, sh);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = a_rr;
        if (PRE < 2) part_rz[blockIdx.x] = PRE == 1 ? a_rz : a_rr;      // PRE 0: z = r; PRE 2, 3: <r,z> comes later
    }
}

void launch_update_r_two_kernel(int precond_fused, int64_t n, Scalars *scal, const double *part_pq, int n_part_pq,
                                const double *q, double *r, const double *dinv, double *z, double *part_rz,
                                double *part_rr, int grid, hipStream_t s) {
    if (precond_fused == 0)
        hipLaunchKernelGGL((k_update_r<0, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r,
                           dinv, z, part_rz, part_rr, 1, RideArgs{nullptr, nullptr, 0});
    else if (precond_fused == 1)
        hipLaunchKernelGGL((k_update_r<1, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r,
                           dinv, z, part_rz, part_rr, 1, RideArgs{nullptr, nullptr, 0});
    else
        hipLaunchKernelGGL((k_update_r<2, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r,
                           dinv, z, part_rz, part_rr, 1, RideArgs{nullptr, nullptr, 0});
}

// Two-kernel iteration: state before the first update (see fused_head).
__global__ void k_fused_init(Scalars *sc) {
    if (threadIdx.x == 0) {
        sc->rz_prev = __builtin_huge_val();   // beta_0 = <r,z>_0 / inf = 0  =>  p_0 = z_0
        sc->alpha = 0.0;                      // no deferred x update yet
        sc->done_seen = sc->done;             // "the INITIAL iterate already passed the test": what KA's head reads (see fused_head)
    }
}

void launch_fused_init(Scalars *scal, hipStream_t s) { hipLaunchKernelGGL(k_fused_init, dim3(1), dim3(64), 0, s, scal); }

// End of a two-kernel solve: the deferred x += alpha_{k-1} p_{k-1} (cg.py:79) and, when the loop ran out of
// updates, the test of the last iterate (cg.py:86-88; the status stays MAX_ITER unless it passes).
__global__ __launch_bounds__(kBlock) void k_final_fused(int64_t n, Scalars *sc, const double *__restrict__ part_rr,
                                                        int n_part, double *hist, int hist_cap, double *__restrict__ x,
                                                        const double *__restrict__ p0, const double *__restrict__ p1) {
    __shared__ double sh[4];
    const int k = sc->k;
    if (k >= 1) {
        const double alpha = sc->alpha;
        const double *__restrict__ p = ((k - 1) & 1) ? p1 : p0;
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] = x[i] + alpha * p[i];
    }
    if (blockIdx.x == 0) {
        const int done = sc->done;                         // uniform: written by an earlier kernel only
        if (!done) {
            const double rr = reduce_partials(part_rr, n_part, sh);
            if (threadIdx.x == 0) {
                if (k >= 1) {                              // iterate k has not been tested yet
                    const double res = rr / sc->bb;
                    const bool conv = (res < sc->rtol_sq) || (rr < sc->atol_sq);
                    if (k < hist_cap) hist[k] = res;
                    sc->res = res;
                    sc->status = conv ? DPCG_OK : (!(res == res) ? DPCG_BREAKDOWN : DPCG_MAX_ITER);
                } else {
                    sc->status = DPCG_MAX_ITER;
                }
                sc->done = 1;
            }
        }
    }
}

void launch_final_fused(int64_t n, Scalars *scal, const double *part_rr, int n_part, double *hist, int hist_cap,
                        double *x, const double *p0, const double *p1, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_final_fused, dim3(grid), dim3(kBlock), 0, s, n, scal, part_rr, n_part, hist, hist_cap, x, p0,
                       p1);
}

void launch_update_r(int precond_fused, int64_t n, Scalars *scal, const double *part_pq, int n_part_pq,
                     const double *q, double *r, const double *dinv, double *z, double *part_rz, double *part_rr,
                     int grid, hipStream_t s, int store_z, bool nt) {
    const RideArgs none{nullptr, nullptr, 0};
    if (nt && precond_fused == 1 && !store_z) {
        hipLaunchKernelGGL((k_update_r<1, false, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr, store_z, none);
        return;
    }
    if (nt && precond_fused == 0) {
        hipLaunchKernelGGL((k_update_r<0, false, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr, store_z, none);
        return;
    }
    if (precond_fused == 0)
        hipLaunchKernelGGL(k_update_r<0>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr, store_z, none);
    else if (precond_fused == 1)
        hipLaunchKernelGGL(k_update_r<1>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr, store_z, none);
    else
        hipLaunchKernelGGL(k_update_r<2>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr, store_z, none);
}

// K2 (two_kernel: KB) for a preconditioner applied by colour sweeps: the first level of the lower solve rides along (k_update_r<3>)
void launch_update_r_ride(int64_t n, Scalars *scal, const double *part_pq, int n_part_pq, const double *q, double *r,
                          const double *first_level_diag, const int32_t *pos, double *lm_out, int first_level_rows,
                          double *part_rr, int grid, hipStream_t s, bool two_kernel) {
    const RideArgs ride{pos, lm_out, first_level_rows};
    if (two_kernel)
        hipLaunchKernelGGL((k_update_r<3, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, first_level_diag,
                           (double *)nullptr, (double *)nullptr, part_rr, 0, ride);
    else
        hipLaunchKernelGGL((k_update_r<3, false>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, first_level_diag,
                           (double *)nullptr, (double *)nullptr, part_rr, 0, ride);
}

// part[b] = partial of <a,b>; skipped once the solve is done (scal may be null: always run).
__global__ __launch_bounds__(kBlock) void k_dot_partials(int64_t n, const Scalars *__restrict__ sc,
                                                         const double *__restrict__ a, const double *__restrict__ b,
                                                         double *__restrict__ part) {
    __shared__ double sh[4];
    if (sc && sc->done) return;
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc += a[i] * b[i];
    const double tot = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

void launch_dot_partials(int64_t n, const Scalars *scal, const double *a, const double *b, double *part, int grid,
                         hipStream_t s) {
    hipLaunchKernelGGL(k_dot_partials, dim3(grid), dim3(kBlock), 0, s, n, scal, a, b, part);
}

// x += alpha p; p = z + beta p (cg.py:79,82-83); optionally also the fp32 copy of p that the
// mixed-precision SpMV gathers.
template <bool P32, bool MANY>   // MANY: more than kVecPartSlots * kBlock partials of <r,z> (an M-apply's SpMV summed them)
__global__ __launch_bounds__(kBlock) void k_update_xp(int64_t n, Scalars *__restrict__ sc,
                                                      const double *__restrict__ part_rz,
                                                      const double *__restrict__ part_rr, int n_part,
                                                      const double *__restrict__ z, double *__restrict__ p,
                                                      double *__restrict__ x, float *__restrict__ p32,
                                                      double *__restrict__ hist, int hist_cap,
                                                      const double *__restrict__ zd, int n_part_rr) {
    __shared__ double sh[4];
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double2 *__restrict__ z2 = reinterpret_cast<const double2 *>(z);
    double2 *__restrict__ p2 = reinterpret_cast<double2 *>(p);
    double2 *__restrict__ x2 = reinterpret_cast<double2 *>(x);
    float2 *__restrict__ f2 = reinterpret_cast<float2 *>(p32);
    const double2 *__restrict__ zd2 = reinterpret_cast<const double2 *>(zd);
    double2 za = make_double2(0, 0), pa = za, xa = za, da = za;
    bool have = i < n2;
    if (have) {
        za = z2[i];
        pa = p2[i];
        xa = x2[i];
        if (zd) da = zd2[i];
    }
    // `done_seen`, not `done`: workgroup 0 of THIS launch sets `done`, and a workgroup dispatched after that must still
    // apply x += alpha p for its rows (cg.py:79 precedes the test of cg.py:86).
    // as in K2: one exposed round trip at the head (more than kVecPartSlots * kBlock partials -- an M-apply's SpMV summed
    // <r,z> -- take the loop form)
    constexpr bool many = MANY;
    EarlyPartials<kVecPartSlots> ep;
    ep.request(part_rz, many ? 1 : n_part);
    const int done_seen = sc->done_seen;
    double rz_old = sc->rz, alpha = sc->alpha;
    pin_scalar(rz_old);
    pin_scalar(alpha);
    ep.land();
    if (done_seen) return;
    const double rz_new = many ? reduce_partials(part_rz, n_part, sh) : ep.reduce(n_part, sh);
    const double beta = rz_new / rz_old;                                // cg.py:82
    if (blockIdx.x == 0) {                                              // cg.py:86 + the test of cg.py:71
        const double rr = reduce_partials(part_rr, n_part_rr, sh);
        if (threadIdx.x == 0) record_and_test(sc, rr, rz_new, hist, hist_cap, sc->k + 1);
    }
    while (have) {
        const int64_t cur = i;
        double2 zc = za;
        const double2 pc = pa, xc = xa, dc = da;
        i += stride;
        have = i < n2;
        if (have) {
            za = z2[i];
            pa = p2[i];
            xa = x2[i];
            if (zd) da = zd2[i];
        }
        if (zd) {                                                       // z = dinv * r, recomputed (cg.py:81)
            zc.x = dc.x * zc.x;
            zc.y = dc.y * zc.y;
        }
        double2 xn, pn;
        xn.x = xc.x + alpha * pc.x;                                     // cg.py:79
        xn.y = xc.y + alpha * pc.y;
        pn.x = zc.x + beta * pc.x;                                      // cg.py:83
        pn.y = zc.y + beta * pc.y;
        x2[cur] = xn;
        p2[cur] = pn;
        if (P32) f2[cur] = make_float2((float)pn.x, (float)pn.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t e = n - 1;
        const double pe = p[e];
        x[e] = x[e] + alpha * pe;
        const double pn = (zd ? zd[e] * z[e] : z[e]) + beta * pe;
        p[e] = pn;
        if (P32) p32[e] = (float)pn;
    }
}

void launch_update_xp(int64_t n, Scalars *scal, const double *part_rz, const double *part_rr, int n_part,
                      const double *z, double *p, double *x, float *p32, double *hist, int hist_cap, int grid,
                      hipStream_t s, const double *zd, int n_part_rr) {
    const bool many = n_part > kVecPartSlots * kBlock;
#define DPCG_K3(P32V, MANYV)                                                                                           \
    hipLaunchKernelGGL((k_update_xp<P32V, MANYV>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_rz, part_rr, n_part, z, p, \
                       x, p32, hist, hist_cap, zd, n_part_rr)
    if (p32) {
        if (many) DPCG_K3(true, true);
        else DPCG_K3(true, false);
    } else {
        if (many) DPCG_K3(false, true);
        else DPCG_K3(false, false);
    }
#undef DPCG_K3
}

// K3 with the x update deferred.  x is only an output, and every write costs more than a read here, so x is brought
// up to date every SECOND update:  even update j  -> p_{j+1} = z + beta p_j  into the other p buffer (p_j survives);
//                                  odd update j+1 -> x = (x + alpha_j p_j) + alpha_{j+1} p_{j+1}, then p_{j+2} over p_j.
// Same operations in the same order as x += alpha p every update (cg.py:79): bit-identical iterates; per two updates
// one x read and one x write are replaced by one extra read of p.  k_final_deferred applies a pending half.
template <bool P32, bool ODD, bool MANY, bool NT = false>   // NT: as in k_update_r -- x (read and rewritten every second update) and the new p
__global__ __launch_bounds__(kBlock) void k_update_xp_deferred(int64_t n, Scalars *__restrict__ sc,
                                                               const double *__restrict__ part_rz,
                                                               const double *__restrict__ part_rr, int n_part,
                                                               const double *__restrict__ z, const double *p_in,
                                                               double *p_out, double *__restrict__ x,
                                                               float *__restrict__ p32, double *__restrict__ hist,
                                                               int hist_cap, const double *__restrict__ zd, int n_part_rr) {
    __shared__ double sh[4];
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double2 *__restrict__ z2 = reinterpret_cast<const double2 *>(z);
    const double2 *pi2 = reinterpret_cast<const double2 *>(p_in);
    double2 *po2 = reinterpret_cast<double2 *>(p_out);          // ODD: holds p of the previous update until overwritten
    double2 *__restrict__ x2 = reinterpret_cast<double2 *>(x);
    float2 *__restrict__ f2 = reinterpret_cast<float2 *>(p32);
    const double2 *__restrict__ zd2 = reinterpret_cast<const double2 *>(zd);
    double2 za = make_double2(0, 0), pa = za, qa = za, xa = za, da = za;
    bool have = i < n2;
    if (have) {
        za = z2[i];
        pa = pi2[i];
        if (ODD) {
            qa = po2[i];
            xa = NT ? nt_load2(x2 + i) : x2[i];
        }
        if (zd) da = zd2[i];
    }
    constexpr bool many = MANY;                                         // see k_update_xp
    EarlyPartials<kVecPartSlots> ep;                                    // as in K2: one exposed round trip at the head
    ep.request(part_rz, many ? 1 : n_part);
    const int done_seen = sc->done_seen;                                // see k_update_xp: never `done` here
    double rz_old = sc->rz, alpha = sc->alpha;
    double alpha_prev = ODD ? sc->alpha_prev : 0.0;                     // written by the even update before this one
    pin_scalar(rz_old);
    pin_scalar(alpha);
    pin_scalar(alpha_prev);
    ep.land();
    if (done_seen) return;
    const double rz_new = many ? reduce_partials(part_rz, n_part, sh) : ep.reduce(n_part, sh);
    const double beta = rz_new / rz_old;                                // cg.py:82
    if (blockIdx.x == 0) {                                              // cg.py:86 + the test of cg.py:71
        const double rr = reduce_partials(part_rr, n_part_rr, sh);
        if (threadIdx.x == 0) {
            if (!ODD) sc->alpha_prev = alpha;                           // read by the next (odd) update only
            record_and_test(sc, rr, rz_new, hist, hist_cap, sc->k + 1);
        }
    }
    while (have) {
        const int64_t cur = i;
        double2 zc = za;
        const double2 pc = pa, qc = qa, xc = xa, dc = da;
        i += stride;
        have = i < n2;
        if (have) {
            za = z2[i];
            pa = pi2[i];
            if (ODD) {
                qa = po2[i];
                xa = NT ? nt_load2(x2 + i) : x2[i];
            }
            if (zd) da = zd2[i];
        }
        if (zd) {                                                       // z = dinv * r, recomputed (cg.py:81)
            zc.x = dc.x * zc.x;
            zc.y = dc.y * zc.y;
        }
        if (ODD) {
            double2 xn;
            xn.x = xc.x + alpha_prev * qc.x;                            // cg.py:79 of the previous update ...
            xn.y = xc.y + alpha_prev * qc.y;
            xn.x = xn.x + alpha * pc.x;                                 // ... and of this one
            xn.y = xn.y + alpha * pc.y;
            if (NT) nt_store2(xn, x2 + cur);
            else x2[cur] = xn;
        }
        double2 pn;
        pn.x = zc.x + beta * pc.x;                                      // cg.py:83
        pn.y = zc.y + beta * pc.y;
        if (NT) nt_store2(pn, po2 + cur);
        else po2[cur] = pn;
        if (P32) f2[cur] = make_float2((float)pn.x, (float)pn.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t e = n - 1;
        const double pe = p_in[e];
        if (ODD) {
            const double xe = x[e] + alpha_prev * p_out[e];
            x[e] = xe + alpha * pe;
        }
        const double pn = (zd ? zd[e] * z[e] : z[e]) + beta * pe;
        p_out[e] = pn;
        if (P32) p32[e] = (float)pn;
    }
}

void launch_update_xp_deferred(bool odd, int64_t n, Scalars *scal, const double *part_rz, const double *part_rr, int n_part,
                               const double *z, const double *p_in, double *p_out, double *x, float *p32, double *hist,
                               int hist_cap, int grid, hipStream_t s, const double *zd, int n_part_rr, bool nt) {
    const bool many = n_part > kVecPartSlots * kBlock;
    if (nt && !p32 && !many) {
        if (odd)
            hipLaunchKernelGGL((k_update_xp_deferred<false, true, false, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_rz, part_rr,
                               n_part, z, p_in, p_out, x, p32, hist, hist_cap, zd, n_part_rr);
        else
            hipLaunchKernelGGL((k_update_xp_deferred<false, false, false, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_rz, part_rr,
                               n_part, z, p_in, p_out, x, p32, hist, hist_cap, zd, n_part_rr);
        return;
    }
#define DPCG_K3D(P32V, ODDV, MANYV)                                                                                     \
    hipLaunchKernelGGL((k_update_xp_deferred<P32V, ODDV, MANYV>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_rz,     \
                       part_rr, n_part, z, p_in, p_out, x, p32, hist, hist_cap, zd, n_part_rr)
#define DPCG_K3D_MANY(P32V, ODDV) \
    do {                          \
        if (many) DPCG_K3D(P32V, ODDV, true); \
        else DPCG_K3D(P32V, ODDV, false);     \
    } while (0)
    if (p32) {
        if (odd) DPCG_K3D_MANY(true, true);
        else DPCG_K3D_MANY(true, false);
    } else {
        if (odd) DPCG_K3D_MANY(false, true);
        else DPCG_K3D_MANY(false, false);
    }
#undef DPCG_K3D_MANY
#undef DPCG_K3D
}

// End of a solve in the deferred-x form: after an odd number of updates x still lacks alpha_{k-1} p_{k-1}; then the
// status bookkeeping of k_final_check.
__global__ __launch_bounds__(kBlock) void k_final_deferred(int64_t n, Scalars *sc, double *__restrict__ x,
                                                           const double *__restrict__ p0, const double *__restrict__ p1) {
    const int k = sc->k;
    if (k & 1) {
        const double alpha = sc->alpha;
        const double *__restrict__ p = ((k - 1) & 1) ? p1 : p0;
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] = x[i] + alpha * p[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && !sc->done) {
        sc->status = DPCG_MAX_ITER;
        sc->done = 1;
    }
}

void launch_final_deferred(int64_t n, Scalars *scal, double *x, const double *p0, const double *p1, int grid,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_final_deferred, dim3(grid), dim3(kBlock), 0, s, n, scal, x, p0, p1);
}

// After the last permitted update (cg.py:70 exhausted): the test has already been recorded by K3.
__global__ void k_final_check(Scalars *sc) {
    if (threadIdx.x == 0 && !sc->done) {
        sc->status = DPCG_MAX_ITER;
        sc->done = 1;
    }
}

void launch_final_check(Scalars *scal, hipStream_t s) {
    hipLaunchKernelGGL(k_final_check, dim3(1), dim3(64), 0, s, scal);
}

// Start of a solve (cg.py:62-66): p = z, partials of <b,b>, <r,z> and of the first tested quantity
// (<z,z> for the reference's quirk at cg.py:66, <r,r> for scipy-style).
template <bool P32>
__global__ __launch_bounds__(kBlock) void k_init_state(int64_t n, const double *__restrict__ b,
                                                       const double *__restrict__ r, const double *__restrict__ z,
                                                       double *__restrict__ p, float *__restrict__ p32,
                                                       double *__restrict__ part_bb, double *__restrict__ part_rz,
                                                       double *__restrict__ part_rr, int init_check_r) {
    __shared__ double sh[4];
    double a_bb = 0.0, a_rz = 0.0, a_t = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double bi = b[i], ri = r[i], zi = z[i];
        p[i] = zi;                                                      // cg.py:62
        if (P32) p32[i] = (float)zi;
        a_bb += bi * bi;
        a_rz += ri * zi;
        a_t += init_check_r ? ri * ri : zi * zi;                        // cg.py:66 tests zk
    }
    const double t_bb = block_sum(a_bb, sh);
    const double t_rz = block_sum(a_rz, sh);
    const double t_t = block_sum(a_t, sh);
    if (threadIdx.x == 0) {
        part_bb[blockIdx.x] = t_bb;
        part_rz[blockIdx.x] = t_rz;
        part_rr[blockIdx.x] = t_t;
    }
}

void launch_init_state(int64_t n, Scalars *scal, const double *b, const double *r, const double *z, double *p,
                       float *p32, double *part_bb, double *part_rz, double *part_rr, int init_check_r, int grid,
                       hipStream_t s) {
    (void)scal;
    if (p32)
        hipLaunchKernelGGL(k_init_state<true>, dim3(grid), dim3(kBlock), 0, s, n, b, r, z, p, p32, part_bb, part_rz,
                           part_rr, init_check_r);
    else
        hipLaunchKernelGGL(k_init_state<false>, dim3(grid), dim3(kBlock), 0, s, n, b, r, z, p, p32, part_bb, part_rz,
                           part_rr, init_check_r);
}

__global__ __launch_bounds__(kBlock) void k_finalize_init(Scalars *sc, const double *__restrict__ part_bb, double *part_rz,
                                                          double *part_t, int n_part, double rtol_sq, double atol_sq,
                                                          double *hist, int hist_cap, unsigned long long *progress,
                                                          int canon_cap) {
    __shared__ double sh[4];
    const double bb = reduce_partials(part_bb, n_part, sh);
    const double rz = reduce_partials(part_rz, n_part, sh);
    const double tt = reduce_partials(part_t, n_part, sh);             // <z0,z0> (cg.py:66) or <r0,r0>
    // canonical form for the first KA of a two-kernel solve, which may sum a different number of partials than the
    // initial state wrote: entry 0 = the sum (these very bits), the rest zero (x + 0.0 == x)
    __syncthreads();
    for (int i = threadIdx.x; i < canon_cap; i += kBlock) {
        part_rz[i] = i == 0 ? rz : 0.0;
        if (part_t != part_rz) part_t[i] = i == 0 ? tt : 0.0;
    }
    if (threadIdx.x == 0) {
        sc->bb = bb;
        sc->rz = rz;
        sc->alpha = 0.0;
        sc->rtol_sq = rtol_sq;
        sc->atol_sq = atol_sq;
        sc->done = 0;
        sc->status = DPCG_MAX_ITER;
        sc->done_seen = 0;
        sc->progress = progress;
        record_and_test(sc, tt, rz, hist, hist_cap, 0);                 // cg.py:66-67 and the first cg.py:71
    }
}

void launch_finalize_init(Scalars *scal, const double *part_bb, double *part_rz, double *part_t,
                          int n_part, double rtol_sq, double atol_sq, double *hist, int hist_cap,
                          unsigned long long *progress, hipStream_t s, int canon_cap) {
    hipLaunchKernelGGL(k_finalize_init, dim3(1), dim3(kBlock), 0, s, scal, part_bb, part_rz, part_t, n_part, rtol_sq,
                       atol_sq, hist, hist_cap, progress, canon_cap);
}

// r = b - A x0 (cg.py:60), ax = A x0 computed by the SpMV before.
__global__ __launch_bounds__(kBlock) void k_residual(int64_t n, const double *__restrict__ b,
                                                     const double *__restrict__ ax, double *__restrict__ r) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) r[i] = b[i] - ax[i];
}

void launch_residual(int64_t n, const double *b, const double *ax, double *r, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_residual, dim3(grid), dim3(kBlock), 0, s, n, b, ax, r);
}

// z = dinv .* r (Jacobi apply outside the fused path).
__global__ __launch_bounds__(kBlock) void k_scale(int64_t n, const double *__restrict__ dinv,
                                                  const double *__restrict__ r, double *__restrict__ z) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) z[i] = dinv[i] * r[i];
}

void launch_scale(int64_t n, const double *dinv, const double *r, double *z, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_scale, dim3(grid), dim3(kBlock), 0, s, n, dinv, r, z);
}

// dinv[i] = 1 / a_ii (test.py:76); flags a missing or non-positive diagonal.
__global__ __launch_bounds__(kBlock) void k_extract_dinv(int64_t n, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, double *__restrict__ dinv,
                                                         int *bad) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double d = 0.0;
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (col[k] == i) d = val[k];
        if (!(d > 0.0)) atomicExch(bad, 1);
        dinv[i] = 1.0 / d;
    }
}

void launch_extract_dinv(const CsrDev &A, double *dinv, int *bad_flag, hipStream_t s) {
    int64_t g = (A.n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_extract_dinv, dim3((int)g), dim3(kBlock), 0, s, A.n, A.rowptr, A.col, A.val, dinv, bad_flag);
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void k_convert(int64_t n, const TI *__restrict__ in, TO *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = (TO)in[i];
}

static int conv_grid(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
void launch_f64_to_f32(int64_t n, const double *in, float *out, hipStream_t s) {
    hipLaunchKernelGGL((k_convert<double, float>), dim3(conv_grid(n)), dim3(kBlock), 0, s, n, in, out);
}
void launch_f32_to_f64(int64_t n, const float *in, double *out, hipStream_t s) {
    hipLaunchKernelGGL((k_convert<float, double>), dim3(conv_grid(n)), dim3(kBlock), 0, s, n, in, out);
}

// e = x - x_true (cg.py:27,43): only for conjugate_gradient(..., x_true=...).
__global__ __launch_bounds__(kBlock) void k_anorm_err(int64_t n, const Scalars *__restrict__ sc,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ x_true, double *__restrict__ e) {
    (void)sc;  // runs even when `done` is set: the iterate the test fired on still gets its error recorded
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) e[i] = x[i] - x_true[i];
}

void launch_anorm_err(int64_t n, const Scalars *scal, const double *x, const double *x_true, double *e, int grid,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_anorm_err, dim3(grid), dim3(kBlock), 0, s, n, scal, x, x_true, e);
}

// err_hist[k] = <e, A e> (cg.py:29,45) for the iterate the loop has just produced.
__global__ __launch_bounds__(kBlock) void k_record_err(const Scalars *__restrict__ sc, const double *__restrict__ part,
                                                       int n_part, double *err_hist, int hist_cap) {
    __shared__ double sh[4];
    const double v = reduce_partials(part, n_part, sh);
    if (threadIdx.x == 0 && sc->k < hist_cap) err_hist[sc->k] = v;
}

void launch_record_err(const Scalars *scal, const double *part, int n_part, double *err_hist, int hist_cap,
                       int at_k_minus_one, hipStream_t s) {
    (void)at_k_minus_one;
    hipLaunchKernelGGL(k_record_err, dim3(1), dim3(kBlock), 0, s, scal, part, n_part, err_hist, hist_cap);
}

__global__ __launch_bounds__(kBlock) void k_dot_final(const double *__restrict__ part, int n_part, double *out) {
    __shared__ double sh[4];
    const double v = reduce_partials(part, n_part, sh);
    if (threadIdx.x == 0) *out = v;
}

void launch_dot_final(const double *part, int n_part, double *out_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kBlock), 0, s, part, n_part, out_dev);
}

}  // namespace dpcg
