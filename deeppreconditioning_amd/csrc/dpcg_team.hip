// Whole-solve kernel for MID-SIZE systems (6 145 .. 65 536 rows): one TEAM of 32 workgroups per system, up to eight
// systems per launch -- hand-written for gfx950 (MI355X: 256 CUs in 8 XCDs of 32, wave64).
//
// A PCG update of such a system is two or three launches whose work (a 5 MB SpMV out of L2) takes 1-2 us each while every
// kernel boundary costs ~2 us and the ramp of a 256-CU grid as much again: 9.6 us per update at 65 536 rows (BASELINE
// config 2), 0.13 of the HBM roofline, more than half the chip idle.  Here the whole solve (cg.py:58-90) is ONE launch:
//   * 8 x 32 workgroups of 1024 threads, one per CU; workgroup b belongs to team b & 7 with rank b >> 3.  Workgroups are
//     dealt round-robin over the XCDs, so a team normally sits on ONE XCD -- its 32 CUs share that XCD's 4 MiB L2, which
//     holds the team's matrix slice and vectors -- but nothing depends on it: every hand-off below is agent-scope, so any
//     placement is correct, the usual one merely fast.
//   * a workgroup owns whole 1024-row slabs of the slab-ELL copy of A (dpcg_small.hip: lanes read consecutive
//     addresses; a row's entries keep their CSR order, so q = A p is the oracle's row sum bit for bit); x, r, p, q of its
//     rows live in registers for the whole solve; p is also published to HBM/L2 because the neighbours gather it.
//   * three team barriers per update (p published | <p,Ap> partials | <r,z>, <r,r> partials).  A barrier is one agent-scope
//     atomic add per workgroup on the team's counter after every wave has drained its stores (s_waitcnt vmcnt(0)), then a
//     relaxed agent-scope poll by one lane; all shared data (p, the 32 partials per reduction) is stored and loaded with
//     agent scope (sc1: served by the L2, never by a stale L1 line).  Reductions are two-stage and ordered (wave DPP tree
//     -> 16 wave sums -> 32 workgroup partials summed in rank order by every workgroup), so all workgroups hold bit-identical
//     alpha / beta and take the stopping decision of cg.py:71 identically, with no broadcast.
//   * every wait is bounded by wall time (4 s): a team whose member never arrives reports DPCG_ERR_STATE instead of hanging.
// M is I or Jacobi (what a mid-size system is solved with when setup time matters); everything else keeps the multi-launch path.
#include <algorithm>

#include "dpcg_device.h"

namespace dpcg {

constexpr int kTeamSize = 32;          // workgroups per team (the CUs of one XCD)
constexpr int kTeams = 8;              // teams per launch (the XCDs)
constexpr int kTeamThreads = 1024;
constexpr int kTeamMaxSlabs = 2;       // 1024-row slabs per workgroup: n <= 32 * 2 * 1024 (four slabs spill: 172 B of scratch per lane)


namespace {

constexpr unsigned long long kTeamSpinTicks = 4ull * 100000000ull;     // 4 s of the 100 MHz constant clock

__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) {
    return __hip_atomic_load(const_cast<double *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every wave drains its own stores, the workgroup meets, ONE lane arrives (agent-scope add) and polls (relaxed agent-scope
// load) until all kTeamSize workgroups have; the others wait at the workgroup barrier that lane joins afterwards.
// Returns false when the wait ran out (or another workgroup's did).
__device__ __forceinline__ bool team_barrier(unsigned int *bar, unsigned int &target, int *err, int *s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    target += kTeamSize;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        unsigned long long t0 = 0;
        int ok = 1;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > kTeamSpinTicks) {
                    atomicExch(err, 1);
                    ok = 0;
                    break;
                }
            }
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

// Sum over the 1024 threads of the workgroup in a fixed order; every thread gets the result.  sh: 16 doubles.
__device__ __forceinline__ double team_block_sum(double v, double *sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kTeamThreads / 64; ++w) s += sh[w];
    return s;
}

template <int RPT>
__global__ __launch_bounds__(kTeamThreads) void k_pcg_team(const TeamDesc *__restrict__ descs, int nsys) {
    const int team = blockIdx.x & (kTeams - 1), rank = blockIdx.x >> 3;
    if (team >= nsys || rank >= kTeamSize) return;
    __shared__ double sh[16];
    __shared__ double s_part[2 * kTeamSize];
    __shared__ int s_flag;
    const TeamDesc d = descs[team];
    const int t = threadIdx.x;
    const int slabs = (d.n + kTeamThreads - 1) / kTeamThreads;
    const int per = (slabs + kTeamSize - 1) / kTeamSize;            // <= RPT (checked by the launcher)
    int row[RPT], len[RPT];
    double x[RPT], r[RPT], p[RPT], q[RPT], dv[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int slab = rank * per + k;
        const int i = slab * kTeamThreads + t;
        const bool valid = k < per && i < d.n;
        row[k] = valid ? i : -1;
        len[k] = valid ? d.rp[i + 1] - d.rp[i] : 0;
        x[k] = r[k] = p[k] = q[k] = 0.0;
        dv[k] = (valid && d.precond == DPCG_PRECOND_JACOBI) ? d.dinv[i] : 1.0;
    }
    unsigned int target = 0;
    bool alive = true;
    // q = A p for the own rows, p gathered with agent-scope loads (the other workgroups' stores), row sums in CSR order
    auto spmv = [&]() {
        int lmax = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            q[k] = 0.0;
            lmax = len[k] > lmax ? len[k] : lmax;
        }
#pragma unroll 2
        for (int j = 0; j < lmax; ++j) {
            int cc[RPT];
            double vv[RPT], pv[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const bool on = j < len[k];
                const size_t o = ((size_t)(row[k] < 0 ? 0 : row[k] / kTeamThreads) * d.W + (on ? j : 0)) * kTeamThreads + t;
                cc[k] = on ? d.ell_col[o] : 0;
                vv[k] = on ? d.ell_val[o] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) pv[k] = ld_agent(d.p + cc[k]);
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (j < len[k]) q[k] += vv[k] * pv[k];
        }
    };
    auto publish_p = [&]() {
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row[k] >= 0) st_agent(d.p + row[k], p[k]);
    };
    // two team-wide sums at once: slots s0, s1 of d.part; every workgroup returns the same bits
    auto team_sum2 = [&](double a, double b2, int s0, int s1, double &ra, double &rb) -> bool {
        a = team_block_sum(a, sh);
        b2 = team_block_sum(b2, sh);
        if (t == 0) {
            st_agent(d.part + s0 * kTeamSize + rank, a);
            st_agent(d.part + s1 * kTeamSize + rank, b2);
        }
        if (!team_barrier(d.bar, target, d.err, &s_flag)) return false;
        if (t < kTeamSize) {
            s_part[t] = ld_agent(d.part + s0 * kTeamSize + t);
            s_part[kTeamSize + t] = ld_agent(d.part + s1 * kTeamSize + t);
        }
        __syncthreads();
        ra = rb = 0.0;
#pragma unroll
        for (int w = 0; w < kTeamSize; ++w) {
            ra += s_part[w];
            rb += s_part[kTeamSize + w];
        }
        __syncthreads();
        return true;
    };
    // ---- cg.py:58-67 -------------------------------------------------------------------------------------------------
    double bb_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (row[k] >= 0) {
            const double bi = d.b[row[k]];
            bb_loc += bi * bi;
            r[k] = bi;
            if (d.x0) x[k] = d.x0[row[k]];
        }
    if (d.x0) {                                                   // r = b - A x0 (cg.py:60)
#pragma unroll
        for (int k = 0; k < RPT; ++k) p[k] = x[k];
        publish_p();
        alive = team_barrier(d.bar, target, d.err, &s_flag);
        if (alive) {
            spmv();
#pragma unroll
            for (int k = 0; k < RPT; ++k) r[k] = r[k] - q[k];
            alive = team_barrier(d.bar, target, d.err, &s_flag);   // everybody has read x0 out of d.p before p overwrites it
        }
    }
    double rz_loc = 0.0, t0_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const double z = dv[k] * r[k];                            // cg.py:61 (M = I: dv = 1, the product is exact)
        p[k] = z;                                                 // cg.py:62
        if (row[k] >= 0) {
            rz_loc += r[k] * z;
            t0_loc += d.init_check_r ? r[k] * r[k] : z * z;        // cg.py:66: the first test is on z
        }
    }
    double bb = 0.0, dummy = 0.0, rz = 0.0, tt = 0.0;
    if (alive) alive = team_sum2(bb_loc, rz_loc, 0, 1, bb, rz);
    // (a slot is written again only after a barrier that every workgroup reaches AFTER its reads of the slot; the
    // second operand of a one-value sum goes to slot 3, which nobody looks at)
    if (alive) alive = team_sum2(t0_loc, 0.0, 2, 3, tt, dummy);
    double res = tt / bb;
    int k_done = 0, status = DPCG_MAX_ITER;
    bool stop = false;
    if (alive) {
        if (rank == 0 && t == 0 && d.hist_cap > 0) d.hist[0] = res;
        const bool conv = (res < d.rtol_sq) || (tt < d.atol_sq);
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
    // ---- cg.py:70-87 -------------------------------------------------------------------------------------------------
    while (alive && !stop && k_done < d.max_iter) {
        publish_p();
        if (!(alive = team_barrier(d.bar, target, d.err, &s_flag))) break;
        spmv();                                                   // cg.py:75
        double pq_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row[k] >= 0) pq_loc += q[k] * p[k];
        double pq = 0.0;
        if (!(alive = team_sum2(pq_loc, 0.0, 0, 3, pq, dummy))) break;
        const double alpha = rz / pq;                             // cg.py:78
        double rz_new_loc = 0.0, rr_loc = 0.0;
        double z[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            x[k] = x[k] + alpha * p[k];                           // cg.py:79
            r[k] = r[k] - alpha * q[k];                           // cg.py:80
            z[k] = dv[k] * r[k];                                  // cg.py:81
            if (row[k] >= 0) {
                rz_new_loc += r[k] * z[k];
                rr_loc += r[k] * r[k];
            }
        }
        double rz_new = 0.0, rr = 0.0;
        if (!(alive = team_sum2(rz_new_loc, rr_loc, 1, 2, rz_new, rr))) break;
        const double beta = rz_new / rz;                          // cg.py:82
#pragma unroll
        for (int k = 0; k < RPT; ++k) p[k] = z[k] + beta * p[k];  // cg.py:83
        rz = rz_new;
        res = rr / bb;                                            // cg.py:86
        ++k_done;
        if (rank == 0 && t == 0 && k_done < d.hist_cap) d.hist[k_done] = res;
        const bool conv = (res < d.rtol_sq) || (rr < d.atol_sq);  // cg.py:71, tested before the next update's work
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (row[k] >= 0) d.x[row[k]] = x[k];
    if (rank == 0 && t == 0) {
        Scalars *sc = d.out;
        sc->k = k_done;
        sc->res = res;
        sc->bb = bb;
        sc->status = alive ? status : DPCG_ERR_STATE;
        sc->done = 1;
    }
}

}  // namespace

int team_max_rows() { return kTeamSize * kTeamMaxSlabs * kTeamThreads; }

// nsys <= 8 systems; descs_dev: device array of TeamDesc.  Returns DPCG_OK or a negative status.
int launch_pcg_team(const TeamDesc *descs_dev, int nsys, int max_slabs_per_wg, hipStream_t s) {
    const dim3 grid(kTeams * kTeamSize), block(kTeamThreads);
    if (max_slabs_per_wg <= 1) hipLaunchKernelGGL(k_pcg_team<1>, grid, block, 0, s, descs_dev, nsys);
    else if (max_slabs_per_wg == 2) hipLaunchKernelGGL(k_pcg_team<2>, grid, block, 0, s, descs_dev, nsys);
    else return DPCG_ERR_INVALID;
    return DPCG_OK;
}

}  // namespace dpcg
