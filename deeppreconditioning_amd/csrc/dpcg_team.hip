// Whole-solve kernel for MID-SIZE systems (6 145 .. 65 536 rows): one TEAM of 32 workgroups per system, up to eight
// systems per launch -- hand-written for gfx950 (MI355X: 256 CUs in 8 XCDs of 32, wave64).
//
// A PCG update of such a system is two or three launches whose work (a 5 MB SpMV out of L2) takes 1-2 us each while every
// kernel boundary costs ~2 us and the ramp of a 256-CU grid as much again: 9.6 us per update at 65 536 rows (BASELINE
// config 2), 0.13 of the HBM roofline, more than half the chip idle.  Here the whole solve (cg.py:58-90) is ONE launch:
//   * 8 x 32 workgroups of 1024 threads, one per CU; workgroup b belongs to team b & 7 with rank b >> 3.  Workgroups are
//     dealt round-robin over the XCDs, so a team normally sits on ONE XCD -- its 32 CUs share that XCD's 4 MiB L2, which
//     holds the team's matrix slice and vectors.  Every workgroup reports its XCD (XCC_ID) at the start of a solve: when all 32
//     agree, what the team publishes (z, p, the reduction slots) is stored plainly -- write-back into the shared L2, where the
//     readers' agent-scope loads find it (0.26 instead of 0.46 us per hand-off; 256^2: 10.8 -> 8.5 us per update) --, otherwise
//     with agent scope, so any placement is correct, the usual one fast.
//   * a workgroup owns whole 1024-row slabs of the slab-ELL copy of A (dpcg_small.hip: lanes read consecutive
//     addresses; a row's entries keep their CSR order, so q = A p is the oracle's row sum bit for bit); x, r, p, q of its
//     rows live in registers for the whole solve; z and p are also published to L2 because the neighbours gather them.
//   * TWO team barriers per update (<p,Ap> partials | z, p published and <r,z>, <r,r> partials): the entries of p_k that a
//     row gathers are recomputed as z_k[c] + beta p_{k-1}[c] from the published vectors (the owner's expression and bits), so
//     publishing needs no barrier of its own; the matrix slice of a workgroup (rows of <= 7 entries) is read ONCE, into LDS.
//     The reductions double as the barriers: a workgroup's partial is ONE 8-byte agent-scope store into a slot that held a
//     reserved NaN pattern, and wave 0 of every workgroup polls the 2 x 32 slots themselves -- no counter, no atomic, no second
//     round trip for the values (three slot sets rotate; a workgroup re-arms the next set as it writes).  Only the x0 path uses
//     the counter barrier: one agent-scope
//     atomic add per workgroup on the team's counter after every wave has drained its stores (s_waitcnt vmcnt(0)), then a
//     relaxed agent-scope poll by one lane; all shared data (p, the 32 partials per reduction) is stored and loaded with
//     agent scope (sc1: served by the L2, never by a stale L1 line).  Reductions are two-stage and ordered (wave DPP tree
//     -> 16 wave sums -> 32 workgroup partials summed in rank order by every workgroup), so all workgroups hold bit-identical
//     alpha / beta and take the stopping decision of cg.py:71 identically, with no broadcast.
//   * the launch is refused up front when the occupancy query does not admit the kernel on a CU, and every wait is bounded by wall
//     time (20 ms; a healthy hand-off takes 0.3-0.5 us): a team whose member never arrives -- somebody else holding CUs -- reports
//     DPCG_ERR_STATE instead of hanging, and the caller solves through the multi-launch path.
// M is I or Jacobi (what a mid-size system is solved with when setup time matters); everything else keeps the multi-launch path.
#include <algorithm>

#include "dpcg_device.h"

namespace dpcg {

constexpr int kTeamSize = 32;          // workgroups per team (the CUs of one XCD)
constexpr int kTeams = 8;              // teams per launch (the XCDs)
constexpr int kTeamThreads = 1024;
constexpr int kTeamMaxSlabs = 2;       // 1024-row slabs per workgroup: n <= 32 * 2 * 1024 (four slabs spill: 172 B of scratch per lane)


namespace {

constexpr unsigned long long kTeamSpinTicks = 2000000ull;              // 20 ms of the 100 MHz constant clock (a healthy hand-off takes 0.3-0.5 us)
constexpr unsigned long long kTeamPending = 0x7ff8dead0badbeefULL;     // a quiet NaN that no arithmetic here produces: "slot not written yet"

__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Between workgroups of ONE XCD a plain (write-back) store is enough: they share that XCD's L2, which then holds the line the
// readers' agent-scope loads ask for -- the store need not travel to the memory side first (tools/hop_lab: 0.26 us per hand-off
// instead of 0.46).  `local` is established per solve (see the kernel): every workgroup of the team on the same XCD.
__device__ __forceinline__ void st_team(double *p, double v, bool local) {
    if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int team_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 0xf);
}
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

// Sum over the 1024 threads of the workgroup in a fixed order; every thread gets the result.  sh: 2 x 32 doubles, the two halves
// used in turn (`phase`), so ONE barrier per sum suffices: whoever writes a half again has passed the barrier of the sum in between,
// behind which every thread had read it.  (A workgroup barrier here also waits for the wave's outstanding stores: ~0.35 us each in
// the update loop, DPCG_TEAM_TRACE.)
__device__ __forceinline__ double team_block_sum(double v, double *sh, int &phase) {
    double *slot = sh + (phase & 1) * 32;
    ++phase;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 63) slot[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kTeamThreads / 64; ++w) s += slot[w];
    return s;
}

// Two such sums behind the one barrier (each in the order of team_block_sum).
__device__ __forceinline__ void team_block_sum2(double &a, double &b, double *sh, int &phase) {
    double *slot = sh + (phase & 1) * 32;
    ++phase;
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 63) {
        slot[threadIdx.x >> 6] = a;
        slot[16 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < kTeamThreads / 64; ++w) {
        sa += slot[w];
        sb += slot[16 + w];
    }
    a = sa;
    b = sb;
}

// RPT: 1024-row slabs per workgroup; WMAX: entries per row.  The matrix slice is read ONCE per solve into LDS -- used as
// per-thread private storage (slot [k][j][t] belongs to thread t: conflict-free, no barrier): values fp64, columns 16-bit
// (n <= 65 536) -- 10 bytes per entry, 143 KB for two slabs of 7-entry rows; in registers the same slice spilled (260 B of
// scratch per lane at 128 VGPRs).
template <int RPT, int WMAX, bool TRACE>   // TRACE (DPCG_TEAM_TRACE): phase timers -- a variant of its own, so that the timers' registers are not the solve's
__global__ __launch_bounds__(kTeamThreads) void k_pcg_team(const TeamDesc *__restrict__ descs, int nsys) {
    const int team = blockIdx.x & (kTeams - 1), rank = blockIdx.x >> 3;
    if (team >= nsys || rank >= kTeamSize) return;
    extern __shared__ __attribute__((aligned(16))) double team_smem[];
    double *lv = team_smem;                                                        // [RPT][WMAX][1024] values
    unsigned short *lc = reinterpret_cast<unsigned short *>(lv + RPT * WMAX * kTeamThreads);   // [RPT][WMAX][1024] columns
    __shared__ double sh[64];
    int sum_phase = 0;
    __shared__ double s_part[2][2 * kTeamSize];      // (two sets in turn: no barrier behind the last read of one)
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
#pragma unroll
        for (int j = 0; j < WMAX; ++j) {
            const bool on = j < len[k];
            const size_t o = ((size_t)(valid ? slab : 0) * d.W + (on ? j : 0)) * kTeamThreads + t;
            lc[(k * WMAX + j) * kTeamThreads + t] = (unsigned short)(on ? d.ell_col[o] : 0);
            lv[(k * WMAX + j) * kTeamThreads + t] = on ? d.ell_val[o] : 0.0;
        }
    }
    unsigned int target = 0;
    bool alive = true;
    // is the whole team on one XCD (the usual placement: workgroups are dealt round-robin over the XCDs and a team's share one
    // residue)?  Every workgroup reports its XCD, the team meets once, everybody reads the 32 answers.  DPCG_TEAM_LOCAL=0 (passed as
    // xcc == null) keeps agent-scope stores.
    bool local = false;
    if (d.xcc) {
        if (t == 0) __hip_atomic_store(d.xcc + rank, team_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alive = team_barrier(d.bar, target, d.err, &s_flag);
        if (alive) {
            const int mine = __hip_atomic_load(d.xcc + rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int same = 1;
            for (int w = 0; w < kTeamSize; ++w)
                same &= __hip_atomic_load(d.xcc + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine ? 1 : 0;
            local = same != 0;
        }
    }
    // q = A p_k for the own rows.  p_k is not stored anywhere as a whole: the gathered entries are RECOMPUTED as
    // Z[c] + beta * Pold[c] from the published z_k and p_{k-1} -- the expression (and, with contraction off, the bits) of the
    // owner's p_k = z + beta p_{k-1} (cg.py:83) -- so that publishing and the <r,z> reduction share ONE barrier.  All 2 * WMAX
    // gathers of a row are in flight together (one round trip to the L2); row sums in CSR order.
    auto spmv = [&](const double *__restrict__ Z, const double *__restrict__ Pold, double beta) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            double zc[WMAX], pc[WMAX];
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int c = lc[(k * WMAX + j) * kTeamThreads + t];
                zc[j] = ld_agent(Z + c);
                pc[j] = ld_agent(Pold + c);
            }
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < WMAX; ++j)
                if (j < len[k]) acc += lv[(k * WMAX + j) * kTeamThreads + t] * (zc[j] + beta * pc[j]);
            q[k] = acc;
        }
    };
    // Two team-wide sums at once, and a team barrier in the same breath.  The 2 x 32 partials ARE the flags: a slot holds a
    // reserved NaN pattern until its owner stores the value (one 8-byte agent-scope store), and the readers -- lanes 0-63 of
    // wave 0, one slot each -- poll the slots themselves: no counter, no atomic, no second round trip to fetch the values.
    // FOUR slot sets rotate: generation g uses set g % 4 and, writing it, a workgroup re-arms its slots of set (g + 2) % 4 --
    // last used by generation g - 2, which everybody finished reading before anybody could write generation g - 1 -- i.e. TWO
    // generations ahead of their next use, and the publishing lane drains its earlier stores before it stores a value: whoever
    // has seen this workgroup's generation-g value knows that its re-arm of set (g + 1) % 4 (issued at generation g - 1) has
    // landed, so a reader that moves on to generation g + 1 cannot find a stale, non-pending value there.  (With three sets the
    // re-arm of the NEXT set and the current value were two unordered stores to different addresses -- other channel, other XCD:
    // a reader could have summed the value of three generations ago into alpha / beta.)  `publish`: the workgroup's stores of z
    // and p must be visible to whoever passes this point, so every wave drains them first.  Every workgroup returns the same bits.
    unsigned int gen = 0;
    const bool timed = TRACE && d.dbg != nullptr && rank == 0 && t == 0;     // DPCG_TEAM_TRACE: where an update's time goes (ticks of rank 0)
    unsigned long long tk_drain = 0, tk_bsum = 0;
    auto team_sum2 = [&](double a, double b2, bool publish, double &ra, double &rb, bool two = true) -> bool {
        unsigned long long q0 = timed ? wall_clock64() : 0;
        if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (timed) { const unsigned long long q1 = wall_clock64(); tk_drain += q1 - q0; q0 = q1; }
        // (a workgroup barrier inside: behind every wave's drain.  One sum, or two behind the same barrier: a block sum with two barriers
        // was ~0.6 us of rank 0's time, DPCG_TEAM_TRACE -- four of them 2.5 of the 8.9 us of an update at 10K rows)
        if (two) team_block_sum2(a, b2, sh, sum_phase);
        else a = team_block_sum(a, sh, sum_phase);
        if (timed) tk_bsum += wall_clock64() - q0;
        const double pend = __longlong_as_double((long long)kTeamPending);
        double *cur = d.part + (gen & 3) * (2 * kTeamSize), *nxt = d.part + ((gen + 2) & 3) * (2 * kTeamSize);
        double *const sp = s_part[gen & 1];
        ++gen;
        if (t == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the re-arm stores of the previous generation have landed
            st_team(nxt + rank, pend, local);
            st_team(nxt + kTeamSize + rank, pend, local);
            st_team(cur + rank, a, local);
            st_team(cur + kTeamSize + rank, b2, local);
        }
        if (t < 2 * kTeamSize) {                                  // wave 0: lane l polls slot l
            double v = ld_agent(cur + t);
            unsigned spins = 0;
            unsigned long long t0 = 0;
            int ok = 1;
            while (__ballot((unsigned long long)__double_as_longlong(v) == kTeamPending) != 0) {
                __builtin_amdgcn_s_sleep(1);
                if ((unsigned long long)__double_as_longlong(v) == kTeamPending) v = ld_agent(cur + t);
                if ((++spins & 1023u) == 0) {
                    const unsigned long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > kTeamSpinTicks || __hip_atomic_load(d.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        atomicExch(d.err, 1);
                        ok = 0;
                        break;
                    }
                }
            }
            sp[t] = v;
            if (t == 0) s_flag = ok;
        }
        __syncthreads();
        if (!s_flag) return false;
        ra = rb = 0.0;
#pragma unroll
        for (int w = 0; w < kTeamSize; ++w) {
            ra += sp[w];
            rb += sp[kTeamSize + w];
        }
        return true;       // (this set is written again two reductions on, behind the block sums' barriers of the next one)
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
    if (d.x0) {                                                   // r = b - A x0 (cg.py:60): x0 published as "z", beta = 0
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row[k] >= 0) {
                st_agent(d.z + row[k], x[k]);
                st_agent(d.p1 + row[k], 0.0);
            }
        alive = team_barrier(d.bar, target, d.err, &s_flag);
        if (alive) {
            spmv(d.z, d.p1, 0.0);
#pragma unroll
            for (int k = 0; k < RPT; ++k) r[k] = r[k] - q[k];
            alive = team_barrier(d.bar, target, d.err, &s_flag);   // everybody has read x0 out of d.z before z_0 overwrites it
        }
    }
    double rz_loc = 0.0, t0_loc = 0.0;
    double z[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        z[k] = dv[k] * r[k];                                      // cg.py:61 (M = I: dv = 1, the product is exact)
        p[k] = z[k];                                              // cg.py:62
        if (row[k] >= 0) {
            rz_loc += r[k] * z[k];
            t0_loc += d.init_check_r ? r[k] * r[k] : z[k] * z[k];  // cg.py:66: the first test is on z
            st_team(d.z + row[k], z[k], local);                   // p_0 = z_0 + 0 * p_{-1} with p_{-1} = 0 (buffer 1)
            st_team(d.p1 + row[k], 0.0, local);
        }
    }
    double bb = 0.0, dummy = 0.0, rz = 0.0, tt = 0.0;
    if (alive) alive = team_sum2(bb_loc, rz_loc, true, bb, rz);    // z_0 and p_{-1} = 0 are published behind this point
    if (alive) alive = team_sum2(t0_loc, 0.0, false, tt, dummy, false);
    double res = tt / bb, beta = 0.0;
    int k_done = 0, status = DPCG_MAX_ITER;
    bool stop = false;
    if (alive) {
        if (rank == 0 && t == 0 && d.hist_cap > 0) d.hist[0] = res;
        const bool conv = (res < d.rtol_sq) || (tt < d.atol_sq);
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
    // ---- cg.py:70-87: two team barriers per update ---------------------------------------------------------------------
    unsigned long long tk[5] = {0, 0, 0, 0, 0}, tk_bsum_a = 0;
    const unsigned long long tk_start = timed ? wall_clock64() : 0;
    tk_drain = tk_bsum = 0;
    while (alive && !stop && k_done < d.max_iter) {
        // update k_done: p_k = z_k + beta_k p_{k-1}; the neighbours' entries from Z = z_k and P[(k+1) & 1] = p_{k-1}
        double *Pold = (k_done & 1) ? d.p0 : d.p1, *Pnew = (k_done & 1) ? d.p1 : d.p0;
        unsigned long long c0 = timed ? wall_clock64() : 0;
        spmv(d.z, Pold, beta);                                    // cg.py:75
        double pq_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row[k] >= 0) pq_loc += q[k] * p[k];
        double pq = 0.0;
        if (timed) { const unsigned long long c1 = wall_clock64(); tk[0] += c1 - c0; c0 = c1; }
        const unsigned long long bs0 = tk_bsum;
        if (!(alive = team_sum2(pq_loc, 0.0, false, pq, dummy, false))) break;     // barrier A: every SpMV of this update is done
        if (timed) { const unsigned long long c1 = wall_clock64(); tk[1] += c1 - c0; c0 = c1; tk_bsum_a += tk_bsum - bs0; }
        const double alpha = rz / pq;                             // cg.py:78
        double rz_new_loc = 0.0, rr_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            x[k] = x[k] + alpha * p[k];                           // cg.py:79
            r[k] = r[k] - alpha * q[k];                           // cg.py:80
            z[k] = dv[k] * r[k];                                  // cg.py:81
            if (row[k] >= 0) {
                rz_new_loc += r[k] * z[k];
                rr_loc += r[k] * r[k];
                st_team(d.z + row[k], z[k], local);               // z_{k+1} and p_k for the neighbours' next gathers
                st_team(Pnew + row[k], p[k], local);
            }
        }
        double rz_new = 0.0, rr = 0.0;
        if (timed) { const unsigned long long c1 = wall_clock64(); tk[2] += c1 - c0; c0 = c1; }
        if (!(alive = team_sum2(rz_new_loc, rr_loc, true, rz_new, rr))) break;   // barrier B: z, p published and <r,z>, <r,r> known
        if (timed) tk[3] += wall_clock64() - c0;
        beta = rz_new / rz;                                       // cg.py:82
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
        if (alive && row[k] >= 0) d.x[row[k]] = x[k];
    if (timed) {
        d.dbg[0] = tk[0]; d.dbg[1] = tk[1]; d.dbg[2] = tk[2]; d.dbg[3] = tk[3]; d.dbg[4] = wall_clock64() - tk_start;
        d.dbg[5] = tk_bsum_a; d.dbg[6] = tk_drain; d.dbg[7] = tk_bsum - tk_bsum_a;
    }
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

int team_max_row_len() { return 7; }

// nsys <= 8 systems; descs_dev: device array of TeamDesc.  max_row_len: longest row of any of them (<= 7: the matrix slice lives
// in registers).  Returns DPCG_OK or a negative status.
int launch_pcg_team(const TeamDesc *descs_dev, int nsys, int max_slabs_per_wg, int max_row_len, hipStream_t s, bool trace) {
    const dim3 grid(kTeams * kTeamSize), block(kTeamThreads);
    if (max_slabs_per_wg > kTeamMaxSlabs || max_row_len > 7) return DPCG_ERR_INVALID;
    // Co-residency is what the team's waits assume: the occupancy query must admit the kernel on a CU (with the CU count >= 256 checked by
    // the caller that is every workgroup of the grid); a kernel that does not fit is refused up front (DPCG_ERR_STATE: the caller takes
    // the multi-launch path), and every wait inside is bounded (20 ms) for the case that somebody else holds CUs.
#define DPCG_TEAM_LAUNCH_T(RPTV, WV, TV)                                                                                   \
    do {                                                                                                                   \
        const int lds = RPTV * WV * kTeamThreads * 10;                                                                     \
        static int resident = -1;                                                                                          \
        if (resident < 0) {                                                                                                \
            if (hipFuncSetAttribute((const void *)k_pcg_team<RPTV, WV, TV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != \
                hipSuccess)                                                                                                \
                return DPCG_ERR_HIP;                                                                                       \
            int per_cu = 0;                                                                                                \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_pcg_team<RPTV, WV, TV>, kTeamThreads, \
                                                             (size_t)lds) != hipSuccess)                                   \
                return DPCG_ERR_HIP;                                                                                       \
            resident = per_cu;                                                                                             \
        }                                                                                                                  \
        if (resident < 1) return DPCG_ERR_STATE;                                                                           \
        hipLaunchKernelGGL((k_pcg_team<RPTV, WV, TV>), grid, block, (size_t)lds, s, descs_dev, nsys);                      \
    } while (0)
#define DPCG_TEAM_LAUNCH(RPTV, WV)                     \
    do {                                               \
        if (trace) DPCG_TEAM_LAUNCH_T(RPTV, WV, true); \
        else DPCG_TEAM_LAUNCH_T(RPTV, WV, false);      \
    } while (0)
    if (max_slabs_per_wg <= 1) {
        if (max_row_len <= 5) DPCG_TEAM_LAUNCH(1, 5);
        else DPCG_TEAM_LAUNCH(1, 7);
    } else {
        if (max_row_len <= 5) DPCG_TEAM_LAUNCH(2, 5);
        else DPCG_TEAM_LAUNCH(2, 7);
    }
#undef DPCG_TEAM_LAUNCH
#undef DPCG_TEAM_LAUNCH_T
    return DPCG_OK;
}

}  // namespace dpcg
