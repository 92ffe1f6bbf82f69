// Level-scheduled sparse triangular solves and the IC(0) numeric factorisation.
// Compiled with -ffp-contract=off so that a*b+c is two roundings, as in the CPU reference path (scipy/ATen CSR row
// sums, unfused torch mul+add at cg.py:79-83); in-order sums then reproduce the oracle bit for bit.
#include <algorithm>
#include <cstdlib>

#include "dpcg_device.h"

namespace dpcg {

// ------------------------------------------------------------------------------------------------
// Level-scheduled sparse triangular solves (LLT_SOLVE: z = L^-T (L^-1 r)).
// Rows of one level are independent; one thread owns a row and subtracts its products in column
// order, then divides by the diagonal -- bit-identical to sequential substitution.
// Lower factor: diagonal LAST in the row.  Upper (L^T as CSR): diagonal FIRST.
// ------------------------------------------------------------------------------------------------
// One wide level, CSR-stream style: a workgroup takes 256 consecutive rows of the level-ordered copy, streams
// their contiguous val/col segment coalesced, parks v*out[col] in LDS and lets thread j subtract the products
// of row j in column order.  The diagonal's slot is skipped (its "product" is never read).
template <bool UPPER>
__global__ __launch_bounds__(kBlock) void k_sptrsv_level_stream(const int32_t *__restrict__ rows, int j0, int count,
                                                                const int32_t *__restrict__ lo_rp,
                                                                const int32_t *__restrict__ lo_ci,
                                                                const double *__restrict__ lo_v,
                                                                const double *__restrict__ rhs, double *out, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    constexpr int U = kStreamCap / kBlock;
    __shared__ double prod[kStreamCap];
    const int t = threadIdx.x;
    const int jb = j0 + blockIdx.x * kBlock;
    const int jend = (jb + kBlock < j0 + count) ? jb + kBlock : j0 + count;
    const int j = jb + t;
    const int base = lo_rp[jb];
    const int cnt = lo_rp[jend] - base;
    int rs = 0, re = 0, i = 0;
    double bi = 0.0;
    if (j < jend) {
        rs = lo_rp[j] - base;
        re = lo_rp[j + 1] - base;
        i = rows[j];
        bi = rhs[i];
    }
    const int last = cnt > 0 ? cnt - 1 : 0;
    int c[U];
    double a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = t + u * kBlock;
        const int kk = k < cnt ? k : last;
        c[u] = lo_ci[base + kk];
        a[u] = lo_v[base + kk];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = t + u * kBlock;
        if (k < cnt) prod[k] = a[u] * out[c[u]];
    }
    __syncthreads();
    if (j < jend) {
        double acc = bi;
        const int ks = UPPER ? rs + 1 : rs, ke = UPPER ? re : re - 1;
        for (int k = ks; k < ke; ++k) acc -= prod[k];
        out[i] = acc / lo_v[base + (UPPER ? rs : re - 1)];
    }
}

// One wide level on the fixed-width records of the sync-free kernel (Levels::sf_meta / sf_val): a thread owns a row, and all
// it needs -- three column indices, three values, the diagonal, its own index -- comes from loads addressed by the level-order
// position alone; then the right-hand side and the (already final) solution entries are gathered.  Two dependent memory
// round trips instead of the three of the LDS-staged kernel above (row extents -> entries -> gathers), no LDS and no
// barrier, hence 8 workgroups per CU in flight.  Same arithmetic and order.  Rows with more than three off-diagonal
// entries (meta.x == -2) walk the level-ordered copy.
// A row's fixed-width record (Levels::sf_meta / sf_val, width W = 3 or 6) in registers.
template <int W>
struct SfRec {
    int col[W];
    int own;
    double v[W];
    double diag;
};

template <int W>
__device__ __forceinline__ SfRec<W> load_sf_record(const int32_t *__restrict__ meta, const double *__restrict__ val, int64_t j) {
    constexpr int S = W == 3 ? 4 : (W == 6 ? 8 : 16);     // record stride: {c0..c(W-1), row, pad} / {v0..v(W-1), diagonal, pad}
    int m[S];
    double w[S];
    const int4 *mp = reinterpret_cast<const int4 *>(meta) + j * (S / 4);
    const double2 *vp = reinterpret_cast<const double2 *>(val) + j * (S / 2);
#pragma unroll
    for (int q = 0; q < S / 4; ++q) {
        const int4 t = mp[q];
        m[4 * q] = t.x; m[4 * q + 1] = t.y; m[4 * q + 2] = t.z; m[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int q = 0; q < S / 2; ++q) {
        if (2 * q > W) break;                             // (padding: not loaded)
        const double2 t = vp[q];
        w[2 * q] = t.x; w[2 * q + 1] = t.y;
    }
    SfRec<W> r;
#pragma unroll
    for (int q = 0; q < W; ++q) {
        r.col[q] = m[q];
        r.v[q] = w[q];
    }
    r.own = m[W];
    r.diag = w[W];
    return r;
}

template <bool UPPER, int W>
__global__ __launch_bounds__(kBlock) void k_sptrsv_level_rec(int j0, int count, const int32_t *__restrict__ lo_rp,
                                                             const int32_t *__restrict__ lo_ci,
                                                             const double *__restrict__ lo_v, const int32_t *__restrict__ meta,
                                                             const double *__restrict__ val,
                                                             const double *__restrict__ rhs, double *out, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= count) return;
    const int j = j0 + idx;
    const SfRec<W> r = load_sf_record<W>(meta, val, j);
    double acc = rhs[r.own];
    if (r.col[0] == -2) {
        const int s = lo_rp[j], e = lo_rp[j + 1];
        const int ks = UPPER ? s + 1 : s, ke = UPPER ? e : e - 1;
        for (int k = ks; k < ke; ++k) acc -= lo_v[k] * out[lo_ci[k]];
    } else {
        double y[W];
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = out[r.col[q] < 0 ? r.own : r.col[q]];
#pragma unroll
        for (int q = 0; q < W; ++q)
            if (r.col[q] >= 0) acc -= r.v[q] * y[q];
    }
    out[r.own] = acc / r.diag;
}

// One level, one thread per row (rows too long for the LDS product buffer).
template <bool UPPER>
__global__ __launch_bounds__(kBlock) void k_sptrsv_level(const int32_t *__restrict__ rows, int j0, int count,
                                                         const int32_t *__restrict__ lo_rp,
                                                         const int32_t *__restrict__ lo_ci,
                                                         const double *__restrict__ lo_v,
                                                         const double *__restrict__ rhs, double *out, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= count) return;
    const int j = j0 + idx, i = rows[j];
    const int s = lo_rp[j], e = lo_rp[j + 1];
    double acc = rhs[i];
    const int ks = UPPER ? s + 1 : s, ke = UPPER ? e : e - 1;
    for (int k = ks; k < ke; ++k) acc -= lo_v[k] * out[lo_ci[k]];
    out[i] = acc / lo_v[UPPER ? s : e - 1];
}

// A run of narrow levels walked by ONE workgroup of 1024 threads with a barrier between levels
// (cheaper than one launch per level: ~1.5 us boundary each).  Values handed from level to level
// go through L2 with agent-scope (sc1) accesses so no wave reads a stale line from its CU's L1.
constexpr int kMergedBlock = 1024;
template <bool UPPER>
__global__ __launch_bounds__(kMergedBlock) void k_sptrsv_merged(const int32_t *__restrict__ rows,
                                                                const int32_t *__restrict__ level_ptr, int lvl_lo,
                                                                int lvl_hi, const int32_t *__restrict__ lo_rp,
                                                                const int32_t *__restrict__ lo_ci,
                                                                const double *__restrict__ lo_v,
                                                                const double *__restrict__ rhs, double *out, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    for (int lvl = lvl_lo; lvl < lvl_hi; ++lvl) {
        const int lo = level_ptr[lvl], hi = level_ptr[lvl + 1];
        for (int j = lo + (int)threadIdx.x; j < hi; j += kMergedBlock) {
            const int i = rows[j];
            const int s = lo_rp[j], e = lo_rp[j + 1];
            double acc = rhs[i];
            const int ks = UPPER ? s + 1 : s, ke = UPPER ? e : e - 1;
            for (int k = ks; k < ke; ++k)
                acc -= lo_v[k] * __hip_atomic_load(out + lo_ci[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double res = acc / lo_v[UPPER ? s : e - 1];
            __hip_atomic_store(out + i, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();  // includes s_waitcnt vmcnt(0): this level's stores have reached L2
    }
}

// The same walk with the dependencies handed over through an LDS ring instead of L2.  In level order the
// entries a banded factor's rows depend on sit within the last W positions (for a 5-point grid: the previous
// anti-diagonal), so y lives in a ring of W doubles indexed by position & (W-1); entries solved before this
// segment are read from `out` (written by earlier kernels).  Everything that does not depend on y -- the row
// id, rhs, extents and the first three entries of the NEXT level's rows -- is loaded one level ahead, so a
// level costs an LDS round trip, a division and a barrier instead of three dependent L2 round trips.
constexpr int kRingE = 3;   // entries per row held in registers one level ahead
template <bool UPPER>
__global__ __launch_bounds__(kMergedBlock) void k_sptrsv_ring(const int32_t *__restrict__ rows,
                                                              const int32_t *__restrict__ level_ptr, int lvl_lo,
                                                              int lvl_hi, const int32_t *__restrict__ lo_rp,
                                                              const int32_t *__restrict__ lo_ci,
                                                              const int32_t *__restrict__ lo_cpos,
                                                              const double *__restrict__ lo_v,
                                                              const double *__restrict__ rhs, double *out,
                                                              int seg_start, int W, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    extern __shared__ __attribute__((aligned(16))) double ring[];
    struct Row {
        int j, i, s, e;
        double b, diag;
        int cpos[kRingE], ci[kRingE];
        double v[kRingE];
    };
    const int t = threadIdx.x;
    auto load_row = [&](Row &r, int j, int hi) {
        r.j = j < hi ? j : -1;
        if (r.j < 0) return;
        r.i = rows[j];
        r.s = lo_rp[j];
        r.e = lo_rp[j + 1];
        r.b = rhs[r.i];
        r.diag = lo_v[UPPER ? r.s : r.e - 1];
        const int ks = UPPER ? r.s + 1 : r.s, ke = UPPER ? r.e : r.e - 1;
#pragma unroll
        for (int m = 0; m < kRingE; ++m) {
            const int k = ks + m < ke ? ks + m : ks;      // clamped: an unused slot re-reads a valid entry
            r.cpos[m] = ks < ke ? lo_cpos[k] : 0;
            r.ci[m] = ks < ke ? lo_ci[k] : 0;
            r.v[m] = ks < ke ? lo_v[k] : 0.0;
        }
    };
    auto solve_row = [&](const Row &r) {
        if (r.j < 0) return;
        const int ks = UPPER ? r.s + 1 : r.s, ke = UPPER ? r.e : r.e - 1;
        double acc = r.b;
#pragma unroll
        for (int m = 0; m < kRingE; ++m)
            if (ks + m < ke) {
                const double yv = r.cpos[m] >= seg_start ? ring[r.cpos[m] & (W - 1)] : out[r.ci[m]];
                acc -= r.v[m] * yv;
            }
        for (int k = ks + kRingE; k < ke; ++k) {           // longer rows: the rest straight from memory
            const int cp = lo_cpos[k];
            const double yv = cp >= seg_start ? ring[cp & (W - 1)] : out[lo_ci[k]];
            acc -= lo_v[k] * yv;
        }
        const double y = acc / r.diag;
        ring[r.j & (W - 1)] = y;
        out[r.i] = y;
    };
    // Two register sets in ping-pong (no register moves: a move of a register with a load in flight would wait
    // for it).  Each thread owns at most two rows of a level (levels of a merged run have <= 2048 rows).
    auto load_level = [&](Row &a, Row &b, int lvl) {
        if (lvl < lvl_hi) {
            const int lo = level_ptr[lvl], hi = level_ptr[lvl + 1];
            load_row(a, lo + t, hi);
            load_row(b, lo + t + (int)blockDim.x, hi);
        } else {
            a.j = b.j = -1;
        }
    };
    // LDS-only hand-off: wait for this wave's ring writes, then the workgroup barrier.  A __syncthreads() would
    // also drain vmcnt, i.e. wait for the out[] store and for the prefetch loads that must stay in flight.
    auto level_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    Row a0, b0, a1, b1;
    load_level(a0, b0, lvl_lo);
    for (int lvl = lvl_lo; lvl < lvl_hi; lvl += 2) {
        load_level(a1, b1, lvl + 1);        // flies while level `lvl` is solved
        solve_row(a0);
        solve_row(b0);
        level_barrier();
        if (lvl + 1 >= lvl_hi) break;
        load_level(a0, b0, lvl + 2);
        solve_row(a1);
        solve_row(b1);
        level_barrier();
    }
}

// The ring walk with a deep software pipeline.  A level of a 5-point grid's factor is ~100 rows: the LDS round
// trip, three multiply-adds, the division and a barrier take ~0.2 us, but fetching a row's data only one level
// ahead (k_sptrsv_ring: row id -> extents -> entries, two dependent trips to L2) costs ~1.4 us per level.  Here a
// row's data sits in fixed-width records addressed by its level-order position alone (Levels::pk_meta / pk_val,
// right-hand side pre-gathered into level order), four 16-byte loads per row, and is requested D levels ahead;
// the segment's level offsets are staged in LDS up front.  Arithmetic and its order are those of every other
// SpTRSV kernel here (ascending columns, one product and one subtraction at a time, then the division).
__global__ __launch_bounds__(kBlock) void k_gather_lo(const int32_t *__restrict__ rows, const double *__restrict__ rhs,
                                                      double *__restrict__ b_lo, int j0, int count, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx < count) b_lo[j0 + idx] = rhs[rows[j0 + idx]];
}

// The prefetch is chunked: while the C levels of one chunk are solved out of one register set, the records of
// the next chunk fly into the other set; at a chunk boundary one s_waitcnt vmcnt(0) retires them (by then C levels
// of work have covered the memory latency).  The explicit wait + register "touch" keeps the compiler from
// inserting its own conservative vmcnt(0) at each first use, which would also wait for the loads just issued.
// FACTOR = 1: IC(0) of a cross-term-free pattern as a recurrence on the diagonals (see k_sptrsv_strips), for a factor whose
// whole schedule is this one walk: records = the entries of tril(A), ring / out = the diagonals of L, fac = the records of L.
// FACTOR = 2: the general incomplete factorisation on rows of at most three off-diagonal entries -- cross terms and ICT's
// drop rule included (the pattern with level-1 fill of a 5-point grid: the harness's default technique at its own size).
// The ring then holds a row's WHOLE record {l0, l1, l2, diagonal} (4 doubles per position); xdesc[j] names, for the entry
// pairs (0,1), (0,2), (1,2) of row j, the slot of column c_p in row c_q's record (2 bits each, 0 = no such entry: no cross
// term), thr[4j + q] = threshold * ||A(c_q:n, c_q)||_1 (null: nothing is dropped).  Operation order as k_ic0_level: entries
// in ascending columns; per entry the cross terms in ascending common column, one product and one subtraction at a time; the
// quotient; the drop test |v| * d < thr; the diagonal minus the squares of the KEPT values in order; the root.
template <bool UPPER, int C, int ROWS, int FACTOR = 0>   // ROWS rows of a level per thread, C levels per prefetch chunk
__global__ __launch_bounds__(512) void k_sptrsv_ring_pipe(const int32_t *__restrict__ level_ptr, int lvl_lo,
                                                          int lvl_hi, const int32_t *__restrict__ lo_rp,
                                                          const int32_t *__restrict__ lo_ci,
                                                          const int32_t *__restrict__ lo_cpos,
                                                          const double *__restrict__ lo_v,
                                                          const int4 *__restrict__ pk_meta,
                                                          const double2 *__restrict__ pk_val,
                                                          const double *__restrict__ b_lo, double *out, int seg_start,
                                                          int W, const int *done, double2 *__restrict__ fac = nullptr,
                                                          const int32_t *__restrict__ xdesc = nullptr,
                                                          const double2 *__restrict__ thr = nullptr) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    extern __shared__ __attribute__((aligned(16))) double ring[];
    int *lp = reinterpret_cast<int *>(ring + (FACTOR == 2 ? 4 * W : W));   // level offsets of this segment, padded with empty levels
    const int t = threadIdx.x, T = blockDim.x;
    const int nl = lvl_hi - lvl_lo;
    const int nchunks = (nl + C - 1) / C;
    const int seg_end = level_ptr[lvl_hi];
    for (int i = t; i <= (nchunks + 2) * C; i += T) lp[i] = i <= nl ? level_ptr[lvl_lo + i] : seg_end;
    __syncthreads();
    const int jmax = seg_end - 1;
    struct Row {
        int j;           // level-order position, -1 for a lane without a row in this level 
        int4 m;          // cpos0..2, original row
        double2 v01, v2d;
        double b;
        int xd;          // FACTOR 2: cross-term slots
        double2 t01;     // FACTOR 2 with a drop rule: thresholds of entries 0, 1
        double t2;       //                            ... and 2
    };
    auto load_row = [&](Row &r, int j, int hi) {
        const int jc = j < hi ? j : jmax;                 // lanes without a row load a valid record and ignore it
        r.j = j < hi ? j : -1;
        r.m = pk_meta[jc];
        r.v01 = pk_val[2 * (int64_t)jc];
        r.v2d = pk_val[2 * (int64_t)jc + 1];
        r.b = FACTOR ? 0.0 : b_lo[jc];
        r.xd = 0;
        r.t01 = make_double2(0.0, 0.0);
        r.t2 = 0.0;
        if (FACTOR == 2) {
            r.xd = xdesc[jc];
            if (thr) {
                r.t01 = thr[2 * (int64_t)jc];
                r.t2 = thr[2 * (int64_t)jc + 1].x;
            }
        }
    };
    auto load_chunk = [&](Row (&S)[C][ROWS], int chunk) {
#pragma unroll
        for (int d = 0; d < C; ++d) {
            const int rel = chunk * C + d;
            const int lo = lp[rel], hi = lp[rel + 1];
#pragma unroll
            for (int h = 0; h < ROWS; ++h) load_row(S[d][h], lo + t + h * T, hi);
        }
    };
    // all outstanding loads have landed; tell the compiler so (the registers are "redefined" here)
    auto retire = [&](Row (&S)[C][ROWS]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < C; ++d)
#pragma unroll
            for (int h = 0; h < ROWS; ++h) {
                Row &r = S[d][h];
                asm volatile("" : "+v"(r.m.x), "+v"(r.m.y), "+v"(r.m.z), "+v"(r.m.w), "+v"(r.v01.x), "+v"(r.v01.y),
                             "+v"(r.v2d.x), "+v"(r.v2d.y), "+v"(r.b));
                if (FACTOR == 2) asm volatile("" : "+v"(r.xd), "+v"(r.t01.x), "+v"(r.t01.y), "+v"(r.t2));
            }
    };
    auto solve_row = [&](const Row &r) {
        const bool valid = r.j >= 0;
        double acc = r.b;
        if (valid && r.m.x == -2) {                       // long row, or one that reaches back before the segment
            const int s = lo_rp[r.j], e = lo_rp[r.j + 1];
            const int ks = UPPER ? s + 1 : s, ke = UPPER ? e : e - 1;
            for (int k = ks; k < ke; ++k) {
                const int cp = lo_cpos[k];
                const double yv = cp >= seg_start ? ring[cp & (W - 1)] : out[lo_ci[k]];
                acc -= lo_v[k] * yv;
            }
        } else if (FACTOR == 2) {
            // positions of the three dependency rows in the ring of records
            const int p0 = 4 * ((r.m.x < 0 ? 0 : r.m.x) & (W - 1)), p1 = 4 * ((r.m.y < 0 ? 0 : r.m.y) & (W - 1)),
                      p2 = 4 * ((r.m.z < 0 ? 0 : r.m.z) & (W - 1));
            const int s01 = (r.xd & 3) - 1, s02 = ((r.xd >> 2) & 3) - 1, s12 = ((r.xd >> 4) & 3) - 1;
            const double d0 = ring[p0 + 3], d1 = ring[p1 + 3], d2 = ring[p2 + 3];
            // the cross operands (slot 0 read when there is none: never used)
            const double c01 = ring[p1 + (s01 < 0 ? 0 : s01)], c02 = ring[p2 + (s02 < 0 ? 0 : s02)],
                         c12 = ring[p2 + (s12 < 0 ? 0 : s12)];
            double l0 = 0.0, l1 = 0.0, l2 = 0.0;
            if (r.m.x >= 0) {
                l0 = r.v01.x / d0;
                if (thr && fabs(l0) * d0 < r.t01.x) l0 = 0.0;
            }
            if (r.m.y >= 0) {
                double a1 = r.v01.y;
                if (s01 >= 0) a1 -= l0 * c01;
                l1 = a1 / d1;
                if (thr && fabs(l1) * d1 < r.t01.y) l1 = 0.0;
            }
            if (r.m.z >= 0) {
                double a2 = r.v2d.x;
                if (s02 >= 0) a2 -= l0 * c02;
                if (s12 >= 0) a2 -= l1 * c12;
                l2 = a2 / d2;
                if (thr && fabs(l2) * d2 < r.t2) l2 = 0.0;
            }
            acc = r.v2d.y;                                   // A_ii
            if (r.m.x >= 0) acc -= l0 * l0;
            if (r.m.y >= 0) acc -= l1 * l1;
            if (r.m.z >= 0) acc -= l2 * l2;
            acc = sqrt(acc);
            if (valid) {
                fac[2 * (int64_t)r.j] = make_double2(l0, l1);
                fac[2 * (int64_t)r.j + 1] = make_double2(l2, acc);
                double *slot = ring + 4 * (r.j & (W - 1));
                slot[0] = l0; slot[1] = l1; slot[2] = l2; slot[3] = acc;
                out[r.m.w] = acc;
            }
            return;
        } else {
            const double y0 = ring[(r.m.x < 0 ? 0 : r.m.x) & (W - 1)];
            const double y1 = ring[(r.m.y < 0 ? 0 : r.m.y) & (W - 1)];
            const double y2 = ring[(r.m.z < 0 ? 0 : r.m.z) & (W - 1)];
            if (FACTOR) {
                double l0 = 0.0, l1 = 0.0, l2 = 0.0;
                acc = r.v2d.y;                               // A_ii
                if (r.m.x >= 0) { l0 = r.v01.x / y0; acc -= l0 * l0; }
                if (r.m.y >= 0) { l1 = r.v01.y / y1; acc -= l1 * l1; }
                if (r.m.z >= 0) { l2 = r.v2d.x / y2; acc -= l2 * l2; }
                acc = sqrt(acc);
                if (valid) {
                    fac[2 * (int64_t)r.j] = make_double2(l0, l1);
                    fac[2 * (int64_t)r.j + 1] = make_double2(l2, acc);
                }
            } else {
                if (r.m.x >= 0) acc -= r.v01.x * y0;
                if (r.m.y >= 0) acc -= r.v01.y * y1;
                if (r.m.z >= 0) acc -= r.v2d.x * y2;
            }
        }
        const double y = FACTOR ? acc : acc / r.v2d.y;
        if (valid) {
            ring[r.j & (W - 1)] = y;
            out[r.m.w] = y;
        }
    };
    // LDS-only hand-off (see k_sptrsv_ring): no vmcnt drain, the prefetched loads stay in flight across it
    auto level_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto solve_chunk = [&](Row (&S)[C][ROWS]) {
#pragma unroll
        for (int d = 0; d < C; ++d) {
#pragma unroll
            for (int h = 0; h < ROWS; ++h) solve_row(S[d][h]);
            level_barrier();
        }
    };
    Row S0[C][ROWS], S1[C][ROWS];
    load_chunk(S0, 0);
    for (int c = 0; c < nchunks; c += 2) {
        retire(S0);
        load_chunk(S1, c + 1);                            // flies while chunk c is solved
        solve_chunk(S0);
        if (c + 1 >= nchunks) break;
        retire(S1);
        load_chunk(S0, c + 2);
        solve_chunk(S1);
    }
}

// ------------------------------------------------------------------------------------------------
// Several levels in ONE multi-workgroup launch, without barriers ("sync-free" triangular solve): in level order a
// workgroup takes 256 consecutive rows -- whatever levels they belong to -- and every row polls the solution entries it
// depends on until they have been written.  An entry carries its own flag: `out` is filled with a reserved NaN bit
// pattern before the launch (k_fill_pending) and an 8-byte agent-scope store of the value replaces it, so a single
// relaxed 8-byte agent-scope load is both the test and the read (no fence, no second word).  Rows of earlier levels
// sit at lower level-order positions, and the 256-row blocks are handed out through a ticket counter, so everything a
// workgroup waits for belongs to a workgroup that has already started: no deadlock whatever the dispatch order; lanes
// of one wave may depend on each other, hence the store inside the poll loop.  Against one launch per level this
// removes (levels - 1) kernel boundaries per segment and lets consecutive levels overlap: a row starts as soon as ITS
// entries are there.  Arithmetic as everywhere else: ascending columns, one product and one subtraction at a time,
// then the division -- bit-identical to sequential substitution.
// ------------------------------------------------------------------------------------------------
// A wait for another workgroup's entry is bounded by WALL TIME (the 100 MHz constant clock), not by a poll count: a chip
// shared with other processes, a profiler that serialises dispatches or a pre-empted predecessor stretch a perfectly valid
// wait; only a malformed schedule waits for seconds.  The clock is looked at every 4096 polls.
constexpr unsigned long long kSpinTicks = 4ull * 100000000ull;     // 4 s
__device__ __forceinline__ bool spin_expired(unsigned &spins, unsigned long long &t_start) {
    if ((++spins & 4095u) != 0) return false;
    const unsigned long long now = wall_clock64();
    if (t_start == 0) {
        t_start = now;
        return false;
    }
    return now - t_start > kSpinTicks;
}
// level-major sync-free solve: blocks of rows per ticket (see k_sptrsv_syncfree_rec).  Measured with 4: a 2-level factor
// gains (104.7 -> 63.6 us per apply at 1M rows) but a 19-level one loses badly (134 -> 301 us: the blocks of a ticket are
// walked one after another, so a level takes four turns to complete); few-level factors of that size now take one launch
// per level (build_levels), so one block per ticket it is.
constexpr int kSfSub = 1;
constexpr unsigned long long kPendingBits = 0x7ff8dead0badbeefULL;   // a quiet NaN that no arithmetic here produces
__device__ __forceinline__ bool is_pending(double y) { return (unsigned long long)__double_as_longlong(y) == kPendingBits; }

__global__ __launch_bounds__(kBlock) void k_fill_pending(const int32_t *__restrict__ rows, int j0, int count,
                                                         double *__restrict__ out, const int *done) {
    if (done && *done) return;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx < count) out[rows ? rows[j0 + idx] : j0 + idx] = __longlong_as_double((long long)kPendingBits);
}

// The kernel works on fixed-width records (Levels::sf_meta / sf_val): everything a row needs -- three column indices,
// three (or six: SfRec) values, the diagonal, its own index -- comes from a few 16-byte loads addressed by the level-order
// position alone, issued the moment the block is drawn; no row extents to wait for, no LDS stage, no workgroup barrier
// besides the ticket hand-out.  Rows with more entries than a record holds (col[0] == -2) walk lo_rowptr.
// The grid is PERSISTENT and sized to the wavefront (about twice the widest level): a workgroup draws the next 256-row
// block when it has finished one, so only rows near the front are resident and polling.  (With one workgroup per block
// and the whole factor resident, thousands of waves polled entries tens of levels away; their requests saturated the
// L2 and a hop cost 3.6 us instead of the ~0.5 us an idle chip needs -- tools/hop_lab.)  The ticket word and an exit
// counter sit side by side; the last workgroup to leave zeroes both for the next launch.
// Measured per apply (two solves), IC(0) of natural-order grids: 64^3 (190 levels) 1.05 ms vs 1.41 ms with one launch per
// level; 100^3 (298 levels) 1.47 vs 2.46 ms.  A run of VERY wide levels (the scrambled 1M-DoF factor: 19 levels of ~52K
// rows) is the other way round -- 0.43 ms vs 0.31 ms: there the level bodies, not the boundaries, are the cost, the
// polling loads bypass the L1, and rows several levels ahead of the front poll for nothing -- in the HANDLE's numbering such
// runs keep one launch per level (build_levels: mean level width > 16384); a factor made of them is solved level-major instead
// (Levels::level_major: contiguous levels, ONE launch of this kernel per solve with a window of 0.85 x the widest level and
// 512 rows per ticket; rhs_map / refill: see SptrsvIo::fused_entry).
// Tickets.  Every atomic on the ticket word is served one after another (~12 ns each: 88 per us on this chip), and a solve
// of 1M rows in 512-row tickets drew 1954 of them plus one out-of-range draw and one exit count per workgroup -- 3600
// serialised atomics = 43 of the kernel's 49 us, whatever the number of levels (found with a 2-level factor: IC(0) in
// red-black order ran no faster than the 19-level one).  Now a ticket is SUB consecutive blocks of BS rows, walked in order
// by the workgroup that drew it (a block's rows depend only on earlier positions: earlier blocks of the same ticket or
// earlier tickets, so the no-deadlock argument is unchanged), and the exit count is read off the ticket itself: every
// workgroup draws exactly one ticket past the end, and the one that draws the LAST of those zeroes the word for the next
// launch.  Atomics per solve: tickets + workgroups.
// Two more variants measured on the scrambled 1M-DoF IC(0) factor (19 levels; 132.5 us per apply as is) and dropped: two / four
// rows per LANE polled in one loop (half / a quarter of the tickets): 176 / 271 us; a second ticket per workgroup whose
// records are loaded while the current block polls, the third drawn meanwhile (ticket and record latency off the chain):
// 138 us.  Neither the tickets nor the per-block load latency bound that solve: its two kernels move 2 x 112 MB of records
// (about 2 x 28 us at the rate the colour sweeps reach) and pay ~1.1 us for each of the 19 hand-offs.
template <bool UPPER, int BS, int W, int SUB = 1>   // BS rows (= threads) per block, SUB blocks per ticket, records of width W
__global__ __launch_bounds__(BS) void k_sptrsv_syncfree_rec(int j0, int count, const int32_t *__restrict__ lo_rp,
                                                            const int32_t *__restrict__ lo_ci,
                                                            const double *__restrict__ lo_v,
                                                            const int32_t *__restrict__ meta,
                                                            const double *__restrict__ val,
                                                            const double *__restrict__ rhs, double *out,
                                                            unsigned int *ticket /* next ticket; [1] unused */,
                                                            int nblocks, const int *done, int *err,
                                                            const int32_t *__restrict__ rhs_map /* null: rhs[own] */,
                                                            double *__restrict__ refill /* null, or preset to pending */) {
    __shared__ unsigned int s_lb;
    const int t = threadIdx.x;
    if (done && *done) return;                      // nothing drawn: the counter stays zero
    const unsigned int ntickets = ((unsigned int)nblocks + SUB - 1) / SUB;
    for (;;) {
        __syncthreads();
        if (t == 0) s_lb = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned int tk = s_lb;
        if (tk >= ntickets) {
            // one out-of-range draw per workgroup: the last of them leaves the word at zero for the next launch
            if (t == 0 && tk == ntickets + gridDim.x - 1) atomicExch(ticket, 0u);
            break;
        }
      for (int sub = 0; sub < SUB; ++sub) {
        const unsigned int lb = tk * SUB + sub;
        if (lb >= (unsigned int)nblocks) break;
        const int j = j0 + (int)lb * BS + t;
        const bool valid = j < j0 + count;
        const int jc = valid ? j : j0;              // lanes without a row load a valid record and ignore it
        const int rhs_at = rhs_map ? rhs_map[jc] : -1;
        const SfRec<W> r = load_sf_record<W>(meta, val, jc);
        const double bi = rhs[rhs_map ? rhs_at : r.own];
        if (refill && valid) refill[j] = __longlong_as_double((long long)kPendingBits);
        bool stored = !valid;
        unsigned spins = 0;
        unsigned long long t_wait = 0;
        // ONE loop for every lane of the wave, left by the whole wave at once (lanes may wait for each other, and a
        // store on an exit path would only run after every lane has left).  Rows that fit the record (<= W entries): all
        // entries are asked for at once, the pending ones again, and consumed in column order.  Longer rows
        // (col[0] == -2) walk the level-ordered copy one entry at a time.
        const bool longrow = r.col[0] == -2;
        const double pend = __longlong_as_double((long long)kPendingBits);
        double y[W];
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = (!longrow && r.col[q] >= 0) ? pend : 0.0;
        int k = 0, ke = 0;
        double acc = bi;
        if (longrow) {
            const int s = lo_rp[jc], e = lo_rp[jc + 1];
            k = UPPER ? s + 1 : s;
            ke = UPPER ? e : e - 1;
        }
        for (;;) {
            if (!stored) {
                bool ready = false, waited = false;
                if (longrow) {
                    // up to four entries per turn: their columns, then their values of `out`, all at once; consumed in order as
                    // far as they have arrived (nothing is kept across turns: the columns come from the L2 again).  One entry
                    // per turn -- two dependent round trips each -- made a 20-entry row cost 40 trips.
                    if (k < ke) {
                        constexpr int CH = 4;
                        int cc[CH];
                        double vv[CH], yy[CH];
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const int kk = k + c < ke ? k + c : ke - 1;
                            cc[c] = lo_ci[kk];
                            vv[c] = lo_v[kk];
                        }
#pragma unroll
                        for (int c = 0; c < CH; ++c) yy[c] = __hip_atomic_load(out + cc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bool open = true;                  // still consuming the prefix that has arrived
                        const int k_before = k;
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            open = open && k_before + c < ke && !is_pending(yy[c]);
                            if (open) {
                                acc -= vv[c] * yy[c];
                                ++k;
                            }
                        }
                        if (k != k_before) { spins = 0; t_wait = 0; }
                        else waited = true;
                    }
                    ready = k >= ke;
                } else {
                    ready = true;
#pragma unroll
                    for (int q = 0; q < W; ++q) {
                        if (is_pending(y[q])) y[q] = __hip_atomic_load(out + r.col[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int q = 0; q < W; ++q) ready = ready && !is_pending(y[q]);
                    waited = !ready;
                    if (ready) {
#pragma unroll
                        for (int q = 0; q < W; ++q)
                            if (r.col[q] >= 0) acc -= r.v[q] * y[q];
                    }
                }
                if (waited && spin_expired(spins, t_wait)) {   // bounded: never hang the device on a malformed schedule
                    atomicExch(err, 1);
                    acc = __builtin_nan("");
                    ready = true;
                }
                if (ready) {
                    __hip_atomic_store(out + r.own, acc / r.diag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stored = true;
                } else if (waited) {
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (__ballot(!stored) == 0) break;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// The level-major sync-free solve in CSR-STREAM form.  The record kernel above moves 96 bytes of fixed-width record per row of a
// factor with rows of up to six entries (128 with the vectors) although such a row holds 3.5 entries on average; this form streams
// the level-ordered copy itself -- 12 bytes per entry, 4 per row extent -- the way the colour sweeps do: a ticket is a block of
// <= 256 consecutive rows of ONE level (Levels::sfs_blk: blocks never straddle a level, so no row of a block depends on another
// one of it), lane k of the block takes entry k of its contiguous val / position segment and polls the solution entry it needs
// (8 in flight per lane), parks value x solution in LDS, and after a barrier thread j subtracts row j's products in column order and
// divides by the diagonal (which travels through its product slot): the arithmetic and order of every other triangular-solve
// kernel, bit-identical to sequential substitution.  rhs_map / refill as in the record kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sfs_block_max(const int32_t *__restrict__ blk, int nblk, const int32_t *__restrict__ lo_rp,
                                                          int *out) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b < nblk) atomicMax(out, lo_rp[blk[2 * b + 1]] - lo_rp[blk[2 * b]]);
}
void launch_sfs_block_max(const int32_t *blk, int nblk, const int32_t *lo_rowptr, int *out_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_sfs_block_max, dim3((nblk + kBlock - 1) / kBlock), dim3(kBlock), 0, s, blk, nblk, lo_rowptr, out_dev);
}

template <bool UPPER, int BS>      // BS rows (= threads) per block
__global__ __launch_bounds__(BS) void k_sptrsv_syncfree_stream(const int32_t *__restrict__ blk, int nblk,
                                                                   const int32_t *__restrict__ lo_rp,
                                                                   const int32_t *__restrict__ lo_cp,
                                                                   const double *__restrict__ lo_v,
                                                                   const double *__restrict__ rhs, double *out,
                                                                   unsigned int *ticket, const int *done, int *err,
                                                                   const int32_t *__restrict__ rhs_map,
                                                                   double *__restrict__ refill) {
    constexpr int U = kStreamCap / kBlock;      // entries per thread: BS * U product slots
    __shared__ double prod[BS * U];
    __shared__ unsigned int s_lb;
    const int t = threadIdx.x;
    if (done && *done) return;
    const double pend = __longlong_as_double((long long)kPendingBits);
    for (;;) {
        __syncthreads();                                 // (prod of the previous block has been read)
        if (t == 0) s_lb = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned int b = s_lb;
        if (b >= (unsigned int)nblk) {
            if (t == 0 && b == (unsigned int)nblk + gridDim.x - 1) atomicExch(ticket, 0u);   // the last leaver re-arms the word
            break;
        }
        const int jb = blk[2 * b], jend = blk[2 * b + 1];
        const int j = jb + t;
        const bool live = j < jend;
        const int base = lo_rp[jb];
        const int cnt = lo_rp[jend] - base;
        int rs = 0, re = 0;
        double bi = 0.0;
        if (live) {
            rs = lo_rp[j] - base;
            re = lo_rp[j + 1] - base;
            bi = rhs[rhs_map ? rhs_map[j] : j];
            if (refill) refill[j] = pend;
        }
        int c[U];
        double a[U], y[U];
        const int last = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = t + u * BS;
            const int kk = k < cnt ? k : last;
            c[u] = lo_cp[base + kk];
            a[u] = lo_v[base + kk];
        }
        // an entry whose position lies before the block is a dependency (earlier level); the others are the rows' own diagonals
        bool waiting = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool dep = (t + u * BS) < cnt && c[u] < jb;
            y[u] = dep ? __hip_atomic_load(out + c[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1.0;
            waiting = waiting || is_pending(y[u]);
        }
        unsigned spins = 0;
        unsigned long long t_wait = 0;
        while (waiting) {                                // every entry this lane waits for belongs to an earlier ticket
            __builtin_amdgcn_s_sleep(1);
            waiting = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (is_pending(y[u])) {
                    y[u] = __hip_atomic_load(out + c[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    waiting = waiting || is_pending(y[u]);
                }
            }
            if (waiting && spin_expired(spins, t_wait)) {   // bounded: never hang the device on a malformed schedule
                atomicExch(err, 1);
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (is_pending(y[u])) y[u] = __builtin_nan("");
                waiting = false;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = t + u * BS;
            if (k < cnt) prod[k] = a[u] * y[u];          // (a diagonal: times 1.0 -- the slot holds the diagonal itself)
        }
        __syncthreads();
        if (live) {
            double acc = bi;
            const int k0 = UPPER ? rs + 1 : rs, k1 = UPPER ? re : re - 1;
            for (int k = k0; k < k1; ++k) acc -= prod[k];
            const double d = prod[UPPER ? rs : re - 1];
            __hip_atomic_store(out + j, acc / d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Strip-pipelined triangular solve (banded factors with many narrow levels: natural / RCM-ordered grids).
// The level walk through an LDS ring (k_sptrsv_ring_pipe) costs ~0.35 us per level but runs on ONE CU; the sync-free
// kernel uses the whole chip but pays a ~2.7 us hand-off per level.  Here the rows are cut into STRIPS of consecutive
// row indices (for a grid: slabs of planes), one workgroup per strip, all strips in ONE launch:
//   * inside a strip the rows are walked in strip-LOCAL level order (levels computed with the dependencies on other
//     strips ignored); entries of the own strip are handed from level to level through the LDS ring;
//   * an entry of an EARLIER strip is polled in `out` (reserved-NaN "pending" pattern, 8-byte agent-scope loads; the
//     owners store with agent scope) -- the strips form a software pipeline in which strip s trails strip s - 1 by the
//     hand-off latency ONCE, not once per level.
// Strips are handed out through a ticket, so a strip's predecessors have always started.  Records, prefetch chunks and
// the LDS-only level barrier are those of k_sptrsv_ring_pipe; arithmetic and its order are those of every other SpTRSV
// kernel here -- bit-identical to sequential substitution.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_strip_prepare(const int32_t *__restrict__ rows, const double *__restrict__ rhs,
                                                          double *__restrict__ b_lo, double *out, int count, const int *done) {
    if (done && *done) return;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx < count) {
        const int i = rows[idx];
        b_lo[idx] = rhs[i];
        out[i] = __longlong_as_double((long long)kPendingBits);
    }
}

// FACTOR: the same walk computes IC(0) instead of a solve, for a pattern WITHOUT cross terms (no two lower entries of a row
// are themselves joined by an entry: every 5- / 7-point grid in any numbering -- checked by k_ic0_cross_terms, not assumed)
// and rows of at most three off-diagonal entries.  Then L_ik = A_ik / L_kk and L_ii = sqrt(A_ii - sum_k L_ik^2): the only
// thing a row needs of an earlier row is its DIAGONAL, one number per dependency -- the data flow of the lower solve, with
// the records holding the entries of tril(A) and `out` / the ring the diagonals of L.  Operation order as k_ic0_level:
// ascending columns, the quotient, then one product and one subtraction at a time, then the root.  The factor's values go
// to fac (the records of the SAME plan, which thereby becomes the plan of L) and are scattered to CSR by k_strip_factor_scatter.
// FACTOR = 2: the general form (cross terms, ICT's drop rule) for rows of at most three off-diagonal entries, as in
// k_sptrsv_ring_pipe: the ring holds whole records {l0, l1, l2, diagonal}, xdesc names the cross-term slots, thr the drop thresholds;
// what another strip needs of a row -- its diagonal AND its off-diagonal values -- is published in out (diagonals) and out3
// (3 values per row), every value self-validating ("pending" until stored), asked for per chunk and polled per value.
template <bool UPPER, int C, int ROWS, int FACTOR = 0>
__global__ __launch_bounds__(512) void k_sptrsv_strips(const int32_t *__restrict__ level_ptr, int nlev,
                                                       const int32_t *__restrict__ lo_rp, const int32_t *__restrict__ lo_ci,
                                                       const int32_t *__restrict__ lo_cpos, const double *__restrict__ lo_v,
                                                       const int4 *__restrict__ pk_meta, const double2 *__restrict__ pk_val,
                                                       const double *__restrict__ b_lo, double *out, int W,
                                                       int ring_reach, unsigned int *ticket, const int *done, int *err,
                                                       long long *trace /* development: per strip {start, end, polls, levels} */,
                                                       double2 *__restrict__ fac = nullptr /* FACTOR: records of L, by position */,
                                                       const int32_t *__restrict__ xdesc = nullptr,
                                                       const double2 *__restrict__ thr = nullptr, double *out3 = nullptr) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    extern __shared__ __attribute__((aligned(16))) double ring[];
    const long long t_start = trace ? (long long)wall_clock64() : 0;
    int n_polls = 0;
    int *lp = reinterpret_cast<int *>(ring + (FACTOR == 2 ? 4 * W : W));   // level offsets of this strip, padded with empty levels
    __shared__ unsigned int s_strip;
    __shared__ int s_nl;
    const int t = threadIdx.x, T = blockDim.x;
    if (t == 0) s_strip = atomicAdd(ticket, 1u);
    __syncthreads();
    const int lvl_lo = (int)s_strip * nlev;
    const int seg_start = level_ptr[lvl_lo], seg_end = level_ptr[lvl_lo + nlev];
    for (int i = t; i <= nlev + 3 * C; i += T) lp[i] = i <= nlev ? level_ptr[lvl_lo + i] : seg_end;
    __syncthreads();
    if (t == 0) {                                       // trailing empty levels of a short strip are not walked
        int nl = nlev;
        while (nl > 0 && lp[nl - 1] == seg_end) --nl;
        s_nl = nl;
    }
    __syncthreads();
    const int nl = s_nl;
    const int nchunks = (nl + C - 1) / C;
    const int jmax = seg_end > seg_start ? seg_end - 1 : seg_start;
    struct Row {
        int j;           // position, -1 for a lane without a row in this level
        int4 m;          // d0..d2, own row index
        double2 v01, v2d;
        double b;
        int xd;          // FACTOR 2: cross-term slots
        double2 t01;     // FACTOR 2 with a drop rule: thresholds of entries 0, 1
        double t2;       //                            ... and 2
    };
    auto load_row = [&](Row &r, int j, int hi) {
        const int jc = j < hi ? j : jmax;                 // lanes without a row load a valid record and ignore it
        r.j = j < hi ? j : -1;
        r.m = pk_meta[jc];
        r.v01 = pk_val[2 * (int64_t)jc];
        r.v2d = pk_val[2 * (int64_t)jc + 1];
        r.b = FACTOR ? 0.0 : b_lo[jc];
        r.xd = 0;
        r.t01 = make_double2(0.0, 0.0);
        r.t2 = 0.0;
        if (FACTOR == 2) {
            r.xd = xdesc[jc];
            if (thr) {
                r.t01 = thr[2 * (int64_t)jc];
                r.t2 = thr[2 * (int64_t)jc + 1].x;
            }
        }
    };
    auto load_chunk = [&](Row (&S)[C][ROWS], int chunk) {
#pragma unroll
        for (int d = 0; d < C; ++d) {
            const int rel = chunk * C + d;
            const int lo = lp[rel], hi = lp[rel + 1];
#pragma unroll
            for (int h = 0; h < ROWS; ++h) load_row(S[d][h], lo + t + h * T, hi);
        }
    };
    auto retire = [&](Row (&S)[C][ROWS]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < C; ++d)
#pragma unroll
            for (int h = 0; h < ROWS; ++h) {
                Row &r = S[d][h];
                asm volatile("" : "+v"(r.m.x), "+v"(r.m.y), "+v"(r.m.z), "+v"(r.m.w), "+v"(r.v01.x), "+v"(r.v01.y),
                             "+v"(r.v2d.x), "+v"(r.v2d.y), "+v"(r.b));
                if (FACTOR == 2) asm volatile("" : "+v"(r.xd), "+v"(r.t01.x), "+v"(r.t01.y), "+v"(r.t2));
            }
    };
    // an entry of an earlier strip: poll until its owner has stored it (bounded)
    auto poll = [&](int col) {
        double y;
        ++n_polls;
        unsigned spins = 0;
        unsigned long long t_wait = 0;
        for (;;) {
            y = __hip_atomic_load(out + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!is_pending(y)) break;
            if (spin_expired(spins, t_wait)) {
                atomicExch(err, 1);
                y = __builtin_nan("");
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        return y;
    };
    // Entries of EARLIER strips are asked for once per CHUNK, all at once, before the chunk's levels are walked (the records
    // that name them have just been retired); a value that is still pending then is polled when its row is solved.  The
    // hand-off latency is paid once per C levels instead of once per level, and the steady-state lag of a strip behind
    // its predecessor grows accordingly.
    struct Ext { double y0, y1, y2, c01, c02, c12; };
    const double pend = __longlong_as_double((long long)kPendingBits);
    auto ask = [&](int d, bool valid) {
        return (valid && d <= -2) ? __hip_atomic_load(out + (-2 - d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    };
    // FACTOR 2: off-diagonal value `slot` of the row of an EARLIER strip (column -2 - d), asked for / polled like a diagonal
    auto ask3 = [&](int d, int slot, bool valid) {
        return (valid && d <= -2 && slot >= 0)
                   ? __hip_atomic_load(out3 + 3 * (int64_t)(-2 - d) + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    };
    auto poll3 = [&](int col, int slot) {
        double y;
        ++n_polls;
        unsigned spins = 0;
        unsigned long long t_wait = 0;
        for (;;) {
            y = __hip_atomic_load(out3 + 3 * (int64_t)col + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!is_pending(y)) break;
            if (spin_expired(spins, t_wait)) {
                atomicExch(err, 1);
                y = __builtin_nan("");
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        return y;
    };
    auto solve_row = [&](const Row &r, const Ext &e) {
        const bool valid = r.j >= 0;
        double acc = r.b;
        if (valid && r.m.x == (int)0x80000000) {          // long row: entries from the level-ordered copy
            const int s = lo_rp[r.j], e2 = lo_rp[r.j + 1];
            const int ks = UPPER ? s + 1 : s, ke = UPPER ? e2 : e2 - 1;
            for (int k = ks; k < ke; ++k) {
                const int cp = lo_cpos[k];
                const bool near = cp >= seg_start && cp < seg_end && r.j - cp <= ring_reach;   // as k_strip_records decides
                const double yv = near ? ring[cp & (W - 1)] : poll(lo_ci[k]);
                acc -= lo_v[k] * yv;
            }
        } else if (FACTOR == 2) {
            const int p0 = 4 * ((r.m.x < 0 ? 0 : r.m.x) & (W - 1)), p1 = 4 * ((r.m.y < 0 ? 0 : r.m.y) & (W - 1)),
                      p2 = 4 * ((r.m.z < 0 ? 0 : r.m.z) & (W - 1));
            const int s01 = (r.xd & 3) - 1, s02 = ((r.xd >> 2) & 3) - 1, s12 = ((r.xd >> 4) & 3) - 1;
            // diagonals and cross operands of the dependency rows: from the ring (own strip) or published by an earlier strip
            double d0 = ring[p0 + 3], d1 = ring[p1 + 3], d2 = ring[p2 + 3];
            double c01 = ring[p1 + (s01 < 0 ? 0 : s01)], c02 = ring[p2 + (s02 < 0 ? 0 : s02)], c12 = ring[p2 + (s12 < 0 ? 0 : s12)];
            if (valid) {
                if (r.m.x <= -2) d0 = is_pending(e.y0) ? poll(-2 - r.m.x) : e.y0;
                if (r.m.y <= -2) {
                    d1 = is_pending(e.y1) ? poll(-2 - r.m.y) : e.y1;
                    if (s01 >= 0) c01 = is_pending(e.c01) ? poll3(-2 - r.m.y, s01) : e.c01;
                }
                if (r.m.z <= -2) {
                    d2 = is_pending(e.y2) ? poll(-2 - r.m.z) : e.y2;
                    if (s02 >= 0) c02 = is_pending(e.c02) ? poll3(-2 - r.m.z, s02) : e.c02;
                    if (s12 >= 0) c12 = is_pending(e.c12) ? poll3(-2 - r.m.z, s12) : e.c12;
                }
            }
            double l0 = 0.0, l1 = 0.0, l2 = 0.0;
            if (r.m.x != -1) {
                l0 = r.v01.x / d0;
                if (thr && fabs(l0) * d0 < r.t01.x) l0 = 0.0;
            }
            if (r.m.y != -1) {
                double a1 = r.v01.y;
                if (s01 >= 0) a1 -= l0 * c01;
                l1 = a1 / d1;
                if (thr && fabs(l1) * d1 < r.t01.y) l1 = 0.0;
            }
            if (r.m.z != -1) {
                double a2 = r.v2d.x;
                if (s02 >= 0) a2 -= l0 * c02;
                if (s12 >= 0) a2 -= l1 * c12;
                l2 = a2 / d2;
                if (thr && fabs(l2) * d2 < r.t2) l2 = 0.0;
            }
            acc = r.v2d.y;                                   // A_ii
            if (r.m.x != -1) acc -= l0 * l0;
            if (r.m.y != -1) acc -= l1 * l1;
            if (r.m.z != -1) acc -= l2 * l2;
            acc = sqrt(acc);
            if (valid) {
                fac[2 * (int64_t)r.j] = make_double2(l0, l1);
                fac[2 * (int64_t)r.j + 1] = make_double2(l2, acc);
                double *slot = ring + 4 * (r.j & (W - 1));
                slot[0] = l0; slot[1] = l1; slot[2] = l2; slot[3] = acc;
                const int row = r.m.w & 0x3fffffff;
                if (r.m.w & (1 << 30)) {
                    __hip_atomic_store(out3 + 3 * (int64_t)row, l0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(out3 + 3 * (int64_t)row + 1, l1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(out3 + 3 * (int64_t)row + 2, l2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(out + row, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    out[row] = acc;
                }
            }
            return;
        } else {
            double y0 = ring[(r.m.x < 0 ? 0 : r.m.x) & (W - 1)];
            double y1 = ring[(r.m.y < 0 ? 0 : r.m.y) & (W - 1)];
            double y2 = ring[(r.m.z < 0 ? 0 : r.m.z) & (W - 1)];
            if (valid) {
                if (r.m.x <= -2) y0 = is_pending(e.y0) ? poll(-2 - r.m.x) : e.y0;
                if (r.m.y <= -2) y1 = is_pending(e.y1) ? poll(-2 - r.m.y) : e.y1;
                if (r.m.z <= -2) y2 = is_pending(e.y2) ? poll(-2 - r.m.z) : e.y2;
            }
            if (FACTOR) {
                double l0 = 0.0, l1 = 0.0, l2 = 0.0;
                acc = r.v2d.y;                               // A_ii
                if (r.m.x != -1) { l0 = r.v01.x / y0; acc -= l0 * l0; }
                if (r.m.y != -1) { l1 = r.v01.y / y1; acc -= l1 * l1; }
                if (r.m.z != -1) { l2 = r.v2d.x / y2; acc -= l2 * l2; }
                acc = sqrt(acc);                             // (a non-positive pivot: NaN or 0, found by the scatter pass)
                if (valid) {
                    fac[2 * (int64_t)r.j] = make_double2(l0, l1);
                    fac[2 * (int64_t)r.j + 1] = make_double2(l2, acc);
                }
            } else {
                if (r.m.x != -1) acc -= r.v01.x * y0;
                if (r.m.y != -1) acc -= r.v01.y * y1;
                if (r.m.z != -1) acc -= r.v2d.x * y2;
            }
        }
        const double y = FACTOR ? acc : acc / r.v2d.y;
        if (valid) {
            ring[r.j & (W - 1)] = y;
            // only rows that somebody reads from `out` DURING the launch are published (write-through); the rest is a plain
            // store, visible at the kernel's end like any other
            const int row = r.m.w & 0x3fffffff;
            if (r.m.w & (1 << 30)) __hip_atomic_store(out + row, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else out[row] = y;
        }
    };
    auto level_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto solve_chunk = [&](Row (&S)[C][ROWS]) {
        Ext E[C][ROWS];
#pragma unroll
        for (int d = 0; d < C; ++d)
#pragma unroll
            for (int h = 0; h < ROWS; ++h) {
                const Row &r = S[d][h];
                const bool v = r.j >= 0 && r.m.x != (int)0x80000000;
                E[d][h].y0 = ask(r.m.x, v);
                E[d][h].y1 = ask(r.m.y, v);
                E[d][h].y2 = ask(r.m.z, v);
                E[d][h].c01 = E[d][h].c02 = E[d][h].c12 = 0.0;
                if (FACTOR == 2) {
                    E[d][h].c01 = ask3(r.m.y, (r.xd & 3) - 1, v);
                    E[d][h].c02 = ask3(r.m.z, ((r.xd >> 2) & 3) - 1, v);
                    E[d][h].c12 = ask3(r.m.z, ((r.xd >> 4) & 3) - 1, v);
                }
            }
        (void)pend;
#pragma unroll
        for (int d = 0; d < C; ++d) {
#pragma unroll
            for (int h = 0; h < ROWS; ++h) solve_row(S[d][h], E[d][h]);
            level_barrier();
        }
    };
    Row S0[C][ROWS], S1[C][ROWS];
    if (nchunks > 0) {
        load_chunk(S0, 0);
        for (int c = 0; c < nchunks; c += 2) {
            retire(S0);
            load_chunk(S1, c + 1);                            // flies while chunk c is solved
            solve_chunk(S0);
            if (c + 1 >= nchunks) break;
            retire(S1);
            load_chunk(S0, c + 2);
            solve_chunk(S1);
        }
    }
    if (trace) {
        atomicAdd(reinterpret_cast<unsigned long long *>(trace + 4 * s_strip + 2), (unsigned long long)n_polls);
        if (t == 0) {
            trace[4 * s_strip] = t_start;
            trace[4 * s_strip + 1] = (long long)wall_clock64();
            trace[4 * s_strip + 3] = nl;
        }
    }
    if (t == 0 && atomicAdd(ticket + 1, 1u) == gridDim.x - 1) {      // last one out: zero the counters for the next launch
        atomicExch(ticket, 0u);
        atomicExch(ticket + 1, 0u);
    }
}

constexpr int kRingChunk = 6;   // levels per prefetch chunk of k_sptrsv_ring_pipe (3 with two rows per thread)
static bool level_rec_disabled() {   // DPCG_LEVEL_REC=0: the LDS-staged level kernel instead of the record one (A/B)
    static const bool off = [] { const char *e = getenv("DPCG_LEVEL_REC"); return e && e[0] == '0'; }();
    return off;
}
static bool ring_pipe_disabled() {
    static const bool off = [] { const char *e = getenv("DPCG_RING_PIPE"); return e && e[0] == '0'; }();
    return off;
}

constexpr int kStripChunk = 4;   // levels per prefetch chunk of k_sptrsv_strips (2 with two rows per thread)

// The strip kernel may need more than the default 64 KiB of dynamic LDS (ring of up to 8192 doubles + level offsets):
// raise the limit once, at setup time (never inside a stream capture).
void init_strip_kernels() {
    static bool done_once = false;
    if (done_once) return;
    done_once = true;
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<true, kStripChunk, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<false, kStripChunk, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<true, kStripChunk / 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<false, kStripChunk / 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<false, kStripChunk, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<false, kStripChunk / 2, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<false, kStripChunk / 2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    (void)hipFuncSetAttribute((const void *)k_sptrsv_strips<false, kStripChunk / 4, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
}

// IC(0) through a strip plan built on the pattern of tril(A) (k_sptrsv_strips<..., FACTOR>): diag[] (n doubles, by the index
// the plan addresses its vectors with) receives the diagonal of L, fac the records {l0, l1, l2, diagonal} by position.
// xdesc != null: the general form (cross terms; thr != null: with the drop rule); offd: 3 n doubles, the published off-diagonal values.
// Returns false when the ring of whole records does not fit the LDS.
bool launch_strip_factor(const Levels &lv, double *diag, double *fac, int64_t n, hipStream_t s, const int32_t *xdesc, const double *thr,
                         double *offd) {
    const Levels::Strips &sp = lv.strips;
    constexpr int CH = kStripChunk;
    const size_t lds = (size_t)sp.W * (xdesc ? 4 : 1) * sizeof(double) + (size_t)(sp.nlev + 3 * CH + 8) * sizeof(int);
    if (lds > (xdesc ? 144 : 128) * 1024) return false;
    hipLaunchKernelGGL(k_fill_pending, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, nullptr, 0, (int)n, diag,
                       nullptr);
    if (xdesc)
        hipLaunchKernelGGL(k_fill_pending, dim3((unsigned)((3 * n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, nullptr, 0, (int)(3 * n),
                           offd, nullptr);
#define DPCG_STRIP_FACTOR(CV, ROWSV, FV)                                                                                       \
    hipLaunchKernelGGL((k_sptrsv_strips<false, CV, ROWSV, FV>), dim3(sp.n_strips), dim3(sp.threads), lds, s, sp.level_ptr_dev,   \
                       sp.nlev, sp.lo_rowptr, sp.lo_col, sp.lo_cpos, sp.lo_val, (const int4 *)sp.meta, (const double2 *)sp.val, \
                       nullptr, diag, sp.W, sp.ring_reach, sp.ticket, nullptr, lv.spin_err, nullptr, (double2 *)fac, xdesc,   \
                       (const double2 *)thr, offd)
    // (the general form holds twice the state per row: half the prefetch chunk keeps it in registers)
    if (sp.rows_per_thread == 1) {
        if (xdesc) DPCG_STRIP_FACTOR(CH / 2, 1, 2);
        else DPCG_STRIP_FACTOR(CH, 1, 1);
    } else {
        if (xdesc) DPCG_STRIP_FACTOR(CH / 4, 2, 2);
        else DPCG_STRIP_FACTOR(CH / 2, 2, 1);
    }
#undef DPCG_STRIP_FACTOR
    return true;
}

// The same through the one-workgroup LDS-ring walk (a factor of <= 131 072 rows whose schedule is ONE ring segment: 2-D grids).
// Returns false when the schedule is not of that form.
// xdesc != null: the general form (cross terms; thr != null: with the drop rule) -- see k_sptrsv_ring_pipe, FACTOR = 2.
bool launch_ring_factor(const Levels &lv, double *diag, double *fac, hipStream_t s, const int32_t *xdesc, const double *thr) {
    if (lv.segments.size() != 1 || lv.strips.n_strips > 0 || !lv.pk_meta) return false;
    const Levels::Segment &seg = lv.segments[0];
    const size_t ring_doubles = (size_t)seg.ring_w * (xdesc ? 4 : 1);
    const size_t lds = ring_doubles * sizeof(double) + (size_t)(seg.hi - seg.lo + 4 * kRingChunk) * sizeof(int);
    if (!(seg.merged && seg.ring_w > 0) || ring_pipe_disabled() || seg.max_width > 1024 || lds > 64 * 1024) return false;
    const int seg_start = lv.level_ptr[seg.lo], width = seg.max_width;
#define DPCG_RING_FACTOR(CV, ROWSV, FV, threads)                                                                              \
    hipLaunchKernelGGL((k_sptrsv_ring_pipe<false, CV, ROWSV, FV>), dim3(1), dim3(threads), lds, s, lv.level_ptr_dev, seg.lo,   \
                       seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, (const int4 *)lv.pk_meta,                      \
                       (const double2 *)lv.pk_val, nullptr, diag, seg_start, seg.ring_w, nullptr, (double2 *)fac, xdesc,      \
                       (const double2 *)thr)
    if (width <= 512) {
        int threads = (width + 63) / 64 * 64;
        threads = threads < 64 ? 64 : threads;
        if (xdesc) DPCG_RING_FACTOR(4, 1, 2, threads);              // (a chunk of 4: the general form's records are 22 registers a row; 6 spilled)
        else DPCG_RING_FACTOR(kRingChunk, 1, 1, threads);
    } else {
        const int threads = ((width + 1) / 2 + 63) / 64 * 64;
        if (xdesc) DPCG_RING_FACTOR(2, 2, 2, threads);
        else DPCG_RING_FACTOR(kRingChunk / 2, 2, 1, threads);
    }
#undef DPCG_RING_FACTOR
    return true;
}

// fac (records by position) -> the factor's CSR values and the plan's level-ordered copy; frow[j] = the factor row at position
// j.  bad: the smallest row with a non-positive pivot, as INT_MAX - row (0: none).
__global__ __launch_bounds__(kBlock) void k_strip_factor_scatter(int64_t n, const int32_t *__restrict__ frow,
                                                                 const int32_t *__restrict__ rp, const double2 *__restrict__ fac,
                                                                 double *__restrict__ fval, const int32_t *__restrict__ lo_rp,
                                                                 double *__restrict__ lo_val, int *bad) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const int i = frow[j];
        const int a = rp[i], b = rp[i + 1], la = lo_rp[j];
        const double2 f01 = fac[2 * j], f2d = fac[2 * j + 1];
        const double l[3] = {f01.x, f01.y, f2d.x};
        for (int q = 0; q < b - a - 1 && q < 3; ++q) {
            fval[a + q] = l[q];
            lo_val[la + q] = l[q];
        }
        fval[b - 1] = f2d.y;
        lo_val[la + (b - a - 1)] = f2d.y;
        if (!(f2d.y > 0.0)) atomicMax(bad, 0x7fffffff - i);
    }
}

void launch_strip_factor_scatter(int64_t n, const int32_t *frow, const int32_t *rp, const double *fac, double *fval,
                                 const int32_t *lo_rp, double *lo_val, int *bad, hipStream_t s) {
    int grid = (int)((n + kBlock - 1) / kBlock);
    grid = grid > 4096 ? 4096 : grid;
    hipLaunchKernelGGL(k_strip_factor_scatter, dim3(grid), dim3(kBlock), 0, s, n, frow, rp, (const double2 *)fac, fval, lo_rp,
                       lo_val, bad);
}

// Level-major solve, way in: dst[j] = src[map[j]], and the solution vector preset to the sync-free kernels' "pending"
// pattern where such a kernel follows.  kLmPerThread elements per thread, all index loads first, then all gathers: two
// memory round trips per thread whatever the count (the index loads do not wait for the `done` word either).
// (An XCD-aware work list -- every XCD gathering from its own eighth of every level -- was measured: no difference.)
constexpr int kLmPerThread = 4;
__global__ __launch_bounds__(kBlock) void k_lm_enter(const int32_t *__restrict__ map, const double *__restrict__ src,
                                                     double *__restrict__ dst, double *__restrict__ pending, int count,
                                                     const int *done) {
    const int base = blockIdx.x * (kBlock * kLmPerThread) + threadIdx.x;
    int a[kLmPerThread];
#pragma unroll
    for (int q = 0; q < kLmPerThread; ++q) {
        const int j = base + q * kBlock;
        a[q] = map[j < count ? j : 0];
    }
    if (done && *done) return;
    double v[kLmPerThread];
#pragma unroll
    for (int q = 0; q < kLmPerThread; ++q) v[q] = src[a[q]];
    const double pend = __longlong_as_double((long long)kPendingBits);
#pragma unroll
    for (int q = 0; q < kLmPerThread; ++q) {
        const int j = base + q * kBlock;
        if (j < count) {
            dst[j] = v[q];
            if (pending) pending[j] = pend;
        }
    }
}

// Level-major solve, way out: dst[i] = src[pos[i]] (the result back in the handle's numbering), optionally with the
// per-workgroup partials of <dotv, dst>.  Same load batching as the way in.
__global__ __launch_bounds__(kBlock) void k_lm_finish(int64_t n, const int32_t *__restrict__ pos, const double *__restrict__ src,
                                                      double *__restrict__ dst, const double *__restrict__ dotv,
                                                      double *__restrict__ part, const int *done, double *__restrict__ refill) {
    __shared__ double sh[4];
    if (done && *done) return;
    double acc = 0.0;
    const double pend = __longlong_as_double((long long)kPendingBits);
    const int64_t step = (int64_t)gridDim.x * (kBlock * kLmPerThread);
    for (int64_t base = (int64_t)blockIdx.x * (kBlock * kLmPerThread) + threadIdx.x; base < n; base += step) {
        int a[kLmPerThread];
        double v[kLmPerThread], w[kLmPerThread];
#pragma unroll
        for (int q = 0; q < kLmPerThread; ++q) {
            const int64_t i = base + q * kBlock;
            a[q] = pos[i < n ? i : 0];
            w[q] = (dotv && i < n) ? dotv[i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kLmPerThread; ++q) v[q] = src[a[q]];
#pragma unroll
        for (int q = 0; q < kLmPerThread; ++q) {
            const int64_t i = base + q * kBlock;
            if (i < n) {
                dst[i] = v[q];
                acc += w[q] * v[q];
                if (refill) refill[i] = pend;
            }
        }
    }
    if (part) {
        const double tot = block_sum(acc, sh);
        if (threadIdx.x == 0) part[blockIdx.x] = tot;
    }
}

bool single_syncfree_segment(const Levels &lv) {
    return lv.level_major && lv.strips.n_strips == 0 && lv.segments.size() == 1 && lv.segments[0].syncfree;
}

void launch_fill_pending(double *v, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_pending, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, nullptr, 0, (int)n, v, nullptr);
}

// The schedule proper: strips, or the segments one after another.  `rows`/`cols`: how the kernels address the vectors (by row
// in the handle's numbering, or by level-order position for a level-major factor); rhs_map / refill: see SptrsvIo::fused_entry.
static void launch_schedule(int64_t n, const Levels &lv, bool upper, bool lm, const int32_t *rows, const int32_t *cols,
                            const double *rhs, double *out, hipStream_t s, const int *done, const int32_t *rhs_map,
                            double *sf_refill) {
    if (lv.strips.n_strips > 0) {
        const Levels::Strips &sp = lv.strips;
        const int count = (int)n;
        hipLaunchKernelGGL(k_strip_prepare, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, sp.rows, rhs, sp.b_lo, out,
                           count, done);
        constexpr int CH = kStripChunk;
        const size_t lds = (size_t)sp.W * sizeof(double) + (size_t)(sp.nlev + 3 * CH + 8) * sizeof(int);
        // DPCG_STRIP_TRACE=1 (development, never under capture): the timeline of the strips of every launch on stderr
        static const bool trace_on = [] { const char *e = getenv("DPCG_STRIP_TRACE"); return e && e[0] == '1'; }();
        long long *trace_buf = nullptr;
        if (trace_on && hipMalloc(&trace_buf, (size_t)sp.n_strips * 4 * sizeof(long long)) == hipSuccess)
            (void)hipMemsetAsync(trace_buf, 0, (size_t)sp.n_strips * 4 * sizeof(long long), s);
#define DPCG_STRIPS(UP, CV, ROWSV)                                                                                        \
    do {                                                                                                                  \
        hipLaunchKernelGGL((k_sptrsv_strips<UP, CV, ROWSV>), dim3(sp.n_strips), dim3(sp.threads), lds, s, sp.level_ptr_dev, \
                           sp.nlev, sp.lo_rowptr, sp.lo_col, sp.lo_cpos, sp.lo_val, (const int4 *)sp.meta,                 \
                           (const double2 *)sp.val, sp.b_lo, out, sp.W, sp.ring_reach, sp.ticket, done, lv.spin_err,       \
                           trace_buf);                                                                                     \
    } while (0)
        if (sp.rows_per_thread == 1) {
            if (upper) DPCG_STRIPS(true, CH, 1);
            else DPCG_STRIPS(false, CH, 1);
        } else {
            if (upper) DPCG_STRIPS(true, CH / 2, 2);
            else DPCG_STRIPS(false, CH / 2, 2);
        }
#undef DPCG_STRIPS
        if (trace_buf) {
            std::vector<long long> h((size_t)sp.n_strips * 4);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h.data(), trace_buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            (void)hipFree(trace_buf);
            long long t0 = h[0];
            for (int q = 0; q < sp.n_strips; ++q) t0 = std::min(t0, h[4 * (size_t)q]);
            fprintf(stderr, "[strip trace] %s, %d strips (100 MHz clock: us)\n", upper ? "upper" : "lower", sp.n_strips);
            for (int q = 0; q < sp.n_strips; ++q)
                fprintf(stderr, "  strip %3d  start %8.2f  end %8.2f  levels %4lld  polls %6lld\n", q, (h[4 * (size_t)q] - t0) / 100.0,
                        (h[4 * (size_t)q + 1] - t0) / 100.0, h[4 * (size_t)q + 3], h[4 * (size_t)q + 2]);
        }
        return;
    }
    int seg_index = -1;
    for (const auto &seg : lv.segments) {
        ++seg_index;
        if (seg.syncfree) {
            const int j0 = lv.level_ptr[seg.lo], cnt = lv.level_ptr[seg.hi] - j0;
            const int nblocks = (cnt + kBlock - 1) / kBlock;
            // persistent grid ~ a few times the widest level: the front and the rows about to join it (DPCG_SF_FACTOR:
            // development knob for that multiple)
            static const double factor_env = [] { const char *e = getenv("DPCG_SF_FACTOR"); return e ? atof(e) : 0.0; }();
            // (level-major factors: wide levels, where the polling loads of rows ahead of the front are what costs.  Scrambled
            // 1M-DoF IC(0), us per apply at 0.35 / 0.5 / 0.7 / 0.85 / 1 / 1.2 / 1.5 / 2 / 3 / 4 x the widest level:
            // 193 / 165 / 148 / 143 / 148 / 154 / 157 / 173 / 239 / 358)
            // The best window shrinks, relative to the level, as the levels widen (PCG update at 0.85 / 1.7 / 3 / 6 x the widest level:
            // 64^3, 14.5K rows a level: 85.6 / 72.2 / 70.9 / 102 us; 256^2, 5K: 50.9 / 45.2 / 44.8 / 46.0; 40^3, 3.4K: 84.1 / 76.7 /
            // 78.7 / 85.0; 1024^2, 58K: 130 / 146 / 153 / 154; 100^3, 53K: 136 / 153 (1.7)): 2.5 x up to a MEAN width of 16K rows, 0.85 x from 40K.
            const double w_lm = (double)cnt / (double)(seg.hi - seg.lo);          // mean level width of the segment
            const double factor_lm = w_lm <= 16000.0 ? 2.5 : (w_lm >= 40000.0 ? 0.85 : 2.5 - (w_lm - 16000.0) / 24000.0 * 1.65);
            const double factor = factor_env > 0.0 ? factor_env : (lm ? factor_lm : 2.0);
            int grid = ((int)(factor * seg.max_width) + kBlock - 1) / kBlock + 4;
            grid = grid < 16 ? 16 : grid;
            grid = grid > nblocks ? nblocks : grid;
            grid = grid > 2048 ? 2048 : grid;       // 8 workgroups per CU: all resident
            if (lm && lv.sfs_blk) {                     // CSR-stream form (see k_sptrsv_syncfree_stream)
                constexpr int SB = kSfsBlock;
                int g = ((int)(factor * seg.max_width) + SB - 1) / SB + 4;
                g = g > lv.sfs_nblk ? lv.sfs_nblk : g;
                g = g > 2048 * 256 / SB ? 2048 * 256 / SB : g;
                if (upper)
                    hipLaunchKernelGGL((k_sptrsv_syncfree_stream<true, SB>), dim3(g), dim3(SB), 0, s, lv.sfs_blk, lv.sfs_nblk, lv.lo_rowptr,
                                       cols, lv.lo_val, rhs, out, reinterpret_cast<unsigned int *>(lv.tickets + seg_index), done,
                                       lv.spin_err, rhs_map, sf_refill);
                else
                    hipLaunchKernelGGL((k_sptrsv_syncfree_stream<false, SB>), dim3(g), dim3(SB), 0, s, lv.sfs_blk, lv.sfs_nblk, lv.lo_rowptr,
                                       cols, lv.lo_val, rhs, out, reinterpret_cast<unsigned int *>(lv.tickets + seg_index), done,
                                       lv.spin_err, rhs_map, sf_refill);
                continue;
            }
            if (lm) {                                   // (lm_out was filled with the pending pattern by the way-in pass)
#define DPCG_SF_LM(UP, BSV, WV)                                                                                              \
    do {                                                                                                                     \
        const int nb = (cnt + BSV - 1) / BSV;                                                                                \
        int g = ((int)(factor * seg.max_width) + BSV - 1) / BSV + 4;                                                         \
        g = g > (nb + kSfSub - 1) / kSfSub ? (nb + kSfSub - 1) / kSfSub : g;                                                 \
        g = g > 2048 * 256 / BSV ? 2048 * 256 / BSV : g;                                                                     \
        hipLaunchKernelGGL((k_sptrsv_syncfree_rec<UP, BSV, WV, kSfSub>), dim3(g), dim3(BSV), 0, s, j0, cnt, lv.lo_rowptr, cols, \
                           lv.lo_val, lv.sf_meta, lv.sf_val, rhs, out,                                                       \
                           reinterpret_cast<unsigned int *>(lv.tickets + seg_index), nb, done, lv.spin_err, rhs_map,        \
                           sf_refill);                                                                                       \
    } while (0)
                // 512 rows per ticket (measured on the scrambled 1M-DoF factor: 256 / 512 / 1024 rows: 241 / 233 / 233-254 us per apply)
                if (lv.rec_w == 14) {
                    if (upper) DPCG_SF_LM(true, 512, 14);
                    else DPCG_SF_LM(false, 512, 14);
                } else if (lv.rec_w == 6) {
                    if (upper) DPCG_SF_LM(true, 512, 6);
                    else DPCG_SF_LM(false, 512, 6);
                } else {
                    if (upper) DPCG_SF_LM(true, 512, 3);
                    else DPCG_SF_LM(false, 512, 3);
                }
#undef DPCG_SF_LM
                continue;
            }
            hipLaunchKernelGGL(k_fill_pending, dim3(nblocks), dim3(kBlock), 0, s, rows, j0, cnt, out, done);
#define DPCG_SF(UP, WV)                                                                                                      \
    hipLaunchKernelGGL((k_sptrsv_syncfree_rec<UP, kBlock, WV>), dim3(grid), dim3(kBlock), 0, s, j0, cnt, lv.lo_rowptr, cols,  \
                       lv.lo_val, lv.sf_meta, lv.sf_val, rhs, out, reinterpret_cast<unsigned int *>(lv.tickets + seg_index),  \
                       nblocks, done, lv.spin_err, nullptr, nullptr)
            if (lv.rec_w == 14) {
                if (upper) DPCG_SF(true, 14);
                else DPCG_SF(false, 14);
            } else if (lv.rec_w == 6) {
                if (upper) DPCG_SF(true, 6);
                else DPCG_SF(false, 6);
            } else {
                if (upper) DPCG_SF(true, 3);
                else DPCG_SF(false, 3);
            }
#undef DPCG_SF
            continue;
        }
        if (seg.merged && seg.ring_w > 0 && lv.pk_meta && !ring_pipe_disabled() && seg.max_width <= 1024 &&
            (size_t)seg.ring_w * sizeof(double) + (size_t)(seg.hi - seg.lo + 4 * kRingChunk) * sizeof(int) <= 64 * 1024) {
            const int seg_start = lv.level_ptr[seg.lo], seg_rows = lv.level_ptr[seg.hi] - seg_start;
            const size_t lds = (size_t)seg.ring_w * sizeof(double) + (size_t)(seg.hi - seg.lo + 4 * kRingChunk) * sizeof(int);
            const int width = seg.max_width;
            hipLaunchKernelGGL(k_gather_lo, dim3((seg_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, s, lv.rows, rhs,
                               lv.b_lo, seg_start, seg_rows, done);
#define DPCG_RING_PIPE(UP, CV, ROWSV, threads)                                                                      \
    hipLaunchKernelGGL((k_sptrsv_ring_pipe<UP, CV, ROWSV>), dim3(1), dim3(threads), lds, s, lv.level_ptr_dev, seg.lo,  \
                       seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, (const int4 *)lv.pk_meta,            \
                       (const double2 *)lv.pk_val, lv.b_lo, out, seg_start, seg.ring_w, done)
            if (width <= 512) {                              // one row per thread
                int threads = (width + 63) / 64 * 64;
                threads = threads < 64 ? 64 : threads;
                if (upper) DPCG_RING_PIPE(true, kRingChunk, 1, threads);
                else DPCG_RING_PIPE(false, kRingChunk, 1, threads);
            } else {                                         // two rows per thread
                const int threads = ((width + 1) / 2 + 63) / 64 * 64;
                if (upper) DPCG_RING_PIPE(true, kRingChunk / 2, 2, threads);
                else DPCG_RING_PIPE(false, kRingChunk / 2, 2, threads);
            }
#undef DPCG_RING_PIPE
            continue;
        }
        if (seg.merged && seg.ring_w > 0) {
            const size_t lds = (size_t)seg.ring_w * sizeof(double);
            const int seg_start = lv.level_ptr[seg.lo];
            // as few waves as the widest level needs (two rows per thread): a barrier among 4 waves is cheaper
            int width = 0;
            for (int q = seg.lo; q < seg.hi; ++q) width = std::max(width, lv.level_ptr[q + 1] - lv.level_ptr[q]);
            int threads = ((width + 1) / 2 + 63) / 64 * 64;
            threads = threads < 64 ? 64 : (threads > kMergedBlock ? kMergedBlock : threads);
            if (upper)
                hipLaunchKernelGGL(k_sptrsv_ring<true>, dim3(1), dim3(threads), lds, s, lv.rows, lv.level_ptr_dev,
                                   seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, rhs, out, seg_start,
                                   seg.ring_w, done);
            else
                hipLaunchKernelGGL(k_sptrsv_ring<false>, dim3(1), dim3(threads), lds, s, lv.rows, lv.level_ptr_dev,
                                   seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, rhs, out, seg_start,
                                   seg.ring_w, done);
            continue;
        }
        if (seg.merged) {
            if (upper)
                hipLaunchKernelGGL(k_sptrsv_merged<true>, dim3(1), dim3(kMergedBlock), 0, s, lv.rows, lv.level_ptr_dev,
                                   seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_val, rhs, out, done);
            else
                hipLaunchKernelGGL(k_sptrsv_merged<false>, dim3(1), dim3(kMergedBlock), 0, s, lv.rows,
                                   lv.level_ptr_dev, seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_val, rhs, out, done);
            continue;
        }
        for (int l = seg.lo; l < seg.hi; ++l) {
            const int j0 = lv.level_ptr[l], cnt = lv.level_ptr[l + 1] - j0;
            const int grid = (cnt + kBlock - 1) / kBlock;
#define DPCG_TRSV(KERNEL, UP) \
    hipLaunchKernelGGL(KERNEL<UP>, dim3(grid), dim3(kBlock), 0, s, lv.rows, j0, cnt, lv.lo_rowptr, lv.lo_col, lv.lo_val, rhs, out, done)
            if (lv.sf_meta && (lm || !level_rec_disabled())) {
#define DPCG_LEVEL_REC(UP, WV)                                                                                            \
    hipLaunchKernelGGL((k_sptrsv_level_rec<UP, WV>), dim3(grid), dim3(kBlock), 0, s, j0, cnt, lv.lo_rowptr, cols, lv.lo_val, \
                       lv.sf_meta, lv.sf_val, rhs, out, done)
                if (lv.rec_w == 14) {
                    if (upper) DPCG_LEVEL_REC(true, 14);
                    else DPCG_LEVEL_REC(false, 14);
                } else if (lv.rec_w == 6) {
                    if (upper) DPCG_LEVEL_REC(true, 6);
                    else DPCG_LEVEL_REC(false, 6);
                } else {
                    if (upper) DPCG_LEVEL_REC(true, 3);
                    else DPCG_LEVEL_REC(false, 3);
                }
#undef DPCG_LEVEL_REC
            } else if (lv.stream_ok) {
                if (upper) DPCG_TRSV(k_sptrsv_level_stream, true);
                else DPCG_TRSV(k_sptrsv_level_stream, false);
            } else {
                if (upper) DPCG_TRSV(k_sptrsv_level, true);
                else DPCG_TRSV(k_sptrsv_level, false);
            }
#undef DPCG_TRSV
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Colour sweep: one VERY wide level of a level-major factor (IC(0) in multicolour order: 2 .. 4 levels of n / colours rows)
// as ONE bandwidth-bound launch with everything an apply needs fused in, so that z = L^-T (L^-1 r) is 2 x levels launches
// and nothing else (no way-in gather pass, no way-out pass, no dot launch):
//   * CSR-stream on the level-ordered copy (positions as columns): a workgroup streams the contiguous val / position
//     segment of 256 consecutive rows, parks v * out[position] in LDS, thread j subtracts row j's products in column order
//     and divides by the diagonal -- the arithmetic and order of k_sptrsv_level_stream, bit-identical to sequential
//     substitution; a row without off-diagonal entries (every row of the first colour) costs its 4-byte extent, not a
//     96-byte record;
//   * the right-hand side is gathered through `src_map` (the handle's r through the row list; the lower solve's result
//     through lm_from_lower), the result goes to out[position] for the later levels and, when `dst` is given, to
//     dst[rows[position]] -- the handle's numbering; with `dotv` the workgroup's share of <dotv, result> goes to
//     part[blockIdx.x] (persistent grid: the partial count does not depend on the level's size).
// ------------------------------------------------------------------------------------------------
template <bool UPPER>
__global__ __launch_bounds__(kBlock) void k_lm_sweep(int j0, int count, const int32_t *__restrict__ lo_rp,
                                                     const int32_t *__restrict__ lo_cp, const double *__restrict__ lo_v,
                                                     const double *__restrict__ src, const int32_t *__restrict__ src_map,
                                                     double *out, double *__restrict__ dst, const int32_t *__restrict__ rows,
                                                     const double *__restrict__ dotv, double *__restrict__ part,
                                                     double *__restrict__ out2, const int32_t *__restrict__ map2, int accumulate,
                                                     const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    constexpr int U = kStreamCap / kBlock;
    __shared__ double prod[kStreamCap];
    __shared__ double sh[4];
    const int t = threadIdx.x;
    const int nblk = (count + kBlock - 1) / kBlock;
    double dot = 0.0;
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int jb = j0 + blk * kBlock;
        const int jend = (jb + kBlock < j0 + count) ? jb + kBlock : j0 + count;
        const int j = jb + t;
        const int base = lo_rp[jb];
        const int cnt = lo_rp[jend] - base;
        int rs = 0, re = 0, own = 0;
        double bi = 0.0, dv = 0.0;
        if (j < jend) {
            rs = lo_rp[j] - base;
            re = lo_rp[j + 1] - base;
            own = rows[j];
            bi = src[src_map ? src_map[j] : j];
            if (dotv) dv = dotv[own];
        }
        const int last = cnt > 0 ? cnt - 1 : 0;
        int c[U];
        double a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = t + u * kBlock;
            const int kk = k < cnt ? k : last;
            c[u] = lo_cp[base + kk];
            a[u] = lo_v[base + kk];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = t + u * kBlock;
            // the diagonal's slot (its column is the row itself: not solved yet) is parked too and never read
            if (k < cnt) prod[k] = a[u] * out[c[u]];
        }
        __syncthreads();
        if (j < jend) {
            double acc = bi;
            const int ks = UPPER ? rs + 1 : rs, ke = UPPER ? re : re - 1;
            for (int k = ks; k < ke; ++k) acc -= prod[k];
            const double d = lo_v[base + (UPPER ? rs : re - 1)];
            double y = acc / d;
            if (out2) {                    // (see k_lm_sweep_tile: the row also opens the paired upper solve)
                y = y / d;
                out2[map2[j]] = y;
            } else {
                out[j] = y;
            }
            if (dst) dst[own] = y;
            dot += dv * y;
        }
        __syncthreads();          // prod is reused by the next block
    }
    if (part) {     // (`accumulate`: an earlier launch of this apply has stored its share in the slot already)
        const double tot = block_sum(dot, sh);
        if (t == 0) part[blockIdx.x] = accumulate ? part[blockIdx.x] + tot : tot;
    }
}

// The same sweep with the solution entries a block gathers staged in LDS (Levels::sw_*: the x-tile plan of the level, as
// k_spmv_tile's): the block's chunks of `out` arrive by coalesced 512-byte wave loads one block ahead instead of one gather
// per entry through the L2, the matrix stream is values + 16-bit local indices read as aligned pairs, and a row's diagonal
// travels through its product slot (local index kTileDiag: times 1.0).  Same arithmetic, same order, same bits.
// `out2` (the last level of a lower solve paired with an upper one whose first level holds the same rows): the row also
// opens the upper solve -- z = y / d goes to out2[map2[j]], to dst and into the dot product; y itself is not stored.
// NT: the factor's stream (values, local indices) is read non-temporally -- a factor beyond the Infinity Cache, as k_spmv_tile's NT.
template <bool UPPER, int XT, bool NT = false>
__global__ __launch_bounds__(kBlock) void k_lm_sweep_tile(int j0, int count, int n_total, const int32_t *__restrict__ lo_rp,
                                                          const double *__restrict__ lo_v, const uint16_t *__restrict__ lidx,
                                                          const int32_t *__restrict__ chunks, const int32_t *__restrict__ nchunks,
                                                          int tile_doubles, const double *__restrict__ src,
                                                          const int32_t *__restrict__ src_map, double *out, double *__restrict__ dst,
                                                          const int32_t *__restrict__ rows, const double *__restrict__ dotv,
                                                          double *__restrict__ part, double *__restrict__ out2,
                                                          const int32_t *__restrict__ map2, int accumulate, const int *done,
                                                          int cyclic) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    constexpr int U = kStreamCap / kBlock;
    constexpr int UP = U / 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *xs = smem;                      // the staged chunks of `out`
    double *prod = smem + tile_doubles;     // products, kStreamCap + 4 doubles
    double *sh = prod + kStreamCap + 4;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nblk = (count + kBlock - 1) / kBlock;
    // slabs, or -- a factor beyond the Infinity Cache -- the blocks dealt out cyclically so that the grid walks the streams together
    // (see k_spmv_tile)
    int b_lo, b_hi, b_step = 1;
    if (cyclic) {
        b_lo = virtual_block();               // (inside a pass an XCD takes one contiguous run of blocks: see k_spmv_tile)
        b_hi = nblk;
        b_step = gridDim.x;
    } else {
        split_range(nblk, virtual_block(), b_lo, b_hi);
    }
    typedef double VPair __attribute__((ext_vector_type(2)));
    typedef unsigned short IPair __attribute__((ext_vector_type(2)));
    constexpr int XP = (XT * (kBlock / 64) + 7) / 8;
    VPair a[UP], xt[XP];
    IPair li[UP];
    int cnt = 0, base = 0, rs = 0, re = 0, nc = 0, own = 0, m2 = 0;
    double bi = 0.0, dv = 0.0;
    bool live = false;
    auto fetch = [&](int blk) {
        const int jb = j0 + blk * kBlock;
        const int jend = (jb + kBlock < j0 + count) ? jb + kBlock : j0 + count;
        const int j = jb + t;
        base = lo_rp[jb] & ~1;
        cnt = lo_rp[jend] - base;
        live = j < jend;
        rs = re = 0;
        if (live) {
            struct __attribute__((packed, aligned(4))) Ext { int32_t s, e; };
            const Ext ext = *reinterpret_cast<const Ext *>(lo_rp + j);
            rs = ext.s - base;
            re = ext.e - base;
            own = rows[j];
            bi = src[src_map ? src_map[j] : j];
            if (dotv) dv = dotv[own];
            if (map2) m2 = map2[j];
        }
        const int lastp = cnt > 0 ? (cnt - 1) / 2 : 0;
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int pr = t + u * kBlock;
            const int64_t kabs = base + 2 * (int64_t)(pr <= lastp ? pr : lastp);
            const VPair *vp = reinterpret_cast<const VPair *>(lo_v + kabs);          // (aligned pairs: see k_spmv_tile)
            const IPair *ip = reinterpret_cast<const IPair *>(lidx + kabs);
            a[u] = NT ? __builtin_nontemporal_load(vp) : *vp;
            li[u] = NT ? __builtin_nontemporal_load(ip) : *ip;
        }
        nc = nchunks[blk];
        const int32_t *__restrict__ cl = chunks + (int64_t)blk * kTileMaxChunks;
        const int my_cid = cl[lane < kTileMaxChunks ? lane : 0];
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int ci = 2 * (wv + u * (kBlock / 64)) + (lane >> 5);
            const int cid = __shfl(my_cid, ci);
            const int64_t gi = (int64_t)(ci < nc ? cid : 0) * kTileChunk + 2 * (lane & 31);
            xt[u] = *reinterpret_cast<const VPair *>(out + (gi < n_total ? gi : ((n_total - 1) & ~1)));
        }
    };
    if (b_lo < b_hi) fetch(b_lo);
    double dot = 0.0;
    for (int blk = b_lo; blk < b_hi; blk += b_step) {
        const int j = j0 + blk * kBlock + t;
        const int ks = rs, ke = re, cnt_cur = cnt, own_c = own, m2_c = m2;
        const double bi_c = bi, dv_c = dv;
        const bool live_c = live;
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int ci = 2 * (wv + u * (kBlock / 64)) + (lane >> 5);
            if (ci < nc) *reinterpret_cast<VPair *>(xs + ci * kTileChunk + 2 * (lane & 31)) = xt[u];
        }
        __syncthreads();                    // tile complete (and every thread is past the previous row sums)
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int k = 2 * (t + u * kBlock);
            if (k < cnt_cur) {
                const unsigned short l0 = li[u][0], l1 = li[u][1];
                const double x0 = xs[l0 == kTileDiag ? 0 : l0], x1 = xs[l1 == kTileDiag ? 0 : l1];
                double2 pp;
                pp.x = a[u][0] * (l0 == kTileDiag ? 1.0 : x0);
                pp.y = a[u][1] * (l1 == kTileDiag ? 1.0 : x1);
                *reinterpret_cast<double2 *>(prod + k) = pp;
            }
        }
        if (blk + b_step < b_hi) fetch(blk + b_step);  // the next block's stream, right-hand side and chunks are in flight from here on
        __syncthreads();
        if (live_c) {
            double acc = bi_c;
            const int k0 = UPPER ? ks + 1 : ks, k1 = UPPER ? ke : ke - 1;
            for (int k = k0; k < k1; ++k) acc -= prod[k];
            const double d = prod[UPPER ? ks : ke - 1];
            double y = acc / d;
            if (out2) {
                y = y / d;
                out2[m2_c] = y;
            } else {
                out[j] = y;
            }
            if (dst) dst[own_c] = y;
            dot += dv_c * y;
        }
    }
    if (part) {
        __syncthreads();
        const double tot = block_sum(dot, sh);
        if (t == 0) part[blockIdx.x] = accumulate ? part[blockIdx.x] + tot : tot;
    }
}

// The whole factor as colour sweeps (Levels::sweep).  `src` / `src_map`: where the right-hand side comes from; dst: the result in
// the handle's numbering (null: by position in lv.lm_out only).
static void launch_sweeps(const Levels &lv, bool upper, const double *src, const int32_t *src_map, double *dst, const double *dotv,
                          double *part, hipStream_t s, const int *done, const SptrsvIo *io) {
    const int first = (io && io->skip_first) ? 1 : 0;
    for (int l = first; l < lv.n_levels; ++l) {
        const int j0 = lv.level_ptr[l], cnt = lv.level_ptr[l + 1] - j0;
        // the last level of a lower solve whose rows open the paired upper solve: z instead of y, into the upper numbering
        const bool emit = !upper && io && io->pair_out && l == lv.n_levels - 1;
        const int grid = lv.sweep_grid;
        double *pl = part;
        // the first launch of the apply that sums <r,z> stores its shares, the later ones add theirs: the lower solve's last level
        // when it opens the upper one, otherwise the upper solve's first level
        const int acc = emit ? 0 : ((upper && l == 0) ? 0 : 1);
        double *o2 = emit ? io->pair_out : nullptr;
        const int32_t *m2 = emit ? lv.lm_to_upper : nullptr;
        double *d = emit ? io->pair_dst : dst;
        const double *dw = (emit || upper) ? dotv : nullptr;
        if (!dw) pl = nullptr;
        const int mc = l < (int)lv.sw_max_chunks.size() ? lv.sw_max_chunks[l] : 0;
        if (mc > 0) {
            const int tile_doubles = mc * kTileChunk;
            const size_t lds = (size_t)(tile_doubles + kStreamCap + 8) * sizeof(double);
            const int32_t *ch = lv.sw_chunks + (size_t)lv.sw_blk0[l] * kTileMaxChunks, *nch = lv.sw_nchunks + lv.sw_blk0[l];
#define DPCG_SWEEP_TILE(UP_, XT_)                                                                                              \
    do {                                                                                                                       \
    if (lv.sweep_nt)                                                                                                           \
        hipLaunchKernelGGL((k_lm_sweep_tile<UP_, XT_, true>), dim3(grid), dim3(kBlock), lds, s, j0, cnt, (int)lv.level_ptr.back(), \
                           lv.lo_rowptr, lv.lo_val, lv.sw_lidx, ch, nch, tile_doubles, src, src_map, lv.lm_out, d, lv.rows, dw, pl, \
                           o2, m2, acc, done, lv.sweep_cyclic ? 1 : 0);                                                            \
    else                                                                                                                       \
    hipLaunchKernelGGL((k_lm_sweep_tile<UP_, XT_>), dim3(grid), dim3(kBlock), lds, s, j0, cnt, (int)lv.level_ptr.back(), \
                       lv.lo_rowptr, lv.lo_val, lv.sw_lidx, ch, nch, tile_doubles, src, src_map, lv.lm_out, d, lv.rows, dw, pl, \
                       o2, m2, acc, done, 0);                                                                                  \
    } while (0)
            if (upper) {
                if (mc <= 20) DPCG_SWEEP_TILE(true, 5);
                else DPCG_SWEEP_TILE(true, (kTileMaxChunks * kTileChunk / kBlock));
            } else {
                if (mc <= 20) DPCG_SWEEP_TILE(false, 5);
                else DPCG_SWEEP_TILE(false, (kTileMaxChunks * kTileChunk / kBlock));
            }
#undef DPCG_SWEEP_TILE
            continue;
        }
        if (upper)
            hipLaunchKernelGGL(k_lm_sweep<true>, dim3(grid), dim3(kBlock), 0, s, j0, cnt, lv.lo_rowptr, lv.lo_cpos, lv.lo_val,
                               src, src_map, lv.lm_out, d, lv.rows, dw, pl, o2, m2, acc, done);
        else
            hipLaunchKernelGGL(k_lm_sweep<false>, dim3(grid), dim3(kBlock), 0, s, j0, cnt, lv.lo_rowptr, lv.lo_cpos, lv.lo_val,
                               src, src_map, lv.lm_out, d, lv.rows, dw, pl, o2, m2, acc, done);
    }
}

void launch_sptrsv(const CsrDev &T, const Levels &lv, bool upper, const double *rhs, double *out, hipStream_t s,
                   const int *done, SptrsvIo *io) {
    const int64_t n = T.n;    // the level-ordered copy in `lv` carries the factor
    if (io) io->dot_done = false;
    const bool lm = lv.level_major && lv.strips.n_strips == 0;
    if (!lm) {
        launch_schedule(n, lv, upper, false, lv.rows, lv.lo_col, rhs, out, s, done, nullptr, nullptr);
        return;
    }
    if (lv.sweep) {
        // colour sweeps: every level one fused launch; the upper solve of a paired apply reads the lower one's result by position
        const bool chained = io && io->lm_in;
        const bool keep = io && io->keep_lm;
        const bool dot = io && io->dot_with && io->dot_part;
        launch_sweeps(lv, upper, chained ? io->lm_in : rhs, chained ? lv.lm_from_lower : lv.rows, keep ? nullptr : out,
                      dot ? io->dot_with : nullptr, dot ? io->dot_part : nullptr, s, done, io);
        if (dot && upper) {
            io->dot_done = true;
            io->dot_count = lv.sweep_grid;
        }
        return;
    }
    // Level-major: way in (a pass, or fused into the single sync-free kernel), the schedule on positions, way out.
    const bool fused_entry = io && io->fused_entry && single_syncfree_segment(lv);
    const bool chained = io && io->lm_in;
    const int32_t *map = chained ? lv.lm_from_lower : lv.rows;
    const double *src = chained ? io->lm_in : rhs;
    if (fused_entry) {
        launch_schedule(n, lv, upper, true, nullptr, lv.lo_cpos, src, lv.lm_out, s, done, map, upper ? nullptr : io->refill);
    } else {
        bool any_syncfree = false;
        for (const auto &seg : lv.segments) any_syncfree = any_syncfree || seg.syncfree;
        const int count = (int)n;
        hipLaunchKernelGGL(k_lm_enter, dim3((count + kBlock * kLmPerThread - 1) / (kBlock * kLmPerThread)), dim3(kBlock), 0, s, map,
                           src, lv.lm_rhs, any_syncfree ? lv.lm_out : nullptr, count, done);
        launch_schedule(n, lv, upper, true, nullptr, lv.lo_cpos, lv.lm_rhs, lv.lm_out, s, done, nullptr, nullptr);
    }
    if (io && io->keep_lm) return;                            // the next solve picks the result up in lm_out
    const bool dot = io && io->dot_with && io->dot_part && io->dot_grid > 0;
    int grid = dot ? io->dot_grid : (int)((n + kBlock * kLmPerThread - 1) / (kBlock * kLmPerThread));
    grid = grid > 2048 ? 2048 : grid;
    hipLaunchKernelGGL(k_lm_finish, dim3(grid), dim3(kBlock), 0, s, n, lv.lm_pos, lv.lm_out, out, dot ? io->dot_with : nullptr,
                       dot ? io->dot_part : nullptr, done, fused_entry ? io->refill : nullptr);
    if (dot) {
        io->dot_done = true;
        io->dot_count = grid;
    }
    // a standalone lower solve consumed the "pending" preset of lm_out: put it back (Levels: the invariant)
    if (!upper && !fused_entry && single_syncfree_segment(lv))
        hipLaunchKernelGGL(k_fill_pending, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, nullptr, 0, (int)n,
                           lv.lm_out, done);
}

// ------------------------------------------------------------------------------------------------
// IC(0) numeric factorisation, level by level (stands in for ilupp.ichol0, test.py:83).
// Row i of L needs the finished rows j < i of its own pattern -- the dependency DAG of the lower
// solve, so the same level sets apply.  One thread owns a row; every sum runs over ascending columns,
// one product and one subtraction at a time (two roundings): the order of the CPU restatement, so the
// factor is bit-identical to it.  lv holds tril(A) on entry and L on exit (diagonal last in a row).
// ------------------------------------------------------------------------------------------------
// DROP (ICT, oracle/oracle.py::ict): an off-diagonal entry v = acc / L_jj is stored as 0 when |v| * L_jj < tau * colnorm[j].
template <bool DROP>
__global__ __launch_bounds__(kBlock) void k_ic0_level(const int32_t *__restrict__ rows, int j0, int count,
                                                      const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                      double *lv, int *bad, const double *__restrict__ colnorm, double tau) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= count) return;
    const int i = rows[j0 + idx];
    const int s_i = rp[i], e_i = rp[i + 1];
    for (int k = s_i; k < e_i; ++k) {
        const int j = ci[k];
        const int s_j = rp[j], e_j = rp[j + 1];
        double acc = lv[k];
        int a = s_i, b = s_j;
        while (a < k && b < e_j - 1) {
            const int ca = ci[a], cb = ci[b];
            if (ca == cb) {
                acc -= lv[a] * lv[b];
                ++a;
                ++b;
            } else if (ca < cb) ++a;
            else ++b;
        }
        if (j < i) {
            const double d = lv[e_j - 1];
            double v = acc / d;
            if (DROP && fabs(v) * d < tau * colnorm[j]) v = 0.0;
            lv[k] = v;
        } else {
            if (!(acc > 0.0)) atomicExch(bad, i + 1);
            lv[k] = sqrt(acc);
        }
    }
}

void launch_ic0_level(const int32_t *rows, int j0, int count, const int32_t *rp, const int32_t *ci, double *lv, int *bad,
                      hipStream_t s, const double *colnorm, double tau) {
    if (colnorm)
        hipLaunchKernelGGL(k_ic0_level<true>, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, rows, j0, count, rp, ci,
                           lv, bad, colnorm, tau);
    else
        hipLaunchKernelGGL(k_ic0_level<false>, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, rows, j0, count, rp,
                           ci, lv, bad, colnorm, tau);
}

}  // namespace dpcg
