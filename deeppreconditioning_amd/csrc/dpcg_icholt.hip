// Thresholded incomplete Cholesky as ILU++ defines it: the device routine behind dpcg_set_precond_icholt (contract:
// oracle/oracle.py::icholt; it replaces `ilupp.icholt(matrix, add_fill_in=1, threshold=0.1)`, the DEFAULT incomplete-Cholesky
// technique of the reference's harness, test.py:81-88).  ILU++'s dual-threshold rule (Saad's ILUT(p, tau) on the lower triangle):
// column k of L is  w = A[k:, k] - sum_{j < k, L_kj kept} L_kj L[k:, j]  (ascending j), d = sqrt(w_k); of the off-diagonal
// candidates those below tau * ||w||_2 go, of the rest the nnz(A[k+1:, k]) + add_fill_in largest stay (ties: smaller row).
//
// The pattern of a column depends on the VALUES of every earlier column that reaches it (which fill survives), so unlike
// IC(0) / IC(level) there is no symbolic phase to derive level sets from, and in a banded numbering every column depends on
// its predecessor through fill: the algorithm is a sequence of n small steps.  It runs as ONE wave that walks the columns in
// order -- the 64 lanes share a column's work: the loads of its dependencies' entries, the updates of distinct candidates
// (an LDS hash table keyed by row), the rank-counting sorts of the few candidates -- with every sum in the oracle's order
// (ascending j, then ascending row; one product and one subtraction at a time; -ffp-contract=off), so the factor equals the
// CPU restatement bit for bit.  Two dependent memory round trips per column (see k_icholt); times by size: tools/icholt_probe.py,
// profiles/r04_icholt_probe.txt.  It is the setup of a technique the reference runs on ~2K-row systems.
#include "dpcg_host.h"
#include "dpcg_prims.h"

namespace dpcg {

constexpr int kIctCap = 64;       // kept entries per row and per column of L (beyond: DPCG_ERR_INVALID)
constexpr int kIctCand = 256;     // candidates of one column
constexpr int kIctHash = 1024;    // LDS hash slots (power of two)

enum { ICHOLT_OK = 0, ICHOLT_PIVOT = 1, ICHOLT_CAND = 2, ICHOLT_ROWCAP = 3, ICHOLT_COLCAP = 4, ICHOLT_NODIAG = 5 };

// The kernel is ONE wave: what a lane stores, another lane of the same wave loads later.  The vector-memory operations of one wave
// go through its CU's L1 in order, so wavefront scope is all the ordering these hand-offs need -- plain loads and stores, no wait
// for a store's acknowledgement (agent-scope loads + `s_waitcnt vmcnt(0)` per column cost two to three extra round trips), and
// the fences / barriers below are compiler-level only.
__device__ __forceinline__ int ld_i(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ double ld_d(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void st_i(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void st_d(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
// the lanes of the wave meet (LDS and global memory written before are read after): no instruction, only an order for the compiler
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// rcnt[i]: kept entries of row i so far; row i's list at [i * kIctCap ..]: column j, L_ij, and rcc = (position after L_ij in column
// j's list) + 256 * (kept entries of column j), i.e. where the entries of column j below row i start and end; column j's kept
// (row, value) at [j * kIctCap ..], rows ascending; diag[k] = L_kk.  status[0] = code, status[1] = column.
// Per column two dependent round trips remain: the entries of its dependencies, and the counters of the rows it appends to (beside
// which the next row's list is fetched; the entry this column itself adds to that list is handed over in registers).  The next
// row of A is fetched a column ahead.
__global__ __launch_bounds__(64) void k_icholt(int n, const int32_t *__restrict__ arp, const int32_t *__restrict__ aci,
                                               const double *__restrict__ av, int add_fill, double tau, int *rcnt, int *rcol,
                                               int *rcc, double *rval, int *crow, double *cval, double *diag, int *status) {
    __shared__ int hkey[kIctHash];
    __shared__ double hval[kIctHash];
    __shared__ int used[kIctCand];
    __shared__ int ci[kIctCand], si[kIctCand], keep[kIctCand];
    __shared__ double cv[kIctCand], sv[kIctCand];
    __shared__ int dj[kIctCap], dst[kIctCap], doff[kIctCap + 1];
    __shared__ double dv[kIctCap];
    __shared__ int s_nused, s_err, s_pk, s_fwd;
    __shared__ double s_diag, s_fwd_l;
    const int lane = threadIdx.x;
    for (int q = lane; q < kIctHash; q += 64) hkey[q] = -1;
    if (lane == 0) { s_nused = 0; s_err = 0; }
    wave_sync();
    // slot of `key` in the table (inserted with value 0 when absent)
    auto slot_of = [&](int key) -> int {
        unsigned hsl = ((unsigned)key * 2654435761u) >> 22;
        for (;;) {
            const int old = atomicCAS(&hkey[hsl], -1, key);
            if (old == -1) {
                const int u = atomicAdd(&s_nused, 1);
                if (u < kIctCand) used[u] = (int)hsl;
                else s_err = ICHOLT_CAND;
                hval[hsl] = 0.0;
                return (int)hsl;
            }
            if (old == key) return (int)hsl;
            hsl = (hsl + 1) & (kIctHash - 1);
        }
    };
    // row k of A a column ahead: p0 = arp[k], p1 = arp[k + 1], p2 = arp[k + 2]; this lane's entry of row k in (ac, ax)
    int p0 = arp[0], p1 = arp[1], p2 = n > 1 ? arp[2] : arp[1];
    int ac = -1;
    double ax = 0.0;
    if (p0 + lane < p1) { ac = aci[p0 + lane]; ax = av[p0 + lane]; }
    // row k's list of L, fetched while column k - 1 was being appended (row 0: empty)
    int m = 0, rj = 0, rc = 0;
    double rv = 0.0;
    for (int k = 0; k < n; ++k) {
        const int a0 = p0, a1 = p1;
        const int my_c = ac;
        const double my_v = ax;
        // ---- fetch ahead: row k + 1 of A
        const int p3 = k + 3 <= n ? arp[k + 3] : p2;
        ac = -1;
        if (k + 1 < n && p1 + lane < p2) { ac = aci[p1 + lane]; ax = av[p1 + lane]; }
        p0 = p1; p1 = p2; p2 = p3;
        // ---- column k of A (= row k of the symmetric matrix, entries at or right of the diagonal)
        if (lane == 0) { s_pk = 0; s_diag = __builtin_nan(""); s_fwd = 0; }
        wave_sync();
        if (my_c == k) s_diag = my_v;
        else if (my_c > k) {
            hval[slot_of(my_c)] = my_v;
            atomicAdd(&s_pk, 1);
        }
        for (int q = a0 + 64 + lane; q < a1; q += 64) {        // (a row of A with more than 64 entries)
            const int c = aci[q];
            const double v = av[q];
            if (c == k) s_diag = v;
            else if (c > k) {
                hval[slot_of(c)] = v;
                atomicAdd(&s_pk, 1);
            }
        }
        // the dependencies j (ascending: columns finish in order) with their L_kj, and where their entries below row k lie
        const int d_at = rc & 255, d_len = lane < m ? (rc >> 8) - d_at : 0;
        if (lane < m) { dj[lane] = rj; dv[lane] = rv; dst[lane] = rj * kIctCap + d_at; }
        {   // exclusive scan of the lengths over the lanes
            int x = d_len;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(x, off);
                if (lane >= off) x += y;
            }
            if (lane < m) doff[lane + 1] = x;
            if (lane == 0) doff[0] = 0;
        }
        wave_sync();
        double dg = s_diag;
        if (!(dg == dg)) {                                   // no diagonal entry
            if (lane == 0) { status[0] = ICHOLT_NODIAG; status[1] = k; }
            return;
        }
        for (int d = 0; d < m; ++d) dg = dg - dv[d] * dv[d];
        const int total = doff[m];
        for (int base = 0; base < total; base += 64) {
            const int q = base + lane;
            int d = -1, i = -1;
            double prod = 0.0;
            if (q < total) {
                d = 0;                                        // the dependency whose entries hold pair q
                for (int o = 1; o < m; ++o) d += doff[o] <= q ? 1 : 0;
                const int at = dst[d] + (q - doff[d]);
                i = ld_i(crow + at);
                prod = dv[d] * ld_d(cval + at);
            }
            // updates of one candidate must come in ascending j: one dependency at a time, its entries side by side
            const int d_lo = __shfl(d, 0);
            const int last = (total - base < 64 ? total - base : 64) - 1;
            const int d_hi = __shfl(d, last);
            for (int dd = d_lo; dd <= d_hi; ++dd) {
                if (d == dd) {
                    const int sl = slot_of(i);
                    hval[sl] = hval[sl] - prod;
                }
                wave_sync();
            }
        }
        wave_sync();
        const int nused = s_nused, pk = s_pk + add_fill;
        if (s_err || nused > kIctCand || pk > kIctCap) {
            if (lane == 0) { status[0] = s_err ? s_err : (nused > kIctCand ? ICHOLT_CAND : ICHOLT_COLCAP); status[1] = k; }
            return;
        }
        if (!(dg > 0.0)) {
            if (lane == 0) { status[0] = ICHOLT_PIVOT; status[1] = k; }
            return;
        }
        const double dk = sqrt(dg);
        // ---- the candidates, sorted by row (rank counting), the table cleared on the way
        for (int c = lane; c < nused; c += 64) {
            const int sl = used[c];
            ci[c] = hkey[sl];
            cv[c] = hval[sl];
            hkey[sl] = -1;
        }
        if (lane == 0) s_nused = 0;
        wave_sync();
        for (int c = lane; c < nused; c += 64) {
            const int me = ci[c];
            int rank = 0;
            for (int o = 0; o < nused; ++o) rank += ci[o] < me ? 1 : 0;
            si[rank] = me;
            sv[rank] = cv[c];
        }
        wave_sync();
        double ss = 0.0;
        for (int c = 0; c < nused; ++c) ss = ss + sv[c] * sv[c];
        const double bound = tau * sqrt(ss);
        int nk = 0;
        for (int c = 0; c < nused; ++c) nk += !(fabs(sv[c]) < bound) ? 1 : 0;
        for (int c = lane; c < nused; c += 64) {
            int kp = !(fabs(sv[c]) < bound) ? 1 : 0;
            if (kp && nk > pk) {                              // the pk largest: how many kept candidates come before this one
                const double mine = fabs(sv[c]);
                int better = 0;
                for (int o = 0; o < nused; ++o) {
                    const double other = fabs(sv[o]);
                    if (!(other < bound) && (other > mine || (other == mine && o < c))) ++better;
                }
                kp = better < pk ? 1 : 0;
            }
            keep[c] = kp;
        }
        wave_sync();
        int nkept = 0;
        for (int c = 0; c < nused; ++c) nkept += keep[c];
        // ---- fetch ahead: row k + 1's list as the columns before k left it (every slot; the count says which hold entries)
        int nm = 0, nrj = 0, nrc = 0;
        double nrv = 0.0;
        if (k + 1 < n) {
            nm = ld_i(rcnt + k + 1);
            nrj = ld_i(rcol + (size_t)(k + 1) * kIctCap + lane);
            nrc = ld_i(rcc + (size_t)(k + 1) * kIctCap + lane);
            nrv = ld_d(rval + (size_t)(k + 1) * kIctCap + lane);
        }
        // ---- column k of L, and its entries appended to their rows' lists
        bool overflow = false;
        for (int c = lane; c < nused; c += 64) {
            if (!keep[c]) continue;
            int pos = 0;
            for (int o = 0; o < c; ++o) pos += keep[o];
            const int i = si[c];
            const double l = sv[c] / dk;
            st_i(crow + (size_t)k * kIctCap + pos, i);
            st_d(cval + (size_t)k * kIctCap + pos, l);
            const int cur = ld_i(rcnt + i);
            if (cur >= kIctCap) { overflow = true; continue; }
            const int link = (pos + 1) + 256 * nkept;
            st_i(rcol + (size_t)i * kIctCap + cur, k);
            st_i(rcc + (size_t)i * kIctCap + cur, link);
            st_d(rval + (size_t)i * kIctCap + cur, l);
            st_i(rcnt + i, cur + 1);
            if (i == k + 1) { s_fwd = link; s_fwd_l = l; }    // (pos = 0: link > 0)
        }
        if (lane == 0) diag[k] = dk;
        if (__ballot(overflow)) {
            if (lane == 0) { status[0] = ICHOLT_ROWCAP; status[1] = k; }
            return;
        }
        wave_sync();
        // the next row's list: what was fetched, plus the entry this column added to it
        m = nm; rj = nrj; rc = nrc; rv = nrv;
        if (s_fwd) {
            if (lane == m) { rj = k; rc = s_fwd; rv = s_fwd_l; }
            m += 1;
        }
        wave_sync();
    }
}

// L as CSR: row i = its list (columns ascending) followed by the diagonal
__global__ __launch_bounds__(kBlock) void k_icholt_count(int n, const int *__restrict__ rcnt, int32_t *__restrict__ cnt) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) cnt[i] = rcnt[i] + 1;
    if (i == n) cnt[i] = 0;
}
__global__ __launch_bounds__(kBlock) void k_icholt_emit(int n, const int *__restrict__ rcnt, const int *__restrict__ rcol,
                                                        const double *__restrict__ rval, const double *__restrict__ diag,
                                                        const int32_t *__restrict__ rp, int32_t *__restrict__ col,
                                                        double *__restrict__ val) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int m = rcnt[i], at = rp[i];
    for (int q = 0; q < m; ++q) {
        col[at + q] = rcol[(size_t)i * kIctCap + q];
        val[at + q] = rval[(size_t)i * kIctCap + q];
    }
    col[at + m] = i;
    val[at + m] = diag[i];
}

}  // namespace dpcg

// Factor `A` (the caller's matrix: a symmetric CSR with sorted columns) into `Lf` (owned CSR, diagonal last).  Leaves Lf empty on failure.
int icholt_factor(const CsrDev &A, int add_fill_in, double threshold, CsrDev &Lf, hipStream_t s) {
    const int64_t n = A.n;
    if (n > 0x7fffffff / kIctCap) return invalid("dpcg_set_precond_icholt: too many rows for the per-row lists");
    int *rcnt = nullptr, *rcol = nullptr, *rcc = nullptr, *crow = nullptr, *status = nullptr;
    int32_t *cnt = nullptr;
    double *rval = nullptr, *cval = nullptr, *diag = nullptr;
    Lf = CsrDev{};
    Lf.n = n;
    Lf.owned = true;
    auto cleanup = [&](int st) {
        dev_free(rcnt); dev_free(rcol); dev_free(rcc); dev_free(crow); dev_free(status); dev_free(cnt);
        dev_free(rval); dev_free(cval); dev_free(diag);
        if (st < 0) free_csr(Lf);
        return st;
    };
    int st = DPCG_OK;
    const int64_t wide = n * kIctCap;
    if ((st = dev_alloc(&rcnt, n)) < 0 || (st = dev_alloc(&rcol, wide)) < 0 || (st = dev_alloc(&rcc, wide)) < 0 ||
        (st = dev_alloc(&rval, wide)) < 0 || (st = dev_alloc(&crow, wide)) < 0 ||
        (st = dev_alloc(&cval, wide)) < 0 || (st = dev_alloc(&diag, n)) < 0 || (st = dev_alloc(&status, 2)) < 0 ||
        (st = dev_alloc(&cnt, n + 1)) < 0 || (st = dev_alloc(&Lf.rowptr, n + 1)) < 0)
        return cleanup(st);
    hipError_t e = hipMemsetAsync(rcnt, 0, (size_t)n * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(status, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return cleanup(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
    hipLaunchKernelGGL(k_icholt, dim3(1), dim3(64), 0, s, (int)n, A.rowptr, A.col, A.val, add_fill_in, threshold, rcnt, rcol, rcc,
                       rval, crow, cval, diag, status);
    int h_status[2] = {0, 0};
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_status, status, sizeof(h_status), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(hip_fail(e, "icholt", __FILE__, __LINE__));
    if (h_status[0] != ICHOLT_OK) {
        const std::string at = " at column " + std::to_string(h_status[1]);
        switch (h_status[0]) {
        case ICHOLT_PIVOT: set_error("icholt: non-positive pivot" + at); return cleanup(DPCG_ERR_PIVOT);
        case ICHOLT_NODIAG: set_error("icholt: missing diagonal entry" + at); return cleanup(DPCG_ERR_PIVOT);
        case ICHOLT_CAND: set_error("icholt: more than 256 candidates" + at); return cleanup(DPCG_ERR_INVALID);
        case ICHOLT_ROWCAP: set_error("icholt: a row of L would keep more than 64 entries (reached" + at + ")"); return cleanup(DPCG_ERR_INVALID);
        default: set_error("icholt: nnz + add_fill_in exceeds 64 entries" + at); return cleanup(DPCG_ERR_INVALID);
        }
    }
    const unsigned grid = (unsigned)((n + 1 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_icholt_count, dim3(grid), dim3(kBlock), 0, s, (int)n, rcnt, cnt);
    if ((st = exclusive_scan_i32(cnt, Lf.rowptr, n + 1, s)) < 0) return cleanup(st);
    int32_t lnnz = 0;
    e = hipMemcpyAsync(&lnnz, Lf.rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(hip_fail(e, "icholt: row pointers", __FILE__, __LINE__));
    Lf.nnz = lnnz;
    if ((st = dev_alloc(&Lf.col, lnnz)) < 0 || (st = dev_alloc(&Lf.val, lnnz)) < 0) return cleanup(st);
    hipLaunchKernelGGL(k_icholt_emit, dim3(grid), dim3(kBlock), 0, s, (int)n, rcnt, rcol, rval, diag, Lf.rowptr, Lf.col, Lf.val);
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return cleanup(hip_fail(e, "icholt: emit", __FILE__, __LINE__));
    return cleanup(DPCG_OK);
}
