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
// CPU restatement bit for bit.  One dependent memory round trip per column (see k_icholt); times by size: tools/icholt_probe.py,
// profiles/r04_icholt_probe.txt.  It is the setup of a technique the reference runs on ~2K-row systems: those -- every system whose
// factor fits one CU's LDS -- go through k_icholt_lds below, a pipeline of four waves over an LDS-resident factor (round 5:
// 2.4 ms instead of 4.8 at 2.4K rows, profiles/r05_icholt_probe.txt; what is left is one wave's instruction stream per column);
// factors beyond the LDS, up to 8192 rows, through the same pipeline over a workspace in memory (5.5K rows: 8.7 ms instead of 12.4).
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

// value of lane `l` (the same for every lane of the wave: a loop counter, a first-lane read) -- the wave's loops over a handful of
// dependencies / candidates read each other's registers this way instead of going through LDS (64+ cycles a dependent read)
__device__ __forceinline__ int lane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int first_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// rcnt[i]: kept entries of row i so far; row i's list at [i * kIctCap ..]: column j, L_ij, and rcc = (position after L_ij in column
// j's list) + 256 * (kept entries of column j), i.e. where the entries of column j below row i start and end; column j's kept
// (row, value) at [j * kIctCap ..], rows ascending; diag[k] = L_kk.  status[0] = code, status[1] = column.
//
// ONE dependent round trip per column: the entries of its dependencies.  Everything else is fetched ahead or written behind --
// the next row of A a column ahead; row k + 1's list while column k is computed (the entry column k itself adds to it is handed
// over in registers); the counters of the rows column k appends to are requested when the column is done, and the appends
// themselves are issued in column k + 1, after ITS dependencies' entries have been requested.
// Row k's list lives in registers (lane d: dependency d) and so do the candidates (lane c: candidate c): a pair (row, product)
// finds its candidate by a compare across the wave.  A column with many pairs, or more than 64 candidates, goes through the LDS
// hash table (keyed by row) and, beyond 64 candidates, through the LDS selection; `use_regs` = 0 forces that path (tests).
__global__ __launch_bounds__(64) void k_icholt(int n, const int32_t *__restrict__ arp, const int32_t *__restrict__ aci,
                                               const double *__restrict__ av, int add_fill, double tau, int *rcnt, int *rcol,
                                               int *rcc, double *rval, int *crow, double *cval, double *diag, int *status,
                                               int use_regs) {
    constexpr int kNone = 0x7fffffff;
    __shared__ int hkey[kIctHash];
    __shared__ double hval[kIctHash];
    __shared__ int used[kIctCand];
    __shared__ int ci[kIctCand], si[kIctCand], keep[kIctCand];
    __shared__ double cv[kIctCand], sv[kIctCand];
    __shared__ int s_nused, s_err, s_pk, s_fwd;
    __shared__ double s_diag, s_fwd_l;
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;    // the lanes before this one
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
    // row k's list of L, fetched while column k - 1 was computed (row 0: empty)
    int m = 0, rj = 0, rc = 0;
    double rv = 0.0;
    // the appends of column k - 1, still to be issued: this lane's kept entry (row pa_row, value pa_l, link pa_link) goes to slot
    // pa_cur (requested at the end of column k - 1) of its row's list
    bool pa = false;
    int pa_row = 0, pa_link = 0, pa_cur = 0;
    double pa_l = 0.0;
    bool overflow = false;
    auto issue_appends = [&](int col) {
        if (pa) {
            if (pa_cur >= kIctCap) overflow = true;
            else {
                st_i(rcol + (size_t)pa_row * kIctCap + pa_cur, col);
                st_i(rcc + (size_t)pa_row * kIctCap + pa_cur, pa_link);
                st_d(rval + (size_t)pa_row * kIctCap + pa_cur, pa_l);
                st_i(rcnt + pa_row, pa_cur + 1);
            }
        }
        pa = false;
    };
    for (int k = 0; k < n; ++k) {
        const int a0 = p0, a1 = p1;
        const int my_c = ac;
        const double my_v = ax;
        // ---- fetch ahead: row k + 1 of A
        const int p3 = k + 3 <= n ? arp[k + 3] : p2;
        ac = -1;
        if (k + 1 < n && p1 + lane < p2) { ac = aci[p1 + lane]; ax = av[p1 + lane]; }
        p0 = p1; p1 = p2; p2 = p3;
        if (lane == 0) { s_pk = 0; s_diag = __builtin_nan(""); s_fwd = 0; }
        wave_sync();
        // ---- the dependencies j (ascending: columns finish in order; lane d holds dependency d): L_kj, and where the entries of
        // column j below row k lie
        const int d_at = rc & 255, d_len = lane < m ? (rc >> 8) - d_at : 0, d_st = rj * kIctCap + d_at;
        int d_off = 0, total = 0;                              // exclusive scan of the lengths over the lanes
        for (int d = 0; d < m; ++d) {
            const int len = lane_i(d_len, d);
            if (lane > d) d_off += len;
            total += len;
        }
        // pair q of the flattened (dependency, entry) list: its dependency, address and L_kj
        auto pair_of = [&](int q, int &d, int &at, double &lkj) {
            d = 0;
            at = lane_i(d_st, 0) + q;
            lkj = lane_d(rv, 0);
            for (int o = 1; o < m; ++o) {
                const int off = lane_i(d_off, o), st = lane_i(d_st, o);
                const double l = lane_d(rv, o);
                if (off <= q) { d = o; at = st + (q - off); lkj = l; }
            }
        };
        int d0 = -1, i0 = -1;
        double prod0 = 0.0;
        if (total > 0) {                                       // the first 64 pairs: requested before anything else
            int at;
            pair_of(lane, d0, at, prod0);
            if (lane < total) {
                i0 = ld_i(crow + at);
                prod0 = prod0 * ld_d(cval + at);
            } else {
                d0 = -1;
            }
        }
        // ---- written behind: the appends of column k - 1; fetched ahead: row k + 1's list as the columns before k leave it (every
        // slot; the count says which hold entries)
        issue_appends(k - 1);
        if (__ballot(overflow)) {
            if (lane == 0) { status[0] = ICHOLT_ROWCAP; status[1] = k - 1; }
            return;
        }
        int nm = 0, nrj = 0, nrc = 0;
        double nrv = 0.0;
        if (k + 1 < n) {
            nm = ld_i(rcnt + k + 1);
            nrj = ld_i(rcol + (size_t)(k + 1) * kIctCap + lane);
            nrc = ld_i(rcc + (size_t)(k + 1) * kIctCap + lane);
            nrv = ld_d(rval + (size_t)(k + 1) * kIctCap + lane);
        }
        // ---- column k of A (= row k of the symmetric matrix, entries at or right of the diagonal), its diagonal reduced
        const int alen = a1 - a0;
        const unsigned long long at_diag = __ballot(my_c == k);
        int pk = __popcll(__ballot(my_c > k));
        double dg = at_diag ? lane_d(my_v, __ffsll((long long)at_diag) - 1) : __builtin_nan("");
        const bool regs = use_regs && alen + total <= 64;
        int me = kNone;                                        // this lane's candidate (regs): row, value; nl lanes in use
        double val = 0.0;
        int nl = 0;
        if (regs) {
            if (my_c > k) { me = my_c; val = my_v; }
            nl = alen;
        } else {
            if (my_c > k) hval[slot_of(my_c)] = my_v;
            if (alen > 64) {                                   // (a row of A with more than 64 entries)
                for (int q = a0 + 64 + lane; q < a1; q += 64) {
                    const int c = aci[q];
                    const double v = av[q];
                    if (c == k) s_diag = v;
                    else if (c > k) {
                        hval[slot_of(c)] = v;
                        atomicAdd(&s_pk, 1);
                    }
                }
                wave_sync();
                pk += first_i(s_pk);
                const double late = s_diag;
                if (late == late) dg = late;
            }
        }
        pk += add_fill;
        if (!(dg == dg)) {                                   // no diagonal entry
            if (lane == 0) { status[0] = ICHOLT_NODIAG; status[1] = k; }
            return;
        }
        for (int d = 0; d < m; ++d) {
            const double l = lane_d(rv, d);
            dg = dg - l * l;
        }
        // ---- the updates: of one candidate in ascending j (one product, one subtraction at a time)
        if (regs) {
            for (int q = 0; q < total; ++q) {
                const int ip = lane_i(i0, q);
                const double pp = lane_d(prod0, q);
                if (__ballot(me == ip)) {
                    if (me == ip) val = val - pp;
                } else {
                    if (lane == nl) { me = ip; val = 0.0 - pp; }
                    ++nl;
                }
            }
        } else {
            for (int base = 0; base < total; base += 64) {
                int d = d0, i = i0;
                double prod = prod0;
                if (base > 0) {
                    int at;
                    pair_of(base + lane, d, at, prod);
                    if (base + lane < total) {
                        i = ld_i(crow + at);
                        prod = prod * ld_d(cval + at);
                    } else {
                        d = -1;
                    }
                }
                const int d_lo = lane_i(d, 0);
                const int last = (total - base < 64 ? total - base : 64) - 1;
                const int d_hi = lane_i(d, last);
                for (int dd = d_lo; dd <= d_hi; ++dd) {        // one dependency at a time, its entries side by side
                    if (d == dd) {
                        const int sl = slot_of(i);
                        hval[sl] = hval[sl] - prod;
                    }
                    wave_sync();
                }
            }
            wave_sync();
            nl = first_i(s_nused);
            if (s_err || nl > kIctCand) {
                if (lane == 0) { status[0] = ICHOLT_CAND; status[1] = k; }
                return;
            }
            if (nl <= 64) {                                    // out of the table into the registers (the table cleared on the way)
                if (lane < nl) {
                    const int sl = used[lane];
                    me = hkey[sl];
                    val = hval[sl];
                    hkey[sl] = -1;
                }
                if (lane == 0) s_nused = 0;
            }
        }
        if (pk > kIctCap) {
            if (lane == 0) { status[0] = ICHOLT_COLCAP; status[1] = k; }
            return;
        }
        if (!(dg > 0.0)) {
            if (lane == 0) { status[0] = ICHOLT_PIVOT; status[1] = k; }
            return;
        }
        const double dk = sqrt(dg);
        if (nl <= 64) {
            // ---- the candidates sorted by row (rank counting), norm, threshold, the pk largest
            int rank = 0;
            for (int o = 0; o < nl; ++o) rank += lane_i(me, o) < me ? 1 : 0;
            const int ncand = __popcll(__ballot(me != kNone));
            if (me != kNone) { si[rank] = me; sv[rank] = val; }
            wave_sync();
            const int row = lane < ncand ? si[lane] : kNone;
            const double w = lane < ncand ? sv[lane] : 0.0;
            const double sq = w * w, aw = fabs(w);
            double ss = 0.0;
            for (int c = 0; c < ncand; ++c) ss = ss + lane_d(sq, c);
            const double bound = tau * sqrt(ss);
            const bool pass = lane < ncand && !(aw < bound);
            const unsigned long long passed = __ballot(pass);
            bool kp = pass;
            if (__popcll(passed) > pk) {                      // how many kept candidates come before this one (ties: smaller row)
                int better = 0;
                for (int o = 0; o < ncand; ++o) {
                    if (!((passed >> o) & 1ull)) continue;
                    const double other = lane_d(aw, o);
                    better += (other > aw || (other == aw && o < lane)) ? 1 : 0;
                }
                kp = pass && better < pk;
            }
            const unsigned long long kept = __ballot(kp);
            const int nkept = __popcll(kept), pos = __popcll(kept & below);
            // ---- column k of L; its entries' rows' counters requested (the appends follow in the next column)
            if (kp) {
                const double l = w / dk;
                st_i(crow + (size_t)k * kIctCap + pos, row);
                st_d(cval + (size_t)k * kIctCap + pos, l);
                pa = true;
                pa_row = row;
                pa_l = l;
                pa_link = (pos + 1) + 256 * nkept;
                pa_cur = ld_i(rcnt + row);
                if (row == k + 1) { s_fwd = pa_link; s_fwd_l = l; }    // (link > 0)
            }
        } else {
            // ---- the same through LDS for a column of up to kIctCand candidates (its appends issued at once)
            const int nused = nl;
            for (int c = lane; c < nused; c += 64) {
                const int sl = used[c];
                ci[c] = hkey[sl];
                cv[c] = hval[sl];
                hkey[sl] = -1;
            }
            if (lane == 0) s_nused = 0;
            wave_sync();
            for (int c = lane; c < nused; c += 64) {
                const int mine = ci[c];
                int rank = 0;
                for (int o = 0; o < nused; ++o) rank += ci[o] < mine ? 1 : 0;
                si[rank] = mine;
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
                if (kp && nk > pk) {
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
                if (i == k + 1) { s_fwd = link; s_fwd_l = l; }
            }
            if (__ballot(overflow)) {
                if (lane == 0) { status[0] = ICHOLT_ROWCAP; status[1] = k; }
                return;
            }
        }
        if (lane == 0) diag[k] = dk;
        wave_sync();
        // the next row's list: what was fetched, plus the entry this column adds to it
        m = first_i(nm); rj = nrj; rc = nrc; rv = nrv;
        const int fwd = first_i(s_fwd);
        if (fwd) {
            if (lane == m) { rj = k; rc = fwd; rv = s_fwd_l; }
            m += 1;
        }
        wave_sync();
    }
    issue_appends(n - 1);
    if (__ballot(overflow)) {
        if (lane == 0) { status[0] = ICHOLT_ROWCAP; status[1] = n - 1; }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same factorisation for systems whose factor fits ONE CU's LDS (the reference's sizes: ~2K rows, test.py:130-135), as a pipeline
// of W waves of one workgroup.  The algorithm stays a chain -- column k needs L_kj L_ij of its last dependency j = k - 1 -- but only
// THAT update and the selection (two square roots, a division, the sum of squares) are on the chain: the rest of a column's work (its
// row of A, its earlier dependencies, the candidates they create) is done by another wave while the columns before it finish.
//   * wave w takes the columns k = w, w + W, ...;
//   * the factor lives in an LDS pool in column order (columns finish in order, so the pool is L's strict lower triangle in CSC:
//     pval[p], pmeta[p] = row | column << 12 | (entries of the column after this one) << 24), each row's kept entries chained through
//     pnext[p] in ascending column (rhead / rtail);
//   * a wave walks its row's chain as far as it reaches -- one entry = one dependency: L_kj = pval[p], the entries of column j below
//     row k follow at p + 1 .. p + rem -- and waits at its end for either a new link or `done >= k` (every column before k final:
//     the chain is complete); then it selects, appends its column to the pool, links its entries into their rows' chains and
//     publishes done = k + 1;
//   * the LAST dependency (column k - 1, in a banded numbering) does not wait for that publication: a wave that has walked its chain
//     to the end while done = k - 1 polls the MAILBOX of column k - 1 -- the kept (row, value) pairs, written the moment they are
//     selected -- and goes on to its own selection while its predecessor is still publishing (it waits for done >= k only before
//     it appends: the ends of the chains are its predecessor's until then).
// One wave's LDS operations are performed in order and a store is performed for all its lanes before the wave's next LDS operation:
// entries are written before the links that lead to them, links before `done` -- the fences are compiler-level only.
// Anything beyond the plain case (a row of A of more than 64 entries, more than 64 candidates, a row of L of more than 64 entries,
// a breakdown) ends the kernel with ICHOLT_RETRY: the one-wave kernel above runs the factorisation again and reports what it finds.
// Every sum in the oracle's order, as above: the factor is the same, bit for bit.
constexpr int kLdsNil = 0xffff;
constexpr int kLdsMaxRows = 4096;            // 12 bits of pmeta for the row, 12 for the column
constexpr int kGmMaxRows = 8192;             // (the workspace form: 13 + 13 bits)
constexpr int kLdsMail = 8;                  // mailboxes: the column a wave has just selected, for the wave of the next column
enum { ICHOLT_RETRY = 6 };

__device__ __forceinline__ int lds_ld_i(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st_i(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_ld_u16(const uint16_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st_u16(uint16_t *p, int v) { __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// pool, chains and staged A (in LDS, or -- GM -- in a workspace in memory), and what stays in LDS either way
size_t icholt_big_bytes(int n, int pool_cap) {
    const size_t pool = (size_t)pool_cap + 64, rows = ((size_t)n + 3) & ~(size_t)3;
    return pool * 8 + rows * 8 + pool * 4 + (pool + (pool & 1)) * 2 + 5 * rows * 2;
}
size_t icholt_small_bytes(int waves) { return (size_t)kLdsMail * 64 * 8 + ((size_t)waves + kLdsMail) * 64 * 4; }
size_t icholt_lds_bytes(int n, int pool_cap, int waves) { return icholt_small_bytes(waves) + icholt_big_bytes(n, pool_cap); }

// GM = true: the same pipeline for factors beyond one CU's LDS (up to 8192 rows and 65 K kept entries: the reference's larger meshes,
// 5.5 K rows): pool, chains and the staged A live in a workspace in memory -- the waves of ONE workgroup reach it through agent-scope
// loads (a line another wave wrote may sit stale in the CU's L1) and order their stores with a release fence where the LDS form
// relies on the order of LDS operations; `done`, the mailboxes (the chain's hand-off) and the scan scratch stay in LDS.
template <bool GM, typename T>
__device__ __forceinline__ T sh_ld(const T *p) {
    if (GM) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool GM, typename T>
__device__ __forceinline__ void sh_st(T *p, T v) {
    if (GM) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// what was stored before is there for whoever sees what is stored after
template <bool GM>
__device__ __forceinline__ void sh_release() {
    if (GM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else wave_sync();
}

template <int W, bool TR = false, bool GM = false>
__global__ __launch_bounds__(W * 64) void k_icholt_lds(int n, const int32_t *__restrict__ arp, const int32_t *__restrict__ aci,
                                                       const double *__restrict__ av, int add_fill, double tau, int pool_cap,
                                                       int32_t *lrp, int32_t *lci, double *lv, int *status, double *ws) {
    constexpr int kNone = 0x7fffffff;
    constexpr int RB = GM ? 13 : 12;                                    // bits of pmeta for the row, and for the column; the rest: entries after this one
    constexpr uint32_t RM = (1u << RB) - 1u;
    extern __shared__ double smem[];
    const int pool = pool_cap + 64, rows = (n + 3) & ~3;
    double *const mb_val = smem;                                        // [kLdsMail][64]   a column's kept values ...
    int *const si_all = (int *)(mb_val + kLdsMail * 64);                // [W][64]          (the scans of the row lengths)
    int *const mb_row = si_all + W * 64;                                // [kLdsMail][64]   ... and their rows, ascending
    double *const big = GM ? ws : (double *)(mb_row + kLdsMail * 64);   // pool, chains, staged A: LDS, or the workspace
    double *const pval = big;                                           // [pool]
    double *const dgl = pval + pool;                                    // [rows]           the diagonal of L
    uint32_t *const pmeta = (uint32_t *)(dgl + rows);                   // [pool]
    uint16_t *const pnext = (uint16_t *)(pmeta + pool);                 // [pool (+1)]
    uint16_t *const rhead = pnext + pool + (pool & 1);                  // [rows]
    uint16_t *const rtail = rhead + rows;
    uint16_t *const rcnt = rtail + rows;
    uint16_t *const ub = rcnt + rows;                                   // [rows]  where column k of A is staged: sum of the count bounds before it
    uint16_t *const acnt = ub + rows;                                   // [rows]  entries of A below the diagonal in column k
    __shared__ int s_done, s_abort, mb_tag[kLdsMail], mb_info[kLdsMail];     // mailbox k & 7: tag = k + 1 once column k's entries are in it, info = kept | pool position << 8
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int i = threadIdx.x; i < rows; i += W * 64) { sh_st<GM>(rhead + i, (uint16_t)kLdsNil); sh_st<GM>(rtail + i, (uint16_t)kLdsNil); sh_st<GM>(rcnt + i, (uint16_t)0); }
    if (threadIdx.x == 0) { s_done = 0; s_abort = 0; }
    if (threadIdx.x < kLdsMail) mb_tag[threadIdx.x] = 0;
    __syncthreads();
    auto give_up = [&](int col) {
        if (lane == 0) {
            if (atomicCAS(&s_abort, 0, 1) == 0) { status[0] = ICHOLT_RETRY; status[1] = col; }
        }
    };
    // ---- A into the pool, once: column k's entries below the diagonal (= row k right of it) at ub[k] = the sum of the count bounds
    // p_j = nnz(A[j+1:, j]) + add_fill of the columns before it.  Column j's kept entries end at or before ub[j + 1] (it keeps at most
    // p_j, and starts at or before ub[j]), so a staged column is still there when its wave reads it -- at the START of its work on
    // that column -- and the loop below touches no memory outside the LDS.
    {
        constexpr int T = W * 64;
        bool bad = false;
        const int per = (n + T - 1) / T, r_lo = min(n, (int)threadIdx.x * per), r_hi = min(n, r_lo + per);
        int mine = 0;
        for (int k = r_lo; k < r_hi; ++k) {
            int cu = 0;
            double dv = 0.0;
            bool has = false;
            for (int q = arp[k]; q < arp[k + 1]; ++q) {
                const int c = aci[q];
                if (c > k) ++cu;
                else if (c == k) { dv = av[q]; has = true; }
            }
            if (!has || cu + add_fill > kIctCap) bad = true;
            sh_st<GM>(acnt + k, (uint16_t)cu);
            sh_st<GM>(dgl + k, dv);
            mine += cu + add_fill;
        }
        si_all[threadIdx.x] = mine;
        if (bad) atomicExch(&s_abort, 2);
        __syncthreads();
        int at = 0;
        for (int u = 0; u < (int)threadIdx.x; ++u) at += si_all[u];
        if (threadIdx.x == T - 1 && at + mine > pool_cap) atomicExch(&s_abort, 2);
        __syncthreads();
        if (s_abort) {                    // (a missing diagonal, a count bound over the cap, a pattern that is not symmetric: the one-wave kernel reports it)
            if (threadIdx.x == 0) { status[0] = ICHOLT_RETRY; status[1] = 0; }
            return;
        }
        for (int k = r_lo; k < r_hi; ++k) {
            sh_st<GM>(ub + k, (uint16_t)at);
            int j = at;
            for (int q = arp[k]; q < arp[k + 1]; ++q) {
                const int c = aci[q];
                if (c > k) { sh_st<GM>(pmeta + j, (uint32_t)c); sh_st<GM>(pval + j, av[q]); ++j; }
            }
            at = j + add_fill;
        }
        if (GM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // (the staged columns are in memory before anybody reads them)
        __syncthreads();
    }
    const long long t_start = wall_clock64();
    bool alive = true;
    long long tr[6] = {0, 0, 0, 0, 0, 0}, tc = 0;      // TR: cycles in [0] row of A, [1] waiting, [2] dependencies, [3] selection, [4] publication
    auto lap = [&](int which) {
        if (TR) {
            const long long now = clock64();
            tr[which] += now - tc;
            tc = now;
        }
    };
    if (TR) tc = clock64();
    for (int k = w; k < n && alive; k += W) {
        // ---- column k of A out of the pool: candidates below the diagonal (lane c: candidate c, rows ascending), the diagonal
        const int u0 = (int)sh_ld<GM>(ub + k), cu = (int)sh_ld<GM>(acnt + k);
        const int pk = cu + add_fill;
        int me = lane < cu ? (int)sh_ld<GM>(pmeta + u0 + lane) : kNone, nl = cu;
        double val = lane < cu ? sh_ld<GM>(pval + u0 + lane) : 0.0;
        double dg = sh_ld<GM>(dgl + k);
        // ---- the dependencies, in ascending j as the chain of row k holds them
        int prev = kLdsNil, spins = 0, base = 0;
        bool by_mail = false;           // the last dependency came through the mailbox: column k - 1 may still be publishing
        // one (row, product) pair of a dependency: the candidate of that row, or a new one (no branch: a taken branch costs more)
        auto update = [&](int ip, double pp) {
            const bool hit = me == ip;
            const bool fresh = __ballot(hit) == 0ull;
            const bool take = fresh && lane == nl;
            const double upd = (take ? 0.0 : val) - pp;
            val = (hit || take) ? upd : val;
            me = take ? ip : me;
            nl += fresh ? 1 : 0;
        };
        lap(0);
        for (;;) {
            const int d = first_i(lds_ld_i(&s_done));            // columns final | pool entries in use << 14
            wave_sync();
            const int p = first_i((int)(prev == kLdsNil ? sh_ld<GM>(rhead + k) : sh_ld<GM>(pnext + prev)));
            if (p == kLdsNil) {
                lap(1);
                const int nd = d & 0x3fff;
                if (nd >= k) { base = d >> 14; break; }
                if (nd == k - 1) {
                    // every column before k - 1 is final and its links are walked: only column k - 1 can still add a dependency, and
                    // it hands its entries over the moment it has selected them -- before it publishes them.  ONE word is polled
                    // (the waves that wait share the LDS with the wave that works)
                    const int slot = (k - 1) & (kLdsMail - 1);
                    for (;;) {
                        if (first_i(lds_ld_i(mb_tag + slot)) == k) break;
                        if ((++spins & 63) == 0 && (lds_ld_i(&s_abort) || wall_clock64() - t_start > 100000000ll)) { alive = false; break; }
                    }
                    if (!alive) { give_up(k); break; }
                    wave_sync();
                    const int info = first_i(lds_ld_i(mb_info + slot));
                    const int erow = mb_row[slot * 64 + lane];
                    const double ev = mb_val[slot * 64 + lane];
                    const int nk = info & 0xff;
                    const unsigned long long hit = __ballot(lane < nk && erow == k);
                    if (hit) {
                        const int hl = __ffsll((long long)hit) - 1;
                        const double lkj = lane_d(ev, hl);
                        dg = dg - lkj * lkj;
                        const double prod = lkj * ev;
                        for (int q = hl + 1; q < nk; ++q) update(lane_i(erow, q), lane_d(prod, q));
                    }
                    base = (info >> 8) + nk;
                    by_mail = true;
                    lap(2);
                    break;
                }
                if (lds_ld_i(&s_abort)) { alive = false; break; }
                if ((++spins & 1023) == 0 && wall_clock64() - t_start > 100000000ll) { give_up(k); alive = false; break; }   // 1 s
                __builtin_amdgcn_s_sleep(2);          // (two or more columns ahead of the chain: no hurry, and the LDS is the others')
                continue;
            }
            wave_sync();
            const uint32_t em = sh_ld<GM>(pmeta + p + lane);
            const double ev = sh_ld<GM>(pval + p + lane);
            const double lkj = lane_d(ev, 0);
            const unsigned em0 = (unsigned)lane_i((int)em, 0);
            const int rem = (int)(em0 >> (2 * RB));
            dg = dg - lkj * lkj;
            const int erow = (int)(em & RM);
            const double prod = lkj * ev;
            for (int q = 1; q <= rem; ++q) update(lane_i(erow, q), lane_d(prod, q));
            if (nl > 64) { give_up(k); alive = false; break; }
            prev = p;
            lap(2);
            // the link of column k - 1: no column before k can add another, and the pool ends behind that column
            if ((int)((em0 >> RB) & RM) == k - 1) { base = p + rem + 1; break; }
        }
        if (nl > 64) { give_up(k); break; }
        if (!alive) break;
        if (!(dg > 0.0)) { give_up(k); break; }
        // ---- the candidates sorted by row (rank counting; lane `rank` receives), norm, threshold, the pk largest (as k_icholt)
        // (loops of a fixed length for the usual sizes: lanes beyond the candidates hold kNone / 0 and change nothing)
        int rank = 0;
        if (nl <= 8) {
#pragma unroll
            for (int o = 0; o < 8; ++o) rank += lane_i(me, o) < me ? 1 : 0;
        } else if (nl <= 16) {
#pragma unroll
            for (int o = 0; o < 16; ++o) rank += lane_i(me, o) < me ? 1 : 0;
        } else {
            for (int o = 0; o < nl; ++o) rank += lane_i(me, o) < me ? 1 : 0;
        }
        const int ncand = __popcll(__ballot(me != kNone));
        const long long vb = __double_as_longlong(val);
        const int row_s = __builtin_amdgcn_ds_permute(rank << 2, me);
        const unsigned lo_s = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)vb);
        const unsigned hi_s = (unsigned)__builtin_amdgcn_ds_permute(rank << 2, (int)(vb >> 32));
        const int row = lane < ncand ? row_s : kNone;
        const double wv = lane < ncand ? __longlong_as_double((long long)(((unsigned long long)hi_s << 32) | lo_s)) : 0.0;
        const double sq = wv * wv, aw = fabs(wv);
        double ss = 0.0;
        if (ncand <= 8) {                                          // (+ 0.0 behind the last candidate: the same sum)
#pragma unroll
            for (int c = 0; c < 8; ++c) ss = ss + lane_d(sq, c);
        } else {
            for (int c = 0; c < ncand; ++c) ss = ss + lane_d(sq, c);
        }
        const double both = sqrt(lane == 0 ? ss : dg);                 // lane 0: the norm; the others: the pivot
        const double dk = lane_d(both, 1);
        const double bound = tau * lane_d(both, 0);
        const bool pass = lane < ncand && !(aw < bound);
        const unsigned long long passed = __ballot(pass);
        bool kp = pass;
        if (__popcll(passed) > pk) {
            int better = 0;
            const double mag = pass ? aw : -1.0;                   // (a candidate below the threshold beats nobody)
            if (ncand <= 8) {
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const double other = lane_d(mag, o);
                    better += (other > aw || (other == aw && o < lane)) ? 1 : 0;
                }
            } else {
                for (int o = 0; o < ncand; ++o) {
                    const double other = lane_d(mag, o);
                    better += (other > aw || (other == aw && o < lane)) ? 1 : 0;
                }
            }
            kp = pass && better < pk;
        }
        const unsigned long long kept = __ballot(kp);
        const int nkept = __popcll(kept), pos = __popcll(kept & below);
        lap(3);
        // ---- the column is handed to the wave of column k + 1 (its mailbox) the moment it is selected ...
        if (base + nkept > pool_cap) { give_up(k); break; }
        const double lq = wv / dk;
        {
            const int slot = k & (kLdsMail - 1);
            if (kp) { mb_row[slot * 64 + pos] = row; mb_val[slot * 64 + pos] = lq; }
            if (lane == 0) lds_st_i(mb_info + slot, nkept | base << 8);
            wave_sync();
            if (lane == 0) lds_st_i(mb_tag + slot, k + 1);
        }
        // ---- ... then joins the pool, its entries their rows' chains, done = k + 1.  A column that came through the mailbox waits for
        // its predecessor to have published (the ends of the chains it appends to are that column's; usually long done)
        if (by_mail) {
            int turns = 0;
            while ((first_i(lds_ld_i(&s_done)) & 0x3fff) < k) {
                if (lds_ld_i(&s_abort) || ((++turns & 1023) == 0 && wall_clock64() - t_start > 100000000ll)) { alive = false; break; }
            }
            if (!alive) { give_up(k); break; }
        }
        wave_sync();
        const int t_tail = kp ? (int)sh_ld<GM>(rtail + row) : kLdsNil, t_cnt = kp ? (int)sh_ld<GM>(rcnt + row) : 0;
        if (__ballot(kp && t_cnt >= kIctCap)) { give_up(k); break; }
        const int at = base + pos;
        // (in this order: whoever sees a link finds the entries behind it and the new ends of the chains; `done` only ever grows)
        if (kp) {
            sh_st<GM>(pval + at, lq);
            sh_st<GM>(pmeta + at, (uint32_t)row | (uint32_t)k << RB | (uint32_t)(nkept - 1 - pos) << (2 * RB));
            sh_st<GM>(pnext + at, (uint16_t)kLdsNil);
            sh_st<GM>(rtail + row, (uint16_t)at);
            sh_st<GM>(rcnt + row, (uint16_t)(t_cnt + 1));
        }
        if (lane == 0) sh_st<GM>(dgl + k, dk);
        sh_release<GM>();
        if (kp) {
            if (t_tail == kLdsNil) sh_st<GM>(rhead + row, (uint16_t)at);
            else sh_st<GM>(pnext + t_tail, (uint16_t)at);
        }
        sh_release<GM>();
        if (lane == 0) atomicMax(&s_done, (k + 1) | (base + nkept) << 14);
        wave_sync();
        lap(4);
    }
    if (TR && lane == 0)
        for (int q = 0; q < 5; ++q) status[8 + w * 8 + q] = (int)(tr[q] >> 4);
    if (GM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (s_abort) return;
    // ---- L as CSR (row i = its chain, columns ascending, then the diagonal): row pointers by a scan over the threads' runs of rows
    constexpr int T = W * 64;
    const int per = (n + T - 1) / T, r_lo = min(n, (int)threadIdx.x * per), r_hi = min(n, r_lo + per);
    int mine = 0;
    for (int i = r_lo; i < r_hi; ++i) mine += (int)sh_ld<GM>(rcnt + i) + 1;
    int *const sums = si_all;                                             // [T]
    sums[threadIdx.x] = mine;
    __syncthreads();
    int at = 0;
    for (int u = 0; u < (int)threadIdx.x; ++u) at += sums[u];
    if (threadIdx.x == T - 1) { lrp[n] = at + mine; status[2] = at + mine; }
    for (int i = r_lo; i < r_hi; ++i) {
        lrp[i] = at;
        for (int p = (int)sh_ld<GM>(rhead + i); p != kLdsNil; p = (int)sh_ld<GM>(pnext + p)) {
            lci[at] = (int)((sh_ld<GM>(pmeta + p) >> RB) & RM);
            lv[at] = sh_ld<GM>(pval + p);
            ++at;
        }
        lci[at] = i;
        lv[at] = sh_ld<GM>(dgl + i);
        ++at;
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
    // ---- systems whose factor fits one CU's LDS: the pipeline of waves (k_icholt_lds).  DPCG_ICHOLT_LDS=0: never (development / tests);
    // DPCG_ICHOLT_WAVES = 4 | 8 | 16 (development)
    {
        const char *e_lds = getenv("DPCG_ICHOLT_LDS"), *e_waves = getenv("DPCG_ICHOLT_WAVES");
        const bool lds_on = !(e_lds && e_lds[0] == '0');
        const int waves_arg = e_waves ? atoi(e_waves) : 4, waves = waves_arg == 8 || waves_arg == 16 ? waves_arg : 4;   // (one a SIMD)
        // kept entries: at most nnz(A[k+1:, k]) + add_fill_in a column = the strict upper triangle of A + n add_fill_in
        const int64_t pool_cap = (A.nnz - n + 1) / 2 + n * (int64_t)add_fill_in + 1;
        // DPCG_ICHOLT_LDS = 2: the workspace form even where the LDS form fits (development / tests)
        const bool in_lds = n <= kLdsMaxRows && icholt_lds_bytes((int)n, (int)pool_cap, waves) + 64 <= 160 * 1024 && !(e_lds && e_lds[0] == '2');
        const bool in_mem = !in_lds && n <= kGmMaxRows;
        if (lds_on && n >= 1 && pool_cap < 0xff00 && (in_lds || in_mem)) {
            const int64_t cap_nnz = pool_cap + n;
            double *ws = nullptr;
            if ((st = dev_alloc(&status, 8 + 16 * 8)) < 0 || (st = dev_alloc(&Lf.rowptr, n + 1)) < 0 ||
                (st = dev_alloc(&Lf.col, cap_nnz)) < 0 || (st = dev_alloc(&Lf.val, cap_nnz)) < 0 ||
                (in_mem && (st = dev_alloc(&ws, (int64_t)(icholt_big_bytes((int)n, (int)pool_cap) / 8 + 2))) < 0)) {
                dev_free(ws);
                return cleanup(st);
            }
            hipError_t e = hipMemsetAsync(status, 0, 4 * sizeof(int), s);
            if (e != hipSuccess) { dev_free(ws); return cleanup(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__)); }
            const size_t lds = in_lds ? icholt_lds_bytes((int)n, (int)pool_cap, waves) : icholt_small_bytes(waves);
            const bool trace = getenv("DPCG_ICHOLT_TRACE") != nullptr;          // (development: cycles by phase, per wave, on stderr)
#define DPCG_ICHOLT_LDS_LAUNCH(WV, TRC, GMV)                                                                                         \
    do {                                                                                                                             \
        e = hipFuncSetAttribute((const void *)k_icholt_lds<WV, TRC, GMV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
        if (e == hipSuccess)                                                                                                         \
            hipLaunchKernelGGL((k_icholt_lds<WV, TRC, GMV>), dim3(1), dim3(WV * 64), lds, s, (int)n, A.rowptr, A.col, A.val,         \
                               add_fill_in, threshold, (int)pool_cap, Lf.rowptr, Lf.col, Lf.val, status, ws);                        \
    } while (0)
            if (in_mem) {
                if (trace) DPCG_ICHOLT_LDS_LAUNCH(4, true, true);
                else if (waves == 8) DPCG_ICHOLT_LDS_LAUNCH(8, false, true);
                else DPCG_ICHOLT_LDS_LAUNCH(4, false, true);
            } else if (trace && waves == 8) DPCG_ICHOLT_LDS_LAUNCH(8, true, false);
            else if (trace) DPCG_ICHOLT_LDS_LAUNCH(4, true, false);
            else if (waves == 8) DPCG_ICHOLT_LDS_LAUNCH(8, false, false);
            else if (waves == 16) DPCG_ICHOLT_LDS_LAUNCH(16, false, false);
            else DPCG_ICHOLT_LDS_LAUNCH(4, false, false);
#undef DPCG_ICHOLT_LDS_LAUNCH
            int h_st[3] = {0, 0, 0};
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(h_st, status, sizeof(h_st), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            dev_free(ws);                                                       // (the stream is idle)
            if (e != hipSuccess) return cleanup(hip_fail(e, "icholt (LDS)", __FILE__, __LINE__));
            if (trace) {
                int h_tr[16 * 8] = {0};
                const int tw = (waves == 8 && !in_mem) ? 8 : 4;
                (void)hipMemcpy(h_tr, status + 8, sizeof(h_tr), hipMemcpyDeviceToHost);
                for (int wv = 0; wv < tw; ++wv)
                    fprintf(stderr, "[icholt lds] wave %d: cycles per column (of %lld): row of A %.0f, waiting %.0f, dependencies %.0f, selection %.0f, publication %.0f\n",
                            wv, (long long)n, h_tr[wv * 8 + 0] * 16.0 * tw / n, h_tr[wv * 8 + 1] * 16.0 * tw / n, h_tr[wv * 8 + 2] * 16.0 * tw / n,
                            h_tr[wv * 8 + 3] * 16.0 * tw / n, h_tr[wv * 8 + 4] * 16.0 * tw / n);
            }
            if (h_st[0] == ICHOLT_OK) {
                Lf.nnz = h_st[2];
                return cleanup(DPCG_OK);
            }
            // not the plain case: the one-wave kernel runs it again (and names what is wrong, if something is)
            dev_free(status); dev_free(Lf.rowptr); dev_free(Lf.col); dev_free(Lf.val);
            status = nullptr; Lf.rowptr = nullptr; Lf.col = nullptr; Lf.val = nullptr;
        }
    }
    const int64_t wide = n * kIctCap;
    if ((st = dev_alloc(&rcnt, n)) < 0 || (st = dev_alloc(&rcol, wide)) < 0 || (st = dev_alloc(&rcc, wide)) < 0 ||
        (st = dev_alloc(&rval, wide)) < 0 || (st = dev_alloc(&crow, wide)) < 0 ||
        (st = dev_alloc(&cval, wide)) < 0 || (st = dev_alloc(&diag, n)) < 0 || (st = dev_alloc(&status, 2)) < 0 ||
        (st = dev_alloc(&cnt, n + 1)) < 0 || (st = dev_alloc(&Lf.rowptr, n + 1)) < 0)
        return cleanup(st);
    // DPCG_ICHOLT_REGS=0: every column through the LDS hash table (the path of columns with many pairs; development / test knob)
    const char *knob = getenv("DPCG_ICHOLT_REGS");
    const int use_regs = !(knob && knob[0] == '0');
    hipError_t e = hipMemsetAsync(rcnt, 0, (size_t)n * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(status, 0, 2 * sizeof(int), s);
    if (e != hipSuccess) return cleanup(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
    hipLaunchKernelGGL(k_icholt, dim3(1), dim3(64), 0, s, (int)n, A.rowptr, A.col, A.val, add_fill_in, threshold, rcnt, rcol, rcc,
                       rval, crow, cval, diag, status, use_regs);
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
