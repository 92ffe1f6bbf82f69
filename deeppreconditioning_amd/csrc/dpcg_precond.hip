// Host side of libdpcg.so, part 2: preconditioner setup -- Jacobi, explicit M, L L^T in multiply or solve mode and
// IC(0).  The host orchestrates; the analysis itself (transpose, level sets, level-ordered copies, ring-segment
// records, tril(A), the numeric factorisation) runs on the device.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dpcg_host.h"
#include "dpcg_prims.h"

// ------------------------------------------------------------------------------------------------
// preconditioners
// ------------------------------------------------------------------------------------------------
extern "C" int dpcg_set_precond_none(dpcg_handle_t h) {
    if (!h) return invalid("NULL handle");
    SetupScope scope(nullptr, true);
    free_precond(h);
    return DPCG_OK;
}

extern "C" int dpcg_set_precond_callback(dpcg_handle_t h, dpcg_precond_fn fn, void *user) {
    if (!h || !fn) return invalid("dpcg_set_precond_callback: NULL handle or function");
    SetupScope scope(nullptr, true);
    free_precond(h);
    h->precond_fn = fn;
    h->precond_user = user;
    h->precond = DPCG_PRECOND_CALLBACK;
    return DPCG_OK;
}

extern "C" int dpcg_set_precond_jacobi(dpcg_handle_t h, const double *dinv, int memspace, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);          // (the preconditioner being replaced may be in use on another stream)
    free_precond(h);
    DPCG_TRY(dev_alloc(&h->dinv, h->A.n));
    if (dinv && h->perm) {                           // the caller's numbering -> the handle's
        double *tmp = nullptr;
        const double *src = dinv;
        hipError_t e = hipSuccess;
        if (memspace == DPCG_HOST) {
            DPCG_TRY(dev_alloc(&tmp, h->A.n));
            e = hipMemcpyAsync(tmp, dinv, (size_t)h->A.n * sizeof(double), hipMemcpyHostToDevice, s);
            src = tmp;
        }
        if (e == hipSuccess) {
            launch_gather_f64(h->A.n, h->perm, src, h->dinv, s);
            e = hipStreamSynchronize(s);
        }
        dev_free(tmp);                                 // on every path, error or not
        DPCG_HIP(e);
    } else if (dinv) {
        DPCG_HIP(hipMemcpyAsync(h->dinv, dinv, (size_t)h->A.n * sizeof(double),
                                memspace == DPCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
        DPCG_HIP(hipStreamSynchronize(s));
    } else {
        int *d_bad = nullptr, h_bad = 0;
        DPCG_TRY(dev_alloc(&d_bad, 1));
        hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(int), s);
        if (e == hipSuccess) {
            launch_extract_dinv(h->A, h->dinv, d_bad, s);
            e = hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        dev_free(d_bad);
        DPCG_HIP(e);
        if (h_bad) {
            dev_free(h->dinv);
            set_error("Jacobi: missing or non-positive diagonal entry");
            return DPCG_ERR_PIVOT;
        }
    }
    h->precond = DPCG_PRECOND_JACOBI;
    return DPCG_OK;
}

extern "C" int dpcg_set_precond_csr(dpcg_handle_t h, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                    const double *val, int memspace, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    if (nnz <= 0 || !rowptr || !col || !val) return invalid("dpcg_set_precond_csr: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);          // (the preconditioner being replaced may be in use on another stream)
    free_precond(h);
    if (h->perm) {                                   // M arrives in the caller's numbering: iterate with P M P^T
        CsrDev Mu;
        DPCG_TRY(upload_csr(Mu, h->A.n, nnz, rowptr, col, val, DPCG_F64, memspace, 1, s));
        const int st = permute_csr(Mu, h->perm, h->iperm, h->M, s);
        free_csr(Mu);
        DPCG_TRY(st);
    } else {
        DPCG_TRY(upload_csr(h->M, h->A.n, nnz, rowptr, col, val, DPCG_F64, memspace, 1, s));
    }
    DPCG_TRY(make_plan(h->M, h->planM, s));
    h->precond = DPCG_PRECOND_CSR;
    return DPCG_OK;
}

// ------------------------------------------------------------------------------------------------
// L L^T preconditioners: the structural analysis (validation, transpose, level sets, level-ordered copies, ring
// records) runs on the device on data that is already in HBM (dpcg_analysis.hip + dpcg_prims.h); the host only sees
// the level offsets (one int per level) to plan the launches.
// ------------------------------------------------------------------------------------------------
namespace {
struct LevelSort;
int numeric_incomplete_cholesky(const LevelSort &ls, int64_t n, CsrDev &F, int *bad_dev, const double *colnorm, double tau,
                                hipStream_t s);
// DPCG_SYNCFREE=0 keeps one launch per wide level (development A/B)
bool syncfree_enabled() {
    static const bool on = [] { const char *e = getenv("DPCG_SYNCFREE"); return !(e && e[0] == '0'); }();
    return on;
}

template <typename T>
struct DevBuf {                       // scoped device allocation for the setup routines
    T *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { dev_free(p); }
    int alloc(int64_t count) { dev_free(p); return dev_alloc(&p, count); }
    T *release() { T *q = p; p = nullptr; return q; }
};

// rows sorted by (level, row), the level of every sorted position, the level offsets (device and host)
struct LevelSort {
    DevBuf<int32_t> rows;
    DevBuf<uint32_t> lvl_of_pos;
    DevBuf<int32_t> level_ptr_dev;
    std::vector<int32_t> level_ptr;
};

// `order_by` (may be null): inside a level the rows are ordered by order_by[row] instead of by row.  On a reordered handle that
// is the handle's (RCM) numbering: rows of a level are independent, so any order gives the same bits, and in this one the
// solve kernels read r, write z and gather their entries at ascending, clustered addresses instead of all over the vector
// (the caller's numbering is the scattered one -- that is why the handle was reordered).
int compute_levels(int64_t n, const int32_t *rp, const int32_t *ci, bool upper, LevelSort &out, hipStream_t s,
                   const int32_t *order_by = nullptr) {
    DevBuf<int32_t> level, iota, ctl;
    DPCG_TRY(level.alloc(n));
    DPCG_TRY(iota.alloc(n));
    DPCG_TRY(ctl.alloc(4));                                   // [0] ticket, [1] error flag, [2] max level
    DPCG_HIP(hipMemsetAsync(level.p, 0xff, (size_t)n * sizeof(int32_t), s));   // -1 = not known yet
    DPCG_HIP(hipMemsetAsync(ctl.p, 0, 4 * sizeof(int32_t), s));
    launch_levels_syncfree(n, rp, ci, upper, level.p, reinterpret_cast<unsigned int *>(ctl.p), ctl.p + 1, s);
    DPCG_CHECK_LAUNCH();
    DPCG_TRY(reduce_max_i32(level.p, ctl.p + 2, n, s));
    int32_t h_ctl[4] = {0, 0, 0, 0};
    DPCG_HIP(hipMemcpyAsync(h_ctl, ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    if (h_ctl[1] || h_ctl[2] < 0) {
        set_error("level analysis: the factor's dependency graph is not acyclic in row order");
        return DPCG_ERR_INVALID;
    }
    const int nl = h_ctl[2] + 1;
    DPCG_TRY(out.rows.alloc(n));
    DPCG_TRY(out.lvl_of_pos.alloc(n));
    DPCG_TRY(out.level_ptr_dev.alloc((int64_t)nl + 1));
    launch_iota(n, iota.p, s);
    if (order_by) {
        DevBuf<uint64_t> key, key_sorted;
        DPCG_TRY(key.alloc(n));
        DPCG_TRY(key_sorted.alloc(n));
        launch_level_keys(n, level.p, order_by, key.p, s);
        DPCG_TRY(sort_pairs_u64_i32(key.p, key_sorted.p, iota.p, out.rows.p, n, 32 + bits_for((uint64_t)h_ctl[2]), s));
        launch_key_levels(n, key_sorted.p, out.lvl_of_pos.p, s);
    } else {
        // stable: rows stay ascending inside a level
        DPCG_TRY(sort_pairs_u32_i32(reinterpret_cast<const uint32_t *>(level.p), out.lvl_of_pos.p, iota.p, out.rows.p, n,
                                    bits_for((uint64_t)h_ctl[2]), s));
    }
    launch_group_offsets(n, out.lvl_of_pos.p, nl, out.level_ptr_dev.p, s);
    out.level_ptr.resize((size_t)nl + 1);
    DPCG_HIP(hipMemcpyAsync(out.level_ptr.data(), out.level_ptr_dev.p, out.level_ptr.size() * sizeof(int32_t),
                            hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// Level sets of L^T from those of L (k_reverse_levels): no second level analysis -- a chain of n_levels dependent hand-offs,
// 0.4-0.9 ms for the factors of 65K-1M rows.  `lo` must still own its arrays (build_levels takes them over).
int reversed_levels(int64_t n, const LevelSort &lo, LevelSort &up, hipStream_t s) {
    const int nl = (int)lo.level_ptr.size() - 1;
    DPCG_TRY(up.rows.alloc(n));
    DPCG_TRY(up.lvl_of_pos.alloc(n));
    DPCG_TRY(up.level_ptr_dev.alloc((int64_t)nl + 1));
    launch_reverse_levels(n, lo.rows.p, lo.lvl_of_pos.p, lo.level_ptr_dev.p, nl, up.rows.p, up.lvl_of_pos.p, s);
    up.level_ptr.resize((size_t)nl + 1);
    for (int l = 0; l <= nl; ++l) up.level_ptr[(size_t)l] = (int32_t)(n - lo.level_ptr[(size_t)(nl - l)]);
    DPCG_HIP(hipMemcpyAsync(up.level_ptr_dev.p, up.level_ptr.data(), up.level_ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DPCG_HIP(hipStreamSynchronize(s));     // (the host vector is the source of an asynchronous copy)
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// The numeric phase of IC(0) / ICT on the pattern held in F (values: the matrix entries, zeros at fill positions): one
// launch per level, one thread per row.  (A sync-free single launch, rows polling `ready` flags, was measured and
// dropped: every read of another row then has to bypass the L1, and with the whole factor resident the polling drowned
// the front -- 1024^2: 56.6 ms vs 10.9 ms for 2047 launches; 100^3: 19.0 vs 2.7 ms.)
int numeric_incomplete_cholesky(const LevelSort &ls, int64_t n, CsrDev &F, int *bad_dev, const double *colnorm, double tau,
                                hipStream_t s) {
    (void)n;
    // (also measured and dropped: ONE workgroup walking a run of narrow levels with a barrier per level -- 1024^2: 24.2 ms, the
    // dependent agent-scope loads of a row cost more than the launches they replace)
    const int nl = (int)ls.level_ptr.size() - 1;
    for (int l = 0; l < nl; ++l)
        launch_ic0_level(ls.rows.p, ls.level_ptr[l], ls.level_ptr[l + 1] - ls.level_ptr[l], F.rowptr, F.col, F.val, bad_dev, s,
                         colnorm, tau);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// Everything launch_sptrsv needs for one factor (rp/ci/v: the factor in its own row order, device).
// `relabel` (may be null): old -> new index map of a reordered handle.  The factor is the caller's; only the indices the
// solve kernels use to address the right-hand side and the solution are mapped, so the arithmetic -- and its order --
// is that of the factor in the caller's numbering.
// Level-major numbering (Levels::level_major) pays when there are few, wide levels: every level then touches every line
// of three vectors once, and the two extra passes (in, out) cost about two such levels.  DPCG_LEVEL_MAJOR=0/1: off / forced.
static bool level_major_wanted(int64_t n, int n_levels) {
    static const int knob = [] { const char *e = getenv("DPCG_LEVEL_MAJOR"); return e ? atoi(e) : -1; }();
    if (knob == 0 || !syncfree_enabled()) return false;
    if (knob == 1) return true;
    // a multicolour-ordered factor has 2 .. ~8 levels of n / colours rows: the case this form is made for
    if (n >= 32768 && n_levels >= 2 && n_levels < 6 && n / n_levels >= 4096) return true;
    // measured, IC(0) of scrambled grids, us per PCG update without / with: 100^3 (19 levels of 53K rows on average) 270 / 146,
    // 64^3 (18 x 14.5K) 134 / 91, 1024^2 (18 x 58K) 201 / 141, 256^2 (13 x 5K) 78 / 54, 40^3 (19 x 3.4K) 83 / 92
    return n >= 32768 && n_levels >= 6 && n / n_levels >= 4096;
}

static bool level_major_syncfree() {              // development knob: DPCG_LM_SYNCFREE=0 keeps one launch per wide level
    static const bool on = [] { const char *e = getenv("DPCG_LM_SYNCFREE"); return !(e && e[0] == '0'); }();
    return on;
}

// `long_rows`: most rows hold more entries than the three of the LDS-ring / strip records (a 27-point stencil, a factor with
// fill): those kernels would walk nearly every row entry by entry, so such a factor takes the sync-free kernels with their
// wider records instead (27-point 64^3, IC(0): 24.6 ms per update in strips, see DESIGN).
// fewest rows per level (on average) for which a factor of <= 4 levels is solved by colour sweeps (DPCG_SWEEP_MIN_ROWS: development
// knob).  Before the sweeps were tiled and paired the single sync-free launch won below 131 072 rows a level; now, per PCG update
// with IC(0) in red-black order (tools/mc_probe.py): 256^2 (2 x 32 768 rows) 25.5 -> 17.3 us, 40^3 (2 x 32 000) 29.6 -> 26.6 us.
static int64_t sweep_min_rows() {
    static const int64_t v = [] { const char *e = getenv("DPCG_SWEEP_MIN_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)16384; }();
    return v;
}

// Up to 4 levels of >= sweep_min_rows() rows each; a fifth when the levels are very wide (>= 131 072 rows on average).  Measured at
// 1M rows on greedy colourings (tools/mc_probe.py, us per PCG update, sync-free launch -> sweeps): 5 levels 119 -> 98, 7 levels
// 123 -> 162 / 152 (hashed-priority colour classes scatter a level's columns: the gather sweep, many launches).
// DPCG_SWEEP_MAX_LEVELS: development knob (then for any width).
static bool sweep_levels_ok(int64_t n, int n_levels) {
    static const int knob = [] { const char *e = getenv("DPCG_SWEEP_MAX_LEVELS"); return e ? atoi(e) : 0; }();
    if (n_levels < 1 || n / n_levels < sweep_min_rows()) return false;
    if (knob > 0) return n_levels <= knob;
    return n_levels <= 4 || (n_levels == 5 && n / n_levels >= 131072);
}

int build_levels(Levels &lv, LevelSort &ls, int64_t n, int64_t nnz, const int32_t *rp, const int32_t *ci, const double *v,
                 hipStream_t s, const int32_t *relabel = nullptr, bool upper = false, bool long_rows = false) {
    constexpr int kMergeMax = 2048;  // levels this narrow are walked by one workgroup
    PhaseTimer pt(s);
    lv.level_ptr = ls.level_ptr;
    lv.n_levels = (int)ls.level_ptr.size() - 1;
    lv.level_major = level_major_wanted(n, lv.n_levels);
    const std::vector<int32_t> &level_ptr = lv.level_ptr;
    lv.rows = ls.rows.release();
    lv.level_ptr_dev = ls.level_ptr_dev.release();
    // level-ordered copy of the factor (row j = original row rows[j])
    DevBuf<int32_t> len, pos, flag;
    DPCG_TRY(len.alloc(n + 1));
    DPCG_TRY(pos.alloc(n));
    DPCG_TRY(flag.alloc(1));
    DPCG_TRY(dev_alloc(&lv.lo_rowptr, n + 1));
    DPCG_TRY(dev_alloc(&lv.lo_col, nnz));
    DPCG_TRY(dev_alloc(&lv.lo_cpos, nnz));
    DPCG_TRY(dev_alloc(&lv.lo_val, nnz));
    launch_lo_lengths(n, lv.rows, rp, len.p, pos.p, s);
    DPCG_TRY(exclusive_scan_i32(len.p, lv.lo_rowptr, n + 1, s));
    pt.mark("  alloc + lengths + scan");
    launch_lo_copy(n, lv.rows, rp, ci, v, pos.p, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, s);
    pt.mark("  level-ordered copy");
    if (relabel) {
        launch_relabel(n, relabel, lv.rows, s);
        launch_relabel(nnz, relabel, lv.lo_col, s);
    }
    DPCG_HIP(hipMemsetAsync(flag.p, 0, sizeof(int32_t), s));
    launch_stream_fit(n, ls.lvl_of_pos.p, lv.level_ptr_dev, lv.lo_rowptr, flag.p, s);
    // segments: runs of narrow levels are merged (one workgroup walks them), wide levels launch one by one
    lv.segments.clear();
    std::vector<int32_t> seg_of_level((size_t)lv.n_levels, -1), seg_start;
    std::vector<int> merged_index;                     // index into lv.segments of merged segment q
    int l = 0;
    while (l < lv.n_levels) {
        const bool narrow = (level_ptr[l + 1] - level_ptr[l]) <= kMergeMax;
        int e = l + 1;
        while (e < lv.n_levels && ((level_ptr[e + 1] - level_ptr[e]) <= kMergeMax) == narrow &&
               (!narrow || e - l < kRingMaxLevels))
            ++e;
        Levels::Segment seg;
        seg.lo = l;
        seg.hi = e;
        seg.merged = narrow && (e - l) >= 2;
        seg.ring_w = 0;
        seg.max_width = 0;
        for (int q = l; q < e; ++q) seg.max_width = std::max<int>(seg.max_width, level_ptr[q + 1] - level_ptr[q]);
        if (seg.merged) {
            for (int q = l; q < e; ++q) seg_of_level[q] = (int32_t)seg_start.size();
            seg_start.push_back(level_ptr[l]);
            merged_index.push_back((int)lv.segments.size());
        }
        lv.segments.push_back(seg);
        l = e;
    }
    int32_t h_flag = 0;
    DPCG_HIP(hipMemcpyAsync(&h_flag, flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    bool any_ring = false;
    DevBuf<int32_t> d_seg_of_level;
    if (!seg_start.empty()) {
        // LDS ring: in level order, how far back do a merged segment's rows reach (within the segment)?
        DevBuf<int32_t> d_seg_start, d_maxdist;
        const int64_t nseg = (int64_t)seg_start.size();
        DPCG_TRY(d_seg_of_level.alloc(lv.n_levels));
        DPCG_TRY(d_seg_start.alloc(nseg));
        DPCG_TRY(d_maxdist.alloc(nseg));
        DPCG_HIP(hipMemcpyAsync(d_seg_of_level.p, seg_of_level.data(), seg_of_level.size() * sizeof(int32_t),
                                hipMemcpyHostToDevice, s));
        DPCG_HIP(hipMemcpyAsync(d_seg_start.p, seg_start.data(), seg_start.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        DPCG_HIP(hipMemsetAsync(d_maxdist.p, 0, (size_t)nseg * sizeof(int32_t), s));
        launch_ring_reach(n, ls.lvl_of_pos.p, d_seg_of_level.p, d_seg_start.p, lv.lo_rowptr, lv.lo_cpos, d_maxdist.p, s);
        std::vector<int32_t> maxdist((size_t)nseg, 0);
        DPCG_HIP(hipMemcpyAsync(maxdist.data(), d_maxdist.p, maxdist.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        for (int64_t q = 0; q < nseg; ++q) {
            Levels::Segment &seg = lv.segments[(size_t)merged_index[(size_t)q]];
            int64_t w = 64;
            while (w < (int64_t)maxdist[(size_t)q] + seg.max_width + 1) w *= 2;
            if (w <= 8192) {                             // 64 KiB of LDS at most
                seg.ring_w = (int)w;
                any_ring = true;
            }
        }
    } else {
        DPCG_HIP(hipStreamSynchronize(s));
    }
    lv.stream_ok = h_flag == 0;
    pt.mark("  segments + ring reach");
    // Everything that is not walked through an LDS ring -- wide levels, and narrow runs whose reach is too long for the
    // ring -- goes to the sync-free multi-workgroup kernel, neighbouring such segments as ONE launch.  A single level on
    // its own keeps the plain level kernel (nothing inside it to wait for).
    if (syncfree_enabled()) {
        // Classes of segments: RING (a run of narrow levels the pipelined LDS-ring kernel can walk: reach inside the ring,
        // levels <= 1024 rows, LDS fits), NARROW (any other run whose levels average <= kSyncfreeMaxMeanWidth rows: one
        // sync-free launch) and WIDE (one launch per level, on records).  The old one-level-ahead ring walk
        // (k_sptrsv_ring) is no longer chosen: on the 6-level tail of the scrambled 1M-DoF factor it took 45 us of a
        // 205 us solve.  Neighbouring segments of the same class merge; NARROW and WIDE runs stay apart, so that the
        // narrow tail of a wide-level factor does not fall back to a launch per level.
        static const int64_t kSyncfreeMaxMeanWidth = [] {
            const char *e = getenv("DPCG_SF_MAX_MEAN_WIDTH");      // development knob
            return e ? (int64_t)atoll(e) : (int64_t)16384;
        }();
        auto cls = [&](const Levels::Segment &seg) {
            const size_t lds = (size_t)seg.ring_w * sizeof(double) + (size_t)(seg.hi - seg.lo + 24) * sizeof(int);
            // (a level-major factor has no ring segments: the ring kernels address the vectors by row)
            if (!lv.level_major && !long_rows && seg.merged && seg.ring_w > 0 && seg.max_width <= 1024 && lds <= 64 * 1024)
                return 0;                                                                                 // RING
            // level-major: the whole factor as ONE sync-free launch -- except a factor of a few VERY wide levels (IC(0) in
            // multicolour order at 1M rows: 2 levels of 500K): there a launch per level on the records wins (measured per
            // apply, 100^3 red-black: 104.7 us sync-free, 63.6 with four blocks per ticket, 59.4 one launch per level;
            // 256^2 red-black, 2 x 32K rows: 19.8 sync-free, 30.6 per level)
            const bool few_very_wide = sweep_levels_ok(n, lv.n_levels);
            if (lv.level_major && level_major_syncfree() && !few_very_wide) return 1;
            if (lv.level_major && few_very_wide) return 2;
            const int64_t rows_in_seg = (int64_t)level_ptr[seg.hi] - level_ptr[seg.lo];
            return rows_in_seg / (seg.hi - seg.lo) <= kSyncfreeMaxMeanWidth ? 1 : 2;                     // NARROW : WIDE
        };
        // first split what the width-based pass produced into per-level pieces where it is not a ring, then re-merge by class
        std::vector<Levels::Segment> pieces;
        for (const auto &seg : lv.segments) {
            if (cls(seg) == 0) {
                pieces.push_back(seg);
                continue;
            }
            for (int q = seg.lo; q < seg.hi; ++q) {
                Levels::Segment one = seg;
                one.lo = q;
                one.hi = q + 1;
                one.merged = false;
                one.ring_w = 0;
                one.max_width = level_ptr[q + 1] - level_ptr[q];
                pieces.push_back(one);
            }
        }
        std::vector<Levels::Segment> merged_segs;
        std::vector<int> merged_cls;
        for (const auto &seg : pieces) {
            const int c = cls(seg);
            if (c != 0 && !merged_segs.empty() && merged_cls.back() == c) {
                Levels::Segment &b = merged_segs.back();
                b.hi = seg.hi;
                b.max_width = std::max(b.max_width, seg.max_width);
                continue;
            }
            merged_segs.push_back(seg);
            merged_cls.push_back(c);
        }
        for (size_t q = 0; q < merged_segs.size(); ++q) {
            Levels::Segment &seg = merged_segs[q];
            if (merged_cls[q] == 0) continue;
            seg.merged = false;
            seg.ring_w = 0;
            seg.syncfree = merged_cls[q] == 1 && seg.hi - seg.lo >= 2;   // a single level on its own: nothing to wait for inside
        }
        lv.segments.swap(merged_segs);
        any_ring = false;
        for (const auto &seg : lv.segments) any_ring = any_ring || (seg.merged && seg.ring_w > 0);
    }
    {
        const int64_t nseg = (int64_t)lv.segments.size();
        DPCG_TRY(dev_alloc(&lv.tickets, nseg));
        DPCG_TRY(dev_alloc(&lv.spin_err, 1));
        DPCG_HIP(hipMemsetAsync(lv.tickets, 0, (size_t)nseg * sizeof(unsigned long long), s));
        DPCG_HIP(hipMemsetAsync(lv.spin_err, 0, sizeof(int), s));
        // colour sweeps (see Levels::sweep): a level-major factor of a few very wide levels whose blocks fit the LDS product buffer
        static const bool sweeps_on = [] { const char *e = getenv("DPCG_SWEEPS"); return !(e && e[0] == '0'); }();
        if (sweeps_on && lv.level_major && lv.stream_ok && sweep_levels_ok(n, lv.n_levels)) {
            lv.sweep = true;
            int64_t widest = 0;
            for (int l = 0; l < lv.n_levels; ++l) widest = std::max<int64_t>(widest, level_ptr[l + 1] - level_ptr[l]);
            const int64_t blocks = (widest + kBlock - 1) / kBlock;
            lv.sweep_grid = (int)std::max<int64_t>(1, std::min<int64_t>(blocks, kMaxSpmvGrid));
            // x-tile plans of the levels (k_lm_sweep_tile); a level without off-diagonal entries, or with a block whose columns
            // are too spread out, keeps the gather sweep
            static const bool tiles_on = [] { const char *e = getenv("DPCG_SWEEP_TILES"); return !(e && e[0] == '0'); }();
            if (tiles_on) {
                lv.sw_blk0.assign((size_t)lv.n_levels + 1, 0);
                for (int l = 0; l < lv.n_levels; ++l)
                    lv.sw_blk0[(size_t)l + 1] = lv.sw_blk0[(size_t)l] + (level_ptr[l + 1] - level_ptr[l] + kBlock - 1) / kBlock;
                const int64_t nb = lv.sw_blk0.back();
                DPCG_TRY(dev_alloc(&lv.sw_chunks, nb * kTileMaxChunks));
                DPCG_TRY(dev_alloc(&lv.sw_nchunks, nb));
                DPCG_TRY(dev_alloc(&lv.sw_lidx, nnz + 4));
                DPCG_HIP(hipMemsetAsync(lv.sw_lidx + nnz, 0, 4 * sizeof(uint16_t), s));
                DevBuf<int32_t> d_flags;
                DPCG_TRY(d_flags.alloc(2 * (int64_t)lv.n_levels));
                std::vector<int32_t> h_flags((size_t)2 * lv.n_levels);
                for (int l = 0; l < lv.n_levels; ++l) { h_flags[(size_t)2 * l] = 1; h_flags[(size_t)2 * l + 1] = 0; }
                DPCG_HIP(hipMemcpyAsync(d_flags.p, h_flags.data(), h_flags.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
                for (int l = 0; l < lv.n_levels; ++l)
                    launch_sweep_tile_plan(level_ptr[l], level_ptr[l + 1] - level_ptr[l], lv.lo_rowptr, lv.lo_cpos,
                                           lv.sw_chunks + (size_t)lv.sw_blk0[(size_t)l] * kTileMaxChunks, lv.sw_nchunks + lv.sw_blk0[(size_t)l],
                                           lv.sw_lidx, reinterpret_cast<int *>(d_flags.p) + 2 * l, s);
                DPCG_HIP(hipMemcpyAsync(h_flags.data(), d_flags.p, h_flags.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                DPCG_HIP(hipStreamSynchronize(s));
                static const int nt_knob = [] { const char *e = getenv("DPCG_SWEEP_NT"); return e ? atoi(e) : -1; }();
                lv.sweep_nt = nt_knob >= 0 ? nt_knob != 0 : 10.0 * (double)nnz + 40.0 * (double)n >= 256e6;   // (measured: 160^3 -2.4 %, 256^3 -1.7 % per update)
                lv.sw_max_chunks.assign((size_t)lv.n_levels, 0);
                int most = 0;
                for (int l = 0; l < lv.n_levels; ++l) {
                    if (h_flags[(size_t)2 * l] == 1) lv.sw_max_chunks[(size_t)l] = h_flags[(size_t)2 * l + 1];
                    most = std::max(most, lv.sw_max_chunks[(size_t)l]);
                }
                if (most > 0) {             // as make_plan: workgroups per CU by the LDS a block takes, a multiple of 8
                    const size_t lds = (size_t)(most * kTileChunk + kStreamCap + 8) * sizeof(double);
                    const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
                    static const int grid_knob = [] { const char *e = getenv("DPCG_SWEEP_GRID"); return e ? atoi(e) : 0; }();
                    static const int cyc_knob = [] { const char *e = getenv("DPCG_SWEEP_CYCLIC"); return e ? atoi(e) : -1; }();
                    lv.sweep_cyclic = cyc_knob >= 0 ? cyc_knob != 0 : lv.sweep_nt;
                    const int grid_cap = grid_knob > 0 ? grid_knob : (lv.sweep_cyclic ? 768 : 1024);
                    int g = (int)std::min<int64_t>(blocks, std::min<int64_t>(std::min(per_cu * 256, grid_cap), kMaxSpmvGrid));
                    if (g > 8) g -= g % 8;
                    lv.sweep_grid = std::max(1, g);
                }
            }
        }
        bool any = false;
        for (const auto &seg : lv.segments) any = any || !(seg.merged && seg.ring_w > 0);
        if (any && !lv.sweep) {                          // records for the sync-free and the per-level kernels
            // width of the records: 3 entries unless more than 2 % of the rows are longer (they would take the slow general
            // path: three dependent loads before the first entry, then one entry at a time), then 6, then 14
            DevBuf<int32_t> n_long;
            DPCG_TRY(n_long.alloc(2));
            DPCG_HIP(hipMemsetAsync(n_long.p, 0, 2 * sizeof(int32_t), s));
            launch_count_long_rows(n, lv.lo_rowptr, 3, n_long.p, s);
            launch_count_long_rows(n, lv.lo_rowptr, 6, n_long.p + 1, s);
            int32_t h_long[2] = {0, 0};
            DPCG_HIP(hipMemcpyAsync(h_long, n_long.p, sizeof(h_long), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            lv.rec_w = (int64_t)h_long[1] * 50 > n ? 14 : ((int64_t)h_long[0] * 50 > n ? 6 : 3);
            const int64_t stride = lv.rec_w == 14 ? 16 : (lv.rec_w == 6 ? 8 : 4);
            DPCG_TRY(dev_alloc(&lv.sf_meta, n * stride));
            DPCG_TRY(dev_alloc(&lv.sf_val, n * stride));
            if (lv.level_major)
                launch_sf_records(n, nullptr, lv.lo_rowptr, lv.lo_cpos, lv.lo_val, upper, lv.sf_meta, lv.sf_val, lv.rec_w, s);
            else
                launch_sf_records(n, lv.rows, lv.lo_rowptr, lv.lo_col, lv.lo_val, upper, lv.sf_meta, lv.sf_val, lv.rec_w, s);
        }
        if (lv.level_major) {
            DPCG_TRY(dev_alloc(&lv.lm_pos, n));
            DPCG_TRY(dev_alloc(&lv.lm_rhs, n));
            DPCG_TRY(dev_alloc(&lv.lm_out, n));
            launch_invert_positions(n, lv.rows, lv.lm_pos, s);
            launch_fill_pending(lv.lm_out, n, s);        // the invariant of Levels::lm_out
            // The whole factor ONE sync-free launch: in CSR-stream form (k_sptrsv_syncfree_stream; blocks of <= 512 rows of one level
            // each) when the rows are long -- records of width 14 move 256 bytes per row whatever it holds.  Measured per PCG update
            // at 1M rows, records -> stream (tools/sfs_probe.py, profiles/r04_sfs_probe.txt): Delaunay IC(0) (19 levels, rows of up to
            // 12 entries) 188 -> 148 us; the scrambled 100^3 / 1024^2 factors (width 6 / 3) 136 -> 140 / 131 -> 131; 64^3 72 -> 103
            // (blocks of 256 or 1024 rows are worse everywhere: 179 / 168 us on 100^3 -- tickets, not bytes, bound these solves).
            // DPCG_SF_STREAM=0/1 overrides.
            static const int sfs_knob = [] { const char *e = getenv("DPCG_SF_STREAM"); return e ? atoi(e) : -1; }();
            const bool sfs_on = sfs_knob >= 0 ? sfs_knob != 0 : lv.rec_w == 14;
            if (sfs_on && !lv.sweep && lv.segments.size() == 1 && lv.segments[0].syncfree) {
                std::vector<int32_t> blk;
                for (int l = 0; l < lv.n_levels; ++l)
                    for (int j0 = level_ptr[l]; j0 < level_ptr[l + 1]; j0 += kSfsBlock) {
                        blk.push_back(j0);
                        blk.push_back(std::min(j0 + kSfsBlock, level_ptr[l + 1]));
                    }
                const int nblk = (int)(blk.size() / 2);
                DevBuf<int32_t> most;
                DPCG_TRY(most.alloc(1));
                DPCG_TRY(dev_alloc(&lv.sfs_blk, (int64_t)blk.size()));
                DPCG_HIP(hipMemcpyAsync(lv.sfs_blk, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
                DPCG_HIP(hipMemsetAsync(most.p, 0, sizeof(int32_t), s));
                launch_sfs_block_max(lv.sfs_blk, nblk, lv.lo_rowptr, reinterpret_cast<int *>(most.p), s);
                int32_t h_most = 0;
                DPCG_HIP(hipMemcpyAsync(&h_most, most.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
                DPCG_HIP(hipStreamSynchronize(s));       // (blk is a local)
                if (h_most > kStreamCap / kBlock * kSfsBlock) dev_free(lv.sfs_blk);      // a block's entries do not fit the product buffer: records
                else lv.sfs_nblk = nblk;
            }
        }
    }
    pt.mark("  sync-free records");
    if (any_ring) {                                      // fixed-width row records (see Levels::pk_meta)
        std::vector<int32_t> ring_start((size_t)lv.n_levels, -1);
        for (const auto &seg : lv.segments)
            if (seg.ring_w > 0)
                for (int q = seg.lo; q < seg.hi; ++q) ring_start[(size_t)q] = level_ptr[seg.lo];
        DPCG_HIP(hipMemcpyAsync(d_seg_of_level.p, ring_start.data(), ring_start.size() * sizeof(int32_t),
                                hipMemcpyHostToDevice, s));
        DPCG_TRY(dev_alloc(&lv.pk_meta, n * 4));
        DPCG_TRY(dev_alloc(&lv.pk_val, n * 4));
        DPCG_TRY(dev_alloc(&lv.b_lo, n));
        launch_ring_records(n, ls.lvl_of_pos.p, d_seg_of_level.p, lv.rows, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val,
                            lv.pk_meta, lv.pk_val, s);
        DPCG_HIP(hipStreamSynchronize(s));
        pt.mark("  ring records");
    }
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// Strip plan of a factor (see k_sptrsv_strips): tried for banded factors with many levels; kept when every strip-local level
// fits one workgroup, the LDS ring covers the reach inside a strip and most entries stay inside their strip.  Leaves
// lv.strips.n_strips = 0 otherwise (the level schedule built above is then what launch_sptrsv uses).
bool strips_enabled() {
    static const bool on = [] { const char *e = getenv("DPCG_STRIPS"); return !(e && e[0] == '0'); }();
    return on;
}

// `levels_known` false: the global level sets have not been computed (IC(0) by strips builds the plan BEFORE anything else);
// the plan is then tried for a banded pattern (band <= n / 32) and kept only with >= 32 strip-local levels.
// `factor_rows` (may be null): receives the factor's row at every position (sp.rows holds the handle's index after relabelling).
int build_strips(Levels &lv, int64_t n, int64_t nnz, const int32_t *rp, const int32_t *ci, const double *v, bool upper,
                 const int32_t *relabel, hipStream_t s, bool levels_known = true, DevBuf<int32_t> *factor_rows = nullptr) {
    if (!strips_enabled() || (levels_known && lv.n_levels < 64) || n < 32768) return DPCG_OK;
    static const int64_t target_rows = [] { const char *e = getenv("DPCG_STRIP_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)16384; }();
    if (n >= (int64_t)1 << 30) return DPCG_OK;                // bit 30 of a row index is the "published" mark
    // a factor whose whole schedule is one LDS-ring walk and that is small enough for one CU to stream keeps that walk
    // (measured at 256^2, 511 levels: 0.36 ms per apply on the ring, 0.66 ms in 8 strips)
    if (n <= 131072 && lv.segments.size() == 1 && lv.segments[0].merged && lv.segments[0].ring_w > 0 && lv.pk_meta &&
        lv.segments[0].max_width <= 1024)
        return DPCG_OK;
    PhaseTimer pt(s);
    DevBuf<int32_t> level, iota, ctl, len, pos, exported;
    DevBuf<uint32_t> key, key_sorted;
    DPCG_TRY(level.alloc(n)); DPCG_TRY(iota.alloc(n)); DPCG_TRY(ctl.alloc(8)); DPCG_TRY(key.alloc(n)); DPCG_TRY(key_sorted.alloc(n));
    DPCG_TRY(exported.alloc(n));
    DPCG_HIP(hipMemsetAsync(ctl.p, 0, 8 * sizeof(int32_t), s));
    // strips are cut at multiples of the band (for a grid: whole planes / grid lines), so that an entry one band back sits one
    // strip-local level back and not dozens
    launch_max_band(n, rp, ci, upper, reinterpret_cast<int *>(ctl.p + 4), s);
    int32_t bands[2] = {0, 0};
    DPCG_HIP(hipMemcpyAsync(bands, ctl.p + 4, sizeof(bands), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    const int32_t band = bands[0] < 1 ? 1 : bands[0], inner = bands[1] < 1 ? 1 : bands[1];
    if (!levels_known && (int64_t)band * 32 > n) return DPCG_OK;
    // One attempt at a plan for a given strip map; leaves lv.strips.n_strips = 0 when the plan is not kept.
    auto attempt = [&](const StripMap &map, int64_t S) -> int {
        DPCG_HIP(hipMemsetAsync(level.p, 0xff, (size_t)n * sizeof(int32_t), s));
        DPCG_HIP(hipMemsetAsync(ctl.p, 0, 8 * sizeof(int32_t), s));
        launch_levels_syncfree(n, rp, ci, upper, level.p, reinterpret_cast<unsigned int *>(ctl.p), ctl.p + 1, s, map);
        DPCG_TRY(reduce_max_i32(level.p, ctl.p + 2, n, s));
        int32_t h_ctl[8] = {0};
        DPCG_HIP(hipMemcpyAsync(h_ctl, ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (h_ctl[1] || h_ctl[2] < 0) return DPCG_OK;            // (cannot happen: the global analysis succeeded)
        const int nlev = h_ctl[2] + 1;
        if ((int64_t)S * nlev > (int64_t)1 << 24) return DPCG_OK;
        if (!levels_known && nlev < 32) return DPCG_OK;
        struct Pending {                                          // the plan under construction: released unless it is kept
            Levels::Strips sp;
            bool keep = false;
            ~Pending() {
                if (keep) return;
                dev_free(sp.rows); dev_free(sp.level_ptr_dev); dev_free(sp.lo_rowptr); dev_free(sp.lo_col); dev_free(sp.lo_cpos);
                dev_free(sp.lo_val); dev_free(sp.val); dev_free(sp.b_lo); dev_free(sp.meta); dev_free(sp.ticket);
            }
        } pending;
        Levels::Strips &sp = pending.sp;
        auto drop = [&]() { return DPCG_OK; };                    // (~Pending frees)
        DPCG_TRY(dev_alloc(&sp.rows, n));
        DPCG_TRY(dev_alloc(&sp.level_ptr_dev, S * nlev + 1));
        launch_strip_keys(n, level.p, map, nlev, upper, key.p, s);
        launch_iota(n, iota.p, s);
        DPCG_TRY(sort_pairs_u32_i32(key.p, key_sorted.p, iota.p, sp.rows, n, bits_for((uint64_t)(S * nlev)), s));
        launch_group_offsets(n, key_sorted.p, (int)(S * nlev), sp.level_ptr_dev, s);
        std::vector<int32_t> lptr((size_t)(S * nlev) + 1);
        DPCG_HIP(hipMemcpyAsync(lptr.data(), sp.level_ptr_dev, lptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        int width = 0;
        for (size_t q = 0; q + 1 < lptr.size(); ++q) width = std::max(width, lptr[q + 1] - lptr[q]);
        if (width > 1024) {
            if (pt.on) fprintf(stderr, "[dpcg setup] strip plan dropped: a strip-local level has %d rows (> 1024)\n", width);
            return drop();
        }
        // level-ordered copy in (strip, level, row) order
        DPCG_TRY(len.alloc(n + 1)); DPCG_TRY(pos.alloc(n));
        DPCG_TRY(dev_alloc(&sp.lo_rowptr, n + 1)); DPCG_TRY(dev_alloc(&sp.lo_col, nnz)); DPCG_TRY(dev_alloc(&sp.lo_cpos, nnz));
        DPCG_TRY(dev_alloc(&sp.lo_val, nnz)); DPCG_TRY(dev_alloc(&sp.meta, n * 4)); DPCG_TRY(dev_alloc(&sp.val, n * 4));
        DPCG_TRY(dev_alloc(&sp.b_lo, n)); DPCG_TRY(dev_alloc(&sp.ticket, 2));
        launch_lo_lengths(n, sp.rows, rp, len.p, pos.p, s);
        DPCG_TRY(exclusive_scan_i32(len.p, sp.lo_rowptr, n + 1, s));
        launch_lo_copy(n, sp.rows, rp, ci, v, pos.p, sp.lo_rowptr, sp.lo_col, sp.lo_cpos, sp.lo_val, s);
        if (factor_rows) {
            DPCG_TRY(factor_rows->alloc(n));
            DPCG_HIP(hipMemcpyAsync(factor_rows->p, sp.rows, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        }
        if (relabel) {
            launch_relabel(n, relabel, sp.rows, s);
            launch_relabel(nnz, relabel, sp.lo_col, s);
        }
        DPCG_HIP(hipMemsetAsync(ctl.p, 0, 8 * sizeof(int32_t), s));
        DPCG_HIP(hipMemsetAsync(sp.ticket, 0, 2 * sizeof(unsigned int), s));
        constexpr int kRingReach = 8192 - 1024 - 1;               // what a ring of 8192 doubles covers beside the widest level
        DPCG_HIP(hipMemsetAsync(exported.p, 0, (size_t)n * sizeof(int32_t), s));
        launch_strip_records(n, key_sorted.p, nlev, sp.level_ptr_dev, sp.rows, sp.lo_rowptr, sp.lo_col, sp.lo_cpos, sp.lo_val, upper,
                             kRingReach, sp.meta, sp.val, exported.p, reinterpret_cast<int *>(ctl.p), s);
        DPCG_HIP(hipMemcpyAsync(h_ctl, ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        DPCG_CHECK_LAUNCH();
        const int64_t reach = h_ctl[0], external = h_ctl[1], offdiag = nnz - n;
        if (h_ctl[3] > 0) {                                       // an entry in a LATER strip: tickets could not guarantee progress
            if (pt.on) fprintf(stderr, "[dpcg setup] strip plan (%d parts) dropped: %d entries live in later strips\n", map.parts, h_ctl[3]);
            return drop();
        }
        int64_t W = 64;
        while (W < reach + width + 1) W *= 2;
        if (W > 8192 || (offdiag > 0 && external * 2 > offdiag)) {                   // ring too long, or mostly foreign entries
            if (pt.on)
                fprintf(stderr, "[dpcg setup] strip plan dropped: reach %lld + width %d -> ring %lld, %lld of %lld entries foreign\n",
                        (long long)reach, width, (long long)W, (long long)external, (long long)offdiag);
            return drop();
        }
        sp.long_rows = h_ctl[2];
        sp.n_strips = (int)S;
        sp.nlev = nlev;
        sp.W = (int)W;
        sp.ring_reach = kRingReach;
        sp.rows_per_thread = width <= 512 ? 1 : 2;
        sp.threads = ((width + sp.rows_per_thread - 1) / sp.rows_per_thread + 63) / 64 * 64;
        sp.threads = sp.threads < 64 ? 64 : sp.threads;
        lv.strips = sp;
        pending.keep = true;
        init_strip_kernels();
        // the global level schedule's big arrays are not used when the strip plan is: give the memory back
        dev_free(lv.lo_rowptr); dev_free(lv.lo_col); dev_free(lv.lo_cpos); dev_free(lv.lo_val);
        dev_free(lv.pk_meta); dev_free(lv.pk_val); dev_free(lv.b_lo); dev_free(lv.sf_meta); dev_free(lv.sf_val);
        return DPCG_OK;
    };
    // First the two-way cut (slabs of whole bands x parts of a band: for a grid, pencils): the longest dependency chain then
    // crosses about slabs + parts strip boundaries (2.7 us each) instead of slabs * parts.  Kept only if every entry lives in
    // the own or an EARLIER strip (true for natural-order grids; checked, not assumed).
    static const bool two_way = [] { const char *e = getenv("DPCG_STRIP_PARTS"); return !(e && e[0] == '1' && e[1] == 0); }();
    const int64_t want = (n + target_rows - 1) / target_rows;              // strips of ~target_rows rows
    // (a pattern whose entries point further along the band -- ICT's fill entry (i, i - nx + 1) -- cannot be cut two ways)
    int32_t along = 0;
    if (two_way && band >= 128 && want >= 8) {
        DPCG_HIP(hipMemsetAsync(ctl.p + 6, 0, sizeof(int32_t), s));
        launch_points_along_band(n, rp, ci, upper, band, reinterpret_cast<int *>(ctl.p + 6), s);
        DPCG_HIP(hipMemcpyAsync(&along, ctl.p + 6, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
    }
    if (two_way && band >= 128 && want >= 8 && !along) {
        int parts = (int)std::llround(std::sqrt((double)want));
        parts = std::max(2, std::min<int>(16, std::min<int>(parts, band / 32)));
        StripMap map;
        map.band = band;
        // parts are cut at multiples of the inner band (for a 3-D grid: whole grid lines -- a part that starts in the middle of a
        // line trails its neighbour by half a strip instead of a few levels: measured, 100^3, 72 us instead of 12 us per part)
        map.sub_w = ((band + parts - 1) / parts + inner - 1) / inner * inner;
        map.parts = (band + map.sub_w - 1) / map.sub_w;
        int64_t slabs = std::max<int64_t>(1, (want + map.parts - 1) / map.parts);
        int64_t rows64 = ((n + slabs - 1) / slabs + band - 1) / band * band;
        slabs = (n + rows64 - 1) / rows64;
        if (slabs * map.parts <= 256 && rows64 < ((int64_t)1 << 31)) {
            map.rows = (int)rows64;
            DPCG_TRY(attempt(map, slabs * map.parts));
            if (lv.strips.n_strips > 0) {
                pt.mark(upper ? "strip plan (L^T, two-way)" : "strip plan (L, two-way)");
                return DPCG_OK;
            }
        }
    }
    int64_t per = (target_rows + band / 2) / band;
    per = per < 1 ? 1 : per;
    int64_t strip_rows64 = per * band;
    int64_t S = (n + strip_rows64 - 1) / strip_rows64;
    if (S > 256) {                                            // keep every strip's workgroup resident
        strip_rows64 = ((n + 255) / 256 + band - 1) / band * band;
        S = (n + strip_rows64 - 1) / strip_rows64;
    }
    if (S < 4) return DPCG_OK;
    StripMap map;
    map.rows = (int)strip_rows64;
    map.band = band;
    DPCG_TRY(attempt(map, S));
    pt.mark(upper ? "strip plan (L^T)" : "strip plan (L)");
    return DPCG_OK;
}

// L^T as CSR (columns ascending, diagonal first): stable sort of the entries by column.
int transpose_lower(const CsrDev &L, CsrDev &Lt, hipStream_t s) {
    const int64_t n = L.n, nnz = L.nnz;
    DevBuf<int32_t> row_of, order_in, order;
    DevBuf<uint32_t> cols_sorted;
    DPCG_TRY(row_of.alloc(nnz));
    DPCG_TRY(order_in.alloc(nnz));
    DPCG_TRY(order.alloc(nnz));
    DPCG_TRY(cols_sorted.alloc(nnz));
    Lt = CsrDev();
    Lt.n = n;
    Lt.nnz = nnz;
    Lt.owned = true;
    DPCG_TRY(dev_alloc(&Lt.rowptr, n + 1));
    DPCG_TRY(dev_alloc(&Lt.col, nnz));
    DPCG_TRY(dev_alloc(&Lt.val, nnz));
    launch_row_of(n, L.rowptr, row_of.p, s);
    launch_iota(nnz, order_in.p, s);
    DPCG_TRY(sort_pairs_u32_i32(reinterpret_cast<const uint32_t *>(L.col), cols_sorted.p, order_in.p, order.p, nnz,
                                bits_for((uint64_t)(n - 1)), s));
    launch_group_offsets(nnz, cols_sorted.p, (int)n, Lt.rowptr, s);
    launch_transpose_gather(nnz, order.p, row_of.p, L.val, Lt.col, Lt.val, s);
    DPCG_HIP(hipStreamSynchronize(s));
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// h->L holds a lower-triangular factor on the device (owned by the handle): validate it, build L^T, the SpMV plans and,
// in solve mode, the level schedules.  `lower_levels`: level analysis of L when the caller already has it (IC(0)).
// `lower_prebuilt`: the schedule of L when IC(0) was factored THROUGH it (a strip plan -- its global level sets were never
// computed, n_levels = -1: counted when dpcg_get_info asks -- or a one-segment LDS-ring schedule).
// `upper_levels`: level sets of L^T the caller already has (the reversed level sets of L, taken before a prebuilt schedule took
// those over).
int finish_llt(dpcg_system *h, int mode, hipStream_t s, LevelSort *lower_levels = nullptr, Levels *lower_prebuilt = nullptr,
               LevelSort *upper_levels = nullptr) {
    const int64_t n = h->A.n;
    DevBuf<int32_t> flags;
    DPCG_TRY(flags.alloc(1));
    DPCG_HIP(hipMemsetAsync(flags.p, 0, sizeof(int32_t), s));
    launch_check_lower(h->L, reinterpret_cast<int *>(flags.p), s);
    int32_t h_flags = 0, h_last = 0;
    DPCG_HIP(hipMemcpyAsync(&h_flags, flags.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipMemcpyAsync(&h_last, h->L.rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    if (h_flags & 1) return invalid("L: lower triangular with ascending columns and the diagonal stored last in every row");
    if (h_flags & 2) {
        set_error("L: non-positive diagonal");
        return DPCG_ERR_PIVOT;
    }
    if (h_last != h->L.nnz) return invalid("L: rowptr[n] != nnz");
    PhaseTimer pt(s);
    DPCG_TRY(transpose_lower(h->L, h->Lt, s));
    pt.mark("transpose");
    // factor index -> handle index: the factor's own numbering (fmap: multicolour IC(0)) or the caller's (iperm, if reordered)
    const int32_t *fm = h->fmap ? h->fmap : h->iperm;
    if (h->perm && mode == DPCG_PRECOND_LLT_MULTIPLY) {
        DPCG_TRY(permute_csr(h->L, h->perm, h->iperm, h->Lp, s));
        DPCG_TRY(permute_csr(h->Lt, h->perm, h->iperm, h->Ltp, s));
        DPCG_TRY(make_plan(h->Lp, h->planL, s));
        DPCG_TRY(make_plan(h->Ltp, h->planLt, s));
    } else if (mode == DPCG_PRECOND_LLT_MULTIPLY) {          // (the triangular solves run on their schedules, not on SpMV plans)
        DPCG_TRY(make_plan(h->L, h->planL, s));
        DPCG_TRY(make_plan(h->Lt, h->planLt, s));
    }
    pt.mark("plans");
    if (mode == DPCG_PRECOND_LLT_SOLVE) {
        LevelSort own, up_own;
        LevelSort &up = upper_levels ? *upper_levels : up_own;
        bool have_up = upper_levels != nullptr;
        // Large factors try the strip plan FIRST and build the level schedule (level-ordered copy, ring / sync-free records)
        // only when it is not kept; small ones build the schedule first because the choice depends on it.
        auto schedule = [&](Levels &lv, LevelSort &ls, const CsrDev &F, bool upper) -> int {
            // more than 2 % of the rows longer than a ring / strip record: no ring walk, no strips (see build_levels; measured on
            // a 48^3 grid with one extra lower neighbour on a share of the rows: 5 % such rows turn the strip plan's 263 us per
            // update into 772, the sync-free kernels take 458 whatever the share -- tools/longrow_share_probe.py)
            DevBuf<int32_t> n_long;
            DPCG_TRY(n_long.alloc(1));
            DPCG_HIP(hipMemsetAsync(n_long.p, 0, sizeof(int32_t), s));
            launch_count_long_rows(n, F.rowptr, 3, n_long.p, s);
            int32_t h_long = 0;
            DPCG_HIP(hipMemcpyAsync(&h_long, n_long.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            // (DPCG_LONG_ROW_PCT: development knob for that share, in per cent)
            static const int64_t pct = [] { const char *e = getenv("DPCG_LONG_ROW_PCT"); return e ? (int64_t)atoll(e) : (int64_t)2; }();
            const bool long_rows = (int64_t)h_long * 100 > n * pct;
            if (long_rows) {
                DPCG_TRY(build_levels(lv, ls, n, F.nnz, F.rowptr, F.col, F.val, s, fm, upper, true));
                return DPCG_OK;
            }
            const bool strips_first = n > 131072;
            if (strips_first) {
                lv.level_ptr = ls.level_ptr;
                lv.n_levels = (int)ls.level_ptr.size() - 1;
                DPCG_TRY(dev_alloc(&lv.spin_err, 1));
                DPCG_HIP(hipMemsetAsync(lv.spin_err, 0, sizeof(int), s));
                DPCG_TRY(build_strips(lv, n, F.nnz, F.rowptr, F.col, F.val, upper, fm, s));
                if (lv.strips.n_strips > 0) return DPCG_OK;
                dev_free(lv.spin_err);
            }
            DPCG_TRY(build_levels(lv, ls, n, F.nnz, F.rowptr, F.col, F.val, s, fm, upper));
            if (!strips_first) DPCG_TRY(build_strips(lv, n, F.nnz, F.rowptr, F.col, F.val, upper, fm, s));
            if (lv.strips.n_strips > 0) lv.level_major = false;
            return DPCG_OK;
        };
        if (lower_prebuilt) {                                 // (the handle owns the schedule from here on)
            h->lvlL = *lower_prebuilt;
            *lower_prebuilt = Levels();
            if (h->lvlL.strips.n_strips > 0) h->lvlL.n_levels = -1;
        } else {
            if (!lower_levels) {
                DPCG_TRY(compute_levels(n, h->L.rowptr, h->L.col, false, own, s, fm));
                lower_levels = &own;
                pt.mark("levels(L)");
            }
            // the level sets of L^T are those of L read backwards (taken before the schedule of L takes the arrays over)
            static const bool reverse_on = [] { const char *e = getenv("DPCG_REVERSE_LEVELS"); return !(e && e[0] == '0'); }();
            if (reverse_on) {
                DPCG_TRY(reversed_levels(n, *lower_levels, up, s));
                have_up = true;
            }
            DPCG_TRY(schedule(h->lvlL, *lower_levels, h->L, false));
            pt.mark("schedule(L)");
        }
        // L^T has as many levels as L (the longest dependency chain read backwards): when L took the strip plan, L^T tries it
        // straight away and its global level sets are only computed if that fails
        if (h->lvlL.strips.n_strips > 0) {
            h->lvlU.n_levels = h->lvlL.n_levels;
            DPCG_TRY(dev_alloc(&h->lvlU.spin_err, 1));
            DPCG_HIP(hipMemsetAsync(h->lvlU.spin_err, 0, sizeof(int), s));
            DPCG_TRY(build_strips(h->lvlU, n, h->Lt.nnz, h->Lt.rowptr, h->Lt.col, h->Lt.val, true, fm, s, h->lvlL.n_levels >= 0));
            if (h->lvlU.strips.n_strips == 0) dev_free(h->lvlU.spin_err);
        }
        if (h->lvlU.strips.n_strips == 0) {
            if (!have_up) {
                DPCG_TRY(compute_levels(n, h->Lt.rowptr, h->Lt.col, true, up, s, fm));
                pt.mark("levels(L^T)");
            }
            DPCG_TRY(schedule(h->lvlU, up, h->Lt, true));
        }
        pt.mark("schedule(L^T)");
        if (h->lvlL.level_major && h->lvlU.level_major) {      // the lower result feeds the upper solve without leaving level-major order
            DPCG_TRY(dev_alloc(&h->lvlU.lm_from_lower, n));
            launch_compose_positions(n, h->lvlU.rows, h->lvlL.lm_pos, h->lvlU.lm_from_lower, s);
            // colour sweeps: the first level of L^T holds the rows of L's last level (no dependants; equal counts: the same set)
            // -- the last lower sweep can open the upper solve (SptrsvIo::pair_out)
            const Levels &lo = h->lvlL, &up = h->lvlU;
            if (lo.sweep && up.sweep && lo.n_levels == up.n_levels && lo.n_levels >= 2 && lo.sweep_grid == up.sweep_grid &&
                up.level_ptr[1] - up.level_ptr[0] == lo.level_ptr[(size_t)lo.n_levels] - lo.level_ptr[(size_t)lo.n_levels - 1]) {
                static const bool pair_on = [] { const char *e = getenv("DPCG_SWEEP_PAIR"); return !(e && e[0] == '0'); }();
                if (pair_on) {
                    DPCG_TRY(dev_alloc(&h->lvlL.lm_to_upper, n));
                    launch_invert_positions(n, h->lvlU.lm_from_lower, h->lvlL.lm_to_upper, s);
                }
            }
            // ... and L's first level (diagonal only) can ride on the kernel that updates r (Levels::ride_diag)
            static const bool ride_on = [] { const char *e = getenv("DPCG_SWEEP_RIDE"); return !(e && e[0] == '0'); }();
            if (ride_on && lo.sweep && lo.n_levels >= 2) {
                DPCG_TRY(dev_alloc(&h->lvlL.ride_diag, n));
                DPCG_HIP(hipMemsetAsync(h->lvlL.ride_diag, 0, (size_t)n * sizeof(double), s));    // (only first-level rows are read)
                launch_scatter_f64(lo.level_ptr[1], lo.rows, lo.lo_val, h->lvlL.ride_diag, s);
            }
            DPCG_CHECK_LAUNCH();
        }
    }
    h->precond = mode;
    return DPCG_OK;
}
}  // namespace

extern "C" int dpcg_set_precond_llt(dpcg_handle_t h, int mode, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                    const double *val, int memspace, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    if (mode != DPCG_PRECOND_LLT_MULTIPLY && mode != DPCG_PRECOND_LLT_SOLVE) return invalid("bad LLT mode");
    if (nnz <= 0 || !rowptr || !col || !val) return invalid("dpcg_set_precond_llt: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);          // (the preconditioner being replaced may be in use on another stream)
    free_precond(h);
    // host arrays are uploaded, device arrays (a factor straight from the CNN) copied device-to-device: either way the
    // analysis below works on HBM-resident data
    DPCG_TRY(upload_csr(h->L, h->A.n, nnz, rowptr, col, val, DPCG_F64, memspace, 1, s));
    const int st = finish_llt(h, mode, s);
    if (st < 0) free_precond(h);
    return st;
}

// The values of a parked IC(0) in multicolour order again (dpcg_system::Parked), from the handle's CURRENT matrix values: the same
// tril(Q A Q^T), the same level-by-level factorisation in the same order as a full setup -- the same bits -- without the
// colouring, the permutation, the level analysis, the transposition, the schedules and their tile plans.  The entry maps are
// built at the first refresh by sending entry NUMBERS through the very kernels the setup sends values through.
namespace {
int refresh_parked_ic0(dpcg_system *h, hipStream_t s) {
    dpcg_system::Parked &P = h->parked;
    const int64_t n = h->A.n, nnzl = P.L.nnz;
    PhaseTimer pt(s);
    if (!P.map_al) {
        DevBuf<double> ids, lids;
        DevBuf<int32_t> lcol;
        DPCG_TRY(ids.alloc(h->A.nnz));
        DPCG_TRY(lids.alloc(nnzl));
        DPCG_TRY(lcol.alloc(nnzl));
        launch_iota_f64(h->A.nnz, ids.p, s);
        CsrDev Aid = h->A, Ac;                               // (borrowed pattern, entry numbers as values)
        Aid.val = ids.p;
        Aid.val32 = nullptr;
        Aid.owned = false;
        int st = permute_csr(Aid, h->mc_perm, h->mc_iperm, Ac, s);
        if (st < 0) {
            free_csr(Ac);
            return st;
        }
        launch_tril_copy(n, Ac.rowptr, Ac.col, Ac.val, P.L.rowptr, lcol.p, lids.p, s);
        st = dev_alloc(&P.map_al, nnzl);
        if (st >= 0) launch_f64_to_i32(nnzl, lids.p, P.map_al, s);
        free_csr(Ac);
        DPCG_TRY(st);
        // L^T's entries in terms of L's
        launch_iota_f64(nnzl, lids.p, s);
        CsrDev Lid = P.L, Ltid;
        Lid.val = lids.p;
        Lid.val32 = nullptr;
        Lid.owned = false;
        st = transpose_lower(Lid, Ltid, s);
        if (st >= 0) st = dev_alloc(&P.t_order, nnzl);
        if (st >= 0) launch_f64_to_i32(nnzl, Ltid.val, P.t_order, s);
        free_csr(Ltid);
        DPCG_TRY(st);
        // level-order position -> factor row (the schedules list handle indices)
        DPCG_TRY(dev_alloc(&P.rows_l, n));
        DPCG_TRY(dev_alloc(&P.rows_u, n));
        launch_compose_positions(n, P.lvlL.rows, h->mc_iperm, P.rows_l, s);
        launch_compose_positions(n, P.lvlU.rows, h->mc_iperm, P.rows_u, s);
        DPCG_HIP(hipStreamSynchronize(s));
        DPCG_CHECK_LAUNCH();
        pt.mark("refresh: entry maps");
    }
    DevBuf<int32_t> bad;
    DPCG_TRY(bad.alloc(1));
    DPCG_HIP(hipMemsetAsync(bad.p, 0, sizeof(int32_t), s));
    launch_gather_f64(nnzl, P.map_al, h->A.val, P.L.val, s);                   // tril(Q A Q^T)
    const std::vector<int32_t> &lp = P.lvlL.level_ptr;                          // the levels of tril(A)'s pattern are L's
    for (int l = 0; l + 1 < (int)lp.size(); ++l)
        launch_ic0_level(P.rows_l, lp[(size_t)l], lp[(size_t)l + 1] - lp[(size_t)l], P.L.rowptr, P.L.col, P.L.val, reinterpret_cast<int *>(bad.p),
                         s, nullptr, 0.0);
    launch_gather_f64(nnzl, P.t_order, P.L.val, P.Lt.val, s);
    launch_lo_values(n, P.rows_l, P.L.rowptr, P.L.val, P.lvlL.lo_rowptr, P.lvlL.lo_val, s);
    launch_lo_values(n, P.rows_u, P.Lt.rowptr, P.Lt.val, P.lvlU.lo_rowptr, P.lvlU.lo_val, s);
    if (P.lvlL.ride_diag) launch_scatter_f64(lp[1], P.lvlL.rows, P.lvlL.lo_val, P.lvlL.ride_diag, s);
    int32_t h_bad = 0;
    DPCG_HIP(hipMemcpyAsync(&h_bad, bad.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    DPCG_CHECK_LAUNCH();
    pt.mark("refresh: values");
    if (h_bad) {
        set_error("IC(0): non-positive pivot at row " + std::to_string(h_bad - 1));
        return DPCG_ERR_PIVOT;
    }
    free_precond(h, true);
    h->L = P.L;          P.L = CsrDev();
    h->Lt = P.Lt;        P.Lt = CsrDev();
    h->lvlL = P.lvlL;    P.lvlL = Levels();
    h->lvlU = P.lvlU;    P.lvlU = Levels();
    h->fmap = h->mc_perm;
    h->fmap_inv = h->mc_iperm;
    h->precond_colors = P.colors;
    h->precond = DPCG_PRECOND_LLT_SOLVE;
    P.valid = false;     // (the maps stay for the next round)
    return DPCG_OK;
}
}  // namespace

// IC(0) of A on its lower-triangular pattern (stands in for ilupp.ichol0, test.py:83), entirely on the device: tril(A)
// by count / scan / copy, the level sets of its pattern, then the numeric factorisation one launch per level
// (k_ic0_level) in the operation order of the CPU restatement (bit-identical factor).  The handle keeps its previous
// preconditioner when the factorisation fails.
extern "C" int dpcg_set_precond_ic0_ordered(dpcg_handle_t h, int mode, int ordering, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    if (mode != DPCG_PRECOND_LLT_MULTIPLY && mode != DPCG_PRECOND_LLT_SOLVE) return invalid("bad LLT mode");
    if (ordering != DPCG_ORDER_CALLER && ordering != DPCG_ORDER_MULTICOLOR) return invalid("dpcg_set_precond_ic0_ordered: bad ordering");
    if (ordering == DPCG_ORDER_MULTICOLOR && mode != DPCG_PRECOND_LLT_SOLVE)
        return invalid("dpcg_set_precond_ic0_ordered: the multicolour ordering serves the triangular solves (mode LLT_SOLVE)");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);          // (the preconditioner being replaced may be in use on another stream)
    const int64_t n = h->A.n;
    if (h->parked.valid) {              // new values on the pattern of a parked factor: only the values are computed again
        static const bool keep = [] { const char *ev = getenv("DPCG_KEEP_COLORING"); return !(ev && ev[0] == '0'); }();
        if (ordering == DPCG_ORDER_MULTICOLOR && mode == DPCG_PRECOND_LLT_SOLVE && keep && h->mc_perm) {
            const int st = refresh_parked_ic0(h, s);
            if (st < 0) free_parked(h);
            return st;
        }
    }
    // DPCG_ORDER_MULTICOLOR: IC(0) of Q A Q^T, Q = the handle's matrix colour by colour (dpcg_reorder.hip: multicolor_order).
    // The permuted matrix is a temporary; the factor stays in that numbering and is addressed through fmap.
    CsrDev Ac;
    int32_t *cperm = nullptr, *ciperm = nullptr;
    int n_colors = 0;
    PhaseTimer pt0(s);
    if (ordering == DPCG_ORDER_MULTICOLOR) {
        // the colouring looks at the pattern only: computed once per handle numbering (kept across dpcg_update_values)
        static const bool keep_coloring = [] { const char *ev = getenv("DPCG_KEEP_COLORING"); return !(ev && ev[0] == '0'); }();
        if (!h->mc_perm || !keep_coloring) {
            int32_t *p = nullptr, *ip = nullptr;
            int nc = 0;
            DPCG_TRY(multicolor_order(h->A, &p, &ip, &nc, s));
            if (h->fmap == h->mc_perm) {       // the attached factor keeps the arrays it was built with until it is replaced
                h->mc_perm = h->mc_iperm = nullptr;
            } else {
                dev_free(h->mc_perm);
                dev_free(h->mc_iperm);
            }
            h->mc_perm = p;
            h->mc_iperm = ip;
            h->mc_colors = nc;
        }
        cperm = h->mc_perm;
        ciperm = h->mc_iperm;
        n_colors = h->mc_colors;
        pt0.mark("multicolour ordering");
        const int stp = permute_csr(h->A, cperm, ciperm, Ac, s);
        pt0.mark("Q A Q^T");
        if (stp < 0) {
            free_csr(Ac);
            return stp;
        }
    }
    // otherwise the factor of the CALLER's matrix (what ilupp.ichol0 would be handed), also when the handle iterates on P A P^T
    const CsrDev &Asrc = ordering == DPCG_ORDER_MULTICOLOR ? Ac : (h->perm ? h->A_user : h->A);
    CsrDev Lf;
    Lf.n = n;
    Lf.owned = true;
    DevBuf<int32_t> cnt, flags;
    auto fail = [&](int st) {
        free_csr(Lf);
        free_csr(Ac);
        return st;
    };
    int st = DPCG_OK;
    if ((st = cnt.alloc(n + 1)) < 0 || (st = flags.alloc(2)) < 0 || (st = dev_alloc(&Lf.rowptr, n + 1)) < 0) return fail(st);
    hipError_t e = hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), s);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
    launch_tril_count(n, Asrc.rowptr, Asrc.col, cnt.p, reinterpret_cast<int *>(flags.p), s);
    if ((st = exclusive_scan_i32(cnt.p, Lf.rowptr, n + 1, s)) < 0) return fail(st);
    int32_t h_flags[2] = {0, 0}, lnnz = 0;
    e = hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&lnnz, Lf.rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(hip_fail(e, "IC(0): tril(A)", __FILE__, __LINE__));
    if (h_flags[0]) {
        set_error("IC(0): missing diagonal entry");
        return fail(DPCG_ERR_PIVOT);
    }
    Lf.nnz = lnnz;
    if ((st = dev_alloc(&Lf.col, lnnz)) < 0 || (st = dev_alloc(&Lf.val, lnnz)) < 0) return fail(st);
    PhaseTimer pt(s);
    launch_tril_copy(n, Asrc.rowptr, Asrc.col, Asrc.val, Lf.rowptr, Lf.col, Lf.val, s);
    pt.mark("tril(A)");
    // A pattern without cross terms whose schedule is a walk of many narrow levels (natural-order / RCM-ordered grids) is factored
    // THROUGH that schedule, built on the pattern of tril(A) -- one launch -- and the schedule then is L's:
    //   * n > 131 072, banded: the strip plan (measured at 1024^2, 2047 levels: 10.9 ms of level-by-level launches and 3.6 ms
    //     of level analysis before);
    //   * smaller, one LDS-ring segment (2-D grids up to 362^2): the one-workgroup ring walk (256^2: 511 launches, 2.3 ms).
    Levels pre;                       // the schedule built on tril(A); becomes the handle's when kept
    auto fail2 = [&](int st2) {
        free_levels(pre);
        return fail(st2);
    };
    bool through_schedule = false, plain = false, short_rows = false;   // plain: no cross terms, rows of <= 3 off-diagonal entries
    static const bool schedule_factor_on = [] { const char *ev = getenv("DPCG_IC0_STRIPS"); return !(ev && ev[0] == '0'); }();
    DevBuf<double> diag, fac;
    // records of L by position -> CSR values + the schedule's own copies; pivot check; the records' value array is swapped in
    auto harvest = [&](const int32_t *frows, const int32_t *lo_rp, double *lo_val, double **rec_val, const char *what) -> int {
        launch_strip_factor_scatter(n, frows, Lf.rowptr, fac.p, Lf.val, lo_rp, lo_val, reinterpret_cast<int *>(flags.p) + 1, s);
        int32_t h_spin = 0;
        hipError_t e2 = hipGetLastError();
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&h_spin, pre.spin_err, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);
        if (e2 != hipSuccess) return hip_fail(e2, what, __FILE__, __LINE__);
        if (h_spin) {
            set_error("IC(0) through the schedule: a bounded wait ran out");
            return DPCG_ERR_HIP;
        }
        if (h_flags[1]) h_flags[1] = (0x7fffffff - h_flags[1]) + 1;      // (row + 1, as the level kernels report it)
        std::swap(*rec_val, fac.p);
        return DPCG_OK;
    };
    if (schedule_factor_on && n >= 4096 && ordering == DPCG_ORDER_CALLER) {
        e = hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), s);
        if (e != hipSuccess) return fail(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
        launch_ic0_cross_terms(n, Lf.rowptr, Lf.col, reinterpret_cast<int *>(flags.p), s);
        e = hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(hip_fail(e, "IC(0): pattern check", __FILE__, __LINE__));
        plain = h_flags[0] == 0;
        short_rows = (h_flags[0] & 2) == 0;
        e = hipMemsetAsync(flags.p, 0, 2 * sizeof(int32_t), s);
        if (e != hipSuccess) return fail(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
        h_flags[0] = h_flags[1] = 0;
    }
    if (short_rows && n > 131072) {
        DevBuf<int32_t> frows, xdesc;
        DevBuf<double> offd;
        if ((st = dev_alloc(&pre.spin_err, 1)) < 0) return fail2(st);
        e = hipMemsetAsync(pre.spin_err, 0, sizeof(int), s);
        if (e != hipSuccess) return fail2(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
        if ((st = build_strips(pre, n, Lf.nnz, Lf.rowptr, Lf.col, Lf.val, false, h->iperm, s, false, &frows)) < 0) return fail2(st);
        bool launched = false;
        if (pre.strips.n_strips > 0 && pre.strips.long_rows == 0) {
            pt.mark("strip plan (tril A)");
            if ((st = diag.alloc(n)) < 0 || (st = fac.alloc(4 * n)) < 0) return fail2(st);
            if (!plain) {                                   // cross terms: the general form of the walk
                if ((st = xdesc.alloc(n)) < 0 || (st = offd.alloc(3 * n)) < 0) return fail2(st);
                launch_ring_factor_desc(n, pre.strips.lo_rowptr, pre.strips.lo_col, pre.strips.lo_cpos, nullptr, 0.0, xdesc.p, nullptr, s);
            }
            launched = launch_strip_factor(pre, diag.p, fac.p, n, s, plain ? nullptr : xdesc.p, nullptr, plain ? nullptr : offd.p);
        }
        if (launched) {
            if ((st = harvest(frows.p, pre.strips.lo_rowptr, pre.strips.lo_val, &pre.strips.val, "IC(0) through the strip plan")) < 0)
                return fail2(st);
            through_schedule = true;
            pt.mark("numeric IC(0) by strips");
        } else {
            free_levels(pre);
        }
    }
    LevelSort ls_first, ls_again, up_from_lower;
    LevelSort *ls = &ls_first;
    bool have_upper = false;
    if (!through_schedule) {
        if ((st = compute_levels(n, Lf.rowptr, Lf.col, false, *ls, s, cperm ? cperm : h->iperm)) < 0) return fail2(st);
        pt.mark("levels(tril A)");
        const int nl = (int)ls->level_ptr.size() - 1;
        if (short_rows && !h->perm && n <= 131072 && nl >= 64) {
            // (build_levels takes the level sets over: if the schedule turns out not to be one ring walk, they are computed again)
            if (mode == DPCG_PRECOND_LLT_SOLVE && (st = reversed_levels(n, *ls, up_from_lower, s)) < 0) return fail2(st);
            if ((st = build_levels(pre, *ls, n, Lf.nnz, Lf.rowptr, Lf.col, Lf.val, s, nullptr, false)) < 0) return fail2(st);
            if ((st = diag.alloc(n)) < 0 || (st = fac.alloc(4 * n)) < 0) return fail2(st);
            DevBuf<int32_t> xdesc;                          // a pattern with cross terms: the general form of the walk
            if (!plain) {
                if ((st = xdesc.alloc(n)) < 0) return fail2(st);
                launch_ring_factor_desc(n, pre.lo_rowptr, pre.lo_col, pre.lo_cpos, nullptr, 0.0, xdesc.p, nullptr, s);
            }
            if (launch_ring_factor(pre, diag.p, fac.p, s, plain ? nullptr : xdesc.p, nullptr)) {
                if ((st = harvest(pre.rows, pre.lo_rowptr, pre.lo_val, &pre.pk_val, "IC(0) through the ring walk")) < 0) return fail2(st);
                through_schedule = true;
                have_upper = mode == DPCG_PRECOND_LLT_SOLVE;
                pt.mark("numeric IC(0) by the ring walk");
            } else {
                free_levels(pre);
                ls = &ls_again;
                if ((st = compute_levels(n, Lf.rowptr, Lf.col, false, *ls, s, cperm ? cperm : h->iperm)) < 0) return fail2(st);
            }
        }
    }
    if (!through_schedule) {
        if ((st = numeric_incomplete_cholesky(*ls, n, Lf, reinterpret_cast<int *>(flags.p) + 1, nullptr, 0.0, s)) < 0) return fail2(st);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail2(hip_fail(e, "IC(0): numeric factorisation", __FILE__, __LINE__));
        pt.mark("numeric IC(0)");
    }
    if (h_flags[1]) {
        set_error("IC(0): non-positive pivot at row " + std::to_string(h_flags[1] - 1));
        return fail2(DPCG_ERR_PIVOT);
    }
    free_precond(h);
    free_csr(Ac);
    h->L = Lf;
    h->fmap = cperm;                  // factor index -> handle index (null: the caller's numbering)
    h->fmap_inv = ciperm;
    h->precond_colors = n_colors;
    const bool keep_schedule = through_schedule && mode == DPCG_PRECOND_LLT_SOLVE;
    st = finish_llt(h, mode, s, through_schedule ? nullptr : ls, keep_schedule ? &pre : nullptr, have_upper ? &up_from_lower : nullptr);
    free_levels(pre);                 // (multiply mode: the schedule only served the factorisation)
    if (st < 0) free_precond(h);
    return st;
}

// The number of levels of a factor whose schedule never needed them (strip plans built before the level analysis): counted
// on the first request, for dpcg_get_info.
int count_levels_on_demand(dpcg_system *h) {
    if (h->precond != DPCG_PRECOND_LLT_SOLVE || h->lvlL.n_levels >= 0) return DPCG_OK;
    const int32_t *fm = h->fmap ? h->fmap : h->iperm;
    SetupScope scope(nullptr);
    LevelSort ls;
    DPCG_TRY(compute_levels(h->A.n, h->L.rowptr, h->L.col, false, ls, nullptr, fm));
    h->lvlL.n_levels = (int)ls.level_ptr.size() - 1;
    h->lvlU.n_levels = h->lvlL.n_levels;   // (the longest dependency chain of L, read backwards)
    return DPCG_OK;
}

extern "C" int dpcg_set_precond_ic0(dpcg_handle_t h, int mode, dpcg_stream_t stream) {
    return dpcg_set_precond_ic0_ordered(h, mode, DPCG_ORDER_CALLER, stream);
}

// perm_host[k] = the CALLER's row that sits at position k of the factor's numbering (identity for DPCG_ORDER_CALLER)
extern "C" int dpcg_get_precond_ordering(dpcg_handle_t h, int *n_colors, int32_t *perm_host) {
    if (!h) return invalid("NULL handle");
    if (n_colors) *n_colors = h->precond_colors;
    if (!perm_host) return DPCG_OK;
    const int64_t n = h->A.n;
    if (!h->fmap) {
        for (int64_t k = 0; k < n; ++k) perm_host[k] = (int32_t)k;
        return DPCG_OK;
    }
    DPCG_HIP(hipMemcpy(perm_host, h->fmap, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));    // handle indices
    if (h->perm) {                                                                                  // -> caller's rows
        std::vector<int32_t> p((size_t)n);
        DPCG_HIP(hipMemcpy(p.data(), h->perm, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < n; ++k) perm_host[k] = p[(size_t)perm_host[k]];
    }
    return DPCG_OK;
}

// ICT -- thresholded incomplete Cholesky with level-1 fill (contract: oracle/oracle.py::ict; stands in for
// ilupp.icholt(add_fill_in=1, threshold=0.1), the harness's default technique, test.py:81-88; ilupp absent: unpinned).
// Symbolic phase (pattern with level-1 fill), level sets, numeric phase (the IC(0) kernel with the drop rule) and the
// compaction all run on the device; like IC(0) it factors the CALLER's matrix and leaves the previous preconditioner
// in place when it fails.
extern "C" int dpcg_set_precond_ict(dpcg_handle_t h, int mode, int fill_in, double threshold, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    if (mode != DPCG_PRECOND_LLT_MULTIPLY && mode != DPCG_PRECOND_LLT_SOLVE) return invalid("bad LLT mode");
    if (fill_in < 0 || !(threshold >= 0.0)) return invalid("dpcg_set_precond_ict: fill_in >= 0 and threshold >= 0");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);          // (the preconditioner being replaced may be in use on another stream)
    const int64_t n = h->A.n;
    const CsrDev &Asrc = h->perm ? h->A_user : h->A;
    CsrDev S, Lf;                       // S: the pattern with fill (values: A, then the factor with stored zeros)
    S.n = Lf.n = n;
    S.owned = Lf.owned = true;
    DevBuf<int32_t> cnt, flags;
    DevBuf<double> colnorm;
    auto fail = [&](int st) {
        free_csr(S);
        free_csr(Lf);
        return st;
    };
    int st = DPCG_OK;
    if ((st = cnt.alloc(n + 1)) < 0 || (st = flags.alloc(4)) < 0 || (st = colnorm.alloc(n)) < 0 ||
        (st = dev_alloc(&S.rowptr, n + 1)) < 0 || (st = dev_alloc(&Lf.rowptr, n + 1)) < 0)
        return fail(st);
    // flags: [0] symbolic phase, [1] pivot (row + 1) of the level kernels, [2] pattern check, [3] pivot of the ring walk
    hipError_t e = hipMemsetAsync(flags.p, 0, 4 * sizeof(int32_t), s);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
    PhaseTimer pt(s);
    launch_colnorm1(n, Asrc.rowptr, Asrc.col, Asrc.val, colnorm.p, s);
    launch_ict_pattern(false, n, Asrc.rowptr, Asrc.col, Asrc.val, fill_in > 0 ? 1 : 0, cnt.p, nullptr, nullptr, nullptr,
                       reinterpret_cast<int *>(flags.p), s);
    if ((st = exclusive_scan_i32(cnt.p, S.rowptr, n + 1, s)) < 0) return fail(st);
    int32_t h_flags[2] = {0, 0}, snnz = 0;
    e = hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&snnz, S.rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail(hip_fail(e, "ICT: symbolic phase", __FILE__, __LINE__));
    if (h_flags[0] & 1) {
        set_error("ICT: missing diagonal entry");
        return fail(DPCG_ERR_PIVOT);
    }
    if (h_flags[0] & 4) {
        set_error("ICT: a row of tril(A) has more than 192 entries (use IC(0) for such matrices)");
        return fail(DPCG_ERR_INVALID);
    }
    S.nnz = snnz;
    if ((st = dev_alloc(&S.col, snnz)) < 0 || (st = dev_alloc(&S.val, snnz)) < 0) return fail(st);
    launch_ict_pattern(true, n, Asrc.rowptr, Asrc.col, Asrc.val, fill_in > 0 ? 1 : 0, nullptr, S.rowptr, S.col, S.val,
                       reinterpret_cast<int *>(flags.p), s);
    LevelSort ls_first, ls_again;
    LevelSort *ls = &ls_first;
    pt.mark("ICT: pattern with fill");
    bool through_ring = false;       // (the numeric phase ran through a schedule: the ring walk or the strips)
    static const bool ring_factor_on = [] { const char *ev = getenv("DPCG_IC0_STRIPS"); return !(ev && ev[0] == '0'); }();
    int32_t h_pat = 2;               // pattern check: bit 1 = a row with more than three off-diagonal entries
    if (ring_factor_on && n >= 4096) {
        launch_ic0_cross_terms(n, S.rowptr, S.col, reinterpret_cast<int *>(flags.p) + 2, s);
        e = hipMemcpyAsync(&h_pat, flags.p + 2, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(hip_fail(e, "ICT: pattern check", __FILE__, __LINE__));
    }
    // Beyond 131 072 rows a banded pattern takes the strip walk (general form), before any level analysis
    if (!(h_pat & 2) && n > 131072) {
        Levels tmp;
        DevBuf<int32_t> frows, xdesc;
        DevBuf<double> thr, diag, fac, offd;
        auto fail_tmp = [&](int st2) {
            free_levels(tmp);
            return fail(st2);
        };
        if ((st = dev_alloc(&tmp.spin_err, 1)) < 0) return fail_tmp(st);
        e = hipMemsetAsync(tmp.spin_err, 0, sizeof(int), s);
        if (e != hipSuccess) return fail_tmp(hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__));
        if ((st = build_strips(tmp, n, S.nnz, S.rowptr, S.col, S.val, false, nullptr, s, false, &frows)) < 0) return fail_tmp(st);
        if (tmp.strips.n_strips > 0 && tmp.strips.long_rows == 0) {
            if ((st = xdesc.alloc(n)) < 0 || (st = thr.alloc(4 * n)) < 0 || (st = diag.alloc(n)) < 0 || (st = fac.alloc(4 * n)) < 0 ||
                (st = offd.alloc(3 * n)) < 0)
                return fail_tmp(st);
            launch_ring_factor_desc(n, tmp.strips.lo_rowptr, tmp.strips.lo_col, tmp.strips.lo_cpos, colnorm.p, threshold, xdesc.p, thr.p, s);
            if (launch_strip_factor(tmp, diag.p, fac.p, n, s, xdesc.p, thr.p, offd.p)) {
                launch_strip_factor_scatter(n, frows.p, S.rowptr, fac.p, S.val, tmp.strips.lo_rowptr, tmp.strips.lo_val,
                                            reinterpret_cast<int *>(flags.p) + 3, s);
                int32_t h_spin = 0;
                e = hipMemcpyAsync(&h_spin, tmp.spin_err, sizeof(int), hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipStreamSynchronize(s);
                if (e != hipSuccess) return fail_tmp(hip_fail(e, "ICT through the strip plan", __FILE__, __LINE__));
                if (h_spin) {
                    set_error("ICT through the strip plan: a bounded wait ran out");
                    return fail_tmp(DPCG_ERR_HIP);
                }
                through_ring = true;
                pt.mark("ICT: numeric (strips)");
            }
        }
        free_levels(tmp);
    }
    if (!through_ring) {
        if ((st = compute_levels(n, S.rowptr, S.col, false, *ls, s)) < 0) return fail(st);
        pt.mark("ICT: levels");
    }
    // A C2-size factor whose rows hold at most three off-diagonal entries (level-1 fill on a 5-point grid) and whose schedule is ONE
    // LDS-ring segment is factored by the one-workgroup walk -- cross terms and the drop rule included (k_sptrsv_ring_pipe, FACTOR = 2)
    // -- instead of one launch per level (256^2: 766 launches, 4.8 of the setup's 6.2 ms).  The schedule built for that is a
    // temporary: dropping changes the pattern, and L gets its own below.
    if (!through_ring && n <= 131072 && (int)ls->level_ptr.size() - 1 >= 64) {
        if (!(h_pat & 2)) {
            Levels tmp;
            DevBuf<int32_t> xdesc;
            DevBuf<double> thr, diag, fac;
            auto fail_tmp = [&](int st2) {
                free_levels(tmp);
                return fail(st2);
            };
            if ((st = build_levels(tmp, *ls, n, S.nnz, S.rowptr, S.col, S.val, s, nullptr, false)) < 0) return fail_tmp(st);
            if ((st = xdesc.alloc(n)) < 0 || (st = thr.alloc(4 * n)) < 0 || (st = diag.alloc(n)) < 0 || (st = fac.alloc(4 * n)) < 0)
                return fail_tmp(st);
            launch_ring_factor_desc(n, tmp.lo_rowptr, tmp.lo_col, tmp.lo_cpos, colnorm.p, threshold, xdesc.p, thr.p, s);
            if (launch_ring_factor(tmp, diag.p, fac.p, s, xdesc.p, thr.p)) {
                launch_strip_factor_scatter(n, tmp.rows, S.rowptr, fac.p, S.val, tmp.lo_rowptr, tmp.lo_val,
                                            reinterpret_cast<int *>(flags.p) + 3, s);
                through_ring = true;
            }
            free_levels(tmp);
            if (!through_ring) {                                   // (build_levels took the level sets over)
                ls = &ls_again;
                if ((st = compute_levels(n, S.rowptr, S.col, false, *ls, s)) < 0) return fail(st);
            }
        }
    }
    if (!through_ring &&
        (st = numeric_incomplete_cholesky(*ls, n, S, reinterpret_cast<int *>(flags.p) + 1, colnorm.p, threshold, s)) < 0)
        return fail(st);
    // compaction: the dropped entries are stored zeros
    pt.mark(through_ring ? "ICT: numeric (through the schedule)" : "ICT: numeric (launch per level)");
    launch_count_kept(n, S.rowptr, S.col, S.val, cnt.p, s);
    if ((st = exclusive_scan_i32(cnt.p, Lf.rowptr, n + 1, s)) < 0) return fail(st);
    int32_t lnnz = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&lnnz, Lf.rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int32_t h_ring_pivot = 0;
    if (e == hipSuccess && through_ring) {
        e = hipMemcpyAsync(&h_ring_pivot, flags.p + 3, sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess) return fail(hip_fail(e, "ICT: numeric factorisation", __FILE__, __LINE__));
    if (h_ring_pivot) h_flags[1] = (0x7fffffff - h_ring_pivot) + 1;          // (row + 1, as the level kernels report it)
    if (h_flags[1]) {
        set_error("ICT: non-positive pivot at row " + std::to_string(h_flags[1] - 1));
        return fail(DPCG_ERR_PIVOT);
    }
    Lf.nnz = lnnz;
    if ((st = dev_alloc(&Lf.col, lnnz)) < 0 || (st = dev_alloc(&Lf.val, lnnz)) < 0) return fail(st);
    launch_copy_kept(n, S.rowptr, S.col, S.val, Lf.rowptr, Lf.col, Lf.val, s);
    e = hipStreamSynchronize(s);
    free_csr(S);
    if (e != hipSuccess) return fail(hip_fail(e, "ICT: compaction", __FILE__, __LINE__));
    free_precond(h);
    h->L = Lf;
    pt.mark("ICT: compaction");
    st = finish_llt(h, mode, s);
    if (st < 0) free_precond(h);
    return st;
}

// ilupp.icholt as ILU++ defines it (contract: oracle/oracle.py::icholt; device routine: dpcg_icholt.hip).  Like IC(0) / ICT it
// factors the CALLER's matrix and leaves the previous preconditioner in place when it fails.
extern "C" int dpcg_set_precond_icholt(dpcg_handle_t h, int mode, int add_fill_in, double threshold, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    if (mode != DPCG_PRECOND_LLT_MULTIPLY && mode != DPCG_PRECOND_LLT_SOLVE) return invalid("bad LLT mode");
    if (add_fill_in < 0 || !(threshold >= 0.0)) return invalid("dpcg_set_precond_icholt: add_fill_in >= 0 and threshold >= 0");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);          // (the preconditioner being replaced may be in use on another stream)
    const CsrDev &Asrc = h->perm ? h->A_user : h->A;
    PhaseTimer pt(s);
    CsrDev Lf;
    int st = icholt_factor(Asrc, add_fill_in, threshold, Lf, s);
    if (st < 0) return st;
    pt.mark("icholt: columns");
    free_precond(h);
    h->L = Lf;
    st = finish_llt(h, mode, s);
    if (st < 0) free_precond(h);
    return st;
}

extern "C" int dpcg_get_factor(dpcg_handle_t h, int32_t *rowptr, int32_t *col, double *val) {
    if (!h) return invalid("NULL handle");
    if (h->precond != DPCG_PRECOND_LLT_MULTIPLY && h->precond != DPCG_PRECOND_LLT_SOLVE) {
        set_error("dpcg_get_factor: no L factor set");
        return DPCG_ERR_STATE;
    }
    DPCG_HIP(hipMemcpy(rowptr, h->L.rowptr, (size_t)(h->L.n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    DPCG_HIP(hipMemcpy(col, h->L.col, (size_t)h->L.nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    DPCG_HIP(hipMemcpy(val, h->L.val, (size_t)h->L.nnz * sizeof(double), hipMemcpyDeviceToHost));
    return DPCG_OK;
}
