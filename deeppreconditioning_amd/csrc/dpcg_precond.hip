// Host side of libdpcg.so, part 2: preconditioner setup -- Jacobi, explicit M, L L^T in multiply or solve mode (transpose,
// level sets, level-ordered copies, ring-segment records) and IC(0) (symbolic part on the host, numeric on the device).
#include <algorithm>
#include <cstring>

#include "dpcg_host.h"

// ------------------------------------------------------------------------------------------------
// preconditioners
// ------------------------------------------------------------------------------------------------
extern "C" int dpcg_set_precond_none(dpcg_handle_t h) {
    if (!h) return invalid("NULL handle");
    free_precond(h);
    return DPCG_OK;
}

extern "C" int dpcg_set_precond_jacobi(dpcg_handle_t h, const double *dinv, int memspace, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    hipStream_t s = (hipStream_t)stream;
    free_precond(h);
    DPCG_TRY(dev_alloc(&h->dinv, h->A.n));
    if (dinv) {
        DPCG_HIP(hipMemcpyAsync(h->dinv, dinv, (size_t)h->A.n * sizeof(double),
                                memspace == DPCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
        DPCG_HIP(hipStreamSynchronize(s));
    } else {
        int *d_bad = nullptr, h_bad = 0;
        DPCG_TRY(dev_alloc(&d_bad, 1));
        DPCG_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), s));
        launch_extract_dinv(h->A, h->dinv, d_bad, s);
        DPCG_HIP(hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        dev_free(d_bad);
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
    free_precond(h);
    DPCG_TRY(upload_csr(h->M, h->A.n, nnz, rowptr, col, val, DPCG_F64, memspace, 1, s));
    DPCG_TRY(make_plan(h->M, h->planM, s));
    h->precond = DPCG_PRECOND_CSR;
    return DPCG_OK;
}

// Level sets of a triangular CSR factor on the host (setup): level(i) = 1 + max level of the rows
// it depends on.  Rows are then grouped by level; runs of narrow levels become one merged segment.
static void build_levels_host(int64_t n, const std::vector<int32_t> &rp, const std::vector<int32_t> &ci, bool upper,
                              std::vector<int32_t> &rows_sorted, std::vector<int32_t> &level_ptr) {
    std::vector<int32_t> level((size_t)n, 0);
    int32_t max_level = 0;
    if (!upper) {
        for (int64_t i = 0; i < n; ++i) {
            int32_t l = 0;
            for (int32_t k = rp[i]; k < rp[i + 1] - 1; ++k) l = std::max(l, level[ci[k]] + 1);
            level[i] = l;
            max_level = std::max(max_level, l);
        }
    } else {
        for (int64_t i = n - 1; i >= 0; --i) {
            int32_t l = 0;
            for (int32_t k = rp[i] + 1; k < rp[i + 1]; ++k) l = std::max(l, level[ci[k]] + 1);
            level[i] = l;
            max_level = std::max(max_level, l);
        }
    }
    const int nl = max_level + 1;
    level_ptr.assign((size_t)nl + 1, 0);
    for (int64_t i = 0; i < n; ++i) level_ptr[level[i] + 1]++;
    for (int l = 0; l < nl; ++l) level_ptr[l + 1] += level_ptr[l];
    rows_sorted.resize((size_t)n);
    std::vector<int32_t> cursor(level_ptr.begin(), level_ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) rows_sorted[cursor[level[i]]++] = (int32_t)i;
}

static int upload_levels(Levels &lv, const std::vector<int32_t> &rows_sorted, const std::vector<int32_t> &level_ptr,
                         const int32_t *rp, const int32_t *ci, const double *v, hipStream_t s) {
    constexpr int kMergeMax = 2048;  // levels this narrow are walked by one 1024-thread workgroup
    lv.level_ptr = level_ptr;
    lv.n_levels = (int)level_ptr.size() - 1;
    // level-ordered copy of the factor (row j = original row rows_sorted[j])
    const int64_t n = (int64_t)rows_sorted.size();
    std::vector<int32_t> lo_rp((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; ++j) lo_rp[j + 1] = lo_rp[j] + (rp[rows_sorted[j] + 1] - rp[rows_sorted[j]]);
    const int64_t nnz = lo_rp[n];
    std::vector<int32_t> lo_ci((size_t)nnz);
    std::vector<double> lo_v((size_t)nnz);
    for (int64_t j = 0; j < n; ++j) {
        const int32_t src = rp[rows_sorted[j]], len = rp[rows_sorted[j] + 1] - src, dst = lo_rp[j];
        std::copy(ci + src, ci + src + len, lo_ci.begin() + dst);
        std::copy(v + src, v + src + len, lo_v.begin() + dst);
    }
    lv.stream_ok = true;
    for (int l = 0; l < lv.n_levels && lv.stream_ok; ++l)
        for (int32_t jb = level_ptr[l]; jb < level_ptr[l + 1]; jb += kStreamRows) {
            const int32_t je = std::min<int32_t>(jb + kStreamRows, level_ptr[l + 1]);
            if (lo_rp[je] - lo_rp[jb] > kStreamCap) { lv.stream_ok = false; break; }
        }
    // level-order position of every entry's column
    std::vector<int32_t> pos((size_t)n), lo_cp((size_t)nnz);
    for (int64_t j = 0; j < n; ++j) pos[rows_sorted[j]] = (int32_t)j;
    for (int64_t k = 0; k < nnz; ++k) lo_cp[k] = pos[lo_ci[k]];
    DPCG_TRY(dev_alloc(&lv.lo_rowptr, n + 1));
    DPCG_TRY(dev_alloc(&lv.lo_col, nnz));
    DPCG_TRY(dev_alloc(&lv.lo_cpos, nnz));
    DPCG_TRY(dev_alloc(&lv.lo_val, nnz));
    DPCG_HIP(hipMemcpyAsync(lv.lo_cpos, lo_cp.data(), lo_cp.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DPCG_HIP(hipMemcpyAsync(lv.lo_rowptr, lo_rp.data(), lo_rp.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DPCG_HIP(hipMemcpyAsync(lv.lo_col, lo_ci.data(), lo_ci.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DPCG_HIP(hipMemcpyAsync(lv.lo_val, lo_v.data(), lo_v.size() * sizeof(double), hipMemcpyHostToDevice, s));
    DPCG_TRY(dev_alloc(&lv.rows, (int64_t)rows_sorted.size()));
    DPCG_TRY(dev_alloc(&lv.level_ptr_dev, (int64_t)level_ptr.size()));
    DPCG_HIP(hipMemcpyAsync(lv.rows, rows_sorted.data(), rows_sorted.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DPCG_HIP(hipMemcpyAsync(lv.level_ptr_dev, level_ptr.data(), level_ptr.size() * sizeof(int32_t),
                            hipMemcpyHostToDevice, s));
    DPCG_HIP(hipStreamSynchronize(s));
    lv.segments.clear();
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
            // LDS ring: in level order, how far back do this segment's rows reach (within the segment)?
            const int32_t seg_start = level_ptr[l];
            int64_t maxdist = 0, width = 0;
            for (int q = l; q < e; ++q) width = std::max<int64_t>(width, level_ptr[q + 1] - level_ptr[q]);
            for (int32_t j = seg_start; j < level_ptr[e]; ++j)
                for (int32_t k = lo_rp[j]; k < lo_rp[j + 1]; ++k)
                    if (lo_cp[k] >= seg_start && lo_cp[k] < j) maxdist = std::max<int64_t>(maxdist, j - lo_cp[k]);
            int64_t w = 64;
            while (w < maxdist + width + 1) w *= 2;
            if (w <= 8192) seg.ring_w = (int)w;            // 64 KiB of LDS at most
        }
        lv.segments.push_back(seg);
        l = e;
    }
    // fixed-width row records for the ring segments (see Levels::pk_meta)
    bool any_ring = false;
    for (const auto &seg : lv.segments) any_ring = any_ring || seg.ring_w > 0;
    if (any_ring) {
        std::vector<int32_t> meta((size_t)n * 4, -1);
        std::vector<double> pv((size_t)n * 4, 0.0);
        for (const auto &seg : lv.segments) {
            if (seg.ring_w <= 0) continue;
            const int32_t seg_start = level_ptr[seg.lo];
            for (int32_t j = seg_start; j < level_ptr[seg.hi]; ++j) {
                const int32_t a = lo_rp[j], b = lo_rp[j + 1], row = rows_sorted[j];
                // the diagonal is the first entry of a row of L^T and the last of a row of L
                const bool diag_first = lo_ci[a] == row && (b - a == 1 || lo_ci[b - 1] != row);
                const int32_t ks = diag_first ? a + 1 : a, ke = diag_first ? b : b - 1;
                meta[(size_t)j * 4 + 3] = row;
                pv[(size_t)j * 4 + 3] = lo_v[diag_first ? a : b - 1];
                bool fast = ke - ks <= 3;
                for (int32_t k = ks; k < ke && fast; ++k) fast = lo_cp[k] >= seg_start;
                if (!fast) {
                    meta[(size_t)j * 4] = -2;
                    continue;
                }
                for (int32_t k = ks; k < ke; ++k) {
                    meta[(size_t)j * 4 + (k - ks)] = lo_cp[k];
                    pv[(size_t)j * 4 + (k - ks)] = lo_v[k];
                }
            }
        }
        DPCG_TRY(dev_alloc(&lv.pk_meta, n * 4));
        DPCG_TRY(dev_alloc(&lv.pk_val, n * 4));
        DPCG_TRY(dev_alloc(&lv.b_lo, n));
        DPCG_HIP(hipMemcpyAsync(lv.pk_meta, meta.data(), meta.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        DPCG_HIP(hipMemcpyAsync(lv.pk_val, pv.data(), pv.size() * sizeof(double), hipMemcpyHostToDevice, s));
        DPCG_HIP(hipStreamSynchronize(s));
    }
    return DPCG_OK;
}

static int set_llt_from_host(dpcg_system *h, int mode, int64_t nnz, const int32_t *rp_in, const int32_t *ci_in,
                             const double *v_in, hipStream_t s) {
    const int64_t n = h->A.n;
    // validate: lower triangular, ascending columns, diagonal last and positive
    for (int64_t i = 0; i < n; ++i) {
        const int32_t a = rp_in[i], b = rp_in[i + 1];
        if (b <= a || ci_in[b - 1] != i) return invalid("L: every row needs its diagonal stored last");
        for (int32_t k = a; k < b - 1; ++k)
            if (ci_in[k] >= ci_in[k + 1]) return invalid("L: columns must ascend within a row (lower triangular)");
        if (!(v_in[b - 1] > 0.0)) {
            set_error("L: non-positive diagonal");
            return DPCG_ERR_PIVOT;
        }
    }
    if (rp_in[n] != nnz) return invalid("L: rowptr[n] != nnz");
    DPCG_TRY(upload_csr(h->L, n, nnz, rp_in, ci_in, v_in, DPCG_F64, DPCG_HOST, 1, s));
    // L^T as CSR: counting transpose, stable in the row index so columns ascend and the diagonal is first
    std::vector<int32_t> trp((size_t)n + 1, 0), tci((size_t)nnz);
    std::vector<double> tv((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) trp[ci_in[k] + 1]++;
    for (int64_t i = 0; i < n; ++i) trp[i + 1] += trp[i];
    {
        std::vector<int32_t> cur(trp.begin(), trp.end() - 1);
        for (int64_t i = 0; i < n; ++i)
            for (int32_t k = rp_in[i]; k < rp_in[i + 1]; ++k) {
                const int32_t dst = cur[ci_in[k]]++;
                tci[dst] = (int32_t)i;
                tv[dst] = v_in[k];
            }
    }
    DPCG_TRY(upload_csr(h->Lt, n, nnz, trp.data(), tci.data(), tv.data(), DPCG_F64, DPCG_HOST, 1, s));
    DPCG_TRY(make_plan(h->L, h->planL, s));
    DPCG_TRY(make_plan(h->Lt, h->planLt, s));
    if (mode == DPCG_PRECOND_LLT_SOLVE) {
        std::vector<int32_t> rp(rp_in, rp_in + n + 1), ci(ci_in, ci_in + nnz), rows, lptr;
        build_levels_host(n, rp, ci, false, rows, lptr);
        DPCG_TRY(upload_levels(h->lvlL, rows, lptr, rp_in, ci_in, v_in, s));
        build_levels_host(n, trp, tci, true, rows, lptr);
        DPCG_TRY(upload_levels(h->lvlU, rows, lptr, trp.data(), tci.data(), tv.data(), s));
    }
    h->precond = mode;
    return DPCG_OK;
}

extern "C" int dpcg_set_precond_llt(dpcg_handle_t h, int mode, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                                    const double *val, int memspace, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    if (mode != DPCG_PRECOND_LLT_MULTIPLY && mode != DPCG_PRECOND_LLT_SOLVE) return invalid("bad LLT mode");
    if (nnz <= 0 || !rowptr || !col || !val) return invalid("dpcg_set_precond_llt: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    free_precond(h);
    const int64_t n = h->A.n;
    if (memspace == DPCG_HOST) return set_llt_from_host(h, mode, nnz, rowptr, col, val, s);
    // device-resident factor (e.g. straight from the CNN): the structural analysis runs on the host
    std::vector<int32_t> rp((size_t)n + 1), ci((size_t)nnz);
    std::vector<double> v((size_t)nnz);
    DPCG_HIP(hipMemcpyAsync(rp.data(), rowptr, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipMemcpyAsync(ci.data(), col, ci.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipMemcpyAsync(v.data(), val, v.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    return set_llt_from_host(h, mode, nnz, rp.data(), ci.data(), v.data(), s);
}

// IC(0) of A on its lower-triangular pattern (stands in for ilupp.ichol0, test.py:83).  The symbolic part
// (tril pattern, level sets) is integer work on the host; the numeric factorisation runs on the device,
// one launch per level (k_ic0_level), in the operation order of the CPU restatement (bit-identical factor).
extern "C" int dpcg_set_precond_ic0(dpcg_handle_t h, int mode, dpcg_stream_t stream) {
    if (!h) return invalid("NULL handle");
    if (mode != DPCG_PRECOND_LLT_MULTIPLY && mode != DPCG_PRECOND_LLT_SOLVE) return invalid("bad LLT mode");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = h->A.n, nnz = h->A.nnz;
    std::vector<int32_t> rp((size_t)n + 1), ci((size_t)nnz);
    std::vector<double> v((size_t)nnz);
    DPCG_HIP(hipMemcpyAsync(rp.data(), h->A.rowptr, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipMemcpyAsync(ci.data(), h->A.col, ci.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipMemcpyAsync(v.data(), h->A.val, v.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> lrp((size_t)n + 1, 0), lci;
    std::vector<double> lv;
    lci.reserve((size_t)nnz / 2 + n);
    lv.reserve((size_t)nnz / 2 + n);
    for (int64_t i = 0; i < n; ++i) {
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k)
            if (ci[k] <= i) {
                lci.push_back(ci[k]);
                lv.push_back(v[k]);
            }
        lrp[i + 1] = (int32_t)lci.size();
        if (lrp[i + 1] == lrp[i] || lci.back() != i) {
            set_error("IC(0): missing diagonal entry");
            return DPCG_ERR_PIVOT;
        }
    }
    const int64_t lnnz = (int64_t)lci.size();
    std::vector<int32_t> rows, lptr;
    build_levels_host(n, lrp, lci, false, rows, lptr);
    int32_t *d_rp = nullptr, *d_ci = nullptr, *d_rows = nullptr;
    double *d_lv = nullptr;
    int *d_bad = nullptr, h_bad = 0;
    int st = DPCG_OK;
    auto cleanup = [&]() { dev_free(d_rp); dev_free(d_ci); dev_free(d_rows); dev_free(d_lv); dev_free(d_bad); };
    if ((st = dev_alloc(&d_rp, n + 1)) < 0 || (st = dev_alloc(&d_ci, lnnz)) < 0 || (st = dev_alloc(&d_rows, n)) < 0 ||
        (st = dev_alloc(&d_lv, lnnz)) < 0 || (st = dev_alloc(&d_bad, 1)) < 0) {
        cleanup();
        return st;
    }
    hipError_t e = hipMemcpyAsync(d_rp, lrp.data(), lrp.size() * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ci, lci.data(), lci.size() * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lv, lv.data(), lv.size() * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(int), s);
    if (e == hipSuccess) {
        const int nl = (int)lptr.size() - 1;
        for (int l = 0; l < nl; ++l) launch_ic0_level(d_rows, lptr[l], lptr[l + 1] - lptr[l], d_rp, d_ci, d_lv, d_bad, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(lv.data(), d_lv, lv.size() * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    DPCG_HIP(e);
    if (h_bad) {
        set_error("IC(0): non-positive pivot at row " + std::to_string(h_bad - 1));
        return DPCG_ERR_PIVOT;
    }
    free_precond(h);
    return set_llt_from_host(h, mode, lnnz, lrp.data(), lci.data(), lv.data(), s);
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
