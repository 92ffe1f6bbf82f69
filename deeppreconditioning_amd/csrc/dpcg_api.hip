// Host side of libdpcg.so: the C ABI declared in include/dpcg.h, handle management, setup analysis
// (row-block plan, IC(0), transpose, level sets) and the PCG driver that replays the iteration
// kernels from a hipGraph without a host round trip per iteration.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "dpcg_internal.h"

using namespace dpcg;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
namespace dpcg {
static thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
    return DPCG_ERR_HIP;
}
}  // namespace dpcg

static int invalid(const char *msg) {
    set_error(msg);
    return DPCG_ERR_INVALID;
}

#define DPCG_TRY(expr)               \
    do {                             \
        int _st = (expr);            \
        if (_st < 0) return _st;     \
    } while (0)

#define DPCG_CHECK_LAUNCH() DPCG_HIP(hipGetLastError())

extern "C" int dpcg_version(void) { return 100; }

extern "C" const char *dpcg_status_string(int st) {
    switch (st) {
        case DPCG_OK: return "ok";
        case DPCG_MAX_ITER: return "max_iter reached";
        case DPCG_BREAKDOWN: return "breakdown (NaN/Inf in the recurrence)";
        case DPCG_ERR_INVALID: return "invalid argument";
        case DPCG_ERR_HIP: return "HIP runtime error";
        case DPCG_ERR_NOMEM: return "out of memory";
        case DPCG_ERR_PIVOT: return "IC(0): non-positive pivot";
        case DPCG_ERR_STATE: return "invalid state";
        default: return "unknown status";
    }
}

extern "C" const char *dpcg_last_error(void) { return g_last_error.c_str(); }

extern "C" int dpcg_device_info(int *cu_count, int64_t *hbm_bytes, char *name, int name_len) {
    int dev = 0;
    DPCG_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    DPCG_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s", prop.gcnArchName);
    }
    return DPCG_OK;
}

// ------------------------------------------------------------------------------------------------
// memory helpers
// ------------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(T **p, int64_t count) {
    *p = nullptr;
    if (count <= 0) count = 1;
    hipError_t e = hipMalloc((void **)p, (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? DPCG_ERR_NOMEM : DPCG_ERR_HIP;
    }
    return DPCG_OK;
}

template <typename T>
static void dev_free(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

static void free_csr(CsrDev &c) {
    if (c.owned) {
        dev_free(c.rowptr);
        dev_free(c.col);
        dev_free(c.val);
    }
    dev_free(c.val32);  // the converted copy is always ours
    c = CsrDev();
}

static void free_levels(Levels &l) {
    dev_free(l.rows);
    dev_free(l.level_ptr_dev);
    dev_free(l.lo_rowptr);
    dev_free(l.lo_col);
    dev_free(l.lo_cpos);
    dev_free(l.lo_val);
    dev_free(l.pk_meta);
    dev_free(l.pk_val);
    dev_free(l.b_lo);
    l = Levels();
}

static int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    return (int)g;
}

// Upload (or adopt) a CSR matrix.  val_dtype F32 input keeps the fp32 array and adds an fp64 copy.
static int upload_csr(CsrDev &out, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col, const void *val,
                      int val_dtype, int memspace, int copy, hipStream_t s) {
    out = CsrDev();
    out.n = n;
    out.nnz = nnz;
    const hipMemcpyKind kind = memspace == DPCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const bool borrow = (memspace == DPCG_DEVICE && !copy);
    if (borrow) {
        out.owned = false;
        out.rowptr = const_cast<int32_t *>(rowptr);
        out.col = const_cast<int32_t *>(col);
        if (val_dtype == DPCG_F64) {
            out.val = const_cast<double *>((const double *)val);
        } else {
            // borrowed fp32 values: the fp64 copy is ours; freed through val32/val bookkeeping below
            DPCG_TRY(dev_alloc(&out.val32, nnz));
            DPCG_HIP(hipMemcpyAsync(out.val32, val, (size_t)nnz * sizeof(float), hipMemcpyDeviceToDevice, s));
            double *v64 = nullptr;
            DPCG_TRY(dev_alloc(&v64, nnz));
            launch_f32_to_f64(nnz, out.val32, v64, s);
            // adopt ownership of everything to keep freeing simple
            int32_t *rp = nullptr, *ci = nullptr;
            DPCG_TRY(dev_alloc(&rp, n + 1));
            DPCG_TRY(dev_alloc(&ci, nnz));
            DPCG_HIP(hipMemcpyAsync(rp, rowptr, (size_t)(n + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            DPCG_HIP(hipMemcpyAsync(ci, col, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            out.rowptr = rp;
            out.col = ci;
            out.val = v64;
            out.owned = true;
        }
        return DPCG_OK;
    }
    out.owned = true;
    DPCG_TRY(dev_alloc(&out.rowptr, n + 1));
    DPCG_TRY(dev_alloc(&out.col, nnz));
    DPCG_TRY(dev_alloc(&out.val, nnz));
    DPCG_HIP(hipMemcpyAsync(out.rowptr, rowptr, (size_t)(n + 1) * sizeof(int32_t), kind, s));
    DPCG_HIP(hipMemcpyAsync(out.col, col, (size_t)nnz * sizeof(int32_t), kind, s));
    if (val_dtype == DPCG_F64) {
        DPCG_HIP(hipMemcpyAsync(out.val, val, (size_t)nnz * sizeof(double), kind, s));
    } else {
        DPCG_TRY(dev_alloc(&out.val32, nnz));
        DPCG_HIP(hipMemcpyAsync(out.val32, val, (size_t)nnz * sizeof(float), kind, s));
        launch_f32_to_f64(nnz, out.val32, out.val, s);
    }
    DPCG_HIP(hipStreamSynchronize(s));  // host source buffers may be released by the caller
    return DPCG_OK;
}

// Choose the SpMV kernel: CSR-stream when every 256-row block's non-zeros fit the LDS product
// buffer (stencils, OpenFOAM-like rows), otherwise CSR-vector with lanes-per-row ~ mean row length.
static void free_plan(SpmvPlan &plan) {
    dev_free(plan.tile_chunks);
    dev_free(plan.tile_nchunks);
    dev_free(plan.tile_lidx);
    plan = SpmvPlan();
}

static int make_plan(const CsrDev &A, SpmvPlan &plan, hipStream_t s, bool allow_tile = false) {
    free_plan(plan);
    int *d_max = nullptr;
    DPCG_TRY(dev_alloc(&d_max, 1));
    DPCG_HIP(hipMemsetAsync(d_max, 0, sizeof(int), s));
    launch_block_nnz_max(A, kStreamRows, d_max, s);
    int h_max = 0;
    DPCG_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    dev_free(d_max);
    const char *force = getenv("DPCG_SPMV_KERNEL");
    const bool force_vector = force && strcmp(force, "vector") == 0;
    if (h_max <= kStreamCap && !force_vector) {
        plan.kernel = SPMV_STREAM;
        plan.nrb = (int)((A.n + kStreamRows - 1) / kStreamRows);
        // 8 workgroups per CU while the matrix stream stays in the 256 MiB Infinity Cache, 6 per CU once
        // it comes from HBM (measured: tools/lib_lab, 100^3: 20.7 vs 21.7 us; 256^3: 457 vs 424 us)
        const double stream_bytes = 12.0 * (double)A.nnz + 20.0 * (double)A.n;
        int cap = stream_bytes < 192e6 ? kMaxSpmvGrid : (kMaxSpmvGrid * 3) / 4;
        // x-tile variant: stage the block's chunks of x in LDS when the columns of every block form a few runs
        // (2 B less per non-zero, no x re-gather, and the next block's stream AND x chunks prefetched into
        // registers).  Measured (tools/lib_lab, tools/c4_probe.py): 256^3 350-362 vs 419 us, 100^3 20.1 vs
        // 20.9 us, 1024^2 +2.7 % its/s; it loses when there are too few row blocks to keep 6 workgroups per CU
        // busy (256^2, 256 blocks: 86K vs 91K its/s), hence the block-count threshold.
        const bool force_stream = force && strcmp(force, "stream") == 0;
        const bool force_tile = force && strcmp(force, "tile") == 0;
        if (allow_tile && !force_stream && A.nnz > 0 && (force_tile || plan.nrb >= kTileMinBlocks)) {
            int *d_flags = nullptr, h_flags[2] = {1, 0};
            DPCG_TRY(dev_alloc(&plan.tile_chunks, (int64_t)plan.nrb * kTileMaxChunks));
            DPCG_TRY(dev_alloc(&plan.tile_nchunks, plan.nrb));
            DPCG_TRY(dev_alloc(&plan.tile_lidx, A.nnz + 2));                   // read in aligned pairs
            DPCG_HIP(hipMemsetAsync(plan.tile_lidx + A.nnz, 0, 2 * sizeof(uint16_t), s));
            DPCG_TRY(dev_alloc(&d_flags, 2));
            DPCG_HIP(hipMemcpyAsync(d_flags, h_flags, sizeof(h_flags), hipMemcpyHostToDevice, s));
            launch_tile_plan(A, plan.nrb, plan.tile_chunks, plan.tile_nchunks, plan.tile_lidx, d_flags, s);
            DPCG_HIP(hipMemcpyAsync(h_flags, d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            dev_free(d_flags);
            if (h_flags[0] == 1 && h_flags[1] > 0) {
                plan.kernel = SPMV_TILE;
                plan.tile_max_chunks = h_flags[1];
                const size_t lds = (size_t)(h_flags[1] * kTileChunk + kStreamCap + 6) * sizeof(double);
                const int per_cu = (int)std::min<size_t>(8, (160 * 1024) / lds);
                cap = std::min(cap, per_cu * 256);
            } else {
                dev_free(plan.tile_chunks);
                dev_free(plan.tile_nchunks);
                dev_free(plan.tile_lidx);
            }
        }
        int g = plan.nrb < cap ? plan.nrb : cap;
        if (g > 8) g -= g % 8;
        plan.grid = g < 1 ? 1 : g;
    } else {
        plan.kernel = SPMV_VECTOR;
        const double mean = A.n > 0 ? (double)A.nnz / (double)A.n : 1.0;
        int tpr = 2;
        while (tpr < 64 && tpr < mean) tpr *= 2;
        plan.tpr = tpr;
        const int64_t ngroups = (A.n + (kBlock / tpr) - 1) / (kBlock / tpr);
        int g = ngroups < kMaxSpmvGrid ? (int)ngroups : kMaxSpmvGrid;
        if (g > 8) g -= g % 8;
        plan.grid = g < 1 ? 1 : g;
    }
    return DPCG_OK;
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct HandleExtras {
    hipStream_t cap_stream = nullptr;
    unsigned long long *prog_host = nullptr;  // pinned + mapped: the solve's progress word
    unsigned long long *prog_dev = nullptr;   // device-side address of the same word
};
// kept outside dpcg_system so the struct in the header stays POD-like; the registry itself is guarded so that
// handles may be created/destroyed from several host threads (one handle is still used by one thread at a time)
#include <map>
#include <mutex>
struct ExtrasRegistry {
    std::mutex mu;
    std::map<dpcg_system *, HandleExtras> m;
    HandleExtras &operator[](dpcg_system *h) {
        std::lock_guard<std::mutex> lock(mu);
        return m[h];   // std::map nodes are stable: the reference stays valid while the handle lives
    }
    bool take(dpcg_system *h, HandleExtras &out) {
        std::lock_guard<std::mutex> lock(mu);
        auto it = m.find(h);
        if (it == m.end()) return false;
        out = it->second;
        m.erase(it);
        return true;
    }
};
static ExtrasRegistry &extras() {
    static ExtrasRegistry r;
    return r;
}

static void drop_graph(dpcg_system *h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
    h->graph_key = -1;
}

static void free_ell(SmallEll &e);
static void free_precond(dpcg_system *h) {
    drop_graph(h);
    free_ell(h->ell_m);
    free_ell(h->ell_t);
    dev_free(h->dinv);
    free_csr(h->M);
    free_csr(h->L);
    free_csr(h->Lt);
    free_levels(h->lvlL);
    free_levels(h->lvlU);
    h->precond = DPCG_PRECOND_NONE;
}

extern "C" int dpcg_create(dpcg_handle_t *out, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                           const void *val, int val_dtype, int memspace, int copy, dpcg_stream_t stream) {
    if (!out) return invalid("dpcg_create: out is NULL");
    *out = nullptr;
    if (n <= 0 || nnz < 0 || !rowptr || (nnz > 0 && (!col || !val))) return invalid("dpcg_create: bad sizes/pointers");
    if (nnz > 2147483647LL || n > 2147483646LL) return invalid("dpcg_create: int32 CSR limits exceeded");
    if (val_dtype != DPCG_F64 && val_dtype != DPCG_F32) return invalid("dpcg_create: bad val_dtype");
    int ndev = 0;
    DPCG_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) {
        set_error("no HIP device visible");
        return DPCG_ERR_HIP;
    }
    hipStream_t s = (hipStream_t)stream;
    dpcg_system *h = new dpcg_system();
    int st = upload_csr(h->A, n, nnz, rowptr, col, val, val_dtype, memspace, copy, s);
    if (st >= 0) st = make_plan(h->A, h->planA, s, true);
    HandleExtras ex;
    if (st >= 0 && hipStreamCreateWithFlags(&ex.cap_stream, hipStreamNonBlocking) != hipSuccess) st = DPCG_ERR_HIP;
    if (st >= 0 && hipHostMalloc((void **)&ex.prog_host, 64, hipHostMallocMapped) != hipSuccess) st = DPCG_ERR_HIP;
    if (st >= 0 && hipHostGetDevicePointer((void **)&ex.prog_dev, ex.prog_host, 0) != hipSuccess) st = DPCG_ERR_HIP;
    if (st >= 0 && hipHostMalloc((void **)&h->scal_host, sizeof(Scalars), hipHostMallocDefault) != hipSuccess)
        st = DPCG_ERR_HIP;
    if (st >= 0) st = dev_alloc(&h->scal, 1);
    extras()[h] = ex;
    if (st < 0) {
        dpcg_destroy(h);
        return st;
    }
    h->vec_grid = grid_for(n);
    *out = h;
    return DPCG_OK;
}

extern "C" int dpcg_destroy(dpcg_handle_t h) {
    if (!h) return DPCG_OK;
    (void)hipDeviceSynchronize();
    free_precond(h);
    free_csr(h->A);
    free_plan(h->planA);
    free_ell(h->ell_a);
    dev_free(h->x); dev_free(h->r); dev_free(h->z); dev_free(h->p); dev_free(h->p2); dev_free(h->q); dev_free(h->t); dev_free(h->e);
    dev_free(h->p32);
    dev_free(h->part_pq); dev_free(h->part_rz); dev_free(h->part_rr); dev_free(h->part_bb);
    dev_free(h->scal); dev_free(h->hist); dev_free(h->err_hist); dev_free(h->small_desc);
    if (h->scal_host) (void)hipHostFree(h->scal_host);
    HandleExtras ex;
    if (extras().take(h, ex)) {
        if (ex.cap_stream) (void)hipStreamDestroy(ex.cap_stream);
        if (ex.prog_host) (void)hipHostFree(ex.prog_host);
    }
    delete h;
    return DPCG_OK;
}

// Two-kernel updates (fused_head in dpcg_device.h) trade one kernel boundary and one pass over p for a second
// gather per non-zero.  Measured with tools/fuse_probe.py (Jacobi PCG, its/s, two- vs three-kernel): 16K rows
// +18 %, 65K +20 %, 147K +12 %, 262K +8 %, 512K -6 %, 1M -8 % (scrambled 1M: -48 %).  So: systems below the x-tile
// threshold, where an update is launch-bound rather than bandwidth-bound.
static bool fuse_eligible(const dpcg_system *h, int flags, const double *x_true) {
    static const int64_t max_rows = [] {
        const char *e = getenv("DPCG_FUSE_MAX_ROWS");
        return e ? (int64_t)atoll(e) : (int64_t)kTileMinBlocks * kStreamRows;
    }();
    if (x_true || (flags & (DPCG_NO_FUSE | DPCG_SPMV_F32))) return false;
    if ((flags & DPCG_VAL32_IF_LOSSLESS) && h->A.val32_lossless == 1) return false;
    if (h->A.n >= max_rows) return false;
    return h->planA.kernel == SPMV_STREAM || h->planA.kernel == SPMV_TILE;
}

static FuseArgs fuse_args(dpcg_system *h) {
    FuseArgs fa;
    fa.z = h->precond == DPCG_PRECOND_NONE ? h->r : h->z;
    fa.p0 = h->p;
    fa.p1 = h->p2;
    fa.xvec = h->x;
    fa.part_rz = h->part_rz;
    fa.part_rr = h->part_rr;
    fa.n_part = h->vec_grid;
    fa.hist = h->hist;
    fa.hist_cap = h->hist_cap;
    return fa;
}

extern "C" int dpcg_get_info(dpcg_handle_t h, int64_t *n, int64_t *nnz, int *spmv_kernel, int *precond_kind,
                             int64_t *precond_nnz, int *n_levels_lower, int *n_levels_upper) {
    if (!h) return invalid("dpcg_get_info: NULL handle");
    if (n) *n = h->A.n;
    if (nnz) *nnz = h->A.nnz;
    if (spmv_kernel) *spmv_kernel = h->planA.kernel + (fuse_eligible(h, 0, nullptr) ? 16 : 0);   // +16: two-kernel updates
    if (precond_kind) *precond_kind = h->precond;
    if (precond_nnz) *precond_nnz = h->precond == DPCG_PRECOND_CSR ? h->M.nnz : h->L.nnz;
    if (n_levels_lower) *n_levels_lower = h->lvlL.n_levels;
    if (n_levels_upper) *n_levels_upper = h->lvlU.n_levels;
    return DPCG_OK;
}

static int ensure_work(dpcg_system *h, int max_iter, bool need_f32, bool need_err) {
    const int64_t n = h->A.n;
    if (!h->x) {
        DPCG_TRY(dev_alloc(&h->x, n));
        DPCG_TRY(dev_alloc(&h->r, n));
        DPCG_TRY(dev_alloc(&h->z, n));
        DPCG_TRY(dev_alloc(&h->p, n));
        DPCG_TRY(dev_alloc(&h->q, n));
        DPCG_TRY(dev_alloc(&h->t, n));
        DPCG_TRY(dev_alloc(&h->part_pq, kMaxSpmvGrid));
        DPCG_TRY(dev_alloc(&h->part_rz, kMaxGrid));
        DPCG_TRY(dev_alloc(&h->part_rr, kMaxGrid));
        DPCG_TRY(dev_alloc(&h->part_bb, kMaxSpmvGrid));
    }
    if (need_err && !h->e) DPCG_TRY(dev_alloc(&h->e, n));
    if (need_f32) {
        if (!h->p32) {
            DPCG_TRY(dev_alloc(&h->p32, n));
            drop_graph(h);
        }
        if (!h->A.val32) {
            DPCG_TRY(dev_alloc(&h->A.val32, h->A.nnz));
            launch_f64_to_f32(h->A.nnz, h->A.val, h->A.val32, nullptr);
            DPCG_HIP(hipDeviceSynchronize());
        }
    }
    if (h->hist_cap < max_iter + 1) {
        drop_graph(h);  // graph nodes hold the old pointers
        dev_free(h->hist);
        dev_free(h->err_hist);
        h->hist_cap = max_iter + 1;
        DPCG_TRY(dev_alloc(&h->hist, h->hist_cap));
        DPCG_TRY(dev_alloc(&h->err_hist, h->hist_cap));
    }
    return DPCG_OK;
}

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

// ------------------------------------------------------------------------------------------------
// standalone operators
// ------------------------------------------------------------------------------------------------
extern "C" int dpcg_spmv(dpcg_handle_t h, const double *x, double *y, dpcg_stream_t stream) {
    if (!h || !x || !y) return invalid("dpcg_spmv: NULL argument");
    launch_spmv(h->A, h->planA, x, y, nullptr, nullptr, (hipStream_t)stream);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_spmv_f32(dpcg_handle_t h, const float *x, float *y, dpcg_stream_t stream) {
    if (!h || !x || !y) return invalid("dpcg_spmv_f32: NULL argument");
    if (!h->A.val32) {
        DPCG_TRY(dev_alloc(&h->A.val32, h->A.nnz));
        launch_f64_to_f32(h->A.nnz, h->A.val, h->A.val32, (hipStream_t)stream);
    }
    launch_spmv_f32out(h->A, h->planA, x, y, (hipStream_t)stream);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// z = M r for the handle's preconditioner (cg.py:61,81).  `t` is the handle's scratch vector.
static int apply_precond(dpcg_system *h, const double *r, double *z, hipStream_t s, bool in_loop = false) {
    switch (h->precond) {
        case DPCG_PRECOND_NONE:
            if (z != r) DPCG_HIP(hipMemcpyAsync(z, r, (size_t)h->A.n * sizeof(double), hipMemcpyDeviceToDevice, s));
            break;
        case DPCG_PRECOND_JACOBI:
            launch_scale(h->A.n, h->dinv, r, z, h->vec_grid, s);
            break;
        case DPCG_PRECOND_CSR:
            launch_spmv(h->M, h->planM, r, z, nullptr, nullptr, s);
            break;
        case DPCG_PRECOND_LLT_MULTIPLY:
            launch_spmv(h->Lt, h->planLt, r, h->t, nullptr, nullptr, s);
            launch_spmv(h->L, h->planL, h->t, z, nullptr, nullptr, s);
            break;
        case DPCG_PRECOND_LLT_SOLVE:
            launch_sptrsv(h->L, h->lvlL, false, r, h->t, s, in_loop ? &h->scal->done : nullptr);
            launch_sptrsv(h->Lt, h->lvlU, true, h->t, z, s, in_loop ? &h->scal->done : nullptr);
            break;
        default:
            set_error("unknown preconditioner kind");
            return DPCG_ERR_STATE;
    }
    return DPCG_OK;
}

extern "C" int dpcg_precond_apply(dpcg_handle_t h, const double *r, double *z, dpcg_stream_t stream) {
    if (!h || !r || !z) return invalid("dpcg_precond_apply: NULL argument");
    DPCG_TRY(ensure_work(h, 0, false, false));
    DPCG_TRY(apply_precond(h, r, z, (hipStream_t)stream));
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_sptrsv(dpcg_handle_t h, int upper, const double *rhs, double *out, dpcg_stream_t stream) {
    if (!h || !rhs || !out) return invalid("dpcg_sptrsv: NULL argument");
    if (h->precond != DPCG_PRECOND_LLT_SOLVE) {
        set_error("dpcg_sptrsv: needs dpcg_set_precond_llt/ic0 in LLT_SOLVE mode");
        return DPCG_ERR_STATE;
    }
    if (upper) launch_sptrsv(h->Lt, h->lvlU, true, rhs, out, (hipStream_t)stream);
    else launch_sptrsv(h->L, h->lvlL, false, rhs, out, (hipStream_t)stream);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_dot(int64_t n, const double *a, const double *b, double *out_host, dpcg_stream_t stream) {
    if (n <= 0 || !a || !b || !out_host) return invalid("dpcg_dot: bad argument");
    hipStream_t s = (hipStream_t)stream;
    double *part = nullptr;
    DPCG_TRY(dev_alloc(&part, kMaxGrid + 1));
    const int g = grid_for(n);
    launch_dot_partials(n, nullptr, a, b, part, g, s);
    launch_dot_final(part, g, part + kMaxGrid, s);
    hipError_t e = hipMemcpyAsync(out_host, part + kMaxGrid, sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    dev_free(part);
    DPCG_HIP(e);
    return DPCG_OK;
}

// Times the SpMV+<p,Ap> kernel in exactly the instantiation the PCG loop launches (iteration head
// included), `repeats` back-to-back launches bracketed by HIP events on `stream`.
extern "C" int dpcg_spmv_dot_bench(dpcg_handle_t h, const double *x, double *y, int repeats, float *ms_per_launch,
                                   dpcg_stream_t stream) {
    if (!h || !x || !y || repeats <= 0 || !ms_per_launch) return invalid("dpcg_spmv_dot_bench: bad argument");
    hipStream_t s = (hipStream_t)stream;
    DPCG_TRY(ensure_work(h, 0, false, false));
    // a never-finishing control block: done = 0 (thresholds 0, <b,b> = 1)
    DPCG_HIP(hipMemsetAsync(h->part_rr, 0, kMaxGrid * sizeof(double), s));
    const double one = 1.0;
    DPCG_HIP(hipMemcpyAsync(h->part_rr, &one, sizeof(double), hipMemcpyHostToDevice, s));
    launch_finalize_init(h->scal, h->part_rr, h->part_rr, h->part_rr, 1, 0.0, 0.0, h->hist, 0, nullptr, s);
    IterCtl ctl{h->scal};
    // the kernel a default solve launches: KA of the two-kernel iteration (update 0: beta = 0, so y = A x still
    // holds with z := x) or the plain SpMV + <p,Ap> kernel of the three-kernel form
    const bool fused = fuse_eligible(h, 0, nullptr);
    FuseArgs fa;
    if (fused) {
        if (!h->p2) {
            DPCG_TRY(dev_alloc(&h->p2, h->A.n));
            drop_graph(h);
        }
        DPCG_HIP(hipMemsetAsync(h->p2, 0, (size_t)h->A.n * sizeof(double), s));
        DPCG_HIP(hipMemsetAsync(h->x, 0, (size_t)h->A.n * sizeof(double), s));
        DPCG_HIP(hipMemsetAsync(h->part_rz, 0, kMaxGrid * sizeof(double), s));
        launch_fused_init(h->scal, s);
        fa = fuse_args(h);
        fa.z = x;
    }
    auto go = [&]() {
        if (fused) launch_spmv_fused(h->A, h->planA, fa, y, h->part_pq, h->scal, s);
        else launch_spmv(h->A, h->planA, x, y, h->part_pq, &ctl, s);
    };
    hipEvent_t e0, e1;
    DPCG_HIP(hipEventCreate(&e0));
    DPCG_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) go();  // warm-up
    DPCG_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < repeats; ++i) go();
    DPCG_HIP(hipEventRecord(e1, s));
    DPCG_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DPCG_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / (float)repeats;
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// ------------------------------------------------------------------------------------------------
// the solve
// ------------------------------------------------------------------------------------------------
static int default_chunk() {
    const char *e = getenv("DPCG_CHUNK");
    int c = e ? atoi(e) : 8;
    return c < 1 ? 1 : (c > 256 ? 256 : c);
}

// One PCG update (cg.py:75-86) as kernel launches on `s`.
static int enqueue_iteration(dpcg_system *h, int flags, const double *x_true, hipStream_t s) {
    const int64_t n = h->A.n;
    const bool f32 = (flags & DPCG_SPMV_F32) != 0;
    if (fuse_eligible(h, flags, x_true)) {
        // KA: test of the current iterate, p = z + beta p, deferred x += alpha p, q = A p, partials of <p,q>
        launch_spmv_fused(h->A, h->planA, fuse_args(h), h->q, h->part_pq, h->scal, s);          // cg.py:71,83,79,75
        // KB: alpha; r -= alpha q; (z = M r fused); partials <r,z>, <r,r>; k += 1                cg.py:78,80-82,86
        const int pre = h->precond == DPCG_PRECOND_NONE ? 0 : (h->precond == DPCG_PRECOND_JACOBI ? 1 : 2);
        launch_update_r_two_kernel(pre, n, h->scal, h->part_pq, h->planA.grid, h->q, h->r, h->dinv, h->z, h->part_rz,
                                   h->part_rr, h->vec_grid, s);
        if (pre == 2) {
            DPCG_TRY(apply_precond(h, h->r, h->z, s, true));                                     // cg.py:81
            launch_dot_partials(n, h->scal, h->r, h->z, h->part_rz, h->vec_grid, s);             // cg.py:82
        }
        return DPCG_OK;
    }
    IterCtl ctl{h->scal};
    // K1: (skip when done) Ap = A p + partials of <p,Ap>           cg.py:71,75,78
    const bool v32 = !f32 && (flags & DPCG_VAL32_IF_LOSSLESS) && h->A.val32_lossless == 1;
    if (f32) launch_spmv_f32in(h->A, h->planA, h->p32, h->p, h->q, h->part_pq, &ctl, s);
    else if (v32) launch_spmv_val32(h->A, h->planA, h->p, h->q, h->part_pq, &ctl, s);
    else launch_spmv(h->A, h->planA, h->p, h->q, h->part_pq, &ctl, s);
    // K2: alpha; r -= alpha Ap; (z = M r fused); partials <r,z>, <r,r>              cg.py:78,80-82,86
    const int pre = h->precond == DPCG_PRECOND_NONE ? 0 : (h->precond == DPCG_PRECOND_JACOBI ? 1 : 2);
    double *z = pre == 0 ? h->r : h->z;
    launch_update_r(pre, n, h->scal, h->part_pq, h->planA.grid, h->q, h->r, h->dinv, h->z, h->part_rz, h->part_rr,
                    h->vec_grid, s);
    if (pre == 2) {
        DPCG_TRY(apply_precond(h, h->r, h->z, s, true));                             // cg.py:81
        launch_dot_partials(n, h->scal, h->r, h->z, h->part_rz, h->vec_grid, s);     // cg.py:82
    }
    // K3: beta; x += alpha p; p = z + beta p; workgroup 0: stopping test of the new iterate   cg.py:79,82-83,86,71
    launch_update_xp(n, h->scal, h->part_rz, h->part_rr, h->vec_grid, z, h->p, h->x, f32 ? h->p32 : nullptr, h->hist,
                     h->hist_cap, h->vec_grid, s);
    if (x_true) {                                                                    // cg.py:43-45
        launch_anorm_err(n, h->scal, h->x, x_true, h->e, h->vec_grid, s);
        launch_spmv(h->A, h->planA, h->e, h->t, h->part_bb, nullptr, s);
        launch_record_err(h->scal, h->part_bb, h->planA.grid, h->err_hist, h->hist_cap, 0, s);
    }
    return DPCG_OK;
}

static int ensure_graph(dpcg_system *h, int flags, int chunk) {
    const int key = (h->precond << 8) | (flags & (DPCG_SPMV_F32 | DPCG_VAL32_IF_LOSSLESS | DPCG_NO_FUSE)) |
                    (h->A.val32_lossless == 1 ? 64 : 0) | (fuse_eligible(h, flags, nullptr) ? 128 : 0);
    if (h->graph_exec && h->graph_key == key && h->graph_chunk == chunk) return DPCG_OK;
    drop_graph(h);
    HandleExtras &ex = extras()[h];
    hipGraph_t graph = nullptr;
    DPCG_HIP(hipStreamBeginCapture(ex.cap_stream, hipStreamCaptureModeThreadLocal));
    int st = DPCG_OK;
    for (int i = 0; i < chunk && st >= 0; ++i) st = enqueue_iteration(h, flags, nullptr, ex.cap_stream);
    hipError_t e = hipStreamEndCapture(ex.cap_stream, &graph);
    if (st < 0) {
        if (graph) (void)hipGraphDestroy(graph);
        return st;
    }
    DPCG_HIP(e);
    e = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    DPCG_HIP(e);
    h->graph_key = key;
    h->graph_chunk = chunk;
    return DPCG_OK;
}

// kernel launches one preconditioner application costs (the SpTRSVs launch once per wide level)
static int precond_launches(const dpcg_system *h) {
    auto trsv = [](const Levels &lv) {
        int c = 0;
        for (const auto &seg : lv.segments) c += seg.merged ? 2 : seg.hi - seg.lo;
        return c;
    };
    switch (h->precond) {
        case DPCG_PRECOND_CSR: return 1;
        case DPCG_PRECOND_LLT_MULTIPLY: return 2;
        case DPCG_PRECOND_LLT_SOLVE: return trsv(h->lvlL) + trsv(h->lvlU);
        default: return 0;
    }
}

namespace {
// Host side of one solve.  The GPU never waits for the host: iterations are enqueued ahead of the
// progress word that K3 posts to pinned memory, as a replayed hipGraph of `chunk` updates (launch-bound
// small systems) or update by update (large systems, where one update outlasts its three launches).
struct Solve {
    dpcg_system *h = nullptr;
    hipStream_t s = nullptr;
    int max_iter = 0, flags = 0, chunk = 8;
    const double *x_true = nullptr;
    bool use_graph = true;
    int enq = 0;             // updates enqueued so far
    bool complete = false;
    double t_iter = 0.0;     // measured seconds per update (0 = not known yet)
    bool fused = false;           // two-kernel updates (x lags one update behind until finish())
    bool many_launches = false;   // an update is dozens of small launches (level-scheduled SpTRSV): always replay a graph
    std::chrono::steady_clock::time_point t0;

    volatile unsigned long long *prog() { return extras()[h].prog_host; }

    int enqueue_some() {
        const bool graph_now = use_graph && (max_iter - enq) >= chunk && (many_launches || !(t_iter > 25e-6));
        if (graph_now) {
            DPCG_HIP(hipGraphLaunch(h->graph_exec, s));
            enq += chunk;
        } else {
            DPCG_TRY(enqueue_iteration(h, flags, x_true, s));
            enq += 1;
        }
        return DPCG_OK;
    }

    // how many updates to keep enqueued beyond the last one the GPU reported
    int run_ahead() const {
        if (t_iter <= 0.0) return 2 * chunk;
        const double cover = 150e-6;  // host launch + scheduling latency to hide
        int it = (int)(cover / t_iter) + 2;
        if (t_iter > 25e-6 && !(many_launches && use_graph)) return it < 3 ? 3 : it;
        const int chunks = (it + chunk - 1) / chunk + 1;
        return chunks * chunk;
    }

    // Enqueue the start of the solve (cg.py:58-67); the timer starts after the initial residual /
    // preconditioner work has drained, as the reference's does (cg.py:69).
    int start(const double *b, const double *x0, double rtol_sq, double atol_sq) {
        const int64_t n = h->A.n;
        const bool f32 = (flags & DPCG_SPMV_F32) != 0;
        HandleExtras &ex = extras()[h];
        DPCG_TRY(ensure_work(h, max_iter, f32, x_true != nullptr));
        if ((flags & DPCG_VAL32_IF_LOSSLESS) && h->A.val32_lossless == 0) {   // decide once per matrix
            int *d_lossy = nullptr, lossy = 0;
            DPCG_TRY(dev_alloc(&d_lossy, 1));
            if (!h->A.val32) DPCG_TRY(dev_alloc(&h->A.val32, h->A.nnz));
            DPCG_HIP(hipMemsetAsync(d_lossy, 0, sizeof(int), s));
            launch_val32_check(h->A.nnz, h->A.val, h->A.val32, d_lossy, s);
            DPCG_HIP(hipMemcpyAsync(&lossy, d_lossy, sizeof(int), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            dev_free(d_lossy);
            h->A.val32_lossless = lossy ? -1 : 1;
        }
        fused = fuse_eligible(h, flags, x_true);
        if (fused && !h->p2) {
            DPCG_TRY(dev_alloc(&h->p2, n));
            drop_graph(h);
        }
        const int per_update = 3 + precond_launches(h);
        many_launches = per_update >= 16;
        if (many_launches) chunk = std::max(1, std::min(chunk, 1024 / per_update));   // keep the graph at ~1K nodes
        use_graph = !(flags & DPCG_NO_GRAPH) && !x_true && max_iter >= chunk;
        if (use_graph) DPCG_TRY(ensure_graph(h, flags, chunk));
        *ex.prog_host = 0;
        if (x0) {
            DPCG_HIP(hipMemcpyAsync(h->x, x0, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            launch_spmv(h->A, h->planA, h->x, h->q, nullptr, nullptr, s);
            launch_residual(n, b, h->q, h->r, h->vec_grid, s);                       // cg.py:60
        } else {
            DPCG_HIP(hipMemsetAsync(h->x, 0, (size_t)n * sizeof(double), s));        // cg.py:58
            DPCG_HIP(hipMemcpyAsync(h->r, b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        double *z = h->precond == DPCG_PRECOND_NONE ? h->r : h->z;
        if (h->precond != DPCG_PRECOND_NONE) DPCG_TRY(apply_precond(h, h->r, h->z, s));   // cg.py:61
        launch_init_state(n, h->scal, b, h->r, z, h->p, f32 ? h->p32 : nullptr, h->part_bb, h->part_rz, h->part_rr,
                          (flags & DPCG_INIT_CHECK_R) ? 1 : 0, h->vec_grid, s);
        launch_finalize_init(h->scal, h->part_bb, h->part_rz, h->part_rr, h->vec_grid, rtol_sq, atol_sq, h->hist,
                             h->hist_cap, ex.prog_dev, s);
        if (fused) {
            DPCG_HIP(hipMemsetAsync(h->p2, 0, (size_t)n * sizeof(double), s));       // "p_{-1}": multiplied by beta_0 = 0
            launch_fused_init(h->scal, s);
        }
        if (x_true) {                                                                // cg.py:27-29
            launch_anorm_err(n, h->scal, h->x, x_true, h->e, h->vec_grid, s);
            launch_spmv(h->A, h->planA, h->e, h->t, h->part_bb, nullptr, s);
            launch_record_err(h->scal, h->part_bb, h->planA.grid, h->err_hist, h->hist_cap, 0, s);
        }
        DPCG_CHECK_LAUNCH();
        DPCG_HIP(hipStreamSynchronize(s));
        t0 = std::chrono::steady_clock::now();                                       // cg.py:69
        if (max_iter == 0) complete = true;
        return DPCG_OK;
    }

    // Advance.  Returns a negative status on error, 1 when the solve is complete, 0 otherwise.
    int step(bool blocking) {
        if (complete) return 1;
        for (;;) {
            const unsigned long long v = *prog();
            const int k = (int)(v >> 1);
            if ((v & 1ull) || k >= max_iter) {
                complete = true;
                return 1;
            }
            if (k >= 4) t_iter = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / k;
            const int target = run_ahead();
            while (enq < max_iter && enq - k < target) DPCG_TRY(enqueue_some());
            if (!blocking) return 0;
            // wait for the progress word to move; watch the stream so that a fault cannot hang the host
            bool moved = false;
            for (int spin = 0; spin < 4000 && !moved; ++spin) {
                moved = *prog() != v;
                if (!moved) __builtin_ia32_pause();
            }
            if (moved) continue;
            const hipError_t q = hipStreamQuery(s);
            if (q == hipErrorNotReady) continue;
            DPCG_HIP(q);
            // stream drained: every enqueued update has run, the word is final for them
            if (*prog() == v && enq > k) {
                set_error("PCG driver: enqueued updates finished without reporting progress");
                return DPCG_ERR_STATE;
            }
        }
    }

    int finish(double *x, int *iters, double *final_res, double *seconds, double *res_history, double *err_history) {
        const int64_t n = h->A.n;
        if (fused)
            launch_final_fused(n, h->scal, h->part_rr, h->vec_grid, h->hist, h->hist_cap, h->x, h->p, h->p2, h->vec_grid,
                               s);
        else
            launch_final_check(h->scal, s);
        DPCG_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(Scalars), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        const auto t1 = std::chrono::steady_clock::now();                            // cg.py:88
        const Scalars sc = *h->scal_host;
        if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
        if (iters) *iters = sc.k;                                                    // cg.py:90
        if (final_res) *final_res = sc.res;
        if (res_history)
            DPCG_HIP(hipMemcpyAsync(res_history, h->hist, (size_t)(sc.k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
        if (err_history && x_true)
            DPCG_HIP(hipMemcpyAsync(err_history, h->err_hist, (size_t)(sc.k + 1) * sizeof(double),
                                    hipMemcpyDeviceToHost, s));
        if (x) DPCG_HIP(hipMemcpyAsync(x, h->x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        DPCG_HIP(hipStreamSynchronize(s));
        DPCG_CHECK_LAUNCH();
        return sc.status;
    }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
// small systems: the whole solve in one launch, one workgroup per system (dpcg_small.hip)
// ------------------------------------------------------------------------------------------------
static bool small_eligible(const dpcg_system *h, int flags, const double *x_true) {
    static const bool enabled = [] { const char *e = getenv("DPCG_SMALL"); return !(e && e[0] == '0'); }();
    if (!enabled || x_true || (flags & (DPCG_SPMV_F32 | DPCG_NO_SMALL))) return false;
    if (h->A.n > kSmallMaxN) return false;
    return h->precond == DPCG_PRECOND_NONE || h->precond == DPCG_PRECOND_JACOBI || h->precond == DPCG_PRECOND_CSR ||
           h->precond == DPCG_PRECOND_LLT_MULTIPLY;
}

static void free_ell(SmallEll &e) {
    dev_free(e.col);
    dev_free(e.val);
    e = SmallEll();
}

static int build_ell(const CsrDev &A, SmallEll &e, hipStream_t s) {
    if (e.col) return DPCG_OK;
    int *d_w = nullptr, w = 0;
    DPCG_TRY(dev_alloc(&d_w, 1));
    DPCG_HIP(hipMemsetAsync(d_w, 0, sizeof(int), s));
    launch_max_row_len((int)A.n, A.rowptr, d_w, s);
    DPCG_HIP(hipMemcpyAsync(&w, d_w, sizeof(int), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    dev_free(d_w);
    const int64_t slabs = (A.n + 1023) / 1024;
    e.W = w < 1 ? 1 : w;
    DPCG_TRY(dev_alloc(&e.col, slabs * e.W * 1024));
    DPCG_TRY(dev_alloc(&e.val, slabs * e.W * 1024));
    launch_build_ell((int)A.n, A.rowptr, A.col, A.val, e.W, e.col, e.val, s);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// slab-ELL copies of the matrices the small-system kernel multiplies by (built once per matrix)
static int ensure_small(dpcg_system *h, hipStream_t s) {
    DPCG_TRY(build_ell(h->A, h->ell_a, s));
    if (h->precond == DPCG_PRECOND_CSR) DPCG_TRY(build_ell(h->M, h->ell_m, s));
    if (h->precond == DPCG_PRECOND_LLT_MULTIPLY) {
        DPCG_TRY(build_ell(h->L, h->ell_m, s));
        DPCG_TRY(build_ell(h->Lt, h->ell_t, s));
    }
    return DPCG_OK;
}

static SmallDesc make_small_desc(dpcg_system *h, const double *b, const double *x0, double *x, double rtol_sq,
                                 double atol_sq, int max_iter, int flags) {
    SmallDesc d;
    memset(&d, 0, sizeof(d));
    d.n = (int)h->A.n;
    d.precond = h->precond;
    d.max_iter = max_iter;
    d.init_check_r = (flags & DPCG_INIT_CHECK_R) ? 1 : 0;
    d.hist_cap = h->hist_cap;
    d.lds_vectors = h->precond == DPCG_PRECOND_CSR ? 2 : (h->precond == DPCG_PRECOND_LLT_MULTIPLY ? 3 : 1);
    d.variant = small_variant((int)h->A.n, h->ell_a.W, h->precond);
    if (d.variant % 16 != 0) d.lds_vectors = 3;   // register-matrix variants: p, x and dinv live in LDS
    d.rp = h->A.rowptr; d.dinv = h->dinv;
    d.ell_a = h->ell_a; d.ell_m = h->ell_m; d.ell_t = h->ell_t;
    if (h->precond == DPCG_PRECOND_CSR) d.m_rp = h->M.rowptr;
    if (h->precond == DPCG_PRECOND_LLT_MULTIPLY) { d.m_rp = h->L.rowptr; d.t_rp = h->Lt.rowptr; }
    d.b = b; d.x0 = x0; d.x = x ? x : h->x; d.hist = h->hist;
    d.rtol_sq = rtol_sq; d.atol_sq = atol_sq;
    d.out = h->scal;
    return d;
}

static int small_variant_bit(const SmallDesc &d) { return d.variant == 4 * 16 + 7 ? 2 : (d.variant == 6 * 16 + 5 ? 4 : 1); }
static int small_lds_bytes(const SmallDesc &d) { return (int)(((size_t)d.lds_vectors * d.n + 64) * sizeof(double)); }

static int solve_small_one(dpcg_system *h, const double *b, const double *x0, double *x, double rtol_sq, double atol_sq,
                           int max_iter, int flags, hipStream_t s, int *iters, double *final_res, double *seconds,
                           double *res_history) {
    DPCG_TRY(ensure_work(h, max_iter, false, false));
    DPCG_TRY(ensure_small(h, s));
    if (!h->small_desc) DPCG_TRY(dev_alloc(&h->small_desc, 1));
    const SmallDesc d = make_small_desc(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags);
    DPCG_HIP(hipMemcpyAsync(h->small_desc, &d, sizeof(d), hipMemcpyHostToDevice, s));
    DPCG_HIP(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();                                // cg.py:69 (the launch is the loop)
    DPCG_TRY(launch_pcg_small(h->small_desc, 1, small_lds_bytes(d), 1 << h->precond, small_variant_bit(d), s));
    DPCG_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();                                // cg.py:88
    DPCG_CHECK_LAUNCH();
    const Scalars sc = *h->scal_host;
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (iters) *iters = sc.k;
    if (final_res) *final_res = sc.res;
    if (res_history) {
        DPCG_HIP(hipMemcpyAsync(res_history, h->hist, (size_t)(sc.k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
    }
    return sc.status;
}

static int check_solve_args(dpcg_handle_t h, const double *b, int max_iter, int flags, const double *x_true,
                            double *err_history) {
    if (!h || !b) return invalid("dpcg_solve: NULL handle or b");
    if (max_iter < 0) return invalid("dpcg_solve: max_iter < 0");
    if ((x_true == nullptr) != (err_history == nullptr) && x_true == nullptr)
        return invalid("dpcg_solve: err_history needs x_true");
    if ((flags & DPCG_SPMV_F32) && x_true) return invalid("dpcg_solve: x_true tracking is fp64 only");
    if (h->precond == DPCG_PRECOND_JACOBI && !h->dinv) return DPCG_ERR_STATE;
    return DPCG_OK;
}

extern "C" int dpcg_solve(dpcg_handle_t h, const double *b, const double *x0, double *x, double rtol_sq,
                          double atol_sq, int max_iter, int flags, dpcg_stream_t stream, int *iters,
                          double *final_res, double *seconds, double *res_history, const double *x_true,
                          double *err_history) {
    DPCG_TRY(check_solve_args(h, b, max_iter, flags, x_true, err_history));
    if (small_eligible(h, flags, x_true))
        return solve_small_one(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags, (hipStream_t)stream, iters, final_res,
                               seconds, res_history);
    Solve sv;
    sv.h = h;
    sv.s = (hipStream_t)stream;
    sv.max_iter = max_iter;
    sv.flags = flags;
    sv.chunk = default_chunk();
    sv.x_true = x_true;
    DPCG_TRY(sv.start(b, x0, rtol_sq, atol_sq));
    for (;;) {
        const int r = sv.step(true);
        if (r < 0) return r;
        if (r == 1) break;
    }
    return sv.finish(x, iters, final_res, seconds, res_history, err_history);
}

extern "C" int dpcg_solve_batch(int count, dpcg_handle_t *handles, const double *const *b, const double *const *x0,
                                double *const *x, double rtol_sq, double atol_sq, int max_iter, int flags,
                                int n_streams, int *iters, double *final_res, double *seconds, int *status) {
    if (count <= 0 || !handles || !b) return invalid("dpcg_solve_batch: bad arguments");
    if (n_streams < 1) n_streams = 1;
    if (n_streams > 8) n_streams = 8;
    if (n_streams > count) n_streams = count;
    for (int i = 0; i < count; ++i) DPCG_TRY(check_solve_args(handles[i], b[i], max_iter, flags, nullptr, nullptr));
    bool all_small = true;
    for (int i = 0; i < count; ++i) all_small = all_small && small_eligible(handles[i], flags, nullptr);
    if (all_small) {
        // one launch, one workgroup (one CU) per system
        std::vector<SmallDesc> descs((size_t)count);
        int lds = 0, kinds = 0, variants = 0;
        for (int i = 0; i < count; ++i) {
            kinds |= 1 << handles[i]->precond;
            DPCG_TRY(ensure_work(handles[i], max_iter, false, false));
            DPCG_TRY(ensure_small(handles[i], nullptr));
            descs[i] = make_small_desc(handles[i], b[i], x0 ? x0[i] : nullptr, x ? x[i] : nullptr, rtol_sq, atol_sq,
                                       max_iter, flags);
            lds = std::max(lds, small_lds_bytes(descs[i]));
            variants |= small_variant_bit(descs[i]);
        }
        SmallDesc *d_descs = nullptr;
        Scalars *d_out = nullptr;            // one contiguous result array: a single copy back for the whole batch
        std::vector<Scalars> out((size_t)count);
        DPCG_TRY(dev_alloc(&d_descs, count));
        int st_alloc = dev_alloc(&d_out, count);
        if (st_alloc < 0) { dev_free(d_descs); return st_alloc; }
        for (int i = 0; i < count; ++i) descs[i].out = d_out + i;
        hipError_t e = hipMemcpy(d_descs, descs.data(), descs.size() * sizeof(SmallDesc), hipMemcpyHostToDevice);
        const auto t0 = std::chrono::steady_clock::now();
        int st = e == hipSuccess ? launch_pcg_small(d_descs, count, lds, kinds, variants, nullptr) : DPCG_ERR_HIP;
        if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, out.size() * sizeof(Scalars), hipMemcpyDeviceToHost);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        dev_free(d_descs);
        dev_free(d_out);
        DPCG_HIP(e);
        if (st < 0) return st;
        int worst_small = DPCG_OK;
        for (int i = 0; i < count; ++i) {
            if (iters) iters[i] = out[i].k;
            if (final_res) final_res[i] = out[i].res;
            if (seconds) seconds[i] = sec;   // the batch ran as one launch
            if (status) status[i] = out[i].status;
            worst_small = std::max(worst_small, out[i].status);
        }
        return worst_small;
    }
    std::vector<hipStream_t> streams((size_t)n_streams, nullptr);
    for (auto &st : streams) DPCG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<Solve> sv((size_t)count);
    std::vector<int> state((size_t)count, 0);  // 0 = waiting, 1 = running, 2 = finished
    std::vector<int> slot_owner((size_t)n_streams, -1);
    int worst = DPCG_OK, next = 0, done = 0, err = 0;
    while (done < count && !err) {
        bool progressed = false;
        for (int sl = 0; sl < n_streams && !err; ++sl) {
            int i = slot_owner[sl];
            if (i < 0) {
                if (next >= count) continue;
                i = next++;
                slot_owner[sl] = i;
                Solve &v = sv[i];
                v.h = handles[i];
                v.s = streams[sl];
                v.max_iter = max_iter;
                v.flags = flags;
                v.chunk = default_chunk();
                int st = v.start(b[i], x0 ? x0[i] : nullptr, rtol_sq, atol_sq);
                if (st < 0) { err = st; break; }
                state[i] = 1;
                progressed = true;
            }
            Solve &v = sv[i];
            const int r = v.step(false);
            if (r < 0) { err = r; break; }
            if (r == 1) {
                const int st = v.finish(x ? x[i] : nullptr, iters ? &iters[i] : nullptr,
                                        final_res ? &final_res[i] : nullptr, seconds ? &seconds[i] : nullptr, nullptr,
                                        nullptr);
                if (st < 0) { err = st; break; }
                if (status) status[i] = st;
                worst = std::max(worst, st);
                state[i] = 2;
                slot_owner[sl] = -1;
                ++done;
                progressed = true;
            }
        }
        if (!progressed) std::this_thread::yield();
    }
    for (auto &st : streams) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    return err ? err : worst;
}

// ------------------------------------------------------------------------------------------------
// generators and the batched COO SpMV
// ------------------------------------------------------------------------------------------------
extern "C" int dpcg_poisson_sizes(int dim, int64_t n, int64_t *rows, int64_t *nnz) {
    if ((dim != 2 && dim != 3) || n < 1) return invalid("dpcg_poisson_sizes: dim must be 2 or 3, n >= 1");
    const int64_t N = dim == 2 ? n * n : n * n * n;
    const int64_t z = dim == 2 ? 5 * n * n - 4 * n : 7 * n * n * n - 6 * n * n;
    if (z > 2147483647LL) return invalid("dpcg_poisson_sizes: nnz exceeds int32 CSR");
    if (rows) *rows = N;
    if (nnz) *nnz = z;
    return DPCG_OK;
}

extern "C" int dpcg_gen_poisson(int dim, int64_t n, int32_t *rowptr, int32_t *col, void *val, int val_dtype,
                                dpcg_stream_t stream) {
    int64_t rows = 0, nnz = 0;
    DPCG_TRY(dpcg_poisson_sizes(dim, n, &rows, &nnz));
    if (!rowptr || !col || !val) return invalid("dpcg_gen_poisson: NULL output");
    launch_gen_poisson(dim, n, rowptr, col, val, val_dtype, (hipStream_t)stream);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_batched_coo_spmv(int64_t nnz, const int32_t *indices, const float *features, int batch,
                                     int64_t dof, const float *vectors, float *out, int transpose,
                                     dpcg_stream_t stream) {
    if (nnz < 0 || batch <= 0 || dof <= 0 || !vectors || !out || (nnz > 0 && (!indices || !features)))
        return invalid("dpcg_batched_coo_spmv: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DPCG_HIP(hipMemsetAsync(out, 0, (size_t)batch * (size_t)dof * sizeof(float), s));   // utils.py:29
    if (nnz > 0) launch_batched_coo_spmv(nnz, indices, features, batch, dof, vectors, out, transpose, s);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_batched_coo_edge(int64_t nnz, const int32_t *indices, int batch, int64_t dof, const float *a,
                                     const float *c, float *out, int transpose, dpcg_stream_t stream) {
    if (nnz < 0 || batch <= 0 || dof <= 0 || (nnz > 0 && (!indices || !a || !c || !out)))
        return invalid("dpcg_batched_coo_edge: bad arguments");
    if (nnz > 0) launch_batched_coo_edge(nnz, indices, batch, dof, a, c, out, transpose, (hipStream_t)stream);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}
