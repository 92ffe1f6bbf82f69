// Host side of libdpcg.so, part 1: the C ABI declared in include/dpcg.h for errors, handles, the SpMV plan,
// introspection, the standalone operators and the generators.  Preconditioner setup lives in dpcg_precond.hip, the
// PCG driver in dpcg_solve.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dpcg_host.h"

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

int invalid(const char *msg) {
    set_error(msg);
    return DPCG_ERR_INVALID;
}


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
void free_csr(CsrDev &c) {
    if (c.owned) {
        dev_free(c.rowptr);
        dev_free(c.col);
        dev_free(c.val);
    }
    dev_free(c.val32);  // the converted copy is always ours
    c = CsrDev();
}

void free_levels(Levels &l) {
    dev_free(l.rows);
    dev_free(l.level_ptr_dev);
    dev_free(l.lo_rowptr);
    dev_free(l.lo_col);
    dev_free(l.lo_cpos);
    dev_free(l.lo_val);
    dev_free(l.pk_meta);
    dev_free(l.pk_val);
    dev_free(l.b_lo);
    dev_free(l.strips.rows); dev_free(l.strips.level_ptr_dev); dev_free(l.strips.lo_rowptr); dev_free(l.strips.lo_col);
    dev_free(l.strips.lo_cpos); dev_free(l.strips.lo_val); dev_free(l.strips.val); dev_free(l.strips.b_lo);
    dev_free(l.strips.meta); dev_free(l.strips.ticket);
    dev_free(l.sf_meta);
    dev_free(l.sfs_blk);
    dev_free(l.sf_val);
    dev_free(l.tickets);
    dev_free(l.spin_err);
    dev_free(l.lm_pos);
    dev_free(l.lm_from_lower);
    dev_free(l.lm_to_upper);
    dev_free(l.ride_diag);
    dev_free(l.sw_chunks);
    dev_free(l.sw_nchunks);
    dev_free(l.sw_lidx);
    dev_free(l.lm_rhs);
    dev_free(l.lm_out);
    l = Levels();
}

int grid_for(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    static const int64_t cap = [] { const char *e = getenv("DPCG_VEC_GRID"); return e ? (int64_t)atoll(e) : (int64_t)kMaxGrid; }();   // development knob
    if (g > cap) g = cap;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g < 1) g = 1;
    return (int)g;
}

// Upload (or adopt) a CSR matrix.  val_dtype F32 input keeps the fp32 array and adds an fp64 copy.
int upload_csr(CsrDev &out, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col, const void *val,
                      int val_dtype, int memspace, int copy, hipStream_t s) {
    out = CsrDev();
    out.n = n;
    out.nnz = nnz;
    const hipMemcpyKind kind = memspace == DPCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    // the SpMV kernels read col / val as aligned pairs: device arrays that are not 16-byte aligned are copied, not borrowed
    const bool aligned16 = ((((uintptr_t)col) | ((uintptr_t)val)) & 15) == 0;
    const bool borrow = (memspace == DPCG_DEVICE && !copy && aligned16);
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
void free_plan(SpmvPlan &plan) {
    dev_free(plan.tile_chunks);
    dev_free(plan.tile_nchunks);
    dev_free(plan.tile_lidx);
    plan = SpmvPlan();
}

int make_plan(const CsrDev &A, SpmvPlan &plan, hipStream_t s, bool allow_tile) {
    free_plan(plan);
    int *d_max = nullptr;
    DPCG_TRY(dev_alloc(&d_max, 1));
    DPCG_HIP(hipMemsetAsync(d_max, 0, sizeof(int), s));
    launch_block_nnz_max(A, kStreamRows, d_max, s);
    int h_max = 0;
    DPCG_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    if (A.n <= chip_max_rows()) {                        // (only the whole-solve kernels of mid-size and cache-sized systems ask)
        int *d_bl = nullptr, h_bl[2] = {0, 0};
        DPCG_TRY(dev_alloc(&d_bl, 2));
        DPCG_HIP(hipMemsetAsync(d_bl, 0, 2 * sizeof(int), s));
        launch_band_and_len(A, d_bl, s);
        DPCG_HIP(hipMemcpyAsync(h_bl, d_bl, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        dev_free(d_bl);
        plan.max_band = h_bl[0];
        plan.max_row_len = h_bl[1];
    }
    dev_free(d_max);
    const char *force = getenv("DPCG_SPMV_KERNEL");
    const bool force_vector = force && strcmp(force, "vector") == 0;
    if (h_max <= kStreamCap - 1 && !force_vector) {   // - 1: the alignment slot of the pair loads
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
            int *d_flags = nullptr, h_flags[3] = {1, 0, 0};    // every block planned | widest tile | blocks without a tile
            DPCG_TRY(dev_alloc(&plan.tile_chunks, (int64_t)plan.nrb * kTileMaxChunks));
            DPCG_TRY(dev_alloc(&plan.tile_nchunks, plan.nrb));
            DPCG_TRY(dev_alloc(&plan.tile_lidx, A.nnz + 4));                   // read in aligned groups of 2 or 4
            DPCG_HIP(hipMemsetAsync(plan.tile_lidx + A.nnz, 0, 4 * sizeof(uint16_t), s));
            DPCG_TRY(dev_alloc(&d_flags, 3));
            DPCG_HIP(hipMemcpyAsync(d_flags, h_flags, sizeof(h_flags), hipMemcpyHostToDevice, s));
            launch_tile_plan(A, plan.nrb, plan.tile_chunks, plan.tile_nchunks, plan.tile_lidx, d_flags, s);
            DPCG_HIP(hipMemcpyAsync(h_flags, d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            dev_free(d_flags);
            // Blocks that touch more than kTileMaxChunks chunks can gather instead (k_spmv_tile<..., MIX>, same bits).  OFF by default
            // (DPCG_TILE_MIX_MAX = largest share of such blocks in percent): on the system that motivated it -- the 1M-row quadtree
            // mesh in OpenFOAM's numbering, 12 % of the blocks at 41-51 chunks, the others at 25 on average -- the plan is SLOWER than
            // the gather kernel (19.7 vs 16.7 us: tiles of 40 chunks stage 6.4 x the vector and leave 4 workgroups per CU); what that
            // system wants is the reordering (14.2 us).  profiles/r04_mesh_probe_mix.txt
            const char *mix_env = getenv("DPCG_TILE_MIX_MAX");
            const int mix_max = mix_env ? atoi(mix_env) : 0;
            const bool mix_ok = h_flags[2] == 0 || (int64_t)h_flags[2] * 100 <= (int64_t)mix_max * plan.nrb;
            if (h_flags[0] == 1 && h_flags[1] > 0 && mix_ok) {
                plan.kernel = SPMV_TILE;
                plan.tile_max_chunks = h_flags[1];
                plan.tile_mixed = h_flags[2] > 0;
                // streams that cannot stay in the 256 MiB Infinity Cache are read (and y written) non-temporally
                // (DPCG_SPMV_NT=0/1 overrides: development knob)
                static const int nt_knob = [] { const char *e = getenv("DPCG_SPMV_NT"); return e ? atoi(e) : -1; }();
                plan.stream_nt = nt_knob >= 0 ? nt_knob != 0 : stream_bytes >= 512e6;
                // ... and walked by the whole grid together (row blocks dealt out cyclically) instead of in slabs
                // (DPCG_SPMV_CYCLIC=0/1 overrides: development knob)
                static const int cyc_knob = [] { const char *e = getenv("DPCG_SPMV_CYCLIC"); return e ? atoi(e) : -1; }();
                plan.cyclic = cyc_knob >= 0 ? cyc_knob : (plan.stream_nt ? 2 : 0);     // 2: an XCD's blocks of a pass are one run (256^3: 272 -> 267 us)
                const size_t lds = (size_t)(h_flags[1] * kTileChunk + kStreamCap + 8) * sizeof(double);
                // (cyclic: THREE workgroups per CU -- with the grid walking together, each workgroup's own one-block-ahead prefetch
                // carries the latency, and fewer workgroups keep the window the chip reads at any instant narrow; 256^3, us per
                // launch at 2 / 3 / 4 / 5 / 6 / 8 per CU: 299 / 272 / 284 / 292 / 293 / 294, slabs at 6: 298 -- profiles/r04_spmv_cyclic_ab.txt)
                static const int wg_knob = [] { const char *e = getenv("DPCG_SPMV_WG_PER_CU"); return e ? atoi(e) : 0; }();   // development
                const int want = wg_knob >= 1 ? wg_knob : (plan.cyclic ? 3 : 8);
                const int per_cu = (int)std::min<size_t>((size_t)want, (160 * 1024) / lds);
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
        int tpr = 2;                       // lanes per row; a lane takes two non-zeros per trip
        while (tpr < 64 && 2 * tpr < mean) tpr *= 2;
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
ExtrasRegistry &extras() {
    static ExtrasRegistry r;
    return r;
}

void drop_graph(dpcg_system *h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->graph_exec = nullptr;
    h->graph_key = -1;
}

void free_parked(dpcg_system *h) {
    dpcg_system::Parked &p = h->parked;
    free_csr(p.L);
    free_csr(p.Lt);
    free_levels(p.lvlL);
    free_levels(p.lvlU);
    dev_free(p.map_al);
    dev_free(p.t_order);
    dev_free(p.rows_l);
    dev_free(p.rows_u);
    p.valid = false;
    p.colors = 0;
}

// dpcg_update_values: an IC(0) in multicolour order applied by colour sweeps keeps everything its PATTERN determined
// (dpcg_system::Parked); the handle has no preconditioner until the next setup, which then only computes values.
static void park_precond(dpcg_system *h) {
    static const bool on = [] { const char *e = getenv("DPCG_PARK_IC0"); return !(e && e[0] == '0'); }();
    if (!on || h->precond != DPCG_PRECOND_LLT_SOLVE || !h->fmap || h->fmap != h->mc_perm) return;
    if (!h->lvlL.sweep || !h->lvlU.sweep || h->lvlL.n_levels < 1) return;
    dpcg_system::Parked &p = h->parked;
    p.L = h->L;          h->L = CsrDev();
    p.Lt = h->Lt;        h->Lt = CsrDev();
    p.lvlL = h->lvlL;    h->lvlL = Levels();
    p.lvlU = h->lvlU;    h->lvlU = Levels();
    p.colors = h->precond_colors;
    p.valid = true;       // (map_al / t_order / rows_*: kept from an earlier refresh, or built at the first one)
}

void free_precond(dpcg_system *h, bool keep_parked) {
    if (!keep_parked) free_parked(h);
    drop_graph(h);
    free_ell(h->ell_m);
    free_ell(h->ell_t);
    dev_free(h->dinv);
    free_csr(h->M);
    free_csr(h->L);
    free_csr(h->Lt);
    free_csr(h->Lp);
    free_csr(h->Ltp);
    free_levels(h->lvlL);
    free_levels(h->lvlU);
    free_plan(h->planM);
    free_plan(h->planL);
    free_plan(h->planLt);
    free_chip_trsv_lists(h->trsv_l);
    free_chip_trsv_lists(h->trsv_u);
    dev_free(h->trsv_lv0);
    dev_free(h->trsv_diag0);
    dev_free(h->trsv_fval);
    dev_free(h->trsv_fcol);
    dev_free(h->trsv_fmeta);
    h->trsv_rpt = h->trsv_wmax = h->trsv_band = 0;
    h->trsv_state = 0;
    if (h->fmap != h->mc_perm) dev_free(h->fmap);            // (the handle's cached colouring is not the preconditioner's to free)
    if (h->fmap_inv != h->mc_iperm) dev_free(h->fmap_inv);
    h->fmap = h->fmap_inv = nullptr;
    h->precond_colors = 0;
    h->precond_fn = nullptr;
    h->precond_user = nullptr;
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
    SetupScope scope(s);
    dpcg_system *h = new dpcg_system();
    PhaseTimer pt(s);
    int st = upload_csr(h->A, n, nnz, rowptr, col, val, val_dtype, memspace, copy, s);
    pt.mark("create: upload");
    if (st >= 0) st = make_plan(h->A, h->planA, s, true);
    pt.mark("create: plan");
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
    SetupScope scope(nullptr, true);                      // (waits for the device: the arrays may be in use on any stream)
    free_precond(h);
    free_csr(h->A);
    free_csr(h->A_user);
    dev_free(h->mc_perm); dev_free(h->mc_iperm);
    dev_free(h->perm_val_map);
    dev_free(h->perm); dev_free(h->iperm); dev_free(h->pb); dev_free(h->pxt); dev_free(h->pv0); dev_free(h->pv1);
    dev_free(h->pf0); dev_free(h->pf1);
    free_plan(h->planA);
    free_ell(h->ell_a);
    dev_free(h->x); dev_free(h->r); dev_free(h->z); dev_free(h->p); dev_free(h->p2); dev_free(h->q); dev_free(h->t); dev_free(h->e);
    dev_free(h->p32);
    dev_free(h->part_pq); dev_free(h->part_rz); dev_free(h->part_rr); dev_free(h->part_bb);
    dev_free(h->scal); dev_free(h->hist); dev_free(h->err_hist); dev_free(h->small_desc);
    dev_free(h->team_desc); dev_free(h->team_part); dev_free(h->team_sync);
    dev_free(h->chip_part); dev_free(h->chip_zp); dev_free(h->chip_rt);
    if (h->scal_host) (void)hipHostFree(h->scal_host);
    HandleExtras ex;
    if (extras().take(h, ex)) {
        if (ex.cap_stream) (void)hipStreamDestroy(ex.cap_stream);
        if (ex.prog_host) (void)hipHostFree(ex.prog_host);
    }
    delete h;
    return DPCG_OK;
}

// ------------------------------------------------------------------------------------------------
// reordering (see include/dpcg.h and dpcg_reorder.hip)
// ------------------------------------------------------------------------------------------------
extern "C" int dpcg_reorder(dpcg_handle_t h, int mode, dpcg_stream_t stream, int *applied) {
    if (!h) return invalid("dpcg_reorder: NULL handle");
    if (applied) *applied = h->perm ? 1 : 0;
    if (mode == DPCG_REORDER_NONE || h->perm) return DPCG_OK;
    if (mode != DPCG_REORDER_AUTO && mode != DPCG_REORDER_ALWAYS && mode != DPCG_REORDER_REGIONS) return invalid("dpcg_reorder: bad mode");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);
    // Everything that can fail is built FIRST, into locals; the handle is switched over only when all of it exists
    // (a failure half way would otherwise leave the old plan on the new matrix).
    auto adopt = [&](int32_t *perm, int32_t *iperm, CsrDev &B, SpmvPlan &planB) {
        free_precond(h);                 // an attached preconditioner referred to the old matrix
        dev_free(h->mc_perm);            // ... and so did a cached colouring
        dev_free(h->mc_iperm);
        h->mc_colors = 0;
        free_ell(h->ell_a);
        drop_graph(h);
        dev_free(h->A.val32);            // recreated on demand from the reordered values
        h->A.val32_lossless = 0;
        free_plan(h->planA);
        h->A_user = h->A;                // ownership (or the borrow) moves with the struct
        h->A = B;
        h->planA = planB;
        h->perm = perm;
        h->iperm = iperm;
        if (applied) *applied = 1;
    };
    // order -> P A P^T -> its SpMV plan, all in locals; *keep = false (AUTO by regions) when the plan is not the x-tile one
    auto build = [&](bool by_regions, bool tile_or_nothing, bool *keep) -> int {
        PhaseTimer pt(s);
        int32_t *perm = nullptr, *iperm = nullptr;
        int st = by_regions ? region_order(h->A, region_order_default_regions(h->A.n), &perm, &iperm, s) : rcm_order(h->A, &perm, &iperm, nullptr, s);
        pt.mark(by_regions ? "reorder: regions" : "reorder: RCM");
        if (st < 0) return st;
        CsrDev B;
        SpmvPlan planB;
        st = permute_csr(h->A, perm, iperm, B, s);
        pt.mark("reorder: permute");
        if (st >= 0) st = make_plan(B, planB, s, true);
        pt.mark("reorder: plan");
        *keep = st >= 0 && !(tile_or_nothing && planB.kernel != SPMV_TILE);
        if (!*keep) {
            dev_free(perm);
            dev_free(iperm);
            free_csr(B);
            free_plan(planB);
            return st;
        }
        adopt(perm, iperm, B, planB);
        return DPCG_OK;
    };
    bool kept = false;
    if (mode == DPCG_REORDER_AUTO) {
        // only where it pays: systems beyond one XCD's L2 reach whose plan is the gather kernel (no x-tile plan, or too few row
        // blocks for one) and
        //  (a) whose gather really is scattered (measured 64^3 scrambled: 31K -> 55K it/s): reverse Cuthill-McKee; or
        //  (b) whose gather looks fine on average but whose x-tile plan was REFUSED -- some row blocks touch more than
        //      kTileMaxChunks chunks of x (an OpenFOAM numbering: refinement appends cells) -- : the region-by-region numbering
        //      (dpcg_reorder.hip: a few ms where RCM walks 2000 levels), kept only when the x-tile plan takes the result.
        // DPCG_REORDER_REGIONS = 0: never by regions; 2: by regions ahead of RCM in case (a) too (development knob).
        if (h->planA.kernel != SPMV_STREAM || h->A.n < kReorderMinRows) return DPCG_OK;
        PhaseTimer ptg(s);
        DPCG_TRY(gather_line_ratio(h->A, &h->gather_ratio, s));
        ptg.mark("reorder: gather ratio");
        static const int regions_knob = [] { const char *e = getenv("DPCG_REORDER_REGIONS"); return e ? atoi(e) : 1; }();
        const bool scattered = h->gather_ratio > 4.0;
        const bool tile_refused = h->planA.nrb >= kTileMinBlocks && !getenv("DPCG_SPMV_KERNEL");
        // (a banded system of short rows is solved in one launch by the whole chip, dpcg_chip.hip: a region numbering would cost it its band)
        const bool chip_shape = h->A.n <= chip_max_rows() && h->planA.max_row_len >= 1 && h->planA.max_row_len <= chip_max_row_len(h->A.n) &&
                                h->planA.max_band >= 0 && h->planA.max_band <= chip_max_band();
        if (tile_refused && !chip_shape && regions_knob >= (scattered ? 2 : 1)) {
            const int st = build(true, true, &kept);
            if (st < 0 && st != DPCG_ERR_INVALID) return st;
            if (kept) return DPCG_OK;
        }
        if (!scattered) return DPCG_OK;
    }
    const int st = build(mode == DPCG_REORDER_REGIONS, false, &kept);
    if (st == DPCG_ERR_INVALID && mode == DPCG_REORDER_AUTO) return DPCG_OK;   // not a symmetric pattern: AUTO leaves the handle as it is
    return st;
}

// New values on the pattern the handle was created with (the next pressure system of the same mesh): everything that
// depends only on the pattern stays -- the SpMV plan with its x-tile lists, the reordering -- and everything that holds
// values is refreshed or dropped (the permuted copy, the fp32 copy, the slab-ELL slices, the captured graphs, the
// preconditioner: attach one again).
extern "C" int dpcg_update_values(dpcg_handle_t h, const void *val, int val_dtype, int memspace, dpcg_stream_t stream) {
    if (!h || !val) return invalid("dpcg_update_values: NULL handle or values");
    if (val_dtype != DPCG_F64 && val_dtype != DPCG_F32) return invalid("dpcg_update_values: bad val_dtype");
    if (memspace != DPCG_HOST && memspace != DPCG_DEVICE) return invalid("dpcg_update_values: bad memspace");
    hipStream_t s = (hipStream_t)stream;
    SetupScope scope(s, true);           // (the old values may be in use on another stream)
    CsrDev &U = h->perm ? h->A_user : h->A;      // the matrix in the caller's numbering
    const int64_t nnz = U.nnz;
    bool val32_fresh = false;                    // U.val32 holds the NEW values in fp32
    if (!U.owned) {
        // borrowed arrays (dpcg_create with copy = 0): the new values are borrowed the same way -- possibly the same buffer,
        // rewritten in place
        if (memspace != DPCG_DEVICE || val_dtype != DPCG_F64 || (((uintptr_t)val) & 15) != 0)
            return invalid("dpcg_update_values: a handle that borrows its arrays takes 16-byte aligned fp64 device values");
        U.val = const_cast<double *>((const double *)val);
    } else {
        const hipMemcpyKind kind = memspace == DPCG_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        if (val_dtype == DPCG_F64) {
            DPCG_HIP(hipMemcpyAsync(U.val, val, (size_t)nnz * sizeof(double), kind, s));
        } else {
            if (!U.val32) DPCG_TRY(dev_alloc(&U.val32, nnz));
            DPCG_HIP(hipMemcpyAsync(U.val32, val, (size_t)nnz * sizeof(float), kind, s));
            launch_f32_to_f64(nnz, U.val32, U.val, s);
            val32_fresh = true;
        }
        DPCG_HIP(hipStreamSynchronize(s));       // a host source buffer may be released by the caller
    }
    if (h->perm) {                               // P A P^T again: same pattern, so only its values move
        if (!h->perm_val_map) {
            // which entry of the caller's matrix each entry of P A P^T is: entry NUMBERS sent through the permutation once
            double *ids = nullptr;
            DPCG_TRY(dev_alloc(&ids, nnz));
            launch_iota_f64(nnz, ids, s);
            CsrDev Uid = h->A_user, B;
            Uid.val = ids;
            Uid.val32 = nullptr;
            Uid.owned = false;
            int st = permute_csr(Uid, h->perm, h->iperm, B, s);
            if (st >= 0) st = dev_alloc(&h->perm_val_map, nnz);
            if (st >= 0) launch_f64_to_i32(nnz, B.val, h->perm_val_map, s);
            free_csr(B);
            dev_free(ids);
            DPCG_TRY(st);
        }
        launch_gather_f64(nnz, h->perm_val_map, h->A_user.val, h->A.val, s);
        dev_free(h->A.val32);                    // (a copy of the old values)
        if (!val32_fresh) dev_free(h->A_user.val32);
    } else if (!val32_fresh) {
        dev_free(h->A.val32);                    // an fp32 original or copy of the OLD values: made again on demand
    }
    h->A.val32_lossless = 0;                     // decided again on demand
    if (!h->parked.valid) park_precond(h);
    free_precond(h, true);
    free_ell(h->ell_a);
    drop_graph(h);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_get_permutation(dpcg_handle_t h, int *reordered, int32_t *perm_host, double *gather_ratio) {
    if (!h) return invalid("dpcg_get_permutation: NULL handle");
    if (reordered) *reordered = h->perm ? 1 : 0;
    if (gather_ratio) *gather_ratio = h->gather_ratio;
    if (perm_host && h->perm)
        DPCG_HIP(hipMemcpy(perm_host, h->perm, (size_t)h->A.n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return DPCG_OK;
}

// Two-kernel updates (fused_head in dpcg_device.h) trade one kernel boundary and one pass over p for a second
// gather per non-zero.  Measured with tools/fuse_probe.py (Jacobi PCG, its/s, two- vs three-kernel): 16K rows
// +18 %, 65K +20 %, 147K +12 %, 262K +8 %, 512K -6 %, 1M -8 % (scrambled 1M: -48 %).  So: systems below the x-tile
// threshold, where an update is launch-bound rather than bandwidth-bound.
bool fuse_eligible(const dpcg_system *h, int flags, const double *x_true) {
    static const int64_t max_rows = [] {
        const char *e = getenv("DPCG_FUSE_MAX_ROWS");
        return e ? (int64_t)atoll(e) : (int64_t)kTileMinBlocks * kStreamRows;
    }();
    if (x_true || (flags & (DPCG_NO_FUSE | DPCG_SPMV_F32))) return false;
    if ((flags & DPCG_VAL32_IF_LOSSLESS) && h->A.val32_lossless == 1) return false;
    if (h->A.n >= max_rows) return false;
    return h->planA.kernel == SPMV_STREAM || h->planA.kernel == SPMV_TILE;
}

FuseArgs fuse_args(dpcg_system *h) {
    FuseArgs fa;
    fa.z = h->precond == DPCG_PRECOND_NONE ? h->r : h->z;
    fa.p0 = h->p;
    fa.p1 = h->p2;
    fa.xvec = h->x;
    fa.part_rz = h->part_rz;
    fa.part_rr = h->part_rr;
    fa.n_part = rz_partial_count(h);
    fa.n_part_rr = h->vec_grid;
    fa.hist = h->hist;
    fa.hist_cap = h->hist_cap;
    return fa;
}

extern "C" int dpcg_get_info(dpcg_handle_t h, int64_t *n, int64_t *nnz, int *spmv_kernel, int *precond_kind,
                             int64_t *precond_nnz, int *n_levels_lower, int *n_levels_upper) {
    if (!h) return invalid("dpcg_get_info: NULL handle");
    if (n) *n = h->A.n;
    if (nnz) *nnz = h->A.nnz;
    if (spmv_kernel)   // +16: two-kernel updates; x-tile kernel: +32 non-temporal streams, +64 some blocks gather, +128 cyclic row blocks
        *spmv_kernel = h->planA.kernel + (fuse_eligible(h, 0, nullptr) ? 16 : 0) +
                       (h->planA.kernel == SPMV_TILE && h->planA.stream_nt ? 32 : 0) +
                       (h->planA.kernel == SPMV_TILE && h->planA.tile_mixed ? 64 : 0) +
                       (h->planA.kernel == SPMV_TILE && h->planA.cyclic ? 128 : 0);
    if (precond_kind) *precond_kind = h->precond;
    if (precond_nnz) *precond_nnz = h->precond == DPCG_PRECOND_CSR ? h->M.nnz : h->L.nnz;
    if ((n_levels_lower || n_levels_upper) && h->lvlL.n_levels < 0) DPCG_TRY(count_levels_on_demand(h));
    if (n_levels_lower) *n_levels_lower = h->lvlL.n_levels;
    if (n_levels_upper) *n_levels_upper = h->lvlU.n_levels;
    return DPCG_OK;
}

int ensure_work(dpcg_system *h, int max_iter, bool need_f32, bool need_err) {
    const int64_t n = h->A.n;
    if (!h->x) {
        DPCG_TRY(dev_alloc(&h->x, n));
        DPCG_TRY(dev_alloc(&h->r, n));
        DPCG_TRY(dev_alloc(&h->z, n));
        DPCG_TRY(dev_alloc(&h->p, n));
        DPCG_TRY(dev_alloc(&h->q, n));
        DPCG_TRY(dev_alloc(&h->t, n));
        DPCG_TRY(dev_alloc(&h->part_pq, kMaxSpmvGrid));
        DPCG_TRY(dev_alloc(&h->part_rz, kMaxSpmvGrid));   // an M-apply's SpMV may leave up to its grid's worth of <r,z> partials
        DPCG_TRY(dev_alloc(&h->part_rr, kMaxSpmvGrid));
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
            DPCG_HIP(device_wide_wait());
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
// standalone operators
// ------------------------------------------------------------------------------------------------
// scratch vectors of the standalone operators on a reordered handle (the caller's vectors keep the caller's numbering)
static int ensure_perm_scratch(dpcg_system *h, bool f32) {
    if (!h->pv0) DPCG_TRY(dev_alloc(&h->pv0, h->A.n));
    if (!h->pv1) DPCG_TRY(dev_alloc(&h->pv1, h->A.n));
    if (f32) {
        if (!h->pf0) DPCG_TRY(dev_alloc(&h->pf0, h->A.n));
        if (!h->pf1) DPCG_TRY(dev_alloc(&h->pf1, h->A.n));
    }
    return DPCG_OK;
}

extern "C" int dpcg_spmv(dpcg_handle_t h, const double *x, double *y, dpcg_stream_t stream) {
    if (!h || !x || !y) return invalid("dpcg_spmv: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    if (h->perm) {                                   // y = P^T (P A P^T) P x
        DPCG_TRY(ensure_perm_scratch(h, false));
        launch_gather_f64(h->A.n, h->perm, x, h->pv0, s);
        launch_spmv(h->A, h->planA, h->pv0, h->pv1, nullptr, nullptr, s);
        launch_scatter_f64(h->A.n, h->perm, h->pv1, y, s);
    } else {
        launch_spmv(h->A, h->planA, x, y, nullptr, nullptr, s);
    }
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_spmv_f32(dpcg_handle_t h, const float *x, float *y, dpcg_stream_t stream) {
    if (!h || !x || !y) return invalid("dpcg_spmv_f32: NULL argument");
    if (!h->A.val32) {
        DPCG_TRY(dev_alloc(&h->A.val32, h->A.nnz));
        launch_f64_to_f32(h->A.nnz, h->A.val, h->A.val32, (hipStream_t)stream);
    }
    if (h->perm) {
        DPCG_TRY(ensure_perm_scratch(h, true));
        launch_gather_f32(h->A.n, h->perm, x, h->pf0, (hipStream_t)stream);
        launch_spmv_f32out(h->A, h->planA, h->pf0, h->pf1, (hipStream_t)stream);
        launch_scatter_f32(h->A.n, h->perm, h->pf1, y, (hipStream_t)stream);
    } else {
        launch_spmv_f32out(h->A, h->planA, x, y, (hipStream_t)stream);
    }
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

// z = M r for the handle's preconditioner (cg.py:61,81).  `t` is the handle's scratch vector.
int rz_partial_count(const dpcg_system *h) {
    if (h->precond == DPCG_PRECOND_CSR) return h->planM.grid;
    if (h->precond == DPCG_PRECOND_LLT_MULTIPLY) return h->planL.grid;
    if (h->precond == DPCG_PRECOND_LLT_SOLVE && h->lvlU.sweep) return h->lvlU.sweep_grid;   // colour sweeps
    return h->vec_grid;
}

int apply_precond(dpcg_system *h, const double *r, double *z, hipStream_t s, bool in_loop, double *part_rz, int *n_part_rz,
                  bool lower_first_done) {
    if (n_part_rz) *n_part_rz = 0;
    switch (h->precond) {
        case DPCG_PRECOND_NONE:
            if (z != r) DPCG_HIP(hipMemcpyAsync(z, r, (size_t)h->A.n * sizeof(double), hipMemcpyDeviceToDevice, s));
            break;
        case DPCG_PRECOND_JACOBI:
            launch_scale(h->A.n, h->dinv, r, z, h->vec_grid, s);
            break;
        case DPCG_PRECOND_CSR:                 // the SpMV sums <r, M r> on the way when asked to
            if (part_rz && n_part_rz) {
                launch_spmv_xdot(h->M, h->planM, r, r, z, part_rz, s);
                *n_part_rz = h->planM.grid;
            } else {
                launch_spmv(h->M, h->planM, r, z, nullptr, nullptr, s);
            }
            break;
        case DPCG_PRECOND_LLT_MULTIPLY:       // on a reordered handle the SpMVs read P L^T P^T and P L P^T
            launch_spmv(h->perm ? h->Ltp : h->Lt, h->planLt, r, h->t, nullptr, nullptr, s);
            if (part_rz && n_part_rz) {
                launch_spmv_xdot(h->perm ? h->Lp : h->L, h->planL, h->t, r, z, part_rz, s);
                *n_part_rz = h->planL.grid;
            } else {
                launch_spmv(h->perm ? h->Lp : h->L, h->planL, h->t, z, nullptr, nullptr, s);
            }
            break;
        case DPCG_PRECOND_CALLBACK:
            if (h->perm) {                        // the caller's function sees the caller's numbering; t and q (dead
                launch_scatter_f64(h->A.n, h->perm, r, h->t, s);          // between K2 and the next K1) are the scratch
                h->precond_fn(h->precond_user, h->t, h->q, h->A.n, (dpcg_stream_t)s);
                launch_gather_f64(h->A.n, h->perm, h->q, z, s);
            } else {
                h->precond_fn(h->precond_user, r, z, h->A.n, (dpcg_stream_t)s);
            }
            break;
        case DPCG_PRECOND_LLT_SOLVE: {
            SptrsvIo lower_io, upper_io;
            if (h->lvlL.level_major && h->lvlU.level_major && h->lvlU.lm_from_lower) {
                lower_io.keep_lm = true;                   // y stays in L's level-major numbering, L^T gathers it from there
                upper_io.lm_in = h->lvlL.lm_out;
                if (single_syncfree_segment(h->lvlL) && single_syncfree_segment(h->lvlU)) {
                    // no way-in passes: the two solve kernels gather their right-hand sides themselves; the lower solve presets
                    // L^T's solution vector to "pending", the way-out pass does the same for L's (for the next apply)
                    lower_io.fused_entry = upper_io.fused_entry = true;
                    lower_io.refill = h->lvlU.lm_out;
                    upper_io.refill = h->lvlL.lm_out;
                }
            }
            if (part_rz && n_part_rz) {                    // <r,z> summed by the kernel that takes z out of level-major order
                upper_io.dot_with = r;
                upper_io.dot_part = part_rz;
                upper_io.dot_grid = h->vec_grid;
            }
            if (lower_io.keep_lm && h->lvlL.sweep && h->lvlU.sweep && h->lvlL.lm_to_upper) {
                // colour sweeps: the last lower level opens the upper solve (its rows are L^T's first level)
                lower_io.pair_out = h->lvlU.lm_out;
                lower_io.pair_dst = z;
                lower_io.dot_with = upper_io.dot_with;
                lower_io.dot_part = upper_io.dot_part;
                upper_io.skip_first = true;
            }
            lower_io.skip_first = lower_first_done;        // (the first lower level rode on K2: ride_eligible)
            launch_sptrsv(h->L, h->lvlL, false, r, h->t, s, in_loop ? &h->scal->done : nullptr, &lower_io);
            launch_sptrsv(h->Lt, h->lvlU, true, h->t, z, s, in_loop ? &h->scal->done : nullptr, &upper_io);
            if (upper_io.dot_done) *n_part_rz = upper_io.dot_count;
            break;
        }
        default:
            set_error("unknown preconditioner kind");
            return DPCG_ERR_STATE;
    }
    return DPCG_OK;
}

extern "C" int dpcg_precond_apply(dpcg_handle_t h, const double *r, double *z, dpcg_stream_t stream) {
    if (!h || !r || !z) return invalid("dpcg_precond_apply: NULL argument");
    DPCG_TRY(ensure_work(h, 0, false, false));
    if (h->perm) {
        hipStream_t s = (hipStream_t)stream;
        DPCG_TRY(ensure_perm_scratch(h, false));
        launch_gather_f64(h->A.n, h->perm, r, h->pv0, s);
        DPCG_TRY(apply_precond(h, h->pv0, h->pv1, s));
        launch_scatter_f64(h->A.n, h->perm, h->pv1, z, s);
    } else {
        DPCG_TRY(apply_precond(h, r, z, (hipStream_t)stream));
    }
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_sptrsv(dpcg_handle_t h, int upper, const double *rhs, double *out, dpcg_stream_t stream) {
    if (!h || !rhs || !out) return invalid("dpcg_sptrsv: NULL argument");
    if (h->precond != DPCG_PRECOND_LLT_SOLVE) {
        set_error("dpcg_sptrsv: needs dpcg_set_precond_llt/ic0 in LLT_SOLVE mode");
        return DPCG_ERR_STATE;
    }
    hipStream_t s = (hipStream_t)stream;
    const double *in = rhs;
    double *res = out;
    if (h->perm) {       // the schedules are relabelled to the handle's numbering; the factor itself is the caller's
        DPCG_TRY(ensure_perm_scratch(h, false));
        launch_gather_f64(h->A.n, h->perm, rhs, h->pv0, s);
        in = h->pv0;
        res = h->pv1;
    }
    if (upper) launch_sptrsv(h->Lt, h->lvlU, true, in, res, s);
    else launch_sptrsv(h->L, h->lvlL, false, in, res, s);
    if (h->perm) launch_scatter_f64(h->A.n, h->perm, h->pv1, out, s);
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
        DPCG_HIP(hipMemsetAsync(h->part_rz, 0, kMaxSpmvGrid * sizeof(double), s));
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

// The streaming ceiling of THIS box, measured with the library's own kernel shape (16-byte lane accesses, XCD-contiguous
// slabs, 2048 workgroups): per 16 bytes written, n_read x 16 contiguous bytes are read and summed (write = 1), or the reads
// are only reduced (write = 0, out_bytes then sizes the read stream: n_read x out_bytes).  What an HBM-bound kernel of this
// library can at best reach on the box it runs on.
extern "C" int dpcg_stream_bench(int n_read, int write, int nontemporal, int64_t out_bytes, int repeats, float *ms_per_launch,
                                 int64_t *bytes_per_launch, dpcg_stream_t stream) {
    if (out_bytes < 16 || repeats <= 0 || !ms_per_launch || n_read < 1) return invalid("dpcg_stream_bench: bad argument");
    hipStream_t s = (hipStream_t)stream;
    double *in = nullptr, *out = nullptr, *part = nullptr;
    const int64_t n = out_bytes / 8;
    const int64_t in_doubles = n * n_read + 2 * kBlock * 2 * n_read * 2;          // one tile of slack behind the last slab
    DPCG_TRY(dev_alloc(&in, in_doubles));
    int st = dev_alloc(&out, write ? n + 4 * kBlock : 1);
    if (st >= 0) st = dev_alloc(&part, kMaxSpmvGrid);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    int64_t moved = 0;
    hipError_t e = hipSuccess;
    if (st >= 0) {
        e = hipMemsetAsync(in, 0, (size_t)in_doubles * sizeof(double), s);
        if (e == hipSuccess) e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        // a footprint inside the Infinity Cache (256 MiB) streams best with one element in flight per lane and 6 workgroups
        // per CU, one beyond it with two and 8 (tools/stream_lab; profiles/r03_stream_lab.txt)
        const bool resident = (int64_t)(n_read + (write ? 1 : 0)) * out_bytes < ((int64_t)200 << 20);
        // nontemporal: bit 0 = non-temporal accesses, bit 1 = the streams WALKED TOGETHER instead of in slabs (k_stream_walk): a copy as
        // one 16-byte element per thread (the guide's float4-copy shape), the other ratios on a persistent grid of 2048
        const bool walk = (nontemporal & 2) != 0;
        const int64_t flat = (out_bytes / 16 + kBlock - 1) / kBlock;
        const int grid = walk ? (int)(n_read == 1 && flat <= 0x7fffffff ? flat : kMaxSpmvGrid) : (resident ? 1536 : kMaxSpmvGrid);
        const int in_flight = walk ? 0 : (resident ? 1 : 2);
        nontemporal &= 1;
        for (int i = 0; i < 2 && e == hipSuccess; ++i)
            moved = launch_stream_bench(n_read, write != 0, nontemporal != 0, out_bytes, in, out, part, grid, s, in_flight);
        if (e == hipSuccess && moved < 0) st = invalid("dpcg_stream_bench: n_read must be 1, 2, 4 or 11");
        if (e == hipSuccess && st >= 0) {
            e = hipEventRecord(e0, s);
            for (int i = 0; i < repeats; ++i)
                launch_stream_bench(n_read, write != 0, nontemporal != 0, out_bytes, in, out, part, grid, s, in_flight);
            if (e == hipSuccess) e = hipEventRecord(e1, s);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e == hipSuccess) e = hipGetLastError();
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    dev_free(in);
    dev_free(out);
    dev_free(part);
    DPCG_HIP(e);
    if (st < 0) return st;
    *ms_per_launch = ms / (float)repeats;
    if (bytes_per_launch) *bytes_per_launch = moved;
    return DPCG_OK;
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

// Panels of `ncols` right-hand sides (the sparse `inverse_loss`, metrics.py:34-55)
extern "C" int dpcg_batched_coo_spmm(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof,
                                     int ncols, const float *panel, float *out, int transpose, dpcg_stream_t stream) {
    if (nnz < 0 || batch <= 0 || dof <= 0 || ncols <= 0 || !panel || !out || (nnz > 0 && (!indices || !features)))
        return invalid("dpcg_batched_coo_spmm: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DPCG_HIP(hipMemsetAsync(out, 0, (size_t)batch * (size_t)dof * (size_t)ncols * sizeof(float), s));
    if (nnz > 0) launch_batched_coo_spmm(nnz, indices, features, batch, dof, ncols, panel, out, transpose, s);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_batched_coo_sddmm(int64_t nnz, const int32_t *indices, int batch, int64_t dof, int ncols, const float *g,
                                      const float *panel, float *out, int transpose, dpcg_stream_t stream) {
    if (nnz < 0 || batch <= 0 || dof <= 0 || ncols <= 0 || (nnz > 0 && (!indices || !g || !panel || !out)))
        return invalid("dpcg_batched_coo_sddmm: bad arguments");
    if (nnz > 0) launch_batched_coo_sddmm(nnz, indices, batch, dof, ncols, g, panel, out, transpose, (hipStream_t)stream);
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}
