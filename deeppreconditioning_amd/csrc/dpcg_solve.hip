// Host side of libdpcg.so, part 3: the PCG driver -- enqueues the iteration kernels (as replayed hipGraph chunks or
// update by update) ahead of a progress word the GPU posts to pinned memory, the whole-solve kernel for small systems,
// and the batch entry point.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "dpcg_host.h"
#include "dpcg_prims.h"

// ------------------------------------------------------------------------------------------------
// the solve
// ------------------------------------------------------------------------------------------------
static int default_chunk() {
    const char *e = getenv("DPCG_CHUNK");
    int c = e ? atoi(e) : 8;
    return c < 1 ? 1 : (c > 256 ? 256 : c);
}

// The x update is deferred to every second update (k_update_xp_deferred) in the three-kernel form; x_true tracking
// needs x after every update.
static bool defer_x_eligible(const dpcg_system *h, int flags, const double *x_true) {
    static const bool enabled = [] { const char *e = getenv("DPCG_DEFER_X"); return !(e && e[0] == '0'); }();
    return enabled && !x_true && !fuse_eligible(h, flags, x_true);
}

// The vector kernels of a system whose streams exceed the Infinity Cache (the x-tile SpMV reads them non-temporally: 256^3) store what
// they produce non-temporally too.  DPCG_VEC_NT=0/1: development knob.
static bool vec_nt(const dpcg_system *h) {
    static const int knob = [] { const char *e = getenv("DPCG_VEC_NT"); return e ? atoi(e) : -1; }();
    return knob >= 0 ? knob != 0 : (h->planA.kernel == SPMV_TILE && h->planA.stream_nt);
}

// One PCG update (cg.py:75-86) as kernel launches on `s`; `j` = index of this update within the solve (its parity
// selects the p buffer in the deferred-x form: a replayed graph chunk has an even length and starts at an even j).
static int enqueue_iteration(dpcg_system *h, int flags, const double *x_true, hipStream_t s, int j) {
    const int64_t n = h->A.n;
    const bool f32 = (flags & DPCG_SPMV_F32) != 0;
    // colour sweeps: the first level of the lower solve (rows without dependencies: y = r / d) rides on the kernel that updates r
    const bool ride = h->precond == DPCG_PRECOND_LLT_SOLVE && h->lvlL.sweep && h->lvlL.ride_diag && h->lvlL.n_levels >= 2;
    if (fuse_eligible(h, flags, x_true)) {
        // KA: test of the current iterate, p = z + beta p, deferred x += alpha p, q = A p, partials of <p,q>
        launch_spmv_fused(h->A, h->planA, fuse_args(h), h->q, h->part_pq, h->scal, s);          // cg.py:71,83,79,75
        // KB: alpha; r -= alpha q; (z = M r fused); partials <r,z>, <r,r>; k += 1                cg.py:78,80-82,86
        const int pre = h->precond == DPCG_PRECOND_NONE ? 0 : (h->precond == DPCG_PRECOND_JACOBI ? 1 : 2);
        if (ride)
            launch_update_r_ride(n, h->scal, h->part_pq, h->planA.grid, h->q, h->r, h->lvlL.ride_diag, h->lvlL.lm_pos, h->lvlL.lm_out,
                                 h->lvlL.level_ptr[1], h->part_rr, h->vec_grid, s, true);
        else
            launch_update_r_two_kernel(pre, n, h->scal, h->part_pq, h->planA.grid, h->q, h->r, h->dinv, h->z, h->part_rz,
                                       h->part_rr, h->vec_grid, s);
        if (pre == 2) {
            int np = 0;
            DPCG_TRY(apply_precond(h, h->r, h->z, s, true, h->part_rz, &np, ride));              // cg.py:81 (+ cg.py:82 when fused)
            if (np == 0) launch_dot_partials(n, h->scal, h->r, h->z, h->part_rz, h->vec_grid, s);   // cg.py:82
        }
        return DPCG_OK;
    }
    IterCtl ctl{h->scal};
    // K1: (skip when done) Ap = A p + partials of <p,Ap>           cg.py:71,75,78
    const bool v32 = !f32 && (flags & DPCG_VAL32_IF_LOSSLESS) && h->A.val32_lossless == 1;
    const bool defer_x = defer_x_eligible(h, flags, x_true);
    double *p_cur = (defer_x && (j & 1)) ? h->p2 : h->p;               // p_j
    double *p_next = (defer_x && !(j & 1)) ? h->p2 : h->p;             // where p_{j+1} goes (in place without deferral)
    if (f32) launch_spmv_f32in(h->A, h->planA, h->p32, p_cur, h->q, h->part_pq, &ctl, s);
    else if (v32) launch_spmv_val32(h->A, h->planA, p_cur, h->q, h->part_pq, &ctl, s);
    else launch_spmv(h->A, h->planA, p_cur, h->q, h->part_pq, &ctl, s);
    // K2: alpha; r -= alpha Ap; (z = M r fused); partials <r,z>, <r,r>              cg.py:78,80-82,86
    const int pre = h->precond == DPCG_PRECOND_NONE ? 0 : (h->precond == DPCG_PRECOND_JACOBI ? 1 : 2);
    // Jacobi: z = dinv .* r is never stored -- K2 only needs it for <r,z>, and K3 recomputes the same product from r and
    // dinv (same rounding, same bits).  That moves 8 B per row from a write in K2 to a read in K3; writes are the dearer
    // direction (measured: 34.9 -> 33.1 us per update at 1M DoF).
    const bool z_on_the_fly = pre == 1;
    const double *z = pre == 2 ? h->z : h->r;
    if (ride)
        launch_update_r_ride(n, h->scal, h->part_pq, h->planA.grid, h->q, h->r, h->lvlL.ride_diag, h->lvlL.lm_pos, h->lvlL.lm_out,
                             h->lvlL.level_ptr[1], h->part_rr, h->vec_grid, s);
    else
        launch_update_r(pre, n, h->scal, h->part_pq, h->planA.grid, h->q, h->r, h->dinv, h->z, h->part_rz, h->part_rr,
                        h->vec_grid, s, z_on_the_fly ? 0 : 1, vec_nt(h));
    const int np_rz = pre == 2 ? rz_partial_count(h) : h->vec_grid;
    if (pre == 2) {
        int np = 0;
        DPCG_TRY(apply_precond(h, h->r, h->z, s, true, h->part_rz, &np, ride));      // cg.py:81 (+ cg.py:82 when fused)
        if (np == 0) launch_dot_partials(n, h->scal, h->r, h->z, h->part_rz, h->vec_grid, s);   // cg.py:82
    }
    // K3: beta; x += alpha p; p = z + beta p; workgroup 0: stopping test of the new iterate   cg.py:79,82-83,86,71
    if (defer_x)
        launch_update_xp_deferred((j & 1) != 0, n, h->scal, h->part_rz, h->part_rr, np_rz, z, p_cur, p_next, h->x,
                                  f32 ? h->p32 : nullptr, h->hist, h->hist_cap, h->vec_grid, s,
                                  z_on_the_fly ? h->dinv : nullptr, h->vec_grid, vec_nt(h));
    else
        launch_update_xp(n, h->scal, h->part_rz, h->part_rr, np_rz, z, h->p, h->x, f32 ? h->p32 : nullptr, h->hist,
                         h->hist_cap, h->vec_grid, s, z_on_the_fly ? h->dinv : nullptr, h->vec_grid);
    if (x_true) {                                                                    // cg.py:43-45
        launch_anorm_err(n, h->scal, h->x, x_true, h->e, h->vec_grid, s);
        launch_spmv(h->A, h->planA, h->e, h->t, h->part_bb, nullptr, s);
        launch_record_err(h->scal, h->part_bb, h->planA.grid, h->err_hist, h->hist_cap, 0, s);
    }
    return DPCG_OK;
}

static int ensure_graph(dpcg_system *h, int flags, int chunk) {
    const int key = (h->precond << 8) | (flags & (DPCG_SPMV_F32 | DPCG_VAL32_IF_LOSSLESS | DPCG_NO_FUSE)) |
                    (h->A.val32_lossless == 1 ? 64 : 0) | (fuse_eligible(h, flags, nullptr) ? 128 : 0) |
                    (defer_x_eligible(h, flags, nullptr) ? 1024 : 0) | (vec_nt(h) ? 2048 : 0);
    if (h->graph_exec && h->graph_key == key && h->graph_chunk == chunk) return DPCG_OK;
    drop_graph(h);
    HandleExtras &ex = extras()[h];
    hipGraph_t graph = nullptr;
    int st = DPCG_OK;
    hipError_t e = hipSuccess;
    {
        CaptureGuard no_device_wide_waits_meanwhile;          // (another host thread's hipFree / hipDeviceSynchronize would void the capture)
        // (relaxed mode: what OTHER host threads call meanwhile -- torch allocating or synchronising -- is their business; the
        // capture stream itself is private to this handle)
        DPCG_HIP(hipStreamBeginCapture(ex.cap_stream, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < chunk && st >= 0; ++i) st = enqueue_iteration(h, flags, nullptr, ex.cap_stream, i);
        e = hipStreamEndCapture(ex.cap_stream, &graph);
    }
    if (e != hipSuccess) {
        // The capture was voided from outside -- HIP refuses a device-wide wait while ANY stream captures, and void-s the capture
        // too: another host thread of the application calling torch.cuda.synchronize() at that moment does it
        // (tools/thread_probe.py).  Not an error of this solve: it runs update by update, the next one captures again.
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        return DPCG_OK;                                  // (h->graph_exec stays null: the caller checks)
    }
    if (st < 0) {
        if (graph) (void)hipGraphDestroy(graph);
        return st;
    }
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
        if (lv.strips.n_strips > 0) return 2;
        int c = 0;
        for (const auto &seg : lv.segments) c += (seg.merged || seg.syncfree) ? 2 : seg.hi - seg.lo;
        return c;
    };
    switch (h->precond) {
        case DPCG_PRECOND_CSR: return 1;
        case DPCG_PRECOND_CALLBACK: return 1;
        case DPCG_PRECOND_LLT_MULTIPLY: return 2;
        case DPCG_PRECOND_LLT_SOLVE: return trsv(h->lvlL) + trsv(h->lvlU);
        default: return 0;
    }
}

// A sync-free triangular solve whose bounded poll ran out (cannot happen with a schedule built by this library) has
// stored NaNs; report it instead of a silent breakdown.
int check_spin_errors(dpcg_system *h, hipStream_t s) {
    if (h->precond != DPCG_PRECOND_LLT_SOLVE) return DPCG_OK;
    for (Levels *lv : {&h->lvlL, &h->lvlU}) {
        if (!lv->spin_err) continue;
        int e = 0;
        DPCG_HIP(hipMemcpyAsync(&e, lv->spin_err, sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (e) {
            (void)hipMemsetAsync(lv->spin_err, 0, sizeof(int), s);
            // the failed solve stored NaNs: put the "all pending between solves" invariant of the lower factor's level-major
            // solution vector back (Levels::lm_out), so that the handle stays usable
            if (single_syncfree_segment(h->lvlL) && h->lvlL.lm_out) launch_fill_pending(h->lvlL.lm_out, h->A.n, s);
            (void)hipStreamSynchronize(s);
            set_error("sync-free triangular solve: a row waited (4 s) for an entry that was never written");
            return DPCG_ERR_STATE;
        }
    }
    return DPCG_OK;
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
    bool defer_x = false;         // three-kernel updates with x brought up to date every second update
    bool many_launches = false;   // an update is dozens of small launches (level-scheduled SpTRSV): always replay a graph
    bool alternate = false;       // test knob DPCG_DRIVER_ALTERNATE, see enqueue_some
    unsigned calls = 0;
    std::chrono::steady_clock::time_point t0;

    volatile unsigned long long *prog() { return extras()[h].prog_host; }

    int enqueue_some() {
        // Replayed chunks for short updates (the launch overhead is what a chunk amortises) AND for long ones.  (Until round 6 updates longer
        // than 25 us were enqueued launch by launch: nothing to amortise there, it seemed -- but at 256^3 the queue idles 5.9 us between the
        // last launch of one host call and the first of the next (the K3 -> K1 gap of the traces of rounds 4-5: with or without non-temporal
        // stores), and not inside a replayed graph: 521.2 -> 516.8 us per update.  In between -- 25 .. 100 us, the 1M-row launch path, where
        // no such gap shows -- single launches stay: a chunk overshoots the converged solve by up to its length.  DPCG_GRAPH_ALWAYS=0: the old rule.)
        static const bool graph_long = [] { const char *e = getenv("DPCG_GRAPH_ALWAYS"); return !(e && e[0] == '0'); }();
        static const double long_s = [] { const char *e = getenv("DPCG_GRAPH_LONG_US"); return (e ? atof(e) : 100.0) * 1e-6; }();
        bool graph_now = use_graph && (max_iter - enq) >= chunk && (many_launches || !(t_iter > 25e-6) || (graph_long && t_iter > long_s));
        // DPCG_DRIVER_ALTERNATE (tests): mix single updates and replayed chunks -- 1, 8, 1, 1, 8, ... -- which is what a
        // per-update time hovering around the 25 us threshold does to the choice above
        if (alternate) graph_now = graph_now && (calls++ % 3) != 0;
        // Deferred-x form: the captured chunk reads p_j from h->p for even j, so a replay must start at an even update;
        // after an odd number of single updates one more single update restores the parity.
        if (graph_now && defer_x && (enq & 1)) graph_now = false;
        if (graph_now) {
            DPCG_HIP(hipGraphLaunch(h->graph_exec, s));
            enq += chunk;
        } else {
            DPCG_TRY(enqueue_iteration(h, flags, x_true, s, enq));
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
        defer_x = defer_x_eligible(h, flags, x_true);
        {
            const char *e = getenv("DPCG_DRIVER_ALTERNATE");   // read per solve: tests switch it on and off
            alternate = e && e[0] == '1';
        }
        if ((fused || defer_x) && !h->p2) {
            DPCG_TRY(dev_alloc(&h->p2, n));
            drop_graph(h);
        }
        const int per_update = 3 + precond_launches(h);
        many_launches = per_update >= 16;
        if (many_launches) chunk = std::max(1, std::min(chunk, 1024 / per_update));   // keep the graph at ~1K nodes
        if (defer_x && (chunk & 1)) chunk += 1;   // a replayed chunk must start at an even update (p buffer parity)
        use_graph = !(flags & DPCG_NO_GRAPH) && !x_true && max_iter >= chunk && h->precond != DPCG_PRECOND_CALLBACK;
        if (use_graph) {
            DPCG_TRY(ensure_graph(h, flags, chunk));
            if (!h->graph_exec) use_graph = false;       // a voided capture (see ensure_graph)
        }
        *ex.prog_host = 0;
        if (h->perm) {                     // b, x0, x_true arrive in the caller's numbering
            if (!h->pb) DPCG_TRY(dev_alloc(&h->pb, n));
            launch_gather_f64(n, h->perm, b, h->pb, s);
            b = h->pb;
            if (x_true) {
                if (!h->pxt) DPCG_TRY(dev_alloc(&h->pxt, n));
                launch_gather_f64(n, h->perm, x_true, h->pxt, s);
                x_true = h->pxt;
            }
        }
        if (x0) {
            if (h->perm) launch_gather_f64(n, h->perm, x0, h->x, s);
            else DPCG_HIP(hipMemcpyAsync(h->x, x0, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
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
                             h->hist_cap, ex.prog_dev, s, kMaxSpmvGrid);
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
        else if (defer_x)
            launch_final_deferred(n, h->scal, h->x, h->p, h->p2, h->vec_grid, s);
        else
            launch_final_check(h->scal, s);
        DPCG_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(Scalars), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        const auto t1 = std::chrono::steady_clock::now();                            // cg.py:88 (the loop only)
        DPCG_TRY(check_spin_errors(h, s));
        const Scalars sc = *h->scal_host;
        if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
        if (iters) *iters = sc.k;                                                    // cg.py:90
        if (final_res) *final_res = sc.res;
        bool pending = false;
        if (res_history) {
            DPCG_HIP(hipMemcpyAsync(res_history, h->hist, (size_t)(sc.k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
            pending = true;
        }
        if (err_history && x_true) {
            DPCG_HIP(hipMemcpyAsync(err_history, h->err_hist, (size_t)(sc.k + 1) * sizeof(double),
                                    hipMemcpyDeviceToHost, s));
            pending = true;
        }
        // x is handed over in stream order: the copy is enqueued on the caller's stream, a second host sync is only
        // needed for the host-side history buffers
        if (x) {
            if (h->perm) launch_scatter_f64(n, h->perm, h->x, x, s);       // back to the caller's numbering
            else DPCG_HIP(hipMemcpyAsync(x, h->x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        if (pending) DPCG_HIP(hipStreamSynchronize(s));
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
    if (h->A.n > kSmallMaxN || h->perm) return false;
    return h->precond == DPCG_PRECOND_NONE || h->precond == DPCG_PRECOND_JACOBI || h->precond == DPCG_PRECOND_CSR ||
           h->precond == DPCG_PRECOND_LLT_MULTIPLY;
}

void free_ell(SmallEll &e) {
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

// ------------------------------------------------------------------------------------------------
// mid-size systems: the whole solve in one launch, a team of 32 workgroups per system (dpcg_team.hip)
// ------------------------------------------------------------------------------------------------
// Rows beyond which ONE system (M = I / Jacobi) takes the team kernel instead of the one-workgroup kernel: that one holds up to 6 rows
// per thread -- 2.7 us per update at 2.4K rows, 3.0 at 4 096 (4 rows per thread), but 6.7-7.5 at 4.9K-6.1K (5-6 rows: the matrix no
// longer stays in registers) where a team takes 5.0 (tools/small_vs_team_probe.py).  Batches of such systems keep the one-workgroup
// kernel (256 of them in one launch).  DPCG_TEAM_MIN_ROWS: development knob.
static int team_min_rows() {
    static const int v = [] { const char *e = getenv("DPCG_TEAM_MIN_ROWS"); return e ? atoi(e) : 4096; }();
    return v;
}
static bool team_eligible(const dpcg_system *h, int flags, const double *x_true) {
    static const bool enabled = [] {
        const char *e = getenv("DPCG_TEAM");
        if (e && e[0] == '0') return false;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        return cus >= 256;                       // eight teams of 32 workgroups, one workgroup per CU, all resident
    }();
    // (DPCG_VAL32_IF_LOSSLESS is a permission about how the matrix is STREAMED; the one-launch forms keep it on chip in fp64 and
    // return the same bits)
    if (!enabled || x_true || (flags & (DPCG_SPMV_F32 | DPCG_NO_TEAM | DPCG_NO_FUSE))) return false;
    if (h->A.n <= team_min_rows() || h->A.n > team_max_rows() || h->perm) return false;
    if (h->planA.max_row_len < 1 || h->planA.max_row_len > team_max_row_len()) return false;   // rows live in registers
    if (h->planA.kernel == SPMV_VECTOR) return false;     // long rows: the multi-launch path's row-sharing kernel
    return h->precond == DPCG_PRECOND_NONE || h->precond == DPCG_PRECOND_JACOBI;
}

// One mid-size system, default flags: the team kernel (with the team on one XCD -- the usual placement -- 5.0-6.3 us per update up to
// 32K rows, 7.3-9.3 up to 64K, against 9.6-14.6 for the launches: profiles/r04_team_trace.txt).  DPCG_NO_SMALL ("no whole-solve
// kernel") keeps it on the launches, as it does for the one-workgroup kernel; DPCG_TEAM_SINGLE=0: development knob.
static bool single_team_default(const dpcg_system *h, int flags) {
    static const bool on = [] { const char *e = getenv("DPCG_TEAM_SINGLE"); return !(e && e[0] == '0'); }();
    (void)h;
    return on && !(flags & (DPCG_NO_SMALL | DPCG_NO_GRAPH));
}

extern "C" int dpcg_get_reduction_geometry(dpcg_handle_t h, int32_t out[16]) {
    if (!h || !out) return invalid("dpcg_get_reduction_geometry: NULL argument");
    out[0] = h->planA.grid;
    out[1] = h->planA.nrb;
    out[2] = h->planA.kernel == SPMV_TILE ? h->planA.cyclic : 0;       // 0 slabs, 1 cyclic, 2 cyclic with XCD runs inside a pass
    out[3] = h->vec_grid;
    out[4] = fuse_eligible(h, 0, nullptr) ? 1 : 0;
    out[5] = h->planA.kernel | (h->planA.kernel == SPMV_VECTOR ? h->planA.tpr << 8 : 0);      // (+ lanes per row of the CSR-vector kernel)
    // threads of the one-workgroup solve a default call takes (0: not that form)
    out[6] = small_eligible(h, 0, nullptr) ? (small_variant((int)h->A.n, h->planA.max_row_len, h->precond) % 16 != 0 ? 768 : 1024) : 0;
    out[7] = team_eligible(h, 0, nullptr) ? (single_team_default(h, 0) ? 2 : 1) : 0;      // 2: a single default solve takes that form too
    // who sums <r,z> in a multi-launch update (cg.py:82): 0 k_update_r (M = I, Jacobi), 1 k_dot_partials, 2 k_lm_finish (way out of a
    // level-major solve), 3 the SpMV that applied M (its plan in out[9..11]), 4 the colour sweeps (out[12..15]), 9 a tree the checker
    // does not restate
    int rzk = 0;
    const SpmvPlan *pm = nullptr;
    switch (h->precond) {
        case DPCG_PRECOND_NONE: case DPCG_PRECOND_JACOBI: rzk = 0; break;
        case DPCG_PRECOND_CSR: pm = &h->planM; break;
        case DPCG_PRECOND_LLT_MULTIPLY: pm = &h->planL; break;
        case DPCG_PRECOND_LLT_SOLVE:
            rzk = h->lvlU.sweep ? 9 : ((h->lvlU.level_major && h->lvlU.strips.n_strips == 0) ? 2 : 1);
            break;
        default: rzk = 1; break;
    }
    out[9] = out[10] = out[11] = 0;
    if (pm) {
        rzk = 3;
        out[9] = pm->grid;
        out[10] = pm->nrb;
        out[11] = pm->kernel == SPMV_TILE ? pm->cyclic : 0;
        // lanes per row where a CSR-vector kernel applies M (bits 8-15: M or L; 16-23: L^T of an L L^T product)
        if (pm->kernel == SPMV_VECTOR) out[11] |= pm->tpr << 8;
        if (h->precond == DPCG_PRECOND_LLT_MULTIPLY && h->planLt.kernel == SPMV_VECTOR) out[11] |= h->planLt.tpr << 16;
    }
    out[12] = out[13] = out[14] = out[15] = 0;
    if (rzk == 9 && h->precond == DPCG_PRECOND_LLT_SOLVE && h->lvlU.sweep && h->lvlU.n_levels <= 16) {
        // colour sweeps: <r,z> is summed launch by launch over the levels of the upper solve -- the first of them by the lower solve's
        // last launch when that one opens the upper solve (apply_precond) -- every workgroup adding its share to its slot.
        // [12] launches, [13] workgroups, [14] two bits per launch, first launch lowest: how its workgroups walk the level's row blocks
        // (0 slabs by virtual block, 1 blocks b, b + G, ..., 2 the same by virtual block), [15] 1 = the first launch is the lower solve's
        const Levels &lo = h->lvlL, &up = h->lvlU;
        const bool paired = lo.level_major && up.level_major && up.lm_from_lower && lo.sweep && lo.lm_to_upper;
        auto mode = [](const Levels &lv, int l) {
            const bool tiled = l < (int)lv.sw_max_chunks.size() && lv.sw_max_chunks[(size_t)l] > 0;
            return tiled ? (lv.sweep_cyclic ? 2 : 0) : 1;
        };
        int packed = 0;
        for (int l = 0; l < up.n_levels; ++l) packed |= ((l == 0 && paired) ? mode(lo, lo.n_levels - 1) : mode(up, l)) << (2 * l);
        rzk = 4;
        out[12] = up.n_levels;
        out[13] = up.sweep_grid;
        out[14] = packed;
        out[15] = paired ? 1 : 0;
    }
    out[8] = rzk;
    return DPCG_OK;
}

static std::mutex &team_launch_mutex() {
    static std::mutex *m = new std::mutex();
    return *m;
}

// Back-off of the one-launch forms (one workgroup team / whole chip): they assume that all of their workgroups become co-resident, and
// a launch that cannot (RCCL kernels or another process holding CUs, a long kernel on another stream) spins for the full 20 ms bound
// before the call goes on through the launches.  Three such timeouts in a row and the one-launch forms are skipped for a cool-down
// (2 s, doubling up to 32 s while the re-probes keep failing); a launch that completes clears it.  One warning per process.
struct CoResidency {
    std::atomic<int> misses{0};
    std::atomic<long long> closed_until_ns{0};
    std::atomic<int> cooldown_s{2};
    std::atomic<bool> warned{false};
    static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    bool open() const { return now_ns() >= closed_until_ns.load(std::memory_order_relaxed); }
    void launched_fine() {
        misses.store(0, std::memory_order_relaxed);
        cooldown_s.store(2, std::memory_order_relaxed);
    }
    void timed_out() {
        if (misses.fetch_add(1, std::memory_order_relaxed) + 1 < 3) return;
        const int cd = cooldown_s.load(std::memory_order_relaxed);
        closed_until_ns.store(now_ns() + (long long)cd * 1000000000ll, std::memory_order_relaxed);
        cooldown_s.store(std::min(2 * cd, 32), std::memory_order_relaxed);
        misses.store(2, std::memory_order_relaxed);            // (the re-probe after the cool-down closes it again at its first timeout)
        if (!warned.exchange(true))
            fprintf(stderr, "[dpcg] the one-launch solve kernels could not become co-resident three times in a row (somebody else holds CUs): "
                            "solving through the multi-launch path, re-probing every %d s and up\n", cd);
    }
};
static CoResidency &co_residency() {
    static CoResidency *c = new CoResidency();
    return *c;
}

static int ensure_team(dpcg_system *h, hipStream_t s) {
    DPCG_TRY(build_ell(h->A, h->ell_a, s));
    if (h->ell_a.W > team_max_row_len()) return invalid("team solve: a row has more than 7 entries");
    if (!h->p2) {
        DPCG_TRY(dev_alloc(&h->p2, h->A.n));
        drop_graph(h);
    }
    if (!h->team_part) DPCG_TRY(dev_alloc(&h->team_part, 4 * 2 * 32 + 8 + 16));      // (+ 8 words of phase times, DPCG_TEAM_TRACE; + 32 ints: the XCDs)
    if (!h->team_sync) DPCG_TRY(dev_alloc(&h->team_sync, 2));
    return DPCG_OK;
}

static TeamDesc make_team_desc(dpcg_system *h, const double *b, const double *x0, double *x, double rtol_sq, double atol_sq,
                               int max_iter, int flags) {
    TeamDesc d;
    memset(&d, 0, sizeof(d));
    d.n = (int)h->A.n;
    d.precond = h->precond;
    d.max_iter = max_iter;
    d.init_check_r = (flags & DPCG_INIT_CHECK_R) ? 1 : 0;
    d.hist_cap = h->hist_cap;
    d.W = h->ell_a.W;
    d.rp = h->A.rowptr;
    d.dinv = h->dinv;
    d.ell_col = h->ell_a.col;
    d.ell_val = h->ell_a.val;
    d.b = b; d.x0 = x0; d.x = x ? x : h->x; d.hist = h->hist;
    d.z = h->z; d.p0 = h->p; d.p1 = h->p2;
    d.rtol_sq = rtol_sq; d.atol_sq = atol_sq;
    d.out = h->scal;
    d.bar = h->team_sync;
    d.part = h->team_part;
    d.err = reinterpret_cast<int *>(h->team_sync + 1);
    d.xcc = reinterpret_cast<int *>(h->team_part + 4 * 2 * 32 + 8);
    static const bool trace = [] { const char *e = getenv("DPCG_TEAM_TRACE"); return e && e[0] == '1'; }();
    d.dbg = trace ? reinterpret_cast<unsigned long long *>(h->team_part + 4 * 2 * 32) : nullptr;
    return d;
}

static int team_slabs_per_wg(int64_t n) {
    const int64_t slabs = (n + 1023) / 1024;
    return (int)((slabs + 31) / 32);
}

static int solve_team_one(dpcg_system *h, const double *b, const double *x0, double *x, double rtol_sq, double atol_sq,
                          int max_iter, int flags, hipStream_t s, int *iters, double *final_res, double *seconds,
                          double *res_history) {
    DPCG_TRY(ensure_work(h, max_iter, false, false));
    DPCG_TRY(ensure_team(h, s));
    if (!h->team_desc) {
        TeamDesc *td = nullptr;
        DPCG_TRY(dev_alloc(&td, 1));
        h->team_desc = td;
    }
    const TeamDesc d = make_team_desc(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags);
    DPCG_HIP(hipMemcpyAsync(h->team_desc, &d, sizeof(d), hipMemcpyHostToDevice, s));
    DPCG_HIP(hipMemsetAsync(h->team_sync, 0, 2 * sizeof(unsigned int), s));
    launch_fill_pending(h->team_part, 4 * 2 * 32, s);                                // every reduction slot: "not written yet"
    DPCG_HIP(hipStreamSynchronize(s));
    // one team launch at a time per process: the workgroups of a team wait for each other, and two such launches dispatched at
    // once from two host threads could each hold part of the chip waiting for the rest of it (the 20 ms bound would end that)
    std::lock_guard<std::mutex> one_team_launch(team_launch_mutex());
    const auto t0 = std::chrono::steady_clock::now();                                // cg.py:69 (the launch is the loop)
    DPCG_TRY(launch_pcg_team(static_cast<const TeamDesc *>(h->team_desc), 1, team_slabs_per_wg(h->A.n), h->ell_a.W, s, d.dbg != nullptr));
    DPCG_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();                                // cg.py:88
    DPCG_CHECK_LAUNCH();
    const Scalars sc = *h->scal_host;
    if (sc.status < 0) {
        co_residency().timed_out();
        set_error("team solve: a workgroup waited (20 ms) for a team member that never arrived");
        return sc.status;
    }
    co_residency().launched_fine();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (iters) *iters = sc.k;
    if (final_res) *final_res = sc.res;
    if (d.dbg && sc.k > 0) {
        unsigned long long w[8];
        DPCG_HIP(hipMemcpy(w, d.dbg, sizeof(w), hipMemcpyDeviceToHost));
        const double us = 0.01 / sc.k;       // 100 MHz ticks -> us per update
        fprintf(stderr, "[dpcg team] %d updates, us per update on rank 0: SpMV %.2f, sum <p,Ap> %.2f (block sums %.2f, hand-off %.2f), vector update + publish %.2f, "
                "sum <r,z> %.2f (drain %.2f, block sums %.2f, hand-off %.2f), total %.2f\n", sc.k, w[0] * us, w[1] * us, w[5] * us, (w[1] - w[5]) * us, w[2] * us,
                w[3] * us, w[6] * us, w[7] * us, (w[3] - w[6] - w[7]) * us, w[4] * us);
    }
    if (res_history) {
        DPCG_HIP(hipMemcpyAsync(res_history, h->hist, (size_t)(sc.k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
    }
    return sc.status;
}

// ------------------------------------------------------------------------------------------------
// cache-sized systems: the whole solve in one launch, the whole chip as one team (dpcg_chip.hip)
// ------------------------------------------------------------------------------------------------
// 65 537 .. 1 048 576 rows, M = I / Jacobi; rows of <= 7 entries (9 up to 524 288 rows), half-bandwidth < 32 768 (stencils; meshes after the
// library's RCM): matrix and vectors resident -- otherwise, rows of up to 24 entries: the vectors resident, the matrix streamed;
// matrix and vectors stay in registers and LDS for the whole solve.  DPCG_CHIP=0 / DPCG_CHIP_MIN_ROWS: development knobs.
// the resident form: every row in the slots of its thread, every column within the 16-bit reach of its row
static bool chip_resident_shape(const dpcg_system *h, bool f32_slots = false) {
    return h->planA.max_row_len >= 1 && h->planA.max_row_len <= chip_max_row_len(h->A.n, f32_slots) && h->planA.max_band >= 0 &&
           h->planA.max_band <= chip_max_band();
}
static bool chip_eligible(const dpcg_system *h, int flags, const double *x_true) {
    static const bool enabled = [] {
        const char *e = getenv("DPCG_CHIP");
        if (e && e[0] == '0') return false;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        return cus >= chip_workgroups();         // one workgroup per CU, all resident
    }();
    static const int min_rows = [] { const char *e = getenv("DPCG_CHIP_MIN_ROWS"); return e ? atoi(e) : team_max_rows(); }();
    // (DPCG_SPMV_F32: with x0 = 0, see the callers; DPCG_VAL32_IF_LOSSLESS: a permission about how the matrix is streamed -- resident in
    // fp64 the results are the same bits, so the flag does not keep a system off the chip)
    if (!enabled || x_true || (flags & (DPCG_NO_TEAM | DPCG_NO_FUSE))) return false;
    if (h->A.n <= min_rows || h->A.n > chip_max_rows()) return false;
    if (h->planA.max_row_len < 1 || h->planA.max_band < 0) return false;
    if (h->precond != DPCG_PRECOND_NONE && h->precond != DPCG_PRECOND_JACOBI) return false;
    if (chip_resident_shape(h)) return true;
    // rows too long or columns too far for the resident form: the same kernel with the matrix streamed (dpcg_chip.hip MODE 5) -- fp64,
    // rows of up to 24 entries.  Measured against the launches, us per update (profiles/r05_chip_stream_probe.txt): 1M-row quadtree meshes
    // (rows of up to 9) 21.0-22.1 / 28.8-28.9, Delaunay graphs (rows of up to 21) of 1M rows 25.5 / 31.3, 500K 14.3 / 21.7, 250K 8.9 / 17.9,
    // 100K 6.8 / 12.6.  DPCG_CHIP_STREAM=0: never (development)
    const char *e = getenv("DPCG_CHIP_STREAM");
    if (e && e[0] == '0') return false;
    return h->planA.max_row_len <= chip_stream_max_row_len();
}
// a plain call takes it (DPCG_NO_SMALL = "no whole-solve kernel for one system" keeps the launches, as for the other two)
static bool chip_default(const dpcg_system *h, int flags) {
    (void)h;
    return !(flags & (DPCG_NO_SMALL | DPCG_NO_GRAPH));
}
static int chip_rows_per_wg(int64_t n) { return (int)((n + chip_workgroups() - 1) / chip_workgroups()); }

static int solve_chip_one(dpcg_system *h, const double *b, const double *x0, double *x, double rtol_sq, double atol_sq,
                          int max_iter, int flags, hipStream_t s, int *iters, double *final_res, double *seconds,
                          double *res_history) {
    const int64_t n = h->A.n;
    DPCG_TRY(ensure_work(h, max_iter, false, false));
    const int kSlots = chip_slot_doubles();                           // reduction slots (doubles), then 8 trace words, the flag, 256 XCD ids
    if (!h->chip_part) DPCG_TRY(dev_alloc(&h->chip_part, kSlots + 8 * 256 + 2 + 128));      // slots | trace words | flag | XCD ids
    if (!h->chip_zp) DPCG_TRY(dev_alloc(&h->chip_zp, chip_zp_doubles(n)));
    if (h->perm) {                         // b and x0 arrive in the caller's numbering
        if (!h->pb) DPCG_TRY(dev_alloc(&h->pb, n));
        launch_gather_f64(n, h->perm, b, h->pb, s);
        b = h->pb;
        if (x0) {
            launch_gather_f64(n, h->perm, x0, h->t, s);
            x0 = h->t;
        }
    }
    ChipDesc d;
    memset(&d, 0, sizeof(d));
    d.n = (int)n;
    d.precond = h->precond;
    d.max_iter = max_iter;
    d.init_check_r = (flags & DPCG_INIT_CHECK_R) ? 1 : 0;
    d.hist_cap = h->hist_cap;
    d.per = chip_rows_per_wg(n);
    d.rp = h->A.rowptr; d.ci = h->A.col; d.val = h->A.val; d.dinv = h->dinv;
    d.b = b; d.x0 = x0;
    d.x = (x && !h->perm) ? x : h->x;
    d.hist = h->hist;
    d.zp = h->chip_zp;
    d.rtol_sq = rtol_sq; d.atol_sq = atol_sq;
    d.out = h->scal;
    d.part = h->chip_part;
    d.err = reinterpret_cast<int *>(h->chip_part + kSlots + 8 * 256);
    d.band = h->planA.max_band;
    d.f32 = (flags & DPCG_SPMV_F32) ? 1 : 0;
    d.rp_nnz = (int)std::min<int64_t>(h->A.nnz, 0x1fffffff);
    // (config 5 with x0 = 0 stores the values as fp32: twice the slots -- rows of 9 entries resident at any size)
    d.stream_cap = (chip_resident_shape(h, d.f32 != 0 && !x0) || h->A.nnz > 0x1fffffff) ? 0 : h->planA.max_row_len * 64;     // (products of the 64 rows of a wave)
    { const char *e = getenv("DPCG_CHIP_BENCH"); d.bench = e ? atoi(e) : 0; }
    static const bool trace = [] { const char *e = getenv("DPCG_CHIP_TRACE"); return e && e[0] == '1'; }();
    if (d.f32 && (d.bench || trace || x0)) return DPCG_ERR_STATE;                      // (the caller goes on with the launches)
    if (d.stream_cap > 0 && (d.bench || trace)) return DPCG_ERR_STATE;
    static const bool plain_ok = [] { const char *e = getenv("DPCG_CHIP_LOCAL"); return !(e && e[0] == '0'); }();   // development: 0 = everything written through
    d.xcc = plain_ok ? reinterpret_cast<int *>(h->chip_part + kSlots + 8 * 256 + 2) : nullptr;
    d.dbg = trace ? reinterpret_cast<unsigned long long *>(h->chip_part + kSlots) : nullptr;
    const int st0 = launch_pcg_chip(d, h->planA.max_row_len, s, true);                // refused up front when it cannot be resident
    if (st0 != DPCG_OK) return st0;
    launch_fill_pending(h->chip_part, kSlots, s);                                     // every reduction slot: "not written yet"
    DPCG_HIP(hipMemsetAsync(d.err, 0, 2 * sizeof(int), s));
    DPCG_HIP(hipStreamSynchronize(s));
    // one whole-chip launch at a time per process (see solve_team_one)
    std::lock_guard<std::mutex> one_team_launch(team_launch_mutex());
    // DPCG_CHIP_EVENTS=1 (read per solve; bench.py's roofline leg): HIP events on the launch stream around the kernel
    static hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const char *ev_env = getenv("DPCG_CHIP_EVENTS");
    const bool events = ev_env && ev_env[0] == '1';
    if (events && !ev0) {
        DPCG_HIP(hipEventCreate(&ev0));
        DPCG_HIP(hipEventCreate(&ev1));
    }
    const auto t0 = std::chrono::steady_clock::now();                                // cg.py:69 (the launch is the loop)
    if (events) DPCG_HIP(hipEventRecord(ev0, s));
    DPCG_TRY(launch_pcg_chip(d, h->planA.max_row_len, s));
    if (events) DPCG_HIP(hipEventRecord(ev1, s));
    DPCG_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();                                // cg.py:88
    if (events) {
        float ms = 0.0f;
        DPCG_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        h->chip_trace_x[5] = (double)ms;
    }
    DPCG_CHECK_LAUNCH();
    const Scalars sc = *h->scal_host;
    if (sc.status < 0) {
        co_residency().timed_out();
        set_error("chip solve: a workgroup waited (20 ms) for one that never became resident");
        return sc.status;
    }
    co_residency().launched_fine();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (iters) *iters = sc.k;
    if (final_res) *final_res = sc.res;
    if (d.dbg) {
        std::vector<unsigned long long> w(8 * 256);
        DPCG_HIP(hipMemcpy(w.data(), d.dbg, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const double us = sc.k > 0 ? 0.01 / sc.k : 0.0;       // 100 MHz ticks -> us per update
        for (int i = 0; i < 7; ++i) h->chip_trace_us[i] = w[i] * us;                 // workgroup 0
        h->chip_trace_us[7] = (double)sc.k;
        // over the 256 workgroups: the slowest and the mean SpMV phase, the slowest publish phase
        double sp_max = 0, sp_sum = 0, pb_max = 0, pb_sum = 0;
        for (int g = 0; g < 256; ++g) {
            sp_max = std::max(sp_max, w[8 * g] * us); sp_sum += w[8 * g] * us;
            pb_max = std::max(pb_max, w[8 * g + 2] * us); pb_sum += w[8 * g + 2] * us;
        }
        h->chip_trace_x[0] = sp_max; h->chip_trace_x[1] = sp_sum / 256; h->chip_trace_x[2] = pb_max; h->chip_trace_x[3] = pb_sum / 256;
        h->chip_trace_x[4] = (double)(w[7] & 1ull);
        static const bool print = [] { const char *e = getenv("DPCG_CHIP_TRACE_PRINT"); return e && e[0] == '1'; }();
        if (print && sc.k > 0) {
            fprintf(stderr, "[dpcg chip] %d updates, groups on one XCD each: %d; us per update on workgroup 0: SpMV %.2f, sum <p,Ap> %.2f (of it waiting for the slots %.2f), "
                    "vector update + publish %.2f, sum <r,z> %.2f (waiting %.2f), total %.2f; SpMV phase over the workgroups: mean %.2f max %.2f; publish: mean %.2f max %.2f\n",
                    sc.k, (int)(w[7] & 1ull), w[0] * us, w[1] * us, w[5] * us, w[2] * us, w[3] * us, w[6] * us, w[4] * us, sp_sum / 256, sp_max, pb_sum / 256, pb_max);
            if (getenv("DPCG_CHIP_TRACE_ALL")) {
                for (int g = 0; g < 256; ++g) fprintf(stderr, "%s%.1f/%.1f", g % 32 ? " " : "\n   ", w[8 * g] * us, w[8 * g + 2] * us);
                fprintf(stderr, "\n");
            }
        }
    }
    bool pending = false;
    if (res_history) {
        DPCG_HIP(hipMemcpyAsync(res_history, h->hist, (size_t)(sc.k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
        pending = true;
    }
    if (x && h->perm) launch_scatter_f64(n, h->perm, h->x, x, s);                    // back to the caller's numbering
    if (pending) DPCG_HIP(hipStreamSynchronize(s));
    return sc.status;
}

// M = L L^T multiplied (the learned technique, IC multiplied) beyond the one-workgroup kernel: the whole chip, L and L^T resident
static const CsrDev &llt_l(const dpcg_system *h) { return h->perm ? h->Lp : h->L; }
static const CsrDev &llt_t(const dpcg_system *h) { return h->perm ? h->Ltp : h->Lt; }
static bool chip_llt_tagged() {       // DPCG_CHIP_LLT_SYNC=0 (development): plain vectors and a chip-wide barrier per product
    static const bool on = [] { const char *e = getenv("DPCG_CHIP_LLT_SYNC"); return !(e && e[0] == '0'); }();
    return on;
}
static bool chip_llt_eligible(const dpcg_system *h, int flags, const double *x_true) {
    static const bool enabled = [] {
        const char *e = getenv("DPCG_CHIP");
        if (e && e[0] == '0') return false;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        return cus >= chip_workgroups();
    }();
    if (!enabled || x_true || h->precond != DPCG_PRECOND_LLT_MULTIPLY) return false;
    if (flags & (DPCG_SPMV_F32 | DPCG_NO_TEAM | DPCG_NO_FUSE)) return false;
    if (h->A.n <= kSmallMaxN || h->A.n > chip_llt_max_rows()) return false;
    if (h->planA.max_row_len < 1 || h->planA.max_row_len > 7) return false;        // (dpcg_chip_llt.hip: rows of A of <= 7 entries)
    const int ml = std::max(h->planL.max_row_len, h->planLt.max_row_len);
    if (h->planL.max_row_len < 1 || h->planLt.max_row_len < 1 || ml > chip_llt_max_row_len()) return false;
    if (ml > 8 && h->A.n > chip_llt_max_rows() / 2) return false;      // (two rows a thread of 16-entry factor rows: beyond the registers)
    const int band = std::max(h->planA.max_band, std::max(h->planL.max_band, h->planLt.max_band));
    if (h->planA.max_band < 0 || h->planL.max_band < 0 || h->planLt.max_band < 0 || band > chip_max_band()) return false;
    return true;
}

static int solve_chip_llt_one(dpcg_system *h, const double *b, const double *x0, double *x, double rtol_sq, double atol_sq,
                              int max_iter, int flags, hipStream_t s, int *iters, double *final_res, double *seconds,
                              double *res_history) {
    const int64_t n = h->A.n;
    DPCG_TRY(ensure_work(h, max_iter, false, false));
    const int kSlots = chip_slot_doubles();
    if (!h->chip_part) DPCG_TRY(dev_alloc(&h->chip_part, kSlots + 8 * 256 + 2 + 128));
    if (!h->chip_zp) DPCG_TRY(dev_alloc(&h->chip_zp, chip_zp_doubles(n)));
    if (!h->chip_rt) DPCG_TRY(dev_alloc(&h->chip_rt, 2 * chip_zp_doubles(n)));      // two vectors of 2 x (n + pad) granules
    if (h->perm) {
        if (!h->pb) DPCG_TRY(dev_alloc(&h->pb, n));
        launch_gather_f64(n, h->perm, b, h->pb, s);
        b = h->pb;
        if (x0) {
            launch_gather_f64(n, h->perm, x0, h->t, s);
            x0 = h->t;
        }
    }
    const CsrDev &Lm = llt_l(h), &Tm = llt_t(h);
    ChipLltDesc d;
    memset(&d, 0, sizeof(d));
    d.n = (int)n;
    d.max_iter = max_iter;
    d.init_check_r = (flags & DPCG_INIT_CHECK_R) ? 1 : 0;
    d.hist_cap = h->hist_cap;
    d.per = chip_rows_per_wg(n);
    d.band = std::max(h->planA.max_band, std::max(h->planL.max_band, h->planLt.max_band));
    d.rp = h->A.rowptr; d.ci = h->A.col; d.val = h->A.val;
    d.lrp = Lm.rowptr; d.lci = Lm.col; d.lval = Lm.val;
    d.trp = Tm.rowptr; d.tci = Tm.col; d.tval = Tm.val;
    d.b = b; d.x0 = x0;
    d.x = (x && !h->perm) ? x : h->x;
    d.hist = h->hist;
    d.zp = h->chip_zp;
    d.rpub = h->chip_rt;
    d.tpub = h->chip_rt + chip_zp_doubles(n);
    d.rtol_sq = rtol_sq; d.atol_sq = atol_sq;
    d.out = h->scal;
    d.part = h->chip_part;
    d.err = reinterpret_cast<int *>(h->chip_part + kSlots + 8 * 256);
    static const bool plain_ok = [] { const char *e = getenv("DPCG_CHIP_LOCAL"); return !(e && e[0] == '0'); }();
    d.xcc = plain_ok ? reinterpret_cast<int *>(h->chip_part + kSlots + 8 * 256 + 2) : nullptr;
    // r and t = L^T r reach the neighbours as self-validating granules keyed by a per-launch nonce (DPCG_CHIP_LLT_SYNC=0, development:
    // plain vectors and a chip-wide barrier per product)
    const bool tagged = chip_llt_tagged();
    static std::atomic<unsigned> launch_nonce{0};
    unsigned nonce = 0;
    if (tagged)
        do { nonce = ++launch_nonce; } while (nonce == 0);
    d.nonce = nonce;
    const int max_l = std::max(h->planL.max_row_len, h->planLt.max_row_len);
    const int st0 = launch_pcg_chip_llt(d, h->planA.max_row_len, max_l, s, true);
    if (st0 != DPCG_OK) return st0;
    launch_fill_pending(h->chip_part, kSlots, s);
    DPCG_HIP(hipMemsetAsync(d.err, 0, 2 * sizeof(int), s));
    DPCG_HIP(hipStreamSynchronize(s));
    std::lock_guard<std::mutex> one_team_launch(team_launch_mutex());
    const auto t0 = std::chrono::steady_clock::now();                                // cg.py:69 (the launch is the loop)
    DPCG_TRY(launch_pcg_chip_llt(d, h->planA.max_row_len, max_l, s));
    DPCG_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();                                // cg.py:88
    DPCG_CHECK_LAUNCH();
    const Scalars sc = *h->scal_host;
    if (sc.status < 0) {
        co_residency().timed_out();
        set_error("chip solve (M = L L^T): a workgroup waited (20 ms) for one that never became resident");
        return sc.status;
    }
    co_residency().launched_fine();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (iters) *iters = sc.k;
    if (final_res) *final_res = sc.res;
    bool pending = false;
    if (res_history) {
        DPCG_HIP(hipMemcpyAsync(res_history, h->hist, (size_t)(sc.k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
        pending = true;
    }
    if (x && h->perm) launch_scatter_f64(n, h->perm, h->x, x, s);
    if (pending) DPCG_HIP(hipStreamSynchronize(s));
    return sc.status;
}

// M = (L L^T)^-1 by two triangular solves on the whole chip (dpcg_chip_trsv.hip): A resident, L and L^T streamed as per-wave block lists
// (built here once per preconditioner: dependency levels of both triangles by the sync-free analysis, then the lists in the chip kernel's
// geometry), y and z handed from level to level as self-validating granules.
static bool chip_trsv_enabled() {
    static const bool on = [] {
        const char *e = getenv("DPCG_CHIP");
        if (e && e[0] == '0') return false;
        e = getenv("DPCG_CHIP_TRSV");
        if (e && e[0] == '0') return false;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        return cus >= chip_workgroups();
    }();
    return on;
}
static int chip_trsv_min_rows() {
    // (from 1 024 rows: measured with IC(0) in multicolour order, us per update, launches -> this kernel: 1 674 rows (quadtree mesh, 4 colours)
    // 29.4 -> 14.9, 4 268 rows 32.6 -> 16.5, 10 000 rows 25.5 -> 9.1, 13 824 rows 32.3 -> 11.7 -- an update of the launches is launch-bound there)
    static const int v = [] { const char *e = getenv("DPCG_CHIP_TRSV_MIN_ROWS"); return e ? atoi(e) : 1024; }();
    return v;
}
// what can be told without the lists (they are built at the first solve)
static bool chip_trsv_shape(const dpcg_system *h, int flags, const double *x_true) {
    if (!chip_trsv_enabled() || x_true || h->precond != DPCG_PRECOND_LLT_SOLVE || h->trsv_state < 0) return false;
    if (flags & (DPCG_SPMV_F32 | DPCG_NO_TEAM | DPCG_NO_FUSE)) return false;
    if (h->A.n < chip_trsv_min_rows() || h->A.n > chip_max_rows()) return false;
    if (!chip_resident_shape(h)) return false;                       // (A resident: rows of <= 7 entries, 9 up to 524 288 rows; 16-bit column offsets)
    if (h->L.nnz <= 0 || h->Lt.nnz <= 0) return false;
    return true;
}
static void free_chip_trsv(dpcg_system *h) {
    free_chip_trsv_lists(h->trsv_l);
    free_chip_trsv_lists(h->trsv_u);
    dev_free(h->trsv_lv0);
    dev_free(h->trsv_diag0);
    dev_free(h->trsv_fval);
    dev_free(h->trsv_fcol);
    dev_free(h->trsv_fmeta);
    h->trsv_rpt = h->trsv_wmax = h->trsv_band = 0;
}
static int ensure_chip_trsv(dpcg_system *h, hipStream_t s) {
    if (h->trsv_state != 0) return DPCG_OK;
    const int64_t n = h->A.n;
    PhaseTimer pt(s);
    int32_t *lvl[2] = {nullptr, nullptr}, *ctl = nullptr;
    auto done = [&](int state, int code) {
        dev_free(lvl[0]); dev_free(lvl[1]); dev_free(ctl);
        if (state < 0) free_chip_trsv(h);
        h->trsv_state = state;
        return code;
    };
    int st;
    if ((st = dev_alloc(&lvl[0], n)) < 0 || (st = dev_alloc(&lvl[1], n)) < 0) return done(0, st);
    if ((st = dev_alloc(&ctl, 4)) < 0) return done(0, st);
    // factor index <-> handle index: the factor's own numbering (multicolour IC(0)) or the caller's (a reordered handle)
    const int32_t *handle_of_f = h->fmap ? h->fmap : h->iperm;
    const int32_t *f_of_handle = h->fmap ? h->fmap_inv : h->perm;
    const int per = chip_rows_per_wg(n);
    int nlev[2] = {0, 0};
    for (int upper = 0; upper < 2; ++upper) {                          // dependency levels of both triangles (sync-free analysis, dpcg_analysis.hip)
        const CsrDev &F = upper ? h->Lt : h->L;
        DPCG_HIP(hipMemsetAsync(lvl[upper], 0xff, (size_t)n * sizeof(int32_t), s));
        DPCG_HIP(hipMemsetAsync(ctl, 0, 4 * sizeof(int32_t), s));
        launch_levels_syncfree(n, F.rowptr, F.col, upper != 0, lvl[upper], reinterpret_cast<unsigned int *>(ctl), ctl + 1, s);
        if ((st = reduce_max_i32(lvl[upper], ctl + 2, n, s)) < 0) return done(0, st);
        int32_t h_ctl[4] = {0, 0, 0, 0};
        DPCG_HIP(hipMemcpyAsync(h_ctl, ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (h_ctl[1] || h_ctl[2] < 0) return done(-1, DPCG_OK);
        nlev[upper] = h_ctl[2] + 1;
        // Every level is a hand-off from one workgroup to another -- publish, become visible, be gathered: ~1.2 us inside an XCD, ~3 across
        // -- and a chain of them is all a many-level solve is: measured at 216 K / 512 K rows with IC(0) in a scattered caller's order, 17 /
        // 18 levels: 66 / 94 us per update here against 72 / 87 for the launches (whose sync-free kernels wait in the same way).  Few levels
        // (multicolour orders: 2-9) are where this form wins (2-4 x); beyond 16 (18 up to two rows a thread) the launches keep the solve (natural orders of grids: hundreds).
        // (the three development knobs of this routine are read per plan, not per process: a plan is built once per preconditioner)
        // (up to two rows a thread -- 262 144 rows -- a level is cheaper here, the level's next block being gathered ahead: 18)
        const int level_default = chip_rows_per_wg(n) <= 2 * chip_threads() ? 18 : 16;
        const int level_limit = [&] { const char *e = getenv("DPCG_CHIP_TRSV_MAX_LEVELS"); return e ? std::min(atoi(e), chip_trsv_max_levels()) : level_default; }();
        if (nlev[upper] > level_limit) return done(-1, DPCG_OK);
    }
    h->trsv_l.n_levels = nlev[0];
    h->trsv_u.n_levels = nlev[1];
    pt.mark("chip trsv: levels");
    // <= 4 rows a thread: the factor resident beside the matrix
    const bool resident_on = [] { const char *e = getenv("DPCG_CHIP_TRSV_RESIDENT"); return !(e && e[0] == '0'); }();
    const int rpt = chip_trsv_resident_rpt(per), wmax = rpt ? chip_trsv_resident_wmax(h->planA.max_row_len, rpt) : 0;
    if (resident_on && rpt && wmax) {
        int misfit = 0, band = 0;
        if ((st = build_chip_trsv_resident((int)n, per, rpt, wmax, h->L, h->Lt, lvl[0], lvl[1], f_of_handle, handle_of_f, &h->trsv_fval, &h->trsv_fcol,
                                           &h->trsv_fmeta, &misfit, &band, &h->trsv_tstride, s)) < 0)
            return done(-1, st);
        pt.mark("chip trsv: resident plan");
        if (!misfit && std::max(band, h->planA.max_band) <= chip_max_band()) {
            h->trsv_rpt = rpt;
            h->trsv_wmax = wmax;
            h->trsv_band = band;
            return done(1, DPCG_OK);
        }
        dev_free(h->trsv_fval); dev_free(h->trsv_fcol); dev_free(h->trsv_fmeta);       // (a factor with fill: the streamed form may still take it)
    }
    // Beyond 524 288 rows (8 rows a thread) the factor cannot sit beside the matrix and would be STREAMED (the block lists below).  Measured at
    // 1M rows that form loses to the launches (IC(0) in multicolour order: 66 against 58 us per update -- every dependent step of a block is a
    // memory-side round trip of ~1.2 us, and the kernel spills): it is kept for factors the resident form refuses at <= 4 rows a thread
    // (fill: a row's L and L^T parts beyond its slots) and, beyond, behind DPCG_CHIP_TRSV_STREAM=1 (development).
    const bool stream_big = [] { const char *e = getenv("DPCG_CHIP_TRSV_STREAM"); return e && e[0] == '1'; }();
    if (!rpt && !stream_big) return done(-1, DPCG_OK);
    if (!h->trsv_lv0 && (st = dev_alloc(&h->trsv_lv0, (int64_t)chip_workgroups() * chip_threads())) < 0) return done(0, st);
    if (!h->trsv_diag0 && (st = dev_alloc(&h->trsv_diag0, chip_trsv_diag_doubles())) < 0) return done(0, st);
    DPCG_HIP(hipMemsetAsync(h->trsv_lv0, 0, (size_t)chip_workgroups() * chip_threads() * sizeof(int32_t), s));
    for (int upper = 0; upper < 2; ++upper) {
        const CsrDev &F = upper ? h->Lt : h->L;
        ChipTrsvLists &out = upper ? h->trsv_u : h->trsv_l;
        if ((st = build_chip_trsv_lists((int)n, per, nlev[upper], F, lvl[upper], f_of_handle, handle_of_f, upper != 0, out, h->trsv_lv0, h->trsv_diag0, s)) < 0) return done(-1, st);
        out.n_levels = nlev[upper];
        if (out.max_row > chip_trsv_max_factor_row() || out.max_row > std::max(h->planA.max_row_len, 5) - 1 ||
            std::max(out.band, h->planA.max_band) > chip_max_band())
            return done(-1, DPCG_OK);
        pt.mark(upper ? "chip lists (L^T)" : "chip lists (L)");
    }
    return done(1, DPCG_OK);
}

static int solve_chip_trsv_one(dpcg_system *h, const double *b, const double *x0, double *x, double rtol_sq, double atol_sq,
                               int max_iter, int flags, hipStream_t s, int *iters, double *final_res, double *seconds,
                               double *res_history) {
    const int64_t n = h->A.n;
    DPCG_TRY(ensure_chip_trsv(h, s));
    if (h->trsv_state != 1) return DPCG_ERR_STATE;                 // this factor keeps the launches
    DPCG_TRY(ensure_work(h, max_iter, false, false));
    const int kSlots = chip_slot_doubles();
    if (!h->chip_part) DPCG_TRY(dev_alloc(&h->chip_part, kSlots + 8 * 256 + 2 + 128));
    if (!h->chip_zp) DPCG_TRY(dev_alloc(&h->chip_zp, chip_zp_doubles(n)));
    if (!h->chip_rt) DPCG_TRY(dev_alloc(&h->chip_rt, 2 * chip_zp_doubles(n)));
    if (h->perm) {
        if (!h->pb) DPCG_TRY(dev_alloc(&h->pb, n));
        launch_gather_f64(n, h->perm, b, h->pb, s);
        b = h->pb;
        if (x0) {
            launch_gather_f64(n, h->perm, x0, h->t, s);
            x0 = h->t;
        }
    }
    ChipTrsvDesc d;
    memset(&d, 0, sizeof(d));
    d.n = (int)n;
    d.max_iter = max_iter;
    d.init_check_r = (flags & DPCG_INIT_CHECK_R) ? 1 : 0;
    d.hist_cap = h->hist_cap;
    d.per = chip_rows_per_wg(n);
    d.band = std::max(h->planA.max_band, h->trsv_rpt ? h->trsv_band : std::max(h->trsv_l.band, h->trsv_u.band));
    d.rp = h->A.rowptr; d.ci = h->A.col; d.val = h->A.val;
    d.b = b; d.x0 = x0;
    d.x = (x && !h->perm) ? x : h->x;
    d.hist = h->hist;
    d.zp = h->chip_zp;
    d.ypub = h->chip_rt;
    d.zpub = h->chip_rt + chip_zp_doubles(n);
    d.first_l = h->trsv_l.first_blk; d.first_u = h->trsv_u.first_blk;
    d.blk_l = h->trsv_l.blk; d.blk_u = h->trsv_u.blk;
    d.val_l = h->trsv_l.val; d.val_u = h->trsv_u.val;
    d.col_l = h->trsv_l.col; d.col_u = h->trsv_u.col;
    d.nent_l = h->trsv_l.n_ent; d.nent_u = h->trsv_u.n_ent;
    d.lv0 = h->trsv_lv0; d.diag0 = h->trsv_diag0;
    d.fval = h->trsv_fval; d.fcol = h->trsv_fcol; d.fmeta = h->trsv_fmeta;
    d.nlev_l = h->trsv_l.n_levels; d.nlev_u = h->trsv_u.n_levels;
    d.tstride = h->trsv_tstride;
    const bool resident = h->trsv_rpt != 0;
    auto launch = [&](bool check_only) {
        return resident ? launch_pcg_chip_trsv_resident(d, h->trsv_rpt, h->trsv_wmax, s, check_only)
                        : launch_pcg_chip_trsv(d, h->planA.max_row_len, std::max(h->trsv_l.max_row, h->trsv_u.max_row), s, check_only);
    };
    d.rtol_sq = rtol_sq; d.atol_sq = atol_sq;
    d.out = h->scal;
    d.part = h->chip_part;
    d.err = reinterpret_cast<int *>(h->chip_part + kSlots + 8 * 256);
    static const bool plain_ok = [] { const char *e = getenv("DPCG_CHIP_LOCAL"); return !(e && e[0] == '0'); }();
    d.xcc = plain_ok ? reinterpret_cast<int *>(h->chip_part + kSlots + 8 * 256 + 2) : nullptr;
    static std::atomic<unsigned> launch_nonce{0x40000000u};
    unsigned nonce = 0;
    do { nonce = ++launch_nonce; } while (nonce == 0);
    d.nonce = nonce;
    static const bool trace = [] { const char *e = getenv("DPCG_CHIP_TRACE"); return e && e[0] == '1'; }();
    unsigned long long *dbg = nullptr;
    if (trace) {
        DPCG_TRY(dev_alloc(&dbg, 256 * 64));
        d.dbg = dbg;
    }
    const int st0 = launch(true);
    if (st0 != DPCG_OK) return st0;
    launch_fill_pending(h->chip_part, kSlots, s);
    DPCG_HIP(hipMemsetAsync(d.err, 0, 2 * sizeof(int), s));
    DPCG_HIP(hipStreamSynchronize(s));
    std::lock_guard<std::mutex> one_team_launch(team_launch_mutex());
    static hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const char *ev_env = getenv("DPCG_CHIP_EVENTS");
    const bool events = ev_env && ev_env[0] == '1';
    if (events && !ev0) {
        DPCG_HIP(hipEventCreate(&ev0));
        DPCG_HIP(hipEventCreate(&ev1));
    }
    const auto t0 = std::chrono::steady_clock::now();                                // cg.py:69 (the launch is the loop)
    if (events) DPCG_HIP(hipEventRecord(ev0, s));
    DPCG_TRY(launch(false));
    if (events) DPCG_HIP(hipEventRecord(ev1, s));
    DPCG_HIP(hipMemcpyAsync(h->scal_host, h->scal, sizeof(Scalars), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();                                // cg.py:88
    if (events) {
        float ms = 0.0f;
        DPCG_HIP(hipEventElapsedTime(&ms, ev0, ev1));
        h->chip_trace_x[5] = (double)ms;
    }
    DPCG_CHECK_LAUNCH();
    const Scalars sc = *h->scal_host;
    if (sc.status < 0) {
        co_residency().timed_out();
        set_error("chip solve (triangular solves): a workgroup waited (20 ms) for one that never became resident");
        return sc.status;
    }
    co_residency().launched_fine();
    if (dbg) {                                                 // us per update by phase of the apply: mean and slowest wave of the chip
        std::vector<unsigned long long> w(256 * 64);
        DPCG_HIP(hipMemcpy(w.data(), dbg, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        dev_free(dbg);
        const double us = sc.k > 0 ? 0.01 / (sc.k + 1) : 0.0;   // (the applies: one per update and the first)
        static const char *names[8] = {"L sweep+prologue", "L blocks", "L polling again", "L^T sweep+prologue", "L^T blocks", "L^T polling again", "wait behind the apply", "blocks that polled again (count per apply)"};
        fprintf(stderr, "[dpcg chip trsv] %d updates; us per apply by phase, mean over the 2048 waves / slowest wave:\n", sc.k);
        for (int ph = 0; ph < 8; ++ph) {
            double sum = 0, mx = 0;
            for (int wv = 0; wv < 2048; ++wv) {
                const double val = (double)w[(size_t)wv * 8 + ph] * (ph == 7 ? (sc.k > 0 ? 1.0 / (sc.k + 1) : 0.0) : us);
                sum += val;
                mx = std::max(mx, val);
            }
            fprintf(stderr, "    %-44s %8.2f / %8.2f\n", names[ph], sum / 2048, mx);
        }
    }
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (iters) *iters = sc.k;
    if (final_res) *final_res = sc.res;
    bool pending = false;
    if (res_history) {
        DPCG_HIP(hipMemcpyAsync(res_history, h->hist, (size_t)(sc.k + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
        pending = true;
    }
    if (x && h->perm) launch_scatter_f64(n, h->perm, h->x, x, s);
    if (pending) DPCG_HIP(hipStreamSynchronize(s));
    return sc.status;
}

extern "C" int dpcg_debug_occupy(int workgroups, double milliseconds, dpcg_stream_t stream) {
    if (workgroups < 1 || workgroups > 4096 || !(milliseconds > 0.0) || milliseconds > 2000.0) return invalid("dpcg_debug_occupy: bad arguments");
    DPCG_TRY(launch_occupy(workgroups, milliseconds, (hipStream_t)stream));
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_debug_l2_gather(int granules_per_group, int reps, const int32_t offsets[7], int depth, int written_through, dpcg_stream_t stream,
                                    double *gbs, double *us_per_pass, int *groups_local) {
    if (!offsets || granules_per_group < 32 * chip_threads() || granules_per_group % 32 != 0 || granules_per_group > (1 << 22) || reps < 1 || reps > 100000 ||
        (depth != 2 && depth != 4) || written_through < 0 || written_through > 2)
        return invalid("dpcg_debug_l2_gather: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int kSlots = chip_slot_doubles();
    double *table = nullptr, *part = nullptr;
    int *ints = nullptr;                   // 7 offsets | err (2) | xcc (257)
    unsigned long long *ticks = nullptr;
    DPCG_TRY(dev_alloc(&table, (size_t)8 * granules_per_group * 2));
    DPCG_TRY(dev_alloc(&part, kSlots));
    DPCG_TRY(dev_alloc(&ints, 8 + 2 + 260));
    DPCG_TRY(dev_alloc(&ticks, 256 + 1));
    int st = DPCG_OK;
    std::vector<unsigned long long> w(256);
    int flags[2] = {0, 0};
    hipError_t e = hipMemsetAsync(ints, 0, (8 + 2 + 260) * sizeof(int), s);
    if (e == hipSuccess) e = hipMemcpyAsync(ints, offsets, 7 * sizeof(int), hipMemcpyHostToDevice, s);
    launch_fill_pending(part, kSlots, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> one_team_launch(team_launch_mutex());
        st = launch_l2_gather_probe(table, granules_per_group, reps, ints, depth, written_through, part, ints + 8, ints + 10, ticks,
                                    reinterpret_cast<unsigned *>(ticks + 256), s);
        if (st == DPCG_OK) e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess && st == DPCG_OK) e = hipMemcpy(w.data(), ticks, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess && st == DPCG_OK) e = hipMemcpy(flags, ints + 8, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && st == DPCG_OK) e = hipMemcpy(flags + 1, ints + 10 + 256, sizeof(int), hipMemcpyDeviceToHost);
    dev_free(table); dev_free(part); dev_free(ints); dev_free(ticks);
    DPCG_HIP(e);
    if (st != DPCG_OK) return st;
    unsigned long long worst = 0;
    for (unsigned long long x : w) worst = std::max(worst, x);
    if (flags[0] || worst == 0) {
        set_error("dpcg_debug_l2_gather: the workgroups never became co-resident");
        return DPCG_ERR_STATE;
    }
    const double us = (double)worst * 0.01;                   // 100 MHz
    const double bytes = (double)reps * 256.0 * 512.0 * 8.0 * 7.0 * 16.0;
    if (gbs) *gbs = bytes / (us * 1.0e-6) / 1.0e9;
    if (us_per_pass) *us_per_pass = us / reps;
    if (groups_local) *groups_local = flags[1];
    return DPCG_OK;
}

extern "C" int dpcg_get_chip_info(dpcg_handle_t h, int32_t out[8], double trace_us[8]) {
    if (!h || !out) return invalid("dpcg_get_chip_info: NULL argument");
    if (chip_trsv_shape(h, 0, nullptr) && h->trsv_state == 0) (void)ensure_chip_trsv(h, nullptr);     // (whether the factor fits is known once its lists exist)
    const bool el = chip_eligible(h, 0, nullptr) || chip_llt_eligible(h, 0, nullptr) || (chip_trsv_shape(h, 0, nullptr) && h->trsv_state == 1);
    out[0] = el ? (chip_default(h, 0) ? 2 : 1) : 0;          // 2: a plain dpcg_solve takes the chip kernel
    out[1] = chip_workgroups();
    out[2] = chip_threads();
    out[3] = chip_rows_per_wg(h->A.n);
    out[4] = h->planA.max_row_len;
    out[5] = h->planA.max_band;
    if (trace_us)
        for (int i = 0; i < 8; ++i) trace_us[i] = h->chip_trace_us[i];
    // bit 0: the last traced chip solve kept plainly stored copies (every group on one XCD); bits 8-15: lanes that share a row in the
    // form a plain solve takes now (2: M = L L^T multiplied with 16-entry factor rows on <= 256 rows a workgroup; a row's terms of the
    // dot products then sit in the even lanes)
    const bool split = chip_llt_eligible(h, 0, nullptr) && std::max(h->planL.max_row_len, h->planLt.max_row_len) > 8 &&
                       chip_rows_per_wg(h->A.n) <= chip_threads() / 2 && chip_llt_tagged();
    out[6] = (int)h->chip_trace_x[4] | ((split ? 2 : 1) << 8);
    out[7] = (int)(h->chip_trace_x[5] * 1.0e6);             // DPCG_CHIP_EVENTS=1: the last chip kernel between HIP events on its stream, ns
    return DPCG_OK;
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
    const bool one_launch_open = co_residency().open();      // (false during the cool-down after repeated co-residency timeouts)
    const bool team_first = one_launch_open && team_eligible(h, flags, x_true) && ((flags & DPCG_TEAM) || single_team_default(h, flags));
    if (small_eligible(h, flags, x_true) && !team_first)
        return solve_small_one(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags, (hipStream_t)stream, iters, final_res,
                               seconds, res_history);
    // a single team by default (tools/team_crossover_probe.py)
    if (team_first) {
        const int st = solve_team_one(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags, (hipStream_t)stream, iters, final_res,
                                      seconds, res_history);
        if (st != DPCG_ERR_STATE) return st;
        // the team never became co-resident (a plain launch assumes it): the multi-launch path below needs no such thing
    }
    // cache-sized systems on the whole chip
    if (one_launch_open && chip_eligible(h, flags, x_true) && ((flags & DPCG_TEAM) || chip_default(h, flags))) {
        const int st = solve_chip_one(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags, (hipStream_t)stream, iters, final_res,
                                      seconds, res_history);
        if (st != DPCG_ERR_STATE) return st;
        // the workgroups never became co-resident, or the kernel was refused up front: the multi-launch path needs no such thing
    }
    if (one_launch_open && chip_trsv_shape(h, flags, x_true) && ((flags & DPCG_TEAM) || chip_default(h, flags))) {
        const int st = solve_chip_trsv_one(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags, (hipStream_t)stream, iters, final_res,
                                           seconds, res_history);
        if (st != DPCG_ERR_STATE) return st;
        // (the factor does not fit the form -- too many levels, rows too long -- or the workgroups never became co-resident: the launches)
    }
    if (one_launch_open && chip_llt_eligible(h, flags, x_true) && ((flags & DPCG_TEAM) || chip_default(h, flags))) {
        const int st = solve_chip_llt_one(h, b, x0, x, rtol_sq, atol_sq, max_iter, flags, (hipStream_t)stream, iters, final_res,
                                          seconds, res_history);
        if (st != DPCG_ERR_STATE) return st;
    }
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
    {   // a handle owns ONE set of work vectors: the same handle twice would share them across streams
        std::vector<dpcg_handle_t> sorted(handles, handles + count);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
            return invalid("dpcg_solve_batch: the handles must be distinct (create one handle per system in flight)");
    }
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
        // descriptor / result arrays live across calls (hipMalloc + hipFree cost ~0.1 ms each: as much as the launch)
        struct BatchScratch {
            SmallDesc *descs = nullptr;
            Scalars *out = nullptr;
            int cap = 0;
            ~BatchScratch() {          // (allocated through the block cache: freed through it)
                dev_free(descs);
                dev_free(out);
            }
        };
        static thread_local BatchScratch scratch;
        if (scratch.cap < count) {
            dev_free(scratch.descs);
            dev_free(scratch.out);
            scratch.cap = 0;
            DPCG_TRY(dev_alloc(&scratch.descs, count));
            DPCG_TRY(dev_alloc(&scratch.out, count));
            scratch.cap = count;
        }
        SmallDesc *d_descs = scratch.descs;
        Scalars *d_out = scratch.out;        // one contiguous result array: a single copy back for the whole batch
        std::vector<Scalars> out((size_t)count);
        for (int i = 0; i < count; ++i) descs[i].out = d_out + i;
        hipError_t e = hipMemcpy(d_descs, descs.data(), descs.size() * sizeof(SmallDesc), hipMemcpyHostToDevice);
        const auto t0 = std::chrono::steady_clock::now();
        int st = e == hipSuccess ? launch_pcg_small(d_descs, count, lds, kinds, variants, nullptr) : DPCG_ERR_HIP;
        if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, out.size() * sizeof(Scalars), hipMemcpyDeviceToHost);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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
    bool all_chip = true;
    for (int i = 0; i < count; ++i) all_chip = all_chip && chip_eligible(handles[i], flags, nullptr);
    if (all_chip && co_residency().open() && ((flags & DPCG_TEAM) || chip_default(handles[0], flags))) {
        // cache-sized systems: the whole chip serves one system at a time (7-13 us per update against 30 for the launches -- nothing to
        // interleave), and a batch member's result is bit for bit that of its single solve
        int worst_chip = DPCG_OK;
        bool refused = false;
        for (int i = 0; i < count && !refused; ++i) {
            int it = 0;
            double fr = 0.0, sec = 0.0;
            const int st = solve_chip_one(handles[i], b[i], x0 ? x0[i] : nullptr, x ? x[i] : nullptr, rtol_sq, atol_sq, max_iter, flags,
                                          nullptr, &it, &fr, &sec, nullptr);
            if (st == DPCG_ERR_STATE) { refused = true; break; }     // never co-resident: the whole batch takes the streams below
            if (st < 0) return st;
            if (iters) iters[i] = it;
            if (final_res) final_res[i] = fr;
            if (seconds) seconds[i] = sec;
            if (status) status[i] = st;
            worst_chip = std::max(worst_chip, st);
        }
        if (!refused) {
            DPCG_HIP(hipStreamSynchronize(nullptr));
            return worst_chip;
        }
    }
    bool all_team = true;
    for (int i = 0; i < count; ++i) all_team = all_team && team_eligible(handles[i], flags, nullptr);
    if (all_team && co_residency().open() && ((flags & DPCG_TEAM) || single_team_default(handles[0], flags))) {      // (any count: one team already beats the launches)
        // up to eight systems per launch, one team (normally: one XCD) each; the launches follow one another
        struct TeamScratch {
            TeamDesc *descs = nullptr;
            ~TeamScratch() { dev_free(descs); }
        };
        static thread_local TeamScratch scratch;
        if (!scratch.descs) DPCG_TRY(dev_alloc(&scratch.descs, 8));
        int worst_team = DPCG_OK;
        bool team_timed_out = false;       // a team waited in vain for its members (the chip shared with another process, a long kernel
                                           // holding CUs: the launch is a plain one, co-residency is assumed, not guaranteed)
        for (int g0 = 0; g0 < count && !team_timed_out; g0 += 8) {
            const int ng = std::min(8, count - g0);
            TeamDesc descs[8];
            int slabs = 1, wmax = 1;
            for (int i = 0; i < ng; ++i) {
                dpcg_system *hi = handles[g0 + i];
                DPCG_TRY(ensure_work(hi, max_iter, false, false));
                DPCG_TRY(ensure_team(hi, nullptr));
                descs[i] = make_team_desc(hi, b[g0 + i], x0 ? x0[g0 + i] : nullptr, x ? x[g0 + i] : nullptr, rtol_sq, atol_sq,
                                          max_iter, flags);
                DPCG_HIP(hipMemsetAsync(hi->team_sync, 0, 2 * sizeof(unsigned int), nullptr));
                launch_fill_pending(hi->team_part, 4 * 2 * 32, nullptr);
                slabs = std::max(slabs, team_slabs_per_wg(hi->A.n));
                wmax = std::max(wmax, hi->ell_a.W);
            }
            DPCG_HIP(hipMemcpy(scratch.descs, descs, (size_t)ng * sizeof(TeamDesc), hipMemcpyHostToDevice));
            std::lock_guard<std::mutex> one_team_launch(team_launch_mutex());        // (see solve_team_one)
            const auto t0 = std::chrono::steady_clock::now();
            DPCG_TRY(launch_pcg_team(scratch.descs, ng, slabs, wmax, nullptr));
            DPCG_HIP(hipStreamSynchronize(nullptr));
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            DPCG_CHECK_LAUNCH();
            for (int i = 0; i < ng; ++i) {
                dpcg_system *hi = handles[g0 + i];
                DPCG_HIP(hipMemcpy(hi->scal_host, hi->scal, sizeof(Scalars), hipMemcpyDeviceToHost));
                const Scalars sc = *hi->scal_host;
                if (sc.status < 0) {
                    team_timed_out = true;
                    co_residency().timed_out();
                    break;
                }
                if (iters) iters[g0 + i] = sc.k;
                if (final_res) final_res[g0 + i] = sc.res;
                if (seconds) seconds[g0 + i] = sec;      // the group ran as one launch
                if (status) status[g0 + i] = sc.status;
                worst_team = std::max(worst_team, sc.status);
            }
        }
        if (!team_timed_out) return worst_team;
        // not an error: the whole batch runs again through the multi-launch path below, which needs no co-residency
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "[dpcg] team solve: a workgroup waited 20 ms for a team member that never became resident; "
                            "the batch is solved through the multi-launch path instead\n");
        }
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
