// Host-side internals shared by dpcg_api.hip, dpcg_precond.hip and dpcg_solve.hip (not part of the ABI).
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dpcg_internal.h"

using namespace dpcg;

#define DPCG_TRY(expr)               \
    do {                             \
        int _st = (expr);            \
        if (_st < 0) return _st;     \
    } while (0)

#define DPCG_CHECK_LAUNCH() DPCG_HIP(hipGetLastError())

// sets dpcg_last_error() and returns DPCG_ERR_INVALID
int invalid(const char *msg);

// DPCG_SETUP_TRACE=1: phase times of the setup routines on stderr (development)
struct PhaseTimer {
    bool on;
    hipStream_t s;
    std::chrono::steady_clock::time_point t;
    explicit PhaseTimer(hipStream_t stream) : s(stream) {
        static const bool enabled = [] { const char *e = getenv("DPCG_SETUP_TRACE"); return e && e[0] == '1'; }();
        on = enabled;
        if (on) {
            (void)hipStreamSynchronize(s);
            t = std::chrono::steady_clock::now();
        }
    }
    void mark(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[dpcg setup] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// ---- device memory ---------------------------------------------------------------------------------------
template <typename T>
inline int dev_alloc(T **p, int64_t count) {
    *p = nullptr;
    if (count <= 0) count = 1;
    hipError_t e = cached_alloc((void **)p, (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? DPCG_ERR_NOMEM : DPCG_ERR_HIP;
    }
    return DPCG_OK;
}

template <typename T>
inline void dev_free(T *&p) {
    if (p) cached_free(p);
    p = nullptr;
}

void free_csr(CsrDev &c);
// dpcg_icholt.hip: ILU++-style thresholded incomplete Cholesky of a symmetric CSR matrix into an owned lower-triangular CSR
int icholt_factor(const CsrDev &A, int add_fill_in, double threshold, CsrDev &Lf, hipStream_t s);
void free_parked(dpcg_system *h);                       // dpcg_api.hip: the multicolour IC(0) parked by dpcg_update_values
void free_levels(Levels &l);
int count_levels_on_demand(dpcg_system *h);   // dpcg_precond.hip
void free_plan(SpmvPlan &plan);
void free_ell(SmallEll &e);
int grid_for(int64_t n);
int upload_csr(CsrDev &out, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col, const void *val,
               int val_dtype, int memspace, int copy, hipStream_t s);
// Chooses the SpMV kernel of a matrix (gather / vector / x-tile) and builds the tile plan where it applies.
int make_plan(const CsrDev &A, SpmvPlan &plan, hipStream_t s, bool allow_tile = false);

// ---- handle ----------------------------------------------------------------------------------------------
struct HandleExtras {
    hipStream_t cap_stream = nullptr;
    unsigned long long *prog_host = nullptr;  // pinned + mapped: the solve's progress word
    unsigned long long *prog_dev = nullptr;   // device-side address of the same word
};
// kept outside dpcg_system so the struct in the header stays POD-like; the registry itself is guarded so that
// handles may be created/destroyed from several host threads (one handle is still used by one thread at a time)
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
ExtrasRegistry &extras();
void drop_graph(dpcg_system *h);
void free_precond(dpcg_system *h, bool keep_parked = false);   // (by default a parked factor goes too)
// work vectors, partial buffers, history (grown on demand)
int ensure_work(dpcg_system *h, int max_iter, bool need_f32, bool need_err);
// whether a solve with these flags runs two-kernel updates, and the extra operands of its SpMV kernel
bool fuse_eligible(const dpcg_system *h, int flags, const double *x_true);
FuseArgs fuse_args(dpcg_system *h);
// z = M r for the handle's preconditioner (cg.py:61,81); in_loop: kernels return at once when the solve is done
// part_rz / n_part_rz (may be null): where the apply may leave per-workgroup partials of <r,z> (cg.py:82) when its last kernel
// can sum them on the way; *n_part_rz = their count, or 0 when the caller has to launch the dot product itself
// lower_first_done: the first level of the lower solve has been computed already (colour sweeps: it rode on K2)
int apply_precond(dpcg_system *h, const double *r, double *z, hipStream_t s, bool in_loop = false, double *part_rz = nullptr,
                  int *n_part_rz = nullptr, bool lower_first_done = false);
// number of <r,z> partials an in-loop apply of the handle's preconditioner leaves (vec_grid when it leaves none)
int rz_partial_count(const dpcg_system *h);
int check_spin_errors(dpcg_system *h, hipStream_t s);
