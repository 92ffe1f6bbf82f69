/*
 * dpcg.h -- C ABI of the MI355X-native preconditioned-CG solve path (libdpcg.so).
 *
 * Drop-in boundary for the hot path of jsappl/DeepPreconditioning.
 * The reference has no native interface of its own: its "operator API" is Python duck typing
 * (`A @ v`, `M @ v`, `torch.inner`) inside uibk/deep_preconditioning/cg.py.  Each entry point
 * below names the reference lines whose work it replaces.  The Python mirror of the reference
 * signatures (deeppreconditioning_amd/cg.py, utils.py) binds these symbols with ctypes; the stub
 * a reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - Plain pointers and sizes only; no exceptions cross the ABI; every function returns a
 *     dpcg_status (negative = error, dpcg_last_error() gives the message for this thread).
 *   - Vectors are fp64, length n, DEVICE pointers unless a parameter says "host".
 *   - CSR: int32 rowptr[n+1], int32 col[nnz] (ascending inside a row), fp64 or fp32 val[nnz].
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue on it;
 *     only dpcg_solve*, dpcg_create (host inputs) and the setup routines synchronise it.
 *   - The caller owns every buffer it passes.  The handle owns its copies/analysis data (row-block
 *     maps, level sets, transposed factor, work vectors, graph).  A handle may be used by one
 *     thread at a time.
 */
#ifndef DPCG_H
#define DPCG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dpcg_system *dpcg_handle_t;
typedef void *dpcg_stream_t; /* hipStream_t */

enum dpcg_status {
    DPCG_OK = 0,            /* converged: res < rtol_sq (cg.py:71)                                  */
    DPCG_MAX_ITER = 1,      /* max_iter updates done without meeting the test (cg.py:70)            */
    DPCG_BREAKDOWN = 2,     /* NaN/Inf in the recurrence (<Ap,p> = 0 ...); reference spins silently */
    DPCG_ERR_INVALID = -1,  /* bad argument                                                         */
    DPCG_ERR_HIP = -2,      /* a HIP runtime call failed (no device, launch failure, ...)           */
    DPCG_ERR_NOMEM = -3,    /* allocation failed                                                    */
    DPCG_ERR_PIVOT = -4,    /* IC(0): non-positive pivot                                            */
    DPCG_ERR_STATE = -5     /* call order error (e.g. solve mode without a factor)                  */
};

enum dpcg_dtype { DPCG_F64 = 0, DPCG_F32 = 1 };
enum dpcg_memspace { DPCG_DEVICE = 0, DPCG_HOST = 1 };

/* How `zk = M @ rk` (cg.py:61,81) is applied. */
enum dpcg_precond {
    DPCG_PRECOND_NONE = 0,         /* M = I                         test.py:70-72  (vanilla)        */
    DPCG_PRECOND_JACOBI = 1,       /* M = diag(1/a_ii)              test.py:74-79                   */
    DPCG_PRECOND_CSR = 2,          /* z = M r, M an explicit CSR    test.py:88,105 (M = L L^T)      */
    DPCG_PRECOND_LLT_MULTIPLY = 3, /* z = L (L^T r), same operator as test.py:102-105, never formed */
    DPCG_PRECOND_LLT_SOLVE = 4,    /* z = L^-T (L^-1 r), level-scheduled SpTRSV (north_star)        */
    DPCG_PRECOND_CALLBACK = 5      /* z = M r by a caller-supplied function (the duck-typed `M @ rk`) */
};

/* dpcg_solve flags */
enum dpcg_solve_flags {
    DPCG_INIT_CHECK_R = 1,   /* first test on <r0,r0> (scipy cg, utils.py:66-72) instead of the
                                reference's <z0,z0> (cg.py:66)                                      */
    DPCG_SPMV_F32 = 2,       /* mixed precision: A@p with fp32 val and fp32 p, fp64 everywhere else */
    DPCG_NO_GRAPH = 4,       /* launch kernels one by one instead of replaying a hipGraph           */
    DPCG_NO_SMALL = 8,       /* no whole-solve kernel for ONE system: neither the one-workgroup kernel (<= 6144 rows) nor the team
                                kernel (4 097 .. 65 536 rows, M = I / Jacobi) -- the multi-launch path */
    DPCG_VAL32_IF_LOSSLESS = 16, /* stream the matrix values as fp32 when every value survives the round trip
                                fp64 -> fp32 -> fp64 unchanged (true for the reference's data, which is fp32
                                upcast to fp64: data_set.py:121, test.py:68): 8 instead of 12 bytes per non-zero,
                                products and sums still fp64, results bit-identical.  Ignored when lossy. */
    DPCG_NO_FUSE = 32,       /* run an update as three kernels (SpMV | r,z | x,p) instead of the default two, in
                                which the SpMV kernel also forms p = z + beta p (cg.py:83) and the deferred
                                x += alpha p (cg.py:79); same arithmetic, bit-identical results              */
    DPCG_NO_TEAM = 64,       /* do not use the one-launch whole-solve kernel for mid-size systems (6 145 .. 65 536 rows,
                                M = I or Jacobi: a team of 32 workgroups per system, up to eight systems per launch) */
    DPCG_TEAM = 128          /* use that kernel whatever the other flags say (it serves one system and batches by default:
                                5.0-9.3 us per update against 9.6-14.6 for the launches; eight teams together 4.4x their rate) */
};

/* ---- library ------------------------------------------------------------------------------- */
int dpcg_version(void);
const char *dpcg_status_string(int status);
const char *dpcg_last_error(void);
/* Device facts for the roofline report: CU count, HBM bytes, gcnArchName into name[name_len]. */
int dpcg_device_info(int *cu_count, int64_t *hbm_bytes, char *name, int name_len);
/* Device blocks freed by the setup routines are kept for the next setup (hipFree costs ~0.14 ms a block and a
 * preconditioner setup frees dozens; the cache is bounded by DPCG_CACHE_MB, default 1024, 0 = off).  This returns them
 * to the driver.  No reference counterpart: the reference's setups run in ilupp / scipy on the host (test.py:81-88). */
int dpcg_release_cached_memory(void);

/* ---- system handle: the operator A (test.py:61-68 / train.py:93-95 pass it dense; here CSR) -- */
/* copy = 0 with DEVICE pointers borrows the arrays (caller keeps them alive); otherwise copied. */
int dpcg_create(dpcg_handle_t *out, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                const void *val, int val_dtype, int memspace, int copy, dpcg_stream_t stream);
int dpcg_destroy(dpcg_handle_t h);
/* Introspection (any out pointer may be NULL).  spmv_kernel: 0 gather (CSR-stream), 1 CSR-vector, 2 x-tile; +16 when
 * a default solve runs two-kernel updates (see DPCG_NO_FUSE), +32 when the x-tile kernel reads its once-read streams and
 * writes y non-temporally (streams beyond the Infinity Cache), +64 when some 256-row blocks of the x-tile plan touch too many
 * places of x for an LDS tile and gather instead, +128 when the row blocks are dealt out to the workgroups cyclically instead of
 * in slabs.  precond_nnz: nnz of M (CSR) or of L. */
int dpcg_get_info(dpcg_handle_t h, int64_t *n, int64_t *nnz, int *spmv_kernel, int *precond_kind,
                  int64_t *precond_nnz, int *n_levels_lower, int *n_levels_upper);

/* ---- bandwidth-reducing reordering (BASELINE config 3: unstructured OpenFOAM numbering) ------------------------
 * The reference hands the solver whatever numbering the mesh generator produced (generate_data.py:67-74 reads the
 * OpenFOAM dump as is); a numbering that scatters neighbours makes every x[col] of `A @ p` (cg.py:75) its own cache
 * line.  dpcg_reorder computes a reverse Cuthill-McKee order ON THE DEVICE and lets the handle iterate on P A P^T.
 * Transparent to the caller: b, x0, x, x_true, dinv, M and L keep the CALLER's numbering in every call (vectors are
 * gathered / scattered on the device, matrices permuted at setup; an IC(0) / L factor is the factor of the caller's
 * matrix, its level schedule merely relabelled).  Call it before attaching a preconditioner (an attached one is
 * dropped).  mode: DPCG_REORDER_AUTO reorders only when the system has >= 65536 rows, its SpMV plan is the gather
 * kernel (no x-tile plan) AND either the measured x-gather traffic (distinct 128-byte lines per 256-row block) exceeds
 * 4x the bytes used (reverse Cuthill-McKee), or the traffic is fine on average but some row blocks are too scattered
 * for the x-tile plan (OpenFOAM appends refined cells): then the cheap REGION-BY-REGION numbering is tried -- many
 * breadth-first searches grown at once, numbered region by region and ring by ring -- and kept when the x-tile plan
 * takes the result.  DPCG_REORDER_ALWAYS: reverse Cuthill-McKee always; DPCG_REORDER_REGIONS: region by region always.
 * *applied (may be NULL) = 1 when the handle now iterates on a reordered matrix.
 * PCG is invariant under symmetric permutation up to the order of floating-point sums: iterates agree with the
 * unpermuted solve to rounding, and to 1e-10 with the CPU reference run on P A P^T (dpcg_get_permutation). */
enum dpcg_reorder_mode { DPCG_REORDER_NONE = 0, DPCG_REORDER_AUTO = 1, DPCG_REORDER_ALWAYS = 2, DPCG_REORDER_REGIONS = 3 };
/* New values on the SAME sparsity pattern: val[nnz] in the order of the arrays dpcg_create was given (the caller's
 * promise -- only values are passed).  The reference builds a fresh tensor per sample (data_set.py / test.py:61-68);
 * the pressure systems of one mesh share their pattern, and the SpMV plan and the reordering depend on nothing else,
 * so the next system costs an upload and a value permutation instead of a create (1M-DoF unstructured system: ~21 ms
 * with reordering).  The preconditioner is dropped (it was computed from the old values): attach one again -- an IC(0)
 * in multicolour order (dpcg_set_precond_ic0_ordered) keeps what its PATTERN determined meanwhile, so attaching it again
 * only computes values (1M DoF: 0.25 ms instead of 2-3.5 ms); any other preconditioner call frees that.  A handle
 * that borrows its arrays (copy = 0) borrows `val` likewise: fp64, device, 16-byte aligned; it may be the same buffer
 * rewritten in place. */
int dpcg_update_values(dpcg_handle_t h, const void *val, int val_dtype, int memspace, dpcg_stream_t stream);
int dpcg_reorder(dpcg_handle_t h, int mode, dpcg_stream_t stream, int *applied);
/* *reordered = 0/1; perm_host (may be NULL): int32[n], perm[new] = old (row `new` of the iterated matrix is the
 * caller's row `old`); gather_ratio (may be NULL): x-gather line traffic / bytes used of the caller's matrix, as
 * measured by the last dpcg_reorder call (0 if never measured). */
int dpcg_get_permutation(dpcg_handle_t h, int *reordered, int32_t *perm_host, double *gather_ratio);

/* ---- preconditioner M (test.py:70-105) ------------------------------------------------------- */
int dpcg_set_precond_none(dpcg_handle_t h);
/* dinv = NULL: 1/diag(A) is extracted on the device (test.py:76). */
int dpcg_set_precond_jacobi(dpcg_handle_t h, const double *dinv, int memspace, dpcg_stream_t stream);
int dpcg_set_precond_csr(dpcg_handle_t h, int64_t nnz, const int32_t *rowptr, const int32_t *col, const double *val,
                         int memspace, dpcg_stream_t stream);
/* L: lower-triangular CSR, columns ascending, diagonal stored LAST in each row.
 * mode = DPCG_PRECOND_LLT_MULTIPLY or DPCG_PRECOND_LLT_SOLVE.  Builds L^T and (solve) level sets. */
int dpcg_set_precond_llt(dpcg_handle_t h, int mode, int64_t nnz, const int32_t *rowptr, const int32_t *col,
                         const double *val, int memspace, dpcg_stream_t stream);
/* IC(0) of A (stands in for ilupp.ichol0, test.py:83), then as dpcg_set_precond_llt. */
int dpcg_set_precond_ic0(dpcg_handle_t h, int mode, dpcg_stream_t stream);
/* IC(0) in an ordering of the library's choice, applied by triangular solves.  DPCG_ORDER_CALLER is dpcg_set_precond_ic0
 * (the factor ilupp.ichol0 would return for the matrix as the caller numbered it, test.py:83).  DPCG_ORDER_MULTICOLOR factors
 * Q A Q^T with the unknowns listed colour by colour (two colours by breadth-first parity when the mesh graph is bipartite --
 * every 5- / 7-point grid -- otherwise a deterministic greedy colouring): the factor's dependency graph is then only as deep
 * as the number of colours, so the two triangular solves of an apply are a handful of wide, fully parallel sweeps instead of
 * hundreds of dependent levels.  It is a DIFFERENT preconditioner from the reference's (same algorithm, another elimination
 * order: iteration counts differ, typically +20 % against a natural ordering), offered because on this hardware it is the
 * form in which an incomplete-Cholesky apply beats Jacobi to the solution.  Caller-visible vectors keep the caller's numbering.
 * dpcg_get_factor then returns L in the factor's numbering; dpcg_get_precond_ordering gives that numbering:
 * perm_host[k] = the caller's row at factor position k (int32[n], host; identity for DPCG_ORDER_CALLER), *n_colors. */
enum dpcg_ordering { DPCG_ORDER_CALLER = 0, DPCG_ORDER_MULTICOLOR = 1 };
int dpcg_set_precond_ic0_ordered(dpcg_handle_t h, int mode, int ordering, dpcg_stream_t stream);
int dpcg_get_precond_ordering(dpcg_handle_t h, int *n_colors, int32_t *perm_host);
/* ICT with LEVEL-1 FILL: thresholded incomplete Cholesky on a static pattern, then as dpcg_set_precond_llt.  NOT what
 * ilupp.icholt computes (that is dpcg_set_precond_icholt below: a per-column entry count, not a fill level); kept because its
 * pattern is known before the values are, so it factors level-parallel on the device at any size.  The contract
 * (oracle/oracle.py::ict) is: pattern = tril(A) plus the fill created by eliminating with original entries only
 * (fill_in >= 1; 0 = no fill); row-wise numeric phase in the operation order of IC(0); an off-diagonal entry v = acc /
 * L_jj is dropped when |v| * L_jj < threshold * ||A(j:n, j)||_1 (the rule MATLAB documents for ichol 'ict').
 * fill_in = 0, threshold = 0 reproduces dpcg_set_precond_ic0 bit for bit. */
int dpcg_set_precond_ict(dpcg_handle_t h, int mode, int fill_in, double threshold, dpcg_stream_t stream);
/* ilupp.icholt(A, add_fill_in, threshold) as ILU++ defines it -- the DEFAULT incomplete-Cholesky technique of the reference's
 * harness (test.py:81-88: `ilupp.icholt(matrix, add_fill_in=1, threshold=0.1)`; ilupp 1.0.2, uv.lock:952, wraps ILU++), then as
 * dpcg_set_precond_llt.  The published algorithm (J. Mayer, ILU++, PAMM 7 (2007); Y. Saad, ILUT(p, tau), 1994, on the lower
 * triangle): column by column, w = A[k:, k] - sum_{j<k} L_kj L[k:, j]; d = sqrt(w_k); off-diagonal candidates below
 * threshold * ||w_offdiag||_2 are dropped, of the rest the nnz(A[k+1:, k]) + add_fill_in largest are kept (ties: smaller row).
 * The ilupp binary is absent from the build image: the restatement (oracle/oracle.py::icholt) is pinned to the published
 * description, not to ilupp's output; the device factor equals the restatement bit for bit.  Limits: at most 64 kept entries
 * per row / column of L and 256 candidates per column (DPCG_ERR_INVALID beyond); a non-positive pivot is DPCG_ERR_PIVOT.  The
 * columns are walked in order by one wave (the pattern of a column depends on the values before it): milliseconds at the
 * reference's 2.4K-22K rows, ~3 us per row beyond. */
int dpcg_set_precond_icholt(dpcg_handle_t h, int mode, int add_fill_in, double threshold, dpcg_stream_t stream);
/* The reference's operator protocol asks of M nothing but `M @ rk` (cg.py:61,81).  An M that is not a matrix this library
 * can hold (a Python object with __matmul__, a multigrid cycle, ...) is applied through a function the caller supplies:
 * fn(user, r, z, n, stream) must ENQUEUE z = M r on `stream` (device pointers, caller's numbering; it is called from the
 * thread inside dpcg_solve, once per update, and keeps being called for the few updates the driver has enqueued beyond
 * convergence).  Slow path by construction -- one host call per update, no graph replay, vectors gathered / scattered
 * around the call on a reordered handle -- while SpMV, dots and vector updates stay the HIP kernels. */
typedef void (*dpcg_precond_fn)(void *user, const double *r, double *z, int64_t n, dpcg_stream_t stream);
int dpcg_set_precond_callback(dpcg_handle_t h, dpcg_precond_fn fn, void *user);
/* The geometry of the handle's reductions, for a checker that wants to sum in the same order (oracle/pcg_oracle.c,
 * orc_set_dot_tree: with it the CPU restatement reproduces the multi-launch solve's residual history BIT FOR BIT for M = I /
 * Jacobi): out[0] = workgroups of the SpMV kernel of the PCG loop, out[1] = its 256-row blocks, out[2] = 1 when the blocks are
 * dealt out cyclically, 2 when cyclically with an XCD's blocks of a pass contiguous (0: contiguous slabs), out[3] = workgroups of the vector kernels, out[4] = 1 when a solve with default
 * flags runs two-kernel updates, out[5] = the SpMV kernel in bits 0-7 (0 gather, 1 vector, 2 x-tile) and, for the vector kernel, its
 * lanes per row in bits 8-15, out[6] = threads of the one-workgroup
 * solve when a default call takes that form (0 otherwise), out[7] = 1 when the system is eligible for the team solve (2: and a
 * single default solve takes it),
 * out[8] = who sums <r,z> behind the CURRENT preconditioner in a multi-launch update (0 the r-update kernel, 1 a separate dot
 * launch, 2 the way-out pass of a level-major triangular solve, 3 the SpMV that applied M -- its grid / row blocks / walk in
 * out[9..11]; out[11] bits 8-15: lanes per row when that SpMV is the vector kernel, bits 16-23: the same for the L^T product of
 * M = L L^T --, 4 the colour sweeps of a triangular solve: out[12] launches that add to <r,z> (the levels of the upper solve, first
 * to last), out[13] workgroups of each, out[14] two bits per launch, first launch lowest (how a workgroup walks the level's 256-row
 * blocks: 0 its slab by virtual block, 1 blocks b, b + G, ..., 2 the same by virtual block), out[15] = 1 when the first of them is the
 * lower solve's last launch --, 9 a tree the checker does not restate: more than 16 sweeps). */
int dpcg_get_reduction_geometry(dpcg_handle_t h, int32_t out[16]);
/* The whole-chip solve (dpcg_chip.hip: 65 537 .. 1 048 576 rows, M = I / Jacobi, cg.py:58-90 in ONE launch of 256 workgroups -- rows of
 * <= 7 entries (9 up to 524 288 rows) within 32 767 columns of the diagonal: matrix and vectors stay in registers and LDS for the whole
 * solve; otherwise, rows of <= 24 entries: the vectors stay, the matrix is streamed every update; the same rows per thread and the same
 * reduction trees either way).  out[0] = 1 when the
 * system with its CURRENT preconditioner is eligible (2: and a plain dpcg_solve takes that form), out[1] = workgroups (256),
 * out[2] = threads of each (512), out[3] = rows per workgroup (ceil(n / 256): workgroup v owns rows v * out[3] .., thread t of it
 * rows v * out[3] + t + 512 k -- what a checker needs to add the dot products in the kernel's order), out[4] = longest row,
 * out[5] = largest |col - row| (-1: not measured, systems beyond 1 048 576 rows), out[6]: bit 0 = the last traced chip solve found
 * every group of 32 workgroups on one XCD (and kept plainly stored copies in that XCD's L2), bits 8-15 = lanes that share a row (2: M = L L^T
 * multiplied with 16-entry factor rows; row v * out[3] + t / 2 then speaks through the even lane t of its pair in the dot products), out[7] = with DPCG_CHIP_EVENTS=1 in the
 * environment, the duration of the last chip kernel in nanoseconds, between HIP events on the stream it ran on.  trace_us (may be NULL): with DPCG_CHIP_TRACE=1 in
 * the environment, microseconds per update that workgroup 0 spent in the phases of the last chip solve -- [0] q = A p (the SpMV
 * phase), [1] sum <p,Ap> incl. its barrier, [2] vector updates + publishing, [3] sum <r,z>, <r,r> incl. its barrier, [4] the whole
 * loop, [5] / [6] of [1] / [3]: waiting for the other workgroups' slots, [7] = the number of updates.  No reference counterpart
 * (the reference's loop is host Python, cg.py:70-87). */
int dpcg_get_chip_info(dpcg_handle_t h, int32_t out[8], double trace_us[8]);
/* Test hook: enqueue on `stream` a kernel of `workgroups` workgroups that each take a whole CU (all of its LDS) and spin for
 * `milliseconds` -- what a long-running kernel of another stream or process does to the co-residency the one-launch solves (team,
 * chip) rely on.  They bound every wait (20 ms) and fall back to the multi-launch path; the tests hold them to that with this call.
 * No reference counterpart. */
int dpcg_debug_occupy(int workgroups, double milliseconds, dpcg_stream_t stream);
/* Measurement hook (bench.py `roofline.frac_of_measured_ceiling`): the rate at which the chip serves what the whole-chip solve kernel
 * (dpcg_chip.hip) gathers -- 16-byte granules stored plainly by the workgroups of the owner's XCD, read with agent-scope loads, 64
 * consecutive granules per wave instruction -- with nothing else going on: 256 workgroups x 512 threads, `granules_per_group` granules per
 * XCD (a multiple of 32, >= 16 384), `reps` passes of 8 x 7 gathers per thread at `offsets` (in granules, relative to the thread's rows,
 * wrapped inside the XCD's part), `depth` (2 | 4) rows' gathers in flight per lane; `written_through` = 1: the table written through
 * instead of plainly (the lines still stay in the writer's L2); 2: written through AND gathered by the neighbouring XCD (every gather
 * then leaves the L2 for the memory side: the path of the granules another XCD owns).  Out: GB/s of gathered bytes, us per pass, whether every group sat on one XCD.
 * DPCG_ERR_STATE when the workgroups could not be co-resident.  No reference counterpart. */
int dpcg_debug_l2_gather(int granules_per_group, int reps, const int32_t offsets[7], int depth, int written_through, dpcg_stream_t stream,
                         double *gbs, double *us_per_pass, int *groups_local);
/* Copy the current factor L out (host arrays sized from dpcg_get_info's precond_nnz). */
int dpcg_get_factor(dpcg_handle_t h, int32_t *rowptr, int32_t *col, double *val);

/* ---- standalone operators (roofline benches, unit parity, duck-typed `@`) -------------------- */
int dpcg_spmv(dpcg_handle_t h, const double *x, double *y, dpcg_stream_t stream);          /* cg.py:60,75 */
int dpcg_spmv_f32(dpcg_handle_t h, const float *x, float *y, dpcg_stream_t stream);        /* config C5   */
int dpcg_precond_apply(dpcg_handle_t h, const double *r, double *z, dpcg_stream_t stream); /* cg.py:61,81 */
int dpcg_sptrsv(dpcg_handle_t h, int upper, const double *rhs, double *out, dpcg_stream_t stream);
/* *out_host = <a,b> (torch.inner, cg.py:17,76,78,82); deterministic two-stage reduction. */
int dpcg_dot(int64_t n, const double *a, const double *b, double *out_host, dpcg_stream_t stream);
/* The SpMV fused with <p,Ap> exactly as launched inside the PCG iteration (for kernel timing). */
/* (times the kernel a default solve launches: with two-kernel updates that is the SpMV fused with the vector update) */
int dpcg_spmv_dot_bench(dpcg_handle_t h, const double *x, double *y, int repeats, float *ms_per_launch,
                        dpcg_stream_t stream);

/* The HBM streaming ceiling of this box with the library's own access shape (SURVEY.md 8-d2 asks for a measured ceiling
 * beside the 8 TB/s spec; nothing in the reference corresponds -- its loop runs on torch CPU/CUDA ops, cg.py:75-86): per 16
 * bytes written, n_read x 16 contiguous bytes are read (n_read = 1 copy, 2 triad, 4, or 11 = the read:write ratio of a
 * 7-point CSR SpMV); write = 0: read-only, n_read x out_bytes are only reduced.  nontemporal: bit 0 = non-temporal loads and
 * stores; bit 1 = the streams are WALKED TOGETHER by the whole grid (workgroup b takes pieces b, b + G, ...) instead of one
 * contiguous slab per workgroup -- a copy then is one 16-byte element per thread, the "float4 copy" MI355X_MICROARCH.md quotes
 * 6.29 TB/s for (measured here: 6.3; slabs 5.3).  `repeats` launches between HIP events on `stream`; *bytes_per_launch = reads + writes of one launch. */
int dpcg_stream_bench(int n_read, int write, int nontemporal, int64_t out_bytes, int repeats, float *ms_per_launch,
                      int64_t *bytes_per_launch, dpcg_stream_t stream);

/* ---- the solve: cg.py:50-90 (PCG) and cg.py:20-47 (CG = PCG with M = I, test on r) ----------- */
/*
 * b, x0 (may be NULL = zeros, cg.py:58), x (out, may be NULL): device fp64[n].  x is written in stream order: valid for
 * work enqueued on `stream` after the call, and for the host once `stream` is synchronised (the scalar outputs and the
 * host-side histories are complete on return).
 * Stop when res_k = <r_k,r_k>/<b,b> < rtol_sq or <r_k,r_k> < atol_sq (cg.py:15-17,71: SQUARED
 * ratio; atol_sq = 0 for the reference, 1e-12 for generate_data.py:107), k = 0 tested on
 * <z_0,z_0>/<b,b> unless DPCG_INIT_CHECK_R (cg.py:66).  At most max_iter updates (cg.py:70).
 * iters = completed updates = len(errors)-1 (cg.py:90).  seconds = host wall time around the
 * iteration loop, device-synchronised (cg.py:69,88).  res_history: host fp64[max_iter+1] or NULL,
 * entry k is what the reference appends to `errors` (cg.py:67,88).
 * x_true / err_history (both NULL for PCG): when given, err_history[k] = (x_k-x_true)^T A (x_k-x_true)
 * (cg.py:27-29,43-45), one extra SpMV per iteration as in the reference.
 * Returns DPCG_OK / DPCG_MAX_ITER / DPCG_BREAKDOWN or a negative error.
 */
int dpcg_solve(dpcg_handle_t h, const double *b, const double *x0, double *x, double rtol_sq, double atol_sq,
               int max_iter, int flags, dpcg_stream_t stream, int *iters, double *final_res, double *seconds,
               double *res_history, const double *x_true, double *err_history);

/* `count` independent systems (train.py:90-108 / test.py:121-149 loop over samples): handles[i],
 * b[i], x0[i], x[i] as above; outputs are arrays of length count.  Systems are interleaved on
 * `n_streams` internal streams (1..8).  Returns the worst status. */
int dpcg_solve_batch(int count, dpcg_handle_t *handles, const double *const *b, const double *const *x0,
                     double *const *x, double rtol_sq, double atol_sq, int max_iter, int flags, int n_streams,
                     int *iters, double *final_res, double *seconds, int *status);

/* ---- synthetic pressure-Poisson systems generated on the device -------------- */
/* dim = 2: 5-point, n*n rows; dim = 3: 7-point, n^3 rows.  Sizes via dpcg_poisson_sizes. */
int dpcg_poisson_sizes(int dim, int64_t n, int64_t *rows, int64_t *nnz);
int dpcg_gen_poisson(int dim, int64_t n, int32_t *rowptr, int32_t *col, void *val, int val_dtype,
                     dpcg_stream_t stream);

/* ---- sparse_matvec_mul (utils.py:15-43): batched COO SpMV / SpMV^T, fp32 ---------------------- */
/* indices: int32[nnz*3] rows of (batch,row,col); features fp32[nnz]; vectors/out fp32[batch*dof]. */
int dpcg_batched_coo_spmv(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof,
                          const float *vectors, float *out, int transpose, dpcg_stream_t stream);

/* Gradient of sparse_matvec_mul w.r.t. the matrix entries (training through frobenius_loss, metrics.py:28-29):
 * out[k] = a[batch_k, row_k] * c[batch_k, col_k] with (row,col) as in dpcg_batched_coo_spmv for `transpose`. */
int dpcg_batched_coo_edge(int64_t nnz, const int32_t *indices, int batch, int64_t dof, const float *a, const float *c,
                          float *out, int transpose, dpcg_stream_t stream);

/* The same two operators on PANELS of `ncols` right-hand sides (fp32[batch*dof*ncols], row-major [b][row][c]): the
 * building blocks of `inverse_loss` (metrics.py:34-55, the loss train.py:59 minimises) WITHOUT the reference's dense
 * N x N products: || L L^T A - I ||_F is accumulated over panels of columns J as L (L^T A[:, J]) - I[:, J].
 * spmm: out[b, row_k, :] += features[k] * panel[b, col_k, :]; sddmm: out[k] = <g[b, row_k, :], panel[b, col_k, :]> (the
 * gradient of spmm with respect to features[k]).  (row, col) swap under `transpose` as in dpcg_batched_coo_spmv. */
int dpcg_batched_coo_spmm(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof, int ncols,
                          const float *panel, float *out, int transpose, dpcg_stream_t stream);
int dpcg_batched_coo_sddmm(int64_t nnz, const int32_t *indices, int batch, int64_t dof, int ncols, const float *g,
                           const float *panel, float *out, int transpose, dpcg_stream_t stream);

/* ---- coordinate triplets -> CSR on the device (the reference's file formats are COO: scipy npz,
 * generate_data.py:109; OpenFOAM `i,j,value` dump, pEqn.H:98-108; StAn npz, data_set.py:186-188) ------------- */
/* rows/cols int32[nnz], vals fp64[nnz]: device.  Stable sort by (row,col), duplicates summed in storage order.
 * rowptr int32[n+1], col_out int32[nnz], val_out fp64[nnz]: device, capacity nnz; *nnz_out = unique entries. */
int dpcg_coo_to_csr(int64_t n, int64_t nnz, const int32_t *rows, const int32_t *cols, const double *vals,
                    int32_t *rowptr, int32_t *col_out, double *val_out, int64_t *nnz_out, dpcg_stream_t stream);

/* ---- the CNN that emits L: PreconditionerNet's sparse convolutions (model.py:13-59) --------------
 * The reference runs them through spconv (CUDA only, pyproject.toml:20).  Here: a PLAN per sparsity pattern -- every
 * layer's active sites and its rulebook, built on the device -- and a FORWARD that runs the layers as gathered GEMMs on the
 * fp32 matrix cores with bias and PReLU (model.py:28,37) fused, the last pointwise layer fused with model.py:53-57 (strict
 * upper part zeroed, softplus on the diagonal), L written straight into a lower-triangular CSR with fp64 values -- what
 * dpcg_set_precond_llt takes; test.py:102-105 densifies instead.  Regular sparse convolution, stride 1, windows up to
 * 2 x 2: an output site is active when its window holds an input site; out(y, x) = sum in(y + ky - ph, x + kx - pw) W[ky, kx].
 *   indices: int32 (nnz, 3) rows of (batch, row, col), device, sorted by (batch, row, col) (what data_set.py:122 emits);
 *   kernel_hw / padding_hw: host, 2 ints per layer;  weights[l]: device fp32 (C_out, kh, kw, C_in) -- spconv's KRSC layout;
 *   biases[l] (may be NULL), prelu[l] (device, ONE slope as nn.PReLU(); NULL = no activation after layer l): device fp32. */
typedef struct dpcg_convnet_plan *dpcg_convnet_plan_t;
int dpcg_convnet_plan_create(dpcg_convnet_plan_t *out, int batch, int64_t height, int64_t width, int64_t nnz,
                             const int32_t *indices, int n_layers, const int32_t *kernel_hw, const int32_t *padding_hw,
                             dpcg_stream_t stream);
/* The same for ANOTHER pattern in an existing plan, reusing its device memory (a plan per matrix of a data set then costs
 * no allocations after the first).  On failure the plan is left empty (usable only for another rebuild or destroy). */
int dpcg_convnet_plan_rebuild(dpcg_convnet_plan_t plan, int batch, int64_t height, int64_t width, int64_t nnz,
                              const int32_t *indices, int n_layers, const int32_t *kernel_hw, const int32_t *padding_hw,
                              dpcg_stream_t stream);
int dpcg_convnet_plan_destroy(dpcg_convnet_plan_t plan);
/* active sites and image size after layer `layer`; nnz_lower: entries with col <= row of the LAST layer's sites */
int dpcg_convnet_plan_info(dpcg_convnet_plan_t plan, int layer, int64_t *sites, int64_t *height, int64_t *width,
                           int64_t *nnz_lower);
/* the output sites as (sites, 3) indices sorted by (batch, row, col), and the pattern of the lower-triangular CSR over
 * batch * height rows (sample b = rows [b * height, (b + 1) * height)); device arrays, any may be NULL */
int dpcg_convnet_plan_output(dpcg_convnet_plan_t plan, int32_t *indices_out, int32_t *lower_rowptr, int32_t *lower_col,
                             dpcg_stream_t stream);
/* channels: host, n_layers + 1.  features_in: device fp32 (nnz, channels[0]).  features_out: device fp32 (sites, channels[n])
 * or NULL.  lower_val: device fp64 [nnz_lower] or NULL; lower_softplus = 1 applies model.py:53-57 (both need a pointwise last
 * layer with one output channel).  Enqueues on `stream`; hidden features live in plan-owned buffers. */
int dpcg_convnet_forward(dpcg_convnet_plan_t plan, const int32_t *channels, const float *const *weights,
                         const float *const *biases, const float *const *prelu, const float *features_in,
                         float *features_out, double *lower_val, int lower_softplus, dpcg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPCG_H */
