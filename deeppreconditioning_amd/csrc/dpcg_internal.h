// Internal declarations shared by the HIP translation units of libdpcg.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/dpcg.h"

namespace dpcg {

// ---- launch geometry -----------------------------------------------------------------------
// 256-thread workgroups = 4 wave64s, one per SIMD of a CU.  Persistent-style grids: at most
// kMaxGrid workgroups (4 per CU on the 256 CUs of an MI355X) that stride over the work, so the
// number of reduction partials is bounded and every kernel after a reduction can re-reduce the
// partials itself (deterministically) instead of waiting on a host round trip or a float atomic.
constexpr int kBlock = 256;
constexpr int kMaxGrid = 1024;        // vector kernels: 4 workgroups per CU (same box, 1M DoF, us per Jacobi update: 512 -> 31.5, 1024 -> 31.2, 1536 -> 32.5, 2048 -> 33.3)
constexpr int kMaxSpmvGrid = 2048;    // SpMV: 8 workgroups per CU = 32 waves per CU (40 VGPRs, 16 KiB LDS each)
constexpr int kStreamCap = 2048;      // products staged in LDS per 256-row block (16 KiB)
constexpr int kStreamRows = 256;      // rows per row-block of the CSR-stream SpMV
constexpr int kSfsBlock = 512;        // rows per block of the CSR-stream form of the level-major sync-free solve

enum SpmvKernel { SPMV_STREAM = 0, SPMV_VECTOR = 1, SPMV_TILE = 2 };
constexpr int kTileChunk = 64;        // x is staged in LDS in chunks of 64 doubles (512 B, one wave-load)
constexpr int kTileMaxChunks = 40;    // at most 40 chunks (20 KiB) per 256-row block
constexpr int kRingMaxLevels = 8192;   // levels per LDS-ring segment (their offsets are staged in LDS)
constexpr int kTileMinBlocks = 1536;   // fewest 256-row blocks for which the x-tile SpMV is chosen
constexpr int kReorderMinRows = 65536;  // DPCG_REORDER_AUTO leaves smaller systems alone (x stays cache-resident anyway)
constexpr int kTileTableMax = 4096;   // chunk-id span a block may cover (262,144 columns)
constexpr uint16_t kTileDiag = 0xffff; // local index of a row's own diagonal in the x-tile plan of a colour sweep (no slot: times 1.0)

struct CsrDev {
    int64_t n = 0, nnz = 0;
    int32_t *rowptr = nullptr;
    int32_t *col = nullptr;
    double *val = nullptr;    // fp64 values (always present after create)
    float *val32 = nullptr;   // fp32 copy (created on demand / when given fp32)
    int val32_lossless = 0;   // 0 unknown, 1 every (double)(float)val == val, -1 not
    bool owned = false;
};

struct SpmvPlan {
    int kernel = SPMV_VECTOR;
    int grid = 1;
    int nrb = 0;          // row-blocks of kStreamRows rows (stream kernel)
    int tpr = 4;          // threads per row (vector kernel)
    // x-tile plan (SPMV_TILE): per row-block the list of 64-double chunks of x its columns touch, and per
    // non-zero a 16-bit index into the LDS tile those chunks are staged in.  Owned by the plan.
    int32_t *tile_chunks = nullptr;    // [nrb][kTileMaxChunks]
    int32_t *tile_nchunks = nullptr;   // [nrb]
    uint16_t *tile_lidx = nullptr;     // [nnz]
    int tile_max_chunks = 0;           // largest chunk count of any block (sizes the dynamic LDS)
    bool stream_nt = false;            // x-tile kernel: once-read streams and y non-temporal (streams beyond the Infinity Cache)
    bool tile_mixed = false;           // x-tile kernel: some blocks have no tile (tile_nchunks = -1) and gather instead
    int cyclic = 0;                    // x-tile kernel: row blocks dealt out cyclically (b, b + G, ...) instead of in slabs; 2: XCD runs inside a pass (see k_spmv_tile)
    int max_row_len = 0;               // longest row (the team / chip kernels keep rows of <= 7 entries on chip); systems of <= 1 048 576 rows
    int max_band = -1;                 // largest |col - row| (the chip kernel keeps columns as 16-bit offsets from the row); -1: not measured
};

// Device-resident scalar state of one solve.  Only block 0 of a kernel writes it; everybody else
// reads values written by an EARLIER kernel, so no intra-kernel hand-off is needed.  This holds for `done` too:
// K3 (whose workgroup 0 sets it) never reads it -- it tests `done_seen`, the copy that the head of the same update
// (K1, an earlier kernel) latched -- so a workgroup of K3 that is dispatched after workgroup 0 has finished cannot
// skip its slice of x += alpha p (cg.py:79 precedes the test of cg.py:86).
struct Scalars {
    double rz;         // <r,z> of the current iterate (cg.py:76)
    double rz_next;    // <r,z> of the iterate K3 has just produced; K1 rotates it into rz
    double alpha;      // step length of the current update (cg.py:78), written by K2 for K3
    double bb;         // <b,b> (cg.py:17), computed once
    double res;        // last tested squared relative residual
    double rtol_sq;    // cg.py:71 threshold
    double atol_sq;    // absolute threshold on <r,r> (0 for the reference)
    double alpha_prev; // deferred-x form only: the step length of the previous update
    double rz_prev;    // two-kernel iteration only: <r,z> of the previous iterate (+inf before the first update)
    int k;             // completed updates
    int done;          // 1 once the stopping test held (kernels become no-ops)
    int status;        // dpcg_status of the solve
    int done_seen;     // `done` as the head (K1) of the current update saw it; what K3 tests (see above)
    // Host-visible progress word (pinned, mapped): (k << 1) | done, written by the one thread that runs
    // the stopping test.  The host steers its run-ahead from it without copies, events or syncs.
    unsigned long long *progress;
};

// One system of the whole-solve kernel for small systems (dpcg_small.hip): everything a workgroup needs.
constexpr int kSmallMaxN = 6144;      // 6 rows per thread of a 1024-thread workgroup; 3 LDS vectors = 144 KiB
struct SmallEll {                     // slab-ELL copy of a small matrix (dpcg_small.hip)
    int W = 0;                        // entries per row (max row length)
    int32_t *col = nullptr;
    double *val = nullptr;
};
struct SmallDesc {
    int n, precond, max_iter, init_check_r, hist_cap, lds_vectors;
    int variant;                      // rows-per-thread * 16 + register width of the kernel variant (dpcg_small.hip)
    const int32_t *rp;                // row pointers of A (row lengths)
    const double *dinv;
    const int32_t *m_rp;              // row pointers of CSR M, or of L for LLT_MULTIPLY
    const int32_t *t_rp;              // row pointers of L^T for LLT_MULTIPLY
    SmallEll ell_a, ell_m, ell_t;
    const double *b, *x0;
    double *x, *hist;
    double rtol_sq, atol_sq;
    Scalars *out;
};

struct Levels {
    int n_levels = 0;
    int32_t *rows = nullptr;               // device: rows sorted by (level, row)
    std::vector<int32_t> level_ptr;        // host: offsets into rows, n_levels+1
    // [lo,hi) levels; merged = one workgroup walks them; ring_w > 0: the solution entries the segment's rows depend
    // on lie within the last ring_w level-order positions, so they are handed from level to level through LDS
    // syncfree: the segment's rows (several levels) are solved by ONE multi-workgroup launch in which a row polls the
    // entries it depends on until they have been written (k_sptrsv_syncfree_rec)
    struct Segment { int lo, hi; bool merged; int ring_w; int max_width; bool syncfree = false; };
    std::vector<Segment> segments;
    int32_t *level_ptr_dev = nullptr;      // device copy of level_ptr
    // the factor once more, rows stored in level order (row j of this copy = original row rows[j]): the
    // rows of a level are contiguous, so a level streams its val/col segment coalesced like the SpMV does
    int32_t *lo_rowptr = nullptr;
    int32_t *lo_col = nullptr;
    int32_t *lo_cpos = nullptr;            // level-order position of each entry's column (position of row lo_col[k])
    double *lo_val = nullptr;
    bool stream_ok = false;                // every 256-row block of every wide level fits the LDS product buffer
    // Fixed-width records of the rows of ring segments, indexed by level-order position j, so that a row's data
    // can be requested several levels ahead without first reading its extents:
    //   pk_meta[j] = {cpos0, cpos1, cpos2, original row}: level-order positions of the first three off-diagonal
    //                columns (-1 = no such entry; cpos0 = -2: the row takes the general path through lo_rowptr);
    //   pk_val[4j..4j+3] = {v0, v1, v2, diagonal}.
    int32_t *pk_meta = nullptr;
    double *pk_val = nullptr;
    double *b_lo = nullptr;                // scratch: the right-hand side gathered into level order
    int32_t *sf_meta = nullptr;            // records of the sync-free kernel (dpcg_analysis.hip: k_sf_records)
    double *sf_val = nullptr;
    int rec_w = 3;                         // entries a record holds (3, 6 or 14); longer rows walk the level-ordered copy
    // Level-major sync-free solve in CSR-stream form (k_sptrsv_syncfree_stream): 256-row blocks that never straddle a level,
    // {first position, end position} per block; null: the record kernel
    int32_t *sfs_blk = nullptr;
    int sfs_nblk = 0;
    // Strip-pipelined solve (k_sptrsv_strips): the rows once more, sorted by (strip, strip-local level, row), with their
    // own level offsets (n_strips * nlev + 1 entries), level-ordered factor copy and records.  n_strips == 0: not used.
    struct Strips {
        int n_strips = 0, nlev = 0, W = 0, ring_reach = 0, threads = 0, rows_per_thread = 1, long_rows = 0;
        int32_t *rows = nullptr, *level_ptr_dev = nullptr, *lo_rowptr = nullptr, *lo_col = nullptr, *lo_cpos = nullptr;
        double *lo_val = nullptr, *val = nullptr, *b_lo = nullptr;
        int32_t *meta = nullptr;
        unsigned int *ticket = nullptr;    // [0] strips handed out, [1] exits
    } strips;
    unsigned long long *tickets = nullptr; // one monotonic block-ticket counter per segment (sync-free segments use theirs)
    int *spin_err = nullptr;               // set by a sync-free kernel whose bounded poll ran out
    // Level-major solve (few, wide levels whose rows lie all over the vector: every level would otherwise touch every
    // line of the right-hand side, of the solution and -- through its gathers -- of the solution again).  The solve runs
    // in the factor's OWN level-order numbering: position j holds row rows[j]; records and the level-ordered copy address
    // columns by position (lo_cpos), so a level reads and writes one contiguous run and gathers from the runs before it.
    // One gather brings the right-hand side in (lm_rhs), one takes the result out (through lm_pos = the inverse of rows).
    bool level_major = false;
    int32_t *lm_pos = nullptr;             // handle index -> level-order position
    int32_t *lm_from_lower = nullptr;      // L^T only: position here -> position in L's numbering (lower result feeds the upper solve)
    double *lm_rhs = nullptr, *lm_out = nullptr;
    // Colour sweeps (a level-major factor of a few very wide levels whose 256-row blocks fit the LDS product buffer --
    // IC(0) in multicolour order): one CSR-stream launch per level that gathers its right-hand side itself, writes the
    // result by position AND in the handle's numbering and sums <r,z> on the way (k_lm_sweep): no way-in / way-out passes.
    bool sweep = false;
    // workgroups per level launch.  An apply leaves sweep_grid partials of <r,z>: workgroup i of every launch that contributes
    // adds its share to slot i (the first such launch of the apply stores instead) -- one fixed order of additions per slot,
    // whatever the number of levels
    int sweep_grid = 0;
    // Tiled sweeps: an x-tile plan (as SpmvPlan's) per level of the level-ordered copy -- blocks of 256 rows counted from the
    // level's first row; the solution entries a block gathers are staged in LDS in 64-entry chunks (k_lm_sweep_tile).
    int32_t *sw_chunks = nullptr, *sw_nchunks = nullptr;
    uint16_t *sw_lidx = nullptr;
    std::vector<int> sw_blk0;              // host: first block of each level in sw_chunks / sw_nchunks
    std::vector<int> sw_max_chunks;        // host: per level the largest chunk count of a block (0: the level keeps the gather sweep)
    bool sweep_nt = false;                 // the factor's stream exceeds the Infinity Cache: read non-temporally
    bool sweep_cyclic = false;             // ... and its blocks dealt out cyclically, three workgroups per CU (as the x-tile SpMV's)
    // L only, when L^T's first level is L's last one (levels of L^T = levels of L reversed): position here -> position in
    // L^T's numbering, so that the last lower sweep can emit the first level of the upper solve (z = y / d) as well
    int32_t *lm_to_upper = nullptr;
    // L only: the diagonal of the rows of the FIRST level, by row (1.0 elsewhere) -- in the PCG loop that level (y = r / d,
    // nothing to wait for) rides on the kernel that updates r (k_update_r<3>), and the sweeps start at the second level
    double *ride_diag = nullptr;
    // Invariant (factors that are one sync-free launch, single_syncfree_segment): between solves the LOWER factor's lm_out
    // holds the sync-free kernels' "pending" pattern everywhere -- set up by build_levels, restored by whoever consumed the
    // values (the way-out pass of a paired apply; launch_sptrsv itself after a standalone lower solve).  A solve with
    // SptrsvIo::fused_entry relies on it.  (A host-side flag instead would be wrong under stream capture.)
};

// Plumbing between the two solves of one preconditioner apply (all optional)
struct SptrsvIo {
    const double *lm_in = nullptr;         // the right-hand side is the lm_out of the lower solve: gathered through lm_from_lower
    bool keep_lm = false;                  // leave the result in lm_out only (the next solve picks it up there)
    const double *dot_with = nullptr;      // on the way out, also per-workgroup partials of <dot_with, result> ...
    double *dot_part = nullptr;            // ... into dot_part[0 .. dot_grid)
    int dot_grid = 0;
    bool dot_done = false;                 // set by launch_sptrsv when it did sum them
    int dot_count = 0;                     // ... and how many partials it left (dot_grid, or what the colour sweeps leave)
    // No way-in pass: the (single, sync-free) solve kernel gathers its right-hand side through the map itself; lm_out must be
    // all-pending (see Levels: the invariant for the lower factor; the lower solve's `refill` for the upper one).  `refill`: another vector of n entries that the
    // kernel (lower solve) or the way-out pass (upper solve) presets to the pending pattern for whoever solves next.
    bool fused_entry = false;
    double *refill = nullptr;
    // Colour sweeps, paired apply: the LOWER solve's last level also emits the first level of the upper solve (rows without
    // dependants: z = y / d) -- by position into pair_out[lm_to_upper[j]], by row into pair_dst, its share of <dot_with, z>
    // STORED into dot_part[0 .. sweep_grid); the UPPER solve then starts at its second level (skip_first) and adds to those.
    double *pair_out = nullptr, *pair_dst = nullptr;
    bool skip_first = false;
};
bool single_syncfree_segment(const Levels &lv);   // level-major, the whole factor one sync-free launch
void launch_fill_pending(double *v, int64_t n, hipStream_t s);
// largest entry count of the blocks {blk[2b], blk[2b+1]} of a level-ordered copy -> *out_dev (atomicMax)
void launch_sfs_block_max(const int32_t *blk, int nblk, const int32_t *lo_rowptr, int *out_dev, hipStream_t s);

// (dpcg_chip_trsv.hip; the comment is with ChipTrsvDesc below)
struct ChipTrsvLists {
    int32_t *first_blk = nullptr, *first_ent = nullptr;     // 2049 each
    int4 *blk = nullptr;
    double *val = nullptr;
    int32_t *col = nullptr;
    int n_blk = 0, n_ent = 0;
    int band = 0;              // largest |col - row| of the factor in the handle's numbering
    int max_row = 0;           // most off-diagonal entries of a row
    int n_levels = 0;
};
}  // namespace dpcg

struct dpcg_system {
    dpcg::CsrDev A;
    dpcg::SpmvPlan planA;
    int precond = DPCG_PRECOND_NONE;
    double *dinv = nullptr;
    dpcg::CsrDev M;
    dpcg::SpmvPlan planM;
    dpcg_precond_fn precond_fn = nullptr;   // DPCG_PRECOND_CALLBACK
    void *precond_user = nullptr;
    dpcg::CsrDev L, Lt;                   // the factor and its transpose in the CALLER's numbering
    dpcg::CsrDev Lp, Ltp;                 // reordered handle, multiply mode: P L P^T and P L^T P^T (what the SpMVs read)
    // A factor that lives in a numbering of its own (IC(0) in multicolour order): fmap[factor index] = handle index,
    // fmap_inv the inverse; owned.  Null: the factor is in the caller's numbering (handle index through iperm, if any).
    int32_t *fmap = nullptr, *fmap_inv = nullptr;
    int precond_colors = 0;               // colours of that ordering (0: caller's ordering)
    // The multicolour ordering of the handle's PATTERN (and its inverse), kept across preconditioner setups and
    // dpcg_update_values -- the colouring only looks at the pattern; fmap / fmap_inv alias these arrays while a factor in that
    // ordering is attached.  Dropped when the handle is renumbered or destroyed.
    int32_t *mc_perm = nullptr, *mc_iperm = nullptr;
    int mc_colors = 0;
    // An IC(0) in multicolour order, applied by colour sweeps, that dpcg_update_values PARKED instead of freeing (the handle has
    // no preconditioner meanwhile): everything that depends on the pattern only -- the factor's pattern, both schedules with their
    // tile plans and maps -- is kept, and the next dpcg_set_precond_ic0_ordered(multicolour, solve) only computes the values
    // again (refresh_parked_ic0): gather tril(Q A Q^T) through map_al, the numeric factorisation level by level, L^T's values
    // through t_order, the values of the two level-ordered copies.  Any other preconditioner call frees it.
    struct Parked {
        bool valid = false;
        dpcg::CsrDev L, Lt;
        dpcg::Levels lvlL, lvlU;
        int colors = 0;
        int32_t *map_al = nullptr;      // entry of L -> entry of the handle's A            (built at the first refresh)
        int32_t *t_order = nullptr;     // entry of L^T -> entry of L
        int32_t *rows_l = nullptr, *rows_u = nullptr;   // level-order position -> factor row, for L and for L^T
    } parked;
    dpcg::SpmvPlan planL, planLt;
    // dpcg_reorder: the handle iterates on A = P A_user P^T; perm[new] = old, iperm[old] = new (device)
    int32_t *perm = nullptr, *iperm = nullptr;
    dpcg::CsrDev A_user;                  // the caller's matrix once A has been replaced by the reordered one
    int32_t *perm_val_map = nullptr;      // entry of A -> entry of A_user (built by the first dpcg_update_values: later ones only gather)
    double gather_ratio = 0.0;            // measured x-gather line traffic / bytes used of the caller's matrix
    double *pb = nullptr, *pxt = nullptr; // b / x_true gathered into the handle's numbering
    double *pv0 = nullptr, *pv1 = nullptr;   // scratch of the standalone operators on a reordered handle
    float *pf0 = nullptr, *pf1 = nullptr;
    dpcg::Levels lvlL, lvlU;
    // work vectors (fp64[n]) and reduction partials
    double *x = nullptr, *r = nullptr, *z = nullptr, *p = nullptr, *q = nullptr, *t = nullptr, *e = nullptr;
    double *p2 = nullptr;                 // second direction buffer of the two-kernel iteration
    float *p32 = nullptr;
    double *part_pq = nullptr, *part_rz = nullptr, *part_rr = nullptr, *part_bb = nullptr;
    dpcg::Scalars *scal = nullptr;        // device
    dpcg::Scalars *scal_host = nullptr;   // pinned
    double *hist = nullptr;               // device, hist_cap doubles
    double *err_hist = nullptr;           // device, hist_cap doubles (x_true mode)
    int hist_cap = 0;
    int vec_grid = 1;
    // cached iteration graph
    dpcg::SmallDesc *small_desc = nullptr;   // device, one entry (single small solves)
    void *team_desc = nullptr;               // device, one TeamDesc (single mid-size solves, dpcg_team.hip)
    double *team_part = nullptr;             // the team's reduction partials (4 x 32 doubles)
    unsigned int *team_sync = nullptr;       // [0] barrier counter, [1] error flag
    double *chip_part = nullptr;             // the chip kernel's reduction slots (4 x 256 x 2 doubles) + 8 trace words + the error flag
    double *chip_zp = nullptr;               // ... and its published granules
    double *chip_rt = nullptr;               // M = L L^T multiplied on the chip: the published r and t = L^T r
    // M = L L^T solved on the chip (dpcg_chip_trsv.hip): the block lists of L and L^T, built at the first solve with this preconditioner
    // (trsv_state: 0 not tried yet, 1 built, -1 this factor does not fit the form) and dropped with it
    dpcg::ChipTrsvLists trsv_l, trsv_u;
    int32_t *trsv_lv0 = nullptr;
    double *trsv_diag0 = nullptr;
    double *trsv_fval = nullptr;             // the resident form's plan (<= 524 288 rows): trsv_rpt != 0
    int32_t *trsv_fcol = nullptr, *trsv_fmeta = nullptr;
    int trsv_rpt = 0, trsv_wmax = 0, trsv_band = 0, trsv_tstride = 512;
    int trsv_state = 0;
    double chip_trace_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // ... over the 256 workgroups: SpMV phase max / mean, publish max / mean, `local`
    double chip_trace_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // DPCG_CHIP_TRACE: us per update by phase of the last chip solve ([7] = updates)
    dpcg::SmallEll ell_a, ell_m, ell_t;      // slab-ELL copies of A, M (or L), L^T for the small-system kernel
    hipGraphExec_t graph_exec = nullptr;
    int graph_key = -1;
    int graph_chunk = 0;
};

namespace dpcg {

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const std::string &msg);
// ---- stream captures vs device-wide waits (dpcg_mem.hip) ----
// A device-wide wait (hipDeviceSynchronize, the implicit one of hipFree) issued while ANOTHER host thread is capturing a stream
// invalidates that capture ("operation failed due to a previous error during capture": four threads setting up and solving at
// once, tools/thread_probe.py).  Captures hold this lock shared, device-wide waits hold it exclusively -- they wait for the
// captures in progress (a few hundred microseconds) and keep new ones out while they run.
struct CaptureGuard {              // around hipStreamBeginCapture .. hipStreamEndCapture
    CaptureGuard();
    ~CaptureGuard();
};
hipError_t device_wide_wait();     // hipDeviceSynchronize under the exclusive lock
hipError_t device_free(void *p);   // hipFree under the exclusive lock

// ---- device memory with a block cache (dpcg_mem.hip) ----
hipError_t cached_alloc(void **out, size_t bytes);
void cached_free(void *p);
void release_cached_memory();
size_t cached_memory_bytes();
// Opened by the ABI entry points that build something on a stream: blocks freed inside are reusable at once by the same
// scope (one stream: ordered), and go to the process-wide pool when the scope ends -- after the stream has been waited
// for, so a setup call's results are complete when it returns.  wait_for_device: the blocks about to be freed may be in
// use on other streams (dpcg_destroy).
struct SetupScope {
    hipStream_t stream;
    int device = 0;
    bool owner = false;
    std::multimap<size_t, void *> idle;
    explicit SetupScope(hipStream_t s, bool wait_for_device = false);
    ~SetupScope();
    SetupScope(const SetupScope &) = delete;
    SetupScope &operator=(const SetupScope &) = delete;
};
int hip_fail(hipError_t e, const char *what, const char *file, int line);
#define DPCG_HIP(call)                                                         \
    do {                                                                       \
        hipError_t _e = (call);                                                \
        if (_e != hipSuccess) return ::dpcg::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

// ---- kernel launchers (dpcg_spmv.hip, dpcg_pcg.hip, dpcg_sptrsv.hip, dpcg_setup.hip) -----------------------------------------------------
// y = A x.  If part_pq != nullptr also writes per-workgroup partials of <x, y> (plan.grid of them).
// `ctl` (may be null): when given, the kernel is the head of a PCG update: it returns at once when
// the device-resident `done` word is set and rotates rz_next -> rz (see dpcg_device.h).
struct IterCtl {
    Scalars *scal;
};
// Extra operands of the SpMV kernels in the two-kernel iteration (see fused_head in dpcg_device.h): the kernel
// first forms p_k = z + beta p_{k-1} and x += alpha_{k-1} p_{k-1} (cg.py:83,79), then q = A p_k (cg.py:75).
struct FuseArgs {
    const double *z;          // preconditioned residual of the current iterate
    double *p0, *p1;          // direction vectors: update k writes P[k & 1] and reads P[(k + 1) & 1]
    double *xvec;             // the iterate, one update behind
    const double *part_rz;    // partials of <r,z> and <r,r> of the current iterate (K2 or the initial state)
    const double *part_rr;
    int n_part;               // partials of <r,z>: vec_grid, or the grid of the SpMV that applied M and summed <r,z> on the way
    int n_part_rr;            // partials of <r,r> (always KB's: vec_grid)
    double *hist;
    int hist_cap;
};
void launch_spmv_fused(const CsrDev &A, const SpmvPlan &plan, const FuseArgs &fa, double *q, double *part_pq,
                       Scalars *scal, hipStream_t s);
void launch_fused_init(Scalars *scal, hipStream_t s);
// K3 with the x update deferred to every second update (see k_update_xp_deferred); `odd`: this is update 1, 3, 5, ...
void launch_update_xp_deferred(bool odd, int64_t n, Scalars *scal, const double *part_rz, const double *part_rr, int n_part,
                               const double *z, const double *p_in, double *p_out, double *x, float *p32, double *hist,
                               int hist_cap, int grid, hipStream_t s, const double *zd, int n_part_rr, bool nt = false);
void launch_final_deferred(int64_t n, Scalars *scal, double *x, const double *p0, const double *p1, int grid,
                           hipStream_t s);
void launch_final_fused(int64_t n, Scalars *scal, const double *part_rr, int n_part, double *hist, int hist_cap,
                        double *x, const double *p0, const double *p1, int grid, hipStream_t s);
void launch_spmv(const CsrDev &A, const SpmvPlan &plan, const double *x, double *y, double *part_pq,
                 const IterCtl *ctl, hipStream_t s);
// y = A x with per-workgroup partials of <xdot, y> (plan.grid of them): the last SpMV of an M-apply sums <r,z> on the way
void launch_spmv_xdot(const CsrDev &A, const SpmvPlan &plan, const double *x, const double *xdot, double *y, double *part,
                      hipStream_t s);
void launch_spmv_f32in(const CsrDev &A, const SpmvPlan &plan, const float *x32, const double *x64, double *y,
                       double *part_pq, const IterCtl *ctl, hipStream_t s);
// fp32-stored values, fp64 x and arithmetic (exact when the values are fp32-representable)
void launch_spmv_val32(const CsrDev &A, const SpmvPlan &plan, const double *x, double *y, double *part_pq,
                       const IterCtl *ctl, hipStream_t s);
// *lossy_dev is set to 1 if some value changes under fp64 -> fp32 -> fp64; also fills val32
void launch_val32_check(int64_t nnz, const double *val, float *val32, int *lossy_dev, hipStream_t s);
void launch_spmv_f32out(const CsrDev &A, const SpmvPlan &plan, const float *x32, float *y32, hipStream_t s);

void launch_update_r_two_kernel(int precond_fused, int64_t n, Scalars *scal, const double *part_pq, int n_part_pq,
                                const double *q, double *r, const double *dinv, double *z, double *part_rz,
                                double *part_rr, int grid, hipStream_t s);
void launch_update_r(int precond_fused, int64_t n, Scalars *scal, const double *part_pq, int n_part_pq,
                     const double *q, double *r, const double *dinv, double *z, double *part_rz, double *part_rr,
                     int grid, hipStream_t s, int store_z = 1, bool nt = false);
void launch_update_r_ride(int64_t n, Scalars *scal, const double *part_pq, int n_part_pq, const double *q, double *r,
                          const double *first_level_diag, const int32_t *pos, double *lm_out, int first_level_rows,
                          double *part_rr, int grid, hipStream_t s, bool two_kernel = false);
void launch_dot_partials(int64_t n, const Scalars *scal, const double *a, const double *b, double *part, int grid,
                         hipStream_t s);
void launch_update_xp(int64_t n, Scalars *scal, const double *part_rz, const double *part_rr, int n_part,
                      const double *z, double *p, double *x, float *p32, double *hist, int hist_cap, int grid,
                      hipStream_t s, const double *zd, int n_part_rr);   // zd: z is not stored, K3 forms zd .* z itself
void launch_final_check(Scalars *scal, hipStream_t s);
void launch_init_state(int64_t n, Scalars *scal, const double *b, const double *r, const double *z, double *p,
                       float *p32, double *part_bb, double *part_rz, double *part_rr, int init_check_r, int grid,
                       hipStream_t s);
// part_rz / part_t are left in canonical form (entry 0 = the sum, entries 1 .. canon_cap-1 = 0), so that the first KA may
// read them with whatever partial count the later updates use
void launch_finalize_init(Scalars *scal, const double *part_bb, double *part_rz, double *part_t,
                          int n_part, double rtol_sq, double atol_sq, double *hist, int hist_cap,
                          unsigned long long *progress, hipStream_t s, int canon_cap = 0);
void launch_residual(int64_t n, const double *b, const double *ax, double *r, int grid, hipStream_t s);
void launch_scale(int64_t n, const double *dinv, const double *r, double *z, int grid, hipStream_t s);
void launch_extract_dinv(const CsrDev &A, double *dinv, int *bad_flag, hipStream_t s);
void launch_f64_to_f32(int64_t n, const double *in, float *out, hipStream_t s);
void launch_f32_to_f64(int64_t n, const float *in, double *out, hipStream_t s);
void launch_anorm_err(int64_t n, const Scalars *scal, const double *x, const double *x_true, double *e, int grid,
                      hipStream_t s);
void launch_record_err(const Scalars *scal, const double *part, int n_part, double *err_hist, int hist_cap,
                       int at_k_minus_one, hipStream_t s);
void launch_dot_final(const double *part, int n_part, double *out_dev, hipStream_t s);

void init_strip_kernels();
// IC(0) through a strip plan (dpcg_sptrsv.hip: k_sptrsv_strips<..., FACTOR>)
bool launch_strip_factor(const Levels &lv, double *diag, double *fac, int64_t n, hipStream_t s, const int32_t *xdesc = nullptr,
                         const double *thr = nullptr, double *offd = nullptr);
bool launch_ring_factor(const Levels &lv, double *diag, double *fac, hipStream_t s, const int32_t *xdesc = nullptr,
                        const double *thr = nullptr);
void launch_ring_factor_desc(int64_t n, const int32_t *lo_rp, const int32_t *lo_ci, const int32_t *lo_cp, const double *colnorm,
                             double tau, int32_t *xdesc, double *thr, hipStream_t s);
void launch_strip_factor_scatter(int64_t n, const int32_t *frow, const int32_t *rp, const double *fac, double *fval,
                                 const int32_t *lo_rp, double *lo_val, int *bad, hipStream_t s);
// done: optional device flag (Scalars::done); when set the kernels return at once
void launch_sptrsv(const CsrDev &T, const Levels &lv, bool upper, const double *rhs, double *out, hipStream_t s,
                   const int *done = nullptr, SptrsvIo *io = nullptr);
void launch_invert_positions(int64_t n, const int32_t *rows, int32_t *pos, hipStream_t s);          // pos[rows[j]] = j
void launch_compose_positions(int64_t n, const int32_t *rows, const int32_t *pos, int32_t *out, hipStream_t s);   // out[j] = pos[rows[j]]

// Builds the x-tile plan of A on the device; *ok = 1 when every block is tileable, *max_chunks its widest tile.
void launch_sweep_tile_plan(int j0, int count, const int32_t *lo_rowptr, const int32_t *lo_cpos, int32_t *chunks, int32_t *nchunks,
                            uint16_t *lidx, int *ok_and_max_dev, hipStream_t s);
void launch_tile_plan(const CsrDev &A, int nrb, int32_t *chunks, int32_t *nchunks, uint16_t *lidx, int *ok_and_max_dev,
                      hipStream_t s);
void launch_max_row_len(int n, const int32_t *rp, int *out_dev, hipStream_t s);
void launch_build_ell(int n, const int32_t *rp, const int32_t *ci, const double *v, int W, int32_t *ell_col,
                      double *ell_val, hipStream_t s);
int small_variant(int n, int max_row_len, int precond);
int launch_pcg_small(const SmallDesc *descs_dev, int count, int lds_bytes, int kinds_mask, int variants_mask,
                     hipStream_t s);
// dpcg_team.hip: whole-solve kernel for mid-size systems, a team of 32 workgroups per system
struct TeamDesc {
    int n, precond, max_iter, init_check_r, hist_cap, W;
    const int32_t *rp;
    const double *dinv;
    const int32_t *ell_col;
    const double *ell_val;
    const double *b, *x0;
    double *x, *hist;
    double *z, *p0, *p1;       // n doubles each: z_k and p_k / p_{k-1} (double-buffered) as the other workgroups see them
    double rtol_sq, atol_sq;
    Scalars *out;
    unsigned int *bar;         // the team's barrier counter (zero at launch)
    double *part;              // 3 sets x 2 x kTeamSize doubles, all preset to the "pending" pattern at launch
    int *err;
    int *xcc;                  // 32 words: the XCD every workgroup of the team found itself on (exchanged once per solve)
    unsigned long long *dbg;   // DPCG_TEAM_TRACE=1: 8 words, ticks (100 MHz) rank 0 spent per phase of the updates; else null
};

// dpcg_chip.hip: whole-solve kernel for cache-sized systems (65 537 .. 1 048 576 rows), the whole chip as one team
struct ChipDesc {
    int n, precond, max_iter, init_check_r, hist_cap, per;   // per = ceil(n / 256): rows per workgroup
    const int32_t *rp, *ci;
    const double *val, *dinv;
    const double *b, *x0;
    double *x, *hist;
    double *zp;                // 2 x (n + 4096) granules {z_{k+1}[i], p_k[i]}: the copy read inside a group and the written-through one
    int band;                  // largest |col - row| of the matrix
    int bench;                 // development (DPCG_CHIP_BENCH): the kernel variant without the gathers of q = A p (q = p) that never stops before max_iter
    int stream_cap;            // > 0: the matrix is streamed every update (MODE 5: rows too long or columns too far for the resident form); entries the
                               // 64 rows a wave owns in one slot hold at most (its LDS product buffer, doubles)
    int rp_nnz;                // entries of the matrix (the extent of the streamed form's col / val buffers)
    int f32;                   // DPCG_SPMV_F32 (BASELINE config 5): matrix values and p of `A @ p` stored in fp32, products and sums in fp64; x0 = 0 only
    int *xcc;                  // 256 words: the XCD every workgroup found itself on (exchanged once per solve); null: never store plainly
    double rtol_sq, atol_sq;
    Scalars *out;
    double *part;              // chip_slot_doubles() doubles, all preset to the "pending" pattern at launch
    int *err;
    unsigned long long *dbg;   // DPCG_CHIP_TRACE=1: 8 words per workgroup, ticks (100 MHz) its thread 0 spent per phase of the updates; else null
};
// dpcg_chip_trsv.hip: M applied by two triangular solves (L y = r, L^T z = y) inside the whole-chip kernel.  The block lists of one triangular
// factor in the chip kernel's geometry: per chip wave (workgroup v, wave w: index 8 v + w) the blocks [first_blk[i], first_blk[i + 1]), each
// {k | W << 8 | level << 16, lane mask lo, hi, first entry}; entries {val, col} compacted over a block's active lanes.
struct ChipTrsvDesc {
    int n, max_iter, init_check_r, hist_cap, per, band;
    const int32_t *rp, *ci;
    const double *val;
    const double *b, *x0;
    double *x, *hist;
    double *zp;                // 2 x (n + 4096) granules {z, p}
    double *ypub, *zpub;       // 2 x (n + 4096) self-validating granules each: y = L^-1 r and z = L^-T y as the other workgroups see them
    const int32_t *first_l, *first_u;
    const int4 *blk_l, *blk_u;
    const double *val_l, *val_u;
    const int32_t *col_l, *col_u;
    int nent_l, nent_u;
    // the resident form (<= 4 rows a thread): the factor by workgroup, slot, thread (k_trsv_res_plan), the level counts
    const double *fval;
    const int32_t *fcol, *fmeta;
    int nlev_l, nlev_u;
    int tstride;               // entries per slot in fval / fcol / fmeta: 512, or rows-per-workgroup rounded up to 64 (small systems)
    const int32_t *lv0;        // [256][512]: bits 0-7 the thread's slots whose row has no dependency in L, bits 8-15 in L^T
    const double *diag0;       // [256][8][512]: the factor's diagonal by workgroup, slot, thread
    double rtol_sq, atol_sq;
    Scalars *out;
    double *part;
    int *err;
    int *xcc;
    unsigned int nonce;        // per launch: keys the self-validating granules
    unsigned long long *dbg;   // DPCG_CHIP_TRACE=1: 64 words per workgroup (8 waves x 8 phases of the apply), ticks of the 100 MHz clock; else null
};
int chip_trsv_max_levels();
int chip_trsv_max_factor_row();
int build_chip_trsv_lists(int n, int per, int nlev, const CsrDev &F, const int32_t *lvl, const int32_t *f_of_handle, const int32_t *handle_of_f, bool upper,
                          ChipTrsvLists &out, int32_t *lv0, double *diag0, hipStream_t s);
int64_t chip_trsv_diag_doubles();
int build_chip_trsv_resident(int n, int per, int rpt, int wmax, const CsrDev &L, const CsrDev &U, const int32_t *lvl_l, const int32_t *lvl_u,
                             const int32_t *f_of_handle, const int32_t *handle_of_f, double **fval, int32_t **fcol, int32_t **fmeta, int *misfit, int *band,
                             int *tstride_out, hipStream_t s);
int chip_trsv_resident_rpt(int per);
int chip_trsv_resident_wmax(int max_a, int rpt);
int launch_pcg_chip_trsv_resident(const ChipTrsvDesc &d, int rpt, int wmax, hipStream_t s, bool check_only = false);
void free_chip_trsv_lists(ChipTrsvLists &l);
int launch_pcg_chip_trsv(const ChipTrsvDesc &d, int max_a, int max_l, hipStream_t s, bool check_only = false);
// dpcg_chip_llt.hip: the same for M = L L^T multiplied (6 145 .. 262 144 rows, rows of A <= 7 and of L, L^T <= 16 entries)
struct ChipLltDesc {
    int n, max_iter, init_check_r, hist_cap, per, band;
    const int32_t *rp, *ci;
    const double *val;
    const int32_t *lrp, *lci;          // L (what z = L t reads) ...
    const double *lval;
    const int32_t *trp, *tci;          // ... and L^T (what t = L^T r reads), both in the numbering the handle iterates in
    const double *tval;
    const double *b, *x0;
    double *x, *hist;
    double *zp;                // 2 x (n + 4096) granules {z, p}
    double *rpub, *tpub;       // 2 x (n + 4096) granules each: r and t = L^T r as the other workgroups see them
    double rtol_sq, atol_sq;
    Scalars *out;
    double *part;
    int *err;
    int *xcc;
    unsigned int nonce;        // != 0: r and t travel as self-validating granules keyed by (nonce, generation); unique per launch
};
int launch_l2_gather_probe(double *table, int per_group, int reps, const int *offs7_dev, int depth, int sc1_only, double *part, int *err, int *xcc,
                           unsigned long long *ticks, unsigned *sink, hipStream_t s);
int chip_llt_max_rows();
int chip_llt_max_row_len();
int launch_pcg_chip_llt(const ChipLltDesc &d, int max_a, int max_l, hipStream_t s, bool check_only = false);
int chip_max_rows();
int chip_max_row_len(int64_t n, bool f32_slots = false);
int chip_max_band();
int chip_stream_max_row_len();    // rows of the streamed form (dpcg_chip.hip MODE 5): the LDS holds the products of 512 of them
int chip_workgroups();
int chip_threads();
int chip_slot_doubles();          // reduction slots (doubles) of a chip solve
int64_t chip_zp_doubles(int64_t n);  // granule storage (doubles) of a chip solve
void launch_band_and_len(const CsrDev &A, int *out2_zeroed_dev, hipStream_t s);
int launch_pcg_chip(const ChipDesc &d, int max_row_len, hipStream_t s, bool check_only = false);
int launch_occupy(int workgroups, double ms, hipStream_t s);
int team_max_rows();
int team_max_row_len();
int launch_pcg_team(const TeamDesc *descs_dev, int nsys, int max_slabs_per_wg, int max_row_len, hipStream_t s, bool trace = false);
void launch_ic0_level(const int32_t *rows, int j0, int count, const int32_t *rp, const int32_t *ci, double *lv, int *bad,
                      hipStream_t s, const double *colnorm = nullptr, double tau = 0.0);   // colnorm: ICT drop rule
void launch_colnorm1(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, double *c, hipStream_t s);
void launch_ict_pattern(bool write, int64_t n, const int32_t *rp, const int32_t *ci, const double *v, int fill, int32_t *cnt,
                        const int32_t *lrp, int32_t *lci, double *lv, int *flags, hipStream_t s);
void launch_count_kept(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, int32_t *cnt, hipStream_t s);
void launch_copy_kept(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, const int32_t *orp, int32_t *oci,
                      double *ov, hipStream_t s);
void launch_block_nnz_max(const CsrDev &A, int rows_per_block, int *out_max_dev, hipStream_t s);
// ---- reordering (dpcg_reorder.hip) ----
void launch_gather_f64(int64_t n, const int32_t *perm, const double *in, double *out, hipStream_t s);    // out[r] = in[perm[r]]
void launch_scatter_f64(int64_t n, const int32_t *perm, const double *in, double *out, hipStream_t s);   // out[perm[r]] = in[r]
void launch_gather_f32(int64_t n, const int32_t *perm, const float *in, float *out, hipStream_t s);
void launch_scatter_f32(int64_t n, const int32_t *perm, const float *in, float *out, hipStream_t s);
void launch_relabel(int64_t count, const int32_t *map, int32_t *idx, hipStream_t s);                    // idx[k] = map[idx[k]]
int gather_line_ratio(const CsrDev &A, double *ratio, hipStream_t s);
int permute_csr(const CsrDev &A, const int32_t *perm, const int32_t *iperm, CsrDev &B, hipStream_t s);
int rcm_order(const CsrDev &A, int32_t **perm_out, int32_t **iperm_out, int *n_components, hipStream_t s);
int multicolor_order(const CsrDev &A, int32_t **perm_out, int32_t **iperm_out, int *n_colors, hipStream_t s);
// region-by-region numbering (dpcg_reorder.hip): `regions` searches grown at once, numbered region by region, ring by ring
int region_order(const CsrDev &A, int regions, int32_t **perm_out, int32_t **iperm_out, hipStream_t s);
int region_order_default_regions(int64_t n);
// ---- structural analysis of triangular factors on the device (dpcg_analysis.hip) ----
void launch_check_lower(const CsrDev &L, int *flags, hipStream_t s);
void launch_row_of(int64_t n, const int32_t *rp, int32_t *row_of, hipStream_t s);
void launch_iota(int64_t count, int32_t *out, hipStream_t s);
void launch_group_offsets(int64_t count, const uint32_t *keys_sorted, int groups, int32_t *ptr, hipStream_t s);
void launch_transpose_gather(int64_t nnz, const int32_t *perm, const int32_t *row_of, const double *val, int32_t *tcol,
                             double *tval, hipStream_t s);
// Which strip of the strip-pipelined solve a row belongs to (idx = the row index, counted from the END for an upper factor).
// Slabs of `rows` consecutive indices (a multiple of the band: whole planes / grid lines); parts > 1 cuts every slab once more
// by the position inside the band, (idx % band) / sub_w -- for a grid: pencils instead of slabs, so that the longest
// dependency chain crosses ~(slabs + parts) strip boundaries instead of `slabs * parts`.  rows == 0: no strips.
struct StripMap {
    int rows = 0, band = 1, sub_w = 1, parts = 1;
    __host__ __device__ int64_t strip_of(int64_t idx) const {
        return parts <= 1 ? idx / rows : (idx / rows) * parts + (idx % band) / sub_w;
    }
};
void launch_levels_syncfree(int64_t n, const int32_t *rp, const int32_t *ci, bool upper, int32_t *level,
                            unsigned int *ticket_zeroed, int *err, hipStream_t s, StripMap map = StripMap());
void launch_strip_keys(int64_t n, const int32_t *level, StripMap map, int nlev, bool upper, uint32_t *key, hipStream_t s);
void launch_strip_records(int64_t n, const uint32_t *key_of_pos, int nlev, const int32_t *level_ptr, const int32_t *rows,
                          const int32_t *lo_rp, const int32_t *lo_ci, const int32_t *lo_cp, const double *lo_v, bool upper,
                          int ring_reach, int32_t *meta, double *pv, int32_t *exported_zeroed, int *stats, hipStream_t s);
void launch_max_band(int64_t n, const int32_t *rp, const int32_t *ci, bool upper, int *out_dev, hipStream_t s);
void launch_reverse_levels(int64_t n, const int32_t *rows_lo, const uint32_t *lvl_lo, const int32_t *level_ptr_lo, int nl,
                           int32_t *rows_up, uint32_t *lvl_up, hipStream_t s);
void launch_points_along_band(int64_t n, const int32_t *rp, const int32_t *ci, bool upper, int band, int *flag_zeroed, hipStream_t s);
void launch_ic0_cross_terms(int64_t n, const int32_t *rp, const int32_t *ci, int *flags_zeroed, hipStream_t s);
void launch_level_keys(int64_t n, const int32_t *level, const int32_t *order_by, uint64_t *key, hipStream_t s);
void launch_key_levels(int64_t n, const uint64_t *key_sorted, uint32_t *lvl, hipStream_t s);
void launch_lo_lengths(int64_t n, const int32_t *rows, const int32_t *rp, int32_t *len, int32_t *pos, hipStream_t s);
void launch_lo_copy(int64_t n, const int32_t *rows, const int32_t *rp, const int32_t *ci, const double *v,
                    const int32_t *pos, const int32_t *lo_rp, int32_t *lo_ci, int32_t *lo_cp, double *lo_v,
                    hipStream_t s);
void launch_stream_fit(int64_t n, const uint32_t *lvl_of_pos, const int32_t *level_ptr, const int32_t *lo_rp, int *flag,
                       hipStream_t s);
void launch_ring_reach(int64_t n, const uint32_t *lvl_of_pos, const int32_t *seg_of_level, const int32_t *seg_start,
                       const int32_t *lo_rp, const int32_t *lo_cp, int32_t *maxdist, hipStream_t s);
void launch_ring_records(int64_t n, const uint32_t *lvl_of_pos, const int32_t *ring_start_of_level, const int32_t *rows,
                         const int32_t *lo_rp, const int32_t *lo_ci, const int32_t *lo_cp, const double *lo_v,
                         int32_t *meta, double *pv, hipStream_t s);
void launch_sf_records(int64_t n, const int32_t *rows, const int32_t *lo_rp, const int32_t *lo_ci, const double *lo_v,
                       bool upper, int32_t *meta, double *pv, int width, hipStream_t s);
void launch_count_long_rows(int64_t n, const int32_t *lo_rp, int limit, int *counter, hipStream_t s);
void launch_tril_count(int64_t n, const int32_t *rp, const int32_t *ci, int32_t *cnt, int *flag, hipStream_t s);
void launch_iota_f64(int64_t count, double *out, hipStream_t s);
void launch_f64_to_i32(int64_t count, const double *in, int32_t *out, hipStream_t s);
void launch_lo_values(int64_t n, const int32_t *rows, const int32_t *rp, const double *v, const int32_t *lo_rp, double *lo_v,
                      hipStream_t s);
void launch_tril_copy(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, const int32_t *lrp, int32_t *lci,
                      double *lv, hipStream_t s);
void launch_gen_poisson(int dim, int64_t n, int32_t *rowptr, int32_t *col, void *val, int val_dtype, hipStream_t s);
int64_t launch_stream_bench(int n_read, bool write, bool nt, int64_t out_bytes, const double *in, double *out, double *part,
                            int grid, hipStream_t s, int in_flight = 2);
void launch_batched_coo_edge(int64_t nnz, const int32_t *indices, int batch, int64_t dof, const float *a, const float *c,
                             float *out, int transpose, hipStream_t s);
void launch_batched_coo_spmm(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof, int ncols,
                             const float *B, float *out, int transpose, hipStream_t s);
void launch_batched_coo_sddmm(int64_t nnz, const int32_t *indices, int batch, int64_t dof, int ncols, const float *G,
                              const float *B, float *out, int transpose, hipStream_t s);
void launch_batched_coo_spmv(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof,
                             const float *vectors, float *out, int transpose, hipStream_t s);

// ---- device helpers shared by the kernel translation units --------------------------------------------
// Cross-lane moves on the DPP path (no LDS crossbar round trip as with ds_bpermute/__shfl): ~8 cycles per step
// instead of ~70.  A 64-bit value moves as two 32-bit halves.  Lanes without a source lane read 0 (bound_ctrl).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes of a wave in a fixed order; the result is valid in lane 63.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x114, 0xf>(v);   // row_shr:4
    v += dpp_move<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of each row holds the row sum
    v += dpp_move<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return v;
}


}  // namespace dpcg
