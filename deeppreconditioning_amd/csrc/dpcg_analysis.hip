// Structural analysis of triangular factors on the device (setup of LLT_SOLVE / IC(0)): validation, transpose, level
// sets, level-ordered copies, LDS-ring records.  Integer work on data that already lives in HBM -- a factor that comes
// "straight from the CNN" (dpcg_set_precond_llt with DEVICE pointers) never crosses PCIe.  Sort / scan come from
// dpcg_prims.h; everything else is a plain kernel.
#include "dpcg_device.h"

namespace dpcg {

namespace {
inline int grid_rows(int64_t n, int cap = 8192) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// validation: lower triangular, columns ascending, diagonal stored last and positive
// flags[0]: bit 0 structure, bit 1 pivot
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_check_lower(int64_t n, const int32_t *__restrict__ rp,
                                                        const int32_t *__restrict__ ci, const double *__restrict__ v,
                                                        int *flags) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int a = rp[i], b = rp[i + 1];
        if (b <= a || ci[b - 1] != i) {
            bad |= 1;
            continue;
        }
        for (int k = a; k < b - 1; ++k)
            if (ci[k] >= ci[k + 1] || ci[k] < 0) bad |= 1;
        if (!(v[b - 1] > 0.0)) bad |= 2;
    }
    if (bad) atomicOr(flags, bad);
}

void launch_check_lower(const CsrDev &L, int *flags, hipStream_t s) {
    hipLaunchKernelGGL(k_check_lower, dim3(grid_rows(L.n)), dim3(kBlock), 0, s, L.n, L.rowptr, L.col, L.val, flags);
}

// ------------------------------------------------------------------------------------------------
// transpose pieces
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_row_of(int64_t n, const int32_t *__restrict__ rp, int32_t *__restrict__ row_of) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        for (int k = rp[i]; k < rp[i + 1]; ++k) row_of[k] = (int32_t)i;
}

void launch_row_of(int64_t n, const int32_t *rp, int32_t *row_of, hipStream_t s) {
    hipLaunchKernelGGL(k_row_of, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rp, row_of);
}

__global__ __launch_bounds__(kBlock) void k_iota(int64_t count, int32_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < count; k += stride) out[k] = (int32_t)k;
}

void launch_iota(int64_t count, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_iota, dim3(grid_rows(count)), dim3(kBlock), 0, s, count, out);
}

// key64[i] = level[i] << 32 | (order_by ? order_by[i] : i): rows sorted by it are grouped by level and, inside a level, ordered
// by `order_by` (the handle's numbering on a reordered handle: the solve kernels then touch r / z in ascending addresses)
__global__ __launch_bounds__(kBlock) void k_level_keys(int64_t n, const int32_t *__restrict__ level,
                                                       const int32_t *__restrict__ order_by, uint64_t *__restrict__ key) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        key[i] = ((uint64_t)(uint32_t)level[i] << 32) | (uint32_t)(order_by ? order_by[i] : (int32_t)i);
}

__global__ __launch_bounds__(kBlock) void k_key_levels(int64_t n, const uint64_t *__restrict__ key, uint32_t *__restrict__ lvl) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) lvl[i] = (uint32_t)(key[i] >> 32);
}

// The level sets of a lower factor, read backwards, are level sets of its transpose: all dependencies of the upper solve run from
// a higher lower-level to a lower one, and the rows of one lower level do not depend on each other in either direction.  Position
// j of lower level l (range [p0, p1)) becomes position (n - p1) + (j - p0) of upper level nl - 1 - l: the rows of a level keep
// their order.  lvl_lo[j] = lower level of position j.
__global__ __launch_bounds__(kBlock) void k_reverse_levels(int64_t n, const int32_t *__restrict__ rows_lo,
                                                           const uint32_t *__restrict__ lvl_lo,
                                                           const int32_t *__restrict__ level_ptr_lo, int nl,
                                                           int32_t *__restrict__ rows_up, uint32_t *__restrict__ lvl_up) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const int l = (int)lvl_lo[j];
        const int p0 = level_ptr_lo[l], p1 = level_ptr_lo[l + 1];
        const int64_t ju = (n - p1) + (j - p0);
        rows_up[ju] = rows_lo[j];
        lvl_up[ju] = (uint32_t)(nl - 1 - l);
    }
}

void launch_reverse_levels(int64_t n, const int32_t *rows_lo, const uint32_t *lvl_lo, const int32_t *level_ptr_lo, int nl,
                           int32_t *rows_up, uint32_t *lvl_up, hipStream_t s) {
    hipLaunchKernelGGL(k_reverse_levels, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows_lo, lvl_lo, level_ptr_lo, nl, rows_up, lvl_up);
}

void launch_level_keys(int64_t n, const int32_t *level, const int32_t *order_by, uint64_t *key, hipStream_t s) {
    hipLaunchKernelGGL(k_level_keys, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, level, order_by, key);
}
void launch_key_levels(int64_t n, const uint64_t *key_sorted, uint32_t *lvl, hipStream_t s) {
    hipLaunchKernelGGL(k_key_levels, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, key_sorted, lvl);
}

// Offsets of the groups of a SORTED key array: ptr[g] = first position with key >= g, for g = 0..groups (ptr[groups] =
// count).  Empty groups get the offset of the next non-empty one.
__global__ __launch_bounds__(kBlock) void k_group_offsets(int64_t count, const uint32_t *__restrict__ keys, int groups,
                                                          int32_t *__restrict__ ptr) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k <= count; k += stride) {
        const int64_t lo = k == 0 ? 0 : (int64_t)keys[k - 1] + 1;
        const int64_t hi = k == count ? groups : (int64_t)keys[k];
        for (int64_t g = lo; g <= hi; ++g) ptr[g] = (int32_t)k;
    }
}

void launch_group_offsets(int64_t count, const uint32_t *keys_sorted, int groups, int32_t *ptr, hipStream_t s) {
    hipLaunchKernelGGL(k_group_offsets, dim3(grid_rows(count + 1)), dim3(kBlock), 0, s, count, keys_sorted, groups, ptr);
}

// Entries of the transposed matrix from the sort permutation: tcol[d] = row_of[perm[d]], tval[d] = val[perm[d]].
__global__ __launch_bounds__(kBlock) void k_transpose_gather(int64_t nnz, const int32_t *__restrict__ perm,
                                                             const int32_t *__restrict__ row_of,
                                                             const double *__restrict__ val, int32_t *__restrict__ tcol,
                                                             double *__restrict__ tval) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x; d < nnz; d += stride) {
        const int32_t k = perm[d];
        tcol[d] = row_of[k];
        tval[d] = val[k];
    }
}

void launch_transpose_gather(int64_t nnz, const int32_t *perm, const int32_t *row_of, const double *val, int32_t *tcol,
                             double *tval, hipStream_t s) {
    hipLaunchKernelGGL(k_transpose_gather, dim3(grid_rows(nnz)), dim3(kBlock), 0, s, nnz, perm, row_of, val, tcol, tval);
}

// ------------------------------------------------------------------------------------------------
// Level sets: level(i) = 1 + max level of the rows i depends on (0 without dependencies).
// ONE launch, no host round trip: a thread owns a row and polls the levels of its dependencies until they have been
// written (level[] starts at -1); values travel as 4-byte agent-scope accesses, flag and payload in one word.  Rows are
// handed out in dependency-compatible order -- ascending for a lower factor, descending for an upper one -- through a
// ticket counter, so every row a workgroup waits for belongs to a workgroup that has already started: no deadlock
// whatever the dispatch order.  The store sits inside the poll loop because lanes of one wave may depend on each other.
// ------------------------------------------------------------------------------------------------
// map.rows > 0: STRIP-LOCAL levels -- dependencies on rows of another strip (StripMap::strip_of) are ignored; see the
// strip-pipelined solve.
template <bool UPPER>
__global__ __launch_bounds__(kBlock) void k_levels_syncfree(int64_t n, const int32_t *__restrict__ rp,
                                                            const int32_t *__restrict__ ci, int32_t *level,
                                                            unsigned int *ticket, int *err, StripMap map) {
    __shared__ unsigned int s_lb;
    if (threadIdx.x == 0) s_lb = atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t idx = (int64_t)s_lb * kBlock + threadIdx.x;
    if (idx >= n) return;                            // (lanes past the end leave before the loop: exec-masked for good)
    const int64_t i = UPPER ? n - 1 - idx : idx;
    const int s = rp[i], e = rp[i + 1];
    int k = UPPER ? s + 1 : s;                       // the diagonal is first (upper) or last (lower)
    const int ke = UPPER ? e : e - 1;
    int lvl = 0;
    int c = k < ke ? ci[k] : 0;
    unsigned spins = 0;
    bool stored = false;
    const int64_t my_strip = map.rows > 0 ? map.strip_of(idx) : 0;
    auto foreign = [&](int col) {                     // a dependency that lives in another strip does not count
        if (map.rows <= 0) return false;
        const int64_t cidx = UPPER ? n - 1 - col : col;
        return map.strip_of(cidx) != my_strip;
    };
    // The loop is left by the whole wave at once (ballot): were lanes to leave one by one, the compiler could move the
    // store onto the exit path, where a SIMT machine executes it only after EVERY lane has left -- a lane waiting for
    // the level of a row owned by another lane of its own wave would then wait forever.
    for (;;) {
        if (!stored && k < ke) {
            const int l = foreign(c) ? 0x7fffffff : __hip_atomic_load(level + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (l == 0x7fffffff) {
                ++k;
                if (k < ke) c = ci[k];
            } else if (l >= 0) {
                lvl = l + 1 > lvl ? l + 1 : lvl;
                ++k;
                if (k < ke) c = ci[k];
                spins = 0;
            } else if (++spins > (1u << 24)) {       // bounded: a malformed factor must not hang the device
                atomicExch(err, 1);
                k = ke;
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (!stored && k >= ke) {
            __hip_atomic_store(level + i, lvl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stored = true;
        }
        if (__ballot(!stored) == 0) break;
    }
}

void launch_levels_syncfree(int64_t n, const int32_t *rp, const int32_t *ci, bool upper, int32_t *level,
                            unsigned int *ticket_zeroed, int *err, hipStream_t s, StripMap map) {
    const int grid = (int)((n + kBlock - 1) / kBlock);
    if (upper)
        hipLaunchKernelGGL(k_levels_syncfree<true>, dim3(grid), dim3(kBlock), 0, s, n, rp, ci, level, ticket_zeroed, err, map);
    else
        hipLaunchKernelGGL(k_levels_syncfree<false>, dim3(grid), dim3(kBlock), 0, s, n, rp, ci, level, ticket_zeroed, err, map);
}

// ------------------------------------------------------------------------------------------------
// Strip-pipelined triangular solve: setup pieces (see k_sptrsv_strips in dpcg_sptrsv.hip)
// ------------------------------------------------------------------------------------------------
// key[i] = strip(i) * nlev + local level(i)
__global__ __launch_bounds__(kBlock) void k_strip_keys(int64_t n, const int32_t *__restrict__ level, StripMap map, int nlev,
                                                       int upper, uint32_t *__restrict__ key) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int64_t idx = upper ? n - 1 - i : i;
        key[i] = (uint32_t)(map.strip_of(idx) * nlev + level[i]);
    }
}

void launch_strip_keys(int64_t n, const int32_t *level, StripMap map, int nlev, bool upper, uint32_t *key, hipStream_t s) {
    hipLaunchKernelGGL(k_strip_keys, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, level, map, nlev, upper ? 1 : 0, key);
}

// Records of the strip solve, one per position j of the (strip, level, row) order:
//   meta[j] = {d0, d1, d2, row};  d >= 0: position of an entry of the SAME strip (read from the LDS ring);
//             d == -1: no such entry;  d <= -2: an entry of an EARLIER strip, column -2 - d (polled in `out`);
//             d0 == INT_MIN: more than three off-diagonal entries, the row walks lo_rowptr;
//   val[4j..4j+3] = {v0, v1, v2, diagonal}.
// stats[0] = max reach (j - position of an own-strip entry), stats[1] = entries of earlier strips, stats[2] = long rows,
// stats[3] = entries that live in a LATER strip (the plan is then not usable: strips are handed out in order, and a strip
// may only wait for strips that have started).
__global__ __launch_bounds__(kBlock) void k_strip_records(int64_t n, const uint32_t *__restrict__ key_of_pos, int nlev,
                                                          const int32_t *__restrict__ level_ptr,
                                                          const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ lo_rp, const int32_t *__restrict__ lo_ci,
                                                          const int32_t *__restrict__ lo_cp, const double *__restrict__ lo_v,
                                                          int upper, int ring_reach, int32_t *__restrict__ meta,
                                                          double *__restrict__ pv, int32_t *exported, int *stats) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int reach = 0, ext = 0, longrows = 0, later = 0;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        int m[4] = {-1, -1, -1, -1};
        double w[4] = {0.0, 0.0, 0.0, 0.0};
        const int strip = (int)(key_of_pos[j] / (uint32_t)nlev);
        const int start = level_ptr[strip * nlev], end = level_ptr[(strip + 1) * nlev];
        const int a = lo_rp[j], b = lo_rp[j + 1];
        const int ks = upper ? a + 1 : a, ke = upper ? b : b - 1;
        m[3] = rows[j];
        w[3] = lo_v[upper ? a : b - 1];
        // through the ring: an entry of the own strip at most ring_reach positions back; everything else -- entries of
        // earlier strips, and own entries too far back for the ring -- is read from `out`, whose owner must then publish it
        for (int k = ks; k < ke; ++k) {
            const int cp = lo_cp[k];
            const bool near = cp >= start && cp < end && (int)j - cp <= ring_reach;
            if (near) {
                reach = (int)j - cp > reach ? (int)j - cp : reach;
            } else {
                ++ext;
                exported[lo_ci[k]] = 1;
                if ((int)(key_of_pos[cp] / (uint32_t)nlev) > strip) ++later;
            }
            if (ke - ks <= 3) {
                m[k - ks] = near ? cp : -2 - lo_ci[k];
                w[k - ks] = lo_v[k];
            }
        }
        if (ke - ks > 3) {
            m[0] = (int)0x80000000;
            ++longrows;
        }
        reinterpret_cast<int4 *>(meta)[j] = make_int4(m[0], m[1], m[2], m[3]);
        reinterpret_cast<double2 *>(pv)[2 * j] = make_double2(w[0], w[1]);
        reinterpret_cast<double2 *>(pv)[2 * j + 1] = make_double2(w[2], w[3]);
    }
    if (reach) atomicMax(stats, reach);
    if (ext) atomicAdd(stats + 1, ext);
    if (longrows) atomicAdd(stats + 2, longrows);
    if (later) atomicAdd(stats + 3, later);
}

// rows some other row reads from `out` during the launch get bit 30 of their own-row field set: they are stored write-through
__global__ __launch_bounds__(kBlock) void k_strip_mark_exported(int64_t n, const int32_t *__restrict__ exported, int32_t *meta) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const int row = meta[4 * j + 3];
        if (exported[row]) meta[4 * j + 3] = row | (1 << 30);
    }
}

// out[0] = max over the rows of (row - smallest column) for a lower factor, (largest column - row) for an upper one: the band;
// out[1] = max over the rows of the SECOND largest such distance: for a 3-D grid the length of a grid line (the band being a
// plane), 1 for a 2-D grid -- what the parts of a two-way strip cut are aligned to.
__global__ __launch_bounds__(kBlock) void k_max_band(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                     int upper, int *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int m = 0, m2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int a = rp[i], b = rp[i + 1];
        if (b > a) {
            const int d = upper ? ci[b - 1] - (int)i : (int)i - ci[a];
            m = d > m ? d : m;
        }
        if (b - a > 2) {                                  // (the diagonal is one of the entries)
            const int d2 = upper ? ci[b - 2] - (int)i : (int)i - ci[a + 1];
            m2 = d2 > m2 ? d2 : m2;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_down(m, off), o2 = __shfl_down(m2, off);
        m = o > m ? o : m;
        m2 = o2 > m2 ? o2 : m2;
    }
    if ((threadIdx.x & 63) == 0) {
        if (m > 0) atomicMax(out, m);
        if (m2 > 0) atomicMax(out + 1, m2);
    }
}

// Can IC(0) on this lower pattern (diagonal last in a row) run as a recurrence on the diagonals alone?  flags |= 1: two
// off-diagonal entries (i, a), (i, b), a < b, of some row are joined by an entry (b, a) -- a cross term L_ia * L_ba in
// the update of L_ib; flags |= 2: a row with more than three off-diagonal entries (longer than a strip record).
__global__ __launch_bounds__(kBlock) void k_ic0_cross_terms(int64_t n, const int32_t *__restrict__ rp,
                                                            const int32_t *__restrict__ ci, int *flags) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int a = rp[i], b = rp[i + 1] - 1;               // off-diagonals [a, b)
        if (b - a > 3) f |= 2;
        for (int q = a + 1; q < b; ++q) {
            const int cq = ci[q];
            const int sa = rp[cq], sb = rp[cq + 1] - 1;
            for (int p = a; p < q; ++p) {
                const int cp = ci[p];
                for (int k = sa; k < sb; ++k)
                    if (ci[k] == cp) f |= 1;
            }
        }
    }
    if (f) atomicOr(flags, f);
}

void launch_ic0_cross_terms(int64_t n, const int32_t *rp, const int32_t *ci, int *flags_zeroed, hipStream_t s) {
    hipLaunchKernelGGL(k_ic0_cross_terms, dim3(grid_rows(n, 1024)), dim3(kBlock), 0, s, n, rp, ci, flags_zeroed);
}

// Descriptors of the general factorisation through the ring walk (k_sptrsv_ring_pipe, FACTOR = 2), per level-order position j
// of a lower pattern whose rows hold at most three off-diagonal entries (lo_*: the level-ordered copy, diagonal last):
//   xdesc[j]: 2 bits for each of the entry pairs (p, q) = (0,1), (0,2), (1,2): 1 + the slot of column c_p among the
//             off-diagonal entries of row c_q, 0 when row c_q has no such entry (no cross term);
//   thr[4j + q] = tau * colnorm[c_q] (the drop threshold of entry q; thr == nullptr: nothing is dropped).
__global__ __launch_bounds__(kBlock) void k_ring_factor_desc(int64_t n, const int32_t *__restrict__ lo_rp,
                                                             const int32_t *__restrict__ lo_ci, const int32_t *__restrict__ lo_cp,
                                                             const double *__restrict__ colnorm, double tau,
                                                             int32_t *__restrict__ xdesc, double *__restrict__ thr) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const int a = lo_rp[j], b = lo_rp[j + 1] - 1;                 // off-diagonals [a, b)
        const int m = b - a < 3 ? b - a : 3;
        int desc = 0;
        for (int q = 1; q < m; ++q) {
            const int pq = lo_cp[a + q];                               // position of row c_q
            const int sa = lo_rp[pq], sb = lo_rp[pq + 1] - 1;
            for (int p = 0; p < q; ++p) {
                const int cp = lo_ci[a + p];
                int slot = 0;
                for (int k = sa; k < sb && k < sa + 3; ++k)
                    if (lo_ci[k] == cp) slot = 1 + (k - sa);
                const int pair = q == 1 ? 0 : 1 + p;                   // (0,1) -> 0, (0,2) -> 1, (1,2) -> 2
                desc |= slot << (2 * pair);
            }
        }
        xdesc[j] = desc;
        if (thr) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            for (int q = 0; q < m; ++q) t[q] = tau * colnorm[lo_ci[a + q]];
            reinterpret_cast<double2 *>(thr)[2 * j] = make_double2(t[0], t[1]);
            reinterpret_cast<double2 *>(thr)[2 * j + 1] = make_double2(t[2], t[3]);
        }
    }
}

void launch_ring_factor_desc(int64_t n, const int32_t *lo_rp, const int32_t *lo_ci, const int32_t *lo_cp, const double *colnorm,
                             double tau, int32_t *xdesc, double *thr, hipStream_t s) {
    hipLaunchKernelGGL(k_ring_factor_desc, dim3(grid_rows(n, 1024)), dim3(kBlock), 0, s, n, lo_rp, lo_ci, lo_cp, colnorm, tau, xdesc,
                       colnorm ? thr : nullptr);
}

// flag[0] = 1 when some off-diagonal entry sits FURTHER along the band than its row (column mod band > row mod band for a lower
// factor, < for an upper one): in a two-way strip cut it would live in a later part of an earlier slab -- e.g. the level-1 fill
// entry (i, i - nx + 1) of a 5-point grid.  Such a pattern takes slabs straight away.
__global__ __launch_bounds__(kBlock) void k_points_along_band(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                              int upper, int band, int *flag) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int im = (int)(i % band);
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
            const int c = ci[k];
            if (c == (int)i) continue;
            const int cm = c % band;
            if (upper ? cm < im && c - (int)i >= band - im : cm > im && (int)i - c >= im + 1) f = 1;
        }
    }
    if (f) atomicExch(flag, 1);
}

void launch_points_along_band(int64_t n, const int32_t *rp, const int32_t *ci, bool upper, int band, int *flag_zeroed, hipStream_t s) {
    hipLaunchKernelGGL(k_points_along_band, dim3(grid_rows(n, 1024)), dim3(kBlock), 0, s, n, rp, ci, upper ? 1 : 0, band, flag_zeroed);
}

void launch_max_band(int64_t n, const int32_t *rp, const int32_t *ci, bool upper, int *out_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_max_band, dim3(grid_rows(n, 1024)), dim3(kBlock), 0, s, n, rp, ci, upper ? 1 : 0, out_dev);
}

void launch_strip_records(int64_t n, const uint32_t *key_of_pos, int nlev, const int32_t *level_ptr, const int32_t *rows,
                          const int32_t *lo_rp, const int32_t *lo_ci, const int32_t *lo_cp, const double *lo_v, bool upper,
                          int ring_reach, int32_t *meta, double *pv, int32_t *exported_zeroed, int *stats, hipStream_t s) {
    hipLaunchKernelGGL(k_strip_records, dim3(grid_rows(n, 1024)), dim3(kBlock), 0, s, n, key_of_pos, nlev, level_ptr, rows, lo_rp,
                       lo_ci, lo_cp, lo_v, upper ? 1 : 0, ring_reach, meta, pv, exported_zeroed, stats);
    hipLaunchKernelGGL(k_strip_mark_exported, dim3(grid_rows(n, 1024)), dim3(kBlock), 0, s, n, exported_zeroed, meta);
}

// ------------------------------------------------------------------------------------------------
// level-ordered copy of the factor: row j of the copy = original row rows[j]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_lo_lengths(int64_t n, const int32_t *__restrict__ rows,
                                                       const int32_t *__restrict__ rp, int32_t *__restrict__ len,
                                                       int32_t *__restrict__ pos) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j <= n; j += stride) {
        if (j == n) {
            len[j] = 0;
            continue;
        }
        const int i = rows[j];
        len[j] = rp[i + 1] - rp[i];
        pos[i] = (int32_t)j;                          // level-order position of original row i
    }
}

void launch_lo_lengths(int64_t n, const int32_t *rows, const int32_t *rp, int32_t *len, int32_t *pos, hipStream_t s) {
    hipLaunchKernelGGL(k_lo_lengths, dim3(grid_rows(n + 1)), dim3(kBlock), 0, s, n, rows, rp, len, pos);
}

// (a workgroup copies the entries of kBlock consecutive level-ordered rows: they are contiguous in the copy, so its lanes walk them
// side by side -- coalesced stores -- and find an entry's row by bisection of the rows' offsets in LDS; one thread per row with a
// loop over its entries took 1.8 ms for the 67M entries of a 256^3 factor)
__global__ __launch_bounds__(kBlock) void k_lo_copy(int64_t n, const int32_t *__restrict__ rows,
                                                    const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                    const double *__restrict__ v, const int32_t *__restrict__ pos,
                                                    const int32_t *__restrict__ lo_rp, int32_t *__restrict__ lo_ci,
                                                    int32_t *__restrict__ lo_cp, double *__restrict__ lo_v) {
    __shared__ int s_dst[kBlock + 1], s_src[kBlock];
    const int tid = threadIdx.x;
    for (int64_t j0 = (int64_t)blockIdx.x * kBlock; j0 < n; j0 += (int64_t)gridDim.x * kBlock) {
        const int nrows = (int)(n - j0 < kBlock ? n - j0 : kBlock);
        if (tid < nrows) {
            s_src[tid] = rp[rows[j0 + tid]];
            s_dst[tid] = lo_rp[j0 + tid];
        }
        if (tid == 0) s_dst[nrows] = lo_rp[j0 + nrows];
        __syncthreads();
        const int e1 = s_dst[nrows];
        for (int e = s_dst[0] + tid; e < e1; e += kBlock) {
            int lo = 0, hi = nrows - 1;                       // the last row that starts at or before e (empty rows come before it)
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_dst[mid] <= e) lo = mid;
                else hi = mid - 1;
            }
            const int src = s_src[lo] + (e - s_dst[lo]);
            const int c = ci[src];
            lo_ci[e] = c;
            lo_cp[e] = pos[c];
            lo_v[e] = v[src];
        }
        __syncthreads();
    }
}

void launch_lo_copy(int64_t n, const int32_t *rows, const int32_t *rp, const int32_t *ci, const double *v,
                    const int32_t *pos, const int32_t *lo_rp, int32_t *lo_ci, int32_t *lo_cp, double *lo_v,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_lo_copy, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows, rp, ci, v, pos, lo_rp, lo_ci, lo_cp,
                       lo_v);
}

// Does every 256-row block of every level fit the LDS product buffer of the stream kernels?  *flag = 1 if not.
__global__ __launch_bounds__(kBlock) void k_stream_fit(int64_t n, const uint32_t *__restrict__ lvl_of_pos,
                                                       const int32_t *__restrict__ level_ptr,
                                                       const int32_t *__restrict__ lo_rp, int *flag) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const int l = (int)lvl_of_pos[j];
        const int lo = level_ptr[l], hi = level_ptr[l + 1];
        if (((int)j - lo) % kStreamRows != 0) continue;
        const int je = (int)j + kStreamRows < hi ? (int)j + kStreamRows : hi;
        if (lo_rp[je] - lo_rp[j] > kStreamCap) atomicExch(flag, 1);
    }
}

void launch_stream_fit(int64_t n, const uint32_t *lvl_of_pos, const int32_t *level_ptr, const int32_t *lo_rp, int *flag,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_stream_fit, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, lvl_of_pos, level_ptr, lo_rp, flag);
}

// For the rows of merged segments: how far back (in level order, within the segment) does a row reach?
// seg_of_level[l] = index of the merged segment level l belongs to, or -1; maxdist[seg] receives the maximum.
__global__ __launch_bounds__(kBlock) void k_ring_reach(int64_t n, const uint32_t *__restrict__ lvl_of_pos,
                                                       const int32_t *__restrict__ seg_of_level,
                                                       const int32_t *__restrict__ seg_start,
                                                       const int32_t *__restrict__ lo_rp,
                                                       const int32_t *__restrict__ lo_cp, int32_t *maxdist) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t first = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (int64_t j0 = first - threadIdx.x % 64; j0 < n; j0 += stride) {     // whole waves stay in the loop together
        const int64_t j = j0 + threadIdx.x % 64;
        int sg = -1, d = 0;
        if (j < n) {
            sg = seg_of_level[lvl_of_pos[j]];
            if (sg >= 0) {
                const int start = seg_start[sg];
                for (int k = lo_rp[j]; k < lo_rp[j + 1]; ++k) {
                    const int cp = lo_cp[k];
                    if (cp >= start && cp < j) d = (int)j - cp > d ? (int)j - cp : d;
                }
            }
        }
        // positions of a segment are contiguous, so a wave almost always sits inside one segment: one atomic per wave
        // (a million atomics on the single counter of a 2-D factor's one merged segment took 12 ms)
        const int sg0 = __shfl(sg, 0);
        if (__ballot(sg != sg0) == 0) {
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_down(d, off);
                d = o > d ? o : d;
            }
            if (threadIdx.x % 64 == 0 && sg0 >= 0 && d > 0) atomicMax(maxdist + sg0, d);
        } else if (sg >= 0 && d > 0) {
            atomicMax(maxdist + sg, d);
        }
    }
}

void launch_ring_reach(int64_t n, const uint32_t *lvl_of_pos, const int32_t *seg_of_level, const int32_t *seg_start,
                       const int32_t *lo_rp, const int32_t *lo_cp, int32_t *maxdist, hipStream_t s) {
    hipLaunchKernelGGL(k_ring_reach, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, lvl_of_pos, seg_of_level, seg_start, lo_rp,
                       lo_cp, maxdist);
}

// Fixed-width records of the rows of LDS-ring segments (Levels::pk_meta / pk_val).  ring_start_of_level[l] = first
// level-order position of the ring segment level l belongs to, or -1 when the level is not in one.
__global__ __launch_bounds__(kBlock) void k_ring_records(int64_t n, const uint32_t *__restrict__ lvl_of_pos,
                                                         const int32_t *__restrict__ ring_start_of_level,
                                                         const int32_t *__restrict__ rows,
                                                         const int32_t *__restrict__ lo_rp,
                                                         const int32_t *__restrict__ lo_ci,
                                                         const int32_t *__restrict__ lo_cp,
                                                         const double *__restrict__ lo_v, int32_t *__restrict__ meta,
                                                         double *__restrict__ pv) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        int m[4] = {-1, -1, -1, -1};
        double w[4] = {0.0, 0.0, 0.0, 0.0};
        const int start = ring_start_of_level[lvl_of_pos[j]];
        if (start >= 0) {
            const int a = lo_rp[j], b = lo_rp[j + 1], row = rows[j];
            // the diagonal is the first entry of a row of L^T and the last of a row of L
            const bool diag_first = lo_ci[a] == row && (b - a == 1 || lo_ci[b - 1] != row);
            const int ks = diag_first ? a + 1 : a, ke = diag_first ? b : b - 1;
            m[3] = row;
            w[3] = lo_v[diag_first ? a : b - 1];
            bool fast = ke - ks <= 3;
            for (int k = ks; k < ke && fast; ++k) fast = lo_cp[k] >= start;
            if (!fast) {
                m[0] = -2;
            } else {
                for (int k = ks; k < ke; ++k) {
                    m[k - ks] = lo_cp[k];
                    w[k - ks] = lo_v[k];
                }
            }
        }
        reinterpret_cast<int4 *>(meta)[j] = make_int4(m[0], m[1], m[2], m[3]);
        reinterpret_cast<double2 *>(pv)[2 * j] = make_double2(w[0], w[1]);
        reinterpret_cast<double2 *>(pv)[2 * j + 1] = make_double2(w[2], w[3]);
    }
}

void launch_ring_records(int64_t n, const uint32_t *lvl_of_pos, const int32_t *ring_start_of_level, const int32_t *rows,
                         const int32_t *lo_rp, const int32_t *lo_ci, const int32_t *lo_cp, const double *lo_v,
                         int32_t *meta, double *pv, hipStream_t s) {
    hipLaunchKernelGGL(k_ring_records, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, lvl_of_pos, ring_start_of_level, rows,
                       lo_rp, lo_ci, lo_cp, lo_v, meta, pv);
}

// Fixed-width records for the sync-free kernel (Levels::sf_meta / sf_val), one per level-order position j:
//   sf_meta[j] = {c0, c1, c2, row}: the (relabelled) column indices of the first three off-diagonal entries (-1 = no
//                such entry; c0 = -2: more than three, the row walks lo_rowptr instead) and its own row index;
//   sf_val[4j..4j+3] = {v0, v1, v2, diagonal}.
// Factors with many longer rows (an unstructured mesh in its own numbering: up to 6 lower neighbours on a 7-point graph) take
// the width-6 form (Levels::rec_w): 8 ints {c0..c5, row, -} and 8 doubles {v0..v5, diagonal, -} per row; factors with fill
// (ICT), or of wider stencils (27-point: 13 lower entries), the width-14 form, 16 + 16.
// Addressed by position alone, so a row's data can be requested before anything about the row is known.
template <int W>   // W = 3 / 6 / 14 entries; a record is S = 4 / 8 / 16 ints {c0..c(W-1), row, -...} and S doubles {v0..v(W-1), diag, -...}
__global__ __launch_bounds__(kBlock) void k_sf_records(int64_t n, const int32_t *__restrict__ rows,
                                                       const int32_t *__restrict__ lo_rp, const int32_t *__restrict__ lo_ci,
                                                       const double *__restrict__ lo_v, int upper, int32_t *__restrict__ meta,
                                                       double *__restrict__ pv) {
    constexpr int S = W == 3 ? 4 : (W == 6 ? 8 : 16);                  // record stride (ints / doubles)
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        int m[S];
        double w[S];
        for (int q = 0; q < S; ++q) {
            m[q] = -1;
            w[q] = 0.0;
        }
        const int a = lo_rp[j], b = lo_rp[j + 1];
        const int ks = upper ? a + 1 : a, ke = upper ? b : b - 1;      // the diagonal is first (L^T) or last (L)
        m[W] = rows ? rows[j] : (int)j;                              // rows == null: level-major records (own position)
        w[W] = lo_v[upper ? a : b - 1];
        if (ke - ks > W) {
            m[0] = -2;
        } else {
            for (int k = ks; k < ke; ++k) {
                m[k - ks] = lo_ci[k];
                w[k - ks] = lo_v[k];
            }
        }
        for (int q = 0; q < S; q += 4)
            reinterpret_cast<int4 *>(meta)[j * (S / 4) + q / 4] = make_int4(m[q], m[q + 1], m[q + 2], m[q + 3]);
        for (int q = 0; q < S; q += 2) reinterpret_cast<double2 *>(pv)[j * (S / 2) + q / 2] = make_double2(w[q], w[q + 1]);
    }
}

__global__ __launch_bounds__(kBlock) void k_invert_positions(int64_t n, const int32_t *__restrict__ rows, int32_t *__restrict__ pos) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) pos[rows[j]] = (int32_t)j;
}

void launch_invert_positions(int64_t n, const int32_t *rows, int32_t *pos, hipStream_t s) {
    hipLaunchKernelGGL(k_invert_positions, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows, pos);
}

__global__ __launch_bounds__(kBlock) void k_compose_positions(int64_t n, const int32_t *__restrict__ rows,
                                                              const int32_t *__restrict__ pos, int32_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) out[j] = pos[rows[j]];
}

void launch_compose_positions(int64_t n, const int32_t *rows, const int32_t *pos, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_compose_positions, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows, pos, out);
}

void launch_sf_records(int64_t n, const int32_t *rows, const int32_t *lo_rp, const int32_t *lo_ci, const double *lo_v,
                       bool upper, int32_t *meta, double *pv, int width, hipStream_t s) {
    if (width == 14)
        hipLaunchKernelGGL(k_sf_records<14>, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows, lo_rp, lo_ci, lo_v, upper ? 1 : 0, meta, pv);
    else if (width == 6)
        hipLaunchKernelGGL(k_sf_records<6>, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows, lo_rp, lo_ci, lo_v, upper ? 1 : 0, meta, pv);
    else
        hipLaunchKernelGGL(k_sf_records<3>, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows, lo_rp, lo_ci, lo_v, upper ? 1 : 0, meta, pv);
}

// *counter += rows of the level-ordered copy with more than `limit` off-diagonal entries (one atomic per wave)
__global__ __launch_bounds__(kBlock) void k_count_long_rows(int64_t n, const int32_t *__restrict__ lo_rp, int limit, int *counter) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int c = 0;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) c += (lo_rp[j + 1] - lo_rp[j] - 1 > limit) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(counter, c);
}

void launch_count_long_rows(int64_t n, const int32_t *lo_rp, int limit, int *counter, hipStream_t s) {
    hipLaunchKernelGGL(k_count_long_rows, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, lo_rp, limit, counter);
}

// ------------------------------------------------------------------------------------------------
// tril(A) for IC(0): count, then copy, entries with col <= row; *flag = 1 when a row's last kept entry is not its diagonal
// ------------------------------------------------------------------------------------------------
// (both kernels: a workgroup takes kBlock consecutive rows, whose entries are contiguous, and its lanes walk the entries side by
// side, finding an entry's row by bisection of the row offsets in LDS -- coalesced loads and stores; a thread per row with a loop
// over its entries: 1.6 ms for the 117M entries of a 256^3 system)
__device__ __forceinline__ int row_of_entry(const int *s_off, int nrows, int e) {     // the last row that starts at or before e
    int lo = 0, hi = nrows - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_off[mid] <= e) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(kBlock) void k_tril_count(int64_t n, const int32_t *__restrict__ rp,
                                                       const int32_t *__restrict__ ci, int32_t *__restrict__ cnt,
                                                       int *flag) {
    __shared__ int s_rp[kBlock + 1], s_cnt[kBlock], s_diag[kBlock];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) cnt[n] = 0;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < n; i0 += (int64_t)gridDim.x * kBlock) {
        const int nrows = (int)(n - i0 < kBlock ? n - i0 : kBlock);
        if (tid < nrows) s_rp[tid] = rp[i0 + tid];
        if (tid == 0) s_rp[nrows] = rp[i0 + nrows];
        s_cnt[tid] = 0;
        s_diag[tid] = 0;
        __syncthreads();
        const int e1 = s_rp[nrows];
        for (int e = s_rp[0] + tid; e < e1; e += kBlock) {
            const int r = row_of_entry(s_rp, nrows, e);
            const int c = ci[e], i = (int)i0 + r;
            if (c <= i) atomicAdd(&s_cnt[r], 1);
            if (c == i) s_diag[r] = 1;
        }
        __syncthreads();
        if (tid < nrows) {
            cnt[i0 + tid] = s_cnt[tid];
            if (!s_diag[tid]) atomicExch(flag, 1);            // (columns ascend: the last kept entry is the diagonal iff it is there)
        }
        __syncthreads();
    }
}

void launch_tril_count(int64_t n, const int32_t *rp, const int32_t *ci, int32_t *cnt, int *flag, hipStream_t s) {
    hipLaunchKernelGGL(k_tril_count, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rp, ci, cnt, flag);
}

__global__ __launch_bounds__(kBlock) void k_tril_copy(int64_t n, const int32_t *__restrict__ rp,
                                                      const int32_t *__restrict__ ci, const double *__restrict__ v,
                                                      const int32_t *__restrict__ lrp, int32_t *__restrict__ lci,
                                                      double *__restrict__ lv) {
    __shared__ int s_dst[kBlock + 1], s_src[kBlock];
    const int tid = threadIdx.x;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock; i0 < n; i0 += (int64_t)gridDim.x * kBlock) {
        const int nrows = (int)(n - i0 < kBlock ? n - i0 : kBlock);
        if (tid < nrows) {
            s_src[tid] = rp[i0 + tid];
            s_dst[tid] = lrp[i0 + tid];
        }
        if (tid == 0) s_dst[nrows] = lrp[i0 + nrows];
        __syncthreads();
        const int e1 = s_dst[nrows];
        for (int e = s_dst[0] + tid; e < e1; e += kBlock) {     // (columns ascend: the kept entries of a row are its first ones)
            const int r = row_of_entry(s_dst, nrows, e);
            const int src = s_src[r] + (e - s_dst[r]);
            lci[e] = ci[src];
            lv[e] = v[src];
        }
        __syncthreads();
    }
}

// ---- value-only refresh of a factor whose pattern is kept (dpcg_precond.hip: refresh_parked_ic0) -------------------------------
__global__ __launch_bounds__(kBlock) void k_iota_f64(int64_t count, double *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < count; k += stride) out[k] = (double)k;
}
void launch_iota_f64(int64_t count, double *out, hipStream_t s) {
    hipLaunchKernelGGL(k_iota_f64, dim3(grid_rows(count)), dim3(kBlock), 0, s, count, out);
}
__global__ __launch_bounds__(kBlock) void k_f64_to_i32(int64_t count, const double *__restrict__ in, int32_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < count; k += stride) out[k] = (int32_t)in[k];
}
void launch_f64_to_i32(int64_t count, const double *in, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_f64_to_i32, dim3(grid_rows(count)), dim3(kBlock), 0, s, count, in, out);
}
// the values of the level-ordered copy again: position j holds factor row rows[j], entries in the row's own order (k_lo_copy)
__global__ __launch_bounds__(kBlock) void k_lo_values(int64_t n, const int32_t *__restrict__ rows, const int32_t *__restrict__ rp,
                                                      const double *__restrict__ v, const int32_t *__restrict__ lo_rp,
                                                      double *__restrict__ lo_v) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        const int i = rows[j];
        const int src = rp[i], len = rp[i + 1] - src, dst = lo_rp[j];
        for (int k = 0; k < len; ++k) lo_v[dst + k] = v[src + k];
    }
}
void launch_lo_values(int64_t n, const int32_t *rows, const int32_t *rp, const double *v, const int32_t *lo_rp, double *lo_v,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_lo_values, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rows, rp, v, lo_rp, lo_v);
}

void launch_tril_copy(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, const int32_t *lrp, int32_t *lci,
                      double *lv, hipStream_t s) {
    hipLaunchKernelGGL(k_tril_copy, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rp, ci, v, lrp, lci, lv);
}

// ------------------------------------------------------------------------------------------------
// ICT: thresholded incomplete Cholesky with level-1 fill (contract: oracle/oracle.py::ict; stands in for
// ilupp.icholt(add_fill_in=1, threshold=0.1), the reference harness's default technique, test.py:81-88)
// ------------------------------------------------------------------------------------------------
// c[j] = || A(j:n, j) ||_1 = sum of |a_jk| over the entries of row j with column >= j (A symmetric)
__global__ __launch_bounds__(kBlock) void k_colnorm1(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                     const double *__restrict__ v, double *__restrict__ c) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += stride) {
        double sum = 0.0;
        for (int k = rp[j]; k < rp[j + 1]; ++k)
            if (ci[k] >= j) sum += fabs(v[k]);
        c[j] = sum;
    }
}

void launch_colnorm1(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, double *c, hipStream_t s) {
    hipLaunchKernelGGL(k_colnorm1, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rp, ci, v, c);
}

// Pattern of row i with level-1 fill: tril(A)_i plus every j < i that shares a neighbour k < j with i.  One thread per
// row keeps the sorted set in a private array (rows are short; a row whose set would exceed kIctRowCap keeps tril(A)_i
// only and raises *overflow).  WRITE = false: cnt[i] = size; WRITE = true: entries to lci / lv (a_ij, or 0 for fill).
constexpr int kIctRowCap = 192;
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_ict_pattern(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                        const double *__restrict__ v, int fill, int32_t *__restrict__ cnt,
                                                        const int32_t *__restrict__ lrp, int32_t *__restrict__ lci,
                                                        double *__restrict__ lv, int *flags) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= n; i += stride) {
        if (i == n) {
            if (!WRITE) cnt[i] = 0;
            continue;
        }
        int set[kIctRowCap];
        int m = 0, last = -1;
        bool over = false;
        for (int k = rp[i]; k < rp[i + 1]; ++k)
            if (ci[k] <= i) {
                if (m < kIctRowCap) set[m++] = ci[k];
                else over = true;
                last = ci[k];
            }
        const int base = m;                                  // tril(A)_i, ascending; re-read from A below, `set` grows
        if (last != i) atomicOr(flags, 1);                   // missing diagonal
        if (over) atomicOr(flags, 4);                        // tril(A)_i itself does not fit the private set: refused
        if (fill && !over) {
            for (int p = rp[i]; p < rp[i + 1] && !over; ++p) {
                const int k = ci[p];                         // original entries only: level-1 fill
                if (k >= i) continue;
                for (int q = rp[k]; q < rp[k + 1]; ++q) {
                    const int j = ci[q];
                    if (j <= k || j >= i) continue;
                    int lo = 0;
                    while (lo < m && set[lo] < j) ++lo;
                    if (lo < m && set[lo] == j) continue;
                    if (m >= kIctRowCap) {
                        over = true;
                        break;
                    }
                    for (int t = m; t > lo; --t) set[t] = set[t - 1];
                    set[lo] = j;
                    ++m;
                }
            }
            if (over) {                                      // fall back to tril(A)_i for this row
                atomicOr(flags, 2);
                m = 0;
                for (int k = rp[i]; k < rp[i + 1]; ++k)
                    if (ci[k] <= i && m < kIctRowCap) set[m++] = ci[k];
            }
        }
        (void)base;
        if (!WRITE) {
            cnt[i] = m;
        } else {
            const int d = lrp[i];
            int q = rp[i];
            for (int t = 0; t < m; ++t) {
                while (q < rp[i + 1] && ci[q] < set[t]) ++q;
                lci[d + t] = set[t];
                lv[d + t] = (q < rp[i + 1] && ci[q] == set[t]) ? v[q] : 0.0;
            }
        }
    }
}

void launch_ict_pattern(bool write, int64_t n, const int32_t *rp, const int32_t *ci, const double *v, int fill, int32_t *cnt,
                        const int32_t *lrp, int32_t *lci, double *lv, int *flags, hipStream_t s) {
    if (write)
        hipLaunchKernelGGL(k_ict_pattern<true>, dim3(grid_rows(n + 1)), dim3(kBlock), 0, s, n, rp, ci, v, fill, cnt, lrp, lci, lv, flags);
    else
        hipLaunchKernelGGL(k_ict_pattern<false>, dim3(grid_rows(n + 1)), dim3(kBlock), 0, s, n, rp, ci, v, fill, cnt, lrp, lci, lv, flags);
}

// compaction after the numeric phase: dropped entries are stored zeros; the diagonal always stays
__global__ __launch_bounds__(kBlock) void k_count_kept(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                       const double *__restrict__ v, int32_t *__restrict__ cnt) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= n; i += stride) {
        int c = 0;
        if (i < n)
            for (int k = rp[i]; k < rp[i + 1]; ++k) c += (v[k] != 0.0 || ci[k] == i) ? 1 : 0;
        cnt[i] = c;
    }
}

__global__ __launch_bounds__(kBlock) void k_copy_kept(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                      const double *__restrict__ v, const int32_t *__restrict__ orp,
                                                      int32_t *__restrict__ oci, double *__restrict__ ov) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        int d = orp[i];
        for (int k = rp[i]; k < rp[i + 1]; ++k)
            if (v[k] != 0.0 || ci[k] == i) {
                oci[d] = ci[k];
                ov[d] = v[k];
                ++d;
            }
    }
}

void launch_count_kept(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, int32_t *cnt, hipStream_t s) {
    hipLaunchKernelGGL(k_count_kept, dim3(grid_rows(n + 1)), dim3(kBlock), 0, s, n, rp, ci, v, cnt);
}
void launch_copy_kept(int64_t n, const int32_t *rp, const int32_t *ci, const double *v, const int32_t *orp, int32_t *oci,
                      double *ov, hipStream_t s) {
    hipLaunchKernelGGL(k_copy_kept, dim3(grid_rows(n)), dim3(kBlock), 0, s, n, rp, ci, v, orp, oci, ov);
}

}  // namespace dpcg
