// Bandwidth-reducing symmetric reordering of the system matrix, computed on the device: P A P^T with P the reverse
// Cuthill-McKee order.  A pressure matrix whose numbering scatters neighbours (an OpenFOAM mesh after refinement, the
// scrambled stand-in of BASELINE config 3) makes every x[col] of the SpMV its own 128-byte line; in RCM order the
// columns of a 256-row block fall into a few runs again, so the x-tile kernel applies.
//
//   1. breadth-first level structure from a pseudo-peripheral vertex (George-Liu: repeat the search from a
//      minimum-degree vertex of the last level while the eccentricity grows), frontier by frontier, the frontier
//      sizes and offsets living on the device (no host round trip per level);
//   2. Cuthill-McKee positions inside the level structure, level by level: a vertex is keyed by the position of its
//      first-numbered neighbour in the previous level, vertices with the same key are ranked by (degree, index);
//      three small launches per level and no sort: children are counted per parent, the counts scanned, siblings ranked
//      by looking at the parent's neighbour list.  The result depends only on the graph and the start vertex;
//   3. reverse, permute the matrix (thread per row: relabel, insertion sort by the new column).
// Further connected components are appended in the same way (at most kMaxComponents searches; whatever is left then
// keeps its relative order at the end).
#include <algorithm>
#include <vector>

#include "dpcg_host.h"
#include "dpcg_prims.h"

namespace dpcg {

namespace {
constexpr int kMaxComponents = 64;
constexpr int kBfsGrid = 512;        // persistent-style grid of a frontier step
constexpr int kBfsBatch = 64;        // frontier steps enqueued between two looks at the frontier sizes

inline int rows_grid(int64_t n, int cap = 8192) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ __launch_bounds__(kBlock) void k_degrees(int64_t n, const int32_t *__restrict__ rp, int32_t *__restrict__ deg) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) deg[i] = rp[i + 1] - rp[i];
}

// best = min over the vertices v of list[lo..hi) (list == nullptr: all vertices) that are still unvisited when
// `only_unvisited`, of (deg[v] << 32 | v)
__global__ __launch_bounds__(kBlock) void k_min_degree(const int32_t *__restrict__ list, int64_t lo, int64_t hi,
                                                       const int32_t *__restrict__ deg, const int32_t *__restrict__ level,
                                                       int only_unvisited, unsigned long long *best) {
    __shared__ unsigned long long sh[kBlock / 64];
    unsigned long long m = ~0ull;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = lo + (int64_t)blockIdx.x * kBlock + threadIdx.x; k < hi; k += stride) {
        const int v = list ? list[k] : (int)k;
        if (only_unvisited && level[v] >= 0) continue;
        const unsigned long long key = ((unsigned long long)(unsigned)deg[v] << 32) | (unsigned)v;
        m = key < m ? key : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(m, off);
        m = o < m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) m = sh[w] < m ? sh[w] : m;
        if (m != ~0ull) atomicMin(best, m);
    }
}

// seed a search: vertex v becomes the only member of level L, stored at order[at]
__global__ void k_bfs_seed(int v, int L, int at, int32_t *level, int32_t *order, int32_t *start, int32_t *count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        level[v] = L;
        order[at] = v;
        start[L] = at;
        count[L] = 1;
    }
}

// One frontier step: the vertices of level L (order[start[L] .. +count[L])) claim their unvisited neighbours for level
// L + 1 and append them behind the frontier.  start[L] and count[L] are final when this kernel runs.
__global__ __launch_bounds__(kBlock) void k_bfs_step(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                     int32_t *level, int32_t *order, int32_t *start, int32_t *count, int L) {
    const int s0 = start[L], c0 = count[L];
    if (c0 == 0) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) start[L + 1] = s0 + c0;
    const int stride = gridDim.x * kBlock;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < c0; idx += stride) {
        const int v = order[s0 + idx];
        const int ks = rp[v], ke = rp[v + 1];
        // eight neighbours at a time, phase by phase (columns, their levels, the claims): a handful of dependent memory round trips per
        // vertex instead of four per NEIGHBOUR (measured on the 10^4-vertex frontiers of a 1M-row 3-D system: 14 us a level)
        for (int k0 = ks; k0 < ke; k0 += 8) {
            int u[8], lu[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) u[q] = k0 + q < ke ? ci[k0 + q] : v;
#pragma unroll
            for (int q = 0; q < 8; ++q) lu[q] = u[q] != v ? level[u[q]] : 0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (lu[q] < 0) lu[q] = atomicCAS(&level[u[q]], -1, L + 1);          // -1: this thread claimed it
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (lu[q] == -1 && u[q] != v) order[s0 + c0 + atomicAdd(&count[L + 1], 1)] = u[q];
        }
    }
}

// A run of NARROW frontiers as one launch of one workgroup: a 2-D mesh of a million cells has ~2000 levels of ~500 vertices, and a
// launch per level costs ~7 us of which the work is a fraction (three searches: 42 of the 80 ms of its reordering).  Here the levels
// follow each other behind workgroup barriers -- what a thread appends to `order`, the next level's threads (same CU, same L1) read
// -- until the frontier is empty or wider than `cap`; out[0] = the first level NOT expanded, out[1] = its width.  What it saves is the
// launches, not the ~5 dependent memory accesses of a level: on a graph that does not fit the L2 those cost ~6 us a level either way
// (1M-cell mesh, frontiers of ~500: 45.9 ms against 42.3 for the launches), so the cap is narrow there (narrow_cap()).
constexpr int kNarrowCap = 2048;
__global__ __launch_bounds__(1024) void k_bfs_narrow(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, int32_t *level,
                                                     int32_t *order, int32_t *start, int32_t *count, int L, int max_level, int cap, int *out) {
    __shared__ int s_next, s_s0, s_c0;
    const int t = threadIdx.x;
    if (t == 0) {
        s_s0 = start[L];
        s_c0 = count[L];
        s_next = 0;
    }
    __syncthreads();
    for (;;) {
        const int s0 = s_s0, c0 = s_c0;
        if (c0 == 0 || c0 > cap || L >= max_level) break;
        for (int idx = t; idx < c0; idx += 1024) {
            const int v = order[s0 + idx];
            const int ks = rp[v], ke = rp[v + 1];
            for (int k0 = ks; k0 < ke; k0 += 8) {            // (as k_bfs_step: columns, their levels, the claims, phase by phase)
                int u[8], lu[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) u[q] = k0 + q < ke ? ci[k0 + q] : v;
#pragma unroll
                for (int q = 0; q < 8; ++q) lu[q] = u[q] != v ? level[u[q]] : 0;   // (a stale -1 only costs the atomic below)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (lu[q] < 0) lu[q] = atomicCAS(&level[u[q]], -1, L + 1);      // -1: this thread claimed it
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (lu[q] == -1 && u[q] != v) order[s0 + c0 + atomicAdd(&s_next, 1)] = u[q];
            }
        }
        __syncthreads();
        if (t == 0) {
            const int c1 = s_next;
            start[L + 1] = s0 + c0;
            count[L + 1] = c1;
            s_s0 = s0 + c0;
            s_c0 = c1;
            s_next = 0;
        }
        ++L;
        __syncthreads();
    }
    if (t == 0) {
        out[0] = L;
        out[1] = s_c0;
    }
}

// Cuthill-McKee, step A for level L: key[u] = position of u's first-numbered neighbour in level L - 1; that parent's
// child counter goes up by one.
__global__ __launch_bounds__(kBlock) void k_cm_keys(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                    const int32_t *__restrict__ level, const int32_t *__restrict__ order,
                                                    const int32_t *__restrict__ pos, int32_t *__restrict__ key,
                                                    int32_t *child, int s1, int c1, int L, int *err) {
    const int stride = gridDim.x * kBlock;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < c1; idx += stride) {
        const int u = order[s1 + idx];
        int mp = 0x7fffffff;
        const int ks = rp[u], ke = rp[u + 1];
        for (int k0 = ks; k0 < ke; k0 += 8) {            // eight neighbours at a time: columns, levels, positions (three round trips)
            int v[8], lv8[8], pv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ci[k0 + q < ke ? k0 + q : ke - 1];
#pragma unroll
            for (int q = 0; q < 8; ++q) lv8[q] = level[v[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) pv[q] = lv8[q] == L - 1 ? pos[v[q]] : 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 8; ++q) mp = pv[q] < mp ? pv[q] : mp;
        }
        key[u] = mp;
        // A vertex the search reached through a PARENT's row finds no level L - 1 neighbour in its OWN row only when the
        // pattern is not structurally symmetric (a one-sided entry, a triangle handed over by mistake): no parent to
        // count it under -- flag it and leave it unplaced (rcm_order then reports the pattern instead of a wrong order).
        if (mp == 0x7fffffff) atomicExch(err, 1);
        else atomicAdd(&child[mp], 1);
    }
}

// exclusive scan of child[lo .. lo+cnt) into base[lo .. lo+cnt), one workgroup: every thread sums its own run of consecutive
// elements, ONE block scan of the 1024 run sums, then the runs are written out (a chunk-by-chunk scan cost 20 barriers per 1024
// elements: ~15 us for the 10^4-vertex levels of a 3-D grid, the largest of the three launches of a level).
__global__ __launch_bounds__(1024) void k_scan_range(const int32_t *__restrict__ child, int32_t *__restrict__ base, int lo, int cnt) {
    __shared__ int sh[1024];
    const int t = threadIdx.x;
    const int per = (cnt + 1023) / 1024;
    const int a = t * per < cnt ? t * per : cnt, b = a + per < cnt ? a + per : cnt;
    int sum = 0;
    for (int i = a; i < b; ++i) sum += child[lo + i];
    sh[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = t >= off ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    int run = sh[t] - sum;                               // exclusive prefix of this thread's run
    for (int i = a; i < b; ++i) {
        const int v = child[lo + i];
        base[lo + i] = run;
        run += v;
    }
}

// step B: final position of u = start of its level + children of earlier parents + rank among its siblings
__global__ __launch_bounds__(kBlock) void k_cm_place(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                     const int32_t *__restrict__ level, const int32_t *__restrict__ order,
                                                     const int32_t *__restrict__ deg, const int32_t *__restrict__ key,
                                                     const int32_t *__restrict__ base, const int32_t *vertex_at_prev,
                                                     int32_t *__restrict__ pos, int32_t *vertex_at, int s1, int c1,
                                                     int L) {
    const int stride = gridDim.x * kBlock;
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < c1; idx += stride) {
        const int u = order[s1 + idx];
        const int mp = key[u];
        if (mp == 0x7fffffff) continue;                   // no parent (flagged by k_cm_keys)
        const int p = vertex_at_prev[mp];
        const int du = deg[u];
        int rank = 0;
        const int ks = rp[p], ke = rp[p + 1];
        for (int k0 = ks; k0 < ke; k0 += 8) {            // eight siblings-to-be at a time, phase by phase
            int w[8], lw[8], kw[8], dw[8];
            bool dup[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q;
                w[q] = k < ke ? ci[k] : u;
                dup[q] = k < ke && k > ks && ci[k - 1] == w[q];   // a duplicated entry (adjacent: columns ascend) counts once
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) lw[q] = level[w[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                kw[q] = key[w[q]];
                dw[q] = deg[w[q]];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (!dup[q] && w[q] != u && lw[q] == L && kw[q] == mp && (dw[q] < du || (dw[q] == du && w[q] < u))) ++rank;
        }
        const int np = s1 + base[mp] + rank;
        pos[u] = np;
        vertex_at[np] = u;
    }
}

// Cuthill-McKee positions of a run of NARROW levels (this one and the one before it of at most 1024 vertices) as one launch of one
// workgroup: the same order as k_cm_keys / k_scan_range / k_cm_place give -- a level sorted by (position of the first-numbered
// parent, degree, index) -- with the children counted, scanned and ranked among their siblings in LDS (the three launches of a level
// are ~15 dependent memory round trips, 18.6 us; 37 of the 80 ms of a 1M-cell 2-D mesh's reordering).
__global__ __launch_bounds__(1024) void k_cm_narrow(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                    const int32_t *__restrict__ level, const int32_t *__restrict__ order,
                                                    const int32_t *__restrict__ deg, const int32_t *__restrict__ start,
                                                    const int32_t *__restrict__ count, int32_t *pos, int32_t *vertex_at, int L_first,
                                                    int L_last, int *err) {
    __shared__ int s_deg[1024], s_u[1024], s_child[1024], s_base[1024], s_grp[1024], s_wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int L = L_first; L <= L_last; ++L) {
        const int s1 = start[L], c1 = count[L], s0 = start[L - 1];
        s_child[t] = 0;
        __syncthreads();
        int u = -1, mp = 0x7fffffff, du = 0, slot = 0;
        if (t < c1) {
            u = order[s1 + t];
            const int ks = rp[u], ke = rp[u + 1];
            for (int k0 = ks; k0 < ke; k0 += 8) {            // eight neighbours at a time: columns, levels, positions
                int v[8], lv8[8], pv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = ci[k0 + q < ke ? k0 + q : ke - 1];
#pragma unroll
                for (int q = 0; q < 8; ++q) lv8[q] = level[v[q]];
#pragma unroll
                for (int q = 0; q < 8; ++q) pv[q] = lv8[q] == L - 1 ? pos[v[q]] : 0x7fffffff;
#pragma unroll
                for (int q = 0; q < 8; ++q) mp = pv[q] < mp ? pv[q] : mp;
            }
            du = deg[u];
            if (mp == 0x7fffffff) atomicExch(err, 1);       // (no parent in its own row: the pattern is not symmetric)
            else slot = atomicAdd(&s_child[mp - s0], 1);     // arrival among the parent's children (any order)
        }
        s_deg[t] = du;
        s_u[t] = u;
        __syncthreads();
        {   // exclusive scan of the children counts over the (at most 1024) parents
            const int mine = s_child[t];
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off);
                if (lane >= off) incl += y;
            }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            int before = 0;
            for (int w = 0; w < wave; ++w) before += s_wsum[w];
            s_base[t] = before + incl - mine;
        }
        __syncthreads();
        const bool placed = u >= 0 && mp != 0x7fffffff;
        if (placed) s_grp[s_base[mp - s0] + slot] = t;
        __syncthreads();
        if (placed) {
            const int b = s_base[mp - s0], e = b + s_child[mp - s0];
            int rank = 0;
            for (int j = b; j < e; ++j) {
                const int o = s_grp[j];
                const int dw = s_deg[o], w = s_u[o];
                if (o != t && (dw < du || (dw == du && w < u))) ++rank;
            }
            const int np = s1 + b + rank;
            pos[u] = np;
            vertex_at[np] = u;
        }
        __syncthreads();
    }
}

__global__ void k_cm_root(const int32_t *__restrict__ order, int32_t *pos, int32_t *vertex_at, int at) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int v = order[at];
        pos[v] = at;
        vertex_at[at] = v;
    }
}

// vertices no search reached: flag them (for the scan that appends them in index order)
__global__ __launch_bounds__(kBlock) void k_flag_unvisited(int64_t n, const int32_t *__restrict__ level, int32_t *__restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) flag[i] = level[i] < 0 ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_place_unvisited(int64_t n, const int32_t *__restrict__ level,
                                                            const int32_t *__restrict__ offs, int at, int32_t *__restrict__ pos) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        if (level[i] < 0) pos[i] = at + offs[i];
}

// reverse Cuthill-McKee: new index of vertex u = n - 1 - pos[u]
// perm arrives filled with -1; a position out of range is flagged, a position taken twice leaves another slot at -1
__global__ __launch_bounds__(kBlock) void k_finish_perm(int64_t n, const int32_t *__restrict__ pos, int32_t *__restrict__ perm,
                                                        int32_t *__restrict__ iperm, int *err) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < n; u += stride) {
        const int64_t q = pos[u];
        if (q < 0 || q >= n) {
            atomicExch(err, 1);
            continue;
        }
        const int r = (int)(n - 1 - q);
        iperm[u] = r;
        perm[r] = (int32_t)u;
    }
}

// perm is a bijection iff every slot was written and maps back: perm[r] = u with iperm[u] = r
__global__ __launch_bounds__(kBlock) void k_check_perm(int64_t n, const int32_t *__restrict__ perm, const int32_t *__restrict__ iperm,
                                                       int *err) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) {
        const int u = perm[r];
        if (u < 0 || u >= n || iperm[u] != r) atomicExch(err, 1);
    }
}

// ---- applying a permutation ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_perm_lengths(int64_t n, const int32_t *__restrict__ perm,
                                                         const int32_t *__restrict__ rp, int32_t *__restrict__ len) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r <= n; r += stride)
        len[r] = r < n ? rp[perm[r] + 1] - rp[perm[r]] : 0;
}

// row r of P A P^T = row perm[r] of A with columns relabelled through iperm, sorted ascending (the values move with their columns,
// so every product of a row sum is the same number as before, only their order in the sum follows the new column order).
// A row of up to kPermStage entries is staged in the thread's own LDS column: its loads are independent of each other, the
// insertion sort runs in LDS and the row is written once (sorting in place in global memory -- a dependent load of the entry just
// written per step -- took 6.2 ms for the 117M entries of a 256^3 system); longer rows sort in place.
constexpr int kPermStage = 16;
__global__ __launch_bounds__(kBlock) void k_perm_rows(int64_t n, const int32_t *__restrict__ perm,
                                                      const int32_t *__restrict__ iperm, const int32_t *__restrict__ rp,
                                                      const int32_t *__restrict__ ci, const double *__restrict__ v,
                                                      const int32_t *__restrict__ nrp, int32_t *__restrict__ nci,
                                                      double *__restrict__ nv) {
    __shared__ int sc[kPermStage][kBlock];
    __shared__ double sx[kPermStage][kBlock];
    const int tid = threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) {
        const int old = perm[r];
        const int src = rp[old], len = rp[old + 1] - src, dst = nrp[r];
        if (len <= kPermStage) {
            for (int k = 0; k < len; ++k) {
                sc[k][tid] = iperm[ci[src + k]];
                sx[k][tid] = v[src + k];
            }
            for (int k = 1; k < len; ++k) {
                const int c = sc[k][tid];
                if (sc[k - 1][tid] <= c) continue;               // (already in place: the common case)
                const double x = sx[k][tid];
                int q = k;
                while (q > 0 && sc[q - 1][tid] > c) {
                    sc[q][tid] = sc[q - 1][tid];
                    sx[q][tid] = sx[q - 1][tid];
                    --q;
                }
                sc[q][tid] = c;
                sx[q][tid] = x;
            }
            for (int k = 0; k < len; ++k) {
                nci[dst + k] = sc[k][tid];
                nv[dst + k] = sx[k][tid];
            }
            continue;
        }
        for (int k = 0; k < len; ++k) {
            const int c = iperm[ci[src + k]];
            const double x = v[src + k];
            int q = dst + k;
            while (q > dst && nci[q - 1] > c) {
                nci[q] = nci[q - 1];
                nv[q] = nv[q - 1];
                --q;
            }
            nci[q] = c;
            nv[q] = x;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_gather_vec(int64_t n, const int32_t *__restrict__ perm, const T *__restrict__ in,
                                                       T *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) out[r] = in[perm[r]];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scatter_vec(int64_t n, const int32_t *__restrict__ perm, const T *__restrict__ in,
                                                        T *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) out[perm[r]] = in[r];
}

__global__ __launch_bounds__(kBlock) void k_relabel(int64_t count, const int32_t *__restrict__ map, int32_t *__restrict__ idx) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < count; k += stride) idx[k] = map[idx[k]];
}

// Locality of the x gather: distinct 128-byte lines of x (16 doubles) that the columns of each 256-row block touch,
// summed over the blocks.  One workgroup per block, an open-addressing set in LDS.
constexpr int kLineSetSlots = 8192;
__global__ __launch_bounds__(kBlock) void k_block_lines(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                        int nrb, unsigned long long *total) {
    __shared__ int set[kLineSetSlots];
    __shared__ int distinct;
    for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        for (int e = threadIdx.x; e < kLineSetSlots; e += kBlock) set[e] = -1;
        if (threadIdx.x == 0) distinct = 0;
        __syncthreads();
        const int64_t r0 = (int64_t)rb * kStreamRows;
        const int64_t r1 = r0 + kStreamRows < n ? r0 + kStreamRows : n;
        const int lo = rp[r0], hi = rp[r1];
        int mine = 0;
        for (int k = lo + threadIdx.x; k < hi; k += kBlock) {
            const int line = ci[k] >> 4;
            unsigned h = ((unsigned)line * 2654435761u) & (kLineSetSlots - 1);
            for (int probe = 0; probe < kLineSetSlots; ++probe) {
                const int prev = atomicCAS(&set[h], -1, line);
                if (prev == -1) {
                    ++mine;
                    break;
                }
                if (prev == line) break;
                h = (h + 1) & (kLineSetSlots - 1);
            }
        }
        if (mine) atomicAdd(&distinct, mine);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(total, (unsigned long long)distinct);
        __syncthreads();
    }
}

template <typename T>
struct Buf {
    T *p = nullptr;
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    ~Buf() { dev_free(p); }
    int alloc(int64_t count) { dev_free(p); return dev_alloc(&p, count); }
    T *release() { T *q = p; p = nullptr; return q; }
};
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
void launch_gather_f64(int64_t n, const int32_t *perm, const double *in, double *out, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_vec<double>, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, perm, in, out);
}
void launch_scatter_f64(int64_t n, const int32_t *perm, const double *in, double *out, hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_vec<double>, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, perm, in, out);
}
void launch_gather_f32(int64_t n, const int32_t *perm, const float *in, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_vec<float>, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, perm, in, out);
}
void launch_scatter_f32(int64_t n, const int32_t *perm, const float *in, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_vec<float>, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, perm, in, out);
}
void launch_relabel(int64_t count, const int32_t *map, int32_t *idx, hipStream_t s) {
    hipLaunchKernelGGL(k_relabel, dim3(rows_grid(count)), dim3(kBlock), 0, s, count, map, idx);
}

// lines * 128 bytes fetched for the x gather per byte of x values used: ~1 for banded blocks, 16 when every column is
// its own line
int gather_line_ratio(const CsrDev &A, double *ratio, hipStream_t s) {
    *ratio = 0.0;
    if (A.nnz <= 0) return DPCG_OK;
    Buf<unsigned long long> total;
    DPCG_TRY(total.alloc(1));
    DPCG_HIP(hipMemsetAsync(total.p, 0, sizeof(unsigned long long), s));
    const int nrb = (int)((A.n + kStreamRows - 1) / kStreamRows);
    hipLaunchKernelGGL(k_block_lines, dim3(nrb < 4096 ? nrb : 4096), dim3(kBlock), 0, s, A.n, A.rowptr, A.col, nrb, total.p);
    unsigned long long h = 0;
    DPCG_HIP(hipMemcpyAsync(&h, total.p, sizeof(h), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    DPCG_CHECK_LAUNCH();
    *ratio = (double)h * 128.0 / ((double)A.nnz * 8.0);
    return DPCG_OK;
}

// B = P A P^T for perm[new] = old, iperm[old] = new (device arrays); B owns its arrays.
int permute_csr(const CsrDev &A, const int32_t *perm, const int32_t *iperm, CsrDev &B, hipStream_t s) {
    const int64_t n = A.n;
    B = CsrDev();
    B.n = n;
    B.nnz = A.nnz;
    B.owned = true;
    Buf<int32_t> len;
    DPCG_TRY(len.alloc(n + 1));
    DPCG_TRY(dev_alloc(&B.rowptr, n + 1));
    DPCG_TRY(dev_alloc(&B.col, A.nnz));
    DPCG_TRY(dev_alloc(&B.val, A.nnz));
    hipLaunchKernelGGL(k_perm_lengths, dim3(rows_grid(n + 1)), dim3(kBlock), 0, s, n, perm, A.rowptr, len.p);
    DPCG_TRY(exclusive_scan_i32(len.p, B.rowptr, n + 1, s));
    hipLaunchKernelGGL(k_perm_rows, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, perm, iperm, A.rowptr, A.col, A.val, B.rowptr,
                       B.col, B.val);
    DPCG_HIP(hipStreamSynchronize(s));
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

namespace {
// Breadth-first search from `root` as level L0 at order position `at`.  Returns the number of levels found and the
// number of vertices reached; start/count (host copies) are filled for the levels of this search.
int bfs(const CsrDev &A, int root, int L0, int at, int32_t *level, int32_t *order, int32_t *start, int32_t *count,
        std::vector<int32_t> &h_start, std::vector<int32_t> &h_count, int *n_levels, hipStream_t s) {
    hipLaunchKernelGGL(k_bfs_seed, dim3(1), dim3(64), 0, s, root, L0, at, level, order, start, count);
    static const bool narrow_on = [] { const char *e = getenv("DPCG_RCM_NARROW"); return !(e && e[0] == '0'); }();   // development knob
    Buf<int> where;
    DPCG_TRY(where.alloc(2));
    int L = L0;
    int32_t last = 1;                                      // width of level L (the seed)
    // measured, create + RCM, a launch per level -> narrow runs: 60K-cell mesh 16.1 -> 6.4 ms, 30K rows in two components 12.0 -> 4.1,
    // a 5000-vertex path 131 -> 28.5, 600^2 scrambled 38.6 -> 21.5 (tools/rcm_narrow_ab.py)
    const int cap = A.n <= 524288 ? kNarrowCap : 128;
    for (;;) {
        if (narrow_on && last <= cap) {                    // narrow frontiers: one workgroup walks them, level after level
            hipLaunchKernelGGL(k_bfs_narrow, dim3(1), dim3(1024), 0, s, A.rowptr, A.col, level, order, start, count, L, (int)A.n + 1, cap,
                               where.p);
            int h_where[2] = {0, 0};
            DPCG_HIP(hipMemcpyAsync(h_where, where.p, sizeof(h_where), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            L = h_where[0];
            last = h_where[1];
            if (last == 0 || (int64_t)L >= A.n + 1) break;
        }
        for (int b = 0; b < kBfsBatch; ++b)
            hipLaunchKernelGGL(k_bfs_step, dim3(kBfsGrid), dim3(kBlock), 0, s, A.rowptr, A.col, level, order, start, count, L + b);
        L += kBfsBatch;
        DPCG_HIP(hipMemcpyAsync(&last, count + L, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (last == 0 || (int64_t)L >= A.n + 1) break;
    }
    const int span = L - L0 + 1;
    h_start.resize((size_t)span);
    h_count.resize((size_t)span);
    DPCG_HIP(hipMemcpyAsync(h_count.data(), count + L0, (size_t)span * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipMemcpyAsync(h_start.data(), start + L0, (size_t)span * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    int nl = 0;
    while (nl < span && h_count[(size_t)nl] > 0) ++nl;
    *n_levels = nl;
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}
}  // namespace

// perm_out[new] = old and iperm_out[old] = new (device, n entries each, owned by the caller afterwards).
int rcm_order(const CsrDev &A, int32_t **perm_out, int32_t **iperm_out, int *n_components, hipStream_t s) {
    const int64_t n = A.n;
    if (n + 2 + kBfsBatch > 2147483000LL) return invalid("reordering: system too large");
    Buf<int32_t> deg, level, order, start, count, pos, vertex_at, key, child, base, perm, iperm;
    Buf<unsigned long long> best;
    Buf<int> err;
    DPCG_TRY(err.alloc(1));
    DPCG_HIP(hipMemsetAsync(err.p, 0, sizeof(int), s));
    const int64_t nlv = n + 2 + 2 * kBfsBatch;
    DPCG_TRY(deg.alloc(n)); DPCG_TRY(level.alloc(n)); DPCG_TRY(order.alloc(n + 1)); DPCG_TRY(start.alloc(nlv));
    DPCG_TRY(count.alloc(nlv)); DPCG_TRY(pos.alloc(n)); DPCG_TRY(vertex_at.alloc(n + 1)); DPCG_TRY(key.alloc(n));
    DPCG_TRY(child.alloc(n + 1)); DPCG_TRY(base.alloc(n + 1)); DPCG_TRY(perm.alloc(n)); DPCG_TRY(iperm.alloc(n));
    DPCG_TRY(best.alloc(1));
    hipLaunchKernelGGL(k_degrees, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, A.rowptr, deg.p);
    DPCG_HIP(hipMemsetAsync(level.p, 0xff, (size_t)n * sizeof(int32_t), s));
    DPCG_HIP(hipMemsetAsync(count.p, 0, (size_t)nlv * sizeof(int32_t), s));
    DPCG_HIP(hipMemsetAsync(child.p, 0, (size_t)(n + 1) * sizeof(int32_t), s));
    DPCG_HIP(hipMemsetAsync(pos.p, 0xff, (size_t)n * sizeof(int32_t), s));      // -1: not placed
    auto min_degree = [&](const int32_t *list, int64_t lo, int64_t hi, int only_unvisited, int *v) -> int {
        DPCG_HIP(hipMemsetAsync(best.p, 0xff, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_min_degree, dim3(rows_grid(hi - lo, 1024)), dim3(kBlock), 0, s, list, lo, hi, deg.p, level.p,
                           only_unvisited, best.p);
        unsigned long long h = 0;
        DPCG_HIP(hipMemcpyAsync(&h, best.p, sizeof(h), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        *v = h == ~0ull ? -1 : (int)(h & 0xffffffffu);
        return DPCG_OK;
    };
    PhaseTimer pt(s);
    int n_bfs = 0;
    std::vector<int32_t> h_start, h_count;
    struct Comp { int L0, nl; std::vector<int32_t> start, count; };
    std::vector<Comp> comps;
    int visited = 0, next_level = 0;
    while (visited < n && (int)comps.size() < kMaxComponents) {
        int root = -1;
        DPCG_TRY(min_degree(nullptr, 0, n, 1, &root));
        if (root < 0) break;
        int nl = 0;
        DPCG_TRY(bfs(A, root, next_level, visited, level.p, order.p, start.p, count.p, h_start, h_count, &nl, s));
        ++n_bfs;
        if (comps.empty()) {
            // pseudo-peripheral start vertex for the first (normally the only) component
            // ONE move: from the vertex of smallest degree to the smallest-degree vertex of its last level.  Moving on while the
            // search gets deeper (George-Liu, up to four more searches) cost a 1M-cell 2-D mesh one search of ~2000 levels more
            // (64.1 -> 49.4 ms to create the system), a Delaunay graph two (37.6 -> 25.6), for the same SpMV time (14.1 / 17.1 us):
            // tools/rcm_narrow_ab.py.  DPCG_RCM_SWEEPS: development knob.
            static const int sweeps = [] { const char *e = getenv("DPCG_RCM_SWEEPS"); return e ? atoi(e) : 1; }();
            for (int sweep = 0; sweep < sweeps; ++sweep) {
                const int last = nl - 1;
                int cand = -1;
                DPCG_TRY(min_degree(order.p, h_start[(size_t)last], (int64_t)h_start[(size_t)last] + h_count[(size_t)last], 0, &cand));
                if (cand < 0 || cand == root) break;
                // search again from the candidate: forget this component
                DPCG_HIP(hipMemsetAsync(level.p, 0xff, (size_t)n * sizeof(int32_t), s));
                DPCG_HIP(hipMemsetAsync(count.p, 0, (size_t)nlv * sizeof(int32_t), s));
                int nl2 = 0;
                std::vector<int32_t> st2, ct2;
                DPCG_TRY(bfs(A, cand, next_level, visited, level.p, order.p, start.p, count.p, st2, ct2, &nl2, s));
                ++n_bfs;
                const bool deeper = nl2 > nl;
                root = cand;
                nl = nl2;
                h_start.swap(st2);
                h_count.swap(ct2);
                if (!deeper) break;
            }
        }
        Comp c;
        c.L0 = next_level;
        c.nl = nl;
        c.start.assign(h_start.begin(), h_start.begin() + nl);
        c.count.assign(h_count.begin(), h_count.begin() + nl);
        int reached = 0;
        for (int l = 0; l < nl; ++l) reached += c.count[(size_t)l];
        visited += reached;
        next_level += nl;
        // later searches index count[] / start[] from next_level on: clear what this search's empty tail steps touched
        comps.push_back(std::move(c));
    }
    if (pt.on) fprintf(stderr, "[dpcg setup]   %d breadth-first searches, %d levels\n", n_bfs, next_level);
    pt.mark("  RCM: searches");
    // Cuthill-McKee positions, component by component, level by level.  (A run of narrow levels as ONE launch of ONE workgroup,
    // the three steps with workgroup barriers in between: 19.4 us per level at 1024^2 against 17.8 for the three launches -- a
    // single workgroup pays every dependent load in full.  Both level loops were also tried as ONE persistent launch
    // each -- 64 workgroups, a grid barrier per phase, agent-scope accesses for what crosses workgroups: 23 / 56 us per level at 1M
    // rows (3-D) against 14 / 29 us for the launches below; with release / acquire fences instead, which write the L2 back, 39 / 100 us.)
    for (const Comp &c : comps) {
        hipLaunchKernelGGL(k_cm_root, dim3(1), dim3(64), 0, s, order.p, pos.p, vertex_at.p, c.start[0]);
        static const bool narrow_on = [] { const char *e = getenv("DPCG_RCM_NARROW"); return !(e && e[0] == '0'); }();
        for (int l = 1; l < c.nl; ++l) {
            if (narrow_on && c.count[(size_t)l] <= 1024 && c.count[(size_t)l - 1] <= 1024) {     // a run of narrow levels: one launch
                int e = l + 1;
                while (e < c.nl && c.count[(size_t)e] <= 1024) ++e;
                hipLaunchKernelGGL(k_cm_narrow, dim3(1), dim3(1024), 0, s, A.rowptr, A.col, level.p, order.p, deg.p, start.p, count.p, pos.p,
                                   vertex_at.p, c.L0 + l, c.L0 + e - 1, err.p);
                l = e - 1;
                continue;
            }
            const int L = c.L0 + l, s1 = c.start[(size_t)l], c1 = c.count[(size_t)l];
            const int s0 = c.start[(size_t)l - 1], c0 = c.count[(size_t)l - 1];
            const int g = rows_grid(c1, 1024);
            hipLaunchKernelGGL(k_cm_keys, dim3(g), dim3(kBlock), 0, s, A.rowptr, A.col, level.p, order.p, pos.p, key.p, child.p,
                               s1, c1, L, err.p);
            hipLaunchKernelGGL(k_scan_range, dim3(1), dim3(1024), 0, s, child.p, base.p, s0, c0);
            hipLaunchKernelGGL(k_cm_place, dim3(g), dim3(kBlock), 0, s, A.rowptr, A.col, level.p, order.p, deg.p, key.p, base.p,
                               vertex_at.p, pos.p, vertex_at.p, s1, c1, L);
        }
    }
    pt.mark("  RCM: positions");
    if (visited < n) {   // more components than searches: the rest keeps its relative order at the end
        Buf<int32_t> flag, offs;
        DPCG_TRY(flag.alloc(n));
        DPCG_TRY(offs.alloc(n));
        hipLaunchKernelGGL(k_flag_unvisited, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, level.p, flag.p);
        DPCG_TRY(exclusive_scan_i32(flag.p, offs.p, n, s));
        hipLaunchKernelGGL(k_place_unvisited, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, level.p, offs.p, visited, pos.p);
    }
    // pos[] of a vertex that was never placed is whatever the allocation held: start from an invalid value
    DPCG_HIP(hipMemsetAsync(perm.p, 0xff, (size_t)n * sizeof(int32_t), s));
    DPCG_HIP(hipMemsetAsync(iperm.p, 0xff, (size_t)n * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_finish_perm, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, pos.p, perm.p, iperm.p, err.p);
    // The construction yields a permutation for a structurally symmetric pattern without duplicate entries; it is CHECKED,
    // not assumed: a one-sided entry leaves a vertex without a parent, a duplicated column ranks two siblings alike.
    hipLaunchKernelGGL(k_check_perm, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, perm.p, iperm.p, err.p);
    int bad = 0;
    DPCG_HIP(hipMemcpyAsync(&bad, err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    DPCG_CHECK_LAUNCH();
    if (bad)
        return invalid("reordering: the sparsity pattern is not structurally symmetric or has duplicate entries "
                       "(reverse Cuthill-McKee needs the pattern of a symmetric matrix, columns ascending)");
    if (n_components) *n_components = (int)comps.size() + (visited < n ? 1 : 0);
    *perm_out = perm.release();
    *iperm_out = iperm.release();
    return DPCG_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Multicolour ordering (for IC(0) applied by triangular solves): vertices of one colour share no edge, so ordered colour
// by colour the factor's dependency graph is only as deep as the number of colours -- 2 for a bipartite mesh graph (every
// 5- / 7-point grid, scrambled or not), a handful in general -- instead of the hundreds of levels of a natural ordering.
// Fewer, wider levels is what the solve kernels want; the price is a somewhat weaker preconditioner (measured with the
// CPU oracle on the scrambled 48^3 system: 42 iterations / 19 levels in the caller's order, 51 / 2 red-black, Jacobi 102).
//   1. two colours by breadth-first parity when that is a proper colouring (checked edge by edge);
//   2. otherwise greedy colouring in the order of a hashed priority (Jones-Plassmann): in every round the uncoloured
//      vertices that beat all their uncoloured neighbours take the smallest colour none of their neighbours has.  A vertex's
//      colour depends only on its higher-priority neighbours, so the result is that of the SEQUENTIAL greedy colouring in
//      priority order whatever the timing: deterministic.
// perm[new] = old lists the vertices colour by colour, in ascending index order inside a colour (stable).
// ------------------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ unsigned jp_priority(int v) {
    unsigned x = (unsigned)v * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    return x;
}

// Peel ranks (an approximate smallest-degree-last order, a handful of rounds): in round r every remaining vertex whose degree
// among the remaining ones is at most max(T, smallest remaining degree) leaves and gets rank r; whoever outlasts the rounds keeps
// the top rank.  Greedy colouring in DESCENDING rank then gives a vertex at most ~T already coloured neighbours when its turn comes,
// i.e. about T + 1 colours: with T = 4 the Delaunay graph (average degree 6) takes 5 colours in 15 rounds where the hashed order
// takes 6 (and 7 before the recolouring passes) -- a level less for the triangular solves, and under the sweep kernels' limit.
__global__ __launch_bounds__(kBlock) void k_peel_init(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                      int32_t *__restrict__ rank, int32_t *__restrict__ deg, int top, int *ctl) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int lo = 0x7fffffff;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        int d = 0;
        for (int k = rp[v]; k < rp[v + 1]; ++k) d += ci[k] != v ? 1 : 0;
        deg[v] = d;
        rank[v] = top;
        lo = d < lo ? d : lo;
    }
    if (lo != 0x7fffffff) atomicMin(ctl, lo);
}
__global__ __launch_bounds__(kBlock) void k_peel_mark(int64_t n, const int32_t *__restrict__ deg, int32_t *__restrict__ rank, int round,
                                                      int top, int T, const int *__restrict__ min_alive) {
    const int lo = *min_alive;
    const int thr = T > lo ? T : lo;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride)
        if (rank[v] == top && deg[v] <= thr) rank[v] = round;
}
__global__ __launch_bounds__(kBlock) void k_peel_degrees(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                         const int32_t *__restrict__ rank, int top, int32_t *__restrict__ deg,
                                                         int *min_alive) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int lo = 0x7fffffff;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        if (rank[v] != top) continue;
        int d = 0;
        for (int k = rp[v]; k < rp[v + 1]; ++k) {
            const int u = ci[k];
            d += (u != v && rank[u] == top) ? 1 : 0;
        }
        deg[v] = d;
        lo = d < lo ? d : lo;
    }
    if (lo != 0x7fffffff) atomicMin(min_alive, lo);
}

__global__ __launch_bounds__(kBlock) void k_jp_round(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                     int32_t *color, int *remaining, int *err, const int32_t *__restrict__ rank) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int left = 0;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        if (__hip_atomic_load(color + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) continue;
        const unsigned pv = jp_priority((int)v);
        const int rv = rank ? rank[v] : 0;
        bool top = true;
        unsigned long long used = 0ull;
        for (int k = rp[v]; k < rp[v + 1] && top; ++k) {
            const int u = ci[k];
            if (u == v) continue;
            const int cu = __hip_atomic_load(color + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cu >= 0) used |= 1ull << cu;
            else {
                const unsigned pu = jp_priority(u);
                const int ru = rank ? rank[u] : 0;                  // the later a vertex left the peel, the earlier it is coloured
                if (ru > rv || (ru == rv && (pu > pv || (pu == pv && u > v)))) top = false;
            }
        }
        if (top) {
            const int c = __ffsll((long long)~used) - 1;          // smallest free colour
            if (c < 0 || c >= 63) atomicExch(err, 1);
            else __hip_atomic_store(color + v, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            ++left;
        }
    }
    if (left) atomicAdd(remaining, left);
}

// Iterated greedy (Culberson): the graph is coloured AFRESH, greedily, class by class of the previous colouring -- a class is an
// independent set, so its vertices can all be coloured at once, each with the smallest colour none of its already recoloured
// neighbours has.  The new colouring never has more colours than the old one; taking the classes in another order (last to first)
// tends to need fewer.  Deterministic: a vertex reads neighbours of other classes only, and those do not change during the launch.
__global__ __launch_bounds__(kBlock) void k_recolor_class(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                          const int32_t *__restrict__ old_color, int32_t *new_color, int cls) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        if (old_color[v] != cls) continue;
        unsigned long long used = 0ull;
        for (int k = rp[v]; k < rp[v + 1]; ++k) {
            const int u = ci[k];
            const int cu = u != v ? new_color[u] : -1;
            if (cu >= 0) used |= 1ull << cu;
        }
        new_color[v] = __ffsll((long long)~used) - 1;
    }
}

// A LAST colour class of a handful of vertices (greedy colourings of meshes with triangles leave such: 2-6 vertices of a fifth
// colour on a 1M-row quadtree mesh) costs the triangular solves a level of its own -- two launches for nothing.  One wave folds it
// away where a local move does: a vertex v of the last class takes colour c < last when no neighbour has c, or when exactly ONE
// neighbour u has c and u can move to another colour c2 < last that none of u's neighbours has.  Vertices are taken one after
// another in index order (moves see the moves before them): deterministic.  Whatever stays keeps its colour: never more colours.
constexpr int kFoldMax = 256;          // largest last class the fold looks at (one wave, ~3 us a vertex)
__global__ __launch_bounds__(kBlock) void k_collect_class(int64_t n, const int32_t *__restrict__ color, int cls, int *count, int *list) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride)
        if (color[v] == cls) {
            const int at = atomicAdd(count, 1);
            if (at < kFoldMax) list[at] = (int)v;
        }
}
__global__ __launch_bounds__(64) void k_fold_tiny_class(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                        int32_t *color, int last, const int *__restrict__ count_in,
                                                        const int *__restrict__ list_in, int *left) {
    __shared__ int list[kFoldMax];
    const int lane = threadIdx.x;
    const int s_count = *count_in;
    if (s_count > kFoldMax) {                                      // (not tiny after all: leave it)
        if (lane == 0) *left = s_count;
        return;
    }
    // the members in index order, whatever order the collection appended them in
    for (int e = lane; e < s_count; e += 64) {
        const int mine = list_in[e];
        int rank = 0;
        for (int o = 0; o < s_count; ++o) rank += list_in[o] < mine ? 1 : 0;
        list[rank] = mine;
    }
    __syncthreads();
    const int count = s_count;
    int remaining = s_count;
    for (int q = 0; q < count; ++q) {
        const int v = list[q];
        const int dv = rp[v + 1] - rp[v];
        if (dv > 64) continue;
        const int u = lane < dv ? ci[rp[v] + lane] : v;            // lane <-> neighbour (the diagonal entry counts as "v itself")
        const int cu = u != v ? __hip_atomic_load(color + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
        bool done = false;
        for (int c = 0; c < last && !done; ++c) {
            const unsigned long long has = __ballot(cu == c);
            const int cnt = __popcll(has);
            if (cnt == 0) {
                if (lane == 0) __hip_atomic_store(color + v, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
            } else if (cnt == 1) {
                const int src = __ffsll((long long)has) - 1;
                const int w = __shfl(u, src);                          // the one neighbour of colour c
                const int dw = rp[w + 1] - rp[w];
                if (dw > 64) continue;
                const int x = lane < dw ? ci[rp[w] + lane] : w;
                const int cx = (x != w && x != v) ? __hip_atomic_load(color + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
                for (int c2 = 0; c2 < last; ++c2) {
                    if (c2 == c) continue;
                    if (__ballot(cx == c2) == 0) {                     // w may take c2 (v leaves `last`, so v does not block it)
                        if (lane == 0) {
                            __hip_atomic_store(color + w, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(color + v, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        done = true;
                        break;
                    }
                }
            }
        }
        if (done) --remaining;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the next vertex reads what this one stored
        __syncthreads();
    }
    if (lane == 0) *left = remaining;
}

__global__ __launch_bounds__(kBlock) void k_color_histogram(int64_t n, const int32_t *__restrict__ color, int *hist) {
    __shared__ int sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) atomicAdd(&sh[color[v] & 63], 1);
    __syncthreads();
    if (threadIdx.x < 64 && sh[threadIdx.x]) atomicAdd(hist + threadIdx.x, sh[threadIdx.x]);
}

static int h_hist_last_nonempty(const int *hist) {
    for (int c = 63; c >= 0; --c)
        if (hist[c] > 0) return hist[c];
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_remap_colors(int64_t n, int32_t *color, const int32_t *__restrict__ map) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) color[v] = map[color[v]];
}

__global__ __launch_bounds__(kBlock) void k_parity_colors(int64_t n, const int32_t *__restrict__ level, int32_t *__restrict__ color) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) color[v] = level[v] < 0 ? -1 : (level[v] & 1);
}

// flag = 1 when some edge joins two vertices of one colour (or a vertex has no colour)
__global__ __launch_bounds__(kBlock) void k_check_coloring(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                           const int32_t *__restrict__ color, int *flag) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        const int cv = color[v];
        bool bad = cv < 0;
        for (int k = rp[v]; k < rp[v + 1]; ++k) bad = bad || (ci[k] != v && color[ci[k]] == cv);
        if (bad) atomicExch(flag, 1);
    }
}

// ---- two colours by REGIONS: kRegions seeds spread over the index range grow level-synchronously, so that the searches meet after
// a fraction of the graph's diameter (a single breadth-first search of a 256^2 grid is 511 dependent steps, of a 100^3 grid 298: at
// ~3 launches a step that was 3 of the 4.4 / 5.3 ms of an IC(0) setup in multicolour order).  state[v] = region << 1 | parity within
// the region, step[v] = the step at which v joined (-1: not yet).
constexpr int kRegionsSmall = 256, kRegionsLarge = 1024;      // seeds: 1024 from 2M vertices on (a 256^3 grid: ~40 -> ~25 growth steps of 0.5 ms each)
constexpr int kRegionBatch = 16;      // growth steps enqueued between two looks at the visited count

__global__ void k_region_seed(int64_t n, int32_t *step, int32_t *state, int *visited, int kRegions, int32_t *front, int *front_n) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= kRegions) return;
    // hashed positions: evenly spaced indices would line the seeds up along one edge of a naturally ordered grid
    // (two seeds may coincide -- the later region simply stays empty)
    const int64_t v = (int64_t)(((unsigned long long)jp_priority(r + 1) * (unsigned long long)n) >> 32);
    if (atomicCAS(step + v, -1, 0) == -1) {
        state[v] = r << 1;
        atomicAdd(visited, 1);
        front[atomicAdd(front_n, 1)] = (int32_t)v;
    }
}

// One growth step by a scan of every vertex (systems below 2M rows, where a step is a few microseconds and one launch instead of
// two matters more than the vertices looked at in vain): an unvisited vertex with a neighbour that joined at step `cur` joins the region of the FIRST such neighbour in
// its row (ascending columns: deterministic), with the opposite parity
// `skip` (may be null): entries the growth does not walk along (edges that lie on a triangle, k_triangle_edges)
__global__ __launch_bounds__(kBlock) void k_region_grow(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                        const uint8_t *__restrict__ skip, int32_t *step, int32_t *state, int cur,
                                                        int *visited) {
    __shared__ int sh_new;
    if (threadIdx.x == 0) sh_new = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int joined = 0;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        if (__hip_atomic_load(step + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) continue;
        for (int k = rp[v]; k < rp[v + 1]; ++k) {
            const int u = ci[k];
            if (skip && skip[k]) continue;
            if (__hip_atomic_load(step + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cur) {
                state[v] = state[u] ^ 1;           // (state[u] was written by an earlier launch)
                __hip_atomic_store(step + v, cur + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ++joined;
                break;
            }
        }
    }
    if (joined) atomicAdd(&sh_new, joined);
    __syncthreads();
    if (threadIdx.x == 0 && sh_new) atomicAdd(visited, sh_new);
}

// One growth step, from the FRONT (the vertices that joined at step `cur`; at 256^3 a scan of all 16.8M vertices per step was 12.5 of
// the 15 ms of this ordering): k_region_push offers every unvisited neighbour of a front vertex u the parent u -- cand[v] = the
// smallest such u, which is the first neighbour of v's (ascending) row that joined at step cur, whatever order the lanes run in --
// and lists v once (the offer that finds cand[v] unset); k_region_settle then gives the listed vertices their region and the
// opposite parity of their parent.  front_n[cur & 3] = length of the front of step cur ([(cur + 2) & 3] is cleared on the way).
constexpr int kCandUnset = 0x7f7f7f7f;      // (hipMemset of 0x7f)
__global__ __launch_bounds__(kBlock) void k_region_push(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                        const uint8_t *__restrict__ skip, const int32_t *__restrict__ step,
                                                        int32_t *cand, const int32_t *__restrict__ front, int32_t *next, int *front_n,
                                                        int cur) {
    const int count = front_n[cur & 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) front_n[(cur + 2) & 3] = 0;
    const int stride = gridDim.x * kBlock, lane = threadIdx.x & 63;
    int *const next_n = front_n + ((cur + 1) & 3);
    for (int t0 = blockIdx.x * kBlock; t0 < count; t0 += stride) {      // (the same trip count for every lane of a wave)
        const int t = t0 + threadIdx.x;
        unsigned long long won = 0;          // the entries of u's row whose vertex this lane lists
        int r0 = 0;
        if (t < count) {
            const int u = front[t];
            r0 = rp[u];
            for (int k = r0; k < rp[u + 1]; ++k) {
                if (skip && skip[k]) continue;
                const int v = ci[k];
                if (step[v] >= 0) continue;
                if (atomicMin(cand + v, u) != kCandUnset) continue;
                if (k - r0 < 64) won |= 1ull << (k - r0);
                else next[atomicAdd(next_n, 1)] = v;
            }
        }
        // one counter update per wave (16.8M single-address atomics were most of a 256^3 growth)
        const int mine = __popcll(won);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(incl, off);
            if (lane >= off) incl += y;
        }
        const int total = __shfl(incl, 63);
        if (total == 0) continue;
        int base = 0;
        if (lane == 63) base = atomicAdd(next_n, total);
        int at = __shfl(base, 63) + incl - mine;
        while (won) {
            const int kb = __ffsll((long long)won) - 1;
            won &= won - 1;
            next[at++] = ci[r0 + kb];
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_region_settle(int32_t *step, int32_t *state, const int32_t *__restrict__ cand,
                                                          const int32_t *__restrict__ next, const int *__restrict__ front_n, int cur,
                                                          int *visited) {
    const int count = front_n[(cur + 1) & 3];
    if (blockIdx.x == 0 && threadIdx.x == 0 && count) atomicAdd(visited, count);
    const int stride = gridDim.x * kBlock;
    for (int t = blockIdx.x * kBlock + threadIdx.x; t < count; t += stride) {
        const int v = next[t];
        state[v] = state[cand[v]] ^ 1;           // (the parent's state was written by an earlier launch)
        step[v] = cur + 1;
    }
}

// rel[2 * (a * kRegions + b) + 0] counts the edges that join regions a and b with opposite parities (the two regions agree as they
// are), [.. + 1] those that join equal parities (one of the two has to be flipped); a bipartite graph fills one of the two only
__global__ __launch_bounds__(kBlock) void k_region_relations(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                             const uint8_t *__restrict__ skip, const int32_t *__restrict__ step,
                                                             const int32_t *__restrict__ state, unsigned int *rel, int kRegions) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        if (step[v] < 0) continue;
        const int sv = state[v];
        for (int k = rp[v]; k < rp[v + 1]; ++k) {
            const int u = ci[k];
            if (u <= v || (skip && skip[k]) || step[u] < 0) continue;
            const int su = state[u];
            if ((su >> 1) == (sv >> 1)) continue;   // inside a region: the final edge-by-edge check looks at those
            const int which = ((su ^ sv) & 1) ? 0 : 1;
            atomicAdd(rel + 2 * ((size_t)(sv >> 1) * kRegions + (su >> 1)) + which, 1u);
            atomicAdd(rel + 2 * ((size_t)(su >> 1) * kRegions + (sv >> 1)) + which, 1u);
        }
    }
}

// the related pairs of regions (a < b) as a list {a, b, opposite, equal}: a region touches a dozen others, so the host reads some
// 10K entries instead of the kRegions^2 matrix (8 MB and a million pairs to look at with 1024 regions)
__global__ __launch_bounds__(kBlock) void k_region_pairs(const unsigned int *__restrict__ rel, int kRegions, int4 *__restrict__ out,
                                                         int *count) {
    const int64_t total = (int64_t)kRegions * kRegions, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += stride) {
        const int a = (int)(idx / kRegions), b = (int)(idx % kRegions);
        if (a >= b) continue;
        const unsigned opposite = rel[2 * idx], equal = rel[2 * idx + 1];
        if (opposite | equal) out[atomicAdd(count, 1)] = make_int4(a, b, (int)opposite, (int)equal);
    }
}

// mask[v] = -1 (a candidate) for the vertices of the regions of component `comp`, 0 otherwise: k_min_degree's "unvisited" filter
__global__ __launch_bounds__(kBlock) void k_region_mask(int64_t n, const int32_t *__restrict__ step, const int32_t *__restrict__ state,
                                                        const int32_t *__restrict__ comp_of, int comp, int32_t *__restrict__ mask) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride)
        mask[v] = (step[v] >= 0 && comp_of[state[v] >> 1] == comp) ? -1 : 0;
}

// (a vertex no region reached -- only possible when edges were skipped -- stays uncoloured: -1)
__global__ __launch_bounds__(kBlock) void k_region_colors(int64_t n, const int32_t *__restrict__ step, const int32_t *__restrict__ state,
                                                          const int32_t *__restrict__ flip, int32_t *__restrict__ color) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride)
        color[v] = step[v] < 0 ? -1 : ((state[v] & 1) ^ flip[state[v] >> 1]);
}

// skip[k] = 1 for the entries (v, u) whose endpoints share a neighbour: the edge lies on a triangle.  A mesh graph that is bipartite
// but for refinement interfaces / a few extra couplings has its odd cycles there; without those edges what is left can be
// two-coloured, and the conflicts then sit on the skipped edges only.  Rows hold ascending columns: a merge per entry.
__global__ __launch_bounds__(kBlock) void k_triangle_edges(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                           uint8_t *__restrict__ skip) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        const int vs = rp[v], ve = rp[v + 1];
        for (int k = vs; k < ve; ++k) {
            const int u = ci[k];
            bool tri = false;
            if (u != v) {
                int a = vs, b = rp[u];
                const int be = rp[u + 1];
                while (a < ve && b < be) {
                    const int ca = ci[a], cb = ci[b];
                    if (ca == cb) {
                        if (ca != v && ca != u) { tri = true; break; }
                        ++a; ++b;
                    } else if (ca < cb) ++a;
                    else ++b;
                }
            }
            skip[k] = tri ? 1 : 0;
        }
    }
}

// How far from bipartite?  Of the rows v = 0, stride, 2 stride, ... : out[0] counts them, out[1] those with an entry on a triangle (a
// proven odd cycle through the vertex).  A Delaunay graph or a grid with a diagonal per cell: every vertex; a grid: none; a mesh
// with refinement interfaces or a few extra couplings: the vertices there.
__global__ __launch_bounds__(kBlock) void k_triangle_sample(int64_t n, int64_t stride_rows, const int32_t *__restrict__ rp,
                                                            const int32_t *__restrict__ ci, int *out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int seen = 0, touched = 0;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t * stride_rows < n; t += stride) {
        const int64_t v = t * stride_rows;
        const int vs = rp[v], ve = rp[v + 1];
        bool tri = false;
        for (int k = vs; k < ve && !tri; ++k) {
            const int u = ci[k];
            if (u == v) continue;
            int a = vs, b = rp[u];
            const int be = rp[u + 1];
            while (a < ve && b < be) {
                const int ca = ci[a], cb = ci[b];
                if (ca == cb) {
                    if (ca != v && ca != u) { tri = true; break; }
                    ++a; ++b;
                } else if (ca < cb) ++a;
                else ++b;
            }
        }
        ++seen;
        touched += tri ? 1 : 0;
    }
    if (seen) atomicAdd(out, seen);
    if (touched) atomicAdd(out + 1, touched);
}

// Repair of an almost proper two-colouring (a mesh that is bipartite but for a few odd cycles -- refinement interfaces, a handful
// of extra couplings): of every edge that joins two vertices of one colour the endpoint of larger index loses its colour
// (-1); Jones-Plassmann then colours those vertices around the ones that keep theirs.  *marked counts them.
__global__ __launch_bounds__(kBlock) void k_mark_conflicts(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                           const int32_t *__restrict__ color, int32_t *__restrict__ out, int *marked) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int mine = 0;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        const int cv = color[v];
        bool lose = cv < 0;
        for (int k = rp[v]; k < rp[v + 1]; ++k) {
            const int u = ci[k];
            lose = lose || (u < v && color[u] == cv);
        }
        out[v] = lose ? -1 : cv;
        mine += lose ? 1 : 0;
    }
    if (mine) atomicAdd(marked, mine);
}

__global__ __launch_bounds__(kBlock) void k_invert_perm(int64_t n, const int32_t *__restrict__ perm, int32_t *__restrict__ iperm) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) iperm[perm[r]] = (int32_t)r;
}
}  // namespace

// Two colours through regions (see k_region_seed).  *colored = true when `color` holds a proper 2-colouring in which the vertex
// of smallest (degree, index) of every connected component has colour 0 -- what the search-by-search parity colouring below
// produces too, so the two agree vertex by vertex.  false (nothing to report): an odd cycle, a component without a seed, more
// than a handful of components -- the caller goes on with the other methods.
// *nearly (may be null): when the graph has odd cycles the walk still hands out flips (first relation seen wins) and `color` holds
// an IMPROPER two-colouring whose conflicts the caller may repair (repair_two_coloring) -- *nearly = true then.
// skip (may be null, with nearly): entries the regions do not grow along nor relate through (k_triangle_edges); vertices they then
// cannot reach stay uncoloured.
static int two_colors_by_regions(const CsrDev &A, int32_t *color, int *flags, bool *colored, bool *nearly, const uint8_t *skip,
                                 hipStream_t s, int32_t *const scratch[4] /* four vectors of n entries the caller does not need yet */) {
    const int64_t n = A.n;
    const int kRegions = n >= (1 << 21) ? kRegionsLarge : kRegionsSmall;
    *colored = false;
    if (nearly) *nearly = false;
    bool odd_cycle = false;
    if (n < 4 * kRegions) return DPCG_OK;
    struct View { int32_t *p; } step{scratch[0]}, state{scratch[1]}, deg{scratch[2]}, mask{scratch[3]};   // (hipMalloc of 4 x 67 MB at 256^3: ~8 ms)
    Buf<int32_t> d_small;
    Buf<unsigned int> rel;
    Buf<unsigned long long> best;
    DPCG_TRY(d_small.alloc(2 * kRegions)); DPCG_TRY(rel.alloc(2 * (int64_t)kRegions * kRegions)); DPCG_TRY(best.alloc(1));
    PhaseTimer pt(s);
    DPCG_HIP(hipMemsetAsync(step.p, 0xff, (size_t)n * sizeof(int32_t), s));
    DPCG_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(int), s));
    // during the growth: cand in `deg`, the two fronts in `mask` and in `color` (all three are written afresh afterwards)
    int32_t *const cand = deg.p, *const front[2] = {mask.p, color};
    Buf<int> front_n;
    DPCG_TRY(front_n.alloc(4));
    DPCG_HIP(hipMemsetAsync(front_n.p, 0, 4 * sizeof(int), s));
    DPCG_HIP(hipMemsetAsync(cand, 0x7f, (size_t)n * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_region_seed, dim3((kRegions + 63) / 64), dim3(64), 0, s, n, step.p, state.p, flags, kRegions, front[0], front_n.p);
    int visited = 0, cur = 0;
    const int grow_grid = rows_grid(n / 4 + 1, 2048);
    const bool from_front = n >= (1 << 21);       // (the same regions either way: measured 256^2 0.39 -> 0.57 ms from the front, 256^3 12.5 -> 8.2)
    for (;;) {
        for (int b = 0; b < kRegionBatch; ++b, ++cur) {
            if (!from_front) {
                hipLaunchKernelGGL(k_region_grow, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, skip, step.p, state.p, cur,
                                   flags);
                continue;
            }
            hipLaunchKernelGGL(k_region_push, dim3(grow_grid), dim3(kBlock), 0, s, A.rowptr, A.col, skip, step.p, cand, front[cur & 1],
                               front[(cur + 1) & 1], front_n.p, cur);
            hipLaunchKernelGGL(k_region_settle, dim3(grow_grid), dim3(kBlock), 0, s, step.p, state.p, cand, front[(cur + 1) & 1], front_n.p, cur,
                               flags);
        }
        int now = 0;
        DPCG_HIP(hipMemcpyAsync(&now, flags, sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (now == n) break;
        if (now == visited) {                        // a component without a seed
            if (skip && nearly && (int64_t)(n - now) * 8 <= n) break;     // (skipped edges: a few unreachable vertices are repaired later)
            if (pt.on) fprintf(stderr, "[dpcg setup]   regions: %lld of %lld vertices not reached after %d steps\n", (long long)(n - now), (long long)n, cur);
            return DPCG_OK;
        }
        visited = now;
    }
    pt.mark("  regions: growth");
    DPCG_HIP(hipMemsetAsync(rel.p, 0, 2 * (size_t)kRegions * kRegions * sizeof(unsigned int), s));
    hipLaunchKernelGGL(k_region_relations, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, skip, step.p, state.p, rel.p, kRegions);
    Buf<int4> pairs;
    Buf<int> n_pairs;
    DPCG_TRY(pairs.alloc((int64_t)kRegions * kRegions / 2));
    DPCG_TRY(n_pairs.alloc(1));
    DPCG_HIP(hipMemsetAsync(n_pairs.p, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_region_pairs, dim3(rows_grid((int64_t)kRegions * kRegions, 1024)), dim3(kBlock), 0, s, rel.p, kRegions, pairs.p, n_pairs.p);
    int h_n_pairs = 0;
    DPCG_HIP(hipMemcpyAsync(&h_n_pairs, n_pairs.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    std::vector<int4> h_pairs((size_t)h_n_pairs);
    if (h_n_pairs) DPCG_HIP(hipMemcpyAsync(h_pairs.data(), pairs.p, h_pairs.size() * sizeof(int4), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    // every region's neighbours in ascending order (the list arrives in the order of the atomics: sorted here, so that the walk below
    // -- and with it, for a graph with odd cycles, which relation is seen first -- is the same from run to run)
    struct Rel { int b; unsigned opposite, equal; };
    std::vector<std::vector<Rel>> adj((size_t)kRegions);
    for (const int4 &q : h_pairs) {
        adj[(size_t)q.x].push_back(Rel{q.y, (unsigned)q.z, (unsigned)q.w});
        adj[(size_t)q.y].push_back(Rel{q.x, (unsigned)q.z, (unsigned)q.w});
    }
    for (auto &list : adj) std::sort(list.begin(), list.end(), [](const Rel &x, const Rel &y) { return x.b < y.b; });
    // walk the graph of regions: flip[b] relative to the first region of its component
    std::vector<int32_t> flip((size_t)kRegions, 0), comp((size_t)kRegions, -1), stack;
    int n_comp = 0;
    for (int r0 = 0; r0 < kRegions; ++r0) {
        if (comp[(size_t)r0] >= 0) continue;
        comp[(size_t)r0] = n_comp;
        stack.assign(1, r0);
        while (!stack.empty()) {
            const int a = stack.back();
            stack.pop_back();
            for (const Rel &nb : adj[(size_t)a]) {
                const int b = nb.b;
                const unsigned opposite = nb.opposite, equal = nb.equal;
                const unsigned bits = (opposite ? 1u : 0u) | (equal ? 2u : 0u);
                if (bits == 3u) {                                                 // an odd cycle through the two regions
                    if (!nearly) return DPCG_OK;
                    odd_cycle = true;
                }
                const int want = flip[(size_t)a] ^ (equal > opposite ? 1 : 0);    // (nearly bipartite: the majority decides)
                if (comp[(size_t)b] < 0) {
                    comp[(size_t)b] = n_comp;
                    flip[(size_t)b] = want;
                    stack.push_back(b);
                } else if (flip[(size_t)b] != want && bits != 3u) {
                    if (!nearly) return DPCG_OK;                                  // an odd cycle through several regions
                    odd_cycle = true;
                }
            }
        }
        ++n_comp;
    }
    // (an empty region -- its seed coincided with another one -- is a component of its own without vertices: harmless)
    std::vector<int32_t> h_small((size_t)2 * kRegions);
    for (int r = 0; r < kRegions; ++r) h_small[(size_t)r] = comp[(size_t)r];
    DPCG_HIP(hipMemcpyAsync(d_small.p, h_small.data(), kRegions * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_degrees, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, A.rowptr, deg.p);
    int real_comps = 0;
    for (int c = 0; c < n_comp; ++c) {
        // the component's vertex of smallest (degree, index) gets colour 0
        DPCG_HIP(hipMemsetAsync(best.p, 0xff, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_region_mask, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, step.p, state.p, d_small.p, c, mask.p);
        hipLaunchKernelGGL(k_min_degree, dim3(rows_grid(n, 1024)), dim3(kBlock), 0, s, (const int32_t *)nullptr, (int64_t)0, n, deg.p, mask.p, 1,
                           best.p);
        unsigned long long hb = 0;
        DPCG_HIP(hipMemcpyAsync(&hb, best.p, sizeof(hb), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (hb == ~0ull) continue;                                                 // an empty region
        if (++real_comps > 8) {                                                    // many components: the searches below handle them
            if (pt.on) fprintf(stderr, "[dpcg setup]   regions: more than 8 components\n");
            return DPCG_OK;
        }
        int32_t sv = 0;
        DPCG_HIP(hipMemcpyAsync(&sv, state.p + (hb & 0xffffffffu), sizeof(int32_t), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (((sv & 1) ^ flip[(size_t)(sv >> 1)]) != 0)
            for (int r = 0; r < kRegions; ++r)
                if (comp[(size_t)r] == c) flip[(size_t)r] ^= 1;
    }
    DPCG_HIP(hipMemcpyAsync(d_small.p + kRegions, flip.data(), kRegions * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_region_colors, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, step.p, state.p, d_small.p + kRegions, color);
    DPCG_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(k_check_coloring, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, A.rowptr, A.col, color, flags);
    int bad = 0;
    DPCG_HIP(hipMemcpyAsync(&bad, flags, sizeof(int), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    *colored = bad == 0;
    if (nearly) *nearly = bad != 0;
    if (pt.on) fprintf(stderr, "[dpcg setup]   regions: %d components, odd cycle %d, proper %d\n", real_comps, (int)odd_cycle, (int)*colored);
    (void)odd_cycle;
    pt.mark("  regions: relations, flips, check");
    return DPCG_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Region-by-region numbering: a CHEAP locality order for meshes whose graph has a long diameter.  Reverse Cuthill-McKee walks the
// levels of ONE breadth-first search -- ~2000 dependent levels on a 1M-cell 2-D mesh, 49 ms -- where `regions` searches started at
// once meet after a few dozen steps.  Vertices are numbered region by region (the regions in breadth-first order of the graph of
// regions, so that neighbouring regions lie near each other), inside a region by the step at which they joined (ring by ring
// around the seed) and in the caller's order inside a ring.  A 256-row block is then a few rings of one region and its columns
// are those rings, their neighbours and pieces of the adjacent regions' outer rings: the 1M-row quadtree mesh in OpenFOAM's
// numbering (refinement appends cells: 12 % of its row blocks touch 41-51 chunks of x, the x-tile plan refuses) comes out at
// 15 chunks a block on average and 38 at most with 2048 regions -- what RCM gives (15 / 20) at a fraction of its price.
// Deterministic: a vertex joins the region of the first neighbour of its row that joined a step earlier (k_region_grow) or of the
// smallest such neighbour (k_region_push / k_region_settle: the same vertex, rows hold ascending columns); the sort is stable.
// perm[new] = old, iperm[old] = new (device, owned by the caller afterwards); vertices no region reached keep their order at the end.
namespace {
// seeds for the numbering: as k_region_seed, but where two seeds fall on one vertex the SMALLER region takes it whatever order the
// threads run in (state arrives filled with 0x7f: the numbering must be the same from run to run, a colouring need not care)
__global__ void k_region_seed_min(int64_t n, int32_t *step, int32_t *state, int *visited, int regions, int32_t *front, int *front_n) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= regions) return;
    const int64_t v = (int64_t)(((unsigned long long)jp_priority(r + 1) * (unsigned long long)n) >> 32);
    if (atomicCAS(step + v, -1, 0) == -1) {
        atomicAdd(visited, 1);
        front[atomicAdd(front_n, 1)] = (int32_t)v;
    }
    atomicMin(state + v, r << 1);
}

__global__ __launch_bounds__(kBlock) void k_region_keys(int64_t n, const int32_t *__restrict__ step, const int32_t *__restrict__ state,
                                                        const int32_t *__restrict__ rank, int regions, uint32_t *__restrict__ key,
                                                        int32_t *__restrict__ vertex) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n; v += stride) {
        const int st = step[v];
        key[v] = st < 0 ? (uint32_t)regions << 8 : ((uint32_t)rank[state[v] >> 1] << 8 | (uint32_t)(st < 255 ? st : 255));
        vertex[v] = (int32_t)v;
    }
}
}  // namespace

int region_order_default_regions(int64_t n) {
    int regions = 256;
    while (regions < 4096 && (int64_t)regions * 512 < n) regions *= 2;       // ~512 vertices a region: two row blocks
    while (regions > 1 && (int64_t)regions * 16 > n) regions /= 2;           // (small systems: at least 16)
    return regions;
}

int region_order(const CsrDev &A, int regions, int32_t **perm_out, int32_t **iperm_out, hipStream_t s) {
    const int64_t n = A.n;
    if (regions < 1 || regions > 4096 || n < regions) return invalid("region_order: 1 .. 4096 regions, not more than rows");
    PhaseTimer pt(s);
    Buf<int32_t> step, state, cand, front0, front1, rank, vertex, perm, iperm;
    Buf<uint32_t> key, key_sorted;
    Buf<int> flags, front_n;
    DPCG_TRY(step.alloc(n)); DPCG_TRY(state.alloc(n)); DPCG_TRY(flags.alloc(4)); DPCG_TRY(front_n.alloc(4)); DPCG_TRY(rank.alloc(regions));
    DPCG_HIP(hipMemsetAsync(step.p, 0xff, (size_t)n * sizeof(int32_t), s));
    DPCG_HIP(hipMemsetAsync(flags.p, 0, 4 * sizeof(int), s));
    DPCG_HIP(hipMemsetAsync(front_n.p, 0, 4 * sizeof(int), s));
    const bool from_front = n >= (1 << 21);              // (as two_colors_by_regions: a scan of every vertex per step below that)
    DPCG_TRY(front0.alloc(from_front ? n : regions));
    if (from_front) {
        DPCG_TRY(front1.alloc(n)); DPCG_TRY(cand.alloc(n));
        DPCG_HIP(hipMemsetAsync(cand.p, 0x7f, (size_t)n * sizeof(int32_t), s));
    }
    DPCG_HIP(hipMemsetAsync(state.p, 0x7f, (size_t)n * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_region_seed_min, dim3((regions + 63) / 64), dim3(64), 0, s, n, step.p, state.p, flags.p, regions, front0.p, front_n.p);
    int32_t *const front[2] = {front0.p, front1.p};
    const int grow_grid = rows_grid(n / 4 + 1, 2048);
    int visited = 0, cur = 0;
    for (;;) {
        for (int b = 0; b < kRegionBatch; ++b, ++cur) {
            if (!from_front) {
                hipLaunchKernelGGL(k_region_grow, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, (const uint8_t *)nullptr,
                                   step.p, state.p, cur, flags.p);
                continue;
            }
            hipLaunchKernelGGL(k_region_push, dim3(grow_grid), dim3(kBlock), 0, s, A.rowptr, A.col, (const uint8_t *)nullptr, step.p, cand.p,
                               front[cur & 1], front[(cur + 1) & 1], front_n.p, cur);
            hipLaunchKernelGGL(k_region_settle, dim3(grow_grid), dim3(kBlock), 0, s, step.p, state.p, cand.p, front[(cur + 1) & 1], front_n.p,
                               cur, flags.p);
        }
        int now = 0;
        DPCG_HIP(hipMemcpyAsync(&now, flags.p, sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (now == n || now == visited) { visited = now; break; }     // (stalled: components without a seed stay at the end)
        visited = now;
    }
    pt.mark("  regions: growth");
    // the graph of regions, and its breadth-first order (neighbours ascending: the same from run to run)
    std::vector<int32_t> h_rank((size_t)regions, -1);
    {
        Buf<unsigned int> rel;
        Buf<int4> pairs;
        Buf<int> n_pairs;
        DPCG_TRY(rel.alloc(2 * (int64_t)regions * regions)); DPCG_TRY(pairs.alloc((int64_t)regions * regions / 2 + 1)); DPCG_TRY(n_pairs.alloc(1));
        DPCG_HIP(hipMemsetAsync(rel.p, 0, 2 * (size_t)regions * regions * sizeof(unsigned int), s));
        DPCG_HIP(hipMemsetAsync(n_pairs.p, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_region_relations, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, (const uint8_t *)nullptr, step.p,
                           state.p, rel.p, regions);
        hipLaunchKernelGGL(k_region_pairs, dim3(rows_grid((int64_t)regions * regions, 1024)), dim3(kBlock), 0, s, rel.p, regions, pairs.p,
                           n_pairs.p);
        int h_n_pairs = 0;
        DPCG_HIP(hipMemcpyAsync(&h_n_pairs, n_pairs.p, sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        std::vector<int4> h_pairs((size_t)h_n_pairs);
        if (h_n_pairs) DPCG_HIP(hipMemcpyAsync(h_pairs.data(), pairs.p, h_pairs.size() * sizeof(int4), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        std::vector<std::vector<int>> adj((size_t)regions);
        for (const int4 &q : h_pairs) {
            adj[(size_t)q.x].push_back(q.y);
            adj[(size_t)q.y].push_back(q.x);
        }
        for (auto &list : adj) std::sort(list.begin(), list.end());
        std::vector<int> queue;
        queue.reserve((size_t)regions);
        for (int r0 = 0; r0 < regions; ++r0) {
            if (h_rank[(size_t)r0] >= 0) continue;
            h_rank[(size_t)r0] = (int32_t)queue.size();
            queue.push_back(r0);
            for (size_t head = queue.size() - 1; head < queue.size(); ++head)
                for (const int b : adj[(size_t)queue[head]])
                    if (h_rank[(size_t)b] < 0) {
                        h_rank[(size_t)b] = (int32_t)queue.size();
                        queue.push_back(b);
                    }
        }
    }
    DPCG_HIP(hipMemcpyAsync(rank.p, h_rank.data(), (size_t)regions * sizeof(int32_t), hipMemcpyHostToDevice, s));
    pt.mark("  regions: graph of regions");
    DPCG_TRY(key.alloc(n)); DPCG_TRY(key_sorted.alloc(n)); DPCG_TRY(vertex.alloc(n)); DPCG_TRY(perm.alloc(n)); DPCG_TRY(iperm.alloc(n));
    hipLaunchKernelGGL(k_region_keys, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, step.p, state.p, rank.p, regions, key.p, vertex.p);
    int bits = 9;
    while ((1 << (bits - 8)) <= regions) ++bits;
    DPCG_TRY(sort_pairs_u32_i32(key.p, key_sorted.p, vertex.p, perm.p, n, bits, s));      // (returns with the stream idle: h_rank may go)
    hipLaunchKernelGGL(k_invert_perm, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, perm.p, iperm.p);
    DPCG_CHECK_LAUNCH();
    pt.mark("  regions: keys, sort");
    if (pt.on) fprintf(stderr, "[dpcg setup]   regions: %d regions, %d growth steps, %lld of %lld vertices reached\n", regions, cur, (long long)visited, (long long)n);
    *perm_out = perm.release();
    *iperm_out = iperm.release();
    return DPCG_OK;
}

int multicolor_order(const CsrDev &A, int32_t **perm_out, int32_t **iperm_out, int *n_colors, hipStream_t s) {
    const int64_t n = A.n;
    if (n + 2 + kBfsBatch > 2147483000LL) return invalid("multicolour ordering: system too large");
    Buf<int32_t> color, perm, iperm, iota;
    Buf<uint32_t> key_sorted;
    Buf<int> flags;
    DPCG_TRY(color.alloc(n)); DPCG_TRY(perm.alloc(n)); DPCG_TRY(iperm.alloc(n)); DPCG_TRY(iota.alloc(n));
    DPCG_TRY(key_sorted.alloc(n)); DPCG_TRY(flags.alloc(4));
    bool colored = false;
    PhaseTimer pt(s);
    static const bool regions_on = [] { const char *e = getenv("DPCG_COLOR_REGIONS"); return !(e && e[0] == '0'); }();
    bool nearly = false, repaired = false;
    static const bool repair_on = [] { const char *e = getenv("DPCG_COLOR_REPAIR"); return !(e && e[0] == '0'); }();
    int32_t *const scratch[4] = {perm.p, iperm.p, iota.p, reinterpret_cast<int32_t *>(key_sorted.p)};     // (free until the sort by colour)
    // -1. how far from bipartite is the graph?  Triangles through a sample of the vertices (every 64th; all of a small graph).  More
    // than an eighth of them on a triangle (a Delaunay graph, a grid with diagonals: all of them): no two big classes to be had -- the
    // two-colouring attempts below would each walk the whole graph to find that out (1M-row meshes: 45 of the 49 ms of this
    // function) -- straight to the greedy colouring.  Some, but few: the graph has odd cycles, so the plain regions (0.) are skipped
    // and the regions without the triangle edges (0b.) tried at once.  None: as a grid.
    bool far = false, some = false;
    if (regions_on) {
        const int64_t stride_rows = n >= 4096 ? 64 : 1;
        DPCG_HIP(hipMemsetAsync(flags.p, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(k_triangle_sample, dim3(rows_grid((n + stride_rows - 1) / stride_rows, 256)), dim3(kBlock), 0, s, n, stride_rows,
                           A.rowptr, A.col, flags.p);
        int h_tri[2] = {0, 0};
        DPCG_HIP(hipMemcpyAsync(h_tri, flags.p, sizeof(h_tri), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        some = h_tri[1] > 0;
        far = (int64_t)h_tri[1] * 8 > (int64_t)h_tri[0];
        if (pt.on) fprintf(stderr, "[dpcg setup]   %d of %d sampled vertices on a triangle\n", h_tri[1], h_tri[0]);
        pt.mark("  triangle sample");
    }
    bool odd_proven = some;
    if (regions_on && !some) DPCG_TRY(two_colors_by_regions(A, color.p, flags.p, &colored, repair_on ? &nearly : nullptr, nullptr, s, scratch));   // 0. two colours, many searches at once
    odd_proven = odd_proven || nearly;      // (every vertex reached and the colouring improper: an odd cycle)
    if (regions_on && repair_on && some && !far) nearly = true;
    if (!colored && nearly) {
        // 0b. nearly bipartite (odd cycles, but few): the regions once more without the edges that lie on triangles -- breadth-first
        // parity follows every shortcut, an extra coupling would flip the cone of vertices behind it -- then the two big classes stay
        // and the few vertices on conflicting edges are recoloured
        Buf<uint8_t> skip;
        DPCG_TRY(skip.alloc(A.nnz));
        hipLaunchKernelGGL(k_triangle_edges, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, skip.p);
        nearly = false;
        DPCG_TRY(two_colors_by_regions(A, color.p, flags.p, &colored, &nearly, skip.p, s, scratch));
        pt.mark("  regions without triangle edges");
    }
    if (!colored && nearly) {
        DPCG_HIP(hipMemsetAsync(flags.p, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(k_mark_conflicts, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, color.p, iota.p, flags.p + 2);
        int marked = 0;
        DPCG_HIP(hipMemcpyAsync(&marked, flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (pt.on) fprintf(stderr, "[dpcg setup]   nearly two-coloured: %d of %lld vertices on conflicting edges\n", marked, (long long)n);
        if ((int64_t)marked * 8 <= n) {
            DPCG_HIP(hipMemcpyAsync(color.p, iota.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            repaired = true;
        }
    }
    if (!colored && !repaired && !odd_proven) {   // 1. breadth-first parity, component by component (pointless once an odd cycle is known)
        Buf<int32_t> deg, level, order, start, count;
        Buf<unsigned long long> best;
        const int64_t nlv = n + 2 + 2 * kBfsBatch;
        DPCG_TRY(deg.alloc(n)); DPCG_TRY(level.alloc(n)); DPCG_TRY(order.alloc(n + 1)); DPCG_TRY(start.alloc(nlv));
        DPCG_TRY(count.alloc(nlv)); DPCG_TRY(best.alloc(1));
        hipLaunchKernelGGL(k_degrees, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, A.rowptr, deg.p);
        DPCG_HIP(hipMemsetAsync(level.p, 0xff, (size_t)n * sizeof(int32_t), s));
        DPCG_HIP(hipMemsetAsync(count.p, 0, (size_t)nlv * sizeof(int32_t), s));
        std::vector<int32_t> h_start, h_count;
        int visited = 0, next_level = 0, searches = 0;
        while (visited < n && searches < kMaxComponents) {
            DPCG_HIP(hipMemsetAsync(best.p, 0xff, sizeof(unsigned long long), s));
            hipLaunchKernelGGL(k_min_degree, dim3(rows_grid(n, 1024)), dim3(kBlock), 0, s, (const int32_t *)nullptr, (int64_t)0, n,
                               deg.p, level.p, 1, best.p);
            unsigned long long hb = 0;
            DPCG_HIP(hipMemcpyAsync(&hb, best.p, sizeof(hb), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            if (hb == ~0ull) break;
            int nl = 0;
            // every search starts at an EVEN level number, so that parity means the same in every component
            next_level += next_level & 1;
            DPCG_TRY(bfs(A, (int)(hb & 0xffffffffu), next_level, visited, level.p, order.p, start.p, count.p, h_start, h_count, &nl, s));
            for (int l = 0; l < nl; ++l) visited += h_count[(size_t)l];
            next_level += nl;
            ++searches;
        }
        if (visited == n) {
            hipLaunchKernelGGL(k_parity_colors, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, level.p, color.p);
            DPCG_HIP(hipMemsetAsync(flags.p, 0, 4 * sizeof(int), s));
            hipLaunchKernelGGL(k_check_coloring, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, A.rowptr, A.col, color.p, flags.p);
            int bad = 0;
            DPCG_HIP(hipMemcpyAsync(&bad, flags.p, sizeof(int), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            colored = bad == 0;
        }
        pt.mark("  breadth-first parity");
    }
    if (!colored) {   // 2. Jones-Plassmann greedy colouring (of everything, or of the vertices the repair left uncoloured)
        if (!repaired) DPCG_HIP(hipMemsetAsync(color.p, 0xff, (size_t)n * sizeof(int32_t), s));
        DPCG_HIP(hipMemsetAsync(flags.p, 0, 4 * sizeof(int), s));
        // peel ranks for the greedy order (DPCG_COLOR_PEEL=0: the hashed order alone, as in round 3; changes the ORDERING)
        Buf<int32_t> prank, pdeg;
        Buf<int> pctl;
        static const int peel_rounds = [] { const char *e = getenv("DPCG_COLOR_PEEL"); return e ? atoi(e) : 24; }();
        const int32_t *rank_dev = nullptr;
        if (!repaired && peel_rounds > 0) {
            DPCG_TRY(prank.alloc(n));
            DPCG_TRY(pdeg.alloc(n));
            DPCG_TRY(pctl.alloc(1));
            const int T = std::max(2, (int)(((A.nnz - n) + n / 2) / std::max<int64_t>(n, 1)) - 2);      // average degree (rounded) - 2
            DPCG_HIP(hipMemsetAsync(pctl.p, 0x7f, sizeof(int), s));
            hipLaunchKernelGGL(k_peel_init, dim3(rows_grid(n, 1024)), dim3(kBlock), 0, s, n, A.rowptr, A.col, prank.p, pdeg.p, peel_rounds, pctl.p);
            for (int r = 0; r < peel_rounds; ++r) {
                hipLaunchKernelGGL(k_peel_mark, dim3(rows_grid(n, 1024)), dim3(kBlock), 0, s, n, pdeg.p, prank.p, r, peel_rounds, T, pctl.p);
                DPCG_HIP(hipMemsetAsync(pctl.p, 0x7f, sizeof(int), s));
                hipLaunchKernelGGL(k_peel_degrees, dim3(rows_grid(n, 1024)), dim3(kBlock), 0, s, n, A.rowptr, A.col, prank.p, peel_rounds, pdeg.p, pctl.p);
            }
            rank_dev = prank.p;
            pt.mark("  peel ranks");
        }
        int rounds = 0;
        for (;;) {
            int h[2] = {0, 0};
            for (int b = 0; b < 4; ++b) {
                DPCG_HIP(hipMemsetAsync(flags.p, 0, sizeof(int), s));            // [0] remaining after this round, [1] error
                hipLaunchKernelGGL(k_jp_round, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, color.p, flags.p,
                                   flags.p + 1, rank_dev);
                ++rounds;
            }
            DPCG_HIP(hipMemcpyAsync(h, flags.p, sizeof(h), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            if (h[1]) return invalid("multicolour ordering: a vertex needs more than 63 colours");
            if (h[0] == 0) break;
            if (rounds > 4096) return invalid("multicolour ordering: the colouring did not finish");
        }
        hipLaunchKernelGGL(k_check_coloring, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, A.rowptr, A.col, color.p, flags.p + 2);
        int bad = 0;
        DPCG_HIP(hipMemcpyAsync(&bad, flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, s));
        DPCG_HIP(hipStreamSynchronize(s));
        if (bad) return invalid("multicolour ordering: improper colouring (is the pattern structurally symmetric?)");
        if (pt.on) fprintf(stderr, "[dpcg setup]   greedy colouring: %d rounds\n", rounds);
        pt.mark("  greedy rounds");
        // fewer colours = fewer levels for the triangular solves: a few passes of iterated greedy over the classes, last to first
        // (not after a repair: the two big classes are what the repair is for)
        static const int passes_knob = [] { const char *e = getenv("DPCG_RECOLOR_PASSES"); return e ? atoi(e) : 3; }();
        const int passes = repaired ? 1 : passes_knob;
        Buf<int> hist;
        Buf<int32_t> map;
        DPCG_TRY(hist.alloc(64));
        DPCG_TRY(map.alloc(64));
        for (int pass = 0; pass < passes; ++pass) {
            DPCG_HIP(hipMemsetAsync(hist.p, 0, 64 * sizeof(int), s));
            hipLaunchKernelGGL(k_color_histogram, dim3(rows_grid(n, 1024)), dim3(kBlock), 0, s, n, color.p, hist.p);
            int h_hist[64];
            DPCG_HIP(hipMemcpyAsync(h_hist, hist.p, sizeof(h_hist), hipMemcpyDeviceToHost, s));
            DPCG_HIP(hipStreamSynchronize(s));
            // empty classes are squeezed out, the others keep their order
            int32_t h_map[64];
            int used = 0;
            for (int c = 0; c < 64; ++c) h_map[c] = h_hist[c] > 0 ? used++ : 0;
            DPCG_HIP(hipMemcpyAsync(map.p, h_map, sizeof(h_map), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_remap_colors, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, color.p, map.p);
            DPCG_HIP(hipStreamSynchronize(s));      // (h_map is on the stack)
            if (used <= 2 || pass == passes - 1) {
                // a last class of a handful of vertices is folded into the others where a local move does (never more colours)
                static const bool fold_on = [] { const char *e = getenv("DPCG_COLOR_FOLD"); return !(e && e[0] == '0'); }();
                if (fold_on && used > 2 && h_hist_last_nonempty(h_hist) <= kFoldMax) {
                    Buf<int> members;
                    DPCG_TRY(members.alloc(kFoldMax + 2));
                    DPCG_HIP(hipMemsetAsync(members.p, 0, sizeof(int), s));
                    hipLaunchKernelGGL(k_collect_class, dim3(rows_grid(n, 1024)), dim3(kBlock), 0, s, n, color.p, used - 1, members.p, members.p + 1);
                    hipLaunchKernelGGL(k_fold_tiny_class, dim3(1), dim3(64), 0, s, n, A.rowptr, A.col, color.p, used - 1, members.p,
                                       members.p + 1, members.p + kFoldMax + 1);
                    DPCG_HIP(hipMemsetAsync(flags.p + 2, 0, sizeof(int), s));
                    hipLaunchKernelGGL(k_check_coloring, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, A.rowptr, A.col, color.p, flags.p + 2);
                    int bad_fold = 0;
                    DPCG_HIP(hipMemcpyAsync(&bad_fold, flags.p + 2, sizeof(int), hipMemcpyDeviceToHost, s));
                    DPCG_HIP(hipStreamSynchronize(s));
                    if (bad_fold) return invalid("multicolour ordering: folding the last class left an improper colouring");
                }
                break;
            }
            DPCG_HIP(hipMemsetAsync(iota.p, 0xff, (size_t)n * sizeof(int32_t), s));          // (iota: free until the sort below)
            for (int c = used - 1; c >= 0; --c)
                hipLaunchKernelGGL(k_recolor_class, dim3(rows_grid(n, 2048)), dim3(kBlock), 0, s, n, A.rowptr, A.col, color.p, iota.p, c);
            DPCG_HIP(hipMemcpyAsync(color.p, iota.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        }
    }
    int32_t cmax = 0;
    DPCG_TRY(reduce_max_i32(color.p, reinterpret_cast<int32_t *>(flags.p + 3), n, s));
    DPCG_HIP(hipMemcpyAsync(&cmax, flags.p + 3, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DPCG_HIP(hipStreamSynchronize(s));
    pt.mark("  colours");
    launch_iota(n, iota.p, s);
    // stable sort by colour: ascending vertex index inside a colour
    DPCG_TRY(sort_pairs_u32_i32(reinterpret_cast<const uint32_t *>(color.p), key_sorted.p, iota.p, perm.p, n, bits_for((uint64_t)cmax), s));
    hipLaunchKernelGGL(k_invert_perm, dim3(rows_grid(n)), dim3(kBlock), 0, s, n, perm.p, iperm.p);
    DPCG_HIP(hipStreamSynchronize(s));
    DPCG_CHECK_LAUNCH();
    pt.mark("  sort by colour");
    if (n_colors) *n_colors = cmax + 1;
    *perm_out = perm.release();
    *iperm_out = iperm.release();
    return DPCG_OK;
}

}  // namespace dpcg
