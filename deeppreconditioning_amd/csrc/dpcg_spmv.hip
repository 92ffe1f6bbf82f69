// SpMV kernels of libdpcg.so -- hand-written HIP for gfx950 (MI355X, CDNA4, wave64): CSR-stream (gather), CSR-vector
// and x-tile kernels, the tile plan, and their dispatch.  HBM-bandwidth work: coalesced streams, LDS staging, enough
// workgroups in flight; no MFMA (there is no dense contraction).
// Compiled with -ffp-contract=off so that a*b+c is two roundings, as in the CPU reference path (scipy/ATen CSR row
// sums, unfused torch mul+add at cg.py:79-83); in-order sums then reproduce the oracle bit for bit.
#include <algorithm>
#include <type_traits>

#include "dpcg_device.h"

namespace dpcg {

// ------------------------------------------------------------------------------------------------
// CSR-stream SpMV (rows with few non-zeros: 5/7-point stencils, OpenFOAM-like matrices).
//
// A workgroup takes 256 consecutive rows.  Their val[]/col[] segment is contiguous in CSR, so the
// 256 threads stream it with fully coalesced loads (lane i <-> non-zero base+i), multiply by the
// gathered x[col] (served by L1/L2: neighbouring rows share columns) and park the products in LDS.
// After a barrier thread i adds up the products of row i IN COLUMN ORDER -- the same order and
// rounding as a sequential CPU CSR row sum.  LDS reads are conflict-free for odd row lengths
// (stride 5 or 7 doubles over 32 lanes).  Algorithmic bytes: nnz*(wv+4) + (n+1)*4 + 2*n*wx.
// ------------------------------------------------------------------------------------------------
template <typename VT, typename XT, bool CTL, bool DOT, typename YT, bool FUSE = false>
__global__ __launch_bounds__(kBlock) void k_spmv_stream(int64_t n, const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col,
                                                        const VT *__restrict__ val, const XT *__restrict__ x,
                                                        const double *__restrict__ xdot, YT *__restrict__ y,
                                                        int nrb, double *__restrict__ part_pq, IterCtlDev ctl,
                                                        FuseArgs fa) {
    constexpr int U = kStreamCap / kBlock;  // non-zeros per thread per row-block, read as U/2 aligned pairs
    __shared__ __attribute__((aligned(16))) double prod[kStreamCap + 2];
    __shared__ double sh[8];
    const int t = threadIdx.x;
    // contiguous ranges of row-blocks per (virtual) workgroup, the remainder spread evenly over the
    // grid (so every XCD slab carries the same load); 32-bit scalar arithmetic only
    const int v = virtual_block();
    int rb_lo, rb_hi;
    split_range(nrb, v, rb_lo, rb_hi);
    int c[U];
    VT a[U];
    int cnt = 0, base = 0, rs = 0, re = 0;   // rs/re stay absolute until the row-sum phase (no early wait)
    // Matrix stream of one row-block -> registers, all loads of a thread in flight at once.  The slice is read as
    // aligned PAIRS of consecutive non-zeros, lane i <-> pair i (16-byte value loads, 8-byte column loads: the
    // texture-address unit issues per instruction, not per byte).  Element 2u+h of a thread is slot 2 (t + 256 u) + h,
    // slots counted from the even index at or below the block's first non-zero: slot 0 may belong to the previous
    // block and the slot after the last non-zero to the next one or to nobody -- their columns are forced to 0 so
    // that nothing is gathered out of range, and their products are never read.
    typedef VT VPair __attribute__((ext_vector_type(2)));
    typedef int IPair __attribute__((ext_vector_type(2)));
    auto fetch = [&](int rb) {
        const int64_t r0 = (int64_t)rb * kStreamRows;
        const int64_t row = r0 + t;
        const int64_t rlast = (r0 + kStreamRows < n) ? r0 + kStreamRows : n;
        const int first = rowptr[r0];
        base = first & ~1;
        cnt = rowptr[rlast] - base;
        rs = re = 0;
        if (row < n) {
            rs = rowptr[row];
            re = rowptr[row + 1];
        }
        const int lastp = cnt > 0 ? (cnt - 1) >> 1 : 0;
#pragma unroll
        for (int u = 0; u < U / 2; ++u) {
            const int pr = t + u * kBlock;
            const int64_t kabs = cnt > 0 ? base + 2 * (int64_t)(pr <= lastp ? pr : lastp) : 0;
            const VPair av = *reinterpret_cast<const VPair *>(val + kabs);
            const IPair cv = *reinterpret_cast<const IPair *>(col + kabs);
            const int k = 2 * pr;
            a[2 * u] = av.x;
            a[2 * u + 1] = av.y;
            c[2 * u] = (k < cnt && base + k >= first) ? cv.x : 0;
            c[2 * u + 1] = (k + 1 < cnt) ? cv.y : 0;
        }
    };
    // The first row-block's loads are issued before the `done` word is looked at.
    if (rb_lo < rb_hi) fetch(rb_lo);
    double alpha = 0.0, beta = 0.0;
    const double *__restrict__ p_old = nullptr;
    double *__restrict__ p_new = nullptr;
    // FUSE: the operands of p_k = z + beta p_{k-1} for the gathered columns and the own row.  Those of the first
    // row block are requested BEFORE the head (whose partial reduction is two dependent round trips), so the
    // launch-bound systems this form serves (one row block per workgroup) overlap the two latencies.
    double zg[FUSE ? U : 1], pg[FUSE ? U : 1], zo = 0.0, po = 0.0, xo = 0.0;
    auto gather_fused = [&](int rb) {
        const int64_t row = (int64_t)rb * kStreamRows + t;
        if (row < n) {
            zo = fa.z[row];
            po = p_old[row];
            xo = fa.xvec[row];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            zg[u] = fa.z[c[u]];
            pg[u] = p_old[c[u]];
        }
    };
    if (FUSE) {
        const int kpar = ctl.scal->k & 1;
        p_new = kpar ? fa.p1 : fa.p0;
        p_old = kpar ? fa.p0 : fa.p1;
        if (rb_lo < rb_hi) gather_fused(rb_lo);
        if (!fused_head(ctl.scal, fa, sh, alpha, beta)) return;
    } else if (CTL) {
        if (!iteration_head(ctl)) return;
    }
    double acc = 0.0;
    for (int rb = rb_lo; rb < rb_hi; ++rb) {
        const int64_t row = (int64_t)rb * kStreamRows + t;
        const int ks = rs - base, ke = re - base;
        if (FUSE && rb != rb_lo) gather_fused(rb);
        double xv[U];
        if (FUSE) {
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = zg[u] + beta * pg[u];                 // = p_k[c], cg.py:83
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = (double)x[c[u]];
        }
#pragma unroll
        for (int u = 0; u < U / 2; ++u) {
            const int k = 2 * (t + u * kBlock);
            if (k < cnt) {                  // one 16-byte LDS store per pair
                double2 pp;
                pp.x = (double)a[2 * u] * xv[2 * u];
                pp.y = (double)a[2 * u + 1] * xv[2 * u + 1];
                *reinterpret_cast<double2 *>(prod + k) = pp;
            }
        }
        __syncthreads();
        if (row < n) {
            double s = 0.0;
            for (int k = ks; k < ke; ++k) s += prod[k];
            y[row] = (YT)s;
            if (FUSE) {
                const double pn = zo + beta * po;                                      // cg.py:83
                p_new[row] = pn;
                fa.xvec[row] = xo + alpha * po;                                        // cg.py:79, one update late
                acc += s * pn;
            } else if (DOT) {
                acc += s * xdot[row];
            }
        }
        __syncthreads();
        if (rb + 1 < rb_hi) fetch(rb + 1);
    }
    if (DOT) {
        const double tot = block_sum(acc, sh);
        if (t == 0) part_pq[blockIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------------------------------
// CSR-vector SpMV (longer rows: M = L L^T, dilated learned factors).  TPR lanes share a row, each
// strides over its non-zeros, then a shuffle tree combines them (order differs from the sequential
// sum: parity for this kernel is tolerance-based, ~1 ulp of the row's magnitude).
// ------------------------------------------------------------------------------------------------
template <int TPR, typename VT, typename XT, bool CTL, bool DOT, typename YT>
__global__ __launch_bounds__(kBlock) void k_spmv_vector(int64_t n, const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col,
                                                        const VT *__restrict__ val, const XT *__restrict__ x,
                                                        const double *__restrict__ xdot, YT *__restrict__ y,
                                                        double *__restrict__ part_pq, IterCtlDev ctl) {
    __shared__ double sh[4];
    if (CTL) {
        if (!iteration_head(ctl)) return;
    }
    constexpr int RPB = kBlock / TPR;  // rows per workgroup step
    const int t = threadIdx.x;
    const int lane = t % TPR;
    const int v = virtual_block();
    const int ngroups = (int)((n + RPB - 1) / RPB);
    int g_lo, g_hi;
    split_range(ngroups, v, g_lo, g_hi);
    double acc = 0.0;
    for (int g = g_lo; g < g_hi; ++g) {
        const int64_t row = (int64_t)g * RPB + t / TPR;
        double s = 0.0;
        if (row < n) {
            // each lane takes aligned PAIRS of consecutive non-zeros (16-byte value, 8-byte column loads): half the
            // lanes per row, so twice the rows in flight -- a 65K-row factor with 15 entries per row fits the chip in
            // one pass.  A pair may start one entry before the row or end one after it: those halves are skipped.
            typedef VT VPair __attribute__((ext_vector_type(2)));
            typedef int IPair __attribute__((ext_vector_type(2)));
            const int rs = rowptr[row], re = rowptr[row + 1];
            for (int k = (rs & ~1) + 2 * lane; k < re; k += 2 * TPR) {
                const VPair av = *reinterpret_cast<const VPair *>(val + k);
                const IPair cv = *reinterpret_cast<const IPair *>(col + k);
                if (k >= rs) s += (double)av.x * (double)x[cv.x];
                if (k + 1 < re) s += (double)av.y * (double)x[cv.y];
            }
        }
#pragma unroll
        for (int off = TPR / 2; off > 0; off >>= 1) s += __shfl_down(s, off, TPR);
        if (lane == 0 && row < n) {
            y[row] = (YT)s;
            if (DOT) acc += s * xdot[row];
        }
    }
    if (DOT) {
        const double tot = block_sum(acc, sh);
        if (t == 0) part_pq[blockIdx.x] = tot;
    }
}


// ------------------------------------------------------------------------------------------------
// CSR SpMV with the x-vector tile staged in LDS (banded / stencil-like matrices).
//
// The columns of a 256-row block fall into a few runs (5-point: i-n, i, i+n; 7-point: five runs).  At setup
// (k_tile_plan) each block gets the list of 64-double chunks of x it touches and every non-zero a 16-bit
// index into the LDS image of those chunks.  At run time the block stages its chunks with coalesced 512-B
// wave loads, streams val[] (8 B) and the local index (2 B instead of the 4-B column) coalesced, takes
// x from LDS instead of gathering it through L1/L2, and finishes like the CSR-stream kernel: products
// parked in LDS, thread i adds row i in column order -- the same bits as the gather kernels and the CPU.
// Per non-zero the matrix stream shrinks from 12 to 10 bytes and the global gathers disappear.
// ------------------------------------------------------------------------------------------------
// `diag_from` (colour sweeps: the rows are one level of a level-ordered triangular factor, columns are positions): a column
// >= diag_from is the row's own diagonal -- no chunk is staged for it and its local index is kTileDiag (the kernel
// multiplies it by 1.0, so the product slot holds the diagonal itself).  INT_MAX: a plain matrix.
__global__ __launch_bounds__(kBlock) void k_tile_plan(int64_t n, const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col, int nrb,
                                                      int32_t *__restrict__ chunks, int32_t *__restrict__ nchunks,
                                                      uint16_t *__restrict__ lidx, int *ok_and_max, int diag_from,
                                                      int allow_flagged /* ok_and_max[2] counts the blocks without a tile */) {
    __shared__ int s_min, s_max, s_nc;
    __shared__ uint16_t slot_of[kTileTableMax];
    const int t = threadIdx.x;
    for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int64_t r0 = (int64_t)rb * kStreamRows;
        const int64_t rlast = (r0 + kStreamRows < n) ? r0 + kStreamRows : n;
        const int base = rowptr[r0], cnt = rowptr[rlast] - base;
        if (t == 0) { s_min = 0x7fffffff; s_max = -1; s_nc = 0; }
        __syncthreads();
        int lo = 0x7fffffff, hi = -1;
        for (int k = t; k < cnt; k += kBlock) {
            const int c = col[base + k];
            if (c >= diag_from) continue;
            lo = c < lo ? c : lo;
            hi = c > hi ? c : hi;
        }
        if (hi >= 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
        __syncthreads();
        if (cnt == 0 || s_max < 0) {                      // no entries (or diagonals only): nothing to stage
            if (t == 0) nchunks[rb] = 0;
            for (int k = t; k < cnt; k += kBlock) lidx[base + k] = kTileDiag;
            __syncthreads();
            continue;
        }
        const int cb = s_min / kTileChunk;
        const int span = s_max / kTileChunk - cb + 1;
        // not tileable: no room for the alignment slots of the grouped loads
        if (cnt > kStreamCap - 3) {
            if (t == 0) { nchunks[rb] = 0; atomicExch(&ok_and_max[0], 0); }
            __syncthreads();
            continue;
        }
        if (span > kTileTableMax) {
            // Columns far apart (a system numbered colour by colour, periodic couplings, a second region): too wide for the
            // table of chunk ids, yet possibly only a few chunks.  The distinct chunk ids go through a 128-entry hash table
            // instead; slots are still handed out in ascending chunk order, so the plan is what the table would have made.
            int *hkeys = reinterpret_cast<int *>(slot_of), *hrank = hkeys + 128;
            __shared__ int s_over;
            if (t < 128) hkeys[t] = -1;
            if (t == 0) s_over = 0;
            __syncthreads();
            for (int k = t; k < cnt; k += kBlock) {
                const int c = col[base + k];
                if (c >= diag_from) continue;
                const int ch = c / kTileChunk;
                unsigned hsl = ((unsigned)ch * 2654435761u) >> 25;
                int probes = 0;
                for (; probes < 128; ++probes, hsl = (hsl + 1) & 127) {
                    const int old = atomicCAS(&hkeys[hsl], -1, ch);
                    if (old == -1) { atomicAdd(&s_nc, 1); break; }
                    if (old == ch) break;
                }
                if (probes == 128) s_over = 1;
            }
            __syncthreads();
            const bool ok = !s_over && s_nc <= kTileMaxChunks;
            if (t == 0) {
                if (!ok) {
                    if (allow_flagged) {                       // too many chunks: this block gathers (k_spmv_tile<..., MIX>)
                        nchunks[rb] = -1;
                        atomicAdd(&ok_and_max[2], 1);
                    } else {
                        nchunks[rb] = 0;
                        atomicExch(&ok_and_max[0], 0);
                    }
                } else {
                    int keys[kTileMaxChunks], where[kTileMaxChunks], nc = 0;
                    for (int e = 0; e < 128; ++e)
                        if (hkeys[e] >= 0) {                      // insertion sort by chunk id
                            int q = nc++;
                            for (; q > 0 && keys[q - 1] > hkeys[e]; --q) { keys[q] = keys[q - 1]; where[q] = where[q - 1]; }
                            keys[q] = hkeys[e];
                            where[q] = e;
                        }
                    for (int q = 0; q < nc; ++q) {
                        chunks[(int64_t)rb * kTileMaxChunks + q] = keys[q];
                        hrank[where[q]] = q;
                    }
                    nchunks[rb] = nc;
                    atomicMax(&ok_and_max[1], nc);
                }
            }
            __syncthreads();
            if (ok)
                for (int k = t; k < cnt; k += kBlock) {
                    const int c = col[base + k];
                    if (c >= diag_from) { lidx[base + k] = kTileDiag; continue; }
                    const int ch = c / kTileChunk;
                    unsigned hsl = ((unsigned)ch * 2654435761u) >> 25;
                    while (hkeys[hsl] != ch) hsl = (hsl + 1) & 127;
                    lidx[base + k] = (uint16_t)(hrank[hsl] * kTileChunk + c % kTileChunk);
                }
            __syncthreads();
            continue;
        }
        for (int e = t; e < span; e += kBlock) slot_of[e] = 0;
        __syncthreads();
        for (int k = t; k < cnt; k += kBlock) {
            const int c = col[base + k];
            if (c < diag_from) slot_of[c / kTileChunk - cb] = 1;
        }
        __syncthreads();
        {   // ascending chunk ids -> slots 1..nc: thread t numbers the marks of its run of the table behind a workgroup-wide scan
            // (one thread walking up to 4096 entries was 7 ms of a 256^3 plan)
            __shared__ int s_wave[kBlock / 64];
            const int per = (span + kBlock - 1) / kBlock, e0 = t * per, e1 = (e0 + per < span) ? e0 + per : span;
            int mine = 0;
            for (int e = e0; e < e1; ++e) mine += slot_of[e] ? 1 : 0;
            int incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(incl, off);
                if ((t & 63) >= off) incl += y;
            }
            if ((t & 63) == 63) s_wave[t >> 6] = incl;
            __syncthreads();
            int before = incl - mine, total = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) {
                if (w < (t >> 6)) before += s_wave[w];
                total += s_wave[w];
            }
            int nc = before;
            for (int e = e0; e < e1; ++e)
                if (slot_of[e]) {
                    if (nc < kTileMaxChunks) chunks[(int64_t)rb * kTileMaxChunks + nc] = cb + e;
                    slot_of[e] = (uint16_t)(++nc);
                }
            if (t == 0) {
                s_nc = total;
                nchunks[rb] = total <= kTileMaxChunks ? total : (allow_flagged ? -1 : 0);   // -1: too many chunks, this block gathers (k_spmv_tile<..., MIX>)
                if (total > kTileMaxChunks) {
                    if (allow_flagged) atomicAdd(&ok_and_max[2], 1);
                    else atomicExch(&ok_and_max[0], 0);
                } else atomicMax(&ok_and_max[1], total);
            }
        }
        __syncthreads();
        if (s_nc <= kTileMaxChunks)
            for (int k = t; k < cnt; k += kBlock) {
                const int c = col[base + k];
                lidx[base + k] = c >= diag_from ? kTileDiag : (uint16_t)((slot_of[c / kTileChunk - cb] - 1) * kTileChunk + c % kTileChunk);
            }
        __syncthreads();
    }
}

void launch_tile_plan(const CsrDev &A, int nrb, int32_t *chunks, int32_t *nchunks, uint16_t *lidx, int *ok_and_max_dev,
                      hipStream_t s) {
    const int grid = nrb < 2048 ? nrb : 2048;
    hipLaunchKernelGGL(k_tile_plan, dim3(grid), dim3(kBlock), 0, s, A.n, A.rowptr, A.col, nrb, chunks, nchunks, lidx,
                       ok_and_max_dev, 0x7fffffff, 1);            // (ok_and_max_dev: three ints)
}

// The same plan for `count` consecutive rows of a level-ordered factor copy that start at row j0 (one level of a colour sweep):
// blocks are counted from j0, the arrays of block b are chunks[b * kTileMaxChunks ..] / nchunks[b]; lidx is indexed like cpos.
void launch_sweep_tile_plan(int j0, int count, const int32_t *lo_rowptr, const int32_t *lo_cpos, int32_t *chunks, int32_t *nchunks,
                            uint16_t *lidx, int *ok_and_max_dev, hipStream_t s) {
    const int nrb = (count + kStreamRows - 1) / kStreamRows;
    const int grid = nrb < 2048 ? nrb : 2048;
    hipLaunchKernelGGL(k_tile_plan, dim3(grid), dim3(kBlock), 0, s, (int64_t)count, lo_rowptr + j0, lo_cpos, nrb, chunks, nchunks,
                       lidx, ok_and_max_dev, j0, 0);
}

// XT: x-tile elements staged per thread (tile_max_chunks * 64 / 256, rounded up).  VT / XV: storage types of the
// matrix values and of the staged vector (fp32 in the mixed-precision and lossless-fp32 modes; products and sums are
// fp64 either way, and <p,Ap> always uses the fp64 vector `xdot`).
// NT: the once-read streams (matrix values, local indices, row extents) are loaded and y is stored NON-TEMPORALLY -- for
// systems whose streams exceed the 256 MiB Infinity Cache, where caching them only evicts the x chunks that ARE reused
// (tools/stream_lab on the same box: an 11-reads-per-write stream 5.05 -> 5.22 TB/s with non-temporal accesses).
// MIX: some blocks of the plan touch more than kTileMaxChunks chunks (nchunks = -1: an OpenFOAM numbering whose refined cells were
// appended couples a block to a few dozen places) -- such a block gathers x[col] through the L2 like the CSR-stream kernel does, its
// value stream already prefetched; every other block keeps its LDS tile.  Same products, same order, same bits.
template <bool CTL, bool DOT, int XT, typename VT, typename XV, bool NT = false, bool MIX = false>
__global__ __launch_bounds__(kBlock) void k_spmv_tile(int64_t n, const int32_t *__restrict__ rowptr,
                                                      const VT *__restrict__ val,
                                                      const uint16_t *__restrict__ lidx,
                                                      const int32_t *__restrict__ chunks,
                                                      const int32_t *__restrict__ nchunks,
                                                      const XV *__restrict__ x, const double *__restrict__ xdot,
                                                      double *__restrict__ y, int nrb, int tile_doubles,
                                                      double *__restrict__ part_pq, IterCtlDev ctl, int64_t nnz, int cyclic,
                                                      const int32_t *__restrict__ col) {
    constexpr int U = kStreamCap / kBlock;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    XV *xs = reinterpret_cast<XV *>(smem);  // the staged x chunks of this block (tile_doubles slots reserved)
    double *prod = smem + tile_doubles;     // products, kStreamCap + 4 doubles
    double *sh = prod + kStreamCap + 4;     // 4 doubles for the block reduction
    const int t = threadIdx.x;
    // Which row blocks a workgroup takes.  Slabs (cache-resident systems): workgroup v owns one contiguous range, the slabs of an
    // XCD are neighbours, so the x halo of neighbouring row blocks is shared in that XCD's L2.  Cyclic (`cyclic` != 0: streams
    // from HBM): row block b, b + G, b + 2 G, ... -- the whole grid walks the matrix TOGETHER, so at any instant the chip reads one
    // contiguous window of the streams instead of G distant slabs (fewer DRAM pages open at once: tools/stream_lab, an 11 : 1
    // stream 5.3 -> 5.6 TB/s, non-temporal 5.4 -> 6.0); row blocks a plane apart still meet in one XCD when G is a multiple of 8.
    int rb_lo, rb_hi, rb_step = 1;
    if (cyclic) {
        // (cyclic == 2: inside every pass of G blocks an XCD takes a contiguous run of them -- the window the chip reads is the
        // same, the x halo of neighbouring row blocks is shared in that XCD's L2 again)
        rb_lo = cyclic == 2 ? virtual_block() : (int)blockIdx.x;
        rb_hi = nrb;
        rb_step = gridDim.x;
    } else {
        split_range(nrb, virtual_block(), rb_lo, rb_hi);
    }
    constexpr int G = 16 / sizeof(VT);      // consecutive non-zeros per 16-byte value load: 2 (fp64) or 4 (fp32)
    constexpr int UP = U / G;               // such groups per thread
    typedef VT VPair __attribute__((ext_vector_type(G)));            // native vectors: one load, stay in registers
    typedef unsigned short IPair __attribute__((ext_vector_type(G)));   // the group's 16-bit local indices (4 or 8 bytes)
    VPair a[UP];
    IPair li[UP];
    constexpr int XP = (XT * (kBlock / 64) + 7) / 8;   // wave instructions that stage two chunks each
    typedef XV XPair __attribute__((ext_vector_type(2)));
    XPair xt[XP];
    int cnt = 0, base = 0, rs = 0, re = 0, nc = 0;
    const int lane = t & 63, wv = t >> 6;
    // Everything block `rb` needs from memory -> registers: its slice of the matrix stream AND its x chunks
    // (wave w stages chunks w, w+4, ...: 64 lanes x 8 B = one 512-B run; the chunk id is wave-uniform and travels
    // through the scalar unit).  Issued one block ahead, so the loads fly during the previous block's phases.
    // The matrix slice is read as aligned PAIRS of consecutive non-zeros, lane i <-> pair i: 16-byte value loads
    // (1 KiB per wave instruction) and 4-byte index loads instead of 8- and 2-byte ones -- the texture-address unit
    // issues per instruction, not per byte.  Slots are counted from the even index at or below the block's first
    // non-zero, so slot 0 may belong to the previous block (its product is never read).
    auto fetch = [&](int rb) {
        const int64_t r0 = (int64_t)rb * kStreamRows;
        const int64_t row = r0 + t;
        const int64_t rlast = (r0 + kStreamRows < n) ? r0 + kStreamRows : n;
        base = rowptr[r0] & ~(G - 1);
        cnt = rowptr[rlast] - base;
        rs = re = 0;
        if (row < n) {                      // both extents in one 8-byte load (4-byte aligned: fine for global memory)
            struct __attribute__((packed, aligned(4))) Ext { int32_t s, e; };
            const Ext *ep = reinterpret_cast<const Ext *>(rowptr + row);
            const Ext ext = *ep;
            rs = ext.s;
            re = ext.e;
        }
        const int lastp = cnt > 0 ? (cnt - 1) / G : 0;
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int pr = t + u * kBlock;
            const int64_t kabs = base + G * (int64_t)(pr <= lastp ? pr : lastp);
            // Unconditional aligned pair loads.  The pair that holds the matrix's very last non-zero when nnz is odd reads
            // 8 (4) bytes past the array: inside the same aligned 16 (8) bytes, hence the same page -- never a fault --
            // and its product lands in a slot no row sum reads.  (The launcher checks the 16-byte alignment of val / x.)
            const VPair *vp = reinterpret_cast<const VPair *>(val + (cnt > 0 ? kabs : 0));
            const IPair *ip = reinterpret_cast<const IPair *>(lidx + (cnt > 0 ? kabs : 0));      // lidx is padded by 4 entries
            a[u] = NT ? __builtin_nontemporal_load(vp) : *vp;
            li[u] = NT ? __builtin_nontemporal_load(ip) : *ip;
        }
        nc = nchunks[rb];
        const int32_t *__restrict__ cl = chunks + (int64_t)rb * kTileMaxChunks;
        // two chunks per wave instruction: lanes 0-31 take chunk 2q, lanes 32-63 chunk 2q+1, two elements per lane.
        // The block's chunk ids are read once per wave (lane l holds id l, always inside the 40-entry list) and handed
        // to the lanes that need them through the cross-lane network: one memory instruction instead of one per pair.
        const int my_cid = cl[lane < kTileMaxChunks ? lane : 0];
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int ci = 2 * (wv + u * (kBlock / 64)) + (lane >> 5);
            const int cid = __shfl(my_cid, ci);
            // clamped to the last aligned pair that holds a vector element (an element past the end is never indexed)
            const int64_t gi = (int64_t)(ci < nc ? cid : 0) * kTileChunk + 2 * (lane & 31);
            xt[u] = *reinterpret_cast<const XPair *>(x + (gi < n ? gi : ((n - 1) & ~(int64_t)1)));
        }
    };
    if (rb_lo < rb_hi) fetch(rb_lo);
    if (CTL) {
        if (!iteration_head(ctl)) return;
    }
    double acc = 0.0;
    for (int rb = rb_lo; rb < rb_hi; rb += rb_step) {
        const int64_t row = (int64_t)rb * kStreamRows + t;
        const int ks = rs - base, ke = re - base, cnt_cur = cnt;
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int ci = 2 * (wv + u * (kBlock / 64)) + (lane >> 5);
            if (ci < nc) *reinterpret_cast<XPair *>(xs + ci * kTileChunk + 2 * (lane & 31)) = xt[u];
        }
        __syncthreads();                    // tile complete (and every thread is past the previous row sums)
        if (MIX && nc < 0) {                // a block without a tile: its columns from the CSR, x through the L2
            typedef int CGroup __attribute__((ext_vector_type(G)));
            const int lastp = cnt_cur > 0 ? (cnt_cur - 1) / G : 0;
            CGroup cg[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int pr = t + u * kBlock;
                const int64_t kabs = base + G * (int64_t)(pr <= lastp ? pr : lastp);
                cg[u] = *reinterpret_cast<const CGroup *>(col + (cnt_cur > 0 ? kabs : 0));    // (col is padded like lidx: see make_plan)
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int k = G * (t + u * kBlock);
                XV xg[G];
#pragma unroll
                for (int e = 0; e < G; ++e) xg[e] = x[cg[u][e]];
                if (k < cnt_cur) {
#pragma unroll
                    for (int e = 0; e < G; e += 2) {
                        double2 pp;
                        pp.x = (double)a[u][e] * (double)xg[e];
                        pp.y = (double)a[u][e + 1] * (double)xg[e + 1];
                        *reinterpret_cast<double2 *>(prod + k + e) = pp;
                    }
                }
            }
        } else {
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int k = G * (t + u * kBlock);
            if (k < cnt_cur) {              // 16-byte LDS stores (slots up to cnt_cur + G - 1 may be written: never read)
#pragma unroll
                for (int e = 0; e < G; e += 2) {
                    double2 pp;
                    pp.x = (double)a[u][e] * (double)xs[li[u][e]];
                    pp.y = (double)a[u][e + 1] * (double)xs[li[u][e + 1]];
                    *reinterpret_cast<double2 *>(prod + k + e) = pp;
                }
            }
        }
        }
        if (rb + rb_step < rb_hi) fetch(rb + rb_step);   // next block's stream and x chunks are in flight from here on
        __syncthreads();
        if (row < n) {
            double s = 0.0;
            for (int k = ks; k < ke; ++k) s += prod[k];
            if (NT) __builtin_nontemporal_store(s, y + row);
            else y[row] = s;
            if (DOT) acc += s * xdot[row];
        }
    }
    if (DOT) {
        const double tot = block_sum(acc, sh);
        if (t == 0) part_pq[blockIdx.x] = tot;
    }
}

static IterCtlDev to_dev(const IterCtl *c) {
    IterCtlDev d{nullptr};
    if (c) d = IterCtlDev{c->scal};
    return d;
}

template <typename VT, typename XT, typename YT>
static void spmv_dispatch(const CsrDev &A, const SpmvPlan &plan, const VT *val, const XT *x, const double *xdot,
                          YT *y, double *part_pq, const IterCtl *ctl, hipStream_t s) {
    const IterCtlDev d = to_dev(ctl);
    const bool dot = part_pq != nullptr;
    const bool c = ctl != nullptr;
#define DPCG_LAUNCH_STREAM(CTLV, DOTV)                                                                     \
    hipLaunchKernelGGL((k_spmv_stream<VT, XT, CTLV, DOTV, YT>), dim3(plan.grid), dim3(kBlock), 0, s, A.n, \
                       A.rowptr, A.col, val, x, xdot, y, plan.nrb, part_pq, d, FuseArgs{})
#define DPCG_LAUNCH_VECTOR(TPRV, CTLV, DOTV)                                                                     \
    hipLaunchKernelGGL((k_spmv_vector<TPRV, VT, XT, CTLV, DOTV, YT>), dim3(plan.grid), dim3(kBlock), 0, s, A.n, \
                       A.rowptr, A.col, val, x, xdot, y, part_pq, d)
#define DPCG_VECTOR_CASE(TPRV)                         \
    case TPRV:                                         \
        if (c && dot) DPCG_LAUNCH_VECTOR(TPRV, true, true);   \
        else if (dot) DPCG_LAUNCH_VECTOR(TPRV, false, true);  \
        else DPCG_LAUNCH_VECTOR(TPRV, false, false);          \
        break
    // the x-tile kernel reads val and x as aligned pairs: a misaligned caller buffer takes the gather kernel
    const bool pair_aligned = (((uintptr_t)val) & 15) == 0 && (((uintptr_t)x) & (2 * sizeof(XT) - 1)) == 0;   // 16-byte value groups, x pairs
    if (plan.kernel == SPMV_TILE && std::is_same<YT, double>::value && pair_aligned) {
        const int tile_doubles = plan.tile_max_chunks * kTileChunk;
        const size_t lds = (size_t)(tile_doubles + kStreamCap + 8) * sizeof(double);
#define DPCG_LAUNCH_TILE_X(CTLV, DOTV, XTV, NTV)                                                                         \
    hipLaunchKernelGGL((k_spmv_tile<CTLV, DOTV, XTV, VT, XT, NTV>), dim3(plan.grid), dim3(kBlock), lds, s, A.n, A.rowptr, \
                       val, plan.tile_lidx, plan.tile_chunks, plan.tile_nchunks, x, xdot, (double *)y, plan.nrb,         \
                       tile_doubles, part_pq, d, A.nnz, plan.cyclic, A.col)
#define DPCG_LAUNCH_TILE_MIX(CTLV, DOTV, XTV)                                                                                  \
    hipLaunchKernelGGL((k_spmv_tile<CTLV, DOTV, XTV, VT, XT, false, true>), dim3(plan.grid), dim3(kBlock), lds, s, A.n, A.rowptr, \
                       val, plan.tile_lidx, plan.tile_chunks, plan.tile_nchunks, x, xdot, (double *)y, plan.nrb,             \
                       tile_doubles, part_pq, d, A.nnz, plan.cyclic, A.col)
#define DPCG_LAUNCH_TILE(CTLV, DOTV)                                                          \
    do {                                                                                      \
        if (plan.tile_mixed) {                                                                \
            if (plan.tile_max_chunks <= 20) DPCG_LAUNCH_TILE_MIX(CTLV, DOTV, 5);              \
            else DPCG_LAUNCH_TILE_MIX(CTLV, DOTV, (kTileMaxChunks * kTileChunk / kBlock));    \
        } else if (plan.stream_nt) {                                                          \
            if (plan.tile_max_chunks <= 20) DPCG_LAUNCH_TILE_X(CTLV, DOTV, 5, true);          \
            else DPCG_LAUNCH_TILE_X(CTLV, DOTV, (kTileMaxChunks * kTileChunk / kBlock), true);  \
        } else {                                                                              \
            if (plan.tile_max_chunks <= 20) DPCG_LAUNCH_TILE_X(CTLV, DOTV, 5, false);         \
            else DPCG_LAUNCH_TILE_X(CTLV, DOTV, (kTileMaxChunks * kTileChunk / kBlock), false); \
        }                                                                                     \
    } while (0)
        if (c && dot) DPCG_LAUNCH_TILE(true, true);
        else if (dot) DPCG_LAUNCH_TILE(false, true);
        else DPCG_LAUNCH_TILE(false, false);
#undef DPCG_LAUNCH_TILE
#undef DPCG_LAUNCH_TILE_MIX
#undef DPCG_LAUNCH_TILE_X
    } else if (plan.kernel == SPMV_STREAM || plan.kernel == SPMV_TILE) {
        if (c && dot) DPCG_LAUNCH_STREAM(true, true);
        else if (dot) DPCG_LAUNCH_STREAM(false, true);
        else DPCG_LAUNCH_STREAM(false, false);
    } else {
        switch (plan.tpr) {
            DPCG_VECTOR_CASE(2);
            DPCG_VECTOR_CASE(4);
            DPCG_VECTOR_CASE(8);
            DPCG_VECTOR_CASE(16);
            DPCG_VECTOR_CASE(32);
            default:
                DPCG_VECTOR_CASE(64);
        }
    }
#undef DPCG_VECTOR_CASE
#undef DPCG_LAUNCH_VECTOR
#undef DPCG_LAUNCH_STREAM
}

// KA of the two-kernel iteration (see fused_head): the gather kernel with FUSE.  A system whose plan is the x-tile
// kernel is too large for this form to pay (dpcg_api.hip: fuse_eligible); should it be asked for anyway, the gather
// kernel runs over the same row blocks.
void launch_spmv_fused(const CsrDev &A, const SpmvPlan &plan, const FuseArgs &fa, double *q, double *part_pq,
                       Scalars *scal, hipStream_t s) {
    const IterCtlDev d{scal};
    hipLaunchKernelGGL((k_spmv_stream<double, double, true, true, double, true>), dim3(plan.grid), dim3(kBlock), 0, s,
                       A.n, A.rowptr, A.col, A.val, nullptr, nullptr, q, plan.nrb, part_pq, d, fa);
}

void launch_spmv(const CsrDev &A, const SpmvPlan &plan, const double *x, double *y, double *part_pq,
                 const IterCtl *ctl, hipStream_t s) {
    spmv_dispatch<double, double, double>(A, plan, A.val, x, x, y, part_pq, ctl, s);
}

void launch_spmv_xdot(const CsrDev &A, const SpmvPlan &plan, const double *x, const double *xdot, double *y, double *part,
                      hipStream_t s) {
    spmv_dispatch<double, double, double>(A, plan, A.val, x, xdot, y, part, nullptr, s);
}

// Mixed precision (config C5): fp32 matrix values and fp32 gathered vector, fp64 products/sums.
void launch_spmv_f32in(const CsrDev &A, const SpmvPlan &plan, const float *x32, const double *x64, double *y,
                       double *part_pq, const IterCtl *ctl, hipStream_t s) {
    spmv_dispatch<float, float, double>(A, plan, A.val32, x32, x64, y, part_pq, ctl, s);
}

void launch_spmv_val32(const CsrDev &A, const SpmvPlan &plan, const double *x, double *y, double *part_pq,
                       const IterCtl *ctl, hipStream_t s) {
    spmv_dispatch<float, double, double>(A, plan, A.val32, x, x, y, part_pq, ctl, s);
}

__global__ __launch_bounds__(kBlock) void k_val32_check(int64_t nnz, const double *__restrict__ val,
                                                        float *__restrict__ val32, int *lossy) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride) {
        const double v = val[k];
        const float f = (float)v;
        val32[k] = f;
        bad = bad || ((double)f != v);
    }
    if (bad) atomicExch(lossy, 1);
}

void launch_val32_check(int64_t nnz, const double *val, float *val32, int *lossy_dev, hipStream_t s) {
    int64_t g = (nnz + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_val32_check, dim3((int)g), dim3(kBlock), 0, s, nnz, val, val32, lossy_dev);
}

void launch_spmv_f32out(const CsrDev &A, const SpmvPlan &plan, const float *x32, float *y32, hipStream_t s) {
    spmv_dispatch<float, float, float>(A, plan, A.val32, x32, nullptr, y32, nullptr, nullptr, s);
}

}  // namespace dpcg
