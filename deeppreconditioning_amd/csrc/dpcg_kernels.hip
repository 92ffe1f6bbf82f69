// Device kernels of libdpcg.so -- hand-written HIP for gfx950 (MI355X, CDNA4, wave64).
//
// Everything here is HBM-bandwidth work (SpMV, triangular solves, dots, axpys): no MFMA.  The
// rules that matter are coalesced streams, LDS staging, enough workgroups in flight and as few
// kernel boundaries / bytes per PCG iteration as possible.  Compiled with -ffp-contract=off so
// that a*b+c is two roundings, as in the CPU reference path (scipy/ATen CSR row sums, unfused
// torch mul+add at cg.py:79-83); the in-order row sums below then reproduce the oracle bit for bit.
#include <algorithm>
#include <type_traits>

#include "dpcg_internal.h"

namespace dpcg {

// ------------------------------------------------------------------------------------------------
// Deterministic reductions: wave64 shuffle tree -> 4 wave sums in LDS -> fixed-order add.
// ------------------------------------------------------------------------------------------------
// Sum over the 256 threads of the workgroup; every thread gets the result.  sh: 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    __syncthreads();  // sh may still be read from a previous use
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// Two sums at once (one pair of barriers).  sh: 8 doubles of LDS.
__device__ __forceinline__ void block_sum2(double &a, double &b, double *sh) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) {
        sh[threadIdx.x >> 6] = a;
        sh[4 + (threadIdx.x >> 6)] = b;
    }
    __syncthreads();
    a = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    b = ((sh[4] + sh[5]) + sh[6]) + sh[7];
}

// Every workgroup re-reduces the <= kMaxGrid partials a previous kernel wrote: same order in every
// workgroup, so all of them hold bit-identical scalars without any inter-workgroup hand-off.
__device__ __forceinline__ double reduce_partials(const double *__restrict__ part, int n_part, double *sh) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n_part; i += kBlock) s += part[i];
    return block_sum(s, sh);
}

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous slab of the
// work so the x-vector halo of neighbouring row-blocks is shared in that XCD's 4 MiB L2.
// Placement is a speed matter only; any mapping gives the same result.
__device__ __forceinline__ int virtual_block() {
    const int G = gridDim.x, b = blockIdx.x;
    return (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
}


// [lo,hi) = floor(v*total/G), floor((v+1)*total/G) for G = gridDim.x <= 2048, without 64-bit division:
// total = q*G + r  =>  floor(v*total/G) = v*q + floor(v*r/G), and v*r < G*G fits 32 bits.
__device__ __forceinline__ void split_range(int total, int v, int &lo, int &hi) {
    const unsigned G = gridDim.x;
    const unsigned q = (unsigned)total / G, r = (unsigned)total - q * G;
    lo = (int)((unsigned)v * q + ((unsigned)v * r) / G);
    hi = (int)((unsigned)(v + 1) * q + ((unsigned)(v + 1) * r) / G);
}

// ------------------------------------------------------------------------------------------------
// Control flow of the iteration without a host round trip (cg.py:70-71):
//   * the stopping test for iterate k+1 is evaluated by workgroup 0 of K3 (the last kernel of
//     update k+1), which writes done / k / history / rz_next into the device-resident Scalars;
//   * K1 (the SpMV, head of the next update) only reads the `done` word an EARLIER kernel wrote and
//     workgroup 0 rotates rz_next -> rz.  No kernel reads a scalar that the same kernel writes.
// Once `done` is set every later kernel of the replayed graph returns at once.
// ------------------------------------------------------------------------------------------------
struct IterCtlDev {
    Scalars *scal;
};

__device__ __forceinline__ bool iteration_head(const IterCtlDev &c) {
    Scalars *sc = c.scal;
    if (sc->done) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->rz = sc->rz_next;     // <r,z> of the current iterate, cg.py:76
    return true;
}

// The test of cg.py:71 on the iterate that update k has just produced; one thread of one workgroup.
__device__ __forceinline__ void record_and_test(Scalars *sc, double rr, double rz_next, double *hist, int hist_cap,
                                                int k) {
    const double res = rr / sc->bb;                                     // cg.py:15-17
    const bool conv = (res < sc->rtol_sq) || (rr < sc->atol_sq);        // cg.py:71
    if (k < hist_cap) hist[k] = res;                                    // cg.py:67,88
    sc->res = res;
    sc->k = k;
    sc->rz_next = rz_next;
    int done = 0;
    if (conv) { done = 1; sc->status = DPCG_OK; }
    else if (!(res == res)) { done = 1; sc->status = DPCG_BREAKDOWN; }
    if (done) sc->done = 1;
    if (sc->progress)   // one posted 8-byte write to pinned host memory per update
        __hip_atomic_store(sc->progress, ((unsigned long long)k << 1) | (unsigned long long)done, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------------
// Two-kernel iteration.  CG has two global reductions per update (<p,Ap> and <r,z>), so two kernels is the floor:
//   KA (an SpMV kernel with FUSE): every workgroup re-reduces the partials of <r,z> and <r,r> that KB (or the
//      initial state) left, runs the stopping test of cg.py:71 on the current iterate k, forms
//      beta = <r,z>_k / <r,z>_{k-1} (cg.py:82), then  p_k = z + beta p_{k-1} (cg.py:83) and the deferred
//      x += alpha_{k-1} p_{k-1} (cg.py:79) for its own rows, and q = A p_k with the partials of <p_k,q>.  The
//      columns it gathers are recomputed as z[c] + beta p_{k-1}[c]: the same expression, hence the same bits,
//      as the stored p_k.  p is double-buffered (P[k & 1]) because neighbours still read p_{k-1}.
//   KB (k_update_r<PRE, true>): alpha, r, z, partials as before; workgroup 0 also advances k and rz_prev.
// x lags one update behind; k_final_fused applies the last one.  Before the first update rz_prev = +inf and
// alpha = 0, P[1] = 0, so update 0 degenerates to p_0 = z_0, x unchanged.  Scalars obey the same rule as in the
// three-kernel form: nobody reads a word that the same kernel writes (`done` excepted: a workgroup that sees it
// set early returns, which is what it would have decided anyway).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool fused_head(Scalars *sc, const FuseArgs &f, double *sh, double &alpha, double &beta) {
    if (sc->done) return false;
    const int k = sc->k;                                   // updates completed (written by KB / the initial state)
    double rz = 0.0, rr = 0.0;
    for (int i = threadIdx.x; i < f.n_part; i += kBlock) {
        rz += f.part_rz[i];
        rr += f.part_rr[i];
    }
    block_sum2(rz, rr, sh);                                // the arithmetic of reduce_partials, twice
    alpha = sc->alpha;
    beta = rz / sc->rz_prev;                               // cg.py:82
    bool stop = false;
    if (k > 0) {                                           // iterate 0 was tested by k_finalize_init (cg.py:66)
        const double res = rr / sc->bb;                    // cg.py:15-17
        const bool conv = (res < sc->rtol_sq) || (rr < sc->atol_sq);   // cg.py:71
        stop = conv || !(res == res);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (k < f.hist_cap) f.hist[k] = res;           // cg.py:88
            sc->res = res;
            if (stop) {
                sc->status = conv ? DPCG_OK : DPCG_BREAKDOWN;
                sc->done = 1;
                if (sc->progress)
                    __hip_atomic_store(sc->progress, ((unsigned long long)k << 1) | 1ull, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (stop) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc->rz = rz;  // <r,z> of the current iterate, read by KB (cg.py:76)
    return true;
}

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
    constexpr int U = kStreamCap / kBlock;  // (col,val) loads per thread per row-block
    __shared__ double prod[kStreamCap];
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
    // matrix stream of one row-block -> registers (all 2U loads of a thread in flight at once)
    auto fetch = [&](int rb) {
        const int64_t r0 = (int64_t)rb * kStreamRows;
        const int64_t row = r0 + t;
        const int64_t rlast = (r0 + kStreamRows < n) ? r0 + kStreamRows : n;
        base = rowptr[r0];
        cnt = rowptr[rlast] - base;
        rs = re = 0;
        if (row < n) {
            rs = rowptr[row];
            re = rowptr[row + 1];
        }
        const int32_t *__restrict__ cb = col + base;
        const VT *__restrict__ vb = val + base;
        const int last = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = t + u * kBlock;
            const int kk = k < cnt ? k : last;
            c[u] = cnt > 0 ? cb[kk] : 0;
            a[u] = cnt > 0 ? vb[kk] : (VT)0;
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
        for (int u = 0; u < U; ++u) {
            const int k = t + u * kBlock;
            if (k < cnt) prod[k] = (double)a[u] * xv[u];
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
            const int rs = rowptr[row], re = rowptr[row + 1];
            for (int k = rs + lane; k < re; k += TPR) s += (double)val[k] * (double)x[col[k]];
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
__global__ __launch_bounds__(kBlock) void k_tile_plan(int64_t n, const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col, int nrb,
                                                      int32_t *__restrict__ chunks, int32_t *__restrict__ nchunks,
                                                      uint16_t *__restrict__ lidx, int *ok_and_max) {
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
            lo = c < lo ? c : lo;
            hi = c > hi ? c : hi;
        }
        if (hi >= 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
        __syncthreads();
        if (cnt == 0) {
            if (t == 0) nchunks[rb] = 0;
            __syncthreads();
            continue;
        }
        const int cb = s_min / kTileChunk;
        const int span = s_max / kTileChunk - cb + 1;
        // not tileable: columns too spread out, or no room for the alignment slot of the pair loads
        if (span > kTileTableMax || cnt > kStreamCap - 1) {
            if (t == 0) { nchunks[rb] = 0; atomicExch(&ok_and_max[0], 0); }
            __syncthreads();
            continue;
        }
        for (int e = t; e < span; e += kBlock) slot_of[e] = 0;
        __syncthreads();
        for (int k = t; k < cnt; k += kBlock) slot_of[col[base + k] / kTileChunk - cb] = 1;
        __syncthreads();
        if (t == 0) {                                     // ascending chunk ids -> slots 1..nc
            int nc = 0;
            for (int e = 0; e < span; ++e)
                if (slot_of[e]) {
                    if (nc < kTileMaxChunks) chunks[(int64_t)rb * kTileMaxChunks + nc] = cb + e;
                    slot_of[e] = (uint16_t)(++nc);
                }
            s_nc = nc;
            nchunks[rb] = nc <= kTileMaxChunks ? nc : 0;
            if (nc > kTileMaxChunks) atomicExch(&ok_and_max[0], 0);
            else atomicMax(&ok_and_max[1], nc);
        }
        __syncthreads();
        if (s_nc <= kTileMaxChunks)
            for (int k = t; k < cnt; k += kBlock) {
                const int c = col[base + k];
                lidx[base + k] = (uint16_t)((slot_of[c / kTileChunk - cb] - 1) * kTileChunk + c % kTileChunk);
            }
        __syncthreads();
    }
}

void launch_tile_plan(const CsrDev &A, int nrb, int32_t *chunks, int32_t *nchunks, uint16_t *lidx, int *ok_and_max_dev,
                      hipStream_t s) {
    const int grid = nrb < 2048 ? nrb : 2048;
    hipLaunchKernelGGL(k_tile_plan, dim3(grid), dim3(kBlock), 0, s, A.n, A.rowptr, A.col, nrb, chunks, nchunks, lidx,
                       ok_and_max_dev);
}

// XT: x-tile elements staged per thread (tile_max_chunks * 64 / 256, rounded up).  VT / XV: storage types of the
// matrix values and of the staged vector (fp32 in the mixed-precision and lossless-fp32 modes; products and sums are
// fp64 either way, and <p,Ap> always uses the fp64 vector `xdot`).
template <bool CTL, bool DOT, int XT, typename VT, typename XV>
__global__ __launch_bounds__(kBlock) void k_spmv_tile(int64_t n, const int32_t *__restrict__ rowptr,
                                                      const VT *__restrict__ val,
                                                      const uint16_t *__restrict__ lidx,
                                                      const int32_t *__restrict__ chunks,
                                                      const int32_t *__restrict__ nchunks,
                                                      const XV *__restrict__ x, const double *__restrict__ xdot,
                                                      double *__restrict__ y, int nrb, int tile_doubles,
                                                      double *__restrict__ part_pq, IterCtlDev ctl, int64_t nnz) {
    constexpr int U = kStreamCap / kBlock;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    XV *xs = reinterpret_cast<XV *>(smem);  // the staged x chunks of this block (tile_doubles slots reserved)
    double *prod = smem + tile_doubles;     // products, kStreamCap + 2 doubles
    double *sh = prod + kStreamCap + 2;     // 4 doubles for the block reduction
    const int t = threadIdx.x;
    const int v = virtual_block();
    int rb_lo, rb_hi;
    split_range(nrb, v, rb_lo, rb_hi);
    constexpr int UP = U / 2;               // pairs of consecutive non-zeros per thread
    struct alignas(2 * sizeof(VT)) VPair { VT x, y; };
    VPair a[UP];
    uint32_t li[UP];                        // two 16-bit local indices per register
    XV xt[XT];
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
        base = rowptr[r0] & ~1;
        cnt = rowptr[rlast] - base;
        rs = re = 0;
        if (row < n) {
            rs = rowptr[row];
            re = rowptr[row + 1];
        }
        const int lastp = cnt > 0 ? (cnt - 1) >> 1 : 0;
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int pr = t + u * kBlock;
            const int64_t kabs = base + 2 * (int64_t)(pr <= lastp ? pr : lastp);
            if (cnt > 0) {
                if (kabs + 1 < nnz) {
                    a[u] = *reinterpret_cast<const VPair *>(val + kabs);
                } else {                    // the matrix's very last non-zero when nnz is odd
                    a[u].x = val[kabs];
                    a[u].y = (VT)0;
                }
                li[u] = *reinterpret_cast<const uint32_t *>(lidx + kabs);   // lidx is padded to an even length
            } else {
                a[u].x = a[u].y = (VT)0;
                li[u] = 0;
            }
        }
        nc = nchunks[rb];
        const int32_t *__restrict__ cl = chunks + (int64_t)rb * kTileMaxChunks;
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int ci = wv + u * (kBlock / 64);
            if (ci < nc) {
                const int chunk = __builtin_amdgcn_readfirstlane(cl[ci]);
                const int64_t gi = (int64_t)chunk * kTileChunk + lane;
                xt[u] = gi < n ? x[gi] : (XV)0;
            }
        }
    };
    if (rb_lo < rb_hi) fetch(rb_lo);
    if (CTL) {
        if (!iteration_head(ctl)) return;
    }
    double acc = 0.0;
    for (int rb = rb_lo; rb < rb_hi; ++rb) {
        const int64_t row = (int64_t)rb * kStreamRows + t;
        const int ks = rs - base, ke = re - base, cnt_cur = cnt;
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int ci = wv + u * (kBlock / 64);
            if (ci < nc) xs[ci * kTileChunk + lane] = xt[u];
        }
        __syncthreads();                    // tile complete (and every thread is past the previous row sums)
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int k = 2 * (t + u * kBlock);
            if (k < cnt_cur) {              // one 16-byte LDS store per pair (slot cnt_cur may be written: never read)
                double2 pp;
                pp.x = (double)a[u].x * (double)xs[li[u] & 0xffffu];
                pp.y = (double)a[u].y * (double)xs[li[u] >> 16];
                *reinterpret_cast<double2 *>(prod + k) = pp;
            }
        }
        if (rb + 1 < rb_hi) fetch(rb + 1);   // next block's stream and x chunks are in flight from here on
        __syncthreads();
        if (row < n) {
            double s = 0.0;
            for (int k = ks; k < ke; ++k) s += prod[k];
            y[row] = s;
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
    if (plan.kernel == SPMV_TILE && std::is_same<YT, double>::value) {
        const int tile_doubles = plan.tile_max_chunks * kTileChunk;
        const size_t lds = (size_t)(tile_doubles + kStreamCap + 6) * sizeof(double);
#define DPCG_LAUNCH_TILE_X(CTLV, DOTV, XTV)                                                                          \
    hipLaunchKernelGGL((k_spmv_tile<CTLV, DOTV, XTV, VT, XT>), dim3(plan.grid), dim3(kBlock), lds, s, A.n, A.rowptr,  \
                       val, plan.tile_lidx, plan.tile_chunks, plan.tile_nchunks, x, xdot, (double *)y, plan.nrb,     \
                       tile_doubles, part_pq, d, A.nnz)
#define DPCG_LAUNCH_TILE(CTLV, DOTV)                                                  \
    do {                                                                              \
        if (plan.tile_max_chunks <= 20) DPCG_LAUNCH_TILE_X(CTLV, DOTV, 5);            \
        else DPCG_LAUNCH_TILE_X(CTLV, DOTV, (kTileMaxChunks * kTileChunk / kBlock));  \
    } while (0)
        if (c && dot) DPCG_LAUNCH_TILE(true, true);
        else if (dot) DPCG_LAUNCH_TILE(false, true);
        else DPCG_LAUNCH_TILE(false, false);
#undef DPCG_LAUNCH_TILE
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

// ------------------------------------------------------------------------------------------------
// Vector updates of the iteration (cg.py:78-83) in two passes of 40 bytes per row each:
//   K2 (k_update_r):  alpha;  r -= alpha q;  z = dinv r;  partials <r,z>, <r,r>     reads q,r,dinv  writes r,z
//   K3 (k_update_xp): beta;   x += alpha p;  p = z + beta p                         reads z,p,x     writes x,p
// p is read once for both of its uses.  All vectors are handle-owned (256-B aligned): 16-byte
// accesses, loads of the next pair issued before the current pair is consumed, and the first loads
// issued before the partial reduction so that its latency is hidden.
// ------------------------------------------------------------------------------------------------
// PRE: 0 = M = I (z aliases r, not stored), 1 = Jacobi fused, 2 = generic M (z computed later).
template <int PRE, bool F2 = false>   // F2: KB of the two-kernel iteration (workgroup 0 also advances k and rz_prev)
__global__ __launch_bounds__(kBlock) void k_update_r(int64_t n, Scalars *__restrict__ sc,
                                                     const double *__restrict__ part_pq, int n_part_pq,
                                                     const double *__restrict__ q, double *__restrict__ r,
                                                     const double *__restrict__ dinv, double *__restrict__ z,
                                                     double *__restrict__ part_rz, double *__restrict__ part_rr) {
    __shared__ double sh[8];
    if (sc->done) return;
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double2 *__restrict__ q2 = reinterpret_cast<const double2 *>(q);
    const double2 *__restrict__ d2 = reinterpret_cast<const double2 *>(dinv);
    double2 *__restrict__ r2 = reinterpret_cast<double2 *>(r);
    double2 *__restrict__ z2 = reinterpret_cast<double2 *>(z);
    double2 qa = make_double2(0, 0), ra = qa, da = qa;
    bool have = i < n2;
    if (have) {
        qa = q2[i];
        ra = r2[i];
        if (PRE == 1) da = d2[i];
    }
    const double pq = reduce_partials(part_pq, n_part_pq, sh);
    const double rz_cur = sc->rz;
    const double alpha = rz_cur / pq;                                   // cg.py:78
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->alpha = alpha;                                              // read by K3 / KA (a later kernel)
        if (F2) {
            const int k1 = sc->k + 1;                                   // this update is complete once KB has run
            sc->rz_prev = rz_cur;                                       // read by the next KA only
            sc->k = k1;
            if (sc->progress)
                __hip_atomic_store(sc->progress, (unsigned long long)k1 << 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    double a_rz = 0.0, a_rr = 0.0;
    while (have) {
        const int64_t cur = i;
        const double2 qc = qa, rc = ra, dc = da;
        i += stride;
        have = i < n2;
        if (have) {
            qa = q2[i];
            ra = r2[i];
            if (PRE == 1) da = d2[i];
        }
        double2 rn;
        rn.x = rc.x - alpha * qc.x;                                     // cg.py:80
        rn.y = rc.y - alpha * qc.y;
        r2[cur] = rn;
        a_rr += rn.x * rn.x;                                            // cg.py:86
        a_rr += rn.y * rn.y;
        if (PRE == 1) {
            double2 zn;
            zn.x = dc.x * rn.x;                                         // cg.py:81 (M = diag(1/a_ii))
            zn.y = dc.y * rn.y;
            z2[cur] = zn;
            a_rz += rn.x * zn.x;                                        // cg.py:82
            a_rz += rn.y * zn.y;
        }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {               // odd tail element
        const int64_t e = n - 1;
        const double rn = r[e] - alpha * q[e];
        r[e] = rn;
        a_rr += rn * rn;
        if (PRE == 1) {
            const double zn = dinv[e] * rn;
            z[e] = zn;
            a_rz += rn * zn;
        }
    }
    block_sum2(a_rr, a_rz, sh);
    if (threadIdx.x == 0) {
        part_rr[blockIdx.x] = a_rr;
        if (PRE != 2) part_rz[blockIdx.x] = PRE == 1 ? a_rz : a_rr;     // PRE 0: z = r; PRE 2: <r,z> comes later
    }
}

void launch_update_r_two_kernel(int precond_fused, int64_t n, Scalars *scal, const double *part_pq, int n_part_pq,
                                const double *q, double *r, const double *dinv, double *z, double *part_rz,
                                double *part_rr, int grid, hipStream_t s) {
    if (precond_fused == 0)
        hipLaunchKernelGGL((k_update_r<0, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r,
                           dinv, z, part_rz, part_rr);
    else if (precond_fused == 1)
        hipLaunchKernelGGL((k_update_r<1, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r,
                           dinv, z, part_rz, part_rr);
    else
        hipLaunchKernelGGL((k_update_r<2, true>), dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r,
                           dinv, z, part_rz, part_rr);
}

// Two-kernel iteration: state before the first update (see fused_head).
__global__ void k_fused_init(Scalars *sc) {
    if (threadIdx.x == 0) {
        sc->rz_prev = __builtin_huge_val();   // beta_0 = <r,z>_0 / inf = 0  =>  p_0 = z_0
        sc->alpha = 0.0;                      // no deferred x update yet
    }
}

void launch_fused_init(Scalars *scal, hipStream_t s) { hipLaunchKernelGGL(k_fused_init, dim3(1), dim3(64), 0, s, scal); }

// End of a two-kernel solve: the deferred x += alpha_{k-1} p_{k-1} (cg.py:79) and, when the loop ran out of
// updates, the test of the last iterate (cg.py:86-88; the status stays MAX_ITER unless it passes).
__global__ __launch_bounds__(kBlock) void k_final_fused(int64_t n, Scalars *sc, const double *__restrict__ part_rr,
                                                        int n_part, double *hist, int hist_cap, double *__restrict__ x,
                                                        const double *__restrict__ p0, const double *__restrict__ p1) {
    __shared__ double sh[4];
    const int k = sc->k;
    if (k >= 1) {
        const double alpha = sc->alpha;
        const double *__restrict__ p = ((k - 1) & 1) ? p1 : p0;
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) x[i] = x[i] + alpha * p[i];
    }
    if (blockIdx.x == 0) {
        const int done = sc->done;                         // uniform: written by an earlier kernel only
        if (!done) {
            const double rr = reduce_partials(part_rr, n_part, sh);
            if (threadIdx.x == 0) {
                if (k >= 1) {                              // iterate k has not been tested yet
                    const double res = rr / sc->bb;
                    const bool conv = (res < sc->rtol_sq) || (rr < sc->atol_sq);
                    if (k < hist_cap) hist[k] = res;
                    sc->res = res;
                    sc->status = conv ? DPCG_OK : (!(res == res) ? DPCG_BREAKDOWN : DPCG_MAX_ITER);
                } else {
                    sc->status = DPCG_MAX_ITER;
                }
                sc->done = 1;
            }
        }
    }
}

void launch_final_fused(int64_t n, Scalars *scal, const double *part_rr, int n_part, double *hist, int hist_cap,
                        double *x, const double *p0, const double *p1, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_final_fused, dim3(grid), dim3(kBlock), 0, s, n, scal, part_rr, n_part, hist, hist_cap, x, p0,
                       p1);
}

void launch_update_r(int precond_fused, int64_t n, Scalars *scal, const double *part_pq, int n_part_pq,
                     const double *q, double *r, const double *dinv, double *z, double *part_rz, double *part_rr,
                     int grid, hipStream_t s) {
    if (precond_fused == 0)
        hipLaunchKernelGGL(k_update_r<0>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr);
    else if (precond_fused == 1)
        hipLaunchKernelGGL(k_update_r<1>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr);
    else
        hipLaunchKernelGGL(k_update_r<2>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_pq, n_part_pq, q, r, dinv, z,
                           part_rz, part_rr);
}

// part[b] = partial of <a,b>; skipped once the solve is done (scal may be null: always run).
__global__ __launch_bounds__(kBlock) void k_dot_partials(int64_t n, const Scalars *__restrict__ sc,
                                                         const double *__restrict__ a, const double *__restrict__ b,
                                                         double *__restrict__ part) {
    __shared__ double sh[4];
    if (sc && sc->done) return;
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc += a[i] * b[i];
    const double tot = block_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

void launch_dot_partials(int64_t n, const Scalars *scal, const double *a, const double *b, double *part, int grid,
                         hipStream_t s) {
    hipLaunchKernelGGL(k_dot_partials, dim3(grid), dim3(kBlock), 0, s, n, scal, a, b, part);
}

// x += alpha p; p = z + beta p (cg.py:79,82-83); optionally also the fp32 copy of p that the
// mixed-precision SpMV gathers.
template <bool P32>
__global__ __launch_bounds__(kBlock) void k_update_xp(int64_t n, Scalars *__restrict__ sc,
                                                      const double *__restrict__ part_rz,
                                                      const double *__restrict__ part_rr, int n_part,
                                                      const double *__restrict__ z, double *__restrict__ p,
                                                      double *__restrict__ x, float *__restrict__ p32,
                                                      double *__restrict__ hist, int hist_cap) {
    __shared__ double sh[4];
    if (sc->done) return;
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const double2 *__restrict__ z2 = reinterpret_cast<const double2 *>(z);
    double2 *__restrict__ p2 = reinterpret_cast<double2 *>(p);
    double2 *__restrict__ x2 = reinterpret_cast<double2 *>(x);
    float2 *__restrict__ f2 = reinterpret_cast<float2 *>(p32);
    double2 za = make_double2(0, 0), pa = za, xa = za;
    bool have = i < n2;
    if (have) {
        za = z2[i];
        pa = p2[i];
        xa = x2[i];
    }
    const double rz_new = reduce_partials(part_rz, n_part, sh);
    const double beta = rz_new / sc->rz;                                // cg.py:82
    const double alpha = sc->alpha;
    if (blockIdx.x == 0) {                                              // cg.py:86 + the test of cg.py:71
        const double rr = reduce_partials(part_rr, n_part, sh);
        if (threadIdx.x == 0) record_and_test(sc, rr, rz_new, hist, hist_cap, sc->k + 1);
    }
    while (have) {
        const int64_t cur = i;
        const double2 zc = za, pc = pa, xc = xa;
        i += stride;
        have = i < n2;
        if (have) {
            za = z2[i];
            pa = p2[i];
            xa = x2[i];
        }
        double2 xn, pn;
        xn.x = xc.x + alpha * pc.x;                                     // cg.py:79
        xn.y = xc.y + alpha * pc.y;
        pn.x = zc.x + beta * pc.x;                                      // cg.py:83
        pn.y = zc.y + beta * pc.y;
        x2[cur] = xn;
        p2[cur] = pn;
        if (P32) f2[cur] = make_float2((float)pn.x, (float)pn.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t e = n - 1;
        const double pe = p[e];
        x[e] = x[e] + alpha * pe;
        const double pn = z[e] + beta * pe;
        p[e] = pn;
        if (P32) p32[e] = (float)pn;
    }
}

void launch_update_xp(int64_t n, Scalars *scal, const double *part_rz, const double *part_rr, int n_part,
                      const double *z, double *p, double *x, float *p32, double *hist, int hist_cap, int grid,
                      hipStream_t s) {
    if (p32)
        hipLaunchKernelGGL(k_update_xp<true>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_rz, part_rr, n_part, z, p,
                           x, p32, hist, hist_cap);
    else
        hipLaunchKernelGGL(k_update_xp<false>, dim3(grid), dim3(kBlock), 0, s, n, scal, part_rz, part_rr, n_part, z, p,
                           x, p32, hist, hist_cap);
}

// After the last permitted update (cg.py:70 exhausted): the test has already been recorded by K3.
__global__ void k_final_check(Scalars *sc) {
    if (threadIdx.x == 0 && !sc->done) {
        sc->status = DPCG_MAX_ITER;
        sc->done = 1;
    }
}

void launch_final_check(Scalars *scal, hipStream_t s) {
    hipLaunchKernelGGL(k_final_check, dim3(1), dim3(64), 0, s, scal);
}

// Start of a solve (cg.py:62-66): p = z, partials of <b,b>, <r,z> and of the first tested quantity
// (<z,z> for the reference's quirk at cg.py:66, <r,r> for scipy-style).
template <bool P32>
__global__ __launch_bounds__(kBlock) void k_init_state(int64_t n, const double *__restrict__ b,
                                                       const double *__restrict__ r, const double *__restrict__ z,
                                                       double *__restrict__ p, float *__restrict__ p32,
                                                       double *__restrict__ part_bb, double *__restrict__ part_rz,
                                                       double *__restrict__ part_rr, int init_check_r) {
    __shared__ double sh[4];
    double a_bb = 0.0, a_rz = 0.0, a_t = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double bi = b[i], ri = r[i], zi = z[i];
        p[i] = zi;                                                      // cg.py:62
        if (P32) p32[i] = (float)zi;
        a_bb += bi * bi;
        a_rz += ri * zi;
        a_t += init_check_r ? ri * ri : zi * zi;                        // cg.py:66 tests zk
    }
    const double t_bb = block_sum(a_bb, sh);
    const double t_rz = block_sum(a_rz, sh);
    const double t_t = block_sum(a_t, sh);
    if (threadIdx.x == 0) {
        part_bb[blockIdx.x] = t_bb;
        part_rz[blockIdx.x] = t_rz;
        part_rr[blockIdx.x] = t_t;
    }
}

void launch_init_state(int64_t n, Scalars *scal, const double *b, const double *r, const double *z, double *p,
                       float *p32, double *part_bb, double *part_rz, double *part_rr, int init_check_r, int grid,
                       hipStream_t s) {
    (void)scal;
    if (p32)
        hipLaunchKernelGGL(k_init_state<true>, dim3(grid), dim3(kBlock), 0, s, n, b, r, z, p, p32, part_bb, part_rz,
                           part_rr, init_check_r);
    else
        hipLaunchKernelGGL(k_init_state<false>, dim3(grid), dim3(kBlock), 0, s, n, b, r, z, p, p32, part_bb, part_rz,
                           part_rr, init_check_r);
}

__global__ __launch_bounds__(kBlock) void k_finalize_init(Scalars *sc, const double *__restrict__ part_bb,
                                                          const double *__restrict__ part_rz,
                                                          const double *__restrict__ part_t, int n_part,
                                                          double rtol_sq, double atol_sq, double *hist, int hist_cap,
                                                          unsigned long long *progress) {
    __shared__ double sh[4];
    const double bb = reduce_partials(part_bb, n_part, sh);
    const double rz = reduce_partials(part_rz, n_part, sh);
    const double tt = reduce_partials(part_t, n_part, sh);             // <z0,z0> (cg.py:66) or <r0,r0>
    if (threadIdx.x == 0) {
        sc->bb = bb;
        sc->rz = rz;
        sc->alpha = 0.0;
        sc->rtol_sq = rtol_sq;
        sc->atol_sq = atol_sq;
        sc->done = 0;
        sc->status = DPCG_MAX_ITER;
        sc->pad = 0;
        sc->progress = progress;
        record_and_test(sc, tt, rz, hist, hist_cap, 0);                 // cg.py:66-67 and the first cg.py:71
    }
}

void launch_finalize_init(Scalars *scal, const double *part_bb, const double *part_rz, const double *part_t,
                          int n_part, double rtol_sq, double atol_sq, double *hist, int hist_cap,
                          unsigned long long *progress, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize_init, dim3(1), dim3(kBlock), 0, s, scal, part_bb, part_rz, part_t, n_part, rtol_sq,
                       atol_sq, hist, hist_cap, progress);
}

// r = b - A x0 (cg.py:60), ax = A x0 computed by the SpMV before.
__global__ __launch_bounds__(kBlock) void k_residual(int64_t n, const double *__restrict__ b,
                                                     const double *__restrict__ ax, double *__restrict__ r) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) r[i] = b[i] - ax[i];
}

void launch_residual(int64_t n, const double *b, const double *ax, double *r, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_residual, dim3(grid), dim3(kBlock), 0, s, n, b, ax, r);
}

// z = dinv .* r (Jacobi apply outside the fused path).
__global__ __launch_bounds__(kBlock) void k_scale(int64_t n, const double *__restrict__ dinv,
                                                  const double *__restrict__ r, double *__restrict__ z) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) z[i] = dinv[i] * r[i];
}

void launch_scale(int64_t n, const double *dinv, const double *r, double *z, int grid, hipStream_t s) {
    hipLaunchKernelGGL(k_scale, dim3(grid), dim3(kBlock), 0, s, n, dinv, r, z);
}

// dinv[i] = 1 / a_ii (test.py:76); flags a missing or non-positive diagonal.
__global__ __launch_bounds__(kBlock) void k_extract_dinv(int64_t n, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, double *__restrict__ dinv,
                                                         int *bad) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double d = 0.0;
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (col[k] == i) d = val[k];
        if (!(d > 0.0)) atomicExch(bad, 1);
        dinv[i] = 1.0 / d;
    }
}

void launch_extract_dinv(const CsrDev &A, double *dinv, int *bad_flag, hipStream_t s) {
    int64_t g = (A.n + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_extract_dinv, dim3((int)g), dim3(kBlock), 0, s, A.n, A.rowptr, A.col, A.val, dinv, bad_flag);
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void k_convert(int64_t n, const TI *__restrict__ in, TO *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = (TO)in[i];
}

static int conv_grid(int64_t n) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
void launch_f64_to_f32(int64_t n, const double *in, float *out, hipStream_t s) {
    hipLaunchKernelGGL((k_convert<double, float>), dim3(conv_grid(n)), dim3(kBlock), 0, s, n, in, out);
}
void launch_f32_to_f64(int64_t n, const float *in, double *out, hipStream_t s) {
    hipLaunchKernelGGL((k_convert<float, double>), dim3(conv_grid(n)), dim3(kBlock), 0, s, n, in, out);
}

// e = x - x_true (cg.py:27,43): only for conjugate_gradient(..., x_true=...).
__global__ __launch_bounds__(kBlock) void k_anorm_err(int64_t n, const Scalars *__restrict__ sc,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ x_true, double *__restrict__ e) {
    (void)sc;  // runs even when `done` is set: the iterate the test fired on still gets its error recorded
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) e[i] = x[i] - x_true[i];
}

void launch_anorm_err(int64_t n, const Scalars *scal, const double *x, const double *x_true, double *e, int grid,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_anorm_err, dim3(grid), dim3(kBlock), 0, s, n, scal, x, x_true, e);
}

// err_hist[k] = <e, A e> (cg.py:29,45) for the iterate the loop has just produced.
__global__ __launch_bounds__(kBlock) void k_record_err(const Scalars *__restrict__ sc, const double *__restrict__ part,
                                                       int n_part, double *err_hist, int hist_cap) {
    __shared__ double sh[4];
    const double v = reduce_partials(part, n_part, sh);
    if (threadIdx.x == 0 && sc->k < hist_cap) err_hist[sc->k] = v;
}

void launch_record_err(const Scalars *scal, const double *part, int n_part, double *err_hist, int hist_cap,
                       int at_k_minus_one, hipStream_t s) {
    (void)at_k_minus_one;
    hipLaunchKernelGGL(k_record_err, dim3(1), dim3(kBlock), 0, s, scal, part, n_part, err_hist, hist_cap);
}

__global__ __launch_bounds__(kBlock) void k_dot_final(const double *__restrict__ part, int n_part, double *out) {
    __shared__ double sh[4];
    const double v = reduce_partials(part, n_part, sh);
    if (threadIdx.x == 0) *out = v;
}

void launch_dot_final(const double *part, int n_part, double *out_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kBlock), 0, s, part, n_part, out_dev);
}

// ------------------------------------------------------------------------------------------------
// Level-scheduled sparse triangular solves (LLT_SOLVE: z = L^-T (L^-1 r)).
// Rows of one level are independent; one thread owns a row and subtracts its products in column
// order, then divides by the diagonal -- bit-identical to sequential substitution.
// Lower factor: diagonal LAST in the row.  Upper (L^T as CSR): diagonal FIRST.
// ------------------------------------------------------------------------------------------------
// One wide level, CSR-stream style: a workgroup takes 256 consecutive rows of the level-ordered copy, streams
// their contiguous val/col segment coalesced, parks v*out[col] in LDS and lets thread j subtract the products
// of row j in column order.  The diagonal's slot is skipped (its "product" is never read).
template <bool UPPER>
__global__ __launch_bounds__(kBlock) void k_sptrsv_level_stream(const int32_t *__restrict__ rows, int j0, int count,
                                                                const int32_t *__restrict__ lo_rp,
                                                                const int32_t *__restrict__ lo_ci,
                                                                const double *__restrict__ lo_v,
                                                                const double *__restrict__ rhs, double *out, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    constexpr int U = kStreamCap / kBlock;
    __shared__ double prod[kStreamCap];
    const int t = threadIdx.x;
    const int jb = j0 + blockIdx.x * kBlock;
    const int jend = (jb + kBlock < j0 + count) ? jb + kBlock : j0 + count;
    const int j = jb + t;
    const int base = lo_rp[jb];
    const int cnt = lo_rp[jend] - base;
    int rs = 0, re = 0, i = 0;
    double bi = 0.0;
    if (j < jend) {
        rs = lo_rp[j] - base;
        re = lo_rp[j + 1] - base;
        i = rows[j];
        bi = rhs[i];
    }
    const int last = cnt > 0 ? cnt - 1 : 0;
    int c[U];
    double a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = t + u * kBlock;
        const int kk = k < cnt ? k : last;
        c[u] = lo_ci[base + kk];
        a[u] = lo_v[base + kk];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = t + u * kBlock;
        if (k < cnt) prod[k] = a[u] * out[c[u]];
    }
    __syncthreads();
    if (j < jend) {
        double acc = bi;
        const int ks = UPPER ? rs + 1 : rs, ke = UPPER ? re : re - 1;
        for (int k = ks; k < ke; ++k) acc -= prod[k];
        out[i] = acc / lo_v[base + (UPPER ? rs : re - 1)];
    }
}

// One level, one thread per row (rows too long for the LDS product buffer).
template <bool UPPER>
__global__ __launch_bounds__(kBlock) void k_sptrsv_level(const int32_t *__restrict__ rows, int j0, int count,
                                                         const int32_t *__restrict__ lo_rp,
                                                         const int32_t *__restrict__ lo_ci,
                                                         const double *__restrict__ lo_v,
                                                         const double *__restrict__ rhs, double *out, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= count) return;
    const int j = j0 + idx, i = rows[j];
    const int s = lo_rp[j], e = lo_rp[j + 1];
    double acc = rhs[i];
    const int ks = UPPER ? s + 1 : s, ke = UPPER ? e : e - 1;
    for (int k = ks; k < ke; ++k) acc -= lo_v[k] * out[lo_ci[k]];
    out[i] = acc / lo_v[UPPER ? s : e - 1];
}

// A run of narrow levels walked by ONE workgroup of 1024 threads with a barrier between levels
// (cheaper than one launch per level: ~1.5 us boundary each).  Values handed from level to level
// go through L2 with agent-scope (sc1) accesses so no wave reads a stale line from its CU's L1.
constexpr int kMergedBlock = 1024;
template <bool UPPER>
__global__ __launch_bounds__(kMergedBlock) void k_sptrsv_merged(const int32_t *__restrict__ rows,
                                                                const int32_t *__restrict__ level_ptr, int lvl_lo,
                                                                int lvl_hi, const int32_t *__restrict__ lo_rp,
                                                                const int32_t *__restrict__ lo_ci,
                                                                const double *__restrict__ lo_v,
                                                                const double *__restrict__ rhs, double *out, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    for (int lvl = lvl_lo; lvl < lvl_hi; ++lvl) {
        const int lo = level_ptr[lvl], hi = level_ptr[lvl + 1];
        for (int j = lo + (int)threadIdx.x; j < hi; j += kMergedBlock) {
            const int i = rows[j];
            const int s = lo_rp[j], e = lo_rp[j + 1];
            double acc = rhs[i];
            const int ks = UPPER ? s + 1 : s, ke = UPPER ? e : e - 1;
            for (int k = ks; k < ke; ++k)
                acc -= lo_v[k] * __hip_atomic_load(out + lo_ci[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double res = acc / lo_v[UPPER ? s : e - 1];
            __hip_atomic_store(out + i, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();  // includes s_waitcnt vmcnt(0): this level's stores have reached L2
    }
}

// The same walk with the dependencies handed over through an LDS ring instead of L2.  In level order the
// entries a banded factor's rows depend on sit within the last W positions (for a 5-point grid: the previous
// anti-diagonal), so y lives in a ring of W doubles indexed by position & (W-1); entries solved before this
// segment are read from `out` (written by earlier kernels).  Everything that does not depend on y -- the row
// id, rhs, extents and the first three entries of the NEXT level's rows -- is loaded one level ahead, so a
// level costs an LDS round trip, a division and a barrier instead of three dependent L2 round trips.
constexpr int kRingE = 3;   // entries per row held in registers one level ahead
template <bool UPPER>
__global__ __launch_bounds__(kMergedBlock) void k_sptrsv_ring(const int32_t *__restrict__ rows,
                                                              const int32_t *__restrict__ level_ptr, int lvl_lo,
                                                              int lvl_hi, const int32_t *__restrict__ lo_rp,
                                                              const int32_t *__restrict__ lo_ci,
                                                              const int32_t *__restrict__ lo_cpos,
                                                              const double *__restrict__ lo_v,
                                                              const double *__restrict__ rhs, double *out,
                                                              int seg_start, int W, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    extern __shared__ __attribute__((aligned(16))) double ring[];
    struct Row {
        int j, i, s, e;
        double b, diag;
        int cpos[kRingE], ci[kRingE];
        double v[kRingE];
    };
    const int t = threadIdx.x;
    auto load_row = [&](Row &r, int j, int hi) {
        r.j = j < hi ? j : -1;
        if (r.j < 0) return;
        r.i = rows[j];
        r.s = lo_rp[j];
        r.e = lo_rp[j + 1];
        r.b = rhs[r.i];
        r.diag = lo_v[UPPER ? r.s : r.e - 1];
        const int ks = UPPER ? r.s + 1 : r.s, ke = UPPER ? r.e : r.e - 1;
#pragma unroll
        for (int m = 0; m < kRingE; ++m) {
            const int k = ks + m < ke ? ks + m : ks;      // clamped: an unused slot re-reads a valid entry
            r.cpos[m] = ks < ke ? lo_cpos[k] : 0;
            r.ci[m] = ks < ke ? lo_ci[k] : 0;
            r.v[m] = ks < ke ? lo_v[k] : 0.0;
        }
    };
    auto solve_row = [&](const Row &r) {
        if (r.j < 0) return;
        const int ks = UPPER ? r.s + 1 : r.s, ke = UPPER ? r.e : r.e - 1;
        double acc = r.b;
#pragma unroll
        for (int m = 0; m < kRingE; ++m)
            if (ks + m < ke) {
                const double yv = r.cpos[m] >= seg_start ? ring[r.cpos[m] & (W - 1)] : out[r.ci[m]];
                acc -= r.v[m] * yv;
            }
        for (int k = ks + kRingE; k < ke; ++k) {           // longer rows: the rest straight from memory
            const int cp = lo_cpos[k];
            const double yv = cp >= seg_start ? ring[cp & (W - 1)] : out[lo_ci[k]];
            acc -= lo_v[k] * yv;
        }
        const double y = acc / r.diag;
        ring[r.j & (W - 1)] = y;
        out[r.i] = y;
    };
    // Two register sets in ping-pong (no register moves: a move of a register with a load in flight would wait
    // for it).  Each thread owns at most two rows of a level (levels of a merged run have <= 2048 rows).
    auto load_level = [&](Row &a, Row &b, int lvl) {
        if (lvl < lvl_hi) {
            const int lo = level_ptr[lvl], hi = level_ptr[lvl + 1];
            load_row(a, lo + t, hi);
            load_row(b, lo + t + (int)blockDim.x, hi);
        } else {
            a.j = b.j = -1;
        }
    };
    // LDS-only hand-off: wait for this wave's ring writes, then the workgroup barrier.  A __syncthreads() would
    // also drain vmcnt, i.e. wait for the out[] store and for the prefetch loads that must stay in flight.
    auto level_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    Row a0, b0, a1, b1;
    load_level(a0, b0, lvl_lo);
    for (int lvl = lvl_lo; lvl < lvl_hi; lvl += 2) {
        load_level(a1, b1, lvl + 1);        // flies while level `lvl` is solved
        solve_row(a0);
        solve_row(b0);
        level_barrier();
        if (lvl + 1 >= lvl_hi) break;
        load_level(a0, b0, lvl + 2);
        solve_row(a1);
        solve_row(b1);
        level_barrier();
    }
}

// The ring walk with a deep software pipeline.  A level of a 5-point grid's factor is ~100 rows: the LDS round
// trip, three multiply-adds, the division and a barrier take ~0.2 us, but fetching a row's data only one level
// ahead (k_sptrsv_ring: row id -> extents -> entries, two dependent trips to L2) costs ~1.4 us per level.  Here a
// row's data sits in fixed-width records addressed by its level-order position alone (Levels::pk_meta / pk_val,
// right-hand side pre-gathered into level order), four 16-byte loads per row, and is requested D levels ahead;
// the segment's level offsets are staged in LDS up front.  Arithmetic and its order are those of every other
// SpTRSV kernel here (ascending columns, one product and one subtraction at a time, then the division).
__global__ __launch_bounds__(kBlock) void k_gather_lo(const int32_t *__restrict__ rows, const double *__restrict__ rhs,
                                                      double *__restrict__ b_lo, int j0, int count, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx < count) b_lo[j0 + idx] = rhs[rows[j0 + idx]];
}

// The prefetch is chunked: while the C levels of one chunk are solved out of one register set, the records of
// the next chunk fly into the other set; at a chunk boundary one s_waitcnt vmcnt(0) retires them (by then C levels
// of work have covered the memory latency).  The explicit wait + register "touch" keeps the compiler from
// inserting its own conservative vmcnt(0) at each first use, which would also wait for the loads just issued.
template <bool UPPER, int C, int ROWS>   // ROWS rows of a level per thread, C levels per prefetch chunk
__global__ __launch_bounds__(512) void k_sptrsv_ring_pipe(const int32_t *__restrict__ level_ptr, int lvl_lo,
                                                          int lvl_hi, const int32_t *__restrict__ lo_rp,
                                                          const int32_t *__restrict__ lo_ci,
                                                          const int32_t *__restrict__ lo_cpos,
                                                          const double *__restrict__ lo_v,
                                                          const int4 *__restrict__ pk_meta,
                                                          const double2 *__restrict__ pk_val,
                                                          const double *__restrict__ b_lo, double *out, int seg_start,
                                                          int W, const int *done) {
    if (done && *done) return;   // the solve has converged: the rest of the enqueued updates are no-ops
    extern __shared__ __attribute__((aligned(16))) double ring[];
    int *lp = reinterpret_cast<int *>(ring + W);        // level offsets of this segment, padded with empty levels
    const int t = threadIdx.x, T = blockDim.x;
    const int nl = lvl_hi - lvl_lo;
    const int nchunks = (nl + C - 1) / C;
    const int seg_end = level_ptr[lvl_hi];
    for (int i = t; i <= (nchunks + 2) * C; i += T) lp[i] = i <= nl ? level_ptr[lvl_lo + i] : seg_end;
    __syncthreads();
    const int jmax = seg_end - 1;
    struct Row {
        int j;           // level-order position, -1 for a lane without a row in this level 
        int4 m;          // cpos0..2, original row
        double2 v01, v2d;
        double b;
    };
    auto load_row = [&](Row &r, int j, int hi) {
        const int jc = j < hi ? j : jmax;                 // lanes without a row load a valid record and ignore it
        r.j = j < hi ? j : -1;
        r.m = pk_meta[jc];
        r.v01 = pk_val[2 * (int64_t)jc];
        r.v2d = pk_val[2 * (int64_t)jc + 1];
        r.b = b_lo[jc];
    };
    auto load_chunk = [&](Row (&S)[C][ROWS], int chunk) {
#pragma unroll
        for (int d = 0; d < C; ++d) {
            const int rel = chunk * C + d;
            const int lo = lp[rel], hi = lp[rel + 1];
#pragma unroll
            for (int h = 0; h < ROWS; ++h) load_row(S[d][h], lo + t + h * T, hi);
        }
    };
    // all outstanding loads have landed; tell the compiler so (the registers are "redefined" here)
    auto retire = [&](Row (&S)[C][ROWS]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < C; ++d)
#pragma unroll
            for (int h = 0; h < ROWS; ++h) {
                Row &r = S[d][h];
                asm volatile("" : "+v"(r.m.x), "+v"(r.m.y), "+v"(r.m.z), "+v"(r.m.w), "+v"(r.v01.x), "+v"(r.v01.y),
                             "+v"(r.v2d.x), "+v"(r.v2d.y), "+v"(r.b));
            }
    };
    auto solve_row = [&](const Row &r) {
        const bool valid = r.j >= 0;
        double acc = r.b;
        if (valid && r.m.x == -2) {                       // long row, or one that reaches back before the segment
            const int s = lo_rp[r.j], e = lo_rp[r.j + 1];
            const int ks = UPPER ? s + 1 : s, ke = UPPER ? e : e - 1;
            for (int k = ks; k < ke; ++k) {
                const int cp = lo_cpos[k];
                const double yv = cp >= seg_start ? ring[cp & (W - 1)] : out[lo_ci[k]];
                acc -= lo_v[k] * yv;
            }
        } else {
            const double y0 = ring[(r.m.x < 0 ? 0 : r.m.x) & (W - 1)];
            const double y1 = ring[(r.m.y < 0 ? 0 : r.m.y) & (W - 1)];
            const double y2 = ring[(r.m.z < 0 ? 0 : r.m.z) & (W - 1)];
            if (r.m.x >= 0) acc -= r.v01.x * y0;
            if (r.m.y >= 0) acc -= r.v01.y * y1;
            if (r.m.z >= 0) acc -= r.v2d.x * y2;
        }
        const double y = acc / r.v2d.y;
        if (valid) {
            ring[r.j & (W - 1)] = y;
            out[r.m.w] = y;
        }
    };
    // LDS-only hand-off (see k_sptrsv_ring): no vmcnt drain, the prefetched loads stay in flight across it
    auto level_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto solve_chunk = [&](Row (&S)[C][ROWS]) {
#pragma unroll
        for (int d = 0; d < C; ++d) {
#pragma unroll
            for (int h = 0; h < ROWS; ++h) solve_row(S[d][h]);
            level_barrier();
        }
    };
    Row S0[C][ROWS], S1[C][ROWS];
    load_chunk(S0, 0);
    for (int c = 0; c < nchunks; c += 2) {
        retire(S0);
        load_chunk(S1, c + 1);                            // flies while chunk c is solved
        solve_chunk(S0);
        if (c + 1 >= nchunks) break;
        retire(S1);
        load_chunk(S0, c + 2);
        solve_chunk(S1);
    }
}

constexpr int kRingChunk = 6;   // levels per prefetch chunk of k_sptrsv_ring_pipe (3 with two rows per thread)
static bool ring_pipe_disabled() {
    static const bool off = [] { const char *e = getenv("DPCG_RING_PIPE"); return e && e[0] == '0'; }();
    return off;
}

void launch_sptrsv(const CsrDev &T, const Levels &lv, bool upper, const double *rhs, double *out, hipStream_t s,
                   const int *done) {
    (void)T;  // the level-ordered copy in `lv` carries the factor
    for (const auto &seg : lv.segments) {
        if (seg.merged && seg.ring_w > 0 && lv.pk_meta && !ring_pipe_disabled() && seg.max_width <= 1024 &&
            (size_t)seg.ring_w * sizeof(double) + (size_t)(seg.hi - seg.lo + 4 * kRingChunk) * sizeof(int) <= 64 * 1024) {
            const int seg_start = lv.level_ptr[seg.lo], seg_rows = lv.level_ptr[seg.hi] - seg_start;
            const size_t lds = (size_t)seg.ring_w * sizeof(double) + (size_t)(seg.hi - seg.lo + 4 * kRingChunk) * sizeof(int);
            const int width = seg.max_width;
            hipLaunchKernelGGL(k_gather_lo, dim3((seg_rows + kBlock - 1) / kBlock), dim3(kBlock), 0, s, lv.rows, rhs,
                               lv.b_lo, seg_start, seg_rows, done);
#define DPCG_RING_PIPE(UP, CV, ROWSV, threads)                                                                      \
    hipLaunchKernelGGL((k_sptrsv_ring_pipe<UP, CV, ROWSV>), dim3(1), dim3(threads), lds, s, lv.level_ptr_dev, seg.lo,  \
                       seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, (const int4 *)lv.pk_meta,            \
                       (const double2 *)lv.pk_val, lv.b_lo, out, seg_start, seg.ring_w, done)
            if (width <= 512) {                              // one row per thread
                int threads = (width + 63) / 64 * 64;
                threads = threads < 64 ? 64 : threads;
                if (upper) DPCG_RING_PIPE(true, kRingChunk, 1, threads);
                else DPCG_RING_PIPE(false, kRingChunk, 1, threads);
            } else {                                         // two rows per thread
                const int threads = ((width + 1) / 2 + 63) / 64 * 64;
                if (upper) DPCG_RING_PIPE(true, kRingChunk / 2, 2, threads);
                else DPCG_RING_PIPE(false, kRingChunk / 2, 2, threads);
            }
#undef DPCG_RING_PIPE
            continue;
        }
        if (seg.merged && seg.ring_w > 0) {
            const size_t lds = (size_t)seg.ring_w * sizeof(double);
            const int seg_start = lv.level_ptr[seg.lo];
            // as few waves as the widest level needs (two rows per thread): a barrier among 4 waves is cheaper
            int width = 0;
            for (int q = seg.lo; q < seg.hi; ++q) width = std::max(width, lv.level_ptr[q + 1] - lv.level_ptr[q]);
            int threads = ((width + 1) / 2 + 63) / 64 * 64;
            threads = threads < 64 ? 64 : (threads > kMergedBlock ? kMergedBlock : threads);
            if (upper)
                hipLaunchKernelGGL(k_sptrsv_ring<true>, dim3(1), dim3(threads), lds, s, lv.rows, lv.level_ptr_dev,
                                   seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, rhs, out, seg_start,
                                   seg.ring_w, done);
            else
                hipLaunchKernelGGL(k_sptrsv_ring<false>, dim3(1), dim3(threads), lds, s, lv.rows, lv.level_ptr_dev,
                                   seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_cpos, lv.lo_val, rhs, out, seg_start,
                                   seg.ring_w, done);
            continue;
        }
        if (seg.merged) {
            if (upper)
                hipLaunchKernelGGL(k_sptrsv_merged<true>, dim3(1), dim3(kMergedBlock), 0, s, lv.rows, lv.level_ptr_dev,
                                   seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_val, rhs, out, done);
            else
                hipLaunchKernelGGL(k_sptrsv_merged<false>, dim3(1), dim3(kMergedBlock), 0, s, lv.rows,
                                   lv.level_ptr_dev, seg.lo, seg.hi, lv.lo_rowptr, lv.lo_col, lv.lo_val, rhs, out, done);
            continue;
        }
        for (int l = seg.lo; l < seg.hi; ++l) {
            const int j0 = lv.level_ptr[l], cnt = lv.level_ptr[l + 1] - j0;
            const int grid = (cnt + kBlock - 1) / kBlock;
#define DPCG_TRSV(KERNEL, UP) \
    hipLaunchKernelGGL(KERNEL<UP>, dim3(grid), dim3(kBlock), 0, s, lv.rows, j0, cnt, lv.lo_rowptr, lv.lo_col, lv.lo_val, rhs, out, done)
            if (lv.stream_ok) {
                if (upper) DPCG_TRSV(k_sptrsv_level_stream, true);
                else DPCG_TRSV(k_sptrsv_level_stream, false);
            } else {
                if (upper) DPCG_TRSV(k_sptrsv_level, true);
                else DPCG_TRSV(k_sptrsv_level, false);
            }
#undef DPCG_TRSV
        }
    }
}

// ------------------------------------------------------------------------------------------------
// IC(0) numeric factorisation, level by level (stands in for ilupp.ichol0, test.py:83).
// Row i of L needs the finished rows j < i of its own pattern -- the dependency DAG of the lower
// solve, so the same level sets apply.  One thread owns a row; every sum runs over ascending columns,
// one product and one subtraction at a time (two roundings): the order of the CPU restatement, so the
// factor is bit-identical to it.  lv holds tril(A) on entry and L on exit (diagonal last in a row).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_ic0_level(const int32_t *__restrict__ rows, int j0, int count,
                                                      const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                      double *lv, int *bad) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= count) return;
    const int i = rows[j0 + idx];
    const int s_i = rp[i], e_i = rp[i + 1];
    for (int k = s_i; k < e_i; ++k) {
        const int j = ci[k];
        const int s_j = rp[j], e_j = rp[j + 1];
        double acc = lv[k];
        int a = s_i, b = s_j;
        while (a < k && b < e_j - 1) {
            const int ca = ci[a], cb = ci[b];
            if (ca == cb) {
                acc -= lv[a] * lv[b];
                ++a;
                ++b;
            } else if (ca < cb) ++a;
            else ++b;
        }
        if (j < i) lv[k] = acc / lv[e_j - 1];
        else {
            if (!(acc > 0.0)) atomicExch(bad, i + 1);
            lv[k] = sqrt(acc);
        }
    }
}

void launch_ic0_level(const int32_t *rows, int j0, int count, const int32_t *rp, const int32_t *ci, double *lv, int *bad,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_ic0_level, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, rows, j0, count, rp, ci, lv,
                       bad);
}

// ------------------------------------------------------------------------------------------------
// Setup helpers
// ------------------------------------------------------------------------------------------------
// max over row-blocks of the non-zeros in `rows_per_block` consecutive rows (stream-kernel test).
__global__ __launch_bounds__(kBlock) void k_block_nnz_max(int64_t n, const int32_t *__restrict__ rowptr,
                                                          int rows_per_block, int *out_max) {
    const int64_t nrb = (n + rows_per_block - 1) / rows_per_block;
    int m = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t rb = (int64_t)blockIdx.x * kBlock + threadIdx.x; rb < nrb; rb += stride) {
        const int64_t r0 = rb * rows_per_block;
        const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
        const int c = rowptr[r1] - rowptr[r0];
        m = c > m ? c : m;
    }
    atomicMax(out_max, m);
}

void launch_block_nnz_max(const CsrDev &A, int rows_per_block, int *out_max_dev, hipStream_t s) {
    const int64_t nrb = (A.n + rows_per_block - 1) / rows_per_block;
    int64_t g = (nrb + kBlock - 1) / kBlock;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_block_nnz_max, dim3((int)g), dim3(kBlock), 0, s, A.n, A.rowptr, rows_per_block, out_max_dev);
}

// Closed-form 5-point / 7-point Poisson CSR (kron(I,T)+kron(T,I)[+...], T = tridiag(-1,2,-1)),
// one thread per row, columns ascending.  Generated in HBM so the 256^3 systems (1.4 GB each) never
// cross PCIe.
template <typename VT>
__global__ __launch_bounds__(kBlock) void k_gen_poisson(int dim, int64_t n, int32_t *__restrict__ rowptr,
                                                        int32_t *__restrict__ col, VT *__restrict__ val) {
    const int64_t n2 = n * n;
    const int64_t N = dim == 2 ? n2 : n2 * n;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= N; i += stride) {
        // non-zeros before row i: full stencil minus the neighbours cut off by each face
        int64_t before;
        if (dim == 2) {
            const int64_t lo_y = i < n ? i : n;                         // rows with iy == 0
            const int64_t hi_y = i > n * (n - 1) ? i - n * (n - 1) : 0; // rows with iy == n-1
            before = 5 * i - lo_y - hi_y - (i + n - 1) / n - i / n;
        } else {
            const int64_t lo_z = i < n2 ? i : n2;
            const int64_t hi_z = i > n2 * (n - 1) ? i - n2 * (n - 1) : 0;
            const int64_t planes = i / n2, rem = i % n2;
            const int64_t lo_y = planes * n + (rem < n ? rem : n);
            const int64_t hi_y = planes * n + (rem > n * (n - 1) ? rem - n * (n - 1) : 0);
            before = 7 * i - lo_z - hi_z - lo_y - hi_y - (i + n - 1) / n - i / n;
        }
        rowptr[i] = (int32_t)before;
        if (i == N) continue;
        int64_t k = before;
        const int64_t ix = i % n, iy = (i / n) % n, iz = i / n2;
        if (dim == 3 && iz > 0) { col[k] = (int32_t)(i - n2); val[k++] = (VT)-1; }
        if (iy > 0) { col[k] = (int32_t)(i - n); val[k++] = (VT)-1; }
        if (ix > 0) { col[k] = (int32_t)(i - 1); val[k++] = (VT)-1; }
        col[k] = (int32_t)i;
        val[k++] = (VT)(dim == 2 ? 4 : 6);
        if (ix < n - 1) { col[k] = (int32_t)(i + 1); val[k++] = (VT)-1; }
        if (iy < n - 1) { col[k] = (int32_t)(i + n); val[k++] = (VT)-1; }
        if (dim == 3 && iz < n - 1) { col[k] = (int32_t)(i + n2); val[k++] = (VT)-1; }
    }
}

void launch_gen_poisson(int dim, int64_t n, int32_t *rowptr, int32_t *col, void *val, int val_dtype, hipStream_t s) {
    const int64_t N = dim == 2 ? n * n : n * n * n;
    int64_t g = (N + 1 + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;
    if (val_dtype == DPCG_F32)
        hipLaunchKernelGGL(k_gen_poisson<float>, dim3((int)g), dim3(kBlock), 0, s, dim, n, rowptr, col, (float *)val);
    else
        hipLaunchKernelGGL(k_gen_poisson<double>, dim3((int)g), dim3(kBlock), 0, s, dim, n, rowptr, col,
                           (double *)val);
}

// sparse_matvec_mul (utils.py:26-41): out[b, row] += feature * vec[b, col] over COO triples.
// One lane per triple, fp32 atomics on the output -- the same scatter-add the reference's own
// CUDA path performs (torch scatter_reduce on a GPU tensor is an atomicAdd).
__global__ __launch_bounds__(kBlock) void k_batched_coo_spmv(int64_t nnz, const int32_t *__restrict__ idx,
                                                             const float *__restrict__ feat, int batch, int64_t dof,
                                                             const float *__restrict__ vec, float *__restrict__ out,
                                                             int transpose) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride) {
        const int b = idx[3 * k];
        const int r = idx[3 * k + (transpose ? 2 : 1)];                 // utils.py:27
        const int c = idx[3 * k + (transpose ? 1 : 2)];                 // utils.py:28
        if (b < 0 || b >= batch || r < 0 || r >= dof || c < 0 || c >= dof) continue;
        atomicAdd(out + (int64_t)b * dof + r, feat[k] * vec[(int64_t)b * dof + c]);   // utils.py:32,36-41
    }
}

// Per-triple products out[k] = a[b, row_k] * c[b, col_k]: the gradient of sparse_matvec_mul with respect to the
// matrix entries (d/d feature_k of sum_b <g_b, A_b v_b> = g[b,row_k] * v[b,col_k]); needed to train through
// `frobenius_loss` (metrics.py:28-29).
__global__ __launch_bounds__(kBlock) void k_batched_coo_edge(int64_t nnz, const int32_t *__restrict__ idx, int batch,
                                                             int64_t dof, const float *__restrict__ a,
                                                             const float *__restrict__ c, float *__restrict__ out,
                                                             int transpose) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride) {
        const int b = idx[3 * k];
        const int r = idx[3 * k + (transpose ? 2 : 1)];
        const int cc = idx[3 * k + (transpose ? 1 : 2)];
        const bool ok = b >= 0 && b < batch && r >= 0 && r < dof && cc >= 0 && cc < dof;
        out[k] = ok ? a[(int64_t)b * dof + r] * c[(int64_t)b * dof + cc] : 0.0f;
    }
}

void launch_batched_coo_edge(int64_t nnz, const int32_t *indices, int batch, int64_t dof, const float *a, const float *c,
                             float *out, int transpose, hipStream_t s) {
    int64_t g = (nnz + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_batched_coo_edge, dim3((int)g), dim3(kBlock), 0, s, nnz, indices, batch, dof, a, c, out,
                       transpose);
}

void launch_batched_coo_spmv(int64_t nnz, const int32_t *indices, const float *features, int batch, int64_t dof,
                             const float *vectors, float *out, int transpose, hipStream_t s) {
    int64_t g = (nnz + kBlock - 1) / kBlock;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_batched_coo_spmv, dim3((int)g), dim3(kBlock), 0, s, nnz, indices, features, batch, dof,
                       vectors, out, transpose);
}

}  // namespace dpcg
