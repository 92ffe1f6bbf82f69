// Sparse-convolution forward of the reference's PreconditionerNet (uibk/deep_preconditioning/model.py:13-59) -- the CNN
// whose output IS the preconditioner's L factor -- hand-written for gfx950 (MI355X, wave64).  spconv, which the
// reference uses, ships CUDA only (pyproject.toml:20); nothing of it is used or translated here.
//
// The "image" is the matrix itself: one active site per stored entry (batch, row, col) of tril(A).  A regular sparse
// convolution (spconv.SparseConv2d, stride 1) activates every output site whose window holds an input site
// (model.py:27,36,40).  Two phases:
//
//   plan (once per sparsity pattern: dpcg_convnet_plan_create).  Every layer's active sites are kept as CSR over the image
//     rows (rowptr over batch * height rows, sorted columns) -- the image's rows are the matrix's rows, so a layer's site
//     set is built row by row: one thread per OUTPUT row merges the <= kh * kw shifted input rows that can feed it (short
//     sorted lists: count, scan, fill).  The fill pass also writes the rulebook in OUTPUT-stationary form: nbr[o][k] = the
//     input site that reaches output site o through kernel offset k, or -1.  No sort, no hash table, no atomics.
//   forward (dpcg_convnet_forward).  out[o, :] = bias + sum_k in[nbr[o][k], :] W_k: a gathered GEMM with M = sites,
//     K = kh * kw * C_in, N = C_out, on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and sums,
//     16 sites x 16 channels per instruction; the four K-quarters of the instruction are the four offsets of a 2 x 2
//     window, so a lane's A operand is ONE neighbour's feature row, read as 16-byte loads).  Weights staged once per
//     workgroup in LDS in B-operand order; bias and PReLU (model.py:28,37) fused into the epilogue; no atomics (each output
//     row has one writer), so the result is bitwise reproducible.  The last layer (pointwise, one output channel,
//     model.py:40) is fused with the post-processing of model.py:53-57 -- strict upper part zeroed, softplus on the
//     diagonal -- and writes the factor straight into the lower-triangular CSR (fp64 values) that LLtMultiply consumes:
//     nothing is densified (test.py:103-104 densifies), nothing sorted.
#include <algorithm>
#include <vector>

#include "dpcg_host.h"
#include "dpcg_prims.h"

namespace dpcg {
namespace {

constexpr int kMaxTaps = 4;          // kernel windows up to 2 x 2 (the reference's layers: 1 x 1 and 2 x 2)
constexpr int kConvMaxLayers = 16;

struct LayerPlan {
    int kh = 1, kw = 1, ph = 0, pw = 0;
    int64_t h_in = 0, w_in = 0, h_out = 0, w_out = 0;
    bool same_sites = true;          // pointwise, no padding: the site set of the input
    int64_t sites = 0;               // active output sites
    int32_t *rowptr = nullptr;       // [batch * h_out + 1]   (owned unless same_sites)
    int32_t *col = nullptr;          // [sites]
    int32_t *nbr = nullptr;          // [sites * kh * kw] input site per kernel offset, -1 = none (null when same_sites)
};

}  // namespace
}  // namespace dpcg

struct dpcg_convnet_plan {
    int batch = 0, n_layers = 0;
    int64_t height = 0, width = 0, nnz_in = 0;
    int32_t *rowptr0 = nullptr, *col0 = nullptr;      // the input's sites as CSR over batch * height rows
    dpcg::LayerPlan layers[dpcg::kConvMaxLayers];
    // output pattern: row of every site, and the lower-triangular CSR (col <= row) the factor is emitted into
    int32_t *site_row = nullptr;                      // [sites_out] image row (0 .. h_out) of the site
    int32_t *site_batch = nullptr;                    // [sites_out]
    int32_t *lower_rowptr = nullptr;                  // [batch * h_out + 1]
    int32_t *lower_col = nullptr;                     // [nnz_lower]
    int32_t *lower_pos = nullptr;                     // [sites_out] position in the lower CSR, -1 for col > row
    int64_t nnz_lower = 0;
    float *buf[2] = {nullptr, nullptr};               // ping-pong feature buffers, grown on demand
    int64_t buf_cap[2] = {0, 0};
    // Every array above lives in one of these slabs, handed out in a fixed order; dpcg_convnet_plan_rebuild draws from them
    // again (a slab grows only when the new pattern needs more), so a stream of similar matrices -- one plan per matrix --
    // costs no device allocations after the first (35 hipMalloc / hipFree pairs were 2.9 of a plan's 3.2 ms).
    std::vector<std::pair<void *, size_t>> slabs;
    size_t slab_cursor = 0;
};

namespace dpcg {
namespace {

template <typename T>
int plan_alloc(dpcg_convnet_plan *p, T **out, int64_t count) {
    const size_t bytes = (size_t)(count < 1 ? 1 : count) * sizeof(T);
    *out = nullptr;
    if (p->slab_cursor < p->slabs.size()) {
        auto &sl = p->slabs[p->slab_cursor];
        if (sl.second < bytes) {
            (void)device_free(sl.first);
            sl = {nullptr, 0};
            const size_t grown = bytes + bytes / 4;
            if (hipMalloc(&sl.first, grown) != hipSuccess) {
                set_error("dpcg_convnet: device allocation failed");
                return DPCG_ERR_NOMEM;
            }
            sl.second = grown;
        }
        *out = reinterpret_cast<T *>(sl.first);
        ++p->slab_cursor;
        return DPCG_OK;
    }
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) {
        set_error("dpcg_convnet: device allocation failed");
        return DPCG_ERR_NOMEM;
    }
    p->slabs.emplace_back(q, bytes);
    ++p->slab_cursor;
    *out = reinterpret_cast<T *>(q);
    return DPCG_OK;
}

inline int grid_rows(int64_t n, int cap = 4096) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---- plan kernels ----------------------------------------------------------------------------------------------
// Input sites (nnz, 3) = (batch, row, col), sorted by (batch, row, col): row pointers over batch * H rows, the column
// array, and a check of the order.  Thread i owns site i and fills the row pointers of the rows that START at or before
// it and after the previous site's row (empty rows in between included).
__global__ __launch_bounds__(kBlock) void k_sites_to_csr(int64_t nnz, const int32_t *__restrict__ idx, int batch, int64_t H,
                                                         int64_t W, int32_t *__restrict__ rowptr, int32_t *__restrict__ col,
                                                         int *bad) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t rows = (int64_t)batch * H;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += stride) {
        const int b = idx[3 * i], y = idx[3 * i + 1], x = idx[3 * i + 2];
        if (b < 0 || b >= batch || y < 0 || y >= H || x < 0 || x >= W) {
            atomicExch(bad, 1);
            continue;
        }
        const int64_t r = (int64_t)b * H + y;
        int64_t rprev = -1;
        if (i > 0) {
            const int pb = idx[3 * i - 3], py = idx[3 * i - 2], px = idx[3 * i - 1];
            rprev = (int64_t)pb * H + py;
            if (rprev > r || (rprev == r && px >= x)) atomicExch(bad, 2);      // not sorted / duplicate site
        }
        col[i] = x;
        for (int64_t rr = rprev + 1; rr <= r; ++rr) rowptr[rr] = (int32_t)i;
        if (i == nnz - 1)
            for (int64_t rr = r + 1; rr <= rows; ++rr) rowptr[rr] = (int32_t)nnz;
    }
}

struct ConvGeom {
    int kh, kw, ph, pw;
    int batch;
    int64_t h_in, w_in, h_out, w_out;
};

// One thread per OUTPUT row: merge of the kh * kw shifted input rows that feed it.  Input site (y, x) reaches output site
// (y - ky + ph, x - kx + pw) through W[ky, kx] (out(oy, ox) = sum in(oy + ky - ph, ox + kx - pw) W[ky, kx]).
// FILL = false: len[r] = distinct output columns.  FILL = true: columns and rulebook at rowptr_out[r].
template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_conv_rows(ConvGeom g, const int32_t *__restrict__ rp_in,
                                                      const int32_t *__restrict__ col_in, int32_t *__restrict__ len,
                                                      const int32_t *__restrict__ rp_out, int32_t *__restrict__ col_out,
                                                      int32_t *__restrict__ nbr) {
    const int64_t rows_out = (int64_t)g.batch * g.h_out;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int taps = g.kh * g.kw;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r <= rows_out; r += stride) {
        if (r == rows_out) {
            if (!FILL) len[r] = 0;
            continue;
        }
        const int64_t b = r / g.h_out, oy = r - b * g.h_out;
        int cur[kMaxTaps], end[kMaxTaps], shift[kMaxTaps];
#pragma unroll
        for (int t = 0; t < kMaxTaps; ++t) {
            cur[t] = end[t] = 0;
            shift[t] = 0;
            if (t < taps) {
                const int ky = t / g.kw, kx = t - ky * g.kw;
                const int64_t y = oy + ky - g.ph;
                if (y >= 0 && y < g.h_in) {
                    cur[t] = rp_in[b * g.h_in + y];
                    end[t] = rp_in[b * g.h_in + y + 1];
                    shift[t] = g.pw - kx;
                    while (cur[t] < end[t] && col_in[cur[t]] + shift[t] < 0) ++cur[t];      // clipped on the left
                }
            }
        }
        int count = 0;
        int64_t at = FILL ? rp_out[r] : 0;
        for (;;) {
            int best = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < kMaxTaps; ++t)
                if (cur[t] < end[t]) {
                    const int v = col_in[cur[t]] + shift[t];
                    best = v < best ? v : best;
                }
            if (best == 0x7fffffff || best >= g.w_out) break;      // exhausted, or clipped on the right (sorted)
            if (FILL) col_out[at] = best;
#pragma unroll
            for (int t = 0; t < kMaxTaps; ++t) {
                int src = -1;
                if (cur[t] < end[t] && col_in[cur[t]] + shift[t] == best) src = cur[t]++;
                if (FILL && t < taps) nbr[at * taps + t] = src;
            }
            ++at;
            ++count;
        }
        if (!FILL) len[r] = count;
    }
}

// rows of the sites, and the lower-triangular part (col <= row: a prefix of every sorted row)
__global__ __launch_bounds__(kBlock) void k_lower_count(int64_t rows, int64_t H, const int32_t *__restrict__ rp,
                                                        const int32_t *__restrict__ col, int32_t *__restrict__ len,
                                                        int32_t *__restrict__ site_row, int32_t *__restrict__ site_batch) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r <= rows; r += stride) {
        if (r == rows) {
            len[r] = 0;
            continue;
        }
        const int y = (int)(r % H), b = (int)(r / H);
        int c = 0;
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
            site_row[k] = y;
            site_batch[k] = b;
            c += col[k] <= y ? 1 : 0;
        }
        len[r] = c;
    }
}

__global__ __launch_bounds__(kBlock) void k_lower_fill(int64_t rows, int64_t H, const int32_t *__restrict__ rp,
                                                       const int32_t *__restrict__ col, const int32_t *__restrict__ lrp,
                                                       int32_t *__restrict__ lcol, int32_t *__restrict__ lpos) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < rows; r += stride) {
        const int y = (int)(r % H);
        int at = lrp[r];
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
            if (col[k] <= y) {
                lcol[at] = col[k];
                lpos[k] = at++;
            } else {
                lpos[k] = -1;
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_out_indices(int64_t sites, const int32_t *__restrict__ site_batch,
                                                        const int32_t *__restrict__ site_row, const int32_t *__restrict__ col,
                                                        int32_t *__restrict__ idx) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < sites; i += stride) {
        idx[3 * i] = site_batch[i];
        idx[3 * i + 1] = site_row[i];
        idx[3 * i + 2] = col[i];
    }
}

// ---- forward kernels -------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float prelu(float v, float slope) { return v >= 0.f ? v : slope * v; }

// 2 x 2 sparse convolution as a gathered GEMM on v_mfma_f32_16x16x4_f32.  One wave = 16 output sites x COUT channels.
// Lane l = (m = l & 15, q = l >> 4): A operand = feature row of neighbour q of site m (K-quarter q = kernel offset q),
// B operand = W_q[ci][co] from LDS.  D[(l >> 4) * 4 + i][l & 15] in register i of the accumulator.
// LDS image of the weights: quarter q at q * QSTRIDE (QSTRIDE = CIN * COUT + 16 * NB floats: the pad puts the quarters that
// one LDS lane group reads on disjoint banks), inside a quarter [ci][m][nb] so that a lane reads its NB = COUT / 16
// B values of one k-step as one 4 * NB-byte word.
template <int CIN, int COUT, bool ACT>
__global__ __launch_bounds__(kBlock) void k_sconv2x2_mfma(int64_t n_out, const int32_t *__restrict__ nbr,
                                                          const float *__restrict__ in, const float *__restrict__ w,
                                                          const float *__restrict__ bias, const float *__restrict__ slope_p,
                                                          float *__restrict__ out) {
    constexpr int NB = COUT / 16;
    constexpr int QSTRIDE = CIN * COUT + 16 * NB;
    __shared__ __attribute__((aligned(16))) float wl[4 * QSTRIDE];
    // w is KRSC: w[co][q][ci]
    for (int e = threadIdx.x; e < COUT * 4 * CIN; e += kBlock) {
        const int co = e / (4 * CIN), rem = e - co * 4 * CIN, q = rem / CIN, ci = rem - q * CIN;
        wl[q * QSTRIDE + ci * COUT + (co & 15) * NB + (co >> 4)] = w[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    const float slope = ACT ? slope_p[0] : 0.f;
    float bias_r[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias_r[nb] = bias ? bias[nb * 16 + m] : 0.f;
    const float *wq = wl + q * QSTRIDE + m * NB;
    const int64_t n_tiles = (n_out + 15) >> 4;
    for (int64_t tile = (int64_t)blockIdx.x * (kBlock / 64) + wave; tile < n_tiles; tile += (int64_t)gridDim.x * (kBlock / 64)) {
        const int64_t o = tile * 16 + m;
        const int src = o < n_out ? nbr[o * 4 + q] : -1;
        // unconditional 16-byte loads from a clamped row; an absent neighbour contributes zeros (multiplied away below).
        // K is walked in chunks of 16 channels: the next chunk's four loads are in flight while this chunk's 16 * NB
        // matrix instructions issue, and only one chunk of B values is live (a fully unrolled K loop lets the compiler
        // hoist every LDS read: 212 VGPRs at 64 -> 32 channels, two waves per SIMD).
        const f32x4 *__restrict__ p = reinterpret_cast<const f32x4 *>(in + (int64_t)(src < 0 ? 0 : src) * CIN);
        const float keep = src < 0 ? 0.f : 1.f;
        // NB == 1: two accumulator chains (even / odd k) keep the matrix pipe issuing every 32 cycles (40-cycle dependent latency)
        constexpr int CH = NB == 1 ? 2 : NB;
        f32x4 acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 cur[4], nxt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = p[u];
#pragma unroll 1
        for (int c0 = 0; c0 < CIN; c0 += 16) {
            if (c0 + 16 < CIN) {
#pragma unroll
                for (int u = 0; u < 4; ++u) nxt[u] = p[(c0 + 16) / 4 + u];
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = c0 + jj;
                const float a = cur[jj >> 2][jj & 3] * keep;
                float bv[NB];
                if constexpr (NB == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(wq + j * COUT);
                    bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
                } else if constexpr (NB == 2) {
                    const float2 t = *reinterpret_cast<const float2 *>(wq + j * COUT);
                    bv[0] = t.x; bv[1] = t.y;
                } else {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) bv[nb] = wq[j * COUT + nb];
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = NB == 1 ? (jj & 1) : nb;
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nb], acc[c], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = tile * 16 + q * 4 + i;
                float v = (NB == 1 ? acc[0][i] + acc[1][i] : acc[nb][i]) + bias_r[nb];
                if (ACT) v = prelu(v, slope);
                if (row < n_out) out[row * COUT + nb * 16 + m] = v;
            }
        }
    }
}

// Any window up to 2 x 2, any channel counts (the first layer has ONE input channel; odd sizes): one thread per
// (site, output channel), plain fp32 FMAs in the order offset-major, channel-minor.
template <bool ACT>
__global__ __launch_bounds__(kBlock) void k_sconv_generic(int64_t n_out, int taps, const int32_t *__restrict__ nbr, int cin,
                                                          int cout, const float *__restrict__ in, const float *__restrict__ w,
                                                          const float *__restrict__ bias, const float *__restrict__ slope_p,
                                                          float *__restrict__ out) {
    const float slope = ACT ? slope_p[0] : 0.f;
    const int64_t total = n_out * cout, stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t o = e / cout;
        const int co = (int)(e - o * cout);
        float s = bias ? bias[co] : 0.f;
        for (int t = 0; t < taps; ++t) {
            const int64_t src = nbr ? nbr[o * taps + t] : o;
            if (src < 0) continue;
            const float *__restrict__ f = in + src * cin;
            const float *__restrict__ wk = w + ((int64_t)co * taps + t) * cin;
            for (int ci = 0; ci < cin; ++ci) s = fmaf(f[ci], wk[ci], s);
        }
        out[e] = ACT ? prelu(s, slope) : s;
    }
}

// Last layer (pointwise, one output channel, model.py:40) + model.py:53-57: strict upper part zeroed, softplus on the
// diagonal (torch's: x > 20 ? x : log1p(exp(x))); writes the site's feature and, for col <= row, the fp64 value of L in
// the lower-triangular CSR.  One thread per site, the feature row read as 16-byte loads.
template <int CIN>
__global__ __launch_bounds__(kBlock) void k_sconv_final(int64_t sites, const float *__restrict__ in, const float *__restrict__ w,
                                                        const float *__restrict__ bias, const int32_t *__restrict__ site_row,
                                                        const int32_t *__restrict__ col, const int32_t *__restrict__ lpos,
                                                        int post, float *__restrict__ feat_out, double *__restrict__ lower_val) {
    float wr[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) wr[c] = w[c];
    const float b0 = bias ? bias[0] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < sites; i += stride) {
        const f32x4 *__restrict__ p = reinterpret_cast<const f32x4 *>(in + i * CIN);
        float s = b0;
#pragma unroll
        for (int u = 0; u < CIN / 4; ++u) {
            const f32x4 v = p[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(v[e], wr[4 * u + e], s);
        }
        if (post) {
            const int y = site_row[i], x = col[i];
            if (y < x) s = 0.f;                                                         // model.py:53-54
            else if (y == x) s = s > 20.f ? s : log1pf(expf(s));                        // model.py:56-57
        }
        if (feat_out) feat_out[i] = s;
        if (lower_val) {
            const int at = lpos[i];
            if (at >= 0) lower_val[at] = (double)s;
        }
    }
}

template <int CIN, int COUT>
void launch_mfma(int64_t n_out, const int32_t *nbr, const float *in, const float *w, const float *bias, const float *slope,
                 float *out, hipStream_t s) {
    const int64_t tiles = (n_out + 15) / 16;
    int64_t g = (tiles + 3) / 4;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    if (slope)
        hipLaunchKernelGGL((k_sconv2x2_mfma<CIN, COUT, true>), dim3((int)g), dim3(kBlock), 0, s, n_out, nbr, in, w, bias, slope, out);
    else
        hipLaunchKernelGGL((k_sconv2x2_mfma<CIN, COUT, false>), dim3((int)g), dim3(kBlock), 0, s, n_out, nbr, in, w, bias, slope, out);
}

// true when an MFMA instantiation exists for (cin, cout) and it was launched
bool try_mfma(int cin, int cout, int64_t n_out, const int32_t *nbr, const float *in, const float *w, const float *bias,
              const float *slope, float *out, hipStream_t s) {
    static const bool enabled = [] { const char *e = getenv("DPCG_CNN_MFMA"); return !(e && e[0] == '0'); }();
    if (!enabled) return false;
#define DPCG_CONV_CASE(CI, CO)                                                   \
    if (cin == CI && cout == CO) {                                               \
        launch_mfma<CI, CO>(n_out, nbr, in, w, bias, slope, out, s);             \
        return true;                                                             \
    }
    DPCG_CONV_CASE(16, 16) DPCG_CONV_CASE(16, 32) DPCG_CONV_CASE(16, 64)
    DPCG_CONV_CASE(32, 16) DPCG_CONV_CASE(32, 32) DPCG_CONV_CASE(32, 64)
    DPCG_CONV_CASE(64, 16) DPCG_CONV_CASE(64, 32) DPCG_CONV_CASE(64, 64)
#undef DPCG_CONV_CASE
    return false;
}


}  // namespace
}  // namespace dpcg

using namespace dpcg;

extern "C" int dpcg_convnet_plan_destroy(dpcg_convnet_plan_t p) {
    if (!p) return DPCG_OK;
    for (auto &sl : p->slabs)
        if (sl.first) (void)device_free(sl.first);
    dev_free(p->buf[0]);
    dev_free(p->buf[1]);
    delete p;
    return DPCG_OK;
}

// (Re)builds `p` for a pattern, drawing its arrays from the plan's slabs.  On failure the plan is left EMPTY (n_layers = 0)
// but alive: its memory can serve the next rebuild.
static int build_plan(dpcg_convnet_plan *p, int batch, int64_t height, int64_t width, int64_t nnz, const int32_t *indices,
                      int n_layers, const int32_t *kernel_hw, const int32_t *padding_hw, dpcg_stream_t stream) {
    if (batch <= 0 || height <= 0 || width <= 0 || nnz <= 0 || !indices || n_layers <= 0 || n_layers > kConvMaxLayers ||
        !kernel_hw || !padding_hw)
        return invalid("dpcg_convnet_plan_create: bad arguments");
    if ((int64_t)batch * (height + 2 * n_layers) >= 2147483000LL || nnz >= 2147483000LL)
        return invalid("dpcg_convnet_plan_create: batch * height or nnz exceeds int32");
    for (int l = 0; l < n_layers; ++l) {
        const int kh = kernel_hw[2 * l], kw = kernel_hw[2 * l + 1], ph = padding_hw[2 * l], pw = padding_hw[2 * l + 1];
        if (kh < 1 || kw < 1 || kh * kw > kMaxTaps || ph < 0 || pw < 0)
            return invalid("dpcg_convnet_plan_create: windows up to 2 x 2 (stride 1) are supported");
    }
    hipStream_t s = (hipStream_t)stream;
    p->slab_cursor = 0;
    p->batch = batch;
    p->n_layers = n_layers;
    p->height = height;
    p->width = width;
    p->nnz_in = nnz;
    p->nnz_lower = 0;
    for (auto &L : p->layers) L = LayerPlan();
    int st = DPCG_OK;
    int *d_bad = nullptr;
    int32_t *len = nullptr;
    void *scan_ws = nullptr;
    auto fail = [&](int code) {
        p->n_layers = 0;
        return code;
    };
    // one workspace for every scan of this build (the longest: batch * (height + n_layers) + 1 row counts)
    const int64_t max_rows = (int64_t)batch * (height + 2 * n_layers) + 1;
    const size_t scan_bytes = scan_workspace_bytes(max_rows) + 256;
#define PLAN_TRY(expr)                     \
    do {                                   \
        st = (expr);                       \
        if (st < 0) return fail(st);       \
    } while (0)
#define PLAN_HIP(call)                                                          \
    do {                                                                        \
        hipError_t _e = (call);                                                 \
        if (_e != hipSuccess) return fail(hip_fail(_e, #call, __FILE__, __LINE__)); \
    } while (0)
    const int64_t rows0 = (int64_t)batch * height;
    PLAN_TRY(plan_alloc(p, &p->rowptr0, rows0 + 1));
    PLAN_TRY(plan_alloc(p, &p->col0, nnz));
    PLAN_TRY(plan_alloc(p, &d_bad, 1));
    PLAN_TRY(plan_alloc(p, reinterpret_cast<char **>(&scan_ws), (int64_t)scan_bytes));
    PLAN_TRY(plan_alloc(p, &len, max_rows));
    PLAN_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_sites_to_csr, dim3(grid_rows(nnz)), dim3(kBlock), 0, s, nnz, indices, batch, height, width, p->rowptr0,
                       p->col0, d_bad);
    int bad = 0;
    PLAN_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s));
    PLAN_HIP(hipStreamSynchronize(s));
    if (bad) {
        set_error(bad == 1 ? "dpcg_convnet_plan_create: a site lies outside the batch / image"
                           : "dpcg_convnet_plan_create: sites must be sorted by (batch, row, col) without duplicates");
        return fail(DPCG_ERR_INVALID);
    }
    const int32_t *rp_in = p->rowptr0, *col_in = p->col0;
    int64_t h = height, w = width, sites = nnz;
    for (int l = 0; l < n_layers; ++l) {
        LayerPlan &L = p->layers[l];
        L.kh = kernel_hw[2 * l]; L.kw = kernel_hw[2 * l + 1]; L.ph = padding_hw[2 * l]; L.pw = padding_hw[2 * l + 1];
        L.h_in = h; L.w_in = w;
        L.h_out = h + 2 * L.ph - L.kh + 1;
        L.w_out = w + 2 * L.pw - L.kw + 1;
        if (L.h_out <= 0 || L.w_out <= 0) {
            set_error("dpcg_convnet_plan_create: a layer's output is empty");
            return fail(DPCG_ERR_INVALID);
        }
        L.same_sites = L.kh == 1 && L.kw == 1 && L.ph == 0 && L.pw == 0;
        if (L.same_sites) {
            L.rowptr = const_cast<int32_t *>(rp_in);
            L.col = const_cast<int32_t *>(col_in);
            L.sites = sites;
        } else {
            const int64_t rows_out = (int64_t)batch * L.h_out;
            const ConvGeom g{L.kh, L.kw, L.ph, L.pw, batch, h, w, L.h_out, L.w_out};
            PLAN_TRY(plan_alloc(p, &L.rowptr, rows_out + 1));
            hipLaunchKernelGGL(k_conv_rows<false>, dim3(grid_rows(rows_out + 1)), dim3(kBlock), 0, s, g, rp_in, col_in, len,
                               (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
            PLAN_TRY(exclusive_scan_i32_ws(len, L.rowptr, rows_out + 1, scan_ws, scan_bytes, s));
            int32_t total = 0;
            PLAN_HIP(hipMemcpyAsync(&total, L.rowptr + rows_out, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            PLAN_HIP(hipStreamSynchronize(s));
            if (total <= 0) {
                set_error("dpcg_convnet_plan_create: a layer has no active output site (or more than 2^31)");
                return fail(DPCG_ERR_INVALID);
            }
            L.sites = total;
            PLAN_TRY(plan_alloc(p, &L.col, L.sites));
            PLAN_TRY(plan_alloc(p, &L.nbr, L.sites * L.kh * L.kw));
            hipLaunchKernelGGL(k_conv_rows<true>, dim3(grid_rows(rows_out + 1)), dim3(kBlock), 0, s, g, rp_in, col_in,
                               (int32_t *)nullptr, (const int32_t *)L.rowptr, L.col, L.nbr);
        }
        rp_in = L.rowptr;
        col_in = L.col;
        h = L.h_out;
        w = L.w_out;
        sites = L.sites;
    }
    // rows of the output sites and the lower-triangular CSR the factor is written into
    {
        const LayerPlan &L = p->layers[n_layers - 1];
        const int64_t rows = (int64_t)batch * L.h_out;
        PLAN_TRY(plan_alloc(p, &p->site_row, L.sites));
        PLAN_TRY(plan_alloc(p, &p->site_batch, L.sites));
        PLAN_TRY(plan_alloc(p, &p->lower_rowptr, rows + 1));
        PLAN_TRY(plan_alloc(p, &p->lower_pos, L.sites));
        hipLaunchKernelGGL(k_lower_count, dim3(grid_rows(rows + 1)), dim3(kBlock), 0, s, rows, L.h_out, L.rowptr, L.col, len,
                           p->site_row, p->site_batch);
        PLAN_TRY(exclusive_scan_i32_ws(len, p->lower_rowptr, rows + 1, scan_ws, scan_bytes, s));
        int32_t total = 0;
        PLAN_HIP(hipMemcpyAsync(&total, p->lower_rowptr + rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        PLAN_HIP(hipStreamSynchronize(s));
        p->nnz_lower = total;
        PLAN_TRY(plan_alloc(p, &p->lower_col, p->nnz_lower));
        hipLaunchKernelGGL(k_lower_fill, dim3(grid_rows(rows)), dim3(kBlock), 0, s, rows, L.h_out, L.rowptr, L.col,
                           p->lower_rowptr, p->lower_col, p->lower_pos);
    }
    PLAN_HIP(hipStreamSynchronize(s));
    PLAN_HIP(hipGetLastError());
#undef PLAN_TRY
#undef PLAN_HIP
    return DPCG_OK;
}

extern "C" int dpcg_convnet_plan_create(dpcg_convnet_plan_t *out, int batch, int64_t height, int64_t width, int64_t nnz,
                                        const int32_t *indices, int n_layers, const int32_t *kernel_hw,
                                        const int32_t *padding_hw, dpcg_stream_t stream) {
    if (!out) return invalid("dpcg_convnet_plan_create: NULL out");
    *out = nullptr;
    dpcg_convnet_plan *p = new dpcg_convnet_plan();
    const int st = build_plan(p, batch, height, width, nnz, indices, n_layers, kernel_hw, padding_hw, stream);
    if (st < 0) {
        dpcg_convnet_plan_destroy(p);
        return st;
    }
    *out = p;
    return DPCG_OK;
}

extern "C" int dpcg_convnet_plan_rebuild(dpcg_convnet_plan_t plan, int batch, int64_t height, int64_t width, int64_t nnz,
                                         const int32_t *indices, int n_layers, const int32_t *kernel_hw,
                                         const int32_t *padding_hw, dpcg_stream_t stream) {
    if (!plan) return invalid("dpcg_convnet_plan_rebuild: NULL plan");
    return build_plan(plan, batch, height, width, nnz, indices, n_layers, kernel_hw, padding_hw, stream);
}


extern "C" int dpcg_convnet_plan_info(dpcg_convnet_plan_t p, int layer, int64_t *sites, int64_t *height, int64_t *width,
                                      int64_t *nnz_lower) {
    if (!p || layer < 0 || layer >= p->n_layers) return invalid("dpcg_convnet_plan_info: bad plan or layer (or a plan whose rebuild failed)");
    const LayerPlan &L = p->layers[layer];
    if (sites) *sites = L.sites;
    if (height) *height = L.h_out;
    if (width) *width = L.w_out;
    if (nnz_lower) *nnz_lower = p->nnz_lower;
    return DPCG_OK;
}

extern "C" int dpcg_convnet_plan_output(dpcg_convnet_plan_t p, int32_t *indices_out, int32_t *lower_rowptr,
                                        int32_t *lower_col, dpcg_stream_t stream) {
    if (!p || p->n_layers <= 0) return invalid("dpcg_convnet_plan_output: NULL or empty plan");
    hipStream_t s = (hipStream_t)stream;
    const LayerPlan &L = p->layers[p->n_layers - 1];
    if (indices_out)
        hipLaunchKernelGGL(k_out_indices, dim3(grid_rows(L.sites)), dim3(kBlock), 0, s, L.sites, p->site_batch, p->site_row, L.col,
                           indices_out);
    if (lower_rowptr)
        DPCG_HIP(hipMemcpyAsync(lower_rowptr, p->lower_rowptr, (size_t)((int64_t)p->batch * L.h_out + 1) * sizeof(int32_t),
                                hipMemcpyDeviceToDevice, s));
    if (lower_col)
        DPCG_HIP(hipMemcpyAsync(lower_col, p->lower_col, (size_t)p->nnz_lower * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}

extern "C" int dpcg_convnet_forward(dpcg_convnet_plan_t p, const int32_t *channels, const float *const *weights,
                                    const float *const *biases, const float *const *prelu, const float *features_in,
                                    float *features_out, double *lower_val, int lower_softplus, dpcg_stream_t stream) {
    if (!p || !channels || !weights || !biases || !prelu || !features_in) return invalid("dpcg_convnet_forward: NULL argument");
    if (p->n_layers <= 0) return invalid("dpcg_convnet_forward: empty plan (its last rebuild failed)");
    const int n = p->n_layers;
    for (int l = 0; l <= n; ++l)
        if (channels[l] < 1 || channels[l] > 1024) return invalid("dpcg_convnet_forward: bad channel count");
    for (int l = 0; l < n; ++l)
        if (!weights[l]) return invalid("dpcg_convnet_forward: NULL weight");
    const LayerPlan &last = p->layers[n - 1];
    const bool fused_final = last.same_sites && channels[n] == 1 && !prelu[n - 1] &&
                             (channels[n - 1] == 16 || channels[n - 1] == 32 || channels[n - 1] == 64);
    if ((lower_val || lower_softplus) && !fused_final)
        return invalid("dpcg_convnet_forward: the lower-triangular output needs a pointwise last layer with one output channel "
                       "and 16 / 32 / 64 input channels");
    if (!fused_final && !features_out) return invalid("dpcg_convnet_forward: NULL features_out");
    hipStream_t s = (hipStream_t)stream;
    // ping-pong buffers for the hidden features
    int64_t need = 0;
    for (int l = 0; l < n - 1; ++l) need = std::max<int64_t>(need, p->layers[l].sites * channels[l + 1]);
    if (!fused_final) need = std::max<int64_t>(need, 0);
    for (int b = 0; b < 2; ++b)
        if (p->buf_cap[b] < need) {
            dev_free(p->buf[b]);
            p->buf_cap[b] = 0;
            DPCG_TRY(dev_alloc(&p->buf[b], need));
            p->buf_cap[b] = need;
        }
    const float *in = features_in;
    for (int l = 0; l < n; ++l) {
        const LayerPlan &L = p->layers[l];
        const int cin = channels[l], cout = channels[l + 1], taps = L.kh * L.kw;
        const bool is_last = l == n - 1;
        if (is_last && fused_final) {
            const int g = grid_rows(L.sites, 2048);
#define DPCG_FINAL(CI)                                                                                                  \
    hipLaunchKernelGGL((k_sconv_final<CI>), dim3(g), dim3(kBlock), 0, s, L.sites, in, weights[l], biases[l], p->site_row, \
                       L.col, p->lower_pos, lower_softplus, features_out, lower_val)
            if (cin == 16) DPCG_FINAL(16);
            else if (cin == 32) DPCG_FINAL(32);
            else DPCG_FINAL(64);
#undef DPCG_FINAL
            break;
        }
        float *dst = is_last ? features_out : p->buf[l & 1];
        const bool mfma = taps == 4 && L.kh == 2 && L.nbr &&
                          try_mfma(cin, cout, L.sites, L.nbr, in, weights[l], biases[l], prelu[l], dst, s);
        if (!mfma) {
            const int g = grid_rows(L.sites * cout, 4096);
            if (prelu[l])
                hipLaunchKernelGGL((k_sconv_generic<true>), dim3(g), dim3(kBlock), 0, s, L.sites, taps, (const int32_t *)L.nbr, cin,
                                   cout, in, weights[l], biases[l], prelu[l], dst);
            else
                hipLaunchKernelGGL((k_sconv_generic<false>), dim3(g), dim3(kBlock), 0, s, L.sites, taps, (const int32_t *)L.nbr, cin,
                                   cout, in, weights[l], biases[l], prelu[l], dst);
        }
        in = dst;
    }
    DPCG_CHECK_LAUNCH();
    return DPCG_OK;
}
