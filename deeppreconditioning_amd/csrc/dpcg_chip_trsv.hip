// Whole-solve kernel for M = (L L^T)^-1 APPLIED BY TWO TRIANGULAR SOLVES (z = L^-T (L^-1 r): the reference's incomplete-Cholesky techniques
// when they are solved with rather than multiplied, test.py:81-88; BASELINE config 3: "level-scheduled L / L^T trisolve") -- hand-written for
// gfx950 (MI355X), the whole chip as one team like dpcg_chip.hip, cg.py:58-90 in ONE launch.
//
// The multi-launch update with such a preconditioner is the three launches of the Jacobi update plus one launch per level (colour sweeps)
// or a sync-free launch per triangle: 61 us per update at 1M rows with IC(0) in multicolour order, 136 in the caller's order, against 12.5
// for Jacobi in the whole-chip kernel -- every triangular-solve preconditioner lost to Jacobi in time to solution.  Here:
//   * geometry, residency of A (values in LDS / registers, 16-bit column offsets), vectors in registers, the published granules {z, p},
//     the two chip-wide exchanges per update, placement and visibility: exactly dpcg_chip.hip (M = I / Jacobi).
//   * the factor does not fit beside A, so L and L^T are STREAMED every update, each wave its own list of BLOCKS prepared once per
//     preconditioner (k_trsv_plan below).  A block = at most one row per lane, all of ONE dependency level: every lane that still has a row
//     of that level among its slots gives its lowest such slot, so a level takes as many blocks as its busiest lane has rows in it.  A
//     header {width W, level, lane mask, first entry} and W + 1 entry columns compacted over the active lanes (entry j of the active lane
//     of rank r at base + j * count + r: perfectly coalesced, no padding for the lanes that sit out), the diagonal last, the lane's slot in
//     its record.  Entries are {value (8 B), column in the handle's numbering (4 B; -1: the row is shorter)}; a row's entries keep the factor's CSR order, so
//     acc = rhs; acc -= l_ij * y_j (in order); y_i = acc / l_ii  is sequential substitution to the bit (contraction off).
//   * a wave walks its blocks level by level (all of level l, then l + 1 ...); what a row depends on lies in lower levels, i.e. in EARLIER
//     blocks of whoever owns it -- so bounded polling cannot deadlock (the wave at the lowest level always finds its operands published).
//   * y and z travel as SELF-VALIDATING 16-byte granules {value, value ^ key(launch, generation)} (dpcg_chip_llt.hip): a gather is accepted
//     only when its halves differ by exactly the expected key, otherwise read again -- the dependency hand-off needs no flag, no ticket,
//     no barrier.  Three blocks are in flight per wave: columns fetched two blocks ahead, values and operand gathers one block ahead.
//   * the dot products run in the whole-chip tree (thread's rows in slot order, wave tree, 8 wave sums, two hops): the CPU restatement's
//     form "chip" with kind llt_solve covers it unchanged -- history, count and x bit for bit.
// Any lower-triangular factor whose rows have at most WL off-diagonal entries and whose two dependency graphs have at most 64 levels is
// taken (IC(0) in multicolour order: 2-9 levels; in a scattered caller's order: ~20); natural orders of grids (hundreds of levels) keep the
// launches.
#include <algorithm>

#include "dpcg_chip_device.h"
#include "dpcg_host.h"

namespace dpcg {

using namespace chip;

namespace {

constexpr int kTrsvMaxLevels = 64;       // (the level fields of the plans hold 6 bits; which factors a plain solve takes: chip_trsv_level_limit())

// ---- the plan: block lists of one triangular factor in the chip kernel's geometry ----------------------------------------------------------
// One launch of 256 x 512 in the solve kernel's geometry (v = blockIdx here: no placement involved).  WRITE = false: counts[wave] =
// {blocks, entries}; WRITE = true: headers and entries from the scanned offsets.  frp / fci / fval: the factor in ITS numbering (L: diagonal
// last; L^T: diagonal first); lvl[f]: dependency level of factor row f; f_of_handle / handle_of_f: the maps between the handle's numbering
// and the factor's (null: the same numbering).
template <bool WRITE>
__global__ __launch_bounds__(kChipThreads) void k_trsv_plan(int n, int per, int nlev, const int32_t *__restrict__ frp, const int32_t *__restrict__ fci,
                                                            const double *__restrict__ fval, const int32_t *__restrict__ lvl,
                                                            const int32_t *__restrict__ f_of_handle, const int32_t *__restrict__ handle_of_f, int upper,
                                                            int2 *counts, const int32_t *__restrict__ first_blk, const int32_t *__restrict__ first_ent,
                                                            int4 *blk, double *ent_val, int32_t *ent_col, int *band_wl /* [0] band, [1] longest row (off-diagonal) */,
                                                            int32_t *lv0 /* WRITE: [v][t] |= slots whose row sits in level 0, << (8 * upper) */,
                                                            double *diag0 /* WRITE, lower only: [v][k][t] the factor's diagonal */) {
    const int t = threadIdx.x, v = blockIdx.x, lane = t & 63, wave = v * (kChipThreads / 64) + (t >> 6);
    int lv[kChipMaxRpt], ln[kChipMaxRpt], rs[kChipMaxRpt];
    int band = 0, wl = 0;
#pragma unroll
    for (int k = 0; k < kChipMaxRpt; ++k) {
        const int loc = k * kChipThreads + t, i = v * per + loc;
        const bool valid = loc < per && i < n;
        lv[k] = -1; ln[k] = 0; rs[k] = 0;
        if (valid) {
            const int f = f_of_handle ? f_of_handle[i] : i;
            const int s = frp[f], e = frp[f + 1];
            lv[k] = lvl[f];
            ln[k] = e - s - 1;
            rs[k] = s;
            wl = ln[k] > wl ? ln[k] : wl;
            if (!WRITE) {
                for (int q = s; q < e; ++q) {
                    const int c = handle_of_f ? handle_of_f[fci[q]] : fci[q];
                    const int dl = c > i ? c - i : i - c;
                    band = dl > band ? dl : band;
                }
            }
        }
    }
    int nb = 0, ne = 0;
    if (WRITE) {
        nb = first_blk[wave];
        ne = first_ent[wave];
        // rows without dependencies (level 0) are no blocks: the solve sweeps them slot by slot -- a mask per thread, the diagonals by slot
        unsigned m0 = 0;
#pragma unroll
        for (int k = 0; k < kChipMaxRpt; ++k) {
            m0 |= (lv[k] == 0 ? 1u : 0u) << k;
            if (!upper && lv[k] >= 0) diag0[((size_t)v * kChipMaxRpt + k) * kChipThreads + t] = fval[rs[k] + ln[k]];
        }
        if (upper) lv0[v * kChipThreads + t] |= (int)(m0 << 8);
        else lv0[v * kChipThreads + t] = (int)m0;
    }
    for (int l = 1; l < nlev; ++l) {
        unsigned mk = 0;                                      // the slots of this lane whose row sits in level l
#pragma unroll
        for (int k = 0; k < kChipMaxRpt; ++k) mk |= (lv[k] == l ? 1u : 0u) << k;
        // a block takes ONE row of every lane that still has one in this level (its lowest slot): as many blocks as the busiest lane has rows
        while (__ballot(mk != 0) != 0) {
            const bool active = mk != 0;
            const unsigned long long m = __ballot(active);
            const int kl = active ? __builtin_ctz(mk) : 0;
            int len = 0, rs0 = 0;
#pragma unroll
            for (int k = 0; k < kChipMaxRpt; ++k)
                if (k == kl) { len = ln[k]; rs0 = rs[k]; }
            int W = active ? len : 0;
            for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(W, off); W = o > W ? o : W; }
            const int cnt = __popcll(m);
            if (WRITE) {
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (lane == 0) blk[nb] = make_int4(W | (l << 8), (int)(unsigned)m, (int)(unsigned)(m >> 32), ne);
                if (active) {
                    for (int j = 0; j < W; ++j) {
                        const int e = ne + j * cnt + rank;
                        if (j < len) {
                            const int q = rs0 + (upper ? 1 + j : j);
                            ent_col[e] = (handle_of_f ? handle_of_f[fci[q]] : fci[q]) + 1;      // (+ 1: 0 = no entry, which is also what a load out of range returns)
                            ent_val[e] = fval[q];
                        } else {
                            ent_col[e] = 0;
                            ent_val[e] = 0.0;
                        }
                    }
                    const int ed = ne + W * cnt + rank;
                    ent_col[ed] = kl;                             // (the diagonal's record names the lane's slot)
                    ent_val[ed] = fval[upper ? rs0 : rs0 + len];
                }
            }
            ++nb;
            ne += (W + 1) * cnt;
            mk &= mk - 1u;
        }
    }
    if (!WRITE) {
        if (lane == 0) counts[wave] = make_int2(nb, ne);
        for (int off = 32; off > 0; off >>= 1) {
            const int ob = __shfl_xor(band, off), ow = __shfl_xor(wl, off);
            band = ob > band ? ob : band;
            wl = ow > wl ? ow : wl;
        }
        if (lane == 0) {
            atomicMax(&band_wl[0], band);
            atomicMax(&band_wl[1], wl);
        }
    }
}

// exclusive scan of the 2048 per-wave counts (one workgroup): first_blk / first_ent [2049]
__global__ __launch_bounds__(1024) void k_trsv_scan(const int2 *__restrict__ counts, int32_t *first_blk, int32_t *first_ent) {
    constexpr int NW = kChipWGs * (kChipThreads / 64);
    __shared__ int sb[1024], se[1024];
    const int t = threadIdx.x;
    const int2 a = counts[2 * t], b = counts[2 * t + 1];
    sb[t] = a.x + b.x;
    se[t] = a.y + b.y;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int vb = t >= off ? sb[t - off] : 0, ve = t >= off ? se[t - off] : 0;
        __syncthreads();
        sb[t] += vb;
        se[t] += ve;
        __syncthreads();
    }
    const int eb = sb[t] - a.x - b.x, ee = se[t] - a.y - b.y;      // exclusive
    first_blk[2 * t] = eb;
    first_ent[2 * t] = ee;
    first_blk[2 * t + 1] = eb + a.x;
    first_ent[2 * t + 1] = ee + a.y;
    if (t == 1023) {
        first_blk[NW] = sb[t];
        first_ent[NW] = se[t];
    }
}

// ---- the solve ------------------------------------------------------------------------------------------------------------------------------
// RPT: rows per thread (2, 4, 8); WMAX: entry slots per row of A (5, 7, 9); WL: off-diagonal entries per factor row (WMAX - 1 for IC(0)).
// TRACE (development, DPCG_CHIP_TRACE=1): ticks of the 100 MHz clock by phase of the preconditioner apply, per wave (lane 0), summed over the solve
template <int RPT, int WMAX, int WL, bool TRACE = false>
__global__ __launch_bounds__(kChipThreads) void k_pcg_chip_trsv(const ChipTrsvDesc d) {
    constexpr int NS = RPT * WMAX;
    constexpr int NLDS = NS < kChipLdsSlots ? NS : kChipLdsSlots;
    constexpr int NREG = NS - NLDS;
    extern __shared__ __attribute__((aligned(16))) double chip_lv[];   // [NLDS][512]
    __shared__ double sh[2 * 16];
    __shared__ double s_res[2][2];
    __shared__ int s_flag;
    __shared__ unsigned long long s_tr[TRACE ? 8 * 8 : 1];        // [wave][phase]: 0 / 3 prologue + level-0 sweep (L / L^T), 1 / 4 the blocks, 2 / 5 of it polling
                                                                  // again, 6 the workgroup's wait behind the apply, 7 blocks that had to poll again
    const int t = threadIdx.x, lane = t & 63;
    const int v = ((int)blockIdx.x & 7) * (kChipWGs / 8) + ((int)blockIdx.x >> 3);
    const int row0 = v * d.per + t;
    const int grp = (int)blockIdx.x & 7, rank = (int)blockIdx.x >> 3;
    const int glo = grp * (kChipWGs / 8) * d.per;
    const int ghi = (glo + (kChipWGs / 8) * d.per < d.n) ? glo + (kChipWGs / 8) * d.per : d.n;
    const int remote_base = (d.n + kChipZpPad) * 16;
    const unsigned gran_bytes = 2u * (unsigned)(d.n + kChipZpPad) * 16u;
    const __amdgpu_buffer_rsrc_t zp_rs = chip_rsrc(d.zp, gran_bytes);
    const __amdgpu_buffer_rsrc_t y_rs = chip_rsrc(d.ypub, gran_bytes);
    const __amdgpu_buffer_rsrc_t w_rs = chip_rsrc(d.zpub, gran_bytes);

    // ---- A's slice and the vectors of the own rows: read once (dpcg_chip.hip) ----------------------------------------------------------
    double vr[NREG > 0 ? NREG : 1];
    unsigned dl[(NS + 1) / 2];
    constexpr int LB = WMAX > 7 ? 8 : 4;
    constexpr unsigned LV = 1u << (LB - 1), LM = LV - 1u;
    static_assert(LB * RPT <= 32, "row lengths of a thread in one register");
    unsigned lens = 0;
    double x[RPT], r[RPT], p[RPT], q[RPT], z[RPT];
    double bb_loc = 0.0;
    int rs_k[RPT], len_k[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int loc = k * kChipThreads + t, i = row0 + k * kChipThreads;
        const bool valid = loc < d.per && i < d.n;
        const int ic = valid ? i : 0;
        const int rs = d.rp[ic], re = d.rp[ic + 1];
        const double bi = d.b[ic];
        const double xi = d.x0 ? d.x0[ic] : 0.0;
        rs_k[k] = rs;
        len_k[k] = valid ? re - rs : 0;
        lens |= (valid ? (LV | (unsigned)(re - rs)) : 0u) << (LB * k);
        x[k] = valid ? xi : 0.0;
        r[k] = valid ? bi : 0.0;
        p[k] = q[k] = z[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if ((lens >> (LB * k)) & LV) bb_loc += r[k] * r[k];
        const int i = row0 + k * kChipThreads;
        int cj[WMAX];
        double aj[WMAX];
#pragma unroll
        for (int j = 0; j < WMAX; ++j) {
            const int e = j < len_k[k] ? rs_k[k] + j : 0;
            cj[j] = d.ci[e];
            aj[j] = d.val[e];
        }
#pragma unroll
        for (int j = 0; j < WMAX; ++j) {
            const int s = k * WMAX + j;
            const bool on = j < len_k[k];
            const int c = on ? cj[j] : i;
            const double a = on ? aj[j] : 0.0;
            const unsigned del = (unsigned)(c - i + 32768) & 0xffffu;
            if (s & 1) dl[s >> 1] |= del << 16;
            else dl[s >> 1] = del;
            if (s < NREG) vr[s < NREG ? s : 0] = a;
            else chip_lv[(s - NREG) * kChipThreads + t] = a;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    bool local = false;
    auto row_on = [&](int k) -> bool { return ((lens >> (LB * k)) & LV) != 0; };
    const unsigned lv0 = (unsigned)d.lv0[v * kChipThreads + t];      // bits 0-7: slots whose row has no dependency in L; 8-15: in L^T

    // q = A p_k for the own rows, the gathered entries of p_k recomputed from the granules {z_k, p_{k-1}} (dpcg_chip.hip)
    auto spmv = [&](double beta) {
        u32x4 g[2][WMAX];
        int tl = t;
        asm volatile("" : "+v"(tl));
        const double *lvt = chip_lv + tl;
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        const int local_shift = grp * 128;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));
#pragma unroll
        for (int e = 0; e < (NS + 1) / 2; ++e) asm volatile("" : "+v"(dl[e]));
        asm volatile("" : "+v"(lens));
        auto request = [&](int k, u32x4 (&gk)[WMAX]) {
            const int rowk = row0 + k * kChipThreads;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int s = k * WMAX + j;
                const int del = (int)((dl[s >> 1] >> (16 * (s & 1))) & 0xffffu);
                const int c = rowk + del - 32768;
                const bool own = (unsigned)(c - glo_l) < (unsigned)span_l;
                gk[j] = __builtin_amdgcn_raw_buffer_load_b128(zp_rs, c * 16 + (own ? local_shift : remote_base), 0, kSc1);
            }
        };
        request(0, g[0]);
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            if (k + 1 < RPT) request(k + 1, g[(k + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int len = (int)((lens >> (LB * k)) & LM);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int s = k * WMAX + j;
                const double a = s < NREG ? vr[s < NREG ? s : 0] : lvt[(s - NREG) * kChipThreads];
                const double pc = lo_f64(g[k & 1][j]) + beta * hi_f64(g[k & 1][j]);  // = p_k[c], cg.py:83
                if (j < len) acc += a * pc;
            }
            q[k] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (TRACE && t < 64) s_tr[t] = 0;
    auto tick = [&]() -> unsigned long long { return TRACE ? wall_clock64() : 0ull; };
    auto tr_add = [&](int phase, unsigned long long dt) {
        if (TRACE && lane == 0) s_tr[(t >> 6) * 8 + phase] += dt;
    };
    Exchange X;
    X.part_rs = chip_rsrc(d.part, (unsigned)kChipSlotBytes);
    X.v = v; X.grp = grp; X.rank = rank; X.sh = sh; X.s_res = s_res; X.s_flag = &s_flag; X.err = d.err;
    auto chip_sum2 = [&](double a, double b2, bool publish, double &ra, double &rb) -> bool { return exchange2(X, a, b2, publish, ra, rb); };
    unsigned far_rows = 0xffu;
    int row0_l = row0;
    auto publish_zp = [&](int k, double zk, double pk) {
        const int o = (row0_l + k * kChipThreads) * 16;
        if (local) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + grp * 128, 0, 0);
        if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + remote_base, 0, kSc1);
    };

    // ---- one triangular solve: the wave walks its blocks of factor `fi` (0: L y = rhs, 1: L^T z = rhs) --------------------------------
    const int wave_id = __builtin_amdgcn_readfirstlane(v * (kChipThreads / 64) + (t >> 6));
    unsigned pub_gen = 0;
    auto make_key = [&](unsigned gen) -> unsigned long long {
        return (((unsigned long long)d.nonce << 32) | (unsigned long long)gen) * 0x9E3779B97F4A7C15ull | 1ull;
    };
    auto tri_solve = [&](int fi, const __amdgpu_buffer_rsrc_t &out_rs, const double (&rhs)[RPT], double (&out)[RPT], unsigned long long key) __attribute__((always_inline)) -> bool {
        const int4 *__restrict__ blk = fi ? d.blk_u : d.blk_l;
        const __amdgpu_buffer_rsrc_t ev_rs = chip_rsrc(fi ? d.val_u : d.val_l, (unsigned)(fi ? d.nent_u : d.nent_l) * 8u);
        const __amdgpu_buffer_rsrc_t ec_rs = chip_rsrc(fi ? d.col_u : d.col_l, (unsigned)(fi ? d.nent_u : d.nent_l) * 4u);
        const int32_t *wf = fi ? d.first_u : d.first_l;
        const int b0 = __builtin_amdgcn_readfirstlane(wf[wave_id]), b1 = __builtin_amdgcn_readfirstlane(wf[wave_id + 1]);
        const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));
        // Three blocks are in flight per wave: the COLUMNS of block b + 2 are fetched while the VALUES of block b + 1 are fetched and its operands
        // gathered (their addresses come from columns that arrived an iteration ago), while block b is validated, solved and published.  The
        // two register sets of each stage are named and used in turn (the loop advances two blocks per trip): a copy of registers whose
        // loads are in flight would wait for them.  Headers run three blocks ahead, as scalar loads.
        struct Hdr { int hw, base, cnt; unsigned mlo, mhi; };
        struct CSt { int c[WL]; };
        struct DSt { int off[WL]; int k; double a[WL]; double dg; u32x4 g[WL]; };
        constexpr int kNoCol = 0x7ffffff0;                        // (a gather offset out of range: no request, zeros returned)
        auto header = [&](int b) __attribute__((always_inline)) -> Hdr {                         // (scalar loads: b is the same for the whole wave)
            const int bu = __builtin_amdgcn_readfirstlane(b);    // (the loop's exit also hangs on a per-lane flag: without this the index is a vector value)
            // (read through the constant address space: the lists are written once per preconditioner, and only a load the compiler knows to be
            // invariant becomes a SCALAR load -- a vector load here waits on vmcnt(0), i.e. on everything fetched ahead)
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            typedef const i32x4 __attribute__((address_space(4))) *const_i32x4_ptr;
            const const_i32x4_ptr cb = (const_i32x4_ptr)(unsigned long long)blk;
            i32x4 h = {0, 0, 0, 0};
            if (bu < b1) h = cb[bu];
            Hdr H;
            H.hw = __builtin_amdgcn_readfirstlane(h.x) & 0xff;
            H.mlo = (unsigned)__builtin_amdgcn_readfirstlane(h.y);
            H.mhi = (unsigned)__builtin_amdgcn_readfirstlane(h.z);
            H.base = __builtin_amdgcn_readfirstlane(h.w);
            H.cnt = __popc(H.mlo) + __popc(H.mhi);
            return H;
        };
        auto lane_on = [&](const Hdr &H) __attribute__((always_inline)) -> bool { return ((lane < 32 ? H.mlo >> lane : H.mhi >> (lane - 32)) & 1u) != 0; };
        auto lane_rank = [&](const Hdr &H) __attribute__((always_inline)) -> int { return __builtin_amdgcn_mbcnt_hi(H.mhi, __builtin_amdgcn_mbcnt_lo(H.mlo, 0u)); };
        // (no branch and no select around these loads: a lane that sits out, and an entry column beyond the block's width, load from an
        // offset out of range -- no request, zeros returned -- and a stored column of 0 means "no entry": every load of a stage is issued
        // back to back and nothing waits before the stage's consumer does)
        auto issue_cols = [&](const Hdr &H, CSt &C) __attribute__((always_inline)) {
            const bool act = lane_on(H);
            const int e0 = H.base + lane_rank(H);
#pragma unroll
            for (int j = 0; j < WL; ++j) C.c[j] = __builtin_amdgcn_raw_buffer_load_b32(ec_rs, (act && j < H.hw) ? (e0 + j * H.cnt) * 4 : kNoCol, 0, 0);
        };
        auto issue_data = [&](const Hdr &H, const CSt &C, DSt &D) __attribute__((always_inline)) {
            const bool act = lane_on(H);
            const int e0 = H.base + lane_rank(H);
            const int ed = act ? e0 + H.hw * H.cnt : 0x0fffffff;
            D.k = __builtin_amdgcn_raw_buffer_load_b32(ec_rs, ed * 4, 0, 0);
            const u32x2 dv2 = __builtin_amdgcn_raw_buffer_load_b64(ev_rs, ed * 8, 0, 0);
            D.dg = __hiloint2double((int)dv2.y, (int)dv2.x);
#pragma unroll
            for (int j = 0; j < WL; ++j) {
                const u32x2 av = __builtin_amdgcn_raw_buffer_load_b64(ev_rs, (act && j < H.hw) ? (e0 + j * H.cnt) * 8 : kNoCol, 0, 0);
                D.a[j] = __hiloint2double((int)av.y, (int)av.x);
            }
#pragma unroll
            for (int j = 0; j < WL; ++j) {
                const int c = C.c[j] - 1;                         // (-1: no entry)
                const bool own = (unsigned)(c - glo_l) < (unsigned)span_l;
                D.off[j] = c >= 0 ? c * 16 + (own ? grp * 128 : remote_base) : kNoCol;
                D.g[j] = __builtin_amdgcn_raw_buffer_load_b128(out_rs, D.off[j], 0, kSc1);
            }
        };
        int ok = 1;                                               // (kept the same for the whole wave: the loop below must stay a uniform one)
        auto finish = [&](const Hdr &H, DSt &D) __attribute__((always_inline)) {
            const bool act = lane_on(H);
            const int k = D.k;
            auto stale = [&]() -> bool {
                bool bad = false;
#pragma unroll
                for (int j = 0; j < WL; ++j) bad = bad || (D.off[j] != kNoCol && ((D.g[j].x ^ D.g[j].z) != klo || (D.g[j].y ^ D.g[j].w) != khi));
                return bad;
            };
            // (the first look stands outside the polling loop: there it waits for THIS block's gathers only -- the loads of the two blocks behind
            // it stay in flight -- whereas a loop that gathers again must wait for everything at its head)
            if (__ballot(stale()) != 0) {
                const unsigned long long tp0 = tick();
                tr_add(7, 1);
                unsigned spins = 0;
                unsigned long long t0 = 0;
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (stale()) {
#pragma unroll
                        for (int j = 0; j < WL; ++j) D.g[j] = __builtin_amdgcn_raw_buffer_load_b128(out_rs, D.off[j], 0, kSc1);
                    }
                    if (__ballot(stale()) == 0) break;
                    if ((++spins & 255u) == 0) {
                        const unsigned long long now = wall_clock64();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > kChipSpinTicks || __hip_atomic_load(d.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            atomicExch(d.err, 1);
                            ok = 0;
                            break;
                        }
                    }
                }
                tr_add(3 * fi + 2, tick() - tp0);
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            // (a lane's slot k is not a constant: the row's operand is SELECTED out of the registers -- each candidate made opaque first, or
            // the optimiser folds the chain into an indexed load and moves the vectors to scratch memory)
            double acc = rhs[0];
#pragma unroll
            for (int kk = 1; kk < RPT; ++kk) {
                double cand = rhs[kk];
                asm volatile("" : "+v"(cand));
                acc = k == kk ? cand : acc;
            }
#pragma unroll
            for (int j = 0; j < WL; ++j)
                if (D.off[j] != kNoCol) acc -= D.a[j] * lo_f64(D.g[j]);
            const double res = acc / (act ? D.dg : 1.0);
            if (act) {
                const int o = (row0 + k * kChipThreads) * 16;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(res), tag = bits ^ key;
                u32x4 w;
                w.x = (unsigned)bits; w.y = (unsigned)(bits >> 32); w.z = (unsigned)tag; w.w = (unsigned)(tag >> 32);
                if (local) __builtin_amdgcn_raw_buffer_store_b128(w, out_rs, o + grp * 128, 0, 0);
                if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(w, out_rs, o + remote_base, 0, kSc1);
            }
#pragma unroll
            for (int kk = 0; kk < RPT; ++kk) {
                double keep = out[kk];
                asm volatile("" : "+v"(keep));
                out[kk] = (act && k == kk) ? res : keep;
            }
        };
        const unsigned long long tk0 = tick();
        Hdr H0 = header(b0), H1 = header(b0 + 1), H2 = header(b0 + 2);
        CSt Ca, Cb;
        DSt Da, Db;
        issue_cols(H0, Ca);
        issue_cols(H1, Cb);
        // the rows without dependencies (level 0), slot by slot: out = rhs / l_ii, published at once -- no entries, no operands, no block
        {
            const unsigned m0 = (lv0 >> (8 * fi)) & 0xffu;
            const __amdgpu_buffer_rsrc_t dg_rs = chip_rsrc(d.diag0, (unsigned)(kChipWGs * kChipMaxRpt * kChipThreads) * 8u);
            double dg[RPT];
#pragma unroll
            for (int kk = 0; kk < RPT; ++kk) {
                const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(dg_rs, ((m0 >> kk) & 1u) ? ((v * kChipMaxRpt + kk) * kChipThreads + t) * 8 : kNoCol, 0, 0);
                dg[kk] = __hiloint2double((int)w2.y, (int)w2.x);
            }
#pragma unroll
            for (int kk = 0; kk < RPT; ++kk) {
                const bool on0 = ((m0 >> kk) & 1u) != 0;
                const double res = rhs[kk] / (on0 ? dg[kk] : 1.0);
                if (on0) {
                    const int o = (row0 + kk * kChipThreads) * 16;
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(res), tag = bits ^ key;
                    u32x4 w;
                    w.x = (unsigned)bits; w.y = (unsigned)(bits >> 32); w.z = (unsigned)tag; w.w = (unsigned)(tag >> 32);
                    if (local) __builtin_amdgcn_raw_buffer_store_b128(w, out_rs, o + grp * 128, 0, 0);
                    if ((far_rows >> kk) & 1u) __builtin_amdgcn_raw_buffer_store_b128(w, out_rs, o + remote_base, 0, kSc1);
                }
                out[kk] = on0 ? res : out[kk];
            }
        }
        issue_data(H0, Ca, Da);
        const unsigned long long tk1 = tick();
        tr_add(3 * fi, tk1 - tk0);
        for (int b = b0; b < b1 && ok != 0; b += 2) {
            const Hdr H3 = header(b + 3);
            issue_cols(H2, Ca);                                   // block b + 2 (Ca's columns went into the gathers of block b)
            issue_data(H1, Cb, Db);                               // block b + 1
            finish(H0, Da);                                       // block b
            const Hdr H4 = header(b + 4);
            issue_cols(H3, Cb);                                   // block b + 3
            issue_data(H2, Ca, Da);                               // block b + 2
            finish(H1, Db);                                       // block b + 1 (beyond the last block: an empty header -- no lane, no request)
            H0 = H2; H1 = H3; H2 = H4;
        }
        tr_add(3 * fi + 1, tick() - tk1);
        return ok != 0;
    };
    // z = L^-T (L^-1 r) for the own rows (cg.py:61,81); y lives in q's registers (q is dead between cg.py:80 and the next cg.py:75)
    auto apply_m = [&]() -> bool {
        const unsigned long long key_y = make_key(++pub_gen);
        bool ok = tri_solve(0, y_rs, r, q, key_y);
        const unsigned long long key_z = make_key(++pub_gen);
        if (ok) ok = tri_solve(1, w_rs, q, z, key_z);
        const unsigned long long ts0 = tick();
        const bool all_ok = __syncthreads_and(ok ? 1 : 0) != 0;
        tr_add(6, tick() - ts0);
        return all_ok;
    };

    bool alive = true;
    double dummy = 0.0, dummy2 = 0.0;
    if (d.xcc) {
        local = groups_on_one_xcd(X, d.xcc, alive);
        X.local = local;
        if (local) {
            far_rows = 0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = row0 + k * kChipThreads;
                if (i < glo + d.band || i >= ghi - d.band) far_rows |= 1u << k;
            }
        }
    }
    // ---- cg.py:58-67 -------------------------------------------------------------------------------------------------
    if (alive && d.x0) {                                          // r = b - A x0 (cg.py:60): x0 published as "z", beta = 0
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) publish_zp(k, x[k], 0.0);
        alive = chip_sum2(0.0, 0.0, true, dummy, dummy2);
        if (alive) {
            spmv(0.0);
#pragma unroll
            for (int k = 0; k < RPT; ++k) r[k] = r[k] - q[k];
            alive = chip_sum2(0.0, 0.0, false, dummy, dummy2);    // everybody has read x0 out of the granules before z_0 overwrites them
        }
    }
    if (alive) alive = apply_m();                                 // cg.py:61
    double rz_loc = 0.0, t0_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        p[k] = z[k];                                              // cg.py:62
        if (row_on(k)) {
            rz_loc += r[k] * z[k];
            t0_loc += d.init_check_r ? r[k] * r[k] : z[k] * z[k];  // cg.py:66: the first test is on z
            publish_zp(k, z[k], 0.0);                             // p_0 = z_0 + 0 * p_{-1}
        }
    }
    double bb = 0.0, rz = 0.0, tt = 0.0;
    if (alive) alive = chip_sum2(bb_loc, rz_loc, true, bb, rz);
    if (alive) alive = chip_sum2(t0_loc, 0.0, false, tt, dummy);
    double res = tt / bb, beta = 0.0;
    int k_done = 0, status = DPCG_MAX_ITER;
    bool stop = false;
    if (alive) {
        if (v == 0 && t == 0 && d.hist_cap > 0) d.hist[0] = res;
        const bool conv = (res < d.rtol_sq) || (tt < d.atol_sq);
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
    // ---- cg.py:70-87 -----------------------------------------------------------------------------------------------------
    while (alive && !stop && k_done < d.max_iter) {
        spmv(beta);                                               // cg.py:75
        double pq_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) pq_loc += q[k] * p[k];
        double pq = 0.0;
        if (!(alive = chip_sum2(pq_loc, 0.0, false, pq, dummy))) break;         // every SpMV of this update is done
        const double alpha = rz / pq;                             // cg.py:78
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            x[k] = x[k] + alpha * p[k];                           // cg.py:79
            r[k] = r[k] - alpha * q[k];                           // cg.py:80
        }
        if (!(alive = apply_m())) break;                          // cg.py:81
        double rz_new_loc = 0.0, rr_loc = 0.0;
        asm volatile("" : "+v"(row0_l), "+v"(far_rows));
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) {
                rz_new_loc += r[k] * z[k];
                rr_loc += r[k] * r[k];
                publish_zp(k, z[k], p[k]);                        // z_{k+1} and p_k for the next update's gathers
            }
        double rz_new = 0.0, rr = 0.0;
        if (!(alive = chip_sum2(rz_new_loc, rr_loc, true, rz_new, rr))) break;
        beta = rz_new / rz;                                       // cg.py:82
#pragma unroll
        for (int k = 0; k < RPT; ++k) p[k] = z[k] + beta * p[k];  // cg.py:83
        rz = rz_new;
        res = rr / bb;                                            // cg.py:86
        ++k_done;
        if (v == 0 && t == 0 && k_done < d.hist_cap) d.hist[k_done] = res;
        const bool conv = (res < d.rtol_sq) || (rr < d.atol_sq);  // cg.py:71
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (alive && row_on(k)) d.x[row0 + k * kChipThreads] = x[k];
    if (TRACE && d.dbg) {
        __syncthreads();
        if (t < 64) d.dbg[v * 64 + t] = s_tr[t];
    }
    if (v == 0 && t == 0) {
        Scalars *sc = d.out;
        sc->k = k_done;
        sc->res = res;
        sc->bb = bb;
        sc->status = alive ? status : DPCG_ERR_STATE;
        sc->done = 1;
    }
}

// ---- the RESIDENT form (<= 4 rows a thread: up to 524 288 rows) ---------------------------------------------------------------------------
// With four rows a thread the factor fits on the chip BESIDE the matrix: A's slice moves to registers (28 values + 14 registers of 16-bit
// column offsets), and the LDS holds, per row, the off-diagonal entries of its row of L (in L's CSR order), then those of its row of L^T
// (in L^T's), the diagonal in the row's last slot -- for an incomplete Cholesky factor WITHOUT fill the two rows together are the
// off-diagonal pattern of A's row, so a row needs exactly the slots of A's -- values as doubles [slot][thread], columns as 16-bit offsets
// from the row, two to a word.  Nothing is streamed and there are no block lists: a row's dependency levels in L and in L^T sit in a
// register (6 bits each), a wave derives its blocks from them on the fly -- level by level, every lane that still has a row in the level
// gives its lowest such slot -- reads the entries of the lane's slot out of the LDS (per-lane addresses), gathers the operands,
// validates, solves, publishes.  The arithmetic and the hand-off (self-validating granules) are those of the streamed form.
//
// k_trsv_res_plan: one launch of 256 x 512 in the solve kernel's geometry writes what the solve kernel loads once: fval / fcol
// [v][slot][t] (slot = k * WMAX + j; columns in the handle's numbering, -1: none) and fmeta [v][k][t] = nL | nU << 4 | levelL << 8 |
// levelU << 14 | 1 << 20 (the row exists); flags[0] |= 1 when a row's two factor rows do not fit its WMAX - 1 slots.
__global__ __launch_bounds__(kChipThreads) void k_trsv_res_plan(int n, int per, int rpt, int wmax, const int32_t *__restrict__ lrp, const int32_t *__restrict__ lci,
                                                                const double *__restrict__ lval, const int32_t *__restrict__ urp, const int32_t *__restrict__ uci,
                                                                const double *__restrict__ uval, const int32_t *__restrict__ lvl_l, const int32_t *__restrict__ lvl_u,
                                                                const int32_t *__restrict__ f_of_handle, const int32_t *__restrict__ handle_of_f,
                                                                double *fval, int32_t *fcol, int32_t *fmeta, int *flags /* [0] misfit, [1] band */,
                                                                int tstride /* threads of a workgroup that can own a row: 512, or per rounded up to 64 for small systems --
                                                                               the arrays hold that many entries per slot */) {
    const int t = threadIdx.x, v = blockIdx.x;
    int band = 0;
    for (int k = 0; k < rpt && t < tstride; ++k) {
        const int loc = k * kChipThreads + t, i = v * per + loc;
        const bool valid = loc < per && i < n;
        const size_t mbase = ((size_t)v * rpt + k) * tstride + t;
        int meta = 0;
        for (int j = 0; j < wmax; ++j) {
            const size_t e = ((size_t)v * rpt * wmax + (size_t)k * wmax + j) * tstride + t;
            fval[e] = j == wmax - 1 ? 1.0 : 0.0;
            fcol[e] = -1;
        }
        if (valid) {
            const int f = f_of_handle ? f_of_handle[i] : i;
            const int ls = lrp[f], le = lrp[f + 1], us = urp[f], ue = urp[f + 1];
            const int nl = le - ls - 1, nu = ue - us - 1;
            if (nl + nu > wmax - 1 || nl > 15 || nu > 15) {
                atomicOr(&flags[0], 1);
            } else {
                for (int j = 0; j < nl; ++j) {                    // L: diagonal last
                    const size_t e = ((size_t)v * rpt * wmax + (size_t)k * wmax + j) * tstride + t;
                    const int c = handle_of_f ? handle_of_f[lci[ls + j]] : lci[ls + j];
                    fval[e] = lval[ls + j];
                    fcol[e] = c;
                    const int dlt = c > i ? c - i : i - c;
                    band = dlt > band ? dlt : band;
                }
                for (int j = 0; j < nu; ++j) {                    // L^T: diagonal first
                    const size_t e = ((size_t)v * rpt * wmax + (size_t)k * wmax + nl + j) * tstride + t;
                    const int c = handle_of_f ? handle_of_f[uci[us + 1 + j]] : uci[us + 1 + j];
                    fval[e] = uval[us + 1 + j];
                    fcol[e] = c;
                    const int dlt = c > i ? c - i : i - c;
                    band = dlt > band ? dlt : band;
                }
                fval[((size_t)v * rpt * wmax + (size_t)k * wmax + wmax - 1) * tstride + t] = lval[le - 1];
                meta = nl | (nu << 4) | (lvl_l[f] << 8) | (lvl_u[f] << 14) | (1 << 20);
            }
        }
        fmeta[mbase] = meta;
    }
    for (int off = 32; off > 0; off >>= 1) { const int ob = __shfl_xor(band, off); band = ob > band ? ob : band; }
    if ((t & 63) == 0) atomicMax(&flags[1], band);
}

template <int RPT, int WMAX, bool TRACE = false>
__global__ __launch_bounds__(kChipThreads) void k_pcg_chip_trsv_res(const ChipTrsvDesc d) {
    constexpr int NS = RPT * WMAX;
    constexpr int WL = WMAX - 1;
    constexpr int NOFF = (NS + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double res_lds[];   // factor values [NS][512], then column offsets [NOFF][512] words
    __shared__ double sh[2 * 16];
    __shared__ double s_res[2][2];
    __shared__ int s_flag;
    __shared__ unsigned long long s_tr[TRACE ? 8 * 8 : 1];
    double *const f_val = res_lds;
    unsigned *const f_off = reinterpret_cast<unsigned *>(res_lds + NS * kChipThreads);
    const int t = threadIdx.x, lane = t & 63;
    const int v = ((int)blockIdx.x & 7) * (kChipWGs / 8) + ((int)blockIdx.x >> 3);
    const int row0 = v * d.per + t;
    const int grp = (int)blockIdx.x & 7, rank = (int)blockIdx.x >> 3;
    const int glo = grp * (kChipWGs / 8) * d.per;
    const int ghi = (glo + (kChipWGs / 8) * d.per < d.n) ? glo + (kChipWGs / 8) * d.per : d.n;
    const int remote_base = (d.n + kChipZpPad) * 16;
    const unsigned gran_bytes = 2u * (unsigned)(d.n + kChipZpPad) * 16u;
    const __amdgpu_buffer_rsrc_t zp_rs = chip_rsrc(d.zp, gran_bytes);
    const __amdgpu_buffer_rsrc_t y_rs = chip_rsrc(d.ypub, gran_bytes);
    const __amdgpu_buffer_rsrc_t w_rs = chip_rsrc(d.zpub, gran_bytes);

    // ---- A's slice (registers), the factor's (LDS), the vectors of the own rows: read once -------------------------------------------------
    double va[NS];
    unsigned dl[NOFF];
    constexpr int LB = WMAX > 7 ? 8 : 4;
    constexpr unsigned LV = 1u << (LB - 1), LM = LV - 1u;
    static_assert(LB * RPT <= 32, "row lengths of a thread in one register");
    unsigned lens = 0;
    int meta[RPT];                           // nL | nU << 4 | levelL << 8 | levelU << 14 | exists << 20
    double x[RPT], r[RPT], p[RPT], q[RPT], z[RPT];
    double bb_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int loc = k * kChipThreads + t, i = row0 + k * kChipThreads;
        const bool valid = loc < d.per && i < d.n;
        const int ic = valid ? i : 0;
        const int rs = d.rp[ic], re = d.rp[ic + 1];
        const int len = valid ? re - rs : 0;
        lens |= (valid ? (LV | (unsigned)len) : 0u) << (LB * k);
        x[k] = valid ? (d.x0 ? d.x0[ic] : 0.0) : 0.0;
        r[k] = valid ? d.b[ic] : 0.0;
        p[k] = q[k] = z[k] = 0.0;
        if (valid) bb_loc += r[k] * r[k];
        meta[k] = t < d.tstride ? d.fmeta[(v * RPT + k) * d.tstride + t] : 0;
#pragma unroll
        for (int j = 0; j < WMAX; ++j) {
            const int s = k * WMAX + j;
            const bool on = j < len;
            const int e = on ? rs + j : 0;
            const int c = on ? d.ci[e] : i;
            va[s] = on ? d.val[e] : 0.0;
            const unsigned del = (unsigned)(c - i + 32768) & 0xffffu;
            if (s & 1) dl[s >> 1] |= del << 16;
            else dl[s >> 1] = del;
        }
    }
    {
        unsigned fo[NOFF];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = row0 + k * kChipThreads;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int s = k * WMAX + j;
                const size_t e = ((size_t)v * NS + s) * d.tstride + t;
                const bool has = t < d.tstride;                   // (small systems: the plan holds the threads that can own a row)
                f_val[s * kChipThreads + t] = has ? d.fval[has ? e : 0] : (j == WMAX - 1 ? 1.0 : 0.0);
                const int c = has ? d.fcol[has ? e : 0] : -1;
                const unsigned del = c >= 0 ? (unsigned)(c - i + 32768) & 0xffffu : 0u;
                if (s & 1) fo[s >> 1] |= del << 16;
                else fo[s >> 1] = del;
            }
        }
#pragma unroll
        for (int w = 0; w < NOFF; ++w) f_off[w * kChipThreads + t] = fo[w];
    }
    bool local = false;
    auto row_on = [&](int k) -> bool { return ((lens >> (LB * k)) & LV) != 0; };

    // q = A p_k for the own rows, the gathered entries of p_k recomputed from the granules {z_k, p_{k-1}} (dpcg_chip.hip)
    auto spmv = [&](double beta) {
        u32x4 g[2][WMAX];
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        const int local_shift = grp * 128;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));
#pragma unroll
        for (int e = 0; e < NOFF; ++e) asm volatile("" : "+v"(dl[e]));
        asm volatile("" : "+v"(lens));
        auto request = [&](int k, u32x4 (&gk)[WMAX]) {
            const int rowk = row0 + k * kChipThreads;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int s = k * WMAX + j;
                const int del = (int)((dl[s >> 1] >> (16 * (s & 1))) & 0xffffu);
                const int c = rowk + del - 32768;
                const bool own = (unsigned)(c - glo_l) < (unsigned)span_l;
                gk[j] = __builtin_amdgcn_raw_buffer_load_b128(zp_rs, c * 16 + (own ? local_shift : remote_base), 0, kSc1);
            }
        };
        request(0, g[0]);
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            if (k + 1 < RPT) request(k + 1, g[(k + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int len = (int)((lens >> (LB * k)) & LM);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const double pc = lo_f64(g[k & 1][j]) + beta * hi_f64(g[k & 1][j]);  // = p_k[c], cg.py:83
                if (j < len) acc += va[k * WMAX + j] * pc;
            }
            q[k] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (TRACE && t < 64) s_tr[t] = 0;
    auto tick = [&]() -> unsigned long long { return TRACE ? wall_clock64() : 0ull; };
    auto tr_add = [&](int phase, unsigned long long dt) {
        if (TRACE && lane == 0) s_tr[(t >> 6) * 8 + phase] += dt;
    };
    Exchange X;
    X.part_rs = chip_rsrc(d.part, (unsigned)kChipSlotBytes);
    X.v = v; X.grp = grp; X.rank = rank; X.sh = sh; X.s_res = s_res; X.s_flag = &s_flag; X.err = d.err;
    auto chip_sum2 = [&](double a, double b2, bool publish, double &ra, double &rb) -> bool { return exchange2(X, a, b2, publish, ra, rb); };
    unsigned far_rows = 0xffu;
    int row0_l = row0;
    auto publish_zp = [&](int k, double zk, double pk) {
        const int o = (row0_l + k * kChipThreads) * 16;
        if (local) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + grp * 128, 0, 0);
        if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + remote_base, 0, kSc1);
    };
    unsigned pub_gen = 0;
    auto make_key = [&](unsigned gen) -> unsigned long long {
        return (((unsigned long long)d.nonce << 32) | (unsigned long long)gen) * 0x9E3779B97F4A7C15ull | 1ull;
    };
    constexpr int kNoCol = 0x7ffffff0;
    // one triangular solve (fi = 0: L y = rhs, 1: L^T z = rhs) on the resident factor
    auto tri_solve = [&](int fi, int nlev, const __amdgpu_buffer_rsrc_t &out_rs, const double (&rhs)[RPT], double (&out)[RPT], unsigned long long key) __attribute__((always_inline)) -> bool {
        const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));
        int tl = t;
        asm volatile("" : "+v"(tl));                              // (keeps the LDS addresses of the loop below out of hoisted registers)
        const double *fv = f_val + tl;
        const unsigned *fo = f_off + tl;
        const int lsh = 8 + 6 * fi;
        auto publish = [&](int k, double res) {
            const int o = (row0 + k * kChipThreads) * 16;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(res), tag = bits ^ key;
            u32x4 w;
            w.x = (unsigned)bits; w.y = (unsigned)(bits >> 32); w.z = (unsigned)tag; w.w = (unsigned)(tag >> 32);
            if (local) __builtin_amdgcn_raw_buffer_store_b128(w, out_rs, o + grp * 128, 0, 0);
            if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(w, out_rs, o + remote_base, 0, kSc1);
        };
        const unsigned long long tk0 = tick();
        // level 0: no dependencies -- slot by slot
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const bool on0 = (meta[k] >> 20) != 0 && ((meta[k] >> lsh) & 63) == 0;
            const double res = rhs[k] / (on0 ? fv[(k * WMAX + WMAX - 1) * kChipThreads] : 1.0);
            if (on0) publish(k, res);
            out[k] = on0 ? res : out[k];
        }
        const unsigned long long tk1 = tick();
        tr_add(3 * fi, tk1 - tk0);
        // The blocks of the levels >= 1, derived on the fly: `lvl_cur` walks the levels, `mk` holds the lane's slots of that level not yet
        // taken; a block = every lane with such a slot gives its lowest one: its operands are gathered (their addresses out of the LDS),
        // validated, the rows solved and published.
        struct Blk { int k; bool act; int e0, cnt; u32x4 g[WL]; };
        int lvl_cur = 0;
        unsigned mk = 0;
        // byte offset of the granule entry j of block B gathers (out of the LDS every time it is wanted: six registers a block less)
        auto gather_off = [&](const Blk &B, int j) __attribute__((always_inline)) -> int {
            const bool on = j < B.cnt;
            const int slc = on ? B.e0 + j : 0;
            const unsigned w = fo[(slc >> 1) * kChipThreads];
            const int del = (int)((w >> (16 * (slc & 1))) & 0xffffu);
            const int c = row0 + B.k * kChipThreads + del - 32768;
            const bool own = (unsigned)(c - glo_l) < (unsigned)span_l;
            return on ? c * 16 + (own ? grp * 128 : remote_base) : kNoCol;
        };
        auto next_block = [&](Blk &B) __attribute__((always_inline)) -> bool {          // false: the levels are used up (the same for the whole wave)
            while (__ballot(mk != 0) == 0) {
                if (++lvl_cur >= nlev) return false;
                mk = 0;
#pragma unroll
                for (int k = 0; k < RPT; ++k) mk |= (((meta[k] >> 20) != 0 && ((meta[k] >> lsh) & 63) == lvl_cur) ? 1u : 0u) << k;
            }
            B.act = mk != 0;
            B.k = B.act ? __builtin_ctz(mk) : 0;
            mk &= mk - 1u;
            int mt = meta[0];
#pragma unroll
            for (int kk = 1; kk < RPT; ++kk) {
                int cand = meta[kk];
                asm volatile("" : "+v"(cand));                    // (opaque: see the selection of the row's operand below)
                mt = B.k == kk ? cand : mt;
            }
            const int nl = mt & 15, nu = (mt >> 4) & 15;
            B.e0 = B.k * WMAX + (fi ? nl : 0);
            B.cnt = B.act ? (fi ? nu : nl) : 0;
#pragma unroll
            for (int j = 0; j < WL; ++j) B.g[j] = __builtin_amdgcn_raw_buffer_load_b128(out_rs, gather_off(B, j), 0, kSc1);
            return true;
        };
        int ok = 1;
        auto finish = [&](Blk &B) __attribute__((always_inline)) {
            auto stale = [&]() -> bool {
                bool bad = false;
#pragma unroll
                for (int j = 0; j < WL; ++j) bad = bad || (j < B.cnt && ((B.g[j].x ^ B.g[j].z) != klo || (B.g[j].y ^ B.g[j].w) != khi));
                return bad;
            };
            // the entries' values and the diagonal: out of the LDS while the gathers land
            double a[WL];
#pragma unroll
            for (int j = 0; j < WL; ++j) a[j] = fv[(j < B.cnt ? B.e0 + j : 0) * kChipThreads];
            const double dg = fv[(B.k * WMAX + WMAX - 1) * kChipThreads];
            if (__ballot(stale()) != 0) {
                const unsigned long long tp0 = tick();
                tr_add(7, 1);
                unsigned spins = 0;
                unsigned long long t0 = 0;
                for (;;) {
                    __builtin_amdgcn_s_sleep(1);
                    if (stale()) {
#pragma unroll
                        for (int j = 0; j < WL; ++j) B.g[j] = __builtin_amdgcn_raw_buffer_load_b128(out_rs, gather_off(B, j), 0, kSc1);
                    }
                    if (__ballot(stale()) == 0) break;
                    if ((++spins & 255u) == 0) {
                        const unsigned long long now = wall_clock64();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > kChipSpinTicks || __hip_atomic_load(d.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            atomicExch(d.err, 1);
                            ok = 0;
                            break;
                        }
                    }
                }
                tr_add(3 * fi + 2, tick() - tp0);
            }
            ok = __builtin_amdgcn_readfirstlane(ok);
            const int k = B.k;
            double acc = rhs[0];
#pragma unroll
            for (int kk = 1; kk < RPT; ++kk) {
                double cand = rhs[kk];
                asm volatile("" : "+v"(cand));
                acc = k == kk ? cand : acc;
            }
#pragma unroll
            for (int j = 0; j < WL; ++j)
                if (j < B.cnt) acc -= a[j] * lo_f64(B.g[j]);
            const double res = acc / (B.act ? dg : 1.0);
            if (B.act) publish(k, res);
#pragma unroll
            for (int kk = 0; kk < RPT; ++kk) {
                double keep = out[kk];
                asm volatile("" : "+v"(keep));
                out[kk] = (B.act && k == kk) ? res : keep;
            }
        };
        // Blocks of ONE level do not depend on each other: while a block is validated and solved, the operands of the level's next block
        // are already gathered (two named register sets in turn).  Never across a level boundary: there the operands are not published yet
        // -- gathering ahead made twice as many blocks poll again (512 K rows, 18 levels: 94 -> 112 us per update).
        // (With four rows a thread the second register set spills and the lookahead loses -- 512 K rows, two colours: 23.2 -> 25.0 us per
        // update; with the iterate x moved into the LDS to make room it fits without a spill and gains nothing: 24.3 against 24.1-24.3 -- so it
        // is taken up to two rows a thread: 167 K-row quadtree mesh, four colours: 26.9 -> 24.5; its caller's order, 13 levels: 57.6 -> 47.6.)
        if constexpr (RPT <= 2) {
            Blk Ba, Bb;
            bool have_a = false;
            for (;;) {
                if (!have_a && !next_block(Ba)) break;
                have_a = false;
                const bool ahead_b = __ballot(mk != 0) != 0;      // the level has another block
                if (ahead_b) next_block(Bb);
                finish(Ba);
                if (ok == 0) break;
                if (!ahead_b) continue;
                if (__ballot(mk != 0) != 0) have_a = next_block(Ba);
                finish(Bb);
                if (ok == 0) break;
            }
        } else {
            Blk B;
            while (ok != 0 && next_block(B)) finish(B);
        }
        tr_add(3 * fi + 1, tick() - tk1);
        return ok != 0;
    };
    auto apply_m = [&]() -> bool {
        const unsigned long long key_y = make_key(++pub_gen);
        bool ok = tri_solve(0, d.nlev_l, y_rs, r, q, key_y);
        const unsigned long long key_z = make_key(++pub_gen);
        if (ok) ok = tri_solve(1, d.nlev_u, w_rs, q, z, key_z);
        const unsigned long long ts0 = tick();
        const bool all_ok = __syncthreads_and(ok ? 1 : 0) != 0;
        tr_add(6, tick() - ts0);
        return all_ok;
    };

    __syncthreads();                                              // (the factor's slice is in the LDS)
    bool alive = true;
    double dummy = 0.0, dummy2 = 0.0;
    if (d.xcc) {
        local = groups_on_one_xcd(X, d.xcc, alive);
        X.local = local;
        if (local) {
            far_rows = 0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = row0 + k * kChipThreads;
                if (i < glo + d.band || i >= ghi - d.band) far_rows |= 1u << k;
            }
        }
    }
    // ---- cg.py:58-67 -------------------------------------------------------------------------------------------------
    if (alive && d.x0) {
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) publish_zp(k, x[k], 0.0);
        alive = chip_sum2(0.0, 0.0, true, dummy, dummy2);
        if (alive) {
            spmv(0.0);
#pragma unroll
            for (int k = 0; k < RPT; ++k) r[k] = r[k] - q[k];
            alive = chip_sum2(0.0, 0.0, false, dummy, dummy2);
        }
    }
    if (alive) alive = apply_m();                                 // cg.py:61
    double rz_loc = 0.0, t0_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        p[k] = z[k];                                              // cg.py:62
        if (row_on(k)) {
            rz_loc += r[k] * z[k];
            t0_loc += d.init_check_r ? r[k] * r[k] : z[k] * z[k];  // cg.py:66
            publish_zp(k, z[k], 0.0);
        }
    }
    double bb = 0.0, rz = 0.0, tt = 0.0;
    if (alive) alive = chip_sum2(bb_loc, rz_loc, true, bb, rz);
    if (alive) alive = chip_sum2(t0_loc, 0.0, false, tt, dummy);
    double res = tt / bb, beta = 0.0;
    int k_done = 0, status = DPCG_MAX_ITER;
    bool stop = false;
    if (alive) {
        if (v == 0 && t == 0 && d.hist_cap > 0) d.hist[0] = res;
        const bool conv = (res < d.rtol_sq) || (tt < d.atol_sq);
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
    // ---- cg.py:70-87 -----------------------------------------------------------------------------------------------------
    while (alive && !stop && k_done < d.max_iter) {
        spmv(beta);                                               // cg.py:75
        double pq_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) pq_loc += q[k] * p[k];
        double pq = 0.0;
        if (!(alive = chip_sum2(pq_loc, 0.0, false, pq, dummy))) break;
        const double alpha = rz / pq;                             // cg.py:78
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            x[k] = x[k] + alpha * p[k];                           // cg.py:79
            r[k] = r[k] - alpha * q[k];                           // cg.py:80
        }
        if (!(alive = apply_m())) break;                          // cg.py:81
        double rz_new_loc = 0.0, rr_loc = 0.0;
        asm volatile("" : "+v"(row0_l), "+v"(far_rows));
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) {
                rz_new_loc += r[k] * z[k];
                rr_loc += r[k] * r[k];
                publish_zp(k, z[k], p[k]);
            }
        double rz_new = 0.0, rr = 0.0;
        if (!(alive = chip_sum2(rz_new_loc, rr_loc, true, rz_new, rr))) break;
        beta = rz_new / rz;                                       // cg.py:82
#pragma unroll
        for (int k = 0; k < RPT; ++k) p[k] = z[k] + beta * p[k];  // cg.py:83
        rz = rz_new;
        res = rr / bb;                                            // cg.py:86
        ++k_done;
        if (v == 0 && t == 0 && k_done < d.hist_cap) d.hist[k_done] = res;
        const bool conv = (res < d.rtol_sq) || (rr < d.atol_sq);  // cg.py:71
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (alive && row_on(k)) d.x[row0 + k * kChipThreads] = x[k];
    if (TRACE && d.dbg) {
        __syncthreads();
        if (t < 64) d.dbg[v * 64 + t] = s_tr[t];
    }
    if (v == 0 && t == 0) {
        Scalars *sc = d.out;
        sc->k = k_done;
        sc->res = res;
        sc->bb = bb;
        sc->status = alive ? status : DPCG_ERR_STATE;
        sc->done = 1;
    }
}

template <int RPT, int WMAX, bool TRACE = false>
int chip_trsv_res_launch(const ChipTrsvDesc &d, hipStream_t s, bool check_only) {
    constexpr int NS = RPT * WMAX;
    const int lds = NS * kChipThreads * (int)sizeof(double) + ((NS + 1) / 2) * kChipThreads * (int)sizeof(unsigned);
    static int resident = -1;
    if (resident < 0) {
        if (hipFuncSetAttribute((const void *)k_pcg_chip_trsv_res<RPT, WMAX, TRACE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return DPCG_ERR_HIP;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_pcg_chip_trsv_res<RPT, WMAX, TRACE>, kChipThreads, (size_t)lds) != hipSuccess)
            return DPCG_ERR_HIP;
        resident = per_cu;
    }
    if (resident < 1) return DPCG_ERR_STATE;
    if (check_only) return DPCG_OK;
    hipLaunchKernelGGL((k_pcg_chip_trsv_res<RPT, WMAX, TRACE>), dim3(kChipWGs), dim3(kChipThreads), (size_t)lds, s, d);
    return DPCG_OK;
}

template <int RPT, int WMAX, int WL, bool TRACE = false>
int chip_trsv_launch(const ChipTrsvDesc &d, hipStream_t s, bool check_only) {
    constexpr int NS = RPT * WMAX;
    constexpr int NLDS = NS < kChipLdsSlots ? NS : kChipLdsSlots;
    const int lds = NLDS * kChipThreads * (int)sizeof(double);
    static int resident = -1;
    if (resident < 0) {
        if (hipFuncSetAttribute((const void *)k_pcg_chip_trsv<RPT, WMAX, WL, TRACE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return DPCG_ERR_HIP;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_pcg_chip_trsv<RPT, WMAX, WL, TRACE>, kChipThreads, (size_t)lds) != hipSuccess)
            return DPCG_ERR_HIP;
        resident = per_cu;
    }
    if (resident < 1) return DPCG_ERR_STATE;
    if (check_only) return DPCG_OK;
    hipLaunchKernelGGL((k_pcg_chip_trsv<RPT, WMAX, WL, TRACE>), dim3(kChipWGs), dim3(kChipThreads), (size_t)lds, s, d);
    return DPCG_OK;
}

}  // namespace

int chip_trsv_max_levels() { return kTrsvMaxLevels; }
int chip_trsv_max_factor_row() { return 8; }
int64_t chip_trsv_diag_doubles() { return (int64_t)kChipWGs * kChipMaxRpt * kChipThreads; }      // off-diagonal entries of a factor row

// The block lists of one triangular factor (see the header).  lvl: dependency level per factor row (device).  Allocates plan arrays.
int build_chip_trsv_lists(int n, int per, int nlev, const CsrDev &F, const int32_t *lvl, const int32_t *f_of_handle, const int32_t *handle_of_f, bool upper,
                          ChipTrsvLists &out, int32_t *lv0, double *diag0, hipStream_t s) {
    constexpr int NW = kChipWGs * (kChipThreads / 64);
    int2 *counts = nullptr;
    int *band_wl = nullptr;
    int st = DPCG_OK;
    auto fail = [&](int code) { dev_free(counts); dev_free(band_wl); return code; };
    if ((st = dev_alloc(&counts, NW)) < 0) return fail(st);
    if ((st = dev_alloc(&band_wl, 2)) < 0) return fail(st);
    if ((st = dev_alloc(&out.first_blk, NW + 1)) < 0) return fail(st);
    if ((st = dev_alloc(&out.first_ent, NW + 1)) < 0) return fail(st);
    if (hipMemsetAsync(band_wl, 0, 2 * sizeof(int), s) != hipSuccess) return fail(DPCG_ERR_HIP);
    hipLaunchKernelGGL(k_trsv_plan<false>, dim3(kChipWGs), dim3(kChipThreads), 0, s, n, per, nlev, F.rowptr, F.col, F.val, lvl, f_of_handle, handle_of_f,
                       upper ? 1 : 0, counts, (const int32_t *)nullptr, (const int32_t *)nullptr, (int4 *)nullptr, (double *)nullptr, (int32_t *)nullptr, band_wl,
                       (int32_t *)nullptr, (double *)nullptr);
    hipLaunchKernelGGL(k_trsv_scan, dim3(1), dim3(1024), 0, s, counts, out.first_blk, out.first_ent);
    int tot[2] = {0, 0}, bw[2] = {0, 0};
    if (hipMemcpyAsync(&tot[0], out.first_blk + NW, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return fail(DPCG_ERR_HIP);
    if (hipMemcpyAsync(&tot[1], out.first_ent + NW, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return fail(DPCG_ERR_HIP);
    if (hipMemcpyAsync(bw, band_wl, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return fail(DPCG_ERR_HIP);
    if (hipStreamSynchronize(s) != hipSuccess) return fail(DPCG_ERR_HIP);
    out.n_blk = tot[0];
    out.n_ent = tot[1];
    out.band = bw[0];
    out.max_row = bw[1];
    if ((st = dev_alloc(&out.blk, (int64_t)std::max(tot[0], 1))) < 0) return fail(st);
    if ((st = dev_alloc(&out.val, (int64_t)std::max(tot[1], 1))) < 0) return fail(st);
    if ((st = dev_alloc(&out.col, (int64_t)std::max(tot[1], 1))) < 0) return fail(st);
    hipLaunchKernelGGL(k_trsv_plan<true>, dim3(kChipWGs), dim3(kChipThreads), 0, s, n, per, nlev, F.rowptr, F.col, F.val, lvl, f_of_handle, handle_of_f,
                       upper ? 1 : 0, (int2 *)nullptr, out.first_blk, out.first_ent, out.blk, out.val, out.col, (int *)nullptr, lv0, diag0);
    if (hipGetLastError() != hipSuccess) return fail(DPCG_ERR_HIP);
    dev_free(counts);
    dev_free(band_wl);
    return DPCG_OK;
}

void free_chip_trsv_lists(ChipTrsvLists &l) {
    dev_free(l.first_blk); dev_free(l.first_ent); dev_free(l.blk); dev_free(l.val); dev_free(l.col);
    l = ChipTrsvLists();
}

// The resident form's plan (see k_trsv_res_plan): rpt = rows a thread (1, 2, 4), wmax = the solve kernel's slots per row (5, 7, 9).
// *misfit: some row's two factor rows do not fit its wmax - 1 slots; *band: largest |col - row| of the factor in the handle's numbering.
int build_chip_trsv_resident(int n, int per, int rpt, int wmax, const CsrDev &L, const CsrDev &U, const int32_t *lvl_l, const int32_t *lvl_u,
                             const int32_t *f_of_handle, const int32_t *handle_of_f, double **fval, int32_t **fcol, int32_t **fmeta, int *misfit, int *band,
                             int *tstride_out, hipStream_t s) {
    int *flags = nullptr;
    int st;
    if ((st = dev_alloc(&flags, 2)) < 0) return st;
    // a system of a few thousand rows gives a workgroup a handful of them: the plan holds only the threads that can own one
    const int tstride = per >= kChipThreads ? kChipThreads : std::min(kChipThreads, ((per + 63) / 64) * 64);
    *tstride_out = tstride;
    const int64_t slots = (int64_t)kChipWGs * rpt * wmax * tstride;
    if ((st = dev_alloc(fval, slots)) < 0 || (st = dev_alloc(fcol, slots)) < 0 || (st = dev_alloc(fmeta, (int64_t)kChipWGs * rpt * tstride)) < 0) {
        dev_free(flags);
        return st;
    }
    int h_flags[2] = {0, 0};
    hipError_t e = hipMemsetAsync(flags, 0, 2 * sizeof(int), s);
    hipLaunchKernelGGL(k_trsv_res_plan, dim3(kChipWGs), dim3(kChipThreads), 0, s, n, per, rpt, wmax, L.rowptr, L.col, L.val, U.rowptr, U.col, U.val, lvl_l, lvl_u,
                       f_of_handle, handle_of_f, *fval, *fcol, *fmeta, flags, tstride);
    if (e == hipSuccess) e = hipMemcpyAsync(h_flags, flags, sizeof(h_flags), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    dev_free(flags);
    if (e != hipSuccess) return DPCG_ERR_HIP;
    *misfit = h_flags[0];
    *band = h_flags[1];
    return DPCG_OK;
}
int chip_trsv_resident_rpt(int per) { const int rpt = (per + kChipThreads - 1) / kChipThreads; return rpt <= 2 ? 2 : (rpt <= 4 ? 4 : 0); }
int chip_trsv_resident_wmax(int max_a, int rpt) { return max_a <= 5 ? 5 : (max_a <= 7 ? 7 : (max_a <= 9 && rpt <= 2 ? 9 : 0)); }

int launch_pcg_chip_trsv_resident(const ChipTrsvDesc &d, int rpt, int wmax, hipStream_t s, bool check_only) {
    const bool tr = d.dbg != nullptr;
    if (rpt == 2 && wmax == 5) return tr ? DPCG_ERR_STATE : chip_trsv_res_launch<2, 5>(d, s, check_only);
    if (rpt == 2 && wmax == 7) return tr ? chip_trsv_res_launch<2, 7, true>(d, s, check_only) : chip_trsv_res_launch<2, 7>(d, s, check_only);
    if (rpt == 2 && wmax == 9) return tr ? DPCG_ERR_STATE : chip_trsv_res_launch<2, 9>(d, s, check_only);
    if (rpt == 4 && wmax == 5) return tr ? DPCG_ERR_STATE : chip_trsv_res_launch<4, 5>(d, s, check_only);
    if (rpt == 4 && wmax == 7) return tr ? chip_trsv_res_launch<4, 7, true>(d, s, check_only) : chip_trsv_res_launch<4, 7>(d, s, check_only);
    return DPCG_ERR_STATE;
}

// max_a: longest row of A; max_l: most off-diagonal entries of a factor row.  DPCG_OK, DPCG_ERR_STATE (cannot be resident / shape not
// compiled), or a negative status.
int launch_pcg_chip_trsv(const ChipTrsvDesc &d, int max_a, int max_l, hipStream_t s, bool check_only) {
    if (max_a < 1 || max_l < 0 || d.per < 1 || d.per > kChipThreads * kChipMaxRpt) return DPCG_ERR_INVALID;
    const int rpt = (d.per + kChipThreads - 1) / kChipThreads;
    const int wa = max_a <= 5 ? 5 : (max_a <= 7 ? 7 : 9);
    if (max_a > 9 || (wa == 9 && rpt > 4)) return DPCG_ERR_STATE;
    if (max_l > wa - 1) return DPCG_ERR_STATE;                 // (factor rows of an incomplete Cholesky WITHOUT fill: a subset of A's row)
    if (d.dbg) {                                               // (development: the traced variants exist for 7-entry rows only)
        if (wa != 7) return DPCG_ERR_STATE;
        return rpt <= 2 ? chip_trsv_launch<2, 7, 6, true>(d, s, check_only) : (rpt <= 4 ? chip_trsv_launch<4, 7, 6, true>(d, s, check_only) : chip_trsv_launch<8, 7, 6, true>(d, s, check_only));
    }
#define DPCG_TRSV_R(RPTV) (wa == 5 ? chip_trsv_launch<RPTV, 5, 4>(d, s, check_only) : chip_trsv_launch<RPTV, 7, 6>(d, s, check_only))
#define DPCG_TRSV_R9(RPTV) (wa == 9 ? chip_trsv_launch<RPTV, 9, 8>(d, s, check_only) : DPCG_TRSV_R(RPTV))
    if (rpt <= 2) return DPCG_TRSV_R9(2);
    if (rpt <= 4) return DPCG_TRSV_R9(4);
    return DPCG_TRSV_R(8);
#undef DPCG_TRSV_R
#undef DPCG_TRSV_R9
}

}  // namespace dpcg
