// Whole-solve kernel for LARGE cache-sized systems (65 537 .. 1 048 576 rows, rows of <= 7 entries, M = I / Jacobi): the WHOLE CHIP as
// one team -- hand-written for gfx950 (MI355X: 256 CUs in 8 XCDs, wave64, 512 KiB of vector registers and 160 KiB of LDS per CU).
//
// The multi-launch update of the 1M-DoF headline system (cg.py:70-87) is three launches: the SpMV streams the matrix (83 MB) out of
// the Infinity Cache at 0.79 of the HBM peak in 16.4 us, the two vector kernels stream r, p, x, dinv (64-88 MB) in another 14.8 us
// behind two kernel boundaries -- every update, although NOTHING of it changes but four vectors.  The chip holds 128 MiB of
// registers and 40 MiB of LDS: the matrix (8 + 2 bytes per entry) AND the vectors fit.  So here the solve (cg.py:58-90) is ONE launch:
//   * 256 workgroups of 512 threads, one per CU (the kernel takes all of a CU's LDS, so the hardware cannot place two).  Workgroup
//     v (its "virtual" index: blocks b, b + 8, ... share an XCD, and v = (b % 8) * 32 + b / 8 gives each XCD one contiguous eighth
//     of the rows -- placement is a speed matter only) owns rows [v * per, (v + 1) * per), per = ceil(n / 256); thread t owns rows
//     v * per + t + 512 k, k < RPT <= 8: consecutive lanes, consecutive rows, so what a wave gathers for one entry slot of a
//     banded matrix is one contiguous run.
//   * the matrix slice of a thread is read ONCE per solve: the values of its RPT x WMAX entry slots go to LDS (39 slots per thread:
//     conflict-free private columns [slot][t]) and, beyond that, to registers; the columns are kept as 16-bit offsets from the row
//     (col - row + 32768: the handle's matrix must have a half-bandwidth below 32 768 -- stencils and RCM-ordered meshes do),
//     two to a register.  x, r, p, q, dinv of the own rows live in registers for the whole solve.
//   * what the other workgroups need of an update is PUBLISHED as 16-byte granules {z_{k+1}[i], p_k[i]} (one store per row and
//     update); a row gathers one granule per entry and recomputes p_{k+1}[c] = z_{k+1}[c] + beta p_k[c] -- the owner's
//     expression, and with contraction off the owner's bits (cg.py:83) -- so that publishing needs no barrier of its own.
//   * TWO chip-wide reductions per update (<p,Ap> | <r,z>, <r,r>), which double as the barriers, in two hops: the 32 workgroups of a
//     GROUP (equal blockIdx % 8: one XCD under the usual round-robin placement) exchange their partial pairs through 16-byte slots
//     that held a reserved NaN pattern -- one store each, wave 0 of every workgroup polls the group's 32 slots -- and sum them in
//     one fixed tree; eight members of each group then hand the group's pair to the eight groups (one 128-byte line of eight slots
//     per destination group, polled by that group's 32 workgroups only: a flat exchange had all 256 CUs polling the same 32 lines
//     and cost 4 us a barrier).  Every workgroup adds the same values in the same order: bit-identical alpha / beta everywhere,
//     the decision of cg.py:71 taken identically, no broadcast.  Four slot sets rotate, a slot is re-armed two generations ahead
//     behind a drain (dpcg_team.hip explains why two).
//   * VISIBILITY.  Data another workgroup reads is stored with agent scope (sc1: written through to the memory side) and loaded
//     with sc1 loads that bypass the CU's L1, every wave drains its stores before its workgroup signals -- correct for ANY
//     placement (MI355X_MICROARCH.md, "Valid forms").  Written through, however, a line leaves the L2, and every one of the
//     seven gathers of a granule then travels to the memory side: 117 MB per update, 25-30 us.  So each workgroup reports its XCD
//     (XCC_ID) once per solve, and when every group does sit on one XCD the granules are kept TWICE: a copy stored plainly -- it
//     stays in the XCD's shared L2, complete there once the storing wave's vmcnt has drained, where the group's sc1 loads hit
//     it -- and, only for the rows within the matrix's bandwidth of the group's first and last row, a written-through copy for the
//     neighbouring groups.  A gather takes the plain copy when the column belongs to the own group, the other one otherwise; the
//     group-level slots are stored plainly too.  Any other placement keeps everything written through.  (A stale line anywhere
//     would change the residual history, which the tests compare bit for bit with the CPU restatement over 1.3e9 gathers a solve.)
//   * co-residency is checked up front (occupancy query x CUs >= 256) and every wait is bounded (20 ms): a launch whose workgroups
//     cannot all become resident (another process's kernel holding CUs) reports DPCG_ERR_STATE and the caller solves through the
//     multi-launch path instead.
// Row sums run in CSR order and every dot product in a fixed tree (restated by the tests' CPU checker, form "chip"), so history, count and
// x equal the CPU restatement's bit for bit.
#include <algorithm>
#include <type_traits>

#include "dpcg_chip_device.h"

namespace dpcg {

using namespace chip;

namespace {

// RPT: rows per thread (2, 4, 8); WMAX: entry slots per row (5, 7; 9 up to four rows a thread: unstructured meshes); JAC: M = diag(1 / a_ii) (else M = I: z = r, no dinv registers).
template <int RPT, int WMAX, bool JAC, int MODE>
__global__ __launch_bounds__(kChipThreads) void k_pcg_chip(const ChipDesc d) {
    constexpr bool TRACE = MODE == 1;      // (development -- MODE 2: q = p instead of the gathers, the loop never stops before max_iter; MODE 3: the gathers
                                           // issued, but every one out of range: no memory request, zeros returned)
    // MODE 4: BASELINE config 5 -- `A @ pk` (cg.py:75) with the matrix values and pk STORED in fp32, products and sums in fp64 (the
    // contract of DPCG_SPMV_F32: the CPU restatement's mixed product).  The values are rounded once, where they are read; a gathered p is
    // rounded where it is recomputed.  Everything else is the fp64 arithmetic of MODE 0 (x0 = 0 only: cg.py:60 reads the fp64 A).
    constexpr bool F32 = MODE == 4 || MODE == 6;                      // (6: the streamed form with fp32-stored operands)
    // MODE 5: the matrix is NOT resident -- rows too long (or columns too far) for the slots above: a 1M-row finite-volume mesh with
    // rows of 9, a Delaunay graph with rows of 21.  Same skeleton (vectors in registers, published granules, two exchanges an update),
    // but q = A p streams the workgroup's CSR every update, wave by wave (see spmv_stream below).
    constexpr bool STREAM = MODE == 5 || MODE == 6;                           // (WMAX then carries the ring's group size, see SU below)
    constexpr int NS = RPT * WMAX;                                   // entry slots of a thread
    // (MODE 4 -- config 5: the values ARE fp32 numbers, and they are STORED as such: 4-byte slots, twice as many in the same LDS -- rows of
    // 9 entries at 8 rows a thread, i.e. the 1M-row finite-volume meshes, resident)
    constexpr bool F32_SLOTS = MODE == 4;
    constexpr int kLdsCap = F32_SLOTS ? 2 * kChipLdsSlots : kChipLdsSlots;
    constexpr int NLDS = NS < kLdsCap ? NS : kLdsCap;                // ... whose values live in LDS
    constexpr int NREG = NS - NLDS;                                  // ... and in registers (the first NREG slots)
    extern __shared__ __attribute__((aligned(16))) double chip_lv[];   // [NLDS][512]: slot s of thread t at [(s - NREG) * 512 + t] (fp32 slots: floats)
    float *const chip_lf = reinterpret_cast<float *>(chip_lv);
    __shared__ double sh[2 * 16];          // block sums: two halves used in turn, 2 x 8 wave sums each
    __shared__ double s_res[2][2];         // the reduced pair, two sets in turn
    __shared__ int s_flag;
    const int t = threadIdx.x;
    const int v = ((int)blockIdx.x & 7) * (kChipWGs / 8) + ((int)blockIdx.x >> 3);
    const int row0 = v * d.per + t;        // row of slot k: row0 + 512 k
    const int grp = (int)blockIdx.x & 7, rank = (int)blockIdx.x >> 3;       // v = 32 grp + rank
    const int glo = grp * (kChipWGs / 8) * d.per;                           // the group's rows: [glo, ghi)
    const int ghi = (glo + (kChipWGs / 8) * d.per < d.n) ? glo + (kChipWGs / 8) * d.per : d.n;
    // the granules: a plainly stored copy (read inside the group; group g's part shifted by 128 g bytes so that no line belongs to
    // two groups) and a written-through copy behind it (see the header)
    const int remote_base = (d.n + kChipZpPad) * 16;
    const __amdgpu_buffer_rsrc_t zp_rs = chip_rsrc(d.zp, 2u * (unsigned)(d.n + kChipZpPad) * 16u);
    const __amdgpu_buffer_rsrc_t part_rs = chip_rsrc(d.part, (unsigned)kChipSlotBytes);

    // ---- the matrix slice and the vectors of the own rows: read once ------------------------------------------------------
    double vr[NREG > 0 ? NREG : 1];
    unsigned dl[(NS + 1) / 2];
    constexpr int LB = WMAX > 7 ? 8 : 4;    // bits per row in `lens`: the top one = the row exists, below it the length
    constexpr unsigned LV = 1u << (LB - 1), LM = LV - 1u;
    static_assert(LB * RPT <= 64, "row lengths of a thread in one or two registers");
    typedef typename std::conditional<(LB * RPT > 32), unsigned long long, unsigned>::type lens_t;
    lens_t lens = 0;
    double x[RPT], r[RPT], p[RPT], q[RPT], dv[JAC ? RPT : 1];
    double bb_loc = 0.0;
    // (unconditional loads from clamped addresses -- a predicated load is a branch with a wait of its own, and 56 of them in a row
    // were 56 dependent round trips: the extents of all rows first, then one row's entries at a time, all of them in flight together)
    int rs_k[RPT], len_k[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int loc = k * kChipThreads + t, i = row0 + k * kChipThreads;
        const bool valid = loc < d.per && i < d.n;
        const int ic = valid ? i : 0;
        const int rs = d.rp[ic], re = d.rp[ic + 1];
        const double bi = d.b[ic];
        const double xi = d.x0 ? d.x0[ic] : 0.0;
        const double di = JAC ? d.dinv[ic] : 1.0;
        rs_k[k] = rs;
        len_k[k] = valid ? re - rs : 0;
        lens |= (lens_t)(valid ? (LV | (STREAM ? 0u : (unsigned)(re - rs))) : 0u) << (LB * k);
        x[k] = valid ? xi : 0.0;
        r[k] = valid ? bi : 0.0;
        p[k] = q[k] = 0.0;
        if (JAC) dv[k] = valid ? di : 1.0;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if ((unsigned)(lens >> (LB * k)) & LV) bb_loc += r[k] * r[k];
        if (STREAM) continue;
        const int i = row0 + k * kChipThreads;
        int cj[WMAX];
        double aj[WMAX];
#pragma unroll
        for (int j = 0; j < WMAX; ++j) {
            const int e = j < len_k[k] ? rs_k[k] + j : 0;          // (entry 0 exists: nnz >= n >= 1)
            cj[j] = d.ci[e];
            aj[j] = d.val[e];
        }
#pragma unroll
        for (int j = 0; j < WMAX; ++j) {
            const int s = k * WMAX + j;
            const bool on = j < len_k[k];
            const int c = on ? cj[j] : i;
            const double a = on ? (F32 ? (double)(float)aj[j] : aj[j]) : 0.0;
            const unsigned del = (unsigned)(c - i + 32768) & 0xffffu;
            if (s & 1) dl[s >> 1] |= del << 16;
            else dl[s >> 1] = del;
            if (s < NREG) vr[s < NREG ? s : 0] = a;
            else if (F32_SLOTS) chip_lf[(s - NREG) * kChipThreads + t] = (float)a;
            else chip_lv[(s - NREG) * kChipThreads + t] = a;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    bool local = false;                     // every group on one XCD (established below, once per solve)
    auto row_on = [&](int k) -> bool { return ((unsigned)(lens >> (LB * k)) & LV) != 0; };

    // q = A p_{k} for the own rows; the gathered entries of p_k are recomputed from the published granules {z_k, p_{k-1}}.
    // All WMAX gathers of a row are in flight together, the next row's are issued while this row's are consumed (the compiler
    // schedules the unrolled loop); row sums in CSR order.
    auto spmv = [&](double beta) {
        u32x4 g[2][WMAX];
        // the LDS-resident values never change, so the optimiser would hoist their loads out of the update loop -- into registers the
        // kernel does not have.  An opaque copy of the thread's column index per call keeps the loads where they are.
        int tl = t;
        asm volatile("" : "+v"(tl));
        const double *lvt = chip_lv + tl;
        const float *lft = chip_lf + tl;
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        const int local_shift = grp * 128;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));            // (and the 56 `own` lane masks)
        // (likewise the gather addresses: they are loop-invariant too, and 56 hoisted addresses are 56 registers)
#pragma unroll
        for (int e = 0; e < (NS + 1) / 2; ++e) asm volatile("" : "+v"(dl[e]));
        asm volatile("" : "+v"(lens));       // (and the 56 `j < len` lane masks)
        auto request = [&](int k, u32x4 (&gk)[WMAX]) {
            const int rowk = row0 + k * kChipThreads;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int s = k * WMAX + j;
                const int del = (int)((dl[s >> 1] >> (16 * (s & 1))) & 0xffffu);
                const int c = rowk + del - 32768;
                const bool own = (unsigned)(c - glo_l) < (unsigned)span_l;              // the column's owner is in this group: the plain copy
                gk[j] = __builtin_amdgcn_raw_buffer_load_b128(zp_rs, MODE == 3 ? (int)0x7ffffff0 : c * 16 + (own ? local_shift : remote_base), 0, kSc1);
            }
        };
        request(0, g[0]);
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            if (k + 1 < RPT) request(k + 1, g[(k + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);                     // (keeps the scheduler from hoisting every row's gathers to the top:
            const int len = (int)((unsigned)(lens >> (LB * k)) & LM);        //  two rows' granules in flight is what the registers hold)
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < WMAX; ++j) {
                const int s = k * WMAX + j;
                const double a = s < NREG ? vr[s < NREG ? s : 0] : (F32_SLOTS ? (double)lft[(s - NREG) * kChipThreads] : lvt[(s - NREG) * kChipThreads]);
                const double pc64 = lo_f64(g[k & 1][j]) + beta * hi_f64(g[k & 1][j]);  // = p_k[c], cg.py:83
                const double pc = F32 ? (double)(float)pc64 : pc64;
                if (j < len) acc += a * pc;
            }
            q[k] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- MODE 5: the streamed product, wave by wave: the 64 rows a wave owns in a slot are 64 CONSECUTIVE rows, their entries one
    // contiguous run of col / val.  Lane l takes entry e0 + l, gathers the granule of its column, parks value x p in the WAVE's part
    // of the LDS, and then adds its own row's products in column order.  No workgroup barrier: the eight waves run their slots
    // independently, one waits for memory while another adds.  LDS: products [8 waves][d.stream_cap], then per thread and slot the
    // row's offset into its wave's run and its length.
    double *const st_prod = chip_lv + (t >> 6) * (STREAM ? d.stream_cap : 0);
    int *const st_rs = reinterpret_cast<int *>(chip_lv + (STREAM ? (kChipThreads / 64) * d.stream_cap : 0));
    int *const st_len = st_rs + RPT * kChipThreads;
    int *const st_run = st_len + RPT * kChipThreads + (t >> 6) * 2 * (RPT + 1);      // this wave's run of slot k: [st_run[2k], st_run[2k + 1])
    const __amdgpu_buffer_rsrc_t ci_rs = chip_rsrc(const_cast<int32_t *>(d.ci), STREAM ? (unsigned)d.rp_nnz * 4u : 4u);
    const __amdgpu_buffer_rsrc_t val_rs = chip_rsrc(const_cast<double *>(d.val), STREAM ? (unsigned)d.rp_nnz * 8u : 8u);
    if (STREAM) {
        const int wg_lo = v * d.per, wg_hi = (wg_lo + d.per < d.n) ? wg_lo + d.per : d.n;
        const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int f0 = wg_lo + k * kChipThreads + wv * 64;
            const int first = f0 < wg_hi ? f0 : wg_hi, last = first + 64 < wg_hi ? first + 64 : wg_hi;
            const int q0 = d.rp[first], q1 = d.rp[last];
            if ((t & 63) == 0) { st_run[2 * k] = q0; st_run[2 * k + 1] = q1; }
            st_rs[k * kChipThreads + t] = len_k[k] > 0 ? rs_k[k] - q0 : 0;
            st_len[k * kChipThreads + t] = len_k[k];
        }
        if ((t & 63) == 0) { st_run[2 * RPT] = 0; st_run[2 * RPT + 1] = 0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // A wave's groups of 64 SU entries pass through a ring: the col / val of a group are fetched SNF iterations before its granules are
    // gathered, the granules SNG iterations before its products are parked; the first SNF groups of the NEXT update are fetched when this
    // one's product ends (the matrix does not change) and ride through the exchanges in registers.  The list of a wave's groups is the
    // same every update: it is written to LDS once (st_grp: slot | last group of its run << 8, first entry, the run's bounds).
    // (the geometry of the ring by measurement on the 1M-row meshes, us per update for SU / SNF = 3/2, 4/1, 4/2, 5/1, 6/1: quadtree in region
    // order 23.4 23.3 24.3 23.3 21.7, in RCM order 21.9 21.9 22.9 22.3 20.6, Delaunay 28.3 25.6 26.1 26.4 27.6 -- a run of 64 short rows
    // in ONE group of 384, longer rows in groups of 256; the launcher passes SU as WMAX)
    constexpr int SU = STREAM ? WMAX : 3, SNG = 1, SNF = 1, SND = SNG + SNF;
    int sc[SND + 1][SU];
    double sa[SND + 1][SU];
    int4 *const st_grp = reinterpret_cast<int4 *>(st_run + 2 * (RPT + 1) * (kChipThreads / 64 - (t >> 6))) + (t >> 6) * kChipStreamGroups;
    int n_grp = 0;
    if (STREAM) {
        if ((t & 63) == 0) {                                      // (once per solve: a few dozen groups a wave)
            int g = 0;
            for (int k = 0; k < RPT; ++k) {
                const int s0 = st_run[2 * k], s1 = st_run[2 * k + 1];
                for (int e = s0; e < s1 && g < kChipStreamGroups; e += SU * 64) {
                    const int last = e + SU * 64 >= s1 ? 1 : 0;
                    st_grp[g++] = make_int4(k | last << 8, e, s0, s1);
                }
            }
            st_run[2 * RPT] = g;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n_grp = __builtin_amdgcn_readfirstlane(st_run[2 * RPT]);
    }
    auto st_desc = [&](int g) -> int4 {                           // group g of this wave (beyond the last: an empty one)
        int4 dd = make_int4(RPT, 0, 0, 0);
        if (g < n_grp) dd = st_grp[g];
        return make_int4(__builtin_amdgcn_readfirstlane(dd.x), __builtin_amdgcn_readfirstlane(dd.y), __builtin_amdgcn_readfirstlane(dd.z),
                         __builtin_amdgcn_readfirstlane(dd.w));
    };
    auto st_fetch = [&](const int4 dd, int (&cc)[SU], double (&aa)[SU]) {
        const int eb = dd.y, s0 = dd.z, s1 = dd.w, lane = t & 63;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (eb + u * 64 >= s1) continue;                      // (the whole chunk lies beyond the run -- or there is no group: the same for every lane)
            const int e = eb + u * 64 + lane;
            const int ec = e < s1 ? e : s0;
            cc[u] = __builtin_amdgcn_raw_buffer_load_b32(ci_rs, ec * 4, 0, 0);
            const u32x2 av2 = __builtin_amdgcn_raw_buffer_load_b64(val_rs, ec * 8, 0, 0);
            aa[u] = __longlong_as_double((long long)(((unsigned long long)av2.y << 32) | av2.x));
        }
    };
    auto st_prefetch = [&]() {                                    // the first SNF groups: before the first product, and behind every product
#pragma unroll
        for (int j = 0; j < SNF; ++j) st_fetch(st_desc(j), sc[j], sa[j]);
    };
    if (STREAM) {
#pragma unroll
        for (int j = 0; j <= SND; ++j)
#pragma unroll
            for (int u = 0; u < SU; ++u) { sc[j][u] = 0; sa[j][u] = 0.0; }
        st_prefetch();
    }
    auto spmv_stream = [&](double beta) {
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        const int local_shift = grp * 128;
        const int lane = t & 63;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));
        u32x4 sg[SNG + 1][SU];
        auto gather = [&](const int (&cc)[SU], u32x4 (&gg)[SU]) {
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const bool own = (unsigned)(cc[u] - glo_l) < (unsigned)span_l;
                gg[u] = __builtin_amdgcn_raw_buffer_load_b128(zp_rs, cc[u] * 16 + (own ? local_shift : remote_base), 0, kSc1);
            }
        };
        // stage j = group i + j; stages 0 .. SNF - 1 arrive fetched
        int4 dg[SND + 1];
#pragma unroll
        for (int j = 0; j <= SND; ++j) dg[j] = st_desc(j);
#pragma unroll
        for (int j = SNF; j < SND; ++j) st_fetch(dg[j], sc[j], sa[j]);
#pragma unroll
        for (int j = 0; j < SNG; ++j)
            if (j < n_grp) gather(sc[j], sg[j]);
        for (int i = 0; i < n_grp; ++i) {
            const int4 dnew = st_desc(i + SND + 1);               // (asked for now, wanted when the ring turns)
            st_fetch(dg[SND], sc[SND], sa[SND]);
            if (i + SNG < n_grp) gather(sc[SNG], sg[SNG]);
            const int k0 = dg[0].x & 0xff, e0 = dg[0].y, s0 = dg[0].z, s1 = dg[0].w;
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int e = e0 + u * 64 + lane;
                const double pc64 = lo_f64(sg[0][u]) + beta * hi_f64(sg[0][u]);     // = p_k[c], cg.py:83
                const double pc = F32 ? (double)(float)pc64 : pc64, av = F32 ? (double)(float)sa[0][u] : sa[0][u];
                if (e < s1) st_prod[e - s0] = av * pc;
            }
            if (dg[0].x >> 8) {                                   // the run is complete: every lane adds its row, in column order
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int rs = st_rs[k0 * kChipThreads + t], len = st_len[k0 * kChipThreads + t];
                double pv[8];                                     // (eight reads in flight, then the sum in order; longer rows go on one by one)
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = st_prod[rs + (j < len ? j : 0)];
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = j < len ? acc + pv[j] : acc;
                for (int j = 8; j < len; ++j) acc += st_prod[rs + j];
#pragma unroll
                for (int kk = 0; kk < RPT; ++kk)
                    if (kk == k0) q[kk] = acc;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();                  // (the next run's products overwrite these)
            }
            // the ring turns
#pragma unroll
            for (int j = 0; j < SND; ++j) {
                dg[j] = dg[j + 1];
#pragma unroll
                for (int u = 0; u < SU; ++u) { sc[j][u] = sc[j + 1][u]; sa[j][u] = sa[j + 1][u]; }
            }
#pragma unroll
            for (int j = 0; j < SNG; ++j)
#pragma unroll
                for (int u = 0; u < SU; ++u) sg[j][u] = sg[j + 1][u];
            dg[SND] = dnew;
        }
        st_prefetch();                                            // (for the next product: in flight through the exchanges)
    };

    // Two chip-wide sums at once, and a chip barrier in the same breath (see the header).  `publish`: the workgroup's granule stores
    // must be visible to whoever passes this point, so every wave drains them first.  Every workgroup returns the same bits.
    // DPCG_CHIP_TRACE: where an update's time goes -- ticks (100 MHz) of thread 0 of EVERY workgroup, accumulated in LDS (registers
    // are what this kernel does not have): [0] q = A p, [1] sum <p,Ap>, [2] vector update + publish, [3] sum <r,z>, [4] the loop,
    // [5] / [6] of [1] / [3] waiting for slots, [7] last stamp
    __shared__ unsigned long long s_tk[8];
    const bool timed = TRACE && d.dbg != nullptr && t == 0;
    if (TRACE && t < 8) s_tk[t] = 0;
    auto stamp = [&](int idx) {             // adds the time since the previous stamp to phase idx (idx < 0: only restarts the clock)
        if (timed) {
            const unsigned long long now = wall_clock64();
            if (idx >= 0) s_tk[idx] += now - s_tk[7];
            s_tk[7] = now;
        }
    };
    Exchange X;
    X.part_rs = part_rs; X.v = v; X.grp = grp; X.rank = rank; X.sh = sh; X.s_res = s_res; X.s_flag = &s_flag; X.err = d.err;
    X.wait_acc = timed ? &s_tk[5] : nullptr;
    auto chip_sum2 = [&](double a, double b2, bool publish, double &ra, double &rb) -> bool { return exchange2(X, a, b2, publish, ra, rb); };
    unsigned far_rows = 0xffu;              // bit k: row k of this thread is gathered by another group (all of them until `local` holds)
    int row0_l = row0;                      // an opaque copy per update: 16 hoisted store addresses are 16 registers the loop does not have
    auto publish = [&](int k, double zk, double pk) {
        const int o = (row0_l + k * kChipThreads) * 16;
        if (local) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + grp * 128, 0, 0);
        if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + remote_base, 0, kSc1);
    };

    bool alive = true;
    double dummy = 0.0, dummy2 = 0.0;
    // ---- where the groups sit: every workgroup reports its XCD, everybody reads the 256 answers ---------------------------
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
    if (d.x0) {                                                   // r = b - A x0 (cg.py:60): x0 published as "z", beta = 0
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) publish(k, x[k], 0.0);
        alive = chip_sum2(0.0, 0.0, true, dummy, dummy2);
        if (alive) {
            if (STREAM) spmv_stream(0.0);
            else spmv(0.0);
#pragma unroll
            for (int k = 0; k < RPT; ++k) r[k] = r[k] - q[k];
            alive = chip_sum2(0.0, 0.0, false, dummy, dummy2);    // everybody has read x0 out of the granules before z_0 overwrites them
        }
    }
    double rz_loc = 0.0, t0_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const double zk = JAC ? dv[JAC ? k : 0] * r[k] : r[k];    // cg.py:61
        p[k] = zk;                                                // cg.py:62
        if (row_on(k)) {
            rz_loc += r[k] * zk;
            t0_loc += d.init_check_r ? r[k] * r[k] : zk * zk;     // cg.py:66: the first test is on z
            publish(k, zk, 0.0);                                  // p_0 = z_0 + 0 * p_{-1} with p_{-1} = 0
        }
    }
    double bb = 0.0, rz = 0.0, tt = 0.0;
    if (alive) alive = chip_sum2(bb_loc, rz_loc, true, bb, rz);    // z_0 and p_{-1} = 0 are published behind this point
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
    // ---- cg.py:70-87: two chip barriers per update -----------------------------------------------------------------------
    if (TRACE) {
        __syncthreads();
        if (t < 8) s_tk[t] = 0;
        __syncthreads();
    }
    const unsigned long long tk_start = timed ? wall_clock64() : 0;
    stamp(-1);
    while (alive && !stop && k_done < d.max_iter) {
        if (MODE == 2) {                                          // (development: what an update costs WITHOUT the gathers)
#pragma unroll
            for (int k = 0; k < RPT; ++k) q[k] = p[k];
        } else {
            if (STREAM) spmv_stream(beta);
            else spmv(beta);                                      // cg.py:75
        }
        double pq_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) pq_loc += q[k] * p[k];
        double pq = 0.0;
        stamp(0);
        if (timed) X.wait_acc = &s_tk[5];
        if (!(alive = chip_sum2(pq_loc, 0.0, false, pq, dummy))) break;         // barrier A: every SpMV of this update is done
        stamp(1);
        if (timed) X.wait_acc = &s_tk[6];
        const double alpha = rz / pq;                             // cg.py:78
        double rz_new_loc = 0.0, rr_loc = 0.0;
        asm volatile("" : "+v"(row0_l), "+v"(far_rows));
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            x[k] = x[k] + alpha * p[k];                           // cg.py:79
            r[k] = r[k] - alpha * q[k];                           // cg.py:80
            const double zk = JAC ? dv[JAC ? k : 0] * r[k] : r[k];   // cg.py:81
            if (row_on(k)) {
                rz_new_loc += r[k] * zk;
                rr_loc += r[k] * r[k];
                publish(k, zk, p[k]);                             // z_{k+1} and p_k for the next update's gathers
            }
        }
        double rz_new = 0.0, rr = 0.0;
        stamp(2);
        if (!(alive = chip_sum2(rz_new_loc, rr_loc, true, rz_new, rr))) break;  // barrier B: granules published, <r,z>, <r,r> known
        stamp(3);
        beta = rz_new / rz;                                       // cg.py:82
#pragma unroll
        for (int k = 0; k < RPT; ++k) p[k] = (JAC ? dv[JAC ? k : 0] * r[k] : r[k]) + beta * p[k];   // cg.py:83 (z recomputed: the same product, the same bits)
        rz = rz_new;
        res = rr / bb;                                            // cg.py:86
        ++k_done;
        if (v == 0 && t == 0 && k_done < d.hist_cap) d.hist[k_done] = res;
        const bool conv = (res < d.rtol_sq) || (rr < d.atol_sq);  // cg.py:71, tested before the next update's work
        if (MODE == 2 || MODE == 3) continue;                                 // (development: a fixed number of updates whatever the numbers do)
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
    // (after a wait that ran out the workgroups are not at the same update: nothing is stored -- d.x may be the caller's buffer, and the
    // caller may have passed it as x0 too, which the multi-launch path that takes over must find untouched)
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (alive && row_on(k)) d.x[row0 + k * kChipThreads] = x[k];
    if (timed) {                            // eight words per workgroup: [4] the loop, [7] (workgroup 0) `local` in bit 0
        unsigned long long *o = d.dbg + 8 * v;
        o[0] = s_tk[0]; o[1] = s_tk[1]; o[2] = s_tk[2]; o[3] = s_tk[3]; o[4] = wall_clock64() - tk_start;
        o[5] = s_tk[5]; o[6] = s_tk[6]; o[7] = local ? 1ull : 0ull;
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

// largest |col - row| and longest row of a CSR pattern (what decides whether the chip kernel can hold it)
__global__ __launch_bounds__(kBlock) void k_band_and_len(int64_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                         int *out /* [0] band, [1] row length */) {
    int band = 0, len = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const int s = rp[i], e = rp[i + 1];
        len = e - s > len ? e - s : len;
        for (int k = s; k < e; ++k) {      // every entry: nothing here may assume that a caller's columns ascend within a row
            const int dlt = ci[k] - (int)i;
            const int ad = dlt < 0 ? -dlt : dlt;
            band = ad > band ? ad : band;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int ob = __shfl_down(band, off), ol = __shfl_down(len, off);
        band = ob > band ? ob : band;
        len = ol > len ? ol : len;
    }
    // one pair of atomics per WORKGROUP (atomics on one word are served one after another, ~12 ns each: a pair per wave of a
    // 1M-row matrix was 0.19 ms)
    __shared__ int s_band[kBlock / 64], s_len[kBlock / 64];
    if ((threadIdx.x & 63) == 0) {
        s_band[threadIdx.x >> 6] = band;
        s_len[threadIdx.x >> 6] = len;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            band = s_band[w] > band ? s_band[w] : band;
            len = s_len[w] > len ? s_len[w] : len;
        }
        atomicMax(&out[0], band);
        atomicMax(&out[1], len);
    }
}

template <int RPT, int WMAX, bool JAC, int MODE>
int chip_launch(const ChipDesc &d, hipStream_t s, bool check_only) {
    constexpr int NS = RPT * WMAX;
    constexpr int kLdsCap = MODE == 4 ? 2 * kChipLdsSlots : kChipLdsSlots;        // (MODE 4: 4-byte value slots)
    constexpr int NLDS = NS < kLdsCap ? NS : kLdsCap;
    // (MODE 5: the product buffer of one slot, the rows' offsets and lengths, the bounds of the runs; the attribute and the occupancy
    // are those of the largest buffer the form admits)
    const int lds_max5 = chip_stream_max_row_len() * kChipThreads * (int)sizeof(double) + RPT * kChipThreads * 8 + 1024 + (kChipThreads / 64) * kChipStreamGroups * 16;
    const int lds = (MODE == 5 || MODE == 6) ? (kChipThreads / 64) * d.stream_cap * (int)sizeof(double) + RPT * kChipThreads * 8 + 1024 + (kChipThreads / 64) * kChipStreamGroups * 16 : NLDS * kChipThreads * (MODE == 4 ? (int)sizeof(float) : (int)sizeof(double));
    const int lds_attr = (MODE == 5 || MODE == 6) ? lds_max5 : lds;
    static int resident = -1;              // workgroups the occupancy query admits per CU (once per instantiation)
    if (resident < 0) {
        if (hipFuncSetAttribute((const void *)k_pcg_chip<RPT, WMAX, JAC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr) != hipSuccess)
            return DPCG_ERR_HIP;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_pcg_chip<RPT, WMAX, JAC, MODE>, kChipThreads, (size_t)lds_attr) !=
            hipSuccess)
            return DPCG_ERR_HIP;
        resident = per_cu;
    }
    if (resident < 1) return DPCG_ERR_STATE;       // the kernel does not fit a CU: refused up front
    if (check_only) return DPCG_OK;
    hipLaunchKernelGGL((k_pcg_chip<RPT, WMAX, JAC, MODE>), dim3(kChipWGs), dim3(kChipThreads), (size_t)lds, s, d);
    return DPCG_OK;
}

}  // namespace

// Test hook (dpcg_debug_occupy): `workgroups` workgroups that each take a whole CU (all of its LDS) and spin for `ticks` of the
// 100 MHz clock -- what a long-running kernel of another stream or process does to the whole-solve kernels' co-residency.
__global__ __launch_bounds__(256) void k_occupy(unsigned long long ticks, int *sink) {
    extern __shared__ int occ_lds[];
    occ_lds[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (sink && occ_lds[(threadIdx.x + 1) & 255] < 0) *sink = 1;
}

int launch_occupy(int workgroups, double ms, hipStream_t s) {
    const int lds = 160 * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DPCG_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_occupy, dim3(workgroups), dim3(256), (size_t)lds, s, (unsigned long long)(ms * 1.0e5), (int *)nullptr);
    return DPCG_OK;
}

// ---- the measured ceiling of the chip kernel's gathers (bench.py: `roofline.frac_of_measured_ceiling`) ---------------------------------
// The whole-chip solve serves the operands of q = A p out of the eight L2s: 16-byte granules, stored plainly by the workgroups of the
// owner's XCD, gathered with sc1 loads, 64 consecutive granules (1 KiB) per wave instruction.  This kernel does THAT and nothing else:
// the same 256 x 512 geometry and placement, the same table per XCD (`per_group` granules of 16 bytes, written plainly by the group's
// own workgroups inside the launch, behind the same exchange), `reps` passes in which every thread gathers RPT x 7 granules at the
// given offsets from its rows (wrapped into its group's part of the table) and xors them into a register.  DEPTH rows' gathers are
// in flight per lane (the solve kernel: 2 -- what its registers hold).  Ticks of the 100 MHz clock per workgroup -> ticks[v].
namespace {
template <int DEPTH>
__global__ __launch_bounds__(kChipThreads) void k_l2_gather_probe(double *table, int per_group, int reps, const int *offs7, double *part, int *err, int *xcc,
                                                                  unsigned long long *ticks, unsigned *sink, int sc1_only) {
    constexpr int RPT = 8, W = 7;
    extern __shared__ __attribute__((aligned(16))) double probe_lds[];     // (unused: the solve kernel's footprint -- one workgroup per CU, the same placement)
    __shared__ double sh[2 * 16];
    __shared__ double s_res[2][2];
    __shared__ int s_flag;
    const int t = threadIdx.x;
    const int grp = (int)blockIdx.x & 7, rank = (int)blockIdx.x >> 3;
    const int v = grp * (kChipWGs / 8) + rank;
    const int per = per_group / (kChipWGs / 8);                 // granules a workgroup writes
    const int glo = grp * per_group;
    const int rglo = sc1_only == 2 ? ((grp + 1) & 7) * per_group : glo;      // (2: gather the part the NEXT group wrote through -- another XCD's)
    const __amdgpu_buffer_rsrc_t rs = chip_rsrc(table, 8u * (unsigned)per_group * 16u);
    Exchange X;
    X.part_rs = chip_rsrc(part, (unsigned)kChipSlotBytes);
    X.v = v; X.grp = grp; X.rank = rank; X.sh = sh; X.s_res = s_res; X.s_flag = &s_flag; X.err = err;
    bool alive = true;
    const bool local = groups_on_one_xcd(X, xcc, alive) && !sc1_only;
    X.local = local;
    for (int loc = t; loc < per; loc += kChipThreads) {
        const int i = glo + rank * per + loc;
        if (local) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2((double)i, 1.0), rs, i * 16, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2((double)i, 1.0), rs, i * 16, 0, kSc1);
    }
    double d0 = 0.0, d1 = 0.0;
    if (alive) alive = exchange2(X, 0.0, 0.0, true, d0, d1);
    int off[W];
#pragma unroll
    for (int j = 0; j < W; ++j) off[j] = offs7[j];
    int row[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) row[k] = (rank * per + t + k * kChipThreads) % per_group;      // position inside the group's part
    unsigned acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int rep = 0; alive && rep < reps; ++rep) {
        u32x4 g[DEPTH][W];
#pragma unroll
        for (int k = 0; k < RPT; ++k) asm volatile("" : "+v"(row[k]));
        auto request = [&](int k, u32x4 (&gk)[W]) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                int c = row[k] + off[j];
                c = c < 0 ? c + per_group : (c >= per_group ? c - per_group : c);
                gk[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (rglo + c) * 16, 0, kSc1);
            }
        };
#pragma unroll
        for (int k = 0; k < DEPTH - 1; ++k) request(k, g[k]);
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            if (k + DEPTH - 1 < RPT) request(k + DEPTH - 1, g[(k + DEPTH - 1) % DEPTH]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < W; ++j) acc ^= g[k % DEPTH][j].x ^ g[k % DEPTH][j].z;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (t == 0) ticks[v] = alive ? t1 - t0 : 0ull;
    if (acc == 0x9e3779b9u) { *sink = acc; probe_lds[t] = 1.0; }
    if (v == 0 && t == 0) xcc[kChipWGs] = local ? 1 : 0;
}

}  // namespace

// reps passes of 256 x 512 x 8 x 7 sixteen-byte gathers; out_ticks: 256 words; xcc: 257 ints ([256] <- the groups sat on one XCD each)
int launch_l2_gather_probe(double *table, int per_group, int reps, const int *offs7_dev, int depth, int sc1_only, double *part, int *err, int *xcc,
                           unsigned long long *ticks, unsigned *sink, hipStream_t s) {
    if (per_group < kChipThreads * (kChipWGs / 8) || per_group % (kChipWGs / 8) != 0 || reps < 1) return DPCG_ERR_INVALID;
    const int lds = kChipLdsSlots * kChipThreads * (int)sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)k_l2_gather_probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DPCG_ERR_HIP;
        if (hipFuncSetAttribute((const void *)k_l2_gather_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DPCG_ERR_HIP;
        attr_set = true;
    }
    if (depth == 4) hipLaunchKernelGGL(k_l2_gather_probe<4>, dim3(kChipWGs), dim3(kChipThreads), (size_t)lds, s, table, per_group, reps, offs7_dev, part, err, xcc, ticks, sink, sc1_only);
    else hipLaunchKernelGGL(k_l2_gather_probe<2>, dim3(kChipWGs), dim3(kChipThreads), (size_t)lds, s, table, per_group, reps, offs7_dev, part, err, xcc, ticks, sink, sc1_only);
    return DPCG_OK;
}

int chip_max_rows() { return kChipWGs * kChipThreads * kChipMaxRpt; }
// rows of up to 9 entries (unstructured meshes) while a thread holds at most four rows (524 288 rows), 7 beyond
// ... with fp32-stored values (DPCG_SPMV_F32: 4-byte slots) rows of 9 entries at any size
int chip_max_row_len(int64_t n, bool f32_slots) { return (f32_slots || n <= (int64_t)kChipWGs * kChipThreads * 4) ? 9 : 7; }
int chip_stream_max_row_len() { return 24; }      // 24 x 512 x 8 B = 98 KB of products + 33 KB of row tables (8 rows a thread) + 8 KB of group lists
int chip_max_band() { return 32767; }
int chip_workgroups() { return kChipWGs; }
int chip_threads() { return kChipThreads; }
int chip_slot_doubles() { return kChipSlotBytes / 8; }
int64_t chip_zp_doubles(int64_t n) { return 4 * (n + kChipZpPad); }

void launch_band_and_len(const CsrDev &A, int *out2_zeroed_dev, hipStream_t s) {
    int64_t g = (A.n + kBlock - 1) / kBlock;
    if (g > 512) g = 512;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_band_and_len, dim3((int)g), dim3(kBlock), 0, s, A.n, A.rowptr, A.col, out2_zeroed_dev);
}

// One system on the whole chip.  max_row_len <= 7; d.per = ceil(n / 256) <= 4096.  check_only: the occupancy query alone.
// Returns DPCG_OK, DPCG_ERR_STATE when the kernel cannot be resident on every CU, or a negative HIP status.
int launch_pcg_chip(const ChipDesc &d, int max_row_len, hipStream_t s, bool check_only) {
    const int rpt = (d.per + kChipThreads - 1) / kChipThreads;
    const bool jac = d.precond == DPCG_PRECOND_JACOBI;
    if (d.stream_cap > 0) {                // the streamed form (MODE 5)
        if (max_row_len < 1 || max_row_len > chip_stream_max_row_len() || d.stream_cap < max_row_len * 64 || d.per < 1 ||
            d.per > kChipThreads * kChipMaxRpt || d.bench || d.dbg || (d.f32 && d.x0))
            return DPCG_ERR_INVALID;
        // entries of a lane per group (the kernel's SU, passed as WMAX): 6 where a run of 64 rows fits ONE group of 384 (<= 5.6 entries a
        // row on average), else 4
        const bool short_rows = (double)d.rp_nnz <= 5.6 * (double)d.n;
#define DPCG_CHIP_S3(RPTV, SUV, MV) (jac ? chip_launch<RPTV, SUV, true, MV>(d, s, check_only) : chip_launch<RPTV, SUV, false, MV>(d, s, check_only))
#define DPCG_CHIP_S2(RPTV, SUV) (d.f32 ? DPCG_CHIP_S3(RPTV, SUV, 6) : DPCG_CHIP_S3(RPTV, SUV, 5))
#define DPCG_CHIP_S(RPTV) (short_rows ? DPCG_CHIP_S2(RPTV, 6) : DPCG_CHIP_S2(RPTV, 4))
        if (rpt <= 2) return DPCG_CHIP_S(2);
        if (rpt <= 4) return DPCG_CHIP_S(4);
        return DPCG_CHIP_S(8);
#undef DPCG_CHIP_S
#undef DPCG_CHIP_S2
#undef DPCG_CHIP_S3
    }
    const int mode = d.bench == 3 ? 3 : (d.bench ? 2 : (d.dbg != nullptr ? 1 : (d.f32 ? 4 : 0)));
    if (max_row_len < 1 || max_row_len > ((rpt <= 4 || mode == 4) ? 9 : 7) || d.per < 1 || d.per > kChipThreads * kChipMaxRpt) return DPCG_ERR_INVALID;
    if (d.f32 && (mode != 4 || d.x0)) return DPCG_ERR_INVALID;
    if (rpt > 4 && max_row_len > 7) return jac ? chip_launch<8, 9, true, 4>(d, s, check_only) : chip_launch<8, 9, false, 4>(d, s, check_only);   // (4-byte slots)
#define DPCG_CHIP_T(RPTV, WV, JV) (mode == 4 ? chip_launch<RPTV, WV, JV, 4>(d, s, check_only) : mode == 3 ? chip_launch<RPTV, WV, JV, 3>(d, s, check_only) : mode == 2 ? chip_launch<RPTV, WV, JV, 2>(d, s, check_only) : (mode == 1 ? chip_launch<RPTV, WV, JV, 1>(d, s, check_only) : chip_launch<RPTV, WV, JV, 0>(d, s, check_only)))
#define DPCG_CHIP_W(RPTV, WV) (jac ? DPCG_CHIP_T(RPTV, WV, true) : DPCG_CHIP_T(RPTV, WV, false))
#define DPCG_CHIP_R(RPTV) (max_row_len <= 5 ? DPCG_CHIP_W(RPTV, 5) : DPCG_CHIP_W(RPTV, 7))
#define DPCG_CHIP_R9(RPTV) (max_row_len <= 5 ? DPCG_CHIP_W(RPTV, 5) : (max_row_len <= 7 ? DPCG_CHIP_W(RPTV, 7) : DPCG_CHIP_W(RPTV, 9)))
    if (rpt <= 2) return DPCG_CHIP_R9(2);
    if (rpt <= 4) return DPCG_CHIP_R9(4);
    return DPCG_CHIP_R(8);
#undef DPCG_CHIP_R
#undef DPCG_CHIP_R9
#undef DPCG_CHIP_W
#undef DPCG_CHIP_T
}

}  // namespace dpcg
