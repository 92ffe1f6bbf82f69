// Whole-solve kernel for M = L L^T MULTIPLIED (z = L (L^T r): the reference's learned technique and its incomplete-Cholesky ones,
// test.py:81-88,100-105, train.py:99-106) on mid-size systems -- BASELINE config 2: 256^2 = 65 536 rows with the CNN-emitted factor
// of 15 entries a row -- hand-written for gfx950 (MI355X), the whole chip as one team like dpcg_chip.hip.
//
// The multi-launch update of that system is five launches of 1-2 us of work each behind 2 us boundaries: 21.3 us per update, 0.2 of the
// roofline (here: 9.2 us).  A team on one XCD (dpcg_team.hip) cannot hold it: L and L^T are 23.5 MB, six times an XCD's L2.  Over 256 CUs they are
// 92 KB each: here A, L^T and L of a workgroup's rows live in LDS / registers for the whole solve (39 eight-byte value slots per
// thread in LDS, the rest in registers; columns as 16-bit offsets from the row), one launch, cg.py:58-90:
//   * 256 workgroups x 512 threads, workgroup v owns rows [v * per, (v + 1) * per), thread t rows v * per + t + 512 k (k < RPT <= 2).
//   * FOUR hand-offs per update: <p,Ap> | r published -> t = L^T r | t published -> z = L t | z, p published and <r,z>, <r,r>.
//     The first and the last are the chip-wide reductions of dpcg_chip.hip (chip::exchange2).  The two in the middle sum nothing:
//     r and t travel as SELF-VALIDATING 16-byte granules {value, value ^ key(solve, generation)} that a gather reads again until
//     its halves differ by the expected key -- the gather is its own synchronisation, no drain, no flag, no barrier (config 2's
//     shape, us per update: chip-wide barriers 10.5, generation words of the neighbouring workgroups 9.2, this 7.7; 14.7 in launches).
//     As in dpcg_chip.hip a row recomputes the entries of p_k it gathers from the published granules {z_k, p_{k-1}}.
//   * visibility, placement and co-residency: exactly as dpcg_chip.hip (written-through copies for the rows other groups gather,
//     plainly stored copies inside a group when every group sits on one XCD; sc1 loads; bounded waits; DPCG_ERR_STATE -> the caller's
//     multi-launch path).
// Row sums of all three products run in CSR order, dot products in the chip tree: history, count and x equal the CPU restatement's
// (form "chip", kind llt_multiply) bit for bit.
#include <algorithm>

#include "dpcg_chip_device.h"

namespace dpcg {

using namespace chip;

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// RPT: rows per thread (1, 2); WA: entry slots per row of A (5, 7); WL: entry slots per row of L and of L^T (4, 8, 16).
// SPLIT (RPT == 1, <= 256 rows per workgroup -- config 2: otherwise half the lanes would idle while the others gather and validate 2 x 16
// granules a row): TWO lanes per row.  Both hold the row's vectors and compute q = A p (redundantly); of a factor row lane h of the
// pair holds, gathers and validates entries [h * WL/2, (h + 1) * WL/2), the products are then added in CSR order -- the even lane's,
// its sum handed to the odd lane, the odd lane's -- so the row sum keeps its bits; dot products count a row once (even lanes).
template <int RPT, int WA, int WL, bool SPLIT = false>
__global__ __launch_bounds__(kChipThreads) void k_pcg_chip_llt(const ChipLltDesc d) {
    static_assert(!SPLIT || (RPT == 1 && WL % 2 == 0), "two lanes per row: one row per pair");
    constexpr int WLS = SPLIT ? WL / 2 : WL;                         // factor-row entries a thread holds
    constexpr int WR = WA + 2 * WLS;                                 // entry slots of a thread's row: A | L^T | L
    constexpr int NS = RPT * WR;
    constexpr int NLDS = NS < kChipLdsSlots ? NS : kChipLdsSlots;
    constexpr int NREG = NS - NLDS;                                  // the first NREG slots live in registers
    extern __shared__ __attribute__((aligned(16))) double chip_lv[];   // [NLDS][512]
    __shared__ double sh[2 * 16];
    __shared__ double s_res[2][2];
    __shared__ int s_flag;
    const int t = threadIdx.x;
    const int tr = SPLIT ? (t >> 1) : t;                             // the thread's row slot in the workgroup
    const int half = SPLIT ? (t & 1) : 0;
    const bool counts = half == 0;                                   // (of a pair, the even lane publishes and counts in the dot products)
    const int v = ((int)blockIdx.x & 7) * (kChipWGs / 8) + ((int)blockIdx.x >> 3);
    const int grp = (int)blockIdx.x & 7, rank = (int)blockIdx.x >> 3;
    const int row0 = v * d.per + tr;
    const int glo = grp * (kChipWGs / 8) * d.per;
    const int ghi = (glo + (kChipWGs / 8) * d.per < d.n) ? glo + (kChipWGs / 8) * d.per : d.n;
    const int zp_remote = (d.n + kChipZpPad) * 16, v8_remote = (d.n + kChipZpPad) * 8;
    const __amdgpu_buffer_rsrc_t zp_rs = chip_rsrc(d.zp, 2u * (unsigned)(d.n + kChipZpPad) * 16u);
    const __amdgpu_buffer_rsrc_t r_rs = chip_rsrc(d.rpub, 2u * (unsigned)(d.n + kChipZpPad) * 16u);     // (16-byte granules; the 8-byte forms use half)
    const __amdgpu_buffer_rsrc_t t_rs = chip_rsrc(d.tpub, 2u * (unsigned)(d.n + kChipZpPad) * 16u);

    // ---- the three matrix slices and the vectors of the own rows: read once --------------------------------------------------
    double vr[NREG > 0 ? NREG : 1];
    unsigned dl[(NS + 1) / 2];
    unsigned lens[RPT];                     // per row: bit 31 = the row exists; length in A (bits 0-4), L^T (5-9), L (10-14)
    double x[RPT], r[RPT], p[RPT], q[RPT];
    double bb_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int loc = k * kChipThreads + tr, i = row0 + k * kChipThreads;
        const bool valid = loc < d.per && i < d.n;
        const int ic = valid ? i : 0;
        const int a0 = d.rp[ic], a1 = d.rp[ic + 1];
        int t0 = d.trp[ic], t1 = d.trp[ic + 1], l0 = d.lrp[ic], l1 = d.lrp[ic + 1];
        if (SPLIT) {                                                  // this lane's half of the factor rows
            t0 = t0 + half * WLS < t1 ? t0 + half * WLS : t1;
            t1 = t0 + WLS < t1 ? t0 + WLS : t1;
            l0 = l0 + half * WLS < l1 ? l0 + half * WLS : l1;
            l1 = l0 + WLS < l1 ? l0 + WLS : l1;
        }
        const double bi = d.b[ic];
        const double xi = d.x0 ? d.x0[ic] : 0.0;
        lens[k] = valid ? (0x80000000u | (unsigned)(a1 - a0) | ((unsigned)(t1 - t0) << 5) | ((unsigned)(l1 - l0) << 10)) : 0u;
        x[k] = valid ? xi : 0.0;
        r[k] = valid ? bi : 0.0;
        p[k] = q[k] = 0.0;
        if (valid && counts) bb_loc += bi * bi;
        auto take = [&](int first, int W, const int32_t *ci, const double *val, int e0, int len) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const int s = k * WR + first + j;
                const bool on = valid && j < len;
                const int e = on ? e0 + j : 0;
                const int c = on ? ci[e] : i;
                const double a = on ? val[e] : 0.0;
                const unsigned del = (unsigned)(c - i + 32768) & 0xffffu;
                if (s & 1) dl[s >> 1] |= del << 16;
                else dl[s >> 1] = del;
                if (s < NREG) vr[s < NREG ? s : 0] = a;
                else chip_lv[(s - NREG) * kChipThreads + t] = a;
            }
        };
        take(0, WA, d.ci, d.val, a0, a1 - a0);
        __builtin_amdgcn_sched_barrier(0);
        take(WA, WLS, d.tci, d.tval, t0, t1 - t0);
        __builtin_amdgcn_sched_barrier(0);
        take(WA + WLS, WLS, d.lci, d.lval, l0, l1 - l0);
        __builtin_amdgcn_sched_barrier(0);
    }
    auto row_on = [&](int k) -> bool { return (lens[k] & 0x80000000u) != 0; };
    auto row_counts = [&](int k) -> bool { return counts && (lens[k] & 0x80000000u) != 0; };   // ... and this lane speaks for it
    bool local = false;

    Exchange X;
    X.part_rs = chip_rsrc(d.part, (unsigned)kChipSlotBytes);
    X.v = v; X.grp = grp; X.rank = rank; X.sh = sh; X.s_res = s_res; X.s_flag = &s_flag; X.err = d.err;
    double dummy = 0.0, dummy2 = 0.0;
    auto data_barrier = [&]() -> bool { return exchange2(X, 0.0, 0.0, true, dummy, dummy2); };
    // y_k = (row of the matrix whose slots start at `first`) . vec, the vector's entries fetched by `fetch(column, is_own_group)`;
    // all W fetches of a row in flight together, sums in CSR order
    unsigned far_rows = (1u << RPT) - 1u;
    auto row_products = [&](int first, int W, int lshift, double (&y)[RPT], auto &&fetch) {
        int tl = t;
        asm volatile("" : "+v"(tl));                      // (keeps the LDS reads and the addresses inside the update loop: see dpcg_chip.hip)
        const double *lvt = chip_lv + tl;
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));
#pragma unroll
        for (int e = 0; e < (NS + 1) / 2; ++e) asm volatile("" : "+v"(dl[e]));
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            asm volatile("" : "+v"(lens[k]));
            const int rowk = row0 + k * kChipThreads;
            const int len = (int)((lens[k] >> lshift) & 31u);
            double g[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < W) {
                    const int s = k * WR + first + j;
                    const int del = (int)((dl[s >> 1] >> (16 * (s & 1))) & 0xffffu);
                    const int c = rowk + del - 32768;
                    g[j] = fetch(c, (unsigned)(c - glo_l) < (unsigned)span_l);
                }
            }
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < W) {
                    const int s = k * WR + first + j;
                    const double a = s < NREG ? vr[s < NREG ? s : 0] : lvt[(s - NREG) * kChipThreads];
                    if (j < len) acc += a * g[j];
                }
            }
            y[k] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto fetch8 = [&](const __amdgpu_buffer_rsrc_t &rs) {
        return [&, rs](int c, bool own) -> double {
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, c * 8 + (own ? grp * 128 : v8_remote), 0, kSc1);
            return __hiloint2double((int)w.y, (int)w.x);
        };
    };
    auto publish8 = [&](const __amdgpu_buffer_rsrc_t &rs, int k, double val) {
        const int o = (row0 + k * kChipThreads) * 8;
        u32x2 w;
        w.x = (unsigned)__double2loint(val);
        w.y = (unsigned)__double2hiint(val);
        if (local) __builtin_amdgcn_raw_buffer_store_b64(w, rs, o + grp * 128, 0, 0);
        if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b64(w, rs, o + v8_remote, 0, kSc1);
    };
    auto publish_zp = [&](int k, double zk, double pk) {
        const int o = (row0 + k * kChipThreads) * 16;
        if (local) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + grp * 128, 0, 0);
        if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(zk, pk), zp_rs, o + zp_remote, 0, kSc1);
    };
    // SELF-VALIDATING granules for r and t = L^T r: {value, value ^ key}, the key a function of the solve and of the publication's
    // generation.  Each half is an aligned 8-byte word (written and read whole); a reader accepts a granule only when its halves
    // differ by exactly the key it expects, so a stale granule (another key), a half-written one (halves of two generations: they
    // differ by the key only if the two values are equal, and then the value is right) and whatever the buffer held before are all
    // refused and read again.  The gather of a product is thereby its own synchronisation: no drain, no flag, no barrier between the
    // three products of an update.  Rewriting is safe: two chip-wide exchanges lie between a vector's gathers and its next publication.
    // (nonce == 0 -- development knob DPCG_CHIP_LLT_SYNC=0 -- keeps plain 8-byte vectors and a chip-wide barrier per product.)
    auto make_key = [&](unsigned gen) -> unsigned long long {
        return (((unsigned long long)d.nonce << 32) | (unsigned long long)gen) * 0x9E3779B97F4A7C15ull | 1ull;
    };
    auto publish_tagged = [&](const __amdgpu_buffer_rsrc_t &rs, int k, double val, unsigned long long key) {
        const int o = (row0 + k * kChipThreads) * 16;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val), tag = bits ^ key;
        u32x4 w;
        w.x = (unsigned)bits; w.y = (unsigned)(bits >> 32); w.z = (unsigned)tag; w.w = (unsigned)(tag >> 32);
        if (local) __builtin_amdgcn_raw_buffer_store_b128(w, rs, o + grp * 128, 0, 0);
        if ((far_rows >> k) & 1u) __builtin_amdgcn_raw_buffer_store_b128(w, rs, o + zp_remote, 0, kSc1);
    };
    auto products_tagged = [&](int first, int W, int lshift, double (&y)[RPT], const __amdgpu_buffer_rsrc_t &rs, unsigned long long key) -> bool {
        int tl = t;
        asm volatile("" : "+v"(tl));
        const double *lvt = chip_lv + tl;
        int glo_l = glo, span_l = local ? ghi - glo : 0;
        asm volatile("" : "+s"(glo_l), "+s"(span_l));
#pragma unroll
        for (int e = 0; e < (NS + 1) / 2; ++e) asm volatile("" : "+v"(dl[e]));
        const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            asm volatile("" : "+v"(lens[k]));
            const int rowk = row0 + k * kChipThreads;
            const int len = (int)((lens[k] >> lshift) & 31u);
            u32x4 g[16];
            int off[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < W) {
                    const int s = k * WR + first + j;
                    const int del = (int)((dl[s >> 1] >> (16 * (s & 1))) & 0xffffu);
                    const int c = rowk + del - 32768;
                    off[j] = c * 16 + ((unsigned)(c - glo_l) < (unsigned)span_l ? grp * 128 : zp_remote);
                    g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], 0, kSc1);
                }
            }
            unsigned spins = 0;
            unsigned long long t0 = 0;
            for (;;) {
                bool bad = false;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < W) bad = bad || (j < len && ((g[j].x ^ g[j].z) != klo || (g[j].y ^ g[j].w) != khi));
                if (__ballot(bad) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                if (bad) {                       // (the whole row again: one divergent region, not one saved exec mask per entry)
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < W) g[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], 0, kSc1);
                }
                if ((++spins & 255u) == 0) {
                    const unsigned long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > kChipSpinTicks || __hip_atomic_load(d.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        atomicExch(d.err, 1);
                        ok = false;
                        break;
                    }
                }
            }
            double acc = 0.0;
            if (SPLIT) {
                double prod[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j < W) {
                        const int s = k * WR + first + j;
                        const double a = s < NREG ? vr[s < NREG ? s : 0] : lvt[(s - NREG) * kChipThreads];
                        prod[j] = a * lo_f64(g[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < W && half == 0 && j < len) acc += prod[j];          // the row's first entries, in order ...
                const double first_half = __shfl(acc, (int)(threadIdx.x & 63u) & ~1);
                if (half == 1) {
                    acc = first_half;
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < W && j < len) acc += prod[j];                   // ... then the rest onto that sum: the CSR order
                }
                acc = __shfl(acc, (int)(threadIdx.x & 63u) | 1);                // both lanes of the pair hold the row's sum
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j < W) {
                        const int s = k * WR + first + j;
                        const double a = s < NREG ? vr[s < NREG ? s : 0] : lvt[(s - NREG) * kChipThreads];
                        if (j < len) acc += a * lo_f64(g[j]);
                    }
                }
            }
            y[k] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
        return __syncthreads_and(ok ? 1 : 0) != 0;       // (a wait that ran out anywhere in the workgroup ends the solve for all of it)
    };
    // q = A p_k, the gathered entries recomputed from the granules {z_k, p_{k-1}} (cg.py:75,83)
    auto spmv_a = [&](double beta) {
        row_products(0, WA, 0, q, [&](int c, bool own) -> double {
            const u32x4 gq = __builtin_amdgcn_raw_buffer_load_b128(zp_rs, c * 16 + (own ? grp * 128 : zp_remote), 0, kSc1);
            return lo_f64(gq) + beta * hi_f64(gq);
        });
    };
    // z = L (L^T r) for the own rows (cg.py:61,81): r published, a chip barrier, t = L^T r published, a chip barrier, z = L t
    double z[RPT];
    unsigned pub_gen = 0;
    auto apply_m = [&]() -> bool {
        if (d.nonce) {                                            // self-validating granules: the gathers synchronise
            const unsigned long long key_r = make_key(++pub_gen);
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (row_on(k) && counts) publish_tagged(r_rs, k, r[k], key_r);
            double tv[RPT];
            if (!products_tagged(WA, WLS, 5, tv, r_rs, key_r)) return false;
            const unsigned long long key_t = make_key(++pub_gen);
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (row_on(k) && counts) publish_tagged(t_rs, k, tv[k], key_t);
            return products_tagged(WA + WLS, WLS, 10, z, t_rs, key_t);
        }
        if (SPLIT) return false;                                  // (two lanes per row: the self-validating form only; the launcher sees to it)
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) publish8(r_rs, k, r[k]);
        if (!data_barrier()) return false;
        double tv[RPT];
        row_products(WA, WLS, 5, tv, fetch8(r_rs));
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_on(k)) publish8(t_rs, k, tv[k]);
        if (!data_barrier()) return false;
        row_products(WA + WLS, WLS, 10, z, fetch8(t_rs));
        return true;
    };

    bool alive = true;
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
            if (row_counts(k)) publish_zp(k, x[k], 0.0);
        alive = data_barrier();
        if (alive) {
            spmv_a(0.0);
#pragma unroll
            for (int k = 0; k < RPT; ++k) r[k] = r[k] - q[k];
            alive = exchange2(X, 0.0, 0.0, false, dummy, dummy2);  // everybody has read x0 out of the granules before z_0 overwrites them
        }
    }
    if (alive) alive = apply_m();                                 // cg.py:61
    double rz_loc = 0.0, t0_loc = 0.0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        p[k] = z[k];                                              // cg.py:62
        if (row_counts(k)) {
            rz_loc += r[k] * z[k];
            t0_loc += d.init_check_r ? r[k] * r[k] : z[k] * z[k];  // cg.py:66: the first test is on z
            publish_zp(k, z[k], 0.0);                             // p_0 = z_0 + 0 * p_{-1}
        }
    }
    double bb = 0.0, rz = 0.0, tt = 0.0;
    if (alive) alive = exchange2(X, bb_loc, rz_loc, true, bb, rz);
    if (alive) alive = exchange2(X, t0_loc, 0.0, false, tt, dummy);
    double res = tt / bb, beta = 0.0;
    int k_done = 0, status = DPCG_MAX_ITER;
    bool stop = false;
    if (alive) {
        if (v == 0 && t == 0 && d.hist_cap > 0) d.hist[0] = res;
        const bool conv = (res < d.rtol_sq) || (tt < d.atol_sq);
        if (conv) { stop = true; status = DPCG_OK; }
        else if (!(res == res)) { stop = true; status = DPCG_BREAKDOWN; }
    }
    // ---- cg.py:70-87: four chip synchronisations per update --------------------------------------------------------------
    while (alive && !stop && k_done < d.max_iter) {
        spmv_a(beta);                                             // cg.py:75
        double pq_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_counts(k)) pq_loc += q[k] * p[k];
        double pq = 0.0;
        if (!(alive = exchange2(X, pq_loc, 0.0, false, pq, dummy))) break;       // every SpMV of this update is done
        const double alpha = rz / pq;                             // cg.py:78
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            x[k] = x[k] + alpha * p[k];                           // cg.py:79
            r[k] = r[k] - alpha * q[k];                           // cg.py:80
        }
        if (!(alive = apply_m())) break;                          // cg.py:81
        double rz_new_loc = 0.0, rr_loc = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (row_counts(k)) {
                rz_new_loc += r[k] * z[k];
                rr_loc += r[k] * r[k];
                publish_zp(k, z[k], p[k]);                        // z_{k+1} and p_k for the next update's gathers
            }
        double rz_new = 0.0, rr = 0.0;
        if (!(alive = exchange2(X, rz_new_loc, rr_loc, true, rz_new, rr))) break;
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
        if (alive && row_counts(k)) d.x[row0 + k * kChipThreads] = x[k];     // (nothing after a wait that ran out: see dpcg_chip.hip)
    if (v == 0 && t == 0) {
        Scalars *sc = d.out;
        sc->k = k_done;
        sc->res = res;
        sc->bb = bb;
        sc->status = alive ? status : DPCG_ERR_STATE;
        sc->done = 1;
    }
}

template <int RPT, int WA, int WL, bool SPLIT>
int chip_llt_launch(const ChipLltDesc &d, hipStream_t s, bool check_only) {
    constexpr int NS = RPT * (WA + 2 * (SPLIT ? WL / 2 : WL));
    constexpr int NLDS = NS < kChipLdsSlots ? NS : kChipLdsSlots;
    const int lds = NLDS * kChipThreads * (int)sizeof(double);
    static int resident = -1;
    if (resident < 0) {
        if (hipFuncSetAttribute((const void *)k_pcg_chip_llt<RPT, WA, WL, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return DPCG_ERR_HIP;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_pcg_chip_llt<RPT, WA, WL, SPLIT>, kChipThreads, (size_t)lds) !=
            hipSuccess)
            return DPCG_ERR_HIP;
        resident = per_cu;
    }
    if (resident < 1) return DPCG_ERR_STATE;
    if (check_only) return DPCG_OK;
    hipLaunchKernelGGL((k_pcg_chip_llt<RPT, WA, WL, SPLIT>), dim3(kChipWGs), dim3(kChipThreads), (size_t)lds, s, d);
    return DPCG_OK;
}

}  // namespace

int chip_llt_max_rows() { return kChipWGs * kChipThreads * 2; }
int chip_llt_max_row_len() { return 16; }

// max_a: longest row of A (<= 7); max_l: longest row of L or of L^T (<= 16).  Returns DPCG_OK, DPCG_ERR_STATE when the kernel cannot be
// resident on every CU, or a negative status.
int launch_pcg_chip_llt(const ChipLltDesc &d, int max_a, int max_l, hipStream_t s, bool check_only) {
    if (max_a < 1 || max_a > 7 || max_l < 1 || max_l > 16 || d.per < 1 || d.per > 2 * kChipThreads) return DPCG_ERR_INVALID;
    const int rpt = (d.per + kChipThreads - 1) / kChipThreads;
    if (rpt > 1 && max_l > 8) return DPCG_ERR_STATE;      // (not compiled: two rows a thread of 16-entry factor rows spill)
    // 16-entry factor rows on <= 256 rows per workgroup (config 2): two lanes per row -- the self-validating form only
    const bool split = max_l > 8 && d.per <= kChipThreads / 2 && d.nonce != 0;
#define DPCG_LLT_L1(WAV) (max_l <= 4 ? chip_llt_launch<1, WAV, 4, false>(d, s, check_only) : (max_l <= 8 ? chip_llt_launch<1, WAV, 8, false>(d, s, check_only) : (split ? chip_llt_launch<1, WAV, 16, true>(d, s, check_only) : chip_llt_launch<1, WAV, 16, false>(d, s, check_only))))
#define DPCG_LLT_L2(WAV) (max_l <= 4 ? chip_llt_launch<2, WAV, 4, false>(d, s, check_only) : chip_llt_launch<2, WAV, 8, false>(d, s, check_only))
    return rpt <= 1 ? (max_a <= 5 ? DPCG_LLT_L1(5) : DPCG_LLT_L1(7)) : (max_a <= 5 ? DPCG_LLT_L2(5) : DPCG_LLT_L2(7));
#undef DPCG_LLT_L1
#undef DPCG_LLT_L2
}

}  // namespace dpcg
