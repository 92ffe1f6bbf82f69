// Device-side pieces shared by the whole-chip solve kernels (dpcg_chip.hip: M = I / Jacobi; dpcg_chip_llt.hip: M = L L^T multiplied):
// geometry, the 16-byte granule helpers, and the two-hop chip-wide exchange that is both their reduction and their barrier.
// Include from .hip files only.
#pragma once

#include "dpcg_device.h"

namespace dpcg {
namespace chip {

constexpr int kChipWGs = 256;           // one workgroup per CU
constexpr int kChipThreads = 512;       // 8 waves: two per SIMD, 256 vector registers per lane
constexpr int kChipMaxRpt = 8;          // rows per thread: n <= 256 * 512 * 8
constexpr int kChipLdsSlots = 39;       // 8-byte value slots per thread kept in LDS: 39 * 512 * 8 = 159 744 B of the CU's 163 840
constexpr unsigned long long kChipSpinTicks = 2000000ull;               // 20 ms of the 100 MHz constant clock
constexpr unsigned long long kChipPending = 0x7ff8dead0badbeefULL;      // a quiet NaN that no arithmetic here produces
constexpr int kChipZpPad = 4096;        // granules of slack behind each copy (group shifts; rows that do not exist gather there)
constexpr int kChipS1Bytes = 4 * kChipWGs * 16;                // group-level slots: 4 sets x 256 workgroups x 16 B
constexpr int kChipSlotBytes = kChipS1Bytes + 4 * 8 * 8 * 16;  // + chip-level slots: 4 sets x 8 destination groups x 8 source groups
constexpr int kSc1 = 16;                // cache-policy operand of the buffer builtins on gfx950: bit 4 = sc1 (agent scope)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kChipStreamGroups = 64;      // groups of 192 entries a wave walks per update in the streamed form (8 slots x 64 rows x 24 entries / 192)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t chip_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double lo_f64(const u32x4 &g) { return __hiloint2double((int)g.y, (int)g.x); }
__device__ __forceinline__ double hi_f64(const u32x4 &g) { return __hiloint2double((int)g.w, (int)g.z); }
__device__ __forceinline__ u32x4 pack_f64x2(double a, double b) {
    u32x4 g;
    g.x = (unsigned)__double2loint(a);
    g.y = (unsigned)__double2hiint(a);
    g.z = (unsigned)__double2loint(b);
    g.w = (unsigned)__double2hiint(b);
    return g;
}
__device__ __forceinline__ bool row_valid_bits(unsigned lens, int k) { return ((lens >> (4 * k)) & 8u) != 0; }
__device__ __forceinline__ bool is_pending(const u32x4 &g) {
    return (g.x == (unsigned)(kChipPending & 0xffffffffu) && g.y == (unsigned)(kChipPending >> 32)) ||
           (g.z == (unsigned)(kChipPending & 0xffffffffu) && g.w == (unsigned)(kChipPending >> 32));
}


// State of a workgroup's side of the chip-wide exchange.  LDS: sh = 2 x 16 doubles (block sums, two halves in turn), s_res = 2 x 2
// doubles (the reduced pair, two sets in turn), s_flag one int.
struct Exchange {
    __amdgpu_buffer_rsrc_t part_rs;         // the slot array: kChipSlotBytes, all preset to the pending pattern at launch
    int v, grp, rank;                       // the workgroup's virtual index = 32 grp + rank
    unsigned gen = 0;
    int sum_phase = 0;
    bool local = false;                     // every group on one XCD: group-level slots are stored plainly (they stay in that XCD's L2)
    double *sh;
    double (*s_res)[2];
    int *s_flag;
    int *err;                               // device flag: a wait ran out somewhere
    unsigned long long *wait_acc = nullptr; // trace: LDS word that thread 0 adds the slot-polling time to (null: no trace)
};

// polls one slot per lane (lanes < count of wave 0) until none is pending; false when the wait ran out
__device__ __forceinline__ bool poll_slots(const Exchange &X, u32x4 &sv, int off, int count) {
    const int t = threadIdx.x;
    const bool mine = t < count;
    sv = pack_f64x2(0.0, 0.0);
    if (mine) sv = __builtin_amdgcn_raw_buffer_load_b128(X.part_rs, off, 0, kSc1);
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while (__ballot(mine && is_pending(sv)) != 0) {
        __builtin_amdgcn_s_sleep(1);
        if (mine && is_pending(sv)) sv = __builtin_amdgcn_raw_buffer_load_b128(X.part_rs, off, 0, kSc1);
        if ((++spins & 255u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > kChipSpinTicks || __hip_atomic_load(X.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                atomicExch(X.err, 1);
                return false;
            }
        }
    }
    return true;
}

// Two chip-wide sums at once, and a chip barrier in the same breath.  In two hops: the 32 workgroups of a group exchange their pairs
// through 16-byte slots that held a reserved NaN pattern (one store each; wave 0 of every workgroup polls the group's 32 slots) and
// sum them in the wave tree (lanes 32-63 add +0.0); eight members of each group then hand the group's pair to the eight groups
// (one 128-byte line of eight slots per destination group) and everybody sums its group's line in the wave tree (lanes 8-63 add
// +0.0).  Every workgroup adds the same values in the same order: bit-identical results everywhere, no broadcast.  Four slot sets
// rotate; a slot is re-armed two generations ahead behind a drain.  `publish`: the workgroup's stores must be visible to whoever
// passes this point, so every wave drains them first.  Returns false when a wait ran out (20 ms) here or anywhere.
__device__ __forceinline__ bool exchange2(Exchange &X, double a, double b2, bool publish, double &ra, double &rb) {
    const int t = threadIdx.x;
    if (publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double *slot = X.sh + (X.sum_phase & 1) * 16;
    ++X.sum_phase;
    a = wave_sum(a);
    b2 = wave_sum(b2);
    if ((t & 63) == 63) {
        slot[t >> 6] = a;
        slot[8 + (t >> 6)] = b2;
    }
    __syncthreads();                                           // (behind every wave's drain)
    const int set_cur = (int)(X.gen & 3u), set_nxt = (int)((X.gen + 2u) & 3u);
    double *const sres = X.s_res[X.gen & 1u];
    ++X.gen;
    if (t < 64) {                                              // wave 0 does the exchange
        const unsigned plo = (unsigned)(kChipPending & 0xffffffffu), phi = (unsigned)(kChipPending >> 32);
        u32x4 pend;
        pend.x = plo; pend.y = phi; pend.z = plo; pend.w = phi;
        const bool timed = X.wait_acc != nullptr && t == 0;
        const unsigned long long w0 = timed ? wall_clock64() : 0;
        // hop 1: the group's 32 pairs
        if (t == 0) {
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int w = 0; w < kChipThreads / 64; ++w) {
                sa += slot[w];
                sb += slot[8 + w];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-arm of the previous generation has landed
            const int o_n = (set_nxt * kChipWGs + X.v) * 16, o_c = (set_cur * kChipWGs + X.v) * 16;
            if (X.local) {
                __builtin_amdgcn_raw_buffer_store_b128(pend, X.part_rs, o_n, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(sa, sb), X.part_rs, o_c, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(pend, X.part_rs, o_n, 0, kSc1);
                __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(sa, sb), X.part_rs, o_c, 0, kSc1);
            }
        }
        u32x4 sv;
        int ok = poll_slots(X, sv, (set_cur * kChipWGs + X.grp * 32 + t) * 16, 32) ? 1 : 0;
        const double ga = wave_sum(lo_f64(sv)), gb = wave_sum(hi_f64(sv));      // (lanes 32-63 add +0.0)
        // hop 2: eight members of the group hand its pair to the eight groups, everybody sums the eight pairs of its group's line
        if (t == 63 && X.rank < 8) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int o_n = kChipS1Bytes + ((set_nxt * 8 + X.rank) * 8 + X.grp) * 16, o_c = kChipS1Bytes + ((set_cur * 8 + X.rank) * 8 + X.grp) * 16;
            __builtin_amdgcn_raw_buffer_store_b128(pend, X.part_rs, o_n, 0, kSc1);
            __builtin_amdgcn_raw_buffer_store_b128(pack_f64x2(ga, gb), X.part_rs, o_c, 0, kSc1);
        }
        if (ok) ok = poll_slots(X, sv, kChipS1Bytes + ((set_cur * 8 + X.grp) * 8 + t) * 16, 8) ? 1 : 0;
        if (timed) *X.wait_acc += wall_clock64() - w0;
        const double ta = wave_sum(lo_f64(sv)), tb = wave_sum(hi_f64(sv));      // (lanes 8-63 add +0.0)
        if (t == 63) {
            sres[0] = ta;
            sres[1] = tb;
            *X.s_flag = ok;
        }
    }
    __syncthreads();
    if (!*X.s_flag) return false;
    ra = sres[0];
    rb = sres[1];
    return true;       // (sres is written again two reductions on, behind the barriers of the next one)
}

// Where the groups sit: every workgroup reports its XCD (XCC_ID), everybody reads the 256 answers; true when each group of 32
// workgroups (equal blockIdx % 8) found itself on ONE XCD.  xcc: 256 device words.  alive: false when the exchange timed out.
__device__ __forceinline__ bool groups_on_one_xcd(Exchange &X, int *xcc, bool &alive) {
    const int t = threadIdx.x;
    if (t == 0) {
        unsigned xid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
        __hip_atomic_store(xcc + X.v, (int)(xid & 0xf), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double d0 = 0.0, d1 = 0.0;
    alive = exchange2(X, 0.0, 0.0, true, d0, d1);
    if (!alive) return false;
    int same = 1;
    if (t < kChipWGs) {
        const int mine = __hip_atomic_load(xcc + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int first = __hip_atomic_load(xcc + (t & ~31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        same = mine == first ? 1 : 0;
    }
    return __syncthreads_and(same) != 0;
}

}  // namespace chip
}  // namespace dpcg
