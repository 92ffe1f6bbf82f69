// Device-wide sort / scan primitives used by the setup routines (COO -> CSR, transpose, level sets, reordering).
// Thin, non-template entry points over rocPRIM, compiled once in dpcg_prims.hip; every call allocates its own scratch
// (setup paths only -- nothing here runs inside the PCG iteration).  All return a dpcg_status.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpcg {

// Stable LSD radix sort of (key, value) pairs on bits [0, end_bit) of the key.  in/out must not alias.
int sort_pairs_u64_i32(const uint64_t *keys_in, uint64_t *keys_out, const int32_t *vals_in, int32_t *vals_out,
                       int64_t count, int end_bit, hipStream_t s);
int sort_pairs_u32_i32(const uint32_t *keys_in, uint32_t *keys_out, const int32_t *vals_in, int32_t *vals_out,
                       int64_t count, int end_bit, hipStream_t s);
// out[i] = in[0] + ... + in[i-1] (exclusive) or ... + in[i] (inclusive); in == out is allowed.
int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t count, hipStream_t s);
int inclusive_scan_i32(const int32_t *in, int32_t *out, int64_t count, hipStream_t s);
// exclusive scan in a caller-provided workspace of scan_workspace_bytes(count) bytes: no allocation, no synchronisation
size_t scan_workspace_bytes(int64_t count);
int exclusive_scan_i32_ws(const int32_t *in, int32_t *out, int64_t count, void *workspace, size_t workspace_bytes, hipStream_t s);
// *out_dev = max(in[0..count))
int reduce_max_i32(const int32_t *in, int32_t *out_dev, int64_t count, hipStream_t s);
// number of bits needed to represent values in [0, max_value]
inline int bits_for(uint64_t max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

}  // namespace dpcg
