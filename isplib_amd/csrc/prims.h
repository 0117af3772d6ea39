// prims.h -- the library's device primitives (rocPRIM radix sorts and exclusive scans), instantiated ONCE in prims.hip.
// Each rocPRIM sort is ~4 MB of code objects per instantiation; four translation units used to carry their own copies
// (graph_prep, stream_plan, reorder, backward_det: 16.9 MB of library, a quarter of it needed).  temp == nullptr: the call
// only reports the scratch size in temp_bytes (rocPRIM's convention); no call synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace isplib {

// stable LSD radix sorts on bits [begin_bit, end_bit) of the key
hipError_t sort_pairs_u32(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st);
hipError_t sort_pairs_u32_f32(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const float *vals_in,
                              float *vals_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st);
hipError_t sort_keys_u64(void *temp, size_t &temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t n, unsigned begin_bit,
                         unsigned end_bit, hipStream_t st);
// exclusive prefix sums starting at 0
hipError_t scan_exclusive_i32(void *temp, size_t &temp_bytes, const int *in, int *out, size_t n, hipStream_t st);
hipError_t scan_exclusive_i64(void *temp, size_t &temp_bytes, const int64_t *in, int64_t *out, size_t n, hipStream_t st);

}  // namespace isplib
