// prims.hip -- the one translation unit that instantiates rocPRIM (see prims.h)
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "prims.h"

namespace isplib {

hipError_t sort_pairs_u32(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in,
                          uint32_t *vals_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st) {
   return rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t *, uint32_t *, const uint32_t *, uint32_t *>(
       temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st, false);
}

hipError_t sort_pairs_u32_f32(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const float *vals_in,
                              float *vals_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st) {
   return rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t *, uint32_t *, const float *, float *>(
       temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st, false);
}

hipError_t sort_keys_u64(void *temp, size_t &temp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t n, unsigned begin_bit,
                         unsigned end_bit, hipStream_t st) {
   return rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, n, begin_bit, end_bit, st, false);
}

hipError_t scan_exclusive_i32(void *temp, size_t &temp_bytes, const int *in, int *out, size_t n, hipStream_t st) {
   return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0, n, rocprim::plus<int>(), st, false);
}

hipError_t scan_exclusive_i64(void *temp, size_t &temp_bytes, const int64_t *in, int64_t *out, size_t n, hipStream_t st) {
   return rocprim::exclusive_scan(temp, temp_bytes, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), st, false);
}

}  // namespace isplib
