// backward_det.hip -- the backward of SpMM-max/min WITHOUT atomics (isplib_spmm_minmax_bw_det_hip).
//
// The reference's CPU backward (csrc/fusedmm.cpp:410-451 / 477-517: gather, mul, masked_fill, scatter_add_) is
// deterministic; the one-pass scatter of backward.hip (float atomics) is not: two launches can differ in the last
// bits.  This form gives bitwise reproducible gradients at about the same cost:
//   grad_mat[j, c] = sum over the rows i whose arg[i, c] names an entry of column j, of val[arg] * grad_out[i, c]:
//       every (i, c) becomes a (key = j * k + c, value) pair -- the key IS the flat index of its destination -- the
//       pairs are sorted by key with a STABLE radix sort (rocPRIM; equal keys keep ascending i), and the thread that
//       finds the head of a run adds the run up in that order and stores it.  No two threads write one element.
//   grad_val[a]    = sum over the features c of row i with arg[i, c] == a of mat[indx[a], c] * grad_out[i, c]:
//       every destination of row i is an entry OF row i, so one thread per row walks its k features in order and
//       updates grad_val in place.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/isplib_hip.h"
#include "common.h"

#include <rocprim/rocprim.hpp>

namespace isplib {

__global__ __launch_bounds__(256) void minmax_pairs_kernel(int64_t total, int64_t k, int64_t nnz, uint32_t limit,
                                                           const int64_t *__restrict__ indx, const float *__restrict__ val,
                                                           const int64_t *__restrict__ arg, const float *__restrict__ grad_out,
                                                           uint32_t *__restrict__ keys, float *__restrict__ vals) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int64_t a = arg[t];
      uint32_t key = limit;                            // "no winner" (arg == nnz): sorts behind every destination
      float v = 0.0f;
      if (a >= 0 && a < nnz) {
         key = (uint32_t)(indx[a] * k + t % k);
         v = (val ? val[a] : 1.0f) * grad_out[t];
      }
      keys[t] = key;
      vals[t] = v;
   }
}

__global__ __launch_bounds__(256) void minmax_runs_kernel(int64_t total, uint32_t limit, const uint32_t *__restrict__ keys,
                                                          const float *__restrict__ vals, float *__restrict__ grad_mat) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const uint32_t key = keys[t];
      if (key >= limit || (t > 0 && keys[t - 1] == key)) continue;      // not the head of a run of destinations
      float acc = vals[t];
      for (int64_t u = t + 1; u < total && keys[u] == key; u++) acc += vals[u];
      grad_mat[key] = acc;
   }
}

__global__ __launch_bounds__(256) void minmax_dval_rows_kernel(int64_t m, int64_t k, int64_t nnz, const int64_t *__restrict__ indx,
                                                               const float *__restrict__ mat, const int64_t *__restrict__ arg,
                                                               const float *__restrict__ grad_out, float *grad_val) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
      const int64_t *ar = arg + i * k;
      const float *gr = grad_out + i * k;
      for (int64_t c = 0; c < k; c++) {
         const int64_t a = ar[c];
         if (a < 0 || a >= nnz) continue;
         grad_val[a] += mat[indx[a] * k + c] * gr[c];   // a is an entry of row i: this thread is its only writer
      }
   }
}

static inline unsigned bits_for(uint64_t n) {
   unsigned b = 1;
   while (b < 32 && ((uint64_t)1 << b) < n) b++;
   return b;
}

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

static hipError_t pair_sort_temp(int64_t total, unsigned bits, size_t *bytes) {
   *bytes = 0;
   return rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t *, uint32_t *, const float *, float *>(
       nullptr, *bytes, nullptr, nullptr, nullptr, nullptr, (size_t)total, 0u, bits, (hipStream_t)0, false);
}

}  // namespace isplib

using namespace isplib;

extern "C" size_t isplib_spmm_minmax_bw_workspace_bytes(int64_t m, int64_t n, int64_t k) {
   if (m <= 0 || n <= 0 || k <= 0) return 256;
   const double dest = (double)n * (double)k;
   if (dest + 1.0 >= 4294967295.0 || (double)m * (double)k >= 2147483647.0 * 2.0) return 0;      // not served: use the atomic form
   const int64_t total = m * k;
   size_t temp = 0;
   if (pair_sort_temp(total, bits_for((uint64_t)(n * k + 1)), &temp) != hipSuccess) { (void)hipGetLastError(); return 0; }
   return 4 * up256((size_t)total * 4) + up256(temp) + 256;
}

extern "C" int isplib_spmm_minmax_bw_det_hip(int64_t m, int64_t n, int64_t k, int64_t nnz, const int64_t *indx, const float *val,
                                             const float *mat, const int64_t *arg, const float *grad_out, float *grad_mat,
                                             float *grad_val, void *workspace, size_t workspace_bytes, void *stream) {
   clear_error();
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "isplib_spmm_minmax_bw_det_hip: negative dimension");
   hipStream_t st = (hipStream_t)stream;
   if (grad_mat && n * k > 0) ISPLIB_HIP_TRY(hipMemsetAsync(grad_mat, 0, (size_t)n * (size_t)k * sizeof(float), st));
   if (grad_val && nnz > 0) ISPLIB_HIP_TRY(hipMemsetAsync(grad_val, 0, (size_t)nnz * sizeof(float), st));
   const int64_t total = m * k;
   if (total == 0 || nnz == 0 || (!grad_mat && !grad_val)) return ISPLIB_SUCCESS;
   if (!indx || !arg || !grad_out) return fail(ISPLIB_FAIL, "isplib_spmm_minmax_bw_det_hip: null operand");
   if (grad_val && !mat) return fail(ISPLIB_FAIL, "isplib_spmm_minmax_bw_det_hip: grad_val needs mat");
   if (grad_val) {
      int64_t blocks = (m + 255) / 256;
      if (blocks > 256 * 32) blocks = 256 * 32;
      hipLaunchKernelGGL(minmax_dval_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, m, k, nnz, indx, mat, arg, grad_out, grad_val);
      const int rc = check_launch("minmax_dval_rows_kernel");
      if (rc) return rc;
   }
   if (!grad_mat) return ISPLIB_SUCCESS;
   const size_t need = isplib_spmm_minmax_bw_workspace_bytes(m, n, k);
   if (need == 0) return fail(ISPLIB_NO_OPT_IMPL, "isplib_spmm_minmax_bw_det_hip: n*k or m*k beyond 32-bit keys (use isplib_spmm_minmax_bw_hip)");
   if (!workspace || workspace_bytes < need) return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_spmm_minmax_bw_det_hip: workspace too small");
   if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "isplib_spmm_minmax_bw_det_hip: workspace must be 256-byte aligned");
   const size_t plane = up256((size_t)total * 4);
   char *w = (char *)workspace;
   uint32_t *keys_in = (uint32_t *)w, *keys_out = (uint32_t *)(w + plane);
   float *vals_in = (float *)(w + 2 * plane), *vals_out = (float *)(w + 3 * plane);
   void *temp = w + 4 * plane;
   const uint32_t limit = (uint32_t)(n * k);
   const unsigned bits = bits_for((uint64_t)limit + 1);
   size_t temp_bytes = 0;
   ISPLIB_HIP_TRY(pair_sort_temp(total, bits, &temp_bytes));
   int64_t blocks = (total + 255) / 256;
   if (blocks > 256 * 32) blocks = 256 * 32;
   hipLaunchKernelGGL(minmax_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, total, k, nnz, limit, indx, val, arg, grad_out,
                      keys_in, vals_in);
   int rc = check_launch("minmax_pairs_kernel");
   if (rc) return rc;
   ISPLIB_HIP_TRY((rocprim::radix_sort_pairs<rocprim::default_config, const uint32_t *, uint32_t *, const float *, float *>(
       temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)total, 0u, bits, st, false)));
   hipLaunchKernelGGL(minmax_runs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, total, limit, keys_out, vals_out, grad_mat);
   return check_launch("minmax_runs_kernel");
}
