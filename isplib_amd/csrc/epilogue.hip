// epilogue.hip -- the dense passes either side of the fused GCN aggregation (torch.ops.isplib.gcn_norm_spmm; the
// `normalize=True` callers, tests/dist/gcn/pyg-sparse.py:61-62).  The aggregation itself applies the self
// loop, D^-1/2, bias and ReLU when it writes a finished row (isplib_epilogue); what is left around it are row-wise passes
// over N x K matrices, HBM-bound, which the operator layer used to compose from ATen calls -- four in the backward, one of
// them a column sum over a tall-skinny matrix that ATen's reduce kernel ran 100x below the memory rate (1.16 ms for the
// 38 MB of a 233K x 41 matrix: round 4's kernel trace of the GCN epoch).  Here each side is ONE pass:
//   isplib_row_scale_hip           y[i,:] = scale[i] * x[i,:]             (forward: the right-hand D^-1/2, written straight
//                                                                          at the pitch the gather wants)
//   isplib_masked_scale_colsum_hip g = dz * (out > 0); gy = g * scale[i]; grad_bias[c] = sum_i g[i,c]   (backward prologue)
// No atomics: the column sums are per-block partials folded in one fixed order (bitwise reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/isplib_hip.h"
#include "common.h"

namespace isplib {

constexpr int EP_THREADS = 256;
constexpr int EP_ROWS = 512;                 // rows of one block

__global__ __launch_bounds__(EP_THREADS) void row_scale_kernel(int64_t n, int64_t k, const float *__restrict__ x, int64_t ldx,
                                                               const float *__restrict__ scale, float *__restrict__ y, int64_t ldy,
                                                               int kp) {
   const int tx = (int)threadIdx.x % kp, ty = (int)threadIdx.x / kp, ry = EP_THREADS / kp;
   const int64_t r0 = (int64_t)blockIdx.x * EP_ROWS, r1 = r0 + EP_ROWS < n ? r0 + EP_ROWS : n;
   for (int64_t r = r0 + ty; r < r1; r += ry) {
      const float s = scale[r];
      for (int64_t c = tx; c < ldy; c += kp) y[r * ldy + c] = c < k ? s * x[r * ldx + c] : 0.0f;      // the pitch's padding is defined
   }
}

// one block: EP_ROWS rows; the thread (ty, tx) walks rows ty, ty + RY, ... of column tx (+ kp, ...): a wave reads whole rows
__global__ __launch_bounds__(EP_THREADS) void masked_scale_colsum_kernel(int64_t n, int64_t k, const float *__restrict__ dz, int64_t lddz,
                                                                         const float *__restrict__ out, int64_t ldo,
                                                                         const float *__restrict__ scale, float *__restrict__ gy,
                                                                         int64_t ldgy, float *__restrict__ partial, int kp) {
   __shared__ float s_acc[EP_THREADS];
   const int tx = (int)threadIdx.x % kp, ty = (int)threadIdx.x / kp, ry = EP_THREADS / kp;
   const int64_t r0 = (int64_t)blockIdx.x * EP_ROWS, r1 = r0 + EP_ROWS < n ? r0 + EP_ROWS : n;
   const int64_t cend = gy && ldgy > k ? ldgy : k;             // gy at a wider pitch: its padding columns are written (0)
   for (int64_t c0 = 0; c0 < cend; c0 += kp) {
      const int64_t c = c0 + tx;
      float acc = 0.0f;
      if (c < cend) {
         for (int64_t r = r0 + ty; r < r1; r += ry) {
            float v = 0.0f;
            if (c < k) {
               v = dz[r * lddz + c];
               if (out && !(out[r * ldo + c] > 0.0f)) v = 0.0f;
               acc += v;
            }
            if (gy) gy[r * ldgy + c] = scale ? v * scale[r] : v;
         }
      }
      if (partial) {
         s_acc[threadIdx.x] = acc;
         __syncthreads();
         if (ty == 0 && c < k) {
            float t = 0.0f;
            for (int q = 0; q < ry; q++) t += s_acc[q * kp + tx];        // fixed order
            partial[(int64_t)blockIdx.x * k + c] = t;
         }
         __syncthreads();
      }
   }
}

// one wave per column: lane l adds the partials of blocks l, l + 64, ... in ascending order, then the 64 lane sums are added by a
// fixed butterfly -- one order of additions whatever the launch, and no chain of `blocks` dependent loads in one thread (the first
// form, one thread per column, took 0.10 ms for 455 blocks: as long as the pass it finishes)
__global__ __launch_bounds__(256) void colsum_fold_kernel(int64_t blocks, int64_t k, const float *__restrict__ partial, float *__restrict__ grad_bias) {
   const int lane = threadIdx.x & 63;
   const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
   if (c >= k) return;                                     // wave-uniform
   float t = 0.0f;
   for (int64_t b = lane; b < blocks; b += 64) t += partial[b * k + c];
#pragma unroll
   for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
   if (lane == 0) grad_bias[c] = t;
}

static int lanes_for(int64_t k) {
   int kp = 16;
   while (kp < 256 && kp < k) kp *= 2;
   return kp;
}

}  // namespace isplib

using namespace isplib;

extern "C" int isplib_row_scale_hip(int64_t n, int64_t k, const float *x, int64_t ldx, const float *scale, float *y, int64_t ldy,
                                    void *stream) {
   clear_error();
   if (n < 0 || k < 0) return fail(ISPLIB_FAIL, "isplib_row_scale_hip: negative dimension");
   if (n == 0 || k == 0) return ISPLIB_SUCCESS;
   if (!x || !scale || !y) return fail(ISPLIB_FAIL, "isplib_row_scale_hip: null operand");
   if (ldx < k || ldy < k) return fail(ISPLIB_FAIL, "isplib_row_scale_hip: leading dimension smaller than k");
   const int64_t blocks = (n + EP_ROWS - 1) / EP_ROWS;
   if (blocks > 0x7FFFFFFF) return fail(ISPLIB_FAIL, "isplib_row_scale_hip: too many rows");
   hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)blocks), dim3(EP_THREADS), 0, (hipStream_t)stream, n, k, x, ldx, scale, y, ldy,
                      lanes_for(ldy));
   return check_launch("row_scale_kernel");
}

extern "C" size_t isplib_masked_scale_colsum_workspace_bytes(int64_t n, int64_t k) {
   if (n <= 0 || k <= 0) return 256;
   const size_t blocks = (size_t)((n + EP_ROWS - 1) / EP_ROWS);
   return (blocks * (size_t)k * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" int isplib_masked_scale_colsum_hip(int64_t n, int64_t k, const float *dz, int64_t lddz, const float *out, int64_t ldo,
                                              const float *scale, float *gy, int64_t ldgy, float *grad_bias, void *workspace,
                                              size_t workspace_bytes, void *stream) {
   clear_error();
   if (n < 0 || k < 0) return fail(ISPLIB_FAIL, "isplib_masked_scale_colsum_hip: negative dimension");
   hipStream_t st = (hipStream_t)stream;
   if (k == 0) return ISPLIB_SUCCESS;
   if (n == 0) {
      if (grad_bias) ISPLIB_HIP_TRY(hipMemsetAsync(grad_bias, 0, (size_t)k * sizeof(float), st));
      return ISPLIB_SUCCESS;
   }
   if (!dz || (!gy && !grad_bias)) return fail(ISPLIB_FAIL, "isplib_masked_scale_colsum_hip: null operand");
   if (lddz < k || (out && ldo < k) || (gy && ldgy < k)) return fail(ISPLIB_FAIL, "isplib_masked_scale_colsum_hip: leading dimension smaller than k");
   const int64_t blocks = (n + EP_ROWS - 1) / EP_ROWS;
   if (blocks > 0x7FFFFFFF) return fail(ISPLIB_FAIL, "isplib_masked_scale_colsum_hip: too many rows");
   if (grad_bias) {
      if (!workspace || workspace_bytes < isplib_masked_scale_colsum_workspace_bytes(n, k))
         return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_masked_scale_colsum_hip: workspace too small");
      if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "isplib_masked_scale_colsum_hip: workspace must be 256-byte aligned");
   }
   float *partial = grad_bias ? (float *)workspace : nullptr;
   hipLaunchKernelGGL(masked_scale_colsum_kernel, dim3((unsigned)blocks), dim3(EP_THREADS), 0, st, n, k, dz, lddz, out, ldo, scale, gy, ldgy,
                      partial, lanes_for(gy && ldgy > k ? ldgy : k));
   int rc = check_launch("masked_scale_colsum_kernel");
   if (rc) return rc;
   if (grad_bias) {
      hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, st, blocks, k, partial, grad_bias);
      rc = check_launch("colsum_fold_kernel");
   }
   return rc;
}
