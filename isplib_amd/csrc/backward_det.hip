// backward_det.hip -- the backward of SpMM-max/min WITHOUT atomics (isplib_spmm_minmax_bw_det_hip).
//
// The reference's CPU backward (csrc/fusedmm.cpp:410-451 / 477-517: gather, mul, masked_fill, scatter_add_) is
// deterministic; the one-pass scatter of backward.hip (float atomics) is not: two launches can differ in the last
// bits.  This form gives bitwise reproducible gradients at about the same cost:
//   grad_mat[j, c] = sum over the rows i whose arg[i, c] names an entry of column j, of val[arg] * grad_out[i, c]:
//       every (i, c) becomes a (key = j * k + c, value) pair -- the key IS the flat index of its destination -- the
//       pairs are sorted by key with a STABLE radix sort (rocPRIM; equal keys keep ascending i), and every run of equal
//       keys is added up in that order (two levels, see below).  No two threads write one element.
//   grad_val[a]    = sum over the features c of row i with arg[i, c] == a of mat[indx[a], c] * grad_out[i, c]:
//       every destination of row i is an entry OF row i, so one wave per row adds the features of each entry in
//       lane order and updates grad_val in place.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/isplib_hip.h"
#include "common.h"

#include "prims.h"

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

// The same pairs from (destination row, value) arrays instead of (arg, indx, val, grad_out): what a rank of the 1-D row partition
// holds after the exchange of the max / min backward (isplib_amd/dist.py) -- every rank's winners' columns and weighted gradients,
// all-gathered in rank (= global row) order; only the destinations inside this rank's own rows [lo, lo + n) are kept.
__global__ __launch_bounds__(256) void dest_pairs_kernel(int64_t total, int64_t k, int64_t n, int64_t lo, uint32_t limit,
                                                         const int32_t *__restrict__ dest, const float *__restrict__ gval,
                                                         uint32_t *__restrict__ keys, float *__restrict__ vals) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int64_t d = (int64_t)dest[t] - lo;
      const bool mine = dest[t] >= 0 && d >= 0 && d < n;
      keys[t] = mine ? (uint32_t)(d * k + t % k) : limit;
      vals[t] = mine ? gval[t] : 0.0f;
   }
}

// Run sums in two levels, so that a destination with thousands of contributions (a hub column wins in many rows) does
// not leave one thread adding them up alone.  Level 1: every thread walks its own RUN_CHUNK consecutive sorted pairs;
// runs that begin and end inside the chunk are finished there; of a run that crosses chunk borders the thread keeps
// its piece: `open` = the piece of a run that begins here and goes on, `cont` = the piece of a run that began earlier.
// Level 2: the thread of every chunk with an open run adds the following chunks' cont pieces from left to right.
// Every sum is formed in one fixed order (ascending row inside a chunk, then chunk by chunk).
constexpr int RUN_CHUNK = 32;
enum { RUN_OPEN = 1, RUN_CONT_ENDS = 2, RUN_CONT_GOES_ON = 4 };

// (Round 4, second session: a tile of 256 chunks is staged through LDS by coalesced loads -- 8,192 keys and values, rows of 33
// words so that thread t walking row t meets no bank conflict -- where every thread used to walk its own 32 consecutive
// pairs straight from memory, 64 different cache lines per wave instruction: the kernel took as long as the radix sort in
// front of it, 0.43 ms for the 14.9 M pairs of K=64.  Same chunks, same order of every addition: the same bits.)
constexpr int RUN_THREADS = 256, RUN_PITCH = RUN_CHUNK + 1;

__global__ __launch_bounds__(RUN_THREADS) void minmax_chunks_kernel(int64_t total, uint32_t limit, const uint32_t *__restrict__ keys,
                                                                    const float *__restrict__ vals, float *__restrict__ grad_mat,
                                                                    float *__restrict__ open_sum, float *__restrict__ cont_sum,
                                                                    unsigned char *__restrict__ state) {
   __shared__ uint32_t sk[RUN_THREADS * RUN_PITCH];
   __shared__ float sv[RUN_THREADS * RUN_PITCH];
   const int64_t nchunks = (total + RUN_CHUNK - 1) / RUN_CHUNK;
   const int64_t ntiles = (nchunks + RUN_THREADS - 1) / RUN_THREADS;
   const int tid = (int)threadIdx.x;
   for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t tbase = tile * (int64_t)(RUN_THREADS * RUN_CHUNK);
#pragma unroll 4
      for (int e = tid; e < RUN_THREADS * RUN_CHUNK; e += RUN_THREADS) {
         const int64_t gi = tbase + e;
         const int at = (e / RUN_CHUNK) * RUN_PITCH + (e % RUN_CHUNK);
         sk[at] = gi < total ? keys[gi] : 0u;
         sv[at] = gi < total ? vals[gi] : 0.0f;
      }
      __syncthreads();
      const int64_t t = tile * RUN_THREADS + tid;
      if (t < nchunks) {
         const int64_t base = t * RUN_CHUNK, end = base + RUN_CHUNK < total ? base + RUN_CHUNK : total;
         const int n = (int)(end - base);
         const uint32_t *k = sk + tid * RUN_PITCH;
         const float *v = sv + tid * RUN_PITCH;
         const bool has_prev = base > 0;
         const uint32_t prev_key = !has_prev ? 0u : (tid > 0 ? sk[(tid - 1) * RUN_PITCH + RUN_CHUNK - 1] : keys[base - 1]);
         // the key behind this chunk's last pair (only looked at when the chunk is full and not the last one)
         const uint32_t next_key = end < total ? (tid + 1 < RUN_THREADS ? sk[(tid + 1) * RUN_PITCH] : keys[end]) : 0u;
         unsigned st = 0;
         float o = 0.0f, c = 0.0f;
         int i = 0;
         while (i < n) {
            const uint32_t key = k[i];
            float acc = v[i];
            int j = i + 1;
            while (j < n && k[j] == key) acc += v[j++];
            const bool begins_here = !(i == 0 && has_prev && key == prev_key);
            const bool ends_here = j < n || end == total || next_key != key;
            if (key < limit) {
               if (begins_here && ends_here) grad_mat[key] = acc;
               else if (begins_here) { o = acc; st |= RUN_OPEN; }
               else { c = acc; st |= ends_here ? RUN_CONT_ENDS : RUN_CONT_GOES_ON; }
            }
            i = j;
         }
         open_sum[t] = o;
         cont_sum[t] = c;
         state[t] = (unsigned char)st;
      }
      __syncthreads();                                     // the next tile overwrites the rows
   }
}

__global__ __launch_bounds__(256) void minmax_open_runs_kernel(int64_t total, const uint32_t *__restrict__ keys,
                                                               const float *__restrict__ open_sum, const float *__restrict__ cont_sum,
                                                               const unsigned char *__restrict__ state, float *__restrict__ grad_mat) {
   const int64_t nchunks = (total + RUN_CHUNK - 1) / RUN_CHUNK;
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nchunks; t += stride) {
      if (!(state[t] & RUN_OPEN)) continue;
      const uint32_t key = keys[(t + 1) * RUN_CHUNK - 1];                 // an open run reaches the end of its (full) chunk
      float acc = open_sum[t];
      for (int64_t u = t + 1; u < nchunks; u++) {
         const unsigned su = state[u];
         if (!(su & (RUN_CONT_ENDS | RUN_CONT_GOES_ON))) break;
         acc += cont_sum[u];
         if (su & RUN_CONT_ENDS) break;
      }
      grad_mat[key] = acc;
   }
}

// grad_val: one wave per row, lanes over the features.  Every lane adds up, in lane order, the contributions of all
// lanes of its 64-feature group that point at the same stored entry; the first such lane adds the sum to grad_val.
// The groups of a row follow each other in the same wave: one fixed order per entry, nobody else writes it.
__global__ __launch_bounds__(256) void minmax_dval_rows_kernel(int64_t m, int64_t k, int64_t nnz, const int64_t *__restrict__ indx,
                                                               const float *__restrict__ mat, const int64_t *__restrict__ arg,
                                                               const float *__restrict__ grad_out, float *grad_val) {
   const int lane = threadIdx.x & 63;
   const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
   for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < m; i += waves) {
      for (int64_t c0 = 0; c0 < k; c0 += 64) {
         const int64_t c = c0 + lane;
         int64_t a = -1;
         float v = 0.0f;
         if (c < k) {
            a = arg[i * k + c];
            if (a >= 0 && a < nnz) v = mat[indx[a] * k + c] * grad_out[i * k + c];
            else a = -1;
         }
         const int a_lo = (int)(uint32_t)a, a_hi = (int)(uint32_t)((uint64_t)a >> 32);
         const int v_bits = __float_as_int(v);
         float acc = 0.0f;
         bool first = a >= 0;
#pragma unroll 8
         for (int l = 0; l < 64; l++) {
            const bool same = a >= 0 && __builtin_amdgcn_readlane(a_lo, l) == a_lo && __builtin_amdgcn_readlane(a_hi, l) == a_hi;
            const float vl = __int_as_float(__builtin_amdgcn_readlane(v_bits, l));
            if (same) acc += vl;
            if (same && l < lane) first = false;
         }
         if (first) grad_val[a] += acc;
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
   return sort_pairs_u32_f32(nullptr, *bytes, nullptr, nullptr, nullptr, nullptr, (size_t)total, 0u, bits, (hipStream_t)0);
}

}  // namespace isplib

using namespace isplib;

static int sorted_run_sums(int64_t total, uint32_t limit, void *workspace, float *grad_mat, hipStream_t st);

extern "C" size_t isplib_spmm_minmax_bw_workspace_bytes(int64_t m, int64_t n, int64_t k) {
   if (m <= 0 || n <= 0 || k <= 0) return 256;
   const double dest = (double)n * (double)k;
   if (dest + 1.0 >= 4294967295.0 || (double)m * (double)k >= 2147483647.0 * 2.0) return 0;      // not served: use the atomic form
   const int64_t total = m * k;
   size_t temp = 0;
   if (pair_sort_temp(total, bits_for((uint64_t)(n * k + 1)), &temp) != hipSuccess) { (void)hipGetLastError(); return 0; }
   const size_t nchunks = ((size_t)total + RUN_CHUNK - 1) / RUN_CHUNK;
   return 4 * up256((size_t)total * 4) + up256(temp) + 2 * up256(nchunks * 4) + up256(nchunks) + 256;
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
      int64_t blocks = (m + 3) / 4;                 // one wave per row
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
   uint32_t *keys_in = (uint32_t *)w;
   float *vals_in = (float *)(w + 2 * plane);
   const uint32_t limit = (uint32_t)(n * k);
   int64_t blocks = (total + 255) / 256;
   if (blocks > 256 * 32) blocks = 256 * 32;
   hipLaunchKernelGGL(minmax_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, total, k, nnz, limit, indx, val, arg, grad_out,
                      keys_in, vals_in);
   const int rc = check_launch("minmax_pairs_kernel");
   if (rc) return rc;
   return sorted_run_sums(total, limit, workspace, grad_mat, st);
}

// the second half of both entries: the (key, value) pairs in the first and third plane of the workspace are sorted by key
// (stable) and every run of equal keys < limit is added up in order into grad_mat[key]
static int sorted_run_sums(int64_t total, uint32_t limit, void *workspace, float *grad_mat, hipStream_t st) {
   const size_t plane = up256((size_t)total * 4);
   char *w = (char *)workspace;
   uint32_t *keys_in = (uint32_t *)w, *keys_out = (uint32_t *)(w + plane);
   float *vals_in = (float *)(w + 2 * plane), *vals_out = (float *)(w + 3 * plane);
   void *temp = w + 4 * plane;
   const unsigned bits = bits_for((uint64_t)limit + 1);
   size_t temp_bytes = 0;
   ISPLIB_HIP_TRY(pair_sort_temp(total, bits, &temp_bytes));
   int rc;
   ISPLIB_HIP_TRY(sort_pairs_u32_f32(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)total, 0u, bits, st));
   const size_t nchunks = ((size_t)total + RUN_CHUNK - 1) / RUN_CHUNK;
   float *open_sum = (float *)((char *)temp + up256(temp_bytes));
   float *cont_sum = (float *)((char *)open_sum + up256(nchunks * 4));
   unsigned char *state = (unsigned char *)((char *)cont_sum + up256(nchunks * 4));
   int64_t cblocks = ((int64_t)nchunks + 255) / 256;
   if (cblocks > 256 * 32) cblocks = 256 * 32;
   hipLaunchKernelGGL(minmax_chunks_kernel, dim3((unsigned)cblocks), dim3(RUN_THREADS), 0, st, total, limit, keys_out, vals_out, grad_mat,
                      open_sum, cont_sum, state);
   rc = check_launch("minmax_chunks_kernel");
   if (rc) return rc;
   hipLaunchKernelGGL(minmax_open_runs_kernel, dim3((unsigned)cblocks), dim3(256), 0, st, total, keys_out, open_sum, cont_sum, state, grad_mat);
   return check_launch("minmax_open_runs_kernel");
}

// grad_mat[d - lo, c] = sum, in ascending i, of gval[i, c] over the (i, c) with dest[i, c] = d in [lo, lo + n): the local half of the
// max / min backward under the 1-D row partition (isplib_amd/dist.py: every rank all-gathers its winners' columns and weighted
// gradients and keeps the destinations that are its own rows).  Same sort and run sums as isplib_spmm_minmax_bw_det_hip, same
// workspace (isplib_spmm_minmax_bw_workspace_bytes(m, n, k)); dest < 0 = no winner.  No atomics: bitwise reproducible.
extern "C" int isplib_scatter_rows_det_hip(int64_t m, int64_t n, int64_t k, int64_t lo, const int32_t *dest, const float *gval,
                                           float *grad_mat, void *workspace, size_t workspace_bytes, void *stream) {
   clear_error();
   if (m < 0 || n < 0 || k < 0) return fail(ISPLIB_FAIL, "isplib_scatter_rows_det_hip: negative dimension");
   hipStream_t st = (hipStream_t)stream;
   if (!grad_mat && n * k > 0) return fail(ISPLIB_FAIL, "isplib_scatter_rows_det_hip: null operand");
   if (n * k > 0) ISPLIB_HIP_TRY(hipMemsetAsync(grad_mat, 0, (size_t)n * (size_t)k * sizeof(float), st));
   const int64_t total = m * k;
   if (total == 0 || n == 0) return ISPLIB_SUCCESS;
   if (!dest || !gval) return fail(ISPLIB_FAIL, "isplib_scatter_rows_det_hip: null operand");
   const size_t need = isplib_spmm_minmax_bw_workspace_bytes(m, n, k);
   if (need == 0) return fail(ISPLIB_NO_OPT_IMPL, "isplib_scatter_rows_det_hip: n*k or m*k beyond 32-bit keys");
   if (!workspace || workspace_bytes < need) return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_scatter_rows_det_hip: workspace too small");
   if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "isplib_scatter_rows_det_hip: workspace must be 256-byte aligned");
   const size_t plane = up256((size_t)total * 4);
   const uint32_t limit = (uint32_t)(n * k);
   int64_t blocks = (total + 255) / 256;
   if (blocks > 256 * 32) blocks = 256 * 32;
   hipLaunchKernelGGL(dest_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, total, k, n, lo, limit, dest, gval,
                      (uint32_t *)workspace, (float *)((char *)workspace + 2 * plane));
   const int rc = check_launch("dest_pairs_kernel");
   if (rc) return rc;
   return sorted_run_sums(total, limit, workspace, grad_mat, st);
}
