// reorder.hip -- a locality ordering of the rows of a square graph, on the device: synchronous label propagation
// (Raghavan, Albert, Kumara 2007) + a stable sort of the rows by label.  For fusedMM_csr_ordered_hip (spmm.hip): with a
// dense operand larger than every cache (the ogbn-products shape: 2.5 GB at K = 256) the only reuse a gathered row can
// have is the graph's own -- rows of one community gather mostly each other -- and it is only realised when those rows
// are worked on at the same time behind the same L2.  One-off per graph, integer work only; rocPRIM radix sorts plus
// three small kernels.  isplib_amd/reorder.py is the same algorithm in torch ops (the tests hold the two to identical
// labels).  No reference counterpart (the reference's CPU kernel walks rows in index order, csrc/fusedmm.cpp:198).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>

#include "prims.h"

#include "../../include/isplib_hip.h"
#include "common.h"

namespace isplib {

static inline unsigned ro_grid(int64_t items) {
   const int64_t b = (items + 255) / 256;
   return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

// key of stored entry e: (its row << 32) | the label its column carries
__global__ __launch_bounds__(256) void ro_keys_kernel(int64_t m, int64_t nnz, const int64_t *__restrict__ rowptr,
                                                      const int64_t *__restrict__ col, const uint32_t *__restrict__ labels,
                                                      uint64_t *__restrict__ keys) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
      int64_t lo = 0, hi = m;                            // last row with rowptr[row] <= e
      while (hi - lo > 1) {
         const int64_t mid = (lo + hi) >> 1;
         if (rowptr[mid] <= e) lo = mid; else hi = mid;
      }
      keys[e] = ((uint64_t)lo << 32) | (uint64_t)labels[col[e]];
   }
}

// the label most entries of a row carry; ties: a per-round hash of the label, then the larger label (the same
// arithmetic as isplib_amd/reorder.py: label_propagation)
__global__ __launch_bounds__(256) void ro_mode_kernel(int64_t m, const int64_t *__restrict__ rowptr, const uint64_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ labels, uint32_t *__restrict__ out, int64_t salt,
                                                      unsigned long long *__restrict__ changed) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   unsigned long long mine = 0;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
      const int64_t b = rowptr[i], e = rowptr[i + 1];
      uint32_t best_lab = labels[i];
      int64_t best = -1;
      int64_t p = b;
      while (p < e) {
         const uint32_t lab = (uint32_t)keys[p];
         int64_t q = p + 1;
         while (q < e && (uint32_t)keys[q] == lab) q++;
         const int64_t tie = (int64_t)((((uint64_t)lab * 2654435761ull + (uint64_t)salt) >> 7) & 0xFFFFFull);
         const int64_t score = ((q - p) << 20) + tie;
         if (score > best || (score == best && lab > best_lab)) { best = score; best_lab = lab; }
         p = q;
      }
      if (best_lab != labels[i]) mine++;
      out[i] = best_lab;
   }
   if (mine) atomicAdd(changed, mine);
}

__global__ __launch_bounds__(256) void ro_iota_kernel(int64_t m, uint32_t *__restrict__ a, uint32_t *__restrict__ b) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
      a[i] = (uint32_t)i;
      if (b) b[i] = (uint32_t)i;
   }
}

__global__ __launch_bounds__(256) void ro_positions_kernel(int64_t m, const int32_t *__restrict__ order, int32_t *__restrict__ pos) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) pos[order ? order[i] : i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void ro_near_kernel(int64_t m, int64_t nnz, const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
                                                      const int32_t *__restrict__ pos, int64_t window, unsigned long long *__restrict__ near) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   unsigned long long mine = 0;
   for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
      int64_t lo = 0, hi = m;
      while (hi - lo > 1) {
         const int64_t mid = (lo + hi) >> 1;
         if (rowptr[mid] <= e) lo = mid; else hi = mid;
      }
      const int64_t d = (int64_t)pos[lo] - (int64_t)pos[col[e]];
      mine += (d <= window && -d <= window) ? 1 : 0;
   }
   if (mine) atomicAdd(near, mine);
}

static int ro_bits(int64_t v) {
   int b = 1;
   while (b < 63 && (1LL << b) < v) b++;
   return b;
}

static size_t ro_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct RoTemp { size_t keys, pairs; };
static RoTemp ro_temp(int64_t m, int64_t nnz) {
   RoTemp t = {0, 0};
   (void)sort_keys_u64(nullptr, t.keys, nullptr, nullptr, (size_t)(nnz > 0 ? nnz : 1), 0, 64, (hipStream_t)0);
   (void)sort_pairs_u32(nullptr, t.pairs, nullptr, nullptr, nullptr, nullptr, (size_t)(m > 0 ? m : 1), 0, 32, (hipStream_t)0);
   return t;
}

}  // namespace isplib

using namespace isplib;

extern "C" size_t isplib_community_order_workspace_bytes(int64_t m, int64_t nnz) {
   if (m <= 0) return 256;
   const RoTemp t = ro_temp(m, nnz);
   const size_t e = (size_t)(nnz > 0 ? nnz : 1), r = (size_t)m;
   return 2 * ro_align(e * 8) + 4 * ro_align(r * 4) + ro_align(t.keys > t.pairs ? t.keys : t.pairs) + 512;
}

extern "C" int isplib_community_order_hip(int64_t m, int64_t nnz, const int64_t *rowptr, const int64_t *col, int rounds, int seed,
                                          int32_t *order, int32_t *labels_out, int *rounds_run, void *workspace,
                                          size_t workspace_bytes, void *stream) {
   clear_error();
   if (m < 0 || nnz < 0) return fail(ISPLIB_FAIL, "isplib_community_order_hip: negative dimension");
   if (m == 0) return ISPLIB_SUCCESS;
   if (m >= (1LL << 31) || nnz >= (1LL << 40)) return fail(ISPLIB_FAIL, "isplib_community_order_hip: m < 2^31 required");
   if (!rowptr || !order || (nnz > 0 && !col)) return fail(ISPLIB_FAIL, "isplib_community_order_hip: null operand");
   if (rounds < 0) rounds = 0;
   if (!workspace || workspace_bytes < isplib_community_order_workspace_bytes(m, nnz))
      return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_community_order_hip: workspace too small (isplib_community_order_workspace_bytes)");
   if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "isplib_community_order_hip: workspace must be 256-byte aligned");
   hipStream_t st = (hipStream_t)stream;
   const RoTemp t = ro_temp(m, nnz);
   const size_t e = (size_t)(nnz > 0 ? nnz : 1), r = (size_t)m;
   char *w = (char *)workspace;
   uint64_t *keys_a = (uint64_t *)w; w += ro_align(e * 8);
   uint64_t *keys_b = (uint64_t *)w; w += ro_align(e * 8);
   uint32_t *lab_a = (uint32_t *)w; w += ro_align(r * 4);
   uint32_t *lab_b = (uint32_t *)w; w += ro_align(r * 4);
   uint32_t *ids = (uint32_t *)w; w += ro_align(r * 4);
   uint32_t *lab_sorted = (uint32_t *)w; w += ro_align(r * 4);
   void *temp = (void *)w; w += ro_align(t.keys > t.pairs ? t.keys : t.pairs);
   unsigned long long *changed = (unsigned long long *)w;
   hipLaunchKernelGGL(ro_iota_kernel, dim3(ro_grid(m)), dim3(256), 0, st, m, lab_a, ids);
   int rc = check_launch("ro_iota_kernel");
   if (rc) return rc;
   const int end_bit = 32 + ro_bits(m);
   int done = 0;
   for (int round = 0; round < rounds && nnz > 0; round++) {
      ISPLIB_HIP_TRY(hipMemsetAsync(changed, 0, sizeof(unsigned long long), st));
      hipLaunchKernelGGL(ro_keys_kernel, dim3(ro_grid(nnz)), dim3(256), 0, st, m, nnz, rowptr, col, lab_a, keys_a);
      if ((rc = check_launch("ro_keys_kernel")) != 0) return rc;
      size_t tb = t.keys;
      ISPLIB_HIP_TRY(sort_keys_u64(temp, tb, (const uint64_t *)keys_a, keys_b, (size_t)nnz, 0, end_bit < 64 ? end_bit : 64, st));
      const int64_t salt = (int64_t)(seed + round) * 40503 + 12345;
      hipLaunchKernelGGL(ro_mode_kernel, dim3(ro_grid(m)), dim3(256), 0, st, m, rowptr, keys_b, lab_a, lab_b, salt, changed);
      if ((rc = check_launch("ro_mode_kernel")) != 0) return rc;
      unsigned long long host = 0;
      ISPLIB_HIP_TRY(hipMemcpyAsync(&host, changed, sizeof(host), hipMemcpyDeviceToHost, st));
      ISPLIB_HIP_TRY(hipStreamSynchronize(st));
      uint32_t *sw = lab_a; lab_a = lab_b; lab_b = sw;
      done = round + 1;
      if ((double)host < 0.01 * (double)m) break;        // fewer than 1 % of the rows changed: settled
   }
   if (rounds_run) *rounds_run = done;
   if (labels_out) ISPLIB_HIP_TRY(hipMemcpyAsync(labels_out, lab_a, r * 4, hipMemcpyDeviceToDevice, st));
   // rows grouped by label, in index order inside a group: a stable sort of (label, row id)
   size_t tb = t.pairs;
   ISPLIB_HIP_TRY(sort_pairs_u32(temp, tb, (const uint32_t *)lab_a, lab_sorted, (const uint32_t *)ids, (uint32_t *)order, (size_t)m, 0,
                                 ro_bits(m) < 32 ? ro_bits(m) : 32, st));
   return ISPLIB_SUCCESS;
}

// share of the stored entries whose column sits within `window` positions of its row in `order` (NULL: index order):
// what an order found, without running a kernel on it; synchronises the stream
extern "C" int isplib_order_locality_hip(int64_t m, int64_t nnz, const int64_t *rowptr, const int64_t *col, const int32_t *order,
                                         int64_t window, double *share, void *workspace, size_t workspace_bytes, void *stream) {
   clear_error();
   if (!share) return fail(ISPLIB_FAIL, "isplib_order_locality_hip: share is NULL");
   *share = 0.0;
   if (m <= 0 || nnz <= 0) return ISPLIB_SUCCESS;
   if (m >= (1LL << 31)) return fail(ISPLIB_FAIL, "isplib_order_locality_hip: m < 2^31 required");
   if (!rowptr || !col) return fail(ISPLIB_FAIL, "isplib_order_locality_hip: null operand");
   const size_t need = ro_align((size_t)m * 4) + 256;
   if (!workspace || workspace_bytes < need) return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_order_locality_hip: workspace too small (4 m + 512 bytes)");
   hipStream_t st = (hipStream_t)stream;
   int32_t *pos = (int32_t *)workspace;
   unsigned long long *near = (unsigned long long *)((char *)workspace + ro_align((size_t)m * 4));
   ISPLIB_HIP_TRY(hipMemsetAsync(near, 0, sizeof(unsigned long long), st));
   hipLaunchKernelGGL(ro_positions_kernel, dim3(ro_grid(m)), dim3(256), 0, st, m, order, pos);
   int rc = check_launch("ro_positions_kernel");
   if (rc) return rc;
   hipLaunchKernelGGL(ro_near_kernel, dim3(ro_grid(nnz)), dim3(256), 0, st, m, nnz, rowptr, col, pos, window, near);
   if ((rc = check_launch("ro_near_kernel")) != 0) return rc;
   unsigned long long host = 0;
   ISPLIB_HIP_TRY(hipMemcpyAsync(&host, near, sizeof(host), hipMemcpyDeviceToHost, st));
   ISPLIB_HIP_TRY(hipStreamSynchronize(st));
   *share = (double)host / (double)nnz;
   return ISPLIB_SUCCESS;
}
