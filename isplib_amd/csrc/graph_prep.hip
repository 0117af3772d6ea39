// graph_prep.hip -- CSR -> CSC operands on the device (the step immediately
// before the path).  Replaces the host-side torch_sparse storage calls the
// reference wrapper forces (isplib/__init__.py:67-73: row(), rowcount(),
// csr2csc(), colptr()) and the two nnz-sized gathers it caches per graph
// (:79-80 for sum, :86-99 for mean), which cost seconds per graph on the CPU.
//
// csr2csc is a STABLE sort of CSR positions by column id, so CSC order within a
// column is ascending CSR position = ascending row -- the order torch_sparse
// produces and the order that fixes which edge wins a max/min tie in A^T.
// The sort is rocPRIM's LSD radix sort (stable by construction) on 32-bit keys
// truncated to ceil(log2(n)) bits; everything else is hand-written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>

#include "prims.h"

#include "../../include/isplib_hip.h"
#include "common.h"

namespace isplib {

__device__ __forceinline__ int64_t row_of(const int64_t *__restrict__ rowptr, int64_t m, int64_t p) {
   // last r in [0, m) with rowptr[r] <= p  (empty rows are skipped naturally)
   int64_t lo = 0, hi = m;   // invariant: rowptr[lo] <= p < rowptr[hi]
   while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid] <= p) lo = mid; else hi = mid;
   }
   return lo;
}

__global__ __launch_bounds__(256) void row_ids_kernel(int64_t m, int64_t nnz, const int64_t *__restrict__ rowptr,
                                                      int64_t *__restrict__ row) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride)
      row[p] = row_of(rowptr, m, p);
}

__global__ __launch_bounds__(256) void make_keys_kernel(int64_t nnz, const int64_t *__restrict__ col,
                                                        uint32_t *__restrict__ keys, uint32_t *__restrict__ pos) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) {
      keys[p] = (uint32_t)col[p];
      pos[p] = (uint32_t)p;
   }
}

// colptr[c] = first CSC position whose column is >= c  (keys sorted ascending)
__global__ __launch_bounds__(256) void colptr_kernel(int64_t n, int64_t nnz, const uint32_t *__restrict__ keys,
                                                     int64_t *__restrict__ colptr) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= n; c += stride) {
      int64_t lo = 0, hi = nnz;   // first index in [0, nnz] with keys[idx] >= c
      while (lo < hi) {
         const int64_t mid = (lo + hi) >> 1;
         if ((int64_t)keys[mid] < c) lo = mid + 1; else hi = mid;
      }
      colptr[c] = lo;
   }
}

__global__ __launch_bounds__(256) void finalize_kernel(int64_t m, int64_t nnz, const int64_t *__restrict__ rowptr,
                                                       const float *__restrict__ val, int mean_scale,
                                                       const uint32_t *__restrict__ pos_sorted,
                                                       int64_t *__restrict__ csr2csc, int64_t *__restrict__ row_t,
                                                       float *__restrict__ val_t) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += stride) {
      const int64_t p = (int64_t)pos_sorted[q];
      const int64_t r = row_of(rowptr, m, p);
      if (csr2csc) csr2csc[q] = p;
      if (row_t) row_t[q] = r;
      if (val_t) {
         float v = val ? val[p] : 1.0f;
         if (mean_scale) {
            const int64_t deg = rowptr[r + 1] - rowptr[r];
            v = v / (float)(deg > 1 ? deg : 1);
         }
         val_t[q] = v;
      }
   }
}

// sliceptr[i*(S+1)+s] = first CSR position of row i whose column is >= s*width (columns sorted in-row)
__global__ __launch_bounds__(256) void slices_kernel(int64_t m, int64_t width, int slices,
                                                     const int64_t *__restrict__ pntrb,
                                                     const int64_t *__restrict__ pntre,
                                                     const int64_t *__restrict__ indx,
                                                     int64_t *__restrict__ sliceptr) {
   const int64_t per = slices + 1;
   const int64_t total = m * per;
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int64_t row = t / per;
      const int s = (int)(t - row * per);
      const int64_t b = pntrb[row], e = pntre[row];
      int64_t lo = b, hi = e;
      if (s == 0) hi = b;
      else if (s == slices) lo = e;
      const int64_t target = (int64_t)s * width;
      while (lo < hi) {
         const int64_t mid = (lo + hi) >> 1;
         if (indx[mid] < target) lo = mid + 1; else hi = mid;
      }
      sliceptr[t] = lo;
   }
}

// flag = 1 if any row has a descending column pair (the slice table is then meaningless)
__global__ __launch_bounds__(256) void sorted_check_kernel(int64_t m, const int64_t *__restrict__ pntrb,
                                                           const int64_t *__restrict__ pntre,
                                                           const int64_t *__restrict__ indx, int *flag) {
   const int lane = threadIdx.x & 63;
   const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
   bool bad = false;
   for (int64_t row = wave; row < m; row += nwaves) {
      const int64_t b = pntrb[row], e = pntre[row];
      for (int64_t p = b + 1 + lane; p < e; p += 64) bad |= indx[p] < indx[p - 1];
   }
   if (bad) atomicOr(flag, 1);
}

// ---- task plan of the task-list SpMM schedule (fusedMM_csr_tasks_hip) ------------------------
// Segments are listed slice-major (slice s holds positions [s*m, (s+1)*m)); the eight XCD lanes own contiguous
// runs of that list cut at equal edge mass (plan_lanes_kernel), so any slice count >= 1 works.  Rows with fewer
// than short_row edges are not sliced: their whole row is one segment homed on slice row % slices.
__device__ __forceinline__ void plan_segment(int64_t row, int sp, int slices, int short_row,
                                             const int64_t *__restrict__ pntrb, const int64_t *__restrict__ pntre,
                                             const int64_t *__restrict__ sliceptr, int64_t &b, int64_t &e) {
   const int s = sp;
   const int64_t rb = pntrb[row], re = pntre[row];
   if (re - rb < short_row) {
      b = rb;
      e = (int)(row % slices) == s ? re : rb;
   } else {
      const int64_t *t = sliceptr + (size_t)row * (size_t)(slices + 1) + s;
      b = t[0];
      e = t[1];
   }
}

__global__ __launch_bounds__(256) void plan_count_kernel(int64_t m, int slices, int chunk, int short_row,
                                                         const int64_t *__restrict__ pntrb,
                                                         const int64_t *__restrict__ pntre,
                                                         const int64_t *__restrict__ sliceptr, int *__restrict__ cnt,
                                                         int64_t *__restrict__ ecnt) {
   const int64_t total = m * slices;
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= total; i += stride) {
      if (i == total) { cnt[i] = 0; ecnt[i] = 0; continue; }
      int64_t b, e;
      plan_segment(i % m, (int)(i / m), slices, short_row, pntrb, pntre, sliceptr, b, e);
      cnt[i] = (int)((e - b + chunk - 1) / chunk);
      ecnt[i] = e - b;
   }
}

// lane boundaries by EDGE mass: lane x starts at the first segment whose edge prefix reaches x/8 of all edges
// (a skewed graph -- R-MAT's low column ids carry most edges -- otherwise leaves most XCDs idle at the end)
__global__ void plan_lanes_kernel(int64_t items, const int64_t *__restrict__ eoff, const int *__restrict__ seg_off,
                                  int *__restrict__ lanes) {
   const int x = threadIdx.x;
   if (x > 8) return;
   const int64_t total = eoff[items - 1];
   const int64_t target = x == 8 ? total + 1 : (total / 8) * x;
   int64_t lo = 0, hi = items - 1;           // first segment index whose exclusive edge prefix is >= target
   while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (eoff[mid] < target) lo = mid + 1; else hi = mid;
   }
   lanes[x] = x == 0 ? 0 : seg_off[x == 8 ? items - 1 : lo];
}

__global__ __launch_bounds__(256) void plan_fill_kernel(int64_t m, int slices, int chunk, int short_row,
                                                        const int64_t *__restrict__ pntrb,
                                                        const int64_t *__restrict__ pntre,
                                                        const int64_t *__restrict__ sliceptr,
                                                        const int *__restrict__ seg_off, int *__restrict__ task_row,
                                                        int64_t *__restrict__ task_b, int *__restrict__ task_len) {
   const int64_t total = m * slices;
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int t0 = seg_off[i], t1 = seg_off[i + 1];
      if (t1 == t0) continue;
      int64_t b, e;
      const int64_t row = i % m;
      plan_segment(row, (int)(i / m), slices, short_row, pntrb, pntre, sliceptr, b, e);
      for (int t = t0; t < t1; t++) {
         const int64_t cb = b + (int64_t)(t - t0) * chunk;
         task_row[t] = (int)row;
         task_b[t] = cb;
         task_len[t] = (int)((e - cb) < chunk ? (e - cb) : chunk);
      }
   }
}

static hipError_t plan_scan_temp_bytes(int64_t items, size_t *bytes) {
   size_t a = 0, b = 0;
   hipError_t e = scan_exclusive_i32(nullptr, a, nullptr, nullptr, (size_t)items, (hipStream_t)0);
   if (e != hipSuccess) return e;
   e = scan_exclusive_i64(nullptr, b, nullptr, nullptr, (size_t)items, (hipStream_t)0);
   *bytes = a > b ? a : b;
   return e;
}

static inline unsigned grid_for(int64_t n) {
   int64_t b = (n + 255) / 256;
   if (b < 1) b = 1;
   if (b > 256 * 16) b = 256 * 16;
   return (unsigned)b;
}

static inline unsigned key_bits(int64_t n) {
   unsigned bits = 1;
   while (bits < 32 && ((int64_t)1 << bits) < n) bits++;
   return bits;
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static hipError_t sort_temp_bytes(int64_t n, int64_t nnz, size_t *bytes) {
   *bytes = 0;
   return sort_pairs_u32(nullptr, *bytes, nullptr, nullptr, nullptr, nullptr, (size_t)nnz, 0u, key_bits(n), (hipStream_t)0);
}

}  // namespace isplib

using namespace isplib;

extern "C" int isplib_csr_row_ids_hip(int64_t m, int64_t nnz, const int64_t *rowptr, int64_t *row, void *stream) {
   clear_error();
   if (m < 0 || nnz < 0) return fail(ISPLIB_FAIL, "isplib_csr_row_ids_hip: negative dimension");
   if (nnz == 0) return ISPLIB_SUCCESS;
   if (!rowptr || !row || m == 0) return fail(ISPLIB_FAIL, "isplib_csr_row_ids_hip: null operand");
   hipLaunchKernelGGL(row_ids_kernel, dim3(grid_for(nnz)), dim3(256), 0, (hipStream_t)stream, m, nnz, rowptr, row);
   return check_launch("row_ids_kernel");
}

extern "C" size_t isplib_csr2csc_workspace_bytes(int64_t m, int64_t n, int64_t nnz) {
   (void)m;
   if (nnz <= 0) return 256;
   size_t temp = 0;
   if (sort_temp_bytes(n, nnz, &temp) != hipSuccess) return 0;
   return 4 * align_up((size_t)nnz * sizeof(uint32_t)) + align_up(temp) + 256;
}

extern "C" int isplib_csr2csc_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                  const float *val, int mean_scale, int64_t *colptr, int64_t *csr2csc,
                                  int64_t *row_t, float *val_t, void *workspace, size_t workspace_bytes,
                                  void *stream) {
   clear_error();
   if (m < 0 || n < 0 || nnz < 0) return fail(ISPLIB_FAIL, "isplib_csr2csc_hip: negative dimension");
   if (n > 0x7fffffffLL || nnz > 0xffffffffLL) return fail(ISPLIB_FAIL, "isplib_csr2csc_hip: n < 2^31 and nnz < 2^32 required");
   if (!colptr) return fail(ISPLIB_FAIL, "isplib_csr2csc_hip: colptr is required");
   hipStream_t st = (hipStream_t)stream;
   if (nnz == 0) {
      ISPLIB_HIP_TRY(hipMemsetAsync(colptr, 0, (size_t)(n + 1) * sizeof(int64_t), st));
      return ISPLIB_SUCCESS;
   }
   if (!rowptr || !col || !workspace) return fail(ISPLIB_FAIL, "isplib_csr2csc_hip: null operand");
   size_t temp = 0;
   ISPLIB_HIP_TRY(sort_temp_bytes(n, nnz, &temp));
   const size_t arr = align_up((size_t)nnz * sizeof(uint32_t));
   if (workspace_bytes < 4 * arr + align_up(temp)) return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_csr2csc_hip: workspace too small");
   char *w = (char *)workspace;
   uint32_t *keys_in = (uint32_t *)w;
   uint32_t *keys_out = (uint32_t *)(w + arr);
   uint32_t *pos_in = (uint32_t *)(w + 2 * arr);
   uint32_t *pos_out = (uint32_t *)(w + 3 * arr);
   void *tmp = (void *)(w + 4 * arr);

   hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(nnz)), dim3(256), 0, st, nnz, col, keys_in, pos_in);
   int rc = check_launch("make_keys_kernel");
   if (rc) return rc;
   ISPLIB_HIP_TRY(sort_pairs_u32(tmp, temp, keys_in, keys_out, pos_in, pos_out, (size_t)nnz, 0u, key_bits(n), st));
   hipLaunchKernelGGL(colptr_kernel, dim3(grid_for(n + 1)), dim3(256), 0, st, n, nnz, keys_out, colptr);
   rc = check_launch("colptr_kernel");
   if (rc) return rc;
   if (csr2csc || row_t || val_t) {
      hipLaunchKernelGGL(finalize_kernel, dim3(grid_for(nnz)), dim3(256), 0, st, m, nnz, rowptr, val, mean_scale,
                         pos_out, csr2csc, row_t, val_t);
      rc = check_launch("finalize_kernel");
   }
   return rc;
}

extern "C" size_t isplib_spmm_slices_bytes(int64_t m, int slices) {
   if (m < 0 || slices < 1) return 0;
   return (size_t)m * (size_t)(slices + 1) * sizeof(int64_t);
}

extern "C" int isplib_spmm_slices_build_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *pntrb,
                                            const int64_t *pntre, const int64_t *indx, int slices, int64_t *sliceptr,
                                            int32_t *unsorted_flag, void *stream) {
   clear_error();
   if (m < 0 || n < 0 || nnz < 0) return fail(ISPLIB_FAIL, "isplib_spmm_slices_build_hip: negative dimension");
   if (slices < 1 || slices > ISPLIB_MAX_SLICES) return fail(ISPLIB_FAIL, "isplib_spmm_slices_build_hip: slices must be in [1, 4096]");
   hipStream_t st = (hipStream_t)stream;
   if (unsorted_flag) ISPLIB_HIP_TRY(hipMemsetAsync(unsorted_flag, 0, sizeof(int32_t), st));
   if (m == 0) return ISPLIB_SUCCESS;
   if (!pntrb || !pntre || !sliceptr || (nnz > 0 && !indx)) return fail(ISPLIB_FAIL, "isplib_spmm_slices_build_hip: null operand");
   const int64_t width = (n + slices - 1) / slices > 0 ? (n + slices - 1) / slices : 1;
   hipLaunchKernelGGL(slices_kernel, dim3(grid_for(m * (slices + 1))), dim3(256), 0, st, m, width, slices, pntrb, pntre,
                      indx, sliceptr);
   int rc = check_launch("slices_kernel");
   if (rc) return rc;
   if (unsorted_flag && nnz > 0) {
      hipLaunchKernelGGL(sorted_check_kernel, dim3(grid_for(m * 64)), dim3(256), 0, st, m, pntrb, pntre, indx,
                         (int *)unsorted_flag);
      rc = check_launch("sorted_check_kernel");
   }
   return rc;
}

extern "C" size_t isplib_spmm_tasks_plan_workspace_bytes(int64_t m, int slices) {
   if (m <= 0 || slices <= 0) return 256;
   const int64_t items = m * slices + 1;
   size_t temp = 0;
   if (plan_scan_temp_bytes(items, &temp) != hipSuccess) return 0;
   return align_up((size_t)items * sizeof(int)) + 2 * align_up((size_t)items * sizeof(int64_t)) + align_up(temp) + 512;
}

extern "C" int isplib_spmm_tasks_count_hip(int64_t m, const int64_t *pntrb, const int64_t *pntre,
                                           const int64_t *sliceptr, int slices, int chunk, int short_row,
                                           int32_t *seg_off, void *workspace, size_t workspace_bytes,
                                           isplib_task_plan_info *info, void *stream) {
   clear_error();
   if (m < 0) return fail(ISPLIB_FAIL, "isplib_spmm_tasks_count_hip: negative dimension");
   if (slices < 1 || slices > ISPLIB_MAX_SLICES) return fail(ISPLIB_FAIL, "isplib_spmm_tasks_count_hip: slices must be in [1, 4096]");
   if (chunk < 64 || short_row < 0) return fail(ISPLIB_FAIL, "isplib_spmm_tasks_count_hip: chunk >= 64 and short_row >= 0 required");
   if (!info || !seg_off) return fail(ISPLIB_FAIL, "isplib_spmm_tasks_count_hip: null operand");
   if (m * slices + 1 > 0x7fffffffLL) return fail(ISPLIB_FAIL, "isplib_spmm_tasks_count_hip: m*slices must be < 2^31");
   hipStream_t st = (hipStream_t)stream;
   memset(info, 0, sizeof(*info));
   info->slices = slices; info->chunk = chunk; info->short_row = short_row;
   if (m == 0) {
      ISPLIB_HIP_TRY(hipMemsetAsync(seg_off, 0, sizeof(int32_t), st));
      return ISPLIB_SUCCESS;
   }
   if (!pntrb || !pntre || !sliceptr || !workspace) return fail(ISPLIB_FAIL, "isplib_spmm_tasks_count_hip: null operand");
   const int64_t items = m * slices + 1;
   size_t temp = 0;
   ISPLIB_HIP_TRY(plan_scan_temp_bytes(items, &temp));
   const size_t cnt_bytes = align_up((size_t)items * sizeof(int)), e_bytes = align_up((size_t)items * sizeof(int64_t));
   if (workspace_bytes < cnt_bytes + 2 * e_bytes + align_up(temp) + 256)
      return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_spmm_tasks_count_hip: workspace too small");
   char *w = (char *)workspace;
   int *cnt = (int *)w;
   int64_t *ecnt = (int64_t *)(w + cnt_bytes), *eoff = (int64_t *)(w + cnt_bytes + e_bytes);
   void *tmp = w + cnt_bytes + 2 * e_bytes;
   int *lanes = (int *)(w + cnt_bytes + 2 * e_bytes + align_up(temp));
   hipLaunchKernelGGL(plan_count_kernel, dim3(grid_for(items)), dim3(256), 0, st, m, slices, chunk, short_row, pntrb,
                      pntre, sliceptr, cnt, ecnt);
   int rc = check_launch("plan_count_kernel");
   if (rc) return rc;
   ISPLIB_HIP_TRY(scan_exclusive_i32(tmp, temp, (const int *)cnt, (int *)seg_off, (size_t)items, st));
   ISPLIB_HIP_TRY(scan_exclusive_i64(tmp, temp, (const int64_t *)ecnt, eoff, (size_t)items, st));
   hipLaunchKernelGGL(plan_lanes_kernel, dim3(1), dim3(64), 0, st, items, eoff, seg_off, lanes);
   rc = check_launch("plan_lanes_kernel");
   if (rc) return rc;
   // the one host round trip of the plan: the task count and the eight (edge-balanced) lane boundaries
   int host[9];
   ISPLIB_HIP_TRY(hipMemcpyAsync(host, lanes, 9 * sizeof(int), hipMemcpyDeviceToHost, st));
   ISPLIB_HIP_TRY(hipStreamSynchronize(st));
   for (int x = 0; x < 9; x++) info->lane_off[x] = host[x];
   info->n_tasks = host[8];
   return ISPLIB_SUCCESS;
}

extern "C" int isplib_spmm_tasks_fill_hip(int64_t m, const int64_t *pntrb, const int64_t *pntre,
                                          const int64_t *sliceptr, const isplib_task_plan_info *info,
                                          const int32_t *seg_off, int32_t *task_row, int64_t *task_b,
                                          int32_t *task_len, void *stream) {
   clear_error();
   if (!info) return fail(ISPLIB_FAIL, "isplib_spmm_tasks_fill_hip: null plan info");
   if (m <= 0 || info->n_tasks == 0) return ISPLIB_SUCCESS;
   if (!pntrb || !pntre || !sliceptr || !seg_off || !task_row || !task_b || !task_len)
      return fail(ISPLIB_FAIL, "isplib_spmm_tasks_fill_hip: null operand");
   hipLaunchKernelGGL(plan_fill_kernel, dim3(grid_for(m * info->slices)), dim3(256), 0, (hipStream_t)stream, m,
                      info->slices, info->chunk, info->short_row, pntrb, pntre, sliceptr, seg_off, task_row, task_b,
                      task_len);
   return check_launch("plan_fill_kernel");
}

// ---- 32-bit copy of the column ids for the task kernels (include/isplib_hip.h) -------------------
namespace isplib {
__global__ __launch_bounds__(256) void pack_indices_kernel(int64_t nnz, const int64_t *__restrict__ indx,
                                                           int32_t *__restrict__ indx32) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) indx32[i] = (int32_t)indx[i];
}
}  // namespace isplib

extern "C" int isplib_pack_indices_hip(int64_t nnz, const int64_t *indx, int32_t *indx32, void *stream) {
   clear_error();
   if (nnz < 0) return fail(ISPLIB_FAIL, "isplib_pack_indices_hip: negative nnz");
   if (nnz == 0) return ISPLIB_SUCCESS;
   if (!indx || !indx32) return fail(ISPLIB_FAIL, "isplib_pack_indices_hip: null operand");
   hipLaunchKernelGGL(pack_indices_kernel, dim3(grid_for(nnz)), dim3(256), 0, (hipStream_t)stream, nnz, indx, indx32);
   return check_launch("pack_indices_kernel");
}
