// graph_runtime.hip -- isplib_graph: a handle that owns everything a graph needs for the fast path.
//
// The stateless entries (fusedMM_csr_hip, fusedMM_csr_tasks_hip, ...) leave allocation and caching to the
// caller; isplib_amd/plan.py + sparse.py do that for torch hosts.  This file is the same thing for hosts
// without torch (the reference's C++ layer, csrc/fusedmm.cpp:113-203, would keep one handle per graph where
// isplib/__init__.py:35-40 keeps its pointer-keyed dicts): per-graph operands built lazily on the device,
// cached per slice count, one grow-only workspace, the schedule picked by the same measured rule.
//   forward   out = A (x) y            isplib_graph_spmm
//   backward  dX  = A^T dY             isplib_graph_spmm_backward   (sum, or mean with val/deg weights,
//                                                                    csrc/fusedmm.cpp:285,375)
// First use of a (side, slice count) allocates and synchronises the stream once; after that every call is
// asynchronous and allocation-free.  A handle is not thread-safe (serialise the CALLS; the operator library does so
// with a mutex), but it may be used on several streams: each stream gets its own workspace.
#include <map>
#include <new>

#include "common.h"


using namespace isplib;

namespace {

struct Plan {
   bool usable = false;          // false: rows not column-sorted (or too large for 32-bit task ids) -> plain kernel
   int64_t n_tasks = 0;
   int32_t *task_row = nullptr, *task_len = nullptr, *seg_off = nullptr;
   int64_t *task_b = nullptr;
   int64_t lane_off[9] = {0};
};

struct Side {                    // A, or A^T, as CSR
   int64_t m = 0, n = 0, nnz = 0;
   const int64_t *rowptr = nullptr, *col = nullptr;
   const float *val = nullptr;
   double cv2 = -1.0;            // squared coefficient of variation of the row degrees (-1: not measured yet)
   int unit = -1;                // val examined once: 1 = every weight is exactly 1.0f (isplib/__init__.py:51-57
                                 // materialises unit weights as a ones vector) -> the kernels skip the value stream
   int32_t *col32 = nullptr;
   std::map<int, Plan> plans;
   // stream plans (sum / mean) per (streams, slices, chunk); the weights they were last given (the plans own a copy in
   // stream order: `val` of the side, or the mean backward's weights)
   // `other`: a second copy of the weights in stream order, for the side that serves two kinds of them in turn (A^T: the
   // sum backward's val[csr2csc] and the mean backward's val[csr2csc] / deg): a model that mixes sum and mean aggregation
   // on one weighted graph swaps the two copies instead of re-gathering nnz weights at every backward
   struct Stream {
      isplib_stream_plan plan;
      const float *vals_of = nullptr; bool has_vals = false; uint64_t gen = 0;
      float *other = nullptr; const float *other_of = nullptr; uint64_t other_gen = 0;
   };
   std::map<uint64_t, Stream> streams;
   // plain-kernel rows in a community order (reorder.hip), for dense operands larger than the Infinity Cache: found on
   // first need (order_state 0 -> 1: tried and kept in `order`, or 2: tried, no structure found / not applicable), or
   // given by the caller (isplib_graph_set_row_order: state 3, borrowed, never freed here)
   int32_t *order = nullptr;
   int order_state = 0;
   bool stream_refused = false;          // the sum / mean stream builder declined this side (outside its domain): task list / plain
   bool minmax_stream_refused = false;   // the max / min stream builder declined this side (rows not column-sorted): task list
};

__global__ __launch_bounds__(256) void not_all_ones_kernel(int64_t nnz, const float *__restrict__ val, int *__restrict__ flag) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   bool other = false;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) other |= val[i] != 1.0f;
   if (other) atomicOr(flag, 1);
}

// sum of the row degrees' squares (double), for the degree-skew test of the slice rule
__global__ __launch_bounds__(256) void degree_squares_kernel(int64_t m, const int64_t *__restrict__ rowptr, double *__restrict__ out) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   double acc = 0.0;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
      const double d = (double)(rowptr[i + 1] - rowptr[i]);
      acc += d * d;
   }
   for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
   if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

void free_side(Side &s, bool owns_arrays) {
   for (auto &kv : s.plans) {
      (void)hipFree(kv.second.task_row); (void)hipFree(kv.second.task_len);
      (void)hipFree(kv.second.seg_off); (void)hipFree(kv.second.task_b);
   }
   s.plans.clear();
   for (auto &kv : s.streams) { isplib_stream_plan_free(&kv.second.plan); (void)hipFree(kv.second.other); }
   s.streams.clear();
   (void)hipFree(s.col32);
   if (s.order_state == 1) (void)hipFree(s.order);
   s.col32 = nullptr;
   if (owns_arrays) {
      (void)hipFree(const_cast<int64_t *>(s.rowptr)); (void)hipFree(const_cast<int64_t *>(s.col));
      (void)hipFree(const_cast<float *>(s.val));
   }
}

}  // namespace

struct isplib_graph {
   Side fwd, bwd;                // bwd = A^T, built on first backward call
   bool has_bwd = false;
   float *mean_val_t = nullptr;  // val[csr2csc] / max(deg(row),1): the mean backward's weights
   float *scaled = nullptr;      // grow-only [m][k] copy of dy with rows scaled by 1 / max(deg,1): the mean backward of a
   size_t scaled_bytes = 0;      // unit-weight graph needs no edge weights (A^T diag(1/deg) dY = A^T (diag(1/deg) dY))
   uint64_t val_gen = 0;         // bumped by isplib_graph_set_values: copies of the weights older than this are stale
   bool bwd_vals_stale = false;  // the transposed weights (bwd.val, mean_val_t) predate the last isplib_graph_set_values
   struct Work { void *ptr = nullptr; size_t bytes = 0; };
   std::map<hipStream_t, Work> works;   // one grow-only workspace per stream the handle has been used on
   int forced_slices = -1;       // -1: isplib_suggest_slices
   bool order_given = false;     // isplib_graph_set_row_order was called: order_t_given is A^T's order when that side is built
   const int32_t *order_t_given = nullptr;
};

#define TRY_ALLOC(ptr, bytes)                                                                              \
   do {                                                                                                    \
      if (hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 256) != hipSuccess) {                             \
         (void)hipGetLastError();                                                                          \
         return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_graph: device allocation failed");                     \
      }                                                                                                    \
   } while (0)

extern "C" int isplib_suggest_slices(int64_t m, int64_t n, int64_t nnz, int64_t k, int minmax) {
   // Measured on MI355X: about 7 MB of the dense operand per slice (2x an XCD's L2),
   // never fewer than k/20 slices on a graph with work for the whole chip, at least ~20 edges per row and
   // slice; wide k is swept in column panels (64 wide when rows are whole cache lines, else 128) and counts
   // as the panel width.  0 = plain row-per-wave kernel.
   if (m <= 0 || n <= 0 || k <= 0) return 0;
   (void)minmax;             // both families run in the same panels (isplib_hip_tune keys 4 / 5)
   if (k % 32 == 0) { if (k >= 96) k = 64; }          // rows of whole cache lines: 64-column panels
   else if (k >= 192) k = 128;                        // ragged rows: 128-column panels
   const double avg_deg = (double)nnz / (double)m;
   if (nnz < (1 << 20) || avg_deg < 64.0) return 0;
   const double by_cache = (double)n * (double)k * 4.0 / (double)(7 << 20);
   double s = by_cache > (double)k / 20.0 ? by_cache : (double)k / 20.0;
   if (avg_deg / 20.0 < s) s = avg_deg / 20.0;
   int r = (int)(s + 0.5);
   return r < 1 ? 1 : (r > 64 ? 64 : r);
}

extern "C" int isplib_suggest_slices_whole_rows(int64_t m, int64_t n, int64_t nnz, int64_t k) {
   // The same rule for kernels that need every column of a row per edge (SDDMM's dot product, the generic
   // pipeline's ROP) and therefore cannot run in column panels: slices sized by the full row width.
   if (m <= 0 || n <= 0 || k <= 0) return 0;
   const double avg_deg = (double)nnz / (double)m;
   if (nnz < (1 << 20) || avg_deg < 64.0) return 0;
   const double by_cache = (double)n * (double)k * 4.0 / (double)(7 << 20);
   const double floor_k = (double)(k < 128 ? k : 128) / 20.0;
   double s = by_cache > floor_k ? by_cache : floor_k;
   if (avg_deg / 20.0 < s) s = avg_deg / 20.0;
   const int r = (int)(s + 0.5);
   return r < 1 ? 1 : (r > 64 ? 64 : r);
}

extern "C" int isplib_graph_create(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                   const float *val, isplib_graph **out) {
   clear_error();
   if (!out) return fail(ISPLIB_FAIL, "isplib_graph_create: out is NULL");
   *out = nullptr;
   if (m < 0 || n < 0 || nnz < 0 || n > 0x7fffffffLL) return fail(ISPLIB_FAIL, "isplib_graph_create: bad dimension (n must be < 2^31)");
   if (!rowptr || (nnz > 0 && !col)) return fail(ISPLIB_FAIL, "isplib_graph_create: null operand");
   isplib_graph *g = new (std::nothrow) isplib_graph();
   if (!g) return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_graph_create: host allocation failed");
   g->fwd.m = m; g->fwd.n = n; g->fwd.nnz = nnz; g->fwd.rowptr = rowptr; g->fwd.col = col; g->fwd.val = val;
   *out = g;
   return ISPLIB_SUCCESS;
}

extern "C" void isplib_graph_destroy(isplib_graph *g) {
   if (!g) return;
   (void)hipDeviceSynchronize();           // nothing of ours may still be in flight
   free_side(g->fwd, false);
   free_side(g->bwd, true);
   (void)hipFree(g->mean_val_t);
   (void)hipFree(g->scaled);
   for (auto &kv : g->works) (void)hipFree(kv.second.ptr);
   delete g;
}

extern "C" int isplib_graph_set_slices(isplib_graph *g, int slices) {
   clear_error();
   if (!g) return fail(ISPLIB_FAIL, "isplib_graph_set_slices: null handle");
   if (slices < -1 || slices > ISPLIB_MAX_SLICES) return fail(ISPLIB_FAIL, "isplib_graph_set_slices: -1 (rule), 0 (plain) or 1..4096");
   g->forced_slices = slices;
   return ISPLIB_SUCCESS;
}

// A caller-given row order must be a permutation of [0, rows): an entry out of range would be an out-of-bounds row read and
// write in the plain kernel, a repeated one a row computed twice and another never.  Checked once, on the device.
__global__ __launch_bounds__(256) void order_check_kernel(int64_t rows, const int32_t *__restrict__ order, int *__restrict__ seen, int *__restrict__ bad) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
      const int r = order[i];
      if (r < 0 || (int64_t)r >= rows) atomicOr(bad, 1);
      else if (atomicExch(&seen[r], 1) != 0) atomicOr(bad, 2);
   }
}

static int check_row_order(int64_t rows, const int32_t *order) {
   if (!order || rows <= 0) return ISPLIB_SUCCESS;
   int *seen = nullptr, host = 0;
   if (hipMalloc((void **)&seen, ((size_t)rows + 1) * sizeof(int)) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_graph_set_row_order: device allocation failed");
   }
   // The order arrays carry no stream: whatever produced them (a kernel on a non-blocking side stream, say) must be complete
   // before they are read here, and the null stream does not wait for such streams -- so the device is drained first.  This
   // is a once-per-graph call that already allocates and copies back; one more synchronisation costs it nothing.
   hipError_t err = hipDeviceSynchronize();
   if (err == hipSuccess) err = hipMemset(seen, 0, ((size_t)rows + 1) * sizeof(int));
   if (err == hipSuccess) {
      int64_t blocks = (rows + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(order_check_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, rows, order, seen, seen + rows);
      err = hipGetLastError();                               // read ONCE: a second call would report success
      if (err == hipSuccess) err = hipMemcpy(&host, seen + rows, sizeof(int), hipMemcpyDeviceToHost);
   }
   (void)hipFree(seen);
   if (err != hipSuccess) return hip_fail(err, "isplib_graph_set_row_order: checking the order");
   if (host) return fail(ISPLIB_FAIL, host & 1 ? "isplib_graph_set_row_order: an entry of the order is outside [0, rows)"
                                               : "isplib_graph_set_row_order: the order is not a permutation (a row appears twice)");
   return ISPLIB_SUCCESS;
}

extern "C" int isplib_graph_set_row_order(isplib_graph *g, const int32_t *order, const int32_t *order_t) {
   // The order the plain kernel takes the rows of A (order) / of A^T (order_t) in: borrowed device arrays of m / n int32,
   // position -> row; NULL = index order and no search for one.  Speed only.  Each given order is checked once, here, to
   // be a permutation (one pass on the device, synchronous); a bad one is refused and nothing changes.
   clear_error();
   if (!g) return fail(ISPLIB_FAIL, "isplib_graph_set_row_order: null handle");
   Side *sides[2] = {&g->fwd, &g->bwd};
   const int32_t *given[2] = {order, order_t};
   for (int i = 0; i < 2; i++) {
      const int rc = check_row_order(sides[i]->m > 0 ? sides[i]->m : (i == 0 ? g->fwd.m : g->fwd.n), given[i]);
      if (rc) return rc;
   }
   for (int i = 0; i < 2; i++) {
      if (sides[i]->order_state == 1) (void)hipFree(sides[i]->order);
      sides[i]->order = const_cast<int32_t *>(given[i]);
      sides[i]->order_state = 3;
   }
   g->order_t_given = order_t;
   g->order_given = true;
   return ISPLIB_SUCCESS;
}

// The plain kernel's row order of a side: the caller's, or -- square graphs whose dense operand is larger than the
// Infinity Cache -- the community order, looked for once and kept only if it found structure (at least a fifth of the
// stored entries, and twice the index order's share, within the ~1024 rows an XCD has in flight)
static const int32_t *side_row_order(isplib_graph *g, Side &s, int64_t k, hipStream_t st) {
   if (s.order_state != 0) return s.order;
   s.order_state = 2;
   if (s.m != s.n || s.m >= (1LL << 31) || s.nnz <= 0 || (double)s.n * (double)k * 4.0 <= 256.0 * 1048576.0) {
      if ((double)s.n * (double)k * 4.0 <= 256.0 * 1048576.0) s.order_state = 0;      // a wider call may still want one
      return nullptr;
   }
   const size_t ws_bytes = isplib_community_order_workspace_bytes(s.m, s.nnz);
   void *ws = nullptr;
   int32_t *order = nullptr;
   if (hipMalloc(&ws, ws_bytes) != hipSuccess || hipMalloc((void **)&order, (size_t)s.m * sizeof(int32_t)) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(ws); (void)hipFree(order);
      clear_error();
      return nullptr;
   }
   double before = 0.0, after = 0.0;
   int rc = isplib_community_order_hip(s.m, s.nnz, s.rowptr, s.col, 8, 0, order, nullptr, nullptr, ws, ws_bytes, st);
   if (!rc) rc = isplib_order_locality_hip(s.m, s.nnz, s.rowptr, s.col, nullptr, 1024, &before, ws, ws_bytes, st);
   if (!rc) rc = isplib_order_locality_hip(s.m, s.nnz, s.rowptr, s.col, order, 1024, &after, ws, ws_bytes, st);
   (void)hipFree(ws);
   if (rc || after < 0.2 || after < 2.0 * before) {
      (void)hipFree(order);
      clear_error();
      return nullptr;
   }
   s.order = order;
   s.order_state = 1;
   (void)g;
   return s.order;
}

extern "C" int isplib_graph_set_values(isplib_graph *g, const float *val) {
   // New weights for the same structure (an optimiser stepped them in place, or the caller passes another tensor):
   // everything keyed on (rowptr, col) -- task plans, stream plans' words, packed column ids, the CSC structure --
   // stays; what is derived from the VALUES is refreshed lazily by the next call that needs it, on that call's
   // stream: the unit-weight test, the stream plans' copies of the weights (one gather through the plan's
   // permutation), the transposed weights of the backward (the two csr2csc passes again, into the same arrays).
   clear_error();
   if (!g) return fail(ISPLIB_FAIL, "isplib_graph_set_values: null handle");
   g->fwd.val = val;
   g->fwd.unit = -1;
   g->val_gen++;
   if (g->has_bwd) {
      g->bwd.unit = -1;
      g->bwd_vals_stale = true;
   }
   return ISPLIB_SUCCESS;
}

static int ensure_work(isplib_graph *g, size_t bytes, hipStream_t st, isplib_graph::Work **out) {
   isplib_graph::Work &w = g->works[st];
   *out = &w;
   if (w.bytes >= bytes) return ISPLIB_SUCCESS;
   if (w.ptr) {
      ISPLIB_HIP_TRY(hipStreamSynchronize(st));          // an earlier call on this stream may still be reading the old one
      (void)hipFree(w.ptr);
      w.ptr = nullptr; w.bytes = 0;
   }
   TRY_ALLOC(w.ptr, bytes);
   w.bytes = bytes;
   return ISPLIB_SUCCESS;
}

// slice table -> task counts -> task arrays, all on the device; one stream synchronisation (inside count)
static int build_plan(isplib_graph *g, Side &s, int slices, hipStream_t st, Plan &p) {
   const int64_t m = s.m;
   if ((double)m * slices + (double)s.nnz / 1024.0 + 1.0 >= 2147483647.0) return ISPLIB_SUCCESS;   // int32 task ids
   int64_t *table = nullptr;
   int32_t *flag = nullptr;
   void *tmp = nullptr;
   TRY_ALLOC(table, isplib_spmm_slices_bytes(m, slices));
   int rc = ISPLIB_SUCCESS;
   int32_t unsorted = 0;
   isplib_task_plan_info info;
   const size_t tmp_bytes = isplib_spmm_tasks_plan_workspace_bytes(m, slices);
   do {
      if (hipMalloc((void **)&flag, 256) != hipSuccess || hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 256) != hipSuccess ||
          hipMalloc((void **)&p.seg_off, ((size_t)m * slices + 1) * sizeof(int32_t)) != hipSuccess) {
         (void)hipGetLastError();
         rc = fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_graph: device allocation failed");
         break;
      }
      rc = isplib_spmm_slices_build_hip(m, s.n, s.nnz, s.rowptr, s.rowptr + 1, s.col, slices, table, flag, st);
      if (rc) break;
      if (hipMemcpyAsync(&unsorted, flag, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess) { rc = hip_fail(hipGetLastError(), "isplib_graph: reading the sortedness flag"); break; }
      if (unsorted) break;                  // plan stays unusable: the plain kernel serves this graph
      rc = isplib_spmm_tasks_count_hip(m, s.rowptr, s.rowptr + 1, table, slices, 1024, 128, p.seg_off, tmp, tmp_bytes, &info, st);
      if (rc) break;
      p.n_tasks = info.n_tasks;
      for (int x = 0; x < 9; x++) p.lane_off[x] = info.lane_off[x];
      const size_t nt = (size_t)(p.n_tasks > 0 ? p.n_tasks : 1);
      if (hipMalloc((void **)&p.task_row, nt * sizeof(int32_t)) != hipSuccess || hipMalloc((void **)&p.task_len, nt * sizeof(int32_t)) != hipSuccess ||
          hipMalloc((void **)&p.task_b, nt * sizeof(int64_t)) != hipSuccess) {
         (void)hipGetLastError();
         rc = fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_graph: device allocation failed");
         break;
      }
      rc = isplib_spmm_tasks_fill_hip(m, s.rowptr, s.rowptr + 1, table, &info, p.seg_off, p.task_row, p.task_b, p.task_len, st);
      if (rc) break;
      if (!s.col32 && s.nnz > 0) {
         if (hipMalloc((void **)&s.col32, (size_t)s.nnz * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); s.col32 = nullptr; }
         else if ((rc = isplib_pack_indices_hip(s.nnz, s.col, s.col32, st)) != 0) break;
      }
      if (hipStreamSynchronize(st) != hipSuccess) { rc = hip_fail(hipGetLastError(), "isplib_graph: plan build"); break; }
      p.usable = true;
   } while (0);
   (void)hipFree(table); (void)hipFree(flag); (void)hipFree(tmp);
   (void)g;
   return rc;
}

// 1 if the side's own weights are all exactly 1.0f (then x * 1.0f == x bit for bit and the stream can be skipped)
static int weights_are_unit(Side &s, hipStream_t st) {
   if (s.unit >= 0) return s.unit;
   if (!s.val || s.nnz == 0) return s.unit = 0;
   int *flag = nullptr, host = 1;
   if (hipMalloc((void **)&flag, 256) != hipSuccess) { (void)hipGetLastError(); return 0; }     // undecided: try again later
   bool ok = hipMemsetAsync(flag, 0, sizeof(int), st) == hipSuccess;
   if (ok) {
      const int64_t blocks = (s.nnz + 255) / 256;
      hipLaunchKernelGGL(not_all_ones_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, s.nnz, s.val, flag);
      ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
           hipStreamSynchronize(st) == hipSuccess;
   }
   (void)hipFree(flag);
   if (!ok) { (void)hipGetLastError(); return 0; }
   return s.unit = host ? 0 : 1;
}

// The rule's 7 MB per slice leans on the popularity skew of real degree distributions (the hot rows of y stay in
// the L2).  A graph whose degrees hardly vary has no hot rows, and its slices must be closer to the L2 size:
// CV^2 of the row degrees under 0.25 -> half as many columns per slice again (uniform random graph of the Reddit
// size, K=128: 4.09 ms with 8 slices, 3.49 ms with 12).  Measured once per side (one small reduction + sync).
static int skew_adjusted(Side &s, int slices, hipStream_t st, int cap = 64) {
   if (slices <= 0 || s.m <= 0 || s.nnz <= 0) return slices;
   if (s.cv2 < 0.0) {
      double *acc = nullptr, host = 0.0;
      if (hipMalloc((void **)&acc, 256) != hipSuccess) { (void)hipGetLastError(); return slices; }
      bool ok = hipMemsetAsync(acc, 0, sizeof(double), st) == hipSuccess;
      if (ok) {
         const int64_t blocks = (s.m + 255) / 256;
         hipLaunchKernelGGL(degree_squares_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, s.m, s.rowptr, acc);
         ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&host, acc, sizeof(double), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipStreamSynchronize(st) == hipSuccess;
      }
      (void)hipFree(acc);
      if (!ok) { (void)hipGetLastError(); return slices; }
      const double mean = (double)s.nnz / (double)s.m;
      s.cv2 = host / (double)s.m / (mean * mean) - 1.0;
      if (s.cv2 < 0.0) s.cv2 = 0.0;
   }
   if (s.cv2 >= 0.25) return slices;
   const int more = (int)(1.5 * slices + 0.5);
   return more > cap ? cap : more;
}

// The side's sum / mean stream plan for width k, built on first use with the weights `val` (NULL: unit weights); *out stays
// null where the stream schedule does not serve the shape (the rule, a forced slice count, a refusal of the builder, no
// room for the plan): not an error, the caller's other schedules take the call.
static int side_stream_plan(isplib_graph *g, Side &s, const float *val, int64_t k, int64_t ldy, hipStream_t st, Side::Stream **out) {
   *out = nullptr;
   int st_streams = 0, st_slices = 0, st_chunk = 0;
   const bool y_in_one_descriptor = (double)s.n * (double)ldy * 4.0 <= 3.5 * 1073741824.0;      // with the caller's ldy, not k
   if (g->forced_slices >= 0 || s.stream_refused || ldy >= (1LL << 22) || !y_in_one_descriptor ||
       !isplib_suggest_stream_weighted(s.m, s.n, s.nnz, k, val != nullptr, &st_streams, &st_slices, &st_chunk))
      return ISPLIB_SUCCESS;
   st_slices = skew_adjusted(s, st_slices, st, 512);      // no degree skew: slices closer to the L2 size (31 -> 47: 3.21 -> 3.00 ms)
   const uint64_t key = ((uint64_t)st_streams << 48) | ((uint64_t)st_slices << 32) | (uint64_t)(uint32_t)st_chunk;
   auto it = s.streams.find(key);
   if (it == s.streams.end()) {
      Side::Stream fresh;
      const int rc = isplib_stream_plan_build_hip(s.m, s.n, s.nnz, s.rowptr, s.col, val, st_streams, st_slices, st_chunk, 0, &fresh.plan, st);
      if (rc == ISPLIB_SUCCESS) {
         fresh.vals_of = val; fresh.has_vals = val != nullptr; fresh.gen = g->val_gen;
         it = s.streams.emplace(key, fresh).first;
      } else if (rc == ISPLIB_FAIL) {
         s.stream_refused = true;                  // outside the builder's domain: not an error of this call; the
         clear_error();                            // task list / plain kernel serve the graph
         return ISPLIB_SUCCESS;
      } else if (rc != ISPLIB_NOT_ENOUGH_MEM) {
         return rc;
      } else {
         clear_error();                            // no room for the plan: the task list / plain kernel need less
         return ISPLIB_SUCCESS;
      }
   }
   *out = &it->second;
   return ISPLIB_SUCCESS;
}

static int run_side(isplib_graph *g, Side &s, const float *val, int32_t imessage, int64_t k, const float *y, int64_t ldy,
                    float *z, int64_t ldz, int64_t *z_arg, hipStream_t st) {
   if (val && val == s.val && weights_are_unit(s, st) == 1) val = nullptr;
   const int minmax = (imessage & 0xF0000) != ISPLIB_AOP_ADD;
   // sum / mean on graphs with work for the whole chip: the stream schedule (rows resident in LDS, the plan's own copy of
   // the edges), unless a slice count was forced
   const bool y_in_one_descriptor = (double)s.n * (double)ldy * 4.0 <= 3.5 * 1073741824.0;      // with the caller's ldy, not k
   if (!minmax) {
      Side::Stream *sp = nullptr;
      int rc = side_stream_plan(g, s, val, k, ldy, st, &sp);
      if (rc) return rc;
      if (sp) {
         if (sp->vals_of != val || sp->has_vals != (val != nullptr) || (val && sp->gen != g->val_gen)) {
            // other weights than last time (sum vs mean backward), or new contents (isplib_graph_set_values)
            if (val && sp->has_vals && sp->plan.vals) {
               // both weighted: the copy being replaced is parked in `other`; if `other` already holds what is wanted
               // (same source array, same generation) the two are swapped and nothing is gathered
               float *parked = const_cast<float *>(sp->plan.vals);
               const float *parked_of = sp->vals_of;
               const uint64_t parked_gen = sp->gen;
               const bool hit = sp->other && sp->other_of == val && sp->other_gen == g->val_gen;
               sp->plan.vals = sp->other;                  // NULL: set_values allocates a fresh array
               if (!hit) {
                  rc = isplib_stream_plan_set_values_hip(&sp->plan, val, st);
                  if (rc) {
                     // a fresh array may have been allocated before the gather failed: it is neither `parked` nor `other`
                     if (sp->plan.vals && sp->plan.vals != parked && sp->plan.vals != sp->other) (void)hipFree(const_cast<float *>(sp->plan.vals));
                     sp->plan.vals = parked;
                     return rc;
                  }
               }
               sp->other = parked; sp->other_of = parked_of; sp->other_gen = parked_gen;
            } else {
               rc = isplib_stream_plan_set_values_hip(&sp->plan, val, st);
               if (rc) return rc;
            }
            sp->vals_of = val; sp->has_vals = val != nullptr; sp->gen = g->val_gen;
         }
         const size_t need = isplib_spmm_stream_workspace_bytes(&sp->plan);
         isplib_graph::Work *w = nullptr;
         rc = ensure_work(g, need, st, &w);
         if (rc) return rc;
         return fusedMM_csr_stream_hip(imessage, s.m, s.n, k, s.nnz, s.rowptr, s.rowptr + 1, &sp->plan, y, ldy, z, ldz, w->ptr, w->bytes, nullptr, st);
      }
   }
   // max / min on such graphs: the stream schedule's own kernel and plan geometry, for column-sorted rows
   int mm_streams = 0, mm_slices = 0, mm_chunk = 0;
   // (the max / min entry admits dense operands under 2 GiB WITH THE CALLER'S ldy -- lanes past column k carry 2^31 in their
   // column term -- while the rule only sees k: a padded leading dimension that crosses it stays on the task list)
   const bool y_under_2gib = (double)s.n * (double)ldy * 4.0 < 2147483648.0;
   if (minmax && g->forced_slices < 0 && !s.minmax_stream_refused && ldy < (1LL << 22) && y_in_one_descriptor && y_under_2gib &&
       isplib_suggest_stream_minmax(s.m, s.n, s.nnz, k, &mm_streams, &mm_slices, &mm_chunk)) {
      mm_slices = skew_adjusted(s, mm_slices, st, 512);
      const uint64_t key = (1ULL << 63) | ((uint64_t)mm_streams << 48) | ((uint64_t)mm_slices << 32) | (uint64_t)(uint32_t)mm_chunk;
      auto it = s.streams.find(key);
      if (it == s.streams.end()) {
         Side::Stream fresh;
         const int rc = isplib_stream_plan_build_minmax_hip(s.m, s.n, s.nnz, s.rowptr, s.col, val, mm_streams, mm_slices, mm_chunk, 0, &fresh.plan, st);
         if (rc == ISPLIB_SUCCESS) {
            fresh.vals_of = val; fresh.has_vals = val != nullptr; fresh.gen = g->val_gen;
            it = s.streams.emplace(key, fresh).first;
         } else if (rc == ISPLIB_FAIL) {
            s.minmax_stream_refused = true;           // unsorted rows (or outside the builder's domain): not an error of this call
            clear_error();
         } else if (rc != ISPLIB_NOT_ENOUGH_MEM) {
            return rc;
         }
      }
      if (it != s.streams.end()) {
         Side::Stream &sp = it->second;
         if (sp.vals_of != val || sp.has_vals != (val != nullptr) || (val && sp.gen != g->val_gen)) {
            const int rc = isplib_stream_plan_set_values_hip(&sp.plan, val, st);
            if (rc) return rc;
            sp.vals_of = val; sp.has_vals = val != nullptr; sp.gen = g->val_gen;
         }
         const size_t need = isplib_spmm_stream_minmax_workspace_bytes(&sp.plan);
         isplib_graph::Work *w = nullptr;
         const int rc = ensure_work(g, need, st, &w);
         if (rc) return rc;
         return fusedMM_csr_stream_minmax_hip(imessage, s.m, s.n, k, s.nnz, s.rowptr, s.rowptr + 1, &sp.plan, y, ldy, z, ldz, z_arg, w->ptr,
                                              w->bytes, st);
      }
   }
   int slices = g->forced_slices >= 0 ? g->forced_slices : isplib_suggest_slices(s.m, s.n, s.nnz, k, minmax);
   if (k < 4 || (double)s.n * (double)ldy * 4.0 > 3.5 * 1073741824.0) slices = 0;      // outside the task entry's domain
   if (g->forced_slices < 0) slices = skew_adjusted(s, slices, st);
   if (slices > 0 && g->forced_slices < 0) {
      // The panel rule halves the slice count to make tasks long enough.  If they are long anyway (hub-dominated
      // graphs: >= 120 edges per task on the panel plan), the whole-row plan with one pass is the better schedule.
      const int whole = isplib_suggest_slices_whole_rows(s.m, s.n, s.nnz, k);
      if (whole > slices) {
         auto it = s.plans.find(slices);
         if (it == s.plans.end()) {
            Plan p;
            const int rc = build_plan(g, s, slices, st, p);
            if (rc) {
               (void)hipFree(p.task_row); (void)hipFree(p.task_len); (void)hipFree(p.seg_off); (void)hipFree(p.task_b);
               return rc;
            }
            it = s.plans.emplace(slices, p).first;
         }
         if (it->second.usable && it->second.n_tasks > 0 && (double)s.nnz / (double)it->second.n_tasks >= 120.0) slices = whole;
      }
   }
   if (slices > 0) {
      auto it = s.plans.find(slices);
      if (it == s.plans.end()) {
         Plan p;
         const int rc = build_plan(g, s, slices, st, p);
         if (rc) {
            (void)hipFree(p.task_row); (void)hipFree(p.task_len); (void)hipFree(p.seg_off); (void)hipFree(p.task_b);
            return rc;
         }
         it = s.plans.emplace(slices, p).first;
      }
      const Plan &p = it->second;
      if (p.usable) {
         const size_t need = isplib_spmm_tasks_workspace_bytes(imessage, p.n_tasks, k);
         isplib_graph::Work *w = nullptr;
         const int rc = ensure_work(g, need, st, &w);
         if (rc) return rc;
         return fusedMM_csr_tasks_hip(imessage, s.m, s.n, k, s.nnz, val, s.col, s.col32, s.rowptr, s.rowptr + 1, p.n_tasks,
                                      p.task_row, p.task_b, p.task_len, p.seg_off, slices, p.lane_off, y, ldy, z, ldz, z_arg,
                                      w->ptr, w->bytes, st);
      }
   }
   // the plain kernel; rows in a community order where the dense operand is beyond every cache and the graph has structure
   return fusedMM_csr_ordered_hip(imessage, s.m, s.n, k, s.nnz, val, s.col, s.rowptr, s.rowptr + 1, side_row_order(g, s, k, st), y, ldy,
                                  z, ldz, z_arg, st);
}

extern "C" int isplib_graph_spmm(isplib_graph *g, int32_t imessage, int64_t k, const float *y, int64_t ldy, float *z,
                                 int64_t ldz, int64_t *z_arg, void *stream) {
   clear_error();
   if (!g) return fail(ISPLIB_FAIL, "isplib_graph_spmm: null handle");
   return run_side(g, g->fwd, g->fwd.val, imessage, k, y, ldy, z, ldz, z_arg, (hipStream_t)stream);
}

// the weights of A^T again after isplib_graph_set_values: the same two stable sorts, into the arrays the handle has
static int refresh_transposed_values(isplib_graph *g, hipStream_t st) {
   const Side &a = g->fwd;
   Side &t = g->bwd;
   const size_t e = (size_t)(a.nnz > 0 ? a.nnz : 1);
   float *val_t = const_cast<float *>(t.val);
   if (a.val && !val_t) {
      TRY_ALLOC(val_t, e * sizeof(float));
      t.val = val_t;
   } else if (!a.val && val_t) {
      ISPLIB_HIP_TRY(hipStreamSynchronize(st));          // an earlier backward on this stream may still read it
      (void)hipFree(val_t);
      t.val = val_t = nullptr;
   }
   const size_t ws_bytes = isplib_csr2csc_workspace_bytes(a.m, a.n, a.nnz);
   isplib_graph::Work *w = nullptr;
   int rc = ensure_work(g, ws_bytes, st, &w);
   if (rc) return rc;
   int64_t *colptr = const_cast<int64_t *>(t.rowptr), *row_t = const_cast<int64_t *>(t.col);
   if (a.val) rc = isplib_csr2csc_hip(a.m, a.n, a.nnz, a.rowptr, a.col, a.val, 0, colptr, nullptr, row_t, val_t, w->ptr, w->bytes, st);
   if (!rc) rc = isplib_csr2csc_hip(a.m, a.n, a.nnz, a.rowptr, a.col, a.val, 1, colptr, nullptr, row_t, g->mean_val_t, w->ptr, w->bytes, st);
   if (rc) return rc;
   g->bwd_vals_stale = false;
   return ISPLIB_SUCCESS;
}

static int ensure_transpose(isplib_graph *g, hipStream_t st) {
   if (g->has_bwd) return g->bwd_vals_stale ? refresh_transposed_values(g, st) : ISPLIB_SUCCESS;
   const Side &a = g->fwd;
   Side t;
   t.m = a.n; t.n = a.m; t.nnz = a.nnz;
   int64_t *colptr = nullptr, *row_t = nullptr;
   float *val_t = nullptr;
   void *ws = nullptr;
   const size_t ws_bytes = isplib_csr2csc_workspace_bytes(a.m, a.n, a.nnz);
   int rc = ISPLIB_SUCCESS;
   const size_t e = (size_t)(a.nnz > 0 ? a.nnz : 1);
   if (hipMalloc((void **)&colptr, ((size_t)a.n + 1) * sizeof(int64_t)) != hipSuccess || hipMalloc((void **)&row_t, e * sizeof(int64_t)) != hipSuccess ||
       (a.val && hipMalloc((void **)&val_t, e * sizeof(float)) != hipSuccess) || hipMalloc((void **)&g->mean_val_t, e * sizeof(float)) != hipSuccess ||
       hipMalloc(&ws, ws_bytes ? ws_bytes : 256) != hipSuccess) {
      (void)hipGetLastError();
      rc = fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_graph: device allocation failed (transpose)");
   }
   // two passes of the same stable sort: the plain weights (sum backward), then val / max(deg,1) (mean backward)
   if (!rc && a.val) rc = isplib_csr2csc_hip(a.m, a.n, a.nnz, a.rowptr, a.col, a.val, 0, colptr, nullptr, row_t, val_t, ws, ws_bytes, st);
   if (!rc) rc = isplib_csr2csc_hip(a.m, a.n, a.nnz, a.rowptr, a.col, a.val, 1, colptr, nullptr, row_t, g->mean_val_t, ws, ws_bytes, st);
   if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = hip_fail(hipGetLastError(), "isplib_graph: transpose");
   (void)hipFree(ws);
   if (rc) {
      (void)hipFree(colptr); (void)hipFree(row_t); (void)hipFree(val_t); (void)hipFree(g->mean_val_t);
      g->mean_val_t = nullptr;
      return rc;
   }
   t.rowptr = colptr; t.col = row_t; t.val = val_t;
   g->bwd = t;
   if (g->order_given) { g->bwd.order = const_cast<int32_t *>(g->order_t_given); g->bwd.order_state = 3; }
   g->has_bwd = true;
   return ISPLIB_SUCCESS;
}

// out[i][:] = in[i][:] / max(deg_i, 1)
__global__ __launch_bounds__(256) void scale_rows_by_degree_kernel(int64_t m, int64_t k, const int64_t *__restrict__ rowptr,
                                                                   const float *__restrict__ in, int64_t ld_in, float *__restrict__ out) {
   const int64_t total = m * k, stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int64_t r = i / k, c = i - r * k;
      const int64_t d = rowptr[r + 1] - rowptr[r];
      out[i] = in[r * ld_in + c] / (float)(d > 1 ? d : 1);
   }
}

extern "C" int isplib_graph_spmm_backward(isplib_graph *g, int mean, int64_t k, const float *dy, int64_t lddy, float *dx,
                                          int64_t lddx, void *stream) {
   clear_error();
   if (!g) return fail(ISPLIB_FAIL, "isplib_graph_spmm_backward: null handle");
   hipStream_t st = (hipStream_t)stream;
   const int rc = ensure_transpose(g, st);
   if (rc) return rc;
   if (mean && g->fwd.m > 0 && k > 0 && (!g->fwd.val || weights_are_unit(g->fwd, st) == 1)) {
      // unit weights: scale the rows of dy once and run the SUM on A^T without edge weights (no weight stream)
      const size_t need = (size_t)g->fwd.m * (size_t)k * sizeof(float);
      if (g->scaled_bytes < need) {
         if (g->scaled) { ISPLIB_HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(g->scaled); g->scaled = nullptr; g->scaled_bytes = 0; }
         TRY_ALLOC(g->scaled, need);
         g->scaled_bytes = need;
      }
      const int64_t total = g->fwd.m * k, blocks = (total + 255) / 256;
      hipLaunchKernelGGL(scale_rows_by_degree_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, g->fwd.m, k,
                         g->fwd.rowptr, dy, lddy, g->scaled);
      const int rl = check_launch("scale_rows_by_degree_kernel");
      if (rl) return rl;
      return run_side(g, g->bwd, nullptr, ISPLIB_MSG_SPMM_SUM, k, g->scaled, k, dx, lddx, nullptr, st);
   }
   return run_side(g, g->bwd, mean ? g->mean_val_t : g->bwd.val, ISPLIB_MSG_SPMM_SUM, k, dy, lddy, dx, lddx, nullptr, st);
}

// dA[e] = <y[col[e], :], g[row(e), :]> (/ max(deg,1) for mean): the SDDMM the reference leaves commented out
// (csrc/fusedmm.cpp:270,351), over a task plan whose slice count is chosen for WHOLE rows of y
extern "C" int isplib_graph_sddmm(isplib_graph *g, int mean, int64_t k, const float *y, int64_t ldy, const float *gm,
                                  int64_t ldg, float *dval, void *stream) {
   clear_error();
   if (!g) return fail(ISPLIB_FAIL, "isplib_graph_sddmm: null handle");
   hipStream_t st = (hipStream_t)stream;
   Side &s = g->fwd;
   // (The forward's stream plan could serve dA too -- isplib_sddmm_stream_hip, now in the experimental library: built,
   // parity-tested, and slower than the task list below on every shape measured: Reddit shape K=128 4.04 ms against 3.54.)
   int slices = g->forced_slices >= 0 ? g->forced_slices : isplib_suggest_slices_whole_rows(s.m, s.n, s.nnz, k);
   if (k < 4 || k > 1024 || (double)s.n * (double)ldy * 4.0 > 3.5 * 1073741824.0) slices = 0;
   if (slices > 0) {
      auto it = s.plans.find(slices);
      if (it == s.plans.end()) {
         Plan p;
         const int rc = build_plan(g, s, slices, st, p);
         if (rc) {
            (void)hipFree(p.task_row); (void)hipFree(p.task_len); (void)hipFree(p.seg_off); (void)hipFree(p.task_b);
            return rc;
         }
         it = s.plans.emplace(slices, p).first;
      }
      const Plan &p = it->second;
      if (p.usable)
         return isplib_sddmm_csr_tasks_hip(s.m, s.n, k, s.col, s.col32, s.rowptr, s.rowptr + 1, p.n_tasks, p.task_row, p.task_b,
                                           p.task_len, p.lane_off, y, ldy, gm, ldg, mean, dval, st);
   }
   return isplib_sddmm_csr_hip(s.m, k, s.col, s.rowptr, s.rowptr + 1, y, ldy, gm, ldg, mean, dval, st);
}
