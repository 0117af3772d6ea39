// torch_ops.cpp -- torch.ops.isplib.* for device tensors, on top of the C ABI.
//
// Mirrors the reference's operator + autograd layer (csrc/fusedmm.cpp:206-570):
// same op names and argument lists, same saved-for-backward operands, same
// backward formulas -- with the launcher (csrc/fusedmm.cpp:113-203) replaced by
// fusedMM_csr_hip and the ATen scatter chain of max/min backward replaced by
// one fused kernel.  Host-side only: no kernels here, torch is used for memory,
// the current stream and autograd bookkeeping.
//
// Deliberate differences from the reference (each is a defect there):
//   * tensors must live on the GPU; there is NO CPU path (a CPU tensor raises).
//   * value == None means unit weights and the value stream is never read (the
//     reference substitutes ones_like(col), an int64 tensor, :241,:329).
//   * the cached transpose operands (value_index_select / row_index_select /
//     new_row / new_rowcount) may be None: they are then built on the device
//     when backward needs them (the reference dereferences them blindly,
//     :246-247,:333).
//   * grad_value of sum/mean is computed (SDDMM); the reference returns an
//     undefined tensor (:268-272,:349-353).
//   * outputs are torch::empty: the kernel writes every element, including the
//     0 / nnz the reference gets from zeros()/full_like() fills (:147-152,171).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/isplib_hip.h"

namespace {

using at::Tensor;
using c10::optional;
using torch::autograd::AutogradContext;
using torch::autograd::Variable;
using torch::autograd::variable_list;

enum Reduction { R_SUM = 0, R_MAX = 1, R_MIN = 2, R_MEAN = 3 };   // codes of csrc/fusedmm.cpp:168-186

void check_status(int st, const char *what) {
   TORCH_CHECK(st == ISPLIB_SUCCESS, what, " failed with status ", st, ": ", isplib_hip_last_error());
}

void check_index(const Tensor &t, const char *name) {
   TORCH_CHECK(t.defined(), "isplib: `", name, "` is missing");
   TORCH_CHECK(t.is_cuda(), "isplib: `", name, "` must be a GPU tensor -- isplib_amd has no CPU path");
   TORCH_CHECK(t.scalar_type() == at::kLong, "isplib: `", name, "` must be int64 (csrc/fusedmm.cpp:43)");
}

void check_float(const Tensor &t, const char *name) {
   TORCH_CHECK(t.defined(), "isplib: `", name, "` is missing");
   TORCH_CHECK(t.is_cuda(), "isplib: `", name, "` must be a GPU tensor -- isplib_amd has no CPU path");
   TORCH_CHECK(t.scalar_type() == at::kFloat, "isplib: `", name, "` must be float32 (csrc/fusedmm.cpp:44)");
}

void *current_stream(const Tensor &t) { return (void *)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// ISPLIB_DEBUG=1: per-operator device time on stderr under the reference's own labels (the commented-out timers
// of csrc/fusedmm.cpp:52-53,252-253,288-289,...: FUSEDMM_SPMM_<RED>_{FW,BW}).  Synchronises per op: debugging only.
struct OpTimer {
   const char *name;
   bool on;
   hipStream_t st = nullptr;
   hipEvent_t a = nullptr, b = nullptr;
   OpTimer(const char *n, const Tensor &t) : name(n) {
      static const bool enabled = [] { const char *e = std::getenv("ISPLIB_DEBUG"); return e && *e && *e != '0'; }();
      on = enabled && t.defined() && t.is_cuda();
      if (!on) return;
      st = (hipStream_t)current_stream(t);
      on = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess && hipEventRecord(a, st) == hipSuccess;
   }
   ~OpTimer() {
      if (on && hipEventRecord(b, st) == hipSuccess && hipEventSynchronize(b) == hipSuccess) {
         float ms = 0.0f;
         if (hipEventElapsedTime(&ms, a, b) == hipSuccess) std::fprintf(stderr, "%s: %.3f ms\n", name, ms);
      }
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
   }
};

// fusedmm_spmm_fw, csrc/fusedmm.cpp:113-203
// `plan`: per-graph schedule operands (isplib_amd/plan.py, include/isplib_hip.h):
//   {}                                                   plain row kernel
//   {sliceptr}                                           column-sliced kernel
//   {task_row, task_b, task_len, seg_off, lane_off_cpu}  task-list kernel
using Plan = std::vector<Tensor>;
// task plan = {task_row, task_b, task_len, seg_off, lane_off (host)} + optionally the 32-bit copy of col
static inline bool is_task_plan(const Plan &p) { return p.size() == 5 || p.size() == 6; }
// stream plan (sum / mean; isplib_stream_plan) = {words, vals (empty = unit weights), wave_step_off, wave_row, wave_part,
// hub_row, hub_off, meta (host int64: rows, cols, slices, gens, waves_per_gen, rows_per_wave, streams, n_steps, n_parts, n_hub)}
static inline bool is_stream_plan(const Plan &p) { return p.size() == 8 || p.size() == 9; }   // 9: + perm (max / min)
// what the reference-schema operators pass: "no plan was given, choose for me" (one undefined tensor), as opposed
// to the empty plan of the *_planned operators, which means "the plain kernel, please"
static inline Plan auto_plan() { return Plan{Tensor()}; }
static inline bool is_auto_plan(const Plan &p) { return p.size() == 1 && !p[0].defined(); }
static inline const int32_t *plan_col32(const Plan &p, const Tensor &col) {
   if (p.size() != 6) return nullptr;
   TORCH_CHECK(p[5].scalar_type() == at::kInt && p[5].numel() == col.numel() && p[5].device() == col.device(),
               "isplib: the plan's packed column ids do not match col");
   return p[5].data_ptr<int32_t>();
}
// the C struct of a stream plan handed over as a tensor list (is_stream_plan), checked against its own meta
static isplib_stream_plan stream_plan_of(const Plan &plan) {
   const Tensor &words = plan[0], &vals = plan[1], &step_off = plan[2], &wave_row = plan[3], &wave_part = plan[4],
                &hub_row = plan[5], &hub_off = plan[6], &meta = plan[7];
   TORCH_CHECK(!meta.is_cuda() && meta.scalar_type() == at::kLong && meta.numel() == 11, "isplib: stream plan meta must be 11 host int64");
   TORCH_CHECK(words.is_cuda() && words.scalar_type() == at::kInt && step_off.scalar_type() == at::kLong &&
                   wave_row.scalar_type() == at::kInt && wave_part.scalar_type() == at::kInt &&
                   hub_row.scalar_type() == at::kInt && hub_off.scalar_type() == at::kInt &&
                   (vals.numel() == 0 || (vals.scalar_type() == at::kFloat && vals.numel() == words.numel())),
               "isplib: malformed stream plan");
   const int64_t *mt = meta.data_ptr<int64_t>();
   isplib_stream_plan sp;
   sp.rows = mt[0]; sp.cols = mt[1]; sp.slices = (int32_t)mt[2]; sp.gens = (int32_t)mt[3]; sp.waves_per_gen = (int32_t)mt[4];
   sp.rows_per_wave = (int32_t)mt[5]; sp.streams = (int32_t)mt[6]; sp.n_steps = mt[7]; sp.n_parts = mt[8]; sp.n_hub = mt[9]; sp.chunk = (int32_t)mt[10];
   TORCH_CHECK(words.numel() == sp.n_steps * sp.streams && step_off.numel() == (int64_t)sp.gens * sp.waves_per_gen + 1 &&
                   wave_row.numel() == (int64_t)sp.gens * sp.waves_per_gen * sp.rows_per_wave && wave_part.numel() == wave_row.numel() &&
                   hub_row.numel() == sp.n_hub && hub_off.numel() == sp.n_hub + 1,
               "isplib: stream plan arrays do not match its meta");
   sp.words = words.data_ptr<int32_t>();
   sp.vals = vals.numel() ? vals.data_ptr<float>() : nullptr;
   sp.wave_step_off = step_off.data_ptr<int64_t>();
   sp.wave_row = wave_row.data_ptr<int32_t>();
   sp.wave_part = wave_part.data_ptr<int32_t>();
   sp.hub_row = sp.n_hub ? hub_row.data_ptr<int32_t>() : nullptr;
   sp.hub_off = hub_off.data_ptr<int32_t>();
   sp.perm = nullptr;
   if (plan.size() == 9) {
      TORCH_CHECK(plan[8].is_cuda() && plan[8].scalar_type() == at::kInt && plan[8].numel() == words.numel(), "isplib: stream plan perm must be int32, one per word");
      sp.perm = plan[8].data_ptr<int32_t>();
   }
   return sp;
}

// ---- per-graph handles for callers that pass no plan (the reference-schema operators) --------------------------
// iSpLib's Python keeps its per-graph operands in dicts keyed by raw data pointers (isplib/__init__.py:35-40,50) and
// calls the operators with just (rowptr, col, value, mat).  To give that caller the fast schedules, the operator
// library keeps one isplib_graph per STRUCTURE (rowptr, col, dense rows) it has seen -- keyed by the same pointers,
// but every hit is checked against weak references to the very tensors (and their version counters), so a
// freed-and-reused address can never serve a stale plan.  The weights are an argument of the call, not part of the
// key: another `value` tensor, or the same one stepped in place by an optimiser (its version counter moves), keeps
// the plans, the packed column ids and the CSC structure and only tells the handle that its copies of the weights
// are stale (isplib_graph_set_values: no synchronisation, nothing freed).  Small graphs never get here.
struct GraphCacheEntry {
   c10::weak_intrusive_ptr<c10::TensorImpl> rowptr, col, value;
   uint32_t v_rowptr = 0, v_col = 0, v_value = 0;
   bool has_value = false;
   isplib_graph *handle = nullptr;
   GraphCacheEntry(const Tensor &r, const Tensor &c, const Tensor &v)
       : rowptr(r.getIntrusivePtr()), col(c.getIntrusivePtr()), value(v.defined() ? v.getIntrusivePtr() : c.getIntrusivePtr()),
         v_rowptr(r._version()), v_col(c._version()), v_value(v.defined() ? v._version() : 0), has_value(v.defined()) {}
   bool alive() const { return !rowptr.expired() && !col.expired(); }
   bool same_structure(const Tensor &r, const Tensor &c) const {
      return rowptr.lock() == r.getIntrusivePtr() && col.lock() == c.getIntrusivePtr() && v_rowptr == r._version() && v_col == c._version();
   }
   bool same_values(const Tensor &v) const {
      return has_value == v.defined() && (!v.defined() || (value.lock() == v.getIntrusivePtr() && v_value == v._version()));
   }
   void remember_values(const Tensor &v, const Tensor &c) {
      value = c10::weak_intrusive_ptr<c10::TensorImpl>(v.defined() ? v.getIntrusivePtr() : c.getIntrusivePtr());
      v_value = v.defined() ? v._version() : 0;
      has_value = v.defined();
   }
};
struct GraphKey {
   const void *rowptr, *col;
   int64_t n;
   bool operator==(const GraphKey &o) const { return rowptr == o.rowptr && col == o.col && n == o.n; }
};
struct GraphKeyHash {
   size_t operator()(const GraphKey &k) const {
      return std::hash<const void *>()(k.rowptr) ^ (std::hash<const void *>()(k.col) << 1) ^ std::hash<int64_t>()(k.n);
   }
};
std::mutex g_graph_mutex;
std::unordered_map<GraphKey, GraphCacheEntry, GraphKeyHash> g_graphs;
// Handles whose graph is gone or has changed.  isplib_graph_destroy synchronises the device (nothing of the handle's may
// still be in flight when its memory goes), which must not happen inside somebody's forward call: retired handles wait
// here until the next NEW graph gets its handle -- a point where the call synchronises anyway (the plan builders do) and
// is about to allocate what the dead handles hold (plans of A and A^T, packed ids, CSC arrays, workspaces: several GB
// for a Reddit-sized graph): a job that cycles through large graphs never holds more than the graphs retired since the
// last one was created -- or until graph_cache_clear().  graph_cache_size() counts them in.
std::vector<isplib_graph *> g_retired;

void retire_handle_locked(isplib_graph *h) { g_retired.push_back(h); }

void destroy_retired_locked() {
   for (isplib_graph *r : g_retired) isplib_graph_destroy(r);
   g_retired.clear();
}

// the handle of the structure (rowptr, col) with N dense rows; g_graph_mutex must be held.  with_values: the call is
// going to read the weights (SpMM and its backward), so the handle must hold `value` as it is now; SDDMM does not
// read them and takes the handle as it finds it.
isplib_graph *graph_handle_locked(const Tensor &rowptr, const Tensor &col, const Tensor &value, int64_t N, bool with_values = true) {
   const int64_t M = rowptr.numel() - 1, nnz = col.numel();
   const GraphKey key{rowptr.data_ptr(), col.data_ptr(), N};
   auto it = g_graphs.find(key);
   if (it != g_graphs.end() && !it->second.same_structure(rowptr, col)) {
      retire_handle_locked(it->second.handle);
      g_graphs.erase(it);
      it = g_graphs.end();
   }
   if (it == g_graphs.end()) {
      for (auto dead = g_graphs.begin(); dead != g_graphs.end();) {      // graphs whose tensors are gone
         if (!dead->second.alive()) {
            retire_handle_locked(dead->second.handle);
            dead = g_graphs.erase(dead);
         } else {
            ++dead;
         }
      }
      destroy_retired_locked();
      const Tensor first = with_values ? value : Tensor();
      GraphCacheEntry e(rowptr, col, first);
      check_status(isplib_graph_create(M, N, nnz, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                       first.defined() ? first.data_ptr<float>() : nullptr, &e.handle),
                   "isplib_graph_create");
      it = g_graphs.emplace(key, std::move(e)).first;
   } else if (with_values && !it->second.same_values(value)) {
      check_status(isplib_graph_set_values(it->second.handle, value.defined() ? value.data_ptr<float>() : nullptr), "isplib_graph_set_values");
      it->second.remember_values(value, col);
   }
   return it->second.handle;
}

bool spmm_through_handle(int32_t msg, const Tensor &rowptr, const Tensor &col, const Tensor &value, const Tensor &mat,
                         Tensor &out, Tensor &arg) {
   const int64_t M = rowptr.numel() - 1, N = mat.size(0), K = mat.size(1), nnz = col.numel();
   if (isplib_suggest_slices(M, N, nnz, K, (msg & 0xF0000) != ISPLIB_AOP_ADD) <= 0) return false;
   std::lock_guard<std::mutex> lock(g_graph_mutex);        // also serialises the handle's shared workspace
   isplib_graph *handle = graph_handle_locked(rowptr, col, value, N);
   const int st = isplib_graph_spmm(handle, msg, K, mat.data_ptr<float>(), K, out.data_ptr<float>(), K,
                                    arg.defined() ? arg.data_ptr<int64_t>() : nullptr, current_stream(mat));
   if (st == ISPLIB_NOT_ENOUGH_MEM) return false;      // no room for the plan beside torch's pool: the plain kernel needs none
   check_status(st, "isplib_graph_spmm");
   return true;
}

// want_arg = false (max / min through the *_values operators: nobody will ask which edge won): on a stream plan the
// launch then leaves the positions out altogether (isplib_hip.h: fusedMM_csr_stream_minmax_hip with z_arg = NULL) and the
// second tensor of the result is undefined; every other schedule computes them as always
std::tuple<Tensor, Tensor> spmm_fw(const Tensor &rowptr_, const Tensor &col_, const optional<Tensor> &value_,
                                   const Tensor &mat_, int reduction, const Plan &plan = Plan(), bool want_arg = true) {
   check_index(rowptr_, "rowptr");
   check_index(col_, "col");
   check_float(mat_, "mat");
   TORCH_CHECK(mat_.dim() == 2, "isplib: `mat` must be 2-D [N, K] (csrc/fusedmm.cpp:121-122)");
   TORCH_CHECK(rowptr_.dim() == 1 && rowptr_.numel() >= 1, "isplib: `rowptr` must be 1-D with M+1 entries");
   c10::DeviceGuard guard(mat_.device());
   const Tensor rowptr = rowptr_.contiguous(), col = col_.contiguous(), mat = mat_.contiguous();   // :140
   Tensor value;
   if (value_.has_value() && value_->defined()) {
      check_float(*value_, "value");
      value = value_->contiguous();
      TORCH_CHECK(value.numel() == col.numel(), "isplib: `value` and `col` differ in length");
   }
   TORCH_CHECK(rowptr.device() == mat.device() && col.device() == mat.device(), "isplib: operands on different devices");
   const int64_t M = rowptr.numel() - 1, N = mat.size(0), K = mat.size(1), nnz = col.numel();
   Tensor out = at::empty({M, K}, mat.options());
   Tensor arg;
   int32_t msg = ISPLIB_MSG_SPMM_SUM;
   if (reduction == R_MAX) msg = ISPLIB_MSG_SPMM_MAX;
   if (reduction == R_MIN) msg = ISPLIB_MSG_SPMM_MIN;
   if (reduction == R_MEAN) msg = ISPLIB_MSG_SPMM_MEAN;
   const int64_t *rp = rowptr.data_ptr<int64_t>();
   const bool tasks_fit = K >= 4 && (double)N * (double)K * 4.0 <= 3.5 * 1073741824.0;
   // (a stream plan for a shape outside the stream entry's domain -- dense operand over 3.5 GiB, k < 4 -- is not an error:
   // the graph is served by the kernels below, which read `col` / `value`)
   const bool minmax_op = reduction == R_MAX || reduction == R_MIN;      // (the max / min stream entry serves dense operands under 2 GiB)
   const bool on_stream = is_stream_plan(plan) && M > 0 && K >= 4 && (double)N * (double)K * 4.0 <= 3.5 * 1073741824.0 &&
                          !(minmax_op && (double)N * (double)K * 4.0 >= 2.0 * 1073741824.0);
   if (minmax_op && (want_arg || !on_stream)) arg = at::empty({M, K}, rowptr.options());
   if (on_stream) {
      // the plan carries the edges (and the weights) in its own order: `col` / `value` are not read
      const isplib_stream_plan sp = stream_plan_of(plan);
      // A line-friendly pitch for 33..47 columns (the GCN's 41 classes): what the address pipeline charges for is the
      // 128-byte line a gather touches, and a 164-byte row at its packed pitch straddles 2.25 of them on average, at a
      // 192-byte pitch exactly 2 -- Reddit shape K=41: 1.365 -> 1.285 ms, the copy 0.015 (scripts/exp_round4.py k41;
      // 176 B: no gain, 256 B: 1.314).  Only the gathered operand is copied; the output stays packed.
      const float *y = mat.data_ptr<float>();
      int64_t ldy = K;
      Tensor pitched;
      if (K > 32 && K < 48 && N >= (1 << 16) && (double)N * 48.0 * 4.0 < 2.0 * 1073741824.0) {
         pitched = at::empty({N, 48}, mat.options());
         pitched.narrow(1, 0, K).copy_(mat);
         y = pitched.data_ptr<float>();
         ldy = 48;
      }
      if (reduction == R_MAX || reduction == R_MIN) {
         TORCH_CHECK(sp.perm != nullptr, "isplib: max / min on a stream plan need its permutation (a 9-element plan)");
         const size_t ws = isplib_spmm_stream_minmax_workspace_bytes(&sp);
         Tensor work = at::empty({(int64_t)ws}, mat.options().dtype(at::kByte));
         const int st = fusedMM_csr_stream_minmax_hip(msg, M, N, K, nnz, rp, rp + 1, &sp, y, ldy, out.data_ptr<float>(), K,
                                                      arg.defined() ? arg.data_ptr<int64_t>() : nullptr, work.data_ptr(), ws, current_stream(mat));
         check_status(st, "fusedMM_csr_stream_minmax_hip");
         return std::make_tuple(out, arg);
      }
      const size_t ws = isplib_spmm_stream_workspace_bytes(&sp);
      Tensor work = at::empty({(int64_t)ws}, mat.options().dtype(at::kByte));
      const int st = fusedMM_csr_stream_hip(msg, M, N, K, nnz, rp, rp + 1, &sp, y, ldy, out.data_ptr<float>(), K,
                                            work.data_ptr(), ws, nullptr, current_stream(mat));
      check_status(st, "fusedMM_csr_stream_hip");
      return std::make_tuple(out, arg);
   }
   if (is_task_plan(plan) && tasks_fit && M > 0 && K > 0) {
      const Tensor &task_row = plan[0], &task_b = plan[1], &task_len = plan[2], &seg_off = plan[3], &lane = plan[4];
      TORCH_CHECK(task_row.is_cuda() && task_row.scalar_type() == at::kInt && task_len.scalar_type() == at::kInt &&
                      seg_off.scalar_type() == at::kInt && task_b.scalar_type() == at::kLong,
                  "isplib: malformed task plan");
      TORCH_CHECK(!lane.is_cuda() && lane.scalar_type() == at::kLong && lane.numel() == 9, "isplib: lane_off must be 9 host int64");
      TORCH_CHECK((seg_off.numel() - 1) % M == 0, "isplib: task plan does not match rowptr");
      const int nsl = (int)((seg_off.numel() - 1) / M);
      const int64_t n_tasks = task_row.numel();
      const size_t ws = isplib_spmm_tasks_workspace_bytes(msg, n_tasks, K);
      Tensor work = at::empty({(int64_t)ws}, mat.options().dtype(at::kByte));
      const int st = fusedMM_csr_tasks_hip(msg, M, N, K, nnz, value.defined() ? value.data_ptr<float>() : nullptr,
                                           col.data_ptr<int64_t>(), plan_col32(plan, col), rp, rp + 1, n_tasks, task_row.data_ptr<int32_t>(),
                                           task_b.data_ptr<int64_t>(), task_len.data_ptr<int32_t>(),
                                           seg_off.data_ptr<int32_t>(), nsl, lane.data_ptr<int64_t>(),
                                           mat.data_ptr<float>(), K, out.data_ptr<float>(), K,
                                           arg.defined() ? arg.data_ptr<int64_t>() : nullptr, work.data_ptr(), ws,
                                           current_stream(mat));
      check_status(st, "fusedMM_csr_tasks_hip");
      return std::make_tuple(out, arg);
   }
   if (plan.size() == 1 && plan[0].defined() && plan[0].scalar_type() == at::kInt && M > 0 && K > 0) {
      // a row order (int32 [M], position -> row): the plain kernel with the rows taken in that order (operands larger
      // than the Infinity Cache on graphs with community structure; fusedMM_csr_ordered_hip) -- the same bits as below
      const Tensor order = plan[0].contiguous();
      TORCH_CHECK(order.is_cuda() && order.numel() == M, "isplib: the row order must hold one int32 position per row");
      const int st = fusedMM_csr_ordered_hip(msg, M, N, K, nnz, value.defined() ? value.data_ptr<float>() : nullptr,
                                             col.data_ptr<int64_t>(), rp, rp + 1, order.data_ptr<int32_t>(), mat.data_ptr<float>(), K,
                                             out.data_ptr<float>(), K, arg.defined() ? arg.data_ptr<int64_t>() : nullptr,
                                             current_stream(mat));
      check_status(st, "fusedMM_csr_ordered_hip");
      return std::make_tuple(out, arg);
   }
   if (plan.size() == 1 && plan[0].defined() && M > 0 && K > 0) {
      check_index(plan[0], "slices");
      const Tensor table = plan[0].contiguous();
      TORCH_CHECK(table.numel() % M == 0 && table.numel() / M >= 9, "isplib: slice table does not match rowptr");
      const int nsl = (int)(table.numel() / M - 1);
      const size_t ws = isplib_spmm_sliced_workspace_bytes(msg, M, K, nsl);
      Tensor work = at::empty({(int64_t)ws}, mat.options().dtype(at::kByte));
      const int st = fusedMM_csr_sliced_hip(msg, M, N, K, nnz, value.defined() ? value.data_ptr<float>() : nullptr,
                                            col.data_ptr<int64_t>(), rp, rp + 1, table.data_ptr<int64_t>(), nsl,
                                            mat.data_ptr<float>(), K, out.data_ptr<float>(), K,
                                            arg.defined() ? arg.data_ptr<int64_t>() : nullptr, work.data_ptr(), ws,
                                            current_stream(mat));
      check_status(st, "fusedMM_csr_sliced_hip");
      return std::make_tuple(out, arg);
   }
   // no plan given (the reference's own call pattern): large graphs go through a cached per-graph handle
   if (is_auto_plan(plan) && M > 0 && K > 0 && rowptr.is_same(rowptr_) && col.is_same(col_) &&
       (!value.defined() || value.is_same(*value_)) && spmm_through_handle(msg, rowptr, col, value, mat, out, arg))
      return std::make_tuple(out, arg);
   const int st = fusedMM_csr_hip(msg, M, N, K, 1.0f, nnz, M, N, value.defined() ? value.data_ptr<float>() : nullptr,
                                  col.data_ptr<int64_t>(), rp, rp + 1, nullptr, K, mat.data_ptr<float>(), K, 0.0f,
                                  out.data_ptr<float>(), K, arg.defined() ? arg.data_ptr<int64_t>() : nullptr,
                                  current_stream(mat));
   check_status(st, "fusedMM_csr_hip");
   return std::make_tuple(out, arg);
}

struct Transposed {
   Tensor colptr, row_t, val_t;
};

// A^T operands built on the device (isplib/__init__.py:79-80 / :86-99 equivalents)
Transposed build_transpose(const Tensor &rowptr, const Tensor &col, const Tensor &value, int64_t ncols, bool mean) {
   c10::DeviceGuard guard(col.device());
   const int64_t M = rowptr.numel() - 1, nnz = col.numel();
   Transposed t;
   t.colptr = at::empty({ncols + 1}, rowptr.options());
   t.row_t = at::empty({nnz}, rowptr.options());
   t.val_t = at::empty({nnz}, col.options().dtype(at::kFloat));
   const size_t ws = isplib_csr2csc_workspace_bytes(M, ncols, nnz);
   TORCH_CHECK(ws > 0, "isplib_csr2csc_workspace_bytes failed: ", isplib_hip_last_error());
   Tensor work = at::empty({(int64_t)ws}, col.options().dtype(at::kByte));
   const int st = isplib_csr2csc_hip(M, ncols, nnz, rowptr.data_ptr<int64_t>(), col.data_ptr<int64_t>(),
                                     value.defined() ? value.data_ptr<float>() : nullptr, mean ? 1 : 0,
                                     t.colptr.data_ptr<int64_t>(), nullptr, t.row_t.data_ptr<int64_t>(),
                                     t.val_t.data_ptr<float>(), work.data_ptr(), ws, current_stream(col));
   check_status(st, "isplib_csr2csc_hip");
   return t;
}

Tensor sddmm(const Tensor &rowptr, const Tensor &col, const Tensor &mat, const Tensor &grad_out, bool mean,
             const std::vector<Tensor> &plan = {}, const Tensor &value = Tensor()) {
   c10::DeviceGuard guard(mat.device());
   const Tensor g = grad_out.contiguous(), y = mat.contiguous();
   const int64_t M = rowptr.numel() - 1, N = y.size(0), K = y.size(1);
   Tensor dval = at::empty({col.numel()}, y.options());
   const int64_t *rp = rowptr.data_ptr<int64_t>();
   // The dot product needs whole rows of y, so the SpMM's plan (sized for 64-column panels) is the wrong one here:
   // large graphs go through the graph's handle, which keeps a plan sized for whole rows (Reddit K=128: 3.7 ms with
   // 16 slices, 4.6 ms on the SpMM's 8).  dA does not read the weights (and the weights tensor autograd hands back
   // here is a different object from the forward's): the structure's handle is used as it is.
   (void)value;
   if (K >= 4 && rowptr.is_contiguous() && col.is_contiguous() && isplib_suggest_slices_whole_rows(M, N, col.numel(), K) > 0) {
      std::lock_guard<std::mutex> lock(g_graph_mutex);
      isplib_graph *handle = graph_handle_locked(rowptr, col, Tensor(), N, /*with_values=*/false);
      const int st = isplib_graph_sddmm(handle, mean ? 1 : 0, K, y.data_ptr<float>(), K, g.data_ptr<float>(), K,
                                        dval.data_ptr<float>(), current_stream(y));
      if (st != ISPLIB_NOT_ENOUGH_MEM) {
         check_status(st, "isplib_graph_sddmm");
         return dval;
      }
   }
   if (is_task_plan(plan) && K >= 4 && K <= 1024 && (double)N * (double)K * 4.0 <= 3.5 * 1073741824.0) {
      const int st = isplib_sddmm_csr_tasks_hip(M, N, K, col.data_ptr<int64_t>(), plan_col32(plan, col), rp, rp + 1, plan[0].numel(),
                                                plan[0].data_ptr<int32_t>(), plan[1].data_ptr<int64_t>(),
                                                plan[2].data_ptr<int32_t>(), plan[4].data_ptr<int64_t>(),
                                                y.data_ptr<float>(), K, g.data_ptr<float>(), K, mean ? 1 : 0,
                                                dval.data_ptr<float>(), current_stream(y));
      check_status(st, "isplib_sddmm_csr_tasks_hip");
      return dval;
   }
   const int st = isplib_sddmm_csr_hip(M, K, col.data_ptr<int64_t>(), rp, rp + 1, y.data_ptr<float>(), K,
                                       g.data_ptr<float>(), K, mean ? 1 : 0, dval.data_ptr<float>(), current_stream(y));
   check_status(st, "isplib_sddmm_csr_hip");
   return dval;
}

// sum-SpMM with unit weights and the fused epilogue out = act(row_scale * (A y + self) + bias); the fold kernel
// applies it when a task plan is given (isplib_epilogue), otherwise it is composed from ATen ops.
// a [N, K] matrix whose rows may sit at a wider pitch (a view of an [N, pitch] buffer): usable as it is by the stream entry
static bool row_strided(const Tensor &t) { return t.dim() == 2 && t.stride(1) == 1 && (t.size(0) <= 1 || t.stride(0) >= t.size(1)); }

Tensor epilogue_spmm(const Tensor &rowptr, const Tensor &col, const Plan &plan, const Tensor &y_, const Tensor &self_,
                     const Tensor &row_scale, const Tensor &bias, bool relu) {
   const int64_t M = rowptr.numel() - 1, N = y_.size(0), K = y_.size(1), nnz = col.numel();
   const bool tasks_fit = is_task_plan(plan) && K >= 4 && M > 0 && (double)N * (double)K * 4.0 <= 3.5 * 1073741824.0;
   if (is_stream_plan(plan) && M > 0 && K >= 4) {
      // the stream kernel applies the same epilogue when it writes a finished row (hub rows: in their fold)
      c10::DeviceGuard guard(y_.device());
      const Tensor y = row_strided(y_) ? y_ : y_.contiguous();          // the gather's own pitch is kept (GcnNormSpmm)
      const int64_t ldy = N > 1 ? y.stride(0) : K;
      TORCH_CHECK((double)N * (double)ldy * 4.0 <= 3.5 * 1073741824.0, "isplib: dense operand beyond one buffer descriptor");
      const Tensor self = self_.defined() ? (row_strided(self_) ? self_ : self_.contiguous()) : Tensor();
      const Tensor rs = row_scale.defined() ? row_scale.contiguous() : Tensor();
      const Tensor bs = bias.defined() ? bias.contiguous() : Tensor();
      if (self.defined()) TORCH_CHECK(self.size(0) == M && self.size(1) == K, "isplib: `self` must be [M, K]");
      if (rs.defined()) TORCH_CHECK(rs.numel() == M, "isplib: `row_scale` must have M entries");
      if (bs.defined()) TORCH_CHECK(bs.numel() == K, "isplib: `bias` must have K entries");
      const isplib_stream_plan sp = stream_plan_of(plan);
      TORCH_CHECK(sp.vals == nullptr, "isplib: the fused epilogue is defined for unit weights");
      const Tensor rp_c = rowptr.contiguous();
      Tensor out = at::empty({M, K}, y.options());
      const size_t ws = isplib_spmm_stream_workspace_bytes(&sp);
      Tensor work = at::empty({(int64_t)ws}, y.options().dtype(at::kByte));
      isplib_epilogue ep;
      ep.row_scale = rs.defined() ? rs.data_ptr<float>() : nullptr;
      ep.self = self.defined() ? self.data_ptr<float>() : nullptr;
      ep.ld_self = self.defined() && M > 1 ? self.stride(0) : K;
      ep.bias = bs.defined() ? bs.data_ptr<float>() : nullptr;
      ep.relu = relu ? 1 : 0;
      const int64_t *rp = rp_c.data_ptr<int64_t>();
      const int st = fusedMM_csr_stream_hip(ISPLIB_MSG_SPMM_SUM, M, N, K, nnz, rp, rp + 1, &sp, y.data_ptr<float>(), ldy, out.data_ptr<float>(), K,
                                            work.data_ptr(), ws, &ep, current_stream(y));
      check_status(st, "fusedMM_csr_stream_hip");
      return out;
   }
   const Tensor y = y_.contiguous();
   if (!tasks_fit) {
      Tensor out = std::get<0>(spmm_fw(rowptr, col, c10::nullopt, y, R_SUM, plan));
      if (self_.defined()) out = out + self_;
      if (row_scale.defined()) out = out * row_scale.unsqueeze(1);
      if (bias.defined()) out = out + bias;
      return relu ? at::relu(out) : out;
   }
   c10::DeviceGuard guard(y.device());
   const Tensor self = self_.defined() ? self_.contiguous() : Tensor();
   const Tensor rs = row_scale.defined() ? row_scale.contiguous() : Tensor();
   const Tensor bs = bias.defined() ? bias.contiguous() : Tensor();
   if (self.defined()) TORCH_CHECK(self.size(0) == M && self.size(1) == K, "isplib: `self` must be [M, K]");
   if (rs.defined()) TORCH_CHECK(rs.numel() == M, "isplib: `row_scale` must have M entries");
   if (bs.defined()) TORCH_CHECK(bs.numel() == K, "isplib: `bias` must have K entries");
   const Tensor rp_c = rowptr.contiguous(), col_c = col.contiguous();
   Tensor out = at::empty({M, K}, y.options());
   const int nsl = (int)((plan[3].numel() - 1) / M);
   const int64_t n_tasks = plan[0].numel();
   const size_t ws = isplib_spmm_tasks_workspace_bytes(ISPLIB_MSG_SPMM_SUM, n_tasks, K);
   Tensor work = at::empty({(int64_t)ws}, y.options().dtype(at::kByte));
   isplib_epilogue ep;
   ep.row_scale = rs.defined() ? rs.data_ptr<float>() : nullptr;
   ep.self = self.defined() ? self.data_ptr<float>() : nullptr;
   ep.ld_self = K;
   ep.bias = bs.defined() ? bs.data_ptr<float>() : nullptr;
   ep.relu = relu ? 1 : 0;
   const int64_t *rp = rp_c.data_ptr<int64_t>();
   const int st = fusedMM_csr_tasks_epilogue_hip(
       ISPLIB_MSG_SPMM_SUM, M, N, K, nnz, nullptr, col_c.data_ptr<int64_t>(), plan_col32(plan, col_c), rp, rp + 1, n_tasks,
       plan[0].data_ptr<int32_t>(), plan[1].data_ptr<int64_t>(), plan[2].data_ptr<int32_t>(), plan[3].data_ptr<int32_t>(), nsl,
       plan[4].data_ptr<int64_t>(), y.data_ptr<float>(), K, out.data_ptr<float>(), K, work.data_ptr(), ws, &ep,
       current_stream(y));
   check_status(st, "fusedMM_csr_tasks_epilogue_hip");
   return out;
}

Tensor or_undef(const optional<Tensor> &t) { return t.has_value() ? *t : Tensor(); }

// AutogradContext::needs_input_grad() is indexed by autograd EDGE, i.e. by position among
// the tensor arguments that are actually present -- absent optionals take no edge.
int64_t edge_of(std::initializer_list<bool> present_before) {
   int64_t e = 0;
   for (bool p : present_before) e += p ? 1 : 0;
   return e;
}
bool present(const optional<Tensor> &t) { return t.has_value() && t->defined(); }

// ---- sum: csrc/fusedmm.cpp:210-294 ---------------------------------------------------------
class SpmmSum : public torch::autograd::Function<SpmmSum> {
 public:
   static variable_list forward(AutogradContext *ctx, optional<Variable> opt_row, Variable rowptr, Variable col,
                                optional<Variable> opt_value, optional<Variable> opt_colptr,
                                optional<Variable> opt_csr2csc, Variable mat, optional<Variable> value_index_select,
                                optional<Variable> row_index_select, Plan plan, Plan plan_t) {
      const bool has_value = opt_value.has_value() && opt_value->defined();
      OpTimer timer("FUSEDMM_SPMM_SUM_FW", mat);
      auto out = std::get<0>(spmm_fw(rowptr, col, opt_value, mat, R_SUM, plan));   // :244
      ctx->saved_data["plan_t"] = plan_t;
      ctx->saved_data["plan"] = plan;
      ctx->saved_data["has_value"] = has_value;
      ctx->saved_data["value_edge"] = edge_of({present(opt_row), true, true});
      ctx->saved_data["mat_edge"] =
          edge_of({present(opt_row), true, true, has_value, present(opt_colptr), present(opt_csr2csc)});
      ctx->save_for_backward({or_undef(opt_row), rowptr, col, or_undef(opt_value), or_undef(opt_colptr),
                              or_undef(opt_csr2csc), mat, or_undef(value_index_select), or_undef(row_index_select)});
      return {out};
   }

   static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
      OpTimer timer("FUSEDMM_SPMM_SUM_BW", grad_outs[0]);
      const bool has_value = ctx->saved_data["has_value"].toBool();
      auto grad_out = grad_outs[0];
      auto saved = ctx->get_saved_variables();
      auto rowptr = saved[1], col = saved[2], value = saved[3], colptr = saved[4], mat = saved[6],
           value_sel = saved[7], row_sel = saved[8];

      auto grad_value = Variable();
      if (has_value && ctx->needs_input_grad(ctx->saved_data["value_edge"].toInt()))   // :269-272 (SDDMM, commented out there)
         grad_value = sddmm(rowptr, col, mat, grad_out, false, ctx->saved_data["plan"].toTensorVector(), value);

      auto grad_mat = Variable();
      if (ctx->needs_input_grad(ctx->saved_data["mat_edge"].toInt())) {
         // :285  grad_mat = fusedmm_spmm_fw(colptr, row_index_select, value_index_select, grad_out)
         if (colptr.defined() && row_sel.defined() && (value_sel.defined() || !has_value)) {
            optional<Tensor> v = has_value ? optional<Tensor>(value_sel) : c10::nullopt;
            grad_mat = std::get<0>(spmm_fw(colptr, row_sel, v, grad_out, R_SUM, ctx->saved_data["plan_t"].toTensorVector()));
         } else {
            auto t = build_transpose(rowptr, col, has_value ? value : Tensor(), mat.size(0), false);
            optional<Tensor> v = has_value ? optional<Tensor>(t.val_t) : c10::nullopt;
            grad_mat = std::get<0>(spmm_fw(t.colptr, t.row_t, v, grad_out, R_SUM));
         }
      }
      return {Variable(), Variable(), Variable(), grad_value, Variable(), Variable(),
              grad_mat,   Variable(), Variable(), Variable(),  Variable()};
   }
};

// ---- mean: csrc/fusedmm.cpp:296-384 --------------------------------------------------------
class SpmmMean : public torch::autograd::Function<SpmmMean> {
 public:
   static variable_list forward(AutogradContext *ctx, optional<Variable> opt_row, Variable rowptr, Variable col,
                                optional<Variable> opt_value, optional<Variable> opt_rowcount,
                                optional<Variable> opt_colptr, optional<Variable> opt_csr2csc, Variable mat,
                                optional<Variable> new_row, optional<Variable> new_rowcount, Plan plan, Plan plan_t) {
      const bool has_value = opt_value.has_value() && opt_value->defined();
      OpTimer timer("FUSEDMM_SPMM_MEAN_FW", mat);
      auto out = std::get<0>(spmm_fw(rowptr, col, opt_value, mat, R_MEAN, plan));   // :331
      ctx->saved_data["plan_t"] = plan_t;
      ctx->saved_data["plan"] = plan;
      ctx->saved_data["has_value"] = has_value;
      ctx->saved_data["value_edge"] = edge_of({present(opt_row), true, true});
      ctx->saved_data["mat_edge"] = edge_of({present(opt_row), true, true, has_value, present(opt_rowcount),
                                             present(opt_colptr), present(opt_csr2csc)});
      ctx->save_for_backward({or_undef(opt_row), rowptr, col, or_undef(opt_value), or_undef(opt_rowcount),
                              or_undef(opt_colptr), or_undef(opt_csr2csc), mat, or_undef(new_row),
                              or_undef(new_rowcount)});
      return {out};
   }

   static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
      OpTimer timer("FUSEDMM_SPMM_MEAN_BW", grad_outs[0]);
      const bool has_value = ctx->saved_data["has_value"].toBool();
      auto grad_out = grad_outs[0];
      auto saved = ctx->get_saved_variables();
      auto rowptr = saved[1], col = saved[2], value = saved[3], colptr = saved[5], mat = saved[7], new_row = saved[8],
           new_rowcount = saved[9];

      auto grad_value = Variable();
      if (has_value && ctx->needs_input_grad(ctx->saved_data["value_edge"].toInt()))
         grad_value = sddmm(rowptr, col, mat, grad_out, true, ctx->saved_data["plan"].toTensorVector(), value);   // :350-353

      auto grad_mat = Variable();
      if (ctx->needs_input_grad(ctx->saved_data["mat_edge"].toInt())) {
         // :375  grad_mat = fusedmm_spmm_fw(colptr, new_row, new_rowcount, grad_out)   (a SUM on A^T)
         const bool cached = colptr.defined() && new_row.defined() && new_rowcount.defined() &&
                             new_row.numel() == col.numel() && new_rowcount.numel() == col.numel() &&
                             new_rowcount.scalar_type() == at::kFloat && new_row.is_cuda();
         // an unweighted graph needs no edge weights here either: (A^T diag(1/deg)) dY = A^T (diag(1/deg) dY) -- the rows
         // of dY are scaled once (M x K elementwise) and the SUM on A^T runs with unit weights, i.e. without a weight stream
         const bool unit_planned = !has_value && colptr.defined() && new_row.defined() && !new_rowcount.defined() &&
                                   new_row.numel() == col.numel() && new_row.is_cuda();
         if (cached) {
            grad_mat = std::get<0>(spmm_fw(colptr, new_row, optional<Tensor>(new_rowcount), grad_out, R_SUM,
                                           ctx->saved_data["plan_t"].toTensorVector()));
         } else if (unit_planned) {
            const Tensor deg = (rowptr.slice(0, 1) - rowptr.slice(0, 0, -1)).clamp_min(1).to(grad_out.scalar_type());
            const Tensor gy = grad_out / deg.unsqueeze(1);
            grad_mat = std::get<0>(spmm_fw(colptr, new_row, c10::nullopt, gy, R_SUM, ctx->saved_data["plan_t"].toTensorVector()));
         } else {
            auto t = build_transpose(rowptr, col, has_value ? value : Tensor(), mat.size(0), true);
            grad_mat = std::get<0>(spmm_fw(t.colptr, t.row_t, optional<Tensor>(t.val_t), grad_out, R_SUM));
         }
      }
      return {Variable(), Variable(), Variable(), grad_value, Variable(), Variable(),
              Variable(), grad_mat,   Variable(), Variable(), Variable(),  Variable()};
   }
};

// ---- max / min: csrc/fusedmm.cpp:386-452, 454-518 ------------------------------------------
template <int RED>
class SpmmMinMax : public torch::autograd::Function<SpmmMinMax<RED>> {
 public:
   static variable_list forward(AutogradContext *ctx, Variable rowptr, Variable col, optional<Variable> opt_value,
                                Variable mat, Plan plan) {
      const bool has_value = opt_value.has_value() && opt_value->defined();
      OpTimer timer(RED == R_MAX ? "FUSEDMM_SPMM_MAX_FW" : "FUSEDMM_SPMM_MIN_FW", mat);
      auto result = spmm_fw(rowptr, col, opt_value, mat, RED, plan);   // :397 / :465
      auto out = std::get<0>(result);
      auto arg_out = std::get<1>(result);
      ctx->saved_data["has_value"] = has_value;
      ctx->save_for_backward({col, or_undef(opt_value), mat, arg_out});
      ctx->mark_non_differentiable({arg_out});                    // :403 / :470
      return {out, arg_out};
   }

   static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
      OpTimer timer(RED == R_MAX ? "FUSEDMM_SPMM_MAX_BW" : "FUSEDMM_SPMM_MIN_BW", grad_outs[0]);
      const bool has_value = ctx->saved_data["has_value"].toBool();
      auto saved = ctx->get_saved_variables();
      auto col = saved[0], value = saved[1], mat = saved[2], arg_out = saved[3];
      const Tensor grad_out = grad_outs[0].contiguous();
      // edges: rowptr 0, col 1, value 2 (if present), mat last
      const bool need_val = has_value && ctx->needs_input_grad(2);
      const bool need_mat = ctx->needs_input_grad(has_value ? 3 : 2);
      auto grad_value = Variable(), grad_mat = Variable();
      if (need_val || need_mat) {
         c10::DeviceGuard guard(mat.device());
         const Tensor y = mat.contiguous();
         const int64_t M = arg_out.size(0), N = y.size(0), K = y.size(1), nnz = col.numel();
         if (need_val) grad_value = at::empty({nnz}, y.options());
         if (need_mat) grad_mat = at::empty({N, K}, y.options());
         // one fused pass for :417-446 (mask, gather, mul, masked_fill, scatter_add x2)
         // the atomic-free form (bitwise reproducible gradients) wherever its 32-bit keys reach; else the one-pass scatter
         const size_t ws = isplib_spmm_minmax_bw_workspace_bytes(M, N, K);
         if (ws > 0) {
            Tensor work = at::empty({(int64_t)ws}, y.options().dtype(at::kByte));
            const int st = isplib_spmm_minmax_bw_det_hip(
                M, N, K, nnz, col.data_ptr<int64_t>(), has_value ? value.data_ptr<float>() : nullptr, y.data_ptr<float>(),
                arg_out.data_ptr<int64_t>(), grad_out.data_ptr<float>(), need_mat ? grad_mat.data_ptr<float>() : nullptr,
                need_val ? grad_value.data_ptr<float>() : nullptr, work.data_ptr(), ws, current_stream(y));
            check_status(st, "isplib_spmm_minmax_bw_det_hip");
         } else {
            const int st = isplib_spmm_minmax_bw_hip(
                M, N, K, nnz, col.data_ptr<int64_t>(), has_value ? value.data_ptr<float>() : nullptr, y.data_ptr<float>(),
                arg_out.data_ptr<int64_t>(), grad_out.data_ptr<float>(), need_mat ? grad_mat.data_ptr<float>() : nullptr,
                need_val ? grad_value.data_ptr<float>() : nullptr, current_stream(y));
            check_status(st, "isplib_spmm_minmax_bw_hip");
         }
      }
      return {Variable(), Variable(), grad_value, grad_mat, Variable()};
   }
};

// The pitch (floats per row) a gathered [N, K] operand of the stream schedule should be written at: 48 for 33..47 columns
// on large graphs (a 164-byte row at its packed pitch straddles 2.25 cache lines on average, at 192 bytes exactly 2: spmm_fw
// above copies for the same reason); K otherwise.
static int64_t gather_pitch(int64_t N, int64_t K, bool stream) {
   return (stream && K > 32 && K < 48 && N >= (1 << 16) && (double)N * 48.0 * 4.0 < 2.0 * 1073741824.0) ? 48 : K;
}

// y = D^-1/2 X in one pass (isplib_row_scale_hip), written at the gather's pitch: a [N, K] view
static Tensor gcn_row_scale(const Tensor &mat_, const Tensor &dinv_, bool stream) {
   c10::DeviceGuard guard(mat_.device());
   const Tensor mat = row_strided(mat_) ? mat_ : mat_.contiguous();
   const Tensor dinv = dinv_.contiguous();
   const int64_t N = mat.size(0), K = mat.size(1);
   TORCH_CHECK(dinv.numel() == N, "isplib: `dinv` must have one entry per row of `mat`");
   const int64_t ld = gather_pitch(N, K, stream);
   Tensor buf = at::empty({N, ld}, mat.options());
   const int st = isplib_row_scale_hip(N, K, mat.data_ptr<float>(), N > 1 ? mat.stride(0) : K, dinv.data_ptr<float>(), buf.data_ptr<float>(),
                                       ld, current_stream(mat));
   check_status(st, "isplib_row_scale_hip");
   return buf.narrow(1, 0, K);
}

// ---- GCN's normalised aggregation, fused: relu(D^-1/2 (A + I) D^-1/2 X + b) with unit-weight A ----------
// (the `normalize=True` callers, tests/dist/gcn/pyg-sparse.py:61-62; not an operator of the reference)
class GcnNormSpmm : public torch::autograd::Function<GcnNormSpmm> {
 public:
   static variable_list forward(AutogradContext *ctx, Variable rowptr, Variable col, Variable mat, Variable dinv,
                                optional<Variable> opt_colptr, optional<Variable> opt_row_t, Plan plan, Plan plan_t,
                                optional<Variable> opt_bias, bool relu) {
      check_index(rowptr, "rowptr"); check_index(col, "col"); check_float(mat, "mat"); check_float(dinv, "dinv");
      TORCH_CHECK(mat.dim() == 2 && rowptr.numel() - 1 == mat.size(0), "isplib: gcn_norm_spmm needs a square graph and [N, K] features");
      const Tensor bias = or_undef(opt_bias);
      const Tensor y = gcn_row_scale(mat, dinv, is_stream_plan(plan));      // D^-1/2 X, one pass, at the gather's pitch
      Tensor out = epilogue_spmm(rowptr, col, plan, y, y, dinv, bias, relu);
      ctx->saved_data["relu"] = relu;
      ctx->saved_data["plan_t"] = plan_t;
      ctx->saved_data["bias_edge"] = (int64_t)(4 + (present(opt_colptr) ? 1 : 0) + (present(opt_row_t) ? 1 : 0));
      ctx->saved_data["has_bias"] = bias.defined();
      ctx->save_for_backward({rowptr, col, dinv, or_undef(opt_colptr), or_undef(opt_row_t), relu ? out : Tensor()});
      return {out};
   }

   static variable_list backward(AutogradContext *ctx, variable_list grad_outs) {
      auto saved = ctx->get_saved_variables();
      auto rowptr = saved[0], col = saved[1], dinv = saved[2], colptr = saved[3], row_t = saved[4], out = saved[5];
      const Tensor dz = grad_outs[0].contiguous();
      check_float(dz, "grad_out");
      auto grad_bias = Variable(), grad_mat = Variable();
      const bool need_bias = ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(ctx->saved_data["bias_edge"].toInt());
      const bool need_mat = ctx->needs_input_grad(2);
      if (need_bias || need_mat) {
         // ONE pass over dZ (and the saved output, for the ReLU mask): gY = (dZ . [out > 0]) D^-1/2 written at the pitch the
         // backward's gather wants, and the bias gradient as a tall-skinny column sum (per-block partials + one fold,
         // deterministic) -- the four ATen passes this replaces included a 1.2 ms reduce_kernel on a 38 MB matrix
         c10::DeviceGuard guard(dz.device());
         Plan plan_t = ctx->saved_data["plan_t"].toTensorVector();
         if (need_mat && (!colptr.defined() || !row_t.defined())) {
            auto t = build_transpose(rowptr, col, Tensor(), rowptr.numel() - 1, false);
            colptr = t.colptr; row_t = t.row_t; plan_t.clear();
         }
         const int64_t N = dz.size(0), K = dz.size(1);
         const int64_t ld = need_mat ? gather_pitch(N, K, is_stream_plan(plan_t)) : K;
         Tensor gy_buf = need_mat ? at::empty({N, ld}, dz.options()) : Tensor();
         if (need_bias) grad_bias = at::empty({K}, dz.options());
         const Tensor mask = ctx->saved_data["relu"].toBool() ? out.contiguous() : Tensor();
         const Tensor scale = dinv.contiguous();
         const size_t ws = isplib_masked_scale_colsum_workspace_bytes(N, K);
         Tensor work = at::empty({(int64_t)ws}, dz.options().dtype(at::kByte));
         const int st = isplib_masked_scale_colsum_hip(N, K, dz.data_ptr<float>(), K, mask.defined() ? mask.data_ptr<float>() : nullptr, K,
                                                       scale.data_ptr<float>(), need_mat ? gy_buf.data_ptr<float>() : nullptr, ld,
                                                       need_bias ? grad_bias.data_ptr<float>() : nullptr, work.data_ptr(), ws,
                                                       current_stream(dz));
         check_status(st, "isplib_masked_scale_colsum_hip");
         if (need_mat) {
            const Tensor gy = gy_buf.narrow(1, 0, K);
            grad_mat = epilogue_spmm(colptr, row_t, plan_t, gy, gy, dinv, Tensor(), false);   // D (A^T + I) D dZ
         }
      }
      return {Variable(), Variable(), grad_mat, Variable(), Variable(), Variable(), Variable(), Variable(), grad_bias, Variable()};
   }
};

Tensor gcn_norm_spmm(Tensor rowptr, Tensor col, Tensor mat, Tensor dinv, optional<Tensor> colptr, optional<Tensor> row_t,
                     Plan plan, Plan plan_t, optional<Tensor> bias, bool relu) {
   return GcnNormSpmm::apply(rowptr, col, mat, dinv, colptr, row_t, plan, plan_t, bias, relu)[0];
}

// ---- op wrappers: csrc/fusedmm.cpp:520-563 -------------------------------------------------
Tensor fusedmm_spmm_add(optional<Tensor> opt_row, Tensor rowptr, Tensor col, optional<Tensor> opt_value,
                        optional<Tensor> opt_colptr, optional<Tensor> opt_csr2csc, Tensor mat,
                        optional<Tensor> value_index_select, optional<Tensor> row_index_select) {
   return SpmmSum::apply(opt_row, rowptr, col, opt_value, opt_colptr, opt_csr2csc, mat, value_index_select,
                         row_index_select, auto_plan(), auto_plan())[0];
}

// same operators fed with the per-graph schedule operands of A (forward) and A^T (backward)
Tensor fusedmm_spmm_planned(Tensor rowptr, Tensor col, optional<Tensor> opt_value, optional<Tensor> opt_colptr,
                            Tensor mat, optional<Tensor> value_t, optional<Tensor> row_t, Plan plan, Plan plan_t) {
   return SpmmSum::apply(optional<Tensor>(), rowptr, col, opt_value, opt_colptr, optional<Tensor>(), mat, value_t,
                         row_t, plan, plan_t)[0];
}

Tensor fusedmm_spmm_mean_planned(Tensor rowptr, Tensor col, optional<Tensor> opt_value, optional<Tensor> opt_colptr,
                                 Tensor mat, optional<Tensor> row_t, optional<Tensor> mean_value_t, Plan plan,
                                 Plan plan_t) {
   return SpmmMean::apply(optional<Tensor>(), rowptr, col, opt_value, optional<Tensor>(), opt_colptr,
                          optional<Tensor>(), mat, row_t, mean_value_t, plan, plan_t)[0];
}

std::tuple<Tensor, Tensor> fusedmm_spmm_max_planned(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat,
                                                    Plan plan);
std::tuple<Tensor, Tensor> fusedmm_spmm_min_planned(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat,
                                                    Plan plan);

Tensor fusedmm_spmm_mean(optional<Tensor> opt_row, Tensor rowptr, Tensor col, optional<Tensor> opt_value,
                         optional<Tensor> opt_rowcount, optional<Tensor> opt_colptr, optional<Tensor> opt_csr2csc,
                         Tensor mat, optional<Tensor> new_row, optional<Tensor> new_rowcount) {
   return SpmmMean::apply(opt_row, rowptr, col, opt_value, opt_rowcount, opt_colptr, opt_csr2csc, mat, new_row,
                          new_rowcount, auto_plan(), auto_plan())[0];
}

std::tuple<Tensor, Tensor> fusedmm_spmm_max(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat) {
   auto r = SpmmMinMax<R_MAX>::apply(rowptr, col, opt_value, mat, auto_plan());
   return std::make_tuple(r[0], r[1]);
}

std::tuple<Tensor, Tensor> fusedmm_spmm_max_planned(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat,
                                                    Plan plan) {
   auto r = SpmmMinMax<R_MAX>::apply(rowptr, col, opt_value, mat, plan);
   return std::make_tuple(r[0], r[1]);
}

std::tuple<Tensor, Tensor> fusedmm_spmm_min_planned(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat,
                                                    Plan plan) {
   auto r = SpmmMinMax<R_MIN>::apply(rowptr, col, opt_value, mat, plan);
   return std::make_tuple(r[0], r[1]);
}

// values only, no autograd node: what the plug-in calls for max / min when no gradient can be asked for (the patched matmul
// returns the tensor alone, isplib/__init__.py:143,145 + SURVEY 8a P1, so the positions would be computed and dropped)
Tensor fusedmm_spmm_max_values(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat, Plan plan) {
   OpTimer timer("FUSEDMM_SPMM_MAX_VALUES", mat);
   return std::get<0>(spmm_fw(rowptr, col, opt_value, mat, R_MAX, plan, /* want_arg = */ false));
}

Tensor fusedmm_spmm_min_values(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat, Plan plan) {
   OpTimer timer("FUSEDMM_SPMM_MIN_VALUES", mat);
   return std::get<0>(spmm_fw(rowptr, col, opt_value, mat, R_MIN, plan, /* want_arg = */ false));
}

std::tuple<Tensor, Tensor> fusedmm_spmm_min(Tensor rowptr, Tensor col, optional<Tensor> opt_value, Tensor mat) {
   auto r = SpmmMinMax<R_MIN>::apply(rowptr, col, opt_value, mat, auto_plan());
   return std::make_tuple(r[0], r[1]);
}

// introspection / housekeeping of the per-graph handle cache of the reference-schema operators
int64_t graph_cache_size() {
   std::lock_guard<std::mutex> lock(g_graph_mutex);
   return (int64_t)(g_graphs.size() + g_retired.size());
}
void graph_cache_clear() {
   std::lock_guard<std::mutex> lock(g_graph_mutex);
   for (auto &kv : g_graphs) isplib_graph_destroy(kv.second.handle);
   g_graphs.clear();
   destroy_retired_locked();
}

void performDummySpMM(int64_t flag) { performDummySpMM_hip(flag, (void *)c10::hip::getCurrentHIPStream().stream()); }

}  // namespace

// schemas as registered at csrc/fusedmm.cpp:565-570
TORCH_LIBRARY(isplib, m) {
   m.def("fusedmm_spmm(Tensor? row, Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr, Tensor? csr2csc, "
         "Tensor mat, Tensor? value_index_select, Tensor? row_index_select) -> Tensor",
         &fusedmm_spmm_add);
   m.def("fusedmm_spmm_mean(Tensor? row, Tensor rowptr, Tensor col, Tensor? value, Tensor? rowcount, Tensor? colptr, "
         "Tensor? csr2csc, Tensor mat, Tensor? new_row, Tensor? new_rowcount) -> Tensor",
         &fusedmm_spmm_mean);
   m.def("fusedmm_spmm_max(Tensor rowptr, Tensor col, Tensor? value, Tensor mat) -> (Tensor, Tensor)",
         &fusedmm_spmm_max);
   m.def("fusedmm_spmm_min(Tensor rowptr, Tensor col, Tensor? value, Tensor mat) -> (Tensor, Tensor)",
         &fusedmm_spmm_min);
   m.def("performDummySpMM(int flag) -> ()", &performDummySpMM);
   m.def("graph_cache_size() -> int", &graph_cache_size);
   m.def("graph_cache_clear() -> ()", &graph_cache_clear);
   // additions (not in the reference): the same operators fed with per-graph schedule operands
   // (plan = [] | [sliceptr] | [task_row, task_b, task_len, seg_off, lane_off_cpu]; plan_t: the same for A^T)
   m.def("fusedmm_spmm_planned(Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr, Tensor mat, Tensor? value_t, "
         "Tensor? row_t, Tensor[] plan, Tensor[] plan_t) -> Tensor",
         &fusedmm_spmm_planned);
   m.def("fusedmm_spmm_mean_planned(Tensor rowptr, Tensor col, Tensor? value, Tensor? colptr, Tensor mat, Tensor? row_t, "
         "Tensor? mean_value_t, Tensor[] plan, Tensor[] plan_t) -> Tensor",
         &fusedmm_spmm_mean_planned);
   m.def("fusedmm_spmm_max_planned(Tensor rowptr, Tensor col, Tensor? value, Tensor mat, Tensor[] plan) -> (Tensor, Tensor)",
         &fusedmm_spmm_max_planned);
   m.def("fusedmm_spmm_min_planned(Tensor rowptr, Tensor col, Tensor? value, Tensor mat, Tensor[] plan) -> (Tensor, Tensor)",
         &fusedmm_spmm_min_planned);
   m.def("fusedmm_spmm_max_values(Tensor rowptr, Tensor col, Tensor? value, Tensor mat, Tensor[] plan) -> Tensor", &fusedmm_spmm_max_values);
   m.def("fusedmm_spmm_min_values(Tensor rowptr, Tensor col, Tensor? value, Tensor mat, Tensor[] plan) -> Tensor", &fusedmm_spmm_min_values);
   m.def("gcn_norm_spmm(Tensor rowptr, Tensor col, Tensor mat, Tensor dinv, Tensor? colptr, Tensor? row_t, Tensor[] plan, "
         "Tensor[] plan_t, Tensor? bias, bool relu) -> Tensor",
         &gcn_norm_spmm);
}
