// c_abi_demo.cpp -- the C ABI used from a plain C++/HIP host: no torch, no Python.
//
// This is what a non-Python caller of the reference's fusedMM_csr (csrc/fusedMM.h:77-99) does after
// switching to the device entry point: own the device buffers, call fusedMM_csr_hip with the reference's
// argument pattern (pntrb = rowptr, pntre = rowptr + 1, csrc/fusedmm.cpp:198) plus a stream, and -- for the
// fast path -- build the per-graph task plan once and call fusedMM_csr_tasks_hip.
// The result is checked against a sequential host loop (sum: 1e-5 * sum|val*x|; max: bit-exact incl. arg).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_abi_demo.cpp -Lisplib_amd -lisplib_hip \
//         -Wl,-rpath,$PWD/isplib_amd -o examples/c_abi_demo && examples/c_abi_demo
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "isplib_hip.h"

#define HIP_OK(x)                                                                         \
   do {                                                                                   \
      hipError_t e_ = (x);                                                                \
      if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
   } while (0)
#define ISP_OK(x)                                                                         \
   do {                                                                                   \
      int s_ = (x);                                                                       \
      if (s_ != ISPLIB_SUCCESS) { printf("isplib status %d (%s) at line %d\n", s_, isplib_hip_last_error(), __LINE__); return 3; } \
   } while (0)

template <class T> static T *to_device(const std::vector<T> &h) {
   T *d = nullptr;
   if (hipMalloc(&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
   (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
   return d;
}

int main() {
   const int64_t m = 3000, n = 3000, k = 64;
   uint64_t rng = 88172645463325252ull;
   auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
   std::vector<int64_t> rowptr(m + 1, 0), col;
   std::vector<float> val;
   for (int64_t i = 0; i < m; i++) {
      const int deg = i == 7 ? 2500 : (i % 97 == 0 ? 0 : 40 + (int)(next() % 200));   // a hub, some empty rows
      std::vector<char> seen(n, 0);
      std::vector<int64_t> cs;
      while ((int)cs.size() < deg) { const int64_t c = (int64_t)(next() % n); if (!seen[c]) { seen[c] = 1; cs.push_back(c); } }
      for (int64_t c = 0; c < n; c++) if (seen[c]) { col.push_back(c); val.push_back((float)(next() % 1000) / 1000.0f); }
      rowptr[i + 1] = (int64_t)col.size();
   }
   const int64_t nnz = (int64_t)col.size();
   std::vector<float> x((size_t)n * k);
   for (auto &v : x) v = (float)((int)(next() % 7) - 3);                              // integer features: ties for max

   int64_t *d_rowptr = to_device(rowptr), *d_col = to_device(col);
   float *d_val = to_device(val), *d_x = to_device(x);
   float *d_out = nullptr; int64_t *d_arg = nullptr;
   HIP_OK(hipMalloc(&d_out, (size_t)m * k * sizeof(float)));
   HIP_OK(hipMalloc(&d_arg, (size_t)m * k * sizeof(int64_t)));
   hipStream_t st; HIP_OK(hipStreamCreate(&st));

   // host reference: the sequential loop of the oracle
   std::vector<float> ref_sum((size_t)m * k, 0.f), mag((size_t)m * k, 0.f), ref_max((size_t)m * k, 0.f);
   std::vector<int64_t> ref_arg((size_t)m * k, nnz);
   for (int64_t i = 0; i < m; i++)
      for (int64_t c = 0; c < k; c++) {
         float s = 0.f, a = 0.f, best = -FLT_MAX; int64_t bj = nnz;
         for (int64_t j = rowptr[i]; j < rowptr[i + 1]; j++) {
            const float t = val[j] * x[(size_t)col[j] * k + c];
            s += t; a += std::fabs(t);
            if (t > best) { best = t; bj = j; }
         }
         ref_sum[i * k + c] = s; mag[i * k + c] = a;
         ref_max[i * k + c] = rowptr[i + 1] > rowptr[i] ? best : 0.f; ref_arg[i * k + c] = bj;
      }
   std::vector<float> out((size_t)m * k); std::vector<int64_t> arg((size_t)m * k);
   auto check_sum = [&](const char *what) {
      (void)hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost);
      for (size_t i = 0; i < out.size(); i++)
         if (std::fabs(out[i] - ref_sum[i]) > 1e-5f * mag[i] + 1e-30f) { printf("%s: sum mismatch at %zu: %g vs %g\n", what, i, out[i], ref_sum[i]); return false; }
      return true;
   };
   auto check_max = [&](const char *what) {
      (void)hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost);
      (void)hipMemcpy(arg.data(), d_arg, arg.size() * sizeof(int64_t), hipMemcpyDeviceToHost);
      for (size_t i = 0; i < out.size(); i++)
         if (out[i] != ref_max[i] || arg[i] != ref_arg[i]) { printf("%s: max mismatch at %zu: %g/%lld vs %g/%lld\n", what, i, out[i], (long long)arg[i], ref_max[i], (long long)ref_arg[i]); return false; }
      return true;
   };

   // 1. reference-signature entry point (no preparation)
   ISP_OK(fusedMM_csr_hip(ISPLIB_MSG_SPMM_SUM, m, n, k, 1.0f, nnz, m, n, d_val, d_col, d_rowptr, d_rowptr + 1, nullptr, k,
                          d_x, k, 0.0f, d_out, k, nullptr, st));
   HIP_OK(hipStreamSynchronize(st));
   if (!check_sum("fusedMM_csr_hip")) return 1;
   ISP_OK(fusedMM_csr_hip(ISPLIB_MSG_SPMM_MAX, m, n, k, 1.0f, nnz, m, n, d_val, d_col, d_rowptr, d_rowptr + 1, nullptr, k,
                          d_x, k, 0.0f, d_out, k, d_arg, st));
   HIP_OK(hipStreamSynchronize(st));
   if (!check_max("fusedMM_csr_hip")) return 1;

   // 2. task-list schedule: plan once per graph, then every SpMM
   const int S = 8;
   int64_t *d_slices = nullptr; int32_t *d_flag = nullptr, *d_seg = nullptr;
   HIP_OK(hipMalloc(&d_slices, isplib_spmm_slices_bytes(m, S)));
   HIP_OK(hipMalloc(&d_flag, sizeof(int32_t)));
   HIP_OK(hipMalloc(&d_seg, ((size_t)m * S + 1) * sizeof(int32_t)));
   ISP_OK(isplib_spmm_slices_build_hip(m, n, nnz, d_rowptr, d_rowptr + 1, d_col, S, d_slices, d_flag, st));
   int32_t unsorted = 1;
   HIP_OK(hipMemcpyAsync(&unsorted, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
   HIP_OK(hipStreamSynchronize(st));
   if (unsorted) { printf("rows are not column-sorted\n"); return 1; }
   const size_t pws = isplib_spmm_tasks_plan_workspace_bytes(m, S);
   void *d_pws = nullptr; HIP_OK(hipMalloc(&d_pws, pws));
   isplib_task_plan_info info;
   ISP_OK(isplib_spmm_tasks_count_hip(m, d_rowptr, d_rowptr + 1, d_slices, S, 256, 64, d_seg, d_pws, pws, &info, st));
   int32_t *d_trow = nullptr, *d_tlen = nullptr; int64_t *d_tb = nullptr;
   HIP_OK(hipMalloc(&d_trow, (size_t)info.n_tasks * 4 + 16)); HIP_OK(hipMalloc(&d_tlen, (size_t)info.n_tasks * 4 + 16));
   HIP_OK(hipMalloc(&d_tb, (size_t)info.n_tasks * 8 + 16));
   ISP_OK(isplib_spmm_tasks_fill_hip(m, d_rowptr, d_rowptr + 1, d_slices, &info, d_seg, d_trow, d_tb, d_tlen, st));
   const size_t ws = isplib_spmm_tasks_workspace_bytes(ISPLIB_MSG_SPMM_MAX, info.n_tasks, k);
   void *d_ws = nullptr; HIP_OK(hipMalloc(&d_ws, ws));
   HIP_OK(hipMemsetAsync(d_out, 0xff, (size_t)m * k * sizeof(float), st));
   int32_t *d_col32 = nullptr; HIP_OK(hipMalloc(&d_col32, (size_t)nnz * sizeof(int32_t)));      // once per graph: 4-byte ids
   ISP_OK(isplib_pack_indices_hip(nnz, d_col, d_col32, st));
   ISP_OK(fusedMM_csr_tasks_hip(ISPLIB_MSG_SPMM_SUM, m, n, k, nnz, d_val, d_col, d_col32, d_rowptr, d_rowptr + 1, info.n_tasks, d_trow,
                                d_tb, d_tlen, d_seg, S, info.lane_off, d_x, k, d_out, k, nullptr, d_ws, ws, st));
   HIP_OK(hipStreamSynchronize(st));
   if (!check_sum("fusedMM_csr_tasks_hip")) return 1;
   ISP_OK(fusedMM_csr_tasks_hip(ISPLIB_MSG_SPMM_MAX, m, n, k, nnz, d_val, d_col, /*indx32*/ nullptr, d_rowptr, d_rowptr + 1, info.n_tasks, d_trow,
                                d_tb, d_tlen, d_seg, S, info.lane_off, d_x, k, d_out, k, d_arg, d_ws, ws, st));
   HIP_OK(hipStreamSynchronize(st));
   if (!check_max("fusedMM_csr_tasks_hip")) return 1;

   // 2b. the same through an isplib_graph handle: the library owns the plan, the packed ids and the workspace
   isplib_graph *graph = nullptr;
   ISP_OK(isplib_graph_create(m, n, nnz, d_rowptr, d_col, d_val, &graph));
   ISP_OK(isplib_graph_set_slices(graph, S));                       // -1 would apply isplib_suggest_slices
   HIP_OK(hipMemsetAsync(d_out, 0xff, (size_t)m * k * sizeof(float), st));
   ISP_OK(isplib_graph_spmm(graph, ISPLIB_MSG_SPMM_SUM, k, d_x, k, d_out, k, nullptr, st));
   HIP_OK(hipStreamSynchronize(st));
   if (!check_sum("isplib_graph_spmm")) return 1;
   ISP_OK(isplib_graph_spmm(graph, ISPLIB_MSG_SPMM_MAX, k, d_x, k, d_out, k, d_arg, st));
   HIP_OK(hipStreamSynchronize(st));
   if (!check_max("isplib_graph_spmm")) return 1;
   isplib_graph_destroy(graph);

   // 3. error behaviour: the reference's status codes -- a flag value csrc/fusedMM.h does not define, and a
   //    user-defined stage (function pointers cannot cross to the device; see fusedMM_csr_udef_hip's menu)
   if (fusedMM_csr_hip(0x11108, m, n, k, 1.0f, nnz, m, n, d_val, d_col, d_rowptr, d_rowptr + 1, nullptr, k, d_x, k, 0.0f,
                       d_out, k, nullptr, st) != ISPLIB_NO_OPT_IMPL) { printf("expected NO_OPT_IMPL\n"); return 1; }
   if (fusedMM_csr_hip(0x1110F, m, n, k, 1.0f, nnz, m, n, d_val, d_col, d_rowptr, d_rowptr + 1, nullptr, k, d_x, k, 0.0f,
                       d_out, k, nullptr, st) != ISPLIB_UNDEFINED_USER_FUNCTION) { printf("expected UNDEFINED_USER_FUNCTION\n"); return 1; }
   printf("c_abi_demo ok: m=%lld nnz=%lld k=%lld, %lld tasks; sum within 1e-5, max/arg bit-exact, both entry points\n",
          (long long)m, (long long)nnz, (long long)k, (long long)info.n_tasks);
   return 0;
}
