// stream_plan.hip -- native builder of the stream schedule's plan (isplib_stream_plan, include/isplib_hip.h) for hosts
// without torch: isplib_stream_plan_build_hip allocates and fills the plan's device arrays, isplib_stream_plan_free
// releases them.  The same construction as isplib_amd/plan.py: stream_plan_arrays (which the tests replay on the CPU):
//   1. rows over `chunk` edges are dealt edge by edge, round robin, to ceil(deg / chunk) virtual rows;
//   2. virtual rows, longest first (stable radix sort), are dealt to the streams in rounds of one row per stream, the
//      longest row of a round to the stream holding the fewest edges so far (stable sort of the loads per round);
//   3. every edge gets the key (stream, column slice, row of the stream); a stable radix sort of the keys yields each
//      stream's words in walking order -- slice by slice, row by row, CSR order inside;
//   4. the `streams` streams of a wave are interleaved step by step and padded to the longest.
// Integer work only; every sort is stable and nothing depends on atomics' order, so the plan is the same on every build.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/isplib_hip.h"
#include "common.h"

#include "prims.h"

namespace isplib {

static inline unsigned sp_grid(int64_t n) {
   int64_t b = (n + 255) / 256;
   if (b < 1) b = 1;
   if (b > 256 * 64) b = 256 * 64;
   return (unsigned)b;
}

static inline unsigned sp_bits(uint64_t n) {
   unsigned b = 1;
   while (b < 32 && ((uint64_t)1 << b) < n) b++;
   return b;
}

__global__ __launch_bounds__(256) void sp_row_chunks_kernel(int64_t m, int64_t chunk, const int64_t *__restrict__ rowptr,
                                                            int *__restrict__ nchunk, int *__restrict__ hub_flag) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
      const int64_t deg = rowptr[i + 1] - rowptr[i];
      const int64_t c = (deg + chunk - 1) / chunk;
      nchunk[i] = c < 1 ? 1 : (int)c;
      hub_flag[i] = c > 1 ? 1 : 0;
   }
}

// virtual rows of row i: first[i] .. first[i] + nchunk[i]; keys = ~length (ascending sort = longest first)
__global__ __launch_bounds__(256) void sp_vrows_kernel(int64_t m, const int64_t *__restrict__ rowptr, const int *__restrict__ nchunk,
                                                       const int *__restrict__ first, const int *__restrict__ hub_idx,
                                                       const int *__restrict__ hub_flag, int *__restrict__ vrow_row,
                                                       int *__restrict__ vlen, uint32_t *__restrict__ sort_key,
                                                       uint32_t *__restrict__ sort_val, int *__restrict__ vhub,
                                                       int *__restrict__ hub_row) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
      const int64_t deg = rowptr[i + 1] - rowptr[i];
      const int nc = nchunk[i], f = first[i];
      for (int ci = 0; ci < nc; ci++) {
         const int len = (int)((deg - ci + nc - 1) / nc);       // edges ci, ci + nc, ci + 2 nc, ...
         vrow_row[f + ci] = (int)i;
         vlen[f + ci] = len;
         sort_key[f + ci] = ~(uint32_t)len;
         sort_val[f + ci] = (uint32_t)(f + ci);
         vhub[f + ci] = nc > 1 ? 1 : 0;
      }
      if (hub_flag[i]) hub_row[hub_idx[i]] = (int)i;
   }
}

__global__ __launch_bounds__(256) void sp_iota_kernel(int64_t n, uint32_t *__restrict__ out) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint32_t)i;
}

// round r of the deal: the i-th longest virtual row of the round goes to the stream with the i-th smallest load
__global__ __launch_bounds__(256) void sp_deal_kernel(int64_t cnt, int round, const uint32_t *__restrict__ order,
                                                      const uint32_t *__restrict__ streams_by_load, const int *__restrict__ vlen,
                                                      int *__restrict__ sid, int *__restrict__ rnd, uint32_t *loads) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
      const uint32_t item = order[i], s = streams_by_load[i];
      sid[item] = (int)s;
      rnd[item] = round;
      loads[s] += (uint32_t)vlen[item];
   }
}

__global__ __launch_bounds__(256) void sp_wave_steps_kernel(int64_t nw, int streams, const uint32_t *__restrict__ loads,
                                                            int64_t *__restrict__ steps, int64_t *__restrict__ loads64) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += stride) {
      uint32_t mx = 0;
      for (int g = 0; g < streams; g++) {
         const uint32_t l = loads[w * streams + g];
         loads64[w * streams + g] = (int64_t)l;
         mx = l > mx ? l : mx;
      }
      steps[w] = (int64_t)mx;
   }
}

// rows of the edges by binary search in rowptr (no nnz-sized row array kept), then the sort key of every edge
__device__ __forceinline__ int64_t sp_row_of(int64_t e, int64_t m, const int64_t *rowptr) {
   int64_t lo = 0, hi = m;                                  // last row with rowptr[row] <= e
   while (hi - lo > 1) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (rowptr[mid] <= e) lo = mid; else hi = mid;
   }
   return lo;
}

// snake != 0 (max / min plans): a stream walks its rows forwards in even slices and backwards in odd ones, so the last row of
// slice s is the first row of slice s + 1 and that change of row -- an LDS swap in the max / min kernel -- disappears (one of
// `per` per slice: K=64 1.794 -> 1.786 ms, K=32 0.858 -> 0.845; every row's own word order is unchanged, the results are
// bit-identical).  Sum / mean plans keep the plain order: there a row that continues across a slice boundary without a flush
// changes the association of its sum, and the launch got 0.5 % slower.
__global__ __launch_bounds__(256) void sp_edge_keys_kernel(int64_t m, int64_t nnz, int64_t width, int slices, int per, int snake,
                                                           const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col,
                                                           const int *__restrict__ nchunk, const int *__restrict__ first,
                                                           const int *__restrict__ sid, const int *__restrict__ rnd,
                                                           uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
      const int64_t row = sp_row_of(e, m, rowptr);
      const int v = first[row] + (int)((e - rowptr[row]) % nchunk[row]);
      const int64_t sl = col[e] / width;
      keys[e] = (uint32_t)(((int64_t)sid[v] * slices + sl) * per + ((snake && (sl & 1)) ? per - 1 - rnd[v] : rnd[v]));
      vals[e] = (uint32_t)e;
   }
}

// padding words: column n (the gather reads 0 through the range check); the local row is the first row of the word's own
// stream (sum / mean: adding 0 changes nothing) or, pad_row >= 0 (max / min plans: a 0 could win), the kernel's spare row
__global__ __launch_bounds__(256) void sp_pad_kernel(int64_t n_words, int streams, int per, int pad_row, uint32_t null_col, int32_t *__restrict__ words,
                                                     int32_t *__restrict__ perm, float *__restrict__ vals) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
      const uint32_t lrow = pad_row >= 0 ? (uint32_t)pad_row : (uint32_t)(i % streams) * (uint32_t)per;
      words[i] = (int32_t)((lrow << 24) | null_col);
      perm[i] = -1;
      if (vals) vals[i] = 0.0f;
   }
}

__global__ __launch_bounds__(256) void sp_place_kernel(int64_t m, int64_t nnz, int streams, int per, const int64_t *__restrict__ rowptr,
                                                       const int64_t *__restrict__ col, const float *__restrict__ val,
                                                       const int *__restrict__ nchunk, const int *__restrict__ first,
                                                       const int *__restrict__ sid, const int *__restrict__ rnd,
                                                       const uint32_t *__restrict__ sorted_e, const int64_t *__restrict__ stream_start,
                                                       const int64_t *__restrict__ wave_step_off, int32_t *__restrict__ words,
                                                       int32_t *__restrict__ perm, float *__restrict__ vals) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += stride) {
      const int64_t e = (int64_t)sorted_e[q];
      const int64_t row = sp_row_of(e, m, rowptr);
      const int v = first[row] + (int)((e - rowptr[row]) % nchunk[row]);
      const int s = sid[v];
      const int64_t p = q - stream_start[s];                 // position inside the stream
      const int64_t idx = (wave_step_off[s / streams] + p) * streams + s % streams;
      const uint32_t lrow = (uint32_t)((s % streams) * per + rnd[v]);
      words[idx] = (int32_t)((lrow << 24) | (uint32_t)col[e]);
      perm[idx] = (int32_t)e;
      if (vals) vals[idx] = val[e];
   }
}

__global__ __launch_bounds__(256) void sp_wave_rows_kernel(int64_t nv, int streams, int per, int rows_per_wave,
                                                           const int *__restrict__ vrow_row, const int *__restrict__ vhub,
                                                           const int *__restrict__ part_of, const int *__restrict__ sid,
                                                           const int *__restrict__ rnd, int32_t *__restrict__ wave_row,
                                                           int32_t *__restrict__ wave_part, const int *__restrict__ first,
                                                           const int *__restrict__ hub_idx, int32_t *__restrict__ hub_off) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
      const int s = sid[v];
      const int64_t at = (int64_t)(s / streams) * rows_per_wave + (s % streams) * per + rnd[v];
      const int row = vrow_row[v];
      wave_row[at] = row;
      wave_part[at] = vhub[v] ? part_of[v] : -1;
      if (vhub[v] && first[row] == (int)v) hub_off[hub_idx[row]] = part_of[v];    // first piece of a hub row
   }
}

__global__ __launch_bounds__(256) void sp_gather_vals_kernel(int64_t n_words, const int32_t *__restrict__ perm, const float *__restrict__ src,
                                                             float *__restrict__ vals) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) vals[i] = perm[i] >= 0 ? src[perm[i]] : 0.0f;
}

struct SpTemp {                                             // frees every temporary on every exit path
   std::vector<void *> ptrs;
   ~SpTemp() { for (void *p : ptrs) (void)hipFree(p); }
   template <class T> bool alloc(T **out, size_t count) {
      void *p = nullptr;
      if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return false; }
      ptrs.push_back(p);
      *out = (T *)p;
      return true;
   }
};

template <class T> static bool sp_alloc_out(T **out, size_t count) {
   void *p = nullptr;
   if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return false; }
   *out = (T *)p;
   return true;
}

static hipError_t sp_sort(void *temp, size_t &temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                          size_t n, unsigned bits, hipStream_t st) {
   return sort_pairs_u32(temp, temp_bytes, kin, kout, vin, vout, n, 0u, bits, st);
}

}  // namespace isplib

using namespace isplib;

extern "C" void isplib_stream_plan_free(isplib_stream_plan *plan) {
   if (!plan) return;
   (void)hipDeviceSynchronize();                           // nothing of the plan's may still be in flight
   (void)hipFree(const_cast<int32_t *>(plan->words)); (void)hipFree(const_cast<float *>(plan->vals));
   (void)hipFree(const_cast<int64_t *>(plan->wave_step_off)); (void)hipFree(const_cast<int32_t *>(plan->wave_row));
   (void)hipFree(const_cast<int32_t *>(plan->wave_part)); (void)hipFree(const_cast<int32_t *>(plan->hub_row));
   (void)hipFree(const_cast<int32_t *>(plan->hub_off)); (void)hipFree(const_cast<int32_t *>(plan->perm));
   memset(plan, 0, sizeof(*plan));
}

extern "C" int isplib_stream_plan_set_values_hip(isplib_stream_plan *plan, const float *val, void *stream) {
   clear_error();
   if (!plan || !plan->perm) return fail(ISPLIB_FAIL, "isplib_stream_plan_set_values_hip: not a plan built by isplib_stream_plan_build_hip");
   const int64_t n_words = plan->n_steps * plan->streams;
   if (!val) {                                              // unit weights: the kernel then skips the stream
      if (plan->vals) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(const_cast<float *>(plan->vals)); plan->vals = nullptr; }
      return ISPLIB_SUCCESS;
   }
   if (!plan->vals) {
      float *v = nullptr;
      if (!sp_alloc_out(&v, (size_t)n_words)) return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_set_values_hip: device allocation failed");
      plan->vals = v;
   }
   if (n_words > 0) {
      hipLaunchKernelGGL(sp_gather_vals_kernel, dim3(sp_grid(n_words)), dim3(256), 0, (hipStream_t)stream, n_words, plan->perm, val,
                         const_cast<float *>(plan->vals));
      return check_launch("sp_gather_vals_kernel");
   }
   return ISPLIB_SUCCESS;
}

static int stream_plan_build(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col, const float *val,
                             int streams, int rpw, int resident, int slices, int chunk, int waves_per_gen,
                             isplib_stream_plan *out, void *stream, int pad_row = -1) {
   if (!out) return fail(ISPLIB_FAIL, "isplib_stream_plan_build_hip: out is NULL");
   memset(out, 0, sizeof(*out));
   if (m <= 0 || n <= 0 || nnz < 0 || !rowptr || (nnz > 0 && !col)) return fail(ISPLIB_FAIL, "isplib_stream_plan_build_hip: bad operand");
   if (n >= (1LL << 24) || nnz >= (1LL << 31) || m >= (1LL << 31)) return fail(ISPLIB_FAIL, "isplib_stream_plan_build_hip: n < 2^24, nnz < 2^31, m < 2^31 required");
   if (slices < 1 || slices > 4096 || chunk < 1 || chunk >= (1 << 24)) return fail(ISPLIB_FAIL, "isplib_stream_plan_build_hip: slices in [1, 4096], chunk in [1, 2^24)");
   if (waves_per_gen <= 0) waves_per_gen = resident;
   const int per = rpw / streams;
   hipStream_t st = (hipStream_t)stream;
   SpTemp T;
   // 1. virtual rows
   int *nchunk, *hub_flag, *first, *hub_idx;
   if (!T.alloc(&nchunk, (size_t)m) || !T.alloc(&hub_flag, (size_t)m) || !T.alloc(&first, (size_t)m + 1) || !T.alloc(&hub_idx, (size_t)m + 1))
      return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   hipLaunchKernelGGL(sp_row_chunks_kernel, dim3(sp_grid(m)), dim3(256), 0, st, m, (int64_t)chunk, rowptr, nchunk, hub_flag);
   int rc = check_launch("sp_row_chunks_kernel");
   if (rc) return rc;
   size_t scan_bytes = 0;
   ISPLIB_HIP_TRY(scan_exclusive_i32(nullptr, scan_bytes, nullptr, nullptr, (size_t)m, st));
   void *scan_tmp;
   {
      char *p;
      if (!T.alloc(&p, scan_bytes + 256)) return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
      scan_tmp = p;
   }
   size_t sb = scan_bytes;
   ISPLIB_HIP_TRY(scan_exclusive_i32(scan_tmp, sb, (const int *)nchunk, first, (size_t)m, st));
   sb = scan_bytes;
   ISPLIB_HIP_TRY(scan_exclusive_i32(scan_tmp, sb, (const int *)hub_flag, hub_idx, (size_t)m, st));
   int last[4];                                             // first[m-1], nchunk[m-1], hub_idx[m-1], hub_flag[m-1]
   ISPLIB_HIP_TRY(hipMemcpyAsync(&last[0], first + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
   ISPLIB_HIP_TRY(hipMemcpyAsync(&last[1], nchunk + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
   ISPLIB_HIP_TRY(hipMemcpyAsync(&last[2], hub_idx + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
   ISPLIB_HIP_TRY(hipMemcpyAsync(&last[3], hub_flag + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
   ISPLIB_HIP_TRY(hipStreamSynchronize(st));
   const int64_t nv = (int64_t)last[0] + last[1], n_hub = (int64_t)last[2] + last[3];
   const int64_t per_gen = (int64_t)waves_per_gen * rpw;
   const int64_t gens = (nv + per_gen - 1) / per_gen;
   const int64_t nw = gens * waves_per_gen, ns = nw * streams;
   if (gens > 4096 || ns >= (1LL << 31) || (double)ns * slices * per >= 4294967295.0)
      return fail(ISPLIB_FAIL, "isplib_stream_plan_build_hip: graph too large for this geometry (32-bit stream keys)");
   int *vrow_row, *vlen, *vhub, *part_of, *sid, *rnd;
   uint32_t *k_in, *k_out, *v_in, *v_out, *loads, *lk_out, *lv_in, *lv_out;
   const size_t big = (size_t)(nnz > nv ? nnz : nv) + 1;
   if (!T.alloc(&vrow_row, (size_t)nv) || !T.alloc(&vlen, (size_t)nv) || !T.alloc(&vhub, (size_t)nv) || !T.alloc(&part_of, (size_t)nv + 1) ||
       !T.alloc(&sid, (size_t)nv) || !T.alloc(&rnd, (size_t)nv) || !T.alloc(&k_in, big) || !T.alloc(&k_out, big) || !T.alloc(&v_in, big) ||
       !T.alloc(&v_out, big) || !T.alloc(&loads, (size_t)ns) || !T.alloc(&lk_out, (size_t)ns) || !T.alloc(&lv_in, (size_t)ns) ||
       !T.alloc(&lv_out, (size_t)ns))
      return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   int32_t *hub_row = nullptr, *hub_off = nullptr;
   if (!sp_alloc_out(&hub_row, (size_t)n_hub) || !sp_alloc_out(&hub_off, (size_t)n_hub + 1)) {
      (void)hipFree(hub_row); (void)hipFree(hub_off);
      return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   }
   out->hub_row = hub_row; out->hub_off = hub_off;         // from here on isplib_stream_plan_free(out) releases what was allocated
#define SP_FAIL(code, msg) do { const int c_ = fail(code, msg); isplib_stream_plan_free(out); return c_; } while (0)
#define SP_TRY(expr) do { const hipError_t e_ = (expr); if (e_ != hipSuccess) { const int c_ = hip_fail(e_, #expr); isplib_stream_plan_free(out); return c_; } } while (0)
#define SP_LAUNCHED(name) do { const int c_ = check_launch(name); if (c_) { isplib_stream_plan_free(out); return c_; } } while (0)
   hipLaunchKernelGGL(sp_vrows_kernel, dim3(sp_grid(m)), dim3(256), 0, st, m, rowptr, nchunk, first, hub_idx, hub_flag, vrow_row, vlen, k_in, v_in,
                      vhub, hub_row);
   SP_LAUNCHED("sp_vrows_kernel");
   size_t sbv = 0;
   SP_TRY(scan_exclusive_i32(nullptr, sbv, nullptr, nullptr, (size_t)nv, st));
   char *scan_tmp2;
   if (!T.alloc(&scan_tmp2, sbv + 256)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   SP_TRY(scan_exclusive_i32(scan_tmp2, sbv, (const int *)vhub, part_of, (size_t)nv, st));
   int part_last[2];
   SP_TRY(hipMemcpyAsync(&part_last[0], part_of + (nv - 1), sizeof(int), hipMemcpyDeviceToHost, st));
   SP_TRY(hipMemcpyAsync(&part_last[1], vhub + (nv - 1), sizeof(int), hipMemcpyDeviceToHost, st));
   // 2. longest first, then the deal in rounds
   size_t sort_bytes = 0, sort_small = 0;
   SP_TRY(sp_sort(nullptr, sort_bytes, nullptr, nullptr, nullptr, nullptr, big, 32, st));
   SP_TRY(sp_sort(nullptr, sort_small, nullptr, nullptr, nullptr, nullptr, (size_t)ns, 32, st));
   char *sort_tmp;
   if (!T.alloc(&sort_tmp, (sort_bytes > sort_small ? sort_bytes : sort_small) + 256)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   size_t tb = sort_bytes;
   SP_TRY(sp_sort(sort_tmp, tb, k_in, k_out, v_in, v_out, (size_t)nv, 32, st));      // v_out = virtual rows, longest first
   SP_TRY(hipMemsetAsync(loads, 0, (size_t)ns * sizeof(uint32_t), st));
   hipLaunchKernelGGL(sp_iota_kernel, dim3(sp_grid(ns)), dim3(256), 0, st, ns, lv_in);
   SP_LAUNCHED("sp_iota_kernel");
   const int rounds = (int)((nv + ns - 1) / ns);
   if (rounds > per) SP_FAIL(ISPLIB_FAIL, "isplib_stream_plan_build_hip: internal: more rounds than rows per stream");
   for (int r = 0; r < rounds; r++) {
      const int64_t cnt = nv - (int64_t)r * ns < ns ? nv - (int64_t)r * ns : ns;
      tb = sort_small;
      SP_TRY(sp_sort(sort_tmp, tb, loads, lk_out, lv_in, lv_out, (size_t)ns, 32, st));
      hipLaunchKernelGGL(sp_deal_kernel, dim3(sp_grid(cnt)), dim3(256), 0, st, cnt, r, v_out + (int64_t)r * ns, lv_out, vlen, sid, rnd, loads);
      SP_LAUNCHED("sp_deal_kernel");
   }
   // 3. stream starts, wave steps
   int64_t *steps, *loads64, *stream_start, *wave_step_off = nullptr;
   if (!T.alloc(&steps, (size_t)nw) || !T.alloc(&loads64, (size_t)ns) || !T.alloc(&stream_start, (size_t)ns) || !sp_alloc_out(&wave_step_off, (size_t)nw + 1))
      SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   out->wave_step_off = wave_step_off;
   hipLaunchKernelGGL(sp_wave_steps_kernel, dim3(sp_grid(nw)), dim3(256), 0, st, nw, streams, loads, steps, loads64);
   SP_LAUNCHED("sp_wave_steps_kernel");
   size_t s64 = 0;
   SP_TRY(scan_exclusive_i64(nullptr, s64, nullptr, nullptr, (size_t)ns, st));
   char *scan_tmp3;
   if (!T.alloc(&scan_tmp3, s64 + 256)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   size_t s64b = s64;
   SP_TRY(scan_exclusive_i64(scan_tmp3, s64b, (const int64_t *)loads64, stream_start, (size_t)ns, st));
   s64b = s64;
   SP_TRY(scan_exclusive_i64(scan_tmp3, s64b, (const int64_t *)steps, wave_step_off, (size_t)nw, st));
   int64_t tail[2];
   SP_TRY(hipMemcpyAsync(&tail[0], wave_step_off + (nw - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
   SP_TRY(hipMemcpyAsync(&tail[1], steps + (nw - 1), sizeof(int64_t), hipMemcpyDeviceToHost, st));
   SP_TRY(hipStreamSynchronize(st));
   const int64_t n_steps = tail[0] + tail[1];
   const int64_t n_parts = (int64_t)part_last[0] + part_last[1];
   SP_TRY(hipMemcpyAsync(wave_step_off + nw, &n_steps, sizeof(int64_t), hipMemcpyHostToDevice, st));
   const int32_t n_parts32 = (int32_t)n_parts;
   SP_TRY(hipMemcpyAsync(hub_off + n_hub, &n_parts32, sizeof(int32_t), hipMemcpyHostToDevice, st));
   // 4. the words
   const int64_t n_words = n_steps * streams;
   int32_t *words = nullptr, *perm = nullptr, *wave_row = nullptr, *wave_part = nullptr;
   float *vals = nullptr;
   if (!sp_alloc_out(&words, (size_t)n_words)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   out->words = words;
   if (!sp_alloc_out(&perm, (size_t)n_words)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   out->perm = perm;
   if (val) {
      if (!sp_alloc_out(&vals, (size_t)n_words)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
      out->vals = vals;
   }
   if (!sp_alloc_out(&wave_row, (size_t)nw * rpw)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   out->wave_row = wave_row;
   if (!sp_alloc_out(&wave_part, (size_t)nw * rpw)) SP_FAIL(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_hip: device allocation failed");
   out->wave_part = wave_part;
   SP_TRY(hipMemsetAsync(wave_row, 0xFF, (size_t)nw * rpw * sizeof(int32_t), st));
   SP_TRY(hipMemsetAsync(wave_part, 0xFF, (size_t)nw * rpw * sizeof(int32_t), st));
   hipLaunchKernelGGL(sp_wave_rows_kernel, dim3(sp_grid(nv)), dim3(256), 0, st, nv, streams, per, rpw, vrow_row, vhub, part_of, sid, rnd, wave_row,
                      wave_part, first, hub_idx, hub_off);
   SP_LAUNCHED("sp_wave_rows_kernel");
   if (n_words > 0) {
      hipLaunchKernelGGL(sp_pad_kernel, dim3(sp_grid(n_words)), dim3(256), 0, st, n_words, streams, per, pad_row, (uint32_t)n, words, perm, vals);
      SP_LAUNCHED("sp_pad_kernel");
   }
   if (nnz > 0) {
      const int64_t width = (n + slices - 1) / slices;
      hipLaunchKernelGGL(sp_edge_keys_kernel, dim3(sp_grid(nnz)), dim3(256), 0, st, m, nnz, width, slices, per, pad_row >= 0 ? 1 : 0, rowptr, col, nchunk, first,
                         sid, rnd, k_in, v_in);
      SP_LAUNCHED("sp_edge_keys_kernel");
      tb = sort_bytes;
      SP_TRY(sp_sort(sort_tmp, tb, k_in, k_out, v_in, v_out, (size_t)nnz, sp_bits((uint64_t)ns * slices * per + 1), st));
      hipLaunchKernelGGL(sp_place_kernel, dim3(sp_grid(nnz)), dim3(256), 0, st, m, nnz, streams, per, rowptr, col, val, nchunk, first, sid, rnd, v_out,
                         stream_start, wave_step_off, words, perm, vals);
      SP_LAUNCHED("sp_place_kernel");
   }
   SP_TRY(hipStreamSynchronize(st));                       // the temporaries go out of scope here
#undef SP_FAIL
#undef SP_TRY
#undef SP_LAUNCHED
   out->rows = m; out->cols = n; out->slices = slices; out->gens = (int32_t)gens; out->waves_per_gen = waves_per_gen;
   out->rows_per_wave = rpw; out->streams = streams; out->chunk = chunk;
   out->n_steps = n_steps; out->n_parts = n_parts; out->n_hub = n_hub;
   return ISPLIB_SUCCESS;
}

extern "C" int isplib_stream_plan_build_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                            const float *val, int streams, int slices, int chunk, int waves_per_gen,
                                            isplib_stream_plan *out, void *stream) {
   clear_error();
   int rpw = 0, resident = 0;
   if (isplib_spmm_stream_geometry(streams, &rpw, &resident) != ISPLIB_SUCCESS) return ISPLIB_FAIL;
   return stream_plan_build(m, n, nnz, rowptr, col, val, streams, rpw, resident, slices, chunk, waves_per_gen, out, stream);
}

// 1 if some row's columns do not ascend (duplicates are fine)
__global__ __launch_bounds__(256) void sp_unsorted_kernel(int64_t m, int64_t nnz, const int64_t *__restrict__ rowptr,
                                                          const int64_t *__restrict__ col, int *__restrict__ flag) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   bool bad = false;
   for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; e < nnz; e += stride)
      if (col[e] < col[e - 1] && e > rowptr[sp_row_of(e, m, rowptr)]) bad = true;
   if (bad) atomicOr(flag, 1);
}

extern "C" int isplib_stream_plan_build_minmax_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                                   const float *val, int streams, int slices, int chunk, int waves_per_gen,
                                                   isplib_stream_plan *out, void *stream) {
   clear_error();
   int rpw = 0, resident = 0;
   if (isplib_spmm_stream_minmax_geometry(streams, &rpw, &resident) != ISPLIB_SUCCESS) return ISPLIB_FAIL;
   if (out) memset(out, 0, sizeof(*out));
   if (m <= 0 || nnz < 0 || !rowptr || (nnz > 0 && !col)) return fail(ISPLIB_FAIL, "isplib_stream_plan_build_minmax_hip: bad operand");
   // the kernel's tie rule (first strictly better candidate in stream order = lowest CSR position) holds for rows whose
   // columns ascend: anything else is refused here and stays on the task list
   if (nnz > 1) {
      int *flag = nullptr, host = 1;
      if (hipMalloc((void **)&flag, 256) != hipSuccess) { (void)hipGetLastError(); return fail(ISPLIB_NOT_ENOUGH_MEM, "isplib_stream_plan_build_minmax_hip: device allocation failed"); }
      hipStream_t st = (hipStream_t)stream;
      bool ok = hipMemsetAsync(flag, 0, sizeof(int), st) == hipSuccess;
      if (ok) {
         hipLaunchKernelGGL(sp_unsorted_kernel, dim3(sp_grid(nnz)), dim3(256), 0, st, m, nnz, rowptr, col, flag);
         ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
              hipStreamSynchronize(st) == hipSuccess;
      }
      (void)hipFree(flag);
      if (!ok) return hip_fail(hipGetLastError(), "isplib_stream_plan_build_minmax_hip: sortedness check");
      if (host) return fail(ISPLIB_FAIL, "isplib_stream_plan_build_minmax_hip: rows are not column-sorted (use the task list)");
   }
   return stream_plan_build(m, n, nnz, rowptr, col, val, streams, rpw, resident, slices, chunk, waves_per_gen, out, stream, /* pad_row = the spare row */ rpw);
}

// A plan for fusedMM_csr_udef_stream_hip (fusedmm_stream.hip): the max / min kernel's shape -- two LDS planes per row, so half
// the rows per wave of a sum plan, and padding words that carry the kernel's spare row -- at the generic kernel's own geometry
// (isplib_fusedmm_stream_geometry).  No requirement on the order of a row's columns: the words are summed, not compared.
extern "C" int isplib_stream_plan_build_fusedmm_hip(int64_t m, int64_t n, int64_t nnz, const int64_t *rowptr, const int64_t *col,
                                                    int streams, int slices, int chunk, int waves_per_gen,
                                                    isplib_stream_plan *out, void *stream) {
   clear_error();
   int rpw = 0, resident = 0;
   if (isplib_fusedmm_stream_geometry(streams, &rpw, &resident) != ISPLIB_SUCCESS) return ISPLIB_FAIL;
   return stream_plan_build(m, n, nnz, rowptr, col, nullptr, streams, rpw, resident, slices, chunk, waves_per_gen, out, stream, /* pad_row = the spare row */ rpw);
}
