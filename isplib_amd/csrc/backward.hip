// backward.hip -- the two backward kernels that are not "SpMM on A^T":
//   * fused scatter for SpMM-max/min (replaces the 6-8 ATen passes of
//     reference csrc/fusedmm.cpp:410-451 / 477-517),
//   * SDDMM-style dA for sum/mean (the call the reference leaves commented out,
//     csrc/fusedmm.cpp:270,351).
// dX of sum/mean needs no kernel of its own: it is fusedMM_csr_hip on the CSC
// operands, exactly as the reference does at csrc/fusedmm.cpp:285,375.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/isplib_hip.h"
#include "common.h"
#include "gather.h"

namespace isplib {

int g_sddmm_panel_cols = 0;   // experimental knob (isplib_hip_tune_experimental(12, cols)): column-panel width of the task-list SDDMM; 0 = whole rows

// One thread per (row, feature) element of arg/grad_out; consecutive lanes hold
// consecutive features of one row, so the reads are coalesced and each wave's
// atomics to one destination row are contiguous where args agree.
__global__ __launch_bounds__(256) void minmax_bw_kernel(int64_t total, int64_t k, int64_t nnz,
                                                        const int64_t *__restrict__ indx,
                                                        const float *__restrict__ val,
                                                        const float *__restrict__ mat,
                                                        const int64_t *__restrict__ arg,
                                                        const float *__restrict__ grad_out,
                                                        float *grad_mat, float *grad_val) {
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int64_t a = arg[t];
      if (a < 0 || a >= nnz) continue;            // nnz = "no winner" sentinel
      const int64_t c = t % k;
      const int64_t j = indx[a];
      const float go = grad_out[t];
      if (grad_mat) atomicAdd(grad_mat + j * k + c, (val ? val[a] : 1.0f) * go);
      if (grad_val) atomicAdd(grad_val + a, mat[j * k + c] * go);
   }
}

template <int VEC> __device__ __forceinline__ float dot_chunk(const float *p, const float *q);
template <> __device__ __forceinline__ float dot_chunk<4>(const float *p, const float *q) {
   const float4 a = *reinterpret_cast<const float4 *>(p);
   const float4 b = *reinterpret_cast<const float4 *>(q);
   return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
template <> __device__ __forceinline__ float dot_chunk<1>(const float *p, const float *q) { return p[0] * q[0]; }

struct SddmmArgs {
   int64_t m, k;
   const int64_t *indx, *pntrb, *pntre;
   const float *y;
   int64_t ldy;
   const float *g;
   int64_t ldg;
   int mean;
   float *dval;
   int long_row;
};

// one wave, edges [rb, re) of row `row`; G = 64/LPR edges per step
template <int VEC, int LPR>
__device__ __forceinline__ void sddmm_edges(const SddmmArgs &a, int64_t row, int64_t rb, int64_t re, float scale) {
   constexpr int G = 64 / LPR;
   constexpr int U = 4;
   const int lane = threadIdx.x & 63;
   const int g = lane / LPR, lc = lane % LPR;
   const float *gr = a.g + (size_t)row * (size_t)a.ldg;
   const int src = ((lane % (G * U)) % G) * LPR + transposed_owner<U, LPR>((lane % (G * U)) / G);   // (see sddmm_task_kernel)
   for (int64_t base = rb; base < re; base += 64) {
      const int64_t p = base + lane;
      const int c_l = p < re ? (int)a.indx[p] : 0;
      const int64_t left = re - base;
      const int cnt = left < 64 ? (int)left : 64;
      float res = 0.0f;                                   // lane l collects the result of edge l of the batch: one store per batch
      for (int s = 0; s < cnt; s += G * U) {
         float part[U];
#pragma unroll
         for (int u = 0; u < U; u++) {
            const int ei = s + u * G + g;
            const bool ok = ei < cnt;
            const int cc = __shfl(c_l, ei & 63);
            const float *yr = a.y + (size_t)cc * (size_t)a.ldy;
            float acc = 0.0f;
            if (ok)
               for (int64_t c = (int64_t)lc * VEC; c < a.k; c += LPR * VEC) acc += dot_chunk<VEC>(yr + c, gr + c);
            part[u] = acc;
         }
         int mine;
         const float t = reduce_transposed<U, LPR>(part, lc, mine);
         const float got = __shfl(t, src);
         res = (s / (G * U) == lane / (G * U)) ? got : res;
      }
      if (lane < cnt) a.dval[base + lane] = res * scale;
   }
}

template <int VEC, int LPR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void sddmm_csr_kernel(const SddmmArgs a) {
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: task/row bookkeeping lives in SGPRs
   const int64_t row0 = (int64_t)blockIdx.x * WAVES;
   const int64_t row = row0 + wave;
   if (row < a.m) {
      const int64_t b = a.pntrb[row], e = a.pntre[row];
      const int64_t deg = e - b;
      if (deg > 0 && deg <= a.long_row) {
         const float scale = a.mean ? 1.0f / (float)deg : 1.0f;
         sddmm_edges<VEC, LPR>(a, row, b, e, scale);
      }
   }
   for (int r = 0; r < WAVES; r++) {
      const int64_t lr = row0 + r;
      if (lr >= a.m) break;
      const int64_t b = a.pntrb[lr], e = a.pntre[lr];
      const int64_t deg = e - b;
      if (deg <= a.long_row) continue;
      int64_t chunk = (deg + WAVES - 1) / WAVES;
      chunk = (chunk + 63) & ~(int64_t)63;
      int64_t cb = b + (int64_t)wave * chunk, ce = cb + chunk;
      if (cb > e) cb = e;
      if (ce > e) ce = e;
      const float scale = a.mean ? 1.0f / (float)deg : 1.0f;
      sddmm_edges<VEC, LPR>(a, lr, cb, ce, scale);
   }
}

// ---- SDDMM over the task plan of the SpMM (include/isplib_hip.h, fusedMM_csr_tasks_hip) ----------
// One wave per task: the task's edges all lie in one column slice, tasks are grouped by XCD lane, so
// the gathers of y enjoy the same L2 affinity as the task-list SpMM -- and since dval is per edge,
// no partials and no combine are needed at all.  g[row, :] sits in registers for the whole task.
constexpr unsigned SD_BUF_LIMIT = 0xE0000000u, SD_BUF_OOB = 0xF0000000u;
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int sd_v4i_t;

struct SddmmTaskArgs {
   int64_t k;
   const int64_t *indx, *pntrb, *pntre;
   const int32_t *indx32;
   const float *y;
   int64_t ldy;
   unsigned ybytes;
   const float *g;
   int64_t ldg;
   int mean;
   float *dval;
   const int *task_row;
   const int64_t *task_b;
   const int *task_len;
   int64_t lane_off[9];
};

// ACCUM: a later column panel of the same call -- the dot products of this panel are added to what the earlier panels
// stored (the old values of a 64-edge batch are fetched before its gathers are issued, so their latency hides behind them; a
// task's results are consecutive CSR positions, read and written coalesced)
template <int LPR, int NCH, int WAVES, bool ACCUM>
__global__ __launch_bounds__(WAVES * 64, NCH == 1 ? 8 : 1) void sddmm_task_kernel(const SddmmTaskArgs a) {
   constexpr int G = 64 / LPR, U = 4;
   const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   const unsigned xcd = blockIdx.x & 7u, within = blockIdx.x >> 3;
   const int64_t t = a.lane_off[xcd] + (int64_t)within * WAVES + wave;
   if (t >= a.lane_off[xcd + 1]) return;
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);
   const int row = a.task_row[t];
   const int64_t b = a.task_b[t], e = b + a.task_len[t];
   const int64_t deg = a.pntre[row] - a.pntrb[row];
   const float scale = a.mean ? 1.0f / (float)(deg > 1 ? deg : 1) : 1.0f;
   // this lane's columns of g[row]; ragged k: the last vector is shifted back, its duplicate components zeroed
   unsigned cbyte[NCH];
   float gv[NCH][4];
#pragma unroll
   for (int j = 0; j < NCH; j++) {
      int c = (j * LPR + lc) * 4;
      int vfirst = 0;
      const bool ok = c < (int)a.k;
      if (ok && c + 4 > (int)a.k) { vfirst = c + 4 - (int)a.k; c = (int)a.k - 4; }
      cbyte[j] = ok ? (unsigned)c * 4u : SD_BUF_OOB;
      const float *gr = a.g + (size_t)row * (size_t)a.ldg + c;
#pragma unroll
      for (int v = 0; v < 4; v++) gv[j][v] = (ok && v >= vfirst) ? gr[v] : 0.0f;
   }
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   // Results leave the wave ONCE per 64-edge batch, coalesced (round 4): a step finishes G * U dot products in scattered
   // lanes -- edge j of the step in the lanes of slot j % G that the butterfly leaves value j / G in -- and used to store them
   // at once, 4 bytes from each of G * U lanes: one store instruction per step, and the address pipeline this kernel is bound
   // by (98.6 % busy) pays for every instruction, however few lanes it carries.  Now lane l of the wave collects the result of
   // edge l of the batch with one cross-lane read per step (the LDS pipe is idle here), and one 256-byte store follows the
   // batch: 1.8 M store instructions per launch instead of 14.3 M on the Reddit shape.
   constexpr int EPS = G * U;                                   // edges per step
   const int src = ((lane % EPS) % G) * LPR + transposed_owner<U, LPR>((lane % EPS) / G);
   const int my_step = lane / EPS;
   for (int64_t base = b; base < e; base += 64) {
      const int64_t p = base + lane;
      const unsigned off_l = p < e ? (a.indx32 ? (unsigned)a.indx32[p] : (unsigned)a.indx[p]) * ldyb : SD_BUF_OOB;
      const int64_t left = e - base;
      const int cnt = left < 64 ? (int)left : 64;
      float old = 0.0f, res = 0.0f;
      if (ACCUM && lane < cnt) old = a.dval[p];                 // (its latency hides behind the batch's gathers)
#pragma unroll 1
      for (int s = 0; s < cnt; s += EPS) {
         sd_v4i_t yv[U][NCH];
#pragma unroll
         for (int u = 0; u < U; u++) {
            const unsigned off = (unsigned)__shfl((int)off_l, (s + u * G + g) & 63);
#pragma unroll
            for (int j = 0; j < NCH; j++) {
               const unsigned o = cbyte[j] >= SD_BUF_OOB ? SD_BUF_OOB : off + cbyte[j];
               yv[u][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
            }
         }
         float d[U];
#pragma unroll
         for (int u = 0; u < U; u++) {
            d[u] = 0.0f;
#pragma unroll
            for (int j = 0; j < NCH; j++)
#pragma unroll
               for (int v = 0; v < 4; v++) d[u] = fmaf(__int_as_float(yv[u][j][v]), gv[j][v], d[u]);
         }
         int mine;
         const float sum = reduce_transposed<U, LPR>(d, lc, mine);
         const float got = __shfl(sum, src);
         res = (s / EPS == my_step) ? got : res;
      }
      if (lane < cnt) a.dval[p] = ACCUM ? old + res * scale : res * scale;
   }
}

template <int LPR, int NCH>
static int launch_sddmm_tasks(const SddmmTaskArgs &a, hipStream_t st, bool accum = false) {
   constexpr int WAVES = 4;
   int64_t most = 0;
   for (int x = 0; x < 8; x++) most = (a.lane_off[x + 1] - a.lane_off[x]) > most ? (a.lane_off[x + 1] - a.lane_off[x]) : most;
   const int64_t gx = 8 * ((most + WAVES - 1) / WAVES);
   if (gx > 0x7fffffffLL) return ISPLIB_FAIL;
   if (gx == 0) return ISPLIB_SUCCESS;
   if (accum) hipLaunchKernelGGL((sddmm_task_kernel<LPR, NCH, WAVES, true>), dim3((unsigned)gx), dim3(WAVES * 64), 0, st, a);
   else hipLaunchKernelGGL((sddmm_task_kernel<LPR, NCH, WAVES, false>), dim3((unsigned)gx), dim3(WAVES * 64), 0, st, a);
   return check_launch("sddmm_task_kernel");
}

template <int VEC, int LPR>
static int launch_sddmm(const SddmmArgs &a, hipStream_t st) {
   constexpr int WAVES = 4;
   const int64_t nb = (a.m + WAVES - 1) / WAVES;
   if (nb > 0x7fffffffLL) return ISPLIB_FAIL;
   hipLaunchKernelGGL((sddmm_csr_kernel<VEC, LPR, WAVES>), dim3((unsigned)nb), dim3(WAVES * 64), 0, st, a);
   return check_launch("sddmm_csr_kernel");
}

}  // namespace isplib

using namespace isplib;

extern "C" int isplib_spmm_minmax_bw_hip(int64_t m, int64_t n, int64_t k, int64_t nnz, const int64_t *indx,
                                         const float *val, const float *mat, const int64_t *arg,
                                         const float *grad_out, float *grad_mat, float *grad_val, void *stream) {
   clear_error();
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "isplib_spmm_minmax_bw_hip: negative dimension");
   hipStream_t st = (hipStream_t)stream;
   if (grad_mat && n * k > 0) ISPLIB_HIP_TRY(hipMemsetAsync(grad_mat, 0, (size_t)n * (size_t)k * sizeof(float), st));
   if (grad_val && nnz > 0) ISPLIB_HIP_TRY(hipMemsetAsync(grad_val, 0, (size_t)nnz * sizeof(float), st));
   const int64_t total = m * k;
   if (total == 0 || nnz == 0 || (!grad_mat && !grad_val)) return ISPLIB_SUCCESS;
   if (!indx || !arg || !grad_out) return fail(ISPLIB_FAIL, "isplib_spmm_minmax_bw_hip: null operand");
   if (grad_val && !mat) return fail(ISPLIB_FAIL, "isplib_spmm_minmax_bw_hip: grad_val needs mat");
   int64_t blocks = (total + 255) / 256;
   if (blocks > 256 * 32) blocks = 256 * 32;   // grid-stride beyond 32 blocks per CU
   hipLaunchKernelGGL(minmax_bw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, total, k, nnz, indx, val, mat, arg,
                      grad_out, grad_mat, grad_val);
   return check_launch("minmax_bw_kernel");
}

extern "C" int isplib_sddmm_csr_hip(int64_t m, int64_t k, const int64_t *indx, const int64_t *pntrb,
                                    const int64_t *pntre, const float *y, int64_t ldy, const float *g, int64_t ldg,
                                    int mean, float *dval, void *stream) {
   clear_error();
   if (m < 0 || k < 0) return fail(ISPLIB_FAIL, "isplib_sddmm_csr_hip: negative dimension");
   if (m == 0) return ISPLIB_SUCCESS;
   if (!indx || !pntrb || !pntre || !dval || (k > 0 && (!y || !g)))
      return fail(ISPLIB_FAIL, "isplib_sddmm_csr_hip: null operand");
   if (ldy < k || ldg < k) return fail(ISPLIB_FAIL, "isplib_sddmm_csr_hip: leading dimension smaller than k");
   SddmmArgs a;
   a.m = m; a.k = k; a.indx = indx; a.pntrb = pntrb; a.pntre = pntre;
   a.y = y; a.ldy = ldy; a.g = g; a.ldg = ldg; a.mean = mean ? 1 : 0; a.dval = dval;
   a.long_row = 4096;
   hipStream_t st = (hipStream_t)stream;
   const uintptr_t al = (uintptr_t)y | (uintptr_t)g;
   const bool v4 = (k % 4 == 0) && (ldy % 4 == 0) && (ldg % 4 == 0) && ((al & 15) == 0);
   if (v4) {
      const int64_t w = k / 4;
      if (w <= 8) return launch_sddmm<4, 8>(a, st);
      if (w <= 16) return launch_sddmm<4, 16>(a, st);
      if (w <= 32) return launch_sddmm<4, 32>(a, st);
      return launch_sddmm<4, 64>(a, st);
   }
   if (k <= 16) return launch_sddmm<1, 16>(a, st);
   if (k <= 32) return launch_sddmm<1, 32>(a, st);
   return launch_sddmm<1, 64>(a, st);
}

extern "C" int isplib_sddmm_csr_tasks_hip(int64_t m, int64_t n, int64_t k, const int64_t *indx, const int32_t *indx32,
                                          const int64_t *pntrb, const int64_t *pntre, int64_t n_tasks, const int32_t *task_row,
                                          const int64_t *task_b, const int32_t *task_len,
                                          const int64_t *lane_off_host, const float *y, int64_t ldy, const float *g,
                                          int64_t ldg, int mean, float *dval, void *stream) {
   clear_error();
   if (m < 0 || n < 0 || k < 0 || n_tasks < 0) return fail(ISPLIB_FAIL, "isplib_sddmm_csr_tasks_hip: negative dimension");
   if (m == 0 || n_tasks == 0) return ISPLIB_SUCCESS;
   if (k < 4 || k > 1024) return fail(ISPLIB_FAIL, "isplib_sddmm_csr_tasks_hip: 4 <= k <= 1024 required (use isplib_sddmm_csr_hip)");
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > SD_BUF_LIMIT) return fail(ISPLIB_FAIL, "isplib_sddmm_csr_tasks_hip: dense operand larger than 3.5 GiB");
   if (!indx || !pntrb || !pntre || !task_row || !task_b || !task_len || !lane_off_host || !y || !g || !dval)
      return fail(ISPLIB_FAIL, "isplib_sddmm_csr_tasks_hip: null operand");
   if (ldy < k || ldg < k) return fail(ISPLIB_FAIL, "isplib_sddmm_csr_tasks_hip: leading dimension smaller than k");
   SddmmTaskArgs a;
   a.k = k; a.indx = indx; a.indx32 = indx32; a.pntrb = pntrb; a.pntre = pntre; a.y = y; a.ldy = ldy; a.ybytes = (unsigned)yb;
   a.g = g; a.ldg = ldg; a.mean = mean ? 1 : 0; a.dval = dval;
   a.task_row = task_row; a.task_b = task_b; a.task_len = task_len;
   for (int x = 0; x < 9; x++) a.lane_off[x] = lane_off_host[x];
   if (a.lane_off[0] != 0 || a.lane_off[8] != n_tasks) return fail(ISPLIB_FAIL, "isplib_sddmm_csr_tasks_hip: lane_off must run from 0 to n_tasks");
   hipStream_t st = (hipStream_t)stream;
   // Column panels (isplib_hip_tune(12, cols); 0 = whole rows): a dot product is a sum over columns, so panel c adds its
   // share to what panels 0..c-1 stored.  A panel of the dense operand is cols / k of its size: the slices of a whole-row
   // plan then fit the L2 (Reddit shape, K=128, 16 slices: 7.5 MB whole, 3.7 MB per 64-column panel).  Every panel but the
   // last is `cols` wide (whole cache lines when cols is a multiple of 32); a tail under 4 columns joins the panel before it.
   const int64_t pw = (g_sddmm_panel_cols >= 32 && (g_sddmm_panel_cols % 32) == 0 && k >= 2 * (int64_t)g_sddmm_panel_cols &&
                       (ldy % 32) == 0 && (ldg % 32) == 0) ? g_sddmm_panel_cols : k;
   for (int64_t c0 = 0; c0 < k;) {
      int64_t c1 = c0 + pw;
      if (c1 + 4 > k) c1 = k;
      SddmmTaskArgs p = a;
      p.k = c1 - c0;
      p.y = y + c0;
      p.g = g + c0;
      p.ybytes = (unsigned)(yb - (unsigned long long)c0 * 4ull);
      const int64_t w = (p.k + 3) / 4;
      const bool acc = c0 > 0;
      int rc;
      if (w <= 8) rc = launch_sddmm_tasks<8, 1>(p, st, acc);
      else if (w <= 16) rc = launch_sddmm_tasks<16, 1>(p, st, acc);
      else if (w <= 32) rc = launch_sddmm_tasks<32, 1>(p, st, acc);
      else if (w <= 64) rc = launch_sddmm_tasks<64, 1>(p, st, acc);
      else if (w <= 128) rc = launch_sddmm_tasks<64, 2>(p, st, acc);
      else rc = launch_sddmm_tasks<64, 4>(p, st, acc);
      if (rc) return rc;
      c0 = c1;
   }
   return ISPLIB_SUCCESS;
}
