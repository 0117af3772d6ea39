// spmm_tasks.hip -- the task-list schedule of the SpMM (fusedMM_csr_tasks_hip): see include/isplib_hip.h and
// DESIGN.md 4.2.  Shares the gather loop (gather.h) with the plain and column-sliced kernels of spmm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>

#include "../../include/isplib_hip.h"
#include "common.h"
#include "gather.h"

namespace isplib {

// ---- task-list schedule -------------------------------------------------------------------
// The per-graph plan (isplib_spmm_tasks_* / isplib_amd/plan.py) cuts every non-empty (row, column
// slice) segment into chunks of at most T edges; a chunk is a TASK = one wave.  Tasks are stored
// slice-major and cut into eight contiguous runs of equal edge mass, one per XCD lane (blockIdx % 8),
// so the L2 affinity of the sliced kernel is kept while
//   * hub rows become many independent tasks (no 4-wave cooperative phase, no LDS, no barrier),
//   * empty segments cost nothing (no wave, no partial plane to write and re-read),
//   * short rows are not sliced at all (their whole row is one task, homed on slice row % slices).
// Task t writes partial[t][0:k]; combine_tasks_kernel folds a row's partials in slice order, then
// chunk order (= ascending CSR position), so results stay bitwise reproducible and max/min ties
// still go to the lowest edge id.
struct TaskArgs {
   int64_t m, k, nnz;
   const float *val;
   const int64_t *indx, *pntrb, *pntre;
   const int32_t *indx32;    // the column ids packed to 32 bits, or null
   const float *y;
   int64_t ldy;
   unsigned ybytes;
   float *z;
   int64_t ldz;
   int64_t *z_arg;
   int mean, slices;
   int empty_init;           // max / min: an empty row holds the launcher's init value (-+FLT_MAX) instead of 0
   const int *task_row;      // [n_tasks]
   const int64_t *task_b;    // [n_tasks] first CSR position
   const int *task_len;      // [n_tasks] edges (<= T)
   const int *seg_off;       // [slices*m + 1], slice-major (slice, row) -> first task of the segment
   int64_t lane_off[9];      // tasks of XCD lane x: [lane_off[x], lane_off[x+1])
   int tpw;                  // tasks per wave (consecutive tasks of one lane)
   float *part_val;          // [n_tasks][k]
   int *part_idx;            // [n_tasks][k] row-relative edge ids (max/min)
   // optional epilogue of the fold (sum / mean only; isplib_epilogue): out = act(row_scale[i] * (acc + self[i,c]) + bias[c])
   const float *ep_row_scale, *ep_self, *ep_bias;
   int64_t ep_ld_self;
   int ep_relu;
};

template <int OP, int LPR, int NCH, int WAVES, int ADDR>
__global__ __launch_bounds__(WAVES * 64, (min_waves_of<OP, LPR, NCH, ADDR>())) void spmm_task_kernel(const TaskArgs a) {
   constexpr int VEC = 4;
   constexpr int U = unroll_of<OP, NCH, ADDR, true, LPR>();
   constexpr int PANEL = LPR * VEC * NCH;
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: task/row bookkeeping lives in SGPRs
   const int g = lane / LPR, lc = lane % LPR;
   const unsigned xcd = blockIdx.x & 7u, within = blockIdx.x >> 3;
   const int64_t t0 = a.lane_off[xcd] + ((int64_t)within * WAVES + wave) * a.tpw;
   const int64_t t_end = (t0 + a.tpw) < a.lane_off[xcd + 1] ? (t0 + a.tpw) : a.lane_off[xcd + 1];
   if (t0 >= t_end) return;                            // no barrier anywhere below
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);

   int ccol[NCH], vfirst[NCH];
   bool cok[NCH];
#pragma unroll
   for (int j = 0; j < NCH; j++) {
      ccol[j] = (int)blockIdx.y * PANEL + (j * LPR + lc) * VEC;
      cok[j] = ccol[j] < a.k;
      vfirst[j] = 0;
      if (cok[j] && ccol[j] + 4 > (int)a.k) { vfirst[j] = ccol[j] + 4 - (int)a.k; ccol[j] = (int)a.k - 4; }
   }
   // PIPE (pipelined_tasks(), gather.h: panels of 64 columns and more except weighted max/min, plus the unit-weight
   // sum at 32 columns and fewer): software pipeline over the wave's tasks.  While task t gathers, the edge metadata of
   // task t+1 is already on its way (vector load) and the task record of t+2 too (scalar loads), so a task no longer
   // starts with two dependent round trips to memory (K=128 sum: 3.39 -> 3.25 ms with two tasks per wave; max
   // 4.13 -> 4.00 ms and K=32 sum 0.96 -> 0.92 ms at 6 gathers in flight).  The others keep the plain loop: the two
   // extra live values spill there (weighted max 4.5 -> 5.5 ms when tried).
   constexpr bool PIPE = pipelined_tasks<OP, LPR, NCH, ADDR>();
   const unsigned ldyb_pre = (unsigned)a.ldy * 4u;
   int64_t b_n = 0, e_n = 0, b_nn = 0, e_nn = 0;
   int row_n = 0, row_nn = 0;
   unsigned off_n = 0u;
   float val_n = 0.0f;
   if (PIPE) {
      b_n = a.task_b[t0]; e_n = b_n + a.task_len[t0]; row_n = a.task_row[t0];
      load_edge_batch<ADDR == 2>(a, b_n, e_n, ldyb_pre, off_n, val_n);
      if (t0 + 1 < t_end) { b_nn = a.task_b[t0 + 1]; e_nn = b_nn + a.task_len[t0 + 1]; row_nn = a.task_row[t0 + 1]; }
   }
   for (int64_t t = t0; t < t_end; t++) {              // tpw consecutive tasks of this lane per wave
      int row;
      int64_t b, e;
      unsigned off_c = 0u;
      float val_c = 0.0f;
      if (PIPE) {
         row = row_n; b = b_n; e = e_n; off_c = off_n; val_c = val_n;
         b_n = b_nn; e_n = e_nn; row_n = row_nn;
         if (t + 1 < t_end) load_edge_batch<ADDR == 2>(a, b_n, e_n, ldyb_pre, off_n, val_n);
         if (t + 2 < t_end) { b_nn = a.task_b[t + 2]; e_nn = b_nn + a.task_len[t + 2]; row_nn = a.task_row[t + 2]; }
      } else {
         row = a.task_row[t];
         b = a.task_b[t];
         e = b + a.task_len[t];
      }
      const int64_t row_b = OP == OP_ADD ? b : a.pntrb[row];
      float acc[NCH][VEC];
      int bi[NCH][VEC];
#pragma unroll
      for (int j = 0; j < NCH; j++)
#pragma unroll
         for (int v = 0; v < VEC; v++) { acc[j][v] = identity<OP>(); bi[j][v] = INT_MAX; }
      wave_edges_buf<OP, ADDR == 2, LPR, NCH, U, TaskArgs, PIPE>(a, rsrc, row_b, b, e, ccol, cok, acc, bi, off_c, val_c);
      slot_reduce<OP, VEC, LPR, NCH>(acc, bi);
      if (g == 0) {
         const size_t off = (size_t)t * (size_t)a.k;
#pragma unroll
         for (int j = 0; j < NCH; j++) {
            if (!cok[j]) continue;
            // partial rows are written once and read once by the fold: non-temporal, so they do not push rows of y out
            // of the L2 (3.44 -> 3.39 ms at K=128)
            float *pp = a.part_val + off + ccol[j];
            if (vfirst[j] == 0 && ((uintptr_t)pp & 15) == 0) {
               typedef float f4nt __attribute__((ext_vector_type(4)));
               const f4nt tv = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
               __builtin_nontemporal_store(tv, reinterpret_cast<f4nt *>(pp));
            } else {
               store_tail<VEC>(pp, acc[j], vfirst[j]);
            }
            if (OP != OP_ADD) {
#pragma unroll
               for (int v = 0; v < VEC; v++)
                  if (v >= vfirst[j]) __builtin_nontemporal_store(bi[j][v], a.part_idx + off + ccol[j] + v);
            }
         }
      }
   }
}

template <int OP, int VEC>
__global__ __launch_bounds__(256) void combine_tasks_kernel(const TaskArgs a) {
   const int64_t kv = a.k / VEC;
   const int64_t total = a.m * kv;
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int64_t row = i / kv;
      const int c = (int)(i - row * kv) * VEC;
      float acc[VEC];
      int bi[VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) { acc[v] = identity<OP>(); bi[v] = INT_MAX; }
      for (int s = 0; s < a.slices; s++) {                 // natural slice order = ascending CSR position
         const int *so = a.seg_off + (size_t)s * (size_t)a.m + row;
         const int t1 = so[1];
         for (int t = so[0]; t < t1; t++) {
            const size_t off = (size_t)t * (size_t)a.k + c;
            float p[VEC];
            load_vec<VEC>(a.part_val + off, p);
#pragma unroll
            for (int v = 0; v < VEC; v++) {
               if (OP == OP_ADD) {
                  acc[v] += p[v];
               } else {
                  const int oi = a.part_idx[off + v];
                  const bool take = better<OP>(p[v], oi, acc[v], bi[v]);
                  acc[v] = take ? p[v] : acc[v];
                  bi[v] = take ? oi : bi[v];
               }
            }
         }
      }
      const int64_t rb = a.pntrb[row];
      const int64_t deg = a.pntre[row] - rb;
      if (OP == OP_ADD) {
         if (a.mean) {
            const float d = (float)(deg > 1 ? deg : 1);
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = acc[v] / d;
         }
         if (a.ep_self) {                                   // e.g. the self loop of GCN's (A + I)
            const float *sr = a.ep_self + (size_t)row * (size_t)a.ep_ld_self + c;
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] += sr[v];
         }
         if (a.ep_row_scale) {                              // e.g. D^-1/2 on the left
            const float rs = a.ep_row_scale[row];
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] *= rs;
         }
         if (a.ep_bias) {
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] += a.ep_bias[c + v];
         }
         if (a.ep_relu) {
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] = acc[v] > 0.0f ? acc[v] : 0.0f;
         }
      } else if (deg <= 0) {
#pragma unroll
         for (int v = 0; v < VEC; v++) acc[v] = a.empty_init ? identity<OP>() : 0.0f;
      }
      store_vec<VEC>(a.z + (size_t)row * (size_t)a.ldz + c, acc);
      if (OP != OP_ADD && a.z_arg) {
         int64_t *ar = a.z_arg + (size_t)row * (size_t)a.ldz + c;
#pragma unroll
         for (int v = 0; v < VEC; v++) ar[v] = bi[v] == INT_MAX ? a.nnz : rb + (int64_t)bi[v];
      }
   }
}

template <int OP, int LPR, int NCH, int ADDR>
static int launch_tasks_cfg(const TaskArgs &a0, hipStream_t st) {
   constexpr int WAVES = 4;
   TaskArgs a = a0;
   if (g_tasks_per_wave <= 0) a.tpw = pipelined_tasks<OP, LPR, NCH, ADDR>() ? 2 : 1;      // the pipeline needs a successor
   int64_t most = 0;
   for (int x = 0; x < 8; x++) most = (a.lane_off[x + 1] - a.lane_off[x]) > most ? (a.lane_off[x + 1] - a.lane_off[x]) : most;
   const int64_t gx = 8 * ((most + (int64_t)WAVES * a.tpw - 1) / ((int64_t)WAVES * a.tpw));
   if (gx > 0x7fffffffLL) return ISPLIB_FAIL;
   constexpr int PANEL = LPR * 4 * NCH;
   const unsigned ny = (unsigned)((a.k + PANEL - 1) / PANEL);
   if (gx > 0) {
      hipLaunchKernelGGL((spmm_task_kernel<OP, LPR, NCH, WAVES, ADDR>), dim3((unsigned)gx, ny, 1), dim3(WAVES * 64), 0, st, a);
      const int rc = check_launch("spmm_task_kernel");
      if (rc) return rc;
   }
   const uintptr_t al = (uintptr_t)a.z;
   int64_t blocks;
   if (a.k % 4 == 0 && a.ldz % 4 == 0 && (al & 15) == 0) {
      blocks = (a.m * (a.k / 4) + 255) / 256; if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((combine_tasks_kernel<OP, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
   } else if (a.k % 2 == 0 && a.ldz % 2 == 0 && (al & 7) == 0) {
      blocks = (a.m * (a.k / 2) + 255) / 256; if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((combine_tasks_kernel<OP, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
   } else {
      blocks = (a.m * a.k + 255) / 256; if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((combine_tasks_kernel<OP, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
   }
   return check_launch("combine_tasks_kernel");
}

template <int OP, int ADDR>
static int launch_tasks_op(const TaskArgs &a, hipStream_t st) {
   const int64_t width = (a.k + 3) / 4;
   if (width <= 8) return launch_tasks_cfg<OP, 8, 1, ADDR>(a, st);
   if (width <= 16) return launch_tasks_cfg<OP, 16, 1, ADDR>(a, st);
   if (width <= 32) return launch_tasks_cfg<OP, 32, 1, ADDR>(a, st);
   if (width <= 64) return launch_tasks_cfg<OP, 64, 1, ADDR>(a, st);
   if (width <= 128) return launch_tasks_cfg<OP, 64, 2, ADDR>(a, st);
   return launch_tasks_cfg<OP, 64, 4, ADDR>(a, st);
}

int combine_task_partials(int aop, int64_t m, int64_t k, int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                          const int *seg_off, int slices, int mean, float *part_val, int *part_idx, float *z, int64_t ldz,
                          int64_t *z_arg, hipStream_t st) {
   TaskArgs a = {};
   a.m = m; a.k = k; a.nnz = nnz; a.pntrb = pntrb; a.pntre = pntre; a.seg_off = seg_off; a.slices = slices; a.mean = mean;
   a.part_val = part_val; a.part_idx = part_idx; a.z = z; a.ldz = ldz; a.z_arg = z_arg;
   a.empty_init = empty_row_init();
   const bool v4 = k % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)z & 15) == 0;
   int64_t blocks = (m * (v4 ? k / 4 : k) + 255) / 256;
   if (blocks > 8192) blocks = 8192;
   if (blocks < 1) return ISPLIB_SUCCESS;
   const dim3 grid((unsigned)blocks), block(256);
   if (aop == 1) { if (v4) hipLaunchKernelGGL((combine_tasks_kernel<OP_ADD, 4>), grid, block, 0, st, a); else hipLaunchKernelGGL((combine_tasks_kernel<OP_ADD, 1>), grid, block, 0, st, a); }
   else if (aop == 2) { if (v4) hipLaunchKernelGGL((combine_tasks_kernel<OP_MAX, 4>), grid, block, 0, st, a); else hipLaunchKernelGGL((combine_tasks_kernel<OP_MAX, 1>), grid, block, 0, st, a); }
   else { if (v4) hipLaunchKernelGGL((combine_tasks_kernel<OP_MIN, 4>), grid, block, 0, st, a); else hipLaunchKernelGGL((combine_tasks_kernel<OP_MIN, 1>), grid, block, 0, st, a); }
   return check_launch("combine_tasks_kernel");
}

}  // namespace isplib

using namespace isplib;

extern "C" size_t isplib_spmm_tasks_workspace_bytes(int32_t imessage, int64_t n_tasks, int64_t k) {
   if (n_tasks <= 0 || k <= 0) return 256;
   const size_t plane = ((size_t)n_tasks * (size_t)k * sizeof(float) + 255) & ~(size_t)255;
   return plane * (((imessage & 0xF0000) != ISPLIB_AOP_ADD) ? 2 : 1);
}

static int tasks_entry(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const float *val,
                       const int64_t *indx, const int32_t *indx32, const int64_t *pntrb, const int64_t *pntre, int64_t n_tasks,
                       const int32_t *task_row, const int64_t *task_b, const int32_t *task_len, const int32_t *seg_off,
                       int slices, const int64_t *lane_off_host, const float *y, int64_t ldy, float *z, int64_t ldz,
                       int64_t *z_arg, void *workspace, size_t workspace_bytes, const isplib_epilogue *ep,
                       void *stream) {
   clear_error();
   const int32_t vop = imessage & 0xF, rop = imessage & 0xF0, sop = imessage & 0xF00, vsc = imessage & 0xF000,
                 aop = imessage & 0xF0000;
   if (vop != ISPLIB_VOP_COPY_RHS || rop != ISPLIB_ROP_NOOP || sop != ISPLIB_SOP_COPY ||
       (vsc != ISPLIB_VSC_MUL && vsc != ISPLIB_VSC_MEAN) ||
       (aop != ISPLIB_AOP_ADD && aop != ISPLIB_AOP_MAX && aop != ISPLIB_AOP_MIN) ||
       (vsc == ISPLIB_VSC_MEAN && aop != ISPLIB_AOP_ADD))
      return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_tasks_hip: message outside the SpMM set");
   if (m < 0 || n < 0 || k < 0 || nnz < 0 || n_tasks < 0) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: negative dimension");
   if (m == 0 || k == 0) return ISPLIB_SUCCESS;
   if (slices < 1 || slices > ISPLIB_MAX_SLICES) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: slices must be in [1, 4096]");
   if (k < 4) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: k >= 4 required (use fusedMM_csr_hip)");
   if (ldy < k || ldz < k) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: leading dimension smaller than k");
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > BUF_LIMIT) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: dense operand larger than 3.5 GiB (use fusedMM_csr_hip)");
   if (!pntrb || !pntre || !z || !seg_off || !lane_off_host || (n_tasks > 0 && (!task_row || !task_b || !task_len || !indx || !y)))
      return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: null operand");
   const size_t need = isplib_spmm_tasks_workspace_bytes(imessage, n_tasks, k);
   if (!workspace || workspace_bytes < need) return fail(ISPLIB_NOT_ENOUGH_MEM, "fusedMM_csr_tasks_hip: workspace too small");
   if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: workspace must be 256-byte aligned");
   TaskArgs a;
   a.m = m; a.k = k; a.nnz = nnz; a.val = val; a.indx = indx; a.indx32 = indx32; a.pntrb = pntrb; a.pntre = pntre;
   a.y = y; a.ldy = ldy; a.ybytes = (unsigned)yb; a.z = z; a.ldz = ldz; a.z_arg = z_arg;
   a.mean = (vsc == ISPLIB_VSC_MEAN) ? 1 : 0; a.slices = slices;
   a.empty_init = empty_row_init();
   a.task_row = task_row; a.task_b = task_b; a.task_len = task_len; a.seg_off = seg_off;
   for (int x = 0; x < 9; x++) a.lane_off[x] = lane_off_host[x];
   if (a.lane_off[0] != 0 || a.lane_off[8] != n_tasks) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: lane_off must run from 0 to n_tasks");
   for (int x = 0; x < 8; x++)
      if (a.lane_off[x + 1] < a.lane_off[x]) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_hip: lane_off must be non-decreasing");
   a.tpw = g_tasks_per_wave > 0 ? g_tasks_per_wave : 1;      // 0: chosen per kernel variant at launch
   a.ep_row_scale = a.ep_self = a.ep_bias = nullptr; a.ep_ld_self = 0; a.ep_relu = 0;
   if (ep) {
      if (aop != ISPLIB_AOP_ADD) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_epilogue_hip: the epilogue is defined for sum / mean only");
      if (ep->self && ep->ld_self < k) return fail(ISPLIB_FAIL, "fusedMM_csr_tasks_epilogue_hip: ld_self smaller than k");
      a.ep_row_scale = ep->row_scale; a.ep_self = ep->self; a.ep_ld_self = ep->ld_self; a.ep_bias = ep->bias;
      a.ep_relu = ep->relu ? 1 : 0;
   }
   a.part_val = (float *)workspace;
   const size_t plane = ((size_t)n_tasks * (size_t)k * sizeof(float) + 255) & ~(size_t)255;
   a.part_idx = aop == ISPLIB_AOP_ADD ? nullptr : (int *)((char *)workspace + plane);
   hipStream_t st = (hipStream_t)stream;
   // Wide feature matrices are swept in column panels of g_panel_cols floats, one complete pass
   // (task kernel + combine) per panel on the same stream: a panel of y is n*panel*4 bytes, which stays
   // inside the 256 MiB Infinity Cache when the whole y does not, and every pass runs at the efficiency of
   // the well-filled K = panel case.  Panels only change which columns a launch touches, never a result.
   // 64-column passes are the most efficient ones (sum K=128: two passes 3.45 ms, one pass 3.61 ms; max: 4.16 vs 4.75 ms)
   // -- as long as panel boundaries fall on cache-line boundaries, i.e. rows are a multiple of 128 bytes.  Otherwise
   // every panel would touch partial lines on both sides (K=100: 4.28 ms in 64 + 36 columns, 3.43 ms in one pass), so
   // unaligned rows keep 128-column panels and only from K = 192.
   int panel = aop == ISPLIB_AOP_ADD ? g_panel_cols : g_panel_cols_minmax;
   if ((ldy % 32) != 0 && panel > 0 && panel < 128) panel = 128;
   // The slice count says which schedule the plan was made for: when a slice of WHOLE rows of y already fits the L2
   // budget (<= 9 MB), panels would only shorten the gathers and read the index stream again -- one pass.
   // (The callers pick such a plan when tasks stay long with twice the slices, e.g. hub-dominated R-MAT graphs:
   // 3.43 ms in one pass over 16 slices, 3.67 ms in two passes over 8.)
   if ((double)n * (double)k * 4.0 / (double)slices <= (double)g_one_pass_kib * 1024.0) panel = 0;
   const int64_t pw = (panel >= 4 && k >= panel + panel / 2) ? (int64_t)(panel / 4 * 4) : k;
   for (int64_t c0 = 0; c0 < k; c0 += pw) {
      TaskArgs p = a;
      p.k = (k - c0) < pw ? (k - c0) : pw;
      if (p.k < 4) {                              // a sliver of 1-3 columns: widen it backwards (overlap is rewritten identically)
         p.k = 4;
         c0 = k - 4;
      }
      p.y = y + c0;
      p.z = z + c0;
      p.ep_self = a.ep_self ? a.ep_self + c0 : nullptr;
      p.ep_bias = a.ep_bias ? a.ep_bias + c0 : nullptr;
      p.z_arg = z_arg ? z_arg + c0 : nullptr;
      p.ybytes = (unsigned)(yb - (unsigned long long)c0 * 4ull);
      int rc;
      if (aop == ISPLIB_AOP_ADD) rc = val ? launch_tasks_op<OP_ADD, 2>(p, st) : launch_tasks_op<OP_ADD, 1>(p, st);
      else if (aop == ISPLIB_AOP_MAX) rc = val ? launch_tasks_op<OP_MAX, 2>(p, st) : launch_tasks_op<OP_MAX, 1>(p, st);
      else rc = val ? launch_tasks_op<OP_MIN, 2>(p, st) : launch_tasks_op<OP_MIN, 1>(p, st);
      if (rc) return rc;
      if (c0 + p.k >= k) break;
   }
   return ISPLIB_SUCCESS;
}

extern "C" int fusedMM_csr_tasks_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const float *val,
                                     const int64_t *indx, const int32_t *indx32, const int64_t *pntrb,
                                     const int64_t *pntre, int64_t n_tasks, const int32_t *task_row, const int64_t *task_b,
                                     const int32_t *task_len, const int32_t *seg_off, int slices,
                                     const int64_t *lane_off_host, const float *y, int64_t ldy, float *z,
                                     int64_t ldz, int64_t *z_arg, void *workspace, size_t workspace_bytes,
                                     void *stream) {
   return tasks_entry(imessage, m, n, k, nnz, val, indx, indx32, pntrb, pntre, n_tasks, task_row, task_b, task_len, seg_off,
                      slices, lane_off_host, y, ldy, z, ldz, z_arg, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int fusedMM_csr_tasks_epilogue_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                                              const float *val, const int64_t *indx, const int32_t *indx32,
                                              const int64_t *pntrb, const int64_t *pntre, int64_t n_tasks, const int32_t *task_row,
                                              const int64_t *task_b, const int32_t *task_len, const int32_t *seg_off,
                                              int slices, const int64_t *lane_off_host, const float *y, int64_t ldy,
                                              float *z, int64_t ldz, void *workspace, size_t workspace_bytes,
                                              const isplib_epilogue *epilogue, void *stream) {
   return tasks_entry(imessage, m, n, k, nnz, val, indx, indx32, pntrb, pntre, n_tasks, task_row, task_b, task_len, seg_off,
                      slices, lane_off_host, y, ldy, z, ldz, nullptr, workspace, workspace_bytes, epilogue, stream);
}
