// fusedmm_general.hip -- the generic five-stage FusedMM pipeline for messages other than the four SpMM words
// (include/isplib_hip.h, fusedMM_csr_udef_hip).  iSpLib itself only ever sends the SpMM words
// (csrc/fusedmm.cpp:168-186); the other flag values of csrc/fusedMM.h:18-74 describe the SDDMM-fused patterns of
// the FusedMM paper (graph embedding: sigmoid / t-distribution models; attention-like score-then-aggregate).
// For every row i and stored entry e = (i, j), a = val[e] (1 when val is NULL):
//    VOP   T[c] = f(x[i,c], y[j,c])                 COPY_LHS | COPY_RHS | ADD | SUBL (x-y) | SUBR (y-x) | MAX | MIN
//    ROP   s    = reduce over c                     NOOP (s = 1) | DOT <x_i,T> | ADD_LHS | ADD_RHS (sum T) | NORML | NORMR (sum T^2)
//    SOP   s'   = scalar stage                      NOOP (s' = s) | COPY (s' = a) | UDEF (built-in menu, isplib_sop_udef)
//    VSC   T'[c]= s' * T[c] | s' + T[c] | T[c]      MUL | ADD | NOOP | MEAN (MUL, then the row is divided by max(deg,1))
//    AOP   z[i,c] (+= | max= | min=) T'[c]          ADD | MAX | MIN (z_arg: CSR position of the winner, nnz if none)
// One wave per row, G = 64/LPR edge slots, x[i,:] in registers, fp32 throughout, no atomics (bitwise reproducible).
// This is the reference library's "generic" tier: correct for every supported word, not tuned per pattern.
#include <cfloat>
#include <climits>

#include "common.h"
#include "gather.h"

namespace isplib {

constexpr unsigned GEN_BUF_LIMIT = 0xE0000000u, GEN_BUF_OOB = 0xF0000000u;   // as gather.h's BUF_LIMIT / BUF_OOB
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int gen_v4i_t;

struct GenArgs {
   int64_t m, k, nnz;
   const float *val;
   const int64_t *indx, *pntrb, *pntre;
   const float *x;
   int64_t ldx;
   const float *y;
   int64_t ldy;
   float *z;
   int64_t ldz;
   int64_t *z_arg;
   int vop, rop, sop, vsc, aop;   // stage codes, already shifted down to 0..15
   int sop_udef;
   float sop_param;
   int empty_init;                // max / min: an empty row holds the launcher's init value (-+FLT_MAX) instead of 0
   unsigned ybytes;
   // task form (fusedMM_csr_udef_tasks_hip): the plan of the SpMM task list; partial rows instead of z
   const int32_t *indx32;
   const int *task_row;
   const int64_t *task_b;
   const int *task_len;
   int64_t lane_off[9];
   float *part_val;
   int *part_idx;
};

enum { G_VOP_COPY_LHS = 1, G_VOP_COPY_RHS = 2, G_VOP_ADD = 3, G_VOP_SUBL = 4, G_VOP_SUBR = 5, G_VOP_MAX = 6, G_VOP_MIN = 7 };
enum { G_ROP_NOOP = 0, G_ROP_DOT = 1, G_ROP_ADD_LHS = 2, G_ROP_ADD_RHS = 3, G_ROP_NORML = 4, G_ROP_NORMR = 5 };
enum { G_SOP_NOOP = 0, G_SOP_COPY = 1, G_SOP_UDEF = 15 };
enum { G_VSC_NOOP = 0, G_VSC_MUL = 1, G_VSC_ADD = 2, G_VSC_MEAN = 3 };
enum { G_AOP_ADD = 1, G_AOP_MAX = 2, G_AOP_MIN = 3 };

__device__ __forceinline__ float sop_apply(int kind, float s, float p) {
   switch (kind) {
      case ISPLIB_SOP_SIGMOID: return 1.0f / (1.0f + __expf(-s));
      case ISPLIB_SOP_ONE_MINUS_SIGMOID: return 1.0f - 1.0f / (1.0f + __expf(-s));
      case ISPLIB_SOP_TDIST: return 1.0f / (1.0f + s);
      case ISPLIB_SOP_SCALE: return p * s;
      case ISPLIB_SOP_EXP: return __expf(s);
      case ISPLIB_SOP_LEAKY_EXP: return __expf(s > 0.0f ? s : p * s);
      default: return s;
   }
}

// PAT folds the stage codes of the two hot SDDMM-fused shapes into compile-time constants (the SOP menu entry
// stays a run-time scalar): 1 = COPY_RHS|DOT|UDEF|MUL|ADD (sigmoid embedding, attention scores),
// 2 = SUBR|NORMR|UDEF|MUL|ADD (t-distribution embedding); 0 = every other word, stage codes read from the args.
template <int PAT> struct StageCodes {
   int vop, rop, sop, vsc, aop;
   __device__ __forceinline__ StageCodes(const GenArgs &a)
       : vop(PAT == 1 ? G_VOP_COPY_RHS : PAT == 2 ? G_VOP_SUBR : a.vop), rop(PAT == 1 ? G_ROP_DOT : PAT == 2 ? G_ROP_NORMR : a.rop),
         sop(PAT ? G_SOP_UDEF : a.sop), vsc(PAT ? G_VSC_MUL : a.vsc), aop(PAT ? G_AOP_ADD : a.aop) {}
};

__device__ __forceinline__ float vop_apply(int vop, float xx, float yv) {
   switch (vop) {
      case G_VOP_COPY_LHS: return xx;
      case G_VOP_ADD: return xx + yv;
      case G_VOP_SUBL: return xx - yv;
      case G_VOP_SUBR: return yv - xx;
      case G_VOP_MAX: return fmaxf(xx, yv);
      case G_VOP_MIN: return fminf(xx, yv);
      default: return yv;                                         // COPY_RHS
   }
}

// TASK = false: one wave per row, writes z.  TASK = true: one wave per task of the SpMM plan (a run of one row's
// edges inside one column slice, tasks grouped by XCD lane), writes one partial row per task for the fold.
template <int LPR, int NCH, int WAVES, int PAT, bool TASK>
__global__ __launch_bounds__(WAVES * 64) void fusedmm_general_kernel(const GenArgs a) {
   constexpr int G = 64 / LPR, U = NCH >= 4 ? 2 : 4;
   const StageCodes<PAT> op(a);
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   int64_t row, task = 0, eb, ee;
   if (TASK) {
      const unsigned xcd = blockIdx.x & 7u, within = blockIdx.x >> 3;
      task = a.lane_off[xcd] + (int64_t)within * WAVES + wave;
      if (task >= a.lane_off[xcd + 1]) return;
      row = a.task_row[task];
      eb = a.task_b[task];
      ee = eb + a.task_len[task];
   } else {
      row = (int64_t)blockIdx.x * WAVES + wave;
      if (row >= a.m) return;
      eb = a.pntrb[row];
      ee = a.pntre[row];
   }
   const int64_t rb = a.pntrb[row], re = TASK ? ee : a.pntre[row];   // rb: origin of the row-relative edge ids
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);

   // this lane's columns: 4 consecutive ones per chunk, masked past k (a 16-byte load may run into the next row
   // or past the buffer -- the latter reads 0 -- and the mask drops whatever it brought)
   unsigned cbyte[NCH];
   bool ok[NCH][4];
   float xv[NCH][4];
#pragma unroll
   for (int j = 0; j < NCH; j++) {
      const int c = (j * LPR + lc) * 4;
      cbyte[j] = c < (int)a.k ? (unsigned)c * 4u : GEN_BUF_OOB;
#pragma unroll
      for (int v = 0; v < 4; v++) {
         ok[j][v] = c + v < (int)a.k;
         xv[j][v] = (ok[j][v] && a.x) ? a.x[(size_t)row * (size_t)a.ldx + c + v] : 0.0f;
      }
   }
   float lhs_sum = 0.0f, lhs_sq = 0.0f;                           // row-constant reductions of the left operand
   if (op.rop == G_ROP_ADD_LHS || op.rop == G_ROP_NORML) {
#pragma unroll
      for (int j = 0; j < NCH; j++)
#pragma unroll
         for (int v = 0; v < 4; v++) { lhs_sum += xv[j][v]; lhs_sq += xv[j][v] * xv[j][v]; }
#pragma unroll
      for (int o = LPR / 2; o >= 1; o >>= 1) { lhs_sum += __shfl_xor(lhs_sum, o); lhs_sq += __shfl_xor(lhs_sq, o); }
   }
   const bool edge_rop = op.rop == G_ROP_DOT || op.rop == G_ROP_ADD_RHS || op.rop == G_ROP_NORMR;

   float acc[NCH][4];
   int bi[NCH][4];
   const float init = op.aop == G_AOP_ADD ? 0.0f : (op.aop == G_AOP_MAX ? -FLT_MAX : FLT_MAX);
#pragma unroll
   for (int j = 0; j < NCH; j++)
#pragma unroll
      for (int v = 0; v < 4; v++) { acc[j][v] = init; bi[j][v] = INT_MAX; }

   const unsigned ldyb = (unsigned)a.ldy * 4u;
   for (int64_t base = eb; base < ee; base += 64) {
      const int64_t p = base + lane;
      const unsigned off_l = p < ee ? (a.indx32 ? (unsigned)a.indx32[p] : (unsigned)a.indx[p]) * ldyb : GEN_BUF_OOB;
      const float a_l = (p < ee && a.val) ? a.val[p] : 1.0f;
      const int64_t left = ee - base;
      const int cnt = left < 64 ? (int)left : 64;
#pragma unroll 1
      for (int s0 = 0; s0 < cnt; s0 += G * U) {
         gen_v4i_t yv[U][NCH];
         float aij[U];
#pragma unroll
         for (int u = 0; u < U; u++) {
            const int src = (s0 + u * G + g) & 63;
            const unsigned off = (unsigned)__shfl((int)off_l, src);        // dead edges carry GEN_BUF_OOB: they read 0
            aij[u] = __shfl(a_l, src);
#pragma unroll
            for (int j = 0; j < NCH; j++) {
               const unsigned o = cbyte[j] >= GEN_BUF_OOB ? GEN_BUF_OOB : off + cbyte[j];
               yv[u][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
            }
         }
         // ROP: one partial per edge and lane, summed over the slot's lanes, then handed back to every lane
         float sc[U];
#pragma unroll
         for (int u = 0; u < U; u++) sc[u] = op.rop == G_ROP_ADD_LHS ? lhs_sum : (op.rop == G_ROP_NORML ? lhs_sq : 1.0f);
         if (edge_rop) {
            float part[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
               part[u] = 0.0f;
#pragma unroll
               for (int j = 0; j < NCH; j++)
#pragma unroll
                  for (int v = 0; v < 4; v++) {
                     const float tv = ok[j][v] ? vop_apply(op.vop, xv[j][v], __int_as_float(yv[u][j][v])) : 0.0f;
                     if (op.rop == G_ROP_DOT) part[u] = fmaf(xv[j][v], tv, part[u]);
                     else if (op.rop == G_ROP_ADD_RHS) part[u] += tv;
                     else part[u] = fmaf(tv, tv, part[u]);
                  }
            }
            int mine;
            const float sum = reduce_transposed<U, LPR>(part, lc, mine);
#pragma unroll
            for (int u = 0; u < U; u++) sc[u] = __shfl(sum, g * LPR + transposed_owner<U, LPR>(u));
         }
#pragma unroll
         for (int u = 0; u < U; u++) {
            const int ei = s0 + u * G + g;
            const bool live = ei < cnt;
            float s = sc[u];
            if (op.sop == G_SOP_COPY) s = aij[u];
            else if (op.sop == G_SOP_UDEF) s = sop_apply(a.sop_udef, s, a.sop_param);
#pragma unroll
            for (int j = 0; j < NCH; j++) {
#pragma unroll
               for (int v = 0; v < 4; v++) {
                  float tv = vop_apply(op.vop, xv[j][v], __int_as_float(yv[u][j][v]));
                  if (op.vsc == G_VSC_MUL || op.vsc == G_VSC_MEAN) tv = s * tv;
                  else if (op.vsc == G_VSC_ADD) tv = s + tv;
                  if (op.aop == G_AOP_ADD) {
                     acc[j][v] += live ? tv : 0.0f;
                  } else {
                     const bool win = live && (op.aop == G_AOP_MAX ? tv > acc[j][v] : tv < acc[j][v]);
                     acc[j][v] = win ? tv : acc[j][v];
                     bi[j][v] = win ? (int)(base - rb) + ei : bi[j][v];
                  }
               }
            }
         }
      }
   }
   // fold the G edge slots (ties: lowest CSR position, like the SpMM kernels)
#pragma unroll
   for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < NCH; j++) {
#pragma unroll
         for (int v = 0; v < 4; v++) {
            const float ot = __shfl_xor(acc[j][v], o);
            const int oi = __shfl_xor(bi[j][v], o);
            if (op.aop == G_AOP_ADD) {
               acc[j][v] += ot;
            } else {
               const bool take = op.aop == G_AOP_MAX ? (ot > acc[j][v] || (ot == acc[j][v] && oi < bi[j][v]))
                                                     : (ot < acc[j][v] || (ot == acc[j][v] && oi < bi[j][v]));
               acc[j][v] = take ? ot : acc[j][v];
               bi[j][v] = take ? oi : bi[j][v];
            }
         }
      }
   }
   if (g != 0) return;
   if (TASK) {                                   // partial row of this task; the fold applies mean / empty-row / arg rules
#pragma unroll
      for (int j = 0; j < NCH; j++)
#pragma unroll
         for (int v = 0; v < 4; v++) {
            if (!ok[j][v]) continue;
            const size_t off = (size_t)task * (size_t)a.k + (size_t)(j * LPR + lc) * 4 + v;
            a.part_val[off] = acc[j][v];
            if (op.aop != G_AOP_ADD) a.part_idx[off] = bi[j][v];
         }
      return;
   }
   const float scale = op.vsc == G_VSC_MEAN ? (float)((re - rb) > 1 ? (re - rb) : 1) : 1.0f;
#pragma unroll
   for (int j = 0; j < NCH; j++) {
#pragma unroll
      for (int v = 0; v < 4; v++) {
         if (!ok[j][v]) continue;
         const size_t off = (size_t)row * (size_t)a.ldz + (size_t)(j * LPR + lc) * 4 + v;
         float out = acc[j][v];
         if (op.vsc == G_VSC_MEAN) out = out / scale;
         if (op.aop != G_AOP_ADD) {
            if (re <= rb) out = !a.empty_init ? 0.0f : (op.aop == G_AOP_MAX ? -FLT_MAX : FLT_MAX);   // empty row: 0 (or the launcher's init) / arg = nnz
            if (a.z_arg) a.z_arg[off] = bi[j][v] == INT_MAX ? a.nnz : rb + bi[j][v];
         }
         a.z[off] = out;
      }
   }
}

template <int LPR, int NCH>
static int launch_general(const GenArgs &a, int pat, hipStream_t st) {
   constexpr int WAVES = 4;
   const dim3 block(WAVES * 64);
   if (a.task_row) {
      int64_t most = 0;
      for (int x = 0; x < 8; x++) most = (a.lane_off[x + 1] - a.lane_off[x]) > most ? (a.lane_off[x + 1] - a.lane_off[x]) : most;
      const int64_t gx = 8 * ((most + WAVES - 1) / WAVES);
      if (gx > 0x7fffffffLL) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_tasks_hip: too many tasks for one launch");
      if (gx == 0) return ISPLIB_SUCCESS;
      const dim3 grid((unsigned)gx);
      if (pat == 1) hipLaunchKernelGGL((fusedmm_general_kernel<LPR, NCH, WAVES, 1, true>), grid, block, 0, st, a);
      else if (pat == 2) hipLaunchKernelGGL((fusedmm_general_kernel<LPR, NCH, WAVES, 2, true>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((fusedmm_general_kernel<LPR, NCH, WAVES, 0, true>), grid, block, 0, st, a);
      return check_launch("fusedmm_general_kernel<tasks>");
   }
   const int64_t nb = (a.m + WAVES - 1) / WAVES;
   if (nb > 0x7fffffffLL) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_hip: too many rows for one launch");
   const dim3 grid((unsigned)nb);
   if (pat == 1) hipLaunchKernelGGL((fusedmm_general_kernel<LPR, NCH, WAVES, 1, false>), grid, block, 0, st, a);
   else if (pat == 2) hipLaunchKernelGGL((fusedmm_general_kernel<LPR, NCH, WAVES, 2, false>), grid, block, 0, st, a);
   else hipLaunchKernelGGL((fusedmm_general_kernel<LPR, NCH, WAVES, 0, false>), grid, block, 0, st, a);
   return check_launch("fusedmm_general_kernel");
}

}  // namespace isplib

using namespace isplib;

struct GenPlan {            // task plan of the SpMM (isplib_spmm_tasks_*), or all null for the row-per-wave form
   const int32_t *indx32 = nullptr;
   int64_t n_tasks = 0;
   const int32_t *task_row = nullptr, *task_len = nullptr, *seg_off = nullptr;
   const int64_t *task_b = nullptr;
   int slices = 0;
   const int64_t *lane_off_host = nullptr;
   void *workspace = nullptr;
   size_t workspace_bytes = 0;
};

static int general_entry(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const float *val,
                         const int64_t *indx, const int64_t *pntrb, const int64_t *pntre, const float *x, int64_t ldx,
                         const float *y, int64_t ldy, float beta, float *z, int64_t ldz, int64_t *z_arg, int sop_udef,
                         float sop_param, const GenPlan &plan, void *stream) {
   clear_error();
   GenArgs a = {};
   a.empty_init = empty_row_init();
   a.vop = imessage & 0xF; a.rop = (imessage >> 4) & 0xF; a.sop = (imessage >> 8) & 0xF;
   a.vsc = (imessage >> 12) & 0xF; a.aop = (imessage >> 16) & 0xF;
   if ((imessage >> 20) != 0) return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: unknown bits above the AOP nibble");
   if (a.vop < G_VOP_COPY_LHS || a.vop > G_VOP_MIN)
      return fail(a.vop == 0xF ? ISPLIB_UNDEFINED_USER_FUNCTION : ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: VOP must be COPY_LHS..MIN");
   if (a.rop > G_ROP_NORMR)
      return fail(a.rop == 0xF ? ISPLIB_UNDEFINED_USER_FUNCTION : ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: ROP must be NOOP..NORMR");
   if (a.sop != G_SOP_NOOP && a.sop != G_SOP_COPY && a.sop != G_SOP_UDEF)
      return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: SOP must be NOOP, COPY or UDEF");
   if (a.sop == G_SOP_UDEF && (sop_udef < ISPLIB_SOP_SIGMOID || sop_udef > ISPLIB_SOP_LEAKY_EXP))
      return fail(ISPLIB_UNDEFINED_USER_FUNCTION, "fusedMM_csr_udef_hip: SOP_UDEF needs one of the built-in functions (isplib_sop_udef)");
   if (a.vsc > G_VSC_MEAN)
      return fail(a.vsc == 0xF ? ISPLIB_UNDEFINED_USER_FUNCTION : ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: VSC must be NOOP, MUL, ADD or MEAN");
   if (a.aop < G_AOP_ADD || a.aop > G_AOP_MIN)
      return fail(a.aop == 0xF ? ISPLIB_UNDEFINED_USER_FUNCTION : ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: AOP must be ADD, MAX or MIN");
   if (a.vsc == G_VSC_MEAN && a.aop != G_AOP_ADD) return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: VSC_MEAN is only defined with AOP_ADD");
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_hip: negative dimension");
   if (n > 0x7fffffffLL) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_hip: n must be < 2^31");
   if (k > 1024) return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: the generic pipeline holds a row in registers, k <= 1024");
   if (beta != 0.0f) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_hip: beta must be 0 (z is write-only)");
   if (m == 0 || k == 0) return ISPLIB_SUCCESS;
   if (plan.task_row && k < 4) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_tasks_hip: k >= 4 required (use fusedMM_csr_udef_hip)");
   const bool needs_x = a.vop != G_VOP_COPY_RHS || a.rop == G_ROP_DOT || a.rop == G_ROP_ADD_LHS || a.rop == G_ROP_NORML;
   if (!pntrb || !pntre || !z || (nnz > 0 && (!indx || !y)) || (needs_x && !x))
      return fail(ISPLIB_FAIL, "fusedMM_csr_udef_hip: null operand");
   if (ldy < k || ldz < k || (needs_x && ldx < k)) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_hip: leading dimension smaller than k");
   a.m = m; a.k = k; a.nnz = nnz; a.val = val; a.indx = indx; a.pntrb = pntrb; a.pntre = pntre;
   a.x = needs_x ? x : nullptr; a.ldx = ldx; a.y = y; a.ldy = ldy; a.z = z; a.ldz = ldz; a.z_arg = z_arg;
   a.sop_udef = sop_udef; a.sop_param = sop_param;
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > GEN_BUF_LIMIT) return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_hip: dense operand larger than 3.5 GiB");
   a.ybytes = (unsigned)yb;
   const int word = imessage & 0xFFFFF;
   const int pat = word == (0x2 | 0x10 | 0xF00 | 0x1000 | 0x10000) ? 1 : word == (0x5 | 0x50 | 0xF00 | 0x1000 | 0x10000) ? 2 : 0;
   hipStream_t st = (hipStream_t)stream;
   if (plan.task_row) {
      if (plan.n_tasks < 0 || plan.slices < 1 || plan.slices > ISPLIB_MAX_SLICES)
         return fail(ISPLIB_FAIL, "fusedMM_csr_udef_tasks_hip: bad plan (n_tasks >= 0, slices in [1, 4096])");
      if (!plan.task_b || !plan.task_len || !plan.seg_off || !plan.lane_off_host)
         return fail(ISPLIB_FAIL, "fusedMM_csr_udef_tasks_hip: null plan operand");
      const size_t plane = ((size_t)plan.n_tasks * (size_t)k * sizeof(float) + 255) & ~(size_t)255;
      const size_t need = plane * (a.aop == G_AOP_ADD ? 1 : 2);
      if (!plan.workspace || plan.workspace_bytes < need) return fail(ISPLIB_NOT_ENOUGH_MEM, "fusedMM_csr_udef_tasks_hip: workspace too small");
      if (((uintptr_t)plan.workspace & 255) != 0) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_tasks_hip: workspace must be 256-byte aligned");
      a.indx32 = plan.indx32; a.task_row = plan.task_row; a.task_b = plan.task_b; a.task_len = plan.task_len;
      for (int xl = 0; xl < 9; xl++) a.lane_off[xl] = plan.lane_off_host[xl];
      if (a.lane_off[0] != 0 || a.lane_off[8] != plan.n_tasks) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_tasks_hip: lane_off must run from 0 to n_tasks");
      a.part_val = (float *)plan.workspace;
      a.part_idx = a.aop == G_AOP_ADD ? nullptr : (int *)((char *)plan.workspace + plane);
   }
   const int64_t w = (k + 3) / 4;
   int rc;
   if (w <= 8) rc = launch_general<8, 1>(a, pat, st);
   else if (w <= 16) rc = launch_general<16, 1>(a, pat, st);
   else if (w <= 32) rc = launch_general<32, 1>(a, pat, st);
   else if (w <= 64) rc = launch_general<64, 1>(a, pat, st);
   else if (w <= 128) rc = launch_general<64, 2>(a, pat, st);
   else rc = launch_general<64, 4>(a, pat, st);
   if (rc || !plan.task_row) return rc;
   return combine_task_partials(a.aop, m, k, nnz, pntrb, pntre, plan.seg_off, plan.slices, a.vsc == G_VSC_MEAN ? 1 : 0, a.part_val,
                                a.part_idx, z, ldz, z_arg, st);
}

extern "C" int fusedMM_csr_udef_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, float alpha, int64_t nnz,
                                    int64_t rows, int64_t cols, const float *val, const int64_t *indx,
                                    const int64_t *pntrb, const int64_t *pntre, const float *x, int64_t ldx,
                                    const float *y, int64_t ldy, float beta, float *z, int64_t ldz, int64_t *z_arg,
                                    int sop_udef, float sop_param, void *stream) {
   (void)alpha; (void)rows; (void)cols;
   return general_entry(imessage, m, n, k, nnz, val, indx, pntrb, pntre, x, ldx, y, ldy, beta, z, ldz, z_arg, sop_udef, sop_param,
                        GenPlan(), stream);
}

extern "C" int fusedMM_csr_udef_tasks_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const float *val,
                                          const int64_t *indx, const int32_t *indx32, const int64_t *pntrb,
                                          const int64_t *pntre, const float *x, int64_t ldx, int64_t n_tasks,
                                          const int32_t *task_row, const int64_t *task_b, const int32_t *task_len,
                                          const int32_t *seg_off, int slices, const int64_t *lane_off_host, const float *y,
                                          int64_t ldy, float *z, int64_t ldz, int64_t *z_arg, int sop_udef, float sop_param,
                                          void *workspace, size_t workspace_bytes, void *stream) {
   if (!task_row) {
      clear_error();
      return fail(ISPLIB_FAIL, "fusedMM_csr_udef_tasks_hip: task_row is required");
   }
   GenPlan plan;
   plan.indx32 = indx32; plan.n_tasks = n_tasks; plan.task_row = task_row; plan.task_b = task_b; plan.task_len = task_len;
   plan.seg_off = seg_off; plan.slices = slices; plan.lane_off_host = lane_off_host; plan.workspace = workspace;
   plan.workspace_bytes = workspace_bytes;
   return general_entry(imessage, m, n, k, nnz, val, indx, pntrb, pntre, x, ldx, y, ldy, 0.0f, z, ldz, z_arg, sop_udef, sop_param,
                        plan, stream);
}
