// gather.h -- the gather-reduce inner loops shared by every SpMM schedule (plain, column-sliced, task list):
// vector load/store helpers, the (value, edge id) comparator of max/min, the 64-bit-address and the
// buffer-descriptor forms of "one wave walks an edge range", the cross-slot butterfly, and the unroll /
// occupancy choices.  Device code only; included by spmm.hip and spmm_tasks.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>

namespace isplib {

enum { OP_ADD = 0, OP_MAX = 1, OP_MIN = 2 };

template <int VEC> __device__ __forceinline__ void load_vec(const float *p, float (&r)[VEC]);
template <> __device__ __forceinline__ void load_vec<4>(const float *p, float (&r)[4]) {
   const float4 t = *reinterpret_cast<const float4 *>(p);
   r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
}
template <> __device__ __forceinline__ void load_vec<2>(const float *p, float (&r)[2]) {
   const float2 t = *reinterpret_cast<const float2 *>(p);
   r[0] = t.x; r[1] = t.y;
}
template <> __device__ __forceinline__ void load_vec<1>(const float *p, float (&r)[1]) { r[0] = *p; }

template <int VEC> __device__ __forceinline__ void store_vec(float *p, const float (&r)[VEC]);
template <> __device__ __forceinline__ void store_vec<4>(float *p, const float (&r)[4]) {
   *reinterpret_cast<float4 *>(p) = make_float4(r[0], r[1], r[2], r[3]);
}
template <> __device__ __forceinline__ void store_vec<2>(float *p, const float (&r)[2]) {
   *reinterpret_cast<float2 *>(p) = make_float2(r[0], r[1]);
}
template <> __device__ __forceinline__ void store_vec<1>(float *p, const float (&r)[1]) { *p = r[0]; }

// 16-byte store that may sit on any 4-byte boundary (ragged K); vfirst leading components belong to
// the neighbouring lane (the last vector of a row is shifted back to end at column k) and are skipped
struct __attribute__((packed, aligned(4))) f4u_t { float x, y, z, w; };
template <int VEC> __device__ __forceinline__ void store_tail(float *p, const float (&r)[VEC], int vfirst) {
   if (VEC == 4) {
      if (vfirst == 0 && ((uintptr_t)p & 15) == 0) {
         store_vec<VEC>(p, r);
      } else if (vfirst == 0) {
         f4u_t t; t.x = r[0]; t.y = r[1]; t.z = r[2]; t.w = r[VEC - 1];
         *reinterpret_cast<f4u_t *>(p) = t;
      } else {
#pragma unroll
         for (int v = 1; v < VEC; v++) if (v >= vfirst) p[v] = r[v];
      }
   } else {
      store_vec<VEC>(p, r);
   }
}

// (value, edge id) comparator: does candidate (t, i) replace (bt, bi)?
template <int OP> __device__ __forceinline__ bool better(float t, int i, float bt, int bi) {
   if (OP == OP_MAX) return (t > bt) || (t == bt && i < bi);
   return (t < bt) || (t == bt && i < bi);
}

template <int OP> __device__ __forceinline__ float identity() {
   return OP == OP_ADD ? 0.0f : (OP == OP_MAX ? -FLT_MAX : FLT_MAX);
}

// One wave walks edges [rb, re) of a row that starts at CSR position row_b.
// acc: running sum (OP_ADD) or running best value; bi: row-relative edge id of
// the best (OP_MAX/MIN), INT_MAX = none yet.
template <int OP, int VEC, int LPR, int NCH, int U, class Args>
__device__ __forceinline__ void wave_edges(const Args &a, int64_t row_b, int64_t rb, int64_t re,
                                           const int (&ccol)[NCH], const bool (&cok)[NCH],
                                           float (&acc)[NCH][VEC], int (&bi)[NCH][VEC]) {
   constexpr int G = 64 / LPR;
   const int lane = threadIdx.x & 63;
   const int g = lane / LPR;
   for (int64_t base = rb; base < re; base += 64) {
      const int64_t p = base + lane;
      int c_l = 0;
      float v_l = 0.0f;
      if (p < re) {
         c_l = (int)a.indx[p];
         v_l = a.val ? a.val[p] : 1.0f;
      }
      const int64_t left = re - base;
      const int cnt = left < 64 ? (int)left : 64;
      const int rel0 = (int)(base - row_b);
      for (int s = 0; s < cnt; s += G * U) {
         float t[U][NCH][VEC];
         float vv[U];
         bool ok[U];
#pragma unroll
         for (int u = 0; u < U; u++) {
            const int ei = s + u * G + g;
            ok[u] = ei < cnt;
            const int cc = __shfl(c_l, ei & 63);
            vv[u] = __shfl(v_l, ei & 63);
            const float *yr = a.y + (size_t)cc * (size_t)a.ldy;
#pragma unroll
            for (int j = 0; j < NCH; j++) {
               if (ok[u] && cok[j]) {
                  load_vec<VEC>(yr + ccol[j], t[u][j]);
               } else {
#pragma unroll
                  for (int v = 0; v < VEC; v++) t[u][j][v] = 0.0f;
               }
            }
         }
         if (OP == OP_ADD) {            // a step's values are summed first, then enter the running sum as one term
#pragma unroll
            for (int j = 0; j < NCH; j++) {
#pragma unroll
               for (int v = 0; v < VEC; v++) {
                  float part = 0.0f;
#pragma unroll
                  for (int u = 0; u < U; u++) part = ok[u] ? fmaf(vv[u], t[u][j][v], part) : part;
                  acc[j][v] += part;
               }
            }
            continue;
         }
#pragma unroll
         for (int u = 0; u < U; u++) {
            const int ei = s + u * G + g;
#pragma unroll
            for (int j = 0; j < NCH; j++) {
#pragma unroll
               for (int v = 0; v < VEC; v++) {
                  if (OP == OP_ADD) {
                     acc[j][v] = ok[u] ? fmaf(vv[u], t[u][j][v], acc[j][v]) : acc[j][v];
                  } else {
                     const float tt = vv[u] * t[u][j][v];
                     const bool win = ok[u] && cok[j] && (OP == OP_MAX ? tt > acc[j][v] : tt < acc[j][v]);
                     acc[j][v] = win ? tt : acc[j][v];
                     bi[j][v] = win ? rel0 + ei : bi[j][v];
                  }
               }
            }
         }
      }
   }
}

// Fast form of wave_edges for 16-B lanes when the whole dense operand is addressable with a
// 32-bit byte offset (n*ldy*4 <= BUF_LIMIT): y is read through a buffer descriptor, so
//   * the row offset is ONE 32-bit multiply per edge, done before the cross-lane hand-off
//     (64 edges per coalesced metadata load), and one add per gather;
//   * out-of-range lanes carry an offset past the descriptor's size: the hardware range
//     check returns 0 for them, so the loop has no branches and no exec-mask flips;
//   * HAS_VAL = false (unit weights) never touches the value stream and adds instead of fma.
constexpr unsigned BUF_LIMIT = 0xE0000000u;   // bytes addressable; offsets >= BUF_OOB read as 0
constexpr unsigned BUF_OOB = 0xF0000000u;     // + any column offset (< 2^24) stays < 2^32: never wraps

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int v4i_t;

// UU gathers per slot issued back to back for the edges [s, s + G*UU) of the current 64-edge batch
template <int OP, bool HAS_VAL, int LPR, int NCH, int UU>
__device__ __forceinline__ void buf_step(const __amdgpu_buffer_rsrc_t rsrc, unsigned off_l, float v_l, int s, int cnt,
                                         int rel0, int g, const unsigned (&cbyte)[NCH], const unsigned (&poison)[NCH],
                                         const bool (&cok)[NCH], float (&acc)[NCH][4], int (&bi)[NCH][4]) {
   constexpr int G = 64 / LPR;
   v4i_t t[UU][NCH];
   float vv[UU];
#pragma unroll
   for (int u = 0; u < UU; u++) {
      const int ei = (s + u * G + g) & 63;
      const unsigned off = (unsigned)__shfl((int)off_l, ei);
      if (HAS_VAL) vv[u] = __shfl(v_l, ei);
#pragma unroll
      for (int j = 0; j < NCH; j++) {
         // masked edge: off = BUF_OOB, + cbyte (< 2^24) cannot wrap; masked column: OR-ed past the limit
         const unsigned o = (off + cbyte[j]) | poison[j];
         t[u][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
      }
   }
   if (OP == OP_ADD) {
      // The UU values of a step are summed among themselves first and enter the running sum as ONE term: the same
      // number of adds (one more for weighted rows), but the long dependent chain on acc is UU times shorter, and so
      // is the worst-case rounding growth on rows of many like-signed terms (measured on a 2,903-edge row over 3
      // distinct columns: 0.1-0.3 of the 1e-5 tolerance; the fp32 sequential oracle is at 4.8).
#pragma unroll
      for (int j = 0; j < NCH; j++) {
#pragma unroll
         for (int v = 0; v < 4; v++) {
            float part = HAS_VAL ? vv[0] * __int_as_float(t[0][j][v]) : __int_as_float(t[0][j][v]);
#pragma unroll
            for (int u = 1; u < UU; u++)
               part = HAS_VAL ? fmaf(vv[u], __int_as_float(t[u][j][v]), part) : part + __int_as_float(t[u][j][v]);
            acc[j][v] += part;
         }
      }
      return;
   }
#pragma unroll
   for (int u = 0; u < UU; u++) {
      const int ei = s + u * G + g;
#pragma unroll
      for (int j = 0; j < NCH; j++) {
#pragma unroll
         for (int v = 0; v < 4; v++) {
            const float x = __int_as_float(t[u][j][v]);
            if (OP == OP_ADD) {
               acc[j][v] = HAS_VAL ? fmaf(vv[u], x, acc[j][v]) : acc[j][v] + x;
            } else {
               const float tt = HAS_VAL ? vv[u] * x : x;
               const bool win = (ei < cnt) && cok[j] && (OP == OP_MAX ? tt > acc[j][v] : tt < acc[j][v]);
               acc[j][v] = win ? tt : acc[j][v];
               bi[j][v] = win ? rel0 + ei : bi[j][v];
            }
         }
      }
   }
}

// metadata of the 64 edges [p0, p0 + 64) n [.., re): byte offset of the dense row (past the descriptor when masked), weight
template <bool HAS_VAL, class Args>
__device__ __forceinline__ void load_edge_batch(const Args &a, int64_t p0, int64_t re, unsigned ldyb, unsigned &off_l, float &v_l) {
   const int64_t p = p0 + (threadIdx.x & 63);
   off_l = BUF_OOB;
   v_l = 0.0f;
   if (p < re) {
      off_l = (a.indx32 ? (unsigned)a.indx32[p] : (unsigned)a.indx[p]) * ldyb;
      if (HAS_VAL) v_l = a.val[p];
   }
}

// PRE: the metadata of the first batch was loaded by the caller (prefetched while the previous task was gathering)
template <int OP, bool HAS_VAL, int LPR, int NCH, int U, class Args, bool PRE = false>
__device__ __forceinline__ void wave_edges_buf(const Args &a, const __amdgpu_buffer_rsrc_t rsrc, int64_t row_b,
                                               int64_t rb, int64_t re, const int (&ccol)[NCH], const bool (&cok)[NCH],
                                               float (&acc)[NCH][4], int (&bi)[NCH][4], unsigned pre_off = 0u,
                                               float pre_val = 0.0f) {
   constexpr int G = 64 / LPR;
   constexpr int UT = U >= 4 ? 2 : 1;   // tail granularity: fewer all-masked gathers on short segments
   const int lane = threadIdx.x & 63;
   const int g = lane / LPR;
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   unsigned cbyte[NCH], poison[NCH];   // lanes whose columns lie beyond k read past the descriptor too
#pragma unroll
   for (int j = 0; j < NCH; j++) {
      cbyte[j] = (unsigned)ccol[j] * 4u;
      poison[j] = cok[j] ? 0u : BUF_OOB;
   }
   for (int64_t base = rb; base < re; base += 64) {
      unsigned off_l;
      float v_l;
      if (PRE && base == rb) { off_l = pre_off; v_l = pre_val; }
      else load_edge_batch<HAS_VAL>(a, base, re, ldyb, off_l, v_l);
      const int64_t left = re - base;
      const int cnt = left < 64 ? (int)left : 64;
      const int rel0 = (int)(base - row_b);
      int s = 0;
      for (; s + G * U <= cnt; s += G * U)
         buf_step<OP, HAS_VAL, LPR, NCH, U>(rsrc, off_l, v_l, s, cnt, rel0, g, cbyte, poison, cok, acc, bi);
      // The remainder (< G*U edges) in steps of two gathers.  Deeper tail steps were tried in round 2 (U/2 gathers per
      // step: fewer dependent round trips for a short segment) and lost: lanes past the end of a task still cost
      // their slot in the address pipeline although the range check returns 0 without memory traffic (sweep
      // schedule, 26-edge tasks: 3.83 -> 5.09 ms; task list unchanged).
      for (; s < cnt; s += G * UT)
         buf_step<OP, HAS_VAL, LPR, NCH, UT>(rsrc, off_l, v_l, s, cnt, rel0, g, cbyte, poison, cok, acc, bi);
   }
}

// butterfly over the G edge slots of a wave; every lane ends with the result
template <int OP, int VEC, int LPR, int NCH>
__device__ __forceinline__ void slot_reduce(float (&acc)[NCH][VEC], int (&bi)[NCH][VEC]) {
#pragma unroll
   for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
      for (int j = 0; j < NCH; j++) {
#pragma unroll
         for (int v = 0; v < VEC; v++) {
            const float ot = __shfl_xor(acc[j][v], off);
            if (OP == OP_ADD) {
               acc[j][v] += ot;
            } else {
               const int oi = __shfl_xor(bi[j][v], off);
               const bool take = better<OP>(ot, oi, acc[j][v], bi[j][v]);
               acc[j][v] = take ? ot : acc[j][v];
               bi[j][v] = take ? oi : bi[j][v];
            }
         }
      }
   }
}

// gathers issued back to back per slot (U) and the occupancy the register allocator is held to; both
// measured on MI355X: the unit-weight sum kernel fits 8 waves/SIMD at U = 8 (62 VGPRs), the weighted
// one needs U = 6 for 7, and max/min carry (value, id) pairs, so U = 4 keeps them at 8.
template <int OP, int NCH, int ADDR, bool TASK = false, int LPR = 64> constexpr int unroll_of() {
   if (NCH > 1) return (8 / NCH) > 2 ? 8 / NCH : 2;
   if (TASK) {          // the task kernel keeps less state per wave: 8 gathers in flight fit 64 VGPRs almost everywhere
      if (LPR == 8 && OP != OP_ADD && ADDR == 2) return 4;
      if (LPR == 8 && OP == OP_ADD && ADDR == 1) return 6;
      if (LPR == 8 && (OP != OP_ADD || ADDR == 2)) return 6;
      if (LPR >= 32 && OP != OP_ADD) return 4;      // measured: 8 is no faster unit-weight and 14 % slower weighted
      if (LPR >= 16 && (OP != OP_ADD || ADDR == 2)) return 6;      // room for the pipelined task loop's prefetch
      return 8;
   }
   if (ADDR != 0 && OP != OP_ADD) return 4;
   if (ADDR == 2) return 6;
   return 8;
}
template <int OP, int LPR, int NCH, int ADDR> constexpr int min_waves_of() {
   if (NCH != 1 || ADDR == 0) return 1;
   return 8;
}

// U per-edge partial dot products per lane, each to be summed over the LPR lanes of its slot.  Instead of
// log2(LPR) shuffles per value, the first log2(U) halving steps also halve the number of live values (a lane
// keeps the values whose index bit matches its own lane bit and hands the others over), so U values cost
// U-1 + log2(LPR) - log2(U) shuffles instead of U * log2(LPR).  Returns the finished sum of edge `mine`
// (valid in every lane of the owning LPR/U-lane group).
// x of lane (lane ^ O) for O = 1, 2, 4, 8 by data-parallel-primitive moves inside the 16-lane DPP row -- no LDS round trip
// and no wait, where a ds_bpermute costs both (SDDMM on the stream plan: 4.50 -> ... ms); wider
// partners cross DPP rows and stay on ds_bpermute.  gfx9 has no row_xmask: xor 1 / 2 are quad permutations, xor 8 is
// the row rotated by 8, xor 4 is the half-row mirror (i -> 7 - i) followed by the quad reversal (j -> 3 - j).
template <int O> __device__ __forceinline__ float lane_xor(float x) {
   if constexpr (O == 1 || O == 2 || O == 4 || O == 8) {
      int v = __float_as_int(x);
      if constexpr (O == 1) v = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);          // quad_perm:[1,0,3,2]
      else if constexpr (O == 2) v = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);     // quad_perm:[2,3,0,1]
      else if constexpr (O == 8) v = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);    // row_ror:8
      else {
         v = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);                            // row_half_mirror
         v = __builtin_amdgcn_update_dpp(0, v, 0x1B, 0xF, 0xF, false);                             // quad_perm:[3,2,1,0]
      }
      return __int_as_float(v);
   } else {
      return __shfl_xor(x, O);
   }
}

template <int U, int N, int O> struct transposed_steps {
   __device__ __forceinline__ static void run(float (&d)[U], int lc, int &mine) {
      if constexpr (N > 1) {
         const bool hi = (lc & O) != 0;
         mine |= hi ? (N >> 1) : 0;
#pragma unroll
         for (int i = 0; i < N / 2; i++) {
            const float send = hi ? d[i] : d[i + N / 2];
            const float keep = hi ? d[i + N / 2] : d[i];
            d[i] = keep + lane_xor<O>(send);
         }
         transposed_steps<U, N / 2, O / 2>::run(d, lc, mine);
      } else if constexpr (O >= 1) {
         d[0] += lane_xor<O>(d[0]);
         transposed_steps<U, 1, O / 2>::run(d, lc, mine);
      }
   }
};

template <int U, int LPR>
__device__ __forceinline__ float reduce_transposed(float (&d)[U], int lc, int &mine) {
   static_assert(U == 1 || U == 2 || U == 4 || U == 8, "U must be a power of two <= 8");
   static_assert(LPR >= U, "slot narrower than the values to transpose");
   mine = 0;
   transposed_steps<U, U, LPR / 2>::run(d, lc, mine);
   return d[0];
}

template <int U, int LPR>
__device__ __forceinline__ float reduce_transposed_bpermute(float (&d)[U], int lc, int &mine) {
   static_assert(U == 1 || U == 2 || U == 4 || U == 8, "U must be a power of two <= 8");
   static_assert(LPR >= U, "slot narrower than the values to transpose");
   mine = 0;
   int o = LPR / 2;
#pragma unroll
   for (int n = U; n > 1; n >>= 1, o >>= 1) {
      const bool hi = (lc & o) != 0;
      mine |= hi ? (n >> 1) : 0;
#pragma unroll
      for (int i = 0; i < n / 2; i++) {
         const float send = hi ? d[i] : d[i + n / 2];
         const float keep = hi ? d[i + n / 2] : d[i];
         d[i] = keep + __shfl_xor(send, o);
      }
   }
#pragma unroll
   for (; o >= 1; o >>= 1) d[0] += __shfl_xor(d[0], o);
   return d[0];
}

// lane (within its LPR-lane slot) that ends up owning value u of reduce_transposed
template <int U, int LPR> __device__ __forceinline__ constexpr int transposed_owner(int u) {
   int lane = 0, o = LPR / 2;
   for (int n = U; n > 1; n >>= 1, o >>= 1) lane |= (u & (n >> 1)) ? o : 0;
   return lane;
}

// fold of per-task partial rows (combine_tasks_kernel, spmm_tasks.hip) for other task kernels; aop = 1 add, 2 max, 3 min
int combine_task_partials(int aop, int64_t m, int64_t k, int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                          const int *seg_off, int slices, int mean, float *part_val, int *part_idx, float *z, int64_t ldz,
                          int64_t *z_arg, hipStream_t st);

// task-kernel variants that run the software-pipelined task loop (spmm_tasks.hip) and take two tasks per wave
template <int OP, int LPR, int NCH, int ADDR> constexpr bool pipelined_tasks() {
   return ADDR != 0 && NCH == 1 && (LPR >= 16 || (OP == OP_ADD && ADDR == 1)) && !(OP != OP_ADD && ADDR == 2);   // weighted max/min would spill
}

// tuning knobs (isplib_hip_tune), defined in spmm.hip
extern int g_force_lpr, g_addr_mode, g_tasks_per_wave, g_panel_cols, g_panel_cols_minmax, g_one_pass_kib;
extern int g_sddmm_panel_cols;       // defined in backward.hip

}  // namespace isplib
