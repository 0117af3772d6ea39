// sweep_common.h -- what the kernels of the row-resident schedules share: the launch arguments, the finish of a row, the
// fold of hub rows' partial rows and the compile-time geometry.  Included by spmm_sweep.hip (the stream schedule: the
// default of sum / mean / max / min), fusedmm_stream.hip (the generic FusedMM words on the same front end) and
// experimental/experimental.hip (the sweep, hybrid and stream-SDDMM forms: measured, slower, kept out of the default library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>

#include "../../include/isplib_hip.h"
#include "common.h"
#include "gather.h"

// geometry of the 64-column stream kernel (compile-time: rows per wave, 64-word batch registers per lane, workgroups per CU)
#ifndef ISPLIB_STREAM_NV4
#define ISPLIB_STREAM_NV4 64
#endif
#ifndef ISPLIB_STREAM_NBW4
#define ISPLIB_STREAM_NBW4 2
#endif
#ifndef ISPLIB_STREAM_WGS4
#define ISPLIB_STREAM_WGS4 2
#endif
// cache-policy bits of the stream kernel's gathers (experiment: sc0 = 1, nt = 2, sc1 = 16; every setting measured no faster)
#ifndef ISPLIB_EXP_GATHER_AUX
#define ISPLIB_EXP_GATHER_AUX 0
#endif
// geometry of the hybrid kernel (hot rows of y in LDS): rows per wave, batch registers, table rows (the last one zero),
// hot-word registers, for 64-column slots (streams = 4) and 32-column slots (streams = 8)
#ifndef ISPLIB_HYB4_NV
#define ISPLIB_HYB4_NV 64
#endif
#ifndef ISPLIB_HYB4_NBW
#define ISPLIB_HYB4_NBW 2
#endif
#ifndef ISPLIB_HYB4_HT
#define ISPLIB_HYB4_HT 128
#endif
#ifndef ISPLIB_HYB4_HWR
#define ISPLIB_HYB4_HWR 4
#endif
#ifndef ISPLIB_HYB8_NV
#define ISPLIB_HYB8_NV 128
#endif
#ifndef ISPLIB_HYB8_NBW
#define ISPLIB_HYB8_NBW 4
#endif
#ifndef ISPLIB_HYB8_HT
#define ISPLIB_HYB8_HT 256
#endif
#ifndef ISPLIB_HYB8_HWR
#define ISPLIB_HYB8_HWR 8
#endif
// the same for the 32-column stream kernel (8-lane slots, k <= 32).  Round 3: 128 rows per wave x 2 workgroups per CU hold
// the Reddit shape's 246 K (virtual) rows in ONE generation of 2,048 waves -- one dispatch per pass instead of two -- with
// 32 gathers in flight per wave: K=32 0.715 ms against 0.813 with 64 rows / 16 in flight / 3 workgroups per CU (128 rows
// with 16 or 24 in flight: 0.760 / 0.729; 96 rows: 0.778)
#ifndef ISPLIB_STREAM_NV8
#define ISPLIB_STREAM_NV8 128
#endif
#ifndef ISPLIB_STREAM_NBW8
#define ISPLIB_STREAM_NBW8 4
#endif
#ifndef ISPLIB_STREAM_WGS8
#define ISPLIB_STREAM_WGS8 2
#endif
// the same for max / min (a second LDS plane holds the winners' positions: half the rows per wave of the sum kernel);
// 64-column slots, and 32-column slots (k <= 32: eight rows per gather, 32 gathers in flight from four batch registers)
#ifndef ISPLIB_STREAM_MM_NV
#define ISPLIB_STREAM_MM_NV 32
#endif
#ifndef ISPLIB_STREAM_MM_NBW
#define ISPLIB_STREAM_MM_NBW 2
#endif
#ifndef ISPLIB_STREAM_MM_WGS
#define ISPLIB_STREAM_MM_WGS 2
#endif
#ifndef ISPLIB_STREAM_MM8_NV
#define ISPLIB_STREAM_MM8_NV 64
#endif
#ifndef ISPLIB_STREAM_MM8_NBW
#define ISPLIB_STREAM_MM8_NBW 4
#endif
#ifndef ISPLIB_STREAM_MM8_WGS
#define ISPLIB_STREAM_MM8_WGS 2
#endif

namespace isplib {

struct SweepArgs {
   int64_t k, nnz;
   const float *val;
   const int64_t *indx, *pntrb, *pntre;
   const int32_t *indx32;
   const float *y;
   int64_t ldy;
   unsigned ybytes;
   float *z;
   int64_t ldz;
   int64_t *z_arg;
   int mean;
   int empty_init;                 // max / min: an empty row holds the launcher's init value (-+FLT_MAX) instead of 0
   const int32_t *wave_row;        // [waves][NVMAX] row of the slot, -1 = unused slot
   const int32_t *wave_part;       // [waves][NVMAX] -1: the slot is a whole row (written to z); else index of its partial row
   const int64_t *wave_task_off;   // [waves + 1]
   const int64_t *task_b;          // [n_tasks] first CSR position
   const int32_t *task_meta;       // [n_tasks] (slot << 24) | edges
   int wave_base, wave_count;      // waves of this launch (one generation): [wave_base, wave_base + wave_count)
   // stream form (spmm_stream_kernel): the plan's own copy of the edges, in the order the waves walk them
   const int32_t *words;           // [steps][G] (local row << 24) | column; padding = (the slot's first row << 24) | n
   const float *vals;              // [steps][G] weights in the same order, or null (unit weights)
   const int64_t *wave_step_off;   // [waves + 1] first step of a wave
   unsigned null_word;
   const int32_t *ids;             // stream form, max / min: [steps][G] CSR position of every word (the plan's perm), -1 = padding
   int abs_ids;                    // part_idx holds absolute CSR positions (stream form) instead of row-relative ones
   // hybrid form (spmm_hybrid_kernel): the hottest rows of y of every column slice are served from an LDS table
   const int32_t *hot_rows;        // [slices][HT] column id of every table row of a slice; n = unused / the all-zero last row
   const int32_t *hot_words;       // [hot steps][G] (local row << 24) | table row, per (wave, slice) chunk
   const int64_t *hot_step_off;    // [waves * slices + 1] first hot step of a (wave, slice) chunk
   int slices;
   // SDDMM over the stream plan (sddmm_stream_kernel): dval[perm[word]] (+)= <y[col], g[row]>
   const float *g;                 // [m][ldg] the other dense operand (grad_out)
   int64_t ldg;
   float *dval;                    // [nnz]
#ifdef ISPLIB_EXP_WAVE_TIMES
   unsigned long long *dbg;        // experiment (scripts/exp_wave_times.py): [wave][4] s_memtime at start / loop entry / loop exit / end
#endif
   float *part_val;                // [n_parts][k]
   int *part_idx;                  // [n_parts][k] row-relative edge ids (max/min)
   const int32_t *hub_row, *hub_off;
   int64_t n_hub;
   const float *ep_row_scale, *ep_self, *ep_bias;
   int64_t ep_ld_self;
   int ep_relu;
};

// finished value of a whole row: mean scale / epilogue (sum, mean), empty-row value and absolute arg (max, min)
template <int OP>
__device__ __forceinline__ void finish_row(const SweepArgs &a, int row, int c, float (&v)[4], int (&bi)[4], int64_t (&arg)[4]) {
   const int64_t rb = a.pntrb[row];
   const int64_t deg = a.pntre[row] - rb;
   if (OP == OP_ADD) {
      if (a.mean) {
         const float d = (float)(deg > 1 ? deg : 1);
#pragma unroll
         for (int i = 0; i < 4; i++) v[i] = v[i] / d;
      }
      if (a.ep_self) {
         const float *sr = a.ep_self + (size_t)row * (size_t)a.ep_ld_self + c;
#pragma unroll
         for (int i = 0; i < 4; i++) v[i] += sr[i];
      }
      if (a.ep_row_scale) {
         const float rs = a.ep_row_scale[row];
#pragma unroll
         for (int i = 0; i < 4; i++) v[i] *= rs;
      }
      if (a.ep_bias) {
#pragma unroll
         for (int i = 0; i < 4; i++) v[i] += a.ep_bias[c + i];
      }
      if (a.ep_relu) {
#pragma unroll
         for (int i = 0; i < 4; i++) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
      }
   } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
         if (deg <= 0) v[i] = a.empty_init ? identity<OP>() : 0.0f;
         arg[i] = bi[i] == INT_MAX ? a.nnz : (a.abs_ids ? (int64_t)bi[i] : rb + (int64_t)bi[i]);
      }
   }
}

// rows cut into several virtual rows: fold their partial rows in chunk order (= ascending CSR position); VEC = 1 serves
// panels whose width is not a multiple of 4 (stream schedule at ragged k)
template <int OP, int VEC>
__global__ __launch_bounds__(256) void sweep_hub_fold_kernel(const SweepArgs a) {
   const int64_t kv = a.k / VEC;
   const int64_t total = a.n_hub * kv;
   const int64_t stride = (int64_t)gridDim.x * blockDim.x;
   for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int64_t h = i / kv;
      const int c = (int)(i - h * kv) * VEC;
      float v[4];
      int bi[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { v[q] = identity<OP>(); bi[q] = INT_MAX; }
      const int p1 = a.hub_off[h + 1];
      for (int p = a.hub_off[h]; p < p1; p++) {
         const size_t po = (size_t)p * (size_t)a.k + c;
         float t[VEC];
         load_vec<VEC>(a.part_val + po, t);
#pragma unroll
         for (int q = 0; q < VEC; q++) {
            if (OP == OP_ADD) {
               v[q] += t[q];
            } else {
               // a values-only launch (z_arg == NULL) never wrote part_idx: the chunk's ordinal stands in for its position --
               // chunks are in CSR order, so among equal values the earliest chunk stays, exactly as with real positions
               const int oi = a.z_arg ? a.part_idx[po + q] : p;
               const bool take = better<OP>(t[q], oi, v[q], bi[q]);
               v[q] = take ? t[q] : v[q];
               bi[q] = take ? oi : bi[q];
            }
         }
      }
      const int row = a.hub_row[h];
      int64_t arg[4];
      if (VEC == 4) {
         finish_row<OP>(a, row, c, v, bi, arg);
         store_vec<4>(a.z + (size_t)row * (size_t)a.ldz + c, v);
      } else {                                             // one column: the row finish reads four, so do it by hand
         const int64_t rb = a.pntrb[row], deg = a.pntre[row] - rb;
         if (OP == OP_ADD) {
            if (a.mean) v[0] = v[0] / (float)(deg > 1 ? deg : 1);
            if (a.ep_self) v[0] += a.ep_self[(size_t)row * (size_t)a.ep_ld_self + c];
            if (a.ep_row_scale) v[0] *= a.ep_row_scale[row];
            if (a.ep_bias) v[0] += a.ep_bias[c];
            if (a.ep_relu) v[0] = v[0] > 0.0f ? v[0] : 0.0f;
         } else {
            if (deg <= 0) v[0] = a.empty_init ? identity<OP>() : 0.0f;
            arg[0] = bi[0] == INT_MAX ? a.nnz : (a.abs_ids ? (int64_t)bi[0] : rb + (int64_t)bi[0]);
         }
         a.z[(size_t)row * (size_t)a.ldz + c] = v[0];
      }
      if (OP != OP_ADD && a.z_arg) {
         int64_t *ar = a.z_arg + (size_t)row * (size_t)a.ldz + c;
#pragma unroll
         for (int q = 0; q < VEC; q++) ar[q] = arg[q];
      }
   }
}

// Geometry of the stream kernels.  Persistent waves only stay on the same column slices while FEW of them share a SIMD:
// a SIMD's memory instructions go to its oldest ready wave first, so with 8 waves per SIMD the waves of a CU finish
// one after the other (L2 hit rate 48 % at 32 slices; 67-75 % with 4; the compulsory misses only with 2).  The bytes
// in flight that keep a CU's address pipeline busy (~256 KB) therefore come from depth, not from occupancy:
// WGS workgroups (of 4 waves, one per SIMD) per CU, each wave with U = 64 * NBW / G gathers of 1 KiB in flight.
template <int LPR, int NVMAX, int WGS> constexpr int stream_wgs_per_cu() {
   return 163840 / (4 * NVMAX * LPR * 4 * 4) < WGS ? 163840 / (4 * NVMAX * LPR * 4 * 4) : WGS;
}

// the one geometry per slot width (lanes per row slot = 64 / streams) that the entry launches: rows per wave, batch
// registers and workgroups per CU (measured on the Reddit shape, K = 128 in 64-column panels)
struct StreamGeom { int nvmax, nbw, wgs; };
static StreamGeom stream_geom(int streams, bool minmax = false) {
   if (minmax) return streams == 8 ? StreamGeom{ISPLIB_STREAM_MM8_NV, ISPLIB_STREAM_MM8_NBW, ISPLIB_STREAM_MM8_WGS}
                                   : StreamGeom{ISPLIB_STREAM_MM_NV, ISPLIB_STREAM_MM_NBW, ISPLIB_STREAM_MM_WGS};
   if (streams == 2) return {32, 1, 2};     // 128-column panels: U = 32 gathers of 1 KiB per wave
   if (streams == 4) return {ISPLIB_STREAM_NV4, ISPLIB_STREAM_NBW4, ISPLIB_STREAM_WGS4};
   return {ISPLIB_STREAM_NV8, ISPLIB_STREAM_NBW8, ISPLIB_STREAM_WGS8};   // 32-column panels
}

static int stream_resident_waves(int streams, int cus, bool minmax = false) {
   const StreamGeom ge = stream_geom(streams, minmax);
   const int lpr = 64 / streams;
   const int lds = minmax ? 2 * 4 * (ge.nvmax + 1) * lpr * 4 * 4 : 4 * ge.nvmax * lpr * 4 * 4;
   int wgs = 163840 / lds;
   if (wgs > ge.wgs) wgs = ge.wgs;
   return cus * wgs * 4;
}

static inline int device_cus() {
   int dev = 0, cus = 0;
   if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      (void)hipGetLastError();
      cus = 256;                                          // MI355X
   }
   return cus;
}

// The domain of the stream entries and of isplib_stream_plan_build_hip, with ldy = k (a contiguous dense operand; callers
// with a padded leading dimension check n * ldy themselves): the dense operand inside one buffer descriptor (3.5 GiB) and
// 32-bit edge positions.  A shape outside it is simply not offered the schedule -- it runs on the task list or the plain
// kernel as before the stream schedule existed -- instead of being offered and then refused with an error.
static bool stream_domain_ok(int64_t n, int64_t k, int64_t nnz) {
   return (unsigned long long)n * (unsigned long long)k * 4ull <= (unsigned long long)BUF_LIMIT && nnz < (1LL << 31);
}

}  // namespace isplib
