// experimental.hip -- three forms of the row-resident schedules that were built, are bit-exact and tested, and were MEASURED
// SLOWER than what the default library runs (DESIGN_HISTORY.md): kept buildable and tested in their own
// library (libisplib_hip_exp.so, include/isplib_hip_experimental.h) so that the default ABI carries no losers.
//   fusedMM_csr_sweep_hip    the round-2 "sweep" schedule: rows resident in LDS, one latency chain per (row, slice) segment
//   fusedMM_csr_hybrid_hip   the stream schedule with the hottest rows of y of every slice served from an LDS table (2.76-2.90 ms
//                            against 2.68: the workgroup barrier per slice)
//   isplib_sddmm_stream_hip  the SDDMM on the stream plan (4.04 / 3.43 ms against 3.45-3.52 on the task list)
// plus the knobs that only these (and the task-list SDDMM's column panels) have: isplib_hip_tune_experimental.
#include "../sweep_common.h"
#include "../../../include/isplib_hip_experimental.h"

namespace isplib {

// LDS of one workgroup (4 waves x NVMAX rows x one panel row, values and for max / min the ids) and the workgroups
// a CU holds at once: the launch bound (one wave of each workgroup per SIMD) and the plan's wave count follow from it
template <int OP, int LPR, int NVMAX> constexpr int sweep_lds_bytes() { return (OP == OP_ADD ? 1 : 2) * 4 * NVMAX * LPR * 4 * 4; }
template <int OP, int LPR, int NVMAX> constexpr int sweep_wgs_per_cu() {
   return 163840 / sweep_lds_bytes<OP, LPR, NVMAX>() < 8 ? 163840 / sweep_lds_bytes<OP, LPR, NVMAX>() : 8;
}

template <int OP, int LPR, int ADDR, int NVMAX>
__global__ __launch_bounds__(256, (sweep_wgs_per_cu<OP, LPR, NVMAX>())) void spmm_sweep_kernel(const SweepArgs a) {
   constexpr int VEC = 4, WAVES = 4, G = 64 / LPR, PANEL = LPR * VEC;
   constexpr int U = unroll_of<OP, 1, ADDR, true, LPR>();
   constexpr int PLANES = OP == OP_ADD ? 1 : 2;
   // one array (two objects can cost a drained pipeline: guide 5, item 4a): values, then ids
   __shared__ __attribute__((aligned(16))) float s_all[PLANES * WAVES * NVMAX * PANEL];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   const int wl = (int)blockIdx.x * WAVES + wave;
   if (wl >= a.wave_count) return;                       // no barrier anywhere below
   const int64_t w = (int64_t)a.wave_base + wl;
   float *my_val = s_all + wave * (NVMAX * PANEL);
   int *my_idx = reinterpret_cast<int *>(s_all + WAVES * NVMAX * PANEL) + wave * (NVMAX * PANEL);
#pragma unroll
   for (int i = 0; i < NVMAX * PANEL / 256; i++) {
      *reinterpret_cast<float4 *>(my_val + (i * 64 + lane) * 4) = make_float4(identity<OP>(), identity<OP>(), identity<OP>(), identity<OP>());
      if (OP != OP_ADD) *reinterpret_cast<int4 *>(my_idx + (i * 64 + lane) * 4) = make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX);
   }
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);
   int ccol[1];
   bool cok[1];
   ccol[0] = lc * VEC;
   cok[0] = ccol[0] < a.k;
   // max/min: first CSR position of the row of slot `lane` (edge ids are kept relative to the row start)
   int64_t slot_rb = 0;
   if (OP != OP_ADD && lane < NVMAX) {
      const int r = a.wave_row[(size_t)w * NVMAX + lane];
      slot_rb = r >= 0 ? a.pntrb[r] : 0;
   }
   const int64_t t0 = a.wave_task_off[w], t_end = a.wave_task_off[w + 1];
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   int64_t b_n = 0, b_nn = 0;
   int m_n = 0, m_nn = 0;
   unsigned off_n = 0u;
   float val_n = 0.0f;
   if (t0 < t_end) {
      b_n = a.task_b[t0]; m_n = a.task_meta[t0];
      load_edge_batch<ADDR == 2>(a, b_n, b_n + (m_n & 0xFFFFFF), ldyb, off_n, val_n);
      if (t0 + 1 < t_end) { b_nn = a.task_b[t0 + 1]; m_nn = a.task_meta[t0 + 1]; }
   }
   for (int64_t t = t0; t < t_end; t++) {
      const int64_t b = b_n, e = b_n + (m_n & 0xFFFFFF);
      const int slot = __builtin_amdgcn_readfirstlane(m_n >> 24);
      const unsigned off_c = off_n;
      const float val_c = val_n;
      b_n = b_nn; m_n = m_nn;
      if (t + 1 < t_end) load_edge_batch<ADDR == 2>(a, b_n, b_n + (m_n & 0xFFFFFF), ldyb, off_n, val_n);
      if (t + 2 < t_end) { b_nn = a.task_b[t + 2]; m_nn = a.task_meta[t + 2]; }
      int64_t row_b = b;
      if (OP != OP_ADD) {
         const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(slot_rb & 0xffffffffLL), slot);
         const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((uint64_t)slot_rb >> 32), slot);
         row_b = (int64_t)(((uint64_t)hi << 32) | lo);
      }
      float acc[1][VEC];
      int bi[1][VEC];
#pragma unroll
      for (int v = 0; v < VEC; v++) { acc[0][v] = identity<OP>(); bi[0][v] = INT_MAX; }
      wave_edges_buf<OP, ADDR == 2, LPR, 1, U, SweepArgs, true>(a, rsrc, row_b, b, e, ccol, cok, acc, bi, off_c, val_c);
      slot_reduce<OP, VEC, LPR, 1>(acc, bi);
      if (g == 0) {                                      // fold into the slot's LDS row: this wave is its only writer
         float4 *pv = reinterpret_cast<float4 *>(my_val + slot * PANEL + lc * VEC);
         float4 cur = *pv;
         if (OP == OP_ADD) {
            cur.x += acc[0][0]; cur.y += acc[0][1]; cur.z += acc[0][2]; cur.w += acc[0][3];
            *pv = cur;
         } else {
            int4 *pi = reinterpret_cast<int4 *>(my_idx + slot * PANEL + lc * VEC);
            int4 ci = *pi;
            bool tk;
            tk = better<OP>(acc[0][0], bi[0][0], cur.x, ci.x); cur.x = tk ? acc[0][0] : cur.x; ci.x = tk ? bi[0][0] : ci.x;
            tk = better<OP>(acc[0][1], bi[0][1], cur.y, ci.y); cur.y = tk ? acc[0][1] : cur.y; ci.y = tk ? bi[0][1] : ci.y;
            tk = better<OP>(acc[0][2], bi[0][2], cur.z, ci.z); cur.z = tk ? acc[0][2] : cur.z; ci.z = tk ? bi[0][2] : ci.z;
            tk = better<OP>(acc[0][3], bi[0][3], cur.w, ci.w); cur.w = tk ? acc[0][3] : cur.w; ci.w = tk ? bi[0][3] : ci.w;
            *pv = cur;
            *pi = ci;
         }
      }
   }
   // write-out: G slots per step, the LPR lanes of a slot hold one row of the panel
#pragma unroll 1
   for (int j0 = 0; j0 < NVMAX; j0 += G) {
      const int j = j0 + g;
      if (j >= NVMAX) continue;
      const int row = a.wave_row[(size_t)w * NVMAX + j];
      if (row < 0 || !cok[0]) continue;
      const int part = a.wave_part[(size_t)w * NVMAX + j];
      const float4 t4 = *reinterpret_cast<const float4 *>(my_val + j * PANEL + lc * VEC);
      float v[4] = {t4.x, t4.y, t4.z, t4.w};
      int bi[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
      if (OP != OP_ADD) {
         const int4 i4 = *reinterpret_cast<const int4 *>(my_idx + j * PANEL + lc * VEC);
         bi[0] = i4.x; bi[1] = i4.y; bi[2] = i4.z; bi[3] = i4.w;
      }
      const int c = ccol[0];
      if (part >= 0) {                                   // a chunk of a hub row: folded by sweep_hub_fold_kernel
         const size_t po = (size_t)part * (size_t)a.k + c;
         store_vec<4>(a.part_val + po, v);
         if (OP != OP_ADD) *reinterpret_cast<int4 *>(a.part_idx + po) = make_int4(bi[0], bi[1], bi[2], bi[3]);
         continue;
      }
      int64_t arg[4];
      finish_row<OP>(a, row, c, v, bi, arg);
      store_vec<4>(a.z + (size_t)row * (size_t)a.ldz + c, v);
      if (OP != OP_ADD && a.z_arg) {
         int64_t *ar = a.z_arg + (size_t)row * (size_t)a.ldz + c;
#pragma unroll
         for (int i = 0; i < 4; i++) ar[i] = arg[i];
      }
   }
}

// ---- SDDMM on the stream front end -------------------------------------------------------------------------------------
// dA[e] = <y[col[e], :], g[row[e], :]> (csrc/fusedmm.cpp:270,351: the call the reference leaves commented out) walks the
// same edges and gathers the same rows of y as the SpMM, so it runs on the same plan and the same front end: word streams,
// one full 1-KiB gather per step, U gathers in flight per wave, two waves per SIMD.  What changes is what sits in LDS and
// what comes out: the wave's rows of g (one panel, pre-scaled by 1 / max(deg, 1) for mean) take the place of the row
// accumulators -- a lane keeps the four floats of the row its slot is on in registers and re-reads them from LDS when the
// stream turns to another row -- and every step yields one dot product per slot: four steps are summed over the slot's
// lanes by the transposed butterfly (gather.h), handed to the lane that holds the word's CSR position (the plan's `perm`,
// loaded 64 per register like the words) and written once per batch.  A panel is 256 / streams columns; later panels add
// to what the earlier ones stored (ACCUM: the old values are fetched at the start of the batch, behind its gathers).
// Every edge is owned by one (wave, slot, step): plain stores, no atomics, bitwise reproducible.
template <int LPR, int NVMAX, int NBW, int WGS, bool ACCUM>
__global__ __launch_bounds__(256, (stream_wgs_per_cu<LPR, NVMAX, WGS>())) void sddmm_stream_kernel(const SweepArgs a) {
   constexpr int WAVES = 4, G = 64 / LPR, PANEL = LPR * 4, U = 64 * NBW / G;
   constexpr int PER = NVMAX / G, WAVE_FLOATS = NVMAX * PANEL;
   constexpr int Q = LPR / 4;                            // lanes that end up with the sum of one of four steps
   static_assert(U % 4 == 0 && (4 * G) <= 64 && 64 % (4 * G) == 0, "four steps' words lie in one batch register");
   __shared__ __attribute__((aligned(16))) float s_all[WAVES * WAVE_FLOATS];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   const int wl = (int)blockIdx.x * WAVES + wave;
   if (wl >= a.wave_count) return;                       // no barrier anywhere below
   const int64_t w = (int64_t)a.wave_base + wl;
   float *my = s_all + wave * WAVE_FLOATS;
   const bool cok = lc * 4 < a.k;
   int ccol = lc * 4, vfirst = 0;
   if (cok && ccol + 4 > (int)a.k) { vfirst = ccol + 4 - (int)a.k; ccol = (int)a.k - 4; }
   const unsigned cbyte = (unsigned)ccol * 4u, poison = cok ? 0u : BUF_OOB;
   float *lane_base = my + lc * 4;
   // the wave's rows of g into LDS: slot q's LPR lanes hold one row of the panel; the components a shifted last vector
   // shares with its neighbour count once (zeroed here), unused local rows read as zero
#pragma unroll 1
   for (int jj = 0; jj < PER; jj++) {
      const int lrow = g * PER + jj;
      const int row = a.wave_row[(size_t)w * NVMAX + lrow];
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (row >= 0 && cok) {
         const float *gr = a.g + (size_t)row * (size_t)a.ldg + ccol;
         float sc = 1.0f;
         if (a.mean) {
            const int64_t deg = a.pntre[row] - a.pntrb[row];
            sc = 1.0f / (float)(deg > 1 ? deg : 1);
         }
         const int skip = vfirst > a.ep_relu - ccol ? vfirst : a.ep_relu - ccol;     // (ep_relu: columns an earlier panel covered)
         v.x = skip > 0 ? 0.0f : gr[0] * sc;
         v.y = skip > 1 ? 0.0f : gr[1] * sc;
         v.z = skip > 2 ? 0.0f : gr[2] * sc;
         v.w = skip > 3 ? 0.0f : gr[3] * sc;
      }
      *reinterpret_cast<float4 *>(lane_base + lrow * PANEL) = v;
   }
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);
   const int64_t s0 = a.wave_step_off[w], s1 = a.wave_step_off[w + 1];
   const int64_t nwords = (s1 - s0) * G;
   const int32_t *wp = a.words + s0 * G;
   const int32_t *pp = a.ids + s0 * G;
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   const unsigned pad_word = ((unsigned)((lane % G) * PER) << 24) | a.null_word;
   auto load_words = [&](int64_t first, unsigned (&word)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         word[q] = i < nwords ? (unsigned)wp[i] : pad_word;
      }
   };
   auto load_perm = [&](int64_t first, int (&pos)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         pos[q] = i < nwords ? pp[i] : -1;
      }
   };
   unsigned w1[NBW], w2[NBW];
   int pc[NBW], pn[NBW];                                  // CSR positions of the batch being consumed and of the next
   float old[NBW], res[NBW];
   v4i_t t[U];
   unsigned la[U];
   auto issue = [&](int u, const unsigned (&word_l)[NBW]) {
      const unsigned word = (unsigned)__shfl((int)word_l[(u * G) / 64], (u * G) % 64 + g);
      const unsigned o = (__umul24(word & 0xFFFFFFu, ldyb) + cbyte) | poison;
      la[u] = (word >> 24) * (unsigned)PANEL;
      t[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
   };
   load_words(0, w1);
   load_perm(0, pc);
#pragma unroll
   for (int u = 0; u < U; u++) issue(u, w1);
   load_words(64 * NBW, w1);
   load_perm(64 * NBW, pn);
   load_words(128 * NBW, w2);
   // the lane of the slot's group that owns step (word lane / G) % 4 of this lane's word, after the butterfly
   const int src_lane = (lane % G) * LPR + ((lane / G) % 4) * Q;
   const int64_t nb = (nwords + 64 * NBW - 1) / (64 * NBW);
   for (int64_t b = 0; b < nb; b++) {
      if (ACCUM) {
#if defined(ISPLIB_EXP_SDDMM_STREAM_ORDER)
#pragma unroll
         for (int q = 0; q < NBW; q++) old[q] = a.dval[(s0 * G + b * 64 * NBW + q * 64 + lane) % a.nnz];
#elif defined(ISPLIB_EXP_SDDMM_NOSTORE)
         old[0] = 0.0f;
#else
#pragma unroll
         for (int q = 0; q < NBW; q++) old[q] = pc[q] >= 0 ? a.dval[pc[q]] : 0.0f;
#endif
      }
      float d[4];
      float4 gq[4];
#pragma unroll
      for (int u = 0; u < U; u++) {
         // (round 5) g's row of every step straight from LDS, the four reads of a group issued together -- one ds_read_b128 per step
         // on a pipe that is otherwise idle -- instead of kept in registers behind a compare, a branch and four copies per step
         // (the trick that took the FusedMM stream kernel from 39 to 28 vector instructions per step)
         if ((u & 3) == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) gq[q] = *reinterpret_cast<const float4 *>(lane_base + la[u + q]);
         }
         const float4 gv = gq[u & 3];
         d[u & 3] = fmaf(__int_as_float(t[u][0]), gv.x, fmaf(__int_as_float(t[u][1]), gv.y,
                    fmaf(__int_as_float(t[u][2]), gv.z, __int_as_float(t[u][3]) * gv.w)));
         issue(u, w1);
         if ((u & 3) == 3) {
            int mine;
            const float sum = reduce_transposed<4, LPR>(d, lc, mine);
            const float mv = __shfl(sum, src_lane);
            constexpr int WPG = 4 * G;                   // words of four steps
            const int first = ((u - 3) * G) % 64;
            if (lane >= first && lane < first + WPG) res[((u - 3) * G) / 64] = mv;
         }
      }
#if defined(ISPLIB_EXP_SDDMM_STREAM_ORDER)                  // timing experiment: results stored in stream order (coalesced), no perm
#pragma unroll
      for (int q = 0; q < NBW; q++) a.dval[(s0 * G + b * 64 * NBW + q * 64 + lane) % a.nnz] = res[q] + (ACCUM ? old[q] : 0.0f);
#elif defined(ISPLIB_EXP_SDDMM_NOSTORE)                    // timing experiment: one store per wave instead of one per word
      if (b + 1 == nb) a.dval[w] = res[0] + res[NBW - 1] + (ACCUM ? old[0] : 0.0f);
#else
#pragma unroll
      for (int q = 0; q < NBW; q++)
         if (pc[q] >= 0) a.dval[pc[q]] = ACCUM ? old[q] + res[q] : res[q];
#endif
#pragma unroll
      for (int q = 0; q < NBW; q++) { w1[q] = w2[q]; pc[q] = pn[q]; }
      load_words((b + 3) * 64 * NBW, w2);
      load_perm((b + 2) * 64 * NBW, pn);
   }
}

// ---- hybrid form: the stream kernel with the hottest rows of y served from LDS ------------------------------------------
// What bounds the stream kernel is the CU's address pipeline: every gathered row costs it the same ~6.5 cycles per
// 256 bytes whether the L2 hits or not.  The only bytes that do not pay that are bytes that never enter
// it.  Here the HT - 1 most-referenced rows of y of every column slice (the plan picks them: in-degree, per slice, the
// same table for every workgroup) are staged ONCE per slice and workgroup into a table beside the row accumulators --
// by LDS-DMA (buffer_load ... lds), no registers -- and the edges that point at them (the "hot" words: local row, table
// row) are served by ds_read_b128 at LDS rate while the "cold" words keep the gather pipeline busy exactly as in
// spmm_stream_kernel.  One workgroup of 8 waves per CU (two per SIMD, as there) shares the table; the accumulators take
// 128 KB, the table the remaining 32 KB of the CU's 160 KB.
//   * The cold stream is free-running as before (no slice boundaries, U gathers in flight at all times).  It is cut
//     into `slices` PHASES of equal batch counts; phase p of every wave is, to within a few percent, its slice p.
//   * Phase p: hot chunk p (the wave's hot edges of slice p, from the table of slice p) -> barrier (everyone is done
//     with table p) -> the DMA loads of table p+1 and the loads of hot words p+1 are issued -> the phase's cold
//     batches -> counted vmcnt wait (the DMA loads are older than the last batch's gathers: they have landed) ->
//     barrier (table p+1 complete).  Nothing drains the gather pipeline: while a wave works through its hot chunk its
//     U gathers keep landing, and the staging loads travel behind them.
//   * Order: a row's contributions are added in program order of the one wave that owns it -- hot chunk 0, cold
//     phase 0, hot chunk 1, ... -- so every sum is formed in one fixed order: bitwise reproducible, no atomics.
// Sum / mean with unit weights (the weighted launch stays on spmm_stream_kernel).
// One LDS-DMA gather: every lane fetches 16 bytes at byte offset `voff` of the buffer and the wave's 1 KiB lands at LDS
// address `lds_addr` + 16 * lane -- no registers.  Inline asm on purpose: the builtin (__builtin_amdgcn_raw_ptr_buffer_
// load_lds) makes the compiler's wait-count pass guard EVERY later LDS access of the wave (the accumulator flushes)
// with a wait for the DMA, i.e. for all gathers issued before it -- a drained pipeline per slice.  Hidden from the pass,
// the DMA only makes its counted waits for younger loads a little conservative (it under-counts what is outstanding,
// never over-counts); visibility of the table is ordered by hand (counted vmcnt + workgroup barrier).  M0 is restored.
typedef int v4i_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_dma_b128(v4i_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
   unsigned keep;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(voff), "s"(rsrc), "s"(lds_addr)
                : "memory");
}

// s_waitcnt immediate of gfx9: vmcnt = {simm16[15:14], simm16[3:0]}, expcnt = simm16[6:4] (7: no wait), lgkmcnt = simm16[11:8]
constexpr int waitcnt_vm_lgkm0(int vm) { return (vm & 15) | (7 << 4) | (0 << 8) | ((vm >> 4) << 14); }

template <int LPR, int NVMAX, int NBW, int HT, int HWR>
__global__ __launch_bounds__(512, 2) void spmm_hybrid_kernel(const SweepArgs a) {
   constexpr int WAVES = 8, G = 64 / LPR, PANEL = LPR * 4, U = 64 * NBW / G;
   constexpr int PER = NVMAX / G, WAVE_FLOATS = NVMAX * PANEL;
   constexpr int HS = 64 / G;                             // hot steps held by one word register
   constexpr int SHARE = HT / WAVES, SI = SHARE / G;      // table rows / DMA instructions of one wave per slice
   static_assert(NVMAX <= 256 && NVMAX % G == 0 && HT % (WAVES * G) == 0 && SHARE <= 64 && HT <= 65536, "geometry");
   static_assert((WAVES * WAVE_FLOATS + HT * PANEL) * 4 <= 163840, "accumulators + table must fit the CU's LDS");
   static_assert(U + NBW <= 63, "counted vmcnt wait");
   __shared__ __attribute__((aligned(16))) float s_all[WAVES * WAVE_FLOATS + HT * PANEL];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   const int wl = (int)blockIdx.x * WAVES + wave;         // the entry launches whole workgroups: wave_count % 8 == 0
   const int64_t w = (int64_t)a.wave_base + wl;
   float *my = s_all + wave * WAVE_FLOATS;
   float *table = s_all + WAVES * WAVE_FLOATS;
   for (int i = lane * 4; i < WAVE_FLOATS; i += 256)
      *reinterpret_cast<float4 *>(my + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
   for (int i = lane * 4; i < SHARE * PANEL; i += 256)    // the table too: its last row (and rows no slice fills) must read 0
      *reinterpret_cast<float4 *>(table + wave * SHARE * PANEL + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);
   const bool cok = lc * 4 < a.k;
   int ccol = lc * 4, vfirst = 0;
   if (cok && ccol + 4 > (int)a.k) { vfirst = ccol + 4 - (int)a.k; ccol = (int)a.k - 4; }
   const unsigned cbyte = (unsigned)ccol * 4u, poison = cok ? 0u : BUF_OOB;
   float *lane_base = my + lc * 4;
   const float *table_lane = table + lc * 4;
   // the same descriptor as rsrc, as four SGPRs for the DMA's asm operand; the table's address in LDS
   const unsigned long long ybase = (unsigned long long)a.y;
   const v4i_rsrc_t rsrc_dma = {(int)(unsigned)ybase, (int)((ybase >> 32) & 0xFFFFu), (int)a.ybytes, 0x00020000};
   const unsigned table_lds = (unsigned)(size_t)(__attribute__((address_space(3))) float *)table;
   const int64_t s0 = a.wave_step_off[w], s1 = a.wave_step_off[w + 1];
   const int64_t nwords = (s1 - s0) * G;
   const int32_t *wp = a.words + s0 * G;
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   const unsigned pad_word = ((unsigned)((lane % G) * PER) << 24) | a.null_word;
   const unsigned pad_hot = ((unsigned)((lane % G) * PER) << 24) | (unsigned)(HT - 1);
   auto load_words = [&](int64_t first, unsigned (&word)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         word[q] = i < nwords ? (unsigned)wp[i] : pad_word;
      }
   };
   unsigned wA[NBW], wB[NBW];                              // even / odd batches: used in turn, never copied (spmm_stream_kernel)
   v4i_t t[U];
   unsigned la[U];
   auto issue = [&](int u, const unsigned (&word_l)[NBW]) {
      const unsigned word = (unsigned)__shfl((int)word_l[(u * G) / 64], (u * G) % 64 + g);
      const unsigned o = (__umul24(word & 0xFFFFFFu, ldyb) + cbyte) | poison;
      la[u] = (word >> 24) * (unsigned)PANEL;
      t[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
   };
   // hot operands of a phase: the chunk's words (HWR registers of 64, loaded a phase ahead) and the ids of the table
   // rows this wave stages (one register, two phases ahead)
   const int S = a.slices;
   const int64_t *hoff = a.hot_step_off + w * S;
   // chunk offsets, relative to the wave's first, 63 phases per register (entries p0 .. p0 + 63), picked by v_readlane:
   // a load of hoff[p] at the point of use would be a vector load (the compiler cannot prove the array unwritten) whose
   // wait drains the gather pipeline; the next block's register is loaded a whole block ahead
   const int64_t hbase = hoff[0];
   const int hbase_lo = (int)hbase;
   // (the low words as loaded -- no arithmetic on the loaded value, which would need it at once; the difference of two
   // low words is the chunk offset as long as a wave's hot words stay below 2^31)
   auto load_hblock = [&](int p0) -> int { return p0 + lane <= S ? reinterpret_cast<const int *>(hoff)[2 * (p0 + lane)] : 0; };
   int hp0 = 0;
   int hblk = load_hblock(0), hblk_next = load_hblock(63);
   unsigned hw[HWR];
   int hn = 0;
   auto load_hot = [&](int p) {
      if (p - hp0 >= 63) {
         hp0 += 63;
         hblk = hblk_next;
         hblk_next = load_hblock(hp0 + 63);
      }
      const int o0 = __builtin_amdgcn_readlane(hblk, p - hp0) - hbase_lo, o1 = __builtin_amdgcn_readlane(hblk, p - hp0 + 1) - hbase_lo;
      hn = o1 - o0;
      const int32_t *hp = a.hot_words + (hbase + o0) * G;
#pragma unroll
      for (int q = 0; q < HWR; q++) {
         const int i = q * 64 + lane;
         hw[q] = i < hn * G ? (unsigned)hp[i] : pad_hot;
      }
   };
   auto load_ids = [&](int p) -> int {
      return (p < S && lane < SHARE) ? a.hot_rows[(size_t)p * HT + wave * SHARE + lane] : (int)a.null_word;
   };
   auto stage = [&](int ids) {
#pragma unroll
      for (int j = 0; j < SI; j++) {
         const unsigned cid = (unsigned)__shfl(ids, j * G + g);
         const unsigned o = (__umul24(cid, ldyb) + cbyte) | poison;
         lds_dma_b128(rsrc_dma, o, table_lds + (unsigned)((wave * SHARE + j * G) * PANEL * 4));
      }
   };
   unsigned cur = (unsigned)(g * PER * PANEL);
   float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   auto flush = [&]() {
      float4 *p = reinterpret_cast<float4 *>(lane_base + cur);
      float4 o = *p;
      o.x += acc[0]; o.y += acc[1]; o.z += acc[2]; o.w += acc[3];
      *p = o;
   };
   // prologue: the cold pipeline is filled, table 0 staged (own zeroing of the table first: DMA writes are not ordered
   // behind ds_writes), hot chunk 0 and the ids of table 1 loaded
   load_words(0, wA);
   load_words(64 * NBW, wB);
#pragma unroll
   for (int u = 0; u < U; u++) issue(u, wA);
   load_words(128 * NBW, wA);
   int ids = load_ids(0);
   asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
   stage(ids);
   ids = load_ids(1);
   load_hot(0);
   __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));       // (the builtin: the compiler's wait-count pass must see it, below)
   asm volatile("s_barrier" ::: "memory");
   // the cold stream in PAIRS of batches (A, B): a phase is a whole number of pairs
   const int64_t npairs = ((nwords + 64 * NBW - 1) / (64 * NBW) + 1) / 2;
   int64_t b = 0;                                          // pairs done
   const int pairs_q = (int)(npairs / S), pairs_r = (int)(npairs % S);
   int pairs_err = 0;
   auto run_batch = [&](int64_t bb, unsigned (&wnext)[NBW]) {
#pragma unroll
      for (int u = 0; u < U; u++) {
         if (la[u] != cur) {
            flush();
            cur = la[u];
            acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
         }
#pragma unroll
         for (int v = 0; v < 4; v++) acc[v] += __int_as_float(t[u][v]);
         __builtin_amdgcn_sched_barrier(0);                // the old t[u] is consumed before the new one is issued: no copy
         issue(u, wnext);
      }
      load_words((bb + 3) * 64 * NBW, wnext);
   };
   for (int p = 0; p < S; p++) {
      // ---- hot chunk p: groups of four steps (four ds_bpermute, four ds_read_b128, then the adds) ----
#ifdef ISPLIB_EXP_HYB_SKIP_HOT                            // timing experiment: what the schedule costs with the hot edges free
      if (false) {
#else
      if (hn > 0) {
#endif
         flush();
         acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
#pragma unroll
         for (int q = 0; q < HWR; q++) {
            if (q * HS >= hn) break;
#pragma unroll
            for (int j0 = 0; j0 < HS; j0 += 4) {
               if (q * HS + j0 >= hn) break;
               unsigned wd[4];
               float4 xv[4];
#pragma unroll
               for (int i = 0; i < 4; i++) wd[i] = (unsigned)__shfl((int)hw[q], (j0 + i) * G + g);
#pragma unroll
               for (int i = 0; i < 4; i++) xv[i] = *reinterpret_cast<const float4 *>(table_lane + (wd[i] & 0xFFFFu) * (unsigned)PANEL);
#pragma unroll
               for (int i = 0; i < 4; i++) {
                  const unsigned lrow = (wd[i] >> 24) * (unsigned)PANEL;
                  if (lrow != cur) {
                     flush();
                     cur = lrow;
                     acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
                  }
                  acc[0] += xv[i].x; acc[1] += xv[i].y; acc[2] += xv[i].z; acc[3] += xv[i].w;
               }
            }
         }
         flush();
         acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
      }
      // everyone is done with table p: the next one may land on it
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (p + 1 < S) {
         stage(ids);
         ids = load_ids(p + 2);
         load_hot(p + 1);
      } else {
         hn = 0;
      }
      // ---- the cold batches of phase p: spmm_stream_kernel's loop ----
      // this phase's share of the pairs (spread evenly; never none: a wave with fewer pairs than slices runs pairs of
      // padding words, gathers the range check answers with 0).  After at least one pair the DMA loads of table p+1 and
      // the loads of the next hot words and ids are older than the last batch's U gathers and NBW word loads:
      // vmcnt(U + NBW) says they have landed and leaves the gathers in flight.  The wait is the builtin, straight after
      // a loop that always runs, so that the compiler's own wait-count pass sees it and knows the hot words are there
      // (with the wait in inline asm, or behind an `if (any pairs) ... else wait for everything`, it guarded their first
      // use and the next staging with vmcnt(0): two drained pipelines per slice).
      int quota = pairs_q;
      pairs_err += pairs_r;
      if (pairs_err >= S) { pairs_err -= S; quota++; }
      const int64_t b_end = b + (quota > 0 ? quota : 1);
      do {
         run_batch(2 * b, wB);
         run_batch(2 * b + 1, wA);
      } while (++b < b_end);
      __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(U + NBW));
      asm volatile("s_barrier" ::: "memory");              // table p+1 complete
   }
   flush();
#pragma unroll 1
   for (int jj = 0; jj < PER; jj++) {
      const int lrow = g * PER + jj;
      const int row = a.wave_row[(size_t)w * NVMAX + lrow];
      if (row < 0 || !cok) continue;
      const int part = a.wave_part[(size_t)w * NVMAX + lrow];
      const float4 t4 = *reinterpret_cast<const float4 *>(lane_base + lrow * PANEL);
      float v[4] = {t4.x, t4.y, t4.z, t4.w};
      int bi[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
      const int c = ccol;
      if (part >= 0) {
         store_tail<4>(a.part_val + (size_t)part * (size_t)a.k + c, v, vfirst);
         continue;
      }
      int64_t arg[4];
      finish_row<OP_ADD>(a, row, c, v, bi, arg);
      store_tail<4>(a.z + (size_t)row * (size_t)a.ldz + c, v, vfirst);
   }
}

template <int OP, int LPR, int ADDR>
static int launch_sweep_nv(const SweepArgs &a, int nvmax, hipStream_t st) {
   const unsigned blocks = (unsigned)((a.wave_count + 3) / 4);
   if (blocks == 0) return ISPLIB_SUCCESS;
   if (nvmax == 8) hipLaunchKernelGGL((spmm_sweep_kernel<OP, LPR, ADDR, 8>), dim3(blocks), dim3(256), 0, st, a);
   else if (nvmax == 16) hipLaunchKernelGGL((spmm_sweep_kernel<OP, LPR, ADDR, 16>), dim3(blocks), dim3(256), 0, st, a);
   else if constexpr (OP == OP_ADD) hipLaunchKernelGGL((spmm_sweep_kernel<OP, LPR, ADDR, 32>), dim3(blocks), dim3(256), 0, st, a);
   else return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: max / min need rows_per_wave <= 16");
   return check_launch("spmm_sweep_kernel");
}

template <int OP, int ADDR>
static int launch_sweep_op(const SweepArgs &a, int nvmax, hipStream_t st) {
   const int64_t width = (a.k + 3) / 4;
   if (width <= 8) return launch_sweep_nv<OP, 8, ADDR>(a, nvmax, st);
   if (width <= 16) return launch_sweep_nv<OP, 16, ADDR>(a, nvmax, st);
   if constexpr (OP == OP_ADD) return launch_sweep_nv<OP, 32, ADDR>(a, nvmax, st);
   return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: max / min panels are at most 64 columns");
}

// waves of one launch that are resident together (what a plan's waves_per_gen should not exceed)
static int sweep_resident_waves(bool add, int64_t pk, int nvmax, int cus) {
   const int lpr = pk <= 32 ? 8 : (pk <= 64 ? 16 : 32);
   const int lds = (add ? 1 : 2) * 4 * nvmax * lpr * 4 * 4;
   int wgs = 163840 / lds;
   if (wgs > 8) wgs = 8;
   return cus * wgs * 4;
}



int g_sweep_panel = 64;     // tuning knob (isplib_hip_tune_experimental(9, w)): column-panel width of the sweep schedule, 32 / 64 / 128

}  // namespace isplib

using namespace isplib;

extern "C" int isplib_spmm_sweep_resident_waves(int32_t imessage, int64_t k, int rows_per_wave) {
   clear_error();
   if (k <= 0 || (rows_per_wave != 8 && rows_per_wave != 16 && rows_per_wave != 32)) return 0;
   int dev = 0, cus = 0;
   if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      (void)hipGetLastError();
      cus = 256;                                          // MI355X
   }
   const bool add = (imessage & 0xF0000) == ISPLIB_AOP_ADD;
   int panel = g_sweep_panel;
   if (panel != 32 && panel != 64 && panel != 128) panel = 64;
   if (!add && panel > 64) panel = 64;
   return sweep_resident_waves(add, k < panel ? k : panel, rows_per_wave, cus);
}

extern "C" size_t isplib_spmm_sweep_workspace_bytes(int32_t imessage, const isplib_sweep_plan *plan, int64_t k) {
   if (!plan || plan->n_parts <= 0 || k <= 0) return 256;
   const int64_t pk = k < 128 ? k : 128;                  // widest panel of a pass
   const size_t plane = ((size_t)plan->n_parts * (size_t)pk * sizeof(float) + 255) & ~(size_t)255;
   return plane * (((imessage & 0xF0000) != ISPLIB_AOP_ADD) ? 2 : 1);
}

extern "C" int fusedMM_csr_sweep_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const float *val,
                                     const int64_t *indx, const int32_t *indx32, const int64_t *pntrb,
                                     const int64_t *pntre, const isplib_sweep_plan *plan, const float *y, int64_t ldy,
                                     float *z, int64_t ldz, int64_t *z_arg, void *workspace, size_t workspace_bytes,
                                     const isplib_epilogue *ep, void *stream) {
   clear_error();
   const int32_t vop = imessage & 0xF, rop = imessage & 0xF0, sop = imessage & 0xF00, vsc = imessage & 0xF000,
                 aop = imessage & 0xF0000;
   if (vop != ISPLIB_VOP_COPY_RHS || rop != ISPLIB_ROP_NOOP || sop != ISPLIB_SOP_COPY ||
       (vsc != ISPLIB_VSC_MUL && vsc != ISPLIB_VSC_MEAN) ||
       (aop != ISPLIB_AOP_ADD && aop != ISPLIB_AOP_MAX && aop != ISPLIB_AOP_MIN) ||
       (vsc == ISPLIB_VSC_MEAN && aop != ISPLIB_AOP_ADD))
      return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_sweep_hip: message outside the SpMM set");
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: negative dimension");
   if (m == 0 || k == 0) return ISPLIB_SUCCESS;
   if (!plan) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: plan is required");
   if (plan->rows != m) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: the plan was built for another row count");
   if (plan->gens < 1 || plan->waves_per_gen < 1 || (plan->rows_per_wave != 8 && plan->rows_per_wave != 16 && plan->rows_per_wave != 32))
      return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: bad plan geometry (rows_per_wave must be 8, 16 or 32)");
   if (aop != ISPLIB_AOP_ADD && plan->rows_per_wave > 16)
      return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: max / min need rows_per_wave <= 16 (two LDS planes)");
   if ((k % 4) != 0 || (ldy % 4) != 0 || (ldz % 4) != 0 || ((uintptr_t)y & 15) != 0 || ((uintptr_t)z & 15) != 0)
      return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: k, ldy, ldz must be multiples of 4 and y, z 16-byte aligned (use fusedMM_csr_tasks_hip)");
   if (ldy < k || ldz < k) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: leading dimension smaller than k");
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > BUF_LIMIT) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: dense operand larger than 3.5 GiB (use fusedMM_csr_hip)");
   if (!pntrb || !pntre || !z || !y || (nnz > 0 && !indx) || !plan->wave_row || !plan->wave_part || !plan->wave_task_off ||
       (plan->n_tasks > 0 && (!plan->task_b || !plan->task_meta)) || (plan->n_hub > 0 && (!plan->hub_row || !plan->hub_off)))
      return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: null operand");
   const size_t need = isplib_spmm_sweep_workspace_bytes(imessage, plan, k);
   if (plan->n_parts > 0) {
      if (!workspace || workspace_bytes < need) return fail(ISPLIB_NOT_ENOUGH_MEM, "fusedMM_csr_sweep_hip: workspace too small");
      if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: workspace must be 256-byte aligned");
   }
   SweepArgs a = {};
   a.empty_init = empty_row_init();
   a.k = k; a.nnz = nnz; a.val = val; a.indx = indx; a.indx32 = indx32; a.pntrb = pntrb; a.pntre = pntre;
   a.y = y; a.ldy = ldy; a.ybytes = (unsigned)yb; a.z = z; a.ldz = ldz; a.z_arg = z_arg;
   a.mean = (vsc == ISPLIB_VSC_MEAN) ? 1 : 0;
   a.wave_row = plan->wave_row; a.wave_part = plan->wave_part; a.wave_task_off = plan->wave_task_off;
   a.task_b = plan->task_b; a.task_meta = plan->task_meta;
   a.wave_base = 0; a.wave_count = 0;
   a.hub_row = plan->hub_row; a.hub_off = plan->hub_off; a.n_hub = plan->n_hub;
   a.ep_row_scale = a.ep_self = a.ep_bias = nullptr; a.ep_ld_self = 0; a.ep_relu = 0;
   if (ep) {
      if (aop != ISPLIB_AOP_ADD) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: the epilogue is defined for sum / mean only");
      if (ep->self && ep->ld_self < k) return fail(ISPLIB_FAIL, "fusedMM_csr_sweep_hip: ld_self smaller than k");
      a.ep_row_scale = ep->row_scale; a.ep_self = ep->self; a.ep_ld_self = ep->ld_self; a.ep_bias = ep->bias;
      a.ep_relu = ep->relu ? 1 : 0;
   }
   const int64_t pk_max = k < 128 ? k : 128;
   const size_t plane = ((size_t)(plan->n_parts > 0 ? plan->n_parts : 0) * (size_t)pk_max * sizeof(float) + 255) & ~(size_t)255;
   a.part_val = (float *)workspace;
   a.part_idx = (aop == ISPLIB_AOP_ADD || !workspace) ? nullptr : (int *)((char *)workspace + plane);
   hipStream_t st = (hipStream_t)stream;
   // column panels: one complete sweep (every generation, then the hub fold) per panel on the same stream; panels
   // never change a result.  64 columns = 256-byte gathers, 16 KB of LDS per workgroup at 16 rows per wave.
   int panel = g_sweep_panel;
   if (panel != 32 && panel != 64 && panel != 128) panel = 64;
   if ((ldy % 32) != 0 && panel < 128) panel = 128;       // rows that are not whole cache lines: fewer, wider panels
   if (aop != ISPLIB_AOP_ADD && panel > 64) panel = 64;   // two LDS planes
   const int64_t pw = k > panel ? panel : k;
   for (int64_t c0 = 0; c0 < k; c0 += pw) {
      SweepArgs p = a;
      p.k = (k - c0) < pw ? (k - c0) : pw;
      p.y = y + c0;
      p.z = z + c0;
      p.ep_self = a.ep_self ? a.ep_self + c0 : nullptr;
      p.ep_bias = a.ep_bias ? a.ep_bias + c0 : nullptr;
      p.z_arg = z_arg ? z_arg + c0 : nullptr;
      p.ybytes = (unsigned)(yb - (unsigned long long)c0 * 4ull);
      for (int gen = 0; gen < plan->gens; gen++) {
         p.wave_base = gen * plan->waves_per_gen;
         p.wave_count = plan->waves_per_gen;
         int rc;
         if (aop == ISPLIB_AOP_ADD) rc = val ? launch_sweep_op<OP_ADD, 2>(p, plan->rows_per_wave, st) : launch_sweep_op<OP_ADD, 1>(p, plan->rows_per_wave, st);
         else if (aop == ISPLIB_AOP_MAX) rc = val ? launch_sweep_op<OP_MAX, 2>(p, plan->rows_per_wave, st) : launch_sweep_op<OP_MAX, 1>(p, plan->rows_per_wave, st);
         else rc = val ? launch_sweep_op<OP_MIN, 2>(p, plan->rows_per_wave, st) : launch_sweep_op<OP_MIN, 1>(p, plan->rows_per_wave, st);
         if (rc) return rc;
      }
      if (plan->n_hub > 0) {
         int64_t blocks = (plan->n_hub * (p.k / 4) + 255) / 256;
         if (blocks > 4096) blocks = 4096;
         if (aop == ISPLIB_AOP_ADD) hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_ADD, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         else if (aop == ISPLIB_AOP_MAX) hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_MAX, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         else hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_MIN, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         const int rc = check_launch("sweep_hub_fold_kernel");
         if (rc) return rc;
      }
   }
   return ISPLIB_SUCCESS;
}

// ---- SDDMM on the stream front end: entry ---------------------------------------------------------------------------
template <int LPR, bool ACCUM>
static int launch_sddmm_stream(const SweepArgs &a, hipStream_t st) {
   const unsigned blocks = (unsigned)((a.wave_count + 3) / 4);
   if (blocks == 0) return ISPLIB_SUCCESS;
   if constexpr (LPR == 32) hipLaunchKernelGGL((sddmm_stream_kernel<32, 32, 1, 2, ACCUM>), dim3(blocks), dim3(256), 0, st, a);
   else if constexpr (LPR == 16) hipLaunchKernelGGL((sddmm_stream_kernel<16, ISPLIB_STREAM_NV4, ISPLIB_STREAM_NBW4, ISPLIB_STREAM_WGS4, ACCUM>), dim3(blocks), dim3(256), 0, st, a);
   else hipLaunchKernelGGL((sddmm_stream_kernel<8, ISPLIB_STREAM_NV8, ISPLIB_STREAM_NBW8, ISPLIB_STREAM_WGS8, ACCUM>), dim3(blocks), dim3(256), 0, st, a);
   return check_launch("sddmm_stream_kernel");
}

extern "C" int isplib_sddmm_stream_hip(int64_t m, int64_t n, int64_t k, int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                                       const isplib_stream_plan *plan, const float *y, int64_t ldy, const float *g, int64_t ldg,
                                       int mean, float *dval, void *stream) {
   clear_error();
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: negative dimension");
   if (m == 0 || nnz == 0) return ISPLIB_SUCCESS;
   if (!plan) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: plan is required");
   if (plan->rows != m || plan->cols != n) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: the plan was built for another shape");
   if (n >= (1LL << 24) || ldy >= (1LL << 22)) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: n must be < 2^24 and ldy < 2^22 (24-bit address arithmetic)");
   if (nnz >= (1LL << 31)) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: nnz < 2^31 required (32-bit CSR positions in the plan)");
   if (plan->streams != 2 && plan->streams != 4 && plan->streams != 8)
      return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: bad plan geometry (streams 2, 4 or 8)");
   if (plan->gens < 1 || plan->waves_per_gen < 1 || plan->rows_per_wave != stream_geom(plan->streams).nvmax)
      return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: bad plan geometry (a sum / mean plan of isplib_spmm_stream_geometry is required)");
   if (plan->n_steps > 0 && !plan->perm) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: the plan's perm array is required (the CSR position of every word)");
   if (k < 4) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: k >= 4 required (use isplib_sddmm_csr_hip)");
   if (ldy < k || ldg < k) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: leading dimension smaller than k");
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > BUF_LIMIT) return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: dense operand larger than 3.5 GiB (use isplib_sddmm_csr_hip)");
   if (!pntrb || !pntre || !y || !g || !dval || !plan->wave_row || !plan->wave_step_off || (plan->n_steps > 0 && !plan->words))
      return fail(ISPLIB_FAIL, "isplib_sddmm_stream_hip: null operand");
   SweepArgs a = {};
   a.empty_init = empty_row_init();
   a.k = k; a.nnz = nnz; a.pntrb = pntrb; a.pntre = pntre;
   a.y = y; a.ldy = ldy; a.ybytes = (unsigned)yb;
   a.mean = mean ? 1 : 0;
   a.ids = plan->perm;
   a.wave_row = plan->wave_row; a.wave_part = plan->wave_part;
   a.words = plan->words; a.wave_step_off = plan->wave_step_off; a.null_word = (unsigned)n;
   a.g = g; a.ldg = ldg; a.dval = dval;
   hipStream_t st = (hipStream_t)stream;
   const int64_t pw = 256 / plan->streams;
   bool first = true;
   for (int64_t c0 = 0; c0 < k; c0 += pw) {
      SweepArgs p = a;
      p.k = (k - c0) < pw ? (k - c0) : pw;
      int64_t at = c0;
      p.ep_relu = 0;                               // (reused: leading columns of the panel that an earlier panel covered)
      if (p.k < 4) {                               // a sliver of 1-3 columns: the panel is the last four columns, of which the
         p.ep_relu = (int)(4 - p.k);               // first 4 - sliver were part of the previous panel's dot products and count as 0
         p.k = 4;
         at = k - 4;
      }
      p.y = y + at;
      p.g = g + at;
      p.ybytes = (unsigned)(yb - (unsigned long long)at * 4ull);
      for (int gen = 0; gen < plan->gens; gen++) {
         p.wave_base = gen * plan->waves_per_gen;
         p.wave_count = plan->waves_per_gen;
         int rc;
         if (plan->streams == 2) rc = first ? launch_sddmm_stream<32, false>(p, st) : launch_sddmm_stream<32, true>(p, st);
         else if (plan->streams == 4) rc = first ? launch_sddmm_stream<16, false>(p, st) : launch_sddmm_stream<16, true>(p, st);
         else rc = first ? launch_sddmm_stream<8, false>(p, st) : launch_sddmm_stream<8, true>(p, st);
         if (rc) return rc;
      }
      first = false;
   }
   return ISPLIB_SUCCESS;
}

// ---- hybrid form: entry ---------------------------------------------------------------------------------------------
struct HybridGeom { int nvmax, nbw, ht, hwr; };
static HybridGeom hybrid_geom(int streams) {
   if (streams == 8) return {ISPLIB_HYB8_NV, ISPLIB_HYB8_NBW, ISPLIB_HYB8_HT, ISPLIB_HYB8_HWR};
   return {ISPLIB_HYB4_NV, ISPLIB_HYB4_NBW, ISPLIB_HYB4_HT, ISPLIB_HYB4_HWR};
}

extern "C" int isplib_spmm_hybrid_geometry(int streams, int *rows_per_wave, int *waves_resident, int *table_rows, int *hot_cap) {
   clear_error();
   if (streams != 4 && streams != 8) return fail(ISPLIB_FAIL, "isplib_spmm_hybrid_geometry: streams must be 4 (64-column slots) or 8 (32-column slots)");
   const HybridGeom ge = hybrid_geom(streams);
   if (rows_per_wave) *rows_per_wave = ge.nvmax;
   if (waves_resident) *waves_resident = device_cus() * 8;            // one workgroup of 8 waves per CU
   if (table_rows) *table_rows = ge.ht;
   if (hot_cap) *hot_cap = ge.hwr * (64 / streams);
   return ISPLIB_SUCCESS;
}

extern "C" size_t isplib_spmm_hybrid_workspace_bytes(const isplib_hybrid_plan *plan) {
   return isplib_spmm_stream_workspace_bytes(plan ? &plan->cold : nullptr);
}

extern "C" int fusedMM_csr_hybrid_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const int64_t *pntrb,
                                      const int64_t *pntre, const isplib_hybrid_plan *hp, const float *y, int64_t ldy, float *z,
                                      int64_t ldz, void *workspace, size_t workspace_bytes, const isplib_epilogue *ep, void *stream) {
   clear_error();
   if (imessage != ISPLIB_MSG_SPMM_SUM && imessage != ISPLIB_MSG_SPMM_MEAN)
      return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_hybrid_hip: sum and mean only");
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: negative dimension");
   if (m == 0 || k == 0) return ISPLIB_SUCCESS;
   if (!hp) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: plan is required");
   const isplib_stream_plan *plan = &hp->cold;
   if (plan->rows != m || plan->cols != n) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: the plan was built for another shape");
   if (n >= (1LL << 24) || ldy >= (1LL << 22)) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: n must be < 2^24 and ldy < 2^22 (24-bit address arithmetic)");
   if (plan->streams != 4 && plan->streams != 8) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: bad plan geometry (streams 4 or 8)");
   const HybridGeom ge = hybrid_geom(plan->streams);
   if (plan->gens < 1 || plan->waves_per_gen < 8 || (plan->waves_per_gen % 8) != 0 || plan->rows_per_wave != ge.nvmax ||
       hp->table_rows != ge.ht || hp->hot_cap > ge.hwr * (64 / plan->streams) || plan->slices < 1)
      return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: bad plan geometry (isplib_spmm_hybrid_geometry reports rows per wave, table rows and the hot-step cap; waves_per_gen must be a multiple of 8)");
   if (plan->vals) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: unit weights only (weighted graphs: fusedMM_csr_stream_hip)");
   if (k < 4) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: k >= 4 required (use fusedMM_csr_hip)");
   if (ldy < k || ldz < k) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: leading dimension smaller than k");
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > BUF_LIMIT) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: dense operand larger than 3.5 GiB (use fusedMM_csr_hip)");
   if (!pntrb || !pntre || !z || !y || !plan->wave_row || !plan->wave_part || !plan->wave_step_off || (plan->n_steps > 0 && !plan->words) ||
       (plan->n_hub > 0 && (!plan->hub_row || !plan->hub_off)) || !hp->hot_rows || !hp->hot_step_off || (hp->n_hot_steps > 0 && !hp->hot_words))
      return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: null operand");
   if (plan->n_parts > 0) {
      if (!workspace || workspace_bytes < isplib_spmm_stream_workspace_bytes(plan)) return fail(ISPLIB_NOT_ENOUGH_MEM, "fusedMM_csr_hybrid_hip: workspace too small");
      if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: workspace must be 256-byte aligned");
   }
   SweepArgs a = {};
   a.empty_init = empty_row_init();
   a.k = k; a.nnz = nnz; a.pntrb = pntrb; a.pntre = pntre;
   a.y = y; a.ldy = ldy; a.ybytes = (unsigned)yb; a.z = z; a.ldz = ldz;
   a.mean = imessage == ISPLIB_MSG_SPMM_MEAN ? 1 : 0;
   a.abs_ids = 1;
   a.wave_row = plan->wave_row; a.wave_part = plan->wave_part;
   a.words = plan->words; a.wave_step_off = plan->wave_step_off; a.null_word = (unsigned)n;
   a.hub_row = plan->hub_row; a.hub_off = plan->hub_off; a.n_hub = plan->n_hub;
   a.hot_rows = hp->hot_rows; a.hot_words = hp->hot_words; a.hot_step_off = hp->hot_step_off; a.slices = plan->slices;
   a.part_val = (float *)workspace;
   if (ep) {
      if (ep->self && ep->ld_self < k) return fail(ISPLIB_FAIL, "fusedMM_csr_hybrid_hip: ld_self smaller than k");
      a.ep_row_scale = ep->row_scale; a.ep_self = ep->self; a.ep_ld_self = ep->ld_self; a.ep_bias = ep->bias;
      a.ep_relu = ep->relu ? 1 : 0;
   }
   hipStream_t st = (hipStream_t)stream;
   const int64_t pw = 256 / plan->streams;
   for (int64_t c0 = 0; c0 < k; c0 += pw) {
      SweepArgs p = a;
      p.k = (k - c0) < pw ? (k - c0) : pw;
      if (p.k < 4) {
         p.k = 4;
         c0 = k - 4;
      }
      p.y = y + c0;
      p.z = z + c0;
      p.ep_self = a.ep_self ? a.ep_self + c0 : nullptr;
      p.ep_bias = a.ep_bias ? a.ep_bias + c0 : nullptr;
      p.ybytes = (unsigned)(yb - (unsigned long long)c0 * 4ull);
      for (int gen = 0; gen < plan->gens; gen++) {
         p.wave_base = gen * plan->waves_per_gen;
         p.wave_count = plan->waves_per_gen;
         const unsigned blocks = (unsigned)(p.wave_count / 8);
         if (plan->streams == 4)
            hipLaunchKernelGGL((spmm_hybrid_kernel<16, ISPLIB_HYB4_NV, ISPLIB_HYB4_NBW, ISPLIB_HYB4_HT, ISPLIB_HYB4_HWR>), dim3(blocks), dim3(512), 0, st, p);
         else
            hipLaunchKernelGGL((spmm_hybrid_kernel<8, ISPLIB_HYB8_NV, ISPLIB_HYB8_NBW, ISPLIB_HYB8_HT, ISPLIB_HYB8_HWR>), dim3(blocks), dim3(512), 0, st, p);
         const int rc = check_launch("spmm_hybrid_kernel");
         if (rc) return rc;
      }
      if (plan->n_hub > 0) {
         const bool v4 = (p.k % 4) == 0 && (p.ldz % 4) == 0 && ((uintptr_t)p.z & 15) == 0 && (!p.ep_self || ((p.ep_ld_self % 4) == 0 && ((uintptr_t)p.ep_self & 15) == 0));
         int64_t blocks = (plan->n_hub * (v4 ? p.k / 4 : p.k) + 255) / 256;
         if (blocks > 4096) blocks = 4096;
         if (v4) hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_ADD, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         else hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_ADD, 1>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         const int rc = check_launch("sweep_hub_fold_kernel");
         if (rc) return rc;
      }
   }
   return ISPLIB_SUCCESS;
}

extern "C" int isplib_internal_set_sddmm_panel_cols(int cols);      // libisplib_hip.so (spmm.hip); not a public entry

extern "C" int isplib_hip_tune_experimental(int key, int value) {
   if (key == 9 && (value == 32 || value == 64 || value == 128)) { g_sweep_panel = value; return ISPLIB_SUCCESS; }
   if (key == 12) return isplib_internal_set_sddmm_panel_cols(value);
   return ISPLIB_FAIL;
}
