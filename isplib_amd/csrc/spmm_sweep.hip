// spmm_sweep.hip -- the STREAM schedule of the SpMM (fusedMM_csr_stream_hip, fusedMM_csr_stream_minmax_hip: the default of
// sum / mean / max / min on graphs with work for the whole chip): see include/isplib_hip.h and DESIGN.md 4.3-4.4.
//
// The task-list schedule (spmm_tasks.hip) gets its L2 locality from the ORDER in which the hardware hands out
// workgroups (slice-major task list), and pays for it with one partial row per task, written to HBM and folded by a
// second kernel.  Here the locality comes from TIME instead: a launch holds only as many waves as are resident at
// once, every wave owns a fixed set of (virtual) rows whose running sums live in LDS, and all waves walk the column
// slices 0, 1, 2, ... of their own rows together.  Equal edge mass per wave (plan) keeps them in step, so at any
// moment an XCD's L2 serves one or two slices -- which can therefore be as small as the L2 -- and nothing but the
// finished rows ever leaves the CU: no partial rows (but for hub rows cut into virtual rows), no fold, z written once.
// (The file keeps its name from the round-2 "sweep" schedule, the first of this family; that kernel, the LDS hot-row hybrid
// and the stream-plan SDDMM -- all measured slower than what is here -- live in experimental/experimental.hip.)
#include "sweep_common.h"

namespace isplib {

// ---- stream form ------------------------------------------------------------------------------------------------
// The sweep above still pays one latency chain per (row, slice) segment -- and with L2-sized slices a segment is
// 15 edges.  Here a wave does not see segments at all.  The plan gives each of the G = 64 / LPR slots of a wave its
// own STREAM: the edges of the slot's NVMAX / G rows, slice by slice, as 4-byte words (local row << 24 | column) in
// the plan's own copy of the index array.  Step i of a wave gathers word i of each of its G streams -- one 1-KiB
// buffer load, always full -- and every lane adds the four floats it receives to the running sum of the row its
// slot is on (registers), which moves to and from the slot's LDS row when the stream changes rows.  The slots of a
// wave own disjoint rows and a wave's LDS operations execute in order, so every sum is formed in one fixed order.
// U gathers are in flight per wave at all times, across row and slice boundaries alike; there is no butterfly, no
// masked tail, no per-segment bookkeeping.  Sum / mean only.
template <int LPR, bool HAS_VAL, int NVMAX, int NBW, int WGS>
__global__ __launch_bounds__(256, (stream_wgs_per_cu<LPR, NVMAX, WGS>())) void spmm_stream_kernel(const SweepArgs a) {
   constexpr int WAVES = 4, G = 64 / LPR, PANEL = LPR * 4, U = 64 * NBW / G;     // a batch = NBW x 64 words = U steps
   constexpr int PER = NVMAX / G;                         // rows of a slot
   constexpr int WAVE_FLOATS = NVMAX * PANEL;
   static_assert(NVMAX <= 256 && NVMAX % G == 0, "the local row is the top byte of a word");
   __shared__ __attribute__((aligned(16))) float s_all[WAVES * WAVE_FLOATS];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   const int wl = (int)blockIdx.x * WAVES + wave;
   if (wl >= a.wave_count) return;                       // no barrier anywhere below
   const int64_t w = (int64_t)a.wave_base + wl;
#ifdef ISPLIB_EXP_WAVE_TIMES
   const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
   float *my = s_all + wave * WAVE_FLOATS;
   for (int i = lane * 4; i < WAVE_FLOATS; i += 256)
      *reinterpret_cast<float4 *>(my + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);
   // a lane holds columns lc*4 .. lc*4+3 of the panel; when k is not a multiple of 4 the last lane's vector is shifted
   // back to END at column k (its first `vfirst` components repeat the neighbour's columns and are never stored), so
   // no load reaches past a row and rows need only 4-byte alignment (the GCN's K = 41 runs here instead of the task list)
   const bool cok = lc * 4 < a.k;
   int ccol = lc * 4, vfirst = 0;
   if (cok && ccol + 4 > (int)a.k) { vfirst = ccol + 4 - (int)a.k; ccol = (int)a.k - 4; }
   const unsigned cbyte = (unsigned)ccol * 4u, poison = cok ? 0u : BUF_OOB;
   float *lane_base = my + lc * 4;                        // a lane's four columns of a row are contiguous
   const int64_t s0 = a.wave_step_off[w], s1 = a.wave_step_off[w + 1];
   const int64_t nwords = (s1 - s0) * G;
   const int32_t *wp = a.words + s0 * G;
   const float *vp = HAS_VAL ? a.vals + s0 * G : nullptr;
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   // lane i of batch register q holds word q*64 + i = (step (q*64 + i) / G, slot i % G); past the end of the wave: the
   // padding word of the slot (column n: the gather reads 0 through the range check; the row is one of the slot's own)
   const unsigned pad_word = ((unsigned)((lane % G) * PER) << 24) | a.null_word;
   auto load_words = [&](int64_t first, unsigned (&word)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         word[q] = i < nwords ? (unsigned)wp[i] : pad_word;
      }
   };
   auto load_vals = [&](int64_t first, float (&val)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         val[q] = HAS_VAL && i < nwords ? vp[i] : 0.0f;
      }
   };
   unsigned w1[NBW], w2[NBW];                             // the words of the next batch and of the one after it
   float v0[NBW], v1[NBW];                                // the weights of the batch being consumed and of the next: a weight is
                                                          // needed one batch later than its word, so it is loaded one batch later
   v4i_t t[U];
   unsigned la[U];
   // one ds_bpermute hands a slot its word of the step (the LDS pipe is otherwise idle; picking it with v_readlane +
   // v_cndmask cost 16 vector instructions per step); column * row pitch is a 24-bit multiply (n < 2^24, pitch < 2^24,
   // product < 2^32: checked by the entry)
   // (a weight is fetched from its batch register when its gather is consumed, one step ahead: a ring of U weights
   // beside the U gathers in flight costs 30 registers)
   auto issue = [&](int u, const unsigned (&word_l)[NBW]) {
      const unsigned word = (unsigned)__shfl((int)word_l[(u * G) / 64], (u * G) % 64 + g);
      const unsigned o = (__umul24(word & 0xFFFFFFu, ldyb) + cbyte) | poison;
      la[u] = (word >> 24) * (unsigned)PANEL;
      t[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, ISPLIB_EXP_GATHER_AUX);
   };
   load_words(0, w1);
   load_vals(0, v0);
#pragma unroll
   for (int u = 0; u < U; u++) issue(u, w1);
   load_words(64 * NBW, w1);
   load_vals(64 * NBW, v1);
   load_words(128 * NBW, w2);
   // The row a slot is working on keeps its running sum in registers; it moves to the slot's LDS row when the stream
   // turns to another row (every ~deg / slices edges) and is picked up again from there when the stream comes back
   // in the next slice.  Plain read-add-write by the only lanes that ever touch that LDS row: LDS float atomics
   // (one ds_add_f32 per gathered float) ran 25x slower than the gathers they were meant to keep up with.
   unsigned cur = (unsigned)(g * PER * PANEL);
   float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   auto flush = [&]() {
      float4 *p = reinterpret_cast<float4 *>(lane_base + cur);
      float4 o = *p;
      o.x += acc[0]; o.y += acc[1]; o.z += acc[2]; o.w += acc[3];
      *p = o;
   };
   const int64_t nb = (nwords + 64 * NBW - 1) / (64 * NBW);
#ifdef ISPLIB_EXP_WAVE_TIMES
   const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
   for (int64_t b = 0; b < nb; b++) {
      // the U gathers of batch b are in flight; each one consumed is replaced by the same step of batch b + 1
      float vnext = HAS_VAL ? __shfl(v0[0], g) : 0.0f;
#pragma unroll
      for (int u = 0; u < U; u++) {
         const float vcur = vnext;
         if (HAS_VAL && u + 1 < U) vnext = __shfl(v0[((u + 1) * G) / 64], ((u + 1) * G) % 64 + g);
         if (la[u] != cur) {                             // per lane: the slots of a wave change rows at different steps
            flush();
            cur = la[u];
            acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
         }
#pragma unroll
         for (int v = 0; v < 4; v++) {
            const float x = __int_as_float(t[u][v]);
            acc[v] = HAS_VAL ? fmaf(vcur, x, acc[v]) : acc[v] + x;
         }
         issue(u, w1);
      }
#pragma unroll
      for (int q = 0; q < NBW; q++) { w1[q] = w2[q]; v0[q] = v1[q]; }
      load_words((b + 3) * 64 * NBW, w2);
      load_vals((b + 2) * 64 * NBW, v1);
   }
   flush();
#ifdef ISPLIB_EXP_WAVE_TIMES
   const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif
   // write-out: slot q owns the local rows [q * PER, (q + 1) * PER); its LPR lanes hold one row of the panel.  The rows' ids
   // (two dependent global loads per row) are fetched for the whole slot first and the rows then go out four at a time: a
   // row-at-a-time loop pays every row's memory latencies one after the other (2 % of a dispatch, scripts/exp_wave_times.py)
   int row_[PER], part_[PER];
#pragma unroll
   for (int jj = 0; jj < PER; jj++) {
      row_[jj] = cok ? a.wave_row[(size_t)w * NVMAX + g * PER + jj] : -1;
      part_[jj] = a.wave_part[(size_t)w * NVMAX + g * PER + jj];
   }
#pragma unroll 4
   for (int jj = 0; jj < PER; jj++) {
      const int lrow = g * PER + jj;
      const int row = row_[jj];
      if (row < 0) continue;
      const int part = part_[jj];
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
#ifdef ISPLIB_EXP_WAVE_TIMES
   if (a.dbg && lane == 0) {
      a.dbg[(size_t)wl * 4 + 0] = t_start;
      a.dbg[(size_t)wl * 4 + 1] = t_loop;
      a.dbg[(size_t)wl * 4 + 2] = t_loop_end;
      a.dbg[(size_t)wl * 4 + 3] = __builtin_amdgcn_s_memtime();
   }
#endif
}

// max / min on the stream schedule.  The running sum becomes the best value so far and the WORD INDEX (position in the
// wave's stream) at which it was met; a second LDS plane keeps those indices beside the values.  "Strictly better
// wins" in stream order: the plan walks the edges of a row slice by slice and, inside a slice, in CSR order, so for rows
// whose columns ascend the stream order of a row IS its CSR order and the first of equal candidates -- the lowest CSR
// position, the reference's tie rule -- is the one that stays (the plan builders refuse graphs with unsorted rows for
// this kernel).  No position travels with the gathers: the word index is arithmetic (batch, step, slot), and only the
// M x K winners are translated to CSR positions, through the plan's `perm`, when a row is written out.
//
// Round 4: the registers CARRY the row's pair.  Rounds 2-3 started every visit of a row from the identity and merged
// the visit's winners into the row's LDS pair when the stream left it (read both planes, four compare-selects, write both
// planes, reset eight registers: ~20 vector instructions per change of row on top of ~25 per step -- the ISA showed 40-45
// vector instructions per step against 13 for the sum kernel, i.e. more vector-ALU cycles per step than the 26 the
// address pipeline needs for the gather: the kernel was ALU-bound and wanted FEW changes of row, 8 slices of 7.5 MB,
// 67 % L2 hits).  Now a change of row is a SWAP: the pair in the registers is stored to the row it belongs to, the pair
// of the new row is loaded and the compare simply goes on -- two 16-byte LDS writes and two reads, no vector ALU work at
// all -- so the row's best so far meets every later candidate directly ("strictly better" against the EARLIER stream
// position keeps the tie rule), and the slices can be as small as the sum kernel's.  Padding words (column n: the gather
// returns 0, which would beat negative values) carry the local row NVMAX, a spare LDS row of the wave that is never
// written out: the plan builders emit it for max / min plans, so the loop neither tests for padding nor masks the column.
// (Also built in round 4 and dropped: 16-bit row-relative ORDINALS in place of the 32-bit word indices -- 388 bytes of LDS per
// row instead of 512, 48 rows per wave, three generations on the Reddit shape instead of four, no permutation lookup at
// write-out; bit-exact, and slower: K=64 weighted 1.89 against 1.81 ms, unit 1.63 against 1.61 -- the swap then packs and
// unpacks and moves a per-row word count as well, which costs what the saved dispatch gave.)
typedef float v2f_t __attribute__((ext_vector_type(2)));

// ARG = false (z_arg == NULL: values only, e.g. the aggregation of an inference pass, where no backward will ask which edge
// won): the index registers, their selects -- four of the 16-18 vector instructions of a step, in a loop that is bound by
// them (profiles/r04_experiments.txt) --, the index plane and the permutation lookups at write-out all drop out; same plan,
// same values bit for bit.  K=64 on the Reddit shape: 1.80 -> 1.6 ms with U(0,1) weights, 1.61 -> 1.4 with unit weights.
template <int OP, int LPR, bool HAS_VAL, int NVMAX, int NBW, int WGS, bool ARG>
__global__ __launch_bounds__(256, (stream_wgs_per_cu<LPR, 2 * (NVMAX + 1), WGS>())) void spmm_stream_minmax_kernel(const SweepArgs a) {
   constexpr int WAVES = 4, G = 64 / LPR, PANEL = LPR * 4, U = 64 * NBW / G;
   constexpr int PER = NVMAX / G;
   constexpr int ROWS = NVMAX + 1;                        // + the spare row of the padding words
   constexpr int WAVE_DWORDS = 2 * ROWS * PANEL;          // a wave's block: the values, then their indices
   constexpr int IDX0 = ROWS * PANEL;
   constexpr int NONE = INT_MAX;                          // "no winner yet"
   static_assert(ROWS <= 256 && NVMAX % G == 0, "the local row (and the spare row) is the top byte of a word");
   __shared__ __attribute__((aligned(16))) float s_all[WAVES * WAVE_DWORDS];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   const int wl = (int)blockIdx.x * WAVES + wave;
   if (wl >= a.wave_count) return;                       // no barrier anywhere below
   const int64_t w = (int64_t)a.wave_base + wl;
   float *my = s_all + wave * WAVE_DWORDS;
   for (int i = lane * 4; i < ROWS * PANEL; i += 256)
      *reinterpret_cast<float4 *>(my + i) = make_float4(identity<OP>(), identity<OP>(), identity<OP>(), identity<OP>());
   if (ARG)
      for (int i = lane * 4; i < ROWS * PANEL; i += 256)
         *reinterpret_cast<int4 *>(my + IDX0 + i) = make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX);
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);
   const bool cok = lc * 4 < a.k;
   int ccol = lc * 4, vfirst = 0;
   if (cok && ccol + 4 > (int)a.k) { vfirst = ccol + 4 - (int)a.k; ccol = (int)a.k - 4; }
   // lanes beyond column k gather nothing: their column term is 2^31, past the end of the descriptor (the entry admits dense
   // operands under 2 GiB only, so row offset + 2^31 neither stays inside the descriptor nor wraps) -- no OR in the loop
   const unsigned cbyte = cok ? (unsigned)ccol * 4u : 0x80000000u;
   float *lane_base = my + lc * 4;                        // a lane's four values of a row ...
   int *lane_idx = reinterpret_cast<int *>(my + IDX0) + lc * 4;      // ... and their indices
   const int64_t s0 = a.wave_step_off[w], s1 = a.wave_step_off[w + 1];
   const int64_t nwords = (s1 - s0) * G;
   const int32_t *wp = a.words + s0 * G;
   const float *vp = HAS_VAL ? a.vals + s0 * G : nullptr;
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   const unsigned pad_word = ((unsigned)NVMAX << 24) | a.null_word;      // past the end of the wave: the spare row too
   auto load_words = [&](int64_t first, unsigned (&word)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         word[q] = i < nwords ? (unsigned)wp[i] : pad_word;
      }
   };
   auto load_vals = [&](int64_t first, float (&val)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         val[q] = HAS_VAL && i < nwords ? vp[i] : 0.0f;
      }
   };
   const int g4 = g * 4;                                  // byte address of lane g for ds_bpermute
   unsigned w1[NBW], w2[NBW];
   float v0[NBW], v1[NBW];                               // the weights of the batch being consumed and of the next: a weight is
                                                         // needed one batch later than its word, so it is loaded one batch later
   v4i_t t[U];
   unsigned la[U];
   // (a weight is fetched from its batch register when its gather is consumed, one step ahead -- a ring of U weights
   // beside the U gathers in flight does not fit the register file).  The 24-bit multiply reads the column straight out
   // of the word: its top byte, the local row, is outside the bits the instruction looks at.
   // the slot's word of step u of a batch register set: lane (u * G) % 64 + g of register (u * G) / 64
   auto fetch = [&](int u, const unsigned (&word_l)[NBW]) {
      return (unsigned)__builtin_amdgcn_ds_bpermute(g4 + (int)(((u * G) % 64) * 4), (int)word_l[(u * G) / 64]);
   };
   // la[u] keeps the WORD of step u as it is: its top byte, the local row, is compared byte against byte when the gather is
   // consumed (one SDWA compare) and only becomes an LDS offset inside the rare change of row -- no shift per step
   auto issue = [&](int u, unsigned word) {
      const unsigned o = __umul24(word, ldyb) + cbyte;
      la[u] = word;
      t[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
   };
   load_words(0, w1);
   load_vals(0, v0);
#pragma unroll
   for (int u = 0; u < U; u++) issue(u, fetch(u, w1));
   load_words(64 * NBW, w1);
   load_vals(64 * NBW, v1);
   load_words(128 * NBW, w2);
   unsigned cur = (unsigned)(g * PER * PANEL);            // LDS offset of the row the registers hold ...
   unsigned curw = (unsigned)(g * PER) << 24;             // ... and that row as the top byte of a word
   float acc[4];
   int bi[4];
#pragma unroll
   for (int v = 0; v < 4; v++) { acc[v] = identity<OP>(); bi[v] = NONE; }
   // change of row: the pair held goes to the row it belongs to, the new row's pair comes in.  The slots of a wave own
   // disjoint rows (the spare row is written by all and read back by nobody who cares) and a wave's LDS operations execute
   // in order, so a later visit of a row reads what the last one stored.
   // (reads first: the wave then waits for the loads only -- the stores just have to be issued)
   auto swap_to = [&](unsigned nxt_word) {
      asm volatile("" : "+v"(nxt_word));                 // keeps the LDS address arithmetic inside the branch (one vector instruction per step otherwise)
      const unsigned nxt = (nxt_word >> 24) * (unsigned)PANEL;
      curw = nxt_word;
      const float4 o = *reinterpret_cast<const float4 *>(lane_base + nxt);
      if constexpr (ARG) {
         const int4 oi = *reinterpret_cast<const int4 *>(lane_idx + nxt);
         *reinterpret_cast<float4 *>(lane_base + cur) = make_float4(acc[0], acc[1], acc[2], acc[3]);
         *reinterpret_cast<int4 *>(lane_idx + cur) = make_int4(bi[0], bi[1], bi[2], bi[3]);
         bi[0] = oi.x; bi[1] = oi.y; bi[2] = oi.z; bi[3] = oi.w;
      } else {
         *reinterpret_cast<float4 *>(lane_base + cur) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
      acc[0] = o.x; acc[1] = o.y; acc[2] = o.z; acc[3] = o.w;
      cur = nxt;
   };
   const int64_t nb = (nwords + 64 * NBW - 1) / (64 * NBW);
   for (int64_t b = 0; b < nb; b++) {
      const int widx0 = (int)(b * (64 * NBW)) + g;        // word index of this lane's slot at step 0 of the batch
      float vnext = HAS_VAL ? __int_as_float(__builtin_amdgcn_ds_bpermute(g4, __float_as_int(v0[0]))) : 0.0f;
      // the word of the gather that REPLACES step u's is fetched one step ahead: the cross-lane read then has a whole step
      // (the row check and the compares) to land instead of being waited for straight away
      unsigned wnext = fetch(0, w1);
#pragma unroll
      for (int u = 0; u < U; u++) {
         const float vcur = vnext;
         const unsigned wcur = wnext;
         if (HAS_VAL && u + 1 < U)
            vnext = __int_as_float(__builtin_amdgcn_ds_bpermute(g4 + (int)((((u + 1) * G) % 64) * 4), __float_as_int(v0[((u + 1) * G) / 64])));
         if (u + 1 < U) wnext = fetch(u + 1, w1);
         // per lane: the slots of a wave change rows at different steps.  ONE vector instruction: the top bytes of the two words
         // compared in place (SDWA), the lane mask handed to the branch as it is -- written in C++ the test becomes xor + compare,
         // and with the row offset kept per step, shift + compare (round 4, second session: 18 -> 17 instructions per step)
         unsigned long long row_changes;
         asm("v_cmp_ne_u32_sdwa %0, %1, %2 src0_sel:BYTE_3 src1_sel:BYTE_3" : "=s"(row_changes) : "v"(la[u]), "v"(curw));
         if (__builtin_amdgcn_inverse_ballot_w64(row_changes)) swap_to(la[u]);
         const int mark = widx0 + u * G;                 // what a winner of this step is remembered by: its word index
         v2f_t x01 = {__int_as_float(t[u][0]), __int_as_float(t[u][1])};
         v2f_t x23 = {__int_as_float(t[u][2]), __int_as_float(t[u][3])};
         if (HAS_VAL) {                                  // two packed multiplies instead of four
            const v2f_t vv = {vcur, vcur};
            x01 *= vv;
            x23 *= vv;
         }
         const float tt[4] = {x01.x, x01.y, x23.x, x23.y};
#pragma unroll
         for (int v = 0; v < 4; v++) {
            const bool win = OP == OP_MAX ? tt[v] > acc[v] : tt[v] < acc[v];      // NaN never wins, as in the oracle
            acc[v] = win ? tt[v] : acc[v];
            if (ARG) bi[v] = win ? mark : bi[v];
         }
         issue(u, wcur);
      }
#pragma unroll
      for (int q = 0; q < NBW; q++) { w1[q] = w2[q]; v0[q] = v1[q]; }
      load_words((b + 3) * 64 * NBW, w2);
      load_vals((b + 2) * 64 * NBW, v1);
   }
   *reinterpret_cast<float4 *>(lane_base + cur) = make_float4(acc[0], acc[1], acc[2], acc[3]);
   if (ARG) *reinterpret_cast<int4 *>(lane_idx + cur) = make_int4(bi[0], bi[1], bi[2], bi[3]);
   // write-out, ALL rows of the slot at once: the winners' word indices become CSR positions through the plan's permutation --
   // four dependent loads per row and lane, which a row-at-a-time loop waits for PER times over (the gather registers are free
   // by now: every load of the slot's PER rows is in flight before the first one is needed)
   const int32_t *ids = a.ids + s0 * G;
   int row_[PER], part_[PER], best_[PER][4];
   float val_[PER][4];
#pragma unroll
   for (int jj = 0; jj < PER; jj++) {
      const int lrow = g * PER + jj;
      row_[jj] = cok ? a.wave_row[(size_t)w * NVMAX + lrow] : -1;
      part_[jj] = a.wave_part[(size_t)w * NVMAX + lrow];
      const float4 t4 = *reinterpret_cast<const float4 *>(lane_base + lrow * PANEL);
      val_[jj][0] = t4.x; val_[jj][1] = t4.y; val_[jj][2] = t4.z; val_[jj][3] = t4.w;
      if constexpr (ARG) {
         const int4 i4 = *reinterpret_cast<const int4 *>(lane_idx + lrow * PANEL);
         best_[jj][0] = i4.x; best_[jj][1] = i4.y; best_[jj][2] = i4.z; best_[jj][3] = i4.w;
      } else {
         best_[jj][0] = best_[jj][1] = best_[jj][2] = best_[jj][3] = INT_MAX;
      }
   }
   if constexpr (ARG) {
#pragma unroll
      for (int jj = 0; jj < PER; jj++)
#pragma unroll
         for (int i = 0; i < 4; i++)   // (a word index outside the wave's stream -- INT_MAX = no winner -- never reaches the subscript)
            best_[jj][i] = (row_[jj] >= 0 && (unsigned)best_[jj][i] < (unsigned)nwords) ? ids[best_[jj][i]] : INT_MAX;
   }
#pragma unroll
   for (int jj = 0; jj < PER; jj++) {
      const int row = row_[jj];
      if (row < 0) continue;
      const int c = ccol;
      if (part_[jj] >= 0) {
         const size_t po = (size_t)part_[jj] * (size_t)a.k + c;
         store_tail<4>(a.part_val + po, val_[jj], vfirst);
         if constexpr (ARG) {
#pragma unroll
            for (int i = 0; i < 4; i++) if (i >= vfirst) a.part_idx[po + i] = best_[jj][i];
         }
         continue;
      }
      int64_t arg[4];
      finish_row<OP>(a, row, c, val_[jj], best_[jj], arg);
      store_tail<4>(a.z + (size_t)row * (size_t)a.ldz + c, val_[jj], vfirst);
      if (a.z_arg) {
         int64_t *ar = a.z_arg + (size_t)row * (size_t)a.ldz + c;
#pragma unroll
         for (int i = 0; i < 4; i++) if (i >= vfirst) ar[i] = arg[i];
      }
   }
}


#ifdef ISPLIB_EXP_WAVE_TIMES
static unsigned long long *g_dbg = nullptr;      // set by isplib_debug_wave_times; one slab per launch, four launches
static int g_dbg_launch = 0;
#endif
template <int LPR, bool HAS_VAL>
static int launch_stream(const SweepArgs &a_in, hipStream_t st) {
   SweepArgs a = a_in;
#ifdef ISPLIB_EXP_WAVE_TIMES
   a.dbg = g_dbg ? g_dbg + (size_t)(g_dbg_launch++ % 4) * (size_t)a.wave_count * 4 : nullptr;
#endif
   const unsigned blocks = (unsigned)((a.wave_count + 3) / 4);
   if (blocks == 0) return ISPLIB_SUCCESS;
   if constexpr (LPR == 32) hipLaunchKernelGGL((spmm_stream_kernel<32, HAS_VAL, 32, 1, 2>), dim3(blocks), dim3(256), 0, st, a);
   else if constexpr (LPR == 16) hipLaunchKernelGGL((spmm_stream_kernel<16, HAS_VAL, ISPLIB_STREAM_NV4, ISPLIB_STREAM_NBW4, ISPLIB_STREAM_WGS4>), dim3(blocks), dim3(256), 0, st, a);
   else hipLaunchKernelGGL((spmm_stream_kernel<8, HAS_VAL, ISPLIB_STREAM_NV8, ISPLIB_STREAM_NBW8, ISPLIB_STREAM_WGS8>), dim3(blocks), dim3(256), 0, st, a);
   return check_launch("spmm_stream_kernel");
}

template <int OP, bool HAS_VAL>
static int launch_stream_minmax(const SweepArgs &a, hipStream_t st, int streams) {
   const unsigned blocks = (unsigned)((a.wave_count + 3) / 4);
   if (blocks == 0) return ISPLIB_SUCCESS;
   if (a.z_arg) {
      if (streams == 8) hipLaunchKernelGGL((spmm_stream_minmax_kernel<OP, 8, HAS_VAL, ISPLIB_STREAM_MM8_NV, ISPLIB_STREAM_MM8_NBW, ISPLIB_STREAM_MM8_WGS, true>), dim3(blocks), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((spmm_stream_minmax_kernel<OP, 16, HAS_VAL, ISPLIB_STREAM_MM_NV, ISPLIB_STREAM_MM_NBW, ISPLIB_STREAM_MM_WGS, true>), dim3(blocks), dim3(256), 0, st, a);
   } else {                                               // values only
      if (streams == 8) hipLaunchKernelGGL((spmm_stream_minmax_kernel<OP, 8, HAS_VAL, ISPLIB_STREAM_MM8_NV, ISPLIB_STREAM_MM8_NBW, ISPLIB_STREAM_MM8_WGS, false>), dim3(blocks), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((spmm_stream_minmax_kernel<OP, 16, HAS_VAL, ISPLIB_STREAM_MM_NV, ISPLIB_STREAM_MM_NBW, ISPLIB_STREAM_MM_WGS, false>), dim3(blocks), dim3(256), 0, st, a);
   }
   return check_launch("spmm_stream_minmax_kernel");
}

}  // namespace isplib

using namespace isplib;


// ---- stream form: entry ---------------------------------------------------------------------------------------------
extern "C" int isplib_spmm_stream_geometry(int streams, int *rows_per_wave, int *waves_resident) {
   clear_error();
   if (streams != 2 && streams != 4 && streams != 8) return fail(ISPLIB_FAIL, "isplib_spmm_stream_geometry: streams must be 2, 4 or 8");
   if (rows_per_wave) *rows_per_wave = stream_geom(streams).nvmax;
   if (waves_resident) *waves_resident = stream_resident_waves(streams, device_cus());
   return ISPLIB_SUCCESS;
}

extern "C" int isplib_spmm_stream_minmax_geometry(int streams, int *rows_per_wave, int *waves_resident) {
   clear_error();
   if (streams != 4 && streams != 8) return fail(ISPLIB_FAIL, "isplib_spmm_stream_minmax_geometry: streams must be 4 (64-column slots) or 8 (32-column slots)");
   if (rows_per_wave) *rows_per_wave = stream_geom(streams, true).nvmax;
   if (waves_resident) *waves_resident = stream_resident_waves(streams, device_cus(), true);
   return ISPLIB_SUCCESS;
}

static int suggest_stream_geom(int64_t m, int64_t n, int64_t nnz, int st, int rpw, int resident, double slice_bytes, double chunk_div,
                               int *slices, int *chunk) {
   const int64_t per_gen = (int64_t)rpw * resident;
   const int64_t gens = (m + per_gen - 1) / per_gen;
   if ((double)nnz / (double)gens / 8.0 < 3.0 * (double)n) return 0;
   const double panel_bytes = 1024.0 / st;
   int sl = (int)((double)n * panel_bytes / slice_bytes + 0.5);
   sl = sl < 1 ? 1 : (sl > 512 ? 512 : sl);
   int64_t ch = (int64_t)((double)nnz / ((double)gens * resident * st) / chunk_div);
   ch = ch < 256 ? 256 : (ch > (1 << 20) ? (1 << 20) : ch);
   if (slices) *slices = sl;
   if (chunk) *chunk = (int)ch;
   return 1;
}
extern "C" int isplib_suggest_stream_weighted(int64_t m, int64_t n, int64_t nnz, int64_t k, int weighted, int *streams, int *slices, int *chunk) {
   // When does the stream schedule pay, and with which plan?  Measured on MI355X:
   //   * slots of 8 lanes (32-column panels) up to k = 32, of 16 lanes (64-column panels) up to 64 and from 128 on, of
   //     32 lanes (one 128-column pass) in between -- 64 + 36 columns in two passes cost K=100 3.23 ms, one pass 2.63
   //     (task list 3.17); K=72: 2.92 / 2.30 (2.42); K=96: 2.51 / 2.30 (2.56); K=160 wants 64 + 64 + 32: 3.85 / 4.97 (4.15);
   //   * WEIGHTED graphs at whole multiples of 128 columns: 128-column slots as well -- a pass reads the plan's weight stream
   //     (4 bytes per edge, as large as the word stream) once per panel, and half as many panels halve that: round 4, same
   //     box, alternating: K=128 2.90 -> 2.82 ms (63 slices; 72: 2.87), K=256 5.89 -> 5.74; with unit weights the two
   //     geometries tie (2.69-2.72 against 2.69-2.70), so those stay on 64-column slots;
   //   * a column slice of ~1.9 MB of the panel (Reddit shape: 32 slices at 64 columns, 16 at 32);
   //   * every generation of waves sweeps the whole dense operand once per XCD, so the rows a generation holds must
   //     reuse each row of it often: edges per generation and XCD >= 3 x rows of y (Reddit shape: 31; one rank's shard
   //     of it at 8 / 16 / 32 ranks: 7.7 / 3.9 / 1.9 -- stream 0.37 / 0.22-0.24 / 0.16-0.18 ms against 0.43 / 0.24 /
   //     0.16-0.19 ms on the task list, scripts/exp_shard_schedule.py; the ogbn-products shape, mean degree 50 over 2.4 M
   //     rows: 0.3 -- such graphs stay on the plain kernel);
   //   * rows longer than ~0.3 of a stream's share of the edges are dealt to several virtual rows (chunk).
   clear_error();
   if (m <= 0 || n <= 0 || nnz <= 0 || k < 4 || n >= (1LL << 24) || nnz < (1LL << 22)) return 0;
   if (!stream_domain_ok(n, k, nnz)) return 0;         // what the entry and the plan builder would refuse: not offered
   int st = k <= 32 ? 8 : (k <= 64 ? 4 : (k < 128 ? 2 : 4));
   if (weighted && k >= 128 && (k % 128) == 0) st = 2;
   int rpw = 0, resident = 0;
   if (isplib_spmm_stream_geometry(st, &rpw, &resident) != ISPLIB_SUCCESS || rpw <= 0 || resident <= 0) return 0;
   if (!suggest_stream_geom(m, n, nnz, st, rpw, resident, 1.9e6, 3.4, slices, chunk)) {
      if (st != 2 || k < 128) return 0;
      st = 4;                                          // (fewer rows per wave on the wide slots: the reuse rule may still accept the narrow ones)
      if (isplib_spmm_stream_geometry(st, &rpw, &resident) != ISPLIB_SUCCESS || rpw <= 0 || resident <= 0) return 0;
      if (!suggest_stream_geom(m, n, nnz, st, rpw, resident, 1.9e6, 3.4, slices, chunk)) return 0;
   }
   if (streams) *streams = st;
   return 1;
}

extern "C" int isplib_suggest_stream(int64_t m, int64_t n, int64_t nnz, int64_t k, int *streams, int *slices, int *chunk) {
   return isplib_suggest_stream_weighted(m, n, nnz, k, 0, streams, slices, chunk);      // unit weights
}

extern "C" int isplib_suggest_stream_minmax(int64_t m, int64_t n, int64_t nnz, int64_t k, int *streams, int *slices, int *chunk) {
   // max / min: 32-column slots up to k = 32, 64-column slots above; the same reuse rule; rows must be column-sorted.
   // Round 4 (the row's pair rides in registers, a change of row is an LDS swap without vector-ALU work), Reddit shape,
   // K = 64, U(0,1) weights / unit weights, ms: 8 slices 1.97 / 1.73, 12: 1.85 / 1.68, 16: 1.81 / 1.63, 20: 1.82 / 1.61,
   // 24: 1.84 / 1.63, 31: 1.92 / 1.73, 48: 2.01 / 1.82 -- slices of ~3.3 MB of the panel (rounds 2-3, when every change of
   // row cost a read-compare-write of both planes: 7.5 MB, 2.04 / 1.82); K = 32 on 32-column slots: 8-16 slices 0.87-0.89,
   // 24: 0.94.  Rows cut at ~0.85 of a stream's share on 64-column slots (chunk 3000: 1.81, 2057: 1.84, 1028: 1.91), ~0.6
   // on 32-column ones (2057: 0.873, 3000: 0.890).  Task list: 2.35 / 1.96 (K=128 weighted 3.71 against 4.54).
   clear_error();
   if (m <= 0 || n <= 0 || nnz <= 0 || k < 4 || n >= (1LL << 24) || nnz < (1LL << 22) || nnz >= (1LL << 31)) return 0;
   if (!stream_domain_ok(n, k, nnz) || (unsigned long long)n * (unsigned long long)k * 4ull >= (1ull << 31)) return 0;   // the max / min entry: under 2 GiB
   const int st = k <= 32 ? 8 : 4;
   int rpw = 0, resident = 0;
   if (isplib_spmm_stream_minmax_geometry(st, &rpw, &resident) != ISPLIB_SUCCESS || rpw <= 0 || resident <= 0) return 0;
   if (!suggest_stream_geom(m, n, nnz, st, rpw, resident, 3.3e6, st == 8 ? 1.7 : 1.17, slices, chunk)) return 0;
   if (streams) *streams = st;
   return 1;
}

extern "C" size_t isplib_spmm_stream_workspace_bytes(const isplib_stream_plan *plan) {
   if (!plan || plan->n_parts <= 0) return 256;
   const size_t pk = (size_t)(256 / (plan->streams > 0 ? plan->streams : 4));
   return ((size_t)plan->n_parts * pk * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" size_t isplib_spmm_stream_minmax_workspace_bytes(const isplib_stream_plan *plan) {
   return 2 * isplib_spmm_stream_workspace_bytes(plan);   // values, then CSR positions
}

static int stream_run(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const int64_t *pntrb, const int64_t *pntre,
                      const isplib_stream_plan *plan, const float *y, int64_t ldy, float *z, int64_t ldz, int64_t *z_arg,
                      void *workspace, size_t workspace_bytes, const isplib_epilogue *ep, void *stream) {
   clear_error();
   const bool mm = imessage == ISPLIB_MSG_SPMM_MAX || imessage == ISPLIB_MSG_SPMM_MIN;
   if (imessage != ISPLIB_MSG_SPMM_SUM && imessage != ISPLIB_MSG_SPMM_MEAN && !mm)
      return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_stream_hip: message outside the SpMM set");
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: negative dimension");
   if (m == 0 || k == 0) return ISPLIB_SUCCESS;
   if (!plan) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: plan is required");
   if (plan->rows != m || plan->cols != n) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: the plan was built for another shape");
   if (n >= (1LL << 24) || ldy >= (1LL << 22)) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: n must be < 2^24 and ldy < 2^22 (24-bit address arithmetic)");
   if (plan->streams != 2 && plan->streams != 4 && plan->streams != 8)
      return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: bad plan geometry (streams 2, 4 or 8)");
   if (mm && plan->streams != 4 && plan->streams != 8) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: max / min run on 4- or 8-stream plans (isplib_spmm_stream_minmax_geometry)");
   if (plan->gens < 1 || plan->waves_per_gen < 1 || plan->rows_per_wave != stream_geom(plan->streams, mm).nvmax)
      return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: bad plan geometry (rows_per_wave must be what isplib_spmm_stream_geometry / _minmax_geometry reports)");
   if (mm && plan->n_steps > 0 && !plan->perm) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: max / min need the plan's perm array (the winners' CSR positions)");
   if (mm && nnz >= (1LL << 31)) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: max / min need nnz < 2^31");
   if (mm && ep) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: the epilogue is defined for sum / mean only");
   if (mm && (unsigned long long)n * (unsigned long long)ldy * 4ull >= (1ull << 31))
      return fail(ISPLIB_FAIL, "fusedMM_csr_stream_minmax_hip: dense operand of 2 GiB or more (use fusedMM_csr_tasks_hip)");
   if (k < 4) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: k >= 4 required (use fusedMM_csr_hip)");
   if (ldy < k || ldz < k) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: leading dimension smaller than k");
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > BUF_LIMIT) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: dense operand larger than 3.5 GiB (use fusedMM_csr_hip)");
   if (!pntrb || !pntre || !z || !y || !plan->wave_row || !plan->wave_part || !plan->wave_step_off ||
       (plan->n_steps > 0 && !plan->words) || (plan->n_hub > 0 && (!plan->hub_row || !plan->hub_off)))
      return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: null operand");
   if (plan->n_parts > 0) {
      if (!workspace || workspace_bytes < (mm ? 2 : 1) * isplib_spmm_stream_workspace_bytes(plan)) return fail(ISPLIB_NOT_ENOUGH_MEM, "fusedMM_csr_stream_hip: workspace too small");
      if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: workspace must be 256-byte aligned");
   }
   SweepArgs a = {};
   a.empty_init = empty_row_init();
   a.k = k; a.nnz = nnz; a.pntrb = pntrb; a.pntre = pntre;
   a.y = y; a.ldy = ldy; a.ybytes = (unsigned)yb; a.z = z; a.ldz = ldz; a.z_arg = z_arg;
   a.mean = imessage == ISPLIB_MSG_SPMM_MEAN ? 1 : 0;
   a.ids = plan->perm; a.abs_ids = 1;
   a.wave_row = plan->wave_row; a.wave_part = plan->wave_part;
   a.words = plan->words; a.vals = plan->vals; a.wave_step_off = plan->wave_step_off; a.null_word = (unsigned)n;
   a.hub_row = plan->hub_row; a.hub_off = plan->hub_off; a.n_hub = plan->n_hub;
   a.part_val = (float *)workspace;
   a.part_idx = mm && workspace ? (int *)((char *)workspace + isplib_spmm_stream_workspace_bytes(plan)) : nullptr;
   if (ep) {
      if (ep->self && ep->ld_self < k) return fail(ISPLIB_FAIL, "fusedMM_csr_stream_hip: ld_self smaller than k");
      a.ep_row_scale = ep->row_scale; a.ep_self = ep->self; a.ep_ld_self = ep->ld_self; a.ep_bias = ep->bias;
      a.ep_relu = ep->relu ? 1 : 0;
   }
   hipStream_t st = (hipStream_t)stream;
   const int64_t pw = 256 / plan->streams;                // panel width follows the plan: a slot is 64 / streams lanes x 4 floats
   for (int64_t c0 = 0; c0 < k; c0 += pw) {
      SweepArgs p = a;
      p.k = (k - c0) < pw ? (k - c0) : pw;
      if (p.k < 4) {                              // a sliver of 1-3 columns: widen it backwards (the overlap is rewritten identically)
         p.k = 4;
         c0 = k - 4;
      }
      p.y = y + c0;
      p.z = z + c0;
      p.ep_self = a.ep_self ? a.ep_self + c0 : nullptr;
      p.ep_bias = a.ep_bias ? a.ep_bias + c0 : nullptr;
      p.z_arg = z_arg ? z_arg + c0 : nullptr;
      p.ybytes = (unsigned)(yb - (unsigned long long)c0 * 4ull);
      for (int gen = 0; gen < plan->gens; gen++) {          // one dispatch per generation (all in one launch: measured slower, 2.73 against 2.68 ms)
         p.wave_base = gen * plan->waves_per_gen;
         p.wave_count = plan->waves_per_gen;
         int rc;
         if (imessage == ISPLIB_MSG_SPMM_MAX) rc = plan->vals ? launch_stream_minmax<OP_MAX, true>(p, st, plan->streams) : launch_stream_minmax<OP_MAX, false>(p, st, plan->streams);
         else if (imessage == ISPLIB_MSG_SPMM_MIN) rc = plan->vals ? launch_stream_minmax<OP_MIN, true>(p, st, plan->streams) : launch_stream_minmax<OP_MIN, false>(p, st, plan->streams);
         else if (plan->streams == 2) rc = plan->vals ? launch_stream<32, true>(p, st) : launch_stream<32, false>(p, st);
         else if (plan->streams == 4) rc = plan->vals ? launch_stream<16, true>(p, st) : launch_stream<16, false>(p, st);
         else rc = plan->vals ? launch_stream<8, true>(p, st) : launch_stream<8, false>(p, st);
         if (rc) return rc;
      }
      if (plan->n_hub > 0) {
         const bool v4 = (p.k % 4) == 0 && (p.ldz % 4) == 0 && ((uintptr_t)p.z & 15) == 0 && (!p.ep_self || ((p.ep_ld_self % 4) == 0 && ((uintptr_t)p.ep_self & 15) == 0));
         int64_t blocks = (plan->n_hub * (v4 ? p.k / 4 : p.k) + 255) / 256;
         if (blocks > 4096) blocks = 4096;
         if (imessage == ISPLIB_MSG_SPMM_MAX) {
            if (v4) hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_MAX, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_MAX, 1>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         } else if (imessage == ISPLIB_MSG_SPMM_MIN) {
            if (v4) hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_MIN, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_MIN, 1>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         } else if (v4) hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_ADD, 4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         else hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_ADD, 1>), dim3((unsigned)blocks), dim3(256), 0, st, p);
         const int rc = check_launch("sweep_hub_fold_kernel");
         if (rc) return rc;
      }
   }
   return ISPLIB_SUCCESS;
}

extern "C" int fusedMM_csr_stream_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                                      const int64_t *pntrb, const int64_t *pntre, const isplib_stream_plan *plan,
                                      const float *y, int64_t ldy, float *z, int64_t ldz, void *workspace,
                                      size_t workspace_bytes, const isplib_epilogue *ep, void *stream) {
   if (imessage != ISPLIB_MSG_SPMM_SUM && imessage != ISPLIB_MSG_SPMM_MEAN) {
      clear_error();
      return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_stream_hip: sum and mean only (max / min: fusedMM_csr_stream_minmax_hip)");
   }
   return stream_run(imessage, m, n, k, nnz, pntrb, pntre, plan, y, ldy, z, ldz, nullptr, workspace, workspace_bytes, ep, stream);
}

extern "C" int fusedMM_csr_stream_minmax_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz,
                                             const int64_t *pntrb, const int64_t *pntre, const isplib_stream_plan *plan,
                                             const float *y, int64_t ldy, float *z, int64_t ldz, int64_t *z_arg,
                                             void *workspace, size_t workspace_bytes, void *stream) {
   if (imessage != ISPLIB_MSG_SPMM_MAX && imessage != ISPLIB_MSG_SPMM_MIN) {
      clear_error();
      return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_stream_minmax_hip: max and min only");
   }
   return stream_run(imessage, m, n, k, nnz, pntrb, pntre, plan, y, ldy, z, ldz, z_arg, workspace, workspace_bytes, nullptr, stream);
}

#ifdef ISPLIB_EXP_WAVE_TIMES
// experiment builds only (scripts/exp_wave_times.py; not declared in include/isplib_hip.h)
extern "C" void isplib_debug_wave_times(unsigned long long *buf) { isplib::g_dbg = buf; isplib::g_dbg_launch = 0; }
#endif
