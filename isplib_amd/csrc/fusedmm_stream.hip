// fusedmm_stream.hip -- the SDDMM-fused FusedMM words on the stream front end (fusedMM_csr_udef_stream_hip).
//
// The generic pipeline z_i = AOP_j VSC(SOP(ROP(VOP(x_i, y_j)))) (csrc/fusedMM.h:18-74) gathers the same rows of y over the same
// edges as the SpMM; on the task list (fusedmm_general.hip) its two hot shapes -- the sigmoid / attention family
// COPY_RHS|DOT|UDEF|MUL|ADD and the t-distribution family SUBR|NORMR|UDEF|MUL|ADD of the FusedMM paper's graph-embedding
// kernels -- took 6.3 ms at K=128 on the Reddit shape where the SpMM that gathers the same rows takes 2.7.  Here they run
// on the stream schedule's front end (spmm_sweep.hip): word streams, one full 1-KiB gather per step, 32 gathers in flight
// per wave, two waves per SIMD, rows resident in LDS.  What differs from the SpMM:
//   * the reduce stage needs the WHOLE row of y before anything can be accumulated, so a slot spans the full width (k <= 128:
//     32-lane slots, two rows per gather; k <= 64: 16-lane; k <= 32: 8-lane) and there are no column panels;
//   * two LDS planes per row: x_i (read when the slot's stream turns to the row) and the accumulator z_i -- the max / min
//     kernel's LDS budget, so plans have its shape: half the rows per wave of a sum plan plus the spare row that padding
//     words point at (x = 0, gathered y = 0: whatever a padding step computes lands in a row nobody writes out);
//   * four steps at a time: their four partial dot products (or squared distances) are summed over the slot's lanes by ONE
//     transposed butterfly (gather.h), the scalar stage runs once on the lanes that end up owning a sum (reciprocals by v_rcp_f32),
//     four cross-lane reads
//     hand every lane its step's scalar, and only then are the four gathered rows scaled into the accumulator and their
//     gathers re-issued (28-32 in flight instead of 32).
// Every row is accumulated by the one wave that owns it, in stream order: no atomics, bitwise reproducible.  Hub rows cut
// into virtual rows leave partial rows that sweep_hub_fold_kernel adds up, as for the SpMM.
#include "sweep_common.h"

namespace isplib {

// (reciprocals by v_rcp_f32, 1 ulp: an IEEE division is ten vector instructions in a loop that is bound by them -- the ISA of the
// first form of this kernel had 39 per step against the SpMM's 13 -- and the results are held to 1e-4 of the largest |z| anyway)
__device__ __forceinline__ float sop_menu(int kind, float s, float p) {
   switch (kind) {
      case ISPLIB_SOP_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __expf(-s));
      case ISPLIB_SOP_ONE_MINUS_SIGMOID: return 1.0f - __builtin_amdgcn_rcpf(1.0f + __expf(-s));
      case ISPLIB_SOP_TDIST: return __builtin_amdgcn_rcpf(1.0f + s);
      case ISPLIB_SOP_SCALE: return p * s;
      case ISPLIB_SOP_EXP: return __expf(s);
      case ISPLIB_SOP_LEAKY_EXP: return __expf(s > 0.0f ? s : p * s);
      default: return s;
   }
}

// geometry per slot width: rows per wave (without the spare row), 64-word batch registers, workgroups per CU
struct GenStreamGeom { int nvmax, nbw, wgs; };
static inline GenStreamGeom gen_stream_geom(int streams) {
   if (streams == 2) return {16, 1, 2};     // 128-column slots: 2 rows per gather, 32 gathers in flight
   if (streams == 4) return {32, 2, 2};     // 64-column slots
   return {64, 4, 2};                       // 32-column slots
}

// PAT 1: T = y_j, s = f(<x_i, y_j>);   PAT 2: T = y_j - x_i, s = f(|T|^2);   z_i += s * T
template <int PAT, int LPR, int NVMAX, int NBW, int WGS>
__global__ __launch_bounds__(256, (stream_wgs_per_cu<LPR, 2 * (NVMAX + 1), WGS>())) void fusedmm_stream_kernel(const SweepArgs a, const int sop_udef,
                                                                                                             const float sop_param) {
   constexpr int WAVES = 4, G = 64 / LPR, PANEL = LPR * 4, U = 64 * NBW / G;
   constexpr int PER = NVMAX / G, ROWS = NVMAX + 1, WAVE_FLOATS = 2 * ROWS * PANEL, Z0 = ROWS * PANEL;
   static_assert(U % 4 == 0 && ROWS <= 256 && NVMAX % G == 0, "four steps per butterfly; the local row is the top byte of a word");
   __shared__ __attribute__((aligned(16))) float s_all[WAVES * WAVE_FLOATS];
   const int lane = threadIdx.x & 63;
   const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane / LPR, lc = lane % LPR;
   const int wl = (int)blockIdx.x * WAVES + wave;
   if (wl >= a.wave_count) return;                       // no barrier anywhere below
   const int64_t w = (int64_t)a.wave_base + wl;
   float *my = s_all + wave * WAVE_FLOATS;
   const bool cok = lc * 4 < a.k;                         // k % 4 == 0 (entry): a lane's four columns are all inside or all outside
   const int ccol = lc * 4;
   const unsigned cbyte = (unsigned)ccol * 4u, poison = cok ? 0u : BUF_OOB;
   float *lane_x = my + lc * 4;                           // a lane's four columns of a row of x ...
   float *lane_z = my + Z0 + lc * 4;                      // ... and of its accumulator
   // the wave's rows of x into the first plane (unused local rows and the spare row: 0), zeros into the second
#pragma unroll 1
   for (int jj = 0; jj <= PER; jj++) {
      const int lrow = jj < PER ? g * PER + jj : NVMAX;
      const int row = jj < PER ? a.wave_row[(size_t)w * NVMAX + lrow] : -1;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (row >= 0 && cok) v = *reinterpret_cast<const float4 *>(a.g + (size_t)row * (size_t)a.ldg + ccol);
      *reinterpret_cast<float4 *>(lane_x + lrow * PANEL) = v;
      *reinterpret_cast<float4 *>(lane_z + lrow * PANEL) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
   }
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.y), 0, (int)a.ybytes, 0x00020000);
   const int64_t s0 = a.wave_step_off[w], s1 = a.wave_step_off[w + 1];
   const int64_t nwords = (s1 - s0) * G;
   const int32_t *wp = a.words + s0 * G;
   const unsigned ldyb = (unsigned)a.ldy * 4u;
   const unsigned pad_word = ((unsigned)NVMAX << 24) | a.null_word;      // past the end of the wave: the spare row
   auto load_words = [&](int64_t first, unsigned (&word)[NBW]) {
#pragma unroll
      for (int q = 0; q < NBW; q++) {
         const int64_t i = first + q * 64 + lane;
         word[q] = i < nwords ? (unsigned)wp[i] : pad_word;
      }
   };
   unsigned w1[NBW], w2[NBW];
   v4i_t t[U];
   unsigned la[U];
   auto issue = [&](int u, const unsigned (&word_l)[NBW]) {
      const unsigned word = (unsigned)__shfl((int)word_l[(u * G) / 64], (u * G) % 64 + g);
      const unsigned o = (__umul24(word & 0xFFFFFFu, ldyb) + cbyte) | poison;
      la[u] = (word >> 24) * (unsigned)PANEL;
      t[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o, 0, 0);
   };
   load_words(0, w1);
#pragma unroll
   for (int u = 0; u < U; u++) issue(u, w1);
   load_words(64 * NBW, w1);
   load_words(128 * NBW, w2);
   unsigned curz = (unsigned)(g * PER * PANEL);           // the row whose accumulator the registers hold
   float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
   auto flush = [&]() {
      float4 *p = reinterpret_cast<float4 *>(lane_z + curz);
      float4 o = *p;
      o.x += acc[0]; o.y += acc[1]; o.z += acc[2]; o.w += acc[3];
      *p = o;
   };
   const int64_t nb = (nwords + 64 * NBW - 1) / (64 * NBW);
   for (int64_t b = 0; b < nb; b++) {
#pragma unroll
      for (int u0 = 0; u0 < U; u0 += 4) {
         float d[4];
         // x_i of each of the four steps straight from its LDS row (one ds_read_b128 per step, issued together: the LDS pipe is
         // otherwise idle).  Keeping the row in registers and re-reading it when the stream turns to another row -- what the
         // accumulator does below -- cost a compare, a branch and four register copies per step here; the read costs none.
         float4 xq[4];
#pragma unroll
         for (int q = 0; q < 4; q++) xq[q] = *reinterpret_cast<const float4 *>(lane_x + la[u0 + q]);
#pragma unroll
         for (int q = 0; q < 4; q++) {
            const int u = u0 + q;
            const float4 xv = xq[q];
            float y0 = __int_as_float(t[u][0]), y1 = __int_as_float(t[u][1]), y2 = __int_as_float(t[u][2]), y3 = __int_as_float(t[u][3]);
            if (PAT == 2) {                               // T = y - x replaces y in the registers of the gather
               y0 -= xv.x; y1 -= xv.y; y2 -= xv.z; y3 -= xv.w;
               t[u][0] = __float_as_int(y0); t[u][1] = __float_as_int(y1); t[u][2] = __float_as_int(y2); t[u][3] = __float_as_int(y3);
               d[q] = fmaf(y0, y0, fmaf(y1, y1, fmaf(y2, y2, y3 * y3)));
            } else {
               d[q] = fmaf(y0, xv.x, fmaf(y1, xv.y, fmaf(y2, xv.z, y3 * xv.w)));
            }
         }
         int mine;
         const float sum = reduce_transposed<4, LPR>(d, lc, mine);
         const float sown = sop_menu(sop_udef, sum, sop_param);      // meaningful on the lanes that own a step's sum
         float s[4];
#pragma unroll
         for (int q = 0; q < 4; q++) s[q] = __shfl(sown, g * LPR + transposed_owner<4, LPR>(q));
#pragma unroll
         for (int q = 0; q < 4; q++) {
            const int u = u0 + q;
            if (la[u] != curz) {
               flush();
               curz = la[u];
               acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
            }
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v] = fmaf(s[q], __int_as_float(t[u][v]), acc[v]);
         }
#pragma unroll
         for (int q = 0; q < 4; q++) issue(u0 + q, w1);
      }
#pragma unroll
      for (int q = 0; q < NBW; q++) w1[q] = w2[q];
      load_words((b + 3) * 64 * NBW, w2);
   }
   flush();
   // write-out: slot q owns the local rows [q * PER, (q + 1) * PER); its LPR lanes hold one row
   int row_[PER], part_[PER];
#pragma unroll
   for (int jj = 0; jj < PER; jj++) {
      row_[jj] = cok ? a.wave_row[(size_t)w * NVMAX + g * PER + jj] : -1;
      part_[jj] = a.wave_part[(size_t)w * NVMAX + g * PER + jj];
   }
#pragma unroll
   for (int jj = 0; jj < PER; jj++) {
      if (row_[jj] < 0) continue;
      const float4 v = *reinterpret_cast<const float4 *>(lane_z + (g * PER + jj) * PANEL);
      float *dst = part_[jj] >= 0 ? a.part_val + (size_t)part_[jj] * (size_t)a.k + ccol : a.z + (size_t)row_[jj] * (size_t)a.ldz + ccol;
      *reinterpret_cast<float4 *>(dst) = v;
   }
}

template <int PAT>
static int launch_fusedmm_stream(const SweepArgs &a, int streams, int sop_udef, float sop_param, hipStream_t st) {
   const unsigned blocks = (unsigned)((a.wave_count + 3) / 4);
   if (blocks == 0) return ISPLIB_SUCCESS;
   if (streams == 2) hipLaunchKernelGGL((fusedmm_stream_kernel<PAT, 32, 16, 1, 2>), dim3(blocks), dim3(256), 0, st, a, sop_udef, sop_param);
   else if (streams == 4) hipLaunchKernelGGL((fusedmm_stream_kernel<PAT, 16, 32, 2, 2>), dim3(blocks), dim3(256), 0, st, a, sop_udef, sop_param);
   else hipLaunchKernelGGL((fusedmm_stream_kernel<PAT, 8, 64, 4, 2>), dim3(blocks), dim3(256), 0, st, a, sop_udef, sop_param);
   return check_launch("fusedmm_stream_kernel");
}

}  // namespace isplib

using namespace isplib;

// the two words this front end serves (enum values of csrc/fusedMM.h:18-74): everything else stays on fusedMM_csr_udef(_tasks)_hip
static int stream_pattern(int32_t imessage) {
   if (imessage == (0x2 | 0x10 | 0xF00 | 0x1000 | 0x10000)) return 1;      // COPY_RHS | DOT   | UDEF | MUL | ADD
   if (imessage == (0x5 | 0x50 | 0xF00 | 0x1000 | 0x10000)) return 2;      // SUBR     | NORMR | UDEF | MUL | ADD
   return 0;
}

extern "C" int isplib_fusedmm_stream_geometry(int streams, int *rows_per_wave, int *waves_resident) {
   clear_error();
   if (streams != 2 && streams != 4 && streams != 8) return fail(ISPLIB_FAIL, "isplib_fusedmm_stream_geometry: streams must be 2, 4 or 8");
   const GenStreamGeom ge = gen_stream_geom(streams);
   if (rows_per_wave) *rows_per_wave = ge.nvmax;
   if (waves_resident) {
      const int lds = 2 * 4 * (ge.nvmax + 1) * (64 / streams) * 4 * 4;
      int wgs = 163840 / lds;
      if (wgs > ge.wgs) wgs = ge.wgs;
      *waves_resident = device_cus() * wgs * 4;
   }
   return ISPLIB_SUCCESS;
}

extern "C" int isplib_suggest_fusedmm_stream(int32_t imessage, int64_t m, int64_t n, int64_t nnz, int64_t k, int *streams, int *slices, int *chunk) {
   // the SpMM's reuse rule on this kernel's geometry (half the rows per wave of a sum plan: twice the generations, each sweeping y
   // once per XCD): edges per generation and XCD >= 3 x rows of y.  Slices of ~2.6 MB of y on 128-column slots, ~3.8 MB on the
   // narrower ones; rows cut at ~0.4 of a stream's share.  Reddit shape, round 5, sigmoid word, ms (task list: 5.55 / 2.74 / 1.52):
   //   K=128: 7 / 15 / 31 / 46 / 62 slices 4.89 / 4.14 / 3.94 / 3.79 / 3.85; chunk 1457 / 2914 / 5828 (31 slices): 3.78 / 3.94 / 4.14
   //   K=64 : 4 / 8 / 16 / 24 / 32 slices 2.10 / 1.91 / 1.88 / 1.92 / 1.96;  chunk 1457 / 2914 / 5828: 1.87 / 1.88 / 2.17
   //   K=32 : 2 / 4 / 8 / 12 / 16 slices 1.08 / 1.04 / 1.04 / 1.07 / 1.08;   chunk 1457 / 2914 / 5828: 1.02 / 1.04 / 1.31
   clear_error();
   if (!stream_pattern(imessage) || m <= 0 || n <= 0 || nnz < (1LL << 22) || nnz >= (1LL << 31) || k < 4 || k > 128 || (k % 4) != 0 || n >= (1LL << 24)) return 0;
   if (!stream_domain_ok(n, k, nnz)) return 0;
   const int st = k <= 32 ? 8 : (k <= 64 ? 4 : 2);
   int rpw = 0, resident = 0;
   if (isplib_fusedmm_stream_geometry(st, &rpw, &resident) != ISPLIB_SUCCESS) return 0;
   const int64_t per_gen = (int64_t)rpw * resident;
   const int64_t gens = (m + per_gen - 1) / per_gen;
   if ((double)nnz / (double)gens / 8.0 < 3.0 * (double)n) return 0;
   int sl = (int)((double)n * (1024.0 / st) / (st == 2 ? 2.6e6 : 3.8e6) + 0.5);
   sl = sl < 1 ? 1 : (sl > 512 ? 512 : sl);
   int64_t ch = (int64_t)((double)nnz / ((double)gens * resident * st) / 2.4);
   ch = ch < 256 ? 256 : (ch > (1 << 20) ? (1 << 20) : ch);
   if (streams) *streams = st;
   if (slices) *slices = sl;
   if (chunk) *chunk = (int)ch;
   return 1;
}

extern "C" int fusedMM_csr_udef_stream_hip(int32_t imessage, int64_t m, int64_t n, int64_t k, int64_t nnz, const int64_t *pntrb,
                                           const int64_t *pntre, const isplib_stream_plan *plan, const float *x, int64_t ldx,
                                           const float *y, int64_t ldy, float *z, int64_t ldz, int sop_udef, float sop_param,
                                           void *workspace, size_t workspace_bytes, void *stream) {
   clear_error();
   const int pat = stream_pattern(imessage);
   if (!pat) return fail(ISPLIB_NO_OPT_IMPL, "fusedMM_csr_udef_stream_hip: COPY_RHS|DOT|UDEF|MUL|ADD and SUBR|NORMR|UDEF|MUL|ADD only (other words: fusedMM_csr_udef_hip)");
   if (sop_udef < ISPLIB_SOP_SIGMOID || sop_udef > ISPLIB_SOP_LEAKY_EXP)
      return fail(ISPLIB_UNDEFINED_USER_FUNCTION, "fusedMM_csr_udef_stream_hip: SOP_UDEF needs a built-in function (enum isplib_sop_udef)");
   if (m < 0 || n < 0 || k < 0 || nnz < 0) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: negative dimension");
   if (m == 0 || k == 0) return ISPLIB_SUCCESS;
   if (!plan) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: plan is required");
   if (plan->rows != m || plan->cols != n) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: the plan was built for another shape");
   if (plan->streams != 2 && plan->streams != 4 && plan->streams != 8) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: bad plan geometry (streams 2, 4 or 8)");
   if (plan->gens < 1 || plan->waves_per_gen < 1 || plan->rows_per_wave != gen_stream_geom(plan->streams).nvmax)
      return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: bad plan geometry (a plan of isplib_stream_plan_build_fusedmm_hip is required)");
   if (k < 4 || (k % 4) != 0 || k > 256 / plan->streams)
      return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: k must be a multiple of 4 within the plan's slot width (256 / streams columns); use fusedMM_csr_udef_tasks_hip");
   if (ldy < k || ldz < k || ldx < k || (ldx % 4) != 0 || (ldz % 4) != 0 || ((uintptr_t)x & 15) != 0 || ((uintptr_t)z & 15) != 0)
      return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: ldx, ldz multiples of 4 and >= k, x and z 16-byte aligned");
   if (n >= (1LL << 24) || ldy >= (1LL << 22)) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: n must be < 2^24 and ldy < 2^22 (24-bit address arithmetic)");
   const unsigned long long yb = (unsigned long long)n * (unsigned long long)ldy * 4ull;
   if (yb > BUF_LIMIT) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: dense operand larger than 3.5 GiB");
   if (!pntrb || !pntre || !x || !y || !z || !plan->wave_row || !plan->wave_part || !plan->wave_step_off || (plan->n_steps > 0 && !plan->words) ||
       (plan->n_hub > 0 && (!plan->hub_row || !plan->hub_off)))
      return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: null operand");
   if (plan->n_parts > 0) {
      if (!workspace || workspace_bytes < ((size_t)plan->n_parts * (size_t)k * sizeof(float))) return fail(ISPLIB_NOT_ENOUGH_MEM, "fusedMM_csr_udef_stream_hip: workspace too small");
      if (((uintptr_t)workspace & 255) != 0) return fail(ISPLIB_FAIL, "fusedMM_csr_udef_stream_hip: workspace must be 256-byte aligned");
   }
   SweepArgs a = {};
   a.k = k; a.nnz = nnz; a.pntrb = pntrb; a.pntre = pntre;
   a.y = y; a.ldy = ldy; a.ybytes = (unsigned)yb; a.z = z; a.ldz = ldz;
   a.g = x; a.ldg = ldx;
   a.abs_ids = 1;
   a.wave_row = plan->wave_row; a.wave_part = plan->wave_part;
   a.words = plan->words; a.wave_step_off = plan->wave_step_off; a.null_word = (unsigned)n;
   a.hub_row = plan->hub_row; a.hub_off = plan->hub_off; a.n_hub = plan->n_hub;
   a.part_val = (float *)workspace;
   hipStream_t st = (hipStream_t)stream;
   for (int gen = 0; gen < plan->gens; gen++) {
      a.wave_base = gen * plan->waves_per_gen;
      a.wave_count = plan->waves_per_gen;
      const int rc = pat == 1 ? launch_fusedmm_stream<1>(a, plan->streams, sop_udef, sop_param, st) : launch_fusedmm_stream<2>(a, plan->streams, sop_udef, sop_param, st);
      if (rc) return rc;
   }
   if (plan->n_hub > 0) {
      int64_t blocks = (plan->n_hub * (k / 4) + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL((sweep_hub_fold_kernel<OP_ADD, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
      const int rc = check_launch("sweep_hub_fold_kernel");
      if (rc) return rc;
   }
   return ISPLIB_SUCCESS;
}
