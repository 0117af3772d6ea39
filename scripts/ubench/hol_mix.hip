// hol_mix.hip -- microbenchmark (round 5): does a wave whose gathers MISS the L2 slow down the waves of the same CU whose gathers hit?
// One workgroup of 8 waves per CU (two per SIMD, the stream kernel's occupancy).  Waves 0-3 ("hit" waves) do `steps` 1-KiB gathers from a
// 1 MB table (L2 hits) and time themselves; waves 4-7 are, per run,
//   idle      : return at once                                  -> what four hit waves per CU do alone
//   hit too   : gather from the same 1 MB table until the hit waves are done   -> plain sharing of the CU's address path
//   missing   : gather random rows of a 4 GB table (every line an L2 miss, ~1,100 clocks) until the hit waves are done
// If the CU returned data strictly in order ACROSS waves, the hit waves of the third run would crawl at the misses' pace; if the order is
// per wave only, they lose about what the second run loses.  Prints clocks per gather of the hit waves and what the other four waves got done.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/hol_mix.hip -o scripts/ubench/hol_mix ; run: scripts/ubench/hol_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int v4i_t;
constexpr int DEPTH = 16;

// role of waves 4-7: 0 idle, 1 hit table, 2 miss table
__global__ __launch_bounds__(512, 1) void mix_kernel(const float *hit, unsigned hit_rows, const float *miss, unsigned miss_rows, int steps, int role,
                                                    unsigned long long *clocks, unsigned long long *others_done) {
   __shared__ int done;
   const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   if (threadIdx.x == 0) done = 0;
   __syncthreads();
   const bool hit_wave = wave < 4;
   if (!hit_wave && role == 0) return;
   const bool use_miss = !hit_wave && role == 2;
   const float *table = use_miss ? miss : hit;
   const unsigned rows = use_miss ? miss_rows : hit_rows;
   // a 4 GB table does not fit one buffer descriptor's 32-bit offsets comfortably: 64-bit global loads
   const int g = lane >> 4, lc = lane & 15;
   unsigned seed = ((blockIdx.x * 8 + wave) * 4 + g) * 2654435761u + 12345u;
   auto next_row = [&]() -> const v4i_t * {
      seed = seed * 1664525u + 1013904223u;
      const unsigned r = (unsigned)(((unsigned long long)(seed >> 4) * rows) >> 28);      // uniform in [0, rows)
      return reinterpret_cast<const v4i_t *>(table + (size_t)r * 64) + lc;
   };
   float acc[4] = {0.f, 0.f, 0.f, 0.f};
   v4i_t t[DEPTH];
#pragma unroll
   for (int u = 0; u < DEPTH; u++) t[u] = *next_row();
   const unsigned long long t0 = wall_clock64();
   long long count = 0;
   for (int s = 0; hit_wave ? s < steps : true; s += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
#pragma unroll
         for (int v = 0; v < 4; v++) acc[v] += __int_as_float(t[u][v]);
         t[u] = *next_row();
      }
      count += DEPTH;
      if (!hit_wave && __hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 4) break;     // every wave reaches an exit
   }
   const unsigned long long t1 = wall_clock64();
#pragma unroll
   for (int u = 0; u < DEPTH; u++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[v] += __int_as_float(t[u][v]);
   if (lane == 0) {
      if (hit_wave) {
         clocks[blockIdx.x * 4 + wave] = t1 - t0;
         atomicAdd(&done, 1);
      } else {
         others_done[blockIdx.x * 4 + (wave - 4)] = (unsigned long long)count;
      }
   }
   if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) clocks[0] = 0;
}

int main(int argc, char **argv) {
   const int steps = argc > 1 ? atoi(argv[1]) : 8192;
   const unsigned hit_rows = 4096;                                  // x 256 B = 1 MB
   const unsigned miss_rows = argc > 2 ? (unsigned)atoi(argv[2]) : (1u << 24);   // x 256 B = 4 GB
   hipDeviceProp_t p;
   (void)hipGetDeviceProperties(&p, 0);
   const int cus = p.multiProcessorCount;
   int wall_khz = 0;
   (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
   const double core_per_wall = (p.clockRate > 0 && wall_khz > 0) ? (double)p.clockRate / wall_khz : 24.0;
   float *hit, *miss;
   unsigned long long *clocks, *others;
   (void)hipMalloc(&hit, (size_t)hit_rows * 256);
   if (hipMalloc(&miss, (size_t)miss_rows * 256) != hipSuccess) { printf("no room for the %u-row table\n", miss_rows); return 1; }
   (void)hipMalloc(&clocks, (size_t)cus * 4 * 8);
   (void)hipMalloc(&others, (size_t)cus * 4 * 8);
   (void)hipMemset(hit, 0, (size_t)hit_rows * 256);
   (void)hipMemset(miss, 0, (size_t)miss_rows * 256);
   printf("device %s: %d CUs, core clock %d kHz, wall clock %d kHz; hit table 1 MB, miss table %.1f GB; %d gathers per hit wave, %d in flight per wave\n", p.name, cus,
          p.clockRate, wall_khz, miss_rows * 256.0 / 1e9, steps, DEPTH);
   const char *names[3] = {"waves 4-7 idle", "waves 4-7 gather from the 1 MB table too", "waves 4-7 gather from the 4 GB table (misses)"};
   std::vector<unsigned long long> h(cus * 4), o(cus * 4);
   for (int rep = 0; rep < 2; rep++)
      for (int role = 0; role < 3; role++) {
         (void)hipMemset(others, 0, (size_t)cus * 4 * 8);
         hipLaunchKernelGGL(mix_kernel, dim3(cus), dim3(512), 0, 0, hit, hit_rows, miss, miss_rows, steps, role, clocks, others);
         (void)hipDeviceSynchronize();
         (void)hipMemcpy(h.data(), clocks, h.size() * 8, hipMemcpyDeviceToHost);
         (void)hipMemcpy(o.data(), others, o.size() * 8, hipMemcpyDeviceToHost);
         double sum = 0, osum = 0;
         for (auto v : h) sum += (double)v;
         for (auto v : o) osum += (double)v;
         const double core_clocks = sum / h.size() * core_per_wall;        // mean per hit wave
         printf("%-50s hit waves: %8.0f core clocks for %d gathers = %6.1f clocks per gather and wave (%5.1f per gather and CU); waves 4-7 did %.0f gathers each\n",
                names[role], core_clocks, steps, core_clocks / steps, core_clocks / steps / 4.0, osum / o.size());
      }
   return 0;
}
