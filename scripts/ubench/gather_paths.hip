// gather_paths.hip -- microbenchmark (round 5): what a CU can gather per clock from an L2-RESIDENT table, by return path.
//   A  buffer_load_dwordx4 into VGPRs (what every SpMM kernel of this library does): 4 rows x 256 B per wave instruction
//   B  buffer_load_dwordx4 ... lds (LDS-DMA: the same address work, the data lands in LDS, no VGPR write-back)
//   C  both interleaved, R VGPR gathers per LDS-DMA gather
// If B (or C) moved more bytes per clock than A, the 26 cycles per 1-KiB gather that bound the stream kernel would be a
// property of the VGPR return path and a kernel could buy bandwidth by mixing paths; if not, they are the address side's.
// Every wave: STEPS steps of one 1-KiB gather (rows chosen by a per-slot LCG inside a table of `rows` rows of 256 B), DEPTH
// gathers in flight, 2 waves per SIMD (the stream kernel's occupancy).  Prints ns per launch and bytes / clock / CU.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/gather_paths.hip -o /tmp/gather_paths ; run: /tmp/gather_paths
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int v4i_t;
typedef int v4i_rsrc_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma_b128(v4i_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
   unsigned keep;
   asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(voff), "s"(rsrc), "s"(lds_addr)
                : "memory");
}

constexpr int DEPTH = 16;        // gathers in flight per wave and path
constexpr int LDS_PER_WAVE = DEPTH * 1024;

// MODE 0: VGPR only; 1: LDS-DMA only; 2: R VGPR gathers, then one LDS-DMA gather;
// MODE 3: VGPR only, R gathers in flight, issued from inline assembly with HAND-WRITTEN waits (s_waitcnt vmcnt(R - 1) before every
// consumption).  Mode 0 leaves the waits to the compiler, which joins the state of the prologue (all gathers issued back to
// back) with the loop's and ends up with a ladder vmcnt(15) ... vmcnt(1) in the first half of the loop body and no wait in the
// second: every wave drains its pipeline once per DEPTH steps.  Mode 3 is what the hardware does when it is never drained.
template <int MODE, int R>
__global__ __launch_bounds__(256, 2) void gather_kernel(const float *table, unsigned rows, int steps, float *out) {
   __shared__ __attribute__((aligned(16))) float ring[4 * LDS_PER_WAVE / 4];       // 64 KB per workgroup: two workgroups per CU
   const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
   const int g = lane >> 4, lc = lane & 15;
   __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, (int)(rows * 256u), 0x00020000);
   v4i_rsrc_t rs;
   {
      const uint64_t base = (uint64_t)table;
      rs.x = (int)(uint32_t)base; rs.y = (int)(uint32_t)(base >> 32); rs.z = (int)(rows * 256u); rs.w = 0x00020000;
   }
   unsigned seed = (blockIdx.x * 4 + wave) * 4 + g + 12345u;
   auto next_off = [&]() -> unsigned {
      seed = seed * 1664525u + 1013904223u;
      return ((seed >> 8) % rows) * 256u + (unsigned)lc * 16u;
   };
   const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(size_t)(__attribute__((address_space(3))) float *)ring + (unsigned)wave * LDS_PER_WAVE));   // LDS byte address of the wave's ring
   float acc[4] = {0.f, 0.f, 0.f, 0.f};
   if constexpr (MODE == 3) {
      v4i_t tt[R];
#pragma unroll
      for (int u = 0; u < R; u++) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(tt[u]) : "v"(next_off()), "s"(rs));
      for (int s = 0; s < steps; s += R) {
#pragma unroll
         for (int u = 0; u < R; u++) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tt[u]) : "n"(R - 1));
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v] += __int_as_float(tt[u][v]);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(tt[u]) : "v"(next_off()), "s"(rs));
         }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < R; u++) asm volatile("" : "+v"(tt[u]));
      if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
      return;
   }
   v4i_t t[DEPTH];
   if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < DEPTH; u++) t[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)next_off(), 0, 0);
   }
   for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; u++) {
         if (MODE != 1) {
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v] += __int_as_float(t[u][v]);
            t[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)next_off(), 0, 0);
         }
         if (MODE == 1 || (MODE == 2 && (u % R) == R - 1)) lds_dma_b128(rs, next_off(), ring_base + (unsigned)(u * 1024));
      }
      if (MODE != 0) {                                   // consume one slot of the ring now and then (keeps the DMA honest)
         asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
         const float4 r = *reinterpret_cast<const float4 *>(ring + wave * (LDS_PER_WAVE / 4) + lane * 4);
         acc[0] += r.x;
      }
   }
   if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < DEPTH; u++)
#pragma unroll
         for (int v = 0; v < 4; v++) acc[v] += __int_as_float(t[u][v]);
   }
   if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}

template <int MODE, int R>
static void run(const char *name, const float *table, unsigned rows, int steps, float *out, int cus, double mhz) {
   const int blocks = cus * 2;
   hipEvent_t a, b;
   hipEventCreate(&a); hipEventCreate(&b);
   for (int w = 0; w < 2; w++) hipLaunchKernelGGL((gather_kernel<MODE, R>), dim3(blocks), dim3(256), 0, 0, table, rows, steps, out);
   hipEventRecord(a, 0);
   const int reps = 5;
   for (int w = 0; w < reps; w++) hipLaunchKernelGGL((gather_kernel<MODE, R>), dim3(blocks), dim3(256), 0, 0, table, rows, steps, out);
   hipEventRecord(b, 0);
   hipEventSynchronize(b);
   float ms = 0.f;
   hipEventElapsedTime(&ms, a, b);
   ms /= reps;
   const double per_wave = (MODE == 0 || MODE == 3 ? 1.0 : MODE == 1 ? 1.0 : 1.0 + 1.0 / R) * steps;      // gathers of 1 KiB per wave
   const double bytes = per_wave * 1024.0 * blocks * 4;
   const double clocks = ms * 1e-3 * mhz * 1e6;
   printf("%-34s %8.3f ms  %7.2f TB/s  %6.2f B/clk/CU  %6.1f clk per 1-KiB gather per CU\n", name, ms, bytes / (ms * 1e-3) / 1e12,
          bytes / clocks / cus, clocks * cus / (per_wave * blocks * 4));
}

int main(int argc, char **argv) {
   const unsigned rows = argc > 1 ? (unsigned)atoi(argv[1]) : 4096;          // x 256 B: 1 MB by default (inside every XCD's L2)
   const int steps = argc > 2 ? atoi(argv[2]) : 16384;
   hipDeviceProp_t p;
   hipGetDeviceProperties(&p, 0);
   const int cus = p.multiProcessorCount;
   const double mhz = p.clockRate / 1000.0;
   float *table, *out;
   hipMalloc(&table, (size_t)rows * 256);
   hipMalloc(&out, 256);
   hipMemset(table, 0, (size_t)rows * 256);
   printf("device: %s, %d CUs, %.0f MHz; table %u rows x 256 B = %.1f MB; %d steps per wave, %d in flight per path, 8 waves per CU\n", p.name, cus, mhz,
          rows, rows * 256.0 / 1e6, steps, DEPTH);
   run<0, 1>("A  VGPR gathers", table, rows, steps, out, cus, mhz);
   run<1, 1>("B  LDS-DMA gathers", table, rows, steps, out, cus, mhz);
   run<2, 1>("C  1 VGPR : 1 LDS-DMA", table, rows, steps, out, cus, mhz);
   run<2, 2>("C  2 VGPR : 1 LDS-DMA", table, rows, steps, out, cus, mhz);
   run<2, 4>("C  4 VGPR : 1 LDS-DMA", table, rows, steps, out, cus, mhz);
   run<0, 1>("A  VGPR gathers (again)", table, rows, steps, out, cus, mhz);
   run<3, 8>("D  VGPR, hand-counted, 8 in flight", table, rows, steps, out, cus, mhz);
   run<3, 16>("D  VGPR, hand-counted, 16 in flight", table, rows, steps, out, cus, mhz);
   run<3, 32>("D  VGPR, hand-counted, 32 in flight", table, rows, steps, out, cus, mhz);
   run<3, 48>("D  VGPR, hand-counted, 48 in flight", table, rows, steps, out, cus, mhz);
   hipFree(table); hipFree(out);
   return 0;
}
