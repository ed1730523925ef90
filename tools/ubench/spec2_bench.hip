// Wave roles, second look (round 3).  spec_bench.hip synchronised loaders and matrix waves with `vmcnt(0)` + a barrier
// EVERY iteration, which exposes the DMA round trip each time.  Here the two roles run (a) free of each other, (b) with a
// barrier per iteration but a COUNTED wait that leaves one iteration of DMA in flight - the shape a real pipeline has.
// Per workgroup and iteration: 24 x 1 KB LDS-DMA (or register loads) and 96 MFMA 32x32x16 f16, as in spec_bench.
//   MODE 0  matrix waves only (waves 0-3: 24 MFMA each)
//   MODE 1  loader waves only (waves 4-7: 6 DMA each, vmcnt(6) throttle)
//   MODE 2  both roles, no synchronisation between them
//   MODE 3  both roles, one barrier per iteration, loaders wait vmcnt(6) before it (previous iteration landed)
//   MODE 4  mixed: all 8 waves, 3 DMA interleaved with 12 MFMA, vmcnt(3) throttle, no barrier
//   MODE 5  mixed, 4 waves only (6 DMA interleaved with 24 MFMA each), vmcnt(6), no barrier
//   MODE 6  as 2, matrix waves also read 16 x ds_read_b128 per iteration (fragment traffic)
//   MODE 7  as 3 with the fragment reads
//   MODE 8  as 2 with REGISTER loads in the loaders
//   MODE 9  as 2, loaders on waves 4-5 only (12 DMA each)
// GATHER: every group of 8 lanes reads a pseudo-random 128-byte row chunk of a `span`-byte table instead of hot lines.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool GATHER>
__global__ void __launch_bounds__(512, 1) k_spec2(const char *src, float *sink, int iters, uint32_t span_rows) {
  __shared__ __attribute__((aligned(128))) char lds[8 * 6 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (MODE == 5 && wave >= 4) return;
  f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
  f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  a[0] = (_Float16)lane;
  const char *my = src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  constexpr bool MIXED = MODE == 4 || MODE == 5;
  const bool matrix = MIXED || (MODE != 1 && wave < 4);
  const bool loader = MIXED || (MODE != 0 && (MODE == 9 ? (wave == 4 || wave == 5) : wave >= 4));
  constexpr int NDMA = MODE == 4 ? 3 : (MODE == 9 ? 12 : 6);
  constexpr int NMMA = MODE == 4 ? 12 : 24;
  constexpr bool BARRIER = MODE == 3 || MODE == 7;
  constexpr bool FRAGS = MODE == 6 || MODE == 7;
  uint32_t rng = blockIdx.x * 9781u + wave * 131u + (lane >> 3) * 7u + 12345u;
  auto src_of = [&](int it, int q) -> const char * {
    if (GATHER) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t row = (rng >> 8) % span_rows;
      return src + (size_t)row * 128 + (lane & 7) * 16;
    }
    return my + ((it * NDMA + q) & 63) * 1024;
  };
  f16x8 fr[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) fr[q] = a;
  for (int it = 0; it < iters; ++it) {
    if (MIXED) {
#pragma unroll
      for (int q = 0; q < NDMA; ++q) {
        char *dst = lds + ((wave * 6 + q) % 48) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_of(it, q),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NMMA / NDMA; m += 4) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc3, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NDMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      if (loader) {
        if (MODE == 8) {
          u32x4 r[NDMA];
#pragma unroll
          for (int q = 0; q < NDMA; ++q) r[q] = *reinterpret_cast<const u32x4 *>(src_of(it, q));
#pragma unroll
          for (int q = 0; q < NDMA; ++q) asm volatile("" ::"v"(r[q]));
        } else {
#pragma unroll
          for (int q = 0; q < NDMA; ++q) {
            char *dst = lds + ((wave * 6 + q) % 48) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_of(it, q),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
          }
          if (NDMA == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
      }
      if (matrix) {
        if (FRAGS) {
#pragma unroll
          for (int q = 0; q < 16; ++q)
            fr[q] = *reinterpret_cast<const f16x8 *>(lds + (((q * 64 + lane) * 16 + it * 16) & (48 * 1024 - 16)));
        }
#pragma unroll
        for (int m = 0; m < NMMA; m += 4) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FRAGS ? fr[m & 15] : a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, FRAGS ? fr[(m + 1) & 15] : a, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FRAGS ? fr[(m + 2) & 15] : a, b, acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, FRAGS ? fr[(m + 3) & 15] : a, acc3, 0, 0, 0);
        }
      }
      if (BARRIER) __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  sink[(size_t)blockIdx.x * 512 + tid] = s + lds[tid];
}

#define LAUNCH(M)                                                                                                           \
  if (mode == M) {                                                                                                          \
    if (gather) hipLaunchKernelGGL((k_spec2<M, true>), dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters, span_rows); \
    else hipLaunchKernelGGL((k_spec2<M, false>), dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters, span_rows);       \
  }

extern "C" int ub_spec2(const void *src, void *sink, int iters, int mode, int gather, uint32_t span_rows, int grid, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9)
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
