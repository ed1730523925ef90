// Does a wave that issues LDS-DMA gathers block its own matrix work?  Two kernels with the SAME totals per workgroup
// and iteration (24 x 1 KB DMA instructions from hot lines, 96 MFMA 32x32x16 f16, one barrier):
//   MIXED  every one of the 8 waves issues 3 DMAs, then 12 MFMAs
//   SPEC   waves 4-7 issue 6 DMAs each, waves 0-3 run 24 MFMAs each (one loader + one matrix wave per SIMD)
// plus the two halves alone (DMA only / MFMA only) for reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0 mixed, 1 specialised, 2 DMA only (all waves 3 each), 3 MFMA only (all waves 12 each), 4 specialised with the
// matrix waves on SIMDs {0,2} and the loaders on {1,3} (waves 0,4,1,5 vs 2,6,3,7), 5 specialised with REGISTER loads
// instead of LDS-DMA, 6 mixed with register loads
template <int MODE>
__global__ void __launch_bounds__(512, 2) k_spec(const char *src, float *sink, int iters) {
  __shared__ __attribute__((aligned(128))) char lds[8 * 6 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc0 = {0}, acc1 = {0};
  f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  a[0] = (_Float16)lane;
  const char *my = src + (size_t)(blockIdx.x & 63) * 65536 + lane * 16;
  if ((MODE == 7 || MODE == 8) && wave >= 4) __builtin_amdgcn_s_setprio(3);     // loaders outrank the matrix waves
  for (int it = 0; it < iters; ++it) {
    constexpr bool SPEC = MODE == 1 || MODE == 4 || MODE == 5 || MODE == 7 || MODE == 8;
    const bool is_matrix_wave = MODE == 4 ? ((wave & 3) < 2) : wave < 4;
    const bool loader = SPEC ? !is_matrix_wave : true;
    const bool matrix = SPEC ? is_matrix_wave : true;
    const int ndma = SPEC ? 6 : 3, nmma = SPEC ? 24 : 12;
    if (loader && MODE != 3 && MODE < 9) {
      if (MODE == 5 || MODE == 6) {
        u32x4 r[6];
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (q < ndma) r[q] = *reinterpret_cast<const u32x4 *>(my + ((it * 6 + q) & 63) * 1024);
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (q < ndma) asm volatile("" :: "v"(r[q]));
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (q < ndma) {
            char *dst = lds + (wave * 6 + q) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(my + ((it * 6 + q) & 63) * 1024),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
          }
      }
    }
    if (MODE == 9 || MODE == 10) {
      f16x8 fr[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) fr[q] = *reinterpret_cast<const f16x8 *>(lds + ((q * 64 + lane + it) & 2047) * 16);
      if (MODE == 9) {
#pragma unroll
        for (int q = 0; q < 12; q += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) a += fr[q];
    } else if (matrix && MODE != 2) {
#pragma unroll
      for (int q = 0; q < 24; q += 2)
        if (q < nmma) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
          if (MODE == 8) __builtin_amdgcn_s_sleep(1);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  sink[(size_t)blockIdx.x * 512 + tid] = s + lds[tid];
}

extern "C" int ub_spec(const void *src, void *sink, int iters, int mode, int grid, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(k_spec<0>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 1) hipLaunchKernelGGL(k_spec<1>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 2) hipLaunchKernelGGL(k_spec<2>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 3) hipLaunchKernelGGL(k_spec<3>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 4) hipLaunchKernelGGL(k_spec<4>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 5) hipLaunchKernelGGL(k_spec<5>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 7) hipLaunchKernelGGL(k_spec<7>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 8) hipLaunchKernelGGL(k_spec<8>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 9) hipLaunchKernelGGL(k_spec<9>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 10) hipLaunchKernelGGL(k_spec<10>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  if (mode == 6) hipLaunchKernelGGL(k_spec<6>, dim3(grid), dim3(512), 0, st, (const char *)src, (float *)sink, iters);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
