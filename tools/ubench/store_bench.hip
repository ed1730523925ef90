// How fast can gfx950 WRITE?  1 GiB output written by different per-wave store shapes (the epilogues' patterns).
//  0: contiguous       lane l writes 16 B at wave_base + 16 l (1 KiB per instruction)
//  1: row chunks 32 B  rows of ROWB bytes; one instruction = 32 rows x 32 B (lane pair per row), 4 instructions
//                      complete a 128-byte line of each row (the operand-emitting epilogue)
//  2: same as 1 with nontemporal stores
//  3: row chunks 128 B one instruction = 8 rows x 128 B (8 lanes per row)
//  4: contiguous, nontemporal
//  5: row chunks 32 B, the 4 instructions of a line issued back to back by the same lanes (as 1) but rows taken
//     in an order that keeps a wave inside one 4 KiB page
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_store(char *out, int64_t rows, int rowb) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  const f4 v = {1.f + lane, 2.f, 3.f, 4.f};
  if (MODE == 0 || MODE == 4) {
    const int64_t total = rows * rowb;
    for (int64_t off = wave * 1024 + lane * 16; off < total; off += nwaves * 1024) {
      if (MODE == 4) __builtin_nontemporal_store(v, (f4 *)(out + off));
      else *(f4 *)(out + off) = v;
    }
  } else if (MODE == 1 || MODE == 2) {
    const int l31 = lane & 31, h = lane >> 5;
    const int groups = rowb / 128;
    for (int64_t blk = wave; blk * 32 < rows; blk += nwaves) {
      const int64_t row = blk * 32 + l31;
      if (row >= rows) continue;
      char *r = out + row * rowb;
      for (int g = 0; g < groups; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          char *p = r + g * 128 + m * 32 + h * 16;
          if (MODE == 2) {
            __builtin_nontemporal_store(v, (f4 *)p);
            __builtin_nontemporal_store(v, (f4 *)(p + 64));
          } else {
            *(f4 *)p = v;
            *(f4 *)(p + 64) = v;
          }
        }
    }
  } else if (MODE == 3) {
    const int l7 = lane & 7, r8 = lane >> 3;
    const int groups = rowb / 128;
    for (int64_t blk = wave; blk * 8 < rows; blk += nwaves) {
      const int64_t row = blk * 8 + r8;
      if (row >= rows) continue;
      char *r = out + row * rowb;
      for (int g = 0; g < groups; ++g) *(f4 *)(r + g * 128 + l7 * 16) = v;
    }
  }
}

template <int MODE>
static void run(char *buf, int64_t rows, int rowb, int grid, const char *name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, buf, rows, rowb);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, buf, rows, rowb);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double gb = (double)rows * rowb / 1e9;
  printf("{\"pattern\": \"%s\", \"rowb\": %d, \"grid\": %d, \"GB\": %.3f, \"us\": %.1f, \"TBps\": %.3f}\n", name, rowb, grid, gb,
         ms / reps * 1e3, gb / (ms / reps * 1e-3) / 1e3);
}

int main() {
  const int rowb = 1536;
  const int64_t rows = 631626;
  char *buf;
  hipMalloc(&buf, rows * rowb + 4096);
  for (int grid : {1024, 4096, 16384}) {
    run<0>(buf, rows, rowb, grid, "contiguous");
    run<4>(buf, rows, rowb, grid, "contiguous nontemporal");
    run<1>(buf, rows, rowb, grid, "rows 32B chunks");
    run<2>(buf, rows, rowb, grid, "rows 32B chunks nontemporal");
    run<3>(buf, rows, rowb, grid, "rows 128B chunks");
  }
  return 0;
}
