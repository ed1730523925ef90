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

// Read + write mixes in the shape of a k = 1 stream (conv_lin.hip): per 32-row block a wave reads 32 x INB bytes and writes
// 32 x OUTB bytes; persistent waves (grid x 4 of them) walk the blocks.
//  RW 0: the kernel's shape - every instruction 32 rows x 32 B (lane pair per row)
//  RW 1: the same bytes, every instruction 1 KiB contiguous (a block's rows are contiguous in memory)
//  RW 2: reads as the kernel's, writes contiguous          RW 3: reads contiguous, writes as the kernel's
template <int RW, int INB, int OUTB>
__global__ void __launch_bounds__(256) k_rw(const char *in, char *out, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  const int64_t nblk = rows / 32;
  for (int64_t blk = wave; blk < nblk; blk += nwaves) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const char *ib = in + blk * 32 * INB;
    char *ob = out + blk * 32 * OUTB;
    f4 v[INB / 32];
#pragma unroll
    for (int k = 0; k < INB / 32; ++k) {
      if (RW == 0 || RW == 2 || RW >= 4) v[k] = *(const f4 *)(ib + l31 * INB + k * 32 + h * 16);
      else v[k] = *(const f4 *)(ib + k * 1024 + lane * 16);
    }
#pragma unroll
    for (int k = 0; k < INB / 32; ++k) acc += v[k];
    if (RW == 4 || RW == 5) {
      // the operand-emitting epilogue's order: per 128-byte group the pieces at +0 and +64 (hi and lo halves of 16 channels),
      // then - after the next 16 channels' arithmetic (RW 4: 512 clocks of sleep; RW 5: nothing) - the pieces at +32 and +96
#pragma unroll
      for (int g = 0; g < OUTB / 128; ++g)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          char *p = ob + l31 * OUTB + g * 128 + m * 32 + h * 16;
          *(f4 *)p = acc;
          *(f4 *)(p + 64) = acc;
          if (RW == 4) __builtin_amdgcn_s_sleep(8);
        }
    } else {
#pragma unroll
    for (int k = 0; k < OUTB / 32; ++k) {
      if (RW == 0 || RW == 3) *(f4 *)(ob + l31 * OUTB + k * 32 + h * 16) = acc;
      else *(f4 *)(ob + k * 1024 + lane * 16) = acc;
    }
    }
  }
}
template <int RW, int INB, int OUTB>
static void run_rw(const char *in, char *out, int64_t rows, int grid, const char *name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_rw<RW, INB, OUTB>), dim3(grid), dim3(256), 0, 0, in, out, rows);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_rw<RW, INB, OUTB>), dim3(grid), dim3(256), 0, 0, in, out, rows);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double gb = (double)rows * (INB + OUTB) / 1e9;
  printf("{\"pattern\": \"%s\", \"in_row_B\": %d, \"out_row_B\": %d, \"grid\": %d, \"GB\": %.3f, \"us\": %.1f, \"TBps\": %.3f}\n", name, INB,
         OUTB, grid, gb, ms / reps * 1e3, gb / (ms / reps * 1e-3) / 1e3);
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

int main(int argc, char **argv) {
  if (argc > 1 && atoi(argv[1]) == 1) {      // the k = 1 stream mixes: 64 -> 128 channels (split f16 in, split f16 out), 380 k rows
    const int64_t rows = 380032;
    char *in, *out;
    hipMalloc(&in, rows * 512);
    hipMalloc(&out, rows * 1024);
    hipMemset(in, 0, rows * 512);
    for (int grid : {256, 512, 1024}) {
      run_rw<0, 256, 512>(in, out, rows, grid, "rw rows 32B / rows 32B");
      run_rw<1, 256, 512>(in, out, rows, grid, "rw contiguous / contiguous");
      run_rw<2, 256, 512>(in, out, rows, grid, "rw rows 32B / contiguous");
      run_rw<3, 256, 512>(in, out, rows, grid, "rw contiguous / rows 32B");
      run_rw<5, 256, 512>(in, out, rows, grid, "rw rows 32B / epilogue order");
      run_rw<4, 256, 512>(in, out, rows, grid, "rw rows 32B / epilogue order, 512 clk between a line's halves");
    }
    for (int grid : {512, 1024}) {
      run_rw<0, 512, 1024>(in, out, rows, grid, "rw rows 32B / rows 32B");
      run_rw<1, 512, 1024>(in, out, rows, grid, "rw contiguous / contiguous");
    }
    return 0;
  }
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
