// What does ds_read_b64_tr_b16 return?  LDS holds element index i at position i (u16); every lane passes its own
// 8-byte-aligned address (pattern selectable) and prints the 4 elements it got.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));

__global__ void probe(int pattern, int *out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int elem;                                     // first element this lane points at
  if (pattern == 0) elem = l * 4;               // lane-linear: 16 lanes cover 64 consecutive elements
  else if (pattern == 1) elem = (l & 15) / 4 * 64 + (l & 3) * 4 + (l >> 4) * 16;   // row stride 64 elems: [4 rows][16 cols] per group, groups side by side
  else elem = (l & 3) * 4 + ((l & 15) >> 2) * 32 + (l >> 4) * 128;                  // row stride 32 elems
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int main() {
  int *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "   ");
  }
  return 0;
}
