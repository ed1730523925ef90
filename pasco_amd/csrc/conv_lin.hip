// Tall "linear" layers on pre-split operands: kernel volume 1, identity map, 64 or 128 input channels - the point MLP,
// the 1x1 heads and the K / V / mask-feature projections of the mask transformer (hundreds of thousands of rows, a
// reduction of 2..4 stages).  k_conv_dma spends such a launch in its prologue and epilogue (one 128-row tile per
// workgroup, nothing to overlap with: 1.5 TB/s).  Here the WEIGHTS are the resident operand:
//
//   * the whole [cout][cin] split kernel sits in LDS (<= 128 KB), loaded once per workgroup, laid out
//     [granule (k chunk, plane, half)][column] so that a fragment read is 32 consecutive 16-byte granules;
//   * every wave streams its own blocks of 32 * TM rows (persistent, grid-stride): the activation fragments go from
//     global memory straight to registers in MFMA layout (row = lane & 31: one 16-byte load per lane and k chunk), the
//     next block is loaded while the current one is multiplied, the result tiles leave through the shared epilogue
//     (h2_store_tile: bias / BN / activation / residual / table residual / operand emission) - no barrier after the
//     weight load, no LDS traffic for activations.
// Bound: HBM (rows in + rows out once); MFMA time is ~1/6 of that.
#include "conv_h2_common.h"

constexpr int LIN_WAVES = 8;

template <int CIN>
struct LinShape {
  static constexpr int KCH = CIN / 16;               // 16-channel MFMA steps
  static constexpr int TM = 1;                       // 32-row tiles per wave block (two fragment sets live: current + next)
  static constexpr int MAXCOLS = CIN == 64 ? 384 : 256;
  static constexpr int LDS_BYTES = MAXCOLS * CIN * 4;
};

template <int CIN, bool EMIT>
__global__ void __launch_bounds__(LIN_WAVES * 64) k_conv_lin(ConvArgsH a, int colsp, int64_t nblk) {
  using S = LinShape<CIN>;
  constexpr int KCH = S::KCH, TM = S::TM;
  __shared__ __attribute__((aligned(128))) char lds[S::LDS_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int rowb = (CIN / 32) * 128;                 // bytes per operand row (activations and kernel rows alike)

  // ---- kernel -> LDS: granule g = gid * colsp + col, gid = (kc * 2 + plane) * 2 + half ----------------------------------
  {
    const char *w = (const char *)a.w_split;
    const int ngran = KCH * 4 * colsp;
    for (int g = threadIdx.x; g < ngran; g += LIN_WAVES * 64) {
      const int gid = g / colsp, col = g - gid * colsp;
      const int kc = gid >> 2, plane = (gid >> 1) & 1, half = gid & 1;
      const int c = col < a.cout ? col : a.cout - 1;
      const uint4 v = *reinterpret_cast<const uint4 *>(w + (int64_t)c * rowb + (kc >> 1) * 128 + plane * 64 + (kc & 1) * 32 + half * 16);
      *reinterpret_cast<uint4 *>(lds + (int64_t)g * 16) = v;
    }
  }
  __syncthreads();

  const char *xin = (const char *)a.in_split;
  const int lane_off = h * 16;                       // + (kc >> 1) * 128 + plane * 64 + (kc & 1) * 32
  const int ncolt = colsp >> 5;
  const int64_t stride = (int64_t)gridDim.x * LIN_WAVES;

  f16x8 xh[TM][KCH], xl[TM][KCH], nh[TM][KCH], nl[TM][KCH];
  auto load = [&](int64_t blk, f16x8 (&dh)[TM][KCH], f16x8 (&dl)[TM][KCH]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int64_t row = blk * (TM * 32) + i * 32 + l31;
      if (row >= a.n_out) row = a.n_out - 1;
      const char *p = xin + row * rowb + lane_off;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        dh[i][kc] = *reinterpret_cast<const f16x8 *>(p + (kc >> 1) * 128 + (kc & 1) * 32);
        dl[i][kc] = *reinterpret_cast<const f16x8 *>(p + (kc >> 1) * 128 + 64 + (kc & 1) * 32);
      }
    }
  };

  int64_t blk = (int64_t)blockIdx.x * LIN_WAVES + wave;
  if (blk < nblk) load(blk, xh, xl);
  for (; blk < nblk; blk += stride) {
    const bool more = blk + stride < nblk;
    if (more) load(blk + stride, nh, nl);
    const int64_t m0 = blk * (TM * 32);
    for (int j = 0; j < ncolt; ++j) {
      f32x16 acc[TM][1];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;
      const char *wb = lds + ((int64_t)h * colsp + j * 32 + l31) * 16;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        const f16x8 wh = *reinterpret_cast<const f16x8 *>(wb + (int64_t)(kc * 4 + 0) * colsp * 16);
        const f16x8 wl = *reinterpret_cast<const f16x8 *>(wb + (int64_t)(kc * 4 + 2) * colsp * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          // small terms first, like the tile kernels (same sums bit for bit)
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[i][kc], acc[i][0], 0, 0, 0);
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[i][kc], acc[i][0], 0, 0, 0);
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[i][kc], acc[i][0], 0, 0, 0);
        }
      }
      h2_store_tile<TM, 1, EMIT>(a, acc, m0, j * 32, 0, 0, h, l31);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
          xh[i][kc] = nh[i][kc];
          xl[i][kc] = nl[i][kc];
        }
    }
  }
}

static int lin_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

// -1: shape not served (the caller goes on to k_conv_dma / k_conv_h2)
int ph_conv_lin_try(const ConvArgsH &a, hipStream_t st) {
  if (a.kvol != 1 || a.nbr != nullptr || a.ksplit != 1 || a.in_split == nullptr || a.w_split == nullptr) return -1;
  if (!(a.cin == 64 || a.cin == 128) || a.cpad != a.cin || a.n_in < a.n_out) return -1;
  const int colsp = ((a.cout + 31) / 32) * 32;
  if (colsp > (a.cin == 64 ? LinShape<64>::MAXCOLS : LinShape<128>::MAXCOLS)) return -1;
  if (a.n_out < 32768) return -1;                    // the weight load per workgroup needs rows to amortise over
  const bool emit = a.out_split != nullptr;
  const int tm = 1;
  const int64_t nblk = (a.n_out + tm * 32 - 1) / (tm * 32);
  int64_t grid = (nblk + LIN_WAVES - 1) / LIN_WAVES;
  if (grid > lin_cus()) grid = lin_cus();
  ConvArgsH args = a;
  args.n_row_tiles = (int)nblk;
  args.n_col_tiles = 1;
#define LIN_GO(CIN_, EMIT_) hipLaunchKernelGGL((k_conv_lin<CIN_, EMIT_>), dim3((unsigned)grid), dim3(LIN_WAVES * 64), 0, st, args, colsp, nblk)
  if (a.cin == 64) {
    if (emit) LIN_GO(64, true); else LIN_GO(64, false);
  } else {
    if (emit) LIN_GO(128, true); else LIN_GO(128, false);
  }
#undef LIN_GO
  PH_LAUNCH_CHECK();
  ph_record_cfg(2, tm * 32, colsp, 16, 1, emit ? 1 : 0, 6, LIN_WAVES);
  return 0;
}
