// k = 1 products (linear layers, 1x1x1 convolutions) of 64 / 128 input channels as a ROW STREAM.
//
// Why not k_conv_dma: a k = 1 launch gives a 128 x 128 tile two to four 32-channel stages of matrix work, so a
// workgroup's life is its start-up (kernel arguments, index table, first DMA: ~20 %), its epilogue (45 - 70 %) and the
// drain of its stores, one after the other, with two workgroups per CU as the only overlap (tools/dma_trace.py,
// profiles/r4v_dma_trace_k1_phases.txt): 0.27 - 0.44 of the HBM rate on launches that are pure streams.
//
// Here the weights of one 128-column tile ([cin][128] split f16: 32 / 64 KB) are put in LDS ONCE per workgroup, and every
// WAVE then walks 32-row blocks on its own: the block's operand rows global -> registers (each lane the 16-byte runs of its
// row the MFMA wants: no LDS, no barrier), the weight fragments from LDS, 24 / 48 MFMAs, the shared epilogue
// (conv_h2_common.h) straight from the accumulators.  The next block's rows are requested between the current block's
// matrix work and its epilogue, and no wave ever waits for another: 8 independent streams per CU.
//
// Measured (profiles/README.md, round 4 "k = 1 streams"): 2 - 14 % faster than k_conv_dma on the launches it takes - not
// the factor the structure promised.  What bounds both kernels is the memory side of THIS access shape: the bare read + write
// mix of the 64 -> 128 stream takes 59 us with the MFMA layout's 32 bytes per row and instruction (one write request per
// lane pair: 6.1 M requests for 195 MB, TCP_TCC_WRITE_REQ) against 44 us with contiguous stores and 40 us fully contiguous
// (tools/ubench/store_bench 1); the kernels take 83 - 90 us, their compute (26 us alone) not overlapping the drain of their
// own stores: loads and stores retire in order on one counter, so the wait for the next block's rows is also a wait for this
// block's stores.  A counted wait that leaves the stores in flight was built and measured (-5 %), then removed again.
//
// Same products in the same order as k_conv_dma / k_conv_h2 (per 32-channel chunk: k-steps 0, 1; per step hi.lo, lo.hi,
// hi.hi) and the same epilogue: results are bit-identical (tests/test_hip_lin.py).
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "conv_h2_common.h"

constexpr int LIN_BN = 128;      // output channels of one workgroup's weight tile
constexpr int LIN_NT = 256;      // 4 waves

template <int NCH>
struct LinRows {                 // one lane's part of a 32-row block: [chunk][k-step] runs of 8 channels, hi and lo halves
  f16x8 hi[NCH][2], lo[NCH][2];
};

template <int NCH, bool EMIT>
__global__ void __launch_bounds__(LIN_NT, 2) k_conv_lin(ConvArgsH a) {
  constexpr int W_BYTES = NCH * LIN_BN * 128;
  __shared__ __attribute__((aligned(128))) char lds[W_BYTES + H2_PAR_VECS * LIN_BN * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int n0 = (int)blockIdx.y * LIN_BN;
  const int cout = a.cout;
  const uint32_t rsb = 4u * (uint32_t)a.cpad;      // bytes per operand row

  // ---- the tile's weights -> LDS: [chunk][column][8 x 16 B], run p of column n at slot p ^ ((n >> 1) & 7) (the
  // fragment reads of 32 consecutive columns then touch every bank group once) -----------------------------------------
  static_assert((NCH * LIN_BN * 8) % (8 * LIN_NT) == 0, "whole batches of 8 loads per thread");
  for (int u0 = 0; u0 < NCH * LIN_BN * 8; u0 += 8 * LIN_NT) {      // 8 loads in flight per thread, then their LDS stores
    uint4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = u0 + q * LIN_NT + tid;
      const int p = u & 7, n = (u >> 3) & (LIN_BN - 1), c = u / (LIN_BN * 8);
      int nn = n0 + n;
      nn = nn < cout ? nn : cout - 1;              // columns >= cout are never stored: any finite row will do
      v[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(a.w_split) + (int64_t)nn * rsb + c * 128 + p * 16);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = u0 + q * LIN_NT + tid;
      const int p = u & 7, n = (u >> 3) & (LIN_BN - 1), c = u / (LIN_BN * 8);
      *reinterpret_cast<uint4 *>(lds + c * (LIN_BN * 128) + n * 128 + ((p ^ ((n >> 1) & 7)) << 4)) = v[q];
    }
  }
  lds_float *par = (lds_float *)(lds + W_BYTES);
  h2_stage_params(a, n0, LIN_BN, tid, par);
  __syncthreads();                                 // the only barrier of the kernel

  const int64_t nblk = (a.n_out + 31) / 32;
  const int64_t nw = (int64_t)gridDim.x * (LIN_NT / 64);
  int64_t b = (int64_t)blockIdx.x * (LIN_NT / 64) + wave;
  if (b >= nblk) return;
  const char *in_base = reinterpret_cast<const char *>(a.in_split) + h * 16;
  auto load_rows = [&](int64_t blk, LinRows<NCH> &r) {
    const int64_t row = blk * 32 + l31;
    int idx = -1;
    if (row < a.n_out) idx = a.nbr ? a.nbr[row] : (int)row;
    if (idx >= 0) {
      const char *src = in_base + (uint64_t)(uint32_t)idx * rsb;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          r.hi[c][ks] = *reinterpret_cast<const f16x8 *>(src + c * 128 + ks * 32);
          r.lo[c][ks] = *reinterpret_cast<const f16x8 *>(src + c * 128 + 64 + ks * 32);
        }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int q = 0; q < 8; ++q) r.hi[c][ks][q] = (_Float16)0.f, r.lo[c][ks][q] = (_Float16)0.f;
    }
  };

  // weight fragment of (chunk, column block j, k-step, half): column j * 32 + l31, run (half * 4 + ks * 2 + h)
  const int sw = (l31 >> 1) & 7;
  const uint32_t wrow_off = (uint32_t)l31 * 128;
  uint32_t xo[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int part = 0; part < 2; ++part) xo[ks][part] = (uint32_t)(((part * 4 + ks * 2 + h) ^ sw) << 4);

  LinRows<NCH> cur;
  load_rows(b, cur);
  for (; b < nblk; b += nw) {
    f32x16 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
    const char *wrow = lds + wrow_off;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = *reinterpret_cast<const f16x8 *>(wrow + c * (LIN_BN * 128) + j * 4096 + xo[ks][0]);
          bl[j] = *reinterpret_cast<const f16x8 *>(wrow + c * (LIN_BN * 128) + j * 4096 + xo[ks][1]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // weights first: transposed accumulator block (lane = output row); smallest terms first
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], cur.lo[c][ks], acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], cur.hi[c][ks], acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], cur.hi[c][ks], acc[0][j], 0, 0, 0);
        }
      }
    // the next block's rows into the registers the matrix work has just released: in flight under the epilogue
    // (128 channels: 64 registers of rows on top of the epilogue's would spill - requested after it; the CU's other waves cover)
    if (NCH <= 2 && b + nw < nblk) load_rows(b + nw, cur);
    __builtin_amdgcn_sched_barrier(0);
    // the epilogue's column arithmetic and per-channel vectors are the same for every block: left alone, the compiler
    // hoists all of it out of the loop (hundreds of registers, spilled).  Two values it cannot see through keep it inside.
    int n0b = n0;
    uint32_t par_off = W_BYTES;
    asm volatile("" : "+s"(n0b), "+s"(par_off));
    const ParLds plb{(lds_float *)(lds + par_off), n0b, LIN_BN};
    h2_store_tile<1, 4, EMIT, ParLds>(a, acc, b * 32, n0b, 0, 0, h, l31, plb);
    __builtin_amdgcn_sched_barrier(0);
    if (NCH > 2 && b + nw < nblk) load_rows(b + nw, cur);
  }
}

// ph_conv_desc.route & PH_ROUTE_LIN_NEVER sends a launch back to k_conv_dma (tests/test_hip_lin.py: the two are bit-identical).
// Development build: PASCO_CONV_LIN=0 / ph_conv_lin_set for same-process A/B runs (bits: 0 = on, 8.. = workgroups per CU)
#ifdef PH_DEV
static int g_lin_on = [] { const char *e = PH_DEV_ENV("PASCO_CONV_LIN"); return e == nullptr ? 1 : atoi(e); }();
extern "C" void ph_conv_lin_set(int on) { g_lin_on = on; }
#else
constexpr int g_lin_on = 1;
#endif

// Takes k = 1 launches on pre-split operands with 64 or 128 (padded) input channels; -1 = not served.
int ph_conv_lin_try(const ConvArgsH &a_in, hipStream_t st) {
  if (!(g_lin_on & 1) || (a_in.route & PH_ROUTE_LIN_NEVER)) return -1;
  if (a_in.kvol != 1 || a_in.ksplit != 1 || a_in.tile_k != nullptr || a_in.out_rows != nullptr) return -1;
  if (a_in.in_split == nullptr || a_in.w_split == nullptr || (a_in.cpad != 64 && a_in.cpad != 128)) return -1;
  if (a_in.n_out < 1) return -1;
  ConvArgsH a = a_in;
  a.ablate = 0;
  a.n_col_tiles = (a.cout + LIN_BN - 1) / LIN_BN;
  const int64_t nblk = (a.n_out + 31) / 32;
  const bool emit = a.out_split != nullptr;
  // every workgroup stages the weight tile once: as many workgroups as stay resident (2 per CU), fewer for short launches
  const int per_cu = (g_lin_on >> 8) & 15;
  const int wpc = per_cu ? per_cu : 2;          // 206 - 256 registers: two waves per SIMD
  int64_t gx = (256 * (int64_t)wpc) / a.n_col_tiles;
  const int64_t need = (nblk + 3) / 4;
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)a.n_col_tiles);
  if (a.cpad == 64) {
    if (emit) hipLaunchKernelGGL((k_conv_lin<2, true>), grid, dim3(LIN_NT), 0, st, a);
    else hipLaunchKernelGGL((k_conv_lin<2, false>), grid, dim3(LIN_NT), 0, st, a);
  } else {
    if (emit) hipLaunchKernelGGL((k_conv_lin<4, true>), grid, dim3(LIN_NT), 0, st, a);
    else hipLaunchKernelGGL((k_conv_lin<4, false>), grid, dim3(LIN_NT), 0, st, a);
  }
  PH_LAUNCH_CHECK();
  ph_record_cfg(2, 32, LIN_BN, 32, 1, emit ? 1 : 0, 7, 4);
  return 0;
}
