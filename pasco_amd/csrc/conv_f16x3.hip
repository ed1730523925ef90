// Sparse convolution forward, split-precision variant: fp32 operands are split into an f16 "hi" and an
// f16 "lo" part and the product is formed as  hi*hi + hi*lo + lo*hi  on v_mfma_f32_32x32x16_f16 with
// fp32 accumulation (the dropped lo*lo term is 2^-22 relative).  Three f16 MFMAs of K = 16 replace eight
// fp32 MFMAs of K = 2: 5.3x less matrix-pipe time per contraction element at fp32-class accuracy
// (measured against an fp64 reference in tests/test_hip_f16x3.py).
//
// OPT-IN (ph_conv_desc.mma_mode = 1).  Requirements: cin % 8 == 0, cout % 4 == 0, |activations| < 65504
// (f16 range; the weights are pre-scaled by a power of two so that their lo parts stay normal, the
// accumulator is unscaled in the epilogue).  Weights arrive pre-split and pre-transposed
// ([K][cout][cin] f16 hi / lo, prepared once per layer by the host) so that both MFMA operands are
// 16-byte contiguous k-runs in LDS.
//
// Structure = the fp32 kernel's (conv.hip): output-stationary tile, gather prologue, fused epilogue,
// register-staged software pipeline, XCD-aware tile order.
#include <stdio.h>
#include <stdlib.h>

#include "conv_h2_common.h"

template <int BM, int KC, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(HV_THREADS) k_conv_f16x3(ConvArgsH a) {
  constexpr int BN = WN * TN * 32;
  constexpr int HKC = KC;
  constexpr int H_LD = KC + 8;
  constexpr int A_TPR = KC / 4;                          // threads x float4 per row chunk
  constexpr int A_RPP = HV_THREADS / A_TPR;              // rows per pass
  constexpr int A_PASSES = BM / A_RPP;
  constexpr int B_SPR = KC / 8;                          // 16-byte (8 x f16) slots per weight row
  constexpr int B_SLOTS = (BN * B_SPR + HV_THREADS - 1) / HV_THREADS;   // per thread, per hi / lo
  static_assert(WM * WN == 4 && WM * TM * 32 == BM, "tile shape");
  static_assert(B_SLOTS >= 1, "loader shape");

  __shared__ __attribute__((aligned(16))) _Float16 Ah[BM * H_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Al[BM * H_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Bh[BN * H_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Bl[BN * H_LD];

  const int nwg = gridDim.x;
  const int cpx = nwg >> 3;
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * cpx + (bid >> 3);
  const int ntiles = a.n_row_tiles * a.n_col_tiles;
  if (tile >= ntiles) return;
  const int row_tile = tile / a.n_col_tiles;
  const int col_tile = tile - row_tile * a.n_col_tiles;
  const int64_t m0 = (int64_t)row_tile * BM;
  const int n0 = col_tile * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int cin = a.cin, cout = a.cout;
  const int nchunks = (cin + HKC - 1) / HKC;
  // split-K over the kernel offsets (few-row layers: the dense bottleneck's 245 offsets on 6.7 k rows)
  const int kper = (a.kvol + a.ksplit - 1) / a.ksplit;
  const int k_begin = (int)blockIdx.y * kper;
  const int k_end = (k_begin + kper < a.kvol) ? k_begin + kper : a.kvol;
  const int nstages = (k_end > k_begin ? k_end - k_begin : 0) * nchunks;
  const int a_c4 = tid % A_TPR;
  const int a_r0 = tid / A_TPR;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[A_PASSES];
  f16x8 rbh[B_SLOTS], rbl[B_SLOTS];
  bool xbad = false;  // this thread converted an activation outside the f16 range (or a NaN)
  int idx_cur[A_PASSES], idx_nxt[A_PASSES];
  int cur_c0 = 0;

  auto load_idx = [&](int k, int *dst) {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int64_t row = m0 + a_r0 + p * A_RPP;
      int idx = -1;
      if (row < a.n_out) idx = a.nbr ? a.nbr[(int64_t)k * a.n_out + row] : (int)row;
      dst[p] = idx;
    }
  };

  auto load_stage = [&](int k, int c0) {
    cur_c0 = c0;
    const int cbase = c0 + a_c4 * 4;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int idx = idx_cur[p];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx >= 0 && cbase < cin) v = *reinterpret_cast<const float4 *>(a.in + (int64_t)idx * cin + cbase);
      ra[p] = v;
    }
    // weights: thread slot -> (output channel n, 8-channel segment of the 32-channel chunk)
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) {
      const int slot = tid + q * HV_THREADS;
      const int n = slot / B_SPR, seg = slot % B_SPR;
      const int c = c0 + seg * 8;
      f16x8 vh = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
      if (slot < BN * B_SPR && n0 + n < cout && c < cin) {
        const int64_t off = ((int64_t)k * cout + n0 + n) * cin + c;
        vh = *reinterpret_cast<const f16x8 *>(a.w_hi + off);
        vl = *reinterpret_cast<const f16x8 *>(a.w_lo + off);
      }
      rbh[q] = vh;
      rbl[q] = vl;
    }
  };

  auto store_stage = [&](unsigned valid) {
    const int cbase = cur_c0 + a_c4 * 4;
    float ps[4] = {1.f, 1.f, 1.f, 1.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.has_pro) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cbase + j < cin) {
          if (a.pro_scale) ps[j] = a.pro_scale[cbase + j];
          if (a.pro_shift) pb[j] = a.pro_shift[cbase + j];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      float v[4] = {ra[p].x, ra[p].y, ra[p].z, ra[p].w};
      const bool ok = (valid >> p) & 1u;
      f16x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = v[j];
        if (a.has_pro) {   // separate multiply and add, like ph_split_rows (mode 2) and the C restatement
#pragma clang fp contract(off)
          const float m = x * ps[j];
          x = h_act(m + pb[j], a.pro_neg);
        }
        if (!ok || cbase + j >= cin) x = 0.f;
        x *= a.act_pow2;
        xbad |= h_out_of_range(x);
        const _Float16 xh = (_Float16)x;
        hi[j] = xh;
        lo[j] = (_Float16)(x - (float)xh);
      }
      const int o = (a_r0 + p * A_RPP) * H_LD + a_c4 * 4;
      *reinterpret_cast<f16x4 *>(&Ah[o]) = hi;
      *reinterpret_cast<f16x4 *>(&Al[o]) = lo;
    }
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) {
      const int slot = tid + q * HV_THREADS;
      const int n = slot / B_SPR, seg = slot % B_SPR;
      if (slot >= BN * B_SPR) continue;
      *reinterpret_cast<f16x8 *>(&Bh[n * H_LD + seg * 8]) = rbh[q];
      *reinterpret_cast<f16x8 *>(&Bl[n * H_LD + seg * 8]) = rbl[q];
    }
  };

  auto compute_stage = [&]() {
#pragma unroll
    for (int ks = 0; ks < HKC / 16; ++ks) {
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = ((wm * TM + i) * 32 + l31) * H_LD + ks * 16 + h * 8;
        ah[i] = *reinterpret_cast<const f16x8 *>(&Ah[o]);
        al[i] = *reinterpret_cast<const f16x8 *>(&Al[o]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = ((wn * TN + j) * 32 + l31) * H_LD + ks * 16 + h * 8;
        bh[j] = *reinterpret_cast<const f16x8 *>(&Bh[o]);
        bl[j] = *reinterpret_cast<const f16x8 *>(&Bl[o]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  auto valid_bits = [&]() {
    unsigned v = 0;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) v |= (idx_cur[p] >= 0 ? 1u : 0u) << p;
    return v;
  };

  if (nstages > 0) {
    load_idx(k_begin, idx_cur);
    if (k_begin + 1 < k_end) load_idx(k_begin + 1, idx_nxt);
    load_stage(k_begin, 0);
  }
  unsigned valid = valid_bits();
  int k = k_begin, chunk = 0;
  for (int s = 0; s < nstages; ++s) {
    store_stage(valid);
    __syncthreads();
    if (s + 1 < nstages) {
      if (++chunk == nchunks) {
        chunk = 0;
        ++k;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) idx_cur[p] = idx_nxt[p];
        if (k + 1 < k_end) load_idx(k + 1, idx_nxt);
      }
      load_stage(k, chunk * HKC);
      valid = valid_bits();
    }
    compute_stage();
    __syncthreads();
  }

  if (a.status != nullptr && xbad) atomicOr(a.status, 1);

  if (a.ksplit > 1) {   // raw partial sums; k_splitk_epilogue reduces them in a fixed order
    float *part = a.partial + (int64_t)blockIdx.y * a.n_out * cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      if (col >= cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < a.n_out) part[row * cout + col] = acc[i][j][r];
        }
    }
    return;
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + l31;
    if (col >= cout) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
    const float es = a.epi_scale ? a.epi_scale[col] : 1.f;
    const float eb = a.epi_shift ? a.epi_shift[col] : 0.f;
    const float es2 = a.epi2_scale ? a.epi2_scale[col] : 1.f;
    const float eb2 = a.epi2_shift ? a.epi2_shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= a.n_out) continue;
        float v = acc[i][j][r] * a.w_unscale + bias;
        v = h_act(v * es + eb, a.epi_neg);
        if (a.has_tail) {
          v = v * es2 + eb2;
          if (a.residual) v += a.residual[row * cout + col];
          v = h_act(v, a.res_neg);
        }
        a.out[row * cout + col] = v;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------
// Mode 2: both operands arrive pre-split (ph_split_rows) in the layout the LDS tiles use, so the gather is
// a pure 16-byte copy: no per-gather conversion (the in-kernel split of mode 1 spends ~2x the matrix-pipe
// time in VALU converting every row up to 27 times, profiles/README.md "SQ counters, split kernel").
// A row of a tile = KC/32 groups of [32 hi | 32 lo] f16 (+8 pad): LD = 2*KC + 8.
// ------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(128))) const uint32_t ph_zero_line[64] = {0};   // 256 zero bytes for absent neighbours

// EMIT = the launch also writes the next convolution's operand (out_split).  A separate instantiation: the
// emission code raises the kernel's register budget (one wave of occupancy less), which launches that do not
// emit should not pay for.
template <int BM, int KC, int WM, int WN, int TM, int TN, bool EMIT>
__global__ void __launch_bounds__(HV_THREADS) k_conv_h2(ConvArgsH a) {
  constexpr int BN = WN * TN * 32;
  constexpr int LD = 2 * KC + 8;
  constexpr int SPR = KC / 4;                 // 16-byte slots per tile row (hi + lo)
  constexpr int RPP = HV_THREADS / SPR;       // tile rows filled per pass
  constexpr int A_PASSES = BM / RPP;
  constexpr int B_PASSES = BN / RPP;
  static_assert(WM * WN == 4 && WM * TM * 32 == BM, "tile shape");
  static_assert(BM % RPP == 0, "loader shape");

  __shared__ __attribute__((aligned(16))) _Float16 As[BM * LD];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[BN * LD];

  const int nwg = gridDim.x;
  const int cpx = nwg >> 3;
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * cpx + (bid >> 3);
  const int ntiles = a.n_row_tiles * a.n_col_tiles;
  if (tile >= ntiles) return;
  // one of a window / gather pair: the window kernel (conv_win.hip) serves the map when the predicate holds
  if (a.win_gather && ph_win_pred(a.win_stats, a.win_which, (a.n_out + 127) / 128)) return;
  const int row_tile = tile / a.n_col_tiles;
  const int col_tile = tile - row_tile * a.n_col_tiles;
  const int64_t m0 = (int64_t)row_tile * BM;
  const int n0 = col_tile * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int cout = a.cout;
  const int nchunks = a.cpad / KC;
  const int kper = (a.kvol + a.ksplit - 1) / a.ksplit;
  const int k_begin = (int)blockIdx.y * kper;
  const int k_end = (k_begin + kper < a.kvol) ? k_begin + kper : a.kvol;
  const int nstages = (k_end > k_begin ? k_end - k_begin : 0) * nchunks;
  const int l_j = tid % SPR;                  // slot inside the row chunk
  const int l_r = tid / SPR;                  // first tile row of this thread
  const uint32_t rs = 2u * (uint32_t)a.cpad;  // f16 per operand row

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f16x8 ra[A_PASSES], rb[B_PASSES];
  int idx_cur[A_PASSES], idx_nxt[A_PASSES];
  const _Float16 *zero_row = reinterpret_cast<const _Float16 *>(ph_zero_line);
  // weight rows of this thread: n0 + l_r + q * RPP, clamped to the last output channel - columns >= cout of the
  // tile are never stored, so they may multiply any finite row and the loads need no predicate
  static_assert(BN % RPP == 0, "weight loader shape");
  int64_t boff[B_PASSES];
#pragma unroll
  for (int q = 0; q < B_PASSES; ++q) {
    int n = n0 + l_r + q * RPP;
    n = n < cout ? n : cout - 1;
    boff[q] = (int64_t)n * rs + l_j * 8;
  }

  auto load_idx = [&](int k, int *dst) {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int64_t row = m0 + l_r + p * RPP;
      int idx = -1;
      if (row < a.n_out) idx = a.nbr ? a.nbr[(int64_t)k * a.n_out + row] : (int)row;
      dst[p] = idx;
    }
  };

  auto load_stage = [&](int k, int chunk) {
    const int coff = chunk * (2 * KC) + l_j * 8;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      // rows without a neighbour read a zero line instead of being predicated off (no exec juggling per load)
      const int idx = idx_cur[p];
      const _Float16 *src = idx >= 0 ? a.in_split + (uint64_t)(uint32_t)idx * rs + coff : zero_row + l_j * 8;
      ra[p] = *reinterpret_cast<const f16x8 *>(src);
    }
    const _Float16 *wk = a.w_split + (int64_t)k * cout * rs + chunk * (2 * KC);
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) rb[q] = *reinterpret_cast<const f16x8 *>(wk + boff[q]);
  };

  auto store_stage = [&]() {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p)
      *reinterpret_cast<f16x8 *>(&As[(l_r + p * RPP) * LD + l_j * 8]) = ra[p];
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) *reinterpret_cast<f16x8 *>(&Bs[(l_r + q * RPP) * LD + l_j * 8]) = rb[q];
  };

  auto compute_stage = [&]() {
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      const int co = (ks >> 1) * 64 + (ks & 1) * 16 + h * 8;   // hi run; the lo run is 32 further
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = ((wm * TM + i) * 32 + l31) * LD + co;
        ah[i] = *reinterpret_cast<const f16x8 *>(&As[o]);
        al[i] = *reinterpret_cast<const f16x8 *>(&As[o + 32]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = ((wn * TN + j) * 32 + l31) * LD + co;
        bh[j] = *reinterpret_cast<const f16x8 *>(&Bs[o]);
        bl[j] = *reinterpret_cast<const f16x8 *>(&Bs[o + 32]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // weights as the first operand: the accumulator is the TRANSPOSED 32x32 block - lane = output row,
          // registers = 4-channel runs - so the epilogue moves float4s (smallest terms first)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
  };

  if (nstages > 0) {
    load_idx(k_begin, idx_cur);
    if (k_begin + 1 < k_end) load_idx(k_begin + 1, idx_nxt);
    load_stage(k_begin, 0);
  }
  int k = k_begin, chunk = 0;
  for (int s = 0; s < nstages; ++s) {
    store_stage();
    __syncthreads();
    if (s + 1 < nstages) {
      if (++chunk == nchunks) {
        chunk = 0;
        ++k;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) idx_cur[p] = idx_nxt[p];
        if (k + 1 < k_end) load_idx(k + 1, idx_nxt);
      }
      load_stage(k, chunk);
    }
    compute_stage();
    __syncthreads();
  }

  h2_store_tile<TM, TN, EMIT>(a, acc, m0, n0, wm, wn, h, l31);
}

// fp32 rows -> [hi x32 | lo x32] groups; one thread per 8 channels.  Channels >= c (pad to 32) are zero.
constexpr unsigned PH_MAGNITUDE_BLOCKS = 256;
__global__ void __launch_bounds__(256) k_split_rows(const float *__restrict__ in, int64_t n, int c, int cpad,
                                                     const float *__restrict__ ps, const float *__restrict__ pb,
                                                     int has_pro, float neg, float pow2, _Float16 *__restrict__ out,
                                                     int32_t *status) {
  const int segs = cpad >> 3;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * segs) return;
  const int64_t row = t / segs;
  const int c0 = (int)(t - row * segs) * 8;
  f16x8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < c) {   // c % 8 == 0: the segment is entirely inside or entirely padding
    const float4 v0 = *reinterpret_cast<const float4 *>(in + row * c + c0);
    const float4 v1 = *reinterpret_cast<const float4 *>(in + row * c + c0 + 4);
    float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    bool bad = false, big = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = x[j];
      if (has_pro) {   // separate multiply and add (no FMA contraction): bit-identical to the C restatement in oracle/
#pragma clang fp contract(off)
        const float m = v * (ps ? ps[c0 + j] : 1.f);
        v = h_act(m + (pb ? pb[c0 + j] : 0.f), neg);
      }
      v *= pow2;
      bad |= h_out_of_range(v);
      big |= fabsf(v) >= PH_SPLIT_FULL_PRECISION;
      const _Float16 vh = (_Float16)v;
      hi[j] = vh;
      lo[j] = (_Float16)(v - (float)vh);
    }
    if (status != nullptr && bad) atomicOr(status, 1);   // |v| > 65504 or NaN
    // bit 6: the operand holds a value whose lo half is a normal f16 (full 22-bit split).  A tensor WITHOUT one is all tiny (or
    // all zero): its hi / lo pairs carry fewer and fewer bits (lo is subnormal below 2^-3 scaled, hi itself below 2^-14) - the guarded
    // module path (ph_conv_desc.exact_if) sends such a tensor to the exact kernel.  Looked for in the first PH_MAGNITUDE_BLOCKS
    // workgroups only (the first ~8 k rows of a 64-channel tensor): a miss there costs a launch of the exact kernel, never a wrong
    // result, and the rest of the grid pays nothing (a read-before-atomic in every wave measured +20 % on this HBM-bound pass).
    if (status != nullptr && blockIdx.x < PH_MAGNITUDE_BLOCKS) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(big);
      if (m != 0 && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m) &&
          (__atomic_load_n(status, __ATOMIC_RELAXED) & PH_STATUS_MAGNITUDE) == 0)
        atomicOr(status, PH_STATUS_MAGNITUDE);
    }
  }
  _Float16 *dst = out + (row * (cpad >> 5) + (c0 >> 5)) * 64 + (c0 & 31);
  *reinterpret_cast<f16x8 *>(dst) = hi;
  *reinterpret_cast<f16x8 *>(dst + 32) = lo;
}

extern "C" int ph_split_rows(const float *in, int64_t n, int32_t c, const float *pro_scale, const float *pro_shift,
                             int32_t pro_act, float slope, int32_t exp2, void *out_split, int32_t *status,
                             ph_stream_t stream) {
  PH_REQUIRE(n >= 0 && c > 0 && c % 8 == 0, "split_rows: needs c %% 8 == 0 (c=%d)", c);
  PH_REQUIRE(exp2 >= -16 && exp2 <= 16, "split_rows: exp2 out of range (%d)", exp2);
  if (n == 0) return 0;
  PH_REQUIRE(in && out_split, "split_rows: null buffer");
  PH_REQUIRE((((uintptr_t)in | (uintptr_t)out_split) & 15) == 0, "split_rows: 16-byte alignment");
  const int cpad = (c + 31) / 32 * 32;
  const int64_t total = n * (cpad / 8);
  const float neg = pro_act == PH_ACT_RELU ? 0.f : (pro_act == PH_ACT_LEAKY ? slope : 1.f);
  const int has_pro = (pro_scale || pro_shift || pro_act != PH_ACT_NONE) ? 1 : 0;
  hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ph_stream(stream), in, n, c, cpad,
                     pro_scale, pro_shift, has_pro, neg, ldexpf(1.f, exp2), (_Float16 *)out_split, status);
  PH_LAUNCH_CHECK();
  return 0;
}

// out = epilogue( sum_s partial[s] )  - one thread per 4-channel run, splits summed in index order
__global__ void __launch_bounds__(256) k_splitk_epilogue(ConvArgsH a) {
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t total = a.n_out * a.cout;
  if (t >= total) return;
  const int64_t row = t / a.cout;
  const int col = (int)(t - row * a.cout);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < a.ksplit; ++s) {
    const float4 p = *reinterpret_cast<const float4 *>(a.partial + (int64_t)s * total + t);
    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
  }
  const float av[4] = {acc.x, acc.y, acc.z, acc.w};
  float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.has_tail && a.residual) rs = *reinterpret_cast<const float4 *>(a.residual + t);
  if (a.has_tail && a.axis_table) {
    int64_t aoff[3];
    if (ph_axis_offsets(a, row, aoff) && a.status != nullptr) atomicOr(a.status, 4);
    const float4 tb = ph_axis_residual4(a, aoff, col);
    rs = make_float4(tb.x + rs.x, tb.y + rs.y, tb.z + rs.z, tb.w + rs.w);
  }
  const float r4[4] = {rs.x, rs.y, rs.z, rs.w};
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = col + q;
    float x = av[q] * a.w_unscale + (a.bias ? a.bias[c] : 0.f);
    x = h_act(x * (a.epi_scale ? a.epi_scale[c] : 1.f) + (a.epi_shift ? a.epi_shift[c] : 0.f), a.epi_neg);
    if (a.has_tail) {
      x = x * (a.epi2_scale ? a.epi2_scale[c] : 1.f) + (a.epi2_shift ? a.epi2_shift[c] : 0.f);
      x = h_act(x + r4[q], a.res_neg);
    }
    v[q] = x;
  }
  if (a.out) *reinterpret_cast<float4 *>(a.out + t) = make_float4(v[0], v[1], v[2], v[3]);
  if (a.out_split) {
    float sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sc[q] = a.osp_scale ? a.osp_scale[col + q] : 1.f;
      sh[q] = a.osp_shift ? a.osp_shift[col + q] : 0.f;
    }
    const bool obad = emit_split4(v, sc, sh, a.osp_has, a.osp_neg, a.act_pow2,
                                  a.out_split + (row * (a.cout >> 5) + (col >> 5)) * 64 + (col & 31));
    if (a.status != nullptr && obad) atomicOr(a.status, 1);
  }
}

int ph_launch_splitk_epilogue(const ConvArgsH &args, hipStream_t st) {
  const int64_t total = args.n_out * args.cout;
  hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, args);
  PH_LAUNCH_CHECK();
  return 0;
}

template <int BM, int KC, int WM, int WN, int TM, int TN>
static int launch_h(const ConvArgsH &a, hipStream_t st) {
  constexpr int BN = WN * TN * 32;
  ConvArgsH args = a;
  args.n_row_tiles = (int)((a.n_out + BM - 1) / BM);
  args.n_col_tiles = (a.cout + BN - 1) / BN;
  const int ntiles = args.n_row_tiles * args.n_col_tiles;
  const int grid = ((ntiles + 7) / 8) * 8;
  hipLaunchKernelGGL((k_conv_f16x3<BM, KC, WM, WN, TM, TN>), dim3(grid, args.ksplit), dim3(HV_THREADS), 0, st, args);
  PH_LAUNCH_CHECK();
  if (args.ksplit > 1) {
    const int64_t total = a.n_out * a.cout;
    hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, args);
    PH_LAUNCH_CHECK();
  }
  ph_record_cfg(1, BM, BN, KC, args.ksplit, 0, 1, 4);
  return 0;
}

template <int BM, int KC, int WM, int WN, int TM, int TN>
static int launch_h2(const ConvArgsH &a, hipStream_t st) {
  constexpr int BN = WN * TN * 32;
  ConvArgsH args = a;
  args.n_row_tiles = (int)((a.n_out + BM - 1) / BM);
  args.n_col_tiles = (a.cout + BN - 1) / BN;
  const int ntiles = args.n_row_tiles * args.n_col_tiles;
  const int grid = ((ntiles + 7) / 8) * 8;
  if (args.out_split != nullptr && args.ksplit == 1)
    hipLaunchKernelGGL((k_conv_h2<BM, KC, WM, WN, TM, TN, true>), dim3(grid, 1), dim3(HV_THREADS), 0, st, args);
  else
    hipLaunchKernelGGL((k_conv_h2<BM, KC, WM, WN, TM, TN, false>), dim3(grid, args.ksplit), dim3(HV_THREADS), 0, st, args);
  PH_LAUNCH_CHECK();
  if (args.ksplit > 1) {
    const int64_t total = a.n_out * a.cout;
    hipLaunchKernelGGL(k_splitk_epilogue, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, args);
    PH_LAUNCH_CHECK();
  }
  ph_record_cfg(2, BM, BN, KC, args.ksplit, (args.out_split != nullptr && args.ksplit == 1) ? 1 : 0, 2, 4);
  return 0;
}

// Tuning overrides (development only), read ONCE when the library first launches a split convolution:
//   PASCO_CONVH_MID=0      disable the 64-row / 64-channel-stage choice for thin 128-channel layers
//   PASCO_CONVH_CFG=bm,kc  force tile height / stage depth where the shape allows it
//   PASCO_CONVH_KSPLIT=n   force the split over the kernel offsets
struct ConvHKnobs {
  bool mid_on = true;
  bool has_cfg = false;
  int cfg_bm = 0, cfg_kc = 0;
  bool has_ksplit = false;
  int ksplit = 0;
  bool dma_all = false;
  bool win_on = true;      // PASCO_CONV_WIN=0: no LDS-window kernel
  bool win_wide = false;   // PASCO_CONV_WIN=2: also on 128-wide tiles
  bool rl_on = true;       // PASCO_CONV_RL=0: ignore row lists (walk all offsets of one-pair maps)
  bool dma_on = true;      // PASCO_CONV_DMA=0: keep every launch on the register-staged k_conv_h2 (A/B comparisons)
  bool wide_on = true;     // PASCO_CONV_WIDE=0: no 256 x 256 tiles (conv_wide.hip) for the 256-output-channel gather launches
  ConvHKnobs() {
    if (const char *e = PH_DEV_ENV("PASCO_CONV_DMA")) {
      dma_on = atoi(e) != 0;
      dma_all = atoi(e) == 2;      // 2: every tile width on the DMA kernel
    }
    if (const char *e = PH_DEV_ENV("PASCO_CONV_WIDE")) wide_on = atoi(e) != 0;
    if (const char *e = PH_DEV_ENV("PASCO_CONVH_MID")) mid_on = atoi(e) != 0;
    if (const char *e = PH_DEV_ENV("PASCO_CONV_WIN")) {
      win_on = atoi(e) != 0;
      win_wide = atoi(e) == 2;
    }
    if (const char *e = PH_DEV_ENV("PASCO_CONV_RL")) rl_on = atoi(e) != 0;
    if (const char *e = PH_DEV_ENV("PASCO_CONVH_CFG")) has_cfg = sscanf(e, "%d,%d", &cfg_bm, &cfg_kc) >= 1;
    if (const char *e = PH_DEV_ENV("PASCO_CONVH_KSPLIT")) {
      has_ksplit = true;
      ksplit = atoi(e);
    }
  }
};
static const ConvHKnobs &convh_knobs() {
  static const ConvHKnobs k;
  return k;
}

// called from ph_conv_fwd (conv.hip) when desc->mma_mode is 1 or 2
int ph_conv_fwd_f16x3(const ph_conv_desc *d, hipStream_t st) {
  const bool pre = d->mma_mode == 2;
  PH_REQUIRE(d->cin % 8 == 0 && d->cout % 4 == 0, "conv_fwd(f16x3): needs cin %% 8 == 0 and cout %% 4 == 0");
  if (pre) {
    PH_REQUIRE(d->in_split && d->w_split, "conv_fwd(f16x3, mode 2): pre-split operands missing");
    PH_REQUIRE((((uintptr_t)d->in_split | (uintptr_t)d->w_split) & 15) == 0, "conv_fwd(f16x3): 16-byte alignment");
    PH_REQUIRE(d->n_in < ((int64_t)1 << 31), "conv_fwd(f16x3): n_in too large");
    PH_REQUIRE((((uintptr_t)d->out | (uintptr_t)d->residual | (uintptr_t)d->splitk_ws) & 15) == 0,
               "conv_fwd(f16x3, mode 2): out / residual / splitk_ws must be 16-byte aligned");
  } else {
    PH_REQUIRE(d->w_f16_hi && d->w_f16_lo, "conv_fwd(f16x3): pre-split weights missing");
    PH_REQUIRE((((uintptr_t)d->in | (uintptr_t)d->w_f16_hi | (uintptr_t)d->w_f16_lo) & 15) == 0,
               "conv_fwd(f16x3): 16-byte alignment");
  }
  ConvArgsH a;
  a.in_split = (const _Float16 *)d->in_split;
  a.w_split = (const _Float16 *)d->w_split;
  a.w_frag = (const _Float16 *)d->w_frag;
  a.cpad = (d->cin + 31) / 32 * 32;
  a.out_split = (_Float16 *)d->out_split;
  a.osp_scale = d->osp_scale;
  a.osp_shift = d->osp_shift;
  a.osp_has = (d->osp_scale || d->osp_shift || d->osp_act != PH_ACT_NONE) ? 1 : 0;
  a.osp_neg = d->osp_act == PH_ACT_RELU ? 0.f : (d->osp_act == PH_ACT_LEAKY ? d->epi_slope : 1.f);
  PH_REQUIRE(d->split_exp2 >= -16 && d->split_exp2 <= 16, "conv_fwd(f16x3): split_exp2 out of range (%d)", d->split_exp2);
  a.act_pow2 = ldexpf(1.f, d->split_exp2);
  {   // the epilogue reads the per-channel vectors as float4 when all of them allow it
    uintptr_t bits = 0;
    const float *vecs[] = {d->bias, d->epi_scale, d->epi_shift, d->epi2_scale, d->epi2_shift, d->osp_scale, d->osp_shift};
    for (const float *v : vecs) bits |= (uintptr_t)v;
    a.par_vec = ((bits & 15) == 0 && d->cout % 4 == 0) ? 1 : 0;
  }
  if (d->out_split) {
    PH_REQUIRE(pre, "conv_fwd(f16x3): out_split needs mma_mode 2");
    PH_REQUIRE(d->cout % 32 == 0 && (((uintptr_t)d->out_split) & 15) == 0, "conv_fwd(f16x3): out_split needs cout %% 32 == 0");
  }
  PH_REQUIRE(d->out || d->out_split, "conv_fwd(f16x3): no output buffer");
  a.in = d->in;
  a.w_hi = (const _Float16 *)d->w_f16_hi;
  a.w_lo = (const _Float16 *)d->w_f16_lo;
  a.nbr = d->nbr;
  a.out = d->out;
  a.n_in = d->n_in;
  a.n_out = d->n_out;
  a.cin = d->cin;
  a.cout = d->cout;
  a.kvol = d->kvol;
  a.pro_scale = d->pro_scale;
  a.pro_shift = d->pro_shift;
  a.bias = d->bias;
  a.epi_scale = d->epi_scale;
  a.epi_shift = d->epi_shift;
  a.epi2_scale = d->epi2_scale;
  a.epi2_shift = d->epi2_shift;
  a.residual = d->residual;
  auto neg_of = [&](int act) { return act == PH_ACT_RELU ? 0.f : (act == PH_ACT_LEAKY ? d->epi_slope : 1.f); };
  a.pro_neg = neg_of(d->pro_act);
  a.epi_neg = neg_of(d->epi_act);
  a.res_neg = neg_of(d->res_act);
  a.w_unscale = d->w_unscale;
  a.has_pro = (d->pro_scale || d->pro_shift || d->pro_act != PH_ACT_NONE) ? 1 : 0;   // mode 2: already applied by ph_split_rows
  a.has_tail = (d->residual || d->epi2_scale || d->epi2_shift || d->res_act != PH_ACT_NONE || d->axis_table) ? 1 : 0;
  a.axis_table = d->axis_table;
  a.axis_coords = d->axis_coords;
  a.axis_lo = d->axis_lo;
  a.axis_rows = d->axis_rows;
  if (d->axis_table) {
    PH_REQUIRE(pre, "conv_fwd: axis_table needs mma_mode 2");
    PH_REQUIRE(d->axis_coords && d->axis_rows > 0 && d->cout % 4 == 0 && (((uintptr_t)d->axis_table | (uintptr_t)d->axis_coords) & 15) == 0,
               "conv_fwd: axis_table needs axis_coords, axis_rows > 0, 16-byte alignment");
  }
  a.n_row_tiles = a.n_col_tiles = 0;
  a.status = d->status;
  a.zero = nullptr;
  a.ablate = 0;
  a.route = d->route;
  a.out_rows = nullptr;
  a.tile_k = nullptr;
  a.nbr_stride = d->n_out;
  for (int q = 0; q < 4; ++q) a.gdim[q] = d->grid_dims[q];
  for (int q = 0; q < 3; ++q) a.gker[q] = d->grid_kernel[q];
  a.grid_upw = 0;
  a.tail_ws = d->splitk_ws;
  a.tail_ws_bytes = d->splitk_ws_bytes;
  a.win_rows = d->win_rows;
  a.win_cnt = d->win_cnt;
  a.win_slots = d->win_slots;
  a.win_stats = nullptr;
  a.win_which = 0;
  a.win_gather = 0;
  const int bn = d->cout <= 32 ? 32 : (d->cout <= 64 ? 64 : 128);
  const int64_t ncol = (d->cout + bn - 1) / bn;
  // tile height: the tallest tile that still gives >= 2 workgroups per CU (profiles/r1e_op_bench.json)
  int bm = 128;
  if (bn >= 64 && ((d->n_out + 127) / 128) * ncol < 2 * 256) bm = 64;
  if (bn == 128 && bm == 64 && ((d->n_out + 63) / 64) * ncol < 2 * 256) bm = 32;
  int kc = (d->cin % 64 == 0 && d->cin >= 256) ? 64 : 32;   // deeper stages pay only for wide layers (profiles/r1f_op_bench.json)
  // 128-channel layers with fewer than ~1.5 waves of 128-row tiles (stride-2 levels of the pruned scene): 64-row
  // tiles with 64-channel stages measured 304 -> 250 us (profiles/README.md, r1k sweep); env PASCO_CONVH_MID=0 disables
  const ConvHKnobs &knobs = convh_knobs();
  if (pre && knobs.mid_on && bn == 128 && d->cin == 128 && d->kvol > 1 && ((d->n_out + 127) / 128) * ncol < 3 * 256) {
    if (bm == 128) bm = 64;
    kc = 64;
  }
  if (pre && a.cpad % 64 != 0) kc = 32;
  const bool env = knobs.has_cfg;
  if (env) {
    const int em = knobs.cfg_bm, ek = knobs.cfg_kc;
    if (em == 128 || (em == 64 && bn >= 64) || (em == 32 && bn == 128)) bm = em;
    if (ek == 32 || (ek == 64 && (!pre || a.cpad % 64 == 0))) kc = ek;
  }
  // Few-row layers (dense bottleneck: 245 offsets on 6.7 k rows; stride-4/8 layers): every workgroup streams
  // the whole W[k] slab sequence from L2, so small row tiles multiply the weight traffic.  Keep the tall
  // 128-row tile and get the parallelism from a split over the kernel offsets instead (partial sums reduced
  // in a fixed order by k_splitk_epilogue).  The caller provides the scratch (desc->splitk_ws), else no split.
  a.ksplit = 1;
  a.partial = nullptr;
  {
    const bool se = knobs.has_ksplit;
    const int64_t t128 = ((d->n_out + 127) / 128) * ncol;
    int want = (int)(2048 / (t128 > 0 ? t128 : 1));
    if (want > 8) want = 8;
    if (want > d->kvol / 4) want = d->kvol / 4;
    // round 3: the number of slices by a cost model instead (128-wide tiles; both operand modes, so that they keep choosing alike): the launch runs in
    // ceil(tiles x slices / 512) rounds of resident workgroups, a round lasts as long as its longest slice, and every slice
    // adds a partial-sum round trip - e.g. 146 tiles x 27 offsets: 6 slices = 2 rounds of 5 offsets, 3 slices = ONE round of 9
    // (profiles/r3z_layer_ab_ksplit_model.txt).  PASCO_CONVH_KSPLIT_MODEL=0: the rule above.
    static const bool model_on = [] { const char *e = PH_DEV_ENV("PASCO_CONVH_KSPLIT_MODEL"); return e == nullptr || atoi(e) != 0; }();
    if (model_on && bn == 128 && t128 < 2 * 256 && d->kvol >= 8 && d->splitk_ws != nullptr) {
      const double stage_us = 1.3, fixed_us = 5.0;           // one 32-channel stage of a workgroup sharing its CU; launch / prologue
      const double nchunks = (double)(a.cpad / 32);
      int best = 1;
      double best_us = 1e30;
      for (int ks = 1; ks <= 12 && ks <= d->kvol / 3; ++ks) {
        const int kper = (d->kvol + ks - 1) / ks;
        if (kper > (d->kvol > 100 ? 31 : 32) || (d->kvol + kper - 1) / kper != ks) continue;   // beyond the index table (the dense
                                                       // GEMMs keep its last slot for their stage list, conv_dma.hip) / same as fewer slices
        if (ks > 1 && d->splitk_ws_bytes < (int64_t)ks * d->n_out * d->cout * 4) continue;
        const double rounds = (double)((t128 * ks + 511) / 512);
        double us = rounds * (kper * nchunks * stage_us + fixed_us);
        if (ks > 1) us += (double)ks * (double)d->n_out * d->cout * 4.0 / 3.0e6 + fixed_us;   // the reduction: 3 TB/s
        if (us < best_us) best_us = us, best = ks;
      }
      want = best;
    }
    if (se) want = knobs.ksplit;
    const bool room = d->splitk_ws != nullptr && d->splitk_ws_bytes >= (int64_t)want * d->n_out * d->cout * 4;
    if (bn == 128 && t128 < 2 * 256 && want >= 2 && room && !env) {
      bm = 128;
      a.ksplit = want;
      a.partial = (float *)d->splitk_ws;
    } else if (se && want >= 2 && room) {
      a.ksplit = want;
      a.partial = (float *)d->splitk_ws;
    }
  }
  // the dense-grid promise (the bottleneck's implicit GEMMs): windows of the input instead of gathers, its own split over the units
  if (pre && !env && d->grid_dims[0] > 0 && d->nbr != nullptr) {
    const int rc = ph_conv_grid_try(a, st);
    if (rc >= 0) return rc;
  }
  // one-pair-per-row maps with row lists (generative transposed convolutions): every 128-position tile of the list is the
  // k = 1 product of one kernel offset; outputs go to the rows the list names
  if (pre && !env && d->rl_in && d->rl_out && d->rl_tile_k && d->rl_rows > 0 && knobs.rl_on && (bn == 64 || bn == 128)) {
    ConvArgsH b = a;
    b.nbr = d->rl_in;
    b.out_rows = d->rl_out;
    b.tile_k = d->rl_tile_k;
    b.n_out = d->rl_rows;
    b.ksplit = 1;
    b.partial = nullptr;
    b.win_stats = nullptr;
    b.win_gather = 0;
    const int rc = ph_conv_dma_try(b, bn, st);
    if (rc == 0) ph_record_cfg(2, 128, bn, 32, 1, a.out_split != nullptr ? 1 : 0, 3, 4);
    if (rc >= 0) return rc;
  }
  // 3x3x3 convolutions on big maps with window tables: launch the window kernel AND the gather kernel below; the
  // device-side predicate (window passes per tile, ph_win_build) lets exactly one of them do the work.  Only where no
  // split over the offsets was chosen (its reduction kernel would run unconditionally).
  bool win_pair = false;
  // Measured (profiles/README.md, round 2): 64-wide tiles 638 -> 540 us on the 683 k-row map (two workgroups per CU);
  // 128-wide tiles lose to the gather kernel (one workgroup per CU: 668 vs 567 us) and stay there unless
  // PASCO_CONV_WIN=2.
  // (a 256-wide window workgroup for 256-channel layers was built and measured in round 2: 568 vs 547 us for the gather
  // kernel - removed in round 5, profiles/README.md)
  if (pre && knobs.win_on && !env && d->kvol == 27 &&
      (bn == 64 || (bn == 128 && (knobs.win_wide || (d->route & PH_ROUTE_WIN_ALWAYS)))) &&
      a.ksplit == 1 &&
      d->win_rows && d->win_cnt && d->win_slots && d->win_stats) {
    a.win_stats = d->win_stats;
    a.win_which = (bn == 64 ? 0 : 1) | ph_win_force_bits(d->route);
    a.win_gather = 0;
    if (int rc = ph_conv_win_launch(a, bn, st)) return rc;
    a.win_gather = 1;
    win_pair = true;
    if (bm != 128) bm = 128;   // pairs are only formed on big maps: the gather side keeps its 128-row tile
  }
  auto gather_launch = [&]() -> int {
  // default for pre-split operands: the LDS-DMA pipelined kernel (conv_dma.hip; 128-row tiles, parallelism of few-row
    // layers from the split over the offsets chosen above); it declines slices of more than 32 offsets
    // measured (profiles/README.md, round 2): 6-9 % faster than k_conv_h2 on 128-channel-wide tiles, a few % slower on 64-wide
    if (pre && knobs.wide_on && !env && bn == 128 && (d->cout == 256 || d->cout == 128) && d->kvol >= 8 && d->nbr != nullptr) {
      const int rc = ph_conv_wide_try(a, st);
      if (rc >= 0) return rc;
    }
    // k = 1 streams (linear layers, 1x1x1 convolutions) of 64 / 128 input channels onto 128-wide tiles: the row-stream kernel
    if (pre && knobs.dma_on && !env && bn == 128 && d->kvol == 1 && a.ksplit == 1) {
      const int rc = ph_conv_lin_try(a, st);
      if (rc >= 0) return rc;
    }
    if (pre && knobs.dma_on && !env && (bn == 128 || knobs.dma_all)) {
      ConvArgsH b = a;
      if (bm != 128) {        // the choice above was made for a shorter tile: redo the split decision for 128 rows
        const int64_t t128 = ((d->n_out + 127) / 128) * ncol;
        int want = (int)(1024 / (t128 > 0 ? t128 : 1));
        if (want > 8) want = 8;
        if (want > d->kvol / 4) want = d->kvol / 4;
        const bool room = d->splitk_ws != nullptr && d->splitk_ws_bytes >= (int64_t)want * d->n_out * d->cout * 4;
        if (t128 < 256 && want >= 2 && room) {
          b.ksplit = want;
          b.partial = (float *)d->splitk_ws;
        }
      }
      const int rc = ph_conv_dma_try(b, bn, st);
      if (rc >= 0) return rc;
    }
#define PH_H_CASE(BM_, WM_, WN_, TM_, TN_)                                                                \
    if (bm == BM_) {                                                                                        \
      if (pre) return kc == 64 ? launch_h2<BM_, 64, WM_, WN_, TM_, TN_>(a, st) : launch_h2<BM_, 32, WM_, WN_, TM_, TN_>(a, st); \
      return kc == 64 ? launch_h<BM_, 64, WM_, WN_, TM_, TN_>(a, st) : launch_h<BM_, 32, WM_, WN_, TM_, TN_>(a, st);            \
    }
    if (bn == 32) {
      PH_H_CASE(128, 4, 1, 1, 1);
    } else if (bn == 64) {
      PH_H_CASE(128, 4, 1, 1, 2);
      PH_H_CASE(64, 2, 2, 1, 1);
    } else {
      PH_H_CASE(128, 2, 2, 2, 2);
      PH_H_CASE(64, 2, 2, 1, 2);
      PH_H_CASE(32, 1, 4, 1, 1);
    }
#undef PH_H_CASE
    ph_set_error("conv_fwd(f16x3): no kernel for bm=%d bn=%d", bm, bn);
    return 1;
  };
  const int rc = gather_launch();
  if (rc == 0 && win_pair)   // a window / gather pair: which one worked is the device's decision
    ph_record_cfg(2, 128, bn, 32, 1, a.out_split != nullptr ? 1 : 0, 5, 4);
  return rc;
}
