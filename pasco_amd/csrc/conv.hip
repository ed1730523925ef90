// Sparse convolution forward for gfx950: output-stationary gather -> LDS tile -> fp32 MFMA.
//
//   out[o,:] = epi( sum_k [nbr[k][o] >= 0] * pro(in[nbr[k][o],:]) @ W[k] + bias )   (pasco_hip.h)
//
// One workgroup (256 threads = 4 wave64) owns BM (128/64/32) output rows x BN output channels and walks
// the (kernel offset, input-channel chunk) stages.  Per stage the BM neighbour rows are gathered
// with 16-byte coalesced loads (a 32/64-float chunk of a row = 8/16 lanes x float4) into an LDS A tile,
// the matching W[k] slab goes into an LDS B tile, and the waves run v_mfma_f32_32x32x2_f32
// (exact fp32, SURVEY.md 8(d): C >= 128 layers are FLOP-bound in fp32, C = 64 sits at the ridge).
// The output tile lives in accumulator registers for the whole kernel-offset loop and is written
// exactly once -- no scatter-add, no atomics, HBM traffic = B_alg of SURVEY.md 8(d).
// BN/ReLU of the producing layer is fused as a gather prologue (valid rows only), bias / BN /
// activation / residual as the epilogue, so ResidualBlocks (mink.py:618-658) need no elementwise
// passes.
//
// LDS layouts (bank maths in MI355X_MICROARCH.md "LDS"):
//   As[BM][BKC + 4]  row-major; a lane reads its 4 k-values with one ds_read_b128; row stride of
//                    36 / 68 dwords makes any 16 rows with distinct (row mod 16) hit distinct 16-byte
//                    slots -> conflict free for the b128 lane groups.
//   Bs[BKC][BN + 4]  k-major; lanes 0..31 read 32 consecutive dwords (ds_read_b32).
// The contraction index inside an 8-wide step is permuted (hardware k-half h <-> k = 4h + s) so
// that A needs one wide read per 4 MFMAs; A and B use the same permutation.
#include <stdlib.h>

#include "ph_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CV_THREADS = 256;

struct ConvArgs {
  const float *in;
  const float *w;
  const int32_t *nbr;
  float *out;
  int64_t n_in, n_out;
  int cin, cout, kvol;
  int pro_act;
  const float *pro_scale, *pro_shift, *bias, *epi_scale, *epi_shift, *epi2_scale, *epi2_shift, *residual;
  int epi_act, res_act;
  float slope;
  int n_row_tiles, n_col_tiles;
  const int32_t *pred;     // ph_conv_desc.exact_if: work only when (*pred & 1) != 0 or (*pred & PH_STATUS_MAGNITUDE) == 0
};

// activations: neg = 1 (none), 0 (ReLU), slope (leaky); NaN-preserving select (ph_common.h)
__device__ __forceinline__ float act_neg_of(int act, float slope) {
  return act == PH_ACT_RELU ? 0.f : (act == PH_ACT_LEAKY ? slope : 1.f);
}
__device__ __forceinline__ float act_apply(float v, float neg) { return ph_act(v, neg); }

thread_local ph_conv_cfg_rec ph_last_cfg = {{-1, 0, 0, 0, 0, 0, -1, 0}};

extern "C" int ph_conv_last_config(int32_t *h_out8) {
  PH_REQUIRE(h_out8 != nullptr, "conv_last_config: null buffer");
  for (int i = 0; i < 8; ++i) h_out8[i] = ph_last_cfg.v[i];
  return 0;
}

// Tile = BM output rows x BN output channels, BKC input channels per LDS stage.
// WM x WN waves (4 in total), each TM x TN MFMA tiles of 32x32:  BM = WM*TM*32, BN = WN*TN*32.
template <int BM, int BKC, int WM, int WN, int TM, int TN, bool VEC_A, bool VEC_B>
__global__ void __launch_bounds__(CV_THREADS) k_conv_mfma(ConvArgs a) {
  constexpr int BN = WN * TN * 32;
  constexpr int A_LD = BKC + 4;
  constexpr int B_LD = BN + 4;
  constexpr int A_TPR = BKC / 4;                 // threads (float4 slots) per gathered row chunk
  constexpr int A_RPP = CV_THREADS / A_TPR;      // rows per pass
  constexpr int A_PASSES = BM / A_RPP;           // float4 slots per thread
  constexpr int B_SLOTS = BKC * (BN / 4) / CV_THREADS;
  static_assert(WM * WN == 4 && WM * TM * 32 == BM, "tile shape");
  static_assert(A_PASSES >= 1 && B_SLOTS >= 1, "loader shape");

  __shared__ __attribute__((aligned(16))) float As[BM * A_LD];
  __shared__ __attribute__((aligned(16))) float Bs[BKC * B_LD];

  if (a.pred != nullptr) {      // guarded launch (ph_conv_desc.exact_if): nothing to redo unless the split operand overflowed or was all tiny
    const int w = *a.pred;
    if ((w & 1) == 0 && (w & PH_STATUS_MAGNITUDE) != 0) return;
  }

  // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run of
  // tiles so that neighbouring row tiles (shared gathered rows) meet in one L2.
  const int nwg = gridDim.x;
  const int cpx = nwg >> 3;  // grid is padded to a multiple of 8
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
  const int h = lane >> 5;   // MFMA k-half
  const int l31 = lane & 31;

  const int cin = a.cin, cout = a.cout;
  const float pro_neg = act_neg_of(a.pro_act, a.slope);
  const float epi_neg = act_neg_of(a.epi_act, a.slope);
  const float res_neg = act_neg_of(a.res_act, a.slope);
  const int nchunks = (cin + BKC - 1) / BKC;
  const int nstages = a.kvol * nchunks;

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
  float4 rb[B_SLOTS];
  int idx_cur[A_PASSES], idx_nxt[A_PASSES];
  int cur_c0 = 0;

  // neighbour rows of this thread's A slots for kernel offset k (prefetched one offset ahead so
  // that the dependent row loads never wait for an index load)
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
      if (idx >= 0) {
        const float *src = a.in + (int64_t)idx * cin + cbase;
        if (VEC_A) {
          if (cbase < cin) v = *reinterpret_cast<const float4 *>(src);
        } else {
          if (cbase + 0 < cin) v.x = src[0];
          if (cbase + 1 < cin) v.y = src[1];
          if (cbase + 2 < cin) v.z = src[2];
          if (cbase + 3 < cin) v.w = src[3];
        }
      }
      ra[p] = v;
    }
    const float *wk = a.w + (int64_t)k * cin * cout;
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) {
      const int slot = tid + q * CV_THREADS;
      const int kr = slot / (BN / 4);
      const int n4 = slot - kr * (BN / 4);
      const int c = c0 + kr;
      const int n = n0 + n4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < cin) {
        const float *src = wk + (int64_t)c * cout + n;
        if (VEC_B) {
          if (n < cout) v = *reinterpret_cast<const float4 *>(src);
        } else {
          if (n + 0 < cout) v.x = src[0];
          if (n + 1 < cout) v.y = src[1];
          if (n + 2 < cout) v.z = src[2];
          if (n + 3 < cout) v.w = src[3];
        }
      }
      rb[q] = v;
    }
  };

  const bool has_pro = (a.pro_scale != nullptr) || (a.pro_shift != nullptr) || a.pro_act != PH_ACT_NONE;

  // registers -> LDS, applying the gather prologue (BN affine + activation) to valid rows only.
  // `valid` holds one bit per pass for the stage whose data sits in ra[].
  auto store_stage = [&](unsigned valid) {
    const int cbase = cur_c0 + a_c4 * 4;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_pro) {
      float s4[4] = {1.f, 1.f, 1.f, 1.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cbase + j < cin) {
          if (a.pro_scale) s4[j] = a.pro_scale[cbase + j];
          if (a.pro_shift) b4[j] = a.pro_shift[cbase + j];
        }
      }
      ps = make_float4(s4[0], s4[1], s4[2], s4[3]);
      pb = make_float4(b4[0], b4[1], b4[2], b4[3]);
    }
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      float4 v = ra[p];
      if (has_pro && ((valid >> p) & 1u)) {
        v.x = act_apply(v.x * ps.x + pb.x, pro_neg);
        v.y = act_apply(v.y * ps.y + pb.y, pro_neg);
        v.z = act_apply(v.z * ps.z + pb.z, pro_neg);
        v.w = act_apply(v.w * ps.w + pb.w, pro_neg);
        if (cbase + 0 >= cin) v.x = 0.f;  // channels beyond cin must stay zero
        if (cbase + 1 >= cin) v.y = 0.f;
        if (cbase + 2 >= cin) v.z = 0.f;
        if (cbase + 3 >= cin) v.w = 0.f;
      }
      *reinterpret_cast<float4 *>(&As[(a_r0 + p * A_RPP) * A_LD + a_c4 * 4]) = v;
    }
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) {
      const int slot = tid + q * CV_THREADS;
      const int kr = slot / (BN / 4);
      const int n4 = slot - kr * (BN / 4);
      *reinterpret_cast<float4 *>(&Bs[kr * B_LD + n4 * 4]) = rb[q];
    }
  };

  auto compute_stage = [&]() {
#pragma unroll
    for (int k8 = 0; k8 < BKC / 8; ++k8) {
      float4 av[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + l31;
        av[i] = *reinterpret_cast<const float4 *>(&As[row * A_LD + k8 * 8 + h * 4]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float bv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[j] = Bs[(k8 * 8 + h * 4 + s) * B_LD + (wn * TN + j) * 32 + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float aval = (s == 0) ? av[i].x : (s == 1) ? av[i].y : (s == 2) ? av[i].z : av[i].w;
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bv[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  auto valid_bits = [&]() {
    unsigned v = 0;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) v |= (idx_cur[p] >= 0 ? 1u : 0u) << p;
    return v;
  };

  load_idx(0, idx_cur);
  if (a.kvol > 1) load_idx(1, idx_nxt);
  load_stage(0, 0);
  unsigned valid = valid_bits();
  int k = 0, chunk = 0;
  for (int s = 0; s < nstages; ++s) {
    store_stage(valid);
    __syncthreads();
    if (s + 1 < nstages) {
      if (++chunk == nchunks) {
        chunk = 0;
        ++k;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) idx_cur[p] = idx_nxt[p];
        if (k + 1 < a.kvol) load_idx(k + 1, idx_nxt);
      }
      load_stage(k, chunk * BKC);
      valid = valid_bits();
    }
    compute_stage();
    __syncthreads();
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + l31;
    if (col >= cout) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
    const float es = a.epi_scale ? a.epi_scale[col] : 1.f;
    const float eb = a.epi_shift ? a.epi_shift[col] : 0.f;
    const float es2 = a.epi2_scale ? a.epi2_scale[col] : 1.f;
    const float eb2 = a.epi2_shift ? a.epi2_shift[col] : 0.f;
    const bool tail = a.residual || a.epi2_scale || a.epi2_shift || a.res_act != PH_ACT_NONE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= a.n_out) continue;
        float v = acc[i][j][r] + bias;
        v = act_apply(v * es + eb, epi_neg);
        if (tail) {
          v = v * es2 + eb2;
          if (a.residual) v += a.residual[row * cout + col];
          v = act_apply(v, res_neg);
        }
        a.out[row * cout + col] = v;
      }
    }
  }
}

template <int BM, int BKC, int WM, int WN, int TM, int TN>
static int launch_conv(const ConvArgs &a, hipStream_t st) {
  constexpr int BN = WN * TN * 32;
  ConvArgs args = a;
  args.n_row_tiles = (int)((a.n_out + BM - 1) / BM);
  args.n_col_tiles = (a.cout + BN - 1) / BN;
  const int ntiles = args.n_row_tiles * args.n_col_tiles;
  const int grid = ((ntiles + 7) / 8) * 8;
  const bool va = (a.cin % 4 == 0) && (((uintptr_t)a.in & 15) == 0);
  const bool vb = (a.cout % 4 == 0) && (((uintptr_t)a.w & 15) == 0);
  if (va && vb)
    hipLaunchKernelGGL((k_conv_mfma<BM, BKC, WM, WN, TM, TN, true, true>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  else if (va)
    hipLaunchKernelGGL((k_conv_mfma<BM, BKC, WM, WN, TM, TN, true, false>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  else if (vb)
    hipLaunchKernelGGL((k_conv_mfma<BM, BKC, WM, WN, TM, TN, false, true>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  else
    hipLaunchKernelGGL((k_conv_mfma<BM, BKC, WM, WN, TM, TN, false, false>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  PH_LAUNCH_CHECK();
  ph_record_cfg(0, BM, BN, BKC, 1, 0, 0, 4);
  return 0;
}

// Tile selection (measured on MI355X, profiles/r1b_op_bench.json; two restructurings of this kernel
// - a mask-sorted walk order with per-group offset skipping, and a double-buffered / LDS-index variant,
// both in the git history - measured 10-20 % slower: they cost registers / LDS, hence resident
// workgroups, and this kernel lives on thread-level parallelism; see profiles/README.md): 64-row tiles beat 128-row tiles
// at every PaSCo layer shape (more workgroups in flight hide the gather latency better); layers
// whose 64-row tiling cannot cover the 256 CUs twice drop to 32 rows.  PASCO_CONV_CFG="bm" overrides.
static int pick_cfg(const ConvArgs &a, int *bm) {
  const int bn = a.cout <= 32 ? 32 : (a.cout <= 64 ? 64 : 128);
  const int64_t ncol = (a.cout + bn - 1) / bn;
  int m = bn == 32 ? 128 : 64;
  if (bn == 128 && ((a.n_out + 63) / 64) * ncol < 2 * 256) m = 32;
  static const char *env = PH_DEV_ENV("PASCO_CONV_CFG");   // tuning override, read once
  if (env) {
    const int em = atoi(env);
    if (em == 128 || (em == 64 && bn >= 64) || (em == 32 && bn == 128)) m = em;
  }
  *bm = m;
  return bn;
}

extern "C" int ph_conv_fwd(const ph_conv_desc *d, ph_stream_t stream) {
  PH_REQUIRE(d != nullptr, "conv_fwd: null desc");
  PH_REQUIRE(d->cin > 0 && d->cout > 0 && d->kvol >= 1 && d->kvol <= 4096,
             "conv_fwd: bad shape cin=%d cout=%d kvol=%d", d->cin, d->cout, d->kvol);
  PH_REQUIRE(d->n_out >= 0 && d->n_out < 0x7FFFFF00, "conv_fwd: bad n_out");
  if (d->n_out == 0) return 0;
  PH_REQUIRE(d->nbr != nullptr || (d->kvol == 1 && d->n_in == d->n_out),
             "conv_fwd: identity map needs kvol == 1 and n_in == n_out");
  // mode 2 reads only the pre-split operands and may write only the split output
  PH_REQUIRE((d->out || (d->mma_mode == 2 && d->out_split)) && (d->mma_mode == 2 || (d->in && d->weight)), "conv_fwd: null tensor");
  if (d->mma_mode == 1 || d->mma_mode == 2) return ph_conv_fwd_f16x3(d, ph_stream(stream));
  PH_REQUIRE(d->mma_mode == 0, "conv_fwd: unknown mma_mode %d", d->mma_mode);
  PH_REQUIRE(d->axis_table == nullptr, "conv_fwd: axis_table is served by mma_mode 2");
  ConvArgs a;
  a.in = d->in;
  a.w = d->weight;
  a.nbr = d->nbr;
  a.out = d->out;
  a.n_in = d->n_in;
  a.n_out = d->n_out;
  a.cin = d->cin;
  a.cout = d->cout;
  a.kvol = d->kvol;
  a.pro_act = d->pro_act;
  a.pro_scale = d->pro_scale;
  a.pro_shift = d->pro_shift;
  a.bias = d->bias;
  a.epi_scale = d->epi_scale;
  a.epi_shift = d->epi_shift;
  a.epi2_scale = d->epi2_scale;
  a.epi2_shift = d->epi2_shift;
  a.residual = d->residual;
  a.epi_act = d->epi_act;
  a.res_act = d->res_act;
  a.slope = d->epi_slope;
  a.n_row_tiles = a.n_col_tiles = 0;
  a.pred = d->exact_if;
  hipStream_t st = ph_stream(stream);
  int bm = 128;
  const int bn = pick_cfg(a, &bm);
#define PH_CONV_CASE(BM_, WM_, WN_, TM_, TN_) \
  if (bm == BM_) return launch_conv<BM_, 32, WM_, WN_, TM_, TN_>(a, st)
  if (bn == 32) {
    PH_CONV_CASE(128, 4, 1, 1, 1);
  } else if (bn == 64) {
    PH_CONV_CASE(128, 4, 1, 1, 2);
    PH_CONV_CASE(64, 2, 2, 1, 1);
  } else {
    PH_CONV_CASE(128, 2, 2, 2, 2);
    PH_CONV_CASE(64, 2, 2, 1, 2);
    PH_CONV_CASE(32, 1, 4, 1, 1);
  }
#undef PH_CONV_CASE
  ph_set_error("conv_fwd: no kernel for bm=%d bn=%d", bm, bn);
  return 1;
}
