// Sparse convolution forward for gfx950: output-stationary gather -> LDS tile -> fp32 MFMA.
//
//   out[o,:] = epi( sum_k [nbr[k][o] >= 0] * pro(in[nbr[k][o],:]) @ W[k] + bias )   (pasco_hip.h)
//
// One workgroup (256 threads = 4 wave64) owns BM (128/64/32) output rows x BN output channels and
// walks the (kernel offset, 32-input-channel chunk) stages.  The output tile lives in accumulator
// registers for the whole walk and is written exactly once -- no scatter-add, no atomics; HBM traffic
// is B_alg of SURVEY.md 8(d).  BN/ReLU of the producing layer is fused as a gather prologue (valid
// rows only), bias / BN / activation / second BN / residual as the epilogue, so ResidualBlocks
// (mink.py:618-658) need no elementwise passes.
//
// Structure of the walk (v3):
//   * the tile's neighbour indices for up to 32 offsets sit in LDS (`idx_s`), loaded coalesced once;
//     a per-row 32-bit offset mask is OR-reduced over each 32-row MFMA tile with wave shuffles
//     (offset compaction): offsets no row of the workgroup has are never staged, and a wave skips
//     the MFMAs of a row tile that lacks the offset;
//   * stages are double-buffered in LDS: while the waves run the MFMAs of stage s, the 16-byte
//     coalesced gathers of stage s+1 (a 32-float chunk of a row = 8 lanes x float4; W[k] slab from L2)
//     are in flight into registers and are written to the other buffer after the MFMAs - one barrier
//     per stage;
//   * loads are branch-free (clamped address + select), activations are branch-free
//     (max(v,0) + neg*min(v,0)), LDS fragment reads are issued one 8-wide k-step ahead of their MFMAs.
// v_mfma_f32_32x32x2_f32 is exact fp32 (C >= 128 layers are FLOP-bound in fp32).
//
// LDS layouts (bank maths in MI355X_MICROARCH.md "LDS"):
//   As[2][BM][36]    row-major; a lane reads its 4 k-values with one ds_read_b128; the 36-dword row
//                    stride puts 16 rows with distinct (row mod 16) on distinct 16-byte slots.
//   Bs[2][32][BN+4]  k-major; lanes 0..31 read 32 consecutive dwords (ds_read_b32).
// The contraction index inside an 8-wide step is permuted (hardware k-half h <-> k = 4h + s) so
// that A needs one wide read per 4 MFMAs; A and B use the same permutation.
#include <stdlib.h>

#include "ph_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CV_THREADS = 256;
constexpr int BKC = 32;
constexpr int A_LD = BKC + 4;
constexpr int KCH = 32;  // kernel offsets whose indices are resident in LDS at a time

struct ConvArgs {
  const float *in;
  const float *w;
  const int32_t *nbr;
  float *out;
  int64_t n_in, n_out;
  int cin, cout, kvol;
  const float *pro_scale, *pro_shift, *bias, *epi_scale, *epi_shift, *epi2_scale, *epi2_shift, *residual;
  float pro_neg, epi_neg, res_neg;  // activation as max(v,0) + neg*min(v,0): 1 = none, 0 = ReLU, slope = leaky
  int has_pro, has_tail;
  int n_row_tiles, n_col_tiles;
  const int32_t *perm;
};

__device__ __forceinline__ float act_neg(float v, float neg) { return fmaxf(v, 0.f) + neg * fminf(v, 0.f); }

// WM x WN waves (4 in total), each TM x TN MFMA tiles of 32x32:  BM = WM*TM*32, BN = WN*TN*32.
template <int BM, int WM, int WN, int TM, int TN, bool VEC_A, bool VEC_B>
__global__ void __launch_bounds__(CV_THREADS) k_conv_mfma(ConvArgs a) {
  constexpr int BN = WN * TN * 32;
  constexpr int B_LD = BN + 4;
  constexpr int A_TPR = BKC / 4;                 // threads (float4 slots) per gathered row chunk (8)
  constexpr int A_RPP = CV_THREADS / A_TPR;      // rows per pass (32)
  constexpr int A_PASSES = BM / A_RPP;           // float4 slots per thread
  constexpr int B_SLOTS = BKC * (BN / 4) / CV_THREADS;
  constexpr int NT = BM / 32;                    // 32-row MFMA tiles of the workgroup
  static_assert(WM * WN == 4 && WM * TM * 32 == BM, "tile shape");
  static_assert(A_PASSES >= 1 && B_SLOTS >= 1, "loader shape");

  __shared__ __attribute__((aligned(16))) float As[2][BM * A_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BKC * B_LD];
  __shared__ int idx_s[KCH][BM];
  __shared__ unsigned tmask_s[NT];

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
  const int nchunks = (cin + BKC - 1) / BKC;
  const int a_c4 = tid & (A_TPR - 1);
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
  unsigned ra_valid = 0;   // bit p: ra[p] holds a real row (of the stage currently in registers)
  int ra_c0 = 0;

  // ---- global -> registers.  Addresses are clamped to something readable and the loaded values are
  // kept raw: masking happens in store_stage, AFTER the MFMAs of the running stage, so the loads
  // stay in flight across the whole compute phase (a select right after the load would force the
  // s_waitcnt here).
  auto load_stage = [&](int kk /*offset within the resident chunk*/, int kabs, int c0) {
    ra_c0 = c0;
    ra_valid = 0;
    const int cbase = c0 + a_c4 * 4;
    const bool cok = cbase < cin;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int idx = idx_s[kk][a_r0 + p * A_RPP];
      const bool ok = idx >= 0;
      ra_valid |= (ok ? 1u : 0u) << p;
      const float *src = a.in + (int64_t)(ok ? idx : 0) * cin + (cok ? cbase : 0);
      if (VEC_A) {
        ra[p] = *reinterpret_cast<const float4 *>(src);
      } else {
        float4 v;
        v.x = src[0];
        v.y = src[(cbase + 1 < cin) ? 1 : 0];
        v.z = src[(cbase + 2 < cin) ? 2 : 0];
        v.w = src[(cbase + 3 < cin) ? 3 : 0];
        ra[p] = v;
      }
    }
    const float *wk = a.w + (int64_t)kabs * cin * cout;
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) {
      const int slot = tid + q * CV_THREADS;
      const int kr = slot / (BN / 4);
      const int n4 = slot - kr * (BN / 4);
      const int c = c0 + kr;
      const int n = n0 + n4 * 4;
      const bool ok = (c < cin) && (n < cout);
      const float *src = wk + (int64_t)(ok ? c : 0) * cout + (ok ? n : 0);
      if (VEC_B) {
        rb[q] = *reinterpret_cast<const float4 *>(src);
      } else {
        float4 v;
        v.x = src[0];
        v.y = src[(ok && n + 1 < cout) ? 1 : 0];
        v.z = src[(ok && n + 2 < cout) ? 2 : 0];
        v.w = src[(ok && n + 3 < cout) ? 3 : 0];
        rb[q] = v;
      }
    }
  };

  // ---- registers -> LDS buffer `buf`: zero what is not there (absent neighbours, channels beyond
  // cin / cout) and apply the gather prologue (BN affine + activation) to real rows only ---------------
  auto store_stage = [&](int buf) {
    const int cbase = ra_c0 + a_c4 * 4;
    float ps[4], pb[4];
    bool chan[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      chan[j] = cbase + j < cin;
      const int c = chan[j] ? cbase + j : 0;
      ps[j] = (a.has_pro && a.pro_scale) ? a.pro_scale[c] : 1.f;
      pb[j] = (a.has_pro && a.pro_shift) ? a.pro_shift[c] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const bool ok = (ra_valid >> p) & 1u;
      float4 v = ra[p];
      if (a.has_pro) {
        v.x = act_neg(v.x * ps[0] + pb[0], a.pro_neg);
        v.y = act_neg(v.y * ps[1] + pb[1], a.pro_neg);
        v.z = act_neg(v.z * ps[2] + pb[2], a.pro_neg);
        v.w = act_neg(v.w * ps[3] + pb[3], a.pro_neg);
      }
      v.x = (ok && chan[0]) ? v.x : 0.f;
      v.y = (ok && chan[1]) ? v.y : 0.f;
      v.z = (ok && chan[2]) ? v.z : 0.f;
      v.w = (ok && chan[3]) ? v.w : 0.f;
      *reinterpret_cast<float4 *>(&As[buf][(a_r0 + p * A_RPP) * A_LD + a_c4 * 4]) = v;
    }
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) {
      const int slot = tid + q * CV_THREADS;
      const int kr = slot / (BN / 4);
      const int n4 = slot - kr * (BN / 4);
      const int n = n0 + n4 * 4;
      const bool rok = ra_c0 + kr < cin;
      float4 v = rb[q];
      v.x = (rok && n + 0 < cout) ? v.x : 0.f;
      v.y = (rok && n + 1 < cout) ? v.y : 0.f;
      v.z = (rok && n + 2 < cout) ? v.z : 0.f;
      v.w = (rok && n + 3 < cout) ? v.w : 0.f;
      *reinterpret_cast<float4 *>(&Bs[buf][kr * B_LD + n4 * 4]) = v;
    }
  };

  // ---- MFMAs of one stage from LDS buffer `buf`; fragments are read one k8-step ahead ------------------
  unsigned mymask[TM];   // offsets (of the resident chunk) each of this wave's row tiles has
  auto compute_stage = [&](int buf, int kk) {
    bool on[TM];
    bool any_on = false;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      on[i] = (mymask[i] >> kk) & 1u;
      any_on |= on[i];
    }
    if (!any_on) return;
    const float *Ab = As[buf];
    const float *Bb = Bs[buf];
    float4 av[2][TM];
    float bv[2][4][TN];
    auto read_frag = [&](int slot, int k8) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        av[slot][i] = *reinterpret_cast<const float4 *>(&Ab[((wm * TM + i) * 32 + l31) * A_LD + k8 * 8 + h * 4]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bv[slot][s][j] = Bb[(k8 * 8 + h * 4 + s) * B_LD + (wn * TN + j) * 32 + l31];
    };
    read_frag(0, 0);
#pragma unroll
    for (int k8 = 0; k8 < BKC / 8; ++k8) {
      const int cur = k8 & 1;
      if (k8 + 1 < BKC / 8) read_frag(cur ^ 1, k8 + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the look-ahead LDS reads ahead of this step's MFMAs
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (!on[i]) continue;
          const float aval = (s == 0) ? av[cur][i].x : (s == 1) ? av[cur][i].y : (s == 2) ? av[cur][i].z : av[cur][i].w;
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bv[cur][s][j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  // ---- walk: chunks of <= 32 kernel offsets whose indices are resident in LDS ----------------------------
  for (int k0 = 0; k0 < a.kvol; k0 += KCH) {
    const int kn = (a.kvol - k0) < KCH ? (a.kvol - k0) : KCH;
    if (k0 > 0) __syncthreads();   // previous chunk's idx_s / buffers no longer read
    for (int e = tid; e < kn * BM; e += CV_THREADS) {
      const int kk = e / BM, r = e - kk * BM;
      const int64_t row = m0 + r;
      int idx = -1;
      if (row < a.n_out) idx = a.nbr ? a.nbr[(int64_t)(k0 + kk) * a.n_out + row] : (int)row;
      idx_s[kk][r] = idx;
    }
    __syncthreads();
    // offset compaction: per-row masks OR-reduced over every 32-row tile (wave shuffles)
    for (int t = wave; t < NT; t += 4) {
      unsigned m = 0;
      if (lane < 32) {
        for (int kk = 0; kk < kn; ++kk) m |= (idx_s[kk][t * 32 + lane] >= 0 ? 1u : 0u) << kk;
      }
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) m |= __shfl_xor(m, d);
      if (lane == 0) tmask_s[t] = m;
    }
    __syncthreads();
    unsigned kmask = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) kmask |= tmask_s[t];
#pragma unroll
    for (int i = 0; i < TM; ++i) mymask[i] = tmask_s[wm * TM + i];

    if (kmask != 0u) {
      int kk = __builtin_ctz(kmask);
      int chunk = 0;
      int buf = 0;
      load_stage(kk, k0 + kk, 0);
      store_stage(0);
      __syncthreads();
      for (;;) {
        // next stage
        int kk2 = kk, c2 = chunk + 1;
        if (c2 == nchunks) {
          c2 = 0;
          const unsigned rest = (kk + 1 < 32) ? (kmask >> (kk + 1)) : 0u;
          kk2 = rest ? kk + 1 + __builtin_ctz(rest) : KCH;
        }
        const bool more = kk2 < KCH;
        if (more) load_stage(kk2, k0 + kk2, c2 * BKC);
        compute_stage(buf, kk);
        if (!more) break;
        store_stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        kk = kk2;
        chunk = c2;
      }
    }
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + l31;
    const bool cok = col < cout;
    const int cc = cok ? col : 0;
    const float bias = a.bias ? a.bias[cc] : 0.f;
    const float es = a.epi_scale ? a.epi_scale[cc] : 1.f;
    const float eb = a.epi_shift ? a.epi_shift[cc] : 0.f;
    const float es2 = a.epi2_scale ? a.epi2_scale[cc] : 1.f;
    const float eb2 = a.epi2_shift ? a.epi2_shift[cc] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t srow = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (srow >= a.n_out || !cok) continue;
        const int64_t row = a.perm ? (int64_t)a.perm[srow] : srow;
        float v = acc[i][j][r] + bias;
        v = act_neg(v * es + eb, a.epi_neg);
        if (a.has_tail) {
          v = v * es2 + eb2;
          if (a.residual) v += a.residual[row * cout + col];
          v = act_neg(v, a.res_neg);
        }
        a.out[row * cout + col] = v;
      }
    }
  }
}

template <int BM, int WM, int WN, int TM, int TN>
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
    hipLaunchKernelGGL((k_conv_mfma<BM, WM, WN, TM, TN, true, true>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  else if (va)
    hipLaunchKernelGGL((k_conv_mfma<BM, WM, WN, TM, TN, true, false>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  else if (vb)
    hipLaunchKernelGGL((k_conv_mfma<BM, WM, WN, TM, TN, false, true>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  else
    hipLaunchKernelGGL((k_conv_mfma<BM, WM, WN, TM, TN, false, false>), dim3(grid), dim3(CV_THREADS), 0, st, args);
  PH_LAUNCH_CHECK();
  return 0;
}

// Tile selection (measured on MI355X, profiles/r1b_op_bench.json): 64-row tiles beat 128-row tiles
// at every PaSCo layer shape (more workgroups in flight hide the gather latency better); layers
// whose 64-row tiling cannot cover the 256 CUs twice drop to 32 rows.  PASCO_CONV_CFG="bm" overrides.
static int pick_cfg(const ConvArgs &a, int *bm) {
  const int bn = a.cout <= 32 ? 32 : (a.cout <= 64 ? 64 : 128);
  const int64_t ncol = (a.cout + bn - 1) / bn;
  int m = bn == 32 ? 128 : 64;
  if (bn == 128 && ((a.n_out + 63) / 64) * ncol < 2 * 256) m = 32;
  const char *env = getenv("PASCO_CONV_CFG");
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
  PH_REQUIRE(d->in && d->weight && d->out, "conv_fwd: null tensor");
  if (d->n_out == 0) return 0;
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
  a.has_pro = (d->pro_scale || d->pro_shift || d->pro_act != PH_ACT_NONE) ? 1 : 0;
  a.has_tail = (d->residual || d->epi2_scale || d->epi2_shift || d->res_act != PH_ACT_NONE) ? 1 : 0;
  a.n_row_tiles = a.n_col_tiles = 0;
  a.perm = d->perm;
  PH_REQUIRE((d->perm == nullptr) == (d->gmask == nullptr), "conv_fwd: perm and gmask come together");
  PH_REQUIRE(d->perm == nullptr || d->nbr != nullptr, "conv_fwd: a schedule needs its neighbour table");
  hipStream_t st = ph_stream(stream);
  int bm = 128;
  const int bn = pick_cfg(a, &bm);
#define PH_CONV_CASE(BM_, WM_, WN_, TM_, TN_) \
  if (bm == BM_) return launch_conv<BM_, WM_, WN_, TM_, TN_>(a, st)
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
