// Sparse convolution forward, split-precision products, 256 x 256 tiles for 256-output-channel layers (round 3; kernel id 6,
// `k_conv_wide`).
//
// What bounds k_conv_dma (conv_dma.hip) on the wide layers, measured (profiles/README.md "round 3"): the CU's texture-address
// unit accepts one LDS-DMA instruction (1 KB) per ~20 clocks, the four waves of a workgroup take turns, and a wave whose DMA is
// waiting for its turn is stalled IN ORDER - the MFMAs behind it do not issue.  With 128 x 128 tiles a stage moves 32 KB per
// 768 matrix clocks: the address unit is busy 83 % of the matrix time and the two co-resident workgroups' waves spend their
// time queueing for it (matrix pipe 55 % busy).  A micro-benchmark (tools/ubench/spec2_bench.hip, profiles/r3a_*) shows the
// regime that does overlap: waves that interleave their own DMAs with their own MFMAs, two per SIMD, with the address unit
// under ~50 % - and that a single wave per SIMD cannot hide its own queueing however its stream is ordered (k_conv_big, round
// 3: 536 us against 526 us).
// So: ONE workgroup per CU, 8 waves (two per SIMD) as 4 x 2, each wave 64 rows x 128 channels (48 MFMAs per 32-channel
// stage), tile 256 x 256 - every gathered row and every weight row is loaded once per 256 x 256 x 32 products:
// 64 KB per 3072 matrix clocks per SIMD pair = 21 B/clk/CU, address unit 42 % busy.  Two LDS stage buffers; ONE fragment set
// per wave used at half-stage granularity (the k = 0..15 half of a stage multiplies while the k = 16..31 half is read, and
// the next stage's first half is read under the second half's products); one barrier and one vmcnt(0) per stage, in the
// middle: by then the DMA of the next stage has had a full stage to land.  DMA, fragment reads and address arithmetic sit
// between the MFMAs (one read per two MFMAs, one DMA per three).
//
// Same operand layout, per-accumulator product order and epilogue (conv_h2_common.h) as k_conv_dma: bit-identical results
// for ksplit == 1.
#include <stdlib.h>

#include "conv_h2_common.h"

constexpr int WIDE_KMAX = 27;     // kernel offsets one workgroup walks (index table in LDS); more -> split over the offsets

// TN = 4: 256 x 256 tiles (256 output channels); TN = 2: 256 x 128 tiles (128 output channels: the address unit is busy 62 %
// of the matrix time instead of 42 % - still the better side of k_conv_dma's 83 % where the map fills the chip in whole rounds)
template <int TN, bool EMIT>
__global__ void __launch_bounds__(512, 2) k_conv_wide(ConvArgsH a) {
  constexpr int WM = 4, WN = 2, TM = 2;
  constexpr int NT = 512;
  constexpr int BM = WM * TM * 32;           // 256
  constexpr int BN = WN * TN * 32;           // 256 or 128
  constexpr int RPP = NT / 8;                // tile rows one DMA pass covers: 8 waves x 8 rows x 128 B
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
  constexpr int L = A_PASSES + B_PASSES;     // DMA instructions per thread and stage (8 / 6)
  constexpr int NMH = TM * TN * 3;           // MFMAs per wave and half stage (24 / 12)
  constexpr int NRH = (TM + TN) * 2;         // fragment reads per wave and half stage (12 / 8)
  static_assert(A_PASSES == 4 && (L == 8 || L == 6), "one 16-byte index read per thread and stage");
  __shared__ __attribute__((aligned(128))) char lds[2 * STAGE + WIDE_KMAX * BM * 4];

  const int nwg = gridDim.x;
  const int cpx = nwg >> 3;
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * cpx + (bid >> 3);
  if (tile >= a.n_row_tiles * a.n_col_tiles) return;
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
  const int nchunks = a.cpad >> 5;
  const int kper = (a.kvol + a.ksplit - 1) / a.ksplit;
  const int k_begin = (int)blockIdx.y * kper;
  const int k_end = (k_begin + kper < a.kvol) ? k_begin + kper : a.kvol;
  const int kcount = k_end > k_begin ? k_end - k_begin : 0;
  const int nstages = kcount * nchunks;
  const uint32_t rsb = 4u * (uint32_t)a.cpad;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nstages > 0) {
    // ---- neighbour indices of the tile -> LDS, layout [offset][l_r = tile row mod 32][p = tile row / 32] ---------------------
    int *idx_lds = reinterpret_cast<int *>(lds + 2 * STAGE);
    for (int i = tid; i < kcount * BM; i += NT) {
      const int k = i / BM, r = i - k * BM;
      const int64_t row = m0 + r;
      int idx = -1;
      if (row < a.n_out) idx = a.nbr ? a.nbr[(int64_t)(k_begin + k) * a.nbr_stride + row] : (int)row;
      idx_lds[k * BM + (r % RPP) * A_PASSES + (r / RPP)] = idx;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- DMA geometry of this thread: tile row l_r + 32 p, 16-byte slot l_j; source chunk swizzled (as k_conv_dma) ----------
    const int l_j = tid & 7;
    const int l_r = tid >> 3;
    const uint32_t sj16 = (uint32_t)((l_j ^ ((l_r >> 1) & 7)) << 4);
    const uint64_t in_base = (uint64_t)reinterpret_cast<uintptr_t>(a.in_split) + sj16;
    const uint64_t zero_src = (uint64_t)reinterpret_cast<uintptr_t>(a.zero) + sj16;
    const uint64_t w_base = (uint64_t)reinterpret_cast<uintptr_t>(a.w_split) + sj16;
    uint32_t boff[B_PASSES];
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) {
      int n = n0 + l_r + q * RPP;
      n = n < cout ? n : cout - 1;
      boff[q] = (uint32_t)n * rsb;
    }
    const int64_t wslab = (int64_t)cout * rsb;
    typedef int i32x4 __attribute__((ext_vector_type(4)));

    auto stage_kc = [&](int s, int &k, uint32_t &coff) {   // the tail re-loads the last stage instead of branching
      const int sc = s < nstages - 1 ? s : nstages - 1;
      k = sc / nchunks;
      coff = (uint32_t)(sc - k * nchunks) << 7;
    };
    struct Src {
      uint64_t a[A_PASSES];
      uint64_t w;
    };
    auto prep = [&](int s, Src &src) {                      // one LDS read + VALU: source addresses of stage s
      int k;
      uint32_t coff;
      stage_kc(s, k, coff);
      const i32x4 idx = *reinterpret_cast<const i32x4 *>(idx_lds + k * BM + l_r * A_PASSES);
#pragma unroll
      for (int p = 0; p < A_PASSES; ++p) {
        const int ix = idx[p];
        uint64_t v = in_base + (uint64_t)(uint32_t)(ix < 0 ? 0 : ix) * rsb + coff;
        asm volatile("" : "+v"(v));
        src.a[p] = ix >= 0 ? v : zero_src;
      }
      src.w = w_base + (uint64_t)((int64_t)(k_begin + k) * wslab) + coff;
    };
    auto dma_one = [&](const Src &src, char *sbuf, int t) {
      if (t < A_PASSES) {
        char *dst = sbuf + (t * RPP + wave * 8) * 128;     // wave-uniform; the DMA adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src.a[t],
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      } else {
        const int q = t - A_PASSES;
        char *dst = sbuf + A_BYTES + (q * RPP + wave * 8) * 128;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(src.w + boff[q]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
    };
    auto fire_all = [&](const Src &src, char *sbuf) {
#pragma unroll
      for (int t = 0; t < L; ++t) dma_one(src, sbuf, t);
    };

    // ---- fragments: lane = (tile row l31 of a 32-row block, k-half h); chunk c of row r sits at c ^ ((r >> 1) & 7) -------------
    const int sw = (l31 >> 1) & 7;
    uint32_t xo[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int part = 0; part < 2; ++part) xo[ks][part] = (uint32_t)(((part * 4 + ks * 2 + h) ^ sw) << 4);
    const uint32_t arow = (uint32_t)((wm * TM * 32 + l31) * 128);
    const uint32_t brow = (uint32_t)(A_BYTES + (wn * TN * 32 + l31) * 128);
    struct Frag {                // one stage: halves ks = 0 / 1 used independently
      f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    };
    Frag f;
    // the r-th of the NRH fragment reads of half `ks` (A hi / lo per i, then B hi / lo per j)
    auto read_one = [&](const char *base, int ks, int r) {
      if (r < 2 * TM) {
        const int i = r >> 1;
        if ((r & 1) == 0) f.ah[ks][i] = *reinterpret_cast<const f16x8 *>(base + arow + i * 4096 + xo[ks][0]);
        else f.al[ks][i] = *reinterpret_cast<const f16x8 *>(base + arow + i * 4096 + xo[ks][1]);
      } else {
        const int j = (r - 2 * TM) >> 1;
        if (((r - 2 * TM) & 1) == 0) f.bh[ks][j] = *reinterpret_cast<const f16x8 *>(base + brow + j * 4096 + xo[ks][0]);
        else f.bl[ks][j] = *reinterpret_cast<const f16x8 *>(base + brow + j * 4096 + xo[ks][1]);
      }
    };
    // the m-th of the NMH MFMAs of half `ks`: weights first (transposed accumulator block); per accumulator the smallest
    // terms first, its three products TM * TN issues apart
    auto mfma_one = [&](int ks, int m) {
      const int ij = m % (TM * TN), p = m / (TM * TN);
      const int i = ij / TN, j = ij % TN;
      if (p == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.al[ks][i], acc[i][j], 0, 0, 0);
      else if (p == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
    };
    Src src;
    // first half of stage s: its k = 0..15 products from registers, its k = 16..31 fragments from `cur`
    auto half_a = [&](const char *cur) {
#pragma unroll
      for (int g = 0; g < NRH; ++g) {
        read_one(cur, 1, g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = g * NMH / NRH; m < (g + 1) * NMH / NRH; ++m) mfma_one(0, m);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // second half: the k = 16..31 products; the DMA of stage s + 2 into `cur` (free: every wave passed the barrier after its
    // last read of it), the first-half fragments of stage s + 1 from `nxt`, the addresses of stage s + 3
    auto half_b = [&](char *cur, const char *nxt, int s) {
#pragma unroll
      for (int g = 0; g < NRH; ++g) {
        if (g < L) dma_one(src, cur, g);
        read_one(nxt, 0, g);
        if (g == L) prep(s + 3, src);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = g * NMH / NRH; m < (g + 1) * NMH / NRH; ++m) mfma_one(1, m);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    char *b0 = lds, *b1 = lds + STAGE;
    prep(0, src);
    fire_all(src, b0);
    prep(1, src);
    fire_all(src, b1);
    prep(2, src);
    if (L == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // stage 0 landed (stage 1 may fly)
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < NRH; ++r) read_one(b0, 0, r);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the middle of a stage: this wave's second-half fragments are in registers and its share of the next stage has landed;
    // after the barrier that holds for every wave: `cur` may be overwritten, `nxt` may be read
#define WIDE_MIDDLE()                                           \
  do {                                                          \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                               \
  } while (0)
    for (int s = 0; s < nstages; s += 2) {
      half_a(b0);
      WIDE_MIDDLE();
      half_b(b0, b1, s);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (s + 1 >= nstages) break;
      half_a(b1);
      WIDE_MIDDLE();
      half_b(b1, b0, s + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // clamped tail loads: nothing may land after the epilogue starts
#undef WIDE_MIDDLE
  }

  h2_store_tile<TM, TN, EMIT>(a, acc, m0, n0, wm, wn, h, l31);
}

// ph_conv_desc.route (tests/test_hip_wide.py): PH_ROUTE_WIDE_ALWAYS = every served shape whatever its size, PH_ROUTE_WIDE_NEVER = none

template <int TN>
static int launch_wide(const ConvArgsH &a, hipStream_t st) {
  constexpr int BN = 2 * TN * 32;
  ConvArgsH args = a;
  args.n_row_tiles = (int)((a.n_out + 255) / 256);
  args.n_col_tiles = (a.cout + BN - 1) / BN;
  const int ntiles = args.n_row_tiles * args.n_col_tiles;
  const int grid = ((ntiles + 7) / 8) * 8;
  const bool emit = args.out_split != nullptr && args.ksplit == 1;
  if (emit) hipLaunchKernelGGL((k_conv_wide<TN, true>), dim3(grid, 1), dim3(512), 0, st, args);
  else hipLaunchKernelGGL((k_conv_wide<TN, false>), dim3(grid, args.ksplit), dim3(512), 0, st, args);
  PH_LAUNCH_CHECK();
  if (args.ksplit > 1) {
    if (int rc = ph_launch_splitk_epilogue(args, st)) return rc;
  }
  ph_record_cfg(2, 256, BN, 32, args.ksplit, emit ? 1 : 0, 6, 8);
  return 0;
}

// Takes gather launches with 256 or 128 output channels and >= 8 kernel offsets; -1 = not served (the caller goes on to
// k_conv_dma).  One workgroup per CU.  256 channels: maps with at least ~160 row tiles run unsplit (one round); fewer row
// tiles are split over the kernel offsets so that the launch has about one workgroup per CU, each slice at most WIDE_KMAX
// offsets.  128 channels: only where the row tiles fill the CUs in (nearly) whole rounds - k_conv_dma's two workgroups per
// CU and its tail split serve the other sizes better.
int ph_conv_wide_try(const ConvArgsH &a_in, hipStream_t st) {
  if ((a_in.cout != 256 && a_in.cout != 128) || a_in.kvol < 8 || a_in.tile_k != nullptr || a_in.win_gather || (a_in.cpad & 31) ||
      a_in.nbr == nullptr)
    return -1;
  const int64_t trow = (a_in.n_out + 255) / 256;
  // measured (profiles/r3n_layer_ab_wide.txt): 477 vs 520 us at 53 k rows, 162 vs 188 us at 15.6 k rows; below ~12 k rows
  // (fewer than 48 row tiles: a deep split over the offsets) and on the bottleneck's 245 / 75-offset products k_conv_dma's
  // smaller tiles win
  if (a_in.route & PH_ROUTE_WIDE_NEVER) return -1;
  if (!(a_in.route & PH_ROUTE_WIDE_ALWAYS)) {
    if (a_in.kvol > WIDE_KMAX) return -1;
    if (a_in.cout == 256 && trow < 48) return -1;
    if (a_in.cout == 128) {      // whole rounds only: the last round at least ~60 % full, no split over the offsets
      const int64_t rest = trow % 256;
      if (trow >= 256 && rest != 0 && rest < 150 && a_in.ksplit == 1 && a_in.partial == nullptr && a_in.tail_ws != nullptr) {
        // whole rounds + a few row tiles (the encoder's 71.5 k-row level: 280 tiles): the whole rounds here (0.35 of the matrix peak
        // against k_conv_dma's 0.24), the left-over rows on k_conv_dma, split over the offsets so that they fill the chip
        const int64_t r0 = (trow - rest) * 256, tail_rows = a_in.n_out - r0;
        int ks = (int)(512 / ((tail_rows + 127) / 128));
        if (ks > a_in.kvol / 3) ks = a_in.kvol / 3;
        if (ks > 12) ks = 12;
        while (ks > 1 && (a_in.kvol + (a_in.kvol + ks - 1) / ks - 1) / ((a_in.kvol + ks - 1) / ks) != ks) --ks;      // no empty slice
        const char *zero0 = ph_dma_zero_line();
        if (ks >= 2 && zero0 != nullptr && (int64_t)ks * tail_rows * a_in.cout * 4 <= a_in.tail_ws_bytes) {
          ConvArgsH head = a_in;
          head.zero = zero0;
          head.ablate = 0;
          head.n_out = r0;
          head.ksplit = 1;
          head.partial = nullptr;
          if (int rc = launch_wide<2>(head, st)) return rc;
          ConvArgsH tail = a_in;
          tail.n_out = tail_rows;
          tail.nbr = a_in.nbr + r0;
          if (tail.out) tail.out = a_in.out + r0 * a_in.cout;
          if (tail.out_split) tail.out_split = a_in.out_split + r0 * a_in.cout * 2;
          if (tail.residual) tail.residual = a_in.residual + r0 * a_in.cout;
          if (tail.axis_coords) tail.axis_coords = a_in.axis_coords + r0 * 4;
          tail.ksplit = ks;
          tail.partial = (float *)a_in.tail_ws;
          return ph_conv_dma_try(tail, 128, st);
        }
      }
      if (trow < 150 || (rest != 0 && rest < 150)) return -1;
    }
  }
  const char *zero = ph_dma_zero_line();
  if (zero == nullptr) return -1;
  ConvArgsH a = a_in;
  a.zero = zero;
  a.ablate = 0;
  int ks = (a.kvol + WIDE_KMAX - 1) / WIDE_KMAX;                     // what the index table demands
  if (trow * ks < 160) {                                             // well under one workgroup per CU: split further
    int want = (int)((256 + trow / 2) / trow);
    const int kmax = a.kvol / 3 > 0 ? a.kvol / 3 : 1;
    if (want > kmax) want = kmax;
    if (want > ks) ks = want;
  }
  a.ksplit = 1;
  a.partial = nullptr;
  if (ks > 1) {
    if (a.tail_ws == nullptr || (int64_t)ks * a.n_out * a.cout * 4 > a.tail_ws_bytes) return -1;
    a.ksplit = ks;
    a.partial = (float *)a.tail_ws;
  }
  return a.cout == 256 ? launch_wide<4>(a, st) : launch_wide<2>(a, st);
}
