// Sparse convolution forward, split-precision products, ONE wave per SIMD owning its whole instruction stream
// (round 3; kernel id 6, `k_conv_big`).
//
// Why another kernel.  k_conv_dma (conv_dma.hip) runs two 4-wave workgroups per CU; each wave alternates a fragment-read
// phase and a DMA + MFMA phase between two barriers and relies on the OTHER workgroup's wave of its SIMD to fill the matrix
// pipe meanwhile.  Measured on gfx950 (profiles/r3a_spec2_wave_roles.txt, profiles/r2d_dma_ablate.txt):
//   * LDS-DMA issued by one wave and MFMAs issued by ANOTHER wave of the same SIMD do not overlap (roles split over two
//     waves: 998 + 421 -> 1257 clk per iteration), while a wave that interleaves its own DMAs with its own MFMAs hides them
//     completely (988 clk = the MFMA time), with one or two such waves per SIMD;
//   * on the 256 -> 256 layer the gather kernel takes 602 us = its DMA pipeline alone (328 us) + its matrix work (~270 us):
//     the two co-resident workgroups add their phases instead of overlapping them; the matrix pipe is 55 % busy.
// So this kernel gives every SIMD ONE wave that never leaves the matrix stream: a 128 x 256 tile (all 256 output channels:
// every gathered row is loaded once, not once per 128-channel column tile), 64 x 128 per wave (48 MFMAs per 32-channel
// stage), and per stage 12 LDS-DMA instructions + 24 fragment reads + the address arithmetic issued BETWEEN those MFMAs
// (one DMA and two reads per four MFMAs: ~1.3 of the ~5 issue slots a 32-clock MFMA leaves).  Three LDS stage buffers +
// register double-buffered fragments: while stage s multiplies from registers, stage s + 1 is read from LDS and stage s + 3
// is in flight into the buffer stage s was read from; one barrier per stage, counted vmcnt (two stages of DMA latency
// slack).  Bytes through the vector-memory path per MFMA clock: 48 KB / 1536 clk = 31 B/clk/CU against 43 for 128 x 128.
//
// Same operand layout, tile algebra per 32 x 32 block, accumulation order over (offset, chunk) and epilogue
// (conv_h2_common.h) as k_conv_dma: results are bit-identical to it for ksplit == 1.
#include <stdlib.h>

#include "conv_h2_common.h"

constexpr int BIG_KMAX = 28;      // kernel offsets one workgroup walks (index table in LDS); more -> split over the offsets

template <int TM, int TN, bool EMIT>
__global__ void __launch_bounds__(256, 1) k_conv_big(ConvArgsH a) {
  constexpr int WM = 2, WN = 2;
  constexpr int BM = WM * TM * 32;           // 128 (TM = 2) or 256
  constexpr int BN = WN * TN * 32;           // 256 (TN = 4) or 128
  constexpr int RPP = 32;                    // tile rows one DMA pass covers: 4 waves x 8 rows x 128 B
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
  constexpr int L = A_PASSES + B_PASSES;     // DMA instructions per thread and stage (12)
  constexpr int NM = 2 * TM * TN * 3;        // MFMAs per wave and stage (48)
  constexpr int NR = 2 * (TM + TN) * 2;      // fragment reads per wave and stage (24)
  static_assert(A_PASSES == 4 && L == 12, "one 16-byte index read per thread and stage; vmcnt(12)");
  __shared__ __attribute__((aligned(128))) char lds[3 * STAGE + BIG_KMAX * BM * 4];

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
    int *idx_lds = reinterpret_cast<int *>(lds + 3 * STAGE);
    for (int i = tid; i < kcount * BM; i += 256) {
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
    struct Frag {
      f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    };
    // the r-th of the NR fragment reads of a stage (order: ks, then A hi / lo per i, then B hi / lo per j)
    auto read_one = [&](const char *base, Frag &f, int r) {
      const int ks = r / (2 * (TM + TN)), q = r % (2 * (TM + TN));
      if (q < 2 * TM) {
        const int i = q >> 1;
        if ((q & 1) == 0) f.ah[ks][i] = *reinterpret_cast<const f16x8 *>(base + arow + i * 4096 + xo[ks][0]);
        else f.al[ks][i] = *reinterpret_cast<const f16x8 *>(base + arow + i * 4096 + xo[ks][1]);
      } else {
        const int j = (q - 2 * TM) >> 1;
        if (((q - 2 * TM) & 1) == 0) f.bh[ks][j] = *reinterpret_cast<const f16x8 *>(base + brow + j * 4096 + xo[ks][0]);
        else f.bl[ks][j] = *reinterpret_cast<const f16x8 *>(base + brow + j * 4096 + xo[ks][1]);
      }
    };
    auto read_all = [&](const char *base, Frag &f) {
#pragma unroll
      for (int r = 0; r < NR; ++r) read_one(base, f, r);
    };
    // the m-th of the NM MFMAs of a stage: weights first (transposed accumulator block), smallest terms first
    auto mfma_one = [&](const Frag &f, int m) {
      // accumulator-major inside a product class: the three products of one accumulator (in this order: smallest terms
      // first) are TM * TN issues apart, so no MFMA waits on the one before it and an instruction slipped between two
      // MFMAs never sits inside a dependent pair
      const int ij = m % (TM * TN), p = (m / (TM * TN)) % 3, ks = m / (3 * TM * TN);
      const int i = ij / TN, j = ij % TN;
      if (p == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.al[ks][i], acc[i][j], 0, 0, 0);
      else if (p == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
    };
    // one stage: matrix work of `fc` (registers), the fragments of the next stage from `rbuf` into `fn`, the DMA of stage sd
    // into `dbuf` - one DMA and two fragment reads per four MFMAs
    // `src` = the source addresses of stage sd (prepared during the previous stage); the last group prepares stage sd + 1's
    Src src;
    auto stage_body = [&](const Frag &fc, Frag &fn, const char *rbuf, char *dbuf, int sd) {
#pragma unroll
      for (int g = 0; g < L; ++g) {
        dma_one(src, dbuf, g);
#pragma unroll
        for (int r = g * NR / L; r < (g + 1) * NR / L; ++r) read_one(rbuf, fn, r);
        if (g == L - 1) prep(sd + 1, src);     // index read + address arithmetic under this group's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = g * NM / L; m < (g + 1) * NM / L; ++m) mfma_one(fc, m);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // stage boundary: at most the L youngest DMA instructions (the stage issued last) still fly, this wave's fragment reads are
    // done; after the barrier every wave's share of the next stage's buffer has landed and the buffer read last is free
#define BIG_BOUNDARY()                                            \
  do {                                                            \
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                                 \
  } while (0)

    char *b0 = lds, *b1 = lds + STAGE, *b2 = lds + 2 * STAGE;
    Frag f0, f1;
    prep(0, src);
    fire_all(src, b0);
    prep(1, src);
    fire_all(src, b1);
    prep(2, src);
    fire_all(src, b2);
    prep(3, src);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // stage 0 landed (stages 1, 2 may fly)
    __builtin_amdgcn_s_barrier();
    read_all(b0, f0);
    BIG_BOUNDARY();                                        // stage 1 landed, buffer 0 free
    // steady state, three stages per trip (the buffer roles rotate with period 3, the fragment sets with period 2):
    //   stage s from f_(s & 1); read buffer (s + 1) % 3; DMA of stage s + 3 into buffer s % 3
    int s = 0;
    while (s < nstages) {
      stage_body(f0, f1, b1, b0, s + 3);             // s
      BIG_BOUNDARY();
      if (++s >= nstages) break;
      stage_body(f1, f0, b2, b1, s + 3);             // s + 1
      BIG_BOUNDARY();
      if (++s >= nstages) break;
      stage_body(f0, f1, b0, b2, s + 3);             // s + 2
      BIG_BOUNDARY();
      if (++s >= nstages) break;
      stage_body(f1, f0, b1, b0, s + 3);             // s + 3: as the first with the fragment sets swapped
      BIG_BOUNDARY();
      if (++s >= nstages) break;
      stage_body(f0, f1, b2, b1, s + 3);
      BIG_BOUNDARY();
      if (++s >= nstages) break;
      stage_body(f1, f0, b0, b2, s + 3);
      BIG_BOUNDARY();
      ++s;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // clamped tail loads: nothing may land after the epilogue starts
#undef BIG_BOUNDARY
  }

  h2_store_tile<TM, TN, EMIT>(a, acc, m0, n0, wm, wn, h, l31);
}

template <int TM, int TN>
static int launch_big(const ConvArgsH &a, hipStream_t st) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  ConvArgsH args = a;
  args.n_row_tiles = (int)((a.n_out + BM - 1) / BM);
  args.n_col_tiles = (a.cout + BN - 1) / BN;
  const int ntiles = args.n_row_tiles * args.n_col_tiles;
  const int grid = ((ntiles + 7) / 8) * 8;
  const bool emit = args.out_split != nullptr && args.ksplit == 1;
  if (emit) hipLaunchKernelGGL((k_conv_big<TM, TN, true>), dim3(grid, 1), dim3(256), 0, st, args);
  else hipLaunchKernelGGL((k_conv_big<TM, TN, false>), dim3(grid, args.ksplit), dim3(256), 0, st, args);
  PH_LAUNCH_CHECK();
  if (args.ksplit > 1) {
    if (int rc = ph_launch_splitk_epilogue(args, st)) return rc;
  }
  ph_record_cfg(2, BM, BN, 32, args.ksplit, emit ? 1 : 0, 6, 4);
  return 0;
}

// Takes gather launches with 256 output channels and >= 8 kernel offsets; -1 = not served (the caller goes on to k_conv_dma).
// One workgroup per CU: the split over the kernel offsets is chosen here so that the launch has about two rounds of
// workgroups (or one full round for few-row layers), each slice at most BIG_KMAX offsets.
int ph_conv_big_try(const ConvArgsH &a_in, hipStream_t st) {
  if (a_in.cout != 256 || a_in.kvol < 8 || a_in.tile_k != nullptr || a_in.win_gather || (a_in.cpad & 31)) return -1;
  const char *zero = ph_dma_zero_line();
  if (zero == nullptr) return -1;
  ConvArgsH a = a_in;
  a.zero = zero;
  a.ablate = 0;
  const int64_t trow = (a.n_out + 127) / 128;
  int ks = (a.kvol + BIG_KMAX - 1) / BIG_KMAX;                       // what the index table demands
  if (trow * ks < 256) {                                             // fewer workgroups than CUs: split further
    int want = (int)((256 + trow - 1) / trow);
    const int kmax = a.kvol / 3 > 0 ? a.kvol / 3 : 1;
    if (want > kmax) want = kmax;
    if (want > ks) ks = want;
  }
  a.ksplit = 1;
  a.partial = nullptr;
  if (ks > 1) {
    if (a.tail_ws == nullptr || (int64_t)ks * a.n_out * a.cout * 4 > a.tail_ws_bytes) {
      if ((a.kvol + BIG_KMAX - 1) / BIG_KMAX > 1) return -1;         // cannot run unsplit and there is no scratch for the split
      ks = 1;
    } else {
      a.ksplit = ks;
      a.partial = (float *)a.tail_ws;
    }
  }
  return launch_big<2, 4>(a, st);
}
