// Sparse convolution forward, split-precision products, operands streamed global -> LDS by the LDS-DMA path
// (`global_load_lds_dwordx4`) in a software pipeline with counted `vmcnt`:
//
//   * no staging registers and no ds_write pass: a gathered row chunk (128 B = [32 hi | 32 lo] f16 of one voxel)
//     lands in LDS as 8 lanes x 16 B of one wave instruction; the weights' tile rows the same way;
//   * two LDS stage buffers + register double-buffered MFMA fragments = three stages deep: while the matrix
//     pipe works on stage s from registers, the fragments of stage s+1 are read from LDS and the DMA of stage
//     s+2 (issued one step earlier) is in flight; `s_waitcnt vmcnt(L)` (L = DMA instructions per stage and
//     thread) retires exactly the stage about to be read and leaves the younger one outstanding across the
//     barrier (MI355X_MICROARCH.md "LDS-DMA requests stay in flight across s_barrier");
//   * the loop holds NO other vector-memory instruction (hipcc would answer any ordinary load with vmcnt(0),
//     cdna_hip_programming.md section 5): the neighbour indices of the tile are parked in LDS once, up front;
//   * LDS tile rows are 128 B without padding (the DMA destination is lane-linear); the 16-byte chunk c of tile
//     row r holds operand chunk c ^ ((r >> 1) & 7) - applied to the per-lane SOURCE address and to the fragment
//     reads - which makes every ds_read_b128 lane group hit 16 distinct bank quads;
//   * ONE __shared__ object (a second one makes the compiler drain vmcnt before every k-step).
//
// Measured motivation (tools/ubench, profiles/r2a_gather_bench.json): the bare gather of the biggest 64-channel
// layer takes ~105 us by this path against 630 us for the whole k_conv_h2 launch whose matrix work is ~150 us:
// the register-staged two-barrier loop serialises load, LDS store and MFMA phases instead of overlapping them.
//
// Same tile algebra, accumulation order and epilogue as k_conv_h2 (conv_h2_common.h): results are bit-identical.
#include <mutex>
#include <stdlib.h>

#include <hip/hip_runtime.h>
#include "ph_common.h"
#ifdef PH_DEV
// development build only (tools/dma_trace.py): shader-clock stamps of the first 64 workgroups' phases when bit 7 of the ablate mask is set
__device__ unsigned long long g_dma_trace[64 * 8];
#define DMA_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if ((a.ablate & 0x80) && threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0)                              \
      g_dma_trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();                                           \
  } while (0)
#else
#define DMA_STAMP(i) do { } while (0)
#endif
#define H2_STAMP(i) DMA_STAMP(i)
#include "conv_h2_common.h"

// kernel volumes above this are the bottleneck's (7, 7, 5) implicit GEMMs: their empty (tile, offset) stages are dropped and their
// slices interleaved (measured: 544 -> 478 us on the 245-offset products; the 75-offset ones, whose light and heavy tiles cannot
// balance over 4 slices, lose 4 % to the bookkeeping and stay as they were: profiles/r4p_layer_ab_dense_stage_skip.txt)
#ifdef PH_DEV
extern "C" int ph_dma_trace_read(unsigned long long *host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dma_trace), sizeof(g_dma_trace)) == hipSuccess ? 0 : 2;
}
#endif
constexpr int DMA_DENSE_KVOL = 100;
constexpr int DMA_KMAX = 32;   // kernel offsets one workgroup walks (its slice of the split over the offsets)

#ifdef PH_DEV
// Development build only: bit 8 of the mask selects the 256-row / 8-wave tiles (measured no faster: one workgroup per CU convoys).
// The phase-ablation bits of round 2 (skip MFMAs / gathers / fragment reads; profiles/r2c - r2e) are gone from the kernel:
// their runtime branches cost ~3 % in the loop.
static int g_dma_ablate = 0;
static int g_dma_tall = 0;
int ph_dma_ablate_bits() { return g_dma_ablate; }
extern "C" void ph_conv_dma_set_ablate(int mask) { g_dma_ablate = mask & 0xff; g_dma_tall = (mask & 0x100) ? 1 : 0; }
#else
constexpr int g_dma_ablate = 0, g_dma_tall = 0;     // the product library has no experiment state
int ph_dma_ablate_bits() { return 0; }
#endif

// One workgroup = WAVES (4 or 8) waves as WM x WN, tile BM = WM*TM*32 rows (128 / 256) x BN = WN*TN*32 channels,
// 32 input channels per stage.  Everything that crosses the vector-memory path (gathered rows AND the weight tile of
// every workgroup) is paid at ~32 B/clk/CU whether it hits L1 / L2 or not (profiles/README.md, k_conv_dma ablation):
// the 256-row tile halves the weight bytes per output row.
template <int WAVES, int WM, int WN, int TM, int TN, bool EMIT, bool INTER = false>
__global__ void __launch_bounds__(WAVES * 64, 2) k_conv_dma(ConvArgsH a) {
  constexpr int NT = WAVES * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int RPP = NT / 8;                // tile rows one DMA pass covers: every wave 8 rows x 128 B
  static_assert(WM * WN == WAVES && BM == 4 * RPP && BN % RPP == 0, "tile shape");
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_PASSES = BM / RPP;
  constexpr int B_PASSES = BN / RPP;
  constexpr int L = A_PASSES + B_PASSES;     // DMA instructions per thread and stage
  __shared__ __attribute__((aligned(128))) char lds[2 * STAGE + DMA_KMAX * BM * 4];

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
  DMA_STAMP(0);
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int cout = a.cout;
  const int nchunks = a.cpad >> 5;
  const int kper = (a.kvol + a.ksplit - 1) / a.ksplit;
  int k_begin = (int)blockIdx.y * kper;
  const int k_end = (k_begin + kper < a.kvol) ? k_begin + kper : a.kvol;
  int kcount = k_end > k_begin ? k_end - k_begin : 0;
  // dense implicit GEMMs (kvol > DMA_DENSE_KVOL) slice the offsets INTERLEAVED (slice s walks s, s + ksplit, ...): which offsets of a tile
  // are empty depends on the tile's z plane and on the offset's dz, the slowest-running index of the enumeration - contiguous
  // slices would be all-empty or all-full, and a launch is as long as its fullest workgroups
  int kstride = 1;
  if (a.kvol > DMA_DENSE_KVOL && a.tile_k == nullptr && a.ksplit > 1 && !PH_ABLATE(a, 4)) {
    kstride = a.ksplit;
    k_begin = (int)blockIdx.y;
    kcount = k_begin < a.kvol ? (a.kvol - k_begin + kstride - 1) / kstride : 0;
  }
  if (a.tile_k != nullptr) {          // row lists: this row tile is ONE kernel offset's k = 1 product (-1: unused tile)
    k_begin = a.tile_k[row_tile];
    if (k_begin < 0) return;
    kcount = 1;
  }
  int nstages = kcount * nchunks;
  const uint32_t rsb = 4u * (uint32_t)a.cpad;   // bytes per operand row

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nstages > 0) {
    // ---- neighbour indices of the tile -> LDS (the only ordinary loads before the epilogue) ----------------
    // layout [offset][l_r = tile row mod RPP][p = tile row / RPP]: the four rows one thread feeds are one 16-byte read
    int *idx_lds = reinterpret_cast<int *>(lds + 2 * STAGE);
    for (int i = tid; i < kcount * BM; i += NT) {
      const int k = i / BM, r = i - k * BM;
      const int64_t row = m0 + r;
      int idx = -1;
      if (row < a.n_out) idx = a.tile_k ? a.nbr[row] : (a.nbr ? a.nbr[(int64_t)(k_begin + k * kstride) * a.nbr_stride + row] : (int)row);
      idx_lds[k * BM + (r % RPP) * 4 + (r / RPP)] = idx;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DMA_STAMP(7);

    // ---- dense implicit GEMMs (the bottleneck's 245-offset products): offsets none of the tile's rows has a neighbour at
    // (beyond the grid's faces: the bottleneck orders its sites z-major so that a 128-row tile lies in ONE z plane, and
    // 30 % of the (tile, offset) stages of the (7, 7, 5) kernels are then empty) are dropped from the stage
    // list: the walk goes over kmap[0 .. nvalid).  Empty stages add exact zeros, so the sums do not change -----------------
    // the list lives in the LAST offset slot of the index table (such launches walk at most DMA_KMAX - 1 offsets, ph_conv_dma_try:
    // the tile's 80 KB of LDS are what lets two workgroups share a CU - not one byte more)
    static_assert(BM >= 2 * DMA_KMAX + 4, "the stage list fits one offset slot of the index table");
    int *kmap = idx_lds + (DMA_KMAX - 1) * BM;         // [DMA_KMAX] valid offset slots, then [DMA_KMAX] flags, then the count
    const bool compact = a.kvol > DMA_DENSE_KVOL && a.tile_k == nullptr && !PH_ABLATE(a, 4);
    if (compact) {
      int *kflag = kmap + DMA_KMAX;
      for (int k = wave; k < kcount; k += WAVES) {
        bool any = false;
        for (int r = lane; r < BM; r += 64) any |= idx_lds[k * BM + r] >= 0;
        const unsigned long long bal = __ballot(any);
        if (lane == 0) kflag[k] = bal != 0ull ? 1 : 0;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) {
        int nv = 0;
        for (int k = 0; k < kcount; ++k)
          if (kflag[k]) kmap[nv++] = k;
        kflag[DMA_KMAX] = nv;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      nstages = kflag[DMA_KMAX] * nchunks;
    }
    if (nstages > 0) {

    // ---- DMA geometry of this thread: tile row l_r + RPP p, 16-byte slot l_j; source chunk swizzled ---------
    const int l_j = tid & 7;
    const int l_r = tid >> 3;
    const uint32_t sj16 = (uint32_t)((l_j ^ ((l_r >> 1) & 7)) << 4);
    const uint64_t in_base = (uint64_t)reinterpret_cast<uintptr_t>(a.in_split) + sj16;
    const uint64_t zero_src = (uint64_t)reinterpret_cast<uintptr_t>(a.zero) + sj16;
    const uint64_t w_base = (uint64_t)reinterpret_cast<uintptr_t>(a.w_split) + sj16;
    uint32_t boff[B_PASSES];   // byte offset of this thread's weight rows inside one offset's [cout][rs] slab
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) {
      int n = n0 + l_r + q * RPP;
      n = n < cout ? n : cout - 1;   // columns >= cout are never stored: any finite row will do
      boff[q] = (uint32_t)n * rsb;
    }
    const int64_t wslab = (int64_t)cout * rsb;
    static_assert(A_PASSES == 4, "one 16-byte index read per thread and stage");
    typedef int i32x4 __attribute__((ext_vector_type(4)));

    // stage s -> (offset slot, channel chunk); the tail re-loads the last stage instead of branching
    auto stage_kc = [&](int s, int &k, uint32_t &coff) {
      const int sc = s < nstages - 1 ? s : nstages - 1;
      const int kk = sc / nchunks;
      coff = (uint32_t)(sc - kk * nchunks) << 7;
      k = compact ? kmap[kk] : kk;
    };
    auto load_idx = [&](int s) {
      int k;
      uint32_t coff;
      stage_kc(s, k, coff);
      return *reinterpret_cast<const i32x4 *>(idx_lds + k * BM + l_r * 4);
    };
    struct Src {
      uint64_t a[A_PASSES];
      uint64_t w;
    };
    auto prep = [&](int s, const i32x4 &idx, Src &src) {   // VALU only: source addresses of stage s
      int k;
      uint32_t coff;
      stage_kc(s, k, coff);
#pragma unroll
      for (int p = 0; p < A_PASSES; ++p) {
        const int ix = idx[p];
        uint64_t v = in_base + (uint64_t)(uint32_t)(ix < 0 ? 0 : ix) * rsb + coff;
        asm volatile("" : "+v"(v));            // materialise before the select: a select, not a branch per row
        src.a[p] = ix >= 0 ? v : zero_src;
      }
      src.w = w_base + (uint64_t)((int64_t)(k_begin + k * kstride) * wslab) + coff;
    };
    auto fire = [&](const Src &src, int buf) {
      char *abuf = lds + buf * STAGE;
#pragma unroll
      for (int p = 0; p < A_PASSES; ++p) {
        char *dst = abuf + (p * RPP + wave * 8) * 128;   // wave-uniform; the DMA adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src.a[p],
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < B_PASSES; ++q) {
        char *dst = abuf + A_BYTES + (q * RPP + wave * 8) * 128;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(src.w + boff[q]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
    };

    // ---- fragment reads: lane = (tile row l31 of a 32-row block, k-half h); chunk c of row r sits at c ^ sw ---
    const int sw = (l31 >> 1) & 7;
    uint32_t xo[2][2];   // [ks][hi / lo] byte offset of the lane's 16-byte run inside its tile row
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int part = 0; part < 2; ++part) xo[ks][part] = (uint32_t)(((part * 4 + ks * 2 + h) ^ sw) << 4);
    const uint32_t arow = (uint32_t)((wm * TM * 32 + l31) * 128);
    const uint32_t brow = (uint32_t)(A_BYTES + (wn * TN * 32 + l31) * 128);

    struct Frag {
      f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    };
    auto readfrag = [&](int buf, Frag &f) {
      const char *base = lds + buf * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f.ah[ks][i] = *reinterpret_cast<const f16x8 *>(base + arow + i * 4096 + xo[ks][0]);
          f.al[ks][i] = *reinterpret_cast<const f16x8 *>(base + arow + i * 4096 + xo[ks][1]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f.bh[ks][j] = *reinterpret_cast<const f16x8 *>(base + brow + j * 4096 + xo[ks][0]);
          f.bl[ks][j] = *reinterpret_cast<const f16x8 *>(base + brow + j * 4096 + xo[ks][1]);
        }
      }
    };
    auto mfma = [&](const Frag &f) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            // weights first: transposed accumulator block (lane = output row); smallest terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.al[ks][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
          }
    };
    // INTER: the DMA instructions of a stage issued BETWEEN its matrix instructions (one DMA, then its share of the MFMAs)
    // instead of all in front of them
    auto fire_mfma = [&](const Src &src, int buf, const Frag &f) {
      char *abuf = lds + buf * STAGE;
      constexpr int ND = A_PASSES + B_PASSES, NM = 2 * TM * TN * 3;
#pragma unroll
      for (int t = 0; t < ND; ++t) {
        if (t < A_PASSES) {
          char *dst = abuf + (t * RPP + wave * 8) * 128;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src.a[t],
                                           (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        } else {
          const int q = t - A_PASSES;
          char *dst = abuf + A_BYTES + (q * RPP + wave * 8) * 128;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(src.w + boff[q]),
                                           (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = t * NM / ND; m < (t + 1) * NM / ND; ++m) {
          const int p = m % 3, ij = (m / 3) % (TM * TN), ks = m / (3 * TM * TN);
          const int i = ij / TN, j = ij % TN;
          if (p == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.al[ks][i], acc[i][j], 0, 0, 0);
          else if (p == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // wait until at most L DMA instructions (= the younger stage) are outstanding, then rendezvous
#define DMA_WAIT_STAGE()                                                  \
  do {                                                                    \
    if (L == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");          \
    else if (L == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     \
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                 \
    __builtin_amdgcn_s_barrier();                                         \
  } while (0)
#define DMA_READS_DONE()                                  \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                         \
  } while (0)
    static_assert(L == 5 || L == 6 || L == 8, "vmcnt immediates");

    Frag f0, f1;
    Src src;
    i32x4 ix;
    // ---- short launches (k = 1 products: linear layers, row-list tiles; nstages = cin / 32 <= 8): the long pipeline below
    // keeps its loop branch-free by re-loading the last stage up to three more times and draining them before the epilogue -
    // for a two-stage launch that is 5 stage loads for 2 stages of work.  Here every stage is loaded exactly once: the DMA of
    // stage s + 2 goes into the buffer stage s was read from, the last waits are vmcnt(0) -----------------------------------
    if (nstages <= 8 && !PH_ABLATE(a, 1)) {
      auto wait_landed = [&](bool younger_in_flight) {
        if (younger_in_flight) {
          if (L == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          else if (L == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      };
      prep(0, load_idx(0), src);
      fire(src, 0);
      if (nstages > 1) {
        prep(1, load_idx(1), src);
        fire(src, 1);
      }
      DMA_STAMP(1);
      for (int s = 0; s < nstages; ++s) {
        const int buf = s & 1;
        wait_landed(s + 1 < nstages);                  // stage s landed (stage s + 1 may still fly)
        if (s == 0) DMA_STAMP(2);
        readfrag(buf, f0);
        if (s + 2 < nstages) {                         // buffer `buf` is free once every wave holds its fragments
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          prep(s + 2, load_idx(s + 2), src);
          fire(src, buf);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma(f0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    prep(0, load_idx(0), src);
    fire(src, 0);
    prep(1, load_idx(1), src);
    fire(src, 1);
    ix = load_idx(2);
    DMA_WAIT_STAGE();            // stage 0 landed (stage 1 may still fly)
    prep(2, ix, src);
    readfrag(0, f0);
    DMA_READS_DONE();            // every wave has its stage-0 fragments: buffer 0 is free
    fire(src, 0);
    ix = load_idx(3);            // index reads run one step ahead of the address arithmetic that consumes them
    for (int s = 0; s < nstages; s += 2) {
      // stage s from f0; fragments of stage s + 1 from buffer 1, then the DMA of stage s + 3 into buffer 1 is FIRED
      // BEFORE the matrix work of stage s so that it flies underneath it
      DMA_WAIT_STAGE();
      prep(s + 3, ix, src);
      readfrag(1, f1);
      DMA_READS_DONE();
      if (INTER) {
        ix = load_idx(s + 4);
        __builtin_amdgcn_sched_barrier(0);
        fire_mfma(src, 1, f0);
      } else {
        fire(src, 1);
        ix = load_idx(s + 4);
        __builtin_amdgcn_sched_barrier(0);
        mfma(f0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // stage s + 1 from f1; fragments of stage s + 2 from buffer 0; DMA of stage s + 4 into buffer 0
      DMA_WAIT_STAGE();
      prep(s + 4, ix, src);
      readfrag(0, f0);
      DMA_READS_DONE();
      if (INTER) {
        ix = load_idx(s + 5);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nstages) fire_mfma(src, 0, f1);
        else fire(src, 0);
      } else {
        fire(src, 0);
        ix = load_idx(s + 5);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nstages) mfma(f1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // clamped tail loads: nothing may land after the epilogue starts
    }
#undef DMA_WAIT_STAGE
#undef DMA_READS_DONE
    }
  }

  DMA_STAMP(3);
  // every load of the epilogue ahead of its first store (conv_h2_common.h, ParGlobal): 600 -> 510 us on the 64 -> 384 projections
  // together with the epilogue's VALU diet, 547 without the staging (profiles/r3s_layer_ab_epilogue_diet_vs_base.txt)
  h2_store_tile_staged<TM, TN, EMIT, NT, BN, 2 * STAGE + DMA_KMAX * BM * 4>(a, acc, m0, n0, wm, wn, h, l31, tid, lds);
  if (PH_ABLATE(a, 0x80)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DMA_STAMP(4);
}

template <int WAVES, int WM, int WN, int TM, int TN>
static int launch_dma(const ConvArgsH &a, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  ConvArgsH args = a;
  args.n_row_tiles = (int)((a.n_out + BM - 1) / BM);
  args.n_col_tiles = (a.cout + BN - 1) / BN;
  const int ntiles = args.n_row_tiles * args.n_col_tiles;
  const int grid = ((ntiles + 7) / 8) * 8;
  const bool emit = args.out_split != nullptr && args.ksplit == 1;
  static const bool inter = [] { const char *e = PH_DEV_ENV("PASCO_CONV_DMA_INTER"); return e == nullptr || atoi(e) != 0; }();
  if (inter && WAVES == 4) {     // default: DMA instructions between the MFMAs (5-10 % on every layer; =0: in front of them)
    if (emit)
      hipLaunchKernelGGL((k_conv_dma<WAVES, WM, WN, TM, TN, true, true>), dim3(grid, 1), dim3(WAVES * 64), 0, st, args);
    else
      hipLaunchKernelGGL((k_conv_dma<WAVES, WM, WN, TM, TN, false, true>), dim3(grid, args.ksplit), dim3(WAVES * 64), 0, st, args);
  } else if (emit)
    hipLaunchKernelGGL((k_conv_dma<WAVES, WM, WN, TM, TN, true>), dim3(grid, 1), dim3(WAVES * 64), 0, st, args);
  else
    hipLaunchKernelGGL((k_conv_dma<WAVES, WM, WN, TM, TN, false>), dim3(grid, args.ksplit), dim3(WAVES * 64), 0, st, args);
  PH_LAUNCH_CHECK();
  if (args.ksplit > 1) {
    if (int rc = ph_launch_splitk_epilogue(args, st)) return rc;
  }
  ph_record_cfg(2, BM, BN, 32, args.ksplit, emit ? 1 : 0, 4, WAVES);
  return 0;
}

// 256 zero bytes per device for rows without a neighbour: the one piece of device memory the library owns
// (allocated at the first launch on a device, never inside a stream capture: convolutions are not captured)
const char *ph_dma_zero_line() {
  static const char *zero[64] = {nullptr};
  static std::mutex mu;                     // serving loops launch from several threads
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (zero[dev] == nullptr) {
    void *p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;   // once per device
    zero[dev] = (const char *)p;
  }
  return zero[dev];
}

// scratch the caller offered for splits over the kernel offsets (ph_conv_desc.splitk_ws), carried in the argument block
static inline void *g_tail_ws(const ConvArgsH &a) { return a.tail_ws; }
static inline int64_t g_tail_ws_bytes(const ConvArgsH &a) { return a.tail_ws_bytes; }

// Takes the launch when the shape fits the DMA pipeline; returns -1 when the caller should use k_conv_h2.
// `a` arrives fully prepared (tile-independent fields, ksplit / partial chosen by the caller for 128-row tiles).
int ph_conv_dma_try(const ConvArgsH &a_in, int bn, hipStream_t st) {
  const int kper = (a_in.kvol + a_in.ksplit - 1) / a_in.ksplit;
  if (kper > DMA_KMAX - (a_in.kvol > DMA_DENSE_KVOL ? 1 : 0)) return -1;     // dense GEMMs: the last index slot holds the stage list
  const char *zero = ph_dma_zero_line();
  if (zero == nullptr) return -1;
  ConvArgsH a = a_in;
  a.zero = zero;
  a.ablate = g_dma_ablate & 0x85;      // experiment switches (tools/layer_ab.py): bit 0 = the long pipeline for short launches too,
                                    // bit 2 = no dropping of empty (tile, offset) stages in the dense implicit GEMMs
  // timing experiments (wrong results by design): 0x10 drops the per-axis table residual, 0x40 the dense residual
  if (g_dma_ablate & 0x10) a.axis_table = nullptr;
  if (g_dma_ablate & 0x40) a.residual = nullptr;
  // 256-row tiles (8 waves, one workgroup per CU) when they still cover every CU about twice; else 128-row tiles
  const int64_t ncol = (a.cout + bn - 1) / bn;
  const bool tall = bn >= 64 && ((a.n_out + 255) / 256) * ncol * a.ksplit >= 2 * 256 && g_dma_tall;
  if (bn == 32) return launch_dma<4, 4, 1, 1, 1>(a, st);
  if (bn == 64) return tall ? launch_dma<8, 8, 1, 1, 2>(a, st) : launch_dma<4, 4, 1, 1, 2>(a, st);
  if (tall) return launch_dma<8, 4, 2, 2, 2>(a, st);
  // ---- tail split (128-wide tiles, two workgroups per CU = 512 resident tiles): every tile of a launch does the same work
  // (all kvol offsets, whatever the map's sparsity), so a launch of T tiles runs in ceil(T / 512) rounds and a small last
  // round leaves most of the chip idle for a whole tile time (559 tiles: 2 rounds for 1.09 rounds of work).  When fewer than
  // half a round of row tiles is left over, they go into a second launch that is split over the kernel offsets so that it
  // fills the chip with short workgroups; only those rows pay the partial-sum reduction. ----------------------------------
  static const bool tail_on = [] { const char *e = PH_DEV_ENV("PASCO_CONV_TAIL"); return e == nullptr || atoi(e) != 0; }();
  const int64_t trow = (a.n_out + 127) / 128;
  const int64_t slots = 512 / ncol;               // row tiles of one full round
  const int64_t rounds = trow / slots, rest = trow - rounds * slots;
  if (tail_on && a.ksplit == 1 && a.tile_k == nullptr && a.nbr != nullptr && a.kvol >= 8 && !a.win_gather && rounds >= 1 &&
      rest > 0 && 2 * rest <= slots && a_in.partial == nullptr && g_tail_ws(a_in) != nullptr) {
    int ks = (int)(slots / rest);
    if (ks > a.kvol / 3) ks = a.kvol / 3;
    if (ks > 16) ks = 16;
    // no empty slices: the kernel gives every slice ceil(kvol / ks) offsets, so e.g. 27 offsets in 8 slices of 4 leaves the
    // eighth with nothing - its workgroups would only write (and the reduction read) zeros
    while (ks > 1 && (a.kvol + (a.kvol + ks - 1) / ks - 1) / ((a.kvol + ks - 1) / ks) != ks) --ks;
    const int64_t r0 = (trow - rest) * 128, tail_rows = a.n_out - r0;
    if (ks >= 2 && (int64_t)ks * tail_rows * a.cout * 4 <= g_tail_ws_bytes(a_in)) {
      ConvArgsH head = a;
      head.n_out = r0;
      if (int rc = launch_dma<4, 2, 2, 2, 2>(head, st)) return rc;
      ConvArgsH tail = a;
      tail.n_out = tail_rows;
      tail.nbr = a.nbr + r0;
      if (tail.out) tail.out = a.out + r0 * a.cout;
      if (tail.out_split) tail.out_split = a.out_split + r0 * a.cout * 2;
      if (tail.residual) tail.residual = a.residual + r0 * a.cout;
      if (tail.axis_coords) tail.axis_coords = a.axis_coords + r0 * 4;
      tail.ksplit = ks;
      tail.partial = (float *)g_tail_ws(a_in);
      return launch_dma<4, 2, 2, 2, 2>(tail, st);     // records its configuration: the launch reads as "ksplit"
    }
  }
  return launch_dma<4, 2, 2, 2, 2>(a, st);
}

#ifdef PH_DEV
// development build only (tools/occupancy.py): resident workgroups per CU of the main instantiations
extern "C" int ph_conv_dma_occupancy(int which) {
  int n = -1;
  hipError_t e = hipSuccess;
  switch (which) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_dma<4, 2, 2, 2, 2, false>, 256, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_dma<4, 2, 2, 2, 2, true>, 256, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_dma<4, 4, 1, 1, 2, false>, 256, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_dma<4, 4, 1, 1, 1, false>, 256, 0); break;
    default: break;
  }
  return e == hipSuccess ? n : -1;
}
#endif
