// Sparse 3^3 convolution on LDS-resident input WINDOWS.
//
// Why (profiles/README.md, round 2): every byte that crosses the vector-memory path - gathered rows and weight
// tiles alike, L1 / L2 hits included, register or LDS-DMA destination alike - is paid at ~32 B/clk/CU on gfx950, and
// an output-stationary tile of BM rows re-gathers every input row once per kernel offset that touches it: 22 x BM
// row chunks per stage sweep for the decoder's stride-1 map.  But the 27 x 128 neighbour entries of 128 CONSECUTIVE
// output rows name only ~2.5 x 128 DISTINCT input rows (tools/window_stats.py: median 326 / p99 425 on the 683 k-row
// map, exactly 320 on the stride-4 level; the generative decoder emits rows parent-major, i.e. in octree order, so a
// run of rows is a compact brick).  So: per (tile, 32-channel chunk) the distinct rows - the tile's WINDOW - are
// DMA'd into LDS once, and the 27 offsets' MFMA stages read their activation fragments from the window through a
// per-(offset, row) slot map; only the weight tile of each stage still streams.  Gather traffic drops 8-10x and the
// kernel leaves the vector-memory roof for the matrix pipe.
//
//   ph_win_build      per kernel map (cached by the caller next to the neighbour table): for every 128-row tile the
//                     sorted list of distinct input rows + the u16 slot of every (offset, row) entry
//   k_conv_win        the convolution; tiles whose window exceeds the LDS capacity run several passes over disjoint
//                     slot ranges (entries outside the pass read a zero row) - correct for ANY map, fast for local ones
//   device predicate  the window statistics stay on the device (no host read): every eligible convolution launches
//                     both this kernel and the LDS-DMA gather kernel, each with the predicate that makes exactly one
//                     of them return at once (mean passes per tile <= 1.25 -> windows)
//
// Accumulation order = pass, 32-channel chunk, kernel offset (the gather kernels run offset-major): same products,
// different fp32 summation order - compared against the oracle / fp64 at 1e-6-class tolerance, not bit-for-bit.
#include "conv_h2_common.h"

constexpr int WIN_BM = 128;
constexpr int WIN_KV = 27;
constexpr int WIN_CAP = WIN_BM * WIN_KV;      // a tile can never name more distinct rows than it has entries

// ---- window build -------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t win_hash(uint32_t k) {
  k ^= k >> 16;
  k *= 0x7feb352du;
  k ^= k >> 15;
  k *= 0x846ca68bu;
  k ^= k >> 16;
  return k;
}

// stats[i] = sum over the tiles of their window passes at capacity i.  Nearly every tile has ONE pass: the "1" of all tiles
// is added once per launch (by the first workgroup of the kernel that sees every tile) and a tile adds only
// its EXCESS passes - two same-address atomics per tile were what the 5 334-tile build spent most of its time on.
__device__ __forceinline__ void win_count_all_tiles(int32_t *stats) {      // thread 0 of workgroup 0, first thing in the kernel
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    atomicAdd(&stats[0], (int)gridDim.x);
    atomicAdd(&stats[1], (int)gridDim.x);
  }
}
__device__ __forceinline__ void win_count_passes(int32_t *stats, int cnt, int wmax_a, int wmax_b) {
  const int pa = cnt > 0 ? (cnt + wmax_a - 1) / wmax_a : 1, pb = cnt > 0 ? (cnt + wmax_b - 1) / wmax_b : 1;
  if (pa > 1) atomicAdd(&stats[0], pa - 1);
  if (pb > 1) atomicAdd(&stats[1], pb - 1);
}

// one workgroup per 128-row tile.  Fast path: the tile's input rows span < 2^18 consecutive indices (octree-ordered maps:
// median span 44 k on the 683 k-row map) -> a bitmap of the span in LDS; the rank of an index = set bits below it (word
// prefix sums + popcount): no hash probes, no sort.  Wider spans (maps without locality) take the hash set + bitonic sort.
__global__ void __launch_bounds__(256) k_win_build(const int32_t *__restrict__ nbr, int64_t n_out, int32_t *__restrict__ win_rows,
                                                     int32_t *__restrict__ win_cnt, uint16_t *__restrict__ slots,
                                                     int32_t *__restrict__ stats, int wmax_a, int wmax_b, int only_marked) {
  constexpr int HT = 8192;                       // hash slots (>= 2 x WIN_CAP) = bitmap words of the fast path
  constexpr int SPAN_MAX = HT * 32;              // 262 144 indices
  __shared__ int32_t keys[HT];                   // hash keys | bitmap words
  __shared__ int32_t uniq[4096];                 // distinct rows (sort path) | word prefix sums (bitmap path: 8192 u16 halves)
  __shared__ int32_t count, s_min, s_max;
  __shared__ int32_t wsum[256];
  __shared__ int32_t ent[WIN_CAP];               // the tile's 27 x 128 neighbour entries: read from global memory ONCE
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int64_t m0 = tile * WIN_BM;
  if (!only_marked) win_count_all_tiles(stats);
  if (only_marked && win_cnt[tile] != -1) return;      // k_win_build_fast served this tile (it marks the others with -1)
  if (tid == 0) {
    count = 0;
    s_min = 0x7FFFFFFF;
    s_max = -1;
  }
  __syncthreads();
  {   // index span of the tile
    int lo = 0x7FFFFFFF, hi = -1;
    for (int e = tid; e < WIN_CAP; e += 256) {
      const int k = e / WIN_BM, r = e - k * WIN_BM;
      const int64_t row = m0 + r;
      const int idx = row < n_out ? nbr[(int64_t)k * n_out + row] : -1;
      ent[e] = idx;
      if (idx >= 0) {
        lo = idx < lo ? idx : lo;
        hi = idx > hi ? idx : hi;
      }
    }
    if (hi >= 0) {
      atomicMin(&s_min, lo);
      atomicMax(&s_max, hi);
    }
  }
  __syncthreads();
  const int base = s_min, top = s_max;
  int32_t *wr = win_rows + tile * WIN_CAP;
  uint16_t *sl = slots + tile * WIN_CAP;
  if (top < 0 || top - base < SPAN_MAX) {
    // ---- bitmap path ---------------------------------------------------------------------------------------------------
    const int nwords = top < 0 ? 0 : ((top - base) >> 5) + 1;
    for (int i = tid; i < nwords; i += 256) keys[i] = 0;
    __syncthreads();
    for (int e = tid; e < WIN_CAP; e += 256) {
      const int idx = ent[e];
      if (idx >= 0) atomicOr(&keys[(idx - base) >> 5], 1 << ((idx - base) & 31));
    }
    __syncthreads();
    // exclusive prefix of the word popcounts: thread t owns words [t * per, (t + 1) * per)
    const int per = (nwords + 255) / 256;
    int mine = 0;
    for (int w = tid * per; w < (tid + 1) * per && w < nwords; ++w) mine += __popc((unsigned)keys[w]);
    wsum[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {           // Hillis-Steele inclusive scan over the 256 partial sums
      const int v = tid >= d ? wsum[tid - d] : 0;
      __syncthreads();
      wsum[tid] += v;
      __syncthreads();
    }
    const int cnt = wsum[255];
    int run = wsum[tid] - mine;                   // exclusive prefix of this thread's first word
    uint16_t *pre = reinterpret_cast<uint16_t *>(uniq);     // prefix per word (cnt <= 3456 fits 16 bits)
    for (int w = tid * per; w < (tid + 1) * per && w < nwords; ++w) {
      pre[w] = (uint16_t)run;
      unsigned bits = (unsigned)keys[w];
      while (bits) {                              // the distinct rows in ascending order
        const int b = __ffs(bits) - 1;
        wr[run++] = base + (w << 5) + b;
        bits &= bits - 1;
      }
    }
    __syncthreads();
    if (tid == 0) {
      win_cnt[tile] = cnt;
      win_count_passes(stats, cnt, wmax_a, wmax_b);
    }
    for (int e = tid; e < WIN_CAP; e += 256) {
      const int idx = ent[e];
      uint16_t s = 0xFFFFu;
      if (idx >= 0) {
        const int off = idx - base, w = off >> 5;
        s = (uint16_t)(pre[w] + __popc((unsigned)keys[w] & ((1u << (off & 31)) - 1u)));
      }
      sl[e] = s;
    }
    return;
  }
  // ---- hash set + bitonic sort (maps without locality) -------------------------------------------------------------------
  for (int i = tid; i < HT; i += 256) keys[i] = -1;
  __syncthreads();
  // distinct input rows of the tile: LDS hash set, first inserter appends
  for (int e = tid; e < WIN_CAP; e += 256) {
    const int idx = ent[e];
    if (idx < 0) continue;
    uint32_t h = win_hash((uint32_t)idx) & (HT - 1);
    for (;;) {
      const int old = atomicCAS(&keys[h], -1, idx);
      if (old == -1) {
        const int pos = atomicAdd(&count, 1);
        uniq[pos] = idx;
        break;
      }
      if (old == idx) break;
      h = (h + 1) & (HT - 1);
    }
  }
  __syncthreads();
  const int cnt = count;
  // ascending order (input rows of neighbouring output rows are neighbours: consecutive lanes -> consecutive slots)
  int np2 = 1;
  while (np2 < cnt) np2 <<= 1;
  for (int i = cnt + tid; i < np2; i += 256) uniq[i] = 0x7FFFFFFF;
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (np2 >> 1); i += 256) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const int a = uniq[lo], b = uniq[hi];
        if ((a > b) == up) {
          uniq[lo] = b;
          uniq[hi] = a;
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < cnt; i += 256) wr[i] = uniq[i];
  if (tid == 0) {
    win_cnt[tile] = cnt;
    win_count_passes(stats, cnt, wmax_a, wmax_b);
  }
  // slot of every (offset, row) entry: rank of its input row in the sorted list
  for (int e = tid; e < WIN_CAP; e += 256) {
    const int idx = ent[e];
    uint16_t s = 0xFFFFu;
    if (idx >= 0) {
      int lo = 0, hi = cnt - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (uniq[mid] < idx) lo = mid + 1;
        else hi = mid;
      }
      s = (uint16_t)lo;
    }
    sl[e] = s;
  }
}

// The common case in a kernel of its own (round 3): tiles whose input rows span < 2^17 indices (every tile of the decoder's
// octree-ordered maps).  Same bitmap rank as above with 24 KB of LDS instead of 63 (six workgroups per CU instead of two: the
// kernel is bound by the latency of its few dependent phases, not by bytes), the 14 neighbour entries of a thread loaded
// back to back, the span and the word prefix reduced inside the waves (2 barriers instead of 18).  A wider tile is marked
// with win_cnt = -1 and left to k_win_build.
constexpr int WINF_WORDS = 4096;                 // bitmap words: spans below 131 072 indices
__global__ void __launch_bounds__(256) k_win_build_fast(const int32_t *__restrict__ nbr, int64_t n_out, int32_t *__restrict__ win_rows,
                                                          int32_t *__restrict__ win_cnt, uint16_t *__restrict__ slots,
                                                          int32_t *__restrict__ stats, int wmax_a, int wmax_b) {
  constexpr int PER_T = (WIN_CAP + 255) / 256;   // 14 entries per thread, kept in registers through every phase
  __shared__ uint32_t bm[WINF_WORDS];
  __shared__ uint16_t pre[WINF_WORDS];
  __shared__ int32_t s_min, s_max;
  __shared__ int32_t wave_sum[4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t m0 = tile * WIN_BM;
  win_count_all_tiles(stats);
  if (tid == 0) {
    s_min = 0x7FFFFFFF;
    s_max = -1;
  }
  int v[PER_T];
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {              // all loads in flight before the first use
    const int e = tid + 256 * i;
    const int k = e / WIN_BM, r = e - k * WIN_BM;
    const int64_t row = m0 + r;
    v[i] = (e < WIN_CAP && row < n_out) ? nbr[(int64_t)k * n_out + row] : -1;
  }
  __syncthreads();
  int lo = 0x7FFFFFFF, hi = -1;
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    if (v[i] >= 0) {
      lo = v[i] < lo ? v[i] : lo;
      hi = v[i] > hi ? v[i] : hi;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const int ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  if (lane == 0 && hi >= 0) {
    atomicMin(&s_min, lo);
    atomicMax(&s_max, hi);
  }
  __syncthreads();
  const int base = s_min, top = s_max;
  if (top >= 0 && top - base >= WINF_WORDS * 32) {          // uniform: k_win_build takes the tile
    if (tid == 0) win_cnt[tile] = -1;
    return;
  }
  const int nwords = top < 0 ? 0 : ((top - base) >> 5) + 1;
  for (int i = tid; i < nwords; i += 256) bm[i] = 0u;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER_T; ++i)
    if (v[i] >= 0) atomicOr(&bm[(v[i] - base) >> 5], 1u << ((v[i] - base) & 31));
  __syncthreads();
  // exclusive prefix of the word popcounts: thread t owns words [t * per, (t + 1) * per); scan inside the wave, then over
  // the four wave totals
  const int per = (nwords + 255) / 256;
  int mine = 0;
  for (int w = tid * per; w < (tid + 1) * per && w < nwords; ++w) mine += __popc(bm[w]);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  int run = incl - mine;
  for (int w = 0; w < wave; ++w) run += wave_sum[w];
  const int cnt = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
  int32_t *wr = win_rows + tile * WIN_CAP;
  uint16_t *sl = slots + tile * WIN_CAP;
  for (int w = tid * per; w < (tid + 1) * per && w < nwords; ++w) {
    pre[w] = (uint16_t)run;
    unsigned bits = bm[w];
    while (bits) {                                // the distinct rows in ascending order
      const int b = __ffs(bits) - 1;
      wr[run++] = base + (w << 5) + b;
      bits &= bits - 1;
    }
  }
  __syncthreads();
  if (tid == 0) {
    win_cnt[tile] = cnt;
    win_count_passes(stats, cnt, wmax_a, wmax_b);
  }
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int e = tid + 256 * i;
    if (e >= WIN_CAP) continue;
    uint16_t sv = 0xFFFFu;
    if (v[i] >= 0) {
      const int off = v[i] - base, w = off >> 5;
      sv = (uint16_t)(pre[w] + __popc(bm[w] & ((1u << (off & 31)) - 1u)));
    }
    sl[e] = sv;
  }
}

// window capacities (rows) of the two kernel shapes: 64-wide tiles keep two workgroups per CU, 128-wide one
constexpr int WIN_MAX_64 = 416;
constexpr int WIN_MAX_128 = 512;

extern "C" int ph_win_build(const int32_t *nbr, int32_t kvol, int64_t n_out, int32_t *win_rows, int32_t *win_cnt,
                            uint16_t *win_slots, int32_t *win_stats, ph_stream_t stream) {
  PH_REQUIRE(kvol == WIN_KV, "win_build: serves 3x3x3 kernel maps (kvol 27), got %d", kvol);
  PH_REQUIRE(n_out >= 0 && n_out < ((int64_t)1 << 31), "win_build: bad n_out");
  if (n_out == 0) return 0;
  PH_REQUIRE(nbr && win_rows && win_cnt && win_slots && win_stats, "win_build: null buffer");
  const int64_t ntiles = (n_out + WIN_BM - 1) / WIN_BM;
  hipStream_t st = ph_stream(stream);
  PH_CHECK_HIP(hipMemsetAsync(win_stats, 0, 4 * sizeof(int32_t), st));
  static const bool fast_on = [] { const char *e = PH_DEV_ENV("PASCO_WIN_BUILD_FAST"); return e == nullptr || atoi(e) != 0; }();
  if (fast_on) {
    hipLaunchKernelGGL(k_win_build_fast, dim3((unsigned)ntiles), dim3(256), 0, st, nbr, n_out, win_rows, win_cnt, win_slots,
                       win_stats, WIN_MAX_64, WIN_MAX_128);
    PH_LAUNCH_CHECK();
  }
  // every tile (PASCO_WIN_BUILD_FAST=0), or the tiles the fast kernel marked: wide spans, maps without locality
  hipLaunchKernelGGL(k_win_build, dim3((unsigned)ntiles), dim3(256), 0, st, nbr, n_out, win_rows, win_cnt, win_slots,
                     win_stats, WIN_MAX_64, WIN_MAX_128, fast_on ? 1 : 0);
  PH_LAUNCH_CHECK();
  return 0;
}

// ph_conv_desc.route (tests/test_hip_win.py): PH_ROUTE_WIN_ALWAYS = the pair runs on windows (multi-pass path on maps without
// locality), PH_ROUTE_WIN_NEVER = never, neither = the device-side predicate decides

// ---- the convolution -------------------------------------------------------------------------------------------
// WAVES waves as WM x WN: 64-wide tiles run 4 waves (two workgroups per CU), 128-wide tiles 8 waves (one workgroup
// per CU, two waves per SIMD; with 4 waves the single wave per SIMD could not hide its own LDS / barrier latency:
// measured 25 % slower than the gather kernel)
template <int WAVES, int WM, int WN, int TM, int TN, int WMAX, bool EMIT>
__global__ void __launch_bounds__(WAVES * 64, 2) k_conv_win(ConvArgsH a) {
  constexpr int NT = WAVES * 64;
  constexpr int RPP = NT / 8;                       // tile / window rows one DMA pass covers
  constexpr int BM = WIN_BM;
  constexpr int BN = WN * TN * 32;
  static_assert(WM * WN == WAVES && WM * TM * 32 == BM, "tile shape");
  constexpr int B_BYTES = BN * 128;
  constexpr int B_PASSES = BN / RPP;
  constexpr int L = B_PASSES;                       // DMA instructions per thread and stage (weights only)
  constexpr int WIN_BYTES = (WMAX + 1) * 128;       // + the zero row
  constexpr int W_PASSES = WMAX / RPP;
  static_assert(WMAX % RPP == 0 && BN % RPP == 0, "window capacity / tile width vs DMA pass");
  constexpr int OFF_RING = WIN_BYTES;
  constexpr int OFF_SLOT = OFF_RING + 2 * B_BYTES;
  constexpr int OFF_WIDX = OFF_SLOT + WIN_CAP * 2;
  __shared__ __attribute__((aligned(128))) char lds[OFF_WIDX + WMAX * 4];

  const int nwg = gridDim.x;
  const int cpx = nwg >> 3;
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * cpx + (bid >> 3);
  const int ntiles = a.n_row_tiles * a.n_col_tiles;
  if (tile >= ntiles) return;
  if (!ph_win_pred(a.win_stats, a.win_which, a.n_row_tiles)) return;   // the gather kernel serves this map
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
  const uint32_t rsb = 4u * (uint32_t)a.cpad;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- tile constants: slot map -> LDS, zero row -------------------------------------------------------------
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.win_slots + (int64_t)row_tile * WIN_CAP);
    uint4 *dst = reinterpret_cast<uint4 *>(lds + OFF_SLOT);
    for (int i = tid; i < WIN_CAP * 2 / 16; i += NT) dst[i] = src[i];
    if (tid < 8) reinterpret_cast<uint4 *>(lds + WMAX * 128)[tid] = make_uint4(0, 0, 0, 0);
  }
  const int cnt = a.win_cnt[row_tile];
  const int npass = cnt > 0 ? (cnt + WMAX - 1) / WMAX : 1;
  const uint16_t *slot_lds = reinterpret_cast<const uint16_t *>(lds + OFF_SLOT);
  int *widx = reinterpret_cast<int *>(lds + OFF_WIDX);

  // ---- DMA geometry --------------------------------------------------------------------------------------------
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

  auto fire_w = [&](int k, uint32_t coff, int buf) {      // weight tile of offset k (clamped) -> ring buffer
    const int kc = k < WIN_KV - 1 ? k : WIN_KV - 1;
    const uint64_t wk = w_base + (uint64_t)((int64_t)kc * wslab) + coff;
    char *bbuf = lds + OFF_RING + buf * B_BYTES;
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) {
      char *dst = bbuf + (q * RPP + wave * 8) * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(wk + boff[q]),
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
  };

  // ---- fragments ---------------------------------------------------------------------------------------------------
  const int swb = (l31 >> 1) & 7;               // weight tile rows are tile rows: swizzle by the row itself
  uint32_t xb[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int part = 0; part < 2; ++part) xb[ks][part] = (uint32_t)(((part * 4 + ks * 2 + h) ^ swb) << 4);
  const uint32_t brow = (uint32_t)(OFF_RING + (wn * TN * 32 + l31) * 128);
  struct Frag {
    f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  };
  // activation fragments come from the window through the slot map: entry (k, row) -> window row slot - base, or the
  // zero row when the entry is empty / belongs to another pass
  // the slot of stage k + 1 is read one stage ahead (slot_next): the dependent LDS round trip slot -> fragment address is
  // not waited for at the head of every stage
  uint32_t slot_next[TM];
  auto slot_of = [&](int k, int i) -> uint32_t {
    const int kc = k < WIN_KV - 1 ? k : WIN_KV - 1;
    return slot_lds[kc * BM + (wm * TM + i) * 32 + l31];
  };
  auto readfrag = [&](int k, int buf, int base, int wp, Frag &f) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const uint32_t slot = slot_next[i];
      slot_next[i] = slot_of(k + 1, i);
      const uint32_t local = slot - (uint32_t)base;
      const uint32_t wr = local < (uint32_t)wp ? local : (uint32_t)WMAX;
      const uint32_t abase = wr * 128u;
      const uint32_t sw = (wr >> 1) & 7u;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f.ah[ks][i] = *reinterpret_cast<const f16x8 *>(lds + abase + (((uint32_t)(ks * 2 + h) ^ sw) << 4));
        f.al[ks][i] = *reinterpret_cast<const f16x8 *>(lds + abase + (((uint32_t)(4 + ks * 2 + h) ^ sw) << 4));
      }
    }
    const char *bb = lds + buf * B_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f.bh[ks][j] = *reinterpret_cast<const f16x8 *>(bb + brow + j * 4096 + xb[ks][0]);
        f.bl[ks][j] = *reinterpret_cast<const f16x8 *>(bb + brow + j * 4096 + xb[ks][1]);
      }
  };
  auto mfma = [&](const Frag &f) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.al[ks][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
        }
  };
  Frag f0, f1;
  for (int pass = 0; pass < npass; ++pass) {
    const int base = pass * WMAX;
    const int wp = cnt - base < WMAX ? cnt - base : WMAX;
    // input rows of this pass -> LDS (ordinary loads, outside the DMA-counted region)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const int32_t *wr = a.win_rows + (int64_t)row_tile * WIN_CAP + base;
      for (int i = tid; i < WMAX; i += NT) widx[i] = i < wp ? wr[i] : -1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const uint32_t coff = (uint32_t)chunk << 7;
      // weights of offsets 0 and 1 first: they fly under the window load
      fire_w(0, coff, 0);
      fire_w(1, coff, 1);
      // the window: row w of the pass -> LDS row w (lane-linear), source chunk swizzled by w
#pragma unroll
      for (int p = 0; p < W_PASSES; ++p) {
        if (p * RPP < wp) {                        // uniform
          const int ix = widx[p * RPP + l_r];
          uint64_t v = in_base + (uint64_t)(uint32_t)(ix < 0 ? 0 : ix) * rsb + coff;
          asm volatile("" : "+v"(v));
          const uint64_t src = ix >= 0 ? v : zero_src;
          char *dst = lds + (p * RPP + wave * 8) * 128;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                           (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // offset loop, ONE barrier per stage (a barrier costs ~150 clk here: two per stage were a third of the loop):
      //   fragments of stage k + 1 are read from ring buffer (k + 1) & 1 while the matrix pipe works on stage k;
      //   then one rendezvous certifies both "every wave has finished reading that buffer" (lgkmcnt) and "every wave's
      //   weight DMA of stage k + 2 has landed in the other buffer" (vmcnt) - after it the buffer just read is refilled
      //   with stage k + 3 and the next iteration may read stage k + 2.
#pragma unroll
      for (int i = 0; i < TM; ++i) slot_next[i] = slot_of(0, i);
      readfrag(0, 0, base, wp, f0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      fire_w(2, coff, 0);
#define WIN_STAGE_END()                                                   \
  do {                                                                    \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_s_barrier();                                         \
  } while (0)
      for (int k = 0; k < WIN_KV; k += 2) {
        readfrag(k + 1, 1, base, wp, f1);
        __builtin_amdgcn_sched_barrier(0);
        mfma(f0);
        __builtin_amdgcn_sched_barrier(0);
        WIN_STAGE_END();
        fire_w(k + 3, coff, 1);
        readfrag(k + 2, 0, base, wp, f0);
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < WIN_KV) mfma(f1);
        __builtin_amdgcn_sched_barrier(0);
        WIN_STAGE_END();
        fire_w(k + 4, coff, 0);
      }
#undef WIN_STAGE_END
      // drain the clamped tail loads before the window and the ring are reused
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  h2_store_tile<TM, TN, EMIT>(a, acc, m0, n0, wm, wn, h, l31);
}

template <int WAVES, int WM, int WN, int TM, int TN, int WMAX>
static int launch_win(const ConvArgsH &a, hipStream_t st) {
  constexpr int BN = WN * TN * 32;
  ConvArgsH args = a;
  args.n_row_tiles = (int)((a.n_out + WIN_BM - 1) / WIN_BM);
  args.n_col_tiles = (a.cout + BN - 1) / BN;
  const int ntiles = args.n_row_tiles * args.n_col_tiles;
  const int grid = ((ntiles + 7) / 8) * 8;
  const bool emit = args.out_split != nullptr;
  if (emit)
    hipLaunchKernelGGL((k_conv_win<WAVES, WM, WN, TM, TN, WMAX, true>), dim3(grid), dim3(WAVES * 64), 0, st, args);
  else
    hipLaunchKernelGGL((k_conv_win<WAVES, WM, WN, TM, TN, WMAX, false>), dim3(grid), dim3(WAVES * 64), 0, st, args);
  PH_LAUNCH_CHECK();
  return 0;
}

// Launches the window kernel (it returns at once on maps the predicate hands to the gather kernel).  bn: 64 or 128.
int ph_conv_win_launch(const ConvArgsH &a, int bn, hipStream_t st) {
  ConvArgsH b = a;
  b.ksplit = 1;
  b.partial = nullptr;
  b.zero = ph_dma_zero_line();
  b.ablate = ph_dma_ablate_bits();
  if (b.zero == nullptr) {
    ph_set_error("conv_fwd(windows): no zero line");
    return 2;
  }
  if (bn == 64) {
    b.win_which = 0 | ph_win_force_bits(b.route);
    // 64-wide outputs: offset-parallel waves on 16-channel chunks (conv_wop.hip, round 6) when the caller gave the fragment-order
    // copy of the kernel; else (and PASCO_WOP=0 in the development build) the row-parallel window kernel below
    static const bool wop2 = [] { const char *e = PH_DEV_ENV("PASCO_WOP"); return e == nullptr || atoi(e) != 0; }();
    if (wop2 && b.cout <= 64 && b.w_frag != nullptr) {
      ConvArgsH args = b;
      args.n_row_tiles = (int)((b.n_out + WIN_BM - 1) / WIN_BM);
      args.n_col_tiles = 1;
      return ph_conv_wop2_launch(args, st);
    }
    return launch_win<4, 4, 1, 1, 2, WIN_MAX_64>(b, st);
  }
  b.win_which = 1 | ph_win_force_bits(b.route);
  return launch_win<8, 2, 4, 2, 1, WIN_MAX_128>(b, st);
}
