// Sparse 3^3 convolution of the 64-channel levels on LDS-resident input windows, round 6 (`k_conv_wop2`, kernel id 5 like the
// kernel it replaces: the window side of a window / gather pair).
//
// What held k_conv_wop (conv_win.hip, round 4) at 0.25 of the f16 matrix peak (profiles/r4e_wop_trace_v3.txt, r5zz PMC: matrix
// pipe 50 % busy): 35 % of a workgroup's life lies OUTSIDE its offset loops - the tile's tables (one round trip), the window
// DMA of each 32-channel chunk (nothing to multiply while it flies), the reduction, the epilogue - and inside the loop a wave
// alone on its SIMD (its partner workgroup being in one of those phases) stalls at the head of every offset on the LDS reads
// it has just issued.  Same arithmetic here (offset-parallel waves: wave w owns offsets w, w + 4, ..., every weight fragment
// global -> registers, 128 rows x 64 columns of accumulators per wave, partial sums added through LDS in a fixed order), but:
//   * 16-CHANNEL window chunks in TWO LDS buffers: the window of chunk c + 1 (64 bytes per row: 16 hi | 16 lo values) is
//     DMA'd while chunk c multiplies - one DMA instruction per wave at the head of each of the 7 offsets, every one of them
//     issued unconditionally (rows beyond the window read the zero line) so that the counted vmcnt waits are constants;
//   * the offsets of a chunk are straight-line code (7 x 24 MFMAs, no branch, no loop-carried fragment copies): the
//     fragments of offset t + 1 are requested under the products of offset t - rows 0..63 before the second group of 12
//     MFMAs, rows 64..127 after it -, the slots of offset t + 1 at its head, the weights one whole offset ahead;
//   * T tiles per workgroup: the next tile's slot map and row list are DMA'd into a second table buffer during the
//     current tile, its first window during the current tile's last chunk - the set-up round trips are paid once per
//     workgroup, and the next tile's window lands under the reduction and the epilogue's stores.
// Wave 3 owns 6 offsets; its seventh (k = 27) multiplies the zero row: no branch in the body.
//
// Summation order per accumulator: pass, 16-channel chunk, own offsets ascending; then across waves (own part first, the other
// waves' ascending) - another fp32 order than k_conv_wop's (32-channel chunks), the same products; held to the oracle / fp64
// at the tolerance of tests/test_hip_win.py.
#include "conv_h2_common.h"

constexpr int W2_BM = 128;
constexpr int W2_KV = 27;
constexpr int W2_CAP = W2_BM * W2_KV;              // entries of a tile's slot map / row list (ph_win_build's layout)
constexpr int W2_MAX = 448;                        // window rows per pass: 7 DMA passes of 64 rows
constexpr int W2_ROWB = 64;                        // bytes of a window row
constexpr int W2_WIN = (W2_MAX + 1) * W2_ROWB;     // + the zero row (index W2_MAX)
constexpr int W2_TAB_SLOT = W2_CAP * 2;            // u16 slot map
constexpr int W2_TAB = W2_TAB_SLOT + W2_MAX * 4;   // + the first W2_MAX entries of the row list
constexpr int W2_PIECES = W2_TAB / 16;             // 16-byte pieces of a table buffer (544)
static_assert(W2_TAB % 16 == 0 && W2_WIN % 64 == 0 && W2_TAB_SLOT % 16 == 0, "alignment of the LDS regions");

__device__ __forceinline__ f16x8 w2_gld(const char *p) {
  f16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ f16x8 w2_lds(uint32_t addr) {
  f16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t w2_lds16(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ int w2_lds32(uint32_t addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
struct W2W {                 // weight fragments of one (offset, chunk): [column block][hi, lo]
  f16x8 bh[2], bl[2];
};
struct W2A {                 // activation fragments of two row blocks
  f16x8 h[2], l[2];
};
#define W2_PIN_W(w) "+v"((w).bh[0]), "+v"((w).bh[1]), "+v"((w).bl[0]), "+v"((w).bl[1])
#define W2_PIN_A(f) "+v"((f).h[0]), "+v"((f).h[1]), "+v"((f).l[0]), "+v"((f).l[1])

#ifdef PH_DEV
// development build only (tools/wop_trace.py): shader-clock stamps of waves 0 and 3 of 64 workgroups from the middle of the grid -
// per step: start, the middle of each of the 7 offsets, before / after the step's barrier; then reduction and epilogue.  A stamp is a
// scalar memory read (it shares lgkmcnt with the counted LDS waits): only placed where the kernel waits for lgkmcnt(0) anyway.
constexpr int W2_TRACE_N = 64;
__device__ unsigned long long g_w2_trace[64 * 2 * W2_TRACE_N];
static int g_w2_trace_on = 0;
extern "C" void ph_wop2_trace_enable(int on) { g_w2_trace_on = on; }
extern "C" int ph_wop2_trace_read(unsigned long long *host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w2_trace), sizeof(g_w2_trace)) == hipSuccess ? 0 : 2;
}
#define W2_STAMP(i)                                                                                                        \
  do {                                                                                                                     \
    const int i_ = (i);                                                                                                    \
    if (TRACE && trace_wg >= 0 && lane == 0 && (wave == 0 || wave == 3) && i_ < W2_TRACE_N)                                \
      reinterpret_cast<unsigned long long *>(lds + OFF_TAB + NTAB * W2_TAB)[(wave == 3 ? W2_TRACE_N : 0) + i_] =           \
          __builtin_readcyclecounter();                                                                                    \
  } while (0)
#else
#define W2_STAMP(i) do { } while (0)
#endif

template <bool EMIT, int T, bool TRACE = false>
__global__ void __launch_bounds__(256, 2) k_conv_wop2(ConvArgsH a) {
  constexpr int NTAB = T > 1 ? 2 : 1;
  constexpr int OFF_TAB = 2 * W2_WIN;
  __shared__ __attribute__((aligned(128))) char lds[OFF_TAB + NTAB * W2_TAB + (TRACE ? 2 * 64 * 8 : 0)];

  const int nwg = gridDim.x;
  const int cpx = nwg >> 3;
  const int bid = blockIdx.x;
  const int tile0 = ((bid & 7) * cpx + (bid >> 3)) * T;
  if (tile0 >= a.n_row_tiles) return;
  {
    const int which = a.win_which;       // ph_win_pred
    const int32_t st_w = a.win_stats[which & 1];
    const bool windows = (which & 0x100) ? true : ((which & 0x200) ? false : (int64_t)st_w * 4 <= (int64_t)a.n_row_tiles * 5);
    if (!windows) return;                // the gather kernel serves this map
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int trace_wg = (TRACE && (int)blockIdx.x >= (int)gridDim.x / 2 && (int)blockIdx.x < (int)gridDim.x / 2 + 64) ? (int)blockIdx.x - (int)gridDim.x / 2 : -1;
  (void)trace_wg;
  int stamp = 0;
  (void)stamp;
  W2_STAMP(stamp++);
  const int h = lane >> 5;
  const int l31 = lane & 31;
  const int cout = a.cout;
  const uint32_t rsb = 4u * (uint32_t)a.cpad;
  const int nch = a.cpad >> 4;
#ifdef PH_DEV
  // development build: the ablation mask in a register the compiler cannot re-load from the kernel arguments inside the counted
  // waits (a scalar load shares lgkmcnt with them)
  int abl_bits = a.ablate;
  asm volatile("" : "+s"(abl_bits));
#define W2_ABL(bits) (abl_bits & (bits))
#else
#define W2_ABL(bits) 0
#endif
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;

  // ---- DMA geometry: thread = (window row l_r of a 64-row pass, 16-byte granule l_j); the granule it FETCHES is l_j ^ ((row >> 2) & 3)
  // (fragment reads of consecutive window rows then spread over all banks); granule g of a chunk: 0 / 1 = hi k 0..7 / 8..15, 2 / 3 = lo
  const int l_j = tid & 3;
  const int l_r = tid >> 2;
  const int gf = l_j ^ ((l_r >> 2) & 3);
  const uint32_t gofs = (uint32_t)((gf >> 1) * 64 + (gf & 1) * 16);
  const uint64_t in_base = (uint64_t)reinterpret_cast<uintptr_t>(a.in_split) + gofs;
  const uint64_t zero_src = (uint64_t)reinterpret_cast<uintptr_t>(a.zero) + (uint32_t)(l_j << 4);
  auto chunk_off = [](int c) -> uint32_t { return (uint32_t)((c >> 1) * 128 + (c & 1) * 32); };

  // one table buffer <- slot map + head of the row list of `tile` (3 DMA instructions per wave at most)
  auto tab_dma = [&](int tile, int tb) {
    const char *sl = reinterpret_cast<const char *>(a.win_slots + (int64_t)tile * W2_CAP);
    const char *rw = reinterpret_cast<const char *>(a.win_rows + (int64_t)tile * W2_CAP);
#pragma unroll
    for (int q = 0; q < (W2_PIECES + 255) / 256; ++q) {
      const int i = q * 256 + tid;
      if (i < W2_PIECES) {
        const char *src = i < W2_TAB_SLOT / 16 ? sl + i * 16 : rw + (i - W2_TAB_SLOT / 16) * 16;
        char *dst = lds + OFF_TAB + tb * W2_TAB + (q * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
    }
  };
  // pass p (64 rows) of a window: row list in LDS at `rows_lds` (byte address), `wp` rows valid, channels at byte `coff` of the operand row
  auto win_dma_ix = [&](int ix, int p, int wp, uint32_t coff, int buf) {
    if (W2_ABL(0x8)) return;       // ... without the window DMA
    const bool ok = (p * 64 + l_r) < wp;
    uint64_t v = in_base + (uint64_t)(uint32_t)(ok ? ix : 0) * rsb + coff;
    asm volatile("" : "+v"(v));
    const uint64_t src = ok ? v : zero_src;
    char *dst = lds + buf * W2_WIN + (p * 64 + wave * 16) * W2_ROWB;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  };

  // weight fragments: lane (column l31 of column block j, k-slots 8 h .. 8 h + 7 of the chunk) = 16 bytes of row (k * cout + column)
  // with the fragment-order copy (a.w_frag): lane's 16 bytes of fragment (k, c, j, hi / lo) at ((k nch + c) 4 + 2 j + part) 1024 + 16 lane -
  // every load instruction reads one contiguous kilobyte
  const bool frag = a.w_frag != nullptr;
  const char *wrow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int n = j * 32 + l31;
    n = n < cout ? n : cout - 1;
    wrow[j] = frag ? reinterpret_cast<const char *>(a.w_frag) + j * 2048 + lane * 16
                   : reinterpret_cast<const char *>(a.w_split) + (uint64_t)n * rsb + h * 16;
  }
  const uint64_t wslab = (uint64_t)cout * rsb;
  const uint32_t lo_delta = frag ? 1024u : 64u;
  auto load_w = [&](int k, int c, W2W &f) {
    if (W2_ABL(0x4)) return;       // ... without the weight loads
    const int kc = k < W2_KV ? k : W2_KV - 1;
    const uint64_t off = frag ? (uint64_t)(uint32_t)((kc * nch + c) << 12) : (uint64_t)kc * wslab + chunk_off(c);
    f.bh[0] = w2_gld(wrow[0] + off);
    f.bl[0] = w2_gld(wrow[0] + off + lo_delta);
    f.bh[1] = w2_gld(wrow[1] + off);
    f.bl[1] = w2_gld(wrow[1] + off + lo_delta);
  };

  f32x16 acc[4][2];
  // fragment reads of row blocks i0, i0 + 1: wofs = window row * 64 + ((h ^ swizzle) << 4), lo granule = hi granule ^ 2
  auto issue_a = [&](const uint32_t (&wofs)[4], uint32_t bufbase, int i0, W2A &f) {
    if (W2_ABL(0x2)) return;       // development build: timing without the fragment reads
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t ad = bufbase + wofs[i0 + u];
      f.h[u] = w2_lds(ad);
      f.l[u] = w2_lds(ad ^ 32u);
    }
  };
  auto mfma12 = [&](const W2W &w, const W2A &f, int i0) {
    if (W2_ABL(0x1)) return;       // ... without the products
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i0 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.bh[j], f.l[u], acc[i0 + u][j], 0, 0, 0);
        acc[i0 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.bl[j], f.h[u], acc[i0 + u][j], 0, 0, 0);
        acc[i0 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.bh[j], f.h[u], acc[i0 + u][j], 0, 0, 0);
      }
  };
  // slot -> byte offset of the window row inside a buffer (+ this lane's hi granule)
  auto row_ofs = [&](uint32_t slot, int base, int wp, bool kvalid) -> uint32_t {
    const uint32_t local = slot - (uint32_t)base;
    const uint32_t r = (kvalid && local < (uint32_t)wp) ? local : (uint32_t)W2_MAX;
    return r * (uint32_t)W2_ROWB + ((((r >> 2) & 3u) ^ (uint32_t)h) << 4);
  };

  // ---- first tile: tables, then its first window (the only exposed round trips of the workgroup) -----------------------------
  tab_dma(tile0, 0);
  int cnt = __builtin_amdgcn_readfirstlane(a.win_cnt[tile0]);
  if (tid < 4) reinterpret_cast<uint4 *>(lds + W2_MAX * W2_ROWB)[tid] = make_uint4(0, 0, 0, 0);                 // zero rows of both buffers
  else if (tid < 8) reinterpret_cast<uint4 *>(lds + W2_WIN + W2_MAX * W2_ROWB)[tid - 4] = make_uint4(0, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const int wp = cnt < W2_MAX ? cnt : W2_MAX;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
      const int ix = *reinterpret_cast<const int *>(lds + OFF_TAB + W2_TAB_SLOT + (p * 64 + l_r) * 4);
      win_dma_ix(ix, p, wp, chunk_off(0), 0);
    }
  }
  W2W w0, w1;
  load_w(wave, 0, w0);
  int par = 0;                                         // window buffer of the step about to run
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : W2_PIN_W(w0)::"memory");
  __builtin_amdgcn_s_barrier();

#pragma unroll
  for (int ti = 0; ti < T; ++ti) {
    const int tile = tile0 + ti;
    if (tile >= a.n_row_tiles) break;                  // uniform
    const int tb = ti & (NTAB - 1);
    const bool has_next = (ti + 1 < T) && (tile + 1 < a.n_row_tiles);
    int cnt_next = 0;
    if (has_next) {
      tab_dma(tile + 1, (ti + 1) & (NTAB - 1));        // lands during this tile's first chunk (the chunk boundary certifies it)
      // the load completes HERE (a scalar load shares lgkmcnt with the counted LDS waits below)
      cnt_next = __builtin_amdgcn_readfirstlane(a.win_cnt[tile + 1]);
    }
    const int64_t m0 = (int64_t)tile * W2_BM;
    const int npass = cnt > 0 ? (cnt + W2_MAX - 1) / W2_MAX : 1;
    const int nsteps = nch * npass;                    // nch = 16-channel chunks of the input rows
    const uint32_t tab_slots = lds0 + (uint32_t)(OFF_TAB + tb * W2_TAB) + (uint32_t)(l31 * 2);        // + (k * 128 + 32 i) * 2
    const uint32_t tab_rows = lds0 + (uint32_t)(OFF_TAB + tb * W2_TAB + W2_TAB_SLOT) + (uint32_t)(l_r * 4);
    const uint32_t nxt_rows = lds0 + (uint32_t)(OFF_TAB + ((ti + 1) & (NTAB - 1)) * W2_TAB + W2_TAB_SLOT) + (uint32_t)(l_r * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int s = 0, pass = 0, c = 0; s < nsteps; ++s, c = (c + 1 == nch ? 0 : c + 1), pass += (c == 0 ? 1 : 0)) {
      const int base = pass * W2_MAX;
      const int wp = cnt - base < W2_MAX ? cnt - base : W2_MAX;
      // the window this step's DMAs fetch: the next chunk of this pass, the first chunk of the next tile, or nothing (zero line)
      const bool same = c + 1 < nch;
      const bool nxt_tile = !same && s + 1 == nsteps && has_next;
      const uint32_t n_rows = same ? tab_rows : nxt_rows;
      const int n_wp = same ? wp : (nxt_tile ? (cnt_next < W2_MAX ? cnt_next : W2_MAX) : 0);
      const uint32_t n_coff = same ? chunk_off(c + 1) : chunk_off(0);
      const int cn = same ? c + 1 : 0;                 // chunk of the weights prefetched at the last offset
      const uint32_t bufbase = lds0 + (uint32_t)(par * W2_WIN);
      const int nbuf = par ^ 1;

      // the window of this step has landed (every wave waited for its DMAs before the barrier that ended the previous step)
      uint32_t sl[4], wofs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sl[i] = w2_lds16(tab_slots + (uint32_t)((wave * W2_BM + i * 32) * 2));
      int ixn = w2_lds32(n_rows);                      // row of DMA pass 0
      W2_STAMP(stamp++);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sl[0]), "+v"(sl[1]), "+v"(sl[2]), "+v"(sl[3]), "+v"(ixn)::"memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) wofs[i] = row_ofs(sl[i], base, wp, true);
      W2A fa, fb;
      __builtin_amdgcn_sched_barrier(0);
      issue_a(wofs, bufbase, 0, fa);
      issue_a(wofs, bufbase, 2, fb);
      __builtin_amdgcn_sched_barrier(0);

#define W2_OFFSET(t, WC, WN)                                                                                                  \
  do {                                                                                                                        \
    const int kn = wave + 4 * ((t) + 1);                                                                                      \
    /* head: this offset's first fragments and the row of its DMA pass have landed (the second group of reads may fly); */  \
    /* then the slots of the next offset, the window DMA, the next offset's (chunk's) weights */                            \
    asm volatile("s_waitcnt lgkmcnt(4)" : W2_PIN_A(fa), "+v"(ixn)::"memory");                                                 \
    if (W2_ABL(0x2)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ixn)::"memory");   /* ablated reads: the counts above do not hold */ \
    if ((t) < 6) {                                                                                                            \
      const int ks = kn < W2_KV ? kn : W2_KV - 1;                                                                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) sl[i] = w2_lds16(tab_slots + (uint32_t)((ks * W2_BM + i * 32) * 2));     \
    }                                                                                                                         \
    win_dma_ix(ixn, (t), n_wp, n_coff, nbuf);                                                                                 \
    if ((t) < 6) load_w(kn, c, WN);                                                                                           \
    else load_w(wave, cn, WN);                                                                                                \
    /* this offset's weights (requested one offset ahead; at t = 0 they were waited for before the step's barrier) */      \
    if ((t) > 0 && !W2_ABL(0x20)) asm volatile("s_waitcnt vmcnt(5)" : W2_PIN_W(WC)::"memory");   /* 0x20: weights not waited for */ \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    mfma12(WC, fa, 0);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    W2_STAMP(stamp++);                                                                                                        \
    if ((t) < 6) {                                                                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sl[0]), "+v"(sl[1]), "+v"(sl[2]), "+v"(sl[3]), W2_PIN_A(fb)::"memory");      \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) wofs[i] = row_ofs(sl[i], base, wp, kn < W2_KV);                           \
      __builtin_amdgcn_sched_barrier(0);                                                                                      \
      issue_a(wofs, bufbase, 0, fa);                                                                                          \
      ixn = w2_lds32(n_rows + (uint32_t)(((t) + 1) * 64 * 4));                                                                \
    } else {                                                                                                                  \
      asm volatile("s_waitcnt lgkmcnt(0)" : W2_PIN_A(fb)::"memory");                                                          \
    }                                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    mfma12(WC, fb, 2);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    if ((t) < 6) issue_a(wofs, bufbase, 2, fb);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
  } while (0)

      W2_OFFSET(0, w0, w1);
      W2_OFFSET(1, w1, w0);
      W2_OFFSET(2, w0, w1);
      W2_OFFSET(3, w1, w0);
      W2_OFFSET(4, w0, w1);
      W2_OFFSET(5, w1, w0);
      W2_OFFSET(6, w0, w1);
#undef W2_OFFSET
      // this wave's DMAs of the next window and the weights prefetched at the last offset have landed, its reads of this window
      // are done; the next step starts with w0
      W2_STAMP(stamp++);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : W2_PIN_W(w1)::"memory");
      W2_STAMP(stamp++);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      w0 = w1;
      __builtin_amdgcn_s_barrier();
      par ^= 1;
      if (!same && s + 1 < nsteps) {
        // a further pass of a window beyond the LDS capacity (rare: maps the predicate lets through have <= 1.25 passes per
        // tile on average): its row list segment and first window, synchronously
        const int nb = (pass + 1) * W2_MAX;
        const int nwp = cnt - nb < W2_MAX ? cnt - nb : W2_MAX;
        const int32_t *wrp = a.win_rows + (int64_t)tile * W2_CAP + nb;
        int *rows = reinterpret_cast<int *>(lds + OFF_TAB + tb * W2_TAB + W2_TAB_SLOT);
        for (int i = tid; i < W2_MAX; i += 256) rows[i] = i < nwp ? wrp[i] : 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int p = 0; p < 7; ++p) {
          const int ix = rows[p * 64 + l_r];
          win_dma_ix(ix, p, nwp, chunk_off(0), par);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }

    // ---- sum of the four partial accumulators through the window buffer that is NOT receiving the next tile's window: four
    // rounds, one row block each: the three other waves write their part, the owner adds: own part, then the others ascending
    if (W2_ABL(0x10)) continue;    // ... without reduction and epilogue
    W2_STAMP(stamp++);
    f32x16 fin[1][2];
    if constexpr (T == 1) {
      // one tile per workgroup: both window buffers are free - two rounds of two row blocks (48 KB): in round b the waves write
      // their partial sums of blocks 2 b and 2 b + 1 (the owner keeps its own), waves 2 b and 2 b + 1 add: own part, then the others'
      float *red = reinterpret_cast<float *>(lds);       // [block in round][writer: the 3 other waves][j][r][lane]
      static_assert(2 * W2_WIN >= 2 * 3 * 2 * 16 * 64 * 4, "two row blocks of partial sums fit the window buffers");
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = 2 * b + u;
          if (wave != i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) red[(((u * 3 + (wave < i ? wave : wave - 1)) * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
          }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = 2 * b + u;
          if (wave == i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              fin[0][j] = acc[i][j];
#pragma unroll
              for (int sw = 0; sw < 3; ++sw)
#pragma unroll
                for (int r = 0; r < 16; ++r) fin[0][j][r] += red[(((u * 3 + sw) * 2 + j) * 16 + r) * 64 + lane];
            }
          }
        }
        if (b == 0) __syncthreads();
      }
    } else {
    // four rounds of one row block through the window buffer that is NOT receiving the next tile's window
    float *red = reinterpret_cast<float *>(lds + (par ^ 1) * W2_WIN);      // [writer: the 3 other waves][j][r][lane]: 24 KB
    static_assert(W2_WIN >= 3 * 2 * 16 * 64 * 4, "one round of the reduction fits a window buffer");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (wave != i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((wave < i ? wave : wave - 1) * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (wave == i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          fin[0][j] = acc[i][j];
#pragma unroll
          for (int sw = 0; sw < 3; ++sw)
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[0][j][r] += red[((sw * 2 + j) * 16 + r) * 64 + lane];
        }
      }
      if (i < 3) __syncthreads();
    }
    }
    W2_STAMP(stamp++);
    // the epilogue with every load ahead of its first store (conv_h2_common.h: the plain form serialises its four column groups
    // on the vmcnt order of loads behind stores: 10.9 k of a workgroup's 85 k clocks, tools/wop2_trace.py): per-channel vectors
    // and each lane's residual values staged in the window buffers, which nothing reads any more
    if constexpr (T == 1) h2_store_tile_staged<1, 2, EMIT, 256, 64, 2 * W2_WIN>(a, fin, m0, 0, wave, 0, h, l31, tid, lds);
    else h2_store_tile<1, 2, EMIT>(a, fin, m0, 0, wave, 0, h, l31);
    W2_STAMP(stamp++);
#ifdef PH_DEV
    if (TRACE && trace_wg >= 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      W2_STAMP(stamp++);                 // the epilogue's stores have completed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < 2 * W2_TRACE_N)
        g_w2_trace[trace_wg * 2 * W2_TRACE_N + tid] = reinterpret_cast<unsigned long long *>(lds + OFF_TAB + NTAB * W2_TAB)[tid];
    }
#endif
    cnt = cnt_next;
    if (has_next) {
      // every owner is done reading the reduction buffer before the next tile's DMAs write it (its first window landed before the
      // last step's barrier); the epilogue's stores stay in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

// Launches the window side of a pair for 64-wide outputs (conv_win.hip: ph_conv_win_launch).  tiles_per_wg: 1 or 2.
int ph_conv_wop2_launch(const ConvArgsH &args, int tiles_per_wg, hipStream_t st) {
  const int nwg = (args.n_row_tiles + tiles_per_wg - 1) / tiles_per_wg;
  const int grid = ((nwg + 7) / 8) * 8;
  const bool emit = args.out_split != nullptr;
#ifdef PH_DEV
  if (g_w2_trace_on && !emit && tiles_per_wg == 1) {      // never on a launch that has to write the next layer's operand
    hipLaunchKernelGGL((k_conv_wop2<false, 1, true>), dim3(grid), dim3(256), 0, st, args);
    PH_LAUNCH_CHECK();
    return 0;
  }
#endif
  if (tiles_per_wg == 2) {
    if (emit) hipLaunchKernelGGL((k_conv_wop2<true, 2>), dim3(grid), dim3(256), 0, st, args);
    else hipLaunchKernelGGL((k_conv_wop2<false, 2>), dim3(grid), dim3(256), 0, st, args);
  } else {
    if (emit) hipLaunchKernelGGL((k_conv_wop2<true, 1>), dim3(grid), dim3(256), 0, st, args);
    else hipLaunchKernelGGL((k_conv_wop2<false, 1>), dim3(grid), dim3(256), 0, st, args);
  }
  PH_LAUNCH_CHECK();
  return 0;
}
