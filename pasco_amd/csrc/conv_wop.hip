// Sparse 3^3 convolution of the 64-channel levels on LDS-resident input windows, round 6 (`k_conv_wop2`, kernel id 5 like the
// kernel it replaces: the window side of a window / gather pair).
//
// What held k_conv_wop (conv_win.hip, round 4) at 0.25 of the f16 matrix peak, measured this round with shader-clock stamps
// on workgroups from the middle of the grid (tools/wop2_trace.py) and with parts of the kernel switched off
// (tools/wop_ablate.py): a workgroup lives ~85 k clocks for 21.5 k clocks of matrix work per wave; 28 % of that life is outside
// the offset loops (tables + first window 7 k, reduction 5.5 k, an epilogue of 11 k that serialised its four column groups on
// the in-order vmcnt of loads behind stores), and inside the loops a wave ALONE on its SIMD (its partner workgroup being in
// one of those phases) needs ~1 800 clocks per offset for 768 clocks of products: its own ~60 address instructions, 13 LDS /
// memory instructions and their waits sit BETWEEN its groups of 12 MFMAs instead of under them.
// Same arithmetic here (offset-parallel waves: wave w owns offsets w, w + 4, ..., every weight fragment global -> registers,
// 128 rows x 64 columns of accumulators per wave, partial sums added through LDS in a fixed order), but:
//   * 16-CHANNEL window chunks in TWO LDS buffers: the window of chunk c + 1 (64 bytes per row: 16 hi | 16 lo values) is
//     DMA'd while chunk c multiplies - one DMA instruction per wave and offset, every one of them issued unconditionally
//     (rows beyond the window read the zero line) so that the counted vmcnt waits are constants;
//   * the slot map is turned into LDS BYTE OFFSETS once per tile and pass (in place, under the first window's DMA): slot ->
//     row * 64 + swizzle, "no neighbour / other pass" -> the zero row: 3 address instructions per fragment pair instead of 15;
//   * the offsets of a chunk are straight-line code (7 x 24 MFMAs) and every other instruction of an offset is placed in a gap
//     BETWEEN two MFMAs: the next offset's addresses, the window DMA, the next offset's weights (one contiguous kilobyte per
//     load instruction from the fragment-order copy of the kernel, ph_conv_desc.w_frag) under the first 12, the slot reads of
//     the offset after next under the second 12; fragment reads between the groups, each a group of products ahead;
//   * two rounds of reduction through both (by then free) window buffers, and the epilogue's loads (per-channel vectors to LDS,
//     residual values to registers) issued ahead of the reduction: no load of the epilogue waits behind one of its stores.
// PERSISTENT workgroups (tiles drawn from per-XCD queues by atomics, the next tile's ticket / tables / first window fetched under
// the current tile) were built this round and give bit-identical results, but the compiler spills 250 registers around the
// tile loop and reloads two of them inside the offsets (each reload drains vmcnt): 31 % SLOWER (profiles/
// r6m_wop2_persistent_rejected.txt) - removed; the ~8 k clocks at the head of every tile (10 % of its life) stay exposed.
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
constexpr int W2_ZOFS = W2_MAX * W2_ROWB;          // byte offset of the zero row (its swizzle is 0)
constexpr int W2_TAB_SLOT = W2_CAP * 2;            // u16 slot map
constexpr int W2_TAB = W2_TAB_SLOT + W2_MAX * 4;   // + the first W2_MAX entries of the row list
constexpr int W2_PIECES = W2_TAB / 16;             // 16-byte pieces of the table buffer (544)
constexpr int OFF_TAB = 2 * W2_WIN;
static_assert(W2_TAB % 16 == 0 && W2_WIN % 64 == 0 && W2_TAB_SLOT % 16 == 0, "alignment of the LDS regions");
static_assert(W2_ZOFS + 48 < 65536, "row offsets fit the u16 entries of the slot map");

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
// ... with the constant part of the address in the instruction (a register per constant address is hoisted out of every loop and
// spilled: the slot of (offset k, row block i) sits at k * 256 + i * 64 behind the wave's base)
#define W2_LDS16_IMM(dst, addr, IMM) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory")
#define W2_LDS32_IMM(dst, addr, IMM) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory")
struct W2W {                 // weight fragments of one (offset, chunk): [column block][hi, lo]
  f16x8 bh[2], bl[2];
};
struct W2A {                 // activation fragments of two row blocks
  f16x8 h[2], l[2];
};
#define W2_PIN_W(w) "+v"((w).bh[0]), "+v"((w).bh[1]), "+v"((w).bl[0]), "+v"((w).bl[1])
#define W2_PIN_A(f) "+v"((f).h[0]), "+v"((f).h[1]), "+v"((f).l[0]), "+v"((f).l[1])
#define W2_SB() __builtin_amdgcn_sched_barrier(0)

#ifdef PH_DEV
// development build only (tools/wop2_trace.py): shader-clock stamps of waves 0 and 3 of 64 workgroups from the middle of the grid -
// per step: start, the middle of each of the 7 offsets, before / after the step's barrier; then reduction and epilogue.  A stamp is a
// scalar memory read (it shares lgkmcnt with the counted LDS waits): each is followed by a wait for lgkmcnt(0).
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
    if (TRACE && trace_wg >= 0) {                                                                                          \
      if (lane == 0 && (wave == 0 || wave == 3) && i_ < W2_TRACE_N)                                                        \
        reinterpret_cast<unsigned long long *>(lds + OFF_TAB + W2_TAB)[(wave == 3 ? W2_TRACE_N : 0) + i_] =                \
            __builtin_readcyclecounter();                                                                                  \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
    }                                                                                                                      \
  } while (0)
#else
#define W2_STAMP(i) do { } while (0)
#endif

template <bool EMIT, bool TRACE = false>
__global__ void __launch_bounds__(256, 2) k_conv_wop2(ConvArgsH a) {
  __shared__ __attribute__((aligned(128))) char lds[OFF_TAB + W2_TAB + (TRACE ? 2 * 64 * 8 : 0)];

  const int nwg = gridDim.x;
  const int cpx = nwg >> 3;
  const int bid = blockIdx.x;
  const int tile = (bid & 7) * cpx + (bid >> 3);
  if (tile >= a.n_row_tiles) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wave >= 0 && wave < 4);
  const int trace_wg = (TRACE && (int)blockIdx.x >= (int)gridDim.x / 2 && (int)blockIdx.x < (int)gridDim.x / 2 + 64) ? (int)blockIdx.x - (int)gridDim.x / 2 : -1;
  (void)trace_wg;
  int stamp = 0;
  (void)stamp;
  W2_STAMP(stamp++);
  const int h = lane >> 5;
  const int l31 = lane & 31;
  const uint32_t hx = (uint32_t)(h << 4);
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
  auto chunk_off = [](int c) -> uint32_t { return (uint32_t)((c >> 1) * 128 + (c & 1) * 32); };

  // the table buffer <- slot map + head of the row list of the tile (3 DMA instructions per wave at most)
  auto tab_dma = [&]() {
    const char *sl = reinterpret_cast<const char *>(a.win_slots + (int64_t)tile * W2_CAP);
    const char *rw = reinterpret_cast<const char *>(a.win_rows + (int64_t)tile * W2_CAP);
#pragma unroll
    for (int q = 0; q < (W2_PIECES + 255) / 256; ++q) {
      const int i = q * 256 + tid;
      if (i < W2_PIECES) {
        const char *src = i < W2_TAB_SLOT / 16 ? sl + i * 16 : rw + (i - W2_TAB_SLOT / 16) * 16;
        char *dst = lds + OFF_TAB + (q * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
    }
  };
  // source address of this thread's 16 bytes of DMA pass p (64 rows) of a window: row `ix` of the operand, `wp` rows valid
  // (a window row beyond `wp` is never named by a slot - absent neighbours go to the zero row -, so what lands in it does not matter:
  // such a lane fetches row 0 of the operand instead of selecting a second base pointer)
  auto dma_src = [&](int ix, int p, int wp, uint32_t coff) -> uint64_t {
    const bool ok = (p * 64 + l_r) < wp;
    uint64_t v = in_base + (uint64_t)(uint32_t)(ok ? ix : 0) * rsb + coff;
    asm volatile("" : "+v"(v));
    return v;
  };
  auto dma_go = [&](uint64_t src, int p, int buf) {
    char *dst = lds + buf * W2_WIN + (p * 64 + wave * 16) * W2_ROWB;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  };
  // slot map -> byte offsets inside a window buffer, in place: entry (k, r) = window row * 64 + (swizzle << 4) for a row of this pass,
  // the zero row for "no neighbour" / a row of another pass.  Two entries per 32-bit word.
  auto tab_transform = [&](int base, int wp) {
    uint32_t *w = reinterpret_cast<uint32_t *>(lds + OFF_TAB);
    for (int i = tid; i < W2_CAP / 2; i += 256) {
      const uint32_t v = w[i];
      uint32_t o = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const uint32_t local = ((v >> (16 * e)) & 0xFFFFu) - (uint32_t)base;
        const uint32_t ofs = local < (uint32_t)wp ? local * (uint32_t)W2_ROWB + (((local >> 2) & 3u) << 4) : (uint32_t)W2_ZOFS;
        o |= ofs << (16 * e);
      }
      w[i] = o;
    }
  };

  // weight fragments in fragment order: lane's 16 bytes of fragment (k, c, j, hi / lo) at ((k nch + c) 4 + 2 j + part) 1024 + 16 lane
  const char *wbase = reinterpret_cast<const char *>(a.w_frag) + lane * 16;
  auto w_addr = [&](int k, int c) -> const char * {
    const int kc = k < W2_KV ? k : W2_KV - 1;
    return wbase + (uint64_t)(uint32_t)((kc * nch + c) << 12);
  };
#define W2_GLD(dst, ptr, OFFS) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFFS : "=v"(dst) : "v"(ptr) : "memory")

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // the n-th of the 12 products of row blocks i0, i0 + 1 with one weight set: per accumulator the smallest terms first (hi x lo,
  // lo x hi, hi x hi), its three products four issues apart
  auto mf = [&](const W2W &w, const W2A &f, int i0, int n) {
    if (W2_ABL(0x1)) return;
    const int p = n >> 2, u = (n >> 1) & 1, j = n & 1;
    if (p == 0) acc[i0 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.bh[j], f.l[u], acc[i0 + u][j], 0, 0, 0);
    else if (p == 1) acc[i0 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.bl[j], f.h[u], acc[i0 + u][j], 0, 0, 0);
    else acc[i0 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.bh[j], f.h[u], acc[i0 + u][j], 0, 0, 0);
  };

  // ---- tables, then the first window (the exposed round trips of the workgroup): the table DMA goes out FIRST, the predicate of the
  // pair, the tile's row count and the first weights are fetched under it; the slot map is rewritten under the window's DMA -------
  tab_dma();
  W2W w0, w1;
  {
    const char *wp0 = w_addr(wave, 0);
    W2_GLD(w0.bh[0], wp0, 0);
    W2_GLD(w0.bl[0], wp0, 1024);
    W2_GLD(w0.bh[1], wp0, 2048);
    W2_GLD(w0.bl[1], wp0, 3072);
  }
  const int cnt = __builtin_amdgcn_readfirstlane(a.win_cnt[tile]);
  bool windows;
  {
    const int which = a.win_which;       // ph_win_pred
    const int32_t st_w = a.win_stats[which & 1];
    windows = (which & 0x100) ? true : ((which & 0x200) ? false : (int64_t)st_w * 4 <= (int64_t)a.n_row_tiles * 5);
  }
  if (tid < 4) reinterpret_cast<uint4 *>(lds + W2_ZOFS)[tid] = make_uint4(0, 0, 0, 0);                 // zero rows of both buffers
  else if (tid < 8) reinterpret_cast<uint4 *>(lds + W2_WIN + W2_ZOFS)[tid - 4] = make_uint4(0, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : W2_PIN_W(w0)::"memory");
  if (!windows) return;                  // the gather kernel serves this map (nothing of this workgroup is in flight any more)
  __builtin_amdgcn_s_barrier();
  const int npass = cnt > 0 ? (cnt + W2_MAX - 1) / W2_MAX : 1;
  const int nsteps = nch * npass;
  const uint32_t tab_slots = lds0 + (uint32_t)OFF_TAB + (uint32_t)(l31 * 2);                          // + (k * 128 + 32 i) * 2
  const uint32_t tab_rows = lds0 + (uint32_t)(OFF_TAB + W2_TAB_SLOT) + (uint32_t)(l_r * 4);           // + 64 p * 4
  const uint32_t slot_w = tab_slots + (uint32_t)(wave * W2_BM * 2);      // this wave's first offset; its offset wave + 4 t: + 4 t * 256
  {
    const int wp = cnt < W2_MAX ? cnt : W2_MAX;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
      const int ix = *reinterpret_cast<const int *>(lds + OFF_TAB + W2_TAB_SLOT + (p * 64 + l_r) * 4);
      dma_go(dma_src(ix, p, wp, chunk_off(0)), p, 0);
    }
    tab_transform(0, wp);
  }
  int par = 0;                                         // window buffer of the step about to run
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int64_t m0 = (int64_t)tile * W2_BM;
  // row offsets of this wave's first two offsets: the same for every chunk of a pass - read once, not at the head of every step
  uint32_t s0[4], s1[4];
  auto first_slots = [&]() {
    W2_LDS16_IMM(s0[0], slot_w, 0); W2_LDS16_IMM(s0[1], slot_w, 64); W2_LDS16_IMM(s0[2], slot_w, 128); W2_LDS16_IMM(s0[3], slot_w, 192);
    W2_LDS16_IMM(s1[0], slot_w, 1024); W2_LDS16_IMM(s1[1], slot_w, 1088); W2_LDS16_IMM(s1[2], slot_w, 1152); W2_LDS16_IMM(s1[3], slot_w, 1216);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0[0]), "+v"(s0[1]), "+v"(s0[2]), "+v"(s0[3]), "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3])::"memory");
  };
  first_slots();

  for (int s = 0, pass = 0, c = 0; s < nsteps; ++s, c = (c + 1 == nch ? 0 : c + 1), pass += (c == 0 ? 1 : 0)) {
    const int base = pass * W2_MAX;
    const int wp = cnt - base < W2_MAX ? cnt - base : W2_MAX;
    // the window this step's DMAs fetch: the next chunk of this pass, or nothing (zero line: the last chunk of a pass)
    const bool same = c + 1 < nch;
    const int n_wp = same ? wp : 0;
    const uint32_t n_coff = same ? chunk_off(c + 1) : chunk_off(0);
    const int cn = same ? c + 1 : 0;                 // chunk of the weights prefetched at the last offset
    const uint32_t bufbase = lds0 + (uint32_t)(par * W2_WIN);
    const int nbuf = par ^ 1;

    // ---- step prologue: the first fragments (addresses from the pass's first slots) - in flight, in this order: fa(0) + ixn(0), fb(0)
    uint32_t sl[4], ad[4];
    int ixn;
    W2A fa, fb;
    W2_STAMP(stamp++);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ad[i] = bufbase + (s0[i] ^ hx);
      sl[i] = s1[i];
    }
    W2_SB();
    if (!W2_ABL(0x2)) {
      fa.h[0] = w2_lds(ad[0]); fa.l[0] = w2_lds(ad[0] ^ 32u); fa.h[1] = w2_lds(ad[1]); fa.l[1] = w2_lds(ad[1] ^ 32u);
    }
    ixn = w2_lds32(tab_rows);
    if (!W2_ABL(0x2)) {
      fb.h[0] = w2_lds(ad[2]); fb.l[0] = w2_lds(ad[2] ^ 32u); fb.h[1] = w2_lds(ad[3]); fb.l[1] = w2_lds(ad[3] ^ 32u);
    }
    W2_SB();

    // One offset.  G1 = the 12 products of row blocks 0, 1 (fa), G2 = those of row blocks 2, 3 (fb); everything else in their gaps:
    //   G1: [wait fa, ixn (, this offset's weights)] mf, [wait sl] mf, next offset's weights (address + 4 loads) between the next
    //       four, the window DMA of this offset, the next offset's four fragment addresses, one per gap;
    //   between: the reads of fa(t + 1) and of the row of the next DMA pass;
    //   G2: [wait fb] the slot reads of offset t + 2 in the first four gaps;
    //   after: the reads of fb(t + 1).
    // LDS queue at the head of an offset (oldest first): fa(t) 4, ixn(t) 1, sl(t + 1) 4, fb(t) 4 (t = 0: the slots are in registers).
#define W2_OFFSET(t, WC, WN)                                                                                                  \
  do {                                                                                                                        \
    const int kn = wave + 4 * ((t) + 1);                                                                                      \
    const char *wpn = (t) < 6 ? w_addr(kn, c) : w_addr(wave, cn);                                                             \
    if ((t) > 0 && (t) < 6) asm volatile("s_waitcnt lgkmcnt(8)" : W2_PIN_A(fa), "+v"(ixn)::"memory");                         \
    else asm volatile("s_waitcnt lgkmcnt(4)" : W2_PIN_A(fa), "+v"(ixn)::"memory");                                            \
    if (W2_ABL(0x2)) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ixn)::"memory");   /* ablated reads: the counts do not hold */   \
    if ((t) > 0) asm volatile("s_waitcnt vmcnt(1)" : W2_PIN_W(WC)::"memory");      /* this offset's weights (the DMA behind them may fly) */ \
    W2_SB(); mf(WC, fa, 0, 0); W2_SB();                                                                                       \
    if ((t) > 0 && (t) < 6) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(sl[0]), "+v"(sl[1]), "+v"(sl[2]), "+v"(sl[3])::"memory"); \
    W2_SB(); mf(WC, fa, 0, 1); W2_SB();                                                                                       \
    if (!W2_ABL(0x4)) W2_GLD(WN.bh[0], wpn, 0);                                                                               \
    W2_SB(); mf(WC, fa, 0, 2); W2_SB();                                                                                       \
    if (!W2_ABL(0x4)) W2_GLD(WN.bl[0], wpn, 1024);                                                                            \
    W2_SB(); mf(WC, fa, 0, 3); W2_SB();                                                                                       \
    if (!W2_ABL(0x4)) W2_GLD(WN.bh[1], wpn, 2048);                                                                            \
    W2_SB(); mf(WC, fa, 0, 4); W2_SB();                                                                                       \
    if (!W2_ABL(0x4)) W2_GLD(WN.bl[1], wpn, 3072);                                                                            \
    W2_SB(); mf(WC, fa, 0, 5); W2_SB();                                                                                       \
    const uint64_t dsrc = dma_src(ixn, (t), n_wp, n_coff);                                                                    \
    W2_SB(); mf(WC, fa, 0, 6); W2_SB();                                                                                       \
    if (!W2_ABL(0x8)) dma_go(dsrc, (t), nbuf);                                                                                \
    W2_SB(); mf(WC, fa, 0, 7); W2_SB();                                                                                       \
    if ((t) < 6) {                                                                                                            \
      const bool kv = kn < W2_KV;                                                                                             \
      ad[0] = bufbase + (kv ? sl[0] ^ hx : (uint32_t)W2_ZOFS + hx);                                                           \
      W2_SB(); mf(WC, fa, 0, 8); W2_SB();                                                                                     \
      ad[1] = bufbase + (kv ? sl[1] ^ hx : (uint32_t)W2_ZOFS + hx);                                                           \
      W2_SB(); mf(WC, fa, 0, 9); W2_SB();                                                                                     \
      ad[2] = bufbase + (kv ? sl[2] ^ hx : (uint32_t)W2_ZOFS + hx);                                                           \
      W2_SB(); mf(WC, fa, 0, 10); W2_SB();                                                                                    \
      ad[3] = bufbase + (kv ? sl[3] ^ hx : (uint32_t)W2_ZOFS + hx);                                                           \
      W2_SB(); mf(WC, fa, 0, 11); W2_SB();                                                                                    \
      if (!W2_ABL(0x2)) {                                                                                                     \
        fa.h[0] = w2_lds(ad[0]); fa.l[0] = w2_lds(ad[0] ^ 32u); fa.h[1] = w2_lds(ad[1]); fa.l[1] = w2_lds(ad[1] ^ 32u);       \
      }                                                                                                                       \
      W2_LDS32_IMM(ixn, tab_rows, ((t) + 1) * 64 * 4);                                                                        \
    } else {                                                                                                                  \
      W2_SB(); mf(WC, fa, 0, 8); W2_SB(); mf(WC, fa, 0, 9); W2_SB(); mf(WC, fa, 0, 10); W2_SB(); mf(WC, fa, 0, 11); W2_SB();   \
    }                                                                                                                         \
    W2_STAMP(stamp++);                                                                                                        \
    if ((t) < 6) asm volatile("s_waitcnt lgkmcnt(5)" : W2_PIN_A(fb)::"memory");                                               \
    else asm volatile("s_waitcnt lgkmcnt(0)" : W2_PIN_A(fb)::"memory");                                                       \
    W2_SB(); mf(WC, fb, 2, 0); W2_SB();                                                                                       \
    /* offset t + 2 of this wave: k = wave + 4 (t + 2); wave 3's k = 27 reads the head of the row list - its entries are not used (kv) */ \
    if ((t) < 5) W2_LDS16_IMM(sl[0], slot_w, (4 * ((t) + 2) * W2_BM + 0) * 2);                                                \
    W2_SB(); mf(WC, fb, 2, 1); W2_SB();                                                                                       \
    if ((t) < 5) W2_LDS16_IMM(sl[1], slot_w, (4 * ((t) + 2) * W2_BM + 32) * 2);                                               \
    W2_SB(); mf(WC, fb, 2, 2); W2_SB();                                                                                       \
    if ((t) < 5) W2_LDS16_IMM(sl[2], slot_w, (4 * ((t) + 2) * W2_BM + 64) * 2);                                               \
    W2_SB(); mf(WC, fb, 2, 3); W2_SB();                                                                                       \
    if ((t) < 5) W2_LDS16_IMM(sl[3], slot_w, (4 * ((t) + 2) * W2_BM + 96) * 2);                                               \
    W2_SB(); mf(WC, fb, 2, 4); W2_SB(); mf(WC, fb, 2, 5); W2_SB(); mf(WC, fb, 2, 6); W2_SB(); mf(WC, fb, 2, 7); W2_SB();      \
    mf(WC, fb, 2, 8); W2_SB(); mf(WC, fb, 2, 9); W2_SB(); mf(WC, fb, 2, 10); W2_SB(); mf(WC, fb, 2, 11); W2_SB();             \
    if ((t) < 6 && !W2_ABL(0x2)) {                                                                                            \
      fb.h[0] = w2_lds(ad[2]); fb.l[0] = w2_lds(ad[2] ^ 32u); fb.h[1] = w2_lds(ad[3]); fb.l[1] = w2_lds(ad[3] ^ 32u);         \
    }                                                                                                                         \
    W2_SB();                                                                                                                  \
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
    w0 = w1;
    __builtin_amdgcn_s_barrier();
    par ^= 1;
    if (!same && s + 1 < nsteps) {
      // a further pass of a window beyond the LDS capacity (rare: maps the predicate lets through have <= 1.25 passes per tile
      // on average): the tables again, the next segment of the row list, the slot map for that pass, its first window - synchronously
      const int nb = (pass + 1) * W2_MAX;
      const int nwp = cnt - nb < W2_MAX ? cnt - nb : W2_MAX;
      tab_dma();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int32_t *wrp = a.win_rows + (int64_t)tile * W2_CAP + nb;
      int *rows = reinterpret_cast<int *>(lds + OFF_TAB + W2_TAB_SLOT);
      for (int i = tid; i < W2_MAX; i += 256) rows[i] = i < nwp ? wrp[i] : 0;
      tab_transform(nb, nwp);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int p = 0; p < 7; ++p) dma_go(dma_src(rows[p * 64 + l_r], p, nwp, chunk_off(0)), p, par);
      first_slots();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  if (W2_ABL(0x10)) return;            // development build: timing without reduction and epilogue

  // ---- the epilogue's loads FIRST: the per-channel vectors into the (free) table buffer, this lane's residual values into
  // registers - they fly under the reduction, and no load of the epilogue waits behind one of its stores (the plain form
  // serialises its four column groups on the in-order vmcnt of loads behind stores: 11 k of a workgroup's 85 k clocks) ----------
  W2_STAMP(stamp++);
  lds_float *epar = (lds_float *)(lds + OFF_TAB);
  h2_stage_params(a, 0, 64, tid, epar);
  TailRegs<8> tail;
  if (a.has_tail) h2_load_tail_regs<1, 2>(a, m0, 0, wave, 0, h, l31, tail);
  // ---- sum of the four partial accumulators through both window buffers (free now): two rounds of two row blocks (48 KB): in
  // round b the waves write their partial sums of blocks 2 b and 2 b + 1 (the owner keeps its own), waves 2 b and 2 b + 1 add:
  // own part first, then the other waves' ascending - a fixed order.  16 bytes per lane and LDS instruction.
  f32x16 fin[1][2];
  {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v *red = reinterpret_cast<f32x4v *>(lds);       // [block in round][writer: the 3 other waves][j][r / 4][lane] x 4 floats
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
            for (int r4 = 0; r4 < 4; ++r4) {
              f32x4v v;
              v[0] = acc[i][j][4 * r4]; v[1] = acc[i][j][4 * r4 + 1]; v[2] = acc[i][j][4 * r4 + 2]; v[3] = acc[i][j][4 * r4 + 3];
              red[(((u * 3 + (wave < i ? wave : wave - 1)) * 2 + j) * 4 + r4) * 64 + lane] = v;
            }
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
              for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4v v = red[(((u * 3 + sw) * 2 + j) * 4 + r4) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) fin[0][j][4 * r4 + q] += v[q];
              }
          }
        }
      }
      if (b == 0) __syncthreads();
    }
  }
  W2_STAMP(stamp++);
  h2_store_tile<1, 2, EMIT, ParLds, TailRegs<8>>(a, fin, m0, 0, wave, 0, h, l31, ParLds{epar, 0, 64}, tail);
  W2_STAMP(stamp++);
#ifdef PH_DEV
  if (TRACE && trace_wg >= 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    W2_STAMP(stamp++);                 // the epilogue's stores have completed
    __syncthreads();
    if (tid < 2 * W2_TRACE_N)
      g_w2_trace[trace_wg * 2 * W2_TRACE_N + tid] = reinterpret_cast<unsigned long long *>(lds + OFF_TAB + W2_TAB)[tid];
  }
#endif
}

// Launches the window side of a pair for 64-wide outputs (conv_win.hip: ph_conv_win_launch); needs args.w_frag.
int ph_conv_wop2_launch(const ConvArgsH &args, hipStream_t st) {
  const int grid = ((args.n_row_tiles + 7) / 8) * 8;
  const bool emit = args.out_split != nullptr;
#ifdef PH_DEV
  if (g_w2_trace_on && !emit) {        // never on a launch that has to write the next layer's operand
    hipLaunchKernelGGL((k_conv_wop2<false, true>), dim3(grid), dim3(256), 0, st, args);
    PH_LAUNCH_CHECK();
    return 0;
  }
#endif
  if (emit) hipLaunchKernelGGL((k_conv_wop2<true>), dim3(grid), dim3(256), 0, st, args);
  else hipLaunchKernelGGL((k_conv_wop2<false>), dim3(grid), dim3(256), 0, st, args);
  PH_LAUNCH_CHECK();
  return 0;
}
