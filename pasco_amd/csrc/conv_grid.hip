// Convolution over a FULL dense grid, split-precision products (round 6; kernel id 8, `k_conv_grid`): the dense bottleneck's
// (3, 3, 1) / (5, 5, 3) / (7, 7, 5) stages (layers.py:656-726: torch Conv3d on the densified stride-8 level) as implicit GEMMs.
//
// What bounds k_conv_dma on those launches, measured (conv_dma.hip, profiles/README.md "round 3"): the CU's texture-address unit -
// one LDS-DMA instruction per ~20 clocks, 32 of them per 128 x 128 x 32 stage, half of them the per-row GATHER of the activation
// rows.  On a dense grid the gather is arithmetic: with the sites enumerated (b, z, x, y), y fastest, and the kernel offsets
// enumerated y fastest too (ph_conv_desc.grid_dims / grid_kernel), the ky offsets of one (dx, dz) GROUP read the SAME run of
// input rows shifted by one row each - tile row r at offset iy reads row r + iy of the run
//     [first site of the tile + dx * Y + dz * X * Y - ky / 2,  + 256 + ky - 1).
// So: 256 x 128 tiles, 8 waves as 4 x 2, one workgroup per CU (k_conv_wide's shape for 128 columns).  A UNIT of work is one
// (group, 32-channel chunk): its run of <= 262 rows x 128 B is copied to an LDS WINDOW once (33 DMA instructions per workgroup)
// and serves ky stages; a stage loads only its 16 KB of weights.  Per 256 x 128 x 32 stage the vector-memory path moves
// 16 KB + 33 KB / ky instead of 48 KB (k_conv_wide) or 2 x 32 KB (two k_conv_dma tiles): the address unit is ~25 % busy.
// Whether a neighbour exists is a coordinate test per lane (x + dx and z + dz once per unit, y + dy per stage); rows without one read a
// zero line of LDS.  Groups none of the tile's rows can reach (dz beyond the 4-deep grid: 30 % of a (7, 7, 5) kernel) are dropped
// from the tile's unit list; the launch is split over the UNITS, interleaved (slice s of a tile walks its units s, s + slices, ...),
// partial sums reduced in slice order by k_splitk_epilogue - any number of slices balances to one unit; in a launch of one round of
// workgroups a tile takes as many slices as its unit count asks for (inner z planes reach a third more groups than outer ones).
//
// Measured (profiles/r6t_grid_ab.txt, r6w_grid_balance.txt, r6v_pmc_grid_wide2.txt): 455 / 177 / 51 us per launch for the 245 / 75 / 9-offset
// shapes against k_conv_dma's 470 - 540 / 195 - 213 / 52; matrix pipe 43 % busy.  The address unit is no longer what bounds it; a
// barrier-free variant with wave-private windows and a variant with wave roles were built, are parity-green and no faster (r6s).
//
// Pipeline = k_conv_wide's: two weight stage buffers, fragments at half-stage granularity, ONE barrier per stage in its middle;
// two window buffers - the next unit's window is fetched under this unit's first ky - 1 stages.
// The products and the per-accumulator order of the three split products are k_conv_dma's; the fp32 summation order over
// (offset, chunk) is this kernel's own (group, chunk, dy) - held to the oracle and to fp64 by tests/test_hip_grid.py.
#include <math.h>
#include <stdlib.h>

#include "conv_h2_common.h"

constexpr int GR_WROWS = 264;                 // window rows: 256 + ky - 1 <= 262, in pieces of 8 rows (one DMA instruction of a wave)
constexpr int GR_PIECES = GR_WROWS / 8;       // 33
constexpr int GR_ROUNDS = (GR_PIECES + 7) / 8;   // 5 rounds of 8 waves
constexpr int GR_WIN = GR_WROWS * 128;
constexpr int GR_BST = 128 * 128;             // one weight stage: 128 columns x 128 B
constexpr int GR_OFF_WIN = 2 * GR_BST;
constexpr int GR_OFF_ZERO = GR_OFF_WIN + 2 * GR_WIN;
constexpr int GR_OFF_MISC = GR_OFF_ZERO + 128;
constexpr int GR_LDS = GR_OFF_MISC + 64;

template <bool EMIT>
__global__ void __launch_bounds__(512, 2) k_conv_grid(ConvArgsH a) {
  constexpr int WN = 2, TM = 2, TN = 2;
  constexpr int NT = 512;
  constexpr int BM = 256, BN = 128;
  constexpr int NMH = TM * TN * 3;           // MFMAs per wave and half stage (12)
  constexpr int NRH = (TM + TN) * 2;         // fragment reads per wave and half stage (8)
  __shared__ __attribute__((aligned(128))) char lds[GR_LDS];

  // workgroup -> tile WITHOUT the XCD grouping of the other kernels: consecutive workgroups go to consecutive XCDs, so each XCD's
  // 32 CUs get tiles from every z plane of the grid (tiles of the inner planes carry a third more work than the outer ones; eight
  // neighbouring tiles on one XCD put 35 workgroups on the inner planes' 32 CUs).  What neighbouring tiles share - the weights -
  // all tiles share
  int tile = blockIdx.x;
  if (PH_ABLATE(a, 0x200)) tile = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);   // development build: the XCD grouping
  if (tile >= a.n_row_tiles * a.n_col_tiles) return;
  const int row_tile = tile / a.n_col_tiles;
  const int col_tile = tile - row_tile * a.n_col_tiles;
  const int64_t m0 = (int64_t)row_tile * BM;
  const int n0 = col_tile * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5;
  const int l31 = lane & 31;

  const int GX = a.gdim[1], GY = a.gdim[2], GZ = a.gdim[3];
  const int kx = a.gker[0], ky = a.gker[1], kz = a.gker[2];
  const int hx = kx >> 1, hy = ky >> 1, hz = kz >> 1;
  const int ngroups = kx * kz;
  const int cout = a.cout;
  const int nchunks = a.cpad >> 5;
  const uint32_t rsb = 4u * (uint32_t)a.cpad;
  const int slice = (int)blockIdx.y;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- which groups any row of the tile can reach: one bit per group ------------------------------------------------------------
  unsigned *gmask = reinterpret_cast<unsigned *>(lds + GR_OFF_MISC);
  if (tid < 2) gmask[tid] = 0u;
  if (tid < 32) reinterpret_cast<float *>(lds + GR_OFF_ZERO)[tid] = 0.f;
  __syncthreads();
  {
    const int r = tid & 255;
    const int64_t s = m0 + r;
    const bool ok = s < a.n_out;
    const int t = (int)((ok ? s : 0) / GY);
    const int x = t % GX, zz = (t / GX) % GZ;
    for (int g = tid >> 8; g < ngroups; g += 2) {
      const int ix = g % kx, iz = g / kx;
      const bool v = ok && (unsigned)(x + ix - hx) < (unsigned)GX && (unsigned)(zz + iz - hz) < (unsigned)GZ;
      if (__ballot(v) != 0ull && lane == 0) atomicOr(&gmask[g >> 5], 1u << (g & 31));
    }
  }
  __syncthreads();
  const unsigned long long gm =
      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(gmask[0]) |
      ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(gmask[1]) << 32);
  const int nvalid = __builtin_popcountll(gm);
  const int total_units = nvalid * nchunks;
  // slices of THIS tile: tiles in the grid's outer z planes reach fewer (dx, dz) groups than inner ones (21 against 28 of a (7, 7, 5)
  // kernel) - a tile takes as many of the launch's slices as keeps a workgroup's share near the launcher's target; the slices it
  // does not take store zeros (the reduction sums all of them)
  int nsl = a.ksplit;
  if (a.grid_upw > 0) {
    nsl = (total_units + (a.grid_upw >> 1)) / a.grid_upw;
    nsl = nsl < 1 ? 1 : (nsl > a.ksplit ? a.ksplit : nsl);
  }
  const int nunits = (slice < nsl && slice < total_units) ? (total_units - slice + nsl - 1) / nsl : 0;
  const int nstages = nunits * ky;

  if (nstages > 0) {
    // ---- units of this slice, walked in order: unit n = (vi-th reachable group, chunk c), slice + n * ksplit = vi * nchunks + c.
    // Every wave runs this bookkeeping on the CU's ONE scalar unit: no division, nothing per stage that can be had per unit ----
    struct Unit {
      int n, vi, g, c;
    };
    const int kq = nsl / nchunks, kr = nsl - kq * nchunks;
    const int inv_kx = 65536 / kx + 1;                      // g / kx = (g * inv_kx) >> 16 for g < 64, kx <= 9
    auto first_unit = [&]() {
      Unit u;
      u.n = 0;
      u.vi = slice / nchunks;
      u.c = slice - u.vi * nchunks;
      unsigned long long m = gm;
      for (int q = 0; q < u.vi; ++q) m &= m - 1;          // drop the vi lowest set bits
      u.g = __builtin_ctzll(m);
      return u;
    };
    auto next_unit = [&](const Unit &p) {
      Unit u;
      u.n = p.n + 1;
      u.c = p.c + kr;
      u.vi = p.vi + kq;
      if (u.c >= nchunks) {
        u.c -= nchunks;
        ++u.vi;
      }
      unsigned long long m = gm & ~((2ull << p.g) - 1ull);   // groups above p.g
      int g = p.g;
      for (int q = p.vi; q < u.vi; ++q) {
        g = __builtin_ctzll(m);
        m &= m - 1;
      }
      u.g = g;
      return u;
    };

    // ---- DMA geometry: weights as k_conv_wide (tile row l_r + 64 q, 16-byte slot l_j, source chunk swizzled) ----------------------
    const int l_j = tid & 7;
    const int l_r = tid >> 3;
    const uint32_t sj16 = (uint32_t)((l_j ^ ((l_r >> 1) & 7)) << 4);
    const uint64_t w_base = (uint64_t)reinterpret_cast<uintptr_t>(a.w_split) + sj16;
    uint32_t boff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int n = n0 + l_r + q * 64;
      n = n < cout ? n : cout - 1;
      boff[q] = (uint32_t)n * rsb;
    }
    const int64_t wslab = (int64_t)cout * rsb;
    const uint64_t in_base = (uint64_t)reinterpret_cast<uintptr_t>(a.in_split);
    const int wl_r = lane >> 3;                              // window piece: row of the piece, slot l_j
    const int n_last = (int)(a.n_in - 1);                    // < 2^31 (ph_conv_fwd_f16x3)
    const uint32_t wdst0 = (uint32_t)(wave * 1024);          // LDS offset of this wave's piece of round 0 inside a window
    const uint32_t bdst0 = (uint32_t)(wave * 1024);

    auto fire_b = [&](uint64_t src, char *bbuf, int q) {
      if (PH_ABLATE(a, 2)) return;
      char *dst = bbuf + bdst0 + q * 8192;                   // wave-uniform; the DMA adds lane * 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(src + boff[q]),
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    // window piece p = 8 round + wave of a unit whose run starts at site `first` (may be negative), chunk byte offset `coff`, window
    // buffer at LDS offset `wbuf`: window row R holds site first + R (clamped into the grid: rows beyond it are never read as
    // neighbours), 16-byte slot j holds source chunk j ^ ((R >> 1) & 7)
    auto fire_win = [&](int first, uint32_t coff, uint32_t wbuf, int round) {
      if (PH_ABLATE(a, 1)) return;
      const int R = round * 64 + wave * 8 + wl_r;
      int site = first + R;
      site = site < 0 ? 0 : (site > n_last ? n_last : site);
      const uint64_t src = in_base + (uint64_t)(uint32_t)site * rsb + coff + (uint32_t)((l_j ^ ((R >> 1) & 7)) << 4);
      char *dst = lds + wbuf + round * 8192 + wdst0;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    // per unit: where its window comes from
    struct WinSrc {
      int first;
      uint32_t coff, wbuf;
    };
    auto win_of = [&](const Unit &u) {
      const int iz = (u.g * inv_kx) >> 16, ix = u.g - iz * kx;
      WinSrc w;
      w.first = (int)m0 + (ix - hx) * GY + (iz - hz) * GX * GY - hy;
      w.coff = (uint32_t)u.c << 7;
      w.wbuf = (uint32_t)(GR_OFF_WIN + (u.n & 1) * GR_WIN);
      return w;
    };

    // ---- this lane's two activation rows (blocks i = 0, 1 of the wave): grid coordinates ------------------------------------------
    int ay[TM], ax[TM], az[TM];
    bool aok[TM];
    uint32_t rbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      rbase[i] = (uint32_t)(wm * 64 + i * 32 + l31);
      const int64_t s = m0 + rbase[i];
      aok[i] = s < a.n_out;
      const int sc = aok[i] ? (int)s : 0;
      const int t = sc / GY;
      ay[i] = sc - t * GY;
      ax[i] = t % GX;
      az[i] = (t / GX) % GZ;
    }
    const uint32_t hq = (uint32_t)h << 4;
    struct ASet {                    // where the lane's fragments of one stage sit: byte offset of the window row (or the zero line), its swizzle
      uint32_t row[TM], sw[TM];
    };
    bool gv[TM];
    uint32_t cwb = 0;                // window buffer of the unit whose products run
    auto unit_valid = [&](const Unit &u) {
      const int iz = (u.g * inv_kx) >> 16, ix = u.g - iz * kx;
      cwb = (uint32_t)(GR_OFF_WIN + (u.n & 1) * GR_WIN);
#pragma unroll
      for (int i = 0; i < TM; ++i)
        gv[i] = aok[i] && (unsigned)(ax[i] + ix - hx) < (unsigned)GX && (unsigned)(az[i] + iz - hz) < (unsigned)GZ;
    };
    auto a_set = [&](int iy, ASet &s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const uint32_t R = rbase[i] + (uint32_t)iy;
        const bool v = gv[i] && (unsigned)(ay[i] + iy - hy) < (unsigned)GY;
        s.row[i] = v ? cwb + R * 128u : (uint32_t)GR_OFF_ZERO;
        s.sw[i] = ((R >> 1) & 7u) << 4;
      }
    };

    // ---- fragments ------------------------------------------------------------------------------------------------------------
    const int sw = (l31 >> 1) & 7;
    uint32_t xo[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int part = 0; part < 2; ++part) xo[ks][part] = (uint32_t)(((part * 4 + ks * 2 + h) ^ sw) << 4);
    const uint32_t brow = (uint32_t)((wn * TN * 32 + l31) * 128);
    struct Frag {
      f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
    };
    Frag f;
    ASet aset;
    // the r-th of the NRH fragment reads of half `ks`: activations (window) hi / lo per i, then weights hi / lo per j
    auto read_one = [&](const char *bbuf, int ks, int r) {
      if (PH_ABLATE(a, 8)) return;
      if (r < 2 * TM) {
        const int i = r >> 1, part = r & 1;
        const uint32_t o = aset.row[i] + ((((uint32_t)(part * 4 + ks * 2) << 4) | hq) ^ aset.sw[i]);
        if (part == 0) f.ah[ks][i] = *reinterpret_cast<const f16x8 *>(lds + o);
        else f.al[ks][i] = *reinterpret_cast<const f16x8 *>(lds + o);
      } else {
        const int j = (r - 2 * TM) >> 1;
        if (((r - 2 * TM) & 1) == 0) f.bh[ks][j] = *reinterpret_cast<const f16x8 *>(bbuf + brow + j * 4096 + xo[ks][0]);
        else f.bl[ks][j] = *reinterpret_cast<const f16x8 *>(bbuf + brow + j * 4096 + xo[ks][1]);
      }
    };
    auto mfma_one = [&](int ks, int m) {
      if (PH_ABLATE(a, 4)) return;
      const int ij = m % (TM * TN), p = m / (TM * TN);
      const int i = ij / TN, j = ij % TN;
      if (p == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.al[ks][i], acc[i][j], 0, 0, 0);
      else if (p == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[ks][j], f.ah[ks][i], acc[i][j], 0, 0, 0);
    };

    // ---- positions: `cu` / `ciy` = the stage whose products run, `du` / `diy` = the stage whose weights are fetched next ------------
    Unit cu = first_unit(), du = cu;
    int ciy = 0, diy = 0;
    bool nu_ok = nunits > 1;
    Unit nu = nu_ok ? next_unit(cu) : cu;                    // the unit after `cu` (its window is fetched under cu's stages)
    WinSrc nw = win_of(nu);
    uint64_t wsrc = w_base + (uint64_t)((int64_t)(ky * du.g) * wslab) + ((uint32_t)du.c << 7);
    auto d_step = [&]() {                                     // the tail re-loads the last stage instead of branching
      if (du.n == nunits - 1 && diy == ky - 1) return;
      if (++diy == ky) {
        diy = 0;
        du = next_unit(du);
        wsrc = w_base + (uint64_t)((int64_t)(ky * du.g) * wslab) + ((uint32_t)du.c << 7);
      } else {
        wsrc += (uint64_t)wslab;
      }
    };
    // rounds of window pieces per stage: all GR_ROUNDS within the unit's first ky - 1 stages
    const int rps = (GR_ROUNDS + ky - 2) / (ky - 1);

    char *b0 = lds, *b1 = lds + GR_BST;
    // prologue: window of the first unit, weights of stages 0 and 1
    {
      const WinSrc w0 = win_of(cu);
      for (int q = 0; q < GR_ROUNDS; ++q)
        if (q * 8 + wave < GR_PIECES) fire_win(w0.first, w0.coff, w0.wbuf, q);
    }
    fire_b(wsrc, b0, 0);
    fire_b(wsrc, b0, 1);
    d_step();
    fire_b(wsrc, b1, 0);
    fire_b(wsrc, b1, 1);
    d_step();                                                 // wsrc: stage 2, fired in the second half of stage 0
    unit_valid(cu);
    a_set(0, aset);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < NRH; ++r) read_one(b0, 0, r);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // first half of a stage: its k = 0..15 products from registers, its k = 16..31 fragments from the window / `cur`
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
    // second half: the k = 16..31 products; the weights of the stage after next into `cur` (free: every wave passed the barrier
    // after its last read of it), this stage's share of the next unit's window, the first-half fragments of the next stage
    auto half_b = [&](char *cur, const char *nxt) {
      // this stage's rounds of the next unit's window (before the position moves on)
      int wq = (nu_ok && ciy < ky - 1) ? ciy * rps : GR_ROUNDS;
      const int wq_end = wq + rps;
      const WinSrc w = nw;
      // the next stage's position and fragment addresses
      if (!(cu.n == nunits - 1 && ciy == ky - 1)) {
        if (++ciy == ky) {
          ciy = 0;
          cu = nu;
          nu_ok = cu.n + 1 < nunits;
          if (nu_ok) {
            nu = next_unit(cu);
            nw = win_of(nu);
          }
          unit_valid(cu);
        }
      }
      if (!PH_ABLATE(a, 64)) a_set(ciy, aset);
#pragma unroll
      for (int g = 0; g < NRH; ++g) {
        if (g < 2) fire_b(wsrc, cur, g);
        if (g >= 2 && g < 5) {
          if (wq < wq_end && wq * 8 + wave < GR_PIECES) fire_win(w.first, w.coff, w.wbuf, wq);
          ++wq;
        }
        read_one(nxt, 0, g);
        if (g == 5) d_step();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = g * NMH / NRH; m < (g + 1) * NMH / NRH; ++m) mfma_one(1, m);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#define GRID_MIDDLE()                                                            \
  do {                                                                           \
    if (PH_ABLATE(a, 16)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             \
    if (!PH_ABLATE(a, 32)) __builtin_amdgcn_s_barrier();                         \
  } while (0)
    for (int s = 0; s < nstages; s += 2) {
      half_a(b0);
      GRID_MIDDLE();
      half_b(b0, b1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (s + 1 >= nstages) break;
      half_a(b1);
      GRID_MIDDLE();
      half_b(b1, b0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // clamped tail loads: nothing may land after the epilogue starts
#undef GRID_MIDDLE
  }

  h2_store_tile<TM, TN, EMIT>(a, acc, m0, n0, wm, wn, h, l31);
}

#ifdef PH_DEV
// development build only (tools/grid_ab.py): phase ablation of k_conv_grid - 1 window DMA, 2 weight DMA, 4 MFMAs, 8 fragment reads,
// 16 no vmcnt wait, 32 no barrier, 64 no fragment-address update.  Wrong sums, right instruction stream otherwise.
static int g_grid_ablate = 0;
extern "C" void ph_conv_grid_set_ablate(int mask) { g_grid_ablate = mask; }
#else
constexpr int g_grid_ablate = 0;
#endif

// reachable (dx, dz) groups of an average tile: all dx, the dz that stay inside the grid
static double grid_groups(const int *gd, const int *gk) {
  double zavg = 0.0;
  for (int z = 0; z < gd[3]; ++z)
    for (int iz = 0; iz < gk[2]; ++iz) zavg += (z + iz - gk[2] / 2 >= 0 && z + iz - gk[2] / 2 < gd[3]) ? 1.0 : 0.0;
  return gk[0] * zavg / gd[3];
}

// Serves mode-2 launches that carry the dense-grid promise (ph_conv_desc.grid_dims / grid_kernel) with 128 k output channels and whole
// 32-channel chunks; -1 = not served (the caller goes on to the gather kernels, which read `nbr`).  The split over the units comes
// from a cost model like ph_conv_fwd_f16x3's: rounds of one workgroup per CU x the longest slice + the reduction's round trip.
int ph_conv_grid_try(const ConvArgsH &a_in, hipStream_t st) {
  const int *gd = a_in.gdim, *gk = a_in.gker;
  if (gd[0] <= 0 || a_in.tile_k != nullptr || a_in.win_gather || (a_in.cpad & 31) || a_in.cout % 128 != 0 || a_in.in_split == nullptr ||
      a_in.axis_table != nullptr)
    return -1;
  if (a_in.route & PH_ROUTE_GRID_NEVER) return -1;
  const int kx = gk[0], ky = gk[1], kz = gk[2];
  if (ky < 3 || ky > 7 || !(kx & 1) || !(ky & 1) || !(kz & 1) || kx * kz > 64 || kx * ky * kz != a_in.kvol) return -1;
  const int64_t sites = (int64_t)gd[0] * gd[1] * gd[2] * gd[3];
  if (sites != a_in.n_out || sites != a_in.n_in || gd[2] < 1) return -1;
  ConvArgsH a = a_in;
  a.n_row_tiles = (int)((a.n_out + 255) / 256);
  a.n_col_tiles = a.cout / 128;
  const int64_t tiles = (int64_t)a.n_row_tiles * a.n_col_tiles;
  const int nchunks = a.cpad >> 5;
  const double units = grid_groups(gd, gk) * nchunks;
  const double stage_us = 1.0, fixed_us = 8.0;
  int best = 1;
  double best_us = 1e30;
  for (int ks = 1; ks <= 12; ++ks) {
    if (ks > 1 && (a.tail_ws == nullptr || a.tail_ws_bytes < (int64_t)ks * a.n_out * a.cout * 4)) break;
    if (ks > 1 && units / ks < 2.0) break;
    const double rounds = (double)((tiles * ks + 255) / 256);
    double us = rounds * (ceil(units / ks) * ky * stage_us + fixed_us);
    if (ks > 1) us += (double)ks * (double)a.n_out * a.cout * 4.0 / 3.0e6 + 5.0;
    if (us < best_us) best_us = us, best = ks;
  }
  if (const char *e = PH_DEV_ENV("PASCO_GRID_KS")) {              // development build: the number of slices by hand (tools/grid_ab.py)
    const int ks = atoi(e);
    if (ks >= 1 && (ks == 1 || (a.tail_ws != nullptr && a.tail_ws_bytes >= (int64_t)ks * a.n_out * a.cout * 4))) best = ks;
  }
  a.grid_upw = 0;
  if (best > 1 && tiles * best <= 256) {
    // ONE round of workgroups: the launch lasts as long as its longest workgroup, and the model's slices are for a tile with the
    // AVERAGE number of reachable groups - tiles in the grid's inner z planes reach a third more, a tile across two planes more still.
    // Aim at the same units per workgroup everywhere, with as many slices in the launch as the fullest tile takes (measured on
    // the 75-offset launches: -5 .. -8 %).  Launches of several rounds balance by themselves - their workgroups are handed out as CUs
    // fall free - and lose 12 % to the extra slices (245 offsets: 9 -> 12): profiles/r6w_grid_balance.txt
    const int upw = (int)(units / best + 0.5) > 0 ? (int)(units / best + 0.5) : 1;
    int zmax = 0;
    for (int z = 0; z < gd[3]; ++z) {
      int c = 0;
      for (int iz = 0; iz < kz; ++iz) c += (z + iz - kz / 2 >= 0 && z + iz - kz / 2 < gd[3]) ? 1 : 0;
      zmax = c > zmax ? c : zmax;
    }
    const int zreach = zmax + 1 < kz ? zmax + 1 : kz;               // a tile across a plane boundary
    int ksmax = (kx * zreach * nchunks + upw / 2) / upw;
    ksmax = ksmax > 12 ? 12 : ksmax;
    while (ksmax > best && a.tail_ws_bytes < (int64_t)ksmax * a.n_out * a.cout * 4) --ksmax;
    const char *e = PH_DEV_ENV("PASCO_GRID_UPW");                  // development build: 0 = the same slices for every tile
    if (e == nullptr || atoi(e) != 0) {
      if (ksmax > best) best = ksmax;
      a.grid_upw = upw;
    }
  }
  a.ksplit = best;
  a.partial = best > 1 ? (float *)a.tail_ws : nullptr;
  a.win_stats = nullptr;
  a.ablate = g_grid_ablate;
  const int grid = (int)((tiles + 7) / 8) * 8;
  const bool emit = a.out_split != nullptr && a.ksplit == 1;
  if (emit) hipLaunchKernelGGL((k_conv_grid<true>), dim3(grid, 1), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((k_conv_grid<false>), dim3(grid, a.ksplit), dim3(512), 0, st, a);
  PH_LAUNCH_CHECK();
  if (a.ksplit > 1) {
    if (int rc = ph_launch_splitk_epilogue(a, st)) return rc;
  }
  ph_record_cfg(2, 256, 128, 32, a.ksplit, emit ? 1 : 0, 8, 8);
  return 0;
}
