// Shared pieces of the split-precision convolution kernels (conv_f16x3.hip: k_conv_f16x3 / k_conv_h2,
// conv_dma.hip: k_conv_dma): argument block, activation / range helpers, the tile epilogue.
#pragma once
#include <type_traits>

#include "ph_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int HV_THREADS = 256;
// KC input channels per stage (32 or 64); LDS rows hold KC + 8 f16 (80 / 144 bytes: 16-byte aligned and
// conflict-free for the b128 fragment reads)

__device__ __forceinline__ float h_act(float v, float neg) { return ph_act(v, neg); }
// range flag of a value about to become an f16 operand: true for |t| > 65504 AND for NaN
__device__ __forceinline__ bool h_out_of_range(float t) { return !(fabsf(t) <= 65504.f); }

struct ConvArgsH {
  const float *in;
  const _Float16 *w_hi;   // [kvol][cout][cin]
  const _Float16 *w_lo;
  const int32_t *nbr;
  float *out;
  int64_t n_in, n_out;
  int cin, cout, kvol;
  const float *pro_scale, *pro_shift, *bias, *epi_scale, *epi_shift, *epi2_scale, *epi2_shift, *residual;
  float pro_neg, epi_neg, res_neg, w_unscale;
  int has_pro, has_tail;
  int n_row_tiles, n_col_tiles;
  int32_t *status;
  int ksplit;       // > 1: blockIdx.y walks one slice of the kernel offsets and stores raw partial sums
  float *partial;   // [ksplit][n_out][cout]
  // mode 2 (both operands pre-split by ph_split_rows): rows of cpad/32 groups [hi x32 | lo x32]
  const _Float16 *in_split;   // [n_in][cpad/32][2][32]
  const _Float16 *w_split;    // [kvol][cout][cpad/32][2][32]
  const _Float16 *w_frag;     // optional (ph_conv_desc.w_frag): the same values in fragment order [kvol][cpad/16][2][2][64][8]
  int cpad;
  // optional second output: split operand of act(out * osp_scale + osp_shift) for the next convolution
  _Float16 *out_split;        // [n_out][cout/32][2][32]  (cout % 32 == 0)
  const float *osp_scale, *osp_shift;
  float osp_neg;
  int osp_has;
  float act_pow2;             // 2^split_exp2: scale of the activation operand (mode 1 gather, emitted out_split)
  const char *zero;           // k_conv_dma: >= 256 zero bytes in device memory (rows without a neighbour read them)
  int ablate;                 // development build only (PH_ABLATE): 0 in the product library, its branches compile out
  int route;                  // ph_conv_desc.route (PH_ROUTE_*): per-call kernel choice of the parity tests, 0 = the library decides
  // LDS-window tables of the kernel map (conv_win.hip); win_stats != nullptr: the launch is one of a window / gather
  // pair and returns at once unless the device-side predicate picks it
  const int32_t *win_rows, *win_cnt, *win_stats;
  const uint16_t *win_slots;
  int win_which;              // statistics slot of the tile shape (0: 64-wide / 416 rows, 1: 128-wide / 512 rows)
  int win_gather;             // 1 on the gather kernel of the pair (inverted predicate)
  // per-axis table residual (ph_conv_desc.axis_table): [3][axis_rows][cout], coordinates [n_out][4]
  const float *axis_table;
  const int32_t *axis_coords;
  int axis_lo, axis_rows;
  // row lists (ph_conv_desc.rl_*): output row of tile row r = out_rows[r] (-1 = padding), kernel offset of row tile t = tile_k[t]
  const int32_t *out_rows, *tile_k;
  int par_vec;                // every per-channel vector (bias, epi*, osp*) is 16-byte aligned: the epilogue loads them as float4
  void *tail_ws;              // host-side only: the caller's split scratch (ph_conv_desc.splitk_ws) for ph_conv_dma_try's tail split
  int64_t tail_ws_bytes;
  int gdim[4], gker[3];       // ph_conv_desc.grid_dims / grid_kernel (all zero: no dense-grid promise); conv_grid.hip
  int grid_upw;               // k_conv_grid: units of work per workgroup the launcher aims at (a tile takes round(its units / this) slices)
  int64_t nbr_stride;         // k_conv_dma: rows of one offset's segment of `nbr` (= the map's n_out; a launch over a row range of
                              // the map has its own, smaller n_out and row-shifted pointers - ph_conv_dma_try's tail split)
};

// element offsets of the three table rows (x, y, z) of one output row (coordinates clamped to the table)
// -> true when a coordinate lies outside the table (its row was clamped: the caller raises status bit 2)
__device__ __forceinline__ bool ph_axis_offsets(const ConvArgsH &a, int64_t row, int64_t (&off)[3]) {
  const int4 c = *reinterpret_cast<const int4 *>(a.axis_coords + row * 4);
  const int v[3] = {c.y, c.z, c.w};
  bool clamped = false;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    int idx = v[ax] - a.axis_lo;
    clamped |= idx < 0 || idx >= a.axis_rows;
    idx = idx < 0 ? 0 : (idx >= a.axis_rows ? a.axis_rows - 1 : idx);
    off[ax] = ((int64_t)ax * a.axis_rows + idx) * a.cout;
  }
  return clamped;
}
// t0[x] + t1[y] + t2[z] for 4 consecutive channels of one output row
__device__ __forceinline__ float4 ph_axis_residual4(const ConvArgsH &a, const int64_t (&off)[3], int col) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float4 t = *reinterpret_cast<const float4 *>(a.axis_table + off[ax] + col);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  return s;
}
__device__ __forceinline__ float4 ph_axis_residual4(const ConvArgsH &a, int64_t row, int col) {
  int64_t off[3];
  ph_axis_offsets(a, row, off);
  return ph_axis_residual4(a, off, col);
}
// 4 consecutive entries of a per-channel vector (or `dflt` when the vector is absent / the run is past the last channel)
__device__ __forceinline__ void ph_par4(const ConvArgsH &a, const float *p, int col, bool ok, float dflt, float (&dst)[4]) {
  dst[0] = dst[1] = dst[2] = dst[3] = dflt;
  if (p == nullptr || !ok) return;
  if (a.par_vec) {
    const float4 v = *reinterpret_cast<const float4 *>(p + col);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = p[col + q];
  }
}

// mean window passes per tile <= 1.25 -> the window kernel serves the map (which: bit 8 / 9 = test override:
// always / never windows)
__device__ __forceinline__ bool ph_win_pred(const int32_t *stats, int which, int64_t n_row_tiles) {
  if (which & 0x100) return true;
  if (which & 0x200) return false;
  return (int64_t)stats[which & 1] * 4 <= n_row_tiles * 5;
}
// ph_conv_desc.route -> the predicate override bits of win_which
static inline int ph_win_force_bits(int route) { return (route & PH_ROUTE_WIN_ALWAYS) ? 0x100 : ((route & PH_ROUTE_WIN_NEVER) ? 0x200 : 0); }
// conv_dma.hip: 256 zero bytes of device memory for absent neighbours
const char *ph_dma_zero_line();
int ph_dma_ablate_bits();   // 0 in the product library (development build: tools/dma_ablate.py)
// conv_win.hip
int ph_conv_win_launch(const ConvArgsH &a, int bn, hipStream_t st);
// conv_wop.hip: the window kernel of the 64-wide outputs (args.n_row_tiles set by the caller)
int ph_conv_wop2_launch(const ConvArgsH &args, hipStream_t st);

// conv_f16x3.hip: reduction + epilogue of a split over the kernel offsets (after a launch with args.ksplit > 1)
int ph_launch_splitk_epilogue(const ConvArgsH &args, hipStream_t st);
// conv_dma.hip: the LDS-DMA pipelined kernel; -1 = shape not served (caller falls back to k_conv_h2)
int ph_conv_dma_try(const ConvArgsH &a, int bn, hipStream_t st);
// conv_lin.hip: k = 1 products of 64 / 128 input channels as a row stream (weights in LDS, no barriers); -1 = shape not served
int ph_conv_lin_try(const ConvArgsH &a, hipStream_t st);
// conv_wide.hip: 256 x 256 tiles, 8 waves, one workgroup per CU (256 output channels); -1 = shape not served
int ph_conv_wide_try(const ConvArgsH &a, hipStream_t st);
// conv_grid.hip: launches with the dense-grid promise (a.gdim), activations from LDS windows; -1 = shape not served
int ph_conv_grid_try(const ConvArgsH &a, hipStream_t st);

// hi / lo halves of four values -> the [hi x32 | lo x32] group layout (dst points at the run's hi slot)
__device__ __forceinline__ bool emit_split4(const float v[4], const float *sc, const float *sh, int has, float neg,
                                            float pow2, _Float16 *dst) {
  f16x4 hi, lo;
  bool bad = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float t = v[q];
    if (has) {   // separate multiply and add, like ph_split_rows and the C restatement
#pragma clang fp contract(off)
      const float m = t * (sc ? sc[q] : 1.f);
      t = h_act(m + (sh ? sh[q] : 0.f), neg);
    }
    t *= pow2;
    bad |= h_out_of_range(t);
    const _Float16 th = (_Float16)t;
    hi[q] = th;
    lo[q] = (_Float16)(t - (float)th);
  }
  *reinterpret_cast<f16x4 *>(dst) = hi;
  *reinterpret_cast<f16x4 *>(dst + 32) = lo;
  return bad;
}



// Where the epilogue reads the per-channel vectors from: global memory (default), or a copy the kernel staged in LDS
// (h2_stage_params).  `which`: 0 bias, 1 epi_scale, 2 epi_shift, 3 epi2_scale, 4 epi2_shift, 5 osp_scale, 6 osp_shift.
//
// Why the staged forms exist (round 3, profiles/README.md "epilogue round trips"): vector loads and stores share the in-order
// vmcnt counter, and the stores of the epilogue may alias anything the compiler knows - so in the plain form the loads of the
// second group of channels (per-channel vectors, table rows, residual) wait for the STORES of the first to complete, and so
// on: four store round trips per tile (~2.5 us each under load), 10 of the 20 us a workgroup of a k = 1 launch lives.  Staged:
// every load of the epilogue is issued before its first store (per-channel vectors cooperatively into LDS, each lane's table +
// residual values into LDS slots of its own), then the stores go out back to back.
struct ParGlobal {
  __device__ __forceinline__ void get(const ConvArgsH &a, int which, const float *p, int col, bool ok, float dflt,
                                      float (&dst)[4]) const {
    (void)which;
    ph_par4(a, p, col, ok, dflt, dst);
  }
};
typedef __attribute__((address_space(3))) float lds_float;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_float4;
constexpr int H2_PAR_VECS = 7;
struct ParLds {          // [H2_PAR_VECS][bn] floats in LDS, defaults filled in (absent vector, channel past the last)
  const lds_float *base;
  int n0, bn;
  __device__ __forceinline__ void get(const ConvArgsH &a, int which, const float *p, int col, bool ok, float dflt,
                                      float (&dst)[4]) const {
    (void)a; (void)p; (void)ok; (void)dflt;
    const f32x4 v = *reinterpret_cast<const lds_float4 *>(base + which * bn + (col - n0));
    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
  }
};
// every thread of the workgroup (nt of them, nt >= bn) calls it; a barrier must follow before ParLds is used
__device__ __forceinline__ void h2_stage_params(const ConvArgsH &a, int n0, int bn, int tid, lds_float *dst) {
  if (tid >= bn) return;
  const int col = n0 + tid;
  const bool ok = col < a.cout;
#define H2_STAGE_ONE(v, ptr, dflt) dst[(v) * bn + tid] = ((ptr) != nullptr && ok) ? (ptr)[col] : (dflt)
  H2_STAGE_ONE(0, a.bias, 0.f);
  H2_STAGE_ONE(1, a.epi_scale, 1.f);
  H2_STAGE_ONE(2, a.epi_shift, 0.f);
  H2_STAGE_ONE(3, a.epi2_scale, 1.f);
  H2_STAGE_ONE(4, a.epi2_shift, 0.f);
  H2_STAGE_ONE(5, a.osp_scale, 1.f);
  H2_STAGE_ONE(6, a.osp_shift, 0.f);
#undef H2_STAGE_ONE
}

// The tail's per-element addends (table rows + dense residual, summed in the plain form's order) of one lane: read where they
// are used (TailInline), or from LDS slots the lane filled before its first store (h2_stage_tail).
struct TailInline {
  static constexpr bool staged = false;
  __device__ __forceinline__ float4 get(int slot) const { (void)slot; return make_float4(0.f, 0.f, 0.f, 0.f); }
};
struct TailLds {
  static constexpr bool staged = true;
  const lds_float4 *slots;   // [slot][nt] float4, this lane's column: slots + tid
  int nt;
  __device__ __forceinline__ float4 get(int slot) const {
    const f32x4 v = slots[slot * nt];
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};
// ... or in REGISTERS of the lane (a kernel whose accumulators die before its epilogue and that has something to do between the loads
// and the stores: conv_wop.hip issues them ahead of its reduction): slot numbers are compile-time constants after unrolling
template <int NSLOT>
struct TailRegs {
  static constexpr bool staged = true;
  float4 v[NSLOT];
  __device__ __forceinline__ float4 get(int slot) const { return v[slot]; }
};
// h2_stage_tail's loads into a TailRegs (all TN column blocks; slot = ((i * TN + j) * 2 + m) * 2 + u)
template <int TM, int TN>
__device__ __forceinline__ void h2_load_tail_regs(const ConvArgsH &a, int64_t m0, int n0, int wm, int wn, int h, int l31,
                                                  TailRegs<TM * TN * 4> &t) {
  const int cout = a.cout;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t r = m0 + (wm * TM + i) * 32 + l31;
    const int64_t orow = r < a.n_out ? (a.out_rows ? (int64_t)a.out_rows[r] : r) : -1;
    int64_t axis_off[3] = {0, 0, 0};
    if (a.axis_table && orow >= 0) {
      if (ph_axis_offsets(a, orow, axis_off) && a.status != nullptr) atomicOr(a.status, 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = n0 + (wn * TN + j) * 32 + 16 * m + 8 * u + 4 * h;
          const bool ok = orow >= 0 && col < cout;
          float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.residual && ok) rs = *reinterpret_cast<const float4 *>(a.residual + orow * cout + col);
          if (a.axis_table && ok) {
            const float4 tb = ph_axis_residual4(a, axis_off, col);
            rs = make_float4(tb.x + rs.x, tb.y + rs.y, tb.z + rs.z, tb.w + rs.w);
          }
          t.v[((i * TN + j) * 2 + m) * 2 + u] = rs;
        }
  }
}
// slot of (i, j, m, u) within passes of JB column blocks: ((i * JB + (j - j0)) * 2 + m) * 2 + u
template <int TM, int TN, int JB>
__device__ __forceinline__ void h2_stage_tail(const ConvArgsH &a, int64_t m0, int n0, int wm, int wn, int h, int l31, int j0,
                                              lds_float4 *slots, int nt) {
  const int cout = a.cout;
  int64_t orow[TM];
  int64_t axis_off[TM][3];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t r = m0 + (wm * TM + i) * 32 + l31;
    orow[i] = r < a.n_out ? (a.out_rows ? (int64_t)a.out_rows[r] : r) : -1;
    axis_off[i][0] = axis_off[i][1] = axis_off[i][2] = 0;
    if (a.axis_table && orow[i] >= 0 && j0 == 0) {
      if (ph_axis_offsets(a, orow[i], axis_off[i]) && a.status != nullptr) atomicOr(a.status, 4);
    } else if (a.axis_table && orow[i] >= 0) {
      ph_axis_offsets(a, orow[i], axis_off[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jj = 0; jj < JB; ++jj)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = n0 + (wn * TN + j0 + jj) * 32 + 16 * m + 8 * u + 4 * h;
          const bool ok = orow[i] >= 0 && col < cout;
          float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.residual && ok) rs = *reinterpret_cast<const float4 *>(a.residual + orow[i] * cout + col);
          if (a.axis_table && ok) {
            const float4 t = ph_axis_residual4(a, axis_off[i], col);
            rs = make_float4(t.x + rs.x, t.y + rs.y, t.z + rs.z, t.w + rs.w);
          }
          f32x4 v;
          v[0] = rs.x; v[1] = rs.y; v[2] = rs.z; v[3] = rs.w;
          slots[(((i * JB + jj) * 2 + m) * 2 + u) * nt] = v;
        }
}

// `J0`, `JB`: the column blocks j = J0 .. J0 + JB - 1 of the wave's TN (a staged tail whose slots do not all fit in LDS goes
// in passes; the raw-partial-sum form ignores them and must be called with the full range).
template <int TM, int TN, bool EMIT, class PAR = ParGlobal, class TAILV = TailInline, int J0 = 0, int JB = TN>
__device__ __forceinline__ void h2_store_tile(const ConvArgsH &a, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm,
                                              int wn, int h, int l31, const PAR &par = PAR(), const TAILV &tailv = TAILV()) {
  const int cout = a.cout;
  // accumulator layout (transposed block): acc[i][j][4g + q] = out[row = m0 + (wm*TM+i)*32 + l31]
  //                                                          [col = n0 + (wn*TN+j)*32 + 8g + 4h + q]
  if (a.ksplit > 1) {   // raw partial sums; k_splitk_epilogue reduces them in a fixed order
    float *part = a.partial + (int64_t)blockIdx.y * a.n_out * cout;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t row = m0 + (wm * TM + i) * 32 + l31;
      if (row >= a.n_out) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + (wn * TN + j) * 32 + 8 * g + 4 * h;
          if (col >= cout) continue;
          *reinterpret_cast<float4 *>(part + row * cout + col) =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
    }
    return;
  }

  // output row of each of this lane's tile rows (-1: beyond the map / list padding); row lists name it explicitly
  int64_t orow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t r = m0 + (wm * TM + i) * 32 + l31;
    orow[i] = r < a.n_out ? (a.out_rows ? (int64_t)a.out_rows[r] : r) : -1;
  }
  // table residual: the three table rows of each of this lane's rows, found once per tile (or by the caller: axis_pre)
  int64_t axis_off[TM][3];
  if (a.axis_table && !TAILV::staged) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int64_t row = orow[i];
      if (row >= 0) {
        if (ph_axis_offsets(a, row, axis_off[i]) && a.status != nullptr) atomicOr(a.status, 4);
      } else {
        axis_off[i][0] = axis_off[i][1] = axis_off[i][2] = 0;
      }
    }
  }
  // VALU diet (round 3: the k = 1 launches are bound by the epilogue's VALU work - ~35 instructions per output value before):
  // identity activations are skipped (h_act(v, 1) == v bit for bit), the range flag is one compare into a wave mask, the
  // half-wave exchange is one v_permlane32_swap per pair of values
  unsigned long long obad = 0;
  // row base pointers, once per tile (the 64-bit row products do not belong next to every store)
  // (typed as global memory: a pointer that went through the register pin below would otherwise be stored through with
  // FLAT instructions, which count on lgkmcnt as well - every LDS read after them would wait for the stores)
  typedef __attribute__((address_space(1))) float gfloat;
  typedef __attribute__((address_space(1))) _Float16 ghalf;
  typedef __attribute__((address_space(1))) f16x8 ghalf8;
  gfloat *out_row[TM];
  ghalf *osp_row[TM];
  const int colbase = n0 + wn * TN * 32 + 4 * h;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (TM * TN > 4) {
      out_row[i] = nullptr, osp_row[i] = nullptr;
      continue;
    }
    const int64_t row = orow[i] >= 0 ? orow[i] : 0;
    out_row[i] = (gfloat *)(a.out + row * cout + colbase);
    osp_row[i] = EMIT ? (ghalf *)(a.out_split + row * (cout >> 5) * 64 + (colbase >> 5) * 64 + (colbase & 31) + 4 * h) : nullptr;
    if (TM * TN <= 4) {      // keep them in registers (the compiler would re-derive them at every store)
      asm volatile("" : "+v"(out_row[i]));
      if (EMIT) asm volatile("" : "+v"(osp_row[i]));
    }
  }
  // instances: with / without activations, with / without an operand prologue (no per-value branches or selects)
  // (the 256 x 256 tiles of k_conv_wide have no registers to spare: there the prologue stays a run-time flag and the
  // addresses are derived at the stores - the extra instances and the pinned pointers cost it spills INSIDE its main loop)
  constexpr bool FULL = TM * TN <= 4;
  auto body = [&](auto acts_tag, auto osp_tag, auto tail_tag) {
  constexpr bool ACTS = decltype(acts_tag)::value;
  const bool TAIL = decltype(tail_tag)::value == 2 ? a.has_tail != 0 : decltype(tail_tag)::value == 1;
  const bool OSP = decltype(osp_tag)::value == 2 ? a.osp_has != 0 : decltype(osp_tag)::value == 1;
#pragma unroll
  for (int j = J0; j < J0 + JB; ++j)
#pragma unroll
    for (int m = 0; m < 2; ++m) {       // pairs of 4-channel runs: g = 2m, 2m + 1
      const int cbase = n0 + (wn * TN + j) * 32 + 16 * m;
      if (cbase >= cout) continue;      // uniform over the wave (cout % 4 == 0; with out_split cout % 32 == 0)
      float bias[2][4], es[2][4], eb[2][4], es2[2][4], eb2[2][4];
      bool cok[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = cbase + 8 * u + 4 * h;
        cok[u] = col < cout;
        par.get(a, 0, a.bias, col, cok[u], 0.f, bias[u]);
        par.get(a, 1, a.epi_scale, col, cok[u], 1.f, es[u]);
        par.get(a, 2, a.epi_shift, col, cok[u], 0.f, eb[u]);
        par.get(a, 3, a.epi2_scale, col, cok[u], 1.f, es2[u]);
        par.get(a, 4, a.epi2_shift, col, cok[u], 0.f, eb2[u]);
      }
      // operand emission: after a half-wave exchange this lane owns 8 consecutive channels cbase + 8h .. + 7
      float sc[8], sh[8];
      if (EMIT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) sc[q] = 1.f, sh[q] = 0.f;
        if (OSP) {
          float t4[4];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            par.get(a, 5, a.osp_scale, cbase + 8 * h + 4 * u, true, 1.f, t4);
#pragma unroll
            for (int q = 0; q < 4; ++q) sc[4 * u + q] = t4[q];
            par.get(a, 6, a.osp_shift, cbase + 8 * h + 4 * u, true, 0.f, t4);
#pragma unroll
            for (int q = 0; q < 4; ++q) sh[4 * u + q] = t4[q];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t row = orow[i];
        const bool rok = row >= 0;
        float v[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int g = 2 * m + u;
          const int col = cbase + 8 * u + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[u][q] = (acc[i][j][4 * g + q] * a.w_unscale + bias[u][q]) * es[u][q] + eb[u][q];
          if (ACTS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = h_act(v[u][q], a.epi_neg);
          }
          if (TAIL) {
            float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TAILV::staged) {
              rs = tailv.get(((i * JB + (j - J0)) * 2 + m) * 2 + u);
            } else {
              if (a.residual && rok && cok[u]) rs = *reinterpret_cast<const float4 *>(a.residual + row * cout + col);
              if (a.axis_table && rok && cok[u]) {   // table rows first, then the dense residual (as the C restatement)
                const float4 t = ph_axis_residual4(a, axis_off[i], col);
                rs = make_float4(t.x + rs.x, t.y + rs.y, t.z + rs.z, t.w + rs.w);
              }
            }
            const float r4[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) v[u][q] = v[u][q] * es2[u][q] + eb2[u][q] + r4[q];
            if (ACTS) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[u][q] = h_act(v[u][q], a.res_neg);
            }
          }
          if ((!EMIT || a.out) && rok && cok[u])
          {
            f32x4 o;
            o[0] = v[u][0]; o[1] = v[u][1]; o[2] = v[u][2]; o[3] = v[u][3];
            if (FULL) *reinterpret_cast<__attribute__((address_space(1))) f32x4 *>(out_row[i] + (j * 32 + 16 * m + 8 * u)) = o;
            else *reinterpret_cast<f32x4 *>(a.out + row * cout + col) = o;
          }
        }
        if (EMIT) {
          // lanes l and l ^ 32 hold the same row: h = 0 keeps run u = 0 and takes the partner's u = 0 (channels
          // +4..7); h = 1 takes the partner's u = 1 (channels +8..11) and keeps its own u = 1
          // (v_permlane32_swap: lanes 32..63 of the first register <-> lanes 0..31 of the second)
          float w8[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0][q]), __float_as_uint(v[1][q]), false, false);
            w8[q] = __uint_as_float(sw[0]);
            w8[4 + q] = __uint_as_float(sw[1]);
          }
          if (rok) {
            f16x8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              float t = w8[q];
              if (OSP) {   // separate multiply and add, like ph_split_rows and the C restatement
#pragma clang fp contract(off)
                const float mm = t * sc[q];
                t = mm + sh[q];
                if (ACTS) t = h_act(t, a.osp_neg);
              }
              t *= a.act_pow2;
              obad |= __builtin_amdgcn_fcmpf(fabsf(t), 65504.f, 10);     // unordered or greater: |t| > 65504 or NaN
              const _Float16 th = (_Float16)t;
              hi[q] = th;
              lo[q] = (_Float16)(t - (float)th);
            }
            // column cbase + 8 h of the row: group (col >> 5), slot (col & 31); osp_row holds the lane's part
            const int col8 = cbase + 8 * h;
            if (FULL) {
              ghalf *dst = osp_row[i] + j * 64 + 16 * m;
              *reinterpret_cast<ghalf8 *>(dst) = hi;
              *reinterpret_cast<ghalf8 *>(dst + 32) = lo;
            } else {
              _Float16 *dst = a.out_split + (row * (cout >> 5) + (col8 >> 5)) * 64 + (col8 & 31);
              *reinterpret_cast<f16x8 *>(dst) = hi;
              *reinterpret_cast<f16x8 *>(dst + 32) = lo;
            }
          }
        }
      }
    }
  };
  const bool acts = a.epi_neg != 1.f || a.res_neg != 1.f || a.osp_neg != 1.f;
  typedef std::integral_constant<int, 0> off_t;
  typedef std::integral_constant<int, 1> on_t;
  typedef std::integral_constant<int, 2> runtime_t;
  auto with_acts = [&](auto osp_tag, auto tail_tag) {
    if (acts) body(std::true_type(), osp_tag, tail_tag);
    else body(std::false_type(), osp_tag, tail_tag);
  };
  if (!FULL) {
    if (a.has_tail) with_acts(runtime_t(), on_t());
    else with_acts(runtime_t(), off_t());
  } else if (EMIT && a.osp_has) {
    if (a.has_tail) with_acts(on_t(), on_t());
    else with_acts(on_t(), off_t());
  } else {
    if (a.has_tail) with_acts(off_t(), on_t());
    else with_acts(off_t(), off_t());
  }
  if (EMIT && a.status != nullptr && obad != 0) atomicOr(a.status, 1);
}

#ifndef H2_STAMP
#define H2_STAMP(i) do {} while (0)        // conv_dma.hip's development trace defines it (tools/dma_trace.py)
#endif

// Staged epilogue of a one-tile workgroup (see ParGlobal): `scratch` = SCRATCH bytes of LDS that nothing reads or writes any
// more once every wave has arrived here (the stage buffers and index table of the main loop).  NT threads, BN-wide tile.
template <int TM, int TN, bool EMIT, int NT, int BN, int SCRATCH, int J0 = 0>
__device__ __forceinline__ void h2_staged_passes(const ConvArgsH &a, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm, int wn,
                                                 int h, int l31, const ParLds &pl, lds_float4 *slots) {
  constexpr int PAR_BYTES = H2_PAR_VECS * BN * 4;
  constexpr int PER_J = TM * 4 * NT * 16;
  constexpr int FIT = (SCRATCH - PAR_BYTES) / PER_J;
  static_assert(FIT >= 1, "LDS scratch too small for one column block of tail slots");
  constexpr int JB = FIT < TN - J0 ? FIT : TN - J0;
  h2_stage_tail<TM, TN, JB>(a, m0, n0, wm, wn, h, l31, J0, slots, NT);
  if (J0 == 0) {   // the per-channel vectors every thread staged: visible after the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    H2_STAMP(6);
  }
  h2_store_tile<TM, TN, EMIT, ParLds, TailLds, J0, JB>(a, acc, m0, n0, wm, wn, h, l31, pl, TailLds{slots, NT});
  if constexpr (J0 + JB < TN) h2_staged_passes<TM, TN, EMIT, NT, BN, SCRATCH, J0 + JB>(a, acc, m0, n0, wm, wn, h, l31, pl, slots);
}
template <int TM, int TN, bool EMIT, int NT, int BN, int SCRATCH>
__device__ __forceinline__ void h2_store_tile_staged(const ConvArgsH &a, f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm,
                                                     int wn, int h, int l31, int tid, char *scratch) {
  if (a.ksplit > 1) {
    h2_store_tile<TM, TN, EMIT>(a, acc, m0, n0, wm, wn, h, l31);
    return;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // every wave is done with the main loop's LDS
  H2_STAMP(5);
  lds_float *par = (lds_float *)scratch;
  h2_stage_params(a, n0, BN, tid, par);
  const ParLds pl{par, n0, BN};
  if (a.residual == nullptr && a.axis_table == nullptr) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    H2_STAMP(6);
    h2_store_tile<TM, TN, EMIT, ParLds>(a, acc, m0, n0, wm, wn, h, l31, pl);
    return;
  }
  lds_float4 *slots = (lds_float4 *)(scratch + H2_PAR_VECS * BN * 4) + tid;
  h2_staged_passes<TM, TN, EMIT, NT, BN, SCRATCH>(a, acc, m0, n0, wm, wn, h, l31, pl, slots);
}
