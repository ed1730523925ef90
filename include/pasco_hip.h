/*
 * pasco_hip.h -- flat C ABI of libpascohip.so (MI355X / gfx950 sparse-voxel operator library).
 *
 * This is the drop-in boundary of the hot path: the entry points a
 * MinkowskiEngine-style Python frontend binds instead of upstream's pybind11 module
 * `MinkowskiEngineBackend._C` (CoordinateMapManager / ConvolutionForward / LocalPoolingForward /
 * PruningForward ...).  MinkowskiEngine v0.5.4 is pinned by the reference at README.md:90 and is
 * NOT vendored under /root/reference, so each entry point cites the PaSCo call site whose
 * `ME.*` operator it serves.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / torch.cuda storage) unless named `h_*`;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises;
 *   - the library never allocates or frees caller tensors; scratch comes in through `ws`;
 *   - coordinates are int32 rows (batch, x, y, z); features are fp32 row-major [N, C];
 *   - row indices are int32; "-1" means "no row";
 *   - return value 0 = ok, non-zero = error, text via ph_last_error() (thread local).
 *
 * The same header is compiled with -DPH_ORACLE by oracle/ to give the CPU restatement the
 * identical signatures under the `pho_` prefix (host pointers, `stream` ignored).  The oracle
 * is test infrastructure only.
 */
#ifndef PASCO_HIP_H_
#define PASCO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifdef PH_ORACLE
#define PH_FN(name) pho_##name
#else
#define PH_FN(name) ph_##name
#endif

/* 2: ph_conv_desc grew (split_exp2, out_split, window / axis-table / row-list blocks), ph_map_insert and ph_split_rows gained
 * their `status` argument.  A caller built against another version must be rebuilt: the binding checks the version AND the size
 * of ph_conv_desc before the first call.
 * 3: ph_panop_*.  4: ph_conv_desc.route (was reserved2) replaces the process-global test hooks of rounds 2 - 5; w_frag.
 * 5: ph_conv_desc.grid_dims / grid_kernel (the dense-grid promise). */
#define PH_ABI_VERSION 5
#define PH_MAX_KVOL 64 /* largest kernel volume of one nbr_build / pooling call (4x4x4 window); conv_fwd takes tables of up to 4096 offsets (dense bottleneck: 7x7x5 = 245) */

/* ph_conv_desc.route bits */
#define PH_ROUTE_WIN_ALWAYS 0x1  /* 3x3x3 maps with window tables: the LDS-window kernel, whatever the map's locality */
#define PH_ROUTE_WIN_NEVER 0x2   /* ... the per-offset gather kernel */
#define PH_ROUTE_WIDE_ALWAYS 0x4 /* 128 / 256 output channels, >= 8 offsets: the 256-row tile kernel at any size */
#define PH_ROUTE_WIDE_NEVER 0x8
#define PH_ROUTE_LIN_NEVER 0x10  /* k = 1 products: not the row-stream kernel */
#define PH_ROUTE_GRID_NEVER 0x20 /* dense-grid promise (grid_dims): ignored, the gather kernels read nbr */

/* status word bit 6, written by ph_split_rows only: the operand holds at least one value with |x * 2^exp2| >= PH_SPLIT_FULL_PRECISION,
 * i.e. one whose hi / lo split carries the full 22 bits.  Informational in a stream's status word; in the word of a GUARDED
 * convolution (ph_conv_desc.exact_if) its ABSENCE - a tensor of tiny values only: 2^-19 relative error at |x| = 2^-9 growing to
 * 2^-11 at 2^-19, unscaled with exp2 = 5 - sends the launch to the exact fp32 kernel like an overflow does.  The device library
 * looks for such a value among the first 65 536 eight-channel segments of the tensor only (a miss costs a launch of the exact
 * kernel, never a wrong result); the checker looks at every value. */
#define PH_STATUS_MAGNITUDE 0x40
#define PH_SPLIT_FULL_PRECISION 0.0625f

/* activation codes for fused prologue / epilogue */
#define PH_ACT_NONE 0
#define PH_ACT_RELU 1
#define PH_ACT_LEAKY 2 /* negative slope passed separately */

typedef void *ph_stream_t;

int PH_FN(abi_version)(void);
/* sizeof(ph_conv_desc) as the library was compiled: the second half of the handshake (a binding whose mirror of the struct
 * has another size would hand over misaligned fields). */
int PH_FN(conv_desc_size)(void);
const char *PH_FN(last_error)(void);

/* Scratch bytes any call below may need for `n` rows. */
int64_t PH_FN(workspace_bytes)(int64_t n);

/* ---------------------------------------------------------------------------------------------
 * Coordinate hash map (serves every `ME.SparseTensor(features, coordinates)` constructor:
 * pasco/models/net_panoptic_sparse.py:549, augmenter.py:26, unet3d_sparse_v2.py:207-212,
 * decoder_v3.py:141-146, transformer_predictor_v2.py:203-205,231,258-262, ensembler.py:122).
 *
 * Table = open-addressing, linear probing; `tkeys[cap]` packed 64-bit coordinate keys,
 * `tvals[cap]` row index.  cap must be a power of two >= 2*n.  Duplicated coordinates keep the
 * FIRST occurrence (upstream default quantisation mode); unique rows keep their input order.
 *   row2uniq[n]   out: unique-row index of every input row
 *   uniq_rows[n]  out: input row of unique row j (first n_uniq entries valid)
 *   n_uniq        out: device scalar
 *   status        in/out (or NULL): bit 1 is raised when a coordinate cannot be packed into the 64-bit key (batch index
 *                 outside 0 .. 1023 or a coordinate outside -2^17 .. 2^17 - 1: it would alias another voxel).  Lookups
 *                 (map_find, nbr_build) answer -1 for such coordinates.
 * ------------------------------------------------------------------------------------------- */
int PH_FN(map_insert)(const int32_t *coords, int64_t n, uint64_t *tkeys, int32_t *tvals,
                      int64_t cap, int32_t *row2uniq, int32_t *uniq_rows, int32_t *n_uniq,
                      void *ws, int64_t ws_bytes, int32_t *status, ph_stream_t stream);

/* Look up `n` query coordinates; out_rows[i] = row or -1.  (Union add decoder_v3.py:163,
 * attention-mask lookup transformer_predictor_v2.py:276-279.) */
int PH_FN(map_find)(const int32_t *query, int64_t n, const uint64_t *tkeys, const int32_t *tvals,
                    int64_t cap, int32_t *out_rows, ph_stream_t stream);

/* out = floor(c / ts) * ts on x,y,z (batch kept).  Coordinates of a strided conv / pool output
 * (k=2,s=2 down-convs mink.py:509-511 used at encoder_v2.py:124,133,142; MinkowskiMaxPooling
 * transformer_predictor_v2.py:100-102). */
int PH_FN(coords_floor)(const int32_t *coords, int64_t n, int32_t ts, int32_t *out,
                        ph_stream_t stream);

/* Generative expansion: out[i*8 + k] = c_i + delta_k * ts_out, delta_k in {0,1}^3, x fastest.
 * (`MinkowskiConvolutionTranspose(k=2, stride=2, expand_coordinates=True)`, mink.py:524-527.) */
int PH_FN(coords_expand)(const int32_t *coords, int64_t n, int32_t ts_out, int32_t *out,
                         ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel map.  Output-stationary neighbour table nbr[kvol][n_out]:
 *     nbr[k][o] = row of (c_o + offsets[k]) in the input map, or -1.
 * `h_offsets` is a HOST array [kvol][3] of coordinate offsets already multiplied by the tensor
 * stride (and negated for transposed convs).  Offset enumeration order (x fastest; odd kernels
 * centred, even kernels start at 0) is fixed by the Python frontend.
 * Serves every MinkowskiConvolution / ConvolutionTranspose / MaxPooling with kernel volume > 1.
 * ------------------------------------------------------------------------------------------- */
int PH_FN(nbr_build)(const int32_t *out_coords, int64_t n_out, const uint64_t *in_tkeys,
                     const int32_t *in_tvals, int64_t in_cap, const int32_t *h_offsets,
                     int32_t kvol, int32_t *nbr, ph_stream_t stream);

/* nbr_build for a map onto ITSELF with a symmetric kernel (offset[K-1-k] = -offset[k], K odd; `coords` are the rows the
 * table was built from, unique): same table, half the probes - a hit at offset k also is the entry of the mirrored
 * offset.  Every stride-1 3x3x3 convolution of the U-Net (mink.py:625-638). */
int PH_FN(nbr_build_same)(const int32_t *coords, int64_t n, const uint64_t *tkeys, const int32_t *tvals,
                          int64_t cap, const int32_t *h_offsets, int32_t kvol, int32_t *nbr, ph_stream_t stream);

/* COO kernel map in upstream's form: for every offset k the (in_row, out_row) pairs, sorted by
 * out_row.  pairs_in / pairs_out have kvol*n_out capacity, segment k starts at k*n_out;
 * counts[kvol] is a device array. */
int PH_FN(kmap_compact)(const int32_t *nbr, int32_t kvol, int64_t n_out, int32_t *pairs_in,
                        int32_t *pairs_out, int32_t *counts, void *ws, int64_t ws_bytes,
                        ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse convolution forward (MinkowskiConvolution k=3 mink.py:625-638, decoder_v3.py:267-282;
 * k=2,s=2 mink.py:509-511; transposed k=2,s=2 mink.py:524-527; k=1 encoder_v2.py:109-111,
 * decoder_v3.py:103-105,133-135).
 *
 *   out[o,:] = epi( sum_k [nbr[k][o] >= 0] * pro(in[nbr[k][o], :]) @ W[k] + bias )
 *   pro(x)  = act_pro(x * pro_scale + pro_shift)          (per input channel; BN-eval + ReLU)
 *   epi(y)  = act_res( act_epi(y * epi_scale + epi_shift) * epi2_scale + epi2_shift + residual[o,:] )
 *             (second affine = the extra BatchNorm that follows a down-conv block,
 *              encoder_v2.py:124-126; with residual == NULL act_res is its activation)
 *
 * nbr == NULL means the identity map (kernel volume 1, n_out == n_in).  Any of pro_scale /
 * pro_shift / bias / epi_scale / epi_shift / epi2_scale / epi2_shift / residual may be NULL.  W is [kvol, cin, cout].
 * ------------------------------------------------------------------------------------------- */
typedef struct ph_conv_desc {
  const float *in;       /* [n_in, cin] */
  const float *weight;   /* [kvol, cin, cout] */
  const int32_t *nbr;    /* [kvol, n_out] or NULL */
  float *out;            /* [n_out, cout] */
  int64_t n_in;
  int64_t n_out;
  int32_t cin;
  int32_t cout;
  int32_t kvol;
  int32_t pro_act;       /* PH_ACT_* applied after pro_scale/pro_shift */
  const float *pro_scale; /* [cin] */
  const float *pro_shift; /* [cin] */
  const float *bias;      /* [cout] */
  const float *epi_scale; /* [cout] */
  const float *epi_shift; /* [cout] */
  int32_t epi_act;
  float epi_slope;        /* negative slope for PH_ACT_LEAKY (pro and epi) */
  const float *residual;  /* [n_out, cout] added after epi_act */
  int32_t res_act;        /* activation after the second affine / residual add */
  int32_t split_exp2;     /* modes 1 / 2: the activation operand stands for x * 2^split_exp2 (mode 1 scales the gathered
                             rows before splitting them; mode 2: in_split was made with this exp2; out_split is emitted
                             with it).  The caller folds 2^-split_exp2 into w_unscale.  Keeps the lo halves of small
                             activations out of the f16 subnormal range: |x| * 2^split_exp2 in [2^-3, 65504] has the full
                             22-bit operand; the range flag fires above 65504 * 2^-split_exp2 */
  const float *epi2_scale; /* [cout] second per-channel affine, applied after epi_act */
  const float *epi2_shift; /* [cout] */
  /* matrix-core mode: 0 = fp32 MFMA (exact fp32, default); 1 = opt-in split precision: the product is
   * formed as hi*hi + hi*lo + lo*hi of f16 halves with fp32 accumulation (fp32-class accuracy at ~5x less
   * matrix-pipe time).  Mode 1 needs the weights pre-split, pre-transposed and pre-scaled:
   * w_f16_hi / w_f16_lo = f16 [kvol, cout, cin] of (weight * 2^e); w_unscale = 2^-e; cin % 8 == 0. */
  int32_t mma_mode;
  float w_unscale;
  const void *w_f16_hi;
  const void *w_f16_lo;
  void *splitk_ws;        /* optional scratch for mode 1: few-row layers may split the kernel offsets over
                             several workgroups and reduce [splits, n_out, cout] partial sums in a fixed order */
  int64_t splitk_ws_bytes;
  int32_t *status;        /* optional device word; mode 1 ORs bit 0 into it when a gathered activation
                             exceeds the f16 range (|x| > 65504) - the caller must then redo the layer in mode 0;
                             bit 2 when a coordinate fell outside axis_table's rows (the row was clamped) */
  /* mode 2 = mode 1 with BOTH operands pre-split by ph_split_rows (the gather becomes a 16-byte copy: no
   * per-gather conversion).  in_split = ph_split_rows(in [n_in, cin], pro_*, split_exp2) - the prologue is applied there
   * and NOT again by the device; w_split = ph_split_rows of the [kvol*cout, cin] rows of (weight * 2^e)
   * transposed to [kvol, cout, cin]; w_unscale = 2^-e.  `in` / `weight` / w_f16_* are not read by the device
   * library in mode 2 (the checker build reads in / weight / pro_*).  The range flag is raised by ph_split_rows. */
  const void *in_split;
  const void *w_split;
  /* mode 2, optional second output (cout % 32 == 0): out_split = ph_split_rows(out, osp_scale, osp_shift,
   * osp_act, epi_slope, split_exp2) - the operand of the NEXT convolution, with that convolution's prologue already applied -
   * written by the same launch.  With out_split given, `out` may be NULL (fp32 result not needed). */
  void *out_split;
  const float *osp_scale; /* [cout] */
  const float *osp_shift; /* [cout] */
  int32_t osp_act;
  /* Kernel choice of THIS call, for parity tests (0 = the library decides - every caller of the product path).  Every route computes
   * the same products in another fp32 summation order; the bits only pin which hand-written kernel serves a shape that more than
   * one could serve, so that a test can hold each of them to the oracle (tests/test_hip_win.py, test_hip_wide.py, test_hip_lin.py).
   * No process-global state: two threads may use different routes at once. */
  int32_t route;
  /* mode 2, optional: LDS-window tables of the kernel map (ph_win_build; 3x3x3 maps only).  When all four are given
   * the library may serve the launch from per-tile input windows instead of per-offset gathers; which of the two
   * kernels does the work is decided on the device from win_stats (no host read).  Results are the same products
   * in a different fp32 summation order. */
  const int32_t *win_rows;   /* [ntiles][27*128] */
  const int32_t *win_cnt;    /* [ntiles] */
  const uint16_t *win_slots; /* [ntiles][27][128] */
  const int32_t *win_stats;  /* [4] */
  /* mode 2, optional per-axis TABLE residual (the sine position encoding of the mask transformer,
   * position_encoding.py:90-135 via transformer_predictor_v2.py:143,150, without materialising it): the row
   *     axis_table[0][x - lo] + axis_table[1][y - lo] + axis_table[2][z - lo]      (each [cout])
   * with (x, y, z) = axis_coords[o][1..3] is added where `residual` is added (coordinate values are clamped to the
   * table).  axis_table is fp32 [3][axis_rows][cout]; axis_coords int32 [n_out][4]. */
  const float *axis_table;
  const int32_t *axis_coords;
  int32_t axis_lo;
  int32_t axis_rows;
  /* Row lists of a kernel map in which every output row has exactly ONE (offset, input row) pair - the generative transposed
   * convolutions (mink.py:524-527).  rl_in / rl_out [rl_rows] from ph_rowlist_pack: position i of the list pairs input row
   * rl_in[i] with output row rl_out[i] (-1 = padding); tile t (128 positions) belongs to kernel offset rl_tile_k[t] (-1 =
   * unused tile).  With them (mma_mode 2) the launch is a set of k = 1 products, one per 128-position tile, instead of a walk
   * over all kvol offsets with kvol - 1 empty entries per row; `nbr` still describes the same map (and is what a checker
   * reads).  NULL = not given. */
  const int32_t *rl_in, *rl_out, *rl_tile_k;
  int64_t rl_rows;
  int32_t rl_tiles;       /* entries of rl_tile_k (an upper bound of the used tiles) */
  /* mode 0 only, optional: a device word that GUARDS the launch - it does its work only when (*exact_if & 1) != 0 (an operand
   * left the f16 range) or (*exact_if & PH_STATUS_MAGNITUDE) == 0 (the operand holds tiny values only: round 6) and
   * returns at once otherwise.  The guarded form of the split path, no host read: ph_split_rows and the mode-2 launch get
   * `status` = this word (bit 0 = an operand left the f16 range), then the same convolution is launched in mode 0 with
   * exact_if = this word and the same `out`: the exact fp32 result replaces the split one exactly when it has to.  What
   * the plain MinkowskiConvolution modules of pasco_amd.me do (round 5). */
  const int32_t *exact_if;
  /* mode 2, optional, cout <= 64 and kvol == 27: the rows of w_split once more in FRAGMENT ORDER for the window kernel of the
   * 64-wide outputs (conv_wop.hip): f16 [kvol][cpad / 16][2][2][64][8] - per (offset, 16-channel chunk c, column block j of 32,
   * hi / lo) the 64 lanes' 16 bytes back to back: lane l31 + 32 h holds channels 16 c + 8 h .. + 7 of column min(32 j + l31,
   * cout - 1).  A fragment load then reads 1 KB of whole cache lines instead of 32 bytes of each of 32 rows (round 6: the
   * weight loads were what saturated the vector-memory path of that kernel).  Same values as w_split; NULL = not given. */
  const void *w_frag;
  /* mode 2, optional PROMISE about the map (round 6; all zero = none): `nbr` is the stride-1 kernel map of a (kx, ky, kz) box - odd
   * sizes, dilation 1 - on the FULL dense grid [B, X, Y, Z] (n_in == n_out == B * X * Y * Z) with the sites enumerated (b, z, x, y),
   * y running fastest, and the offsets enumerated k = iy + ky * (ix + kx * iz), offset = (ix - kx / 2, iy - ky / 2, iz - kz / 2).
   * grid_dims = {B, X, Y, Z}, grid_kernel = {kx, ky, kz}.  The library may then form neighbour rows by arithmetic and serve the ky
   * offsets of one (dx, dz) from one LDS window of the input (conv_grid.hip) - the dense bottleneck, whose torch Conv3d stages
   * (layers.py:656-726) such a launch stands for.  `nbr` must still be given and BE that map: the checker build and the other
   * kernels read it, and a promise the map does not keep gives wrong sums without a flag. */
  int32_t grid_dims[4];
  int32_t grid_kernel[3];
  int32_t reserved3;
} ph_conv_desc;

int PH_FN(conv_fwd)(const ph_conv_desc *desc, ph_stream_t stream);

/* Input windows of a 3x3x3 kernel map (stride-1 convolutions: MinkowskiConvolution k=3, mink.py:625-638,
 * decoder_v3.py:267-282), built once per map and cached by the caller next to `nbr`.  For every tile of 128
 * consecutive output rows:
 *   win_rows[tile][0 .. cnt)   the DISTINCT input rows its 27 x 128 neighbour entries name, ascending
 *   win_cnt[tile]              their number (<= 27 * 128)
 *   win_slots[tile][k][r]      position of nbr[k][tile*128 + r] in that list, 0xFFFF for "no neighbour"
 *   win_stats[0 / 1]           sum over tiles of ceil(cnt / 416) and ceil(cnt / 512) (window passes of the two
 *                              kernel shapes; device-side choice between window and gather kernels)
 * Buffers: win_rows int32 [ntiles * 3456], win_cnt int32 [ntiles], win_slots uint16 [ntiles * 3456],
 * win_stats int32 [4]; ntiles = ceil(n_out / 128).  (No upstream counterpart: ME gathers per offset.) */
int PH_FN(win_build)(const int32_t *nbr, int32_t kvol, int64_t n_out, int32_t *win_rows, int32_t *win_cnt,
                     uint16_t *win_slots, int32_t *win_stats, ph_stream_t stream);

/* Diagnostics for the parity tests: which kernel instantiation the LAST conv_fwd of the calling thread launched.
 * h_out8 (HOST array of 8 ints) = { mma_mode, tile rows, tile channels, input channels per stage, splits over the
 * kernel offsets, 1 if the launch wrote out_split, kernel id (0 k_conv_mfma, 1 k_conv_f16x3, 2 k_conv_h2,
 * 3 k_conv_rl = row-list k = 2 kernel, 4 k_conv_dma, 5 k_conv_win + k_conv_dma pair with the device-side choice), waves per workgroup }; all -1 / 0 before the first launch.  The checker build reports
 * kernel id -1.  (No upstream counterpart: ME picks its kernel inside ConvolutionForwardKernelGPU.) */
int PH_FN(conv_last_config)(int32_t *h_out8);

/* Operand preparation for mma_mode 2.  x = act(in * pro_scale + pro_shift) * 2^exp2 (any of them NULL /
 * PH_ACT_NONE; the power of two is exact), hi = f16(x) (round to nearest even), lo = f16(x - hi); out_split is
 * f16 [n][cpad/32][2][32] with cpad = c rounded up to 32: group g holds channels 32g..32g+31 as 32 hi values then
 * 32 lo values; channels >= c are zero.  c % 8 == 0.  ORs bit 0 into *status (optional) when some |x| > 65504 or
 * is NaN.  exp2: activations use the convolution's split_exp2, pre-scaled weights 0. */
int PH_FN(split_rows)(const float *in, int64_t n, int32_t c, const float *pro_scale, const float *pro_shift,
                      int32_t pro_act, float slope, int32_t exp2, void *out_split, int32_t *status, ph_stream_t stream);

/* Local max pooling over a neighbour table (MinkowskiMaxPooling,
 * transformer_predictor_v2.py:100-102,234-236).  Rows without any neighbour give 0. */
int PH_FN(maxpool_fwd)(const float *in, int32_t c, const int32_t *nbr, int32_t kvol,
                       int64_t n_out, float *out, ph_stream_t stream);

/* Sine position encoding of voxel coordinates (PositionEmbeddingSineSparse, normalize=True:
 * pasco/models/transformer/position_encoding.py via transformer_predictor_v2.py:93-95,190-199).
 * coords = int32 rows of `cstride` ints whose x,y,z start at column `coff`; out [n, 3*f] fp32.  Per axis a:
 *   c = float(coord); c = c / (c + 1e-6f) * scale;  ang_i = c / dim_t[i]  (dim_t [f], given by the caller)
 *   out[row, a*f + p]       = sin(ang_{2p})      p < f/2
 *   out[row, a*f + f/2 + p] = cos(ang_{2p+1})
 * f even.  `table` (optional, [tab_n, f] fp32 = this same function evaluated on the coordinate values
 * tab_lo .. tab_lo + tab_n - 1, one axis) turns the evaluation into a lookup for coordinates inside the table
 * range (the encoding of an axis depends on that axis' integer value only); values outside are computed. */
int PH_FN(sine_pe)(const int32_t *coords, int64_t n, int32_t cstride, int32_t coff, int32_t f,
                   const float *dim_t, float scale, const float *table, int32_t tab_lo, int32_t tab_n,
                   float *out, ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Pruning / row movement (ME.MinkowskiPruning decoder_v3.py:159,421,427,432,496-497,
 * misc.py:17,26; SparseTensor.__add__ decoder_v3.py:163).
 * ------------------------------------------------------------------------------------------- */
/* Stable compaction of a byte mask: keep_rows[j] = j-th row with mask != 0. */
int PH_FN(mask_compact)(const uint8_t *mask, int64_t n, int32_t *keep_rows, int32_t *n_keep,
                        void *ws, int64_t ws_bytes, ph_stream_t stream);

/* dst[j,:] = src[rows[j],:]   (rows[j] == -1 -> zeros).  4-byte elements, c per row. */
int PH_FN(gather_rows)(const void *src, int32_t c, const int32_t *rows, int64_t n_out, void *dst,
                       ph_stream_t stream);

/* dst[rows[i],:] += src[i,:] for rows[i] >= 0; rows must be unique (no atomics). */
int PH_FN(scatter_add_rows)(const float *src, int32_t c, const int32_t *rows, int64_t n_src,
                            float *dst, ph_stream_t stream);

/* Keep masks of the decoder in one pass (decoder_v3.py:148-158 completion keep = OR over the subnets' occupied masks;
 * :411-420 panoptic keep = occupied, or the first 1000 rows when nothing is occupied, AND inside the subnet's box):
 *   out[r] = K[r] && lo <= coords[r].xyz <= hi,   K[r] = OR_i kept_i[r]
 * srcs: HOST array of n_src <= 8 device pointers, all of one kind: 0 = bytes (non-zero = kept), 1 = int32 (>= 0 = kept: the
 * rows ph_map_find returns).  lo / hi: device int32[3] or both NULL (no box test; coords may then be NULL).
 * fallback_rows > 0: when K is false everywhere, K[r] = (r < fallback_rows) instead - decided on the device through the
 * scratch word any_word (int32, clobbered).  out: bytes 0 / 1. */
int PH_FN(keep_mask)(const void *const *srcs, int32_t n_src, int32_t kind, const int32_t *coords, int64_t n,
                     const int32_t *lo, const int32_t *hi, int64_t fallback_rows, uint8_t *out, int32_t *any_word,
                     ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Semantic ensembling on the canonical grid in one pass (Ensembler.ensemble_sem_compl + the confidence map of
 * Net.forward: pasco/models/ensembler.py:159-187, net_panoptic_sparse.py:252-310).  For every canonical site s and every
 * subnet i (m <= 8):  p_i = softmax(logits_i[rows_i[s], :])   or the one-hot of class 0 when rows_i[s] < 0
 * (the reference resamples the probabilities densely and sets class 0 to 1 where nothing lands);
 *   out[i][s, :] = p_i,  out[m][s, :] = (p_0 + ... + p_{m-1}) / m,  conf[i][s] = max_c out[i][s, c]   (conf optional).
 * logits_i fp32 [n_i, c]; rows_i int32 [n_sites] (row of the subnet's voxel that the site samples, ph_map_find of the
 * transformed site); out[i] fp32 [n_sites, c] channels-last rows (the returned dense [c, X, Y, Z] tensors are views). */
typedef struct ph_sem_ens_desc {
  int32_t m;
  int32_t c;
  int64_t n_sites;
  const float *logits[8];
  const int32_t *rows[8];
  float *out[9];
  float *conf[9]; /* each may be NULL */
} ph_sem_ens_desc;

int PH_FN(sem_ensemble)(const ph_sem_ens_desc *desc, ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense <-> sparse (SparseTensor.dense augmenter.py:17-18, unet3d_sparse_v2.py:196-198,
 * transformer_predictor_v2.py:263-274; ME.to_sparse augmenter.py:22, unet3d_sparse_v2.py:202,
 * ensembler.py:117).
 * dense layout [B, C, X, Y, Z] fp32.  to_dense does NOT zero `dense` (caller zero-fills).
 * Site of row i: ((c_i - min) / ts) on each axis; rows falling outside are skipped.
 * ------------------------------------------------------------------------------------------- */
int PH_FN(to_dense)(const float *feats, const int32_t *coords, int64_t n, int32_t c,
                    const int32_t *h_min3, int32_t ts, const int32_t *h_dims4 /*B,X,Y,Z*/,
                    float *dense, ph_stream_t stream);

/* to_sparse: rows for sites with any non-zero channel, lexicographic (b,x,y,z) order.
 *   site_rows[B*X*Y*Z] scratch/out: row of each site or -1
 *   out_coords[n_sites,4] (capacity B*X*Y*Z), n_rows device scalar.
 * Features are then read with dense_gather. */
int PH_FN(to_sparse_coords)(const float *dense, int32_t c, const int32_t *h_dims4,
                            int32_t *out_coords, int32_t *n_rows, void *ws, int64_t ws_bytes,
                            ph_stream_t stream);

/* feats[i, ch] = dense[b_i, ch, x_i, y_i, z_i] for coords given in site units. */
int PH_FN(dense_gather)(const float *dense, int32_t c, const int32_t *h_dims4,
                        const int32_t *site_coords, int64_t n, float *feats, ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Masked cross-attention of <= 128 queries over N voxel keys (CrossAttentionLayer of the mask
 * transformer: pasco/models/transformer/blocks.py:73-92, called at
 * transformer_predictor_v2.py:167-173 with the mask of :220-289).
 *   q [B, H, Qn, Dh] (pre-scaled by 1/sqrt(Dh)), k / v [B, N, H*Dh], out [B, Qn, H*Dh]
 *   bits [B, N, 4]: bit q of the 128-bit word = query q may attend that key (NULL = no mask)
 *   any  [B, 4]   : OR of bits over the keys; a query with no allowed key attends everywhere
 *                   (transformer_predictor_v2.py:163-164)
 * attn_mask_pack builds bits / any from vals [B*N, Qn]: allowed = (vals != 0), or (vals > 0) when
 * `positive_only` (mask logits: sigmoid(l) > 0.5 <=> l > 0, transformer_predictor_v2.py:226).
 * bits_orpool ORs the bit rows of the children of every coarse voxel (the reference max-pools the
 * 0/1 mask, transformer_predictor_v2.py:232-236); bits_or_reduce ORs all rows of a batch (`any`).
 * ------------------------------------------------------------------------------------------- */
int64_t PH_FN(attn_workspace_bytes)(int64_t n, int32_t b, int32_t h, int32_t qn, int32_t dh);

int PH_FN(attn_mask_pack)(const float *vals, int64_t n, int32_t b, int32_t qn, int32_t positive_only,
                          uint32_t *bits, uint32_t *any, ph_stream_t stream);

int PH_FN(bits_orpool)(const uint32_t *bits_in, const int32_t *nbr, int32_t kvol, int64_t n_out,
                       uint32_t *bits_out, ph_stream_t stream);

int PH_FN(bits_or_reduce)(const uint32_t *bits, int64_t n, int32_t b, uint32_t *any, ph_stream_t stream);

/* The attention mask of one level without the pooled map: level voxel i (batch i / n_per_b, coordinates level_coords[i][1..3],
 * edge s) gets the OR of the bit rows of the fine voxels inside its block [c, c + s)^3, found through the fine map's hash
 * table (tkeys / tvals of ph_map_insert; bits_in rows in that map's row order).  Equals max-pool (kernel s, stride s) ->
 * dense -> index of transformer_predictor_v2.py:232-289 whenever no coordinate lies outside [lo[b], hi[b]] (int32 [B, 3]
 * each, device): `range` (device word, or NULL) gets bit 0 when one does - the dense indexing of the reference then wraps
 * negative indices, which the caller reproduces through bits_orpool + the dense-site lookup. */
int PH_FN(bits_block_or)(const int32_t *level_coords, int64_t m, int64_t n_per_b, int32_t s, const uint64_t *tkeys,
                         const int32_t *tvals, int64_t cap, const uint32_t *bits_in, const int32_t *lo, const int32_t *hi,
                         uint32_t *bits_out, int32_t *range, ph_stream_t stream);

int PH_FN(attn_cross_fwd)(const float *q, const float *k, const float *v, const uint32_t *bits,
                          const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h,
                          int32_t qn, int32_t dh, void *ws, int64_t ws_bytes, ph_stream_t stream);

/* attn_cross_fwd with K and V as split f16 operands [B*N, H*Dh/32, 2, 32] (hi + lo = value * 2^exp2: what a projection
 * launched with ph_conv_desc.out_split writes, so K and V never exist in fp32): every product runs as three f16 MFMAs
 * with fp32 accumulation.  `status` (or NULL): bit 0 is raised when a query value leaves the f16 range after the
 * kernel's own 2^8 scaling (the caller then reruns attn_cross_fwd on fp32 K / V). */
int PH_FN(attn_cross_split)(const float *q, const void *k_split, const void *v_split, int32_t exp2,
                            const uint32_t *bits, const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h,
                            int32_t qn, int32_t dh, void *ws, int64_t ws_bytes, int32_t *status, ph_stream_t stream);

/* The same attention straight on a level's FEATURE operand: the keys and values of a cross-attention level are
 * K = x A + a + pos W_k^T and V = x B + b + pos W_v^T (input projection and K / V projection composed; reference
 * transformer_predictor_v2.py:150,167-173 + transformer/blocks.py:83-86, key = value = bb_feat + pos), so per head
 * q_h K_h^T = (q_h A_h^T) x^T + const + position terms and P V_h = (P x) B_h + b_h + P (position terms): both products run
 * on the rows [x | aug] and K / V are never formed (at the finest level of the benchmark scene: 2 x 970 MB less written
 * and read, two projection launches less).
 *   x_split [B*N, c/32, 2, 32] f16: the level's features as a split operand (value * 2^exp2; ph_split_rows)
 *   aug     [B*N, 16] f16          : position columns of every key (pos_aug below), value * 2^exp2 like x_split
 *   q2      [B, H, Qn, c + 16]     : per head q_h [A_h^T | position coefficients] (host), 1/sqrt(Dh) included
 *   out     [B, Qn, H * (c + 16)]  : Y = softmax(q2 [x | aug]^T + mask) [x | aug]; the caller applies (B_h ; position rows)
 *                                    and the constants (folded into the output projection)
 * The HIP library serves c = 64; bits / any / status as attn_cross_split; workspace = attn_workspace_bytes(.., dh = c + 16).
 *
 * pos_aug: aug[i] = ([c_x == 0], [c_y == 0], [c_z == 0], eps[c_x - tab_lo], eps[c_y - tab_lo], eps[c_z - tab_lo], 0 x 10)
 * for coords [N, 4] (b, x, y, z).  The sine encoding normalises c / (c + 1e-6) * 2 pi (position_encoding.py:100-104): the
 * angle is 0 for c = 0, exactly 2 pi for |c| >= 32 and 2 pi + eps_c in between, which is all a key's position term depends
 * on (eps [tab_n] fp32 in the caller's scale, eps of the value 0 = 0).  Every column is written times 2^exp2 (the feature
 * operand's scale: the Q2 position columns then need no extra factor and meet the f16 range flag only above 256, like the
 * feature columns).  A coordinate outside the table raises status bit 2. */
int PH_FN(attn_cross_feat)(const float *q2, const void *x_split, const void *aug, int32_t c, int32_t exp2,
                           const uint32_t *bits, const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h,
                           int32_t qn, void *ws, int64_t ws_bytes, int32_t *status, ph_stream_t stream);

int PH_FN(pos_aug)(const int32_t *coords, int64_t n, const float *eps, int32_t tab_lo, int32_t tab_n, int32_t exp2, void *aug,
                   int32_t *status, ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Panoptic ensembling on the compacted union of occupied canonical sites (reference: pasco/models/ensembler.py:20-131,
 * which works on dense [100, 256, 256, 32] tensors per subnet).  sel [U] = canonical site id of every union row.
 *   ens_resample: out[u, :] = sigmoid(logits[rows[sel[u]], :]) (rows [n_sites]: the subnet's voxel row of a canonical
 *                 site or -1 -> zeros) = sigmoid + grid_sample(nearest, zero padding) of ensembler.py:44-62;
 *                 flag[u] = the row has a non-zero entry (what ME.to_sparse keeps, misc.py:46-57)
 *   ens_merge   : anchor = (anchor * i + m[:, perm]) / (i + 1), the running mean of the Hungarian-matched query masks
 *                 (ensembler.py:86-98), in place, fp32 operations in that order
 *   ens_finish  : out[u, j] = anchor[u, keep[j]] * (argmax_c sem[sel[u], c] != 0) (matched-IoU filter of the queries and
 *                 the "empty class" zeroing, ensembler.py:100-118); flag as above.  sem [n_sites, c] channels last.
 * ------------------------------------------------------------------------------------------- */
int PH_FN(ens_resample)(const float *logits, int64_t n, int32_t q, const int32_t *rows, const int32_t *sel, int64_t u,
                        float *out, uint8_t *flag, ph_stream_t stream);
int PH_FN(ens_merge)(float *anchor, const float *m, const int32_t *perm, int64_t u, int32_t q, int32_t i,
                     ph_stream_t stream);
int PH_FN(ens_finish)(const float *anchor, int64_t u, int32_t q, const int32_t *keep, int32_t qk, const float *sem,
                      int32_t c, const int32_t *sel, float *out, uint8_t *flag, ph_stream_t stream);

/* Canonical grid seen through a subnet's transform: for every site (x, y, z) of the X x Y x Z grid (lexicographic site id)
 * the voxel index of T applied to its centre, as (0, x', y', z') rows: the reference's `transform` + `sample_scene`
 * addressing (transform_utils.py:60-74,95-117): centre = site * resolution + resolution / 2 + min_bound in float64, cast to
 * fp32, T (device, 3 x 4 leading rows of the 4 x 4, row-major) in fp32 with the sums in the order ((T0 x + T1 y) + T2 z) + T3,
 * then (v - min_bound - resolution / 2) / resolution rounded half to even.  h_min_bound: 3 host floats. */
int PH_FN(project_canonical)(const float *T, int32_t X, int32_t Y, int32_t Z, double resolution, const float *h_min_bound,
                             int32_t *out_coords, ph_stream_t stream);

/* COO kernel map (ph_kmap_compact) -> the padded row lists of ph_conv_desc.rl_*: cap = rl_rows (multiple of 128, >= n_out +
 * 127 * kvol rounded up), tcap = entries of tile_k (>= cap / 128).  Counts stay on the device; `status` (may be NULL): bit 5
 * (32) is raised when the map does not hold exactly one pair per output row (sum of counts != n_out). */
int PH_FN(rowlist_pack)(const int32_t *pairs_in, const int32_t *pairs_out, const int32_t *counts, int32_t kvol, int64_t n_out,
                        int32_t *rl_in, int32_t *rl_out, int32_t *tile_k, int64_t cap, int64_t tcap, int32_t *status,
                        ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input stage: voxel max of the point features + MIMO channel concatenation (CylinderFeat.forward after its PPmodel,
 * pasco/models/unet3d_sparse_v2.py:53-86: torch.unique(dim=0) of the (subnet, voxel) rows + torch_scatter.scatter_max; then
 * Augmenter.merge, pasco/models/augmenter.py:13-27: SparseTensor.dense -> channel cat over the subnets -> ME.to_sparse).
 * Restated without the sort and without the dense [M, C, X, Y, Z] tensor: the merged rows are the occupied sites of the
 * points' bounding box in lexicographic (x, y, z) order, row r holds in channels [b*C, (b+1)*C) the max over subnet b's
 * points in that voxel, 0 where subnet b has none.
 *   points_bounds      out6 (device) = min x, y, z, max x, y, z over xyz int64 [n, 3] (the reference's index dtype)
 *   points_mark        flags[site] = 1 for the site ((x-lo)*dimy + (y-lo))*dimz + (z-lo) of every point; flags is zero-filled
 *                      by the caller, h_lo3 / h_dims3 are HOST ints; a point outside the box raises status bit 3
 *   mask_compact_rank  mask_compact that also writes rank_of[i] = row of kept element i (rows of the occupied sites)
 *   points_link        chains the points of every (merged row, subnet) cell: next[i] = previous head of cell
 *                      rank_of[site(i)] * m + b(i), head[cell] = i (head int32 [v * m] filled with -1 by the caller, next
 *                      int32 [n]); subnet b of point i from h_starts (HOST int64 [m + 1]: first point of every subnet - the
 *                      points of a subnet are contiguous); m <= 8.  The order inside a chain is arbitrary (it feeds a max).
 *   cells_max          out[r, b*c + ch] = max over the chain of cell (r, b) of h[i, ch] (h fp32 [n, c] = the point MLP's
 *                      output), 0 for an empty chain; coords[r] = (0, x, y, z) of sites[r] (int32 [v]: the kept site ids);
 *                      status bit 3 when a row is entirely zero (ME.to_sparse drops such a row: the caller must redo
 *                      the stage on its general path - the row count cannot change without a host read); c % 4 == 0
 * ------------------------------------------------------------------------------------------- */
int PH_FN(points_bounds)(const int64_t *xyz, int64_t n, int32_t *out6, ph_stream_t stream);
int PH_FN(points_mark)(const int64_t *xyz, int64_t n, const int32_t *h_lo3, const int32_t *h_dims3, uint8_t *flags,
                       int32_t *status, ph_stream_t stream);
int PH_FN(mask_compact_rank)(const uint8_t *mask, int64_t n, int32_t *keep_rows, int32_t *rank_of, int32_t *n_keep,
                             void *ws, int64_t ws_bytes, ph_stream_t stream);
int PH_FN(points_link)(const int64_t *xyz, int64_t n, const int64_t *h_starts, int32_t m, const int32_t *h_lo3,
                       const int32_t *h_dims3, const int32_t *rank_of, int32_t *head, int32_t *next, ph_stream_t stream);
int PH_FN(cells_max)(const float *h, int32_t c, const int32_t *head, const int32_t *next, int64_t v, int32_t m,
                     const int32_t *sites, const int32_t *h_lo3, const int32_t *h_dims3, float *out, int32_t *coords,
                     int32_t *status, ph_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Panoptic post-processing (reference: pasco/models/helper.py:91-303 `panoptic_inference`, called for each of the M + 1
 * outputs by Net.step_inference, net_panoptic_sparse.py:578-608).  All tables are int32 [*, 128] rows (q <= 128 queries).
 *   panop_queries  qp fp32 [q, c1] class probabilities (c1 = classes + dustbin, c1 <= 64) -> qtab int32 [4, 128]:
 *                  row 0 kidx[q'] = rank of query q' among the KEPT queries (arg-max class != 0, != dustbin, probability >
 *                  thr: helper.py:135-140; ascending query order) or -1, row 1 kq[k] = query of kept k, row 2 the arg-max
 *                  class of every query (first maximum), row 3 its probability (float bits); nk[0] = number kept
 *   panop_argmax   masks fp32 [n, q] mask probabilities -> per voxel: winner = kept index with the largest prob x mask
 *                  (smallest index on ties; -1 when nothing is kept), own = mask of the winner >= occ_thr, conf = m_w /
 *                  (sum_k m_k + 1e-8), vunc = max_k (p_k m_k) / sum_k (p_k m_k) (helper.py:150-153, 186, 199-206, 235-241);
 *                  areas int32 [2, 128] += (voxels won and owned by kept k, voxels with m_k >= occ_thr); caller zeroes
 *   panop_write    replays the reference's sequential walk over the kept queries (helper.py:188-250: mask_area /
 *                  original_area < overlap_thr skips, `thing_mask` bit c = class c is a thing, stuff segments of one class
 *                  merge and then write the panoptic id only) and writes per voxel panoptic id / semantic class /
 *                  ins_unc (query probability) / vox_conf / vox_unc; seg int32 [5, 128] (may be NULL): rows id, isthing,
 *                  category, query id of every segment, seg[4][0] = number of segments.  No host read anywhere.
 * ------------------------------------------------------------------------------------------- */
int PH_FN(panop_queries)(const float *qp, int32_t q, int32_t c1, float thr, int32_t *qtab, int32_t *nk, ph_stream_t stream);
int PH_FN(panop_argmax)(const float *masks, int64_t n, int32_t q, const int32_t *qtab, float occ_thr, int32_t *winner,
                        uint8_t *own, float *conf, float *vunc, int32_t *areas, ph_stream_t stream);
int PH_FN(panop_write)(int64_t n, const int32_t *winner, const uint8_t *own, const float *conf, const float *vunc,
                       const int32_t *areas, const int32_t *qtab, const int32_t *nk, double overlap_thr, uint64_t thing_mask,
                       int32_t *panoptic, int32_t *semantic, float *ins_unc, float *vox_conf, float *vox_unc, int32_t *seg,
                       ph_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PASCO_HIP_H_ */
