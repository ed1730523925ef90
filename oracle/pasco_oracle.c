/*
 * pasco_oracle.c -- CPU restatement of the sparse-voxel operator semantics on PaSCo's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pasco_amd/ imports, links or executes this file; it is
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
 *
 * PARITY UNPINNED at the operator layer: the arithmetic of this path lives in MinkowskiEngine
 * v0.5.4 (pinned by the reference at README.md:90, imported e.g. at
 * pasco/models/unet3d_sparse_v2.py:6), which is neither vendored under /root/reference nor
 * installed, and the reference holds no tests or golden vectors for it (SURVEY.md section 4 and
 * 8(c)).  This file restates upstream's published semantics as used by the reference's call
 * sites; tests/test_oracle_dense.py pins it against an independent dense formulation
 * (torch conv3d / conv_transpose3d / max_pool3d on zero-filled grids).
 *
 * Semantics restated (one function each):
 *   a1  coordinate map: int32 (b,x,y,z) rows, duplicates keep the first occurrence, unique rows
 *       keep input order                      -> pho_map_insert   (net_panoptic_sparse.py:549)
 *   a3  strided output coordinates floor(c/ts)*ts   -> pho_coords_floor (mink.py:509-511)
 *   a5  generative expansion c + {0,1}^3 * ts_out   -> pho_coords_expand (mink.py:524-527)
 *   a2-a5 kernel map: offsets enumerated x fastest, odd kernels centred, even kernels from 0;
 *       correlation (no flip); W[k] is [cin, cout]  -> pho_nbr_build / pho_kmap_compact / pho_conv_fwd
 *       (mink.py:625-638, decoder_v3.py:267-282, encoder_v2.py:109-111)
 *   a6  eval BatchNorm / ReLU / LeakyReLU as per-channel affine + activation (conv prologue/epilogue)
 *   a7  pruning keeps row order                      -> pho_mask_compact + pho_gather_rows
 *   a8  union add                                    -> pho_map_insert + pho_scatter_add_rows
 *   a9  dense / to_sparse (lexicographic b,x,y,z)     -> pho_to_dense / pho_to_sparse_coords / pho_dense_gather
 *   a10 local max pooling                            -> pho_maxpool_fwd (transformer_predictor_v2.py:100-102)
 *   a12 voxel max of the point features, a13 MIMO merge -> pho_points_* + pho_cells_max
 *       (unet3d_sparse_v2.py:53-86, augmenter.py:13-27)
 *
 * Same C ABI as include/pasco_hip.h with the `pho_` prefix; pointers are host pointers and the
 * stream argument is ignored.  Plain C + OpenMP.
 */
#define PH_ORACLE 1
#include "../include/pasco_hip.h"

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];

static int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

int pho_abi_version(void) { return PH_ABI_VERSION; }
int pho_conv_desc_size(void) { return (int)sizeof(ph_conv_desc); }

/* the restatement has one formulation per operator: nothing to report (kernel id -1) */
int pho_conv_last_config(int32_t *h_out8) {
  if (!h_out8) return 1;
  for (int i = 0; i < 8; ++i) h_out8[i] = (i == 6) ? -1 : 0;
  return 0;
}
const char *pho_last_error(void) { return g_err; }
int64_t pho_workspace_bytes(int64_t n) { return 9 * (n < 0 ? 0 : n) + 4096; }

/* ---- coordinate key (same packing as the device table so tables are interchangeable) -------- */
#define EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define COORD_BIAS (1 << 17)

static uint64_t pack(int b, int x, int y, int z) {
  return ((uint64_t)((uint32_t)b & 0x3FFu) << 54) |
         ((uint64_t)((uint32_t)(x + COORD_BIAS) & 0x3FFFFu) << 36) |
         ((uint64_t)((uint32_t)(y + COORD_BIAS) & 0x3FFFFu) << 18) |
         ((uint64_t)((uint32_t)(z + COORD_BIAS) & 0x3FFFFu));
}

/* batch 0 .. 1023, coordinates -2^17 .. 2^17 - 1: what the key can hold */
static int packable(int b, int x, int y, int z) {
  return (unsigned)b < 1024u && (unsigned)(x + COORD_BIAS) < (1u << 18) && (unsigned)(y + COORD_BIAS) < (1u << 18) &&
         (unsigned)(z + COORD_BIAS) < (1u << 18);
}

static uint64_t mix(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

static int is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

static int find(const uint64_t *tkeys, const int32_t *tvals, uint64_t mask, uint64_t key) {
  uint64_t slot = mix(key) & mask;
  for (;;) {
    uint64_t k = tkeys[slot];
    if (k == key) return tvals[slot];
    if (k == EMPTY_KEY) return -1;
    slot = (slot + 1) & mask;
  }
}

/* a1: sequential insert => first occurrence wins, unique rows numbered in input order. */
int pho_map_insert(const int32_t *coords, int64_t n, uint64_t *tkeys, int32_t *tvals, int64_t cap,
                   int32_t *row2uniq, int32_t *uniq_rows, int32_t *n_uniq, void *ws,
                   int64_t ws_bytes, int32_t *status, ph_stream_t stream) {
  (void)ws; (void)ws_bytes; (void)stream;
  if (!is_pow2(cap) || cap < 2 * n || cap < 2) return fail("map_insert: cap must be pow2 >= 2n");
  for (int64_t i = 0; i < n && status; ++i)
    if (!packable(coords[4 * i], coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3])) *status |= 2;
  for (int64_t s = 0; s < cap; ++s) { tkeys[s] = EMPTY_KEY; tvals[s] = -1; }   /* unset slots: all ones, as in libpascohip */
  uint64_t mask = (uint64_t)cap - 1;
  int32_t count = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t *c = coords + 4 * i;
    uint64_t key = pack(c[0], c[1], c[2], c[3]);
    uint64_t slot = mix(key) & mask;
    while (tkeys[slot] != EMPTY_KEY && tkeys[slot] != key) slot = (slot + 1) & mask;
    if (tkeys[slot] == EMPTY_KEY) {
      tkeys[slot] = key;
      tvals[slot] = uniq_rows ? count : (int32_t)i;
      if (uniq_rows) uniq_rows[count] = (int32_t)i;
      ++count;
    } else if (!uniq_rows) {
      return fail("map_insert: duplicate coordinate at row %lld but caller promised unique", (long long)i);
    }
    if (row2uniq) row2uniq[i] = tvals[slot];
  }
  if (n_uniq) *n_uniq = count;
  return 0;
}

int pho_map_find(const int32_t *query, int64_t n, const uint64_t *tkeys, const int32_t *tvals,
                 int64_t cap, int32_t *out_rows, ph_stream_t stream) {
  (void)stream;
  if (!is_pow2(cap)) return fail("map_find: cap must be pow2");
  uint64_t mask = (uint64_t)cap - 1;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t *c = query + 4 * i;
    out_rows[i] = packable(c[0], c[1], c[2], c[3]) ? find(tkeys, tvals, mask, pack(c[0], c[1], c[2], c[3])) : -1;
  }
  return 0;
}

static int floor_div(int v, int ts) {
  int q = v / ts;
  if ((v % ts) != 0 && ((v < 0) != (ts < 0))) --q;
  return q;
}

/* a3 */
int pho_coords_floor(const int32_t *coords, int64_t n, int32_t ts, int32_t *out, ph_stream_t stream) {
  (void)stream;
  if (ts <= 0) return fail("coords_floor: ts must be > 0");
  for (int64_t i = 0; i < n; ++i) {
    out[4 * i] = coords[4 * i];
    for (int a = 1; a < 4; ++a) out[4 * i + a] = floor_div(coords[4 * i + a], ts) * ts;
  }
  return 0;
}

/* a5 */
int pho_coords_expand(const int32_t *coords, int64_t n, int32_t ts_out, int32_t *out, ph_stream_t stream) {
  (void)stream;
  if (ts_out <= 0) return fail("coords_expand: ts_out must be > 0");
  for (int64_t i = 0; i < n; ++i)
    for (int k = 0; k < 8; ++k) {
      int32_t *o = out + 4 * (i * 8 + k);
      o[0] = coords[4 * i];
      o[1] = coords[4 * i + 1] + (k & 1) * ts_out;
      o[2] = coords[4 * i + 2] + ((k >> 1) & 1) * ts_out;
      o[3] = coords[4 * i + 3] + ((k >> 2) & 1) * ts_out;
    }
  return 0;
}

int pho_nbr_build(const int32_t *out_coords, int64_t n_out, const uint64_t *in_tkeys,
                  const int32_t *in_tvals, int64_t in_cap, const int32_t *h_offsets, int32_t kvol,
                  int32_t *nbr, ph_stream_t stream) {
  (void)stream;
  if (kvol < 1 || kvol > PH_MAX_KVOL) return fail("nbr_build: kvol out of range");
  if (!is_pow2(in_cap)) return fail("nbr_build: cap must be pow2");
  uint64_t mask = (uint64_t)in_cap - 1;
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < n_out; ++o) {
    const int32_t *c = out_coords + 4 * o;
    for (int k = 0; k < kvol; ++k) {
      const int32_t *d = h_offsets + 3 * k;
      nbr[(int64_t)k * n_out + o] = packable(c[0], c[1] + d[0], c[2] + d[1], c[3] + d[2])
                                        ? find(in_tkeys, in_tvals, mask, pack(c[0], c[1] + d[0], c[2] + d[1], c[3] + d[2])) : -1;
    }
  }
  return 0;
}

int pho_kmap_compact(const int32_t *nbr, int32_t kvol, int64_t n_out, int32_t *pairs_in,
                     int32_t *pairs_out, int32_t *counts, void *ws, int64_t ws_bytes,
                     ph_stream_t stream) {
  (void)ws; (void)ws_bytes; (void)stream;
  for (int k = 0; k < kvol; ++k) {
    int64_t base = (int64_t)k * n_out;
    int32_t cnt = 0;
    for (int64_t o = 0; o < n_out; ++o)
      if (nbr[base + o] >= 0) {
        pairs_in[base + cnt] = nbr[base + o];
        pairs_out[base + cnt] = (int32_t)o;
        ++cnt;
      }
    counts[k] = cnt;
  }
  return 0;
}

static float act_apply(float v, int act, float slope) {
  if (act == PH_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == PH_ACT_LEAKY) return v > 0.f ? v : v * slope;
  return v;
}

/* ---- operand split for mma_mode 2 (include/pasco_hip.h ph_split_rows) ------------------------------------
 * IEEE binary16 conversion restated in integer arithmetic (round to nearest even, overflow -> inf,
 * gradual underflow), so the checker does not depend on compiler _Float16 support. */
static uint16_t f32_to_f16_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x0200u : 0u));   /* inf / nan */
  if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);   /* >= 65520 rounds to inf */
  if (ax < 0x33000001u) return (uint16_t)sign;                /* <= 2^-25 rounds to zero */
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;                  /* 24-bit significand */
  int shift;                                                  /* bits dropped from m */
  uint32_t base;
  if (e >= -14) {                                             /* normal half */
    shift = 13;
    base = (uint32_t)(e + 15) << 10;
    m &= 0x7FFFFFu;
  } else {                                                    /* subnormal half: value = m * 2^(e-23), unit 2^-24 */
    shift = 13 + (-14 - e);
    base = 0;
  }
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u);
  const uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;           /* may carry into the exponent: still correct */
  return (uint16_t)(sign | (base + q));
}

static float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const int e = (h >> 10) & 0x1F;
  const uint32_t m = h & 0x3FFu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      float v = (float)m * 5.9604644775390625e-08f;           /* m * 2^-24 */
      memcpy(&x, &v, 4);
      x |= sign;
    }
  } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
  else x = sign | ((uint32_t)(e - 15 + 127) << 23) | (m << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
}

int pho_split_rows(const float *in, int64_t n, int32_t c, const float *pro_scale, const float *pro_shift,
                   int32_t pro_act, float slope, int32_t exp2, void *out_split, int32_t *status, ph_stream_t stream) {
  (void)stream;
  const float pow2 = ldexpf(1.f, exp2);
  if (n < 0 || c <= 0 || c % 8 != 0) return fail("split_rows: needs c % 8 == 0");
  if (n == 0) return 0;
  if (!in || !out_split) return fail("split_rows: null buffer");
  const int cpad = (c + 31) / 32 * 32;
  uint16_t *out = (uint16_t *)out_split;
  const int has_pro = pro_scale || pro_shift || pro_act != PH_ACT_NONE;
  int bad = 0, big = 0;
#pragma omp parallel for schedule(static) reduction(| : bad, big)
  for (int64_t r = 0; r < n; ++r) {
    uint16_t *row = out + r * 2 * cpad;
    for (int ch = 0; ch < cpad; ++ch) {
      uint16_t hi = 0, lo = 0;
      if (ch < c) {
        float v = in[r * c + ch];
        if (has_pro) {
          v = v * (pro_scale ? pro_scale[ch] : 1.f) + (pro_shift ? pro_shift[ch] : 0.f);
          v = act_apply(v, pro_act, slope);
        }
        v *= pow2;
        if (!(fabsf(v) <= 65504.f)) bad = 1;
        if (fabsf(v) >= PH_SPLIT_FULL_PRECISION) big = 1;
        hi = f32_to_f16_bits(v);
        lo = f32_to_f16_bits(v - f16_bits_to_f32(hi));
      }
      row[(ch >> 5) * 64 + (ch & 31)] = hi;
      row[(ch >> 5) * 64 + 32 + (ch & 31)] = lo;
    }
  }
  if (bad && status) *status |= 1;
  if (big && status) *status |= PH_STATUS_MAGNITUDE;      /* include/pasco_hip.h: the operand holds a full-precision value */
  return 0;
}

/* a2-a5: per output row, per kernel offset, row-vector x W[k] accumulate (fp32, like upstream's
 * CPU gather -> SGEMM -> scatter-add). Rows are processed in blocks so W[k] stays cache resident. */
/* rows of a split operand (ph_split_rows layout) back to fp32: hi + lo */
static float *unsplit_rows(const void *split, int64_t n, int c, float scale) {
  const int cpad = (c + 31) / 32 * 32;
  const uint16_t *sp = (const uint16_t *)split;
  float *x = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * c);
  if (!x) return NULL;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const uint16_t *row = sp + r * 2 * cpad;
    for (int ch = 0; ch < c; ++ch)
      x[r * c + ch] = (f16_bits_to_f32(row[(ch >> 5) * 64 + (ch & 31)]) +
                       f16_bits_to_f32(row[(ch >> 5) * 64 + 32 + (ch & 31)])) * scale;
  }
  return x;
}

int pho_conv_fwd(const ph_conv_desc *d, ph_stream_t stream) {
  (void)stream;
  if (!d) return fail("conv_fwd: null desc");
  if (d->cin <= 0 || d->cout <= 0 || d->kvol < 1 || d->kvol > 4096) return fail("conv_fwd: bad shape");
  if (d->n_out == 0) return 0;
  if (d->mma_mode == 0 && d->exact_if && (*d->exact_if & 1) == 0 && (*d->exact_if & PH_STATUS_MAGNITUDE) != 0)
    return 0;   /* guarded launch: nothing to redo (no overflow, and the operand was not all tiny) */
  /* mma_mode 0 / 1: plain fp32 on in / weight (how the device forms the products does not change the values
   * beyond rounding).  mma_mode 2 follows the device's data flow (include/pasco_hip.h): the operands are
   * in_split / w_split, read back as hi + lo, with the prologue already inside in_split; in / weight may be NULL. */
  if (!d->nbr && !(d->kvol == 1 && d->n_in == d->n_out)) return fail("conv_fwd: identity map needs kvol == 1");
  const int cin = d->cin, cout = d->cout;
  const int64_t n_out = d->n_out;
  float *in_tmp = NULL, *w_tmp = NULL, *out_tmp = NULL;
  const float *in = d->in, *weight = d->weight;
  int has_pro = d->pro_scale || d->pro_shift || d->pro_act != PH_ACT_NONE;
  if (d->mma_mode == 2 && d->in_split) {
    in = in_tmp = unsplit_rows(d->in_split, d->n_in, cin, 1.f);   /* = x * 2^split_exp2; w_unscale undoes it */
    has_pro = 0;
  }
  if (!in) return fail("conv_fwd: null input");
  if (d->mma_mode == 2 && d->w_split) {
    /* w_split rows are [kvol * cout][cin] of weight * 2^e; back to [kvol][cin][cout] * 2^-e */
    float *rows = unsplit_rows(d->w_split, (int64_t)d->kvol * cout, cin, d->w_unscale);
    w_tmp = (float *)malloc(sizeof(float) * (size_t)d->kvol * cin * cout);
    for (int64_t k = 0; k < d->kvol; ++k)
      for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c) w_tmp[(k * cin + c) * cout + n] = rows[(k * cout + n) * cin + c];
    free(rows);
    weight = w_tmp;
  }
  if (!weight) { free(in_tmp); return fail("conv_fwd: null weight"); }
  float *outp = d->out;
  if (!outp) {
    if (!d->out_split) { free(in_tmp); free(w_tmp); return fail("conv_fwd: no output buffer"); }
    outp = out_tmp = (float *)malloc(sizeof(float) * (size_t)n_out * cout);
  }
  enum { RB = 32 };
  const int64_t nblocks = (n_out + RB - 1) / RB;
#pragma omp parallel
  {
    float *acc = (float *)malloc(sizeof(float) * RB * cout);
    float *arow = (float *)malloc(sizeof(float) * cin);
#pragma omp for schedule(dynamic, 4)
    for (int64_t blk = 0; blk < nblocks; ++blk) {
      const int64_t o0 = blk * RB;
      const int rows = (int)((n_out - o0) < RB ? (n_out - o0) : RB);
      memset(acc, 0, sizeof(float) * RB * cout);
      for (int k = 0; k < d->kvol; ++k) {
        const float *wk = weight + (int64_t)k * cin * cout;
        for (int r = 0; r < rows; ++r) {
          const int64_t o = o0 + r;
          const int idx = d->nbr ? d->nbr[(int64_t)k * n_out + o] : (int)o;
          if (idx < 0) continue;
          const float *src = in + (int64_t)idx * cin;
          const float *a = src;
          if (has_pro) {
            for (int c = 0; c < cin; ++c) {
              float v = src[c] * (d->pro_scale ? d->pro_scale[c] : 1.f) + (d->pro_shift ? d->pro_shift[c] : 0.f);
              arow[c] = act_apply(v, d->pro_act, d->epi_slope);
            }
            a = arow;
          }
          float *out = acc + (int64_t)r * cout;
          for (int c = 0; c < cin; ++c) {
            const float av = a[c];
            const float *w = wk + (int64_t)c * cout;
            for (int n = 0; n < cout; ++n) out[n] += av * w[n];
          }
        }
      }
      for (int r = 0; r < rows; ++r) {
        const int64_t o = o0 + r;
        for (int n = 0; n < cout; ++n) {
          float v = acc[(int64_t)r * cout + n] + (d->bias ? d->bias[n] : 0.f);
          v = v * (d->epi_scale ? d->epi_scale[n] : 1.f) + (d->epi_shift ? d->epi_shift[n] : 0.f);
          v = act_apply(v, d->epi_act, d->epi_slope);
          if (d->residual || d->epi2_scale || d->epi2_shift || d->res_act != PH_ACT_NONE || d->axis_table) {
            v = v * (d->epi2_scale ? d->epi2_scale[n] : 1.f) + (d->epi2_shift ? d->epi2_shift[n] : 0.f);
            float r = d->residual ? d->residual[o * cout + n] : 0.f;
            if (d->axis_table) {   /* per-axis table residual: t0[x] + t1[y] + t2[z], then the dense residual */
              float t = 0.f;
              for (int ax = 0; ax < 3; ++ax) {
                int idx = d->axis_coords[o * 4 + 1 + ax] - d->axis_lo;
                if ((idx < 0 || idx >= d->axis_rows) && d->status) {   /* clamped: status bit 2 */
#pragma omp atomic
                  *(int32_t *)d->status |= 4;
                }
                idx = idx < 0 ? 0 : (idx >= d->axis_rows ? d->axis_rows - 1 : idx);
                t += d->axis_table[((int64_t)ax * d->axis_rows + idx) * cout + n];
              }
              r += t;
            }
            v += r;
            v = act_apply(v, d->res_act, d->epi_slope);
          }
          outp[o * cout + n] = v;
        }
      }
    }
    free(acc);
    free(arow);
  }
  int rc = 0;
  if (d->out_split) {   /* second output of mma_mode 2: the next convolution's operand */
    if (d->cout % 32 != 0) rc = fail("conv_fwd: out_split needs cout % 32 == 0");
    else rc = pho_split_rows(outp, d->n_out, d->cout, d->osp_scale, d->osp_shift, d->osp_act, d->epi_slope,
                             d->split_exp2, d->out_split, d->status, stream);
  }
  free(in_tmp);
  free(w_tmp);
  free(out_tmp);
  return rc;
}

/* input windows of a 3x3x3 kernel map (include/pasco_hip.h ph_win_build): per 128-row tile the ascending list of
 * distinct input rows and the position of every (offset, row) entry in it */
static int cmp_i32(const void *a, const void *b) {
  const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

int pho_win_build(const int32_t *nbr, int32_t kvol, int64_t n_out, int32_t *win_rows, int32_t *win_cnt,
                  uint16_t *win_slots, int32_t *win_stats, ph_stream_t stream) {
  (void)stream;
  enum { BM = 128, KV = 27, CAP = BM * KV };
  if (kvol != KV) return fail("win_build: serves kvol 27");
  if (n_out == 0) return 0;
  if (!nbr || !win_rows || !win_cnt || !win_slots || !win_stats) return fail("win_build: null buffer");
  const int64_t ntiles = (n_out + BM - 1) / BM;
  long long pa = 0, pb = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : pa, pb)
  for (int64_t t = 0; t < ntiles; ++t) {
    int32_t buf[CAP];
    int m = 0;
    for (int k = 0; k < KV; ++k)
      for (int r = 0; r < BM; ++r) {
        const int64_t row = t * BM + r;
        const int32_t idx = row < n_out ? nbr[(int64_t)k * n_out + row] : -1;
        if (idx >= 0) buf[m++] = idx;
      }
    qsort(buf, (size_t)m, sizeof(int32_t), cmp_i32);
    int cnt = 0;
    for (int i = 0; i < m; ++i)
      if (i == 0 || buf[i] != buf[i - 1]) buf[cnt++] = buf[i];
    int32_t *wr = win_rows + t * CAP;
    for (int i = 0; i < cnt; ++i) wr[i] = buf[i];
    win_cnt[t] = cnt;
    pa += cnt > 0 ? (cnt + 415) / 416 : 1;
    pb += cnt > 0 ? (cnt + 511) / 512 : 1;
    uint16_t *sl = win_slots + t * CAP;
    for (int k = 0; k < KV; ++k)
      for (int r = 0; r < BM; ++r) {
        const int64_t row = t * BM + r;
        const int32_t idx = row < n_out ? nbr[(int64_t)k * n_out + row] : -1;
        uint16_t s = 0xFFFFu;
        if (idx >= 0) {
          int lo = 0, hi = cnt - 1;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (buf[mid] < idx) lo = mid + 1;
            else hi = mid;
          }
          s = (uint16_t)lo;
        }
        sl[k * BM + r] = s;
      }
  }
  win_stats[0] = (int32_t)pa;
  win_stats[1] = (int32_t)pb;
  win_stats[2] = win_stats[3] = 0;
  return 0;
}

/* semantic ensembling in one pass (include/pasco_hip.h ph_sem_ensemble; ensembler.py:159-187) */
int pho_sem_ensemble(const ph_sem_ens_desc *d, ph_stream_t stream) {
  (void)stream;
  if (!d || d->m < 1 || d->m > 8 || d->c < 1 || d->c > 64) return fail("sem_ensemble: bad desc");
  const int m = d->m, c = d->c;
#pragma omp parallel for schedule(static)
  for (int64_t s = 0; s < d->n_sites; ++s) {
    float mean[64], p[64];
    for (int ch = 0; ch < c; ++ch) mean[ch] = 0.f;
    for (int i = 0; i < m; ++i) {
      const int r = d->rows[i][s];
      if (r >= 0) {
        const float *x = d->logits[i] + (int64_t)r * c;
        float mx = x[0];
        for (int ch = 1; ch < c; ++ch) mx = x[ch] > mx ? x[ch] : mx;
        float sum = 0.f;
        for (int ch = 0; ch < c; ++ch) { p[ch] = expf(x[ch] - mx); sum += p[ch]; }
        for (int ch = 0; ch < c; ++ch) p[ch] = p[ch] / sum;
      } else {
        for (int ch = 0; ch < c; ++ch) p[ch] = ch == 0 ? 1.f : 0.f;
      }
      float best = p[0];
      for (int ch = 0; ch < c; ++ch) {
        d->out[i][s * c + ch] = p[ch];
        mean[ch] += p[ch];
        best = p[ch] > best ? p[ch] : best;
      }
      if (d->conf[i]) d->conf[i][s] = best;
    }
    float best = 0.f;
    for (int ch = 0; ch < c; ++ch) {
      const float v = mean[ch] / (float)m;
      d->out[m][s * c + ch] = v;
      best = (ch == 0 || v > best) ? v : best;
    }
    if (d->conf[m]) d->conf[m][s] = best;
  }
  return 0;
}

/* a10 */
int pho_maxpool_fwd(const float *in, int32_t c, const int32_t *nbr, int32_t kvol, int64_t n_out,
                    float *out, ph_stream_t stream) {
  (void)stream;
  if (c <= 0 || kvol < 1 || kvol > PH_MAX_KVOL) return fail("maxpool_fwd: bad shape");
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < n_out; ++o) {
    float *dst = out + o * c;
    int any = 0;
    for (int ch = 0; ch < c; ++ch) dst[ch] = 0.f;
    for (int k = 0; k < kvol; ++k) {
      int r = nbr[(int64_t)k * n_out + o];
      if (r < 0) continue;
      const float *src = in + (int64_t)r * c;
      for (int ch = 0; ch < c; ++ch) dst[ch] = (!any || src[ch] > dst[ch]) ? src[ch] : dst[ch];
      any = 1;
    }
  }
  return 0;
}

/* a7 */
int pho_mask_compact(const uint8_t *mask, int64_t n, int32_t *keep_rows, int32_t *n_keep, void *ws,
                     int64_t ws_bytes, ph_stream_t stream) {
  (void)ws; (void)ws_bytes; (void)stream;
  int32_t cnt = 0;
  for (int64_t i = 0; i < n; ++i)
    if (mask[i]) keep_rows[cnt++] = (int32_t)i;
  *n_keep = cnt;
  return 0;
}

/* ---- a12 / a13: voxel max of the point features + MIMO merge (unet3d_sparse_v2.py:53-86, augmenter.py:13-27) -------- */
int pho_points_bounds(const int64_t *xyz, int64_t n, int32_t *out6, ph_stream_t stream) {
  (void)stream;
  for (int a = 0; a < 3; ++a) out6[a] = INT_MAX, out6[3 + a] = INT_MIN;
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      int64_t v = xyz[i * 3 + a];
      int32_t w = v < INT_MIN ? INT_MIN : (v > INT_MAX ? INT_MAX : (int32_t)v);
      if (w < out6[a]) out6[a] = w;
      if (w > out6[3 + a]) out6[3 + a] = w;
    }
  return 0;
}

static int64_t pts_site(const int64_t *xyz, int64_t i, const int32_t *lo, const int32_t *dim) {
  int64_t x = xyz[i * 3] - lo[0], y = xyz[i * 3 + 1] - lo[1], z = xyz[i * 3 + 2] - lo[2];
  if (x < 0 || y < 0 || z < 0 || x >= dim[0] || y >= dim[1] || z >= dim[2]) return -1;
  return (x * dim[1] + y) * dim[2] + z;
}

int pho_points_mark(const int64_t *xyz, int64_t n, const int32_t *h_lo3, const int32_t *h_dims3, uint8_t *flags,
                    int32_t *status, ph_stream_t stream) {
  (void)stream;
  if (h_dims3[0] <= 0 || h_dims3[1] <= 0 || h_dims3[2] <= 0) return fail("points_mark: bad grid");
  for (int64_t i = 0; i < n; ++i) {
    int64_t s = pts_site(xyz, i, h_lo3, h_dims3);
    if (s >= 0) flags[s] = 1;
    else if (status) *status |= 8;
  }
  return 0;
}

int pho_mask_compact_rank(const uint8_t *mask, int64_t n, int32_t *keep_rows, int32_t *rank_of, int32_t *n_keep, void *ws,
                          int64_t ws_bytes, ph_stream_t stream) {
  (void)ws; (void)ws_bytes; (void)stream;
  int32_t cnt = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (mask[i]) {
      if (rank_of) rank_of[i] = cnt;
      keep_rows[cnt++] = (int32_t)i;
    } else if (rank_of) rank_of[i] = -1;
  }
  *n_keep = cnt;
  return 0;
}

int pho_points_link(const int64_t *xyz, int64_t n, const int64_t *h_starts, int32_t m, const int32_t *h_lo3,
                    const int32_t *h_dims3, const int32_t *rank_of, int32_t *head, int32_t *next, ph_stream_t stream) {
  (void)stream;
  if (m < 1 || m > 8) return fail("points_link: needs 1 <= m <= 8");
  int b = 0;
  for (int64_t i = 0; i < n; ++i) {
    while (b + 1 < m && i >= h_starts[b + 1]) ++b;
    int64_t s = pts_site(xyz, i, h_lo3, h_dims3);
    if (s < 0) {
      next[i] = -1;
      continue;
    }
    int64_t cell = (int64_t)rank_of[s] * m + b;
    next[i] = head[cell];
    head[cell] = (int32_t)i;
  }
  return 0;
}

int pho_cells_max(const float *h, int32_t c, const int32_t *head, const int32_t *next, int64_t v, int32_t m,
                  const int32_t *sites, const int32_t *h_lo3, const int32_t *h_dims3, float *out, int32_t *coords,
                  int32_t *status, ph_stream_t stream) {
  (void)stream;
  if (c <= 0 || c % 4 || m < 1 || m > 8) return fail("cells_max: needs c %% 4 == 0 and 1 <= m <= 8");
  for (int64_t r = 0; r < v; ++r) {
    int nz = 0;
    for (int b = 0; b < m; ++b) {
      float *dst = out + (r * m + b) * c;
      int p = head[r * m + b];
      for (int ch = 0; ch < c; ++ch) dst[ch] = p >= 0 ? h[(int64_t)p * c + ch] : 0.f;
      for (p = p >= 0 ? next[p] : -1; p >= 0; p = next[p])
        for (int ch = 0; ch < c; ++ch)
          if (h[(int64_t)p * c + ch] > dst[ch]) dst[ch] = h[(int64_t)p * c + ch];
      for (int ch = 0; ch < c; ++ch) nz |= dst[ch] != 0.f;
    }
    int s = sites[r];
    int z = s % h_dims3[2], xy = s / h_dims3[2];
    coords[r * 4] = 0;
    coords[r * 4 + 1] = xy / h_dims3[1] + h_lo3[0];
    coords[r * 4 + 2] = xy % h_dims3[1] + h_lo3[1];
    coords[r * 4 + 3] = z + h_lo3[2];
    if (!nz && status) *status |= 8;
  }
  return 0;
}

int pho_gather_rows(const void *src, int32_t c, const int32_t *rows, int64_t n_out, void *dst,
                    ph_stream_t stream) {
  (void)stream;
  if (c <= 0) return fail("gather_rows: c must be > 0");
  const uint32_t *s = (const uint32_t *)src;
  uint32_t *d = (uint32_t *)dst;
  for (int64_t j = 0; j < n_out; ++j) {
    if (rows[j] >= 0) memcpy(d + j * c, s + (int64_t)rows[j] * c, 4 * (size_t)c);
    else memset(d + j * c, 0, 4 * (size_t)c);
  }
  return 0;
}

/* a8 (feature side) */
int pho_scatter_add_rows(const float *src, int32_t c, const int32_t *rows, int64_t n_src, float *dst,
                         ph_stream_t stream) {
  (void)stream;
  if (c <= 0) return fail("scatter_add_rows: c must be > 0");
  for (int64_t i = 0; i < n_src; ++i) {
    if (rows[i] < 0) continue;
    float *d = dst + (int64_t)rows[i] * c;
    for (int ch = 0; ch < c; ++ch) d[ch] += src[i * c + ch];
  }
  return 0;
}

/* a9 */
int pho_to_dense(const float *feats, const int32_t *coords, int64_t n, int32_t c, const int32_t *h_min3,
                 int32_t ts, const int32_t *h_dims4, float *dense, ph_stream_t stream) {
  (void)stream;
  if (c <= 0 || ts <= 0) return fail("to_dense: bad c/ts");
  const int B = h_dims4[0], X = h_dims4[1], Y = h_dims4[2], Z = h_dims4[3];
  const int64_t per_b = (int64_t)X * Y * Z;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t *p = coords + 4 * i;
    int x = floor_div(p[1] - h_min3[0], ts), y = floor_div(p[2] - h_min3[1], ts), z = floor_div(p[3] - h_min3[2], ts);
    /* upstream writes `dense_F[b, :, x, y, z] = F` by advanced indexing [ME-upstream: MinkowskiSparseTensor.dense]: an index
     * in [-dim, 0) wraps around python-style (the padded rows of the attention mask rely on it,
     * transformer_predictor_v2.py:263-279 with SURVEY.md section 9 item 5); beyond that upstream raises - skipped here.
     * Rows are written in order, so the last of several rows of one site wins (torch's CPU index_put order). */
    if (x < 0) x += X;
    if (y < 0) y += Y;
    if (z < 0) z += Z;
    if (p[0] < 0 || p[0] >= B || x < 0 || x >= X || y < 0 || y >= Y || z < 0 || z >= Z) continue;
    int64_t site = ((int64_t)x * Y + y) * Z + z;
    for (int ch = 0; ch < c; ++ch) dense[((int64_t)p[0] * c + ch) * per_b + site] = feats[i * c + ch];
  }
  return 0;
}

int pho_to_sparse_coords(const float *dense, int32_t c, const int32_t *h_dims4, int32_t *out_coords,
                         int32_t *n_rows, void *ws, int64_t ws_bytes, ph_stream_t stream) {
  (void)ws; (void)ws_bytes; (void)stream;
  const int B = h_dims4[0], X = h_dims4[1], Y = h_dims4[2], Z = h_dims4[3];
  const int64_t per_b = (int64_t)X * Y * Z;
  int32_t cnt = 0;
  for (int b = 0; b < B; ++b)
    for (int x = 0; x < X; ++x)
      for (int y = 0; y < Y; ++y)
        for (int z = 0; z < Z; ++z) {
          int64_t site = ((int64_t)x * Y + y) * Z + z;
          int any = 0;
          for (int ch = 0; ch < c && !any; ++ch) any = dense[((int64_t)b * c + ch) * per_b + site] != 0.f;
          if (any) {
            int32_t *o = out_coords + 4 * (int64_t)cnt;
            o[0] = b; o[1] = x; o[2] = y; o[3] = z;
            ++cnt;
          }
        }
  *n_rows = cnt;
  return 0;
}

int pho_dense_gather(const float *dense, int32_t c, const int32_t *h_dims4, const int32_t *site_coords,
                     int64_t n, float *feats, ph_stream_t stream) {
  (void)stream;
  if (c <= 0) return fail("dense_gather: bad c");
  const int B = h_dims4[0], X = h_dims4[1], Y = h_dims4[2], Z = h_dims4[3];
  const int64_t per_b = (int64_t)X * Y * Z;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t *p = site_coords + 4 * i;
    int ok = p[0] >= 0 && p[0] < B && p[1] >= 0 && p[1] < X && p[2] >= 0 && p[2] < Y && p[3] >= 0 && p[3] < Z;
    int64_t site = ok ? ((int64_t)p[1] * Y + p[2]) * Z + p[3] : 0;
    for (int ch = 0; ch < c; ++ch) feats[i * c + ch] = ok ? dense[((int64_t)p[0] * c + ch) * per_b + site] : 0.f;
  }
  return 0;
}

/* ---- masked cross-attention (transformer/blocks.py:73-92; mask transformer_predictor_v2.py:220-289,
 * all-masked rule :163-164): plain two-pass softmax per (b, h, q) in double accumulation-free fp32. */
int64_t pho_attn_workspace_bytes(int64_t n, int32_t b, int32_t h, int32_t qn, int32_t dh) {
  (void)n; (void)b; (void)h; (void)qn; (void)dh;
  return 256;
}

int pho_attn_mask_pack(const float *vals, int64_t n, int32_t b, int32_t qn, int32_t positive_only, uint32_t *bits,
                       uint32_t *any, ph_stream_t stream) {
  (void)stream;
  if (qn < 1 || qn > 128) return fail("attn_mask_pack: bad qn");
  if (any) memset(any, 0, sizeof(uint32_t) * 4 * (size_t)b);
  for (int64_t row = 0; row < (int64_t)b * n; ++row) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int q = 0; q < qn; ++q)
      if (positive_only ? vals[row * qn + q] > 0.f : vals[row * qn + q] != 0.f) w[q >> 5] |= 1u << (q & 31);
    for (int i = 0; i < 4; ++i) {
      bits[row * 4 + i] = w[i];
      if (any) any[(row / n) * 4 + i] |= w[i];
    }
  }
  return 0;
}

int pho_attn_cross_fwd(const float *q, const float *k, const float *v, const uint32_t *bits, const uint32_t *any,
                       float *out, int64_t n, int32_t b, int32_t h, int32_t qn, int32_t dh, void *ws,
                       int64_t ws_bytes, ph_stream_t stream) {
  (void)ws; (void)ws_bytes; (void)stream;
  if (qn < 1 || qn > 128 || dh < 1 || dh > 512) return fail("attn_cross_fwd: bad shape");
  const int D = h * dh;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int bi = 0; bi < b; ++bi)
    for (int task = 0; task < h * qn; ++task) {
      const int hi = task / qn, qi = task % qn;
      const float *qv = q + (((int64_t)bi * h + hi) * qn + qi) * dh;
      int force = bits == NULL;
      if (bits && any) force = !((any[bi * 4 + (qi >> 5)] >> (qi & 31)) & 1u);
      float *sc = (float *)malloc(sizeof(float) * (size_t)n);
      float mx = -INFINITY;
      for (int64_t j = 0; j < n; ++j) {
        int ok = force || ((bits[((int64_t)bi * n + j) * 4 + (qi >> 5)] >> (qi & 31)) & 1u);
        float s = -INFINITY;
        if (ok) {
          const float *kv = k + ((int64_t)bi * n + j) * D + hi * dh;
          double sd = 0.0;
          for (int d = 0; d < dh; ++d) sd += (double)qv[d] * (double)kv[d];
          s = (float)sd;
        }
        sc[j] = s;
        if (s > mx) mx = s;
      }
      float *o = out + ((int64_t)bi * qn + qi) * D + hi * dh;
      for (int d = 0; d < dh; ++d) o[d] = 0.f;
      /* the sums over the keys run over up to ~10^6 terms (631 k keys at S10): accumulated in double - a sequential fp32 sum
       * of that length carries ~3e-5 of rounding of its own, which the decoder layers behind it amplify; the checker should
       * not be the less exact side of a comparison (round 5: tests/test_s10_end_to_end.py, three arithmetics) */
      double l = 0.0, od[512];
      for (int d = 0; d < dh && d < 512; ++d) od[d] = 0.0;
      if (mx > -INFINITY) {
        for (int64_t j = 0; j < n; ++j) {
          if (sc[j] == -INFINITY) continue;
          const double p = (double)expf(sc[j] - mx);
          l += p;
          const float *vv = v + ((int64_t)bi * n + j) * D + hi * dh;
          for (int d = 0; d < dh; ++d) od[d] += p * (double)vv[d];
        }
        for (int d = 0; d < dh; ++d) o[d] = (float)(od[d] / l);
      }
      free(sc);
    }
  return 0;
}


/* The same attention with K and V given as the split f16 operands the projections emit (hi + lo = value * 2^exp2):
 * recombine, undo the power of two, run the fp32 restatement above.  `status` is never raised here (no f16 Q split). */
int pho_attn_cross_split(const float *q, const void *k_split, const void *v_split, int32_t exp2, const uint32_t *bits,
                         const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h, int32_t qn, int32_t dh,
                         void *ws, int64_t ws_bytes, int32_t *status, ph_stream_t stream) {
  (void)status;
  if ((h * dh) % 32 != 0) return fail("attn_cross_split: model width is not a multiple of 32");
  float *k = unsplit_rows(k_split, (int64_t)b * n, h * dh, ldexpf(1.f, -exp2));
  float *v = unsplit_rows(v_split, (int64_t)b * n, h * dh, ldexpf(1.f, -exp2));
  int rc = pho_attn_cross_fwd(q, k, v, bits, any, out, n, b, h, qn, dh, ws, ws_bytes, stream);
  free(k);
  free(v);
  return rc;
}

/* Attention on the level's feature operand (include/pasco_hip.h attn_cross_feat; reference transformer/blocks.py:83-92 with
 * key = value = bb_feat + pos, transformer_predictor_v2.py:150,167-173): rows r = [x | aug] (x = hi + lo of the split operand
 * times 2^-exp2, aug as stored), scores q2 . r, masked softmax over the keys, out = sum_keys p r.  Plain two-pass fp32. */
static float f16_to_f32_at(const void *p, int64_t i) { return f16_bits_to_f32(((const uint16_t *)p)[i]); }

int pho_attn_cross_feat(const float *q2, const void *x_split, const void *aug, int32_t c, int32_t exp2, const uint32_t *bits,
                        const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h, int32_t qn, void *ws,
                        int64_t ws_bytes, int32_t *status, ph_stream_t stream) {
  (void)ws; (void)ws_bytes; (void)status; (void)stream;
  if (c < 32 || c % 32 != 0 || c > 1024) return fail("attn_cross_feat: channels must be a multiple of 32 (<= 1024)");
  if (qn < 1 || qn > 128 || b < 1 || h < 1 || n < 1) return fail("attn_cross_feat: bad shape");
  const int d = c + 16;
  float *x = unsplit_rows(x_split, (int64_t)b * n, c, ldexpf(1.f, -exp2));
  const float aug_unscale = ldexpf(1.f, -exp2);         /* the position columns carry the operand's scale (pho_pos_aug) */
  if (!x) return fail("attn_cross_feat: out of memory");
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int bi = 0; bi < b; ++bi)
    for (int task = 0; task < h * qn; ++task) {
      const int hi = task / qn, qi = task % qn;
      const float *qv = q2 + (((int64_t)bi * h + hi) * qn + qi) * d;
      int force = bits == NULL;
      if (bits && any) force = !((any[bi * 4 + (qi >> 5)] >> (qi & 31)) & 1u);
      float *sc = (float *)malloc(sizeof(float) * (size_t)n);
      float mx = -INFINITY;
      for (int64_t j = 0; j < n; ++j) {
        const int64_t row = (int64_t)bi * n + j;
        int ok = force || ((bits[row * 4 + (qi >> 5)] >> (qi & 31)) & 1u);
        float s = -INFINITY;
        if (ok) {
          double sd = 0.0;
          for (int ch = 0; ch < c; ++ch) sd += (double)qv[ch] * (double)x[row * c + ch];
          for (int ch = 0; ch < 16; ++ch) sd += (double)qv[c + ch] * (double)(f16_to_f32_at(aug, row * 16 + ch) * aug_unscale);
          s = (float)sd;
        }
        sc[j] = s;
        if (s > mx) mx = s;
      }
      float *o = out + ((int64_t)bi * qn + qi) * ((int64_t)h * d) + (int64_t)hi * d;
      for (int ch = 0; ch < d; ++ch) o[ch] = 0.f;
      double l = 0.0, od[1024 + 16];                     /* sums over the keys in double, as in pho_attn_cross_fwd */
      for (int ch = 0; ch < d; ++ch) od[ch] = 0.0;
      if (mx > -INFINITY) {
        for (int64_t j = 0; j < n; ++j) {
          if (sc[j] == -INFINITY) continue;
          const int64_t row = (int64_t)bi * n + j;
          const double p = (double)expf(sc[j] - mx);
          l += p;
          for (int ch = 0; ch < c; ++ch) od[ch] += p * (double)x[row * c + ch];
          for (int ch = 0; ch < 16; ++ch) od[c + ch] += p * (double)(f16_to_f32_at(aug, row * 16 + ch) * aug_unscale);
        }
        for (int ch = 0; ch < d; ++ch) o[ch] = (float)(od[ch] / l);
      }
      free(sc);
    }
  free(x);
  return 0;
}


int pho_pos_aug(const int32_t *coords, int64_t n, const float *eps, int32_t tab_lo, int32_t tab_n, int32_t exp2, void *aug,
                int32_t *status, ph_stream_t stream) {
  (void)stream;
  if (n < 0 || tab_n < 1 || exp2 < -14 || exp2 > 14) return fail("pos_aug: bad shape / exponent");
  const float pow2 = ldexpf(1.f, exp2);                 /* the columns extend the feature operand: same 2^exp2 */
  uint16_t *o = (uint16_t *)aug;
  for (int64_t i = 0; i < n; ++i) {
    for (int k = 0; k < 16; ++k) o[i * 16 + k] = 0;
    for (int ax = 0; ax < 3; ++ax) {
      const int v = coords[i * 4 + 1 + ax];
      int64_t t = (int64_t)v - tab_lo;
      if (t < 0 || t >= tab_n) {
        if (status) *status |= 4;
        t = t < 0 ? 0 : tab_n - 1;
      }
      o[i * 16 + ax] = f32_to_f16_bits(v == 0 ? pow2 : 0.f);
      o[i * 16 + 3 + ax] = f32_to_f16_bits(eps[t] * pow2);
    }
  }
  return 0;
}

/* keep masks of the decoder (decoder_v3.py:148-158, 411-420; include/pasco_hip.h keep_mask) */
int pho_keep_mask(const void *const *srcs, int32_t n_src, int32_t kind, const int32_t *coords, int64_t n, const int32_t *lo,
                  const int32_t *hi, int64_t fallback_rows, uint8_t *out, int32_t *any_word, ph_stream_t stream) {
  (void)stream; (void)any_word;
  if (n_src < 0 || n_src > 8 || (kind != 0 && kind != 1) || n < 0 || fallback_rows < 0) return fail("keep_mask: bad arguments");
  if ((lo == NULL) != (hi == NULL) || (lo != NULL && coords == NULL)) return fail("keep_mask: bounds need lo, hi and coords");
  int any = 0;
  for (int64_t r = 0; r < n; ++r) {
    int k = n_src == 0;            /* no source: the box test alone */
    for (int i = 0; i < n_src; ++i)
      k = k || (kind == 1 ? ((const int32_t *)srcs[i])[r] >= 0 : ((const uint8_t *)srcs[i])[r] != 0);
    out[r] = (uint8_t)k;
    any |= k;
  }
  for (int64_t r = 0; r < n; ++r) {
    int k = out[r];
    if (!any && fallback_rows > 0) k = r < fallback_rows;
    if (k && lo != NULL) {
      const int32_t *c = coords + r * 4 + 1;
      k = c[0] >= lo[0] && c[1] >= lo[1] && c[2] >= lo[2] && c[0] <= hi[0] && c[1] <= hi[1] && c[2] <= hi[2];
    }
    out[r] = (uint8_t)k;
  }
  return 0;
}

int pho_bits_orpool(const uint32_t *bits_in, const int32_t *nbr, int32_t kvol, int64_t n_out, uint32_t *bits_out,
                    ph_stream_t stream) {
  (void)stream;
  if (kvol < 1 || kvol > PH_MAX_KVOL) return fail("bits_orpool: bad kernel volume");
  for (int64_t o = 0; o < n_out; ++o)
    for (int w = 0; w < 4; ++w) {
      uint32_t m = 0;
      for (int k = 0; k < kvol; ++k) {
        int r = nbr[(int64_t)k * n_out + o];
        if (r >= 0) m |= bits_in[(int64_t)r * 4 + w];
      }
      bits_out[o * 4 + w] = m;
    }
  return 0;
}

int pho_bits_or_reduce(const uint32_t *bits, int64_t n, int32_t b, uint32_t *any, ph_stream_t stream) {
  (void)stream;
  for (int bi = 0; bi < b; ++bi)
    for (int w = 0; w < 4; ++w) {
      uint32_t m = 0;
      for (int64_t i = 0; i < n; ++i) m |= bits[((int64_t)bi * n + i) * 4 + w];
      any[bi * 4 + w] = m;
    }
  return 0;
}

/* sine position encoding (include/pasco_hip.h ph_sine_pe; reference: PositionEmbeddingSineSparse with
 * normalize=True as used by transformer_predictor_v2.py:93-95) */
int pho_sine_pe(const int32_t *coords, int64_t n, int32_t cstride, int32_t coff, int32_t f, const float *dim_t,
                float scale, const float *table, int32_t tab_lo, int32_t tab_n, float *out, ph_stream_t stream) {
  (void)stream;
  if (n < 0 || f <= 0 || f % 2 != 0 || cstride < 3 || coff < 0 || coff + 3 > cstride) return fail("sine_pe: bad shape");
  if (n == 0) return 0;
  if (!coords || !dim_t || !out) return fail("sine_pe: null buffer");
  const int half = f / 2;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    for (int a = 0; a < 3; ++a) {
      const int ci = coords[r * cstride + coff + a];
      float *o = out + r * 3 * f + a * f;
      if (table && ci >= tab_lo && ci - tab_lo < tab_n) {   /* lookup: same values by construction */
        memcpy(o, table + (int64_t)(ci - tab_lo) * f, sizeof(float) * (size_t)f);
        continue;
      }
      float c = (float)ci;
      c = c / (c + 1e-6f) * scale;
      for (int p = 0; p < half; ++p) {
        o[p] = sinf(c / dim_t[2 * p]);
        o[half + p] = cosf(c / dim_t[2 * p + 1]);
      }
    }
  }
  return 0;
}

/* ---- panoptic ensembling rows (ensembler.py:44-62, 86-98, 100-118; misc.py:46-57) ---------------------------------- */
int pho_ens_resample(const float *logits, int64_t n, int32_t q, const int32_t *rows, const int32_t *sel, int64_t u,
                     float *out, uint8_t *flag, ph_stream_t stream) {
  (void)stream; (void)n;
  if (q < 1 || q > 128) return fail("ens_resample: bad shape");
  for (int64_t s = 0; s < u; ++s) {
    const int r = rows[sel[s]];
    int any = 0;
    for (int c = 0; c < q; ++c) {
      const float v = r >= 0 ? 1.f / (1.f + expf(-logits[(int64_t)r * q + c])) : 0.f;
      out[s * q + c] = v;
      any |= v != 0.f;
    }
    flag[s] = (uint8_t)any;
  }
  return 0;
}

int pho_ens_merge(float *anchor, const float *m, const int32_t *perm, int64_t u, int32_t q, int32_t i, ph_stream_t stream) {
  (void)stream;
  if (q < 1 || q > 128 || i < 1) return fail("ens_merge: bad shape");
  const float fi = (float)i, den = fi + 1.f;
  float *tmp = (float *)malloc(sizeof(float) * (size_t)q);
  for (int64_t s = 0; s < u; ++s) {
    for (int c = 0; c < q; ++c) {
      volatile float t = anchor[s * q + c] * fi;      /* rounded product, then sum, then quotient (no fused multiply-add) */
      volatile float w = t + m[s * q + perm[c]];
      tmp[c] = w / den;
    }
    memcpy(anchor + s * q, tmp, sizeof(float) * (size_t)q);
  }
  free(tmp);
  return 0;
}

int pho_ens_finish(const float *anchor, int64_t u, int32_t q, const int32_t *keep, int32_t qk, const float *sem, int32_t c,
                   const int32_t *sel, float *out, uint8_t *flag, ph_stream_t stream) {
  (void)stream;
  if (q < 1 || q > 128 || qk < 0 || qk > q || c < 1) return fail("ens_finish: bad shape");
  for (int64_t s = 0; s < u; ++s) {
    const float *sr = sem + (int64_t)sel[s] * c;
    int best = 0;
    for (int k = 1; k < c; ++k)
      if (sr[k] > sr[best]) best = k;                  /* first maximum */
    const float nz = best != 0 ? 1.f : 0.f;
    int any = 0;
    for (int j = 0; j < qk; ++j) {
      const float v = anchor[s * q + keep[j]] * nz;
      out[s * qk + j] = v;
      any |= v != 0.f;
    }
    flag[s] = (uint8_t)any;
  }
  return 0;
}

/* same-map neighbour table of a symmetric kernel: by definition the table pho_nbr_build gives (the device halves its probes) */
int pho_nbr_build_same(const int32_t *coords, int64_t n, const uint64_t *tkeys, const int32_t *tvals, int64_t cap,
                       const int32_t *h_offsets, int32_t kvol, int32_t *nbr, ph_stream_t stream) {
  if (!(kvol & 1)) return fail("nbr_build_same: odd kernel volume expected");
  for (int k = 0; k < kvol; ++k)
    for (int a = 0; a < 3; ++a)
      if (h_offsets[3 * k + a] != -h_offsets[3 * (kvol - 1 - k) + a]) return fail("nbr_build_same: offsets are not symmetric");
  return pho_nbr_build(coords, n, tkeys, tvals, cap, h_offsets, kvol, nbr, stream);
}

/* attention-mask bits of a level straight from the fine map (transformer_predictor_v2.py:232-289 without wrap) */
int pho_bits_block_or(const int32_t *level_coords, int64_t m, int64_t n_per_b, int32_t s, const uint64_t *tkeys,
                      const int32_t *tvals, int64_t cap, const uint32_t *bits_in, const int32_t *lo, const int32_t *hi,
                      uint32_t *bits_out, int32_t *range, ph_stream_t stream) {
  (void)stream;
  if (s < 1 || s > 8 || n_per_b < 1 || !is_pow2(cap)) return fail("bits_block_or: bad shape");
  const uint64_t mask = (uint64_t)cap - 1;
  for (int64_t i = 0; i < m; ++i) {
    const int32_t *c = level_coords + 4 * i;
    const int b = (int)(i / n_per_b);
    if (range)
      for (int a = 0; a < 3; ++a)
        if (c[1 + a] < lo[b * 3 + a] || c[1 + a] > hi[b * 3 + a]) *range |= 1;
    uint32_t acc[4] = {0, 0, 0, 0};
    for (int dx = 0; dx < s; ++dx)
      for (int dy = 0; dy < s; ++dy)
        for (int dz = 0; dz < s; ++dz) {
          const int x = c[1] + dx, y = c[2] + dy, z = c[3] + dz;
          if (!packable(b, x, y, z)) continue;
          const int r = find(tkeys, tvals, mask, pack(b, x, y, z));
          if (r >= 0)
            for (int w = 0; w < 4; ++w) acc[w] |= bits_in[(int64_t)r * 4 + w];
        }
    for (int w = 0; w < 4; ++w) bits_out[i * 4 + w] = acc[w];
  }
  return 0;
}

/* canonical sites through T (transform_utils.py:60-74): float64 centre -> fp32 -> fp32 affine, sums in the reference's order */
int pho_project_canonical(const float *T, int32_t X, int32_t Y, int32_t Z, double resolution, const float *h_min_bound,
                          int32_t *out_coords, ph_stream_t stream) {
  (void)stream;
  if (X < 1 || Y < 1 || Z < 1 || !(resolution > 0)) return fail("project_canonical: bad grid");
  const float resf = (float)resolution, halff = (float)(resolution / 2);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)X * Y * Z; ++i) {
    const int c[3] = {(int)(i / ((int64_t)Y * Z)), (int)((i / Z) % Y), (int)(i % Z)};
    float p[3];
    for (int a = 0; a < 3; ++a) {
      volatile double m = (double)c[a] * resolution;
      volatile double h = m + resolution / 2;
      p[a] = (float)(h + (double)h_min_bound[a]);
    }
    out_coords[4 * i] = 0;
    for (int r = 0; r < 3; ++r) {
      volatile float t0 = T[r * 4 + 0] * p[0];
      volatile float t1 = T[r * 4 + 1] * p[1];
      volatile float v = t0 + t1;
      volatile float t2 = T[r * 4 + 2] * p[2];
      v = v + t2;
      v = v + T[r * 4 + 3];
      v = v - h_min_bound[r];
      v = v - halff;
      v = v / resf;
      out_coords[4 * i + 1 + r] = (int32_t)rintf(v);
    }
  }
  return 0;
}

/* padded row lists of a kernel map given in COO form (one list segment per offset, 128-aligned) */
int pho_rowlist_pack(const int32_t *pairs_in, const int32_t *pairs_out, const int32_t *counts, int32_t kvol, int64_t n_out,
                     int32_t *rl_in, int32_t *rl_out, int32_t *tile_k, int64_t cap, int64_t tcap, int32_t *status,
                     ph_stream_t stream) {
  (void)stream;
  if (kvol < 1 || kvol > PH_MAX_KVOL || cap % 128 != 0 || tcap * 128 < cap) return fail("rowlist_pack: bad shape");
  int64_t pos = 0, pairs = 0;
  for (int k = 0; k < kvol; ++k) pairs += counts[k];
  if (status && pairs != n_out) *status |= 32;          /* not one pair per output row */
  for (int64_t i = 0; i < cap; ++i) rl_in[i] = rl_out[i] = -1;
  for (int64_t t = 0; t < tcap; ++t) tile_k[t] = -1;
  for (int k = 0; k < kvol; ++k) {
    const int64_t padded = ((int64_t)counts[k] + 127) / 128 * 128;
    if (pos + padded > cap) return fail("rowlist_pack: list capacity too small");
    for (int64_t j = 0; j < counts[k]; ++j) {
      rl_in[pos + j] = pairs_in[(int64_t)k * n_out + j];
      rl_out[pos + j] = pairs_out[(int64_t)k * n_out + j];
    }
    for (int64_t t = pos / 128; t < (pos + padded) / 128; ++t) tile_k[t] = k;
    pos += padded;
  }
  return 0;
}


/* ---- panoptic post-processing (pasco/models/helper.py:91-303), see include/pasco_hip.h panop_* ---------------------- */
#define PANOP_QMAX 128

int pho_panop_queries(const float *qp, int32_t q, int32_t c1, float thr, int32_t *qtab, int32_t *nk, ph_stream_t stream) {
  (void)stream;
  if (q < 1 || q > PANOP_QMAX || c1 < 2 || c1 > 64) return fail("panop_queries: 1 <= q <= 128 queries, 2 <= c1 <= 64");
  int k = 0;
  for (int t = 0; t < PANOP_QMAX; ++t) {
    int label = 0;
    float prob = 0.f;
    int keep = 0;
    if (t < q) {                                            /* helper.py:132-140: probs, labels = query_probs.max(-1) */
      prob = qp[(int64_t)t * c1];
      for (int c = 1; c < c1; ++c)
        if (qp[(int64_t)t * c1 + c] > prob) prob = qp[(int64_t)t * c1 + c], label = c;
      keep = label != 0 && label != c1 - 1 && prob > thr;
    }
    qtab[0 * PANOP_QMAX + t] = keep ? k : -1;
    qtab[2 * PANOP_QMAX + t] = label;
    memcpy(&qtab[3 * PANOP_QMAX + t], &prob, 4);
    if (keep) qtab[1 * PANOP_QMAX + k++] = t;
  }
  nk[0] = k;
  return 0;
}

int pho_panop_argmax(const float *masks, int64_t n, int32_t q, const int32_t *qtab, float occ_thr, int32_t *winner,
                     uint8_t *own, float *conf, float *vunc, int32_t *areas, ph_stream_t stream) {
  (void)stream;
  if (q < 1 || q > PANOP_QMAX) return fail("panop_argmax: 1 <= q <= 128 queries");
  for (int64_t r = 0; r < n; ++r) {
    const float *row = masks + r * q;
    float bv = -1.f, bm = 0.f, sm = 0.f, sc = 0.f;
    int bk = -1;
    for (int c = 0; c < q; ++c) {
      const int k = qtab[c];
      if (k < 0) continue;
      float p;
      memcpy(&p, &qtab[3 * PANOP_QMAX + c], 4);
      const float m = row[c], v = p * m;                    /* helper.py:158-160 combined = prob * mask */
      if (m >= occ_thr) areas[PANOP_QMAX + k] += 1;         /* :203-205 original_area */
      if (v > bv) bv = v, bk = k, bm = m;                   /* :186 first maximum */
      sm += m;
      sc += v;
    }
    const int mine = bk >= 0 && bm >= occ_thr;              /* :199-201 */
    if (mine) areas[bk] += 1;
    winner[r] = bk;
    own[r] = (uint8_t)mine;
    conf[r] = bk >= 0 ? bm / (sm + 1e-8f) : 0.f;            /* :150-153 */
    vunc[r] = bk >= 0 ? bv / sc : 0.f;                      /* :236-239 */
  }
  return 0;
}

int pho_panop_write(int64_t n, const int32_t *winner, const uint8_t *own, const float *conf, const float *vunc,
                    const int32_t *areas, const int32_t *qtab, const int32_t *nk, double overlap_thr, uint64_t thing_mask,
                    int32_t *panoptic, int32_t *semantic, float *ins_unc, float *vox_conf, float *vox_unc, int32_t *seg,
                    ph_stream_t stream) {
  (void)stream;
  int s_seg[PANOP_QMAX], s_full[PANOP_QMAX], s_cls[PANOP_QMAX], stuff_seg[64];
  float s_prob[PANOP_QMAX];
  memset(stuff_seg, 0, sizeof(stuff_seg));
  const int K = nk[0];
  int current = 0;
  for (int k = 0; k < K; ++k) {                             /* helper.py:188-250 */
    const int qid = qtab[1 * PANOP_QMAX + k];
    const int cls = qtab[2 * PANOP_QMAX + qid];
    float prob;
    memcpy(&prob, &qtab[3 * PANOP_QMAX + qid], 4);
    const int ma = areas[k], oa = areas[PANOP_QMAX + k];
    s_seg[k] = 0, s_full[k] = 0, s_cls[k] = cls, s_prob[k] = prob;
    if (!(ma > 0 && oa > 0) || (double)ma / (double)oa < overlap_thr) continue;
    const int isthing = cls < 64 && ((thing_mask >> cls) & 1ull);
    if (!isthing) {
      if (cls < 64 && stuff_seg[cls] != 0) {
        s_seg[k] = stuff_seg[cls];
        continue;
      }
      if (cls < 64) stuff_seg[cls] = current + 1;
    }
    ++current;
    s_seg[k] = current, s_full[k] = 1;
    if (seg) {
      seg[0 * PANOP_QMAX + current - 1] = current;
      seg[1 * PANOP_QMAX + current - 1] = isthing;
      seg[2 * PANOP_QMAX + current - 1] = cls;
      seg[3 * PANOP_QMAX + current - 1] = qid;
    }
  }
  if (seg) seg[4 * PANOP_QMAX] = current;
  for (int64_t i = 0; i < n; ++i) {
    const int w = winner[i];
    const int mine = own[i] != 0 && w >= 0;
    const int hit = mine && s_seg[w] != 0, full = mine && s_full[w] != 0;
    panoptic[i] = hit ? s_seg[w] : 0;
    semantic[i] = full ? s_cls[w] : 0;
    ins_unc[i] = full ? s_prob[w] : 0.f;
    vox_conf[i] = full ? conf[i] : 0.f;
    vox_unc[i] = full ? vunc[i] : 0.f;
  }
  return 0;
}
