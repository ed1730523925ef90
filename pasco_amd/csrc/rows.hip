// Row movement, local max pooling and dense <-> sparse conversion kernels (gfx950).
// All of these are pure HBM-bound copies: one float per lane with the
// channel index fastest so that a row of C floats is read/written as contiguous 4*C bytes.
#include "ph_common.h"

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

// ---- gather / scatter-add -------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
    k_gather_rows(const T *__restrict__ src, int c, const int32_t *__restrict__ rows, int64_t n_out,
                  T *__restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * c) return;
  int64_t j = t / c;
  int ch = (int)(t - j * c);
  int r = rows[j];
  T v;
  memset(&v, 0, sizeof(T));
  if (r >= 0) v = src[(int64_t)r * c + ch];
  dst[t] = v;
}

extern "C" int ph_gather_rows(const void *src, int32_t c, const int32_t *rows, int64_t n_out,
                              void *dst, ph_stream_t stream) {
  PH_REQUIRE(c > 0, "gather_rows: c must be > 0");
  if (n_out == 0) return 0;
  hipStream_t st = ph_stream(stream);
  if (c % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    int c4 = c / 4;
    hipLaunchKernelGGL(k_gather_rows<float4>, dim3(nblk(n_out * c4, 256)), dim3(256), 0, st,
                       (const float4 *)src, c4, rows, n_out, (float4 *)dst);
  } else {
    hipLaunchKernelGGL(k_gather_rows<float>, dim3(nblk(n_out * c, 256)), dim3(256), 0, st,
                       (const float *)src, c, rows, n_out, (float *)dst);
  }
  PH_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256)
    k_scatter_add_rows(const float *__restrict__ src, int c, const int32_t *__restrict__ rows,
                       int64_t n_src, float *__restrict__ dst) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_src * c) return;
  int64_t i = t / c;
  int ch = (int)(t - i * c);
  int r = rows[i];
  if (r >= 0) dst[(int64_t)r * c + ch] += src[t];
}

extern "C" int ph_scatter_add_rows(const float *src, int32_t c, const int32_t *rows, int64_t n_src,
                                   float *dst, ph_stream_t stream) {
  PH_REQUIRE(c > 0, "scatter_add_rows: c must be > 0");
  if (n_src == 0) return 0;
  hipLaunchKernelGGL(k_scatter_add_rows, dim3(nblk(n_src * c, 256)), dim3(256), 0,
                     ph_stream(stream), src, c, rows, n_src, dst);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- keep masks of the decoder in one pass (ph_keep_mask) ---------------------------------------------------------------
// out[r] = K[r] && (lo <= coords[r].xyz <= hi), K[r] = OR over the sources of "kept" (kind 0: byte != 0; kind 1: int32 >= 0, the
// output of ph_map_find).  k_keep_mask also raises *any when some K[r] holds; k_keep_fallback (a few workgroups, always
// launched) rewrites the first `fallback` rows to the bounds test alone when nothing was kept - the reference's "nothing kept
// -> keep the first 1000 rows" (decoder_v3.py:415-418), decided on the device.
struct KeepSrcs {
  const void *p[8];
};
__global__ void __launch_bounds__(256)
    k_keep_mask(KeepSrcs srcs, int n_src, int kind, const int4 *__restrict__ coords, int64_t n, const int32_t *__restrict__ lo,
                const int32_t *__restrict__ hi, uint8_t *__restrict__ out, int32_t *__restrict__ any) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool k = n_src == 0;             // no source: the box test alone (decoder_v3.py:151-158 on generated children)
  if (r < n) {
    for (int i = 0; i < n_src; ++i)
      k = k || (kind == 1 ? reinterpret_cast<const int32_t *>(srcs.p[i])[r] >= 0 : reinterpret_cast<const uint8_t *>(srcs.p[i])[r] != 0);
  }
  if (any != nullptr && __ballot(k) != 0ull && (threadIdx.x & 63) == 0) {
    if (__hip_atomic_load(any, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(any, 1);
  }
  if (r >= n) return;
  bool inside = true;
  if (lo != nullptr) {
    const int4 c = coords[r];
    inside = c.y >= lo[0] && c.z >= lo[1] && c.w >= lo[2] && c.y <= hi[0] && c.z <= hi[1] && c.w <= hi[2];
  }
  out[r] = (k && inside) ? 1 : 0;
}
__global__ void __launch_bounds__(256)
    k_keep_fallback(const int4 *__restrict__ coords, int64_t n, int64_t fallback, const int32_t *__restrict__ lo,
                    const int32_t *__restrict__ hi, uint8_t *__restrict__ out, const int32_t *__restrict__ any) {
  if (*any != 0) return;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || r >= fallback) return;
  bool inside = true;
  if (lo != nullptr) {
    const int4 c = coords[r];
    inside = c.y >= lo[0] && c.z >= lo[1] && c.w >= lo[2] && c.y <= hi[0] && c.z <= hi[1] && c.w <= hi[2];
  }
  out[r] = inside ? 1 : 0;
}

extern "C" int ph_keep_mask(const void *const *srcs, int32_t n_src, int32_t kind, const int32_t *coords, int64_t n,
                            const int32_t *lo, const int32_t *hi, int64_t fallback_rows, uint8_t *out, int32_t *any_word,
                            ph_stream_t stream) {
  PH_REQUIRE(n_src >= 0 && n_src <= 8 && (kind == 0 || kind == 1) && n >= 0 && fallback_rows >= 0, "keep_mask: bad arguments");
  PH_REQUIRE((lo == nullptr) == (hi == nullptr) && (lo == nullptr || coords != nullptr), "keep_mask: bounds need lo, hi and coords");
  PH_REQUIRE(fallback_rows == 0 || any_word != nullptr, "keep_mask: the fallback needs the scratch word");
  if (n == 0) return 0;
  KeepSrcs ks;
  for (int i = 0; i < 8; ++i) ks.p[i] = i < n_src ? srcs[i] : nullptr;
  hipStream_t st = ph_stream(stream);
  int32_t *any = fallback_rows > 0 ? any_word : nullptr;
  if (any) PH_CHECK_HIP(hipMemsetAsync(any, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(k_keep_mask, dim3(nblk(n, 256)), dim3(256), 0, st, ks, n_src, kind, (const int4 *)coords, n, lo, hi, out, any);
  PH_LAUNCH_CHECK();
  if (any) {
    const int64_t m = n < fallback_rows ? n : fallback_rows;
    hipLaunchKernelGGL(k_keep_fallback, dim3(nblk(m, 256)), dim3(256), 0, st, (const int4 *)coords, n, fallback_rows, lo, hi, out,
                       any);
    PH_LAUNCH_CHECK();
  }
  return 0;
}

// ---- local max pooling ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_maxpool(const float *__restrict__ in, int c, const int32_t *__restrict__ nbr, int kvol,
              int64_t n_out, float *__restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * c) return;
  int64_t o = t / c;
  int ch = (int)(t - o * c);
  float m = 0.f;
  bool any = false;
  for (int k = 0; k < kvol; ++k) {
    int r = nbr[(int64_t)k * n_out + o];
    if (r >= 0) {
      float v = in[(int64_t)r * c + ch];
      m = any ? fmaxf(m, v) : v;
      any = true;
    }
  }
  out[t] = m;
}

extern "C" int ph_maxpool_fwd(const float *in, int32_t c, const int32_t *nbr, int32_t kvol,
                              int64_t n_out, float *out, ph_stream_t stream) {
  PH_REQUIRE(c > 0 && kvol >= 1 && kvol <= PH_MAX_KVOL, "maxpool_fwd: bad shape");
  if (n_out == 0) return 0;
  hipLaunchKernelGGL(k_maxpool, dim3(nblk(n_out * c, 256)), dim3(256), 0, ph_stream(stream), in, c,
                     nbr, kvol, n_out, out);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- dense <-> sparse -------------------------------------------------------------------------------
struct Dims4 {
  int b, x, y, z;
};

__device__ __forceinline__ int floor_div(int v, int ts) {
  int q = v / ts;
  if ((v % ts) != 0 && ((v < 0) != (ts < 0))) --q;
  return q;
}

__global__ void __launch_bounds__(256)
    k_to_dense(const float *__restrict__ feats, const int4 *__restrict__ coords, int64_t n, int c,
               int mx, int my, int mz, int ts, Dims4 d, float *__restrict__ dense) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  int64_t i = t / c;
  int ch = (int)(t - i * c);
  int4 p = coords[i];
  int x = floor_div(p.y - mx, ts), y = floor_div(p.z - my, ts), z = floor_div(p.w - mz, ts);
  // upstream's dense() assigns by advanced indexing: an index in [-dim, 0) wraps around python-style (the padded rows of the
  // attention mask rely on it, transformer_predictor_v2.py:263-279); anything further out is skipped (upstream raises).
  // Several rows on one site: unordered here, as in upstream's GPU index_put.
  if (x < 0) x += d.x;
  if (y < 0) y += d.y;
  if (z < 0) z += d.z;
  if (p.x < 0 || p.x >= d.b || x < 0 || x >= d.x || y < 0 || y >= d.y || z < 0 || z >= d.z) return;
  int64_t site = ((int64_t)x * d.y + y) * d.z + z;
  dense[((int64_t)p.x * c + ch) * ((int64_t)d.x * d.y * d.z) + site] = feats[t];
}

extern "C" int ph_to_dense(const float *feats, const int32_t *coords, int64_t n, int32_t c,
                           const int32_t *h_min3, int32_t ts, const int32_t *h_dims4, float *dense,
                           ph_stream_t stream) {
  PH_REQUIRE(c > 0 && ts > 0, "to_dense: bad c/ts");
  if (n == 0) return 0;
  Dims4 d{h_dims4[0], h_dims4[1], h_dims4[2], h_dims4[3]};
  hipLaunchKernelGGL(k_to_dense, dim3(nblk(n * c, 256)), dim3(256), 0, ph_stream(stream), feats,
                     (const int4 *)coords, n, c, h_min3[0], h_min3[1], h_min3[2], ts, d, dense);
  PH_LAUNCH_CHECK();
  return 0;
}

// flags[site] = any channel != 0 ; sites are enumerated (b, x, y, z) lexicographically
__global__ void __launch_bounds__(256)
    k_site_flags(const float *__restrict__ dense, int c, int64_t nsite_per_b, int64_t nsites,
                 uint8_t *__restrict__ flags) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nsites) return;
  int64_t b = s / nsite_per_b;
  int64_t site = s - b * nsite_per_b;
  const float *p = dense + b * c * nsite_per_b + site;
  bool any = false;
  for (int ch = 0; ch < c; ++ch) any |= (p[(int64_t)ch * nsite_per_b] != 0.f);
  flags[s] = any ? 1 : 0;
}

__global__ void __launch_bounds__(256)
    k_sites_to_coords(const int32_t *__restrict__ keep_rows, const int32_t *__restrict__ n_rows,
                      Dims4 d, int4 *__restrict__ out_coords) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= *n_rows) return;
  int64_t s = keep_rows[j];
  int z = (int)(s % d.z);
  s /= d.z;
  int y = (int)(s % d.y);
  s /= d.y;
  int x = (int)(s % d.x);
  s /= d.x;
  out_coords[j] = make_int4((int)s, x, y, z);
}

extern "C" int ph_to_sparse_coords(const float *dense, int32_t c, const int32_t *h_dims4,
                                   int32_t *out_coords, int32_t *n_rows, void *ws,
                                   int64_t ws_bytes, ph_stream_t stream) {
  Dims4 d{h_dims4[0], h_dims4[1], h_dims4[2], h_dims4[3]};
  int64_t per_b = (int64_t)d.x * d.y * d.z;
  int64_t nsites = per_b * d.b;
  PH_REQUIRE(c > 0 && nsites < 0x7FFFFF00, "to_sparse_coords: bad shape");
  hipStream_t st = ph_stream(stream);
  if (nsites == 0) {
    PH_CHECK_HIP(hipMemsetAsync(n_rows, 0, 4, st));
    return 0;
  }
  PH_REQUIRE(ws_bytes >= ph_workspace_bytes(nsites), "to_sparse_coords: workspace too small");
  // carve: flags[nsites] (padded to 16) | keep_rows[nsites] | block counters
  uint8_t *flags = (uint8_t *)ws;
  int64_t fpad = (nsites + 15) & ~(int64_t)15;
  int32_t *keep_rows = (int32_t *)(flags + fpad);
  char *rest = (char *)(keep_rows + nsites);
  int64_t rest_bytes = ws_bytes - fpad - 4 * nsites;
  hipLaunchKernelGGL(k_site_flags, dim3(nblk(nsites, 256)), dim3(256), 0, st, dense, c, per_b, nsites,
                     flags);
  PH_LAUNCH_CHECK();
  int rc = ph_compact_flags(flags, nsites, keep_rows, nullptr, n_rows, rest, rest_bytes, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_sites_to_coords, dim3(nblk(nsites, 256)), dim3(256), 0, st, keep_rows, n_rows, d,
                     (int4 *)out_coords);
  PH_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256)
    k_dense_gather(const float *__restrict__ dense, int c, Dims4 d, const int4 *__restrict__ sc,
                   int64_t n, float *__restrict__ feats) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * c) return;
  int64_t i = t / c;
  int ch = (int)(t - i * c);
  int4 p = sc[i];
  float v = 0.f;
  if (p.x >= 0 && p.x < d.b && p.y >= 0 && p.y < d.x && p.z >= 0 && p.z < d.y && p.w >= 0 && p.w < d.z) {
    int64_t per_b = (int64_t)d.x * d.y * d.z;
    int64_t site = ((int64_t)p.y * d.y + p.z) * d.z + p.w;
    v = dense[((int64_t)p.x * c + ch) * per_b + site];
  }
  feats[t] = v;
}

extern "C" int ph_dense_gather(const float *dense, int32_t c, const int32_t *h_dims4,
                               const int32_t *site_coords, int64_t n, float *feats,
                               ph_stream_t stream) {
  PH_REQUIRE(c > 0, "dense_gather: bad c");
  if (n == 0) return 0;
  Dims4 d{h_dims4[0], h_dims4[1], h_dims4[2], h_dims4[3]};
  hipLaunchKernelGGL(k_dense_gather, dim3(nblk(n * c, 256)), dim3(256), 0, ph_stream(stream), dense, c,
                     d, (const int4 *)site_coords, n, feats);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- sine position encoding -----------------------------------------------------------------------
// one thread per output element; consecutive threads write consecutive columns of a row
__global__ void __launch_bounds__(256) k_sine_pe(const int32_t *__restrict__ coords, int64_t n, int cstride, int coff, int f,
                                                  const float *__restrict__ dim_t, float scale,
                                                  const float *__restrict__ table, int tab_lo, int tab_n,
                                                  float *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = 3 * f;
  if (t >= n * w) return;
  const int64_t row = t / w;
  const int col = (int)(t - row * w);
  const int axis = col / f;
  const int j = col - axis * f;
  const int half = f >> 1;
  const int ci = coords[row * cstride + coff + axis];
  const unsigned ti = (unsigned)(ci - tab_lo);
  float v;
  if (table != nullptr && ti < (unsigned)tab_n) {
    v = table[(int64_t)ti * f + j];
  } else {
    float c = (float)ci;
    c = c / (c + 1e-6f) * scale;
    if (j < half) v = sinf(c / dim_t[2 * j]);
    else v = cosf(c / dim_t[2 * (j - half) + 1]);
  }
  out[t] = v;
}

extern "C" int ph_sine_pe(const int32_t *coords, int64_t n, int32_t cstride, int32_t coff, int32_t f,
                          const float *dim_t, float scale, const float *table, int32_t tab_lo, int32_t tab_n,
                          float *out, ph_stream_t stream) {
  PH_REQUIRE(n >= 0 && f > 0 && f % 2 == 0 && cstride >= 3 && coff >= 0 && coff + 3 <= cstride, "sine_pe: bad shape");
  PH_REQUIRE(table == nullptr || tab_n > 0, "sine_pe: empty table");
  if (n == 0) return 0;
  PH_REQUIRE(coords && dim_t && out, "sine_pe: null buffer");
  hipLaunchKernelGGL(k_sine_pe, dim3(nblk(n * 3 * f, 256)), dim3(256), 0, ph_stream(stream), coords, n, cstride, coff, f,
                     dim_t, scale, table, tab_lo, tab_n, out);
  PH_LAUNCH_CHECK();
  return 0;
}


// ---- semantic ensembling in one pass (ph_sem_ensemble) ---------------------------------------------------------------
// One thread per canonical site: for every subnet gather its logits row (or take the one-hot of class 0), softmax in
// registers, write the subnet's row, accumulate the mean, keep the row maxima.  HBM bound: per site 4 B of row index and
// up to 4 C bytes of logits per subnet in, 4 C (m + 1) bytes out (one pass instead of softmax / gather / column fix-up /
// stack / mean / max passes of 168 MB tensors each).
template <int C, bool VEC>
__global__ void __launch_bounds__(256) k_sem_ensemble(ph_sem_ens_desc d) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.n_sites) return;
  float mean[C];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) mean[ch] = 0.f;
  for (int i = 0; i < d.m; ++i) {
    float p[C];
    const int r = d.rows[i][s];
    if (r >= 0) {
      const float *x = d.logits[i] + (int64_t)r * C;
      if (VEC) {
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
          const float4 v = reinterpret_cast<const float4 *>(x)[q];
          p[4 * q] = v.x; p[4 * q + 1] = v.y; p[4 * q + 2] = v.z; p[4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) p[ch] = x[ch];
      }
      float mx = p[0];
#pragma unroll
      for (int ch = 1; ch < C; ++ch) mx = fmaxf(mx, p[ch]);
      float sum = 0.f;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        p[ch] = expf(p[ch] - mx);
        sum += p[ch];
      }
#pragma unroll
      for (int ch = 0; ch < C; ++ch) p[ch] = p[ch] / sum;
    } else {
#pragma unroll
      for (int ch = 0; ch < C; ++ch) p[ch] = ch == 0 ? 1.f : 0.f;
    }
    float best = p[0];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      mean[ch] += p[ch];
      best = fmaxf(best, p[ch]);
    }
    float *o = d.out[i] + s * C;
    if (VEC) {
#pragma unroll
      for (int q = 0; q < C / 4; ++q) reinterpret_cast<float4 *>(o)[q] = make_float4(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
    } else {
#pragma unroll
      for (int ch = 0; ch < C; ++ch) o[ch] = p[ch];
    }
    if (d.conf[i]) d.conf[i][s] = best;
  }
  const float inv_m = (float)d.m;
  float best = 0.f;
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    mean[ch] = mean[ch] / inv_m;
    best = ch == 0 ? mean[ch] : fmaxf(best, mean[ch]);
  }
  float *o = d.out[d.m] + s * C;
  if (VEC) {
#pragma unroll
    for (int q = 0; q < C / 4; ++q) reinterpret_cast<float4 *>(o)[q] = make_float4(mean[4 * q], mean[4 * q + 1], mean[4 * q + 2], mean[4 * q + 3]);
  } else {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) o[ch] = mean[ch];
  }
  if (d.conf[d.m]) d.conf[d.m][s] = best;
}

extern "C" int ph_sem_ensemble(const ph_sem_ens_desc *desc, ph_stream_t stream) {
  PH_REQUIRE(desc != nullptr && desc->m >= 1 && desc->m <= 8, "sem_ensemble: 1 <= m <= 8");
  PH_REQUIRE(desc->c == 20 || desc->c == 19, "sem_ensemble: serves 20 (SemanticKITTI) or 19 (SSCBench-KITTI360) classes, got %d", desc->c);
  if (desc->n_sites == 0) return 0;
  for (int i = 0; i < desc->m; ++i) PH_REQUIRE(desc->logits[i] && desc->rows[i] && desc->out[i], "sem_ensemble: null buffer");
  PH_REQUIRE(desc->out[desc->m] != nullptr, "sem_ensemble: null mean buffer");
  hipStream_t st = ph_stream(stream);
  const dim3 grid((unsigned)((desc->n_sites + 255) / 256));
  bool al = true;
  for (int i = 0; i <= desc->m; ++i) al = al && (((uintptr_t)desc->out[i]) & 15) == 0;
  for (int i = 0; i < desc->m; ++i) al = al && (((uintptr_t)desc->logits[i]) & 15) == 0;
  if (desc->c == 20 && al)
    hipLaunchKernelGGL((k_sem_ensemble<20, true>), grid, dim3(256), 0, st, *desc);
  else if (desc->c == 20)
    hipLaunchKernelGGL((k_sem_ensemble<20, false>), grid, dim3(256), 0, st, *desc);
  else
    hipLaunchKernelGGL((k_sem_ensemble<19, false>), grid, dim3(256), 0, st, *desc);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- panoptic ensembling on the compacted union of occupied canonical sites (ensembler.py:20-131; graph/ensemble.py) ----
// One wave64 per union site (a row of <= 128 query columns: lanes l and l + 64), grid-stride over the sites.  Each kernel
// replaces a chain of elementwise / gather passes over [U, Q] tensors (~100 MB each at S10) by one pass.

// out[u] = sigmoid(logits[rows[sel[u]]]) or zeros when the subnet has no voxel there; flag[u] = any non-zero entry
__global__ void __launch_bounds__(256)
    k_ens_resample(const float *__restrict__ logits, int q, const int32_t *__restrict__ rows, const int32_t *__restrict__ sel,
                   int64_t u, float *__restrict__ out, uint8_t *__restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < u; s += nwaves) {
    const int r = rows[sel[s]];
    float v0 = 0.f, v1 = 0.f;
    if (r >= 0) {
      const float *x = logits + (int64_t)r * q;
      if (lane < q) v0 = 1.f / (1.f + expf(-x[lane]));
      if (lane + 64 < q) v1 = 1.f / (1.f + expf(-x[lane + 64]));
    }
    float *o = out + s * q;
    if (lane < q) o[lane] = v0;
    if (lane + 64 < q) o[lane + 64] = v1;
    const unsigned long long any = __ballot(v0 != 0.f || v1 != 0.f);
    if (lane == 0) flag[s] = any != 0ull ? 1 : 0;
  }
}

// anchor[u][c] = (anchor[u][c] * i + m[u][perm[c]]) / (i + 1): the running mean of the matched query masks
__global__ void __launch_bounds__(256)
    k_ens_merge(float *__restrict__ anchor, const float *__restrict__ m, const int32_t *__restrict__ perm, int64_t u, int q,
                float fi) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int p0 = lane < q ? perm[lane] : 0, p1 = lane + 64 < q ? perm[lane + 64] : 0;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float den = fi + 1.f;
  for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < u; s += nwaves) {
    float *a = anchor + s * q;
    const float *b = m + s * q;
    if (lane < q) {
      const float t = a[lane] * fi;
      a[lane] = (t + b[p0]) / den;
    }
    if (lane + 64 < q) {
      const float t = a[lane + 64] * fi;
      a[lane + 64] = (t + b[p1]) / den;
    }
  }
}

// out[u][j] = anchor[u][keep[j]] * (argmax_c sem[sel[u]][c] != 0); flag[u] = any non-zero entry
__global__ void __launch_bounds__(256)
    k_ens_finish(const float *__restrict__ anchor, int q, const int32_t *__restrict__ keep, int qk, const float *__restrict__ sem,
                 int c, const int32_t *__restrict__ sel, int64_t u, float *__restrict__ out, uint8_t *__restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int k0 = lane < qk ? keep[lane] : 0, k1 = lane + 64 < qk ? keep[lane + 64] : 0;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < u; s += nwaves) {
    // class 0 wins the argmax iff no later class is strictly larger (first maximum, as torch.argmax on ties)
    const float *sr = sem + (int64_t)sel[s] * c;
    const float p0 = sr[0];
    const bool larger = lane >= 1 && lane < c && sr[lane] > p0;
    const float nz = __ballot(larger) != 0ull ? 1.f : 0.f;
    const float *a = anchor + s * q;
    float v0 = 0.f, v1 = 0.f;
    if (lane < qk) v0 = a[k0] * nz;
    if (lane + 64 < qk) v1 = a[k1] * nz;
    float *o = out + s * qk;
    if (lane < qk) o[lane] = v0;
    if (lane + 64 < qk) o[lane + 64] = v1;
    const unsigned long long any = __ballot(v0 != 0.f || v1 != 0.f);
    if (lane == 0) flag[s] = any != 0ull ? 1 : 0;
  }
}

static unsigned ens_grid(int64_t u) {
  int64_t g = (u + 3) / 4;
  if (g > 16384) g = 16384;
  return (unsigned)(g < 1 ? 1 : g);
}

extern "C" int ph_ens_resample(const float *logits, int64_t n, int32_t q, const int32_t *rows, const int32_t *sel, int64_t u,
                               float *out, uint8_t *flag, ph_stream_t stream) {
  PH_REQUIRE(q >= 1 && q <= 128 && n >= 0 && u >= 0, "ens_resample: bad shape");
  if (u == 0) return 0;
  PH_REQUIRE(logits && rows && sel && out && flag, "ens_resample: null buffer");
  hipLaunchKernelGGL(k_ens_resample, dim3(ens_grid(u)), dim3(256), 0, ph_stream(stream), logits, q, rows, sel, u, out, flag);
  PH_LAUNCH_CHECK();
  return 0;
}

extern "C" int ph_ens_merge(float *anchor, const float *m, const int32_t *perm, int64_t u, int32_t q, int32_t i,
                            ph_stream_t stream) {
  PH_REQUIRE(q >= 1 && q <= 128 && u >= 0 && i >= 1, "ens_merge: bad shape");
  if (u == 0) return 0;
  PH_REQUIRE(anchor && m && perm, "ens_merge: null buffer");
  hipLaunchKernelGGL(k_ens_merge, dim3(ens_grid(u)), dim3(256), 0, ph_stream(stream), anchor, m, perm, u, q, (float)i);
  PH_LAUNCH_CHECK();
  return 0;
}

extern "C" int ph_ens_finish(const float *anchor, int64_t u, int32_t q, const int32_t *keep, int32_t qk, const float *sem,
                             int32_t c, const int32_t *sel, float *out, uint8_t *flag, ph_stream_t stream) {
  PH_REQUIRE(q >= 1 && q <= 128 && qk >= 0 && qk <= q && c >= 1 && c <= 64 && u >= 0, "ens_finish: bad shape");
  if (u == 0) return 0;
  PH_REQUIRE(anchor && sem && sel && flag && (qk == 0 || (keep && out)), "ens_finish: null buffer");
  hipLaunchKernelGGL(k_ens_finish, dim3(ens_grid(u)), dim3(256), 0, ph_stream(stream), anchor, q, keep, qk, sem, c, sel, u, out,
                     flag);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- canonical voxel centres seen through a subnet's transform (transform_utils.py:60-74; graph/ensemble.py) ---------------
// site id -> (x, y, z) of the X x Y x Z grid (lexicographic) -> metres (float64 affine, as the reference's numpy grid) ->
// fp32 -> T (3 x 4, row-major, fp32 products and sums in the reference's order, no fused multiply-add) -> voxel index
// (round half to even).  One thread per site; out[i] = (0, x', y', z').
__global__ void __launch_bounds__(256)
    k_project_canonical(const float *__restrict__ T, int X, int Y, int Z, double res, float mb0, float mb1, float mb2,
                        int4 *__restrict__ out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)X * Y * Z;
  if (i >= n) return;
  const int z = (int)(i % Z), y = (int)((i / Z) % Y), x = (int)(i / ((int64_t)Y * Z));
  const float mb[3] = {mb0, mb1, mb2};
  const int c[3] = {x, y, z};
  float p[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double m = (double)c[a] * res;
    const double h = m + res / 2;
    p[a] = (float)(h + (double)mb[a]);
  }
  const float resf = (float)res, halff = (float)(res / 2);
  int q[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float t0 = T[r * 4 + 0] * p[0];
    const float t1 = T[r * 4 + 1] * p[1];
    float v = t0 + t1;
    const float t2 = T[r * 4 + 2] * p[2];
    v = v + t2;
    v = v + T[r * 4 + 3];
    v = v - mb[r];
    v = v - halff;
    v = v / resf;
    q[r] = (int)rintf(v);
  }
  out[i] = make_int4(0, q[0], q[1], q[2]);
}

extern "C" int ph_project_canonical(const float *T, int32_t X, int32_t Y, int32_t Z, double resolution, const float *h_min_bound,
                                    int32_t *out_coords, ph_stream_t stream) {
  PH_REQUIRE(X >= 1 && Y >= 1 && Z >= 1 && resolution > 0, "project_canonical: bad grid");
  PH_REQUIRE(T && h_min_bound && out_coords, "project_canonical: null buffer");
  const int64_t n = (int64_t)X * Y * Z;
  hipLaunchKernelGGL(k_project_canonical, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ph_stream(stream), T, X, Y, Z,
                     resolution, h_min_bound[0], h_min_bound[1], h_min_bound[2], (int4 *)out_coords);
  PH_LAUNCH_CHECK();
  return 0;
}
