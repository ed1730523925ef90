// Coordinate maps and kernel maps for gfx950: hash insert / find, strided and generative
// coordinate generation, output-stationary neighbour tables and their COO compaction.
//
// Everything here is HBM / latency bound integer work:
// one thread per coordinate row, 16-byte row loads, wave64 ballot + popcount prefix sums for the
// order-preserving compactions, no atomics on the output side.
#include <stdarg.h>

#include "ph_common.h"

// ---- error plumbing ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void ph_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ph_abi_version(void) { return PH_ABI_VERSION; }
extern "C" int ph_conv_desc_size(void) { return (int)sizeof(ph_conv_desc); }
extern "C" const char *ph_last_error(void) { return g_err; }

// ---- stable compaction --------------------------------------------------------------------------
constexpr int CP_THREADS = 256;
constexpr int CP_ROUNDS = 8;
constexpr int CP_TILE = CP_THREADS * CP_ROUNDS;

extern "C" int64_t ph_workspace_bytes(int64_t n) {
  if (n < 0) n = 0;
  // flags (n) + row_slot (4n) + rank_of (4n) + block counters (PH_MAX_KVOL segments) + slack
  int64_t nb = n / CP_TILE + 2;
  return 9 * n + (int64_t)PH_MAX_KVOL * nb * 4 + 4096;
}

template <class Pred>
__global__ void __launch_bounds__(CP_THREADS)
    k_compact_count(Pred pred, int64_t n, int32_t *__restrict__ block_counts) {
  const int seg = blockIdx.y;
  const int nb = gridDim.x;
  const int64_t base = (int64_t)blockIdx.x * CP_TILE;
  int cnt = 0;
#pragma unroll
  for (int r = 0; r < CP_ROUNDS; ++r) {
    int64_t i = base + r * CP_THREADS + threadIdx.x;
    bool p = (i < n) && pred(seg, i);
    cnt += __popcll(__ballot(p));
  }
  __shared__ int wcnt[CP_THREADS / PH_WAVE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) wcnt[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[(int64_t)seg * nb + blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// exclusive scan of each segment's block counters in place; totals[seg] = segment sum
__global__ void __launch_bounds__(256)
    k_compact_scan(int32_t *__restrict__ block_counts, int nb, int32_t *__restrict__ totals) {
  const int seg = blockIdx.x;
  int32_t *c = block_counts + (int64_t)seg * nb;
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int carry = 0;
  for (int base = 0; base < nb; base += 256) {
    int i = base + threadIdx.x;
    int v = (i < nb) ? c[i] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (i < nb) c[i] = carry + woff + x - v;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[seg] = carry;
}

template <class Pred, class Emit>
__global__ void __launch_bounds__(CP_THREADS)
    k_compact_scatter(Pred pred, Emit emit, int64_t n, const int32_t *__restrict__ block_offsets) {
  const int seg = blockIdx.y;
  const int nb = gridDim.x;
  const int64_t base = (int64_t)blockIdx.x * CP_TILE;
  __shared__ int wcnt[CP_THREADS / PH_WAVE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int running = block_offsets[(int64_t)seg * nb + blockIdx.x];
#pragma unroll 1
  for (int r = 0; r < CP_ROUNDS; ++r) {
    int64_t i = base + r * CP_THREADS + threadIdx.x;
    bool p = (i < n) && pred(seg, i);
    unsigned long long b = __ballot(p);
    int wtot = __popcll(b);
    int wpre = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave] = wtot;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wcnt[w];
    int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (i < n) emit(seg, i, p ? running + woff + wpre : -1);
    running += tot;
    __syncthreads();
  }
}

template <class Pred, class Emit>
static int compact_run(Pred pred, Emit emit, int64_t n, int segs, int32_t *totals, void *ws,
                       int64_t ws_bytes, hipStream_t st) {
  int64_t nb = (n + CP_TILE - 1) / CP_TILE;
  if (nb < 1) nb = 1;
  PH_REQUIRE(ws_bytes >= (int64_t)segs * nb * 4, "compaction workspace too small (%lld < %lld)",
             (long long)ws_bytes, (long long)(segs * nb * 4));
  int32_t *block_counts = (int32_t *)ws;
  dim3 grid((unsigned)nb, (unsigned)segs);
  hipLaunchKernelGGL(k_compact_count<Pred>, grid, dim3(CP_THREADS), 0, st, pred, n, block_counts);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_compact_scan, dim3(segs), dim3(256), 0, st, block_counts, (int)nb, totals);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_compact_scatter<Pred, Emit>), grid, dim3(CP_THREADS), 0, st, pred, emit, n,
                     block_counts);
  PH_LAUNCH_CHECK();
  return 0;
}

struct PredFlag {
  const uint8_t *flags;
  __device__ bool operator()(int, int64_t i) const { return flags[i] != 0; }
};
struct EmitRows {
  int32_t *keep_rows;
  int32_t *rank_of;  // may be null
  __device__ void operator()(int, int64_t i, int rank) const {
    if (rank >= 0) keep_rows[rank] = (int32_t)i;
    if (rank_of) rank_of[i] = rank;
  }
};

int ph_compact_flags(const uint8_t *flags, int64_t n, int32_t *keep_rows, int32_t *rank_of,
                     int32_t *n_keep, void *ws, int64_t ws_bytes, hipStream_t st) {
  return compact_run(PredFlag{flags}, EmitRows{keep_rows, rank_of}, n, 1, n_keep, ws, ws_bytes, st);
}

extern "C" int ph_mask_compact(const uint8_t *mask, int64_t n, int32_t *keep_rows, int32_t *n_keep,
                               void *ws, int64_t ws_bytes, ph_stream_t stream) {
  PH_REQUIRE(n >= 0 && n < 0x7FFFFFFF, "mask_compact: bad n");
  if (n == 0) {
    PH_CHECK_HIP(hipMemsetAsync(n_keep, 0, 4, ph_stream(stream)));
    return 0;
  }
  return ph_compact_flags(mask, n, keep_rows, nullptr, n_keep, ws, ws_bytes, ph_stream(stream));
}

// ---- hash insert --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_insert(const int4 *__restrict__ coords, int64_t n, unsigned long long *__restrict__ tkeys,
             int32_t *__restrict__ tvals, uint64_t mask, int32_t *__restrict__ row_slot, int32_t *__restrict__ status) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  if (status != nullptr && !ph_packable(c.x, c.y, c.z, c.w)) atomicOr(status, 2);   // the key would alias another voxel
  uint64_t key = ph_pack(c.x, c.y, c.z, c.w);
  uint64_t slot = ph_hash(key) & mask;
  for (;;) {
    unsigned long long prev = atomicCAS(&tkeys[slot], (unsigned long long)PH_EMPTY_KEY,
                                        (unsigned long long)key);
    if (prev == PH_EMPTY_KEY || prev == key) break;
    slot = (slot + 1) & mask;
  }
  atomicMin((unsigned int *)&tvals[slot], (unsigned int)i);     // first occurrence (unset slots hold 0xFFFFFFFF)
  if (row_slot) row_slot[i] = (int32_t)slot;
}

struct PredFirst {
  const int32_t *tvals;
  const int32_t *row_slot;
  __device__ bool operator()(int, int64_t i) const { return tvals[row_slot[i]] == (int32_t)i; }
};

__global__ void __launch_bounds__(256)
    k_assign_rank(int64_t n, const int32_t *__restrict__ rank_of,
                  const int32_t *__restrict__ row_slot, int32_t *__restrict__ tvals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = rank_of[i];
  if (r >= 0) tvals[row_slot[i]] = r;
}

__global__ void __launch_bounds__(256)
    k_row2uniq(int64_t n, const int32_t *__restrict__ row_slot, const int32_t *__restrict__ tvals,
               int32_t *__restrict__ row2uniq) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  row2uniq[i] = tvals[row_slot[i]];
}

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

extern "C" int ph_map_insert(const int32_t *coords, int64_t n, uint64_t *tkeys, int32_t *tvals,
                             int64_t cap, int32_t *row2uniq, int32_t *uniq_rows, int32_t *n_uniq,
                             void *ws, int64_t ws_bytes, int32_t *status, ph_stream_t stream) {
  hipStream_t st = ph_stream(stream);
  PH_REQUIRE(n >= 0 && n < 0x3FFFFFFF, "map_insert: bad n=%lld", (long long)n);
  PH_REQUIRE(ph_is_pow2(cap) && cap >= 2 * n && cap >= 2, "map_insert: cap=%lld must be pow2 >= 2n",
             (long long)cap);
  // empty = all ones in both tables (keys: PH_EMPTY_KEY; values: the largest unsigned row, so that the unsigned atomicMin
  // of k_insert keeps the first occurrence): ONE fill when the caller laid the value table out right behind the keys
  if ((const char *)tvals == (const char *)tkeys + (size_t)cap * 8) {
    PH_CHECK_HIP(hipMemsetAsync(tkeys, 0xFF, (size_t)cap * 12, st));
  } else {
    PH_CHECK_HIP(hipMemsetAsync(tkeys, 0xFF, (size_t)cap * 8, st));
    PH_CHECK_HIP(hipMemsetAsync(tvals, 0xFF, (size_t)cap * 4, st));
  }
  if (n == 0) {
    if (n_uniq) PH_CHECK_HIP(hipMemsetAsync(n_uniq, 0, 4, st));
    return 0;
  }
  const uint64_t mask = (uint64_t)cap - 1;
  if (uniq_rows == nullptr) {
    // caller guarantees unique coordinates: rows keep their index.
    hipLaunchKernelGGL(k_insert, dim3(nblk(n, 256)), dim3(256), 0, st, (const int4 *)coords, n,
                       (unsigned long long *)tkeys, tvals, mask, (int32_t *)nullptr, status);
    PH_LAUNCH_CHECK();
    return 0;
  }
  PH_REQUIRE(row2uniq && n_uniq, "map_insert: row2uniq / n_uniq required with uniq_rows");
  PH_REQUIRE(ws_bytes >= ph_workspace_bytes(n), "map_insert: workspace too small");
  // carve: row_slot[n] | rank_of[n] | block counters
  int32_t *row_slot = (int32_t *)ws;
  int32_t *rank_of = row_slot + n;
  char *rest = (char *)(rank_of + n);
  int64_t rest_bytes = ws_bytes - 8 * n;
  hipLaunchKernelGGL(k_insert, dim3(nblk(n, 256)), dim3(256), 0, st, (const int4 *)coords, n,
                     (unsigned long long *)tkeys, tvals, mask, row_slot, status);
  PH_LAUNCH_CHECK();
  int rc = compact_run(PredFirst{tvals, row_slot}, EmitRows{uniq_rows, rank_of}, n, 1, n_uniq, rest,
                       rest_bytes, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_assign_rank, dim3(nblk(n, 256)), dim3(256), 0, st, n, rank_of, row_slot, tvals);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_row2uniq, dim3(nblk(n, 256)), dim3(256), 0, st, n, row_slot, tvals, row2uniq);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- find ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_find(const int4 *__restrict__ q, int64_t n, const uint64_t *__restrict__ tkeys,
           const int32_t *__restrict__ tvals, uint64_t mask, int32_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = q[i];
  out[i] = ph_packable(c.x, c.y, c.z, c.w) ? ph_find(tkeys, tvals, mask, ph_pack(c.x, c.y, c.z, c.w)) : -1;
}

extern "C" int ph_map_find(const int32_t *query, int64_t n, const uint64_t *tkeys,
                           const int32_t *tvals, int64_t cap, int32_t *out_rows,
                           ph_stream_t stream) {
  PH_REQUIRE(ph_is_pow2(cap), "map_find: cap must be pow2");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_find, dim3(nblk(n, 256)), dim3(256), 0, ph_stream(stream),
                     (const int4 *)query, n, tkeys, tvals, (uint64_t)cap - 1, out_rows);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- coordinate generation ----------------------------------------------------------------------
__device__ __forceinline__ int floor_to(int v, int ts) {
  int q = v / ts;
  if ((v % ts) != 0 && ((v < 0) != (ts < 0))) --q;
  return q * ts;
}

__global__ void __launch_bounds__(256)
    k_coords_floor(const int4 *__restrict__ in, int64_t n, int ts, int4 *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = in[i];
  out[i] = make_int4(c.x, floor_to(c.y, ts), floor_to(c.z, ts), floor_to(c.w, ts));
}

extern "C" int ph_coords_floor(const int32_t *coords, int64_t n, int32_t ts, int32_t *out,
                               ph_stream_t stream) {
  PH_REQUIRE(ts > 0, "coords_floor: ts must be > 0");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_coords_floor, dim3(nblk(n, 256)), dim3(256), 0, ph_stream(stream),
                     (const int4 *)coords, n, ts, (int4 *)out);
  PH_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256)
    k_coords_expand(const int4 *__restrict__ in, int64_t n, int ts, int4 *__restrict__ out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 8) return;
  int64_t i = t >> 3;
  int k = (int)(t & 7);
  int4 c = in[i];
  out[t] = make_int4(c.x, c.y + (k & 1) * ts, c.z + ((k >> 1) & 1) * ts, c.w + ((k >> 2) & 1) * ts);
}

extern "C" int ph_coords_expand(const int32_t *coords, int64_t n, int32_t ts_out, int32_t *out,
                                ph_stream_t stream) {
  PH_REQUIRE(ts_out > 0, "coords_expand: ts_out must be > 0");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_coords_expand, dim3(nblk(n * 8, 256)), dim3(256), 0, ph_stream(stream),
                     (const int4 *)coords, n, ts_out, (int4 *)out);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- neighbour table ----------------------------------------------------------------------------
struct NbrOffsets {
  int32_t d[PH_MAX_KVOL][3];
};

__global__ void __launch_bounds__(256)
    k_nbr_build(const int4 *__restrict__ out_coords, int64_t n_out,
                const uint64_t *__restrict__ tkeys, const int32_t *__restrict__ tvals,
                uint64_t mask, NbrOffsets off, int32_t *__restrict__ nbr) {
  const int k = blockIdx.y;
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  int4 c = out_coords[o];
  const int x = c.y + off.d[k][0], y = c.z + off.d[k][1], z = c.w + off.d[k][2];
  // a neighbour beyond the packable range cannot be in the map (insert flags such coordinates)
  nbr[(int64_t)k * n_out + o] = ph_packable(c.x, x, y, z) ? ph_find(tkeys, tvals, mask, ph_pack(c.x, x, y, z)) : -1;
}

extern "C" int ph_nbr_build(const int32_t *out_coords, int64_t n_out, const uint64_t *in_tkeys,
                            const int32_t *in_tvals, int64_t in_cap, const int32_t *h_offsets,
                            int32_t kvol, int32_t *nbr, ph_stream_t stream) {
  PH_REQUIRE(kvol >= 1 && kvol <= PH_MAX_KVOL, "nbr_build: kvol=%d out of range", kvol);
  PH_REQUIRE(ph_is_pow2(in_cap), "nbr_build: cap must be pow2");
  if (n_out == 0) return 0;
  NbrOffsets off;
  memset(&off, 0, sizeof(off));
  memcpy(off.d, h_offsets, sizeof(int32_t) * 3 * kvol);
  dim3 grid(nblk(n_out, 256), (unsigned)kvol);
  hipLaunchKernelGGL(k_nbr_build, grid, dim3(256), 0, ph_stream(stream), (const int4 *)out_coords,
                     n_out, in_tkeys, in_tvals, (uint64_t)in_cap - 1, off, nbr);
  PH_LAUNCH_CHECK();
  return 0;
}

// Same-map tables of a symmetric kernel (out_coords ARE the rows of the table, offset[K-1-k] = -offset[k], centre in the
// middle): row i is the neighbour of row o at offset k  <=>  row o is the neighbour of row i at offset K-1-k.  Only the first
// K/2 offsets are probed; every hit also writes its mirror entry (one writer per entry), the centre column is the identity.
__global__ void __launch_bounds__(256)
    k_nbr_build_same(const int4 *__restrict__ coords, int64_t n, const uint64_t *__restrict__ tkeys,
                     const int32_t *__restrict__ tvals, uint64_t mask, NbrOffsets off, int kvol, int32_t *__restrict__ nbr) {
  const int k = blockIdx.y;                 // 0 .. K/2
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  if (k == kvol / 2) {
    nbr[(int64_t)k * n + o] = (int32_t)o;
    return;
  }
  int4 c = coords[o];
  const int x = c.y + off.d[k][0], y = c.z + off.d[k][1], z = c.w + off.d[k][2];
  const int j = ph_packable(c.x, x, y, z) ? ph_find(tkeys, tvals, mask, ph_pack(c.x, x, y, z)) : -1;
  nbr[(int64_t)k * n + o] = j;
  if (j >= 0) nbr[(int64_t)(kvol - 1 - k) * n + j] = (int32_t)o;
}

extern "C" int ph_nbr_build_same(const int32_t *coords, int64_t n, const uint64_t *tkeys, const int32_t *tvals, int64_t cap,
                                 const int32_t *h_offsets, int32_t kvol, int32_t *nbr, ph_stream_t stream) {
  PH_REQUIRE(kvol >= 1 && kvol <= PH_MAX_KVOL && (kvol & 1), "nbr_build_same: odd kernel volume <= %d expected, got %d", PH_MAX_KVOL, kvol);
  PH_REQUIRE(ph_is_pow2(cap), "nbr_build_same: cap must be pow2");
  for (int k = 0; k < kvol; ++k)
    for (int a = 0; a < 3; ++a)
      PH_REQUIRE(h_offsets[3 * k + a] == -h_offsets[3 * (kvol - 1 - k) + a], "nbr_build_same: offsets are not symmetric");
  if (n == 0) return 0;
  hipStream_t st = ph_stream(stream);
  NbrOffsets off;
  memset(&off, 0, sizeof(off));
  memcpy(off.d, h_offsets, sizeof(int32_t) * 3 * kvol);
  const int half = kvol / 2;
  if (half > 0)   // mirror half: -1 unless a hit writes it
    PH_CHECK_HIP(hipMemsetAsync(nbr + (int64_t)(half + 1) * n, 0xFF, (size_t)half * (size_t)n * 4, st));
  dim3 grid(nblk(n, 256), (unsigned)(half + 1));
  hipLaunchKernelGGL(k_nbr_build_same, grid, dim3(256), 0, st, (const int4 *)coords, n, tkeys, tvals, (uint64_t)cap - 1, off,
                     kvol, nbr);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- COO kernel map ------------------------------------------------------------------------------
struct PredNbr {
  const int32_t *nbr;
  int64_t n_out;
  __device__ bool operator()(int seg, int64_t i) const { return nbr[(int64_t)seg * n_out + i] >= 0; }
};
struct EmitPairs {
  const int32_t *nbr;
  int64_t n_out;
  int32_t *pairs_in;
  int32_t *pairs_out;
  __device__ void operator()(int seg, int64_t i, int rank) const {
    if (rank < 0) return;
    int64_t base = (int64_t)seg * n_out;
    pairs_in[base + rank] = nbr[base + i];
    pairs_out[base + rank] = (int32_t)i;
  }
};

extern "C" int ph_kmap_compact(const int32_t *nbr, int32_t kvol, int64_t n_out, int32_t *pairs_in,
                               int32_t *pairs_out, int32_t *counts, void *ws, int64_t ws_bytes,
                               ph_stream_t stream) {
  PH_REQUIRE(kvol >= 1 && kvol <= PH_MAX_KVOL, "kmap_compact: kvol=%d out of range", kvol);
  if (n_out == 0) {
    PH_CHECK_HIP(hipMemsetAsync(counts, 0, 4 * kvol, ph_stream(stream)));
    return 0;
  }
  return compact_run(PredNbr{nbr, n_out}, EmitPairs{nbr, n_out, pairs_in, pairs_out}, n_out, kvol,
                     counts, ws, ws_bytes, ph_stream(stream));
}


// ---- row lists of a one-pair-per-row kernel map (generative transposed convolutions) -----------------------------------------
// From the COO form (ph_kmap_compact: pairs_in / pairs_out [K, n_out], counts [K]) to one padded list: the pairs of offset k
// occupy positions [off_k, off_k + counts[k]), off_k = sum_{j<k} roundup128(counts[j]); padding holds -1; tile t (128
// positions) belongs to offset tile_k[t] (-1 beyond the last tile).  Everything is sized by upper bounds, the counts stay
// on the device.
__global__ void __launch_bounds__(256)
    k_rowlist_pack(const int32_t *__restrict__ pairs_in, const int32_t *__restrict__ pairs_out, const int32_t *__restrict__ counts,
                   int kvol, int64_t n_out, int32_t *__restrict__ rl_in, int32_t *__restrict__ rl_out,
                   int32_t *__restrict__ tile_k, int64_t cap, int64_t tcap, int32_t *__restrict__ status) {
  __shared__ int64_t off[PH_MAX_KVOL + 1];
  if (threadIdx.x == 0) {
    int64_t o = 0, pairs = 0;
    for (int k = 0; k < kvol; ++k) {
      off[k] = o;
      o += ((int64_t)counts[k] + 127) / 128 * 128;
      pairs += counts[k];
    }
    off[kvol] = o;
    // the caller's promise "exactly one pair per output row": a map that breaks it would leave output rows unwritten
    if (status != nullptr && blockIdx.x == 0 && pairs != n_out) atomicOr(status, 32);
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += stride) {
    int k = 0;
    while (k < kvol && i >= off[k + 1]) ++k;
    int vi = -1, vo = -1;
    if (k < kvol) {
      const int64_t j = i - off[k];
      if (j < counts[k]) {
        vi = pairs_in[(int64_t)k * n_out + j];
        vo = pairs_out[(int64_t)k * n_out + j];
      }
    }
    rl_in[i] = vi;
    rl_out[i] = vo;
    if ((i & 127) == 0 && (i >> 7) < tcap) tile_k[i >> 7] = k < kvol ? k : -1;
  }
}

extern "C" int ph_rowlist_pack(const int32_t *pairs_in, const int32_t *pairs_out, const int32_t *counts, int32_t kvol,
                               int64_t n_out, int32_t *rl_in, int32_t *rl_out, int32_t *tile_k, int64_t cap, int64_t tcap,
                               int32_t *status, ph_stream_t stream) {
  PH_REQUIRE(kvol >= 1 && kvol <= PH_MAX_KVOL, "rowlist_pack: kvol=%d out of range", kvol);
  PH_REQUIRE(cap % 128 == 0 && cap >= n_out + (int64_t)kvol * 127 - (kvol * 127) % 128 && tcap * 128 >= cap,
             "rowlist_pack: list capacity %lld too small for %lld rows and %d offsets", (long long)cap, (long long)n_out, kvol);
  if (cap == 0) return 0;
  int64_t blocks = (cap + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_rowlist_pack, dim3((unsigned)blocks), dim3(256), 0, ph_stream(stream), pairs_in, pairs_out, counts, kvol,
                     n_out, rl_in, rl_out, tile_k, cap, tcap, status);
  PH_LAUNCH_CHECK();
  return 0;
}
