// Input stage on the device: the voxel max of `CylinderFeat`
// (pasco/models/unet3d_sparse_v2.py:53-86: sorted unique of the points' (subnet, voxel) rows + torch_scatter.scatter_max)
// and the MIMO channel concatenation of `Augmenter.merge` (pasco/models/augmenter.py:13-27: dense -> cat -> ME.to_sparse)
// as ONE sort-free pass over the points:
//
//   * the merged tensor's rows are the occupied sites of the points' bounding box in lexicographic (x, y, z) order - what
//     ME.to_sparse enumerates.  A byte flag per site (k_pts_mark) + the stable compaction the coordinate maps already use
//     (ph_compact_flags: wave64 ballot / popcount prefix) give every occupied site its row: no sort, no hash table;
//   * the points of one (subnet, voxel) cell are chained through ONE integer exchange per point (k_pts_link: head[cell] <-
//     point, next[point] <- old head); a wave per merged row then walks its cells' chains (two points on average), takes the
//     max with 16-byte loads and writes the row once, zeros where a subnet has no point in the voxel - what the dense detour
//     would hold (k_cells_max).  A float atomic max per (point, channel) was measured first: 24 M device-scope atomics =
//     567 us at S10 against ~50 us for the chained form;
//   * an all-zero row (ME.to_sparse would drop it) cannot be removed without a host read: k_cells_max raises status bit 3
//     instead and the caller redoes the stage on its general path (measure-zero event for real features).
//
// All of it is HBM bound integer + max work: one thread per point, one wave per merged row, 16-byte accesses.
#include <limits.h>

#include "ph_common.h"

// ---- bounds ---------------------------------------------------------------------------------------------------------
__global__ void k_pts_bounds_init(int32_t *out6) {
  if (threadIdx.x < 3) out6[threadIdx.x] = INT_MAX;
  else if (threadIdx.x < 6) out6[threadIdx.x] = INT_MIN;
}

__global__ void __launch_bounds__(256) k_pts_bounds(const int64_t *__restrict__ xyz, int64_t n, int32_t *__restrict__ out6) {
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int64_t v64 = xyz[i * 3 + a];
      const int v = v64 < INT_MIN ? INT_MIN : (v64 > INT_MAX ? INT_MAX : (int)v64);
      lo[a] = v < lo[a] ? v : lo[a];
      hi[a] = v > hi[a] ? v : hi[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int l2 = __shfl_xor(lo[a], off), h2 = __shfl_xor(hi[a], off);
      lo[a] = l2 < lo[a] ? l2 : lo[a];
      hi[a] = h2 > hi[a] ? h2 : hi[a];
    }
  }
  // one set of six atomics per WORKGROUP (a first version had one per wave: 36 k same-address atomics = 0.4 ms at S10)
  __shared__ int red[4][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) red[wave][a] = lo[a], red[wave][3 + a] = hi[a];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    int v = red[0][a];
    for (int w = 1; w < 4; ++w) v = a < 3 ? (red[w][a] < v ? red[w][a] : v) : (red[w][a] > v ? red[w][a] : v);
    if (a < 3) atomicMin(&out6[a], v);
    else atomicMax(&out6[a], v);
  }
}

extern "C" int ph_points_bounds(const int64_t *xyz, int64_t n, int32_t *out6, ph_stream_t stream) {
  PH_REQUIRE(n >= 0 && out6 != nullptr, "points_bounds: bad arguments");
  hipStream_t st = ph_stream(stream);
  hipLaunchKernelGGL(k_pts_bounds_init, dim3(1), dim3(64), 0, st, out6);
  if (n > 0) {
    int64_t grid = (n + 255) / 256;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(k_pts_bounds, dim3((unsigned)grid), dim3(256), 0, st, xyz, n, out6);
  }
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- site of a point ------------------------------------------------------------------------------------------------
struct PtsGrid {
  int lo[3];
  int dim[3];
};

__device__ __forceinline__ int64_t pts_site(const int64_t *__restrict__ xyz, int64_t i, const PtsGrid &g) {
  const int64_t x = xyz[i * 3] - g.lo[0], y = xyz[i * 3 + 1] - g.lo[1], z = xyz[i * 3 + 2] - g.lo[2];
  if ((uint64_t)x >= (uint64_t)g.dim[0] || (uint64_t)y >= (uint64_t)g.dim[1] || (uint64_t)z >= (uint64_t)g.dim[2]) return -1;
  return (x * g.dim[1] + y) * g.dim[2] + z;
}

static int pts_grid(const int32_t *h_lo3, const int32_t *h_dims3, PtsGrid &g) {
  int64_t vol = 1;
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = h_lo3[a];
    g.dim[a] = h_dims3[a];
    if (g.dim[a] <= 0) return 1;
    vol *= g.dim[a];
  }
  return vol < 0x7FFFFFFF ? 0 : 1;
}

__global__ void __launch_bounds__(256) k_pts_mark(const int64_t *__restrict__ xyz, int64_t n, PtsGrid g, uint8_t *__restrict__ flags,
                                                  int32_t *__restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = pts_site(xyz, i, g);
  if (s >= 0) flags[s] = 1;
  else if (status != nullptr) atomicOr(status, 8);     // a point outside the box the caller gave: the stage must be redone
}

extern "C" int ph_points_mark(const int64_t *xyz, int64_t n, const int32_t *h_lo3, const int32_t *h_dims3, uint8_t *flags,
                              int32_t *status, ph_stream_t stream) {
  PtsGrid g;
  PH_REQUIRE(n >= 0 && pts_grid(h_lo3, h_dims3, g) == 0, "points_mark: bad grid (volume must stay below 2^31 sites)");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_pts_mark, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ph_stream(stream), xyz, n, g, flags, status);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- stable compaction with the rank of every kept element (row of every occupied site) -------------------------------
extern "C" int ph_mask_compact_rank(const uint8_t *mask, int64_t n, int32_t *keep_rows, int32_t *rank_of, int32_t *n_keep,
                                    void *ws, int64_t ws_bytes, ph_stream_t stream) {
  PH_REQUIRE(n >= 0 && n < 0x7FFFFFFF, "mask_compact_rank: bad n");
  if (n == 0) {
    PH_CHECK_HIP(hipMemsetAsync(n_keep, 0, 4, ph_stream(stream)));
    return 0;
  }
  return ph_compact_flags(mask, n, keep_rows, rank_of, n_keep, ws, ws_bytes, ph_stream(stream));
}

// ---- chains of the points of every (merged row, subnet) cell -----------------------------------------------------------------
struct PtsStarts {
  int64_t s[9];      // first point of subnet b (s[m] = n)
  int m;
};

__global__ void __launch_bounds__(256) k_pts_link(const int64_t *__restrict__ xyz, int64_t n, PtsStarts st, PtsGrid g,
                                                  const int32_t *__restrict__ rank_of, int32_t *__restrict__ head,
                                                  int32_t *__restrict__ next) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = pts_site(xyz, i, g);
  if (s < 0) {
    next[i] = -1;
    return;
  }
  int b = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k) b += (k < st.m && i >= st.s[k]) ? 1 : 0;
  next[i] = atomicExch(&head[(int64_t)rank_of[s] * st.m + b], (int32_t)i);
}

extern "C" int ph_points_link(const int64_t *xyz, int64_t n, const int64_t *h_starts, int32_t m, const int32_t *h_lo3,
                              const int32_t *h_dims3, const int32_t *rank_of, int32_t *head, int32_t *next, ph_stream_t stream) {
  PtsGrid g;
  PH_REQUIRE(n >= 0 && n < 0x7FFFFFFF && m >= 1 && m <= 8, "points_link: needs 1 <= m <= 8 and n < 2^31");
  PH_REQUIRE(pts_grid(h_lo3, h_dims3, g) == 0, "points_link: bad grid");
  if (n == 0) return 0;
  PtsStarts st;
  st.m = m;
  for (int k = 0; k <= 8; ++k) st.s[k] = k <= m ? h_starts[k] : n;
  hipLaunchKernelGGL(k_pts_link, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ph_stream(stream), xyz, n, st, g, rank_of, head, next);
  PH_LAUNCH_CHECK();
  return 0;
}

// ---- max over every cell's chain, merged row written once, coordinates, all-zero-row flag ------------------------------
__global__ void __launch_bounds__(256) k_cells_max(const float *__restrict__ h, int c, const int32_t *__restrict__ head,
                                                   const int32_t *__restrict__ next, int64_t v, int m, const int32_t *__restrict__ sites,
                                                   PtsGrid g, float *__restrict__ out, int32_t *__restrict__ coords,
                                                   int32_t *__restrict__ status) {
  // one wave per merged row; lane -> (subnet b, 4-channel run) pairs of the row's m * c channels
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= v) return;
  const int quads = c >> 2;
  bool nz = false;
  for (int q = lane; q < m * quads; q += 64) {
    const int b = q / quads, cq = q - b * quads;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);       // no point of subnet b in this voxel: the dense detour holds 0
    int p = head[row * m + b];
    if (p >= 0) {
      acc = *reinterpret_cast<const float4 *>(h + (int64_t)p * c + 4 * cq);
      for (p = next[p]; p >= 0; p = next[p]) {
        const float4 x = *reinterpret_cast<const float4 *>(h + (int64_t)p * c + 4 * cq);
        acc.x = x.x > acc.x ? x.x : acc.x;
        acc.y = x.y > acc.y ? x.y : acc.y;
        acc.z = x.z > acc.z ? x.z : acc.z;
        acc.w = x.w > acc.w ? x.w : acc.w;
      }
    }
    nz |= (acc.x != 0.f) | (acc.y != 0.f) | (acc.z != 0.f) | (acc.w != 0.f);
    *reinterpret_cast<float4 *>(out + (row * m + b) * c + 4 * cq) = acc;
  }
  const bool any = __ballot(nz) != 0ull;
  if (lane == 0) {
    const int s = sites[row];
    const int z = s % g.dim[2], xy = s / g.dim[2];
    const int y = xy % g.dim[1], x = xy / g.dim[1];
    *reinterpret_cast<int4 *>(coords + row * 4) = make_int4(0, x + g.lo[0], y + g.lo[1], z + g.lo[2]);
    if (!any && status != nullptr) atomicOr(status, 8);      // ME.to_sparse would drop this row: the caller redoes the stage
  }
}

extern "C" int ph_cells_max(const float *h, int32_t c, const int32_t *head, const int32_t *next, int64_t v, int32_t m,
                            const int32_t *sites, const int32_t *h_lo3, const int32_t *h_dims3, float *out, int32_t *coords,
                            int32_t *status, ph_stream_t stream) {
  PtsGrid g;
  PH_REQUIRE(v >= 0 && c > 0 && c % 4 == 0 && m >= 1 && m <= 8, "cells_max: needs c %% 4 == 0 and 1 <= m <= 8");
  PH_REQUIRE(pts_grid(h_lo3, h_dims3, g) == 0, "cells_max: bad grid");
  PH_REQUIRE((((uintptr_t)h | (uintptr_t)out | (uintptr_t)coords) & 15) == 0, "cells_max: 16-byte alignment");
  if (v == 0) return 0;
  const int64_t threads = v * 64;
  hipLaunchKernelGGL(k_cells_max, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ph_stream(stream), h, c, head, next, v, m,
                     sites, g, out, coords, status);
  PH_LAUNCH_CHECK();
  return 0;
}
