// Panoptic post-processing on the device (include/pasco_hip.h panop_*; reference pasco/models/helper.py:91-303).
//
// The reference walks the kept queries in a Python loop with ~6 host reads per query and materialises [K, 256, 256, 32]
// dense tensors.  Here: one tiny kernel classifies the queries, one pass over the [n, q] mask probabilities finds every
// voxel's winning query and the per-query areas (a wave per row: lanes read consecutive columns, 256-byte runs), and one
// element-wise pass writes the per-voxel outputs after every workgroup has replayed the reference's sequential walk over the
// <= 128 kept queries from the areas.  No host read in between: the segment table comes back with the results.
#include "ph_common.h"

#define PANOP_QMAX 128

// ---- queries -----------------------------------------------------------------------------------
// qtab int32 [4][128]: 0 kidx[q] (rank among the kept queries, -1 = not kept), 1 kq[k] (query of kept k), 2 class[q],
// 3 prob[q] (float bits); nk[0] = K
__global__ void __launch_bounds__(PANOP_QMAX)
    k_panop_queries(const float *__restrict__ qp, int q, int c1, float thr, int32_t *__restrict__ qtab, int32_t *__restrict__ nk) {
  __shared__ int s_keep[PANOP_QMAX];
  const int t = threadIdx.x;
  int label = 0;
  float prob = 0.f;
  int keep = 0;
  if (t < q) {
    const float *row = qp + (int64_t)t * c1;
    prob = row[0];
    for (int c = 1; c < c1; ++c) {          // first maximum, like torch.max
      const float v = row[c];
      if (v > prob) prob = v, label = c;
    }
    keep = (label != 0 && label != c1 - 1 && prob > thr) ? 1 : 0;
  }
  s_keep[t] = keep;
  __syncthreads();
  int rank = 0;
  for (int i = 0; i < t; ++i) rank += s_keep[i];
  qtab[0 * PANOP_QMAX + t] = keep ? rank : -1;
  qtab[2 * PANOP_QMAX + t] = label;
  qtab[3 * PANOP_QMAX + t] = __float_as_int(prob);
  if (keep) qtab[1 * PANOP_QMAX + rank] = t;
  if (t == PANOP_QMAX - 1) nk[0] = rank + keep;
}

// ---- per-voxel competition ---------------------------------------------------------------------
// one wave per row (grid-stride over the rows): lane l holds columns l and l + 64
__global__ void __launch_bounds__(256)
    k_panop_argmax(const float *__restrict__ masks, int64_t n, int q, const int32_t *__restrict__ qtab, float occ_thr,
                   int32_t *__restrict__ winner, uint8_t *__restrict__ own, float *__restrict__ conf,
                   float *__restrict__ vunc, int32_t *__restrict__ areas) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int c0 = lane, c1 = lane + 64;
  const int k0 = c0 < q ? qtab[c0] : -1;
  const int k1 = c1 < q ? qtab[c1] : -1;
  const float p0 = c0 < q ? __int_as_float(qtab[3 * PANOP_QMAX + c0]) : 0.f;
  const float p1 = c1 < q ? __int_as_float(qtab[3 * PANOP_QMAX + c1]) : 0.f;
  int orig0 = 0, orig1 = 0, area0 = 0, area1 = 0;    // lane-private counts: original area of its columns, won area of kept l / l + 64
  for (int64_t r = wave; r < n; r += nwaves) {
    const float *row = masks + r * q;
    const float m0 = k0 >= 0 ? row[c0] : 0.f;
    const float m1 = k1 >= 0 ? row[c1] : 0.f;
    orig0 += (k0 >= 0 && m0 >= occ_thr) ? 1 : 0;
    orig1 += (k1 >= 0 && m1 >= occ_thr) ? 1 : 0;
    const float v0 = p0 * m0, v1 = p1 * m1;
    // best (value, kept index) of the lane: kept indices ascend with the column, so "first maximum" = smallest index
    float bv = -1.f, bm = 0.f;
    int bk = 0x7fffffff;
    if (k0 >= 0) bv = v0, bk = k0, bm = m0;
    if (k1 >= 0 && (v1 > bv)) bv = v1, bk = k1, bm = m1;
    float sm = (k0 >= 0 ? m0 : 0.f) + (k1 >= 0 ? m1 : 0.f);
    float sc = (k0 >= 0 ? v0 : 0.f) + (k1 >= 0 ? v1 : 0.f);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(bv, off);
      const int ok = __shfl_xor(bk, off);
      const float om = __shfl_xor(bm, off);
      // NaN-free inputs (probabilities); ties go to the smaller kept index
      if (ov > bv || (ov == bv && ok < bk)) bv = ov, bk = ok, bm = om;
      sm += __shfl_xor(sm, off);
      sc += __shfl_xor(sc, off);
    }
    const bool any = bk != 0x7fffffff;
    const bool mine = any && bm >= occ_thr;
    if (mine) {
      if ((bk & 63) == lane) {
        if (bk < 64) ++area0; else ++area1;
      }
    }
    if (lane == 0) {
      winner[r] = any ? bk : -1;
      own[r] = mine ? 1 : 0;
      conf[r] = any ? bm / (sm + 1e-8f) : 0.f;
      vunc[r] = any ? bv / sc : 0.f;
    }
  }
  // kept index k lives in lane k & 63, slot k >> 6; original areas are per COLUMN -> translate to the kept index
  if (area0) atomicAdd(&areas[lane], area0);
  if (area1) atomicAdd(&areas[64 + lane], area1);
  if (orig0) atomicAdd(&areas[PANOP_QMAX + k0], orig0);
  if (orig1) atomicAdd(&areas[PANOP_QMAX + k1], orig1);
}

// ---- segments + per-voxel outputs ----------------------------------------------------------------
struct PanopWriteArgs {
  int64_t n;
  const int32_t *winner;
  const uint8_t *own;
  const float *conf, *vunc;
  const int32_t *areas, *qtab, *nk;
  double overlap_thr;
  uint64_t thing_mask;      // bit c set: class c is a "thing"
  int32_t *panoptic, *semantic;
  float *ins_unc, *vox_conf, *vox_unc;
  int32_t *seg;             // [5][128]: id, isthing, category, query id; seg[4 * 128] = number of segments
};

__global__ void __launch_bounds__(256) k_panop_write(PanopWriteArgs a) {
  __shared__ int s_seg[PANOP_QMAX];       // segment id written for voxels won by kept k (0 = nothing)
  __shared__ int s_full[PANOP_QMAX];      // the query opened a segment: class / confidences are written too
  __shared__ int s_cls[PANOP_QMAX];
  __shared__ float s_prob[PANOP_QMAX];
  if (threadIdx.x == 0) {
    // the reference's walk over the kept queries (helper.py:188-250): overlap test, stuff segments of one class merge
    const int K = a.nk[0];
    int current = 0;
    int stuff_seg[64];
    for (int c = 0; c < 64; ++c) stuff_seg[c] = 0;
    const bool emit = blockIdx.x == 0 && a.seg != nullptr;
    for (int k = 0; k < K; ++k) {
      const int qid = a.qtab[1 * PANOP_QMAX + k];
      const int cls = a.qtab[2 * PANOP_QMAX + qid];
      const float prob = __int_as_float(a.qtab[3 * PANOP_QMAX + qid]);
      const int ma = a.areas[k], oa = a.areas[PANOP_QMAX + k];
      s_seg[k] = 0, s_full[k] = 0, s_cls[k] = cls, s_prob[k] = prob;
      if (!(ma > 0 && oa > 0) || (double)ma / (double)oa < a.overlap_thr) continue;
      const bool isthing = cls < 64 && ((a.thing_mask >> cls) & 1ull);
      if (!isthing) {
        if (cls < 64 && stuff_seg[cls] != 0) {
          s_seg[k] = stuff_seg[cls];        // merged: only the panoptic id is written
          continue;
        }
        if (cls < 64) stuff_seg[cls] = current + 1;
      }
      ++current;
      s_seg[k] = current, s_full[k] = 1;
      if (emit) {
        a.seg[0 * PANOP_QMAX + current - 1] = current;
        a.seg[1 * PANOP_QMAX + current - 1] = isthing ? 1 : 0;
        a.seg[2 * PANOP_QMAX + current - 1] = cls;
        a.seg[3 * PANOP_QMAX + current - 1] = qid;
      }
    }
    if (emit) a.seg[4 * PANOP_QMAX] = current;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int w = a.winner[i];
  const bool mine = a.own[i] != 0 && w >= 0;
  const bool hit = mine && s_seg[w] != 0;
  const bool full = mine && s_full[w] != 0;
  a.panoptic[i] = hit ? s_seg[w] : 0;
  a.semantic[i] = full ? s_cls[w] : 0;
  a.ins_unc[i] = full ? s_prob[w] : 0.f;
  a.vox_conf[i] = full ? a.conf[i] : 0.f;
  a.vox_unc[i] = full ? a.vunc[i] : 0.f;
}

extern "C" int ph_panop_queries(const float *qp, int32_t q, int32_t c1, float thr, int32_t *qtab, int32_t *nk,
                                ph_stream_t stream) {
  PH_REQUIRE(q >= 1 && q <= PANOP_QMAX && c1 >= 2 && c1 <= 64, "panop_queries: 1 <= q <= 128 queries, 2 <= c1 <= 64");
  hipLaunchKernelGGL(k_panop_queries, dim3(1), dim3(PANOP_QMAX), 0, ph_stream(stream), qp, q, c1, thr, qtab, nk);
  PH_LAUNCH_CHECK();
  return 0;
}

extern "C" int ph_panop_argmax(const float *masks, int64_t n, int32_t q, const int32_t *qtab, float occ_thr, int32_t *winner,
                               uint8_t *own, float *conf, float *vunc, int32_t *areas, ph_stream_t stream) {
  PH_REQUIRE(q >= 1 && q <= PANOP_QMAX, "panop_argmax: 1 <= q <= 128 queries");
  if (n == 0) return 0;
  int64_t blocks = (n + 3) / 4;                      // four waves per workgroup, a row per wave and trip
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_panop_argmax, dim3((unsigned)blocks), dim3(256), 0, ph_stream(stream), masks, n, q, qtab, occ_thr,
                     winner, own, conf, vunc, areas);
  PH_LAUNCH_CHECK();
  return 0;
}

extern "C" int ph_panop_write(int64_t n, const int32_t *winner, const uint8_t *own, const float *conf, const float *vunc,
                              const int32_t *areas, const int32_t *qtab, const int32_t *nk, double overlap_thr,
                              uint64_t thing_mask, int32_t *panoptic, int32_t *semantic, float *ins_unc, float *vox_conf,
                              float *vox_unc, int32_t *seg, ph_stream_t stream) {
  PanopWriteArgs a{n, winner, own, conf, vunc, areas, qtab, nk, overlap_thr, thing_mask, panoptic, semantic, ins_unc, vox_conf,
                   vox_unc, seg};
  const int64_t blocks = n > 0 ? (n + 255) / 256 : 1;   // n = 0: the segment table is still written
  hipLaunchKernelGGL(k_panop_write, dim3((unsigned)blocks), dim3(256), 0, ph_stream(stream), a);
  PH_LAUNCH_CHECK();
  return 0;
}
