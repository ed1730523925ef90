// Masked cross-attention of a few queries (<= 128) over very many voxel keys, fp32, gfx950.
//
// Serves the CrossAttentionLayer of PaSCo's mask transformer (reference:
// pasco/models/transformer/blocks.py:73-92 called at transformer_predictor_v2.py:167-173): per
// subnet b and head h, O = softmax(Q K^T + mask) V with Q [100 x 48], K,V [N x 48], N up to ~4e5.
// The reference materialises the [B*8, 100, N] score tensor; here every wave64 streams a
// contiguous run of 16-key tiles with an online softmax and keeps everything in registers:
//
//   S^T[key][q] = sum_d K[key][d] Q[q][d]     v_mfma_f32_16x16x4_f32, A = K fragment straight from
//                                              global (float4 per lane), B = Q held in registers
//   P = exp(S^T - m)                           lane (q, g) holds keys 4g..4g+3 of its query: the
//                                              row max / row sum need two __shfl_xor steps only
//   O^T[d][q] += sum_key V[key][d] P[key][q]   A = V fragment from global, B = P registers as is
//
// The C/D layout of the 16x16 MFMA (col = lane & 15, row = 4*(lane >> 4) + reg) makes the S^T
// registers directly usable as the B operand of the second product - no LDS, no barrier, no
// cross-lane traffic for P.  Work is split over N (flash-decoding style): each wave writes a
// partial (m, l, O) and k_attn_merge combines them.  Mask = 1 bit per (key, query), 16 B per key.
#include <stdlib.h>

#include "ph_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AttnArgs {
  const float *q;        // [B, H, Qn, Dh]  (already scaled by 1/sqrt(Dh))
  const float *k;        // [B, N, H*Dh]
  const float *v;        // [B, N, H*Dh]
  const uint32_t *bits;  // [B, N, 4] allow bits (bit q of the 128-bit word) or null
  const uint32_t *any;   // [B, 4] OR over keys of bits (queries with no allowed key attend everywhere) or null
  float *part;           // [B*H*splits][QT*16][Dh+4]  (O row, then m, l, 2 pad)
  float *out;            // [B, Qn, H*Dh]
  int64_t n;
  int B, H, Qn, Dh;
  int splits;            // waves per (b, h)
  int tiles_per_wave;    // 16-key tiles per wave
  int qp;                // queries per partial record (64 per query block of the launch)
};

// One wave = one key range x one block of QT query tiles (blockIdx.y).  QT = 4 keeps the wave at ~150 VGPRs
// (3 waves per SIMD; the 7-tile variant needed 356 and ran one wave per SIMD with nothing to hide latency).
template <int QT, int DT>
__global__ void __launch_bounds__(256) k_attn_cross(AttnArgs a) {
  constexpr int DH = DT * 16;
  const int qt0 = (int)blockIdx.y * QT;   // first query tile of this block
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int64_t w = (int64_t)blockIdx.x * 4 + wave_in_block;
  const int64_t total = (int64_t)a.B * a.H * a.splits;
  if (w >= total) return;
  const int bh = (int)(w / a.splits);
  const int split = (int)(w - (int64_t)bh * a.splits);
  const int b = bh / a.H, h = bh - b * a.H;
  const int qi = lane & 15, g = lane >> 4;
  const int D = a.H * DH;

  // Q fragments: qreg[qt][j][c] = Q[b, h, qt*16 + qi, 16j + 4g + c]
  f32x4 qreg[QT][DT];
  unsigned force[QT];  // query attends everywhere (no mask given, or nothing allowed anywhere)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qq = (qt0 + qt) * 16 + qi;
    const bool qv = qq < a.Qn;
#pragma unroll
    for (int j = 0; j < DT; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (qv) v = *reinterpret_cast<const f32x4 *>(a.q + (((int64_t)b * a.H + h) * a.Qn + qq) * DH + 16 * j + 4 * g);
      qreg[qt][j] = v;
    }
    unsigned f = 1u;
    if (a.bits != nullptr) {
      f = 0u;
      if (a.any != nullptr && qv) f = ((a.any[b * 4 + (qq >> 5)] >> (qq & 31)) & 1u) ? 0u : 1u;
    }
    force[qt] = f;
  }

  float m[QT], l[QT];
  f32x4 acc[QT][DT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m[qt] = -INFINITY;
    l[qt] = 0.f;
#pragma unroll
    for (int j = 0; j < DT; ++j) acc[qt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int64_t t0 = (int64_t)split * a.tiles_per_wave;
  const int64_t ntile_all = (a.n + 15) / 16;
  int64_t t1 = t0 + a.tiles_per_wave;
  if (t1 > ntile_all) t1 = ntile_all;
  const float *kb = a.k + (int64_t)b * a.n * D + h * DH;
  const float *vb = a.v + (int64_t)b * a.n * D + h * DH;

  for (int64_t t = t0; t < t1; ++t) {
    const int64_t nb = t * 16;
    // ---- K fragment: lane (key = qi, g) loads dims 16j + 4g .. +3 ---------------------------------
    f32x4 kf[DT];
    {
      const int64_t key = nb + qi;
      const bool kv = key < a.n;
#pragma unroll
      for (int j = 0; j < DT; ++j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kv) v = *reinterpret_cast<const f32x4 *>(kb + key * D + 16 * j + 4 * g);
        kf[j] = v;
      }
    }
    // ---- mask words of the 4 keys this lane's S registers refer to ---------------------------------
    uint4 mb[4];
    bool kin[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t key = nb + 4 * g + r;
      kin[r] = key < a.n;
      mb[r] = make_uint4(0u, 0u, 0u, 0u);
      if (a.bits != nullptr && kin[r]) mb[r] = *reinterpret_cast<const uint4 *>(a.bits + ((int64_t)b * a.n + key) * 4);
    }
    // ---- V fragments: lane (dim = qi, g), step r -> V[nb + 4g + r][16*dt + qi] ----------------------
    float vf[DT][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t key = nb + 4 * g + r;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) vf[dt][r] = kin[r] ? vb[key * D + 16 * dt + qi] : 0.f;
    }

#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      // S^T tile [16 keys x 16 queries]
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < DT; ++j) {
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][0], qreg[qt][j][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][1], qreg[qt][j][1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][2], qreg[qt][j][2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j][3], qreg[qt][j][3], s, 0, 0, 0);
      }
      // mask: this lane's query is qt*16 + qi; register r is key nb + 4g + r
      const int qq = (qt0 + qt) * 16 + qi;
      const int wsel = qq >> 5, bsel = qq & 31;
      float sv[4];
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned word = wsel == 0 ? mb[r].x : (wsel == 1 ? mb[r].y : (wsel == 2 ? mb[r].z : mb[r].w));
        const bool ok = kin[r] && (force[qt] || ((word >> bsel) & 1u));
        sv[r] = ok ? s[r] : -INFINITY;
        tmax = fmaxf(tmax, sv[r]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m[qt], tmax);
      const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __expf(m[qt] - m_safe);   // m = -inf -> 0 (nothing accumulated yet)
      m[qt] = m_new;
      float p[4];
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = __expf(sv[r] - m_safe);
        psum += p[r];
      }
      l[qt] = l[qt] * alpha + psum;
      if (alpha != 1.f) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[qt][dt] *= alpha;
      }
      // O^T += V^T P
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dt][r], p[r], acc[qt][dt], 0, 0, 0);
      }
    }
  }

  // ---- partial result of this wave: per query m, l (summed over the 4 key groups), O[q][d] -----------
  float *pw = a.part + w * (int64_t)a.qp * (DH + 4);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float lt = l[qt];
    lt += __shfl_xor(lt, 16);
    lt += __shfl_xor(lt, 32);
    const int qq = (qt0 + qt) * 16 + qi;
    float *row = pw + (int64_t)qq * (DH + 4);
    if (g == 0) {
      row[DH] = m[qt];
      row[DH + 1] = lt;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      *reinterpret_cast<f32x4 *>(row + 16 * dt + 4 * g) = acc[qt][dt];  // dims 16dt + 4g + r
  }
}

// One workgroup per (b, h, 16-query tile): combine the per-wave partials.
template <int QT, int DT>
__global__ void __launch_bounds__(256) k_attn_merge(AttnArgs a) {
  constexpr int DH = DT * 16;
  const int QP = a.qp;
  constexpr int PS = DH + 4;
  __shared__ float Ms[16];
  __shared__ float Ls[16];
  const int bh = blockIdx.x;
  const int qt = blockIdx.y;
  const int b = bh / a.H, h = bh - b * a.H;
  const float *base = a.part + ((int64_t)bh * a.splits * QP + qt * 16) * PS;   // + s * QP * PS + ql * PS
  // 16 threads per query reduce (m, l) over the splits
  {
    const int ql = threadIdx.x >> 4, part = threadIdx.x & 15;
    float M = -INFINITY;
    for (int s = part; s < a.splits; s += 16) M = fmaxf(M, base[((int64_t)s * QP + ql) * PS + DH]);
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) M = fmaxf(M, __shfl_xor(M, d));
    const float Msafe = (M == -INFINITY) ? 0.f : M;
    float L = 0.f;
    for (int s = part; s < a.splits; s += 16) {
      const float *row = base + ((int64_t)s * QP + ql) * PS;
      L += row[DH + 1] * __expf(row[DH] - Msafe);
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) L += __shfl_xor(L, d);
    if (part == 0) {
      Ms[ql] = Msafe;
      Ls[ql] = L;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * DH; e += blockDim.x) {
    const int ql = e / DH, d = e - ql * DH;
    const int q = qt * 16 + ql;
    if (q >= a.Qn) continue;
    float o = 0.f;
    for (int s = 0; s < a.splits; ++s) {
      const float *row = base + ((int64_t)s * QP + ql) * PS;
      o += row[d] * __expf(row[DH] - Ms[ql]);
    }
    const float L = Ls[ql];
    a.out[((int64_t)b * a.Qn + q) * (a.H * DH) + h * DH + d] = L > 0.f ? o / L : 0.f;
  }
}

// vals [R, Qn] -> bits [R, 4] ; any[b, 4] |= bits (R = B * N, b = row / N).  One wave64 per row: two
// coalesced 256-byte loads, two ballots.
__global__ void __launch_bounds__(256)
    k_mask_pack(const float *__restrict__ vals, int64_t rows, int64_t n_per_b, int qn, int positive_only,
                uint32_t *__restrict__ bits, uint32_t *__restrict__ any) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < rows; row += nwaves) {
    const float *src = vals + row * qn;
    const float v0 = lane < qn ? src[lane] : 0.f;
    const float v1 = 64 + lane < qn ? src[64 + lane] : 0.f;
    const unsigned long long b0 = __ballot(positive_only ? v0 > 0.f : v0 != 0.f);
    const unsigned long long b1 = __ballot(positive_only ? v1 > 0.f : v1 != 0.f);
    if (lane < 4) {
      const uint32_t word = lane == 0 ? (uint32_t)b0 : lane == 1 ? (uint32_t)(b0 >> 32)
                          : lane == 2 ? (uint32_t)b1 : (uint32_t)(b1 >> 32);
      bits[row * 4 + lane] = word;
      if (any != nullptr && word != 0u) {
        uint32_t *dst = any + (row / n_per_b) * 4 + lane;
        if ((__hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & word) != word) atomicOr(dst, word);
      }
    }
  }
}

extern "C" int ph_attn_mask_pack(const float *vals, int64_t n, int32_t b, int32_t qn, int32_t positive_only,
                                 uint32_t *bits, uint32_t *any, ph_stream_t stream) {
  PH_REQUIRE(qn >= 1 && qn <= 128 && b >= 1 && n >= 0, "attn_mask_pack: bad shape");
  hipStream_t st = ph_stream(stream);
  if (any) PH_CHECK_HIP(hipMemsetAsync(any, 0, (size_t)b * 16, st));
  const int64_t rows = (int64_t)b * n;
  if (rows == 0) return 0;
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_mask_pack, dim3((unsigned)blocks), dim3(256), 0, st, vals, rows, n, qn, positive_only, bits,
                     any);
  PH_LAUNCH_CHECK();
  return 0;
}

// bits_out[o] = OR_k bits_in[nbr[k][o]]   (one thread per (o, word))
__global__ void __launch_bounds__(256)
    k_bits_orpool(const uint32_t *__restrict__ bin, const int32_t *__restrict__ nbr, int kvol, int64_t n_out,
                  uint32_t *__restrict__ bout) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * 4) return;
  const int64_t o = t >> 2;
  const int w = (int)(t & 3);
  uint32_t m = 0;
  for (int k = 0; k < kvol; ++k) {
    const int r = nbr[(int64_t)k * n_out + o];
    if (r >= 0) m |= bin[(int64_t)r * 4 + w];
  }
  bout[t] = m;
}

extern "C" int ph_bits_orpool(const uint32_t *bits_in, const int32_t *nbr, int32_t kvol, int64_t n_out,
                              uint32_t *bits_out, ph_stream_t stream) {
  PH_REQUIRE(kvol >= 1 && kvol <= PH_MAX_KVOL, "bits_orpool: bad kernel volume");
  if (n_out == 0) return 0;
  hipLaunchKernelGGL(k_bits_orpool, dim3((unsigned)((n_out * 4 + 255) / 256)), dim3(256), 0, ph_stream(stream),
                     bits_in, nbr, kvol, n_out, bits_out);
  PH_LAUNCH_CHECK();
  return 0;
}

// any[b][w] = OR over the n rows of batch b (wave OR-reduction, one atomicOr per wave)
__global__ void __launch_bounds__(256)
    k_bits_or_reduce(const uint32_t *__restrict__ bits, int64_t n, uint32_t *__restrict__ any) {
  const int b = blockIdx.y;
  const uint32_t *src = bits + (int64_t)b * n * 4;
  uint32_t m = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * 4; t += (int64_t)gridDim.x * blockDim.x)
    m |= src[t];   // t & 3 is constant per thread (stride is a multiple of 4)
  // lanes with equal (lane & 3) hold the same word
#pragma unroll
  for (int d = 4; d < 64; d <<= 1) m |= __shfl_xor(m, d);
  if ((threadIdx.x & 63) < 4 && m != 0u) atomicOr(any + b * 4 + (threadIdx.x & 3), m);
}

extern "C" int ph_bits_or_reduce(const uint32_t *bits, int64_t n, int32_t b, uint32_t *any, ph_stream_t stream) {
  PH_REQUIRE(b >= 1 && n >= 0, "bits_or_reduce: bad shape");
  hipStream_t st = ph_stream(stream);
  PH_CHECK_HIP(hipMemsetAsync(any, 0, (size_t)b * 16, st));
  if (n == 0) return 0;
  int64_t blocks = (n * 4 + 255) / 256;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(k_bits_or_reduce, dim3((unsigned)blocks, (unsigned)b), dim3(256), 0, st, bits, n, any);
  PH_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t ph_attn_workspace_bytes(int64_t n, int32_t b, int32_t h, int32_t qn, int32_t dh) {
  // at most 2048 + b*h partial records, each as large as the launch writes: 64 queries per query block
  // (1 or 2 blocks), whatever qn is
  const int64_t qp = qn <= 64 ? 64 : 128;
  return (int64_t)(2048 + (int64_t)b * h * 4) * qp * (dh + 4) * 4 + 256;
}

extern "C" int ph_attn_cross_fwd(const float *q, const float *k, const float *v, const uint32_t *bits,
                                 const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h,
                                 int32_t qn, int32_t dh, void *ws, int64_t ws_bytes, ph_stream_t stream) {
  PH_REQUIRE(q && k && v && out, "attn_cross_fwd: null tensor");
  PH_REQUIRE(dh == 48, "attn_cross_fwd: head dim %d not served (48)", dh);
  PH_REQUIRE(qn >= 1 && qn <= 128, "attn_cross_fwd: %d queries not served (1..128)", qn);
  PH_REQUIRE(b >= 1 && h >= 1 && n >= 1, "attn_cross_fwd: bad shape");
  PH_REQUIRE(ws_bytes >= ph_attn_workspace_bytes(n, b, h, qn, dh), "attn_cross_fwd: workspace too small");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.bits = bits; a.any = any; a.part = (float *)ws; a.out = out;
  a.n = n; a.B = b; a.H = h; a.Qn = qn; a.Dh = dh;
  const int64_t ntile = (n + 15) / 16;
  const int bh = b * h;
  // aim at ~2048 waves (256 CUs x 4 SIMDs x 2) in total, multiples of 4 waves per (b, h)
  int64_t splits = 2048 / bh;
  if (splits < 1) splits = 1;
  if (splits > ntile) splits = ntile;
  splits = ((splits + 3) / 4) * 4;
  int64_t tpw = (ntile + splits - 1) / splits;
  if (tpw < 1) tpw = 1;
  splits = (ntile + tpw - 1) / tpw;
  a.splits = (int)splits;
  a.tiles_per_wave = (int)tpw;
  hipStream_t st = ph_stream(stream);
  const int64_t waves = (int64_t)bh * splits;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  a.qp = qn <= 64 ? 64 : 128;   // 2-tile query blocks were measured slower (K / V re-read by four waves)
  hipLaunchKernelGGL((k_attn_cross<4, 3>), dim3(grid, a.qp / 64), dim3(256), 0, st, a);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_attn_merge<4, 3>), dim3(bh, (qn + 15) / 16), dim3(256), 0, st, a);
  PH_LAUNCH_CHECK();
  return 0;
}
