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

// ---------------------------------------------------------------------------------------------------------------------
// The same attention on split f16 operands (what the K / V projections emit: ph_conv_desc.out_split):
// every product is three v_mfma_f32_32x32x16_f16 (hi hi + hi lo + lo hi, fp32 accumulate), 16x the fp32 MFMA rate, so
// the kernel is bound by streaming K and V once instead of by the matrix pipe.
//
//   workgroup = 4 waves = the 4 query tiles (32 queries each) of one (subnet b, head h, key range); the waves share the
//   K / V / mask tiles of 32 keys, which arrive by LDS-DMA (global_load_lds) into a ring of 3 stages, one barrier per tile
//   S^T[32 keys x 32 q] = K Q^T        A = K fragment (ds_read_b128: key = lane & 31, 8 dims), B = Q hi / lo in registers
//   online softmax per lane            lane (q = lane & 31, h2 = lane >> 5) owns keys 8g + 4 h2 + r of its query: one
//                                      exchange with lane ^ 32 for the row max, none for P
//   O^T[dims x 32 q] += V^T P          B = P straight from the S registers (k-slot i of key chunk c <-> key
//                                      16c + 8 (i >> 2) + 4 h2 + (i & 3)); A = V^T through ds_read_b64_tr_b16, the LDS
//                                      transpose read of gfx950, with the same key order.  dims are tiled 0..31 | 32..63:
//                                      the second tile's upper half is beyond the 48-dim head and never stored.
// Heads of one key range sit on the same XCD (blockIdx -> (range, head) below): the 128-byte operand groups two heads
// share are fetched once per L2.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnSplitArgs {
  AttnArgs a;                 // k / v unused; part / out / bits / any / shapes / splits (tiles_per_wave = 32-key tiles per range)
  const _Float16 *ks, *vs;    // split operands [B*N, D/32, 2, 32]
  float kv_unscale;           // 2^-exp2 of the K and V operands
  int *status;                // |Q * 2^8| beyond the f16 range -> bit 1 (the caller reruns in fp32), or null
  int groups;                 // B * splits
};

constexpr int AS_KT = 32;                     // keys per tile
constexpr int AS_K_BYTES = 12 * 32 * 16;      // [granule (chunk, plane, half)][key] 16-byte granules
constexpr int AS_V_BYTES = 48 * 128;          // [(plane, key quad, chunk)] blocks of [4 keys][16 dims]
constexpr int AS_M_BYTES = 4 * 32 * 4;        // [word][key]
constexpr int AS_STAGE = AS_K_BYTES + AS_V_BYTES + AS_M_BYTES;
constexpr int AS_STAGES = 3;
constexpr float AS_QSCALE = 256.f;            // Q is split as q * 2^8
constexpr float AS_PSCALE = 4096.f;           // P is split as p * 2^12 (cancels between O and l)

// LDS reads of the tile loop go through inline asm: the compiler cannot tell them from the LDS-DMA writes in flight for the
// next tiles and would wait vmcnt(0) in front of them (the counted waits below are the synchronisation)
__device__ __forceinline__ h16x8 as_rd128(uint32_t addr) {
  h16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 as_rd128u(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ s16x4 as_rdtr(uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <bool MASK, bool INTER = true>
__global__ void __launch_bounds__(256) k_attn_split(AttnSplitArgs s) {
  const AttnArgs &a = s.a;
  __shared__ __attribute__((aligned(16))) char lds[AS_STAGE * AS_STAGES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h2 = lane >> 5;
  // blockIdx -> (key range group, head): the H heads of a group are consecutive on one XCD
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int group = (j / a.H) * 8 + xcd, h = j % a.H;
  if (group >= s.groups) return;
  const int b = group / a.splits, split = group - b * a.splits;
  const int D = a.H * 48;
  const int rowb = (D >> 5) * 128;            // bytes per operand row
  const int64_t ntile_all = (a.n + AS_KT - 1) / AS_KT;
  const int64_t t0 = (int64_t)split * a.tiles_per_wave;
  int64_t t1 = t0 + a.tiles_per_wave;
  if (t1 > ntile_all) t1 = ntile_all;
  const int ntile = (int)(t1 - t0);

  // ---- DMA plan: 12 sixteen-byte instructions per tile (K 0..5, V 0..5; 64 granules each, lane-linear in LDS), three per
  // wave (e = wave, wave + 4, wave + 8), + the two mask-word instructions on waves 0 and 1 ----------------------------------
  // K instr i: granule gid = 2i + h2 of key l31;  V instr i: block 8i + (lane >> 3), granule (key in quad, dim half) = lane & 7
  int d_key[3], d_off[3], d_dst[3];
  const char *d_base[3];
  const char *kbase = (const char *)s.ks + (int64_t)b * a.n * rowb;
  const char *vbase = (const char *)s.vs + (int64_t)b * a.n * rowb;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = wave + 4 * k;                 // wave-uniform
    if (e < 6) {
      const int gid = 2 * e + h2;
      const int c = gid >> 2, plane = (gid >> 1) & 1, half = gid & 1;
      const int ch = 3 * h + c;
      d_key[k] = l31;
      d_off[k] = (ch >> 1) * 128 + plane * 64 + (ch & 1) * 32 + half * 16;
      d_dst[k] = e * 1024;
      d_base[k] = kbase;
    } else {
      const int i = e - 6;
      const int blk = 8 * i + (lane >> 3);
      const int vc = blk % 3, pk = blk / 3;
      const int vplane = pk >> 3, kq = pk & 7;
      const int vch = 3 * h + vc;
      d_key[k] = 4 * kq + ((lane & 7) >> 1);
      d_off[k] = (vch >> 1) * 128 + vplane * 64 + (vch & 1) * 32 + (lane & 1) * 16;
      d_dst[k] = AS_K_BYTES + i * 1024;
      d_base[k] = vbase;
    }
  }
  constexpr bool has_bits = MASK;
  const bool mask_wave = has_bits && wave < 2;          // mask words [w][key]: instr m = wave: w = 2m + h2
  const char *mbase = has_bits ? (const char *)a.bits + (int64_t)b * a.n * 16 + (2 * wave + h2) * 4 : nullptr;

  auto fire_one = [&](int tile_local, int k) {       // one of the wave's three 16-byte DMA instructions of a tile
    const int64_t nb = (t0 + tile_local) * AS_KT;
    char *st = lds + (tile_local % AS_STAGES) * AS_STAGE;
    int64_t key = nb + d_key[k];
    if (key >= a.n) key = a.n - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(d_base[k] + key * rowb + d_off[k]),
                                     (__attribute__((address_space(3))) void *)(st + d_dst[k]), 16, 0, 0);
  };
  auto fire_mask = [&](int tile_local) {
    if (mask_wave) {
      const int64_t nb = (t0 + tile_local) * AS_KT;
      char *st = lds + (tile_local % AS_STAGES) * AS_STAGE;
      int64_t key = nb + l31;
      if (key >= a.n) key = a.n - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(mbase + key * 16),
                                       (__attribute__((address_space(3))) void *)(st + AS_K_BYTES + AS_V_BYTES + wave * 256), 4, 0, 0);
    }
  };
  auto fire = [&](int tile_local) {
    const int64_t nb = (t0 + tile_local) * AS_KT;
    char *st = lds + (tile_local % AS_STAGES) * AS_STAGE;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int64_t key = nb + d_key[k];
      if (key >= a.n) key = a.n - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(d_base[k] + key * rowb + d_off[k]),
                                       (__attribute__((address_space(3))) void *)(st + d_dst[k]), 16, 0, 0);
    }
    if (mask_wave) {
      int64_t key = nb + l31;
      if (key >= a.n) key = a.n - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(mbase + key * 16),
                                       (__attribute__((address_space(3))) void *)(st + AS_K_BYTES + AS_V_BYTES + wave * 256), 4, 0, 0);
    }
  };
  const int per_tile = mask_wave ? 4 : 3;

  // ---- Q fragments of this wave's query tile: B[k = dim][n = q]: lane (q = l31, dims 8 h2 .. + 7 of chunk c) --------------
  const int qt = wave;
  const int qq = qt * 32 + l31;
  const bool qv = qq < a.Qn;
  const bool wave_live = qt * 32 < a.Qn;
  h16x8 qh[3], ql[3];
  bool qbad = false;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] = 0.f;
    if (qv) {
      const float *src = a.q + (((int64_t)b * a.H + h) * a.Qn + qq) * 48 + 16 * c + 8 * h2;
      const float4 x0 = *reinterpret_cast<const float4 *>(src), x1 = *reinterpret_cast<const float4 *>(src + 4);
      v8[0] = x0.x; v8[1] = x0.y; v8[2] = x0.z; v8[3] = x0.w; v8[4] = x1.x; v8[5] = x1.y; v8[6] = x1.z; v8[7] = x1.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = v8[i] * AS_QSCALE;
      qbad |= !(fabsf(t) <= 65504.f);
      const _Float16 th = (_Float16)t;
      qh[c][i] = th;
      ql[c][i] = (_Float16)(t - (float)th);
    }
  }
  if (s.status != nullptr && qbad) atomicOr(s.status, 1);
  unsigned force = 1u;
  if (has_bits) {
    force = 0u;
    if (a.any != nullptr && qv) force = ((a.any[b * 4 + qt] >> l31) & 1u) ? 0u : 1u;
  }
  // scores are kept in the log2 domain (the unscale carries log2(e)): p = exp2(s - m) is one v_exp_f32, no multiply per value
  const float s_unscale = s.kv_unscale * (1.f / AS_QSCALE) * 1.4426950408889634f;

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) oacc[0][i] = oacc[1][i] = 0.f;

  // fragment read addresses (LDS byte addresses within stage 0)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
  const uint32_t k_rd = lds0 + (h2 * 32 + l31) * 16;                             // + (c * 2 + plane) * 1024
  const int i16 = lane & 15, dsel = (lane >> 4) & 1;
  uint32_t v_rd[2];                                                               // per dim tile: + ((plane * 8 + kq) * 3) * 128
  v_rd[0] = lds0 + AS_K_BYTES + dsel * 128 + (i16 >> 2) * 32 + (i16 & 3) * 8;
  v_rd[1] = lds0 + AS_K_BYTES + 2 * 128 + (i16 >> 2) * 32 + (i16 & 3) * 8;       // chunk 2 for both halves (dims 48..63 do not
                                                                                  // exist: any finite operand, rows never stored)
  const uint32_t m_rd = lds0 + AS_K_BYTES + AS_V_BYTES + (qt * 32 + 4 * h2) * 4; // + 32 g
  const unsigned lanebit = 1u << l31;

  if (ntile > 0) fire(0);
  if (ntile > 1) fire(1);
  for (int t = 0; t < ntile; ++t) {
    // stage t landed (this wave's part), everybody past tile t - 1: then refill the stage tile t - 1 used
    if (t + 1 < ntile) {
      if (per_tile == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bool pre = t + 2 < ntile;
    if (pre && (!wave_live || !INTER)) fire(t + 2);          // idle query tiles still carry their share of the DMA
    if (!wave_live) continue;
    const uint32_t so = (uint32_t)((t % AS_STAGES) * AS_STAGE);
    const int64_t nb = (t0 + t) * AS_KT;
    const int nvalid = a.n - nb < AS_KT ? (int)(a.n - nb) : AS_KT;

    // ---- S^T = K Q^T ---------------------------------------------------------------------------------------------------
    h16x8 kh[3], kl[3];
    u32x4 mw[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      kh[c] = as_rd128(so + k_rd + (c * 2 + 0) * 1024);
      kl[c] = as_rd128(so + k_rd + (c * 2 + 1) * 1024);
    }
    if (MASK) {
#pragma unroll
      for (int g = 0; g < 4; ++g) mw[g] = as_rd128u(so + m_rd + 32 * g);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(kh[0]), "+v"(kh[1]), "+v"(kh[2]), "+v"(kl[0]), "+v"(kl[1]), "+v"(kl[2]), "+v"(mw[0]), "+v"(mw[1]),
                     "+v"(mw[2]), "+v"(mw[3])::"memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(kh[0]), "+v"(kh[1]), "+v"(kh[2]), "+v"(kl[0]), "+v"(kl[1]), "+v"(kl[2])::"memory");
    }
    // V^T fragments of the whole tile: in flight under the score MFMAs and the softmax
    s16x4 vt[2][2][2][2];                        // [dim tile][key chunk][plane][r]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int plane = 0; plane < 2; ++plane)
#pragma unroll
          for (int r = 0; r < 2; ++r)
            vt[mt][kc][plane][r] = as_rdtr(so + v_rd[mt] + (uint32_t)(((plane * 8 + 4 * kc + 2 * r) * 3) * 128) + (uint32_t)(h2 * 384));
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
    // the DMA of tile t + 2 goes out between the score MFMAs (a wave stalled at the issue of a DMA leaves the matrix
    // pipe the MFMAs it already issued)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (INTER && pre) fire_one(t + 2, c);
      __builtin_amdgcn_sched_barrier(0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], qh[c], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], ql[c], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[c], qh[c], sacc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (INTER && pre) fire_mask(t + 2);
    // ---- mask + online softmax: this lane's query is qq; register 4g + r is key nb + 8g + 4 h2 + r ----------------------
    // sv stays in the accumulator's raw scale: the maximum does not care, and the unscale (log2 domain) rides in the one fma
    // that forms the exponent; the 2^12 scale of P is part of that fma's addend
    float sv[16];
    float tmax = -INFINITY;
    if (nvalid == AS_KT) {      // full tile (uniform): no key-validity test; mask bit and the all-masked rule in one v_and_or
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned wr[4] = {mw[g][0], mw[g][1], mw[g][2], mw[g][3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned keep = MASK ? ((wr[r] & lanebit) | force) : 1u;
          const float x = keep ? sacc[4 * g + r] : -INFINITY;
          sv[4 * g + r] = x;
          tmax = fmaxf(tmax, x);
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned wr[4] = {mw[g][0], mw[g][1], mw[g][2], mw[g][3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bool ok = 8 * g + r + 4 * h2 < nvalid;
          if (MASK) ok = ok && (force || (wr[r] & lanebit));
          const float x = ok ? sacc[4 * g + r] : -INFINITY;
          sv[4 * g + r] = x;
          tmax = fmaxf(tmax, x);
        }
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);                            // raw scale
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * s_unscale);      // m_run = -inf -> 0 (nothing accumulated yet)
    m_run = m_new;
    const float e_off = 12.f - m_safe * s_unscale;                     // log2(AS_PSCALE) - m in the log2 domain
    h16x8 ph[2], pl[2];
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[i], s_unscale, e_off));
      psum += p;
      const _Float16 th = (_Float16)p;
      ph[i >> 3][i & 7] = th;
      pl[i >> 3][i & 7] = (_Float16)(p - (float)th);
    }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        oacc[0][i] *= alpha;
        oacc[1][i] *= alpha;
      }
    }
    // ---- O^T += V^T P ---------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vt[0][0][0][0]), "+v"(vt[0][0][0][1]), "+v"(vt[0][0][1][0]), "+v"(vt[0][0][1][1]), "+v"(vt[0][1][0][0]),
                   "+v"(vt[0][1][0][1]), "+v"(vt[0][1][1][0]), "+v"(vt[0][1][1][1]), "+v"(vt[1][0][0][0]), "+v"(vt[1][0][0][1]),
                   "+v"(vt[1][0][1][0]), "+v"(vt[1][0][1][1]), "+v"(vt[1][1][0][0]), "+v"(vt[1][1][0][1]), "+v"(vt[1][1][1][0]),
                   "+v"(vt[1][1][1][1])::"memory");
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        h16x8 vh, vl;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const h16x4 yh = __builtin_bit_cast(h16x4, vt[mt][kc][0][r]);
          const h16x4 yl = __builtin_bit_cast(h16x4, vt[mt][kc][1][r]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vh[4 * r + e] = yh[e];
            vl[4 * r + e] = yl[e];
          }
        }
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kc], oacc[mt], 0, 0, 0);
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kc], oacc[mt], 0, 0, 0);
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kc], oacc[mt], 0, 0, 0);
      }
  }

  // ---- partial record of this (b, h, range): per query O[48], m, l (k_attn_merge's format) -------------------------------
  if (!wave_live) return;
  const int64_t w = ((int64_t)b * a.H + h) * a.splits + split;
  float *row = a.part + (w * a.qp + qq) * (48 + 4);
  float lt = l_run + __shfl_xor(l_run, 32);
  if (h2 == 0) {
    row[48] = m_run * s_unscale * 0.6931471805599453f;      // raw scale -> log2 domain -> natural log (k_attn_merge's domain)
    row[49] = lt;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * mt + 8 * g + 4 * h2;
      if (d >= 48) continue;
      *reinterpret_cast<float4 *>(row + d) = make_float4(oacc[mt][4 * g] * s.kv_unscale, oacc[mt][4 * g + 1] * s.kv_unscale,
                                                         oacc[mt][4 * g + 2] * s.kv_unscale, oacc[mt][4 * g + 3] * s.kv_unscale);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention straight on a level's FEATURE operand (ph_attn_cross_feat): K and V are never formed.
//
// A level's keys and values are K = x A + a + pos W_k^T, V = x B + b + pos W_v^T with x the level's C = 64 feature channels
// (input projection and K / V projection composed, CrossAttentionLayer.composed_kv).  Per head h
//     S = q_h K_h^T = (q_h A_h^T) x^T + [constant over the keys: drops out of the softmax] + q_h . (pos terms)
//     O_h = P V_h   = (P x) B_h + b_h + P . (pos terms)
// and the position term of a key depends on its three integer coordinates only, through the rows tab[t] of the sine table.
// The reference's encoding normalises c / (c + 1e-6) * 2 pi (position_encoding.py:100-104), so tab[t] takes three kinds of
// values: tab[0], the far value tab[inf] (every |t| >= 32: the ratio is exactly 1 in fp32), and tab[inf] + eps_t G for the
// handful of small |t| (eps_t = the fp32 angle's offset from 2 pi, G = d tab / d angle; the second-order term is < 1e-10).
// So a key carries, next to its 64 channels, 16 POSITION COLUMNS (ph_pos_aug): [c_a == 0] and eps_{c_a} for the three axes,
// and both products run on the 80-column row [x | aug]:
//     S^T[32 keys x 32 q]   = [x | aug] Q2^T           Q2 = q_h [A_h^T | D0 | G | 0]   (host, [Q, 80] per (subnet, head))
//     Y^T[80 x 32 q]       += [x | aug]^T P             O_h = Y (B_h ; D0v ; Gv) + const (host, folded into out_proj)
// ONE operand tile per 32 keys (8 KB + 1 KB) serves every head and both products; the 2 x 970 MB K / V operands of the
// finest level (written once, read once) and their two projection launches are gone.
//
// LDS tile: nine 1 KB blocks (chunk of 16 channels, plane hi / lo; the position block has no lo plane) of 64 sixteen-byte
// granules (key, half).  Granule position  pos = half * 32 + (key & 16) + (((key & 15) ^ (half << 2)) ^ ((chunk & 1) << 3)):
// the key-row reads of the S product (ds_read_b128, lane = (key, half)) and the TRANSPOSED reads of the Y product
// (ds_read_b64_tr_b16: a group of 16 lanes reads [4 keys x 16 channels], two groups per pass take the two chunks of a
// 32-channel tile) are both bank-conflict free on the same copy.  LDS-DMA writes a block lane-linearly, so the permutation
// lives in the global address each lane loads.
struct AttnFeatArgs {
  AttnArgs a;                 // q = Q2 [B, H, Qn, 80]; out [B, Qn, H * 80]; k / v unused
  const _Float16 *xs;         // split feature operand [B*N, 2, 2, 32]
  const _Float16 *aug;        // position columns [B*N, 16]
  float kv_unscale;           // 2^-exp2 of the feature operand
  int *status;
  int groups;                 // B * splits
};

constexpr int AF_D = 80;                      // columns of a key row: 64 channels + 16 position columns
constexpr int AF_X_BYTES = 8 * 1024;
constexpr int AF_A_BYTES = 1024;
constexpr int AF_M_BYTES = 4 * 32 * 4;
constexpr int AF_STAGE = AF_X_BYTES + AF_A_BYTES + AF_M_BYTES;
constexpr int AF_STAGES = 3;

template <bool MASK>
__global__ void __launch_bounds__(256, 2) k_attn_feat(AttnFeatArgs s) {
  const AttnArgs &a = s.a;
  __shared__ __attribute__((aligned(16))) char lds[AF_STAGE * AF_STAGES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, h2 = lane >> 5;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int group = (j / a.H) * 8 + xcd, h = j % a.H;
  if (group >= s.groups) return;
  const int b = group / a.splits, split = group - b * a.splits;
  const int64_t ntile_all = (a.n + AS_KT - 1) / AS_KT;
  const int64_t t0 = (int64_t)split * a.tiles_per_wave;
  int64_t t1 = t0 + a.tiles_per_wave;
  if (t1 > ntile_all) t1 = ntile_all;
  const int ntile = (int)(t1 - t0);

  // ---- DMA plan: blocks e = 0 .. 7 = (chunk e >> 1, plane e & 1) of the feature operand, e = 8 the position block; wave w
  // carries blocks w and w + 4, wave 2 the position block, waves 0 and 1 the mask words (one 4-byte instruction each) ------
  const int dh2 = lane >> 5;                                     // half of the granule this lane carries
  int d_key[2], d_off[2];
  const char *xbase = (const char *)s.xs + (int64_t)b * a.n * 256;
  const char *abase = (const char *)s.aug + (int64_t)b * a.n * 32;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = wave + 4 * k, cc = e >> 1, plane = e & 1;
    d_key[k] = (lane & 16) | (((lane & 15) ^ (dh2 << 2)) ^ ((cc & 1) << 3));
    d_off[k] = (cc >> 1) * 128 + plane * 64 + (cc & 1) * 32 + dh2 * 16;
  }
  const int a_key = (lane & 16) | ((lane & 15) ^ (dh2 << 2));   // position block: chunk 4 (even)
  const bool mask_wave = MASK && wave < 2;
  const bool aug_wave = wave == 2;
  const char *mbase = MASK ? (const char *)a.bits + (int64_t)b * a.n * 16 + (2 * wave + h2) * 4 : nullptr;
  const int per_tile = wave == 3 ? 2 : ((mask_wave || aug_wave) ? 3 : 2);

  auto fire_x = [&](int tile_local, int k) {
    const int64_t nb = (t0 + tile_local) * AS_KT;
    char *st = lds + (tile_local % AF_STAGES) * AF_STAGE;
    int64_t key = nb + d_key[k];
    if (key >= a.n) key = a.n - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(xbase + key * 256 + d_off[k]),
                                     (__attribute__((address_space(3))) void *)(st + (wave + 4 * k) * 1024), 16, 0, 0);
  };
  auto fire_third = [&](int tile_local) {
    const int64_t nb = (t0 + tile_local) * AS_KT;
    char *st = lds + (tile_local % AF_STAGES) * AF_STAGE;
    if (aug_wave) {
      int64_t key = nb + a_key;
      if (key >= a.n) key = a.n - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(abase + key * 32 + dh2 * 16),
                                       (__attribute__((address_space(3))) void *)(st + AF_X_BYTES), 16, 0, 0);
    } else if (mask_wave) {
      int64_t key = nb + l31;
      if (key >= a.n) key = a.n - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)(mbase + key * 16),
                                       (__attribute__((address_space(3))) void *)(st + AF_X_BYTES + AF_A_BYTES + wave * 256), 4, 0, 0);
    }
  };
  auto fire = [&](int tile_local) {
    fire_x(tile_local, 0);
    fire_x(tile_local, 1);
    fire_third(tile_local);
  };

  // ---- Q2 fragments of this wave's query tile: B[k = column][n = q]: lane (q = l31, columns 8 h2 .. + 7 of chunk c) ------
  const int qt = wave;
  const int qq = qt * 32 + l31;
  const bool qv = qq < a.Qn;
  const bool wave_live = qt * 32 < a.Qn;
  h16x8 qh[5], ql[5];
  bool qbad = false;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    float v8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] = 0.f;
    if (qv) {
      const float *src = a.q + (((int64_t)b * a.H + h) * a.Qn + qq) * AF_D + 16 * c + 8 * h2;
      const float4 x0 = *reinterpret_cast<const float4 *>(src), x1 = *reinterpret_cast<const float4 *>(src + 4);
      v8[0] = x0.x; v8[1] = x0.y; v8[2] = x0.z; v8[3] = x0.w; v8[4] = x1.x; v8[5] = x1.y; v8[6] = x1.z; v8[7] = x1.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = v8[i] * AS_QSCALE;      // position columns too: their key columns carry the operand's 2^exp2 (ph_pos_aug)
      qbad |= !(fabsf(t) <= 65504.f);
      const _Float16 th = (_Float16)t;
      qh[c][i] = th;
      ql[c][i] = (_Float16)(t - (float)th);
    }
  }
  if (s.status != nullptr && qbad) atomicOr(s.status, 1);
  unsigned force = 1u;
  if (MASK) {
    force = 0u;
    if (a.any != nullptr && qv) force = ((a.any[b * 4 + qt] >> l31) & 1u) ? 0u : 1u;
  }
  const float s_unscale = s.kv_unscale * (1.f / AS_QSCALE) * 1.4426950408889634f;

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 oacc[3];
#pragma unroll
  for (int i = 0; i < 16; ++i) oacc[0][i] = oacc[1][i] = oacc[2][i] = 0.f;

  // ---- fragment read addresses (LDS byte addresses within stage 0) -------------------------------------------------------
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
  // key rows: granule (key = l31, half = h2) of block (chunk, plane): + (chunk * 2 + plane) * 1024, ^ 128 for odd chunks
  const uint32_t k_rd = lds0 + (uint32_t)((h2 * 32 + (l31 & 16) + ((l31 & 15) ^ (h2 << 2))) * 16);
  // transposed reads: the lane's group of 16 is (h2, dsel); lane (rr = key in quad, jp = 8-byte piece) -> granule
  // (key 16 kc + 8 r + 4 h2 + rr, half jp >> 1) of chunk cc = 2 mt + dsel (position tile: chunk 4 | a finite stand-in)
  const int i16 = lane & 15, dsel = (lane >> 4) & 1, rr = i16 >> 2, jp = i16 & 3, hq = jp >> 1;
  uint32_t v_rd[3][2];                        // [dim tile][r]: + plane * 1024 + kc * 256
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    const int cc = mt < 2 ? 2 * mt + dsel : (dsel ? 3 : 4);
    const int par = cc & 1;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      v_rd[mt][r] = lds0 + (uint32_t)(cc * 2048 + (hq * 32 + 8 * (r ^ par) + 4 * (h2 ^ hq) + rr) * 16 + (jp & 1) * 8);
  }
  const uint32_t m_rd = lds0 + AF_X_BYTES + AF_A_BYTES + (qt * 32 + 4 * h2) * 4;   // + 32 g
  const unsigned lanebit = 1u << l31;

  if (ntile > 0) fire(0);
  if (ntile > 1) fire(1);
  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) {
      if (per_tile == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const bool pre = t + 2 < ntile;
    if (pre && !wave_live) fire(t + 2);          // idle query tiles still carry their share of the DMA
    if (!wave_live) continue;
    const uint32_t so = (uint32_t)((t % AF_STAGES) * AF_STAGE);
    const int64_t nb = (t0 + t) * AS_KT;
    const int nvalid = a.n - nb < AS_KT ? (int)(a.n - nb) : AS_KT;

    // ---- S^T = [x | aug] Q2^T ------------------------------------------------------------------------------------------
    h16x8 kh[4], kl[4], ka;
    u32x4 mw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t ad = so + (k_rd ^ (uint32_t)((c & 1) << 7)) + (uint32_t)(c * 2048);
      kh[c] = as_rd128(ad);
      kl[c] = as_rd128(ad + 1024);
    }
    ka = as_rd128(so + k_rd + 8 * 1024);
    if (MASK) {
#pragma unroll
      for (int g = 0; g < 4; ++g) mw[g] = as_rd128u(so + m_rd + 32 * g);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(kh[0]), "+v"(kh[1]), "+v"(kh[2]), "+v"(kh[3]), "+v"(kl[0]), "+v"(kl[1]), "+v"(kl[2]), "+v"(kl[3]),
                     "+v"(ka), "+v"(mw[0]), "+v"(mw[1]), "+v"(mw[2]), "+v"(mw[3])::"memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(kh[0]), "+v"(kh[1]), "+v"(kh[2]), "+v"(kh[3]), "+v"(kl[0]), "+v"(kl[1]), "+v"(kl[2]), "+v"(kl[3]),
                     "+v"(ka)::"memory");
    }
    // [x | aug]^T fragments of the whole tile: in flight under the score MFMAs and the softmax
    s16x4 vt[2][2][2][2];                        // [dim tile][key chunk][plane][r]
    s16x4 va[2][2];                              // position tile: [key chunk][r]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int plane = 0; plane < 2; ++plane)
#pragma unroll
          for (int r = 0; r < 2; ++r) vt[mt][kc][plane][r] = as_rdtr(so + v_rd[mt][r] + (uint32_t)(plane * 1024 + kc * 256));
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int r = 0; r < 2; ++r) va[kc][r] = as_rdtr(so + v_rd[2][r] + (uint32_t)(kc * 256));
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (pre) {
        if (c < 2) fire_x(t + 2, c);
        else if (c == 2) fire_third(t + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], qh[c], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], ql[c], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[c], qh[c], sacc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qh[4], sacc, 0, 0, 0);
    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, ql[4], sacc, 0, 0, 0);
    // ---- mask + online softmax (as k_attn_split) -----------------------------------------------------------------------
    float sv[16];
    float tmax = -INFINITY;
    if (nvalid == AS_KT) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned wr[4] = {mw[g][0], mw[g][1], mw[g][2], mw[g][3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned keep = MASK ? ((wr[r] & lanebit) | force) : 1u;
          const float x = keep ? sacc[4 * g + r] : -INFINITY;
          sv[4 * g + r] = x;
          tmax = fmaxf(tmax, x);
        }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned wr[4] = {mw[g][0], mw[g][1], mw[g][2], mw[g][3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bool ok = 8 * g + r + 4 * h2 < nvalid;
          if (MASK) ok = ok && (force || (wr[r] & lanebit));
          const float x = ok ? sacc[4 * g + r] : -INFINITY;
          sv[4 * g + r] = x;
          tmax = fmaxf(tmax, x);
        }
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_safe) * s_unscale);
    m_run = m_new;
    const float e_off = 12.f - m_safe * s_unscale;
    h16x8 ph[2], pl[2];
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[i], s_unscale, e_off));
      psum += p;
      const _Float16 th = (_Float16)p;
      ph[i >> 3][i & 7] = th;
      pl[i >> 3][i & 7] = (_Float16)(p - (float)th);
    }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        oacc[0][i] *= alpha;
        oacc[1][i] *= alpha;
        oacc[2][i] *= alpha;
      }
    }
    // ---- Y^T += [x | aug]^T P ------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vt[0][0][0][0]), "+v"(vt[0][0][0][1]), "+v"(vt[0][0][1][0]), "+v"(vt[0][0][1][1]), "+v"(vt[0][1][0][0]),
                   "+v"(vt[0][1][0][1]), "+v"(vt[0][1][1][0]), "+v"(vt[0][1][1][1]), "+v"(vt[1][0][0][0]), "+v"(vt[1][0][0][1]),
                   "+v"(vt[1][0][1][0]), "+v"(vt[1][0][1][1]), "+v"(vt[1][1][0][0]), "+v"(vt[1][1][0][1]), "+v"(vt[1][1][1][0]),
                   "+v"(vt[1][1][1][1]), "+v"(va[0][0]), "+v"(va[0][1]), "+v"(va[1][0]), "+v"(va[1][1])::"memory");
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        h16x8 vh, vl;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const h16x4 yh = __builtin_bit_cast(h16x4, vt[mt][kc][0][r]);
          const h16x4 yl = __builtin_bit_cast(h16x4, vt[mt][kc][1][r]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vh[4 * r + e] = yh[e];
            vl[4 * r + e] = yl[e];
          }
        }
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kc], oacc[mt], 0, 0, 0);
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kc], oacc[mt], 0, 0, 0);
        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kc], oacc[mt], 0, 0, 0);
      }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      h16x8 vh;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const h16x4 yh = __builtin_bit_cast(h16x4, va[kc][r]);
#pragma unroll
        for (int e = 0; e < 4; ++e) vh[4 * r + e] = yh[e];
      }
      oacc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kc], oacc[2], 0, 0, 0);
      oacc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kc], oacc[2], 0, 0, 0);
    }
  }

  // ---- partial record of this (b, h, range): per query Y[80], m, l (k_attn_merge's format) -------------------------------
  if (!wave_live) return;
  const int64_t w = ((int64_t)b * a.H + h) * a.splits + split;
  float *row = a.part + (w * a.qp + qq) * (AF_D + 4);
  float lt = l_run + __shfl_xor(l_run, 32);
  if (h2 == 0) {
    row[AF_D] = m_run * s_unscale * 0.6931471805599453f;
    row[AF_D + 1] = lt;
  }
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * mt + 8 * g + 4 * h2;
      if (d >= AF_D) continue;
      // feature AND position columns of the keys carry the operand's 2^exp2 (ph_pos_aug writes them pre-scaled, so that
      // their Q2 columns need no extra factor and raise the f16 range flag only above 256 like the feature columns: ADVICE r4)
      const float us = s.kv_unscale;
      *reinterpret_cast<float4 *>(row + d) = make_float4(oacc[mt][4 * g] * us, oacc[mt][4 * g + 1] * us, oacc[mt][4 * g + 2] * us,
                                                         oacc[mt][4 * g + 3] * us);
    }
}

// Position columns of the keys (see k_attn_feat): aug[i] = [c_x == 0, c_y == 0, c_z == 0, eps[c_x], eps[c_y], eps[c_z], 0 ...]
// x 2^exp2 (the scale of the feature operand they extend: exact powers of two, 32.0 and eps * 32 are normal f16 values)
// as f16; eps [tab_n] (fp32, already in the caller's scale; eps of the value 0 must be 0) serves the coordinate values
// tab_lo .. tab_lo + tab_n - 1; a coordinate outside raises status bit 2 (as the per-axis tables of ph_conv_desc do).
__global__ void __launch_bounds__(256)
    k_pos_aug(const int4 *__restrict__ coords, int64_t n, const float *__restrict__ eps, int tab_lo, int tab_n, float pow2,
              uint4 *__restrict__ aug, int *__restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  const int v[3] = {c.y, c.z, c.w};
  _Float16 o[6];
  bool bad = false;
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    unsigned t = (unsigned)(v[ax] - tab_lo);
    if (t >= (unsigned)tab_n) {
      bad = true;
      t = t > 0x7fffffffu ? 0u : (unsigned)(tab_n - 1);
    }
    o[ax] = (_Float16)(v[ax] == 0 ? pow2 : 0.f);
    o[3 + ax] = (_Float16)(eps[t] * pow2);
  }
  if (bad && status != nullptr) atomicOr(status, 4);
  uint4 w0;
  w0.x = (uint32_t)__builtin_bit_cast(uint16_t, o[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, o[1]) << 16);
  w0.y = (uint32_t)__builtin_bit_cast(uint16_t, o[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, o[3]) << 16);
  w0.z = (uint32_t)__builtin_bit_cast(uint16_t, o[4]) | ((uint32_t)__builtin_bit_cast(uint16_t, o[5]) << 16);
  w0.w = 0u;
  aug[2 * i] = w0;
  aug[2 * i + 1] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int ph_pos_aug(const int32_t *coords, int64_t n, const float *eps, int32_t tab_lo, int32_t tab_n, int32_t exp2,
                          void *aug, int32_t *status, ph_stream_t stream) {
  PH_REQUIRE(n >= 0 && tab_n >= 1 && exp2 >= -14 && exp2 <= 14, "pos_aug: bad shape / exponent");
  if (n == 0) return 0;
  PH_REQUIRE(coords && eps && aug, "pos_aug: null buffer");
  hipLaunchKernelGGL(k_pos_aug, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ph_stream(stream), (const int4 *)coords, n, eps,
                     tab_lo, tab_n, ldexpf(1.f, exp2), (uint4 *)aug, status);
  PH_LAUNCH_CHECK();
  return 0;
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

// bits_out[i] = OR of the bit rows of the fine voxels inside level voxel i's s^3 block [c, c + s)^3 of batch i / n_per_b,
// looked up in the fine map's hash table (absent children contribute nothing); range[0] |= 1 when a coordinate of the
// level lies outside [lo[b], hi[b]] (the caller then takes the exact dense-index path of the reference's wrap semantics)
__global__ void __launch_bounds__(256)
    k_bits_block_or(const int4 *__restrict__ level, int64_t m, int64_t n_per_b, int s, const uint64_t *__restrict__ tkeys,
                    const int32_t *__restrict__ tvals, uint64_t mask, const uint4 *__restrict__ bin,
                    const int32_t *__restrict__ lo, const int32_t *__restrict__ hi, uint4 *__restrict__ bout,
                    int32_t *__restrict__ range) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int4 c = level[i];
  const int b = (int)(i / n_per_b);
  if (range != nullptr) {
    const bool out = c.y < lo[b * 3] || c.z < lo[b * 3 + 1] || c.w < lo[b * 3 + 2] || c.y > hi[b * 3] || c.z > hi[b * 3 + 1] ||
                     c.w > hi[b * 3 + 2];
    if (out) atomicOr(range, 1);
  }
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
  for (int dx = 0; dx < s; ++dx)
    for (int dy = 0; dy < s; ++dy)
      for (int dz = 0; dz < s; ++dz) {
        const int x = c.y + dx, y = c.z + dy, z = c.w + dz;
        if (!ph_packable(b, x, y, z)) continue;
        const int r = ph_find(tkeys, tvals, mask, ph_pack(b, x, y, z));
        if (r >= 0) {
          const uint4 v = bin[r];
          acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w;
        }
      }
  bout[i] = acc;
}

extern "C" int ph_bits_block_or(const int32_t *level_coords, int64_t m, int64_t n_per_b, int32_t s, const uint64_t *tkeys,
                                const int32_t *tvals, int64_t cap, const uint32_t *bits_in, const int32_t *lo,
                                const int32_t *hi, uint32_t *bits_out, int32_t *range, ph_stream_t stream) {
  PH_REQUIRE(s >= 1 && s <= 8 && n_per_b >= 1 && m >= 0, "bits_block_or: bad shape");
  PH_REQUIRE(ph_is_pow2(cap), "bits_block_or: cap must be pow2");
  PH_REQUIRE(range == nullptr || (lo != nullptr && hi != nullptr), "bits_block_or: range flag needs the bounds");
  if (m == 0) return 0;
  hipLaunchKernelGGL(k_bits_block_or, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ph_stream(stream),
                     (const int4 *)level_coords, m, n_per_b, s, tkeys, tvals, (uint64_t)cap - 1, (const uint4 *)bits_in, lo, hi,
                     (uint4 *)bits_out, range);
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

extern "C" int ph_attn_cross_split(const float *q, const void *k_split, const void *v_split, int32_t exp2, const uint32_t *bits,
                                   const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h, int32_t qn, int32_t dh,
                                   void *ws, int64_t ws_bytes, int32_t *status, ph_stream_t stream) {
  PH_REQUIRE(q && k_split && v_split && out, "attn_cross_split: null tensor");
  PH_REQUIRE(dh == 48, "attn_cross_split: head dim %d not served (48)", dh);
  PH_REQUIRE((h * dh) % 32 == 0, "attn_cross_split: model width %d is not a multiple of 32", h * dh);
  PH_REQUIRE(qn >= 1 && qn <= 128, "attn_cross_split: %d queries not served (1..128)", qn);
  PH_REQUIRE(b >= 1 && h >= 1 && n >= 1, "attn_cross_split: bad shape");
  PH_REQUIRE(exp2 >= -14 && exp2 <= 14, "attn_cross_split: operand exponent %d", exp2);
  // (kv_unscale = 2^-exp2 is therefore finite and positive: the running maximum is taken on the raw accumulator, which is
  // only order-preserving for a positive scale)
  PH_REQUIRE(ws_bytes >= ph_attn_workspace_bytes(n, b, h, qn, dh), "attn_cross_split: workspace too small");
  AttnSplitArgs s;
  AttnArgs &a = s.a;
  a.q = q; a.k = nullptr; a.v = nullptr; a.bits = bits; a.any = any; a.part = (float *)ws; a.out = out;
  a.n = n; a.B = b; a.H = h; a.Qn = qn; a.Dh = dh;
  s.ks = (const _Float16 *)k_split; s.vs = (const _Float16 *)v_split;
  s.kv_unscale = ldexpf(1.f, -exp2);
  s.status = status;
  const int64_t ntile = (n + AS_KT - 1) / AS_KT;
  const int bh = b * h;
  // 512 workgroups = one round of the two resident per CU (round 4; 1536 before: every key range leaves a partial record that
  // k_attn_merge reads again - 229 -> 181 us and 104 -> 66 us at the two coarser levels' sizes, merge included); at most the
  // 2048 + 4 b h partial records the workspace holds
  static const int target = [] { const char *e = PH_DEV_ENV("PASCO_ATTN_SPLIT_WGS"); return e ? atoi(e) : 512; }();
  int64_t splits = (target > 0 && target <= 2048 ? target : 512) / bh;
  if (splits < 1) splits = 1;
  if (splits > ntile) splits = ntile;
  int64_t tpw = (ntile + splits - 1) / splits;
  splits = (ntile + tpw - 1) / tpw;
  a.splits = (int)splits;
  a.tiles_per_wave = (int)tpw;
  a.qp = qn <= 64 ? 64 : 128;
  s.groups = b * (int)splits;
  hipStream_t st = ph_stream(stream);
  const int64_t groups8 = ((int64_t)s.groups + 7) / 8;       // groups per XCD
  static const bool plain = [] { const char *e = PH_DEV_ENV("PASCO_ATTN_INTER"); return e != nullptr && atoi(e) == 0; }();
  const dim3 grid((unsigned)(groups8 * h * 8));
  if (plain) {
    if (bits != nullptr) hipLaunchKernelGGL((k_attn_split<true, false>), grid, dim3(256), 0, st, s);
    else hipLaunchKernelGGL((k_attn_split<false, false>), grid, dim3(256), 0, st, s);
  } else if (bits != nullptr) hipLaunchKernelGGL((k_attn_split<true, true>), grid, dim3(256), 0, st, s);
  else hipLaunchKernelGGL((k_attn_split<false, true>), grid, dim3(256), 0, st, s);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_attn_merge<4, 3>), dim3(bh, (qn + 15) / 16), dim3(256), 0, st, a);
  PH_LAUNCH_CHECK();
  return 0;
}

extern "C" int ph_attn_cross_feat(const float *q2, const void *x_split, const void *aug, int32_t c, int32_t exp2,
                                  const uint32_t *bits, const uint32_t *any, float *out, int64_t n, int32_t b, int32_t h,
                                  int32_t qn, void *ws, int64_t ws_bytes, int32_t *status, ph_stream_t stream) {
  PH_REQUIRE(q2 && x_split && aug && out, "attn_cross_feat: null tensor");
  PH_REQUIRE(c == 64, "attn_cross_feat: %d feature channels not served (64)", c);
  PH_REQUIRE(qn >= 1 && qn <= 128, "attn_cross_feat: %d queries not served (1..128)", qn);
  PH_REQUIRE(b >= 1 && h >= 1 && n >= 1, "attn_cross_feat: bad shape");
  PH_REQUIRE(exp2 >= -14 && exp2 <= 14, "attn_cross_feat: operand exponent %d", exp2);
  PH_REQUIRE(ws_bytes >= ph_attn_workspace_bytes(n, b, h, qn, AF_D), "attn_cross_feat: workspace too small");
  AttnFeatArgs s;
  AttnArgs &a = s.a;
  a.q = q2; a.k = nullptr; a.v = nullptr; a.bits = bits; a.any = any; a.part = (float *)ws; a.out = out;
  a.n = n; a.B = b; a.H = h; a.Qn = qn; a.Dh = AF_D;
  s.xs = (const _Float16 *)x_split; s.aug = (const _Float16 *)aug;
  s.kv_unscale = ldexpf(1.f, -exp2);
  s.status = status;
  const int64_t ntile = (n + AS_KT - 1) / AS_KT;
  const int bh = b * h;
  // key ranges per (subnet, head): two workgroups are resident per CU (256 registers per wave), so 512 workgroups are one
  // round; every partial record is 43 KB that k_attn_merge reads again (2048 workgroups: 88 MB of partials, 157 us of merge;
  // 846 -> 756 us at the finest level's size, merge included)
  static const int target = [] { const char *e = PH_DEV_ENV("PASCO_ATTN_FEAT_WGS"); return e ? atoi(e) : 512; }();
  int64_t splits = (target > 0 && target <= 2048 ? target : 512) / bh;      // at most the 2048 + 4 b h records the workspace holds
  if (splits < 1) splits = 1;
  if (splits > ntile) splits = ntile;
  int64_t tpw = (ntile + splits - 1) / splits;
  splits = (ntile + tpw - 1) / tpw;
  a.splits = (int)splits;
  a.tiles_per_wave = (int)tpw;
  a.qp = qn <= 64 ? 64 : 128;
  s.groups = b * (int)splits;
  hipStream_t st = ph_stream(stream);
  const int64_t groups8 = ((int64_t)s.groups + 7) / 8;
  const dim3 grid((unsigned)(groups8 * h * 8));
  if (bits != nullptr) hipLaunchKernelGGL((k_attn_feat<true>), grid, dim3(256), 0, st, s);
  else hipLaunchKernelGGL((k_attn_feat<false>), grid, dim3(256), 0, st, s);
  PH_LAUNCH_CHECK();
  hipLaunchKernelGGL((k_attn_merge<4, 5>), dim3(bh, (qn + 15) / 16), dim3(256), 0, st, a);
  PH_LAUNCH_CHECK();
  return 0;
}
