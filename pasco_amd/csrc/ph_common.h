// Shared device/host helpers for libpascohip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pasco_hip.h"

#define PH_WAVE 64

// ---- development build (-DPH_DEV: pasco_amd/build.py build_hip(dev=True) -> libpascohip_dev.so, loaded only by tools/) ---------
// The PRODUCT library has no process-global experiment state, no environment switches and no exports beyond include/pasco_hip.h:
// PH_DEV_ENV answers "unset", PH_ABLATE(...) is the constant 0 (its branches compile out of the kernels), the trace buffers,
// the ablation masks and their extern "C" setters do not exist.
#ifdef PH_DEV
#include <stdlib.h>
#define PH_DEV_ENV(name) getenv(name)
#define PH_ABLATE(a, bits) ((a).ablate & (bits))
#else
#define PH_DEV_ENV(name) (static_cast<const char *>(nullptr))
#define PH_ABLATE(a, bits) 0
#endif

// ---- error plumbing ---------------------------------------------------------------------------
void ph_set_error(const char *fmt, ...);

#define PH_CHECK_HIP(expr)                                                            \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      ph_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)

#define PH_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      ph_set_error(__VA_ARGS__);  \
      return 1;                   \
    }                             \
  } while (0)

#define PH_LAUNCH_CHECK() PH_CHECK_HIP(hipGetLastError())

// ---- coordinate key -----------------------------------------------------------------------------
// (b, x, y, z) -> 64 bit: b in 10 bits, x/y/z biased by 2^17 into 18 bits each.
#define PH_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define PH_COORD_BIAS (1 << 17)

__host__ __device__ __forceinline__ uint64_t ph_pack(int b, int x, int y, int z) {
  return ((uint64_t)((uint32_t)b & 0x3FFu) << 54) |
         ((uint64_t)((uint32_t)(x + PH_COORD_BIAS) & 0x3FFFFu) << 36) |
         ((uint64_t)((uint32_t)(y + PH_COORD_BIAS) & 0x3FFFFu) << 18) |
         ((uint64_t)((uint32_t)(z + PH_COORD_BIAS) & 0x3FFFFu));
}

// what ph_pack can represent: batch 0 .. 1023, coordinates -2^17 .. 2^17 - 1 (anything else would alias another key)
__host__ __device__ __forceinline__ bool ph_packable(int b, int x, int y, int z) {
  return (unsigned)b < 1024u && (unsigned)(x + PH_COORD_BIAS) < (1u << 18) && (unsigned)(y + PH_COORD_BIAS) < (1u << 18) &&
         (unsigned)(z + PH_COORD_BIAS) < (1u << 18);
}

__host__ __device__ __forceinline__ uint64_t ph_hash(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__device__ __forceinline__ int ph_find(const uint64_t *__restrict__ tkeys,
                                       const int32_t *__restrict__ tvals, uint64_t mask,
                                       uint64_t key) {
  uint64_t slot = ph_hash(key) & mask;
  for (;;) {
    uint64_t k = tkeys[slot];
    if (k == key) return tvals[slot];
    if (k == PH_EMPTY_KEY) return -1;
    slot = (slot + 1) & mask;
  }
}

static inline hipStream_t ph_stream(ph_stream_t s) { return (hipStream_t)s; }

// ---- fused activations ---------------------------------------------------------------------------
// neg = 1 (none), 0 (ReLU), slope (leaky).  Selects instead of max / min arithmetic: NaN stays NaN (the torch
// graph of the reference propagates it; fmaxf(v, 0) + neg * fminf(v, 0) turns it into 0), -inf under ReLU is +0
// (not 0 * inf), and every finite value gets exactly what the C restatement in oracle/ computes.
__device__ __forceinline__ float ph_act(float v, float neg) {
  const float n = neg * v;
  return (v > 0.f || v != v) ? v : (neg == 0.f ? 0.f : n);
}

// ---- which convolution kernel the last ph_conv_fwd of this thread launched (ph_conv_last_config) --
struct ph_conv_cfg_rec {
  int32_t v[8];   // mma_mode, bm, bn, kc, ksplit, emit, kernel id (0 k_conv_mfma, 1 k_conv_f16x3, 2 k_conv_h2, ...), waves
};
extern thread_local ph_conv_cfg_rec ph_last_cfg;
static inline void ph_record_cfg(int mode, int bm, int bn, int kc, int ksplit, int emit, int kid, int waves) {
  ph_last_cfg = ph_conv_cfg_rec{{mode, bm, bn, kc, ksplit, emit, kid, waves}};
}

static inline int ph_is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// ---- stable compaction primitive (coords.hip) --------------------------------------------------
// flags[n] (bytes, non-zero = keep).  Writes keep_rows[rank] = i, optional rank_of[i] (-1 when
// dropped), and *n_keep.  ws needs ph_workspace_bytes(n).
int ph_compact_flags(const uint8_t *flags, int64_t n, int32_t *keep_rows, int32_t *rank_of,
                     int32_t *n_keep, void *ws, int64_t ws_bytes, hipStream_t st);

// split-precision convolution (conv_f16x3.hip), reached through ph_conv_fwd when desc->mma_mode == 1
int ph_conv_fwd_f16x3(const ph_conv_desc *d, hipStream_t st);
