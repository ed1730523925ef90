// Micro-benchmark of the gather path of the sparse convolution (no MFMA, no epilogue): how fast can a CU pull the
// 128-byte operand-row chunks of a real kernel map out of L2 / Infinity Cache, by access shape and by the number
// of stages kept in flight?  Driven by tools/ubench/run_gather_bench.py on the S10 maps.
//   MODE 0  full-line: 8 lanes x 16 B cover one row chunk (what k_conv_h2 issues), register destination
//   MODE 1  fragment-shaped: lane = (row, k-half), 4 instructions x 32 B pieces per row chunk (MFMA B-operand order)
//   MODE 2  full-line, direct to LDS (global_load_lds_dwordx4), ring of DEPTH buffers, indices resident in LDS
// A stage = one (kernel offset, 32-channel chunk): 128 rows x 128 B per workgroup.  DEPTH = stages of loads in
// flight; indices are fetched 2 * DEPTH stages ahead so that waiting for them never drains younger data loads.
// The tail re-loads the last stage (clamped) instead of branching: the steady state has no control flow.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct GArgs {
  const char *in_split;   // [n_in][rs bytes]
  const int32_t *nbr;     // [kvol][n_out]
  int64_t n_out;
  int kvol, nchunks;      // chunks of 128 B per row
  uint32_t rs;            // row stride in bytes
  u32x4 *sink;            // [grid * 256]
  int n_tiles;
  const char *zero;       // >= 256 zero bytes (rows without a neighbour read these)
};


template <int MODE, int DEPTH, int LDS_PAD>
__global__ void __launch_bounds__(256) k_gather(GArgs a) {
  extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
  constexpr int BM = 128;
  const int nwg = gridDim.x, cpx = nwg >> 3, bid = blockIdx.x;
  const int tile = (bid & 7) * cpx + (bid >> 3);
  if (tile >= a.n_tiles) return;
  const int64_t m0 = (int64_t)tile * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (LDS_PAD > 0 && tid == 0) dyn_lds[LDS_PAD - 1] = 1;   // occupancy limiter only
  u32x4 acc = {0, 0, 0, 0};
  const int nstages = a.kvol * a.nchunks;
  const int last = nstages - 1;
  const int nch = a.nchunks;

  if (MODE == 0) {
    const int l_j = tid & 7, l_r = tid >> 3;   // 32 rows per pass, 4 passes
    int64_t rowp[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = m0 + l_r + p * 32;
      rowp[p] = r < a.n_out ? r : a.n_out - 1;
    }
    u32x4 ring[DEPTH][4];
    int idxr[DEPTH][4];
    auto load_idx = [&](int s, int *dst) {
      const int sc = s < last ? s : last;
      const int k = sc / nch;
#pragma unroll
      for (int p = 0; p < 4; ++p) dst[p] = a.nbr[(int64_t)k * a.n_out + rowp[p]];
    };
    auto issue = [&](int s, int slot, const int *idx) {
      const int sc = s < last ? s : last;
      const int chunk = sc % nch;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const char *src = idx[p] >= 0 ? a.in_split + (uint64_t)(uint32_t)idx[p] * a.rs + chunk * 128 + l_j * 16
                                      : a.zero + l_j * 16;
        ring[slot][p] = *reinterpret_cast<const u32x4 *>(src);
      }
    };
    {
      int tmp[DEPTH][4];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) load_idx(d, tmp[d]);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) load_idx(DEPTH + d, idxr[d]);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) issue(d, d, tmp[d]);
    }
    for (int s0 = 0; s0 < nstages; s0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int s = s0 + d;
#pragma unroll
        for (int p = 0; p < 4; ++p) acc ^= ring[d][p];
        int nidx[4];
        load_idx(s + 2 * DEPTH, nidx);        // issued BEFORE this iteration's data loads
        issue(s + DEPTH, d, idxr[d]);
#pragma unroll
        for (int p = 0; p < 4; ++p) idxr[d][p] = nidx[p];
      }
    }
  } else if (MODE == 1) {
    // fragment-shaped: wave owns rows wave*32 .. +31; lane = (row l31, k-half h); 4 x 16 B per lane per stage:
    // pieces (h0 h1) (l0 l1) (h2 h3) (l2 l3) of the row chunk's [hi 64 B | lo 64 B]
    const int l31 = lane & 31, h = lane >> 5;
    int64_t row = m0 + wave * 32 + l31;
    row = row < a.n_out ? row : a.n_out - 1;
    u32x4 ring[DEPTH][4];
    int idxr[DEPTH];
    auto load_idx1 = [&](int s) {
      const int sc = s < last ? s : last;
      return a.nbr[(int64_t)(sc / nch) * a.n_out + row];
    };
    auto issue = [&](int s, int slot, int idx) {
      const int sc = s < last ? s : last;
      const int chunk = sc % nch;
      const char *base = idx >= 0 ? a.in_split + (uint64_t)(uint32_t)idx * a.rs + chunk * 128 : a.zero;
      ring[slot][0] = *reinterpret_cast<const u32x4 *>(base + 0 + h * 16);
      ring[slot][1] = *reinterpret_cast<const u32x4 *>(base + 64 + h * 16);
      ring[slot][2] = *reinterpret_cast<const u32x4 *>(base + 32 + h * 16);
      ring[slot][3] = *reinterpret_cast<const u32x4 *>(base + 96 + h * 16);
    };
    {
      int tmp[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) tmp[d] = load_idx1(d);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) idxr[d] = load_idx1(DEPTH + d);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) issue(d, d, tmp[d]);
    }
    for (int s0 = 0; s0 < nstages; s0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int s = s0 + d;
#pragma unroll
        for (int p = 0; p < 4; ++p) acc ^= ring[d][p];
        const int nidx = load_idx1(s + 2 * DEPTH);
        issue(s + DEPTH, d, idxr[d]);
        idxr[d] = nidx;
      }
    }
  } else {
    // direct to LDS: ring of DEPTH buffers of 16 KB; each wave-instruction lands 1 KB = 8 rows x 128 B
    // (lane-linear destination: base + lane * 16).  All neighbour indices of the tile are staged in LDS first
    // (kvol * 128 ints) so that the loop contains LDS-DMA loads only and vmcnt can be counted by hand.
    const int l_j = tid & 7, l_r = tid >> 3;
    int *idx_lds = reinterpret_cast<int *>(dyn_lds + DEPTH * 16384);
    for (int i = tid; i < a.kvol * BM; i += 256) {
      const int k = i / BM, r = i % BM;
      int64_t row = m0 + r;
      row = row < a.n_out ? row : a.n_out - 1;
      idx_lds[i] = a.nbr[(int64_t)k * a.n_out + row];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto issue = [&](int s, int slot) {
      const int sc = s < last ? s : last;
      const int k = sc / nch, chunk = sc % nch;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int idx = idx_lds[k * BM + l_r + p * 32];
        const char *src = idx >= 0 ? a.in_split + (uint64_t)(uint32_t)idx * a.rs + chunk * 128 + l_j * 16
                                   : a.zero + l_j * 16;
        char *dst = dyn_lds + slot * 16384 + (p * 32 + wave * 8) * 128;   // wave-uniform base; lane adds 16 B
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d);
    for (int s0 = 0; s0 < nstages; s0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int s = s0 + d;
        // at most DEPTH-1 younger stages (4 loads each) stay outstanding -> stage s has landed
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc ^= *reinterpret_cast<const u32x4 *>(dyn_lds + d * 16384 + tid * 16);   // consume: 16 B per lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(s + DEPTH, d);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  a.sink[(int64_t)bid * 256 + tid] = acc;
}

template <int MODE, int DEPTH, int LDS_PAD>
static int launch(const GArgs &a, hipStream_t st) {
  const int grid = ((a.n_tiles + 7) / 8) * 8;
  size_t dyn = MODE == 2 ? (size_t)DEPTH * 16384 + (size_t)a.kvol * 128 * 4 : 0;
  if (LDS_PAD > (int)dyn) dyn = LDS_PAD;
  if (dyn > 65536) {
    if (hipFuncSetAttribute((const void *)k_gather<MODE, DEPTH, LDS_PAD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)dyn) != hipSuccess)
      return 3;
  }
  hipLaunchKernelGGL((k_gather<MODE, DEPTH, LDS_PAD>), dim3(grid), dim3(256), dyn, st, a);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int ub_gather(const void *in_split, const int32_t *nbr, int64_t n_out, int kvol, int nchunks, uint32_t rs,
                         void *sink, const void *zero, int mode, int depth, int lds_pad_kb, void *stream) {
  GArgs a;
  a.in_split = (const char *)in_split;
  a.nbr = nbr;
  a.n_out = n_out;
  a.kvol = kvol;
  a.nchunks = nchunks;
  a.rs = rs;
  a.sink = (u32x4 *)sink;
  a.n_tiles = (int)((n_out + 127) / 128);
  a.zero = (const char *)zero;
  hipStream_t st = (hipStream_t)stream;
#define CASE(M, D, P) if (mode == M && depth == D && lds_pad_kb == P) return launch<M, D, P * 1024>(a, st)
  CASE(0, 1, 0); CASE(0, 2, 0); CASE(0, 4, 0);
  CASE(0, 1, 40); CASE(0, 2, 40); CASE(0, 4, 40);     // <= 4 workgroups per CU
  CASE(0, 2, 80); CASE(0, 4, 80);                      // <= 2 workgroups per CU
  CASE(1, 1, 0); CASE(1, 2, 0); CASE(1, 4, 0);
  CASE(1, 2, 40); CASE(1, 4, 40);
  CASE(2, 1, 0); CASE(2, 2, 0); CASE(2, 3, 0); CASE(2, 4, 0);
  CASE(2, 2, 80); CASE(2, 4, 80);
#undef CASE
  return 2;
}
