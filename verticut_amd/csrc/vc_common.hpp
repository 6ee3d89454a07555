// Shared host/device definitions for the gfx950 Hamming k-NN engine.
// Wave width is 64 everywhere (CDNA4); nothing here is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/verticut_gpu.h"

#define VC_WAVE 64
#define VC_MAX_K 8192u           // LDS bitonic capacity of the select kernel (64 KiB of uint64)
#define VC_SORT_CAP 8192u
#define VC_PAD_ITEMS 8192ull     // column stride granularity: every scan chunk shape divides this
#define VC_MAX_W 8               // 512-bit codes
#define VC_PACK_INF 0xFFFFFFFFFFFFFFFFull
#define VC_SHIST_COPIES 16       // partial histograms per bootstrap stage (spreads the flush atomics over L2 channels)

// native 16-byte vector (two uint64): one global_load_dwordx4 per lane
typedef unsigned long long vc_u64x2 __attribute__((ext_vector_type(2)));

// ---- synthetic generator (definition shared with oracle/vc_oracle.cc gen_one) ----------------
__host__ __device__ inline uint64_t vc_mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
#define VC_SALT_CENTRE 0xC3A5C85C97CB3127ull
#define VC_SALT_ITEM 0xA0761D6478BD642Full

// result packing: search_worker.cc:254-256 (id | dist << 32)
__host__ __device__ inline uint64_t vc_pack(uint32_t dist, uint32_t id) { return ((uint64_t)dist << 32) | id; }

// ---- wave64 helpers --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t vc_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Wave64 scan / reduction on the DPP cross-lane path (row_shr inside the rows of 16, row_bcast:15 / :31 across them -- gfx9
// encodings): no LDS traffic and, unlike a __shfl ladder (ds_bpermute), no per-lane index registers -- hipcc hoisted those
// twelve index computations to kernel entry and kept them alive through kernels that sit at their VGPR limit (r04:
// mih_query_kernel spilled `lane` itself next to them).
#define VC_DPP_ROW_SHR(n) (0x110 + (n))
#define VC_DPP_ROW_BCAST15 0x142
#define VC_DPP_ROW_BCAST31 0x143
__device__ __forceinline__ uint32_t vc_wave_incl_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, VC_DPP_ROW_SHR(1), 0xf, 0xf, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, VC_DPP_ROW_SHR(2), 0xf, 0xf, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, VC_DPP_ROW_SHR(4), 0xf, 0xf, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, VC_DPP_ROW_SHR(8), 0xf, 0xf, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, VC_DPP_ROW_BCAST15, 0xa, 0xf, false);
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, VC_DPP_ROW_BCAST31, 0xc, 0xf, false);
  return x;
}

// exclusive prefix sum over the 64 lanes of a wave; total = the wave's sum (wave-uniform)
__device__ __forceinline__ uint32_t vc_wave_excl_scan(uint32_t v, uint32_t& total) {
  const uint32_t x = vc_wave_incl_scan(v);
  total = (uint32_t)__builtin_amdgcn_readlane((int)x, VC_WAVE - 1);
  return x - v;
}

// minimum over the 64 lanes of a wave (wave-uniform result)
__device__ __forceinline__ uint32_t vc_wave_min(uint32_t v) {
#define VC_MIN_STEP(ctrl, row_mask)                                                                                                  \
  {                                                                                                                                  \
    const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, row_mask, 0xf, false); /* no source: keeps v */ \
    v = y < v ? y : v;                                                                                                               \
  }
  VC_MIN_STEP(VC_DPP_ROW_SHR(1), 0xf)
  VC_MIN_STEP(VC_DPP_ROW_SHR(2), 0xf)
  VC_MIN_STEP(VC_DPP_ROW_SHR(4), 0xf)
  VC_MIN_STEP(VC_DPP_ROW_SHR(8), 0xf)
  VC_MIN_STEP(VC_DPP_ROW_BCAST15, 0xa)
  VC_MIN_STEP(VC_DPP_ROW_BCAST31, 0xc)
#undef VC_MIN_STEP
  return (uint32_t)__builtin_amdgcn_readlane((int)v, VC_WAVE - 1);
}

// in-place bitonic sort of a[0..P) (P a power of two) in LDS or global memory by one block of `nthreads` threads
__device__ __forceinline__ void vc_bitonic_lds(uint64_t* a, uint32_t P, uint32_t nthreads) {
  for (uint32_t size = 2; size <= P; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < (P >> 1); i += nthreads) {
        const uint32_t lo = 2 * i - (i & (stride - 1));
        const uint32_t hi = lo + stride;
        const uint64_t x = a[lo], y = a[hi];
        const bool up = (lo & size) == 0;
        if ((x > y) == up) {
          a[lo] = y;
          a[hi] = x;
        }
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ uint32_t vc_ld_relaxed(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- engine-internal structures ----------------------------------------------------------------
struct VcScanParams {
  const uint64_t* cols;   // W columns of `stride` uint64 each: word j of item i at cols[j*stride + i]
  uint64_t stride;
  uint64_t n;             // valid items
  uint64_t nchunks;       // ceil(n / chunk_items)
  uint32_t id_base;
  uint32_t qt;            // queries in this tile
  uint32_t k;
  uint32_t cap;           // candidate ring entries per query
  uint32_t hist_stride;   // uint32 per query in hist
  uint32_t wrap;          // diagnostic only (VC_SCAN_WRAP): loads wrap onto the first `wrap` chunks (cache-resident)
  const uint64_t* queries;  // [qt][W]
  uint32_t* tau;            // [qt] distance threshold, shared by every wave of the grid
  uint32_t* count;          // [qt] append cursor
  uint32_t* hist;           // [qt][hist_stride] histogram of appended distances
  uint64_t* buf;            // [qt][cap] appended packed candidates
  const uint64_t* limit;    // optional [qt]: append only packed values <= limit[q] (ring-overflow recovery)
  uint32_t diag;            // diagnostic only (VC_SCAN_DIAG): N > 0 replaces the verify arithmetic by s_sleep N per query
  uint32_t qs;              // stride, in words, between consecutive queries' entries of tau[] and count[] (>= 1)
  uint64_t* trace;          // diagnostic only (VC_SCAN_TRACE): [grid][2] s_memrealtime at block start / end
  uint64_t resident;        // chunks [0, resident) are read with plain (Infinity-Cache-allocating) loads, the rest non-temporal
  // Threshold bootstrap folded into the verify prologue (small tiles): when shist is set, every block cuts the sampled
  // distance histograms itself ([shist_copies] partial copies shist_cstride words apart, row q at shist + q * hist_stride)
  // instead of reading tau[] -- no vc_tau_init_kernel launch, and no coherent read of the threshold lines by every block
  const uint32_t* shist;
  uint64_t shist_cstride;
  uint32_t shist_copies;
  uint32_t bits;
};
// The linear path gives every query its own 128-byte line for its threshold and for its ring cursor: both are read /
// updated coherently by every wave that enters the rare path, coherent traffic to one line is served by ONE memory
// channel, and eight queries sharing a line overloaded that channel enough to slow the whole (channel-interleaved)
// code stream by 5-14 % depending on where the allocation happened to land (tools/placement_probe.py).
#define VC_QUERY_LINE_WORDS 32u
