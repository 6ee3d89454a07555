// ============================================================================
// vc_sort.hip -- the index builder's primitives, hand-written for gfx950 (wave64): a stable LSD radix sort of
// (key, id) pairs and an exclusive prefix sum.  Replaces the hipcub calls of round 1 (the product now links no
// device library at all).
//
// Used by vc_mih_build (vc_mih.hip) for rule a12 of SURVEY.md section 8: bucket (t, key) holds the records whose
// substring t equals key IN APPEND (= id) ORDER (build_hash_tables.cc:36-64), i.e. the ids sorted by key with a
// STABLE sort -- ids enter in ascending order, so every bucket is an ascending id run.
//
// Radix sort, 8-bit digits, one pass per digit (1 / 2 / 4 passes for 8- / 16- / 32-bit substrings):
//   rs_hist_kernel     per block (RS_BLOCK_ITEMS consecutive items) a 256-bin digit histogram -> hist[digit][block]
//   exclusive scan     over the digit-major histogram = global start of every (digit, block) run, stable by layout
//   rs_scatter_kernel  per block, sub-tile by sub-tile (4096 items): every wave ranks its 64 items per round among
//                      the lanes with the same digit (8 ballots -> peer mask -> popcount below the lane), the per-wave
//                      digit counters live in LDS, the sub-tile is staged in LDS in sorted order and written out as
//                      contiguous runs (one run per digit), the block's running digit offsets advance.
// Exclusive scan: reduce per 8192-element block -> scan of the block sums by one block -> scan with the block base.
// ============================================================================
#include "vc_internal.hpp"

#define RS_THREADS 256u
#define RS_WAVES (RS_THREADS / VC_WAVE)
#define RS_TILE 4096u                    // items ranked, staged and written per sub-tile (16 per thread)
#define RS_ROUNDS (RS_TILE / RS_THREADS) // 16 rounds of one item per lane
#define RS_SUBTILES 4u
#define RS_BLOCK_ITEMS (RS_TILE * RS_SUBTILES)   // 16384 consecutive items per block
#define SCAN_THREADS 256u
#define SCAN_ITEMS 8192u                 // elements per block of the scan kernels (32 per thread)

namespace {

__global__ void __launch_bounds__(RS_THREADS) rs_hist_kernel(const uint32_t* __restrict__ keys, uint64_t n, uint32_t shift,
                                                             uint32_t nblocks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * RS_BLOCK_ITEMS;
  for (uint32_t i = threadIdx.x; i < RS_BLOCK_ITEMS; i += RS_THREADS) {
    const uint64_t idx = base + i;
    if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];   // digit-major: the scan order is (digit, block)
}

// offs = exclusive scan of hist (digit-major): offs[d * nblocks + b] = first output position of block b's digit-d run
__global__ void __launch_bounds__(RS_THREADS) rs_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                uint64_t n, uint32_t shift, uint32_t nblocks,
                                                                const uint32_t* __restrict__ offs) {
  __shared__ uint32_t s_wcnt[RS_WAVES][256];   // per wave: items of each digit seen so far in this sub-tile
  __shared__ uint32_t s_dstart[257];           // sub-tile: start of every digit's run in sorted order
  __shared__ uint32_t s_run[256];              // block: next global output position of every digit
  __shared__ uint32_t s_key[RS_TILE], s_val[RS_TILE];
  __shared__ uint32_t s_wsum[RS_WAVES];
  const uint32_t tid = threadIdx.x, lane = vc_lane(), wave = tid / VC_WAVE;
  s_run[tid] = offs[(uint64_t)tid * nblocks + blockIdx.x];
  const uint64_t bbase = (uint64_t)blockIdx.x * RS_BLOCK_ITEMS;

  for (uint32_t st = 0; st < RS_SUBTILES; ++st) {
    const uint64_t tbase = bbase + (uint64_t)st * RS_TILE;
    if (tbase >= n) break;                      // block-uniform
    const uint32_t tcount = (uint32_t)min((uint64_t)RS_TILE, n - tbase);
    for (uint32_t w = 0; w < RS_WAVES; ++w) s_wcnt[w][tid] = 0;
    __syncthreads();
    // ---- rank: wave w owns items [w * 1024, (w + 1) * 1024) of the sub-tile, 64 consecutive items per round, so that
    // (wave, round, lane) order is input order (the sort must be stable)
    uint32_t key[RS_ROUNDS], val[RS_ROUNDS], rnk[RS_ROUNDS];
#pragma unroll
    for (uint32_t j = 0; j < RS_ROUNDS; ++j) {
      const uint32_t li = (wave * RS_ROUNDS + j) * VC_WAVE + lane;   // index inside the sub-tile
      const bool ok = li < tcount;
      key[j] = ok ? keys_in[tbase + li] : 0xFFFFFFFFu;
      val[j] = ok ? vals_in[tbase + li] : 0u;
    }
#pragma unroll
    for (uint32_t j = 0; j < RS_ROUNDS; ++j) {
      const uint32_t li = (wave * RS_ROUNDS + j) * VC_WAVE + lane;
      const bool ok = li < tcount;
      const uint32_t d = (key[j] >> shift) & 255u;
      uint64_t peers = __ballot(ok);            // lanes of this round with the same digit (padding lanes excluded)
#pragma unroll
      for (uint32_t b = 0; b < 8; ++b) {
        const uint64_t m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
      }
      const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
      uint32_t old = 0;
      if (ok && below == 0) {                   // the lowest peer lane keeps the wave's counter of this digit
        old = s_wcnt[wave][d];
        s_wcnt[wave][d] = old + (uint32_t)__popcll(peers);
      }
      const uint32_t leader = peers ? (uint32_t)__ffsll((long long)peers) - 1u : lane;
      old = __shfl(old, leader, VC_WAVE);
      rnk[j] = old + below;                     // rank among this wave's items of digit d
    }
    __syncthreads();
    // ---- digit totals of the sub-tile, starts of the digit runs (exclusive scan over 256 digits), wave bases per digit
    {
      uint32_t tot = 0, wb[RS_WAVES];
#pragma unroll
      for (uint32_t w = 0; w < RS_WAVES; ++w) {
        wb[w] = tot;
        tot += s_wcnt[w][tid];
      }
      uint32_t wtot;
      const uint32_t ex = vc_wave_excl_scan(tot, wtot);
      if (lane == 0) s_wsum[wave] = wtot;
      __syncthreads();
      uint32_t base = 0;
      for (uint32_t w = 0; w < wave; ++w) base += s_wsum[w];
      const uint32_t start = base + ex;
      s_dstart[tid] = start;
      if (tid == 255) s_dstart[256] = start + tot;
#pragma unroll
      for (uint32_t w = 0; w < RS_WAVES; ++w) s_wcnt[w][tid] = start + wb[w];   // now: sorted position of the wave's first item of the digit
    }
    __syncthreads();
    // ---- stage the sub-tile in sorted order
#pragma unroll
    for (uint32_t j = 0; j < RS_ROUNDS; ++j) {
      const uint32_t li = (wave * RS_ROUNDS + j) * VC_WAVE + lane;
      if (li < tcount) {
        const uint32_t d = (key[j] >> shift) & 255u;
        const uint32_t pos = s_wcnt[wave][d] + rnk[j];
        s_key[pos] = key[j];
        s_val[pos] = val[j];
      }
    }
    __syncthreads();
    // ---- write out: consecutive sorted positions of one digit are consecutive global addresses
    for (uint32_t pos = tid; pos < tcount; pos += RS_THREADS) {
      const uint32_t k = s_key[pos];
      const uint32_t d = (k >> shift) & 255u;
      const uint32_t g = s_run[d] + (pos - s_dstart[d]);
      keys_out[g] = k;
      vals_out[g] = s_val[pos];
    }
    __syncthreads();
    s_run[tid] += s_dstart[tid + 1] - s_dstart[tid];
    __syncthreads();
  }
}

// ---- exclusive scan ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t& total) {
  const uint32_t lane = vc_lane(), wave = threadIdx.x / VC_WAVE;
  uint32_t wtot;
  const uint32_t ex = vc_wave_excl_scan(v, wtot);
  if (lane == 0) s_w[wave] = wtot;
  __syncthreads();
  uint32_t base = 0;
  total = 0;
  for (uint32_t w = 0; w < SCAN_THREADS / VC_WAVE; ++w) {
    if (w < wave) base += s_w[w];
    total += s_w[w];
  }
  __syncthreads();
  return base + ex;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_reduce_kernel(const uint32_t* __restrict__ in, uint64_t L, uint32_t* __restrict__ bsum) {
  __shared__ uint32_t s_w[SCAN_THREADS / VC_WAVE];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_ITEMS;
  uint32_t v = 0;
  for (uint32_t i = threadIdx.x; i < SCAN_ITEMS; i += SCAN_THREADS)
    if (base + i < L) v += in[base + i];
  uint32_t total;
  (void)block_excl_scan(v, s_w, total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one block: exclusive scan of the block sums in place (nb of them, any number: chunks of SCAN_THREADS with a carry)
__global__ void __launch_bounds__(SCAN_THREADS) scan_sums_kernel(uint32_t* __restrict__ bsum, uint32_t nb) {
  __shared__ uint32_t s_w[SCAN_THREADS / VC_WAVE];
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < nb; c0 += SCAN_THREADS) {
    const uint32_t i = c0 + threadIdx.x;
    const uint32_t v = i < nb ? bsum[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan(v, s_w, total);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_apply_kernel(const uint32_t* in, uint32_t* out, uint64_t L,   // in == out allowed
                                                                 
                                                                  const uint32_t* __restrict__ bsum) {
  __shared__ uint32_t s_w[SCAN_THREADS / VC_WAVE];
  constexpr uint32_t PER = SCAN_ITEMS / SCAN_THREADS;   // consecutive elements per thread
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_ITEMS + (uint64_t)threadIdx.x * PER;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (uint32_t i = 0; i < PER; ++i) {
    v[i] = base + i < L ? in[base + i] : 0u;
    sum += v[i];
  }
  uint32_t total;
  uint32_t run = bsum[blockIdx.x] + block_excl_scan(sum, s_w, total);
#pragma unroll
  for (uint32_t i = 0; i < PER; ++i) {
    if (base + i < L) out[base + i] = run;
    run += v[i];
  }
}

}  // namespace

size_t vc_scan_work_words(uint64_t L) { return (size_t)((L + SCAN_ITEMS - 1) / SCAN_ITEMS) + 32; }

hipError_t vc_exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t L, uint32_t* d_work, hipStream_t s) {
  if (L == 0) return hipSuccess;
  const uint32_t nb = (uint32_t)((L + SCAN_ITEMS - 1) / SCAN_ITEMS);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, d_in, L, d_work);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, d_work, nb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_THREADS), 0, s, d_in, d_out, L, d_work);
  return hipGetLastError();
}

uint32_t vc_radix_sort_passes(uint32_t key_bits) { return (key_bits + 7) / 8; }

size_t vc_radix_sort_work_words(uint64_t n) {
  const uint64_t nblocks = (n + RS_BLOCK_ITEMS - 1) / RS_BLOCK_ITEMS;
  const uint64_t L = 256 * std::max<uint64_t>(nblocks, 1);
  return (size_t)L + vc_scan_work_words(L);
}

// Stable sort of n (key, value) pairs by the low key_bits bits of the key.  Input in (keys[0], vals[0]); the passes
// ping-pong between the two buffer pairs; the result is in pair (passes & 1).
hipError_t vc_radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint64_t n, uint32_t key_bits, uint32_t* d_work, hipStream_t s) {
  const uint32_t passes = vc_radix_sort_passes(key_bits);
  if (n == 0) return hipSuccess;
  const uint32_t nblocks = (uint32_t)((n + RS_BLOCK_ITEMS - 1) / RS_BLOCK_ITEMS);
  const uint64_t L = 256ull * nblocks;
  uint32_t* d_hist = d_work;
  uint32_t* d_scan = d_work + L;
  int cur = 0;
  for (uint32_t p = 0; p < passes; ++p) {
    const uint32_t shift = 8 * p;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nblocks), dim3(RS_THREADS), 0, s, keys[cur], n, shift, nblocks, d_hist);
    hipError_t r = vc_exclusive_scan_u32(d_hist, d_hist, L, d_scan, s);
    if (r != hipSuccess) return r;
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblocks), dim3(RS_THREADS), 0, s, keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, shift,
                       nblocks, d_hist);
    r = hipGetLastError();
    if (r != hipSuccess) return r;
    cur ^= 1;
  }
  return hipSuccess;
}
