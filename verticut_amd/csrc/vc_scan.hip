// ============================================================================
// vc_scan.hip -- the verify path: full-code XOR/popcount over packed uint64 code columns,
// threshold filter, wavefront ballot/prefix-sum compaction, top-k select.   gfx950 only.
//
// Replaces (reference, CPU):
//   Pilaf/image_tools.h:21-33      compute_hamming_dist  -> vc_dist<W>()
//   src/linear_search.cc:44-57     scan + max-heap(k)    -> vc_scan_kernel + vc_select_kernel
//   src/search_worker.cc:249-257   candidate verify+pack -> same device functions (vc_mih.hip)
//   src/search_worker.cc:179-199   master-side heap      -> vc_select_kernel (ring or gathered lists)
//
// Roofline: HBM read.  Algorithmic bytes per launch of vc_scan_kernel = N * B/8 (every code
// byte exactly once per query tile); integer work = qt * N * (B/32 xor + B/32 v_bcnt + ~1).
// ============================================================================
#include <stdio.h>
#include <stdlib.h>

#include "vc_internal.hpp"

// Build with -DVC_SCAN_DIAGNOSTICS=1 (VC_BUILD_DIAG=1 python -m verticut_amd.build) to enable the VC_SCAN_WRAP /
// VC_SCAN_DIAG study knobs; they cost registers, so the product build leaves them out.
#ifndef VC_SCAN_DIAGNOSTICS
#define VC_SCAN_DIAGNOSTICS 0
#endif
// code bytes are read once per launch: non-temporal loads keep them from displacing the query/threshold lines
#ifndef VC_SCAN_NT
#define VC_SCAN_NT 1
#endif
// waves per SIMD the 256-thread variants must leave room for (3 -> up to 168 VGPRs, 4 -> 128)
#ifndef VC_SCAN_MIN_WAVES
#define VC_SCAN_MIN_WAVES 4
#endif
// software-pipeline the LDS query reads through two register sets (costs 2*W+2 VGPRs)
#ifndef VC_SCAN_QUERY_PREFETCH
#define VC_SCAN_QUERY_PREFETCH 1
#endif
// ring-fill stride at which the chip-wide threshold is re-derived from the histogram
// waves per SIMD the small-tile form (<= 8 queries, <= 128-bit codes) is compiled for
#ifndef VC_SCAN_SMALL_WAVES
#define VC_SCAN_SMALL_WAVES 4
#endif
#ifndef VC_SCAN_RECUT_EVERY
#define VC_SCAN_RECUT_EVERY 32u
#endif

namespace {

// ------------------------------------------------------------------------------------------
// data-movement kernels
// ------------------------------------------------------------------------------------------
// Streaming-read probe in the verify kernel's access pattern: persistent blocks walk chunks of 2048 items grid-strided
// and read every column of a chunk (16 B per lane, non-temporal).  vc_create times it for a few candidate column
// strides (see there).
template <int W>
__global__ void __launch_bounds__(256) vc_stream_probe_kernel(const uint64_t* __restrict__ cols, uint64_t stride,
                                                              uint64_t nchunks, uint64_t* __restrict__ sink) {
  uint64_t acc = 0;
  for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    vc_u64x2 v[4][W];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < W; ++j)
        v[u][j] = __builtin_nontemporal_load(
            reinterpret_cast<const vc_u64x2*>(cols + (uint64_t)j * stride + c * 2048 + (uint64_t)u * 512 + 2 * threadIdx.x));
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < W; ++j) acc += v[u][j].x ^ v[u][j].y;
  }
  if (acc == 0x9E3779B97F4A7C15ull) *sink = acc;   // keeps the loads alive; never true on the zero-filled buffer
}

__global__ void __launch_bounds__(256) vc_fill_synth_kernel(uint64_t* __restrict__ cols, uint64_t stride, uint32_t W,
                                                            uint64_t first_local, uint64_t n, uint64_t first_gid,
                                                            uint64_t seed, uint32_t kind, uint32_t n_centres,
                                                            uint32_t max_flips) {
  const uint64_t sm = vc_mix64(seed);
  const uint64_t cm = vc_mix64(seed ^ VC_SALT_CENTRE);
  const uint64_t im = vc_mix64(seed ^ VC_SALT_ITEM);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t gid = first_gid + i;
    uint64_t w[VC_MAX_W];
    if (kind == VC_SYNTH_UNIFORM) {
      for (uint32_t j = 0; j < W; ++j) w[j] = vc_mix64(sm ^ (gid * 16 + j));
    } else {
      const uint64_t r0 = vc_mix64(im ^ gid);
      const uint64_t c = r0 % n_centres;
      const uint32_t nflips = (uint32_t)((r0 >> 32) % (max_flips + 1));
      for (uint32_t j = 0; j < W; ++j) w[j] = vc_mix64(cm ^ (c * 16 + j));
      for (uint32_t t = 0; t < nflips; ++t) {
        const uint32_t pos = (uint32_t)(vc_mix64(r0 + t + 1) % (64u * W));
        for (uint32_t j = 0; j < W; ++j)
          if (j == (pos >> 6)) w[j] ^= 1ull << (pos & 63);
      }
    }
    for (uint32_t j = 0; j < W; ++j) cols[j * stride + first_local + i] = w[j];
  }
}

// row-major records (as in the reference's code file, build_hash_tables.cc:42) -> uint64 columns
__global__ void __launch_bounds__(256) vc_rows_to_cols_kernel(const uint64_t* __restrict__ rows,
                                                              uint64_t* __restrict__ cols, uint64_t stride, uint32_t W,
                                                              uint64_t first_local, uint64_t n) {
  const uint64_t total = n * W;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = e / W;
    const uint32_t j = (uint32_t)(e % W);
    cols[j * stride + first_local + i] = rows[e];
  }
}

__global__ void __launch_bounds__(256) vc_gather_rows_kernel(const uint64_t* __restrict__ cols, uint64_t stride,
                                                             uint32_t W, const uint32_t* __restrict__ ids,
                                                             uint32_t n_ids, uint64_t* __restrict__ rows) {
  const uint32_t total = n_ids * W;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const uint32_t i = e / W, j = e % W;
    rows[e] = cols[j * stride + ids[i]];
  }
}

// ------------------------------------------------------------------------------------------
// Hamming distance of one item (W uint64 words held in registers) to a query.
// hipcc lowers each 64-bit popcount to two v_bcnt_u32_b32 with the running sum as accumulator.
// ------------------------------------------------------------------------------------------
// v_bcnt_u32_b32 d, x, acc  ==  popcount(x) + acc.  Spelled as asm so the four (eight, ...) word counts of a
// code stay ONE accumulate chain: left to itself hipcc forms two half chains and spends a v_add3 per item.
__device__ __forceinline__ uint32_t vc_bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
__device__ __forceinline__ uint32_t vc_bcnt0(uint32_t x) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x));
  return r;
}

template <int W>
__device__ __forceinline__ uint32_t vc_dist(const uint64_t (&c)[W], const uint64_t (&q)[W]) {
  uint32_t d = 0;
#pragma unroll
  for (int j = 0; j < W; ++j) {  // one accumulating v_bcnt_u32_b32 per 32-bit half, single dependency chain
    const uint64_t x = c[j] ^ q[j];
    d = (uint32_t)__builtin_popcount((uint32_t)x) + d;
    d = (uint32_t)__builtin_popcount((uint32_t)(x >> 32)) + d;
  }
  return d;
}

// ------------------------------------------------------------------------------------------
// Threshold bootstrap: exact distance histogram of the first s_items codes, per query.
// One LDS histogram per (block, query sub-tile); flushed with one global atomic per non-empty bin.
// ------------------------------------------------------------------------------------------
#define VC_SAMPLE_QSUB 32

// smallest d with  sum_{d' <= d} h[d'] >= k  (0xFFFFFFFF if the histogram holds fewer than k).  One wave.
// The histogram may be split into `copies` partial histograms `cstride` words apart (the bootstrap kernels flush into
// several copies so that their same-address atomics spread over L2 channels); a bin's count is the sum over copies.
__device__ __forceinline__ uint32_t vc_hist_cut(const uint32_t* h, uint32_t nbins, uint32_t k, bool atomic_reads,
                                                uint32_t copies = 1, uint64_t cstride = 0) {
  const uint32_t lane = vc_lane();
  const uint32_t bpl = (nbins + VC_WAVE - 1) / VC_WAVE;  // consecutive bins per lane
  auto bin_count = [&](uint32_t bin) {
    uint32_t c = 0;
    for (uint32_t r = 0; r < copies; ++r) c += atomic_reads ? vc_ld_relaxed(h + r * cstride + bin) : h[r * cstride + bin];
    return c;
  };
  uint32_t mine = 0;
  for (uint32_t i = 0; i < bpl; ++i) {
    const uint32_t bin = lane * bpl + i;
    if (bin < nbins) mine += bin_count(bin);
  }
  uint32_t total;
  uint32_t run = vc_wave_excl_scan(mine, total);
  uint32_t cand = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < bpl; ++i) {
    const uint32_t bin = lane * bpl + i;
    if (bin < nbins) {
      run += bin_count(bin);
      if (run >= k && cand == 0xFFFFFFFFu) cand = bin;
    }
  }
  return vc_wave_min(cand);
}

struct VcSampleParams {
  const uint64_t* cols;
  uint64_t stride;
  uint64_t s_items;          // histogram the first s_items codes
  const uint64_t* queries;   // [qt][W]
  uint32_t* shist;           // [VC_SHIST_COPIES][qt][hs] zeroed by the caller; block b flushes into copy b % COPIES
  const uint32_t* tau;       // [qt] refine: only distances <= tau[q] are counted
  uint32_t qt, hs, refine;
  uint32_t qs;               // words between consecutive queries' tau entries
};

// Stage kernel of the threshold bootstrap.  Blocks histogram their share of the prefix in LDS and flush with one
// global atomic per non-empty bin; vc_tau_init_kernel turns the finished histogram into the threshold.
template <int W>
__global__ void __launch_bounds__(256) vc_sample_hist_kernel(const VcSampleParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t q0 = blockIdx.y * VC_SAMPLE_QSUB;
  const uint32_t nq = min((uint32_t)VC_SAMPLE_QSUB, p.qt - q0);
  const uint32_t hs = p.hs;
  uint64_t* sq = (uint64_t*)smem;                                      // [nq][W]
  uint32_t* lh = (uint32_t*)(smem + (size_t)VC_SAMPLE_QSUB * W * 8);   // [nq][hs]
  uint32_t* sthr = lh + (size_t)VC_SAMPLE_QSUB * hs;                   // [nq] only distances <= thr are counted
  for (uint32_t i = threadIdx.x; i < nq * W; i += blockDim.x) sq[i] = p.queries[(uint64_t)q0 * W + i];
  for (uint32_t i = threadIdx.x; i < nq * hs; i += blockDim.x) lh[i] = 0;
  for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) sthr[i] = p.refine ? p.tau[(size_t)(q0 + i) * p.qs] : 0xFFFFFFFFu;
  __syncthreads();

  const uint64_t npairs = (p.s_items + 1) / 2;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  // one item pair per lane per round, the next round's pair already in flight (register double buffer)
  auto fetch = [&](vc_u64x2(&v)[W], uint64_t pr) {
    const uint64_t pc = pr < npairs ? pr : npairs - 1;   // clamp: the duplicate is never counted
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] = *reinterpret_cast<const vc_u64x2*>(p.cols + j * p.stride + 2 * pc);
  };
  uint64_t pr = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  vc_u64x2 cur[W], nxt[W];
  if (pr < npairs) fetch(cur, pr);
  for (; pr < npairs; pr += step) {
    fetch(nxt, pr + step);
    uint64_t a[W], b[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      a[j] = cur[j].x;
      b[j] = cur[j].y;
    }
    const bool okb = 2 * pr + 1 < p.s_items;
    for (uint32_t q = 0; q < nq; ++q) {
      uint64_t qw[W];
#pragma unroll
      for (int j = 0; j < W; ++j) qw[j] = sq[q * W + j];
      const uint32_t t = sthr[q];
      const uint32_t da = vc_dist<W>(a, qw), db = vc_dist<W>(b, qw);
      if (da <= t) atomicAdd(&lh[q * hs + da], 1u);
      if (okb && db <= t) atomicAdd(&lh[q * hs + db], 1u);
    }
#pragma unroll
    for (int j = 0; j < W; ++j) cur[j] = nxt[j];
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < nq * hs; i += blockDim.x) {
    const uint32_t c = lh[i];
    if (c) atomicAdd(&p.shist[(uint64_t)(blockIdx.x % VC_SHIST_COPIES) * p.qt * hs + (uint64_t)q0 * hs + i], c);
  }
}

// tau[q] = smallest d with cumulative count >= k.  First stage without a cut (fewer than k samples) = accept
// everything; refining stage without a cut = keep the previous bound.  One wave per query.  (A "last block to
// arrive" epilogue in the stage kernel was tried instead of this launch: 2048 tickets on one counter cost 0.1 ms.)
__global__ void __launch_bounds__(64) vc_tau_init_kernel(const uint32_t* __restrict__ shist, uint32_t hs, uint32_t k,
                                                         uint32_t bits, uint32_t* __restrict__ tau, uint32_t refine,
                                                         uint32_t qs) {
  const uint32_t q = blockIdx.x;
  const uint32_t cut = vc_hist_cut(shist + (uint64_t)q * hs, bits + 1, k, false, VC_SHIST_COPIES, (uint64_t)gridDim.x * hs);
  uint32_t* tq = tau + (size_t)q * qs;
  if (threadIdx.x == 0) *tq = (cut == 0xFFFFFFFFu) ? (refine ? *tq : bits) : (refine ? min(cut, *tq) : cut);
}

// ------------------------------------------------------------------------------------------
// THE verify kernel.
//
// Layout: word j of item i at cols[j*stride + i].  A lane owns the item pair (2p, 2p+1) and reads
// it with one 16-byte load per column, so every wave-instruction fetches 1 KiB contiguous.
// A block walks chunks of 2*BLK*U items grid-strided; the next chunk's loads are issued before the
// current chunk is verified (register double buffer), so HBM requests stay in flight under the VALU work.
//
// Query tile: staged once per block in LDS ([qt][W] words + [qt] thresholds) and streamed with
// wave-uniform (broadcast) ds_reads against the register-resident code tile -- the DB is read once
// per tile whatever qt is.
//
// Filter: per query a distance threshold tau (k-th best distance known so far, chip-wide).  Hot loop =
// xor, v_bcnt accumulate, min, one compare per query.  Rare path (a lane beat tau): refresh tau from
// HBM, ballot + prefix-sum the survivors of the wave, reserve ring space with ONE atomic per wave and
// surviving item pair, store packed dist<<32|id (search_worker.cc:254-256), bump the chip-wide distance
// histogram and, every VC_SCAN_RECUT_EVERY ring entries, re-derive tau from it (smallest d whose
// cumulative count reaches k).  tau only ever decreases and a
// stale tau is merely conservative, so no ordering between waves is needed for correctness.
// ------------------------------------------------------------------------------------------
// Rare path of the verify kernel (a lane beat the threshold).  It is inlined -- an out-of-line call would have to
// save tile registers whose hand-issued loads are still in flight -- and works on the caller's register tile one item
// pair at a time: recompute the pair's two distances, and only when a lane of the wave survives reserve ring slots for
// that pair and append.  Nothing but wave-uniform scalars lives across pairs, so the path needs no VGPRs beyond the
// hot loop's own temporaries (an earlier version that kept all 2*U distances across ONE reservation spilled in the
// chunk loop; one that re-read the items from memory paid 2*U loaded-HBM round trips per entry).
struct VcScanRare {
  uint64_t n;
  uint32_t* tau;
  uint32_t* count;
  uint32_t* hist;
  uint64_t* buf;
  const uint64_t* limit;
  uint32_t id_base, k, cap, hist_stride, qs;
  uint32_t* dbg;   // diagnostic build + VC_SCAN_TRACE: [0] rare-path entries, [1] entries that appended, [2] appended items,
                   // [3] re-cuts, [8..63] appended items by pass of the block's chunk loop
};

template <int W, int U, int BLK>
__device__ __forceinline__ void vc_scan_slow(const VcScanRare& p, const vc_u64x2 (&r)[U][W], const uint64_t (&qw)[W],
                                             uint32_t q, uint64_t chunk_base, uint32_t* st) {
  const uint32_t lane = vc_lane();
  // The thread index enters through an opaque statement INSIDE the rare branch: otherwise hipcc hoists the item ids
  // of all U pairs (invariant across the query loop) to tile level and keeps them alive through the hot loop.
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  uint32_t* const tau_q = p.tau + (size_t)q * p.qs;
  // chip-wide threshold, never looser than the block's own (with the bootstrap folded into the prologue the chip-wide word
  // is published by block 0 only, possibly after this wave got here)
  uint32_t t = min(vc_ld_relaxed(tau_q), st[q]);
  const uint64_t lim = p.limit ? p.limit[q] : VC_PACK_INF;   // recovery pass: exact packed bound (ties cannot refill the ring)
  uint64_t* ring = p.buf + (uint64_t)q * p.cap;
  uint32_t* hist = p.hist + (uint64_t)q * p.hist_stride;
  uint32_t seen = 0;   // ring fill after this wave's last append (wave-uniform)
  bool recut = false;  // an append of this wave crossed a VC_SCAN_RECUT_EVERY boundary of the ring fill
#if VC_SCAN_DIAGNOSTICS
  uint32_t dbg_items = 0;
  if (p.dbg && lane == 0) atomicAdd(p.dbg + 0, 1u);
#endif
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint64_t a[W], b[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      a[j] = r[u][j].x;
      b[j] = r[u][j].y;
    }
    const uint64_t ia = chunk_base + (uint64_t)u * 2 * BLK + 2 * tid;
    const uint32_t da = vc_dist<W>(a, qw), db = vc_dist<W>(b, qw);
    const uint64_t pa = vc_pack(da, p.id_base + (uint32_t)ia), pb = vc_pack(db, p.id_base + (uint32_t)ia + 1);
    const bool oka = da <= t && ia < p.n && pa <= lim, okb = db <= t && ia + 1 < p.n && pb <= lim;
    const uint32_t cnt = (uint32_t)oka + (uint32_t)okb;
    if (__ballot(cnt != 0) == 0) continue;
    uint32_t total;
    uint32_t slot = vc_wave_excl_scan(cnt, total);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(p.count + (size_t)q * p.qs, total);
    base = __builtin_amdgcn_readfirstlane(base);
    slot += base;
    seen = base + total;
#if VC_SCAN_DIAGNOSTICS
    dbg_items += total;
#endif
    recut = recut || (base / VC_SCAN_RECUT_EVERY != seen / VC_SCAN_RECUT_EVERY) || (base < p.k && seen >= p.k);
    if (oka) {
      if (slot < p.cap) ring[slot] = pa;
      ++slot;
      atomicAdd(hist + da, 1u);
    }
    if (okb) {
      if (slot < p.cap) ring[slot] = pb;
      atomicAdd(hist + db, 1u);
    }
  }
  // re-derive the chip-wide threshold from the histogram of everything appended so far -- not on every entry: the
  // histogram lines are read coherently by every wave that gets here, and same-line coherent accesses serialise in
  // L2 (~27 ns each), so only the wave whose append crosses a multiple of VC_SCAN_RECUT_EVERY entries does it
#if VC_SCAN_DIAGNOSTICS
  if (p.dbg && lane == 0) {
    if (dbg_items) {
      atomicAdd(p.dbg + 1, 1u);
      atomicAdd(p.dbg + 2, dbg_items);
      atomicAdd(p.dbg + 8 + min((uint32_t)(chunk_base / (2ull * BLK * U) / gridDim.x), 55u), dbg_items);
    }
    if (seen >= p.k && recut) atomicAdd(p.dbg + 3, 1u);
  }
#endif
  if (seen >= p.k && recut) {
    const uint32_t cut = vc_hist_cut(hist, t + 1, p.k, true);
    if (cut < t) {
      t = cut;
      if (lane == 0) atomicMin(tau_q, t);
    }
  }
  if (lane == 0) st[q] = t;
}

// Counted wait for hand-issued tile loads: at most N vector-memory operations younger than the tile may still be in
// flight.  The tile's registers pass through the asm as read-write operands, so every use is ordered behind the wait.
template <int N, int U, int W>
__device__ __forceinline__ void vc_tile_wait(vc_u64x2 (&r)[U][W]) {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63));   // 6-bit counter on gfx9; waiting for fewer is only stricter
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < W; ++j) asm volatile("" : "+v"(r[u][j]));
}

// QT > 0: the "small tile" form for p.qt <= QT <= 8 queries (the HBM-bound passes): the query loop is unrolled at compile
// time, the LDS query words are read at immediate offsets (no address arithmetic, no loop branch, no second register
// set for a software pipeline), and the distances are tested in "matching bits" form: the staged words are the
// COMPLEMENTED query words and each item's accumulate chain starts at tau, so acc = tau + B - dist and
// dist <= tau <=> acc >= B; the 2*U chains of a tile are OR-reduced (v_or3_b32) and tested with one compare -- B is a
// power of two and acc < 2B, so the OR is >= B exactly when some chain is.  69 VALU instructions per query and 8-item
// tile instead of 74 and no per-query loop overhead: 2.5-3.5 % off a whole pass at 8 queries (profiles/r02_sweeps.md).
// (Compiling it for a fifth wave per SIMD -- <= 96 VGPRs, MINW = 5 -- spills into the loop, and a spill reload waits
// vmcnt(0), i.e. for the prefetched tile: no gain, and unsafe next to hand-issued loads.)
template <int W, int U, int BLK, int NB, int QT = 0, int MINW = 0>
__global__ void __launch_bounds__(BLK, MINW ? MINW : (BLK == 512 ? 2 : ((W >= 4 || NB * U * W > 16) ? 3 : VC_SCAN_MIN_WAVES))) vc_scan_kernel(const VcScanParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* sq = (uint64_t*)smem;                             // [qt][W]  query tile
  uint32_t* st = (uint32_t*)(smem + (size_t)(QT ? QT : p.qt) * W * 8);    // [qt]     block-local copy of tau (QT > 0: p.qt == QT)
  for (uint32_t i = threadIdx.x; i < p.qt * W; i += BLK) sq[i] = QT ? ~p.queries[i] : p.queries[i];
  if (QT > 0 && p.shist) {
    // bootstrap threshold = smallest d whose sampled cumulative count reaches k (what vc_tau_init_kernel computes); fewer
    // than k samples = accept everything.  One wave per query; block 0 publishes the chip-wide word (cleaned to ~0).
    for (uint32_t q = threadIdx.x / VC_WAVE; q < p.qt; q += BLK / VC_WAVE) {
      uint32_t cut = vc_hist_cut(p.shist + (uint64_t)q * p.hist_stride, p.bits + 1, p.k, false, p.shist_copies, p.shist_cstride);
      if (cut == 0xFFFFFFFFu) cut = p.bits;
      if ((threadIdx.x & (VC_WAVE - 1)) == 0) {
        st[q] = cut;
        if (blockIdx.x == 0) atomicMin(p.tau + (size_t)q * p.qs, cut);
      }
    }
  } else {
    for (uint32_t i = threadIdx.x; i < p.qt; i += BLK) st[i] = (VC_SCAN_DIAGNOSTICS && p.wrap) ? 0u : vc_ld_relaxed(p.tau + (size_t)i * p.qs);  // wrap: rare path off
  }
  __syncthreads();

  constexpr uint64_t CH = 2ull * BLK * U;
  const VcScanRare rare{p.n, p.tau, p.count, p.hist, p.buf, p.limit, p.id_base, p.k, p.cap, p.hist_stride, p.qs,
                        (VC_SCAN_DIAGNOSTICS && p.trace) ? (uint32_t*)(p.trace + 2 * 8192) : nullptr};
  vc_u64x2 ra[U][W], rb[NB >= 2 ? U : 1][W], rc[NB >= 3 ? U : 1][W];

  // Prefetch cursor: the chunk the next load() fetches.  Tile loads are issued by hand (inline asm): the address is a
  // wave-uniform base (SGPR pair, scalar arithmetic) plus ONE constant 32-bit lane offset, so a chunk's U*W loads
  // cost no VALU address math and no VGPR address pairs.  Left to itself hipcc builds 64-bit VGPR addresses in the
  // registers of the tile it is about to load and then has to wait (vmcnt(1)) for the tile still in flight before it
  // may issue the next one -- which serialises the prefetch exactly when HBM and VALU time are balanced.  The
  // compiler does not track asm loads, so tile_wait<N>() carries the counted s_waitcnt and ties the tile's registers
  // to it (no use can be scheduled above it).  vmcnt retires in order, so the compiler's own waits for its (rare
  // path) loads stay correct -- merely conservative -- with ours interleaved.  Past the block's last chunk the
  // cursor stays put (re-reading that chunk is an L2 hit) so the number of loads in flight is the same on every path.
  const uint64_t G = gridDim.x;
  uint64_t pf = blockIdx.x;
  const uint32_t lane_off = threadIdx.x * (uint32_t)sizeof(vc_u64x2);
  auto load = [&](vc_u64x2(&r)[U][W]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < W; ++j) {
        // (diagnostic build, VC_SCAN_WRAP: the stream wraps onto the first `wrap` chunks = cache resident)
        const uint64_t c = (VC_SCAN_DIAGNOSTICS && p.wrap) ? pf % p.wrap : pf;
        const uint64_t* sbase = p.cols + (uint64_t)j * p.stride + c * CH + (uint64_t)u * 2 * BLK;
        // The base goes through an s_mov inside the asm: hipcc cannot see the VMEM instruction in here, so it would
        // not pad the 5 wait states gfx9 needs between a VALU write of an SGPR (v_readfirstlane, v_cmp) and a VMEM
        // read of it, should it ever compute the base that way; an SGPR written by the SALU has no such hazard.
        const uint64_t* sb;
#if VC_SCAN_NT
        // The stream is non-temporal -- except the first p.resident chunks of the database, read with plain loads: nt
        // loads do not allocate in the 256 MB Infinity Cache, plain ones do, so that prefix (and nothing else the pass
        // reads) stays cache-resident from one pass to the next and costs no HBM traffic after the first.
        if (pf < p.resident)
          asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1" : "=&v"(r[u][j]), "=&s"(sb) : "v"(lane_off), "s"(sbase));
        else
          asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1 nt" : "=&v"(r[u][j]), "=&s"(sb) : "v"(lane_off), "s"(sbase));
#else
        asm volatile("s_mov_b64 %1, %3\n\tglobal_load_dwordx4 %0, %2, %1" : "=&v"(r[u][j]), "=&s"(sb) : "v"(lane_off), "s"(sbase));
#endif
      }
    pf += pf + G < p.nchunks ? G : 0;
  };

  // one query against the register-resident code tile: xor + accumulating v_bcnt per 32-bit word, running min
  // over the tile's 2*U items, ONE compare + branch per query.
  auto one_query = [&](const vc_u64x2(&r)[U][W], const uint64_t(&qv)[W], uint32_t t, uint32_t q, uint64_t chunk) {
#if VC_SCAN_DIAGNOSTICS
    if (p.diag) {  // diagnostic: a wave "computes" for the same wall time without using the VALU (power/overlap study)
      for (uint32_t i = 0; i < p.diag; ++i) __builtin_amdgcn_s_sleep(1);
      return;
    }
#endif
    const uint64_t(&qw)[W] = qv;  // VGPR operands: an SGPR source halves the v_xor rate on gfx950 (measured, tools/ubench_bank.hip)
    // per item pair: W*2 v_xor + W*2 accumulating v_bcnt (two interleaved chains), running min over the tile
    uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t da, db;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        const uint64_t xa = r[u][j].x ^ qw[j], xb = r[u][j].y ^ qw[j];
        da = j ? vc_bcnt_acc((uint32_t)xa, da) : vc_bcnt0((uint32_t)xa);
        db = j ? vc_bcnt_acc((uint32_t)xb, db) : vc_bcnt0((uint32_t)xb);
        da = vc_bcnt_acc((uint32_t)(xa >> 32), da);
        db = vc_bcnt_acc((uint32_t)(xb >> 32), db);
      }
      dmin = min(dmin, min(da, db));
    }
    if (__ballot(dmin <= t) != 0) vc_scan_slow<W, U, BLK>(rare, r, qw, q, chunk * CH, st);
  };

  // The query stream is software-pipelined through two register sets: the ds_reads (wave-uniform address =
  // LDS broadcast) of query q+1 are in flight while query q is verified, so no wave parks on lgkmcnt.
  auto ldq = [&](uint64_t(&qw)[W], uint32_t& t, uint32_t q) {
#pragma unroll
    for (int j = 0; j < W; ++j) qw[j] = sq[q * W + j];
    t = st[q];
  };
  // small-tile form of one query (QT > 0): w = complemented query words, t = tau; true iff some lane of the wave has an
  // item within tau
  auto one_query_c = [&](const vc_u64x2(&r)[U][W], const uint32_t(&w)[2 * W], uint32_t t) -> bool {
    uint32_t orv = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint32_t da = t, db = t;
#pragma unroll
      for (int j = 0; j < W; ++j) {
        da = vc_bcnt_acc((uint32_t)r[u][j].x ^ w[2 * j], da);
        db = vc_bcnt_acc((uint32_t)r[u][j].y ^ w[2 * j], db);
        da = vc_bcnt_acc((uint32_t)(r[u][j].x >> 32) ^ w[2 * j + 1], da);
        db = vc_bcnt_acc((uint32_t)(r[u][j].y >> 32) ^ w[2 * j + 1], db);
      }
      orv |= da | db;
    }
    return __ballot(orv >= 64u * W) != 0;
  };
  auto verify = [&](const vc_u64x2(&r)[U][W], uint64_t chunk) {
    if constexpr (QT > 0) {   // exactly QT queries (p.qt == QT): no guards, no loop
      const uint32_t* sw = (const uint32_t*)sq;
      uint32_t hm = 0;   // wave-uniform mask of the queries with a candidate in this tile: ONE rare-path copy behind the loop
      uint32_t wq[2][2 * W], tq[2];   // the LDS reads of query q + 1 are in flight while query q is verified
#pragma unroll
      for (int j = 0; j < 2 * W; ++j) wq[0][j] = sw[j];
      tq[0] = st[0];
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        if (q + 1 < QT) {
#pragma unroll
          for (int j = 0; j < 2 * W; ++j) wq[(q + 1) & 1][j] = sw[(q + 1) * 2 * W + j];
          tq[(q + 1) & 1] = st[q + 1];
        }
        if (one_query_c(r, wq[q & 1], tq[q & 1])) hm |= 1u << q;
      }
      while (hm) {
        const uint32_t q = (uint32_t)__builtin_ctz(hm);
        hm &= hm - 1u;
        uint64_t qw[W];
#pragma unroll
        for (int j = 0; j < W; ++j) qw[j] = ~sq[q * W + j];
        vc_scan_slow<W, U, BLK>(rare, r, qw, q, chunk * CH, st);
      }
      return;
    }
#if VC_SCAN_QUERY_PREFETCH
    uint64_t qa[W], qb[W];
    uint32_t ta, tb;
    const uint32_t last = p.qt - 1;
    ldq(qa, ta, 0);
    for (uint32_t q = 0; q < p.qt; q += 2) {
      ldq(qb, tb, min(q + 1, last));
      one_query(r, qa, ta, q, chunk);
      ldq(qa, ta, min(q + 2, last));
      if (q + 1 < p.qt) one_query(r, qb, tb, q + 1, chunk);
    }
#else
    for (uint32_t q = 0; q < p.qt; ++q) {   // the other waves of the SIMD cover the LDS latency; saves 2*W+2 VGPRs
      uint64_t qa[W];
      uint32_t ta;
      ldq(qa, ta, q);
      one_query(r, qa, ta, q, chunk);
    }
#endif
  };

  uint64_t chunk = blockIdx.x;
  if (chunk >= p.nchunks) return;
#if VC_SCAN_DIAGNOSTICS
  if (p.trace && threadIdx.x == 0) p.trace[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int T = U * W;  // loads per tile
  if constexpr (NB == 1) {
    // single register buffer: memory latency is covered by the other waves of the SIMD (more of them fit)
    for (; chunk < p.nchunks; chunk += G) {
      load(ra);
      vc_tile_wait<0>(ra);
      verify(ra, chunk);
    }
  } else if constexpr (NB == 2) {
    load(ra);
    for (;;) {
      load(rb);
      vc_tile_wait<T>(ra);
      verify(ra, chunk);
      if ((chunk += G) >= p.nchunks) break;
      load(ra);
      vc_tile_wait<T>(rb);
      verify(rb, chunk);
      if ((chunk += G) >= p.nchunks) break;
    }
  } else {
    // three buffers: two chunks in flight behind the one being verified (absorbs HBM latency jitter)
    load(ra);
    load(rb);
    for (;;) {
      load(rc);
      vc_tile_wait<2 * T>(ra);
      verify(ra, chunk);
      if ((chunk += G) >= p.nchunks) break;
      load(ra);
      vc_tile_wait<2 * T>(rb);
      verify(rb, chunk);
      if ((chunk += G) >= p.nchunks) break;
      load(rb);
      vc_tile_wait<2 * T>(rc);
      verify(rc, chunk);
      if ((chunk += G) >= p.nchunks) break;
    }
  }
#if VC_SCAN_DIAGNOSTICS
  // The cursor's last load (a re-read of the block's final chunk) is never consumed and still in flight here: to the
  // compiler its registers are free, and anything it places in them now is overwritten when the load lands (an exit
  // routine that was tried here -- a call with its pointer argument in v[0:1] -- faulted exactly so).  Code after the
  // loop must drain first; the plain kernel has none.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.trace && threadIdx.x == 0) p.trace[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------
// Select: the k smallest packed values of a per-query candidate set, ascending.
// Source = candidate ring of the scan / MIH verify, or the all-gathered per-shard top-k lists
// (replaces the master-side priority_queue, search_worker.cc:179-199, and
// mpi_coordinator::gather_vectors' consumer).  One 1024-thread block per query.
//   n <= 8192 : whole set bitonic-sorted in LDS.
//   n  > 8192 : 4-pass (11 bit) radix select finds the k-th smallest value, survivors are
//               compacted into LDS and sorted.
// ------------------------------------------------------------------------------------------
struct VcRingSrc {
  const uint64_t* buf;
  const uint32_t* count;
  uint32_t cap;
  const uint32_t* list;   // optional: block b serves ring slot list[b]
  uint32_t mark_overflow; // report count = UINT32_MAX when the ring overflowed (row is then only an upper bound)
  const uint32_t* tau;    // optional final distance thresholds: entries farther than tau[q] cannot be in the top-k
  uint32_t qs;            // words between consecutive queries' entries of count[] and tau[]
  __device__ uint32_t slot(uint32_t b) const { return list ? list[b] : b; }
  // Pre-filter by the final threshold -- but not when the ring overflowed: tau comes from the histogram of ALL
  // survivors, stored or not, so it can lie below every entry that did fit (early arrivals under a loose tau), and
  // the recovery pass needs the k-th best of what fitted as its bound (found by tests/campaign/parity_campaign.py: k = 1,
  // cap = 4, thousands of duplicates -> empty row -> no bound -> "recovery did not converge").
  __device__ uint64_t bound(uint32_t q) const {
    return (tau && count[(size_t)q * qs] <= cap) ? (((uint64_t)tau[(size_t)q * qs] + 1) << 32) : VC_PACK_INF;
  }
  __device__ bool overflowed(uint32_t q) const { return mark_overflow && count[(size_t)q * qs] > cap; }
  __device__ uint32_t size(uint32_t q) const { return min(count[(size_t)q * qs], cap); }
  __device__ uint64_t get(uint32_t q, uint32_t i) const { return buf[(uint64_t)q * cap + i]; }
};
struct VcListsSrc {
  const uint64_t* lists;
  uint32_t n_lists, nq, k;
  __device__ uint32_t slot(uint32_t b) const { return b; }
  __device__ uint64_t bound(uint32_t) const { return VC_PACK_INF; }
  __device__ bool overflowed(uint32_t) const { return false; }
  __device__ uint32_t size(uint32_t) const { return n_lists * k; }
  __device__ uint64_t get(uint32_t q, uint32_t i) const {
    return lists[((uint64_t)(i / k) * nq + q) * k + (i % k)];
  }
};

// The gathered per-shard results of vc_sharded_*: shard g's slot starts g * slot_words words into `base` and holds its
// [nq][k] rows followed, cnt_off (64-bit) words in, by its [nq] counts.  A shard that flags a row (count UINT32_MAX: its ring
// overflowed and the device-side recovery gave up) flags the merged row -- the overflow flags are reduced HERE, on the
// device, instead of a count read-back and a host wait per shard.
struct VcSlotsSrc {
  const uint64_t* base;
  uint64_t slot_words;
  uint32_t cnt_off, n_lists, nq, k;
  __device__ uint32_t slot(uint32_t b) const { return b; }
  __device__ uint64_t bound(uint32_t) const { return VC_PACK_INF; }
  __device__ bool overflowed(uint32_t q) const {
    for (uint32_t g = 0; g < n_lists; ++g)
      if (((const uint32_t*)(base + (uint64_t)g * slot_words + cnt_off))[q] == 0xFFFFFFFFu) return true;
    return false;
  }
  __device__ uint32_t size(uint32_t) const { return n_lists * k; }
  __device__ uint64_t get(uint32_t q, uint32_t i) const { return base[(uint64_t)(i / k) * slot_words + (uint64_t)q * k + (i % k)]; }
};

#define VC_SEL_THREADS 1024
#define VC_RANK_SORT_MAX 1024u

template <class Src>
__global__ void __launch_bounds__(VC_SEL_THREADS) vc_select_kernel(Src src, uint32_t k, uint64_t* __restrict__ out,
                                                                   uint32_t* __restrict__ out_count) {
  __shared__ uint64_t a[VC_SORT_CAP];
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t s_prefix_hi, s_prefix_lo, s_rank, s_fill, s_valid;
  const uint32_t q = src.slot(blockIdx.x);
  const uint32_t n = src.size(q);
  uint32_t P;

  if (n <= VC_SORT_CAP) {
    // compact the entries that can still matter (below the final threshold when the source has one), then sort those
    const uint64_t bound = src.bound(q);
    if (threadIdx.x == 0) s_fill = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += VC_SEL_THREADS) {
      const uint64_t v = src.get(q, i);
      if (v < bound) a[atomicAdd(&s_fill, 1u)] = v;
    }
    __syncthreads();
    const uint32_t fill = s_fill;
    P = 2;
    while (P < fill) P <<= 1;
    for (uint32_t i = fill + threadIdx.x; i < P; i += VC_SEL_THREADS) a[i] = VC_PACK_INF;
  } else {
    // ---- radix select on the 44 low bits (dist < 2048, id 32 bit); padding (INF) never participates
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    uint32_t myvalid = 0;
    for (uint32_t i = threadIdx.x; i < n; i += VC_SEL_THREADS) myvalid += src.get(q, i) != VC_PACK_INF;
    atomicAdd(&s_valid, myvalid);
    __syncthreads();
    const uint32_t valid = s_valid;
    uint64_t thresh = VC_PACK_INF - 1;  // valid <= k: keep every valid entry
    if (valid > k) {
      if (threadIdx.x == 0) {
        s_prefix_hi = 0;
        s_prefix_lo = 0;
        s_rank = k - 1;  // 0-based rank of the k-th smallest
      }
      uint64_t prefix = 0;
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 33 - 11 * pass;
        for (uint32_t i = threadIdx.x; i < 2048; i += VC_SEL_THREADS) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += VC_SEL_THREADS) {
          const uint64_t v = src.get(q, i);
          if (v != VC_PACK_INF && (v >> (shift + 11)) == prefix) atomicAdd(&hist[(uint32_t)(v >> shift) & 0x7FFu], 1u);
        }
        __syncthreads();
        if (threadIdx.x < VC_WAVE) {  // wave 0 locates the digit holding rank s_rank
          const uint32_t lane = threadIdx.x;
          uint32_t mine = 0;
          for (uint32_t i = 0; i < 32; ++i) mine += hist[lane * 32 + i];
          uint32_t total;
          uint32_t run = vc_wave_excl_scan(mine, total);
          const uint32_t rank = s_rank;
          uint32_t found = 0xFFFFFFFFu, before = 0;
          for (uint32_t i = 0; i < 32; ++i) {
            const uint32_t c = hist[lane * 32 + i];
            if (found == 0xFFFFFFFFu && rank >= run && rank < run + c) {
              found = lane * 32 + i;
              before = run;
            }
            run += c;
          }
          if (found != 0xFFFFFFFFu) {  // exactly one lane
            const uint64_t np = (prefix << 11) | found;
            s_prefix_hi = (uint32_t)(np >> 32);
            s_prefix_lo = (uint32_t)np;
            s_rank = rank - before;
          }
        }
        __syncthreads();
        prefix = ((uint64_t)s_prefix_hi << 32) | s_prefix_lo;
      }
      thresh = prefix;  // exact value of the k-th smallest
    }
    if (threadIdx.x == 0) s_fill = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += VC_SEL_THREADS) {
      const uint64_t v = src.get(q, i);
      if (v <= thresh) {
        const uint32_t slot = atomicAdd(&s_fill, 1u);
        if (slot < VC_SORT_CAP) a[slot] = v;
      }
    }
    __syncthreads();
    const uint32_t fill = min(s_fill, (uint32_t)VC_SORT_CAP);
    P = 2;
    while (P < fill) P <<= 1;
    for (uint32_t i = fill + threadIdx.x; i < P; i += VC_SEL_THREADS) a[i] = VC_PACK_INF;
  }
  __shared__ uint64_t srt[VC_RANK_SORT_MAX];
  // More than VC_RANK_SORT_MAX entries but only k wanted (an MIH shell with a few thousand candidates): cut by
  // distance first -- histogram of the distances in LDS, smallest d whose cumulative count reaches k -- and keep the
  // entries up to that distance; they usually fit the counting sort below (8 us) where the full bitonic network of
  // 2048-8192 entries takes 25 us.
  if (P > VC_RANK_SORT_MAX) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2048; i += VC_SEL_THREADS) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < P; i += VC_SEL_THREADS)
      if (a[i] != VC_PACK_INF) atomicAdd(&hist[min((uint32_t)(a[i] >> 32), 2047u)], 1u);
    __syncthreads();
    if (threadIdx.x < VC_WAVE) {   // wave 0: 32 bins per lane
      const uint32_t lane = threadIdx.x;
      uint32_t mine = 0;
      for (uint32_t i = 0; i < 32; ++i) mine += hist[lane * 32 + i];
      uint32_t total;
      uint32_t run = vc_wave_excl_scan(mine, total);
      uint32_t cut = 0xFFFFFFFFu, upto = 0;
      for (uint32_t i = 0; i < 32; ++i) {
        run += hist[lane * 32 + i];
        if (run >= k && cut == 0xFFFFFFFFu) {
          cut = lane * 32 + i;
          upto = run;
        }
      }
      const uint32_t best = vc_wave_min(cut);
      if (best == 0xFFFFFFFFu) {         // fewer than k entries in total: keep everything
        if (lane == 0) { s_prefix_lo = 2047u; s_rank = total; }
      } else if (cut == best) {          // exactly one lane holds the cut bin
        s_prefix_lo = best;
        s_rank = upto;
      }
    }
    __syncthreads();
    const uint32_t dcut = s_prefix_lo, kept = s_rank;
    if (dcut < 2047u && kept <= VC_RANK_SORT_MAX) {   // (bin 2047 collects every larger distance: not a usable cut)
      if (threadIdx.x == 0) s_fill = 0;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < P; i += VC_SEL_THREADS) {
        const uint64_t v = a[i];
        if (v != VC_PACK_INF && (uint32_t)(v >> 32) <= dcut) srt[atomicAdd(&s_fill, 1u)] = v;
      }
      __syncthreads();
      P = 2;
      while (P < kept) P <<= 1;
      for (uint32_t i = threadIdx.x; i < P; i += VC_SEL_THREADS) a[i] = i < kept ? srt[i] : VC_PACK_INF;
    }
  }
  // Small survivor sets (the usual case: a few hundred entries) are ordered by counting: entry i goes to slot
  // #{j : a[j] < a[i]} -- packed values are distinct -- which is P broadcast LDS reads per thread and one barrier
  // instead of the ~log^2 P barriers of the bitonic network.
  const uint64_t* sorted = a;
  if (P <= VC_RANK_SORT_MAX) {
    // All 1024 threads count: VC_SEL_THREADS / P threads share an entry, each walks its slice of a[] (the usual few hundred
    // survivors used to leave three quarters of the block idle while P threads walked all of a[]: 4 of the kernel's 10 us)
    // and the partial ranks meet in hist[] (free here).
    __syncthreads();
    const uint32_t parts = min(VC_SEL_THREADS / P, P);         // P is a power of two, 2 .. 1024: 1 <= parts <= P, both powers of two
    const uint32_t i = threadIdx.x & (P - 1u), part = threadIdx.x / P;
    const uint32_t j0 = part * (P / parts), j1 = j0 + P / parts;
    if (threadIdx.x < P) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t v = a[i];
    if (part >= parts) {
      // (more threads than entry x slice pairs: nothing to do)
    } else if (v != VC_PACK_INF) {
      uint32_t r = 0;
      for (uint32_t j = j0; j < j1; ++j) r += a[j] < v || (a[j] == v && j < i);   // (equal values -- gathered lists that overlap -- keep distinct slots)
      if (parts > 1) atomicAdd(&hist[i], r); else hist[i] = r;
    } else if (part == 0) {
      hist[i] = i;   // padding sits at the tail of a[] already (index >= fill) and stays there
    }
    __syncthreads();
    if (threadIdx.x < P) srt[hist[threadIdx.x]] = a[threadIdx.x];
    __syncthreads();
    sorted = srt;
  } else {
    vc_bitonic_lds(a, P, VC_SEL_THREADS);
  }
  if (threadIdx.x == 0) s_valid = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t i = threadIdx.x; i < k; i += VC_SEL_THREADS) {
    const uint64_t v = i < P ? sorted[i] : VC_PACK_INF;
    out[(uint64_t)q * k + i] = v;
    mine += v != VC_PACK_INF;
  }
  if (out_count) {
    atomicAdd(&s_valid, mine);
    __syncthreads();
    if (threadIdx.x == 0) out_count[q] = src.overflowed(q) ? 0xFFFFFFFFu : s_valid;
  }
}

// ------------------------------------------------------------------------------------------
// Ring-overflow recovery on the device (no host round trip, so the device-pointer API stays exact and asynchronous).
//
// A ring overflows when more than `cap` items lie at or below the k-th distance D: fewer than k items are nearer
// than D (all of them are wanted) and thousands tie at exactly D, of which the canonical contract keeps the
// need = k - #{d < D} SMALLEST ids (linear_search.cc:53 never replaces an equal-distance item, so the reference keeps
// early = low ids too).  ids are positions (id = id_base + position, build_hash_tables.cc:55,61), so the wanted tie
// with the largest id is found by a radix select over the POSITION of the ties:
//   * D and need come from the scan's distance histogram, which is exact for every d <= D (the running threshold
//     never drops below D, so every such item was counted whether its ring slot existed or not);
//   * level 0 histograms position >> s0 of every tie over the whole database (2048 bins), level 1 / 2 refine inside
//     the bin that holds the need-th tie (ranges of n/2048, n/2048^2 items) down to one position p*;
//   * one more pass appends every item with (dist, id) <= (D, id_base + p*): exactly k of them, so the ring cannot
//     overflow again, and one block per query sorts them into the output row.
// Two full passes over the codes + two short ones, whatever the ring size, against the 2..200 host-driven rounds
// of linear_recover() (vc_engine.hip), which stays as the fallback.
//
// One persistent launch after every select: without an overflowed query each block reads the ring cursors and
// leaves (a ~2 us launch); otherwise the passes are separated by grid barriers, so every block must be resident
// (2 blocks of 256 threads per CU, 32 KiB of LDS each -- 64 KiB only for k > 4096) and every spin is bounded: if the grid cannot meet (another
// kernel holds the CUs for seconds) the blocks give up, the row keeps its UINT32_MAX count and the sticky
// `gave_up` counter reports it (vc_device_status).
// ------------------------------------------------------------------------------------------
#define VC_REC_RQ 4u            // overflowed queries recovered per round of passes (they share the two full sweeps)
#define VC_REC_BINS 2048u
#define VC_REC_MAXQ 64u         // = VC_GROUP_QUERIES: queries one select / recover launch serves
#define VC_REC_SPIN_LIMIT (3u << 20)   // x s_sleep 32 (~0.9 us): ~3 s

struct VcRecoverParams {
  const uint64_t* cols;
  uint64_t stride, n;
  const uint64_t* queries;   // [nq][W]
  const uint32_t* count;     // [nq] raw ring cursors of the scan (one per 128-byte line, `qs` words apart)
  const uint32_t* hist;      // [nq][hist_stride] distance histogram of the scan
  uint64_t* ring;            // [nq][cap]
  uint32_t* idhist;          // [VC_REC_MAXQ][3][VC_REC_BINS] scratch, zeroed in here when needed
  uint32_t* rcount;          // [VC_REC_MAXQ] ring cursors of the append pass, 32 words apart
  uint32_t* bar;             // [0] arrivals, [32] give-up flag, [64] departures: all zero between launches
  uint32_t* gave_up;         // sticky count of launches that gave up
  uint64_t* out;             // [nq][k]
  uint32_t* out_count;       // [nq]
  uint32_t nq, k, cap, hist_stride, qs, bits, id_base;
  // per-step state this launch hands back zeroed (the next search call then needs no memset): ring cursors and
  // thresholds (one line per query), the scan's distance histogram, the bootstrap's partial histograms
  uint32_t* clean_count;     // [nq] lines of qs words
  uint32_t* clean_tau;       // [nq] lines of qs words
  uint32_t* clean_shist;     // clean_copies x [.. nq rows of hist_stride ..], copy stride clean_copy_stride words
  uint64_t clean_copy_stride;
  uint32_t clean_copies;     // 0 = leave the state alone
  uint32_t spin_limit;       // bounded spin of the grid barrier, x s_sleep 32 (~0.9 us)
  uint32_t absent;           // test knob (VC_RECOVER_TEST_FAIL): blocks the barrier waits for in vain -> forced give-up
};

// cut of a histogram plus the cumulative count below the cut bin (one wave; same result in every lane)
__device__ __forceinline__ uint32_t vc_hist_cut_below(const uint32_t* h, uint32_t nbins, uint32_t k, uint32_t& below) {
  const uint32_t lane = vc_lane();
  const uint32_t bpl = (nbins + VC_WAVE - 1) / VC_WAVE;
  uint32_t mine = 0;
  for (uint32_t i = 0; i < bpl; ++i) {
    const uint32_t bin = lane * bpl + i;
    if (bin < nbins) mine += vc_ld_relaxed(h + bin);
  }
  uint32_t total;
  uint32_t run = vc_wave_excl_scan(mine, total);
  uint32_t cand = 0xFFFFFFFFu, before = 0;
  for (uint32_t i = 0; i < bpl; ++i) {
    const uint32_t bin = lane * bpl + i;
    if (bin < nbins) {
      const uint32_t c = vc_ld_relaxed(h + bin);
      if (run + c >= k && cand == 0xFFFFFFFFu) {
        cand = bin;
        before = run;
      }
      run += c;
    }
  }
  const uint32_t cut = vc_wave_min(cand);
  const uint64_t owner = __ballot(cand == cut && cut != 0xFFFFFFFFu);
  below = owner ? __shfl(before, __ffsll((long long)owner) - 1, VC_WAVE) : 0u;
  return cut;
}

// grid barrier with a bounded spin; false = some block gave up (every block then leaves)
__device__ __forceinline__ bool vc_rec_barrier(uint32_t* bar, uint32_t& epoch, uint32_t* s_ok, uint32_t spin_limit,
                                               uint32_t absent, uint32_t* gave_up) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the workgroup barrier
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-back must have landed before the arrival shows (hipcc may drop its own wait)
    ++epoch;
    atomicAdd(&bar[0], 1u);
    const uint32_t target = epoch * (gridDim.x + absent);
    bool ok = vc_ld_relaxed(&bar[32]) == 0;   // a block that becomes resident after the others gave up leaves at once
    for (uint32_t spins = 0; ok && vc_ld_relaxed(&bar[0]) < target; ++spins) {
      __builtin_amdgcn_s_sleep(32);
      if (spins > spin_limit || vc_ld_relaxed(&bar[32])) ok = false;
    }
    if (!ok && atomicExch(&bar[32], 1u) == 0u) atomicAdd(gave_up, 1u);   // counted once, when the flag is first set
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate completes asynchronously: hold the barrier until it has
    *s_ok = ok ? 1u : 0u;
  }
  __syncthreads();
  return *s_ok != 0;
}

template <int W>
__global__ void __launch_bounds__(256, 2) vc_recover_kernel(const VcRecoverParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* lh = (uint32_t*)smem;   // [VC_REC_RQ][VC_REC_BINS] block-local position histograms; later the sort buffer
  __shared__ uint32_t s_list[VC_REC_MAXQ];
  __shared__ uint32_t s_nover, s_ok;
  __shared__ uint64_t s_q[VC_REC_RQ][W];
  __shared__ uint32_t s_D[VC_REC_RQ], s_need[VC_REC_RQ];
  __shared__ uint64_t s_lo[VC_REC_RQ], s_lim[VC_REC_RQ];
  const uint32_t lane = vc_lane(), wave = threadIdx.x / VC_WAVE;

  // ---- which queries overflowed?  (ring cursors of the finished scan: the same answer in every block)
  if (threadIdx.x < VC_WAVE) {
    const bool over = threadIdx.x < p.nq && p.count[(size_t)threadIdx.x * p.qs] > p.cap;
    const uint64_t m = __ballot(over);
    if (over) s_list[__popcll(m & ((1ull << lane) - 1ull))] = threadIdx.x;
    if (lane == 0) s_nover = (uint32_t)__popcll(m);
  }
  __syncthreads();
  const uint32_t n_over = s_nover;
  // Last kernel of a search step: hand the step's state back zeroed.  Nothing reads it any more once the overflow
  // list is known (no overflow), or once the last round's barriers are behind (below).
  auto clean_state = [&](bool alone) {
    if (p.clean_copies == 0) return;
    const uint64_t gtid = alone ? threadIdx.x : (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t gsz = alone ? blockDim.x : (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = gtid; i < (uint64_t)p.nq * p.qs; i += gsz) {
      const_cast<uint32_t*>(p.count)[i] = 0;
      p.clean_tau[i] = 0xFFFFFFFFu;   // "no threshold yet": the verify prologue publishes with atomicMin
    }
    const uint64_t row = (uint64_t)p.nq * p.hist_stride;
    for (uint64_t i = gtid; i < row; i += gsz) const_cast<uint32_t*>(p.hist)[i] = 0;
    for (uint64_t i = gtid; i < row * p.clean_copies; i += gsz) p.clean_shist[(i / row) * p.clean_copy_stride + i % row] = 0;
  };
  if (n_over == 0) {
    clean_state(false);
    return;
  }

  uint32_t epoch = 0;
  bool alive = true;
  // scratch of this launch: position histograms and append cursors of every overflowed query
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)n_over * 3 * VC_REC_BINS; i += (uint64_t)gridDim.x * blockDim.x)
    p.idhist[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x < n_over) p.rcount[threadIdx.x * 32] = 0;
  alive = vc_rec_barrier(p.bar, epoch, &s_ok, p.spin_limit, p.absent, p.gave_up);

  // position bits and the digit shifts of the (up to three) levels
  uint32_t P = 1;
  while (P < 32 && (1ull << P) < p.n) ++P;
  const uint32_t nlev = P <= 11 ? 1u : (P <= 22 ? 2u : 3u);
  const uint32_t shifts[3] = {P > 11 ? P - 11 : 0u, P > 22 ? P - 22 : 0u, 0u};

  // one sweep over positions [lo, hi) for the queries [rb, re) of the round.  count: ties (d == D) are histogrammed
  // by (position - base) >> shift into lh[r]; append: items with packed <= s_lim[r] go to the query's ring.
  auto sweep = [&](uint64_t lo, uint64_t hi, uint32_t rb, uint32_t re, bool append, uint32_t shift, uint32_t slot0) {
    if (hi > p.n) hi = p.n;
    if (lo >= hi) return;
    for (uint64_t c = lo / 2048 + blockIdx.x; c * 2048 < hi; c += gridDim.x) {
      vc_u64x2 v[4][W];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < W; ++j)
          v[u][j] = __builtin_nontemporal_load(
              reinterpret_cast<const vc_u64x2*>(p.cols + (uint64_t)j * p.stride + c * 2048 + (uint64_t)u * 512 + 2 * threadIdx.x));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t ia = c * 2048 + (uint64_t)u * 512 + 2 * threadIdx.x;
        uint64_t a[W], b[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
          a[j] = v[u][j].x;
          b[j] = v[u][j].y;
        }
        const bool va = ia >= lo && ia < hi, vb = ia + 1 >= lo && ia + 1 < hi;
        for (uint32_t r = rb; r < re; ++r) {
          uint64_t qw[W];
#pragma unroll
          for (int j = 0; j < W; ++j) qw[j] = s_q[r][j];
          const uint32_t da = vc_dist<W>(a, qw), db = vc_dist<W>(b, qw);
          if (!append) {
            const uint32_t D = s_D[r];
            const uint64_t base = s_lo[r];
            if (va && da == D) atomicAdd(&lh[r * VC_REC_BINS + (uint32_t)((ia - base) >> shift)], 1u);
            if (vb && db == D) atomicAdd(&lh[r * VC_REC_BINS + (uint32_t)((ia + 1 - base) >> shift)], 1u);
          } else {
            const uint64_t lim = s_lim[r];
            const uint64_t pa = vc_pack(da, p.id_base + (uint32_t)ia), pb = vc_pack(db, p.id_base + (uint32_t)ia + 1);
            const bool oka = va && pa <= lim, okb = vb && pb <= lim;
            const uint32_t cnt = (uint32_t)oka + (uint32_t)okb;
            if (__ballot(cnt != 0) == 0) continue;
            uint32_t total;
            uint32_t pos = vc_wave_excl_scan(cnt, total);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&p.rcount[(slot0 + r) * 32], total);
            pos += __builtin_amdgcn_readfirstlane(base);
            uint64_t* ring = p.ring + (uint64_t)s_list[slot0 + r] * p.cap;
            if (oka) {
              if (pos < p.cap) ring[pos] = pa;
              ++pos;
            }
            if (okb && pos < p.cap) ring[pos] = pb;
          }
        }
      }
    }
  };
  // flush lh[r] into the global histogram of (query slot, level), then clear it
  auto flush = [&](uint32_t nr, uint32_t slot0, uint32_t level) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nr * VC_REC_BINS; i += blockDim.x) {
      const uint32_t c = lh[i];
      if (c) atomicAdd(&p.idhist[((uint64_t)(slot0 + i / VC_REC_BINS) * 3 + level) * VC_REC_BINS + i % VC_REC_BINS], c);
      lh[i] = 0;
    }
    __syncthreads();
  };
  // after the barrier: every block locates the bin holding the need-th tie of each query of the round (one wave per query)
  auto locate = [&](uint32_t nr, uint32_t slot0, uint32_t level, uint32_t shift) {
    for (uint32_t r = wave; r < nr; r += blockDim.x / VC_WAVE) {
      uint32_t before;
      const uint32_t b = vc_hist_cut_below(p.idhist + ((uint64_t)(slot0 + r) * 3 + level) * VC_REC_BINS, VC_REC_BINS, s_need[r], before);
      if (lane == 0) {
        if (b == 0xFFFFFFFFu) {   // cannot happen (the ties were counted by the scan); poison the limit so the row stays flagged
          s_need[r] = 0xFFFFFFFFu;
        } else {
          s_lo[r] += (uint64_t)b << shift;
          s_need[r] -= before;
        }
      }
    }
    __syncthreads();
  };

  for (uint32_t r0 = 0; r0 < n_over && alive; r0 += VC_REC_RQ) {
    const uint32_t nr = min(VC_REC_RQ, n_over - r0);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nr * W; i += blockDim.x) s_q[i / W][i % W] = p.queries[(uint64_t)s_list[r0 + i / W] * W + i % W];
    for (uint32_t i = threadIdx.x; i < VC_REC_RQ * VC_REC_BINS; i += blockDim.x) lh[i] = 0;
    for (uint32_t r = wave; r < nr; r += blockDim.x / VC_WAVE) {   // k-th distance D and the ties wanted at D
      uint32_t below;
      const uint32_t D = vc_hist_cut_below(p.hist + (uint64_t)s_list[r0 + r] * p.hist_stride, p.bits + 1, p.k, below);
      if (lane == 0) {
        s_D[r] = D;
        s_need[r] = p.k - below;
        s_lo[r] = 0;
      }
    }
    __syncthreads();
    // level 0: the whole database once for all queries of the round
    sweep(0, p.n, 0, nr, false, shifts[0], r0);
    flush(nr, r0, 0);
    if (!(alive = vc_rec_barrier(p.bar, epoch, &s_ok, p.spin_limit, p.absent, p.gave_up))) break;
    locate(nr, r0, 0, shifts[0]);
    // deeper levels: only the bin that holds the need-th tie, per query
    for (uint32_t l = 1; l < nlev && alive; ++l) {
      for (uint32_t r = 0; r < nr; ++r) sweep(s_lo[r], s_lo[r] + (1ull << shifts[l - 1]), r, r + 1, false, shifts[l], r0);
      flush(nr, r0, l);
      if (!(alive = vc_rec_barrier(p.bar, epoch, &s_ok, p.spin_limit, p.absent, p.gave_up))) break;
      locate(nr, r0, l, shifts[l]);
    }
    if (!alive) break;
    // s_lo[r] is now the position of the last wanted tie: append everything at or below (D, id_base + position)
    if (threadIdx.x < nr)
      s_lim[threadIdx.x] = s_need[threadIdx.x] == 0xFFFFFFFFu ? 0ull : vc_pack(s_D[threadIdx.x], p.id_base + (uint32_t)s_lo[threadIdx.x]);
    __syncthreads();
    sweep(0, p.n, 0, nr, true, 0, r0);
    if (!(alive = vc_rec_barrier(p.bar, epoch, &s_ok, p.spin_limit, p.absent, p.gave_up))) break;
    // block r sorts query r's k entries into its output row
    if (blockIdx.x < nr) {
      const uint32_t r = blockIdx.x, q = s_list[r0 + r];
      const uint32_t got = vc_ld_relaxed(&p.rcount[(r0 + r) * 32]);
      if (got == p.k && p.k <= VC_SORT_CAP) {   // anything else would be a logic error: leave the row flagged
        uint64_t* a = (uint64_t*)smem;
        uint32_t PP = 2;
        while (PP < got) PP <<= 1;
        for (uint32_t i = threadIdx.x; i < PP; i += blockDim.x)
          a[i] = i < got ? __hip_atomic_load(p.ring + (uint64_t)q * p.cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : VC_PACK_INF;
        vc_bitonic_lds(a, PP, blockDim.x);
        for (uint32_t i = threadIdx.x; i < p.k; i += blockDim.x) p.out[(uint64_t)q * p.k + i] = a[i];
        if (threadIdx.x == 0) p.out_count[q] = got;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < VC_REC_RQ * VC_REC_BINS; i += blockDim.x) lh[i] = 0;
      }
    }
  }
  // Leaving.  Completed: every block has passed the last barrier, nobody reads the step state any more, the grid cleans
  // it together.  Given up: blocks that only become resident now must still find the overflow (n_over > 0 above) so that
  // they run into the barrier's give-up flag and are counted here -- so nothing is cleaned until the LAST block leaves,
  // and that block cleans alone.  The last block also restores the barrier words.
  __syncthreads();
  if (alive) clean_state(false);
  __shared__ uint32_t s_last;
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(&p.bar[64], 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (s_last) {
    if (!alive) clean_state(true);
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicExch(&p.bar[0], 0u);
      atomicExch(&p.bar[32], 0u);
      atomicExch(&p.bar[64], 0u);
    }
  }
}

// persistent grid: every block must be resident (a block that waits for a slot would start when the others
// have almost finished), so grid = CUs x blocks the kernel's registers/LDS admit per CU, capped by the request
template <class K>
uint32_t resident_grid(K kernel, int blk, size_t lds, uint32_t n_cu, uint32_t want) {
  // the occupancy answer depends on (kernel, block, LDS) only: remember it per calling thread instead of asking
  // the runtime on every launch (it is on the per-step host path)
  struct Key { const void* k; int blk; size_t lds; int per_cu; };
  static thread_local std::vector<Key> cache;
  int per_cu = 0;
  for (const Key& c : cache)
    if (c.k == (const void*)kernel && c.blk == blk && c.lds == lds) per_cu = c.per_cu;
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, blk, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    cache.push_back(Key{(const void*)kernel, blk, lds, per_cu});
  }
  const uint32_t cap = n_cu * (uint32_t)per_cu;
  return want ? std::min(want, cap) : cap;
}

template <int W>
hipError_t launch_scan_w(const VcScanParams& p, const VcScanShape& sh, size_t lds, uint32_t n_cu, uint32_t want, hipStream_t s) {
#define VC_LAUNCH_QT(NB_, U_, QT_, MW_)                                                                     \
  {                                                                                                         \
    auto kern = vc_scan_kernel<W, U_, 256, NB_, QT_, MW_>;                                                  \
    const size_t lds_q = (size_t)QT_ * (W * 8 + 4);                                                         \
    const uint32_t grid = (uint32_t)std::min<uint64_t>(p.nchunks, resident_grid(kern, 256, lds_q, n_cu, want)); \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_q, s, p);                                           \
    return hipGetLastError();                                                                               \
  }
  // small tiles (<= 8 queries; the diagnostic knobs keep the general form): compile-time query loop, one kernel per count
  if (sh.small && sh.blk == 256 && sh.dbuf == 2 && p.qt <= 8 && !(VC_SCAN_DIAGNOSTICS && (p.wrap || p.diag))) {
    constexpr int UD = W <= 2 ? 4 : (W <= 4 ? 2 : 1);
    constexpr int MW = W <= 2 ? VC_SCAN_SMALL_WAVES : 0;
    if (sh.unroll == UD && sh.dbuf == 2) {
      switch (p.qt) {
        case 1: VC_LAUNCH_QT(2, UD, 1, MW)
        case 2: VC_LAUNCH_QT(2, UD, 2, MW)
        case 3: VC_LAUNCH_QT(2, UD, 3, MW)
        case 4: VC_LAUNCH_QT(2, UD, 4, MW)
        case 5: VC_LAUNCH_QT(2, UD, 5, MW)
        case 6: VC_LAUNCH_QT(2, UD, 6, MW)
        case 7: VC_LAUNCH_QT(2, UD, 7, MW)
        default: VC_LAUNCH_QT(2, UD, 8, MW)
      }
    }
  }
#undef VC_LAUNCH_QT
#define VC_LAUNCH(NB_, U_, B_)                                                                              \
  {                                                                                                         \
    auto kern = vc_scan_kernel<W, U_, B_, NB_>;                                                             \
    const uint32_t grid = (uint32_t)std::min<uint64_t>(p.nchunks, resident_grid(kern, B_, lds, n_cu, want)); \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(B_), lds, s, p);                                              \
  }
  // (shapes with more than 8 tile loads per lane and buffer are never picked -- vc_scan_pick_shape -- and not built:
  // they spill, and a spilling kernel is unsafe next to hand-issued loads)
#define VC_SCAN_CASE(U_, B_)                                                                     \
  if constexpr (U_ * W <= 8) {                                                                   \
    if (sh.unroll == U_ && sh.blk == B_) {                                                       \
      if (sh.dbuf == 3) VC_LAUNCH(3, U_, B_)                                                     \
      else if (sh.dbuf == 2) VC_LAUNCH(2, U_, B_)                                                \
      else VC_LAUNCH(1, U_, B_)                                                                  \
      return hipGetLastError();                                                                  \
    }                                                                                            \
  }
  VC_SCAN_CASE(4, 256)
  VC_SCAN_CASE(2, 256)
  VC_SCAN_CASE(1, 256)
  VC_SCAN_CASE(4, 512)
  VC_SCAN_CASE(2, 512)
  VC_SCAN_CASE(1, 512)
#undef VC_SCAN_CASE
#undef VC_LAUNCH
  return hipErrorInvalidValue;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
VcScanShape vc_scan_pick_shape(uint32_t W, uint32_t qt, size_t* lds_bytes, const VcKnobs* knobs, uint64_t n_items) {
  const size_t lds = (size_t)qt * (W * 8 + 4);
  if (lds_bytes) *lds_bytes = lds;
  VcScanShape sh;
  // 16 B loads in flight per thread per buffer = U*W; keep the two register buffers <= 64 VGPRs each.
  sh.unroll = W <= 2 ? 4 : (W <= 4 ? 2 : 1);
  // big LDS tiles leave room for one block per CU only: use 512 threads to keep 2 waves per SIMD.
  sh.blk = lds > 40 * 1024 ? 512 : 256;
  sh.dbuf = 2;
  sh.small = knobs ? knobs->scan_small : 1;
  // A small database (n_items given): fewer items per lane, so that its chunks reach every wave of the grid -- 2^20 codes in
  // chunks of 2048 items are 512 blocks' worth, two waves per SIMD walking 200 queries each with nothing to hide their chains
  // behind (configs[0]: 0.143 -> 0.077 ms per pass with one item pair per lane)
  while (n_items && sh.unroll > 1 && n_items < sh.chunk_items() * 2048) sh.unroll >>= 1;
  if (knobs && knobs->shape_set) {  // dev knob VC_SCAN_SHAPE "U,BLK,DB" (read at vc_create)
    const int u = knobs->shape_u ? knobs->shape_u : sh.unroll, b = knobs->shape_blk ? knobs->shape_blk : sh.blk, d = knobs->shape_db;
    if ((u == 1 || u == 2 || u == 4) && u * (int)W <= 8) sh.unroll = u;
    if (b == 256 || b == 512) sh.blk = b;
    sh.dbuf = d <= 0 ? 1 : (d >= 3 ? 3 : (d == 1 ? 2 : d));   // 0 -> 1 buffer, 1/2 -> 2, 3 -> 3
  }
  return sh;
}

// milliseconds of the fastest of three timed passes (after one warm-up pass) of the stream probe, < 0 on error
float vc_probe_stream_ms(const uint64_t* cols, uint64_t stride, uint32_t W, uint64_t items, uint64_t* d_sink, uint32_t n_cu,
                         hipStream_t s) {
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess) return -1.f;
  if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1.f; }
  const uint64_t nchunks = items / 2048;
  float best = -1.f;
  for (int rep = 0; rep < 4 && nchunks; ++rep) {
    (void)hipEventRecord(a, s);
    switch (W) {
      case 2: hipLaunchKernelGGL((vc_stream_probe_kernel<2>), dim3(n_cu * 4), dim3(256), 0, s, cols, stride, nchunks, d_sink); break;
      case 4: hipLaunchKernelGGL((vc_stream_probe_kernel<4>), dim3(n_cu * 4), dim3(256), 0, s, cols, stride, nchunks, d_sink); break;
      case 8: hipLaunchKernelGGL((vc_stream_probe_kernel<8>), dim3(n_cu * 4), dim3(256), 0, s, cols, stride, nchunks, d_sink); break;
      default: (void)hipEventDestroy(a); (void)hipEventDestroy(b); return -1.f;
    }
    (void)hipEventRecord(b, s);
    float ms = 0;
    if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) { best = -1.f; break; }
    if (rep && (best < 0 || ms < best)) best = ms;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return best;
}

hipError_t vc_launch_fill_synth(uint64_t* cols, uint64_t stride, uint32_t W, uint64_t first_local, uint64_t n,
                                uint64_t first_gid, uint64_t seed, uint32_t kind, uint32_t n_centres,
                                uint32_t max_flips, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(vc_fill_synth_kernel, dim3(grid), dim3(256), 0, s, cols, stride, W, first_local, n, first_gid,
                     seed, kind, n_centres ? n_centres : 1u, max_flips);
  return hipGetLastError();
}

hipError_t vc_launch_rows_to_cols(const uint64_t* rows, uint64_t* cols, uint64_t stride, uint32_t W,
                                  uint64_t first_local, uint64_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n * W + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(vc_rows_to_cols_kernel, dim3(grid), dim3(256), 0, s, rows, cols, stride, W, first_local, n);
  return hipGetLastError();
}

hipError_t vc_launch_gather_rows(const uint64_t* cols, uint64_t stride, uint32_t W, const uint32_t* d_local_ids,
                                 uint32_t n_ids, uint64_t* d_rows, hipStream_t s) {
  if (n_ids == 0) return hipSuccess;
  const uint32_t grid = std::min<uint32_t>((n_ids * W + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(vc_gather_rows_kernel, dim3(grid), dim3(256), 0, s, cols, stride, W, d_local_ids, n_ids, d_rows);
  return hipGetLastError();
}

hipError_t vc_launch_sample_hist(const uint64_t* cols, uint64_t stride, uint32_t W, uint64_t s_items,
                                 const uint64_t* d_queries, uint32_t qt, uint32_t* d_shist, uint32_t hist_stride,
                                 uint32_t k, uint32_t bits, uint32_t* d_tau, uint32_t qs, bool refine, uint32_t n_cu,
                                 uint32_t blocks_per_cu, hipStream_t s, bool cut) {
  if (qt == 0) return hipSuccess;
  if (s_items == 0) {
    if (!cut) return hipSuccess;   // the consumer cuts the (all-zero) histogram itself   // nothing to sample: the cut of the (zero) histogram = "accept everything"
    hipLaunchKernelGGL(vc_tau_init_kernel, dim3(qt), dim3(64), 0, s, d_shist, hist_stride, k, bits, d_tau, refine ? 1u : 0u, qs);
    return hipGetLastError();
  }
  const uint32_t gy = (qt + VC_SAMPLE_QSUB - 1) / VC_SAMPLE_QSUB;
  const uint64_t npairs = std::max<uint64_t>((s_items + 1) / 2, 1);
  const uint64_t per_cu = blocks_per_cu ? blocks_per_cu : 8;
  const uint32_t gx = (uint32_t)std::min<uint64_t>((npairs + 255) / 256, (uint64_t)n_cu * per_cu);
  const size_t lds = (size_t)VC_SAMPLE_QSUB * W * 8 + (size_t)VC_SAMPLE_QSUB * hist_stride * 4 + VC_SAMPLE_QSUB * 4;
  VcSampleParams p{cols, stride, s_items, d_queries, d_shist, d_tau, qt, hist_stride, refine ? 1u : 0u, qs};
#define VC_SH_CASE(W_)                                                                                   \
  case W_:                                                                                               \
    hipLaunchKernelGGL((vc_sample_hist_kernel<W_>), dim3(gx, gy), dim3(256), lds, s, p);                  \
    break;
  switch (W) {
    VC_SH_CASE(1)
    VC_SH_CASE(2)
    VC_SH_CASE(4)
    VC_SH_CASE(8)
    default:
      return hipErrorInvalidValue;
  }
#undef VC_SH_CASE
  hipError_t r = hipGetLastError();
  if (r != hipSuccess || !cut) return r;
  hipLaunchKernelGGL(vc_tau_init_kernel, dim3(qt), dim3(64), 0, s, d_shist, hist_stride, k, bits, d_tau, refine ? 1u : 0u, qs);
  return hipGetLastError();
}

// true when a tile of qt queries runs the small-tile form of the verify kernel (which can cut the bootstrap histograms itself)
bool vc_scan_is_small(uint32_t W, uint32_t qt, const VcKnobs* knobs, uint64_t n_items) {
  const VcScanShape sh = vc_scan_pick_shape(W, qt, nullptr, knobs, n_items);
  const int ud = W <= 2 ? 4 : (W <= 4 ? 2 : 1);
  return sh.small && sh.blk == 256 && sh.dbuf == 2 && qt <= 8 && sh.unroll == ud && !VC_SCAN_DIAGNOSTICS;
}

hipError_t vc_launch_scan(const VcScanParams& p, uint32_t W, uint32_t n_cu, uint32_t want_blocks, const VcKnobs* knobs,
                          hipStream_t s, uint64_t shape_n) {
  if (p.nchunks == 0 || p.qt == 0) return hipSuccess;
  size_t lds;
  const VcScanShape sh = vc_scan_pick_shape(W, p.qt, &lds, knobs, shape_n);
  switch (W) {
    case 1: return launch_scan_w<1>(p, sh, lds, n_cu, want_blocks, s);
    case 2: return launch_scan_w<2>(p, sh, lds, n_cu, want_blocks, s);
    case 4: return launch_scan_w<4>(p, sh, lds, n_cu, want_blocks, s);
    case 8: return launch_scan_w<8>(p, sh, lds, n_cu, want_blocks, s);
  }
  return hipErrorInvalidValue;
}

hipError_t vc_launch_select_ring(const uint64_t* d_buf, uint32_t cap, const uint32_t* d_count, const uint32_t* d_tau,
                                 uint32_t qs, uint32_t nq, uint32_t k, uint64_t* d_out, uint32_t* d_out_count, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  VcRingSrc src{d_buf, d_count, cap, nullptr, 1u, d_tau, qs};
  hipLaunchKernelGGL((vc_select_kernel<VcRingSrc>), dim3(nq), dim3(VC_SEL_THREADS), 0, s, src, k, d_out, d_out_count);
  return hipGetLastError();
}

hipError_t vc_launch_select_ring_list(const uint64_t* d_buf, uint32_t cap, const uint32_t* d_count, const uint32_t* d_list,
                                      uint32_t n_list, uint32_t k, uint64_t* d_out, uint32_t* d_out_count, hipStream_t s) {
  if (n_list == 0) return hipSuccess;
  VcRingSrc src{d_buf, d_count, cap, d_list, 0u, nullptr, 1u};
  hipLaunchKernelGGL((vc_select_kernel<VcRingSrc>), dim3(n_list), dim3(VC_SEL_THREADS), 0, s, src, k, d_out, d_out_count);
  return hipGetLastError();
}

size_t vc_recover_barrier_offset_words() { return (size_t)VC_REC_MAXQ * 3 * VC_REC_BINS + (size_t)VC_REC_MAXQ * 32; }
size_t vc_recover_scratch_words() {   // idhist | rcount lines | barrier lines (3 x 32 words) | gave_up
  return (size_t)VC_REC_MAXQ * 3 * VC_REC_BINS + (size_t)VC_REC_MAXQ * 32 + 96 + 32;
}

hipError_t vc_launch_recover(const uint64_t* cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t id_base, uint32_t bits,
                             const uint64_t* d_queries, uint32_t nq, uint32_t k, uint64_t* d_ring, uint32_t cap,
                             const uint32_t* d_count, const uint32_t* d_hist, uint32_t hist_stride, uint32_t qs, uint32_t* d_scratch,
                             uint64_t* d_out, uint32_t* d_out_count, uint32_t* d_clean_tau, uint32_t* d_clean_shist,
                             uint64_t clean_copy_stride, uint32_t clean_copies, uint32_t n_cu, uint32_t spin_limit, uint32_t absent,
                             hipStream_t s) {
  if (nq == 0) return hipSuccess;
  if (nq > VC_REC_MAXQ) return hipErrorInvalidValue;
  VcRecoverParams p{};
  p.cols = cols; p.stride = stride; p.n = n; p.queries = d_queries; p.count = d_count; p.hist = d_hist; p.ring = d_ring;
  p.idhist = d_scratch;
  p.rcount = d_scratch + (size_t)VC_REC_MAXQ * 3 * VC_REC_BINS;
  p.bar = p.rcount + (size_t)VC_REC_MAXQ * 32;
  p.gave_up = p.bar + 96;
  p.out = d_out; p.out_count = d_out_count;
  p.nq = nq; p.k = k; p.cap = cap; p.hist_stride = hist_stride; p.qs = qs; p.bits = bits; p.id_base = id_base;
  p.clean_tau = d_clean_tau; p.clean_shist = d_clean_shist; p.clean_copy_stride = clean_copy_stride; p.clean_copies = clean_copies;
  p.spin_limit = spin_limit ? spin_limit : VC_REC_SPIN_LIMIT; p.absent = absent;
  // block-local position histograms (32 KiB), reused as the sort buffer of the final rows (k entries, padded to a power
  // of two); kept small so that the grids of several engines fit on the chip side by side
  uint32_t kp = 2;
  while (kp < k) kp <<= 1;
  const size_t lds = std::max((size_t)VC_REC_RQ * VC_REC_BINS * 4, (size_t)kp * 8);
#define VC_REC_CASE(W_)                                                                                              \
  case W_: {                                                                                                         \
    auto kern = vc_recover_kernel<W_>;                                                                               \
    const uint32_t grid = std::min(2 * n_cu, resident_grid(kern, 256, lds, n_cu, 0));                                \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);                                                       \
    break;                                                                                                           \
  }
  switch (W) {
    VC_REC_CASE(1)
    VC_REC_CASE(2)
    VC_REC_CASE(4)
    VC_REC_CASE(8)
    default:
      return hipErrorInvalidValue;
  }
#undef VC_REC_CASE
  return hipGetLastError();
}

// Merge of G ascending, INF-padded lists of k packed values per query (the per-shard top-k: mpi_coordinator::gather_vectors'
// consumer, search_worker.cc:179-199): the lists are staged in LDS and every entry finds its output position by itself --
// its index in its own list + the entries below it in every other list (binary searches in LDS; ties between lists break
// by list number, so duplicates keep distinct ranks) -- entries ranked below k are written, the tail is padded.  No
// sort, two barriers: ~3 us where the general select kernel (no order assumed) takes 10.
#define VC_MERGE_THREADS 256
#define VC_MERGE_MAX_ENTRIES 6144u     // 48 KiB of LDS
template <class Src>
__global__ void __launch_bounds__(VC_MERGE_THREADS) vc_merge_sorted_kernel(Src src, uint32_t k, uint64_t* __restrict__ out,
                                                                            uint32_t* __restrict__ out_count) {
  extern __shared__ uint64_t m_lists[];
  __shared__ uint32_t s_total;
  const uint32_t q = blockIdx.x, G = src.n_lists, n = G * k;
  if (threadIdx.x == 0) s_total = 0;
  for (uint32_t i = threadIdx.x; i < n; i += VC_MERGE_THREADS) m_lists[i] = src.get(q, i);
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t i = threadIdx.x; i < n; i += VC_MERGE_THREADS) {
    const uint64_t v = m_lists[i];
    if (v == VC_PACK_INF) continue;
    ++mine;
    const uint32_t g = i / k;
    uint32_t rank = i - g * k;
    for (uint32_t h = 0; h < G && rank < k; ++h) {
      if (h == g) continue;
      const uint64_t* l = m_lists + h * k;
      uint32_t lo = 0, hi = k;
      if (h < g) { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (l[mid] <= v) lo = mid + 1; else hi = mid; } }
      else       { while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (l[mid] < v) lo = mid + 1; else hi = mid; } }
      rank += lo;
    }
    if (rank < k) out[(uint64_t)q * k + rank] = v;
  }
  if (mine) atomicAdd(&s_total, mine);
  __syncthreads();
  const uint32_t total = min(s_total, k);
  for (uint32_t i = total + threadIdx.x; i < k; i += VC_MERGE_THREADS) out[(uint64_t)q * k + i] = VC_PACK_INF;
  if (out_count && threadIdx.x == 0) out_count[q] = src.overflowed(q) ? 0xFFFFFFFFu : total;
}

template <class Src>
static hipError_t launch_merge(const Src& src, uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* d_out, uint32_t* d_out_count, hipStream_t s) {
  if ((uint64_t)n_lists * k <= VC_MERGE_MAX_ENTRIES)
    hipLaunchKernelGGL((vc_merge_sorted_kernel<Src>), dim3(nq), dim3(VC_MERGE_THREADS), (size_t)n_lists * k * 8, s, src, k, d_out, d_out_count);
  else   // too many entries for LDS staging: the general select (radix select + sort)
    hipLaunchKernelGGL((vc_select_kernel<Src>), dim3(nq), dim3(VC_SEL_THREADS), 0, s, src, k, d_out, d_out_count);
  return hipGetLastError();
}

hipError_t vc_launch_select_slots(const uint64_t* d_base, uint64_t slot_words, uint32_t cnt_off_words, uint32_t n_lists, uint32_t nq,
                                  uint32_t k, uint64_t* d_out, uint32_t* d_out_count, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  VcSlotsSrc src{d_base, slot_words, cnt_off_words, n_lists, nq, k};
  return launch_merge(src, n_lists, nq, k, d_out, d_out_count, s);
}

hipError_t vc_launch_select_lists(const uint64_t* d_lists, uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* d_out,
                                  uint32_t* d_out_count, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  VcListsSrc src{d_lists, n_lists, nq, k};
  return launch_merge(src, n_lists, nq, k, d_out, d_out_count, s);
}
