// ============================================================================
// vc_mih.hip -- multi-index hashing on gfx950: index build, bucket views, radius-incremental
// k-NN (exact + approximate) and fixed-radius neighbour search.
//
// Replaces (reference, CPU + KV tier):
//   src/build_hash_tables.cc:36-64   bucket contents (rule a12)      -> vc_mih_build (sorted CSR per table)
//   src/generate_bitmap.cc:105-114   bucket-occupancy bitmap          -> mih_bitmap_kernel
//   src/bitmap.cc:22-26              ImageBitmap::get_idx             -> vc_bit_test
//   src/search_worker.cc:230-264     enumerate_entry (shell of keys)  -> combination unranking + Gosper step
//   src/search_worker.cc:246         proxy get(HashIndex)             -> vc_lookup (direct offsets / bitmap rank)
//   src/search_worker.cc:249-257     verify + pack                    -> mih_probe_kernel phase 3
//   src/search_worker.cc:179-207     dedup map + heap + stop rule     -> owner rule + vc_select + mih_commit_kernel
//   src/mpi_coordinator.cc:34-69     gather_vectors / bcast           -> nothing to gather: all tables live in one HBM
//
// Index layout per table t (substring t of every code, s = B/m bits):
//   ids[n]        local ids sorted by (key, id): a bucket is a contiguous id run in append order
//   s <= 16 :     offsets[2^s + 1]  direct-addressed bucket starts
//   s == 32 :     bitmap[2^32 bit] + blockrank[2^24] (set bits before each 256-bit block)
//                 + offsets[U + 1] indexed by rank(key) among the U non-empty buckets
//   bitmap is kept for every s (it is the reference's own prefilter structure, bit v of word v/32).
// ============================================================================

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <sched.h>

#include "vc_internal.hpp"
#include "vc_mih.hpp"

#define MIH_BLK 256
#define MIH_PPT 4                       // probes per thread
#define MIH_EPT 4u                      // bucket entries per thread and round in the verify phase
#define MIH_PCH (MIH_BLK * MIH_PPT)     // probes per block pass
// Queries of one k-NN launch (one block each in mih_query_kernel).  A launch ends when its LONGEST query does, so every launch
// carries a fixed tail: the exact top-100 kernel takes 90 us + 41.7 ns per query at 1e8 records (60 us + 56 ns at 1e9) --
// 14.6 M queries/s in launches of 4096, 18.2 M of 8192, 20.4 M of 16384 (11.9 / 14.6 / 16.5 M at 1e9; profiles/r04_sweeps.md, 9).
// A call's queries are dealt to as few launches as VC_MIH_QTILE (default MIH_QTILE_MAX) allows; the per-slot state grows with
// the largest batch seen (>= MIH_QTILE_MIN slots).
#define MIH_QTILE_MIN 4096u
#define MIH_QTILE_MAX 16384u
#define MIH_QTILE_LIMIT 65536u           // VC_MIH_QTILE is clamped to MIH_QTILE_MIN .. this
#define MIH_RADIUS_TILE 4096u            // queries per tile of the radius search (vc_radius_offsets_kernel: four per thread)
#define MIH_APPROX_FACTOR 20u           // search_worker.h:14

struct VcTableView {
  const uint32_t* offsets;
  const uint32_t* ids;
  const uint32_t* bitmap;
  const uint32_t* blockrank;  // s == 32 only
  // s == 32 only: blockdir[b] = {offsets[blockrank[b]], blockrank[b]}, b = 0 .. 2^24 -- the entry position and the rank of
  // block b's first key in ONE 8-byte record.  When a block's buckets all hold one entry (blockdir[b + 1].x - blockdir[b].x
  // == its set bits: 93 % of the blocks at 1e8 uniform codes) the position of a key's entry is blockdir[b].x + (set bits
  // below the key); otherwise the rank is already at hand and offsets[] is the only further round trip.
  const uint2* blockoff;
  // Optional copy of the codes in THIS table's bucket order (word j of the pos-th entry at bcodes[j*n + pos]), built
  // for substrings <= 16 bit: their buckets hold thousands of items (1526 at 1e8 codes, s = 16), and verifying a
  // bucket through ids[] -> cols[] is an 8-byte gather per word that moves a 64-byte sector each; from the copy it is
  // a contiguous stream.  32-bit substrings keep gathering (0.02 items per bucket).
  const uint64_t* bcodes;
  // Optional, 32-bit substrings of 64- / 128-bit codes: {id, 0, code} of the pos-th entry in ONE 16- / 32-byte record.
  // Their buckets hold one entry, and ids[pos] -> cols[id] are two (three) dependent sectors per hit where this is one.
  const uint4* bent;
  // Optional, 32-bit substrings: DIRECTORY LINES -- one 64-byte line (= one memory sector) per 128 keys that holds the
  // occupancy bits of those keys AND what a hit needs next, so that a probe's granule read and its bucket look-up touch
  // ONE sector (the second access is an L2 hit) where bitmap -> blockoff -> offsets[] were up to three dependent sectors:
  //   words 0..3   occupancy bits of keys [128 L, 128 L + 128)   (== bitmap words 4 L .. 4 L + 3)
  //   word  4      entry position of the line's first bucket     (== offsets[rank of the line's first set bit])
  //   word  5      set bits before the line (rank of its first set bit)
  //   word  6      non-empty buckets in the line | mode << 8
  //   word  7      entries in the line
  //   words 8..15  mode 1 (<= 16 buckets, < 65 536 entries): 16-bit cumulative END offsets of the buckets, rank order;
  //                mode 2 (every bucket <= 4 entries): 2-bit (length - 1) per bucket, rank order;
  //                mode 0: neither fits -- the hit falls back to offsets[rank] (one more dependent sector)
  // 2 GB per table; derived at build / load like blockoff, not part of the index file.
  const uint4* lines;
  uint32_t n_unique;
  uint32_t pad;
};
#define MIH_LINE_MODE_OFFSETS 0u
#define MIH_LINE_MODE_CUM16 1u
#define MIH_LINE_MODE_LEN2 2u

__constant__ uint32_t c_binom[33][33];  // C(n, k), n,k <= 32 (max C(32,16) = 601,080,390 fits uint32)

namespace {

// ------------------------------------------------------------------------------------------
// build kernels
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mih_bcodes_kernel(const uint64_t* __restrict__ cols, uint64_t stride, uint32_t W,
                                                         const uint32_t* __restrict__ ids, uint64_t n,
                                                         uint64_t* __restrict__ out) {
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * W; e += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t j = e / n, pos = e - j * n;
    out[e] = cols[j * stride + ids[pos]];
  }
}

__global__ void __launch_bounds__(256) mih_bent_kernel(const uint64_t* __restrict__ cols, uint64_t stride, uint32_t W,
                                                       const uint32_t* __restrict__ ids, uint64_t n, uint4* __restrict__ out) {
  // a permutation gather: 8 entries per thread in flight (ids first, then their code words), or the 1e9-entry tables of
  // the metric's size take a second each -- one dependent pair of round trips per entry and thread otherwise
  constexpr int U = 8;
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t p0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < n; p0 += step * U) {
    uint32_t id[U];
    uint64_t c0[U], c1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t pos = p0 + u * step;
      id[u] = pos < n ? __builtin_nontemporal_load(ids + pos) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c0[u] = cols[id[u]];
      c1[u] = W == 2 ? cols[stride + id[u]] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t pos = p0 + u * step;
      if (pos >= n) continue;
      if (W == 1) {
        out[pos] = make_uint4(id[u], 0u, (uint32_t)c0[u], (uint32_t)(c0[u] >> 32));
      } else {   // W == 2: 32 bytes per entry, half a sector
        out[2 * pos] = make_uint4(id[u], 0u, (uint32_t)c0[u], (uint32_t)(c0[u] >> 32));
        out[2 * pos + 1] = make_uint4((uint32_t)c1[u], (uint32_t)(c1[u] >> 32), 0u, 0u);
      }
    }
  }
}

__global__ void __launch_bounds__(256) mih_keys_kernel(const uint64_t* __restrict__ col, uint64_t n, uint32_t shift,
                                                       uint32_t mask, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    keys[i] = (uint32_t)(col[i] >> shift) & mask;
    vals[i] = (uint32_t)i;
  }
}

// heads of key runs set the occupancy bit (generate_bitmap.cc:54-58 set_idx) and, for direct tables,
// run heads/tails leave run lengths in counts[] (tail adds p+1, head subtracts p; uint32 wrap-around is exact).
__global__ void __launch_bounds__(256) mih_runs_kernel(const uint32_t* __restrict__ keys, uint64_t n,
                                                       uint32_t* __restrict__ bitmap, uint32_t* __restrict__ counts) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t key = keys[p];
    const bool head = p == 0 || keys[p - 1] != key;
    const bool tail = p + 1 == n || keys[p + 1] != key;
    if (head) atomicOr(&bitmap[key >> 5], 1u << (key & 31));
    if (counts) {
      if (head) atomicSub(&counts[key], (uint32_t)p);
      if (tail) atomicAdd(&counts[key], (uint32_t)(p + 1));
    }
  }
}

__global__ void __launch_bounds__(256) mih_blockpop_kernel(const uint32_t* __restrict__ bitmap, uint32_t nblocks,
                                                           uint32_t* __restrict__ blockpop) {
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += gridDim.x * blockDim.x) {
    const uint4 lo = reinterpret_cast<const uint4*>(bitmap)[2 * (uint64_t)b];
    const uint4 hi = reinterpret_cast<const uint4*>(bitmap)[2 * (uint64_t)b + 1];
    blockpop[b] = __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w) + __popc(hi.x) + __popc(hi.y) +
                  __popc(hi.z) + __popc(hi.w);
  }
}

__global__ void __launch_bounds__(256) mih_blockoff_kernel(const uint32_t* __restrict__ blockrank, const uint32_t* __restrict__ offsets,
                                                           uint32_t nblocks, uint32_t n_unique, uint2* __restrict__ blockoff) {
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= nblocks; b += gridDim.x * blockDim.x) {
    const uint32_t rk = b < nblocks ? blockrank[b] : n_unique;
    blockoff[b] = make_uint2(offsets[rk], rk);
  }
}

// directory lines (VcTableView::lines): one thread per 128-key line
__global__ void __launch_bounds__(256) mih_lines_kernel(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ blockrank,
                                                        const uint32_t* __restrict__ offsets, uint32_t nlines, uint4* __restrict__ lines) {
  for (uint32_t L = blockIdx.x * blockDim.x + threadIdx.x; L < nlines; L += gridDim.x * blockDim.x) {
    const uint4 occ = reinterpret_cast<const uint4*>(bitmap)[L];
    uint32_t rank = blockrank[L >> 1];
    if (L & 1u) {
      const uint4 lo = reinterpret_cast<const uint4*>(bitmap)[L - 1];
      rank += __popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w);
    }
    const uint32_t nb = __popc(occ.x) + __popc(occ.y) + __popc(occ.z) + __popc(occ.w);
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t mode = MIH_LINE_MODE_OFFSETS, pos0 = 0, total = 0;
    if (nb) {
      pos0 = offsets[rank];
      total = offsets[rank + nb] - pos0;
      if (nb <= 16 && total < 65536u) {
        mode = MIH_LINE_MODE_CUM16;
        for (uint32_t i = 0; i < nb; ++i) w[i >> 1] |= (offsets[rank + i + 1] - pos0) << ((i & 1u) * 16u);
      } else if (total <= 4u * nb) {     // (necessary for "every bucket <= 4 entries"; checked bucket by bucket below)
        mode = MIH_LINE_MODE_LEN2;
        uint32_t prev = pos0;
        for (uint32_t i = 0; i < nb; ++i) {
          const uint32_t e = offsets[rank + i + 1], len = e - prev;
          prev = e;
          if (len > 4u) mode = MIH_LINE_MODE_OFFSETS;
          w[i >> 4] |= ((len - 1u) & 3u) << ((i & 15u) * 2u);
        }
      }
    }
    uint4* dst = lines + (uint64_t)L * 4;
    dst[0] = occ;
    dst[1] = make_uint4(pos0, rank, nb | (mode << 8), total);
    dst[2] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[3] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// bucket (entry position, length) of the key whose bit is x (0..127) in a directory line held in registers; false = mode 0
__device__ __forceinline__ bool vc_line_lookup(const uint4& occ, const uint4& dir, const uint4& f0, const uint4& f1, uint32_t x,
                                               uint32_t& pos, uint32_t& len, uint32_t& rk) {
  const uint32_t wq = x >> 5, below_mask = (1u << (x & 31u)) - 1u;
  const uint32_t r = (wq > 0 ? __popc(occ.x) : 0u) + (wq > 1 ? __popc(occ.y) : 0u) + (wq > 2 ? __popc(occ.z) : 0u) +
                     __popc((wq == 0 ? occ.x : wq == 1 ? occ.y : wq == 2 ? occ.z : occ.w) & below_mask);
  rk = dir.y + r;
  const uint32_t mode = dir.z >> 8;
  auto sel8 = [&](uint32_t i) {      // word i of f0 | f1 by masks (a select chain here was turned into a scratch array by hipcc)
    auto pick = [&](uint32_t wv, uint32_t j) { return wv & (0u - (uint32_t)(i == j)); };
    return pick(f0.x, 0) | pick(f0.y, 1) | pick(f0.z, 2) | pick(f0.w, 3) | pick(f1.x, 4) | pick(f1.y, 5) | pick(f1.z, 6) | pick(f1.w, 7);
  };
  if (mode == MIH_LINE_MODE_CUM16) {
    const uint32_t end = (sel8(r >> 1) >> ((r & 1u) * 16u)) & 0xFFFFu;
    const uint32_t beg = r ? (sel8((r - 1u) >> 1) >> (((r - 1u) & 1u) * 16u)) & 0xFFFFu : 0u;
    pos = dir.x + beg;
    len = end - beg;
    return true;
  }
  if (mode == MIH_LINE_MODE_LEN2) {
    uint32_t extra = 0;                 // sum of (length - 1) of the r buckets below: the low 2 r bits of the 256-bit field array
    auto part = [&](uint32_t wv, int i) {
      const int nbits = (int)(2u * r) - 32 * i;
      const uint32_t mk = nbits >= 32 ? 0xFFFFFFFFu : (nbits <= 0 ? 0u : ((1u << nbits) - 1u));
      extra += __popc(wv & 0x55555555u & mk) + 2u * __popc(wv & 0xAAAAAAAAu & mk);
    };
    part(f0.x, 0); part(f0.y, 1); part(f0.z, 2); part(f0.w, 3); part(f1.x, 4); part(f1.y, 5); part(f1.z, 6); part(f1.w, 7);
    pos = dir.x + r + extra;
    len = ((sel8(r >> 4) >> ((r & 15u) * 2u)) & 3u) + 1u;
    return true;
  }
  return false;
}

// ImageBitmap::get_idx (bitmap.cc:22-26)
__device__ __forceinline__ bool vc_bit_test(const uint32_t* bitmap, uint32_t v) { return (bitmap[v >> 5] >> (v & 31)) & 1u; }

// number of set bits strictly below v (rank), from the 256-bit block directory
__device__ __forceinline__ uint32_t vc_rank32(const uint32_t* bitmap, const uint32_t* blockrank, uint32_t v) {
  const uint32_t blk = v >> 8, w = (v >> 5) & 7u;
  uint32_t r = blockrank[blk];
  const uint32_t* b = bitmap + ((uint64_t)blk << 3);
  for (uint32_t i = 0; i < w; ++i) r += __popc(b[i]);
  r += __popc(b[w] & ((1u << (v & 31)) - 1u));
  return r;
}

__global__ void __launch_bounds__(256) mih_ranked_offsets_kernel(const uint32_t* __restrict__ keys, uint64_t n,
                                                                 const uint32_t* __restrict__ bitmap,
                                                                 const uint32_t* __restrict__ blockrank,
                                                                 uint32_t* __restrict__ offsets, uint32_t n_unique) {
  for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t key = keys[p];
    if (p == 0 || keys[p - 1] != key) offsets[vc_rank32(bitmap, blockrank, key)] = (uint32_t)p;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n_unique] = (uint32_t)n;
}

// ------------------------------------------------------------------------------------------
// index-file validation (vc_mih_load): nothing a search kernel dereferences is taken from a file unchecked
// ------------------------------------------------------------------------------------------
// order-independent 64-bit digest of the resident code columns: sum over (item, word) of mix(word ^ mix(position))
__global__ void __launch_bounds__(256) mih_checksum_kernel(const uint64_t* __restrict__ cols, uint64_t stride, uint32_t W,
                                                           uint64_t n, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * W; e += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t j = e / n, i = e - j * n;
    acc += vc_mix64(cols[j * stride + i] ^ vc_mix64(i * 8 + j));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, VC_WAVE);
  if (vc_lane() == 0 && acc) atomicAdd(out, acc);
}

#define MIH_BAD_ID 1u          // ids[] entry outside [0, n)
#define MIH_BAD_DUP 2u         // ids[] is not a permutation
#define MIH_BAD_BUCKET 4u      // an entry does not sit in the bucket its record's key names (or that bucket is marked empty)
#define MIH_BAD_OFFSETS 8u     // offsets[] not monotone from 0 to n
#define MIH_BAD_RANK 16u       // blockrank[] is not the prefix sum of the bitmap's block popcounts / n_unique differs

// every entry: id in range, seen once, and filed under the key of ITS record in the resident columns
__global__ void __launch_bounds__(256) mih_validate_entries_kernel(VcTableView tv, const uint64_t* __restrict__ col, uint64_t n,
                                                                   uint32_t shift, uint32_t mask, uint32_t sbits,
                                                                   uint32_t* __restrict__ seen, uint32_t* __restrict__ bad) {
  uint32_t f = 0;
  for (uint64_t pos = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t id = tv.ids[pos];
    if (id >= n) { f |= MIH_BAD_ID; continue; }
    if (atomicOr(&seen[id >> 5], 1u << (id & 31)) & (1u << (id & 31))) f |= MIH_BAD_DUP;
    const uint32_t key = (uint32_t)(col[id] >> shift) & mask;
    uint32_t a, b;
    if (sbits < 32) {
      a = tv.offsets[key];
      b = tv.offsets[key + 1];
    } else {
      if (!vc_bit_test(tv.bitmap, key)) { f |= MIH_BAD_BUCKET; continue; }
      const uint32_t rk = vc_rank32(tv.bitmap, tv.blockrank, key);
      if (rk >= tv.n_unique) { f |= MIH_BAD_BUCKET; continue; }
      a = tv.offsets[rk];
      b = tv.offsets[rk + 1];
    }
    if (!(a <= pos && pos < b)) f |= MIH_BAD_BUCKET;
  }
  if (f) atomicOr(bad, f);
}

__global__ void __launch_bounds__(256) mih_validate_offsets_kernel(const uint32_t* __restrict__ offsets, uint64_t len, uint64_t n,
                                                                   uint32_t* __restrict__ bad) {
  uint32_t f = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t v = offsets[i];
    if (i == 0 && v != 0) f = MIH_BAD_OFFSETS;
    if (i + 1 == len ? (uint64_t)v != n : v > offsets[i + 1]) f = MIH_BAD_OFFSETS;
  }
  if (f) atomicOr(bad, f);
}

__global__ void __launch_bounds__(256) mih_validate_rank_kernel(const uint32_t* __restrict__ blockrank, const uint32_t* __restrict__ expect,
                                                                const uint32_t* __restrict__ blockpop, uint32_t nblocks,
                                                                uint32_t n_unique, uint32_t* __restrict__ bad) {
  uint32_t f = 0;
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += gridDim.x * blockDim.x) {
    if (blockrank[b] != expect[b]) f = MIH_BAD_RANK;
    if (b + 1 == nblocks && expect[b] + blockpop[b] != n_unique) f = MIH_BAD_RANK;
  }
  if (f) atomicOr(bad, f);
}

// ------------------------------------------------------------------------------------------
// search state
// ------------------------------------------------------------------------------------------
struct MihState {          // all arrays indexed by the query's slot in the tile
  uint64_t* thresh;        // append iff packed < thresh (k-th best so far, or overflow-recovery limit + 1)
  uint64_t* ring;          // [slot][cap]  committed top-k in front, this shell's candidates behind
  uint32_t* count;         // ring fill (may exceed cap: overflow)
  uint32_t* prev;          // committed entries at the start of the shell
  unsigned long long* seen;   // distinct items verified (knn_found_.size())
  unsigned long long* sub;    // table-0 gets issued  (n_sub_reads_)
  unsigned long long* loc;    // table-0 bitmap tests (n_local_reads_)
  uint32_t* radius;        // last shell searched
  uint64_t* topk;          // [slot][k] select output
  uint32_t* topn;          // [slot]
  unsigned long long* work;   // [slot][4] mih_query_kernel: bucket probes, non-empty buckets, bucket entries verified, -
};

struct ProbeParams {
  const uint64_t* cols;
  uint64_t stride;
  const VcTableView* tables;
  const uint64_t* queries;   // [tile][W]
  const uint32_t* list;      // slots to process
  MihState st;
  uint32_t r, nkeys, m, sbits, id_base, flags, cap, count_seen;
  uint64_t n;                // items in the index (stride of the bucket-order code copies)
  uint32_t m_probe;          // tables probed by this launch: 0 .. m_probe-1 (0 = all m; radius search narrows the outer shells)
};

// lookup of bucket (t, key): start offset and length (0 = PROXY_NOT_FOUND)
__device__ __forceinline__ void vc_lookup(const VcTableView& tv, uint32_t sbits, uint32_t key, uint32_t& off,
                                          uint32_t& len, bool& bit) {
  if (sbits < 32) {
    const uint32_t a = tv.offsets[key], b = tv.offsets[key + 1];
    off = a;
    len = b - a;
    bit = len != 0;
  } else {
    bit = vc_bit_test(tv.bitmap, key);
    off = 0;
    len = 0;
    if (bit) {
      const uint32_t rk = vc_rank32(tv.bitmap, tv.blockrank, key);
      off = tv.offsets[rk];
      len = tv.offsets[rk + 1] - off;
    }
  }
}

// j-th r-subset of {0..s-1} in colexicographic order == numeric order of the masks (combinadic unranking)
__device__ __forceinline__ uint32_t vc_unrank(uint32_t j, uint32_t r, uint32_t s) {
  uint32_t mask = 0;
  uint32_t c = s;
  for (uint32_t i = r; i >= 1; --i) {
    do { --c; } while (c_binom[c][i] > j);   // largest c with C(c,i) <= j
    j -= c_binom[c][i];
    mask |= 1u << c;
  }
  return mask;
}

// next mask with the same popcount (Gosper), 64-bit so s = 32 cannot overflow
__device__ __forceinline__ uint32_t vc_next_comb(uint32_t x) {
  const uint64_t v = x;
  const uint64_t c = v & (0 - v);
  const uint64_t rr = v + c;
  return (uint32_t)((((rr ^ v) >> 2) >> (__ffsll((long long)v) - 1)) | rr);
}

template <int W>
__global__ void __launch_bounds__(MIH_BLK) mih_probe_kernel(const ProbeParams p) {
  __shared__ uint32_t s_off[MIH_PCH];
  __shared__ uint32_t s_pref[MIH_PCH + 1];   // lens, then exclusive prefix
  __shared__ uint32_t s_wsum[MIH_BLK / VC_WAVE];
  __shared__ uint32_t s_n, s_leaves, s_hits;

  const uint32_t slot = p.list[blockIdx.z];
  const uint32_t t = blockIdx.y;
  const VcTableView tv = p.tables[t];
  const uint32_t s = p.sbits;
  const uint32_t smask = s == 32 ? 0xFFFFFFFFu : ((1u << s) - 1u);
  const uint32_t lane = vc_lane();
  const uint32_t wave = threadIdx.x / VC_WAVE;

  uint64_t qw[W];
#pragma unroll
  for (int j = 0; j < W; ++j) qw[j] = p.queries[(uint64_t)slot * W + j];
  const uint32_t bitpos = t * s;
  uint32_t qkey = 0;
#pragma unroll
  for (int j = 0; j < W; ++j)
    if ((uint32_t)j == (bitpos >> 6)) qkey = (uint32_t)(qw[j] >> (bitpos & 63)) & smask;

  if (threadIdx.x == 0) { s_n = 0; s_leaves = 0; s_hits = 0; }
  __syncthreads();

  // ---- phase 1: enumerate this block's slice of the shell, look the buckets up, compact non-empty ones
  const uint32_t j0 = blockIdx.x * MIH_PCH + threadIdx.x * MIH_PPT;
  uint32_t offv[MIH_PPT], lenv[MIH_PPT];
  uint32_t nvalid = 0, leaves = 0, hits = 0;
  uint32_t mask = 0;
  if (j0 < p.nkeys) mask = vc_unrank(j0, p.r, s);
#pragma unroll
  for (int i = 0; i < MIH_PPT; ++i) {
    offv[i] = 0;
    lenv[i] = 0;
    if (j0 + i < p.nkeys) {
      if (i) mask = vc_next_comb(mask);
      ++leaves;
      // binaryToInt's sign-extended keys: a probe that flips the substring's top bit keeps the query's
      // high bits and can match nothing (Pilaf/image_tools.h:13); reproduced only on request
      const bool dead = (p.flags & VC_FLAG_REF_SIGNEXT_KEYS) && s < 32 && ((mask >> (s - 1)) & 1u);
      bool bit = false;
      if (!dead) vc_lookup(tv, s, qkey ^ mask, offv[i], lenv[i], bit);
      hits += (p.flags & VC_FLAG_USE_BITMAP) ? (bit ? 1u : 0u) : 1u;
      nvalid += lenv[i] != 0;
    }
  }
  {
    uint32_t wtotal;
    uint32_t pos = vc_wave_excl_scan(nvalid, wtotal);
    uint32_t wbase = 0;
    if (lane == 0 && wtotal) wbase = atomicAdd(&s_n, wtotal);
    wbase = __shfl(wbase, 0, VC_WAVE);
    pos += wbase;
#pragma unroll
    for (int i = 0; i < MIH_PPT; ++i)
      if (lenv[i]) {
        s_off[pos] = offv[i];
        s_pref[pos] = lenv[i];
        ++pos;
      }
    if (t == 0 && p.count_seen) {
      uint32_t tl, th;
      (void)vc_wave_excl_scan(leaves, tl);
      (void)vc_wave_excl_scan(hits, th);
      if (lane == 0) {
        atomicAdd(&s_leaves, tl);
        atomicAdd(&s_hits, th);
      }
    }
  }
  __syncthreads();
  const uint32_t nb = s_n;
  if (t == 0 && p.count_seen && threadIdx.x == 0) {
    atomicAdd(&p.st.sub[slot], (unsigned long long)s_hits);
    if (p.flags & VC_FLAG_USE_BITMAP) atomicAdd(&p.st.loc[slot], (unsigned long long)s_leaves);
  }
  if (nb == 0) return;

  // ---- phase 2: exclusive prefix sum of bucket lengths (block scan, MIH_PPT consecutive entries per thread)
  uint32_t lsum = 0, lv[MIH_PPT];
#pragma unroll
  for (int i = 0; i < MIH_PPT; ++i) {
    const uint32_t idx = threadIdx.x * MIH_PPT + i;
    lv[i] = idx < nb ? s_pref[idx] : 0;
    lsum += lv[i];
  }
  uint32_t wtot;
  uint32_t excl = vc_wave_excl_scan(lsum, wtot);
  if (lane == 0) s_wsum[wave] = wtot;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < MIH_BLK / VC_WAVE; ++w) {
    if ((uint32_t)w < wave) wbase += s_wsum[w];
    total += s_wsum[w];
  }
  excl += wbase;
#pragma unroll
  for (int i = 0; i < MIH_PPT; ++i) {
    const uint32_t idx = threadIdx.x * MIH_PPT + i;
    if (idx < nb) s_pref[idx] = excl;
    excl += lv[i];
  }
  if (threadIdx.x == 0) s_pref[nb] = total;
  __syncthreads();

  // ---- phase 3: balanced expansion of the bucket entries, gather + verify + owner rule + compaction
  const uint64_t thresh = p.st.thresh[slot];
  uint64_t* ring = p.st.ring + (uint64_t)slot * p.cap;
  // MIH_EPT entries per thread and round: their bucket searches, id loads and code loads are all issued before the
  // first one is consumed (a round is a chain LDS search -> global loads -> verify; with one entry per thread the wave
  // spends most of a big-bucket shell waiting on it)
  constexpr uint32_t ROUND = MIH_BLK * MIH_EPT;
  const uint32_t nrounds = (total + ROUND - 1) / ROUND;
  uint32_t seen_acc = 0;   // wave-uniform
  for (uint32_t it = 0; it < nrounds; ++it) {
    uint32_t local[MIH_EPT];
    uint64_t x[MIH_EPT][W];
    bool live[MIH_EPT];
#pragma unroll
    for (uint32_t g = 0; g < MIH_EPT; ++g) {
      const uint32_t e = it * ROUND + g * MIH_BLK + threadIdx.x;
      live[g] = e < total;
      const uint32_t ec = live[g] ? e : 0;   // clamp: entry 0 exists whenever total > 0
      uint32_t lo = 0, hi = nb;               // largest b with s_pref[b] <= e
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pref[mid] <= ec) lo = mid; else hi = mid;
      }
      const uint32_t pos = s_off[lo] + (ec - s_pref[lo]);
      local[g] = tv.ids[pos];
      if (tv.bcodes) {
#pragma unroll
        for (int j = 0; j < W; ++j) x[g][j] = tv.bcodes[(uint64_t)j * p.n + pos];
      } else {
#pragma unroll
        for (int j = 0; j < W; ++j) x[g][j] = p.cols[(uint64_t)j * p.stride + local[g]];
      }
    }
#pragma unroll
    for (uint32_t g = 0; g < MIH_EPT; ++g) {
      bool emit = live[g];
      uint64_t packed = 0;
      if (live[g]) {
        // per-substring distances come free with the full distance (compute_hamming_dist, image_tools.h:21-33)
        uint32_t dist = 0;
        for (uint32_t tt = 0; tt < p.m; ++tt) {
          const uint32_t bp = tt * s;
          uint32_t field = 0;
#pragma unroll
          for (int j = 0; j < W; ++j)
            if ((uint32_t)j == (bp >> 6)) field = (uint32_t)((x[g][j] ^ qw[j]) >> (bp & 63)) & smask;
          const uint32_t d = __popc(field);
          dist += d;
          // owner rule: the item is reported by the first table holding its minimum substring distance, in the
          // shell equal to that distance -- exactly once over the whole radius loop (replaces knn_found_).
          bool reach = true;  // could table tt have fetched this item at shell d?  (sign-extended keys: only if top bits agree)
          if ((p.flags & VC_FLAG_REF_SIGNEXT_KEYS) && s < 32) reach = ((field >> (s - 1)) & 1u) == 0;
          if (tt != t && reach && (d < p.r || (d == p.r && tt < t))) emit = false;
        }
        packed = vc_pack(dist, p.id_base + local[g]);
      }
      const uint64_t emask = __ballot(emit);
      if (emask == 0) continue;
      seen_acc += (uint32_t)__popcll(emask);   // one atomic per wave after the loop: per-step atomics of every wave of
                                               // a query on one counter line serialise in its memory channel
      const bool keep = emit && packed < thresh;
      const uint64_t kmask = __ballot(keep);
      if (kmask == 0) continue;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&p.st.count[slot], (uint32_t)__popcll(kmask));
      base = __shfl(base, 0, VC_WAVE);
      if (keep) {
        const uint32_t pos = base + (uint32_t)__popcll(kmask & ((1ull << lane) - 1ull));
        if (pos < p.cap) ring[pos] = packed;
      }
    }
  }
  if (p.count_seen && lane == 0 && seen_acc) atomicAdd(&p.st.seen[slot], (unsigned long long)seen_acc);
}

// after vc_select: commit the shell for every listed slot, apply the stop rule, build the next lists
struct CommitParams {
  MihState st;
  const uint32_t* list;
  uint32_t* next_list;
  uint32_t* redo_list;
  uint32_t* ctr;        // [0] = n_next, [1] = n_redo
  uint32_t k, cap, r, sbits, stop_mult, approximate, last_shell;
};

__global__ void __launch_bounds__(VC_WAVE) mih_commit_kernel(const CommitParams c) {
  const uint32_t slot = c.list[blockIdx.x];
  const uint32_t raw = c.st.count[slot];
  const uint32_t kk = c.st.topn[slot];
  const uint64_t* top = c.st.topk + (uint64_t)slot * c.k;
  uint64_t* ring = c.st.ring + (uint64_t)slot * c.cap;
  if (raw > c.cap) {
    // ring overflowed: everything that can still matter is <= the k-th best of what did fit.  Rewind to the
    // committed prefix and re-run this shell for this query with that limit (strictly tighter every round).
    if (threadIdx.x == 0) {
      c.st.thresh[slot] = top[c.k - 1] + 1;   // cap >= 4k, so kk == k here
      c.st.count[slot] = c.st.prev[slot];
      c.redo_list[atomicAdd(&c.ctr[1], 1u)] = slot;
    }
    return;
  }
  for (uint32_t i = threadIdx.x; i < kk; i += VC_WAVE) ring[i] = top[i];
  if (threadIdx.x == 0) {
    c.st.count[slot] = kk;
    c.st.prev[slot] = kk;
    const uint64_t kth = kk == c.k ? top[c.k - 1] : VC_PACK_INF;
    c.st.thresh[slot] = kth;
    bool stop;
    if (c.approximate)   // search_worker.cc:136-137: heap of k*20 distinct candidates is full
      stop = c.st.seen[slot] >= (unsigned long long)c.k * MIH_APPROX_FACTOR;
    else                 // search_worker.cc:201-205: size == k && top.dist <= radius * 4 (radius already incremented)
      stop = kk == c.k && (uint32_t)(kth >> 32) <= (c.r + 1) * c.stop_mult;
    if (stop || c.last_shell) {
      c.st.radius[slot] = c.r;   // find() returns radius - 1 = last shell searched
    } else {
      c.next_list[atomicAdd(&c.ctr[0], 1u)] = slot;
    }
  }
}

// queries handed over by mih_query_kernel: committed top-k (st.topk) -> front of the slot's candidate ring
__global__ void __launch_bounds__(256) mih_seed_ring_kernel(MihState st, const uint32_t* __restrict__ list, uint32_t k, uint32_t cap) {
  const uint32_t slot = list[blockIdx.x];
  const uint32_t kk = st.count[slot];
  for (uint32_t i = threadIdx.x; i < kk; i += blockDim.x) st.ring[(uint64_t)slot * cap + i] = st.topk[(uint64_t)slot * k + i];
}

__global__ void __launch_bounds__(256) mih_init_kernel(MihState st, uint32_t nq, uint32_t* list, uint64_t thresh0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  st.thresh[i] = thresh0;
  st.count[i] = 0;
  st.prev[i] = 0;
  st.seen[i] = 0;
  st.sub[i] = 0;
  st.loc[i] = 0;
  st.radius[i] = 0;
  st.topn[i] = 0;
  list[i] = i;
}

__global__ void __launch_bounds__(256) mih_export_kernel(MihState st, const uint32_t* __restrict__ list, uint32_t k, uint32_t cap,
                                                         uint64_t* __restrict__ out, uint32_t* __restrict__ cnt) {
  const uint32_t q = list ? list[blockIdx.x] : blockIdx.x;
  const uint32_t n = min(st.count[q], k);
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) out[(uint64_t)q * k + i] = i < n ? st.ring[(uint64_t)q * cap + i] : VC_PACK_INF;
  if (threadIdx.x == 0) cnt[q] = n;
}


__global__ void __launch_bounds__(256) vc_fill_u32_kernel(uint32_t* p, uint32_t n, uint32_t v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// =============================================================================================================
// mih_order_kernel -- longest-first launch order for mih_query_kernel (k-NN modes).
// One block per query lasts as long as its query: on the bench's near-duplicate workload 6 % of the queries walk on to
// shell 3 and live three times as long as the rest, and the ones that happen to start late are the launch's tail (r04: after
// the last ordinary block ends only those ~260 blocks run, for 13 % of the kernel's time).  How long a query will live is not
// known beforehand, but how many database items share one of its substrings EXACTLY -- the lengths of its m shell-0 buckets
// -- predicts it well (the fifth of a batch with the smallest sums holds 81 % of its shell-3 queries, oracle run on 2 M
// clustered codes): few exact companions = a sparse neighbourhood = a long radius loop.  One thread per query looks its m
// buckets up and sums their lengths; every block of the pre-pass orders its 64 queries by that sum and deals them into order[]
// round-robin with the other blocks -- sorted up to sampling noise, no exchange between blocks (a plain two-class partition,
// fewer than k exact companions first and no order inside the classes, keeps only a quarter of the gain: 0.347 vs 0.316 ms per
// launch at 1e9); the query kernel's blocks read order[] instead of their own index.  Results do not depend on the order.
// =============================================================================================================
#define MO_BLK 256u
__global__ void __launch_bounds__(MO_BLK) mih_order_kernel(const VcTableView* __restrict__ tables, const uint64_t* __restrict__ queries,
                                                           uint32_t W, uint32_t m, uint32_t sbits, uint32_t nq, uint32_t* __restrict__ order) {
  __shared__ uint32_t s_sc[MO_BLK];
  // one thread per (query, table): nq * m / 256 blocks, so that the look-ups of a 4096-query batch spread over 64 CUs
  const uint32_t per_block = MO_BLK / m;                       // queries of a block
  const uint32_t ql = threadIdx.x / m, t = threadIdx.x - ql * m;
  const uint32_t q = blockIdx.x * per_block + ql;
  const uint32_t smask = sbits == 32 ? 0xFFFFFFFFu : ((1u << sbits) - 1u);
  s_sc[threadIdx.x] = 0;
  __syncthreads();
  if (ql < per_block && q < nq) {
    const uint32_t bp = t * sbits;
    const uint32_t key = (uint32_t)(queries[(uint64_t)q * W + (bp >> 6)] >> (bp & 63)) & smask;
    const VcTableView& tv = tables[t];
    uint32_t len = 0;
    if (sbits == 32 && tv.lines) {          // one sector: occupancy, first entry position and the buckets' extents
      const uint4* ln = tv.lines + (uint64_t)(key >> 7) * 4;
      const uint4 occ = ln[0];
      const uint32_t x = key & 127u;
      const uint32_t wv = x < 32 ? occ.x : (x < 64 ? occ.y : (x < 96 ? occ.z : occ.w));
      if ((wv >> (x & 31u)) & 1u) {
        const uint4 dir = ln[1], f0 = ln[2], f1 = ln[3];
        uint32_t pos, rk;
        if (!vc_line_lookup(occ, dir, f0, f1, x, pos, len, rk)) len = tv.offsets[rk + 1] - tv.offsets[rk];
      }
    } else {
      uint32_t off;
      bool bit;
      vc_lookup(tv, sbits, key, off, len, bit);
    }
    if (len) atomicAdd(&s_sc[ql], len);
  }
  __syncthreads();
  // The block orders ITS queries by ascending score and deals them into the launch order round-robin with the other blocks:
  // local rank r of block b -> position r * blocks + b.  Every block holds a random sample of the batch, so the dealt sequence
  // is sorted up to the sampling noise -- with no word exchanged between blocks (a device-scope fence and a ticket per block,
  // for the block that arrives last to sort the whole batch, made this pre-pass 20 us: a release writes an XCD's L2 back).
  const uint32_t nb = gridDim.x, last_cnt = nq - (nb - 1u) * per_block;   // queries of the last block (1 .. per_block)
  const uint32_t mine_n = blockIdx.x + 1u == nb ? last_cnt : per_block;
  if (threadIdx.x < mine_n) {
    const uint32_t sc = s_sc[threadIdx.x];
    uint32_t r = 0;
    for (uint32_t j = 0; j < mine_n; ++j) {
      const uint32_t o = s_sc[j];
      r += o < sc || (o == sc && j < threadIdx.x);
    }
    // ranks >= last_cnt do not exist in the last block: the rows of the deal from there on are one short
    const uint32_t pos = r * nb - (r > last_cnt ? r - last_cnt : 0u) + blockIdx.x;
    order[pos] = blockIdx.x * per_block + threadIdx.x;
  }
}

// =============================================================================================================
// mih_query_kernel -- ONE 256-thread block runs a query's whole radius loop (search_worker.cc:159-218 / 93-157,
// shells 0..r_last) or a whole fixed-radius neighbour search (search_R_neighbors shells of every table up to its pigeonhole radius, :222-227) in ONE
// launch: probe -> bucket lookup -> verify -> top-k merge -> stop rule, shell after shell, with no host round trip
// and no kernel boundary between shells.  Queries are independent, so the grid is simply one block per query; what
// the round-1 loop paid per shell (memset + probe + select + commit + an 8-byte read-back and ~20 us of host
// turnaround, with one 256-thread block per (query, table) even for the 1-key shell) is gone.
//
// 32-bit substrings -- the reference's native shape -- are probed by GRANULE, not by key: the keys of a shell are
// qkey ^ mask with popcount(mask) = r; split mask = (hi: 25 bits, lo: 7 bits).  All keys that share `hi` lie in ONE
// aligned 128-bit granule of the occupancy bitmap (bit v of word v/32, bitmap.cc:22-26), so a lane loads that
// granule once (one 16-byte load) and ANDs it with a per-query mask E_j = {x : popcount(x ^ qlo) = j}, j = r - |hi|
// (radius search: the ball B_j = E_0 | .. | E_j), precomputed in LDS.  Shells 0..4 of a 32-bit substring are 41 449
// keys but only 15 276 granules; a set bit of (granule & mask) is a non-empty bucket of the shell.
// <= 16-bit substrings have direct offsets (two 4-byte loads per key), enumerated by combination unranking + Gosper.
//
// Per block: hits (non-empty buckets) are compacted into an LDS list; when it holds >= MQ_HFLUSH entries (and at the
// end of a shell) it is DRAINED: rank -> offsets for the 32-bit tables, block prefix sum of the bucket lengths,
// balanced expansion of the entries over the threads (binary search in the LDS prefix array), gather of id + code,
// full distance + every substring distance, OWNER RULE (vc_mih.hip mih_probe_kernel), survivors below the running
// threshold appended to an UNSORTED candidate buffer in LDS and counted in a distance histogram.  After a shell one
// wave cuts the histogram (smallest d whose cumulative count reaches k): that is the stop rule's k-th distance and the
// next threshold; the buffer is compacted when it fills and ordered once, when the query ends (k-NN modes; radius
// search sorts its results once at the end).  The first pass of 32-bit substrings covers shells 0 .. group-1 together:
// candidates carry their shell class in the two top bits of the packed value, histogram / seen / bitmap-hit counters
// exist per class, and the stop rule is then evaluated shell by shell -- same results and statistics, one scan / drain
// round instead of one per shell.  A pass gives every thread MQ_GPT_KNN granules (6: shell 2's 1 304 granules of four
// tables are one pass).  A query that is not finished after shell r_last (the shells beyond cost > 10^5 probes) writes
// its state to the slot arrays and joins the `heavy` list, which the multi-block kernels above continue from shell
// r_last + 1.  Round 4: with directory lines (VcTableView::lines, template parameter LINES) a granule is the first 16 bytes of
// its key range's 64-byte line and a hit's bucket look-up reads the rest of the SAME sector; k-NN drains map entries to buckets
// through a bitmap of bucket starts instead of binary searches; the owner rule is unrolled for 32-bit substrings; wave scans run
// on DPP (vc_common.hpp).  The kernel sits at the 128-VGPR limit of its 4 waves per SIMD (tests/test_build_cpu.py guards it):
// result / state pointers are read from the kernel-argument segment where they are used (cold()), per-query counters
// live in LDS, and the rare buffer paths (mq_compact, mq_select_exact) are functions of their own.
// =============================================================================================================
#ifndef MQ_BLK
#define MQ_BLK 256u
#endif
#define MQ_G 4u                         // keys per thread per pass (<= 16-bit substrings)
#define MQ_PASS (MQ_BLK * MQ_G)
// 32-bit substrings: a key = (hi: 32 - LO bits | lo: LO bits), a granule = the 2^LO bitmap bits that share `hi`; LO is a
// template parameter of the kernel.  Radius search (configs[1]: ~31 K granules per query) runs LO = 9: a granule is one
// whole 64-byte sector (four 16-byte loads of a lane, the last three hit the line the first one fetched) and shells 0..4
// of a table are 10 903 sector reads instead of the 15 276 separate sectors of 128-bit granules (+7 % queries/s; the
// rest of a query's sectors are its ~1 900 bucket look-ups).  k-NN (a few hundred granules per query, a latency chain)
// keeps LO = 7: one 16-byte load and four mask words per granule (LO = 9 there: -15 %).
#ifndef MQ_LO_RADIUS
#define MQ_LO_RADIUS 9u
#endif
#ifndef MQ_LO_KNN
#define MQ_LO_KNN 7u
#endif
#define MQ_LO_MAX (MQ_LO_RADIUS > MQ_LO_KNN ? MQ_LO_RADIUS : MQ_LO_KNN)
#ifndef MQ_HMAX
#define MQ_HMAX 1024u                   // LDS hit list (non-empty buckets awaiting a drain)
#endif
#define MQ_HFLUSH (MQ_HMAX / 2)
#ifndef MQ_MINW
#define MQ_MINW 4
#endif
#ifndef MQ_EPT
#define MQ_EPT 4u                        // bucket entries per thread and round of the query kernel's verify phase
#endif
#define MQ_ROUND (MQ_BLK * MQ_EPT)      // bucket entries verified per round
#define MQ_BW 17u                       // row width of the LDS binomial table: C(c, i), c <= 32, i <= 16
#define MQ_MODE_EXACT 0u
#define MQ_MODE_APPROX 1u
#define MQ_MODE_RADIUS 2u

// k-NN candidates of a grouped first pass carry their shell (relative to the pass's first shell) in the two top bits:
// dist << 32 | id uses 42 bits at most.  VC_PACK_INF reads as class 3, which no entry has, so padding sorts last.
#define MQ_TAG_SHIFT 62
#define MQ_TAG_MASK (3ull << MQ_TAG_SHIFT)
#define MQ_MAX_GROUP 3u
#ifndef MQ_LAZY_BINOM
#define MQ_LAZY_BINOM 1
#endif
#ifndef MQ_BMW
#define MQ_BMW 256u                      // words of the drain's entry -> bucket bitmap: drains of up to 8 192 entries use it (one per thread)
#endif
#ifndef MQ_FAST_OWNER
#define MQ_FAST_OWNER 1
#endif
// (r04: a persistent form -- one residency wave of blocks drawing queries from a ticket counter, no workgroup launch per query --
// was built: + 3 % against one block per query inside the same binary, but the loop around the body keeps ~20 more registers
// alive at the 128-VGPR limit: - 5 % net, and every slot busy only made the blocks live longer.  Removed; profiles/r04_sweeps.md.)
static __host__ __device__ inline uint32_t mq_hist_bins(uint32_t W) { return (W * 64u + 1u + 7u) & ~7u; }

struct QueryKernelParams {
  const uint64_t* cols;
  uint64_t stride, n;
  const VcTableView* tables;
  const uint64_t* queries;     // [nq][W]
  MihState st;
  uint32_t m, sbits, id_base, flags, cap, k;
  uint32_t mode;               // MQ_MODE_*
  uint32_t radius;             // MQ_MODE_RADIUS: full-distance radius
  uint32_t stop_mult;
  uint32_t r_last;             // last shell run in here (radius mode: the substring radius of the first n_big tables)
  uint32_t n_big;              // radius mode: tables 0 .. n_big-1 search shells 0 .. r_last,
  uint32_t small_shells;       //              the others shells 0 .. small_shells-1 (0 = not at all)
  uint32_t buf_entries;        // LDS top-k + candidate buffer (power of two, >= k + MQ_ROUND)
  uint32_t* heavy_list;
  uint32_t* heavy_ctr;
  uint64_t* out;               // k-NN: [nq][k] rows
  uint32_t* out_cnt;
  unsigned long long* phase_dbg;  // dev (VC_MIH_PHASES): [8] phase times of mih_query_kernel, summed over the launch; [8..13] block lifetimes by stop shell (0..4, handed over), [16..21] their counts, [24] start of the first block, [25..30] latest end by stop shell, [32 + 8 c ..] the phase times of class c
  const uint32_t* order;       // k-NN: block b serves query order[b] (longest-first, mih_order_kernel); null = query b
  uint32_t use_lines;          // k-NN, 32-bit substrings: probe the directory lines (VcTableView::lines) instead of bitmap + blockoff + offsets
  uint32_t group;              // k-NN, 32-bit substrings: shells 0 .. group-1 share the first pass (1 = one shell per pass)
  uint32_t* radius_hist;       // k-NN: [4] queries of the launch by the shell they stopped in (0, 1, 2, later / handed over)
};

__device__ __forceinline__ uint32_t mq_unrank(const uint32_t* sb, uint32_t j, uint32_t r, uint32_t s) {
  // the shells that matter (r <= 2) in closed form: the table walk below is a chain of ~25 dependent LDS reads per flip
  if (r == 0) return 0u;
  if (r == 1) return 1u << j;
  if (r == 2) {   // j = C(c2, 2) + c1, c1 < c2: c2 = floor((1 + sqrt(1 + 8 j)) / 2), corrected for rounding
    uint32_t c2 = (uint32_t)((1.0f + __fsqrt_rn(1.0f + 8.0f * (float)j)) * 0.5f);
    while (c2 * (c2 - 1) / 2 > j) --c2;
    while ((c2 + 1) * c2 / 2 <= j) ++c2;
    return (1u << c2) | (1u << (j - c2 * (c2 - 1) / 2));
  }
  uint32_t mask = 0, c = s;
  for (uint32_t i = r; i >= 1; --i) {
    do { --c; } while (sb[c * MQ_BW + i] > j);   // largest c with C(c,i) <= j
    j -= sb[c * MQ_BW + i];
    mask |= 1u << c;
  }
  return mask;
}

// smallest d with sum_{d' <= d} h[d'] >= k, 0xFFFFFFFF if the histogram holds fewer than k (one wave, LDS histogram)
__device__ __forceinline__ uint32_t mw_hist_cut(const uint32_t* h, uint32_t nbins, uint32_t k) {
  const uint32_t lane = vc_lane();
  const uint32_t bpl = (nbins + VC_WAVE - 1) / VC_WAVE;
  uint32_t mine = 0;
  for (uint32_t i = 0; i < bpl; ++i) {
    const uint32_t bin = lane * bpl + i;
    if (bin < nbins) mine += h[bin];
  }
  uint32_t total;
  uint32_t run = vc_wave_excl_scan(mine, total);
  uint32_t cand = 0xFFFFFFFFu;
  for (uint32_t i = 0; i < bpl; ++i) {
    const uint32_t bin = lane * bpl + i;
    if (bin < nbins) {
      run += h[bin];
      if (run >= k && cand == 0xFFFFFFFFu) cand = bin;
    }
  }
  return vc_wave_min(cand);
}

// ---- k-NN candidate buffer of mih_query_kernel: s_buf[0 .. *s_ncand) unsorted, every entry tagged with its shell class
// (MQ_TAG_MASK; class 0 = shells evaluated so far + the first shell of the pass).  Rare paths, deliberately NOT inlined
// (all threads of the block call them together).
// Drop what the threshold has overtaken and the classes above max_cls, in place, 256 entries at a time.
__device__ __noinline__ void mq_compact(uint64_t* s_buf, uint32_t* s_ncand, const uint64_t* s_thresh, uint32_t* s_wsum,
                                        uint32_t buf_entries, uint32_t max_cls, bool clear_tags) {
  const uint32_t tid = threadIdx.x, lane = vc_lane(), wave = tid / VC_WAVE;
  __syncthreads();
  const uint32_t fill = min(*s_ncand, buf_entries);
  const uint64_t thr = *s_thresh;
  uint32_t nf = 0;
  for (uint32_t base = 0; base < fill; base += MQ_BLK) {
    const uint32_t i = base + tid;
    const uint64_t v = i < fill ? s_buf[i] : VC_PACK_INF;
    const bool keep = i < fill && (v & ~MQ_TAG_MASK) < thr && (uint32_t)(v >> MQ_TAG_SHIFT) <= max_cls;
    const uint64_t km = __ballot(keep);
    if (lane == 0) s_wsum[wave] = (uint32_t)__popcll(km);
    __syncthreads();                      // every thread has read its entry; the wave counts are visible
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < MQ_BLK / VC_WAVE; ++w) {
      if (w < wave) before += s_wsum[w];
      total += s_wsum[w];
    }
    if (keep) s_buf[nf + before + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = clear_tags ? (v & ~MQ_TAG_MASK) : v;
    nf += total;
    __syncthreads();
  }
  if (tid == 0) *s_ncand = nf;
  __syncthreads();
}

// Exact selection when ties overfill the buffer: sort (class-major: the tag sits in the top bits) and keep the k best of
// every class -- a superset of the top-k of each prefix of shells the stop rule may still be evaluated for.
__device__ __noinline__ void mq_select_exact(uint64_t* s_buf, uint32_t* s_ncand, uint64_t* s_thresh, uint32_t* s_tmp /*[4]*/,
                                             uint32_t buf_entries, uint32_t k) {
  const uint32_t tid = threadIdx.x;
  __syncthreads();
  const uint32_t fill = min(*s_ncand, buf_entries);
  uint32_t P = 2;
  while (P < fill) P <<= 1;
  for (uint32_t i = fill + tid; i < P; i += MQ_BLK) s_buf[i] = VC_PACK_INF;
  vc_bitonic_lds(s_buf, P, MQ_BLK);
  if (tid < 4) s_tmp[tid] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < fill; i += MQ_BLK) atomicAdd(&s_tmp[(uint32_t)(s_buf[i] >> MQ_TAG_SHIFT)], 1u);
  __syncthreads();
  const uint32_t n0 = s_tmp[0], n1 = s_tmp[1], n2 = s_tmp[2];
  const uint32_t k0 = min(n0, k), k1 = min(n1, k), k2 = min(n2, k);
  __syncthreads();
  // the kept entries of classes 1 and 2 move up behind those of the class before (destination <= source: in order, in chunks)
  for (uint32_t c = 1; c <= 2; ++c) {
    const uint32_t src = c == 1 ? n0 : n0 + n1, dst = c == 1 ? k0 : k0 + k1, cnt = c == 1 ? k1 : k2;
    for (uint32_t base = 0; base < cnt; base += MQ_BLK) {
      const uint32_t i = base + tid;
      const uint64_t v = i < cnt ? s_buf[src + i] : 0;
      __syncthreads();
      if (i < cnt) s_buf[dst + i] = v;
      __syncthreads();
    }
  }
  if (tid == 0) {
    *s_ncand = k0 + k1 + k2;
    if (n0 >= k && s_buf[k - 1] + 1 < *s_thresh) *s_thresh = s_buf[k - 1] + 1;   // exact bound for everything still to come
  }
  __syncthreads();
}

template <int W, uint32_t MQ_LO, bool LINES>
__global__ void __launch_bounds__(MQ_BLK, (W <= 2 ? MQ_MINW : 0)) mih_query_kernel(const QueryKernelParams p) {
  constexpr uint32_t MQ_HI = 32u - MQ_LO;
  constexpr uint32_t MQ_GW = (1u << MQ_LO) / 32u;                    // 32-bit words per granule
  constexpr uint32_t MQ_NJ = MQ_LO + 1u;                              // masks per table: j = 0 .. LO flips inside the low part
#ifndef MQ_GPT_KNN
#define MQ_GPT_KNN 6u
#endif
  constexpr uint32_t MQ_G32 = MQ_GW >= 16u ? 1u : (MQ_GW == 4u ? MQ_GPT_KNN : 16u / MQ_GW);   // granules per thread per pass (64 bytes in flight per lane)
  constexpr uint32_t MQ_PASS32 = MQ_BLK * MQ_G32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* s_buf = (uint64_t*)smem;                         // [buf_entries] top-k (sorted) | fresh candidates
  uint32_t* s_key = (uint32_t*)(s_buf + p.buf_entries);      // [MQ_HMAX] bucket key, then bucket offset
  uint32_t* s_pref = s_key + MQ_HMAX;                        // [MQ_HMAX + 1] bucket length, then exclusive prefix
  uint32_t* s_meta = s_pref + MQ_HMAX + 1;                   // [MQ_HMAX] table | substring distance << 8
  uint32_t* s_binom = s_meta + MQ_HMAX;                      // [33][MQ_BW]
  uint32_t* s_mask = s_binom + 33 * MQ_BW;                   // [m][MQ_NJ][MQ_GW] (32-bit substrings only)
  const uint32_t HB = mq_hist_bins(W);
  uint32_t* s_hist = s_mask + (p.sbits == 32 ? p.m * MQ_NJ * MQ_GW : 0u);   // [MQ_MAX_GROUP][HB] (k-NN modes)
  VcTableView* s_tv = (VcTableView*)(((uintptr_t)(s_hist + MQ_MAX_GROUP * HB) + 15) & ~(uintptr_t)15);      // [m]
  // s_seenc / s_hits0c [class]: distinct items verified / table 0's set leaves, per shell class of the current pass
  __shared__ uint32_t s_nh, s_ncand, s_seenc[MQ_MAX_GROUP], s_hits0c[MQ_MAX_GROUP], s_dk, s_wsum[MQ_BLK / VC_WAVE < 4 ? 4 : MQ_BLK / VC_WAVE];   // (s_wsum doubles as the 4 class counters of mq_select_exact)
  __shared__ uint32_t s_segstart[28], s_segh[27], s_segmask[27], s_segr[27], s_nseg;
  __shared__ uint64_t s_thresh;
  // entry -> bucket map of a drain: bit e set iff a bucket of the hit list starts at entry e, + the set bits before every word
  __shared__ uint32_t s_bm[MQ_BMW + 1], s_bmpre[MQ_BMW + 1];

  __shared__ unsigned long long s_t_entry;   // dev (VC_MIH_PHASES): thread 0's clock at entry -- in LDS, not in a register pair that lives to the end
  __shared__ unsigned long long s_ph_last, s_phacc[8];   // (phase times of THIS block: flushed once, by stop shell, when it ends --
  __shared__ uint32_t s_ph_cur;                            //  an atomic per phase change on launch-wide words made the instrumented kernel 2.4 x slower)
  __shared__ unsigned long long s_stat[5];
  const uint32_t slot = p.order ? p.order[blockIdx.x] : blockIdx.x;   // (block-uniform: a scalar load)
  const uint32_t tid = threadIdx.x, lane = vc_lane(), wave = tid / VC_WAVE;
  const uint32_t s = p.sbits, m = p.m;
  const uint32_t smask = s == 32 ? 0xFFFFFFFFu : ((1u << s) - 1u);
  const bool knn = p.mode != MQ_MODE_RADIUS;
  // Result / state pointers are read from the kernel-argument segment where they are used (the end of a query, once):
  // as ordinary by-value arguments they sat in ~40 SGPRs from entry to exit and were spilled to VGPR lanes around every loop.
  auto cold = [&]() {
    const __attribute__((address_space(4))) QueryKernelParams* q =
        (const __attribute__((address_space(4))) QueryKernelParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));   // opaque: the loads stay behind this point
    return q;
  };
  // per block, once: the tables' views (directory lines: a granule is then the first 16 bytes of its key range's 64-byte line,
  // MQ_LO = 7 <=> 128 keys per line)
  constexpr uint32_t gstride = LINES ? 16u : MQ_GW;   // words between consecutive granules
  for (uint32_t i = tid; i < m * (sizeof(VcTableView) / 4); i += MQ_BLK) ((uint32_t*)s_tv)[i] = ((const uint32_t*)p.tables)[i];
  if (LINES) {
    __syncthreads();
    if (tid < m) s_tv[tid].bitmap = (const uint32_t*)s_tv[tid].lines;
  }

  // ---- one query, start to end (a persistent block runs this for every query it draws)
  if (p.phase_dbg && threadIdx.x == 0) s_t_entry = __builtin_amdgcn_s_memrealtime();

  uint64_t qw[W];
#pragma unroll
  for (int j = 0; j < W; ++j) qw[j] = p.queries[(uint64_t)slot * W + j];
  auto qkey = [&](uint32_t t) {
    const uint32_t bp = t * s;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < W; ++j)
      if ((uint32_t)j == (bp >> 6)) v = (uint32_t)(qw[j] >> (bp & 63)) & smask;
    return v;
  };

  // the binomial table feeds combination unranking beyond the closed forms (|hi| >= 3: shells >= 3) and the <= 16-bit key
  // walk; a 32-bit k-NN query that stops in shells 0..2 -- most do -- never reads it, so it is copied when first needed
  auto load_binom = [&]() {         // (the barrier at the top of plan32 / before the first scan publishes it)
    for (uint32_t i = tid; i < 33 * MQ_BW; i += MQ_BLK) s_binom[i] = c_binom[i / MQ_BW][i % MQ_BW];
  };
  if (s != 32 || !knn || !MQ_LAZY_BINOM) load_binom();
  if (s == 32)
    for (uint32_t i = tid; i < m * MQ_NJ * MQ_GW; i += MQ_BLK) s_mask[i] = 0;
  if (knn)
    for (uint32_t i = tid; i < MQ_MAX_GROUP * HB; i += MQ_BLK) s_hist[i] = 0;
  if (tid < MQ_MAX_GROUP) {
    s_seenc[tid] = 0;
    s_hits0c[tid] = 0;
  }
  if (tid == 0) {
    s_nh = 0;
    s_ncand = 0;
    s_thresh = knn ? VC_PACK_INF : vc_pack(cold()->radius + 1, 0);
  }
  __syncthreads();
  if (s == 32) {   // E_j[t] = { x < 2^MQ_LO : popcount(x ^ qlo_t) = j }; radius search: balls B_j = E_0 | ... | E_j
    for (uint32_t i = tid; i < (m << MQ_LO); i += MQ_BLK) {
      const uint32_t t = i >> MQ_LO, x = i & ((1u << MQ_LO) - 1u);
      const uint32_t j = __popc(x ^ (qkey(t) & ((1u << MQ_LO) - 1u)));
      atomicOr(&s_mask[(t * MQ_NJ + j) * MQ_GW + (x >> 5)], 1u << (x & 31));
    }
    __syncthreads();
    if (!knn) {
      for (uint32_t i = tid; i < m * MQ_GW; i += MQ_BLK) {
        const uint32_t t = i / MQ_GW, w = i % MQ_GW;
        uint32_t acc = 0;
        for (uint32_t j = 0; j < MQ_NJ; ++j) {
          acc |= s_mask[(t * MQ_NJ + j) * MQ_GW + w];
          s_mask[(t * MQ_NJ + j) * MQ_GW + w] = acc;
        }
      }
      __syncthreads();
    }
  }

  // k-NN modes keep s_buf UNSORTED: s_buf[0 .. s_ncand) = the candidates that passed the running threshold, tagged with
  // their shell class inside a grouped pass (MQ_TAG_MASK).  The stop rule's k-th distance and the threshold come from a
  // distance histogram in LDS (no sort per shell: the bitonic network after every shell was most of this kernel's
  // instructions); the buffer is compacted when it fills and sorted once, when the query ends or is handed over.
  // VC_MIH_PHASES (dev): where a block's time goes -- phase times in 10 ns ticks, summed per phase over the launch
  auto tick = [&](uint32_t next) {   // ends the current phase, starts `next`
    if (p.phase_dbg && tid == 0) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      s_phacc[s_ph_cur] += now - s_ph_last;
      s_ph_last = now;
      s_ph_cur = next;
    }
  };
  if (p.phase_dbg && tid == 0) {
    for (int i = 0; i < 8; ++i) s_phacc[i] = 0;
    s_ph_last = __builtin_amdgcn_s_memrealtime();
    s_ph_cur = 0;
    s_phacc[7] = s_ph_last - s_t_entry;   // set-up: query, tables, binomials, masks
  }
  auto phase_flush = [&](uint32_t cls) {   // thread 0, at the block's end: cls = stop shell 0..4, 5 = handed over
    for (int i = 0; i < 8; ++i) {
      atomicAdd(&p.phase_dbg[i], s_phacc[i]);
      atomicAdd(&p.phase_dbg[32 + cls * 8 + i], s_phacc[i]);
    }
  };
  uint32_t r_base = 0;             // first shell of the current pass: a candidate's class = its substring distance - r_base (block-uniform)
  uint32_t kk = 0;                 // radius mode: sorted results in s_buf[0..kk)           (block-uniform)
  bool spilled = false;            // radius mode: results went to the global ring unsorted (block-uniform)
  uint32_t ring_fill = 0;          // radius mode: entries already in the global ring       (block-uniform)
  // get_stat counters of table 0 (sub, loc) and the algorithmic work of this query (probes, non-empty buckets, entries): thread 0's,
  // in LDS -- as block-uniform locals they were ten more scalar registers alive from entry to exit
  unsigned long long& sub = s_stat[0];
  unsigned long long& loc = s_stat[1];
  unsigned long long& w_probes = s_stat[2];
  unsigned long long& w_hits = s_stat[3];
  unsigned long long& w_entries = s_stat[4];
  if (tid < 5) s_stat[tid] = 0;   // (a barrier follows before the first use)
  auto ring_row = [&]() { return cold()->st.ring + (uint64_t)slot * cold()->cap; };   // radius mode only (k-NN hand-over goes through st.topk)
  // (r04: publishing the launch's counters to the host from the block that ends last -- a device-scope fence and a ticket per
  // block, so that no reduce kernel stands between the query kernel and the polling host -- made the kernel 60 % SLOWER, 0.443
  // against 0.274 ms per 4096 queries: a device-scope release writes the XCD's L2 back, 4096 times per launch.  The reduce
  // kernel behind the launch keeps that job.)
  auto put_work = [&]() {
    if (tid == 0) {
      cold()->st.work[slot * 4 + 0] = w_probes;
      cold()->st.work[slot * 4 + 1] = w_hits;
      cold()->st.work[slot * 4 + 2] = w_entries;
    }
  };

  // ---- sort s_buf[0 .. kk + ncand) and keep the k smallest: new top-k, new threshold
  auto merge = [&]() {
    __syncthreads();
    const uint32_t total = kk + min(s_ncand, p.buf_entries - kk);
    uint32_t P = 2;
    while (P < total) P <<= 1;
    for (uint32_t i = total + tid; i < P; i += MQ_BLK) s_buf[i] = VC_PACK_INF;
    vc_bitonic_lds(s_buf, P, MQ_BLK);
    kk = knn ? min(p.k, total) : total;
    if (tid == 0) {
      s_ncand = 0;
      if (knn) s_thresh = kk == p.k ? s_buf[p.k - 1] : VC_PACK_INF;
    }
    __syncthreads();
  };
  // ---- radius mode: move the LDS results to the global ring (unsorted; the count keeps running past cap)
  auto flush_ring = [&]() {
    __syncthreads();
    const uint32_t nc = s_ncand, rcap = cold()->cap;
    uint64_t* const ring = ring_row();
    for (uint32_t i = tid; i < nc; i += MQ_BLK)
      if (ring_fill + i < rcap) ring[ring_fill + i] = s_buf[i];
    ring_fill += nc;
    spilled = true;
    __syncthreads();
    if (tid == 0) s_ncand = 0;
    __syncthreads();
  };

  // ---- k-NN: rare paths of the candidate buffer live in functions of their own (mq_compact / mq_select_exact): inlined
  // into every drain() they made the compiler outline drain() itself, closure and all
  auto compact = [&](uint32_t max_cls, bool clear_tags) { mq_compact(s_buf, &s_ncand, &s_thresh, s_wsum, p.buf_entries, max_cls, clear_tags); };
  auto select_exact = [&]() { mq_select_exact(s_buf, &s_ncand, &s_thresh, s_wsum, p.buf_entries, p.k); };
  // ---- k-NN: sorted top-k of what the evaluated shells have seen -> s_buf[0 .. return value)
  auto finish_sort = [&](uint32_t max_cls) -> uint32_t {
    compact(max_cls, true);
    const uint32_t fill = s_ncand;
    if (fill <= MQ_BLK) {
      // the usual case (k + ties survive the final threshold): order by counting -- entry i goes to slot #{j : a[j] < a[i]},
      // packed values are distinct -- fill broadcast LDS reads and two barriers instead of a 36-stage network.
      // (r04: spreading the count over all four waves -- MQ_BLK / P threads per entry, partial ranks met in LDS -- is 5.7 % SLOWER
      // at 1e8, 13.15 vs 13.94 M queries/s: the kernel is bound by the instructions the CU issues in total, and waves that idle
      // here leave their issue slots to the other blocks of the CU.)
      const uint64_t v = tid < fill ? s_buf[tid] : VC_PACK_INF;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < fill; ++j) rank += s_buf[j] < v;
      __syncthreads();
      if (tid < fill) s_buf[rank] = v;
      __syncthreads();
      return min(fill, p.k);
    }
    uint32_t P = 2;
    while (P < fill) P <<= 1;
    for (uint32_t i = fill + tid; i < P; i += MQ_BLK) s_buf[i] = VC_PACK_INF;
    vc_bitonic_lds(s_buf, P, MQ_BLK);
    return min(fill, p.k);
  };

  // ---- drain the hit list: (rank -> offsets), prefix sum, balanced verify
  auto drain = [&]() __attribute__((always_inline)) {
    __syncthreads();
    tick(2);
    const uint32_t H = min(s_nh, MQ_HMAX);
    if (MQ_BMW) {                       // (published by the barrier in front of the prefix sums' second half)
      if (tid < MQ_BMW) s_bm[tid] = 0;
      if (tid == 0) s_bm[MQ_BMW] = 0;
    }
    if (LINES && s == 32) {
      // directory lines: occupancy, first entry position, rank and the buckets' extents sit in the ONE sector the scan just
      // read (four 16-byte loads that hit L2); only a line whose buckets fit neither encoding goes on to offsets[]
      for (uint32_t i = tid; i < H; i += MQ_BLK) {
        const uint32_t key = s_key[i];
        const VcTableView& tv = s_tv[s_meta[i] & 0xFFu];
        const uint4* ln = tv.lines + (uint64_t)(key >> 7) * 4;
        const uint4 occ = ln[0], dir = ln[1], f0 = ln[2], f1 = ln[3];
        uint32_t a, len, rk;
        if (!vc_line_lookup(occ, dir, f0, f1, key & 127u, a, len, rk)) {
          a = tv.offsets[rk];
          len = tv.offsets[rk + 1] - a;
        }
        s_key[i] = a;
        s_pref[i] = len;
      }
      __syncthreads();
    } else if (!LINES && s == 32) {
      for (uint32_t i = tid; i < H; i += MQ_BLK) {
        const uint32_t key = s_key[i];
        const VcTableView& tv = s_tv[s_meta[i] & 0xFFu];
        // set bits of the key's 256-bit block below the key / in all of it (the block is in the sector the scan just read)
        const uint32_t blk = key >> 8, wq = (key >> 5) & 7u;
        const uint32_t* bw = tv.bitmap + ((uint64_t)blk << 3);
        uint32_t below = 0, all = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
          const uint32_t wv = bw[j];
          all += __popc(wv);
          below += j < wq ? __popc(wv) : (j == wq ? __popc(wv & ((1u << (key & 31)) - 1u)) : 0u);
        }
        const uint2 d0 = tv.blockoff[blk], d1 = tv.blockoff[blk + 1];
        uint32_t a, len;
        if (d1.x - d0.x == all) {      // every bucket of the block holds one entry
          a = d0.x + below;
          len = 1;
        } else {
          const uint32_t rk = d0.y + below;
          a = tv.offsets[rk];
          len = tv.offsets[rk + 1] - a;
        }
        s_key[i] = a;
        s_pref[i] = len;
      }
      __syncthreads();
    }
    uint32_t lsum = 0, lv[MQ_HMAX / MQ_BLK];
#pragma unroll
    for (uint32_t i = 0; i < MQ_HMAX / MQ_BLK; ++i) {
      const uint32_t idx = tid * (MQ_HMAX / MQ_BLK) + i;
      lv[i] = idx < H ? s_pref[idx] : 0;
      lsum += lv[i];
    }
    uint32_t wtot;
    uint32_t excl = vc_wave_excl_scan(lsum, wtot);
    if (lane == 0) s_wsum[wave] = wtot;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < MQ_BLK / VC_WAVE; ++w) {
      if (w < wave) wbase += s_wsum[w];
      total += s_wsum[w];
    }
    excl += wbase;
#pragma unroll
    for (uint32_t i = 0; i < MQ_HMAX / MQ_BLK; ++i) {
      const uint32_t idx = tid * (MQ_HMAX / MQ_BLK) + i;
      if (idx < H) {
        s_pref[idx] = excl;
        if (MQ_BMW && excl < MQ_BMW * 32u) atomicOr(&s_bm[excl >> 5], 1u << (excl & 31u));   // (buckets of the list hold >= 1 entry: distinct starts)
      }
      excl += lv[i];
    }
    if (tid == 0) s_pref[H] = total;
    __syncthreads();
    if (tid == 0) {
      w_hits += H;
      w_entries += total;
    }
    // Which bucket does entry e belong to?  Four binary searches in the prefix array per thread and round were the largest
    // single item of a verified entry's instructions AND a chain of ~7 dependent LDS reads each (r04).  A drain of up to
    // 32 * MQ_BMW entries marks every bucket's first entry in a bitmap and keeps the set bits before every word: the bucket of
    // e is then two independent LDS reads and a popcount.
    // (r04, same box: exact top-100 + 2 % at 1e9 -- 1 714 entries per query --, + 0.5 % at 1e8, approximate mode equal; the
    // radius search, whose buckets hold one entry each, loses 1.5 % to the map's two barriers and keeps its searches)
    const bool fastmap = MQ_BMW && knn && total <= MQ_BMW * 32u;     // (block-uniform)
    if (fastmap) {
      const uint32_t c = tid < MQ_BMW ? __popc(s_bm[tid]) : 0u;
      uint32_t wt;
      uint32_t ex = vc_wave_excl_scan(c, wt);
      if (lane == 0) s_wsum[wave] = wt;
      __syncthreads();
#pragma unroll
      for (uint32_t w = 0; w < MQ_BLK / VC_WAVE; ++w)
        if (w < wave) ex += s_wsum[w];
      if (tid < MQ_BMW) s_bmpre[tid] = ex;
      __syncthreads();
    }
    tick(3);

    uint32_t seen_acc = 0, seen_acc1 = 0, seen_acc2 = 0;   // per shell class (wave-uniform)
    for (uint32_t e0 = 0; e0 < total; e0 += MQ_ROUND) {
      // room for one round of survivors behind what the buffer already holds.  The fill is read between two barriers:
      // the first ends the previous round's appends, the second keeps a fast wave's appends of THIS round from being
      // seen by a wave that has not read yet (a lone wave taking the merge branch would deadlock the block).
      __syncthreads();
      const uint32_t fill = s_ncand;
      __syncthreads();
      if (kk + fill + MQ_ROUND > p.buf_entries) {
        if (knn) {
          compact(MQ_MAX_GROUP, false);
          if (s_ncand + MQ_ROUND > p.buf_entries) select_exact();
        } else {
          flush_ring();
        }
      }
      const uint64_t thresh = s_thresh;
      // a wave none of whose lanes has an entry in this round (the tail round of a drain: 1 055 entries are one full round
      // and 31 entries for wave 0) has nothing to search, load or verify -- entry e0 + wave * 64 is its first
      if (e0 + wave * VC_WAVE >= total) continue;
      uint32_t local[MQ_EPT], meta[MQ_EPT];
      uint64_t x[MQ_EPT][W];
      bool live[MQ_EPT];
#pragma unroll
      for (uint32_t g = 0; g < MQ_EPT; ++g) {
        const uint32_t e = e0 + g * MQ_BLK + tid;
        live[g] = e < total;
        const uint32_t ec = live[g] ? e : 0;
        uint32_t lo = 0, hi = H;               // largest b with s_pref[b] <= e
        if (fastmap) {
          const uint32_t wi = ec >> 5;
          lo = s_bmpre[wi] + __popc(s_bm[wi] & (0xFFFFFFFFu >> (31u - (ec & 31u)))) - 1u;   // bucket starts at or before e, minus one
        } else {
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_pref[mid] <= ec) lo = mid; else hi = mid;
          }
        }
        meta[g] = s_meta[lo];
        const VcTableView& tv = s_tv[meta[g] & 0xFFu];
        const uint32_t pos = s_key[lo] + (ec - s_pref[lo]);
        if (W <= 2 && tv.bent) {
          const uint4 rec = tv.bent[(uint64_t)pos * W];
          local[g] = rec.x;
          x[g][0] = ((uint64_t)rec.w << 32) | rec.z;
          if (W == 2) {
            const uint4 rec1 = tv.bent[(uint64_t)pos * 2 + 1];
            x[g][W - 1] = ((uint64_t)rec1.y << 32) | rec1.x;
          }
          continue;
        }
        local[g] = tv.ids[pos];
        if (tv.bcodes) {
#pragma unroll
          for (int j = 0; j < W; ++j) x[g][j] = tv.bcodes[(uint64_t)j * p.n + pos];
        } else {
#pragma unroll
          for (int j = 0; j < W; ++j) x[g][j] = p.cols[(uint64_t)j * p.stride + local[g]];
        }
      }
#pragma unroll
      for (uint32_t g = 0; g < MQ_EPT; ++g) {
        bool emit = live[g];
        uint64_t packed = 0;
        if (live[g] && MQ_FAST_OWNER && s == 32 && m == 2u * W) {
          // the reference's native shape -- 32-bit substrings, every table one half of a code word -- unrolled: the substring
          // distances are the popcounts of the 2 W dwords of x ^ q, no field extraction, no loop over a run-time table count
          const uint32_t t = meta[g] & 0xFFu, dt = meta[g] >> 8;
          uint32_t dist = 0;
#pragma unroll
          for (int j = 0; j < W; ++j) {
            const uint64_t xq = x[g][j] ^ qw[j];
            const uint32_t d0 = __popc((uint32_t)xq), d1 = __popc((uint32_t)(xq >> 32));
            dist += d0 + d1;
            // owner rule: reported by the first table holding the minimum substring distance
            if (((uint32_t)(2 * j) != t && (d0 < dt || (d0 == dt && (uint32_t)(2 * j) < t))) ||
                ((uint32_t)(2 * j + 1) != t && (d1 < dt || (d1 == dt && (uint32_t)(2 * j + 1) < t)))) emit = false;
          }
          packed = vc_pack(dist, p.id_base + local[g]);
        } else if (live[g]) {
          const uint32_t t = meta[g] & 0xFFu, dt = meta[g] >> 8;
          uint32_t dist = 0;
          for (uint32_t tt = 0; tt < m; ++tt) {
            const uint32_t bp = tt * s;
            uint32_t field = 0;
#pragma unroll
            for (int j = 0; j < W; ++j)
              if ((uint32_t)j == (bp >> 6)) field = (uint32_t)((x[g][j] ^ qw[j]) >> (bp & 63)) & smask;
            const uint32_t d = __popc(field);
            dist += d;
            // owner rule (see mih_probe_kernel): reported by the first table holding the minimum substring distance
            bool reach = true;
            if ((p.flags & VC_FLAG_REF_SIGNEXT_KEYS) && s < 32) reach = ((field >> (s - 1)) & 1u) == 0;
            if (tt != t && reach && (d < dt || (d == dt && tt < t))) emit = false;
          }
          packed = vc_pack(dist, p.id_base + local[g]);
        }
        const uint64_t emask = __ballot(emit);
        if (emask == 0) continue;
        const uint32_t cls = knn ? (meta[g] >> 8) - r_base : 0u;   // shell class inside a grouped pass (0 for single-shell passes)
        const uint64_t m1 = __ballot(emit && cls == 1), m2 = __ballot(emit && cls == 2);
        seen_acc1 += (uint32_t)__popcll(m1);
        seen_acc2 += (uint32_t)__popcll(m2);
        seen_acc += (uint32_t)__popcll(emask & ~(m1 | m2));
        const bool keep = emit && packed < thresh;
        const uint64_t kmask = __ballot(keep);
        if (kmask == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_ncand, (uint32_t)__popcll(kmask));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep) {
          s_buf[kk + base + (uint32_t)__popcll(kmask & ((1ull << lane) - 1ull))] = packed | ((uint64_t)cls << MQ_TAG_SHIFT);
          if (knn) atomicAdd(&s_hist[cls * HB + (uint32_t)(packed >> 32)], 1u);
        }
      }
    }
    if (lane == 0 && seen_acc) atomicAdd(&s_seenc[0], seen_acc);
    if (lane == 0 && seen_acc1) atomicAdd(&s_seenc[1], seen_acc1);
    if (lane == 0 && seen_acc2) atomicAdd(&s_seenc[2], seen_acc2);
    __syncthreads();
    if (tid == 0) s_nh = 0;
    __syncthreads();
    tick(1);
  };

  // ---- one shell (or the whole ball) of ALL tables, 32-bit substrings: granule scan over the segments in s_seg*.
  // The tables share one item space (item = table * per_table + pattern index), so the 1-, 26- and 326-granule shells
  // 0..2 of four tables are 1 + 1 + 2 passes, not 4 + 4 + 4: a pass costs a barrier and a memory round trip whatever it holds.
  uint32_t tb_first = 0, tb_count = m;   // the tables a scan covers (radius mode narrows them, see below)
  auto scan32 = [&]() {
    const uint32_t per_table = s_segstart[s_nseg];
    const uint32_t total = per_table * tb_count;
    for (uint32_t base = 0; base < total; base += MQ_PASS32) {
      const uint32_t idx0 = base + tid * MQ_G32;
      uint32_t t = 0, rem = 0, seg = 0, hi = 0;
      if (idx0 < total) {
        t = idx0 / per_table;
        rem = idx0 - t * per_table;
        t += tb_first;
        while (rem >= s_segstart[seg + 1]) ++seg;
        hi = mq_unrank(s_binom, rem - s_segstart[seg], s_segh[seg], MQ_HI);
      }
      uint4 v[MQ_G32][MQ_GW / 4];
      uint32_t gr[MQ_G32], meta0[MQ_G32];   // meta0 = table | |hi| << 8 | mask index << 16 | shell class << 30
#pragma unroll
      for (uint32_t g = 0; g < MQ_G32; ++g) {
        const uint32_t idx = idx0 + g;
        gr[g] = 0; meta0[g] = 0;
#pragma unroll
        for (uint32_t c = 0; c < MQ_GW / 4; ++c) v[g][c] = make_uint4(0, 0, 0, 0);
        if (idx < total) {
          if (g) {
            ++rem;
            if (rem == per_table) {                 // next table: its first pattern
              ++t;
              rem = 0;
              seg = 0;
              hi = (1u << s_segh[0]) - 1u;
            } else if (rem == s_segstart[seg + 1]) {
              ++seg;
              hi = (1u << s_segh[seg]) - 1u;
            } else {
              hi = vc_next_comb(hi);
            }
          }
          gr[g] = (qkey(t) >> MQ_LO) ^ hi;
          meta0[g] = t | (s_segh[seg] << 8) | (s_segmask[seg] << 16) | ((s_segr[seg] - r_base) << 30);   // bits 30..31: shell class of the segment
          const uint4* gp = reinterpret_cast<const uint4*>(s_tv[t].bitmap + (uint64_t)gr[g] * gstride);
#pragma unroll
          for (uint32_t c = 0; c < MQ_GW / 4; ++c) v[g][c] = gp[c];
        }
      }
      uint32_t w[MQ_G32][MQ_GW];
      uint32_t cnt = 0;
#pragma unroll
      for (uint32_t g = 0; g < MQ_G32; ++g) {
        const uint32_t* mk = s_mask + ((meta0[g] & 0xFFu) * MQ_NJ + ((meta0[g] >> 16) & 0xFu)) * MQ_GW;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t cc = 0; cc < MQ_GW / 4; ++cc) {
          w[g][4 * cc + 0] = v[g][cc].x & mk[4 * cc + 0];
          w[g][4 * cc + 1] = v[g][cc].y & mk[4 * cc + 1];
          w[g][4 * cc + 2] = v[g][cc].z & mk[4 * cc + 2];
          w[g][4 * cc + 3] = v[g][cc].w & mk[4 * cc + 3];
          c += __popc(w[g][4 * cc + 0]) + __popc(w[g][4 * cc + 1]) + __popc(w[g][4 * cc + 2]) + __popc(w[g][4 * cc + 3]);
        }
        cnt += c;
      }
      if (p.flags & VC_FLAG_USE_BITMAP) {   // n_sub_reads_ of table 0 = its leaves whose bit is set (search_worker.cc:238-245)
        uint32_t cnt0[MQ_MAX_GROUP] = {0, 0, 0};   // (counted here, not in the loop above: three registers fewer while the granules are in flight)
#pragma unroll
        for (uint32_t g = 0; g < MQ_G32; ++g)
          if ((meta0[g] & 0xFFu) == 0 && tb_first == 0 && idx0 + g < per_table) {
            uint32_t c = 0;
#pragma unroll
            for (uint32_t i = 0; i < MQ_GW; ++i) c += __popc(w[g][i]);
#pragma unroll
            for (uint32_t cc = 0; cc < MQ_MAX_GROUP; ++cc) cnt0[cc] += (meta0[g] >> 30) == cc ? c : 0u;
          }
#pragma unroll
        for (uint32_t cc = 0; cc < MQ_MAX_GROUP; ++cc) {
          uint32_t wt;
          (void)vc_wave_excl_scan(cnt0[cc], wt);
          if (lane == 0 && wt) atomicAdd(&s_hits0c[cc], wt);
        }
      }
      for (;;) {   // append this pass's hits; what does not fit waits for a drain
        if (!__syncthreads_or(cnt != 0)) break;
        uint32_t wtot;
        const uint32_t off = vc_wave_excl_scan(cnt, wtot);
        uint32_t wb = 0;
        if (lane == 0 && wtot) wb = atomicAdd(&s_nh, wtot);
        uint32_t pos = __builtin_amdgcn_readfirstlane(wb) + off;
#pragma unroll
        for (uint32_t g = 0; g < MQ_G32; ++g) {
          const uint32_t qlo = qkey(meta0[g] & 0xFFu) & ((1u << MQ_LO) - 1u);
#pragma unroll
          for (uint32_t i = 0; i < MQ_GW; ++i)
            while (w[g][i] && pos < MQ_HMAX) {
              const uint32_t bb = (uint32_t)__ffs((int)w[g][i]) - 1u;
              w[g][i] &= w[g][i] - 1u;
              const uint32_t x = i * 32 + bb;
              s_key[pos] = (gr[g] << MQ_LO) | x;
              s_meta[pos] = (meta0[g] & 0xFFFFu) + (__popc(x ^ qlo) << 8);
              ++pos;
              --cnt;
            }
        }
        __syncthreads();
        const uint32_t nh = s_nh;
        if (nh >= MQ_HFLUSH) drain();
        if (nh <= MQ_HMAX) break;               // everything fitted
      }
    }
  };

  // ---- one shell of ALL tables, <= 16-bit substrings: direct offsets, keys by combination unranking + Gosper
  auto scan_direct = [&](uint32_t r) {
    const uint32_t nkeys = s_binom[s * MQ_BW + r];
    const uint32_t total = nkeys * tb_count;
    for (uint32_t base = 0; base < total; base += MQ_PASS) {
      const uint32_t j0 = base + tid * MQ_G;
      uint32_t t = 0, rem = 0, mask = 0;
      if (j0 < total) {
        t = j0 / nkeys;
        rem = j0 - t * nkeys;
        t += tb_first;
        mask = mq_unrank(s_binom, rem, r, s);
      }
      uint32_t offv[MQ_G], lenv[MQ_G], tt[MQ_G];
      uint32_t cnt = 0, cnt0 = 0;
#pragma unroll
      for (uint32_t g = 0; g < MQ_G; ++g) {
        offv[g] = 0;
        lenv[g] = 0;
        tt[g] = 0;
        if (j0 + g < total) {
          if (g) {
            ++rem;
            if (rem == nkeys) {                     // next table: its first key of the shell
              ++t;
              rem = 0;
              mask = r ? (1u << r) - 1u : 0u;
            } else {
              mask = vc_next_comb(mask);
            }
          }
          tt[g] = t;
          // binaryToInt's sign-extended keys (Pilaf/image_tools.h:13): a probe that flips the top bit matches nothing
          const bool dead = (p.flags & VC_FLAG_REF_SIGNEXT_KEYS) && ((mask >> (s - 1)) & 1u);
          if (!dead) {
            const uint32_t key = qkey(t) ^ mask;
            const uint32_t a0 = s_tv[t].offsets[key], b0 = s_tv[t].offsets[key + 1];
            offv[g] = a0;
            lenv[g] = b0 - a0;
          }
          cnt += lenv[g] != 0;
          if (t == 0) cnt0 += lenv[g] != 0;
        }
      }
      if (p.flags & VC_FLAG_USE_BITMAP) {
        uint32_t wt;
        (void)vc_wave_excl_scan(cnt0, wt);
        if (lane == 0 && wt) atomicAdd(&s_hits0c[0], wt);
      }
      for (;;) {
        if (!__syncthreads_or(cnt != 0)) break;
        uint32_t wtot;
        const uint32_t off = vc_wave_excl_scan(cnt, wtot);
        uint32_t wb = 0;
        if (lane == 0 && wtot) wb = atomicAdd(&s_nh, wtot);
        uint32_t pos = __builtin_amdgcn_readfirstlane(wb) + off;
#pragma unroll
        for (uint32_t g = 0; g < MQ_G; ++g)
          if (lenv[g] && pos < MQ_HMAX) {
            s_key[pos] = offv[g];
            s_pref[pos] = lenv[g];
            s_meta[pos] = tt[g] | (r << 8);
            lenv[g] = 0;
            ++pos;
            --cnt;
          }
        __syncthreads();
        const uint32_t nh = s_nh;
        if (nh >= MQ_HFLUSH) drain();
        if (nh <= MQ_HMAX) break;
      }
    }
  };

  // ---- segments of one pass over a 32-bit table: exact shell r (masks E_{r-h}) or the whole ball (masks B_{R-h})
  auto plan32 = [&](uint32_t r_lo, uint32_t r_hi, bool ball) {   // (ball: one "shell" r_lo = r_hi with the cumulative masks)
    __syncthreads();
    if (tid == 0) {
      uint32_t ns = 0, start = 0;
      for (uint32_t r = r_lo; r <= r_hi; ++r)
        for (uint32_t h = 0; h <= min(r, MQ_HI); ++h) {
          const uint32_t j = r - h;
          if (!ball && j > MQ_LO) continue;        // the low part holds at most MQ_LO flips
          s_segstart[ns] = start;
          s_segh[ns] = h;
          s_segmask[ns] = min(j, MQ_LO);
          s_segr[ns] = r;
          // C(MQ_HI, h): closed form while the LDS table may not exist yet (it is copied before the pass of shell 3)
          start += h == 0 ? 1u : (h == 1 ? MQ_HI : (h == 2 ? MQ_HI * (MQ_HI - 1u) / 2u : s_binom[MQ_HI * MQ_BW + h]));
          ++ns;
        }
      s_segstart[ns] = start;
      s_nseg = ns;
    }
    __syncthreads();
  };

  tick(1);                         // (phase 0 = set-up: tables, binomials, masks -- but the phase clock starts after them)
  const uint32_t S = s;            // loop bound radius <= n_local_bytes_ * 8 (search_worker.cc:170)
  if (!knn) {
    // fixed-radius neighbour search, every item within the full distance R kept.  Pigeonhole with the sharper radii of
    // multi-index hashing: R = m q + a  =>  tables 0..a search substring radius q, tables a+1..m-1 only q - 1 (were every
    // substring beyond its radius the distance would be >= (a+1)(q+1) + (m-a-1) q = R + 1).  The owner rule of the drain
    // (lowest table among those with the smallest substring distance) needs no change: the radii do not increase with the
    // table number, so the owner of an item within R always lies inside its own radius.  configs[1] (R = 8, m = 2: radii
    // 4 and 3) probes 12 951 instead of 21 806 bitmap sectors per query.
    const uint32_t n_big = min(cold()->n_big, m), small_shells = cold()->small_shells;
    if (s == 32) {
      tb_first = 0; tb_count = n_big;
      plan32(p.r_last, p.r_last, true);
      scan32();
      if (n_big < m && small_shells) {
        tb_first = n_big; tb_count = m - n_big;
        plan32(small_shells - 1, small_shells - 1, true);
        scan32();
      }
    } else {
      for (uint32_t r = 0; r <= p.r_last; ++r) {
        tb_first = 0; tb_count = r < small_shells ? m : n_big;
        scan_direct(r);
      }
    }
    tb_first = 0; tb_count = m;
    if (s_nh) drain();
    __syncthreads();
    if (tid == 0)
      for (uint32_t r = 0; r <= p.r_last; ++r) w_probes += (unsigned long long)(r < small_shells ? m : n_big) * c_binom[s][r];
    put_work();
    if (!spilled) {
      merge();                                       // sorts the LDS results (kk = their number)
      const uint32_t rcap = cold()->cap;
      uint64_t* const ring = ring_row();
      for (uint32_t i = tid; i < kk; i += MQ_BLK)
        if (i < rcap) ring[i] = s_buf[i];
      if (tid == 0) {
        cold()->st.count[slot] = kk;
        cold()->st.topn[slot] = kk <= rcap ? 1u : 0u;     // 1 = the ring segment is already sorted
      }
    } else {
      flush_ring();
      if (tid == 0) {
        cold()->st.count[slot] = ring_fill;
        cold()->st.topn[slot] = 0;
      }
    }
    return;
  }

  for (uint32_t r = 0; r <= p.r_last;) {
    // The first pass of 32-bit substrings covers shells 0 .. group-1 at once (their 1 + 26 (+ 326) granules per table):
    // candidates are tagged with their shell class and counted per class, the stop rule is evaluated shell by shell
    // afterwards -- no result or statistic changes, a scan / drain round is saved per grouped shell
    const uint32_t r_hi = (s == 32 && r == 0) ? min(p.group ? p.group - 1 : 0u, min(p.r_last, MQ_MAX_GROUP - 1)) : r;
    r_base = r;
    if (s == 32) {
      if (MQ_LAZY_BINOM && r_hi >= 3) load_binom();   // (again for every later shell: 561 words next to a pass of >= 10 504 granules)
      plan32(r, r_hi, false);
      scan32();
    } else {
      scan_direct(r);
    }
    if (s_nh) drain();
    __syncthreads();
    tick(4);
    for (uint32_t rr = r; rr <= r_hi; ++rr) {
      const uint32_t cls = rr - r;
      if (cls) {                     // next shell of the pass: its candidates join the evaluated set (class 0 counters)
        for (uint32_t i = tid; i < HB; i += MQ_BLK) {
          s_hist[i] += s_hist[cls * HB + i];
          s_hist[cls * HB + i] = 0;
        }
        if (tid == 0) {
          s_seenc[0] += s_seenc[cls];
          s_seenc[cls] = 0;
          s_hits0c[0] = s_hits0c[cls];
          s_hits0c[cls] = 0;
        }
        __syncthreads();
      }
      // get_stat counters of table 0 (rank 0's, search_worker.cc:24-30): every leaf is a bitmap test when the bitmap
      // is attached (:239) and a get only where the bit is set (:245); without it every leaf is a get
      if (tid == 0) {
        const unsigned long long leaves = c_binom[s][rr];
        w_probes += leaves * m;
        if (p.flags & VC_FLAG_USE_BITMAP) {
          loc += leaves;
          sub += s_hits0c[0];
        } else {
          sub += leaves;
        }
      }
      const uint32_t seen = s_seenc[0];
      __syncthreads();
      if (tid == 0) s_hits0c[0] = 0;
      if (wave == 0) {               // k-th distance among the candidates of the shells evaluated so far
        const uint32_t cut = seen >= p.k ? mw_hist_cut(s_hist, W * 64u + 1u, p.k) : 0xFFFFFFFFu;
        if (lane == 0) s_dk = cut;
      }
      __syncthreads();
      const uint32_t dk = s_dk;
      bool stop;
      if (p.mode == MQ_MODE_APPROX)   // search_worker.cc:136-137: the heap of k*20 distinct candidates is full
        stop = seen >= p.k * MIH_APPROX_FACTOR;
      else                            // search_worker.cc:201-205: size == k && top.dist <= radius * 4 (radius already incremented)
        stop = dk != 0xFFFFFFFFu && dk <= (rr + 1) * p.stop_mult;
      if (tid == 0 && dk != 0xFFFFFFFFu) {   // everything farther than the k-th distance is out for good
        const uint64_t bnd = ((uint64_t)dk + 1) << 32;
        if (bnd < s_thresh) s_thresh = bnd;
      }
      if (stop || rr == S) {
        tick(5);
        const uint32_t kout = finish_sort(cls);   // classes beyond the shell the loop stops in were never seen by it
        put_work();
        for (uint32_t i = tid; i < p.k; i += MQ_BLK) cold()->out[(uint64_t)slot * p.k + i] = i < kout ? s_buf[i] : VC_PACK_INF;
        if (tid == 0) {
          cold()->out_cnt[slot] = kout;
          cold()->st.radius[slot] = rr;     // find() returns radius - 1 = last shell searched
          cold()->st.seen[slot] = seen;
          cold()->st.sub[slot] = sub;
          cold()->st.loc[slot] = loc;
          if (cold()->radius_hist) atomicAdd(&cold()->radius_hist[min(rr, 3u)], 1u);
        }
        tick(6);
        if (p.phase_dbg && tid == 0) {
          const unsigned long long now = __builtin_amdgcn_s_memrealtime();
          atomicAdd(&p.phase_dbg[8 + min(rr, 4u)], now - s_t_entry);
          atomicAdd(&p.phase_dbg[16 + min(rr, 4u)], 1ull);
          atomicMin(&p.phase_dbg[24], s_t_entry);
          atomicMax(&p.phase_dbg[25 + min(rr, 4u)], now);
          phase_flush(min(rr, 4u));
        }
        return;
      }
      __syncthreads();
    }
    if (r_hi != r) {                 // the pass is over: its candidates are ordinary (class 0) entries from now on
      const uint32_t fill = min(s_ncand, p.buf_entries);
      for (uint32_t i = tid; i < fill; i += MQ_BLK) s_buf[i] &= ~MQ_TAG_MASK;
      __syncthreads();
    }
    tick(1);
    r = r_hi + 1;
  }
  tick(5);
  // not finished: hand the query to the multi-block shells (state exactly as mih_commit_kernel leaves it)
  const uint32_t kout = finish_sort(0);
  put_work();
  for (uint32_t i = tid; i < kout; i += MQ_BLK) cold()->st.topk[(uint64_t)slot * p.k + i] = s_buf[i];   // mih_seed_ring_kernel moves it into the ring
  if (tid == 0) {
    cold()->st.count[slot] = kout;
    cold()->st.prev[slot] = kout;
    cold()->st.thresh[slot] = kout == p.k ? s_buf[p.k - 1] : VC_PACK_INF;
    cold()->st.seen[slot] = s_seenc[0];
    cold()->st.sub[slot] = sub;
    cold()->st.loc[slot] = loc;
    cold()->st.radius[slot] = 0;
    cold()->st.topn[slot] = kout;
    cold()->heavy_list[atomicAdd(cold()->heavy_ctr, 1u)] = slot;
    if (cold()->radius_hist) atomicAdd(&cold()->radius_hist[3], 1u);
    // what the radius loop may still cost this query: the k-th distance so far bounds the true one from above, so the loop
    // ends by shell floor(D_bound / mult) at the latest (search_worker.cc:201-205); all shells when k items are not found yet
    const uint32_t r_end = kout == p.k ? min(S, (uint32_t)(s_buf[p.k - 1] >> 32) / max(p.stop_mult, 1u)) : S;
    unsigned long long pred = 0;
    for (uint32_t r = p.r_last + 1; r <= r_end; ++r) pred += (unsigned long long)m * c_binom[s][r];
    cold()->st.work[slot * 4 + 3] = pred;
    if (p.phase_dbg) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      atomicAdd(&p.phase_dbg[13], now - s_t_entry);
      atomicAdd(&p.phase_dbg[21], 1ull);
      atomicMin(&p.phase_dbg[24], s_t_entry);
      atomicMax(&p.phase_dbg[30], now);
      phase_flush(5);
    }
  }
}

// =============================================================================================================
// mih_bucket_stream_kernel -- fixed-radius search over <= 16-bit substrings (BASELINE configs[1], m = 4 x 16 bit):
// the buckets hold ~N / 2^s entries each (1 526 at 1e8 codes), so a query is not a chain of probes but a STREAM of
// bucket entries -- 188 buckets, 287 K entries per query at R = 8 -- and the work is HBM-bound byte work:
//   * a block takes one query (or one of `split` interleaved parts of its probe list), its threads look the probes up
//     once (key by combination unranking, direct offsets: search_worker.cc:230-246) into an LDS list of (offset, length);
//   * every WAVE then streams whole buckets from the table's bucket-order code copy (VcTableView::bcodes: contiguous,
//     two entries per lane and 16-byte load = 1 KB per wave-instruction, MS_UP of them in flight per lane) and tests the
//     FULL distance only: xor + popcount + compare per entry, no id load, no LDS search, no workgroup barrier in the loop;
//   * an entry within the radius pays for the owner rule (per-substring distances) and is staged in the wave's LDS buffer
//     as (table, distance, position); every ~64-128 results the wave reserves ring space with ONE atomic and gathers
//     their ids together (a returning atomic and a gather per group of hits stalled the stream ~500 times per query).
// Algorithmic bytes: probes x 8 (two offsets) + entries x B/8; measured by the kernel's own counters (vc_timing.mih_*).
// =============================================================================================================
#ifndef MS_U
#define MS_U 8u                          // bucket entries per lane in flight
#endif
#ifndef MS_UP
#define MS_UP 8u                         // ... 16-byte pairs per lane in flight (pair path: 128 bytes per lane)
#endif
#ifndef MS_STAGE
#define MS_STAGE 128u                     // results a wave stages in LDS between two flushes (>= 128)
#endif
#ifndef MS_MAXP
#define MS_MAXP 2048u                    // probes of one query (all tables, all shells) held in LDS
#endif

#define MS_TOT_LINES 256u                 // copies of the launch's work counters, one 128-byte line each

struct StreamParams {
  const uint64_t* queries;               // [nq][W]
  const VcTableView* tables;
  uint64_t* ring;                        // [nq][cap]
  uint32_t* count;                       // [nq]
  unsigned long long* totals;            // [MS_TOT_LINES][16]: probes | non-empty buckets | entries | queries, summed by vc_mih_timing
  uint64_t n;                            // entries per table (stride of the bucket-order copies)
  uint32_t m, sbits, id_base, flags, cap, radius;
  uint32_t rsub, n_big, small_shells;    // shells 0..rsub for tables 0..n_big-1, shells 0..small_shells-1 for the others
  uint32_t nprobes, split;
  unsigned long long* trace;             // dev (VC_STREAM_TRACE): per block start | probes looked up | end, 10 ns ticks
};

template <int W>
__global__ void __launch_bounds__(256) mih_bucket_stream_kernel(const StreamParams p) {
  __shared__ uint32_t s_off[MS_MAXP], s_len[MS_MAXP], s_meta[MS_MAXP];
  __shared__ uint64_t s_stage[256 / VC_WAVE][MS_STAGE];   // per wave: results awaiting their ids and ring space
  __shared__ unsigned long long s_tot[2];
  const uint32_t slot = blockIdx.x / p.split, part = blockIdx.x % p.split;
  const uint32_t tid = threadIdx.x, lane = vc_lane(), wave = tid / VC_WAVE;
  const uint32_t s = p.sbits, m = p.m, smask = (1u << s) - 1u;
  uint64_t qw[W];
#pragma unroll
  for (int j = 0; j < W; ++j) qw[j] = p.queries[(uint64_t)slot * W + j];
  auto qkey = [&](uint32_t t) {
    const uint32_t bp = t * s;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < W; ++j)
      if ((uint32_t)j == (bp >> 6)) v = (uint32_t)(qw[j] >> (bp & 63)) & smask;
    return v;
  };
  if (tid < 2) s_tot[tid] = 0;
  if (p.trace && tid == 0) p.trace[3ull * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  // ---- this block's probes: j = part, part + split, ...  ->  (shell r, table t, index in the shell) -> bucket
  uint32_t nloc = 0;
  unsigned long long hits = 0, entries = 0;
  for (uint32_t li = tid;; li += blockDim.x) {
    uint32_t j = part + li * p.split;
    if (j >= p.nprobes) break;
    uint32_t r = 0, t = 0;
    for (;; ++r) {
      const uint32_t per = c_binom[s][r], cnt = (r < p.small_shells ? m : p.n_big) * per;
      if (j < cnt) {
        t = j / per;
        j -= t * per;
        break;
      }
      j -= cnt;
    }
    const uint32_t mask = r ? vc_unrank(j, r, s) : 0u;
    // binaryToInt's sign-extended keys (Pilaf/image_tools.h:13): a probe that flips the top bit matches nothing
    const bool dead = (p.flags & VC_FLAG_REF_SIGNEXT_KEYS) && ((mask >> (s - 1)) & 1u);
    uint32_t off = 0, len = 0;
    if (!dead) {
      const uint32_t key = qkey(t) ^ mask;
      off = p.tables[t].offsets[key];
      len = p.tables[t].offsets[key + 1] - off;
    }
    s_off[li] = off;
    s_len[li] = len;
    s_meta[li] = t | (r << 8);
    hits += len != 0;
    entries += len;
  }
  nloc = (p.nprobes > part) ? (p.nprobes - part + p.split - 1) / p.split : 0;
  if (hits) atomicAdd(&s_tot[0], hits);
  if (entries) atomicAdd(&s_tot[1], entries);
  __syncthreads();
  // (r04: these three atomics used to go to ONE 32-byte record, 5 121 of them per 1024-query launch -- same-address atomics retire
  // one every ~9 ns chip-wide, and wave 0 of every block waited for its own before it could consume its first bucket:
  // 0.44 -> 0.40 ms per launch without them, profiles/r04_stream_trace.txt.  Now MS_TOT_LINES records, one per 128-byte line.)
  if (tid == 0 && p.totals) {
    unsigned long long* const tot = p.totals + (size_t)(blockIdx.x % MS_TOT_LINES) * 16;
    atomicAdd(&tot[1], s_tot[0]);
    atomicAdd(&tot[2], s_tot[1]);
    if (part == 0) atomicAdd(&tot[0], (unsigned long long)p.nprobes);
    if (blockIdx.x == 0) atomicAdd(&tot[3], (unsigned long long)(gridDim.x / p.split));   // queries of the launch
  }
  if (p.trace && tid == 0) p.trace[3ull * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  // ---- every wave streams whole buckets
  uint64_t* const ring = p.ring + (uint64_t)slot * p.cap;
  // Results are staged per wave: a group of hits used to cost the wave a returning global atomic (ring space) and a gather
  // (ids[pos]) on the spot -- two round trips in the middle of the stream, ~500 times per query; now one of each per
  // MS_STAGE - 64 .. MS_STAGE results.
  uint64_t* const stage = s_stage[wave];
  uint32_t fill = 0;                                   // (wave-uniform)
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the wave's own LDS writes, in program order (see mih_tile... history)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&p.count[slot], fill);
    base = __builtin_amdgcn_readfirstlane(base);
    for (uint32_t i = lane; i < fill; i += VC_WAVE) {
      const uint64_t h = stage[i];
      const uint32_t at = base + i;
      if (at < p.cap) ring[at] = vc_pack((uint32_t)(h >> 32) & 0xFFFFu, p.id_base + p.tables[(uint32_t)(h >> 48)].ids[(uint32_t)h]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    fill = 0;
  };
  // one entry of bucket b (table t, substring distance dt of its key), code words x, position pos in the table's entry arrays
  auto entry = [&](const uint64_t (&x)[W], bool valid, uint32_t pos, uint32_t t, uint32_t dt) {
    uint32_t dist = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) dist += (uint32_t)__popcll(x[j] ^ qw[j]);
    bool hit = valid && dist <= p.radius;
    if (__ballot(hit) == 0) return;
    if (hit) {   // owner rule (mih_probe_kernel): reported by the first table holding the minimum substring distance
      for (uint32_t tt = 0; tt < m; ++tt) {
        const uint32_t bp = tt * s;
        uint32_t field = 0;
#pragma unroll
        for (int j = 0; j < W; ++j)
          if ((uint32_t)j == (bp >> 6)) field = (uint32_t)((x[j] ^ qw[j]) >> (bp & 63)) & smask;
        const uint32_t d = __popc(field);
        bool reach = true;
        if (p.flags & VC_FLAG_REF_SIGNEXT_KEYS) reach = ((field >> (s - 1)) & 1u) == 0;
        if (tt != t && reach && (d < dt || (d == dt && tt < t))) hit = false;
      }
    }
    const uint64_t km = __ballot(hit);
    if (km == 0) return;
    // staged in the wave's LDS buffer as (table, distance, entry position); ids and ring space are fetched per flush
    if (hit) stage[fill + (uint32_t)__popcll(km & ((1ull << lane) - 1ull))] = ((uint64_t)t << 48) | ((uint64_t)dist << 32) | pos;
    fill += (uint32_t)__popcll(km);
    if (fill >= MS_STAGE - VC_WAVE) flush();
  };
  // 16-byte loads -- two consecutive entries per lane, 1 KB per wave-instruction -- wherever every column of the copy keeps
  // even positions 16-byte aligned (one-word codes, or an even entry count); the copies are padded by one entry so that the
  // pair that holds the table's last entry stays inside the allocation
  const bool pairs = W == 1 || (p.n & 1ull) == 0;
  // (r04: the wave's buckets as ONE stream of 128-entry segments -- no half-empty second round per 1 526-entry bucket, 4 % idle
  // lane slots instead of 25 % -- was built and is 23 % SLOWER, 0.563 against 0.435 ms per launch on one box: the idle lanes'
  // clamped loads hit the cache and cost next to nothing, while a cursor that crosses bucket boundaries puts LDS reads and
  // scalar bookkeeping between the loads of a round.)
  for (uint32_t b = wave; b < nloc; b += blockDim.x / VC_WAVE) {
    const uint32_t off = s_off[b], len = s_len[b], t = s_meta[b] & 0xFFu, dt = s_meta[b] >> 8;
    if (len == 0) continue;
    const uint64_t* bc = p.tables[t].bcodes;
    if (pairs) {
      const uint32_t a0 = off & ~1u, end = off + len;
      for (uint32_t e0 = a0; e0 < end; e0 += VC_WAVE * MS_UP * 2) {
        vc_u64x2 v[MS_UP][W];
#pragma unroll
        for (uint32_t u = 0; u < MS_UP; ++u) {
          const uint32_t pa = e0 + (u * VC_WAVE + lane) * 2;
          const uint32_t pc = pa < end ? pa : a0;           // clamp: the bucket's first pair exists
#pragma unroll
          for (int j = 0; j < W; ++j) v[u][j] = __builtin_nontemporal_load(reinterpret_cast<const vc_u64x2*>(bc + (uint64_t)j * p.n + pc));
        }
#pragma unroll
        for (uint32_t u = 0; u < MS_UP; ++u) {
          const uint32_t pa = e0 + (u * VC_WAVE + lane) * 2;
          uint64_t xa[W], xb[W];
#pragma unroll
          for (int j = 0; j < W; ++j) {
            xa[j] = v[u][j].x;
            xb[j] = v[u][j].y;
          }
          entry(xa, pa >= off && pa < end, pa, t, dt);
          entry(xb, pa + 1 < end, pa + 1, t, dt);              // (pa + 1 > off always)
        }
      }
      continue;
    }
    for (uint32_t e0 = 0; e0 < len; e0 += VC_WAVE * MS_U) {
      uint64_t x[MS_U][W];
#pragma unroll
      for (uint32_t u = 0; u < MS_U; ++u) {
        const uint32_t e = e0 + u * VC_WAVE + lane;
        const uint32_t pos = off + (e < len ? e : 0u);   // clamp: entry 0 exists
#pragma unroll
        for (int j = 0; j < W; ++j) x[u][j] = __builtin_nontemporal_load(bc + (uint64_t)j * p.n + pos);
      }
#pragma unroll
      for (uint32_t u = 0; u < MS_U; ++u) {
        const uint32_t e = e0 + u * VC_WAVE + lane;
        entry(x[u], e < len, off + e, t, dt);
      }
    }
  }
  if (fill) flush();
  if (p.trace) {
    __syncthreads();
    if (tid == 0) p.trace[3ull * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
  }
}


// =============================================================================================================
// Cost-model switch of the exact k-NN loop: kernels (see VcMihScanFallback, vc_mih.hpp).
//
// The radius loop of search_worker.cc:159-218 stops after shell r iff k items have been seen and the k-th best distance
// K_r among them is <= mult * (r + 1) (:201-205).  An item is SEEN in shell r iff its minimum substring distance is
// <= r; every item with full distance < m (r + 1) has been seen by then (pigeonhole).  With D = the k-th smallest
// distance of the whole database (what the scan finds) and mult = min(m, 4):
//   mult < m:  the loop stops at the first r with mult (r + 1) >= D -- all items at or below D are seen by then, so the
//              result is the scan's top-k and radius = ceil(D / mult) - 1 (0 for D = 0);
//   mult = m:  r0 = D / m.  If m does not divide D (or D = 0) the loop stops at r0 with the scan's top-k.  If D = m r0 > 0
//              it may stop one shell EARLY: after shell r0 - 1 every item below D has been seen, and of the items AT D all
//              but those whose m substrings sit at exactly r0 each; if these make up k, radius = r0 - 1 and the result is
//              the k best among them, else radius = r0 and the result is the scan's top-k.
// The scan's candidate ring holds every item at or below D (the filter accepts d <= tau and tau never drops below D), so
// the ties are examined from the ring.  n_sub_reads = every leaf of shells 0..radius (no bitmap attached on this path).
// =============================================================================================================
__global__ void __launch_bounds__(256) mih_gather_queries_kernel(const uint64_t* __restrict__ q, const uint32_t* __restrict__ list,
                                                                 uint32_t n, uint32_t W, uint64_t* __restrict__ out) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n * W; e += gridDim.x * blockDim.x)
    out[e] = q[(uint64_t)list[e / W] * W + e % W];
}

// heavy queries after the in-kernel shells: those whose predicted remaining probes (work[slot][3], written at hand-over)
// exceed `limit` go to list_a (answered by the verify kernel), the others to list_b (radius loop continues)
__global__ void __launch_bounds__(256) mih_partition_kernel(const uint32_t* __restrict__ list, uint32_t n,
                                                            const unsigned long long* __restrict__ work, unsigned long long limit,
                                                            uint32_t* __restrict__ list_a, uint32_t* __restrict__ list_b, uint32_t* __restrict__ ctr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t slot = list[i];
  if (work[slot * 4 + 3] > limit) list_a[atomicAdd(&ctr[0], 1u)] = slot;
  else list_b[atomicAdd(&ctr[1], 1u)] = slot;
}

#define MR_TIES 8192u
__global__ void __launch_bounds__(256) mih_replay_kernel(const VcMihReplayArgs a) {
  __shared__ uint64_t s_tie[MR_TIES];
  __shared__ uint32_t s_ntie, s_below;
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const uint32_t slot = a.list[q];
  const uint32_t raw = a.lin_count[(size_t)q * a.lin_qs], nres = a.rows_cnt[q];
  const uint64_t* row = a.rows + (uint64_t)q * a.k;
  const uint32_t S = a.sbits, m = a.m;
  auto unresolved = [&]() {
    if (tid == 0) {
      a.unresolved[atomicAdd(a.n_unresolved, 1u)] = slot;
      a.resolved_flag[q] = 0;
    }
  };
  if (raw > a.lin_cap || nres == 0xFFFFFFFFu) { unresolved(); return; }
  uint32_t r_stop = 0, B = 0;
  bool early = false;
  if (tid == 0) { s_ntie = 0; s_below = 0; }
  __syncthreads();
  if (nres < a.k) {
    r_stop = S;                                    // fewer than k items: the loop runs to its last shell (search_worker.cc:170)
  } else {
    const uint32_t D = (uint32_t)(row[a.k - 1] >> 32);
    if (a.stop_mult < m) {
      r_stop = D == 0 ? 0u : (D + a.stop_mult - 1) / a.stop_mult - 1;
    } else {
      const uint32_t r0 = D / m;
      r_stop = r0;
      if (D != 0 && D % m == 0) {
        uint32_t mine = 0;
        for (uint32_t i = tid; i < a.k; i += blockDim.x) mine += (uint32_t)(row[i] >> 32) < D;
        if (mine) atomicAdd(&s_below, mine);
        // ties at D that shell r0 - 1 has seen: some substring below r0 (the substrings sum to m r0, so not all equal r0)
        const uint64_t* ring = a.lin_ring + (uint64_t)q * a.lin_cap;
        const uint32_t smask = S == 32 ? 0xFFFFFFFFu : ((1u << S) - 1u);
        for (uint32_t e = tid; e < raw; e += blockDim.x) {
          const uint64_t v = ring[e];
          if ((uint32_t)(v >> 32) != D) continue;
          const uint32_t local = (uint32_t)v - a.id_base;
          bool all_eq = true;
          for (uint32_t t = 0; t < m; ++t) {
            const uint32_t bp = t * S;
            const uint64_t x = a.cols[(uint64_t)(bp >> 6) * a.stride + local] ^ a.queries[(uint64_t)q * a.W + (bp >> 6)];
            all_eq = all_eq && (uint32_t)__popc((uint32_t)(x >> (bp & 63)) & smask) == r0;
          }
          if (!all_eq) {
            const uint32_t at = atomicAdd(&s_ntie, 1u);
            if (at < MR_TIES) s_tie[at] = v;
          }
        }
        __syncthreads();
        B = s_below;
        const uint32_t nt = s_ntie;
        if (nt > MR_TIES) { unresolved(); return; }   // (block-uniform)
        if (B + nt >= a.k) {
          early = true;
          r_stop = r0 - 1;
          uint32_t P = 2;
          while (P < nt) P <<= 1;
          for (uint32_t i = nt + tid; i < P; i += blockDim.x) s_tie[i] = VC_PACK_INF;
          vc_bitonic_lds(s_tie, P, blockDim.x);
        }
      }
    }
  }
  uint64_t* out = a.tgt.ring + (uint64_t)slot * a.tgt.cap;
  for (uint32_t i = tid; i < nres; i += blockDim.x) out[i] = (early && i >= B) ? s_tie[i - B] : row[i];
  if (tid == 0) {
    a.tgt.count[slot] = nres;
    a.tgt.radius[slot] = r_stop;
    unsigned long long leaves = 0;
    for (uint32_t r = 0; r <= r_stop; ++r) leaves += c_binom[S][r];
    a.tgt.sub[slot] = leaves;
    a.tgt.loc[slot] = 0;
    a.tgt.seen[slot] = 0;
    a.resolved_flag[q] = 1;
  }
}

// distinct items the radius loop would have verified up to shell radius[slot]: minimum substring distance <= radius.
// Up to 8 queries per pass over the database; counters in registers, one atomic per block and query at the end.
#define MC_Q 8
template <int W>
__global__ void __launch_bounds__(256) mih_minsub_count_kernel(const uint64_t* __restrict__ cols, uint64_t stride, uint64_t n, uint32_t m,
                                                               uint32_t sbits, const uint64_t* __restrict__ queries,
                                                               const uint32_t* __restrict__ list, const uint32_t* __restrict__ flag,
                                                               uint32_t nq, const uint32_t* __restrict__ radius,
                                                               unsigned long long* __restrict__ seen) {
  __shared__ uint64_t sq[MC_Q][W];
  __shared__ uint32_t srad[MC_Q];
  for (uint32_t i = threadIdx.x; i < nq * W; i += blockDim.x) sq[i / W][i % W] = queries[i];
  for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) srad[i] = flag[i] ? radius[list[i]] : 0xFFFFFFFFu;   // unsettled: counts nothing useful
  __syncthreads();
  const uint32_t smask = sbits == 32 ? 0xFFFFFFFFu : ((1u << sbits) - 1u);
  unsigned long long cnt[MC_Q];
#pragma unroll
  for (int q = 0; q < MC_Q; ++q) cnt[q] = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t c[W];
#pragma unroll
    for (int j = 0; j < W; ++j) c[j] = __builtin_nontemporal_load(cols + (uint64_t)j * stride + i);
#pragma unroll
    for (int q = 0; q < MC_Q; ++q) {
      if ((uint32_t)q >= nq) break;
      uint32_t ms = 0xFFFFFFFFu;
      for (uint32_t t = 0; t < m; ++t) {
        const uint32_t bp = t * sbits;
        uint32_t field = 0;
#pragma unroll
        for (int j = 0; j < W; ++j)
          if ((uint32_t)j == (bp >> 6)) field = (uint32_t)((c[j] ^ sq[q][j]) >> (bp & 63)) & smask;
        ms = min(ms, (uint32_t)__popc(field));
      }
      cnt[q] += ms <= srad[q] && srad[q] != 0xFFFFFFFFu;
    }
  }
#pragma unroll
  for (int q = 0; q < MC_Q; ++q) {
    if ((uint32_t)q >= nq) break;
    unsigned long long v = cnt[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, VC_WAVE);
    if (vc_lane() == 0 && v && flag[q]) atomicAdd(&seen[list[q]], v);
  }
}


// sum the per-query work counters of one mih_query_kernel launch into the index-wide totals (vc_get_timing); one block
// of 1024 threads: four 32-byte records per thread in flight -- one memory round trip for a tile of 4096 queries (the
// serial form of this loop, 256 threads and one record per iteration, took 16 us behind every launch)
__device__ __forceinline__ void mih_work_reduce_block(const unsigned long long* __restrict__ work, uint32_t nq,
                                                      unsigned long long* __restrict__ totals) {
  __shared__ unsigned long long s_t[3];
  if (threadIdx.x < 3) s_t[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long a = 0, b = 0, c = 0;
  const ulonglong2* w2 = reinterpret_cast<const ulonglong2*>(work);
  for (uint32_t base = 0; base < nq; base += 4 * blockDim.x) {
    ulonglong2 v0[4], v1[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t i = base + u * blockDim.x + threadIdx.x;
      v0[u] = i < nq ? w2[(size_t)i * 2] : make_ulonglong2(0, 0);
      v1[u] = i < nq ? w2[(size_t)i * 2 + 1] : make_ulonglong2(0, 0);
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      a += v0[u].x;
      b += v0[u].y;
      c += v1[u].x;
    }
  }
  for (uint32_t o = VC_WAVE / 2; o; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
    c += __shfl_xor(c, o);
  }
  if (vc_lane() == 0) {
    atomicAdd(&s_t[0], a);
    atomicAdd(&s_t[1], b);
    atomicAdd(&s_t[2], c);
  }
  __syncthreads();
  if (threadIdx.x < 3) totals[threadIdx.x] += s_t[threadIdx.x];
  if (threadIdx.x == 3) totals[3] += nq;
}

__global__ void __launch_bounds__(1024) mih_work_reduce_kernel(const unsigned long long* __restrict__ work, uint32_t nq,
                                                               unsigned long long* __restrict__ totals, uint32_t* __restrict__ ctr,
                                                               volatile uint32_t* __restrict__ host_ctr, uint32_t seq) {
  // the launch's counters (unfinished queries, where the others stopped) go straight to pinned host memory, followed by the
  // launch's sequence number: the host polls that word instead of queueing a copy and waiting for the stream (every query
  // kernel block has exited when this kernel runs, so the counters are final)
  // (block 1 of the two: the system-scope fence takes ~4 us, as long as the summation block 0 does meanwhile)
  if (blockIdx.x == 1) {
    if (host_ctr && threadIdx.x == 0) {
      for (uint32_t i = 2; i < 8; ++i) {
        host_ctr[i] = ctr[i];
        ctr[i] = 0;                        // ready for the next launch: no memset in front of it
      }
      __threadfence_system();
      host_ctr[8] = seq;
    }
    return;
  }
  mih_work_reduce_block(work, nq, totals);
}

// per-query result segments of the radius search -> one contiguous result array, ascending per query (positions beyond
// out_cap are dropped; the caller learns the needed size from the offsets).  One 1024-thread block per query; a segment the
// query kernel already sorted (flag) is copied, an unsorted one is ordered on the way:
//   n <= 8192 : bitonic network in LDS, written straight to its place in `out`;
//   larger    : the same network on the ring segment itself (global memory), padded to a power of two with VC_PACK_INF
//               behind the entries (cap is a power of two), then copied.
// (one launch where the sort and the copy-out used to be two)
__global__ void __launch_bounds__(1024) vc_sort_compact_segments_kernel(uint64_t* __restrict__ ring, uint32_t cap,
                                                                        const uint32_t* __restrict__ count,
                                                                        const uint32_t* __restrict__ sorted_flag,
                                                                        const uint64_t* __restrict__ offs,
                                                                        uint64_t* __restrict__ out, uint64_t out_cap) {
  __shared__ uint64_t a[VC_SORT_CAP];
  const uint32_t q = blockIdx.x;
  const uint32_t n = min(count[q], cap);          // (a segment that outgrew the ring is repeated with a larger one)
  const uint64_t lo = offs[q];
  uint64_t* seg = ring + (uint64_t)q * cap;
  const bool sorted = n < 2 || count[q] > cap || (sorted_flag && sorted_flag[q]);
  uint32_t P = 2;
  while (P < n) P <<= 1;
  if (!sorted && P <= VC_SORT_CAP) {
    for (uint32_t i = threadIdx.x; i < P; i += 1024) a[i] = i < n ? seg[i] : VC_PACK_INF;
    vc_bitonic_lds(a, P, 1024);
    for (uint32_t i = threadIdx.x; i < n; i += 1024)
      if (lo + i < out_cap) out[lo + i] = a[i];
    return;
  }
  if (!sorted) {
    for (uint32_t i = n + threadIdx.x; i < P; i += 1024) seg[i] = VC_PACK_INF;
    vc_bitonic_lds(seg, P, 1024);
  }
  for (uint32_t i = threadIdx.x; i < n; i += 1024)
    if (lo + i < out_cap) out[lo + i] = seg[i];
}

// exclusive prefix of one tile's segment lengths behind the running total of the call:
// offsets[i] = tot[0] + sum_{j<i} count[j];  tot[0] += sum;  tot[1] = max(tot[1], max count)  (overflow detection)
__global__ void __launch_bounds__(1024) vc_radius_offsets_kernel(const uint32_t* __restrict__ count, uint32_t nq,
                                                                 uint64_t* __restrict__ offsets, unsigned long long* tot, uint32_t first,
                                                                 volatile unsigned long long* host_tot, unsigned long long seq,
                                                                 const unsigned long long* __restrict__ work,
                                                                 unsigned long long* __restrict__ totals) {
  if (blockIdx.x == 1) {   // second block (launched when `work` is given): the query kernel's work counters (vc_get_timing)
    mih_work_reduce_block(work, nq, totals);
    return;
  }
  __shared__ uint64_t s_w[1024 / VC_WAVE];
  __shared__ uint32_t s_max;
  const uint32_t lane = vc_lane(), wave = threadIdx.x / VC_WAVE;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  constexpr uint32_t PER = MIH_RADIUS_TILE / 1024;   // consecutive queries per thread
  uint32_t c[PER], mine = 0, mx = 0;
#pragma unroll
  for (uint32_t i = 0; i < PER; ++i) {
    const uint32_t q = threadIdx.x * PER + i;
    c[i] = q < nq ? count[q] : 0u;
    mine += c[i];                                     // (cap <= 2^26: a wave's 64 x PER counts stay below 2^32)
    mx = max(mx, c[i]);
  }
  uint32_t wtot;
  uint32_t ex = vc_wave_excl_scan(mine, wtot);
  if (lane == 0) s_w[wave] = wtot;
  atomicMax(&s_max, mx);
  __syncthreads();
  const uint64_t tot0 = first ? 0ull : tot[0], max0 = first ? 0ull : tot[1];   // (first tile of a call: the totals start over)
  uint64_t base = tot0, total = 0;
  for (uint32_t w = 0; w < 1024 / VC_WAVE; ++w) {
    if (w < wave) base += s_w[w];
    total += s_w[w];
  }
  uint64_t run = base + ex;
#pragma unroll
  for (uint32_t i = 0; i < PER; ++i) {
    const uint32_t q = threadIdx.x * PER + i;
    if (q < nq) offsets[q] = run;
    run += c[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    offsets[nq] = tot0 + total;
    tot[0] = tot0 + total;
    tot[1] = s_max > max0 ? (unsigned long long)s_max : max0;
    if (host_tot && seq) {   // last tile of a call: total and largest segment go to mapped host memory, the call's sequence number last
      host_tot[0] = tot[0];
      host_tot[1] = tot[1];
      __threadfence_system();
      host_tot[2] = seq;
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// per-query result segments (ring[q * cap ..], count[q] entries each, cap a power of two) -> ascending per query at out[offs[q] ..]
// (vc_sharded_search_radius: the shards' concatenated results of a query are ordered by the same kernel the radius search uses)
hipError_t vc_launch_sort_compact_segments(uint64_t* d_ring, uint32_t cap, const uint32_t* d_count, const uint64_t* d_offs,
                                           uint64_t* d_out, uint64_t out_cap, uint32_t nq, hipStream_t s) {
  if (nq == 0) return hipSuccess;
  hipLaunchKernelGGL(vc_sort_compact_segments_kernel, dim3(nq), dim3(1024), 0, s, d_ring, cap, d_count, (const uint32_t*)nullptr, d_offs,
                     d_out, out_cap);
  return hipGetLastError();
}

hipError_t vc_launch_gather_queries(const uint64_t* d_q, const uint32_t* d_list, uint32_t n, uint32_t W, uint64_t* d_out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(mih_gather_queries_kernel, dim3((n * W + 255) / 256), dim3(256), 0, s, d_q, d_list, n, W, d_out);
  return hipGetLastError();
}

hipError_t vc_launch_mih_replay(const VcMihReplayArgs& a, hipStream_t s) {
  if (a.gq == 0) return hipSuccess;
  hipLaunchKernelGGL(mih_replay_kernel, dim3(a.gq), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t vc_launch_minsub_count(const uint64_t* cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t m, uint32_t sbits,
                                  const uint64_t* d_queries, const uint32_t* d_list, const uint32_t* d_flag, uint32_t nq,
                                  const uint32_t* d_radius, unsigned long long* d_seen, uint32_t n_cu, hipStream_t s) {
  for (uint32_t q0 = 0; q0 < nq; q0 += MC_Q) {
    const uint32_t qt = std::min<uint32_t>(MC_Q, nq - q0);
    const dim3 grid((uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)n_cu * 8));
#define MC_CASE(W_) case W_: hipLaunchKernelGGL(mih_minsub_count_kernel<W_>, grid, dim3(256), 0, s, cols, stride, n, m, sbits, d_queries + (size_t)q0 * W, d_list + q0, d_flag + q0, qt, d_radius, d_seen); break;
    switch (W) {
      MC_CASE(1) MC_CASE(2) MC_CASE(4) MC_CASE(8)
      default: return hipErrorInvalidValue;
    }
#undef MC_CASE
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) return r;
  }
  return hipSuccess;
}

// Host wait for a word a kernel publishes to mapped host memory (sequence number last).  The first ~30 us spin hot --
// that is where the win over hipStreamSynchronize lies (~10 us per call) --, after that every probe is followed by a
// pause instruction and, beyond 200 us, a sched_yield: a long launch (uniform queries, 1e9 shells) no longer burns a core,
// and the G threads of the sharded host leg do not starve each other on a cgroup-limited host.  Gives up after 2 ms
// (the caller then waits with hipStreamSynchronize): ADVICE round 3.
template <class T>
static bool poll_mapped_word(volatile T* flag, T expect) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spins = 0;; ++spins) {
    if (*flag == expect) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return true;
    }
    if ((spins & 63u) != 63u) continue;
    const auto dt = std::chrono::steady_clock::now() - t0;
    if (dt > std::chrono::milliseconds(2)) return false;
    if (dt > std::chrono::microseconds(200)) sched_yield();
    else if (dt > std::chrono::microseconds(30))
      for (int i = 0; i < 32; ++i) __builtin_ia32_pause();
  }
}

struct VcMihIndex {
  uint32_t W = 0, m = 0, sbits = 0, id_base = 0, flags = 0, n_cu = 0, cap = 0;
  uint64_t n = 0;
  VcKnobs knobs;   // environment knobs of the owning engine (read at vc_create)
  std::vector<VcTableView> h_tables;
  VcTableView* d_tables = nullptr;
  std::vector<void*> allocs;
  // search tile buffers
  uint32_t tile_k = 0, tile_cap = 0;
  void* d_tile = nullptr;
  size_t tile_bytes = 0;
  uint32_t* d_lists = nullptr;   // 4 slot lists of qtile_max entries + counters + the launch order
  uint32_t* h_ctr = nullptr;     // pinned: the counters the host reads back after every launch sequence
  uint32_t* h_ctr_dev = nullptr; // its device-side alias (mapped): the query kernel's counters are stored there directly
  uint32_t ctr_seq = 0;          // sequence number of the last counter publication (h_ctr[8] when it has landed)
  // measurement (vc_get_timing): event pairs around every mih_query_kernel launch, device totals of its work counters
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  uint32_t launch_tick = 0, launches_all = 0;   // every knobs.timing_every-th launch is timed; all are counted
  unsigned long long* d_totals = nullptr;   // probes | non-empty buckets | entries verified | queries
  unsigned long long* d_stotals = nullptr;  // the stream kernel's share of the same, [MS_TOT_LINES][16]
  uint64_t* d_ring = nullptr;               // [slots of the tile][cap] candidate rings of the multi-block shells (lazy)
  size_t ring_entries = 0;
  size_t lds_per_block = 65536;             // hipDeviceProp.sharedMemPerBlock of the index's device
  uint32_t group_hint = 2;                  // shells grouped into the query kernel's first pass (adapts to where queries stop)
};

#define MIH_CHECK(call)                                                                                  \
  do {                                                                                                   \
    hipError_t _r = (call);                                                                              \
    if (_r != hipSuccess) {                                                                              \
      if (err) *err = std::string(#call) + ": " + hipGetErrorString(_r);                                 \
      return _r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP;                                      \
    }                                                                                                    \
  } while (0)

static size_t query_kernel_lds(uint32_t buf_entries, uint32_t m, uint32_t sbits, uint32_t W, uint32_t lo);
static uint32_t grid_for(uint64_t n, uint32_t n_cu) { return (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)n_cu * 16); }

static size_t device_lds_per_block() {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 65536;
  return prop.sharedMemPerBlock;
}

static bool g_binom_ready[16] = {};

static int upload_binom(std::string* err) {
  int dev = 0;
  MIH_CHECK(hipGetDevice(&dev));
  if (dev < 16 && g_binom_ready[dev]) return VC_OK;
  static uint32_t b[33][33];
  for (int n = 0; n <= 32; ++n)
    for (int k = 0; k <= 32; ++k) b[n][k] = k > n ? 0u : (k == 0 || k == n) ? 1u : b[n - 1][k - 1] + b[n - 1][k];
  MIH_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_binom), b, sizeof b));
  if (dev < 16) g_binom_ready[dev] = true;
  return VC_OK;
}

void vc_mih_free(VcMihIndex* ix) {
  if (!ix) return;
  for (void* p : ix->allocs) (void)hipFree(p);
  (void)hipFree(ix->d_tables);
  (void)hipFree(ix->d_tile);
  (void)hipFree(ix->d_lists);
  if (ix->h_ctr) (void)hipHostFree(ix->h_ctr);
  for (auto& pr : ix->ev_pool) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  (void)hipFree(ix->d_totals);
  (void)hipFree(ix->d_stotals);
  (void)hipFree(ix->d_ring);
  delete ix;
}

// directory lines (VcTableView::lines): 2 GB per 32-bit table, built while that is a small part of what is still free once the
// records of the index itself (ids, offsets, bitmaps, directories, {id, code} records) have their room (VC_MIH_LINES=0/1 overrides)
#define MIH_NLINES (1u << 25)
static bool want_dir_lines(uint32_t sbits, uint32_t m, uint64_t n, uint32_t W, bool want_bent, const VcKnobs& knobs) {
  if (sbits != 32 || n == 0) return false;
  if (knobs.mih_lines >= 0) return knobs.mih_lines != 0;
  // Where most 256-key blocks hold single-entry buckets only (93 % at 1e8 codes) the block directory answers a hit in one round
  // trip too and the lines gain nothing (r04, same box: 13.7 vs 13.6 M queries/s at 1e8, 9.96 -> 10.25 M at 1e9): build them from
  // the size on where a block's buckets are rarely all single (n / 2^32 = 0.07: a third of the blocks)
  if (n < 300000000ull) return false;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  const size_t index_bytes = (size_t)m * (n * 8 + (1ull << 29) + (3ull << 26)) + (want_bent ? (size_t)m * n * 16 * W : 0);
  return free_b > index_bytes && (size_t)m * MIH_NLINES * 64 <= (free_b - index_bytes) / 4;
}

int vc_mih_build(VcMihIndex** out, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t m,
                 uint32_t sbits, uint32_t id_base, uint32_t flags, uint32_t n_cu, uint32_t cand_cap, const VcKnobs& knobs,
                 hipStream_t s, std::string* err) {
  if (sbits != 8 && sbits != 16 && sbits != 32) {
    if (err) *err = "substring width must be 8, 16 or 32 bits (bits / n_tables)";
    return VC_ERR_INVALID;
  }
  int rc = upload_binom(err);
  if (rc) return rc;
  VcMihIndex* ix = new VcMihIndex();
  ix->W = W; ix->m = m; ix->sbits = sbits; ix->id_base = id_base; ix->flags = flags; ix->n_cu = n_cu; ix->cap = cand_cap; ix->n = n;
  ix->knobs = knobs;
  ix->lds_per_block = device_lds_per_block();
  ix->h_tables.resize(m);
  auto fail_free = [&](int code) { vc_mih_free(ix); return code; };
  auto dalloc = [&](void** p, size_t bytes, bool keep) -> hipError_t {
    hipError_t r = hipMalloc(p, std::max<size_t>(bytes, 256));
    if (r == hipSuccess && keep) ix->allocs.push_back(*p);
    return r;
  };

  // bucket-order code copies for the tables whose buckets are big (see VcTableView::bcodes): m more copies of the
  // codes, so only while they fit comfortably (dev knob VC_MIH_BCODES=0/1 overrides)
  bool want_bcodes = sbits <= 16;
  {
    size_t free_b = 0, total_b = 0;
    if (want_bcodes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)m * n * W * 8 > free_b / 3) want_bcodes = false;
    if (knobs.mih_bcodes >= 0) want_bcodes = knobs.mih_bcodes != 0;   // dev knob VC_MIH_BCODES
  }
  bool want_bent = sbits == 32 && W <= 2;   // (VcTableView::bent; VC_MIH_BENT=0/1 overrides)
  {
    size_t free_b = 0, total_b = 0;
    if (want_bent && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)m * n * 16 * W > free_b / 100 * 55) want_bent = false;   // (55 % of the free memory: 128 GB of records at 1e9 x 128 bit next to 50 GB of codes + index on a 288 GB part)
    if (knobs.mih_bent >= 0) want_bent = knobs.mih_bent != 0 && sbits == 32 && W <= 2;
  }
  const bool want_lines = want_dir_lines(sbits, m, n, W, want_bent, knobs);
  const uint64_t nkeyspace = 1ull << sbits;
  const uint64_t bm_words = std::max<uint64_t>(nkeyspace / 32, 8);
  const uint32_t mask = sbits == 32 ? 0xFFFFFFFFu : (uint32_t)(nkeyspace - 1);
  const uint64_t nn = std::max<uint64_t>(n, 1);

  uint32_t *k_in = nullptr, *k_out = nullptr, *v_in = nullptr;
  uint32_t* d_temp = nullptr;      // work area of the sort and of the scans (vc_sort.hip)
  uint32_t* d_scan_in = nullptr;   // counts (direct) or blockpop (ranked)
#define B_CHECK(call)                                                                     \
  do {                                                                                    \
    hipError_t _r = (call);                                                               \
    if (_r != hipSuccess) {                                                               \
      if (err) *err = std::string(#call) + ": " + hipGetErrorString(_r);                  \
      (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(v_in); (void)hipFree(d_temp); (void)hipFree(d_scan_in); \
      return fail_free(_r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP);            \
    }                                                                                     \
  } while (0)

  B_CHECK(dalloc((void**)&k_in, nn * 4, false));
  B_CHECK(dalloc((void**)&k_out, nn * 4, false));
  B_CHECK(dalloc((void**)&v_in, nn * 4, false));
  const uint64_t scan_n = sbits == 32 ? (1ull << 24) : nkeyspace + 1;
  const size_t temp_words = std::max(vc_radix_sort_work_words(n), vc_scan_work_words(scan_n));
  B_CHECK(dalloc((void**)&d_temp, temp_words * 4, false));
  B_CHECK(dalloc((void**)&d_scan_in, (scan_n + 1) * 4, false));
  const uint32_t passes = vc_radix_sort_passes(sbits);

  for (uint32_t t = 0; t < m; ++t) {
    uint32_t *ids = nullptr, *bitmap = nullptr, *offsets = nullptr, *blockrank = nullptr;
    B_CHECK(dalloc((void**)&ids, nn * 4, true));
    B_CHECK(dalloc((void**)&bitmap, bm_words * 4, true));
    B_CHECK(hipMemsetAsync(bitmap, 0, bm_words * 4, s));
    const uint32_t bitpos = t * sbits;
    if (n) {
      // stable LSD radix sort by key (vc_sort.hip): ids stay ascending inside a bucket = append order of
      // build_hash_tables.cc:54-63.  The passes ping-pong; the buffers are ordered so that the last pass lands in
      // (k_out, ids).
      uint32_t* kb[2] = {k_in, k_out};
      uint32_t* vb[2] = {v_in, ids};
      if ((passes & 1u) == 0) { std::swap(kb[0], kb[1]); std::swap(vb[0], vb[1]); }
      hipLaunchKernelGGL(mih_keys_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, d_cols + (uint64_t)(bitpos >> 6) * stride, n,
                         bitpos & 63, mask, kb[0], vb[0]);
      B_CHECK(hipGetLastError());
      B_CHECK(vc_radix_sort_pairs(kb, vb, n, sbits, d_temp, s));   // result in pair (passes & 1) = (k_out, ids)
    }
    VcTableView tv{};
    if (sbits < 32) {
      B_CHECK(dalloc((void**)&offsets, (nkeyspace + 1) * 4, true));
      B_CHECK(hipMemsetAsync(d_scan_in, 0, (nkeyspace + 1) * 4, s));
      if (n) {
        hipLaunchKernelGGL(mih_runs_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, k_out, n, bitmap, d_scan_in);
        B_CHECK(hipGetLastError());
      }
      B_CHECK(vc_exclusive_scan_u32(d_scan_in, offsets, nkeyspace + 1, d_temp, s));
      tv.n_unique = 0;
    } else {
      if (n) {
        hipLaunchKernelGGL(mih_runs_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, k_out, n, bitmap, (uint32_t*)nullptr);
        B_CHECK(hipGetLastError());
      }
      const uint32_t nblocks = 1u << 24;
      B_CHECK(dalloc((void**)&blockrank, (size_t)nblocks * 4, true));
      hipLaunchKernelGGL(mih_blockpop_kernel, dim3(n_cu * 16), dim3(256), 0, s, bitmap, nblocks, d_scan_in);
      B_CHECK(hipGetLastError());
      B_CHECK(vc_exclusive_scan_u32(d_scan_in, blockrank, nblocks, d_temp, s));
      uint32_t last_rank = 0, last_pop = 0;
      B_CHECK(hipMemcpyAsync(&last_rank, blockrank + nblocks - 1, 4, hipMemcpyDeviceToHost, s));
      B_CHECK(hipMemcpyAsync(&last_pop, d_scan_in + nblocks - 1, 4, hipMemcpyDeviceToHost, s));
      B_CHECK(hipStreamSynchronize(s));
      tv.n_unique = last_rank + last_pop;
      B_CHECK(dalloc((void**)&offsets, ((size_t)tv.n_unique + 1) * 4, true));
      hipLaunchKernelGGL(mih_ranked_offsets_kernel, dim3(grid_for(nn, n_cu)), dim3(256), 0, s, k_out, n, bitmap, blockrank,
                         offsets, tv.n_unique);
      B_CHECK(hipGetLastError());
      uint2* blockoff = nullptr;
      B_CHECK(dalloc((void**)&blockoff, ((size_t)nblocks + 1) * 8, true));
      hipLaunchKernelGGL(mih_blockoff_kernel, dim3(n_cu * 16), dim3(256), 0, s, blockrank, offsets, nblocks, tv.n_unique, blockoff);
      B_CHECK(hipGetLastError());
      tv.blockoff = blockoff;
      if (want_lines) {
        uint4* lines = nullptr;
        B_CHECK(dalloc((void**)&lines, (size_t)MIH_NLINES * 64, true));
        hipLaunchKernelGGL(mih_lines_kernel, dim3(n_cu * 16), dim3(256), 0, s, bitmap, blockrank, offsets, MIH_NLINES, lines);
        B_CHECK(hipGetLastError());
        tv.lines = lines;
      }
    }
    tv.offsets = offsets;
    tv.ids = ids;
    tv.bitmap = bitmap;
    tv.blockrank = blockrank;
    tv.bcodes = nullptr;
    if (want_bcodes && n) {
      uint64_t* bc = nullptr;
      B_CHECK(dalloc((void**)&bc, (size_t)n * W * 8 + 16, true));   // (+ one entry: mih_bucket_stream_kernel reads 16-byte pairs)
      hipLaunchKernelGGL(mih_bcodes_kernel, dim3(grid_for(n * W, n_cu)), dim3(256), 0, s, d_cols, stride, W, ids, n, bc);
      B_CHECK(hipGetLastError());
      tv.bcodes = bc;
    }
    tv.bent = nullptr;
    if (want_bent && n) {
      uint4* be = nullptr;
      B_CHECK(dalloc((void**)&be, (size_t)n * 16 * W, true));
      hipLaunchKernelGGL(mih_bent_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, d_cols, stride, W, ids, n, be);
      B_CHECK(hipGetLastError());
      tv.bent = be;
    }
    ix->h_tables[t] = tv;
  }
  B_CHECK(hipMalloc((void**)&ix->d_tables, sizeof(VcTableView) * m));
  B_CHECK(hipMemcpyAsync(ix->d_tables, ix->h_tables.data(), sizeof(VcTableView) * m, hipMemcpyHostToDevice, s));
  B_CHECK(hipStreamSynchronize(s));
  (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(v_in); (void)hipFree(d_temp); (void)hipFree(d_scan_in);
#undef B_CHECK
  *out = ix;
  return VC_OK;
}

// ---- BaseProxy-style views (tests / adapters; not on the search path) --------------------------------
static uint32_t popc32(uint32_t x) { return (uint32_t)__builtin_popcount(x); }

// reference-style key -> internal (masked) key, or false when no bucket can carry that index
static bool internal_key(const VcMihIndex* ix, uint32_t index, uint32_t* key) {
  if (ix->sbits == 32) { *key = index; return true; }
  const uint32_t mask = (1u << ix->sbits) - 1u;
  const uint32_t low = index & mask;
  if (ix->flags & VC_FLAG_REF_SIGNEXT_KEYS) {   // binaryToInt leaves the sign extension of the top byte in the key
    const uint32_t ext = (low >> (ix->sbits - 1)) ? (low | ~mask) : low;
    if (ext != index) return false;
  } else if (index != low) {
    return false;
  }
  *key = low;
  return true;
}

static int bucket_range(VcMihIndex* ix, uint32_t table, uint32_t key, uint32_t* off, uint32_t* len, int* bit,
                        hipStream_t s, std::string* err) {
  const VcTableView& tv = ix->h_tables[table];
  uint32_t word = 0;
  MIH_CHECK(hipMemcpyAsync(&word, tv.bitmap + (key >> 5), 4, hipMemcpyDeviceToHost, s));
  MIH_CHECK(hipStreamSynchronize(s));
  *bit = (word >> (key & 31)) & 1u;
  *off = 0;
  *len = 0;
  if (!*bit) return VC_OK;
  uint32_t idx = key;
  if (ix->sbits == 32) {
    uint32_t blk[8], base = 0;
    MIH_CHECK(hipMemcpyAsync(blk, tv.bitmap + ((uint64_t)(key >> 8) << 3), 32, hipMemcpyDeviceToHost, s));
    MIH_CHECK(hipMemcpyAsync(&base, tv.blockrank + (key >> 8), 4, hipMemcpyDeviceToHost, s));
    MIH_CHECK(hipStreamSynchronize(s));
    const uint32_t w = (key >> 5) & 7u;
    for (uint32_t i = 0; i < w; ++i) base += popc32(blk[i]);
    base += popc32(blk[w] & ((1u << (key & 31)) - 1u));
    idx = base;
  }
  uint32_t ab[2];
  MIH_CHECK(hipMemcpyAsync(ab, tv.offsets + idx, 8, hipMemcpyDeviceToHost, s));
  MIH_CHECK(hipStreamSynchronize(s));
  *off = ab[0];
  *len = ab[1] - ab[0];
  return VC_OK;
}

int vc_mih_bucket(VcMihIndex* ix, uint32_t table, uint32_t index, std::vector<uint32_t>* local_ids, hipStream_t s,
                  std::string* err) {
  local_ids->clear();
  uint32_t key;
  if (!internal_key(ix, index, &key)) return VC_OK;
  uint32_t off, len;
  int bit;
  int rc = bucket_range(ix, table, key, &off, &len, &bit, s, err);
  if (rc) return rc;
  local_ids->resize(len);
  if (len) {
    MIH_CHECK(hipMemcpyAsync(local_ids->data(), ix->h_tables[table].ids + off, (size_t)len * 4, hipMemcpyDeviceToHost, s));
    MIH_CHECK(hipStreamSynchronize(s));
  }
  return VC_OK;
}

int vc_mih_bitmap_test(VcMihIndex* ix, uint32_t table, uint32_t index, int* bit, hipStream_t s, std::string* err) {
  uint32_t key;
  *bit = 0;
  if (!internal_key(ix, index, &key)) return VC_OK;
  uint32_t off, len;
  return bucket_range(ix, table, key, &off, &len, bit, s, err);
}

int vc_mih_bitmap_read(VcMihIndex* ix, uint32_t table, uint64_t word_off, uint64_t n_words, uint32_t* out, hipStream_t s,
                       std::string* err) {
  const uint64_t words = (1ull << ix->sbits) / 32;
  if (word_off + n_words > words) {
    if (err) *err = "bitmap read out of range";
    return VC_ERR_INVALID;
  }
  MIH_CHECK(hipMemcpyAsync(out, ix->h_tables[table].bitmap + word_off, n_words * 4, hipMemcpyDeviceToHost, s));
  MIH_CHECK(hipStreamSynchronize(s));
  return VC_OK;
}

// ---- index persistence ------------------------------------------------------------------
// File = header + per table {n_unique, offsets length, ids[n], offsets[], bitmap[2^s/32], blockrank[2^24] (s == 32)}.
// The bitmap section is byte for byte what generate_bitmap.cc:122-125 writes for that table; the bucket lists
// (build_hash_tables.cc:36-64) are the contiguous id runs delimited by offsets[].
struct VcIndexHeader {
  char magic[8];
  uint32_t version, bits, m, sbits, id_base, reserved;
  uint64_t n;
  uint64_t cols_checksum;   // digest of the code columns the index was built from (mih_checksum_kernel)
  uint64_t file_bytes;      // whole file, header included: a truncated or padded file is refused
};
#define VC_INDEX_VERSION 2u

static int cols_checksum(const uint64_t* d_cols, uint64_t stride, uint32_t W, uint64_t n, uint32_t n_cu, hipStream_t s,
                         uint64_t* out, std::string* err) {
  unsigned long long* d_sum = nullptr;
  MIH_CHECK(hipMalloc((void**)&d_sum, 8));
  hipError_t r = hipMemsetAsync(d_sum, 0, 8, s);
  if (r == hipSuccess && n) {
    hipLaunchKernelGGL(mih_checksum_kernel, dim3(grid_for(n * W, n_cu)), dim3(256), 0, s, d_cols, stride, W, n, d_sum);
    r = hipGetLastError();
  }
  unsigned long long h = 0;
  if (r == hipSuccess) r = hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s);
  if (r == hipSuccess) r = hipStreamSynchronize(s);
  (void)hipFree(d_sum);
  if (r != hipSuccess) {
    if (err) *err = std::string("index checksum: ") + hipGetErrorString(r);
    return VC_ERR_HIP;
  }
  *out = h;
  return VC_OK;
}
static const char kIndexMagic[8] = {'V', 'C', 'M', 'I', 'H', 'I', 'D', 'X'};

static int copy_out(FILE* fh, const void* d_src, size_t bytes, hipStream_t s, std::vector<char>& buf, std::string* err) {
  for (size_t off = 0; off < bytes; off += buf.size()) {
    const size_t cnt = std::min(buf.size(), bytes - off);
    MIH_CHECK(hipMemcpyAsync(buf.data(), (const char*)d_src + off, cnt, hipMemcpyDeviceToHost, s));
    MIH_CHECK(hipStreamSynchronize(s));
    if (fwrite(buf.data(), 1, cnt, fh) != cnt) {
      if (err) *err = "short write to the index file";
      return VC_ERR_INVALID;
    }
  }
  return VC_OK;
}

static int copy_in(FILE* fh, void* d_dst, size_t bytes, hipStream_t s, std::vector<char>& buf, std::string* err) {
  for (size_t off = 0; off < bytes; off += buf.size()) {
    const size_t cnt = std::min(buf.size(), bytes - off);
    if (fread(buf.data(), 1, cnt, fh) != cnt) {
      if (err) *err = "index file is truncated";
      return VC_ERR_INVALID;
    }
    MIH_CHECK(hipMemcpyAsync((char*)d_dst + off, buf.data(), cnt, hipMemcpyHostToDevice, s));
    MIH_CHECK(hipStreamSynchronize(s));
  }
  return VC_OK;
}

int vc_mih_save(VcMihIndex* ix, const uint64_t* d_cols, uint64_t stride, const char* path, hipStream_t s, std::string* err) {
  uint64_t digest = 0;
  int rc0 = cols_checksum(d_cols, stride, ix->W, ix->n, ix->n_cu, s, &digest, err);
  if (rc0) return rc0;
  FILE* fh = fopen(path, "wb");
  if (!fh) {
    if (err) *err = std::string("Can't create file ") + path + ".";
    return VC_ERR_INVALID;
  }
  VcIndexHeader h{};
  memcpy(h.magic, kIndexMagic, 8);
  h.version = VC_INDEX_VERSION; h.bits = ix->W * 64; h.m = ix->m; h.sbits = ix->sbits; h.id_base = ix->id_base; h.n = ix->n;
  h.cols_checksum = digest;
  const uint64_t bm_words = std::max<uint64_t>((1ull << ix->sbits) / 32, 8);
  h.file_bytes = sizeof h;
  for (uint32_t t = 0; t < ix->m; ++t) {
    const uint64_t off_len = ix->sbits == 32 ? (uint64_t)ix->h_tables[t].n_unique + 1 : (1ull << ix->sbits) + 1;
    h.file_bytes += 16 + (ix->n + off_len + bm_words + (ix->sbits == 32 ? (1ull << 24) : 0)) * 4;
  }
  int rc = fwrite(&h, sizeof h, 1, fh) == 1 ? VC_OK : VC_ERR_INVALID;
  std::vector<char> buf(32u << 20);
  for (uint32_t t = 0; t < ix->m && rc == VC_OK; ++t) {
    const VcTableView& tv = ix->h_tables[t];
    const uint64_t off_len = ix->sbits == 32 ? (uint64_t)tv.n_unique + 1 : (1ull << ix->sbits) + 1;
    const uint64_t th[2] = {tv.n_unique, off_len};
    if (fwrite(th, sizeof th, 1, fh) != 1) rc = VC_ERR_INVALID;
    if (rc == VC_OK) rc = copy_out(fh, tv.ids, (size_t)ix->n * 4, s, buf, err);
    if (rc == VC_OK) rc = copy_out(fh, tv.offsets, (size_t)off_len * 4, s, buf, err);
    if (rc == VC_OK) rc = copy_out(fh, tv.bitmap, (size_t)bm_words * 4, s, buf, err);
    if (rc == VC_OK && ix->sbits == 32) rc = copy_out(fh, tv.blockrank, (size_t)(1u << 24) * 4, s, buf, err);
  }
  if (fclose(fh) != 0 && rc == VC_OK) rc = VC_ERR_INVALID;
  if (rc == VC_ERR_INVALID && err && err->empty()) *err = "short write to the index file";
  return rc;
}

int vc_mih_load(VcMihIndex** out, const char* path, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t m,
                uint32_t sbits, uint32_t id_base, uint32_t flags, uint32_t n_cu, uint32_t cand_cap, const VcKnobs& knobs,
                hipStream_t s, std::string* err) {
  FILE* fh = fopen(path, "rb");
  if (!fh) {
    if (err) *err = std::string("Can't open file ") + path + ".";
    return VC_ERR_INVALID;
  }
  VcIndexHeader h{};
  if (fread(&h, sizeof h, 1, fh) != 1 || memcmp(h.magic, kIndexMagic, 8) != 0 || h.version != VC_INDEX_VERSION) {
    fclose(fh);
    if (err) *err = "not a verticut_gpu index file (bad magic or version)";
    return VC_ERR_INVALID;
  }
  if (h.bits != W * 64 || h.m != m || h.sbits != sbits || h.n != n || h.id_base != id_base) {
    fclose(fh);
    if (err) *err = "index file was built for another database shape (bits / n_tables / records / id_base differ)";
    return VC_ERR_STATE;
  }
  {   // the file must be exactly as long as its header says (truncated copies, appended garbage)
    const long at = ftell(fh);
    uint64_t size = 0;
    if (at < 0 || fseek(fh, 0, SEEK_END) != 0) size = 0; else size = (uint64_t)ftell(fh);
    if (size != h.file_bytes || fseek(fh, at, SEEK_SET) != 0) {
      fclose(fh);
      if (err) *err = "index file is truncated or has trailing bytes (size differs from its header)";
      return VC_ERR_INVALID;
    }
  }
  int rc = upload_binom(err);
  if (rc) { fclose(fh); return rc; }
  {   // the index must have been built from THESE records, not merely from a database of the same shape
    uint64_t digest = 0;
    if ((rc = cols_checksum(d_cols, stride, W, n, n_cu, s, &digest, err))) { fclose(fh); return rc; }
    if (digest != h.cols_checksum) {
      fclose(fh);
      if (err) *err = "index file was built from other records than the resident ones (code checksum differs)";
      return VC_ERR_STATE;
    }
  }
  VcMihIndex* ix = new VcMihIndex();
  ix->W = W; ix->m = m; ix->sbits = sbits; ix->id_base = id_base; ix->flags = flags; ix->n_cu = n_cu; ix->cap = cand_cap; ix->n = n;
  ix->knobs = knobs;
  ix->lds_per_block = device_lds_per_block();
  ix->h_tables.resize(m);
  std::vector<char> buf(32u << 20);
  const uint64_t bm_words = std::max<uint64_t>((1ull << sbits) / 32, 8);
  bool want_bcodes = sbits <= 16;
  {
    size_t free_b = 0, total_b = 0;
    if (want_bcodes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)m * n * W * 8 > free_b / 3) want_bcodes = false;
    if (knobs.mih_bcodes >= 0) want_bcodes = knobs.mih_bcodes != 0;
  }
  bool want_bent = sbits == 32 && W <= 2;
  {
    size_t free_b = 0, total_b = 0;
    if (want_bent && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)m * n * 16 * W > free_b / 100 * 55) want_bent = false;   // (55 % of the free memory: 128 GB of records at 1e9 x 128 bit next to 50 GB of codes + index on a 288 GB part)
    if (knobs.mih_bent >= 0) want_bent = knobs.mih_bent != 0 && sbits == 32 && W <= 2;
  }
  const bool want_lines = want_dir_lines(sbits, m, n, W, want_bent, knobs);
  auto dalloc = [&](void** p, size_t bytes) -> int {
    hipError_t r = hipMalloc(p, std::max<size_t>(bytes, 256));
    if (r != hipSuccess) {
      if (err) *err = std::string("index load: ") + hipGetErrorString(r);
      return r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP;
    }
    ix->allocs.push_back(*p);
    return VC_OK;
  };
  for (uint32_t t = 0; t < m && rc == VC_OK; ++t) {
    uint64_t th[2];
    if (fread(th, sizeof th, 1, fh) != 1) { rc = VC_ERR_INVALID; if (err) *err = "index file is truncated"; break; }
    const uint64_t expect = sbits == 32 ? th[0] + 1 : (1ull << sbits) + 1;
    if (th[1] != expect || th[0] > n) { rc = VC_ERR_INVALID; if (err) *err = "index file is corrupt (table directory)"; break; }
    VcTableView tv{};
    uint32_t *ids = nullptr, *offsets = nullptr, *bitmap = nullptr, *blockrank = nullptr;
    if ((rc = dalloc((void**)&ids, (size_t)std::max<uint64_t>(n, 1) * 4))) break;
    if ((rc = dalloc((void**)&offsets, (size_t)th[1] * 4))) break;
    if ((rc = dalloc((void**)&bitmap, (size_t)bm_words * 4))) break;
    if ((rc = copy_in(fh, ids, (size_t)n * 4, s, buf, err))) break;
    if ((rc = copy_in(fh, offsets, (size_t)th[1] * 4, s, buf, err))) break;
    if ((rc = copy_in(fh, bitmap, (size_t)bm_words * 4, s, buf, err))) break;
    if (sbits == 32) {
      if ((rc = dalloc((void**)&blockrank, (size_t)(1u << 24) * 4))) break;
      if ((rc = copy_in(fh, blockrank, (size_t)(1u << 24) * 4, s, buf, err))) break;
    }
    tv.ids = ids; tv.offsets = offsets; tv.bitmap = bitmap; tv.blockrank = blockrank; tv.n_unique = (uint32_t)th[0];
    {   // device validation pass: see the kernels above.  Order matters: offsets and ranks first, entries last (the
        // entry check dereferences offsets[rank(key)], which is only safe to trust once... it is bounds-checked anyway)
      uint32_t *d_bad = nullptr, *d_seen = nullptr, *d_pop = nullptr, *d_exp = nullptr, *d_work = nullptr;
      const uint32_t nblocks = 1u << 24;
      hipError_t r = hipMalloc((void**)&d_bad, 4);
      if (r == hipSuccess) r = hipMemsetAsync(d_bad, 0, 4, s);
      if (r == hipSuccess) r = hipMalloc((void**)&d_seen, (size_t)((n + 31) / 32 + 1) * 4);
      if (r == hipSuccess) r = hipMemsetAsync(d_seen, 0, (size_t)((n + 31) / 32 + 1) * 4, s);
      if (r == hipSuccess) {
        hipLaunchKernelGGL(mih_validate_offsets_kernel, dim3(grid_for(th[1], n_cu)), dim3(256), 0, s, offsets, th[1], n, d_bad);
        r = hipGetLastError();
      }
      if (r == hipSuccess && sbits == 32) {
        r = hipMalloc((void**)&d_pop, (size_t)(nblocks + 1) * 4);
        if (r == hipSuccess) r = hipMalloc((void**)&d_exp, (size_t)nblocks * 4);
        if (r == hipSuccess) r = hipMalloc((void**)&d_work, std::max<size_t>(vc_scan_work_words(nblocks), 64) * 4);
        if (r == hipSuccess) {
          hipLaunchKernelGGL(mih_blockpop_kernel, dim3(n_cu * 16), dim3(256), 0, s, bitmap, nblocks, d_pop);
          r = hipGetLastError();
        }
        if (r == hipSuccess) r = vc_exclusive_scan_u32(d_pop, d_exp, nblocks, d_work, s);
        if (r == hipSuccess) {
          hipLaunchKernelGGL(mih_validate_rank_kernel, dim3(n_cu * 16), dim3(256), 0, s, blockrank, d_exp, d_pop, nblocks, tv.n_unique, d_bad);
          r = hipGetLastError();
        }
      }
      uint32_t bad = 0;
      if (r == hipSuccess) r = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s);
      if (r == hipSuccess) r = hipStreamSynchronize(s);
      if (r == hipSuccess && bad == 0 && n) {   // directory sound: now every entry against the resident records
        const uint32_t bitpos = t * sbits;
        hipLaunchKernelGGL(mih_validate_entries_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, tv, d_cols + (uint64_t)(bitpos >> 6) * stride, n,
                           bitpos & 63, sbits == 32 ? 0xFFFFFFFFu : (uint32_t)((1ull << sbits) - 1), sbits, d_seen, d_bad);
        r = hipGetLastError();
        if (r == hipSuccess) r = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s);
        if (r == hipSuccess) r = hipStreamSynchronize(s);
      }
      (void)hipFree(d_bad); (void)hipFree(d_seen); (void)hipFree(d_pop); (void)hipFree(d_exp); (void)hipFree(d_work);
      if (r != hipSuccess) {
        if (err) *err = std::string("index validation: ") + hipGetErrorString(r);
        rc = r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP;
        break;
      }
      if (bad) {
        char msg[160];
        snprintf(msg, sizeof msg, "index file is corrupt: table %u fails validation (%s%s%s%s%s)", t, bad & MIH_BAD_OFFSETS ? "offsets " : "",
                 bad & MIH_BAD_RANK ? "rank-directory " : "", bad & MIH_BAD_ID ? "id-range " : "", bad & MIH_BAD_DUP ? "duplicate-ids " : "",
                 bad & MIH_BAD_BUCKET ? "bucket-membership " : "");
        if (err) *err = msg;
        rc = VC_ERR_STATE;
        break;
      }
    }
    if (sbits == 32) {   // derived directory, not in the file (see VcTableView::blockoff)
      uint2* blockoff = nullptr;
      if ((rc = dalloc((void**)&blockoff, ((size_t)(1u << 24) + 1) * 8))) break;
      hipLaunchKernelGGL(mih_blockoff_kernel, dim3(n_cu * 16), dim3(256), 0, s, blockrank, offsets, 1u << 24, tv.n_unique, blockoff);
      tv.blockoff = blockoff;
      if (want_lines) {
        uint4* lines = nullptr;
        if ((rc = dalloc((void**)&lines, (size_t)MIH_NLINES * 64))) break;
        hipLaunchKernelGGL(mih_lines_kernel, dim3(n_cu * 16), dim3(256), 0, s, bitmap, blockrank, offsets, MIH_NLINES, lines);
        tv.lines = lines;
      }
    }
    if (want_bcodes && n) {
      uint64_t* bc = nullptr;
      if ((rc = dalloc((void**)&bc, (size_t)n * W * 8 + 16))) break;
      hipLaunchKernelGGL(mih_bcodes_kernel, dim3(grid_for(n * W, n_cu)), dim3(256), 0, s, d_cols, stride, W, ids, n, bc);
      tv.bcodes = bc;
    }
    if (want_bent && n) {
      uint4* be = nullptr;
      if ((rc = dalloc((void**)&be, (size_t)n * 16 * W))) break;
      hipLaunchKernelGGL(mih_bent_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, d_cols, stride, W, ids, n, be);
      tv.bent = be;
    }
    ix->h_tables[t] = tv;
  }
  fclose(fh);
  if (rc == VC_OK) {
    hipError_t r = hipMalloc((void**)&ix->d_tables, sizeof(VcTableView) * m);
    if (r == hipSuccess) r = hipMemcpyAsync(ix->d_tables, ix->h_tables.data(), sizeof(VcTableView) * m, hipMemcpyHostToDevice, s);
    if (r == hipSuccess) r = hipStreamSynchronize(s);
    if (r != hipSuccess) {
      if (err) *err = std::string("index load: ") + hipGetErrorString(r);
      rc = VC_ERR_HIP;
    }
  }
  if (rc != VC_OK) {
    vc_mih_free(ix);
    return rc;
  }
  *out = ix;
  return VC_OK;
}

// ---- search -------------------------------------------------------------------------------------------
// per-slot state of a search tile.  The candidate rings ([slot][cap], 2 GB at the default cap) are used only by the
// multi-block shells and are allocated the first time a query gets there (with_ring).
static uint32_t qtile_max(const VcMihIndex* ix) {
  const int v = ix->knobs.mih_qtile;
  return v > 0 ? std::min(std::max((uint32_t)v, MIH_QTILE_MIN), MIH_QTILE_LIMIT) : MIH_QTILE_MAX;
}
// `slots`: queries of the call's largest launch (the state is laid out for exactly that many, so one call passes one value)
static int ensure_tile(VcMihIndex* ix, uint32_t slots, uint32_t k, uint32_t cap, bool with_ring, MihState* st, std::string* err) {
  const size_t Q = std::max(slots, MIH_QTILE_MIN);
  size_t bytes = 0;
  auto take = [&](size_t b) { size_t o = bytes; bytes += (b + 255) & ~(size_t)255; return o; };
  const size_t o_thresh = take(Q * 8), o_count = take(Q * 4), o_prev = take(Q * 4),
               o_seen = take(Q * 8), o_sub = take(Q * 8), o_loc = take(Q * 8), o_radius = take(Q * 4),
               o_topk = take(Q * (size_t)k * 8), o_topn = take(Q * 4), o_work = take(Q * 4 * 8);
  if (bytes > ix->tile_bytes) {
    if (ix->d_tile) MIH_CHECK(hipFree(ix->d_tile));
    ix->d_tile = nullptr;
    ix->tile_bytes = 0;
    MIH_CHECK(hipMalloc(&ix->d_tile, bytes));
    ix->tile_bytes = bytes;
  }
  if (with_ring && ix->ring_entries < Q * cap) {
    if (ix->d_ring) MIH_CHECK(hipFree(ix->d_ring));
    ix->d_ring = nullptr;
    ix->ring_entries = 0;
    MIH_CHECK(hipMalloc((void**)&ix->d_ring, Q * cap * 8));
    ix->ring_entries = Q * cap;
  }
  if (!ix->d_lists) {   // 4 slot lists + 4 counters + 4 stop-shell counts (the counters start at zero; the reduce kernel re-zeroes what a launch counted)
    const size_t QM = qtile_max(ix);   // (laid out for the largest tile once: the counter block must not move between calls)
    MIH_CHECK(hipMalloc((void**)&ix->d_lists, (5 * QM + 16) * 4));   // ... + the launch order [QM] behind a 16-word counter block
    MIH_CHECK(hipMemset(ix->d_lists + 4 * QM, 0, 64));
  }
  uint8_t* b = (uint8_t*)ix->d_tile;
  st->thresh = (uint64_t*)(b + o_thresh);
  st->ring = ix->d_ring;
  st->count = (uint32_t*)(b + o_count);
  st->prev = (uint32_t*)(b + o_prev);
  st->seen = (unsigned long long*)(b + o_seen);
  st->sub = (unsigned long long*)(b + o_sub);
  st->loc = (unsigned long long*)(b + o_loc);
  st->radius = (uint32_t*)(b + o_radius);
  st->topk = (uint64_t*)(b + o_topk);
  st->topn = (uint32_t*)(b + o_topn);
  st->work = (unsigned long long*)(b + o_work);
  return VC_OK;
}

static hipError_t launch_probe(const ProbeParams& p, uint32_t W, uint32_t n_list, hipStream_t s) {
  const dim3 grid((p.nkeys + MIH_PCH - 1) / MIH_PCH, p.m_probe ? p.m_probe : p.m, n_list);
  switch (W) {
    case 1: hipLaunchKernelGGL(mih_probe_kernel<1>, grid, dim3(MIH_BLK), 0, s, p); break;
    case 2: hipLaunchKernelGGL(mih_probe_kernel<2>, grid, dim3(MIH_BLK), 0, s, p); break;
    case 4: hipLaunchKernelGGL(mih_probe_kernel<4>, grid, dim3(MIH_BLK), 0, s, p); break;
    case 8: hipLaunchKernelGGL(mih_probe_kernel<8>, grid, dim3(MIH_BLK), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static size_t query_kernel_lds(uint32_t buf_entries, uint32_t m, uint32_t sbits, uint32_t W = VC_MAX_W, uint32_t lo = MQ_LO_MAX) {
  return (size_t)buf_entries * 8 + (size_t)(3 * MQ_HMAX + 1) * 4 + (size_t)33 * MQ_BW * 4 +
         (sbits == 32 ? (size_t)m * (lo + 1) * ((1u << lo) / 32u) * 4 : 0) +                    // masks: [m][lo + 1] granules of 2^lo bits
         (size_t)MQ_MAX_GROUP * mq_hist_bins(W) * 4 + 16 +                                      // k-NN distance histograms, one per shell class
         (size_t)m * sizeof(VcTableView) + 16;                                                   // the tables' views
}

static hipError_t launch_query_kernel(const QueryKernelParams& p, uint32_t W, uint32_t nq, hipStream_t s) {
  const size_t lds = query_kernel_lds(p.buf_entries, p.m, p.sbits, W, p.mode == MQ_MODE_RADIUS ? MQ_LO_RADIUS : MQ_LO_KNN);
#define MQ_LAUNCH_K(K_) hipLaunchKernelGGL((K_), dim3(nq), dim3(MQ_BLK), lds, s, p)
#define MQ_LAUNCH(W_)                                                                                           \
  case W_:                                                                                                      \
    if (p.mode == MQ_MODE_RADIUS) MQ_LAUNCH_K((mih_query_kernel<W_, MQ_LO_RADIUS, false>));                     \
    else if (p.use_lines && MQ_LO_KNN == 7u) MQ_LAUNCH_K((mih_query_kernel<W_, MQ_LO_KNN, MQ_LO_KNN == 7u>));   \
    else MQ_LAUNCH_K((mih_query_kernel<W_, MQ_LO_KNN, false>));                                                 \
    break;
  switch (W) {
    MQ_LAUNCH(1)
    MQ_LAUNCH(2)
    MQ_LAUNCH(4)
    MQ_LAUNCH(8)
    default: return hipErrorInvalidValue;
  }
#undef MQ_LAUNCH
#undef MQ_LAUNCH_K
  return hipGetLastError();
}

// launch + measurement: events on the launch stream around the kernel, then the reduction of its work counters
static hipError_t timed_query_launch(VcMihIndex* ix, const QueryKernelParams& p_in, uint32_t W, uint32_t nq, hipStream_t s) {
  QueryKernelParams p = p_in;
  static unsigned long long* d_phase = nullptr;   // dev knob VC_MIH_PHASES: one buffer per process is enough
  if (ix->knobs.mih_phases && p.mode != MQ_MODE_RADIUS) {
    if (!d_phase && hipMalloc((void**)&d_phase, 1024) != hipSuccess) d_phase = nullptr;
    if (d_phase) {
      (void)hipMemsetAsync(d_phase, 0, 1024, s);
      (void)hipMemsetAsync(d_phase + 24, 0xFF, 8, s);
    }
    p.phase_dbg = d_phase;
  }
  if (!ix->d_totals) {
    hipError_t r = hipMalloc((void**)&ix->d_totals, 32);
    if (r == hipSuccess) r = hipMemsetAsync(ix->d_totals, 0, 32, s);
    if (r != hipSuccess) return r;
  }
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  ++ix->launches_all;
  if (ix->ev_used < 4096 && ix->launch_tick++ % std::max(ix->knobs.timing_every, 1u) == 0) {
    if (ix->ev_used == ix->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) == hipSuccess) {
        if (hipEventCreate(&b) == hipSuccess) ix->ev_pool.emplace_back(a, b); else (void)hipEventDestroy(a);
      }
    }
    if (ix->ev_used < ix->ev_pool.size()) ev = &ix->ev_pool[ix->ev_used++];
  }
  if (ev) (void)hipEventRecord(ev->first, s);
  // (approximate k-NN on the 512-bit-granule instantiation -- its deep shells are sector-bound like the radius search's -- measured
  // 10 % slower than on 128-bit granules, 0.81 vs 0.89 M queries/s: r04_sweeps.md)
  hipError_t r = launch_query_kernel(p, W, nq, s);
  if (ev) (void)hipEventRecord(ev->second, s);
  if (r != hipSuccess) return r;
  // (k-NN launches: heavy_ctr = the tile's counter block + 2.  Radius search has no counters to publish, and its work
  // counters are summed by vc_radius_offsets_kernel, which follows anyway)
  if (p.mode != MQ_MODE_RADIUS)
    hipLaunchKernelGGL(mih_work_reduce_kernel, dim3(2), dim3(1024), 0, s, p.st.work, nq, ix->d_totals,
                     p.heavy_ctr ? p.heavy_ctr - 2 : (uint32_t*)nullptr,
                     p.heavy_ctr ? (volatile uint32_t*)ix->h_ctr_dev : (volatile uint32_t*)nullptr, ++ix->ctr_seq);
  if (p.phase_dbg) {
    unsigned long long h[128];
    if (hipMemcpyAsync(h, p.phase_dbg, 1024, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
      fprintf(stderr, "[vc_mih lifetimes] stop shell: blocks, mean lifetime us, last end (us after the first block's start):");
      for (int c = 0; c < 6; ++c)
        if (h[16 + c]) fprintf(stderr, "  %s%d: %llu, %.1f, %.1f", c == 5 ? "handed over after " : "", c == 5 ? (int)p.r_last : c, h[16 + c], h[8 + c] * 0.01 / h[16 + c], (h[25 + c] - h[24]) * 0.01);
      fprintf(stderr, "\n");
      for (int c = 0; c < 6; ++c)
        if (h[16 + c]) {
          const unsigned long long* q = h + 32 + c * 8;
          const double nb = (double)h[16 + c];
          fprintf(stderr, "[vc_mih phases] %s%d (%llu blocks), us per block: set-up %.1f | plan+scan %.1f | directory %.1f | verify %.1f | evaluate %.1f | finish %.1f | out %.1f\n",
                  c == 5 ? "handed over after " : "stop shell ", c == 5 ? (int)p.r_last : c, h[16 + c], (q[7] + q[0]) * 0.01 / nb, q[1] * 0.01 / nb,
                  q[2] * 0.01 / nb, q[3] * 0.01 / nb, q[4] * 0.01 / nb, q[5] * 0.01 / nb, q[6] * 0.01 / nb);
        }
    }
    fprintf(stderr, "[vc_mih phases] %u blocks, us per block: set-up %.1f | plan+scan %.1f | directory %.1f | verify %.1f | evaluate %.1f | finish %.1f | out %.1f\n", nq,
              (h[7] + h[0]) * 0.01 / nq, h[1] * 0.01 / nq, h[2] * 0.01 / nq, h[3] * 0.01 / nq, h[4] * 0.01 / nq, h[5] * 0.01 / nq, h[6] * 0.01 / nq);
  }
  return hipGetLastError();
}

// launch + measurement of mih_bucket_stream_kernel (same records as the query kernels: vc_timing.mih_*)
// blocks of mih_bucket_stream_kernel<W> the chip holds at once
static uint32_t stream_resident_blocks(uint32_t W, uint32_t n_cu) {
  static uint32_t per_cu[9] = {};
  if (W > 8) return n_cu;
  if (!per_cu[W]) {
    int b = 0;
    hipError_t r = hipErrorInvalidValue;
    switch (W) {
      case 1: r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, mih_bucket_stream_kernel<1>, 256, 0); break;
      case 2: r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, mih_bucket_stream_kernel<2>, 256, 0); break;
      case 4: r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, mih_bucket_stream_kernel<4>, 256, 0); break;
      case 8: r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, mih_bucket_stream_kernel<8>, 256, 0); break;
    }
    if (r != hipSuccess) (void)hipGetLastError();
    per_cu[W] = (r == hipSuccess && b > 0) ? (uint32_t)b : 2u;
  }
  return per_cu[W] * n_cu;
}

static hipError_t timed_stream_launch(VcMihIndex* ix, StreamParams sp, uint32_t W, uint32_t nq, hipStream_t s) {
  if (!ix->d_stotals) {
    hipError_t r = hipMalloc((void**)&ix->d_stotals, MS_TOT_LINES * 128);
    if (r == hipSuccess) r = hipMemsetAsync(ix->d_stotals, 0, MS_TOT_LINES * 128, s);
    if (r != hipSuccess) return r;
  }
  sp.totals = ix->d_stotals;
  static const bool dev_trace = getenv("VC_STREAM_TRACE") != nullptr;   // dev: per-block start / look-up / end times on stderr
  const uint32_t nblocks = nq * sp.split;
  unsigned long long* d_trace = nullptr;
  if (dev_trace && hipMalloc((void**)&d_trace, (size_t)nblocks * 24) == hipSuccess) {
    (void)hipMemsetAsync(d_trace, 0, (size_t)nblocks * 24, s);
    sp.trace = d_trace;
  }
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  ++ix->launches_all;
  if (ix->ev_used < 4096 && ix->launch_tick++ % std::max(ix->knobs.timing_every, 1u) == 0) {
    if (ix->ev_used == ix->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) == hipSuccess) {
        if (hipEventCreate(&b) == hipSuccess) ix->ev_pool.emplace_back(a, b); else (void)hipEventDestroy(a);
      }
    }
    if (ix->ev_used < ix->ev_pool.size()) ev = &ix->ev_pool[ix->ev_used++];
  }
  if (ev) (void)hipEventRecord(ev->first, s);
  const dim3 grid(nq * sp.split);
  switch (W) {
    case 1: hipLaunchKernelGGL(mih_bucket_stream_kernel<1>, grid, dim3(256), 0, s, sp); break;
    case 2: hipLaunchKernelGGL(mih_bucket_stream_kernel<2>, grid, dim3(256), 0, s, sp); break;
    case 4: hipLaunchKernelGGL(mih_bucket_stream_kernel<4>, grid, dim3(256), 0, s, sp); break;
    case 8: hipLaunchKernelGGL(mih_bucket_stream_kernel<8>, grid, dim3(256), 0, s, sp); break;
    default: return hipErrorInvalidValue;
  }
  hipError_t r = hipGetLastError();
  if (ev) (void)hipEventRecord(ev->second, s);
  if (d_trace) {   // dev: when do the blocks start, finish their probe look-up and end?  (us after the first block's start)
    std::vector<unsigned long long> h((size_t)nblocks * 3);
    if (hipMemcpyAsync(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
      unsigned long long t0 = ~0ull;
      for (uint32_t b = 0; b < nblocks; ++b) t0 = std::min(t0, h[3 * b]);
      std::vector<double> st, lk, en, life;
      for (uint32_t b = 0; b < nblocks; ++b) {
        st.push_back((h[3 * b] - t0) * 0.01);
        lk.push_back((h[3 * b + 1] - h[3 * b]) * 0.01);
        en.push_back((h[3 * b + 2] - t0) * 0.01);
        life.push_back((h[3 * b + 2] - h[3 * b]) * 0.01);
      }
      auto pr = [&](const char* name, std::vector<double> v) {
        std::sort(v.begin(), v.end());
        auto q = [&](double f) { return v[(size_t)(f * (v.size() - 1))]; };
        fprintf(stderr, "[stream trace] %-10s min/p10/p50/p90/p99/max us: %.1f %.1f %.1f %.1f %.1f %.1f\n", name, q(0), q(.1), q(.5), q(.9), q(.99), q(1));
      };
      fprintf(stderr, "[stream trace] %u blocks (%u queries x %u parts)\n", nblocks, nq, sp.split);
      pr("start", st); pr("look-up", lk); pr("end", en); pr("lifetime", life);
      // blocks alive over time (20 samples of the launch's span)
      const double span = *std::max_element(en.begin(), en.end());
      fprintf(stderr, "[stream trace] blocks alive at 5%%..100%% of %.1f us:", span);
      for (int i = 1; i <= 20; ++i) {
        const double t = span * i / 20.0 - 1e-9;
        uint32_t alive = 0;
        for (uint32_t b = 0; b < nblocks; ++b) alive += st[b] <= t && en[b] > t;
        fprintf(stderr, " %u", alive);
      }
      fprintf(stderr, "\n");
    }
    (void)hipFree(d_trace);
  }
  return r;
}

void vc_mih_timing(VcMihIndex* ix, float* ms, uint32_t* launches, uint64_t totals[4], hipStream_t s) {
  *ms = 0;
  *launches = 0;
  totals[0] = totals[1] = totals[2] = totals[3] = 0;
  if (!ix) return;
  for (size_t i = 0; i < ix->ev_used; ++i) {
    float t = 0;
    if (hipEventSynchronize(ix->ev_pool[i].second) == hipSuccess && hipEventElapsedTime(&t, ix->ev_pool[i].first, ix->ev_pool[i].second) == hipSuccess) {
      *ms += t;
      ++*launches;
    }
  }
  ix->ev_used = 0;
  // launches that were not bracketed by events (timing_every > 1) are priced at the timed ones' average: the work counters
  // below cover ALL launches, and the callers divide them by this launch count
  if (*launches && ix->launches_all > *launches) {
    *ms = *ms / (float)*launches * (float)ix->launches_all;
    *launches = ix->launches_all;
  }
  ix->launches_all = 0;
  if (ix->d_totals) {
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(h, ix->d_totals, 32, hipMemcpyDeviceToHost, s) == hipSuccess && hipMemsetAsync(ix->d_totals, 0, 32, s) == hipSuccess &&
        hipStreamSynchronize(s) == hipSuccess)
      for (int i = 0; i < 4; ++i) totals[i] = h[i];
  }
  if (ix->d_stotals) {
    std::vector<unsigned long long> h((size_t)MS_TOT_LINES * 16);
    if (hipMemcpyAsync(h.data(), ix->d_stotals, MS_TOT_LINES * 128, hipMemcpyDeviceToHost, s) == hipSuccess &&
        hipMemsetAsync(ix->d_stotals, 0, MS_TOT_LINES * 128, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
      for (uint32_t l = 0; l < MS_TOT_LINES; ++l)
        for (int i = 0; i < 4; ++i) totals[i] += h[(size_t)l * 16 + i];
  }
}

static uint32_t binom_host(uint32_t n, uint32_t k) {
  uint64_t r = 1;
  for (uint32_t i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (uint32_t)r;
}

// last shell the one-block-per-query kernel runs: the shells beyond cost more than `budget` bucket probes per query,
// or more than MQ_ENTRY_BUDGET expected bucket entries (probes x average bucket size: one block verifies them
// serially, round after round; the multi-block kernels spread a shell of big buckets over the chip -- 64-bit codes,
// 1e8 items, 16-bit substrings: 1526 entries per bucket, 274 K vs 160 K queries/s at configs[1]), and go to the
// multi-block kernels (one launch sequence per shell)
#define MQ_ENTRY_BUDGET 65536.0
static uint32_t inblock_last_shell(uint32_t S, uint32_t m, uint64_t budget, uint32_t r_cap, double avg_bucket) {
  uint64_t tot = 0;
  uint32_t r_last = 0;
  for (uint32_t r = 0; r <= std::min(S, r_cap); ++r) {
    const uint64_t c = (uint64_t)m * binom_host(S, r);
    if (r > 0 && (tot + c > budget || (double)(tot + c) * avg_bucket > MQ_ENTRY_BUDGET)) break;
    tot += c;
    r_last = r;
  }
  return std::min(r_last, 16u);
}

#define MQ_KNN_BUDGET 600000ull        // probes per query run inside mih_query_kernel (32-bit substrings, m = 4: shells 0..4)
#define MQ_RADIUS_BUDGET 4000000ull

// SearchWorker::get_stat for callers that stay on the device (vc_search_knn_dev_stats): the tile's statistics arrays ->
// vc_query_stats records in device memory, n_results from the rows' counts.  One thread per query.
__global__ void mih_stats_export_kernel(MihState st, const uint32_t* __restrict__ cnt, uint32_t qt, vc_query_stats* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= qt) return;
  vc_query_stats o;
  o.radius = st.radius[i];
  o.n_results = cnt[i];
  o.n_main_reads = 0;
  o.n_sub_reads = st.sub[i];
  o.n_local_reads = st.loc[i];
  o.n_candidates = st.seen[i];
  out[i] = o;
}

// What the verify kernel charges for nq queries (the cost-model switch, vc_engine.hip mih_scan_fallback: passes of up to 32
// queries): a pass streams the shard at ~6 TB/s or, from ~10 queries on, runs at the xor / popcount issue rate -- 8.2 ms per 32
// queries over 1e9 x 128 bit, i.e. 1.3e-13 s per (query x 64-bit word) -- plus ~40 us of launches.
static double scan_cost_s(uint64_t n, uint32_t W, uint32_t nq) {
  auto pass = [&](uint32_t qt) { return std::max((double)n * W * 8 / 6e12, (double)qt * n * W * 1.3e-13) + 40e-6; };
  return (double)(nq / 32) * pass(32) + (nq % 32 ? pass(nq % 32) : 0.0);
}

int vc_mih_search(VcMihIndex* ix, const uint64_t* d_cols, uint64_t stride, uint64_t n, const uint64_t* d_q, uint32_t nq,
                  uint32_t k, bool approximate, uint64_t* d_out, uint32_t* d_cnt, vc_query_stats* host_stats, hipStream_t s,
                  std::string* err, const VcMihScanFallback* fb, vc_query_stats* d_stats) {
  const bool stats = host_stats != nullptr || d_stats != nullptr;   // the cost model prices the statistics pass either way
  if (n != ix->n) {
    if (err) *err = "index is stale: codes were added after vc_build_index()";
    return VC_ERR_STATE;
  }
  int rc = upload_binom(err);
  if (rc) return rc;
  const uint32_t cap = std::max(ix->cap, 4 * k);
  const uint32_t S = ix->sbits;
  // stop multiplier: the reference's literal 4 (search_worker.cc:204); min(m,4) keeps m < 4 exact
  const uint32_t stop_mult = (ix->flags & VC_FLAG_REF_STOP_LITERAL4) ? 4u : std::min(ix->m, 4u);
  MihState st;
  if (!ix->h_ctr) {
    MIH_CHECK(hipHostMalloc((void**)&ix->h_ctr, 64, hipHostMallocMapped));
    ix->h_ctr[8] = 0;
    if (hipHostGetDevicePointer((void**)&ix->h_ctr_dev, ix->h_ctr, 0) != hipSuccess) { (void)hipGetLastError(); ix->h_ctr_dev = nullptr; }
  }   // pageable memory makes the read-back a staged copy
  uint32_t* h_ctr = ix->h_ctr;

  // Shells 0..r_last run inside ONE launch, one block per query (mih_query_kernel); the host reads ONE counter per
  // tile (how many queries are not finished) and only those continue shell by shell through the multi-block kernels.
  uint32_t buf_entries = 1024;     // candidates of one block: up to k of every shell class of a grouped pass + one round of survivors
  while (buf_entries < MQ_MAX_GROUP * k + MQ_ROUND) buf_entries <<= 1;
  // the query kernel's LDS request (top-k + candidate buffer, hit lists, binomials, masks) must fit a workgroup of this
  // device (k = 3073..7168 asks for ~82 KB); otherwise every shell runs through the multi-block kernels
  const bool inblock = ix->knobs.mih_host_loop == 0 && buf_entries <= 8192 && ix->m <= 64 &&
                       query_kernel_lds(buf_entries, ix->m, S, ix->W, MQ_LO_KNN) <= ix->lds_per_block;
  // the call's queries in as few launches as the tile limit allows, equally filled (20 000 queries = 2 x 10 000, not 16 384 + 3 616)
  const uint32_t QM = qtile_max(ix);
  const uint32_t n_tiles = std::max(1u, (nq + QM - 1) / QM);
  const uint32_t tile = std::max(64u, std::min(QM, (((nq + n_tiles - 1) / n_tiles) + 63u) & ~63u));
  if ((rc = ensure_tile(ix, tile, k, cap, !inblock, &st, err))) return rc;
  uint32_t* lists[4] = {ix->d_lists, ix->d_lists + QM, ix->d_lists + 2 * QM, ix->d_lists + 3 * QM};
  uint32_t* d_ctr = ix->d_lists + 4 * QM;
  const double avg_bucket = (double)ix->n / (S >= 32 ? 4294967296.0 : (double)(1ull << S));
  // One block per query: a small batch leaves most of the chip idle while a few blocks walk a big shell, so the probe
  // budget per query shrinks with the batch (a lone query runs shells 0..2 in its block -- 2 116 probes at m = 4 -- and
  // the bigger ones through the multi-block kernels, which spread a shell over all CUs).
  uint64_t knn_budget = MQ_KNN_BUDGET;
  if (nq < 1024) knn_budget = std::max<uint64_t>(2500, MQ_KNN_BUDGET * nq / 1024);
  if (ix->knobs.mih_budget) knn_budget = ix->knobs.mih_budget;   // dev knob VC_MIH_BUDGET
  const uint32_t r_last = inblock_last_shell(S, ix->m, knn_budget, S, avg_bucket);
  const bool trace = ix->knobs.mih_trace;   // VC_MIH_TRACE: per-shell wall times on stderr
  // the scan switch reproduces the radius loop only where that loop is exact and its counters have a closed form
  const bool switch_ok = !approximate && ix->knobs.mih_switch != 0 && !(ix->flags & (VC_FLAG_USE_BITMAP | VC_FLAG_REF_SIGNEXT_KEYS)) &&
                         stop_mult == std::min(ix->m, 4u) && n >= 1;
  // Shells that share the FIRST pass of the query kernel (32-bit substrings): their candidates are tagged by shell and the
  // stop rule is evaluated shell by shell afterwards, so grouping changes no result and no statistic -- it saves a scan /
  // drain round per grouped shell and wastes the probes of the shells behind the one a query stops in.  The depth follows
  // the workload: the kernel counts where the queries of a launch stopped, and the next launch groups up to the shell
  // most of them needed (at most 3; VC_MIH_GROUP fixes it).
  uint32_t group = ix->knobs.mih_group > 0 ? (uint32_t)ix->knobs.mih_group : ix->group_hint;
  group = std::max(1u, std::min(group, std::min(3u, r_last + 1)));

  for (uint32_t q0 = 0; q0 < nq; q0 += tile) {
    const uint32_t qt = std::min(tile, nq - q0);
    uint32_t *cur = lists[0], *nxt = lists[1], *redo = lists[2];
    uint32_t n_cur = qt, r_start = 0, n_heavy = qt;
    if (inblock) {
      QueryKernelParams qp{};
      qp.cols = d_cols; qp.stride = stride; qp.n = ix->n; qp.tables = ix->d_tables; qp.queries = d_q + (size_t)q0 * ix->W;
      qp.st = st; qp.m = ix->m; qp.sbits = S; qp.id_base = ix->id_base; qp.flags = ix->flags; qp.cap = cap; qp.k = k;
      qp.mode = approximate ? MQ_MODE_APPROX : MQ_MODE_EXACT; qp.stop_mult = stop_mult; qp.r_last = r_last;
      qp.buf_entries = buf_entries; qp.heavy_list = cur; qp.heavy_ctr = d_ctr + 2;
      qp.out = d_out + (size_t)q0 * k; qp.out_cnt = d_cnt + q0; qp.group = group; qp.radius_hist = d_ctr + 4;
      qp.use_lines = (S == 32 && !ix->h_tables.empty() && ix->h_tables[0].lines) ? 1u : 0u;
      // (r04, same box: the query kernel 0.353 -> 0.319 ms per 4096 queries at 1e9, 0.277 -> 0.269 ms at 1e8, for a pre-pass of
      // ~3 us; VC_MIH_ORDER=0 switches it off)
      const bool want_order = ix->knobs.mih_order != 0;
      if (want_order && qt >= 4 * ix->n_cu * 2 && ix->m <= MO_BLK) {   // at least two residency waves of blocks: a launch order matters
        uint32_t* d_order = d_ctr + 16;
        const uint32_t per_block = MO_BLK / ix->m;
        hipLaunchKernelGGL(mih_order_kernel, dim3((qt + per_block - 1) / per_block), dim3(MO_BLK), 0, s, (const VcTableView*)ix->d_tables,
                           d_q + (size_t)q0 * ix->W, ix->W, ix->m, S, qt, d_order);
        MIH_CHECK(hipGetLastError());
        qp.order = d_order;
      }
      const auto t_q = std::chrono::steady_clock::now();
      if (!ix->h_ctr_dev) MIH_CHECK(hipMemsetAsync(d_ctr, 0, 32, s));   // (else zeroed at allocation and by every reduce kernel since)
      MIH_CHECK(timed_query_launch(ix, qp, ix->W, qt, s));
      // unfinished queries + where the others stopped: published by the reduce kernel into mapped host memory, sequence
      // number last; polling that word returns ~10 us before hipStreamSynchronize does (VC_MIH_POLL=0: plain synchronise)
      bool landed = false;
      if (ix->h_ctr_dev && ix->knobs.mih_poll) {
        landed = poll_mapped_word((volatile uint32_t*)(h_ctr + 8), (uint32_t)ix->ctr_seq);   // long launches (or a fault): wait properly
      }
      if (!ix->h_ctr_dev) MIH_CHECK(hipMemcpyAsync(h_ctr + 2, d_ctr + 2, 24, hipMemcpyDeviceToHost, s));
      if (!landed) MIH_CHECK(hipStreamSynchronize(s));
      n_heavy = n_cur = h_ctr[2];
      if (S == 32 && qt >= 64)       // next launch: group up to shell 2 when most queries of this one needed it
        ix->group_hint = (uint64_t)(h_ctr[6] + h_ctr[7]) * 10 >= (uint64_t)qt * 6 ? 3u : 2u;
      r_start = r_last + 1;
      if (trace)
        fprintf(stderr, "[vc_mih] shells 0..%u in the query kernel (group %u): %u queries, %u continue, stopped in shell 0/1/2/later %u/%u/%u/%u  %.1f us\n",
                r_last, group, qt, n_cur, h_ctr[4], h_ctr[5], h_ctr[6], h_ctr[7],
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_q).count());
      if (n_heavy) {
        MIH_CHECK(hipMemcpyAsync(lists[3], cur, (size_t)n_heavy * 4, hipMemcpyDeviceToDevice, s));
        if ((rc = ensure_tile(ix, tile, k, cap, true, &st, err))) return rc;   // the rings exist from the first hand-over on
        hipLaunchKernelGGL(mih_seed_ring_kernel, dim3(n_heavy), dim3(256), 0, s, st, (const uint32_t*)cur, k, cap);
        MIH_CHECK(hipGetLastError());
        if (fb && fb->fn && switch_ok) {
          // Cost model, per query: the hand-over left an upper bound of the probes each query may still need (its k-th
          // distance so far bounds its last shell).  The multi-block kernels sustain ~4e10 probes/s; the verify kernel's price per
          // query is scan_cost_s's (x 2.5 with the statistics pass).  Queries beyond the break-even are answered
          // by the scan NOW, stop rule replayed (mih_replay_kernel) -- the others keep their radius loop.
          const double limit = ix->knobs.mih_switch == 2 ? -1.0 : scan_cost_s(n, ix->W, 32) / 32 * (stats ? 2.5 : 1.0) * 4e10;
          MIH_CHECK(hipMemsetAsync(d_ctr, 0, 8, s));
          hipLaunchKernelGGL(mih_partition_kernel, dim3((n_heavy + 255) / 256), dim3(256), 0, s, (const uint32_t*)cur, n_heavy, st.work,
                             limit < 0 ? 0ull : (unsigned long long)limit, nxt, redo, d_ctr);
          MIH_CHECK(hipGetLastError());
          MIH_CHECK(hipMemcpyAsync(h_ctr, d_ctr, 8, hipMemcpyDeviceToHost, s));
          MIH_CHECK(hipStreamSynchronize(s));
          const uint32_t n_scan = h_ctr[0], n_keep = h_ctr[1];
          if (n_scan) {
            VcMihScanTarget tgt{st.ring, cap, st.count, st.radius, st.seen, st.sub, st.loc};
            MIH_CHECK(hipMemsetAsync(d_ctr, 0, 4, s));
            // unresolved queries (ring overflow, too many ties) rejoin the radius loop: appended behind the kept ones
            rc = fb->fn(fb->ctx, d_q + (size_t)q0 * ix->W, nxt, n_scan, k, stop_mult, tgt, stats, redo + n_keep, d_ctr, s);
            if (rc) {
              if (err) *err = "scan fallback of the exact k-NN loop failed";
              return rc;
            }
            MIH_CHECK(hipMemcpyAsync(h_ctr, d_ctr, 4, hipMemcpyDeviceToHost, s));
            MIH_CHECK(hipStreamSynchronize(s));
            if (trace) fprintf(stderr, "[vc_mih] cost model: %u of %u unfinished queries answered by the verify kernel, %u came back\n", n_scan, n_heavy, h_ctr[0]);
            n_cur = n_keep + h_ctr[0];
            std::swap(cur, redo);          // the radius loop goes on with the kept (+ unresolved) queries
          }
        }
      }
    } else {
      hipLaunchKernelGGL(mih_init_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, st, qt, cur, (uint64_t)VC_PACK_INF);
      MIH_CHECK(hipGetLastError());
    }
    bool switched = false;
    for (uint32_t r = r_start; r <= S && n_cur; ++r) {       // search_worker.cc:170: radius <= n_local_bytes_*8
      // Cost model: this shell alone is m * C(S, r) bucket probes per active query (the multi-block kernels sustain ~4e10
      // probes/s plus ~40 us of launches and a host round trip per shell); the verify kernel costs scan_cost_s (x 2.5 when the
      // statistics pass is wanted).  Beyond the break-even the remaining queries are
      // answered by the scan with the stop rule replayed -- same rows, same statistics (see mih_replay_kernel).
      if (fb && fb->fn && switch_ok && !switched) {
        const double est_mih = (double)n_cur * ix->m * binom_host(S, r) / 4e10 + 40e-6;
        const double est_scan = scan_cost_s(n, ix->W, n_cur) * (stats ? 2.5 : 1.0);
        if (est_mih > est_scan || ix->knobs.mih_switch == 2) {
          VcMihScanTarget tgt{st.ring, cap, st.count, st.radius, st.seen, st.sub, st.loc};
          MIH_CHECK(hipMemsetAsync(d_ctr, 0, 8, s));
          rc = fb->fn(fb->ctx, d_q + (size_t)q0 * ix->W, cur, n_cur, k, stop_mult, tgt, stats, nxt, d_ctr, s);
          if (rc) {
            if (err) *err = "scan fallback of the exact k-NN loop failed";
            return rc;
          }
          MIH_CHECK(hipMemcpyAsync(h_ctr, d_ctr, 4, hipMemcpyDeviceToHost, s));
          MIH_CHECK(hipStreamSynchronize(s));
          if (trace) fprintf(stderr, "[vc_mih] shell r=%u: %u queries answered by the verify kernel (cost model), %u continue\n", r, n_cur - h_ctr[0], h_ctr[0]);
          n_cur = h_ctr[0];
          std::swap(cur, nxt);
          switched = true;
          if (n_cur == 0) break;
        }
      }
      const auto t_shell = std::chrono::steady_clock::now();
      ProbeParams p{};
      p.cols = d_cols; p.stride = stride; p.tables = ix->d_tables; p.queries = d_q + (size_t)q0 * ix->W;
      p.st = st; p.r = r; p.nkeys = binom_host(S, r); p.m = ix->m; p.sbits = S; p.id_base = ix->id_base;
      p.flags = ix->flags; p.cap = cap; p.n = ix->n;
      CommitParams c{};
      c.st = st; c.next_list = nxt; c.redo_list = redo; c.ctr = d_ctr; c.k = k; c.cap = cap; c.r = r; c.sbits = S;
      c.stop_mult = stop_mult; c.approximate = approximate; c.last_shell = r == S;
      MIH_CHECK(hipMemsetAsync(d_ctr, 0, 8, s));
      const uint32_t* work = cur;
      uint32_t n_work = n_cur;
      h_ctr[0] = h_ctr[1] = 0;
      for (int round = 0;; ++round) {
        p.list = work;
        p.count_seen = round == 0;
        MIH_CHECK(launch_probe(p, ix->W, n_work, s));
        MIH_CHECK(vc_launch_select_ring_list(st.ring, cap, st.count, work, n_work, k, st.topk, st.topn, s));
        c.list = work;
        hipLaunchKernelGGL(mih_commit_kernel, dim3(n_work), dim3(VC_WAVE), 0, s, c);
        MIH_CHECK(hipGetLastError());
        MIH_CHECK(hipMemcpyAsync(h_ctr, d_ctr, 8, hipMemcpyDeviceToHost, s));
        MIH_CHECK(hipStreamSynchronize(s));
        if (h_ctr[1] == 0) break;
        // overflowed queries go round again with their tightened limit; swap the redo list with a scratch list
        if (round > 64) {
          if (err) *err = "MIH overflow recovery did not converge";
          return VC_ERR_CAPACITY;
        }
        // `cur` has been fully consumed by this round: reuse it as the work list of the recovery round so the
        // commit kernel can refill `redo`
        n_work = h_ctr[1];
        MIH_CHECK(hipMemcpyAsync(cur, redo, (size_t)n_work * 4, hipMemcpyDeviceToDevice, s));
        MIH_CHECK(hipMemsetAsync(d_ctr + 1, 0, 4, s));
        work = cur;
      }
      if (trace)
        fprintf(stderr, "[vc_mih] shell r=%u keys=%u active=%u -> next=%u  %.1f us\n", r, p.nkeys, n_cur, h_ctr[0],
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_shell).count());
      n_cur = h_ctr[0];
      std::swap(cur, nxt);
    }
    // rows of the queries that went through the multi-block shells (the query kernel wrote the others itself)
    if (n_heavy) {
      hipLaunchKernelGGL(mih_export_kernel, dim3(n_heavy), dim3(256), 0, s, st, inblock ? (const uint32_t*)lists[3] : (const uint32_t*)nullptr, k, cap,
                         d_out + (size_t)q0 * k, d_cnt + q0);
      MIH_CHECK(hipGetLastError());
    }
    if (d_stats) {
      hipLaunchKernelGGL(mih_stats_export_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, st, (const uint32_t*)(d_cnt + q0), qt, d_stats + q0);
      MIH_CHECK(hipGetLastError());
    }
    if (host_stats) {
      std::vector<unsigned long long> seen(qt), sub(qt), loc(qt);
      std::vector<uint32_t> rad(qt);
      MIH_CHECK(hipMemcpyAsync(seen.data(), st.seen, qt * 8, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipMemcpyAsync(sub.data(), st.sub, qt * 8, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipMemcpyAsync(loc.data(), st.loc, qt * 8, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipMemcpyAsync(rad.data(), st.radius, qt * 4, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipStreamSynchronize(s));
      for (uint32_t i = 0; i < qt; ++i) {
        vc_query_stats& o = host_stats[q0 + i];
        o.radius = rad[i];
        o.n_results = 0;
        o.n_main_reads = 0;
        o.n_sub_reads = sub[i];
        o.n_local_reads = loc[i];
        o.n_candidates = seen[i];
      }
    }
  }
  return VC_OK;
}

// ---- fixed-radius neighbour search (BASELINE config 2) ----------------------------------------------------
void vc_radius_work_free(VcRadiusWork* w) {
  if (!w) return;
  (void)hipFree(w->d_ring); (void)hipFree(w->d_compact); (void)hipFree(w->d_aux); (void)hipFree(w->d_offs);
  if (w->h_tot) (void)hipHostFree(w->h_tot);
  *w = VcRadiusWork();
}

// All items within full distance <= radius of each query, ascending per query, written to d_out (device memory,
// out_cap entries) with d_offsets[nq + 1] (device).  Everything is enqueued on `s`; the host synchronises ONCE at the
// end to learn the total and whether a query outgrew its ring (then the ring is doubled and the call repeated).
// *total = entries needed (may exceed out_cap: VC_ERR_CAPACITY, offsets valid).
static int radius_search_device(VcMihIndex* ix, bool use_mih, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W,
                                uint32_t id_base, uint32_t n_cu, const VcKnobs* knobs, const uint64_t* d_q, uint32_t nq,
                                uint32_t radius, uint64_t* d_out, uint64_t out_cap, uint64_t* d_offsets, uint64_t* total,
                                VcRadiusWork* wk, hipStream_t s, std::string* err) {
  int rc;
  if (use_mih && (rc = upload_binom(err))) return rc;
  const uint32_t bits = W * 64;
  if (radius > bits) radius = bits;
  // pigeonhole (search_R_neighbors shells, search_worker.cc:222-227) with multi-index hashing's sharper radii:
  // R = m q + a  =>  tables 0..a search substring radius q (n_big of them), the others q - 1 (small_shells shells)
  const uint32_t rq = use_mih ? radius / ix->m : 0, ra = use_mih ? radius % ix->m : 0;
  const uint32_t rsub = use_mih ? std::min(ix->sbits, rq) : 0;
  const uint32_t n_big = use_mih ? std::min(ix->m, ra + 1) : 0;
  const uint32_t small_shells = (use_mih && rq) ? std::min(ix->sbits, rq - 1) + 1 : 0;
  auto tables_at = [&](uint32_t r) { return r < small_shells ? ix->m : n_big; };
  bool inblock = false;
  if (use_mih && ix->knobs.mih_host_loop == 0) {
    uint64_t probes = 0;
    for (uint32_t r = 0; r <= rsub; ++r) probes += (uint64_t)tables_at(r) * binom_host(ix->sbits, r);
    const double avg_bucket = (double)ix->n / (ix->sbits >= 32 ? 4294967296.0 : (double)(1ull << ix->sbits));
    inblock = probes <= MQ_RADIUS_BUDGET && rsub <= 16 && (double)probes * avg_bucket <= MQ_ENTRY_BUDGET;
    if (ix->knobs.mih_stream == 2 && ix->sbits <= 16) inblock = false;   // tests: small databases through the streaming kernel too
  }
  // <= 16-bit substrings whose shells exceed the query kernel's entry budget: stream the buckets (mih_bucket_stream_kernel)
  uint32_t stream_probes = 0;
  bool stream = false;
  if (use_mih && !inblock && ix->knobs.mih_host_loop == 0 && ix->knobs.mih_stream != 0 && ix->sbits <= 16) {
    for (uint32_t r = 0; r <= rsub; ++r) stream_probes += tables_at(r) * binom_host(ix->sbits, r);
    stream = stream_probes <= MS_MAXP;
    for (const VcTableView& tv : ix->h_tables) stream = stream && tv.bcodes != nullptr;
  }
  const uint32_t TQ = use_mih ? MIH_RADIUS_TILE : 64u;
  uint32_t cap = std::max(wk->cap, use_mih ? std::max(ix->cap, 4096u) : 65536u);
  while (cap & (cap - 1)) cap += cap & (0u - cap);   // power of two: the in-place segment sort pads to one
#define R_CHECK(call)                                                      \
  do {                                                                     \
    hipError_t _r = (call);                                                \
    if (_r != hipSuccess) {                                                \
      if (err) *err = std::string(#call) + ": " + hipGetErrorString(_r);   \
      return _r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP;        \
    }                                                                      \
  } while (0)

  const uint32_t hs = (bits + 1 + 7) & ~7u;
  const size_t aux_words = (size_t)TQ * (3 + hs) + 8;   // count[TQ] | tau[TQ] | sorted[TQ] | hist[TQ*hs] | tot (2 x u64)
  if (wk->aux_words < aux_words) {
    (void)hipFree(wk->d_aux);
    wk->d_aux = nullptr;
    R_CHECK(hipMalloc((void**)&wk->d_aux, aux_words * 4));
    wk->aux_words = aux_words;
  }
  if (!wk->h_tot) {
    R_CHECK(hipHostMalloc((void**)&wk->h_tot, 32, hipHostMallocMapped));
    wk->h_tot[2] = 0;
    if (hipHostGetDevicePointer((void**)&wk->h_tot_dev, wk->h_tot, 0) != hipSuccess) { (void)hipGetLastError(); wk->h_tot_dev = nullptr; }
  }
  const bool poll = wk->h_tot_dev && (!knobs || knobs->mih_poll);
  uint32_t* d_count = wk->d_aux;
  uint32_t* d_tau = wk->d_aux + TQ;
  uint32_t* d_sorted = wk->d_aux + 2 * TQ;
  uint32_t* d_hist = wk->d_aux + 3 * TQ;
  unsigned long long* d_tot = (unsigned long long*)(wk->d_aux + (((size_t)TQ * (3 + hs) + 1) & ~(size_t)1));
  MihState st{};

  for (int attempt = 0;; ++attempt) {
    if (wk->cap != cap || !wk->d_ring) {
      (void)hipFree(wk->d_ring);
      wk->d_ring = nullptr;
      R_CHECK(hipMalloc((void**)&wk->d_ring, (size_t)TQ * cap * 8));
      wk->cap = cap;
    }
    if (nq == 0) R_CHECK(hipMemsetAsync(d_tot, 0, 16, s));   // (else the first tile's offsets kernel starts the totals over)
    for (uint32_t q0 = 0; q0 < nq; q0 += TQ) {
      const uint32_t qt = std::min(TQ, nq - q0);
      const uint32_t* sorted_flag = nullptr;
      const unsigned long long* work = nullptr;   // mih_query_kernel's per-query work counters of this tile
      if (use_mih) {
        if ((rc = ensure_tile(ix, MIH_RADIUS_TILE, 1, 1, false, &st, err))) return rc;
        st.ring = wk->d_ring;   // radius search keeps every neighbour: the big ring instead of the tile's
        st.count = d_count;
        st.topn = d_sorted;
        if (inblock) {
          QueryKernelParams qp{};
          qp.cols = d_cols; qp.stride = stride; qp.n = ix->n; qp.tables = ix->d_tables; qp.queries = d_q + (size_t)q0 * W;
          qp.st = st; qp.m = ix->m; qp.sbits = ix->sbits; qp.id_base = id_base; qp.flags = ix->flags; qp.cap = cap; qp.k = 0;
          qp.mode = MQ_MODE_RADIUS; qp.radius = radius; qp.r_last = rsub; qp.n_big = n_big; qp.small_shells = small_shells;
          qp.buf_entries = 2048;
          R_CHECK(timed_query_launch(ix, qp, W, qt, s));
          sorted_flag = d_sorted;
          work = st.work;
        } else if (stream) {
          uint32_t* list = ix->d_lists;
          hipLaunchKernelGGL(mih_init_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, st, qt, list, vc_pack(radius + 1, 0));
          R_CHECK(hipGetLastError());
          StreamParams sp{};
          sp.queries = d_q + (size_t)q0 * W; sp.tables = ix->d_tables; sp.ring = wk->d_ring; sp.count = d_count; sp.n = ix->n;
          sp.m = ix->m; sp.sbits = ix->sbits; sp.id_base = id_base; sp.flags = ix->flags; sp.cap = cap; sp.radius = radius;
          sp.rsub = rsub; sp.n_big = n_big; sp.small_shells = small_shells; sp.nprobes = stream_probes;
          // enough blocks to fill the chip when the batch is small: a query's probe list is dealt out to `split` blocks
          // (r04: 8 x n_cu blocks were 2048 for a tile of 1024 queries, 1.6 residency waves of 1280 -- the last fifth of the launch
          // ran on 768 blocks and fewer; one wave of blocks that are all resident ends 1.3 % earlier, more and smaller blocks pay
          // for their look-ups: 0.396 / 0.391 / 0.406 ms at 8 / 4 / 16 blocks per CU, profiles/r04_stream_trace.txt)
          sp.split = std::max(1u, std::min(std::min(stream_probes, 64u), stream_resident_blocks(W, n_cu) / qt));
          R_CHECK(timed_stream_launch(ix, sp, W, qt, s));
        } else {
          uint32_t* list = ix->d_lists;
          hipLaunchKernelGGL(mih_init_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, st, qt, list, vc_pack(radius + 1, 0));
          R_CHECK(hipGetLastError());
          for (uint32_t r = 0; r <= rsub; ++r) {
            ProbeParams p{};
            p.cols = d_cols; p.stride = stride; p.tables = ix->d_tables; p.queries = d_q + (size_t)q0 * W; p.list = list;
            p.st = st; p.r = r; p.nkeys = binom_host(ix->sbits, r); p.m = ix->m; p.sbits = ix->sbits; p.id_base = id_base;
            p.flags = ix->flags; p.cap = cap; p.count_seen = 1; p.n = ix->n; p.m_probe = tables_at(r);
            R_CHECK(launch_probe(p, W, qt, s));
          }
        }
      } else {
        R_CHECK(hipMemsetAsync(wk->d_aux, 0, (size_t)TQ * (3 + hs) * 4, s));
        hipLaunchKernelGGL(vc_fill_u32_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, d_tau, qt, radius);
        R_CHECK(hipGetLastError());
        size_t lds;
        const VcScanShape sh = vc_scan_pick_shape(W, qt, &lds, knobs);
        VcScanParams p{};
        p.cols = d_cols; p.stride = stride; p.n = n; p.nchunks = (n + sh.chunk_items() - 1) / sh.chunk_items();
        p.id_base = id_base; p.qt = qt; p.k = 0xFFFFFFFFu;   // never re-derive tau: it is the fixed radius
        p.cap = cap; p.hist_stride = hs; p.queries = d_q + (size_t)q0 * W; p.tau = d_tau; p.count = d_count; p.qs = 1;
        p.hist = d_hist; p.buf = wk->d_ring;
        R_CHECK(vc_launch_scan(p, W, n_cu, 0, knobs, s));
      }
      // place the tile's segments behind the previous tiles' (+ the call's totals, + the query kernel's work counters), then
      // order what did not arrive sorted (hand-written bitonic network; the query kernel's small segments do) and copy out
      const unsigned long long seq = (poll && q0 + TQ >= nq) ? ++wk->seq : 0ull;
      hipLaunchKernelGGL(vc_radius_offsets_kernel, dim3(work ? 2 : 1), dim3(1024), 0, s, d_count, qt, d_offsets + q0, d_tot, q0 == 0 ? 1u : 0u,
                         (volatile unsigned long long*)wk->h_tot_dev, seq, work, work ? ix->d_totals : (unsigned long long*)nullptr);
      R_CHECK(hipGetLastError());
      hipLaunchKernelGGL(vc_sort_compact_segments_kernel, dim3(qt), dim3(1024), 0, s, wk->d_ring, cap, d_count, sorted_flag,
                         d_offsets + q0, d_out, out_cap);
      R_CHECK(hipGetLastError());
    }
    // total + largest segment: the last offsets kernel wrote them to mapped host memory; the host polls the sequence word and
    // returns while the last copy-out kernel may still run (results are in stream order; host readers copy behind it)
    bool landed = false;
    if (poll && nq) {
      landed = poll_mapped_word((volatile unsigned long long*)(wk->h_tot + 2), (unsigned long long)wk->seq);
    }
    if (!landed) {
      R_CHECK(hipMemcpyAsync(wk->h_tot, d_tot, 16, hipMemcpyDeviceToHost, s));
      R_CHECK(hipStreamSynchronize(s));
    }
    const uint64_t mx = wk->h_tot[1];
    *total = wk->h_tot[0];
    if (mx <= cap) break;
    uint64_t want = cap;
    while (want < mx) want <<= 1;
    if (attempt > 8 || want * TQ * 8 > (64ull << 30)) {
      if (err) *err = "radius search: a query has more neighbours than the work ring can hold";
      return VC_ERR_CAPACITY;
    }
    cap = (uint32_t)want;
  }
#undef R_CHECK
  if (*total > out_cap) {
    if (err) *err = "radius search: output buffer too small (needed counts are in out_offsets)";
    return VC_ERR_CAPACITY;
  }
  return VC_OK;
}

int vc_radius_search(VcMihIndex* ix, bool use_mih, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W,
                     uint32_t id_base, uint32_t n_cu, const VcKnobs* knobs, const uint64_t* d_q, uint32_t nq, uint32_t radius,
                     uint64_t* out, uint64_t out_cap, uint64_t* out_offsets, bool device_out, VcRadiusWork* wk, hipStream_t s,
                     std::string* err) {
  if (use_mih && n != ix->n) {
    if (err) *err = "index is stale: codes were added after vc_build_index()";
    return VC_ERR_STATE;
  }
  if (use_mih) {
    // a radius whose substring shells cost more probes than scanning the shard costs distance evaluations is answered
    // by the scan (identical results; search_R_neighbors would enumerate up to 2^s keys per table, search_worker.cc:222-264)
    const uint32_t rr = std::min(radius, W * 64), rq = rr / ix->m, ra = rr % ix->m;   // radii q (tables 0..a) and q - 1, as below
    const uint32_t rsub = std::min(ix->sbits, rq), n_big = std::min(ix->m, ra + 1);
    const uint32_t small_shells = rq ? std::min(ix->sbits, rq - 1) + 1 : 0;
    double probes = 0;
    for (uint32_t r = 0; r <= rsub; ++r) probes += (double)(r < small_shells ? ix->m : n_big) * binom_host(ix->sbits, r);
    if (probes > (double)std::max<uint64_t>(n, 1u << 20)) use_mih = false;
  }
  const uint32_t TQ = use_mih ? MIH_RADIUS_TILE : 64u;
  if (wk->tq != TQ) {   // tile shape changed (scan <-> MIH): start over with fresh buffers
    vc_radius_work_free(wk);
    wk->tq = TQ;
  }
  if (device_out) {
    uint64_t total = 0;
    return radius_search_device(ix, use_mih, d_cols, stride, n, W, id_base, n_cu, knobs, d_q, nq, radius, out, out_cap, out_offsets,
                                &total, wk, s, err);
  }
  // host-pointer API: stage in device memory, grow the staging buffer to what the call needs, one copy back
  if (wk->offs_cap < (size_t)nq + 1) {
    (void)hipFree(wk->d_offs);
    wk->d_offs = nullptr;
    if (hipMalloc((void**)&wk->d_offs, ((size_t)nq + 1) * 8) != hipSuccess) {
      if (err) *err = "radius search: offsets allocation failed";
      return VC_ERR_NOMEM;
    }
    wk->offs_cap = (size_t)nq + 1;
  }
  uint64_t total = 0;
  for (int pass = 0; pass < 2; ++pass) {
    if (!wk->d_compact) {
      if (wk->compact_cap == 0) wk->compact_cap = std::max<uint64_t>(out_cap, 1u << 16);
      if (hipMalloc((void**)&wk->d_compact, wk->compact_cap * 8) != hipSuccess) {
        if (err) *err = "radius search: staging allocation failed";
        return VC_ERR_NOMEM;
      }
    }
    std::string e2;
    int rc = radius_search_device(ix, use_mih, d_cols, stride, n, W, id_base, n_cu, knobs, d_q, nq, radius, wk->d_compact, wk->compact_cap,
                                  wk->d_offs, &total, wk, s, &e2);
    if (rc == VC_OK) break;
    if (rc != VC_ERR_CAPACITY || total <= wk->compact_cap || pass == 1) {
      if (err) *err = e2;
      return rc;
    }
    (void)hipFree(wk->d_compact);   // the staging buffer was too small: now the needed size is known
    wk->d_compact = nullptr;
    wk->compact_cap = total * 2;
  }
  if (hipMemcpyAsync(out_offsets, wk->d_offs, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
      (total <= out_cap && total && hipMemcpyAsync(out, wk->d_compact, total * 8, hipMemcpyDeviceToHost, s) != hipSuccess) ||
      hipStreamSynchronize(s) != hipSuccess) {
    if (err) *err = "radius search: copy back failed";
    return VC_ERR_HIP;
  }
  if (total > out_cap) {
    if (err) *err = "radius search: output buffer too small (needed counts are in out_offsets)";
    return VC_ERR_CAPACITY;
  }
  return VC_OK;
}
