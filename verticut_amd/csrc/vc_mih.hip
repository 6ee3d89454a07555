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
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>

#include "vc_internal.hpp"
#include "vc_mih.hpp"

#define MIH_BLK 256
#define MIH_PPT 4                       // probes per thread
#define MIH_EPT 4u                      // bucket entries per thread and round in the verify phase
#define MIH_PCH (MIH_BLK * MIH_PPT)     // probes per block pass
#define MIH_QTILE 256u                  // queries resident per search tile
#define MIH_APPROX_FACTOR 20u           // search_worker.h:14

struct VcTableView {
  const uint32_t* offsets;
  const uint32_t* ids;
  const uint32_t* bitmap;
  const uint32_t* blockrank;  // s == 32 only
  // Optional copy of the codes in THIS table's bucket order (word j of the pos-th entry at bcodes[j*n + pos]), built
  // for substrings <= 16 bit: their buckets hold thousands of items (1526 at 1e8 codes, s = 16), and verifying a
  // bucket through ids[] -> cols[] is an 8-byte gather per word that moves a 64-byte sector each; from the copy it is
  // a contiguous stream.  32-bit substrings keep gathering (0.02 items per bucket).
  const uint64_t* bcodes;
  uint32_t n_unique;
  uint32_t pad;
};

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

__global__ void __launch_bounds__(256) mih_export_kernel(MihState st, uint32_t nq, uint32_t k, uint32_t cap,
                                                         uint64_t* __restrict__ out, uint32_t* __restrict__ cnt) {
  const uint32_t q = blockIdx.x;
  const uint32_t n = min(st.count[q], k);
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) out[(uint64_t)q * k + i] = i < n ? st.ring[(uint64_t)q * cap + i] : VC_PACK_INF;
  if (threadIdx.x == 0) cnt[q] = n;
}

__global__ void __launch_bounds__(256) vc_fill_u32_kernel(uint32_t* p, uint32_t n, uint32_t v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// sorted ring segments -> one contiguous result array (offs = exclusive prefix of the segment lengths)
__global__ void __launch_bounds__(256) vc_compact_segments_kernel(const uint64_t* __restrict__ sorted, uint32_t cap,
                                                                  const uint64_t* __restrict__ offs, uint32_t nq,
                                                                  uint64_t* __restrict__ out) {
  const uint32_t q = blockIdx.x;
  const uint64_t lo = offs[q] - offs[0], n = offs[q + 1] - offs[q];
  for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) out[lo + i] = sorted[(uint64_t)q * cap + i];
}

__global__ void __launch_bounds__(256) vc_seg_bounds_kernel(const uint32_t* count, uint32_t nq, uint32_t cap, uint32_t* beg,
                                                            uint32_t* end) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) {
    beg[i] = i * cap;
    end[i] = i * cap + min(count[i], cap);
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
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
  uint32_t* d_lists = nullptr;   // 3 * MIH_QTILE + 4 counters
  uint32_t* h_ctr = nullptr;     // pinned: the two counters the host reads back after every shell
};

#define MIH_CHECK(call)                                                                                  \
  do {                                                                                                   \
    hipError_t _r = (call);                                                                              \
    if (_r != hipSuccess) {                                                                              \
      if (err) *err = std::string(#call) + ": " + hipGetErrorString(_r);                                 \
      return _r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP;                                      \
    }                                                                                                    \
  } while (0)

static uint32_t grid_for(uint64_t n, uint32_t n_cu) { return (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)n_cu * 16); }

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
  delete ix;
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
  const uint64_t nkeyspace = 1ull << sbits;
  const uint64_t bm_words = std::max<uint64_t>(nkeyspace / 32, 8);
  const uint32_t mask = sbits == 32 ? 0xFFFFFFFFu : (uint32_t)(nkeyspace - 1);
  const uint64_t nn = std::max<uint64_t>(n, 1);

  uint32_t *k_in = nullptr, *k_out = nullptr, *v_in = nullptr;
  void* d_temp = nullptr;
  size_t temp_bytes = 0, scan_bytes = 0;
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
  B_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, k_in, k_out, v_in, v_in, (int64_t)n, 0, (int)sbits, s));
  const uint64_t scan_n = sbits == 32 ? (1ull << 24) : nkeyspace + 1;
  B_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, k_in, k_in, (int64_t)scan_n, s));
  temp_bytes = std::max(temp_bytes, scan_bytes);
  B_CHECK(dalloc(&d_temp, temp_bytes, false));
  B_CHECK(dalloc((void**)&d_scan_in, (scan_n + 1) * 4, false));

  for (uint32_t t = 0; t < m; ++t) {
    uint32_t *ids = nullptr, *bitmap = nullptr, *offsets = nullptr, *blockrank = nullptr;
    B_CHECK(dalloc((void**)&ids, nn * 4, true));
    B_CHECK(dalloc((void**)&bitmap, bm_words * 4, true));
    B_CHECK(hipMemsetAsync(bitmap, 0, bm_words * 4, s));
    const uint32_t bitpos = t * sbits;
    if (n) {
      hipLaunchKernelGGL(mih_keys_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, d_cols + (uint64_t)(bitpos >> 6) * stride, n,
                         bitpos & 63, mask, k_in, v_in);
      B_CHECK(hipGetLastError());
      // stable LSD radix sort by key: ids stay ascending inside a bucket = append order of build_hash_tables.cc:54-63
      B_CHECK(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, k_in, k_out, v_in, ids, (int64_t)n, 0, (int)sbits, s));
    }
    VcTableView tv{};
    if (sbits < 32) {
      B_CHECK(dalloc((void**)&offsets, (nkeyspace + 1) * 4, true));
      B_CHECK(hipMemsetAsync(d_scan_in, 0, (nkeyspace + 1) * 4, s));
      if (n) {
        hipLaunchKernelGGL(mih_runs_kernel, dim3(grid_for(n, n_cu)), dim3(256), 0, s, k_out, n, bitmap, d_scan_in);
        B_CHECK(hipGetLastError());
      }
      B_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_scan_in, offsets, (int64_t)(nkeyspace + 1), s));
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
      B_CHECK(hipcub::DeviceScan::ExclusiveSum(d_temp, temp_bytes, d_scan_in, blockrank, (int64_t)nblocks, s));
      uint32_t last_rank = 0, last_pop = 0;
      B_CHECK(hipMemcpyAsync(&last_rank, blockrank + nblocks - 1, 4, hipMemcpyDeviceToHost, s));
      B_CHECK(hipMemcpyAsync(&last_pop, d_scan_in + nblocks - 1, 4, hipMemcpyDeviceToHost, s));
      B_CHECK(hipStreamSynchronize(s));
      tv.n_unique = last_rank + last_pop;
      B_CHECK(dalloc((void**)&offsets, ((size_t)tv.n_unique + 1) * 4, true));
      hipLaunchKernelGGL(mih_ranked_offsets_kernel, dim3(grid_for(nn, n_cu)), dim3(256), 0, s, k_out, n, bitmap, blockrank,
                         offsets, tv.n_unique);
      B_CHECK(hipGetLastError());
    }
    tv.offsets = offsets;
    tv.ids = ids;
    tv.bitmap = bitmap;
    tv.blockrank = blockrank;
    tv.bcodes = nullptr;
    if (want_bcodes && n) {
      uint64_t* bc = nullptr;
      B_CHECK(dalloc((void**)&bc, (size_t)n * W * 8, true));
      hipLaunchKernelGGL(mih_bcodes_kernel, dim3(grid_for(n * W, n_cu)), dim3(256), 0, s, d_cols, stride, W, ids, n, bc);
      B_CHECK(hipGetLastError());
      tv.bcodes = bc;
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

// ---- search -------------------------------------------------------------------------------------------
static int ensure_tile(VcMihIndex* ix, uint32_t k, uint32_t cap, MihState* st, std::string* err) {
  const size_t Q = MIH_QTILE;
  size_t bytes = 0;
  auto take = [&](size_t b) { size_t o = bytes; bytes += (b + 255) & ~(size_t)255; return o; };
  const size_t o_thresh = take(Q * 8), o_ring = take(Q * cap * 8), o_count = take(Q * 4), o_prev = take(Q * 4),
               o_seen = take(Q * 8), o_sub = take(Q * 8), o_loc = take(Q * 8), o_radius = take(Q * 4),
               o_topk = take(Q * (size_t)k * 8), o_topn = take(Q * 4);
  if (bytes > ix->tile_bytes) {
    if (ix->d_tile) MIH_CHECK(hipFree(ix->d_tile));
    ix->d_tile = nullptr;
    ix->tile_bytes = 0;
    MIH_CHECK(hipMalloc(&ix->d_tile, bytes));
    ix->tile_bytes = bytes;
  }
  if (!ix->d_lists) MIH_CHECK(hipMalloc((void**)&ix->d_lists, (3 * Q + 4) * 4));
  uint8_t* b = (uint8_t*)ix->d_tile;
  st->thresh = (uint64_t*)(b + o_thresh);
  st->ring = (uint64_t*)(b + o_ring);
  st->count = (uint32_t*)(b + o_count);
  st->prev = (uint32_t*)(b + o_prev);
  st->seen = (unsigned long long*)(b + o_seen);
  st->sub = (unsigned long long*)(b + o_sub);
  st->loc = (unsigned long long*)(b + o_loc);
  st->radius = (uint32_t*)(b + o_radius);
  st->topk = (uint64_t*)(b + o_topk);
  st->topn = (uint32_t*)(b + o_topn);
  return VC_OK;
}

static hipError_t launch_probe(const ProbeParams& p, uint32_t W, uint32_t n_list, hipStream_t s) {
  const dim3 grid((p.nkeys + MIH_PCH - 1) / MIH_PCH, p.m, n_list);
  switch (W) {
    case 1: hipLaunchKernelGGL(mih_probe_kernel<1>, grid, dim3(MIH_BLK), 0, s, p); break;
    case 2: hipLaunchKernelGGL(mih_probe_kernel<2>, grid, dim3(MIH_BLK), 0, s, p); break;
    case 4: hipLaunchKernelGGL(mih_probe_kernel<4>, grid, dim3(MIH_BLK), 0, s, p); break;
    case 8: hipLaunchKernelGGL(mih_probe_kernel<8>, grid, dim3(MIH_BLK), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

static uint32_t binom_host(uint32_t n, uint32_t k) {
  uint64_t r = 1;
  for (uint32_t i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return (uint32_t)r;
}

int vc_mih_search(VcMihIndex* ix, const uint64_t* d_cols, uint64_t stride, uint64_t n, const uint64_t* d_q, uint32_t nq,
                  uint32_t k, bool approximate, uint64_t* d_out, uint32_t* d_cnt, vc_query_stats* stats, hipStream_t s,
                  std::string* err) {
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
  if ((rc = ensure_tile(ix, k, cap, &st, err))) return rc;
  uint32_t* lists[3] = {ix->d_lists, ix->d_lists + MIH_QTILE, ix->d_lists + 2 * MIH_QTILE};
  uint32_t* d_ctr = ix->d_lists + 3 * MIH_QTILE;

  for (uint32_t q0 = 0; q0 < nq; q0 += MIH_QTILE) {
    const uint32_t qt = std::min(MIH_QTILE, nq - q0);
    uint32_t *cur = lists[0], *nxt = lists[1], *redo = lists[2];
    hipLaunchKernelGGL(mih_init_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, st, qt, cur, (uint64_t)VC_PACK_INF);
    MIH_CHECK(hipGetLastError());
    uint32_t n_cur = qt;
    const bool trace = ix->knobs.mih_trace;   // VC_MIH_TRACE: per-shell wall times on stderr
    for (uint32_t r = 0; r <= S && n_cur; ++r) {       // search_worker.cc:170: radius <= n_local_bytes_*8
      const auto t_shell = std::chrono::steady_clock::now();
      ProbeParams p{};
      p.cols = d_cols; p.stride = stride; p.tables = ix->d_tables; p.queries = d_q + (size_t)q0 * ix->W;
      p.st = st; p.r = r; p.nkeys = binom_host(S, r); p.m = ix->m; p.sbits = S; p.id_base = ix->id_base;
      p.flags = ix->flags; p.cap = cap; p.n = ix->n;
      CommitParams c{};
      c.st = st; c.next_list = nxt; c.redo_list = redo; c.ctr = d_ctr; c.k = k; c.cap = cap; c.r = r; c.sbits = S;
      c.stop_mult = stop_mult; c.approximate = approximate; c.last_shell = r == S;
      MIH_CHECK(hipMemsetAsync(d_ctr, 0, 16, s));
      const uint32_t* work = cur;
      uint32_t n_work = n_cur;
      if (!ix->h_ctr) MIH_CHECK(hipHostMalloc((void**)&ix->h_ctr, 16, hipHostMallocDefault));   // pageable memory makes the 8-byte read-back a staged copy
      uint32_t* h_ctr = ix->h_ctr;
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
    hipLaunchKernelGGL(mih_export_kernel, dim3(qt), dim3(256), 0, s, st, qt, k, cap, d_out + (size_t)q0 * k, d_cnt + q0);
    MIH_CHECK(hipGetLastError());
    if (stats) {
      std::vector<unsigned long long> seen(qt), sub(qt), loc(qt);
      std::vector<uint32_t> rad(qt);
      MIH_CHECK(hipMemcpyAsync(seen.data(), st.seen, qt * 8, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipMemcpyAsync(sub.data(), st.sub, qt * 8, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipMemcpyAsync(loc.data(), st.loc, qt * 8, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipMemcpyAsync(rad.data(), st.radius, qt * 4, hipMemcpyDeviceToHost, s));
      MIH_CHECK(hipStreamSynchronize(s));
      for (uint32_t i = 0; i < qt; ++i) {
        vc_query_stats& o = stats[q0 + i];
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
  (void)hipFree(w->d_ring); (void)hipFree(w->d_sorted); (void)hipFree(w->d_compact); (void)hipFree(w->d_aux); (void)hipFree(w->d_temp);
  *w = VcRadiusWork();
}

int vc_radius_search(VcMihIndex* ix, bool use_mih, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W,
                     uint32_t id_base, uint32_t n_cu, const uint64_t* d_q, uint32_t nq, uint32_t radius, uint64_t* out,
                     uint64_t out_cap, uint64_t* out_offsets, VcRadiusWork* wk, hipStream_t s, std::string* err) {
  if (use_mih && n != ix->n) {
    if (err) *err = "index is stale: codes were added after vc_build_index()";
    return VC_ERR_STATE;
  }
  int rc;
  if (use_mih && (rc = upload_binom(err))) return rc;
  const uint32_t bits = W * 64;
  if (radius > bits) radius = bits;
  const uint32_t TQ = use_mih ? MIH_QTILE : 64u;
  uint32_t cap = std::max(wk->cap, use_mih ? std::max(ix->cap, 4096u) : 65536u);
  if (wk->tq != TQ) {   // tile shape changed (scan <-> MIH): start over with fresh buffers
    vc_radius_work_free(wk);
    wk->tq = TQ;
  }
  std::vector<uint64_t> result;
  std::vector<uint64_t> offs(nq + 1, 0);

  // work buffers live in *wk across calls (references so the retry logic below can replace them)
  uint64_t*& d_ring = wk->d_ring;
  uint64_t*& d_sorted = wk->d_sorted;
  uint64_t*& d_compact = wk->d_compact;
  uint64_t& compact_cap = wk->compact_cap;
  uint32_t*& d_aux = wk->d_aux;     // count[TQ] | tau[TQ] | beg[TQ] | end[TQ] | hist[TQ*hs]
  void*& d_temp = wk->d_temp;
  size_t& temp_bytes = wk->temp_bytes;
  if (wk->cap != cap) {             // ring size changed: both rings and the sort workspace are re-made below
    (void)hipFree(d_ring); (void)hipFree(d_sorted);
    d_ring = d_sorted = nullptr;
  }
  auto cleanup = [&]() {};          // buffers stay with the engine
#define R_CHECK(call)                                                      \
  do {                                                                     \
    hipError_t _r = (call);                                                \
    if (_r != hipSuccess) {                                                \
      if (err) *err = std::string(#call) + ": " + hipGetErrorString(_r);   \
      cleanup();                                                           \
      return _r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP;        \
    }                                                                      \
  } while (0)

  const uint32_t hs = (bits + 1 + 7) & ~7u;
  if (wk->aux_words < (size_t)TQ * (4 + hs)) {
    (void)hipFree(d_aux);
    d_aux = nullptr;
    R_CHECK(hipMalloc((void**)&d_aux, (size_t)TQ * (4 + hs) * 4));
    wk->aux_words = (size_t)TQ * (4 + hs);
  }
  uint32_t* d_count = d_aux;
  uint32_t* d_tau = d_aux + TQ;
  uint32_t* d_beg = d_aux + 2 * TQ;
  uint32_t* d_end = d_aux + 3 * TQ;
  uint32_t* d_hist = d_aux + 4 * TQ;
  MihState st{};
  std::vector<uint32_t> h_count(TQ);

  for (uint32_t q0 = 0; q0 < nq; q0 += TQ) {
    const uint32_t qt = std::min(TQ, nq - q0);
    for (;;) {  // retry with a larger ring until every query's neighbours fit
      if (!d_ring) {
        R_CHECK(hipMalloc((void**)&d_ring, (size_t)TQ * cap * 8));
        R_CHECK(hipMalloc((void**)&d_sorted, (size_t)TQ * cap * 8));
        wk->cap = cap;
        temp_bytes = 0;
        R_CHECK(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, temp_bytes, d_ring, d_sorted, (int64_t)TQ * cap, (int)TQ,
                                                           d_beg, d_end, 0, 44, s));
        (void)hipFree(d_temp);
        d_temp = nullptr;
        R_CHECK(hipMalloc(&d_temp, std::max<size_t>(temp_bytes, 256)));
      }
      if (use_mih) {
        if ((rc = ensure_tile(ix, 1, 1, &st, err))) { cleanup(); return rc; }
        st.ring = d_ring;   // radius search keeps every neighbour: use the big ring instead of the tile's
        st.count = d_count;
        uint32_t* list = ix->d_lists;
        hipLaunchKernelGGL(mih_init_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, st, qt, list, vc_pack(radius + 1, 0));
        R_CHECK(hipGetLastError());
        // pigeonhole: dist <= R implies some substring within floor(R/m) (search_R_neighbors shells, search_worker.cc:222-227)
        const uint32_t rmax = std::min(ix->sbits, radius / ix->m);
        for (uint32_t r = 0; r <= rmax; ++r) {
          ProbeParams p{};
          p.cols = d_cols; p.stride = stride; p.tables = ix->d_tables; p.queries = d_q + (size_t)q0 * W; p.list = list;
          p.st = st; p.r = r; p.nkeys = binom_host(ix->sbits, r); p.m = ix->m; p.sbits = ix->sbits; p.id_base = id_base;
          p.flags = ix->flags; p.cap = cap; p.count_seen = 1; p.n = ix->n;
          R_CHECK(launch_probe(p, W, qt, s));
        }
      } else {
        R_CHECK(hipMemsetAsync(d_aux, 0, (size_t)TQ * (4 + hs) * 4, s));
        hipLaunchKernelGGL(vc_fill_u32_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, d_tau, qt, radius);
        R_CHECK(hipGetLastError());
        size_t lds;
        const VcScanShape sh = vc_scan_pick_shape(W, qt, &lds, nullptr);
        VcScanParams p{};
        p.cols = d_cols; p.stride = stride; p.n = n; p.nchunks = (n + sh.chunk_items() - 1) / sh.chunk_items();
        p.id_base = id_base; p.qt = qt; p.k = 0xFFFFFFFFu;   // never re-derive tau: it is the fixed radius
        p.cap = cap; p.hist_stride = hs; p.queries = d_q + (size_t)q0 * W; p.tau = d_tau; p.count = d_count; p.qs = 1;
        p.hist = d_hist; p.buf = d_ring;
        R_CHECK(vc_launch_scan(p, W, n_cu, 0, nullptr, s));
      }
      R_CHECK(hipMemcpyAsync(h_count.data(), d_count, qt * 4, hipMemcpyDeviceToHost, s));
      R_CHECK(hipStreamSynchronize(s));
      uint32_t mx = 0;
      for (uint32_t i = 0; i < qt; ++i) mx = std::max(mx, h_count[i]);
      if (mx <= cap) break;
      uint64_t want = 1;
      while (want < mx) want <<= 1;
      if (want * TQ * 16 > (64ull << 30)) {
        if (err) *err = "radius search: a query has more neighbours than the work ring can hold";
        cleanup();
        return VC_ERR_CAPACITY;
      }
      cap = (uint32_t)want;
      (void)hipFree(d_ring); (void)hipFree(d_sorted);
      d_ring = d_sorted = nullptr;
    }
    hipLaunchKernelGGL(vc_seg_bounds_kernel, dim3((qt + 255) / 256), dim3(256), 0, s, d_count, qt, cap, d_beg, d_end);
    R_CHECK(hipGetLastError());
    R_CHECK(hipcub::DeviceSegmentedRadixSort::SortKeys(d_temp, temp_bytes, d_ring, d_sorted, (int64_t)qt * cap, (int)qt, d_beg, d_end,
                                                       0, 44, s));
    for (uint32_t i = 0; i < qt; ++i) offs[q0 + i + 1] = offs[q0 + i] + h_count[i];
    const uint64_t tile_total = offs[q0 + qt] - offs[q0];
    if (tile_total) {   // gather the tile's segments on the device, one copy back
      if (tile_total > compact_cap) {
        (void)hipFree(d_compact);
        d_compact = nullptr;
        compact_cap = tile_total * 2;
        R_CHECK(hipMalloc((void**)&d_compact, compact_cap * 8 + (TQ + 1) * 8));
      }
      uint64_t* d_offs = d_compact + compact_cap;
      R_CHECK(hipMemcpyAsync(d_offs, offs.data() + q0, (size_t)(qt + 1) * 8, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(vc_compact_segments_kernel, dim3(qt), dim3(256), 0, s, d_sorted, cap, d_offs, qt, d_compact);
      R_CHECK(hipGetLastError());
      const size_t base = result.size();
      result.resize(base + tile_total);
      R_CHECK(hipMemcpyAsync(result.data() + base, d_compact, tile_total * 8, hipMemcpyDeviceToHost, s));
      R_CHECK(hipStreamSynchronize(s));
    }
  }
#undef R_CHECK
  memcpy(out_offsets, offs.data(), (nq + 1) * sizeof(uint64_t));
  if (offs[nq] > out_cap) {
    if (err) *err = "radius search: output buffer too small (needed counts are in out_offsets)";
    return VC_ERR_CAPACITY;
  }
  if (offs[nq]) memcpy(out, result.data(), offs[nq] * sizeof(uint64_t));
  return VC_OK;
}
