// Multi-index-hashing side of the engine (vc_mih.hip): index build, bucket views, radius-incremental search.
#pragma once
#include "vc_internal.hpp"

struct VcMihIndex;

int vc_mih_build(VcMihIndex** out, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t m,
                 uint32_t sbits, uint32_t id_base, uint32_t flags, uint32_t n_cu, uint32_t cand_cap, const VcKnobs& knobs,
                 hipStream_t s, std::string* err);
void vc_mih_free(VcMihIndex* ix);
// index persistence: the CSR tables + bitmaps of a built index, and back (the codes themselves travel as a code file)
int vc_mih_save(VcMihIndex* ix, const uint64_t* d_cols, uint64_t stride, const char* path, hipStream_t s, std::string* err);
int vc_mih_load(VcMihIndex** out, const char* path, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t m,
                uint32_t sbits, uint32_t id_base, uint32_t flags, uint32_t n_cu, uint32_t cand_cap, const VcKnobs& knobs,
                hipStream_t s, std::string* err);
// sums and resets the measurement records of mih_query_kernel: device time of its launches (HIP events on the launch
// stream) and the algorithmic work counters {bucket probes, non-empty buckets, bucket entries verified, queries}
void vc_mih_timing(VcMihIndex* ix, float* ms, uint32_t* launches, uint64_t totals[4], hipStream_t s);
// Cost-model switch of the exact k-NN loop (the reference loops blindly to radius 32, search_worker.cc:170):
// when the shells a query still has to probe cost more than streaming the shard once, the engine answers it with the
// verify kernel and REPLAYS the stop rule of search_worker.cc:201-205 on the scan's candidates, so that results and
// statistics are those of the radius loop.  Where a resolved query is left: the front of its candidate ring + the
// statistics arrays of the search tile (mih_export_kernel and the statistics read-back then treat it like any other).
struct VcMihScanTarget {
  uint64_t* ring;            // [slot][cap]
  uint32_t cap;
  uint32_t* count;           // [slot] results written to the ring front
  uint32_t* radius;          // [slot] last shell the radius loop would have searched
  unsigned long long* seen;  // [slot] distinct items that loop would have verified (only when want_stats)
  unsigned long long* sub;   // [slot] table 0's bucket gets (closed form: every leaf of shells 0..radius)
  unsigned long long* loc;   // [slot] 0 (no bitmap attached on this path)
};
// d_q: the tile's queries ([slot][W]); d_list[0..n): slots to answer; d_unresolved / d_n_unresolved: slots the scan could
// not settle (ring overflow, too many ties) -- they continue in the radius loop.  Enqueued on s, no host sync.
typedef int (*vc_mih_scan_fn)(void* ctx, const uint64_t* d_q, const uint32_t* d_list, uint32_t n, uint32_t k, uint32_t stop_mult,
                              const VcMihScanTarget& tgt, bool want_stats, uint32_t* d_unresolved, uint32_t* d_n_unresolved,
                              hipStream_t s);
struct VcMihScanFallback {
  vc_mih_scan_fn fn = nullptr;
  void* ctx = nullptr;
  uint32_t n_cu = 0;
};
// kernels of that switch (vc_mih.hip), launched by the engine between its select and recover launches
hipError_t vc_launch_sort_compact_segments(uint64_t* d_ring, uint32_t cap, const uint32_t* d_count, const uint64_t* d_offs,
                                           uint64_t* d_out, uint64_t out_cap, uint32_t nq, hipStream_t s);
hipError_t vc_launch_gather_queries(const uint64_t* d_q, const uint32_t* d_list, uint32_t n, uint32_t W, uint64_t* d_out, hipStream_t s);
struct VcMihReplayArgs {
  const uint64_t* lin_ring;   // [gq][lin_cap] candidate rings of the scan (complete: every item at or below the k-th distance)
  const uint32_t* lin_count;  // raw ring cursors, lin_qs words apart
  const uint64_t* rows;       // [gq][k] select output (ascending, INF padded)
  const uint32_t* rows_cnt;   // [gq] (UINT32_MAX = ring overflowed)
  const uint64_t* queries;    // [gq][W] gathered queries
  const uint32_t* list;       // [gq] slots
  const uint64_t* cols;
  uint64_t stride;
  uint32_t lin_cap, lin_qs, gq, k, m, sbits, W, stop_mult, id_base;
  VcMihScanTarget tgt;
  uint32_t* unresolved;
  uint32_t* n_unresolved;
  uint32_t* resolved_flag;    // [gq] 1 = settled here
};
hipError_t vc_launch_mih_replay(const VcMihReplayArgs& a, hipStream_t s);
// seen[list[i]] = #{items whose minimum substring distance <= radius[list[i]]} for the queries with flag[i] != 0
hipError_t vc_launch_minsub_count(const uint64_t* cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t m, uint32_t sbits,
                                  const uint64_t* d_queries, const uint32_t* d_list, const uint32_t* d_flag, uint32_t nq,
                                  const uint32_t* d_radius, unsigned long long* d_seen, uint32_t n_cu, hipStream_t s);

// d_q [nq][W]; d_out [nq][k] ascending INF-padded; d_cnt [nq]; stats (host, may be null) filled after a sync.
// d_stats (device, may be null): the same records written by a kernel in stream order, no host wait for them.
int vc_mih_search(VcMihIndex* ix, const uint64_t* d_cols, uint64_t stride, uint64_t n, const uint64_t* d_q, uint32_t nq,
                  uint32_t k, bool approximate, uint64_t* d_out, uint32_t* d_cnt, vc_query_stats* stats, hipStream_t s,
                  std::string* err, const VcMihScanFallback* fb = nullptr, vc_query_stats* d_stats = nullptr);
int vc_mih_bucket(VcMihIndex* ix, uint32_t table, uint32_t index, std::vector<uint32_t>* local_ids, hipStream_t s,
                  std::string* err);
int vc_mih_bitmap_test(VcMihIndex* ix, uint32_t table, uint32_t index, int* bit, hipStream_t s, std::string* err);
int vc_mih_bitmap_read(VcMihIndex* ix, uint32_t table, uint64_t word_off, uint64_t n_words, uint32_t* out, hipStream_t s,
                       std::string* err);
// grow-only device buffers of the radius search, owned by the engine (allocation costs more than a search)
struct VcRadiusWork {
  uint64_t *d_ring = nullptr, *d_compact = nullptr, *d_offs = nullptr;
  uint32_t* d_aux = nullptr;
  unsigned long long* h_tot = nullptr;   // pinned, mapped: total entries | largest segment | sequence number of the call that wrote them
  unsigned long long* h_tot_dev = nullptr;
  unsigned long long seq = 0;
  size_t aux_words = 0, offs_cap = 0;
  uint64_t compact_cap = 0;
  uint32_t cap = 0, tq = 0;
};
void vc_radius_work_free(VcRadiusWork* w);

// All items within full distance <= radius, ascending per query; ix may be null when use_mih is false (linear scan).
// device_out = false: out / out_offsets are host memory (staged through the engine's buffers);
// device_out = true : out (out_cap entries) / out_offsets (nq + 1) are device memory, everything is enqueued on `s`
// and the host synchronises once at the end.
int vc_radius_search(VcMihIndex* ix, bool use_mih, const uint64_t* d_cols, uint64_t stride, uint64_t n, uint32_t W,
                     uint32_t id_base, uint32_t n_cu, const VcKnobs* knobs, const uint64_t* d_q, uint32_t nq, uint32_t radius,
                     uint64_t* out, uint64_t out_cap, uint64_t* out_offsets, bool device_out, VcRadiusWork* work, hipStream_t s,
                     std::string* err);
