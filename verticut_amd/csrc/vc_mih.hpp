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
// d_q [nq][W]; d_out [nq][k] ascending INF-padded; d_cnt [nq]; stats (host, may be null) filled after a sync.
int vc_mih_search(VcMihIndex* ix, const uint64_t* d_cols, uint64_t stride, uint64_t n, const uint64_t* d_q, uint32_t nq,
                  uint32_t k, bool approximate, uint64_t* d_out, uint32_t* d_cnt, vc_query_stats* stats, hipStream_t s,
                  std::string* err);
int vc_mih_bucket(VcMihIndex* ix, uint32_t table, uint32_t index, std::vector<uint32_t>* local_ids, hipStream_t s,
                  std::string* err);
int vc_mih_bitmap_test(VcMihIndex* ix, uint32_t table, uint32_t index, int* bit, hipStream_t s, std::string* err);
int vc_mih_bitmap_read(VcMihIndex* ix, uint32_t table, uint64_t word_off, uint64_t n_words, uint32_t* out, hipStream_t s,
                       std::string* err);
// grow-only device buffers of the radius search, owned by the engine (allocation costs more than a search)
struct VcRadiusWork {
  uint64_t *d_ring = nullptr, *d_compact = nullptr, *d_offs = nullptr;
  uint32_t* d_aux = nullptr;
  unsigned long long* h_tot = nullptr;   // pinned: total entries and largest segment of a call
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
