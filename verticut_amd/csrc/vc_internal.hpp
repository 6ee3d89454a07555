// Host-side launcher declarations shared by the translation units of libverticut_gpu.so.
#pragma once
#include "vc_common.hpp"

// Developer / test knobs from the environment, read ONCE per engine at vc_create (never on a launch path).
struct VcKnobs {
  uint32_t scan_wrap = 0, scan_diag = 0;      // VC_SCAN_WRAP / VC_SCAN_DIAG (diagnostic build only; results wrong by design)
  bool sample2_set = false;                   // VC_SAMPLE2: size of the second bootstrap stage
  uint64_t sample2 = 0;
  bool sample1_set = false;                   // VC_SAMPLE1: size of the first bootstrap stage
  uint64_t sample1 = 0;
  bool shape_set = false;                     // VC_SCAN_SHAPE "U,BLK,DB"
  int shape_u = 0, shape_blk = 0, shape_db = 0;
  uint32_t sample_blocks_per_cu = 4;          // VC_SAMPLE_BLOCKS_PER_CU (8 -> 4 in round 3: -1.5..6 us per 125 M-code step, two A/B pairs)
  bool recover_trace = false;                 // VC_RECOVER_TRACE
  bool mih_trace = false;                     // VC_MIH_TRACE
  bool device_recover = true;                 // VC_DEVICE_RECOVER=0: ring overflow handled by the host-driven fallback only
  int mih_bcodes = -1;                        // VC_MIH_BCODES: -1 auto, 0 / 1 forced
  int mih_bent = -1;                          // VC_MIH_BENT: id + code records in bucket order (32-bit substrings of 64-bit codes): -1 auto, 0 / 1
  int scan_small = 1;                         // VC_SCAN_SMALL=0: the general (LDS-streamed) query loop for every tile size
  uint64_t mih_budget = 0;                    // VC_MIH_BUDGET: probes per query run inside mih_query_kernel (0 = automatic)
  int resident_mb = -1;                       // VC_SCAN_RESIDENT_MB: database prefix kept in the Infinity Cache by the verify pass (-1 = default)
  bool scan_trace = false;                    // VC_SCAN_TRACE=1 (diagnostic build): per-block start / end times of the verify kernel
  int mih_host_loop = 0;                      // VC_MIH_HOST_LOOP=1: one host round trip per shell (the round-1 loop)
  int mih_phases = 0;                         // VC_MIH_PHASES=1 (dev): per-phase times of mih_query_kernel on stderr
  int mih_switch = 1;                         // VC_MIH_SWITCH=0: the exact k-NN loop never switches to the verify kernel; 2: always (tests)
  int mih_stream = 1;                         // VC_MIH_STREAM=0: radius search over <= 16-bit substrings through the per-shell probe kernels
  int tau_fold = 0;                           // VC_TAU_FOLD=1: small tiles cut the bootstrap histograms in the verify prologue instead of a
                                              // vc_tau_init_kernel launch (measured: the 1024 prologues cost 20 us, the launch 5 -- off)
  int mih_poll = 1;                           // VC_MIH_POLL=0: wait for the query kernel's counters with hipStreamSynchronize instead of polling
  int mih_lines = -1;                         // VC_MIH_LINES: directory lines of the 32-bit tables (VcTableView::lines): -1 auto, 0 / 1
  int mih_qtile = 0;                          // VC_MIH_QTILE: most queries of one mih_query_kernel launch (0 = default 16384; 4096 .. 65536)
  int mih_order = 1;                          // VC_MIH_ORDER=0: mih_query_kernel's blocks take the queries in batch order (no longest-first pre-pass, mih_order_kernel)
  uint32_t timing_every = 1;                  // not an environment knob: vc_config.timing_sample under VC_FLAG_LEAN_TIMING (set by vc_create) --
                                              // the MIH kernels' launches are bracketed by events only every N-th time as well
  int mih_group = 0;                          // VC_MIH_GROUP=1..3: shells sharing the query kernel's first pass (0 = adaptive)
  uint32_t recover_spin_limit = 0;            // VC_RECOVER_SPIN_LIMIT: bound of the recovery grid barrier's spin (0 = default, ~3 s)
  uint32_t recover_test_fail = 0;             // VC_RECOVER_TEST_FAIL=N (tests): the first N recover launches wait for a block that never comes
};

// ---- vc_scan.hip ------------------------------------------------------------------------------
// Scan-kernel shape chosen per call: BLK threads, U column loads per thread per chunk.
struct VcScanShape {
  int blk;      // 256 or 512
  int unroll;   // U
  int dbuf;     // register buffers per lane: 1 (rely on other waves), 2 (prefetch next chunk), 3 (two chunks ahead)
  int small;    // tiles of <= 8 queries use the compile-time-unrolled form of the kernel
  uint64_t chunk_items() const { return 2ull * blk * unroll; }
};
VcScanShape vc_scan_pick_shape(uint32_t W, uint32_t qt, size_t* lds_bytes, const VcKnobs* knobs, uint64_t n_items = 0);

float vc_probe_stream_ms(const uint64_t* cols, uint64_t stride, uint32_t W, uint64_t items, uint64_t* d_sink, uint32_t n_cu,
                         hipStream_t s);
hipError_t vc_launch_fill_synth(uint64_t* cols, uint64_t stride, uint32_t W, uint64_t first_local, uint64_t n,
                                uint64_t first_gid, uint64_t seed, uint32_t kind, uint32_t n_centres,
                                uint32_t max_flips, hipStream_t s);
hipError_t vc_launch_rows_to_cols(const uint64_t* rows, uint64_t* cols, uint64_t stride, uint32_t W,
                                  uint64_t first_local, uint64_t n, hipStream_t s);
hipError_t vc_launch_gather_rows(const uint64_t* cols, uint64_t stride, uint32_t W, const uint32_t* d_local_ids,
                                 uint32_t n_ids, uint64_t* d_rows, hipStream_t s);
// One stage of the threshold bootstrap (two launches): histogram of the first s_items codes (refine: only
// distances <= tau[q]), then tau[q] = k-th smallest sampled distance.  d_shist [qt][hist_stride] must be zero.
hipError_t vc_launch_sample_hist(const uint64_t* cols, uint64_t stride, uint32_t W, uint64_t s_items,
                                 const uint64_t* d_queries, uint32_t qt, uint32_t* d_shist, uint32_t hist_stride,
                                 uint32_t k, uint32_t bits, uint32_t* d_tau, uint32_t qs, bool refine, uint32_t n_cu,
                                 uint32_t blocks_per_cu, hipStream_t s, bool cut = true);
bool vc_scan_is_small(uint32_t W, uint32_t qt, const VcKnobs* knobs, uint64_t n_items = 0);
// grid = min(chunks, CUs x resident blocks per CU, want_blocks if non-zero)
// shape_n: 0, or the database size when the caller wants the tile shape fitted to a small database (vc_scan_pick_shape; the
// caller's nchunks must come from the same shape)
hipError_t vc_launch_scan(const VcScanParams& p, uint32_t W, uint32_t n_cu, uint32_t want_blocks, const VcKnobs* knobs,
                          hipStream_t s, uint64_t shape_n = 0);
// ring -> sorted top-k (per query); out padded with VC_PACK_INF
// d_tau (nullable): final per-query distance thresholds of the scan -- farther entries are dropped before sorting
hipError_t vc_launch_select_ring(const uint64_t* d_buf, uint32_t cap, const uint32_t* d_count, const uint32_t* d_tau, uint32_t qs,
                                 uint32_t nq, uint32_t k, uint64_t* d_out, uint32_t* d_out_count, hipStream_t s);
// same, for the ring slots named in d_list (outputs indexed by slot)
hipError_t vc_launch_select_ring_list(const uint64_t* d_buf, uint32_t cap, const uint32_t* d_count, const uint32_t* d_list,
                                      uint32_t n_list, uint32_t k, uint64_t* d_out, uint32_t* d_out_count, hipStream_t s);
// Exact device-side recovery of the rows whose ring overflowed (count > cap), after vc_launch_select_ring on the same
// buffers: no-op launch when nothing overflowed.  d_scratch: vc_recover_scratch_words() words, zero at first use
// (the kernel restores its barrier words itself).  nq <= 64.  clean_copies > 0: as the last kernel of the step it also
// zeroes the step's state for these queries (ring cursors d_count, thresholds d_clean_tau, d_hist, and clean_copies
// partial histograms d_clean_shist + c * clean_copy_stride), so that the next step needs no memset.
size_t vc_recover_scratch_words();
hipError_t vc_launch_recover(const uint64_t* cols, uint64_t stride, uint64_t n, uint32_t W, uint32_t id_base, uint32_t bits,
                             const uint64_t* d_queries, uint32_t nq, uint32_t k, uint64_t* d_ring, uint32_t cap,
                             const uint32_t* d_count, const uint32_t* d_hist, uint32_t hist_stride, uint32_t qs, uint32_t* d_scratch,
                             uint64_t* d_out, uint32_t* d_out_count, uint32_t* d_clean_tau, uint32_t* d_clean_shist,
                             uint64_t clean_copy_stride, uint32_t clean_copies, uint32_t n_cu, uint32_t spin_limit, uint32_t absent,
                             hipStream_t s);
// offset (words) of the three barrier lines (followed by the give-up counter line) inside the recovery scratch
size_t vc_recover_barrier_offset_words();
// ---- vc_sort.hip: the index builder's primitives (hand-written; no device library is linked) --------------------
// exclusive prefix sum of L uint32 (in place allowed); d_work: vc_scan_work_words(L) words
size_t vc_scan_work_words(uint64_t L);
hipError_t vc_exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, uint64_t L, uint32_t* d_work, hipStream_t s);
// stable LSD radix sort of n (key, value) pairs by the low key_bits bits of the key (8-bit digits).  Input in
// (keys[0], vals[0]); the passes ping-pong between the two buffer pairs and the result is in pair
// (vc_radix_sort_passes(key_bits) & 1).  d_work: vc_radix_sort_work_words(n) words.
uint32_t vc_radix_sort_passes(uint32_t key_bits);
size_t vc_radix_sort_work_words(uint64_t n);
hipError_t vc_radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint64_t n, uint32_t key_bits, uint32_t* d_work, hipStream_t s);
// n_lists x [nq][k] sorted lists -> merged top-k
hipError_t vc_launch_select_lists(const uint64_t* d_lists, uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* d_out,
                                  uint32_t* d_out_count, hipStream_t s);
// the same over the gathered shard slots of vc_sharded_* (rows + counts per slot; a flagged shard row flags the merged row)
hipError_t vc_launch_select_slots(const uint64_t* d_base, uint64_t slot_words, uint32_t cnt_off_words, uint32_t n_lists, uint32_t nq,
                                  uint32_t k, uint64_t* d_out, uint32_t* d_out_count, hipStream_t s);
