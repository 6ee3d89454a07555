/* ============================================================================
 * verticut_gpu.h -- C ABI of the MI355X (gfx950) Hamming k-NN engine.
 *
 * Drop-in boundary for VertiCut's search hot path.  Every entry point names the reference
 * interface it replaces (paths relative to the reference tree).  Plain pointers and sizes
 * only: no C++ types, no torch types, no exceptions, no globals; one engine handle is
 * thread-compatible (one caller at a time, like SearchWorker, search_worker.h:35-50).
 *
 * Data model (reference semantics kept):
 *   code      B/8 raw bytes, B in {64,128,256,512}  (N_BINARY_BITS image_search_constants.h:10)
 *   id        uint32 = ordinal of the record in insertion order (+ id_base of the shard)
 *             (build_hash_tables.cc:55,61,69; image_search.proto:4,17)
 *   result    uint64 = id | (uint64)dist << 32      (search_worker.cc:12-13,254-256)
 *   table t   substring t = bytes [t*B/8/m, (t+1)*B/8/m) of the code, key = little-endian
 *             value (Pilaf/image_tools.h:12-18), one table per former MPI rank
 *             (search_worker.cc:99-101, build_hash_tables.cc:26,36-38)
 *
 * All functions return VC_OK (0) or a negative VC_ERR_* code; vc_last_error() gives text.
 * There is no CPU fallback anywhere behind this ABI: without a usable gfx950 device
 * vc_create fails with VC_ERR_NO_DEVICE.
 * ==========================================================================*/
#ifndef VERTICUT_GPU_H
#define VERTICUT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_ABI_VERSION 2

/* ---- status codes (BaseProxy uses 0 = found/done, 1 = not found/fail: base_proxy.h:10-13) */
#define VC_OK 0
#define VC_NOT_FOUND 1          /* == PROXY_NOT_FOUND, only from vc_get_bucket / vc_get_code */
#define VC_ERR_INVALID (-1)     /* bad argument / configuration (reference: assert, search_worker.cc:75) */
#define VC_ERR_NO_DEVICE (-2)   /* no gfx950 device / HIP runtime failure at create */
#define VC_ERR_HIP (-3)         /* HIP call failed (text in vc_last_error) */
#define VC_ERR_NOMEM (-4)       /* device allocation failed */
#define VC_ERR_STATE (-5)       /* call order: e.g. MIH search before vc_build_index */
#define VC_ERR_CAPACITY (-6)    /* more codes than vc_config.capacity, or output buffer too small */

/* ---- search modes (which reference loop the call reproduces) */
#define VC_MODE_LINEAR 0      /* linear_search.cc:39-64  full scan + top-k            */
#define VC_MODE_MIH_EXACT 1   /* search_worker.cc:159-218 search_K_nearest_neighbors   */
#define VC_MODE_MIH_APPROX 2  /* search_worker.cc:93-157  ..._approximate_... (factor 20, search_worker.h:14) */

/* ---- vc_config.flags */
#define VC_FLAG_USE_BITMAP 0x1u        /* attach the bucket-occupancy bitmap (search_worker.cc:238-243);
                                          off = as shipped (:61-62), n_local_reads stays 0 */
#define VC_FLAG_REF_SIGNEXT_KEYS 0x2u  /* reproduce binaryToInt's sign-extended keys for substrings < 32 bit
                                          (Pilaf/image_tools.h:13): probes that flip the substring's top bit
                                          can never match, exactly as in the reference.  Off = masked keys
                                          (exact MIH for every substring width). */
#define VC_FLAG_REF_STOP_LITERAL4 0x4u /* stop rule "kth <= 4*radius" with the literal 4 (search_worker.cc:204)
                                          even when n_tables < 4.  Default: min(n_tables,4), identical to the
                                          reference whenever the reference itself is exact. */

#define VC_FLAG_LEAN_TIMING 0x8u       /* time only the verify-kernel launches (vc_timing.scan_*): no event pair around each
                                          search call, so vc_timing.total_ms / calls stay 0.  Each event record is a barrier
                                          packet in the stream (~4 us); a throughput loop that keeps its own clock sets this. */

/* ---- synthetic data kinds for vc_add_synthetic (the reference ships no data: .gitignore:7-8) */
#define VC_SYNTH_UNIFORM 0
#define VC_SYNTH_CLUSTERED 1

/* ---- streams: every `stream` argument is a hipStream_t; NULL is the HIP null (legacy default) stream -- what
 * PyTorch-ROCm's default stream is -- and VC_STREAM_OWN names the engine's private non-blocking stream, on which
 * the host-pointer calls run unless vc_set_stream says otherwise. */
#define VC_STREAM_OWN ((void*)(intptr_t)-1)

/* ---- output order for k-NN results */
#define VC_ORDER_ASCENDING 0       /* canonical: ascending packed (dist, id) */
#define VC_ORDER_FARTHEST_FIRST 1  /* as SearchWorker::find / linear_search print (search_worker.cc:210-216) */

typedef struct vc_engine vc_engine;

typedef struct vc_config {
  uint32_t abi_version;  /* VC_ABI_VERSION */
  uint32_t bits;         /* code width B: 64, 128, 256 or 512            (args_config.cc binary_bits) */
  uint32_t n_tables;     /* m, substring = B/m bits, 8..32, multiple of 8 (args_config.cc n_tables);
                            0 = linear-only engine */
  uint32_t flags;        /* VC_FLAG_* */
  uint64_t capacity;     /* max number of codes this engine (shard) will hold (image_total) */
  uint32_t id_base;      /* global id of local record 0 (shard offset; ids stay < 2^32) */
  int32_t device;        /* HIP device ordinal, -1 = current */
  uint32_t cand_cap;     /* per-query candidate ring entries, 0 = default (65536) */
  uint32_t scan_blocks;  /* 0 = default grid for the verify kernel (tuning knob) */
  uint32_t query_tile;   /* queries verified per DB pass; 0 = the engine chooses: 32 for databases of 256 MB and more, doubling
                            per halving below (at most 512: a small database's pass is priced by its launches, not its bytes).
                            8 = HBM-bound pass, see DESIGN.md 4.1 */
  uint32_t timing_sample;/* with VC_FLAG_LEAN_TIMING: time only every N-th verify launch (0/1 = every launch) */
  uint32_t reserved[4];
} vc_config;

/* Per-query statistics == SearchWorker::get_stat (search_worker.cc:24-30, search_worker.h:42-45). */
typedef struct vc_query_stats {
  uint32_t radius;        /* last substring shell searched (find's return value radius-1) */
  uint32_t n_results;     /* results written for this query (<= k) */
  uint64_t n_main_reads;  /* always 0 (never incremented in the reference) */
  uint64_t n_sub_reads;   /* bucket gets issued by table 0 (what rank 0's get_stat reports) */
  uint64_t n_local_reads; /* bitmap tests by table 0 (0 unless VC_FLAG_USE_BITMAP) */
  uint64_t n_candidates;  /* distinct DB items verified (owner-rule deduplicated) */
} vc_query_stats;

/* Device-side timing of every search call since the previous vc_get_timing(), measured with HIP events
 * recorded on the stream the kernels were launched on. */
typedef struct vc_timing {
  float total_ms;          /* sum over calls of the whole call's device span */
  float scan_ms;           /* sum over launches of the dominant verify kernel (vc_scan_kernel) */
  uint32_t scan_launches;
  uint32_t calls;
  uint64_t scan_bytes;     /* algorithmic bytes those launches read: launches * N * B/8 */
  /* ABI 2 -- the MIH query kernel (mih_query_kernel: probe + verify + merge of the shells of a batch in one launch) */
  float mih_ms;            /* sum over its launches */
  uint32_t mih_launches;
  uint64_t mih_queries;    /* queries those launches served */
  uint64_t mih_probes;     /* bucket probes = keys enumerated (search_worker.cc:230-264 leaves), all tables */
  uint64_t mih_hits;       /* non-empty buckets looked up (PROXY_FOUND gets, search_worker.cc:246) */
  uint64_t mih_entries;    /* bucket entries verified (search_worker.cc:249-257) */
} vc_timing;

/* ---- lifetime ----------------------------------------------------------------------------
 * replaces: SearchWorker ctor (search_worker.cc:50-63) + BaseProxy::init/close (base_proxy.h:24-28) */
int vc_create(const vc_config* cfg, vc_engine** out);
int vc_destroy(vc_engine* e);
const char* vc_last_error(const vc_engine* e); /* never NULL; e may be NULL for create failures */
const char* vc_strerror(int code);
int vc_abi_version(void);

/* ---- ingest ------------------------------------------------------------------------------
 * replaces: build_hash_tables.cc:40-70 (records appended in file order, id = ordinal) and the
 * ID -> BinaryCode put path used by linear_search.cc:45-46.  `codes` = n * bits/8 raw bytes (host). */
int vc_add_codes(vc_engine* e, const void* codes, uint64_t n);
/* Same, generated on the device: item with global id g gets the code of the shared definition
 * (oracle/vc_oracle.cc gen_one) -- used by bench/tests for the BASELINE.json shapes. */
int vc_add_synthetic(vc_engine* e, uint64_t n, uint64_t seed, uint32_t kind, uint32_t n_centres, uint32_t max_flips);
int vc_size(const vc_engine* e, uint64_t* n);
/* The reference's on-disk inputs, honoured as they are:
 *   code file   headerless records of bits/8 bytes, id = ordinal  (build_hash_tables.cc:40-70, BINARY_CODE_FILE)
 *   bitmap file raw 2^substr_bits-bit LSB-first uint32 words of one table (generate_bitmap.cc:99-125,
 *               read back by bitmap_deamon.cc:41-65)
 * vc_load_code_file appends up to max_records records (0 = all) and reports how many were read;
 * vc_save_code_file writes the resident records back in id order; vc_write_bitmap_file needs vc_build_index. */
int vc_load_code_file(vc_engine* e, const char* path, uint64_t max_records, uint64_t* n_read);
int vc_save_code_file(vc_engine* e, const char* path);
int vc_write_bitmap_file(vc_engine* e, uint32_t table, const char* path);
/* Reads a bitmap file of that format (what bitmap_deamon.cc:41-65 loads) and checks it word for word against the
 * bitmap of the resident index: VC_OK = identical (the file belongs to this database), VC_ERR_STATE = it differs
 * (*n_mismatch_words, may be NULL, says in how many 32-bit words), VC_ERR_INVALID = wrong size / unreadable. */
int vc_read_bitmap_file(vc_engine* e, uint32_t table, const char* path, uint64_t* n_mismatch_words);
/* Index persistence (the step build_hash_tables.cc performs against the KV tier, kept on disk instead): the sorted id
 * runs, bucket offsets, occupancy bitmaps and rank directories of every table.  vc_load_index needs the same records
 * resident (vc_load_code_file / vc_add_codes) and refuses a file built for another shape (VC_ERR_STATE). */
int vc_save_index(vc_engine* e, const char* path);
int vc_load_index(vc_engine* e, const char* path);
/* ID -> BinaryCode get (linear_search.cc:45-46; by-id query path image_search_client.h:23-25).
 * id is a global id; out = bits/8 bytes.  VC_NOT_FOUND if id is not in this shard. */
int vc_get_code(vc_engine* e, uint32_t id, void* out);

/* ---- index -------------------------------------------------------------------------------
 * replaces: build_hash_tables.cc (bucket contents, rule a12) + generate_bitmap.cc:105-114
 * (bit v of table t set iff bucket (t,v) non-empty).  Must follow the last vc_add_*. */
int vc_build_index(vc_engine* e);
/* HashIndex{table_id,index} -> Image_List get (search_worker.cc:224-246, base_proxy.h:18).
 * Writes up to cap (id, code) pairs in append (= id) order; *n = bucket length.
 * Returns VC_OK (PROXY_FOUND) or VC_NOT_FOUND.  ids / codes may be NULL. */
int vc_get_bucket(vc_engine* e, uint32_t table, uint32_t index, uint32_t* ids, void* codes, uint32_t cap, uint32_t* n);
/* ImageBitmap::get_idx (bitmap.cc:22-26) on table t's occupancy bitmap. *bit = 0/1. */
int vc_bitmap_test(vc_engine* e, uint32_t table, uint32_t index, int* bit);
/* Raw bitmap words (uint32, LSB-first, 2^substr_bits bits) as generate_bitmap.cc:122-125 writes them;
 * copies [word_off, word_off+n_words) to host memory. */
int vc_bitmap_read(vc_engine* e, uint32_t table, uint64_t word_off, uint64_t n_words, uint32_t* out);

/* ---- search ------------------------------------------------------------------------------
 * replaces: SearchWorker::find (search_worker.cc:65-89) / linear_search.cc:39-64 for a batch of
 * nq queries (nq * bits/8 raw bytes, host memory).
 *   out    nq * k packed results; query i owns out[i*k .. i*k+counts[i])
 *   counts nq entries (may be NULL): results found (< k only if the DB holds fewer items)
 *   stats  nq entries or NULL
 * Result set = the k smallest (dist, id) pairs among the items the mode's loop has seen
 * (LINEAR: all items).  Ties at the k-th distance resolve to the smallest ids.
 * Host memory: any.  Results of up to 512 KB cross in one copy through a pinned staging buffer of the engine; larger ones go
 * out by DMA directly when `out` is page-locked (hipHostMalloc / hipHostRegister: detected per call), else through two pinned
 * chunks with the DMA of one overlapping the host copy of the other -- a 16 384-query top-100 call returns 13 MB of rows:
 * 15 M queries/s into page-locked memory, 11 M into pageable (1e8 records, exact MIH).  `stats` costs the MIH modes four small
 * read-backs per launch and is skipped when NULL. */
int vc_search_knn(vc_engine* e, const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order,
                  uint64_t* out, uint32_t* counts, vc_query_stats* stats);
/* Device-pointer variant for callers that keep queries/results in HBM (torch / multi-GPU merge).
 * d_queries: nq*bits/8 bytes; d_out: nq*k uint64, ascending, padded with UINT64_MAX; d_counts: nq uint32.
 * Asynchronous on `stream` and ordered like any other work enqueued there when mode == VC_MODE_LINEAR; the MIH
 * modes make the host wait until the batch's query kernel has finished (how many queries continue in the multi-block
 * shells decides what is enqueued next), so their rows are complete when the call returns; results are valid in
 * stream order in every mode.
 * Batch size, MIH modes: a call's queries run in launches of up to 16 384 (VC_MIH_QTILE) and a launch ends with its longest
 * query, i.e. carries a fixed tail of 60-90 us -- exact top-100 over 1e8 records: 14.4 M queries/s in calls of 4 096 queries,
 * 20.7 M in calls of 16 384; the radius search the same way (1 024 -> 4 096 queries per call: + 10 %).
 * A candidate-ring overflow (more than cand_cap items at or below the k-th distance: duplicate-heavy data,
 * linear_search.cc:113-117 mentions 250 000-entry buckets) is recovered exactly ON THE DEVICE in the same stream
 * (radix select over the position of the tied items), so a LINEAR row is always exact.  Only if that
 * recovery gives up (see vc_device_status) a query reports d_counts[i] == UINT32_MAX with an upper-bound row. */
int vc_search_knn_dev(vc_engine* e, const void* d_queries, uint32_t nq, uint32_t k, uint32_t mode,
                      uint64_t* d_out, uint32_t* d_counts, void* stream);
/* Same, plus SearchWorker::get_stat (search_worker.cc:24-30) for callers that stay on the device: d_stats (device
 * memory, nq records, may be NULL) is written by a kernel in stream order -- no host read-back, no extra wait.
 * LINEAR: n_candidates = records scanned, everything else 0.  n_results = d_counts[i]. */
int vc_search_knn_dev_stats(vc_engine* e, const void* d_queries, uint32_t nq, uint32_t k, uint32_t mode,
                            uint64_t* d_out, uint32_t* d_counts, vc_query_stats* d_stats, void* stream);
/* All items within full Hamming distance <= radius of each query (BASELINE config 2; built from
 * search_R_neighbors shells 0..radius/m, search_worker.cc:222-227).  mode LINEAR or MIH_EXACT.
 * out_offsets: nq+1 entries; results of query i at out[out_offsets[i] .. out_offsets[i+1]),
 * ascending packed.  VC_ERR_CAPACITY (with out_offsets filled with the needed counts) if out_cap is too small. */
int vc_search_radius(vc_engine* e, const void* queries, uint32_t nq, uint32_t radius, uint32_t mode,
                     uint64_t* out, uint64_t out_cap, uint64_t* out_offsets);

/* Device-pointer variant (queries and results stay in HBM): d_queries nq*bits/8 bytes, d_out out_cap packed values,
 * d_offsets nq+1 entries, all device memory.  The work is enqueued on `stream`; the host waits once, at the end, for the
 * total (and repeats the call internally with a larger work ring if a query outgrew it).  The copy into d_out may still be
 * running on `stream` when the call returns: the results are valid in stream order, a host reader copies behind it.
 * VC_ERR_CAPACITY if the results do not fit out_cap (d_offsets then holds the needed counts).
 * replaces: the same reference loop as vc_search_radius (search_worker.cc:222-264). */
int vc_search_radius_dev(vc_engine* e, const void* d_queries, uint32_t nq, uint32_t radius, uint32_t mode,
                         uint64_t* d_out, uint64_t out_cap, uint64_t* d_offsets, void* stream);

/* Sticky status of the asynchronous device path: *n_gave_up = calls since the previous vc_device_status() in which the
 * device-side ring-overflow recovery could not complete (its grid never met: the GPU was held by other kernels for
 * seconds); the affected queries kept d_counts[i] == UINT32_MAX.  0 in normal operation.  Synchronises the stream.
 * replaces: nothing in the reference (its find() is synchronous, search_worker.cc:65-89). */
int vc_device_status(vc_engine* e, uint32_t* n_gave_up);

/* ---- multi-GPU merge ---------------------------------------------------------------------
 * replaces: mpi_coordinator::gather_vectors + master-side heap (mpi_coordinator.cc:34-69,
 * search_worker.cc:179-199).  d_lists holds n_lists blocks of nq*k packed values (the all-gathered
 * per-shard top-k, UINT64_MAX padded); writes the merged ascending top-k to d_out (nq*k) and the
 * valid count per query to d_counts (may be NULL).  Asynchronous on `stream`; no engine needed. */
int vc_merge_topk_dev(const uint64_t* d_lists, uint32_t n_lists, uint32_t nq, uint32_t k,
                      uint64_t* d_out, uint32_t* d_counts, void* stream);

/* ---- one process, several GPUs --------------------------------------------------------------
 * replaces: the reference's distribution of this path -- `mpirun -n 4` ranks (run_distributed_search.py:74), the
 * per-radius MPI_Gather / Gatherv / Bcast between them (search_worker.cc:99-101,177,207; mpi_coordinator.cc:26-69)
 * and the master-side dedup + heap (search_worker.cc:179-199).
 * The database is split BY ID RANGE into n_shards shards (shard g holds ids [capacity*g/G, capacity*(g+1)/G) of the
 * id space that starts at engine.id_base; shard g lives on device_ids[g % n_devices]); every shard answers the whole
 * batch for its ids and the per-shard top-k rows (nq*k*8 bytes per shard -- the only inter-GPU traffic) are brought
 * together by ONE exchange per batch and merged by the kernel behind vc_merge_topk_dev on the first device:
 *   VC_EXCHANGE_RCCL       grouped ncclAllGather over xGMI (single-process ncclCommInitAll; one shard per device)
 *   VC_EXCHANGE_PEER_COPY  hipMemcpyPeerAsync into the root's gather buffer (also when shards share a device)
 *   VC_EXCHANGE_AUTO       RCCL when there is one shard per device, more than one device and librccl loads; else peer copy
 * LINEAR results are exactly those of one engine holding everything (the top-k of a union is the top-k of the parts'
 * top-k).  MIH modes: every shard runs to its OWN stop rule, which is exact for the shard, so exact-mode distances
 * equal the single-engine result; statistics report the widest radius and the summed reads / candidates. */
#define VC_MAX_SHARDS 16
#define VC_EXCHANGE_AUTO 0
#define VC_EXCHANGE_PEER_COPY 1
#define VC_EXCHANGE_RCCL 2

typedef struct vc_sharded vc_sharded;
typedef struct vc_sharded_config {
  uint32_t abi_version;            /* VC_ABI_VERSION */
  uint32_t n_shards;               /* G: 1..VC_MAX_SHARDS (the reference's `size`, search_worker.cc:58) */
  uint32_t n_devices;              /* entries of device_ids; 0 = devices 0..min(visible, n_shards)-1 */
  uint32_t exchange;               /* VC_EXCHANGE_* */
  int32_t device_ids[VC_MAX_SHARDS];
  vc_config engine;                /* per-shard template: bits, n_tables, flags, cand_cap, query_tile ...; capacity = TOTAL
                                      records over all shards, id_base = global id of record 0; `device` is ignored */
} vc_sharded_config;

int vc_sharded_create(const vc_sharded_config* cfg, vc_sharded** out);
int vc_sharded_destroy(vc_sharded* h);
const char* vc_sharded_last_error(const vc_sharded* h);     /* h may be NULL for create failures */
int vc_sharded_exchange(const vc_sharded* h, uint32_t* kind); /* the exchange in use: VC_EXCHANGE_PEER_COPY or _RCCL */
/* ingest in global id order (build_hash_tables.cc:40-70), routed to the shard that owns the id */
int vc_sharded_add_codes(vc_sharded* h, const void* codes, uint64_t n);
int vc_sharded_add_synthetic(vc_sharded* h, uint64_t n, uint64_t seed, uint32_t kind, uint32_t n_centres, uint32_t max_flips);
int vc_sharded_size(const vc_sharded* h, uint64_t* n);
int vc_sharded_build_index(vc_sharded* h);
/* ID -> BinaryCode and HashIndex -> Image_List over all shards (a bucket = the shards' buckets in id order) */
int vc_sharded_get_code(vc_sharded* h, uint32_t id, void* out);
int vc_sharded_get_bucket(vc_sharded* h, uint32_t table, uint32_t index, uint32_t* ids, void* codes, uint32_t cap, uint32_t* n);
/* SearchWorker::find / linear_search for a batch over all shards; arguments as vc_search_knn */
int vc_sharded_search_knn(vc_sharded* h, const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order,
                          uint64_t* out, uint32_t* counts, vc_query_stats* stats);
/* Device-resident, stream-ordered form: d_queries (nq*bits/8 bytes), d_out (nq*k), d_counts (nq, may be NULL) and
 * d_stats (nq records, may be NULL) are device memory on the ROOT device (vc_sharded_root_device: the first shard's),
 * `stream` a stream of that device (NULL = its null stream, VC_STREAM_OWN = the handle's own).  The queries reach every
 * other device by one peer copy, the shards of a device run one after the other on that device's stream and the devices
 * concurrently, rows + counts + statistics of a shard travel as one slot (one peer copy per remote shard or one grouped
 * ncclAllGather), the merge kernel reduces the shards' overflow flags on the device (a row whose shard-side recovery
 * gave up reports d_counts[i] == UINT32_MAX, exactly as vc_search_knn_dev does) and a small kernel the statistics.
 * LINEAR: nothing is waited for on the host -- results are valid in stream order.  MIH modes: as in vc_search_knn_dev the
 * host waits inside every shard for its query kernel (lanes of different devices then run on host threads).
 * replaces: search_worker.cc:99-101,177,207 + mpi_coordinator.cc:34-69 for callers that keep the batch in HBM. */
int vc_sharded_search_knn_dev(vc_sharded* h, const void* d_queries, uint32_t nq, uint32_t k, uint32_t mode,
                              uint64_t* d_out, uint32_t* d_counts, vc_query_stats* d_stats, void* stream);
int vc_sharded_root_device(const vc_sharded* h, int* device);
/* All items within full Hamming distance <= radius of each query over all shards; arguments, result layout and
 * VC_ERR_CAPACITY behaviour as vc_search_radius (host buffers).  Every shard searches its id range, the shards' results of a
 * query are brought together and ordered by the radius search's own segment sort on the root device.
 * replaces: search_R_neighbors on every rank + gather_vectors + the master-side dedup (search_worker.cc:177-199,222-264). */
int vc_sharded_search_radius(vc_sharded* h, const void* queries, uint32_t nq, uint32_t radius, uint32_t mode,
                             uint64_t* out, uint64_t out_cap, uint64_t* out_offsets);
/* borrow shard g's engine (bucket views, timing, files); its id range is [*first_id, *first_id + *n_ids) */
int vc_sharded_shard(vc_sharded* h, uint32_t shard, vc_engine** e, uint64_t* first_id, uint64_t* n_ids);

/* ---- measurement ------------------------------------------------------------------------- */
/* Sums and resets the event records (synchronises with the last recorded call). */
int vc_get_timing(const vc_engine* e, vc_timing* t);
/* Stream of the host-pointer calls (default VC_STREAM_OWN). */
int vc_set_stream(vc_engine* e, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VERTICUT_GPU_H */
