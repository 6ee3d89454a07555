// ============================================================================
// vc_sharded.hip -- the vc_sharded_* family of include/verticut_gpu.h: ONE process drives the GPUs of a node.
//
// Replaces the reference's distribution layer for this path:
//   src/search_worker.cc:99-101,177,207  per-radius MPI_Gather / Gatherv / Bcast between the ranks
//   src/mpi_coordinator.cc:34-69         gather_vectors (variable-length concat on rank 0)
//   src/search_worker.cc:179-199         master-side dedup map + heap
// by: the database split BY ID RANGE into shards (one vc_engine each, shard g on device_ids[g % n_devices]); every
// shard answers the whole query batch for its ids; the per-shard top-k rows -- nq * k * 8 bytes per shard, the only
// inter-GPU traffic of a batch -- are brought together by ONE exchange and merged by vc_merge_topk_dev:
//   VC_EXCHANGE_RCCL       single-process ncclCommInitAll over the devices + a grouped ncclAllGather on the shards'
//                          own streams (one shard per device); librccl is opened lazily with dlopen, so the library
//                          has no link-time dependency on it and a process that already carries RCCL (PyTorch)
//                          keeps its copy
//   VC_EXCHANGE_PEER_COPY  hipMemcpyPeerAsync of each shard's rows into the root device's gather buffer, ordered
//                          behind the shard's search by an event; also the path when several shards share a device
// No CPU compute anywhere: the host moves pointers, the merge is the device kernel of vc_scan.hip.
// ============================================================================
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "vc_internal.hpp"
#include "vc_mih.hpp"

namespace {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string* why) {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      if (why) *why = std::string("librccl not loadable: ") + dlerror();
      return false;
    }
#define VC_SYM(field, sym)                                   \
  field = (decltype(field))dlsym(lib, sym);                  \
  if (!field) {                                              \
    if (why) *why = std::string("librccl lacks ") + sym;     \
    dlclose(lib);                                            \
    lib = nullptr;                                           \
    return false;                                            \
  }
    VC_SYM(CommInitAll, "ncclCommInitAll")
    VC_SYM(CommDestroy, "ncclCommDestroy")
    VC_SYM(AllGather, "ncclAllGather")
    VC_SYM(GroupStart, "ncclGroupStart")
    VC_SYM(GroupEnd, "ncclGroupEnd")
    VC_SYM(GetErrorString, "ncclGetErrorString")
#undef VC_SYM
    return true;
  }
};

// One lane per distinct device: the shards that live there run ONE AFTER THE OTHER on the lane's stream -- at most one
// persistent-grid kernel sequence per GPU (the recovery kernel's grid barrier needs its whole grid resident; a neighbour
// shard's full-chip verify grid on another stream could keep it off the CUs for seconds: ADVICE round 3) -- while the
// lanes, i.e. the devices, run concurrently.  The root device's lane runs on the caller's stream.
struct Lane {
  int dev = 0;
  std::vector<uint32_t> shards;
  hipStream_t stream = nullptr;                   // own stream (the root lane uses the caller's)
  hipEvent_t done = nullptr;                      // behind the lane's last shard of a batch
  void* q = nullptr;        size_t q_bytes = 0;   // the batch's queries on this device (peer copy from the root)
  uint64_t* gath = nullptr; size_t gath_bytes = 0;   // [G][slot]: shard g's rows | counts | statistics at slot g; on the root
                                                     // (and, with RCCL, everywhere) the gathered results of all shards
};

}  // namespace

struct vc_sharded {
  vc_sharded_config cfg;
  uint32_t G = 0, D = 0, nbytes = 0;
  uint64_t capacity = 0, n = 0;
  std::vector<vc_engine*> eng;
  std::vector<int> dev;                 // device of shard g
  std::vector<uint64_t> lo, hi;         // ids [lo, hi) relative to cfg.engine.id_base
  std::vector<Lane> lanes;
  std::vector<uint32_t> lane_of;        // shard -> lane
  uint32_t root_lane = 0;
  int root = 0;                         // device that merges
  hipStream_t root_stream = nullptr;    // stream of the host-pointer calls / VC_STREAM_OWN
  hipEvent_t ev_q = nullptr;            // "the batch's queries are ready" on the caller's stream
  void* d_hq = nullptr;         size_t hq_bytes = 0;      // host-pointer API: staged queries, merged rows, counts, statistics
  uint64_t* d_out = nullptr;    size_t out_bytes = 0;
  uint32_t* d_ocnt = nullptr;   size_t ocnt_bytes = 0;
  vc_query_stats* d_ostats = nullptr; size_t ostats_bytes = 0;
  RcclApi rccl;
  std::vector<ncclComm_t> comms;
  uint32_t exchange = VC_EXCHANGE_PEER_COPY;   // what is in use
  std::string err;
};

static thread_local std::string g_sharded_create_err;

static int sfail(vc_sharded* h, int code, const char* fmt, ...) {
  char b[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(b, sizeof b, fmt, ap);
  va_end(ap);
  if (h) h->err = b; else g_sharded_create_err = b;
  return code;
}

#define VS_HIP(h, call)                                                                                        \
  do {                                                                                                         \
    hipError_t _r = (call);                                                                                    \
    if (_r != hipSuccess)                                                                                      \
      return sfail(h, _r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP, "%s: %s", #call, hipGetErrorString(_r)); \
  } while (0)

template <class T>
static int sgrow(vc_sharded* h, T** p, size_t* have, size_t need) {
  if (need <= *have) return VC_OK;
  if (*p) VS_HIP(h, hipFree(*p));
  *p = nullptr;
  *have = 0;
  need = (need + 255) & ~(size_t)255;
  VS_HIP(h, hipMalloc((void**)p, need));
  *have = need;
  return VC_OK;
}

static uint32_t shard_of(const vc_sharded* h, uint64_t pos) {   // pos relative to id_base, < capacity
  uint32_t g = (uint32_t)std::min<uint64_t>(h->G - 1, pos * h->G / std::max<uint64_t>(h->capacity, 1));
  while (g + 1 < h->G && pos >= h->hi[g]) ++g;
  while (g > 0 && pos < h->lo[g]) --g;
  return g;
}

static uint64_t shard_size(const vc_sharded* h, uint32_t g) {   // records shard g holds (ingest fills the id ranges in order)
  return h->n <= h->lo[g] ? 0 : std::min(h->n, h->hi[g]) - h->lo[g];
}

extern "C" {

const char* vc_sharded_last_error(const vc_sharded* h) { return h ? h->err.c_str() : g_sharded_create_err.c_str(); }

int vc_sharded_destroy(vc_sharded* h) {
  if (!h) return VC_OK;
  for (Lane& l : h->lanes) {
    (void)hipSetDevice(l.dev);
    (void)hipDeviceSynchronize();
  }
  if (!h->comms.empty() && h->rccl.CommDestroy)
    for (ncclComm_t c : h->comms)
      if (c) (void)h->rccl.CommDestroy(c);
  for (Lane& l : h->lanes) {
    (void)hipSetDevice(l.dev);
    (void)hipFree(l.q); (void)hipFree(l.gath);
    if (l.done) (void)hipEventDestroy(l.done);
    if (l.stream) (void)hipStreamDestroy(l.stream);
  }
  for (vc_engine* e : h->eng) vc_destroy(e);
  (void)hipSetDevice(h->root);
  (void)hipFree(h->d_hq); (void)hipFree(h->d_out); (void)hipFree(h->d_ocnt); (void)hipFree(h->d_ostats);
  if (h->ev_q) (void)hipEventDestroy(h->ev_q);
  if (h->root_stream) (void)hipStreamDestroy(h->root_stream);
  delete h;
  return VC_OK;
}

int vc_sharded_create(const vc_sharded_config* cfg, vc_sharded** out) {
  if (!cfg || !out) return sfail(nullptr, VC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != VC_ABI_VERSION) return sfail(nullptr, VC_ERR_INVALID, "abi_version %u != %u", cfg->abi_version, VC_ABI_VERSION);
  if (cfg->n_shards == 0 || cfg->n_shards > VC_MAX_SHARDS) return sfail(nullptr, VC_ERR_INVALID, "n_shards must be 1..%d", VC_MAX_SHARDS);
  if (cfg->n_devices > VC_MAX_SHARDS) return sfail(nullptr, VC_ERR_INVALID, "n_devices must be <= %d", VC_MAX_SHARDS);
  if (cfg->exchange > VC_EXCHANGE_RCCL) return sfail(nullptr, VC_ERR_INVALID, "unknown exchange %u", cfg->exchange);
  if (cfg->engine.capacity == 0 || cfg->engine.capacity + (uint64_t)cfg->engine.id_base > 0x100000000ull)
    return sfail(nullptr, VC_ERR_INVALID, "capacity must be > 0 and id_base + capacity <= 2^32 (ids are uint32)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return sfail(nullptr, VC_ERR_NO_DEVICE, "no HIP device visible");
  vc_sharded* h = new vc_sharded();
  h->cfg = *cfg;
  h->G = cfg->n_shards;
  h->D = cfg->n_devices ? cfg->n_devices : (uint32_t)std::min<int>(ndev, (int)cfg->n_shards);
  h->capacity = cfg->engine.capacity;
  h->nbytes = cfg->engine.bits / 8;
  h->eng.assign(h->G, nullptr);
  h->dev.resize(h->G);
  h->lo.resize(h->G);
  h->hi.resize(h->G);
  h->lane_of.resize(h->G);
  int rc = VC_OK;
  for (uint32_t g = 0; g < h->G && rc == VC_OK; ++g) {
    h->dev[g] = cfg->n_devices ? cfg->device_ids[g % h->D] : (int)(g % h->D);
    h->lo[g] = h->capacity * g / h->G;
    h->hi[g] = h->capacity * (g + 1) / h->G;
    if (h->dev[g] < 0 || h->dev[g] >= ndev) { rc = sfail(nullptr, VC_ERR_NO_DEVICE, "device %d of shard %u is out of range (%d visible)", h->dev[g], g, ndev); break; }
    vc_config ec = cfg->engine;
    ec.capacity = std::max<uint64_t>(h->hi[g] - h->lo[g], 1);
    ec.id_base = cfg->engine.id_base + (uint32_t)h->lo[g];
    ec.device = h->dev[g];
    ec.flags |= VC_FLAG_LEAN_TIMING;
    rc = vc_create(&ec, &h->eng[g]);
    if (rc != VC_OK) { g_sharded_create_err = std::string("shard ") + std::to_string(g) + ": " + vc_last_error(nullptr); break; }
    uint32_t li = 0;
    while (li < h->lanes.size() && h->lanes[li].dev != h->dev[g]) ++li;
    if (li == h->lanes.size()) {
      h->lanes.emplace_back();
      Lane& l = h->lanes.back();
      l.dev = h->dev[g];
      if (hipSetDevice(l.dev) != hipSuccess || hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess)
        rc = sfail(nullptr, VC_ERR_HIP, "device %d: stream / event creation failed", l.dev);
    }
    h->lanes[li].shards.push_back(g);
    h->lane_of[g] = li;
  }
  if (rc == VC_OK) {
    h->root = h->dev[0];
    h->root_lane = h->lane_of[0];
    if (hipSetDevice(h->root) != hipSuccess || hipStreamCreateWithFlags(&h->root_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_q, hipEventDisableTiming) != hipSuccess)
      rc = sfail(nullptr, VC_ERR_HIP, "root stream / event creation failed");
  }
  // peer access root <- every other device (hipMemcpyPeerAsync works without it through a staged copy; with it the
  // copy is one xGMI transfer)
  if (rc == VC_OK)
    for (uint32_t g = 0; g < h->G; ++g)
      if (h->dev[g] != h->root) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, h->root, h->dev[g]) == hipSuccess && can) {
          (void)hipSetDevice(h->root);
          hipError_t r = hipDeviceEnablePeerAccess(h->dev[g], 0);
          if (r != hipSuccess && r != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
      }
  // exchange: RCCL needs one shard per device (a communicator rank is a device)
  if (rc == VC_OK) {
    bool one_per_device = h->G == h->D;
    for (uint32_t g = 0; g < h->G && one_per_device; ++g)
      for (uint32_t g2 = 0; g2 < g; ++g2)
        if (h->dev[g] == h->dev[g2]) one_per_device = false;
    const bool want_rccl = cfg->exchange == VC_EXCHANGE_RCCL || (cfg->exchange == VC_EXCHANGE_AUTO && one_per_device && h->G > 1);
    h->exchange = VC_EXCHANGE_PEER_COPY;
    if (want_rccl) {
      std::string why;
      if (!one_per_device) {
        why = "RCCL exchange needs exactly one shard per device";
      } else if (h->rccl.load(&why)) {
        h->comms.assign(h->G, nullptr);
        ncclResult_t nr = h->rccl.CommInitAll(h->comms.data(), (int)h->G, h->dev.data());
        if (nr == ncclSuccess) h->exchange = VC_EXCHANGE_RCCL;
        else { why = std::string("ncclCommInitAll: ") + h->rccl.GetErrorString(nr); h->comms.clear(); }
      }
      if (h->exchange != VC_EXCHANGE_RCCL && cfg->exchange == VC_EXCHANGE_RCCL) rc = sfail(nullptr, VC_ERR_STATE, "%s", why.c_str());
    }
  }
  if (rc != VC_OK) {
    vc_sharded_destroy(h);
    return rc;
  }
  *out = h;
  return VC_OK;
}

int vc_sharded_exchange(const vc_sharded* h, uint32_t* kind) {
  if (!h || !kind) return VC_ERR_INVALID;
  *kind = h->exchange;
  return VC_OK;
}

int vc_sharded_size(const vc_sharded* h, uint64_t* n) {
  if (!h || !n) return VC_ERR_INVALID;
  *n = h->n;
  return VC_OK;
}

int vc_sharded_shard(vc_sharded* h, uint32_t shard, vc_engine** e, uint64_t* first_id, uint64_t* n_ids) {
  if (!h || shard >= h->G) return VC_ERR_INVALID;
  if (e) *e = h->eng[shard];
  if (first_id) *first_id = (uint64_t)h->cfg.engine.id_base + h->lo[shard];
  if (n_ids) *n_ids = h->hi[shard] - h->lo[shard];
  return VC_OK;
}

// records arrive in global id order (build_hash_tables.cc:40-70: id = ordinal of the record in the file) and fill
// the shards' id ranges one after the other
int vc_sharded_add_codes(vc_sharded* h, const void* codes, uint64_t n) {
  if (!h || (!codes && n)) return VC_ERR_INVALID;
  if (h->n + n > h->capacity) return sfail(h, VC_ERR_CAPACITY, "add %llu codes: beyond capacity %llu", (unsigned long long)n, (unsigned long long)h->capacity);
  const uint8_t* src = (const uint8_t*)codes;
  while (n) {
    const uint32_t g = shard_of(h, h->n);
    const uint64_t take = std::min<uint64_t>(n, h->hi[g] - h->n);
    int rc = vc_add_codes(h->eng[g], src, take);
    if (rc) return sfail(h, rc, "shard %u: %s", g, vc_last_error(h->eng[g]));
    src += take * h->nbytes;
    h->n += take;
    n -= take;
  }
  return VC_OK;
}

int vc_sharded_add_synthetic(vc_sharded* h, uint64_t n, uint64_t seed, uint32_t kind, uint32_t n_centres, uint32_t max_flips) {
  if (!h) return VC_ERR_INVALID;
  if (h->n + n > h->capacity) return sfail(h, VC_ERR_CAPACITY, "add_synthetic beyond capacity");
  while (n) {
    const uint32_t g = shard_of(h, h->n);
    const uint64_t take = std::min<uint64_t>(n, h->hi[g] - h->n);
    int rc = vc_add_synthetic(h->eng[g], take, seed, kind, n_centres, max_flips);
    if (rc) return sfail(h, rc, "shard %u: %s", g, vc_last_error(h->eng[g]));
    h->n += take;
    n -= take;
  }
  return VC_OK;
}

int vc_sharded_build_index(vc_sharded* h) {
  if (!h) return VC_ERR_INVALID;
  for (uint32_t g = 0; g < h->G; ++g) {
    if (shard_size(h, g) == 0) continue;   // a store filled below its capacity leaves the trailing shards empty: nothing to index
    int rc = vc_build_index(h->eng[g]);
    if (rc) return sfail(h, rc, "shard %u: %s", g, vc_last_error(h->eng[g]));
  }
  return VC_OK;
}

int vc_sharded_get_code(vc_sharded* h, uint32_t id, void* out) {
  if (!h || !out) return VC_ERR_INVALID;
  if (id < h->cfg.engine.id_base || (uint64_t)id - h->cfg.engine.id_base >= h->n) return VC_NOT_FOUND;
  return vc_get_code(h->eng[shard_of(h, id - h->cfg.engine.id_base)], id, out);
}

// HashIndex -> Image_List over all shards: a bucket is the concatenation of the shards' buckets in shard (= id) order
int vc_sharded_get_bucket(vc_sharded* h, uint32_t table, uint32_t index, uint32_t* ids, void* codes, uint32_t cap, uint32_t* n) {
  if (!h || !n) return VC_ERR_INVALID;
  uint32_t total = 0;
  for (uint32_t g = 0; g < h->G; ++g) {
    if (shard_size(h, g) == 0) continue;
    uint32_t got = 0;
    const uint32_t room = total < cap ? cap - total : 0;
    int rc = vc_get_bucket(h->eng[g], table, index, ids ? ids + std::min(total, cap) : nullptr,
                           codes ? (uint8_t*)codes + (size_t)std::min(total, cap) * h->nbytes : nullptr, room, &got);
    if (rc < 0) return sfail(h, rc, "shard %u: %s", g, vc_last_error(h->eng[g]));
    total += got;
  }
  *n = total;
  return total ? VC_OK : VC_NOT_FOUND;
}

// ---- a batch -------------------------------------------------------------------------------------------------
// Slot of shard g in a gather buffer: [nq][k] rows | [nq] counts (padded to 8 bytes) | [nq] vc_query_stats (when wanted).
// Rows, counts and statistics travel together: ONE peer copy per remote shard, or ONE all-gather.
struct SlotLayout {
  size_t rows, cnt_off, stats_off, words;
  SlotLayout(uint32_t nq, uint32_t k, bool stats) {
    rows = (size_t)nq * k;
    cnt_off = rows;
    stats_off = rows + ((size_t)nq + 1) / 2;
    words = stats_off + (stats ? (size_t)nq * (sizeof(vc_query_stats) / 8) : 0);
  }
};
static_assert(sizeof(vc_query_stats) == 40, "slot layout assumes 5 words per vc_query_stats");

// SearchWorker::get_stat over the shards: every shard stops by its own rule -- the widest radius, the summed work
__global__ void vc_sharded_stats_kernel(const uint64_t* __restrict__ base, uint64_t slot_words, uint64_t stats_off, uint32_t G, uint32_t nq,
                                        const uint32_t* __restrict__ merged_cnt, vc_query_stats* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  vc_query_stats s{};
  for (uint32_t g = 0; g < G; ++g) {
    const vc_query_stats t = ((const vc_query_stats*)(base + (uint64_t)g * slot_words + stats_off))[i];
    s.radius = max(s.radius, t.radius);
    s.n_sub_reads += t.n_sub_reads;
    s.n_local_reads += t.n_local_reads;
    s.n_candidates += t.n_candidates;
  }
  s.n_results = merged_cnt[i];
  out[i] = s;
}

// Everything of a batch is enqueued, nothing is waited for (LINEAR; the MIH modes make the host wait inside each shard's
// vc_search_knn_dev -- how many queries continue decides what is enqueued next -- which is why lanes of different devices
// then run on host threads): queries, rows, counts and statistics live in HBM on the root device, valid in `S` order.
static int sharded_search_dev(vc_sharded* h, const void* d_queries, uint32_t nq, uint32_t k, uint32_t mode, uint64_t* d_out,
                              uint32_t* d_counts, vc_query_stats* d_stats, hipStream_t S) {
  const SlotLayout L(nq, k, d_stats != nullptr);
  const size_t qbytes = (size_t)nq * h->nbytes;
  int rc;
  for (uint32_t li = 0; li < h->lanes.size(); ++li) {
    Lane& l = h->lanes[li];
    const bool everything = li == h->root_lane || h->exchange == VC_EXCHANGE_RCCL;   // gathered results of all shards land here
    VS_HIP(h, hipSetDevice(l.dev));
    if ((rc = sgrow(h, &l.gath, &l.gath_bytes, L.words * 8 * (everything ? h->G : l.shards.back() + 1)))) return rc;
    if (li != h->root_lane && (rc = sgrow(h, (uint8_t**)&l.q, &l.q_bytes, qbytes))) return rc;
  }
  Lane& R = h->lanes[h->root_lane];
  if (!d_counts) {
    VS_HIP(h, hipSetDevice(h->root));
    if ((rc = sgrow(h, &h->d_ocnt, &h->ocnt_bytes, (size_t)nq * 4))) return rc;
    d_counts = h->d_ocnt;
  }
  // ---- the queries reach every device once: an event on the caller's stream, a peer copy per remote lane
  if (h->lanes.size() > 1) {
    VS_HIP(h, hipSetDevice(h->root));
    VS_HIP(h, hipEventRecord(h->ev_q, S));
    for (uint32_t li = 0; li < h->lanes.size(); ++li) {
      if (li == h->root_lane) continue;
      Lane& l = h->lanes[li];
      VS_HIP(h, hipSetDevice(l.dev));
      VS_HIP(h, hipStreamWaitEvent(l.stream, h->ev_q, 0));   // also orders the lane behind the previous batch's copies out of its slots
      VS_HIP(h, hipMemcpyPeerAsync(l.q, l.dev, d_queries, h->root, qbytes, l.stream));
    }
  }
  // ---- every shard answers the batch for its id range, the shards of a device one after the other
  std::vector<int> lane_rc(h->lanes.size(), VC_OK);
  std::vector<std::string> lane_err(h->lanes.size());
  auto run_lane = [&](uint32_t li) {
    Lane& l = h->lanes[li];
    const bool is_root = li == h->root_lane;
    hipStream_t ls = is_root ? S : l.stream;
    const void* q = is_root ? d_queries : l.q;
    auto hip = [&](hipError_t r, const char* what) {
      if (r != hipSuccess && lane_rc[li] == VC_OK) { lane_rc[li] = VC_ERR_HIP; lane_err[li] = std::string(what) + ": " + hipGetErrorString(r); }
      return r == hipSuccess;
    };
    if (!hip(hipSetDevice(l.dev), "hipSetDevice")) return;
    for (uint32_t g : l.shards) {
      uint64_t* slot = l.gath + (size_t)g * L.words;
      if (shard_size(h, g) == 0) {
        // an empty shard (a store filled below its capacity) contributes INF rows, zero counts, zero statistics; its engine
        // is never asked -- an exact radius loop over nothing would walk every shell of every table (search_worker.cc:170)
        if (!hip(hipMemsetAsync(slot, 0xFF, L.rows * 8, ls), "hipMemsetAsync") ||
            !hip(hipMemsetAsync(slot + L.cnt_off, 0, (L.words - L.cnt_off) * 8, ls), "hipMemsetAsync")) return;
        continue;
      }
      const int r = vc_search_knn_dev_stats(h->eng[g], q, nq, k, mode, slot, (uint32_t*)(slot + L.cnt_off),
                                            d_stats ? (vc_query_stats*)(slot + L.stats_off) : nullptr, ls);
      if (r != VC_OK) { lane_rc[li] = r; lane_err[li] = std::string("shard ") + std::to_string(g) + ": " + vc_last_error(h->eng[g]); return; }
    }
    if (!is_root) hip(hipEventRecord(l.done, ls), "hipEventRecord");
  };
  if (mode == VC_MODE_LINEAR || h->lanes.size() == 1) {
    for (uint32_t li = 0; li < h->lanes.size(); ++li) run_lane(li);
  } else {
    std::vector<std::thread> th;
    for (uint32_t li = 0; li < h->lanes.size(); ++li) th.emplace_back(run_lane, li);
    for (auto& t : th) t.join();
  }
  for (uint32_t li = 0; li < h->lanes.size(); ++li)
    if (lane_rc[li]) return sfail(h, lane_rc[li], "%s", lane_err[li].c_str());
  // ---- ONE exchange (replaces gather_vectors, mpi_coordinator.cc:34-69): every shard's slot -> the root's gather buffer
  if (h->exchange == VC_EXCHANGE_RCCL) {
    ncclResult_t nr = h->rccl.GroupStart();
    for (uint32_t g = 0; g < h->G && nr == ncclSuccess; ++g) {   // rank g = shard g = its device's lane; in place: the send slot sits in the receive buffer
      Lane& l = h->lanes[h->lane_of[g]];
      nr = h->rccl.AllGather(l.gath + (size_t)g * L.words, l.gath, L.words, ncclUint64, h->comms[g], h->lane_of[g] == h->root_lane ? S : l.stream);
    }
    const ncclResult_t ne = h->rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return sfail(h, VC_ERR_HIP, "ncclAllGather: %s", h->rccl.GetErrorString(nr));
  } else {
    VS_HIP(h, hipSetDevice(h->root));
    for (uint32_t li = 0; li < h->lanes.size(); ++li) {
      if (li == h->root_lane) continue;      // the root's shards wrote their slots in place
      Lane& l = h->lanes[li];
      VS_HIP(h, hipStreamWaitEvent(S, l.done, 0));
      for (uint32_t g : l.shards)
        VS_HIP(h, hipMemcpyPeerAsync(R.gath + (size_t)g * L.words, h->root, l.gath + (size_t)g * L.words, l.dev, L.words * 8, S));
    }
  }
  // ---- merge (replaces the master's heap, search_worker.cc:179-199); a flagged shard row flags the merged row
  VS_HIP(h, hipSetDevice(h->root));
  VS_HIP(h, vc_launch_select_slots(R.gath, L.words, (uint32_t)L.cnt_off, h->G, nq, k, d_out, d_counts, S));
  if (d_stats) {
    hipLaunchKernelGGL(vc_sharded_stats_kernel, dim3((nq + 255) / 256), dim3(256), 0, S, (const uint64_t*)R.gath, (uint64_t)L.words,
                       (uint64_t)L.stats_off, h->G, nq, (const uint32_t*)d_counts, d_stats);
    VS_HIP(h, hipGetLastError());
  }
  return VC_OK;
}

int vc_sharded_root_device(const vc_sharded* h, int* device) {
  if (!h || !device) return VC_ERR_INVALID;
  *device = h->root;
  return VC_OK;
}

int vc_sharded_search_knn_dev(vc_sharded* h, const void* d_queries, uint32_t nq, uint32_t k, uint32_t mode, uint64_t* d_out,
                              uint32_t* d_counts, vc_query_stats* d_stats, void* stream) {
  if (!h || !d_queries || !d_out || nq == 0 || k == 0 || k > VC_MAX_K || mode > VC_MODE_MIH_APPROX) return VC_ERR_INVALID;
  return sharded_search_dev(h, d_queries, nq, k, mode, d_out, d_counts, d_stats, stream == VC_STREAM_OWN ? h->root_stream : (hipStream_t)stream);
}

int vc_sharded_search_knn(vc_sharded* h, const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order,
                          uint64_t* out, uint32_t* counts, vc_query_stats* stats) {
  if (!h || !queries || !out || nq == 0 || k == 0 || k > VC_MAX_K || mode > VC_MODE_MIH_APPROX || order > VC_ORDER_FARTHEST_FIRST)
    return VC_ERR_INVALID;
  const size_t qbytes = (size_t)nq * h->nbytes, rows = (size_t)nq * k;
  int rc;
  hipStream_t S = h->root_stream;
  VS_HIP(h, hipSetDevice(h->root));
  if ((rc = sgrow(h, (uint8_t**)&h->d_hq, &h->hq_bytes, qbytes))) return rc;
  if ((rc = sgrow(h, &h->d_out, &h->out_bytes, rows * 8))) return rc;
  if ((rc = sgrow(h, &h->d_ocnt, &h->ocnt_bytes, (size_t)nq * 4))) return rc;
  if (stats && (rc = sgrow(h, &h->d_ostats, &h->ostats_bytes, (size_t)nq * sizeof(vc_query_stats)))) return rc;
  VS_HIP(h, hipMemcpyAsync(h->d_hq, queries, qbytes, hipMemcpyHostToDevice, S));
  if ((rc = sharded_search_dev(h, h->d_hq, nq, k, mode, h->d_out, h->d_ocnt, stats ? h->d_ostats : nullptr, S))) return rc;
  std::vector<uint32_t> cnt(nq);
  VS_HIP(h, hipSetDevice(h->root));
  VS_HIP(h, hipMemcpyAsync(out, h->d_out, rows * 8, hipMemcpyDeviceToHost, S));
  VS_HIP(h, hipMemcpyAsync(cnt.data(), h->d_ocnt, (size_t)nq * 4, hipMemcpyDeviceToHost, S));
  if (stats) VS_HIP(h, hipMemcpyAsync(stats, h->d_ostats, (size_t)nq * sizeof(vc_query_stats), hipMemcpyDeviceToHost, S));
  VS_HIP(h, hipStreamSynchronize(S));
  bool gave_up = false;
  for (uint32_t i = 0; i < nq; ++i) gave_up = gave_up || cnt[i] == 0xFFFFFFFFu;
  if (gave_up && mode == VC_MODE_LINEAR) {
    // A shard's device-side ring-overflow recovery gave up (its grid never met; never seen outside tests): the batch is
    // answered again through the shards' host API, whose host-driven recovery always ends, one shard after the other;
    // the rows go back into the root's slots and are merged by the same kernel.
    const SlotLayout L(nq, k, false);
    Lane& R = h->lanes[h->root_lane];
    VS_HIP(h, hipSetDevice(h->root));
    if ((rc = sgrow(h, &R.gath, &R.gath_bytes, L.words * 8 * h->G))) return rc;
    std::vector<uint64_t> slot(L.words);
    for (uint32_t g = 0; g < h->G; ++g) {
      uint32_t* c = (uint32_t*)(slot.data() + L.cnt_off);
      std::fill(slot.begin(), slot.begin() + L.rows, VC_PACK_INF);
      std::fill(slot.begin() + L.rows, slot.end(), 0ull);
      if (shard_size(h, g)) {
        rc = vc_search_knn(h->eng[g], queries, nq, k, VC_MODE_LINEAR, VC_ORDER_ASCENDING, slot.data(), c, nullptr);
        if (rc) return sfail(h, rc, "shard %u: %s", g, vc_last_error(h->eng[g]));
        for (uint32_t i = 0; i < nq; ++i)
          for (uint32_t j = c[i]; j < k; ++j) slot[(size_t)i * k + j] = VC_PACK_INF;
      }
      VS_HIP(h, hipSetDevice(h->root));
      VS_HIP(h, hipMemcpyAsync(R.gath + (size_t)g * L.words, slot.data(), L.words * 8, hipMemcpyHostToDevice, S));
      VS_HIP(h, hipStreamSynchronize(S));   // `slot` is reused by the next shard
    }
    VS_HIP(h, vc_launch_select_slots(R.gath, L.words, (uint32_t)L.cnt_off, h->G, nq, k, h->d_out, h->d_ocnt, S));
    VS_HIP(h, hipMemcpyAsync(out, h->d_out, rows * 8, hipMemcpyDeviceToHost, S));
    VS_HIP(h, hipMemcpyAsync(cnt.data(), h->d_ocnt, (size_t)nq * 4, hipMemcpyDeviceToHost, S));
    VS_HIP(h, hipStreamSynchronize(S));
  }
  for (uint32_t i = 0; i < nq; ++i) {
    if (order == VC_ORDER_FARTHEST_FIRST) std::reverse(out + (size_t)i * k, out + (size_t)i * k + cnt[i]);
    if (counts) counts[i] = cnt[i];
    if (stats) stats[i].n_results = cnt[i];
  }
  return VC_OK;
}

// All items within `radius` of every query over all shards (search_R_neighbors per rank + gather_vectors + the master's dedup,
// search_worker.cc:177-199,222-264: every shard searches its id range, the per-query results of the shards are brought together and
// ordered).  Shard by shard through the shards' host API (a shard's radius search waits for its own total anyway); the
// concatenated segments are ordered ON THE DEVICE by the radius search's own segment sort, one block per query.
int vc_sharded_search_radius(vc_sharded* h, const void* queries, uint32_t nq, uint32_t radius, uint32_t mode, uint64_t* out,
                             uint64_t out_cap, uint64_t* out_offsets) {
  if (!h || !queries || !out_offsets || nq == 0 || (mode != VC_MODE_LINEAR && mode != VC_MODE_MIH_EXACT) || (!out && out_cap))
    return VC_ERR_INVALID;
  std::vector<std::vector<uint64_t>> res(h->G), offs(h->G, std::vector<uint64_t>((size_t)nq + 1, 0));
  for (uint32_t g = 0; g < h->G; ++g) {
    if (shard_size(h, g) == 0) continue;
    uint64_t cap = (uint64_t)nq * 64;
    for (int attempt = 0;; ++attempt) {
      res[g].resize(std::max<uint64_t>(cap, 1));
      const int rc = vc_search_radius(h->eng[g], queries, nq, radius, mode, res[g].data(), cap, offs[g].data());
      if (rc == VC_OK) break;
      if (rc != VC_ERR_CAPACITY || attempt) return sfail(h, rc, "shard %u: %s", g, vc_last_error(h->eng[g]));
      cap = offs[g][nq];   // the needed size came back in the offsets
    }
  }
  // per-query totals -> output offsets; the largest query sizes the segment ring (a power of two, as the sort wants it)
  std::vector<uint32_t> count(nq);
  uint64_t total = 0, mx = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    uint64_t c = 0;
    for (uint32_t g = 0; g < h->G; ++g) c += offs[g][q + 1] - offs[g][q];
    out_offsets[q] = total;
    total += c;
    mx = std::max(mx, c);
    if (c > 0xFFFFFFFFull) return sfail(h, VC_ERR_CAPACITY, "radius search: a query has more than 2^32 neighbours");
    count[q] = (uint32_t)c;
  }
  out_offsets[nq] = total;
  if (total > out_cap) return sfail(h, VC_ERR_CAPACITY, "radius search: output buffer too small (needed counts are in out_offsets)");
  if (total == 0) return VC_OK;
  uint64_t cap2 = 2;
  while (cap2 < mx) cap2 <<= 1;
  if (cap2 > 0x80000000ull || (uint64_t)nq * cap2 * 8 > (4ull << 30))   // (the padded segments are assembled in host memory first)
    return sfail(h, VC_ERR_CAPACITY, "radius search: result segments too large (a query with %llu neighbours x %u queries): search fewer queries per call", (unsigned long long)mx, nq);
  std::vector<uint64_t> ring((size_t)nq * cap2);
  for (uint32_t q = 0; q < nq; ++q) {      // shard after shard = ascending id ranges; the distances interleave: ordered below
    uint64_t* dst = ring.data() + (size_t)q * cap2;
    for (uint32_t g = 0; g < h->G; ++g)
      dst = std::copy(res[g].begin() + offs[g][q], res[g].begin() + offs[g][q + 1], dst);
  }
  hipStream_t S = h->root_stream;
  VS_HIP(h, hipSetDevice(h->root));
  uint64_t *d_ring = nullptr, *d_offs = nullptr, *d_out = nullptr;
  uint32_t* d_count = nullptr;
  auto cleanup = [&]() { (void)hipFree(d_ring); (void)hipFree(d_offs); (void)hipFree(d_out); (void)hipFree(d_count); };
#define VR_HIP(call) do { hipError_t _r = (call); if (_r != hipSuccess) { cleanup(); return sfail(h, _r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP, "%s: %s", #call, hipGetErrorString(_r)); } } while (0)
  VR_HIP(hipMalloc((void**)&d_ring, ring.size() * 8));
  VR_HIP(hipMalloc((void**)&d_offs, ((size_t)nq + 1) * 8));
  VR_HIP(hipMalloc((void**)&d_out, total * 8));
  VR_HIP(hipMalloc((void**)&d_count, (size_t)nq * 4));
  VR_HIP(hipMemcpyAsync(d_ring, ring.data(), ring.size() * 8, hipMemcpyHostToDevice, S));
  VR_HIP(hipMemcpyAsync(d_offs, out_offsets, ((size_t)nq + 1) * 8, hipMemcpyHostToDevice, S));
  VR_HIP(hipMemcpyAsync(d_count, count.data(), (size_t)nq * 4, hipMemcpyHostToDevice, S));
  VR_HIP(vc_launch_sort_compact_segments(d_ring, (uint32_t)cap2, d_count, d_offs, d_out, total, nq, S));
  VR_HIP(hipMemcpyAsync(out, d_out, total * 8, hipMemcpyDeviceToHost, S));
  VR_HIP(hipStreamSynchronize(S));
#undef VR_HIP
  cleanup();
  return VC_OK;
}

}  // extern "C"
