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

struct ShardBuf {
  void* q = nullptr;        size_t q_bytes = 0;     // the batch's queries on the shard's device
  uint64_t* top = nullptr;  size_t top_bytes = 0;   // [nq][k] per-shard top-k
  uint32_t* cnt = nullptr;  size_t cnt_bytes = 0;
  uint64_t* recv = nullptr; size_t recv_bytes = 0;  // RCCL: [G][nq][k] all-gathered rows on this device
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
};

}  // namespace

struct vc_sharded {
  vc_sharded_config cfg;
  uint32_t G = 0, D = 0, nbytes = 0;
  uint64_t capacity = 0, n = 0;
  std::vector<vc_engine*> eng;
  std::vector<int> dev;                 // device of shard g
  std::vector<uint64_t> lo, hi;         // ids [lo, hi) relative to cfg.engine.id_base
  std::vector<ShardBuf> buf;
  int root = 0;                         // device that merges
  hipStream_t root_stream = nullptr;
  uint64_t* d_gather = nullptr; size_t gather_bytes = 0;
  uint64_t* d_out = nullptr;    size_t out_bytes = 0;
  uint32_t* d_ocnt = nullptr;   size_t ocnt_bytes = 0;
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

extern "C" {

const char* vc_sharded_last_error(const vc_sharded* h) { return h ? h->err.c_str() : g_sharded_create_err.c_str(); }

int vc_sharded_destroy(vc_sharded* h) {
  if (!h) return VC_OK;
  for (uint32_t g = 0; g < h->buf.size(); ++g) {
    (void)hipSetDevice(h->dev[g]);
    if (h->buf[g].stream) (void)hipStreamSynchronize(h->buf[g].stream);
  }
  if (!h->comms.empty() && h->rccl.CommDestroy)
    for (ncclComm_t c : h->comms)
      if (c) (void)h->rccl.CommDestroy(c);
  for (uint32_t g = 0; g < h->buf.size(); ++g) {
    (void)hipSetDevice(h->dev[g]);
    ShardBuf& b = h->buf[g];
    (void)hipFree(b.q); (void)hipFree(b.top); (void)hipFree(b.cnt); (void)hipFree(b.recv);
    if (b.done) (void)hipEventDestroy(b.done);
    if (b.stream) (void)hipStreamDestroy(b.stream);
  }
  for (vc_engine* e : h->eng) vc_destroy(e);
  (void)hipSetDevice(h->root);
  (void)hipFree(h->d_gather); (void)hipFree(h->d_out); (void)hipFree(h->d_ocnt);
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
  h->buf.resize(h->G);
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
    if (hipSetDevice(h->dev[g]) != hipSuccess || hipStreamCreateWithFlags(&h->buf[g].stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->buf[g].done, hipEventDisableTiming) != hipSuccess)
      rc = sfail(nullptr, VC_ERR_HIP, "shard %u: stream / event creation failed", g);
  }
  if (rc == VC_OK) {
    h->root = h->dev[0];
    if (hipSetDevice(h->root) != hipSuccess || hipStreamCreateWithFlags(&h->root_stream, hipStreamNonBlocking) != hipSuccess)
      rc = sfail(nullptr, VC_ERR_HIP, "root stream creation failed");
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

// one exchange: the shards' [nq][k] rows -> gathered [G][nq][k] on the root device, ordered on h->root_stream
static int exchange_rows(vc_sharded* h, uint32_t nq, uint32_t k, const uint64_t** d_lists) {
  const size_t rows = (size_t)nq * k;
  if (h->exchange == VC_EXCHANGE_RCCL) {
    for (uint32_t g = 0; g < h->G; ++g) {
      VS_HIP(h, hipSetDevice(h->dev[g]));
      int rc = sgrow(h, &h->buf[g].recv, &h->buf[g].recv_bytes, rows * 8 * h->G);
      if (rc) return rc;
    }
    ncclResult_t nr = h->rccl.GroupStart();
    for (uint32_t g = 0; g < h->G && nr == ncclSuccess; ++g)   // rank g = shard g = device dev[g]; on the shard's own stream
      nr = h->rccl.AllGather(h->buf[g].top, h->buf[g].recv, rows, ncclUint64, h->comms[g], h->buf[g].stream);
    const ncclResult_t ne = h->rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return sfail(h, VC_ERR_HIP, "ncclAllGather: %s", h->rccl.GetErrorString(nr));
    // the root merges from its own copy, behind its shard's stream
    VS_HIP(h, hipSetDevice(h->root));
    VS_HIP(h, hipEventRecord(h->buf[0].done, h->buf[0].stream));
    VS_HIP(h, hipStreamWaitEvent(h->root_stream, h->buf[0].done, 0));
    *d_lists = h->buf[0].recv;
    return VC_OK;
  }
  VS_HIP(h, hipSetDevice(h->root));
  int rc = sgrow(h, &h->d_gather, &h->gather_bytes, rows * 8 * h->G);
  if (rc) return rc;
  for (uint32_t g = 0; g < h->G; ++g) {
    VS_HIP(h, hipStreamWaitEvent(h->root_stream, h->buf[g].done, 0));
    if (h->dev[g] == h->root)
      VS_HIP(h, hipMemcpyAsync(h->d_gather + (size_t)g * rows, h->buf[g].top, rows * 8, hipMemcpyDeviceToDevice, h->root_stream));
    else
      VS_HIP(h, hipMemcpyPeerAsync(h->d_gather + (size_t)g * rows, h->root, h->buf[g].top, h->dev[g], rows * 8, h->root_stream));
  }
  *d_lists = h->d_gather;
  return VC_OK;
}

int vc_sharded_search_knn(vc_sharded* h, const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order,
                          uint64_t* out, uint32_t* counts, vc_query_stats* stats) {
  if (!h || !queries || !out || nq == 0 || k == 0 || k > VC_MAX_K || mode > VC_MODE_MIH_APPROX || order > VC_ORDER_FARTHEST_FIRST)
    return VC_ERR_INVALID;
  const size_t qbytes = (size_t)nq * h->nbytes, rows = (size_t)nq * k;
  std::vector<std::vector<vc_query_stats>> sstat;
  int rc;
  // ---- every shard answers the batch for its id range
  for (uint32_t g = 0; g < h->G; ++g) {
    ShardBuf& b = h->buf[g];
    VS_HIP(h, hipSetDevice(h->dev[g]));
    if ((rc = sgrow(h, (uint8_t**)&b.q, &b.q_bytes, qbytes))) return rc;
    if ((rc = sgrow(h, &b.top, &b.top_bytes, rows * 8))) return rc;
    if ((rc = sgrow(h, &b.cnt, &b.cnt_bytes, (size_t)nq * 4))) return rc;
  }
  // host-API leg: MIH modes (every shard runs its queries to its LOCAL stop rule -- exact for the shard -- and the host
  // API reports the statistics), and the fallback of a linear batch whose device-side ring-overflow recovery gave up.
  // The calls synchronise, so the shards run on threads of their own; rows land in the shards' device buffers.
  auto host_leg = [&](uint32_t leg_mode) -> int {
    sstat.assign(h->G, std::vector<vc_query_stats>(nq));
    std::vector<int> rcs(h->G, VC_OK);
    std::vector<std::vector<uint64_t>> rows_h(h->G, std::vector<uint64_t>(rows));
    auto work = [&](uint32_t g) {
      std::vector<uint32_t> c(nq);
      rcs[g] = vc_search_knn(h->eng[g], queries, nq, k, leg_mode, VC_ORDER_ASCENDING, rows_h[g].data(), c.data(), sstat[g].data());
      if (rcs[g] == VC_OK)
        for (uint32_t i = 0; i < nq; ++i)
          for (uint32_t j = c[i]; j < k; ++j) rows_h[g][(size_t)i * k + j] = VC_PACK_INF;
    };
    if (h->G == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (uint32_t g = 0; g < h->G; ++g) th.emplace_back(work, g);
      for (auto& t : th) t.join();
    }
    for (uint32_t g = 0; g < h->G; ++g)
      if (rcs[g]) return sfail(h, rcs[g], "shard %u: %s", g, vc_last_error(h->eng[g]));
    for (uint32_t g = 0; g < h->G; ++g) {
      ShardBuf& b = h->buf[g];
      VS_HIP(h, hipSetDevice(h->dev[g]));
      VS_HIP(h, hipMemcpyAsync(b.top, rows_h[g].data(), rows * 8, hipMemcpyHostToDevice, b.stream));
      VS_HIP(h, hipStreamSynchronize(b.stream));   // rows_h goes out of scope
      VS_HIP(h, hipEventRecord(b.done, b.stream));
    }
    return VC_OK;
  };
  if (mode == VC_MODE_LINEAR) {
    // asynchronous on every shard's own stream: the devices scan concurrently
    for (uint32_t g = 0; g < h->G; ++g) {
      ShardBuf& b = h->buf[g];
      VS_HIP(h, hipSetDevice(h->dev[g]));
      VS_HIP(h, hipMemcpyAsync(b.q, queries, qbytes, hipMemcpyHostToDevice, b.stream));
      rc = vc_search_knn_dev(h->eng[g], b.q, nq, k, mode, b.top, b.cnt, b.stream);
      if (rc) return sfail(h, rc, "shard %u: %s", g, vc_last_error(h->eng[g]));
      VS_HIP(h, hipEventRecord(b.done, b.stream));
    }
    // a shard whose device-side recovery gave up flags its rows with count UINT32_MAX (vc_search_knn_dev): upper bounds
    // only -- the batch is then answered through the host API, which recovers on the host (never seen outside tests)
    bool gave_up = false;
    std::vector<uint32_t> c(nq);
    for (uint32_t g = 0; g < h->G; ++g) {
      VS_HIP(h, hipSetDevice(h->dev[g]));
      VS_HIP(h, hipMemcpyAsync(c.data(), h->buf[g].cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, h->buf[g].stream));
      VS_HIP(h, hipStreamSynchronize(h->buf[g].stream));
      for (uint32_t i = 0; i < nq; ++i) gave_up = gave_up || c[i] == 0xFFFFFFFFu;
    }
    if (gave_up && (rc = host_leg(VC_MODE_LINEAR))) return rc;
  } else if ((rc = host_leg(mode))) {
    return rc;
  }
  // ---- one exchange + merge (replaces gather_vectors + the master's heap)
  const uint64_t* d_lists = nullptr;
  if ((rc = exchange_rows(h, nq, k, &d_lists))) return rc;
  VS_HIP(h, hipSetDevice(h->root));
  if ((rc = sgrow(h, &h->d_out, &h->out_bytes, rows * 8))) return rc;
  if ((rc = sgrow(h, &h->d_ocnt, &h->ocnt_bytes, (size_t)nq * 4))) return rc;
  if ((rc = vc_merge_topk_dev(d_lists, h->G, nq, k, h->d_out, h->d_ocnt, h->root_stream))) return sfail(h, rc, "merge: %s", vc_last_error(nullptr));
  std::vector<uint32_t> cnt(nq);
  VS_HIP(h, hipMemcpyAsync(out, h->d_out, rows * 8, hipMemcpyDeviceToHost, h->root_stream));
  VS_HIP(h, hipMemcpyAsync(cnt.data(), h->d_ocnt, (size_t)nq * 4, hipMemcpyDeviceToHost, h->root_stream));
  VS_HIP(h, hipStreamSynchronize(h->root_stream));
  for (uint32_t i = 0; i < nq; ++i) {
    if (order == VC_ORDER_FARTHEST_FIRST) std::reverse(out + (size_t)i * k, out + (size_t)i * k + cnt[i]);
    if (counts) counts[i] = cnt[i];
    if (stats) {
      vc_query_stats s{};
      if (mode == VC_MODE_LINEAR) {
        s.n_candidates = h->n;
      } else {
        for (uint32_t g = 0; g < h->G; ++g) {   // every shard stops by its own rule: the widest radius, the summed work
          const vc_query_stats& t = sstat[g][i];
          s.radius = std::max(s.radius, t.radius);
          s.n_sub_reads += t.n_sub_reads;
          s.n_local_reads += t.n_local_reads;
          s.n_candidates += t.n_candidates;
        }
      }
      s.n_results = cnt[i];
      stats[i] = s;
    }
  }
  return VC_OK;
}

}  // extern "C"
