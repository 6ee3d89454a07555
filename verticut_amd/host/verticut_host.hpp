// ============================================================================
// verticut_host.hpp -- C++ host layer above the C ABI (include/verticut_gpu.h), shaped like the
// reference's own interfaces for this path so its callers can switch over:
//
//   vc::BaseProxy<K,V>        same virtuals / return codes as src/base_proxy.h:10-29
//   vc::GpuProxy              BaseProxy over the HBM-resident index:
//                               HashIndex{table_id,index} -> Image_List   (search_worker.cc:224-246)
//                               ID{id}                    -> BinaryCode   (linear_search.cc:45-46)
//   vc::SearchWorker          find(code, nbytes, knn, approximate) / get_knn / get_stat
//                             (src/search_worker.h:25-33); results farthest first (search_worker.cc:210-216)
//   vc::image_search_client   ping / search_image_by_id(id, knn, approximate)
//                             (src/image_search_client.h:12-27), in process instead of msgpack-rpc
//
// The message structs stand in for the protobuf messages of src/image_search.proto:3-27 (same field
// names and accessors; no wire format -- there is no KV tier to talk to any more).
// Header only; needs libverticut_gpu.so at link time.  No CPU search path: every call lands in the C ABI.
// ============================================================================
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <list>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/verticut_gpu.h"

namespace vc {

// ---- src/base_proxy.h:10-13
enum { PROXY_FOUND = 0, PROXY_NOT_FOUND = 1, PROXY_PUT_DONE = 0, PROXY_PUT_FAIL = 1 };

// ---- src/image_search.proto:3-27 (field semantics only)
struct Message { virtual ~Message() {} };
struct ID : Message {
  uint32_t id_ = 0;
  void set_id(uint32_t v) { id_ = v; }
  uint32_t id() const { return id_; }
};
struct BinaryCode : Message {
  std::string code_;
  void set_code(const char* p, size_t n) { code_.assign(p, n); }
  const std::string& code() const { return code_; }
};
struct HashIndex : Message {
  uint32_t table_id_ = 0, index_ = 0;
  void set_table_id(uint32_t v) { table_id_ = v; }
  void set_index(uint32_t v) { index_ = v; }
  uint32_t table_id() const { return table_id_; }
  uint32_t index() const { return index_; }
};
struct ID_Code_Pair : Message {
  uint32_t id_ = 0;
  std::string code_;
  void set_id(uint32_t v) { id_ = v; }
  void set_code(const char* p, size_t n) { code_.assign(p, n); }
  uint32_t id() const { return id_; }
  const std::string& code() const { return code_; }
};
struct Image_List : Message {
  std::vector<ID_Code_Pair> images_;
  int images_size() const { return (int)images_.size(); }
  const ID_Code_Pair& images(int i) const { return images_[i]; }
  ID_Code_Pair* add_images() { images_.emplace_back(); return &images_.back(); }
  void clear_images() { images_.clear(); }
};

// ---- src/base_proxy.h:15-29
template <class K, class V>
class BaseProxy {
 public:
  virtual ~BaseProxy() {}
  virtual int get(const K& key, V& value) = 0;
  virtual int put(const K& key, const V& value) = 0;
  virtual int contain(const K& key) = 0;
  virtual int init(const char* filename) = 0;
  virtual void close() = 0;
};

class EngineError : public std::runtime_error {
 public:
  EngineError(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
  int code() const { return code_; }
 private:
  int code_;
};

// What the adapters below need from the store: one GPU (Engine) or the GPUs of a node (ShardedEngine).
class Backend {
 public:
  virtual ~Backend() {}
  virtual uint32_t bits() const = 0;
  virtual uint32_t n_tables() const = 0;
  uint32_t nbytes() const { return bits() / 8; }
  virtual const char* last_error() const = 0;
  void check(int rc) const {
    if (rc < 0) throw EngineError(rc, last_error());
  }
  virtual int add_codes(const void* codes, uint64_t n) = 0;
  virtual int load_code_file(const char* path, uint64_t max_records, uint64_t* n_read) = 0;
  virtual int build_index() = 0;
  virtual uint64_t size() const = 0;
  virtual int get_code(uint32_t id, void* out) = 0;
  virtual int get_bucket(uint32_t table, uint32_t index, uint32_t* ids, void* codes, uint32_t cap, uint32_t* n) = 0;
  virtual int search_knn(const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order, uint64_t* out,
                         uint32_t* counts, vc_query_stats* stats) = 0;
};

// Owns one vc_engine (one shard / GPU).
class Engine : public Backend {
 public:
  Engine(uint32_t bits, uint32_t n_tables, uint64_t capacity, uint32_t flags = 0, uint32_t id_base = 0, int device = -1) {
    vc_config c{};
    c.abi_version = VC_ABI_VERSION;
    c.bits = bits;
    c.n_tables = n_tables;
    c.capacity = capacity;
    c.flags = flags;
    c.id_base = id_base;
    c.device = device;
    int rc = vc_create(&c, &h_);
    if (rc != VC_OK) throw EngineError(rc, vc_last_error(nullptr));
    bits_ = bits;
    m_ = n_tables;
  }
  ~Engine() override { vc_destroy(h_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  vc_engine* handle() const { return h_; }
  uint32_t bits() const override { return bits_; }
  uint32_t n_tables() const override { return m_; }
  const char* last_error() const override { return vc_last_error(h_); }
  int add_codes(const void* codes, uint64_t n) override { return vc_add_codes(h_, codes, n); }
  int load_code_file(const char* path, uint64_t max_records, uint64_t* n_read) override { return vc_load_code_file(h_, path, max_records, n_read); }
  int build_index() override { return vc_build_index(h_); }
  uint64_t size() const override {
    uint64_t n = 0;
    vc_size(h_, &n);
    return n;
  }
  int get_code(uint32_t id, void* out) override { return vc_get_code(h_, id, out); }
  int get_bucket(uint32_t table, uint32_t index, uint32_t* ids, void* codes, uint32_t cap, uint32_t* n) override {
    return vc_get_bucket(h_, table, index, ids, codes, cap, n);
  }
  int search_knn(const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order, uint64_t* out, uint32_t* counts,
                 vc_query_stats* stats) override {
    return vc_search_knn(h_, queries, nq, k, mode, order, out, counts, stats);
  }
 private:
  vc_engine* h_ = nullptr;
  uint32_t bits_ = 0, m_ = 0;
};

// The GPUs of one node behind the same interface (vc_sharded_*): the database split by id range into n_shards engines,
// one exchange of per-shard top-k per batch (RCCL all-gather over xGMI, or peer copies) + the merge kernel -- what the
// reference does with `mpirun -n 4` ranks, MPI_Gather/Gatherv/Bcast per radius and a master-side heap
// (run_distributed_search.py:74; search_worker.cc:99-101,177-207; mpi_coordinator.cc:26-69).
class ShardedEngine : public Backend {
 public:
  ShardedEngine(uint32_t bits, uint32_t n_tables, uint64_t capacity, uint32_t n_shards, const std::vector<int>& devices = {},
                uint32_t flags = 0, uint32_t id_base = 0, uint32_t exchange = VC_EXCHANGE_AUTO) {
    vc_sharded_config c{};
    c.abi_version = VC_ABI_VERSION;
    c.n_shards = n_shards;
    c.n_devices = (uint32_t)devices.size();
    c.exchange = exchange;
    for (size_t i = 0; i < devices.size() && i < VC_MAX_SHARDS; ++i) c.device_ids[i] = devices[i];
    c.engine.abi_version = VC_ABI_VERSION;
    c.engine.bits = bits;
    c.engine.n_tables = n_tables;
    c.engine.capacity = capacity;
    c.engine.flags = flags;
    c.engine.id_base = id_base;
    c.engine.device = -1;
    int rc = vc_sharded_create(&c, &h_);
    if (rc != VC_OK) throw EngineError(rc, vc_sharded_last_error(nullptr));
    bits_ = bits;
    m_ = n_tables;
  }
  ~ShardedEngine() override { vc_sharded_destroy(h_); }
  ShardedEngine(const ShardedEngine&) = delete;
  ShardedEngine& operator=(const ShardedEngine&) = delete;
  vc_sharded* handle() const { return h_; }
  uint32_t bits() const override { return bits_; }
  uint32_t n_tables() const override { return m_; }
  const char* last_error() const override { return vc_sharded_last_error(h_); }
  int add_codes(const void* codes, uint64_t n) override { return vc_sharded_add_codes(h_, codes, n); }
  int load_code_file(const char* path, uint64_t max_records, uint64_t* n_read) override {   // build_hash_tables.cc:40-70
    FILE* fh = fopen(path, "rb");
    if (!fh) return VC_ERR_INVALID;
    const size_t rec = nbytes(), batch = (size_t)1 << 16;
    std::vector<char> buf(rec * batch);
    uint64_t total = 0;
    int rc = VC_OK;
    while (max_records == 0 || total < max_records) {
      const size_t want = max_records ? (size_t)std::min<uint64_t>(batch, max_records - total) : batch;
      const size_t got = fread(buf.data(), rec, want, fh);
      if (got == 0) break;
      if ((rc = vc_sharded_add_codes(h_, buf.data(), got))) break;
      total += got;
    }
    fclose(fh);
    if (n_read) *n_read = total;
    return rc;
  }
  int build_index() override { return vc_sharded_build_index(h_); }
  uint64_t size() const override {
    uint64_t n = 0;
    vc_sharded_size(h_, &n);
    return n;
  }
  int get_code(uint32_t id, void* out) override { return vc_sharded_get_code(h_, id, out); }
  int get_bucket(uint32_t table, uint32_t index, uint32_t* ids, void* codes, uint32_t cap, uint32_t* n) override {
    return vc_sharded_get_bucket(h_, table, index, ids, codes, cap, n);
  }
  int search_knn(const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order, uint64_t* out, uint32_t* counts,
                 vc_query_stats* stats) override {
    return vc_sharded_search_knn(h_, queries, nq, k, mode, order, out, counts, stats);
  }
 private:
  vc_sharded* h_ = nullptr;
  uint32_t bits_ = 0, m_ = 0;
};

// VC_SHARDS=G [VC_DEVICES=0,1,...]: the drivers' switch from one engine to the sharded store (no argv position is free for
// it: run_distributed_search.py:74-79 fixes them all).  G = 1 (or unset) is the one-GPU Engine -- a one-shard store would
// only add an exchange, a merge and copies --; anything that is not a number in 1..VC_MAX_SHARDS is refused, not guessed at.
// Under VC_SHARDS the printed n_sub_reads / n_local_reads are SUMS over the shards and radius the widest shard's (every
// shard stops by its own rule, verticut_gpu.h), not the figures of one SearchWorker over the whole database.
inline Backend* make_backend(uint32_t bits, uint32_t n_tables, uint64_t capacity, uint32_t flags) {
  const char* g = getenv("VC_SHARDS");
  long shards = 1;
  if (g && *g) {
    char* end = nullptr;
    shards = strtol(g, &end, 10);
    if (end == g || *end != '\0' || shards < 1 || shards > VC_MAX_SHARDS)
      throw EngineError(VC_ERR_INVALID, std::string("VC_SHARDS must be a number in 1..") + std::to_string(VC_MAX_SHARDS) + ", got '" + g + "'");
  }
  if (shards <= 1) return new Engine(bits, n_tables, capacity, flags);
  std::vector<int> devices;
  if (const char* d = getenv("VC_DEVICES"))
    for (const char* p = d; *p;) {
      char* end = nullptr;
      const long v = strtol(p, &end, 10);
      if (end == p || v < 0 || (*end != ',' && *end != '\0'))
        throw EngineError(VC_ERR_INVALID, std::string("VC_DEVICES must be comma-separated device ordinals, got '") + d + "'");
      devices.push_back((int)v);
      p = *end == ',' ? end + 1 : end;
    }
  return new ShardedEngine(bits, n_tables, capacity, (uint32_t)shards, devices, flags);
}

// BaseProxy over the resident index.  put(ID, BinaryCode) appends a record (ids must arrive in order, as
// build_hash_tables.cc:55-69 produces them); put(HashIndex, Image_List) is accepted and ignored because the
// buckets are derived from the records by vc_build_index (rule a12) -- the reference's read-modify-write of
// bucket lists has no equivalent to perform.
class GpuProxy : public BaseProxy<Message, Message> {
 public:
  explicit GpuProxy(Backend* e) : e_(e) {}
  int init(const char*) override { return 0; }   // pilaf_proxy.h:65-80 reads a host list; nothing to connect to here
  void close() override {}
  int contain(const Message&) override { return PROXY_NOT_FOUND; }  // unimplemented in every reference proxy too

  int get(const Message& key, Message& value) override {
    if (const HashIndex* hi = dynamic_cast<const HashIndex*>(&key)) {
      Image_List* out = dynamic_cast<Image_List*>(&value);
      if (!out) return PROXY_NOT_FOUND;
      out->clear_images();
      uint32_t n = 0;
      int rc = e_->get_bucket(hi->table_id(), hi->index(), nullptr, nullptr, 0, &n);
      if (rc == VC_NOT_FOUND) return PROXY_NOT_FOUND;
      e_->check(rc);
      std::vector<uint32_t> ids(n);
      std::string codes((size_t)n * e_->nbytes(), '\0');
      e_->check(e_->get_bucket(hi->table_id(), hi->index(), ids.data(), &codes[0], n, &n));
      for (uint32_t i = 0; i < n; ++i) {
        ID_Code_Pair* p = out->add_images();
        p->set_id(ids[i]);
        p->set_code(codes.data() + (size_t)i * e_->nbytes(), e_->nbytes());
      }
      return PROXY_FOUND;
    }
    if (const ID* id = dynamic_cast<const ID*>(&key)) {
      BinaryCode* out = dynamic_cast<BinaryCode*>(&value);
      if (!out) return PROXY_NOT_FOUND;
      std::string buf(e_->nbytes(), '\0');
      int rc = e_->get_code(id->id(), &buf[0]);
      if (rc == VC_NOT_FOUND) return PROXY_NOT_FOUND;
      e_->check(rc);
      out->set_code(buf.data(), buf.size());
      return PROXY_FOUND;
    }
    return PROXY_NOT_FOUND;
  }

  int put(const Message& key, const Message& value) override {
    if (const ID* id = dynamic_cast<const ID*>(&key)) {
      const BinaryCode* c = dynamic_cast<const BinaryCode*>(&value);
      const uint64_t n = e_->size();
      if (!c || c->code().size() != e_->nbytes() || id->id() != n) return PROXY_PUT_FAIL;
      return e_->add_codes(c->code().data(), 1) == VC_OK ? PROXY_PUT_DONE : PROXY_PUT_FAIL;
    }
    if (dynamic_cast<const HashIndex*>(&key)) return PROXY_PUT_DONE;
    return PROXY_PUT_FAIL;
  }

 private:
  Backend* e_;
};

// ---- src/search_worker.h:18-64
class SearchWorker {
 public:
  struct search_result_st {
    uint32_t image_id;
    uint32_t dist;
  };

  // The reference takes (mpi_coordinator*, BaseProxy*, image_total); the coordinator is gone (all tables
  // live in one HBM), the proxy's engine is what is searched.
  SearchWorker(Backend* engine, int image_total) : e_(engine), image_total_(image_total) {}

  // search_worker.cc:65-89.  approximate -> search_K_approximate_nearest_neighbors (:93-157),
  // else search_K_nearest_neighbors (:159-218).  Farthest first, like the reference's heap drain.
  std::list<search_result_st> find(const char* binary_code, size_t nbytes, int knn, bool approximate) {
    if (nbytes != e_->nbytes() || e_->n_tables() == 0 || nbytes % e_->n_tables() != 0)   // :75 assert
      throw EngineError(VC_ERR_INVALID, "find: nbytes must equal the engine's code size and divide by n_tables");
    result_.clear();
    std::vector<uint64_t> out((size_t)knn);
    uint32_t cnt = 0;
    e_->check(e_->search_knn(binary_code, 1, (uint32_t)knn, approximate ? VC_MODE_MIH_APPROX : VC_MODE_MIH_EXACT,
                             VC_ORDER_FARTHEST_FIRST, out.data(), &cnt, &stat_));
    for (uint32_t i = 0; i < cnt; ++i) result_.push_back({(uint32_t)(out[i] & 0xffffffffu), (uint32_t)(out[i] >> 32)});
    return result_;
  }
  // The same for a BATCH of queries in one ABI call (n_query codes of nbytes each, back to back): the reference's drivers
  // call find() once per query (distributed_image_search.cc:62-85, accuracy_test.cc:72-91); on the GPU a batch shares every
  // launch, so 200 queries cost little more than one.  Results and per-query statistics exactly as n_query find() calls.
  std::vector<std::list<search_result_st> > find_batch(const char* binary_codes, size_t nbytes, uint32_t n_query, int knn,
                                                       bool approximate, std::vector<vc_query_stats>* stats = nullptr) {
    if (nbytes != e_->nbytes() || e_->n_tables() == 0 || nbytes % e_->n_tables() != 0)   // :75 assert
      throw EngineError(VC_ERR_INVALID, "find: nbytes must equal the engine's code size and divide by n_tables");
    std::vector<std::list<search_result_st> > all(n_query);
    if (n_query == 0) return all;
    std::vector<uint64_t> out((size_t)n_query * knn);
    std::vector<uint32_t> cnt(n_query);
    std::vector<vc_query_stats> st(n_query);
    e_->check(e_->search_knn(binary_codes, n_query, (uint32_t)knn, approximate ? VC_MODE_MIH_APPROX : VC_MODE_MIH_EXACT,
                             VC_ORDER_FARTHEST_FIRST, out.data(), cnt.data(), st.data()));
    for (uint32_t q = 0; q < n_query; ++q)
      for (uint32_t i = 0; i < cnt[q]; ++i) {
        const uint64_t v = out[(size_t)q * knn + i];
        all[q].push_back({(uint32_t)(v & 0xffffffffu), (uint32_t)(v >> 32)});
      }
    result_ = all.back();
    stat_ = st.back();
    if (stats) *stats = st;
    return all;
  }
  std::list<search_result_st> get_knn() { return result_; }
  // search_worker.cc:24-30
  void get_stat(uint64_t& n_main_reads, uint64_t& n_sub_reads, uint64_t& n_local_reads, uint32_t& radius) {
    n_main_reads = stat_.n_main_reads;
    n_sub_reads = stat_.n_sub_reads;
    n_local_reads = stat_.n_local_reads;
    radius = stat_.radius;
  }

  // linear_search.cc:39-64 as a member (the reference keeps it as a free function over globals)
  std::list<search_result_st> linear_search(const char* binary_code, int knn) {
    std::vector<uint64_t> out((size_t)knn);
    uint32_t cnt = 0;
    e_->check(e_->search_knn(binary_code, 1, (uint32_t)knn, VC_MODE_LINEAR, VC_ORDER_FARTHEST_FIRST, out.data(), &cnt, nullptr));
    std::list<search_result_st> r;
    for (uint32_t i = 0; i < cnt; ++i) r.push_back({(uint32_t)(out[i] & 0xffffffffu), (uint32_t)(out[i] >> 32)});
    return r;
  }

 private:
  Backend* e_;
  int image_total_;
  std::list<search_result_st> result_;
  vc_query_stats stat_{};
};

// ---- src/image_search_client.h:12-27 (served in process: image_search_server.cc:22-102 without ssh/popen)
class image_search_client {
 public:
  explicit image_search_client(Backend* engine) : e_(engine), worker_(engine, 0) {}
  std::string ping(const std::string& s) { return s; }   // image_search_server.cc:51-53 echoes the string
  // (image id, distance) pairs in the order the worker prints them ("id : dist" lines, farthest first).
  // The by-id path is dead in the reference as shipped (distributed_image_search.cc:116); here the code of
  // image `id` is read back from the resident DB and searched.
  std::list<std::pair<uint32_t, uint32_t> > search_image_by_id(uint32_t id, int knn, bool approximate = false) {
    std::string code(e_->nbytes(), '\0');
    int rc = e_->get_code(id, &code[0]);
    if (rc == VC_NOT_FOUND) throw EngineError(VC_NOT_FOUND, "Can't find match");   // distributed_image_search.cc:98
    e_->check(rc);
    std::list<std::pair<uint32_t, uint32_t> > out;
    for (const auto& r : worker_.find(code.data(), code.size(), knn, approximate)) out.push_back({r.image_id, r.dist});
    return out;
  }
 private:
  Backend* e_;
  SearchWorker worker_;
};

}  // namespace vc
