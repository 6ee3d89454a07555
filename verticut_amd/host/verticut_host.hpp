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

#include <list>
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

// Owns one vc_engine (one shard / GPU).
class Engine {
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
  ~Engine() { vc_destroy(h_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  vc_engine* handle() const { return h_; }
  uint32_t bits() const { return bits_; }
  uint32_t nbytes() const { return bits_ / 8; }
  uint32_t n_tables() const { return m_; }
  void check(int rc) const {
    if (rc < 0) throw EngineError(rc, vc_last_error(h_));
  }
 private:
  vc_engine* h_ = nullptr;
  uint32_t bits_ = 0, m_ = 0;
};

// BaseProxy over the resident index.  put(ID, BinaryCode) appends a record (ids must arrive in order, as
// build_hash_tables.cc:55-69 produces them); put(HashIndex, Image_List) is accepted and ignored because the
// buckets are derived from the records by vc_build_index (rule a12) -- the reference's read-modify-write of
// bucket lists has no equivalent to perform.
class GpuProxy : public BaseProxy<Message, Message> {
 public:
  explicit GpuProxy(Engine* e) : e_(e) {}
  int init(const char*) override { return 0; }   // pilaf_proxy.h:65-80 reads a host list; nothing to connect to here
  void close() override {}
  int contain(const Message&) override { return PROXY_NOT_FOUND; }  // unimplemented in every reference proxy too

  int get(const Message& key, Message& value) override {
    if (const HashIndex* hi = dynamic_cast<const HashIndex*>(&key)) {
      Image_List* out = dynamic_cast<Image_List*>(&value);
      if (!out) return PROXY_NOT_FOUND;
      out->clear_images();
      uint32_t n = 0;
      int rc = vc_get_bucket(e_->handle(), hi->table_id(), hi->index(), nullptr, nullptr, 0, &n);
      if (rc == VC_NOT_FOUND) return PROXY_NOT_FOUND;
      e_->check(rc);
      std::vector<uint32_t> ids(n);
      std::string codes((size_t)n * e_->nbytes(), '\0');
      e_->check(vc_get_bucket(e_->handle(), hi->table_id(), hi->index(), ids.data(), &codes[0], n, &n));
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
      int rc = vc_get_code(e_->handle(), id->id(), &buf[0]);
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
      uint64_t n = 0;
      vc_size(e_->handle(), &n);
      if (!c || c->code().size() != e_->nbytes() || id->id() != n) return PROXY_PUT_FAIL;
      return vc_add_codes(e_->handle(), c->code().data(), 1) == VC_OK ? PROXY_PUT_DONE : PROXY_PUT_FAIL;
    }
    if (dynamic_cast<const HashIndex*>(&key)) return PROXY_PUT_DONE;
    return PROXY_PUT_FAIL;
  }

 private:
  Engine* e_;
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
  SearchWorker(Engine* engine, int image_total) : e_(engine), image_total_(image_total) {}

  // search_worker.cc:65-89.  approximate -> search_K_approximate_nearest_neighbors (:93-157),
  // else search_K_nearest_neighbors (:159-218).  Farthest first, like the reference's heap drain.
  std::list<search_result_st> find(const char* binary_code, size_t nbytes, int knn, bool approximate) {
    if (nbytes != e_->nbytes() || e_->n_tables() == 0 || nbytes % e_->n_tables() != 0)   // :75 assert
      throw EngineError(VC_ERR_INVALID, "find: nbytes must equal the engine's code size and divide by n_tables");
    result_.clear();
    std::vector<uint64_t> out((size_t)knn);
    uint32_t cnt = 0;
    e_->check(vc_search_knn(e_->handle(), binary_code, 1, (uint32_t)knn, approximate ? VC_MODE_MIH_APPROX : VC_MODE_MIH_EXACT,
                            VC_ORDER_FARTHEST_FIRST, out.data(), &cnt, &stat_));
    for (uint32_t i = 0; i < cnt; ++i) result_.push_back({(uint32_t)(out[i] & 0xffffffffu), (uint32_t)(out[i] >> 32)});
    return result_;
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
    e_->check(vc_search_knn(e_->handle(), binary_code, 1, (uint32_t)knn, VC_MODE_LINEAR, VC_ORDER_FARTHEST_FIRST, out.data(),
                            &cnt, nullptr));
    std::list<search_result_st> r;
    for (uint32_t i = 0; i < cnt; ++i) r.push_back({(uint32_t)(out[i] & 0xffffffffu), (uint32_t)(out[i] >> 32)});
    return r;
  }

 private:
  Engine* e_;
  int image_total_;
  std::list<search_result_st> result_;
  vc_query_stats stat_{};
};

// ---- src/image_search_client.h:12-27 (served in process: image_search_server.cc:22-102 without ssh/popen)
class image_search_client {
 public:
  explicit image_search_client(Engine* engine) : e_(engine), worker_(engine, 0) {}
  std::string ping(const std::string& s) { return s; }   // image_search_server.cc:51-53 echoes the string
  // (image id, distance) pairs in the order the worker prints them ("id : dist" lines, farthest first).
  // The by-id path is dead in the reference as shipped (distributed_image_search.cc:116); here the code of
  // image `id` is read back from the resident DB and searched.
  std::list<std::pair<uint32_t, uint32_t> > search_image_by_id(uint32_t id, int knn, bool approximate = false) {
    std::string code(e_->nbytes(), '\0');
    int rc = vc_get_code(e_->handle(), id, &code[0]);
    if (rc == VC_NOT_FOUND) throw EngineError(VC_NOT_FOUND, "Can't find match");   // distributed_image_search.cc:98
    e_->check(rc);
    std::list<std::pair<uint32_t, uint32_t> > out;
    for (const auto& r : worker_.find(code.data(), code.size(), knn, approximate)) out.push_back({r.image_id, r.dist});
    return out;
  }
 private:
  Engine* e_;
  SearchWorker worker_;
};

}  // namespace vc
