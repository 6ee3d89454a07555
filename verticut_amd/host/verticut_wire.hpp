// ============================================================================
// verticut_wire.hpp -- protobuf wire encoding of the reference's records (src/image_search.proto:3-27),
// written out by hand (proto2 varint / length-delimited fields; no libprotobuf needed):
//
//   message ID            { required uint32 id = 1; }
//   message BinaryCode    { required bytes code = 1; }
//   message HashIndex     { required uint32 table_id = 1; required uint32 index = 2; }
//   message ID_Code_Pair  { required uint32 id = 1; required bytes code = 2; }
//   message Image_List    { repeated ID_Code_Pair images = 1; }
//
// These are the byte strings PilafProxy / MemcachedProxy / RedisProxy exchange with the KV tier
// (pilaf_proxy.h:39-62: SerializeToString on put, ParseFromString on get).  With them a BaseProxy over the GPU
// index can answer an unmodified SearchWorker byte for byte; nothing on the GPU search
// path uses them.
// ============================================================================
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "verticut_host.hpp"

namespace vc {
namespace wire {

inline void put_varint(std::string& out, uint64_t v) {
  while (v >= 0x80) {
    out.push_back((char)((v & 0x7F) | 0x80));
    v >>= 7;
  }
  out.push_back((char)v);
}
inline bool get_varint(const std::string& in, size_t& pos, uint64_t& v) {
  v = 0;
  for (int shift = 0; shift < 64 && pos < in.size(); shift += 7) {
    const uint8_t b = (uint8_t)in[pos++];
    v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}
inline void put_bytes(std::string& out, const std::string& b) {
  put_varint(out, b.size());
  out.append(b);
}

// tag = field_number << 3 | wire_type (0 varint, 2 length-delimited)
inline std::string encode(const ID& m) { std::string s(1, '\x08'); put_varint(s, m.id()); return s; }
inline std::string encode(const BinaryCode& m) { std::string s(1, '\x0A'); put_bytes(s, m.code()); return s; }
inline std::string encode(const HashIndex& m) {
  std::string s(1, '\x08');
  put_varint(s, m.table_id());
  s.push_back('\x10');
  put_varint(s, m.index());
  return s;
}
inline std::string encode(const ID_Code_Pair& m) {
  std::string s(1, '\x08');
  put_varint(s, m.id());
  s.push_back('\x12');
  put_bytes(s, m.code());
  return s;
}
inline std::string encode(const Image_List& m) {
  std::string s;
  for (int i = 0; i < m.images_size(); ++i) {
    s.push_back('\x0A');
    put_bytes(s, encode(m.images(i)));
  }
  return s;
}

// Decoders accept fields in any order and skip unknown varint / length-delimited fields, like ParseFromString.
namespace detail {
template <class F>
inline bool walk(const std::string& in, F on_field) {
  size_t pos = 0;
  while (pos < in.size()) {
    uint64_t tag;
    if (!get_varint(in, pos, tag)) return false;
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    if (wt == 0) {
      uint64_t v;
      if (!get_varint(in, pos, v)) return false;
      on_field(field, v, nullptr, 0);
    } else if (wt == 2) {
      uint64_t len;
      if (!get_varint(in, pos, len) || pos + len > in.size()) return false;
      on_field(field, 0, in.data() + pos, (size_t)len);
      pos += (size_t)len;
    } else {
      return false;
    }
  }
  return true;
}
}  // namespace detail

inline bool decode(const std::string& in, ID& m) {
  return detail::walk(in, [&](uint32_t f, uint64_t v, const char*, size_t) { if (f == 1) m.set_id((uint32_t)v); });
}
inline bool decode(const std::string& in, BinaryCode& m) {
  return detail::walk(in, [&](uint32_t f, uint64_t, const char* p, size_t n) { if (f == 1 && p) m.set_code(p, n); });
}
inline bool decode(const std::string& in, HashIndex& m) {
  return detail::walk(in, [&](uint32_t f, uint64_t v, const char*, size_t) {
    if (f == 1) m.set_table_id((uint32_t)v);
    if (f == 2) m.set_index((uint32_t)v);
  });
}
inline bool decode(const std::string& in, ID_Code_Pair& m) {
  return detail::walk(in, [&](uint32_t f, uint64_t v, const char* p, size_t n) {
    if (f == 1 && !p) m.set_id((uint32_t)v);
    if (f == 2 && p) m.set_code(p, n);
  });
}
inline bool decode(const std::string& in, Image_List& m) {
  m.clear_images();
  bool ok = true;
  const bool w = detail::walk(in, [&](uint32_t f, uint64_t, const char* p, size_t n) {
    if (f == 1 && p) ok = decode(std::string(p, n), *m.add_images()) && ok;
  });
  return w && ok;
}

// The KV tier as the reference's proxies see it: serialized key in, serialized value out
// (memcached_proxy.h:35-62 get: key.SerializeToString -> server -> value.ParseFromString).
// A key carrying field 2 is a HashIndex (-> Image_List bytes), otherwise an ID (-> BinaryCode bytes).
inline int kv_get(GpuProxy& proxy, const std::string& key_bytes, std::string* value_bytes) {
  bool has_index = false;
  if (!detail::walk(key_bytes, [&](uint32_t f, uint64_t, const char*, size_t) { has_index |= f == 2; })) return PROXY_NOT_FOUND;
  if (has_index) {
    HashIndex hi;
    Image_List list;
    if (!decode(key_bytes, hi)) return PROXY_NOT_FOUND;
    const int rc = proxy.get(hi, list);
    if (rc == PROXY_FOUND) *value_bytes = encode(list);
    return rc;
  }
  ID id;
  BinaryCode code;
  if (!decode(key_bytes, id)) return PROXY_NOT_FOUND;
  const int rc = proxy.get(id, code);
  if (rc == PROXY_FOUND) *value_bytes = encode(code);
  return rc;
}

}  // namespace wire
}  // namespace vc
