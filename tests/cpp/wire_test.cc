// Test helper (CPU only) for verticut_wire.hpp.
//   wire_test            prints the wire bytes of a few records and checks decode(encode(x)) == x
//   wire_test --vectors  reads one record per line from stdin (field values, as tests/golden/wire_vectors.json holds
//                        them), prints "<hex of encode(record)> <hex of encode(decode(encode(record)))>" per line:
//                        tests/test_oracle_cpu.py compares both with the bytes a real protobuf runtime produced
//     id <u32> | binarycode <hex|-> | hashindex <table> <index> | imagelist <n> {<id> <hex|->}*n
#include <stdio.h>
#include <string.h>

#include <iostream>
#include <sstream>

#include "verticut_wire.hpp"

static std::string hex(const std::string& s) {
  static const char* d = "0123456789abcdef";
  std::string o;
  for (unsigned char c : s) { o.push_back(d[c >> 4]); o.push_back(d[c & 15]); }
  return o;
}
static std::string unhex(const std::string& h) {
  std::string o;
  if (h == "-") return o;
  for (size_t i = 0; i + 1 < h.size(); i += 2) o.push_back((char)strtoul(h.substr(i, 2).c_str(), nullptr, 16));
  return o;
}

static int vectors() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string kind, enc, again;
    in >> kind;
    if (kind == "id") {
      unsigned long long v; in >> v;
      vc::ID m, m2; m.set_id((uint32_t)v);
      enc = vc::wire::encode(m);
      if (!vc::wire::decode(enc, m2)) return 2;
      again = vc::wire::encode(m2);
    } else if (kind == "binarycode") {
      std::string h; in >> h;
      const std::string c = unhex(h);
      vc::BinaryCode m, m2; m.set_code(c.data(), c.size());
      enc = vc::wire::encode(m);
      if (!vc::wire::decode(enc, m2)) return 2;
      again = vc::wire::encode(m2);
    } else if (kind == "hashindex") {
      unsigned long long t, i; in >> t >> i;
      vc::HashIndex m, m2; m.set_table_id((uint32_t)t); m.set_index((uint32_t)i);
      enc = vc::wire::encode(m);
      if (!vc::wire::decode(enc, m2)) return 2;
      again = vc::wire::encode(m2);
    } else if (kind == "imagelist") {
      int n; in >> n;
      vc::Image_List m, m2;
      for (int j = 0; j < n; ++j) {
        unsigned long long id; std::string h; in >> id >> h;
        const std::string c = unhex(h);
        vc::ID_Code_Pair* p = m.add_images();
        p->set_id((uint32_t)id);
        p->set_code(c.data(), c.size());
      }
      enc = vc::wire::encode(m);
      if (!vc::wire::decode(enc, m2) || m2.images_size() != n) return 2;
      again = vc::wire::encode(m2);
    } else if (kind.empty()) {
      continue;
    } else {
      return 3;
    }
    printf("%s %s\n", enc.empty() ? "-" : hex(enc).c_str(), again.empty() ? "-" : hex(again).c_str());
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--vectors")) return vectors();
  vc::ID id; id.set_id(300);
  vc::HashIndex hi; hi.set_table_id(3); hi.set_index(0xFFFF8001u);
  vc::BinaryCode bc; bc.set_code("0123456789123456", 16);           // linear_search.cc:69 sample code
  vc::Image_List il;
  for (uint32_t i = 0; i < 3; ++i) {
    vc::ID_Code_Pair* p = il.add_images();
    p->set_id(i * 1000000u);
    p->set_code("0123456789123456", 16);
  }
  printf("id %s\n", hex(vc::wire::encode(id)).c_str());
  printf("hashindex %s\n", hex(vc::wire::encode(hi)).c_str());
  printf("binarycode %s\n", hex(vc::wire::encode(bc)).c_str());
  printf("imagelist %s\n", hex(vc::wire::encode(il)).c_str());
  vc::ID id2; vc::HashIndex hi2; vc::BinaryCode bc2; vc::Image_List il2;
  bool ok = vc::wire::decode(vc::wire::encode(id), id2) && id2.id() == 300;
  ok = ok && vc::wire::decode(vc::wire::encode(hi), hi2) && hi2.table_id() == 3 && hi2.index() == 0xFFFF8001u;
  ok = ok && vc::wire::decode(vc::wire::encode(bc), bc2) && bc2.code() == bc.code();
  ok = ok && vc::wire::decode(vc::wire::encode(il), il2) && il2.images_size() == 3 && il2.images(2).id() == 2000000u &&
       il2.images(1).code() == "0123456789123456";
  ok = ok && !vc::wire::decode(std::string("\x0A\x7F", 2), bc2);   // truncated length-delimited field
  printf("roundtrip %s\n", ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}
