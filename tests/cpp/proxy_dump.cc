// Test helper: drives vc::GpuProxy exactly the way the reference drives a BaseProxy
// (put(ID, BinaryCode) per record like a loader, get(HashIndex, Image_List) like search_worker.cc:246,
// get(ID, BinaryCode) like linear_search.cc:45-46) and dumps what comes back.
//   proxy_dump <code_file> <n> <bits> <n_tables> <table> <index> [<table> <index> ...]
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "verticut_wire.hpp"

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const uint64_t n = strtoull(argv[2], nullptr, 10);
  const uint32_t bits = atoi(argv[3]), m = atoi(argv[4]), nbytes = bits / 8;
  vc::Engine engine(bits, m, n);
  vc::GpuProxy proxy(&engine);
  if (proxy.init("unused") != 0) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  std::vector<char> rec(nbytes);
  for (uint64_t i = 0; i < n && fread(rec.data(), nbytes, 1, f) == 1; ++i) {
    vc::ID id;
    vc::BinaryCode code;
    id.set_id((uint32_t)i);
    code.set_code(rec.data(), nbytes);
    if (proxy.put(id, code) != vc::PROXY_PUT_DONE) return 3;
  }
  fclose(f);
  {  // out-of-order id is refused, like a failed put
    vc::ID id;
    vc::BinaryCode code;
    id.set_id(12345678);
    code.set_code(rec.data(), nbytes);
    printf("put_out_of_order %d\n", proxy.put(id, code));
  }
  engine.check(vc_build_index(engine.handle()));
  for (int a = 5; a + 1 < argc; a += 2) {
    vc::HashIndex hi;
    hi.set_table_id(atoi(argv[a]));
    hi.set_index((uint32_t)strtoul(argv[a + 1], nullptr, 10));
    vc::Image_List list;
    const int rc = proxy.get(hi, list);
    printf("bucket %u %u rc=%d n=%d\n", hi.table_id(), hi.index(), rc, list.images_size());
    for (int i = 0; i < list.images_size(); ++i) {
      printf("  %u ", list.images(i).id());
      for (unsigned char c : list.images(i).code()) printf("%02x", c);
      printf("\n");
    }
  }
  vc::ID id;
  vc::BinaryCode code;
  id.set_id(7);
  printf("get_id7 rc=%d ", proxy.get(id, code));
  for (unsigned char c : code.code()) printf("%02x", c);
  printf("\n");
  id.set_id((uint32_t)n + 5);
  printf("get_missing rc=%d\n", proxy.get(id, code));
  // byte-level KV view (what a memcached/redis/pilaf server would be asked): first probe as HashIndex bytes, id 7 as ID bytes
  {
    vc::HashIndex hi;
    hi.set_table_id(atoi(argv[5]));
    hi.set_index((uint32_t)strtoul(argv[6], nullptr, 10));
    std::string val;
    const int rc = vc::wire::kv_get(proxy, vc::wire::encode(hi), &val);
    printf("kv_bucket rc=%d ", rc);
    for (unsigned char c : val) printf("%02x", c);
    printf("\n");
    vc::ID k7;
    k7.set_id(7);
    val.clear();
    printf("kv_id7 rc=%d ", vc::wire::kv_get(proxy, vc::wire::encode(k7), &val));
    for (unsigned char c : val) printf("%02x", c);
    printf("\n");
  }
  proxy.close();
  return 0;
}
