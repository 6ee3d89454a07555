// Test helper (CPU only): prints the wire bytes of a few records and checks decode(encode(x)) == x.
#include <stdio.h>

#include "verticut_wire.hpp"

static void hex(const char* name, const std::string& s) {
  printf("%s ", name);
  for (unsigned char c : s) printf("%02x", c);
  printf("\n");
}

int main() {
  vc::ID id; id.set_id(300);
  vc::HashIndex hi; hi.set_table_id(3); hi.set_index(0xFFFF8001u);
  vc::BinaryCode bc; bc.set_code("0123456789123456", 16);           // linear_search.cc:69 sample code
  vc::Image_List il;
  for (uint32_t i = 0; i < 3; ++i) {
    vc::ID_Code_Pair* p = il.add_images();
    p->set_id(i * 1000000u);
    p->set_code("0123456789123456", 16);
  }
  hex("id", vc::wire::encode(id));
  hex("hashindex", vc::wire::encode(hi));
  hex("binarycode", vc::wire::encode(bc));
  hex("imagelist", vc::wire::encode(il));
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
