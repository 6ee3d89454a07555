// TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
//
// Thin extern "C" veneer over the two hot-path pieces of the reference that compile
// standalone from their own sources, where they lie under /root/reference:
//   * Pilaf/image_tools.h  (header only): compute_hamming_dist :21-33, binaryToInt :12-18
//   * src/bitmap.cc + src/bitmap.h      : ImageBitmap get_idx/set_idx/reset_idx :22-38
// The reference sources are NOT copied; they are found through -I at build time
// (oracle/Makefile, target `ref`) and the result goes to oracle/_ref/ (git-ignored).
// Everything else on the path (search_worker.cc, linear_search.cc) needs mpi.h and the
// protoc-generated image_search.pb.h, which this image lacks -> unbuildable here; those
// loops are restated in oracle/vc_oracle.cc and are NOT pinned by this library.
#include <stdint.h>
#include <string>
#include "image_tools.h"   // -I/root/reference/Pilaf
#include "bitmap.h"        // -I/root/reference/src

extern "C" {

int vcref_hamming(const char* a, const char* b, size_t nbytes) {
  std::string s1(a, nbytes), s2(b, nbytes);
  return compute_hamming_dist(s1, s2);
}

uint32_t vcref_binary_to_int(const char* p, int len) { return binaryToInt(p, len); }

// Bitmap round trip through the reference class (owning ctor, so its dtor's free() is valid).
void* vcref_bitmap_new(unsigned long n_bytes) { return new ImageBitmap(n_bytes); }
void vcref_bitmap_free(void* b) { delete static_cast<ImageBitmap*>(b); }
void vcref_bitmap_set(void* b, unsigned long i) { static_cast<ImageBitmap*>(b)->set_idx(i); }
void vcref_bitmap_reset(void* b, unsigned long i) { static_cast<ImageBitmap*>(b)->reset_idx(i); }
int vcref_bitmap_get(void* b, unsigned long i) { return static_cast<ImageBitmap*>(b)->get_idx(i) ? 1 : 0; }
const void* vcref_bitmap_data(void* b) { return static_cast<ImageBitmap*>(b)->data(); }

}  // extern "C"
