// placeholder until the MIH kernels land (next commit)
#include "vc_mih.hpp"
struct VcMihIndex { int dummy; };
static int nyi(std::string* err) { if (err) *err = "MIH path not built yet"; return VC_ERR_STATE; }
int vc_mih_build(VcMihIndex**, const uint64_t*, uint64_t, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, hipStream_t, std::string* err) { return nyi(err); }
void vc_mih_free(VcMihIndex* ix) { delete ix; }
int vc_mih_search(VcMihIndex*, const uint64_t*, uint64_t, uint64_t, const uint64_t*, uint32_t, uint32_t, bool, uint64_t*, uint32_t*, vc_query_stats*, hipStream_t, std::string* err) { return nyi(err); }
int vc_mih_bucket(VcMihIndex*, uint32_t, uint32_t, std::vector<uint32_t>*, hipStream_t, std::string* err) { return nyi(err); }
int vc_mih_bitmap_test(VcMihIndex*, uint32_t, uint32_t, int*, hipStream_t, std::string* err) { return nyi(err); }
int vc_mih_bitmap_read(VcMihIndex*, uint32_t, uint64_t, uint64_t, uint32_t*, hipStream_t, std::string* err) { return nyi(err); }
int vc_radius_search(VcMihIndex*, bool, const uint64_t*, uint64_t, uint64_t, uint32_t, uint32_t, uint32_t, const uint64_t*, uint32_t, uint32_t, uint64_t*, uint64_t, uint64_t*, hipStream_t, std::string* err) { return nyi(err); }
