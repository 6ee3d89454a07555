// ============================================================================
// accuracy-test (GPU build): approximate vs exact k-NN quality, the reference's metrics and output lines.
// Same positional argv as the other MPI drivers (src/accuracy_test.cc, launched by src/run_test.py):
//   accuracy-test <code_file> <image_count> <binary_bits> <substr_len> <k> <server> <read_mode> <approximate>
//                 <query_id> <query_file>
// Per query (accuracy_test.cc:72-96): approximate find, exact find, then one line
//   "<mean exact dist> <mean approx dist> <inaccurate count per query per k>"
// and one line "app time : <s/query>, ex time : <s/query>".
//   mean dist        = sum of result distances / n_query / k                                   (:106-113,126-131)
//   inaccurate count = results in the approximate list (farthest first) that come before the first one
//                      with dist <= the exact list's first (farthest) distance                    (:115-124,133)
// ============================================================================
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <iostream>
#include <vector>

#include "verticut_host.hpp"

typedef std::list<vc::SearchWorker::search_result_st> result_list;

static uint32_t dist_accumulate(const result_list& a) {
  uint32_t total = 0;
  for (const auto& r : a) total += r.dist;
  return total;
}

static uint32_t test_inaccurate(uint32_t dist_threshold, const result_list& app) {
  uint32_t count = 0;
  for (const auto& r : app) {
    if (r.dist <= dist_threshold) break;
    ++count;
  }
  return count;
}

int main(int argc, char** argv) {
  if (argc < 11) {
    fprintf(stderr, "usage: accuracy-test <code_file> <image_count> <binary_bits> <substr_len> <k> <server> <read_mode> "
                    "<approximate> <query_id> <query_file>\n");
    return 2;
  }
  const uint64_t image_count = strtoull(argv[2], nullptr, 10);
  const uint32_t binary_bits = (uint32_t)atoi(argv[3]), substr_len = (uint32_t)atoi(argv[4]);
  const int k = atoi(argv[5]);
  const char* query_file = argv[10];
  if (!binary_bits || !substr_len || binary_bits % substr_len || k <= 0 || !image_count) return 2;
  const uint32_t nbytes = binary_bits / 8;
  try {
    // VC_REF_QUIRKS=1: behave exactly like the reference for substrings < 32 bit and fewer than 4 tables -- binaryToInt's
    // sign-extended bucket keys (Pilaf/image_tools.h:13) and the stop rule's literal 4 (search_worker.cc:204).  Default:
    // masked keys and min(n_tables, 4), which are exact where the reference is not.
    const char* quirks = getenv("VC_REF_QUIRKS");
    const uint32_t flags = (quirks && atoi(quirks)) ? (VC_FLAG_REF_SIGNEXT_KEYS | VC_FLAG_REF_STOP_LITERAL4) : 0u;
    std::unique_ptr<vc::Backend> store(vc::make_backend(binary_bits, binary_bits / substr_len, image_count, flags));   // VC_SHARDS=G: several GPUs
    vc::Backend& engine = *store;
    uint64_t loaded = 0;
    engine.check(engine.load_code_file(argv[1], image_count, &loaded));
    engine.check(engine.build_index());
    vc::SearchWorker worker(&engine, (int)loaded);
    FILE* f = fopen(query_file, "rb");
    if (!f) {
      fprintf(stderr, "Couldn't open file %s\n", query_file);
      return 1;
    }
    uint64_t total_dist_ex = 0, total_dist_app = 0, inaccurate_count = 0;
    double time_app = 0, time_ex = 0;
    int n_query = 0;
    std::vector<char> code(nbytes);
    while (fread(code.data(), nbytes, 1, f) != 0) {
      auto t0 = std::chrono::steady_clock::now();
      result_list result_app = worker.find(code.data(), nbytes, k, true);
      time_app += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      t0 = std::chrono::steady_clock::now();
      result_list result_exact = worker.find(code.data(), nbytes, k, false);
      time_ex += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      n_query++;
      total_dist_ex += dist_accumulate(result_exact);
      total_dist_app += dist_accumulate(result_app);
      if (!result_exact.empty()) inaccurate_count += test_inaccurate(result_exact.front().dist, result_app);
      std::cout << (float)total_dist_ex / n_query / k << " " << (float)total_dist_app / n_query / k << " "
                << (float)inaccurate_count / n_query / k << std::endl;
      std::cout << "app time : " << time_app / n_query << ", ex time : " << time_ex / n_query << std::endl;
    }
    fclose(f);
  } catch (const vc::EngineError& e) {
    fprintf(stderr, "verticut_gpu error %d: %s\n", e.code(), e.what());
    return 1;
  }
  return 0;
}
