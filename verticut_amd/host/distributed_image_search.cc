// ============================================================================
// distributed-image-search (GPU build): the query driver of the reference, same positional argv
// (src/run_distributed_search.py:74-79, parsed at src/distributed_image_search.cc:141-155):
//
//   distributed-image-search <config_path> <image_count> <binary_bits> <substr_len> <k> <server>
//                            <read_mode> <approximate> <query_id> [query_file]
//
//   config_path  here: the raw binary-code file (headerless records of binary_bits/8 bytes, id = ordinal,
//                build_hash_tables.cc:40-70) -- there is no KV tier whose host list could be named
//   image_count  records to load (<= file size)
//   substr_len   substring bits; n_tables = binary_bits / substr_len (one table per former MPI rank)
//   server, read_mode   accepted and ignored (memcached/pilaf/redis selection has no meaning any more)
//   approximate  0/1 (search_worker.cc:93-157 vs :159-218)
//   query_id     >= 0: search the code of that image id (the by-id path the reference declares dead,
//                distributed_image_search.cc:116) and print "id : dist" lines (:70-72 format);
//                < 0: read queries from query_file (raw codes, at most 200: :83-84) and print the
//                "Averate result" line of :87-93
// VC_SHARDS=G (and optionally VC_DEVICES=0,1,..) spreads the records over G GPUs of the node (vc_sharded_*), which is what
// `mpirun -n 4` + the KV tier did for the reference; the printed n_sub_reads / n_local_reads are then SUMS over the shards and
// radius the widest shard's (every shard stops by its own rule), not one SearchWorker's figures.  Set VC_PRINT_RESULTS=1 to print the "id : dist" lines for file queries too; VC_REF_QUIRKS=1 reproduces the reference's
// behaviour for substrings < 32 bit / fewer than 4 tables (sign-extended keys, literal-4 stop rule).
// ============================================================================
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "verticut_host.hpp"

static int usage() {
  fprintf(stderr,
          "usage: distributed-image-search <code_file> <image_count> <binary_bits> <substr_len> <k> <server> <read_mode> "
          "<approximate> <query_id> [query_file]\n");
  return 2;
}

int main(int argc, char** argv) {
  if (argc < 10) return usage();
  const char* code_file = argv[1];
  const uint64_t image_count = strtoull(argv[2], nullptr, 10);
  const uint32_t binary_bits = (uint32_t)atoi(argv[3]);
  const uint32_t substr_len = (uint32_t)atoi(argv[4]);
  const int k = atoi(argv[5]);
  const bool approximate = atoi(argv[8]) != 0;
  const long long query_id = atoll(argv[9]);
  const char* query_file = argc > 10 ? argv[10] : nullptr;
  if (binary_bits == 0 || substr_len == 0 || binary_bits % substr_len || k <= 0 || image_count == 0) return usage();
  const uint32_t nbytes = binary_bits / 8;
  const bool print_results = getenv("VC_PRINT_RESULTS") != nullptr;

  try {
    // VC_REF_QUIRKS=1: behave exactly like the reference for substrings < 32 bit and fewer than 4 tables -- binaryToInt's
    // sign-extended bucket keys (Pilaf/image_tools.h:13) and the stop rule's literal 4 (search_worker.cc:204).  Default:
    // masked keys and min(n_tables, 4), which are exact where the reference is not.
    const char* quirks = getenv("VC_REF_QUIRKS");
    const uint32_t flags = (quirks && atoi(quirks)) ? (VC_FLAG_REF_SIGNEXT_KEYS | VC_FLAG_REF_STOP_LITERAL4) : 0u;
    // VC_SHARDS=G [VC_DEVICES=0,1,..]: the records are split by id range over G engines (the GPUs of the node) behind the
    // same interface; default: one engine on the current device
    std::unique_ptr<vc::Backend> store(vc::make_backend(binary_bits, binary_bits / substr_len, image_count, flags));
    vc::Backend& engine = *store;
    // ---- load: build_hash_tables.cc:40-70 (records in file order, id = ordinal)
    FILE* fh = fopen(code_file, "rb");
    if (!fh) {
      fprintf(stderr, "Can't open file %s.", code_file);
      return 1;
    }
    std::vector<char> buf((size_t)nbytes * (1u << 16));
    uint64_t loaded = 0;
    while (loaded < image_count) {
      const size_t want = (size_t)std::min<uint64_t>(1u << 16, image_count - loaded);
      const size_t got = fread(buf.data(), nbytes, want, fh);
      if (got == 0) break;
      engine.check(engine.add_codes(buf.data(), got));
      loaded += got;
    }
    fclose(fh);
    engine.check(engine.build_index());

    vc::SearchWorker worker(&engine, (int)loaded);
    uint64_t n_main_reads = 0, n_sub_reads = 0, n_local_reads = 0;
    uint32_t radius = 0;

    if (query_id >= 0) {
      vc::image_search_client client(&engine);
      for (const auto& r : client.search_image_by_id((uint32_t)query_id, k, approximate))
        std::cout << r.first << " : " << r.second << std::endl;
      return 0;
    }
    if (!query_file) return usage();
    FILE* f = fopen(query_file, "rb");
    if (!f) {
      fprintf(stderr, "Can't open file %s.", query_file);
      return 1;
    }
    uint64_t n_main_total = 0, n_sub_total = 0, n_local_total = 0, radius_total = 0;
    int n_query = 0;
    // distributed_image_search.cc:62-85 searches the file's queries one by one (at most 200, :83-84); here they are read
    // first and searched in ONE batched call -- same results, same statistics, same output lines (VC_QUERY_BY_QUERY=1
    // keeps the reference's one-find-per-query sequence, e.g. to time single-query latency)
    std::vector<char> qbuf;
    std::vector<char> code(nbytes);
    while (n_query < 200 && fread(code.data(), nbytes, 1, f) != 0) {
      qbuf.insert(qbuf.end(), code.begin(), code.end());
      n_query++;
    }
    const bool one_by_one = getenv("VC_QUERY_BY_QUERY") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::list<vc::SearchWorker::search_result_st> > results;
    std::vector<vc_query_stats> stats;
    if (one_by_one) {
      for (int q = 0; q < n_query; ++q) {
        results.push_back(worker.find(qbuf.data() + (size_t)q * nbytes, nbytes, k, approximate));
        vc_query_stats st{};
        worker.get_stat(st.n_main_reads, st.n_sub_reads, st.n_local_reads, st.radius);
        stats.push_back(st);
      }
    } else {
      results = worker.find_batch(qbuf.data(), nbytes, (uint32_t)n_query, k, approximate, &stats);
    }
    for (int q = 0; q < n_query; ++q) {
      n_main_reads = stats[q].n_main_reads; n_sub_reads = stats[q].n_sub_reads; n_local_reads = stats[q].n_local_reads; radius = stats[q].radius;
      n_main_total += n_main_reads;
      n_local_total += n_local_reads;
      n_sub_total += n_sub_reads;
      radius_total += radius;
      if (print_results) {
        std::cout << "query " << q << std::endl;
        for (const auto& r : results[q]) std::cout << r.image_id << " : " << r.dist << std::endl;
        std::cout << "stat n_main_reads : " << n_main_reads << " , n_sub_reads : " << n_sub_reads
                  << ", n_local_reads : " << n_local_reads << ", radius : " << radius << std::endl;
      }
    }
    fclose(f);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (n_query) {   // :87-93 (rank 0's line; there is one process now)
      std::cout << "Averate result : " << std::endl;
      std::cout << 0 << "  n_main_reads : " << n_main_total / n_query;
      std::cout << " , n_sub_reads : " << n_sub_total / n_query << ", ";
      std::cout << "n_local_reads : " << n_local_total / n_query << ", radius : " << radius_total / n_query << ", ";
      std::cout << "rdma : " << 0 << std::endl;
      std::cout << "while : " << secs << " s, " << n_query << " queries" << std::endl;   // timer "while" (:60)
    }
  } catch (const vc::EngineError& e) {
    fprintf(stderr, "verticut_gpu error %d: %s\n", e.code(), e.what());
    return 1;
  }
  return 0;
}
