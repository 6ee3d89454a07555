// ============================================================================
// vc_engine.hip -- host side of the C ABI declared in include/verticut_gpu.h.
// Owns the HBM-resident code columns, the per-call work buffers and the launch sequence.
// No CPU compute path exists here: every search runs on the device or fails.
// ============================================================================
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

#include "vc_internal.hpp"
#include "vc_mih.hpp"

struct vc_engine {
  vc_config cfg;
  int device = 0;
  uint32_t bits = 0, W = 0, m = 0, sbits = 0, n_cu = 0;
  uint32_t cap = 65536, qtile = 32, scan_blocks = 0;
  bool qtile_auto = true;        // vc_config.query_tile == 0 and no VC_QUERY_TILE: the tile follows the database size (linear_tile)
  uint64_t n = 0, stride = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  uint64_t* d_cols = nullptr;

  // grow-only work buffers
  void* d_stage = nullptr;      size_t stage_bytes = 0;   // ingest staging / query upload
  uint8_t* h_pin = nullptr;     size_t pin_bytes = 0;     // pinned host staging of the host-pointer search calls (queries | rows | counts)
  uint8_t* h_pipe = nullptr;    hipEvent_t pipe_ev[2] = {nullptr, nullptr};   // two pinned chunks: large result sets on their way to pageable memory
  uint64_t* d_q = nullptr;      size_t q_bytes = 0;       // queries [nq][W]
  uint32_t* d_state = nullptr;  size_t state_bytes = 0;   // per tile: count | hist | shist | tau
  uint64_t* d_ring = nullptr;   size_t ring_bytes = 0;    // per tile: [qt][cap]
  uint64_t* d_out = nullptr;    size_t out_bytes = 0;     // [nq][k]
  uint32_t* d_cnt = nullptr;    size_t cnt_bytes = 0;     // [nq] result counts | [nq] raw ring counts
  // exact-MIH cost-model switch (mih_scan_fallback): gathered queries, their scan rows / counts / settled flags
  uint64_t* d_fq = nullptr;     size_t fq_bytes = 0;
  uint64_t* d_frows = nullptr;  size_t frows_bytes = 0;
  uint32_t* d_fcnt = nullptr;   size_t fcnt_bytes = 0;
  uint32_t* d_rec = nullptr;                              // scratch of the device-side ring-overflow recovery (zero at first use)
  // the last kernel of a linear step (vc_recover_kernel) hands the per-step state back zeroed: no memset per step
  const uint32_t* clean_ptr = nullptr;  size_t clean_words = 0;
  uint64_t clean_layout = 0;                              // (queries per group, histogram stride) the clean state is laid out for:
                                                          // the threshold lines are "clean" at ~0, everything else at 0
  uint32_t scan_event_tick = 0;                           // VC_FLAG_LEAN_TIMING: only every timing_sample-th verify launch is timed
  VcKnobs knobs;                                          // environment knobs, read once at vc_create
  uint32_t recover_sabotage = 0;                          // test knob VC_RECOVER_TEST_FAIL: recover launches still to be made to give up

  // timing: event pairs recorded since the last vc_get_timing (calls: whole search calls, scans: verify launches)
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_calls, ev_scans;
  hipEvent_t cur_t0 = nullptr;
  // All search calls share the engine's work buffers, so a call must run after the previous one.  On the same stream
  // that is stream order; when the stream changes, the new call records an event at the tail of the previous stream
  // and waits for it (lazily: the common same-stream loop pays no event at all).
  hipEvent_t last_call = nullptr;
  hipStream_t last_stream = nullptr;
  bool last_stream_valid = false;
  uint64_t scan_bytes = 0;
  vc_timing last{};

  VcMihIndex* mih = nullptr;
  VcRadiusWork radius_work;
  std::string err;
};

static thread_local std::string g_create_err;

static int fail(vc_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_err = buf;
  return code;
}

#define VC_HIP(e, call)                                                                         \
  do {                                                                                          \
    hipError_t _r = (call);                                                                     \
    if (_r != hipSuccess)                                                                       \
      return fail(e, _r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP, "%s: %s (%s:%d)", #call, \
                  hipGetErrorString(_r), __FILE__, __LINE__);                                   \
  } while (0)

template <class T>
static int grow(vc_engine* e, T** p, size_t* have, size_t need) {
  if (need <= *have) return VC_OK;
  if (*p) VC_HIP(e, hipFree(*p));
  *p = nullptr;
  *have = 0;
  need = (need + 255) & ~(size_t)255;
  VC_HIP(e, hipMalloc((void**)p, need));
  *have = need;
  return VC_OK;
}

// environment knobs (developer / test switches): read here, once per engine, never on a launch path
static void read_knobs(VcKnobs* k) {
  if (const char* w = getenv("VC_SCAN_WRAP")) k->scan_wrap = (uint32_t)atoi(w);
  if (const char* w = getenv("VC_SCAN_DIAG")) k->scan_diag = (uint32_t)atoi(w);
  if (const char* w = getenv("VC_SCAN_TRACE")) k->scan_trace = atoi(w) != 0;
  if (const char* w = getenv("VC_SCAN_RESIDENT_MB")) k->resident_mb = std::max(0, atoi(w));
  if (const char* s2 = getenv("VC_SAMPLE2")) { k->sample2_set = true; k->sample2 = strtoull(s2, nullptr, 10); }
  if (const char* s1 = getenv("VC_SAMPLE1")) { k->sample1_set = true; k->sample1 = strtoull(s1, nullptr, 10); }
  if (const char* sh = getenv("VC_SCAN_SHAPE")) {
    int u = 0, b = 0, d = 2;
    if (sscanf(sh, "%d,%d,%d", &u, &b, &d) >= 1) { k->shape_set = true; k->shape_u = u; k->shape_blk = b; k->shape_db = d; }
  }
  if (const char* g = getenv("VC_SAMPLE_BLOCKS_PER_CU")) k->sample_blocks_per_cu = (uint32_t)std::max(1, atoi(g));
  k->recover_trace = getenv("VC_RECOVER_TRACE") != nullptr;
  k->mih_trace = getenv("VC_MIH_TRACE") != nullptr;
  if (const char* r = getenv("VC_DEVICE_RECOVER")) k->device_recover = atoi(r) != 0;
  if (const char* v = getenv("VC_MIH_BCODES")) k->mih_bcodes = atoi(v) != 0;
  if (const char* v = getenv("VC_MIH_BENT")) k->mih_bent = atoi(v) != 0;
  if (const char* v = getenv("VC_MIH_HOST_LOOP")) k->mih_host_loop = atoi(v);
  if (const char* v = getenv("VC_SCAN_SMALL")) k->scan_small = atoi(v);
  if (const char* v = getenv("VC_MIH_BUDGET")) k->mih_budget = strtoull(v, nullptr, 10);
  if (const char* v = getenv("VC_TAU_FOLD")) k->tau_fold = atoi(v);
  if (const char* v = getenv("VC_MIH_STREAM")) k->mih_stream = atoi(v);
  if (const char* v = getenv("VC_MIH_SWITCH")) k->mih_switch = atoi(v);
  if (const char* v = getenv("VC_MIH_PHASES")) k->mih_phases = atoi(v);
  if (const char* v = getenv("VC_MIH_GROUP")) k->mih_group = atoi(v);
  if (const char* v = getenv("VC_MIH_POLL")) k->mih_poll = atoi(v);
  if (const char* v = getenv("VC_MIH_LINES")) k->mih_lines = atoi(v);
  if (const char* v = getenv("VC_MIH_ORDER")) k->mih_order = atoi(v);
  if (const char* v = getenv("VC_MIH_QTILE")) k->mih_qtile = atoi(v);
  if (const char* v = getenv("VC_RECOVER_SPIN_LIMIT")) k->recover_spin_limit = (uint32_t)strtoul(v, nullptr, 10);
  if (const char* v = getenv("VC_RECOVER_TEST_FAIL")) k->recover_test_fail = (uint32_t)strtoul(v, nullptr, 10);
}

static int bind_device(vc_engine* e) {
  VC_HIP(e, hipSetDevice(e->device));
  return VC_OK;
}

extern "C" {

int vc_abi_version(void) { return VC_ABI_VERSION; }

const char* vc_strerror(int code) {
  switch (code) {
    case VC_OK: return "ok";
    case VC_NOT_FOUND: return "not found";
    case VC_ERR_INVALID: return "invalid argument";
    case VC_ERR_NO_DEVICE: return "no usable gfx950 device";
    case VC_ERR_HIP: return "HIP runtime error";
    case VC_ERR_NOMEM: return "out of device memory";
    case VC_ERR_STATE: return "call made in the wrong state";
    case VC_ERR_CAPACITY: return "capacity exceeded";
  }
  return "unknown error";
}

const char* vc_last_error(const vc_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int vc_create(const vc_config* cfg, vc_engine** out) {
  if (!cfg || !out) return fail(nullptr, VC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != VC_ABI_VERSION) return fail(nullptr, VC_ERR_INVALID, "abi_version %u != %u", cfg->abi_version, VC_ABI_VERSION);
  const uint32_t B = cfg->bits;
  if (!(B == 64 || B == 128 || B == 256 || B == 512)) return fail(nullptr, VC_ERR_INVALID, "bits must be 64/128/256/512");
  uint32_t sbits = 0;
  if (cfg->n_tables) {  // search_worker.cc:75 assert(nbytes % size == 0); binaryToInt handles <= 4 bytes
    if ((B / 8) % cfg->n_tables) return fail(nullptr, VC_ERR_INVALID, "code bytes not divisible by n_tables");
    sbits = B / cfg->n_tables;
    if (sbits < 8 || sbits > 32 || sbits % 8) return fail(nullptr, VC_ERR_INVALID, "substring must be 8..32 bits, multiple of 8");
  }
  if (cfg->capacity == 0 || cfg->capacity + (uint64_t)cfg->id_base > 0x100000000ull)
    return fail(nullptr, VC_ERR_INVALID, "capacity must be >0 and id_base+capacity <= 2^32 (ids are uint32)");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, VC_ERR_NO_DEVICE, "no HIP device visible");
  int dev = cfg->device;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(nullptr, VC_ERR_NO_DEVICE, "hipGetDevice failed");
  if (dev >= ndev) return fail(nullptr, VC_ERR_NO_DEVICE, "device %d out of range", dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(nullptr, VC_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, VC_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", dev, prop.gcnArchName);

  vc_engine* e = new vc_engine();
  e->cfg = *cfg;
  e->device = dev;
  e->bits = B;
  e->W = B / 64;
  e->m = cfg->n_tables;
  e->sbits = sbits;
  e->n_cu = (uint32_t)prop.multiProcessorCount;
  read_knobs(&e->knobs);
  if ((cfg->flags & VC_FLAG_LEAN_TIMING) && cfg->timing_sample > 1) e->knobs.timing_every = cfg->timing_sample;
  e->recover_sabotage = e->knobs.recover_test_fail;
  e->cap = cfg->cand_cap ? cfg->cand_cap : 65536u;
  // queries verified per database pass: 8 keeps the pass on the HBM side of the roofline (bench), larger tiles trade
  // bandwidth efficiency for queries/s until the popcount VALU ceiling; default for big batches: 32
  e->qtile = cfg->query_tile ? cfg->query_tile : 32u;
  e->qtile_auto = cfg->query_tile == 0;
  if (const char* s = getenv("VC_QUERY_TILE")) { e->qtile = (uint32_t)std::max(1, atoi(s)); e->qtile_auto = false; }
  e->scan_blocks = cfg->scan_blocks;
  if (const char* s = getenv("VC_SCAN_BLOCKS")) e->scan_blocks = (uint32_t)std::max(1, atoi(s));
  e->stride = (cfg->capacity + VC_PAD_ITEMS - 1) / VC_PAD_ITEMS * VC_PAD_ITEMS;

  int rc = bind_device(e);
  if (rc == VC_OK) {
    hipError_t r = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    // Column stride.  The verify kernel streams the W columns side by side, and how well the memory system keeps up
    // depends on the distance between the columns and on where the allocation landed: at 1e9 x 128-bit codes the
    // same kernel streams anywhere from 6.4 to 7.0 TB/s (tools/placement_probe3.py: a stride of exactly 2^30 items
    // is always fast there, 9 * 2^27 items is always slow at 1.2e9 codes, most strides are a lottery per allocation).
    // Nothing about the address hashing is documented, so a big multi-column database is allocated with room for a
    // few candidate strides and each is timed with a streaming read in the verify kernel's access pattern.
    std::vector<uint64_t> cand{e->stride};
    const bool stride_trace = getenv("VC_STRIDE_TRACE") != nullptr;
    const uint64_t col_min = e->stride * sizeof(uint64_t);
    int tries = 6;
    if (const char* t = getenv("VC_STRIDE_TRIES")) tries = std::max(1, atoi(t));   // 1 = take the first
    if (e->W >= 2 && col_min >= (256ull << 20) && tries > 1) {
      uint64_t p2 = VC_PAD_ITEMS;
      while (p2 < cfg->capacity) p2 <<= 1;
      if (p2 - e->stride <= e->stride / 12) cand.push_back(p2);                       // power of two, if it costs < 8 %
      for (int j = 1; (int)cand.size() < tries; ++j) cand.push_back(e->stride + (uint64_t)j * 4 * VC_PAD_ITEMS);   // + j * 256 KiB
    }
    if (const char* f = getenv("VC_STRIDE_FORCE")) {   // test knob: a given stride (items, multiple of 8192, >= capacity)
      const uint64_t fs = strtoull(f, nullptr, 10);
      if (fs >= e->stride && fs % VC_PAD_ITEMS == 0) {
        cand.assign(1, fs);
        e->stride = fs;
      }
    }
    const uint64_t stride_max = *std::max_element(cand.begin(), cand.end());
    const size_t col_bytes = stride_max * e->W * sizeof(uint64_t);
    if (r == hipSuccess) r = hipMalloc((void**)&e->d_cols, col_bytes);
    if (r == hipSuccess) r = hipMemsetAsync(e->d_cols, 0, col_bytes, e->own_stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->own_stream);
    if (r == hipSuccess && cand.size() > 1) {
      uint64_t* d_sink = nullptr;
      if (hipMalloc((void**)&d_sink, 8) == hipSuccess) {
        float best = -1.f;
        for (uint64_t st : cand) {
          const float ms = vc_probe_stream_ms(e->d_cols, st, e->W, cfg->capacity, d_sink, e->n_cu, e->own_stream);
          if (stride_trace) fprintf(stderr, "[vc stride] %llu items: %.3f ms\n", (unsigned long long)st, ms);
          if (ms > 0 && (best < 0 || ms < best)) {
            best = ms;
            e->stride = st;
          }
        }
        (void)hipFree(d_sink);
      }
    }
    if (r != hipSuccess) rc = fail(nullptr, r == hipErrorOutOfMemory ? VC_ERR_NOMEM : VC_ERR_HIP, "engine setup: %s", hipGetErrorString(r));
  } else {
    g_create_err = e->err;
  }
  if (rc != VC_OK) {
    vc_destroy(e);
    return rc;
  }
  e->stream = e->own_stream;
  *out = e;
  return VC_OK;
}

int vc_destroy(vc_engine* e) {
  if (!e) return VC_OK;
  (void)hipSetDevice(e->device);
  if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
  if (e->mih) vc_mih_free(e->mih);
  vc_radius_work_free(&e->radius_work);
  (void)hipFree(e->d_cols);
  (void)hipFree(e->d_stage);
  if (e->h_pin) (void)hipHostFree(e->h_pin);
  if (e->h_pipe) (void)hipHostFree(e->h_pipe);
  for (hipEvent_t ev : e->pipe_ev)
    if (ev) (void)hipEventDestroy(ev);
  (void)hipFree(e->d_q);
  (void)hipFree(e->d_state);
  (void)hipFree(e->d_ring);
  (void)hipFree(e->d_out);
  (void)hipFree(e->d_cnt);
  (void)hipFree(e->d_rec);
  (void)hipFree(e->d_fq);
  (void)hipFree(e->d_frows);
  (void)hipFree(e->d_fcnt);
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->last_call) (void)hipEventDestroy(e->last_call);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
  return VC_OK;
}

int vc_set_stream(vc_engine* e, void* stream) {
  if (!e) return VC_ERR_INVALID;
  e->stream = stream == VC_STREAM_OWN ? e->own_stream : (hipStream_t)stream;   // NULL = the HIP null stream
  return VC_OK;
}

int vc_size(const vc_engine* e, uint64_t* n) {
  if (!e || !n) return VC_ERR_INVALID;
  *n = e->n;
  return VC_OK;
}

// ---- ingest (build_hash_tables.cc:40-70: append in file order, id = ordinal) ---------------------
int vc_add_codes(vc_engine* e, const void* codes, uint64_t n) {
  if (!e || (!codes && n)) return VC_ERR_INVALID;
  if (e->n + n > e->cfg.capacity) return fail(e, VC_ERR_CAPACITY, "add %llu codes: %llu + n > capacity %llu", (unsigned long long)n, (unsigned long long)e->n, (unsigned long long)e->cfg.capacity);
  int rc = bind_device(e);
  if (rc) return rc;
  const size_t rec = e->bits / 8;
  const uint64_t batch = std::max<uint64_t>(1, (64ull << 20) / rec);
  const uint8_t* src = (const uint8_t*)codes;
  for (uint64_t off = 0; off < n; off += batch) {
    const uint64_t cnt = std::min(batch, n - off);
    if ((rc = grow(e, (uint8_t**)&e->d_stage, &e->stage_bytes, cnt * rec))) return rc;
    VC_HIP(e, hipMemcpyAsync(e->d_stage, src + off * rec, cnt * rec, hipMemcpyHostToDevice, e->stream));
    VC_HIP(e, vc_launch_rows_to_cols((const uint64_t*)e->d_stage, e->d_cols, e->stride, e->W, e->n + off, cnt, e->stream));
    VC_HIP(e, hipStreamSynchronize(e->stream));
  }
  e->n += n;
  if (e->mih) { vc_mih_free(e->mih); e->mih = nullptr; }
  return VC_OK;
}

int vc_add_synthetic(vc_engine* e, uint64_t n, uint64_t seed, uint32_t kind, uint32_t n_centres, uint32_t max_flips) {
  if (!e || kind > VC_SYNTH_CLUSTERED) return VC_ERR_INVALID;
  if (e->n + n > e->cfg.capacity) return fail(e, VC_ERR_CAPACITY, "add_synthetic beyond capacity");
  int rc = bind_device(e);
  if (rc) return rc;
  VC_HIP(e, vc_launch_fill_synth(e->d_cols, e->stride, e->W, e->n, n, (uint64_t)e->cfg.id_base + e->n, seed, kind, n_centres, max_flips, e->stream));
  VC_HIP(e, hipStreamSynchronize(e->stream));
  e->n += n;
  if (e->mih) { vc_mih_free(e->mih); e->mih = nullptr; }
  return VC_OK;
}

// ---- the reference's file formats ------------------------------------------------------------------
int vc_load_code_file(vc_engine* e, const char* path, uint64_t max_records, uint64_t* n_read) {
  if (!e || !path) return VC_ERR_INVALID;
  FILE* fh = fopen(path, "rb");
  if (!fh) return fail(e, VC_ERR_INVALID, "Can't open file %s.", path);   // build_hash_tables.cc:30-33
  const size_t rec = e->bits / 8;
  const size_t batch = std::max<size_t>(1, (64u << 20) / rec);
  std::vector<char> buf(batch * rec);
  uint64_t total = 0;
  int rc = VC_OK;
  while (max_records == 0 || total < max_records) {
    const size_t want = max_records ? (size_t)std::min<uint64_t>(batch, max_records - total) : batch;
    const size_t got = fread(buf.data(), rec, want, fh);   // a trailing partial record is ignored, as fread(...,1,fh) does
    if (got == 0) break;
    if ((rc = vc_add_codes(e, buf.data(), got))) break;
    total += got;
  }
  fclose(fh);
  if (n_read) *n_read = total;
  return rc;
}

int vc_save_code_file(vc_engine* e, const char* path) {
  if (!e || !path) return VC_ERR_INVALID;
  int rc = bind_device(e);
  if (rc) return rc;
  FILE* fh = fopen(path, "wb");
  if (!fh) return fail(e, VC_ERR_INVALID, "Can't create file %s.", path);
  const size_t rec = e->bits / 8;
  const uint64_t batch = std::max<uint64_t>(1, (64ull << 20) / rec);
  std::vector<char> buf(batch * rec);
  std::vector<uint32_t> ids(batch);
  for (uint64_t off = 0; off < e->n; off += batch) {
    const uint32_t cnt = (uint32_t)std::min<uint64_t>(batch, e->n - off);
    if ((rc = grow(e, (uint8_t**)&e->d_stage, &e->stage_bytes, (size_t)cnt * (rec + 4)))) break;
    uint32_t* d_ids = (uint32_t*)((uint8_t*)e->d_stage + (size_t)cnt * rec);
    for (uint32_t i = 0; i < cnt; ++i) ids[i] = (uint32_t)(off + i);
    hipError_t r = hipMemcpyAsync(d_ids, ids.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, e->stream);
    if (r == hipSuccess) r = vc_launch_gather_rows(e->d_cols, e->stride, e->W, d_ids, cnt, (uint64_t*)e->d_stage, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(buf.data(), e->d_stage, (size_t)cnt * rec, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    if (r != hipSuccess) { rc = fail(e, VC_ERR_HIP, "save: %s", hipGetErrorString(r)); break; }
    if (fwrite(buf.data(), rec, cnt, fh) != cnt) { rc = fail(e, VC_ERR_INVALID, "short write to %s", path); break; }
  }
  fclose(fh);
  return rc;
}

int vc_write_bitmap_file(vc_engine* e, uint32_t table, const char* path) {
  if (!e || !path) return VC_ERR_INVALID;
  if (!e->mih) return fail(e, VC_ERR_STATE, "no index built");
  if (table >= e->m) return VC_ERR_INVALID;
  int rc = bind_device(e);
  if (rc) return rc;
  FILE* fh = fopen(path, "wb");
  if (!fh) return fail(e, VC_ERR_INVALID, "Can't create file %s.", path);
  const uint64_t words = (1ull << e->sbits) / 32;
  const uint64_t batch = 16ull << 20;   // 64 MB of words per round
  std::vector<uint32_t> buf((size_t)std::min(words, batch));
  for (uint64_t off = 0; off < words && rc == VC_OK; off += batch) {
    const uint64_t cnt = std::min(batch, words - off);
    rc = vc_mih_bitmap_read(e->mih, table, off, cnt, buf.data(), e->stream, &e->err);
    if (rc == VC_OK && fwrite(buf.data(), 4, cnt, fh) != cnt) rc = fail(e, VC_ERR_INVALID, "short write to %s", path);
  }
  fclose(fh);
  return rc;
}

// bitmap_deamon.cc:41-65 reads the files generate_bitmap.cc:99-125 wrote back into memory; here the file is checked word
// for word against the bitmap the index derived from the resident records (the two are equal exactly when the file was
// generated from the same code file), which is what "attaching" it means for an index whose rank directory depends on it
int vc_read_bitmap_file(vc_engine* e, uint32_t table, const char* path, uint64_t* n_mismatch_words) {
  if (!e || !path) return VC_ERR_INVALID;
  if (n_mismatch_words) *n_mismatch_words = 0;
  if (!e->mih) return fail(e, VC_ERR_STATE, "no index built");
  if (table >= e->m) return VC_ERR_INVALID;
  int rc = bind_device(e);
  if (rc) return rc;
  FILE* fh = fopen(path, "rb");
  if (!fh) return fail(e, VC_ERR_INVALID, "Can't open file %s.", path);   // bitmap_deamon.cc:48-51
  const uint64_t words = (1ull << e->sbits) / 32;
  const uint64_t batch = 16ull << 20;
  std::vector<uint32_t> file_w((size_t)std::min(words, batch)), dev_w((size_t)std::min(words, batch));
  uint64_t bad = 0;
  for (uint64_t off = 0; off < words && rc == VC_OK; off += batch) {
    const uint64_t cnt = std::min(batch, words - off);
    if (fread(file_w.data(), 4, cnt, fh) != cnt) { rc = fail(e, VC_ERR_INVALID, "%s is shorter than 2^%u bits", path, e->sbits); break; }
    rc = vc_mih_bitmap_read(e->mih, table, off, cnt, dev_w.data(), e->stream, &e->err);
    for (uint64_t i = 0; i < cnt && rc == VC_OK; ++i) bad += file_w[i] != dev_w[i];
  }
  if (rc == VC_OK && fgetc(fh) != EOF) rc = fail(e, VC_ERR_INVALID, "%s is longer than 2^%u bits", path, e->sbits);
  fclose(fh);
  if (n_mismatch_words) *n_mismatch_words = bad;
  if (rc == VC_OK && bad) rc = fail(e, VC_ERR_STATE, "%s differs from the bitmap of the resident records in %llu words", path, (unsigned long long)bad);
  return rc;
}

int vc_save_index(vc_engine* e, const char* path) {
  if (!e || !path) return VC_ERR_INVALID;
  if (!e->mih) return fail(e, VC_ERR_STATE, "no index built");
  int rc = bind_device(e);
  if (rc) return rc;
  return vc_mih_save(e->mih, e->d_cols, e->stride, path, e->stream, &e->err);
}

int vc_load_index(vc_engine* e, const char* path) {
  if (!e || !path) return VC_ERR_INVALID;
  if (e->m == 0) return fail(e, VC_ERR_STATE, "engine was created with n_tables = 0 (linear only)");
  int rc = bind_device(e);
  if (rc) return rc;
  if (e->mih) { vc_mih_free(e->mih); e->mih = nullptr; }
  return vc_mih_load(&e->mih, path, e->d_cols, e->stride, e->n, e->W, e->m, e->sbits, e->cfg.id_base, e->cfg.flags, e->n_cu, e->cap,
                     e->knobs, e->stream, &e->err);
}

int vc_get_code(vc_engine* e, uint32_t id, void* out) {
  if (!e || !out) return VC_ERR_INVALID;
  if (id < e->cfg.id_base || (uint64_t)id - e->cfg.id_base >= e->n) return VC_NOT_FOUND;
  int rc = bind_device(e);
  if (rc) return rc;
  const uint64_t local = id - e->cfg.id_base;
  uint64_t w[VC_MAX_W];
  for (uint32_t j = 0; j < e->W; ++j)
    VC_HIP(e, hipMemcpyAsync(&w[j], e->d_cols + j * e->stride + local, 8, hipMemcpyDeviceToHost, e->stream));
  VC_HIP(e, hipStreamSynchronize(e->stream));
  memcpy(out, w, e->bits / 8);
  return VC_OK;
}

// ---- timing helpers -----------------------------------------------------------------------------
// Events live in a grow-only pool and are handed out in pairs; vc_get_timing() sums every pair recorded since
// the previous vc_get_timing() and recycles them.  Beyond VC_MAX_TIMED pairs recording stops (sums stay valid).
#define VC_MAX_TIMED 8192
static hipEvent_t ev_take(vc_engine* e) {
  if (e->ev_used >= 2 * VC_MAX_TIMED) return nullptr;
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}
static int ev_pair(vc_engine* e, hipEvent_t* a, hipEvent_t* b) {
  *a = ev_take(e);
  *b = *a ? ev_take(e) : nullptr;
  if (*a && !*b) { --e->ev_used; *a = nullptr; }
  return VC_OK;
}
static void timing_begin(vc_engine* e) {
  if (e->last_stream_valid && e->last_stream != e->stream) {
    if (!e->last_call && hipEventCreateWithFlags(&e->last_call, hipEventDisableTiming) != hipSuccess) e->last_call = nullptr;
    // the previous stream may be gone by now (the caller's to destroy): then everything it held has been submitted
    // and a device-wide wait is the safe fallback
    if (!e->last_call || hipEventRecord(e->last_call, e->last_stream) != hipSuccess ||
        hipStreamWaitEvent(e->stream, e->last_call, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
    }
  }
  e->cur_t0 = (e->cfg.flags & VC_FLAG_LEAN_TIMING) ? nullptr : ev_take(e);
  if (e->cur_t0) (void)hipEventRecord(e->cur_t0, e->stream);
}
static void timing_end(vc_engine* e) {
  e->last_stream = e->stream;
  e->last_stream_valid = true;
  if (!e->cur_t0) return;
  hipEvent_t t1 = ev_take(e);
  if (!t1) { --e->ev_used; e->cur_t0 = nullptr; return; }
  (void)hipEventRecord(t1, e->stream);
  e->ev_calls.emplace_back(e->cur_t0, t1);
  e->cur_t0 = nullptr;
}

int vc_get_timing(const vc_engine* ce, vc_timing* t) {
  vc_engine* e = const_cast<vc_engine*>(ce);
  if (!e || !t) return VC_ERR_INVALID;
  if (!e->ev_calls.empty() || !e->ev_scans.empty() || e->mih) {
    int rc = bind_device(e);
    if (rc) return rc;
    vc_timing lt{};
    if (e->mih) {
      uint64_t tot[4];
      vc_mih_timing(e->mih, &lt.mih_ms, &lt.mih_launches, tot, e->stream);
      lt.mih_probes = tot[0]; lt.mih_hits = tot[1]; lt.mih_entries = tot[2]; lt.mih_queries = tot[3];
    }
    for (auto& pr : e->ev_calls) {
      float ms = 0;
      VC_HIP(e, hipEventSynchronize(pr.second));
      (void)hipEventElapsedTime(&ms, pr.first, pr.second);
      lt.total_ms += ms;
      lt.calls++;
    }
    for (auto& pr : e->ev_scans) {
      float ms = 0;
      VC_HIP(e, hipEventSynchronize(pr.second));
      (void)hipEventElapsedTime(&ms, pr.first, pr.second);
      lt.scan_ms += ms;
      lt.scan_launches++;
    }
    lt.scan_bytes = e->scan_bytes;
    e->last = lt;
    e->ev_calls.clear();
    e->ev_scans.clear();
    e->ev_used = 0;
    e->scan_bytes = 0;
  }
  *t = e->last;
  return VC_OK;
}

// ---- LINEAR: linear_search.cc:39-64 for a batch of queries ------------------------------------------
struct LinearBufs {
  uint32_t hs, QT, GQ, cap;   // histogram stride, queries per tile (one verify launch), per group (one bootstrap), ring entries
  size_t state_words;
  uint32_t *d_count, *d_hist, *d_shist, *d_shist2, *d_tau;
};

// Queries whose bootstrap is done by ONE pair of sampling launches: the tiles of a group share them (the stage
// kernels read the sampled prefix once per 32 queries instead of once per tile) and one select launch.
#define VC_GROUP_QUERIES 64u
#define VC_RESIDENT_MB_DEFAULT 240   // of the 256 MB Infinity Cache (profiles/r02_sweeps.md: 224-256 MB best, 320 MB thrashes)

// Queries per database pass when the caller left it to the engine (vc_config.query_tile = 0).  A pass over a BIG database is
// priced by its bytes and 32 queries keep it near the VALU / HBM balance point; a pass over a small one is priced by its
// launches (~50 us of bootstrap / verify / select / recover whatever it reads), so the tile grows as the database shrinks:
// 32 from 256 MB on, doubling per halving below, at most 512 -- configs[0] (8 MB, 200 queries per call) runs in ONE pass
// instead of seven: 0.42 -> 0.92 M queries/s (1.3 M with the lane tile of vc_scan_pick_shape fitted to it as well).
static uint32_t linear_tile(const vc_engine* e) {
  if (!e->qtile_auto) return e->qtile;
  const uint64_t bytes = std::max<uint64_t>(e->n, 1) * (e->bits / 8);
  uint32_t t = 32;
  for (uint64_t b = bytes; b < ((uint64_t)256 << 20) && t < 512; b <<= 1) t <<= 1;
  return t;
}

static int linear_bufs(vc_engine* e, uint32_t nq, uint32_t k, LinearBufs* b) {
  b->hs = (e->bits + 1 + 7) & ~7u;
  b->QT = std::min(linear_tile(e), nq);
  b->GQ = std::min(nq, std::max(b->QT, VC_GROUP_QUERIES / b->QT * b->QT));   // whole tiles, >= one tile
  b->cap = std::max(e->cap, 4 * k);
  // count[GQ] and tau[GQ] hold one 128-byte line per query (VC_QUERY_LINE_WORDS); the histograms are dense
  const size_t hist_words = (((size_t)b->GQ * (1 + 2 * (size_t)VC_SHIST_COPIES) * b->hs) + 31) & ~(size_t)31;   // keeps tau[] line-aligned
  b->state_words = (size_t)b->GQ * 2 * VC_QUERY_LINE_WORDS + hist_words;
  int rc;
  if ((rc = grow(e, &e->d_state, &e->state_bytes, b->state_words * 4))) return rc;
  if ((rc = grow(e, &e->d_ring, &e->ring_bytes, (size_t)b->GQ * b->cap * 8))) return rc;
  if (!e->d_rec && e->knobs.device_recover) {
    const size_t bytes = vc_recover_scratch_words() * 4;
    VC_HIP(e, hipMalloc((void**)&e->d_rec, bytes));
    VC_HIP(e, hipMemsetAsync(e->d_rec, 0, bytes, e->stream));
  }
  b->d_count = e->d_state;
  b->d_hist = b->d_count + (size_t)b->GQ * VC_QUERY_LINE_WORDS;
  b->d_shist = b->d_hist + (size_t)b->GQ * b->hs;
  b->d_shist2 = b->d_shist + (size_t)VC_SHIST_COPIES * b->GQ * b->hs;
  b->d_tau = b->d_hist + hist_words;
  return VC_OK;
}

// one verify launch for a tile whose tau is already set (the tile's state starts at query t0 of the group); d_limit may be null
static int scan_tile(vc_engine* e, const LinearBufs& b, const uint64_t* dq, uint32_t qt, uint32_t k, const uint64_t* d_limit,
                     uint32_t t0 = 0, const uint32_t* d_shist = nullptr, uint64_t shist_cstride = 0) {
  // (the shape follows a small database only where the TILE does too -- query_tile left to the engine -- and only for tiles
  // beyond the small-tile form: explicit tiles and <= 8 queries keep the headline's kernel, on any database)
  const uint64_t shape_n = (e->qtile_auto && qt > 8) ? e->n : 0;
  const VcScanShape sh = vc_scan_pick_shape(e->W, qt, nullptr, &e->knobs, shape_n);
  VcScanParams p{};
  p.cols = e->d_cols;
  p.stride = e->stride;
  p.n = e->n;
  p.nchunks = (e->n + sh.chunk_items() - 1) / sh.chunk_items();
  p.id_base = e->cfg.id_base;
  p.qt = qt;
  p.k = k;
  p.cap = b.cap;
  p.hist_stride = b.hs;
  p.queries = dq;
  p.qs = VC_QUERY_LINE_WORDS;
  p.tau = b.d_tau + (size_t)t0 * VC_QUERY_LINE_WORDS;
  p.count = b.d_count + (size_t)t0 * VC_QUERY_LINE_WORDS;
  p.hist = b.d_hist + (size_t)t0 * b.hs;
  p.buf = e->d_ring + (size_t)t0 * b.cap;
  p.limit = d_limit;
  p.shist = d_shist;             // set: the kernel's prologue cuts the bootstrap histograms itself (no vc_tau_init_kernel)
  p.shist_cstride = shist_cstride;
  p.shist_copies = VC_SHIST_COPIES;
  p.bits = e->bits;
  p.wrap = e->knobs.scan_wrap;   // diagnostic build only, results are wrong by design
  p.diag = e->knobs.scan_diag;
  {   // Infinity-Cache-resident prefix (see the load in vc_scan_kernel)
    const uint64_t mb = e->knobs.resident_mb < 0 ? VC_RESIDENT_MB_DEFAULT : (uint64_t)e->knobs.resident_mb;
    p.resident = (mb << 20) / (sh.chunk_items() * (e->bits / 8));
  }
  hipEvent_t a = nullptr, bb = nullptr;
  // VC_FLAG_LEAN_TIMING with vc_config.timing_sample = N > 1: only every N-th verify launch is bracketed by events
  // (an event record is a barrier packet, ~4-5 us each; a 125 M-code shard step is 0.35 ms)
  const uint32_t every = (e->cfg.flags & VC_FLAG_LEAN_TIMING) ? std::max(e->cfg.timing_sample, 1u) : 1u;
  if (e->scan_event_tick++ % every == 0) ev_pair(e, &a, &bb);
  if (a) VC_HIP(e, hipEventRecord(a, e->stream));
  uint64_t* d_trace = nullptr;
  const uint32_t trace_blocks = 8192;
  if (e->knobs.scan_trace) {   // dev knob, diagnostic build only: when does every block of the persistent grid start and end
    VC_HIP(e, hipMalloc((void**)&d_trace, trace_blocks * 16 + 256));   // + the rare-path counters
    VC_HIP(e, hipMemsetAsync(d_trace, 0, trace_blocks * 16 + 256, e->stream));
    p.trace = d_trace;
  }
  VC_HIP(e, vc_launch_scan(p, e->W, e->n_cu, e->scan_blocks, &e->knobs, e->stream, shape_n));
  if (d_trace) {
    std::vector<uint64_t> h(trace_blocks * 2 + 32);
    VC_HIP(e, hipMemcpyAsync(h.data(), d_trace, trace_blocks * 16 + 256, hipMemcpyDeviceToHost, e->stream));
    VC_HIP(e, hipStreamSynchronize(e->stream));
    {
      const uint32_t* c = (const uint32_t*)(h.data() + trace_blocks * 2);
      fprintf(stderr, "[scan trace] rare path: %u entries, %u of them appended %u items, %u re-cuts (%u queries)\n", c[0], c[1], c[2], c[3], qt);
      fprintf(stderr, "[scan trace] appended items by pass of the chunk loop:");
      for (int i = 0; i < 56; ++i) fprintf(stderr, " %u", c[8 + i]);
      fprintf(stderr, "\n");
    }
    (void)hipFree(d_trace);
    std::vector<double> st, en;
    uint64_t t0 = UINT64_MAX;
    for (uint32_t i = 0; i < trace_blocks; ++i)
      if (h[2 * i] && h[2 * i + 1]) t0 = std::min(t0, h[2 * i]);
    for (uint32_t i = 0; i < trace_blocks; ++i)
      if (h[2 * i] && h[2 * i + 1]) { st.push_back((h[2 * i] - t0) * 0.01); en.push_back((h[2 * i + 1] - t0) * 0.01); }   // 100 MHz -> us
    {   // per XCD (blocks map to XCDs round-robin, blockIdx % 8): when does its last block end, and the mean end of its blocks
      double mx[8] = {}, sum[8] = {};
      uint32_t cnt[8] = {};
      for (uint32_t i = 0; i < trace_blocks; ++i)
        if (h[2 * i] && h[2 * i + 1] && h[2 * i + 1] - t0 < (1ull << 40)) {
          const double en_us = (h[2 * i + 1] - t0) * 0.01;
          mx[i & 7] = std::max(mx[i & 7], en_us); sum[i & 7] += en_us; ++cnt[i & 7];
        }
      fprintf(stderr, "[scan trace] per XCD last end / mean end us:");
      for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f/%.0f", mx[x], cnt[x] ? sum[x] / cnt[x] : 0.0);
      fprintf(stderr, "\n");
    }
    if (!st.empty()) {
      std::sort(st.begin(), st.end());
      std::sort(en.begin(), en.end());
      auto q = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
      fprintf(stderr, "[scan trace] %zu blocks, qt %u: start us min/p50/p90/max %.1f %.1f %.1f %.1f | end us min/p10/p50/p90/p99/max %.1f %.1f %.1f %.1f %.1f %.1f\n",
              st.size(), qt, q(st, 0), q(st, .5), q(st, .9), q(st, 1), q(en, 0), q(en, .1), q(en, .5), q(en, .9), q(en, .99), q(en, 1));
    }
  }
  if (a) {
    VC_HIP(e, hipEventRecord(bb, e->stream));
    e->ev_scans.emplace_back(a, bb);
    e->scan_bytes += e->n * (e->bits / 8);
  }
  return VC_OK;
}

// d_q: [nq][W] words on the device.  Results: d_out [nq][k] ascending (INF padded), d_cnt[0..nq) counts
// (UINT32_MAX == the ring overflowed, the device-side recovery did not run or gave up, and the row is only an upper bound).
typedef std::function<int(uint32_t g0, uint32_t gq, const LinearBufs& b)> LinearHook;
static int linear_batch(vc_engine* e, const uint64_t* d_q, uint32_t nq, uint32_t k, uint64_t* d_out, uint32_t* d_cnt,
                        const LinearHook* after_select = nullptr) {
  LinearBufs b;
  int rc = linear_bufs(e, nq, k, &b);
  if (rc) return rc;
  // Threshold bootstrap: exact distance histogram of the first `sample` codes -> tau (k-th best of the sample), so that
  // the verify kernel's first tiles do not flood the ring / histogram atomics from every wave at once (that flood cost
  // ~0.4 ms per launch while eight queries shared one line for their ring cursors and one for their thresholds).
  // Round 1 ran two stages (64 K codes exactly, then 2 M codes counting only distances <= tau1): 4 launches, 26 us.
  // With one line per query the flood is mild and the threshold of ONE exact stage over 1 M codes starts the verify
  // kernel just as well -- 2 launches, 15 us; a 125 M-code shard step 0.397 -> 0.384 ms, 1e9 unchanged
  // (profiles/r02_sweeps.md; below 512 K codes the verify pass pays for the looser threshold, beyond 1 M nothing is
  // gained -- an exact threshold makes a 1e9 pass no faster).  The refining stage stays selectable (VC_SAMPLE2, tests).
  uint64_t sample = std::min<uint64_t>(e->n, std::max<uint64_t>(std::max<uint64_t>(262144, 64ull * k), std::min<uint64_t>(e->n / 16, 1048576)));
  uint64_t sample2 = 0;
  if (e->knobs.sample2_set) sample2 = std::min<uint64_t>(e->n, e->knobs.sample2);   // dev/test knob VC_SAMPLE2
  if (e->knobs.sample1_set) sample = std::min<uint64_t>(e->n, e->knobs.sample1);      // dev/test knob VC_SAMPLE1
  if (sample2 && !e->knobs.sample1_set) sample = std::min<uint64_t>(sample, std::max<uint64_t>(65536, 64ull * k));   // stage 1 only has to seed stage 2
  for (uint32_t g0 = 0; g0 < nq; g0 += b.GQ) {
    const uint32_t gq = std::min(b.GQ, nq - g0);
    const uint64_t* dg = d_q + (size_t)g0 * e->W;
    // per-step state: zero from the previous step's last kernel, or (first use, new buffer, after an error) memset now
    const uint64_t layout = ((uint64_t)b.GQ << 32) | b.hs;
    const bool clean = e->knobs.device_recover && e->clean_ptr == e->d_state && e->clean_words >= b.state_words && e->clean_layout == layout;
    e->clean_ptr = nullptr;
    if (!clean) {
      VC_HIP(e, hipMemsetAsync(e->d_state, 0, e->state_bytes, e->stream));
      VC_HIP(e, hipMemsetAsync(b.d_tau, 0xFF, (size_t)b.GQ * VC_QUERY_LINE_WORDS * 4, e->stream));   // "no threshold yet"
    }
    // small tiles: the verify kernel's prologue turns the sampled histograms into thresholds itself (one launch less and
    // no coherent read of the threshold lines by every block at start)
    const bool fold = !sample2 && e->knobs.tau_fold && vc_scan_is_small(e->W, std::min(b.QT, gq), &e->knobs) &&
                      (gq % b.QT == 0 || vc_scan_is_small(e->W, gq % b.QT, &e->knobs));
    VC_HIP(e, vc_launch_sample_hist(e->d_cols, e->stride, e->W, sample, dg, gq, b.d_shist, b.hs, k, e->bits, b.d_tau, VC_QUERY_LINE_WORDS, false,
                                    e->n_cu, e->knobs.sample_blocks_per_cu, e->stream, !fold));
    if (sample2)
      VC_HIP(e, vc_launch_sample_hist(e->d_cols, e->stride, e->W, sample2, dg, gq, b.d_shist2, b.hs, k, e->bits, b.d_tau, VC_QUERY_LINE_WORDS, true,
                                      e->n_cu, e->knobs.sample_blocks_per_cu, e->stream));
    for (uint32_t t0 = 0; t0 < gq; t0 += b.QT)
      if ((rc = scan_tile(e, b, dg + (size_t)t0 * e->W, std::min(b.QT, gq - t0), k, nullptr, t0,
                          fold ? b.d_shist + (size_t)t0 * b.hs : nullptr, (uint64_t)gq * b.hs))) return rc;
    VC_HIP(e, vc_launch_select_ring(e->d_ring, b.cap, b.d_count, b.d_tau, VC_QUERY_LINE_WORDS, gq, k, d_out + (size_t)g0 * k,
                                    d_cnt + g0, e->stream));
    // (the exact-MIH switch replays the radius loop's stop rule here: rings and cursors of the group are still intact)
    if (after_select && (rc = (*after_select)(g0, gq, b))) return rc;
    // rows whose ring overflowed (count reported as UINT32_MAX) are recomputed exactly on the device: a no-op launch otherwise
    if (e->knobs.device_recover)
      for (uint32_t c0 = 0; c0 < gq; c0 += 64)   // one launch serves 64 queries (a group is larger only when one tile is: query_tile > 64)
        VC_HIP(e, vc_launch_recover(e->d_cols, e->stride, e->n, e->W, e->cfg.id_base, e->bits, dg + (size_t)c0 * e->W, std::min(64u, gq - c0), k,
                                    e->d_ring + (size_t)c0 * b.cap, b.cap, b.d_count + (size_t)c0 * VC_QUERY_LINE_WORDS,
                                    b.d_hist + (size_t)c0 * b.hs, b.hs, VC_QUERY_LINE_WORDS, e->d_rec, d_out + (size_t)(g0 + c0) * k,
                                    d_cnt + g0 + c0, b.d_tau + (size_t)c0 * VC_QUERY_LINE_WORDS, b.d_shist + (size_t)c0 * b.hs,
                                    (uint64_t)gq * b.hs, VC_SHIST_COPIES, e->n_cu, e->knobs.recover_spin_limit,
                                    e->recover_sabotage ? (e->recover_sabotage--, 1u) : 0u, e->stream));
    // the recover kernel handed the group's state back clean (the 16 partial histograms of a group of gq queries sit
    // gq * hs words apart, which is what it was told); the refining stage's histograms (dev knob) are not covered
    if (e->knobs.device_recover && !sample2) {
      e->clean_ptr = e->d_state;
      e->clean_words = e->state_bytes / 4;
      e->clean_layout = layout;
    }
  }
  return VC_OK;
}

// Ring-overflow recovery (host-driven, rare: more than `cap` items at or below the k-th distance).  The truncated
// ring still yields a valid upper bound on the k-th best packed value (the k-th best of what fitted); the query is
// scanned again appending only packed values <= a probe, and the append counter tells EXACTLY how many items lie at
// or below the probe whether they fitted or not.  Per query an interval (lo, hi] is kept with count(<= lo) < k <=
// count(<= hi):
//   * probe = hi; the k-th best of what fitted becomes the new hi.  With entries arriving in random order that
//     quarters the survivors per round (cap >= 4k); but arrival order is NOT random (the same early waves deliver
//     ids just under the limit round after round: tests/campaign/parity_campaign.py case 259 needed > 64 rounds), so
//   * whenever a round fails to halve the survivors the next probe bisects (lo, hi] instead: fewer than k items
//     below it -> lo = probe; otherwise hi = min(probe, k-th best of what fitted).
// Each bisection halves a 64-bit interval, the other steps never widen it: it ends.
static int linear_recover(vc_engine* e, const uint64_t* d_q, uint32_t k, const std::vector<uint32_t>& over,
                          uint64_t* out /*host [nq][k]*/, uint32_t* cnt /*host [nq]*/) {
  LinearBufs b;
  int rc = linear_bufs(e, (uint32_t)over.size(), k, &b);
  if (rc) return rc;
  e->clean_ptr = nullptr;   // this path memsets the state itself and leaves it used
  const size_t W = e->W;
  uint64_t *d_rq = nullptr, *d_lim = nullptr, *d_rout = nullptr;
  uint32_t* d_rcnt = nullptr;
  auto cleanup = [&]() { (void)hipFree(d_rq); (void)hipFree(d_lim); (void)hipFree(d_rout); (void)hipFree(d_rcnt); };
#define RC(call) do { hipError_t _r = (call); if (_r != hipSuccess) { cleanup(); return fail(e, VC_ERR_HIP, "%s: %s", #call, hipGetErrorString(_r)); } } while (0)
  RC(hipMalloc((void**)&d_rq, b.QT * W * 8));
  RC(hipMalloc((void**)&d_lim, b.QT * 8));
  RC(hipMalloc((void**)&d_rout, (size_t)b.QT * k * 8));
  RC(hipMalloc((void**)&d_rcnt, b.QT * 4));
  struct Rec {
    uint32_t q;          // query index in the caller's batch
    uint64_t lo, hi;     // count(<= lo) < k (valid only if has_lo), count(<= hi) >= k
    bool has_lo;
    uint64_t prev;       // survivors of the previous round (0 = none yet)
    bool bisect;         // this round's probe is the midpoint, not hi
    uint64_t probe;
  };
  std::vector<Rec> todo;
  for (uint32_t q : over) todo.push_back(Rec{q, 0, out[(size_t)q * k + k - 1], false, 0, false, 0});
  const bool trace = e->knobs.recover_trace;   // dev knob VC_RECOVER_TRACE
  for (int round = 0; !todo.empty(); ++round) {
    if (round > 200) { cleanup(); return fail(e, VC_ERR_CAPACITY, "ring overflow recovery did not converge"); }
    std::vector<Rec> next;
    for (size_t t0 = 0; t0 < todo.size(); t0 += b.QT) {
      const uint32_t qt = (uint32_t)std::min<size_t>(b.QT, todo.size() - t0);
      std::vector<uint64_t> lim(qt);
      std::vector<uint32_t> tau(qt);
      for (uint32_t i = 0; i < qt; ++i) {
        Rec& r = todo[t0 + i];
        RC(hipMemcpyAsync(d_rq + i * W, d_q + (size_t)r.q * W, W * 8, hipMemcpyDeviceToDevice, e->stream));
        if (!r.bisect) r.probe = r.hi;
        else if (r.has_lo) r.probe = r.lo + (r.hi - r.lo + 1) / 2;   // rounds up: lo < probe <= hi, so the interval always shrinks
        else if (r.hi >> 32) r.probe = ((r.hi >> 32) << 32) - 1;   // first: everything strictly nearer than hi's distance
        else r.probe = r.hi / 2;
        lim[i] = r.probe;
        tau[i] = (uint32_t)(lim[i] >> 32);
      }
      RC(hipMemsetAsync(e->d_state, 0, b.state_words * 4, e->stream));
      RC(hipMemcpyAsync(d_lim, lim.data(), qt * 8, hipMemcpyHostToDevice, e->stream));
      RC(hipMemcpy2DAsync(b.d_tau, VC_QUERY_LINE_WORDS * 4, tau.data(), 4, 4, qt, hipMemcpyHostToDevice, e->stream));
      if ((rc = scan_tile(e, b, d_rq, qt, k, d_lim))) { cleanup(); return rc; }
      RC(vc_launch_select_ring(e->d_ring, b.cap, b.d_count, b.d_tau, VC_QUERY_LINE_WORDS, qt, k, d_rout, d_rcnt, e->stream));
      std::vector<uint64_t> rout((size_t)qt * k);
      std::vector<uint32_t> rcnt(qt), raw(qt);
      RC(hipMemcpyAsync(rout.data(), d_rout, rout.size() * 8, hipMemcpyDeviceToHost, e->stream));
      RC(hipMemcpyAsync(rcnt.data(), d_rcnt, qt * 4, hipMemcpyDeviceToHost, e->stream));
      RC(hipMemcpy2DAsync(raw.data(), 4, b.d_count, VC_QUERY_LINE_WORDS * 4, 4, qt, hipMemcpyDeviceToHost, e->stream));
      RC(hipStreamSynchronize(e->stream));
      for (uint32_t i = 0; i < qt; ++i) {
        Rec r = todo[t0 + i];
        const uint64_t c = raw[i];                     // exact number of items <= probe
        const uint64_t want = std::min<uint64_t>(k, e->n);
        if (trace && i == 0)
          fprintf(stderr, "[vc recover] round %d query %u: %s probe %016llx -> %llu items (cap %u)\n", round, r.q,
                  r.bisect ? "bisect" : "bound ", (unsigned long long)r.probe, (unsigned long long)c, b.cap);
        if (c < want) {                                // only a bisection probe can undershoot
          r.lo = r.probe;
          r.has_lo = true;
          r.bisect = true;
          next.push_back(r);
          continue;
        }
        // a valid row: everything <= probe was counted, the best min(c, cap) >= k of it was stored
        memcpy(out + (size_t)r.q * k, rout.data() + (size_t)i * k, (size_t)k * 8);
        cnt[r.q] = rcnt[i];
        if (c <= b.cap) continue;                      // nothing was dropped: exact, done
        r.hi = std::min(r.probe, rout[(size_t)i * k + k - 1]);
        r.bisect = r.prev != 0 && c * 2 > r.prev;      // poor progress since the last valid round -> bisect next
        r.prev = c;
        next.push_back(r);
      }
    }
    todo.swap(next);
  }
#undef RC
  cleanup();
  return VC_OK;
}

// Cost-model switch of the exact MIH k-NN loop (VcMihScanFallback, vc_mih.hpp): the listed queries are answered by the
// verify kernel in HBM-bound tiles of 8 and the stop rule of search_worker.cc:201-205 is replayed on each tile's
// candidates (mih_replay_kernel) between the select and the recover launch; with statistics wanted, one more pass over
// the shard counts the items the radius loop would have verified (minimum substring distance <= radius).
static int mih_scan_fallback(void* ctx, const uint64_t* d_q, const uint32_t* d_list, uint32_t n, uint32_t k, uint32_t stop_mult,
                             const VcMihScanTarget& tgt, bool want_stats, uint32_t* d_unresolved, uint32_t* d_n_unresolved, hipStream_t s) {
  vc_engine* e = (vc_engine*)ctx;
  int rc;
  if ((rc = grow(e, &e->d_fq, &e->fq_bytes, (size_t)n * e->W * 8))) return rc;
  if ((rc = grow(e, &e->d_frows, &e->frows_bytes, (size_t)n * k * 8))) return rc;
  if ((rc = grow(e, &e->d_fcnt, &e->fcnt_bytes, (size_t)n * 8))) return rc;
  uint32_t* d_flag = e->d_fcnt + n;
  VC_HIP(e, vc_launch_gather_queries(d_q, d_list, n, e->W, e->d_fq, s));
  const LinearHook hook = [&](uint32_t g0, uint32_t gq, const LinearBufs& b) -> int {
    VcMihReplayArgs a{};
    a.lin_ring = e->d_ring; a.lin_count = b.d_count; a.rows = e->d_frows + (size_t)g0 * k; a.rows_cnt = e->d_fcnt + g0;
    a.queries = e->d_fq + (size_t)g0 * e->W; a.list = d_list + g0; a.cols = e->d_cols; a.stride = e->stride;
    a.lin_cap = b.cap; a.lin_qs = VC_QUERY_LINE_WORDS; a.gq = gq; a.k = k; a.m = e->m; a.sbits = e->sbits; a.W = e->W;
    a.stop_mult = stop_mult; a.id_base = e->cfg.id_base; a.tgt = tgt; a.unresolved = d_unresolved; a.n_unresolved = d_n_unresolved;
    a.resolved_flag = d_flag + g0;
    VC_HIP(e, vc_launch_mih_replay(a, s));
    return VC_OK;
  };
  const uint32_t saved_tile = e->qtile;
  const bool saved_auto = e->qtile_auto;
  // 32 queries per pass: a pass of 8 sits on the HBM roofline, but the switch is asked for QUERIES, and per query the VALU-bound
  // pass of 32 is the cheaper one (8.2 ms per 32 against 2.5 ms per 8 at 1e9 x 128 bit: 3 900 against 3 200 queries/s)
  e->qtile = 32;
  e->qtile_auto = false;
  rc = linear_batch(e, e->d_fq, n, k, e->d_frows, e->d_fcnt, &hook);
  e->qtile = saved_tile;
  e->qtile_auto = saved_auto;
  if (rc) return rc;
  if (want_stats)
    VC_HIP(e, vc_launch_minsub_count(e->d_cols, e->stride, e->n, e->W, e->m, e->sbits, e->d_fq, d_list, d_flag, n, tgt.radius, tgt.seen, e->n_cu, s));
  return VC_OK;
}

static int check_knn_args(vc_engine* e, const void* q, uint32_t nq, uint32_t k, uint32_t mode) {
  if (!e || !q || nq == 0) return VC_ERR_INVALID;
  if (k == 0 || k > VC_MAX_K) return fail(e, VC_ERR_INVALID, "k must be in 1..%u", VC_MAX_K);
  if (mode > VC_MODE_MIH_APPROX) return fail(e, VC_ERR_INVALID, "unknown mode %u", mode);
  if (mode != VC_MODE_LINEAR && !e->mih) return fail(e, VC_ERR_STATE, "MIH search needs vc_build_index() first (n_tables=%u)", e->m);
  return VC_OK;
}

int vc_search_knn_dev(vc_engine* e, const void* d_queries, uint32_t nq, uint32_t k, uint32_t mode, uint64_t* d_out,
                      uint32_t* d_counts, void* stream) {
  return vc_search_knn_dev_stats(e, d_queries, nq, k, mode, d_out, d_counts, nullptr, stream);
}

}  // extern "C"

// get_stat of a linear scan (vc_search_knn's host loop writes the same): every item was a candidate, nothing was probed
__global__ void vc_linear_stats_kernel(const uint32_t* __restrict__ cnt, uint32_t nq, uint64_t n, vc_query_stats* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  vc_query_stats o{};
  o.n_results = cnt[i];
  o.n_candidates = n;
  out[i] = o;
}

extern "C" {

int vc_search_knn_dev_stats(vc_engine* e, const void* d_queries, uint32_t nq, uint32_t k, uint32_t mode, uint64_t* d_out,
                            uint32_t* d_counts, vc_query_stats* d_stats, void* stream) {
  int rc = check_knn_args(e, d_queries, nq, k, mode);
  if (rc) return rc;
  if (!d_out) return VC_ERR_INVALID;
  if ((rc = bind_device(e))) return rc;
  hipStream_t saved = e->stream;
  e->stream = stream == VC_STREAM_OWN ? e->own_stream : (hipStream_t)stream;   // NULL = the HIP null stream
  if ((rc = grow(e, &e->d_cnt, &e->cnt_bytes, (size_t)nq * 8))) { e->stream = saved; return rc; }
  uint32_t* cnt = d_counts ? d_counts : e->d_cnt;
  timing_begin(e);
  if (mode == VC_MODE_LINEAR) {
    rc = linear_batch(e, (const uint64_t*)d_queries, nq, k, d_out, cnt);
    if (rc == VC_OK && d_stats) {
      hipLaunchKernelGGL(vc_linear_stats_kernel, dim3((nq + 255) / 256), dim3(256), 0, e->stream, (const uint32_t*)cnt, nq, e->n, d_stats);
      if (hipGetLastError() != hipSuccess) rc = fail(e, VC_ERR_HIP, "vc_linear_stats_kernel launch failed");
    }
  } else {
    const VcMihScanFallback fb{mih_scan_fallback, e, e->n_cu};
    rc = vc_mih_search(e->mih, e->d_cols, e->stride, e->n, (const uint64_t*)d_queries, nq, k, mode == VC_MODE_MIH_APPROX,
                       d_out, cnt, nullptr, e->stream, &e->err, &fb, d_stats);
  }
  timing_end(e);
  e->stream = saved;
  return rc;
}

// is `p` page-locked host memory (hipHostMalloc / hipHostRegister / torch pin_memory)?  Then a copy is one DMA, no staging.
static bool host_pinned(const void* p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();   // ordinary malloc'ed memory is "invalid value" to the runtime
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

#define VC_PIPE_CHUNK ((size_t)1 << 20)
// Large results to PAGEABLE host memory: the runtime's own staged copy is serial (DMA a chunk, memcpy it, next chunk); here the
// DMA of chunk i+1 runs while the host copies chunk i out of the other pinned buffer.  Synchronises the stream.
static int d2h_pipelined(vc_engine* e, void* dst, const void* d_src, size_t bytes) {
  if (!e->h_pipe) {
    if (hipHostMalloc((void**)&e->h_pipe, 2 * VC_PIPE_CHUNK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); e->h_pipe = nullptr; }
    for (hipEvent_t& ev : e->pipe_ev)
      if (e->h_pipe && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
  }
  if (!e->h_pipe || !e->pipe_ev[0] || !e->pipe_ev[1]) {   // no pinned memory to be had: the runtime's staged copy
    VC_HIP(e, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, e->stream));
    VC_HIP(e, hipStreamSynchronize(e->stream));
    return VC_OK;
  }
  const size_t n_chunks = (bytes + VC_PIPE_CHUNK - 1) / VC_PIPE_CHUNK;
  for (size_t c = 0; c <= n_chunks; ++c) {
    if (c < n_chunks) {
      const size_t off = c * VC_PIPE_CHUNK, len = std::min(VC_PIPE_CHUNK, bytes - off);
      VC_HIP(e, hipMemcpyAsync(e->h_pipe + (c & 1) * VC_PIPE_CHUNK, (const uint8_t*)d_src + off, len, hipMemcpyDeviceToHost, e->stream));
      VC_HIP(e, hipEventRecord(e->pipe_ev[c & 1], e->stream));
    }
    if (c > 0) {
      const size_t off = (c - 1) * VC_PIPE_CHUNK, len = std::min(VC_PIPE_CHUNK, bytes - off);
      VC_HIP(e, hipEventSynchronize(e->pipe_ev[(c - 1) & 1]));
      memcpy((uint8_t*)dst + off, e->h_pipe + ((c - 1) & 1) * VC_PIPE_CHUNK, len);
    }
  }
  return VC_OK;
}

int vc_search_knn(vc_engine* e, const void* queries, uint32_t nq, uint32_t k, uint32_t mode, uint32_t order,
                  uint64_t* out, uint32_t* counts, vc_query_stats* stats) {
  int rc = check_knn_args(e, queries, nq, k, mode);
  if (rc) return rc;
  if (!out || order > VC_ORDER_FARTHEST_FIRST) return VC_ERR_INVALID;
  if ((rc = bind_device(e))) return rc;
  const size_t qbytes = (size_t)nq * (e->bits / 8);
  if ((rc = grow(e, &e->d_q, &e->q_bytes, qbytes))) return rc;
  if ((rc = grow(e, &e->d_out, &e->out_bytes, (size_t)nq * k * 8))) return rc;
  if ((rc = grow(e, &e->d_cnt, &e->cnt_bytes, (size_t)nq * 8))) return rc;
  // Pageable host memory makes every async copy a staged, partly synchronous one (~20 us each at this size: a third of what the
  // host-pointer call costs over the device-pointer call).  Batches of up to 512 KB go through a pinned staging buffer of the
  // engine: one memcpy in, one out, true async copies in between.
  // Larger batches (an MIH call of 16 384 queries returns 13 MB of rows): straight DMA when the caller's buffers are page-locked,
  // else the rows travel through two pinned chunks, DMA and host copy overlapped (d2h_pipelined).
  const size_t rows_bytes = (size_t)nq * k * 8, pin_need = ((qbytes + 63) & ~(size_t)63) + rows_bytes + (size_t)nq * 4;
  uint8_t *pin_q = nullptr, *pin_rows = nullptr, *pin_cnt = nullptr;
  const bool small = pin_need <= ((size_t)512 << 10);     // (one staging buffer, one host copy: 8 queries x top-100 and the like)
  const bool out_direct = !small && host_pinned(out);
  if (small) {
    if (pin_need > e->pin_bytes) {
      if (e->h_pin) (void)hipHostFree(e->h_pin);
      e->h_pin = nullptr;
      e->pin_bytes = 0;
      const size_t want = std::max<size_t>(pin_need, 64 << 10);
      if (hipHostMalloc((void**)&e->h_pin, want, hipHostMallocDefault) == hipSuccess) e->pin_bytes = want;
      else { (void)hipGetLastError(); e->h_pin = nullptr; }
    }
    if (e->h_pin) {
      pin_q = e->h_pin;
      pin_rows = e->h_pin + ((qbytes + 63) & ~(size_t)63);
      pin_cnt = pin_rows + rows_bytes;
      memcpy(pin_q, queries, qbytes);
    }
  }
  VC_HIP(e, hipMemcpyAsync(e->d_q, pin_q ? (const void*)pin_q : queries, qbytes, hipMemcpyHostToDevice, e->stream));
  std::vector<uint32_t> cnt(2 * (size_t)nq, 0);
  std::vector<vc_query_stats> st;
  timing_begin(e);
  if (mode == VC_MODE_LINEAR) {
    rc = linear_batch(e, e->d_q, nq, k, e->d_out, e->d_cnt);
  } else {
    if (stats) st.resize(nq);      // (the statistics cost four read-backs and a wait per launch: only when asked for)
    const VcMihScanFallback fb{mih_scan_fallback, e, e->n_cu};
    rc = vc_mih_search(e->mih, e->d_cols, e->stride, e->n, e->d_q, nq, k, mode == VC_MODE_MIH_APPROX, e->d_out,
                       e->d_cnt, stats ? st.data() : nullptr, e->stream, &e->err, &fb);
  }
  timing_end(e);
  if (rc) return rc;
  VC_HIP(e, hipMemcpyAsync(pin_cnt ? (void*)pin_cnt : (void*)cnt.data(), e->d_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, e->stream));
  if (small || out_direct) {
    VC_HIP(e, hipMemcpyAsync(pin_rows ? (void*)pin_rows : (void*)out, e->d_out, rows_bytes, hipMemcpyDeviceToHost, e->stream));
    VC_HIP(e, hipStreamSynchronize(e->stream));
  } else if ((rc = d2h_pipelined(e, out, e->d_out, rows_bytes))) {
    return rc;
  }
  if (pin_rows) {
    memcpy(out, pin_rows, rows_bytes);
    memcpy(cnt.data(), pin_cnt, (size_t)nq * 4);
  }
  if (mode == VC_MODE_LINEAR) {
    // a row still flagged here overflowed its ring and was not recomputed on the device (VC_DEVICE_RECOVER=0, or the
    // recovery grid gave up): host-driven fallback
    std::vector<uint32_t> over;
    for (uint32_t i = 0; i < nq; ++i)
      if (cnt[i] == 0xFFFFFFFFu) over.push_back(i);
    if (!over.empty()) {
      // the device recovery gave up (or is switched off): its last block restores the barrier words itself, but should
      // that block never have run (the grid was cut short) they would stay dirty for the life of the engine -- the
      // stream is idle here, so the three lines are simply rewritten
      if (e->d_rec) VC_HIP(e, hipMemsetAsync(e->d_rec + vc_recover_barrier_offset_words(), 0, 96 * 4, e->stream));
      if ((rc = linear_recover(e, e->d_q, k, over, out, cnt.data()))) return rc;
    }
  }
  for (uint32_t i = 0; i < nq; ++i) {
    if (order == VC_ORDER_FARTHEST_FIRST) std::reverse(out + (size_t)i * k, out + (size_t)i * k + cnt[i]);
    if (counts) counts[i] = cnt[i];
    if (stats) {
      if (mode == VC_MODE_LINEAR) {
        memset(&stats[i], 0, sizeof stats[i]);
        stats[i].n_candidates = e->n;
      } else {
        stats[i] = st[i];
      }
      stats[i].n_results = cnt[i];
    }
  }
  return VC_OK;
}

int vc_device_status(vc_engine* e, uint32_t* n_gave_up) {
  if (!e || !n_gave_up) return VC_ERR_INVALID;
  *n_gave_up = 0;
  if (!e->d_rec) return VC_OK;
  int rc = bind_device(e);
  if (rc) return rc;
  uint32_t* d_flag = e->d_rec + vc_recover_scratch_words() - 32;
  VC_HIP(e, hipMemcpyAsync(n_gave_up, d_flag, 4, hipMemcpyDeviceToHost, e->stream));
  VC_HIP(e, hipMemsetAsync(d_flag, 0, 4, e->stream));
  VC_HIP(e, hipStreamSynchronize(e->stream));
  return VC_OK;
}

int vc_merge_topk_dev(const uint64_t* d_lists, uint32_t n_lists, uint32_t nq, uint32_t k, uint64_t* d_out,
                      uint32_t* d_counts, void* stream) {
  if (!d_lists || !d_out || n_lists == 0 || nq == 0 || k == 0 || k > VC_MAX_K) return VC_ERR_INVALID;
  hipError_t r = vc_launch_select_lists(d_lists, n_lists, nq, k, d_out, d_counts, (hipStream_t)stream);
  if (r != hipSuccess) return fail(nullptr, VC_ERR_HIP, "merge launch: %s", hipGetErrorString(r));
  return VC_OK;
}

// ---- MIH index / bucket views -----------------------------------------------------------------------
int vc_build_index(vc_engine* e) {
  if (!e) return VC_ERR_INVALID;
  if (e->m == 0) return fail(e, VC_ERR_STATE, "engine was created with n_tables = 0 (linear only)");
  int rc = bind_device(e);
  if (rc) return rc;
  if (e->mih) { vc_mih_free(e->mih); e->mih = nullptr; }
  return vc_mih_build(&e->mih, e->d_cols, e->stride, e->n, e->W, e->m, e->sbits, e->cfg.id_base, e->cfg.flags, e->n_cu,
                      e->cap, e->knobs, e->stream, &e->err);
}

int vc_get_bucket(vc_engine* e, uint32_t table, uint32_t index, uint32_t* ids, void* codes, uint32_t cap, uint32_t* n) {
  if (!e || !n) return VC_ERR_INVALID;
  if (!e->mih) return fail(e, VC_ERR_STATE, "no index built");
  if (table >= e->m) return fail(e, VC_ERR_INVALID, "table %u >= n_tables %u", table, e->m);
  int rc = bind_device(e);
  if (rc) return rc;
  std::vector<uint32_t> local;
  rc = vc_mih_bucket(e->mih, table, index, &local, e->stream, &e->err);
  if (rc < 0) return rc;
  *n = (uint32_t)local.size();
  if (local.empty()) return VC_NOT_FOUND;
  const uint32_t take = std::min<uint32_t>(cap, (uint32_t)local.size());
  if (codes && take) {
    const size_t rec = e->bits / 8;
    if ((rc = grow(e, (uint8_t**)&e->d_stage, &e->stage_bytes, take * (rec + 4)))) return rc;
    uint32_t* d_ids = (uint32_t*)((uint8_t*)e->d_stage + (size_t)take * rec);
    VC_HIP(e, hipMemcpyAsync(d_ids, local.data(), take * 4, hipMemcpyHostToDevice, e->stream));
    VC_HIP(e, vc_launch_gather_rows(e->d_cols, e->stride, e->W, d_ids, take, (uint64_t*)e->d_stage, e->stream));
    VC_HIP(e, hipMemcpyAsync(codes, e->d_stage, take * rec, hipMemcpyDeviceToHost, e->stream));
    VC_HIP(e, hipStreamSynchronize(e->stream));
  }
  if (ids)
    for (uint32_t i = 0; i < take; ++i) ids[i] = e->cfg.id_base + local[i];
  return VC_OK;
}

int vc_bitmap_test(vc_engine* e, uint32_t table, uint32_t index, int* bit) {
  if (!e || !bit) return VC_ERR_INVALID;
  if (!e->mih) return fail(e, VC_ERR_STATE, "no index built");
  if (table >= e->m) return VC_ERR_INVALID;
  int rc = bind_device(e);
  if (rc) return rc;
  return vc_mih_bitmap_test(e->mih, table, index, bit, e->stream, &e->err);
}

int vc_bitmap_read(vc_engine* e, uint32_t table, uint64_t word_off, uint64_t n_words, uint32_t* out) {
  if (!e || !out) return VC_ERR_INVALID;
  if (!e->mih) return fail(e, VC_ERR_STATE, "no index built");
  if (table >= e->m) return VC_ERR_INVALID;
  int rc = bind_device(e);
  if (rc) return rc;
  return vc_mih_bitmap_read(e->mih, table, word_off, n_words, out, e->stream, &e->err);
}

int vc_search_radius(vc_engine* e, const void* queries, uint32_t nq, uint32_t radius, uint32_t mode, uint64_t* out,
                     uint64_t out_cap, uint64_t* out_offsets) {
  if (!e || !queries || !out_offsets || nq == 0) return VC_ERR_INVALID;
  if (mode != VC_MODE_LINEAR && mode != VC_MODE_MIH_EXACT) return fail(e, VC_ERR_INVALID, "radius search: mode must be LINEAR or MIH_EXACT");
  if (mode == VC_MODE_MIH_EXACT && !e->mih) return fail(e, VC_ERR_STATE, "MIH search needs vc_build_index() first");
  int rc = bind_device(e);
  if (rc) return rc;
  const size_t qbytes = (size_t)nq * (e->bits / 8);
  if ((rc = grow(e, &e->d_q, &e->q_bytes, qbytes))) return rc;
  VC_HIP(e, hipMemcpyAsync(e->d_q, queries, qbytes, hipMemcpyHostToDevice, e->stream));
  timing_begin(e);
  rc = vc_radius_search(e->mih, mode == VC_MODE_MIH_EXACT, e->d_cols, e->stride, e->n, e->W, e->cfg.id_base, e->n_cu, &e->knobs,
                        e->d_q, nq, radius, out, out_cap, out_offsets, false, &e->radius_work, e->stream, &e->err);
  timing_end(e);
  return rc;
}

int vc_search_radius_dev(vc_engine* e, const void* d_queries, uint32_t nq, uint32_t radius, uint32_t mode, uint64_t* d_out,
                         uint64_t out_cap, uint64_t* d_offsets, void* stream) {
  if (!e || !d_queries || !d_offsets || (!d_out && out_cap) || nq == 0) return VC_ERR_INVALID;
  if (mode != VC_MODE_LINEAR && mode != VC_MODE_MIH_EXACT) return fail(e, VC_ERR_INVALID, "radius search: mode must be LINEAR or MIH_EXACT");
  if (mode == VC_MODE_MIH_EXACT && !e->mih) return fail(e, VC_ERR_STATE, "MIH search needs vc_build_index() first");
  int rc = bind_device(e);
  if (rc) return rc;
  hipStream_t saved = e->stream;
  e->stream = stream == VC_STREAM_OWN ? e->own_stream : (hipStream_t)stream;
  timing_begin(e);
  rc = vc_radius_search(e->mih, mode == VC_MODE_MIH_EXACT, e->d_cols, e->stride, e->n, e->W, e->cfg.id_base, e->n_cu, &e->knobs,
                        (const uint64_t*)d_queries, nq, radius, d_out, out_cap, d_offsets, true, &e->radius_work, e->stream, &e->err);
  timing_end(e);
  e->stream = saved;
  return rc;
}

}  // extern "C"
