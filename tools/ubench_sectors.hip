// Dev tool: the random-sector ceiling of the memory system -- what the MIH kernels' access pattern can flow at.
// Every lane loads `bytes` (4 / 16 / 64) from a uniformly random 64-byte-aligned sector of a buffer of `footprint` bytes,
// U independent loads in flight per lane, no dependence between loads (the kernels' chains are shorter than this:
// it is a ceiling, not a model).  Prints sectors/s and the byte rate of 64-byte requests for several footprints:
// 256 MB (the Infinity Cache), 2 GB (four 512 MB occupancy bitmaps), 16 / 64 / 128 GB (bucket-order records at 1e9).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_sectors.hip -o tools/ubench_sectors && tools/ubench_sectors
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {   // splitmix64 step
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <int BYTES, int U>
__global__ void __launch_bounds__(256) k_sectors(const uint32_t* __restrict__ buf, uint64_t n_sectors, uint32_t iters, uint32_t seed,
                                                 uint32_t* __restrict__ out) {
  uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x10001ull + seed);
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
    uint64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s = mix(s);
      idx[u] = __umul64hi(s, n_sectors) * 16;   // uniform in [0, n_sectors); 16 dwords per 64-byte sector
    }
    if (BYTES == 4) {
      uint32_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = buf[idx[u] + (s >> 60)];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    } else if (BYTES == 32) {
      // a 32-byte record per lane PAIR and instruction: lanes 2i and 2i+1 read the two 16-byte halves of the same record
      // (one sector per pair), halves exchanged across the pair afterwards -- against BYTES == 33: two instructions per lane
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t pi = __shfl(idx[u], (int)(threadIdx.x & 63u & ~1u));   // the pair's record
        v[u] = *reinterpret_cast<const uint4*>(buf + pi + 8 * ((s >> 62) & 1) + 4 * (threadIdx.x & 1u));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w ^ (uint32_t)__shfl_xor((int)v[u].x, 1);
    } else if (BYTES == 33) {
      uint4 v[U][2];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 2; ++c) v[u][c] = *reinterpret_cast<const uint4*>(buf + idx[u] + 8 * ((s >> 62) & 1) + 4 * c);
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0].x ^ v[u][0].w ^ v[u][1].y ^ v[u][1].z;
    } else if (BYTES == 16) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const uint4*>(buf + idx[u] + 4 * ((s >> 62) & 3));
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    } else {
      uint4 v[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][c] = *reinterpret_cast<const uint4*>(buf + idx[u] + 4 * c);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc ^= v[u][c].x ^ v[u][c].y ^ v[u][c].z ^ v[u][c].w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;   // (keeps the loads alive)
}

template <int BYTES, int U>
static void run(const uint32_t* buf, uint64_t footprint, int blocks_per_cu, int n_cu, uint32_t* d_out) {
  const uint64_t n_sectors = footprint / 64;
  const uint32_t iters = 64 / U * 8;
  const dim3 grid(n_cu * blocks_per_cu);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_sectors<BYTES, U>), grid, dim3(256), 0, 0, buf, n_sectors, iters, 1u, d_out);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_sectors<BYTES, U>), grid, dim3(256), 0, 0, buf, n_sectors, iters, 7u + rep, d_out);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  const double sectors = (double)grid.x * 256 * iters * U / (BYTES == 32 ? 2 : 1);
  printf("  %2d-byte loads, %d in flight per lane, %d blocks per CU: %7.3f ms  %6.1f G sectors/s = %5.2f TB/s of 64-byte requests\n", BYTES, U,
         blocks_per_cu, best, sectors / best / 1e6, sectors * 64 / best / 1e9);
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  size_t free_b = 0, total_b = 0;
  CHECK(hipMemGetInfo(&free_b, &total_b));
  const uint64_t sizes_mb[] = {256, 2048, 16384, 65536, 131072};
  uint64_t max_mb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 131072;
  if (max_mb * (1ull << 20) > free_b - (2ull << 30)) max_mb = (free_b - (2ull << 30)) >> 20;
  uint32_t* buf = nullptr;
  uint32_t* d_out = nullptr;
  CHECK(hipMalloc((void**)&buf, max_mb << 20));
  CHECK(hipMalloc((void**)&d_out, 256));
  CHECK(hipMemset(buf, 0, max_mb << 20));   // touch every page
  printf("%s, %d CUs; buffer of %llu MB\n", prop.name, n_cu, (unsigned long long)max_mb);
  for (uint64_t mb : sizes_mb) {
    if (mb > max_mb) break;
    printf("footprint %llu MB\n", (unsigned long long)mb);
    run<16, 4>(buf, mb << 20, 4, n_cu, d_out);    // the k-NN scan's shape: 4 granules per lane, 4 blocks per CU
    run<16, 4>(buf, mb << 20, 8, n_cu, d_out);
    run<16, 8>(buf, mb << 20, 8, n_cu, d_out);
    run<4, 8>(buf, mb << 20, 8, n_cu, d_out);
    run<64, 2>(buf, mb << 20, 8, n_cu, d_out);    // the radius search's 512-bit granules
    run<64, 4>(buf, mb << 20, 8, n_cu, d_out);
    run<33, 4>(buf, mb << 20, 8, n_cu, d_out);    // "33" = a 32-byte record as two 16-byte loads of one lane (the k-NN verify today)
    run<32, 8>(buf, mb << 20, 8, n_cu, d_out);    // "32" = the same record read by a lane pair in ONE instruction
  }
  CHECK(hipFree(buf));
  CHECK(hipFree(d_out));
  return 0;
}
