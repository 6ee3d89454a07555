// Dev tool: the verify kernel's inner loop (vc_scan.hip one_query) on a register-resident code tile, no global loads:
// how many SIMD cycles does one (query x 8-item lane tile) cost, against the 32 v_xor (2 cyc) + 32 v_bcnt (4 cyc) = 192
// the issue rates of tools/ubench_valu.hip predict?  Variants isolate the candidates for the gap.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_verify.hip -o tools/ubench_verify && tools/ubench_verify
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
__device__ __forceinline__ uint32_t bcnt0(uint32_t x) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x));
  return r;
}

// VARIANT 0: as in vc_scan_kernel (branch per query, LDS query prefetch through two register sets)
// VARIANT 1: no branch per query: hits OR-ed, one branch per tile
// VARIANT 2: bcnt only (no xor): floor of the 32 v_bcnt
// VARIANT 3: xor only (xor + or-reduce): floor of the 32 v_xor
// VARIANT 4: as 0 but the query words come from SGPRs (s_load through readfirstlane)
// VARIANT 5: as 1, queries in pairs: two queries share the loop overhead and interleave their chains
template <int VARIANT, int WPS>
__global__ void __launch_bounds__(256, WPS) k(const uint64_t* __restrict__ qg, uint32_t qt, uint32_t iters, uint32_t* out,
                                               unsigned long long* cyc) {
  constexpr int U = 4, W = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* sq = (uint64_t*)smem;
  uint32_t* st = (uint32_t*)(smem + (size_t)qt * W * 8);
  for (uint32_t i = threadIdx.x; i < qt * W; i += 256) sq[i] = qg[i];
  for (uint32_t i = threadIdx.x; i < qt; i += 256) st[i] = 3;   // nothing is ever this near
  __syncthreads();
  u64x2 r[U][W];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < W; ++j) {
      r[u][j].x = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 64 * u + 7 * j) ^ (blockIdx.x * 0xD6E8FEB86659FD93ull);
      r[u][j].y = 0xBF58476D1CE4E5B9ull * (threadIdx.x + 3 + 64 * u + 5 * j) ^ (blockIdx.x * 0x94D049BB133111EBull);
    }
  uint32_t hits = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < W; ++j) asm volatile("" : "+v"(r[u][j]));   // the tile is "new" every iteration
    auto one = [&](const uint64_t(&qw)[W], uint32_t t) -> bool {
      uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t da = 0, db = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if (VARIANT == 2) {
            da = j ? bcnt_acc((uint32_t)r[u][j].x, da) : bcnt0((uint32_t)r[u][j].x);
            db = j ? bcnt_acc((uint32_t)r[u][j].y, db) : bcnt0((uint32_t)r[u][j].y);
            da = bcnt_acc((uint32_t)(r[u][j].x >> 32), da);
            db = bcnt_acc((uint32_t)(r[u][j].y >> 32), db);
          } else if (VARIANT == 3) {
            const uint64_t xa = r[u][j].x ^ qw[j], xb = r[u][j].y ^ qw[j];
            da |= (uint32_t)xa;
            db |= (uint32_t)xb;
            da |= (uint32_t)(xa >> 32);
            db |= (uint32_t)(xb >> 32);
          } else {
            const uint64_t xa = r[u][j].x ^ qw[j], xb = r[u][j].y ^ qw[j];
            da = j ? bcnt_acc((uint32_t)xa, da) : bcnt0((uint32_t)xa);
            db = j ? bcnt_acc((uint32_t)xb, db) : bcnt0((uint32_t)xb);
            da = bcnt_acc((uint32_t)(xa >> 32), da);
            db = bcnt_acc((uint32_t)(xb >> 32), db);
          }
        }
        dmin = min(dmin, min(da, db));
      }
      return dmin <= t;
    };
    if (VARIANT == 4) {
      for (uint32_t q = 0; q < qt; ++q) {
        uint64_t qw[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const uint64_t v = qg[q * W + j];   // uniform address: scalar load
          qw[j] = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);
        }
        if (__ballot(one(qw, 3)) != 0) hits += q;
      }
    } else if (VARIANT == 1 || VARIANT == 5) {
      bool any = false;
      for (uint32_t q = 0; q < qt; ++q) {
        uint64_t qw[W];
#pragma unroll
        for (int j = 0; j < W; ++j) qw[j] = sq[q * W + j];
        any |= one(qw, st[q]);
      }
      if (__ballot(any) != 0) hits += 1;
    } else {
      uint64_t qa[W], qb[W];
      uint32_t ta, tb;
      const uint32_t last = qt - 1;
      auto ldq = [&](uint64_t(&qw)[W], uint32_t& t, uint32_t q) {
#pragma unroll
        for (int j = 0; j < W; ++j) qw[j] = sq[q * W + j];
        t = st[q];
      };
      ldq(qa, ta, 0);
      for (uint32_t q = 0; q < qt; q += 2) {
        ldq(qb, tb, min(q + 1, last));
        if (__ballot(one(qa, ta)) != 0) hits += q;
        ldq(qa, ta, min(q + 2, last));
        if (q + 1 < qt && __ballot(one(qb, tb)) != 0) hits += q + 1;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = hits;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__device__ __forceinline__ uint32_t bcnt_acc_s(uint32_t x, uint32_t acc) {   // accumulator operand in an SGPR
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(acc));
  return r;
}

// fully unrolled over QT = 8 queries, "matching bits" form: operands are the COMPLEMENTED query words, the accumulate
// chain starts at tau, so acc = tau + 128 - d and d <= tau <=> acc >= 128; the 8 chains are OR-reduced (v_or3).
//   MODE 0: query words + tau in SGPRs (loaded once)   MODE 1: query words + tau in VGPRs (loaded once)
//   MODE 2: query words from LDS with immediate offsets (ds_read_b128 per query), tau in SGPRs
template <int MODE, int WPS>
__global__ void __launch_bounds__(256, WPS) k2(const uint64_t* __restrict__ qg, uint32_t iters, uint32_t* out, unsigned long long* cyc) {
  constexpr int U = 4, W = 2, QT = 8;
  __shared__ __attribute__((aligned(16))) uint64_t sq[QT * W];
  for (uint32_t i = threadIdx.x; i < QT * W; i += 256) sq[i] = ~qg[i];
  __syncthreads();
  u64x2 r[U][W];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < W; ++j) {
      r[u][j].x = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 64 * u + 7 * j) ^ (blockIdx.x * 0xD6E8FEB86659FD93ull);
      r[u][j].y = 0xBF58476D1CE4E5B9ull * (threadIdx.x + 3 + 64 * u + 5 * j) ^ (blockIdx.x * 0x94D049BB133111EBull);
    }
  uint32_t nq[QT][2 * W], tau[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const uint64_t v = ~qg[q * W + j];
      nq[q][2 * j] = (uint32_t)v;
      nq[q][2 * j + 1] = (uint32_t)(v >> 32);
      if (MODE == 0) {
        nq[q][2 * j] = __builtin_amdgcn_readfirstlane(nq[q][2 * j]);
        nq[q][2 * j + 1] = __builtin_amdgcn_readfirstlane(nq[q][2 * j + 1]);
      }
    }
    tau[q] = 3;
    if (MODE != 1 && MODE < 3) tau[q] = __builtin_amdgcn_readfirstlane(tau[q] + (qg[0] == 1));
    if (MODE == 1 || MODE >= 3) asm volatile("" : "+v"(tau[q]));
  }
  if (MODE == 1) {
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int j = 0; j < 2 * W; ++j) asm volatile("" : "+v"(nq[q][j]));
  }
  uint32_t hits = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < W; ++j) asm volatile("" : "+v"(r[u][j]));
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      uint32_t w[2 * W];
      if (MODE >= 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(&sq[q * W]);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 2 * W; ++j) w[j] = nq[q][j];
      }
      uint32_t orv = 0;
      if (MODE >= 3) {   // runs: all xors of GRP item pairs first, then their popcounts (sched barriers keep hipcc from re-mixing)
        constexpr int GRP = MODE == 3 ? 1 : (MODE == 4 ? 2 : 4);
#pragma unroll
        for (int u0 = 0; u0 < U; u0 += GRP) {
          uint32_t x[GRP][W][4];
#pragma unroll
          for (int g = 0; g < GRP; ++g)
#pragma unroll
            for (int j = 0; j < W; ++j) {
              x[g][j][0] = (uint32_t)r[u0 + g][j].x ^ w[2 * j];
              x[g][j][1] = (uint32_t)(r[u0 + g][j].x >> 32) ^ w[2 * j + 1];
              x[g][j][2] = (uint32_t)r[u0 + g][j].y ^ w[2 * j];
              x[g][j][3] = (uint32_t)(r[u0 + g][j].y >> 32) ^ w[2 * j + 1];
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < GRP; ++g) {
            uint32_t da = tau[q], db = tau[q];
#pragma unroll
            for (int j = 0; j < W; ++j) {
              da = bcnt_acc(x[g][j][0], da);
              db = bcnt_acc(x[g][j][2], db);
              da = bcnt_acc(x[g][j][1], da);
              db = bcnt_acc(x[g][j][3], db);
            }
            orv |= da | db;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (__ballot(orv >= 128u) != 0) hits += q + 1;
        continue;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t da, db;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          const uint32_t xal = (uint32_t)r[u][j].x ^ w[2 * j], xah = (uint32_t)(r[u][j].x >> 32) ^ w[2 * j + 1];
          const uint32_t xbl = (uint32_t)r[u][j].y ^ w[2 * j], xbh = (uint32_t)(r[u][j].y >> 32) ^ w[2 * j + 1];
          if (j == 0) {
            da = MODE == 1 ? bcnt_acc(xal, tau[q]) : bcnt_acc_s(xal, tau[q]);
            db = MODE == 1 ? bcnt_acc(xbl, tau[q]) : bcnt_acc_s(xbl, tau[q]);
          } else {
            da = bcnt_acc(xal, da);
            db = bcnt_acc(xbl, db);
          }
          da = bcnt_acc(xah, da);
          db = bcnt_acc(xbh, db);
        }
        orv |= da | db;
      }
      if (__ballot(orv >= 128u) != 0) hits += q + 1;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = hits;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int WPS>
void run2(const char* name, const uint64_t* dq, uint32_t* dout, unsigned long long* dcyc) {
  const uint32_t iters = 400, blocks = 256 * WPS, qt = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, dq, 10u, dout, dcyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k2<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, dq, iters, dout, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double units = (double)iters * qt;
  const double ns = ms * 1e6 / units / WPS;
  printf("%-34s qt=%2u waves/SIMD=%d  %.3f ms  %.1f ns per (query x 8-item tile) per SIMD = %.0f cyc @2.4GHz\n", name, qt, WPS, ms, ns, ns * 2.4);
}

template <int VARIANT, int WPS>
void run(const char* name, const uint64_t* dq, uint32_t qt, uint32_t* dout, unsigned long long* dcyc) {
  const uint32_t iters = 400, blocks = 256 * WPS;
  const size_t lds = (size_t)qt * (2 * 8 + 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<VARIANT, WPS>), dim3(blocks), dim3(256), lds, 0, dq, qt, 10u, dout, dcyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<VARIANT, WPS>), dim3(blocks), dim3(256), lds, 0, dq, qt, iters, dout, dcyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  static unsigned long long h[4096];
  hipMemcpy(h, dcyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (uint32_t i = 0; i < blocks; ++i) avg += (double)h[i];
  avg /= blocks;
  // s_memtime ticks at 100 MHz on gfx950? report both: wall-derived cycles at the nominal 2.4 GHz and ticks
  const double units = (double)iters * qt;                         // (query x tile) units per wave
  const double ns_per_unit_per_simd = ms * 1e6 / units / 1.0 / WPS * 1.0;   // WPS waves share one SIMD
  printf("%-34s qt=%2u waves/SIMD=%d  %.3f ms  %.1f ns per (query x 8-item tile) per SIMD = %.0f cyc @2.4GHz  [memtime %.0f ticks/unit/wave]\n",
         name, qt, WPS, ms, ns_per_unit_per_simd, ns_per_unit_per_simd * 2.4, avg / units);
}

int main() {
  uint64_t hq[64 * 2];
  for (int i = 0; i < 128; ++i) hq[i] = 0x123456789ABCDEFull * (i + 1);
  uint64_t* dq;
  uint32_t* dout;
  unsigned long long* dcyc;
  hipMalloc(&dq, sizeof hq);
  hipMemcpy(dq, hq, sizeof hq, hipMemcpyHostToDevice);
  hipMalloc(&dout, 256 * 8 * 256 * 4);
  hipMalloc(&dcyc, 4096 * 8);
  run2<0, 4>("U0 unrolled, SGPR queries", dq, dout, dcyc);
  run2<1, 4>("U1 unrolled, VGPR queries", dq, dout, dcyc);
  run2<2, 4>("U2 unrolled, LDS imm offsets", dq, dout, dcyc);
  run2<3, 4>("U3 LDS imm, runs of 8 xor / 8 bcnt", dq, dout, dcyc);
  run2<4, 4>("U4 LDS imm, runs of 16 / 16", dq, dout, dcyc);
  run2<5, 4>("U5 LDS imm, runs of 32 / 32", dq, dout, dcyc);
  run2<0, 3>("U0 unrolled, SGPR, 3 waves/SIMD", dq, dout, dcyc);
  run2<1, 3>("U1 unrolled, VGPR, 3 waves/SIMD", dq, dout, dcyc);
  run2<0, 5>("U0 unrolled, SGPR, 5 waves/SIMD", dq, dout, dcyc);
  run2<2, 5>("U2 unrolled, LDS, 5 waves/SIMD", dq, dout, dcyc);
  for (uint32_t qt : {8u}) {
    run<0, 4>("0 as vc_scan (branch per query)", dq, qt, dout, dcyc);
    run<1, 4>("1 one branch per tile", dq, qt, dout, dcyc);
    run<2, 4>("2 bcnt only", dq, qt, dout, dcyc);
    run<3, 4>("3 xor only", dq, qt, dout, dcyc);
    run<4, 4>("4 SGPR query words", dq, qt, dout, dcyc);
    run<0, 3>("0 as vc_scan, 3 waves/SIMD", dq, qt, dout, dcyc);
    run<0, 2>("0 as vc_scan, 2 waves/SIMD", dq, qt, dout, dcyc);
    run<1, 2>("1 one branch per tile, 2 waves/SIMD", dq, qt, dout, dcyc);
  }
  return 0;
}
