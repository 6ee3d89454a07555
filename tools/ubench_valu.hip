// Dev tool: per-instruction VALU issue rate on gfx950 (cycles per wave64 instruction per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 17;
  uint32_t b = seed * 77 + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#define ONE(r)                                                                                     \
    if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r) : "v"(b));                        \
    if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r) : "v"(b));                   \
    if (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r) : "v"(b));                        \
    if (OP == 3) asm volatile("v_sad_u8 %0, %1, %1, %0" : "+v"(r) : "v"(b));                     \
    if (OP == 4) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(r) : "v"(b));                \
    if (OP == 5) asm volatile("v_dot8_u32_u4 %0, %1, %1, %0" : "+v"(r) : "v"(b));                \
    if (OP == 6) asm volatile("v_bfi_b32 %0, %1, %1, %0" : "+v"(r) : "v"(b));                    \
    if (OP == 7) asm volatile("v_min3_u32 %0, %1, %1, %0" : "+v"(r) : "v"(b));                   \
    if (OP == 8) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(r) : "v"(b));                \
    if (OP == 9) asm volatile("v_and_or_b32 %0, %1, %1, %0" : "+v"(r) : "v"(b));                 \
    if (OP == 10) asm volatile("v_add3_u32 %0, %1, %1, %0" : "+v"(r) : "v"(b));                  \
    if (OP == 11) asm volatile("v_lshl_add_u32 %0, %1, 1, %0" : "+v"(r) : "v"(b));               \
    if (OP == 12) asm volatile("v_perm_b32 %0, %1, %1, %0" : "+v"(r) : "v"(b));                  \
    if (OP == 13) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(r) : "v"(b));                    \
    if (OP == 14) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r) : "v"(b));                       \
    if (OP == 15) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(r) : "v"(b));
    REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char* name, uint32_t* d, int wps) {
  const int iters = 2000, blocks = 256 * wps;  // one 4-wave block puts one wave on each SIMD of a CU
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 10, 1);
  hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr = (double)blocks * 4 * iters * 64;        // wave-instructions
  double per_simd = winstr / (256.0 * 4);                  // per SIMD
  printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cyc)\n", name, wps, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
  for (int wps : {1, 2, 4, 8}) {
    run<0>("v_xor_b32", d, wps); run<1>("v_bcnt_u32_b32", d, wps); run<2>("v_add_u32", d, wps); run<3>("v_sad_u8", d, wps);
    run<4>("v_dot4_u32_u8", d, wps); run<5>("v_dot8_u32_u4", d, wps); run<6>("v_bfi_b32", d, wps); run<7>("v_min3_u32", d, wps);
    run<8>("v_mad_u32_u24", d, wps); run<9>("v_and_or_b32", d, wps); run<10>("v_add3_u32", d, wps); run<11>("v_lshl_add_u32", d, wps);
    run<12>("v_perm_b32", d, wps); run<13>("v_pk_add_u16", d, wps); run<14>("v_sub_u32", d, wps); run<15>("v_mul_u32_u24", d, wps);
  }
  return 0;
}
