// Dev tool: do same-bank VGPR source operands cost extra issue cycles on gfx950?  (bank = vgpr index mod 4)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23"
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
  asm volatile("v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n"
               "v_mov_b32 v20, %1\n v_mov_b32 v21, %1\n v_mov_b32 v22, %1\n v_mov_b32 v23, %1\n" :: "v"(threadIdx.x + seed), "v"(seed * 3 + threadIdx.x) : CLOB);
  for (int i = 0; i < iters; ++i) {
    // 8 independent chains v10..v17; operand register chosen to be in a different / the same bank as the chain register
    if (OP == 0) asm volatile(REP16("v_xor_b32 v10, v21, v10\n v_xor_b32 v11, v22, v11\n v_xor_b32 v12, v23, v12\n v_xor_b32 v13, v20, v13\n v_xor_b32 v14, v21, v14\n v_xor_b32 v15, v22, v15\n v_xor_b32 v16, v23, v16\n v_xor_b32 v17, v20, v17\n") ::: CLOB);
    if (OP == 1) asm volatile(REP16("v_xor_b32 v10, v22, v10\n v_xor_b32 v11, v23, v11\n v_xor_b32 v12, v20, v12\n v_xor_b32 v13, v21, v13\n v_xor_b32 v14, v22, v14\n v_xor_b32 v15, v23, v15\n v_xor_b32 v16, v20, v16\n v_xor_b32 v17, v21, v17\n") ::: CLOB);
    if (OP == 2) asm volatile(REP16("v_bcnt_u32_b32 v10, v21, v10\n v_bcnt_u32_b32 v11, v22, v11\n v_bcnt_u32_b32 v12, v23, v12\n v_bcnt_u32_b32 v13, v20, v13\n v_bcnt_u32_b32 v14, v21, v14\n v_bcnt_u32_b32 v15, v22, v15\n v_bcnt_u32_b32 v16, v23, v16\n v_bcnt_u32_b32 v17, v20, v17\n") ::: CLOB);
    if (OP == 3) asm volatile(REP16("v_bcnt_u32_b32 v10, v22, v10\n v_bcnt_u32_b32 v11, v23, v11\n v_bcnt_u32_b32 v12, v20, v12\n v_bcnt_u32_b32 v13, v21, v13\n v_bcnt_u32_b32 v14, v22, v14\n v_bcnt_u32_b32 v15, v23, v15\n v_bcnt_u32_b32 v16, v20, v16\n v_bcnt_u32_b32 v17, v21, v17\n") ::: CLOB);
    if (OP == 4) asm volatile(REP16("v_xor_b32 v10, s20, v10\n v_xor_b32 v11, s21, v11\n v_xor_b32 v12, s22, v12\n v_xor_b32 v13, s23, v13\n v_xor_b32 v14, s20, v14\n v_xor_b32 v15, s21, v15\n v_xor_b32 v16, s22, v16\n v_xor_b32 v17, s23, v17\n") ::: CLOB, "s20", "s21", "s22", "s23");
    // the real mix: xor into a temp, bcnt-accumulate the temp (different banks everywhere)
    if (OP == 5) asm volatile(REP16("v_xor_b32 v20, s20, v10\n v_bcnt_u32_b32 v14, v20, v14\n v_xor_b32 v21, s21, v11\n v_bcnt_u32_b32 v15, v21, v15\n v_xor_b32 v22, s22, v12\n v_bcnt_u32_b32 v16, v22, v16\n v_xor_b32 v23, s23, v13\n v_bcnt_u32_b32 v17, v23, v17\n") ::: CLOB, "s20", "s21", "s22", "s23");
    // same mix, bcnt sources in the same bank (v20 & v16 are both bank 0, ...)
    if (OP == 6) asm volatile(REP16("v_xor_b32 v20, s20, v10\n v_bcnt_u32_b32 v16, v20, v16\n v_xor_b32 v21, s21, v11\n v_bcnt_u32_b32 v17, v21, v17\n v_xor_b32 v22, s22, v12\n v_bcnt_u32_b32 v14, v22, v14\n v_xor_b32 v23, s23, v13\n v_bcnt_u32_b32 v15, v23, v15\n") ::: CLOB, "s20", "s21", "s22", "s23");
    if (OP == 7) asm volatile(REP16("v_readfirstlane_b32 s20, v10\n v_readfirstlane_b32 s21, v11\n v_readfirstlane_b32 s22, v12\n v_readfirstlane_b32 s23, v13\n v_readfirstlane_b32 s20, v14\n v_readfirstlane_b32 s21, v15\n v_readfirstlane_b32 s22, v16\n v_readfirstlane_b32 s23, v17\n") ::: CLOB, "s20", "s21", "s22", "s23");
  }
  uint32_t r;
  asm volatile("v_add_u32 %0, v10, v11\n v_add_u32 %0, %0, v12\n v_add_u32 %0, %0, v13\n v_add_u32 %0, %0, v14\n v_add_u32 %0, %0, v15\n v_add_u32 %0, %0, v16\n v_add_u32 %0, %0, v17\n" : "=v"(r) :: CLOB);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> void run(const char* name, uint32_t* d, int wps) {
  const int iters = 1000, blocks = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 10, 1);
  hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, iters, 1); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double per_simd = (double)blocks * 4 * iters * 128 / (256.0 * 4);
  printf("%-34s waves/SIMD=%d  %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, wps, ms, ms * 1e6 / per_simd);
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
  for (int wps : {4, 8}) {
    run<0>("xor vgpr,vgpr different banks", d, wps); run<1>("xor vgpr,vgpr same bank", d, wps);
    run<2>("bcnt vgpr,vgpr different banks", d, wps); run<3>("bcnt vgpr,vgpr same bank", d, wps);
    run<4>("xor sgpr,vgpr", d, wps); run<5>("xor(s)+bcnt mix, clean banks", d, wps); run<6>("xor(s)+bcnt mix, bcnt same bank", d, wps);
    run<7>("v_readfirstlane", d, wps);
  }
  return 0;
}
