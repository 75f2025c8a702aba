// Issue cost (cycles per instruction, ONE wave on a SIMD) of the instructions the block-mode LSTM scan is made of
// (csrc/offline.hip lstm_scan_kernel: 21 v_readlane + 21 v_pk_fma_f32 + 3 v_exp / v_rcp pairs + DPP moves per step):
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
// Every case is 64 copies of one instruction (or pair) between two s_memtime reads, repeated `iters` times; "dep" =
// every instruction reads the result of the one before it, "ind" = four independent chains.
// (The dependent v_pk_fma_f32 chain is timed WITHOUT the wait state the compiler puts between two dependent packed ops
// (s_nop 0): with it a single chain costs 5.5 per instruction, which is why the scan keeps two accumulator chains.)
// Results: profiles/r03_ubench_valu_issue.txt -- 4.4-4.5 cycles for v_fmac / v_pk_fma / v_readlane, 8.6 for v_exp / v_rcp
// (12.4 when the next instruction needs the result), 12.4 for a DPP op that reads the result of the op before it.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"

#define REP64(body) ".rept 16\n" body ".endr\n"

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 0.001f, a1 = 0.5f, a2 = 0.25f, a3 = 0.125f;
  float b = 1.0f + threadIdx.x * 1e-6f, c = 0.999f;
  double p0 = 0.0, p1 = 0.0;       // 64-bit register pairs for the packed ops
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) asm volatile(REP64("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %0, %4, %5\n v_fmac_f32 %0, %4, %5\n v_fmac_f32 %0, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (MODE == 1) asm volatile(REP64("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (MODE == 2) asm volatile("s_mov_b32 s8, 1.0\n" REP64("v_pk_fma_f32 %0, %2, s[8:9], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, %2, s[8:9], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, %2, s[8:9], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, %2, s[8:9], %0 op_sel_hi:[1,0,1]\n") : "+v"(p0), "+v"(p1) : "v"(p1) : "s8", "s9");
    if (MODE == 3) asm volatile("s_mov_b32 s8, 1.0\n" REP64("v_pk_fma_f32 %0, %2, s[8:9], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %2, s[8:9], %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %0, %2, s[8:9], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %2, s[8:9], %1 op_sel_hi:[1,0,1]\n") : "+v"(p0), "+v"(p1) : "v"(p1) : "s8", "s9");
    if (MODE == 4) asm volatile(REP64("v_readlane_b32 s8, %0, 0\n v_readlane_b32 s9, %0, 2\n v_readlane_b32 s10, %0, 4\n v_readlane_b32 s11, %0, 6\n") : "+v"(a0) : : "s8", "s9", "s10", "s11");
    if (MODE == 5) asm volatile(REP64("v_readlane_b32 s8, %0, 0\n s_nop 1\n v_fmac_f32 %0, s8, %1\n v_readlane_b32 s8, %0, 2\n s_nop 1\n v_fmac_f32 %0, s8, %1\n") : "+v"(a0) : "v"(c) : "s8");
    if (MODE == 6) asm volatile(REP64("v_exp_f32 %0, %0\n s_nop 0\n v_exp_f32 %0, %0\n s_nop 0\n v_exp_f32 %0, %0\n s_nop 0\n v_exp_f32 %0, %0\n s_nop 0\n") : "+v"(a0));
    if (MODE == 7) asm volatile(REP64("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    if (MODE == 8) asm volatile(REP64("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    if (MODE == 9) asm volatile(REP64("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n") : "+v"(a0));
    if (MODE == 10) asm volatile(REP64("v_exp_f32 %0, %0\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    if (MODE == 11) asm volatile(REP64("v_pk_mul_f32 %0, %2, %0\n v_pk_add_f32 %1, %2, %1\n v_pk_mul_f32 %0, %2, %0\n v_pk_add_f32 %1, %2, %1\n") : "+v"(p0), "+v"(p1) : "v"(p1));
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = a0 + a1 + a2 + a3 + static_cast<float>(p0 + p1);
}

template <int MODE>
static void run(const char* what, int per_block) {
  float* out; long long* cyc;
  hipMalloc(&out, 64 * sizeof(float)); hipMalloc(&cyc, sizeof(long long));
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 10);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h = 0; hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double n = static_cast<double>(iters) * per_block;
  printf("%-58s %7.2f clock64 ticks / instr   %7.2f ns / instr\n", what, h / n, 1e6 * ms / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_fmac_f32, dependent chain", 64);
  run<1>("v_fmac_f32, four chains", 64);
  run<2>("v_pk_fma_f32 (SGPR broadcast operand), dependent chain", 64);
  run<3>("v_pk_fma_f32 (SGPR broadcast operand), two chains", 64);
  run<4>("v_readlane_b32 back to back", 64);
  run<5>("v_readlane_b32 + s_nop 1 + v_fmac_f32 (per pair)", 32);
  run<6>("v_exp_f32 + s_nop 0, dependent chain", 64);
  run<7>("v_exp_f32, four chains", 64);
  run<8>("v_rcp_f32, four chains", 64);
  run<9>("v_mov_b32_dpp quad_perm + s_nop 1, dependent chain", 64);
  run<10>("1 v_exp_f32 + 3 v_fmac_f32 (per instr)", 64);
  run<11>("v_pk_mul_f32 / v_pk_add_f32, two chains", 64);
  return 0;
}
