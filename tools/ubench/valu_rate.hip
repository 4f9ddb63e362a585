// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD, by opcode,
// with W waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP 64
#define ITERS 2048

template <int OP>
__global__ void k(float* out, float seed) {
  float a[8];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 1e-3f + i;
    p[i] = v2f{a[i], a[i] + 0.5f};
  }
  const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(v2f{c1, c1}), "v"(v2f{c2, c2}));
        if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(v2f{c1, c1}));
        if (OP == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 5) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c2), "v"(c1));
        if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1));
        if (OP == 10) asm volatile("v_cmp_ge_f32 vcc, %0, %1" ::"v"(a[i]), "v"(c1) : "vcc");
        if (OP == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
        if (OP == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(v2f{c2, c2}));
        if (OP == 13) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(c1));
        if (OP == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 15) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        if (OP == 16) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 17) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[i]));
        if (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 19) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 20) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
        if (OP == 21) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 22) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 23) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 24) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(c1) : "s10", "s11");
        if (OP == 25) asm volatile("v_cmp_ge_f32_e64 s[10:11], %0, %1" ::"v"(a[i]), "v"(c1) : "s10", "s11");
        if (OP == 26) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 27) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 28) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 29) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
        if (OP == 30) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i]));
        if (OP == 31) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
        if (OP == 32) asm volatile("v_fmamk_f32 %0, %0, 0x3f8ccccd, %1" : "+v"(a[i]) : "v"(c2));
        if (OP == 33) asm volatile("v_cmp_class_f32 vcc, %0, %1" ::"v"(a[i]), "v"(c1) : "vcc");
        if (OP == 34) asm volatile("v_max_f32 %0, 0x38d1b717, %0" : "+v"(a[i]));
        if (OP == 35) asm volatile("v_lshrrev_b32 %0, 31, %0" : "+v"(a[i]));
        if (OP == 36) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 37) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 38) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 39) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 40) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        if (OP == 41) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(a[i]));
        if (OP == 42) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 43) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 44) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(a[i]));
        if (OP == 45) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
        if (OP == 46) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c1));
        if (OP == 47) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char* name, float* d) {
  for (int wps = 4; wps <= 8; wps *= 2) {  // waves per SIMD
    int threads = 256;                     // 4 waves per block = 1 per SIMD
    int blocks = 256 * wps;                // 256 CUs
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double instr_per_simd = (double)wps * REP * ITERS;
    // cycles at an assumed 2.4 GHz (upper bound on clock): report ns per instr per SIMD
    printf("%-14s waves/SIMD %d : %.3f ms  %.3f ns/instr/SIMD  (= %.2f cyc @2.4GHz, %.2f @2.0GHz)\n", name, wps, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, ms * 1e6 / instr_per_simd * 2.0);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  run<0>("v_fma_f32", d);
  run<1>("v_pk_fma_f32", d);
  run<38>("v_sub_f32", d);
  run<30>("v_mul 2.0 inline", d);
  run<32>("v_fmamk literal", d);
  run<26>("v_fma clamp", d);
  run<4>("v_exp_f32", d);
  run<7>("v_max_f32", d);
  run<34>("v_max literal", d);
  run<21>("v_min_f32", d);
  run<22>("v_max3_f32", d);
  run<8>("v_med3_f32", d);
  run<10>("v_cmp_ge vcc", d);
  run<25>("v_cmp_ge sgpr", d);
  run<33>("v_cmp_class", d);
  run<9>("v_cndmask vcc", d);
  run<24>("v_cndmask sgpr", d);
  run<16>("v_bfi_b32", d);
  run<17>("v_ashrrev_i32", d);
  run<35>("v_lshrrev_b32", d);
  run<18>("v_and_b32", d);
  run<28>("v_or_b32", d);
  run<37>("v_xor_b32", d);
  run<19>("v_sub_u32", d);
  run<27>("v_add_u32", d);
  run<39>("v_mad_u32_u24", d);
  run<20>("v_fract_f32", d);
  run<31>("v_floor_f32", d);
  run<29>("v_cvt_f32_i32", d);
  run<40>("v_ldexp_f32", d);
  run<41>("v_frexp_mant", d);
  run<23>("v_mov_b32", d);
  run<42>("v_bitop3_b32", d);
  run<43>("v_min3_f32", d);
  run<44>("v_med3 0,1", d);
  run<45>("v_not_b32", d);
  run<46>("v_cndmask e64 vcc", d);
  run<47>("v_rsq_f32", d);
  return 0;
}
