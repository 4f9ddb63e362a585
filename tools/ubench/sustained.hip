// Sustained-throughput probe: does the chip hold its clock under a long VALU-only load, and do packed
// FP32 ops (v_pk_fma_f32) deliver more FMA/s than plain v_fma_f32 once DVFS has settled?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP>
__global__ void k(float* out, float seed, int iters) {
  float a[8];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
  const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(v2f{c1, c1}), "v"(v2f{c2, c2}));
        if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (OP == 3) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2)); if ((i & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(p[i].x)); }
        if (OP == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca" : "+v"(a[i]) : "v"(c1), "v"(c2));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}
template <int OP>
void run(const char* name, float* d, double ops_per_instr) {
  const int wps = 8, blocks = 256 * wps, iters = 4096;
  hipEvent_t e[41];
  for (auto& x : e) hipEventCreate(&x);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, iters);
  hipDeviceSynchronize();
  hipEventRecord(e[0]);
  for (int i = 0; i < 40; ++i) { hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, iters); hipEventRecord(e[i + 1]); }
  hipEventSynchronize(e[40]);
  printf("%-16s ms/launch:", name);
  float first = 0, last = 0;
  for (int i = 0; i < 40; ++i) { float ms; hipEventElapsedTime(&ms, e[i], e[i + 1]); if (i < 3) first += ms / 3; if (i >= 37) last += ms / 3; if (i % 4 == 0) printf(" %.2f", ms); }
  double instr = (double)wps * REP * iters;  // per SIMD per launch
  printf("\n    first3 %.3f ms  last3 %.3f ms -> sustained %.2f ns/instr/SIMD, %.1f T lane-ops/s chip\n", first, last,
         last * 1e6 / instr, ops_per_instr * 64 * instr * 1024 / (last * 1e-3) / 1e12);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  run<0>("v_fma_f32", d, 1);
  run<1>("v_pk_fma_f32", d, 2);
  run<2>("v_exp_f32", d, 1);
  run<3>("fma+exp(1:4)", d, 1.25);
  run<4>("v_bitop3", d, 1);
  run<0>("v_fma_f32 again", d, 1);
  return 0;
}
