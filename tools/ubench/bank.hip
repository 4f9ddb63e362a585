// VGPR bank-conflict probe: v_fma_f32 with its three sources in one bank (index mod 4 equal) vs spread.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define BODY(INS) for (int it = 0; it < iters; ++it) { asm volatile(INS INS INS INS INS INS INS INS INS INS INS INS INS INS INS INS ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"); }
__global__ void k_same(float* o, int iters) {
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v28, 0.5\n v_mov_b32 v32, 0.25\n v_mov_b32 v36, 1.0\n v_mov_b32 v40, 1.0\n v_mov_b32 v44, 1.0" ::: "v24","v28","v32","v36","v40","v44");
  BODY("v_fma_f32 v20, v24, v28, v32\n v_fma_f32 v21, v36, v40, v44\n v_fma_f32 v22, v24, v28, v32\n v_fma_f32 v23, v36, v40, v44\n")
  if (iters < 0) o[0] = 1;
}
__global__ void k_spread(float* o, int iters) {
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v29, 0.5\n v_mov_b32 v34, 0.25\n v_mov_b32 v37, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v47, 1.0" ::: "v24","v29","v34","v37","v42","v47");
  BODY("v_fma_f32 v20, v24, v29, v34\n v_fma_f32 v21, v37, v42, v47\n v_fma_f32 v22, v24, v29, v34\n v_fma_f32 v23, v37, v42, v47\n")
  if (iters < 0) o[0] = 1;
}
__global__ void k_two(float* o, int iters) {  // 2-source ops: v_mul same bank vs different
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v28, 0.5\n v_mov_b32 v36, 1.0\n v_mov_b32 v40, 1.0" ::: "v24","v28","v36","v40");
  BODY("v_mul_f32 v20, v24, v28\n v_mul_f32 v21, v36, v40\n v_mul_f32 v22, v24, v28\n v_mul_f32 v23, v36, v40\n")
  if (iters < 0) o[0] = 1;
}
__global__ void k_dep(float* o, int iters) {  // distinct registers every instruction (no operand reuse)
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.25\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0" ::: "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
  BODY("v_fma_f32 v20, v24, v25, v26\n v_fma_f32 v21, v27, v28, v29\n v_fma_f32 v22, v30, v31, v32\n v_fma_f32 v23, v33, v34, v35\n")
  if (iters < 0) o[0] = 1;
}
__global__ void k_pair(float* o, int iters) {  // two of the three sources share a bank
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v28, 0.5\n v_mov_b32 v33, 0.25\n v_mov_b32 v36, 1.0\n v_mov_b32 v40, 1.0\n v_mov_b32 v45, 1.0" ::: "v24","v28","v33","v36","v40","v45");
  BODY("v_fma_f32 v20, v24, v28, v33\n v_fma_f32 v21, v36, v40, v45\n v_fma_f32 v22, v24, v28, v33\n v_fma_f32 v23, v36, v40, v45\n")
  if (iters < 0) o[0] = 1;
}
__global__ void k_pair2(float* o, int iters) {  // src0 and src2 share a bank
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v29, 0.5\n v_mov_b32 v32, 0.25\n v_mov_b32 v36, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v44, 1.0" ::: "v24","v29","v32","v36","v41","v44");
  BODY("v_fma_f32 v20, v24, v29, v32\n v_fma_f32 v21, v36, v41, v44\n v_fma_f32 v22, v24, v29, v32\n v_fma_f32 v23, v36, v41, v44\n")
  if (iters < 0) o[0] = 1;
}
__global__ void k_fmac(float* o, int iters) {  // v_fmac: dst is the third source; dst bank == src bank
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v28, 0.5\n v_mov_b32 v36, 1.0\n v_mov_b32 v40, 1.0" ::: "v24","v28","v36","v40");
  BODY("v_fmac_f32 v20, v24, v28\n v_fmac_f32 v32, v36, v40\n v_fmac_f32 v44, v24, v28\n v_fmac_f32 v20, v36, v40\n")
  if (iters < 0) o[0] = 1;
}
__global__ void k_bitop(float* o, int iters) {
  asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v28, 0.5\n v_mov_b32 v32, 0.25\n v_mov_b32 v36, 1.0\n v_mov_b32 v40, 1.0\n v_mov_b32 v44, 1.0" ::: "v24","v28","v32","v36","v40","v44");
  BODY("v_bitop3_b32 v20, v24, v28, v32 bitop3:0xca\n v_bitop3_b32 v21, v36, v40, v44 bitop3:0xca\n v_bitop3_b32 v22, v24, v28, v32 bitop3:0xca\n v_bitop3_b32 v23, v36, v40, v44 bitop3:0xca\n")
  if (iters < 0) o[0] = 1;
}
template <typename K> void run(const char* n, K kern, float* d) {
  int iters = 2048; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  double instr = 8.0 * 64 * iters;  // per SIMD: 8 waves x 64 instr per iter
  printf("%-10s %.3f ms  %.3f ns/instr/SIMD\n", n, ms, ms * 1e6 / instr);
}
int main() { float* d; hipMalloc(&d, 64); run("same-bank", k_same, d); run("spread", k_spread, d); run("mul2", k_two, d); run("distinct", k_dep, d); run("pair01", k_pair, d); run("pair02", k_pair2, d); run("fmac-same", k_fmac, d); run("bitop-same", k_bitop, d); return 0; }
