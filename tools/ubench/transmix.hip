// Does interleaving transcendental ops with FMA-class ops cost more than clustering them?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define F4 "v_fma_f32 v20, v24, v25, v26\n v_fma_f32 v21, v27, v28, v29\n v_fma_f32 v22, v30, v31, v32\n v_fma_f32 v23, v33, v34, v35\n"
#define E1(r) "v_exp_f32 " r ", " r "\n"
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43"
#define INIT asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.25\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n v_mov_b32 v36, 0.5\n v_mov_b32 v37, 0.5\n v_mov_b32 v38, 0.5\n v_mov_b32 v39, 0.5\n v_mov_b32 v40, 0.5\n v_mov_b32 v41, 0.5\n v_mov_b32 v42, 0.5\n v_mov_b32 v43, 0.5" ::: CLOB);
__global__ void k_inter(float* o, int iters) {  // 8 x [4 fma, 1 exp]
  INIT
  for (int it = 0; it < iters; ++it)
    asm volatile(F4 E1("v36") F4 E1("v37") F4 E1("v38") F4 E1("v39") F4 E1("v40") F4 E1("v41") F4 E1("v42") F4 E1("v43") ::: CLOB);
  if (iters < 0) o[0] = 1;
}
__global__ void k_clust(float* o, int iters) {  // 32 fma then 8 exp
  INIT
  for (int it = 0; it < iters; ++it)
    asm volatile(F4 F4 F4 F4 F4 F4 F4 F4 E1("v36") E1("v37") E1("v38") E1("v39") E1("v40") E1("v41") E1("v42") E1("v43") ::: CLOB);
  if (iters < 0) o[0] = 1;
}
__global__ void k_fma(float* o, int iters) {  // 32 fma only
  INIT
  for (int it = 0; it < iters; ++it) asm volatile(F4 F4 F4 F4 F4 F4 F4 F4 ::: CLOB);
  if (iters < 0) o[0] = 1;
}
__global__ void k_exp(float* o, int iters) {  // 8 exp only
  INIT
  for (int it = 0; it < iters; ++it) asm volatile(E1("v36") E1("v37") E1("v38") E1("v39") E1("v40") E1("v41") E1("v42") E1("v43") ::: CLOB);
  if (iters < 0) o[0] = 1;
}
__global__ void k_dep(float* o, int iters) {  // exp result consumed by the very next fma (trans-use hazard)
  INIT
  for (int it = 0; it < iters; ++it)
    asm volatile(E1("v36") "v_fma_f32 v20, v36, v25, v26\n" F4 E1("v37") "v_fma_f32 v21, v37, v25, v26\n" F4 E1("v38") "v_fma_f32 v22, v38, v25, v26\n" F4 E1("v39") "v_fma_f32 v23, v39, v25, v26\n" F4 ::: CLOB);
  if (iters < 0) o[0] = 1;
}
template <typename K> void run(const char* n, K kern, float* d, int waves) {
  int iters = 4096; hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256 * waves), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(256 * waves), dim3(256), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("%-12s waves/SIMD %d: %.3f ms  -> %.2f ns per loop body per SIMD\n", n, waves, ms, ms * 1e6 / ((double)waves * iters));
}
int main() { float* d; hipMalloc(&d, 64);
  for (int w : {8, 4, 2}) { run("fma32", k_fma, d, w); run("exp8", k_exp, d, w); run("interleaved", k_inter, d, w); run("clustered", k_clust, d, w); run("exp->use", k_dep, d, w); }
  return 0; }
