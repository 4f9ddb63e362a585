// sincos_probe.hip -- how far are v_sin_f32 / v_cos_f32 (input in revolutions) from what the reference's HSV cone computes,
// cos(float32(2*pi) * h) and sin(...) in float32 (model.py:70-75), for h in [0, 1]?  Decides whether the CURLLoss kernels
// may use the hardware instructions instead of the library's range-reduced sinf / cosf.
// Build: hipcc -O3 --offload-arch=gfx950 -o sincos_probe sincos_probe.hip        Run: ./sincos_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

__global__ void k(float* s, float* c, float* sl, float* cl, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  float h = (float)i / (float)n;
  s[i] = __builtin_amdgcn_sinf(h);
  c[i] = __builtin_amdgcn_cosf(h);
  const float a = 6.2831855f * h;
  sl[i] = sinf(a);
  cl[i] = cosf(a);
}

int main() {
  const int n = 1 << 22;
  float *s, *c, *sl, *cl;
  hipMalloc(&s, 4 * (n + 1)), hipMalloc(&c, 4 * (n + 1)), hipMalloc(&sl, 4 * (n + 1)), hipMalloc(&cl, 4 * (n + 1));
  hipLaunchKernelGGL(k, dim3((n + 256) / 256), dim3(256), 0, 0, s, c, sl, cl, n);
  std::vector<float> hs(n + 1), hc(n + 1), hsl(n + 1), hcl(n + 1);
  hipMemcpy(hs.data(), s, 4 * (n + 1), hipMemcpyDeviceToHost), hipMemcpy(hc.data(), c, 4 * (n + 1), hipMemcpyDeviceToHost);
  hipMemcpy(hsl.data(), sl, 4 * (n + 1), hipMemcpyDeviceToHost), hipMemcpy(hcl.data(), cl, 4 * (n + 1), hipMemcpyDeviceToHost);
  double e_hw_exact = 0, e_hw_ref = 0, e_lib_ref = 0, e_ref_exact = 0, sum_hw_ref = 0;
  for (int i = 0; i <= n; ++i) {
    const float h = (float)i / (float)n;
    const float a = 6.2831855f * h;
    const float rs = sinf(a), rc = cosf(a);  // the reference's float32 values (host libm)
    const double xs = sin(2 * M_PI * (double)h), xc = cos(2 * M_PI * (double)h);
    e_hw_exact = fmax(e_hw_exact, fmax(fabs(hs[i] - xs), fabs(hc[i] - xc)));
    e_hw_ref = fmax(e_hw_ref, fmax(fabs((double)hs[i] - rs), fabs((double)hc[i] - rc)));
    e_lib_ref = fmax(e_lib_ref, fmax(fabs((double)hsl[i] - rs), fabs((double)hcl[i] - rc)));
    e_ref_exact = fmax(e_ref_exact, fmax(fabs(rs - xs), fabs(rc - xc)));
    sum_hw_ref += fabs((double)hs[i] - rs) + fabs((double)hc[i] - rc);
  }
  printf("h = i / 2^22, i = 0..2^22\n");
  printf("max |v_sin/v_cos(h)        - exact sin/cos(2 pi h)|          = %.3e\n", e_hw_exact);
  printf("max |v_sin/v_cos(h)        - reference float32 value|        = %.3e   (mean %.3e)\n", e_hw_ref, sum_hw_ref / (2.0 * (n + 1)));
  printf("max |device sinf/cosf(a)   - reference float32 value|        = %.3e\n", e_lib_ref);
  printf("max |reference float32     - exact|  (its own rounding)      = %.3e\n", e_ref_exact);
  return 0;
}
