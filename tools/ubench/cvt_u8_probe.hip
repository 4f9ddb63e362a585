// cvt_u8_probe.hip -- what does v_cvt_pk_u8_f32 do with fractions, negatives, > 255, NaN?  (Candidate for the byte
// egress of the uint8 kernels: mul(255) + ONE instruction that converts, saturates and packs, instead of max/min/cvt/
// shift/or -- usable only if it TRUNCATES like (x*255).astype('uint8') / mul(255).byte().)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned r = 0;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
  out[i] = r;
}
int main() {
  const float v[] = {0.0f, 0.4f, 0.5f, 0.6f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 254.99f, 255.0f, 255.5f,
                     256.0f, 300.0f, 1e9f, -0.4f, -0.6f, -5.0f, NAN, INFINITY, -INFINITY, 127.5f, 128.5f, 0.999999f, 1.9999999f};
  const int n = sizeof(v) / sizeof(v[0]);
  float* din;
  unsigned* dout;
  hipMalloc(&din, n * 4);
  hipMalloc(&dout, n * 4);
  hipMemcpy(din, v, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, n);
  unsigned o[64];
  hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%14.7f -> %3u   (truncation would give %d)\n", v[i], o[i] & 0xff,
                                     isnan(v[i]) ? 0 : (v[i] < 0 ? 0 : (v[i] > 255 ? 255 : (int)v[i])));
  return 0;
}
