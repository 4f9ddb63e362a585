// Occupancy probe (round 3, exp27): the 3-plane copy of copy3.hip (256-thread tiles, non-temporal, U float4 groups per lane)
// with the number of resident workgroups per CU held down by reserving dynamic LDS nobody uses.  Prints what the runtime
// says fits (hipOccupancyMaxActiveBlocksPerMultiprocessor) beside the sustained time: 1000 launches after 200.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_tile(const v4f* __restrict__ in, v4f* __restrict__ out, unsigned n) {
  extern __shared__ float unused_lds[];
  const unsigned img = blockIdx.y;
  const size_t plane = n;
  const v4f* p = in + (size_t)img * 3 * plane;
  v4f* q = out + (size_t)img * 3 * plane;
  unsigned base = blockIdx.x * (256u * U) + threadIdx.x;
  v4f a[U], b[U], c[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned i = min(base + u * 256u, n - 1);
    a[u] = __builtin_nontemporal_load(p + i); b[u] = __builtin_nontemporal_load(p + plane + i); c[u] = __builtin_nontemporal_load(p + 2 * plane + i);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned i = base + u * 256u;
    if (i < n) {
      __builtin_nontemporal_store(a[u] * 1.01f, q + i); __builtin_nontemporal_store(b[u] * 1.01f, q + plane + i);
      __builtin_nontemporal_store(c[u] * 1.01f, q + 2 * plane + i);
    }
  }
}

template <int U>
static void sweep(v4f** in, v4f* out, unsigned n, int B, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
  hipFuncSetAttribute((const void*)k_tile<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const unsigned lds[] = {0, 16384, 20480, 23296, 27136, 32768, 36864, 40960, 45056, 49152, 54528, 65536, 81920, 163840};
  for (unsigned l : lds) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile<U>, 256, l);
    auto launch = [&](v4f* src) { hipLaunchKernelGGL(k_tile<U>, dim3((n + 256 * U - 1) / (256 * U), B), dim3(256), l, 0, src, out, n); };
    for (int i = 0; i < 200; ++i) launch(in[i & 1]);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("U=%d lds=%u: launch failed\n", U, l); continue; }
    hipEventRecord(e0);
    for (int i = 0; i < 1000; ++i) launch(in[i & 1]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 1000;
    printf("copy U=%d  dynamic LDS %6u B  -> %2d workgroups per CU  %7.1f us  %7.1f GB/s\n", U, l, nb, ms * 1e3, 2.0 * bytes / ms / 1e6);
    fflush(stdout);
  }
}

// how many workgroups REALLY share a CU for a given reservation: every workgroup spins a fixed number of clock ticks, so the
// launch takes (workgroups per CU / resident) x spin
__global__ __launch_bounds__(256) void k_spin(unsigned long long ticks, unsigned* sink) {
  extern __shared__ float unused_lds[];
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *sink = 1;
}
static void residency(hipEvent_t e0, hipEvent_t e1) {
  hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int per_cu = 48;
  const unsigned long long ticks = 2000000ull;  // 20 ms at the 100 MHz constant clock
  float base = 0;
  for (unsigned l = 0; l <= 163840; l += (l < 81920 ? 4096 : 16384)) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spin, 256, l);
    hipLaunchKernelGGL(k_spin, dim3(cus * per_cu), dim3(256), l, 0, 1000ull, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_spin, dim3(cus * per_cu), dim3(256), l, 0, ticks / per_cu, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (l == 0) base = ms;
    printf("spin  dynamic LDS %6u B  runtime says %2d per CU  %7.2f ms  -> resident = %.2f x the no-reservation case\n", l, nb, ms, base / ms);
    fflush(stdout);
  }
}

int main() {
  const int B = 32, H = 1000, W = 1500;
  const unsigned n = H * W / 4;
  const size_t bytes = (size_t)B * 3 * n * 16;
  v4f *in[2], *out;
  for (auto& p : in) { hipMalloc(&p, bytes); hipMemset(p, 0x3c, bytes); }
  hipMalloc(&out, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  residency(e0, e1);
  sweep<1>(in, out, n, B, bytes, e0, e1);
  sweep<2>(in, out, n, B, bytes, e0, e1);
  sweep<4>(in, out, n, B, bytes, e0, e1);
  return 0;
}
