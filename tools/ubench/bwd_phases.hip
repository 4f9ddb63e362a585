// bwd_phases.hip -- what would "the forward's trick in reverse" buy the layer backward at best?  
//
// layer_bwd_kernel runs a lane's four pixels one after the other: per pixel ~386 plain VALU instructions (a quarter of them
// v_cmp_e64 -> SGPR pair -> v_cndmask_e64 gates, a tenth with an SGPR coefficient operand) and 23 transcendentals in runs of
// THREE at raised issue priority.  The forward runs four pixels in phases: runs of TWELVE.  This probe issues the backward's
// instruction mix per pixel -- the same multiset of instructions per pixel in every arrangement, dependencies hopping over
// eight registers -- as
//   serial   : pixel after pixel, transcendental runs of 3                      (what the kernel does)
//   phases2  : two pixels per phase, runs of 6, plain runs twice as long
//   phases4  : four pixels per phase, runs of 12
// at a given number of resident waves per SIMD (the real kernel holds 4 at 109-120 VGPRs; two pixels' tapes would leave 3,
// four pixels' 2 at most) and reports the median wave's lifetime per instruction (in-kernel clock) and SIMD cycles per instruction.  No masks to keep alive, no
// tapes, no spills: the UPPER bound of what the rearrangement itself is worth.
//
// Build: hipcc -O3 --offload-arch=gfx950 -o bwd_phases bwd_phases.hip      Run: ./bwd_phases 4 3 2
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"  // (a probe: hip calls unchecked)
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define A(i) "%" #i
#define FMA(i) "v_fma_f32 " A(i) ", " A(i) ", %8, %9\n"
#define MUL(i) "v_mul_f32 " A(i) ", " A(i) ", %8\n"
#define ADD(i) "v_add_f32 " A(i) ", " A(i) ", %9\n"
#define FMAS(i) "v_fma_f32 " A(i) ", " A(i) ", s20, %9\n"                 /* a collapsed-curve coefficient from an SGPR */
#define FMAK(i) "v_fmamk_f32 " A(i) ", " A(i) ", 0x3f8ccccd, %9\n"        /* a matrix constant as a literal */
#define CLAMP(i) "v_max_f32_e64 " A(i) ", " A(i) ", " A(i) " clamp\n"
#define GATE(i, s) "v_cmp_le_f32_e64 s[" #s ":" #s "+1], " A(i) ", %8\n"  /* a gate: compare into an SGPR pair ... */
#define KEEP(i, s) "v_cndmask_b32_e64 " A(i) ", 0, " A(i) ", s[" #s ":" #s "+1]\n"  /* ... applied later */
#define LOG(i) "v_log_f32 " A(i) ", " A(i) "\n"
#define EXP(i) "v_exp_f32 " A(i) ", " A(i) "\n"
#define RCP(i) "v_rcp_f32 " A(i) ", " A(i) "\n"
#define PRIO(n) "s_setprio " #n "\n"

// 16 plain instructions: the backward's mix (mul 4, fma 3 of which one with an SGPR operand, fmamk 1, add 2, clamp 1,
// gate 2, keep 2, and one more mul) -- registers rotate so that no instruction reads its predecessor's result
#define P16(g0, g1)                                                                                                  \
  MUL(0) FMA(1) GATE(2, g0) FMAS(3) MUL(4) KEEP(5, g1) FMAK(6) ADD(7) MUL(1) CLAMP(0) GATE(3, g1) FMA(2) KEEP(4, g0) MUL(6) \
  ADD(5) MUL(7)
#define P32 P16(22, 24) P16(26, 28)
#define P64 P32 P32
#define P96 P64 P32
#define P128 P64 P64
#define T3(OP) OP(0) OP(3) OP(6)
#define T2(OP) OP(1) OP(5)

// one pixel of the backward, N copies per phase:  3 log | 32 | 3 exp | 32 | 3 log | 32 | 3 exp (+3: the cube root's tape) | 64
//   | 3 log | 3 exp | 3 rcp | 96 | 2 rcp | 128     = 384 plain + 23 transcendental
#define R1(X) X
#define R2(X) X X
#define R4(X) X X X X
#define PIXEL_PHASES(R)                                                                                              \
  PRIO(1) R(T3(LOG)) PRIO(0) R(P32) PRIO(1) R(T3(EXP)) PRIO(0) R(P32) PRIO(1) R(T3(LOG)) PRIO(0) R(P32) PRIO(1) R(T3(EXP)) \
  PRIO(0) R(P64) PRIO(1) R(T3(LOG)) R(T3(EXP)) R(T3(RCP)) PRIO(0) R(P96) PRIO(1) R(T2(RCP)) PRIO(0) R(P128)

struct Stamp {
  unsigned long long t0, t1, r0, r1;
};

template <int ARR>
__device__ __forceinline__ void body(float (&a)[8], float c1, float c2);
#define BODY(ARR, TEXT)                                                                                              \
  template <>                                                                                                        \
  __device__ __forceinline__ void body<ARR>(float (&a)[8], float c1, float c2) {                                      \
    asm volatile(TEXT                                                                                                \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])    \
                 : "v"(c1), "v"(c2)                                                                                  \
                 : "s20", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");                                   \
  }
// every arrangement: FOUR pixels per call
BODY(1, PIXEL_PHASES(R1) PIXEL_PHASES(R1) PIXEL_PHASES(R1) PIXEL_PHASES(R1))
BODY(2, PIXEL_PHASES(R2) PIXEL_PHASES(R2))
BODY(4, PIXEL_PHASES(R4))
// the same without any priority changes (what the arrangement is worth when nothing lines the waves up)
#undef PRIO
#define PRIO(n) ""
BODY(11, PIXEL_PHASES(R1) PIXEL_PHASES(R1) PIXEL_PHASES(R1) PIXEL_PHASES(R1))
BODY(14, PIXEL_PHASES(R4))

template <int ARR>
__global__ __launch_bounds__(256) void k(float* out, Stamp* st, float seed, int iters) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
  asm volatile("s_mov_b32 s20, 0x3f7fbe77" ::: "s20");
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) body<ARR>(a, c1, c2);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + threadIdx.x / 64] = Stamp{t0, t1, r0, r1};
}

template <int ARR>
double run(const char* name, float* d, Stamp* dst, int wps) {
  const int n_instr = 4 * (384 + 23);
  const int iters = (1 << 20) / n_instr;
  const int blocks = 256 * wps;  // 4 waves per block = one per SIMD; 256 CUs
  std::vector<Stamp> h(blocks * 4);
  float best_ms = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ARR>, dim3(blocks), dim3(256), 0, 0, d, dst, 1.0f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) best_ms = std::min(best_ms, ms);
  }
  hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (auto& s : h) {
    clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);
    cyc.push_back((double)(s.t1 - s.t0));
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  // the median wave's lifetime in shader-clock ticks per instruction it issued: clock-independent (the runs of one session
  // differ by 4 % in clock while the chip settles), the figure the arrangements are compared on
  const double ticks = cyc[cyc.size() / 2] / ((double)n_instr * iters);
  // ... and SIMD cycles per instruction from the launch's duration: waves x instructions per SIMD over duration x clock
  const double cpi = best_ms * 1e-3 * clk[clk.size() / 2] * 1e9 / ((double)wps * n_instr * iters);
  printf("waves/SIMD %d  %-44s %7.3f ms  clock %.2f GHz  %6.3f wave ticks per instruction  %5.2f SIMD cycles per instruction\n", wps,
         name, best_ms, clk[clk.size() / 2], ticks, cpi);
  fflush(stdout);
  return ticks;
}

int main(int argc, char** argv) {
  float* d;
  Stamp* st;
  hipMalloc(&d, 4096);
  hipMalloc(&st, sizeof(Stamp) * 256 * 8 * 4);
  std::vector<int> wpss;
  for (int i = 1; i < argc; ++i) wpss.push_back(atoi(argv[i]));
  if (wpss.empty()) wpss = {4, 3, 2};
  double base = 0;
  for (int wps : wpss) {
    double s = run<1>("serial (runs of 3)", d, st, wps);
    if (base == 0) base = s;
    double p2 = run<2>("two pixels per phase (runs of 6)", d, st, wps);
    double p4 = run<4>("four pixels per phase (runs of 12)", d, st, wps);
    double s0 = run<11>("serial, no priority changes", d, st, wps);
    double p40 = run<14>("four pixels per phase, no priority changes", d, st, wps);
    (void)base;
    printf("   wave lifetime per instruction vs serial at %d waves/SIMD: two pixels per phase %+.1f %%, four %+.1f %%;  without the "
           "priority changes: serial %+.1f %%, four per phase %+.1f %%\n\n",
           wps, (p2 / s - 1) * 100, (p4 / s - 1) * 100, (s0 / s - 1) * 100, (p40 / s - 1) * 100);
  }
  return 0;
}
