// issue_mix.hip -- how gfx950 issues MIXED VALU streams ("2 or 4 cycles per plain op?").
//
// valu_rate.hip measured pure streams: fma/mul/add/shift/logic ~2.3 cycles per wave-instruction per SIMD (two waves
// share a quad-cycle), packed / min / max / cmp / cvt ~4.2, transcendentals ~8.2.  The fused layer kernel runs at
// ~4.4 cycles per instruction although 2/3 of its instructions are of the 2.3-cycle kind.  This probe runs mixed
// streams (same kinds, different interleavings and dependency shapes) and reports cycles per instruction from the
// in-kernel clock (s_memtime / s_memrealtime), next to the additive prediction from the pure streams.
//
// Build: hipcc -O3 --offload-arch=gfx950 -o issue_mix issue_mix.hip        Run: ./issue_mix [waves_per_simd ...]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

// Every pattern is ONE asm statement (hipcc pads separate inline-asm statements with s_nop, which would take issue
// slots of their own).  Operands: %0-%7 = a[0..7], %8-%15 = p[0..7] (register pairs), %16 = c1, %17 = c2,
// %18 = {c1,c1}, %19 = {c2,c2}.  No instruction reads a transcendental's result in the very next slot (gfx940+
// needs one wait state there).
#define STR_(x) #x
#define STR(x) STR_(x)
#define PIDX0 8
#define PIDX1 9
#define PIDX2 10
#define PIDX3 11
#define PIDX4 12
#define PIDX5 13
#define PIDX6 14
#define PIDX7 15
#define A(i) "%" #i
#define PP(i) "%" STR(PIDX##i)
#define FMA(i) "v_fma_f32 " A(i) ", " A(i) ", %16, %17\n"
#define MUL(i) "v_mul_f32 " A(i) ", " A(i) ", %16\n"
#define SUB(i) "v_sub_f32 " A(i) ", %16, " A(i) "\n"
#define ASHR(i) "v_ashrrev_i32 " A(i) ", 31, " A(i) "\n"
#define BITOP(i) "v_bitop3_b32 " A(i) ", " A(i) ", %16, %17 bitop3:0xca\n"
#define PKFMA(i) "v_pk_fma_f32 " PP(i) ", " PP(i) ", %18, %19\n"
#define PKMUL(i) "v_pk_mul_f32 " PP(i) ", " PP(i) ", %18\n"
#define EXP(i) "v_exp_f32 " A(i) ", " A(i) "\n"
#define LOG(i) "v_log_f32 " A(i) ", " A(i) "\n"
#define MAX(i) "v_max_f32 " A(i) ", " A(i) ", %16\n"
#define FMAS(i) "v_fma_f32 " A(i) ", " A(i) ", s20, %17\n"
#define FMAK(i) "v_fmamk_f32 " A(i) ", " A(i) ", 0x3f8ccccd, %17\n"
#define MULC(i) "v_mul_f32_e64 " A(i) ", " A(i) ", %16 clamp\n"
// dependent pair: a[i] feeds a[j] -- a chain hopping over registers
#define FMAD(i, j) "v_fma_f32 " A(j) ", " A(i) ", %16, %17\n"

#define X8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define X4(M) M(0) M(1) M(2) M(3)

// One repetition of each pattern; N = instructions in it.  NF/NH/NQ = fast / half-rate / quarter-rate counts.
template <int P>
struct Pat;
#define PATTERN(ID, NAME, NFAST, NHALF, NQUART, BODY)                                            \
  template <>                                                                                      \
  struct Pat<ID> {                                                                                 \
    static constexpr const char* name = NAME;                                                      \
    static constexpr int nf = NFAST, nh = NHALF, nq = NQUART;                                      \
    static __device__ __forceinline__ void run(float (&a)[8], v2f (&p)[8], float c1, float c2, v2f k1, v2f k2) {         \
      asm volatile(BODY                                                                            \
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),   \
                     "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])    \
                   : "v"(c1), "v"(c2), "v"(k1), "v"(k2)                                              \
                   : "s20");                                                                       \
    }                                                                                              \
  };

PATTERN(0, "fma x8 (8 independent chains)", 8, 0, 0, X8(FMA))
PATTERN(1, "pk_fma x8", 0, 8, 0, X8(PKFMA))
PATTERN(2, "exp x8", 0, 0, 8, X8(EXP))
PATTERN(3, "fma,ashr alternating", 8, 0, 0, FMA(0) ASHR(1) FMA(2) ASHR(3) FMA(4) ASHR(5) FMA(6) ASHR(7))
PATTERN(4, "fma,mul,sub,ashr,bitop3 soup", 8, 0, 0, FMA(0) MUL(1) SUB(2) ASHR(3) BITOP(4) FMA(5) MUL(6) SUB(7))
PATTERN(5, "fma,pk_fma alternating", 4, 4, 0, FMA(0) PKFMA(1) FMA(2) PKFMA(3) FMA(4) PKFMA(5) FMA(6) PKFMA(7))
PATTERN(6, "fma x8 then pk_fma x8 (runs)", 8, 8, 0, X8(FMA) X8(PKFMA))
PATTERN(7, "fma x4, exp x1", 4, 0, 1, X4(FMA) EXP(4))
PATTERN(8, "fma x8, exp x2", 8, 0, 2, X8(FMA) EXP(0) EXP(1))
PATTERN(9, "fma x32, exp x8 (runs)", 32, 0, 8, X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(EXP))
PATTERN(10, "fma x96, exp x24 (long runs)", 96, 0, 24,
        X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA)
        X8(EXP) X8(EXP) X8(EXP))
PATTERN(11, "select: sub,ashr,bitop3 element-major (dependent neighbours)", 24, 0, 0,
        SUB(0) ASHR(0) BITOP(0) SUB(1) ASHR(1) BITOP(1) SUB(2) ASHR(2) BITOP(2) SUB(3) ASHR(3) BITOP(3)
        SUB(4) ASHR(4) BITOP(4) SUB(5) ASHR(5) BITOP(5) SUB(6) ASHR(6) BITOP(6) SUB(7) ASHR(7) BITOP(7))
PATTERN(12, "select: sub x8, ashr x8, bitop3 x8 (op-major)", 24, 0, 0, X8(SUB) X8(ASHR) X8(BITOP))
PATTERN(13, "fma ONE dependent chain", 8, 0, 0, FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0))
PATTERN(14, "fma TWO dependent chains", 8, 0, 0, FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1))
PATTERN(15, "fma FOUR dependent chains", 8, 0, 0, X4(FMA) X4(FMA))
PATTERN(16, "fma with SGPR operand x8", 8, 0, 0, X8(FMAS))
PATTERN(17, "fmamk literal x8", 8, 0, 0, X8(FMAK))
PATTERN(18, "mul_e64 clamp x8", 8, 0, 0, X8(MULC))
PATTERN(19, "max x8", 0, 8, 0, X8(MAX))
PATTERN(20, "fma,max alternating", 4, 4, 0, FMA(0) MAX(1) FMA(2) MAX(3) FMA(4) MAX(5) FMA(6) MAX(7))
PATTERN(21, "layer-like: 12 log, 12 pk_mul, 12 exp, 24 pk_fma, 72 fast", 72, 36, 24,
        X8(LOG) X4(LOG) X8(PKMUL) X4(PKMUL) X8(EXP) X4(EXP) X8(PKFMA) X8(PKFMA) X8(PKFMA)
        X8(FMA) X8(SUB) X8(ASHR) X8(BITOP) X8(MUL) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP))
PATTERN(22, "layer-like with pk replaced by 2 plain each", 144, 0, 24,
        X8(LOG) X4(LOG) X8(MUL) X8(MUL) X8(MUL) X8(EXP) X4(EXP) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA)
        X8(FMA) X8(SUB) X8(ASHR) X8(BITOP) X8(MUL) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP))
PATTERN(23, "fma dependent hop chain (a[i]->a[i+1])", 8, 0, 0,
        FMAD(0, 1) FMAD(1, 2) FMAD(2, 3) FMAD(3, 4) FMAD(4, 5) FMAD(5, 6) FMAD(6, 7) FMAD(7, 0))
#define NPAT 24

struct Stamp {
  unsigned long long t0, t1, r0, r1;
};

template <int P>
__global__ __launch_bounds__(256) void k(float* out, Stamp* st, float seed, int iters) {
  float a[8];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 1e-3f + i;
    p[i] = v2f{a[i], a[i] + 0.5f};
  }
  const float c1 = seed * 0.999f, c2 = seed * 1e-3f;
  const v2f k1 = v2f{c1, c1}, k2 = v2f{c2, c2};
  asm volatile("s_mov_b32 s20, 0x3f7fbe77" ::: "s20");
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    Pat<P>::run(a, p, c1, c2, k1, k2);
    Pat<P>::run(a, p, c1, c2, k1, k2);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + threadIdx.x / 64] = Stamp{t0, t1, r0, r1};
}

static double g_pure[3] = {0, 0, 0};  // measured cycles per instruction of the pure fast / half / quarter streams

template <int P>
void run(float* d, Stamp* dst, int wps) {
  typedef Pat<P> T;
  const int n_instr = T::nf + T::nh + T::nq;
  const int target = 1 << 20;  // instructions per wave
  const int iters = target / (2 * n_instr);
  const int blocks = 256 * wps;  // 4 waves per block = one per SIMD; 256 CUs
  std::vector<Stamp> h(blocks * 4);
  float best_ms = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {  // the last repetitions run at the settled clock
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, d, dst, 1.0f, iters);
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
  const double ghz = clk[clk.size() / 2];
  const double n_total = 2.0 * n_instr * iters;  // per wave
  // SIMD cycles per instruction: the wave's lifetime is shared by wps waves on its SIMD
  const double cpi = cyc[cyc.size() / 2] / n_total / wps;
  double pred = 0;
  if (g_pure[0] > 0) pred = (T::nf * g_pure[0] + T::nh * g_pure[1] + T::nq * g_pure[2]) / n_instr;
  printf("w/SIMD %d  %-62s %7.3f ms  clock %.2f GHz  %5.2f cyc/instr/SIMD", wps, T::name, best_ms, ghz, cpi);
  if (pred > 0) printf("  additive %5.2f  ratio %.2f", pred, cpi / pred);
  printf("\n");
  fflush(stdout);
  if (P == 0) g_pure[0] = cpi;
  if (P == 1) g_pure[1] = cpi;
  if (P == 2) g_pure[2] = cpi;
}

template <int P>
struct RunAll {
  static void go(float* d, Stamp* st, int wps) {
    RunAll<P - 1>::go(d, st, wps);
    run<P>(d, st, wps);
  }
};
template <>
struct RunAll<-1> {
  static void go(float*, Stamp*, int) {}
};

int main(int argc, char** argv) {
  float* d;
  Stamp* st;
  hipMalloc(&d, 4096);
  hipMalloc(&st, sizeof(Stamp) * 256 * 8 * 4);
  std::vector<int> wpss;
  for (int i = 1; i < argc; ++i) wpss.push_back(atoi(argv[i]));
  if (wpss.empty()) wpss = {8, 4, 2, 1};
  for (int wps : wpss) {
    g_pure[0] = g_pure[1] = g_pure[2] = 0;
    RunAll<NPAT - 1>::go(d, st, wps);
    printf("\n");
  }
  return 0;
}
