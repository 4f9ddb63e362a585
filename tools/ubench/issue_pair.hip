// issue_pair.hip -- WHICH VALU instructions gfx950 issues two-per-quad-cycle (SQ_ACTIVE_INST_VALU2), and what s_setprio does to it.
// Same harness as issue_mix.hip; meant to be run under rocprofv3 --pmc SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_CYCLES.
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
#define CVT(i) "v_cvt_f32_i32 " A(i) ", " A(i) "\n"
#define MULS2(i) "v_mul_f32 " A(i) ", s20, " A(i) "\n"          /* VOP2, SGPR src0 */
#define MULI(i) "v_mul_f32 " A(i) ", 2.0, " A(i) "\n"           /* inline constant */
#define ADDL(i) "v_add_f32 " A(i) ", 0x3f8ccccd, " A(i) "\n"    /* VOP2 literal */
#define MOV(i) "v_mov_b32 " A(i) ", %16\n"
#define MOVS(i) "v_mov_b32 " A(i) ", s20\n"
#define MED3(i) "v_med3_f32 " A(i) ", " A(i) ", 0, 1.0\n"
#define PKADD(i) "v_pk_add_f32 " PP(i) ", " PP(i) ", %18\n"
#define PRIO(n) "s_setprio " #n "\n"
#define NOP "s_nop 0\n"

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

PATTERN(0, "fma x8", 8, 0, 0, X8(FMA))
PATTERN(1, "pk_fma x8", 0, 8, 0, X8(PKFMA))
PATTERN(2, "exp x8", 0, 0, 8, X8(EXP))
PATTERN(3, "fma, fma_sgpr alternating", 8, 0, 0, FMA(0) FMAS(1) FMA(2) FMAS(3) FMA(4) FMAS(5) FMA(6) FMAS(7))
PATTERN(4, "bitop3, fma_sgpr alternating", 8, 0, 0, BITOP(0) FMAS(1) BITOP(2) FMAS(3) BITOP(4) FMAS(5) BITOP(6) FMAS(7))
PATTERN(5, "fma, exp alternating", 4, 0, 4, FMA(0) EXP(1) FMA(2) EXP(3) FMA(4) EXP(5) FMA(6) EXP(7))
PATTERN(6, "bitop3, pk_fma alternating", 4, 4, 0, BITOP(0) PKFMA(1) BITOP(2) PKFMA(3) BITOP(4) PKFMA(5) BITOP(6) PKFMA(7))
PATTERN(7, "ashr, pk_fma alternating", 4, 4, 0, ASHR(0) PKFMA(1) ASHR(2) PKFMA(3) ASHR(4) PKFMA(5) ASHR(6) PKFMA(7))
PATTERN(8, "max, bitop3 alternating", 4, 4, 0, MAX(0) BITOP(1) MAX(2) BITOP(3) MAX(4) BITOP(5) MAX(6) BITOP(7))
PATTERN(9, "max, ashr alternating", 4, 4, 0, MAX(0) ASHR(1) MAX(2) ASHR(3) MAX(4) ASHR(5) MAX(6) ASHR(7))
PATTERN(10, "cvt x8", 0, 8, 0, X8(CVT))
PATTERN(11, "cvt, fma alternating", 4, 4, 0, CVT(0) FMA(1) CVT(2) FMA(3) CVT(4) FMA(5) CVT(6) FMA(7))
PATTERN(12, "mul VOP2 sgpr src0 x8", 8, 0, 0, X8(MULS2))
PATTERN(13, "mul inline const x8", 8, 0, 0, X8(MULI))
PATTERN(14, "add VOP2 literal x8", 8, 0, 0, X8(ADDL))
PATTERN(15, "mov vgpr x8", 8, 0, 0, X8(MOV))
PATTERN(16, "mov sgpr x8", 8, 0, 0, X8(MOVS))
PATTERN(17, "med3 0,1 x8", 0, 8, 0, X8(MED3))
PATTERN(18, "med3, fma alternating", 4, 4, 0, MED3(0) FMA(1) MED3(2) FMA(3) MED3(4) FMA(5) MED3(6) FMA(7))
PATTERN(19, "fma x8 with s_nop between", 8, 0, 0, FMA(0) NOP FMA(1) NOP FMA(2) NOP FMA(3) NOP FMA(4) NOP FMA(5) NOP FMA(6) NOP FMA(7) NOP)
PATTERN(20, "fma x96, exp x24 (as issue_mix P10)", 96, 0, 24,
        X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA)
        X8(EXP) X8(EXP) X8(EXP))
PATTERN(21, "fma x96 at prio 1, exp x24 at prio 0", 96, 0, 24,
        PRIO(1) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA)
        PRIO(0) X8(EXP) X8(EXP) X8(EXP))
PATTERN(22, "fma x96 at prio 0, exp x24 at prio 1", 96, 0, 24,
        PRIO(0) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA) X8(FMA)
        PRIO(1) X8(EXP) X8(EXP) X8(EXP))
PATTERN(23, "layer-like (issue_mix P21)", 72, 36, 24,
        X8(LOG) X4(LOG) X8(PKMUL) X4(PKMUL) X8(EXP) X4(EXP) X8(PKFMA) X8(PKFMA) X8(PKFMA)
        X8(FMA) X8(SUB) X8(ASHR) X8(BITOP) X8(MUL) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP))
PATTERN(24, "layer-like, fast runs at prio 1", 72, 36, 24,
        PRIO(0) X8(LOG) X4(LOG) X8(PKMUL) X4(PKMUL) X8(EXP) X4(EXP) X8(PKFMA) X8(PKFMA) X8(PKFMA)
        PRIO(1) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP) X8(MUL) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP))
PATTERN(25, "layer-like, fast runs at prio 3", 72, 36, 24,
        PRIO(0) X8(LOG) X4(LOG) X8(PKMUL) X4(PKMUL) X8(EXP) X4(EXP) X8(PKFMA) X8(PKFMA) X8(PKFMA)
        PRIO(3) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP) X8(MUL) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP))
PATTERN(26, "layer-like, unpairable runs at prio 1 (control)", 72, 36, 24,
        PRIO(1) X8(LOG) X4(LOG) X8(PKMUL) X4(PKMUL) X8(EXP) X4(EXP) X8(PKFMA) X8(PKFMA) X8(PKFMA)
        PRIO(0) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP) X8(MUL) X8(FMA) X8(SUB) X8(ASHR) X8(BITOP))
PATTERN(27, "fma x4, exp x1 with exp at prio 1", 4, 0, 1, PRIO(0) X4(FMA) PRIO(1) EXP(4))
PATTERN(28, "pk_add, fma alternating", 4, 4, 0, PKADD(0) FMA(1) PKADD(2) FMA(3) PKADD(4) FMA(5) PKADD(6) FMA(7))
PATTERN(29, "exp, bitop3 alternating", 4, 0, 4, EXP(0) BITOP(1) EXP(2) BITOP(3) EXP(4) BITOP(5) EXP(6) BITOP(7))
#define NPAT 30

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
  if (wpss.empty()) wpss = {8};
  for (int wps : wpss) {
    g_pure[0] = g_pure[1] = g_pure[2] = 0;
    RunAll<NPAT - 1>::go(d, st, wps);
    printf("\n");
  }
  return 0;
}
