// curl_math.h -- per-pixel arithmetic of the CURL colour-curve path for gfx950.
//
// One header, two compilations:
//   * hipcc (device): the kernels in curl_kernels.hip include it; transcendentals are the
//     CDNA4 hardware instructions v_log_f32 / v_exp_f32 / v_rcp_f32 (quarter-rate VALU),
//     clamps are v_med3_f32.
//   * g++ (host, -DCURL_HOST_TWIN): tests/ build a test-only twin of the same arithmetic with
//     libm standing in for the three hardware instructions, so the algebra (thresholds, tie
//     handling, constant folding, error budget) is checked against the oracle in the CPU
//     container.  The product never loads the twin: curl_amd/ binds libcurlhip.so only.
//
// Each function cites the reference lines it restates (paths under the reference tree).
// Constants are the float32 roundings of the Python doubles the reference writes, because
// torch casts a Python scalar to the tensor dtype before the op.
#pragma once
#include <math.h>
#include <type_traits>

#if defined(__HIPCC__)
#define CURL_HD __host__ __device__ __forceinline__
#else
#define CURL_HD inline
#endif

namespace curlm {

// ---------------------------------------------------------------- hardware primitives
CURL_HD float hw_log2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_logf(x);  // v_log_f32, 1 ulp, no denormal support (inputs here are >= 1e-4)
#else
  return log2f(x);
#endif
}
CURL_HD float hw_exp2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(x);  // v_exp_f32, 1 ulp (results here are >= 2^-14)
#else
  return exp2f(x);
#endif
}
// 1/x to (almost always) correct rounding: v_rcp_f32 (1 ulp) + one Newton step (2 FMAs).
CURL_HD float rcp_refined(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(x);
  float e = fmaf(-x, r, 1.0f);
  return fmaf(e, r, r);
#else
  return 1.0f / x;
#endif
}
CURL_HD float hw_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);  // v_rcp_f32, 1 ulp
#else
  return 1.0f / x;
#endif
}
CURL_HD float clampf(float x, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(x, lo, hi);  // v_med3_f32
#else
  return fminf(fmaxf(x, lo), hi);
#endif
}
CURL_HD float clamp01(float x) { return clampf(x, 0.0f, 1.0f); }

// to_tensor's byte -> float: b / 255 correctly rounded, for b = 0..255 held exactly in a float, without the
// ~10-instruction IEEE division: 1/255 as a float pair (hi + lo), b*lo rounded, then ONE fma b*hi + that -- equal to
// the division for all 256 bytes (test_twin_math checks them; the plain product b*fl(1/255) is wrong for 126 of them).
CURL_HD float u8_to_unit(float b) {
  const float hi = (float)(1.0 / 255.0);
  const float lo = (float)(1.0 / 255.0 - (double)hi);
  return fmaf(b, hi, b * lo);
}
// n / d for small non-negative integers held in floats (pixel column / width): a Newton-corrected multiply by
// rd = 1/d instead of the ~10-instruction IEEE division; equal to the division for every d <= 8192, n < d
// (test_twin_math.py checks all of them), at most 1 ulp off beyond.
CURL_HD float div_small(float n, float d, float rd) {
  float q = n * rd;
  float e = fmaf(-q, d, n);
  return fmaf(e, rd, q);
}
// (x * 255).astype('uint8') / to_pil_image's mul(255).byte(): truncation; out-of-range saturates
CURL_HD unsigned unit_to_u8(float x) {
  float v = x * 255.0f;
  v = fminf(fmaxf(v, 0.0f), 255.0f);  // fmaxf(NaN, 0) = 0: NaN becomes byte 0 on the device and in the host twin alike
  return (unsigned)(int)v;
}

// the same for a value KNOWN to lie in [0, 1] (the fused byte paths: both ops end in clamp(., 0, 1), and the white
// compositing x m + (1 - m) of two values in [0, 1] stays there): the two saturation instructions go
CURL_HD unsigned unit_to_u8_in_range(float x) { return (unsigned)(int)(x * 255.0f); }

// ---------------------------------------------------------------- branch-free selects
// Measured on MI355X (tools/ubench/valu_rate.hip): v_cndmask_b32 with its mask in VCC -- what hipcc emits
// for `c ? a : b` -- issues once per ~23 cycles per SIMD, against 2 for v_fma/v_mul/v_add/shift/and/or and
// ~3.5 for v_bfi/v_max/v_med3/v_cmp.  The fused kernels held ~17 such selects per pixel (36 % of their
// VALU time), so every select here is built from the sign bit instead: subtract, arithmetic shift, bitwise
// blend.  A bitwise blend also discards a NaN/garbage value in the branch not taken, which is what lets the
// `clamp(min=1e-4)` guards in front of the reference's pow() calls go (they only protect dead branches).
CURL_HD int f2i(float x) { return __builtin_bit_cast(int, x); }
CURL_HD float i2f(int x) { return __builtin_bit_cast(float, x); }
// Keeps the optimiser from folding `mask & a | ~mask & b` back into compare + v_cndmask.
CURL_HD int opaque(int m) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(m));
#endif
  return m;
}
CURL_HD int neg_mask(float x) { return opaque(f2i(x) >> 31); }           // all ones iff sign bit set
CURL_HD int zero_mask(float t) { return opaque((f2i(t) - 1) >> 31); }    // t >= +0: all ones iff t == +0
CURL_HD int nonzero_mask(float t) { return opaque((-f2i(t)) >> 31); }    // t >= +0: all ones iff t > 0
CURL_HD float blend(int mask, float a, float b) {  // mask ? a : b, bit by bit
#if defined(__HIP_DEVICE_COMPILE__)
  return i2f(__builtin_amdgcn_bitop3_b32(mask, f2i(a), f2i(b), 0xCA));  // v_bitop3_b32 (gfx950), one instruction
#else
  return i2f((mask & f2i(a)) | (~mask & f2i(b)));
#endif
}
CURL_HD float keep_if(int mask, float a) { return i2f(mask & f2i(a)); }                             // mask ? a : +0
CURL_HD float drop_if2(int m0, int m1, float a) {                                                   // (m0 | m1) ? +0 : a
#if defined(__HIP_DEVICE_COMPILE__)
  return i2f(__builtin_amdgcn_bitop3_b32(m0, m1, f2i(a), 0x02));  // ~m0 & ~m1 & a, one instruction
#else
  return i2f(~m0 & ~m1 & f2i(a));
#endif
}
CURL_HD float drop_if(int mask, float a) {                                                          // mask ? +0 : a
#if defined(__HIP_DEVICE_COMPILE__)
  return i2f(__builtin_amdgcn_bitop3_b32(mask, f2i(a), f2i(a), 0x0C));  // ~mask & a, one instruction
#else
  return i2f(~mask & f2i(a));
#endif
}
// x <= thr ? a : b   (thr - x is exact-or-nonzero for x != thr: float32 denormals are on, hipcc default)
CURL_HD float select_le(float x, float thr, float a, float b) { return blend(neg_mask(thr - x), b, a); }

// ---------------------------------------------------------------- constants
// colors.py:37-38 / 118-119
constexpr float kSrgbThr = (float)0.04045;
constexpr float kInv1292 = (float)(1.0 / 12.92);
constexpr float kInv1055 = (float)(1.0 / 1.055);
constexpr float kLinThr = (float)0.0031308;
// x ** 2.4 : torch raises to float32(2.4) = 2.4000000953...; 2.4f - 2.0f is exact in float32.
constexpr float kGammaFrac = (float)2.4 - 2.0f;
constexpr float kGamma = (float)2.4;  // torch raises to float32(2.4)
constexpr float kInvGamma = (float)(1.0 / 2.4);
constexpr float kThird = (float)(1.0 / 3.0);
// colors.py:43-47 / 108-111
constexpr double kEpsD = 6.0 / 29.0;
constexpr float kEps = (float)kEpsD;
constexpr float kEps3 = (float)(kEpsD * kEpsD * kEpsD);
constexpr float k3Eps2 = (float)(3.0 * kEpsD * kEpsD);
constexpr float kInv3Eps2 = (float)(1.0 / (3.0 * kEpsD * kEpsD));
constexpr float k4_29 = (float)(4.0 / 29.0);
// (the reference's clamp(min=0.0001) in front of every pow guards only the branch its select discards: dropped)
// colors.py:24,41 : XYZ is multiplied by the float32 reciprocals of the D65 white point
constexpr float kInvXn = 1.0f / 0.950456f;
constexpr float kInvZn = 1.0f / 1.088754f;
constexpr float kXn = 0.950456f;
constexpr float kZn = 1.088754f;
// colors.py:205,240
constexpr float kHsvFloor = (float)1e-9;

struct Px {
  float c0, c1, c2;
};

// ---------------------------------------------------------------- N pixels at a time, transcendentals clustered
// Measured (tools/ubench/transmix.hip): a v_exp/v_log/v_rcp issued between FMA-class instructions costs
// ~5.8 ns per SIMD, the same instruction in a run of its own kind ~4.3 ns (alone 3.5).  The converters are
// therefore written over the N pixels a lane owns, as phases: every phase of transcendentals is a run of
// 3N (or N) back-to-back v_log / v_exp / v_rcp, fenced so the scheduler cannot interleave it again.
#if defined(__HIP_DEVICE_COMPILE__)
#define CURL_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define CURL_FENCE() ((void)0)
#endif

template <int N>
struct PxN {
  float c0[N], c1[N], c2[N];
};

// Issue priority by phase.  gfx950 issues two VALU instructions of DIFFERENT waves in one
// quad-cycle (SQ_ACTIVE_INST_VALU2) when both are plain one-pass instructions (fma/mul/add/sub, shifts, bit ops,
// moves; at most one of the two with an SGPR operand); packed-FP32 and transcendental instructions always go alone.
// With every wave at the same priority the pickers are fed a random mix of heads and almost nothing pairs (2.5 % of
// the layer kernel's instructions, tools/ubench/issue_pair.hip reproduces it).  Raising the priority of a wave while
// it runs its packed / transcendental runs makes those runs drain back to back, which leaves the other waves of the
// SIMD in plain code at the same time -- where they pair.
// Measured on the fused layer (profiles/r02/issue_priority_ab.log, bs32 x 1500x1000): all at one priority 2.5 % of
// the instructions pair and the kernel takes 254 us; transcendental (and packed) runs at priority 1: 26 % pair,
// 217 us; the same with the element-wise helpers as scalar instead of packed code (they can pair, packed ones
// never do): 40 % pair, 213 us -- what is built here: plain code at priority 0, transcendental runs at 1.  (The packed
// helpers, other priority levels and the polynomial model's Horner code at a raised priority were measured and are not
// built: tools/experiments/patches/.)
#if defined(__HIP_DEVICE_COMPILE__)
#define CURL_SETPRIO(n)                  \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_setprio(n);       \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
#else
#define CURL_SETPRIO(n) ((void)0)
#endif
#define CURL_TRANS_BEGIN() CURL_SETPRIO(1)
#define CURL_TRANS_END() CURL_SETPRIO(0)

// Two-float vectors for the polynomial model's packed Horner chains (curl_math_poly.h: an FMA-only stream pairs only 72 %
// of its three-VGPR-operand instructions, so v_pk_fma_f32 wins there).  The converters' element-wise loops below are SCALAR:
// two scalar instructions of different waves share a quad-cycle, which a packed instruction (alone in its quad) only equals
// (measured: layer 217 us packed, 213 us scalar).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float curl_f2 __attribute__((ext_vector_type(2)));
CURL_HD curl_f2 splat2(float k) {
  curl_f2 v;
  v.x = k;
  v.y = k;
  return v;
}
CURL_HD curl_f2 ld2(const float* a, int i) {
  curl_f2 v;
  v.x = a[i];
  v.y = a[i + 1];
  return v;
}
CURL_HD void st2(float* a, int i, curl_f2 v) {
  a[i] = v.x;
  a[i + 1] = v.y;
}
#endif

// y = a*k + c, element-wise over M values (k, c scalars)
template <int M>
CURL_HD void fma_run(float (&y)[M], const float (&a)[M], float k, float c) {
  for (int i = 0; i < M; ++i) y[i] = fmaf(a[i], k, c);
}
// y = a*b element-wise
template <int M>
CURL_HD void mul_run(float (&y)[M], const float (&a)[M], const float (&b)[M]) {
  for (int i = 0; i < M; ++i) y[i] = a[i] * b[i];
}
// y = a*k
template <int M>
CURL_HD void scale_run(float (&y)[M], const float (&a)[M], float k) {
  for (int i = 0; i < M; ++i) y[i] = a[i] * k;
}
// y = k - a   (the sign of thr - x drives every threshold select)
template <int M>
CURL_HD void rsub_run(float (&y)[M], float k, const float (&a)[M]) {
  for (int i = 0; i < M; ++i) y[i] = k - a[i];
}
// The threshold selects of the converters (x <= thr ? a : b, twelve per pixel) on the device: v_cmp_le_f32_e64 into an
// SGPR pair + v_cndmask_b32_e64 (the VOP3 form with its mask in SGPRs issues in 4 cycles; it is the VOP2 form with the
// mask in VCC, what hipcc emits for `?:`, that takes 23).  Two instructions with two VGPR reads each instead of
// sub / ashr / bitop3: 750 instead of 798 instructions per wave and ~0.7 nJ less per select (tools/ubench/energy.hip),
// which is what counts under the board's power cap: layer 239.7 -> 236.8 us, Lab stage 199.1 -> 197.3 us on one box
// (profiles/r02/select_cndmask_ab.log).  The host twin uses the sign-bit form: same values, also for x == thr (a NaN
// takes the second branch here, as the reference's `x <= thr` mask does).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float select_le_hw(float x, float thr, float a, float b) {
  unsigned long long m;
  float r;
  asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x), "s"(thr));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
// (c < mx) ? +0 : val -- the "[c == max]" factor of the hue terms (c <= mx always holds there)
__device__ __forceinline__ float zero_if_less_hw(float c, float mx, float val) {
  unsigned long long m;
  float r;
  asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(c), "v"(mx));
  asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(r) : "v"(val), "s"(m));
  return r;
}
#endif

// out = (x <= thr) ? a : b  element-wise (thr scalar)
template <int M>
CURL_HD void select_le_run(float (&out)[M], const float (&x)[M], float thr, const float (&a)[M], const float (&b)[M]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int i = 0; i < M; ++i) out[i] = select_le_hw(x[i], thr, a[i], b[i]);
  return;
#endif
  float d[M];
  rsub_run(d, thr, x);
#pragma unroll
  for (int i = 0; i < M; ++i) out[i] = blend(neg_mask(d[i]), b[i], a[i]);
}

// The same select with a RARE first branch: out = (x <= thr) ? a(i) : b, where a() is only evaluated -- and the select only
// executed -- if some lane of the wavefront has some x <= thr.  The compares are needed either way; their 64-lane results
// sit in SGPR pairs, OR-ed by the scalar unit, and one wave-uniform branch skips M x (branch value + v_cndmask) VALU
// instructions.  The four threshold selects of the Lab converters pick their linear branch for dark values only (sRGB
// x <= 0.04045, XYZ t <= 0.008856, f <= 6/29, linear v <= 0.0031308): a wave of a photograph's mid-tones takes none of them;
// of uniformly random pixels (the benchmark) ~30 % of the waves skip the two XYZ-side ones.  The kernel runs at the board's
// power cap, where instructions are joules.
#if defined(__HIP_DEVICE_COMPILE__)
// ... and inside the branch the linear value is not computed for every lane and then selected: the multiply (or fma) itself
// runs PREDICATED -- exec narrowed to the lanes whose compare said "linear branch" -- and overwrites the pow branch's value
// in place there: one VALU instruction and two scalar ones per value instead of two VALU instructions (the scalar unit is
// a pipe of its own).  s_and_saveexec / restore, so a caller's partial exec mask is respected.
__device__ __forceinline__ void pred_mul(float& dst, unsigned long long m, float x, float k_uniform) {  // dst = x * k where m
  unsigned long long t;
  asm("s_and_saveexec_b64 %1, %2\n\tv_mul_f32_e32 %0, %4, %3\n\ts_mov_b64 exec, %1"
      : "+v"(dst), "=&s"(t) : "s"(m), "v"(x), "s"(k_uniform) : "scc");  // s_and_saveexec writes SCC
}
__device__ __forceinline__ void pred_mul_clamp(float& dst, unsigned long long m, float x, float k_uniform) {  // clamp01(x * k) where m
  unsigned long long t;
  asm("s_and_saveexec_b64 %1, %2\n\tv_mul_f32_e64 %0, %4, %3 clamp\n\ts_mov_b64 exec, %1"
      : "+v"(dst), "=&s"(t) : "s"(m), "v"(x), "s"(k_uniform) : "scc");  // s_and_saveexec writes SCC
}
__device__ __forceinline__ void pred_fma(float& dst, unsigned long long m, float x, float k_uniform, float c_vgpr) {  // x * k + c where m
  unsigned long long t;
  asm("s_and_saveexec_b64 %1, %2\n\tv_fma_f32 %0, %3, %4, %5\n\ts_mov_b64 exec, %1"
      : "+v"(dst), "=&s"(t) : "s"(m), "v"(x), "s"(k_uniform), "v"(c_vgpr) : "scc");
}
// A(r, m): overwrite r[i] with the linear branch where m[i] (predicated form), or A(av): compute it for every lane.
// BRANCH = false: no wave-uniform skip, only the predicated overwrites (the predicates are consumed one by one instead of
// being held across a branch: for kernels at their register budget)
template <bool BRANCH = true, int M, class A>
__device__ __forceinline__ void select_le_lazy(float (&out)[M], const float (&x)[M], float thr, A a, const float (&b)[M]) {
  unsigned long long m[M], any = BRANCH ? 0ull : 1ull;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m[i]) : "v"(x[i]), "s"(thr));
    if (BRANCH) any |= m[i];
  }
  // the select works IN PLACE on b's registers, so the skipping path is empty (as "out = m ? a : b" on one side and
  // "out = b" on the other the register allocator left twelve copies on the skipping side)
  float r[M];
#pragma unroll
  for (int i = 0; i < M; ++i) r[i] = b[i];
  if (any) {
    if constexpr (std::is_invocable_v<A, float (&)[M], const unsigned long long (&)[M]>) {
      a(r, m);  // predicated overwrites
    } else {
      float av[M];
      a(av);
#pragma unroll
      for (int i = 0; i < M; ++i) asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(av[i]), "s"(m[i]));
    }
  }
#pragma unroll
  for (int i = 0; i < M; ++i) out[i] = r[i];
}
#endif

// y[i] = x[i]^e for M values: M logs, M multiplies, M exps
template <int M>
CURL_HD void pow_run(float (&x)[M], float e) {
  CURL_FENCE();
  CURL_TRANS_BEGIN();
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] = hw_log2(x[i]);
  CURL_FENCE();
  scale_run(x, x, e);
  CURL_TRANS_BEGIN();
  CURL_FENCE();
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] = hw_exp2(x[i]);
  CURL_TRANS_END();
  CURL_FENCE();
}

// row r of a 3x3 matrix applied to plane-major (c*N + i) data: y[i] = m0*a[i] + m1*a[N+i] + m2*a[2N+i]
template <int N>
CURL_HD void mat_row(float (&y)[N], const float (&a)[3 * N], float m0, float m1, float m2) {
  for (int i = 0; i < N; ++i) y[i] = fmaf(m2, a[2 * N + i], fmaf(m1, a[N + i], m0 * a[i]));
}

// ---------------------------------------------------------------- RGB -> Lab   colors.py:27-62
// LAZY: the two threshold selects through select_le_lazy (the fused curve stages ask for it; the polynomial model's
// kernel, already at its register budget, spills 1.1 KB per lane with the twelve predicates live across a branch: 0.9 -> 6 ms)
template <int N, int LAZY = 0>  // 0: eager selects, 1: skipped per wave + predicated, 2: predicated only
CURL_HD void rgb2lab_n(PxN<N>& p) {
  float x[3 * N], g[3 * N], u2[3 * N], lin[3 * N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    x[i] = p.c0[i];
    x[N + i] = p.c1[i];
    x[2 * N + i] = p.c2[i];
  }
  // colors.py:37-38: both branches are evaluated and blended with 0/1 masks in the reference.  The gamma
  // branch is taken only for x > 0.04045, where clamp(x, min=1e-4) is the identity, so the guard is dropped.
  // u^2.4 = u*u * 2^(0.4*log2 u): splitting off u^2 keeps the exponent argument below 2 in magnitude, so the
  // hardware log/exp errors (1 ulp each) cost ~1e-7 relative instead of ~4e-7 (the direct form fails the
  // 1e-5 end-to-end bar on out-of-range inputs).  torch raises to float32(2.4); 2.4f - 2.0f is exact.
  fma_run(g, x, kInv1055, (float)(0.055 / 1.055));
  // (2^(2.4*log2 u) taken directly saves two VALU instructions per value and 1 % of the layer's time, and moves one
  // out-of-range pixel of test_odd_shapes_and_tails over the 1e-5 bar: not built, profiles/r03/exp26_*.log.)
  {
    mul_run(u2, g, g);
    pow_run(g, kGammaFrac);
    mul_run(g, u2, g);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (LAZY != 0) {
    const float (&xr)[3 * N] = x;
    select_le_lazy<LAZY == 1>(x, x, kSrgbThr, [&](float (&r)[3 * N], const unsigned long long (&m)[3 * N]) {
#pragma unroll
      for (int i = 0; i < 3 * N; ++i) pred_mul(r[i], m[i], xr[i], kInv1292);
    }, g);
  } else
#endif
  {
    scale_run(lin, x, kInv1292);
    select_le_run(x, x, kSrgbThr, lin, g);
  }
  // colors.py:10-12,40 (OpenCV matrix) then colors.py:41 (x 1/white, folded into the rows)
  float t[3 * N], f[3 * N];
  {
    float tx[N], ty[N], tz[N];
    mat_row<N>(tx, x, 0.412453f * kInvXn, 0.357580f * kInvXn, 0.180423f * kInvXn);
    mat_row<N>(ty, x, 0.212671f, 0.715160f, 0.072169f);
    mat_row<N>(tz, x, 0.019334f * kInvZn, 0.119193f * kInvZn, 0.950227f * kInvZn);
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = tx[i], t[N + i] = ty[i], t[2 * N + i] = tz[i];
  }
  // colors.py:45-47 (cube root taken only for t > eps^3 > 1e-4)
#pragma unroll
  for (int i = 0; i < 3 * N; ++i) f[i] = t[i];
  pow_run(f, kThird);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (LAZY != 0) {
    const float (&tr)[3 * N] = t;
    select_le_lazy<LAZY == 1>(f, t, kEps3, [&](float (&r)[3 * N], const unsigned long long (&m)[3 * N]) {
      const float c = k4_29;
#pragma unroll
      for (int i = 0; i < 3 * N; ++i) pred_fma(r[i], m[i], tr[i], kInv3Eps2, c);
    }, f);
  } else
#endif
  {
    fma_run(lin, t, kInv3Eps2, k4_29);
    select_le_run(f, t, kEps3, lin, f);
  }
  // colors.py:18-20,50: L = 116 fy - 16, a = 500 (fx - fy), b = 200 (fy - fz);
  // colors.py:57-59: L/100, (a/110 + 1)/2, (b/110 + 1)/2 -- constants folded.
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float fx = f[i], fy = f[N + i], fz = f[2 * N + i];
    p.c0[i] = fmaf(fy, 1.16f, -0.16f);
    p.c1[i] = fmaf(fx - fy, (float)(500.0 / 220.0), 0.5f);
    p.c2[i] = fmaf(fy - fz, (float)(200.0 / 220.0), 0.5f);
  }
}

// ---------------------------------------------------------------- Lab -> RGB   colors.py:88-123
template <int N, bool CLAMP12 = false, int LAZY = 0>  // CLAMP12: also clamp channels 1 and 2 to [0,1] (fused stages, see below); LAZY as in rgb2lab_n
CURL_HD void lab2rgb_n(PxN<N>& p) {
  float X[3 * N], v[3 * N], g[3 * N], lin[3 * N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    // colors.py:97-99 (L*100, (a*2-1)*110, (b*2-1)*110) and colors.py:79-81,104-106
    // (fy = (L+16)/116, fx = fy + a/500, fz = fy - b/200) with the constants folded:
    float fy = fmaf(p.c0[i], (float)(100.0 / 116.0), (float)(16.0 / 116.0));
    float fx = fmaf(p.c1[i], (float)(220.0 / 500.0), fy - (float)(110.0 / 500.0));
    float fz = fmaf(p.c2[i], (float)(-220.0 / 200.0), fy + (float)(110.0 / 200.0));
    X[i] = fx, X[N + i] = fy, X[2 * N + i] = fz;
  }
  {
    // colors.py:110-111 ; x**3.0 is x*x*x in torch (cube taken only for f > eps > 1e-4)
    float cub[3 * N];
    mul_run(cub, X, X);
    mul_run(cub, cub, X);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LAZY != 0) {
      const float (&Xr)[3 * N] = X;
      select_le_lazy<LAZY == 1>(X, X, kEps, [&](float (&r)[3 * N], const unsigned long long (&m)[3 * N]) {
        const float c = -(k3Eps2 * k4_29);
#pragma unroll
        for (int i = 0; i < 3 * N; ++i) pred_fma(r[i], m[i], Xr[i], k3Eps2, c);
      }, cub);
    } else
#endif
    {
      fma_run(lin, X, k3Eps2, -(k3Eps2 * k4_29));
      select_le_run(X, X, kEps, lin, cub);
    }
  }
  {
    // colors.py:114 (x white) folded into the columns of colors.py:71-73,117 (Lindbloom sRGB D65 inverse)
    float r[N], gg[N], b[N];
    mat_row<N>(r, X, 3.2404542f * kXn, -1.5371385f, -0.4985314f * kZn);
    mat_row<N>(gg, X, -0.9692660f * kXn, 1.8760108f, 0.0415560f * kZn);
    mat_row<N>(b, X, 0.0556434f * kXn, -0.2040259f, 1.0572252f * kZn);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = r[i], v[N + i] = gg[i], v[2 * N + i] = b[i];
  }
  // colors.py:118-119 (power taken only for v > 0.0031308 > 1e-4)
#pragma unroll
  for (int i = 0; i < 3 * N; ++i) g[i] = v[i];
  pow_run(g, kInvGamma);
  if (!CLAMP12) {
    fma_run(g, g, 1.055f, -0.055f);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LAZY != 0) {
      const float (&vr)[3 * N] = v;
      select_le_lazy<LAZY == 1>(v, v, kLinThr, [&](float (&r)[3 * N], const unsigned long long (&m)[3 * N]) {
#pragma unroll
        for (int i = 0; i < 3 * N; ++i) pred_mul(r[i], m[i], vr[i], 12.92f);
      }, g);
    } else
#endif
    {
      scale_run(lin, v, 12.92f);
      select_le_run(v, v, kLinThr, lin, g);
    }
  } else {
    // The next consumer (adjust3, curves.py:36) clamps channels 1 and 2 before using them.  A clamp after the
    // bitwise select is a separate v_max; on the two branches it rides on the fma / mul that produce them
    // (clamp(select(a, b)) == select(clamp a, clamp b)), so those two channels go through scalar VOP3 forms.
    float g0[N], v0[N], l0[N], d[3 * N];
#pragma unroll
    for (int i = 0; i < N; ++i) g0[i] = g[i], v0[i] = v[i];
    fma_run(g0, g0, 1.055f, -0.055f);
    scale_run(l0, v0, 12.92f);
    rsub_run(d, kLinThr, v);
#pragma unroll
    for (int i = 0; i < N; ++i) g[i] = g0[i], lin[i] = l0[i];
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LAZY != 0) {
#pragma unroll
      for (int i = N; i < 3 * N; ++i) g[i] = clamp01(fmaf(g[i], 1.055f, -0.055f));
      const float (&vr)[3 * N] = v;
      select_le_lazy<LAZY == 1>(v, v, kLinThr, [&](float (&r)[3 * N], const unsigned long long (&m)[3 * N]) {
#pragma unroll
        for (int i = 0; i < N; ++i) pred_mul(r[i], m[i], vr[i], 12.92f);
#pragma unroll
        for (int i = N; i < 3 * N; ++i) pred_mul_clamp(r[i], m[i], vr[i], 12.92f);
      }, g);
    } else
#endif
    {
#pragma unroll
      for (int i = N; i < 3 * N; ++i) {
        g[i] = clamp01(fmaf(g[i], 1.055f, -0.055f));
        lin[i] = clamp01(v[i] * 12.92f);
      }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int i = 0; i < 3 * N; ++i) v[i] = select_le_hw(v[i], kLinThr, lin[i], g[i]);
#else
#pragma unroll
      for (int i = 0; i < 3 * N; ++i) v[i] = blend(neg_mask(d[i]), g[i], lin[i]);
#endif
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    p.c0[i] = v[i];  // NOT clamped (colors.py:121-123)
    p.c1[i] = v[N + i];
    p.c2[i] = v[2 * N + i];
  }
}

// ---------------------------------------------------------------- RGB -> HSV   colors.py:195-242
// UNIT = the caller guarantees r, g, b in [0,1] (the curve layer: adjust3 has just clamped them) and consumes
// h, s, v through adjust_hsv4, which clamps what it produces.  Then the 1e-9 floors of colors.py:205,240 are
// six v_med3 per pixel that change the layer's output by < 1e-8 (they only move channels that are exactly 0 to
// 1e-9: relative hue/saturation changes of 1e-9/max, scaled by v = max on the way back to RGB), and the
// reciprocal needs no Newton step.
template <int N, bool UNIT = false>
CURL_HD void rgb2hsv_n(PxN<N>& p) {
  float r[N], g[N], b[N], mx[N], nd[N], rdm[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    r[i] = UNIT ? p.c0[i] : clampf(p.c0[i], kHsvFloor, 1.0f);
    g[i] = UNIT ? p.c1[i] : clampf(p.c1[i], kHsvFloor, 1.0f);
    b[i] = UNIT ? p.c2[i] : clampf(p.c2[i], kHsvFloor, 1.0f);
    mx[i] = fmaxf(r[i], fmaxf(g[i], b[i]));
    // nd = -(max - min): +0 when all channels are equal, negative otherwise -- its sign bit IS the "df != 0" mask.
    nd[i] = fminf(r[i], fminf(g[i], b[i])) - mx[i];
    // 1/df and 1/mx from ONE reciprocal: q = 1/(df*mx), 1/df = q*mx, 1/mx = q*df (df*mx >= 1e-25, no underflow).
    // df == 0 gives inf/NaN; hue and saturation are masked to 0 below.
    // UNIT: + 1e-30 keeps the reciprocal finite when df == 0, so that s = df * df * q is an exact 0 there without a mask
    // (one fma instead of mul + and); df * mx >= 6e-8 * mx for any two distinct floats <= mx, so the addend changes q
    // only where mx < 1e-20 -- and v = mx scales everything the HSV stage contributes to the output.
    rdm[i] = UNIT ? fmaf(-nd[i], mx[i], 1e-30f) : -nd[i] * mx[i];
  }
  CURL_FENCE();
  CURL_TRANS_BEGIN();
#pragma unroll
  for (int i = 0; i < N; ++i) rdm[i] = UNIT ? hw_rcp(rdm[i]) : rcp_refined(rdm[i]);
  CURL_TRANS_END();
  CURL_FENCE();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float dfi = rdm[i] * mx[i];
    // colors.py:221-224: the three sextant terms ADD when channels tie for the maximum.
    // [c == mx] as a bit mask: c - mx is negative exactly when c is NOT the maximum (+0 when it is), so one
    // arithmetic shift gives the complement mask and one and-not applies it.
#if defined(__HIP_DEVICE_COMPILE__)
    float t0 = zero_if_less_hw(r[i], mx[i], (g[i] - b[i]) * dfi);
    float t1 = zero_if_less_hw(g[i], mx[i], fmaf(b[i] - r[i], dfi, 2.0f));
    float t2 = zero_if_less_hw(b[i], mx[i], fmaf(r[i] - g[i], dfi, 4.0f));
#else
    float t0 = drop_if(neg_mask(r[i] - mx[i]), (g[i] - b[i]) * dfi);
    float t1 = drop_if(neg_mask(g[i] - mx[i]), fmaf(b[i] - r[i], dfi, 2.0f));
    float t2 = drop_if(neg_mask(b[i] - mx[i]), fmaf(r[i] - g[i], dfi, 4.0f));
#endif
    int live = neg_mask(nd[i]);
    float h = keep_if(live, (t0 + t1) + t2);  // df == 0 -> 0 (colors.py:221)
    // colors.py:225-231: *60, negative hues + 360, /360  ==  (negative sextants + 6) / 6
    // (not fract(h/6): two channels tying for the maximum at g == b add up to h6 = 6 exactly, which must stay
    // hue 1.0, not wrap to 0 -- the hue curves of adjust_hsv are not periodic)
    if (UNIT) h = fmaf(h, (float)(1.0 / 6.0), keep_if(neg_mask(h), 1.0f));
    else h = (h + keep_if(neg_mask(h), 6.0f)) * (float)(1.0 / 6.0);
    float s = nd[i] * (rdm[i] * nd[i]);  // colors.py:234-237: df/mx
    if (!UNIT) s = keep_if(live, s);     // 0 when df == 0 (UNIT: nd == 0 and a finite q give exactly 0)
    p.c0[i] = UNIT ? h : clampf(h, kHsvFloor, 1.0f);  // colors.py:240
    p.c1[i] = UNIT ? s : clampf(s, kHsvFloor, 1.0f);
    p.c2[i] = UNIT ? mx[i] : clampf(mx[i], kHsvFloor, 1.0f);
  }
}

// single-pixel forms (chain kernel, backward recompute, host twin, tests)
CURL_HD Px rgb2lab(Px p) {
  PxN<1> q{{p.c0}, {p.c1}, {p.c2}};
  rgb2lab_n<1>(q);
  return Px{q.c0[0], q.c1[0], q.c2[0]};
}
CURL_HD Px lab2rgb(Px p) {
  PxN<1> q{{p.c0}, {p.c1}, {p.c2}};
  lab2rgb_n<1>(q);
  return Px{q.c0[0], q.c1[0], q.c2[0]};
}
CURL_HD Px rgb2hsv(Px p) {
  PxN<1> q{{p.c0}, {p.c1}, {p.c2}};
  rgb2hsv_n<1>(q);
  return Px{q.c0[0], q.c1[0], q.c2[0]};
}
// powers used by the backward pass
CURL_HD float pow_gamma(float x) { return (x * x) * hw_exp2(kGammaFrac * hw_log2(x)); }
CURL_HD float pow_inv_gamma(float x) { return hw_exp2(kInvGamma * hw_log2(x)); }
CURL_HD float cbrt_pos(float x) { return hw_exp2(kThird * hw_log2(x)); }

// ---------------------------------------------------------------- HSV -> RGB   colors.py:131-177
template <bool UNIT = false>  // UNIT: h, s, v already in [0,1] (straight out of adjust_hsv4's clamps)
CURL_HD Px hsv2rgb(Px p) {
  // colors.py:141-175 in sextant units: clamp(360h - a, 0, 60) * (d/60) == clamp(6h - a/60, 0, 1) * d,
  // so every ramp is one saturating add (v_add_f32 ... clamp) and the /60 disappears.
  // Each channel's two ramps are one trapezoid: for h6 = 6h in [0,6]
  //   clamp(h6-1) - clamp(h6-4) = clamp(2 - |h6-3|),  clamp(h6) - clamp(h6-3) = clamp(2 - |h6-2|),
  //   clamp(h6-2) - clamp(h6-5) = clamp(2 - |h6-4|)   (rise, plateau at 1, fall),
  // so a channel is fma(h, 6, -c), 2 - |.| saturated (one VOP3 with abs and clamp), one fma: 3 instructions instead of 4,
  // and d = v s, q = v - d replace 1-s, v(1-s), v-q: 11 instructions per pixel instead of 15 (the kernel runs at the
  // board's power cap: instructions are joules).  Same piecewise-linear function, roundings of the same size.
  const float hh = UNIT ? p.c0 : clamp01(p.c0), s = UNIT ? p.c1 : clamp01(p.c1), v = UNIT ? p.c2 : clamp01(p.c2);
  const float d = v * s;   // v - p, p = v (1 - s): colors.py:142
  const float q = v - d;
  const float tr = clamp01(2.0f - fabsf(fmaf(hh, 6.0f, -3.0f)));
  const float tg = clamp01(2.0f - fabsf(fmaf(hh, 6.0f, -2.0f)));
  const float tb = clamp01(2.0f - fabsf(fmaf(hh, 6.0f, -4.0f)));
  float r = fmaf(tr, -d, v);   // colors.py:144-150
  float g = fmaf(tg, d, q);    // colors.py:153-159
  float b = fmaf(tb, d, q);    // colors.py:163-168
  Px o;
  o.c0 = clamp01(r);
  o.c1 = clamp01(g);
  o.c2 = clamp01(b);
  return o;
}

template <int N>
CURL_HD void hsv2rgb_n(PxN<N>& p) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Px o = hsv2rgb(Px{p.c0[i], p.c1[i], p.c2[i]});
    p.c0[i] = o.c0, p.c1[i] = o.c1, p.c2[i] = o.c2;
  }
}

// ---------------------------------------------------------------- curves   curves.py:4-38
// Collapsed form of curves.py:31-32: with no clamp on (S*x - j) the sum is exactly
// a + b*x,  a = C0 - sum_j j*slope_j,  b = S*sum_j slope_j  (j = 0..K-3).  (a,b) come from the
// knot-prep kernel (float64 sums of the float32 slopes, rounded once).
struct Affine {
  float a, b;
};
CURL_HD float curve_mul(float x_out, float x_in, Affine k) { return x_out * fmaf(k.b, x_in, k.a); }

// adjust_rgb / adjust_lab (curves.py:90-133,136-180): curves (0->0),(1->1),(2->2); EVERY apply_curve
// clamps all three channels (curves.py:36), so channel 0 meets its curve unclamped while channels
// 1 and 2 are clamped first.
template <bool CLAMPED12 = false>  // CLAMPED12: channels 1 and 2 arrive clamped (lab2rgb_n<N, true>)
CURL_HD Px adjust3(Px p, Affine k0, Affine k1, Affine k2) {
  Px o;
  o.c0 = clamp01(curve_mul(p.c0, p.c0, k0));
  float c1 = CLAMPED12 ? p.c1 : clamp01(p.c1), c2 = CLAMPED12 ? p.c2 : clamp01(p.c2);
  o.c1 = clamp01(curve_mul(c1, c1, k1));
  o.c2 = clamp01(curve_mul(c2, c2, k2));
  return o;
}
// adjust_hsv (curves.py:41-87): H->H, H->S (on the ADJUSTED hue), S->S, V->V.
template <bool UNIT = false>  // UNIT: s and v arrive in [0,1] (rgb2hsv_n<N, true>), their input clamp is the identity
CURL_HD Px adjust_hsv4(Px p, Affine k0, Affine k1, Affine k2, Affine k3) {
  float h = clamp01(curve_mul(p.c0, p.c0, k0));
  float s = UNIT ? p.c1 : clamp01(p.c1), v = UNIT ? p.c2 : clamp01(p.c2);
  s = clamp01(curve_mul(s, h, k1));
  s = clamp01(curve_mul(s, s, k2));
  v = clamp01(curve_mul(v, v, k3));
  Px o{h, s, v};
  return o;
}

// Per-curve host/prep arithmetic: slopes in float32 as the reference forms them (curves.py:19), every
// sum in float64, one rounding at the end.  reg = sum of squared slope differences (curves.py:24).
// The sums are taken in ONE fixed order everywhere -- sixteen interleaved partial sums (partial l takes the terms j = l,
// l + 16, ...), combined pairwise (l with l ^ 8, ^ 4, ^ 2, ^ 1) -- because that is the order sixteen lanes of a wavefront
// produce (knots_prep_kernel and the in-kernel collapse give a curve 16 lanes: the K-long chain of dependent float64
// operations one thread ran was a third of the ~3.4 us prologue a one-frame forward paid, round 5); collapse_curve is the
// same arithmetic by one thread (the chain kernels, the host twin): identical bits.
constexpr int kCollapseLanes = 16;
struct CollapseSums {
  double s, js, r;  // sum slope_j, sum j * slope_j (j <= K-3: curves.py:31 slope[:, :-1]); sum (slope_j - slope_{j-1})^2
};
CURL_HD CollapseSums collapse_partial(const float* C, int K, int l) {
  CollapseSums p{0.0, 0.0, 0.0};
  for (int j = l; j + 1 < K; j += kCollapseLanes) {
    const float cj = C[j], sl = C[j + 1] - cj;
    if (j + 2 < K) {
      p.s += (double)sl;
      p.js += (double)j * (double)sl;
    }
    if (j > 0) {
      const float d = sl - (cj - C[j - 1]);
      p.r += (double)(d * d);
    }
  }
  return p;
}
CURL_HD void collapse_finish(const float* C, int K, const CollapseSums& t, float& a, float& b, float& reg) {
  a = (float)((double)C[0] - t.js);
  b = (float)((double)(K - 1) * t.s);
  reg = (float)t.r;
}
// what lane l holds after the steps that trade with lane l ^ 8, ..., l ^ O (O = 16: the lane's own partial sums): the value
// of S(l, O) = S(l, 2 O) + S(l ^ O, 2 O), by recursion instead of sixteen live partials (five are alive at a time)
template <int O>
CURL_HD CollapseSums collapse_tree(const float* C, int K, int l) {
  if constexpr (O == kCollapseLanes) {
    return collapse_partial(C, K, l);
  } else {
    const CollapseSums x = collapse_tree<2 * O>(C, K, l), y = collapse_tree<2 * O>(C, K, l ^ O);
    return CollapseSums{x.s + y.s, x.js + y.js, x.r + y.r};
  }
}
CURL_HD void collapse_curve(const float* C, int K, float& a, float& b, float& reg) {
  collapse_finish(C, K, collapse_tree<1>(C, K, 0), a, b, reg);
}

// curves.py:31-32 in torch's evaluation order: t_j = slope_j * (S*x - j) rounded term by term, summed
// by ATen's cascade sum (SumKernel.cpp multi_row_sum: sequential into acc0, flushed into acc1 every 16
// terms and into acc2 every 256; acc0 += acc1; acc0 += acc2 at the end), then C0 + sum.  No FMA.
CURL_HD float scale_exact(float x, const float* sl, float c0, int n_terms, float S) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  float sx = S * x;
  float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
  int j = 0;
  for (; j + 16 <= n_terms;) {
    for (int e = 0; e < 16; ++e, ++j) {
      float t = sx - (float)j;
      float p = sl[j] * t;
      acc0 = acc0 + p;
    }
    acc1 = acc1 + acc0;
    acc0 = 0.0f;
    if ((j & 0xF0) == 0) {
      acc2 = acc2 + acc1;
      acc1 = 0.0f;
    }
  }
  for (; j < n_terms; ++j) {
    float t = sx - (float)j;
    float p = sl[j] * t;
    acc0 = acc0 + p;
  }
  acc0 = acc0 + acc1;
  acc0 = acc0 + acc2;
  return c0 + acc0;
}

// Paper eq. (1) (increments clamped to [0,1]) -- NOT the reference's arithmetic, an explicit option:
// scale = C_i + slope_i * frac, i = floor(S x) in [0, K-2].  Knots are uniformly spaced, so the
// interval is a direct index (no search); on the device C and sl live in LDS and the two reads are
// per-lane gathers.
CURL_HD float scale_pwl(float x, const float* C, const float* sl, int K) {
  float sx = (float)(K - 1) * x;
  float fi = clampf(floorf(sx), 0.0f, (float)(K - 2));
  int i = (int)fi;
  float frac = clamp01(sx - fi);
  return fmaf(sl[i], frac, C[i]);
}

// the same with knot and slope interleaved, {C_i, slope_i} at tab[2 i]: one 8-byte gather per lookup (fused kernels)
CURL_HD float scale_pwl_pairs(float x, const float* tab, int K) {
  float sx = (float)(K - 1) * x;
  float fi = clampf(floorf(sx), 0.0f, (float)(K - 2));
  int i = (int)fi;
  float frac = clamp01(sx - fi);
  return fmaf(tab[2 * i + 1], frac, tab[2 * i]);
}

// curves.py:31-32 in torch's evaluation order (scale_exact) from the same interleaved table: C_0 at tab[0], slope_j at
// tab[2 j + 1] -- the validation mode of the fused stages (CURL_F_EXACT_ORDER): every lane reads the same address
// (broadcast), 4 instructions per term instead of one fma per curve
CURL_HD float scale_exact_pairs(float x, const float* tab, int K) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const int n_terms = K - 2;
  const float sx = (float)(K - 1) * x;
  float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
  int j = 0;
  for (; j + 16 <= n_terms;) {
    for (int e = 0; e < 16; ++e, ++j) {
      float t = sx - (float)j;
      float p = tab[2 * j + 1] * t;
      acc0 = acc0 + p;
    }
    acc1 = acc1 + acc0;
    acc0 = 0.0f;
    if ((j & 0xF0) == 0) {
      acc2 = acc2 + acc1;
      acc1 = 0.0f;
    }
  }
  for (; j < n_terms; ++j) {
    float t = sx - (float)j;
    float p = tab[2 * j + 1] * t;
    acc0 = acc0 + p;
  }
  acc0 = acc0 + acc1;
  acc0 = acc0 + acc2;
  return tab[0] + acc0;
}
// MODE 0: the paper's clamped interpolation (CURL_F_PWL); 1: the reference's sum, term by term (CURL_F_EXACT_ORDER)
template <int MODE>
CURL_HD float scale_tab(float x, const float* tab, int K) {
  return MODE == 1 ? scale_exact_pairs(x, tab, K) : scale_pwl_pairs(x, tab, K);
}

struct LayerCoef {
  Affine lab[3], rgb[3], hsv[4];
};

// ---------------------------------------------------------------- fused stages over the N pixels of a lane
// model.py:151-157 : rgb2lab -> adjust_lab -> *mask -> lab2rgb
// BINARY = the mask is known to be exactly 0 or 1 (bool / uint8 masks, or no mask at all): x*1 == x, so the
// multiply is dropped for m == 1, and pixels with m == 0 are finished by the caller (lab_stage_masked_out).
template <bool BINARY, int N, bool CLAMP12 = false>
CURL_HD void lab_stage_n(PxN<N>& p, const float (&m)[N], const Affine* k) {
  rgb2lab_n<N, 1>(p);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Px o = adjust3(Px{p.c0[i], p.c1[i], p.c2[i]}, k[0], k[1], k[2]);
    if (!BINARY) {
      o.c0 *= m[i];
      o.c1 *= m[i];
      o.c2 *= m[i];
    }
    p.c0[i] = o.c0, p.c1[i] = o.c1, p.c2[i] = o.c2;
  }
  lab2rgb_n<N, CLAMP12, 1>(p);
}
// what model.py:154-157 yields where the mask is 0: lab2rgb(0,0,0), the same colour for every such pixel
CURL_HD Px lab_stage_masked_out() { return lab2rgb(Px{0.0f, 0.0f, 0.0f}); }

// model.py:163-169 as a stage of its own: rgb2hsv -> adjust_hsv -> *mask -> hsv2rgb (the layer's RGB residual).
// The input is whatever the caller hands over (not known to be in [0,1]): the 1e-9 floors of colors.py:205,240 and the
// refined reciprocal stay.  adjust_hsv4 clamps h, s, v to [0,1]; times a BINARY mask they are still there, so hsv2rgb's
// input clamps go (UNIT), and hsv2rgb(0,0,0) = (0,0,0): a masked-out pixel is exactly 0, which the final `* m` yields.
template <bool BINARY, int N>
CURL_HD void hsv_stage_n(PxN<N>& p, const float (&m)[N], const Affine* k) {
  rgb2hsv_n<N>(p);  // model.py:163
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Px h = adjust_hsv4(Px{p.c0[i], p.c1[i], p.c2[i]}, k[0], k[1], k[2], k[3]);  // model.py:165
    if (!BINARY) {                                                              // model.py:166
      h.c0 *= m[i];
      h.c1 *= m[i];
      h.c2 *= m[i];
    }
    Px o = hsv2rgb<BINARY>(h);  // model.py:169
    if (BINARY) {               // m in {0, 1}: hsv2rgb(hsv * 0) == 0 == hsv2rgb(hsv) * 0
      o.c0 *= m[i];
      o.c1 *= m[i];
      o.c2 *= m[i];
    }
    p.c0[i] = o.c0, p.c1[i] = o.c1, p.c2[i] = o.c2;
  }
}

// model.py:137-176 minus the dead `feat` lines.  For a BINARY mask every intermediate `* mask`
// (model.py:154,160,166) is the identity where m == 1, and where m == 0 the result is 0 whatever the
// intermediates were (every stage maps finite values to finite values and model.py:170 ends in `* mask`),
// so only the final multiply is kept.
template <bool BINARY, int N>
CURL_HD void curl_layer_n(PxN<N>& p, const float (&m)[N], const LayerCoef& k) {
  PxN<N> in = p;
  lab_stage_n<BINARY, N, true>(p, m, k.lab);  // channels 1, 2 come back clamped: adjust3 would clamp them first
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Px o = adjust3<true>(Px{p.c0[i], p.c1[i], p.c2[i]}, k.rgb[0], k.rgb[1], k.rgb[2]);  // model.py:159
    if (!BINARY) {                                                                  // model.py:160
      o.c0 *= m[i];
      o.c1 *= m[i];
      o.c2 *= m[i];
    }
    p.c0[i] = o.c0, p.c1[i] = o.c1, p.c2[i] = o.c2;
  }
  // a binary mask leaves the clamped RGB of adjust3 untouched: everything up to hsv2rgb stays in [0,1]
  rgb2hsv_n<N, BINARY>(p);  // model.py:163
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Px h = adjust_hsv4<BINARY>(Px{p.c0[i], p.c1[i], p.c2[i]}, k.hsv[0], k.hsv[1], k.hsv[2], k.hsv[3]);  // model.py:165
    if (!BINARY) {                                                                                 // model.py:166
      h.c0 *= m[i];
      h.c1 *= m[i];
      h.c2 *= m[i];
    }
    Px res = hsv2rgb<BINARY>(h);                           // model.py:169
    p.c0[i] = clamp01(in.c0[i] + res.c0) * m[i];           // model.py:170
    p.c1[i] = clamp01(in.c1[i] + res.c1) * m[i];
    p.c2[i] = clamp01(in.c2[i] + res.c2) * m[i];
  }
}

// single-pixel forms
template <bool BINARY>
CURL_HD Px lab_stage(Px in, float m, const Affine* k) {
  PxN<1> q{{in.c0}, {in.c1}, {in.c2}};
  const float mm[1] = {m};
  lab_stage_n<BINARY, 1>(q, mm, k);
  return Px{q.c0[0], q.c1[0], q.c2[0]};
}
template <bool BINARY>
CURL_HD Px hsv_stage(Px in, float m, const Affine* k) {
  PxN<1> q{{in.c0}, {in.c1}, {in.c2}};
  const float mm[1] = {m};
  hsv_stage_n<BINARY, 1>(q, mm, k);
  return Px{q.c0[0], q.c1[0], q.c2[0]};
}
template <bool BINARY>
CURL_HD Px curl_layer(Px in, float m, const LayerCoef& k) {
  PxN<1> q{{in.c0}, {in.c1}, {in.c2}};
  const float mm[1] = {m};
  curl_layer_n<BINARY, 1>(q, mm, k);
  return Px{q.c0[0], q.c1[0], q.c2[0]};
}

}  // namespace curlm
