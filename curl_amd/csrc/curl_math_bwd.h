// curl_math_bwd.h -- reverse-mode derivative of curl_layer (curl_math.h) for one pixel.
//
// What it must reproduce is torch autograd through the reference's eager ops (curves.py, colors.py,
// model.py:137-176), whose conventions are:
//   * torch.clamp passes the gradient where min <= x <= max (boundaries included);
//   * the 0/1 masks (`.le() .gt() .eq() .lt() .ge()`, colors.py:37,45,110,118,223,228) carry no gradient:
//     a blended expression differentiates as the branch that was selected;
//   * torch.max/min over the channel dim send the gradient to the FIRST index attaining the extremum
//     (colors.py:211-212);
//   * the regulariser and the knots enter only through the per-image curve coefficients: with
//     scale = C0 + sum_j slope_j (S x - j), the knot gradient of one curve needs just two pixel sums,
//     P = sum g_scale  and  Q = sum g_scale * x_in   (d/dslope_j = S Q - j P, d/dC0 = P), which the
//     kernel accumulates per image; the chain rule to the raw knots runs once per image (knots_bwd).
//
// Round 3: every stage is ONE function that runs the forward and leaves the factors its pullback needs (`*_t`: a
// "tape" of a few floats and lane predicates) -- nothing is recomputed on the way back: the round-2 code evaluated every
// pow twice and built each clamp gate from two subtractions, two shifts and a bit op (168 mask instructions of 730 per
// pixel; 225 VGPRs, two waves per SIMD).  Gates are lane predicates in SGPR pairs now: `clamp(pre) == pre` is one
// v_cmp_eq_f32_e64, applying it one v_cndmask_b32_e64 (the VOP3 forms: 4 issue cycles; it is the VOP2 `v_cndmask ..., vcc`
// hipcc emits for `?:` that costs 23).  Constant factors of the derivatives (3 of the cube, 1/3 of the cube
// root, 2.4/1.055 ...) ride on the constants of the linear maps behind them.
//
// (Tried and withdrawn: skipping the converters' rare linear branches per wave as the forward kernel does
// (select_le_lazy).  With one pixel per lane at a time a branch covers three values, not twelve, and its compare ->
// SGPR -> scalar OR -> branch latency sits in the only chain the wave has: +2.5 % at full frames, +4.6 % on the training
// crop batch, -0.9 % even where every wave skips -- profiles/r03/exp8_bwd_lazy_branches_lost.log.  Nor do the forward's
// predicated overwrites pay here: -0.4 % at full frames, +2.7 % on the crop batch -- exp19_bwd_predicated_lost.log: every
// exec switch sits in the one dependent chain a wave has.)
//
// Same dual compilation as curl_math.h (device: gfx950 kernels; host: the test-only twin, predicates as bool).
#pragma once
#include "curl_math.h"

namespace curlm {

// ---------------------------------------------------------------- lane predicates
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned long long lmask;  // one bit per lane, in an SGPR pair; & | ~ on it are SALU instructions
#define CURL_LM_CMP(NAME, OP)                                                   \
  __device__ __forceinline__ lmask NAME(float a, float b) {                     \
    lmask m;                                                                    \
    asm("v_cmp_" OP "_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));          \
    return m;                                                                   \
  }                                                                             \
  __device__ __forceinline__ lmask NAME##_u(float a, float uniform_b) {         \
    lmask m;                                                                    \
    asm("v_cmp_" OP "_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(uniform_b));  \
    return m;                                                                   \
  }
CURL_LM_CMP(lm_eq, "eq")
CURL_LM_CMP(lm_le, "le")
CURL_LM_CMP(lm_lt, "lt")
#undef CURL_LM_CMP
__device__ __forceinline__ lmask lm_neg(float a) {  // a < 0
  lmask m;
  asm("v_cmp_gt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(a));
  return m;
}
__device__ __forceinline__ float lm_sel(lmask m, float a, float b) {  // m ? a : b  (the value not taken may be NaN)
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ float lm_keep(lmask m, float a) {  // m ? a : +0
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ lmask lm_and(lmask a, lmask b) { return a & b; }
__device__ __forceinline__ lmask lm_or(lmask a, lmask b) { return a | b; }
__device__ __forceinline__ lmask lm_andn(lmask a, lmask b) { return a & ~b; }  // a and not b
__device__ __forceinline__ lmask lm_nor(lmask a, lmask b) { return ~(a | b); }
#else
typedef bool lmask;
inline lmask lm_eq(float a, float b) { return a == b; }
inline lmask lm_le(float a, float b) { return a <= b; }
inline lmask lm_lt(float a, float b) { return a < b; }
inline lmask lm_eq_u(float a, float b) { return a == b; }
inline lmask lm_le_u(float a, float b) { return a <= b; }
inline lmask lm_lt_u(float a, float b) { return a < b; }
inline lmask lm_neg(float a) { return a < 0.0f; }
inline float lm_sel(lmask m, float a, float b) { return m ? a : b; }
inline float lm_keep(lmask m, float a) { return m ? a : 0.0f; }
inline lmask lm_and(lmask a, lmask b) { return a && b; }
inline lmask lm_or(lmask a, lmask b) { return a || b; }
inline lmask lm_andn(lmask a, lmask b) { return a && !b; }
inline lmask lm_nor(lmask a, lmask b) { return !(a || b); }
#endif

// y = clamp(x, lo, hi) and torch.clamp's gradient gate lo <= x <= hi (boundaries included; -0 passes at lo = 0: -0 == +0)
CURL_HD float clamp_gate(float x, float lo, float hi, lmask& pass) {
  float y = clampf(x, lo, hi);
  pass = lm_eq(y, x);
  return y;
}
// x with its sign flipped where u is NOT negative: -sign(u) x for u != 0  (d(2 - |u|) / du = -sign(u))
CURL_HD float neg_sign_of(float u, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return i2f(__builtin_amdgcn_bitop3_b32(f2i(x), f2i(u), (int)0x80000000u, 0xD2));  // x ^ (~u & 0x80000000): 0xF0 ^ (0x33 & 0xAA), one instruction
#else
  return i2f(f2i(x) ^ (~f2i(u) & (int)0x80000000u));
#endif
}

// ---------------------------------------------------------------- curves
// y = clamp01(x * (a + b x)): forward + tape
struct CurveT {
  float x, ds;  // the (clamped) input, d pre / d x = a + 2 b x
  lmask pass;   // 0 <= pre <= 1
};
CURL_HD float curve_self_t(float x, Affine k, CurveT& t) {
  float s = fmaf(k.b, x, k.a);
  float y = clamp_gate(x * s, 0.0f, 1.0f, t.pass);
  t.x = x;
  t.ds = fmaf(k.b, x, s);
  return y;
}
CURL_HD float curve_self_pull(const CurveT& t, float gy, float& P, float& Q) {
  float g_pre = lm_keep(t.pass, gy);
  float g_s = g_pre * t.x;
  P += g_s;
  Q = fmaf(g_s, t.x, Q);
  return g_pre * t.ds;
}
// y = clamp01(xo * (a + b xi)), xi != xo (adjust_hsv's H -> S curve, curves.py:67)
struct CrossT {
  float xo, xi, s;
  lmask pass;
};
CURL_HD float curve_cross_t(float xo, float xi, Affine k, CrossT& t) {
  t.s = fmaf(k.b, xi, k.a);
  t.xo = xo, t.xi = xi;
  return clamp_gate(xo * t.s, 0.0f, 1.0f, t.pass);
}
CURL_HD void curve_cross_pull(const CrossT& t, Affine k, float gy, float& g_xo, float& g_xi, float& P, float& Q) {
  float g_pre = lm_keep(t.pass, gy);
  float g_s = g_pre * t.xo;
  g_xo = g_pre * t.s;
  g_xi = fmaf(g_s, k.b, g_xi);
  P += g_s;
  Q = fmaf(g_s, t.xi, Q);
}

// adjust3 (curves.py:90-133 / 136-180): channel 0 meets its curve unclamped, channels 1, 2 clamped first (curves.py:36)
struct Adjust3T {
  CurveT c[3];
  lmask in1, in2;  // 0 <= p.c1 <= 1, 0 <= p.c2 <= 1
};
CURL_HD Px adjust3_t(Px p, const Affine* k, Adjust3T& t) {
  Px o;
  o.c0 = curve_self_t(p.c0, k[0], t.c[0]);
  o.c1 = curve_self_t(clamp_gate(p.c1, 0.0f, 1.0f, t.in1), k[1], t.c[1]);
  o.c2 = curve_self_t(clamp_gate(p.c2, 0.0f, 1.0f, t.in2), k[2], t.c[2]);
  return o;
}
CURL_HD Px adjust3_pull(const Adjust3T& t, Px g, float* P, float* Q) {
  Px gi;
  gi.c0 = curve_self_pull(t.c[0], g.c0, P[0], Q[0]);
  gi.c1 = lm_keep(t.in1, curve_self_pull(t.c[1], g.c1, P[1], Q[1]));
  gi.c2 = lm_keep(t.in2, curve_self_pull(t.c[2], g.c2, P[2], Q[2]));
  return gi;
}

// ---------------------------------------------------------------- RGB -> Lab   colors.py:27-62
struct Rgb2LabT {
  float dlin[3];  // d linear_c / d x_c
  float df3[3];   // 3 d f / d t_r  (t^(-2/3), or 3 / (3 eps^2) on the linear branch)
};
CURL_HD Px rgb2lab_t(Px p, Rgb2LabT& t) {
  // No implicit a*b+c contraction in here (the explicit fmaf's stay): hipcc's default contracts across statements depending on
  // how many uses a product has, so the SAME source gave different bits in different callers -- and CURLLoss's backward needs
  // this function to map equal colours to equal values (curl_math_loss.h: sign(lp - lt) must be 0 where pred == target).
#pragma clang fp contract(off)
  const float x[3] = {p.c0, p.c1, p.c2};
  float u[3], e[3], lin[3];
  for (int c = 0; c < 3; ++c) u[c] = fmaf(x[c], kInv1055, (float)(0.055 / 1.055));
  CURL_TRANS_BEGIN();
  for (int c = 0; c < 3; ++c) e[c] = hw_log2(u[c]);
  for (int c = 0; c < 3; ++c) e[c] *= kGammaFrac;
  for (int c = 0; c < 3; ++c) e[c] = hw_exp2(e[c]);  // u^0.4
  CURL_TRANS_END();
  const float k_lin = kInv1292, k_pow = (float)2.4 * kInv1055;
  for (int c = 0; c < 3; ++c) {
    const lmask lo = lm_le_u(x[c], kSrgbThr);  // the linear branch (colors.py:37)
    const float ue = u[c] * e[c];              // u^1.4: d u^2.4 / du = 2.4 u^1.4
    lin[c] = lm_sel(lo, x[c] * kInv1292, u[c] * ue);
    t.dlin[c] = lm_sel(lo, k_lin, ue * k_pow);
  }
  float tt[3], f[3], lg[3];
  tt[0] = fmaf(0.180423f * kInvXn, lin[2], fmaf(0.357580f * kInvXn, lin[1], (0.412453f * kInvXn) * lin[0]));
  tt[1] = fmaf(0.072169f, lin[2], fmaf(0.715160f, lin[1], 0.212671f * lin[0]));
  tt[2] = fmaf(0.950227f * kInvZn, lin[2], fmaf(0.119193f * kInvZn, lin[1], (0.019334f * kInvZn) * lin[0]));
  CURL_TRANS_BEGIN();
  for (int r = 0; r < 3; ++r) lg[r] = hw_log2(tt[r]);
  for (int r = 0; r < 3; ++r) f[r] = hw_exp2(lg[r] * kThird);                     // t^(1/3)
  for (int r = 0; r < 3; ++r) t.df3[r] = hw_exp2(lg[r] * (float)(-2.0 / 3.0));    // t^(-2/3) = 3 d t^(1/3) / dt
  CURL_TRANS_END();
  const float k_df = (3.0f * kInv3Eps2);
  for (int r = 0; r < 3; ++r) {
    const lmask lo = lm_le_u(tt[r], kEps3);  // colors.py:45-47
    f[r] = lm_sel(lo, fmaf(tt[r], kInv3Eps2, k4_29), f[r]);
    t.df3[r] = lm_sel(lo, k_df, t.df3[r]);
  }
  Px o;
  // L = 1.16 fy - 0.16 as 1.16 (fy - 4/29): EXACTLY 0 at black (fy = 4/29), where the reference's float32 (116 * fy rounds to
  // 16.0, colors.py:50-56) and float64 evaluations both give L >= 0 and torch.clamp's gate at 0 passes the gradient; the single
  // fma gave -1.6e-9 there and closed it on every black pixel of a photograph (tests: photograph_dark, the black colour)
  o.c0 = (f[1] - k4_29) * 1.16f;
  o.c1 = fmaf(f[0] - f[1], (float)(500.0 / 220.0), 0.5f);
  o.c2 = fmaf(f[1] - f[2], (float)(200.0 / 220.0), 0.5f);
  return o;
}
CURL_HD Px rgb2lab_pull(const Rgb2LabT& t, Px g) {
  // L = 1.16 fy - 0.16, a = (fx - fy) ka + 0.5, b = (fy - fz) kb + 0.5; the 1/3 of df3 rides on these constants
  const float ka3 = (float)(500.0 / 220.0 / 3.0), kb3 = (float)(200.0 / 220.0 / 3.0), kl3 = (float)(1.16 / 3.0);
  const float ga = g.c1 * ka3, gb = g.c2 * kb3;
  const float g_t0 = ga * t.df3[0];
  const float g_t1 = fmaf(g.c0, kl3, gb - ga) * t.df3[1];
  const float g_t2 = -gb * t.df3[2];
  Px gi;
  gi.c0 = fmaf(g_t2, 0.019334f * kInvZn, fmaf(g_t1, 0.212671f, g_t0 * (0.412453f * kInvXn))) * t.dlin[0];
  gi.c1 = fmaf(g_t2, 0.119193f * kInvZn, fmaf(g_t1, 0.715160f, g_t0 * (0.357580f * kInvXn))) * t.dlin[1];
  gi.c2 = fmaf(g_t2, 0.950227f * kInvZn, fmaf(g_t1, 0.072169f, g_t0 * (0.180423f * kInvXn))) * t.dlin[2];
  return gi;
}

// ---------------------------------------------------------------- Lab -> RGB   colors.py:88-123
struct Lab2RgbT {
  float dX3[3];  // (d f^-1 / d f) / 3 : f^2, or eps^2 on the linear branch
  float dn[3];   // (d gamma / d v) / (1.055 / 2.4) : v^(1/2.4 - 1), or 12.92 * 2.4 / 1.055 on the linear branch
};
CURL_HD Px lab2rgb_t(Px p, Lab2RgbT& t) {
  const float fy = fmaf(p.c0, (float)(100.0 / 116.0), (float)(16.0 / 116.0));
  const float f[3] = {fmaf(p.c1, (float)(220.0 / 500.0), fy - (float)(110.0 / 500.0)), fy,
                      fmaf(p.c2, (float)(-220.0 / 200.0), fy + (float)(110.0 / 200.0))};
  float X[3], v[3], pw[3], rv[3];
  const float k_dx = (k3Eps2 * (float)(1.0 / 3.0));
  for (int i = 0; i < 3; ++i) {
    const lmask lo = lm_le_u(f[i], kEps);  // colors.py:110-111
    const float f2 = f[i] * f[i];
    X[i] = lm_sel(lo, fmaf(f[i], k3Eps2, -(k3Eps2 * k4_29)), f2 * f[i]);
    t.dX3[i] = lm_sel(lo, k_dx, f2);
  }
  v[0] = fmaf(-0.4985314f * kZn, X[2], fmaf(-1.5371385f, X[1], (3.2404542f * kXn) * X[0]));
  v[1] = fmaf(0.0415560f * kZn, X[2], fmaf(1.8760108f, X[1], (-0.9692660f * kXn) * X[0]));
  v[2] = fmaf(1.0572252f * kZn, X[2], fmaf(-0.2040259f, X[1], (0.0556434f * kXn) * X[0]));
  CURL_TRANS_BEGIN();
  for (int r = 0; r < 3; ++r) pw[r] = hw_log2(v[r]);
  for (int r = 0; r < 3; ++r) pw[r] = hw_exp2(pw[r] * kInvGamma);  // v^(1/2.4)
  for (int r = 0; r < 3; ++r) rv[r] = hw_rcp(v[r]);
  CURL_TRANS_END();
  const float k_dn = (float)(12.92 * 2.4 / 1.055);
  float o[3];
  for (int r = 0; r < 3; ++r) {
    const lmask lo = lm_le_u(v[r], kLinThr);  // colors.py:118-119
    o[r] = lm_sel(lo, v[r] * 12.92f, fmaf(pw[r], 1.055f, -0.055f));
    t.dn[r] = lm_sel(lo, k_dn, pw[r] * rv[r]);
  }
  return Px{o[0], o[1], o[2]};  // NOT clamped (colors.py:121-123)
}
CURL_HD Px lab2rgb_pull(const Lab2RgbT& t, Px g) {
  const float gv0 = g.c0 * t.dn[0], gv1 = g.c1 * t.dn[1], gv2 = g.c2 * t.dn[2];
  const float gX0 = fmaf(gv2, 0.0556434f * kXn, fmaf(gv1, -0.9692660f * kXn, gv0 * (3.2404542f * kXn))) * t.dX3[0];
  const float gX1 = fmaf(gv2, -0.2040259f, fmaf(gv1, 1.8760108f, gv0 * -1.5371385f)) * t.dX3[1];
  const float gX2 = fmaf(gv2, 1.0572252f * kZn, fmaf(gv1, 0.0415560f * kZn, gv0 * (-0.4985314f * kZn))) * t.dX3[2];
  // fy = c1 L + c2, fx = ca A + fy - .., fz = cb B + fy + ..; the 3 of dX3 and the 1.055/2.4 of dn ride on these constants
  const float kk = (float)(3.0 * 1.055 / 2.4);
  Px gi;
  gi.c0 = ((gX0 + gX1) + gX2) * (kk * (float)(100.0 / 116.0));
  gi.c1 = gX0 * (kk * (float)(220.0 / 500.0));
  gi.c2 = gX2 * (kk * (float)(-220.0 / 200.0));
  return gi;
}

// ---------------------------------------------------------------- RGB -> HSV   colors.py:195-242
struct Rgb2HsvT {
  lmask pin[3];            // 1e-9 <= input_c <= 1 (colors.py:205)
  lmask e[3];              // channel c attains the maximum (the tie terms of colors.py:221-224 add)
  lmask max1, min0, min1;  // first index attaining max / min (torch.max / torch.min over dim 1): max0 == e[0]
  lmask live, oh, os, ov;  // max != min; the output clamps of colors.py:240 pass
  float df, dfi, mxi, Nn;  // max - min, 1 / (max - min) (0 where flat), 1 / max, the selected hue numerator
};
CURL_HD Px rgb2hsv_t(Px p, Rgb2HsvT& t) {
#pragma clang fp contract(off)  // (as rgb2lab_t)
  const float r = clamp_gate(p.c0, kHsvFloor, 1.0f, t.pin[0]), g = clamp_gate(p.c1, kHsvFloor, 1.0f, t.pin[1]),
              b = clamp_gate(p.c2, kHsvFloor, 1.0f, t.pin[2]);
  const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
  t.e[0] = lm_eq(r, mx), t.e[1] = lm_eq(g, mx), t.e[2] = lm_eq(b, mx);
  t.max1 = lm_andn(t.e[1], t.e[0]);
  t.min0 = lm_eq(r, mn);
  t.min1 = lm_andn(lm_eq(g, mn), t.min0);
  const float nd = mn - mx;
  t.live = lm_neg(nd);
  t.df = -nd;
  CURL_TRANS_BEGIN();
  const float rdf = hw_rcp(t.df);
  t.mxi = hw_rcp(mx);
  CURL_TRANS_END();
  t.dfi = lm_keep(t.live, rdf);
  const float n0 = g - b, n1 = b - r, n2 = r - g;
  t.Nn = (lm_keep(t.e[0], n0) + lm_keep(t.e[1], n1)) + lm_keep(t.e[2], n2);
  const float h6 = lm_keep(t.live, (lm_keep(t.e[0], n0 * t.dfi) + lm_keep(t.e[1], fmaf(n1, t.dfi, 2.0f))) +
                                       lm_keep(t.e[2], fmaf(n2, t.dfi, 4.0f)));
  const float h = (h6 + lm_keep(lm_neg(h6), 6.0f)) * (float)(1.0 / 6.0);
  const float s = t.df * t.mxi;
  Px o;
  o.c0 = clamp_gate(h, kHsvFloor, 1.0f, t.oh);
  o.c1 = clamp_gate(s, kHsvFloor, 1.0f, t.os);
  o.c2 = clamp_gate(mx, kHsvFloor, 1.0f, t.ov);
  return o;
}
CURL_HD Px rgb2hsv_pull(const Rgb2HsvT& t, Px g) {
  const float g_h6 = lm_keep(lm_and(t.oh, t.live), g.c0 * (float)(1.0 / 6.0));
  const float g_s = lm_keep(t.os, g.c1), g_v = lm_keep(t.ov, g.c2);
  const float g_N = g_h6 * t.dfi;
  const float g_df = fmaf(g_s, t.mxi, -((g_N * t.Nn) * t.dfi));
  const float g_mx = fmaf(-(g_s * t.df) * t.mxi, t.mxi, g_v + g_df);
  const float g_mn = -g_df;
  // d N / d (r, g, b) = (e_b - e_g, e_r - e_b, e_g - e_r)
  const float Nr = lm_keep(t.e[0], g_N), Ng = lm_keep(t.e[1], g_N), Nb = lm_keep(t.e[2], g_N);
  const lmask max2 = lm_nor(t.e[0], t.max1), min2 = lm_nor(t.min0, t.min1);
  Px gi;
  gi.c0 = lm_keep(t.pin[0], ((Nb - Ng) + lm_keep(t.e[0], g_mx)) + lm_keep(t.min0, g_mn));
  gi.c1 = lm_keep(t.pin[1], ((Nr - Nb) + lm_keep(t.max1, g_mx)) + lm_keep(t.min1, g_mn));
  gi.c2 = lm_keep(t.pin[2], ((Ng - Nr) + lm_keep(max2, g_mx)) + lm_keep(min2, g_mn));
  return gi;
}

// ---------------------------------------------------------------- adjust_hsv   curves.py:41-87
// UNIT: s and v arrive in [0,1] (straight from rgb2hsv's output clamp, times a 0/1 mask): their input clamps are the
// identity and always pass.
struct AdjustHsvT {
  CurveT hh, ss, vv;  // H->H, S->S, V->V
  CrossT hs;          // H->S on the adjusted hue
  lmask in1, in2;
};
template <bool UNIT>
CURL_HD Px adjust_hsv4_t(Px p, const Affine* k, AdjustHsvT& t) {
  const float h1 = curve_self_t(p.c0, k[0], t.hh);
  const float sc = UNIT ? p.c1 : clamp_gate(p.c1, 0.0f, 1.0f, t.in1), vc = UNIT ? p.c2 : clamp_gate(p.c2, 0.0f, 1.0f, t.in2);
  const float s1 = curve_cross_t(sc, h1, k[1], t.hs);
  Px o;
  o.c0 = h1;
  o.c1 = curve_self_t(s1, k[2], t.ss);
  o.c2 = curve_self_t(vc, k[3], t.vv);
  return o;
}
template <bool UNIT>
CURL_HD Px adjust_hsv4_pull(const AdjustHsvT& t, const Affine* k, Px g, float* P, float* Q) {
  const float g_vc = curve_self_pull(t.vv, g.c2, P[3], Q[3]);
  const float g_s1 = curve_self_pull(t.ss, g.c1, P[2], Q[2]);
  float g_sc, g_h1 = g.c0;
  curve_cross_pull(t.hs, k[1], g_s1, g_sc, g_h1, P[1], Q[1]);
  Px gi;
  gi.c0 = curve_self_pull(t.hh, g_h1, P[0], Q[0]);
  gi.c1 = UNIT ? g_sc : lm_keep(t.in1, g_sc);
  gi.c2 = UNIT ? g_vc : lm_keep(t.in2, g_vc);
  return gi;
}

// ---------------------------------------------------------------- HSV -> RGB   colors.py:131-177
// The trapezoid form of curl_math.h's hsv2rgb: channel = base -/+ d clamp01(2 - |6h - c|).  Its kinks carry torch's
// clamp conventions: clamp(6h - 1) - clamp(6h - 4) passes the gradient of exactly one ramp at 6h = 1, 2, 4, 5 (boundaries
// included), and so does [0 <= 2 - |6h - 3| <= 1] with slope -sign(6h - 3).
// UNIT: h, s, v in [0,1] (adjust_hsv's clamps, times a 0/1 mask): the input clamps pass, and so do the output clamps --
// with d = fl(v s) <= v and q = fl(v - d) every channel lies in [0, v] (rounding is monotone).
struct Hsv2RgbT {
  float ss, vv, d, ur, ug, ub, tr, tg, tb;
  lmask pr, pg, pb;     // 0 <= 2 - |u| <= 1
  lmask ih, is, iv;     // input clamps
  lmask o_r, o_g, o_b;  // output clamps
};
template <bool UNIT>
CURL_HD Px hsv2rgb_t(Px p, Hsv2RgbT& t) {
  const float hh = UNIT ? p.c0 : clamp_gate(p.c0, 0.0f, 1.0f, t.ih);
  t.ss = UNIT ? p.c1 : clamp_gate(p.c1, 0.0f, 1.0f, t.is);
  t.vv = UNIT ? p.c2 : clamp_gate(p.c2, 0.0f, 1.0f, t.iv);
  t.d = t.vv * t.ss;
  const float q = t.vv - t.d;
  t.ur = fmaf(hh, 6.0f, -3.0f), t.ug = fmaf(hh, 6.0f, -2.0f), t.ub = fmaf(hh, 6.0f, -4.0f);
  t.tr = clamp_gate(2.0f - fabsf(t.ur), 0.0f, 1.0f, t.pr);
  t.tg = clamp_gate(2.0f - fabsf(t.ug), 0.0f, 1.0f, t.pg);
  t.tb = clamp_gate(2.0f - fabsf(t.ub), 0.0f, 1.0f, t.pb);
  const float r = fmaf(t.tr, -t.d, t.vv), g = fmaf(t.tg, t.d, q), b = fmaf(t.tb, t.d, q);
  if (UNIT) return Px{r, g, b};
  return Px{clamp_gate(r, 0.0f, 1.0f, t.o_r), clamp_gate(g, 0.0f, 1.0f, t.o_g), clamp_gate(b, 0.0f, 1.0f, t.o_b)};
}
template <bool UNIT>
CURL_HD Px hsv2rgb_pull(const Hsv2RgbT& t, Px g) {
  const float gr = UNIT ? g.c0 : lm_keep(t.o_r, g.c0), gg = UNIT ? g.c1 : lm_keep(t.o_g, g.c1),
              gb = UNIT ? g.c2 : lm_keep(t.o_b, g.c2);
  // r = v - d tr, g = q + d tg, b = q + d tb, q = v - d, d = v s
  const float g_q = gg + gb;
  const float g_d = fmaf(gb, t.tb, fmaf(gg, t.tg, -(gr * t.tr))) - g_q;
  const float g_vv = fmaf(g_d, t.ss, gr + g_q);
  const float g_ss = g_d * t.vv;
  // t = clamp01(2 - |u|), u = 6 h - c: d t / d h = -6 sign(u) inside the gate
  const float g_tr = -(gr * t.d), g_tg = gg * t.d, g_tb = gb * t.d;
  const float g_h = ((lm_keep(t.pr, neg_sign_of(t.ur, g_tr)) + lm_keep(t.pg, neg_sign_of(t.ug, g_tg))) +
                     lm_keep(t.pb, neg_sign_of(t.ub, g_tb))) * 6.0f;
  if (UNIT) return Px{g_h, g_ss, g_vv};
  return Px{lm_keep(t.ih, g_h), lm_keep(t.is, g_ss), lm_keep(t.iv, g_vv)};
}

// ---------------------------------------------------------------- stand-alone pullbacks
// (the polynomial model's and CURLLoss' backward passes, curl_math_poly.h / curl_math_loss.h: forward + pullback of one
// converter at the point p)
CURL_HD float pass01(float x) {  // [0 <= x <= 1] as 1.0 / 0.0
  lmask m;
  clamp_gate(x, 0.0f, 1.0f, m);
  return lm_keep(m, 1.0f);
}
CURL_HD Px rgb2lab_bwd(Px p, Px g) {
  Rgb2LabT t;
  rgb2lab_t(p, t);
  return rgb2lab_pull(t, g);
}
CURL_HD Px lab2rgb_bwd(Px p, Px g) {
  Lab2RgbT t;
  lab2rgb_t(p, t);
  return lab2rgb_pull(t, g);
}
CURL_HD Px rgb2hsv_bwd(Px p, Px g) {
  Rgb2HsvT t;
  rgb2hsv_t(p, t);
  return rgb2hsv_pull(t, g);
}
CURL_HD Px hsv2rgb_bwd(Px p, Px g) {
  Hsv2RgbT t;
  hsv2rgb_t<false>(p, t);
  return hsv2rgb_pull<false>(t, g);
}

// ---------------------------------------------------------------- the whole layer   model.py:137-176
// P, Q [10]: += per curve (0-2 lab, 3-5 rgb, 6-9 hsv).  Returns d loss / d in.
// BINARY: m is exactly 0 or 1.  Where m == 1 the three intermediate `* mask` are the identity; where m == 0 the incoming
// gradient gout * m is 0 and every factor on the tape is finite (the selects discard the branch not taken), so every
// product on the way back is an exact 0: the intermediate multiplies go, as in the forward kernel.
// NEED_GIN = false: the caller wants the knot gradients only (training: the image is data, main.py:287) -- the chain stops
// at the Lab curves' sums; RGB2LAB's pullback, its tape factors (two extra transcendentals per channel) and the residual
// path's add are never computed.  P and Q are the same bits either way.
template <bool BINARY, bool NEED_GIN = true>
CURL_HD Px curl_layer_bwd(Px in, float m, const LayerCoef& k, Px gout, float* P, float* Q) {
  Rgb2LabT t_lab;
  Adjust3T t_al, t_ar;
  Lab2RgbT t_rgb;
  Rgb2HsvT t_hsv;
  AdjustHsvT t_ah;
  Hsv2RgbT t_res;
  // forward, every stage leaving its tape
  Px x = adjust3_t(rgb2lab_t(in, t_lab), k.lab, t_al);
  if (!BINARY) x = Px{x.c0 * m, x.c1 * m, x.c2 * m};  // model.py:154
  x = adjust3_t(lab2rgb_t(x, t_rgb), k.rgb, t_ar);
  if (!BINARY) x = Px{x.c0 * m, x.c1 * m, x.c2 * m};  // model.py:160
  x = adjust_hsv4_t<BINARY>(rgb2hsv_t(x, t_hsv), k.hsv, t_ah);
  if (!BINARY) x = Px{x.c0 * m, x.c1 * m, x.c2 * m};  // model.py:166
  const Px res = hsv2rgb_t<BINARY>(x, t_res);
  // out = clamp01(in + res) * m   (model.py:170)
  lmask o0, o1, o2;
  clamp_gate(in.c0 + res.c0, 0.0f, 1.0f, o0);
  clamp_gate(in.c1 + res.c1, 0.0f, 1.0f, o1);
  clamp_gate(in.c2 + res.c2, 0.0f, 1.0f, o2);
  const Px g_pre{lm_keep(o0, gout.c0 * m), lm_keep(o1, gout.c1 * m), lm_keep(o2, gout.c2 * m)};
  // and back
  Px g = hsv2rgb_pull<BINARY>(t_res, g_pre);
  if (!BINARY) g = Px{g.c0 * m, g.c1 * m, g.c2 * m};
  g = rgb2hsv_pull(t_hsv, adjust_hsv4_pull<BINARY>(t_ah, k.hsv, g, P + 6, Q + 6));
  if (!BINARY) g = Px{g.c0 * m, g.c1 * m, g.c2 * m};
  g = lab2rgb_pull(t_rgb, adjust3_pull(t_ar, g, P + 3, Q + 3));
  if (!BINARY) g = Px{g.c0 * m, g.c1 * m, g.c2 * m};
  g = adjust3_pull(t_al, g, P, Q);
  if constexpr (!NEED_GIN) return g;  // (not d loss / d in: the caller ignores it)
  g = rgb2lab_pull(t_lab, g);
  return Px{g.c0 + g_pre.c0, g.c1 + g_pre.c1, g.c2 + g_pre.c2};
}

// ---- per image: (P, Q) of one curve + d loss / d reg  ->  gradient of that curve's RAW knots (pre-exp).
// C = exp(raw) (already computed); scale = C0 + sum_{j<=K-3} slope_j (S x - j); reg = sum_j (slope_{j+1}-slope_j)^2.
// Everything in float64: K is tiny and this runs once per curve per image.
// One knot of it (the kernel hands every (curve, knot) pair to a thread of its own: as one thread per curve the K-long
// loop of float64 chains was ~10 us of every backward call).
CURL_HD float knot_bwd(const float* C, int K, double P, double Q, double g_reg, int kk) {
  const double S = (double)(K - 1);
  // dL/dslope_j  (slope_j = C[j+1]-C[j], j = 0..K-2)
  auto dslope = [&](int j) -> double {
    if (j < 0 || j > K - 2) return 0.0;
    double v = 0.0;
    if (j <= K - 3) v += S * Q - (double)j * P;  // pixels
    // regulariser: terms (slope_j - slope_{j-1})^2 and (slope_{j+1} - slope_j)^2
    auto sl = [&](int i) { return (double)(C[i + 1] - C[i]); };
    if (j >= 1) v += g_reg * 2.0 * (sl(j) - sl(j - 1));
    if (j + 1 <= K - 2) v -= g_reg * 2.0 * (sl(j + 1) - sl(j));
    return v;
  };
  double gC = dslope(kk - 1) - dslope(kk);
  if (kk == 0) gC += P;  // the C0 term of the scale
  return (float)(gC * (double)C[kk]);  // d exp(raw) = C
}
// The same from the five knots around kk held in registers (c[d] = C[kk - 2 + d]; entries outside the curve are never used:
// the conditions below are knot_bwd's): knots_bwd_kernel asks for them at its very top, long before P and Q exist, so that no
// memory latency is left behind its reductions.  Operation for operation knot_bwd: identical bits.
CURL_HD float knot_bwd5(const float (&c)[5], int K, double P, double Q, double g_reg, int kk) {
  const double S = (double)(K - 1);
  // slope differences as knot_bwd's sl(i) = (double)(C[i + 1] - C[i]) for i = kk - 2 .. kk + 1
  const double sl_m2 = (double)(c[1] - c[0]), sl_m1 = (double)(c[2] - c[1]), sl_0 = (double)(c[3] - c[2]), sl_p1 = (double)(c[4] - c[3]);
  auto dslope = [&](int j, double sl_j, double sl_jm1, double sl_jp1) -> double {
    if (j < 0 || j > K - 2) return 0.0;
    double v = 0.0;
    if (j <= K - 3) v += S * Q - (double)j * P;
    if (j >= 1) v += g_reg * 2.0 * (sl_j - sl_jm1);
    if (j + 1 <= K - 2) v -= g_reg * 2.0 * (sl_jp1 - sl_j);
    return v;
  };
  double gC = dslope(kk - 1, sl_m1, sl_m2, sl_0) - dslope(kk, sl_0, sl_m1, sl_p1);
  if (kk == 0) gC += P;
  return (float)(gC * (double)c[2]);
}
CURL_HD void knots_bwd(const float* C, int K, double P, double Q, double g_reg, float* g_raw) {
  for (int kk = 0; kk < K; ++kk) g_raw[kk] = knot_bwd(C, K, P, Q, g_reg, kk);
}

}  // namespace curlm
