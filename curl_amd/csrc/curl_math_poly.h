// curl_math_poly.h -- the polynomial per-pixel path of the reference's live model:
// ChannelPolyLayer / Deg4MobilePolyLayer (model.py:206-415) and TriSpaceRegNet.generate_residual +
// generate_image (model.py:499-520).  Same dual compilation as curl_math.h.
//
// Per pixel and colour space: 3 outputs, each a polynomial of total degree <= 4 in V = 5 variables (3 colour
// channels + x/W + y/H; 126 coefficients) or V = 3 (35).  Evaluated in multivariate Horner form
// (poly_horner.inc, generated): 125 FMAs per output instead of 121 shared multiplies + 126 FMAs.  On gfx950
// the value type is a packed pair of pixels (v_pk_fma_f32) and the coefficients are wave-uniform: they are
// read through a uniform pointer, i.e. scalar loads into SGPRs that feed the packed FMAs directly.
// ~2.6 kFLOP per pixel: this kernel is VALU-bound by a wide margin (no MFMA shape fits: the contraction has
// 3 output columns per 126-deep dot product, 3/16 of the smallest f32 MFMA tile, at the VALU's own rate).
#pragma once
constexpr int kPolySel = 0;  // the converters' threshold selects inside the polynomial model: 0 eager; 2 (predicated) spills 688 B per lane at the kernel's 128-VGPR budget, 1 (skipped per wave too) 1.1 KB -- round 3
#include "curl_math_bwd.h"

namespace curlm {

#if defined(__HIP_DEVICE_COMPILE__)
CURL_HD curl_f2 poly_fmav(curl_f2 a, curl_f2 v, curl_f2 q) { return __builtin_elementwise_fma(a, v, q); }
#endif
// The scalar form on the device is an opaque v_fma_f32: left as fmaf, hipcc's SLP vectoriser packs the independent scalar
// chains of a lane into v_pk_fma_f32 with a shuffle per operand and spills (poly_layer_kernel: 2 400 spilled registers).
CURL_HD float poly_fmav(float a, float v, float q) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(v), "v"(q));
  return r;
#else
  return fmaf(a, v, q);
#endif
}

// A coefficient as the evaluator's value type F.  seq(): from the consumption-order table; ref(): from the
// reference-order table.  (Storing every coefficient twice, so that the packed type reads {c, c} as one 8-byte
// operand, removes the 170 v_mov per 4 pixels that broadcast odd halves -- and doubles the LDS reads, which the
// four SIMDs of a CU share: measured 8 % slower.)
template <class F>
struct PolyCoef;
template <>
struct PolyCoef<float> {
  static CURL_HD float seq(const float* c, int q) { return c[q]; }
  static CURL_HD float ref(float x) { return x; }
  // the first fma of a Horner chain: its initial value is coefficient QA itself (consumption-order table)
  template <int QA, int QB>
  static CURL_HD float fma_cc(const float* c, float v) { return poly_fmav(c[QA], v, c[QB]); }
  template <int QA>
  static CURL_HD float fmav_c(const float* c, float v, float t) { return poly_fmav(c[QA], v, t); }
};
#if defined(__HIP_DEVICE_COMPILE__)
template <>
struct PolyCoef<curl_f2> {
  static CURL_HD curl_f2 seq(const float* c, int q) { return splat2(c[q]); }
  static CURL_HD curl_f2 ref(float x) { return splat2(x); }
  // Coefficient q lives in half (q & 1) of the aligned 8-byte pair at c + (q & ~1).  v_pk_fma_f32 can take either
  // half of a source pair for BOTH result halves (op_sel / op_sel_hi); hipcc uses that for an addend but builds a
  // {c, c} pair with a v_mov when the coefficient is the multiplicand -- the chain's first fma, 160 per 4 pixels.
  // Written out here, the select costs nothing.
  // (read as half of the aligned 16-byte group, so that the reads stay ds_read_b128: polynomials start on 16-byte
  // boundaries of the LDS table, kSeqStride)
  static __device__ __forceinline__ curl_f2 pair(const float* c, int q) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 g = *reinterpret_cast<const f4*>(c + (q & ~3));
    return (q & 2) ? __builtin_shufflevector(g, g, 2, 3) : __builtin_shufflevector(g, g, 0, 1);
  }
  template <int QA, int QB>
  static __device__ __forceinline__ curl_f2 fma_cc(const float* c, curl_f2 v) {
    const curl_f2 a = pair(c, QA), b = pair(c, QB);
    curl_f2 r;
    if constexpr ((QA & 1) == 0 && (QB & 1) == 0)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(v), "v"(b));
    else if constexpr ((QA & 1) == 1 && (QB & 1) == 0)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(v), "v"(b));
    else if constexpr ((QA & 1) == 0 && (QB & 1) == 1)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(v), "v"(b));
    else
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(v), "v"(b));
    return r;
  }
  template <int QA>
  static __device__ __forceinline__ curl_f2 fmav_c(const float* c, curl_f2 v, curl_f2 t) {
    const curl_f2 a = pair(c, QA);
    curl_f2 r;
    if constexpr ((QA & 1) == 0)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(v), "v"(t));
    else
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(v), "v"(t));
    return r;
  }
};
#endif
#define CURL_POLY_SPLAT(c) (c)
#define CURL_POLY_FMA(a, v, c) poly_fmav(a, v, c)
#define CURL_POLY_FMAV(a, v, q) poly_fmav(a, v, q)
#include "poly_horner.inc"
#undef CURL_POLY_SPLAT
#undef CURL_POLY_FMA
#undef CURL_POLY_FMAV

template <int V>
struct PolyEval;
template <>
struct PolyEval<5> {
  static constexpr int kCoeffs = 126;
  static constexpr int kSeqStride = 128;  // floats per polynomial in the consumption-order (LDS) layout: 16-byte multiple
  template <class F, bool SEQ, int NP>
  static CURL_HD void eval(F (&out)[NP], const F (&v)[NP][5], const float* c) { poly_d4_v5<F, SEQ, NP>(out, v, c); }
  static CURL_HD int order(int pos) { return kPolyOrder_d4_v5[pos]; }
  static constexpr int kChunk = 42, kChunks = 3;
  template <int C>
  static CURL_HD void monomials(float (&m)[42], const float (&pw)[5][5]) { mono_d4_v5<C>(m, pw); }
};
template <>
struct PolyEval<4> {  // 3 colour channels + x/W: the per-row collapsed form of the 5-variable polynomial (below)
  static constexpr int kCoeffs = 70;
  static constexpr int kSeqStride = 72;
  template <class F, bool SEQ, int NP>
  static CURL_HD void eval(F (&out)[NP], const F (&v)[NP][4], const float* c) { poly_d4_v4<F, SEQ, NP>(out, v, c); }
  static CURL_HD int order(int pos) { return kPolyOrder_d4_v4[pos]; }
  static constexpr int kChunk = 35, kChunks = 2;  // the x-folded coefficient gradient (coef_grad_accumulate_foldx)
  template <int C>
  static CURL_HD void monomials(float (&m)[35], const float (&pw)[4][5]) { mono_d4_v4<C>(m, pw); }
};
template <>
struct PolyEval<3> {
  static constexpr int kCoeffs = 35;
  static constexpr int kSeqStride = 36;
  template <class F, bool SEQ, int NP>
  static CURL_HD void eval(F (&out)[NP], const F (&v)[NP][3], const float* c) { poly_d4_v3<F, SEQ, NP>(out, v, c); }
  static CURL_HD int order(int pos) { return kPolyOrder_d4_v3[pos]; }
  static constexpr int kChunk = 35, kChunks = 1;
  template <int C>
  static CURL_HD void monomials(float (&m)[35], const float (&pw)[3][5]) { mono_d4_v3<C>(m, pw); }
};

// out[o][i] = P_o(vars[.][i]) for the N pixels of a lane; coef = [3][NC] of one image and one space, in the
// reference's order (SEQ = false) or permuted into Horner consumption order (SEQ = true).  vars is plane-major.
template <int V, int N, bool SEQ = false>
CURL_HD void poly3_n(float (&out)[3][N], const float (&vars)[V][N], const float* coef) {
  // SEQ layout: every polynomial starts on a 16-byte boundary (kSeqStride), so the sequential reads merge into
  // ds_read_b64 / b128 with 16-bit offsets instead of ds_read2_b32 pairs that need a fresh base register every 1 KB
  constexpr int NC = SEQ ? PolyEval<V>::kSeqStride : PolyEval<V>::kCoeffs;
#if defined(__HIP_DEVICE_COMPILE__)
  // One output polynomial at a time, all pixel pairs of the lane in lock step (poly_horner.inc): the pairs share
  // every coefficient read (one broadcast ds_read_b128 feeds 4 terms x N/2 packed FMAs) and are the ILP of the
  // Horner chain; fences inside the generated code bound how far ahead the reads are hoisted (unfenced, hipcc
  // hoisted them all, needed > 256 VGPRs and spilled).
  constexpr int NP = N / 2;
  if constexpr (NP > 0) {
    curl_f2 v[NP][V], r[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int k = 0; k < V; ++k) {
        v[q][k].x = vars[k][2 * q];
        v[q][k].y = vars[k][2 * q + 1];
      }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      CURL_FENCE();
      PolyEval<V>::template eval<curl_f2, SEQ, NP>(r, v, coef + o * NC);
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        out[o][2 * q] = r[q].x;
        out[o][2 * q + 1] = r[q].y;
      }
    }
    CURL_FENCE();
  }
  if (N & 1) {
    float vs[1][V], rs[1];
#pragma unroll
    for (int k = 0; k < V; ++k) vs[0][k] = vars[k][N - 1];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      PolyEval<V>::template eval<float, SEQ, 1>(rs, vs, coef + o * NC);
      out[o][N - 1] = rs[0];
    }
  }
#else
  for (int i = 0; i < N; ++i) {
    float v[1][V], r[1];
    for (int k = 0; k < V; ++k) v[0][k] = vars[k][i];
    for (int o = 0; o < 3; ++o) {
      PolyEval<V>::template eval<float, SEQ, 1>(r, v, coef + o * NC);
      out[o][i] = r[0];
    }
  }
#endif
}

// torch.sigmoid over M values: 1 / (1 + exp(-x)); exps and reciprocals in runs
template <int M>
CURL_HD void sigmoid_run(float (&x)[M]) {
  scale_run(x, x, (float)(-1.4426950408889634));  // -log2(e)
  CURL_FENCE();
  CURL_TRANS_BEGIN();
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] = hw_exp2(x[i]);
  CURL_FENCE();
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] = 1.0f + x[i];
  CURL_FENCE();
#pragma unroll
  for (int i = 0; i < M; ++i) x[i] = hw_rcp(x[i]);  // 1 ulp: 6e-8 of a value in (0,1)
  CURL_TRANS_END();
  CURL_FENCE();
}

// TriSpaceRegNet.generate_residual (model.py:499-515) [+ generate_image (model.py:517-520) unless residual_only]
// for the N pixels of a lane.  p: RGB in, result out.  xw, yh: x/width and y/height of each pixel
// (cat_coords, model.py:487-497; unused when V == 3).  coef: [3 spaces = R, L, H][3][NC] of this image.
template <int V, int N, bool SEQ = false>
CURL_HD void trispace_n(PxN<N>& p, const float (&xw)[N], const float (&yh)[N], const float* coef, bool residual_only) {
  constexpr int NC = SEQ ? PolyEval<V>::kSeqStride : PolyEval<V>::kCoeffs;
  PxN<N> lab = p, hsv = p;
  rgb2lab_n<N, kPolySel>(lab);
  rgb2hsv_n<N>(hsv);
  float vars[V][N], o[3][N];
  float res[3][N];
  auto fill = [&](const PxN<N>& q) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      vars[0][i] = q.c0[i];
      vars[1][i] = q.c1[i];
      vars[2][i] = q.c2[i];
      if constexpr (V >= 4) vars[3][i] = xw[i];
      if constexpr (V == 5) vars[4][i] = yh[i];
    }
  };
  auto squash = [&]() {  // sigmoid over the 3N outputs
    float flat[3 * N];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) flat[c * N + i] = o[c][i];
    sigmoid_run(flat);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) o[c][i] = flat[c * N + i];
  };
  // RGB space: rgb_res = sigmoid(poly(cat(rgb, x, y), R)); 2 * (rgb_res - 0.5)
  fill(p);
  poly3_n<V, N, SEQ>(o, vars, coef);
  squash();
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) res[c][i] = 2.0f * (o[c][i] - 0.5f);
  // Lab space: lab2rgb(sigmoid(poly(cat(lab, x, y), L)))
  fill(lab);
  poly3_n<V, N, SEQ>(o, vars, coef + 3 * NC);
  squash();
  {
    PxN<N> q;
#pragma unroll
    for (int i = 0; i < N; ++i) q.c0[i] = o[0][i], q.c1[i] = o[1][i], q.c2[i] = o[2][i];
    lab2rgb_n<N, false, kPolySel>(q);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      res[0][i] += 2.0f * (q.c0[i] - 0.5f);
      res[1][i] += 2.0f * (q.c1[i] - 0.5f);
      res[2][i] += 2.0f * (q.c2[i] - 0.5f);
    }
  }
  // HSV space: hsv2rgb(sigmoid(poly(cat(hsv, x, y), H)))
  fill(hsv);
  poly3_n<V, N, SEQ>(o, vars, coef + 6 * NC);
  squash();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    Px q = hsv2rgb<true>(Px{o[0][i], o[1][i], o[2][i]});  // sigmoid outputs: already in [0,1]
    res[0][i] += 2.0f * (q.c0 - 0.5f);
    res[1][i] += 2.0f * (q.c1 - 0.5f);
    res[2][i] += 2.0f * (q.c2 - 0.5f);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (residual_only) {  // is_train=False: final_op returns the residual (model.py:485)
      p.c0[i] = res[0][i], p.c1[i] = res[1][i], p.c2[i] = res[2][i];
    } else {  // generate_image: clamp(img + residual, 0, 1)
      p.c0[i] = clamp01(p.c0[i] + res[0][i]);
      p.c1[i] = clamp01(p.c1[i] + res[1][i]);
      p.c2[i] = clamp01(p.c2[i] + res[2][i]);
    }
  }
}

// ---------------------------------------------------------------- one pixel row at a time: y is a constant
// cat_coords' last variable is row/height: the same value for every pixel of a row.  Writing
//   P(c0, c1, c2, x, y) = sum_m m(c0, c1, c2, x) * (sum_j y^j coef[(m, j)])
// the inner sums are 70 numbers per polynomial and row (Horner in y, <= 4 FMAs each), after which a pixel costs
// 69 FMAs per output instead of 125.  collapse_coef returns the collapsed coefficient the 4-variable Horner
// scheme consumes at position `pos`; c126 = the reference-order coefficients of one polynomial.
CURL_HD float collapse_coef(const float* c126, int pos, float y) {
  const unsigned short* src = kPolyCollapse_d4_v5[kPolyOrder_d4_v4[pos]];
  float acc = 0.0f;
  bool started = false;
#pragma unroll
  for (int j = 4; j >= 0; --j) {
    if (src[j] == 0xFFFF) continue;
    acc = started ? fmaf(acc, y, c126[src[j]]) : c126[src[j]];
    started = true;
  }
  return acc;
}

// The same on the device, from one 16-byte row of the position-indexed table (kPolyFold_d4_v5: one load instead of a
// chain of dependent 2-byte ones) and coefficients that already sit in LDS.  The valid entries are j = 0..4-deg(m);
// starting the Horner chain from 0 makes the first step fma(0, y, c) = c, the value collapse_coef starts from.
CURL_HD float collapse_coef_fold(const float* c126, int pos, float y) {
  const PolyFoldRow e = kPolyFold_d4_v5[pos];
  float acc = 0.0f;
#pragma unroll
  for (int j = 4; j >= 0; --j) {
    const unsigned idx = e.j[j];
    if (idx != 0xFFFFu) acc = fmaf(acc, y, c126[idx]);
  }
  return acc;
}

// ---------------------------------------------------------------- backward of the polynomial path
// Training the fork's live model needs d loss / d coeffs only (the image is data).  Per pixel and space s:
//   y_s = conv_s(sigmoid(P_s(vars_s))),  residual = sum_s 2 (y_s - 0.5),  out = clamp(img + residual)  [or residual]
//   g_P[s][o] = conv_s'(sigma)^T (2 g_res) * sigma (1 - sigma),   d coeffs[s][o][t] += g_P[s][o] * m_t(vars_s)
// Pass 1 (this function) emits, per pixel, the 9 g_P values and the 9 colour variables (coordinates are
// recomputed); pass 2 accumulates the outer products with the monomials (coef_grad_accumulate).
template <int V, int N, bool SEQ>
CURL_HD void trispace_bwd_n(const PxN<N>& in, const float (&xw)[N], const float (&yh)[N], const float* coef,
                            const PxN<N>& gout, bool residual_only, float (&vars)[3][3][N], float (&gP)[3][3][N]) {
  constexpr int NC = SEQ ? PolyEval<V>::kSeqStride : PolyEval<V>::kCoeffs;
  PxN<N> sp[3] = {in, in, in};
  rgb2lab_n<N>(sp[1]);
  rgb2hsv_n<N>(sp[2]);
  float sig[3][3][N];
#pragma unroll
  for (int s = 0; s < 3; ++s) {  // forward recompute: the N pixels share the coefficient reads (poly3_n)
    float v[V][N], o[3][N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      vars[s][0][i] = v[0][i] = sp[s].c0[i];
      vars[s][1][i] = v[1][i] = sp[s].c1[i];
      vars[s][2][i] = v[2][i] = sp[s].c2[i];
      if constexpr (V >= 4) v[3][i] = xw[i];
      if constexpr (V == 5) v[4][i] = yh[i];
    }
    poly3_n<V, N, SEQ>(o, v, coef + s * 3 * NC);
    float flat[3 * N];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) flat[c * N + i] = o[c][i];
    sigmoid_run(flat);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) sig[s][c][i] = flat[c * N + i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {  // reverse mode through the converters, one pixel at a time
    Px s1{sig[1][0][i], sig[1][1][i], sig[1][2][i]}, s2{sig[2][0][i], sig[2][1][i], sig[2][2][i]};
    Px g_res{gout.c0[i], gout.c1[i], gout.c2[i]};
    // the taped converters: their forward values serve generate_image's clamp gate, their tapes the pullbacks (one evaluation
    // each; until round 4 lab2rgb / hsv2rgb ran here and again inside lab2rgb_bwd / hsv2rgb_bwd)
    Lab2RgbT t1;
    Hsv2RgbT t2;
    const Px y1 = lab2rgb_t(s1, t1), y2 = hsv2rgb_t<false>(s2, t2);
    if (!residual_only) {  // generate_image: clamp(img + residual, 0, 1)
      float r0 = 2.0f * (sig[0][0][i] - 0.5f) + 2.0f * (y1.c0 - 0.5f) + 2.0f * (y2.c0 - 0.5f);
      float r1 = 2.0f * (sig[0][1][i] - 0.5f) + 2.0f * (y1.c1 - 0.5f) + 2.0f * (y2.c1 - 0.5f);
      float r2 = 2.0f * (sig[0][2][i] - 0.5f) + 2.0f * (y1.c2 - 0.5f) + 2.0f * (y2.c2 - 0.5f);
      g_res = Px{g_res.c0 * pass01(in.c0[i] + r0), g_res.c1 * pass01(in.c1[i] + r1), g_res.c2 * pass01(in.c2[i] + r2)};
    }
    Px gy{2.0f * g_res.c0, 2.0f * g_res.c1, 2.0f * g_res.c2};
    Px gs[3] = {gy, lab2rgb_pull(t1, gy), hsv2rgb_pull<false>(t2, gy)};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      gP[s][0][i] = gs[s].c0 * sig[s][0][i] * (1.0f - sig[s][0][i]);
      gP[s][1][i] = gs[s].c1 * sig[s][1][i] * (1.0f - sig[s][1][i]);
      gP[s][2][i] = gs[s].c2 * sig[s][2][i] * (1.0f - sig[s][2][i]);
    }
  }
}

template <int V, bool SEQ>
CURL_HD void trispace_bwd_px(Px in, float xw, float yh, const float* coef, Px gout, bool residual_only,
                             float (&vars)[3][3], float (&gP)[3][3]) {
  PxN<1> i1{{in.c0}, {in.c1}, {in.c2}}, g1{{gout.c0}, {gout.c1}, {gout.c2}};
  float x1[1] = {xw}, y1[1] = {yh}, v[3][3][1], g[3][3][1];
  trispace_bwd_n<V, 1, SEQ>(i1, x1, y1, coef, g1, residual_only, v, g);
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int c = 0; c < 3; ++c) vars[s][c] = v[s][c][0], gP[s][c] = g[s][c][0];
}

// Accumulator PAIRS for the coefficient gradients: on gfx950 one v_pk_fma_f32 per pair with the pixel's g as a broadcast
// operand, the same monomial pairs for the three outputs.  (Left to hipcc's vectoriser over 3 x 35 floats, the second
// output's pairs straddled the first's and the monomial products were packed too: 70 moves and shuffles per step next to 51
// packed FMAs.)  NPAIR pairs hold 2 * NPAIR >= T monomials; the last half of an odd T is padding.
#if defined(__HIP_DEVICE_COMPILE__)
typedef curl_f2 grad_pair;
#else
struct grad_pair {
  float x, y;
};
#endif
template <int T, int NPAIR>
CURL_HD void grad_pairs_accumulate(grad_pair (&acc)[3][NPAIR], float (&m)[2 * NPAIR], const float (&gP)[3]) {
  static_assert(2 * NPAIR >= T && 2 * NPAIR - T <= 1, "pairs cover the monomials");
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int j = 0; j < T; ++j) asm volatile("" : "+v"(m[j]));  // scalar products, each in a register of its own choosing
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const curl_f2 g = {gP[o], gP[o]};
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const curl_f2 mk = {m[2 * k], m[2 * k + 1]};
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[o][k]) : "v"(g), "v"(mk));
    }
  }
#else
  for (int o = 0; o < 3; ++o)
    for (int k = 0; k < NPAIR; ++k) {
      acc[o][k].x = fmaf(gP[o], m[2 * k], acc[o][k].x);
      acc[o][k].y = fmaf(gP[o], m[2 * k + 1], acc[o][k].y);
    }
#endif
}
CURL_HD float grad_pair_get(const grad_pair* acc, int j) { return (j & 1) ? acc[j >> 1].y : acc[j >> 1].x; }

// acc[o][j] += gP[o] * m_{C*chunk + j}(v) for one pixel and one chunk of the monomials
template <int V, int C>
CURL_HD void coef_grad_accumulate(grad_pair (&acc)[3][(PolyEval<V>::kChunk + 1) / 2], const float (&v)[V], const float (&gP)[3]) {
  constexpr int T = PolyEval<V>::kChunk, NPAIR = (T + 1) / 2;
  float pw[V][5];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    pw[k][0] = 1.0f;
    pw[k][1] = v[k];
    pw[k][2] = v[k] * v[k];
    pw[k][3] = pw[k][2] * v[k];
    pw[k][4] = pw[k][2] * pw[k][2];
  }
  float m[2 * NPAIR];
  {
    float mt[T];
    PolyEval<V>::template monomials<C>(mt, pw);
#pragma unroll
    for (int j = 0; j < 2 * NPAIR; ++j) m[j] = j < T ? mt[j] : 0.0f;
  }
  grad_pairs_accumulate<T, NPAIR>(acc, m, gP);
}

// The same for the spatial polynomial with the COLUMN coordinate folded out (PolyFoldX, poly_horner.inc): a thread that
// walks down one image column has one x, so it accumulates over the 70 monomials of (c0, c1, c2, y) -- chunk C of 35 --
// and expands by the powers of x once at the end (coef_grad_expand_foldx): 35 monomials + 105 FMAs per pixel and chunk,
// two chunks, instead of 42 + 126 and three.
// The accumulators are pairs (grad_pairs_accumulate above: 35 monomials padded to 36).
constexpr int kFoldXPairs = 18;
typedef grad_pair foldx_pair;
template <int C>
CURL_HD void coef_grad_accumulate_foldx(foldx_pair (&acc)[3][kFoldXPairs], const float (&c)[3], float y, const float (&gP)[3]) {
  const float v[4] = {c[0], c[1], c[2], y};
  float pw[4][5];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pw[k][0] = 1.0f;
    pw[k][1] = v[k];
    pw[k][2] = v[k] * v[k];
    pw[k][3] = pw[k][2] * v[k];
    pw[k][4] = pw[k][2] * pw[k][2];
  }
  float m[2 * kFoldXPairs];
  {
    float m35[35];
    PolyEval<4>::template monomials<C>(m35, pw);
#pragma unroll
    for (int j = 0; j < 2 * kFoldXPairs; ++j) m[j] = j < 35 ? m35[j] : 0.0f;
  }
  grad_pairs_accumulate<35, kFoldXPairs>(acc, m, gP);
}
// e[i] = x^j_i * acc[m'_i] for slice S of chunk C's (monomial, x power) pairs; kPolyFoldXIndex[C][S][i] names the
// reference coefficient each one is the gradient of
template <int C, int S>
CURL_HD void coef_grad_expand_foldx(float (&e)[PolyFoldX<C>::kSlice], const foldx_pair (&acc)[kFoldXPairs], float x) {
  const float x2 = x * x;
  const float xp[5] = {1.0f, x, x2, x2 * x, x2 * x2};
  float a[35];
#pragma unroll
  for (int j = 0; j < 35; ++j) a[j] = (j & 1) ? acc[j >> 1].y : acc[j >> 1].x;
  foldx_expand<C, S>(e, a, xp);
}

}  // namespace curlm
