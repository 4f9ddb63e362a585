// curl_twin.cpp -- TEST-ONLY host twin of the kernel arithmetic.
//
// Compiles curl_amd/csrc/curl_math.h (the exact header the gfx950 kernels include) with g++, libm
// standing in for v_log_f32 / v_exp_f32 / v_rcp_f32, and loops it over host arrays.  tests/ use it in
// the CPU container to check the kernels' algebra against the oracle and the golden vectors before
// anything goes to the GPU box.  The product (curl_amd/) never loads this library.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../curl_amd/csrc/curl_math_bwd.h"
#include "../../curl_amd/csrc/curl_math_poly.h"
#include "../../curl_amd/csrc/curl_math_loss.h"

using namespace curlm;

static void prep(const float* raw, int ncurves, int K, Affine* k, float* knots, float& reg) {
  reg = 0.0f;
  for (int c = 0; c < ncurves; ++c) {
    float* C = knots + c * K;
    for (int j = 0; j < K; ++j) C[j] = (float)std::exp((double)raw[c * K + j]);
    float r;
    collapse_curve(C, K, k[c].a, k[c].b, r);
    reg += r;
  }
}

extern "C" {

// op: 0 rgb2lab, 1 lab2rgb, 2 rgb2hsv, 3 hsv2rgb
int twin_convert(int op, const float* in, float* out, int B, long HW) {
  for (int b = 0; b < B; ++b)
    for (long i = 0; i < HW; ++i) {
      const float* p = in + (size_t)b * 3 * HW + i;
      Px x{p[0], p[HW], p[2 * HW]}, y;
      switch (op) {
        case 0: y = rgb2lab(x); break;
        case 1: y = lab2rgb(x); break;
        case 2: y = rgb2hsv(x); break;
        default: y = hsv2rgb(x); break;
      }
      float* q = out + (size_t)b * 3 * HW + i;
      q[0] = y.c0, q[HW] = y.c1, q[2 * HW] = y.c2;
    }
  return 0;
}

// mode: 0 affine, 1 exact order, 2 pwl.  C already exp'd, [B,K].  reg (nullable) +=.
int twin_apply_curve(const float* img, const float* C, float* out, float* reg, int B, long HW, int K, int cin,
                     int cout, int mode) {
  std::vector<float> sl(K);
  for (int b = 0; b < B; ++b) {
    const float* c = C + (size_t)b * K;
    for (int j = 0; j + 1 < K; ++j) sl[j] = c[j + 1] - c[j];
    float a, bb, r;
    collapse_curve(c, K, a, bb, r);
    if (reg) reg[b] += r;
    for (long i = 0; i < HW; ++i) {
      const float* p = img + (size_t)b * 3 * HW + i;
      float ch[3] = {p[0], p[HW], p[2 * HW]};
      float x = ch[cin];
      float s = mode == 1 ? scale_exact(x, sl.data(), c[0], K - 2, (float)(K - 1))
                          : mode == 2 ? scale_pwl(x, c, sl.data(), K) : fmaf(bb, x, a);
      ch[cout] *= s;
      float* q = out + (size_t)b * 3 * HW + i;
      q[0] = clamp01(ch[0]), q[HW] = clamp01(ch[1]), q[2 * HW] = clamp01(ch[2]);
    }
  }
  return 0;
}

// ncurves 3 (rgb/lab) or 4 (hsv); raw [B, ncurves*K]
int twin_adjust(int ncurves, const float* img, const float* raw, float* out, float* reg, int B, long HW, int K,
                int mode) {
  const int cin4[4] = {0, 0, 1, 2}, cout4[4] = {0, 1, 1, 2};
  std::vector<float> knots(ncurves * K), sl(K);
  for (int b = 0; b < B; ++b) {
    Affine k[4];
    float r;
    prep(raw + (size_t)b * ncurves * K, ncurves, K, k, knots.data(), r);
    if (reg) reg[b] = r;
    for (long i = 0; i < HW; ++i) {
      const float* p = img + (size_t)b * 3 * HW + i;
      Px x{p[0], p[HW], p[2 * HW]}, y;
      if (mode == 0) {
        y = ncurves == 3 ? adjust3(x, k[0], k[1], k[2]) : adjust_hsv4(x, k[0], k[1], k[2], k[3]);
      } else {
        float ch[3] = {x.c0, x.c1, x.c2};
        for (int s = 0; s < ncurves; ++s) {
          const float* c = knots.data() + s * K;
          for (int j = 0; j + 1 < K; ++j) sl[j] = c[j + 1] - c[j];
          int ci = ncurves == 3 ? s : cin4[s], co = ncurves == 3 ? s : cout4[s];
          float sc = mode == 1 ? scale_exact(ch[ci], sl.data(), c[0], K - 2, (float)(K - 1))
                               : scale_pwl(ch[ci], c, sl.data(), K);
          ch[co] *= sc;
          for (int e = 0; e < 3; ++e) ch[e] = clamp01(ch[e]);
        }
        y = Px{ch[0], ch[1], ch[2]};
      }
      float* q = out + (size_t)b * 3 * HW + i;
      q[0] = y.c0, q[HW] = y.c1, q[2 * HW] = y.c2;
    }
  }
  return 0;
}

// stage 0: lab_stage (rawR/rawH ignored), 1: full layer, 2: hsv_stage (rawL/rawR ignored).  mask: float [B,HW] or NULL.
// binary != 0: run the binary-mask specialisation (mask values must be 0 or 1), incl. the masked-out shortcut.
int twin_layer(int stage, const float* img, const float* mask, const float* rawL, const float* rawR,
               const float* rawH, float* out, float* reg, int B, long HW, int Kl, int Kr, int Kh, int binary) {
  std::vector<float> knots(4 * 256);
  for (int b = 0; b < B; ++b) {
    LayerCoef k;
    float rl = 0, rr = 0, rh = 0;
    if (stage != 2) prep(rawL + (size_t)b * 3 * Kl, 3, Kl, k.lab, knots.data(), rl);
    if (stage == 1) prep(rawR + (size_t)b * 3 * Kr, 3, Kr, k.rgb, knots.data(), rr);
    if (stage >= 1) prep(rawH + (size_t)b * 4 * Kh, 4, Kh, k.hsv, knots.data(), rh);
    if (reg) reg[b] = (rl + rr) + rh;
    for (long i = 0; i < HW; ++i) {
      const float* p = img + (size_t)b * 3 * HW + i;
      float m = mask ? mask[(size_t)b * HW + i] : 1.0f;
      Px x{p[0], p[HW], p[2 * HW]};
      Px y;
      if (binary && m == 0.0f)
        y = stage == 0 ? lab_stage_masked_out() : Px{0.0f, 0.0f, 0.0f};
      else if (binary)
        y = stage == 0 ? lab_stage<true>(x, m, k.lab) : stage == 2 ? hsv_stage<true>(x, m, k.hsv) : curl_layer<true>(x, m, k);
      else
        y = stage == 0 ? lab_stage<false>(x, m, k.lab) : stage == 2 ? hsv_stage<false>(x, m, k.hsv) : curl_layer<false>(x, m, k);
      float* q = out + (size_t)b * 3 * HW + i;
      q[0] = y.c0, q[HW] = y.c1, q[2 * HW] = y.c2;
    }
  }
  return 0;
}

// Backward of twin_layer(stage=1): d loss/d img, d loss/d raw knots, given d loss/d out and d loss/d reg.
int twin_layer_bwd(const float* img, const float* mask, const float* rawL, const float* rawR, const float* rawH,
                   const float* gout, const float* greg, float* gimg, float* gL, float* gR, float* gH, int B, long HW,
                   int Kl, int Kr, int Kh, int binary) {
  std::vector<float> kl(3 * Kl), kr(3 * Kr), kh(4 * Kh);
  for (int b = 0; b < B; ++b) {
    LayerCoef k;
    float r;
    prep(rawL + (size_t)b * 3 * Kl, 3, Kl, k.lab, kl.data(), r);
    prep(rawR + (size_t)b * 3 * Kr, 3, Kr, k.rgb, kr.data(), r);
    prep(rawH + (size_t)b * 4 * Kh, 4, Kh, k.hsv, kh.data(), r);
    double P[10] = {0}, Q[10] = {0};
    for (long i = 0; i < HW; ++i) {
      const float* p = img + (size_t)b * 3 * HW + i;
      const float* g = gout + (size_t)b * 3 * HW + i;
      float m = mask ? mask[(size_t)b * HW + i] : 1.0f;
      float Pf[10] = {0}, Qf[10] = {0};
      Px gi = binary ? curl_layer_bwd<true>(Px{p[0], p[HW], p[2 * HW]}, m, k, Px{g[0], g[HW], g[2 * HW]}, Pf, Qf)
                     : curl_layer_bwd<false>(Px{p[0], p[HW], p[2 * HW]}, m, k, Px{g[0], g[HW], g[2 * HW]}, Pf, Qf);
      for (int c = 0; c < 10; ++c) P[c] += Pf[c], Q[c] += Qf[c];
      if (gimg) {
        float* q = gimg + (size_t)b * 3 * HW + i;
        q[0] = gi.c0, q[HW] = gi.c1, q[2 * HW] = gi.c2;
      }
    }
    double gr = greg ? (double)greg[b] : 0.0;
    for (int c = 0; c < 3; ++c) knots_bwd(kl.data() + c * Kl, Kl, P[c], Q[c], gr, gL + (size_t)b * 3 * Kl + c * Kl);
    for (int c = 0; c < 3; ++c) knots_bwd(kr.data() + c * Kr, Kr, P[3 + c], Q[3 + c], gr, gR + (size_t)b * 3 * Kr + c * Kr);
    for (int c = 0; c < 4; ++c) knots_bwd(kh.data() + c * Kh, Kh, P[6 + c], Q[6 + c], gr, gH + (size_t)b * 4 * Kh + c * Kh);
  }
  return 0;
}

// TriSpaceRegNet per-pixel path. coeffs [B,3,3,NC] (R,L,H); V = 5 (spatial) or 3.
int twin_trispace(const float* img, const float* coeffs, float* out, int B, int H, int W, int V, int residual_only) {
  const long HW = (long)H * W;
  const int NC = V == 5 ? 126 : 35;
  for (int b = 0; b < B; ++b)
    for (long i = 0; i < HW; ++i) {
      const float* p = img + (size_t)b * 3 * HW + i;
      PxN<1> q{{p[0]}, {p[HW]}, {p[2 * HW]}};
      const float xw[1] = {(float)(i % W) / (float)W}, yh[1] = {(float)(i / W) / (float)H};
      if (V == 5)
        trispace_n<5, 1>(q, xw, yh, coeffs + (size_t)b * 9 * NC, residual_only != 0);
      else
        trispace_n<3, 1>(q, xw, yh, coeffs + (size_t)b * 9 * NC, residual_only != 0);
      float* o = out + (size_t)b * 3 * HW + i;
      o[0] = q.c0[0], o[HW] = q.c1[0], o[2 * HW] = q.c2[0];
    }
  return 0;
}
// ChannelPolyLayer / Deg4MobilePolyLayer forward: img [B,V,H,W], coeffs [B,3,NC] -> out [B,3,H,W]
// the per-row collapsed evaluation of the spatial polynomial path (what trispace_rows_kernel does)
int twin_trispace_rows(const float* img, const float* coeffs, float* out, int B, int H, int W, int residual_only) {
  const size_t HW = (size_t)H * W;
  for (int b = 0; b < B; ++b) {
    const float* c = coeffs + (size_t)b * 9 * 126;
    for (int r = 0; r < H; ++r) {
      constexpr int NS4 = PolyEval<4>::kSeqStride;
      float row_coef[9 * NS4] = {};
      const float y = (float)r / (float)H;
      for (int q = 0; q < 9; ++q)
        for (int pos = 0; pos < 70; ++pos) row_coef[q * NS4 + pos] = collapse_coef(c + q * 126, pos, y);
      for (int col = 0; col < W; ++col) {
        size_t i = (size_t)r * W + col;
        PxN<1> p{{img[(b * 3 + 0) * HW + i]}, {img[(b * 3 + 1) * HW + i]}, {img[(b * 3 + 2) * HW + i]}};
        float xw[1] = {(float)col / (float)W}, yh[1] = {0.0f};
        trispace_n<4, 1, true>(p, xw, yh, row_coef, residual_only != 0);
        out[(b * 3 + 0) * HW + i] = p.c0[0], out[(b * 3 + 1) * HW + i] = p.c1[0], out[(b * 3 + 2) * HW + i] = p.c2[0];
      }
    }
  }
  return 0;
}

int twin_poly_layer(const float* img, const float* coeffs, float* out, int B, long HW, int V) {
  const int NC = V == 5 ? 126 : 35;
  for (int b = 0; b < B; ++b)
    for (long i = 0; i < HW; ++i) {
      float o[3][1];
      if (V == 5) {
        float v[5][1];
        for (int k = 0; k < 5; ++k) v[k][0] = img[((size_t)b * 5 + k) * HW + i];
        poly3_n<5, 1>(o, v, coeffs + (size_t)b * 3 * NC);
      } else {
        float v[3][1];
        for (int k = 0; k < 3; ++k) v[k][0] = img[((size_t)b * 3 + k) * HW + i];
        poly3_n<3, 1>(o, v, coeffs + (size_t)b * 3 * NC);
      }
      for (int c = 0; c < 3; ++c) out[((size_t)b * 3 + c) * HW + i] = o[c][0];
    }
  return 0;
}

// CURLLoss pointwise terms: sums [B][5] = (sum|p-t|, sum cos, sum|lab|, sum|cone|, sum mask); L planes optional.
int twin_loss_terms(const float* pred, const float* tgt, const float* mask, double* sums, float* Lp, float* Lt, int B,
                    long HW) {
  for (int b = 0; b < B; ++b) {
    double acc[5] = {0, 0, 0, 0, 0};
    long i = 0;
    for (; i + 4 <= HW; i += 4) {  // groups of four through the phase form the HIP kernel runs (loss_terms_n<4>)
      const float* p = pred + (size_t)b * 3 * HW + i;
      const float* t = tgt + (size_t)b * 3 * HW + i;
      curlm::PxN<4> pp, tt;
      float m[4], lp[4], lt[4], sum[4] = {0, 0, 0, 0};
      for (int e = 0; e < 4; ++e) {
        pp.c0[e] = p[e], pp.c1[e] = p[HW + e], pp.c2[e] = p[2 * HW + e];
        tt.c0[e] = t[e], tt.c1[e] = t[HW + e], tt.c2[e] = t[2 * HW + e];
        m[e] = mask ? mask[(size_t)b * HW + i + e] : 1.0f;
        acc[4] += m[e];
      }
      curlm::loss_terms_n<4>(pp, tt, m, sum, lp, lt);
      for (int k = 0; k < 4; ++k) acc[k] += sum[k];
      for (int e = 0; e < 4; ++e) {
        if (Lp) Lp[(size_t)b * HW + i + e] = lp[e];
        if (Lt) Lt[(size_t)b * HW + i + e] = lt[e];
      }
    }
    for (; i < HW; ++i) {
      const float* p = pred + (size_t)b * 3 * HW + i;
      const float* t = tgt + (size_t)b * 3 * HW + i;
      float m = mask ? mask[(size_t)b * HW + i] : 1.0f;
      LossPx o = loss_terms(Px{p[0], p[HW], p[2 * HW]}, Px{t[0], t[HW], t[2 * HW]}, m);
      acc[0] += o.rgb_l1, acc[1] += o.cos_sim, acc[2] += o.lab_l1, acc[3] += o.hsv_l1, acc[4] += m;
      if (Lp) Lp[(size_t)b * HW + i] = o.Lp;
      if (Lt) Lt[(size_t)b * HW + i] = o.Lt;
    }
    for (int k = 0; k < 5; ++k) sums[b * 5 + k] = acc[k];
  }
  return 0;
}
int twin_loss_terms_bwd(const float* pred, const float* tgt, const float* mask, const float* w4, const float* gLp,
                        float* gpred, int B, long HW) {
  const float w[4] = {w4[0], w4[1], w4[2], w4[3]};
  for (int b = 0; b < B; ++b)
    for (long i = 0; i < HW; ++i) {
      const float* p = pred + (size_t)b * 3 * HW + i;
      const float* t = tgt + (size_t)b * 3 * HW + i;
      float m = mask ? mask[(size_t)b * HW + i] : 1.0f;
      Px g = loss_terms_bwd(Px{p[0], p[HW], p[2 * HW]}, Px{t[0], t[HW], t[2 * HW]}, m, w, gLp ? gLp[(size_t)b * HW + i] : 0.0f);
      float* q = gpred + (size_t)b * 3 * HW + i;
      q[0] = g.c0, q[HW] = g.c1, q[2 * HW] = g.c2;
    }
  return 0;
}

}  // extern "C"

// d loss / d coeffs [B,3,3,NC] of twin_trispace, given gout = d loss / d out.
template <int V>
static void trispace_bwd_host(const float* img, const float* coeffs, const float* gout, float* gcoef, int B, int H, int W,
                              int residual_only) {
  constexpr int NC = PolyEval<V>::kCoeffs, T = PolyEval<V>::kChunk;
  const long HW = (long)H * W;
  for (int b = 0; b < B; ++b) {
    std::vector<double> acc(9 * NC, 0.0);
    for (long i = 0; i < HW; ++i) {
      const float* p = img + (size_t)b * 3 * HW + i;
      const float* g = gout + (size_t)b * 3 * HW + i;
      float xw = (float)(i % W) / (float)W, yh = (float)(i / W) / (float)H;
      float vars[3][3], gP[3][3];
      trispace_bwd_px<V, false>(Px{p[0], p[HW], p[2 * HW]}, xw, yh, coeffs + (size_t)b * 9 * NC, Px{g[0], g[HW], g[2 * HW]},
                                residual_only != 0, vars, gP);
      for (int s = 0; s < 3; ++s) {
        float v[V];
        v[0] = vars[s][0], v[1] = vars[s][1], v[2] = vars[s][2];
        if (V == 5) v[V - 2] = xw, v[V - 1] = yh;
        constexpr int NPAIR = (T + 1) / 2;
        curlm::grad_pair a[3][NPAIR];
        auto clear = [&]() {
          for (int o = 0; o < 3; ++o) for (int k = 0; k < NPAIR; ++k) a[o][k] = curlm::grad_pair{0.0f, 0.0f};
        };
        auto flush = [&](int c) {
          for (int o = 0; o < 3; ++o)
            for (int j = 0; j < T; ++j)
              if (c * T + j < NC) acc[(s * 3 + o) * NC + c * T + j] += curlm::grad_pair_get(a[o], j);
        };
        clear();
        coef_grad_accumulate<V, 0>(a, v, gP[s]);
        flush(0);
        if constexpr (PolyEval<V>::kChunks > 1) {
          clear();
          coef_grad_accumulate<V, 1>(a, v, gP[s]);
          flush(1);
          clear();
          coef_grad_accumulate<V, 2>(a, v, gP[s]);
          flush(2);
        }
      }
    }
    for (int k = 0; k < 9 * NC; ++k) gcoef[(size_t)b * 9 * NC + k] = (float)acc[k];
  }
}
// The spatial form's accumulation as the HIP kernel does it (trispace_coef_grad_strip_kernel): per image column, sums over
// the 70 monomials of (c0, c1, c2, y) down the rows, expanded by the powers of that column's x at the end.
template <int C, int S>
static void foldx_flush(const curlm::foldx_pair (&a)[curlm::kFoldXPairs], float x, double* dst) {
  if constexpr (S < curlm::PolyFoldX<C>::kSlices) {
    float e[curlm::PolyFoldX<C>::kSlice];
    curlm::coef_grad_expand_foldx<C, S>(e, a, x);
    for (int i = 0; i < curlm::PolyFoldX<C>::kSlice; ++i) {
      unsigned t = curlm::kPolyFoldXIndex[C][S][i];
      if (t != 0xFFFFu) dst[t] += e[i];
    }
    foldx_flush<C, S + 1>(a, x, dst);
  }
}
static void trispace_bwd_host_foldx(const float* img, const float* coeffs, const float* gout, float* gcoef, int B, int H, int W,
                                    int residual_only) {
  constexpr int NC = 126;
  const long HW = (long)H * W;
  for (int b = 0; b < B; ++b) {
    std::vector<double> acc(9 * NC, 0.0);
    for (int col = 0; col < W; ++col) {
      const float xw = (float)col / (float)W;
      curlm::foldx_pair a0[3][3][curlm::kFoldXPairs] = {}, a1[3][3][curlm::kFoldXPairs] = {};
      for (int row = 0; row < H; ++row) {
        const long i = (long)row * W + col;
        const float* p = img + (size_t)b * 3 * HW + i;
        const float* g = gout + (size_t)b * 3 * HW + i;
        const float yh = (float)row / (float)H;
        float vars[3][3], gP[3][3];
        trispace_bwd_px<5, false>(Px{p[0], p[HW], p[2 * HW]}, xw, yh, coeffs + (size_t)b * 9 * NC, Px{g[0], g[HW], g[2 * HW]},
                                  residual_only != 0, vars, gP);
        for (int s = 0; s < 3; ++s) {
          curlm::coef_grad_accumulate_foldx<0>(a0[s], vars[s], yh, gP[s]);
          curlm::coef_grad_accumulate_foldx<1>(a1[s], vars[s], yh, gP[s]);
        }
      }
      for (int s = 0; s < 3; ++s)
        for (int o = 0; o < 3; ++o) {
          foldx_flush<0, 0>(a0[s][o], xw, acc.data() + (s * 3 + o) * NC);
          foldx_flush<1, 0>(a1[s][o], xw, acc.data() + (s * 3 + o) * NC);
        }
    }
    for (int k = 0; k < 9 * NC; ++k) gcoef[(size_t)b * 9 * NC + k] = (float)acc[k];
  }
}
extern "C" {
int twin_trispace_bwd_foldx(const float* img, const float* coeffs, const float* gout, float* gcoef, int B, int H, int W,
                            int residual_only) {
  trispace_bwd_host_foldx(img, coeffs, gout, gcoef, B, H, W, residual_only);
  return 0;
}
int twin_trispace_bwd(const float* img, const float* coeffs, const float* gout, float* gcoef, int B, int H, int W, int V,
                      int residual_only) {
  if (V == 5) trispace_bwd_host<5>(img, coeffs, gout, gcoef, B, H, W, residual_only);
  else trispace_bwd_host<3>(img, coeffs, gout, gcoef, B, H, W, residual_only);
  return 0;
}
// masked PSNR's sums for one image (psnr.inc's per-pixel term in the kernel's float32; the kernel's block partials are summed in
// float64, here everything is): sse = sum over 3 HW of (clamp(a) m - clamp(b) m)^2, msum = sum of m
int twin_psnr_sums(const float* a, const float* b, const float* mask, long HW, double* sse, double* msum) {
  double s = 0.0, ms = 0.0;
  for (long i = 0; i < HW; ++i) {
    const float m = mask ? mask[i] : 1.0f;
    for (int c = 0; c < 3; ++c) s += (double)curlm::psnr_sq_err(a[c * HW + i], b[c * HW + i], m, 0.0f);
    ms += m;
  }
  *sse = s, *msum = ms;
  return 0;
}
// file-edge scalars: out[b] = u8_to_unit(b) for b = 0..255; q[i] = unit_to_u8(x[i])
int twin_u8_edges(float* unit256, const float* x, unsigned char* q, long n) {
  for (int b = 0; b < 256; ++b) unit256[b] = curlm::u8_to_unit((float)b);
  for (long i = 0; i < n; ++i) q[i] = (unsigned char)curlm::unit_to_u8(x[i]);
  return 0;
}
// number of (n, d) with 0 <= n < d <= dmax where div_small differs from the IEEE division
long twin_div_small_mismatches(int dmax) {
  long bad = 0;
  for (int d = 1; d <= dmax; ++d) {
    const float fd = (float)d, rd = 1.0f / fd;
    for (int n = 0; n < d; ++n) bad += curlm::div_small((float)n, fd, rd) != (float)n / fd;
  }
  return bad;
}
}
