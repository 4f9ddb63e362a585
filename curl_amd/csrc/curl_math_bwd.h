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
// Same dual compilation as curl_math.h (device: gfx950 kernels; host: the test-only twin).
#pragma once
#include "curl_math.h"

namespace curlm {

// torch.clamp's gradient gate, lo <= x <= hi (boundaries included), as a bit mask built from sign bits like the
// forward's selects (no v_cmp / v_cndmask): x - lo and hi - x are both non-negative exactly inside the range.
// "+ 0.0f" turns a -0 difference (x = -0, lo = 0) into +0 so that the boundary passes, as it does in torch.
CURL_HD int range_mask(float x, float lo, float hi) { return ~(neg_mask((x - lo) + 0.0f) | neg_mask(hi - x)); }
CURL_HD int mask01(float x) { return range_mask(x, 0.0f, 1.0f); }
CURL_HD float gate01(float g, float pre) {  // g * [0 <= pre <= 1]: two subtractions, two shifts, one bitop3
  return drop_if2(neg_mask(pre + 0.0f), neg_mask(1.0f - pre), g);
}
CURL_HD float pass01(float pre) { return keep_if(mask01(pre), 1.0f); }
CURL_HD float pass_range(float x, float lo, float hi) { return keep_if(range_mask(x, lo, hi), 1.0f); }
CURL_HD int ge_mask(float a, float b) { return ~neg_mask(a - b); }  // a >= b (finite a, b)

// y = clamp01(xo * (a + b*xi)), xi != xo.  Given gy: accumulates P, Q; returns g_xo, adds to g_xi.
CURL_HD void curve_bwd_cross(float xo, float xi, Affine k, float gy, float& g_xo, float& g_xi, float& P, float& Q) {
  float s = fmaf(k.b, xi, k.a);
  float g_pre = gate01(gy, xo * s);
  float g_s = g_pre * xo;
  g_xo = g_pre * s;
  g_xi += g_s * k.b;
  P += g_s;
  Q += g_s * xi;
}
// y = clamp01(x * (a + b*x)).  Returns g_x.
CURL_HD float curve_bwd_self(float x, Affine k, float gy, float& P, float& Q) {
  float s = fmaf(k.b, x, k.a);
  float g_pre = gate01(gy, x * s);
  float g_s = g_pre * x;
  P += g_s;
  Q += g_s * x;
  return g_pre * fmaf(k.b, x, s);  // s + x*b
}

// adjust3 (curves.py:90-133 / 136-180): channel 0 meets its curve unclamped, channels 1,2 clamped first.
// in: p (inputs), g (gradient of the outputs); returns gradient of the inputs; P,Q += for curves 0..2.
CURL_HD Px adjust3_bwd(Px p, const Affine* k, Px g, float* P, float* Q) {
  Px gi;
  gi.c0 = curve_bwd_self(p.c0, k[0], g.c0, P[0], Q[0]);
  float c1 = clamp01(p.c1), c2 = clamp01(p.c2);
  gi.c1 = gate01(curve_bwd_self(c1, k[1], g.c1, P[1], Q[1]), p.c1);
  gi.c2 = gate01(curve_bwd_self(c2, k[2], g.c2, P[2], Q[2]), p.c2);
  return gi;
}

// ---- RGB -> Lab
CURL_HD Px rgb2lab_bwd(Px p, Px g) {
  float x[3] = {p.c0, p.c1, p.c2}, lin[3], dlin[3];
  for (int c = 0; c < 3; ++c) {
    float u = fmaf(x[c], kInv1055, (float)(0.055 / 1.055));
    float gam = pow_gamma(u);
    int hi = neg_mask(kSrgbThr - x[c]);  // x > threshold: the power branch
    lin[c] = blend(hi, gam, x[c] * kInv1292);
    // d/dx ((x+0.055)/1.055)^2.4 = 2.4/1.055 * u^1.4 = 2.4/1.055 * gam/u
    dlin[c] = blend(hi, ((float)2.4 * kInv1055) * gam * hw_rcp(u), kInv1292);
  }
  const float M[3][3] = {{0.412453f * kInvXn, 0.357580f * kInvXn, 0.180423f * kInvXn},
                         {0.212671f, 0.715160f, 0.072169f},
                         {0.019334f * kInvZn, 0.119193f * kInvZn, 0.950227f * kInvZn}};
  float t[3], df[3];
  for (int r = 0; r < 3; ++r) {
    t[r] = fmaf(M[r][2], lin[2], fmaf(M[r][1], lin[1], M[r][0] * lin[0]));
    float f = cbrt_pos(t[r]);
    df[r] = blend(neg_mask(kEps3 - t[r]), kThird * f * hw_rcp(t[r]), kInv3Eps2);  // (1/3) t^(-2/3) = f/(3t)
  }
  // L = 1.16 fy - 0.16 ; a = (fx - fy) ka + 0.5 ; b = (fy - fz) kb + 0.5
  const float ka = (float)(500.0 / 220.0), kb = (float)(200.0 / 220.0);
  float g_fx = g.c1 * ka;
  float g_fy = g.c0 * 1.16f - g.c1 * ka + g.c2 * kb;
  float g_fz = -g.c2 * kb;
  float g_t[3] = {g_fx * df[0], g_fy * df[1], g_fz * df[2]};
  Px gi;
  float gl[3];
  for (int c = 0; c < 3; ++c) gl[c] = (g_t[0] * M[0][c] + g_t[1] * M[1][c] + g_t[2] * M[2][c]) * dlin[c];
  gi.c0 = gl[0], gi.c1 = gl[1], gi.c2 = gl[2];
  return gi;
}

// ---- Lab -> RGB
CURL_HD Px lab2rgb_bwd(Px p, Px g) {
  const float c1 = (float)(100.0 / 116.0), c2 = (float)(16.0 / 116.0);
  const float ca = (float)(220.0 / 500.0), cb = (float)(-220.0 / 200.0);
  float fy = fmaf(p.c0, c1, c2);
  float fx = fmaf(p.c1, ca, fy - (float)(110.0 / 500.0));
  float fz = fmaf(p.c2, cb, fy + (float)(110.0 / 200.0));
  float f[3] = {fx, fy, fz}, X[3], dX[3];
  for (int i = 0; i < 3; ++i) {
    int hi = neg_mask(kEps - f[i]);
    float f2 = f[i] * f[i];
    X[i] = blend(hi, f2 * f[i], fmaf(f[i], k3Eps2, -(k3Eps2 * k4_29)));
    dX[i] = blend(hi, 3.0f * f2, k3Eps2);
  }
  const float M[3][3] = {{3.2404542f * kXn, -1.5371385f, -0.4985314f * kZn},
                         {-0.9692660f * kXn, 1.8760108f, 0.0415560f * kZn},
                         {0.0556434f * kXn, -0.2040259f, 1.0572252f * kZn}};
  float gv[3], go[3] = {g.c0, g.c1, g.c2};
  for (int r = 0; r < 3; ++r) {
    float v = fmaf(M[r][2], X[2], fmaf(M[r][1], X[1], M[r][0] * X[0]));
    float pw = pow_inv_gamma(v);
    // d/dv (1.055 v^(1/2.4) - 0.055) = 1.055/2.4 * v^(1/2.4 - 1) = 1.055/2.4 * pw / v
    float d = blend(neg_mask(kLinThr - v), (1.055f * kInvGamma) * pw * hw_rcp(v), 12.92f);
    gv[r] = go[r] * d;
  }
  float gX[3];
  for (int i = 0; i < 3; ++i) gX[i] = (gv[0] * M[0][i] + gv[1] * M[1][i] + gv[2] * M[2][i]) * dX[i];
  // fx = c1a*A + fy - .., fz = cb*B + fy + ..
  Px gi;
  gi.c0 = (gX[0] + gX[1] + gX[2]) * c1;
  gi.c1 = gX[0] * ca;
  gi.c2 = gX[2] * cb;
  return gi;
}

// ---- RGB -> HSV
CURL_HD Px rgb2hsv_bwd(Px p, Px g) {
  float q[3] = {p.c0, p.c1, p.c2}, c[3];
  int pin[3];
  for (int i = 0; i < 3; ++i) {
    c[i] = clampf(q[i], kHsvFloor, 1.0f);
    pin[i] = range_mask(q[i], kHsvFloor, 1.0f);
  }
  float r = c[0], gg = c[1], b = c[2];
  float mx = fmaxf(r, fmaxf(gg, b)), mn = fminf(r, fminf(gg, b));
  // first index attaining the max / the min (torch.max/min over dim=1), as masks
  int max0 = ge_mask(r, gg) & ge_mask(r, b), max1 = ~max0 & ge_mask(gg, b), max2 = ~(max0 | max1);
  int min0 = ge_mask(gg, r) & ge_mask(b, r), min1 = ~min0 & ge_mask(b, gg), min2 = ~(min0 | min1);
  float nd = mn - mx;                 // -(max - min): sign set iff the pixel is not flat
  int live = neg_mask(nd);
  float df = -nd;
  float dfi = keep_if(live, hw_rcp(df));
  float mxi = hw_rcp(mx);
  // [c == mx]: c - mx is negative exactly when c is NOT the maximum
  int ner = neg_mask(r - mx), neg_ = neg_mask(gg - mx), neb = neg_mask(b - mx);
  float er = drop_if(ner, 1.0f), eg = drop_if(neg_, 1.0f), eb = drop_if(neb, 1.0f);
  float Nn = drop_if(ner, gg - b) + drop_if(neg_, b - r) + drop_if(neb, r - gg);
  float h6 = keep_if(live, (drop_if(ner, (gg - b) * dfi) + drop_if(neg_, fmaf(b - r, dfi, 2.0f))) +
                               drop_if(neb, fmaf(r - gg, dfi, 4.0f)));
  float h = (h6 + keep_if(neg_mask(h6), 6.0f)) * (float)(1.0 / 6.0);
  float s = df * mxi;
  // output clamp (colors.py:240)
  float g_h = keep_if(range_mask(h, kHsvFloor, 1.0f), g.c0);
  float g_s = keep_if(range_mask(s, kHsvFloor, 1.0f), g.c1);
  float g_v = keep_if(range_mask(mx, kHsvFloor, 1.0f), g.c2);
  float g_h6 = keep_if(live, g_h * (float)(1.0 / 6.0));
  float g_N = g_h6 * dfi;
  float g_df = -(g_h6 * Nn) * dfi * dfi + g_s * mxi;
  float g_mx = -(g_s * df) * mxi * mxi + g_v + g_df;
  float g_mn = -g_df;
  float gc[3];
  gc[0] = g_N * (eb - eg) + keep_if(max0, g_mx) + keep_if(min0, g_mn);
  gc[1] = g_N * (er - eb) + keep_if(max1, g_mx) + keep_if(min1, g_mn);
  gc[2] = g_N * (eg - er) + keep_if(max2, g_mx) + keep_if(min2, g_mn);
  Px gi{keep_if(pin[0], gc[0]), keep_if(pin[1], gc[1]), keep_if(pin[2], gc[2])};
  return gi;
}

// ---- adjust_hsv (curves.py:41-87)
CURL_HD Px adjust_hsv4_bwd(Px p, const Affine* k, Px g, float* P, float* Q) {
  // forward
  float h = p.c0;
  float h1 = clamp01(curve_mul(h, h, k[0]));
  float sc = clamp01(p.c1), vc = clamp01(p.c2);
  float s1 = clamp01(curve_mul(sc, h1, k[1]));
  // backward, last curve first
  float g_vc = curve_bwd_self(vc, k[3], g.c2, P[3], Q[3]);
  float g_s1 = curve_bwd_self(s1, k[2], g.c1, P[2], Q[2]);
  float g_sc, g_h1 = g.c0;
  curve_bwd_cross(sc, h1, k[1], g_s1, g_sc, g_h1, P[1], Q[1]);
  float g_h = curve_bwd_self(h, k[0], g_h1, P[0], Q[0]);
  Px gi{g_h, gate01(g_sc, p.c1), gate01(g_vc, p.c2)};
  return gi;
}

// ---- HSV -> RGB
CURL_HD Px hsv2rgb_bwd(Px p, Px g) {
  float hh = clamp01(p.c0), ss = clamp01(p.c1), vv = clamp01(p.c2);
  float H = hh * 6.0f;
  float q = vv * (1.0f - ss);
  float d = vv - q;
  float a1 = H - 1.0f, a4 = H - 4.0f, a0 = H, a3 = H - 3.0f, a2 = H - 2.0f, a5 = H - 5.0f;
  float R1 = clamp01(a1), R4 = clamp01(a4), G0 = clamp01(a0), G3 = clamp01(a3), B2 = clamp01(a2), B5 = clamp01(a5);
  float r = fmaf(R4, d, fmaf(R1, -d, vv));
  float gn = fmaf(G3, -d, fmaf(G0, d, q));
  float b = fmaf(B5, -d, fmaf(B2, d, q));
  float gr = gate01(g.c0, r), gg = gate01(g.c1, gn), gb = gate01(g.c2, b);
  float g_vv = gr;
  float g_q = gg + gb;
  float g_d = gr * (R4 - R1) + gg * (G0 - G3) + gb * (B2 - B5);
  float g_H = (gr * d) * (pass01(a4) - pass01(a1)) + (gg * d) * (pass01(a0) - pass01(a3)) +
              (gb * d) * (pass01(a2) - pass01(a5));
  // d = vv - q
  g_vv += g_d;
  g_q -= g_d;
  // q = vv (1 - ss)
  g_vv += g_q * (1.0f - ss);
  float g_ss = -g_q * vv;
  Px gi{gate01(6.0f * g_H, p.c0), gate01(g_ss, p.c1), gate01(g_vv, p.c2)};
  return gi;
}

// ---- the whole layer.  P,Q [10]: += per curve (0-2 lab, 3-5 rgb, 6-9 hsv).  Returns d loss / d in.
CURL_HD Px curl_layer_bwd(Px in, float m, const LayerCoef& k, Px gout, float* P, float* Q) {
  // forward, keeping the stage inputs
  Px lab0 = rgb2lab(in);
  Px lab1 = adjust3(lab0, k.lab[0], k.lab[1], k.lab[2]);
  Px lab2{lab1.c0 * m, lab1.c1 * m, lab1.c2 * m};
  Px rgb1 = lab2rgb(lab2);
  Px rgb2 = adjust3(rgb1, k.rgb[0], k.rgb[1], k.rgb[2]);
  Px rgb3{rgb2.c0 * m, rgb2.c1 * m, rgb2.c2 * m};
  Px hsv0 = rgb2hsv(rgb3);
  Px hsv1 = adjust_hsv4(hsv0, k.hsv[0], k.hsv[1], k.hsv[2], k.hsv[3]);
  Px hsv2{hsv1.c0 * m, hsv1.c1 * m, hsv1.c2 * m};
  Px res = hsv2rgb(hsv2);
  // out = clamp01(in + res) * m   (model.py:170)
  Px g_pre{gate01(gout.c0 * m, in.c0 + res.c0), gate01(gout.c1 * m, in.c1 + res.c1),
           gate01(gout.c2 * m, in.c2 + res.c2)};
  Px g = hsv2rgb_bwd(hsv2, g_pre);
  g = Px{g.c0 * m, g.c1 * m, g.c2 * m};
  g = adjust_hsv4_bwd(hsv0, k.hsv, g, P + 6, Q + 6);
  g = rgb2hsv_bwd(rgb3, g);
  g = Px{g.c0 * m, g.c1 * m, g.c2 * m};
  g = adjust3_bwd(rgb1, k.rgb, g, P + 3, Q + 3);
  g = lab2rgb_bwd(lab2, g);
  g = Px{g.c0 * m, g.c1 * m, g.c2 * m};
  g = adjust3_bwd(lab0, k.lab, g, P, Q);
  g = rgb2lab_bwd(in, g);
  return Px{g.c0 + g_pre.c0, g.c1 + g_pre.c1, g.c2 + g_pre.c2};
}

// ---- per image: (P, Q) of one curve + d loss / d reg  ->  gradient of that curve's RAW knots (pre-exp).
// C = exp(raw) (already computed); scale = C0 + sum_{j<=K-3} slope_j (S x - j); reg = sum_j (slope_{j+1}-slope_j)^2.
// Everything in float64: K is tiny and this runs once per curve per image.
CURL_HD void knots_bwd(const float* C, int K, double P, double Q, double g_reg, float* g_raw) {
  const double S = (double)(K - 1);
  for (int kk = 0; kk < K; ++kk) {
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
    g_raw[kk] = (float)(gC * (double)C[kk]);  // d exp(raw) = C
  }
}

}  // namespace curlm
