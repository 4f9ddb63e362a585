// curl_math_loss.h -- the per-pixel terms of the reference's CURLLoss (model.py:78-116):
// masked L1 in RGB, cosine similarity, L1 in clamped Lab, L1 on the HSV cone.  (The MS-SSIM term of
// model.py:103-105 is grouped convolutions on the L plane -- stock PyTorch; this header hands it the L planes
// and takes its gradient back.)  Same dual compilation as curl_math.h.
#pragma once
#include "curl_math_bwd.h"

namespace curlm {

constexpr float kTwoPi = (float)(2 * 3.141592653589793);  // 2*math.pi as float32 (model.py:70)
constexpr float kCosEps = 1e-8f;                          // torch cosine_similarity eps

// sqrt of the cosine term's squared norms.  Device: v_sqrt_f32 (1 ulp) -- the correctly rounded sqrtf expands to a scaled
// Newton sequence with two `v_cndmask ..., vcc` (23 cycles each) per root; a relative 6e-8 on a cosine in
// [-1, 1] is below what the reference's own float32 sums carry.  The quotient likewise: v_rcp_f32 + multiply.
CURL_HD float loss_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}

struct LossPx {
  float rgb_l1, cos_sim, lab_l1, hsv_l1;  // this pixel's contribution to the four sums
  float Lp, Lt;                           // clamped L of prediction and target (for MS-SSIM)
};

// cos and sin of float32(2*pi) * h for h in [0, 1] (model.py:70-75).  On gfx950: v_cos_f32 / v_sin_f32, which take their
// argument in REVOLUTIONS -- over h = i / 2^22 they are within 1.3e-7 of the exact cos / sin(2 pi h), closer than the
// reference's own float32 evaluation is (4.1e-7: its angle is rounded to float32 first), and within 4.6e-7 of the
// reference's values (tools/ubench/sincos_probe.hip, profiles/r02/sincos_probe.log).  The library's sinf / cosf do a
// full-range Payne-Hanek reduction: 100+ integer instructions and a branch each, 60 % of the loss kernel's instructions.
CURL_HD void cos_sin_turns(float h, float& c, float& s) {
#if defined(__HIP_DEVICE_COMPILE__)
  c = __builtin_amdgcn_cosf(h), s = __builtin_amdgcn_sinf(h);
#else
  const float a = kTwoPi * h;
  c = cosf(a), s = sinf(a);
#endif
}

// model.py:65-76 on clamp(rgb2hsv(x), 0, 1).  That clamp is the identity: RGB2HSV ends in clamp(., 1e-9, 1) (colors.py:240,
// rgb2hsv_n / rgb2hsv_t), so h, s, v arrive in [1e-9, 1] and torch.clamp's gradient gate [0 <= x <= 1] always passes --
// round 5 took the three clamps per colour and the three gates of the backward out of the kernels (6 of 241 and 21 of 410
// instructions per pixel; same bits: a clamp of a value inside its interval returns it).
CURL_HD Px hsv_cone(Px hsv) {
#pragma clang fp contract(off)  // (equal colours -> equal cones, whatever the caller does with them: loss_terms_bwd)
  float h = hsv.c0, s = hsv.c1, v = hsv.c2;
  float ca, sa;
  cos_sin_turns(h, ca, sa);
  return Px{v * s * ca, v * s * sa, v};
}

CURL_HD LossPx loss_terms(Px pred, Px tgt, float m) {
  Px p{pred.c0 * m, pred.c1 * m, pred.c2 * m}, t{tgt.c0 * m, tgt.c1 * m, tgt.c2 * m};  // model.py:91
  LossPx o;
  o.rgb_l1 = (fabsf(p.c0 - t.c0) + fabsf(p.c1 - t.c1)) + fabsf(p.c2 - t.c2);  // model.py:93
  float d = p.c0 * t.c0 + p.c1 * t.c1 + p.c2 * t.c2;
  float np = loss_sqrt(p.c0 * p.c0 + p.c1 * p.c1 + p.c2 * p.c2), nt = loss_sqrt(t.c0 * t.c0 + t.c1 * t.c1 + t.c2 * t.c2);
  o.cos_sim = d * hw_rcp(fmaxf(np, kCosEps) * fmaxf(nt, kCosEps));  // model.py:97
  Px lp = rgb2lab(p), lt = rgb2lab(t);                          // model.py:100-101 (+ clamp, model.py:55)
  lp = Px{clamp01(lp.c0), clamp01(lp.c1), clamp01(lp.c2)};
  lt = Px{clamp01(lt.c0), clamp01(lt.c1), clamp01(lt.c2)};
  o.lab_l1 = (fabsf(lp.c0 - lt.c0) + fabsf(lp.c1 - lt.c1)) + fabsf(lp.c2 - lt.c2);
  o.Lp = lp.c0;
  o.Lt = lt.c0;
  Px cp = hsv_cone(rgb2hsv(p)), ct = hsv_cone(rgb2hsv(t));      // model.py:107-109
  o.hsv_l1 = (fabsf(cp.c0 - ct.c0) + fabsf(cp.c1 - ct.c1)) + fabsf(cp.c2 - ct.c2);
  return o;
}

// The same over the N pixels a lane owns, written as phases over 2N colours (N predictions, N targets): the converters'
// transcendental runs are 6N long instead of 3, the cone's cos / sin run 4N (issue priority, curl_math.h).  sum[0..3] +=
// the four terms over the N pixels; the arithmetic per pixel is loss_terms' (the single-pixel converters ARE the N = 1 forms).
template <int N>
CURL_HD void loss_terms_n(const PxN<N>& pred, const PxN<N>& tgt, const float (&m)[N], float (&sum)[4], float (&Lp)[N],
                          float (&Lt)[N]) {
  PxN<2 * N> x;
#pragma unroll
  for (int i = 0; i < N; ++i) {  // model.py:91
    x.c0[i] = pred.c0[i] * m[i], x.c1[i] = pred.c1[i] * m[i], x.c2[i] = pred.c2[i] * m[i];
    x.c0[N + i] = tgt.c0[i] * m[i], x.c1[N + i] = tgt.c1[i] * m[i], x.c2[N + i] = tgt.c2[i] * m[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int j = N + i;
    sum[0] += (fabsf(x.c0[i] - x.c0[j]) + fabsf(x.c1[i] - x.c1[j])) + fabsf(x.c2[i] - x.c2[j]);  // model.py:93
    const float d = x.c0[i] * x.c0[j] + x.c1[i] * x.c1[j] + x.c2[i] * x.c2[j];
    const float np = loss_sqrt(x.c0[i] * x.c0[i] + x.c1[i] * x.c1[i] + x.c2[i] * x.c2[i]);
    const float nt = loss_sqrt(x.c0[j] * x.c0[j] + x.c1[j] * x.c1[j] + x.c2[j] * x.c2[j]);
    sum[1] += d * hw_rcp(fmaxf(np, kCosEps) * fmaxf(nt, kCosEps));  // model.py:97
  }
  {
    PxN<2 * N> lab = x;
    rgb2lab_n<2 * N>(lab);  // model.py:100-101 (+ clamp, model.py:55); eager selects: the lazy / predicated form moved nothing here (exp18)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int j = N + i;
      const float l0 = clamp01(lab.c0[i]), l1 = clamp01(lab.c0[j]);
      sum[2] += (fabsf(l0 - l1) + fabsf(clamp01(lab.c1[i]) - clamp01(lab.c1[j]))) + fabsf(clamp01(lab.c2[i]) - clamp01(lab.c2[j]));
      Lp[i] = l0, Lt[i] = l1;
    }
  }
  {
    PxN<2 * N> hsv = x;
    rgb2hsv_n<2 * N>(hsv);  // model.py:107-109
    float ca[2 * N], sa[2 * N];
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) ca[i] = hsv.c0[i];  // (model.py:66's clamp is the identity on RGB2HSV's output: hsv_cone)
    CURL_FENCE();
    CURL_TRANS_BEGIN();
#pragma unroll
    for (int i = 0; i < 2 * N; ++i) {
      const float h = ca[i];
      cos_sin_turns(h, ca[i], sa[i]);
    }
    CURL_TRANS_END();
    CURL_FENCE();
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int j = N + i;
      const float vi = hsv.c2[i], vj = hsv.c2[j];
      const float ri = vi * hsv.c1[i], rj = vj * hsv.c1[j];
      sum[3] += (fabsf(ri * ca[i] - rj * ca[j]) + fabsf(ri * sa[i] - rj * sa[j])) + fabsf(vi - vj);
    }
  }
}

// Masked PSNR (metric.py:35-47): one channel's squared error under the mask, (clamp(a) m - clamp(b) m)^2, added to acc.
// Two ROUNDED products, then their difference: contracted into fma(a, m, -(b m)) -- hipcc's default -- equal images under a
// fractional float mask left a rounding residue, 160 dB where the reference has +inf (round 4).
CURL_HD float psnr_sq_err(float a, float b, float m, float acc) {
#pragma clang fp contract(off)
  const float pa_m = clamp01(a) * m, pb_m = clamp01(b) * m;  // metric.py:60-61, then :44
  const float d = pa_m - pb_m;
  return fmaf(d, d, acc);
}

// torch.sign (the L1 terms' backward): -1, 0, +1.  Device: two exact scalings by 2^100 take every nonzero float32 past +-1
// (the smallest denormal: 2^-149 * 2^200 = 2^51; an overflow to +-inf is fine), then a median with -1 and +1 -- three plain
// instructions instead of two compare + `v_cndmask ..., vcc` pairs.
CURL_HD float sign0(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float big = 0x1p100f;
  return __builtin_amdgcn_fmed3f((x * big) * big, -1.0f, 1.0f);
#else
  return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
#endif
}

// w * torch.sign(x) in three instructions where `w * sign0(x)` takes four: the second scaling carries the weight,
// (x 2^100) (2^100 w) has the sign of x w and -- for any nonzero float32 x and |w| >= 1e-15 -- a magnitude far above |w|, so the
// median with -|w| and +|w| is exactly w sign(x); x = 0 gives 0.  (The weights are d loss / d sum divided by the pixel count:
// 1e-7 ... 1; an |x| above 2^27 would overflow the first product -- differences of values in [0, 1] do not.)
struct SignW {
  float bw, aw;  // 2^100 w, |w|
};
CURL_HD SignW signw_of(float w) { return SignW{0x1p100f * w, fabsf(w)}; }
CURL_HD float signw(float x, float w, const SignW& k) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)w;
  return __builtin_amdgcn_fmed3f((x * 0x1p100f) * k.bw, -k.aw, k.aw);
#else
  (void)k;
  return w * sign0(x);
#endif
}

// d(sum_k w[k] * term_k + gLp * Lp) / d pred for one pixel; w = weights of (rgb_l1, cos_sim, lab_l1, hsv_l1).
CURL_HD Px loss_terms_bwd(Px pred, Px tgt, float m, const float (&w)[4], float gLp) {
  // The differences below feed torch.sign: contracted into fma(v s, cos, -cone_t) -- exact product minus rounded product -- a
  // difference of EQUAL cones is a rounding residue and its sign +-1, where the reference has sign(0) = 0 (found in round 4 with
  // pred == target pixels; the g++ host twin is built without contraction and never saw it -- tests/conftest.py's second,
  // contracting twin does).  Explicit fmaf's only in here.
#pragma clang fp contract(off)
  Px p{pred.c0 * m, pred.c1 * m, pred.c2 * m}, t{tgt.c0 * m, tgt.c1 * m, tgt.c2 * m};
  // rgb L1
  const SignW k0 = signw_of(w[0]), k2 = signw_of(w[2]), k3 = signw_of(w[3]);  // (loop-invariant: once per wave)
  Px g{signw(p.c0 - t.c0, w[0], k0), signw(p.c1 - t.c1, w[0], k0), signw(p.c2 - t.c2, w[0], k0)};
  // cosine similarity: c = d / (max(np,eps) max(nt,eps))
  float d = fmaf(p.c2, t.c2, fmaf(p.c1, t.c1, p.c0 * t.c0));
  float np = loss_sqrt(fmaf(p.c2, p.c2, fmaf(p.c1, p.c1, p.c0 * p.c0))), nt = loss_sqrt(fmaf(t.c2, t.c2, fmaf(t.c1, t.c1, t.c0 * t.c0)));
  float npc = fmaxf(np, kCosEps), ntc = fmaxf(nt, kCosEps);
  float inv = hw_rcp(npc * ntc);
  // d/dp of 1/max(np,eps) is -p/np^3-ish only when np > eps (sign-bit mask of eps - np; np = 0 gives 0 * inf masked to 0)
  float k = keep_if(neg_mask(kCosEps - np), d * inv * hw_rcp(npc * fmaxf(np, kCosEps)));
  g.c0 = fmaf(w[1], fmaf(-k, p.c0, t.c0 * inv), g.c0);
  g.c1 = fmaf(w[1], fmaf(-k, p.c1, t.c1 * inv), g.c1);
  g.c2 = fmaf(w[1], fmaf(-k, p.c2, t.c2 * inv), g.c2);
  // Lab L1 (+ the MS-SSIM gradient arriving on the clamped L plane)
  // (the prediction's Lab from the TAPED converter: its L is exactly 0 at black -- a prediction the layer's clamp saturated at 0 --
  // where the reference's is, so model.py:55's clamp gate passes there as torch's does; curl_math_bwd.h rgb2lab_t)
  // The target goes through the same function (its tape is dead code): pred == target must give lp == lt bit for bit, as in
  // the reference, or sign(lp - lt) would be +-1 where torch.sign(0) is 0.
  Rgb2LabT tape_lab, tape_unused;
  Px lp = rgb2lab_t(p, tape_lab), lt = rgb2lab_t(t, tape_unused);
  // model.py:55's clamp and its gradient gate [0 <= x <= 1] from ONE compare each (clamp_gate), applied as a select: as
  // `value * pass01(x)` a gate was clamp + compare + select(1.0) + multiply
  lmask gate0, gate1, gate2;
  Px lpc{clamp_gate(lp.c0, 0.0f, 1.0f, gate0), clamp_gate(lp.c1, 0.0f, 1.0f, gate1), clamp_gate(lp.c2, 0.0f, 1.0f, gate2)};
  Px ltc{clamp01(lt.c0), clamp01(lt.c1), clamp01(lt.c2)};
  Px gl{lm_keep(gate0, signw(lpc.c0 - ltc.c0, w[2], k2) + gLp), lm_keep(gate1, signw(lpc.c1 - ltc.c1, w[2], k2)),
        lm_keep(gate2, signw(lpc.c2 - ltc.c2, w[2], k2))};
  Px g_lab = rgb2lab_pull(tape_lab, gl);
  // HSV cone L1
  // (the taped converter for both colours, as for Lab: one evaluation of the prediction's forward instead of rgb2hsv +
  // rgb2hsv_bwd's own, and pred == target gives identical cones)
  Rgb2HsvT tape_hsv, tape_hsv_unused;
  Px hp = rgb2hsv_t(p, tape_hsv);
  Px cp = hsv_cone(hp), ct = hsv_cone(rgb2hsv_t(t, tape_hsv_unused));
  float ge0 = signw(cp.c0 - ct.c0, w[3], k3), ge1 = signw(cp.c1 - ct.c1, w[3], k3), ge2 = signw(cp.c2 - ct.c2, w[3], k3);
  float h = hp.c0, s = hp.c1, v = hp.c2;  // in [1e-9, 1]: model.py:66's clamp is the identity and its gate passes (hsv_cone)
  float ca, sa;
  cos_sin_turns(h, ca, sa);
  const float radial = fmaf(ge1, sa, ge0 * ca);
  Px gh{kTwoPi * v * s * fmaf(ge1, ca, -(ge0 * sa)), v * radial, fmaf(s, radial, ge2)};
  Px g_hsv = rgb2hsv_pull(tape_hsv, gh);
  return Px{(g.c0 + g_lab.c0 + g_hsv.c0) * m, (g.c1 + g_lab.c1 + g_hsv.c1) * m, (g.c2 + g_lab.c2 + g_hsv.c2) * m};
}

}  // namespace curlm
