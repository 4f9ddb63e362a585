// curl_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI for the CURL colour-curve path.
//
// Everything here is a streaming, per-pixel, HBM-bound pass over NCHW float32 planes:
//   * one workgroup = 256 threads = 4 wavefronts of 64; each lane owns U float4 groups per plane,
//     so every wave-instruction moves 1 KiB of one plane, perfectly coalesced;
//   * blockIdx -> (image, chunk); everything that depends on the image (the collapsed curve
//     coefficients) is wave-uniform and is fetched with scalar loads into SGPRs;
//   * curves evaluated in the reference's exact summation order, and the paper-style PWL lookup,
//     stage the slopes/knots of the image in LDS (broadcast reads / per-lane gathers);
//   * no MFMA: there is no contraction on this path (roofline = HBM).
//
// Interface: include/curl_hip.h.  Arithmetic: curl_math.h (cites the reference line by line).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "../../include/curl_hip.h"
#include "curl_math_bwd.h"
#include "curl_math_poly.h"
#include "curl_math_loss.h"

using namespace curlm;

constexpr int kTriSpaceWaves = 4;  // register budget of the polynomial kernel: 128 VGPRs (it needs ~106)

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int fail(int code, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s", what);
  return code;
}
static int hip_fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

// ------------------------------------------------------------------------------------------------
// workspace layout (floats, per image)
//   [0..19]  (a,b) of curve c at [2c],[2c+1]   (segment order: see prep kernel)
//   [20..22] regulariser of segment 0,1,2 ; [23] total
//   [24..26] lab2rgb(0,0,0): what the Lab stage yields where the mask is 0 (model.py:154-157)
//   [29]     (uint) the row's stamp: which knot count / row stride a prep pass filled it for (ws_stamp)
//   [32..)   exp'd knots of every curve, segment after segment
// ------------------------------------------------------------------------------------------------
#define WS_COEF 0
#define WS_REG 20
#define WS_MASKED 24
#define WS_STAMP 29
#define WS_KNOTS 32
#define MAX_CURVES 10

static inline unsigned ws_stride(int n_knots) { return WS_KNOTS + ((unsigned)(n_knots + 3) & ~3u); }
// what a prepared workspace row says about itself: the knot count and row stride it was filled for
__host__ __device__ static inline unsigned ws_stamp(unsigned n_knots, unsigned stride) { return 0x43550000u ^ (n_knots * 2654435761u) ^ stride; }

// Knots per curve as the C ABI carries them (include/curl_hip.h, CURL_K_UNEVEN): K in the low 16 bits; the high 16 bits,
// when non-zero, are the knot count of the segment's LAST curve -- what torch.chunk hands it when the parameter count does
// not divide (curves.py:53,105,152).  Curve c of a segment of n curves starts c * K knots into it.
#define KP_K(p) ((int)((unsigned)(p) & 0xffffu))
#define KP_LAST(p) ((int)(((unsigned)(p) >> 16) ? ((unsigned)(p) >> 16) : ((unsigned)(p) & 0xffffu)))
#define KP_TOTAL(p, n) (((n) - 1) * KP_K(p) + KP_LAST(p))
#define KP_UNEVEN(p) (KP_LAST(p) != KP_K(p))

struct PrepArgs {
  const float* raw[3];  // per segment: [B, ncurves*K] raw (pre-exp) parameters, NULL if absent
  int ncurves[3];
  int K[3];  // packed (KP_*)
  float* ws;
  float* reg_out;  // nullable, [B], assigned
  unsigned stride;
};

// The curves of ONE image, by one 256-thread workgroup: exp in float64 (rounded once to float32: the best estimate of
// torch.exp's float32 result), slopes in float32 as the reference forms them (curves.py:19), every sum in float64.
//   store: write the image's workspace row (and reg_out): knots_prep_kernel's one workgroup per image -- or, when the
//          streaming kernel collapses its curves itself (stream_selfprep_kernel), the image's first workgroup;
//   head:  NULL, or 32 floats of LDS that receive the row's head (coefficients, regularisers, masked-out colour) for the
//          calling workgroup's own use.  Both forms run THIS code: their results are bit-identical.
// raw knot i of image b (i counts through the call's segments).  The three segments' pointers and sizes are taken out of
// the kernel-argument struct as OPAQUE scalars first: a select over `a.raw[s]` is otherwise compiled into an indexed VECTOR
// load from the kernel-argument segment -- the pointer arrived by a global load and the knot by a second, dependent one
// (two memory latencies at the head of every in-kernel collapse; round 5, seen in the ISA: `global_load_dword ... offset:104`).
// (as a macro declaring LOCAL scalars: handed around in a struct they live in memory again and the select turns back into
// an indexed load -- from scratch this time)
#define PREP_SEGS(a)                                                                                                     \
  const float *seg_r0 = (a).raw[0], *seg_r1 = (a).raw[1], *seg_r2 = (a).raw[2];                                            \
  int seg_n0 = KP_TOTAL((a).K[0], (a).ncurves[0]), seg_n1 = KP_TOTAL((a).K[1], (a).ncurves[1]),                           \
      seg_n2 = KP_TOTAL((a).K[2], (a).ncurves[2]);                                                                        \
  asm volatile("" : "+s"(seg_r0), "+s"(seg_r1), "+s"(seg_r2), "+s"(seg_n0), "+s"(seg_n1), "+s"(seg_n2))
// (the pointers lose their address space in the asm: said again, or the knot is a FLAT load, whose return order against the
// tile's global loads is undefined -- the compiler then waits for every load in flight before the first exp)
typedef const float __attribute__((address_space(1))) * prep_gptr;
#define PREP_RAW(b, i, off1, off2)                                                                                       \
  (((prep_gptr)((i) >= (off2) ? seg_r2 : (i) >= (off1) ? seg_r1 : seg_r0))[(size_t)(b) * ((i) >= (off2) ? seg_n2 : (i) >= (off1) ? seg_n1 : seg_n0) + \
                                                                          ((i) - ((i) >= (off2) ? (off2) : (i) >= (off1) ? (off1) : 0))])
__device__ __forceinline__ void prep_offsets(const PrepArgs& a, int (&seg_off)[4]) {
  seg_off[0] = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) seg_off[s + 1] = seg_off[s] + (a.raw[s] ? KP_TOTAL(a.K[s], a.ncurves[s]) : 0);
}
// The lane's FIRST raw knot (i = threadIdx.x), asked for on its own: the in-kernel collapse issues this load BEFORE its tile's
// plane loads -- vector loads return in order, so asked for behind them the knot arrived after 7 KB of pixels per wave and the
// whole prologue started ~1 us late (round 5).
__device__ __forceinline__ float prep_fetch_first(const PrepArgs& a, unsigned b) {
  int seg_off[4];
  prep_offsets(a, seg_off);
  PREP_SEGS(a);
  const int i = (int)threadIdx.x;
  return i < seg_off[3] ? PREP_RAW(b, i, seg_off[1], seg_off[2]) : 0.0f;
}
//   first_raw: prep_fetch_first's value when the caller asked for it early (have_first), else loaded here
__device__ __forceinline__ void prep_image(const PrepArgs& a, unsigned b, bool store, float* head, float first_raw = 0.0f,
                                           bool have_first = false) {
  __shared__ float sC[MAX_CURVES * CURL_MAX_KNOTS];
  __shared__ float sReg[MAX_CURVES];
  float* ws = a.ws + (size_t)b * a.stride;

  int seg_off[4];
  prep_offsets(a, seg_off);
  const int n_total = seg_off[3];
  PREP_SEGS(a);
  // (the lane's first knot outside the loop: inside it the wait for an early-fetched value is a wait for EVERY load in flight)
  int i = threadIdx.x;
  if (i < n_total) {
    const float r = have_first ? first_raw : PREP_RAW(b, i, seg_off[1], seg_off[2]);
    const float c = (float)exp((double)r);  // curves.py:54,106,153
    sC[i] = c;
    if (store) ws[WS_KNOTS + i] = c;
  }
  for (i += 256; i < n_total; i += 256) {  // more than 256 knots per image: K > 25
    const float r = PREP_RAW(b, i, seg_off[1], seg_off[2]);
    const float c = (float)exp((double)r);
    sC[i] = c;
    if (store) ws[WS_KNOTS + i] = c;
  }
  __syncthreads();
  // sixteen lanes per curve (curl_math.h collapse_partial / collapse_finish: the order collapse_curve takes alone); the
  // workgroup's last lane computes the masked-out colour beside them
  int curve0[4];
  curve0[0] = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) curve0[s + 1] = curve0[s] + (a.raw[s] ? a.ncurves[s] : 0);
  const int n_curves = curve0[3];
  const int c = threadIdx.x / kCollapseLanes, l = threadIdx.x % kCollapseLanes;
  static_assert(MAX_CURVES * kCollapseLanes <= 255, "a 16-lane group per curve and one lane to spare in a 256-thread workgroup");
  if (c < n_curves) {
    // (selects over the three segments, not a.K[s]: a run-time index into a kernel argument is a global load)
    const int s = (c >= curve0[2]) ? 2 : (c >= curve0[1]) ? 1 : 0;
    const int nc_s = s == 2 ? a.ncurves[2] : s == 1 ? a.ncurves[1] : a.ncurves[0];
    const int K_s = s == 2 ? a.K[2] : s == 1 ? a.K[1] : a.K[0];
    const int off_s = s == 2 ? seg_off[2] : s == 1 ? seg_off[1] : seg_off[0];
    const int local = c - (s == 2 ? curve0[2] : s == 1 ? curve0[1] : curve0[0]);
    const int K = (local == nc_s - 1) ? KP_LAST(K_s) : KP_K(K_s);  // torch.chunk: the last curve may be shorter
    const float* C = sC + off_s + local * KP_K(K_s);
    CollapseSums t = collapse_partial(C, K, l);
#pragma unroll
    for (int o = kCollapseLanes / 2; o; o >>= 1) {  // (whole 16-lane groups are active or not: lanes trade inside their group)
      t.s += __shfl_xor(t.s, o, kCollapseLanes);
      t.js += __shfl_xor(t.js, o, kCollapseLanes);
      t.r += __shfl_xor(t.r, o, kCollapseLanes);
    }
    if (l == 0) {
      float ca, cb, creg;
      collapse_finish(C, K, t, ca, cb, creg);
      if (store) ws[WS_COEF + 2 * c] = ca, ws[WS_COEF + 2 * c + 1] = cb;
      if (head) head[WS_COEF + 2 * c] = ca, head[WS_COEF + 2 * c + 1] = cb;
      sReg[c] = creg;
    }
  }
  if (threadIdx.x == 255) {
    const Px z = lab_stage_masked_out();
    if (store) ws[WS_MASKED + 0] = z.c0, ws[WS_MASKED + 1] = z.c1, ws[WS_MASKED + 2] = z.c2;
    if (head) head[WS_MASKED + 0] = z.c0, head[WS_MASKED + 1] = z.c1, head[WS_MASKED + 2] = z.c2;
  }
  if (!store) return;  // (the caller's barrier follows: `head` is complete behind it)
  __syncthreads();
  if (threadIdx.x == 0) {
    float seg_reg[3] = {0.0f, 0.0f, 0.0f};
    for (int s = 0; s < 3; ++s) {
      float r = 0.0f;
      for (int k = curve0[s]; k < curve0[s + 1]; ++k) r += sReg[k];  // reg += per curve (curves.py:24)
      seg_reg[s] = r;
    }
    const float tot = (seg_reg[0] + seg_reg[1]) + seg_reg[2];  // model.py:172-174 (rgb + lab) + hsv
    ws[WS_REG + 0] = seg_reg[0], ws[WS_REG + 1] = seg_reg[1], ws[WS_REG + 2] = seg_reg[2], ws[WS_REG + 3] = tot;
    if (a.reg_out) a.reg_out[b] = tot;
    // the row's identity, which CURL_F_WS_READY callers are checked against (knots_bwd_kernel, layer_bwd_kernel)
    reinterpret_cast<unsigned*>(ws)[WS_STAMP] = ws_stamp((unsigned)n_total, a.stride);
  }
}

// One workgroup per image (the two-launch form: every knot-driven entry point runs this first for large launches).
__global__ __launch_bounds__(256) void knots_prep_kernel(PrepArgs a) { prep_image(a, blockIdx.x, true, nullptr); }

#include "kernels/stream.inc"
#include "kernels/ops.inc"
#include "kernels/chain.inc"
#include "kernels/layer_bwd.inc"
#include "kernels/psnr.inc"
#include "kernels/msssim.inc"
#include "kernels/poly_bwd.inc"
#include "kernels/loss.inc"
#include "kernels/edges.inc"
#include "kernels/host_api.inc"
