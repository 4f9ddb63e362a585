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

#include "../../include/curl_hip.h"
#include "curl_math_bwd.h"
#include "curl_math_poly.h"
#include "curl_math_loss.h"

using namespace curlm;

#ifndef CURL_TRISPACE_WAVES
#define CURL_TRISPACE_WAVES 4  // register budget of the polynomial kernel: 128 VGPRs (it needs ~106)
#endif

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
//   [32..)   exp'd knots of every curve, segment after segment
// ------------------------------------------------------------------------------------------------
#define WS_COEF 0
#define WS_REG 20
#define WS_MASKED 24
#define WS_KNOTS 32
#define MAX_CURVES 10

static inline unsigned ws_stride(int n_knots) { return WS_KNOTS + ((unsigned)(n_knots + 3) & ~3u); }

struct PrepArgs {
  const float* raw[3];  // per segment: [B, ncurves*K] raw (pre-exp) parameters, NULL if absent
  int ncurves[3];
  int K[3];
  float* ws;
  float* reg_out;  // nullable, [B], assigned
  unsigned stride;
};

// One workgroup per image.  exp in float64 (rounded once to float32: the best estimate of torch.exp's
// float32 result), slopes in float32 as the reference forms them (curves.py:19), every sum in float64.
__global__ __launch_bounds__(256) void knots_prep_kernel(PrepArgs a) {
  __shared__ float sC[MAX_CURVES * CURL_MAX_KNOTS];
  __shared__ float sReg[MAX_CURVES];
  const unsigned b = blockIdx.x;
  float* ws = a.ws + (size_t)b * a.stride;
  int seg_off[4];
  seg_off[0] = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) seg_off[s + 1] = seg_off[s] + (a.raw[s] ? a.ncurves[s] * a.K[s] : 0);
  const int n_total = seg_off[3];
  for (int i = threadIdx.x; i < n_total; i += 256) {
    int s = (i >= seg_off[2]) ? 2 : (i >= seg_off[1]) ? 1 : 0;
    int local = i - seg_off[s];
    int per_img = a.ncurves[s] * a.K[s];
    float r = a.raw[s][(size_t)b * per_img + local];
    float c = (float)exp((double)r);  // curves.py:54,106,153
    sC[i] = c;
    ws[WS_KNOTS + i] = c;
  }
  __syncthreads();
  // one thread per curve
  int curve0[4];
  curve0[0] = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) curve0[s + 1] = curve0[s] + (a.raw[s] ? a.ncurves[s] : 0);
  const int n_curves = curve0[3];
  const int c = threadIdx.x;
  if (c < n_curves) {
    int s = (c >= curve0[2]) ? 2 : (c >= curve0[1]) ? 1 : 0;
    int K = a.K[s];
    const float* C = sC + seg_off[s] + (c - curve0[s]) * K;
    float ca, cb, creg;
    collapse_curve(C, K, ca, cb, creg);
    ws[WS_COEF + 2 * c] = ca;
    ws[WS_COEF + 2 * c + 1] = cb;
    sReg[c] = creg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.0f;
    float seg_reg[3] = {0.0f, 0.0f, 0.0f};
    for (int s = 0; s < 3; ++s) {
      float r = 0.0f;
      for (int k = curve0[s]; k < curve0[s + 1]; ++k) r += sReg[k];  // reg += per curve (curves.py:24)
      seg_reg[s] = r;
      ws[WS_REG + s] = r;
    }
    tot = (seg_reg[0] + seg_reg[1]) + seg_reg[2];  // model.py:172-174 (rgb + lab) + hsv
    ws[WS_REG + 3] = tot;
    if (a.reg_out) a.reg_out[b] = tot;
    Px z = lab_stage_masked_out();
    ws[WS_MASKED + 0] = z.c0;
    ws[WS_MASKED + 1] = z.c1;
    ws[WS_MASKED + 2] = z.c2;
  }
}

// ------------------------------------------------------------------------------------------------
// streaming skeleton
// ------------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned char v4b __attribute__((ext_vector_type(4)));

struct StreamArgs {
  const float* in;
  float* out;
  const void* mask;   // [B,1,H,W] u8 or f32, or NULL
  const float* coef;  // workspace base (per-image stride below) or NULL
  unsigned coef_stride;
  unsigned n;                 // elements per plane in units of VEC floats
  unsigned blocks_per_image;  // chunks per image
  unsigned n_blocks;          // total
  int no_mem;  // diagnostics: synthesise inputs, suppress stores (VALU-only timing; results undefined)
  unsigned W, H;  // image size in pixels (ops that need pixel coordinates)
  int op_flag;    // op-specific (trispace: residual only)
  const uint8_t* white;  // FMT_U8HWC only: [B,H,W] 'L' mask of infer.py:39,46 (out*m + (1-m), m = L/255) or NULL
  unsigned units, segs;  // Op::kRowTiles only: VEC-pixel groups per image row, blocks per row
};
// pixel formats at the kernel's edges: planar float32 NCHW (what the reference's tensors are), or the file edge's
// interleaved bytes (PIL HWC uint8 in, to_pil_image / astype('uint8') out) converted in registers
#define FMT_F32CHW 0
#define FMT_U8HWC 1

template <int VEC>
struct Pack;
template <>
struct Pack<4> {
  typedef v4f T;
  typedef v4b M;
};
template <>
struct Pack<1> {
  typedef float T;
  typedef unsigned char M;
};

// Non-temporal must be a COMPILE-TIME choice: with a run-time `nt ? __builtin_nontemporal_load(p) : *p`
// the optimiser merges the two loads and drops the hint (no `nt` instruction was ever emitted that way).
// Streaming data is touched once; `nt` keeps it from displacing lines in L2/MALL: +10 % on the 3-plane
// copy pattern of this path (tools/ubench/copy3.hip: 5.56 -> 6.15 TB/s).
template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void st(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
__device__ __forceinline__ float lane(const v4f& v, int e) { return v[e]; }
__device__ __forceinline__ float lane(const float& v, int) { return v; }
__device__ __forceinline__ void set_lane(v4f& v, int e, float x) { v[e] = x; }
__device__ __forceinline__ void set_lane(float& v, int, float x) { v = x; }
__device__ __forceinline__ unsigned char lane_b(const v4b& v, int e) { return v[e]; }
__device__ __forceinline__ unsigned char lane_b(const unsigned char& v, int) { return v; }
__device__ __forceinline__ float mlane(const v4b& v, int e) { return v[e] ? 1.0f : 0.0f; }
__device__ __forceinline__ float mlane(const unsigned char& v, int) { return v ? 1.0f : 0.0f; }

// Op contract:  struct Op { struct K {...}; static K load(const float* ws_image);  // uniform
//                           static Px apply(Px in, float m, const K&); static constexpr bool kMask; }
//
// One block = one tile of 256*U vectors of ONE image: all loads of the tile are issued up front (7*U
// independent 16-byte loads per lane), then the arithmetic, then the stores; latency is covered by the
// other resident waves (a two-deep register prefetch loop was measured and lost to plain occupancy on every
// kernel here: profiles/sweep_r01.md).  Loads clamp their index instead of branching (no divergent
// prologue, no zero-fill); stores are guarded.
template <int VEC, int U, int MK>
struct Tile {
  typename Pack<VEC>::T x0[U], x1[U], x2[U];
  typename Pack<VEC>::T mf[MK == CURL_MASK_F32 ? U : 1];
  typename Pack<VEC>::M mb[MK == CURL_MASK_U8 ? U : 1];
  typename Pack<VEC>::M wm[U];  // FMT_U8HWC: white-background mask bytes
#ifdef CURL_DIAG_NO_DEP
  typename Pack<VEC>::T raw[U];
#endif
};

// a*b rounded, then + c rounded -- what two eager ops produce (HIP's __fmul_rn/__fadd_rn are plain operators and
// get contracted into one fma; the pragma is what hipcc's default fast-honor-pragmas mode respects)
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
  float t = a * b;
  return t + c;
}
__device__ __forceinline__ float byte_of(unsigned w, int k) { return (float)((w >> (8 * k)) & 0xffu); }  // v_cvt_f32_ubyteK
// 4 interleaved RGB pixels = 3 dwords [R0 G0 B0 R1][G1 B1 R2 G2][B2 R3 G3 B3] -> three planes of 4 floats in [0,1]
__device__ __forceinline__ void unpack_rgb4(unsigned w0, unsigned w1, unsigned w2, v4f& r, v4f& g, v4f& b) {
  r = v4f{byte_of(w0, 0), byte_of(w0, 3), byte_of(w1, 2), byte_of(w2, 1)};
  g = v4f{byte_of(w0, 1), byte_of(w1, 0), byte_of(w1, 3), byte_of(w2, 2)};
  b = v4f{byte_of(w0, 2), byte_of(w1, 1), byte_of(w2, 0), byte_of(w2, 3)};
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = u8_to_unit(r[e]), g[e] = u8_to_unit(g[e]), b[e] = u8_to_unit(b[e]);
}

template <int VEC, int U, int MK, bool NT, int FMT>
__device__ __forceinline__ void load_tile(Tile<VEC, U, MK>& t, const StreamArgs& a, const typename Pack<VEC>::T* p0,
                                          size_t plane, size_t mask_off, unsigned base) {
  typedef typename Pack<VEC>::T T;
  typedef typename Pack<VEC>::M M;
  if constexpr (FMT == FMT_U8HWC) {
    // p0 = this image's interleaved bytes.  All loads first (raw words), conversions after.
    unsigned raw[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned i = min(base + u * 256u, a.n - 1u);
      if constexpr (VEC == 4) {
        const unsigned* w = reinterpret_cast<const unsigned*>(p0) + 3 * (size_t)i;  // 12 bytes per lane, contiguous
        raw[u][0] = ld<NT>(w), raw[u][1] = ld<NT>(w + 1), raw[u][2] = ld<NT>(w + 2);
      } else {
        const uint8_t* w = reinterpret_cast<const uint8_t*>(p0) + 3 * (size_t)i;
        raw[u][0] = w[0], raw[u][1] = w[1], raw[u][2] = w[2];
      }
      if (MK == CURL_MASK_U8) t.mb[u] = ld<NT>(reinterpret_cast<const M*>(a.mask) + mask_off + i);
      if (MK == CURL_MASK_F32) t.mf[u] = ld<NT>(reinterpret_cast<const T*>(a.mask) + mask_off + i);
      if (a.white) t.wm[u] = ld<NT>(reinterpret_cast<const M*>(a.white) + mask_off + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (VEC == 4) {
        unpack_rgb4(raw[u][0], raw[u][1], raw[u][2], t.x0[u], t.x1[u], t.x2[u]);
      } else {
        t.x0[u] = u8_to_unit((float)raw[u][0]), t.x1[u] = u8_to_unit((float)raw[u][1]);
        t.x2[u] = u8_to_unit((float)raw[u][2]);
      }
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned i = min(base + u * 256u, a.n - 1u);
    if (a.no_mem) {
      float f = (float)(i & 1023u) * (1.0f / 1024.0f);
      t.x0[u] = T(f);
      t.x1[u] = T(1.0f - f);
      t.x2[u] = T(0.5f * f + 0.1f);
      if (MK == CURL_MASK_U8) t.mb[u] = typename Pack<VEC>::M(1);
      if (MK == CURL_MASK_F32) t.mf[u] = T(1.0f);
      continue;
    }
    t.x0[u] = ld<NT>(p0 + i);
    t.x1[u] = ld<NT>(p0 + plane + i);
    t.x2[u] = ld<NT>(p0 + 2 * plane + i);
#ifdef CURL_DIAG_NO_DEP
    // experiment build only (tools/ab.py): the loads are issued and must land before the stores, but the
    // arithmetic runs on synthetic values -- separates "waiting for data" from "sharing the chip with traffic"
    t.raw[u] = t.x0[u] + t.x1[u] + t.x2[u];
    {
      float f = (float)(i & 1023u) * (1.0f / 1024.0f);
      t.x0[u] = T(f), t.x1[u] = T(1.0f - f), t.x2[u] = T(0.5f * f + 0.1f);
    }
#endif
    if (MK == CURL_MASK_U8) t.mb[u] = ld<NT>(reinterpret_cast<const M*>(a.mask) + mask_off + i);
    if (MK == CURL_MASK_F32) t.mf[u] = ld<NT>(reinterpret_cast<const T*>(a.mask) + mask_off + i);
  }
}

template <class Op, int VEC, int U, int MK, bool NT, int FMT>
__device__ __forceinline__ void compute_store(const Tile<VEC, U, MK>& t, const StreamArgs& a,
                                              typename Pack<VEC>::T* q0, size_t plane, unsigned base,
                                              const typename Op::K& k, bool valid) {
  typedef typename Pack<VEC>::T T;
  constexpr bool kBinary = (MK != CURL_MASK_F32);  // none / bool / uint8: the mask is exactly 0 or 1
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned i = base + u * 256u;
    T y0, y1, y2;
    bool live = true, full = false;
    if (Op::kMask && MK != CURL_MASK_NONE) {
      // Masks are foreground masks: whole waves are often masked out.  Where every lane of the wave has
      // m == 0 for all its pixels the result is a constant (0 for the layer, lab2rgb(0,0,0) for the Lab
      // stage) and the arithmetic is skipped -- a wave-uniform branch (ballot), no divergence.
      bool lane_live = false;
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        lane_live |= (MK == CURL_MASK_U8) ? (mlane(t.mb[u], e) != 0.0f) : (lane(t.mf[u], e) != 0.0f);
      live = __builtin_amdgcn_ballot_w64(lane_live) != 0ull;
      if constexpr (MK == CURL_MASK_U8) {
        // ... and just as often fully inside the foreground: every mask byte of the wave non-zero.  Then the
        // mask is the constant 1 and its conversions, multiplies and blends go (wave-uniform branch again).
        unsigned w;
        if constexpr (VEC == 4) {
          w = __builtin_bit_cast(unsigned, t.mb[u]);
          w = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;  // bit 7 of each byte: byte != 0
          w ^= 0x80808080u;                                           // 0 iff all four are non-zero
        } else {
          w = t.mb[u] ? 0u : 1u;
        }
        full = __builtin_amdgcn_ballot_w64(w == 0u) == ~0ull;
      }
    }
    if (live && full) {
      PxN<VEC> px;
      float one[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        px.c0[e] = lane(t.x0[u], e);
        px.c1[e] = lane(t.x1[u], e);
        px.c2[e] = lane(t.x2[u], e);
        one[e] = 1.0f;
      }
      Op::template apply_n<true, VEC>(px, one, k, i * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        set_lane(y0, e, px.c0[e]);
        set_lane(y1, e, px.c1[e]);
        set_lane(y2, e, px.c2[e]);
      }
    } else if (live) {
      PxN<VEC> px;
      float mm[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        px.c0[e] = lane(t.x0[u], e);
        px.c1[e] = lane(t.x1[u], e);
        px.c2[e] = lane(t.x2[u], e);
        mm[e] = 1.0f;
        if (MK == CURL_MASK_U8) mm[e] = mlane(t.mb[u], e);
        if (MK == CURL_MASK_F32) mm[e] = lane(t.mf[u], e);
      }
      Op::template apply_n<kBinary, VEC>(px, mm, k, i * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        Px o{px.c0[e], px.c1[e], px.c2[e]};
        if (Op::kBlendMaskedOut && MK == CURL_MASK_U8) {
          // the binary specialisation leaves m == 0 pixels to us: overwrite with the masked-out constant
          Px z = Op::masked_out(k);
          int keep = opaque(-(int)(mm[e] != 0.0f));
          o.c0 = blend(keep, o.c0, z.c0);
          o.c1 = blend(keep, o.c1, z.c1);
          o.c2 = blend(keep, o.c2, z.c2);
        }
        set_lane(y0, e, o.c0);
        set_lane(y1, e, o.c1);
        set_lane(y2, e, o.c2);
      }
    } else {
      Px z = Op::masked_out(k);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        set_lane(y0, e, z.c0);
        set_lane(y1, e, z.c1);
        set_lane(y2, e, z.c2);
      }
    }
    bool keep = true;
#ifdef CURL_DIAG_NO_DEP
    keep = lane(t.raw[u], 0) != -123.0f;  // always true; ties the stores to the loads
#endif
    if (a.no_mem) {  // every output feeds the (never true) condition, so nothing can be sunk or dropped
      float chk = 0.0f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) chk += lane(y0, e) + lane(y1, e) + lane(y2, e);
      keep = (chk == -123.0f);
    }
    if constexpr (FMT == FMT_U8HWC) {
      unsigned q[3][VEC];  // [channel][pixel]
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float c[3] = {lane(y0, e), lane(y1, e), lane(y2, e)};
        if (a.white) {  // infer.py:46: out*m + (1-m), two roundings like the eager ops (no contraction)
          float m = u8_to_unit((float)lane_b(t.wm[u], e));
          float one_minus = 1.0f - m;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) c[ch] = mul_then_add(c[ch], m, one_minus);
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) q[ch][e] = unit_to_u8(c[ch]);
      }
      if (i < a.n && valid) {
        if constexpr (VEC == 4) {
          unsigned* w = reinterpret_cast<unsigned*>(q0) + 3 * (size_t)i;
          st<NT>(w, q[0][0] | (q[1][0] << 8) | (q[2][0] << 16) | (q[0][1] << 24));
          st<NT>(w + 1, q[1][1] | (q[2][1] << 8) | (q[0][2] << 16) | (q[1][2] << 24));
          st<NT>(w + 2, q[2][2] | (q[0][3] << 8) | (q[1][3] << 16) | (q[2][3] << 24));
        } else {
          uint8_t* w = reinterpret_cast<uint8_t*>(q0) + 3 * (size_t)i;
          w[0] = (uint8_t)q[0][0], w[1] = (uint8_t)q[1][0], w[2] = (uint8_t)q[2][0];
        }
      }
    } else if (i < a.n && keep && valid) {
      st<NT>(q0 + i, y0);
      st<NT>(q0 + plane + i, y1);
      st<NT>(q0 + 2 * plane + i, y2);
    }
  }
}

template <class Op, int VEC, int U, int MK, bool NT, int FMT = FMT_F32CHW>
__global__ __launch_bounds__(256, Op::kMinWavesPerSimd) void stream_kernel(StreamArgs a) {
  typedef typename Pack<VEC>::T T;
  // grid = (chunks per image, images): both indices are SGPRs, no division
  const unsigned img = blockIdx.y;
  const unsigned chunk = blockIdx.x;
  __builtin_assume(a.n <= (1u << 28));  // H*W <= 2^30 (checked on the host): byte offsets fit 32 bits
  const float* table = a.coef ? a.coef + (size_t)img * a.coef_stride : nullptr;
  __shared__ float s_table[Op::kLdsFloats > 0 ? Op::kLdsFloats : 1];
  unsigned row = 0, base = chunk * (256u * U) + threadIdx.x;
  bool valid = true;
  if constexpr (Op::kRowTiles) {
    row = chunk / a.segs;  // wave-uniform, once per block
    const unsigned in_row = (chunk - row * a.segs) * blockDim.x + threadIdx.x;
    valid = in_row < a.units;                          // lanes past the row end: loads clamp, stores are dropped
    base = row * a.units + min(in_row, a.units - 1u);
  }
  if constexpr (Op::kLdsFloats > 0) {
    // a per-image table too large for SGPRs (1134 polynomial coefficients): one coalesced copy into LDS, then
    // every lane reads the same address (broadcast ds_read_b128, conflict-free)
    if constexpr (Op::kRowTiles) {
      for (unsigned i = threadIdx.x; i < (unsigned)Op::kLdsFloats; i += blockDim.x) s_table[i] = Op::stage_value(table, i, row, a);
    } else {
      for (int i = threadIdx.x; i < Op::kLdsFloats; i += 256) s_table[i] = table[Op::stage_index(i)];
    }
    __syncthreads();
    table = s_table;
  }
  const typename Op::K k = Op::load(table, a);
  const size_t plane = (size_t)a.n;
  // an image is 3 planes of `plane` vectors, or (FMT_U8HWC) plane*VEC pixels of 3 bytes = 3*plane*VEC bytes
  const size_t image_bytes = (FMT == FMT_U8HWC) ? 3 * plane * VEC : 3 * plane * sizeof(T);
  const T* p0 = reinterpret_cast<const T*>(reinterpret_cast<const char*>(a.in) + (size_t)img * image_bytes);
  T* q0 = reinterpret_cast<T*>(reinterpret_cast<char*>(a.out) + (size_t)img * image_bytes);
  const size_t mask_off = (size_t)img * plane;
  Tile<VEC, U, MK> t;
  load_tile<VEC, U, MK, NT, FMT>(t, a, p0, plane, mask_off, base);
  compute_store<Op, VEC, U, MK, NT, FMT>(t, a, q0, plane, base, k, valid);
}

// ------------------------------------------------------------------------------------------------
// ops
// ------------------------------------------------------------------------------------------------
struct NoK {};
struct OpDefaults {
  static constexpr bool kSingleTileShape = false;
  static constexpr int kLdsFloats = 0;  // per-image table the block stages in LDS before the tile (0 = none)
  static constexpr int kMinWavesPerSimd = 1;  // __launch_bounds__ second argument (register budget)
  // true: a block never crosses an image row (grid.x = blocks per row x rows, blockDim.x <= 256 follows the row
  // width) and the LDS table is built per row by Op::stage_value(table, i, row, args)
  static constexpr bool kRowTiles = false;
};
#define CONVERTER_OP(NAME, FN)                                                           \
  struct NAME : OpDefaults {                                                             \
    typedef NoK K;                                                                       \
    static constexpr bool kMask = false;                                                 \
    static constexpr int kUnroll = 2;                                                    \
    static __device__ __forceinline__ K load(const float*, const StreamArgs&) { return K{}; } \
    template <bool, int N>                                                               \
    static __device__ __forceinline__ void apply_n(PxN<N>& p, const float (&)[N], const K&, unsigned) { FN(p); } \
    static constexpr bool kBlendMaskedOut = false;                                       \
    static __device__ __forceinline__ Px masked_out(const K&) { return Px{0.0f, 0.0f, 0.0f}; } \
  };
CONVERTER_OP(OpRgb2Lab, rgb2lab_n<N>)
CONVERTER_OP(OpLab2Rgb, lab2rgb_n<N>)
CONVERTER_OP(OpRgb2Hsv, rgb2hsv_n<N>)
CONVERTER_OP(OpHsv2Rgb, hsv2rgb_n<N>)

__device__ __forceinline__ Affine load_affine(const float* ws, int c) {
  return Affine{ws[WS_COEF + 2 * c], ws[WS_COEF + 2 * c + 1]};
}

struct OpAdjust3 : OpDefaults {  // adjust_rgb / adjust_lab, affine form
  struct K {
    Affine k[3];
  };
  static constexpr bool kMask = false;
  static constexpr int kUnroll = 2;
  static __device__ __forceinline__ K load(const float* ws, const StreamArgs&) {
    K k;
#pragma unroll
    for (int c = 0; c < 3; ++c) k.k[c] = load_affine(ws, c);
    return k;
  }
  template <bool, int N>
  static __device__ __forceinline__ void apply_n(PxN<N>& p, const float (&)[N], const K& k, unsigned) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      Px o = adjust3(Px{p.c0[i], p.c1[i], p.c2[i]}, k.k[0], k.k[1], k.k[2]);
      p.c0[i] = o.c0, p.c1[i] = o.c1, p.c2[i] = o.c2;
    }
  }
  static constexpr bool kBlendMaskedOut = false;
  static __device__ __forceinline__ Px masked_out(const K&) { return Px{0.0f, 0.0f, 0.0f}; }
};
struct OpAdjustHsv : OpDefaults {
  struct K {
    Affine k[4];
  };
  static constexpr bool kMask = false;
  static constexpr int kUnroll = 2;
  static __device__ __forceinline__ K load(const float* ws, const StreamArgs&) {
    K k;
#pragma unroll
    for (int c = 0; c < 4; ++c) k.k[c] = load_affine(ws, c);
    return k;
  }
  template <bool, int N>
  static __device__ __forceinline__ void apply_n(PxN<N>& p, const float (&)[N], const K& k, unsigned) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      Px o = adjust_hsv4(Px{p.c0[i], p.c1[i], p.c2[i]}, k.k[0], k.k[1], k.k[2], k.k[3]);
      p.c0[i] = o.c0, p.c1[i] = o.c1, p.c2[i] = o.c2;
    }
  }
  static constexpr bool kBlendMaskedOut = false;
  static __device__ __forceinline__ Px masked_out(const K&) { return Px{0.0f, 0.0f, 0.0f}; }
};
struct OpLabStage : OpDefaults {
  struct K {
    Affine k[3];
    Px masked;
  };
  static constexpr bool kMask = true;
  static constexpr int kUnroll = 1;  // arithmetic-heavy: occupancy beats per-lane ILP (profiles/sweep_r01.md)
  static __device__ __forceinline__ K load(const float* ws, const StreamArgs&) {
    K k;
#pragma unroll
    for (int c = 0; c < 3; ++c) k.k[c] = load_affine(ws, c);
    k.masked = Px{ws[WS_MASKED], ws[WS_MASKED + 1], ws[WS_MASKED + 2]};
    return k;
  }
  template <bool BINARY, int N>
  static __device__ __forceinline__ void apply_n(PxN<N>& p, const float (&m)[N], const K& k, unsigned) {
    lab_stage_n<BINARY, N>(p, m, k.k);
  }
  static constexpr bool kBlendMaskedOut = true;  // lab_stage<true> computes m == 0 pixels as if m == 1
  static __device__ __forceinline__ Px masked_out(const K& k) { return k.masked; }
};
struct OpLayer : OpDefaults {
  typedef LayerCoef K;
  static constexpr bool kMask = true;
  static constexpr int kUnroll = 1;
  static __device__ __forceinline__ K load(const float* ws, const StreamArgs&) {
    K k;
#pragma unroll
    for (int c = 0; c < 3; ++c) k.lab[c] = load_affine(ws, c);
#pragma unroll
    for (int c = 0; c < 3; ++c) k.rgb[c] = load_affine(ws, 3 + c);
#pragma unroll
    for (int c = 0; c < 4; ++c) k.hsv[c] = load_affine(ws, 6 + c);
    return k;
  }
  template <bool BINARY, int N>
  static __device__ __forceinline__ void apply_n(PxN<N>& p, const float (&m)[N], const K& k, unsigned) {
    curl_layer_n<BINARY, N>(p, m, k);
  }
  static constexpr bool kBlendMaskedOut = false;  // curl_layer ends in `* m` for every mask kind
  static __device__ __forceinline__ Px masked_out(const K&) { return Px{0.0f, 0.0f, 0.0f}; }
};

// TriSpaceRegNet.generate_residual (+ generate_image), model.py:499-520: three degree-4 polynomial layers in
// RGB / Lab / HSV + converters, one pass.  The 9 x NC coefficients of the image are staged in LDS by the block
// (kLdsFloats) and reach the packed FMAs as broadcast ds_read_b128 -> VGPR halves (op_sel).  Reading them
// through a uniform global pointer instead made hipcc hoist all 1134 scalar loads and spill SGPRs into VGPR
// lanes (3000 v_readlane/v_writelane per thread, 4.4 ms per batch).
template <int V>
struct OpTriSpace : OpDefaults {
  struct K {
    const float* coef;
    unsigned W;
    float fW, fH;
    bool residual_only;
  };
  static constexpr bool kMask = false;
  static constexpr int kUnroll = 1;
  static constexpr bool kSingleTileShape = true;
  static constexpr int kLdsFloats = 9 * PolyEval<V>::kCoeffs;
  static constexpr int kMinWavesPerSimd = CURL_TRISPACE_WAVES;
  // LDS position p of polynomial q holds the coefficient the Horner scheme consumes p-th
  static __device__ __forceinline__ int stage_index(int i) {
    constexpr int NC = PolyEval<V>::kCoeffs;
    int q = i / NC, pos = i - q * NC;
    return q * NC + PolyEval<V>::order(pos);
  }
  static constexpr bool kBlendMaskedOut = false;
  static __device__ __forceinline__ K load(const float* coef_img, const StreamArgs& a) {
    return K{coef_img, a.W, (float)a.W, (float)a.H, a.op_flag != 0};
  }
  template <bool, int N>
  static __device__ __forceinline__ void apply_n(PxN<N>& p, const float (&)[N], const K& k, unsigned pix0) {
    float xw[N], yh[N];
    if (V == 5) {  // cat_coords (model.py:487-497): column / width, row / height, true division
      unsigned row = pix0 / k.W, col = pix0 - row * k.W;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        xw[i] = (float)col / k.fW;
        yh[i] = (float)row / k.fH;
        if (++col == k.W) col = 0, ++row;
      }
    }
    trispace_n<V, N, true>(p, xw, yh, k.coef, k.residual_only);
  }
  static __device__ __forceinline__ Px masked_out(const K&) { return Px{0.0f, 0.0f, 0.0f}; }
};

// The spatial polynomial path, one image row per block.  cat_coords' y = row/height is the same for every pixel
// of a row, so the block folds y into the coefficients first (9 polynomials x 70 collapsed coefficients, <= 4
// FMAs each, a handful per thread; collapse_coef) and leaves them in LDS in the consumption order of the
// 4-variable Horner scheme: a pixel then costs 9 x 69 FMAs instead of 9 x 125.
struct OpTriSpaceRows : OpDefaults {
  struct K {
    const float* coef;
    unsigned W;
    float fW, rW;
    bool residual_only;
  };
  static constexpr bool kMask = false;
  static constexpr int kUnroll = 1;
  static constexpr bool kSingleTileShape = true;
  static constexpr bool kRowTiles = true;
  static constexpr int kLdsFloats = 9 * PolyEval<4>::kCoeffs;
  static constexpr int kMinWavesPerSimd = CURL_TRISPACE_WAVES;
  static __device__ __forceinline__ float stage_value(const float* coef_img, unsigned i, unsigned row, const StreamArgs& a) {
    constexpr unsigned NC4 = PolyEval<4>::kCoeffs, NC5 = PolyEval<5>::kCoeffs;
    unsigned q = i / NC4, pos = i - q * NC4;
    return collapse_coef(coef_img + q * NC5, (int)pos, (float)row / (float)a.H);  // row / height: true division
  }
  static constexpr bool kBlendMaskedOut = false;
  static __device__ __forceinline__ K load(const float* coef_row, const StreamArgs& a) {
    return K{coef_row, a.W, (float)a.W, 1.0f / (float)a.W, a.op_flag != 0};
  }
  template <bool, int N>
  static __device__ __forceinline__ void apply_n(PxN<N>& p, const float (&)[N], const K& k, unsigned pix0) {
    float xw[N], yh[N];
    unsigned col = pix0 % k.W;  // a row tile never wraps
#pragma unroll
    for (int i = 0; i < N; ++i) xw[i] = div_small((float)(col + i), k.fW, k.rW), yh[i] = 0.0f;  // column / width
    trispace_n<4, N, true>(p, xw, yh, k.coef, k.residual_only);
  }
  static __device__ __forceinline__ Px masked_out(const K&) { return Px{0.0f, 0.0f, 0.0f}; }
};

// ChannelPolyLayer / Deg4MobilePolyLayer forward (model.py:295-333, 399-415): img [B,V,H,W] -> [B,3,H,W].
template <int V>
__global__ __launch_bounds__(256) void poly_layer_kernel(const float* in, const float* coeffs, float* out, unsigned HW) {
  constexpr int NC = PolyEval<V>::kCoeffs;
  const unsigned img = blockIdx.y;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= HW) return;
  const float* p = in + (size_t)img * V * HW + i;
  float vars[V][1], o[3][1];
#pragma unroll
  for (int k = 0; k < V; ++k) vars[k][0] = p[(size_t)k * HW];
  poly3_n<V, 1>(o, vars, coeffs + (size_t)img * 3 * NC);
  float* q = out + (size_t)img * 3 * HW + i;
#pragma unroll
  for (int c = 0; c < 3; ++c) q[(size_t)c * HW] = o[c][0];
}

// ------------------------------------------------------------------------------------------------
// curve chain with the knots in LDS: reference summation order, or paper-style PWL lookup
// ------------------------------------------------------------------------------------------------
#define CHAIN_MAX 4
struct ChainArgs {
  const float* in;
  float* out;
  const float* knots;  // exp'd knots; step s of image b at knots + b*knot_stride + knot_off[s]
  unsigned knot_stride;
  int n_steps;
  int K;  // knots per curve (same for every step)
  int knot_off[CHAIN_MAX];
  int cin[CHAIN_MAX], cout[CHAIN_MAX];
  unsigned n, blocks_per_image, n_blocks;
  int mode;  // 0 affine (collapsed in the prologue), 1 exact order, 2 PWL
};

template <int VEC, int U>
__global__ __launch_bounds__(256) void chain_kernel(ChainArgs a) {
  typedef typename Pack<VEC>::T T;
  __shared__ float sC[CHAIN_MAX][CURL_MAX_KNOTS];
  __shared__ float sS[CHAIN_MAX][CURL_MAX_KNOTS];
  __shared__ float sAB[CHAIN_MAX][2];
  const unsigned img = blockIdx.y;
  const unsigned chunk = blockIdx.x;
  const int K = a.K;
  // stage this image's knots and slopes in LDS
  for (int i = threadIdx.x; i < a.n_steps * K; i += 256) {
    int s = i / K, j = i - s * K;
    const float* C = a.knots + (size_t)img * a.knot_stride + a.knot_off[s];
    float c = C[j];
    sC[s][j] = c;
    if (j + 1 < K) sS[s][j] = C[j + 1] - c;  // curves.py:19
  }
  __syncthreads();
  if (a.mode == 0 && (int)threadIdx.x < a.n_steps) {
    int s = threadIdx.x;
    float creg;
    collapse_curve(sC[s], K, sAB[s][0], sAB[s][1], creg);
  }
  if (a.mode == 0) __syncthreads();

  const size_t plane = (size_t)a.n;
  const T* p0 = reinterpret_cast<const T*>(a.in) + (size_t)img * 3 * plane;
  T* q0 = reinterpret_cast<T*>(a.out) + (size_t)img * 3 * plane;
  const unsigned base = chunk * (256u * U) + threadIdx.x;
  T x0[U], x1[U], x2[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned i = base + u * 256u;
    if (i < a.n) {
      x0[u] = ld<true>(p0 + i);
      x1[u] = ld<true>(p0 + plane + i);
      x2[u] = ld<true>(p0 + 2 * plane + i);
    }
  }
  const float S = (float)(K - 1);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned i = base + u * 256u;
    if (i < a.n) {
      T y0, y1, y2;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float c0 = lane(x0[u], e), c1 = lane(x1[u], e), c2 = lane(x2[u], e);
        for (int s = 0; s < a.n_steps; ++s) {
          const int ci = a.cin[s], co = a.cout[s];
          float xin = (ci == 0) ? c0 : (ci == 1) ? c1 : c2;
          float scale;
          if (a.mode == 1)
            scale = scale_exact(xin, sS[s], sC[s][0], K - 2, S);
          else if (a.mode == 2)
            scale = scale_pwl(xin, sC[s], sS[s], K);
          else
            scale = fmaf(sAB[s][1], xin, sAB[s][0]);
          // curves.py:35-36: multiply the output channel, clamp the whole image
          float m0 = (co == 0) ? scale : 1.0f, m1 = (co == 1) ? scale : 1.0f, m2 = (co == 2) ? scale : 1.0f;
          c0 = clamp01(c0 * m0);
          c1 = clamp01(c1 * m1);
          c2 = clamp01(c2 * m2);
        }
        set_lane(y0, e, c0);
        set_lane(y1, e, c1);
        set_lane(y2, e, c2);
      }
      st<true>(q0 + i, y0);
      st<true>(q0 + plane + i, y1);
      st<true>(q0 + 2 * plane + i, y2);
    }
  }
}

// reg[b] += sum of squared slope differences of C[b,:]  (curves.py:19,24), C already exp'd.
__global__ void curve_reg_kernel(const float* C, float* reg, int B, int K) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float ca, cb, creg;
  collapse_curve(C + (size_t)b * K, K, ca, cb, creg);
  reg[b] += creg;
}

// ------------------------------------------------------------------------------------------------
// backward of the fused layer
// ------------------------------------------------------------------------------------------------
#define BWD_NACC 20  // P[10], Q[10]
struct BwdArgs {
  const float* in;
  const float* gout;
  float* gin;         // nullable
  const void* mask;
  const float* coef;  // workspace (prep output)
  float* partial;     // [n_blocks][BWD_NACC] block partial sums of P,Q
  unsigned coef_stride, n, blocks_per_image, n_blocks;
};

// wave-wide sum in 6 DPP adds (VALU rate; __shfl_xor compiles to ds_bpermute + a full wait each): the row's 16
// lanes by quad_perm / half-mirror / mirror, then row_bcast15 and row_bcast31.  The total is in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
// stage-major over a lane's M accumulators: consecutive DPP adds are independent (no wait states between them)
template <int CTRL, int ROW_MASK, int R, int M>
__device__ __forceinline__ void dpp_add_all(float (&x)[R][M]) {
#pragma unroll
  for (int o = 0; o < R; ++o)
#pragma unroll
    for (int j = 0; j < M; ++j) x[o][j] = dpp_add<CTRL, ROW_MASK>(x[o][j]);
  CURL_FENCE();
}
template <int R, int M>
__device__ __forceinline__ void wave_sum_lane63(float (&x)[R][M]) {
  dpp_add_all<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  dpp_add_all<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  dpp_add_all<0x141, 0xF>(x);  // row_half_mirror
  dpp_add_all<0x140, 0xF>(x);  // row_mirror: every lane of a row holds the row's sum
  dpp_add_all<0x142, 0xA>(x);  // row_bcast15 into rows 1 and 3
  dpp_add_all<0x143, 0xC>(x);  // row_bcast31 into rows 2 and 3
}
// One tile per block, like the forward.  Per-pixel reverse mode (curl_math_bwd.h) recomputes the forward
// chain in registers; the 20 per-image curve sums are reduced wave -> LDS -> one row of `partial` per block
// (no float atomics: the second pass sums the rows in a fixed order in float64, so results are reproducible).
template <int VEC, int MK>
__global__ __launch_bounds__(256) void layer_bwd_kernel(BwdArgs a) {
  typedef typename Pack<VEC>::T T;
  typedef typename Pack<VEC>::M M;
  __shared__ float sPart[4][BWD_NACC];
  const unsigned img = blockIdx.y;
  const unsigned chunk = blockIdx.x;
  const unsigned bid = img * a.blocks_per_image + chunk;
  const LayerCoef k = OpLayer::load(a.coef + (size_t)img * a.coef_stride, StreamArgs{});
  const size_t plane = (size_t)a.n;
  const T* p0 = reinterpret_cast<const T*>(a.in) + (size_t)img * 3 * plane;
  const T* g0 = reinterpret_cast<const T*>(a.gout) + (size_t)img * 3 * plane;
  const unsigned i = chunk * 256u + threadIdx.x;
  const unsigned ic = min(i, a.n - 1u);
  const bool valid = i < a.n;
  T x0 = p0[ic], x1 = p0[plane + ic], x2 = p0[2 * plane + ic];
  T w0 = g0[ic], w1 = g0[plane + ic], w2 = g0[2 * plane + ic];
  T mf;
  M mb;
  if (MK == CURL_MASK_U8) mb = reinterpret_cast<const M*>(a.mask)[(size_t)img * plane + ic];
  if (MK == CURL_MASK_F32) mf = reinterpret_cast<const T*>(a.mask)[(size_t)img * plane + ic];
  float acc[BWD_NACC];
#pragma unroll
  for (int c = 0; c < BWD_NACC; ++c) acc[c] = 0.0f;
  T y0, y1, y2;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    float m = 1.0f;
    if (MK == CURL_MASK_U8) m = mlane(mb, e);
    if (MK == CURL_MASK_F32) m = lane(mf, e);
    if (!valid) m = 0.0f;  // lanes past the end contribute nothing (m multiplies every path to P, Q)
    float P[10], Q[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) P[c] = Q[c] = 0.0f;
    Px gi = curl_layer_bwd(Px{lane(x0, e), lane(x1, e), lane(x2, e)}, m, k, Px{lane(w0, e), lane(w1, e), lane(w2, e)},
                           P, Q);
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      acc[c] += P[c];
      acc[10 + c] += Q[c];
    }
    set_lane(y0, e, gi.c0);
    set_lane(y1, e, gi.c1);
    set_lane(y2, e, gi.c2);
  }
  if (a.gin && valid) {
    T* q0 = reinterpret_cast<T*>(a.gin) + (size_t)img * 3 * plane;
    q0[i] = y0;
    q0[plane + i] = y1;
    q0[2 * plane + i] = y2;
  }
  // wave sum by DPP (total in lane 63), then the 4 waves through LDS
  float accw[1][BWD_NACC];
#pragma unroll
  for (int c = 0; c < BWD_NACC; ++c) accw[0][c] = acc[c];
  wave_sum_lane63(accw);
  const int wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
  if (lane_id == 63) {
#pragma unroll
    for (int c = 0; c < BWD_NACC; ++c) sPart[wave][c] = accw[0][c];
  }
  __syncthreads();
  if (threadIdx.x < BWD_NACC) {
    int c = threadIdx.x;
    a.partial[(size_t)bid * BWD_NACC + c] = (sPart[0][c] + sPart[1][c]) + (sPart[2][c] + sPart[3][c]);
  }
}

struct KnotsBwdArgs {
  const float* ws;       // prep output (exp'd knots at WS_KNOTS)
  const float* partial;  // [B][blocks_per_image][BWD_NACC]
  const float* greg;     // nullable [B]
  float* graw[3];        // gradients shaped like rawL, rawR, rawH
  int K[3];
  unsigned ws_stride, blocks_per_image;
};

// One block per image: fixed-order float64 reduction of the block partials, then the chain rule
// (P, Q, d reg) -> raw knots of each of the 10 curves (curl_math_bwd.h: knots_bwd).
__global__ __launch_bounds__(256) void knots_bwd_kernel(KnotsBwdArgs a) {
  __shared__ double sAcc[256];
  __shared__ double sPQ[BWD_NACC];
  const unsigned b = blockIdx.x;
  const float* part = a.partial + (size_t)b * a.blocks_per_image * BWD_NACC;
  for (int c = 0; c < BWD_NACC; ++c) {
    double v = 0.0;
    for (unsigned i = threadIdx.x; i < a.blocks_per_image; i += 256) v += (double)part[(size_t)i * BWD_NACC + c];
    sAcc[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sAcc[threadIdx.x] += sAcc[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) sPQ[c] = sAcc[0];
    __syncthreads();
  }
  const int c = threadIdx.x;
  if (c < 10) {
    int s = c < 3 ? 0 : (c < 6 ? 1 : 2);
    int local = c - (s == 0 ? 0 : (s == 1 ? 3 : 6));
    int K = a.K[s];
    int off = (s == 0 ? 0 : (s == 1 ? 3 * a.K[0] : 3 * a.K[0] + 3 * a.K[1])) + local * K;
    const float* C = a.ws + (size_t)b * a.ws_stride + WS_KNOTS + off;
    int per_img = (s == 2 ? 4 : 3) * K;
    float* g = a.graw[s] + (size_t)b * per_img + local * K;
    knots_bwd(C, K, sPQ[c], sPQ[10 + c], a.greg ? (double)a.greg[b] : 0.0, g);
  }
}

// ------------------------------------------------------------------------------------------------
// masked PSNR  (metric.py:35-68)
// ------------------------------------------------------------------------------------------------
// pass 1: per block, sum over its pixels of (clamp(a)*m - clamp(b)*m)^2 over the 3 channels, and sum of m.
__global__ __launch_bounds__(256) void psnr_partial_kernel(const float* a, const float* b, const void* mask,
                                                           int mask_kind, float* partial, unsigned HW,
                                                           unsigned blocks_per_image) {
  __shared__ float sS[4], sM[4];
  const unsigned img = blockIdx.y;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  float se = 0.0f, sm = 0.0f;
  if (i < HW) {
    float m = 1.0f;
    if (mask_kind == CURL_MASK_U8) m = reinterpret_cast<const uint8_t*>(mask)[(size_t)img * HW + i] ? 1.0f : 0.0f;
    if (mask_kind == CURL_MASK_F32) m = reinterpret_cast<const float*>(mask)[(size_t)img * HW + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      size_t o = ((size_t)img * 3 + c) * HW + i;
      float d = clamp01(a[o]) * m - clamp01(b[o]) * m;  // metric.py:60-61 then :44
      se += d * d;
    }
    sm = m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    se += __shfl_xor(se, off, 64);
    sm += __shfl_xor(sm, off, 64);
  }
  if ((threadIdx.x & 63) == 0) sS[threadIdx.x >> 6] = se, sM[threadIdx.x >> 6] = sm;
  __syncthreads();
  if (threadIdx.x == 0) {
    size_t r = ((size_t)img * blocks_per_image + blockIdx.x) * 2;
    partial[r] = (sS[0] + sS[1]) + (sS[2] + sS[3]);
    partial[r + 1] = (sM[0] + sM[1]) + (sM[2] + sM[3]);
  }
}
// pass 2: one block per image, fixed-order float64 sums -> 10*log10(max^2 / mse), mse = SSE / (3 * sum(mask))
__global__ __launch_bounds__(256) void psnr_final_kernel(const float* partial, float* out, unsigned blocks_per_image,
                                                         float max_intensity) {
  __shared__ double s0[256], s1[256];
  const float* p = partial + (size_t)blockIdx.x * blocks_per_image * 2;
  double a = 0.0, b = 0.0;
  for (unsigned i = threadIdx.x; i < blocks_per_image; i += 256) {
    a += (double)p[2 * i];
    b += (double)p[2 * i + 1];
  }
  s0[threadIdx.x] = a;
  s1[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s0[threadIdx.x] += s0[threadIdx.x + off], s1[threadIdx.x] += s1[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double mse = s0[0] / (3.0 * s1[0]);  // metric.py:46-47 (0/0 -> NaN, as in the reference)
    out[blockIdx.x] = (float)(10.0 * log10((double)max_intensity * (double)max_intensity / mse));
  }
}

// ------------------------------------------------------------------------------------------------
// backward of the polynomial path: d loss / d coeffs  (autograd of model.py:499-520 w.r.t. R, L, H)
// ------------------------------------------------------------------------------------------------
#define TRI_BWD_N 2  // pixels per lane in pass 1: one packed Horner chain (4 per lane spilled: 9 KB scratch)
// pass 1: per pixel, the 9 colour variables (planes 0..8) and the 9 upstream gradients g_P[s][o] (planes 9..17)
template <int V>
__global__ __launch_bounds__(256, 2) void trispace_bwd_px_kernel(const float* img, const float* coeffs, const float* gout,
                                                                 float* pxbuf, unsigned HW, unsigned W, float fW, float fH,
                                                                 int residual_only, int vec_ok) {
  constexpr int NC = PolyEval<V>::kCoeffs, N = TRI_BWD_N;
  typedef float VT __attribute__((ext_vector_type(N)));
  __shared__ float s_coef[9 * NC];
  const unsigned b = blockIdx.y;
  const float* table = coeffs + (size_t)b * 9 * NC;
  for (int i = threadIdx.x; i < 9 * NC; i += 256) s_coef[i] = table[OpTriSpace<V>::stage_index(i)];
  __syncthreads();
  const unsigned i0 = (blockIdx.x * 256u + threadIdx.x) * N;
  if (i0 >= HW) return;
  const float* pi = img + (size_t)b * 3 * HW;
  const float* pg = gout + (size_t)b * 3 * HW;
  PxN<N> in, g;
  float xw[N], yh[N];
  if (vec_ok) {  // HW % N == 0 and planes aligned to the vector: i0 + N - 1 < HW
    auto unpack = [](float (&d)[N], const float* p) {
      VT t = *(const VT*)p;
#pragma unroll
      for (int k = 0; k < N; ++k) d[k] = t[k];
    };
    unpack(in.c0, pi + i0), unpack(in.c1, pi + HW + i0), unpack(in.c2, pi + 2 * (size_t)HW + i0);
    unpack(g.c0, pg + i0), unpack(g.c1, pg + HW + i0), unpack(g.c2, pg + 2 * (size_t)HW + i0);
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      unsigned i = min(i0 + k, HW - 1);
      in.c0[k] = pi[i], in.c1[k] = pi[HW + i], in.c2[k] = pi[2 * (size_t)HW + i];
      g.c0[k] = pg[i], g.c1[k] = pg[HW + i], g.c2[k] = pg[2 * (size_t)HW + i];
    }
  }
  unsigned row = i0 / W, col = i0 - row * W;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    xw[k] = (float)col / fW, yh[k] = (float)row / fH;
    if (++col == W) col = 0, ++row;
  }
  float vars[3][3][N], gP[3][3][N];
  trispace_bwd_n<V, N, true>(in, xw, yh, s_coef, g, residual_only != 0, vars, gP);
  float* q = pxbuf + (size_t)b * 18 * HW;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float* qv = q + (size_t)(s * 3 + c) * HW + i0;
      float* qg = q + (size_t)(9 + s * 3 + c) * HW + i0;
      if (vec_ok) {
        VT tv, tg;
#pragma unroll
        for (int k = 0; k < N; ++k) tv[k] = vars[s][c][k], tg[k] = gP[s][c][k];
        *(VT*)qv = tv, *(VT*)qg = tg;
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (i0 + k < HW) qv[k] = vars[s][c][k], qg[k] = gP[s][c][k];
      }
    }
}
// pass 1 for the spatial form, row-tiled like OpTriSpaceRows: the forward recompute runs on the row's collapsed
// 70-coefficient polynomials (69 FMAs per output instead of 125).  grid.x = blocks per row x rows.
__global__ __launch_bounds__(256, 2) void trispace_bwd_px_rows_kernel(const float* img, const float* coeffs, const float* gout,
                                                                      float* pxbuf, unsigned HW, unsigned W, unsigned H,
                                                                      unsigned units, unsigned segs, int residual_only,
                                                                      int vec_ok) {
  constexpr int NC4 = PolyEval<4>::kCoeffs, NC5 = PolyEval<5>::kCoeffs, N = TRI_BWD_N;
  typedef float VT __attribute__((ext_vector_type(N)));
  __shared__ float s_coef[9 * NC4];
  const unsigned b = blockIdx.y, row = blockIdx.x / segs, seg = blockIdx.x - row * segs;
  {
    const float* table = coeffs + (size_t)b * 9 * NC5;
    const float y = (float)row / (float)H;
    for (unsigned i = threadIdx.x; i < 9u * NC4; i += blockDim.x) {
      unsigned q = i / NC4, pos = i - q * NC4;
      s_coef[i] = collapse_coef(table + q * NC5, (int)pos, y);
    }
  }
  __syncthreads();
  const unsigned u = seg * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const unsigned col0 = u * N, i0 = row * W + col0;
  const float* pi = img + (size_t)b * 3 * HW;
  const float* pg = gout + (size_t)b * 3 * HW;
  PxN<N> in, g;
  float xw[N], yh[N];
  if (vec_ok) {  // W % N == 0 and planes aligned to the vector: col0 + N - 1 < W
    auto unpack = [](float (&d)[N], const float* p) {
      VT t = *(const VT*)p;
#pragma unroll
      for (int k = 0; k < N; ++k) d[k] = t[k];
    };
    unpack(in.c0, pi + i0), unpack(in.c1, pi + HW + i0), unpack(in.c2, pi + 2 * (size_t)HW + i0);
    unpack(g.c0, pg + i0), unpack(g.c1, pg + HW + i0), unpack(g.c2, pg + 2 * (size_t)HW + i0);
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      unsigned i = row * W + min(col0 + k, W - 1);
      in.c0[k] = pi[i], in.c1[k] = pi[HW + i], in.c2[k] = pi[2 * (size_t)HW + i];
      g.c0[k] = pg[i], g.c1[k] = pg[HW + i], g.c2[k] = pg[2 * (size_t)HW + i];
    }
  }
  const float fW = (float)W, rW = 1.0f / fW;
#pragma unroll
  for (int k = 0; k < N; ++k) xw[k] = div_small((float)(col0 + k), fW, rW), yh[k] = 0.0f;
  float vars[3][3][N], gP[3][3][N];
  trispace_bwd_n<4, N, true>(in, xw, yh, s_coef, g, residual_only != 0, vars, gP);
  float* q = pxbuf + (size_t)b * 18 * HW;
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float* qv = q + (size_t)(s * 3 + c) * HW + i0;
      float* qg = q + (size_t)(9 + s * 3 + c) * HW + i0;
      if (vec_ok) {
        VT tv, tg;
#pragma unroll
        for (int k = 0; k < N; ++k) tv[k] = vars[s][c][k], tg[k] = gP[s][c][k];
        *(VT*)qv = tv, *(VT*)qg = tg;
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k)
          if (col0 + k < W) qv[k] = vars[s][c][k], qg[k] = gP[s][c][k];
      }
    }
}
// pass 2: block (tile of 256*ppt pixels; space s and monomial chunk C; image b) accumulates g_P[s][o] * m_t in
// registers, reduces over the block, writes one row of partials.
struct CoefGradArgs {
  const float* pxbuf;
  float* partial;
  unsigned HW, W, tiles, ppt, items;  // items = B * 3 * tiles work items (image, space, tile)
  unsigned step_rows, step_cols;      // 256 / W, 256 % W: how (row, col) advances per loop step
  float fW, fH;
};
template <int V, int C>
__device__ __forceinline__ void coef_grad_block(const CoefGradArgs& a, unsigned b, unsigned s, unsigned tile,
                                                float (*sPart)[3 * PolyEval<V>::kChunk]) {
  constexpr int NC = PolyEval<V>::kCoeffs, T = PolyEval<V>::kChunk;
  const unsigned HW = a.HW;
  const float* base = a.pxbuf + (size_t)b * 18 * HW + (size_t)(s * 3) * HW;
  float acc[3][T];
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[o][j] = 0.0f;
  unsigned i = tile * 256u * a.ppt + threadIdx.x;
  unsigned row = i / a.W, col = i - row * a.W;
  const float rW = 1.0f / a.fW, rH = 1.0f / a.fH;
  auto fetch = [&](float (&d)[6], unsigned at) {  // clamped: always a valid pixel, masked below
    const float* p = base + min(at, HW - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = p[(size_t)c * HW], d[3 + c] = p[(size_t)(9 + c) * HW];
  };
  float cur[6], nxt[6];
  fetch(cur, i);
  for (unsigned k = 0; k < a.ppt; ++k) {
    fetch(nxt, i + 256u);  // next step's operands are in flight while this step computes
    float v[V], g[3];
    const bool live = i < HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = cur[c], g[c] = live ? cur[3 + c] : 0.0f;
    if (V == 5) {
      v[V - 2] = div_small((float)col, a.fW, rW);
      v[V - 1] = div_small((float)row, a.fH, rH);
    }
    coef_grad_accumulate<V, C>(acc, v, g);
    CURL_FENCE();  // keep the wait for the prefetch at the end of the step
#pragma unroll
    for (int c = 0; c < 6; ++c) cur[c] = nxt[c];
    i += 256u, row += a.step_rows, col += a.step_cols;
    if (col >= a.W) col -= a.W, ++row;
  }
  const int wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
  wave_sum_lane63(acc);
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int j = 0; j < T; ++j)
      if (lane_id == 63) sPart[wave][o * T + j] = acc[o][j];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * T; idx += 256) {
    int o = idx / T, j = idx - o * T;
    int t = C * T + j;
    if (t < NC)
      a.partial[((size_t)b * a.tiles + tile) * 9 * NC + (s * 3 + o) * NC + t] =
          (sPart[0][idx] + sPart[1][idx]) + (sPart[2][idx] + sPart[3][idx]);
  }
}
// 1-D grid.  Workgroups are dealt round-robin to the 8 XCDs (id % 8); the kChunks blocks of one work item re-read
// the same 6 planes of the tile, so they get ids 8 apart: same XCD, same L2, dispatched together.
template <int V>
__global__ __launch_bounds__(256) void trispace_coef_grad_kernel(CoefGradArgs a) {
  constexpr int CH = PolyEval<V>::kChunks;
  __shared__ float sPart[4][3 * PolyEval<V>::kChunk];
  const unsigned n = blockIdx.x, group = n / (8u * CH), r = n - group * (8u * CH);
  const unsigned chunk = r >> 3, item = group * 8u + (r & 7u);  // block-uniform
  if (item >= a.items) return;
  const unsigned tile = item % a.tiles, bs = item / a.tiles, s = bs % 3u, b = bs / 3u;
  if (chunk == 0) coef_grad_block<V, 0>(a, b, s, tile, sPart);
  if constexpr (CH > 1) {
    if (chunk == 1) coef_grad_block<V, 1>(a, b, s, tile, sPart);
    if (chunk == 2) coef_grad_block<V, 2>(a, b, s, tile, sPart);
  }
}
// pass 3: fixed-order float64 sum of the tile partials -> grad_coeffs [B,3,3,NC]
__global__ __launch_bounds__(256) void trispace_coef_final_kernel(const float* partial, float* gcoef, unsigned tiles, int n) {
  const unsigned b = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const float* p = partial + (size_t)b * tiles * n + k;
  double a = 0.0;
  for (unsigned t = 0; t < tiles; ++t) a += (double)p[(size_t)t * n];
  gcoef[(size_t)b * n + k] = (float)a;
}

// ------------------------------------------------------------------------------------------------
// CURLLoss pointwise terms (model.py:78-116)
// ------------------------------------------------------------------------------------------------
#define LOSS_NSUM 5  // sum|p-t|, sum cos, sum|lab|, sum|cone|, sum mask
__device__ __forceinline__ float load_mask(const void* mask, int mask_kind, size_t i) {
  if (mask_kind == CURL_MASK_U8) return reinterpret_cast<const uint8_t*>(mask)[i] ? 1.0f : 0.0f;
  if (mask_kind == CURL_MASK_F32) return reinterpret_cast<const float*>(mask)[i];
  return 1.0f;
}
__global__ __launch_bounds__(256) void loss_terms_kernel(const float* pred, const float* tgt, const void* mask,
                                                         int mask_kind, float* partial, float* Lp, float* Lt,
                                                         unsigned HW, unsigned blocks_per_image) {
  __shared__ float sP[4][LOSS_NSUM];
  const unsigned img = blockIdx.y;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  float acc[LOSS_NSUM] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (i < HW) {
    size_t o = (size_t)img * 3 * HW + i, mo = (size_t)img * HW + i;
    float m = load_mask(mask, mask_kind, mo);
    LossPx r = loss_terms(Px{pred[o], pred[o + HW], pred[o + 2 * (size_t)HW]},
                          Px{tgt[o], tgt[o + HW], tgt[o + 2 * (size_t)HW]}, m);
    acc[0] = r.rgb_l1, acc[1] = r.cos_sim, acc[2] = r.lab_l1, acc[3] = r.hsv_l1, acc[4] = m;
    if (Lp) Lp[mo] = r.Lp;
    if (Lt) Lt[mo] = r.Lt;
  }
#pragma unroll
  for (int c = 0; c < LOSS_NSUM; ++c) {
    float v = acc[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sP[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < LOSS_NSUM) {
    int c = threadIdx.x;
    partial[((size_t)img * blocks_per_image + blockIdx.x) * LOSS_NSUM + c] = (sP[0][c] + sP[1][c]) + (sP[2][c] + sP[3][c]);
  }
}
__global__ __launch_bounds__(256) void loss_terms_final_kernel(const float* partial, double* sums, unsigned blocks_per_image) {
  __shared__ double sA[256];
  const float* p = partial + (size_t)blockIdx.x * blocks_per_image * LOSS_NSUM;
  for (int c = 0; c < LOSS_NSUM; ++c) {
    double v = 0.0;
    for (unsigned i = threadIdx.x; i < blocks_per_image; i += 256) v += (double)p[(size_t)i * LOSS_NSUM + c];
    sA[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) sA[threadIdx.x] += sA[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) sums[(size_t)blockIdx.x * LOSS_NSUM + c] = sA[0];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void loss_terms_bwd_kernel(const float* pred, const float* tgt, const void* mask,
                                                             int mask_kind, const float* w4, const float* gLp,
                                                             float* gpred, unsigned HW) {
  const unsigned img = blockIdx.y;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= HW) return;
  size_t o = (size_t)img * 3 * HW + i, mo = (size_t)img * HW + i;
  const float w[4] = {w4[0], w4[1], w4[2], w4[3]};
  Px g = loss_terms_bwd(Px{pred[o], pred[o + HW], pred[o + 2 * (size_t)HW]},
                        Px{tgt[o], tgt[o + HW], tgt[o + 2 * (size_t)HW]}, load_mask(mask, mask_kind, mo), w,
                        gLp ? gLp[mo] : 0.0f);
  gpred[o] = g.c0;
  gpred[o + HW] = g.c1;
  gpred[o + 2 * (size_t)HW] = g.c2;
}

// ------------------------------------------------------------------------------------------------
// layout edges: u8 HWC <-> f32 CHW
// ------------------------------------------------------------------------------------------------
// One thread per pixel; HWC bytes of a wave are one contiguous 192/256-byte run, CHW floats three
// coalesced 256-byte runs.
__global__ __launch_bounds__(256) void u8hwc_to_f32chw_kernel(const uint8_t* in, float* out, size_t HW, int Cin,
                                                              size_t total) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // pixel index over B*HW
  if (i >= total) return;
  size_t b = i / HW, p = i - b * HW;
  const uint8_t* s = in + i * Cin;
  float* d = out + b * 3 * HW + p;
  d[0] = u8_to_unit((float)s[0]);  // to_tensor: byte -> float, div(255)
  d[HW] = u8_to_unit((float)s[1]);
  d[2 * HW] = u8_to_unit((float)s[2]);
}
__global__ __launch_bounds__(256) void f32chw_to_u8hwc_kernel(const float* in, uint8_t* out, size_t HW,
                                                              size_t total) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  size_t b = i / HW, p = i - b * HW;
  const float* s = in + b * 3 * HW + p;
  uint8_t* d = out + i * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = s[c * HW] * 255.0f;             // evaluate.py:64
    v = fminf(fmaxf(v, 0.0f), 255.0f);        // astype('uint8') of out-of-range is undefined: saturate
    d[c] = (uint8_t)(int)v;                   // truncation toward zero
  }
}

// infer.py:46-47: out*mask + (1-mask) (white background where the mask is 0), then to_pil_image's
// mul(255).byte() -- fused into the egress.  mask: u8 (MK 1) or f32 (MK 2), [B,1,H,W].
__global__ __launch_bounds__(256) void compose_white_u8hwc_kernel(const float* in, const void* mask, int mask_kind,
                                                                  uint8_t* out, size_t HW, size_t total) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  size_t b = i / HW, p = i - b * HW;
  const float* s = in + b * 3 * HW + p;
  float m = (mask_kind == CURL_MASK_U8) ? (reinterpret_cast<const uint8_t*>(mask)[i] ? 1.0f : 0.0f)
                                        : reinterpret_cast<const float*>(mask)[i];
  uint8_t* d = out + i * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (s[c * HW] * m + (1.0f - m)) * 255.0f;
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    d[c] = (uint8_t)(int)v;
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check_img(const void* in, const void* out, int B, int H, int W) {
  if (!in || !out) return fail(CURL_E_NULL, "image pointer is NULL");
  if (B <= 0 || H <= 0 || W <= 0) return fail(CURL_E_SHAPE, "B, H, W must be positive");
  if ((uint64_t)H * (uint64_t)W > (1ull << 30)) return fail(CURL_E_SHAPE, "H*W exceeds 2^30 pixels");
  if (B > 65535) return fail(CURL_E_SHAPE, "B exceeds 65535 images per call");
  return 0;
}
static int check_flags(unsigned flags, unsigned allowed) {
  const unsigned tune = CURL_F_TUNE_UNROLL_MASK | CURL_F_TUNE_NO_NT | CURL_F_DIAG_NO_MEM;
  if (flags & ~(allowed | tune)) return fail(CURL_E_FLAGS, "unsupported flag bit for this entry point");
  if ((flags & CURL_F_EXACT_ORDER) && (flags & CURL_F_PWL))
    return fail(CURL_E_FLAGS, "CURL_F_EXACT_ORDER and CURL_F_PWL are exclusive");
  return 0;
}

struct Geometry {
  int vec;       // 4 or 1
  int unroll;    // 1, 2 or 4
  unsigned n;    // plane length in vec units
  unsigned blocks_per_image, n_blocks, n_images;
  int nt;  // non-temporal loads and stores (float4 kernels)
  unsigned threads = 256;      // block size (row-tiled ops: follows the row width)
  unsigned units = 0, segs = 0;  // row-tiled ops: vec groups per row, blocks per row
};
// Row-tiled ops (Op::kRowTiles): a block covers (part of) ONE image row.  Block size = the multiple of 64 lanes
// (<= 256) that wastes the fewest lanes on the last block of a row, larger preferred: 1500 px = 375 float4 groups
// -> 2 blocks of 192 lanes (2.3 % idle) rather than 256 + 119.
static int make_row_geometry(Geometry& g, int B, int H, int W) {
  if (g.vec == 4 && W % 4 != 0) g.vec = 1;  // rows must start on a float4 boundary
  g.units = (unsigned)(W / g.vec);
  unsigned best_t = 64, best_waste = ~0u;
  for (unsigned t = 64; t <= 256; t += 64) {
    unsigned segs = (g.units + t - 1) / t, waste = segs * t - g.units;
    if (waste <= best_waste) best_waste = waste, best_t = t;
  }
  g.threads = best_t;
  g.segs = (g.units + best_t - 1) / best_t;
  g.n = (unsigned)((size_t)H * W / g.vec);
  uint64_t per_image = (uint64_t)g.segs * (uint64_t)H;
  if (per_image * (uint64_t)B > 0x7fffffffull || per_image > 0x7fffffffull) return fail(CURL_E_SHAPE, "grid too large");
  g.blocks_per_image = (unsigned)per_image;
  g.n_blocks = (unsigned)(per_image * (uint64_t)B);
  return 0;
}

// Library defaults chosen from the sweep in DESIGN.md (profiles/).

static int make_geometry(Geometry& g, const void* p0, const void* p1, const void* pm, int mask_kind, int B, int H, int W,
                         unsigned flags, int default_unroll) {
  size_t HW = (size_t)H * W;
  bool aligned = (HW % 4 == 0) && (((uintptr_t)p0 | (uintptr_t)p1) % 16 == 0);
  if (pm && mask_kind == CURL_MASK_F32 && ((uintptr_t)pm % 16)) aligned = false;
  if (pm && mask_kind == CURL_MASK_U8 && ((uintptr_t)pm % 4)) aligned = false;
  g.vec = aligned ? 4 : 1;
  int u = (int)((flags & CURL_F_TUNE_UNROLL_MASK) >> CURL_F_TUNE_UNROLL_SHIFT);
  if (u == 0) u = default_unroll;
  if (u != 1 && u != 2 && u != 4) return fail(CURL_E_FLAGS, "tuning unroll must be 1, 2 or 4");
  g.unroll = u;
  g.n = (unsigned)(HW / g.vec);
  unsigned per_chunk = 256u * (unsigned)u;
  g.blocks_per_image = (g.n + per_chunk - 1) / per_chunk;
  uint64_t nb = (uint64_t)g.blocks_per_image * (uint64_t)B;
  if (nb > 0x7fffffffull) return fail(CURL_E_SHAPE, "grid too large");
  g.n_blocks = (unsigned)nb;
  g.nt = (flags & CURL_F_TUNE_NO_NT) ? 0 : 1;
  g.n_images = (unsigned)B;
  return 0;
}

template <class Op, int VEC, int MK, bool NT>
static hipError_t launch_u(const Geometry& g, const StreamArgs& a, hipStream_t s) {
  dim3 grid(g.blocks_per_image, g.n_images), block(g.threads);
  if constexpr (Op::kSingleTileShape) {
    // very large ops (polynomial layers) are built for one tile shape only
    hipLaunchKernelGGL((stream_kernel<Op, VEC, 1, MK, NT>), grid, block, 0, s, a);
  } else {
    switch (g.unroll) {
      case 1:
        hipLaunchKernelGGL((stream_kernel<Op, VEC, 1, MK, NT>), grid, block, 0, s, a);
        break;
      case 2:
        hipLaunchKernelGGL((stream_kernel<Op, VEC, 2, MK, NT>), grid, block, 0, s, a);
        break;
      default:
        hipLaunchKernelGGL((stream_kernel<Op, VEC, 4, MK, NT>), grid, block, 0, s, a);
        break;
    }
  }
  return hipGetLastError();
}

template <class Op, int MK>
static hipError_t launch_v(const Geometry& g, const StreamArgs& a, hipStream_t s) {
  if (g.vec == 4) return g.nt ? launch_u<Op, 4, MK, true>(g, a, s) : launch_u<Op, 4, MK, false>(g, a, s);
  return launch_u<Op, 1, MK, false>(g, a, s);  // scalar correctness path: plain accesses
}

template <class Op>
static int launch_stream(const float* in, float* out, const void* mask, int mask_kind, const float* coef,
                         unsigned coef_stride, int B, int H, int W, unsigned flags, hipStream_t s, const char* name,
                         int op_flag = 0) {
  Geometry g;
  if (Op::kSingleTileShape) flags &= ~CURL_F_TUNE_UNROLL_MASK;
  if (int rc = make_geometry(g, in, out, mask, mask_kind, B, H, W, flags, Op::kUnroll)) return rc;
  if constexpr (Op::kRowTiles)
    if (int rc = make_row_geometry(g, B, H, W)) return rc;
  StreamArgs a;
  a.in = in;
  a.out = out;
  a.mask = mask;
  a.coef = coef;
  a.coef_stride = coef_stride;
  a.n = g.n;
  a.blocks_per_image = g.blocks_per_image;
  a.n_blocks = g.n_blocks;
  a.no_mem = (flags & CURL_F_DIAG_NO_MEM) ? 1 : 0;
  a.W = (unsigned)W;
  a.H = (unsigned)H;
  a.op_flag = op_flag;
  a.white = nullptr;
  a.units = g.units, a.segs = g.segs;
  hipError_t e;
  if constexpr (Op::kMask) {
    e = (mask_kind == CURL_MASK_U8)    ? launch_v<Op, CURL_MASK_U8>(g, a, s)
        : (mask_kind == CURL_MASK_F32) ? launch_v<Op, CURL_MASK_F32>(g, a, s)
                                       : launch_v<Op, CURL_MASK_NONE>(g, a, s);
  } else {
    e = launch_v<Op, CURL_MASK_NONE>(g, a, s);
  }
  if (e != hipSuccess) return hip_fail(e, name);
  return 0;
}

// FMT_U8HWC launch: interleaved bytes in and out, optional white-background mask; one tile shape per op
template <class Op, int MK>
static hipError_t launch_u8_v(const Geometry& g, const StreamArgs& a, hipStream_t s) {
  dim3 grid(g.blocks_per_image, g.n_images), block(g.threads);
  constexpr int U = Op::kSingleTileShape ? 1 : Op::kUnroll;
  if (g.vec == 4)
    hipLaunchKernelGGL((stream_kernel<Op, 4, U, MK, true, FMT_U8HWC>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((stream_kernel<Op, 1, U, MK, false, FMT_U8HWC>), grid, block, 0, s, a);
  return hipGetLastError();
}
template <class Op>
static int launch_stream_u8(const uint8_t* in, uint8_t* out, const void* mask, int mask_kind, const uint8_t* white,
                            const float* coef, unsigned coef_stride, int B, int H, int W, hipStream_t s,
                            const char* name, int op_flag = 0) {
  Geometry g;
  if (int rc = make_geometry(g, in, out, mask, mask_kind, B, H, W, 0, Op::kSingleTileShape ? 1 : Op::kUnroll)) return rc;
  // make_geometry asks 16-byte alignment of in/out for the vector path; dword accesses need 4 (also for `white`)
  size_t HW = (size_t)H * W;
  bool aligned = (HW % 4 == 0) && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)white) % 4 == 0);
  if (mask && mask_kind == CURL_MASK_F32 && ((uintptr_t)mask % 16)) aligned = false;
  if (mask && mask_kind == CURL_MASK_U8 && ((uintptr_t)mask % 4)) aligned = false;
  g.vec = aligned ? 4 : 1;
  g.n = (unsigned)(HW / g.vec);
  unsigned per_chunk = 256u * (unsigned)g.unroll;
  g.blocks_per_image = (g.n + per_chunk - 1) / per_chunk;
  if constexpr (Op::kRowTiles)
    if (int rc = make_row_geometry(g, B, H, W)) return rc;
  StreamArgs a{};
  a.in = reinterpret_cast<const float*>(in);
  a.out = reinterpret_cast<float*>(out);
  a.mask = mask;
  a.white = white;
  a.coef = coef;
  a.coef_stride = coef_stride;
  a.n = g.n;
  a.blocks_per_image = g.blocks_per_image;
  a.n_blocks = g.blocks_per_image * (unsigned)B;
  a.no_mem = 0;
  a.W = (unsigned)W;
  a.H = (unsigned)H;
  a.op_flag = op_flag;
  a.units = g.units, a.segs = g.segs;
  hipError_t e;
  if constexpr (Op::kMask) {
    e = (mask_kind == CURL_MASK_U8)    ? launch_u8_v<Op, CURL_MASK_U8>(g, a, s)
        : (mask_kind == CURL_MASK_F32) ? launch_u8_v<Op, CURL_MASK_F32>(g, a, s)
                                       : launch_u8_v<Op, CURL_MASK_NONE>(g, a, s);
  } else {
    e = launch_u8_v<Op, CURL_MASK_NONE>(g, a, s);
  }
  if (e != hipSuccess) return hip_fail(e, name);
  return 0;
}

static int check_mask(const void* mask, int mask_kind) {
  if (mask_kind != CURL_MASK_NONE && mask_kind != CURL_MASK_U8 && mask_kind != CURL_MASK_F32)
    return fail(CURL_E_MASK, "mask_kind must be 0 (none), 1 (u8) or 2 (f32)");
  if (mask_kind != CURL_MASK_NONE && !mask) return fail(CURL_E_MASK, "mask_kind set but mask pointer is NULL");
  return 0;
}
static int check_K(int K) {
  if (K < 2 || K > CURL_MAX_KNOTS) return fail(CURL_E_KNOTS, "knots per curve must be in [2, CURL_MAX_KNOTS]");
  return 0;
}
static int check_ws(const void* ws, size_t bytes, int B, int n_knots) {
  if (!ws) return fail(CURL_E_WORKSPACE, "workspace is NULL");
  if ((uintptr_t)ws % 16) return fail(CURL_E_WORKSPACE, "workspace must be 16-byte aligned");
  if (bytes < curl_workspace_bytes(B, n_knots)) return fail(CURL_E_WORKSPACE, "workspace too small");
  return 0;
}

static int run_prep(const float* r0, int n0, int K0, const float* r1, int n1, int K1, const float* r2, int n2, int K2,
                    float* ws, unsigned stride, float* reg, int B, hipStream_t s) {
  PrepArgs p;
  p.raw[0] = r0;
  p.raw[1] = r1;
  p.raw[2] = r2;
  p.ncurves[0] = n0;
  p.ncurves[1] = n1;
  p.ncurves[2] = n2;
  p.K[0] = K0;
  p.K[1] = K1;
  p.K[2] = K2;
  p.ws = ws;
  p.reg_out = reg;
  p.stride = stride;
  hipLaunchKernelGGL(knots_prep_kernel, dim3(B), dim3(256), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "knots_prep_kernel");
  return 0;
}

template <int VEC>
static hipError_t launch_chain_u(const Geometry& g, const ChainArgs& a, hipStream_t s) {
  dim3 grid(g.blocks_per_image, g.n_images), block(g.threads);
  switch (g.unroll) {
    case 1:
      hipLaunchKernelGGL((chain_kernel<VEC, 1>), grid, block, 0, s, a);
      break;
    case 2:
      hipLaunchKernelGGL((chain_kernel<VEC, 2>), grid, block, 0, s, a);
      break;
    default:
      hipLaunchKernelGGL((chain_kernel<VEC, 4>), grid, block, 0, s, a);
      break;
  }
  return hipGetLastError();
}

static int launch_chain(const float* in, float* out, const float* knots, unsigned knot_stride, int n_steps, int K,
                        const int* knot_off, const int* cin, const int* cout, int mode, int B, int H, int W,
                        unsigned flags, hipStream_t s) {
  Geometry g;
  if (int rc = make_geometry(g, in, out, nullptr, 0, B, H, W, flags, 2)) return rc;
  ChainArgs a;
  a.in = in;
  a.out = out;
  a.knots = knots;
  a.knot_stride = knot_stride;
  a.n_steps = n_steps;
  a.K = K;
  for (int i = 0; i < CHAIN_MAX; ++i) {
    a.knot_off[i] = i < n_steps ? knot_off[i] : 0;
    a.cin[i] = i < n_steps ? cin[i] : 0;
    a.cout[i] = i < n_steps ? cout[i] : 0;
  }
  a.n = g.n;
  a.blocks_per_image = g.blocks_per_image;
  a.n_blocks = g.n_blocks;
  a.mode = mode;
  hipError_t e = (g.vec == 4) ? launch_chain_u<4>(g, a, s) : launch_chain_u<1>(g, a, s);
  if (e != hipSuccess) return hip_fail(e, "chain_kernel");
  return 0;
}

static int chain_mode(unsigned flags) { return (flags & CURL_F_EXACT_ORDER) ? 1 : (flags & CURL_F_PWL) ? 2 : 0; }

// pixels per thread of the accumulation pass: enough to amortise the 3*T-value block reduction (>= 16), few enough
static unsigned tri_ppt(int B, size_t HW) {
  size_t p = HW * 9 * (size_t)B / (256u * 2048u);
  return (unsigned)(p < 16 ? 16 : (p > 64 ? 64 : p));
}
static unsigned tri_tiles(int B, size_t HW) {
  size_t per = 256u * (size_t)tri_ppt(B, HW);
  return (unsigned)((HW + per - 1) / per);
}

template <int V>
static hipError_t launch_trispace_bwd(const float* img, const float* coeffs, const float* gout, float* gcoef, float* pxbuf,
                                      float* partial, int B, int H, int W, int residual_only, hipStream_t s) {
  constexpr int NC = PolyEval<V>::kCoeffs;
  unsigned HW = (unsigned)((size_t)H * W), ppt = tri_ppt(B, HW), tiles = tri_tiles(B, HW);
  const unsigned va = 4 * TRI_BWD_N;
  int vec_ok = (HW % TRI_BWD_N == 0) && ((uintptr_t)img % va == 0) && ((uintptr_t)gout % va == 0) && ((uintptr_t)pxbuf % va == 0);
  unsigned threads = (HW + TRI_BWD_N - 1) / TRI_BWD_N;
  if constexpr (V == 5) {
    // row tiles (see make_row_geometry): block = the multiple of 64 lanes that wastes the fewest at the row end
    int row_vec_ok = vec_ok && (W % TRI_BWD_N == 0);
    unsigned units = ((unsigned)W + TRI_BWD_N - 1) / TRI_BWD_N, best_t = 64, best_waste = ~0u;
    for (unsigned t = 64; t <= 256; t += 64) {
      unsigned sg = (units + t - 1) / t, waste = sg * t - units;
      if (waste <= best_waste) best_waste = waste, best_t = t;
    }
    unsigned segs = (units + best_t - 1) / best_t;
    hipLaunchKernelGGL(trispace_bwd_px_rows_kernel, dim3(segs * (unsigned)H, (unsigned)B), dim3(best_t), 0, s, img, coeffs, gout,
                       pxbuf, HW, (unsigned)W, (unsigned)H, units, segs, residual_only, row_vec_ok);
  } else {
    hipLaunchKernelGGL(trispace_bwd_px_kernel<V>, dim3((threads + 255u) / 256u, (unsigned)B), dim3(256), 0, s, img, coeffs,
                       gout, pxbuf, HW, (unsigned)W, (float)W, (float)H, residual_only, vec_ok);
  }
  CoefGradArgs a{pxbuf, partial, HW, (unsigned)W, tiles, ppt, (unsigned)B * 3u * tiles, 256u / (unsigned)W, 256u % (unsigned)W,
                 (float)W, (float)H};
  hipLaunchKernelGGL(trispace_coef_grad_kernel<V>, dim3((a.items + 7u) / 8u * 8u * PolyEval<V>::kChunks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(trispace_coef_final_kernel, dim3((9 * NC + 255) / 256, (unsigned)B), dim3(256), 0, s, partial, gcoef,
                     tiles, 9 * NC);
  return hipGetLastError();
}

extern "C" {

int curl_version(void) { return 100; }  // 0.1.0

const char* curl_last_error(void) { return g_err; }

size_t curl_workspace_bytes(int B, int n_knots) {
  if (B <= 0 || n_knots < 0) return 0;
  return (size_t)B * ws_stride(n_knots) * sizeof(float);
}

int curl_apply_curve_f32(const float* img, const float* C, float* out, float* reg, int B, int H, int W, int K,
                         int channel_in, int channel_out, unsigned flags, curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!C) return fail(CURL_E_NULL, "C is NULL");
  if (int rc = check_K(K)) return rc;
  if (channel_in < 0 || channel_in > 2 || channel_out < 0 || channel_out > 2)
    return fail(CURL_E_SHAPE, "channel_in/channel_out must be 0, 1 or 2");
  if (int rc = check_flags(flags, CURL_F_EXACT_ORDER | CURL_F_PWL)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (reg) {
    hipLaunchKernelGGL(curve_reg_kernel, dim3((B + 63) / 64), dim3(64), 0, s, C, reg, B, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "curve_reg_kernel");
  }
  int off[1] = {0}, ci[1] = {channel_in}, co[1] = {channel_out};
  return launch_chain(img, out, C, (unsigned)K, 1, K, off, ci, co, chain_mode(flags), B, H, W, flags, s);
}

static int adjust_common(const float* img, const float* raw, float* out, float* reg, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, int K, unsigned flags, curl_stream_t stream,
                         int n_curves) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!raw) return fail(CURL_E_NULL, "raw knot pointer is NULL");
  if (int rc = check_K(K)) return rc;
  if (int rc = check_flags(flags, CURL_F_EXACT_ORDER | CURL_F_PWL)) return rc;
  if (int rc = check_ws(workspace, workspace_bytes, B, n_curves * K)) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  unsigned stride = ws_stride(n_curves * K);
  if (int rc = run_prep(raw, n_curves, K, nullptr, 0, 0, nullptr, 0, 0, ws, stride, reg, B, s)) return rc;
  int mode = chain_mode(flags);
  if (mode == 0) {
    if (n_curves == 3)
      return launch_stream<OpAdjust3>(img, out, nullptr, 0, ws, stride, B, H, W, flags, s, "adjust3");
    return launch_stream<OpAdjustHsv>(img, out, nullptr, 0, ws, stride, B, H, W, flags, s, "adjust_hsv");
  }
  int off[4], ci[4], co[4];
  for (int c = 0; c < n_curves; ++c) off[c] = WS_KNOTS + c * K;
  if (n_curves == 3) {
    for (int c = 0; c < 3; ++c) ci[c] = co[c] = c;  // curves.py:113-126,160-173
  } else {
    ci[0] = 0, co[0] = 0;  // curves.py:61-62
    ci[1] = 0, co[1] = 1;  // curves.py:67-68
    ci[2] = 1, co[2] = 1;  // curves.py:73-74
    ci[3] = 2, co[3] = 2;  // curves.py:79-80
  }
  return launch_chain(img, out, ws, stride, n_curves, K, off, ci, co, mode, B, H, W, flags, s);
}

int curl_adjust_rgb_f32(const float* img, const float* raw, float* out, float* reg, void* workspace,
                        size_t workspace_bytes, int B, int H, int W, int K, unsigned flags, curl_stream_t stream) {
  return adjust_common(img, raw, out, reg, workspace, workspace_bytes, B, H, W, K, flags, stream, 3);
}
int curl_adjust_lab_f32(const float* img, const float* raw, float* out, float* reg, void* workspace,
                        size_t workspace_bytes, int B, int H, int W, int K, unsigned flags, curl_stream_t stream) {
  return adjust_common(img, raw, out, reg, workspace, workspace_bytes, B, H, W, K, flags, stream, 3);
}
int curl_adjust_hsv_f32(const float* img, const float* raw, float* out, float* reg, void* workspace,
                        size_t workspace_bytes, int B, int H, int W, int K, unsigned flags, curl_stream_t stream) {
  return adjust_common(img, raw, out, reg, workspace, workspace_bytes, B, H, W, K, flags, stream, 4);
}

#define CONVERTER_ENTRY(FN, OP)                                                                              \
  int FN(const float* in, float* out, int B, int H, int W, unsigned flags, curl_stream_t stream) {           \
    g_err[0] = 0;                                                                                            \
    if (int rc = check_img(in, out, B, H, W)) return rc;                                                     \
    if (int rc = check_flags(flags, 0)) return rc;                                                           \
    return launch_stream<OP>(in, out, nullptr, 0, nullptr, 0, B, H, W, flags, (hipStream_t)stream, #FN);     \
  }
CONVERTER_ENTRY(curl_rgb2lab_f32, OpRgb2Lab)
CONVERTER_ENTRY(curl_lab2rgb_f32, OpLab2Rgb)
CONVERTER_ENTRY(curl_rgb2hsv_f32, OpRgb2Hsv)
CONVERTER_ENTRY(curl_hsv2rgb_f32, OpHsv2Rgb)

int curl_lab_stage_f32(const float* img, const void* mask, int mask_kind, const float* rawL, float* out, float* reg,
                       void* workspace, size_t workspace_bytes, int B, int H, int W, int Kl, unsigned flags,
                       curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!rawL) return fail(CURL_E_NULL, "rawL is NULL");
  if (int rc = check_mask(mask, mask_kind)) return rc;
  if (int rc = check_K(Kl)) return rc;
  if (int rc = check_flags(flags, 0)) return rc;
  if (int rc = check_ws(workspace, workspace_bytes, B, 3 * Kl)) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  unsigned stride = ws_stride(3 * Kl);
  if (int rc = run_prep(rawL, 3, Kl, nullptr, 0, 0, nullptr, 0, 0, ws, stride, reg, B, s)) return rc;
  return launch_stream<OpLabStage>(img, out, mask_kind ? mask : nullptr, mask_kind, ws, stride, B, H, W, flags, s,
                                   "lab_stage");
}

int curl_layer_fwd_f32(const float* img, const void* mask, int mask_kind, const float* rawL, const float* rawR,
                       const float* rawH, float* out, float* reg, void* workspace, size_t workspace_bytes, int B, int H,
                       int W, int Kl, int Kr, int Kh, unsigned flags, curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!rawL || !rawR || !rawH) return fail(CURL_E_NULL, "rawL/rawR/rawH must all be non-NULL");
  if (int rc = check_mask(mask, mask_kind)) return rc;
  if (int rc = check_K(Kl)) return rc;
  if (int rc = check_K(Kr)) return rc;
  if (int rc = check_K(Kh)) return rc;
  if (int rc = check_flags(flags, 0)) return rc;
  int n_knots = 3 * Kl + 3 * Kr + 4 * Kh;
  if (int rc = check_ws(workspace, workspace_bytes, B, n_knots)) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  unsigned stride = ws_stride(n_knots);
  if (int rc = run_prep(rawL, 3, Kl, rawR, 3, Kr, rawH, 4, Kh, ws, stride, reg, B, s)) return rc;
  return launch_stream<OpLayer>(img, out, mask_kind ? mask : nullptr, mask_kind, ws, stride, B, H, W, flags, s,
                                "curl_layer");
}

int curl_layer_fwd_u8hwc(const uint8_t* img, const void* mask, int mask_kind, const float* rawL, const float* rawR,
                         const float* rawH, const uint8_t* white_mask, uint8_t* out, float* reg, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, int Kl, int Kr, int Kh, unsigned flags,
                         curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!rawL || !rawR || !rawH) return fail(CURL_E_NULL, "rawL/rawR/rawH must all be non-NULL");
  if (int rc = check_mask(mask, mask_kind)) return rc;
  if (int rc = check_K(Kl)) return rc;
  if (int rc = check_K(Kr)) return rc;
  if (int rc = check_K(Kh)) return rc;
  if (flags) return fail(CURL_E_FLAGS, "unsupported flag bit for this entry point");
  int n_knots = 3 * Kl + 3 * Kr + 4 * Kh;
  if (int rc = check_ws(workspace, workspace_bytes, B, n_knots)) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  unsigned stride = ws_stride(n_knots);
  if (int rc = run_prep(rawL, 3, Kl, rawR, 3, Kr, rawH, 4, Kh, ws, stride, reg, B, s)) return rc;
  return launch_stream_u8<OpLayer>(img, out, mask_kind ? mask : nullptr, mask_kind, white_mask, ws, stride, B, H, W, s,
                                   "curl_layer_u8hwc");
}

size_t curl_layer_bwd_scratch_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  size_t HW = (size_t)H * W;
  size_t blocks = (HW + 255) / 256;  // upper bound: the scalar path, one pixel per lane
  return (size_t)B * blocks * BWD_NACC * sizeof(float);
}

int curl_layer_bwd_f32(const float* img, const void* mask, int mask_kind, const float* rawL, const float* rawR,
                       const float* rawH, const float* grad_out, const float* grad_reg, float* grad_img,
                       float* grad_rawL, float* grad_rawR, float* grad_rawH, void* workspace, size_t workspace_bytes,
                       void* scratch, size_t scratch_bytes, int B, int H, int W, int Kl, int Kr, int Kh, unsigned flags,
                       curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, grad_out, B, H, W)) return rc;
  if (!rawL || !rawR || !rawH) return fail(CURL_E_NULL, "rawL/rawR/rawH must all be non-NULL");
  if (!grad_rawL || !grad_rawR || !grad_rawH) return fail(CURL_E_NULL, "grad_rawL/R/H must all be non-NULL");
  if (int rc = check_mask(mask, mask_kind)) return rc;
  if (int rc = check_K(Kl)) return rc;
  if (int rc = check_K(Kr)) return rc;
  if (int rc = check_K(Kh)) return rc;
  if (int rc = check_flags(flags, 0)) return rc;
  int n_knots = 3 * Kl + 3 * Kr + 4 * Kh;
  if (int rc = check_ws(workspace, workspace_bytes, B, n_knots)) return rc;
  if (!scratch || (uintptr_t)scratch % 16 || scratch_bytes < curl_layer_bwd_scratch_bytes(B, H, W))
    return fail(CURL_E_WORKSPACE, "scratch missing, misaligned or smaller than curl_layer_bwd_scratch_bytes");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  unsigned stride = ws_stride(n_knots);
  if (int rc = run_prep(rawL, 3, Kl, rawR, 3, Kr, rawH, 4, Kh, ws, stride, nullptr, B, s)) return rc;
  size_t HW = (size_t)H * W;
  bool aligned = (HW % 4 == 0) && (((uintptr_t)img | (uintptr_t)grad_out | (uintptr_t)grad_img) % 16 == 0);
  if (mask_kind == CURL_MASK_F32 && ((uintptr_t)mask % 16)) aligned = false;
  if (mask_kind == CURL_MASK_U8 && ((uintptr_t)mask % 4)) aligned = false;
  BwdArgs a;
  a.in = img;
  a.gout = grad_out;
  a.gin = grad_img;
  a.mask = mask_kind ? mask : nullptr;
  a.coef = ws;
  a.partial = (float*)scratch;
  a.coef_stride = stride;
  a.n = (unsigned)(HW / (aligned ? 4 : 1));
  a.blocks_per_image = (a.n + 255u) / 256u;
  uint64_t nb = (uint64_t)a.blocks_per_image * (uint64_t)B;
  if (nb > 0x7fffffffull) return fail(CURL_E_SHAPE, "grid too large");
  a.n_blocks = (unsigned)nb;
  dim3 grid(a.blocks_per_image, (unsigned)B), block(256);
#define LAUNCH_BWD(V, M) hipLaunchKernelGGL((layer_bwd_kernel<V, M>), grid, block, 0, s, a)
  if (aligned) {
    if (mask_kind == CURL_MASK_U8) LAUNCH_BWD(4, CURL_MASK_U8);
    else if (mask_kind == CURL_MASK_F32) LAUNCH_BWD(4, CURL_MASK_F32);
    else LAUNCH_BWD(4, CURL_MASK_NONE);
  } else {
    if (mask_kind == CURL_MASK_U8) LAUNCH_BWD(1, CURL_MASK_U8);
    else if (mask_kind == CURL_MASK_F32) LAUNCH_BWD(1, CURL_MASK_F32);
    else LAUNCH_BWD(1, CURL_MASK_NONE);
  }
#undef LAUNCH_BWD
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "layer_bwd_kernel");
  KnotsBwdArgs kb;
  kb.ws = ws;
  kb.partial = (const float*)scratch;
  kb.greg = grad_reg;
  kb.graw[0] = grad_rawL;
  kb.graw[1] = grad_rawR;
  kb.graw[2] = grad_rawH;
  kb.K[0] = Kl;
  kb.K[1] = Kr;
  kb.K[2] = Kh;
  kb.ws_stride = stride;
  kb.blocks_per_image = a.blocks_per_image;
  hipLaunchKernelGGL(knots_bwd_kernel, dim3(B), dim3(256), 0, s, kb);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "knots_bwd_kernel");
  return 0;
}

int curl_trispace_fwd_f32(const float* img, const float* coeffs, float* out, int B, int H, int W, int num_coeffs,
                          unsigned flags, curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!coeffs) return fail(CURL_E_NULL, "coeffs is NULL");
  if (num_coeffs != 126 && num_coeffs != 35)
    return fail(CURL_E_KNOTS, "num_coeffs must be 126 (degree 4, 5 variables) or 35 (degree 4, 3 variables)");
  if (int rc = check_flags(flags, CURL_F_RESIDUAL_ONLY)) return rc;
  int ro = (flags & CURL_F_RESIDUAL_ONLY) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (num_coeffs == 126)
    return launch_stream<OpTriSpaceRows>(img, out, nullptr, 0, coeffs, 9 * 126, B, H, W, flags, s, "trispace_rows", ro);
  return launch_stream<OpTriSpace<3>>(img, out, nullptr, 0, coeffs, 9 * 35, B, H, W, flags, s, "trispace", ro);
}

int curl_trispace_fwd_u8hwc(const uint8_t* img, const float* coeffs, const uint8_t* white_mask, uint8_t* out, int B, int H,
                            int W, int num_coeffs, unsigned flags, curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!coeffs) return fail(CURL_E_NULL, "coeffs is NULL");
  if (num_coeffs != 126 && num_coeffs != 35)
    return fail(CURL_E_KNOTS, "num_coeffs must be 126 (degree 4, 5 variables) or 35 (degree 4, 3 variables)");
  if (flags) return fail(CURL_E_FLAGS, "unsupported flag bit for this entry point (the byte output is an image)");
  hipStream_t s = (hipStream_t)stream;
  if (num_coeffs == 126)
    return launch_stream_u8<OpTriSpaceRows>(img, out, nullptr, 0, white_mask, coeffs, 9 * 126, B, H, W, s,
                                            "trispace_rows_u8hwc");
  return launch_stream_u8<OpTriSpace<3>>(img, out, nullptr, 0, white_mask, coeffs, 9 * 35, B, H, W, s, "trispace_u8hwc");
}

int curl_poly_layer_f32(const float* img, const float* coeffs, float* out, int B, int H, int W, int num_variables,
                        curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, out, B, H, W)) return rc;
  if (!coeffs) return fail(CURL_E_NULL, "coeffs is NULL");
  if (num_variables != 5 && num_variables != 3) return fail(CURL_E_SHAPE, "num_variables must be 5 or 3 (degree 4)");
  unsigned HW = (unsigned)((size_t)H * W);
  dim3 grid((HW + 255u) / 256u, (unsigned)B), block(256);
  if (num_variables == 5)
    hipLaunchKernelGGL(poly_layer_kernel<5>, grid, block, 0, (hipStream_t)stream, img, coeffs, out, HW);
  else
    hipLaunchKernelGGL(poly_layer_kernel<3>, grid, block, 0, (hipStream_t)stream, img, coeffs, out, HW);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "poly_layer_kernel");
  return 0;
}

int curl_u8hwc_to_f32chw(const uint8_t* in, float* out, int B, int H, int W, int Cin, curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(in, out, B, H, W)) return rc;
  if (Cin != 3 && Cin != 4) return fail(CURL_E_SHAPE, "Cin must be 3 (RGB) or 4 (RGBA)");
  size_t HW = (size_t)H * W, total = HW * (size_t)B;
  if ((total + 255) / 256 > 0x7fffffffull) return fail(CURL_E_SHAPE, "grid too large");
  hipLaunchKernelGGL(u8hwc_to_f32chw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, HW, Cin, total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "u8hwc_to_f32chw");
  return 0;
}

int curl_f32chw_to_u8hwc(const float* in, uint8_t* out, int B, int H, int W, curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(in, out, B, H, W)) return rc;
  size_t HW = (size_t)H * W, total = HW * (size_t)B;
  if ((total + 255) / 256 > 0x7fffffffull) return fail(CURL_E_SHAPE, "grid too large");
  hipLaunchKernelGGL(f32chw_to_u8hwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     in, out, HW, total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "f32chw_to_u8hwc");
  return 0;
}

int curl_compose_white_u8hwc(const float* in, const void* mask, int mask_kind, uint8_t* out, int B, int H, int W,
                             curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(in, out, B, H, W)) return rc;
  if (mask_kind != CURL_MASK_U8 && mask_kind != CURL_MASK_F32) return fail(CURL_E_MASK, "mask_kind must be 1 (u8) or 2 (f32)");
  if (!mask) return fail(CURL_E_MASK, "mask is NULL");
  size_t HW = (size_t)H * W, total = HW * (size_t)B;
  if ((total + 255) / 256 > 0x7fffffffull) return fail(CURL_E_SHAPE, "grid too large");
  hipLaunchKernelGGL(compose_white_u8hwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, in, mask, mask_kind, out, HW, total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "compose_white_u8hwc");
  return 0;
}

size_t curl_psnr_scratch_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  size_t blocks = ((size_t)H * W + 255) / 256;
  return (size_t)B * blocks * 2 * sizeof(float);
}

int curl_psnr_f32(const float* a, const float* b, const void* mask, int mask_kind, float* psnr, void* scratch,
                  size_t scratch_bytes, int B, int H, int W, float max_intensity, curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(a, b, B, H, W)) return rc;
  if (!psnr) return fail(CURL_E_NULL, "psnr output is NULL");
  if (int rc = check_mask(mask, mask_kind)) return rc;
  if (!scratch || scratch_bytes < curl_psnr_scratch_bytes(B, H, W))
    return fail(CURL_E_WORKSPACE, "scratch missing or smaller than curl_psnr_scratch_bytes");
  unsigned HW = (unsigned)((size_t)H * W), bpi = (HW + 255u) / 256u;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(psnr_partial_kernel, dim3(bpi, (unsigned)B), dim3(256), 0, s, a, b, mask_kind ? mask : nullptr,
                     mask_kind, (float*)scratch, HW, bpi);
  hipLaunchKernelGGL(psnr_final_kernel, dim3((unsigned)B), dim3(256), 0, s, (const float*)scratch, psnr, bpi,
                     max_intensity);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "psnr kernels");
  return 0;
}

size_t curl_loss_terms_scratch_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * (((size_t)H * W + 255) / 256) * LOSS_NSUM * sizeof(float);
}

int curl_loss_terms_f32(const float* pred, const float* target, const void* mask, int mask_kind, double* sums,
                        float* L_pred, float* L_target, void* scratch, size_t scratch_bytes, int B, int H, int W,
                        curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(pred, target, B, H, W)) return rc;
  if (!sums) return fail(CURL_E_NULL, "sums is NULL");
  if (int rc = check_mask(mask, mask_kind)) return rc;
  if (!scratch || scratch_bytes < curl_loss_terms_scratch_bytes(B, H, W))
    return fail(CURL_E_WORKSPACE, "scratch missing or smaller than curl_loss_terms_scratch_bytes");
  unsigned HW = (unsigned)((size_t)H * W), bpi = (HW + 255u) / 256u;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_terms_kernel, dim3(bpi, (unsigned)B), dim3(256), 0, s, pred, target,
                     mask_kind ? mask : nullptr, mask_kind, (float*)scratch, L_pred, L_target, HW, bpi);
  hipLaunchKernelGGL(loss_terms_final_kernel, dim3((unsigned)B), dim3(256), 0, s, (const float*)scratch, sums, bpi);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "loss_terms kernels");
  return 0;
}

int curl_loss_terms_bwd_f32(const float* pred, const float* target, const void* mask, int mask_kind,
                            const float* weights, const float* grad_L_pred, float* grad_pred, int B, int H, int W,
                            curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(pred, target, B, H, W)) return rc;
  if (!weights || !grad_pred) return fail(CURL_E_NULL, "weights / grad_pred is NULL");
  if (int rc = check_mask(mask, mask_kind)) return rc;
  unsigned HW = (unsigned)((size_t)H * W), bpi = (HW + 255u) / 256u;
  hipLaunchKernelGGL(loss_terms_bwd_kernel, dim3(bpi, (unsigned)B), dim3(256), 0, (hipStream_t)stream, pred, target,
                     mask_kind ? mask : nullptr, mask_kind, weights, grad_L_pred, grad_pred, HW);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "loss_terms_bwd_kernel");
  return 0;
}

size_t curl_trispace_bwd_scratch_bytes(int B, int H, int W, int num_coeffs) {
  if (B <= 0 || H <= 0 || W <= 0 || (num_coeffs != 126 && num_coeffs != 35)) return 0;
  size_t HW = (size_t)H * W;
  return ((size_t)B * 18 * HW + (size_t)B * tri_tiles(B, HW) * 9 * num_coeffs) * sizeof(float);
}

int curl_trispace_bwd_f32(const float* img, const float* coeffs, const float* grad_out, float* grad_coeffs, void* scratch,
                          size_t scratch_bytes, int B, int H, int W, int num_coeffs, unsigned flags,
                          curl_stream_t stream) {
  g_err[0] = 0;
  if (int rc = check_img(img, grad_out, B, H, W)) return rc;
  if (!coeffs || !grad_coeffs) return fail(CURL_E_NULL, "coeffs / grad_coeffs is NULL");
  if (num_coeffs != 126 && num_coeffs != 35) return fail(CURL_E_KNOTS, "num_coeffs must be 126 or 35");
  if (int rc = check_flags(flags, CURL_F_RESIDUAL_ONLY)) return rc;
  if (!scratch || (uintptr_t)scratch % 16 || scratch_bytes < curl_trispace_bwd_scratch_bytes(B, H, W, num_coeffs))
    return fail(CURL_E_WORKSPACE, "scratch missing, misaligned or smaller than curl_trispace_bwd_scratch_bytes");
  size_t HW = (size_t)H * W;
  float* pxbuf = (float*)scratch;
  float* partial = pxbuf + (size_t)B * 18 * HW;
  int ro = (flags & CURL_F_RESIDUAL_ONLY) ? 1 : 0;
  hipError_t e = (num_coeffs == 126)
                     ? launch_trispace_bwd<5>(img, coeffs, grad_out, grad_coeffs, pxbuf, partial, B, H, W, ro, (hipStream_t)stream)
                     : launch_trispace_bwd<3>(img, coeffs, grad_out, grad_coeffs, pxbuf, partial, B, H, W, ro, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "trispace backward kernels");
  return 0;
}

}  // extern "C"
