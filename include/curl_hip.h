/* curl_hip.h -- C ABI of libcurlhip.so: the MI355X (gfx950) implementation of the
 * CURL per-pixel colour-curve hot path.
 *
 * The reference (danielbulhosa/CURL) has no FFI: its boundary for this path is a set
 * of Python call signatures over eager torch ops.  Each entry point below REPLACES one
 * of those call sites; the citation after "replaces:" is the reference interface
 * (file:line under the reference tree).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add at each site.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator
 *     in practice); the library never allocates, frees or retains device memory;
 *   - images are float32, NCHW planar, contiguous: [B,3,H,W]; masks are [B,1,H,W];
 *   - out may alias img (every pixel is read before it is written);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; 0 = default stream)
 *     and the call returns without synchronising;
 *   - re-entrant, no global mutable state; the error string is thread-local;
 *   - return value: 0 = ok; <0 = argument error (CURL_E_*); >0 = a hipError_t.
 *     No exception crosses the boundary and nothing calls exit().
 *
 * Semantics where the reference is broken as written: the
 * adjust_* wrappers seed the regulariser with zeros(B); the layer skips the dead
 * `feat` lines of model.py:152,158,164.
 */
#ifndef CURL_HIP_H
#define CURL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* curl_stream_t; /* hipStream_t */

enum {
  CURL_OK = 0,
  CURL_E_NULL = -1,      /* a required pointer is NULL */
  CURL_E_SHAPE = -2,     /* B/H/W/K/channel out of range */
  CURL_E_KNOTS = -3,     /* knot count unsupported (K < 2 or K > CURL_MAX_KNOTS) */
  CURL_E_WORKSPACE = -4, /* workspace missing, misaligned or too small */
  CURL_E_MASK = -5,      /* mask_kind invalid or mask pointer inconsistent with it */
  CURL_E_FLAGS = -6      /* unknown flag bit */
};

#define CURL_MAX_KNOTS 256
/* Every `K` argument is the number of knots per curve.  torch.chunk(P, n, dim=1) of a parameter count that n does not
 * divide hands the first n-1 curves K = ceil(N/n) knots and the LAST one fewer (curves.py:53,105,152): pass
 * CURL_K_UNEVEN(K, K_last) for that segment (raw pointer: [B, (n-1)*K + K_last]).  Supported by the default (affine)
 * forward forms and the backward; CURL_F_EXACT_ORDER / CURL_F_PWL and curl_apply_curve_f32 take a plain K. */
#define CURL_K_UNEVEN(K, K_last) ((int)((unsigned)(K) | ((unsigned)(K_last) << 16)))

/* mask_kind */
#define CURL_MASK_NONE 0 /* mask pointer ignored (treated as all ones) */
#define CURL_MASK_U8 1   /* uint8/bool, non-zero = 1.0 (data.py:190 hands bool masks) */
#define CURL_MASK_F32 2  /* float32, multiplied as is (infer.py:41) */

/* flags */
#define CURL_F_EXACT_ORDER 0x1u /* evaluate each curve as the in-order fp32 sum of curves.py:31-32
                                   (no FMA contraction, torch's cascade order) instead of the collapsed
                                   affine form a + b*x.  apply_curve / adjust_*: bit-identical to the reference.
                                   curl_lab_stage_f32 / curl_layer_fwd(_slab)_f32: the VALIDATION mode of the fused
                                   stages -- the same in-order sums (knots and slopes of the image in LDS), every
                                   `* mask` of model.py:154,160,166 executed, the converters' 1e-9 floors and refined
                                   reciprocals kept: ~3x the default's time, for checking it, not for production. */
#define CURL_F_PWL 0x2u         /* paper-style piecewise-linear curve (clamp (S*x-j) to [0,1]) -- NOT the
                                   reference's arithmetic; explicit non-parity option.  adjust_* and layer. */
#define CURL_F_RESIDUAL_ONLY 0x4u /* curl_trispace_fwd_f32: write the residual instead of clamp(img + residual)
                                    (TriSpaceRegNet with is_train=False, model.py:485) */
/* tuning bits (0 = library default; used by the bench sweep, never change results) */
#define CURL_F_TUNE_UNROLL_SHIFT 8 /* bits 8..10: float4 groups per thread, 0 = default */
#define CURL_F_TUNE_UNROLL_MASK 0x700u
#define CURL_F_TUNE_BLOCK_SHIFT 11 /* bits 11..12: threads per workgroup of the streaming kernels, 0 = 256, 1 = 128, 2 = 64 */
#define CURL_F_TUNE_BLOCK_MASK 0x1800u
#define CURL_F_TUNE_XCD_SHIFT 13   /* bits 13..14: workgroup -> tile mapping of the streaming kernels: 0 = library default,
                                      1 = plain (workgroup b takes tile b), 2 = XCD-contiguous (the 8 XCDs, which are dealt
                                      workgroups round-robin, each own one contiguous eighth of an image's tiles) */
#define CURL_F_TUNE_XCD_MASK 0x6000u
#define CURL_F_TUNE_OCC_SHIFT 19   /* bits 19..21: resident 256-thread workgroups per CU of the f32 streaming kernels (= waves per
                                     SIMD), held down by reserving unused LDS (floor(128 / k) granules of 1 280 bytes): 0 = library default per operator
                                     (large launches of the light operators run at 2-6), 1 = no cap,
                                     2..7 = k */
#define CURL_F_TUNE_OCC_MASK 0x380000u
#define CURL_F_TUNE_NO_NT 0x8000u     /* plain loads/stores instead of the default non-temporal ones */
#define CURL_F_MASK_FIRST 0x400000u /* bool / uint8 FOREGROUND masks (curl_layer_fwd(_slab)_f32, curl_lab_stage_f32,
                                      curl_hsv_stage_f32, curl_layer_bwd_f32): a wavefront first waits for its mask bytes
                                      and, where all of them are zero, never reads its pixels.  Forward: +0.5 ... +0.8 % on an
                                      all-ones mask, -7 % at 70 % coverage, -25 % at 40 %.  Same results
                                      bit for bit.  Ignored where it does not apply (other mask kinds, scalar kernels). */
#define CURL_F_TUNE_PREP_SHIFT 23
#define CURL_F_TUNE_PREP_MASK 0x1800000u /* where the image's curves are collapsed (exp, slopes, the (a, b) pairs): 0 = the library's
                                            choice -- INSIDE the streaming kernel for launches small enough to be resident in
                                            one or two rounds (one launch instead of two: one frame, the training crop batch),
                                            in a launch of their own (one workgroup per image) otherwise; 1 = always the
                                            separate launch; 2 = always in the kernel (where the entry point has that form).
                                            Forward: curl_layer_fwd(_slab)_f32, curl_lab_stage_f32, curl_hsv_stage_f32,
                                            curl_adjust_*_f32 (affine form).  Results are bit-identical either way. */
#define CURL_F_DIAG_NO_MEM 0x10000u   /* DIAGNOSTICS ONLY: inputs synthesised in registers, stores suppressed --
                                         times the arithmetic alone; the output buffer is left untouched */

#define CURL_F_WS_READY 0x40000u /* curl_layer_bwd_f32: `workspace` is the buffer curl_layer_fwd_f32 (or an earlier backward) was
                                    handed for the SAME raw knots and has not been written since: it already holds the
                                    exp'd knots and collapsed curves, the knot-prep launch is skipped.  Every prepared row
                                    carries a stamp of the knot count and row stride it was filled for: a row nobody filled
                                    for this call's shape (a zeroed buffer, another K) is answered with NaN knot gradients
                                    AND a NaN gradient image for that image (never a plausible-looking one);
                                    that the VALUES belong to these knots remains the caller's word */
#define CURL_F_DIAG_SKIP_PREP 0x20000u /* DIAGNOSTICS ONLY (curl_layer_fwd_f32): the knot-prep launch is skipped and the
                                         workspace is taken to hold an earlier call's result for the same knots; `reg`
                                         is not written.  Measures what the prep launch + its kernel boundary cost. */

int curl_version(void);
/* Thread-local description of the last non-zero return on this thread ("" if none). */
const char* curl_last_error(void);

/* Bytes of device scratch the knot-driven entry points need (holds, per image, the 10 collapsed
 * (a,b) pairs, per-space regularisers and the exp'd knots).  n_knots = total raw knots per image
 * handed to that call (e.g. 160 for the layer, 48 for adjust_rgb). */
size_t curl_workspace_bytes(int B, int n_knots);

/* replaces: curves.apply_curve(img, C, slope_sqr_diff, channel_in, channel_out)  curves.py:4-38
 * C [B,K] are the knots AFTER exp.  reg [B] may be NULL; otherwise reg[b] += sum of squared slope
 * differences (in place, like curves.py:24).  The whole image is clamped to [0,1] (curves.py:36).
 * Default = CURL_F_EXACT_ORDER semantics are available; flags=0 uses the affine collapse. */
int curl_apply_curve_f32(const float* img, const float* C, float* out, float* reg,
                         int B, int H, int W, int K, int channel_in, int channel_out,
                         unsigned flags, curl_stream_t stream);

/* replaces: curves.adjust_rgb(img, R) / adjust_lab(img, L) / adjust_hsv(img, S)
 *           curves.py:90-133 / 136-180 / 41-87
 * raw [B, 3*K] (rgb, lab) or [B, 4*K] (hsv): raw parameters, exp() is applied inside (curves.py:54,106,153).
 * reg [B] (nullable) is ASSIGNED the regulariser (seeded with zero). */
int curl_adjust_rgb_f32(const float* img, const float* raw, float* out, float* reg,
                        void* workspace, size_t workspace_bytes,
                        int B, int H, int W, int K, unsigned flags, curl_stream_t stream);
int curl_adjust_lab_f32(const float* img, const float* raw, float* out, float* reg,
                        void* workspace, size_t workspace_bytes,
                        int B, int H, int W, int K, unsigned flags, curl_stream_t stream);
int curl_adjust_hsv_f32(const float* img, const float* raw, float* out, float* reg,
                        void* workspace, size_t workspace_bytes,
                        int B, int H, int W, int K, unsigned flags, curl_stream_t stream);

/* replaces: colors.RGB2LAB.forward colors.py:27-62 ; LAB2RGB.forward colors.py:88-123 ;
 *           RGB2HSV.forward colors.py:195-242 ; HSV2RGB.forward colors.py:131-177 */
int curl_rgb2lab_f32(const float* in, float* out, int B, int H, int W, unsigned flags, curl_stream_t stream);
int curl_lab2rgb_f32(const float* in, float* out, int B, int H, int W, unsigned flags, curl_stream_t stream);
int curl_rgb2hsv_f32(const float* in, float* out, int B, int H, int W, unsigned flags, curl_stream_t stream);
int curl_hsv2rgb_f32(const float* in, float* out, int B, int H, int W, unsigned flags, curl_stream_t stream);

/* replaces: the first stage of CURLLayer.forward closed back to RGB, model.py:151-157
 *           (rgb2lab -> adjust_lab -> *mask -> lab2rgb) as ONE pass over the pixels.
 * rawL [B, 3*Kl]. reg [B] (nullable) is assigned reg_lab.
 * flags: CURL_F_PWL evaluates the three curves as the PAPER's clamped piecewise-linear interpolation of the knots
 * (knots and slopes of the image staged in LDS, interval = direct index, two per-lane gathers + one fma per curve)
 * instead of the reference's affine form -- a non-parity option, as for curl_adjust_*. */
int curl_lab_stage_f32(const float* img, const void* mask, int mask_kind, const float* rawL,
                       float* out, float* reg, void* workspace, size_t workspace_bytes,
                       int B, int H, int W, int Kl, unsigned flags, curl_stream_t stream);

/* replaces: the third stage of CURLLayer.forward from RGB to its RGB residual, model.py:163-169
 *           (rgb2hsv colors.py:195-242 -> adjust_hsv curves.py:41-87 -> *mask -> hsv2rgb colors.py:131-177) as ONE
 *           pass over the pixels -- with curl_adjust_rgb_f32 and curl_lab_stage_f32 the "one fused kernel per colour
 *           space" of the path.  rawH [B, 4*Kh].  reg [B] (nullable) is assigned reg_hsv.  Where a bool / uint8 mask is
 *           0 the result is hsv2rgb(0,0,0) = 0.  flags: tuning bits only. */
int curl_hsv_stage_f32(const float* img, const void* mask, int mask_kind, const float* rawH,
                       float* out, float* reg, void* workspace, size_t workspace_bytes,
                       int B, int H, int W, int Kh, unsigned flags, curl_stream_t stream);

/* replaces: CURLLayer.forward(img, mask, L, R, H)  model.py:137-176, as ONE pass over the pixels.
 * rawL [B,3*Kl], rawR [B,3*Kr], rawH [B,4*Kh] (the slices L[:, :48], R[:, :48], H[:, :64] of
 * model.py:153,159,165, made contiguous by the caller).  reg [B] (nullable) is assigned
 * reg_rgb + reg_lab + reg_hsv (model.py:172-174).
 * flags: CURL_F_PWL = all ten curves as the paper's clamped piecewise-linear interpolation, knots and slopes of the
 * image staged in LDS (non-parity option, see curl_lab_stage_f32). */
int curl_layer_fwd_f32(const float* img, const void* mask, int mask_kind,
                       const float* rawL, const float* rawR, const float* rawH,
                       float* out, float* reg, void* workspace, size_t workspace_bytes,
                       int B, int H, int W, int Kl, int Kr, int Kh,
                       unsigned flags, curl_stream_t stream);

/* The same pass over ROWS [row0, row0 + rows) of every image only -- the shared-encoder / split-pixels layout of
 * BASELINE.json's north_star: every GPU holds the knots of the whole batch (one RCCL broadcast /
 * all-gather, 640 B per image) and enhances its row slab of every image.
 * replaces: CURLLayer.forward(img[:, :, row0:row0+rows], mask[:, :, row0:row0+rows], L, R, H) (model.py:137-176)
 *           WITHOUT the .contiguous() copy a slice of an NCHW tensor needs: img, mask and out are the base pointers
 *           of the FULL [B,3,H,W] / [B,1,H,W] tensors; only the slab's rows are read and written (in place is
 *           allowed), plane stride H*W.  float4 kernels when H*W, row0*W and rows*W are multiples of 4. */
int curl_layer_fwd_slab_f32(const float* img, const void* mask, int mask_kind,
                            const float* rawL, const float* rawR, const float* rawH,
                            float* out, float* reg, void* workspace, size_t workspace_bytes,
                            int B, int H, int W, int row0, int rows, int Kl, int Kr, int Kh,
                            unsigned flags, curl_stream_t stream);

/* replaces: torch autograd through CURLLayer.forward (model.py:137-176 over curves.py / colors.py), i.e. what
 *           loss.backward() runs for this layer in main.py:287.  One pass over the pixels (forward chain
 *           recomputed in registers) + a per-image pass for the knots.
 * grad_out [B,3,H,W]; grad_reg [B] (nullable = zeros): gradients of the two outputs of curl_layer_fwd_f32.
 * Outputs: grad_img [B,3,H,W] (nullable), grad_rawL/R/H shaped like rawL/R/H (ASSIGNED, required).
 * grad_img == NULL -- the training step's case, the image being data (main.py:287) -- runs a kernel that stops at the
 * curves' sums: no RGB2LAB pullback, 25 instead of 37 B/px, 12-16 % less time; the knot gradients are the same
 * arithmetic (equal to the other variant's within an ulp of the sums: the compiler's fma contraction may differ).
 * workspace: curl_workspace_bytes(B, 3*Kl+3*Kr+4*Kh); scratch: curl_layer_bwd_scratch_bytes(B,H,W) bytes
 * (block partial sums; reduced in a fixed order in float64 -- no float atomics, results are reproducible). */
size_t curl_layer_bwd_scratch_bytes(int B, int H, int W);
int curl_layer_bwd_f32(const float* img, const void* mask, int mask_kind,
                       const float* rawL, const float* rawR, const float* rawH,
                       const float* grad_out, const float* grad_reg,
                       float* grad_img, float* grad_rawL, float* grad_rawR, float* grad_rawH,
                       void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                       int B, int H, int W, int Kl, int Kr, int Kh,
                       unsigned flags, curl_stream_t stream);

/* replaces: TriSpaceRegNet.generate_residual + generate_image  model.py:499-520 -- the per-pixel path of the
 *           fork's live model (infer.py:44-45, main.py:283) -- with polylayer = Deg4MobilePolyLayer
 *           (model.py:336-415) or ChannelPolyLayer(degree=4) (model.py:206-333), as ONE pass over the pixels.
 * coeffs [B,3,3,num_coeffs] = the reshaped head output (model.py:523-526; [:,0]=R, [:,1]=L, [:,2]=H).
 * num_coeffs 126 = spatial model (5 variables: colour + x/W + y/H), 35 = non-spatial (3 variables).
 * coeffs must be 8-byte aligned (CURL_E_SHAPE otherwise; true of any allocation's start): the kernels copy an image's
 * table into LDS as 8-byte pairs.  The same holds for the u8 and backward entry points below.
 * Default output: clamp(img + residual, 0, 1); with CURL_F_RESIDUAL_ONLY the residual itself. */
int curl_trispace_fwd_f32(const float* img, const float* coeffs, float* out, int B, int H, int W,
                          int num_coeffs, unsigned flags, curl_stream_t stream);

/* Rows [row0, row0 + rows) of every image only (see curl_layer_fwd_slab_f32): img and out are the base pointers of
 * the full [B,3,H,W] tensors.  The polynomial's coordinates stay those of the FULL image -- cat_coords' y = row /
 * H (model.py:487-497) with the image's row index and height, not the slab's -- so the rows written are bit-identical
 * to the same rows of curl_trispace_fwd_f32 on the whole image. */
int curl_trispace_fwd_slab_f32(const float* img, const float* coeffs, float* out, int B, int H, int W,
                               int row0, int rows, int num_coeffs, unsigned flags, curl_stream_t stream);

/* replaces: the file-to-file inference path of infer.py:35-47 in ONE launch -- TF.to_tensor (byte/255), the
 *           full-resolution generate_residual + generate_image, `out*tmask + (1-tmask)` and to_pil_image's
 *           mul(255).byte() -- on interleaved bytes: 6 B/px instead of 15 + 24 + 16 through the f32 entry points.
 * img, out: [B,H,W,3] uint8 (PIL's RGB layout).  white_mask: [B,H,W] uint8 ('L' image, m = byte/255) or NULL
 * (no compositing).  flags must be 0 (the byte output is the image, never the residual). */
int curl_trispace_fwd_u8hwc(const uint8_t* img, const float* coeffs, const uint8_t* white_mask, uint8_t* out,
                            int B, int H, int W, int num_coeffs, unsigned flags, curl_stream_t stream);

/* Same file-edge fusion for the curve layer (CURLLayer.forward between evaluate.py:64-66-style byte images):
 * byte/255 -> curl_layer_fwd_f32 semantics (mask, knots, reg as there) -> [white background] -> truncating *255. */
int curl_layer_fwd_u8hwc(const uint8_t* img, const void* mask, int mask_kind, const float* rawL, const float* rawR,
                         const float* rawH, const uint8_t* white_mask, uint8_t* out, float* reg, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, int Kl, int Kr, int Kh, unsigned flags,
                         curl_stream_t stream);

/* replaces: autograd of curl_trispace_fwd_f32 w.r.t. the coefficients (what loss.backward() runs through
 *           TriSpaceRegNet.generate_residual, main.py:287) -- the image is data, its gradient is not produced.
 * grad_out [B,3,H,W] -> grad_coeffs [B,3,3,num_coeffs] (ASSIGNED).  flags: CURL_F_RESIDUAL_ONLY as in the forward.
 * Three passes: per-pixel upstream gradients, register-tiled outer products with the monomials, fixed-order
 * float64 reduction (no atomics).  scratch: curl_trispace_bwd_scratch_bytes (72 B per pixel + tile partials). */
size_t curl_trispace_bwd_scratch_bytes(int B, int H, int W, int num_coeffs);
int curl_trispace_bwd_f32(const float* img, const float* coeffs, const float* grad_out, float* grad_coeffs,
                          void* scratch, size_t scratch_bytes, int B, int H, int W, int num_coeffs,
                          unsigned flags, curl_stream_t stream);

/* replaces: ChannelPolyLayer(degree=4).forward / Deg4MobilePolyLayer.forward  model.py:295-333, 399-415
 * img [B,num_variables,H,W] (num_variables 5 or 3), coeffs [B,3,num_coeffs] -> out [B,3,H,W]. */
int curl_poly_layer_f32(const float* img, const float* coeffs, float* out, int B, int H, int W,
                        int num_variables, curl_stream_t stream);

/* replaces: PIL + TF.to_tensor + transpose.swapimdims_HW3_3HW at the file edge
 *           infer.py:35-40, data.py:133-158, transpose.py:19-31
 * in: uint8 [B,H,W,Cin] with Cin = 3 or 4 (alpha dropped); out: float32 [B,3,H,W] = value/255. */
int curl_u8hwc_to_f32chw(const uint8_t* in, float* out, int B, int H, int W, int Cin, curl_stream_t stream);
/* replaces: (x*255).astype('uint8') + transpose.swapimdims_3HW_HW3   evaluate.py:64-66, transpose.py:4-16
 * TRUNCATING conversion (values are expected in [0,1]; outside, they saturate to 0/255). */
int curl_f32chw_to_u8hwc(const float* in, uint8_t* out, int B, int H, int W, curl_stream_t stream);

/* replaces: `out_img * tmask + (1 - tmask)` + TF.to_pil_image (mul(255).byte())   infer.py:46-47
 * White background where the mask is 0, then the truncating u8 HWC egress, in one pass.
 * mask [B,1,H,W], mask_kind CURL_MASK_U8 or CURL_MASK_F32. */
int curl_compose_white_u8hwc(const float* in, const void* mask, int mask_kind, uint8_t* out,
                             int B, int H, int W, curl_stream_t stream);

/* replaces: PSNRMetric.compute_psnr  metric.py:35-68 (per-image part)
 * psnr[b] = 10 log10(max^2 / mse_b), mse_b = sum((clamp(a)*m - clamp(b)*m)^2) / (3 * sum(m)) over image b.
 * The batch nan-mean of metric.py:66-67 is a [B]-sized host reduction.  mask_kind NONE = all ones.
 * scratch: curl_psnr_scratch_bytes(B,H,W) bytes; fixed-order float64 reduction (reproducible). */
size_t curl_psnr_scratch_bytes(int B, int H, int W);
int curl_psnr_f32(const float* a, const float* b, const void* mask, int mask_kind, float* psnr,
                  void* scratch, size_t scratch_bytes, int B, int H, int W, float max_intensity,
                  curl_stream_t stream);

/* replaces: MSSSIMMetric.compute_ssim per pyramid level + F.avg_pool2d between levels (metric.py:120-166,
 *           185-192) -- the term of CURLLoss.forward at model.py:103-105 and the MS-SSIM of evaluate.py.
 * a, b: [B,C,H,W] float32 (C = 1 for the loss's L planes).  ssims, mcs: [B,5] = per level, the per-image means of
 * the SSIM map and of the contrast-structure map; the caller finishes metric.py:194-208 ((x+1)/2, powers, product)
 * on those ten numbers per image.  window_size odd, <= 11 (Gaussian, sigma 1.5, zero padding); H, W >= 32 (below that the
 * reference's fifth avg_pool2d raises).
 * One launch per level (32x32 tiles, separable window through LDS, all five blurs at once, next level written by
 * the same block) + a fixed-order float64 reduction: deterministic.  scratch: curl_msssim_scratch_bytes, 16-aligned. */
size_t curl_msssim_scratch_bytes(int B, int C, int H, int W);
int curl_msssim_fwd_f32(const float* a, const float* b, float* ssims, float* mcs, void* scratch, size_t scratch_bytes,
                        int B, int C, int H, int W, int window_size, curl_stream_t stream);
/* replaces: autograd of the above w.r.t. `a` (the prediction; `b` is the target): g_ssims, g_mcs [B,5] are
 * d loss / d ssims, d loss / d mcs; grad_a [B,C,H,W] is ASSIGNED.  Stateless: rebuilds the pyramids in scratch. */
int curl_msssim_bwd_f32(const float* a, const float* b, const float* g_ssims, const float* g_mcs, float* grad_a,
                        void* scratch, size_t scratch_bytes, int B, int C, int H, int W, int window_size,
                        curl_stream_t stream);

/* replaces: the per-pixel terms of CURLLoss.forward  model.py:89-109 (masked L1 in RGB, cosine similarity,
 *           L1 in clamped Lab, L1 on the HSV cone) in one pass over prediction and target.
 * sums [B,5] float64 (ASSIGNED), per image: sum|p-t|, sum cos_sim, sum|lab_p-lab_t|, sum|cone_p-cone_t|, sum(mask).
 * The caller forms model.py:93-109 from them: unmasked = 3*sum(mask) over the batch;
 * cosine term = 1 - mean(cos) - mean(not mask) (what model.py:98's broadcast [B,B,H,W] mean evaluates to).
 * L_pred / L_target [B,1,H,W] (nullable): clamped L planes for the MS-SSIM term (model.py:103-105), which stays
 * stock PyTorch.  scratch: curl_loss_terms_scratch_bytes(B,H,W). */
size_t curl_loss_terms_scratch_bytes(int B, int H, int W);
int curl_loss_terms_f32(const float* pred, const float* target, const void* mask, int mask_kind, double* sums,
                        float* L_pred, float* L_target, void* scratch, size_t scratch_bytes,
                        int B, int H, int W, curl_stream_t stream);
/* Backward of the above w.r.t. pred.  weights: DEVICE pointer to 4 floats = d loss / d (each of the four sums);
 * grad_L_pred [B,1,H,W] (nullable): d loss / d L_pred from the MS-SSIM branch.  grad_pred [B,3,H,W] ASSIGNED. */
int curl_loss_terms_bwd_f32(const float* pred, const float* target, const void* mask, int mask_kind,
                            const float* weights, const float* grad_L_pred, float* grad_pred,
                            int B, int H, int W, curl_stream_t stream);

/* replaces: the forward half of a training step of the curve model -- `net_output = net(input, mask)` then
 *           `loss = criterion(net_output, gt, mask)` (main.py:283-285), i.e. CURLLayer.forward (model.py:137-176) followed by
 *           CURLLoss.forward's pointwise terms (model.py:89-109) -- as ONE pass over the pixels: the loss terms are taken on
 *           the prediction while it is in registers (no 12 B/px round trip, no second read of the mask, one launch less;
 *           launches of <= 2 048 workgroups also collapse their curves inside the kernel).
 * Outputs are those of curl_layer_fwd_f32 (out, reg, the workspace row -- hand it to curl_layer_bwd_f32 with
 * CURL_F_WS_READY) and of curl_loss_terms_f32 (sums [B,5] float64, L_pred / L_target [B,1,H,W], nullable), the same
 * bits as the two calls give.  mask / mask_kind: ONE mask for both (main.py passes the same tensor twice).
 * workspace: curl_workspace_bytes(B, 3*Kl+3*Kr+4*Kh); scratch: curl_loss_terms_scratch_bytes(B,H,W).  flags: 0. */
int curl_layer_loss_fwd_f32(const float* img, const void* mask, int mask_kind,
                            const float* rawL, const float* rawR, const float* rawH, const float* target,
                            float* out, float* reg, double* sums, float* L_pred, float* L_target,
                            void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                            int B, int H, int W, int Kl, int Kr, int Kh, unsigned flags, curl_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* CURL_HIP_H */
