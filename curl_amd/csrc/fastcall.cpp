// fastcall.cpp -- compiled host-side binding of the three hot entry points of include/curl_hip.h:
//   curl_layer_fwd_f32       (CURLLayer.forward, model.py:137-176 -- one frame per call in infer.py:44)
//   curl_layer_bwd_f32       (its autograd, main.py:287 -- the training crop batch of main.py:88 / data.py:86)
//   curl_trispace_fwd_u8hwc  (the file-to-file path of infer.py:35-47)
// A 1500x1000 frame is 12 us of kernel; through curl_amd/ops.py the call cost 16.6 us of HOST time (three decorators, the
// argument checks, four torch.empty, 18 ctypes conversions -- profiles/r04/host_prof_after.log).  This module takes the
// tensors, checks that the call is the PLAIN one (every tensor on the current device, float32 / uint8, contiguous, even knot
// counts), allocates the outputs through torch's caching allocator, reads torch's current stream and calls the C ABI -- one
// Python -> C++ transition.  Anything else (a CPU tensor, a broadcast mask, uneven torch.chunk knots, another
// device, an empty image ...) returns None and curl_amd/ops.py runs its fully checked path, which raises the documented
// errors: the validation semantics stay where they are (tests/test_host_logic.py).
//
// It changes no kernel and is not the boundary: the C ABI is.  The library is NOT linked here: curl_amd/_lib.py hands over the
// addresses of the functions of the library IT loaded (CURL_HIP_LIB may select a variant build), so both surfaces always call
// the same code.  PyTorch is plumbing (device memory, stream).
#include <torch/extension.h>

#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>

#include "../../include/curl_hip.h"

namespace {

struct Abi {
  decltype(&curl_layer_fwd_f32) layer_fwd = nullptr;
  decltype(&curl_layer_bwd_f32) layer_bwd = nullptr;
  decltype(&curl_trispace_fwd_u8hwc) tri_u8 = nullptr;
  decltype(&curl_workspace_bytes) ws_bytes = nullptr;
  decltype(&curl_layer_bwd_scratch_bytes) bwd_scratch = nullptr;
} g;

template <class F>
void take(F& slot, const py::dict& d, const char* name) {
  slot = reinterpret_cast<F>(static_cast<uintptr_t>(d[name].cast<unsigned long long>()));
}

void bind_abi(const py::dict& addresses) {
  take(g.layer_fwd, addresses, "curl_layer_fwd_f32");
  take(g.layer_bwd, addresses, "curl_layer_bwd_f32");
  take(g.tri_u8, addresses, "curl_trispace_fwd_u8hwc");
  take(g.ws_bytes, addresses, "curl_workspace_bytes");
  take(g.bwd_scratch, addresses, "curl_layer_bwd_scratch_bytes");
}

// a tensor this path hands to a kernel as it is: on `dev`, of dtype `dt`, contiguous
inline bool plain(const at::Tensor& t, at::ScalarType dt, c10::DeviceIndex dev) {
  return t.is_cuda() && t.get_device() == dev && t.scalar_type() == dt && t.is_contiguous();
}
// raw knots [B, n * K] with n curves of K knots each, 2 <= K <= CURL_MAX_KNOTS (torch.chunk's uneven split: the checked path)
inline int knots_per_curve(const at::Tensor& t, int64_t B, int n, c10::DeviceIndex dev) {
  if (!plain(t, at::kFloat, dev) || t.dim() != 2 || t.size(0) != B) return 0;
  const int64_t N = t.size(1);
  if (N % n != 0) return 0;
  const int64_t K = N / n;
  return (K >= 2 && K <= CURL_MAX_KNOTS) ? (int)K : 0;
}
// -> mask_kind, or -1 for "not the plain case"
inline int mask_kind(const c10::optional<at::Tensor>& mask, int64_t B, int64_t H, int64_t W, c10::DeviceIndex dev,
                     const void*& ptr) {
  ptr = nullptr;
  if (!mask.has_value() || !mask->defined()) return CURL_MASK_NONE;
  const at::Tensor& m = *mask;
  if (!m.is_cuda() || m.get_device() != dev || !m.is_contiguous() || m.dim() != 4 || m.size(0) != B || m.size(1) != 1 ||
      m.size(2) != H || m.size(3) != W)
    return -1;
  ptr = m.data_ptr();
  const auto dt = m.scalar_type();
  if (dt == at::kBool || dt == at::kByte) return CURL_MASK_U8;
  if (dt == at::kFloat) return CURL_MASK_F32;
  return -1;
}
inline bool plain_image(const at::Tensor& img, c10::DeviceIndex& dev) {
  if (!img.is_cuda()) return false;
  dev = img.get_device();
  return dev == c10::hip::current_device() && img.scalar_type() == at::kFloat && img.dim() == 4 && img.size(1) == 3 &&
         img.numel() > 0 && img.is_contiguous() && img.size(0) <= INT_MAX && img.size(2) <= INT_MAX && img.size(3) <= INT_MAX;
}

// -> (out, reg, workspace) | int (the C ABI's non-zero return code) | None (not the plain call)
py::object layer_fwd(const at::Tensor& img, const c10::optional<at::Tensor>& mask, const at::Tensor& L, const at::Tensor& R,
                     const at::Tensor& Hk, unsigned flags, const c10::optional<at::Tensor>& out_arg) {
  c10::DeviceIndex dev;
  if (!g.layer_fwd || !plain_image(img, dev)) return py::none();
  const int64_t B = img.size(0), H = img.size(2), W = img.size(3);
  const int Kl = knots_per_curve(L, B, 3, dev), Kr = knots_per_curve(R, B, 3, dev), Kh = knots_per_curve(Hk, B, 4, dev);
  const void* mptr;
  const int kind = mask_kind(mask, B, H, W, dev, mptr);
  if (!Kl || !Kr || !Kh || kind < 0) return py::none();
  at::Tensor out;
  if (out_arg.has_value() && out_arg->defined()) {  // written in place (may be img itself: every pixel is read before it is written)
    if (!plain(*out_arg, at::kFloat, dev) || out_arg->sizes() != img.sizes()) return py::none();
    out = *out_arg;
  } else {
    out = at::empty_like(img);
  }
  at::Tensor reg = at::empty({B}, img.options());
  const size_t nbytes = g.ws_bytes((int)B, 3 * Kl + 3 * Kr + 4 * Kh);
  at::Tensor ws = at::empty({(int64_t)(nbytes / 4)}, img.options());
  const int rc = g.layer_fwd(img.data_ptr<float>(), mptr, kind, L.data_ptr<float>(), R.data_ptr<float>(), Hk.data_ptr<float>(),
                             out.data_ptr<float>(), reg.data_ptr<float>(), ws.data_ptr(), nbytes, (int)B, (int)H, (int)W, Kl, Kr,
                             Kh, flags, (curl_stream_t)c10::hip::getCurrentHIPStream(dev).stream());
  if (rc != 0) return py::int_(rc);
  return py::make_tuple(std::move(out), std::move(reg), std::move(ws));
}

// -> (grad_img | None, grad_L, grad_R, grad_H) | int | None
py::object layer_bwd(const at::Tensor& img, const c10::optional<at::Tensor>& mask, const at::Tensor& L, const at::Tensor& R,
                     const at::Tensor& Hk, const at::Tensor& grad_out, const c10::optional<at::Tensor>& grad_reg,
                     bool need_grad_img, const c10::optional<at::Tensor>& workspace, unsigned flags) {
  c10::DeviceIndex dev;
  if (!g.layer_bwd || !plain_image(img, dev)) return py::none();
  if (!plain(grad_out, at::kFloat, dev) || grad_out.sizes() != img.sizes()) return py::none();
  const int64_t B = img.size(0), H = img.size(2), W = img.size(3);
  const int Kl = knots_per_curve(L, B, 3, dev), Kr = knots_per_curve(R, B, 3, dev), Kh = knots_per_curve(Hk, B, 4, dev);
  const void* mptr;
  const int kind = mask_kind(mask, B, H, W, dev, mptr);
  if (!Kl || !Kr || !Kh || kind < 0) return py::none();
  const float* greg = nullptr;
  if (grad_reg.has_value() && grad_reg->defined()) {
    if (!plain(*grad_reg, at::kFloat, dev) || grad_reg->dim() != 1 || grad_reg->size(0) != B) return py::none();
    greg = grad_reg->data_ptr<float>();
  }
  const size_t nbytes = g.ws_bytes((int)B, 3 * Kl + 3 * Kr + 4 * Kh);
  at::Tensor ws;
  if (workspace.has_value() && workspace->defined()) {
    // the tensor curl_layer_fwd_f32 filled for the same knots (CURL_F_WS_READY: no knot-prep launch)
    if (!plain(*workspace, at::kFloat, dev) || (size_t)workspace->numel() * 4 < nbytes) return py::none();
    ws = *workspace;
    flags |= CURL_F_WS_READY;
  } else {
    ws = at::empty({(int64_t)(nbytes / 4)}, img.options());
  }
  const size_t sbytes = g.bwd_scratch((int)B, (int)H, (int)W);
  at::Tensor scratch = at::empty({(int64_t)(sbytes / 4)}, img.options());
  at::Tensor gL = at::empty_like(L), gR = at::empty_like(R), gH = at::empty_like(Hk);
  at::Tensor gimg;
  if (need_grad_img) gimg = at::empty_like(img);
  const int rc = g.layer_bwd(img.data_ptr<float>(), mptr, kind, L.data_ptr<float>(), R.data_ptr<float>(), Hk.data_ptr<float>(),
                             grad_out.data_ptr<float>(), greg, need_grad_img ? gimg.data_ptr<float>() : nullptr,
                             gL.data_ptr<float>(), gR.data_ptr<float>(), gH.data_ptr<float>(), ws.data_ptr(), nbytes,
                             scratch.data_ptr(), sbytes, (int)B, (int)H, (int)W, Kl, Kr, Kh, flags,
                             (curl_stream_t)c10::hip::getCurrentHIPStream(dev).stream());
  if (rc != 0) return py::int_(rc);
  py::object gi = need_grad_img ? py::cast(std::move(gimg)) : py::object(py::none());
  return py::make_tuple(std::move(gi), std::move(gL), std::move(gR), std::move(gH));
}

// -> out uint8 [B,H,W,3] | int | None
py::object trispace_fwd_u8hwc(const at::Tensor& img, const at::Tensor& coeffs, const c10::optional<at::Tensor>& white_mask) {
  if (!g.tri_u8 || !img.is_cuda()) return py::none();
  const c10::DeviceIndex dev = img.get_device();
  if (dev != c10::hip::current_device() || !plain(img, at::kByte, dev) || img.dim() != 4 || img.size(3) != 3 || img.numel() == 0)
    return py::none();
  const int64_t B = img.size(0), H = img.size(1), W = img.size(2);
  if (B > INT_MAX || H > INT_MAX || W > INT_MAX) return py::none();
  if (!plain(coeffs, at::kFloat, dev) || coeffs.dim() != 4 || coeffs.size(0) != B || coeffs.size(1) != 3 || coeffs.size(2) != 3 ||
      (coeffs.size(3) != 126 && coeffs.size(3) != 35) || (reinterpret_cast<uintptr_t>(coeffs.data_ptr()) & 7u))
    return py::none();
  const uint8_t* wm = nullptr;
  if (white_mask.has_value() && white_mask->defined()) {
    const at::Tensor& m = *white_mask;
    if (!plain(m, at::kByte, dev) || m.dim() != 3 || m.size(0) != B || m.size(1) != H || m.size(2) != W) return py::none();
    wm = m.data_ptr<uint8_t>();
  }
  at::Tensor out = at::empty_like(img);
  const int rc = g.tri_u8(img.data_ptr<uint8_t>(), coeffs.data_ptr<float>(), wm, out.data_ptr<uint8_t>(), (int)B, (int)H, (int)W,
                          (int)coeffs.size(3), 0u, (curl_stream_t)c10::hip::getCurrentHIPStream(dev).stream());
  if (rc != 0) return py::int_(rc);
  return py::cast(std::move(out));
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled host-side binding of libcurlhip.so's three hot entry points (see fastcall.cpp)";
  m.def("bind_abi", &bind_abi, "addresses of the C-ABI functions of the library curl_amd._lib loaded");
  m.def("layer_fwd", &layer_fwd, py::arg("img"), py::arg("mask"), py::arg("L"), py::arg("R"), py::arg("H"), py::arg("flags") = 0u,
        py::arg("out") = py::none());
  m.def("layer_bwd", &layer_bwd, py::arg("img"), py::arg("mask"), py::arg("L"), py::arg("R"), py::arg("H"), py::arg("grad_out"),
        py::arg("grad_reg"), py::arg("need_grad_img"), py::arg("workspace"), py::arg("flags") = 0u);
  m.def("trispace_fwd_u8hwc", &trispace_fwd_u8hwc, py::arg("img"), py::arg("coeffs"), py::arg("white_mask"));
}
