// A stand-alone host of the C ABI (include/curl_hip.h): no PyTorch, no Python -- what a cgo / JNI / C++ maintainer's binding
// amounts to.  Reads float32 inputs from files, runs CURLLayer.forward and its backward through libcurlhip.so on buffers it
// allocated itself with hipMalloc, on a stream it created, and writes the outputs; tests/test_gpu_cabi.py builds it with hipcc
// and compares the files with what the Python surface produces on the same inputs (bit for bit).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "curl_hip.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define CURL_OK_OR_DIE(x)                                                          \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_ != 0) {                                                                \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, curl_last_error());               \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

static std::vector<char> slurp(const std::string& path) {
  std::vector<char> v;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return v;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)n);
  if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
  fclose(f);
  return v;
}
static bool spit(const std::string& path, const void* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = fwrite(p, 1, n, f) == n;
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: %s <dir> B H W   (dir holds img.f32 mask.u8 L.f32 R.f32 H.f32 gout.f32)\n", argv[0]);
    return 1;
  }
  const std::string dir = argv[1];
  const int B = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]);
  const size_t px = (size_t)B * H * W;
  std::vector<char> img = slurp(dir + "/img.f32"), mask = slurp(dir + "/mask.u8"), L = slurp(dir + "/L.f32"), R = slurp(dir + "/R.f32"),
                    Hk = slurp(dir + "/H.f32"), gout = slurp(dir + "/gout.f32");
  if (img.size() != px * 12 || mask.size() != px || L.size() != (size_t)B * 48 * 4 || R.size() != L.size() || Hk.size() != (size_t)B * 64 * 4 ||
      gout.size() != img.size()) {
    fprintf(stderr, "input files missing or of the wrong size\n");
    return 1;
  }
  printf("curl_version %d\n", curl_version());
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *d_img, *d_out, *d_L, *d_R, *d_H, *d_reg, *d_gout, *d_gimg, *d_gL, *d_gR, *d_gH;
  uint8_t* d_mask;
  void *d_ws, *d_scratch;
  const size_t ws_bytes = curl_workspace_bytes(B, 160), scratch_bytes = curl_layer_bwd_scratch_bytes(B, H, W);
  HIP_OK(hipMalloc(&d_img, img.size()));
  HIP_OK(hipMalloc(&d_out, img.size()));
  HIP_OK(hipMalloc(&d_gout, img.size()));
  HIP_OK(hipMalloc(&d_gimg, img.size()));
  HIP_OK(hipMalloc(&d_mask, mask.size()));
  HIP_OK(hipMalloc(&d_L, L.size()));
  HIP_OK(hipMalloc(&d_R, R.size()));
  HIP_OK(hipMalloc(&d_H, Hk.size()));
  HIP_OK(hipMalloc(&d_gL, L.size()));
  HIP_OK(hipMalloc(&d_gR, R.size()));
  HIP_OK(hipMalloc(&d_gH, Hk.size()));
  HIP_OK(hipMalloc(&d_reg, (size_t)B * 4));
  HIP_OK(hipMalloc(&d_ws, ws_bytes));
  HIP_OK(hipMalloc(&d_scratch, scratch_bytes));
  HIP_OK(hipMemcpyAsync(d_img, img.data(), img.size(), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_gout, gout.data(), gout.size(), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_L, L.data(), L.size(), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_R, R.data(), R.size(), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_H, Hk.data(), Hk.size(), hipMemcpyHostToDevice, stream));
  // model.py:137-176 as one call; then its backward with the workspace the forward filled (CURL_F_WS_READY)
  CURL_OK_OR_DIE(curl_layer_fwd_f32(d_img, d_mask, CURL_MASK_U8, d_L, d_R, d_H, d_out, d_reg, d_ws, ws_bytes, B, H, W, 16, 16, 16, 0,
                                    (curl_stream_t)stream));
  CURL_OK_OR_DIE(curl_layer_bwd_f32(d_img, d_mask, CURL_MASK_U8, d_L, d_R, d_H, d_gout, /*grad_reg*/ nullptr, d_gimg, d_gL, d_gR, d_gH, d_ws,
                                    ws_bytes, d_scratch, scratch_bytes, B, H, W, 16, 16, 16, CURL_F_WS_READY, (curl_stream_t)stream));
  // an argument error is a return code and a message, not a crash
  int rc = curl_layer_fwd_f32(d_img, nullptr, CURL_MASK_U8, d_L, d_R, d_H, d_out, d_reg, d_ws, ws_bytes, B, H, W, 16, 16, 16, 0,
                              (curl_stream_t)stream);
  printf("mask_kind set, mask NULL -> %d (%s)\n", rc, curl_last_error());
  if (rc != CURL_E_MASK) return 4;
  std::vector<char> out(img.size()), gimg(img.size()), reg((size_t)B * 4), gL(L.size()), gR(R.size()), gH(Hk.size());
  HIP_OK(hipMemcpyAsync(out.data(), d_out, out.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(gimg.data(), d_gimg, gimg.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(reg.data(), d_reg, reg.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(gL.data(), d_gL, gL.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(gR.data(), d_gR, gR.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(gH.data(), d_gH, gH.size(), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  bool ok = spit(dir + "/out.f32", out.data(), out.size()) && spit(dir + "/reg.f32", reg.data(), reg.size()) &&
            spit(dir + "/gimg.f32", gimg.data(), gimg.size()) && spit(dir + "/gL.f32", gL.data(), gL.size()) &&
            spit(dir + "/gR.f32", gR.data(), gR.size()) && spit(dir + "/gH.f32", gH.data(), gH.size());
  printf("%s\n", ok ? "done" : "could not write outputs");
  return ok ? 0 : 5;
}
