// libcfdenoise: float32 building blocks on device tensors -- the conditioning producers (cfd_linear_act) and the pieces of
// ConvoFusionVae.decode (cfd_layer_norm, cfd_mha, cfd_add, cfd_zero_rows).
#include "cfd_internal.hpp"

// ---- conditioning producers ---------------------------------------------------------------------------------
int enqueue_linear_act(const float* x, long long n_rows, int K, const float* W, const float* b, int N, int act, float* out, hipStream_t st) {
  const long long gy = (n_rows + 31) / 32;
  if (gy > 65535) return fail(CFD_E_ARG, "too many rows for one launch (%lld)", n_rows);
  hipLaunchKernelGGL(linear_act_kernel<>, dim3((unsigned)((N + 63) / 64), (unsigned)gy), dim3(256), 0, st, x, n_rows, K, W, b, N, act, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CFD_OK : fail(CFD_E_HIP, "linear_act_kernel launch failed: %s", hipGetErrorString(e));
}

extern "C" int cfd_linear_act(cfd_handle c, const float* x, long long n_rows, int K, const float* W, const float* b, int N, int act,
                              float* out, void* stream) {
  if (!c || !x || !W || !out || n_rows < 1 || K < 1 || N < 1 || act < 0 || act > 2) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  return enqueue_linear_act(x, n_rows, K, W, b, N, act, out, (hipStream_t)stream);
}

extern "C" int cfd_layer_norm(cfd_handle c, const float* x, long long rows, int D, const float* gamma, const float* beta, float eps,
                              float* out, void* stream) {
  if (!c || !x || !gamma || !beta || !out || rows < 1 || D < 1 || D > 2048) return fail(CFD_E_ARG, "bad argument (D <= 2048)");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(layernorm_f32_kernel<>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out, rows, D, eps);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_mha(cfd_handle c, const float* q, const float* k, const float* v, int Lq, int Lk, int bs, int E, int H,
                       const uint8_t* key_padding_mask, float* out, void* stream) {
  if (!c || !q || !k || !v || !out || Lq < 1 || Lk < 1 || bs < 1 || H < 1 || E % H) return fail(CFD_E_ARG, "bad argument");
  if (E / H > 64 || Lk > MHA_MAX_KEYS) return fail(CFD_E_SHAPE, "cfd_mha supports head_dim <= 64 and <= %d keys", MHA_MAX_KEYS);
  HIPCHK(hipSetDevice(c->cfg.device));
  const long long items = (long long)Lq * bs * H;
  const float scale = (float)(1.0 / std::sqrt((double)(E / H)));
  hipLaunchKernelGGL(mha_f32_kernel<>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, k, v, key_padding_mask, out, Lq, Lk,
                     bs, E, H, scale);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_add(cfd_handle c, float* x, const float* y, size_t numel, void* stream) {
  if (!c || !x || !y || numel < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(add_f32_kernel<>, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)numel);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_zero_rows(cfd_handle c, float* x, const uint8_t* keep, long long rows, int D, void* stream) {
  if (!c || !x || !keep || rows < 1 || D < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  const long long n = rows * D;
  hipLaunchKernelGGL(zero_rows_f32_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, keep, rows, D);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

