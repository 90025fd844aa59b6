// Developer tool: times gemm_rd_kernel (activation operand direct to registers) against the product kernel on one shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/experiments/gemm_rd/gemm_rd_lab.hip -o tools/experiments/gemm_rd/gemm_rd_lab
//   gemm_rd_lab [J=43904] [rounds=5]          (K = 512, I = 512)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../../convofusion_amd/csrc/gemm_sp.hpp"
#include "../../../convofusion_amd/csrc/rows.hpp"
#include "gemm_rd.hpp"
int g_cfd_naive_gemm = 0;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int J = argc > 1 ? atoi(argv[1]) : 43904, rounds = argc > 2 ? atoi(argv[2]) : 5, K = 512, I = 512;
  float *xf, *yf, *out, *ref;
  char *xs, *ys;
  CK(hipMalloc(&xf, (size_t)I * K * 4)); CK(hipMalloc(&yf, (size_t)J * K * 4)); CK(hipMalloc(&xs, (size_t)I * K * 4)); CK(hipMalloc(&ys, (size_t)J * K * 4));
  CK(hipMalloc(&out, (size_t)J * I * 4)); CK(hipMalloc(&ref, (size_t)J * I * 4));
  long long n = (long long)I * K / 4;
  hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, xf, 1, I * K, 1ull, 0u, 0u, 3u, 0.05f);
  n = (long long)J * K / 4;
  hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, yf, 1, (int)((long long)J * K), 2ull, 0u, 0u, 3u, 1.0f);
  n = (long long)I * (K / 8);
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, xf, xs, (long long)I, K, (long long)K, (long long)K * 4);
  n = (long long)J * (K / 8);
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, yf, ys, (long long)J, K, (long long)K, (long long)K * 4);
  CK(hipDeviceSynchronize());
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.nslot = 1;
  a.X[0] = xs; a.ldx[0] = (long long)K * 4; a.I[0] = I; a.Iclamp[0] = I; a.kt[0] = K / 32;
  a.Y = ys; a.ldy = (long long)K * 4; a.J = J; a.Jclamp = J;
  EpiResid er{out, 0, nullptr};
  EpiNull en{out};
  RdArgs ra{xs, ys, (long long)K * 4, (long long)K * 4, I, J, out, 0};
  struct V { const char* name; int kind; std::vector<float> ms; };
  std::vector<V> vs = {{"product 128x128, no store", 0, {}}, {"product 128x128, residual", 1, {}}, {"reg-direct NS=4 PF=2, no store", 2, {}},
                       {"reg-direct NS=4 PF=2, residual", 3, {}}, {"reg-direct NS=4 PF=3, no store", 4, {}}, {"reg-direct NS=4 PF=3, residual", 5, {}}};
  auto go = [&](int kind) -> hipError_t {
    RdArgs r = ra;
    switch (kind) {
      case 0: return launch_gemm<MODE_PLAIN, EpiNull>(a, en, 1, 1, nullptr, 1);
      case 1: return launch_gemm<MODE_PLAIN, EpiResid>(a, er, 1, 1, nullptr, 1);
      case 2: r.epi = 0; return launch_gemm_rd<16, 4, 2>(r, nullptr);
      case 3: r.epi = 1; return launch_gemm_rd<16, 4, 2>(r, nullptr);
      case 4: r.epi = 0; return launch_gemm_rd<16, 4, 3>(r, nullptr);
      default: r.epi = 1; return launch_gemm_rd<16, 4, 3>(r, nullptr);
    }
  };
  // check the residual variants against the product kernel: out = 0 -> out += D
  std::vector<float> h0((size_t)J * I), h1((size_t)J * I);
  CK(hipMemset(out, 0, (size_t)J * I * 4)); CK(go(1)); CK(hipDeviceSynchronize()); CK(hipMemcpy(h0.data(), out, h0.size() * 4, hipMemcpyDeviceToHost));
  for (int kind : {3, 5}) {
    CK(hipMemset(out, 0, (size_t)J * I * 4)); CK(go(kind)); CK(hipDeviceSynchronize()); CK(hipMemcpy(h1.data(), out, h1.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0; size_t neq = 0;
    for (size_t q = 0; q < h0.size(); ++q) { const double d = (double)h1[q] - h0[q]; num += d * d; den += (double)h0[q] * h0[q]; neq += h1[q] != h0[q]; }
    printf("check kind %d vs product kernel: rel L2 %.3e, %zu of %zu elements differ\n", kind, std::sqrt(num / den), neq, h0.size());
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, nullptr));
      for (int it = 0; it < 10; ++it) (void)go(v.kind);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / 10);
    }
  const double fl = 2.0 * I * (double)J * K;
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    printf("J=%d  %-34s median %7.1f us  min %7.1f us  %6.1f TF algorithmic, %5.1f %% of f16 MFMA peak issued\n", J, v.name, v.ms[v.ms.size() / 2] * 1e3,
           v.ms[0] * 1e3, fl / v.ms[v.ms.size() / 2] / 1e9, 3 * fl / v.ms[v.ms.size() / 2] / 1e9 / 2500 * 100);
  }
  return 0;
}
