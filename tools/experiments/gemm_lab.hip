// Developer tool: stand-alone timing / checking harness for the split-pair MFMA GEMM (gemm_sp.hpp), built as a
// plain HIP executable so that kernel experiments do not need the whole library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/experiments/gemm_lab.hip -o tools/experiments/gemm_lab
//   (with r06_variants/gemm_register_staged_prefetch.patch applied and -DCFD_GEMM_RS=1: cfg 31 = the 128 x 128 tile with register-staged prefetch)
//   gemm_lab J K I rounds  epi:cfg [epi:cfg ...]      epi: r = residual RMW, o = residual RMW without the late
//                                                      prefetch, n = no stores, f = fp32 store, s = SP store
// Variants are timed interleaved, `rounds` times, in one process (median and min are printed); the first launch of
// every variant is checked against the one-thread-per-output kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../convofusion_amd/csrc/gemm_sp.hpp"
#include "gemm_pp.hpp"
#include "../../convofusion_amd/csrc/rows.hpp"

int g_cfd_naive_gemm = 0;
int g_cfd_gemm_cfg = 0;
int g_cfd_small3 = 1;

struct EpiResidOld : EpiResid {
  static constexpr bool kLate = false;
};

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

struct Variant {
  char epi;
  int cfg;
  std::vector<float> ms;
};

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: gemm_lab J K I rounds epi:cfg ...\n");
    return 2;
  }
  const int J = atoi(argv[1]), K = atoi(argv[2]), I = atoi(argv[3]), rounds = atoi(argv[4]);
  std::vector<Variant> vs;
  for (int k = 5; k < argc; ++k) vs.push_back(Variant{argv[k][0], atoi(argv[k] + 2), {}});
  float *xf, *yf, *out, *ref;
  char *xs, *ys, *osp;
  CK(hipMalloc(&xf, (size_t)I * K * 4));
  CK(hipMalloc(&yf, (size_t)J * K * 4));
  CK(hipMalloc(&xs, (size_t)I * K * 4));
  CK(hipMalloc(&ys, (size_t)J * K * 4));
  CK(hipMalloc(&out, (size_t)J * I * 4));
  CK(hipMalloc(&ref, (size_t)J * I * 4));
  CK(hipMalloc(&osp, (size_t)J * I * 4));
  long long n = (long long)I * K / 4;
  hipLaunchKernelGGL(philox_fill_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, xf, 1, I * K, 1ull, 0u, 0u, 3u, 0.05f);
  n = (long long)J * K / 4;
  hipLaunchKernelGGL(philox_fill_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, yf, 1, (int)((long long)J * K), 2ull, 0u, 0u, 3u, 1.0f);
  n = (long long)I * (K / 8);
  hipLaunchKernelGGL(to_split_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, xf, xs, (long long)I, K, (long long)K, (long long)K * 4, (unsigned int*)nullptr);
  n = (long long)J * (K / 8);
  hipLaunchKernelGGL(to_split_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, yf, ys, (long long)J, K, (long long)K, (long long)K * 4, (unsigned int*)nullptr);
  CK(hipDeviceSynchronize());

  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.nslot = 1;
  a.X[0] = xs; a.ldx[0] = (long long)K * 4; a.I[0] = I; a.Iclamp[0] = I; a.kt[0] = K / 32;
  a.Y = ys; a.ldy = (long long)K * 4; a.J = J; a.Jclamp = J;
  EpiF32 ef;
  memset(&ef, 0, sizeof(ef));
  ef.out = out; ef.ldo = I;
  EpiResid er{out, 0, nullptr};
  EpiResidOld eo;
  eo.x = out; eo.obs = 0; eo.bias = nullptr;
  EpiNull en{out};
  EpiSplit es{osp, (long long)I * 4, 0, 0, nullptr, 0, 0};

  auto go = [&](const Variant& v) -> hipError_t {
    if (v.cfg >= 100) {   // persistent ping-pong kernel (cfg = 100 + number of workgroups / 8, 100 -> 256 workgroups)
      PPArgs pa{xs, ys, (long long)K * 4, (long long)K * 4, I, J, K / 32, nullptr, 0, 0};
      const int nb = v.cfg == 100 ? 256 : (v.cfg - 100) * 8;
      switch (v.epi) {
        case 'r': return launch_gemm_pp<PPResid>(pa, PPResid{out}, nullptr, nb);
        case 'n': return launch_gemm_pp<PPNull>(pa, PPNull{out}, nullptr, nb);
        case 's': return launch_gemm_pp<PPSplit<false>>(pa, PPSplit<false>{osp, (long long)I * 4}, nullptr, nb);
        case 'g': return launch_gemm_pp<PPSplit<true>>(pa, PPSplit<true>{osp, (long long)I * 4}, nullptr, nb);
        case 'f': return launch_gemm_pp<PPF32>(pa, PPF32{out, (long long)I, 0}, nullptr, nb);
        default: return hipErrorInvalidValue;
      }
    }
    switch (v.epi) {
      case 'r': return launch_gemm<MODE_PLAIN, EpiResid>(a, er, 1, 1, nullptr, v.cfg);
      case 'o': return launch_gemm<MODE_PLAIN, EpiResidOld>(a, eo, 1, 1, nullptr, v.cfg);
      case 'n': return launch_gemm<MODE_PLAIN, EpiNull>(a, en, 1, 1, nullptr, v.cfg);
      case 's': return launch_gemm<MODE_PLAIN, EpiSplit>(a, es, 1, 1, nullptr, v.cfg);
      case 'G': { EpiSplit eg = es; eg.gelu = 1; return launch_gemm<MODE_PLAIN, EpiSplit>(a, eg, 1, 1, nullptr, v.cfg); }
      default: return launch_gemm<MODE_PLAIN, EpiF32>(a, ef, 1, 1, nullptr, v.cfg);
    }
  };

  // reference result (fp32 store)
  {
    EpiF32 e2 = ef;
    e2.out = ref;
    g_cfd_naive_gemm = 1;
    CK((launch_gemm<MODE_PLAIN, EpiF32>(a, e2, 1, 1, nullptr, 1)));
    g_cfd_naive_gemm = 0;
    CK(hipDeviceSynchronize());
  }
  std::vector<float> href((size_t)J * I), hout((size_t)J * I);
  CK(hipMemcpy(href.data(), ref, (size_t)J * I * 4, hipMemcpyDeviceToHost));
  for (auto& v : vs) {
    if (v.epi == 'n' || v.epi == 's' || v.epi == 'g' || v.epi == 'G') continue;
    CK(hipMemset(out, 0, (size_t)J * I * 4));
    CK(go(v));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hout.data(), out, (size_t)J * I * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t q = 0; q < hout.size(); ++q) {
      const double d = (double)hout[q] - href[q];
      num += d * d;
      den += (double)href[q] * href[q];
    }
    printf("check %c:%d  rel L2 vs naive = %.3e\n", v.epi, v.cfg, std::sqrt(num / den));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 10;
  for (auto& v : vs) { CK(go(v)); }
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) {
      CK(hipEventRecord(e0, nullptr));
      for (int it = 0; it < iters; ++it) (void)go(v);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / iters);
    }
  const double fl = 2.0 * I * (double)J * K;
  for (auto& v : vs) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("J=%d K=%d I=%d  %c:%-2d  median %7.1f us  min %7.1f us   %6.1f TF algorithmic, %5.1f %% of f16 MFMA peak issued\n", J, K, I,
           v.epi, v.cfg, med * 1e3, mn * 1e3, fl / med / 1e9, 3 * fl / med / 1e9 / 2500 * 100);
  }
  return 0;
}
