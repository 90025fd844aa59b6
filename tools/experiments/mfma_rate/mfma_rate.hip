// Developer tool (GPU box): cycles per MFMA of the fp16 forms on gfx950, one wave per SIMD, back-to-back issue on 8 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_rate/mfma_rate.hip -o tools/experiments/mfma_rate/mfma_rate && tools/experiments/mfma_rate/mfma_rate
// Question behind it: would a 16-key P.V step (v_mfma_f32_16x16x16_f16) cost half of the 32-key one (16x16x32)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* out, long long* cyc, int iters) {
  f4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  h8 a8, b8;
  h4 a4, b4;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(0.001f * (threadIdx.x + e)); b8[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
  for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  // (inline assembly on fixed accumulators: written with the builtin, hipcc rotates the eight accumulators through the AGPR file between
  //  iterations -- ~50 v_accvgpr moves per 8 MFMAs -- and the loop measures those)
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n"
                   "v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n"
                   "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_f16 %7, %8, %9, %7\n"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                   : "v"(a8), "v"(b8));
    else
      asm volatile("v_mfma_f32_16x16x16_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x16_f16 %1, %8, %9, %1\n v_mfma_f32_16x16x16_f16 %2, %8, %9, %2\n"
                   "v_mfma_f32_16x16x16_f16 %3, %8, %9, %3\n v_mfma_f32_16x16x16_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x16_f16 %5, %8, %9, %5\n"
                   "v_mfma_f32_16x16x16_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x16_f16 %7, %8, %9, %7\n"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                   : "v"(a4), "v"(b4));
  }
  asm volatile("s_nop 15\ns_nop 15" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 20000;
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 4; ++rep) {
      const int nblk = rep < 2 ? 256 : 8;       // the whole chip (power-limited clock) / 8 CUs
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(rate_kernel<1>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfmas = (double)iters * 8;           // per wave
      const double flop = mfmas * (kind == 0 ? 16384.0 : 8192.0) * nblk * 4;   // blocks x 4 waves
      printf("%s: %.3f ms for %d x 8 MFMAs per wave, one wave per SIMD on %d CUs: %.1f ns per MFMA (%.1f cycles at 2.4 GHz), %.0f TFLOP/s\n",
             kind == 0 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x16_f16", ms, iters, nblk, ms * 1e6 / mfmas, ms * 1e6 / mfmas * 2.4, flop / ms / 1e9);
    }
  }
  return 0;
}
