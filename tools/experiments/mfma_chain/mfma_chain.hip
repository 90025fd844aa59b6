// Developer tool (GPU box, round 6): what a DEPENDENT chain of v_mfma_f32_16x16x32_f16 costs on gfx950 -- 8 MFMAs per loop iteration on
// NACC = 1, 2, 4, 8 accumulators (1: every MFMA reads the previous one's result as SrcC), one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_chain/mfma_chain.hip -o tools/experiments/mfma_chain/mfma_chain
// Question behind it: the fused cross-attention's score accumulator is ONE MFMA tile per 16 keys, so phase A is a
// chain of 16 - 24 dependent MFMAs per wave; in lock-step two waves of a SIMD interleave their chains, in the half-step ping-pong loop a wave is
// alone in phase A.  Does the chain run at the issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(512) chain_kernel(float* out, int iters) {
  f4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  h8 a8, b8;
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(0.001f * ((threadIdx.x & 63) + e)); b8[e] = (_Float16)(0.002f * ((threadIdx.x & 63) - e)); }
  for (int i = 0; i < iters; ++i) {
    if (NACC == 1)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                   "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                   : "+v"(acc[0]) : "v"(a8), "v"(b8));
    else if (NACC == 2)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n"
                   "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n"
                   : "+v"(acc[0]), "+v"(acc[1]) : "v"(a8), "v"(b8));
    else if (NACC == 4)
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_f16 %3, %4, %5, %3\n"
                   "v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_f16 %3, %4, %5, %3\n"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a8), "v"(b8));
    else
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n v_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n"
                   "v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n v_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n v_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n"
                   "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n v_mfma_f32_16x16x32_f16 %7, %8, %9, %7\n"
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                   : "v"(a8), "v"(b8));
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NACC>
static void run(float* out, int waves_per_simd, int nblk) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<NACC>, dim3(nblk), dim3(256 * waves_per_simd), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)iters * 8 * waves_per_simd;      // MFMAs through one SIMD's matrix pipe
  printf("accumulators %d, %d wave(s) per SIMD, %3d CUs: %6.1f ns per MFMA of the SIMD (%5.1f cycles at 2.4 GHz)\n", NACC, waves_per_simd, nblk,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  for (int nblk : {8, 256})
    for (int w : {1, 2}) { run<1>(out, w, nblk); run<2>(out, w, nblk); run<4>(out, w, nblk); run<8>(out, w, nblk); }
  return 0;
}
