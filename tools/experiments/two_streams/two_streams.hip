// Developer experiment: is a chain of dependent read-modify-write kernels on one stream still exact when a second
// stream runs its own chain on its own buffer at the same time?  Each kernel applies one LCG step to every element, with a
// block -> data mapping that changes from kernel to kernel (so every line is touched by a different CU / XCD each time); after K
// kernels every element must equal the K-fold LCG of its start value.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/two_streams/two_streams.hip -o tools/experiments/two_streams/two_streams
//   ./two_streams [kernels per chain] [MiB per buffer] [repetitions] [lds|snap]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) lcg_step(unsigned* x, size_t n, unsigned rot) {
  const size_t blk = ((size_t)blockIdx.x * 7 + rot) % gridDim.x;
  const size_t i = blk * 1024 + threadIdx.x * 4;
  if (i + 3 < n) {
    uint4 v = *reinterpret_cast<uint4*>(x + i);
    v.x = v.x * 1664525u + 1013904223u; v.y = v.y * 1664525u + 1013904223u;
    v.z = v.z * 1664525u + 1013904223u; v.w = v.w * 1664525u + 1013904223u;
    *reinterpret_cast<uint4*>(x + i) = v;
  }
}
// the same step with the element taken through LDS by the LDS-DMA (global_load_lds_dwordx4, as the product's GEMM / attention kernels
// stage their tiles), 64 KB of LDS per workgroup so that two workgroups -- under two streams: of two different launches -- share a CU
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void __launch_bounds__(256) lcg_step_lds(unsigned* x, size_t n, unsigned rot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const size_t blk = ((size_t)blockIdx.x * 7 + rot) % gridDim.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t i = blk * 1024 + threadIdx.x * 4;
  char* slot = smem + ((rot & 3) * 4 + wid) * 1024;      // a different part of the 64 KB each kernel
  __builtin_amdgcn_global_load_lds((gptr_t)(x + blk * 1024 + wid * 256 + lane * 4), (lptr_t)slot, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0070 | (0xF << 8));        // vmcnt(0)
  __builtin_amdgcn_s_barrier();
  uint4 v = *reinterpret_cast<const uint4*>(slot + lane * 16);
  v.x = v.x * 1664525u + 1013904223u; v.y = v.y * 1664525u + 1013904223u;
  v.z = v.z * 1664525u + 1013904223u; v.w = v.w * 1664525u + 1013904223u;
  if (i + 3 < n) *reinterpret_cast<uint4*>(x + i) = v;
}
// a SHORT consumer behind every big step: copies a 256 KB window of the buffer (position changes every kernel) into a log; the log must
// hold the window as it is after exactly k + 1 steps
__global__ void __launch_bounds__(256) snapshot(const unsigned* x, unsigned* log, size_t win0, int k) {
  const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;     // 64 workgroups x 1024 elements
  *reinterpret_cast<uint4*>(log + (size_t)k * 65536 + i) = *reinterpret_cast<const uint4*>(x + win0 + i);
}
__global__ void check_log(const unsigned* log, size_t win_stride, size_t n, int K, unsigned long long* bad) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)K * 65536) return;
  const int k = (int)(idx / 65536);
  const size_t i = ((size_t)k * win_stride) % (n - 65536) / 1024 * 1024 + idx % 65536;
  unsigned v = (unsigned)i;
  for (int q = 0; q <= k; ++q) v = v * 1664525u + 1013904223u;
  if (log[idx] != v) atomicAdd(bad, 1ull);
}
__global__ void check(const unsigned* x, size_t n, unsigned mul, unsigned add, unsigned long long* bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && x[i] != (unsigned)i * mul + add) atomicAdd(bad, 1ull);
}
__global__ void init(unsigned* x, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (unsigned)i;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 500, mib = argc > 2 ? atoi(argv[2]) : 90, reps = argc > 3 ? atoi(argv[3]) : 20;
  const bool use_lds = argc > 4 && argv[4][0] == 'l';
  const bool snap = argc > 4 && argv[4][0] == 's';
  unsigned* logb[2] = {nullptr, nullptr};
  const size_t win_stride = 7777777;
  if (snap) for (int s2 = 0; s2 < 2; ++s2) CHK(hipMalloc(&logb[s2], (size_t)K * 65536 * 4));
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lcg_step_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const size_t n = (size_t)mib * 1024 * 1024 / 4 / 1024 * 1024;
  unsigned* buf[2];
  unsigned long long* bad;
  hipStream_t st[2];
  for (int s = 0; s < 2; ++s) { CHK(hipMalloc(&buf[s], n * 4)); CHK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking)); }
  CHK(hipMalloc(&bad, 16));
  unsigned mul = 1, add = 0;   // K-fold LCG: x -> mul * x + add
  for (int k = 0; k < K; ++k) { mul = mul * 1664525u; add = add * 1664525u + 1013904223u; }
  const unsigned grid = (unsigned)(n / 1024);
  for (int mode = 0; mode < 2; ++mode) {   // 0: one chain after the other, 1: both at once
    unsigned long long total_bad = 0;
    for (int r = 0; r < reps; ++r) {
      CHK(hipMemset(bad, 0, 16));
      for (int s = 0; s < 2; ++s) hipLaunchKernelGGL(init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st[s], buf[s], n);
      CHK(hipDeviceSynchronize());
      for (int k = 0; k < K; ++k)
        for (int s = 0; s < 2; ++s) {
          if (mode == 0 && s == 1) continue;
          if (use_lds) hipLaunchKernelGGL(lcg_step_lds, dim3(grid), dim3(256), 65536, st[s], buf[s], n, (unsigned)(k * 13 + s));
          else hipLaunchKernelGGL(lcg_step, dim3(grid), dim3(256), 0, st[s], buf[s], n, (unsigned)(k * 13 + s));
          if (snap) hipLaunchKernelGGL(snapshot, dim3(64), dim3(256), 0, st[s], buf[s], logb[s], ((size_t)k * win_stride) % (n - 65536) / 1024 * 1024, k);
        }
      if (mode == 0) {
        CHK(hipDeviceSynchronize());
        for (int k = 0; k < K; ++k) {
          if (use_lds) hipLaunchKernelGGL(lcg_step_lds, dim3(grid), dim3(256), 65536, st[1], buf[1], n, (unsigned)(k * 13 + 1));
          else hipLaunchKernelGGL(lcg_step, dim3(grid), dim3(256), 0, st[1], buf[1], n, (unsigned)(k * 13 + 1));
          if (snap) hipLaunchKernelGGL(snapshot, dim3(64), dim3(256), 0, st[1], buf[1], logb[1], ((size_t)k * win_stride) % (n - 65536) / 1024 * 1024, k);
        }
      }
      CHK(hipDeviceSynchronize());
      for (int s = 0; s < 2; ++s) hipLaunchKernelGGL(check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st[s], buf[s], n, mul, add, bad);
      if (snap)
        for (int s = 0; s < 2; ++s)
          hipLaunchKernelGGL(check_log, dim3((unsigned)(((size_t)K * 65536 + 255) / 256)), dim3(256), 0, st[s], logb[s], win_stride, n, K, bad);
      CHK(hipDeviceSynchronize());
      unsigned long long h = 0;
      CHK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
      total_bad += h;
    }
    printf("%s%s: %d repetitions x 2 chains x %d kernels on %d MiB buffers: %llu wrong elements\n", use_lds ? "[LDS-DMA] " : (snap ? "[short consumer behind every step] " : ""), mode ? "two streams at once" : "one stream at a time",
           reps, K, mib, total_bad);
  }
  return 0;
}
