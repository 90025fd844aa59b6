// Round-2 experiment: the split-pair GEMM with the ACTIVATION operand taken straight from global memory into
// registers (in MFMA B-operand layout) instead of through LDS.  The product kernel keeps 64 KB per CU in flight (two workgroups x one
// 32 KB stage), which at the ~2 us loaded latency of the activation stream is what bounds it; the register file (512 KB per CU) is a
// bigger landing zone than LDS (160 KB).  Here: 128 x 128 tile, 2 x 2 waves of 64 x 64; the weight tile (X, L2-resident) still comes
// through an LDS ring of NS stages by the LDS-DMA, the activation fragments (Y, streamed from HBM) are requested PF k-steps ahead
// into registers (8 loads of 16 B per lane and k-step, 32 VGPRs per k-step in flight).  The k-loop is fully unrolled (NKT k-tiles
// of 32), so hipcc's own wait-count bookkeeping is exact and the waits are counted.
//   D[i][j] = sum_k X[i][k] Y[j][k]     X: SP [I][K] (weights), Y: SP [J][K] (tokens); epilogues: none (timing) / x[j][i] += D
#pragma once
#include "../../../convofusion_amd/csrc/cfd_common.hpp"

struct RdArgs {
  const char* X;
  const char* Y;
  long long ldx, ldy;   // bytes
  int I, J;
  float* x;             // residual stream [J][I] (epilogue 1) or sink
  int epi;              // 0: keep the accumulators live, store nothing; 1: x[j][i] += D[i][j]
};

template <int NKT, int NS, int PF>
__global__ void __launch_bounds__(256, 2) gemm_rd_kernel(const RdArgs a) {
  static_assert(PF >= 1 && PF < NS && PF <= 3, "prefetch distance");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XSTAGE = 128 * 128;     // 128 weight rows x 128 B per k-tile
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wid >> 1, wj = wid & 1;
  const int l15 = lane & 15, q4 = lane >> 4, sw = l15 >> 1;
  const int tiles_i = a.I / 128;
  const int ti_blk = blockIdx.x % tiles_i, tj_blk = blockIdx.x / tiles_i;
  const int i0 = ti_blk * 128, j0 = tj_blk * 128;

  // X staging: 16 groups of 8 rows per k-tile, 4 per wave
  const int cpos = lane & 7, rsub = lane >> 3;
  long long xoff[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int r = (wid + 4 * n) * 8 + rsub;
    xoff[n] = (long long)(i0 + r) * a.ldx + ((cpos ^ ((r >> 1) & 7)) << 4);
  }
  auto stage_x = [&](int kt) __attribute__((always_inline)) {
    char* sb = smem + (kt % NS) * XSTAGE;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.X + xoff[n] + (long long)kt * 128), (lptr_t)(sb + (wid + 4 * n) * 1024), 16, 0, 0);
  };
  // Y fragments: lane (n = l15, g = q4) supplies k-slots 8g..8g+7 of row j0 + wj*64 + tj*16 + n
  const char* yrow[4];
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) yrow[tj] = a.Y + (long long)min(j0 + wj * 64 + tj * 16 + l15, a.J - 1) * a.ldy + q4 * 16;
  spx8 yh[PF + 1][4], yl[PF + 1][4];
  auto load_y = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      yh[kt % (PF + 1)][tj] = *reinterpret_cast<const spx8*>(yrow[tj] + (long long)kt * 128);
      yl[kt % (PF + 1)][tj] = *reinterpret_cast<const spx8*>(yrow[tj] + (long long)kt * 128 + 64);
    }
  };
  const int xoff_h = (wi * 64 + l15) * 128 + ((q4 ^ sw) << 4);
  const int xoff_l = (wi * 64 + l15) * 128 + (((4 + q4) ^ sw) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int OPS = 12;   // vector-memory requests of one wave per k-step: 4 X groups + 8 Y fragments
#pragma unroll
  for (int kt = 0; kt < PF; ++kt) {
    stage_x(kt);
    load_y(kt);
  }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (kt + PF < NKT) {
      stage_x(kt + PF);      // ring slot (kt+PF) % NS was last read in iteration kt + PF - NS < kt: free since that iteration's barrier
      load_y(kt + PF);
    }
    // everything of k-step kt has landed: at most the requests of the younger k-steps are outstanding
    constexpr int dummy = 0;
    (void)dummy;
    const int younger = (NKT - 1 - kt) < PF ? (NKT - 1 - kt) : PF;
    if (younger == 3) __builtin_amdgcn_s_waitcnt(((3 * OPS) & 15) | 0x70 | (0xF << 8) | ((((3 * OPS) >> 4) & 3) << 14));
    else if (younger == 2) __builtin_amdgcn_s_waitcnt(((2 * OPS) & 15) | 0x70 | (0xF << 8) | ((((2 * OPS) >> 4) & 3) << 14));
    else if (younger == 1) __builtin_amdgcn_s_waitcnt(((1 * OPS) & 15) | 0x70 | (0xF << 8) | ((((1 * OPS) >> 4) & 3) << 14));
    else __builtin_amdgcn_s_waitcnt(0x70 | (0xF << 8));
    __builtin_amdgcn_s_barrier();          // the X tile of k-step kt is visible to every wave
    const char* sb = smem + (kt % NS) * XSTAGE;
    spx8 xh[4], xl[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      xh[ti] = *reinterpret_cast<const spx8*>(sb + xoff_h + ti * 2048);
      xl[ti] = *reinterpret_cast<const spx8*>(sb + xoff_l + ti * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        acc[ti][tj] = SP_MFMA(xl[ti], yh[kt % (PF + 1)][tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yl[kt % (PF + 1)][tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yh[kt % (PF + 1)][tj], acc[ti][tj], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  // epilogue (native MFMA layout: a lane owns 4 consecutive i of row j)
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      const int i = i0 + (wi * 4 + ti) * 16 + q4 * 4, j = j0 + (wj * 4 + tj) * 16 + l15;
      const f32x4 v = acc[ti][tj];
      if (a.epi == 1) {
        if (j < a.J) {
          float4* p = reinterpret_cast<float4*>(a.x + (long long)j * a.I + i);
          float4 r = *p;
          r.x += v[0]; r.y += v[1]; r.z += v[2]; r.w += v[3];
          *p = r;
        }
      } else {
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
      }
    }
}

template <int NKT, int NS, int PF>
static hipError_t launch_gemm_rd(const RdArgs& a, hipStream_t st) {
  static bool attr = false;
  constexpr int lds = NS * 128 * 128;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rd_kernel<NKT, NS, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int tiles = (a.I / 128) * ((a.J + 127) / 128);
  hipLaunchKernelGGL((gemm_rd_kernel<NKT, NS, PF>), dim3(tiles), dim3(256), lds, st, a);
  return hipGetLastError();
}
