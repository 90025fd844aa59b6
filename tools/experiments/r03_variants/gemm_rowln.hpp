// Row-complete residual product with the FOLLOWING LayerNorm in its epilogue (round 3):
//
//   x[row][:] += bias + Y[row][:] . W^T                      (EpiResid of gemm_sp.hpp: cross_attention.py:572,655,661)
//   out[row][:] = split( act( LN(x[row]) ) )                 (ln_rows_kernel of rows.hpp: the norm1 / norm3 / final LayerNorm, or
//                                                             the TimeBlock's AdaLN + SiLU, cross_attention.py:426-439,568,659)
//
// The 128 x 128 tiles of gemm_sp_kernel hold a quarter of a row, so the LayerNorm that follows every residual product is a
// launch of its own that reads the 90 MB residual stream back (27.6 us at the benchmark shape).  Here a workgroup owns ALL 512
// output features of 64 rows -- 8 waves, wave w = features [64 w, 64 w + 64) x 64 rows, the same 64 x 64 wave tile (4 x 4 MFMA
// tiles, 48 MFMAs per k-step) as the 128 x 128 kernel -- so the row statistics are workgroup-local: the epilogue adds the
// residual, writes x, reduces mean and centred variance through LDS (two passes, like ln_rows_kernel) and stores the
// normalised (+ modulated + SiLU) split-pair operand of the next product.  One workgroup per CU (152 KB of LDS: weight ring of 2
// x 64 KB, activation ring of 3 x 8 KB).
// MEASURED SLOWER than gemm_sp_kernel + ln_rows_kernel at the benchmark shape (one lock-step workgroup per
// CU exposes every memory wait that two independent 4-wave workgroups cover for each other), so it is OFF by default:
// CFD_ROWLN_MIN_ROWS=<n> selects it for residual products of at least n rows (cfd_api.hip: rowln_min_rows); parity-tested
// (tests/test_gpu_sampler.py::test_developer_knobs_keep_parity).
#pragma once
#include "gemm_sp.hpp"

struct RowLnArgs {
  const char* W;      // SP [512][K] weight (row operand, MFMA A)
  const char* Y;      // SP [M][K] activation (column operand, MFMA B)
  int K;              // multiple of 32
  long long M;
  float* x;           // fp32 [M][512] residual stream, updated in place
  const float* bias;  // [512] or null
  // the LayerNorm that follows (same meaning as LnArgs of rows.hpp)
  char* out;          // SP [M][512]
  const float* g;
  const float* b;
  int adaln;
  const float* ss;    // (1 + scale | shift) rows of 1024 floats for this time block, t-row stride ss_tstride
  long long ss_tstride;
  const int* d_step;
  int tmode;          // 0: t-row = *d_step ; 1: t-row = trow0 + row / L
  int L;
  int trow0;
};

#define RL_BI 512
#define RL_BJ 64
#define RL_XS (RL_BI * 128)                 // one stage of the weight tile (64 KB)
#define RL_YS (RL_BJ * 128)                 // one stage of the activation tile (8 KB)
#define RL_YOFF (2 * RL_XS)
#define RL_LDS (2 * RL_XS + 3 * RL_YS)      // X ring of 2, Y ring of 3: 152 KB
#define RL_RS (4 * 64 + 16)                 // epilogue strip row stride in bytes (+16: conflict-free 16-byte writes)
#define RL_STAT_OFF (8 * 16 * RL_RS)        // behind the 8 strips: per pass, partial sums [8 waves][64 rows] + the row statistic [64]

__global__ void __launch_bounds__(512, 2) gemm_rowln_kernel(const RowLnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long j0 = (long long)blockIdx.x * RL_BJ;
  const int nkt = a.K / 32;
  const long long ld = (long long)a.K * 4;

  // ---- staging: a piece = 8 tile rows x 128 B = one global_load_lds_dwordx4 wave-instruction; X tile = 64 pieces (8 per wave),
  //      Y tile = 8 pieces (1 per wave); chunk swizzle (row >> 1) & 7 on the source address (gemm_sp.hpp)
  const int cpos = lane & 7, rsub = lane >> 3;
  long long xoff[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int r = (wid + 8 * n) * 8 + rsub;
    xoff[n] = (long long)r * ld + ((cpos ^ ((r >> 1) & 7)) << 4);
  }
  long long yoff;
  {
    const int r = wid * 8 + rsub;
    const long long row = (j0 + r < a.M) ? j0 + r : a.M - 1;
    yoff = row * ld + ((cpos ^ ((r >> 1) & 7)) << 4);
  }
  auto stage_x = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < 8; ++n)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.W + xoff[n] + (long long)kt * 128), (lptr_t)(smem + buf * RL_XS + (wid + 8 * n) * 1024), 16, 0, 0);
  };
  auto stage_y = [&](int kt, int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((gptr_t)(a.Y + yoff + (long long)kt * 128), (lptr_t)(smem + RL_YOFF + buf * RL_YS + wid * 1024), 16, 0, 0);
  };

  const int l15 = lane & 15, q4 = lane >> 4, sw = l15 >> 1;
  const int xoff_h = (wid * 64 + l15) * 128 + ((q4 ^ sw) << 4);
  const int xoff_l = (wid * 64 + l15) * 128 + (((4 + q4) ^ sw) << 4);
  const int yoff_h = RL_YOFF + l15 * 128 + ((q4 ^ sw) << 4);
  const int yoff_l = RL_YOFF + l15 * 128 + (((4 + q4) ^ sw) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One workgroup per CU: nobody else covers a wait, so the activation tile -- the operand that comes from HBM -- is requested
  // TWO k-steps ahead.  vmcnt retires in order: every wave requests its 8 weight pieces of k-step kt+1 first and its activation
  // piece of k-step kt+2 last, and waits with vmcnt(1): the weights of kt+1 and the (older) activation piece of kt+1 have landed,
  // the youngest request stays in flight across the barrier.
  stage_x(0, 0);
  stage_y(0, 0);
  if (nkt > 1) { stage_y(1, 1); __builtin_amdgcn_s_waitcnt(WAIT_VM_LGKM0(1)); }
  else __builtin_amdgcn_s_waitcnt(WAIT_VM_LGKM0(0));
  __builtin_amdgcn_s_barrier();
  int ybuf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    int yb2 = ybuf + 2;
    if (yb2 >= 3) yb2 -= 3;
    if (kt + 1 < nkt) stage_x(kt + 1, (kt + 1) & 1);     // X buffer (kt+1)&1 and Y buffer (kt+2)%3 were last read in iteration kt-1
    if (kt + 2 < nkt) stage_y(kt + 2, yb2);
    const char* sb = smem + (kt & 1) * RL_XS;
    const char* sy = smem + ybuf * RL_YS;
    spx8 xh[4], xl[4], yh[4], yl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xh[t] = *reinterpret_cast<const spx8*>(sb + xoff_h + t * 2048);
      xl[t] = *reinterpret_cast<const spx8*>(sb + xoff_l + t * 2048);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      yh[t] = *reinterpret_cast<const spx8*>(sy + yoff_h + t * 2048);
      yl[t] = *reinterpret_cast<const spx8*>(sy + yoff_l + t * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);   // all fragment reads of the k-step are issued before its first MFMA
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        acc[ti][tj] = SP_MFMA(xl[ti], yh[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yl[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yh[tj], acc[ti][tj], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nkt) __builtin_amdgcn_s_waitcnt(WAIT_VM_LGKM0(1));
    else __builtin_amdgcn_s_waitcnt(WAIT_VM_LGKM0(0));
    __builtin_amdgcn_s_barrier();
    ybuf = (ybuf == 2) ? 0 : ybuf + 1;
  }

  // ---- epilogue.  Row-major lane mapping (the WIDE mapping of gemm_sp.hpp): 16 lanes per row x 4 consecutive features, 4 rows per
  //      wave instruction; band tj = 16 rows, instruction it: row tj*16 + it*4 + lr.  A lane ends up with its 4 features of 16 rows.
  const int c4 = (lane & 15) * 4, lr = lane >> 4;
  const int col = wid * 64 + c4;
  char* strip = smem + wid * (16 * RL_RS);
  const float4 bias4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* xp = a.x + j0 * CFD_D + col;
  float4 r[4][4];
  // the old values of all 16 rows are requested before the first store (a store to x may alias a later load as far as hipcc knows)
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = tj * 16 + it * 4 + lr;
      r[tj][it] = (j0 + row < a.M) ? *reinterpret_cast<const float4*>(xp + (long long)row * CFD_D) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
  for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
      *reinterpret_cast<f32x4*>(strip + l15 * RL_RS + (ti * 16 + q4 * 4) * 4) = acc[ti][tj];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private strip: no barrier needed
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(strip + (it * 4 + lr) * RL_RS + c4 * 4);
      float4 t = r[tj][it];
      t.x = (t.x + bias4.x) + v[0]; t.y = (t.y + bias4.y) + v[1]; t.z = (t.z + bias4.z) + v[2]; t.w = (t.w + bias4.w) + v[3];   // EpiResid's association
      r[tj][it] = t;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next band overwrites the strip
  }
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = tj * 16 + it * 4 + lr;
      if (j0 + row < a.M) *reinterpret_cast<float4*>(xp + (long long)row * CFD_D) = r[tj][it];
    }

  // row statistics, two passes (mean, then the centred sum of squares) like ln_rows_kernel: 16-lane partial sums of this wave's 64
  // features -> LDS [wave][row] -> one thread per row adds the 8 partials -> LDS -> every lane reads back its 16 rows
  float* stat_base = reinterpret_cast<float*>(smem + RL_STAT_OFF);     // per pass: partial sums [8][64], then the statistic [64]
  auto sum16 = [](float v) __attribute__((always_inline)) -> float {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
  };
  auto reduce_rows = [&](float (&val)[4][4], bool finish_rstd) __attribute__((always_inline)) {
    float* part = stat_base + (finish_rstd ? 9 * 64 : 0);   // (the two passes use disjoint areas: no barrier between them)
    float* stat = part + 8 * 64;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float s = sum16(val[tj][it]);
        if ((lane & 15) == 0) part[wid * 64 + tj * 16 + it * 4 + lr] = s;
      }
    __syncthreads();
    if (threadIdx.x < 64) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += part[w * 64 + threadIdx.x];
      s *= (1.0f / CFD_D);
      stat[threadIdx.x] = finish_rstd ? 1.0f / sqrtf(s + 1e-5f) : s;
    }
    __syncthreads();
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int it = 0; it < 4; ++it) val[tj][it] = stat[tj * 16 + it * 4 + lr];
  };
  float st[4][4];
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int it = 0; it < 4; ++it) st[tj][it] = (r[tj][it].x + r[tj][it].y) + (r[tj][it].z + r[tj][it].w);
  reduce_rows(st, false);   // st = mean of the row
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float4& t = r[tj][it];
      const float m = st[tj][it];
      t.x -= m; t.y -= m; t.z -= m; t.w -= m;
      st[tj][it] = (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    }
  reduce_rows(st, true);    // st = 1 / sqrt(var + eps)

  const float4 g4 = *reinterpret_cast<const float4*>(a.g + col), b4 = *reinterpret_cast<const float4*>(a.b + col);
  const long long trow_all = (a.adaln && !a.tmode) ? (long long)(*a.d_step) : 0;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long long row = j0 + tj * 16 + it * 4 + lr;
      if (row >= a.M) continue;
      const float rs = st[tj][it];
      const float4 t = r[tj][it];
      float y0 = t.x * rs * g4.x + b4.x, y1 = t.y * rs * g4.y + b4.y, y2 = t.z * rs * g4.z + b4.z, y3 = t.w * rs * g4.w + b4.w;
      if (a.adaln) {
        const long long trow = a.tmode ? (a.trow0 + row / a.L) : trow_all;
        const float* sc = a.ss + trow * a.ss_tstride + col;
        const float4 s4 = *reinterpret_cast<const float4*>(sc), h4 = *reinterpret_cast<const float4*>(sc + CFD_D);
        y0 = silu_f(y0 * s4.x + h4.x); y1 = silu_f(y1 * s4.y + h4.y); y2 = silu_f(y2 * s4.z + h4.z); y3 = silu_f(y3 * s4.w + h4.w);
      }
      sp_store4(a.out + row * (CFD_D * 4), col, y0, y1, y2, y3);
    }
}

static hipError_t launch_rowln(const RowLnArgs& a, hipStream_t st) {
  static unsigned long long attr_set = 0;   // per device, like launch_cfg
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_set >> (dev & 63)) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rowln_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS);
    if (e != hipSuccess) return e;
    attr_set |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL(gemm_rowln_kernel, dim3((unsigned)((a.M + RL_BJ - 1) / RL_BJ)), dim3(512), RL_LDS, st, a);
  return hipGetLastError();
}
