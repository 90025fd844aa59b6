// Fused TimeBlock (cross_attention.py:426-439, applied at :575 and :655):
//
//   x[row][:] += W . silu( LayerNorm(x[row]) * (1 + scale_t) + shift_t ) + bias
//
// Round 1 ran it as ln_rows_kernel (x -> h, 4 KB per token of HBM traffic) + gemm_sp_kernel with the residual epilogue (h in, x
// read-modify-write: 6 KB per token); two per layer = 14 % of a step, both HBM-bound.  Here a workgroup owns 64 COMPLETE rows: it
// reads them once into registers (64 rows x 512 floats = 64 VGPRs per lane over 8 waves), makes the LayerNorm statistics there,
// produces the modulated + activated operand h of each 32-deep k-step in registers and writes it to LDS as split pairs, streams W
// through LDS with the LDS-DMA, and adds the product to the rows it still holds: 4 KB per token, one launch.
//
//   thread t of 512: row r = t >> 3, sub = t & 7; holds x[r][32 kt + 4 sub .. +3] for kt = 0..15   (a wave = 8 rows; every global
//                    access of a wave is 8 rows x 128 contiguous bytes)
//   MFMA (v_mfma_f32_16x16x32, split pairs, 3 per product): D^T[n][row] = sum_k W[n][k] h[row][k]; wave w owns n in [64 w, 64 w + 64)
//                    for all 64 rows = 4 x 4 tiles, 48 MFMAs and 16 fragment reads per k-step
//   LDS: two stages of { W tile 512 x 128 B (64 KB, 8 LDS-DMA pieces per wave) | h tile 64 x 128 B (8 KB) }, source-side / write-side
//                    chunk swizzle (row >> 1) & 7 as in gemm_sp.hpp; the folded per-column vectors A = g (1 + scale), B = b (1 + scale)
//                    + shift and the output bias (6 KB); the epilogue re-lays D through the stage area to the row layout of x.
//   The per-column vectors come from LDS, not from global memory: an ordinary load inside the loop would make hipcc drain the LDS-DMA
//   queue in front of every fragment read.
// One timestep for all rows (the sampling loop, cfd_forward with a scalar timestep); per-row timesteps keep the two-launch form.
#pragma once
#include "cfd_common.hpp"

#define TB_ROWS 64
#define TB_THREADS 512
#define TB_STAGE 73728                 // 64 KB W tile + 8 KB h tile
#define TB_POFF (2 * TB_STAGE)         // A | B | bias, 512 floats each
#define TB_LDS (TB_POFF + 3 * 512 * 4)
#define TB_RS (512 * 4 + 16)           // epilogue row stride (bytes)

struct TbArgs {
  float* x;                // [M][512] residual stream, updated in place
  long long M;
  const float* g;          // norm.weight [512]
  const float* b;          // norm.bias [512]
  const float* ss;         // (1 + scale | shift) of THIS time block: 1024 floats per table row
  long long ss_tstride;    // floats between table rows
  const int* d_step;       // table row = *d_step
  const char* W;           // SP [512][512]: out_layers.2.weight
  const float* bias;       // out_layers.2.bias [512]
};

__global__ void __launch_bounds__(TB_THREADS, 2) tb_fused_kernel(const TbArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(TB_ROWS * TB_RS <= TB_POFF, "the epilogue image must not reach the parameter area");
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r = t >> 3, sub = t & 7;
  const int l15 = lane & 15, q4 = lane >> 4, sw = l15 >> 1;
  const long long row0 = (long long)blockIdx.x * TB_ROWS;
  const long long row = row0 + r;
  const bool row_ok = row < a.M;

  // ---- the rows of this workgroup -> registers ------------------------------------------------------------------------------
  float4 xr[16];
  {
    const float* xp = a.x + (row_ok ? row : a.M - 1) * CFD_D + 4 * sub;
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) xr[kt] = *reinterpret_cast<const float4*>(xp + 32 * kt);
  }
  // ---- per-column vectors -> LDS (folded: y = n A + B with n the normalised value; same arithmetic as ln_rows_kernel up to the
  //      association of the two affine maps) ------------------------------------------------------------------------------------
  if (t < 128) {
    const float* sc = a.ss + (long long)(*a.d_step) * a.ss_tstride + 4 * t;
    const float4 g = *reinterpret_cast<const float4*>(a.g + 4 * t), b = *reinterpret_cast<const float4*>(a.b + 4 * t);
    const float4 s1 = *reinterpret_cast<const float4*>(sc), s2 = *reinterpret_cast<const float4*>(sc + CFD_D);
    const float4 bo = *reinterpret_cast<const float4*>(a.bias + 4 * t);
    float4 A, B;
    A.x = g.x * s1.x; A.y = g.y * s1.y; A.z = g.z * s1.z; A.w = g.w * s1.w;
    B.x = b.x * s1.x + s2.x; B.y = b.y * s1.y + s2.y; B.z = b.z * s1.z + s2.z; B.w = b.w * s1.w + s2.w;
    reinterpret_cast<float4*>(smem + TB_POFF)[t] = A;
    reinterpret_cast<float4*>(smem + TB_POFF + 2048)[t] = B;
    reinterpret_cast<float4*>(smem + TB_POFF + 4096)[t] = bo;
  }
  // ---- LayerNorm statistics: two-pass over the 64 values of the lane, 8 lanes per row (eps 1e-5, biased variance) --------------
  float mean, rstd;
  {
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) s += (xr[kt].x + xr[kt].y) + (xr[kt].z + xr[kt].w);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    mean = s * (1.0f / CFD_D);
    float ss = 0.f;
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      const float d0 = xr[kt].x - mean, d1 = xr[kt].y - mean, d2 = xr[kt].z - mean, d3 = xr[kt].w - mean;
      ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
    rstd = 1.0f / sqrtf(ss * (1.0f / CFD_D) + 1e-5f);
  }
  __syncthreads();   // per-column vectors visible; every global load of the prologue has returned (the barrier drains vmcnt)

  // ---- staging ---------------------------------------------------------------------------------------------------------------
  const int cpos = lane & 7, rsub = lane >> 3;
  const unsigned wsrc_lane = (unsigned)((wid * 8 + rsub) * (CFD_D * 4) + ((cpos ^ (((wid & 1) << 2) | (rsub >> 1))) << 4));
  auto stage_w = [&](int kt, int st) __attribute__((always_inline)) {   // piece n of wave `wid`: W rows (wid + 8 n) * 8 .. +7
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      unsigned wl = wsrc_lane;
      const char* base = a.W + (long long)n * (64 * CFD_D * 4) + kt * 128;
      asm volatile("" : "+v"(wl), "+s"(base));
      __builtin_amdgcn_global_load_lds((gptr_t)(base + wl), (lptr_t)(smem + st * TB_STAGE + (wid + 8 * n) * 1024), 16, 0, 0);
    }
  };
  const int hdst = r * 128 + (sub & 1) * 8;
  const int hsw = (r >> 1) & 7;
  auto make_h = [&](int kt, int st) __attribute__((always_inline)) {    // this lane's 4 columns of k-step kt -> LDS (hi, lo)
    const f32x4 A = *reinterpret_cast<const f32x4*>(smem + TB_POFF + (32 * kt + 4 * sub) * 4);
    const f32x4 B = *reinterpret_cast<const f32x4*>(smem + TB_POFF + 2048 + (32 * kt + 4 * sub) * 4);
    const float v[4] = {xr[kt].x, xr[kt].y, xr[kt].z, xr[kt].w};
    spx4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = ((v[e] - mean) * rstd) * A[e] + B[e];
      const float sl = silu_f(y);
      sp_t hi, lo;
      split_f32(sl, hi, lo);
      h[e] = hi;
      l[e] = lo;
    }
    char* hp = smem + st * TB_STAGE + 65536 + hdst;
    *reinterpret_cast<spx4*>(hp + (((sub >> 1) ^ hsw) << 4)) = h;
    *reinterpret_cast<spx4*>(hp + (((4 + (sub >> 1)) ^ hsw) << 4)) = l;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int woff_h = (wid * 64 + l15) * 128 + ((q4 ^ sw) << 4), woff_l = (wid * 64 + l15) * 128 + (((4 + q4) ^ sw) << 4);
  const int hoff_h = 65536 + l15 * 128 + ((q4 ^ sw) << 4), hoff_l = 65536 + l15 * 128 + (((4 + q4) ^ sw) << 4);

  stage_w(0, 0);
  make_h(0, 0);
  __syncthreads();
#pragma unroll   // (fully unrolled: xr[] must be indexed statically to stay in registers)
  for (int kt = 0; kt < 16; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < 16) {
      stage_w(kt + 1, st ^ 1);
      make_h(kt + 1, st ^ 1);
    }
    const char* sb = smem + st * TB_STAGE;
    spx8 wh[4], wl[4], hh[4], hl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wh[i] = *reinterpret_cast<const spx8*>(sb + woff_h + i * 2048);
      wl[i] = *reinterpret_cast<const spx8*>(sb + woff_l + i * 2048);
      hh[i] = *reinterpret_cast<const spx8*>(sb + hoff_h + i * 2048);
      hl[i] = *reinterpret_cast<const spx8*>(sb + hoff_l + i * 2048);
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        acc[ti][tj] = SP_MFMA(wl[ti], hh[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(wh[ti], hl[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(wh[ti], hh[tj], acc[ti][tj], 0, 0, 0);
      }
    __syncthreads();   // stage st^1 complete (the barrier drains this wave's LDS-DMA and LDS writes), stage st free
  }

  // ---- epilogue: D^T tiles -> [row][n] image in LDS -> the row layout of x; x += D + bias -----------------------------------------
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
      *reinterpret_cast<f32x4*>(smem + (16 * tj + l15) * TB_RS + (wid * 64 + 16 * ti + 4 * q4) * 4) = acc[ti][tj];
  __syncthreads();
  if (row_ok) {
    float* xp = a.x + row * CFD_D + 4 * sub;
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(smem + r * TB_RS + (32 * kt + 4 * sub) * 4);
      const f32x4 bo = *reinterpret_cast<const f32x4*>(smem + TB_POFF + 4096 + (32 * kt + 4 * sub) * 4);
      float4 o;   // same association as the residual epilogue of gemm_sp.hpp: (x + bias) + D
      o.x = (xr[kt].x + bo[0]) + d[0]; o.y = (xr[kt].y + bo[1]) + d[1]; o.z = (xr[kt].z + bo[2]) + d[2]; o.w = (xr[kt].w + bo[3]) + d[3];
      *reinterpret_cast<float4*>(xp + 32 * kt) = o;
    }
  }
}
