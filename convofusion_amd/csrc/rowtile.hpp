// Row-tile path: the denoiser forward for SMALL problems (the product shape: one to a few utterances of 16 latent frames, i.e.
// 7 guidance rows x 16 tokens per utterance; denoiser.py:173-386, cross_attention.py:556-664).
//
// At this size every product of the network is a [16 tokens] x [512] x [512] problem per batch row and the step is bound by
// LATENCY, not by bytes or MFMA rate: round 3's tile kernels (64 x 64 tiles, LDS ring, 16 dependent k-steps, separate LayerNorm and
// softmax launches) ran 171 launches of ~8.4 us per step.  Here the unit of work is a "token tile" = 16 consecutive tokens of ONE
// batch row (exactly one MFMA tile high) and every kernel has the same shape:
//
//   prologue   the workgroup builds the A operand of its tile itself -- LayerNorm / AdaLN + SiLU of the 16 residual rows (two-pass
//              statistics, 16 or 32 lanes per row), or softmax probabilities, written to LDS as split pairs -- or reads it straight
//              from an SP matrix in global memory; LayerNorm and softmax are therefore no launches of their own
//   product    16 tokens x 16 features per workgroup with the K axis split over the waves (2-4 k-groups of 32 each): ALL operand
//              loads of the kernel are issued at its start as direct 16-byte fragment loads (weights: global -> registers in MFMA
//              operand layout), so a kernel pays ONE memory round trip, then 6-12 MFMAs per wave, then an LDS reduction over the waves
//   epilogue   by wave 0: bias, GELU, residual add, split-pair store
//
// so that a 512 x 512 product at one utterance is 7 x 32 = 224 independent workgroups (one per CU) instead of 16 workgroups walking
// 16 k-steps, and a layer is 9 launches (LN1+QKV, self-attention core, out-projection, time block 1, cross-attention scores,
// cross-attention softmax + P.V, time block 2, FFN1, FFN2) instead of 19.  Arithmetic is the split-pair product of cfd_common.hpp
// (3 MFMAs, fp32 accumulate) everywhere; the cross-attention uses the folded, timestep-hoisted form of xattn_fused.hpp:
//   score(q, s) = rs_s (q . KA_s + q . (A b_t)) + cbk_s,   x += sum_s P'_s VA_s + (sum_s P'_s) VV b_t + bias,   P' = p rs
// with rs / cbk tabulated for every step of a run at cfd_sample_begin (mem_scale_table_kernel).
//
// Why launches and not one persistent cooperative kernel: MI355X_MICROARCH.md's price list puts a dependent kernel boundary at
// 1.2 - 1.9 us and an XCD-hierarchical grid barrier at 4.1 - 4.8 us (+ the release / acquire fences of every hand-off), and every
// phase here is an all-to-all seam over the 32 workgroups of a batch row (each needs the complete 512-wide rows the others wrote);
// DESIGN.md section 10 has the measured A/B.
#pragma once
#include "cfd_common.hpp"

// Developer build (-DRT_STAMP=1, tools/rt_stamps.py): workgroup (0, 0) of every row-tile launch appends {kernel id, s_memrealtime at
// entry, at operands-arrived, at exit} to a ring in device memory -- the time line of consecutive launches of a captured step.
#ifndef RT_STAMP
#define RT_STAMP 0
#endif
#if RT_STAMP
__device__ unsigned long long g_rt_ring[4 * 4096];
__device__ unsigned int g_rt_seq;
#define RT_T(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define RT_STAMP_OUT(id, t0, t1)                                                                     \
  do {                                                                                               \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                                    \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
      const unsigned long long t2_ = __builtin_amdgcn_s_memrealtime();                               \
      const unsigned k_ = atomicAdd(&g_rt_seq, 1u) & 4095u;                                          \
      g_rt_ring[4 * k_] = (id); g_rt_ring[4 * k_ + 1] = (t0); g_rt_ring[4 * k_ + 2] = (t1); g_rt_ring[4 * k_ + 3] = t2_; \
    }                                                                                                \
  } while (0)
#else
#define RT_T(var) do { } while (0)
#define RT_STAMP_OUT(id, t0, t1) do { } while (0)
#endif

#define RT_MAX_L 32          // self-attention keys of a batch row fit one 32-deep k-step
#define RT_ARG_ROWS 64       // batch rows whose memory instances travel inside the kernel arguments
#define RT_MAX_KEYS 1024     // padded cross-attention keys of all five memories together

template <class T>
__device__ __forceinline__ T rt_sel(const T (&arr)[CFD_NMEM], int j) {
  T v = arr[0];
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if (j == q) v = arr[q];
  return v;
}

// one split-pair product step: acc += x . y over a 32-deep k-group (x: rows -> D rows, 4 per lane; y: rows -> D columns)
__device__ __forceinline__ f32x4 rt_mma(const spx8 xh, const spx8 xl, const spx8 yh, const spx8 yl, f32x4 acc) {
  acc = SP_MFMA(xl, yh, acc, 0, 0, 0);
  acc = SP_MFMA(xh, yl, acc, 0, 0, 0);
  acc = SP_MFMA(xh, yh, acc, 0, 0, 0);
  return acc;
}


// LDS image of a 16-row SP operand: [k-group][row 16][128 B], 16-byte chunk c of a row stored at position c ^ ((row >> 1) & 7)
// (the swizzle of gemm_sp.hpp: conflict-free ds_read_b128 fragment reads)
__device__ __forceinline__ void rt_lfrag(const char* img, int kt, int l15, int q4, spx8& hi, spx8& lo) {
  const int sw = (l15 >> 1) & 7;
  const char* p = img + kt * 2048 + l15 * 128;
  hi = *reinterpret_cast<const spx8*>(p + ((q4 ^ sw) << 4));
  lo = *reinterpret_cast<const spx8*>(p + (((4 + q4) ^ sw) << 4));
}
// LDS-DMA of k-group kt of 16 SP rows (row r at base + min(r, rmax) * ld bytes) into the image slot `slot` (2 KB: [row 16][128 B]):
// two wave-instructions of 8 rows x 128 B, i.e. whole cache lines per instruction (a direct 16-byte fragment load touches
// sixteen HALF lines per instruction and runs at a quarter of the L1's rate: measured 1.5 us to issue 64 KB, tools/rt_stamps.py);
// the swizzle is applied on the source address, as in gemm_sp.hpp.  The caller waits (vmcnt) before reading the slot.
__device__ __forceinline__ void rt_dma_slice(char* slot, const char* base, long long ld, int rmax, int kt, int lane) {
  const int cpos = lane & 7, rsub = lane >> 3;
#pragma unroll
  for (int pc = 0; pc < 2; ++pc) {
    const int r = pc * 8 + rsub;
    const char* src = base + (long long)min(r, rmax) * ld + (long long)kt * 128 + ((cpos ^ ((r >> 1) & 7)) << 4);
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(slot + pc * 1024), 16, 0, 0);
  }
}
#define RT_WAIT_VM0() do { __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)   /* vmcnt(0), lgkmcnt / expcnt untouched */


// Reductions over the LPR lanes that share a prologue row (LPR = 16: one DPP row; 32: two adjacent DPP rows), in the vector ALU:
// quad permutes, row_half_mirror and row_mirror (each lane ends with the value of its whole 16-lane row), then
// v_permlane16_swap for the neighbouring row -- five dependent ds_bpermute round trips per reduction otherwise (__shfl_xor).
template <int CTRL>
__device__ __forceinline__ float rt_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LPR>
__device__ __forceinline__ float rt_row_sum(float v) {
  static_assert(LPR == 16 || LPR == 32, "16 or 32 lanes per row");
  v += rt_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
  v += rt_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
  v += rt_dpp<0x141>(v);    // row_half_mirror
  v += rt_dpp<0x140>(v);    // row_mirror
  if constexpr (LPR == 32) {
    const unsigned xi = __float_as_uint(v);
    auto q = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    v = __uint_as_float(q[0]) + __uint_as_float(q[1]);
  }
  return v;
}
template <int LPR>
__device__ __forceinline__ float rt_row_max(float v) {
  static_assert(LPR == 16 || LPR == 32, "16 or 32 lanes per row");
  v = fmaxf(v, rt_dpp<0xB1>(v));
  v = fmaxf(v, rt_dpp<0x4E>(v));
  v = fmaxf(v, rt_dpp<0x141>(v));
  v = fmaxf(v, rt_dpp<0x140>(v));
  if constexpr (LPR == 32) {
    const unsigned xi = __float_as_uint(v);
    auto q = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
    v = fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
  }
  return v;
}
// 1 / x to float32 rounding noise: v_rcp_f32 (1 ulp) + one Newton step, instead of the ~10-instruction IEEE division sequence
__device__ __forceinline__ float rt_rcp(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}
// sigmoid(x) = 1 / (1 + exp(-x)).  The exponent is clamped at 80: for x < -88.7 exp(-x) is +inf in float32 and the Newton step of rt_rcp
// turns 1 / inf into inf * 0 = NaN -- which split_f32's v_med3 then "clamps" to -65504 (found by the heavy-tailed stress weights:
// tests/golden/make_golden_heavy.py; AdaLN inputs below -89 do not occur with the uniform test weights).  exp(80) = 5.5e34 keeps every
// intermediate a normal number; the result differs from the exact one by less than 2e-35 |x|.
__device__ __forceinline__ float rt_sigmoid(float x) { return rt_rcp(1.0f + __expf(fminf(-x, 80.0f))); }
__device__ __forceinline__ float rt_silu(float x) { return x * rt_sigmoid(x); }

// store 4 consecutive columns (c0 % 4 == 0) of row r into the image
__device__ __forceinline__ void rt_lstore4(char* img, int r, int c0, const float* v) {
  spx4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sp_t a, b;
    split_f32(v[e], a, b);
    h[e] = (v[e] != v[e]) ? (sp_t)v[e] : a;   // a NaN stays a NaN (split_f32's v_med3 clamp would make it -65504): see ln_rows_kernel
    l[e] = b;
  }
  const int sw = (r >> 1) & 7, ch = (c0 & 31) >> 3, half = ((c0 & 31) >> 2) & 1;
  char* p = img + (c0 >> 5) * 2048 + r * 128 + half * 8;
  *reinterpret_cast<spx4*>(p + ((ch ^ sw) << 4)) = h;
  *reinterpret_cast<spx4*>(p + (((4 + ch) ^ sw) << 4)) = l;
}
// fragment of row `row` of a general image: [k-group][nrows][128 B] with the same swizzle (row index inside the image)
__device__ __forceinline__ void rt_lfrag_row(const char* img, int kt, int nrows, int row, int q4, spx8& hi, spx8& lo) {
  const int sw = (row >> 1) & 7;
  const char* p = img + (kt * nrows + row) * 128;
  hi = *reinterpret_cast<const spx8*>(p + ((q4 ^ sw) << 4));
  lo = *reinterpret_cast<const spx8*>(p + (((4 + q4) ^ sw) << 4));
}
// one LDS-DMA piece: 8 rows x 128 B of k-group kt, rows r0 .. r0 + 7 of the image (source row = min(src_row0 + r, rmax))
__device__ __forceinline__ void rt_dma_piece(char* img_rows0, const char* base, long long ld, int src_row0, int rmax, int r0, int kt, int lane) {
  const int cpos = lane & 7, r = r0 + (lane >> 3);
  const char* src = base + (long long)min(src_row0 + r, rmax) * ld + (long long)kt * 128 + ((cpos ^ ((r >> 1) & 7)) << 4);
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(img_rows0 + r0 * 128), 16, 0, 0);
}

// Sum of the waves' partial accumulators of NFB feature blocks (fixed order: bit-reproducible); block i's sum is returned in wave i.
// Every wave parks its partials in ITS OWN staging region (dead once its fragments are in registers), `stride` bytes apart.
template <int NW, int NFB>
__device__ __forceinline__ f32x4 rt_reduce(char* stage0, int stride, int wid, int lane, const f32x4 (&acc)[NFB]) {
#pragma unroll
  for (int i = 0; i < NFB; ++i) reinterpret_cast<f32x4*>(stage0 + wid * stride + i * 1024)[lane] = acc[i];
  __syncthreads();
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wid < NFB) {
    s = reinterpret_cast<const f32x4*>(stage0 + wid * 1024)[lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const f32x4 t = reinterpret_cast<const f32x4*>(stage0 + w * stride + wid * 1024)[lane];
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
  }
  return s;
}

// LayerNorm prologue of a 512-thread workgroup: thread (row pr = tid / 32, lane plr = tid % 32) owns columns 128 i + 4 plr .. + 3,
// i = 0..3 -- 16-byte loads that are contiguous across the row's lanes.  The per-column parameters (LayerNorm weight / bias, and
// up to two more rows) go through LDS: ONE float4 per thread from global memory instead of every row's lanes loading all of them
// (16 x the bytes through the L1, which is what bounds these kernels: tools/rt_stamps.py).
#define RT_LPR 32
__device__ __forceinline__ void rt_rows_load(const float* x, long long row, int plr, float (&v)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 q = *reinterpret_cast<const float4*>(x + row * CFD_D + 128 * i + 4 * plr);
    v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
  }
}
// parameter rows p0..p3 (512 floats each, null = unused) -> par[4][128] float4 in LDS; NPAR * 128 threads load one float4 each
template <int NPAR>
__device__ __forceinline__ float4 rt_par_fetch(const float* p0, const float* p1, const float* p2, const float* p3) {
  const int arr = threadIdx.x >> 7, idx = threadIdx.x & 127;
  const float* src = arr == 0 ? p0 : arr == 1 ? p1 : arr == 2 ? p2 : p3;
  return (arr < NPAR) ? *reinterpret_cast<const float4*>(src + 4 * idx) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void rt_par_get(const char* par, int arr, int i, int plr, float (&o)[4]) {
  const float4 q = reinterpret_cast<const float4*>(par)[arr * 128 + 32 * i + plr];
  o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
}
// statistics of the row: v is centred in place, returns 1 / sqrt(var + eps)
__device__ __forceinline__ float rt_ln_stats(float (&v)[4][4]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += v[i][e];
  const float mean = rt_row_sum<RT_LPR>(s) * (1.0f / CFD_D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[i][e] -= mean; ss += v[i][e] * v[i][e]; }
  return 1.0f / sqrtf(rt_row_sum<RT_LPR>(ss) * (1.0f / CFD_D) + 1e-5f);
}

// ------------------------------------------------------------------------------------------------
// Generic token-tile product:  out[16 tokens][NFB x 16 features] = A[16][K] . W[f0 ..][K]^T
// ------------------------------------------------------------------------------------------------
enum { RT_PRO_SP = 0, RT_PRO_LN = 1, RT_PRO_ADALN = 2 };
enum { RT_EPI_RESID = 0, RT_EPI_SPLIT = 1, RT_EPI_F32 = 2, RT_EPI_EMBED = 3, RT_EPI_QKV = 4 };

struct RtGemmArgs {
  int L, tpr;              // tokens per batch row, token tiles per batch row
  // A operand
  const float* x;          // fp32 [M][512] residual stream: source of the LN / AdaLN prologues
  const char* a_sp;        // SP [M][K] (RT_PRO_SP)
  const float* g;          // LayerNorm weight / bias [512]
  const float* b;
  const float* ss;         // AdaLN: (1 + scale | shift) of this time block AT THIS STEP [1024] (no step index on the device side: a
                           // dependent scalar load in front of the parameter fetch costs every launch ~1 us, tools/rt_stamps.py)
  // W operand
  const char* w;           // SP [N][K]
  const char* w2;          // RT_EPI_QKV: the value projection's weights (16-feature blocks >= nfb_qk, operand roles swapped)
  int nfb_qk;
  const float* bias;       // [N] or null
  // outputs
  const float* xr;         // RESID: the residual rows (fp32 [M][512]) ...
  float* xo;               // ... and where the sum goes (RESID / EMBED); xo == xr: in place
  char* o_sp;              // SPLIT / QKV: SP matrix, ld_o bytes per token row
  long long ld_o;
  int gelu;
  float* pre;              // SPLIT: optional copy of the pre-activation values, fp32 [M][ld_o / 4] (kept for the WEG backward)
  float* o_f32;            // F32: [M][ldo_f]
  int ldo_f;
  char* vt;                // QKV: V^T, SP [Be][512][32 keys]
  const float* bh;         // EMBED: body/hand embedding [2][512], query PE [>= L/2][512]
  const float* qpe;
};

// dynamic LDS of rt_gemm_kernel
constexpr int rt_gemm_lds(int pro, int nt, int kt, int nfb) {
  return (pro == RT_PRO_SP ? 0 : 16 * 2048 + 8192) + (nt / 64) * ((pro == RT_PRO_SP ? 1 : 0) + nfb) * (kt / (nt / 64)) * 2048;
}

template <int PRO, int EPI, int NT, int KT, int NFB>
__global__ void __launch_bounds__(NT) rt_gemm_kernel(const RtGemmArgs a) {
  constexpr int NW = NT / 64;
  constexpr int NK = KT / NW;                // k-groups per wave: kt = wid + NW * n
  static_assert(KT % NW == 0 && NK >= 1, "the k-groups divide evenly over the waves");
  static_assert(PRO == RT_PRO_SP || (KT == CFD_D / 32 && NT == 512), "LayerNorm prologues: 512 columns, 512 threads");
  static_assert(NFB <= NW, "wave i finishes feature block i");
  RT_T(t_in);
  // LDS map.  LN prologues: A image (16 k-groups x 2 KB, written by every wave) | parameter rows (8 KB) | staging.  Staging = per
  // wave a PRIVATE region [A slices NK (SP prologue only) | W slices NFB x NK] x 2 KB, filled by that wave's own LDS-DMA and read
  // back by it alone; afterwards the wave parks its partial sums there.
  constexpr int NA = PRO == RT_PRO_SP ? NK : 0;
  constexpr int STAGE_W = (NA + NFB * NK) * 2048;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* img = smem;
  char* par = smem + 16 * 2048;
  char* stage0 = smem + (PRO == RT_PRO_SP ? 0 : 16 * 2048 + 8192);
  char* wreg = stage0 + wid * STAGE_W;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tile = blockIdx.y;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  const int fb0 = blockIdx.x * NFB;                              // first 16-feature block of this workgroup
  auto blk_swapped = [&](int i) __attribute__((always_inline)) { return EPI == RT_EPI_QKV && fb0 + i >= a.nfb_qk; };   // workgroup-uniform
  auto blk_f0 = [&](int i) __attribute__((always_inline)) { return (blk_swapped(i) ? fb0 + i - a.nfb_qk : fb0 + i) * 16; };

  // ---- 1. every global load of the kernel is issued here ------------------------------------------------
  // prologue rows first (vmcnt retires in order: the LayerNorm can start while the weight slices are still in flight)
  float v[4][4];
  float4 parv = make_float4(0.f, 0.f, 0.f, 0.f);
  const int pr = threadIdx.x >> 5, plr = threadIdx.x & 31;      // prologue: row, lane in row
  if constexpr (PRO != RT_PRO_SP) {
    rt_rows_load(a.x, tok0 + min(pr, nq - 1), plr, v);
  } else {
#pragma unroll
    for (int n = 0; n < NK; ++n) rt_dma_slice(wreg + n * 2048, a.a_sp + (size_t)tok0 * (KT * 128), KT * 128, nq - 1, wid + NW * n, lane);
  }
#pragma unroll
  for (int i = 0; i < NFB; ++i) {
    const char* wb = (blk_swapped(i) ? a.w2 : a.w) + (size_t)blk_f0(i) * (KT * 128);
#pragma unroll
    for (int n = 0; n < NK; ++n) rt_dma_slice(wreg + (NA + i * NK + n) * 2048, wb, KT * 128, 15, wid + NW * n, lane);
  }
  if constexpr (PRO == RT_PRO_ADALN) {
    parv = rt_par_fetch<4>(a.g, a.b, a.ss, a.ss + CFD_D);
  } else if constexpr (PRO == RT_PRO_LN) {
    parv = rt_par_fetch<2>(a.g, a.b, nullptr, nullptr);
  }
  // epilogue operands: wave i finishes feature block i
  float4 ep_r = make_float4(0.f, 0.f, 0.f, 0.f), ep_t = ep_r, ep_t1 = ep_r, ep_t2 = ep_r;
  const bool my_swapped = blk_swapped(wid);                       // (wave-uniform)
  const int fcol = blk_f0(wid) + 4 * q4;                          // standard roles: this lane's 4 features, token l15
  if (wid < NFB) {
    if constexpr (EPI == RT_EPI_RESID) ep_r = *reinterpret_cast<const float4*>(a.xr + (tok0 + min(l15, nq - 1)) * CFD_D + fcol);
    if (a.bias && !my_swapped) ep_t = *reinterpret_cast<const float4*>(a.bias + fcol);
    if constexpr (EPI == RT_EPI_EMBED) {
      const int l = q0 + min(l15, nq - 1);
      ep_t1 = *reinterpret_cast<const float4*>(a.bh + (l & 1) * CFD_D + fcol);
      ep_t2 = *reinterpret_cast<const float4*>(a.qpe + (size_t)(l >> 1) * CFD_D + fcol);
    }
  }
  RT_T(t_iss);

  // ---- 2. prologue: LayerNorm (+ AdaLN, SiLU) of the tile's rows -> split-pair image in LDS ---------------
  if constexpr (PRO != RT_PRO_SP) {
    reinterpret_cast<float4*>(par)[threadIdx.x] = parv;
    const float rstd = rt_ln_stats(v);
    __syncthreads();                                            // the parameter rows are in LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float gg[4], bb[4];
      rt_par_get(par, 0, i, plr, gg);
      rt_par_get(par, 1, i, plr, bb);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] * rstd * gg[e] + bb[e];
      if constexpr (PRO == RT_PRO_ADALN) {
        float sv[4], hv[4];
        rt_par_get(par, 2, i, plr, sv);
        rt_par_get(par, 3, i, plr, hv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = rt_silu(v[i][e] * sv[e] + hv[e]);
      }
      rt_lstore4(img, pr, 128 * i + 4 * plr, v[i]);
    }
  }
  RT_WAIT_VM0();                                                  // this wave's slices have landed
  if constexpr (PRO != RT_PRO_SP) __syncthreads();               // the A image is complete
  spx8 wh[NFB][NK], wl[NFB][NK], ah[NK], al[NK];
#pragma unroll
  for (int n = 0; n < NK; ++n) {
    if constexpr (PRO == RT_PRO_SP) rt_lfrag(wreg, n, l15, q4, ah[n], al[n]);
    else rt_lfrag(img, wid + NW * n, l15, q4, ah[n], al[n]);
  }
#pragma unroll
  for (int i = 0; i < NFB; ++i)
#pragma unroll
    for (int n = 0; n < NK; ++n) rt_lfrag(wreg + (NA + i * NK) * 2048, n, l15, q4, wh[i][n], wl[i][n]);

  // ---- 3. product over this wave's k-groups, then the sum over the waves -----------------------------------
#if RT_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  RT_T(t_ops);
  f32x4 part[NFB];
#pragma unroll
  for (int i = 0; i < NFB; ++i) {
    part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool sw_i = blk_swapped(i);
#pragma unroll
    for (int n = 0; n < NK; ++n) part[i] = sw_i ? rt_mma(ah[n], al[n], wh[i][n], wl[i][n], part[i]) : rt_mma(wh[i][n], wl[i][n], ah[n], al[n], part[i]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's fragment reads are done: its staging region is free
  const f32x4 acc = rt_reduce<NW, NFB>(stage0, STAGE_W, wid, lane, part);
  if (wid >= NFB) return;
#if RT_STAMP
  struct StampAtExit { unsigned long long a, b; int id; __device__ ~StampAtExit() { RT_STAMP_OUT(id, a, b); } } stamp_{t_in, t_ops, (int)(100 * PRO + 10 * EPI + (KT == 32 ? 1 : 0)) | (int)((t_iss - t_in) << 16)};
#endif

  // ---- 4. epilogue (wave i: feature block i).  Standard roles: lane (token l15) holds features fcol .. fcol + 3 ---------------
  if constexpr (EPI == RT_EPI_QKV) {
    if (my_swapped) {   // lane (feature f0 + l15) holds tokens q0 + 4 q4 .. + 3 of V^T; tokens beyond L are stored as zero
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (4 * q4 + r < nq) ? acc[r] : 0.f;
      sp_store4(a.vt + ((size_t)b * CFD_D + blk_f0(wid) + l15) * (RT_MAX_L * 4), q0 + 4 * q4, o[0], o[1], o[2], o[3]);
      return;
    }
  }
  if (l15 >= nq) return;
  const long long tok = tok0 + l15;
  if constexpr (EPI == RT_EPI_RESID) {
    float4 r = ep_r;   // same association as EpiResid: (x + bias) + product
    r.x = (r.x + ep_t.x) + acc[0]; r.y = (r.y + ep_t.y) + acc[1]; r.z = (r.z + ep_t.z) + acc[2]; r.w = (r.w + ep_t.w) + acc[3];
    *reinterpret_cast<float4*>(a.xo + tok * CFD_D + fcol) = r;
  } else if constexpr (EPI == RT_EPI_SPLIT || EPI == RT_EPI_QKV) {
    float o[4] = {acc[0] + ep_t.x, acc[1] + ep_t.y, acc[2] + ep_t.z, acc[3] + ep_t.w};
    if constexpr (EPI == RT_EPI_SPLIT) {
      if (a.pre) *reinterpret_cast<float4*>(a.pre + tok * (a.ld_o / 4) + fcol) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (a.gelu) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = gelu_fast_f(o[r]);
    }
    sp_store4(a.o_sp + tok * a.ld_o, fcol, o[0], o[1], o[2], o[3]);
  } else if constexpr (EPI == RT_EPI_F32) {
    *reinterpret_cast<float4*>(a.o_f32 + tok * a.ldo_f + fcol) = make_float4(acc[0] + ep_t.x, acc[1] + ep_t.y, acc[2] + ep_t.z, acc[3] + ep_t.w);
  } else if constexpr (EPI == RT_EPI_EMBED) {   // ((linear + bh) + pe), the reference's association (EpiEmbed)
    float4 r;
    r.x = ((acc[0] + ep_t.x) + ep_t1.x) + ep_t2.x;
    r.y = ((acc[1] + ep_t.y) + ep_t1.y) + ep_t2.y;
    r.z = ((acc[2] + ep_t.z) + ep_t1.z) + ep_t2.z;
    r.w = ((acc[3] + ep_t.w) + ep_t1.w) + ep_t2.w;
    *reinterpret_cast<float4*>(a.xo + tok * CFD_D + fcol) = r;
  }
}

// ------------------------------------------------------------------------------------------------
// Self-attention core of one (head, token tile): o = softmax(q k^T) v for L <= 32 keys (cross_attention.py:568-572; q is
// pre-scaled in the weights).  The head's q tile (8 KB), the row's keys (16 KB) and V^T (16 KB) come in by LDS-DMA, whole cache lines
// per instruction; every wave then forms the 16 x 32 scores itself (S^T[key][query] = K . Q, 24 MFMAs) and takes 2 of the head's 8
// feature tiles of the P.V product.  The key rows are assigned to MFMA rows so that a lane's 8 score registers are keys
// 8 g .. 8 g + 7: exactly the k-slots it supplies as the second operand of P.V (as in attn_fused.hpp).
// dynamic LDS: 40 KB
// ------------------------------------------------------------------------------------------------
struct RtSelfArgs {
  const char* qk;    // SP [M][1024]: q at k-groups 4 h .. 4 h + 3, k at 16 + 4 h ..
  const char* vt;    // SP [Be][512][32 keys]
  char* o;           // SP [M][512]
  int L, tpr;
};

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) rt_selfattn_kernel(const RtSelfArgs a) {
  RT_T(t_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qimg = smem;                 // [4 k-groups][16 rows][128 B]
  char* kimg = smem + 8192;          // [4 k-groups][32 rows][128 B]
  char* vimg = smem + 8192 + 16384;  // [128 feature rows][128 B]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int h = blockIdx.x, tile = blockIdx.y;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L;
  const char* rows0 = a.qk + (size_t)tok0 * 4096;
  // 40 pieces of 8 rows x 128 B, 10 per wave: wave w takes k-group w of q (2) and of k (4), and feature rows 32 w .. of V^T (4)
#pragma unroll
  for (int pc = 0; pc < 2; ++pc) rt_dma_piece(qimg + wid * 2048, rows0, 4096, q0, a.L - 1, 8 * pc, 4 * h + wid, lane);
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) rt_dma_piece(kimg + wid * 4096, rows0, 4096, 0, a.L - 1, 8 * pc, 16 + 4 * h + wid, lane);
  {
    const char* vb = a.vt + ((size_t)b * CFD_D + h * CFD_HD) * (RT_MAX_L * 4);
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) rt_dma_piece(vimg, vb, RT_MAX_L * 4, 0, CFD_HD - 1, 32 * wid + 8 * pc, 0, lane);
  }
  RT_WAIT_VM0();
  __syncthreads();
  const int key0 = 8 * (l15 >> 2) + (l15 & 3), key1 = key0 + 4;       // MFMA row l15 of key tile 0 / 1
  spx8 qh[4], ql[4], k0h[4], k0l[4], k1h[4], k1l[4], vh[2], vl[2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    rt_lfrag_row(qimg, g, 16, l15, q4, qh[g], ql[g]);
    rt_lfrag_row(kimg, g, 32, key0, q4, k0h[g], k0l[g]);
    rt_lfrag_row(kimg, g, 32, key1, q4, k1h[g], k1l[g]);
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) rt_lfrag_row(vimg, 0, CFD_HD, (2 * wid + n) * 16 + l15, q4, vh[n], vl[n]);
#if RT_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  RT_T(t_ops);
  f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    s0 = rt_mma(k0h[g], k0l[g], qh[g], ql[g], s0);
    s1 = rt_mma(k1h[g], k1l[g], qh[g], ql[g], s1);
  }
  // lane (query l15, g = q4): s0[r] = key 8 g + r, s1[r] = key 8 g + 4 + r
  float p[8];
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p[r] = (8 * q4 + r < a.L) ? s0[r] : -INFINITY;
    p[4 + r] = (8 * q4 + 4 + r < a.L) ? s1[r] : -INFINITY;
    mx = fmaxf(mx, fmaxf(p[r], p[4 + r]));
  }
  mx = xlane_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { p[e] = __expf(p[e] - mx); sum += p[e]; }
  sum = xlane_sum(sum);
  const float inv = rt_rcp(sum);
  spx8 ph, pl;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sp_t hi, lo;
    split_f32(p[e] * inv, hi, lo);
    ph[e] = hi;
    pl[e] = lo;
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    f32x4 o = rt_mma(vh[n], vl[n], ph, pl, f32x4{0.f, 0.f, 0.f, 0.f});   // O^T[feature][query]
    if (l15 < nq)
      sp_store4(a.o + (size_t)(tok0 + q0 + l15) * (CFD_D * 4), h * CFD_HD + (2 * wid + n) * 16 + 4 * q4, o[0], o[1], o[2], o[3]);
  }
  RT_STAMP_OUT(1000, t_in, t_ops);
}

// ------------------------------------------------------------------------------------------------
// Cross-attention, first half: LayerNorm2 of the tile's rows and the scores against one CELL of 32 folded keys of one memory
// (cross_attention.py:578-652, folded + timestep-hoisted form, see the header):
//   score_s = rs_s (q . KA_s + q . (A b_t)) + cbk_s;   out: e_s = exp(score_s - max over the cell), the cell's (max, sum e, sum e rs)
// dynamic LDS: A image 32 KB | parameters 8 KB | 8 x 8 KB staging | c_q, exchange
// ------------------------------------------------------------------------------------------------
// Per-memory scalars of a kernel's arguments, chosen per lane.  Written as a select over the argument ARRAY (rt_sel) the compiler turns
// `select(j == q, load arr[q], ...)` into ONE load at a per-lane address of the kernel-argument segment -- a vector-memory round trip in
// the middle of a prologue, behind a barrier.  Values pinned in scalar registers first stay selects.
#define RT_PIN_S(x) asm volatile("" : "+s"(x))
template <class T>
__device__ __forceinline__ T rt_pick5(int j, T v0, T v1, T v2, T v3, T v4) {
  return j == 0 ? v0 : j == 1 ? v1 : j == 2 ? v2 : j == 3 ? v3 : v4;
}

// Entry b of a memory's instance table in the kernel-argument segment (RtXArgs::inst[j], RtXBwdArgs::inst[j]), b wave-uniform: fetched
// as the aligned dword that holds it, which is a SCALAR load (there is no scalar byte load on gfx9: indexed as bytes the table was read
// with a vector load per memory, and every operand load whose address depends on the instance waited a vector-memory round trip for it).
__device__ __forceinline__ int rt_inst_get(const unsigned char (&tab)[64], int b) {
  const unsigned w = reinterpret_cast<const unsigned*>(&tab[0])[b >> 2];
  return (int)((w >> (8 * (b & 3))) & 255u);
}

struct RtXArgs {
  const float* x;               // fp32 [M][512]: the residual stream in front of the cross-attention block
  float* xo;                    // where x + block(x) goes (xo == x: in place)
  const float* ln_g;            // norm2
  const float* ln_b;
  const float* bias;            // folded cross-attention bias [512]
  int L, tpr, nl, layer;
  const char* K[CFD_NMEM];      // this layer's folded keys: SP [U_j * Sp_j][512]
  const char* VT[CFD_NMEM];     // this layer's folded values^T: SP [U_j][512][Sp_j]
  const float* cbt[CFD_NMEM];   // THIS STEP's key tables [nl + 1][U_j * Sp_j]: plane l = cbk of layer l, plane nl = rs
  const float* kb[CFD_NMEM];    // A_l b_t [512] of this layer at this step
  const float* vb[CFD_NMEM];    // VV_l b_t [512] of this layer at this step
  const int* map[CFD_NMEM];     // batch row -> memory instance (used when the batch has more than RT_ARG_ROWS rows)
  int use_inst;                 // 1: inst[j][b] below holds map[j][b] (a kernel-argument read instead of a dependent global load in
  alignas(4) unsigned char inst[CFD_NMEM][RT_ARG_ROWS];   // front of the operand fetch; instances < 256, <= RT_ARG_ROWS batch rows; read with rt_inst_get)
  int rows[CFD_NMEM];           // U_j * Sp_j
  int S[CFD_NMEM], Sp[CFD_NMEM], off[CFD_NMEM];
  int blk0[CFD_NMEM + 1];       // (unused since the score launch works on 32-key cells of the concatenated key axis)
  int Sp_tot;
  float* sc;                    // fp32 [M][Sp_tot]: e_s = exp(score_s - its cell's maximum)
  float* cst;                   // float4 [M][Sp_tot / 32]: per (token, cell) maximum, sum e_s, sum e_s rs_s
  float* rsp;                   // fp32 [M][Sp_tot]: rs of every key, repeated per token by the score launch
  float* att[CFD_NMEM];         // optional att_mats [Be][nl][L][S_j]; with att_step: a ring of such blocks for the rows [att_b0, att_b0 + att_nb)
  int att_b0, att_nb;           // batch rows that write their maps (as rows 0 .. att_nb - 1 of the block)
  const int* att_step;          // ring slot = *att_step (the sampling run's iteration counter), or null: one block
  long long att_slot[CFD_NMEM]; // floats per slot
};

// The softmax of a memory is assembled from CELLS of 32 keys: the score launch leaves e_s = exp(score_s - m_c) and, per (token, cell),
// (m_c, l_c = sum e_s, sum e_s rs_s); the probability of key s is then e_s * factor, factor = exp(m_c - M) / sum_c' l_c' exp(m_c' - M)
// over the cells [cb, ce) of the key's memory, M their largest m -- no reduction over lanes in the launch that consumes it
// (its 32 feature-block workgroups per tile each computed the whole five-memory softmax before: 3 of that launch's 7.6 us).
// (M and 1 / sum per (token, memory) are made once per workgroup by 80 threads -- rt_xpv_kernel, rt_xbwd_dy_kernel -- and a key's factor
//  is then exp(m_c - M) / sum.)

#define RT_XS_LDS (16 * 2048 + 8192 + 8 * 8192 + 64 + 512)
template <int CFD_KI = 0>
__global__ void __launch_bounds__(512) rt_xscore_kernel(const RtXArgs a) {
  constexpr int NW = 8, NK = 2, STAGE_W = 2 * NK * 2048;     // per wave: 2 key tiles x NK k-groups
  RT_T(t_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img = smem;
  char* par = smem + 16 * 2048;
  char* stage0 = smem + 16 * 2048 + 8192;
  float* cq = reinterpret_cast<float*>(stage0 + NW * STAGE_W);
  float* xch = cq + 16;                                        // [2 tiles][16 tokens] x (max, l, l_rs) exchange
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* wreg = stage0 + wid * STAGE_W;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tile = blockIdx.y, cell = blockIdx.x;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  const int pr = threadIdx.x >> 5, plr = threadIdx.x & 31;
  float v[4][4];
  rt_rows_load(a.x, tok0 + min(pr, nq - 1), plr, v);   // (first: the rows depend on nothing but the tile)
  int j = 0;
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if (cell * 32 >= a.off[q]) j = q;
  const int s0 = cell * 32 - rt_sel(a.off, j);
  const int u = a.use_inst ? rt_inst_get(a.inst[j], b) : rt_sel(a.map, j)[b];
  const int rows = rt_sel(a.rows, j), Sp = rt_sel(a.Sp, j);
  const long long key0 = (long long)u * Sp + s0;
  const float4 parv = rt_par_fetch<3>(a.ln_g, a.ln_b, rt_sel(a.kb, j), nullptr);
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int n = 0; n < NK; ++n)
      rt_dma_slice(wreg + (t2 * NK + n) * 2048, rt_sel(a.K, j) + (size_t)(key0 + 16 * t2) * (CFD_D * 4), CFD_D * 4, 15, wid + NW * n, lane);
  const float* tab = rt_sel(a.cbt, j);
  float4 e_rs = make_float4(0.f, 0.f, 0.f, 0.f), e_cb = e_rs;
  if (wid < 2) {                                        // wave t finishes key tile t
    e_rs = *reinterpret_cast<const float4*>(tab + (long long)a.nl * rows + key0 + 16 * wid + 4 * q4);
    e_cb = *reinterpret_cast<const float4*>(tab + (long long)a.layer * rows + key0 + 16 * wid + 4 * q4);
  }
  // LayerNorm2 -> image; c_q = q . (A b_t)
  reinterpret_cast<float4*>(par)[threadIdx.x] = parv;
  const float rstd = rt_ln_stats(v);
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float gg[4], bb[4], kk[4];
    rt_par_get(par, 0, i, plr, gg);
    rt_par_get(par, 1, i, plr, bb);
    rt_par_get(par, 2, i, plr, kk);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = v[i][e] * rstd * gg[e] + bb[e];
      dot += v[i][e] * kk[e];
    }
    rt_lstore4(img, pr, 128 * i + 4 * plr, v[i]);
  }
  dot = rt_row_sum<RT_LPR>(dot);
  if (plr == 0) cq[pr] = dot;
  RT_WAIT_VM0();
  __syncthreads();
  spx8 kh[2][NK], kl[2][NK], ah[NK], al[NK];
#pragma unroll
  for (int n = 0; n < NK; ++n) {
    rt_lfrag(img, wid + NW * n, l15, q4, ah[n], al[n]);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) rt_lfrag(wreg + t2 * NK * 2048, n, l15, q4, kh[t2][n], kl[t2][n]);
  }
  RT_T(t_ops);
  f32x4 part[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int n = 0; n < NK; ++n) part[t2] = rt_mma(kh[t2][n], kl[t2][n], ah[n], al[n], part[t2]);   // S^T[key][token]
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const f32x4 acc = rt_reduce<NW, 2>(stage0, STAGE_W, wid, lane, part);
#if RT_STAMP
  struct StampAtExit { unsigned long long a, b; __device__ ~StampAtExit() { RT_STAMP_OUT(2000, a, b); } } stamp_{t_in, t_ops};
#endif
  // waves 0 / 1: the scores of their 16 keys, the cell's maximum (both tiles), e = exp(score - max) and the cell sums
  float sc4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  const float rs4[4] = {e_rs.x, e_rs.y, e_rs.z, e_rs.w};
  if (wid < 2) {
    const float c_q = cq[l15];
    const float cb4[4] = {e_cb.x, e_cb.y, e_cb.z, e_cb.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) sc4[r] = rs4[r] * (acc[r] + c_q) + cb4[r];
    const float m16 = xlane_max(fmaxf(fmaxf(sc4[0], sc4[1]), fmaxf(sc4[2], sc4[3])));
    if (q4 == 0) xch[wid * 16 + l15] = m16;
  }
  __syncthreads();
  float e4[4] = {0.f, 0.f, 0.f, 0.f};
  float mcell = 0.f;
  if (wid < 2) {
    mcell = fmaxf(xch[l15], xch[16 + l15]);
    float l4 = 0.f, lr4 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e4[r] = (mcell == -INFINITY) ? 0.f : __expf(sc4[r] - mcell);
      l4 += e4[r];
      lr4 = fmaf(e4[r], rs4[r], lr4);
    }
    l4 = xlane_sum(l4);
    lr4 = xlane_sum(lr4);
    if (q4 == 0) { xch[32 + wid * 16 + l15] = l4; xch[64 + wid * 16 + l15] = lr4; }
  }
  __syncthreads();
  if (wid >= 2 || l15 >= nq) return;
  const long long so = (tok0 + l15) * a.Sp_tot + cell * 32 + 16 * wid + 4 * q4;
  *reinterpret_cast<float4*>(a.sc + so) = make_float4(e4[0], e4[1], e4[2], e4[3]);
  *reinterpret_cast<float4*>(a.rsp + so) = e_rs;   // the keys' scales next to them: the second half then has no load that waits for another
  if (wid == 0 && q4 == 0)
    reinterpret_cast<float4*>(a.cst)[(tok0 + l15) * (a.Sp_tot / 32) + cell] = make_float4(mcell, xch[32 + l15] + xch[48 + l15], xch[64 + l15] + xch[80 + l15], 0.f);
}

// ------------------------------------------------------------------------------------------------
// Cross-attention, second half: P' = p rs from the cells' statistics, x += sum_j (VA_j^T P'_j + (sum P'_j) VV_j b_t) + bias for 16
// features.  MAXKEYS: capacity in padded keys (512: the product shape's 320 keys; 1024).  Thread (row pr, lane plr) owns the 4-key
// chunks plr + 32 n: 16-byte loads contiguous across the row's lanes.
// dynamic LDS: P' image (Sp_tot / 32) x 2 KB | 8 x MAXN x 2 KB staging | cell statistics [16][32] float4 | sum_s P'_s [16][8]
// ------------------------------------------------------------------------------------------------
template <int MAXKEYS>
__global__ void __launch_bounds__(512) rt_xpv_kernel(const RtXArgs a) {
  constexpr int NW = 8;
  constexpr int MAXC = MAXKEYS / 4 / RT_LPR;                     // 4-key chunks per lane
  constexpr int MAXN = MAXKEYS / 32 / NW;                        // k-groups per wave
  constexpr int STAGE_W = MAXN * 2048;
  RT_T(t_in);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KT = a.Sp_tot / 32;
  char* img = smem;                                              // P' image: Sp_tot / 32 k-groups x 2 KB
  char* stage0 = smem + KT * 2048;
  float4* cst = reinterpret_cast<float4*>(stage0 + NW * STAGE_W);   // [16 tokens][32 cells]
  float* ws = reinterpret_cast<float*>(cst + 16 * 32);           // [16 tokens][8]: sum_s P'_s per memory
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* wreg = stage0 + wid * STAGE_W;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int fb = blockIdx.x, tile = blockIdx.y, f0 = fb * 16;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  static_assert(CFD_NMEM == 5, "five named instance indices");
  int u0, u1, u2, u3, u4;   // (named scalars: a local array indexed through rt_sel goes to scratch)
  if (a.use_inst) { u0 = rt_inst_get(a.inst[0], b); u1 = rt_inst_get(a.inst[1], b); u2 = rt_inst_get(a.inst[2], b); u3 = rt_inst_get(a.inst[3], b); u4 = rt_inst_get(a.inst[4], b); }
  else { u0 = a.map[0][b]; u1 = a.map[1][b]; u2 = a.map[2][b]; u3 = a.map[3][b]; u4 = a.map[4][b]; }
  auto inst = [&](int j) __attribute__((always_inline)) { return j == 0 ? u0 : j == 1 ? u1 : j == 2 ? u2 : j == 3 ? u3 : u4; };

  // per-memory scalars (see RT_PIN_S)
  int o1 = a.off[1], o2 = a.off[2], o3 = a.off[3], o4 = a.off[4];
  int sp0 = a.Sp[0], sp1 = a.Sp[1], sp2 = a.Sp[2], sp3 = a.Sp[3], sp4 = a.Sp[4];
  int sz0 = a.S[0], sz1 = a.S[1], sz2 = a.S[2], sz3 = a.S[3], sz4 = a.S[4];
  RT_PIN_S(o1); RT_PIN_S(o2); RT_PIN_S(o3); RT_PIN_S(o4);
  RT_PIN_S(sp0); RT_PIN_S(sp1); RT_PIN_S(sp2); RT_PIN_S(sp3); RT_PIN_S(sp4);
  RT_PIN_S(sz0); RT_PIN_S(sz1); RT_PIN_S(sz2); RT_PIN_S(sz3); RT_PIN_S(sz4);
  const bool any_att = a.att[0] || a.att[1] || a.att[2] || a.att[3] || a.att[4];   // (attention maps wanted: cfd_forward, the WEG evaluation)
  auto mem_of = [&](int key) __attribute__((always_inline)) { return key >= o4 ? 4 : key >= o3 ? 3 : key >= o2 ? 2 : key >= o1 ? 1 : 0; };

  // ---- loads, all requested before anything waits: the V^T slices of this wave's k-groups (LDS-DMA), the tile's e_s and per-key
  // scales (prologue lanes; chunks past the last key load the last chunk and are ignored), the cell statistics, the epilogue's operands.
  // (The statistics used to be loaded AND stored to LDS in front of the DMA: the store waited for every load before it, and the slices
  //  were requested one round trip late.)
  const int nk = (KT - wid + NW - 1) / NW;
#pragma unroll
  for (int n = 0; n < MAXN; ++n) {
    if (n < nk) {
      const int kt = wid + NW * n;
      const int j = mem_of(kt * 32);
      const long long ld = (long long)rt_pick5(j, sp0, sp1, sp2, sp3, sp4) * 4;
      rt_dma_slice(wreg + n * 2048, rt_sel(a.VT, j) + ((size_t)inst(j) * CFD_D + f0) * ld, ld, 15, kt - rt_pick5(j, 0, o1, o2, o3, o4) / 32, lane);
    }
  }
  RT_T(t_x1);
  const int pr = threadIdx.x >> 5, plr = threadIdx.x & 31;
  float s[MAXC][4], rsv[MAXC][4];
  const long long srow = (tok0 + min(pr, nq - 1)) * a.Sp_tot;
#pragma unroll
  for (int n = 0; n < MAXC; ++n) {
    const int c0 = min((plr + RT_LPR * n) * 4, a.Sp_tot - 4);
    const float4 p0 = *reinterpret_cast<const float4*>(a.sc + srow + c0);
    const float4 r0 = *reinterpret_cast<const float4*>(a.rsp + srow + c0);
    s[n][0] = p0.x; s[n][1] = p0.y; s[n][2] = p0.z; s[n][3] = p0.w;
    rsv[n][0] = r0.x; rsv[n][1] = r0.y; rsv[n][2] = r0.z; rsv[n][3] = r0.w;
  }
  RT_T(t_x2);
  const float4 cst_mine = reinterpret_cast<const float4*>(a.cst)[(tok0 + min(pr, nq - 1)) * KT + min(plr, KT - 1)];
  RT_T(t_x3);
  const int fcol = f0 + 4 * q4;
  float4 ep_r = make_float4(0.f, 0.f, 0.f, 0.f), ep_b = ep_r;
  float4 ep_vb[CFD_NMEM];
  if (wid == 0) {
    ep_r = *reinterpret_cast<const float4*>(a.x + (tok0 + min(l15, nq - 1)) * CFD_D + fcol);
    ep_b = *reinterpret_cast<const float4*>(a.bias + fcol);
#pragma unroll
    for (int j = 0; j < CFD_NMEM; ++j) ep_vb[j] = *reinterpret_cast<const float4*>(a.vb[j] + fcol);
  }
  __builtin_amdgcn_sched_barrier(0);
  RT_T(t_a);
  if (plr < KT) cst[pr * 32 + plr] = cst_mine;
#if RT_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    const unsigned k_ = atomicAdd(&g_rt_seq, 1u) & 4095u;
    g_rt_ring[4 * k_] = 3002; g_rt_ring[4 * k_ + 1] = t_x1 - t_in; g_rt_ring[4 * k_ + 2] = t_x2 - t_in; g_rt_ring[4 * k_ + 3] = t_x3 - t_in;
  }
#endif
  __syncthreads();                                                // the cell statistics are in LDS
  // per (token, memory): M, 1 / sum_c l_c exp(m_c - M), sum_s P'_s -- once, by 80 threads (serial walks over a memory's cells by every
  // lane cost more than the softmax they replaced)
  float4* seg = reinterpret_cast<float4*>(ws + 16 * 8);           // [16 tokens][8]
  if (threadIdx.x < 16 * CFD_NMEM) {
    const int r = threadIdx.x / CFD_NMEM, j = threadIdx.x - r * CFD_NMEM;
    const float4* cr = cst + r * 32;
    const int offj = rt_pick5(j, 0, o1, o2, o3, o4), cb = offj >> 5, ce = (offj + rt_pick5(j, sp0, sp1, sp2, sp3, sp4)) >> 5;
    float M = -INFINITY, l = 0.f, w = 0.f;
    for (int k = cb; k < ce; ++k) M = fmaxf(M, cr[k].x);
    for (int k = cb; k < ce; ++k) {
      const float ex = __expf(cr[k].x - M);
      l = fmaf(cr[k].y, ex, l);
      w = fmaf(cr[k].z, ex, w);
    }
    const float il = rt_rcp(l);                                   // all keys of the memory dead: NaN like the reference's softmax
    seg[r * 8 + j] = make_float4(M, il, 0.f, 0.f);
    ws[r * 8 + j] = w * il;
  }
  __syncthreads();
  // ---- probabilities from the cells' statistics; P' -> image ----------------------------------
  const float4* crow = cst + pr * 32;
#pragma unroll
  for (int n = 0; n < MAXC; ++n) {
    const int c0 = (plr + RT_LPR * n) * 4;
    if (c0 < a.Sp_tot) {
      const int j = mem_of(c0);
      const float4 sg = seg[pr * 8 + j];
      const float f = __expf(crow[c0 >> 5].x - sg.x) * sg.y;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[n][e] *= f;
      if (any_att && fb == 0 && pr < nq && b >= a.att_b0 && b < a.att_b0 + a.att_nb) {
        float* att = rt_sel(a.att, j);
        if (att) {
          const int S = rt_pick5(j, sz0, sz1, sz2, sz3, sz4);
          if (a.att_step) att += (long long)(*a.att_step) * rt_sel(a.att_slot, j);
          float* ap = att + (((long long)(b - a.att_b0) * a.nl + a.layer) * a.L + q0 + pr) * S;
          const int k0 = c0 - rt_pick5(j, 0, o1, o2, o3, o4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k0 + e < S) ap[k0 + e] = s[n][e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[n][e] *= rsv[n][e];
      rt_lstore4(img, pr, c0, s[n]);
    }
  }
  RT_T(t_b);
  RT_WAIT_VM0();
  __syncthreads();
  RT_T(t_ops);
#if RT_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // first record: entry, loads issued, softmax done
    const unsigned k_ = atomicAdd(&g_rt_seq, 1u) & 4095u;
    g_rt_ring[4 * k_] = 3001; g_rt_ring[4 * k_ + 1] = t_in; g_rt_ring[4 * k_ + 2] = t_a; g_rt_ring[4 * k_ + 3] = t_b;
  }
  struct StampAtExit { unsigned long long a, b; int id; __device__ ~StampAtExit() { RT_STAMP_OUT(id, a, b); } } stamp_{t_in, t_ops, 3000};
#endif
  f32x4 part[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int n = 0; n < MAXN; ++n)
    if (n < nk) {
      spx8 ph, pl, vh, vl;
      rt_lfrag(img, wid + NW * n, l15, q4, ph, pl);
      rt_lfrag(wreg, n, l15, q4, vh, vl);
      part[0] = rt_mma(vh, vl, ph, pl, part[0]);   // O^T[feature][token]
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const f32x4 acc = rt_reduce<NW, 1>(stage0, STAGE_W, wid, lane, part);
  if (wid != 0 || l15 >= nq) return;
  float o[4] = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
  for (int j = 0; j < CFD_NMEM; ++j) {
    const float w = ws[l15 * 8 + j];
    o[0] += w * ep_vb[j].x; o[1] += w * ep_vb[j].y; o[2] += w * ep_vb[j].z; o[3] += w * ep_vb[j].w;
  }
  float4 r = ep_r;
  r.x = (r.x + ep_b.x) + o[0]; r.y = (r.y + ep_b.y) + o[1]; r.z = (r.z + ep_b.z) + o[2]; r.w = (r.w + ep_b.w) + o[3];
  *reinterpret_cast<float4*>(a.xo + (tok0 + l15) * CFD_D + fcol) = r;
}

// ------------------------------------------------------------------------------------------------
// Start of a captured iteration on the row-tile path: row *d_step of every per-step table -> the fixed "this step" buffers the
// launches of the iteration read (so that none of them has the step index as a dependent scalar load in front of its operands)
// ------------------------------------------------------------------------------------------------
#define RT_NTAB 16
struct RtStepRowsArgs {
  const float* src[RT_NTAB];    // table base; row t at src + t * n4 * 4 floats
  float* dst[RT_NTAB];
  int n4[RT_NTAB];              // float4 per row
  int first[RT_NTAB + 1];       // first workgroup of table k
  int ntab;
  const int* d_step;
};
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) rt_step_rows_kernel(const RtStepRowsArgs a) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < RT_NTAB; ++q)
    if (q < a.ntab && (int)blockIdx.x >= a.first[q]) k = q;
  const float* src = a.src[0];
  float* dst = a.dst[0];
  int n4 = a.n4[0], first = 0;
#pragma unroll
  for (int q = 1; q < RT_NTAB; ++q)
    if (k == q) { src = a.src[q]; dst = a.dst[q]; n4 = a.n4[q]; first = a.first[q]; }
  const int i = ((int)blockIdx.x - first) * 256 + threadIdx.x;
  if (i < n4) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[(long long)(*a.d_step) * n4 + i];
}

// ------------------------------------------------------------------------------------------------
// mem_scale_all_kernel (rows.hpp) for EVERY step of a run at once: table[t][l][key] = cbk, table[t][nl][key] = rs
// grid (ceil(rows / 4), T)
// ------------------------------------------------------------------------------------------------
struct MemScaleTabArgs {
  const char* a_sp;    // SP [rows][512]
  const float* asq;    // [rows]
  long long rows;
  const float* btab;   // [T][512] centred timestep embeddings
  const float* bsq;    // [T]
  const float* ca;     // [nl][rows]
  const float* cbb;    // table row t: cbb[t * cbb_tstride + l]
  long long cbb_tstride;
  int nl;
  float* tab;          // [T][nl + 1][rows]
};

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) mem_scale_table_kernel(const MemScaleTabArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int t = blockIdx.y;
  const char* ap = a.a_sp + row * (CFD_D * 4) + (size_t)(lane >> 2) * 128 + (lane & 3) * 16;
  const spx8 h = *reinterpret_cast<const spx8*>(ap);
  const spx8 l = *reinterpret_cast<const spx8*>(ap + 64);
  const float* bp = a.btab + (long long)t * CFD_D + lane * 8;
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) dot += ((float)h[e] + (float)l[e]) * bp[e];
  dot = wave_sum(dot);
  const float var = (a.asq[row] + 2.0f * dot + a.bsq[t]) * (1.0f / CFD_D);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  float* out = a.tab + (long long)t * (a.nl + 1) * a.rows;
  if (lane == 0) out[(long long)a.nl * a.rows + row] = rstd;
  if (lane < a.nl) out[(long long)lane * a.rows + row] = rstd * (a.ca[(long long)lane * a.rows + row] + a.cbb[(long long)t * a.cbb_tstride + lane]);
}
