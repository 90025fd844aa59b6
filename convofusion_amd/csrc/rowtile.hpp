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

#define RT_MAX_L 32          // self-attention keys of a batch row fit one 32-deep k-step
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

// fragment of k-group kt of an SP row in global memory (row_base points at the row): lane q4 takes k = 8 q4 .. 8 q4 + 7
__device__ __forceinline__ void rt_gfrag(const char* row_base, int kt, int q4, spx8& hi, spx8& lo) {
  const char* p = row_base + (size_t)kt * 128 + q4 * 16;
  hi = *reinterpret_cast<const spx8*>(p);
  lo = *reinterpret_cast<const spx8*>(p + 64);
}

// LDS image of a 16-row SP operand: [k-group][row 16][128 B], 16-byte chunk c of a row stored at position c ^ ((row >> 1) & 7)
// (the swizzle of gemm_sp.hpp: conflict-free ds_read_b128 fragment reads)
__device__ __forceinline__ void rt_lfrag(const char* img, int kt, int l15, int q4, spx8& hi, spx8& lo) {
  const int sw = (l15 >> 1) & 7;
  const char* p = img + kt * 2048 + l15 * 128;
  hi = *reinterpret_cast<const spx8*>(p + ((q4 ^ sw) << 4));
  lo = *reinterpret_cast<const spx8*>(p + (((4 + q4) ^ sw) << 4));
}
// store 8 consecutive columns (c0 % 8 == 0) of row r into the image
__device__ __forceinline__ void rt_lstore8(char* img, int r, int c0, const float* v) {
  spx8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sp_t a, b;
    split_f32(v[e], a, b);
    h[e] = a;
    l[e] = b;
  }
  const int sw = (r >> 1) & 7, ch = (c0 & 31) >> 3;
  char* p = img + (c0 >> 5) * 2048 + r * 128;
  *reinterpret_cast<spx8*>(p + ((ch ^ sw) << 4)) = h;
  *reinterpret_cast<spx8*>(p + (((4 + ch) ^ sw) << 4)) = l;
}

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
__device__ __forceinline__ float rt_silu(float x) { return x * rt_rcp(1.0f + __expf(-x)); }

// sum of the waves' partial accumulators of NFB feature blocks (fixed order: bit-reproducible); block i's sum is returned in wave i
template <int NW, int NFB>
__device__ __forceinline__ f32x4 rt_reduce(char* red, int wid, int lane, const f32x4 (&acc)[NFB]) {
#pragma unroll
  for (int i = 0; i < NFB; ++i) reinterpret_cast<f32x4*>(red)[(i * NW + wid) * 64 + lane] = acc[i];
  __syncthreads();
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wid < NFB) {
    s = reinterpret_cast<const f32x4*>(red)[(wid * NW) * 64 + lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const f32x4 t = reinterpret_cast<const f32x4*>(red)[(wid * NW + w) * 64 + lane];
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
  }
  return s;
}

// LayerNorm statistics of a prologue row: v (this lane's columns) is centred in place, returns 1 / sqrt(var + eps)
template <int LPR, int CH>
__device__ __forceinline__ float rt_ln_stats(float (&v)[CH][8]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[c][e];
  const float mean = rt_row_sum<LPR>(s) * (1.0f / CFD_D);
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[c][e] -= mean; ss += v[c][e] * v[c][e]; }
  return 1.0f / sqrtf(rt_row_sum<LPR>(ss) * (1.0f / CFD_D) + 1e-5f);
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
  const float* ss;         // AdaLN: (1 + scale | shift) of this time block, row t at ss + t * ss_tstride
  long long ss_tstride;
  const int* d_step;
  // W operand
  const char* w;           // SP [N][K]
  const char* w2;          // RT_EPI_QKV: the value projection's weights (feature blocks >= nfb_qk, operand roles swapped)
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

template <int PRO, int EPI, int NT, int KT, int NFB>
__global__ void __launch_bounds__(NT, NT / 128) rt_gemm_kernel(const RtGemmArgs a) {
  constexpr int NW = NT / 64;
  constexpr int NK = KT / NW;                // k-groups per wave: kt = wid + NW * n
  constexpr int LPR = NT / 16;               // lanes per token row in the prologue
  constexpr int CH = CFD_D / (LPR * 8);      // 8-column chunks per lane
  static_assert(KT % NW == 0 && NK >= 1, "the k-groups divide evenly over the waves");
  static_assert(PRO == RT_PRO_SP || KT == CFD_D / 32, "LayerNorm prologues are 512 wide");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img = smem;                                              // A image (LN prologues): 16 k-groups x 2 KB
  char* red = smem + (PRO == RT_PRO_SP ? 0 : 16 * 2048);         // reduction scratch: NFB x NW KB
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tile = blockIdx.y;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  const int fb0 = blockIdx.x * NFB;                              // first 16-feature block of this workgroup
  const bool swapped = EPI == RT_EPI_QKV && fb0 >= a.nfb_qk;     // workgroup-uniform (nfb_qk is a multiple of NFB)
  const int f0 = (swapped ? fb0 - a.nfb_qk : fb0) * 16;

  // ---- 1. every global load of the kernel is issued here ------------------------------------------------
  // prologue rows first (vmcnt retires in order: the LayerNorm can start while the weight fragments are still in flight)
  float v[CH][8];
  float4 lg[CH][2], lb[CH][2], ls[CH][2], lh[CH][2];             // LayerNorm weight / bias, AdaLN (1 + scale) / shift of this lane's columns
  const int pr = threadIdx.x / LPR, plr = threadIdx.x % LPR;     // prologue: row, lane in row
  if constexpr (PRO != RT_PRO_SP) {
    const float* xr = a.x + (tok0 + min(pr, nq - 1)) * CFD_D + plr * 8;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4 p0 = *reinterpret_cast<const float4*>(xr + c * (LPR * 8));
      const float4 p1 = *reinterpret_cast<const float4*>(xr + c * (LPR * 8) + 4);
      v[c][0] = p0.x; v[c][1] = p0.y; v[c][2] = p0.z; v[c][3] = p0.w; v[c][4] = p1.x; v[c][5] = p1.y; v[c][6] = p1.z; v[c][7] = p1.w;
    }
    const float* sc = nullptr;
    if constexpr (PRO == RT_PRO_ADALN) sc = a.ss + (long long)(*a.d_step) * a.ss_tstride;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int c0 = c * (LPR * 8) + plr * 8;
      lg[c][0] = *reinterpret_cast<const float4*>(a.g + c0); lg[c][1] = *reinterpret_cast<const float4*>(a.g + c0 + 4);
      lb[c][0] = *reinterpret_cast<const float4*>(a.b + c0); lb[c][1] = *reinterpret_cast<const float4*>(a.b + c0 + 4);
      if constexpr (PRO == RT_PRO_ADALN) {
        ls[c][0] = *reinterpret_cast<const float4*>(sc + c0); ls[c][1] = *reinterpret_cast<const float4*>(sc + c0 + 4);
        lh[c][0] = *reinterpret_cast<const float4*>(sc + CFD_D + c0); lh[c][1] = *reinterpret_cast<const float4*>(sc + CFD_D + c0 + 4);
      }
    }
  }
  spx8 wh[NFB][NK], wl[NFB][NK], ah[NK], al[NK];
#pragma unroll
  for (int i = 0; i < NFB; ++i) {
    const char* wrow = (swapped ? a.w2 : a.w) + (size_t)(f0 + 16 * i + l15) * (KT * 128);
#pragma unroll
    for (int n = 0; n < NK; ++n) rt_gfrag(wrow, wid + NW * n, q4, wh[i][n], wl[i][n]);
  }
  if constexpr (PRO == RT_PRO_SP) {
    const char* arow = a.a_sp + (size_t)(tok0 + min(l15, nq - 1)) * (KT * 128);
#pragma unroll
    for (int n = 0; n < NK; ++n) rt_gfrag(arow, wid + NW * n, q4, ah[n], al[n]);
  }
  // epilogue operands: wave i finishes feature block i
  float4 ep_r = make_float4(0.f, 0.f, 0.f, 0.f), ep_t = ep_r, ep_t1 = ep_r, ep_t2 = ep_r;
  const int fcol = f0 + 16 * wid + 4 * q4;                       // standard roles: this lane's 4 features, token l15
  if (wid < NFB) {
    if constexpr (EPI == RT_EPI_RESID) ep_r = *reinterpret_cast<const float4*>(a.xr + (tok0 + min(l15, nq - 1)) * CFD_D + fcol);
    if (a.bias && !swapped) ep_t = *reinterpret_cast<const float4*>(a.bias + fcol);
    if constexpr (EPI == RT_EPI_EMBED) {
      const int l = q0 + min(l15, nq - 1);
      ep_t1 = *reinterpret_cast<const float4*>(a.bh + (l & 1) * CFD_D + fcol);
      ep_t2 = *reinterpret_cast<const float4*>(a.qpe + (size_t)(l >> 1) * CFD_D + fcol);
    }
  }

  // ---- 2. prologue: LayerNorm (+ AdaLN, SiLU) of the tile's rows -> split-pair image in LDS ---------------
  if constexpr (PRO != RT_PRO_SP) {
    const float rstd = rt_ln_stats<LPR, CH>(v);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int c0 = c * (LPR * 8) + plr * 8;
      const float gg[8] = {lg[c][0].x, lg[c][0].y, lg[c][0].z, lg[c][0].w, lg[c][1].x, lg[c][1].y, lg[c][1].z, lg[c][1].w};
      const float bb[8] = {lb[c][0].x, lb[c][0].y, lb[c][0].z, lb[c][0].w, lb[c][1].x, lb[c][1].y, lb[c][1].z, lb[c][1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = v[c][e] * rstd * gg[e] + bb[e];
      if constexpr (PRO == RT_PRO_ADALN) {
        const float sv[8] = {ls[c][0].x, ls[c][0].y, ls[c][0].z, ls[c][0].w, ls[c][1].x, ls[c][1].y, ls[c][1].z, ls[c][1].w};
        const float hv[8] = {lh[c][0].x, lh[c][0].y, lh[c][0].z, lh[c][0].w, lh[c][1].x, lh[c][1].y, lh[c][1].z, lh[c][1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = rt_silu(v[c][e] * sv[e] + hv[e]);
      }
      rt_lstore8(img, pr, c0, v[c]);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NK; ++n) rt_lfrag(img, wid + NW * n, l15, q4, ah[n], al[n]);
  }

  // ---- 3. product over this wave's k-groups, then the sum over the waves -----------------------------------
  f32x4 part[NFB];
#pragma unroll
  for (int i = 0; i < NFB; ++i) {
    part[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NK; ++n) part[i] = swapped ? rt_mma(ah[n], al[n], wh[i][n], wl[i][n], part[i]) : rt_mma(wh[i][n], wl[i][n], ah[n], al[n], part[i]);
  }
  const f32x4 acc = rt_reduce<NW, NFB>(red, wid, lane, part);
  if (wid >= NFB) return;

  // ---- 4. epilogue (wave i: feature block i).  Standard roles: lane (token l15) holds features fcol .. fcol + 3 ---------------
  if constexpr (EPI == RT_EPI_QKV) {
    if (swapped) {   // lane (feature f0 + 16 wid + l15) holds tokens q0 + 4 q4 .. + 3 of V^T; tokens beyond L are stored as zero
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (4 * q4 + r < nq) ? acc[r] : 0.f;
      sp_store4(a.vt + ((size_t)b * CFD_D + f0 + 16 * wid + l15) * (RT_MAX_L * 4), q0 + 4 * q4, o[0], o[1], o[2], o[3]);
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
// pre-scaled in the weights).  No LDS: every wave forms the 16 x 32 scores itself (S^T[key][query] = K . Q, 24 MFMAs) and
// takes 2 of the head's 8 feature tiles of the P.V product.  The key rows are assigned to MFMA rows so that a lane's 8
// score registers are keys 8 g .. 8 g + 7: exactly the k-slots it supplies as the second operand of P.V (as in attn_fused.hpp).
// ------------------------------------------------------------------------------------------------
struct RtSelfArgs {
  const char* qk;    // SP [M][1024]: q at k-groups 4 h .. 4 h + 3, k at 16 + 4 h ..
  const char* vt;    // SP [Be][512][32 keys]
  char* o;           // SP [M][512]
  int L, tpr;
};

__global__ void __launch_bounds__(256) rt_selfattn_kernel(const RtSelfArgs a) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int h = blockIdx.x, tile = blockIdx.y;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L;
  const char* qrow = a.qk + (size_t)(tok0 + q0 + min(l15, nq - 1)) * 4096;
  const int key0 = 8 * (l15 >> 2) + (l15 & 3), key1 = key0 + 4;       // MFMA row l15 of key tile 0 / 1
  const char* krow0 = a.qk + (size_t)(tok0 + min(key0, a.L - 1)) * 4096;
  const char* krow1 = a.qk + (size_t)(tok0 + min(key1, a.L - 1)) * 4096;
  spx8 qh[4], ql[4], k0h[4], k0l[4], k1h[4], k1l[4], vh[2], vl[2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    rt_gfrag(qrow, 4 * h + g, q4, qh[g], ql[g]);
    rt_gfrag(krow0, 16 + 4 * h + g, q4, k0h[g], k0l[g]);
    rt_gfrag(krow1, 16 + 4 * h + g, q4, k1h[g], k1l[g]);
  }
#pragma unroll
  for (int n = 0; n < 2; ++n)
    rt_gfrag(a.vt + ((size_t)b * CFD_D + h * CFD_HD + (2 * wid + n) * 16 + l15) * (RT_MAX_L * 4), 0, q4, vh[n], vl[n]);
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
  spx8 ph, pl;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sp_t hi, lo;
    split_f32(p[e] / sum, hi, lo);
    ph[e] = hi;
    pl[e] = lo;
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    f32x4 o = rt_mma(vh[n], vl[n], ph, pl, f32x4{0.f, 0.f, 0.f, 0.f});   // O^T[feature][query]
    if (l15 < nq)
      sp_store4(a.o + (size_t)(tok0 + q0 + l15) * (CFD_D * 4), h * CFD_HD + (2 * wid + n) * 16 + 4 * q4, o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// Cross-attention, first half: LayerNorm2 of the tile's rows and the scores against 16 folded keys of one memory
// (cross_attention.py:578-652, folded + timestep-hoisted form, see the header):
//   sc[token][off_j + s] = rs_s (q . KA_s + q . (A b_t)) + cbk_s
// ------------------------------------------------------------------------------------------------
struct RtXArgs {
  const float* x;               // fp32 [M][512]: the residual stream in front of the cross-attention block
  float* xo;                    // where x + block(x) goes (xo == x: in place)
  const float* ln_g;            // norm2
  const float* ln_b;
  const float* bias;            // folded cross-attention bias [512]
  int L, tpr, nl, layer;
  const int* d_step;
  const char* K[CFD_NMEM];      // this layer's folded keys: SP [U_j * Sp_j][512]
  const char* VT[CFD_NMEM];     // this layer's folded values^T: SP [U_j][512][Sp_j]
  const float* cbt[CFD_NMEM];   // per-step key tables [T][nl + 1][U_j * Sp_j]: plane l = cbk of layer l, plane nl = rs
  const float* kb[CFD_NMEM];    // A_l b_t [512] at kb[j] + t * kb_stride
  const float* vb[CFD_NMEM];    // VV_l b_t [512] at vb[j] + t * vb_stride
  long long kb_stride[CFD_NMEM], vb_stride[CFD_NMEM];
  const int* map[CFD_NMEM];     // batch row -> memory instance
  int rows[CFD_NMEM];           // U_j * Sp_j
  int S[CFD_NMEM], Sp[CFD_NMEM], off[CFD_NMEM];
  int blk0[CFD_NMEM + 1];       // first 16-key block of memory j in the score launch's grid
  int Sp_tot;
  float* sc;                    // fp32 [M][Sp_tot] scores
  float* rsp;                   // fp32 [M][Sp_tot]: rs of every key, repeated per token by the score launch
  float* att[CFD_NMEM];         // optional att_mats [Be][nl][L][S_j]
};

template <int NT>
__global__ void __launch_bounds__(NT) rt_xscore_kernel(const RtXArgs a) {
  constexpr int NW = NT / 64, LPR = NT / 16, CH = CFD_D / (LPR * 8);
  constexpr int NK = 16 / NW;                                    // k-groups per wave (K = 512)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img = smem;
  char* red = smem + 16 * 2048;
  float* cq = reinterpret_cast<float*>(smem + 16 * 2048 + NW * 1024);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tile = blockIdx.y;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  int j = 0;
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if ((int)blockIdx.x >= a.blk0[q]) j = q;
  int blk_first = 0;
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if (j == q) blk_first = a.blk0[q];
  const int s0 = ((int)blockIdx.x - blk_first) * 16;
  const int t = *a.d_step;
  const int u = rt_sel(a.map, j)[b];
  const int rows = rt_sel(a.rows, j), Sp = rt_sel(a.Sp, j);
  const long long key0 = (long long)u * Sp + s0;

  const int pr = threadIdx.x / LPR, plr = threadIdx.x % LPR;
  float v[CH][8];
  {
    const float* xr = a.x + (tok0 + min(pr, nq - 1)) * CFD_D + plr * 8;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float4 p0 = *reinterpret_cast<const float4*>(xr + c * (LPR * 8));
      const float4 p1 = *reinterpret_cast<const float4*>(xr + c * (LPR * 8) + 4);
      v[c][0] = p0.x; v[c][1] = p0.y; v[c][2] = p0.z; v[c][3] = p0.w; v[c][4] = p1.x; v[c][5] = p1.y; v[c][6] = p1.z; v[c][7] = p1.w;
    }
  }
  spx8 kh[NK], kl[NK], ah[NK], al[NK];
  {
    const char* krow = rt_sel(a.K, j) + (size_t)(key0 + l15) * (CFD_D * 4);
#pragma unroll
    for (int n = 0; n < NK; ++n) rt_gfrag(krow, wid + NW * n, q4, kh[n], kl[n]);
  }
  const float* kbp = rt_sel(a.kb, j) + (long long)t * rt_sel(a.kb_stride, j);
  float4 lg[CH][2], lb[CH][2], lk[CH][2];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int c0 = c * (LPR * 8) + plr * 8;
    lg[c][0] = *reinterpret_cast<const float4*>(a.ln_g + c0); lg[c][1] = *reinterpret_cast<const float4*>(a.ln_g + c0 + 4);
    lb[c][0] = *reinterpret_cast<const float4*>(a.ln_b + c0); lb[c][1] = *reinterpret_cast<const float4*>(a.ln_b + c0 + 4);
    lk[c][0] = *reinterpret_cast<const float4*>(kbp + c0); lk[c][1] = *reinterpret_cast<const float4*>(kbp + c0 + 4);
  }
  const float* tab = rt_sel(a.cbt, j) + (long long)t * (a.nl + 1) * rows;
  float4 e_rs = make_float4(0.f, 0.f, 0.f, 0.f), e_cb = e_rs;
  if (wid == 0) {
    e_rs = *reinterpret_cast<const float4*>(tab + (long long)a.nl * rows + key0 + 4 * q4);
    e_cb = *reinterpret_cast<const float4*>(tab + (long long)a.layer * rows + key0 + 4 * q4);
  }
  // LayerNorm2 -> image; c_q = q . (A b_t)
  {
    const float rstd = rt_ln_stats<LPR, CH>(v);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int c0 = c * (LPR * 8) + plr * 8;
      const float gg[8] = {lg[c][0].x, lg[c][0].y, lg[c][0].z, lg[c][0].w, lg[c][1].x, lg[c][1].y, lg[c][1].z, lg[c][1].w};
      const float bb[8] = {lb[c][0].x, lb[c][0].y, lb[c][0].z, lb[c][0].w, lb[c][1].x, lb[c][1].y, lb[c][1].z, lb[c][1].w};
      const float4 k0 = lk[c][0], k1 = lk[c][1];
      const float kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = v[c][e] * rstd * gg[e] + bb[e];
        dot += v[c][e] * kk[e];
      }
      rt_lstore8(img, pr, c0, v[c]);
    }
    dot = rt_row_sum<LPR>(dot);
    if (plr == 0) cq[pr] = dot;
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NK; ++n) rt_lfrag(img, wid + NW * n, l15, q4, ah[n], al[n]);
  f32x4 part[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int n = 0; n < NK; ++n) part[0] = rt_mma(kh[n], kl[n], ah[n], al[n], part[0]);   // S^T[key][token]
  const f32x4 acc = rt_reduce<NW, 1>(red, wid, lane, part);
  if (wid != 0 || l15 >= nq) return;
  const float c_q = cq[l15];
  const float4 o = make_float4(e_rs.x * (acc[0] + c_q) + e_cb.x, e_rs.y * (acc[1] + c_q) + e_cb.y, e_rs.z * (acc[2] + c_q) + e_cb.z,
                               e_rs.w * (acc[3] + c_q) + e_cb.w);
  const long long so = (tok0 + l15) * a.Sp_tot + rt_sel(a.off, j) + s0 + 4 * q4;
  *reinterpret_cast<float4*>(a.sc + so) = o;
  *reinterpret_cast<float4*>(a.rsp + so) = e_rs;   // the keys' scales next to the scores: the second half then has no load that waits for another
}

// ------------------------------------------------------------------------------------------------
// Cross-attention, second half: softmax per memory, P' = p rs, x += sum_j (VA_j^T P'_j + (sum P'_j) VV_j b_t) + bias for 16 features
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(NT) rt_xpv_kernel(const RtXArgs a) {
  constexpr int NW = NT / 64, LPR = NT / 16;
  constexpr int MAXC = RT_MAX_KEYS / 8 / LPR;                    // 8-key chunks per lane
  constexpr int MAXN = RT_MAX_KEYS / 32 / NW;                    // k-groups per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img = smem;                                              // P' image: Sp_tot / 32 k-groups x 2 KB
  char* red = smem + (a.Sp_tot / 32) * 2048;
  float* ws = reinterpret_cast<float*>(red + NW * 1024);         // [16 tokens][8]: sum_s P'_s per memory
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int fb = blockIdx.x, tile = blockIdx.y, f0 = fb * 16;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  const int t = *a.d_step;
  const int KT = a.Sp_tot / 32;
  static_assert(CFD_NMEM == 5, "five named instance indices");
  const int u0 = a.map[0][b], u1 = a.map[1][b], u2 = a.map[2][b], u3 = a.map[3][b], u4 = a.map[4][b];   // (named scalars: a local array indexed through rt_sel goes to scratch)
  auto inst = [&](int j) __attribute__((always_inline)) { return j == 0 ? u0 : j == 1 ? u1 : j == 2 ? u2 : j == 3 ? u3 : u4; };

  // ---- loads: the tile's scores and per-key scales (prologue lanes), the V^T fragments of this wave's k-groups --------
  const int pr = threadIdx.x / LPR, plr = threadIdx.x % LPR;
  float s[MAXC][8], rsv[MAXC][8];
  int cj[MAXC];
  const float* srow = a.sc + (tok0 + min(pr, nq - 1)) * a.Sp_tot;
#pragma unroll
  for (int n = 0; n < MAXC; ++n) {
    const int c0 = (plr + LPR * n) * 8;
    cj[n] = -1;
    if (c0 < a.Sp_tot) {
      int j = 0;
#pragma unroll
      for (int q = 1; q < CFD_NMEM; ++q)
        if (c0 >= a.off[q]) j = q;
      cj[n] = j;
      const float4 p0 = *reinterpret_cast<const float4*>(srow + c0), p1 = *reinterpret_cast<const float4*>(srow + c0 + 4);
      s[n][0] = p0.x; s[n][1] = p0.y; s[n][2] = p0.z; s[n][3] = p0.w; s[n][4] = p1.x; s[n][5] = p1.y; s[n][6] = p1.z; s[n][7] = p1.w;
      const float* rp = a.rsp + (tok0 + min(pr, nq - 1)) * a.Sp_tot + c0;
      const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
      rsv[n][0] = r0.x; rsv[n][1] = r0.y; rsv[n][2] = r0.z; rsv[n][3] = r0.w; rsv[n][4] = r1.x; rsv[n][5] = r1.y; rsv[n][6] = r1.z; rsv[n][7] = r1.w;
    }
  }
  spx8 vh[MAXN], vl[MAXN], ph[MAXN], pl[MAXN];
  const int nk = (KT - wid + NW - 1) / NW;
#pragma unroll
  for (int n = 0; n < MAXN; ++n) {
    if (n < nk) {
      const int kt = wid + NW * n;
      int j = 0;
#pragma unroll
      for (int q = 1; q < CFD_NMEM; ++q)
        if (kt * 32 >= a.off[q]) j = q;
      const int Sp = rt_sel(a.Sp, j);
      const char* vrow = rt_sel(a.VT, j) + ((size_t)inst(j) * CFD_D + f0 + l15) * ((size_t)Sp * 4);
      rt_gfrag(vrow, kt - rt_sel(a.off, j) / 32, q4, vh[n], vl[n]);
    }
  }
  const int fcol = f0 + 4 * q4;
  float4 ep_r = make_float4(0.f, 0.f, 0.f, 0.f), ep_b = ep_r;
  float4 ep_vb[CFD_NMEM];
  if (wid == 0) {
    ep_r = *reinterpret_cast<const float4*>(a.x + (tok0 + min(l15, nq - 1)) * CFD_D + fcol);
    ep_b = *reinterpret_cast<const float4*>(a.bias + fcol);
#pragma unroll
    for (int j = 0; j < CFD_NMEM; ++j) ep_vb[j] = *reinterpret_cast<const float4*>(a.vb[j] + (long long)t * a.vb_stride[j] + fcol);
  }

  // ---- softmax per memory over the row's lanes; P' -> image; sum_s P'_s -> ws ----------------------------------
#pragma unroll
  for (int j = 0; j < CFD_NMEM; ++j) {
    float mx = -INFINITY;
#pragma unroll
    for (int n = 0; n < MAXC; ++n)
      if (cj[n] == j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, s[n][e]);
      }
    mx = rt_row_max<LPR>(mx);
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < MAXC; ++n)
      if (cj[n] == j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[n][e] = __expf(s[n][e] - mx); sum += s[n][e]; }   // all keys dead: NaN, as the reference
      }
    sum = rt_row_sum<LPR>(sum);
    const float inv = rt_rcp(sum);
    float wsum = 0.f;
    float* att = a.att[j];
    const int S = a.S[j];
#pragma unroll
    for (int n = 0; n < MAXC; ++n)
      if (cj[n] == j) {
        const int c0 = (plr + LPR * n) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[n][e] = s[n][e] * inv;
        if (att && fb == 0 && pr < nq) {
          float* ap = att + (((long long)b * a.nl + a.layer) * a.L + q0 + pr) * S;
          const int k0 = c0 - a.off[j];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + e < S) ap[k0 + e] = s[n][e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[n][e] *= rsv[n][e]; wsum += s[n][e]; }
        rt_lstore8(img, pr, c0, s[n]);
      }
    wsum = rt_row_sum<LPR>(wsum);
    if (plr == 0) ws[pr * 8 + j] = wsum;
  }
  __syncthreads();
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < MAXN; ++n)
    if (n < nk) {
      rt_lfrag(img, wid + NW * n, l15, q4, ph[n], pl[n]);
      acc = rt_mma(vh[n], vl[n], ph[n], pl[n], acc);   // O^T[feature][token]
    }
  const f32x4 part[1] = {acc};
  acc = rt_reduce<NW, 1>(red, wid, lane, part);
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
// mem_scale_kernel (rows.hpp) for EVERY step of a run at once: table[t][l][key] = cbk, table[t][nl][key] = rs
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
