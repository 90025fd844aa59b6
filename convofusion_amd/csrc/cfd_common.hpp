// Common types and helpers for the ConvoFusion denoising-loop kernels (gfx950 / CDNA4 only).
//
// Numeric format used by every GEMM operand ("split pair", SP): a float32 value v is carried as two 16-bit
// floats hi = f16(v), lo = f16(v - hi) (fp16 by default, bf16 with -DCFD_SPLIT_F16=0); a product a*b is issued as
// THREE MFMAs (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi) with fp32 accumulation: ~2^-22 (fp16) / 2^-16 (bf16) relative
// operand error, i.e. fp32-class results from the 16-bit matrix cores.  (The reference's 1e-3 budget on the final
// latents cannot be met with plain bf16 operands: SURVEY.md fact 8; measurements: DESIGN.md section 2.)
//
// Memory layout of an SP matrix [R rows][K cols], K % 32 == 0: per row, K/32 groups of 128 bytes, each group =
// 32 hi values (64 B) followed by the 32 lo values (64 B).  One 128-byte line thus carries both halves of a
// 32-deep K-step for one row, which is what one LDS tile row holds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CFD_D 512        // text_encoded_dim (configs/modules/denoiser.yaml:4)
#define CFD_FF 1024      // ff_size (:6)
#define CFD_LAT 128      // latent_dim[-1] (configs/config_cf_beatdnd.yaml latent_dim [1,128])
#define CFD_NHEAD 4      // num_heads (:8)
#define CFD_HD 128       // CFD_D / CFD_NHEAD
#define CFD_NMEM 5       // spkemb, alsn, tlsn, apb, lsnemb (denoiser.py:220)
#define CFD_MAX_LAYERS 16

// Element type of the split pair.  CFD_SPLIT_F16=1 (default): two IEEE half floats (11+11 significant
// bits, ~2^-22 operand error; inputs are saturated to +-65504); CFD_SPLIT_F16=0: two bfloat16 (8+8 bits,
// ~2^-16, full float32 range).  Both run on the same-rate v_mfma_f32_16x16x32_{f16,bf16}.
#ifndef CFD_SPLIT_F16
#define CFD_SPLIT_F16 1
#endif
#if CFD_SPLIT_F16
typedef _Float16 sp_t;
#define SP_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
typedef __bf16 sp_t;
#define SP_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif
typedef __attribute__((ext_vector_type(8))) sp_t spx8;   // 8 halves = one MFMA operand fragment per lane
typedef __attribute__((ext_vector_type(4))) sp_t spx4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void split_f32(float v, sp_t& hi, sp_t& lo) {
#if CFD_SPLIT_F16
  v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
#endif
  hi = (sp_t)v;
  lo = (sp_t)(v - (float)hi);
}

// Saturation census.  split_f32 clamps to +-65504 without telling anybody.  Operands made inside the network (LayerNorm outputs,
// probabilities, GELU / SiLU outputs) are bounded by construction; operands whose magnitude follows the CALLER's data -- the
// weights, the centred memories a_s and the folded projections KA = A a_s, VA = VV a_s of them, and the
// sample / latents handed to an entry point -- are counted when one of them leaves the range, and the entry points refuse to go
// on (cfd_internal.hpp: check_saturation): a clamped key or value is a silently wrong attention.  The counters belong to the HANDLE
// (cfd_handle_s::sat, two words: [CFD_SAT_MEM] weights / memories / their projections, [CFD_SAT_IN] sample / latents) and reach
// the kernels through their arguments: a call reads what ITS launches counted, never another handle's or an earlier call's.
// One compare per element; the atomic only fires on a fault.  sat == nullptr: not counted (micro-benchmarks).
enum { CFD_SAT_MEM = 0, CFD_SAT_IN = 1 };
template <int N>
__device__ __forceinline__ void sat_note(unsigned int* sat, const float* v) {
  bool s = false;
#pragma unroll
  for (int e = 0; e < N; ++e) s = s || !(fabsf(v[e]) <= 65504.0f);   // (NaN counts)
  if (s && sat) atomicAdd(sat, 1u);
}

// store 4 consecutive-column values (col % 4 == 0) of one row into an SP matrix
__device__ __forceinline__ void sp_store4(char* row_base, int col, float a, float b, float c, float d) {
  spx4 h, l;
  sp_t t0, t1;
  split_f32(a, t0, t1); h[0] = t0; l[0] = t1;
  split_f32(b, t0, t1); h[1] = t0; l[1] = t1;
  split_f32(c, t0, t1); h[2] = t0; l[2] = t1;
  split_f32(d, t0, t1); h[3] = t0; l[3] = t1;
  char* p = row_base + (size_t)(col >> 5) * 128 + (col & 31) * 2;
  *reinterpret_cast<spx4*>(p) = h;
  *reinterpret_cast<spx4*>(p + 64) = l;
}

// store 8 consecutive-column values (col % 8 == 0)
__device__ __forceinline__ void sp_store8(char* row_base, int col, const float* v) {
  spx8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sp_t a, b;
    split_f32(v[e], a, b);
    h[e] = a;
    l[e] = b;
  }
  char* p = row_base + (size_t)(col >> 5) * 128 + (col & 31) * 2;
  *reinterpret_cast<spx8*>(p) = h;
  *reinterpret_cast<spx8*>(p + 64) = l;
}

// Developer experiment (-DCFD_SPLIT_SC1=1, round 5): the split-pair OUTPUT of a product stored write-through (`sc1`: the line leaves the
// XCD's L2 instead of staying in it), to see whether the 2.17 x over-fetch of the N = 1024 products' operand panels is their own output
// evicting them.  Off in the product.
#ifndef CFD_SPLIT_SC1
#define CFD_SPLIT_SC1 0
#endif
__device__ __forceinline__ void sp_store8_out(char* row_base, int col, const float* v) {
#if CFD_SPLIT_SC1
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  spx8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sp_t a, b;
    split_f32(v[e], a, b);
    h[e] = a;
    l[e] = b;
  }
  char* p = row_base + (size_t)(col >> 5) * 128 + (col & 31) * 2;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(__builtin_bit_cast(u32x4_t, h)) : "memory");
  asm volatile("global_store_dwordx4 %0, %1, off offset:64 sc1" ::"v"(p), "v"(__builtin_bit_cast(u32x4_t, l)) : "memory");
#else
  sp_store8(row_base, col, v);
#endif
}

// sp_store8 for values that may legitimately be NaN (LayerNorm outputs and probabilities of a row whose softmax had nothing but masked
// keys: the reference returns NaN for such a row, from the attention on through every later LayerNorm to the output).  split_f32's clamp
// (v_med3_f32 returns the minimum of the other two operands for a NaN) would store -65504, a finite and wrong operand; here the NaN is kept.
__device__ __forceinline__ void sp_store8_keep_nan(char* row_base, int col, const float* v) {
  spx8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sp_t a, b;
    split_f32(v[e], a, b);
    h[e] = (v[e] != v[e]) ? (sp_t)v[e] : a;
    l[e] = b;
  }
  char* p = row_base + (size_t)(col >> 5) * 128 + (col & 31) * 2;
  *reinterpret_cast<spx8*>(p) = h;
  *reinterpret_cast<spx8*>(p + 64) = l;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// v_exp_f32-based exponential (__expf): ~1e-7 relative near 0, grows with |x| only where the result is negligible
// sum / max over the 4 lanes {l, l^16, l^32, l^48} that share a query, in the vector ALU (v_permlane32_swap /
// v_permlane16_swap: with both operands the same register, the two results hold the value of the lane's own and of its
// partner's half / row) -- dependent ds_bpermute round trips otherwise
__device__ __forceinline__ float xlane_sum(float x) {
  unsigned xi = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  xi = __float_as_uint(x);
  auto q = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float xlane_max(float x) {
  unsigned xi = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  xi = __float_as_uint(x);
  auto q = __builtin_amdgcn_permlane16_swap(xi, xi, false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_f(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU for the FFN epilogue (45 M evaluations per launch; erff's two-range code costs 41 of the launch's 188 us).
// erf by Abramowitz & Stegun 7.1.26: erfc(a) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-a^2), t = 1 / (1 + p a),
// |error| <= 1.5e-7 -- the float32 noise level of the reference's own erf; 1 + erf(x) is formed as erfc(|x|) for x < 0,
// so the negative tail does not cancel.  |gelu error| <= 0.75e-7 |x|.
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float a = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(fmaf(0.3275911f, a, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erfc_a = poly * __expf(-a * a);
  return 0.5f * x * (x >= 0.f ? 2.0f - erfc_a : erfc_a);
}
