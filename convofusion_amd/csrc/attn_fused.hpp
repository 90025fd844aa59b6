// Fused (flash-style) self-attention for the 4-head, head-dim-128 block (cross_attention.py:568-572):
//   o[b, q, h*128:(h+1)*128] = softmax_k( q_h . k_h ) v_h          (q pre-scaled by 1/sqrt(128) in the weights)
// One workgroup = 8 waves = 128 queries of one (batch row, head); keys are consumed in tiles of 32 (double-buffered in LDS)
// with an online softmax, so neither the score matrix nor the probabilities ever leave the chip.
//
// MFMA orientation (v_mfma_f32_16x16x32, split-pair operands, 3 MFMAs per product):
//   S^T[key][q] = sum_d K[key][d] Q[q][d]      K tile from LDS (A operand), Q fragments in registers (B operand)
//   O^T[f][q]  += sum_key V^T[f][key] P[q][key] V^T tile from LDS (A operand), P straight from the S accumulators
// A lane (q = lane&15, g = lane>>4) ends the first product holding S[q][16t + 4g + r] (t = 0..1, r = 0..3).  Feeding those
// registers as the B operand of the second product means k-slot (8g + e) of 32-key step s is key
// 32s + 16(e>>2) + 4g + (e&3): the V^T tile must list its keys in that order, which the V^T GEMM's epilogue
// produces for free (EpiSplit::perm32) -- no transpose, no LDS round trip for P.
#pragma once
#include "cfd_common.hpp"

struct SelfAttnArgs {
  const char* qk;   // SP [M][1024]: q at columns h*128.., k at columns 512 + h*128..
  const char* vts;  // SP [Be][512][Lv]: V^T per batch row, keys permuted inside every 32-block, Lv = roundup(L, 32)
  char* o;          // SP [M][512]
  int L, Lv;
};

#define SELF_ATTN_WAVES 8
#define SA_WAIT_VM_LGKM0(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x70 | ((((N) >> 4) & 3) << 14))
template <int CFD_KI = 0>
__global__ void __launch_bounds__(SELF_ATTN_WAVES * 64, 4) self_attn_fused_kernel(const SelfAttnArgs a) {
  // two stages of (K tile 16 KB | V^T tile 16 KB): keys are consumed in tiles of 32; the LDS-DMA fill of tile kt + 1 runs
  // under the MFMAs of tile kt (one barrier per tile), and two workgroups per CU cover each other's softmax sections
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 32768, VOFF = 16384;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4, sw = l15 >> 1;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * (SELF_ATTN_WAVES * 16) + wid * 16 + l15;
  const bool qvalid = q < a.L;
  // A wave whose 16 queries all lie beyond L (the second workgroup of L = 196 has three) only helps with staging.
  const bool wave_active = (int)(blockIdx.x * (SELF_ATTN_WAVES * 16)) + wid * 16 < a.L;   // wave-uniform
  const int qc = qvalid ? q : a.L - 1;
  const long long ROW = 4096;  // bytes per qk row (1024 columns)

  const int cpos = lane & 7, rsub = lane >> 3;
  const int nkv = (a.L + 31) / 32;
  // fill of tile kt into stage kt & 1: 32 pieces of 1 KB (8 rows x 128 B), 4 per wave -- K: 4 k-steps x 4 row groups; V^T: 16 row groups
  auto fill = [&](int kt) __attribute__((always_inline)) {
    char* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int g = wid + SELF_ATTN_WAVES * n, ks = g >> 2, rg = g & 3;
      const int r = rg * 8 + rsub;
      const int key = min(kt * 32 + r, a.L - 1);
      const char* src = a.qk + ((long long)b * a.L + key) * ROW + (long long)(16 + h * 4 + ks) * 128 + ((cpos ^ ((r >> 1) & 7)) << 4);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + ks * 4096 + rg * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int rg = wid + SELF_ATTN_WAVES * n;
      const int r = rg * 8 + rsub;
      const char* src = a.vts + ((long long)b * CFD_D + h * 128 + r) * ((long long)a.Lv * 4) + (long long)kt * 128 + ((cpos ^ ((r >> 1) & 7)) << 4);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + VOFF + rg * 1024), 16, 0, 0);
    }
  };
  fill(0);

  // Q fragments (B operand): lane holds d = 32*ks + 8*q4 .. +7 of its query
  spx8 qh[4], ql[4];
  {
    const char* qp = a.qk + ((long long)b * a.L + qc) * ROW + (long long)(h * 4) * 128 + q4 * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qh[ks] = *reinterpret_cast<const spx8*>(qp + ks * 128);
      ql[ks] = *reinterpret_cast<const spx8*>(qp + ks * 128 + 64);
    }
  }
  f32x4 o[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) o[f] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f, mc_run = -INFINITY;   // running maximum, sum, and the exponent reference the sums are relative to

  for (int kt = 0; kt < nkv; ++kt) {
    SA_WAIT_VM_LGKM0(0);               // tile kt (the only fill in flight) has landed; on the first pass also the Q fragments
    __builtin_amdgcn_s_barrier();      // ... everywhere, and every wave is done with tile kt - 1 (the other stage)
    if (kt + 1 < nkv) fill(kt + 1);
    if (!wave_active) continue;
    const char* st = smem + (kt & 1) * STAGE;

    // S^T tile: 32 keys x 16 queries per wave
    f32x4 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char* kp = st + ks * 4096 + (t * 16 + l15) * 128;
        const spx8 xh = *reinterpret_cast<const spx8*>(kp + ((q4 ^ sw) << 4));
        const spx8 xl = *reinterpret_cast<const spx8*>(kp + (((4 + q4) ^ sw) << 4));
        s[t] = SP_MFMA(xl, qh[ks], s[t], 0, 0, 0);
        s[t] = SP_MFMA(xh, ql[ks], s[t], 0, 0, 0);
        s[t] = SP_MFMA(xh, qh[ks], s[t], 0, 0, 0);
      }
    }
    // mask the padding keys, online softmax over this tile (a query's 32 scores live in 4 lanes x 8 registers);
    // exp(x - m) = exp2(x c - m c), c = log2(e)
    constexpr float LOG2E = 1.44269504088896340736f;
    float p[8];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = kt * 32 + 16 * (e >> 2) + 4 * q4 + (e & 3);
      p[e] = key < a.L ? s[e >> 2][e & 3] : -INFINITY;
      mx = fmaxf(mx, p[e]);
    }
    mx = xlane_max(mx);
    const float m_new = fmaxf(m, mx);            // (finite: every tile has at least one valid key)
    const float mc = m_new * LOG2E;
    // The running sums are kept relative to the ROUNDED exponent reference mc of the tile that wrote them (mc_run), so the factor that moves
    // them to this tile's reference is exp2(mc_run - mc): exactly 1 while the maximum stands.  (Until round 5 it was exp2(fma(m, c, -mc)) --
    // the exact m c against the rounded one, i.e. 2^(rounding error of m c) instead of 1: 1 + 2e-6 per tile at ordinary scores, compounding
    // over the tiles, and 0.5 % at the logits of 1e5 the heavy-tailed stress weights produce.)
    const float scale = __builtin_amdgcn_exp2f(mc_run - mc);     // (first tile: mc_run = -inf -> 0, times sums that are still 0)
    mc_run = mc;
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p[e] = __builtin_amdgcn_exp2f(fmaf(p[e], LOG2E, -mc));
      ps += p[e];
    }
    lsum = lsum * scale + xlane_sum(ps);
    m = m_new;
    if (!__all(scale == 1.0f)) {
#pragma unroll
      for (int f = 0; f < 8; ++f) { o[f][0] *= scale; o[f][1] *= scale; o[f][2] *= scale; o[f][3] *= scale; }
    }
    // O^T += V^T P^T : P fragments come straight out of the S registers (k-slot order matches perm32)
    spx8 ph, pl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sp_t hi, lo;
      split_f32(p[e], hi, lo);
      ph[e] = hi;
      pl[e] = lo;
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const char* vp = st + VOFF + (f * 16 + l15) * 128;
      const spx8 xh = *reinterpret_cast<const spx8*>(vp + ((q4 ^ sw) << 4));
      const spx8 xl = *reinterpret_cast<const spx8*>(vp + (((4 + q4) ^ sw) << 4));
      o[f] = SP_MFMA(xl, ph, o[f], 0, 0, 0);
      o[f] = SP_MFMA(xh, pl, o[f], 0, 0, 0);
      o[f] = SP_MFMA(xh, ph, o[f], 0, 0, 0);
    }
  }
  // Epilogue.  The MFMA layout gives a lane 4 features of one query: 8-byte stores in 32-byte row segments.  Each
  // wave re-lays its 16 x 128 block through a private LDS strip instead (the K / V^T tiles are free now) and stores
  // 8 consecutive features per lane, 64 features per pass: 8 lanes cover a query, 8 queries per instruction.
  __syncthreads();   // every wave is done with the last K / V^T tile
  {
    constexpr int RS = 64 * 4 + 16;   // strip row stride in bytes (+16: conflict-free 16-byte writes); 64 features per pass
    char* strip = smem + wid * (16 * RS);
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const f32x4 src = o[half * 4 + f];
        *reinterpret_cast<f32x4*>(strip + l15 * RS + (f * 16 + q4 * 4) * 4) = f32x4{src[0] * inv, src[1] * inv, src[2] * inv, src[3] * inv};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private strip: no barrier needed
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
        const int qq = blockIdx.x * (SELF_ATTN_WAVES * 16) + wid * 16 + row;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(strip + row * RS + c8 * 4);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(strip + row * RS + c8 * 4 + 16);
        if (qq < a.L) {
          const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          sp_store8(a.o + ((long long)b * a.L + qq) * (CFD_D * 4), h * 128 + half * 64 + c8, v);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the second half overwrites the strip
    }
  }
}
