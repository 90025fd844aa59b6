// Row-tile backward sweep of the word-excitation-guidance gradient (convofusion.py:437-496, word_excitation_guidance.py:11-81): the
// reverse of rowtile.hpp's forward on the text-only guidance chunk (one token tile per batch row at the product shape), in
// the SAME folded formulation, so that the five cross-attentions cost two products per layer instead of five chains of four.
//
// Every backward product contracts over a weight's OUTPUT axis, i.e. reads W[k][n] with n contiguous -- the transpose of what
// the split-pair kernels want -- and gradients span many binades, so the products run on the float32 matrix core
// (v_mfma_f32_16x16x4_f32: exact float32 FMA chains, one scalar operand per lane and MFMA): lane (n, k) loads W[k][n] as it
// lies, no transposed weight copies and no operand scaling.  The folded keys / values are read back from their split-pair form
// (hi + lo is exact in float32).  Kernel shape as in rowtile.hpp: a workgroup = 16 tokens x 16 outputs, the K axis split over
// its 8 waves, all operand loads issued up front, an LDS reduction over the waves; the row-complete steps of the backward
// (LayerNorm backward, the TimeBlock's SiLU / modulation backward, the softmax backward) are PROLOGUES of the product that
// consumes them, computed by every workgroup of the tile; workgroup 0 also writes the updated running gradient.
//
// Per layer (top down), g = gradient at the layer's output rows:
//   B1  du = (g + LN1'(next layer's dy1)) W2         ; dh = du * gelu'(pre)            [16 x 1024]
//   B2  dy3 = dh W1                                                                    [16 x 512]
//   B3  dz2 = (g + LN3'(dy3)) Wtb2                   (writes g_a)
//   B4  dP = ((g_a + TB2'(dz2)) (VA + VV b)^T) rs + d_att    (writes g_b; one workgroup per 16 keys)
//   B5  dy2 = (softmax'(dP) rs) KA + (sum dS') (A b)                                   [16 x 512]
//   B6  dz1 = (g_b + LN2'(dy2)) Wtb1                 (writes g_c)
//   B7  dO = (g_c + TB1'(dz1)) Wo                    (writes g_d)
//   B8  self-attention core backward per (row, head): dq | dk | dv                     [16 x 1536]
//   B9  dy1 = [dq | dk | dv] Wqkv                                                      [16 x 512]
// and at the bottom  grad = (g_d + LN1'(dy1)) We  [16 x 128].
#pragma once
#include "rowtile.hpp"

#define RT_MFMA_F32 __builtin_amdgcn_mfma_f32_16x16x4f32

__device__ __forceinline__ float rt_gelu_grad(float y) {      // d/dy of the erf-form GELU (the formula of grad.hpp's EW_GELU_BWD)
  return 0.5f * (1.0f + erff(y * 0.70710678118654752440f)) + y * expf(-0.5f * y * y) * 0.39894228040143267794f;
}

enum { RT_BPRO_ROWS = 0, RT_BPRO_LN = 1, RT_BPRO_TB = 2 };
enum { RT_BEPI_F32 = 0, RT_BEPI_GELU = 1 };

struct RtBwdArgs {
  int L, tpr;
  int K;                 // contraction length: 512 (LN / TB prologues), 1024, 1536
  // prologue inputs
  const float* a;        // ROWS: the operand rows fp32 [M][K];  LN / TB: dy, the gradient at the LayerNorm's / the TimeBlock linear's output [M][512]
  const float* g;        // LN / TB: running gradient rows [M][512] (null: zero)
  const float* x;        // LN / TB: the LayerNorm's input rows [M][512]
  const float* gamma;    // LayerNorm weight
  const float* beta;     // TB: LayerNorm bias
  const float* ss;       // TB: (1 + scale | shift) of this time block at this step [1024]
  float* gout;           // LN / TB: the updated running gradient, written by workgroup x == 0
  // weights: fp32 [K][ldw], outputs n contiguous
  const float* w;
  long long ldw;
  // epilogue
  float* out;            // fp32 [M][ldo]
  long long ldo;
  const float* pre;      // GELU: pre-activations [M][ldo]
};

// LayerNorm backward (+ optional TimeBlock SiLU / modulation backward in front of it) of one prologue row, 16 columns per lane
// (c = 128 i + 4 plr .. + 3), 32 lanes per row.  In: dy (gradient at the LayerNorm output -- TB: at the linear's input side, i.e.
// d/d SiLU output), x, g.  Out: gn = g + d/dx.  Same formulas and reduction structure as layernorm_bwd_f32_kernel (grad.hpp).
template <bool TB>
__device__ __forceinline__ void rt_ln_bwd_row(float (&dy)[4][4], float (&xv)[4][4], const float (&gv)[4][4], const float (&gam)[4][4],
                                              const float (&bet)[4][4], const float (&s1)[4][4], const float (&sh)[4][4], float (&gn)[4][4]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += xv[i][e];
  const float mean = rt_row_sum<32>(s) * (1.0f / CFD_D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { xv[i][e] -= mean; ss += xv[i][e] * xv[i][e]; }
  const float rstd = 1.0f / sqrtf(rt_row_sum<32>(ss) * (1.0f / CFD_D) + 1e-5f);
  float a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xv[i][e] *= rstd;                                   // x hat
      float d = dy[i][e];
      if constexpr (TB) {                                 // through SiLU and the modulation: h = LN(x) (1 + scale) + shift
        const float h = (xv[i][e] * gam[i][e] + bet[i][e]) * s1[i][e] + sh[i][e];
        const float sg = rt_sigmoid(h);               // (the forward's rt_silu: one Newton step on v_rcp, v_exp, exponent clamped)
        d *= (sg * (1.0f + h * (1.0f - sg))) * s1[i][e];
      }
      d *= gam[i][e];
      dy[i][e] = d;                                       // dh
      a1 += d;
      a2 = fmaf(d, xv[i][e], a2);
    }
  const float m1 = rt_row_sum<32>(a1) * (1.0f / CFD_D), m2 = rt_row_sum<32>(a2) * (1.0f / CFD_D);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) gn[i][e] = gv[i][e] + rstd * (dy[i][e] - m1 - xv[i][e] * m2);
}

// loads of the LN / TB prologue for this lane's 16 columns of row `row` (global row index)
struct RtBwdRow {
  float dy[4][4], xv[4][4], gv[4][4], gam[4][4], bet[4][4], s1[4][4], sh[4][4];
};
template <bool TB>
__device__ __forceinline__ void rt_bwd_row_load(RtBwdRow& r, const float* a, const float* g, const float* x, const float* gamma, const float* beta,
                                                const float* sc, long long row, int plr) {
  // every load unconditional and in one basic block: a load inside `g ? ... : 0` is a block of its own, the loads behind it cannot move
  // in front of it, and the ones inside wait for themselves
  const float* gp = g ? g : a;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 128 * i + 4 * plr;
    const float4 d = *reinterpret_cast<const float4*>(a + row * CFD_D + c);
    const float4 xx = *reinterpret_cast<const float4*>(x + row * CFD_D + c);
    float4 gg = *reinterpret_cast<const float4*>(gp + row * CFD_D + c);   // (no running gradient yet: a valid row, zeroed by the select below)
    if (!g) gg = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    r.dy[i][0] = d.x; r.dy[i][1] = d.y; r.dy[i][2] = d.z; r.dy[i][3] = d.w;
    r.xv[i][0] = xx.x; r.xv[i][1] = xx.y; r.xv[i][2] = xx.z; r.xv[i][3] = xx.w;
    r.gv[i][0] = gg.x; r.gv[i][1] = gg.y; r.gv[i][2] = gg.z; r.gv[i][3] = gg.w;
    r.gam[i][0] = ga.x; r.gam[i][1] = ga.y; r.gam[i][2] = ga.z; r.gam[i][3] = ga.w;
    if constexpr (TB) {
      const float4 be = *reinterpret_cast<const float4*>(beta + c);
      const float4 a1 = *reinterpret_cast<const float4*>(sc + c), a2 = *reinterpret_cast<const float4*>(sc + CFD_D + c);
      r.bet[i][0] = be.x; r.bet[i][1] = be.y; r.bet[i][2] = be.z; r.bet[i][3] = be.w;
      r.s1[i][0] = a1.x; r.s1[i][1] = a1.y; r.s1[i][2] = a1.z; r.s1[i][3] = a1.w;
      r.sh[i][0] = a2.x; r.sh[i][1] = a2.y; r.sh[i][2] = a2.z; r.sh[i][3] = a2.w;
    }
  }
}

// float32 operand image of 16 rows in LDS: row stride K + 2 words (conflict-free ds_read_b32 of column k by 16 rows x 2 k's)
#define RT_BSTRIDE(K) ((K) + 2)

// sum over the waves (rt_reduce of rowtile.hpp with one block)
template <int NW>
__device__ __forceinline__ f32x4 rt_reduce1(char* red, int wid, int lane, f32x4 acc) {
  const f32x4 part[1] = {acc};
  return rt_reduce<NW, 1>(red, 1024, wid, lane, part);
}

// ------------------------------------------------------------------------------------------------
// out[16 tokens][16 outputs] = A[16][K] . W[K][n0 .. n0 + 15]      (B1, B2, B3, B6, B7, B9, the embedding's backward)
// grid (N / 16, tiles), 512 threads; dynamic LDS = 16 * (K + 2) * 4 + 8 KB
// ------------------------------------------------------------------------------------------------
template <int PRO, int EPI, int MAXSTEP>
__global__ void __launch_bounds__(512) rt_bwd_gemm_kernel(const RtBwdArgs a) {
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  // K is the kernel instance's (the host checks a.K == MAXSTEP * 32): with a run-time K every load of the unrolled loops below sat in
  // a conditional block of its own, next to its use, and waited for its own round trip -- 16 to 48 of them in a row
  constexpr int K = MAXSTEP * 32, RS = RT_BSTRIDE(K);
  char* red = smem + (size_t)16 * RS * 4;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tile = blockIdx.y, n0 = blockIdx.x * 16;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  const int pr = threadIdx.x >> 5, plr = threadIdx.x & 31;
  const long long prow = tok0 + min(pr, nq - 1);
  RT_T(t_in);

  // ---- loads: the prologue's rows (they lead to the barrier), then this wave's weight column, all requested before anything waits ----
  RtBwdRow r;
  constexpr int NR = PRO == RT_BPRO_ROWS ? MAXSTEP / 4 : 1;     // ROWS: K / 128 float4 per lane (12 for K = 1536)
  float rows[NR][4];
  if constexpr (PRO == RT_BPRO_ROWS) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const float4 d = *reinterpret_cast<const float4*>(a.a + prow * K + 128 * i + 4 * plr);
      rows[i][0] = d.x; rows[i][1] = d.y; rows[i][2] = d.z; rows[i][3] = d.w;
    }
  } else {
    rt_bwd_row_load<PRO == RT_BPRO_TB>(r, a.a, a.g, a.x, a.gamma, a.beta, a.ss, prow, plr);
  }
  const int kbase = wid * (K / NW) + q4;                        // MFMAs of this wave: k = wid * (K / 8) + 4 i + q4, i < MAXSTEP
  float wv[MAXSTEP];
  {
    const float* wp = a.w + (long long)kbase * a.ldw + n0 + l15;
#pragma unroll
    for (int i = 0; i < MAXSTEP; ++i) wv[i] = wp[(long long)(4 * i) * a.ldw];
  }
  __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler sinks the weight loads behind the prologue's waits: a second round trip)
  float4 ep_pre = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (EPI == RT_BEPI_GELU) {
    if (wid == 0) ep_pre = *reinterpret_cast<const float4*>(a.pre + (tok0 + min(l15, nq - 1)) * a.ldo + n0 + 4 * q4);
  }

  // ---- prologue -> float32 image ---------------------------------------------------------------------------
  if constexpr (PRO == RT_BPRO_ROWS) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      *reinterpret_cast<float2*>(img + pr * RS + 128 * i + 4 * plr) = make_float2(rows[i][0], rows[i][1]);
      *reinterpret_cast<float2*>(img + pr * RS + 128 * i + 4 * plr + 2) = make_float2(rows[i][2], rows[i][3]);
    }
  } else {
    float gn[4][4];
    rt_ln_bwd_row<PRO == RT_BPRO_TB>(r.dy, r.xv, r.gv, r.gam, r.bet, r.s1, r.sh, gn);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float2*>(img + pr * RS + 128 * i + 4 * plr) = make_float2(gn[i][0], gn[i][1]);
      *reinterpret_cast<float2*>(img + pr * RS + 128 * i + 4 * plr + 2) = make_float2(gn[i][2], gn[i][3]);
      if (blockIdx.x == 0 && pr < nq) *reinterpret_cast<float4*>(a.gout + (tok0 + pr) * CFD_D + 128 * i + 4 * plr) = make_float4(gn[i][0], gn[i][1], gn[i][2], gn[i][3]);
    }
  }
  __syncthreads();

  // ---- product ---------------------------------------------------------------------------------------------------
#if RT_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  RT_T(t_ops);
  struct StampAtExit { unsigned long long a, b; int id; __device__ ~StampAtExit() { RT_STAMP_OUT(id, a, b); } } stamp_{t_in, t_ops, 5000 + 100 * PRO + 10 * EPI + (MAXSTEP == 16 ? 0 : MAXSTEP == 32 ? 1 : 2)};
#endif
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = img + l15 * RS + kbase;
#pragma unroll
  for (int i = 0; i < MAXSTEP; ++i) acc = RT_MFMA_F32(wv[i], ap[4 * i], acc, 0, 0, 0);   // D[n][token]
  acc = rt_reduce1<NW>(red, wid, lane, acc);
  if (wid != 0 || l15 >= nq) return;
  float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
  if constexpr (EPI == RT_BEPI_GELU) {
    o.x *= rt_gelu_grad(ep_pre.x); o.y *= rt_gelu_grad(ep_pre.y); o.z *= rt_gelu_grad(ep_pre.z); o.w *= rt_gelu_grad(ep_pre.w);
  }
  *reinterpret_cast<float4*>(a.out + (tok0 + l15) * a.ldo + n0 + 4 * q4) = o;
}

// ------------------------------------------------------------------------------------------------
// B4: gradient at the cross-attention probabilities of 16 keys of one memory
//   dP[token][s] = (gn . (VA_s + VV b_t)) rs_s + d_att[token][s]  (tlsn only),   gn = g + TB2'(dz)  (written by workgroup 0)
// ------------------------------------------------------------------------------------------------
struct RtXBwdArgs {
  int L, tpr, nl, layer;
  // TB prologue (B4)
  const float* dz;
  const float* g;
  const float* x;
  const float* gamma;
  const float* beta;
  const float* ss;              // time block 2's (1 + scale | shift) at this step
  float* gout;
  // memories
  const char* K[CFD_NMEM];      // this layer's folded keys: SP [U_j * Sp_j][512]
  const char* VT[CFD_NMEM];     // this layer's folded values^T: SP [U_j][512][Sp_j]
  const float* kb[CFD_NMEM];    // A_l b_t of this layer at this step
  const float* vb[CFD_NMEM];    // VV_l b_t
  const int* map[CFD_NMEM];
  int use_inst;                 // as RtXArgs
  alignas(4) unsigned char inst[CFD_NMEM][RT_ARG_ROWS];
  int S[CFD_NMEM], Sp[CFD_NMEM], off[CFD_NMEM];
  int blk0[CFD_NMEM + 1];
  int Sp_tot;
  const float* sc;              // this layer's e_s = exp(score - cell maximum) [M][Sp_tot] (saved by the forward)
  const float* cst;             // this layer's cell statistics, float4 [M][Sp_tot / 32] (rowtile.hpp, RtXArgs::cst)
  const float* rsp;             // per-key scales [M][Sp_tot]
  const float* d_att;           // gradient at the tlsn probabilities [B][nl][L][S_2]
  float* dP;                    // [M][Sp_tot]
  int dp_from_datt;             // B5 of the top layer: dP is d_att alone (no B4 ran)
  float* dy;                    // B5 output [M][512]
};

template <int CFD_KI = 0>
__global__ void __launch_bounds__(512) rt_xbwd_dp_kernel(const RtXBwdArgs a) {
  constexpr int NW = 8, K = CFD_D, RS = RT_BSTRIDE(CFD_D), NSTEP = K / (4 * NW);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* img = reinterpret_cast<float*>(smem);
  char* red = smem + 16 * RS * 4;
  float* dvb = reinterpret_cast<float*>(red + NW * 1024);
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
  const int u = a.use_inst ? rt_inst_get(a.inst[j], b) : rt_sel(a.map, j)[b];
  const int Sp = rt_sel(a.Sp, j);
  const int pr = threadIdx.x >> 5, plr = threadIdx.x & 31;
  const long long prow = tok0 + min(pr, nq - 1);
  RT_T(t_in);
  RtBwdRow r;
  rt_bwd_row_load<true>(r, a.dz, a.g, a.x, a.gamma, a.beta, a.ss, prow, plr);
  const float* vbp = rt_sel(a.vb, j);
  float vbv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 q = *reinterpret_cast<const float4*>(vbp + 128 * i + 4 * plr);
    vbv[i][0] = q.x; vbv[i][1] = q.y; vbv[i][2] = q.z; vbv[i][3] = q.w;
  }
  // first operand: VA[s0 + l15][f] for f = wid * 64 + 4 i + q4, read from V^T [f][Sp] (split pairs, keys contiguous) -- kept as halves
  // until the product (a conversion next to its load waits for it), and requested before the prologue waits for its rows
  const int kbase = wid * (K / NW) + q4;
  sp_t whi[NSTEP], wlo[NSTEP];
  {
    const int sl = s0 + l15;
    const char* vp = rt_sel(a.VT, j) + ((size_t)u * CFD_D + kbase) * ((size_t)Sp * 4) + (size_t)(sl >> 5) * 128 + (sl & 31) * 2;
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      const char* p = vp + (size_t)(4 * i) * ((size_t)Sp * 4);
      whi[i] = *reinterpret_cast<const sp_t*>(p);
      wlo[i] = *reinterpret_cast<const sp_t*>(p + 64);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 e_rs = make_float4(0.f, 0.f, 0.f, 0.f), e_da = e_rs;
  const long long so = (tok0 + min(l15, nq - 1)) * a.Sp_tot + rt_sel(a.off, j) + s0 + 4 * q4;
  if (wid == 0) {
    e_rs = *reinterpret_cast<const float4*>(a.rsp + so);
    if (j == 2) {
      const int S = a.S[2];
      const float* dp = a.d_att + (((long long)b * a.nl + a.layer) * a.L + q0 + min(l15, nq - 1)) * S;
      const int k0 = s0 + 4 * q4;
      // (loads at clamped indices, zeroed by selects: four conditional loads are four blocks that each wait for their own round trip,
      //  and the workgroups of this memory then ran 2 us behind the others of the launch)
      const float d0 = dp[min(k0, S - 1)], d1 = dp[min(k0 + 1, S - 1)], d2 = dp[min(k0 + 2, S - 1)], d3 = dp[min(k0 + 3, S - 1)];
      e_da.x = k0 < S ? d0 : 0.f; e_da.y = k0 + 1 < S ? d1 : 0.f; e_da.z = k0 + 2 < S ? d2 : 0.f; e_da.w = k0 + 3 < S ? d3 : 0.f;
    }
  }
  float gn[4][4];
  rt_ln_bwd_row<true>(r.dy, r.xv, r.gv, r.gam, r.bet, r.s1, r.sh, gn);
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<float2*>(img + pr * RS + 128 * i + 4 * plr) = make_float2(gn[i][0], gn[i][1]);
    *reinterpret_cast<float2*>(img + pr * RS + 128 * i + 4 * plr + 2) = make_float2(gn[i][2], gn[i][3]);
    if (blockIdx.x == 0 && pr < nq) *reinterpret_cast<float4*>(a.gout + (tok0 + pr) * CFD_D + 128 * i + 4 * plr) = make_float4(gn[i][0], gn[i][1], gn[i][2], gn[i][3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) dot = fmaf(gn[i][e], vbv[i][e], dot);
  }
  dot = rt_row_sum<32>(dot);
  if (plr == 0) dvb[pr] = dot;
  __syncthreads();
#if RT_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  RT_T(t_ops);
  struct StampAtExit { unsigned long long a, b; int id; __device__ ~StampAtExit() { RT_STAMP_OUT(id, a, b); } } stamp_{t_in, t_ops, 6000};
#endif
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = img + l15 * RS + kbase;
#pragma unroll
  for (int i = 0; i < NSTEP; ++i) acc = RT_MFMA_F32((float)whi[i] + (float)wlo[i], ap[4 * i], acc, 0, 0, 0);   // D[key][token]
  acc = rt_reduce1<NW>(red, wid, lane, acc);
  if (wid != 0 || l15 >= nq) return;
  const float d = dvb[l15];
  *reinterpret_cast<float4*>(a.dP + so) = make_float4((acc[0] + d) * e_rs.x + e_da.x, (acc[1] + d) * e_rs.y + e_da.y, (acc[2] + d) * e_rs.z + e_da.z,
                                                      (acc[3] + d) * e_rs.w + e_da.w);
}

// ------------------------------------------------------------------------------------------------
// B5: softmax backward per memory, dS' = dS rs, and the gradient at the LayerNorm2 output
//   dy[token][f] = sum_j ( sum_s dS'_s KA_j[s][f] + (sum_s dS'_s) (A b_t)_j[f] )
// grid (32, tiles); dynamic LDS = 16 * (Sp_tot + 2) * 4 (dS image) + 8 KB (reduction) + 512 (sums) + 8 KB (cell statistics) + 1 KB (per-memory statistics)
// ------------------------------------------------------------------------------------------------
template <int MAXKEYS>
__global__ void __launch_bounds__(512) rt_xbwd_dy_kernel(const RtXBwdArgs a) {
  constexpr int NW = 8, LPR = 32;
  constexpr int MAXC = MAXKEYS / 4 / LPR;                        // 4-key chunks per lane
  constexpr int MAXSTEP = MAXKEYS / (4 * NW);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KS = a.Sp_tot, RS = RT_BSTRIDE(KS);
  float* img = reinterpret_cast<float*>(smem);
  char* red = smem + (size_t)16 * RS * 4;
  float* dcq = reinterpret_cast<float*>(red + NW * 1024);   // [16 tokens][8]
  float4* cst = reinterpret_cast<float4*>(red + NW * 1024 + 512);   // [16 tokens][32 cells]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tile = blockIdx.y, f0 = blockIdx.x * 16;
  const int b = tile / a.tpr, q0 = (tile - b * a.tpr) * 16, nq = min(16, a.L - q0);
  const long long tok0 = (long long)b * a.L + q0;
  RT_T(t_in);
  // this wave's operand: KA[key][f0 + l15] for its keys kbase + 4 i, read from the folded keys (split pairs) -- requested before the
  // prologue, so that the loads run under the softmax backward.  The row of concatenated key k of memory j starts at kb_j + k * 2048
  // with kb_j wave-uniform (the instance of the batch row), so a load's address costs one select chain over four boundaries and one
  // multiply-add.  (With the memory, instance and row length selected per lane and step -- the first version -- ISSUING the loads was
  // 9.3 us of this kernel's 14 us in front of its first MFMA: ~100 vector instructions for each of the 32 unrolled steps.)
  static_assert(CFD_NMEM == 5, "five named instance indices");
  int u0, u1, u2, u3, u4;
  if (a.use_inst) { u0 = rt_inst_get(a.inst[0], b); u1 = rt_inst_get(a.inst[1], b); u2 = rt_inst_get(a.inst[2], b); u3 = rt_inst_get(a.inst[3], b); u4 = rt_inst_get(a.inst[4], b); }
  else { u0 = a.map[0][b]; u1 = a.map[1][b]; u2 = a.map[2][b]; u3 = a.map[3][b]; u4 = a.map[4][b]; }
  const long long ROWB = CFD_D * 4;
  const char* const kb0 = a.K[0] + ((long long)u0 * a.Sp[0] - a.off[0]) * ROWB;
  const char* const kb1 = a.K[1] + ((long long)u1 * a.Sp[1] - a.off[1]) * ROWB;
  const char* const kb2 = a.K[2] + ((long long)u2 * a.Sp[2] - a.off[2]) * ROWB;
  const char* const kb3 = a.K[3] + ((long long)u3 * a.Sp[3] - a.off[3]) * ROWB;
  const char* const kb4 = a.K[4] + ((long long)u4 * a.Sp[4] - a.off[4]) * ROWB;
  int o1 = a.off[1], o2 = a.off[2], o3 = a.off[3], o4 = a.off[4];
  int sp0 = a.Sp[0], sp1 = a.Sp[1], sp2 = a.Sp[2], sp3 = a.Sp[3], sp4 = a.Sp[4];
  RT_PIN_S(o1); RT_PIN_S(o2); RT_PIN_S(o3); RT_PIN_S(o4);      // (per-lane choices below stay selects: rowtile.hpp, RT_PIN_S)
  RT_PIN_S(sp0); RT_PIN_S(sp1); RT_PIN_S(sp2); RT_PIN_S(sp3); RT_PIN_S(sp4);
  const int nstep = KS / (4 * NW);
  const int kbase = wid * (KS / NW) + q4;
  const int fo = ((f0 + l15) >> 5) * 128 + ((f0 + l15) & 31) * 2;   // this lane's feature inside an SP key row
  // prologue loads: scores, dP (or d_att), rs of this lane's 4-key chunks plr + 32 n (16-byte loads contiguous across the row's lanes)
  const int pr = threadIdx.x / LPR, plr = threadIdx.x % LPR;
  const long long prow = tok0 + min(pr, nq - 1);
  float s[MAXC][4], dp[MAXC][4], rsv[MAXC][4];
  int cj[MAXC];
#pragma unroll
  for (int n = 0; n < MAXC; ++n) {
    const int c0r = (plr + LPR * n) * 4, c0 = min(c0r, KS - 4);   // (chunks past the last key load the last chunk and are ignored: cj = -1)
    const int j = c0 >= o4 ? 4 : c0 >= o3 ? 3 : c0 >= o2 ? 2 : c0 >= o1 ? 1 : 0;
    cj[n] = c0r < KS ? j : -1;
    const float4 p0 = *reinterpret_cast<const float4*>(a.sc + prow * KS + c0);
    const float4 r0 = *reinterpret_cast<const float4*>(a.rsp + prow * KS + c0);
    s[n][0] = p0.x; s[n][1] = p0.y; s[n][2] = p0.z; s[n][3] = p0.w;
    rsv[n][0] = r0.x; rsv[n][1] = r0.y; rsv[n][2] = r0.z; rsv[n][3] = r0.w;
  }
  if (!a.dp_from_datt) {        // (one block of loads per case: the branch is outside the unrolled loop)
#pragma unroll
    for (int n = 0; n < MAXC; ++n) {
      const float4 d0 = *reinterpret_cast<const float4*>(a.dP + prow * KS + min((plr + LPR * n) * 4, KS - 4));
      dp[n][0] = d0.x; dp[n][1] = d0.y; dp[n][2] = d0.z; dp[n][3] = d0.w;
    }
  } else {                      // the top layer: dP is d_att alone (tlsn keys), element by element (S_2 need not be a multiple of 4)
    const int S2 = a.S[2];
    const float* da = a.d_att + (((long long)b * a.nl + a.layer) * a.L + q0 + min(pr, nq - 1)) * S2;
#pragma unroll
    for (int n = 0; n < MAXC; ++n) {
      const int k0 = min((plr + LPR * n) * 4, KS - 4) - o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) dp[n][e] = (cj[n] == 2 && k0 + e < S2) ? da[min(max(k0 + e, 0), S2 - 1)] : 0.f;
    }
  }
  const int fcol = f0 + 4 * q4;
  float4 ep_kb[CFD_NMEM];
  if (wid == 0) {
#pragma unroll
    for (int j = 0; j < CFD_NMEM; ++j) ep_kb[j] = *reinterpret_cast<const float4*>(a.kb[j] + fcol);
  }
  // (unconditional loads at clamped keys, converted behind the prologue: a load inside `if (i < nstep)` is a block of its own together
  //  with its conversion, and each step then waits for its own round trip)
  sp_t whi[MAXSTEP], wlo[MAXSTEP];
#pragma unroll
  for (int i = 0; i < MAXSTEP; ++i) {
    const int key = min(kbase + 4 * i, KS - 1);
    const char* base = key >= o4 ? kb4 : key >= o3 ? kb3 : key >= o2 ? kb2 : key >= o1 ? kb1 : kb0;
    const char* p = base + (long long)key * ROWB + fo;
    whi[i] = *reinterpret_cast<const sp_t*>(p);
    wlo[i] = *reinterpret_cast<const sp_t*>(p + 64);
  }
  __builtin_amdgcn_sched_barrier(0);
  RT_T(t_iss);
  if (plr < KS / 32) cst[pr * 32 + plr] = reinterpret_cast<const float4*>(a.cst)[prow * (KS / 32) + plr];
  __syncthreads();
  RT_T(t_s1);
  // the probabilities from the saved e_s and cell statistics (as rt_xpv_kernel makes them: per (token, memory) the maximum and
  // 1 / sum once, by 80 threads), then the softmax backward per memory
  float2* seg = reinterpret_cast<float2*>(cst + 16 * 32);   // [16 tokens][8]
  if (threadIdx.x < 16 * CFD_NMEM) {
    const int r = threadIdx.x / CFD_NMEM, j = threadIdx.x - r * CFD_NMEM;
    const float4* cr = cst + r * 32;
    const int offj = rt_pick5(j, 0, o1, o2, o3, o4), cb = offj >> 5, ce = (offj + rt_pick5(j, sp0, sp1, sp2, sp3, sp4)) >> 5;
    float M = -INFINITY, l = 0.f;
    for (int k = cb; k < ce; ++k) M = fmaxf(M, cr[k].x);
    for (int k = cb; k < ce; ++k) l = fmaf(cr[k].y, __expf(cr[k].x - M), l);
    seg[r * 8 + j] = make_float2(M, rt_rcp(l));
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < MAXC; ++n)
    if (cj[n] >= 0) {
      const int c0 = (plr + LPR * n) * 4;
      const float2 sg = seg[pr * 8 + cj[n]];
      const float f = __expf(cst[pr * 32 + (c0 >> 5)].x - sg.x) * sg.y;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[n][e] *= f;
    }
  RT_T(t_cs);
#pragma unroll 1
  for (int j = 0; j < CFD_NMEM; ++j) {
    float dot = 0.f;
#pragma unroll
    for (int n = 0; n < MAXC; ++n)
      if (cj[n] == j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dot = fmaf(dp[n][e], s[n][e], dot);
      }
    dot = rt_row_sum<LPR>(dot);
    float wsum = 0.f;
#pragma unroll
    for (int n = 0; n < MAXC; ++n)
      if (cj[n] == j) {
        const int c0 = (plr + LPR * n) * 4;
        float ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ds[e] = s[n][e] * (dp[n][e] - dot) * rsv[n][e];   // dS' = p (dP - sum dP p) rs
          wsum += ds[e];
        }
        *reinterpret_cast<float2*>(img + pr * RS + c0) = make_float2(ds[0], ds[1]);
        *reinterpret_cast<float2*>(img + pr * RS + c0 + 2) = make_float2(ds[2], ds[3]);
      }
    wsum = rt_row_sum<LPR>(wsum);
    if (plr == 0) dcq[pr * 8 + j] = wsum;
  }
  __syncthreads();
#if RT_STAMP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  RT_T(t_ops);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // first record: entry -> loads issued, first barrier passed, cell scales applied
    const unsigned k_ = atomicAdd(&g_rt_seq, 1u) & 4095u;
    g_rt_ring[4 * k_] = 6101; g_rt_ring[4 * k_ + 1] = t_iss - t_in; g_rt_ring[4 * k_ + 2] = t_s1 - t_in; g_rt_ring[4 * k_ + 3] = t_cs - t_in;
  }
  struct StampAtExit { unsigned long long a, b; int id; __device__ ~StampAtExit() { RT_STAMP_OUT(id, a, b); } } stamp_{t_in, t_ops, 6100};
#endif
  // product over the keys: this wave's range of Sp_tot / 8 keys
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = img + l15 * RS + kbase;
#pragma unroll
  for (int i = 0; i < MAXSTEP; ++i) {
    const float wv = i < nstep ? (float)whi[i] + (float)wlo[i] : 0.f;   // (a select, not a branch: see the loads; steps past the last
    acc = RT_MFMA_F32(wv, ap[4 * min(i, nstep - 1)], acc, 0, 0, 0);     //  key multiply a valid image element by zero)  D[feature][token]
  }
  acc = rt_reduce1<NW>(red, wid, lane, acc);
  if (wid != 0 || l15 >= nq) return;
  float o[4] = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
  for (int j = 0; j < CFD_NMEM; ++j) {
    const float w = dcq[l15 * 8 + j];
    o[0] += w * ep_kb[j].x; o[1] += w * ep_kb[j].y; o[2] += w * ep_kb[j].z; o[3] += w * ep_kb[j].w;
  }
  *reinterpret_cast<float4*>(a.dy + (tok0 + l15) * CFD_D + fcol) = make_float4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------
// B8: self-attention core backward of one (head, batch row), L <= 32 tokens, plain float32 out of LDS (a few hundred kFLOP):
//   P = softmax(Q K^T) (recomputed), dP = dO V^T, dS = P (dP - rowsum(dP P)), dQ = dS K, dK = dS^T Q, dV = P^T dO
// q is the pre-scaled query (the scale lives in the weights, so dQ is the gradient at the scaled projection's output).
// ------------------------------------------------------------------------------------------------
struct RtSelfBwdArgs {
  const char* qk;     // SP [M][1024] saved by the forward
  const char* vt;     // SP [Be][512][32]
  const float* dO;    // [M][512]: gradient at the attention output (before the out-projection)
  float* dqkv;        // [M][1536]: dq | dk | dv
  int L;
  float qscale;       // 1 / sqrt(head_dim): dq is handed on as the gradient at the UNSCALED query projection (the forward's weights carry the scale)
};

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) rt_selfattn_bwd_kernel(const RtSelfBwdArgs a) {
  constexpr int HD = CFD_HD, RSD = HD + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Q = reinterpret_cast<float*>(smem);          // [32][129]
  float* Kk = Q + RT_MAX_L * RSD;
  float* V = Kk + RT_MAX_L * RSD;
  float* dO = V + RT_MAX_L * RSD;
  float* P = dO + RT_MAX_L * RSD;                      // [32][33]
  float* dS = P + RT_MAX_L * (RT_MAX_L + 1);
  const int h = blockIdx.x, b = blockIdx.y, L = a.L, tid = threadIdx.x;
  const long long tok0 = (long long)b * L;
  RT_T(t_in);
  // operands into LDS as float32, 16-byte loads: 8 consecutive features of a q / k row (hi and lo chunk), 8 consecutive keys of a
  // V^T feature row, 4 consecutive features of a dO row  (element by element this phase was most of the kernel's 15 us)
  for (int e = tid; e < L * (HD / 8); e += 256) {
    const int r = e / (HD / 8), c8 = (e - r * (HD / 8)) * 8, cq = h * HD + c8;
    const char* row = a.qk + (size_t)(tok0 + r) * 4096 + (cq & 31) * 2;
    const spx8 qh = *reinterpret_cast<const spx8*>(row + (cq >> 5) * 128), ql = *reinterpret_cast<const spx8*>(row + (cq >> 5) * 128 + 64);
    const spx8 kh = *reinterpret_cast<const spx8*>(row + (16 + (cq >> 5)) * 128), kl = *reinterpret_cast<const spx8*>(row + (16 + (cq >> 5)) * 128 + 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      Q[r * RSD + c8 + i] = (float)qh[i] + (float)ql[i];
      Kk[r * RSD + c8 + i] = (float)kh[i] + (float)kl[i];
    }
  }
  for (int e = tid; e < HD * (RT_MAX_L / 8); e += 256) {
    const int d = e / (RT_MAX_L / 8), k8 = (e - d * (RT_MAX_L / 8)) * 8;
    if (k8 < L) {
      const char* vr = a.vt + ((size_t)b * CFD_D + h * HD + d) * (RT_MAX_L * 4) + k8 * 2;
      const spx8 vh = *reinterpret_cast<const spx8*>(vr), vl = *reinterpret_cast<const spx8*>(vr + 64);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (k8 + i < L) V[(k8 + i) * RSD + d] = (float)vh[i] + (float)vl[i];
    }
  }
  for (int e = tid; e < L * (HD / 4); e += 256) {
    const int r = e / (HD / 4), c4 = (e - r * (HD / 4)) * 4;
    const float4 q = *reinterpret_cast<const float4*>(a.dO + (tok0 + r) * CFD_D + h * HD + c4);
    dO[r * RSD + c4] = q.x; dO[r * RSD + c4 + 1] = q.y; dO[r * RSD + c4 + 2] = q.z; dO[r * RSD + c4 + 3] = q.w;
  }
  __syncthreads();
#if RT_STAMP
  RT_T(t_ops);
  struct StampAtExit { unsigned long long a, b; int id; __device__ ~StampAtExit() { RT_STAMP_OUT(id, a, b); } } stamp_{t_in, t_ops, 6200};
#endif
  // scores and dP on the float32 matrix core: wave w takes (query tile w & 1, key tile w >> 1); D[key][query], 32 steps over d
  const int lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q4 = lane >> 4;
  const int nqt = (L + 15) / 16;
  {
    const int qt = wid & 1, kt = wid >> 1;
    if (qt < nqt && kt < nqt) {
      const float* qrow = Q + min(qt * 16 + l15, L - 1) * RSD + q4;
      const float* orow = dO + min(qt * 16 + l15, L - 1) * RSD + q4;
      const float* krow = Kk + min(kt * 16 + l15, L - 1) * RSD + q4;
      const float* vrow = V + min(kt * 16 + l15, L - 1) * RSD + q4;
      f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, da = sa;
#pragma unroll 8
      for (int st = 0; st < HD / 4; ++st) {
        sa = RT_MFMA_F32(krow[4 * st], qrow[4 * st], sa, 0, 0, 0);
        da = RT_MFMA_F32(vrow[4 * st], orow[4 * st], da, 0, 0, 0);
      }
      const int q = qt * 16 + l15;
      if (q < L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = kt * 16 + 4 * q4 + r;
          if (k < L) { P[q * (RT_MAX_L + 1) + k] = sa[r]; dS[q * (RT_MAX_L + 1) + k] = da[r]; }
        }
      }
    }
  }
  __syncthreads();
  // softmax of every query row and its backward, in place: 16 lanes per row (keys lane16 and lane16 + 16), 16 rows per pass
  for (int r0 = 0; r0 < L; r0 += 16) {
    const int q = r0 + (tid >> 4), k0 = tid & 15, k1 = k0 + 16;
    const bool qok = q < L, ok0 = qok && k0 < L, ok1 = qok && k1 < L;
    float* pr = P + min(q, L - 1) * (RT_MAX_L + 1);
    float* dr = dS + min(q, L - 1) * (RT_MAX_L + 1);
    const float s0 = ok0 ? pr[k0] : -INFINITY, s1 = ok1 ? pr[k1] : -INFINITY;
    const float mx = rt_row_max<16>(fmaxf(s0, s1));
    const float e0 = ok0 ? __expf(s0 - mx) : 0.f, e1 = ok1 ? __expf(s1 - mx) : 0.f;   // (as rt_selfattn_kernel)
    const float sum = rt_row_sum<16>(e0 + e1);
    const float p0 = e0 / sum, p1 = e1 / sum;
    const float d0 = ok0 ? dr[k0] : 0.f, d1 = ok1 ? dr[k1] : 0.f;
    const float dot = rt_row_sum<16>(fmaf(d0, p0, d1 * p1));
    if (ok0) { pr[k0] = p0; dr[k0] = p0 * (d0 - dot); }
    if (ok1) { pr[k1] = p1; dr[k1] = p1 * (d1 - dot); }
  }
  __syncthreads();
  // dQ = dS K, dK = dS^T Q, dV = P^T dO: output tiles [16 features][16 rows] (lane: 4 consecutive features of row l15), contraction
  // over the other row axis (L / 4 steps); 3 products x nqt row tiles x 8 feature tiles dealt to the 4 waves
  const int nsteps = (L + 3) / 4;
  for (int item = wid; item < 3 * nqt * 8; item += 4) {
    const int prod = item / (nqt * 8), rt = (item / 8) % nqt, ft = item & 7;
    const float* X = prod == 0 ? Kk : (prod == 1 ? Q : dO);           // first operand: X[kk][feature]
    const float* Y = prod == 2 ? P : dS;                                // second operand: dQ: dS[row][kk]; dK, dV: Y[kk][row] (transposed use)
    const int row = rt * 16 + l15, rowc = min(row, L - 1);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // (LDS reads at clamped indices, zeroed by selects, in one block: a read inside a condition waits for itself before the next is issued)
    const float* xp = X + ft * 16 + l15;
    const float* yp = prod == 0 ? Y + rowc * (RT_MAX_L + 1) : Y + rowc;
    const int ystep = prod == 0 ? 1 : RT_MAX_L + 1;
    float xv[RT_MAX_L / 4], yv[RT_MAX_L / 4];
#pragma unroll
    for (int st = 0; st < RT_MAX_L / 4; ++st) {
      const int kk = min(4 * st + q4, L - 1);
      xv[st] = xp[kk * RSD];
      yv[st] = yp[kk * ystep];
    }
#pragma unroll
    for (int st = 0; st < RT_MAX_L / 4; ++st) {
      const bool ok = 4 * st + q4 < L;
      if (st < nsteps) acc = RT_MFMA_F32(ok ? xv[st] : 0.f, (ok && row < L) ? yv[st] : 0.f, acc, 0, 0, 0);
    }
    if (row < L) {
      const float sc = prod == 0 ? a.qscale : 1.0f;
      float* o = a.dqkv + (tok0 + row) * (3 * CFD_D) + prod * CFD_D + h * HD + ft * 16 + 4 * q4;
      *reinterpret_cast<float4*>(o) = make_float4(acc[0] * sc, acc[1] * sc, acc[2] * sc, acc[3] * sc);
    }
  }
}
