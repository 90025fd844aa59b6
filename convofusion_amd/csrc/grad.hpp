// float32 kernels of the word-excitation-guidance (WEG) gradient path: d(loss)/d(latents) through the denoiser
// (reference: torch autograd over Denoiser.forward, convofusion/models/modeltype/convofusion.py:437-496 and
// convofusion/models/tools/word_excitation_guidance.py:11-81).  WEG runs on the text-only guidance chunk with
// test batch size 1 (word_excitation_guidance.py:25), i.e. 16 tokens x 512 features: launch-latency work, so the
// kernels are general (arbitrary strides: every transpose of the backward pass is a view) and exact rather than tuned.
#pragma once
#include "cfd_common.hpp"

// element (z1, z2, r, c) = p[z1 * b1 + z2 * b2 + r * rs + c * cs]   (mirrors cfd_mat in include/cfdenoise.h)
struct MatView {
  const float* p;
  long long rs, cs, b1, b2;
};

// C(z; m, n) = alpha * sum_k A(z; m, k) B(z; k, n) + bias[n] (+ C(z; m, n) when accumulate); ascending-k FMA chain per
// output.  64 x 64 output tile per 256-thread workgroup (4 x 4 per thread), K-step 16 through LDS.
__device__ __forceinline__ float act_f32(float x, int act) {   // 0 none, 1 nn.SiLU, 2 nn.GELU (erf form)
  return act == 1 ? x / (1.0f + expf(-x)) : (act == 2 ? gelu_f(x) : x);
}

// + resid(z; m, n) (same strides as C; the residual connection) ; a_act: activation applied to A's elements as they are read
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) gemm_f32_kernel(MatView A, MatView B, float* C, long long c_rs, long long c_cs, long long c_b1,
                                                       long long c_b2, int M, int N, int K, int nb2, const float* bias, float alpha,
                                                       int accumulate, const float* resid, int a_act) {
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int z1 = blockIdx.z / nb2, z2 = blockIdx.z % nb2;
  const float* a = A.p + z1 * A.b1 + z2 * A.b2;
  const float* b = B.p + z1 * B.b1 + z2 * B.b2;
  float* c = C + z1 * c_b1 + z2 * c_b2;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const bool a_k_fast = A.cs == 1, b_n_fast = B.cs == 1;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      int mm, kk;
      if (a_k_fast) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
      As[kk][mm] = (m0 + mm < M && k0 + kk < K) ? act_f32(a[(long long)(m0 + mm) * A.rs + (long long)(k0 + kk) * A.cs], a_act) : 0.f;
      int nn, kb;
      if (b_n_fast) { nn = e & 63; kb = e >> 6; } else { kb = e & 15; nn = e >> 4; }
      Bs[kb][nn] = (n0 + nn < N && k0 + kb < K) ? b[(long long)(k0 + kb) * B.rs + (long long)(n0 + nn) * B.cs] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = alpha * acc[i][j];
      if (bias) v += bias[n];
      const long long o = (long long)m * c_rs + (long long)n * c_cs;
      if (resid) v += resid[z1 * c_b1 + z2 * c_b2 + o];
      if (accumulate) v += c[o];
      c[o] = v;
    }
  }
}

// The same product for few output rows (the WEG problem: 16 latent tokens, <= a few hundred memory tokens), where the
// tiled kernel above would run 8 workgroups of 32 dependent k-steps: one workgroup = 16 rows x 16 columns, its 256
// threads = 16 columns x 16 k-slices (thread (tn, tk) accumulates k = tk, tk + 16, ... for all 16 rows; A goes through
// LDS in k-chunks of 512, B streams from global memory once), partial sums meet in LDS and are added in ascending tk.
// N / 16 x M / 16 workgroups instead of N / 64 x M / 64, and one memory round trip per k-chunk.
#define R16_KC 512
struct GemmSeg {   // one more (A, B, K) term of the same output: C = alpha * sum_segments A_s B_s + ...
  MatView A, B;
  int K;
};
template <bool MULTI>
__device__ __forceinline__ void gemm_f32_rows16_tile(const MatView& A0, const MatView& B0, float* C, long long c_rs, long long c_cs, long long c_b1,
                                                     long long c_b2, int M, int N, int K0, int nb2, const float* bias, float alpha, int accumulate,
                                                     const float* resid, int a_act, int bx, int by, int bz, const GemmSeg* more = nullptr,
                                                     int n_more = 0) {
  __shared__ __attribute__((aligned(16))) float As[R16_KC][20];   // [k][row], rows padded to 20 (16-byte aligned float4 reads)
  __shared__ float red[16][16][17];                                // [tk][tn][row]
  const int z1 = bz / nb2, z2 = bz % nb2;
  float* c = C + z1 * c_b1 + z2 * c_b2;
  const int m0 = by * 16, n0 = bx * 16;
  const int tid = threadIdx.x;
  // the thread layout follows the first segment's operand orientation (all segments of a launch share it)
  const bool b_n_fast = B0.cs == 1, a_k_fast = A0.cs == 1;
  const int tn = b_n_fast ? (tid & 15) : (tid >> 4), tk = b_n_fast ? (tid >> 4) : (tid & 15);
  const int n = n0 + tn;
  const bool n_ok = n < N;
  float acc[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[m] = 0.f;
  // the epilogue's own operands are requested now, so they are not a second memory round trip after the k-loop
  const int om = tid >> 4, on = tid & 15;
  const bool o_ok = m0 + om < M && n0 + on < N;
  const long long o = (long long)(m0 + om) * c_rs + (long long)(n0 + on) * c_cs;
  const float e_bias = (o_ok && bias) ? bias[n0 + on] : 0.f;
  const float e_res = (o_ok && resid) ? resid[z1 * c_b1 + z2 * c_b2 + o] : 0.f;
  const float e_old = (o_ok && accumulate) ? c[o] : 0.f;
  bool first = true;
  for (int sgi = 0; sgi <= (MULTI ? n_more : 0); ++sgi) {
  MatView A = A0, B = B0;
  int K = K0;
  if (MULTI && sgi) {
    A = more[sgi - 1].A;
    B = more[sgi - 1].B;
    K = more[sgi - 1].K;
  }
  const float* a = A.p + z1 * A.b1 + z2 * A.b2;
  const float* b = B.p + z1 * B.b1 + z2 * B.b2;
  // This is latency-bound work (one workgroup owns a 512-deep dependent chain): every global load of a k-chunk -- the
  // thread's 32 B values and its 32 A values -- is issued before anything waits, so a chunk costs one memory round trip.
  const int am = a_k_fast ? (tid >> 4) : (tid & 15), ak = a_k_fast ? (tid & 15) : (tid >> 4);   // A element (am, ak + 16 i)
  const bool am_ok = m0 + am < M;
  for (int k0 = 0; k0 < K; k0 += R16_KC) {
    const int kc = min(R16_KC, K - k0);
    const float* bp = b + (long long)k0 * B.rs + (long long)n * B.cs;
    const float* ap = a + (long long)(m0 + am) * A.rs + (long long)k0 * A.cs;
    float bv[R16_KC / 16], av[R16_KC / 16];
#pragma unroll
    for (int i = 0; i < R16_KC / 16; ++i) {
      const int kb = tk + 16 * i, ka = ak + 16 * i;
      bv[i] = (n_ok && kb < kc) ? bp[(long long)kb * B.rs] : 0.f;
      av[i] = (am_ok && ka < kc) ? ap[(long long)ka * A.cs] : 0.f;
    }
    if (a_act) {
#pragma unroll
      for (int i = 0; i < R16_KC / 16; ++i) av[i] = (am_ok && ak + 16 * i < kc) ? act_f32(av[i], a_act) : 0.f;
    }
    if (!first) __syncthreads();      // the previous chunk's LDS reads are done
    first = false;
#pragma unroll
    for (int i = 0; i < R16_KC / 16; ++i) As[ak + 16 * i][am] = av[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R16_KC / 16; ++i) {
      const int kk = tk + 16 * i;
      if (kk < kc) {
        const float4* ar = reinterpret_cast<const float4*>(&As[kk][0]);
        const float4 a0 = ar[0], a1 = ar[1], a2 = ar[2], a3 = ar[3];
        const float w = bv[i];
        acc[0] = fmaf(a0.x, w, acc[0]); acc[1] = fmaf(a0.y, w, acc[1]); acc[2] = fmaf(a0.z, w, acc[2]); acc[3] = fmaf(a0.w, w, acc[3]);
        acc[4] = fmaf(a1.x, w, acc[4]); acc[5] = fmaf(a1.y, w, acc[5]); acc[6] = fmaf(a1.z, w, acc[6]); acc[7] = fmaf(a1.w, w, acc[7]);
        acc[8] = fmaf(a2.x, w, acc[8]); acc[9] = fmaf(a2.y, w, acc[9]); acc[10] = fmaf(a2.z, w, acc[10]); acc[11] = fmaf(a2.w, w, acc[11]);
        acc[12] = fmaf(a3.x, w, acc[12]); acc[13] = fmaf(a3.y, w, acc[13]); acc[14] = fmaf(a3.z, w, acc[14]); acc[15] = fmaf(a3.w, w, acc[15]);
      }
    }
  }
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) red[tk][tn][m] = acc[m];
  __syncthreads();
  if (o_ok) {
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) v += red[t][on][om];
    v *= alpha;
    if (bias) v += e_bias;
    if (resid) v += e_res;
    if (accumulate) v += e_old;
    c[o] = v;
  }
}

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) gemm_f32_rows16_kernel(MatView A, MatView B, float* C, long long c_rs, long long c_cs, long long c_b1,
                                                              long long c_b2, int M, int N, int K, int nb2, const float* bias, float alpha,
                                                              int accumulate, const float* resid, int a_act) {
  gemm_f32_rows16_tile<false>(A, B, C, c_rs, c_cs, c_b1, c_b2, M, N, K, nb2, bias, alpha, accumulate, resid, a_act, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Up to five independent products in one launch (the five cross-attentions of a layer differ in weights and memory
// length, not in structure): blockIdx.z walks the groups' batch entries, a workgroup outside its group's extent leaves.
#define GEMM_MAX_GROUPS 5
struct GemmGroup {
  MatView A, B;
  float* C;
  long long c_rs, c_cs, c_b1, c_b2;
  int M, N, K, nb1, nb2;
  const float* bias;
  const float* resid;
  float alpha;
  int accumulate, a_act;
};
struct GemmGroups {
  GemmGroup g[GEMM_MAX_GROUPS];
  int n;
  int zoff[GEMM_MAX_GROUPS + 1];
};
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) gemm_f32_rows16_grouped_kernel(const GemmGroups gs) {
  int gi = 0;
  while (gi + 1 < gs.n && (int)blockIdx.z >= gs.zoff[gi + 1]) ++gi;
  const GemmGroup& g = gs.g[gi];
  if ((int)blockIdx.x * 16 >= g.N || (int)blockIdx.y * 16 >= g.M) return;
  gemm_f32_rows16_tile<false>(g.A, g.B, g.C, g.c_rs, g.c_cs, g.c_b1, g.c_b2, g.M, g.N, g.K, g.nb2, g.bias, g.alpha, g.accumulate, g.resid, g.a_act,
                              blockIdx.x, blockIdx.y, blockIdx.z - gs.zoff[gi]);
}
static inline void launch_gemm_f32_grouped(hipStream_t st, GemmGroups& gs) {
  int mx = 0, nx = 0;
  gs.zoff[0] = 0;
  for (int i = 0; i < gs.n; ++i) {
    mx = gs.g[i].M > mx ? gs.g[i].M : mx;
    nx = gs.g[i].N > nx ? gs.g[i].N : nx;
    gs.zoff[i + 1] = gs.zoff[i] + gs.g[i].nb1 * gs.g[i].nb2;
  }
  hipLaunchKernelGGL(gemm_f32_rows16_grouped_kernel<>, dim3((unsigned)((nx + 15) / 16), (unsigned)((mx + 15) / 16), (unsigned)gs.zoff[gs.n]), dim3(256), 0, st,
                     gs);
}

// C = alpha * (A_0 B_0 + A_1 B_1 + ...) + ...: the terms of one output summed inside the workgroup (the five cross-attention
// query gradients that meet in one tensor: one launch and one pass over the output instead of five read-modify-writes)
struct GemmSum {
  GemmGroup g;                       // output, extents and the first term
  GemmSeg more[GEMM_MAX_GROUPS - 1];
  int n_more;
};
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) gemm_f32_rows16_sum_kernel(const GemmSum s) {
  __shared__ GemmSeg more[GEMM_MAX_GROUPS - 1];        // the extra terms, out of the kernel-argument segment
  if (threadIdx.x < (unsigned)s.n_more) more[threadIdx.x] = s.more[threadIdx.x];
  __syncthreads();
  const GemmGroup& g = s.g;
  gemm_f32_rows16_tile<true>(g.A, g.B, g.C, g.c_rs, g.c_cs, g.c_b1, g.c_b2, g.M, g.N, g.K, g.nb2, g.bias, g.alpha, g.accumulate, g.resid, g.a_act, blockIdx.x,
                             blockIdx.y, blockIdx.z, more, s.n_more);
}
static inline void launch_gemm_f32_sum(hipStream_t st, const GemmSum& s) {
  hipLaunchKernelGGL(gemm_f32_rows16_sum_kernel<>, dim3((unsigned)((s.g.N + 15) / 16), (unsigned)((s.g.M + 15) / 16), (unsigned)(s.g.nb1 * s.g.nb2)),
                     dim3(256), 0, st, s);
}

// launch the product with the kernel that fits its shape
static inline void launch_gemm_f32(hipStream_t st, const MatView& a, const MatView& b, float* C, long long c_rs, long long c_cs, long long c_b1,
                                   long long c_b2, int M, int N, int K, int nb1, int nb2, const float* bias, float alpha, int accumulate,
                                   const float* resid = nullptr, int a_act = 0) {
  if (M <= 256)
    hipLaunchKernelGGL(gemm_f32_rows16_kernel<>, dim3((unsigned)((N + 15) / 16), (unsigned)((M + 15) / 16), (unsigned)(nb1 * nb2)), dim3(256), 0, st, a, b,
                       C, c_rs, c_cs, c_b1, c_b2, M, N, K, nb2, bias, alpha, accumulate, resid, a_act);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<>, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)(nb1 * nb2)), dim3(256), 0, st, a, b, C,
                       c_rs, c_cs, c_b1, c_b2, M, N, K, nb2, bias, alpha, accumulate, resid, a_act);
}

// In-place softmax over the last axis of scores [rows][Lk]; key_padding_mask [batch][Lk] (1 = ignore), the batch of a
// row is row / rows_per_batch.  One wave per row.
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) softmax_f32_kernel(float* s, const uint8_t* kpm, long long rows, int Lk, long long rows_per_batch) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* r = s + row * Lk;
  const uint8_t* mk = kpm ? kpm + (row / rows_per_batch) * Lk : nullptr;
  float mx = -INFINITY;
  for (int c = lane; c < Lk; c += 64)
    if (!mk || !mk[c]) mx = fmaxf(mx, r[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < Lk; c += 64) {
    const float e = (!mk || !mk[c]) ? expf(r[c] - mx) : 0.f;
    r[c] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  for (int c = lane; c < Lk; c += 64) r[c] = r[c] / sum;
}

// ds = p * ((dp + extra) - sum_k (dp + extra) p)   (softmax backward), in place on dp; one wave per row
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) softmax_bwd_f32_kernel(const float* p, float* dp, const float* extra, long long rows, int Lk) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = p + row * Lk;
  float* dr = dp + row * Lk;
  const float* er = extra ? extra + row * Lk : nullptr;
  float dot = 0.f;
  for (int c = lane; c < Lk; c += 64) {
    const float d = dr[c] + (er ? er[c] : 0.f);
    dr[c] = d;
    dot = fmaf(d, pr[c], dot);
  }
  dot = wave_sum(dot);
  for (int c = lane; c < Lk; c += 64) dr[c] = pr[c] * (dr[c] - dot);
}

// dx (+)= rstd * (dh - mean(dh) - xh * mean(dh * xh)),  dh = dy * gamma,  xh = (x - mean) * rstd   (nn.LayerNorm backward
// with respect to its input); one wave per row, D <= 2048
// With (tb_h, tb_e) the incoming gradient is first taken through the TimeBlock's SiLU and modulation:
// dy <- dy * SiLU'(tb_h) * (1 + tb_e[d])   (tb_h the modulated LayerNorm output the forward kept, tb_e its [scale | shift] row).
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) layernorm_bwd_f32_kernel(const float* x, const float* g, const float* dy, float* dx, long long rows, int D,
                                                                float eps, int accumulate, const float* tb_h, const float* tb_e) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * D;
  const float* dr = dy + row * D;
  float v[32], dh[32];
  float s = 0.f;
  // every operand is requested before the first reduction (one memory round trip for the whole row)
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    v[q] = c < D ? xr[c] : 0.f;
    float dyv = c < D ? dr[c] : 0.f;
    if (tb_h && c < D) {
      const float y = tb_h[row * D + c], sg = 1.0f / (1.0f + expf(-y));
      dyv *= (sg * (1.0f + y * (1.0f - sg))) * (1.0f + tb_e[c]);
    }
    dh[q] = c < D ? dyv * g[c] : 0.f;
    s += v[q];
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    const float d = c < D ? v[q] - mean : 0.f;
    v[q] = d;
    ss += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    v[q] *= rstd;
    s1 += dh[q];
    s2 = fmaf(dh[q], v[q], s2);
  }
  const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    if (c < D) {
      float r = rstd * (dh[q] - m1 - v[q] * m2);
      if (accumulate) r += dx[row * D + c];
      dx[row * D + c] = r;
    }
  }
}

// h = LayerNorm(x) * (1 + e[d]) + e[D + d]: the TimeBlock's norm and modulation in one pass (cross_attention.py:433-436)
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) layernorm_mod_f32_kernel(const float* x, const float* g, const float* b, const float* e, float* out, long long rows,
                                                                int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * D;
  float v[32];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    v[q] = c < D ? xr[c] : 0.f;
    s += v[q];
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    const float d = c < D ? v[q] - mean : 0.f;
    v[q] = d;
    ss += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    if (c < D) out[row * D + c] = (v[q] * rstd * g[c] + b[c]) * (1.0f + e[c]) + e[D + c];
  }
}

// Grouped forms of the row kernels above for the five cross-attentions of a layer (blockIdx.y = group).
#define ROW_MAX_GROUPS 5
struct LnGroups {
  const float* x[ROW_MAX_GROUPS];
  const float* g[ROW_MAX_GROUPS];
  const float* b[ROW_MAX_GROUPS];
  float* out[ROW_MAX_GROUPS];
  long long rows[ROW_MAX_GROUPS];
};
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) layernorm_f32_grouped_kernel(const LnGroups gs, int D, float eps) {
  const int gi = blockIdx.y, lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= gs.rows[gi]) return;
  const float* xr = gs.x[gi] + row * D;
  const float *g = gs.g[gi], *b = gs.b[gi];
  float* out = gs.out[gi];
  float v[32];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    v[q] = c < D ? xr[c] : 0.f;
    s += v[q];
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    const float d = c < D ? v[q] - mean : 0.f;
    v[q] = d;
    ss += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int c = lane + 64 * q;
    if (c < D) out[row * D + c] = v[q] * rstd * g[c] + b[c];
  }
}

// softmax / softmax backward over rows that sit in per-batch blocks `bstride` floats apart (rows_per_batch rows of Lk each)
struct SoftmaxGroups {
  float* s[ROW_MAX_GROUPS];            // scores -> probabilities (forward); dp -> ds (backward)
  const float* p[ROW_MAX_GROUPS];      // backward: probabilities
  const float* extra[ROW_MAX_GROUPS];  // backward: gradient arriving at the probabilities directly (or null)
  const uint8_t* kpm[ROW_MAX_GROUPS];  // forward: key padding mask [batch][Lk] (or null)
  long long rows[ROW_MAX_GROUPS], rows_per_batch[ROW_MAX_GROUPS], s_bstride[ROW_MAX_GROUPS], p_bstride[ROW_MAX_GROUPS], e_bstride[ROW_MAX_GROUPS];
  int Lk[ROW_MAX_GROUPS];
};
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) softmax_f32_grouped_kernel(const SoftmaxGroups gs) {
  const int gi = blockIdx.y, lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= gs.rows[gi]) return;
  const int Lk = gs.Lk[gi];
  const long long bt = row / gs.rows_per_batch[gi], lr = row % gs.rows_per_batch[gi];
  float* r = gs.s[gi] + bt * gs.s_bstride[gi] + lr * Lk;
  const uint8_t* mk = gs.kpm[gi] ? gs.kpm[gi] + bt * Lk : nullptr;
  float mx = -INFINITY;
  for (int c = lane; c < Lk; c += 64)
    if (!mk || !mk[c]) mx = fmaxf(mx, r[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < Lk; c += 64) {
    const float e = (!mk || !mk[c]) ? expf(r[c] - mx) : 0.f;
    r[c] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  for (int c = lane; c < Lk; c += 64) r[c] = r[c] / sum;
}
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) softmax_bwd_f32_grouped_kernel(const SoftmaxGroups gs) {
  const int gi = blockIdx.y, lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= gs.rows[gi]) return;
  const int Lk = gs.Lk[gi];
  const long long bt = row / gs.rows_per_batch[gi], lr = row % gs.rows_per_batch[gi];
  const float* pr = gs.p[gi] + bt * gs.p_bstride[gi] + lr * Lk;
  float* dr = gs.s[gi] + bt * gs.s_bstride[gi] + lr * Lk;
  const float* er = gs.extra[gi] ? gs.extra[gi] + bt * gs.e_bstride[gi] + lr * Lk : nullptr;
  float dot = 0.f;
  for (int c = lane; c < Lk; c += 64) {
    const float d = dr[c] + (er ? er[c] : 0.f);
    dr[c] = d;
    dot = fmaf(d, pr[c], dot);
  }
  dot = wave_sum(dot);
  for (int c = lane; c < Lk; c += 64) dr[c] = pr[c] * (dr[c] - dot);
}

enum { EW_SILU = 0, EW_GELU = 1, EW_SILU_BWD = 2, EW_GELU_BWD = 3, EW_AXPY = 4, EW_ADD_BCAST = 5, EW_MODULATE = 6, EW_MODULATE_BWD = 7, EW_NOPS = 8 };

// TimeBlock backward between its output projection and its LayerNorm in one pass: out = a * SiLU'(h) * (1 + e[d])
// (h the modulated LayerNorm output kept by the forward, e the block's [scale | shift] row)
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) tb_bwd_f32_kernel(const float* a, const float* h, const float* e, float* out, long long n, int D) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float y = h[i], sg = 1.0f / (1.0f + expf(-y));
  out[i] = a[i] * (sg * (1.0f + y * (1.0f - sg))) * (1.0f + e[i % D]);
}


// Element-wise pieces of the forward / backward pass over a [R0][R1][D] tensor (index i -> d = i % D, r1 = (i / D) % R1,
// r0 = i / (D R1)):
//   SILU / GELU          out = act(a)                                  (nn.SiLU, exact-erf nn.GELU)
//   SILU_BWD / GELU_BWD  out = a * act'(b)                             (a upstream gradient, b pre-activation)
//   AXPY                 out = a + alpha * b                           (weg.update_latent: latents - lr * grad)
//   ADD_BCAST            out = a + b[r0 * s0 + r1 * s1 + d]            (temb / condition id / position rows)
//   MODULATE             out = a * (1 + b[r1][d]) + b[r1][D + d]       (TimeBlock: b = emb_layers output [R1][2 D], scale first)
//   MODULATE_BWD         out = a * (1 + b[r1][d])
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) ew_f32_kernel(int op, const float* a, const float* b, float* out, long long n, int D, int R1, long long s0,
                                                     long long s1, float alpha) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  float r = 0.f;
  switch (op) {
    case EW_SILU: r = x / (1.0f + expf(-x)); break;
    case EW_GELU: r = gelu_f(x); break;
    case EW_SILU_BWD: {
      const float y = b[i], sg = 1.0f / (1.0f + expf(-y));
      r = x * (sg * (1.0f + y * (1.0f - sg)));
      break;
    }
    case EW_GELU_BWD: {
      const float y = b[i];
      r = x * (0.5f * (1.0f + erff(y * 0.70710678118654752440f)) + y * expf(-0.5f * y * y) * 0.39894228040143267794f);
      break;
    }
    case EW_AXPY: r = x + alpha * b[i]; break;
    case EW_ADD_BCAST: {
      const long long row = i / D;
      r = x + b[(row / R1) * s0 + (row % R1) * s1 + (i % D)];
      break;
    }
    case EW_MODULATE: {
      const long long r1 = (i / D) % R1;
      const int d = (int)(i % D);
      r = x * (1.0f + b[r1 * 2 * D + d]) + b[r1 * 2 * D + D + d];
      break;
    }
    case EW_MODULATE_BWD: {
      const long long r1 = (i / D) % R1;
      r = x * (1.0f + b[r1 * 2 * D + (i % D)]);
      break;
    }
  }
  out[i] = r;
}

// The attend-and-excite objective on the listener-text attention maps and its gradient with respect to them
// (word_excitation_guidance.py:11-81; GaussianSmoothing 3 x 3, sigma 0.5, reflect padding: gaussian_smoothing.py:21-72).
//   att     [B][NL][L][S]  probabilities of the tlsn cross-attention of every layer (Denoiser.forward's att_mats[2])
//   tok_off [B + 1], tok_idx [tok_off[B]]  focus token indices of each sample (text positions, BOS = 0)
//   last    exclusive end of the text slice [1, last) (eot index when normalize_eot, else S - 1)
//   ws      workspace >= B * (3 * L * W + 3 * nt_max) 4-byte words, W = last - 1, nt_max >= tokens of any sample
// outputs: losses [B] (mean over the sample's tokens of max(0, 1 - max attention)), max_att [tok_off[B]],
//          d_att [B][NL][L][S] = d(mean_b losses[b]) / d att.  One workgroup per sample; phases separated by barriers.
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) weg_focus_kernel(const float* att, const int* tok_off, const int* tok_idx, int B, int NL, int L, int S,
                                                        int last, int nt_max, float k00, float k01, float k11, float* ws, float* losses,
                                                        float* max_att, float* d_att) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int W = last - 1;
  const long long LW = (long long)L * W;
  float* sm = ws + (long long)b * (3 * LW + 3 * nt_max);
  float* sg = sm + LW;
  float* dsm = sg + LW;
  float* tok_g = dsm + LW;            // [nt_max] upstream gradient at the token's arg-max cell
  int* tok_l = (int*)(tok_g + nt_max);     // [nt_max] arg-max frame
  int* tok_w = tok_l + nt_max;             // [nt_max] column in the slice
  const float* ab = att + (long long)b * NL * L * S;
  // 1. layer mean, slice [1, last), softmax over the slice
  for (int l = tid; l < L; l += 256) {
    float mx = -INFINITY;
    for (int w = 0; w < W; ++w) {
      float m = 0.f;
      for (int n = 0; n < NL; ++n) m += ab[((long long)n * L + l) * S + 1 + w];
      m = m / (float)NL;
      sm[(long long)l * W + w] = m;
      mx = fmaxf(mx, m);
    }
    float sum = 0.f;
    for (int w = 0; w < W; ++w) {
      const float e = expf(sm[(long long)l * W + w] - mx);
      sm[(long long)l * W + w] = e;
      sum += e;
    }
    for (int w = 0; w < W; ++w) sm[(long long)l * W + w] /= sum;
  }
  __syncthreads();
  // 2. 3 x 3 Gaussian correlation over the reflect-padded map
  const float kk[3][3] = {{k00, k01, k00}, {k01, k11, k01}, {k00, k01, k00}};
  for (long long e = tid; e < LW; e += 256) {
    const int l = (int)(e / W), w = (int)(e % W);
    float v = 0.f;
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) {
        int ll = l + a - 1, ww = w + c - 1;
        ll = ll < 0 ? 1 : (ll >= L ? L - 2 : ll);
        ww = ww < 0 ? 1 : (ww >= W ? W - 2 : ww);
        v += kk[a][c] * sm[(long long)ll * W + ww];
      }
    sg[e] = v;
  }
  __syncthreads();
  // 3. per focus token: max over frames, hinge
  const int t0 = tok_off[b], nt = tok_off[b + 1] - t0;
  for (int t = tid; t < nt; t += 256) {
    const int w = tok_idx[t0 + t] - 1;
    float best = -INFINITY;
    int bl = 0;
    for (int l = 0; l < L; ++l) {
      const float v = sg[(long long)l * W + w];
      if (v > best) { best = v; bl = l; }
    }
    max_att[t0 + t] = best;
    tok_l[t] = bl;
    tok_w[t] = w;
    tok_g[t] = (1.0f - best > 0.f) ? -1.0f / ((float)nt * (float)B) : 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int t = 0; t < nt; ++t) s += fmaxf(0.f, 1.0f - max_att[t0 + t]);
    losses[b] = nt > 0 ? s / (float)nt : 0.f;
  }
  // 4. adjoint of the padded correlation, gathered per cell over the (few) tokens: deterministic, no atomics
  for (long long e = tid; e < LW; e += 256) {
    const int l = (int)(e / W), w = (int)(e % W);
    float v = 0.f;
    for (int t = 0; t < nt; ++t) {
      const float g = tok_g[t];
      if (g == 0.f) continue;
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
          int ll = tok_l[t] + a - 1, ww = tok_w[t] + c - 1;
          ll = ll < 0 ? 1 : (ll >= L ? L - 2 : ll);
          ww = ww < 0 ? 1 : (ww >= W ? W - 2 : ww);
          if (ll == l && ww == w) v += kk[a][c] * g;
        }
    }
    dsm[e] = v;
  }
  __syncthreads();
  // 5. softmax backward, spread over the layers (mean) and written into the full-width maps
  float* db = d_att + (long long)b * NL * L * S;
  for (int l = tid; l < L; l += 256) {
    float dot = 0.f;
    for (int w = 0; w < W; ++w) dot = fmaf(dsm[(long long)l * W + w], sm[(long long)l * W + w], dot);
    for (int s = 0; s < S; ++s) {
      float v = 0.f;
      if (s >= 1 && s < last) {
        const int w = s - 1;
        v = sm[(long long)l * W + w] * (dsm[(long long)l * W + w] - dot) / (float)NL;
      }
      for (int n = 0; n < NL; ++n) db[((long long)n * L + l) * S + s] = v;
    }
  }
}

// The same objective for small maps (L * (last - 1) <= WEG_SMALL_CELLS cells, <= WEG_SMALL_TOK focus tokens per sample: the product shape is
// 16 x 16): the three maps live in LDS, every phase but the two row-serial ones runs one thread per cell, and the layers are read and
// written along the key axis.  Operation for operation the arithmetic of weg_focus_kernel (same summation orders: the results are
// bit-identical); 34 -> ~6 us at the product shape, where the general kernel's 16 active threads walk global memory serially.
#define WEG_SMALL_CELLS 1024
#define WEG_SMALL_TOK 64
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) weg_focus_small_kernel(const float* att, const int* tok_off, const int* tok_idx, int B, int NL, int L, int S,
                                                              int last, float k00, float k01, float k11, float* losses, float* max_att,
                                                              float* d_att) {
  __shared__ float sm[WEG_SMALL_CELLS], sg[WEG_SMALL_CELLS], dsm[WEG_SMALL_CELLS];
  __shared__ float tok_g[WEG_SMALL_TOK], dots[64];
  __shared__ int tok_l[WEG_SMALL_TOK], tok_w[WEG_SMALL_TOK];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int W = last - 1, LW = L * W;
  const float* ab = att + (long long)b * NL * L * S;
  // 1. layer mean over the slice [1, last) (a thread per cell), softmax over the slice (a thread per frame)
  for (int e = tid; e < LW; e += 256) {
    const int l = e / W, w = e - l * W;
    float m = 0.f;
    for (int n = 0; n < NL; ++n) m += ab[((long long)n * L + l) * S + 1 + w];
    sm[e] = m / (float)NL;
  }
  __syncthreads();
  for (int l = tid; l < L; l += 256) {
    float mx = -INFINITY;
    for (int w = 0; w < W; ++w) mx = fmaxf(mx, sm[l * W + w]);
    float sum = 0.f;
    for (int w = 0; w < W; ++w) {
      const float e = expf(sm[l * W + w] - mx);
      sm[l * W + w] = e;
      sum += e;
    }
    for (int w = 0; w < W; ++w) sm[l * W + w] /= sum;
  }
  __syncthreads();
  // 2. 3 x 3 Gaussian correlation over the reflect-padded map
  const float kk[3][3] = {{k00, k01, k00}, {k01, k11, k01}, {k00, k01, k00}};
  for (int e = tid; e < LW; e += 256) {
    const int l = e / W, w = e - l * W;
    float v = 0.f;
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c) {
        int ll = l + a - 1, ww = w + c - 1;
        ll = ll < 0 ? 1 : (ll >= L ? L - 2 : ll);
        ww = ww < 0 ? 1 : (ww >= W ? W - 2 : ww);
        v += kk[a][c] * sm[ll * W + ww];
      }
    sg[e] = v;
  }
  __syncthreads();
  // 3. per focus token: max over frames, hinge
  const int t0 = tok_off[b], nt = tok_off[b + 1] - t0;
  for (int t = tid; t < nt; t += 256) {
    const int w = tok_idx[t0 + t] - 1;
    float best = -INFINITY;
    int bl = 0;
    for (int l = 0; l < L; ++l) {
      const float v = sg[l * W + w];
      if (v > best) { best = v; bl = l; }
    }
    max_att[t0 + t] = best;
    tok_l[t] = bl;
    tok_w[t] = w;
    tok_g[t] = (1.0f - best > 0.f) ? -1.0f / ((float)nt * (float)B) : 0.f;
    dots[t] = fmaxf(0.f, 1.0f - best);   // (hinge terms, summed in token order below)
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int t = 0; t < nt; ++t) s += dots[t];
    losses[b] = nt > 0 ? s / (float)nt : 0.f;
  }
  // 4. adjoint of the padded correlation, gathered per cell over the (few) tokens: deterministic, no atomics
  for (int e = tid; e < LW; e += 256) {
    const int l = e / W, w = e - l * W;
    float v = 0.f;
    for (int t = 0; t < nt; ++t) {
      const float g = tok_g[t];
      if (g == 0.f) continue;
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
          int ll = tok_l[t] + a - 1, ww = tok_w[t] + c - 1;
          ll = ll < 0 ? 1 : (ll >= L ? L - 2 : ll);
          ww = ww < 0 ? 1 : (ww >= W ? W - 2 : ww);
          if (ll == l && ww == w) v += kk[a][c] * g;
        }
    }
    dsm[e] = v;
  }
  __syncthreads();
  // 5. softmax backward (the row's dot product by a thread per frame), spread over the layers (mean) and written into the full-width maps
  for (int l = tid; l < L; l += 256) {
    float dot = 0.f;
    for (int w = 0; w < W; ++w) dot = fmaf(dsm[l * W + w], sm[l * W + w], dot);
    dots[l] = dot;
  }
  __syncthreads();
  float* db = d_att + (long long)b * NL * L * S;
  for (int e = tid; e < L * S; e += 256) {
    const int l = e / S, s = e - l * S;
    float v = 0.f;
    if (s >= 1 && s < last) {
      const int w = s - 1;
      v = sm[l * W + w] * (dsm[l * W + w] - dots[l]) / (float)NL;
    }
    for (int n = 0; n < NL; ++n) db[((long long)n * L + l) * S + s] = v;
  }
}
