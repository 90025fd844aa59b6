// Row-wise / element-wise kernels of the denoising step (HBM-bound; one 64-lane wave per 512-wide row,
// 8 contiguous floats per lane = two 16-byte loads, SP outputs written as 16-byte hi / lo chunks).
#pragma once
#include "cfd_common.hpp"

// ------------------------------------------------------------------------------------------------
// LayerNorm (+ optional AdaLN modulate + SiLU) : x fp32 [M][512] -> SP [M][512]
//   plain : y = (x-mean)*rstd*g + b                                  cross_attention.py:568,578,659
//   adaln : y = silu( LN(x)*(1+scale) + shift )                      TimeBlock.forward :426-439
// ------------------------------------------------------------------------------------------------
struct LnArgs {
  const float* x;
  char* out;
  long long M;
  const float* g;
  const float* b;
  int adaln;
  const float* ss;        // (1+scale | shift) rows of 1024 floats for THIS time block, t-row stride ss_tstride
  long long ss_tstride;
  const int* d_step;
  int tmode;              // 0: t-row = *d_step ; 1: t-row = trow0 + row / L
  int L;
  int trow0;
};

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) ln_rows_kernel(const LnArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const float* xr = a.x + row * CFD_D + lane * 8;
  // every operand is requested before the first is used: the AdaLN rows used to be loaded inside `if (a.adaln)`, behind the statistics,
  // i.e. a second round trip per row (half of the launches of a step are AdaLN ones)
  const long long trow = a.tmode ? (a.trow0 + row / a.L) : (a.d_step ? (long long)(*a.d_step) : 0);   // (null: `ss` is this step's row already)
  const float* sc = a.adaln ? a.ss + trow * a.ss_tstride + lane * 8 : a.g + lane * 8;   // (plain LayerNorm: a valid address, values unused)
  const float* sh = a.adaln ? sc + CFD_D : sc;
  const float4 p = *reinterpret_cast<const float4*>(xr), q = *reinterpret_cast<const float4*>(xr + 4);
  const float4 g0 = *reinterpret_cast<const float4*>(a.g + lane * 8), g1 = *reinterpret_cast<const float4*>(a.g + lane * 8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(a.b + lane * 8), b1 = *reinterpret_cast<const float4*>(a.b + lane * 8 + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
  const float4 h0 = *reinterpret_cast<const float4*>(sh), h1 = *reinterpret_cast<const float4*>(sh + 4);
  __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the parameter loads behind the wait for x)
  float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
  const float gp[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float bp[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += v[e];
  const float mean = wave_sum(s) * (1.0f / CFD_D);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { v[e] -= mean; ss += v[e] * v[e]; }
  const float var = wave_sum(ss) * (1.0f / CFD_D);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * gp[e] + bp[e];
  if (a.adaln) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e] * scv[e] + shv[e]);
  }
  // (a NaN anywhere in the row -- a token whose cross-attention had nothing but masked keys -- makes the whole row NaN, as in the
  //  reference's LayerNorm; the store keeps it a NaN so that it reaches the output: cfd_common.hpp)
  sp_store8_keep_nan(a.out + row * (CFD_D * 4), lane * 8, v);
}

// ------------------------------------------------------------------------------------------------
// Memory preparation (denoiser.py:223-261 time add, :332-353 condition-id + sine PE) followed by the
// per-layer memory LayerNorm's normalisation (cross_attention.py:581-585; its affine is folded into
// the memory-side projection weights at load time):
//   n[u][s][:] = normalise( raw[u][s] + temb[t] + E_cond[j] + pe[s] ),   rows s >= S are zero.
// ------------------------------------------------------------------------------------------------
struct MemPrepArgs {
  const float* raw;   // [U][S][512]
  int U, S, Sp;
  const float* temb;  // [T][512]
  const int* d_step;
  int tmode;          // 0: t-row = *d_step ; 1: t-row = u
  const float* cond;  // [512]
  const float* pe;    // [>=S][512]
  char* n_sp;         // SP [U*Sp][512]
};

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) mem_prep_kernel(const MemPrepArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)a.U * a.Sp) return;
  const int u = (int)(row / a.Sp), s = (int)(row % a.Sp);
  float v[8];
  if (s >= a.S) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  } else {
    const float* r = a.raw + ((long long)u * a.S + s) * CFD_D + lane * 8;
    const float* te = a.temb + (long long)(a.tmode ? u : *a.d_step) * CFD_D + lane * 8;
    const float* ce = a.cond + lane * 8;
    const float* pe = a.pe + (long long)s * CFD_D + lane * 8;
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = ((te[e] + r[e]) + ce[e]) + pe[e];
      sum += v[e];
    }
    const float mean = wave_sum(sum) * (1.0f / CFD_D);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] -= mean; ss += v[e] * v[e]; }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) * (1.0f / CFD_D) + 1e-5f);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= rstd;
  }
  sp_store8(a.n_sp + row * (CFD_D * 4), lane * 8, v);
}

// ------------------------------------------------------------------------------------------------
// Timestep-independent part of the memory preparation (round 2: the memory-side projections leave the loop).
// The memory LayerNorm sees v_s = m_s + temb(t) with m_s = raw_s + E_cond + pe_s constant during a sampling run.  With
// a_s = m_s - mean(m_s) and b = temb(t) - mean(temb(t)):   v_s - mean(v_s) = a_s + b,
//   var_s = (|a_s|^2 + 2 a_s.b + |b|^2) / 512,   n_s = (a_s + b) * rstd_s,   rstd_s = 1 / sqrt(var_s + eps)
// so every folded projection W n_s = rstd_s (W a_s + W b): the big products W a_s are computed ONCE per run from a_s (this
// kernel's output), W b is one vector per timestep, and rstd_s is one scalar per key and step (mem_scale_all_kernel).
//   a[u][s][:] = m - mean(m)  as SP,  asq[u][s] = |a|^2   (rows s >= S: zero)
// ------------------------------------------------------------------------------------------------
struct MemCenterArgs {
  const float* raw;   // [U][S][512]
  int U, S, Sp;
  const float* cond;  // [512]
  const float* pe;    // [>=S][512]
  char* a_sp;         // SP [U*Sp][512]
  float* asq;         // [U*Sp]
  unsigned int* sat;  // the handle's saturation census (cfd_common.hpp), or null
};

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) mem_center_kernel(const MemCenterArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)a.U * a.Sp) return;
  const int u = (int)(row / a.Sp), s = (int)(row % a.Sp);
  float v[8];
  float ss = 0.f;
  if (s >= a.S) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
  } else {
    const float* r = a.raw + ((long long)u * a.S + s) * CFD_D + lane * 8;
    const float* ce = a.cond + lane * 8;
    const float* pe = a.pe + (long long)s * CFD_D + lane * 8;
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = (r[e] + ce[e]) + pe[e];
      sum += v[e];
    }
    const float mean = wave_sum(sum) * (1.0f / CFD_D);
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] -= mean; ss += v[e] * v[e]; }
    ss = wave_sum(ss);
  }
  sat_note<8>(a.sat, v);
  sp_store8(a.a_sp + row * (CFD_D * 4), lane * 8, v);
  if (lane == 0) a.asq[row] = ss;
}

// Per step and memory: rstd of every key from its dot product with the centred timestep embedding, and the key bias of every layer
//   rs[key] = 1 / sqrt((asq + 2 a.b + |b|^2) / 512 + 1e-5)
//   cbk[l][key] = rs * (ca[l][key] + cbb[l])        ca = c_l . a_s (with -inf on dead keys, written by EpiMemK), cbb = c_l . b
// One wave per key; a is read back from its split-pair form (hi + lo).
struct MemScaleArgs {
  const char* a_sp;    // SP [rows][512]
  const float* asq;    // [rows]
  long long rows;
  const float* btab;   // [T][512] centred timestep embeddings
  const float* bsq;    // [T]
  const float* ca;     // [nl][rows]
  const float* cbb;    // table row t: cbb[t * cbb_tstride + l]
  long long cbb_tstride;
  const int* d_step;
  int nl;
  float* rs;           // [rows]
  float* cbk;          // [nl][rows]
};

// every static memory of a step in ONE launch (five launches of this size were five launch latencies: four of them have a few hundred rows)
struct MemScaleAllArgs {
  MemScaleArgs m[CFD_NMEM];
  int first[CFD_NMEM + 1];   // first workgroup of memory slot k (slots packed: n of them)
  int n;
};

__device__ __forceinline__ void mem_scale_rows(const MemScaleArgs& a, long long row, int lane);

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) mem_scale_all_kernel(const MemScaleAllArgs g) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if (q < g.n && (int)blockIdx.x >= g.first[q]) k = q;
  // (copies of the chosen slot's fields: a select per field, no indexing of the argument struct)
  MemScaleArgs a = g.m[0];
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if (k == q) a = g.m[q];
  int first = 0;
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if (k == q) first = g.first[q];
  const long long row = (long long)((int)blockIdx.x - first) * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  mem_scale_rows(a, row, threadIdx.x & 63);
}

__device__ __forceinline__ void mem_scale_rows(const MemScaleArgs& a, long long row, int lane) {
  const int t = *a.d_step;
  const char* ap = a.a_sp + row * (CFD_D * 4) + (size_t)(lane >> 2) * 128 + (lane & 3) * 16;   // 8 consecutive columns: lane * 8
  const spx8 h = *reinterpret_cast<const spx8*>(ap);
  const spx8 l = *reinterpret_cast<const spx8*>(ap + 64);
  const float* bp = a.btab + (long long)t * CFD_D + lane * 8;
  const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
  // (the scalars of the row and the layers' terms are requested with the row, at clamped lanes: loaded where they are used they
  //  were a second and a third round trip behind the reduction)
  const int ll = min(lane, a.nl - 1);
  const float asq = a.asq[row], bsq = a.bsq[t];
  const float ca = a.ca[(long long)ll * a.rows + row], cbb = a.cbb[(long long)t * a.cbb_tstride + ll];
  __builtin_amdgcn_sched_barrier(0);
  const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  float dot = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) dot += ((float)h[e] + (float)l[e]) * bv[e];
  dot = wave_sum(dot);
  const float var = (asq + 2.0f * dot + bsq) * (1.0f / CFD_D);
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  if (lane == 0) a.rs[row] = rstd;
  if (lane < a.nl) a.cbk[(long long)lane * a.rows + row] = rstd * (ca + cbb);
}

// b[t][:] = temb[t][:] - mean(temb[t]) as fp32 and SP, bsq[t] = |b|^2   (one wave per table row)
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) temb_center_kernel(const float* temb, int T, float* b, char* b_sp, float* bsq) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  float v[8];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { v[e] = temb[(long long)row * CFD_D + lane * 8 + e]; sum += v[e]; }
  const float mean = wave_sum(sum) * (1.0f / CFD_D);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { v[e] -= mean; ss += v[e] * v[e]; b[(long long)row * CFD_D + lane * 8 + e] = v[e]; }
  ss = wave_sum(ss);
  sp_store8(b_sp + (long long)row * (CFD_D * 4), lane * 8, v);
  if (lane == 0) bsq[row] = ss;
}

// ------------------------------------------------------------------------------------------------
// Masked softmax over up to 5 key segments of one score row; writes P (SP, zero in the padding) and
// optionally the probabilities the reference returns as att_mats (cross_attention.py:227-234).
// One wave per (b, l) row; a segment of Sp <= 2048 keys lives in registers (<= 4 chunks of 8 / lane).
// ------------------------------------------------------------------------------------------------
#define SM_MAX_CHUNKS 4
struct SoftmaxArgs {
  const float* sc;
  char* P;
  long long ld;        // floats per row (= total padded keys)
  long long rows;
  int rows_per_b;
  int nseg;
  int off[CFD_NMEM], S[CFD_NMEM], Sp[CFD_NMEM];
  const uint8_t* mask[CFD_NMEM];  // [U][S] (1 = padded key); NEVER null (the host passes an all-zero mask)
  int has_mask[CFD_NMEM];         // 0: the mask is the all-zero stand-in (long segments skip the byte loads)
  const int* map[CFD_NMEM];       // b -> u, or null (u = b)
  float* att[CFD_NMEM];           // [Be][nl][L][S] or null
  int layer, nl;
  int skip_seg;                   // segment whose softmax was done by the score product itself (EpiTileSoftmax) ...
  const uint8_t* skip_rows;       // ... for the batch rows flagged here ([Be] bytes), or null
};

// Segments of <= 64 keys (text / activity-bit / listener-id memories) are handled TOGETHER, one 8-lane group per
// segment, so a row costs one load->reduce->store round for all of them instead of one serialised round each.
__device__ __forceinline__ void softmax_store(const SoftmaxArgs& a, int g, int b, int l, char* prow, int c0, float* v) {
  const int S = a.S[g];
  sp_store8_keep_nan(prow, a.off[g] + c0, v);   // (an all-masked row's probabilities are NaN, like the reference's)
  if (a.att[g]) {
    float* att = a.att[g] + (((long long)b * a.nl + a.layer) * a.rows_per_b + l) * S;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e < S) att[c0 + e] = v[e];
  }
}

template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const SoftmaxArgs a) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int b = (int)(row / a.rows_per_b), l = (int)(row % a.rows_per_b);
  const float* srow = a.sc + row * a.ld;
  char* prow = a.P + row * a.ld * 4;

  // ---- short segments: lane group (lane >> 3) <-> the k-th short segment, lane & 7 <-> chunk of 8 keys ------
  {
    int myseg = -1, k = 0;
#pragma unroll
    for (int g = 0; g < CFD_NMEM; ++g)
      if (g < a.nseg && a.Sp[g] <= 64) {
        if (k == (lane >> 3)) myseg = g;
        ++k;
      }
    if (k > 0) {   // wave-uniform
      int S = 1, Sp = 0, off = 0;
      const uint8_t* mk = a.mask[0];
#pragma unroll
      for (int g = 0; g < CFD_NMEM; ++g)
        if (g == myseg) {
          S = a.S[g]; Sp = a.Sp[g]; off = a.off[g];
          mk = a.mask[g] + (long long)(a.map[g] ? a.map[g][b] : b) * S;
        }
      const int c0 = (lane & 7) * 8;
      const bool act = myseg >= 0 && c0 < Sp;
      float v[8];
      float mx = -INFINITY;
      if (act) {
        const float4 p = *reinterpret_cast<const float4*>(srow + off + c0);
        const float4 q = *reinterpret_cast<const float4*>(srow + off + c0 + 4);
        v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int s = c0 + e;
          const bool dead = (s >= S) | (mk[min(s, S - 1)] != 0);   // branch-free: the mask pointer is never null
          v[e] = dead ? -INFINITY : v[e];
          mx = fmaxf(mx, v[e]);
        }
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      float sum = 0.f;
      if (act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = __expf(v[e] - mx); sum += v[e]; }
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if (act) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] / sum;
#pragma unroll
        for (int g = 0; g < CFD_NMEM; ++g)
          if (g == myseg) softmax_store(a, g, b, l, prow, c0, v);
      }
    }
  }

  // ---- long segments: the whole wave per segment, <= 4 chunks of 8 keys per lane in registers -----------------
#pragma unroll
  for (int g = 0; g < CFD_NMEM; ++g) {
    if (g >= a.nseg) break;
    const int S = a.S[g], Sp = a.Sp[g], off = a.off[g];
    if (Sp <= 64) continue;
    if (a.skip_rows && g == a.skip_seg && a.skip_rows[b]) continue;   // wave-uniform
    const uint8_t* mk = a.mask[g] + (long long)(a.map[g] ? a.map[g][b] : b) * S;
    float v[SM_MAX_CHUNKS][8];
    float mx = -INFINITY;
#pragma unroll
    for (int n = 0; n < SM_MAX_CHUNKS; ++n) {
      const int c0 = (lane + 64 * n) * 8;
      if (c0 < Sp) {
        const float4 p = *reinterpret_cast<const float4*>(srow + off + c0);
        const float4 q = *reinterpret_cast<const float4*>(srow + off + c0 + 4);
        v[n][0] = p.x; v[n][1] = p.y; v[n][2] = p.z; v[n][3] = p.w;
        v[n][4] = q.x; v[n][5] = q.y; v[n][6] = q.z; v[n][7] = q.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int s = c0 + e;
          bool dead = s >= S;
          if (a.has_mask[g]) dead |= mk[min(s, S - 1)] != 0;   // wave-uniform condition
          v[n][e] = dead ? -INFINITY : v[n][e];
          mx = fmaxf(mx, v[n][e]);
        }
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < SM_MAX_CHUNKS; ++n) {
      const int c0 = (lane + 64 * n) * 8;
      if (c0 < Sp) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[n][e] = __expf(v[n][e] - mx);  // all-masked row: (-inf)-(-inf) = NaN, as in the reference
          sum += v[n][e];
        }
      }
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int n = 0; n < SM_MAX_CHUNKS; ++n) {
      const int c0 = (lane + 64 * n) * 8;
      if (c0 < Sp) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[n][e] = v[n][e] / sum;
        softmax_store(a, g, b, l, prow, c0, v[n]);
      }
    }
  }
}

// Per-tile softmax statistics (EpiTileSoftmax) -> fold weights of the tile-relative P.V product:
//   m = max_t m_t ;  l = sum_t l_t exp(m_t - m) ;  alpha_t = exp(m_t - m) / l
// (a row whose keys are all masked gives 0 / 0 = NaN, as the reference's softmax does)
template <int CFD_KI = 0>
__global__ void attn_alpha_kernel(const float2* stats, float* alpha, long long rows, int ntiles) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float2* st = stats + row * ntiles;
  float m = -INFINITY;
  for (int t = 0; t < ntiles; ++t) m = fmaxf(m, st[t].x);
  float l = 0.f;
  for (int t = 0; t < ntiles; ++t) l += (st[t].x == -INFINITY) ? 0.f : st[t].y * __expf(st[t].x - m);
  for (int t = 0; t < ntiles; ++t) alpha[row * ntiles + t] = (st[t].x == -INFINITY) ? 0.f / l : __expf(st[t].x - m) / l;
}

// ------------------------------------------------------------------------------------------------
// fp32 [R][K] -> SP [R][K]   (K % 8 == 0); used for weights at load time
// ------------------------------------------------------------------------------------------------
template <int CFD_KI = 0>
__global__ void to_split_kernel(const float* in, char* out, long long R, int K, long long ld_in, long long ld_out_bytes, unsigned int* sat) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kc = K / 8;
  if (idx >= R * kc) return;
  const long long r = idx / kc;
  const int c = (int)(idx % kc) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = in[r * ld_in + c + e];
  sat_note<8>(sat, v);
  sp_store8(out + r * ld_out_bytes, c, v);
}

// ------------------------------------------------------------------------------------------------
// out[r][n] = post( b[n] + sum_k in[r][k] * W[n][k] ),  K = 512.  One wave per output column n keeps its
// weight row in registers and sweeps the R input rows.  (timestep MLP + the 18 TimeBlock emb_layers:
// embeddings.py:298-305, cross_attention.py:432-434)
//   in_act: 0 none, 1 silu on the input;  post: 0 none, 1 silu, 2 "first 512 columns get +1" (1+scale)
// ------------------------------------------------------------------------------------------------
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) small_linear_kernel(const float* in, const int* in_rows, long long ld_in,
                                                           const float* W, const float* bias, float* out,
                                                           long long ld_out, int R, int N, int in_act, int post) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float w[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) w[e] = W[(long long)n * CFD_D + lane * 8 + e];
  const float bn = bias[n];
  for (int r = blockIdx.y; r < R; r += gridDim.y) {
    const float* ir = in + (long long)(in_rows ? in_rows[r] : r) * ld_in + lane * 8;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = ir[e];
      if (in_act == 1) x = silu_f(x);
      s += x * w[e];
    }
    s = wave_sum(s) + bn;
    if (post == 1) s = silu_f(s);
    if (post == 2 && n < CFD_D) s = 1.0f + s;
    if (lane == 0) out[(long long)r * ld_out + n] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// Load-time weight folding in float64:  C[m][n] = alpha * rowscale? ... generic strided product
//   C(m,n) = alpha * sum_k A(m,k) * B(k,n) * (colscale ? colscale[n] : 1) (+ addC ? addC(m,n) : 0)
// A, B float or double with element strides; C float or double, row-major.
// ------------------------------------------------------------------------------------------------
template <class TA, class TB, class TC>
__global__ void fold_mm_kernel(const TA* A, long long sam, long long sak, const TB* B, long long sbk, long long sbn,
                               TC* C, long long ldc, int M, int N, int K, double alpha, const float* colscale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  double acc = 0.0;
  for (int k = 0; k < K; ++k) acc += (double)A[m * sam + k * sak] * (double)B[k * sbk + n * sbn];
  acc *= alpha;
  if (colscale) acc *= (double)colscale[n];
  C[(long long)m * ldc + n] = (TC)acc;
}

// y[m] = alpha * sum_k A(m,k) x[k] (+ add[m])   in float64, output float or double
template <class TA, class TX, class TY>
__global__ void fold_mv_kernel(const TA* A, long long sam, long long sak, const TX* x, const double* add_d,
                               const float* add_f, TY* y, int M, int K, double alpha, const float* rowscale) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) acc += (double)A[m * sam + k * sak] * (double)x[k];
  acc *= alpha;
  if (rowscale) acc *= (double)rowscale[m];
  if (add_d) acc += add_d[m];
  if (add_f) acc += (double)add_f[m];
  y[m] = (TY)acc;
}

// ------------------------------------------------------------------------------------------------
// Sampler element-wise kernels
// ------------------------------------------------------------------------------------------------
struct StepCoef {   // one row per loop iteration i (timestep t_i), float32 as diffusers computes them
  float sb;         // sqrt(1 - abar_t)
  float sa;         // sqrt(abar_t)
  float c0;         // DDPM: x0 coefficient            | DDIM: sqrt(abar_prev)
  float cx;         // DDPM: current-sample coefficient | DDIM: direction coefficient sqrt(1-abar_prev-std^2)
  float sigma;      // noise std (0 when no noise is added)
  float use_noise;  // 1.0 if a N(0,1) draw is added at this step
  float pad0, pad1;
};

// Philox4x32-10 (restated in oracle/philox_ref.py, checked there against the Random123 known answers)
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float4 philox_normal4(uint64_t seed, uint32_t group, uint32_t step, uint32_t utt,
                                                 uint32_t stream) {
  uint32_t w[4];
  philox4x32_10(group, step, utt, stream, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  float u[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) u[e] = ((float)(w[e] >> 8) + 0.5f) * 5.9604644775390625e-08f;  // 2^-24
  const float r0 = sqrtf(-2.0f * logf(u[0])), r1 = sqrtf(-2.0f * logf(u[2]));
  const float t0 = 6.283185307179586f * u[1], t1 = 6.283185307179586f * u[3];
  return make_float4(r0 * cosf(t0), r0 * sinf(t0), r1 * cosf(t1), r1 * sinf(t1));
}

// fill [B][L*128] with N(0,1): stream 1 = initial latents (convofusion.py:412-419)
template <int CFD_KI = 0>
__global__ void philox_fill_kernel(float* out, int B, int per_utt, uint64_t seed, uint32_t step, uint32_t utt0,
                                   uint32_t stream, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int gpu = per_utt / 4;
  if (idx >= (long long)B * gpu) return;
  const int b = (int)(idx / gpu), g = (int)(idx % gpu);
  float4 z = philox_normal4(seed, (uint32_t)g, step, utt0 + b, stream);
  z.x *= scale; z.y *= scale; z.z *= scale; z.w *= scale;
  *reinterpret_cast<float4*>(out + (long long)b * per_utt + g * 4) = z;
}

// Start of loop iteration i: optional in-painting overwrite of the first `pl` tokens
// (unbounded_synthesis.py:70-76, including its aliasing quirk at i == 0), then replicate the latents
// G times into the SP denoiser input (convofusion.py:499-501).
struct BeginArgs {
  float* latents;        // [B][L][128]
  char* sample_sp;       // SP [G*B*L][128]
  int B, L, G;
  const float* preseq;   // [B][pl][128] or null
  float* inoise;         // [B][pl][128] noise used for the overwrite (rewritten at i == 0)
  int pl;
  const StepCoef* coef;
  const int* d_step;
};

template <int CFD_KI = 0>
__global__ void begin_step_kernel(const BeginArgs a) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread = 8 elements
  const long long n8 = (long long)a.B * a.L * (CFD_LAT / 8);
  if (idx >= n8) return;
  const int c = (int)(idx % (CFD_LAT / 8)) * 8;
  const long long bl = idx / (CFD_LAT / 8);
  const int l = (int)(bl % a.L), b = (int)(bl / a.L);
  float* lp = a.latents + bl * CFD_LAT + c;
  float v[8];
  if (a.preseq && l < a.pl && a.d_step[2] == 0) {   // d_step[2] != 0: cfd_sample_inpaint already did this iteration's overwrite
    const int i = *a.d_step;
    const float sa = a.coef[i].sa, sb = a.coef[i].sb;
    const long long po = ((long long)b * a.pl + l) * CFD_LAT + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = sa * a.preseq[po + e] + sb * a.inoise[po + e];
      lp[e] = v[e];
      if (i == 0) a.inoise[po + e] = v[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = lp[e];
  }
  for (int g = 0; g < a.G; ++g)
    sp_store8(a.sample_sp + (((long long)g * a.B + b) * a.L + l) * (CFD_LAT * 4), c, v);
}

// Modality-guidance combine (convofusion.py:527-541) + scheduler step (diffusers 0.14.0 DDPM / DDIM).
struct CfgStepArgs {
  const float* eps;      // [G*B][L][128]
  float* latents;        // [B][L][128] in/out
  int B, L, G;
  float w[8];            // guidance weight of chunk k (k >= 1); chunk 0 is the unconditional one
  int pos[8];            // chunk k's rows start at row pos[k] * B of eps (the engine may reorder chunks internally)
  int kind;              // 0 DDPM, 1 DDIM
  int clip;
  const StepCoef* coef;
  const int* d_step;
  int* advance;          // non-null: the last workgroup to finish advances the loop (d_step[0] += 1, d_step[2] = 0; d_step[3] is its ticket
                         // counter), so the iteration needs no one-thread kernel behind this one
  const float* noise;    // injected [n_steps][B][L][128] or null -> Philox
  unsigned long long seed;
  unsigned int utt0;
};

template <int CFD_KI = 0>
__global__ void cfg_step_kernel(const CfgStepArgs a) {
  const int per_utt = a.L * CFD_LAT;
  const long long n4 = (long long)a.B * per_utt / 4;
  const int i = *a.d_step;
  // grid-stride over groups of 4 elements: the launch has at most one workgroup per CU, because every workgroup ends with a fence and
  // a ticket (one workgroup per 256 groups was 784 of them at the benchmark shape: most of the kernel's 28 us)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long long)gridDim.x * blockDim.x) {
  const StepCoef c = a.coef[i];
  const long long e0 = idx * 4;
  const long long chunk = (long long)a.B * per_utt;
  // every chunk's prediction is requested before the first is used (chunks past G re-read chunk 0 and are ignored): in a loop
  // of run-time length each load sat next to its use and the G round trips ran one after the other -- 28 us at the benchmark shape
  float4 e4[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e4[k] = *reinterpret_cast<const float4*>(a.eps + (k < a.G ? a.pos[k] : a.pos[0]) * chunk + e0);   // (a select of two argument fields: indexed, pos[] is a dependent scalar load per chunk)
  float4 x4 = *reinterpret_cast<const float4*>(a.latents + e0);
  const float u[4] = {e4[0].x, e4[0].y, e4[0].z, e4[0].w};
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // reference association: ((((text + audio) + spk) + apb) + lsnid) + all, each = (g*w)*(e_k - e_0)
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const float e[4] = {e4[k].x, e4[k].y, e4[k].z, e4[k].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float term = a.w[k] * (e[q] - u[q]);
      acc[q] = (k == 1) ? (a.G > 1 ? term : 0.f) : (k < a.G ? acc[q] + term : acc[q]);
    }
  }
  float x[4] = {x4.x, x4.y, x4.z, x4.w};
  float z[4] = {0.f, 0.f, 0.f, 0.f};
  if (c.use_noise != 0.f) {
    if (a.noise) {
      const float4 n4v = *reinterpret_cast<const float4*>(a.noise + (long long)i * chunk + e0);
      z[0] = n4v.x; z[1] = n4v.y; z[2] = n4v.z; z[3] = n4v.w;
    } else {
      const int b = (int)(e0 / per_utt), g = (int)((e0 % per_utt) / 4);
      const float4 n4v = philox_normal4(a.seed, (uint32_t)g, (uint32_t)i, a.utt0 + b, 0u);
      z[0] = n4v.x; z[1] = n4v.y; z[2] = n4v.z; z[3] = n4v.w;
    }
  }
  float o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float eps = (a.G > 1) ? u[q] + acc[q] : u[q];
    float x0 = (x[q] - c.sb * eps) / c.sa;
    if (a.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    float prev;
    if (a.kind == 0) prev = c.c0 * x0 + c.cx * x[q];
    else prev = c.c0 * x0 + c.cx * eps;
    if (c.use_noise != 0.f) prev = prev + c.sigma * z[q];
    o[q] = prev;
  }
  *reinterpret_cast<float4*>(a.latents + e0) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (a.advance) {   // every thread of every workgroup has read the step index above before the last ticket is taken
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(a.advance + 3), 1u);
      if (t == gridDim.x - 1) {
        a.advance[3] = 0;
        a.advance[2] = 0;
        a.advance[0] = i + 1;
      }
    }
  }
}

// The in-painting overwrite of begin_step_kernel alone, ahead of the captured iteration (cfd_sample_inpaint): the WEG
// branch of the rollout alters the latents AFTER the overwrite and BEFORE the replication (unbounded_synthesis.py:70-143).
template <int CFD_KI = 0>
__global__ void inpaint_now_kernel(const BeginArgs a, int* d_step) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)a.B * a.pl * CFD_LAT;
  if (idx == 0) d_step[2] = 1;
  if (idx >= n) return;
  const int c = (int)(idx % CFD_LAT);
  const long long bl = idx / CFD_LAT;
  const int l = (int)(bl % a.pl), b = (int)(bl / a.pl);
  const int i = *a.d_step;
  const float v = a.coef[i].sa * a.preseq[idx] + a.coef[i].sb * a.inoise[idx];
  a.latents[((long long)b * a.L + l) * CFD_LAT + c] = v;
  if (i == 0) a.inoise[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// out[r][n] = act( b[n] + sum_k x[r][k] W[n][k] ), float32 FMA chain in ascending k (nn.Linear + activation).
// Conditioning producers that run once per batch or once per step on a few thousand rows -- the audio encoder
// (audioenc.py:12-21,33-34), the partner-latent projection of the dyadic path (condfuser.py:22-27) -- so a plain
// LDS-tiled fp32 kernel: 32 rows x 64 outputs per 256-thread block, k in chunks of 32.
//   act: 0 none, 1 GELU (erf form, nn.GELU()), 2 LeakyReLU(0.1)
// ------------------------------------------------------------------------------------------------
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) linear_act_kernel(const float* __restrict__ x, long long n_rows, int K, const float* __restrict__ W,
                                                         const float* __restrict__ b, int N, int act, float* __restrict__ out) {
  __shared__ float xs[32][33];
  __shared__ float ws[64][33];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;               // 4 output columns, 2 rows per thread
  const long long r0 = (long long)blockIdx.y * 32;
  const int n0 = blockIdx.x * 64;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q, rr = e >> 5, kk = e & 31;
      const long long r = r0 + rr;
      xs[rr][kk] = (r < n_rows && k0 + kk < K) ? x[r * K + k0 + kk] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q, nn = e >> 5, kk = e & 31;
      ws[nn][kk] = (n0 + nn < N && k0 + kk < K) ? W[(long long)(n0 + nn) * K + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      const float x0 = xs[ty * 2][kk], x1 = xs[ty * 2 + 1][kk];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float w = ws[tx * 4 + c][kk];
        acc[0][c] = fmaf(x0, w, acc[0][c]);
        acc[1][c] = fmaf(x1, w, acc[1][c]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const long long r = r0 + ty * 2 + rr;
    if (r >= n_rows) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + tx * 4 + c;
      if (n >= N) continue;
      float v = acc[rr][c] + (b ? b[n] : 0.f);
      if (act == 1) v = gelu_f(v);
      else if (act == 2) v = v > 0.f ? v : 0.1f * v;
      out[r * N + n] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Generic float32 row kernels for the small transformers either side of the loop (the VAE decoder, vae.py:268-372:
// d_model 128, 2 heads, a few thousand rows, once per batch).  Plain and exact rather than fast.
// ------------------------------------------------------------------------------------------------
// out[r][:] = (x[r][:] - mean) * rstd * g + b   (nn.LayerNorm, eps 1e-5, biased variance); one wave per row, D <= 2048
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) layernorm_f32_kernel(const float* x, const float* g, const float* b, float* out, long long rows, int D,
                                                            float eps) {
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
    if (c < D) out[row * D + c] = v[q] * rstd * g[c] + b[c];
  }
}

// nn.MultiheadAttention core on projected inputs (F.multi_head_attention_forward after the in-projection):
//   q [Lq][bs][E], k / v [Lk][bs][E] (sequence-major rows, as the reference's [L, N, E] tensors), H heads of hd = E / H <= 64,
//   out[lq][b][h*hd + d] = sum_lk softmax_lk( scale * q.k  (+ -inf where key_padding_mask[b][lk]) ) v[lk][b][h*hd + d]
// One wave per (lq, b, h): lanes <-> keys for the scores (kept in LDS), lanes <-> head features for the weighted sum.
#define MHA_MAX_KEYS 1024
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) mha_f32_kernel(const float* q, const float* k, const float* v, const uint8_t* key_padding_mask, float* out,
                                                      int Lq, int Lk, int bs, int E, int H, float scale) {
  __shared__ float sc[4][MHA_MAX_KEYS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long item = (long long)blockIdx.x * 4 + w;
  if (item >= (long long)Lq * bs * H) return;
  const int h = (int)(item % H);
  const int b = (int)((item / H) % bs);
  const int lq = (int)(item / ((long long)H * bs));
  const int hd = E / H;
  const float* qr = q + ((long long)lq * bs + b) * E + h * hd;
  float mx = -INFINITY;
  for (int lk = lane; lk < Lk; lk += 64) {
    const float* kr = k + ((long long)lk * bs + b) * E + h * hd;
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s = fmaf(qr[d] * scale, kr[d], s);   // q is scaled before the product (q_scaled @ k^T)
    if (key_padding_mask && key_padding_mask[(long long)b * Lk + lk]) s = -INFINITY;
    sc[w][lk] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int lk = lane; lk < Lk; lk += 64) {
    const float p = expf(sc[w][lk] - mx);
    sc[w][lk] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are visible to all of its lanes
  if (lane < hd) {
    float acc = 0.f;
    for (int lk = 0; lk < Lk; ++lk) acc = fmaf(sc[w][lk] / sum, v[((long long)lk * bs + b) * E + h * hd + lane], acc);
    out[((long long)lq * bs + b) * E + h * hd + lane] = acc;
  }
}

// x[i] += y[i]  (residual connections); optional row mask: rows with keep[row] == 0 are set to zero afterwards
template <int CFD_KI = 0>
__global__ void add_f32_kernel(float* x, const float* y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] + y[i];
}
template <int CFD_KI = 0>
__global__ void zero_rows_f32_kernel(float* x, const uint8_t* keep, long long rows, int D) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * D && !keep[i / D]) x[i] = 0.f;
}
