// Split-pair "NT" GEMM for gfx950:  D[i][j] = sum_k X[i][k] * Y[j][k], every product issued as 3 MFMAs
// (lo*hi + hi*lo + hi*hi of the fp16 -- or bf16 -- halves, fp32 accumulate; see cfd_common.hpp).
//
//   X ("row operand", MFMA A) and Y ("column operand", MFMA B) are SP matrices with K contiguous, so both
//   fragments are 16-byte LDS reads.  The result tile is handed to an epilogue functor as 4 consecutive i for
//   one j (the v_mfma_f32_16x16x32 C/D layout: row = 4*(lane>>4)+r, col = lane&15), i.e. outputs are written
//   "out[j][i]" with i contiguous.  X = weight[N][K], Y = activation[M][K] gives out[token][feature]; swapping
//   the roles gives the transposed product (V^T for the P.V products, no transpose pass).
//
//   Block = WI x WJ waves, each wave owns TI x TJ MFMA tiles of 16x16; K-step = 32 (one 128-byte SP line per
//   row: 4 hi chunks + 4 lo chunks of 16 B).  Staging: global_load_lds_dwordx4 (16 B/lane, 8 rows x 128 B per
//   wave-instruction) into an LDS ring; the LDS image is lane-linear, the bank swizzle
//   chunk' = chunk ^ ((row>>1)&7) is applied on the per-lane SOURCE address and again on the fragment read
//   (SQ_LDS_BANK_CONFLICT = 0 measured).  Per K-step a wave issues TI*TJ*3 MFMAs for 2*(TI+TJ) ds_read_b128.
//
//   MODE_PLAIN   one problem (optionally batched over blockIdx.y / .z with strides, an index map or a row list)
//   MODE_GROUPED up to 5 problems sharing Y (cross-attention scores against the 5 memories)
//   MODE_SEGK    one problem whose K range is the concatenation of up to 5 X segments (cross-attention P.V over
//                the 5 memories, accumulated in registers)
//
//   Tile configurations are listed at launch_gemm(); DESIGN.md section 7 has the measurements that chose them.
#pragma once
#include <type_traits>

#include "cfd_common.hpp"

// s_waitcnt immediate (gfx9 encoding): vmcnt = N (bits 3:0 and 15:14), expcnt = 7 (no wait), lgkmcnt = 0
#define WAIT_VM_LGKM0(N) ((((N) & 15) | 0x70 | ((((N) >> 4) & 3) << 14)))
#ifndef CFD_WIDE_EPI
#define CFD_WIDE_EPI 1
#endif
// hipcc moves part of a k-step's MFMAs below the barrier that ends the step (20 of 48 in the 128 x 128 kernel), i.e. in
// front of the NEXT step's LDS-DMA requests: the requests are issued 320 cycles later and the vmcnt(0) wait at the barrier
// is covered by fewer MFMAs.  A scheduling fence keeps the whole cluster in front of the barrier.
#ifndef CFD_MFMA_FENCE
#define CFD_MFMA_FENCE 1
#endif
#ifndef CFD_READS_FIRST
#define CFD_READS_FIRST 1
#endif
#define GEMM_SLOTS 5
enum { MODE_PLAIN = 0, MODE_GROUPED = 1, MODE_SEGK = 2 };

struct GemmArgs {
  const char* X[GEMM_SLOTS];
  long long ldx[GEMM_SLOTS];   // bytes
  long long xbs[GEMM_SLOTS];   // bytes, multiplied by (xmap ? xmap[b] : b)
  const int* xmap[GEMM_SLOTS];
  long long xzs;               // bytes per blockIdx.z
  int I[GEMM_SLOTS];           // D rows (store bound)
  int Iclamp[GEMM_SLOTS];      // valid X rows (loads clamp to Iclamp-1)
  int kt[GEMM_SLOTS];          // k-tiles of 32
  int tiles_i[GEMM_SLOTS];
  int tile_start[GEMM_SLOTS + 1];
  int nslot;
  const char* Y;
  long long ldy, ybs, yzs;     // bytes
  int J, Jclamp, tiles_j;
  int super_i, super_j;        // MODE_PLAIN: > 0 selects the super-tiled block order (see kernel)
  const int* brow;             // optional: batch index -> effective batch row (launches over a subset of rows)
  int yk0[GEMM_SLOTS];         // MODE_SEGK: first k-tile of segment s inside Y (segments need not be adjacent)
};

// compile-time-indexed select from a kernel-argument array (a runtime index would force the
// by-value argument struct into scratch memory)
template <class T>
__device__ __forceinline__ T sel5(const T (&arr)[GEMM_SLOTS], int g) {
  T v = arr[0];
#pragma unroll
  for (int q = 1; q < GEMM_SLOTS; ++q)
    if (g == q) v = arr[q];
  return v;
}

// ------------------------------------------------------------------------------------------------
// Epilogues.  operator()(g, b, z, i, j, v): v = D[i..i+3][j] for group g, batch b, z.
// ------------------------------------------------------------------------------------------------
struct EpiF32 {  // out_f32[j][goff+i] = v (+ bias[i] | + key-bias of group g)
  float* out;
  long long ldo, obs, ozs;  // floats
  const float* bias;
  int goff[GEMM_SLOTS];
  const float* gbias[GEMM_SLOTS];
  const int* gmap[GEMM_SLOTS];
  long long gstride[GEMM_SLOTS];
  static constexpr bool kPrefetch = true;   // operands of the epilogue are loaded before the K loop (see kernel)
  typedef float4 Pre;
  __device__ __forceinline__ Pre prefetch(int g, int b, int z, int i, int j) const {
    const float* bb = nullptr;
    const float* gb = sel5(gbias, g);
    if (gb) {
      const int* gm = sel5(gmap, g);
      bb = gb + (long long)(gm ? gm[b] : b) * sel5(gstride, g) + i;
    } else if (bias) bb = bias + i;
    return bb ? *reinterpret_cast<const float4*>(bb) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v, const Pre& t) const {
    float* p = out + (long long)b * obs + (long long)z * ozs + (long long)j * ldo + sel5(goff, g) + i;
    *reinterpret_cast<float4*>(p) = make_float4(v[0] + t.x, v[1] + t.y, v[2] + t.z, v[3] + t.w);
  }
};

struct EpiSplit {  // out_sp[j][coloff + i] = split(act(v + bias[i]))
  char* out;
  long long ldo, obs, ozs;  // bytes
  const float* bias;
  int gelu;
  int perm32;  // store column 32S+16h+4q+r at position 32S+8q+4h+r (k-slot order of attn_fused.hpp's P fragments)
  static constexpr bool kPrefetch = false;
  // wide epilogue: a lane gets 8 consecutive i (i % 8 == 0) -> one 16-byte hi and one 16-byte lo store instead of
  // two 8-byte pairs (half the store instructions for the same bytes)
  static constexpr bool kStore8 = true;
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {
    if (bias) {
      const float4 t = *reinterpret_cast<const float4*>(bias + i);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    if (gelu) { v[0] = gelu_fast_f(v[0]); v[1] = gelu_fast_f(v[1]); v[2] = gelu_fast_f(v[2]); v[3] = gelu_fast_f(v[3]); }
    const int col = perm32 ? ((i & ~31) | (((i >> 2) & 3) << 3) | (((i >> 4) & 1) << 2)) : i;
    sp_store4(out + (long long)b * obs + (long long)z * ozs + (long long)j * ldo, col, v[0], v[1], v[2], v[3]);
  }
  // the bias of a lane's 8 columns is read once per tile (kBias8): re-read behind every store, the load's wait also
  // waits for the store before it
  static constexpr bool kBias8 = true;
  __device__ __forceinline__ void tile_bias8(int i, float4& t0, float4& t1) const {
    t0 = t1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) {
      t0 = *reinterpret_cast<const float4*>(bias + i);
      t1 = *reinterpret_cast<const float4*>(bias + i + 4);
    }
  }
  __device__ __forceinline__ void store8(int g, int b, int z, int i, int j, f32x4 v0, f32x4 v1, float4 t0, float4 t1) const {
    float v[8] = {v0[0] + t0.x, v0[1] + t0.y, v0[2] + t0.z, v0[3] + t0.w, v1[0] + t1.x, v1[1] + t1.y, v1[2] + t1.z, v1[3] + t1.w};
    if (gelu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_fast_f(v[e]);
    }
    char* row = out + (long long)b * obs + (long long)z * ozs + (long long)j * ldo;
    if (perm32) {   // the k-slot permutation moves groups of 4: two 4-wide stores
      const int c0 = (i & ~31) | (((i >> 2) & 3) << 3) | (((i >> 4) & 1) << 2);
      const int c1 = ((i + 4) & ~31) | ((((i + 4) >> 2) & 3) << 3) | ((((i + 4) >> 4) & 1) << 2);
      sp_store4(row, c0, v[0], v[1], v[2], v[3]);
      sp_store4(row, c1, v[4], v[5], v[6], v[7]);
    } else {
      sp_store8_out(row, i, v);
    }
  }
};

// q | k and v^T of the self-attention in ONE grouped launch for batch rows of exactly 16 tokens (the product shape; round 5).
// Group 0 (X = [Wq; Wk], 1024 features): out_qk[j][i] = split(v + bias[i]), as EpiSplit.  Group 1 (X = Wv, 512 features): the value
// projection stored TRANSPOSED, vts[j / 16][i][perm32(j % 16)] -- what the separate batched product with swapped operand roles
// (X = the row's tokens, Y = Wv, EpiSplit::perm32) wrote, but that launch works on half-empty 32-token tiles (20.7 us beside the
// q | k product's 21 at 32 utterances).  The wide epilogue has a wave's band of 16 tokens x 16 TI features in its LDS strip anyway:
// read by columns, a lane gets 8 consecutive tokens of one feature.  Token t = 4 q + r of a 16-token row sits at key position
// 8 q + r of the 32-key block (attn_fused.hpp), so tokens 8 h .. 8 h + 7 fill positions [16 h, 16 h + 4) and [16 h + 8, 16 h + 12);
// the gaps are the padding keys 16 .. 31, stored as zeros (the kernel masks their scores; their values only have to be finite).
struct EpiQkvT {
  char* out_qk;          // SP [M][1024]
  long long ldo;         // bytes per q | k row
  const float* bias;     // [1024] (the value bias is folded into the out-projection's)
  char* vts;             // SP [Be][512][32]
  int natural;           // 1: keys in natural order (rt_selfattn_kernel, rowtile.hpp): tokens 8 h .. 8 h + 7 at positions [8 h, 8 h + 8), zeros from 16 on
  static constexpr bool kPrefetch = false;
  static constexpr bool kStore8 = true;
  static constexpr bool kBias8 = true;
  static constexpr bool kStoreT = true;   // group 1 is stored by store_t
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {
    const float4 t = *reinterpret_cast<const float4*>(bias + i);
    sp_store4(out_qk + (long long)j * ldo, i, v[0] + t.x, v[1] + t.y, v[2] + t.z, v[3] + t.w);
  }
  __device__ __forceinline__ void tile_bias8(int i, float4& t0, float4& t1) const {   // (group 1 reads it too and ignores it: i < 512 there)
    t0 = *reinterpret_cast<const float4*>(bias + i);
    t1 = *reinterpret_cast<const float4*>(bias + i + 4);
  }
  __device__ __forceinline__ void store8(int g, int b, int z, int i, int j, f32x4 v0, f32x4 v1, float4 t0, float4 t1) const {
    const float v[8] = {v0[0] + t0.x, v0[1] + t0.y, v0[2] + t0.z, v0[3] + t0.w, v1[0] + t1.x, v1[1] + t1.y, v1[2] + t1.z, v1[3] + t1.w};
    sp_store8_out(out_qk + (long long)j * ldo, i, v);
  }
  // feature i of the value projection, tokens jb + 8 h .. + 7 (jb % 16 == 0): 16 key positions = 32 bytes of the hi plane and of the lo plane
  __device__ __forceinline__ void store_t(int i, int jb, int h, const float* v) const {
    spx8 hi0, lo0, hi1, lo1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sp_t a, c;
      split_f32(v[e], a, c);
      hi0[e] = a; lo0[e] = c;
      split_f32(v[4 + e], a, c);
      hi1[e] = a; lo1[e] = c;
      hi0[4 + e] = (sp_t)0.f; lo0[4 + e] = (sp_t)0.f; hi1[4 + e] = (sp_t)0.f; lo1[4 + e] = (sp_t)0.f;
    }
    if (natural) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi0[4 + e] = hi1[e]; lo0[4 + e] = lo1[e];      // tokens 8 h .. 8 h + 7 side by side ...
        hi1[e] = (sp_t)0.f; lo1[e] = (sp_t)0.f;        // ... and 8 padding keys
      }
      char* p = vts + ((long long)(jb >> 4) * CFD_D + i) * 128 + h * 16;
      *reinterpret_cast<spx8*>(p) = hi0;
      *reinterpret_cast<spx8*>(p + 32) = hi1;
      *reinterpret_cast<spx8*>(p + 64) = lo0;
      *reinterpret_cast<spx8*>(p + 96) = lo1;
      return;
    }
    char* p = vts + ((long long)(jb >> 4) * CFD_D + i) * 128 + h * 32;
    *reinterpret_cast<spx8*>(p) = hi0;
    *reinterpret_cast<spx8*>(p + 16) = hi1;
    *reinterpret_cast<spx8*>(p + 64) = lo0;
    *reinterpret_cast<spx8*>(p + 80) = lo1;
  }
};

struct EpiResid {  // x[(b*rows_per_b + j)][i] += v + bias[i]   (row length CFD_D floats)
  float* x;
  long long obs;  // floats per batch
  const float* bias;
  // (Reading the residual rows before the K loop was tried: +46 VGPRs and no gain -- 92 vs 86 us on the
  //  43904x512x512 product; the epilogue is bandwidth-, not latency-limited.)
  static constexpr bool kPrefetch = false;
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {
    float* p = x + (long long)b * obs + (long long)j * CFD_D + i;
    float4 r = *reinterpret_cast<const float4*>(p);
    if (bias) {
      const float4 t = *reinterpret_cast<const float4*>(bias + i);
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    r.x += v[0]; r.y += v[1]; r.z += v[2]; r.w += v[3];
    *reinterpret_cast<float4*>(p) = r;
  }
  // Band form used by the wide epilogue: the old values of a 16-row band are requested together and the bias is
  // read once per tile.  (Written item by item, hipcc serialises load -> wait -> store for every 16 bytes -- a store
  // to x may alias the next load from x -- and re-reads the bias behind every store: 32 dependent round trips a tile.)
  static constexpr bool kBand = true;
  __device__ __forceinline__ float4 tile_bias(int i) const {
    return bias ? *reinterpret_cast<const float4*>(bias + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ float4 band_load(int g, int b, int z, int i, int j) const {
    return *reinterpret_cast<const float4*>(x + (long long)b * obs + (long long)j * CFD_D + i);
  }
  __device__ __forceinline__ void band_store(int g, int b, int z, int i, int j, f32x4 v, float4 r, float4 t) const {
    r.x = (r.x + t.x) + v[0]; r.y = (r.y + t.y) + v[1]; r.z = (r.z + t.z) + v[2]; r.w = (r.w + t.w) + v[3];   // same association as above
    *reinterpret_cast<float4*>(x + (long long)b * obs + (long long)j * CFD_D + i) = r;
  }
};

// ---- The algebraic LayerNorm fold (round 6; mid-size problems: cfd_forward.hip `ln_fold`) -------------------------
//   W LN(x) + b = r_sigma (W' x - mu c) + d + b,    W' = W diag(gamma),  c = W' 1,  d = W beta
// The PRODUCER of x (a residual product: EpiResidStat) also stores the split-pair copy of its new rows -- the consumer's operand -- and, per
// row and 32-column slot, the slot's mean and sum of squared deviations (16 slots a row, no atomics).  The CONSUMER (any epilogue wrapped
// in EpiLn) puts a row's statistics together from the 16 slots (Chan's pairwise update with equal counts) while its first operand tiles
// are on their way, and rescales its accumulators in front of its own epilogue.  No ln_rows launch in between.
#define LN_SLOTS 16          // slots per row (CFD_D / 32)
// Sum over the aligned group of 8 consecutive lanes a lane belongs to, in the vector ALU (DPP: quad permutes, then the half-row mirror).
__device__ __forceinline__ float lane_group8_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1, 0, 3, 2]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2, 3, 0, 1]
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return x;
}
struct EpiResidStat {   // EpiResid + xs[row] = split(x_new[row]) + stat[row][slot] = (mean, M2) of the slot's 32 columns
  float* x;
  long long obs;  // floats per batch
  const float* bias;
  char* xs;       // SP [rows][512]
  float* stat;    // [rows][LN_SLOTS][2]
  static constexpr bool kPrefetch = false;
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {   // (narrow epilogue / naive kernel: not used with the fold)
    float* p = x + (long long)b * obs + (long long)j * CFD_D + i;
    float4 r = *reinterpret_cast<const float4*>(p);
    if (bias) {
      const float4 t = *reinterpret_cast<const float4*>(bias + i);
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    r.x += v[0]; r.y += v[1]; r.z += v[2]; r.w += v[3];
    *reinterpret_cast<float4*>(p) = r;
  }
  static constexpr bool kBand = true;
  static constexpr bool kRowStat = true;
  __device__ __forceinline__ float4 tile_bias(int i) const {
    return bias ? *reinterpret_cast<const float4*>(bias + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ float4 band_load(int g, int b, int z, int i, int j) const {
    return *reinterpret_cast<const float4*>(x + (long long)b * obs + (long long)j * CFD_D + i);
  }
  // every lane of the wave takes part (the 8 lanes of a slot reduce in the vector ALU); `valid` = the lane's (i, j) is inside the matrix
  __device__ __forceinline__ void band_store_stat(int g, int b, int z, int i, int j, f32x4 v, float4 r, float4 t, bool valid) const {
    r.x = (r.x + t.x) + v[0]; r.y = (r.y + t.y) + v[1]; r.z = (r.z + t.z) + v[2]; r.w = (r.w + t.w) + v[3];   // same association as EpiResid
    const long long row = (long long)b * obs / CFD_D + j;
    if (valid) {
      *reinterpret_cast<float4*>(x + row * CFD_D + i) = r;
      sp_store4(xs + row * (CFD_D * 4), i, r.x, r.y, r.z, r.w);
    }
    const float mean = lane_group8_sum((r.x + r.y) + (r.z + r.w)) * (1.0f / 32.0f);
    const float dx = r.x - mean, dy = r.y - mean, dz = r.z - mean, dw = r.w - mean;
    const float m2 = lane_group8_sum((dx * dx + dy * dy) + (dz * dz + dw * dw));
    if (valid && (threadIdx.x & 7) == 0) *reinterpret_cast<float2*>(stat + (row * LN_SLOTS + (i >> 5)) * 2) = make_float2(mean, m2);
  }
  __device__ __forceinline__ void band_store(int g, int b, int z, int i, int j, f32x4 v, float4 r, float4 t) const {   // (unused: the kernel calls band_store_stat)
    band_store_stat(g, b, z, i, j, v, r, t, true);
  }
};
// An epilogue E whose product runs on the RAW rows of x (split pairs stored by EpiResidStat) against W' = W diag(gamma): gemm_sp_body
// rescales the accumulators with the rows' statistics before E sees them.  Group g of a grouped launch has its own c / d vectors.
template <class E>
struct EpiLn : E {
  const float* ln_stat;      // [rows][LN_SLOTS][2]; rows indexed like the launch's j (un-batched launches only)
  const float* ln_c[2];      // c = W' 1 per output feature of group g
  const float* ln_d[2];      // d = W beta
  float ln_eps;
  static constexpr bool kLnFold = true;
};

struct EpiNull {  // timing experiments only: keeps the accumulators live, stores nothing
  static constexpr bool kPrefetch = false;
  float* sink;
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {
    asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    if (i < -1) sink[0] = v[0];
  }
};

struct EpiEmbed {  // x0[j][i] = v + bias[i] + bh[(l&1)][i] + qpe[(l>>1)][i],  l = j % L   (denoiser.py:187,316-326)
  float* x;
  const float* bias;
  const float* bh;   // [2][512]
  const float* qpe;  // [>=L/2][512]
  int L;
  static constexpr bool kPrefetch = false;
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {
    const int l = j % L;
    const float4 t0 = *reinterpret_cast<const float4*>(bias + i);
    const float4 t1 = *reinterpret_cast<const float4*>(bh + (l & 1) * CFD_D + i);
    const float4 t2 = *reinterpret_cast<const float4*>(qpe + (long long)(l >> 1) * CFD_D + i);
    // same association as the reference: ((linear + bh) + pe)
    float4 r;
    r.x = ((v[0] + t0.x) + t1.x) + t2.x;
    r.y = ((v[1] + t0.y) + t1.y) + t2.y;
    r.z = ((v[2] + t0.z) + t1.z) + t2.z;
    r.w = ((v[3] + t0.w) + t1.w) + t2.w;
    *reinterpret_cast<float4*>(x + (long long)j * CFD_D + i) = r;
  }
};

struct EpiMemK {  // i < nfeat: K_layer[i/512][j][i%512] = split(v)  (one contiguous [rows][512] SP matrix per layer);
                  // nfeat <= i < nfeat+nl: cbias[i-nfeat][j] = v
  char* kall;
  long long rows;  // memory rows J (= cbias row length)
  float* cbias;
  int nfeat, nl;
  // dead keys (padding rows s >= S of a memory, key-padding mask) get key bias -inf: the fused cross-attention kernel
  // then needs no mask loads, and the three-launch path's softmax masks them (again) anyway
  const uint8_t* mask;   // [U][S] (1 = padded key); never null
  int S, Sp;
  unsigned int* sat;     // the handle's saturation census (cfd_common.hpp), or null
  static constexpr bool kPrefetch = false;
  static constexpr bool kStore8 = true;
  __device__ __forceinline__ void store8(int g, int b, int z, int i, int j, f32x4 v0, f32x4 v1) const {
    if (i + 8 <= nfeat) {   // (nfeat is a multiple of 512: 8 consecutive features never straddle a layer)
      const int layer = i >> 9, o = i & (CFD_D - 1);
      const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      sat_note<8>(sat, v);
      sp_store8(kall + ((long long)layer * rows + j) * (CFD_D * 4), o, v);
    } else {
      (*this)(g, b, z, i, j, v0);
      (*this)(g, b, z, i + 4, j, v1);
    }
  }
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {
    if (i < nfeat) {
      const int layer = i >> 9, o = i & (CFD_D - 1);
      const float vv[4] = {v[0], v[1], v[2], v[3]};
      sat_note<4>(sat, vv);
      sp_store4(kall + ((long long)layer * rows + j) * (CFD_D * 4), o, v[0], v[1], v[2], v[3]);
    } else {
      const int u = j / Sp, sk = j - u * Sp;
      const bool dead = sk >= S || mask[(long long)u * S + min(sk, S - 1)] != 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i - nfeat + e < nl) cbias[(long long)(i - nfeat + e) * rows + j] = dead ? -INFINITY : v[e];
    }
  }
};

struct EpiMemV {  // V^T_layer_u[j/512][i/Sp][j%512][i%Sp] = split(v): one contiguous [512][Sp] SP block per (layer, memory)
  char* vt;
  int Sp, U;
  unsigned int* sat;     // the handle's saturation census, or null
  static constexpr bool kPrefetch = false;
  static constexpr bool kStore8 = true;
  __device__ __forceinline__ void store8(int g, int b, int z, int i, int j, f32x4 v0, f32x4 v1) const {
    const int layer = j >> 9, f = j & (CFD_D - 1);
    const int u = i / Sp, s = i - u * Sp;   // (Sp is a multiple of 32: 8 consecutive rows belong to one memory)
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    sat_note<8>(sat, v);
    sp_store8(vt + (((long long)layer * U + u) * CFD_D + f) * ((long long)Sp * 4), s, v);
  }
  __device__ __forceinline__ void operator()(int g, int b, int z, int i, int j, f32x4 v) const {
    const int layer = j >> 9, f = j & (CFD_D - 1);
    const int u = i / Sp, s = i - u * Sp;
    const float vv[4] = {v[0], v[1], v[2], v[3]};
    sat_note<4>(sat, vv);
    sp_store4(vt + (((long long)layer * U + u) * CFD_D + f) * ((long long)Sp * 4), s, v[0], v[1], v[2], v[3]);
  }
};

template <class E, class = void> struct EpiHasBias8 { static constexpr bool value = false; };
template <class E> struct EpiHasBias8<E, typename std::enable_if<E::kBias8>::type> { static constexpr bool value = true; };
template <class E, class = void> struct EpiHasBand { static constexpr bool value = false; };
template <class E> struct EpiHasBand<E, typename std::enable_if<E::kBand>::type> { static constexpr bool value = true; };
template <class E, class = void> struct EpiHasStore8 { static constexpr bool value = false; };
template <class E> struct EpiHasStore8<E, typename std::enable_if<E::kStore8>::type> { static constexpr bool value = true; };
template <class E, class = void> struct EpiHasStoreT { static constexpr bool value = false; };
template <class E> struct EpiHasStoreT<E, typename std::enable_if<E::kStoreT>::type> { static constexpr bool value = true; };
template <class E, class = void> struct EpiHasRowStat { static constexpr bool value = false; };
template <class E> struct EpiHasRowStat<E, typename std::enable_if<E::kRowStat>::type> { static constexpr bool value = true; };
template <class E, class = void> struct EpiHasLnFold { static constexpr bool value = false; };
template <class E> struct EpiHasLnFold<E, typename std::enable_if<E::kLnFold>::type> { static constexpr bool value = true; };
template <class E, bool P = E::kPrefetch> struct EpiPre { struct type {}; };
template <class E> struct EpiPre<E, true> { typedef typename E::Pre type; };

// ------------------------------------------------------------------------------------------------
// NSTAGE == 2: one k-tile of prefetch, __syncthreads() per k-step (small / ragged tile configs).
// NSTAGE == 3: two k-tiles of prefetch kept in flight ACROSS the per-k-step barrier: counted
//              s_waitcnt vmcnt(GPW) + raw s_barrier (a __syncthreads() would drain the LDS-DMA queue).
//              Requires every wave to issue exactly GPW loads per stage.
// The body takes its position in the launch as arguments (linear block id `lin_in` of an nx x ny x nz grid): gemm_sp_kernel passes
// blockIdx / gridDim.
template <int WI, int WJ, int TI, int TJ, int NSTAGE, int MODE, class Epi>
__device__ __forceinline__ void gemm_sp_body(const GemmArgs& a, const Epi& epi, const long long lin_in, const int nx_in, const int ny_in, const int nz_in) {
  constexpr int NW = WI * WJ;
  constexpr int BI = WI * TI * 16, BJ = WJ * TJ * 16;
  constexpr int STAGE = (BI + BJ) * 128;
  constexpr int XG = BI / 8, YG = BJ / 8;                // 8-row load groups per operand tile
  constexpr int XPW = (XG + NW - 1) / NW, YPW = (YG + NW - 1) / NW;
  constexpr int GPW = XPW + YPW;
  static_assert(NSTAGE == 2 || NSTAGE == 3, "2-stage (one barrier per k-step) or 3-stage (counted vmcnt) loop");
  static_assert(NSTAGE != 3 || (XG % NW == 0 && YG % NW == 0), "3-stage pipeline needs uniform load counts per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wid / WJ, wj = wid % WJ;
  // XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs in LINEAR dispatch order
  // (x fastest, then y, z), and each XCD has a private L2.  Blocks that land on one XCD (linear id % 8 equal)
  // are given a contiguous range of the (batch, tile) space, so the tiles that re-read one activation tile --
  // and, for batched launches, all tiles of one batch row -- share an L2 instead of each fetching from HBM.
  int t, b, z;
  {
    const int nx = nx_in, ny = ny_in;
    const long long total = (long long)nx * ny * nz_in;
    const long long lin = lin_in;
    const long long q = total >> 3, k = lin >> 3;
    const int r = (int)(total & 7), xcd = (int)(lin & 7);
    const long long v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    t = (int)(v % nx);
    const long long bz = v / nx;
    b = (int)(bz % ny);
    z = (int)(bz / ny);
    if (a.brow) b = a.brow[b];
  }
  int g = 0;
  if (MODE == MODE_GROUPED) {
#pragma unroll
    for (int s = 1; s < GEMM_SLOTS; ++s)
      if (s < a.nslot && t >= a.tile_start[s]) g = s;
    int ts = 0;
#pragma unroll
    for (int s = 1; s < GEMM_SLOTS; ++s)
      if (g == s) ts = a.tile_start[s];
    t -= ts;
  }
  const int tiles_i = sel5(a.tiles_i, g);
  int ti_blk = t % tiles_i, tj_blk = t / tiles_i;
  if (MODE == MODE_PLAIN && a.super_i > 0) {
    // both operands exceed an XCD's 4 MiB L2 (memory-side projections: 9.5 MB of weights x 100 MB of rows):
    // walk the tile grid in super_i x super_j super-tiles so each panel is fetched once per super-tile row/column
    // instead of once per tile (measured 3.9 GB -> ~1 GB of fetch per launch)
    const int SI = a.super_i, SJ = a.super_j;
    const int nsi = (tiles_i + SI - 1) / SI;
    const int per_band = tiles_i * SJ;                 // tiles in one band of SJ tile-columns
    const int band = t / per_band, r = t - band * per_band;
    const int wj = min(SJ, a.tiles_j - band * SJ);     // columns in this (possibly last, narrower) band
    const int per_super = SI * wj;
    int sidx = r / per_super, rr = r - sidx * per_super;
    if (sidx >= nsi) { sidx = nsi - 1; rr = r - sidx * per_super; }
    const int hi = min(SI, tiles_i - sidx * SI);       // rows in this (possibly last, shorter) super-tile
    // tiles before this super-tile in the band: sidx * SI * wj (all full-height super-tiles)
    rr = r - sidx * SI * wj;
    ti_blk = sidx * SI + rr % hi;
    tj_blk = band * SJ + rr / hi;
  }
  const int i0 = ti_blk * BI, j0 = tj_blk * BJ;
  const int Ig = sel5(a.I, g);
  const int Iclamp_g = sel5(a.Iclamp, g);

  // X base for this (b, z): one slot in PLAIN / GROUPED mode, one per K segment in SEGK mode
  // (five named scalars, not an array: hipcc turns a select chain over a small local array into an indexed SCRATCH
  //  load in front of every k-step's requests)
  static_assert(GEMM_SLOTS == 5, "the K-segment bases are five named scalars");
  const char *xb0 = nullptr, *xb1 = nullptr, *xb2 = nullptr, *xb3 = nullptr, *xb4 = nullptr;
  const char* xbase_g = nullptr;
  long long ldx_g = 0;
  if (MODE == MODE_SEGK) {
    auto seg_base = [&](int s) __attribute__((always_inline)) -> const char* {
      if (s >= a.nslot) return nullptr;
      const long long bi = a.xmap[s] ? a.xmap[s][b] : b;
      return a.X[s] + bi * a.xbs[s] + (long long)z * a.xzs;
    };
    xb0 = seg_base(0); xb1 = seg_base(1); xb2 = seg_base(2); xb3 = seg_base(3); xb4 = seg_base(4);
  } else {
    const int* xm = sel5(a.xmap, g);
    const long long bi = xm ? xm[b] : b;
    xbase_g = sel5(a.X, g) + bi * sel5(a.xbs, g) + (long long)z * a.xzs;
    ldx_g = sel5(a.ldx, g);
  }
  const char* yb = a.Y + (long long)b * a.ybs + (long long)z * a.yzs;

  // staging bookkeeping: wave `wid` loads X groups wid, wid+NW, ... and Y groups wid, wid+NW, ...
  // (a group = 8 tile rows x 128 B = one global_load_lds_dwordx4 wave-instruction)
  const int cpos = lane & 7, rsub = lane >> 3;
  int xrow[XPW], xchunk[XPW];
  long long yoff[YPW];
#pragma unroll
  for (int n = 0; n < XPW; ++n) {
    const int r = (wid + NW * n) * 8 + rsub;
    xrow[n] = min(i0 + r, Iclamp_g - 1);
    xchunk[n] = (cpos ^ ((r >> 1) & 7)) << 4;
  }
#pragma unroll
  for (int n = 0; n < YPW; ++n) {
    const int r = (wid + NW * n) * 8 + rsub;
    yoff[n] = (long long)min(j0 + r, a.Jclamp - 1) * a.ldy + ((cpos ^ ((r >> 1) & 7)) << 4);
  }
  long long xoff[XPW];
  if (MODE != MODE_SEGK) {
#pragma unroll
    for (int n = 0; n < XPW; ++n) xoff[n] = (long long)xrow[n] * ldx_g + xchunk[n];
  }

  int nkt = 0;
  if (MODE == MODE_SEGK) {
#pragma unroll
    for (int s = 0; s < GEMM_SLOTS; ++s)
      if (s < a.nslot) nkt += a.kt[s];
  } else {
    nkt = sel5(a.kt, g);
  }

  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
    char* sbuf = smem + buf * STAGE;
    int ykt = kt;
    if (MODE == MODE_SEGK) {
      int s = 0, ktl = kt;
#pragma unroll
      for (int q = 0; q < GEMM_SLOTS - 1; ++q)
        if (s == q && q < a.nslot - 1 && ktl >= a.kt[q]) { ktl -= a.kt[q]; s = q + 1; }
      const char* xs = s == 0 ? xb0 : s == 1 ? xb1 : s == 2 ? xb2 : s == 3 ? xb3 : xb4;
      long long ldx = a.ldx[0];
      int yk = a.yk0[0];
#pragma unroll
      for (int q = 1; q < GEMM_SLOTS; ++q)
        if (q == s) { ldx = a.ldx[q]; yk = a.yk0[q]; }
      ykt = yk + ktl;
#pragma unroll
      for (int n = 0; n < XPW; ++n) {
        const int gi = wid + NW * n;
        if (XG % NW == 0 || gi < XG)
          __builtin_amdgcn_global_load_lds((gptr_t)(xs + (long long)xrow[n] * ldx + (long long)ktl * 128 + xchunk[n]),
                                           (lptr_t)(sbuf + gi * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int n = 0; n < XPW; ++n) {
        const int gi = wid + NW * n;
        if (XG % NW == 0 || gi < XG)
          __builtin_amdgcn_global_load_lds((gptr_t)(xbase_g + xoff[n] + (long long)kt * 128), (lptr_t)(sbuf + gi * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < YPW; ++n) {
      const int gi = wid + NW * n;
      if (YG % NW == 0 || gi < YG)
        __builtin_amdgcn_global_load_lds((gptr_t)(yb + yoff[n] + (long long)ykt * 128), (lptr_t)(sbuf + BI * 128 + gi * 1024), 16, 0, 0);
    }
  };

  // fragment read offsets
  const int l15 = lane & 15, q4 = lane >> 4;
  const int sw = l15 >> 1;
  const int xoff_h = (wi * TI * 16 + l15) * 128 + ((q4 ^ sw) << 4);
  const int xoff_l = (wi * TI * 16 + l15) * 128 + (((4 + q4) ^ sw) << 4);
  const int yoff_h = BI * 128 + (wj * TJ * 16 + l15) * 128 + ((q4 ^ sw) << 4);
  const int yoff_l = BI * 128 + (wj * TJ * 16 + l15) * 128 + (((4 + q4) ^ sw) << 4);

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};

  // epilogue operands (residual rows / bias vectors) are requested now and consumed after the K loop
  struct NoPre {};
  typename std::conditional<Epi::kPrefetch, typename EpiPre<Epi>::type, NoPre>::type pre[TI][TJ];
  // Epilogue lane mapping.  Native MFMA layout: a lane owns 4 consecutive i of ONE row j, so a wave store touches 16
  // rows x 64 B (16 half cache lines).  With CFD_WIDE_EPI the accumulator tile goes through LDS once and is handed
  // out again with LPR = 4*TI lanes per row: a wave store then covers 64/LPR rows x (TI*64) contiguous bytes.
  constexpr bool WIDE = CFD_WIDE_EPI && (TI == 2 || TI == 4 || TI == 8);
  constexpr int LPR = TI * 4;            // lanes per output row (wide mapping)
  constexpr int RPI = 64 / LPR;          // rows per wave instruction
  constexpr int NIT = 16 / RPI;          // instructions per 16-row tile band
  auto epi_ij = [&](int tj, int k, int& i, int& j) __attribute__((always_inline)) {   // k: ti (native) or it (wide)
    if (WIDE) {
      i = i0 + wi * TI * 16 + (lane % LPR) * 4;
      j = j0 + (wj * TJ + tj) * 16 + k * RPI + lane / LPR;
    } else {
      i = i0 + (wi * TI + k) * 16 + q4 * 4;
      j = j0 + (wj * TJ + tj) * 16 + l15;
    }
  };
  if constexpr (Epi::kPrefetch) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        int i, j;
        epi_ij(tj, ti, i, j);   // WIDE: NIT == TI, so the [TI][TJ] array has exactly one slot per instruction
        pre[ti][tj] = epi.prefetch(g, b, z, min(i, Ig - 4), min(j, a.J - 1));
      }
  }

  // the bias of a lane's 8 output columns (split-pair stores): requested here, so that its round trip runs under the K loop -- it used
  // to be the first thing the epilogue waited for.  (-DCFD_BIAS8_LATE=1: the old place, for the A/B.)
#ifndef CFD_BIAS8_LATE
#define CFD_BIAS8_LATE 0
#endif
  float4 s8_t0 = make_float4(0.f, 0.f, 0.f, 0.f), s8_t1 = s8_t0;
  if constexpr (WIDE && !CFD_BIAS8_LATE && EpiHasStore8<Epi>::value && EpiHasBias8<Epi>::value) {
    const int i8 = i0 + wi * TI * 16 + (lane % (TI * 2)) * 8;   // (a lane's columns do not depend on the band)
    epi.tile_bias8(min(i8, Ig - 8), s8_t0, s8_t1);
  }

  // LayerNorm fold (EpiLn): thread t < BJ requests the 16 (mean, M2) slots of row j0 + t now; ln_finish() -- called behind the first
  // stage requests -- puts them together and parks (mu, r_sigma) behind the staging ring, where the epilogue reads them.
#ifndef LNF_ABL
#define LNF_ABL 0   // developer timing experiments (results are garbage): 1 = no statistics loads / combine, 2 = no rescale, 4 = no c / d loads
#endif
  constexpr bool LNF = EpiHasLnFold<Epi>::value;
  static_assert(!LNF || MODE != MODE_SEGK, "LayerNorm fold: plain or grouped launches");
  // The statistics are requested by inline assembly and waited for by hand in ln_finish: as ordinary loads hipcc drained the whole queue for them
  // TWICE in the prologue (behind the first stage's requests and again behind the second's), i.e. the first two stages landed one after the other --
  // +1.6 us on every consumer launch.
  f32x4 ln_raw[LNF ? LN_SLOTS / 2 : 1];
  float4 ln_c4[LNF ? TI : 1], ln_d4[LNF ? TI : 1];
  if constexpr (LNF) {
    if (!(LNF_ABL & 1) && (int)threadIdx.x < BJ) {
      const char* sp = reinterpret_cast<const char*>(epi.ln_stat + (long long)min(j0 + (int)threadIdx.x, a.J - 1) * (LN_SLOTS * 2));
#pragma unroll
      for (int q = 0; q < LN_SLOTS / 2; ++q) {
        const char* gp = sp + q * 16;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(ln_raw[q]) : "v"(gp) : "memory");
      }
    }
    // c = W' 1 and d = W beta of this lane's features: requested in front of the stage requests (in-order completion: behind them, the
    // counted waits of the loop would wait for them and with them for the stage in front), used behind the K loop
    const float* cg = g == 0 ? epi.ln_c[0] : epi.ln_c[1];
    const float* dg = g == 0 ? epi.ln_d[0] : epi.ln_d[1];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const int i = (LNF_ABL & 4) ? 0 : min(i0 + (wi * TI + ti) * 16 + q4 * 4, Ig - 4);
      if (LNF_ABL & 4) { ln_c4[ti] = ln_d4[ti] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
      ln_c4[ti] = *reinterpret_cast<const float4*>(cg + i);
      ln_d4[ti] = *reinterpret_cast<const float4*>(dg + i);
    }
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks half of these requests behind the first stage's)
  }
  // `younger`: the vector-memory requests this wave has issued since (the stage requests in front of the call)
  auto ln_finish = [&](auto younger) __attribute__((always_inline)) {
    if constexpr (LNF) {
      if constexpr (!(LNF_ABL & 1)) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(younger)::value) : "memory");
        if ((int)threadIdx.x < BJ) {
#pragma unroll
          for (int q = 0; q < LN_SLOTS / 2; ++q) asm volatile("" : "+v"(ln_raw[q]));
          float ms = 0.f, m2 = 0.f;
#pragma unroll
          for (int q = 0; q < LN_SLOTS / 2; ++q) { ms += ln_raw[q][0] + ln_raw[q][2]; m2 += ln_raw[q][1] + ln_raw[q][3]; }
          const float mu = ms * (1.0f / LN_SLOTS);
          float dev = 0.f;
#pragma unroll
          for (int q = 0; q < LN_SLOTS / 2; ++q) {
            const float d0 = ln_raw[q][0] - mu, d1 = ln_raw[q][2] - mu;
            dev += d0 * d0 + d1 * d1;
          }
          const float var = (m2 + dev * 32.0f) * (1.0f / CFD_D);
          // (an LDS write the compiler does not see: in front of one it does, it waits for EVERY pending LDS-DMA request -- it cannot tell that
          //  the (mu, r_sigma) slots lie behind the staging ring -- and the first barrier would wait for both stages instead of the first)
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          const f32x2_t mr = {mu, 1.0f / sqrtf(var + epi.ln_eps)};
          const unsigned la = (unsigned)(unsigned long long)(lptr_t)(smem + NSTAGE * STAGE + threadIdx.x * 8);
          asm volatile("ds_write_b64 %0, %1" ::"v"(la), "v"(mr) : "memory");
        }
      }
    }
  };

  auto compute2 = [&](const char* sbx, const char* sby) __attribute__((always_inline)) {   // sby: the Y tile's base MINUS BI * 128
    spx8 xh[TI], xl[TI], yh[TJ], yl[TJ];
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      xh[ti] = *reinterpret_cast<const spx8*>(sbx + xoff_h + ti * 2048);
      xl[ti] = *reinterpret_cast<const spx8*>(sbx + xoff_l + ti * 2048);
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      yh[tj] = *reinterpret_cast<const spx8*>(sby + yoff_h + tj * 2048);
      yl[tj] = *reinterpret_cast<const spx8*>(sby + yoff_l + tj * 2048);
    }
#if CFD_READS_FIRST
    __builtin_amdgcn_sched_barrier(0);   // all fragment reads of the k-step are issued before its first MFMA
#endif
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        acc[ti][tj] = SP_MFMA(xl[ti], yh[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yl[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yh[tj], acc[ti][tj], 0, 0, 0);
      }
  };
  auto compute = [&](int buf) __attribute__((always_inline)) { compute2(smem + buf * STAGE, smem + buf * STAGE); };

  if constexpr (NSTAGE == 2) {
    stage(0, 0);
    ln_finish(std::integral_constant<int, GPW>{});
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);
      compute(buf);
#if CFD_MFMA_FENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
      __syncthreads();
    }
  } else {
    // tiles kt+1 and kt+2 are in flight while tile kt is consumed
    stage(0, 0);
    if (nkt > 1) {
      stage(1, 1);
      ln_finish(std::integral_constant<int, 2 * GPW>{});
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      int nb2 = buf + 2;
      if (nb2 >= 3) nb2 -= 3;
      if (kt + 2 < nkt) stage(kt + 2, nb2);   // buffer (kt+2)%3 was last read in iteration kt-1
      compute(buf);
      if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPW) : "memory");   // tile kt+1 has landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of `buf` are done before others overwrite it
      __builtin_amdgcn_s_barrier();
      buf = (buf == 2) ? 0 : buf + 1;
    }
  }

  if constexpr (LNF && !(LNF_ABL & 2)) {   // acc <- r_sigma (acc - mu c) + d: lane (l15, q4) holds row (wj TJ + tj) 16 + l15, features (wi TI + ti) 16 + 4 q4 .. + 3
    const float2* sl = reinterpret_cast<const float2*>(smem + NSTAGE * STAGE);
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const float2 mr = sl[(wj * TJ + tj) * 16 + l15];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        acc[ti][tj][0] = mr.y * (acc[ti][tj][0] - mr.x * ln_c4[ti].x) + ln_d4[ti].x;
        acc[ti][tj][1] = mr.y * (acc[ti][tj][1] - mr.x * ln_c4[ti].y) + ln_d4[ti].y;
        acc[ti][tj][2] = mr.y * (acc[ti][tj][2] - mr.x * ln_c4[ti].z) + ln_d4[ti].z;
        acc[ti][tj][3] = mr.y * (acc[ti][tj][3] - mr.x * ln_c4[ti].w) + ln_d4[ti].w;
      }
    }
  }
  if constexpr (WIDE) {
    static_assert(NIT == TI, "one epilogue instruction per MFMA tile");
    // every wave re-lays its tile band by band through a private LDS strip (the staging ring is free now)
    constexpr int RS = TI * 64 + 16;                 // row stride in bytes (+16: conflict-free 16-byte writes)
    char* strip = smem + wid * (16 * RS);
    __syncthreads();
    constexpr bool BAND = EpiHasBand<Epi>::value && !EpiHasStore8<Epi>::value;
    float4 band_r[BAND ? NIT : 1];
    float4 band_t = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (BAND) {
      band_t = epi.tile_bias(min(i0 + wi * TI * 16 + (lane % LPR) * 4, Ig - 4));   // (a lane's columns do not depend on the band)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        int i, j;
        epi_ij(0, it, i, j);
        band_r[it] = epi.band_load(g, b, z, min(i, Ig - 4), min(j, a.J - 1));
      }
    }
    if constexpr (CFD_BIAS8_LATE && EpiHasStore8<Epi>::value && EpiHasBias8<Epi>::value) {
      const int i8 = i0 + wi * TI * 16 + (lane % (TI * 2)) * 8;   // (a lane's columns do not depend on the band)
      if (i8 + 8 <= Ig) epi.tile_bias8(i8, s8_t0, s8_t1);
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
        *reinterpret_cast<f32x4*>(strip + l15 * RS + (ti * 16 + q4 * 4) * 4) = acc[ti][tj];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private strip: no barrier needed
      if constexpr (EpiHasStoreT<Epi>::value) {
        if (MODE == MODE_GROUPED && g == 1) {   // (workgroup-uniform) the band read by columns: lane = (feature, token half), see EpiQkvT
          static_assert(TI % 2 == 0, "32 features x 2 token halves per wave instruction");
#pragma unroll
          for (int it = 0; it < TI / 2; ++it) {
            const int il = (lane & 31) + 32 * it, hf = lane >> 5;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(strip + (8 * hf + e) * RS + il * 4);
            const int i = i0 + wi * TI * 16 + il, jb = j0 + (wj * TJ + tj) * 16;
            if (i < Ig && jb < a.J) epi.store_t(i, jb, hf, v);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next band overwrites the strip
          continue;
        }
      }
      if constexpr (EpiHasStore8<Epi>::value) {
        // split-pair outputs: LPR8 lanes per row, 8 consecutive i per lane
        constexpr int LPR8 = TI * 2, RPI8 = 64 / LPR8, NIT8 = 16 / RPI8;
#pragma unroll
        for (int it = 0; it < NIT8; ++it) {
          const int i = i0 + wi * TI * 16 + (lane % LPR8) * 8;
          const int j = j0 + (wj * TJ + tj) * 16 + it * RPI8 + lane / LPR8;
          const char* sp = strip + (it * RPI8 + lane / LPR8) * RS + (lane % LPR8) * 32;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 16);
          if (j < a.J) {
            if constexpr (EpiHasBias8<Epi>::value) {
              if (i + 4 < Ig) epi.store8(g, b, z, i, j, v0, v1, s8_t0, s8_t1);
              else if (i < Ig) epi(g, b, z, i, j, v0);
            } else {
              if (i + 4 < Ig) epi.store8(g, b, z, i, j, v0, v1);
              else if (i < Ig) epi(g, b, z, i, j, v0);
            }
          }
        }
      } else if constexpr (EpiHasBand<Epi>::value) {
        // the old values of band tj were requested one band ago (band 0: before the loop); band tj+1's requests go out
        // BEFORE this band's stores, which they must not be reordered with as far as hipcc knows
        f32x4 bv[NIT];
        float4 brn[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          bv[it] = *reinterpret_cast<const f32x4*>(strip + (it * RPI + lane / LPR) * RS + (lane % LPR) * 16);
        if (tj + 1 < TJ) {
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            int i, j;
            epi_ij(tj + 1, it, i, j);
            brn[it] = epi.band_load(g, b, z, min(i, Ig - 4), min(j, a.J - 1));
          }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          int i, j;
          epi_ij(tj, it, i, j);
          if constexpr (EpiHasRowStat<Epi>::value) {   // (every lane takes part in the slot reduction; out-of-range lanes store nothing)
            static_assert(LPR == 8 || LPR == 16, "a 32-column slot = 8 lanes of one row");
            epi.band_store_stat(g, b, z, min(i, Ig - 4), min(j, a.J - 1), bv[it], band_r[it], band_t, i < Ig && j < a.J);
          } else
          if (i < Ig && j < a.J) epi.band_store(g, b, z, i, j, bv[it], band_r[it], band_t);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) band_r[it] = brn[it];
      } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        int i, j;
        epi_ij(tj, it, i, j);
        const f32x4 v = *reinterpret_cast<const f32x4*>(strip + (it * RPI + lane / LPR) * RS + (lane % LPR) * 16);
        if (i < Ig && j < a.J) {
          if constexpr (Epi::kPrefetch) epi(g, b, z, i, j, v, pre[it][tj]);
          else epi(g, b, z, i, j, v);
        }
      }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next band overwrites the strip
    }
  } else {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        int i, j;
        epi_ij(tj, ti, i, j);
        if (i < Ig && j < a.J) {
          if constexpr (Epi::kPrefetch) epi(g, b, z, i, j, acc[ti][tj], pre[ti][tj]);
          else epi(g, b, z, i, j, acc[ti][tj]);
        }
      }
  }
}

template <int WI, int WJ, int TI, int TJ, int NSTAGE, int MODE, class Epi>
__global__ void __launch_bounds__(WI * WJ * 64, (WI * WJ * 64) / 256 * ((WI * TI + WJ * TJ) * 16 * 128 * NSTAGE <= 52 * 1024 ? 3 : 2))
gemm_sp_kernel(const GemmArgs a, const Epi epi) {
  gemm_sp_body<WI, WJ, TI, TJ, NSTAGE, MODE, Epi>(a, epi, blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z),
                                                  (int)gridDim.x, (int)gridDim.y, (int)gridDim.z);
}

// Reference kernel with the same operands / epilogues, one thread per (4 i, 1 j): used by the
// CFD_NAIVE_GEMM=1 debug switch to separate MFMA-path bugs from host-side plumbing bugs.
__device__ __forceinline__ float sp_load(const char* row, int col) {
  const char* p = row + (size_t)(col >> 5) * 128 + (col & 31) * 2;
  return (float)*reinterpret_cast<const sp_t*>(p) + (float)*reinterpret_cast<const sp_t*>(p + 64);
}

template <int MODE, class Epi>
__global__ void gemm_sp_naive_kernel(const GemmArgs a, const Epi epi, int g_fixed) {
  const int b = a.brow ? a.brow[blockIdx.y] : blockIdx.y, z = blockIdx.z;
  const int g = g_fixed;
  const int Ig = a.I[g];
  const int nq = Ig / 4;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)nq * a.J) return;
  const int i = (int)(idx % nq) * 4, j = (int)(idx / nq);
  const char* yrow = a.Y + (long long)b * a.ybs + (long long)z * a.yzs + (long long)min(j, a.Jclamp - 1) * a.ldy;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int kbase = 0;
  const int s0 = (MODE == MODE_SEGK) ? 0 : g, s1 = (MODE == MODE_SEGK) ? a.nslot : g + 1;
  for (int s = s0; s < s1; ++s) {
    const long long bi = a.xmap[s] ? a.xmap[s][b] : b;
    const char* xs = a.X[s] + bi * a.xbs[s] + (long long)z * a.xzs;
    if (MODE == MODE_SEGK) kbase = a.yk0[s] * 32;
    for (int k = 0; k < a.kt[s] * 32; ++k) {
      const float yv = sp_load(yrow, kbase + k);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[e] += sp_load(xs + (long long)min(i + e, a.Iclamp[s] - 1) * a.ldx[s], k) * yv;
    }
    kbase += a.kt[s] * 32;
  }
  if constexpr (Epi::kPrefetch) epi(g, b, z, i, j, f32x4{acc[0], acc[1], acc[2], acc[3]}, epi.prefetch(g, b, z, i, j));
  else epi(g, b, z, i, j, f32x4{acc[0], acc[1], acc[2], acc[3]});
}

// ------------------------------------------------------------------------------------------------
// Host-side launcher
// ------------------------------------------------------------------------------------------------
struct GemmCfgSel { int wi, wj, ti, tj; };

extern int g_cfd_naive_gemm;  // set from CFD_NAIVE_GEMM env at cfd_create

template <int WI, int WJ, int TI, int TJ, int NSTAGE, int MODE, class Epi>
static hipError_t launch_cfg(GemmArgs a, const Epi& epi, int nb, int nz, hipStream_t st) {
  constexpr int BI = WI * TI * 16, BJ = WJ * TJ * 16;
  a.tiles_j = (a.J + BJ - 1) / BJ;
  int total = 0;
  const int ng = (MODE == MODE_GROUPED) ? a.nslot : 1;
  for (int g = 0; g < ng; ++g) {
    a.tiles_i[g] = (a.I[g] + BI - 1) / BI;
    a.tile_start[g] = total;
    total += a.tiles_i[g] * a.tiles_j;
  }
  a.tile_start[ng] = total;
  constexpr int lds = NSTAGE * (BI + BJ) * 128 + (EpiHasLnFold<Epi>::value ? BJ * 8 : 0);   // (+ the rows' (mu, r_sigma) of the LayerNorm fold)
  // the attribute is per device: one bit per device ordinal (a process may hold handles on several GPUs)
  static unsigned long long attr_set = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_set >> (dev & 63)) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sp_kernel<WI, WJ, TI, TJ, NSTAGE, MODE, Epi>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set |= 1ull << (dev & 63);
  }
  hipLaunchKernelGGL((gemm_sp_kernel<WI, WJ, TI, TJ, NSTAGE, MODE, Epi>), dim3(total, nb, nz), dim3(WI * WJ * 64), lds, st, a, epi);
  return hipGetLastError();
}

// Tile configurations (cfg): 0 = chosen from the shape (below); the five shapes the denoising step launches:
//   1 = 128 x 128 (2 x 2 waves of 64 x 64, 2-stage)            the large token-side and memory-side products
//   3 = 128 x 16  (4 waves of 32 x 16, 2-stage)                J <= 16 columns (time tables)
//   6 = 128 x 112 (4 waves of 32 x 112, 2-stage)               J in (128, 224]: per-row attention products of 196 tokens
//  19 = 64 x 64   (2 x 2 waves of 32 x 32, 3-stage counted-vmcnt loop)   small problems: few workgroups per CU, latency-bound
//  20 = 32 x 128  (1 x 4 waves of 32 x 32, 3-stage)            memories of <= 64 keys
//  24 = 128 x 64  (2 x 2 waves of 64 x 32, 3-stage)            small problems with more than 768 tiles of 64 x 64
// The variants measured and rejected in round 1 (3-stage / software-pipelined / deep-prefetch / single-buffer loops,
// 128 x 256, 256 x 128, 128 x 176 and 256 x 176 tiles, the tile-softmax epilogues) live in tools/experiments/gemm_sp_r01_variants.hpp.
template <int MODE, class Epi>
static hipError_t launch_gemm(GemmArgs a, const Epi& epi, int nb, int nz, hipStream_t st, int cfg = 0) {
  if (a.nslot < 1) a.nslot = 1;
  if (g_cfd_naive_gemm) {
    const int ng = (MODE == MODE_GROUPED) ? a.nslot : 1;
    for (int g = 0; g < ng; ++g) {
      const long long n = (long long)(a.I[g] / 4) * a.J;
      if (n == 0) continue;
      hipLaunchKernelGGL((gemm_sp_naive_kernel<MODE, Epi>), dim3((unsigned)((n + 255) / 256), nb, nz), dim3(256), 0, st,
                         a, epi, g);
    }
    return hipGetLastError();
  }
  if (cfg == 0) {
    const int ng = (MODE == MODE_GROUPED) ? a.nslot : 1;
    long long big_tiles = 0;
    for (int g = 0; g < ng; ++g) big_tiles += (long long)((a.I[g] + 127) / 128) * ((a.J + 255) / 256);
    big_tiles *= (long long)nb * nz;
    int imax = 0;
    for (int g = 0; g < ng; ++g) imax = a.I[g] > imax ? a.I[g] : imax;
    if (a.J <= 16) cfg = 3;
    else if (imax <= 64 && a.J >= 96) cfg = 20;
    else if (a.J > 128 && a.J <= 224 && big_tiles * 2 >= 256) cfg = 6;
    else if (big_tiles * 2 >= 384 && a.J >= 96) cfg = 1;
    else cfg = 19;
    // more 64 x 64 tiles than the chip holds at once (3 per CU) but too few for the 128 x 128 class: 128 x 64 tiles, one round of two per CU
    // (the q | k and FFN1 products of 32 utterances at the product shape: 896 tiles -> 448; 21 -> 19 us in the captured step, round 5)
    if (cfg == 19 && nb * nz == 1) {
      long long t64 = 0;
      for (int g = 0; g < ng; ++g) t64 += (long long)((a.I[g] + 63) / 64) * ((a.J + 63) / 64);
      if (t64 > 768 && imax >= 128) cfg = 24;
    }
  }
  switch (cfg) {
    case 1: return launch_cfg<2, 2, 4, 4, 2, MODE, Epi>(a, epi, nb, nz, st);
    case 6: return launch_cfg<4, 1, 2, 7, 2, MODE, Epi>(a, epi, nb, nz, st);
    case 19: return launch_cfg<2, 2, 2, 2, 3, MODE, Epi>(a, epi, nb, nz, st);
    case 20: return launch_cfg<1, 4, 2, 2, 3, MODE, Epi>(a, epi, nb, nz, st);
    case 24: return launch_cfg<2, 2, 4, 2, 3, MODE, Epi>(a, epi, nb, nz, st);
    default: return launch_cfg<4, 1, 2, 1, 2, MODE, Epi>(a, epi, nb, nz, st);
  }
}

// The launches of the LayerNorm fold (EpiResidStat producers, EpiLn<E> consumers): un-batched mid-size problems, for which launch_gemm
// above picks the 64 x 64 class or, with more than 768 such tiles, the 128 x 64 class (cfd_forward.hip checks the range) -- only these
// two are instantiated for the fold's epilogues.
template <int MODE, class Epi>
static hipError_t launch_gemm_midsize(GemmArgs a, const Epi& epi, hipStream_t st) {
  if (a.nslot < 1) a.nslot = 1;
  const int ng = (MODE == MODE_GROUPED) ? a.nslot : 1;
  long long t64 = 0;
  int imax = 0;
  for (int g = 0; g < ng; ++g) {
    t64 += (long long)((a.I[g] + 63) / 64) * ((a.J + 63) / 64);
    imax = a.I[g] > imax ? a.I[g] : imax;
  }
  if (t64 > 768 && imax >= 128) return launch_cfg<2, 2, 4, 2, 3, MODE, Epi>(a, epi, 1, 1, st);
  return launch_cfg<2, 2, 2, 2, 3, MODE, Epi>(a, epi, 1, 1, st);
}
