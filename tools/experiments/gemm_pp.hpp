// Persistent "ping-pong" split-pair GEMM for the large token-side products (gfx950):
//     D[i][j] = bias[i] + sum_k X[i][k] * Y[j][k]        (X = weights [I][K], Y = activations [J][K], both SP)
//
// One workgroup per CU, 8 waves = two groups of four.  Both groups work on the same 128 (i) x 256 (j) output tile,
// group g owning the half j in [128 g, 128 g + 128); every SIMD hosts one wave of each group.  The groups run the
// k-loop half a step out of phase, separated by workgroup barriers:
//
//        barrier   b1        b2        b3        b4
//   A :  read(0) | mfma(0) | read(1) | mfma(1) | ...        read(s) = issue the LDS-DMA of k-tile s+2, read the
//   B :          | read(0) | mfma(0) | read(1) | mfma(1)              fragments of k-tile s from LDS into registers
//
// so each SIMD's matrix pipe is fed by exactly one wave at a time while its partner's LDS reads and global->LDS
// requests are in flight.  (gemm_sp_kernel's two co-resident workgroups drift into lock-step instead -- both waves
// of a SIMD want the matrix pipe, then both wait for LDS: 1.19 PFLOP/s of issued MFMA per extra k vs 1.38 here.)
//
// The kernel is persistent: a workgroup walks a static list of tiles and its k-steps form ONE stream, so the
// LDS-DMA of the next tile's first k-tiles is already in flight while the current tile finishes, and the epilogue
// of tile n (kept in a second accumulator set) is spread over the read phases of tile n+1: one 16 x 16 slice per
// k-step, its residual row requested two steps earlier.  Nothing but the very first fill and the very last
// epilogue of a workgroup is exposed.
//
// Three LDS stages of 48 KB.  Hazards (b_n = n-th barrier after the prologue barrier b0; A reads k-tile s between
// b_2s and b_2s+1, B between b_2s+1 and b_2s+2):
//   RAW  every wave waits until its share of k-tile s+1 has landed before the EVEN barrier b_2s+2; the first read
//        of k-tile s+1 (by A) comes after that barrier.
//   WAR  k-tile s+2 goes to stage (s+2)%3 = (s-1)%3, last read by B before b_2s (B waits lgkmcnt(0) before each of
//        its barriers); A issues its share after b_2s, B after b_2s+1.
// vmcnt counts LDS-DMA, the (inline-asm) residual loads and the stores together, in issue order; the kernel keeps
// a scalar count of issued operations and waits with s_waitcnt vmcnt(issued - sequence number of the needed one).
//
// Tile order: workgroups with equal blockIdx % 8 share an XCD (and its L2); XCD x gets a contiguous range of the
// tile space (i-tiles fastest) and its workgroups take consecutive tiles of that range round by round, so the
// i-tiles that re-read one activation panel run on one L2 at about the same time.
#pragma once
#include "../../convofusion_amd/csrc/gemm_sp.hpp"

#define PP_BI 128
#define PP_BJ 256

// s_waitcnt vmcnt(m) with the largest supported m <= n  (waiting for fewer outstanding operations is always safe)
__device__ __forceinline__ void pp_wait_vm(int n) {
  if (n >= 16) {
    if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (n >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else if (n >= 8) {
    if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else if (n >= 4) {
    if (n >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// PPEpi: what happens to a finished 4 (i) x 1 (j) group of results.  kResid: out[j][i] += v (the old value arrives
// through the kernel's residual pipeline); otherwise store(i, j, v) writes it.  Epilogues must not LOAD anything
// (an ordinary load inside the k-loop makes hipcc drain the LDS-DMA queue): the bias enters through the
// accumulator initial value.
struct PPResid {   // x[j][i] += v          (row length CFD_D floats)
  static constexpr bool kResid = true;
  static constexpr bool kSpread = false;
  float* x;
  __device__ __forceinline__ float* addr(int i, int j) const { return x + (long long)j * CFD_D + i; }
  __device__ __forceinline__ unsigned offset(int i, int j) const { return ((unsigned)j * CFD_D + (unsigned)i) * 4u; }   // bytes, < 4 GB
};
template <bool GELU>
struct PPSplit {   // out_sp[j][i] = split(act(v))
  static constexpr bool kResid = false;
  static constexpr bool kSpread = !GELU;   // (erff in eight switch cases overflows the register file: GELU tiles are
                                           //  written at the tile boundary instead)
  char* out;
  long long ldo;   // bytes
  __device__ __forceinline__ void store(int i, int j, f32x4 v) const {
    if constexpr (GELU) { v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); v[2] = gelu_f(v[2]); v[3] = gelu_f(v[3]); }
    sp_store4(out + (long long)j * ldo, i, v[0], v[1], v[2], v[3]);
  }
};
struct PPF32 {     // out[j][goff + i] = v
  static constexpr bool kResid = false;
  static constexpr bool kSpread = true;
  float* out;
  long long ldo;   // floats
  int goff;
  __device__ __forceinline__ void store(int i, int j, f32x4 v) const {
    *reinterpret_cast<float4*>(out + (long long)j * ldo + goff + i) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
struct PPNull {    // timing experiments: keeps the results live, stores nothing
  static constexpr bool kResid = false;
  static constexpr bool kSpread = true;
  float* sink;
  __device__ __forceinline__ void store(int i, int j, f32x4 v) const {
    asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    if (i < -1) sink[0] = v[0];
  }
};

struct PPArgs {
  const char* X;      // [I][K] SP
  const char* Y;      // [J][K] SP
  long long ldx, ldy; // bytes
  int I, J, kt;       // kt = K / 32
  const float* bias;  // [I] or null
  int tiles_i, tiles_j;
};

template <class Epi>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(const PPArgs a, const Epi epi) {
  constexpr int TI = 4, TJ = 4;
  constexpr int BI = PP_BI, BJ = PP_BJ;
  constexpr int NW = 8;
  constexpr int STAGE = (BI + BJ) * 128;
  constexpr int XPW = BI / 8 / NW, YPW = BJ / 8 / NW;
  constexpr int GPW = XPW + YPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 2, wi = (wid >> 1) & 1, wj = wid & 1;
  const int nkt = a.kt;

  // ---- static tile list of this workgroup
  const int ncu = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
  const int T = a.tiles_i * a.tiles_j;
  const int tq = T >> 3, tr = T & 7;
  const int v0 = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + m;
  const int cnt = tq + (xcd < tr ? 1 : 0);
  const int my_tiles = m < cnt ? (cnt - m + ncu - 1) / ncu : 0;
  if (my_tiles == 0) return;
  const int S = my_tiles * nkt;

  // ---- staging (LDS-DMA) state: runs two k-steps ahead of the compute state
  const int cpos = lane & 7, rsub = lane >> 3;
  unsigned xoff[XPW], yoff[YPW];   // byte offsets from a.X / a.Y (operands are < 4 GB)
  int st_k = 0, st_kt = 0, st_buf = 0, st_s = 0;
  auto stage_next = [&]() __attribute__((always_inline)) {
    if (st_kt == 0) {
      const int v = v0 + ncu * st_k;
      const int i0s = (v % a.tiles_i) * BI, j0s = (v / a.tiles_i) * BJ;
#pragma unroll
      for (int n = 0; n < XPW; ++n) {
        const int r = (wid + NW * n) * 8 + rsub;
        xoff[n] = (unsigned)min(i0s + r, a.I - 1) * (unsigned)a.ldx + ((cpos ^ ((r >> 1) & 7)) << 4);
      }
#pragma unroll
      for (int n = 0; n < YPW; ++n) {
        const int r = (wid + NW * n) * 8 + rsub;
        yoff[n] = (unsigned)min(j0s + r, a.J - 1) * (unsigned)a.ldy + ((cpos ^ ((r >> 1) & 7)) << 4);
      }
    }
    char* sbuf = smem + st_buf * STAGE;
#pragma unroll
    for (int n = 0; n < XPW; ++n)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.X + (xoff[n] + (unsigned)st_kt * 128u)), (lptr_t)(sbuf + (wid + NW * n) * 1024), 16, 0, 0);
#pragma unroll
    for (int n = 0; n < YPW; ++n)
      __builtin_amdgcn_global_load_lds((gptr_t)(a.Y + (yoff[n] + (unsigned)st_kt * 128u)), (lptr_t)(sbuf + BI * 128 + (wid + NW * n) * 1024), 16, 0, 0);
    ++st_s;
    if (++st_kt == nkt) { st_kt = 0; ++st_k; }
    st_buf = (st_buf == 2) ? 0 : st_buf + 1;
  };

  // ---- fragment addressing (same swizzle as gemm_sp_kernel)
  const int l15 = lane & 15, q4 = lane >> 4;
  const int sw = l15 >> 1;
  const int xoff_h = (wi * TI * 16 + l15) * 128 + ((q4 ^ sw) << 4);
  const int xoff_l = (wi * TI * 16 + l15) * 128 + (((4 + q4) ^ sw) << 4);
  const int yrow0 = grp * 128 + wj * TJ * 16;
  const int yoff_h = BI * 128 + (yrow0 + l15) * 128 + ((q4 ^ sw) << 4);
  const int yoff_l = BI * 128 + (yrow0 + l15) * 128 + (((4 + q4) ^ sw) << 4);

  f32x4 acc[TI][TJ], done[TI][TJ];   // running tile / finished tile whose epilogue is in progress
  spx8 xh[TI], xl[TI], yh[TJ], yl[TJ];
  auto load_frags = [&](int buf) __attribute__((always_inline)) {
    const char* sb = smem + buf * STAGE;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      xh[ti] = *reinterpret_cast<const spx8*>(sb + xoff_h + ti * 2048);
      xl[ti] = *reinterpret_cast<const spx8*>(sb + xoff_l + ti * 2048);
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      yh[tj] = *reinterpret_cast<const spx8*>(sb + yoff_h + tj * 2048);
      yl[tj] = *reinterpret_cast<const spx8*>(sb + yoff_l + tj * 2048);
    }
  };
  auto mfma_all = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        acc[ti][tj] = SP_MFMA(xl[ti], yh[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yl[tj], acc[ti][tj], 0, 0, 0);
        acc[ti][tj] = SP_MFMA(xh[ti], yh[tj], acc[ti][tj], 0, 0, 0);
      }
  };
  auto barrier = [&]() __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- vmcnt bookkeeping (wave-uniform scalars)
  int vm = 0;            // vector-memory operations issued so far by this wave
  int seqG_prev = 0;     // vm right after the stage issued in the previous read phase
  int seqG_cur = 0;      // vm right after the stage issued in this read phase

  // ---- tile bookkeeping
  int buf = 0;
  int ci0, cj0;          // origin of the running tile
  int pi0 = 0, pj0 = 0;  // origin of the finished tile
  auto tile_origin = [&](int k, int& i0, int& j0) __attribute__((always_inline)) {
    const int v = v0 + ncu * k;
    i0 = (v % a.tiles_i) * BI;
    j0 = (v / a.tiles_i) * BJ;
  };
  // The bias enters through the accumulator's initial value.  The workgroup's 128 bias values live in LDS behind
  // the staging ring (an ordinary load inside the k-loop would make hipcc drain the LDS-DMA queue, so the global
  // read happens in the prologue and is repeated only when the i-tile changes; with workgroups-per-XCD a multiple
  // of tiles_i -- every shape this library launches -- that never happens).
  float* bias_lds = reinterpret_cast<float*>(smem + 3 * STAGE);
  auto load_bias = [&](int i0) __attribute__((always_inline)) {
    if (threadIdx.x < BI) bias_lds[threadIdx.x] = a.bias ? a.bias[min(i0 + (int)threadIdx.x, a.I - 1)] : 0.f;
  };
  auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_lds + wi * 64 + ti * 16 + q4 * 4);
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = bv;
    }
  };
  // slice c of a wave tile: ti = c & 3, tj = c >> 2; element (i, j) of this lane
  auto slice_ij = [&](int c, int i0, int j0, int& i, int& j) __attribute__((always_inline)) {
    i = i0 + wi * 64 + (c & 3) * 16 + q4 * 4;
    j = j0 + yrow0 + (c >> 2) * 16 + l15;
  };
  // ---- epilogue pipeline.  `done` holds the finished tile.  Store-only epilogues: in read phases 0..7 of the
  // following tile two 16 x 16 slices are written per phase, hidden behind the partner's MFMA phase.  The residual
  // epilogue needs the old values: a load with a VGPR destination inside the k-loop either makes hipcc drain the
  // LDS-DMA queue (ordinary load) or, as inline asm, leaves the in-flight destination registers exposed to the
  // register allocator's copies (tried: requests landing in the freed `done` registers -- hipcc routes them through
  // temporaries and scratch), and an LDS landing slot per wave only fits one request ahead (tried: every read
  // phase then waits ~0.8 us for it).  So a residual tile is written at the tile boundary: all 16 old-value loads
  // first, then the stores; the LDS-DMA of the next tile's first two k-tiles is in flight meanwhile.
  auto write_slice = [&](auto ctag, int i0, int j0) __attribute__((always_inline)) {
    constexpr int c = decltype(ctag)::value;   // compile-time slice index: keeps every register index static
    int i, j;
    slice_ij(c, i0, j0, i, j);
    if constexpr (!Epi::kResid) {
      if (i < a.I && j < a.J) epi.store(i, j, Epi::kSpread ? done[c & 3][c >> 2] : acc[c & 3][c >> 2]);
    }
  };
  auto write_tile_now = [&](int i0, int j0) __attribute__((always_inline)) {
    if constexpr (Epi::kResid) {
#pragma unroll
      for (int cb = 0; cb < 16; cb += 8) {
        f32x4 r[8];
        unsigned off[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          int i, j;
          slice_ij(cb + q, i0, j0, i, j);
          off[q] = epi.offset(min(i, a.I - 4), min(j, a.J - 1));
        }
        float* const xbase = epi.x;
        // eight requests and their wait in ONE asm statement: the destinations are never exposed while in flight,
        // and hipcc sees no ordinary load in the loop nest (it would drain the LDS-DMA queue on every k-step)
        asm volatile(
            "global_load_dwordx4 %0, %8, %16\n\tglobal_load_dwordx4 %1, %9, %16\n\t"
            "global_load_dwordx4 %2, %10, %16\n\tglobal_load_dwordx4 %3, %11, %16\n\t"
            "global_load_dwordx4 %4, %12, %16\n\tglobal_load_dwordx4 %5, %13, %16\n\t"
            "global_load_dwordx4 %6, %14, %16\n\tglobal_load_dwordx4 %7, %15, %16\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
            : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "s"(xbase)
            : "memory");
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          int i, j;
          slice_ij(cb + q, i0, j0, i, j);
          const f32x4 v = r[q] + acc[(cb + q) & 3][(cb + q) >> 2];   // (residual tiles are written straight from acc)
          if (i < a.I && j < a.J) *reinterpret_cast<float4*>(epi.addr(i, j)) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    } else {
#define PP_ALL(C) write_slice(std::integral_constant<int, C>{}, i0, j0);
      PP_ALL(0) PP_ALL(1) PP_ALL(2) PP_ALL(3) PP_ALL(4) PP_ALL(5) PP_ALL(6) PP_ALL(7)
      PP_ALL(8) PP_ALL(9) PP_ALL(10) PP_ALL(11) PP_ALL(12) PP_ALL(13) PP_ALL(14) PP_ALL(15)
#undef PP_ALL
    }
  };
  const bool spread = Epi::kSpread && nkt >= 8;   // the following k-loop hides the epilogue

  // ---- prologue
  tile_origin(0, ci0, cj0);
  load_bias(ci0);
  stage_next(); vm += GPW; seqG_prev = vm;
  if (S > 1) { stage_next(); vm += GPW; }
  seqG_cur = vm;
  pp_wait_vm(vm - seqG_prev);          // k-tile 0 has landed (k-tile 1 may be in flight)
  barrier();                           // b0 (also: the bias values are in LDS)
  init_acc();
  if (grp == 1) barrier();             // b1: group B starts half a step later
  seqG_prev = seqG_cur;                // the newest stage so far is k-tile 1: it must land before b2

  // (nested loops, not one flat k-step stream with an "end of tile" test: hipcc then keeps `done` in place instead
  //  of shuffling 64 registers on every k-step)
  for (int tk = 0; tk < my_tiles; ++tk) {
    const bool have_done = tk > 0 && spread;
    for (int kt = 0; kt < nkt; ++kt) {
      // ================= read phase (the partner wave of this SIMD is in its MFMA phase)
      if (st_s < S) { stage_next(); vm += GPW; seqG_cur = vm; }
      load_frags(buf);
      if (have_done && kt < 8) {
#define PP_CASE(P) case P: write_slice(std::integral_constant<int, 2 * P>{}, pi0, pj0); write_slice(std::integral_constant<int, 2 * P + 1>{}, pi0, pj0); break;
        switch (kt) {
          PP_CASE(0) PP_CASE(1) PP_CASE(2) PP_CASE(3) PP_CASE(4) PP_CASE(5) PP_CASE(6) PP_CASE(7)
          default: break;
        }
#undef PP_CASE
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 1) pp_wait_vm(vm - seqG_prev);
      barrier();
      // ================= MFMA phase
      __builtin_amdgcn_s_setprio(1);
      mfma_all();
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (grp == 0) pp_wait_vm(vm - seqG_prev);
      barrier();
      seqG_prev = seqG_cur;
      buf = (buf == 2) ? 0 : buf + 1;
    }
    // tile finished: hand the accumulators to the epilogue and start the next tile
    pi0 = ci0; pj0 = cj0;
    const bool more = tk + 1 < my_tiles;
    if constexpr (!Epi::kSpread) {
      write_tile_now(pi0, pj0);   // straight from acc
    } else {
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) done[ti][tj] = acc[ti][tj];
      if (!spread || !more) write_tile_now(pi0, pj0);
    }
    if (more) {
      tile_origin(tk + 1, ci0, cj0);
      // (an i-tile change would need a workgroup-wide refresh of bias_lds between barriers; launch_gemm_pp only
      //  accepts grids where a workgroup's i-tile is fixed)
      init_acc();
    }
  }
  if (grp == 0) barrier();
}

template <class Epi>
static hipError_t launch_gemm_pp(PPArgs a, const Epi& epi, hipStream_t st, int nblocks = 256) {
  a.tiles_i = (a.I + PP_BI - 1) / PP_BI;
  a.tiles_j = (a.J + PP_BJ - 1) / PP_BJ;
  constexpr int lds = 3 * (PP_BI + PP_BJ) * 128 + PP_BI * 4;   // staging ring, bias
  if (nblocks % 8 || (nblocks / 8) % a.tiles_i) return hipErrorInvalidValue;   // a workgroup keeps one i-tile (bias in LDS)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_pp_kernel<Epi>), dim3(nblocks), dim3(512), lds, st, a, epi);
  return hipGetLastError();
}
