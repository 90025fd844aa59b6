// Fused cross-attention block, ROLE-SPLIT variant (round 3).  Same arithmetic, work list, tile formats and LDS images as
// xattn_fused_kernel (xattn_fused.hpp -- read its header first); what changes is who does what:
//
//   xattn_fused_kernel   the two waves of a pair (w, w + 4: one SIMD) split the 512-long axes and run the SAME program in lock-step:
//                        both want the matrix pipe in the score / P.V sections and both leave it idle in the softmax, the
//                        exchange of partial scores and the fill issue (33 % MFMA-busy).
//   xattn_role_kernel    the pair splits the WORK: wave w ("A") holds the 16 queries at full depth (128 VGPRs) and does
//                        scores + softmax of key tile n; its SIMD partner w + 4 ("B") holds the whole 512-feature output
//                        accumulator (128 VGPRs) and does P.V of key tile n - 1, whose probabilities A handed over through 2.5 KB of
//                        LDS.  The two waves of a SIMD now run DIFFERENT instruction mixes one tile apart: A's softmax (vector
//                        ALU) and B's P.V (matrix pipe) overlap, there are no partial scores to exchange, and a step needs two
//                        barriers (one per tile half) instead of three.
//
// Per loop iteration n (a "step" = 32 keys), two intervals:
//   interval 1   A: S^T rows of keys 0-15 of tile n   (Ka: 16 k-steps, 48 MFMAs)      B: O^T += Va(n-1) P(n-1)  (16 feature tiles, 48 MFMAs)
//   barrier 1    everybody is done with Ka(n) and Va(n-1)  ->  requests Ka(n+1) (+ key bias / scale), Va(n)
//   interval 2   A: keys 16-31 (Kb, 48 MFMAs), softmax of the tile, P'(n) -> LDS             B: O^T += Vb(n-1) P(n-1)  (48 MFMAs)
//   barrier 2    everybody is done with Kb(n) and Vb(n-1); P'(n) visible  ->  requests Kb(n+1), Vb(n)
// Every fill has one interval to land (the wait in front of each barrier is vmcnt(0)).  A segment list with a FLUSH (two long
// memories) is cut into runs; the pipeline drains at the end of a run, B flushes O into x, and the next run primes again.
#pragma once
#include "xattn_fused.hpp"

#define XR_RS (CFD_D * 4 + 16)                       // flush strip row stride (512 features + pad)
#define XR_HOFF(t) (XA_XOFF + (t) * 4096)            // hand-off record of pair t: [0, 2048) P' hi/lo per lane, [2048, 2560) scale / inv per lane
static_assert(4 * 16 * XR_RS <= XA_XOFF, "B-wave flush strips must not reach the hand-off / key-bias / segment areas");

__global__ void __launch_bounds__(XA_WAVES * 64, 2) xattn_role_kernel(const XAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KOFF = 0, VOFF = 65536;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = wid & 3, role = wid >> 2;            // role 0 = A (scores + softmax), 1 = B (P.V); (w, w + 4) share a SIMD
  const int l15 = lane & 15, q4 = lane >> 4, sw = l15 >> 1;
  const int cpos = lane & 7, rsub = lane >> 3;

  const XaWg* wgp = a.wgs + blockIdx.x;
  const int my_row = wgp->row[tile];
  const int my_q0 = wgp->q0[tile];
  const int seg0 = wgp->seg0, nseg = wgp->nseg;
  const bool active = my_row >= 0;
  const long long tok0 = active ? (long long)my_row * a.L + my_q0 : 0;
  const int nq = active ? min(16, a.L - my_q0) : 0;

  if (threadIdx.x < nseg) reinterpret_cast<int4*>(smem + XA_SEGOFF)[threadIdx.x] = reinterpret_cast<const int4*>(a.segs + seg0)[threadIdx.x];
  float* cq_pair = reinterpret_cast<float*>(smem + XA_CQOFF + tile * XA_CQW);   // the pair's c_q [16][5] (A) ...
  float* wq_pair = cq_pair + 80;                                                  // ... and sum_s P' [16][5] (A writes, B's flush reads)
  const int trow = *a.d_step;
  constexpr int KBOFF = VOFF + 16 * 1024;     // A b of the five memories, parked in the V^T tile buffer until the first Va / Vb fills
  if (wid < CFD_NMEM) {
    const char* kp = reinterpret_cast<const char*>(xa_sel(a.kb, wid) + (long long)trow * xa_sel(a.kb_stride, wid)) + lane * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)kp, (lptr_t)(smem + KBOFF + wid * 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(kp + 1024), (lptr_t)(smem + KBOFF + wid * 2048 + 1024), 16, 0, 0);
  }

  // ---- staging (identical to xattn_fused_kernel: same pieces per wave, same LDS images) ---------------------------------------
  // XR_SPLIT_FILL (default): only the B waves request tiles -- B wave w also does the pieces wave w - 4 has in xattn_fused_kernel
  // (virtual waves fw and fw + 4 below) -- and the A waves issue the L2 touches: a touch that misses sits in the A wave's own
  // vmcnt queue, where nothing waits behind it; in a filling wave it would hold up every younger fill (vmcnt retires in order).
#ifndef XR_SPLIT_FILL
#define XR_SPLIT_FILL 1
#endif
#ifndef XR_ABLATE
#define XR_ABLATE 0   // developer timing builds (results are garbage): 1 = no fragment reads / MFMAs / softmax, 2 = no fills, 4 = no touches
#endif
  const int fw = XR_SPLIT_FILL ? (wid & 3) : wid;             // first virtual fill wave of this wave
  constexpr int NFV = XR_SPLIT_FILL ? 2 : 1;                  // virtual fill waves per filling wave (fw, fw + 4)
  const bool filler = !XR_SPLIT_FILL || role == 1;
  const int kr = (fw & 1) * 8 + rsub;
  const int kkey = 8 * (kr >> 2) + (kr & 3);
  const int ksrc_lane = kkey * (CFD_D * 4) + (fw >> 1) * 128 + ((cpos ^ ((kr >> 1) & 7)) << 4);     // virtual wave fw + 4: + 256
  const int kdst_wave = KOFF + (fw >> 1) * 4096 + (fw & 1) * 1024;                                   //                   + 8192
  const int vsw = (cpos ^ (((fw & 1) << 2) | (rsub >> 1))) << 4;
  struct Tile { const char* k; const char* v; const float* cb; long long rowb; unsigned vlane; unsigned cblane; };
  auto seg_field = [&](int si, int fld) __attribute__((always_inline)) -> int {
    return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(smem + XA_SEGOFF)[si * 4 + fld]);
  };
  auto seg_tile = [&](int si, Tile& t, int& T, int& wm, int& fl, int& j) __attribute__((always_inline)) {
    j = seg_field(si, 0);
    const int u = seg_field(si, 1);
    wm = seg_field(si, 2); fl = seg_field(si, 3);
    const int Sp = xa_sel(a.Sp, j);
    T = Sp / XA_KEYS;
    t.k = xa_sel(a.K, j) + (long long)u * Sp * (CFD_D * 4);
    t.v = xa_sel(a.VT, j) + (long long)u * CFD_D * Sp * 4;
    t.cb = xa_sel(a.cb, j) + (long long)u * Sp;
    t.rowb = (long long)Sp * 4;
    t.vlane = (unsigned)((fw * 8 + rsub) * Sp * 4 + vsw);                                                // virtual wave fw + 4: + 32 rows
    t.cblane = (unsigned)((lane & 31) * 4) + (lane >= 32 ? xa_sel(a.rs_off, j) : 0u);
  };
  auto fill_k = [&](const Tile& t, int hb, int slot) __attribute__((always_inline)) {
    if (!filler || (XR_ABLATE & 2)) return;
#pragma unroll
    for (int v = 0; v < NFV; ++v)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      unsigned kl = (unsigned)ksrc_lane;
      const char* b = t.k + hb * (4 * CFD_D * 4) + n * 512 + v * 256;
      asm volatile("" : "+v"(kl), "+s"(b));
      __builtin_amdgcn_global_load_lds((gptr_t)(b + kl), (lptr_t)(smem + kdst_wave + v * 8192 + hb * 2048 + n * 16384), 16, 0, 0);
    }
    if (hb == 0) {
      unsigned cl = t.cblane;
      asm volatile("" : "+v"(cl));
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(t.cb) + cl), (lptr_t)(smem + XA_CBOFF + slot * 256), 4, 0, 0);
    }
  };
  auto fill_v = [&](const Tile& t, int hb) __attribute__((always_inline)) {
    if (!filler || (XR_ABLATE & 2)) return;
#pragma unroll
    for (int v = 0; v < NFV; ++v)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      unsigned vl = t.vlane;
      const int g = 8 * (n & 1) + 32 * (n >> 1) + 16 * hb + 4 * v;   // 8-row group of virtual wave fw + 4 v (+ fw per wave)
      const char* b = t.v + (long long)g * 8 * t.rowb;
      asm volatile("" : "+v"(vl), "+s"(b));
      __builtin_amdgcn_global_load_lds((gptr_t)(b + vl), (lptr_t)(smem + VOFF + (fw + g) * 1024), 16, 0, 0);
    }
  };
  const int off_h = (q4 ^ sw) << 4, off_l = ((4 + q4) ^ sw) << 4;

  // ---- L2 prefetch.  The K / V^T tiles of the long memory stream from the Infinity Cache / HBM, and every workgroup on an XCD that
  //      reads the same instance waits for the same miss; a fill has one interval to land.  So each of the pf_n workgroups that walk
  //      the instance together touches 1 / pf_n of the lines of the tile XR_PF_DIST steps ahead -- one dword per 128-byte line, K
  //      lines in interval 1 and V^T lines in interval 2 -- and the fills find them in L2.  The touch is the YOUNGEST vector-memory
  //      operation in front of the barrier wait, which then is vmcnt(1): everything older (the fills) has landed, the touch itself
  //      may stay in flight for another interval (vmcnt retires in order).
#ifndef XR_PF_DIST
#define XR_PF_DIST 2
#endif
  // the wait in front of a barrier: fills landed (a filling wave: everything but a touch it issued last), LDS traffic of this wave done
#if XR_SPLIT_FILL
#define XR_WAIT(pf_out_) do { if (filler) XA_WAIT_VM_LGKM0(0); else XA_WAIT_VM_LGKM0(63); } while (0)
#else
#define XR_WAIT(pf_out_) do { if (pf_out_) XA_WAIT_VM_LGKM0(1); else XA_WAIT_VM_LGKM0(0); } while (0)
#endif
  const int pf_n = wgp->pf_n, pf_slot = wgp->pf_slot;
  const int pf_cnt = pf_n > 0 ? (512 + pf_n - 1) / pf_n : 0;               // lines per workgroup, interval and tile
  const bool toucher = !XR_SPLIT_FILL || role == 0;
  const bool pf_wave = XR_PF_DIST > 0 && toucher && wid * 64 < pf_cnt;       // does this wave issue a touch at all? (wave-uniform)
  // (every lane of a touching wave loads SOMETHING -- surplus lanes repeat the share's last line -- so that the instruction is issued
  //  whatever the share: the vmcnt(1) of the caller counts on it)
  const int pf_line = min(pf_slot * pf_cnt + min(wid * 64 + lane, max(pf_cnt - 1, 0)), 511);
  // (an LDS-DMA dword into a scrap area -- the unused c_q slots of the B waves -- not a load into a register: a register written
  //  when the data returns, up to two intervals later, cannot be described to the register allocator)
  auto touch = [&](const char* p) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + XA_CQOFF + (4 + (wid & 3)) * XA_CQW), 4, 0, 0);
  };
  // returns true when a touch was issued (the caller then waits with vmcnt(1))
  auto prefetch = [&](const Tile& t, int kt, int T, int flags, int which) __attribute__((always_inline)) -> bool {
    if (!pf_wave || (XR_ABLATE & 4) || !(flags & XA_ONLINE) || kt + XR_PF_DIST >= T) return false;
    if (which == 0) touch(t.k + (long long)XR_PF_DIST * (XA_KEYS * CFD_D * 4) + (long long)pf_line * 128);
    else touch(t.v + XR_PF_DIST * 128 + (long long)pf_line * t.rowb);
    return true;
  };

  // ---- the step cursor: (segment, key tile) pairs in list order; every wave walks it the same way -------------------------------
  struct Cur { Tile t; int si, kt, T, mask, flags, j; };
  auto cur_first = [&](Cur& c, int si) __attribute__((always_inline)) {
    c.si = si; c.kt = 0;
    seg_tile(si, c.t, c.T, c.mask, c.flags, c.j);
  };
  auto cur_next = [&](Cur& c) __attribute__((always_inline)) {   // caller has checked that a next step exists in this run
    if (c.kt + 1 < c.T) {
      c.kt += 1;
      c.t.k += XA_KEYS * CFD_D * 4; c.t.v += 128; c.t.cb += XA_KEYS;
    } else {
      cur_first(c, c.si + 1);
    }
  };

  // runs: [s_begin, s_end) segments; a run ends behind a segment that carries XA_FLUSH
  int s_begin = 0;
  if (role == 0) {
    // =============================================== A: scores + softmax ===============================================
    // Q fragments at FULL depth: q = LayerNorm2(x[token]); lane (q = l15, g = q4) holds d = 32 c + 8 g .. + 7 for c = 0..15
    spx8 qh[16], ql[16];
    {
      // two passes over the row (the second one hits L1 / L2): holding the 128 floats AND the 128 registers of fragments at the
      // same time does not fit the register file
      const float* xr = a.x + (tok0 + min(l15, max(nq - 1, 0))) * CFD_D + q4 * 8;
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4 r0 = *reinterpret_cast<const float4*>(xr + 32 * c), r1 = *reinterpret_cast<const float4*>(xr + 32 * c + 4);
        sum += ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w));
      }
      const float mean = xlane_sum(sum) * (1.0f / CFD_D);
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        float4 r0 = *reinterpret_cast<const float4*>(xr + 32 * c), r1 = *reinterpret_cast<const float4*>(xr + 32 * c + 4);
        r0.x -= mean; r0.y -= mean; r0.z -= mean; r0.w -= mean; r1.x -= mean; r1.y -= mean; r1.z -= mean; r1.w -= mean;
        ss += ((r0.x * r0.x + r0.y * r0.y) + (r0.z * r0.z + r0.w * r0.w)) + ((r1.x * r1.x + r1.y * r1.y) + (r1.z * r1.z + r1.w * r1.w));
      }
      const float rstd = 1.0f / sqrtf(xlane_sum(ss) * (1.0f / CFD_D) + 1e-5f);
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // [P0] A b (parked in the V^T buffer) and the segment list visible
      // third pass: fragments, and c_q = q . (A b) of every memory from the split values (q = hi + lo), chunk by chunk
      float cacc[CFD_NMEM] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        float4 v0 = *reinterpret_cast<const float4*>(xr + 32 * c), v1 = *reinterpret_cast<const float4*>(xr + 32 * c + 4);
        v0.x -= mean; v0.y -= mean; v0.z -= mean; v0.w -= mean; v1.x -= mean; v1.y -= mean; v1.z -= mean; v1.w -= mean;
        const float* gp = a.ln_g + 32 * c + q4 * 8;
        const float* bp = a.ln_b + 32 * c + q4 * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
        const float y[8] = {v0.x * rstd * g0.x + b0.x, v0.y * rstd * g0.y + b0.y, v0.z * rstd * g0.z + b0.z, v0.w * rstd * g0.w + b0.w,
                            v1.x * rstd * g1.x + b1.x, v1.y * rstd * g1.y + b1.y, v1.z * rstd * g1.z + b1.z, v1.w * rstd * g1.w + b1.w};
        float qf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sp_t hi, lo;
          split_f32(y[e], hi, lo);
          qh[c][e] = hi;
          ql[c][e] = lo;
          qf[e] = (float)hi + (float)lo;
        }
#pragma unroll
        for (int j = 0; j < CFD_NMEM; ++j) {
          const float* kp = reinterpret_cast<const float*>(smem + KBOFF + j * 2048) + q4 * 8 + 32 * c;
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(kp), k1 = *reinterpret_cast<const f32x4*>(kp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) cacc[j] += qf[e] * k0[e] + qf[4 + e] * k1[e];
        }
        // one chunk at a time: otherwise hipcc requests all 16 chunks first and keeps the 256 halves in a register each until it packs them
        asm volatile("" : "+v"(qh[c]), "+v"(ql[c]), "+v"(cacc[0]), "+v"(cacc[1]), "+v"(cacc[2]), "+v"(cacc[3]), "+v"(cacc[4]));
      }
#pragma unroll
      for (int j = 0; j < CFD_NMEM; ++j) {
        const float acc = xlane_sum(cacc[j]);
        if (q4 == 0) { cq_pair[l15 * 5 + j] = acc; wq_pair[l15 * 5 + j] = 0.f; }
      }
    }
    XA_WAIT_VM_LGKM0(0);
    __builtin_amdgcn_s_barrier();   // [P1] everybody is done with the parked A b: the V^T buffer may be filled

    const char* kfrag = smem + KOFF + l15 * 128;
    // S^T tile `t` (0: Ka, 1: Kb) over all 16 k-steps: eight batches of two k-steps, the next batch's 4 fragment reads issued in
    // front of the current batch's 6 MFMAs (two register sets of 16 VGPRs: the wave has 128 VGPRs of Q fragments to carry)
    spx8 fa[4], fb[4];
    auto read_k = [&](spx8 (&fr)[4], int t, int b) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* kp = kfrag + (2 * b + i) * 4096 + t * 2048;
        fr[2 * i] = *reinterpret_cast<const spx8*>(kp + off_h);
        fr[2 * i + 1] = *reinterpret_cast<const spx8*>(kp + off_l);
      }
    };
    auto score_half = [&](f32x4& acc, int t) __attribute__((always_inline)) {
#define XR_MFMA_K(FR, B)                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                       \
    acc = SP_MFMA(FR[2 * i + 1], qh[2 * (B) + i], acc, 0, 0, 0);        \
    acc = SP_MFMA(FR[2 * i], ql[2 * (B) + i], acc, 0, 0, 0);            \
    acc = SP_MFMA(FR[2 * i], qh[2 * (B) + i], acc, 0, 0, 0);            \
  }
      read_k(fa, t, 0);
      read_k(fb, t, 1);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fa, 0)
      __builtin_amdgcn_sched_barrier(0);
      read_k(fa, t, 2);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fb, 1)
      __builtin_amdgcn_sched_barrier(0);
      read_k(fb, t, 3);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fa, 2)
      __builtin_amdgcn_sched_barrier(0);
      read_k(fa, t, 4);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fb, 3)
      __builtin_amdgcn_sched_barrier(0);
      read_k(fb, t, 5);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fa, 4)
      __builtin_amdgcn_sched_barrier(0);
      read_k(fa, t, 6);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fb, 5)
      __builtin_amdgcn_sched_barrier(0);
      read_k(fb, t, 7);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fa, 6)
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_K(fb, 7)
      __builtin_amdgcn_sched_barrier(0);
#undef XR_MFMA_K
    };
    char* hand = smem + XR_HOFF(tile);
#ifndef XR_A_PRIO
#define XR_A_PRIO 1   // the A wave of a SIMD gets the matrix pipe first: it still has the softmax to do when its MFMAs are through,
                      // while its partner has nothing but MFMAs (MI355X_MICROARCH.md, two waves per SIMD, item 2: priority, then age)
#endif
#if XR_A_PRIO
    __builtin_amdgcn_s_setprio(XR_A_PRIO);
#endif

    while (s_begin < nseg) {
      int s_end = s_begin;
      while (s_end < nseg && !(seg_field(s_end, 3) & XA_FLUSH)) ++s_end;
      if (s_end < nseg) ++s_end;
      Cur cur, nxt;
      cur_first(cur, s_begin);
      nxt = cur;
      bool has_cur = true;
      // prime: Ka(first) + key bias, Kb(first)
      fill_k(cur.t, 0, 0);
      fill_k(cur.t, 1, 0);
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // [R0]
      float m = -INFINITY, lsum = 0.f, wl = 0.f;
      int step = 0;
      bool pf_out = false;   // a prefetch touch is the youngest outstanding vector-memory operation
      for (;;) {   // iterations n = 0 .. steps of the run (the last one only drains B)
        bool has_next = false;
        if (has_cur) {
          has_next = (cur.kt + 1 < cur.T) || (cur.si + 1 < s_end);
          if (has_next) { nxt = cur; cur_next(nxt); }
        }
        const int slot = step & 1;
        const bool in_seg = !(XR_ABLATE & 1) && has_cur && active && ((cur.mask >> tile) & 1);
        const bool online = (cur.flags & XA_ONLINE) != 0;
        const float cq = in_seg ? cq_pair[l15 * 5 + cur.j] : 0.f;
        f32x4 s0 = f32x4{cq, cq, cq, cq}, s1 = s0;
        // ---- interval 1
        if (in_seg) score_half(s0, 0);
        XR_WAIT(pf_out);
        __builtin_amdgcn_s_barrier();   // [B1]
        if (has_next) fill_k(nxt.t, 0, slot ^ 1);
        if (has_cur) fill_v(cur.t, 0);
        pf_out = has_cur && prefetch(cur.t, cur.kt, cur.T, cur.flags, 1);
        // ---- interval 2
        if (in_seg) {
          score_half(s1, 1);
          // softmax of the tile: lane (q, g) holds keys 8 g + e, e = 0..7 (s0 = e 0..3, s1 = e 4..7)
          const f32x4 kb0 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + q4 * 32);
          const f32x4 kb1 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + q4 * 32 + 16);
          const f32x4 rs0 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + 128 + q4 * 32);
          const f32x4 rs1 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + 128 + q4 * 32 + 16);
          const float rs[8] = {rs0[0], rs0[1], rs0[2], rs0[3], rs1[0], rs1[1], rs1[2], rs1[3]};
          float p[8];
          p[0] = fmaf(s0[0], rs[0], kb0[0]); p[1] = fmaf(s0[1], rs[1], kb0[1]); p[2] = fmaf(s0[2], rs[2], kb0[2]); p[3] = fmaf(s0[3], rs[3], kb0[3]);
          p[4] = fmaf(s1[0], rs[4], kb1[0]); p[5] = fmaf(s1[1], rs[5], kb1[1]); p[6] = fmaf(s1[2], rs[6], kb1[2]); p[7] = fmaf(s1[3], rs[7], kb1[3]);
          const float mx = xlane_max(fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7]))));
          constexpr float LOG2E = 1.44269504088896340736f;
          float scale = 1.0f;
          if (online) {
            const float m_new = fmaxf(m, mx);
            const bool dead = m_new == -INFINITY;
            const float mc = dead ? 0.f : m_new * LOG2E;
            scale = dead ? 1.0f : __builtin_amdgcn_exp2f(fmaf(m, LOG2E, -mc));
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { p[e] = __builtin_amdgcn_exp2f(fmaf(p[e], LOG2E, -mc)); ps += p[e]; }
            lsum = lsum * scale + xlane_sum(ps);
            m = m_new;
            float pw = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { p[e] *= rs[e]; pw += p[e]; }
            wl = wl * scale + pw;
          } else {
            const float mc = mx * LOG2E;
            float ps = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { p[e] = __builtin_amdgcn_exp2f(fmaf(p[e], LOG2E, -mc)); ps += p[e]; }
            const float inv = 1.0f / xlane_sum(ps);
#pragma unroll
            for (int e = 0; e < 8; ++e) { p[e] = (p[e] * inv) * rs[e]; wl += p[e]; }
          }
          spx8 ph, pl;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const sp_t hi = (sp_t)p[e];
            ph[e] = hi;
            pl[e] = (sp_t)(p[e] - (float)hi);
          }
          const bool last_in_seg = cur.kt + 1 == cur.T;
          float inv_out = 1.0f;
          if (last_in_seg) {
            float wsum = xlane_sum(wl);
            if (online) { inv_out = 1.0f / lsum; wsum *= inv_out; }   // (all keys dead: 0 * inf = NaN, like the reference)
            if (q4 == 0) wq_pair[l15 * 5 + cur.j] = wsum;
            m = -INFINITY; lsum = 0.f; wl = 0.f;
          }
          *reinterpret_cast<spx8*>(hand + lane * 32) = ph;
          *reinterpret_cast<spx8*>(hand + lane * 32 + 16) = pl;
          *reinterpret_cast<float2*>(hand + 2048 + lane * 8) = make_float2(scale, inv_out);
        }
        XR_WAIT(pf_out);
        __builtin_amdgcn_s_barrier();   // [B2]
        if (has_next) fill_k(nxt.t, 1, slot ^ 1);
        if (has_cur) fill_v(cur.t, 1);
        pf_out = has_next && prefetch(nxt.t, nxt.kt, nxt.T, nxt.flags, 0);
        if (!has_cur) break;            // this was the drain iteration
        has_cur = has_next;
        cur = nxt;
        ++step;
      }
      // end of run: B flushes O (two barriers around it)
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // [F0]
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // [F1]
      s_begin = s_end;
    }
  } else {
    // =============================================== B: P.V + residual ==================================================
    f32x4 o[32];   // O^T tiles of features 16 f .. 16 f + 15
#pragma unroll
    for (int f = 0; f < 32; ++f) o[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    XA_WAIT_VM_LGKM0(0);
    __builtin_amdgcn_s_barrier();   // [P0]
    XA_WAIT_VM_LGKM0(0);
    __builtin_amdgcn_s_barrier();   // [P1]
    const char* vfrag = smem + VOFF + l15 * 128;
    const char* hand = smem + XR_HOFF(tile);
    spx8 ph, pl;
    spx8 fa[8], fb[8];
    // feature tiles of a V^T half `hb` (0: Va = tiles 0-7, 16-23; 1: Vb = tiles 8-15, 24-31), in four batches of four
    auto ftile = [](int hb, int b, int i) __attribute__((always_inline)) -> int { return 8 * hb + 16 * (b >> 1) + 4 * (b & 1) + i; };
    auto read_v = [&](spx8 (&fr)[8], int hb, int b) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* vp = vfrag + ftile(hb, b, i) * 2048;
        fr[2 * i] = *reinterpret_cast<const spx8*>(vp + off_h);
        fr[2 * i + 1] = *reinterpret_cast<const spx8*>(vp + off_l);
      }
    };
    auto pv_half = [&](int hb) __attribute__((always_inline)) {
      // (the feature-tile index must be a compile-time constant for o[]: the batches are unrolled by hand)
#define XR_MFMA_V(FR, B)                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                   \
    const int f = (hb == 0 ? 0 : 8) + 16 * ((B) >> 1) + 4 * ((B) & 1) + i;          \
    o[f] = SP_MFMA(FR[2 * i + 1], ph, o[f], 0, 0, 0);                                \
    o[f] = SP_MFMA(FR[2 * i], pl, o[f], 0, 0, 0);                                    \
    o[f] = SP_MFMA(FR[2 * i], ph, o[f], 0, 0, 0);                                    \
  }
      read_v(fa, hb, 0);
      read_v(fb, hb, 1);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_V(fa, 0)
      __builtin_amdgcn_sched_barrier(0);
      read_v(fa, hb, 2);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_V(fb, 1)
      __builtin_amdgcn_sched_barrier(0);
      read_v(fb, hb, 3);
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_V(fa, 2)
      __builtin_amdgcn_sched_barrier(0);
      XR_MFMA_V(fb, 3)
      __builtin_amdgcn_sched_barrier(0);
#undef XR_MFMA_V
    };
    // x[token][:] += O^T (+ folded bias + rank-one terms on the last run), then O = 0: the wave re-lays its 512 x 16 tile through a
    // private LDS strip (every fill has landed and nobody reads the tile buffers: barrier [F0]) and moves whole 1 KB row pieces
    auto flush = [&](bool add_bias) __attribute__((always_inline)) {
      const XAttnArgs* ka = reinterpret_cast<const XAttnArgs*>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
      asm volatile("" : "+s"(ka));
      char* strip = smem + tile * (16 * XR_RS);
#pragma unroll
      for (int f = 0; f < 32; ++f) {
        *reinterpret_cast<f32x4*>(strip + l15 * XR_RS + (f * 16 + q4 * 4) * 4) = o[f];
        o[f] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 vbv[CFD_NMEM];
        if (add_bias) {
          bv = *reinterpret_cast<const float4*>(ka->bias + h * 256 + lane * 4);
#pragma unroll
          for (int j = 0; j < CFD_NMEM; ++j)
            vbv[j] = *reinterpret_cast<const float4*>(ka->vb[j] + (long long)trow * ka->vb_stride[j] + h * 256 + lane * 4);
        }
        float* xp = ka->x + tok0 * CFD_D + h * 256 + lane * 4;
#pragma unroll 1
        for (int g = 0; g < 2; ++g) {
          float4 old[8];
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8)
            if (g * 8 + r8 < nq) old[r8] = *reinterpret_cast<const float4*>(xp + (long long)(g * 8 + r8) * CFD_D);
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) {
            const int r = g * 8 + r8;
            const f32x4 v = *reinterpret_cast<const f32x4*>(strip + r * XR_RS + h * 1024 + lane * 16);
            if (r < nq) {
              float4 t = old[r8];
              float4 c = make_float4(v[0], v[1], v[2], v[3]);
              if (add_bias) {
#pragma unroll
                for (int j = 0; j < CFD_NMEM; ++j) {
                  const float wj = wq_pair[r * 5 + j];
                  c.x += wj * vbv[j].x; c.y += wj * vbv[j].y; c.z += wj * vbv[j].z; c.w += wj * vbv[j].w;
                }
              }
              t.x = (t.x + bv.x) + c.x; t.y = (t.y + bv.y) + c.y; t.z = (t.z + bv.z) + c.z; t.w = (t.w + bv.w) + c.w;
              *reinterpret_cast<float4*>(xp + (long long)r * CFD_D) = t;
            }
          }
        }
      }
      XA_WAIT_VM(0);
    };

    while (s_begin < nseg) {
      int s_end = s_begin;
      while (s_end < nseg && !(seg_field(s_end, 3) & XA_FLUSH)) ++s_end;
      if (s_end < nseg) ++s_end;
      Cur cur, nxt;
      cur_first(cur, s_begin);
      nxt = cur;
      bool has_cur = true;
      fill_k(cur.t, 0, 0);
      fill_k(cur.t, 1, 0);
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // [R0]
      // control of the step B works on (one behind the cursor)
      bool p_valid = false;
      int p_mask = 0, p_flags = 0, p_last = 0;
      int step = 0;
      bool pf_out = false;
      for (;;) {
        bool has_next = false;
        if (has_cur) {
          has_next = (cur.kt + 1 < cur.T) || (cur.si + 1 < s_end);
          if (has_next) { nxt = cur; cur_next(nxt); }
        }
        const int slot = step & 1;
        const bool in_seg = !(XR_ABLATE & 1) && p_valid && active && ((p_mask >> tile) & 1);
        float inv_in = 1.0f;
        // ---- interval 1
        if (in_seg) {
          ph = *reinterpret_cast<const spx8*>(hand + lane * 32);
          pl = *reinterpret_cast<const spx8*>(hand + lane * 32 + 16);
          const float2 si = *reinterpret_cast<const float2*>(hand + 2048 + lane * 8);
          inv_in = si.y;
          if ((p_flags & XA_ONLINE) && !__all(si.x == 1.0f)) {
#pragma unroll
            for (int f = 0; f < 32; ++f) { o[f][0] *= si.x; o[f][1] *= si.x; o[f][2] *= si.x; o[f][3] *= si.x; }
          }
          pv_half(0);
        }
        XR_WAIT(pf_out);
        __builtin_amdgcn_s_barrier();   // [B1]
        if (has_next) fill_k(nxt.t, 0, slot ^ 1);
        if (has_cur) fill_v(cur.t, 0);
        pf_out = has_cur && prefetch(cur.t, cur.kt, cur.T, cur.flags, 1);
        // ---- interval 2
        if (in_seg) {
          pv_half(1);
          if ((p_flags & XA_ONLINE) && p_last) {   // the finished online memory is normalised in registers
#pragma unroll
            for (int f = 0; f < 32; ++f) { o[f][0] *= inv_in; o[f][1] *= inv_in; o[f][2] *= inv_in; o[f][3] *= inv_in; }
          }
        }
        XR_WAIT(pf_out);
        __builtin_amdgcn_s_barrier();   // [B2]
        if (has_next) fill_k(nxt.t, 1, slot ^ 1);
        if (has_cur) fill_v(cur.t, 1);
        pf_out = has_next && prefetch(nxt.t, nxt.kt, nxt.T, nxt.flags, 0);
        if (!has_cur) break;
        p_valid = true; p_mask = cur.mask; p_flags = cur.flags; p_last = cur.kt + 1 == cur.T;
        has_cur = has_next;
        cur = nxt;
        ++step;
      }
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // [F0] every fill has landed, nobody reads the tile buffers: the strips may alias them
      if (active) flush(s_end >= nseg);
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // [F1] strips read back before the next run primes the tile buffers
      s_begin = s_end;
    }
  }
}
