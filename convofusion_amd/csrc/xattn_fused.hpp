// Fused cross-attention block of one decoder layer (cross_attention.py:578-652, folded form of DESIGN.md section 3):
//
//   x[token][:] += sum_j softmax_s( LN2(x[token]) . Kf_j[u_j][s] + cb_j[u_j][s] ) Vf_j[u_j][s][:]  + cross_bias
//
// for the five memories j (one head of width 512 each), u_j = the memory instance the token's batch row maps to.
// Scores, probabilities and the per-memory outputs never leave the chip: the kernel replaces the score products,
// softmax_rows_kernel and the P.V products of the three-launch path (which stays for calls that want att_mats).
//
// Work decomposition.  A workgroup = 8 waves = 4 query tiles of 16 queries (each of ONE batch row) x 2 halves.  The two
// waves of a pair (2 t, 2 t + 1) share a query tile and split the two 512-long axes between them, so that a wave's resident
// state -- its Q fragments (8 k-steps x hi/lo = 64 VGPRs) and its output accumulator O^T[256 features][16 queries]
// (16 MFMA tiles = 64 VGPRs) -- leaves room for two waves per SIMD:
//   phase A   partial S^T[32 keys][16 q] over ITS half of the 512-deep dot product: Kf tile (LDS, A operand) x Q
//             (registers, B operand), 48 MFMAs; the two partial sums meet through 2 KB of LDS per wave
//   phase B   O^T[its 256 features][16 q] += Vf^T tile (LDS, A operand) x P (registers), 48 MFMAs
// Keys arrive in tiles of 32.  A K tile (32 keys x 2 KB) and a V^T tile (512 features x 128 B) are 64 KB each and LDS
// holds exactly one of each, so the tiles are cut into four 32 KB sub-buffers that are consumed by four sub-phases:
//   A0  scores of the tile's first 16-key MFMA tile   (Ka: LDS rows 0-15 of every k-step)
//   A1  scores of the second                          (Kb: rows 16-31)
//   B0  P.V for the first 128 features of each half   (Va)
//   B1  P.V for the other 128 features of each half   (Vb)
// Each sub-phase is 24 MFMAs fed by 16 fragment reads, done in two halves: the 8 reads of a half are issued BEFORE the 12
// MFMAs of the previous half (two register sets), so the LDS latency hides under MFMAs instead of in front of them -- with
// the reads issued only right behind a barrier, two waves per SIMD reach ~45 % of the MFMA rate even with every operand
// resident.  For the reads of the next sub-phase to be issued early, its sub-buffer is declared ready in the MIDDLE of the
// current one: three barriers per step -- mid-A0 (Kb ready; Vb is refilled behind it), end of A1 (the whole V^T tile ready +
// the pair's partial scores exchanged; both K halves are refilled behind it) and mid-B1 (next Ka ready; Va is refilled) --
// so every fill is issued 1.5-2.5 sub-phases ahead of its use.  (A barrier costs 300-500 cycles of wave skew whatever the
// slack of the fill it waits for -- measured with s_memtime stamps -- so there are as few as the four sub-buffers allow.)  The fills are waited for with COUNTED
// s_waitcnt vmcnt(N) + raw s_barrier (a __syncthreads() would drain the queue); for the counts to hold there is no
// ordinary global load inside the loop: the key bias of a tile arrives through LDS with its Ka fill, and the
// workgroup's segment list is copied to LDS once.
//
// MFMA operand orientation (v_mfma_f32_16x16x32, split pairs, 3 MFMAs per product).  Phase A leaves lane (q = lane&15,
// g = lane>>4) with S^T rows 4g..4g+3 of the two 16-key MFMA tiles.  The K tile's LDS row order is chosen at staging time
// (the per-lane SOURCE address of the LDS-DMA) so that those 8 accumulator registers are keys 8g..8g+7: exactly the
// k-slots lane (q, g) supplies as the B operand of phase B against a V^T tile in natural key order.  So P goes from
// the softmax straight into the P.V MFMAs -- no LDS round trip, no permuted V layout.  Both waves of a pair form the
// same sum of the two partial scores (a + b = b + a in floating point), hence bit-identical probabilities.
//
// Softmax.  A memory longer than one tile runs the online (flash) recurrence and is normalised in registers when its
// last tile is done; a memory of <= 32 padded keys is normalised before its single P.V step, so its contribution is
// simply accumulated on top.  With ONE accumulator only one online memory can be pending at a time: the host puts the
// long memories first and asks for a flush (x += O, O = 0) between two of them (the shipped shapes have at most one
// memory that is long at L = 196 -- the audio memory -- so the benchmark shape never flushes before the end).
// Dead keys (padding, key-padding mask) carry cb = -inf (EpiMemK writes it), so the kernel needs no mask loads.
// A row whose keys are all dead gives NaN like the reference's softmax.
//
// Timestep-independent projections (round 2).  The memories' LayerNorm input is m_s + temb(t): with a_s the centred static part
// and b the centred timestep embedding,  Kf n_s = rs_s (KA_s + A b),  cb_s = rs_s (ca_s + c.b),  Vf n_s = rs_s (VA_s + VV b)
// (rows.hpp, mem_center_kernel).  For such a memory the tiles hold KA / VA (computed once per run) and the kernel applies the
// per-step part itself:
//   score(q, s) = rs_s (S_raw(q, s) + c_q) + cbk_s        c_q = q . (A b): per query and memory, made in the prologue
//   O += VA^T P',  P' = p_s rs_s                           and   x += (sum_s P'_s) VV b   in the epilogue (rank one)
// rs_s and cbk_s (one scalar per key each, mem_scale_all_kernel) arrive together in the tile's key-bias piece.  A memory whose
// projections are still made per step (dynamic memories of the dyadic rollout, per-row timesteps) passes rs = 1, A b = VV b = 0:
// same code, the extra terms vanish exactly.
//
// The segment list (which memory instance, which of the four waves take part) is built by the host per problem
// (build_xattn_worklist, cfd_problem.hip): waves of a workgroup share every LDS tile, so a segment whose instance differs
// between the workgroup's batch rows is split into passes.
#pragma once
#include "cfd_common.hpp"
#include <type_traits>

#define XA_TILES 4     // query tiles (of 16 queries) per workgroup
#define XA_WAVES 8     // two waves per query tile
#define XA_KEYS 32
#define XA_F16_MIN_KEYS 128   // padded keys from which a memory counts as LONG for the operand policy (single-fp16 tiles; see OPF below)
// LDS map: K tile | V^T tile (the epilogue strips alias these two and 2 KB more) | partial-score exchange | key bias of
// two steps | the workgroup's segment list
#define XA_XOFF 133120
#define XA_CBOFF (XA_XOFF + XA_WAVES * 2048)
#define XA_SEGOFF (XA_CBOFF + 512)
#define XA_MAXSEG 24
#define XA_CQOFF (XA_SEGOFF + XA_MAXSEG * 16)   // per wave: c_q partial sums [16 queries][5 memories], then sum_s P' [16][5]
#define XA_CQW 640
#define XA_LDS (XA_CQOFF + XA_WAVES * XA_CQW)

struct XaSeg {
  int j;        // memory 0..4
  int u;        // memory instance
  int wmask;    // query tiles (bit t) whose batch row uses this instance
  int flags;    // XA_ONLINE | XA_FLUSH
};
enum { XA_ONLINE = 1, XA_FLUSH = 2, XA_F16 = 4 };   // XA_F16: the segment's memory has single-fp16 tiles in the formats the kernel instance's OPF names

struct XaWg {
  int row[XA_TILES];   // effective-batch row of query tile t, or -1 (idle)
  int q0[XA_TILES];    // first query (token index inside the row) of tile t
  int aux[XA_TILES];   // layer-0 de-duplication: row of XAttnArgs::dd_out this tile's result is stored to / of dd_in it adds (-1: none)
  int one[XA_TILES];   // one-key memory (XAttnArgs::one_j): its instance for this tile's row (-1: none)
  int att[XA_TILES];   // the ATT instance: row of XaAtt's blocks this tile's attention maps go to (-1: the tile keeps none)
  int seg0, nseg;
  int n16;             // how many of the segments -- a prefix of the list: the long memories come first -- carry XA_F16 (single-fp16 tiles)
  int pf_n;            // (unused)
};

// Attention maps kept by the kernel itself (the ATT instance; round 5): a sampling run that wants the reference's per-iteration attention
// dict (convofusion.py:517-523) on the tile kernels.  The tiles the work list marks (XaWg::att: the full-conditioning chunk's rows) store
// what the softmax has in registers anyway -- for an online memory the tile's probabilities RELATIVE to the tile's exponent reference,
// that reference, and at the end of the memory the final reference and 1 / sum; for a single-tile memory the normalised probabilities --
// and att_fixup_kernel turns them into the maps of ring slot *d_step once per step.  One descriptor per layer.
struct XaAtt {
  float* raw;          // [nb][L][sp_tot]: key column off[j] + s of memory j
  float* mc;           // [nb][L][nt]: tile column t0[j] + s / 32
  float* fin;          // [nb][L][CFD_NMEM][2]: final exponent reference, 1 / sum (single-tile memories: 0, 1)
  int nb, sp_tot, nt;
  int off[CFD_NMEM], t0[CFD_NMEM];
};

struct XAttnArgs {
  float* x;                   // fp32 [M][512] residual stream: read (queries = LayerNorm2(x), norm2 of cross_attention.py:578) and updated in place
  const float* ln_g;          // norm2.weight [512]
  const float* ln_b;          // norm2.bias [512]
  const float* bias;          // folded cross-attention bias [512]
  const char* K[CFD_NMEM];    // this layer's folded keys: SP [U_j * Sp_j][512]
  const float* cb[CFD_NMEM];  // this layer's key bias (+ -inf on dead keys): [U_j * Sp_j]
  unsigned rs_off[CFD_NMEM];  // byte offset from cb[j] to the per-key scale rs of memory j (same indexing; a plane of the same buffer)
  const float* kb[CFD_NMEM];  // A_l b  [512] of table row *d_step: kb[j] + *d_step * kb_stride[j]   (zeros, stride 0: projections made per step)
  const float* vb[CFD_NMEM];  // VV_l b [512], likewise
  long long kb_stride[CFD_NMEM], vb_stride[CFD_NMEM];
  const int* d_step;          // or null: kb / vb point at this step's rows
  const char* VT[CFD_NMEM];   // this layer's folded values^T: SP [U_j][512][Sp_j]
  int Sp[CFD_NMEM];
  int L;
  const XaWg* wgs;
  const XaSeg* segs;
  // layer-0 de-duplication (cfd_problem.hip, build_xattn_layer0_lists).  dd_out: the tiles' results (accumulated memories + their rank-one
  // terms, no bias) are STORED to dd_out[aux][query][512] and x is left alone.  dd_in: x += ... + dd_in[aux][query][512].
  float* dd_out;
  const float* dd_in;
  // A memory with ONE key (lsnemb) has no softmax to speak of: its probability is 1, so its contribution to a row is the vector
  //   rs_u (VA_u + VV b_t)      (u: the row's instance, rs_u: the key's scale at this step)
  // and the 32-key tile step that used to produce it (a K and a V^T fill of 64 KB each for one live key, three barriers) is replaced by one
  // 2 KB read in the flush.  one_j < 0: no such memory (or its key can be masked: then it stays a segment).  Such a memory has no segments
  // in the work lists (cfd_problem.hip, build_xattn_worklist).
  int one_j;                  // memory index, or -1
  int one_sp;                 // its padded length (row pitch of one_rs)
  const float* one_va;        // fp32 [U][512]: VA of its key, this layer
  const float* one_rs;        // fp32 [U * one_sp]: the scale plane of this step (key 0 of instance u at u * one_sp)
  long long* stamps;          // XA_STAMP builds only (tools/xa_stamps.py): per wave, cycles per section of the kernel
  const XaAtt* att;           // the ATT instance: this layer's descriptor
};

// float32 copy of the value row of a one-key memory: out[(l * U + u) * 512 + f] = VA_l,u[f] (key 0 of V^T [nl][U][512][Sp], hi + lo)
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) one_key_va_kernel(const char* vt, long long n, int Sp, float* out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const char* p = vt + i * ((long long)Sp * 4);
  out[i] = (float)*reinterpret_cast<const sp_t*>(p) + (float)*reinterpret_cast<const sp_t*>(p + 64);
}

// Single-fp16 key tiles for the fused kernel's OPF instances (once per run, from the split-pair projections; layouts: see the staging
// comment in the kernel).  One thread per 16-byte chunk (8 values: the `hi` halves, which are the values rounded to fp16).
//   which = 0: V^T  in  SP [n_lu][512][Sp]      out [n_lu][T][512][64 B]
//   which = 1: K    in  SP [n_lu * Sp][512]     out [n_lu][T][2][16][16][64 B]          (n_lu = layers x instances, T = Sp / 32)
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) xa_pack16_kernel(const char* in, char* out, long long n_chunks, int Sp, int which) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_chunks) return;
  const int T = Sp / XA_KEYS;
  const int c = (int)(i & 3);                 // chunk of the 64-byte row: values 8 c .. 8 c + 7
  const int r = (int)((i >> 2) & 511);        // row inside the 32 KB tile
  const long long tile = i >> 11;             // (lu, kt)
  const int kt = (int)(tile % T);
  const long long lu = tile / T;
  const char* src;
  int sw;
  if (which == 0) {                            // r = feature
    src = in + (lu * CFD_D + r) * ((long long)Sp * 4) + kt * 128 + c * 16;
    sw = (r >> 2) & 3;
  } else {                                     // r = (t, ks, row i)
    const int t = r >> 8, ks = (r >> 4) & 15, ri = r & 15;
    const int key = kt * XA_KEYS + 8 * (ri >> 2) + 4 * t + (ri & 3);
    src = in + (lu * Sp + key) * (long long)(CFD_D * 4) + ks * 128 + c * 16;
    sw = (ri >> 2) & 3;
  }
  *reinterpret_cast<uint4*>(out + tile * 32768 + r * 64 + ((c ^ sw) << 4)) = *reinterpret_cast<const uint4*>(src);
}

template <class T>
__device__ __forceinline__ T xa_sel(const T (&arr)[CFD_NMEM], int j) {
  T v = arr[0];
#pragma unroll
  for (int q = 1; q < CFD_NMEM; ++q)
    if (j == q) v = arr[q];
  return v;
}

// s_waitcnt immediates (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt = 6:4, lgkmcnt = 11:8).  The waits are the BUILTIN,
// not inline asm: hipcc's own wait-count bookkeeping then knows what has landed and adds no vmcnt(0) of its own in front
// of the first use of a register that an older load filled (an asm wait is invisible to it).
#ifndef XA_STAMP
#define XA_STAMP 0    // developer build: s_memtime stamps around the sections of the kernel (adds ~10 % to its run time)
#endif
#define XA_NSTAMP 16
#if XA_STAMP
#define XA_T(k) do { asm volatile("" ::: "memory"); const long long t_ = __builtin_amdgcn_s_memtime(); acc_[k] += t_ - tprev_; tprev_ = t_; asm volatile("" ::: "memory"); } while (0)
#else
#define XA_T(k) do { } while (0)
#endif
#ifndef XA_ABLATE
#define XA_ABLATE 0   // developer timing experiments, bit mask: 1 = no fills, 2 = no MFMAs, 4 = no fragment reads, 8 = no softmax (results are garbage)
#endif
#if XA_ABLATE & 2
#define XA_MFMA(a_, b_, c_) ([&]() { asm volatile("" ::"v"(a_), "v"(b_)); return c_; }())
#else
#define XA_MFMA(a_, b_, c_) SP_MFMA(a_, b_, c_, 0, 0, 0)
#endif
#if XA_ABLATE & 4
#define XA_FRAG(p_) ([&]() { spx8 z_; for (int e_ = 0; e_ < 8; ++e_) z_[e_] = (sp_t)(float)(lane + e_); asm volatile("" : "+v"(z_)); return z_; }())
#else
#define XA_FRAG(p_) (*reinterpret_cast<const spx8*>(p_))
#endif
#define XA_WAIT_VM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x70 | (0xF << 8) | ((((N) >> 4) & 3) << 14))
#define XA_WAIT_VM_LGKM0(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x70 | ((((N) >> 4) & 3) << 14))

// ATT: the rows of XaAtt also store their attention maps.
// OPF (operand format of the key tiles of LONG memories, round 6; DESIGN.md section 2 "operand policy"): the kernel is paced by the
// L2 -> LDS fills of the K / V^T tiles (64 KB each as split pairs), so a run whose scheduler tolerates it may carry the tiles of its long
// memories (segments flagged XA_F16 by the host: cfd_problem.hip, XA_F16_MIN_KEYS) as ONE fp16 per value:
//   bit 0 (XA_V16): V^T tiles hold VA as single fp16 (32 KB per 32 keys); P' stays a pair, so P.V is 2 MFMAs per product
//                   (VA_hi . P'_lo + VA_hi . P'_hi) -- the LINEAR path of the attention;
//   bit 1 (XA_K16): K tiles hold KA as single fp16; Q stays a pair (2 MFMAs: KA_hi . q_lo + KA_hi . q_hi) -- the EXPONENTIATED path.
//   bit 2 (XA_P16) / bit 3 (XA_Q16): with single-fp16 tiles, P' / the query fragments enter those products as ONE fp16 too (their `hi` half:
//                   the value rounded to fp16): 1 MFMA per product -- plain fp16 attention against the long memories.
// Why long memories only: the rounding of a value (2^-12 relative, independent signs) enters the output weighted by its probability, so
// over N attended keys the absolute error falls like 1 / sqrt(N) -- for the 1500-key audio memory it is ~8x below that of a 24-key text
// memory, whose tiles are 3 of a row's 50 anyway (measured: profiles/r06_xa_operands_*).  Segments without the flag run the split-pair
// loop body; at a change of format between two segments the pipeline drains and is primed again (once per workgroup at the shipped shapes).
// XA_DBUF (bit 4, with both tiles single fp16): a K and a V^T tile are 32 KB each, so the two 64 KB tile buffers hold TWO of each and the
// long memories' steps run a double-buffered pipeline (kt_step_db: every fill a whole step ahead, two barriers per step instead of three).
// Measured on one box, three interleaved rounds (profiles/r06_xa_dbuf_ab.log): with all four operands single fp16 (OPF 15) the kernel is
// 1.0 % faster at the headline shape (3.575 -> 3.539 ms; 81.4 -> 81.9 steps/s) and 2.9 % at the product shape (0.371 -> 0.360 ms), whose
// workgroups run a handful of steps each and wait on fill latency -- so the shipped single-fp16 instance is OPF 15 | XA_DBUF.  (With pairs
// as the other operands, OPF 3, the same pipeline had measured 0.6 % SLOWER, profiles/r06_xa_double_buffer_ab.log: there the step is
// paced by the matrix pipe and the softmax's vector work, which the two waves of a SIMD use one after the other.)
// The single-fp16 tiles come from xa_pack16_kernel (once per run, from the split-pair projections): tile-major and already in the LDS
// image's order, so a fill is a linear copy of 1 KB pieces.  cfd_forward, DDIM runs, runs that keep attention maps and the memories of a
// dynamic run keep pairs (cfd_sample.hip: operand policy of cfd_sample_begin).
enum { XA_V16 = 1, XA_K16 = 2, XA_P16 = 4, XA_Q16 = 8, XA_DBUF = 16 };   // XA_DBUF (with XA_V16 | XA_K16): the double-buffered step, an instance of its own   // (XA_P16 / XA_Q16: with single-fp16 tiles, also the OTHER operand of the product as one fp16: 1 MFMA)
#ifndef XA_ALL_OPF
#define XA_ALL_OPF 0    // 1: developer builds also instantiate the partial combinations (OPF 1, 2, 3, 7, 11: measured in round 6 and dominated by
                        // OPF 15 -- e.g. 78.5 / 79.7 / 79.5 steps/s for 3 / 7 / 11 against 80.9, all at 2.3e-5 on the DDPM-1000 golden -- and OPF 15 on the
                        // three-barrier step, CFD_XA_DB=0: not shipped)
#endif
template <bool ATT, int OPF>
__global__ void __launch_bounds__(XA_WAVES * 64, 2) xattn_fused_kernel(const XAttnArgs a) {
  static_assert(!ATT || OPF == 0, "attention maps: split-pair tiles");
  static_assert(!(OPF & XA_DBUF) || (OPF & 3) == 3, "double-buffered step: both tiles single fp16");
  typedef std::integral_constant<int, OPF> fmt_long;     // format tags of the loop-body instances: flagged segments / all others
  typedef std::integral_constant<int, 0> fmt_pair;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KOFF = 0, VOFF = 65536;
#if XA_STAMP
  long long acc_[XA_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (12 - 15: sections of the prologue)
  long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifndef XA_PAIR_ADJACENT
#define XA_PAIR_ADJACENT 1
#endif
#if XA_PAIR_ADJACENT
  // The pair of a query tile = waves (2 t, 2 t + 1): on DIFFERENT SIMDs.  With all four tiles busy a SIMD still hosts two waves (of two tiles),
  // which is what the main loop's accounting assumes; a short work list that gives a workgroup ONE tile (the product shape: make_xattn_worklist)
  // then has its two computing waves on two SIMDs instead of taking turns on one.
  const int tile = wid >> 1;     // query tile of the pair
  const int half = wid & 1;      // which half of the 512-long axes
  const int partner = wid ^ 1;
#else
  const int tile = wid & 3;      // query tile of the pair (w, w + 4): the two waves share a SIMD
  const int half = wid >> 2;     // which half of the 512-long axes
  const int partner = wid ^ 4;
#endif
  const int l15 = lane & 15, q4 = lane >> 4, sw = l15 >> 1;
  const int cpos = lane & 7, rsub = lane >> 3;

  const XaWg* wgp = a.wgs + blockIdx.x;
  const int my_row = wgp->row[tile];
  const int my_q0 = wgp->q0[tile];
  const int my_aux = wgp->aux[tile];
  const int my_one = wgp->one[tile];
  const int seg0 = wgp->seg0, nseg = wgp->nseg;
  const bool active = my_row >= 0;                     // wave-uniform
  const long long tok0 = active ? (long long)my_row * a.L + my_q0 : 0;
  const int nq = active ? min(16, a.L - my_q0) : 0;    // valid queries of this wave's tile
  bool att_on = false;                                 // (wave-uniform) this wave stores its tile's attention maps: one wave of the pair
  int my_att = -1;
  if constexpr (ATT) { my_att = wgp->att[tile]; att_on = active && half == 0 && my_att >= 0; }

  // the segment list of this workgroup -> LDS (read back with ds_read: no vector-memory traffic inside the loop)
  if (threadIdx.x < nseg) reinterpret_cast<int4*>(smem + XA_SEGOFF)[threadIdx.x] = reinterpret_cast<const int4*>(a.segs + seg0)[threadIdx.x];

  // Q fragments (B operand) of this wave's half of the feature axis, made here from the residual stream:
  //   q = LayerNorm2(x[token])  (cross_attention.py:578; two-pass mean / variance like ln_rows_kernel, eps 1e-5).
  // Lane (q = l15, g = q4) holds d = 32 c + 8 g .. +7 of its query's row for the 32-chunks c = 8 half .. 8 half + 7, its fragments (the four
  // lanes of a query cover the half row: its statistics need one 4-lane reduction, the row's the partner wave's half as well).
  spx8 qh[8], ql[8];
  float* cq_mine = reinterpret_cast<float*>(smem + XA_CQOFF + wid * XA_CQW);
  float* wq_mine = cq_mine + 80;
  const int trow = a.d_step ? *a.d_step : 0;   // (null: kb / vb are this step's rows already -- no dependent scalar load in front of the A b request)
  // A b of the five memories (2 KB each) -> LDS by the LDS-DMA, issued before anything else so that its round trip runs under the
  // row loads and the LayerNorm below.  Parked in the part of the V^T tile buffer that is first filled after the first step's mid-A0
  // barrier (Vb, row groups 16-31; an instance with single-fp16 V^T tiles: row groups 48-63, which belong to Vb in the pair format and
  // are beyond the 32 KB a single-fp16 tile takes); c_q is computed from there behind the first barrier.
  constexpr int KBOFF = VOFF + ((OPF & XA_V16) ? 48 : 16) * 1024;
  if (wid < CFD_NMEM) {
    const char* kp = reinterpret_cast<const char*>(xa_sel(a.kb, wid) + (long long)trow * xa_sel(a.kb_stride, wid)) + lane * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)kp, (lptr_t)(smem + KBOFF + wid * 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(kp + 1024), (lptr_t)(smem + KBOFF + wid * 2048 + 1024), 16, 0, 0);
  } else if (wid < CFD_NMEM + 2) {
    // norm2's weight (wave 5) and bias (wave 6) the same way, behind A b: read as 32 float4 per lane straight from memory they were
    // 32 loads that the registers (the 16 rows are live) only let go out four at a time -- a chain of waits in front of the first fill
    const char* gp = reinterpret_cast<const char*>(wid == CFD_NMEM ? a.ln_g : a.ln_b) + lane * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(smem + KBOFF + wid * 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(gp + 1024), (lptr_t)(smem + KBOFF + wid * 2048 + 1024), 16, 0, 0);
  }
  // Each wave of the pair loads only ITS half of the 16 rows (the chunks it keeps as fragments): the prologue is paced by what a CU can take
  // in (~11 B per clock with every CU in its prologue), and with whole rows in both waves a workgroup read its 128 KB of rows twice.  The
  // row statistics are put together from the two halves' (mean, sum of squared deviations) -- Chan's pairwise update, two-pass inside a
  // half -- exchanged through the pair's exchange area across the barrier below; both waves compute the same two sums of the same two
  // operands, so they normalise with identical statistics.
  float4 r[16];
  float rstd, mean;
  float mean_h, m2_h;
  {
    const float* xr = a.x + (tok0 + min(l15, max(nq - 1, 0))) * CFD_D + 256 * half + q4 * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      r[2 * c] = *reinterpret_cast<const float4*>(xr + 32 * c);
      r[2 * c + 1] = *reinterpret_cast<const float4*>(xr + 32 * c + 4);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += (r[i].x + r[i].y) + (r[i].z + r[i].w);
    mean_h = xlane_sum(sum) * (2.0f / CFD_D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float dx = r[i].x - mean_h, dy = r[i].y - mean_h, dz = r[i].z - mean_h, dw = r[i].w - mean_h;
      ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    m2_h = xlane_sum(ss);
    if (q4 == 0) reinterpret_cast<float2*>(smem + XA_XOFF + wid * 2048)[l15] = float2{mean_h, m2_h};
  }
  XA_T(12);
  XA_WAIT_VM_LGKM0(0);            // row loads consumed; A b, norm2's parameters landed
  __builtin_amdgcn_s_barrier();   // ... and visible, with the segment list and the partner's half-row statistics
  f32x4 o[16];   // O^T tiles of features 256 half + 16 f .. +15
  // The second half of the prologue -- the fragments from the loaded half rows, c_q, the zeroed accumulators -- as a block that runs BEHIND the
  // first tile's fill requests where the instance allows it (finish_queries() below: the fills' round trip then runs under this arithmetic).
  auto finish_queries = [&]() __attribute__((always_inline)) {
  {
    const float2 oth = reinterpret_cast<const float2*>(smem + XA_XOFF + partner * 2048)[l15];
    const float dm = mean_h - oth.x;
    mean = 0.5f * (mean_h + oth.x);
    rstd = 1.0f / sqrtf(((m2_h + oth.y) + dm * dm * (CFD_D / 4)) * (1.0f / CFD_D) + 1e-5f);
  }
  {
    const float* lng = reinterpret_cast<const float*>(smem + KBOFF + CFD_NMEM * 2048);
    const float* lnb = reinterpret_cast<const float*>(smem + KBOFF + (CFD_NMEM + 1) * 2048);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int c = 8 * half + ks;          // (half is wave-uniform: the two candidates are selected, not indexed)
      const float4 v0 = float4{r[2 * ks].x - mean, r[2 * ks].y - mean, r[2 * ks].z - mean, r[2 * ks].w - mean};
      const float4 v1 = float4{r[2 * ks + 1].x - mean, r[2 * ks + 1].y - mean, r[2 * ks + 1].z - mean, r[2 * ks + 1].w - mean};
      const float* gp = lng + 32 * c + q4 * 8;
      const float* bp = lnb + 32 * c + q4 * 8;
      const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
      const float y[8] = {v0.x * rstd * g0.x + b0.x, v0.y * rstd * g0.y + b0.y, v0.z * rstd * g0.z + b0.z, v0.w * rstd * g0.w + b0.w,
                          v1.x * rstd * g1.x + b1.x, v1.y * rstd * g1.y + b1.y, v1.z * rstd * g1.z + b1.z, v1.w * rstd * g1.w + b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sp_t hi, lo;
        split_f32(y[e], hi, lo);
        qh[ks][e] = hi;
        ql[ks][e] = lo;
      }
    }
  }
  XA_T(14);
  // this wave's half of c_q = q . (A b) for every memory, from the fragments (q = hi + lo); kept in LDS, read back by this wave only
  {
    float qf[8][8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = (float)qh[ks][e] + (float)ql[ks][e];
#pragma unroll
    for (int j = 0; j < CFD_NMEM; ++j) {
      const float* kp = reinterpret_cast<const float*>(smem + KBOFF + j * 2048) + 256 * half + q4 * 8;
      float acc = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(kp + 32 * ks), k1 = *reinterpret_cast<const f32x4*>(kp + 32 * ks + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(qf[ks][4 + e], k1[e], __builtin_fmaf(qf[ks][e], k0[e], acc));   // (the library is built with -ffp-contract=off)
      }
      acc = xlane_sum(acc);
      if (q4 == 0) { cq_mine[l15 * 5 + j] = acc; wq_mine[l15 * 5 + j] = 0.f; }
    }
    // the one-key memory: sum_s P'_s = rs_u for every query of the tile (its rank-one term is added with the others in the final flush)
    if (a.one_j >= 0 && my_one >= 0 && q4 == 0) wq_mine[l15 * 5 + a.one_j] = a.one_rs[(long long)my_one * a.one_sp];
  }
  XA_T(15);
#pragma unroll
  for (int f = 0; f < 16; ++f) o[f] = f32x4{0.f, 0.f, 0.f, 0.f};
  XA_WAIT_VM_LGKM0(0);            // the counted waits of the loop start from an empty queue.  (No barrier: the c_q halves are wave-private, and
                                  // the parked A b is overwritten only behind the first step's mid-A0 barrier, which every wave reaches after this point.)
  };

  // ---- staging: one piece = one global_load_lds_dwordx4 wave-instruction = 8 tile rows x 128 B; a 32 KB sub-buffer is
  //      32 pieces = 4 per wave.  K tile LDS image: [k-step 16][row 32][128 B]; LDS row rho = 16 t + i holds key
  //      8 (i>>2) + 4 t + (i&3) of the tile (see header); chunk swizzle (rho>>1)&7 on the source address.
  //      Sub-buffer Ka = rows 0-15, Kb = rows 16-31.  Piece n of wave `wid`: rows (wid&1)*8 .. +7 (+16 for Kb), k-step (wid>>1) + 4 n.
  const int kr = (wid & 1) * 8 + rsub;                                   // row inside the 16-row half
  const int kkey = 8 * (kr >> 2) + (kr & 3);                             // its key (+ 4 for the second half)
  const int ksrc_lane = kkey * (CFD_D * 4) + (wid >> 1) * 128 + ((cpos ^ ((kr >> 1) & 7)) << 4);
  const int kdst_wave = KOFF + (wid >> 1) * 4096 + (wid & 1) * 1024;
  //      V^T tile LDS image: [feature 512][128 B]; Va = features [0,128) + [256,384), Vb = the rest.  Piece n of wave `wid`:
  //      8-row group g = wid + 8 (n&1) + 32 (n>>1) (+16 for Vb); swizzle ((f>>1)&7) = ((wid&1)<<2) | (rsub>>1)
  const int vsw = (cpos ^ (((wid & 1) << 2) | (rsub >> 1))) << 4;
  //      Single-fp16 tiles (OPF): a tile is 32 KB, contiguous in memory and already in the LDS image's order (xa_pack16_kernel):
  //        K  [half t 2][k-step 16][row i 16][64 B], row 16 t + i = key 8 (i>>2) + 4 t + (i&3) as above, 16-byte chunk c at (c ^ (i>>2)&3);
  //           Ka / Kb = the two halves, 16 pieces each = 2 per wave (k-steps wid, wid + 8)
  //        V^T [feature 512][64 B], chunk c at (c ^ (f>>2)&3); Va / Vb as above in 16-row groups: 2 pieces per wave (groups wid, wid + 16, + 8 for Vb)
  //      so a piece's source is tile base + piece * 1 KB + lane * 16.
  const unsigned lane16 = (unsigned)lane * 16u;
  // Per-lane addresses that depend on the FORMAT of the segment at hand live in one set of variables, set per segment (set_format below):
  // an instance with two loop bodies (OPF != 0) otherwise keeps both bodies' loop invariants in registers at once and spills.
  unsigned kfill_lane = (unsigned)ksrc_lane;      // source offset of this lane inside a K piece
  const char *kf_a, *kf_b, *vf_a, *vf_b;          // fragment read bases: pairs: hi / lo chunk of the lane's row; single fp16: the chunk (b unused)

  // A tile = 32 keys of one memory instance: its K rows, its V^T column block, its key bias; rowb = bytes per V^T feature
  // row of that memory; vlane = this lane's byte offset inside a V^T piece (depends on rowb).  All but vlane are wave-uniform.
  struct Tile { const char* k; const char* v; const float* cb; long long rowb; unsigned vlane; unsigned cblane; };
  auto seg_field = [&](int si, int fld) __attribute__((always_inline)) -> int {
    return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(smem + XA_SEGOFF)[si * 4 + fld]);
  };
  auto seg_tile = [&](int si, Tile& t, int& T, int& wm, int& fl, int& j) __attribute__((always_inline)) {   // first tile of segment si
    j = seg_field(si, 0);
    const int u = seg_field(si, 1);
    wm = seg_field(si, 2); fl = seg_field(si, 3);
    const int Sp = xa_sel(a.Sp, j);
    T = Sp / XA_KEYS;
    const bool f16 = OPF != 0 && (fl & XA_F16);     // (wave-uniform) this memory's tiles are single fp16 in the formats OPF names
    t.k = xa_sel(a.K, j) + ((f16 && (OPF & XA_K16)) ? (long long)u * T * 32768 : (long long)u * Sp * (CFD_D * 4));
    t.v = xa_sel(a.VT, j) + ((f16 && (OPF & XA_V16)) ? (long long)u * T * 32768 : (long long)u * CFD_D * Sp * 4);
    t.cb = xa_sel(a.cb, j) + (long long)u * Sp;
    t.rowb = (long long)Sp * 4;
    t.vlane = (f16 && (OPF & XA_V16)) ? lane16 : (unsigned)((wid * 8 + rsub) * Sp * 4 + vsw);    // (512 rows x Sp x 4 B < 4 GiB)
    t.cblane = (unsigned)((lane & 31) * 4) + (lane >= 32 ? xa_sel(a.rs_off, j) : 0u);   // lanes 0-31: key bias, lanes 32-63: key scale
  };
  // fills: K half `hb` (0: Ka, 1: Kb) of tile `t`; with Ka travels the key bias and key scale of the tile (1 piece: 64 x 4 B:
  // 32 biases, 32 scales) into key-bias slot `slot`.  Every address is a wave-uniform 64-bit base
  // (SGPRs) + a loop-invariant 32-bit lane offset: no vector arithmetic per fill.  (The operands are made opaque at every
  // use: otherwise hipcc hoists base + lane offset out of the loop as a 64-bit per-lane pointer and pays vector adds per fill.)
  auto fill_k = [&](auto fc, const Tile& t, int hb, int slot) __attribute__((always_inline)) {
    constexpr bool K16 = (decltype(fc)::value & XA_K16) != 0;
    if (XA_ABLATE & 1) return;
    if constexpr (K16) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        unsigned kl = kfill_lane;
        const char* b = t.k + hb * 16384 + (wid + 8 * n) * 1024;
        asm volatile("" : "+v"(kl), "+s"(b));
        __builtin_amdgcn_global_load_lds((gptr_t)(b + kl), (lptr_t)(smem + KOFF + hb * 16384 + (wid + 8 * n) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        unsigned kl = kfill_lane;
        const char* b = t.k + hb * (4 * CFD_D * 4) + n * 512;
        asm volatile("" : "+v"(kl), "+s"(b));
        __builtin_amdgcn_global_load_lds((gptr_t)(b + kl), (lptr_t)(smem + kdst_wave + hb * 2048 + n * 16384), 16, 0, 0);
      }
    }
    if (hb == 0) {
      unsigned cl = t.cblane;
      asm volatile("" : "+v"(cl));
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(t.cb) + cl), (lptr_t)(smem + XA_CBOFF + slot * 256), 4, 0, 0);
    }
  };
  auto fill_v = [&](auto fc, const Tile& t, int hb) __attribute__((always_inline)) {
    constexpr bool V16 = (decltype(fc)::value & XA_V16) != 0;
    if (XA_ABLATE & 1) return;
    if constexpr (V16) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        unsigned vl = t.vlane;
        const int g = wid + 16 * n + 8 * hb;                       // 16-row group (1 KB)
        const char* b = t.v + g * 1024;
        asm volatile("" : "+v"(vl), "+s"(b));
        __builtin_amdgcn_global_load_lds((gptr_t)(b + vl), (lptr_t)(smem + VOFF + g * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        unsigned vl = t.vlane;
        const int g = 8 * (n & 1) + 32 * (n >> 1) + 16 * hb;       // uniform part of the 8-row group index (+ wid per wave)
        const char* b = t.v + (long long)g * 8 * t.rowb;
        asm volatile("" : "+v"(vl), "+s"(b));
        __builtin_amdgcn_global_load_lds((gptr_t)(b + vl), (lptr_t)(smem + VOFF + (wid + g) * 1024), 16, 0, 0);
      }
    }
  };

  // x[token][256 half ..] += O^T (+ bias) for this wave's queries, then O = 0.  Every wave re-lays its 256 x 16 tile
  // through a private LDS strip (the caller has drained every fill and passed a barrier, so the tile buffers are free)
  // and moves whole 1 KB row pieces.
  // The residual rows a flush adds to (and, in layer 0's second list, the stored results of the first): requested by flush_request(), which
  // the FINAL flush calls in front of the kernel's last wait + barrier -- their round trip then runs under the barrier's skew and the
  // trailing fills' landing instead of behind them; a flush between two online memories requests them itself.
  float4 fl_old[16], fl_extra[16];
  bool fl_requested = false;
  auto flush_request = [&](bool add_bias) __attribute__((always_inline)) {
    const XAttnArgs* ka = reinterpret_cast<const XAttnArgs*>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
    asm volatile("" : "+s"(ka));
    const float* xp = ka->x + tok0 * CFD_D + half * 256 + lane * 4;
    const float* const dd_in = ka->dd_in;
    const bool store_only = add_bias && ka->dd_out != nullptr;                  // (wave-uniform)
    const bool add_extra = add_bias && dd_in != nullptr && my_aux >= 0;
    const long long dd_off = ((long long)max(my_aux, 0) * a.L + my_q0) * CFD_D + half * 256 + lane * 4;
    // all 16 rows are requested before the first is used: one memory round trip per flush instead of four
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      fl_old[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      fl_extra[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nq && !store_only) fl_old[r] = *reinterpret_cast<const float4*>(xp + (long long)r * CFD_D);
      if (r < nq && add_extra) fl_extra[r] = *reinterpret_cast<const float4*>(dd_in + dd_off + (long long)r * CFD_D);
    }
    fl_requested = true;
  };
  auto flush = [&](bool add_bias) __attribute__((always_inline)) {
    // The arguments only the flush needs are re-read from the kernel-argument segment HERE (through a pointer the compiler cannot see
    // through): kept in SGPRs across the main loop they push the loop's own wave-uniform pointers into VGPRs and from there into
    // scratch memory (68 bytes per lane before this; tests/test_cabi_and_host.py checks that no kernel of the library has any).
    const XAttnArgs* ka = reinterpret_cast<const XAttnArgs*>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());   // (constant -> flat address: the same bits)
    asm volatile("" : "+s"(ka));
    constexpr int RS = 256 * 4 + 16;
    static_assert(XA_WAVES * 16 * RS <= XA_XOFF, "the epilogue strips must not reach the exchange / key-bias / segment areas");
    char* strip = smem + wid * (16 * RS);
#pragma unroll
    for (int f = 0; f < 16; ++f)
      *reinterpret_cast<f32x4*>(strip + l15 * RS + (f * 16 + q4 * 4) * 4) = o[f];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private strip: no barrier needed
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 vbv[CFD_NMEM];
    if (add_bias) {
      bv = *reinterpret_cast<const float4*>(ka->bias + half * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < CFD_NMEM; ++j)
        vbv[j] = *reinterpret_cast<const float4*>(ka->vb[j] + (long long)trow * ka->vb_stride[j] + half * 256 + lane * 4);
    }
    float* xp = ka->x + tok0 * CFD_D + half * 256 + lane * 4;
    float4 one_v = make_float4(0.f, 0.f, 0.f, 0.f);   // rs_u VA_u of the one-key memory for this lane's 4 features
    if (add_bias && ka->one_j >= 0 && my_one >= 0) {
      const float rs1 = ka->one_rs[(long long)my_one * ka->one_sp];
      const float4 va = *reinterpret_cast<const float4*>(ka->one_va + (long long)my_one * CFD_D + half * 256 + lane * 4);
      one_v = make_float4(rs1 * va.x, rs1 * va.y, rs1 * va.z, rs1 * va.w);
    }
    float* const dd_out = ka->dd_out;
    const bool store_only = add_bias && dd_out != nullptr;                      // (wave-uniform)
    const long long dd_off = ((long long)max(my_aux, 0) * a.L + my_q0) * CFD_D + half * 256 + lane * 4;
    if (!fl_requested) flush_request(add_bias);
    fl_requested = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(strip + r * RS + lane * 16);
      if (r < nq) {
        float4 t = fl_old[r];
        float4 c = make_float4(v[0], v[1], v[2], v[3]);
        if (add_bias) {   // + (sum_s P'_s) VV b of every memory (the final flush only: the sums of all memories are complete)
#pragma unroll
          for (int j = 0; j < CFD_NMEM; ++j) {
            const float wj = wq_mine[r * 5 + j];
            c.x += wj * vbv[j].x; c.y += wj * vbv[j].y; c.z += wj * vbv[j].z; c.w += wj * vbv[j].w;
          }
          c.x += one_v.x; c.y += one_v.y; c.z += one_v.z; c.w += one_v.w;
        }
        if (store_only) {
          *reinterpret_cast<float4*>(dd_out + dd_off + (long long)r * CFD_D) = c;
        } else {
          const float4 e = fl_extra[r];
          t.x = ((t.x + bv.x) + e.x) + c.x; t.y = ((t.y + bv.y) + e.y) + c.y; t.z = ((t.z + bv.z) + e.z) + c.z; t.w = ((t.w + bv.w) + e.w) + c.w;
          *reinterpret_cast<float4*>(xp + (long long)r * CFD_D) = t;
        }
      }
    }
#pragma unroll
    for (int f = 0; f < 16; ++f) o[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!add_bias) XA_WAIT_VM(0);                        // the counted waits of the loop assume an empty queue (the final flush's stores are left in flight)
  };

  float m = -INFINITY, lsum = 0.f, wl = 0.f, mc_run = -INFINITY;   // online softmax: maximum, sums, and the exponent reference the sums are relative to
  char* xch_mine = smem + XA_XOFF + wid * 2048 + lane * 16;
  const char* xch_other = smem + XA_XOFF + partner * 2048 + lane * 16;
  // fragment reads: half `hf` (4 k-steps / 4 feature tiles) of a sub-phase -> 8 fragments (hi, lo alternating)
  // pairs: rows of 128 B, hi chunk q4 / lo chunk 4 + q4 of row l15 at (chunk ^ l15 >> 1); single fp16: rows of 64 B, chunk q4 at (q4 ^ (l15 >> 2) & 3)
  auto set_format = [&](bool k16, bool v16) __attribute__((always_inline)) {
    int ln = lane;
    if constexpr (OPF != 0) asm volatile("" : "+v"(ln));      // (opaque: computed where the segment starts, not hoisted for both formats)
    const int r15 = ln & 15, g4 = ln >> 4;
    const int s16 = (g4 ^ ((r15 >> 2) & 3)) << 4, sh = (g4 ^ (r15 >> 1)) << 4, sl = ((4 + g4) ^ (r15 >> 1)) << 4;
    kf_a = smem + KOFF + (k16 ? (8 * half) * 1024 + r15 * 64 + s16 : (8 * half) * 4096 + r15 * 128 + sh);
    kf_b = smem + KOFF + (8 * half) * 4096 + r15 * 128 + sl;
    vf_a = smem + VOFF + (v16 ? (16 * half * 16 + r15) * 64 + s16 : (16 * half * 16 + r15) * 128 + sh);
    vf_b = smem + VOFF + (16 * half * 16 + r15) * 128 + sl;
    kfill_lane = k16 ? (unsigned)ln * 16u : (unsigned)ksrc_lane;
  };
  set_format(false, false);
  // (pairs: fr[2 i] = hi, fr[2 i + 1] = lo of fragment i; single fp16: fr[i] = the fragment, fr[4..7] unused)
  auto read_k = [&](auto fc, spx8 (&fr)[8], int t, int hf) __attribute__((always_inline)) {
    constexpr bool K16 = (decltype(fc)::value & XA_K16) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (K16) {
        fr[i] = XA_FRAG(kf_a + t * 16384 + (4 * hf + i) * 1024);
      } else {
        fr[2 * i] = XA_FRAG(kf_a + (4 * hf + i) * 4096 + t * 2048);
        fr[2 * i + 1] = XA_FRAG(kf_b + (4 * hf + i) * 4096 + t * 2048);
      }
    }
  };
  auto read_v = [&](auto fc, spx8 (&fr)[8], int qf) __attribute__((always_inline)) {   // qf = 0..3: feature tiles 4 qf .. 4 qf + 3 of this half
    constexpr bool V16 = (decltype(fc)::value & XA_V16) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (V16) {
        fr[i] = XA_FRAG(vf_a + (4 * qf + i) * 1024);
      } else {
        fr[2 * i] = XA_FRAG(vf_a + (4 * qf + i) * 2048);
        fr[2 * i + 1] = XA_FRAG(vf_b + (4 * qf + i) * 2048);
      }
    }
  };
  auto mfma_k = [&](auto fc, f32x4& acc, const spx8 (&fr)[8], int hf) __attribute__((always_inline)) {
    constexpr bool K16 = (decltype(fc)::value & XA_K16) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (K16) {
        if constexpr ((decltype(fc)::value & XA_Q16) == 0) acc = XA_MFMA(fr[i], ql[4 * hf + i], acc);
        acc = XA_MFMA(fr[i], qh[4 * hf + i], acc);
      } else {
        acc = XA_MFMA(fr[2 * i + 1], qh[4 * hf + i], acc);
        acc = XA_MFMA(fr[2 * i], ql[4 * hf + i], acc);
        acc = XA_MFMA(fr[2 * i], qh[4 * hf + i], acc);
      }
    }
  };
  spx8 ph, pl;
  auto mfma_v = [&](auto fc, const spx8 (&fr)[8], int qf) __attribute__((always_inline)) {
    constexpr bool V16 = (decltype(fc)::value & XA_V16) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (V16) {
        if constexpr ((decltype(fc)::value & XA_P16) == 0) o[4 * qf + i] = XA_MFMA(fr[i], pl, o[4 * qf + i]);
        o[4 * qf + i] = XA_MFMA(fr[i], ph, o[4 * qf + i]);
      } else {
        o[4 * qf + i] = XA_MFMA(fr[2 * i + 1], ph, o[4 * qf + i]);
        o[4 * qf + i] = XA_MFMA(fr[2 * i], pl, o[4 * qf + i]);
        o[4 * qf + i] = XA_MFMA(fr[2 * i], ph, o[4 * qf + i]);
      }
    }
  };
  // softmax of one key tile from the pair's two partial score sets (this wave's s0 / s1 in registers, the partner's in the exchange area):
  // leaves P' (hi / lo) in ph / pl, updates the running maximum / sums and rescales O for an online memory
  // ATT: where this lane's 8 probabilities of key tile kt of memory j go (queries beyond L store nothing)
  auto att_store = [&](int j, int kt, const float* pv, float mc) __attribute__((always_inline)) {
    const XAttnArgs* ka = reinterpret_cast<const XAttnArgs*>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());   // (as in the flush: nothing kept in SGPRs across the loop)
    asm volatile("" : "+s"(ka));
    const XaAtt* ap = ka->att;
    if (l15 < nq) {
      const long long qi = (long long)my_att * ka->L + my_q0 + l15;
      float* rp = ap->raw + qi * ap->sp_tot + ap->off[j] + kt * XA_KEYS + q4 * 8;
      *reinterpret_cast<float4*>(rp) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *reinterpret_cast<float4*>(rp + 4) = make_float4(pv[4], pv[5], pv[6], pv[7]);
      if (q4 == 0) ap->mc[qi * ap->nt + ap->t0[j] + kt] = mc;
    }
  };
  auto att_finish = [&](int j, float mc_fin, float inv_sum) __attribute__((always_inline)) {
    const XAttnArgs* ka = reinterpret_cast<const XAttnArgs*>((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
    asm volatile("" : "+s"(ka));
    const XaAtt* ap = ka->att;
    if (l15 < nq && q4 == 0) {
      float* fp = ap->fin + (((long long)my_att * ka->L + my_q0 + l15) * CFD_NMEM + j) * 2;
      fp[0] = mc_fin; fp[1] = inv_sum;
    }
  };
  auto softmax_tile = [&](const f32x4& s0, const f32x4& s1, int slot, bool online, int j, int kt) __attribute__((always_inline)) {
    float scale = 1.0f;
    // softmax of this tile: lane (q, g) holds keys 8 g + e, e = 0..7 (s0 = e 0..3, s1 = e 4..7)
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(xch_other);
    const f32x4 t1 = *reinterpret_cast<const f32x4*>(xch_other + 1024);
    const f32x4 kb0 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + q4 * 32);
    const f32x4 kb1 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + q4 * 32 + 16);
    const f32x4 rs0 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + 128 + q4 * 32);
    const f32x4 rs1 = *reinterpret_cast<const f32x4*>(smem + XA_CBOFF + slot * 256 + 128 + q4 * 32 + 16);
    const float rs[8] = {rs0[0], rs0[1], rs0[2], rs0[3], rs1[0], rs1[1], rs1[2], rs1[3]};
    float p[8];
    p[0] = fmaf(s0[0] + t0[0], rs[0], kb0[0]); p[1] = fmaf(s0[1] + t0[1], rs[1], kb0[1]);
    p[2] = fmaf(s0[2] + t0[2], rs[2], kb0[2]); p[3] = fmaf(s0[3] + t0[3], rs[3], kb0[3]);
    p[4] = fmaf(s1[0] + t1[0], rs[4], kb1[0]); p[5] = fmaf(s1[1] + t1[1], rs[5], kb1[1]);
    p[6] = fmaf(s1[2] + t1[2], rs[6], kb1[2]); p[7] = fmaf(s1[3] + t1[3], rs[7], kb1[3]);
    const float mx = xlane_max(fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7]))));
    // exp(x - m) = exp2(x c - m c), c = log2(e): one fused multiply-add and one v_exp_f32 per key (the rounding of m c is common
    // to all keys of a row and cancels against the row sum); dead keys carry x = -inf -> 0
    constexpr float LOG2E = 1.44269504088896340736f;
    if (online) {
      const float m_new = fmaxf(m, mx);
      const bool dead = m_new == -INFINITY;             // nothing but dead keys so far: contribute 0, keep m = -inf
      const float mc = dead ? 0.f : m_new * LOG2E;      // (-inf - (-inf) would be NaN)
      // The running sums are relative to the ROUNDED reference mc of the tile that wrote them (mc_run): the factor to this tile's reference is
      // exp2(mc_run - mc), exactly 1 while the maximum stands.  (Until round 5: exp2(fma(m, c, -mc)), the exact m c against the rounded one --
      // 2^(rounding error of m c) per tile instead of 1, compounding over the 47 tiles of the audio memory; attn_fused.hpp has the figures.)
      scale = dead ? 1.0f : __builtin_amdgcn_exp2f(mc_run - mc);     // (first live tile: mc_run = -inf -> 0, times sums that are still 0)
      mc_run = dead ? -INFINITY : mc;
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p[e] = __builtin_amdgcn_exp2f(fmaf(p[e], LOG2E, -mc));
        ps += p[e];
      }
      lsum = lsum * scale + xlane_sum(ps);
      m = m_new;
      if constexpr (ATT) { if (att_on) att_store(j, kt, p, mc); }     // (a dead tile: p = 0 against the reference 0)
      float pw = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { p[e] *= rs[e]; pw += p[e]; }     // P' = p rs: the operand of VA^T P'
      wl = wl * scale + pw;
    } else {
      const float mc = mx * LOG2E;
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        p[e] = __builtin_amdgcn_exp2f(fmaf(p[e], LOG2E, -mc));   // all keys dead: (-inf) - (-inf) = NaN, as in the reference
        ps += p[e];
      }
      const float inv = 1.0f / xlane_sum(ps);
#pragma unroll
      for (int e = 0; e < 8; ++e) p[e] = p[e] * inv;
      if constexpr (ATT) { if (att_on) { att_store(j, kt, p, 0.f); att_finish(j, 0.f, 1.0f); } }
#pragma unroll
      for (int e = 0; e < 8; ++e) { p[e] = p[e] * rs[e]; wl += p[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {      // p <= 1 and rs <= 1 / sqrt(eps) = 316: no saturation needed in front of the fp16 split
      const sp_t hi = (sp_t)p[e];
      ph[e] = hi;
      pl[e] = (sp_t)(p[e] - (float)hi);
    }
    if (online && !__all(scale == 1.0f)) {
#pragma unroll
      for (int f = 0; f < 16; ++f) { o[f][0] *= scale; o[f][1] *= scale; o[f][2] *= scale; o[f][3] *= scale; }
    }
  };
  spx8 fa[8], fb[8];   // the two fragment sets
  bool primed = false;
  int step = 0;
  Tile cur, nseg_t;
  int cT = 1, cmask = 0, cflags = 0, cj = 0, nT = 1, nmask = 0, nflags = 0, nj = 0;
  // A whole single-fp16 K tile (+ its key-bias piece) / V^T tile into slot `sl` of its buffer: 4 (+1) / 4 pieces per wave (XA_DBUF instances)
  auto fill_k_full = [&](const Tile& t, int sl) __attribute__((always_inline)) {
    if (XA_ABLATE & 1) return;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      unsigned kl = lane16;
      const char* b = t.k + (wid + 8 * n) * 1024;
      asm volatile("" : "+v"(kl), "+s"(b));
      __builtin_amdgcn_global_load_lds((gptr_t)(b + kl), (lptr_t)(smem + KOFF + sl * 32768 + (wid + 8 * n) * 1024), 16, 0, 0);
    }
    unsigned cl = t.cblane;
    asm volatile("" : "+v"(cl));
    __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(t.cb) + cl), (lptr_t)(smem + XA_CBOFF + sl * 256), 4, 0, 0);
  };
  auto fill_v_full = [&](const Tile& t, int sl) __attribute__((always_inline)) {
    if (XA_ABLATE & 1) return;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      unsigned vl = lane16;
      const char* b = t.v + (wid + 8 * n) * 1024;
      asm volatile("" : "+v"(vl), "+s"(b));
      __builtin_amdgcn_global_load_lds((gptr_t)(b + vl), (lptr_t)(smem + VOFF + sl * 32768 + (wid + 8 * n) * 1024), 16, 0, 0);
    }
  };
  if (nseg > 0) seg_tile(0, cur, cT, cmask, cflags, cj);
  if constexpr ((OPF & XA_DBUF) != 0) {
    // the first tile of a list that starts with a long memory: requested HERE, in front of the rest of the prologue.  (Slot 0 of both tile
    // buffers: the parked A b / norm2 parameters sit beyond the V^T tile's 32 KB; nobody reads the tile buffers before the first step's B0.)
    if (nseg > 0 && wgp->n16 > 0) {
      fill_k_full(cur, 0);
      fill_v_full(cur, 0);
      primed = true;
    }
  }
  XA_T(13);
  finish_queries();
  XA_T(0);
  // One key-tile step in the format `fc` (fmt_long / fmt_pair): the tile `cur` and the tile after it, `nxt`, are BOTH in that format
  // (the segment loop below sees to it), so every piece count behind a counted wait is a compile-time constant of the instance.
  bool in_seg = false, online = false;
  float cqh = 0.f;
  auto kt_step = [&](auto fc, int kt) __attribute__((always_inline)) {
    constexpr int F = decltype(fc)::value;
    constexpr bool V16 = (F & XA_V16) != 0, K16 = (F & XA_K16) != 0;
    constexpr int NKP = K16 ? 2 : 4, NVP = V16 ? 2 : 4;   // pieces per wave and sub-buffer: what the counted waits count
    const bool last_in_seg = kt + 1 == cT;
    Tile nxt;   // the tile of the step after this one
    nxt.k = last_in_seg ? nseg_t.k : cur.k + (K16 ? 32768 : XA_KEYS * CFD_D * 4);
    nxt.v = last_in_seg ? nseg_t.v : cur.v + (V16 ? 32768 : 128);
    nxt.cb = last_in_seg ? nseg_t.cb : cur.cb + XA_KEYS;
    nxt.rowb = last_in_seg ? nseg_t.rowb : cur.rowb;
    nxt.vlane = last_in_seg ? nseg_t.vlane : cur.vlane;
    nxt.cblane = last_in_seg ? nseg_t.cblane : cur.cblane;
    const int slot = step & 1;
    if (!primed) {   // (re)start of the pipeline: Ka (+ key bias), Kb, Va of this step; Vb follows behind mid-A0
      fill_k(fc, cur, 0, slot);
      fill_k(fc, cur, 1, slot);
      fill_v(fc, cur, 0);
      XA_WAIT_VM_LGKM0(NKP + NVP);       // Ka + key bias landed (Kb and Va are younger)
      __builtin_amdgcn_s_barrier();
      read_k(fc, fa, 0, 0);
      primed = true;
      XA_T(0);
    }
    f32x4 s0 = f32x4{cqh, cqh, cqh, cqh}, s1 = s0;
    // ---- A0 (fa holds its first half) --------------------------------------------------------------------------------
    read_k(fc, fb, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_k(fc, s0, fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    XA_T(1);
    XA_WAIT_VM_LGKM0(NVP);               // Kb landed (Va's pieces are younger); this wave's reads of Ka's first half are done
    __builtin_amdgcn_s_barrier();        // mid-A0: Kb ready; every wave is done with Vb
    XA_T(2);
    fill_v(fc, cur, 1);
    read_k(fc, fa, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) {
      mfma_k(fc, s0, fb, 1);
      *reinterpret_cast<f32x4*>(xch_mine) = s0;
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- A1 ---------------------------------------------------------------------------------------------------------
    read_k(fc, fb, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_k(fc, s1, fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) {
      mfma_k(fc, s1, fb, 1);
      *reinterpret_cast<f32x4*>(xch_mine + 1024) = s1;
    }
    XA_T(3);
    XA_WAIT_VM_LGKM0(0);                 // Va and Vb landed (nothing younger is in flight); partial scores written
    __builtin_amdgcn_s_barrier();        // end of A1: the whole V^T tile ready, partial scores visible, every wave is done with Ka and Kb
    XA_T(4);
    fill_k(fc, nxt, 0, slot ^ 1);
    fill_k(fc, nxt, 1, slot ^ 1);
    read_v(fc, fa, 0);
    if (in_seg && !(XA_ABLATE & 8)) softmax_tile(s0, s1, slot, online, cj, kt);
    // ---- B0 (fa holds its first half) --------------------------------------------------------------------------------
    XA_T(10);
    read_v(fc, fb, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_v(fc, fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    XA_T(5);
    read_v(fc, fa, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_v(fc, fb, 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- B1 ---------------------------------------------------------------------------------------------------------
    read_v(fc, fb, 3);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_v(fc, fa, 2);
    __builtin_amdgcn_sched_barrier(0);
    XA_T(7);
    XA_WAIT_VM_LGKM0(NKP);               // next Ka + key bias landed (the next Kb's pieces are younger)
    __builtin_amdgcn_s_barrier();        // mid-B1: next Ka ready; every wave is done with Va
    XA_T(8);
    fill_v(fc, nxt, 0);
    read_k(fc, fa, 0, 0);                // first half of the next step's A0
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_v(fc, fb, 3);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
    ++step;
    XA_T(9);
  };
  // One key-tile step with BOTH tiles as single fp16, double-buffered (instances with XA_DBUF).  A K tile and a V^T tile are 32 KB each, so
  // the two 64 KB tile buffers hold TWO of each: step n computes out of slot n & 1 while tile n + 1 lands in the other slot -- its K
  // requested behind the step's first barrier, its V^T behind the second, i.e. every fill has a whole step (four sub-phases) to land instead
  // of 1.5 - 2.5, and a step has TWO barriers instead of three:
  //   B0  K(n) has landed and is visible; every wave is done with the other slot's tiles (K since B1 of step n - 1, V^T just now)
  //   B1  V^T(n) has landed; the pair's partial scores are in the exchange area
  // A wave's requests complete in order: behind B0 it has K(n + 1) (4 pieces + the key-bias piece) in flight behind V^T(n) (4), behind B1
  // V^T(n + 1) behind K(n + 1) -- which is what the two counted waits count.
  auto kt_step_db = [&](int kt) __attribute__((always_inline)) {
    const bool last_in_seg = kt + 1 == cT;
    Tile nxt;
    nxt.k = last_in_seg ? nseg_t.k : cur.k + 32768;
    nxt.v = last_in_seg ? nseg_t.v : cur.v + 32768;
    nxt.cb = last_in_seg ? nseg_t.cb : cur.cb + XA_KEYS;
    nxt.rowb = cur.rowb;
    nxt.vlane = cur.vlane;
    nxt.cblane = last_in_seg ? nseg_t.cblane : cur.cblane;
    const int slot = step & 1;
    if (!primed) {   // (re)start of the pipeline
      fill_k_full(cur, slot);
      fill_v_full(cur, slot);
      primed = true;
    }
    const char* kb = kf_a + slot * 32768;
    const char* vb = vf_a + slot * 32768;
    auto rd_k = [&](spx8 (&fr)[8], int t, int hf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fr[i] = XA_FRAG(kb + t * 16384 + (4 * hf + i) * 1024);
    };
    auto rd_v = [&](spx8 (&fr)[8], int qf) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fr[i] = XA_FRAG(vb + (4 * qf + i) * 1024);
    };
    f32x4 s0 = f32x4{cqh, cqh, cqh, cqh}, s1 = s0;
    XA_T(1);
    XA_WAIT_VM_LGKM0(4);                 // K(n) landed (V^T(n)'s 4 pieces are younger)
    __builtin_amdgcn_s_barrier();        // B0
    XA_T(2);
    rd_k(fa, 0, 0);
    rd_k(fb, 0, 1);
    fill_k_full(nxt, slot ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_k(fmt_long{}, s0, fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    rd_k(fa, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) {
      mfma_k(fmt_long{}, s0, fb, 1);
      *reinterpret_cast<f32x4*>(xch_mine) = s0;
    }
    __builtin_amdgcn_sched_barrier(0);
    rd_k(fb, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_k(fmt_long{}, s1, fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) {
      mfma_k(fmt_long{}, s1, fb, 1);
      *reinterpret_cast<f32x4*>(xch_mine + 1024) = s1;
    }
    XA_T(3);
    XA_WAIT_VM_LGKM0(5);                 // V^T(n) landed (K(n + 1): 4 pieces + key bias are younger); partial scores written
    __builtin_amdgcn_s_barrier();        // B1
    XA_T(4);
    rd_v(fa, 0);
    fill_v_full(nxt, slot ^ 1);
    if (in_seg && !(XA_ABLATE & 8)) softmax_tile(s0, s1, slot, online, cj, kt);
    XA_T(10);
    rd_v(fb, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_v(fmt_long{}, fa, 0);
    __builtin_amdgcn_sched_barrier(0);
    XA_T(5);
    rd_v(fa, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_v(fmt_long{}, fb, 1);
    __builtin_amdgcn_sched_barrier(0);
    rd_v(fb, 3);
    __builtin_amdgcn_sched_barrier(0);
    if (in_seg) mfma_v(fmt_long{}, fa, 2);
    __builtin_amdgcn_sched_barrier(0);
    XA_T(7);
    if (in_seg) mfma_v(fmt_long{}, fb, 3);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
    ++step;
    XA_T(9);
  };
  // The segments [s0, s1) of the workgroup's list, all in the format `fc`.
  auto seg_loop = [&](auto fc, int s0, int s1) __attribute__((always_inline)) {
  for (int si = s0; si < s1; ++si) {
    in_seg = active && ((cmask >> tile) & 1);   // wave-uniform, the same for both waves of a pair
    online = (cflags & XA_ONLINE) != 0;
    const bool seg_follows = si + 1 < s1;
    // the first tile of the next segment (this segment's first tile again when there is none in this format: the trailing fills of the
    // last step then land in buffers nobody reads)
    nseg_t = cur; nT = cT; nmask = cmask; nflags = cflags; nj = cj;
    if (seg_follows) seg_tile(si + 1, nseg_t, nT, nmask, nflags, nj);
    // this wave's half of c_q for the segment's memory: the score accumulators START from it, so the pair's partial scores already
    // add up to S_raw + c_q (a + b = b + a: the same in both waves of the pair)
    cqh = cq_mine[l15 * 5 + cj];
    XA_T(11);
    if constexpr ((decltype(fc)::value & XA_DBUF) != 0) {
      for (int kt = 0; kt < cT; ++kt) kt_step_db(kt);
    } else {
      for (int kt = 0; kt < cT; ++kt) kt_step(fc, kt);
    }
    if (in_seg) {
      float wsum = xlane_sum(wl);
      if (online) {   // normalise the finished online memory in registers (all keys dead: 0 * inf = NaN)
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int f = 0; f < 16; ++f) { o[f][0] *= inv; o[f][1] *= inv; o[f][2] *= inv; o[f][3] *= inv; }
        wsum *= inv;
        if constexpr (ATT) { if (att_on) att_finish(cj, mc_run, inv); }
      }
      if (q4 == 0) wq_mine[l15 * 5 + cj] = wsum;
    }
    if (cflags & XA_FLUSH) {   // one accumulator: hand the finished online memory to x before the next one starts
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // every wave is done with Vb and every fill has landed: the strips may alias the tile buffers
      flush(false);
      XA_WAIT_VM_LGKM0(0);
      __builtin_amdgcn_s_barrier();   // strips read back everywhere before the pipeline is primed again
      primed = false;
    }
    m = -INFINITY;
    mc_run = -INFINITY;
    lsum = 0.f;
    wl = 0.f;
    cT = nT; cmask = nmask; cflags = nflags; cj = nj;   // (cur already points at the next segment's first tile)
  }
  };
  if constexpr (OPF != 0) {
    // The host lists a workgroup's long memories first (make_xattn_worklist): the segments with single-fp16 tiles are a PREFIX of the list,
    // n16 of them.  Two loops one behind the other -- not two bodies inside one loop, which hipcc could not fit into 256 registers -- with
    // the pipeline drained and primed again in between (once per workgroup).
    const int n16 = min(wgp->n16, nseg);
    if (n16 > 0) {
      set_format((OPF & XA_K16) != 0, (OPF & XA_V16) != 0);
      seg_loop(fmt_long{}, 0, n16);
    }
    if (n16 < nseg) {
      if (n16 > 0) {
        XA_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();   // nothing of the other format in flight or in use when the pair pipeline starts
        primed = false;
        set_format(false, false);
        seg_tile(n16, cur, cT, cmask, cflags, cj);
      }
      seg_loop(fmt_pair{}, n16, nseg);
    }
  } else {
    seg_loop(fmt_pair{}, 0, nseg);
  }
  flush_request(true);
  XA_WAIT_VM_LGKM0(0);
  __builtin_amdgcn_s_barrier();   // last B1 done everywhere and the trailing (unused) fills have landed: the tile buffers become the epilogue strips
  XA_T(11);
  flush(true);
#if XA_STAMP
  {
    const long long t_ = __builtin_amdgcn_s_memtime();
    acc_[6] += t_ - tprev_;   // the final flush
    if (a.stamps && lane == 0)
      for (int k = 0; k < XA_NSTAMP; ++k) a.stamps[((long long)blockIdx.x * XA_WAVES + wid) * XA_NSTAMP + k] = acc_[k];
  }
#endif
}

// The maps of one step from what the ATT instance stored (XaAtt), into slot *d_step of the caller's ring: block [nb][nl][L][S_j] per memory.
//   p(s) = raw(s) 2^(mc_tile - mc_final) / sum        (raw = 0 stays 0 whatever the exponents: a dead tile in front of live ones;
//                                                      every key dead: 0 x (1 / 0) = NaN like the reference's softmax)
// A one-key memory that the work lists skip (XAttnArgs::one_j) has probability 1.
struct XaFixArgs {
  const XaAtt* att;            // [nl]
  int nl, L, one_j;
  int S[CFD_NMEM];
  float* ring[CFD_NMEM];
  long long slot[CFD_NMEM];    // floats per ring slot
  const int* d_step;
};
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) att_fixup_kernel(const XaFixArgs a) {
  const int l = blockIdx.y;
  const XaAtt at = a.att[l];
  const long long qi = blockIdx.x;                 // (row r, query q): r * L + q
  const int r = (int)(qi / a.L), q = (int)(qi - (long long)r * a.L);
  const int step = *a.d_step;
#pragma unroll
  for (int j = 0; j < CFD_NMEM; ++j) {
    const int S = xa_sel(a.S, j);
    float* out = xa_sel(a.ring, j);
    if (!out) continue;
    out += (long long)step * xa_sel(a.slot, j) + (((long long)r * a.nl + l) * a.L + q) * S;
    if (j == a.one_j) { if (threadIdx.x == 0) out[0] = 1.0f; continue; }
    const float mcf = at.fin[(qi * CFD_NMEM + j) * 2], inv = at.fin[(qi * CFD_NMEM + j) * 2 + 1];
    for (int s = threadIdx.x; s < S; s += 256) {
      const float raw = at.raw[qi * at.sp_tot + at.off[j] + s];
      const float mct = at.mc[qi * at.nt + at.t0[j] + (s >> 5)];
      out[s] = (raw == 0.f ? 0.f : raw * __builtin_amdgcn_exp2f(mct - mcf)) * inv;
    }
  }
}
