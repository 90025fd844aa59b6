// libcfdenoise internals shared by the translation units of the library (cfd_core / cfd_problem / cfd_forward / cfd_sample / cfd_blocks / cfd_weg /
// cfd_dev .hip): the handle, the per-problem workspace, error and launch helpers, and the functions one unit calls in another.
// gfx950 only; no CPU fallback anywhere.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/cfdenoise.h"
#include "../../include/cfdenoise_dev.h"
#include "gemm_sp.hpp"
#include "rows.hpp"
#include "attn_fused.hpp"
#include "xattn_fused.hpp"
#include "rowtile.hpp"


int fail(int code, const char* fmt, ...);      // cfd_core.hip: formats the thread's last-error text (cfd_last_error) and returns `code`
#define HIPCHK(expr)                                                                               \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(CFD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define CHK(expr)            \
  do {                       \
    int _r = (expr);         \
    if (_r != CFD_OK) return _r; \
  } while (0)

static const char* MEM_NAMES[CFD_NMEM] = {"spkemb", "alsn", "tlsn", "apb", "lsnemb"};

struct DBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t n) {
    if (n <= bytes) return CFD_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    HIPCHK(hipMalloc(&p, n));
    bytes = n;
    return CFD_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct LayerW {
  DBuf wqk_sp, bqk, wv_sp, wo_sp, bo2, wtb1_sp, wtb2_sp, w1_sp, w2_sp, cross_bias;
  // the LayerNorm fold's operands (gemm_sp.hpp EpiLn): W' = W diag(gamma) as split pairs for q | k, v and FFN1, and per output feature
  // c = W' 1, d = W beta in ln_cd: [c_qk 1024][d_qk 1024][c_v 512][d_v 512][c_1 1024][d_1 1024]
  DBuf wqk_f, wv_f, w1_f, ln_cd;
  const float *ln1g, *ln1b, *tb1g, *tb1b, *btb1, *ln2g, *ln2b, *tb2g, *tb2b, *btb2, *ln3g, *ln3b, *b1, *b2;
};

struct Problem {
  int Be = 0, L = 0, Lp = 0;
  long long M = 0;
  int U[CFD_NMEM], S[CFD_NMEM], Sp[CFD_NMEM], off[CFD_NMEM], Sp_tot = 0;
  const float* mem[CFD_NMEM];
  const int* map[CFD_NMEM];
  const uint8_t* mask[CFD_NMEM];
  int has_mask[CFD_NMEM];
  float* att[CFD_NMEM];
  // attention ring of a sampling run (cfd_sample_args::att_ring): only the batch rows [att_b0, att_b0 + att_nb) write their maps, into
  // slot *d_step of att[j] (att_slot[j] floats per slot), as rows 0 .. att_nb - 1 of that slot.  att_nb == 0: att[j] is one [Be][nl][L][S_j] block.
  int att_b0 = 0, att_nb = 0;
  bool prev_same = false;       // setup_problem: the workspace still holds the previous cfd_forward's projections of memories of these shapes
  bool att_fused = false;       // the ring is written by the fused cross-attention kernel's ATT instance + att_fixup_kernel (tile kernels)
  long long att_slot[CFD_NMEM] = {0, 0, 0, 0, 0};
  int tmode = 0;  // 0: all rows share the timestep of table row *d_step ; 1: row b uses table row b
  // Sampling loop only: the effective batch is G replicas (chunk-major) of the same B latent rows, so everything
  // before the first cross-attention -- embedding, layer 0's self-attention and first time block -- is identical
  // for the G replicas of an utterance (same input, same timestep; the memories enter only at the cross-attention).
  // It is computed for the first B rows and copied to the other chunks.  0 = off (cfd_forward: arbitrary rows).
  int share_B = 0;
  int T = 1;      // rows in the temb tables
  // Rows that share one memory of the LARGEST memory type in long consecutive runs (the guidance batch repeats
  // the unconditional audio memory for 5 of its 7 chunks): their attention against that memory is one big
  // un-batched product per run instead of a 196-row product per batch row.
  int jbig = -1, nruns = 0, nlong = 0, nshort = 0;
  int run_row0[8], run_len[8], run_u[8];
  // fused cross-attention (xattn_fused.hpp): workgroups of the work list, 0 = the list was not built
  int xa_nwg = 0;
  int xa0_nwg_a = 0, xa0_nwg_b = 0;   // layer-0 de-duplication lists (build_xattn_layer0_lists); 0: one launch
  bool xa_flush = false;               // some work list flushes the accumulator between two online memories (XA_FLUSH): lock-step kernel only
  int xa_one = -1;                     // the one-key memory the fused cross-attention adds as a vector (xattn_fused.hpp, XAttnArgs::one_j), or -1
  int xa_opf = 0;                      // operand format of the fused cross-attention's key tiles of LONG memories in this problem (XA_V16 | XA_K16; 0: split
                                       // pairs).  Only a sampling run sets it (cfd_sample_args::operand_policy), and only when every memory is static and no
                                       // maps are kept
  int xa_f16_mask = 0;                 // bit j: memory j is long enough (XA_F16_MIN_KEYS) for single-fp16 tiles; its segments carry XA_F16
  // memories (bit j) whose folded projections were computed once for the run from the centred static part of the memory
  // (prepare_static_memside); per step they only get their per-key scale and bias (mem_scale_all_kernel)
  int static_mask = 0;
  // small problems (rowtile.hpp): every launch of the forward is a grid of 16-token x 16-feature workgroups; needs every memory static
  bool rt = false;
  int rt_use_inst = 0;                              // the row maps fit the kernel arguments (<= RT_ARG_ROWS rows, instances < 256)
  unsigned char rt_inst[CFD_NMEM][RT_ARG_ROWS];
};

struct RtSave {
  float* x[CFD_MAX_LAYERS + 1][5];   // [l][0] layer input, [1] after self-attention, [2] after time block 1, [3] after cross-attention, [4] after time block 2
  char* qk[CFD_MAX_LAYERS];
  char* vt[CFD_MAX_LAYERS];
  float* sc[CFD_MAX_LAYERS];    // e_s of the cross-attention (rowtile.hpp: rt_xscore_kernel) ...
  float* cst[CFD_MAX_LAYERS];   // ... and its cell statistics
  float* pre[CFD_MAX_LAYERS];
};

// cfd_weg_eval on the row-tile kernels (weg_rt.hpp): the arena of saved activations and gradient buffers
struct WegRtState {
  std::vector<long long> sig;   // shapes and pointers the arena and the problem of wk[1] were prepared for
  RtSave sv;
  float *att = nullptr, *d_att = nullptr, *fws = nullptr, *dP = nullptr, *G[3] = {nullptr, nullptr, nullptr}, *dz = nullptr, *dy = nullptr,
        *dh = nullptr, *dO = nullptr, *dqkv = nullptr;
  int launches = 0;
  int T = 1;                    // rows of wk[1]'s per-timestep tables: 1 (this evaluation's timestep) or every timestep (cfd_weg_args::reuse_memory_side == 2)
};

// One problem's device workspace: everything setup_problem / prepare_static_memside allocate and the launches of a forward touch.
struct Work {
  Problem pb;
  DBuf x, h_sp, qk_sp, vts_sp, ssc, sp_sp, o_sp, u_sp, sc, p_sp, eps, sample_sp;
  DBuf ln_stat;                 // LayerNorm fold: per row 16 slots of (mean, M2) written by the producing residual product (EpiResidStat)
  DBuf n_sp[CFD_NMEM], kall_sp[CFD_NMEM], cb[CFD_NMEM], vt_all[CFD_NMEM];
  DBuf temb_tab, h1_tab, ss_tab, trows, iota, long_rows, short_rows, zero_mask;
  // timestep-independent memory-side projections (rows.hpp mem_center_kernel): per memory the dot products c_l . a_s (ca), |a_s|^2 (asq)
  // and, per table row t, A_l b_t / c_l . b_t (kbtab) and VV_l b_t (vbtab); b_t = centred timestep embedding.  CFD_HOIST_MEMSIDE=0: off.
  DBuf ca[CFD_NMEM], asq[CFD_NMEM], kbtab[CFD_NMEM], vbtab[CFD_NMEM], b_tab, b_sp, bsq, zeros512;
  DBuf xa_wgs, xa_segs, xa_stamps, xa0_wgs_a, xa0_segs_a, xa0_wgs_b, xa0_segs_b, xa_dedup, xa_one_va, xa_att_raw, xa_att_mc, xa_att_fin, xa_att_desc;
  DBuf d_step;  // [0] = loop index, [1] = constant 0, [2] = "this iteration's in-painting overwrite is done" (cfd_sample_inpaint)
  // the timestep-independent memory-side projections the last cfd_forward left in this workspace (cfd_forward_same_memories): valid only
  // from the end of a cfd_forward that made (or reused) all five until the next setup_problem on this workspace
  bool fwd_mem_valid = false;
  int fwd_U[CFD_NMEM] = {0, 0, 0, 0, 0}, fwd_S[CFD_NMEM] = {0, 0, 0, 0, 0}, fwd_Be = 0, fwd_L = 0;
  bool fwd_att = false;
  bool fwd_mask[CFD_NMEM] = {false, false, false, false, false}, fwd_map[CFD_NMEM] = {false, false, false, false, false};
  unsigned long long fwd_wver = 0;
  DBuf rt_vt, rt_cbt[CFD_NMEM];   // row-tile path: V^T of the self-attention, per-step key tables
  DBuf k16[CFD_NMEM], v16[CFD_NMEM];   // single-fp16 key / value tiles of the static memories (xa_pack16_kernel), when pb.xa_opf asks for them
  DBuf rt_cur;                    // row-tile path, sampling run: this step's rows of every per-step table (rt_step_rows_kernel)
  // What the timestep-only tables of this workspace were built from: the table rows' timesteps and the weights' generation.  temb / AdaLN
  // rows (20 launches) and, per memory, A_l b_t / VV_l b_t (kbtab / vbtab: two products each) depend on nothing else, so a run that
  // finds them built for its own timestep list skips them (the rollout opens eleven 1000-step runs per sample, unbounded_synthesis.py:285-468).
  std::vector<int32_t> tt_key;
  long long tt_wver = -1;
  int tt_mem_mask = 0;            // bit j: kbtab[j] / vbtab[j] (and b_tab / b_sp / bsq) hold the products for tt_key
  // row-tile path: where the launches of the current problem find this step's AdaLN rows and A b / VV b vectors (the tables themselves
  // when they have one row, rt_cur otherwise); set by enqueue_rows_rt, read by the WEG reverse sweep (weg_rt.hpp)
  const float* now_ss = nullptr;
  const float* now_kb[CFD_NMEM] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  const float* now_vb[CFD_NMEM] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  void release() {
    DBuf* all[] = {&x, &h_sp, &qk_sp, &vts_sp, &ssc, &sp_sp, &o_sp, &u_sp, &sc, &p_sp, &eps, &sample_sp, &temb_tab, &h1_tab, &ss_tab, &trows, &iota,
                   &long_rows, &short_rows, &zero_mask, &ln_stat, &b_tab, &b_sp, &bsq, &zeros512, &xa_wgs, &xa_segs, &xa_stamps, &xa0_wgs_a, &xa0_segs_a, &xa0_wgs_b, &xa0_segs_b, &xa_dedup, &xa_one_va, &xa_att_raw, &xa_att_mc, &xa_att_fin, &xa_att_desc, &d_step, &rt_vt, &rt_cur};
    for (DBuf* b : all) b->release();
    for (int j = 0; j < CFD_NMEM; ++j) {
      n_sp[j].release(); kall_sp[j].release(); cb[j].release(); vt_all[j].release(); ca[j].release(); asq[j].release(); kbtab[j].release();
      vbtab[j].release(); rt_cbt[j].release(); k16[j].release(); v16[j].release();
    }
  }
};

struct cfd_handle_s {
  cfd_config cfg;
  int nl = 0;
  bool finalized = false;
  std::map<std::string, DBuf> raw;
  std::map<std::string, size_t> raw_numel;
  // prepared weights
  DBuf we_sp, wp_sp, we_all, be_all;
  DBuf wp_f, ln_cd_p;           // the final norm folded into latent_proj: W' as split pairs; [c 128][d 128]
  DBuf wk_all_sp[CFD_NMEM], wv_all_sp[CFD_NMEM];
  std::vector<LayerW> lw;
  int qpe_rows = 0, mpe_rows = 0;
  // timestep sinusoid table
  DBuf tsin;
  int tsin_rows = 0;
  // workspaces (struct Work): wk[0] belongs to cfd_forward / the sampling run (its captured graph holds these pointers), wk[1] to the
  // row-tile WEG evaluation, which runs between two replays of an open run and must not disturb it; `w` is the one in use
  Work wk[2];
  Work* w = &wk[0];
  long long wver = 0;     // generation of the prepared weights (cfd_finalize_weights)
  int setup_launches = 0; // launches the last cfd_sample_begin spent on timestep-only tables (0: served from the cache); test / bench read-out
  // saturation census of THIS handle (cfd_common.hpp): sat[CFD_SAT_MEM] weights / memories / their projections, sat[CFD_SAT_IN] the
  // sample / latents handed to an entry point.  Zeroed at the entry of the calls that count, read at their end.
  DBuf sat;
  bool memside_in_forward = false;   // the last enqueue_denoise ran memory-side projections itself (not hoisted): census still open
  bool run_counts = false;           // the open run's captured iteration contains launches that count into the census (per-step projections)
  unsigned int* sat_mem() const { return sat.as<unsigned int>() + CFD_SAT_MEM; }
  unsigned int* sat_in() const { return sat.as<unsigned int>() + CFD_SAT_IN; }
  bool hoist_memside = true;
  bool use_runs = true;   // CFD_RUNS=0 disables the shared-memory run optimisation of the three-launch attention path
  // The cross-attention block is one fused kernel (xattn_fused.hpp) unless the caller wants att_mats, which only the
  // three-launch path (score products -> softmax_rows_kernel -> P.V products) materialises.  CFD_FUSED_XATTN=0 forces
  // the three-launch path everywhere (parity A/B of the two paths).
  bool fused_xattn = true;
  int fused_xattn_min_wgs = 6;
  int one_key = 1;              // CFD_ONE_KEY=0: a one-key memory (lsnemb) keeps its 32-key tile step in the fused cross-attention
  int want_opf = 0;             // cfd_sample_begin -> setup_problem: the operand policy the run asks for (0 everywhere else)
  int xa_operands = -1;         // CFD_XA_OPERANDS=<0..3>: overrides cfd_sample_args.operand_policy (developer A/B of the fused cross-attention's tile formats)
  int xa_db = -1;               // CFD_XA_DB=0 (developer builds, -DXA_ALL_OPF=1): operand policy 15 on the three-barrier step instead of the double-buffered one
  bool hint_same_mem = false;   // cfd_forward_same_memories: the promise for the NEXT cfd_forward ...
  bool hint_now = false;        // ... taken (and cleared) at that call's very first line, before anything can fail: a call that returns early
                                // must not leave the promise standing for the call after it
  bool census_pending = false;  // a cfd_weg_eval without loss_host left its census unread (it does not wait): settled by the next entry point
  int rt_nfb2_tiles = 14;       // CFD_RT_NFB2_TILES=<token tiles>: from how many token tiles on the row-tile path's 512 x 512 residual products take two feature blocks per workgroup
  int step_rows = 1;            // CFD_STEP_ROWS=0: the tile kernels index the per-step tables with the device step counter themselves
  int att_fused = 1;            // CFD_ATT_FUSED=0: a forward that returns att_mats takes the three-launch cross-attention on the tile kernels (the fused
                                // kernel's ATT instance keeps the maps otherwise: xattn_fused.hpp, XaAtt)
  int ln_fold = -1;             // CFD_LN_FOLD: the algebraic LayerNorm fold of mid-size problems (cfd_forward.hip): -1 by shape, 0 off, 1 wherever the launches allow it
  int qkv_fused = 1;            // CFD_QKV_FUSED=0: batch rows of 16 tokens keep the separate v^T product (EpiQkvT, gemm_sp.hpp); 2: one launch, but
                                // the flash self-attention kernel behind it (1: the row-tile path's attention core)
  int l0_dedup = 1;             // CFD_L0_DEDUP=0: layer 0's cross-attention as one launch over all rows (build_xattn_layer0_lists)
  // Row-tile path for small problems (rowtile.hpp): chosen by SHAPE -- at most rt_max_rows token rows of at most RT_MAX_L tokens per batch
  // row, one timestep for all rows, no dynamic memories.  CFD_ROWTILE=0 turns it off (parity A/B against the tile kernels),
  // CFD_ROWTILE_MAX_ROWS moves the threshold.
  bool rt_on = true;
  long long rt_max_rows = 700;    // measured crossover at the product shape (L = 16), seconds per 1000 steps, row-tile vs tile kernels (profiles/r05_rowtile_crossover.log:
                                  // the short cross-attention work lists of round 5 made the tile kernels faster): 5 utterances 1.04 / 1.22, 6: 1.18 / 1.24, 7: 1.34 / 1.23
  bool share0 = true;       // CFD_SHARE0=0: evaluate the pre-cross-attention part of layer 0 for every guidance replica
  DBuf weg_ws, weg_tok;   // cfd_weg_eval: activation arena, focus-token tables
  // cfd_weg_eval replays its ~400 launches as a hipGraph.  A graph holds its kernels' arguments BY VALUE, so everything the
  // caller passes per call -- latents in, losses / max_att / grad out, the timestep's sinusoid row -- goes through fixed
  // staging buffers (weg_io); round 1's attempt captured the caller's own pointers, which are fresh torch tensors on every
  // call, and so replayed against stale addresses ("wrong gradients when interleaved with the sampling graph").
  // One graph per variant (full evaluation / memory-side results reused), keyed by everything else the launches depend on;
  // a key is run eagerly once (function attributes, warm-up) and captured on its second use.  CFD_WEG_GRAPH=0: always eager.
  DBuf weg_io;
  // Row-tile evaluation (weg_rt.hpp): the product path for small problems; CFD_WEG_ROWTILE=0 keeps the float32 launch sequence of weg_eval.hpp
  bool weg_rt_on = true;
  DBuf weg_rt_ws;
  WegRtState wrt;
  int weg_t_host = 0;   // the evaluation's timestep, copied to wk[1].trows in front of every launch sequence (one-row tables)
  int weg_dstep_host = 0;   // ... and the table row it selects, copied to wk[1].d_step (0 for one-row tables, the timestep for full tables)
  bool weg_graph_on = true;
  struct WegGraph { std::vector<long long> key; int uses = 0; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
  WegGraph weg_graph[2];
  hipEvent_t weg_ev = nullptr;
  long long weg_tok_version = 0;
  std::vector<int32_t> weg_tok_host;
  std::vector<long long> weg_sig;   // timestep, shapes, memory pointers and arena of the last evaluation (reuse_memory_side)
  int weg_launches = 0;
  // profiling
  bool prof = false;
  hipEvent_t pev[2] = {nullptr, nullptr};
  float prof_ms[CFD_PROF_NCLASS];
  int prof_n[CFD_PROF_NCLASS];
  int stop_stage = 0;  // test hook: leave enqueue_denoise after this tap point (0 = run everything)
  int run_iters = 0;                 // loop iterations of the open run (= length of the timestep table)
  hipStream_t own_stream = nullptr;  // non-blocking stream the captured loop iteration replays on
  // sampling run
  bool run_open = false;
  cfd_sample_args sargs;
  hipStream_t run_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  DBuf latents, coef, inoise, mem_own[CFD_NMEM];
  // Internal chunk order of a sampling run: chunk k of the caller's chunk-major batch lives at rows
  // chunk_pos[k] * B.  Chunks whose rows all use ONE shared copy of the largest memory (the unconditional audio
  // memory: 5 of the 7 guidance chunks, not adjacent in the reference's order) are moved next to each other, so
  // their attention against it is one un-batched product instead of one per contiguous run.  Every per-row result
  // is unchanged (rows are independent); the guidance combine reads chunk k at its position.  CFD_PERMUTE=0: off.
  int chunk_pos[8];
  DBuf perm_map[CFD_NMEM];
  bool permute = true;
  int run_pos = 0;
};
typedef cfd_handle_s Ctx;

// Saturation census (cfd_common.hpp) of this handle.  sat_begin zeroes the two counters in stream order at the entry of a call that
// counts; check_saturation reads them (the caller has waited for the stream) and clears them, so an error is reported by the call
// whose launches counted it and never leaks into the next call or another handle.
static inline int sat_begin(Ctx* c, hipStream_t st) {
  HIPCHK(hipMemsetAsync(c->sat.p, 0, 8, st));
  return CFD_OK;
}
static inline int check_saturation(Ctx* c, const char* what) {
  unsigned int n[2] = {0, 0};
  HIPCHK(hipMemcpy(n, c->sat.p, 8, hipMemcpyDeviceToHost));
  if (n[0] == 0 && n[1] == 0) return CFD_OK;
  HIPCHK(hipMemset(c->sat.p, 0, 8));
  if (n[CFD_SAT_MEM])
    return fail(CFD_E_RANGE, "%s: %u groups of values exceed +-65504, the range of the fp16 split-pair operands (weights, centred memories and their "
                             "folded key / value projections must stay inside it); rescale the conditioning input", what, n[CFD_SAT_MEM]);
  return fail(CFD_E_RANGE, "%s: %u groups of values of the sample / latents exceed +-65504, the range of the fp16 split-pair operands", what, n[CFD_SAT_IN]);
}

// A cfd_weg_eval that does not wait (loss_host == NULL) cannot read its own census.  The next entry point of the handle does, before it
// zeroes or reads the counters for its own launches: nothing is dropped and nothing is blamed on the wrong call.
static inline int settle_deferred_census(Ctx* c) {
  if (!c->census_pending) return CFD_OK;
  c->census_pending = false;
  HIPCHK(hipStreamSynchronize(c->own_stream));
  return check_saturation(c, "an earlier cfd_weg_eval (latents, memories / their projections)");
}

static inline const float* rawp(Ctx* c, const std::string& name) {
  auto it = c->raw.find(name);
  return it == c->raw.end() ? nullptr : it->second.as<float>();
}

// ---- profiling brackets ---------------------------------------------------------------------------
struct Bracket {
  Ctx* c;
  int cls;
  hipStream_t st;
  Bracket(Ctx* c_, int cls_, hipStream_t st_) : c(c_), cls(cls_), st(st_) {
    if (c->prof) (void)hipEventRecord(c->pev[0], st);
  }
  ~Bracket() {
    if (c->prof) {
      (void)hipEventRecord(c->pev[1], st);
      (void)hipEventSynchronize(c->pev[1]);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, c->pev[0], c->pev[1]);
      c->prof_ms[cls] += ms;
      c->prof_n[cls] += 1;
    }
  }
};

template <int MODE, class Epi>
static int run_gemm(Ctx* c, int cls, const GemmArgs& a, const Epi& e, int nb, int nz, hipStream_t st, int cfg = 0) {
  Bracket br(c, cls, st);
  hipError_t err = launch_gemm<MODE, Epi>(a, e, nb, nz, st, cfg);
  if (err != hipSuccess) return fail(CFD_E_HIP, "gemm launch failed: %s", hipGetErrorString(err));
  return CFD_OK;
}

template <int MODE, class Epi>
static int run_gemm_midsize(Ctx* c, int cls, const GemmArgs& a, const Epi& e, hipStream_t st) {
  Bracket br(c, cls, st);
  hipError_t err = launch_gemm_midsize<MODE, Epi>(a, e, st);
  if (err != hipSuccess) return fail(CFD_E_HIP, "gemm launch failed: %s", hipGetErrorString(err));
  return CFD_OK;
}

static inline GemmArgs gemm_args() {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.nslot = 1;
  return a;
}

#define LAUNCH(cls, kernel, grid, block, st, ...)                                              \
  do {                                                                                         \
    Bracket _br(c, cls, st);                                                                   \
    hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                               \
    hipError_t _e = hipGetLastError();                                                         \
    if (_e != hipSuccess) return fail(CFD_E_HIP, "%s launch failed: %s", #kernel, hipGetErrorString(_e)); \
  } while (0)

// ---- small kernels of the host code itself (templates: each unit instantiates what it launches) -----------
template <int CFD_KI = 0>
__global__ void scale_copy_kernel(const float* in, float* out, long long n, float s) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * s;
}
template <int CFD_KI = 0>
__global__ void d2f_kernel(const double* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
template <int CFD_KI = 0>
__global__ void f2d_kernel(const float* in, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}

// x[g][:] = x[0][:] for g = 1 .. G-1 (n4 float4 per replica): hands the shared pre-cross-attention state of layer 0
// to every guidance chunk
template <int CFD_KI = 0>
__global__ void replicate_rows_kernel(float4* x, long long n4, int G) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = x[i];
  for (int g = 1; g < G; ++g) x[(long long)g * n4 + i] = v;
}

template <int CFD_KI = 0>
__global__ void fill_f32_kernel(float* p, long long n, float v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}


template <int CFD_KI = 0>
__global__ void sched_step_kernel(const float* eps, const float* noise, float* x, size_t n, StepCoef c, int kind, int clip, float* x0_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float e = eps[i], xv = x[i];
  float x0 = (xv - c.sb * e) / c.sa;
  if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
  if (x0_out) x0_out[i] = x0;
  float prev = (kind == 0) ? c.c0 * x0 + c.cx * xv : c.c0 * x0 + c.cx * e;
  if (c.use_noise != 0.f) prev = prev + c.sigma * noise[i];
  x[i] = prev;
}
template <int CFD_KI = 0>
__global__ void add_noise_kernel(const float* x0, const float* noise, float* out, size_t n, float sa, float sb) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sa * x0[i] + sb * noise[i];
}

// ---- functions one translation unit calls in another ---------------------------------------------------
// cfd_core.hip
int to_sp(Ctx* c, const float* src, long long R, int K, DBuf& dst, long long dst_rows = -1);
// float32 [R][K] (row pitch ld_src floats) -> split pairs (row pitch ld_dst bytes); `sat`: the census counter of the values' class, or null
int enqueue_to_split(Ctx* c, int cls, hipStream_t st, const float* src, char* dst, long long R, int K, long long ld_src, long long ld_dst, unsigned int* sat);
// cfd_problem.hip: work lists, problem set-up, timestep tables, memory-side projections
int build_xattn_worklist(Ctx* c, const cfd_memory mem[CFD_NMEM]);
int build_xattn_layer0_lists(Ctx* c, const cfd_memory mem[CFD_NMEM]);
int setup_att_fused(Ctx* c);
int setup_problem(Ctx* c, int Be, int L, const cfd_memory mem[CFD_NMEM], float* const att[CFD_NMEM], int tmode, int T);
int build_time_tables(Ctx* c, const int32_t* trows_host, int T, hipStream_t st);
int enqueue_time_tables(Ctx* c, int T, hipStream_t st);
int prepare_static_memside(Ctx* c, hipStream_t st, int dynamic_mask, bool want_att, bool reuse = false);
int enqueue_memside(Ctx* c, hipStream_t st);
// cfd_forward.hip: the launches of one denoiser forward
int enqueue_rows(Ctx* c, hipStream_t st, int row0, int nrows);
int enqueue_rows_rt(Ctx* c, hipStream_t st, const RtSave* sv = nullptr);
int enqueue_denoise(Ctx* c, hipStream_t st);
int run_gemm_plain_f32(Ctx* c, int cls, const GemmArgs& a, const EpiF32& e, int nb, int nz, hipStream_t st);   // (the unit that holds the EpiF32 instances)
// cfd_sample.hip
int enqueue_philox_fill(float* out, int B, int per_utt, uint64_t seed, uint32_t step, uint32_t utt0, uint32_t stream_id, float scale, hipStream_t st);
// cfd_blocks.hip
int enqueue_linear_act(const float* x, long long n_rows, int K, const float* W, const float* b, int N, int act, float* out, hipStream_t st);
