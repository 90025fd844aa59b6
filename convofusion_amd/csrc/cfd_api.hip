// libcfdenoise: handle, weight folding, the denoiser forward pipeline, the hipGraph-captured sampler
// and the C ABI declared in include/cfdenoise.h.  gfx950 only; no CPU fallback anywhere.
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
#include "gemm_sp.hpp"
#include "rows.hpp"
#include "attn_fused.hpp"
#include "xattn_fused.hpp"
#include "rowtile.hpp"
#include "grad.hpp"

int g_cfd_naive_gemm = 0;

static thread_local char g_err[1024] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
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
                   &long_rows, &short_rows, &zero_mask, &b_tab, &b_sp, &bsq, &zeros512, &xa_wgs, &xa_segs, &xa_stamps, &xa0_wgs_a, &xa0_segs_a, &xa0_wgs_b, &xa0_segs_b, &xa_dedup, &xa_one_va, &xa_att_raw, &xa_att_mc, &xa_att_fin, &xa_att_desc, &d_step, &rt_vt, &rt_cur};
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
  bool hint_same_mem = false;   // cfd_forward_same_memories: the promise for the NEXT cfd_forward ...
  bool hint_now = false;        // ... taken (and cleared) at that call's very first line, before anything can fail: a call that returns early
                                // must not leave the promise standing for the call after it
  bool census_pending = false;  // a cfd_weg_eval without loss_host left its census unread (it does not wait): settled by the next entry point
  int rt_nfb2_tiles = 14;       // CFD_RT_NFB2_TILES=<token tiles>: from how many token tiles on the row-tile path's 512 x 512 residual products take two feature blocks per workgroup
  int step_rows = 1;            // CFD_STEP_ROWS=0: the tile kernels index the per-step tables with the device step counter themselves
  int att_fused = 1;            // CFD_ATT_FUSED=0: a forward that returns att_mats takes the three-launch cross-attention on the tile kernels (the fused
                                // kernel's ATT instance keeps the maps otherwise: xattn_fused.hpp, XaAtt)
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
static int sat_begin(Ctx* c, hipStream_t st) {
  HIPCHK(hipMemsetAsync(c->sat.p, 0, 8, st));
  return CFD_OK;
}
static int check_saturation(Ctx* c, const char* what) {
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
static int settle_deferred_census(Ctx* c) {
  if (!c->census_pending) return CFD_OK;
  c->census_pending = false;
  HIPCHK(hipStreamSynchronize(c->own_stream));
  return check_saturation(c, "an earlier cfd_weg_eval (latents, memories / their projections)");
}

static const float* rawp(Ctx* c, const std::string& name) {
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

static GemmArgs gemm_args() {
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

// ---- create / destroy -----------------------------------------------------------------------------
extern "C" const char* cfd_last_error(void) { return g_err; }

#ifndef CFD_SOURCE_HASH
#define CFD_SOURCE_HASH "unknown"
#endif
// (the marker lets the binding read the hash from the file before it maps it)
static const char g_source_hash[] = "cfd-src-hash:" CFD_SOURCE_HASH;
extern "C" const char* cfd_source_hash(void) { return g_source_hash + 13; }

extern "C" int cfd_create(const cfd_config* cfg, cfd_handle* out) {
  if (!cfg || !out) return fail(CFD_E_ARG, "null argument");
  if (cfg->latent_dim != CFD_LAT || cfg->text_encoded_dim != CFD_D || cfg->ff_size != CFD_FF ||
      cfg->num_heads != CFD_NHEAD)
    return fail(CFD_E_ARG, "unsupported dimensions: this build is specialised to latent 128, d_model 512, ff 1024, 4 heads "
                           "(configs/modules/denoiser.yaml)");
  if (cfg->num_layers < 1 || cfg->num_layers > CFD_MAX_LAYERS) return fail(CFD_E_ARG, "num_layers out of range");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(CFD_E_ARG, "device %d not present (%d devices)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(CFD_E_ARG, "libcfdenoise is built for gfx950 (MI355X) only; device reports %s", prop.gcnArchName);
  Ctx* c = new Ctx();
  c->cfg = *cfg;
  c->nl = cfg->num_layers;
  c->lw.resize(c->nl);
  const char* env = getenv("CFD_NAIVE_GEMM");
  g_cfd_naive_gemm = (env && atoi(env) != 0) ? 1 : 0;
  env = getenv("CFD_RUNS");
  c->use_runs = !(env && atoi(env) == 0);
  env = getenv("CFD_WEG_GRAPH");
  c->weg_graph_on = !(env && atoi(env) == 0);
  (void)hipEventCreateWithFlags(&c->weg_ev, hipEventDisableTiming);
  env = getenv("CFD_FUSED_XATTN");
  c->fused_xattn = !(env && atoi(env) == 0);
  env = getenv("CFD_FUSED_XATTN_MIN_WGS");
  if (env) c->fused_xattn_min_wgs = atoi(env);
  env = getenv("CFD_L0_DEDUP");
  if (env) c->l0_dedup = atoi(env) != 0;
  env = getenv("CFD_XA_OPERANDS");
  if (env) c->xa_operands = atoi(env) & 3;
  env = getenv("CFD_ONE_KEY");
  if (env) c->one_key = atoi(env) != 0;
  env = getenv("CFD_RT_NFB2_TILES");
  if (env) c->rt_nfb2_tiles = atoi(env);
  env = getenv("CFD_STEP_ROWS");
  if (env) c->step_rows = atoi(env) != 0;
  env = getenv("CFD_ATT_FUSED");
  if (env) c->att_fused = atoi(env) != 0;
  env = getenv("CFD_QKV_FUSED");
  if (env) c->qkv_fused = atoi(env);
  env = getenv("CFD_ROWTILE");
  c->rt_on = !(env && atoi(env) == 0);
  env = getenv("CFD_WEG_ROWTILE");
  c->weg_rt_on = !(env && atoi(env) == 0);
  env = getenv("CFD_ROWTILE_MAX_ROWS");
  if (env) c->rt_max_rows = atoll(env);
  env = getenv("CFD_HOIST_MEMSIDE");
  c->hoist_memside = !(env && atoi(env) == 0);
  env = getenv("CFD_PERMUTE");
  c->permute = !(env && atoi(env) == 0);
  env = getenv("CFD_SHARE0");
  c->share0 = !(env && atoi(env) == 0);
  for (Work& wk : c->wk) {
    if (wk.d_step.ensure(16) != CFD_OK) { delete c; return CFD_E_HIP; }
    if (hipMemset(wk.d_step.p, 0, 16) != hipSuccess) { delete c; return fail(CFD_E_HIP, "memset"); }
  }
  if (c->sat.ensure(8) != CFD_OK) { delete c; return CFD_E_HIP; }
  if (hipMemset(c->sat.p, 0, 8) != hipSuccess) { delete c; return fail(CFD_E_HIP, "memset"); }
  (void)hipEventCreate(&c->pev[0]);
  (void)hipEventCreate(&c->pev[1]);
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(CFD_E_HIP, "stream create"); }
  *out = c;
  return CFD_OK;
}

extern "C" void cfd_destroy(cfd_handle c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  (void)hipDeviceSynchronize();
  if (c->gexec) (void)hipGraphExecDestroy(c->gexec);
  if (c->graph) (void)hipGraphDestroy(c->graph);
  for (auto& wg : c->weg_graph) {
    if (wg.exec) (void)hipGraphExecDestroy(wg.exec);
    if (wg.graph) (void)hipGraphDestroy(wg.graph);
  }
  if (c->weg_ev) (void)hipEventDestroy(c->weg_ev);
  c->weg_io.release();
  c->weg_rt_ws.release();
  c->sat.release();
  for (auto& kv : c->raw) kv.second.release();
  DBuf* all[] = {&c->we_sp, &c->wp_sp, &c->we_all, &c->be_all, &c->tsin, &c->weg_ws, &c->weg_tok, &c->latents, &c->coef, &c->inoise};
  for (DBuf* b : all) b->release();
  c->wk[0].release();
  c->wk[1].release();
  for (int j = 0; j < CFD_NMEM; ++j) {
    c->wk_all_sp[j].release(); c->wv_all_sp[j].release(); c->mem_own[j].release(); c->perm_map[j].release();
  }
  for (auto& l : c->lw) {
    DBuf* lb[] = {&l.wqk_sp, &l.bqk, &l.wv_sp, &l.wo_sp, &l.bo2, &l.wtb1_sp, &l.wtb2_sp, &l.w1_sp, &l.w2_sp, &l.cross_bias};
    for (DBuf* b : lb) b->release();
  }
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->pev[0]) (void)hipEventDestroy(c->pev[0]);
  if (c->pev[1]) (void)hipEventDestroy(c->pev[1]);
  delete c;
}

// ---- weights ----------------------------------------------------------------------------------------
extern "C" int cfd_load_tensor(cfd_handle c, const char* name, const float* data, size_t numel, int is_device) {
  if (!c || !name || !data || numel == 0) return fail(CFD_E_ARG, "null/empty argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  DBuf& b = c->raw[name];
  CHK(b.ensure(numel * sizeof(float)));
  HIPCHK(hipMemcpy(b.p, data, numel * sizeof(float), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  c->raw_numel[name] = numel;
  c->finalized = false;
  c->weg_sig.clear();
  return CFD_OK;
}

static int need(Ctx* c, const std::string& name, size_t numel, const float** out, bool at_least = false) {
  auto it = c->raw_numel.find(name);
  if (it == c->raw_numel.end()) return fail(CFD_E_STATE, "missing tensor '%s' (state-dict key denoiser.%s)", name.c_str(), name.c_str());
  if (at_least ? (it->second < numel || it->second % CFD_D) : (it->second != numel))
    return fail(CFD_E_SHAPE, "tensor '%s' has %zu elements, expected %s%zu", name.c_str(), it->second, at_least ? ">= " : "", numel);
  *out = c->raw[name].as<float>();
  return CFD_OK;
}

static int to_sp(Ctx* c, const float* src, long long R, int K, DBuf& dst, long long dst_rows = -1) {
  if (dst_rows < 0) dst_rows = R;
  CHK(dst.ensure((size_t)dst_rows * K * 4));
  if (dst_rows > R) HIPCHK(hipMemset(dst.p, 0, (size_t)dst_rows * K * 4));
  const long long n = R * (K / 8);
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, src, dst.as<char>(), R, K,
                     (long long)K, (long long)K * 4, c->sat_mem());
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

__global__ void scale_copy_kernel(const float* in, float* out, long long n, float s) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * s;
}
__global__ void d2f_kernel(const double* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
__global__ void f2d_kernel(const float* in, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}

// x[g][:] = x[0][:] for g = 1 .. G-1 (n4 float4 per replica): hands the shared pre-cross-attention state of layer 0
// to every guidance chunk
__global__ void replicate_rows_kernel(float4* x, long long n4, int G) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = x[i];
  for (int g = 1; g < G; ++g) x[(long long)g * n4 + i] = v;
}

extern "C" int cfd_finalize_weights(cfd_handle c) {
  if (!c) return fail(CFD_E_ARG, "null handle");
  HIPCHK(hipSetDevice(c->cfg.device));
  const int D = CFD_D, nl = c->nl;
  const float *t0, *t1;
  CHK(sat_begin(c, 0));
  // embed / projection / tables
  CHK(need(c, "latent_embd.weight", (size_t)D * CFD_LAT, &t0));
  CHK(to_sp(c, t0, D, CFD_LAT, c->we_sp));
  CHK(need(c, "latent_embd.bias", D, &t0));
  CHK(need(c, "latent_proj.weight", (size_t)CFD_LAT * D, &t0));
  CHK(to_sp(c, t0, CFD_LAT, D, c->wp_sp));
  CHK(need(c, "latent_proj.bias", CFD_LAT, &t0));
  CHK(need(c, "time_embedding.linear_1.weight", (size_t)D * D, &t0));
  CHK(need(c, "time_embedding.linear_1.bias", D, &t0));
  CHK(need(c, "time_embedding.linear_2.weight", (size_t)D * D, &t0));
  CHK(need(c, "time_embedding.linear_2.bias", D, &t0));
  CHK(need(c, "bh_embedding.weight", 2 * D, &t0));
  CHK(need(c, "condition_embedding.weight", 5 * D, &t0));
  CHK(need(c, "decoder.norm.weight", D, &t0));
  CHK(need(c, "decoder.norm.bias", D, &t0));
  CHK(need(c, "query_pos.pe", D, &t0, true));
  c->qpe_rows = (int)(c->raw_numel["query_pos.pe"] / D);
  CHK(need(c, "mem_pos.pe", D, &t0, true));
  c->mpe_rows = (int)(c->raw_numel["mem_pos.pe"] / D);

  CHK(c->we_all.ensure((size_t)nl * 2 * 2 * D * D * 4));
  CHK(c->be_all.ensure((size_t)nl * 2 * 2 * D * 4));
  DBuf tmpf, tmpd1, tmpd2, vd1, vd2, accd;
  CHK(tmpf.ensure((size_t)3 * D * D * 4));
  CHK(tmpd1.ensure((size_t)D * D * 8));
  CHK(tmpd2.ensure((size_t)D * D * 8));
  CHK(vd1.ensure(D * 8));
  CHK(vd2.ensure(D * 8));
  CHK(accd.ensure(D * 8));
  const long long kfeat = (long long)nl * D;
  DBuf wk_f[CFD_NMEM], wv_f[CFD_NMEM];
  for (int j = 0; j < CFD_NMEM; ++j) {
    CHK(wk_f[j].ensure((size_t)(kfeat + 32) * D * 4));
    HIPCHK(hipMemset(wk_f[j].p, 0, (size_t)(kfeat + 32) * D * 4));
    CHK(wv_f[j].ensure((size_t)kfeat * D * 4));
  }
  const dim3 blk(256);
  auto grid1 = [](long long n) { return dim3((unsigned)((n + 255) / 256)); };

  for (int l = 0; l < nl; ++l) {
    LayerW& w = c->lw[l];
    const std::string p = "decoder.layers." + std::to_string(l) + ".";
    const float *ipw, *ipb, *ow, *ob;
    // -- self attention: q rows scaled by sqrt(1/head_dim) (F.multi_head_attention_forward "q_scaled")
    CHK(need(c, p + "self_attn.in_proj_weight", (size_t)3 * D * D, &ipw));
    CHK(need(c, p + "self_attn.in_proj_bias", 3 * D, &ipb));
    CHK(need(c, p + "self_attn.out_proj.weight", (size_t)D * D, &ow));
    CHK(need(c, p + "self_attn.out_proj.bias", D, &ob));
    const float qs = (float)std::sqrt(1.0 / (double)CFD_HD);
    float* tf = tmpf.as<float>();
    hipLaunchKernelGGL(scale_copy_kernel, grid1((long long)D * D), blk, 0, 0, ipw, tf, (long long)D * D, qs);
    hipLaunchKernelGGL(scale_copy_kernel, grid1((long long)D * D), blk, 0, 0, ipw + (size_t)D * D, tf + (size_t)D * D,
                       (long long)D * D, 1.0f);
    CHK(to_sp(c, tf, 2 * D, D, w.wqk_sp));
    CHK(w.bqk.ensure(2 * D * 4));
    hipLaunchKernelGGL(scale_copy_kernel, grid1(D), blk, 0, 0, ipb, w.bqk.as<float>(), (long long)D, qs);
    hipLaunchKernelGGL(scale_copy_kernel, grid1(D), blk, 0, 0, ipb + D, w.bqk.as<float>() + D, (long long)D, 1.0f);
    CHK(to_sp(c, ipw + (size_t)2 * D * D, D, D, w.wv_sp));
    CHK(to_sp(c, ow, D, D, w.wo_sp));
    // softmax rows sum to one, so the value bias passes straight through: bo' = bo + Wo bv
    CHK(w.bo2.ensure(D * 4));
    hipLaunchKernelGGL((fold_mv_kernel<float, float, float>), grid1(D), blk, 0, 0, ow, (long long)D, 1LL, ipb + 2 * D,
                       (const double*)nullptr, ob, w.bo2.as<float>(), D, D, 1.0, (const float*)nullptr);
    // -- time blocks
    for (int tb = 0; tb < 2; ++tb) {
      const std::string q = p + (tb == 0 ? "time_block1." : "time_block2.");
      CHK(need(c, q + "emb_layers.1.weight", (size_t)2 * D * D, &t0));
      HIPCHK(hipMemcpy(c->we_all.as<float>() + ((size_t)(2 * l + tb) * 2 * D) * D, t0, (size_t)2 * D * D * 4, hipMemcpyDeviceToDevice));
      CHK(need(c, q + "emb_layers.1.bias", 2 * D, &t0));
      HIPCHK(hipMemcpy(c->be_all.as<float>() + (size_t)(2 * l + tb) * 2 * D, t0, (size_t)2 * D * 4, hipMemcpyDeviceToDevice));
      CHK(need(c, q + "out_layers.2.weight", (size_t)D * D, &t0));
      CHK(to_sp(c, t0, D, D, tb == 0 ? w.wtb1_sp : w.wtb2_sp));
      CHK(need(c, q + "norm.weight", D, tb == 0 ? &w.tb1g : &w.tb2g));
      CHK(need(c, q + "norm.bias", D, tb == 0 ? &w.tb1b : &w.tb2b));
      CHK(need(c, q + "out_layers.2.bias", D, tb == 0 ? &w.btb1 : &w.btb2));
    }
    CHK(need(c, p + "norm1.weight", D, &w.ln1g)); CHK(need(c, p + "norm1.bias", D, &w.ln1b));
    CHK(need(c, p + "norm2.weight", D, &w.ln2g)); CHK(need(c, p + "norm2.bias", D, &w.ln2b));
    CHK(need(c, p + "norm3.weight", D, &w.ln3g)); CHK(need(c, p + "norm3.bias", D, &w.ln3b));
    CHK(need(c, p + "linear1.weight", (size_t)CFD_FF * D, &t0)); CHK(to_sp(c, t0, CFD_FF, D, w.w1_sp));
    CHK(need(c, p + "linear1.bias", CFD_FF, &w.b1));
    CHK(need(c, p + "linear2.weight", (size_t)D * CFD_FF, &t0)); CHK(to_sp(c, t0, D, CFD_FF, w.w2_sp));
    CHK(need(c, p + "linear2.bias", D, &w.b2));
    // -- five single-head cross attentions + att_fuser, folded onto the memory side (DESIGN.md "Folding")
    const float *fw, *fb;
    CHK(need(c, p + "att_fuser.weight", (size_t)D * 5 * D, &fw));
    CHK(need(c, p + "att_fuser.bias", D, &fb));
    hipLaunchKernelGGL(f2d_kernel, grid1(D), blk, 0, 0, fb, accd.as<double>(), D);
    const double cs = std::sqrt(1.0 / (double)D);  // one head of width 512
    for (int j = 0; j < CFD_NMEM; ++j) {
      const std::string a = p + "multihead_attn_" + MEM_NAMES[j];
      const float *gam, *bet;
      CHK(need(c, a + ".in_proj_weight", (size_t)3 * D * D, &ipw));
      CHK(need(c, a + ".in_proj_bias", 3 * D, &ipb));
      CHK(need(c, a + ".out_proj.weight", (size_t)D * D, &ow));
      CHK(need(c, a + ".out_proj.bias", D, &ob));
      CHK(need(c, p + MEM_NAMES[j] + "_norm.weight", D, &gam));
      CHK(need(c, p + MEM_NAMES[j] + "_norm.bias", D, &bet));
      const float *Wq = ipw, *Wk = ipw + (size_t)D * D, *Wv = ipw + (size_t)2 * D * D;
      const float *bq = ipb, *bv = ipb + 2 * D;
      // key side:  A[o][i] = cs * gamma[i] * sum_r Wq[r][o] Wk[r][i]      (scores = y . (A n))
      hipLaunchKernelGGL((fold_mm_kernel<float, float, float>), grid1((long long)D * D), blk, 0, 0, Wq, 1LL, (long long)D, Wk,
                         (long long)D, 1LL, wk_f[j].as<float>() + (size_t)l * D * D, (long long)D, D, D, D, cs, gam);
      //            c[i]  = cs * gamma[i] * sum_r Wk[r][i] bq[r]          (key-dependent part of the q-bias term)
      hipLaunchKernelGGL((fold_mv_kernel<float, float, float>), grid1(D), blk, 0, 0, Wk, 1LL, (long long)D, bq,
                         (const double*)nullptr, (const float*)nullptr, wk_f[j].as<float>() + (size_t)(kfeat + l) * D, D, D, cs, gam);
      // value side: VV = Wf_j Wo Wv diag(gamma)
      hipLaunchKernelGGL((fold_mm_kernel<float, float, double>), grid1((long long)D * D), blk, 0, 0, ow, (long long)D, 1LL, Wv,
                         (long long)D, 1LL, tmpd1.as<double>(), (long long)D, D, D, D, 1.0, (const float*)nullptr);
      hipLaunchKernelGGL((fold_mm_kernel<float, double, float>), grid1((long long)D * D), blk, 0, 0, fw + (size_t)j * D,
                         (long long)5 * D, 1LL, tmpd1.as<double>(), (long long)D, 1LL, wv_f[j].as<float>() + (size_t)l * D * D,
                         (long long)D, D, D, D, 1.0, gam);
      // constant: acc += Wf_j ( Wo (Wv beta + bv) + bo )
      hipLaunchKernelGGL((fold_mv_kernel<float, float, double>), grid1(D), blk, 0, 0, Wv, (long long)D, 1LL, bet,
                         (const double*)nullptr, bv, vd1.as<double>(), D, D, 1.0, (const float*)nullptr);
      hipLaunchKernelGGL((fold_mv_kernel<float, double, double>), grid1(D), blk, 0, 0, ow, (long long)D, 1LL, vd1.as<double>(),
                         (const double*)nullptr, ob, vd2.as<double>(), D, D, 1.0, (const float*)nullptr);
      hipLaunchKernelGGL((fold_mv_kernel<float, double, double>), grid1(D), blk, 0, 0, fw + (size_t)j * D, (long long)5 * D, 1LL,
                         vd2.as<double>(), (const double*)accd.as<double>(), (const float*)nullptr, vd1.as<double>(), D, D, 1.0,
                         (const float*)nullptr);
      HIPCHK(hipMemcpy(accd.p, vd1.p, D * 8, hipMemcpyDeviceToDevice));
    }
    CHK(w.cross_bias.ensure(D * 4));
    hipLaunchKernelGGL(d2f_kernel, grid1(D), blk, 0, 0, accd.as<double>(), w.cross_bias.as<float>(), D);
    HIPCHK(hipGetLastError());
  }
  for (int j = 0; j < CFD_NMEM; ++j) {
    CHK(to_sp(c, wk_f[j].as<float>(), kfeat + 32, D, c->wk_all_sp[j]));
    CHK(to_sp(c, wv_f[j].as<float>(), kfeat, D, c->wv_all_sp[j]));
  }
  HIPCHK(hipDeviceSynchronize());
  CHK(check_saturation(c, "cfd_finalize_weights (a weight or a folded weight product)"));
  for (int j = 0; j < CFD_NMEM; ++j) { wk_f[j].release(); wv_f[j].release(); }
  tmpf.release(); tmpd1.release(); tmpd2.release(); vd1.release(); vd2.release(); accd.release();
  c->finalized = true;
  ++c->wver;
  return CFD_OK;
}

extern "C" int cfd_set_timestep_table(cfd_handle c, const float* rows, int n_rows) {
  if (!c || !rows || n_rows < 1) return fail(CFD_E_ARG, "bad timestep table");
  HIPCHK(hipSetDevice(c->cfg.device));
  CHK(c->tsin.ensure((size_t)n_rows * CFD_D * 4));
  HIPCHK(hipMemcpy(c->tsin.p, rows, (size_t)n_rows * CFD_D * 4, hipMemcpyHostToDevice));
  c->tsin_rows = n_rows;
  ++c->wver;              // (the timestep-only tables cached in the workspaces were built from the previous sinusoid rows: build_time_tables)
  return CFD_OK;
}

// ---- problem setup ----------------------------------------------------------------------------------
// Work list of the fused cross-attention kernel (xattn_fused.hpp).  A wave owns one tile of 16 queries of one batch row;
// the four waves of a workgroup share every K / V^T tile that passes through LDS, so
//  * rows are grouped by the instance of the LARGEST memory they attend to (the guidance batch: 5 of 7 chunks share the
//    unconditional audio memory; the other two chunks of an utterance share its own): a group's query tiles are dealt to
//    workgroups four at a time, and the long key stream is read once per workgroup whatever the rows' other memories are;
//  * per workgroup and memory, one segment per DISTINCT instance among its waves' rows (wave mask says who takes part);
//  * memories longer than one 32-key tile come first (online softmax; a flush of the accumulator between two of them);
//  * workgroups that read the same instance are placed on one XCD (block id % 8) next to each other so the stream is
//    fetched into that XCD's L2 once; big groups are dealt over all XCDs.
struct XaRow {      // one row of a work list
  int xrow;           // row of the residual stream (queries read from it; updated unless the list stores to xa_dedup)
  int inst[CFD_NMEM]; // memory instance per memory
  int aux;            // row of xa_dedup this row's tiles store to / add (-1: none)
  int one;            // instance of the one-key memory (Problem::xa_one), or -1
  int att;            // row of the attention-map blocks this row's tiles store to (Problem::att_fused), or -1
};

static void make_xattn_worklist(const Problem& p, const std::vector<XaRow>& rows, int mem_mask, std::vector<XaWg>& wgs, std::vector<XaSeg>& segs,
                                size_t& n_active, bool& has_flush) {
  const int L = p.L, nqt = (L + 15) / 16;
  wgs.clear(); segs.clear(); n_active = 0;
  // memory order: long (online) memories first, longest first; then the single-tile ones
  int order[CFD_NMEM], n_mem = 0, n_online = 0;
  for (int j = 0; j < CFD_NMEM; ++j)
    if ((mem_mask >> j) & 1) order[n_mem++] = j;
  if (n_mem == 0 || rows.empty()) return;
  std::stable_sort(order, order + n_mem, [&](int a, int b) { return p.Sp[a] > p.Sp[b]; });
  for (int oi = 0; oi < n_mem; ++oi) n_online += p.Sp[order[oi]] > XA_KEYS;
  const int jg = order[0];
  // groups of rows by instance of memory jg, in order of first appearance
  std::vector<int> inst_group(p.U[jg], -1);
  std::vector<std::vector<int>> groups;
  for (size_t r = 0; r < rows.size(); ++r) {
    int& g = inst_group[rows[r].inst[jg]];
    if (g < 0) { g = (int)groups.size(); groups.emplace_back(); }
    groups[g].push_back((int)r);
  }
  // Query tiles per workgroup.  A workgroup has room for XA_TILES = 4 (its K / V^T tiles then serve 64 queries), but a short list must
  // first of all FILL THE CHIP: at the product shape (L = 16: one tile per batch row) 32 utterances are 224 tiles = 56 workgroups for 256
  // CUs, each walking 14 segment steps because its four rows use different instances of the short memories (a pass per instance).  With
  // one tile per workgroup (six of the eight waves only request tile pieces) that is 224 workgroups of 9 steps: half the launch time.
  // Halve while the list stays at or below 128 workgroups.
  int tpw = XA_TILES;
  while (tpw > 1 && (rows.size() * (size_t)nqt + tpw - 1) / tpw <= 128) tpw /= 2;
  std::vector<std::vector<XaWg>> group_wgs(groups.size());
  for (size_t g = 0; g < groups.size(); ++g) {
    std::vector<std::pair<int, int>> tiles;   // (row of `rows`, first query)
    for (int r : groups[g])
      for (int t = 0; t < nqt; ++t) tiles.emplace_back(r, t * 16);
    for (size_t t0 = 0; t0 < tiles.size(); t0 += tpw) {
      XaWg w;
      memset(&w, 0, sizeof(w));
      int vr[XA_TILES];
      for (int k = 0; k < XA_TILES; ++k) {
        const bool on = k < tpw && t0 + k < tiles.size();
        vr[k] = on ? tiles[t0 + k].first : -1;
        w.row[k] = on ? rows[vr[k]].xrow : -1;
        w.aux[k] = on ? rows[vr[k]].aux : -1;
        w.one[k] = on ? rows[vr[k]].one : -1;
        w.att[k] = on ? rows[vr[k]].att : -1;
        w.q0[k] = on ? tiles[t0 + k].second : 0;
      }
      w.seg0 = (int)segs.size();
      int online_seen = 0;
      for (int oi = 0; oi < n_mem; ++oi) {
        const int j = order[oi];
        const bool online = p.Sp[j] > XA_KEYS;
        online_seen += online;
        int done = 0;
        size_t first_seg = segs.size();
        for (int k = 0; k < XA_TILES; ++k) {
          if (vr[k] < 0 || (done >> k) & 1) continue;
          XaSeg sg;
          sg.j = j; sg.u = rows[vr[k]].inst[j]; sg.wmask = 0; sg.flags = (online ? XA_ONLINE : 0) | (((p.xa_f16_mask >> j) & 1) ? XA_F16 : 0);
          for (int k2 = k; k2 < XA_TILES; ++k2)
            if (vr[k2] >= 0 && rows[vr[k2]].inst[j] == sg.u) sg.wmask |= 1 << k2;
          done |= sg.wmask;
          segs.push_back(sg);
        }
        // one accumulator: a finished online memory is flushed to x before the next online memory starts
        if (online && online_seen < n_online && segs.size() > first_seg) { segs.back().flags |= XA_FLUSH; has_flush = true; }
      }
      w.nseg = (int)segs.size() - w.seg0;
      w.n16 = 0;            // the segments with single-fp16 tiles: a prefix of the list (memories in descending length, XA_F16 <=> long enough)
      while (w.n16 < w.nseg && (segs[w.seg0 + w.n16].flags & XA_F16)) ++w.n16;
      for (int k = w.n16; k < w.nseg; ++k) segs[w.seg0 + k].flags &= ~XA_F16;   // (never: the order guarantees it; a flag behind the prefix would be read in the wrong format)
      group_wgs[g].push_back(w);
    }
  }
  // XCD placement: queue x holds the workgroups with block id % 8 == x, in dispatch order
  std::vector<std::vector<XaWg>> queue(8);
  std::vector<size_t> gorder(groups.size());
  for (size_t g = 0; g < groups.size(); ++g) gorder[g] = g;
  std::stable_sort(gorder.begin(), gorder.end(), [&](size_t a, size_t b) { return group_wgs[a].size() > group_wgs[b].size(); });
  auto shortest = [&]() { int q = 0; for (int x = 1; x < 8; ++x) if (queue[x].size() < queue[q].size()) q = x; return q; };
  for (size_t g : gorder) {
    if (group_wgs[g].size() > 32) { for (const XaWg& w : group_wgs[g]) queue[shortest()].push_back(w); }
    else { const int q = shortest(); for (const XaWg& w : group_wgs[g]) queue[q].push_back(w); }
  }
  size_t qlen = 0;
  for (int x = 0; x < 8; ++x) qlen = std::max(qlen, queue[x].size());
  XaWg idle;
  memset(&idle, 0, sizeof(idle));
  for (int k = 0; k < XA_TILES; ++k) { idle.row[k] = -1; idle.aux[k] = -1; idle.one[k] = -1; idle.att[k] = -1; }
  wgs.assign(qlen * 8, idle);
  for (int x = 0; x < 8; ++x) {
    for (size_t i = 0; i < queue[x].size(); ++i) wgs[i * 8 + x] = queue[x][i];
    n_active += queue[x].size();
  }
}

static int read_row_maps(Ctx* c, const cfd_memory mem[CFD_NMEM], std::vector<std::vector<int>>& hm) {
  const Problem& p = c->w->pb;
  hm.assign(CFD_NMEM, std::vector<int>(p.Be));
  for (int j = 0; j < CFD_NMEM; ++j) {
    if (mem[j].row_map) HIPCHK(hipMemcpy(hm[j].data(), mem[j].row_map, (size_t)p.Be * 4, hipMemcpyDeviceToHost));
    else for (int b = 0; b < p.Be; ++b) hm[j][b] = b;
    for (int b = 0; b < p.Be; ++b)
      if (hm[j][b] < 0 || hm[j][b] >= p.U[j]) return fail(CFD_E_ARG, "memory %s: row_map[%d] = %d outside [0, %d)", MEM_NAMES[j], b, hm[j][b], p.U[j]);
  }
  return CFD_OK;
}

static int upload_worklist(DBuf& dw, DBuf& ds, const std::vector<XaWg>& wgs, const std::vector<XaSeg>& segs) {
  CHK(dw.ensure(wgs.size() * sizeof(XaWg)));
  CHK(ds.ensure(segs.size() * sizeof(XaSeg)));
  HIPCHK(hipMemcpy(dw.p, wgs.data(), wgs.size() * sizeof(XaWg), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds.p, segs.data(), segs.size() * sizeof(XaSeg), hipMemcpyHostToDevice));
  return CFD_OK;
}

static int build_xattn_worklist(Ctx* c, const cfd_memory mem[CFD_NMEM]) {
  Problem& p = c->w->pb;
  p.xa_nwg = 0; p.xa0_nwg_a = 0; p.xa0_nwg_b = 0; p.xa_flush = false;
  if (!c->fused_xattn) return CFD_OK;
  std::vector<std::vector<int>> hm;
  CHK(read_row_maps(c, mem, hm));
  // A memory of ONE key whose key no instance masks is added as a vector in the kernel's flush instead of walking a 32-key tile step
  // (xattn_fused.hpp, XAttnArgs::one_j): it gets no segments.  (With the key masked the reference's softmax is NaN: that stays a segment.)
  p.xa_one = -1;
  if (c->one_key && c->hoist_memside && p.tmode == 0) {
    for (int j = CFD_NMEM - 1; j >= 0 && p.xa_one < 0; --j) {
      if (p.S[j] != 1) continue;
      bool alive = true;
      if (mem[j].key_padding_mask) {
        std::vector<uint8_t> mk(p.U[j]);
        HIPCHK(hipMemcpy(mk.data(), mem[j].key_padding_mask, (size_t)p.U[j], hipMemcpyDeviceToHost));
        for (uint8_t v : mk) alive = alive && v == 0;
      }
      if (alive) p.xa_one = j;
    }
  }
  std::vector<XaRow> rows(p.Be);
  for (int b = 0; b < p.Be; ++b) {
    rows[b].xrow = b; rows[b].aux = -1; rows[b].one = p.xa_one >= 0 ? hm[p.xa_one][b] : -1;
    rows[b].att = (p.att_fused && b >= p.att_b0 && b < p.att_b0 + p.att_nb) ? b - p.att_b0 : -1;
    for (int j = 0; j < CFD_NMEM; ++j) rows[b].inst[j] = hm[j][b];
  }
  const int all_mems = ((1 << CFD_NMEM) - 1) & ~(p.xa_one >= 0 ? 1 << p.xa_one : 0);
  std::vector<XaWg> wgs;
  std::vector<XaSeg> segs;
  size_t n_active = 0;
  make_xattn_worklist(p, rows, all_mems, wgs, segs, n_active, p.xa_flush);
  if (wgs.empty() || segs.empty()) return CFD_OK;
  // A handful of workgroups cannot hide their serial walk over the key tiles (3 barriers and a fill round trip per 32 keys
  // with nothing else on the chip).  Round-2 measurements at the product shape, 1000 steps, since the memory-side projections
  // left the loop (the three-launch path still makes them per step): one utterance (2 workgroups) 1.342 s fused against
  // 1.318 s three-launch, four utterances (7 workgroups) 1.370 against 1.404, 16 (28 workgroups, one shard of the product-shape
  // benchmark) 470 against 465 steps/s.  Below 6 workgroups the three-launch path stays (CFD_FUSED_XATTN_MIN_WGS overrides;
  // the test suite sets 0 and runs its small cases through the fused kernel).
  // (counted in workgroups of four query tiles, as measured -- the list itself may deal fewer tiles per workgroup: make_xattn_worklist)
  if ((int)(((size_t)p.Be * ((p.L + 15) / 16) + XA_TILES - 1) / XA_TILES) < c->fused_xattn_min_wgs) return CFD_OK;
  (void)n_active;
  CHK(upload_worklist(c->w->xa_wgs, c->w->xa_segs, wgs, segs));
  p.xa_nwg = (int)wgs.size();
  return CFD_OK;
}

// Layer 0 of the sampling loop: the G guidance chunks of an utterance enter the first cross-attention with the SAME state (the
// replica-independent head, Problem::share_B), so the attention of that state against one memory instance is the same in every
// chunk that uses the instance.  For the longest memory (audio: 1 500 of the 1 573 keys at the benchmark shape) the 7 chunks of an
// utterance use 2 instances -- its own and the shared unconditional one -- so 2 evaluations replace 7:
//   list A  one row per distinct (utterance, instance of the longest memory): that memory only, result (incl. its rank-one
//           timestep term, without the bias) STORED to xa_dedup[aux]
//   list B  every row, the other memories, and x += ... + xa_dedup[aux(row)]
// Exact in real arithmetic; the summation order over the memories differs from the one-launch form (CFD_L0_DEDUP=0), so the G
// chunks of an utterance still leave layer 0 bit-identical where their memories are identical, but a run differs from the
// one-launch form by rounding (measured 2e-5 relative on final latents).
static int build_xattn_layer0_lists(Ctx* c, const cfd_memory mem[CFD_NMEM]) {
  Problem& p = c->w->pb;
  p.xa0_nwg_a = p.xa0_nwg_b = 0;
  if (!c->l0_dedup || p.xa_nwg == 0 || p.share_B <= 0 || p.Be % p.share_B || p.Be == p.share_B) return CFD_OK;
  const int B = p.share_B, G = p.Be / B;
  std::vector<std::vector<int>> hm;
  CHK(read_row_maps(c, mem, hm));
  int jg = 0;
  for (int j = 1; j < CFD_NMEM; ++j)
    if (p.Sp[j] > p.Sp[jg]) jg = j;
  if (p.Sp[jg] < 256) return CFD_OK;   // nothing worth a second launch
  std::vector<XaRow> ra, rb(p.Be);
  std::vector<int> aux_of(p.Be, -1);
  for (int b = 0; b < B; ++b) {
    std::vector<std::pair<int, int>> seen;   // (instance, index in ra)
    for (int g = 0; g < G; ++g) {
      const int row = g * B + b, u = hm[jg][row];
      int idx = -1;
      for (auto& sn : seen)
        if (sn.first == u) idx = sn.second;
      if (idx < 0) {
        idx = (int)ra.size();
        XaRow r;
        r.xrow = row; r.aux = idx; r.one = -1;
        // (every chunk of the utterance enters layer 0 with the same state: the full-conditioning chunk's map against this instance is this row's)
        r.att = (p.att_fused && p.att_b0 + b < p.Be && hm[jg][p.att_b0 + b] == u) ? b : -1;
        for (int j = 0; j < CFD_NMEM; ++j) r.inst[j] = hm[j][row];
        ra.push_back(r);
        seen.emplace_back(u, idx);
      }
      aux_of[row] = idx;
    }
  }
  if (ra.size() * 2 > (size_t)p.Be) return CFD_OK;   // too little repetition
  for (int b = 0; b < p.Be; ++b) {
    rb[b].xrow = b; rb[b].aux = aux_of[b]; rb[b].one = p.xa_one >= 0 ? hm[p.xa_one][b] : -1;
    rb[b].att = (p.att_fused && b >= p.att_b0 && b < p.att_b0 + p.att_nb) ? b - p.att_b0 : -1;
    for (int j = 0; j < CFD_NMEM; ++j) rb[b].inst[j] = hm[j][b];
  }
  std::vector<XaWg> wa, wb;
  std::vector<XaSeg> sa, sb;
  size_t na = 0, nb = 0;
  make_xattn_worklist(p, ra, 1 << jg, wa, sa, na, p.xa_flush);
  make_xattn_worklist(p, rb, ((1 << CFD_NMEM) - 1) & ~(1 << jg) & ~(p.xa_one >= 0 ? 1 << p.xa_one : 0), wb, sb, nb, p.xa_flush);
  if (wa.empty() || wb.empty()) return CFD_OK;
  CHK(upload_worklist(c->w->xa0_wgs_a, c->w->xa0_segs_a, wa, sa));
  CHK(upload_worklist(c->w->xa0_wgs_b, c->w->xa0_segs_b, wb, sb));
  CHK(c->w->xa_dedup.ensure(ra.size() * (size_t)p.L * CFD_D * 4));
  p.xa0_nwg_a = (int)wa.size();
  p.xa0_nwg_b = (int)wb.size();
  return CFD_OK;
}

// Buffers and per-layer descriptors of the attention maps the fused cross-attention kernel keeps (Problem::att_fused; rows att_nb, set by the caller)
static int setup_att_fused(Ctx* c) {
  Problem& pb = c->w->pb;
  XaAtt d;
  memset(&d, 0, sizeof(d));
  d.nb = pb.att_nb;
  for (int j = 0; j < CFD_NMEM; ++j) { d.off[j] = d.sp_tot; d.t0[j] = d.nt; d.sp_tot += pb.Sp[j]; d.nt += pb.Sp[j] / XA_KEYS; }
  const size_t rows = (size_t)pb.att_nb * pb.L;
  CHK(c->w->xa_att_raw.ensure(c->nl * rows * d.sp_tot * 4));
  CHK(c->w->xa_att_mc.ensure(c->nl * rows * d.nt * 4));
  CHK(c->w->xa_att_fin.ensure(c->nl * rows * CFD_NMEM * 2 * 4));
  CHK(c->w->xa_att_desc.ensure(c->nl * sizeof(XaAtt)));
  std::vector<XaAtt> desc(c->nl, d);
  for (int l = 0; l < c->nl; ++l) {
    desc[l].raw = c->w->xa_att_raw.as<float>() + (size_t)l * rows * d.sp_tot;
    desc[l].mc = c->w->xa_att_mc.as<float>() + (size_t)l * rows * d.nt;
    desc[l].fin = c->w->xa_att_fin.as<float>() + (size_t)l * rows * CFD_NMEM * 2;
  }
  HIPCHK(hipMemcpy(c->w->xa_att_desc.p, desc.data(), c->nl * sizeof(XaAtt), hipMemcpyHostToDevice));
  return CFD_OK;
}

static int setup_problem(Ctx* c, int Be, int L, const cfd_memory mem[CFD_NMEM], float* const att[CFD_NMEM], int tmode, int T) {
  // (whatever this call is and however it ends, it may overwrite the memory-side buffers: the previous forward's projections are current
  //  only if cfd_forward says so again at its end)
  const bool mem_was_valid = c->w->fwd_mem_valid;
  c->w->fwd_mem_valid = false;
  if (!c->finalized) return fail(CFD_E_STATE, "weights not finalized");
  if (c->tsin_rows < 1) return fail(CFD_E_STATE, "timestep table not set");
  if (Be < 1 || L < 2) return fail(CFD_E_ARG, "bad batch / length");
  if (L % 2) return fail(CFD_E_SHAPE, "latent length %d is odd (reference: broadcasting error at position_encoding.py:160-161)", L);
  if (L / 2 > c->qpe_rows) return fail(CFD_E_SHAPE, "L/2 = %d exceeds the query PE buffer (%d rows)", L / 2, c->qpe_rows);
  if ((size_t)Be * 4 > c->w->iota.bytes) {   // identity row map (memories passed without de-duplication)
    CHK(c->w->iota.ensure((size_t)Be * 4));
    std::vector<int> id(Be);
    for (int i = 0; i < Be; ++i) id[i] = i;
    HIPCHK(hipMemcpy(c->w->iota.p, id.data(), (size_t)Be * 4, hipMemcpyHostToDevice));
  }
  Problem& p = c->w->pb;
  bool prev_same = mem_was_valid && c->w->fwd_wver == (unsigned long long)c->wver && c->w->fwd_Be == Be && tmode == 0;
  for (int j = 0; j < CFD_NMEM && prev_same; ++j)
    prev_same = c->w->fwd_U[j] == mem[j].U && c->w->fwd_S[j] == mem[j].S && c->w->fwd_mask[j] == (mem[j].key_padding_mask != nullptr) &&
                c->w->fwd_map[j] == (mem[j].row_map != nullptr);
  p.prev_same = prev_same;
  // ... and with the caller's promise that they ARE the same memories (cfd_forward_same_memories covers the row maps and masks), the work
  // lists and instance tables made from them -- several device-to-host reads per call -- are kept as well
  bool any_att_in = false;
  for (int j = 0; j < CFD_NMEM; ++j) any_att_in = any_att_in || (att && att[j]);
  const bool keep_lists = prev_same && c->hint_now && c->w->fwd_L == L && c->w->fwd_att == any_att_in;
  p.Be = Be; p.L = L; p.Lp = (L + 31) / 32 * 32; p.M = (long long)Be * L; p.tmode = tmode; p.T = T;
  p.share_B = 0;
  if (p.Lp > SM_MAX_CHUNKS * 512) return fail(CFD_E_SHAPE, "L = %d exceeds the in-register softmax limit (%d)", L, SM_MAX_CHUNKS * 512);
  int off = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {
    const cfd_memory& m = mem[j];
    if (!m.data || m.U < 1 || m.S < 1) return fail(CFD_E_ARG, "memory %s: null/empty", MEM_NAMES[j]);
    if (!m.row_map && m.U != Be) return fail(CFD_E_ARG, "memory %s: U = %d != Be = %d without a row_map", MEM_NAMES[j], m.U, Be);
    if (tmode == 1 && m.row_map) return fail(CFD_E_ARG, "per-row timesteps need identity memory maps");
    if (m.S > c->mpe_rows)
      return fail(CFD_E_SHAPE, "memory %s has %d tokens, memory PE buffer has %d rows (reference: size mismatch at position_encoding.py:135)",
                  MEM_NAMES[j], m.S, c->mpe_rows);
    p.U[j] = m.U; p.S[j] = m.S; p.Sp[j] = (m.S + 31) / 32 * 32; p.off[j] = off; off += p.Sp[j];
    if (p.Sp[j] > SM_MAX_CHUNKS * 512) return fail(CFD_E_SHAPE, "memory %s: %d keys exceed the in-register softmax limit", MEM_NAMES[j], m.S);
    p.mem[j] = m.data; p.map[j] = m.row_map ? m.row_map : c->w->iota.as<int>(); p.mask[j] = m.key_padding_mask;
    p.att[j] = att ? att[j] : nullptr;
    p.att_slot[j] = 0;
    p.att_b0 = p.att_nb = 0;      // (a sampling run with an attention ring sets them after this call)
    p.att_fused = false;
    p.xa_opf = 0; p.xa_f16_mask = 0;
  }
  p.Sp_tot = off;
  // The run's operand policy (cfd_sample_begin): which memories are long enough for single-fp16 tiles.  The work lists flag their segments
  // (XA_F16); whether the run really takes the single-fp16 kernel instance is decided when everything else about it is known
  // (cfd_sample_begin, prepare_static_memside) -- the pair instance ignores the flag.
  p.xa_opf = c->want_opf;
  if (p.xa_opf)
    for (int j = 0; j < CFD_NMEM; ++j)
      if (p.Sp[j] >= XA_F16_MIN_KEYS) p.xa_f16_mask |= 1 << j;
  if (!p.xa_f16_mask) p.xa_opf = 0;
  {  // memories without a key-padding mask get an all-zero one, so the softmax kernel needs no null test
    size_t need = (size_t)Be * L;   // (the un-fused self-attention softmax indexes it per batch row)
    for (int j = 0; j < CFD_NMEM; ++j) need = std::max(need, (size_t)p.U[j] * p.S[j]);
    if (need > c->w->zero_mask.bytes) {
      CHK(c->w->zero_mask.ensure(need));
      HIPCHK(hipMemset(c->w->zero_mask.p, 0, need));
    }
    for (int j = 0; j < CFD_NMEM; ++j) {
      p.has_mask[j] = p.mask[j] != nullptr;
      if (!p.mask[j]) p.mask[j] = c->w->zero_mask.as<uint8_t>();
    }
  }
  p.jbig = -1; p.nruns = 0; p.nlong = 0; p.nshort = Be;
  if (c->use_runs && tmode == 0) {
    int jb = 0;
    for (int j = 1; j < CFD_NMEM; ++j)
      if (p.Sp[j] > p.Sp[jb]) jb = j;
    // worth it only for long latents and long memories (measured: 20.6 vs 21.2 ms at L=196 / 1500 keys, but
    // 3.60 vs 3.15 ms at L=16 / 161 keys, where the extra launches dominate)
    if (p.Sp[jb] >= 256 && L >= 64 && mem[jb].row_map) {
      std::vector<int> hmap(Be), lrows, srows;
      HIPCHK(hipMemcpy(hmap.data(), mem[jb].row_map, (size_t)Be * 4, hipMemcpyDeviceToHost));
      for (int b0 = 0; b0 < Be;) {
        int b1 = b0 + 1;
        while (b1 < Be && hmap[b1] == hmap[b0]) ++b1;
        if (b1 - b0 >= 4 && p.nruns < 8) {
          p.run_row0[p.nruns] = b0; p.run_len[p.nruns] = b1 - b0; p.run_u[p.nruns] = hmap[b0]; ++p.nruns;
          for (int b = b0; b < b1; ++b) lrows.push_back(b);
        } else {
          for (int b = b0; b < b1; ++b) srows.push_back(b);
        }
        b0 = b1;
      }
      if (p.nruns > 0) {
        p.jbig = jb; p.nlong = (int)lrows.size(); p.nshort = (int)srows.size();
        CHK(c->w->long_rows.ensure(lrows.size() * 4 + 16));
        CHK(c->w->short_rows.ensure(srows.size() * 4 + 16));
        HIPCHK(hipMemcpy(c->w->long_rows.p, lrows.data(), lrows.size() * 4, hipMemcpyHostToDevice));
        if (!srows.empty()) HIPCHK(hipMemcpy(c->w->short_rows.p, srows.data(), srows.size() * 4, hipMemcpyHostToDevice));
      }
    }
  }
  p.rt = c->rt_on && tmode == 0 && !g_cfd_naive_gemm && L <= RT_MAX_L && p.M <= c->rt_max_rows && p.Sp_tot <= RT_MAX_KEYS && c->hoist_memside;
  if (p.rt) {
    if (!keep_lists) p.rt_use_inst = Be <= RT_ARG_ROWS;
    for (int j = 0; j < CFD_NMEM && p.rt_use_inst && !keep_lists; ++j) {
      std::vector<int> hm(Be);
      if (mem[j].row_map) HIPCHK(hipMemcpy(hm.data(), mem[j].row_map, (size_t)Be * 4, hipMemcpyDeviceToHost));
      else for (int b = 0; b < Be; ++b) hm[b] = b;
      for (int b = 0; b < Be; ++b) {
        if (hm[b] < 0 || hm[b] >= p.U[j]) return fail(CFD_E_ARG, "memory %s: row_map[%d] = %d outside [0, %d)", MEM_NAMES[j], b, hm[b], p.U[j]);
        if (hm[b] > 255) p.rt_use_inst = 0;
        p.rt_inst[j][b] = (unsigned char)hm[b];
      }
    }
    CHK(c->w->rt_vt.ensure((size_t)Be * CFD_D * RT_MAX_L * 4));
    HIPCHK(hipMemset(c->w->rt_vt.p, 0, (size_t)Be * CFD_D * RT_MAX_L * 4));   // keys beyond L stay zero
  }
  // A forward that returns att_mats, beyond the row-tile path: the fused cross-attention kernel keeps every row's maps itself (its ATT
  // instance + att_fixup_kernel) instead of the three-launch path with its per-call memory-side projections.
  {
    bool any_att = false;
    for (int j = 0; j < CFD_NMEM; ++j) any_att = any_att || p.att[j];
    p.att_fused = any_att && !p.rt && tmode == 0 && c->att_fused && c->fused_xattn && c->hoist_memside && !g_cfd_naive_gemm;
    if (p.att_fused) { p.att_b0 = 0; p.att_nb = Be; }
  }
  if (!keep_lists) CHK(build_xattn_worklist(c, mem));
  if (p.att_fused && p.xa_nwg <= 0) { p.att_fused = false; p.att_nb = 0; }   // (a list too short for the fused kernel: three-launch path)
  if (p.att_fused && !keep_lists) CHK(setup_att_fused(c));
  const long long M = p.M;
  const int nl = c->nl;
  CHK(c->w->x.ensure((size_t)M * CFD_D * 4));
  CHK(c->w->h_sp.ensure((size_t)M * CFD_D * 4));
  CHK(c->w->qk_sp.ensure((size_t)M * 2 * CFD_D * 4));
  CHK(c->w->vts_sp.ensure((size_t)Be * CFD_D * ((L + 63) / 64 * 64) * 4));
  CHK(c->w->ssc.ensure((size_t)Be * CFD_NHEAD * L * p.Lp * 4));
  CHK(c->w->sp_sp.ensure((size_t)Be * CFD_NHEAD * L * p.Lp * 4));
  CHK(c->w->o_sp.ensure((size_t)M * CFD_D * 4));
  CHK(c->w->u_sp.ensure((size_t)M * CFD_FF * 4));
  CHK(c->w->sc.ensure((size_t)M * p.Sp_tot * 4));
  CHK(c->w->p_sp.ensure((size_t)M * p.Sp_tot * 4));
  CHK(c->w->eps.ensure((size_t)M * CFD_LAT * 4));
  CHK(c->w->sample_sp.ensure((size_t)M * CFD_LAT * 4));
  for (int j = 0; j < CFD_NMEM; ++j) {
    const size_t rows = (size_t)p.U[j] * p.Sp[j];
    CHK(c->w->n_sp[j].ensure(rows * CFD_D * 4));
    CHK(c->w->kall_sp[j].ensure(rows * nl * CFD_D * 4));
    CHK(c->w->cb[j].ensure(rows * (nl + 1) * 4));   // + one plane: the per-key scale of the fused cross-attention kernel
    CHK(c->w->vt_all[j].ensure(rows * nl * CFD_D * 4));
  }
  if (c->w->temb_tab.bytes < (size_t)T * CFD_D * 4 || c->w->ss_tab.bytes < (size_t)T * nl * 2 * 2 * CFD_D * 4) {
    c->w->tt_key.clear();          // (a table that is reallocated is an empty one: the timestep-only tables are rebuilt)
    c->w->tt_mem_mask = 0;
  }
  CHK(c->w->temb_tab.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->h1_tab.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->ss_tab.ensure((size_t)T * nl * 2 * 2 * CFD_D * 4));
  CHK(c->w->trows.ensure((size_t)T * 4));
  return CFD_OK;
}

// temb / TimeBlock modulation tables for the T timesteps in `trows_host` (embeddings.py:298-305,
// cross_attention.py:432-434).  temb depends only on t, so a sampling run computes all of its steps once.
static int enqueue_time_tables(Ctx* c, int T, hipStream_t st);
static int build_time_tables(Ctx* c, const int32_t* trows_host, int T, hipStream_t st) {
  for (int i = 0; i < T; ++i)
    if (trows_host[i] < 0 || trows_host[i] >= c->tsin_rows)
      return fail(CFD_E_ARG, "timestep %d outside the sinusoid table (0..%d)", trows_host[i], c->tsin_rows - 1);
  HIPCHK(hipMemcpyAsync(c->w->trows.p, trows_host, (size_t)T * 4, hipMemcpyHostToDevice, st));
  Work* w = c->w;
  const bool same = w->tt_wver == c->wver && (int)w->tt_key.size() == T && std::equal(w->tt_key.begin(), w->tt_key.end(), trows_host);
  if (same) return CFD_OK;          // temb_tab / ss_tab already hold these rows (and kbtab / vbtab may: tt_mem_mask)
  w->tt_key.assign(trows_host, trows_host + T);
  w->tt_wver = c->wver;
  w->tt_mem_mask = 0;
  c->setup_launches += 2 + 2 * c->nl;
  return enqueue_time_tables(c, T, st);
}

// the launches of build_time_tables: table rows from the timestep indices already in w->trows
static int enqueue_time_tables(Ctx* c, int T, hipStream_t st) {
  const int ry = T < 64 ? T : 64;
  const float* W1 = rawp(c, "time_embedding.linear_1.weight");
  const float* b1 = rawp(c, "time_embedding.linear_1.bias");
  const float* W2 = rawp(c, "time_embedding.linear_2.weight");
  const float* b2 = rawp(c, "time_embedding.linear_2.bias");
  const int NE = c->nl * 2 * 2 * CFD_D;
  LAUNCH(CFD_PROF_OTHER, small_linear_kernel, dim3(CFD_D / 4, ry), dim3(256), st, c->tsin.as<float>(), c->w->trows.as<int>(),
         (long long)CFD_D, W1, b1, c->w->h1_tab.as<float>(), (long long)CFD_D, T, CFD_D, 0, 1);
  LAUNCH(CFD_PROF_OTHER, small_linear_kernel, dim3(CFD_D / 4, ry), dim3(256), st, c->w->h1_tab.as<float>(), (const int*)nullptr,
         (long long)CFD_D, W2, b2, c->w->temb_tab.as<float>(), (long long)CFD_D, T, CFD_D, 0, 0);
  // 18 emb_layers at once: rows (2l+tb)*1024 + n ; "post 2" adds 1 to the scale half
  // (post=2 tests n < 512 within each 1024 block -> handled by launching per time block)
  for (int tb = 0; tb < c->nl * 2; ++tb) {
    LAUNCH(CFD_PROF_OTHER, small_linear_kernel, dim3(2 * CFD_D / 4, ry), dim3(256), st, c->w->temb_tab.as<float>(), (const int*)nullptr,
           (long long)CFD_D, c->we_all.as<float>() + (size_t)tb * 2 * CFD_D * CFD_D, c->be_all.as<float>() + (size_t)tb * 2 * CFD_D,
           c->w->ss_tab.as<float>() + (size_t)tb * 2 * CFD_D, (long long)NE, T, 2 * CFD_D, 1, 2);
  }
  return CFD_OK;
}

__global__ void fill_f32_kernel(float* p, long long n, float v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// Once per cfd_forward / sampling run, after the time tables: the part of the memory-side work that does not depend on the
// timestep (see rows.hpp, mem_center_kernel, and xattn_fused.hpp).  Memories in `dynamic_mask` (contents rewritten between the
// iterations of a run: the dyadic rollout's partner projection) keep their per-step projections, and so does every memory when
// the fused cross-attention kernel is not the one that runs (att_mats wanted, small problems, per-row timesteps).
static int prepare_static_memside(Ctx* c, hipStream_t st, int dynamic_mask, bool want_att, bool reuse = false) {
  Problem& p = c->w->pb;
  const int nl = c->nl;
  const long long ROWB = CFD_D * 4;
  p.static_mask = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {   // scale plane = 1 unless mem_scale_all_kernel writes it
    const long long rows = (long long)p.U[j] * p.Sp[j];
    LAUNCH(CFD_PROF_OTHER, fill_f32_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), st, c->w->cb[j].as<float>() + (size_t)nl * rows, rows, 1.0f);
  }
  if (c->w->zeros512.bytes == 0) {
    CHK(c->w->zeros512.ensure(CFD_D * 4));
    HIPCHK(hipMemsetAsync(c->w->zeros512.p, 0, CFD_D * 4, st));
  }
  if (dynamic_mask) p.rt = false;   // (a memory rewritten between iterations keeps its per-step projections: tile-kernel path)
  if (p.xa_one >= 0 && ((dynamic_mask >> p.xa_one) & 1)) {   // the one-key memory is rewritten between iterations: it needs its segments back
    cfd_memory mem[CFD_NMEM];
    memset(mem, 0, sizeof(mem));
    for (int j = 0; j < CFD_NMEM; ++j) {
      mem[j].data = p.mem[j]; mem[j].U = p.U[j]; mem[j].S = p.S[j]; mem[j].row_map = p.map[j];
      mem[j].key_padding_mask = p.has_mask[j] ? p.mask[j] : nullptr;
    }
    const int keep = c->one_key;
    c->one_key = 0;
    const int r = build_xattn_worklist(c, mem);
    c->one_key = keep;
    CHK(r);
  }
  const bool fused = p.rt || (c->fused_xattn && p.xa_nwg > 0 && !want_att && !g_cfd_naive_gemm);
  if (!fused || !c->hoist_memside || p.tmode != 0) { p.xa_opf = 0; return CFD_OK; }
  const int T = p.T;
  if (c->w->b_tab.bytes < (size_t)T * CFD_D * 4 || c->w->b_sp.bytes < (size_t)T * CFD_D * 4 || c->w->bsq.bytes < (size_t)T * 4)
    c->w->tt_mem_mask = 0;       // (a table that is reallocated is an empty one)
  CHK(c->w->b_tab.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->b_sp.ensure((size_t)T * CFD_D * 4));
  CHK(c->w->bsq.ensure((size_t)T * 4));
  if (c->w->tt_mem_mask == 0) {
    c->w->tt_mem_mask = 0;
    LAUNCH(CFD_PROF_OTHER, temb_center_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), st, c->w->temb_tab.as<float>(), T, c->w->b_tab.as<float>(),
           c->w->b_sp.as<char>(), c->w->bsq.as<float>());
    c->setup_launches += 1;
  }
  for (int j = 0; j < CFD_NMEM; ++j) {
    if ((dynamic_mask >> j) & 1) continue;
    const int rows = p.U[j] * p.Sp[j];
    const int NK = nl * CFD_D + 32;
    CHK(c->w->ca[j].ensure((size_t)rows * nl * 4));
    CHK(c->w->asq[j].ensure((size_t)rows * 4));
    if (c->w->kbtab[j].bytes < (size_t)T * NK * 4 || c->w->vbtab[j].bytes < (size_t)T * nl * CFD_D * 4) c->w->tt_mem_mask &= ~(1 << j);
    CHK(c->w->kbtab[j].ensure((size_t)T * NK * 4));
    CHK(c->w->vbtab[j].ensure((size_t)T * nl * CFD_D * 4));
    const bool have_tb = (c->w->tt_mem_mask >> j) & 1;   // A_l b_t / VV_l b_t of this memory are already there for this timestep list
    MemCenterArgs ma{p.mem[j], p.U[j], p.S[j], p.Sp[j], rawp(c, "condition_embedding.weight") + (size_t)j * CFD_D, rawp(c, "mem_pos.pe"),
                     c->w->n_sp[j].as<char>(), c->w->asq[j].as<float>(), c->sat_mem()};
    // (`reuse`: the previous cfd_forward's memories again, cfd_forward_same_memories -- a_s, KA, ca and VA^T are in place)
    if (!reuse) LAUNCH(CFD_PROF_ROWS, mem_center_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), st, ma);
    if (!reuse) {  // KA = A a_s for all layers, ca = c_l . a_s (-inf on dead keys)
      GemmArgs a = gemm_args();
      a.X[0] = c->wk_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = NK; a.Iclamp[0] = NK; a.kt[0] = CFD_D / 32;
      a.Y = c->w->n_sp[j].as<char>(); a.ldy = ROWB; a.J = rows; a.Jclamp = rows;
      a.super_i = 8; a.super_j = 8;
      EpiMemK e{c->w->kall_sp[j].as<char>(), (long long)rows, c->w->ca[j].as<float>(), nl * CFD_D, nl, p.mask[j], p.S[j], p.Sp[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
    if (!reuse) {  // VA^T
      GemmArgs a = gemm_args();
      a.X[0] = c->w->n_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = rows; a.Iclamp[0] = rows; a.kt[0] = CFD_D / 32;
      a.Y = c->wv_all_sp[j].as<char>(); a.ldy = ROWB; a.J = nl * CFD_D; a.Jclamp = nl * CFD_D;
      a.super_i = 8; a.super_j = 8;
      EpiMemV e{c->w->vt_all[j].as<char>(), p.Sp[j], p.U[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
    if (!have_tb) {  // kbtab[t][:] = [A_l b_t for all l | c_l . b_t]
      GemmArgs a = gemm_args();
      a.X[0] = c->wk_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = NK; a.Iclamp[0] = NK; a.kt[0] = CFD_D / 32;
      a.Y = c->w->b_sp.as<char>(); a.ldy = ROWB; a.J = T; a.Jclamp = T;
      EpiF32 e;
      memset(&e, 0, sizeof(e));
      e.out = c->w->kbtab[j].as<float>(); e.ldo = NK;
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
    if (!have_tb) {  // vbtab[t][:] = VV_l b_t for all l
      GemmArgs a = gemm_args();
      a.X[0] = c->wv_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = nl * CFD_D; a.Iclamp[0] = nl * CFD_D; a.kt[0] = CFD_D / 32;
      a.Y = c->w->b_sp.as<char>(); a.ldy = ROWB; a.J = T; a.Jclamp = T;
      EpiF32 e;
      memset(&e, 0, sizeof(e));
      e.out = c->w->vbtab[j].as<float>(); e.ldo = nl * CFD_D;
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
      c->w->tt_mem_mask |= 1 << j;
      c->setup_launches += 2;
    }
    p.static_mask |= 1 << j;
  }
  if (p.xa_opf && (p.rt || p.static_mask != (1 << CFD_NMEM) - 1)) p.xa_opf = 0;   // (single-fp16 tiles: every memory static, tile kernels)
  if (p.xa_opf) {   // this run's operand policy: the key / value tiles of the fused cross-attention as single fp16, packed tile by tile
    for (int j = 0; j < CFD_NMEM; ++j) {
      if (!((p.xa_f16_mask >> j) & 1)) continue;   // (short memories keep pairs: xattn_fused.hpp, OPF)
      const long long tiles = (long long)nl * p.U[j] * (p.Sp[j] / XA_KEYS), chunks = tiles * 2048;
      if (p.xa_opf & XA_V16) {
        CHK(c->w->v16[j].ensure((size_t)tiles * 32768));
        LAUNCH(CFD_PROF_ROWS, xa_pack16_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), st, c->w->vt_all[j].as<char>(), c->w->v16[j].as<char>(), chunks, p.Sp[j], 0);
      }
      if (p.xa_opf & XA_K16) {
        CHK(c->w->k16[j].ensure((size_t)tiles * 32768));
        LAUNCH(CFD_PROF_ROWS, xa_pack16_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), st, c->w->kall_sp[j].as<char>(), c->w->k16[j].as<char>(), chunks, p.Sp[j], 1);
      }
    }
  }
  if (p.xa_one >= 0 && !p.rt) {   // the one-key memory's value rows as float32 vectors (xattn_fused.hpp, XAttnArgs::one_va).  Also with `reuse`:
                                  // the previous forward of these memories may have had another L or the row-tile path and never made them (one tiny launch)
    const int j = p.xa_one;
    const long long n = (long long)nl * p.U[j] * CFD_D;
    CHK(c->w->xa_one_va.ensure((size_t)n * 4));
    LAUNCH(CFD_PROF_ROWS, one_key_va_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), st, c->w->vt_all[j].as<char>(), n, p.Sp[j], c->w->xa_one_va.as<float>());
  }
  if (p.rt) {   // per-key scale and key bias of every step of the run (the tile-kernel path makes one step's per iteration: mem_scale_all_kernel)
    for (int j = 0; j < CFD_NMEM; ++j) {
      const long long rows = (long long)p.U[j] * p.Sp[j];
      CHK(c->w->rt_cbt[j].ensure((size_t)T * (nl + 1) * rows * 4));
      MemScaleTabArgs a{c->w->n_sp[j].as<char>(), c->w->asq[j].as<float>(), rows, c->w->b_tab.as<float>(), c->w->bsq.as<float>(), c->w->ca[j].as<float>(),
                        c->w->kbtab[j].as<float>() + (size_t)nl * CFD_D, (long long)(nl * CFD_D + 32), nl, c->w->rt_cbt[j].as<float>()};
      LAUNCH(CFD_PROF_ROWS, mem_scale_table_kernel, dim3((unsigned)((rows + 3) / 4), (unsigned)T), dim3(256), st, a);
    }
  }
  return CFD_OK;
}

// ---- the denoiser forward: Denoiser.forward (denoiser.py:173-386) --------------------------------------
// Input: c->w->sample_sp (SP [M][128]); time tables built; output: c->w->eps (fp32 [M][128]).
static int enqueue_rows(Ctx* c, hipStream_t st, int row0, int nrows);

// memory-side work of one forward: shared by every row chunk
static int enqueue_memside(Ctx* c, hipStream_t st) {
  const Problem& p = c->w->pb;
  if (p.rt) return CFD_OK;   // every memory is static and its per-step scalars are tabulated (prepare_static_memside)
  const int nl = c->nl;
  const int* dstep = p.tmode ? c->w->d_step.as<int>() + 1 : c->w->d_step.as<int>();
  const long long ROWB = CFD_D * 4;
  const dim3 blk(256);
  // memories whose projections were made once for the run: this step's per-key scale and key bias
  {
    MemScaleAllArgs g;
    memset(&g, 0, sizeof(g));
    int nwg = 0;
    for (int j = 0; j < CFD_NMEM; ++j) {
      if (!((p.static_mask >> j) & 1)) continue;
      const long long rows = (long long)p.U[j] * p.Sp[j];
      const int NK = nl * CFD_D + 32;
      g.m[g.n] = MemScaleArgs{c->w->n_sp[j].as<char>(), c->w->asq[j].as<float>(), rows, c->w->b_tab.as<float>(), c->w->bsq.as<float>(), c->w->ca[j].as<float>(),
                              c->w->kbtab[j].as<float>() + (size_t)nl * CFD_D, (long long)NK, dstep, nl, c->w->cb[j].as<float>() + (size_t)nl * rows,
                              c->w->cb[j].as<float>()};
      g.first[g.n] = nwg;
      nwg += (int)((rows + 3) / 4);
      ++g.n;
    }
    g.first[g.n] = nwg;
    if (g.n > 0) LAUNCH(CFD_PROF_ROWS, mem_scale_all_kernel, dim3((unsigned)nwg), blk, st, g);
  }
  // 2. memories: + temb + condition id + PE, normalise           (denoiser.py:223-261,332-353)
  for (int j = 0; j < CFD_NMEM; ++j) {
    if ((p.static_mask >> j) & 1) continue;
    MemPrepArgs a{p.mem[j], p.U[j], p.S[j], p.Sp[j], c->w->temb_tab.as<float>(), dstep, p.tmode,
                  rawp(c, "condition_embedding.weight") + (size_t)j * CFD_D, rawp(c, "mem_pos.pe"), c->w->n_sp[j].as<char>()};
    const long long rows = (long long)p.U[j] * p.Sp[j];
    LAUNCH(CFD_PROF_ROWS, mem_prep_kernel, dim3((unsigned)((rows + 3) / 4)), blk, st, a);
  }
  // 3. memory-side projections for ALL layers at once: folded keys (+ key bias) and folded values^T
  for (int j = 0; j < CFD_NMEM; ++j) {
    if ((p.static_mask >> j) & 1) continue;
    const int rows = p.U[j] * p.Sp[j];
    c->memside_in_forward = true;   // these epilogues count into the handle's census: whoever waits for this stream next reads it
    {
      GemmArgs a = gemm_args();
      a.X[0] = c->wk_all_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = nl * CFD_D + 32; a.Iclamp[0] = nl * CFD_D + 32; a.kt[0] = CFD_D / 32;
      a.Y = c->w->n_sp[j].as<char>(); a.ldy = ROWB; a.J = rows; a.Jclamp = rows;
      a.super_i = 8; a.super_j = 8;
      EpiMemK e{c->w->kall_sp[j].as<char>(), (long long)rows, c->w->cb[j].as<float>(), nl * CFD_D, nl, p.mask[j], p.S[j], p.Sp[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
    {
      GemmArgs a = gemm_args();
      a.X[0] = c->w->n_sp[j].as<char>(); a.ldx[0] = ROWB; a.I[0] = rows; a.Iclamp[0] = rows; a.kt[0] = CFD_D / 32;
      a.Y = c->wv_all_sp[j].as<char>(); a.ldy = ROWB; a.J = nl * CFD_D; a.Jclamp = nl * CFD_D;
      a.super_i = 8; a.super_j = 8;
      EpiMemV e{c->w->vt_all[j].as<char>(), p.Sp[j], p.U[j], c->sat_mem()};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_MEM, a, e, 1, 1, st)));
    }
  }

  return CFD_OK;
}

static int enqueue_rows_rt(Ctx* c, hipStream_t st, const RtSave* sv = nullptr);

static int enqueue_denoise(Ctx* c, hipStream_t st) {
  CHK(enqueue_memside(c, st));
  if (c->w->pb.rt) return enqueue_rows_rt(c, st);
  return enqueue_rows(c, st, 0, c->w->pb.Be);
}

// The forward for small problems: launches of 16-token x 16-feature workgroups (rowtile.hpp); same buffers, same tap points.
// With `sv` (the WEG evaluation, weg_rt.hpp) every residual update goes to a buffer of its own, the self-attention operands, the
// cross-attention scores and the FFN pre-activations of every layer are kept, and the pass ends behind the last layer's
// cross-attention (nothing above it reaches the objective).

static int enqueue_rows_rt(Ctx* c, hipStream_t st, const RtSave* sv) {
  const Problem& p = c->w->pb;
  const int nl = c->nl, L = p.L, tpr = (L + 15) / 16, ntile = p.Be * tpr;
  const int* dstep = c->w->d_step.as<int>();
  const int lds_xpv = (p.Sp_tot / 32) * 2048 + 8 * (p.Sp_tot <= 512 ? 2 : 4) * 2048 + 16 * 32 * 16 + 512 + 2048;
#define RT_SET_LDS(kernel, bytes) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes))
  static unsigned long long attr = 0;
  if (!((attr >> (c->cfg.device & 63)) & 1ull)) {
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_EMBED, 256, CFD_LAT / 32, 1>), rt_gemm_lds(RT_PRO_SP, 256, CFD_LAT / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_LN, RT_EPI_QKV, 512, CFD_D / 32, 3>), rt_gemm_lds(RT_PRO_LN, 512, CFD_D / 32, 3));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 1>), rt_gemm_lds(RT_PRO_SP, 256, CFD_D / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 1>), rt_gemm_lds(RT_PRO_ADALN, 512, CFD_D / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_LN, RT_EPI_SPLIT, 512, CFD_D / 32, 2>), rt_gemm_lds(RT_PRO_LN, 512, CFD_D / 32, 2));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_RESID, 512, CFD_FF / 32, 1>), rt_gemm_lds(RT_PRO_SP, 512, CFD_FF / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_LN, RT_EPI_F32, 512, CFD_D / 32, 1>), rt_gemm_lds(RT_PRO_LN, 512, CFD_D / 32, 1));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 2>), rt_gemm_lds(RT_PRO_SP, 256, CFD_D / 32, 2));
    RT_SET_LDS((rt_gemm_kernel<RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 2>), rt_gemm_lds(RT_PRO_ADALN, 512, CFD_D / 32, 2));
    RT_SET_LDS(rt_xscore_kernel, RT_XS_LDS);
    RT_SET_LDS(rt_xpv_kernel<512>, (512 / 32) * 2048 + 8 * 2 * 2048 + 16 * 32 * 16 + 512 + 2048);
    RT_SET_LDS(rt_xpv_kernel<RT_MAX_KEYS>, (RT_MAX_KEYS / 32) * 2048 + 8 * 4 * 2048 + 16 * 32 * 16 + 512 + 2048);
    attr |= 1ull << (c->cfg.device & 63);
  }
#undef RT_SET_LDS
  // The residual stream alternates between two buffers: a time block's workgroups read COMPLETE rows (LayerNorm prologue) while the
  // other workgroups of the tile write their 16 features of the sum, so it must not run in place.  x -> (time block 1) -> h ->
  // (cross-attention) -> x -> (time block 2) -> h -> (FFN) -> x; the products whose prologue reads another matrix (out-projection,
  // FFN2) and the cross-attention's second half touch only their own 16 features of the rows and may update in place.
  float* const xw = c->w->x.as<float>();
  float* const hw = c->w->h_sp.as<float>();   // (the tile-kernel path's LayerNorm output: same bytes, unused here)
  auto X = [&](int l, int k) -> float* { return sv ? sv->x[l][k] : ((k == 2 || k == 4) ? hw : xw); };
  // This step's rows of the per-step tables.  One table row (cfd_forward, the WEG evaluation): the tables themselves.  A sampling
  // run: fixed buffers refreshed by ONE launch at the start of the iteration, so that no launch of the iteration has the step index
  // as a dependent scalar load in front of its operand loads.
  const float* ss_now = c->w->ss_tab.as<float>();
  const float *kb_now[CFD_NMEM], *vb_now[CFD_NMEM], *cbt_now[CFD_NMEM];
  for (int j = 0; j < CFD_NMEM; ++j) { kb_now[j] = c->w->kbtab[j].as<float>(); vb_now[j] = c->w->vbtab[j].as<float>(); cbt_now[j] = c->w->rt_cbt[j].as<float>(); }
  if (p.T > 1) {
    RtStepRowsArgs ra;
    memset(&ra, 0, sizeof(ra));
    size_t off4 = 0;
    int nt = 0, nwg = 0;
    auto add = [&](const float*& now, int nfloat) {
      float* dst = c->w->rt_cur.as<float>() + off4 * 4;
      ra.src[nt] = now; ra.dst[nt] = dst; ra.n4[nt] = nfloat / 4; ra.first[nt] = nwg;
      nwg += (nfloat / 4 + 255) / 256; off4 += (size_t)(nfloat / 4 + 63) / 64 * 64; ++nt;
      now = dst;
    };
    size_t need4 = (size_t)(nl * 4 * CFD_D / 4 + 64);
    for (int j = 0; j < CFD_NMEM; ++j) need4 += (size_t)((nl * CFD_D + 32) / 4 + 64) + (size_t)(nl * CFD_D / 4 + 64) + (size_t)((nl + 1) * p.U[j] * p.Sp[j] / 4 + 64);
    CHK(c->w->rt_cur.ensure(need4 * 16));
    add(ss_now, nl * 4 * CFD_D);
    for (int j = 0; j < CFD_NMEM; ++j) { add(kb_now[j], nl * CFD_D + 32); add(vb_now[j], nl * CFD_D); add(cbt_now[j], (nl + 1) * p.U[j] * p.Sp[j]); }
    ra.ntab = nt; ra.first[nt] = nwg; ra.d_step = dstep;
    LAUNCH(CFD_PROF_OTHER, rt_step_rows_kernel, dim3(nwg), dim3(256), st, ra);
  }
  c->w->now_ss = ss_now;
  for (int j = 0; j < CFD_NMEM; ++j) { c->w->now_kb[j] = kb_now[j]; c->w->now_vb[j] = vb_now[j]; }
  RtGemmArgs base;
  memset(&base, 0, sizeof(base));
  base.L = L; base.tpr = tpr;
  // Two 16-feature blocks per workgroup for the 512 x 512 residual products from two utterances on: half the workgroups, each normalising
  // its 16 rows once for two blocks (one utterance: 0.411 -> 0.421 s per 1000 steps, two: 0.544 -> 0.528, four: 0.882 -> 0.869; same sums
  // in the same order, so bit-identical).  CFD_RT_NFB2_TILES=<token tiles> moves the threshold (read at cfd_create).
  const bool nfb2 = ntile >= c->rt_nfb2_tiles && !sv;
#define RT_LAUNCH(cls, PRO, EPI, NT, KT, NFB, nfeat, args)                                                        \
  do {                                                                                                          \
    Bracket _br(c, cls, st);                                                                                    \
    hipLaunchKernelGGL((rt_gemm_kernel<PRO, EPI, NT, KT, NFB>), dim3((nfeat) / (16 * NFB), ntile), dim3(NT), rt_gemm_lds(PRO, NT, KT, NFB), st, args); \
    hipError_t _e = hipGetLastError();                                                                          \
    if (_e != hipSuccess) return fail(CFD_E_HIP, "row-tile launch failed: %s", hipGetErrorString(_e));          \
  } while (0)
  // 1. latent embedding + body/hand embedding + query PE          (denoiser.py:187,316-326)
  {
    RtGemmArgs a = base;
    a.a_sp = c->w->sample_sp.as<char>(); a.w = c->we_sp.as<char>(); a.bias = rawp(c, "latent_embd.bias");
    a.bh = rawp(c, "bh_embedding.weight"); a.qpe = rawp(c, "query_pos.pe"); a.xo = X(0, 0);
    RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_EMBED, 256, CFD_LAT / 32, 1, CFD_D, a);
  }
  if (c->stop_stage == 1) return CFD_OK;
  auto time_block = [&](const DBuf& w, const float* g, const float* b, const float* bias, int tbidx, const float* xin, float* xout) -> int {
    RtGemmArgs a = base;
    a.x = xin; a.xr = xin; a.xo = xout;
    a.g = g; a.b = b; a.ss = ss_now + (size_t)tbidx * 2 * CFD_D; a.w = w.as<char>(); a.bias = bias;
    if (nfb2) RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 2, CFD_D, a);
    else RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_ADALN, RT_EPI_RESID, 512, CFD_D / 32, 1, CFD_D, a);
    return CFD_OK;
  };
  RtXArgs xa;
  memset(&xa, 0, sizeof(xa));
  xa.L = L; xa.tpr = tpr; xa.nl = nl; xa.Sp_tot = p.Sp_tot; xa.rsp = c->w->p_sp.as<float>();
  int nkb = 0;
  for (int j = 0; j < CFD_NMEM; ++j) {
    xa.map[j] = p.map[j]; xa.rows[j] = p.U[j] * p.Sp[j]; xa.S[j] = p.S[j]; xa.Sp[j] = p.Sp[j]; xa.off[j] = p.off[j];
    xa.cbt[j] = cbt_now[j]; xa.att[j] = p.att[j]; xa.att_slot[j] = p.att_slot[j];
    memcpy(xa.inst[j], p.rt_inst[j], RT_ARG_ROWS);
    xa.blk0[j] = nkb; nkb += p.Sp[j] / 16;
  }
  xa.blk0[CFD_NMEM] = nkb;
  xa.att_b0 = p.att_nb > 0 ? p.att_b0 : 0;
  xa.att_nb = p.att_nb > 0 ? p.att_nb : p.Be;
  xa.att_step = p.att_nb > 0 ? c->w->d_step.as<int>() : nullptr;
  for (int l = 0; l < nl; ++l) {
    const LayerW& w = c->lw[l];
    char* qk = sv ? sv->qk[l] : c->w->qk_sp.as<char>();
    char* vt = sv ? sv->vt[l] : c->w->rt_vt.as<char>();
    // ---- a. self attention: x += Wo softmax(q k^T) v                         (cross_attention.py:568-572)
    {
      RtGemmArgs a = base;   // norm1 + q | k | v^T projections
      a.x = X(l, 0);
      a.g = w.ln1g; a.b = w.ln1b; a.w = w.wqk_sp.as<char>(); a.w2 = w.wv_sp.as<char>(); a.nfb_qk = 2 * CFD_D / 16;
      a.bias = w.bqk.as<float>(); a.o_sp = qk; a.ld_o = 2 * CFD_D * 4; a.vt = vt;
      RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_LN, RT_EPI_QKV, 512, CFD_D / 32, 3, 3 * CFD_D, a);
    }
    {
      RtSelfArgs a{qk, vt, c->w->o_sp.as<char>(), L, tpr};
      Bracket br(c, CFD_PROF_GEMM_ATTN, st);
      hipLaunchKernelGGL(rt_selfattn_kernel, dim3(CFD_NHEAD, ntile), dim3(256), 40960, st, a);
      HIPCHK(hipGetLastError());
    }
    {
      RtGemmArgs a = base;   // out-projection + residual
      a.xr = X(l, 0); a.xo = X(l, 1);
      a.a_sp = c->w->o_sp.as<char>(); a.w = w.wo_sp.as<char>(); a.bias = w.bo2.as<float>();
      if (nfb2) RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 2, CFD_D, a);
      else RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_RESID, 256, CFD_D / 32, 1, CFD_D, a);
    }
    if (c->stop_stage == 2 + 4 * l) return CFD_OK;
    // ---- b. time block 1                                                        (:575, :426-439)
    CHK(time_block(w.wtb1_sp, w.tb1g, w.tb1b, w.btb1, 2 * l, X(l, 1), X(l, 2)));
    if (c->stop_stage == 3 + 4 * l) {   // (test hook: the tap is read from x)
      if (!sv) HIPCHK(hipMemcpyAsync(xw, hw, (size_t)p.M * CFD_D * 4, hipMemcpyDeviceToDevice, st));
      return CFD_OK;
    }
    // ---- c-e. five cross attentions + fuser, folded                             (:578-652)
    {
      RtXArgs a = xa;
      a.x = X(l, 2); a.xo = X(l, 3);
      a.sc = sv ? sv->sc[l] : c->w->sc.as<float>();
      a.cst = sv ? sv->cst[l] : c->w->ssc.as<float>();   // (the tile-kernel path's self-attention score buffer: >= M x 32 float4, unused here)
      a.ln_g = w.ln2g; a.ln_b = w.ln2b; a.bias = w.cross_bias.as<float>(); a.layer = l;
      for (int j = 0; j < CFD_NMEM; ++j) {
        const size_t rows = (size_t)p.U[j] * p.Sp[j];
        a.K[j] = c->w->kall_sp[j].as<char>() + (size_t)l * rows * CFD_D * 4;
        a.VT[j] = c->w->vt_all[j].as<char>() + (size_t)l * rows * CFD_D * 4;
        a.kb[j] = kb_now[j] + (size_t)l * CFD_D;
        a.vb[j] = vb_now[j] + (size_t)l * CFD_D;
      }
      {
        Bracket br(c, CFD_PROF_XATTN, st);
        hipLaunchKernelGGL(rt_xscore_kernel, dim3(p.Sp_tot / 32, ntile), dim3(512), RT_XS_LDS, st, a);
        HIPCHK(hipGetLastError());
      }
      {
        Bracket br(c, CFD_PROF_XATTN, st);
        if (p.Sp_tot <= 512) hipLaunchKernelGGL(rt_xpv_kernel<512>, dim3(CFD_D / 16, ntile), dim3(512), lds_xpv, st, a);
        else hipLaunchKernelGGL(rt_xpv_kernel<RT_MAX_KEYS>, dim3(CFD_D / 16, ntile), dim3(512), lds_xpv, st, a);
        HIPCHK(hipGetLastError());
      }
    }
    if (c->stop_stage == 4 + 4 * l) return CFD_OK;
    if (sv && l == nl - 1) return CFD_OK;
    // ---- f. time block 2                                                        (:655)
    CHK(time_block(w.wtb2_sp, w.tb2g, w.tb2b, w.btb2, 2 * l + 1, X(l, 3), X(l, 4)));
    // ---- g. FFN                                                                 (:659-661)
    {
      RtGemmArgs a = base;   // norm3 + linear1 + GELU
      a.x = X(l, 4);
      a.g = w.ln3g; a.b = w.ln3b; a.w = w.w1_sp.as<char>(); a.bias = w.b1; a.o_sp = c->w->u_sp.as<char>(); a.ld_o = CFD_FF * 4; a.gelu = 1;
      a.pre = sv ? sv->pre[l] : nullptr;
      RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_LN, RT_EPI_SPLIT, 512, CFD_D / 32, 2, CFD_FF, a);
    }
    {
      RtGemmArgs a = base;   // linear2 + residual
      a.xr = X(l, 4); a.xo = X(l + 1, 0);
      a.a_sp = c->w->u_sp.as<char>(); a.w = w.w2_sp.as<char>(); a.bias = w.b2;
      RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_SP, RT_EPI_RESID, 512, CFD_FF / 32, 1, CFD_D, a);
    }
    if (c->stop_stage == 5 + 4 * l) return CFD_OK;
  }
  // 7. final norm + latent projection                                (cross_attention.py:238-239, denoiser.py:382)
  {
    RtGemmArgs a = base;
    a.x = X(nl, 0);
    a.g = rawp(c, "decoder.norm.weight"); a.b = rawp(c, "decoder.norm.bias"); a.w = c->wp_sp.as<char>();
    a.bias = rawp(c, "latent_proj.bias"); a.o_f32 = c->w->eps.as<float>(); a.ldo_f = CFD_LAT;
    RT_LAUNCH(CFD_PROF_GEMM_TOKEN, RT_PRO_LN, RT_EPI_F32, 512, CFD_D / 32, 1, CFD_LAT, a);
  }
#undef RT_LAUNCH
  return CFD_OK;
}

static int enqueue_rows(Ctx* c, hipStream_t st, int row0, int Be) {
  const Problem& p = c->w->pb;
  const int nl = c->nl, L = p.L;
  const long long M = (long long)Be * L;
  const int* dstep = p.tmode ? c->w->d_step.as<int>() + 1 : c->w->d_step.as<int>();
  const long long ROWB = CFD_D * 4;  // bytes per SP row of 512
  const dim3 blk(256);
  const char* sample_sp = c->w->sample_sp.as<char>() + (size_t)row0 * L * CFD_LAT * 4;
  float* eps_out = c->w->eps.as<float>() + (size_t)row0 * L * CFD_LAT;
  const int* mapj[CFD_NMEM];
  float* attj[CFD_NMEM];
  bool want_att = false;
  for (int j = 0; j < CFD_NMEM; ++j) {
    mapj[j] = p.map[j] + row0;
    attj[j] = p.att[j] ? p.att[j] + (size_t)row0 * nl * L * p.S[j] : nullptr;
    want_att = want_att || p.att[j];
  }
  if (p.att_fused) want_att = false;   // (the ring of a sampling run on the tile kernels: the fused kernel keeps the maps itself)
  // one fused kernel per layer for the cross-attention block, unless att_mats are wanted (or the naive debug GEMMs)
  const bool fused_x = c->fused_xattn && p.xa_nwg > 0 && !want_att && !g_cfd_naive_gemm && row0 == 0 && Be == p.Be;

  // rows that run the replica-independent head of the network (see Problem::share_B)
  const bool share = p.share_B > 0 && row0 == 0 && Be == p.Be && Be % p.share_B == 0 && Be > p.share_B && !c->stop_stage;
  const int Bs = share ? p.share_B : Be;
  const long long Ms = (long long)Bs * L;
  // 1. latent embedding + body/hand embedding + query PE          (denoiser.py:187,316-326)
  {
    GemmArgs a = gemm_args();
    a.X[0] = c->we_sp.as<char>(); a.ldx[0] = CFD_LAT * 4; a.I[0] = CFD_D; a.Iclamp[0] = CFD_D; a.kt[0] = CFD_LAT / 32;
    a.Y = sample_sp; a.ldy = CFD_LAT * 4; a.J = (int)Ms; a.Jclamp = (int)Ms;
    EpiEmbed e{c->w->x.as<float>(), rawp(c, "latent_embd.bias"), rawp(c, "bh_embedding.weight"), rawp(c, "query_pos.pe"), L};
    CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
  }
  if (c->stop_stage == 1) return CFD_OK;
  // This step's rows of the per-step tables (as on the row-tile path, enqueue_rows_rt): with one timestep for all rows, a sampling run
  // refreshes fixed buffers with ONE launch at the start of the iteration and cfd_forward points at its single table row, so that no
  // launch has the step index as a dependent scalar load in front of its operand loads (a load that misses in every XCD's L2 after the
  // previous iteration's last workgroup has written it: ~1 us on 18 AdaLN launches and 9 cross-attention prologues per step).
  const float* ss_now = c->w->ss_tab.as<float>();
  const float *kb_now[CFD_NMEM], *vb_now[CFD_NMEM];
  for (int j = 0; j < CFD_NMEM; ++j) { kb_now[j] = c->w->kbtab[j].as<float>(); vb_now[j] = c->w->vbtab[j].as<float>(); }
  const bool rows_now = p.tmode == 0 && c->step_rows;
  if (rows_now && p.T > 1) {
    RtStepRowsArgs ra;
    memset(&ra, 0, sizeof(ra));
    size_t off4 = 0;
    int nt = 0, nwg = 0;
    auto add = [&](const float*& now, int nfloat) {
      float* dst = c->w->rt_cur.as<float>() + off4 * 4;
      ra.src[nt] = now; ra.dst[nt] = dst; ra.n4[nt] = nfloat / 4; ra.first[nt] = nwg;
      nwg += (nfloat / 4 + 255) / 256; off4 += (size_t)(nfloat / 4 + 63) / 64 * 64; ++nt;
      now = dst;
    };
    size_t need4 = (size_t)(nl * 4 * CFD_D / 4 + 64);
    for (int j = 0; j < CFD_NMEM; ++j) need4 += (size_t)((nl * CFD_D + 32) / 4 + 64) + (size_t)(nl * CFD_D / 4 + 64);
    CHK(c->w->rt_cur.ensure(need4 * 16));
    add(ss_now, nl * 4 * CFD_D);
    for (int j = 0; j < CFD_NMEM; ++j)
      if ((p.static_mask >> j) & 1) { add(kb_now[j], nl * CFD_D + 32); add(vb_now[j], nl * CFD_D); }
    ra.ntab = nt; ra.first[nt] = nwg; ra.d_step = dstep;
    LAUNCH(CFD_PROF_OTHER, rt_step_rows_kernel, dim3(nwg), dim3(256), st, ra);
  }
  auto ln = [&](const float* g, const float* b, int adaln, int tbidx, char* out, long long rows) -> int {
    LnArgs a{c->w->x.as<float>(), out, rows, g, b, adaln, (rows_now ? ss_now : c->w->ss_tab.as<float>()) + (size_t)tbidx * 2 * CFD_D,
             (long long)nl * 2 * 2 * CFD_D, rows_now ? nullptr : dstep, p.tmode, L, row0};
    LAUNCH(CFD_PROF_ROWS, ln_rows_kernel, dim3((unsigned)((rows + 3) / 4)), blk, st, a);
    return CFD_OK;
  };
  auto token_gemm_resid = [&](const DBuf& w, int K, const char* y, const float* bias, long long rows) -> int {
    GemmArgs a = gemm_args();
    a.X[0] = w.as<char>(); a.ldx[0] = (long long)K * 4; a.I[0] = CFD_D; a.Iclamp[0] = CFD_D; a.kt[0] = K / 32;
    a.Y = y; a.ldy = (long long)K * 4; a.J = (int)rows; a.Jclamp = (int)rows;
    EpiResid e{c->w->x.as<float>(), 0, bias};
    return run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st);
  };
  // residual product + the LayerNorm that follows it
  auto token_gemm_resid_ln = [&](const DBuf& w, int K, const char* y, const float* bias, long long rows, const float* g, const float* b,
                                 int adaln, int tbidx) -> int {
    CHK(token_gemm_resid(w, K, y, bias, rows));
    return ln(g, b, adaln, tbidx, c->w->h_sp.as<char>(), rows);
  };
  static unsigned long long attr = 0;   // per device (one bit per ordinal): a process may hold handles on several GPUs
  if (!((attr >> (c->cfg.device & 63)) & 1ull)) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attn_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_fused_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS));
    attr |= 1ull << (c->cfg.device & 63);
  }

  bool h_ready = false;   // h_sp already holds norm1(x) of the layer that starts
  for (int l = 0; l < nl; ++l) {
    const LayerW& w = c->lw[l];
    // rows of sub-layers a and b: layer 0 runs them once per utterance when the batch is G replicas of it
    const int Ba = (l == 0) ? Bs : Be;
    const long long Ma = (long long)Ba * L;
    // ---- a. self attention: x += Wo softmax(q k^T) v                         (cross_attention.py:568-572)
    // (norm1 of layers 1.. is made by the previous layer's last residual product when that ran row-complete: `h_ready`)
    if (!h_ready) CHK(ln(w.ln1g, w.ln1b, 0, 0, c->w->h_sp.as<char>(), Ma));
    h_ready = false;
    bool qkv_one_launch = false;
    const int Lv = (L + 31) / 32 * 32;   // (whole 32-key blocks: at L = 16 a 64-key pitch made the v^T product twice the work of the q | k one)
    {
      // q (pre-scaled) and k, token-major ...
      GemmArgs a = gemm_args();
      a.X[0] = w.wqk_sp.as<char>(); a.ldx[0] = ROWB; a.I[0] = 2 * CFD_D; a.Iclamp[0] = 2 * CFD_D; a.kt[0] = CFD_D / 32;
      a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.J = (int)Ma; a.Jclamp = (int)Ma;
      EpiSplit e{c->w->qk_sp.as<char>(), 2 * ROWB, 0, 0, w.bqk.as<float>(), 0, 0};
      // ... and v^T per batch row: vts[b][f][l] (keys in P-fragment order for the fused kernel)
      GemmArgs av = gemm_args();
      av.X[0] = c->w->h_sp.as<char>(); av.ldx[0] = ROWB; av.xbs[0] = (long long)L * ROWB; av.I[0] = Lv; av.Iclamp[0] = L; av.kt[0] = CFD_D / 32;
      av.Y = w.wv_sp.as<char>(); av.ldy = ROWB; av.J = CFD_D; av.Jclamp = CFD_D;
      EpiSplit ev{c->w->vts_sp.as<char>(), (long long)Lv * 4, (long long)CFD_D * Lv * 4, 0, nullptr, 0, 1};
      if (L == 16 && c->qkv_fused && !g_cfd_naive_gemm) {
        // batch rows of exactly 16 tokens: both in ONE grouped launch, the value projection stored transposed by the epilogue (EpiQkvT)
        GemmArgs ag = a;
        ag.nslot = 2;
        ag.X[1] = w.wv_sp.as<char>(); ag.ldx[1] = ROWB; ag.I[1] = CFD_D; ag.Iclamp[1] = CFD_D; ag.kt[1] = CFD_D / 32;
        EpiQkvT eg{c->w->qk_sp.as<char>(), 2 * ROWB, w.bqk.as<float>(), c->w->vts_sp.as<char>(), c->qkv_fused == 1 ? 1 : 0};
        CHK((run_gemm<MODE_GROUPED>(c, CFD_PROF_GEMM_TOKEN, ag, eg, 1, 1, st)));
        qkv_one_launch = true;
      } else {
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, av, ev, Ba, 1, st)));
      }
    }
    if (qkv_one_launch && c->qkv_fused == 1) {
      // one query tile and one key tile per (row, head): the row-tile path's attention core (4 waves that all compute; V^T in natural key
      // order, which EpiQkvT wrote) instead of the flash kernel's 8-wave workgroup with one busy wave
      RtSelfArgs a{c->w->qk_sp.as<char>(), c->w->vts_sp.as<char>(), c->w->o_sp.as<char>(), L, 1};
      Bracket br(c, CFD_PROF_GEMM_ATTN, st);
      hipLaunchKernelGGL(rt_selfattn_kernel, dim3(CFD_NHEAD, Ba), dim3(256), 40 * 1024, st, a);
      HIPCHK(hipGetLastError());
    } else {
      SelfAttnArgs a{c->w->qk_sp.as<char>(), c->w->vts_sp.as<char>(), c->w->o_sp.as<char>(), L, Lv};
      Bracket br(c, CFD_PROF_GEMM_ATTN, st);
      hipLaunchKernelGGL(self_attn_fused_kernel, dim3((L + SELF_ATTN_WAVES * 16 - 1) / (SELF_ATTN_WAVES * 16), CFD_NHEAD, Ba), dim3(SELF_ATTN_WAVES * 64), 65536, st, a);
      HIPCHK(hipGetLastError());
    }
    // out-projection + residual, then time block 1's AdaLN + SiLU                 (:572, :575, :426-439)
    CHK(token_gemm_resid_ln(w.wo_sp, CFD_D, c->w->o_sp.as<char>(), w.bo2.as<float>(), Ma, w.tb1g, w.tb1b, 1, 2 * l));
    if (c->stop_stage == 2 + 4 * l) return CFD_OK;
    // ---- b. time block 1
    CHK(token_gemm_resid(w.wtb1_sp, CFD_D, c->w->h_sp.as<char>(), w.btb1, Ma));
    if (c->stop_stage == 3 + 4 * l) return CFD_OK;
    if (Ma != M) {   // every guidance chunk starts its first cross-attention from the same state
      const long long n4 = Ma * (CFD_D / 4);
      LAUNCH(CFD_PROF_ROWS, replicate_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), blk, st, reinterpret_cast<float4*>(c->w->x.as<float>()), n4,
             (int)(M / Ma));
    }
    // ---- c-e. five cross attentions + fuser, folded                             (:578-652)
    if (fused_x) {   // LayerNorm2 is part of the kernel's prologue
      XAttnArgs a;
      memset(&a, 0, sizeof(a));
      a.x = c->w->x.as<float>(); a.ln_g = w.ln2g; a.ln_b = w.ln2b; a.bias = w.cross_bias.as<float>(); a.L = L;
      for (int j = 0; j < CFD_NMEM; ++j) {
        const size_t rows = (size_t)p.U[j] * p.Sp[j];
        // (single-fp16 tiles of a long memory, Problem::xa_opf / xa_f16_mask: 32 KB per 32 keys = 1 KB per key, tile-major per (layer, instance))
        const bool f16 = (p.xa_f16_mask >> j) & 1;
        a.K[j] = (f16 && (p.xa_opf & XA_K16)) ? c->w->k16[j].as<char>() + (size_t)l * rows * 1024 : c->w->kall_sp[j].as<char>() + (size_t)l * rows * ROWB;
        a.cb[j] = c->w->cb[j].as<float>() + (size_t)l * rows;
        a.VT[j] = (f16 && (p.xa_opf & XA_V16)) ? c->w->v16[j].as<char>() + (size_t)l * rows * 1024 : c->w->vt_all[j].as<char>() + (size_t)l * rows * ROWB;
        a.Sp[j] = p.Sp[j];
        const bool stat = (p.static_mask >> j) & 1;
        a.rs_off[j] = (unsigned)((size_t)(nl - l) * rows * 4);
        a.kb[j] = stat ? kb_now[j] + (size_t)l * CFD_D : c->w->zeros512.as<float>();
        a.kb_stride[j] = stat ? nl * CFD_D + 32 : 0;
        a.vb[j] = stat ? vb_now[j] + (size_t)l * CFD_D : c->w->zeros512.as<float>();
        a.vb_stride[j] = stat ? nl * CFD_D : 0;
      }
      a.d_step = rows_now ? nullptr : dstep;
      a.one_j = -1;
      if (p.xa_one >= 0) {   // (the work lists hold no segments for it: build_xattn_worklist)
        const int j = p.xa_one;
        const size_t rows = (size_t)p.U[j] * p.Sp[j];
        a.one_j = j; a.one_sp = p.Sp[j];
        a.one_va = c->w->xa_one_va.as<float>() + (size_t)l * p.U[j] * CFD_D;
        a.one_rs = c->w->cb[j].as<float>() + (size_t)nl * rows;
      }
      a.wgs = c->w->xa_wgs.as<XaWg>(); a.segs = c->w->xa_segs.as<XaSeg>();
#if XA_STAMP
      CHK(c->w->xa_stamps.ensure((size_t)p.xa_nwg * XA_WAVES * XA_NSTAMP * 8));
      a.stamps = c->w->xa_stamps.as<long long>();
#endif
      Bracket br(c, CFD_PROF_XATTN, st);
      if (p.att_fused) a.att = c->w->xa_att_desc.as<XaAtt>() + l;
      const int opf = p.att_fused ? 0 : p.xa_opf;
      auto launch_xa = [&](int nwg, const XAttnArgs& xa) {
        if (p.att_fused) hipLaunchKernelGGL((xattn_fused_kernel<true, 0>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 1) hipLaunchKernelGGL((xattn_fused_kernel<false, 1>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 2) hipLaunchKernelGGL((xattn_fused_kernel<false, 2>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else if (opf == 3) hipLaunchKernelGGL((xattn_fused_kernel<false, 3>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
        else hipLaunchKernelGGL((xattn_fused_kernel<false, 0>), dim3(nwg), dim3(XA_WAVES * 64), XA_LDS, st, xa);
      };
      if (l == 0 && share && p.xa0_nwg_a > 0) {   // layer-0 de-duplication (build_xattn_layer0_lists): the longest memory once per distinct (utterance, instance) ...
        XAttnArgs a0 = a;
        a0.wgs = c->w->xa0_wgs_a.as<XaWg>(); a0.segs = c->w->xa0_segs_a.as<XaSeg>(); a0.dd_out = c->w->xa_dedup.as<float>(); a0.stamps = nullptr;
        a0.one_j = -1;   // (the one-key memory belongs to the second launch)
        launch_xa(p.xa0_nwg_a, a0);
        HIPCHK(hipGetLastError());
        // ... then the other memories for every row, which also adds the stored results
        a0.wgs = c->w->xa0_wgs_b.as<XaWg>(); a0.segs = c->w->xa0_segs_b.as<XaSeg>(); a0.dd_out = nullptr; a0.dd_in = c->w->xa_dedup.as<float>();
        a0.one_j = a.one_j;
        launch_xa(p.xa0_nwg_b, a0);
        HIPCHK(hipGetLastError());
        if (c->prof) c->prof_n[CFD_PROF_XATTN] += 1;   // (two launches under one bracket)
      } else {
        launch_xa(p.xa_nwg, a);
        HIPCHK(hipGetLastError());
      }
    } else {
    CHK(ln(w.ln2g, w.ln2b, 0, 0, c->w->h_sp.as<char>(), M));
    // Three-launch path (att_mats wanted): scores against the folded keys of every memory.  Long memories and short
    // (<= 64 keys) memories go to different tile shapes; rows in a shared-memory run of the largest memory use one
    // un-batched product per run.
    const bool runs = p.nruns > 0 && row0 == 0 && Be == p.Be;
    auto scores_grouped = [&](bool small, int skip_j, const int* brow, int nb) -> int {
      GemmArgs a = gemm_args();
      EpiF32 e;
      memset(&e, 0, sizeof(e));
      int n = 0;
      for (int j = 0; j < CFD_NMEM; ++j) {
        if ((p.Sp[j] <= 64) != small || j == skip_j) continue;
        a.X[n] = c->w->kall_sp[j].as<char>() + (size_t)l * p.U[j] * p.Sp[j] * ROWB; a.ldx[n] = ROWB;
        a.xbs[n] = (long long)p.Sp[j] * ROWB; a.xmap[n] = mapj[j];
        a.I[n] = p.Sp[j]; a.Iclamp[n] = p.Sp[j]; a.kt[n] = CFD_D / 32;
        e.goff[n] = p.off[j]; e.gbias[n] = c->w->cb[j].as<float>() + (size_t)l * p.U[j] * p.Sp[j]; e.gmap[n] = mapj[j]; e.gstride[n] = p.Sp[j];
        ++n;
      }
      if (!n || nb <= 0) return CFD_OK;
      a.nslot = n; a.brow = brow;
      a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.ybs = (long long)L * ROWB; a.J = L; a.Jclamp = L;
      e.out = c->w->sc.as<float>(); e.ldo = p.Sp_tot; e.obs = (long long)L * p.Sp_tot;
      return run_gemm<MODE_GROUPED>(c, CFD_PROF_GEMM_ATTN, a, e, nb, 1, st);
    };
    CHK(scores_grouped(true, -1, nullptr, Be));   // short memories
    if (!runs) {
      CHK(scores_grouped(false, -1, nullptr, Be));
    } else {
      CHK(scores_grouped(false, -1, c->w->short_rows.as<int>(), p.nshort));
      CHK(scores_grouped(false, p.jbig, c->w->long_rows.as<int>(), p.nlong));
      const int j = p.jbig;
      for (int r = 0; r < p.nruns; ++r) {
        GemmArgs a = gemm_args();
        a.X[0] = c->w->kall_sp[j].as<char>() + ((size_t)l * p.U[j] + p.run_u[r]) * p.Sp[j] * ROWB; a.ldx[0] = ROWB;
        a.I[0] = p.Sp[j]; a.Iclamp[0] = p.Sp[j]; a.kt[0] = CFD_D / 32;
        a.Y = c->w->h_sp.as<char>() + (size_t)p.run_row0[r] * L * ROWB; a.ldy = ROWB; a.J = p.run_len[r] * L; a.Jclamp = a.J;
        EpiF32 e;
        memset(&e, 0, sizeof(e));
        e.out = c->w->sc.as<float>() + (size_t)p.run_row0[r] * L * p.Sp_tot; e.ldo = p.Sp_tot; e.goff[0] = p.off[j];
        e.gbias[0] = c->w->cb[j].as<float>() + ((size_t)l * p.U[j] + p.run_u[r]) * p.Sp[j]; e.gstride[0] = 0;
        CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_ATTN, a, e, 1, 1, st)));
      }
    }
    {
      SoftmaxArgs a;
      memset(&a, 0, sizeof(a));
      a.sc = c->w->sc.as<float>(); a.P = c->w->p_sp.as<char>(); a.ld = p.Sp_tot; a.rows = M; a.rows_per_b = L; a.nseg = CFD_NMEM;
      for (int j = 0; j < CFD_NMEM; ++j) {
        a.off[j] = p.off[j]; a.S[j] = p.S[j]; a.Sp[j] = p.Sp[j]; a.mask[j] = p.mask[j]; a.has_mask[j] = p.has_mask[j]; a.map[j] = mapj[j]; a.att[j] = attj[j];
      }
      a.layer = l; a.nl = nl;
      LAUNCH(CFD_PROF_ROWS, softmax_rows_kernel, dim3((unsigned)((M + 3) / 4)), blk, st, a);
    }
    // x += sum_j P_j . VV_j(n_j) + folded bias
    auto pv_segk = [&](int skip_j, const int* brow, int nb) -> int {
      if (nb <= 0) return CFD_OK;
      GemmArgs a = gemm_args();
      int n = 0;
      for (int j = 0; j < CFD_NMEM; ++j) {
        if (j == skip_j) continue;
        a.X[n] = c->w->vt_all[j].as<char>() + (size_t)l * p.U[j] * CFD_D * p.Sp[j] * 4; a.ldx[n] = (long long)p.Sp[j] * 4;
        a.xbs[n] = (long long)CFD_D * p.Sp[j] * 4; a.xmap[n] = mapj[j];
        a.kt[n] = p.Sp[j] / 32; a.yk0[n] = p.off[j] / 32;
        a.I[n] = CFD_D; a.Iclamp[n] = CFD_D;
        ++n;
      }
      a.nslot = n; a.brow = brow;
      a.Y = c->w->p_sp.as<char>(); a.ldy = (long long)p.Sp_tot * 4; a.ybs = (long long)L * p.Sp_tot * 4; a.J = L; a.Jclamp = L;
      EpiResid e{c->w->x.as<float>(), (long long)L * CFD_D, w.cross_bias.as<float>()};
      return run_gemm<MODE_SEGK>(c, CFD_PROF_GEMM_ATTN, a, e, nb, 1, st);
    };
    if (!runs) {
      CHK(pv_segk(-1, nullptr, Be));
    } else {
      CHK(pv_segk(-1, c->w->short_rows.as<int>(), p.nshort));   // short rows: all segments
      CHK(pv_segk(p.jbig, c->w->long_rows.as<int>(), p.nlong));   // long rows: short segments first ...
      const int j = p.jbig;
      for (int r = 0; r < p.nruns; ++r) {   // ... then the shared audio memory, run by run (disjoint rows)
        GemmArgs a = gemm_args();
        a.X[0] = c->w->vt_all[j].as<char>() + ((size_t)l * p.U[j] + p.run_u[r]) * CFD_D * p.Sp[j] * 4; a.ldx[0] = (long long)p.Sp[j] * 4;
        a.I[0] = CFD_D; a.Iclamp[0] = CFD_D; a.kt[0] = p.Sp[j] / 32;
        a.Y = c->w->p_sp.as<char>() + (size_t)p.run_row0[r] * L * p.Sp_tot * 4 + (size_t)(p.off[j] / 32) * 128;
        a.ldy = (long long)p.Sp_tot * 4; a.J = p.run_len[r] * L; a.Jclamp = a.J;
        EpiResid e{c->w->x.as<float>() + (size_t)p.run_row0[r] * L * CFD_D, 0, nullptr};
        CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_ATTN, a, e, 1, 1, st)));
      }
    }
    }
    if (c->stop_stage == 4 + 4 * l) return CFD_OK;
    // ---- f. time block 2                                                        (:655)
    CHK(ln(w.tb2g, w.tb2b, 1, 2 * l + 1, c->w->h_sp.as<char>(), M));
    // time block 2's projection + residual, then norm3                           (:655, :659)
    CHK(token_gemm_resid_ln(w.wtb2_sp, CFD_D, c->w->h_sp.as<char>(), w.btb2, M, w.ln3g, w.ln3b, 0, 0));
    // ---- g. FFN                                                                 (:659-661)
    {
      GemmArgs a = gemm_args();
      a.X[0] = w.w1_sp.as<char>(); a.ldx[0] = ROWB; a.I[0] = CFD_FF; a.Iclamp[0] = CFD_FF; a.kt[0] = CFD_D / 32;
      a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.J = (int)M; a.Jclamp = (int)M;
      EpiSplit e{c->w->u_sp.as<char>(), (long long)CFD_FF * 4, 0, 0, w.b1, 1, 0};
      CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
    }
    // second FFN product + residual, then the next layer's norm1 (or the decoder's final norm)   (:661, :568; :238-239)
    {
      const float* ng = l + 1 < nl ? c->lw[l + 1].ln1g : rawp(c, "decoder.norm.weight");
      const float* nb = l + 1 < nl ? c->lw[l + 1].ln1b : rawp(c, "decoder.norm.bias");
      if (c->stop_stage == 5 + 4 * l) return token_gemm_resid(w.w2_sp, CFD_FF, c->w->u_sp.as<char>(), w.b2, M);
      CHK(token_gemm_resid_ln(w.w2_sp, CFD_FF, c->w->u_sp.as<char>(), w.b2, M, ng, nb, 0, 0));
      h_ready = true;
    }
  }
  // 7. final norm (made above) + latent projection                                (cross_attention.py:238-239, denoiser.py:382)
  {
    GemmArgs a = gemm_args();
    a.X[0] = c->wp_sp.as<char>(); a.ldx[0] = ROWB; a.I[0] = CFD_LAT; a.Iclamp[0] = CFD_LAT; a.kt[0] = CFD_D / 32;
    a.Y = c->w->h_sp.as<char>(); a.ldy = ROWB; a.J = (int)M; a.Jclamp = (int)M;
    EpiF32 e;
    memset(&e, 0, sizeof(e));
    e.out = eps_out; e.ldo = CFD_LAT; e.bias = rawp(c, "latent_proj.bias");
    CHK((run_gemm<MODE_PLAIN>(c, CFD_PROF_GEMM_TOKEN, a, e, 1, 1, st)));
  }
  if (p.att_fused && fused_x) {   // this step's maps: from what the nine cross-attention launches kept, into slot *d_step of the ring
    XaFixArgs f;
    memset(&f, 0, sizeof(f));
    f.att = c->w->xa_att_desc.as<XaAtt>(); f.nl = nl; f.L = L; f.one_j = p.xa_one; f.d_step = c->w->d_step.as<int>();
    for (int j = 0; j < CFD_NMEM; ++j) { f.S[j] = p.S[j]; f.ring[j] = p.att[j]; f.slot[j] = p.att_slot[j]; }
    LAUNCH(CFD_PROF_ROWS, att_fixup_kernel, dim3((unsigned)(p.att_nb * L), nl), dim3(256), st, f);
  }
  return CFD_OK;
}

// ---- cfd_forward ---------------------------------------------------------------------------------------
extern "C" int cfd_forward(cfd_handle c, const float* sample, int Be, int L, const int32_t* timesteps, int n_t,
                           const cfd_memory mem[CFD_NUM_MEM], float* out, float* const att[CFD_NUM_MEM], void* stream) {
  if (!c) return fail(CFD_E_ARG, "null argument");
  c->hint_now = c->hint_same_mem;   // the promise covers THIS call only, however it ends
  c->hint_same_mem = false;
  if (!sample || !timesteps || !mem || !out) return fail(CFD_E_ARG, "null argument");
  if (c->run_open) return fail(CFD_E_STATE, "a sampling run is open on this handle");
  HIPCHK(hipSetDevice(c->cfg.device));
  CHK(settle_deferred_census(c));
  if (n_t != 1 && n_t != Be) return fail(CFD_E_ARG, "n_t must be 1 or Be");
  hipStream_t st = (hipStream_t)stream;
  const int tmode = (n_t == 1) ? 0 : 1;
  CHK(setup_problem(c, Be, L, mem, att, tmode, n_t));
  CHK(sat_begin(c, st));
  // tmode 0 reads table row d_step[0] which must be 0 outside a sampling run
  HIPCHK(hipMemsetAsync(c->w->d_step.p, 0, 16, st));
  CHK(build_time_tables(c, timesteps, n_t, st));
  // (the sample is split in front of the once-per-call projections, so that ONE wait reads the census of both)
  const long long n = c->w->pb.M * (CFD_LAT / 8);
  LAUNCH(CFD_PROF_OTHER, to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), st, sample, c->w->sample_sp.as<char>(), c->w->pb.M,
         CFD_LAT, (long long)CFD_LAT, (long long)CFD_LAT * 4, c->sat_in());
  {
    bool want_att = false;
    for (int j = 0; j < CFD_NMEM; ++j) want_att = want_att || (att && att[j]);
    const bool reuse = c->hint_now && c->w->pb.prev_same;
    CHK(prepare_static_memside(c, st, 0, want_att && !c->w->pb.att_fused, reuse));
    if (c->w->pb.static_mask) {   // once-per-call projections of the caller's memories: the census is read before they are used
      HIPCHK(hipStreamSynchronize(st));
      CHK(check_saturation(c, "cfd_forward (sample, memories / their projections)"));
    }
  }
  c->memside_in_forward = false;
  CHK(enqueue_denoise(c, st));
  HIPCHK(hipMemcpyAsync(out, c->w->eps.p, (size_t)c->w->pb.M * CFD_LAT * 4, hipMemcpyDeviceToDevice, st));
  if (c->memside_in_forward || !c->w->pb.static_mask) {
    // Paths whose memory-side projections run INSIDE the forward (per-row timesteps, att_mats on the tile kernels, CFD_HOIST_MEMSIDE=0,
    // the three-launch cross-attention): their census -- and the sample's, which no earlier wait has read on these paths -- is read here,
    // so that a clamped projection fails THIS call instead of the next one (on these paths the call therefore returns with `out` complete).
    HIPCHK(hipStreamSynchronize(st));
    c->memside_in_forward = false;
    CHK(check_saturation(c, "cfd_forward (sample, memories / their per-call projections)"));
  }
  {   // what the next call may reuse (cfd_forward_same_memories): all five memories' timestep-independent projections are in the workspace
    Work* w = c->w;
    const Problem& p = w->pb;
    w->fwd_mem_valid = p.tmode == 0 && p.static_mask == (1 << CFD_NMEM) - 1;
    w->fwd_wver = (unsigned long long)c->wver;
    w->fwd_Be = p.Be;
    w->fwd_L = p.L;
    w->fwd_att = false;
    for (int j = 0; j < CFD_NMEM; ++j) w->fwd_att = w->fwd_att || (att && att[j]);
    for (int j = 0; j < CFD_NMEM; ++j) {
      w->fwd_U[j] = p.U[j]; w->fwd_S[j] = p.S[j]; w->fwd_mask[j] = mem[j].key_padding_mask != nullptr; w->fwd_map[j] = mem[j].row_map != nullptr;
    }
  }
  return CFD_OK;
}

extern "C" int cfd_forward_same_memories(cfd_handle c) {
  if (!c) return fail(CFD_E_ARG, "null handle");
  c->hint_same_mem = true;
  return CFD_OK;
}

extern "C" int cfd_profile_forward(cfd_handle c, float ms[CFD_PROF_NCLASS], int launches[CFD_PROF_NCLASS]) {
  if (!c || !ms || !launches) return fail(CFD_E_ARG, "null argument");
  if (c->w->pb.Be == 0) return fail(CFD_E_STATE, "no problem configured (call cfd_forward or cfd_sample_begin first)");
  HIPCHK(hipSetDevice(c->cfg.device));
  if (c->run_open && c->run_pos >= c->run_iters)
    return fail(CFD_E_STATE, "sampling run is complete; profile before the last iteration");
  hipStream_t st = c->run_open ? c->run_stream : nullptr;
  HIPCHK(hipStreamSynchronize(st));
  for (int k = 0; k < CFD_PROF_NCLASS; ++k) { c->prof_ms[k] = 0.f; c->prof_n[k] = 0; }
  c->prof = true;
  int r = enqueue_denoise(c, st);
  c->prof = false;
  HIPCHK(hipStreamSynchronize(st));
  for (int k = 0; k < CFD_PROF_NCLASS; ++k) { ms[k] = c->prof_ms[k]; launches[k] = c->prof_n[k]; }
  return r;
}

// ---- scheduler coefficients: diffusers 0.14.0 DDPMScheduler.step / DDIMScheduler.step, float32 ----------
static void ddpm_coef(const float* ac, int T, int n_inf, int t, StepCoef* o) {
  const int prev_t = t - T / n_inf;
  const float ap_t = ac[t];
  const float ap_prev = prev_t >= 0 ? ac[prev_t] : 1.0f;
  const float bp_t = 1.0f - ap_t, bp_prev = 1.0f - ap_prev;
  const float cur_alpha = ap_t / ap_prev;
  const float cur_beta = 1.0f - cur_alpha;
  o->sb = sqrtf(bp_t);
  o->sa = sqrtf(ap_t);
  o->c0 = (sqrtf(ap_prev) * cur_beta) / bp_t;
  o->cx = sqrtf(cur_alpha) * bp_prev / bp_t;
  float var = bp_prev / bp_t * cur_beta;
  if (var < 1e-20f) var = 1e-20f;
  o->sigma = t > 0 ? sqrtf(var) : 0.0f;
  o->use_noise = t > 0 ? 1.0f : 0.0f;
  o->pad0 = o->pad1 = 0.f;
}
static void ddim_coef(const float* ac, int T, int n_inf, int t, float eta, int set_alpha_to_one, StepCoef* o) {
  const int prev_t = t - T / n_inf;
  const float ap_t = ac[t];
  const float ap_prev = prev_t >= 0 ? ac[prev_t] : (set_alpha_to_one ? 1.0f : ac[0]);
  const float bp_t = 1.0f - ap_t, bp_prev = 1.0f - ap_prev;
  const float var = (bp_prev / bp_t) * (1.0f - ap_t / ap_prev);
  const float std = eta * sqrtf(var);
  o->sb = sqrtf(bp_t);
  o->sa = sqrtf(ap_t);
  o->c0 = sqrtf(ap_prev);
  o->cx = sqrtf(1.0f - ap_prev - std * std);
  o->sigma = std;
  o->use_noise = eta > 0.f ? 1.0f : 0.0f;
  o->pad0 = o->pad1 = 0.f;
}

static int enqueue_loop_iteration(Ctx* c, hipStream_t st) {
  const cfd_sample_args& s = c->sargs;
  const long long n8 = (long long)s.B * s.L * (CFD_LAT / 8);
  BeginArgs ba{c->latents.as<float>(), c->w->sample_sp.as<char>(), s.B, s.L, s.G, s.preseq, c->inoise.as<float>(), s.preseq_len,
               c->coef.as<StepCoef>(), c->w->d_step.as<int>()};
  LAUNCH(CFD_PROF_OTHER, begin_step_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), st, ba);
  CHK(enqueue_denoise(c, st));
  CfgStepArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.eps = c->w->eps.as<float>(); ca.latents = c->latents.as<float>(); ca.B = s.B; ca.L = s.L; ca.G = s.G;
  for (int k = 0; k < 8; ++k) { ca.w[k] = s.guidance_weight[k]; ca.pos[k] = c->chunk_pos[k]; }
  ca.kind = s.scheduler; ca.clip = s.clip_sample; ca.coef = c->coef.as<StepCoef>(); ca.d_step = c->w->d_step.as<int>();
  ca.noise = s.step_noise; ca.seed = s.seed; ca.utt0 = s.first_utterance;
  const long long n4 = (long long)s.B * s.L * CFD_LAT / 4;
  ca.advance = c->w->d_step.as<int>();   // the last workgroup of cfg_step_kernel advances the loop index
  LAUNCH(CFD_PROF_OTHER, cfg_step_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 256)), dim3(256), st, ca);
  return CFD_OK;
}

extern "C" int cfd_sample_begin(cfd_handle c, const cfd_sample_args* args, void* stream) {
  if (!c || !args) return fail(CFD_E_ARG, "null argument");
  if (c->run_open) return fail(CFD_E_STATE, "a sampling run is already open");
  HIPCHK(hipSetDevice(c->cfg.device));
  c->hint_now = c->hint_same_mem = false;
  CHK(settle_deferred_census(c));
  const cfd_sample_args& s = *args;
  if (s.B < 1 || (s.G != 1 && s.G != 7 && (s.G < 1 || s.G > 8))) return fail(CFD_E_ARG, "bad B / G");
  if (s.scheduler != 0 && s.scheduler != 1) return fail(CFD_E_ARG, "scheduler must be 0 (DDPM) or 1 (DDIM)");
  if (!s.alphas_cumprod || s.num_train_timesteps < 1 || s.num_inference_steps < 1 || s.num_inference_steps > s.num_train_timesteps)
    return fail(CFD_E_ARG, "bad scheduler tables");
  if (s.timesteps && (s.num_timesteps < 1 || s.num_timesteps > s.num_train_timesteps)) return fail(CFD_E_ARG, "bad num_timesteps");
  if (!s.timesteps && s.scheduler == 0 && s.num_train_timesteps % s.num_inference_steps)
    return fail(CFD_E_ARG, "DDPM: num_inference_steps = %d does not divide num_train_timesteps = %d: the loop's timestep table for such "
                           "counts differs between diffusers releases (unpinned); pass the scheduler's table in cfd_sample_args.timesteps",
                s.num_inference_steps, s.num_train_timesteps);
  if (s.preseq && (s.preseq_len < 1 || s.preseq_len > s.L)) return fail(CFD_E_ARG, "bad preseq_len");
  hipStream_t st = (hipStream_t)stream;
  c->sargs = s;
  c->run_stream = st;
  c->setup_launches = 0;
  int n_ring = 0;
  for (int j = 0; j < CFD_NMEM; ++j) n_ring += s.att_ring[j] != nullptr;
  if (n_ring != 0 && n_ring != CFD_NMEM) return fail(CFD_E_ARG, "att_ring: give all five buffers or none");
  if (n_ring && s.skip_zero_weight_chunks && s.G > 1 && s.guidance_weight[s.G - 1] == 0.0f)
    return fail(CFD_E_ARG, "att_ring keeps the maps of the LAST guidance chunk: it must be evaluated (skip_zero_weight_chunks = 0)");
  if (s.skip_zero_weight_chunks)   // chunk-major batch: dropping trailing chunks = using the first G' * B rows
    while (c->sargs.G > 1 && s.guidance_weight[c->sargs.G - 1] == 0.0f) c->sargs.G -= 1;
  // N = loop iterations (the length of scheduler.timesteps); n_inf = the count given to set_timesteps, which fixes the
  // stride `prev_t = t - T // n_inf` of the step formulas.  They differ only for a caller-supplied table.
  const int Be = c->sargs.G * s.B, n_inf = s.num_inference_steps, N = s.timesteps ? s.num_timesteps : n_inf, T = s.num_train_timesteps;
  c->sargs.timesteps = nullptr;   // (host pointer: not kept beyond this call)
  c->run_iters = N;
  for (int k = 0; k < 8; ++k) c->chunk_pos[k] = k;
  cfd_memory mem_in[CFD_NMEM];
  for (int j = 0; j < CFD_NMEM; ++j) mem_in[j] = s.mem[j];
  {
    // chunk permutation (see chunk_pos): group the chunks that use one shared copy of the largest memory
    const int G = c->sargs.G, B = s.B;
    bool all_maps = c->permute && G > 2;
    int jb = 0;
    for (int j = 0; j < CFD_NMEM; ++j) {
      if (!s.mem[j].row_map) all_maps = false;
      if (s.mem[j].S > s.mem[jb].S) jb = j;
    }
    if (all_maps) {
      std::vector<int> hm(Be);
      HIPCHK(hipMemcpy(hm.data(), s.mem[jb].row_map, (size_t)Be * 4, hipMemcpyDeviceToHost));
      std::vector<int> key(G);   // the shared memory index of a uniform chunk, or -1
      for (int g = 0; g < G; ++g) {
        key[g] = hm[(size_t)g * B];
        for (int u = 1; u < B; ++u)
          if (hm[(size_t)g * B + u] != key[g]) { key[g] = -1; break; }
      }
      std::vector<int> order;   // order[position] = original chunk: uniform chunks grouped by key, first occurrence first
      std::vector<char> used(G, 0);
      for (int g = 0; g < G; ++g) {
        if (used[g] || key[g] < 0) continue;
        for (int h = g; h < G; ++h)
          if (!used[h] && key[h] == key[g]) { order.push_back(h); used[h] = 1; }
      }
      for (int g = 0; g < G; ++g)
        if (!used[g]) order.push_back(g);
      bool ident = true;
      for (int pnum = 0; pnum < G; ++pnum) ident = ident && order[pnum] == pnum;
      if (!ident) {
        for (int pnum = 0; pnum < G; ++pnum) c->chunk_pos[order[pnum]] = pnum;
        std::vector<int> pm(Be);
        for (int j = 0; j < CFD_NMEM; ++j) {
          HIPCHK(hipMemcpy(hm.data(), s.mem[j].row_map, (size_t)Be * 4, hipMemcpyDeviceToHost));
          for (int pnum = 0; pnum < G; ++pnum)
            for (int u = 0; u < B; ++u) pm[(size_t)pnum * B + u] = hm[(size_t)order[pnum] * B + u];
          CHK(c->perm_map[j].ensure((size_t)Be * 4));
          HIPCHK(hipMemcpy(c->perm_map[j].p, pm.data(), (size_t)Be * 4, hipMemcpyHostToDevice));
          mem_in[j].row_map = c->perm_map[j].as<int32_t>();
        }
      }
    }
  }
  // operand policy of the run (cfd_sample_args::operand_policy): single-fp16 key / value tiles of the long memories for the fused
  // cross-attention kernel -- only where that kernel runs on projections made once per run and keeps no maps
  c->want_opf = (n_ring || s.dynamic_memory_mask) ? 0 : (c->xa_operands >= 0 ? c->xa_operands : (s.operand_policy & 3));
  const int r_setup = setup_problem(c, Be, s.L, mem_in, nullptr, 0, N);
  c->want_opf = 0;
  CHK(r_setup);
  if (c->share0 && c->sargs.G > 1) c->w->pb.share_B = s.B;   // begin_step_kernel writes G identical copies of the B rows
  c->w->pb.att_nb = 0;
  if (n_ring) {
    // The reference keeps att_mats of the full-conditioning chunk of EVERY iteration (convofusion.py:517-523).  On the row-tile path the
    // second cross-attention launch has the probabilities in registers anyway: the rows of the last chunk store them into slot *d_step
    // of the caller's ring, inside the captured iteration -- no second forward, no host round trip.
    Problem& pb = c->w->pb;
    // ... and on the tile kernels the fused cross-attention kernel has them in its softmax: its ATT instance keeps them, att_fixup_kernel
    // normalises them once per step (xattn_fused.hpp, XaAtt).  What cannot keep them: a run without the fused kernel (memories made per
    // step: dynamic memories; the developer switches that turn it off).
    const bool fused_ok = c->fused_xattn && pb.xa_nwg > 0 && c->hoist_memside && !g_cfd_naive_gemm;
    if ((!pb.rt && !fused_ok) || s.dynamic_memory_mask)
      return fail(CFD_E_SHAPE, "att_ring needs the row-tile path or the fused cross-attention kernel (one timestep per step, no dynamic memory): "
                               "this run has L = %d, %lld token rows; take the maps with one forward per iteration instead", s.L, (long long)Be * s.L);
    pb.att_b0 = c->chunk_pos[c->sargs.G - 1] * s.B;
    pb.att_nb = s.B;
    for (int j = 0; j < CFD_NMEM; ++j) {
      pb.att[j] = s.att_ring[j];
      pb.att_slot[j] = (long long)s.B * c->nl * s.L * pb.S[j];
    }
    if (!pb.rt) {
      pb.att_fused = true;
      CHK(setup_att_fused(c));
      CHK(build_xattn_worklist(c, mem_in));   // (once more: the list now says which tiles keep their maps)
      if (pb.xa_nwg <= 0) return fail(CFD_E_SHAPE, "att_ring: the fused cross-attention work list is empty");
    }
  }
  CHK(build_xattn_layer0_lists(c, mem_in));
  {   // (the rest of the operand policy's conditions; prepare_static_memside checks that every memory's projections are made once per run)
    Problem& pb = c->w->pb;
    const bool fused_run = !pb.rt && c->fused_xattn && pb.xa_nwg > 0 && c->hoist_memside && !g_cfd_naive_gemm && !pb.att_fused && !s.dynamic_memory_mask;
    if (!fused_run) pb.xa_opf = 0;
  }
  // timesteps: (arange(N) * (T // N)).round()[::-1] (+ steps_offset for DDIM)
  std::vector<int32_t> ts(N);
  std::vector<StepCoef> coef(N);
  const int ratio = T / n_inf;
  for (int i = 0; i < N; ++i) {
    int t = s.timesteps ? s.timesteps[i] : (N - 1 - i) * ratio + (s.scheduler == 1 ? s.steps_offset : 0);
    if (t < 0 || t >= T) return fail(CFD_E_ARG, "timestep %d out of range", t);
    ts[i] = t;
    if (s.scheduler == 0) ddpm_coef(s.alphas_cumprod, T, n_inf, t, &coef[i]);
    else ddim_coef(s.alphas_cumprod, T, n_inf, t, s.eta, s.set_alpha_to_one, &coef[i]);
  }
  CHK(c->coef.ensure((size_t)N * sizeof(StepCoef)));
  HIPCHK(hipMemcpyAsync(c->coef.p, coef.data(), (size_t)N * sizeof(StepCoef), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(c->w->d_step.p, 0, 16, st));
  CHK(sat_begin(c, st));    // the census this call reads below counts ITS launches only
  CHK(build_time_tables(c, ts.data(), N, st));
  CHK(prepare_static_memside(c, st, s.dynamic_memory_mask, false));
  HIPCHK(hipStreamSynchronize(st));  // ts / coef host vectors go out of scope
  CHK(check_saturation(c, "cfd_sample_begin (memories / their once-per-run projections)"));
  const size_t lat_bytes = (size_t)s.B * s.L * CFD_LAT * 4;
  CHK(c->latents.ensure(lat_bytes));
  if (s.init_latents) {
    HIPCHK(hipMemcpyAsync(c->latents.p, s.init_latents, lat_bytes, hipMemcpyDeviceToDevice, st));
  } else {
    const long long n = (long long)s.B * s.L * CFD_LAT / 4;
    LAUNCH(CFD_PROF_OTHER, philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), st, c->latents.as<float>(), s.B,
           s.L * CFD_LAT, (uint64_t)s.seed, 0u, s.first_utterance, 1u, 1.0f);
  }
  CHK(c->inoise.ensure(s.preseq ? (size_t)s.B * s.preseq_len * CFD_LAT * 4 : 16));
  if (s.preseq) {
    HIPCHK(hipMemcpy2DAsync(c->inoise.p, (size_t)s.preseq_len * CFD_LAT * 4, c->latents.p, (size_t)s.L * CFD_LAT * 4,
                            (size_t)s.preseq_len * CFD_LAT * 4, s.B, hipMemcpyDeviceToDevice, st));
  }
  // capture one loop iteration
  if (c->gexec) { (void)hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
  if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
  // eager warm-up of every kernel variant (sets function attributes outside capture); the iteration is
  // idempotent on the workspace and we restore the state it mutates (latents, in-paint noise, step index).
  {
    DBuf save_lat, save_in;
    CHK(save_lat.ensure(lat_bytes));
    HIPCHK(hipMemcpyAsync(save_lat.p, c->latents.p, lat_bytes, hipMemcpyDeviceToDevice, st));
    if (s.preseq) {
      CHK(save_in.ensure(c->inoise.bytes));
      HIPCHK(hipMemcpyAsync(save_in.p, c->inoise.p, c->inoise.bytes, hipMemcpyDeviceToDevice, st));
    }
    int r = enqueue_loop_iteration(c, st);
    if (r != CFD_OK) return r;
    HIPCHK(hipMemcpyAsync(c->latents.p, save_lat.p, lat_bytes, hipMemcpyDeviceToDevice, st));
    if (s.preseq) HIPCHK(hipMemcpyAsync(c->inoise.p, save_in.p, c->inoise.bytes, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(c->w->d_step.p, 0, 16, st));
    HIPCHK(hipStreamSynchronize(st));
    save_lat.release();
    save_in.release();
  }
  // capture and replay on the handle's own stream (the legacy default stream cannot be captured); all
  // set-up work above was enqueued on the caller's stream and has been waited for.
  hipStream_t cap = c->own_stream;
  c->run_stream = cap;
  c->memside_in_forward = false;
  HIPCHK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  int r = enqueue_loop_iteration(c, cap);
  c->run_counts = c->memside_in_forward;   // (the hoisted / row-tile iteration has no counting launch: cfd_sample_read then skips the census read)
  c->memside_in_forward = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(cap, &g);
  if (r != CFD_OK) { if (g) (void)hipGraphDestroy(g); return r; }
  if (e != hipSuccess) return fail(CFD_E_HIP, "stream capture failed: %s", hipGetErrorString(e));
  c->graph = g;
  HIPCHK(hipGraphInstantiate(&c->gexec, c->graph, nullptr, nullptr, 0));
  c->run_open = true;
  c->run_pos = 0;
  return CFD_OK;
}

extern "C" int cfd_sample_steps(cfd_handle c, int n) {
  if (!c) return fail(CFD_E_ARG, "null handle");
  if (!c->run_open) return fail(CFD_E_STATE, "no sampling run open");
  if (n < 0 || c->run_pos + n > c->run_iters)
    return fail(CFD_E_ARG, "run has %d of %d iterations done; cannot run %d more", c->run_pos, c->run_iters, n);
  HIPCHK(hipSetDevice(c->cfg.device));
  for (int i = 0; i < n; ++i) HIPCHK(hipGraphLaunch(c->gexec, c->run_stream));
  c->run_pos += n;
  return CFD_OK;
}

extern "C" int cfd_dyadic_steps(cfd_handle a, cfd_handle b, const cfd_dyadic_proj* pr, int n) {
  if (!a || !pr || a == b) return fail(CFD_E_ARG, "side A's handle, the projection and (two-handle form) a distinct side B handle are needed");
  if (!a->run_open || (b && !b->run_open)) return fail(CFD_E_STATE, "both sides need an open sampling run");
  if (!pr->w1 || !pr->b1 || !pr->w2 || !pr->b2 || !pr->spk_a || !pr->spk_b || !pr->tmp || pr->hidden < 1 || pr->out_dim != CFD_D)
    return fail(CFD_E_ARG, "bad partner projection");
  const cfd_sample_args& sa = a->sargs;
  if (!(sa.dynamic_memory_mask & 1) || (b && !(b->sargs.dynamic_memory_mask & 1)))
    return fail(CFD_E_STATE, "the speaker memory of the run(s) must be declared dynamic");
  if (b && (sa.B != b->sargs.B || sa.L != b->sargs.L || a->cfg.device != b->cfg.device)) return fail(CFD_E_ARG, "the two sides differ in batch, length or device");
  if (!b && sa.B % 2) return fail(CFD_E_ARG, "merged form: the run holds side A's utterances followed by side B's (even batch)");
  if (n < 0 || a->run_pos + n > a->run_iters || (b && b->run_pos + n > b->run_iters))
    return fail(CFD_E_ARG, "run has %d of %d iterations done; cannot run %d more", a->run_pos, a->run_iters, n);
  HIPCHK(hipSetDevice(a->cfg.device));
  hipStream_t st = a->run_stream;
  if (b) {
    // side B's stream may still hold its set-up or an earlier read: everything below is ordered behind it, and side B's later reads
    // behind everything below (events, no host wait)
    HIPCHK(hipEventRecord(b->weg_ev, b->run_stream));
    HIPCHK(hipStreamWaitEvent(st, b->weg_ev, 0));
  }
  const int Bs = b ? sa.B : sa.B / 2;                  // utterances per side
  const long long rows = (long long)Bs * sa.L;
  const dim3 blk(256);
  const long long gy = (rows + 31) / 32;
  if (gy > 65535) return fail(CFD_E_ARG, "too many rows for one launch (%lld)", rows);
  auto project = [&](const float* lat, float* spk) {
    hipLaunchKernelGGL(linear_act_kernel, dim3((unsigned)((pr->hidden + 63) / 64), (unsigned)gy), blk, 0, st, lat, rows, CFD_LAT, pr->w1, pr->b1, pr->hidden, 1, pr->tmp);
    hipLaunchKernelGGL(linear_act_kernel, dim3((unsigned)((pr->out_dim + 63) / 64), (unsigned)gy), blk, 0, st, (const float*)pr->tmp, rows, pr->hidden, pr->w2, pr->b2,
                       pr->out_dim, 1, spk);
  };
  const float* lat_a = a->latents.as<float>();
  const float* lat_b = b ? b->latents.as<float>() : lat_a + rows * CFD_LAT;
  for (int i = 0; i < n; ++i) {
    project(lat_b, pr->spk_a);                         // A attends to B's latents as they stand at the start of the iteration ...
    project(lat_a, pr->spk_b);                         // ... and B to A's
    HIPCHK(hipGetLastError());
    HIPCHK(hipGraphLaunch(a->gexec, st));
    if (b) HIPCHK(hipGraphLaunch(b->gexec, st));       // same queue, one after the other: no two-queue overlap
  }
  a->run_pos += n;
  if (b) {
    b->run_pos += n;
    HIPCHK(hipEventRecord(a->weg_ev, st));
    HIPCHK(hipStreamWaitEvent(b->run_stream, a->weg_ev, 0));
  }
  return CFD_OK;
}

extern "C" int cfd_sample_position(cfd_handle c) { return (c && c->run_open) ? c->run_pos : -1; }

extern "C" int cfd_sample_read(cfd_handle c, float* out, int close) {
  if (!c || !out) return fail(CFD_E_ARG, "null argument");
  if (!c->run_open) return fail(CFD_E_STATE, "no sampling run open");
  HIPCHK(hipSetDevice(c->cfg.device));
  const size_t lat_bytes = (size_t)c->sargs.B * c->sargs.L * CFD_LAT * 4;
  HIPCHK(hipMemcpyAsync(out, c->latents.p, lat_bytes, hipMemcpyDeviceToDevice, c->run_stream));
  HIPCHK(hipStreamSynchronize(c->run_stream));
  CHK(settle_deferred_census(c));
  // the census of everything the run's iterations counted (per-step projections of a dynamic memory, CFD_HOIST_MEMSIDE=0): read on
  // every read of a run whose captured iteration has such launches, BEFORE the run is closed -- a run that fails here stays open and can be
  // inspected or closed by the caller
  if (c->run_counts) CHK(check_saturation(c, "sampling run (the per-step projections of a memory)"));
  if (close) c->run_open = false;
  return CFD_OK;
}

// ---- stand-alone scheduler ops ----------------------------------------------------------------------------
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
__global__ void add_noise_kernel(const float* x0, const float* noise, float* out, size_t n, float sa, float sb) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sa * x0[i] + sb * noise[i];
}

extern "C" int cfd_scheduler_step(cfd_handle c, int scheduler, const float* ac, int T, int n_inf, int t, int clip, float eta,
                                  int set_alpha_to_one, const float* model_output, const float* noise, float* sample_inout,
                                  size_t numel, float* pred_original_sample, void* stream) {
  if (!c || !ac || !model_output || !sample_inout || t < 0 || t >= T || n_inf < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  StepCoef k;
  if (scheduler == 0) ddpm_coef(ac, T, n_inf, t, &k);
  else ddim_coef(ac, T, n_inf, t, eta, set_alpha_to_one, &k);
  if (k.use_noise != 0.f && !noise) return fail(CFD_E_ARG, "this step adds noise: pass the N(0,1) draw");
  hipLaunchKernelGGL(sched_step_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, model_output, noise,
                     sample_inout, numel, k, scheduler, clip, pred_original_sample);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_add_noise(cfd_handle c, const float* ac, int t, const float* original, const float* noise, float* out,
                             size_t numel, void* stream) {
  if (!c || !ac || !original || !noise || !out || t < 0) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  const float sa = sqrtf(ac[t]), sb = sqrtf(1.0f - ac[t]);
  hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, original, noise, out,
                     numel, sa, sb);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_philox_normal(cfd_handle c, float* out, int B, int per_utt, uint64_t seed, uint32_t step, uint32_t first_utt,
                                 uint32_t stream_id, void* stream) {
  if (!c || !out || B < 1 || per_utt < 4 || per_utt % 4) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  const long long n = (long long)B * per_utt / 4;
  hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, B, per_utt, seed,
                     step, first_utt, stream_id, 1.0f);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

// ---- conditioning producers ---------------------------------------------------------------------------------
extern "C" int cfd_linear_act(cfd_handle c, const float* x, long long n_rows, int K, const float* W, const float* b, int N, int act,
                              float* out, void* stream) {
  if (!c || !x || !W || !out || n_rows < 1 || K < 1 || N < 1 || act < 0 || act > 2) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  const long long gy = (n_rows + 31) / 32;
  if (gy > 65535) return fail(CFD_E_ARG, "too many rows for one launch (%lld)", n_rows);
  hipLaunchKernelGGL(linear_act_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)gy), dim3(256), 0, (hipStream_t)stream, x, n_rows, K, W, b,
                     N, act, out);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_layer_norm(cfd_handle c, const float* x, long long rows, int D, const float* gamma, const float* beta, float eps,
                              float* out, void* stream) {
  if (!c || !x || !gamma || !beta || !out || rows < 1 || D < 1 || D > 2048) return fail(CFD_E_ARG, "bad argument (D <= 2048)");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out, rows, D, eps);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_mha(cfd_handle c, const float* q, const float* k, const float* v, int Lq, int Lk, int bs, int E, int H,
                       const uint8_t* key_padding_mask, float* out, void* stream) {
  if (!c || !q || !k || !v || !out || Lq < 1 || Lk < 1 || bs < 1 || H < 1 || E % H) return fail(CFD_E_ARG, "bad argument");
  if (E / H > 64 || Lk > MHA_MAX_KEYS) return fail(CFD_E_SHAPE, "cfd_mha supports head_dim <= 64 and <= %d keys", MHA_MAX_KEYS);
  HIPCHK(hipSetDevice(c->cfg.device));
  const long long items = (long long)Lq * bs * H;
  const float scale = (float)(1.0 / std::sqrt((double)(E / H)));
  hipLaunchKernelGGL(mha_f32_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, q, k, v, key_padding_mask, out, Lq, Lk,
                     bs, E, H, scale);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_add(cfd_handle c, float* x, const float* y, size_t numel, void* stream) {
  if (!c || !x || !y || numel < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)numel);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_zero_rows(cfd_handle c, float* x, const uint8_t* keep, long long rows, int D, void* stream) {
  if (!c || !x || !keep || rows < 1 || D < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  const long long n = rows * D;
  hipLaunchKernelGGL(zero_rows_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, keep, rows, D);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

// ---- float32 pieces of the WEG gradient path (grad.hpp) and the whole evaluation (weg_eval.hpp) --------------------------
#include "weg_eval.hpp"
#include "rowtile_bwd.hpp"
#include "weg_rt.hpp"

extern "C" int cfd_gemm_f32(cfd_handle c, int M, int N, int K, int nb1, int nb2, const cfd_mat* A, const cfd_mat* B, const cfd_mat* Cm,
                            const float* bias, float alpha, int accumulate, void* stream) {
  if (!c || !A || !B || !Cm || !A->p || !B->p || !Cm->p || M < 1 || N < 1 || K < 1 || nb1 < 1 || nb2 < 1) return fail(CFD_E_ARG, "bad argument");
  if ((long long)nb1 * nb2 > 65535) return fail(CFD_E_SHAPE, "cfd_gemm_f32: at most 65535 batch entries");
  HIPCHK(hipSetDevice(c->cfg.device));
  MatView a{A->p, A->rs, A->cs, A->b1, A->b2}, b{B->p, B->rs, B->cs, B->b1, B->b2};
  launch_gemm_f32((hipStream_t)stream, a, b, const_cast<float*>(Cm->p), Cm->rs, Cm->cs, Cm->b1, Cm->b2, M, N, K, nb1, nb2, bias, alpha, accumulate);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_softmax(cfd_handle c, float* scores, long long rows, int Lk, const uint8_t* key_padding_mask, long long rows_per_batch,
                           void* stream) {
  if (!c || !scores || rows < 1 || Lk < 1 || rows_per_batch < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(softmax_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, key_padding_mask, rows, Lk,
                     rows_per_batch);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_softmax_bwd(cfd_handle c, const float* p, float* dp, const float* extra, long long rows, int Lk, void* stream) {
  if (!c || !p || !dp || rows < 1 || Lk < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(softmax_bwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, extra, rows, Lk);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_layer_norm_bwd(cfd_handle c, const float* x, const float* gamma, const float* dy, float* dx, long long rows, int D, float eps,
                                  int accumulate, void* stream) {
  if (!c || !x || !gamma || !dy || !dx || rows < 1 || D < 1 || D > 2048) return fail(CFD_E_ARG, "bad argument (D <= 2048)");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(layernorm_bwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, dx, rows, D, eps,
                     accumulate, (const float*)nullptr, (const float*)nullptr);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_ew(cfd_handle c, int op, const float* a, const float* b, float* out, size_t numel, int D, int R1, long long s0, long long s1,
                      float alpha, void* stream) {
  if (!c || !a || !out || numel < 1 || op < 0 || op >= EW_NOPS) return fail(CFD_E_ARG, "bad argument");
  if (op >= EW_SILU_BWD && !b) return fail(CFD_E_ARG, "cfd_ew: this op needs the second operand");
  if (op >= EW_ADD_BCAST && (D < 1 || R1 < 1)) return fail(CFD_E_ARG, "cfd_ew: D and R1 must be positive");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(ew_f32_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, a, b, out, (long long)numel,
                     D > 0 ? D : 1, R1 > 0 ? R1 : 1, s0, s1, alpha);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_weg_focus(cfd_handle c, const float* att, int B, int NL, int L, int S, const int32_t* tok_off, const int32_t* tok_idx, int last,
                             int nt_max, const float kernel3[3], float* workspace, float* losses, float* max_att, float* d_att, void* stream) {
  if (!c || !att || !tok_off || !tok_idx || !kernel3 || !workspace || !losses || !max_att || !d_att || B < 1 || NL < 1 || nt_max < 1)
    return fail(CFD_E_ARG, "bad argument");
  // F.pad(..., mode='reflect') with pad 1 needs at least 2 entries per axis (word_excitation_guidance.py:35)
  if (L < 2 || last - 1 < 2 || last > S) return fail(CFD_E_SHAPE, "text slice [1, %d) of %d keys / %d frames is too short for the 3x3 reflect-padded smoothing", last, S, L);
  HIPCHK(hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(weg_focus_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, att, tok_off, tok_idx, B, NL, L, S, last, nt_max,
                     kernel3[0], kernel3[1], kernel3[2], workspace, losses, max_att, d_att);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_weg_eval(cfd_handle c, const cfd_weg_args* a, float* losses, float* max_att, float* grad, float* loss_host, void* stream) {
  if (!c || !a || !a->latents || !a->tok_off || !losses || !max_att || !grad) return fail(CFD_E_ARG, "null argument");
  if (!c->finalized) return fail(CFD_E_STATE, "weights not finalized");
  const int B = a->B, L = a->L, D = c->cfg.text_encoded_dim;
  if (B < 1 || L < 2) return fail(CFD_E_ARG, "bad batch / length");
  if (L % 2) return fail(CFD_E_SHAPE, "latent length %d is odd (reference: broadcasting error at position_encoding.py:160-161)", L);
  if (L / 2 > c->qpe_rows) return fail(CFD_E_SHAPE, "L/2 = %d exceeds the query PE buffer (%d rows)", L / 2, c->qpe_rows);
  if (a->timestep < 0 || a->timestep >= c->tsin_rows) return fail(CFD_E_ARG, "timestep %d outside the timestep table (%d rows)", a->timestep, c->tsin_rows);
  if (D > 2048) return fail(CFD_E_SHAPE, "model width above 2048");
  for (int j = 0; j < CFD_NMEM; ++j) {
    if (!a->mem[j].data || a->mem[j].S < 1) return fail(CFD_E_ARG, "memory %s missing", MEM_NAMES[j]);
    if (a->mem[j].U != B || a->mem[j].row_map) return fail(CFD_E_ARG, "cfd_weg_eval takes one memory per row (U == B, no row map)");
    if (a->mem[j].S > c->mpe_rows) return fail(CFD_E_SHAPE, "memory %s has %d tokens, the memory PE buffer %d rows", MEM_NAMES[j], a->mem[j].S, c->mpe_rows);
  }
  const int St = a->mem[2].S, n_tok = a->tok_off[B];
  if (a->tok_off[0] != 0 || n_tok < 0 || (n_tok > 0 && !a->tok_idx)) return fail(CFD_E_ARG, "bad focus-token table");
  // F.pad(..., mode='reflect') with pad 1 needs at least 2 entries per axis (word_excitation_guidance.py:35)
  if (a->last - 1 < 2 || a->last > St) return fail(CFD_E_SHAPE, "text slice [1, %d) of %d keys is too short for the 3x3 reflect-padded smoothing", a->last, St);
  int nt_max = 1;
  for (int b = 0; b < B; ++b) {
    if (a->tok_off[b + 1] < a->tok_off[b]) return fail(CFD_E_ARG, "bad focus-token table");
    nt_max = std::max(nt_max, a->tok_off[b + 1] - a->tok_off[b]);
  }
  for (int t = 0; t < n_tok; ++t)
    if (a->tok_idx[t] < 1 || a->tok_idx[t] > a->last - 1) return fail(CFD_E_ARG, "focus index %d is outside the text slice [1, %d)", a->tok_idx[t], a->last);
  HIPCHK(hipSetDevice(c->cfg.device));
  c->hint_now = c->hint_same_mem = false;
  CHK(settle_deferred_census(c));
  hipStream_t caller = (hipStream_t)stream;
  // the evaluation runs on the handle's own stream (capturable, and the one the sampling graph replays on: the two
  // serialise); it starts behind whatever the caller has queued on `stream`
  hipStream_t st = c->own_stream;
  HIPCHK(hipEventRecord(c->weg_ev, caller));
  HIPCHK(hipStreamWaitEvent(st, c->weg_ev, 0));
  std::vector<int32_t> tok(a->tok_off, a->tok_off + B + 1);
  tok.insert(tok.end(), a->tok_idx, a->tok_idx + n_tok);
  if (tok != c->weg_tok_host) {                       // focus-token tables to the device (the stream may still read the old copy)
    HIPCHK(hipStreamSynchronize(st));
    CHK(c->weg_tok.ensure((size_t)(B + 1 + std::max(1, n_tok)) * 4));
    HIPCHK(hipMemcpy(c->weg_tok.p, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));
    c->weg_tok_host = tok;
    ++c->weg_tok_version;
  }
  // staging: [latents | timestep row | losses | max_att | grad]
  const size_t n_lat = (size_t)B * L * CFD_LAT, n_max = (size_t)std::max(1, n_tok);
  const size_t o_lat = 0, o_trow = o_lat + n_lat, o_loss = o_trow + (size_t)D, o_max = o_loss + (size_t)((B + 63) / 64 * 64),
               o_grad = o_max + (n_max + 63) / 64 * 64, n_io = o_grad + n_lat;
  if (n_io * 4 > c->weg_io.bytes) {
    HIPCHK(hipStreamSynchronize(st));
    CHK(c->weg_io.ensure(n_io * 4));
  }
  float* io = c->weg_io.as<float>();
  HIPCHK(hipMemcpyAsync(io + o_lat, a->latents, n_lat * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(io + o_trow, c->tsin.as<float>() + (size_t)a->timestep * D, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
  weg::Args wa{io + o_lat, io + o_trow, a->mem, c->weg_tok.as<int32_t>(), c->weg_tok.as<int32_t>() + B + 1,
               a->last, nt_max, {a->kernel3[0], a->kernel3[1], a->kernel3[2]}, io + o_loss, io + o_max, io + o_grad};
  // what the memory-side / time-only part of an evaluation depends on: with args->reuse_memory_side the caller states that the
  // memories' CONTENTS are unchanged too (a refinement loop at one timestep), and those launches are skipped
  // What the memory-side / time-only part of an evaluation depends on.  With args->reuse_memory_side the caller states that the
  // memories' CONTENTS are unchanged too, and those launches are skipped: 1 = same timestep as well (a refinement loop at one
  // timestep), 2 = the timestep may differ (the guided sampling loop: one evaluation per iteration, same conditioning).  The
  // row-tile path serves 2 from tables over ALL timesteps, built at the first such call (row t = timestep t, one launch per
  // evaluation copies the row); the float32 launch sequence treats 2 with a new timestep as 0.
  const bool use_rt = wegrt::eligible(c, a);
  std::vector<long long> sig = {B, L};
  for (int j = 0; j < CFD_NMEM; ++j) {
    sig.push_back(a->mem[j].S);
    sig.push_back((long long)(size_t)a->mem[j].data);
    sig.push_back((long long)(size_t)a->mem[j].key_padding_mask);   // (wegrt::prepare rebuilds the problem when a mask pointer changes: no reuse then)
  }
  weg::Ctx x{c, st, true, reinterpret_cast<char*>(256), 0, B, L, D, D, CFD_OK, std::string(), 0};
  // small problems (the product shape) run on the row-tile kernels, everything else on the float32 launch sequence of weg_eval.hpp
  const void* arena = nullptr;
  bool reuse = false;
  if (use_rt) {
    const bool had_full = c->wrt.T > 1;
    // tables over all timesteps stay while the caller keeps stating that the conditioning is unchanged
    const int T = a->reuse_memory_side == 2 || (a->reuse_memory_side == 1 && had_full) ? c->tsin_rows : 1;
    CHK(wegrt::prepare(c, a, T, st));
    arena = c->weg_rt_ws.p;
    sig.push_back((long long)(size_t)arena);
    sig.push_back(-(long long)T);
    if (T == 1) sig.push_back(a->timestep);
    reuse = a->reuse_memory_side != 0 && sig == c->weg_sig;
    c->wrt.T = T;
    c->weg_t_host = a->timestep;                     // in front of the launch sequence, outside any captured graph
    c->weg_dstep_host = T > 1 ? a->timestep : 0;
    if (T == 1) HIPCHK(hipMemcpyAsync(c->wk[1].trows.p, &c->weg_t_host, 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(c->wk[1].d_step.p, &c->weg_dstep_host, 4, hipMemcpyHostToDevice, st));
  } else {
    sig.push_back(a->timestep);
    weg::run(x, wa);                                 // sizing pass
    if (x.err) return fail(x.err, "missing tensor '%s' (state-dict key denoiser.%s)", x.missing.c_str(), x.missing.c_str());
    if (x.off > c->weg_ws.bytes) HIPCHK(hipStreamSynchronize(st));
    CHK(c->weg_ws.ensure(x.off));
    arena = c->weg_ws.p;
    sig.push_back((long long)(size_t)arena);
    sig.push_back((long long)x.off);
    reuse = a->reuse_memory_side != 0 && sig == c->weg_sig;
  }
  x.dry = false;
  x.base = c->weg_ws.as<char>();
  x.off = 0;
  x.launches = 0;
  x.reuse = reuse;
  c->weg_sig.clear();
  const wegrt::EvalArgs ea{io + o_lat, c->weg_tok.as<int32_t>(), c->weg_tok.as<int32_t>() + B + 1, a->last, nt_max,
                           {a->kernel3[0], a->kernel3[1], a->kernel3[2]}, io + o_loss, io + o_max, io + o_grad};
  auto run_eval = [&]() -> int {                     // the evaluation's launches (this is what a graph captures)
    if (!use_rt) { weg::run(x, wa); return CFD_OK; }
    const int r = wegrt::enqueue(c, st, !x.reuse, ea);
    x.launches = c->wrt.launches;
    return r;
  };
  // everything the launch sequence and its (by-value) kernel arguments depend on, the timestep excepted (its row is staged)
  std::vector<long long> key = {B, L, a->last, nt_max, c->weg_tok_version, (long long)(size_t)c->weg_tok.p, (long long)(size_t)io, (long long)n_io,
                                (long long)(size_t)arena, (long long)x.reuse, (long long)use_rt, (long long)(use_rt ? c->wrt.T : 0)};
  for (int j = 0; j < CFD_NMEM; ++j) {
    key.push_back(a->mem[j].S);
    key.push_back((long long)(size_t)a->mem[j].data);
    key.push_back((long long)(size_t)a->mem[j].key_padding_mask);
  }
  for (int k = 0; k < 3; ++k) { long long bits = 0; memcpy(&bits, &a->kernel3[k], 4); key.push_back(bits); }
  auto& wg = c->weg_graph[x.reuse ? 1 : 0];
  if (wg.key != key) {
    if (wg.exec) { (void)hipGraphExecDestroy(wg.exec); wg.exec = nullptr; }
    if (wg.graph) { (void)hipGraphDestroy(wg.graph); wg.graph = nullptr; }
    wg.key = key;
    wg.uses = 0;
  }
  if (c->weg_graph_on && wg.exec) {
    HIPCHK(hipGraphLaunch(wg.exec, st));
    x.launches = c->weg_launches;
  } else if (c->weg_graph_on && wg.uses >= 1) {       // second use of this key: capture, instantiate, launch
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rr = run_eval();
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rr != CFD_OK) { if (g) (void)hipGraphDestroy(g); return rr; }
    if (e != hipSuccess) return fail(CFD_E_HIP, "capturing the WEG evaluation failed: %s", hipGetErrorString(e));
    wg.graph = g;
    HIPCHK(hipGraphInstantiate(&wg.exec, wg.graph, nullptr, nullptr, 0));
    HIPCHK(hipGraphLaunch(wg.exec, st));
  } else {
    CHK(run_eval());
  }
  ++wg.uses;
  HIPCHK(hipGetLastError());
  c->weg_launches = x.launches;
  c->weg_sig = sig;
  HIPCHK(hipMemcpyAsync(losses, io + o_loss, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(max_att, io + o_max, n_max * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(grad, io + o_grad, n_lat * 4, hipMemcpyDeviceToDevice, st));
  if (loss_host) {                                   // torch.mean(losses) over the batch (word_excitation_guidance.py:80)
    std::vector<float> l(B);
    HIPCHK(hipMemcpyAsync(l.data(), io + o_loss, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    CHK(check_saturation(c, "cfd_weg_eval (latents, memories / their projections)"));   // (without loss_host: read by the next call that waits on this handle)
    float sum = 0.f;
    for (int b = 0; b < B; ++b) sum += l[b];
    *loss_host = sum / (float)B;
  } else {                                           // the caller's stream continues behind the evaluation
    c->census_pending = true;                        // (read by the handle's next entry point: settle_deferred_census)
    HIPCHK(hipEventRecord(c->weg_ev, st));
    HIPCHK(hipStreamWaitEvent(caller, c->weg_ev, 0));
  }
  return CFD_OK;
}

extern "C" int cfd_sample_inpaint(cfd_handle c) {
  if (!c) return fail(CFD_E_ARG, "null handle");
  if (!c->run_open) return fail(CFD_E_STATE, "no sampling run open");
  const cfd_sample_args& s = c->sargs;
  if (!s.preseq || s.preseq_len < 1) return CFD_OK;
  HIPCHK(hipSetDevice(c->cfg.device));
  BeginArgs ba{c->latents.as<float>(), c->w->sample_sp.as<char>(), s.B, s.L, s.G, s.preseq, c->inoise.as<float>(), s.preseq_len,
               c->coef.as<StepCoef>(), c->w->d_step.as<int>()};
  const long long n = (long long)s.B * s.preseq_len * CFD_LAT;
  hipLaunchKernelGGL(inpaint_now_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->run_stream, ba, c->w->d_step.as<int>());
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_sample_write(cfd_handle c, const float* latents) {
  if (!c || !latents) return fail(CFD_E_ARG, "null argument");
  if (!c->run_open) return fail(CFD_E_STATE, "no sampling run open");
  HIPCHK(hipSetDevice(c->cfg.device));
  const size_t lat_bytes = (size_t)c->sargs.B * c->sargs.L * CFD_LAT * 4;
  HIPCHK(hipMemcpyAsync(c->latents.p, latents, lat_bytes, hipMemcpyDeviceToDevice, c->run_stream));
  HIPCHK(hipStreamSynchronize(c->run_stream));
  return CFD_OK;
}

// ---- test hooks -----------------------------------------------------------------------------------------
extern "C" int cfd_debug_stop_stage(cfd_handle c, int stage) {
  if (!c) return fail(CFD_E_ARG, "null handle");
  c->stop_stage = stage;
  return CFD_OK;
}

extern "C" int cfd_debug_read(cfd_handle c, const char* what, float* dst_dev, size_t numel) {
  if (!c || !what || !dst_dev) return fail(CFD_E_ARG, "null argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!strcmp(what, "setup_launches")) {   // launches the last cfd_sample_begin spent on timestep-only tables (0: all served from the cache)
    const float f = (float)c->setup_launches;
    HIPCHK(hipMemcpy(dst_dev, &f, 4, hipMemcpyHostToDevice));
    return CFD_OK;
  }
  if (!strcmp(what, "sat")) {   // the saturation census as one float (not cleared)
    unsigned int n[2] = {0, 0};
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(n, c->sat.p, 8, hipMemcpyDeviceToHost));
    const float f = (float)n[0] + (float)n[1];
    HIPCHK(hipMemcpy(dst_dev, &f, 4, hipMemcpyHostToDevice));
    return CFD_OK;
  }
#if RT_STAMP
  if (!strcmp(what, "rt_ring")) {   // developer build: the launch time line (rowtile.hpp), 4 x 4096 64-bit words + the sequence counter
    HIPCHK(hipDeviceSynchronize());
    if (numel * 4 < sizeof(unsigned long long) * 4 * 4096 + 8) return fail(CFD_E_ARG, "rt_ring needs %zu bytes", sizeof(unsigned long long) * 4 * 4096 + 8);
    HIPCHK(hipMemcpyFromSymbol(dst_dev, HIP_SYMBOL(g_rt_ring), sizeof(unsigned long long) * 4 * 4096, 0, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpyFromSymbol(reinterpret_cast<char*>(dst_dev) + sizeof(unsigned long long) * 4 * 4096, HIP_SYMBOL(g_rt_seq), 4, 0, hipMemcpyDeviceToDevice));
    return CFD_OK;
  }
#endif
  const DBuf* b = nullptr;
  if (!strcmp(what, "x")) b = &c->w->x;
  else if (!strcmp(what, "temb")) b = &c->w->temb_tab;
  else if (!strcmp(what, "ss")) b = &c->w->ss_tab;
  else if (!strcmp(what, "eps")) b = &c->w->eps;
  else if (!strcmp(what, "sc")) b = &c->w->sc;
  else if (!strcmp(what, "ssc")) b = &c->w->ssc;
  else if (!strcmp(what, "xa_stamps")) b = &c->w->xa_stamps;
  else return fail(CFD_E_ARG, "unknown buffer '%s'", what);
  if (numel * 4 > b->bytes) return fail(CFD_E_ARG, "buffer '%s' holds %zu bytes, asked for %zu", what, b->bytes, numel * 4);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(dst_dev, b->p, numel * 4, hipMemcpyDeviceToDevice));
  return CFD_OK;
}

extern "C" int cfd_test_gemm(cfd_handle c, const float* X, const float* Y, float* out, int I, int J, int K, int tile_cfg,
                             void* stream) {
  if (!c || !X || !Y || !out || K % 32 || I % 4 || I < 4 || J < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  DBuf xs, ys;
  CHK(xs.ensure((size_t)I * K * 4));
  CHK(ys.ensure((size_t)J * K * 4));
  long long n = (long long)I * (K / 8);
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, xs.as<char>(), (long long)I, K, (long long)K,
                     (long long)K * 4, (unsigned int*)nullptr);
  n = (long long)J * (K / 8);
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Y, ys.as<char>(), (long long)J, K, (long long)K,
                     (long long)K * 4, (unsigned int*)nullptr);
  GemmArgs a = gemm_args();
  a.X[0] = xs.as<char>(); a.ldx[0] = (long long)K * 4; a.I[0] = I; a.Iclamp[0] = I; a.kt[0] = K / 32;
  a.Y = ys.as<char>(); a.ldy = (long long)K * 4; a.J = J; a.Jclamp = J;
  EpiF32 e;
  memset(&e, 0, sizeof(e));
  e.out = out; e.ldo = I;
  hipError_t err = launch_gemm<MODE_PLAIN, EpiF32>(a, e, 1, 1, st, tile_cfg);
  if (err != hipSuccess) return fail(CFD_E_HIP, "gemm launch failed: %s", hipGetErrorString(err));
  HIPCHK(hipStreamSynchronize(st));
  xs.release();
  ys.release();
  return CFD_OK;
}

// Micro-benchmark hook: `iters` launches of the MFMA GEMM (EpiResid epilogue: x[j][i] += D + bias) on device-resident
// SP operands filled from a float32 pattern; returns the average milliseconds per launch (HIP events).
extern "C" int cfd_bench_gemm(cfd_handle c, int I, int J, int K, int tile_cfg, int iters, float* ms_out) {
  if (!c || !ms_out || K % 32 || I % CFD_D || I < CFD_D || J < 1 || iters < 1) return fail(CFD_E_ARG, "bad argument (I must be a multiple of 512)");
  HIPCHK(hipSetDevice(c->cfg.device));
  DBuf xf, yf, xs, ys, out;
  CHK(xf.ensure((size_t)I * K * 4));
  CHK(yf.ensure((size_t)J * K * 4));
  CHK(xs.ensure((size_t)I * K * 4));
  CHK(ys.ensure((size_t)J * K * 4));
  CHK(out.ensure((size_t)J * I * 4));
  HIPCHK(hipMemset(out.p, 0, (size_t)J * I * 4));
  long long n = (long long)I * K / 4;
  hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, xf.as<float>(), 1, I * K, 1ull, 0u, 0u, 3u, 0.05f);
  n = (long long)J * K / 4;
  hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, yf.as<float>(), 1, (int)((long long)J * K), 2ull, 0u, 0u, 3u, 1.0f);
  if (getenv("CFD_BENCH_ZERO")) {   // power/clock probe: all-zero operands
    HIPCHK(hipMemset(xf.p, 0, (size_t)I * K * 4));
    HIPCHK(hipMemset(yf.p, 0, (size_t)J * K * 4));
  }
  n = (long long)I * (K / 8);
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, xf.as<float>(), xs.as<char>(), (long long)I, K, (long long)K, (long long)K * 4, (unsigned int*)nullptr);
  n = (long long)J * (K / 8);
  hipLaunchKernelGGL(to_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, yf.as<float>(), ys.as<char>(), (long long)J, K, (long long)K, (long long)K * 4, (unsigned int*)nullptr);
  GemmArgs a = gemm_args();
  a.X[0] = xs.as<char>(); a.ldx[0] = (long long)K * 4; a.I[0] = I; a.Iclamp[0] = I; a.kt[0] = K / 32;
  a.Y = ys.as<char>(); a.ldy = (long long)K * 4; a.J = J; a.Jclamp = J;
  EpiResid e{out.as<float>(), 0, nullptr};
  EpiNull en{out.as<float>()};
  EpiF32 ef;
  memset(&ef, 0, sizeof(ef));
  ef.out = out.as<float>(); ef.ldo = I;
  const char* ev = getenv("CFD_BENCH_EPI");
  const int epi_kind = ev ? atoi(ev) : 0;   // 0 residual RMW, 1 no stores, 2 plain fp32 store, 3 split-pair store
  EpiSplit es;
  memset(&es, 0, sizeof(es));
  es.out = out.as<char>(); es.ldo = (long long)I * 4;
  if (epi_kind == 0 && I != CFD_D) return fail(CFD_E_ARG, "the residual epilogue has rows of 512");
  auto go = [&]() -> hipError_t {
    if (epi_kind == 1) return launch_gemm<MODE_PLAIN, EpiNull>(a, en, 1, 1, nullptr, tile_cfg);
    if (epi_kind == 2) return launch_gemm<MODE_PLAIN, EpiF32>(a, ef, 1, 1, nullptr, tile_cfg);
    if (epi_kind == 3) return launch_gemm<MODE_PLAIN, EpiSplit>(a, es, 1, 1, nullptr, tile_cfg);
    return launch_gemm<MODE_PLAIN, EpiResid>(a, e, 1, 1, nullptr, tile_cfg);
  };
  hipError_t err = go();   // warm-up
  if (err != hipSuccess) return fail(CFD_E_HIP, "gemm launch failed: %s", hipGetErrorString(err));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipEventRecord(c->pev[0], nullptr));
  for (int i = 0; i < iters; ++i) (void)go();
  HIPCHK(hipEventRecord(c->pev[1], nullptr));
  HIPCHK(hipEventSynchronize(c->pev[1]));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, c->pev[0], c->pev[1]));
  *ms_out = ms / iters;
  xf.release(); yf.release(); xs.release(); ys.release(); out.release();
  return CFD_OK;
}
