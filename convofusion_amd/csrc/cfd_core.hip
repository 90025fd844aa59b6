// libcfdenoise: the handle (create / destroy, the developer knobs read once per handle), the checkpoint's tensors, and the load-time weight
// folding and re-layout of cfd_finalize_weights.
#include "cfd_internal.hpp"

int g_cfd_naive_gemm = 0;

static thread_local char g_err[1024] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

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
  if (env) c->xa_operands = atoi(env) & 15;
  env = getenv("CFD_LN_FOLD");
  if (env) c->ln_fold = atoi(env);
  env = getenv("CFD_XA_DB");
  if (env) c->xa_db = atoi(env) != 0;
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
  DBuf* all[] = {&c->we_sp, &c->wp_sp, &c->wp_f, &c->ln_cd_p, &c->we_all, &c->be_all, &c->tsin, &c->weg_ws, &c->weg_tok, &c->latents, &c->coef, &c->inoise};
  for (DBuf* b : all) b->release();
  c->wk[0].release();
  c->wk[1].release();
  for (int j = 0; j < CFD_NMEM; ++j) {
    c->wk_all_sp[j].release(); c->wv_all_sp[j].release(); c->mem_own[j].release(); c->perm_map[j].release();
  }
  for (auto& l : c->lw) {
    DBuf* lb[] = {&l.wqk_f, &l.wv_f, &l.w1_f, &l.ln_cd, &l.wqk_sp, &l.bqk, &l.wv_sp, &l.wo_sp, &l.bo2, &l.wtb1_sp, &l.wtb2_sp, &l.w1_sp, &l.w2_sp, &l.cross_bias};
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

int to_sp(Ctx* c, const float* src, long long R, int K, DBuf& dst, long long dst_rows) {
  if (dst_rows < 0) dst_rows = R;
  CHK(dst.ensure((size_t)dst_rows * K * 4));
  if (dst_rows > R) HIPCHK(hipMemset(dst.p, 0, (size_t)dst_rows * K * 4));
  return enqueue_to_split(c, CFD_PROF_OTHER, 0, src, dst.as<char>(), R, K, (long long)K, (long long)K * 4, c->sat_mem());
}

int enqueue_to_split(Ctx* c, int cls, hipStream_t st, const float* src, char* dst, long long R, int K, long long ld_src, long long ld_dst, unsigned int* sat) {
  const long long n = R * (K / 8);
  LAUNCH(cls, to_split_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), st, src, dst, R, K, ld_src, ld_dst, sat);
  return CFD_OK;
}

// The LayerNorm fold's weight side (gemm_sp.hpp EpiLn), one workgroup per output feature r:
//   Wf[r][k] = W[r][k] gamma[k]        c[r] = sum_k (hi + lo)(Wf[r][k])  -- the value the split-pair product sees --        d[r] = sum_k W[r][k] beta[k]
template <int CFD_KI = 0>
__global__ void __launch_bounds__(256) ln_fold_rows_kernel(const float* W, const float* gamma, const float* beta, float* Wf, float* cvec, float* dvec, int K) {
  const int r = blockIdx.x;
  double sc = 0.0, sd = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float w = W[(long long)r * K + k];
    const float wf = w * gamma[k];
    Wf[(long long)r * K + k] = wf;
    sp_t hi, lo;
    split_f32(wf, hi, lo);
    sc += (double)((float)hi + (float)lo);
    sd += (double)w * (double)beta[k];
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = sc; red[1][threadIdx.x] = sd;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { cvec[r] = (float)red[0][0]; dvec[r] = (float)red[1][0]; }
}
static int ln_fold_weight(Ctx* c, const float* W, int R, int K, const float* gamma, const float* beta, DBuf& tmp, DBuf& dst_sp, float* cvec, float* dvec) {
  CHK(tmp.ensure((size_t)R * K * 4));
  hipLaunchKernelGGL(ln_fold_rows_kernel<>, dim3(R), dim3(256), 0, 0, W, gamma, beta, tmp.as<float>(), cvec, dvec, K);
  HIPCHK(hipGetLastError());
  return to_sp(c, tmp.as<float>(), R, K, dst_sp);
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
  DBuf tmpf, tmpd1, tmpd2, vd1, vd2, accd, tmpfold;
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
    const float* ipw_self = nullptr;
    // -- self attention: q rows scaled by sqrt(1/head_dim) (F.multi_head_attention_forward "q_scaled")
    CHK(need(c, p + "self_attn.in_proj_weight", (size_t)3 * D * D, &ipw));
    ipw_self = ipw;
    CHK(need(c, p + "self_attn.in_proj_bias", 3 * D, &ipb));
    CHK(need(c, p + "self_attn.out_proj.weight", (size_t)D * D, &ow));
    CHK(need(c, p + "self_attn.out_proj.bias", D, &ob));
    const float qs = (float)std::sqrt(1.0 / (double)CFD_HD);
    float* tf = tmpf.as<float>();
    hipLaunchKernelGGL(scale_copy_kernel<>, grid1((long long)D * D), blk, 0, 0, ipw, tf, (long long)D * D, qs);
    hipLaunchKernelGGL(scale_copy_kernel<>, grid1((long long)D * D), blk, 0, 0, ipw + (size_t)D * D, tf + (size_t)D * D,
                       (long long)D * D, 1.0f);
    CHK(to_sp(c, tf, 2 * D, D, w.wqk_sp));
    CHK(w.bqk.ensure(2 * D * 4));
    hipLaunchKernelGGL(scale_copy_kernel<>, grid1(D), blk, 0, 0, ipb, w.bqk.as<float>(), (long long)D, qs);
    hipLaunchKernelGGL(scale_copy_kernel<>, grid1(D), blk, 0, 0, ipb + D, w.bqk.as<float>() + D, (long long)D, 1.0f);
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
    {   // the LayerNorm fold's operands: norm1 -> q | k (the scaled copy in tmpf) and v, norm3 -> FFN1
      CHK(w.ln_cd.ensure(6144 * 4));
      float* cd = w.ln_cd.as<float>();
      CHK(ln_fold_weight(c, tmpf.as<float>(), 2 * D, D, w.ln1g, w.ln1b, tmpfold, w.wqk_f, cd, cd + 1024));
      CHK(ln_fold_weight(c, ipw_self + (size_t)2 * D * D, D, D, w.ln1g, w.ln1b, tmpfold, w.wv_f, cd + 2048, cd + 2560));
      CHK(ln_fold_weight(c, t0, CFD_FF, D, w.ln3g, w.ln3b, tmpfold, w.w1_f, cd + 3072, cd + 4096));
    }
    CHK(need(c, p + "linear2.weight", (size_t)D * CFD_FF, &t0)); CHK(to_sp(c, t0, D, CFD_FF, w.w2_sp));
    CHK(need(c, p + "linear2.bias", D, &w.b2));
    // -- five single-head cross attentions + att_fuser, folded onto the memory side (DESIGN.md "Folding")
    const float *fw, *fb;
    CHK(need(c, p + "att_fuser.weight", (size_t)D * 5 * D, &fw));
    CHK(need(c, p + "att_fuser.bias", D, &fb));
    hipLaunchKernelGGL(f2d_kernel<>, grid1(D), blk, 0, 0, fb, accd.as<double>(), D);
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
    hipLaunchKernelGGL(d2f_kernel<>, grid1(D), blk, 0, 0, accd.as<double>(), w.cross_bias.as<float>(), D);
    HIPCHK(hipGetLastError());
  }
  for (int j = 0; j < CFD_NMEM; ++j) {
    CHK(to_sp(c, wk_f[j].as<float>(), kfeat + 32, D, c->wk_all_sp[j]));
    CHK(to_sp(c, wv_f[j].as<float>(), kfeat, D, c->wv_all_sp[j]));
  }
  {   // the decoder's final norm folded into latent_proj
    const float *gF, *bF, *wpw;
    CHK(need(c, "decoder.norm.weight", D, &gF));
    CHK(need(c, "decoder.norm.bias", D, &bF));
    CHK(need(c, "latent_proj.weight", (size_t)CFD_LAT * D, &wpw));
    CHK(c->ln_cd_p.ensure(2 * CFD_LAT * 4));
    CHK(ln_fold_weight(c, wpw, CFD_LAT, D, gF, bF, tmpfold, c->wp_f, c->ln_cd_p.as<float>(), c->ln_cd_p.as<float>() + CFD_LAT));
  }
  HIPCHK(hipDeviceSynchronize());
  CHK(check_saturation(c, "cfd_finalize_weights (a weight or a folded weight product)"));
  for (int j = 0; j < CFD_NMEM; ++j) { wk_f[j].release(); wv_f[j].release(); }
  tmpf.release(); tmpd1.release(); tmpd2.release(); vd1.release(); vd2.release(); accd.release(); tmpfold.release();
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

