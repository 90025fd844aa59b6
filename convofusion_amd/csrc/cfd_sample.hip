// libcfdenoise: the sampling loop -- scheduler coefficients, the captured iteration, cfd_sample_* / cfd_dyadic_steps -- and the stand-alone
// scheduler / RNG entry points.
#include "cfd_internal.hpp"

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
  LAUNCH(CFD_PROF_OTHER, begin_step_kernel<>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), st, ba);
  CHK(enqueue_denoise(c, st));
  CfgStepArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.eps = c->w->eps.as<float>(); ca.latents = c->latents.as<float>(); ca.B = s.B; ca.L = s.L; ca.G = s.G;
  for (int k = 0; k < 8; ++k) { ca.w[k] = s.guidance_weight[k]; ca.pos[k] = c->chunk_pos[k]; }
  ca.kind = s.scheduler; ca.clip = s.clip_sample; ca.coef = c->coef.as<StepCoef>(); ca.d_step = c->w->d_step.as<int>();
  ca.noise = s.step_noise; ca.seed = s.seed; ca.utt0 = s.first_utterance;
  const long long n4 = (long long)s.B * s.L * CFD_LAT / 4;
  ca.advance = c->w->d_step.as<int>();   // the last workgroup of cfg_step_kernel advances the loop index
  LAUNCH(CFD_PROF_OTHER, cfg_step_kernel<>, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 256)), dim3(256), st, ca);
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
  c->want_opf = (n_ring || s.dynamic_memory_mask) ? 0 : (c->xa_operands >= 0 ? c->xa_operands : (s.operand_policy & 15));
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
    CHK(enqueue_philox_fill(c->latents.as<float>(), s.B, s.L * CFD_LAT, (uint64_t)s.seed, 0u, s.first_utterance, 1u, 1.0f, st));
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
    (void)enqueue_linear_act(lat, rows, CFD_LAT, pr->w1, pr->b1, pr->hidden, 1, pr->tmp, st);
    (void)enqueue_linear_act((const float*)pr->tmp, rows, pr->hidden, pr->w2, pr->b2, pr->out_dim, 1, spk, st);
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
extern "C" int cfd_scheduler_step(cfd_handle c, int scheduler, const float* ac, int T, int n_inf, int t, int clip, float eta,
                                  int set_alpha_to_one, const float* model_output, const float* noise, float* sample_inout,
                                  size_t numel, float* pred_original_sample, void* stream) {
  if (!c || !ac || !model_output || !sample_inout || t < 0 || t >= T || n_inf < 1) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  StepCoef k;
  if (scheduler == 0) ddpm_coef(ac, T, n_inf, t, &k);
  else ddim_coef(ac, T, n_inf, t, eta, set_alpha_to_one, &k);
  if (k.use_noise != 0.f && !noise) return fail(CFD_E_ARG, "this step adds noise: pass the N(0,1) draw");
  hipLaunchKernelGGL(sched_step_kernel<>, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, model_output, noise,
                     sample_inout, numel, k, scheduler, clip, pred_original_sample);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

extern "C" int cfd_add_noise(cfd_handle c, const float* ac, int t, const float* original, const float* noise, float* out,
                             size_t numel, void* stream) {
  if (!c || !ac || !original || !noise || !out || t < 0) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  const float sa = sqrtf(ac[t]), sb = sqrtf(1.0f - ac[t]);
  hipLaunchKernelGGL(add_noise_kernel<>, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, original, noise, out,
                     numel, sa, sb);
  HIPCHK(hipGetLastError());
  return CFD_OK;
}

int enqueue_philox_fill(float* out, int B, int per_utt, uint64_t seed, uint32_t step, uint32_t utt0, uint32_t stream_id, float scale, hipStream_t st) {
  const long long n = (long long)B * per_utt / 4;
  hipLaunchKernelGGL(philox_fill_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, B, per_utt, seed, step, utt0, stream_id, scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CFD_OK : fail(CFD_E_HIP, "philox_fill_kernel launch failed: %s", hipGetErrorString(e));
}

extern "C" int cfd_philox_normal(cfd_handle c, float* out, int B, int per_utt, uint64_t seed, uint32_t step, uint32_t first_utt,
                                 uint32_t stream_id, void* stream) {
  if (!c || !out || B < 1 || per_utt < 4 || per_utt % 4) return fail(CFD_E_ARG, "bad argument");
  HIPCHK(hipSetDevice(c->cfg.device));
  return enqueue_philox_fill(out, B, per_utt, seed, step, first_utt, stream_id, 1.0f, (hipStream_t)stream);
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
  hipLaunchKernelGGL(inpaint_now_kernel<>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->run_stream, ba, c->w->d_step.as<int>());
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

