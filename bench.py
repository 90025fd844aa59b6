#!/usr/bin/env python
"""Benchmark of the denoising loop (BASELINE.json metric: denoise-steps/sec, B=32 utterances per GPU,
196-token latents, 1500 audio tokens, 1000-step DDPM schedule, 7-way guidance => denoiser batch 224).

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a launcher environment: bench.py starts N worker processes itself (torch.distributed.run on
      127.0.0.1, one rank per GPU, RCCL); under the driver's own torch.distributed.run launch it is a rank.

One "step" = one iteration of the captured loop for the rank's 32 utterances: replicate latents x7,
denoiser forward, guidance combine, scheduler step.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
B_PER_GPU = 32
L = 196
S = (32, 1500, 32, 8, 1)
G = 7
NL = 9


def canonical_flops_per_step(Be, Lq, Ss, nl=NL):
    """reference formulation, GEMM + attention only, 2 FLOP/MAC, no de-duplication."""
    sS = sum(Ss)
    macs_row = 65536 * Lq * 2 + 524288 + nl * (6553600 * Lq + 1024 * Lq * Lq + 524288 * sS + 1024 * Lq * sS + 1048576)
    return 2.0 * macs_row * Be


def executed_gemm_flops(Be, Lq, Ss, U, nl=NL, shared_rows=None, l0_pairs=None, one_key=True):
    """Algorithmic (single-product) FLOPs of the GEMMs the HIP pipeline actually launches, per class.
    Every product is issued as 3 f16 MFMAs, so MFMA-issued FLOPs are 3x these.  ``shared_rows``: rows that run the
    replica-independent head of the network (embedding, layer 0's self-attention and first time block) when the
    batch is G replicas of them (csrc/cfd_internal.hpp Problem::share_B); the other rows do not launch those products.
    ``l0_pairs``: layer 0's attention against the audio memory is evaluated once per distinct (utterance, instance) pair
    (csrc/cfd_problem.hip build_xattn_layer0_lists; None: for every row).  ``one_key``: the one-key memory (lsnemb) has no tile
    step in the fused kernel (XAttnArgs::one_j): its 32 padded keys are not multiplied at all."""
    M = Be * Lq
    M0 = (shared_rows if shared_rows else Be) * Lq
    B0 = shared_rows if shared_rows else Be
    Lp = (Lq + 31) // 32 * 32
    Sp = [(s + 31) // 32 * 32 for s in Ss]
    tok = 2.0 * M0 * 512 * 128 + 2.0 * M * 128 * 512                      # embed (shared rows) + latent proj
    tok += nl * 2.0 * M * 512 * (1024 + 512 + 512 + 512 + 512 + 1024 + 1024)  # qk, v^T, Wo, TB1, TB2, FFN1, FFN2
    tok -= 2.0 * (M - M0) * 512 * (1024 + 512 + 512 + 512)                # layer 0: qk, v^T, Wo, TB1 on the shared rows only
    mem = sum(2.0 * u * sp * 512 * (2 * nl * 512) for u, sp in zip(U, Sp))
    att = nl * 2.0 * Be * 4 * Lq * Lp * 128 * 2                           # self-attention (fused kernel), padded key axis
    att -= 2.0 * (Be - B0) * 4 * Lq * Lp * 128 * 2                        # layer 0's self-attention on the shared rows only
    xat = nl * sum(2.0 * Be * Lq * sp * 512 * 2 for sp in Sp)             # cross-attention: scores + P.V over the padded keys
    if l0_pairs is not None:
        xat -= 2.0 * (Be - l0_pairs) * Lq * Sp[1] * 512 * 2               # layer 0, audio memory: distinct pairs only
    if one_key:
        xat -= nl * sum(2.0 * Be * Lq * sp * 512 * 2 for s_, sp in zip(Ss, Sp) if s_ == 1)
    return {"gemm_token": tok, "gemm_mem": mem, "gemm_attn": att, "xattn": xat}


def make_model(device, seed=1234):
    from convofusion_amd.denoiser import Denoiser
    abl = SimpleNamespace(SKIP_CONNECT=True, VAE_TYPE="convofusion", DIFF_PE_TYPE="convofusion", CAUSAL_ATTN=False)
    torch.manual_seed(seed)
    m = Denoiser(ablation=abl, nfeats=189, condition="text+audio", latent_dim=[1, 128], ff_size=1024, num_layers=NL,
                 num_heads=4, dropout=0.1, normalize_before=True, activation="gelu", flip_sin_to_cos=True,
                 position_embedding="sine", arch="trans_dec", freq_shift=0, text_encoded_dim=512, audio_encoded_dim=512)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():   # random-init weights of the reference architecture; layers perturbed independently
        for name, p in m.named_parameters():
            if p.dim() >= 2:
                p.add_(0.02 * torch.randn(p.shape, generator=g))
    return m.to(device).eval()


def make_inputs(B, device, seed):
    """Synthetic conditioning in the 7-chunk guidance pattern:
    per memory B conditional tensors + one shared unconditional tensor; text memories mask their last 8 keys."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    cond_chunks = {0: (3, 6), 1: (2, 6), 2: (1, 6), 3: (4, 6), 4: (5, 6)}
    mems, masks = [], {}
    names = ("spkemb", "alsn", "tlsn", "apb", "lsnemb")
    for j, s in enumerate(S):
        cond = torch.randn(B, s, 512, generator=g)
        unc = torch.randn(1, s, 512, generator=g)
        uq = torch.cat([unc, cond], 0)
        rm = torch.zeros(G, B, dtype=torch.long)
        for c in cond_chunks[j]:
            rm[c] = 1 + torch.arange(B)
        mems.append(uq[rm.reshape(-1)].to(device))
        if names[j] in ("spkemb", "tlsn"):
            mk = torch.zeros(G * B, s, dtype=torch.bool)
            mk[:, s - 8:] = True
            masks[names[j]] = mk.to(device)
        else:
            masks[names[j]] = None
    return mems, masks


def _cpu_info():
    """(model string, physical cores) from /proc/cpuinfo; falls back to os.cpu_count()."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or os.cpu_count() or 1)


def cpu_baseline(model, n_warm=3, n_timed=10, budget_s=30.0):
    """The reference's own CPU path -- PyTorch eager float32 (oracle/denoiser_torch.py: the same torch op sequence
    as ``Denoiser.forward``, pinned against the imported reference class by tests/test_oracle_denoiser.py) + the 7-way
    guidance combine + the DDPM step -- timed on this box's host cores (kind = "port").  Bounded sample:
    ``n_warm`` + ``n_timed`` single-utterance steps (Be=7) at timesteps spread over the schedule, one B=4 step (Be=28)
    and, when the single-utterance time says it fits ``budget_s``, one full B=32 step (Be=224) -- the headline
    workload itself; ``value`` is taken from the largest batch that was timed."""
    from oracle import denoiser_torch, sampler_ref, scheduler_ref, weights
    cpu_model, cores = _cpu_info()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    tsd = denoiser_torch.to_torch(weights.extend_pe(sd, 1536))
    sch = scheduler_ref.DDPMSchedulerRef()
    sch.set_timesteps(1000)
    g = torch.Generator().manual_seed(0)

    def step(B, t):
        x = torch.randn(B, L, 128, generator=g)
        mems = [torch.randn(7 * B, s, 512, generator=g) for s in S]
        t0 = time.perf_counter()
        eps, _ = denoiser_torch.denoiser_forward(tsd, torch.cat([x] * 7), int(t), mems, {})
        e = sampler_ref.cfg_combine(eps.numpy(), 7.5)
        sch.step(e, int(t), x.numpy(), noise=np.zeros_like(e))
        return time.perf_counter() - t0

    # thread count: every physical core is not the fastest choice for eager torch on this problem (128 threads on a 2 x 64-core
    # host ran a step in 3.0 s, 8 threads on a small VM in 1.8 s), and a baseline that is slower than it need be flatters the
    # GPU: one step per candidate, keep the fastest
    step(1, 999)
    tried = {}
    for n in sorted({cores, max(1, cores // 2), max(1, cores // 4), max(1, cores // 8), 16, 8} & set(range(1, cores + 1))):
        torch.set_num_threads(n)
        step(1, 998)
        tried[n] = step(1, 997)
    threads = min(tried, key=tried.get)
    torch.set_num_threads(threads)
    for i in range(n_warm):
        step(1, 999 - i)
    ts = [int(t) for t in np.linspace(999, 0, n_timed)]
    t1 = float(np.mean([step(1, t) for t in ts]))
    t4 = step(4, 500)
    sample = (f"torch {torch.__version__} CPU eager fp32 on '{cpu_model}' ({cores} physical cores; threads tried -> s/step: "
              f"{ {k: round(v, 2) for k, v in tried.items()} }, using {threads}): {n_warm} warm-up + {n_timed} timed "
              f"single-utterance steps (Be=7, L={L}, S={S}) at {t1:.3f} s each; one B=4 step (Be=28) {t4:.2f} s")
    per_b32 = t4 * (B_PER_GPU / 4)
    if min(t1 * B_PER_GPU, per_b32) <= budget_s:
        t32 = step(B_PER_GPU, 500)
        sample += f"; one full B=32 step (Be=224) {t32:.2f} s (value = 1 / this)"
        per_b32 = t32
    else:
        sample += f"; a B=32 step counted as 8 B=4 steps (a timed Be=224 step would exceed the {budget_s:.0f} s sample budget)"
    return {"value": 1.0 / per_b32, "unit": "denoise-steps/s (B=32)", "cores": threads, "physical_cores": cores, "kind": "port", "cpu": cpu_model,
            "single_utterance_s": t1, "b4_step_s": t4, "sample": sample}


def secondary_configs(model, device):
    """The BASELINE.json configurations that are not the headline line, on the same box and process (rank 0, N = 1): a few seconds each.
      R   product shape (L=16, 161 audio tokens), B=32: the captured loop's steps/s
      C1  ONE utterance at the product shape, the whole 1000-step DDPM run end to end (row-tile path), with the HBM roofline of
          SURVEY.md section 8d's algorithmic bytes (float32 weights once per step + memories read by every layer + latents)
      WEG one objective + gradient evaluation at that shape (cfd_weg_eval) and a guided single-utterance run
      C4  50-step DDIM, B=32, headline shape: end-to-end latency through sample()
      C5  dyadic reactive path, B=16 per side, headline shape: one lock-step iteration"""
    global L, S
    from convofusion_amd import scheduler, weg
    from convofusion_amd.conditioning import default_fuser
    from convofusion_amd.dyadic import DyadicRun
    from convofusion_amd.sampler import SamplingRun, sample, sample_with_weg
    sk = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=True)
    ddpm = scheduler.DDPMScheduler(variance_type="fixed_small", **sk)
    out = {}
    L0, S0 = L, S

    def sync():
        torch.cuda.synchronize(device)
    try:
        # ---- C4 / C5 at the headline shape
        mems, masks = make_inputs(B_PER_GPU, device, seed=1234)
        ddim = scheduler.DDIMScheduler(set_alpha_to_one=True, steps_offset=0, **sk)
        sample(model, ddim, mems, masks, B=B_PER_GPU, L=L, num_inference_steps=2, seed=0)
        sync()
        t0 = time.perf_counter()
        sample(model, ddim, mems, masks, B=B_PER_GPU, L=L, num_inference_steps=50, eta=0.0, seed=1)
        sync()
        dt = time.perf_counter() - t0
        out["c4_ddim50_b32"] = {"latency_s_end_to_end": dt, "steps_per_s": 50 / dt,
                                "note": "sample(): de-duplication, tables, warm-up iteration, capture, 50 replays, read"}
        del mems, masks
        Bd = 16
        g = torch.Generator().manual_seed(5)
        cond = lambda: [torch.randn(Bd, L if j == 0 else S[j], 512, generator=g).to(device) for j in range(5)]   # noqa: E731
        uncond = [torch.randn(1, L if j == 0 else S[j], 512, generator=g).to(device) for j in range(5)]
        model_b = make_model(device)
        ca, cb_, fus, c5 = cond(), cond(), default_fuser().to(device).eval(), {}
        for key, mb, shared in (("two_handles", model_b, False), ("shared_weights_one_run", None, True)):
            run = DyadicRun(model, mb, ddpm, fus, ca, cb_, uncond, Bd, L, 1000, seed=3, shared_weights=shared)
            run.steps(3)
            run.read()
            t0 = time.perf_counter()
            run.steps(10)
            run.read()
            sync()
            c5[key] = (time.perf_counter() - t0) / 10
            run.read(close=True)
            del run
        del model_b
        out["c5_dyadic_b16x2"] = {"ms_per_lockstep_iteration": c5["shared_weights_one_run"] * 1e3, "iterations_per_s": 1 / c5["shared_weights_one_run"],
                                  "ms_per_lockstep_iteration_two_handles": c5["two_handles"] * 1e3,
                                  "note": "both sides share the denoiser's weights here: one captured iteration of the 32-utterance double batch + the "
                                          "two partner projections per lock-step iteration (DyadicRun(shared_weights=True)); two_handles = two "
                                          "denoisers, two 16-utterance graphs replayed one after the other on one stream"}
        # ---- product shape
        L, S = 16, (24, 161, 24, 8, 1)
        mems, masks = make_inputs(B_PER_GPU, device, seed=1234)
        run = SamplingRun(model, ddpm, mems, masks, B_PER_GPU, L, 1000, guidance_scale=7.5, seed=0)
        run.steps(3)
        run.read()
        t0 = time.perf_counter()
        run.steps(50)
        run.read()
        dt = time.perf_counter() - t0
        run.close()
        # ... and with every iteration's attention maps kept (the reference's default dict; here by the fused cross-attention kernel itself)
        run = SamplingRun(model, ddpm, mems, masks, B_PER_GPU, L, 100, guidance_scale=7.5, seed=0, attention_ring=True)
        run.steps(3)
        run.read()
        t0 = time.perf_counter()
        run.steps(50)
        run.read()
        dt_ring = time.perf_counter() - t0
        run.close()
        del run
        # ... and what a caller gets WITHOUT convofusion_amd.install() (only the yaml edits): the reference's own Python loop (convofusion.py:499-544:
        # replicate x 7, denoiser, guidance combine, scheduler.step) on the HIP Denoiser.forward -- one cfd_forward with att_mats per iteration
        ddpm.set_timesteps(1000)
        lat_py = torch.randn((B_PER_GPU, L, 128), device=device)
        t_py = 0.0
        for it, t in enumerate(ddpm.timesteps[:45]):
            if it == 5:
                sync()
                t0 = time.perf_counter()
            with torch.no_grad():
                npred, _ = model(sample=torch.cat([lat_py] * G), timestep=t, encoder_hidden_states=mems, mem_mask_dict=masks)
            u_, tx_, a_, s_, p_, i_, f_ = npred.chunk(G)
            npred = u_ + 7.5 * (tx_ - u_) + 7.5 * (a_ - u_) + 7.5 * (s_ - u_) + 7.5 * (p_ - u_) + 7.5 * (i_ - u_) + 7.5 * 0 * (f_ - u_)
            lat_py = ddpm.step(npred, t, lat_py).prev_sample
        sync()
        t_py = (time.perf_counter() - t0) / 40
        out["r_product_shape_b32"] = {"steps_per_s": 50 / dt, "ms_per_step": 1000 * dt / 50, "workload": f"B={B_PER_GPU}, L={L}, S={S}",
                                      "all_attention_maps_ms_per_step": 1000 * dt_ring / 50, "all_attention_maps_over_none": dt_ring / dt,
                                      "reference_python_loop_on_hip_denoiser_ms_per_step": 1000 * t_py}
        # ... and the smaller batches of the same shape (8 / 16 utterances: 896 / 1 792 token rows), 200 replays each
        for b_small in (8, 16):
            mems, masks = make_inputs(b_small, device, seed=1234)
            run = SamplingRun(model, ddpm, mems, masks, b_small, L, 1000, guidance_scale=7.5, seed=0)
            run.steps(5)
            run.read()
            t0 = time.perf_counter()
            run.steps(200)
            run.read()
            dt_small = time.perf_counter() - t0
            run.close()
            del run
            out["r_product_shape_b32"][f"b{b_small}_ms_per_step"] = 1000 * dt_small / 200
        mems, masks = make_inputs(1, device, seed=1234)
        run = SamplingRun(model, ddpm, mems, masks, 1, L, 1000, guidance_scale=7.5, seed=0)
        run.steps(2)
        c1_launches = sum(v[1] for v in run.profile().values()) + 2     # + begin_step / cfg_step around the forward
        # host against GPU: how long the replays of 900 iterations take to ENQUEUE (hipGraphLaunch per iteration) and to finish
        run.steps(8)
        run.read()
        sync()
        t0 = time.perf_counter()
        run.steps(900)
        t_enq = time.perf_counter() - t0
        run.read()
        sync()
        t_gpu = time.perf_counter() - t0
        run.close()
        sample(model, ddpm, mems, masks, B=1, L=L, num_inference_steps=4, seed=0)
        sync()
        t0 = time.perf_counter()
        sample(model, ddpm, mems, masks, B=1, L=L, num_inference_steps=1000, seed=0)
        sync()
        dt = time.perf_counter() - t0
        # every iteration's attention maps kept (the reference's default dict, convofusion.py:517-523): the captured iteration stores them
        sample(model, ddpm, mems, masks, B=1, L=L, num_inference_steps=4, seed=0, return_attention="all")
        sync()
        t0 = time.perf_counter()
        _, atts_all = sample(model, ddpm, mems, masks, B=1, L=L, num_inference_steps=1000, seed=0, return_attention="all")
        sync()
        dt_all = time.perf_counter() - t0
        assert len(atts_all) == 1000
        del atts_all
        # set-up of a run (cfd_sample_begin): with the timestep-only tables built for another timestep list / served from the handle's cache
        SamplingRun(model, ddpm, mems, masks, 1, L, 500, guidance_scale=7.5, seed=0).close()
        sync()
        t0 = time.perf_counter()
        r_ = SamplingRun(model, ddpm, mems, masks, 1, L, 1000, guidance_scale=7.5, seed=0)
        sync()
        setup_first = time.perf_counter() - t0
        r_.close()
        t0 = time.perf_counter()
        r_ = SamplingRun(model, ddpm, mems, masks, 1, L, 1000, guidance_scale=7.5, seed=0)
        sync()
        setup_repeat = time.perf_counter() - t0
        r_.close()
        # SURVEY.md section 8d, B = 1 at the product shape: float32 weights once per step + the memories read by each of the 9
        # layers + the latents (in, 7 replicas of eps out)
        n_w = sum(p.numel() for p in model.parameters()) * 4
        alg_bytes = n_w + G * sum(S) * 512 * 4 * NL + (1 + G) * L * 128 * 4
        ach = alg_bytes / (dt / 1000) / 1e9
        out["c1_single_utterance"] = {"s_per_1000": dt, "steps_per_s": 1000 / dt, "launches_per_step": c1_launches,
                                      "gpu_us_per_step": t_gpu / 900 * 1e6, "host_enqueue_us_per_step": t_enq / 900 * 1e6,
                                      "all_attention_maps_s_per_1000": dt_all, "all_attention_maps_over_last": dt_all / dt,
                                      "setup_ms_first": setup_first * 1e3, "setup_ms_repeat": setup_repeat * 1e3,
                                      "workload": f"B=1 (denoiser batch {G}), L={L}, S={S}, 1000-step DDPM end to end through sample()",
                                      "roofline": {"bound": "latency", "launches_per_step": c1_launches, "us_per_launch": t_gpu / 900 * 1e6 / max(c1_launches, 1),
                                                   "algorithmic_bytes_per_step": alg_bytes, "algorithmic_bytes_rate_GBps": ach,
                                                   "algorithmic_bytes_rate_over_hbm_peak": ach / 8000.0, "traffic": None,
                                                   "note": "latency-bound: ~86 dependent launches per step (rowtile.hpp), each a kernel boundary + one memory "
                                                           "round trip; algorithmic bytes = SURVEY.md 8d's figure for the REFERENCE formulation (float32 weights + "
                                                           "memories x 9 layers + latents) -- a rate of that figure, NOT measured traffic of this engine (the folded, "
                                                           "hoisted row-tile path moves fewer bytes, mostly L2-resident split-pair weights), hence no `frac`"}}
        # ---- WEG at the product shape
        gw = torch.Generator().manual_seed(9)
        enc_w = [torch.randn(1, s, 512, generator=gw).to(device) for s in S]
        mask_w = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].to(device)}
        lat_w = torch.randn(1, L, 128, generator=gw).to(device)

        def ev(same, t=500):
            return weg.loss_and_grad(model, lat_w, t, enc_w, mask_w, [[3, 9, 14]], same_conditioning=same)
        for _ in range(3):
            ev(False)
        sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ev(False)
        sync()
        t_full = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            ev(True)
        sync()
        t_same = (time.perf_counter() - t0) / 20
        for k in range(3):       # the guided loop's case: one evaluation per iteration, same conditioning, a new timestep every time
            ev("memories", 999 - k)
        sync()
        t0 = time.perf_counter()
        for k in range(20):
            ev("memories", 900 - 7 * k)
        sync()
        t_newt = (time.perf_counter() - t0) / 20
        g1 = torch.Generator().manual_seed(11)
        cond1 = [torch.randn(1, s, 512, generator=g1) for s in S]
        unc1 = [torch.randn(1, s, 512, generator=g1) for s in S]
        pat = {0: (3, 6), 1: (2, 6), 2: (1, 6), 3: (4, 6), 4: (5, 6)}
        enc7 = [torch.cat([(cond1[j] if c in pat[j] else unc1[j]) for c in range(G)], 0).to(device) for j in range(5)]
        mask7 = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].expand(G, 24).contiguous().to(device)}
        wp = dict(scale_factor=1000, scale_range=[1.0, 0.5], max_iter_to_alter=800, thresholds={0: 0.05, 200: 0.4, 400: 0.6, 600: 0.8}, max_refinement_steps=300)
        sync()
        t0 = time.perf_counter()
        sample_with_weg(model, ddpm, enc7, mask7, [[3, 9, 14]], wp, B=1, L=L, num_inference_steps=1000, seed=1)
        sync()
        out["weg_b1_product_shape"] = {"eval_ms": t_full * 1e3, "eval_same_conditioning_ms": t_same * 1e3, "eval_new_timestep_same_memories_ms": t_newt * 1e3,
                                       "guided_utterance_s": time.perf_counter() - t0,
                                       "note": "cfd_weg_eval on the row-tile kernels (forward with saved activations, objective, float32-MFMA reverse "
                                               "sweep: ~160 launches); eval_ms = new memories (their projections and one timestep's tables are made: ~50 launches more), "
                                               "eval_new_timestep_same_memories_ms = what the guided loop does once per iteration (tables over all timesteps, built once); "
                                               "guided run = configs/assets.yaml:18-23 WEG parameters on random weights"}
    finally:
        L, S = L0, S0
    return out


def gpu_state(device):
    """Clock / power / temperature of the GPU right now (amdsmi through torch.cuda; None where the box does not answer): recorded at
    both ends of the timed region, so that a +-3 % difference between two boxes of the pool -- or a power-capped box -- can be told from a
    regression.  Read OUTSIDE the timed region."""
    st = {}
    for key, fn, scale in (("sclk_mhz", torch.cuda.clock_rate, 1.0), ("power_w", torch.cuda.power_draw, 1.0), ("temperature_c", torch.cuda.temperature, 1.0)):
        try:
            st[key] = round(float(fn(device)) * scale, 1)
        except Exception:      # noqa: BLE001  (no amdsmi, no permission: the measurement goes on without the reading)
            st[key] = None
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[device.index or 0]
        cap = amdsmi.amdsmi_get_power_cap_info(h)
        st["power_cap_w"] = round(float(cap.get("power_cap", 0)) / (1e6 if cap.get("power_cap", 0) > 1e5 else 1.0), 1)
        st["max_power_cap_w"] = round(float(cap.get("max_power_cap", 0)) / (1e6 if cap.get("max_power_cap", 0) > 1e5 else 1.0), 1)
    except Exception:          # noqa: BLE001
        st.setdefault("power_cap_w", None)
    return st


def spawn_ranks(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher environment: start N fresh worker processes (one per
    GPU, torch.distributed.run on 127.0.0.1) BEFORE anything in this process touches the GPU, forward their output
    and exit with their return code.  (Never exec from a process that has initialised HIP.)"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def selftest_cpu(args, world, rank):
    """Launch-path self test on CPU (tests/test_cabi_and_host.py): rendezvous over gloo, the collate all_gather and the
    single JSON line -- with NO denoiser work and no throughput value, so it cannot be mistaken for a measurement."""
    import torch.distributed as dist
    from convofusion_amd.distributed import gather_latents
    from convofusion_amd.distributed import sample_sharded
    seen, shard_ok = 1, True

    def stand_in(enc, masks, B, first_utterance):
        # what the sharding contract promises about a sampler: utterance u's result depends on its GLOBAL id and its own
        # conditioning only (SamplingRun keys Philox by first_utterance + local index) -- a closed form with that property
        ids = first_utterance + torch.arange(B, dtype=torch.float32)
        own = enc[0].reshape(7, B, -1)[3].sum(-1)                       # chunk 3 carries the utterance's own spkemb (make_inputs)
        return (ids * 1000.0 + own)[:, None, None].expand(B, 4, 128).contiguous()
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        local = torch.full((2, 4, 128), float(rank))
        total = gather_latents(local, 2 * world)
        seen = int(total[:, 0, 0].unique().numel())
        # rank r's latents == rows of the single-rank run of the whole batch (here: the stand-in sampler through sample_sharded)
        Bt = 3 * world + 1                                              # ragged on purpose
        g = torch.Generator().manual_seed(5)
        enc = [torch.randn(7 * Bt, 2, 8, generator=g)]
        whole = stand_in(enc, {}, Bt, 0)
        got = sample_sharded(stand_in, enc, {}, Bt)
        shard_ok = bool(torch.equal(got, whole))
        dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "metric": None, "value": None, "n_gpus": world, "ranks_seen": seen, "shards_reproduce_single_rank": shard_ok,
                          "steps": args.steps, "warmup": args.warmup}))
    assert shard_ok
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shape", default="C2", choices=["C2", "R"])
    ap.add_argument("--no-full-loop", action="store_true", help="skip the secondary whole-1000-step-run wall time")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configurations (R, C1, WEG, C4, C5) appended to the line")
    ap.add_argument("--headline-only", action="store_true", help="only the headline workload's launches (profiling passes): no zero-weight-chunk run, "
                    "no whole-run wall time, no other configurations, no CPU baseline")
    ap.add_argument("--selftest-cpu", action="store_true", help="exercise only the N-rank launch path on CPU (gloo); no measurement")
    args = ap.parse_args()
    if args.headline_only:
        args.no_cpu_baseline = args.no_full_loop = args.no_secondary = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: start it as `python bench.py --gpus N` "
                         f"or under torch.distributed.run with --nproc-per-node N")
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py measures the HIP path on an MI355X; there is no CPU fallback"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    global L, S
    if args.shape == "R":
        L, S = 16, (24, 161, 24, 8, 1)

    from convofusion_amd import scheduler
    from convofusion_amd.distributed import gather_latents
    from convofusion_amd.sampler import SamplingRun

    model = make_model(device)
    mems, masks = make_inputs(B_PER_GPU, device, seed=1234 + rank)
    sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                                  beta_schedule="scaled_linear", variance_type="fixed_small", clip_sample=True)
    n_sched = 1000
    assert args.steps + args.warmup < n_sched, "steps + warmup must leave one iteration of the 1000-step run for the profiled forward"

    def open_run(**kw):
        return SamplingRun(model, sch, mems, masks, B_PER_GPU, L, n_sched, guidance_scale=7.5, seed=0, first_utterance=rank * B_PER_GPU, **kw)

    run = open_run()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    run.steps(args.warmup)
    warm = run.read()
    if world > 1:
        gather_latents(warm, world * B_PER_GPU)   # untimed: the collective's first use sets up its channels
    torch.cuda.synchronize()
    state0 = gpu_state(device) if rank == 0 else None
    barrier()
    t0 = time.perf_counter()
    run.steps(args.steps)
    local = run.read()                         # syncs the run's stream
    t_steps = time.perf_counter() - t0         # this rank's K iterations alone (no collective, no barrier)
    total = gather_latents(local, world * B_PER_GPU) if world > 1 else local
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0 - t_steps
    barrier()
    dt = time.perf_counter() - t0
    state1 = gpu_state(device) if rank == 0 else None
    rank_ms = [1000.0 * dt / args.steps]
    rank_steps_ms, rank_gather_ms = [1000.0 * t_steps / args.steps], [1000.0 * t_gather]
    if world > 1:
        import torch.distributed as dist
        tall = [torch.zeros(3, device=device) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt, t_steps, t_gather], device=device))      # a real RCCL collective: one entry per rank that answered
        ranks_seen = len(tall)
        assert ranks_seen == dist.get_world_size() == args.gpus
        rank_ms = [1000.0 * float(t[0].item()) / args.steps for t in tall]
        rank_steps_ms = [1000.0 * float(t[1].item()) / args.steps for t in tall]
        rank_gather_ms = [1000.0 * float(t[2].item()) for t in tall]
        dt = max(float(t[0].item()) for t in tall)
        # the collated tensor holds every rank's shard at ITS rows: rank r's local latents are rows 32 r .. 32 r + 31 on every rank
        assert torch.equal(total[rank * B_PER_GPU:(rank + 1) * B_PER_GPU], local)
    assert torch.isfinite(total).all()
    assert total.shape[0] == world * B_PER_GPU

    # per-kernel-class timing with HIP events on the launch stream: ONE EAGER forward of the whole batch, every launch bracketed by
    # an event pair (cfd_profile_forward).  The brackets cost a few per cent (the classes sum to more than ms_per_step, which is
    # the replayed graph); the roofline objects below are built from these class times, i.e. slightly pessimistic.
    prof = run.profile()
    run.close()

    # secondary measurement (NOT the headline value): same job without evaluating the full-conditioning chunk,
    # whose guidance weight is 7.5 * 0 in the reference (convofusion.py:538) -- identical latents, 6/7 of the work
    dt_skip = None
    if not args.headline_only:
        run2 = open_run(skip_zero_weight_chunks=True)
        run2.steps(args.warmup)
        run2.read()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run2.steps(args.steps)
        run2.read()
        torch.cuda.synchronize()
        dt_skip = time.perf_counter() - t1
        run2.close()

    # secondary measurement (NOT the headline value): the same job with fp16 split pairs EVERYWHERE (operand policy 0: the fused
    # cross-attention's key / value tiles of the audio memory as pairs too, what every round before round 6 ran and what DDIM runs keep)
    dt_pairs = None
    if not args.headline_only:
        run4 = open_run(operands=0)
        run4.steps(args.warmup)
        run4.read()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run4.steps(args.steps)
        run4.read()
        torch.cuda.synchronize()
        dt_pairs = time.perf_counter() - t1
        run4.close()

    # secondary measurement: the ENTIRE 1000-step run of the same job (set-up, capture, 1000 replays, read), wall clock;
    # puts the sustained clock on record next to the short timed window above
    full_loop_s = None
    if not args.no_full_loop and args.shape == "C2":
        torch.cuda.synchronize()
        barrier()
        t2 = time.perf_counter()
        run3 = open_run()
        run3.steps(n_sched)
        fin = run3.read(close=True)
        torch.cuda.synchronize()
        full_loop_s = time.perf_counter() - t2
        assert torch.isfinite(fin).all()
        barrier()

    if rank == 0:
        from convofusion_amd import sampler as _sampler
        policy = int(os.environ["CFD_XA_OPERANDS"]) & 15 if "CFD_XA_OPERANDS" in os.environ else int(_sampler.OPERAND_POLICY[0])
        if policy:
            policy = 15        # (the shipped library implements the four bits together)
        # MFMAs issued per algorithmic product in the fused cross-attention: 3 with split pairs; the long memories' score products take 2
        # with single-f16 keys (policy bit 1), their P.V products 2 with single-f16 values (bit 0); memories below 128 padded keys keep 3
        sp = [(x + 31) // 32 * 32 for x in S]
        long_frac = sum(x for x in sp if x >= 128) / float(sum(x for x in sp if x > 32) + sum(x for x in sp[:4] if x <= 32)) if args.shape == "C2" else 0.0
        # (scores: 3 with pairs, 2 with single-f16 keys, 1 with single-f16 queries too; P.V likewise with values / probabilities)
        sc = 1.0 if (policy & 10) == 10 else (2.0 if policy & 2 else 3.0)
        pv = 1.0 if (policy & 5) == 5 else (2.0 if policy & 1 else 3.0)
        xa_issue = 3.0 - long_frac * (3.0 - 0.5 * (sc + pv))
        Be = G * B_PER_GPU
        U = [B_PER_GPU + 1] * 5
        canon = canonical_flops_per_step(Be, L, S)
        share0 = os.environ.get("CFD_SHARE0", "1") != "0"
        # distinct (utterance, audio instance) pairs of the 7-chunk pattern of make_inputs: the unconditional tensor + the utterance's own
        l0_pairs = 2 * B_PER_GPU if (share0 and os.environ.get("CFD_L0_DEDUP", "1") != "0" and os.environ.get("CFD_FUSED_XATTN", "1") != "0") else None
        ex = executed_gemm_flops(Be, L, S, U, shared_rows=B_PER_GPU if share0 else None, l0_pairs=l0_pairs,
                                 one_key=os.environ.get("CFD_ONE_KEY", "1") != "0" and os.environ.get("CFD_FUSED_XATTN", "1") != "0")
        classes = {k: {"ms": round(v[0], 4), "launches": v[1]} for k, v in prof.items()}
        once_per_run = {k: round(ex[k] / 1e12, 4) for k in ex if prof[k][1] == 0 and ex[k] > 0}
        for k in once_per_run:      # not launched inside an iteration (memory-side projections: made once at cfd_sample_begin)
            ex[k] = 0.0
        # dominant kernel = gemm_sp_kernel (the token-side and memory-side products: the largest share of a step); the fused
        # attention kernels (self_attn_fused_kernel = class gemm_attn, xattn_fused_kernel = class xattn) are listed per class
        gk = ("gemm_token", "gemm_mem")
        dom_ms = sum(prof[k][0] for k in gk)
        dom_n = sum(prof[k][1] for k in gk)
        achieved = sum(ex[k] for k in gk) / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic = xa_traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic_gemm.json")
        if os.path.exists(tpath) and args.shape == "C2":
            tj = json.load(open(tpath))
            traffic, xa_traffic = tj.get("bytes_per_launch_mean"), tj.get("xattn_bytes_per_launch")
        for k in ex:
            classes[k]["algorithmic_tflop"] = round(ex[k] / 1e12, 4)
            classes[k]["tflops_eager"] = round(ex[k] / (prof[k][0] * 1e-3) / 1e12, 1) if prof[k][0] > 0 else None
        xa_ms, xa_n = prof["xattn"]
        xa_ach = ex["xattn"] / (xa_ms * 1e-3) / 1e12 if xa_ms > 0 else 0.0
        # The fractions are quoted on the REPLAYED graph's clock: a class's share of the eager, event-bracketed forward times
        # ms_per_step (the timed region).  The brackets and eager launch gaps inflate the class times by a few per cent; the
        # shares are what the committed rocprofv3 kernel stats of the same command agree with (profiles/rNN_roofline.json,
        # tools/profile_post.py).  The eager figures stay as achieved_eager / frac_eager.
        class_sum = sum(v[0] for v in prof.values())
        ms_step = 1000.0 * dt / args.steps
        scale = ms_step / class_sum if class_sum > 0 else 1.0
        achieved_eager, xa_eager = achieved, xa_ach
        achieved, xa_ach = achieved_eager / scale, xa_eager / scale
        for k in classes:
            classes[k]["ms_eager"] = classes[k].pop("ms")
            classes[k]["ms"] = round(prof[k][0] * scale, 4)          # share of the timed region (replayed graph)
        for k in ex:
            classes[k]["tflops"] = round(ex[k] / (prof[k][0] * scale * 1e-3) / 1e12, 1) if prof[k][0] > 0 else None
            classes[k]["frac_of_peak"] = round(classes[k]["tflops"] / PEAK_BF16_TFLOPS, 4) if classes[k]["tflops"] else None
        xattn_roofline = {"bound": "mfma", "kernel": "xattn_fused_kernel (the largest single symbol of the trace: LayerNorm2, scores, softmax, "
                                                     "P.V and residual update of a layer's five cross-attentions in one launch)",
                          "achieved": xa_ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": xa_ach / PEAK_BF16_TFLOPS,
                          "frac_issued": xa_issue * xa_ach / PEAK_BF16_TFLOPS, "mfma_per_product": xa_issue,
                          "frac_of_issued_peak_div3": xa_ach / (PEAK_BF16_TFLOPS / 3.0),
                          "achieved_eager": xa_eager, "frac_eager": xa_eager / PEAK_BF16_TFLOPS,
                          "algorithmic_tflop_per_step": ex["xattn"] / 1e12,
                          "launches_per_step": xa_n, "avg_launch_ms": xa_ms * scale / max(xa_n, 1), "avg_launch_ms_eager": xa_ms / max(xa_n, 1),
                          "traffic": xa_traffic, "traffic_source": "profiles/hbm_traffic_gemm.json" if xa_traffic else None}
        mfma_ms = sum(prof[k][0] for k in ex) * scale
        all_mfma = sum(ex.values()) / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
        executed = sum(ex.values())
        step_roofline = {"bound": "mfma", "executed_tflop_per_step": executed / 1e12, "executed_tflops": executed / (ms_step * 1e-3) / 1e12,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": executed / (ms_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         "frac_issued": (3.0 * (executed - ex["xattn"]) + xa_issue * ex["xattn"]) / (ms_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         "canonical_tflop_per_step": canon / 1e12, "canonical_tflops": canon / (ms_step * 1e-3) / 1e12,
                         "note": "whole step on the timed region's clock: executed = algorithmic FLOPs of every product launched inside an iteration "
                                 "(after de-duplication: shared layer-0 head, layer-0 audio attention per distinct pair, no one-key tile step); "
                                 "canonical = the reference formulation's FLOPs for the same step (SURVEY.md 8d)"}
        out = {
            "metric": ("denoise-steps/sec (B=32 per GPU, 196-token latent, 1500 audio tokens, 1000-step DDPM schedule)" if args.shape == "C2" else
                       f"denoise-steps/sec at the product shape (NOT the BASELINE metric: B=32 per GPU, L={L}, S={S}, 1000-step DDPM schedule)"),
            "value": world * args.steps / dt,
            "unit": "denoise-steps/s (32-utterance batches, summed over GPUs)",
            "n_gpus": world, "ranks_seen": ranks_seen, "rank_ms_per_step": [round(x, 4) for x in rank_ms],
            "rank_ms_per_step_iterations_only": [round(x, 4) for x in rank_steps_ms],
            "rank_all_gather_ms": [round(x, 4) for x in rank_gather_ms],      # once per timed region (one all_gather of [B, L, 128] fp32 per rank), incl. the wait for the slowest rank
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 (fp16 hi/lo split operands, 3 MFMAs per product, f32 accumulate; f32-equivalent)" + (
                "" if policy == 0 else "; fused cross-attention against the long (audio) memory: single-f16 operands (" +
                ", ".join(n for b, n in ((2, "keys"), (1, "values"), (8, "queries"), (4, "probabilities")) if policy & b) +
                "), f32 accumulate, in DDPM runs (operand policy %d; 1000-step DDPM golden at this shape: 2.3e-5 from the reference, pairs 8e-6, "
                "budget 1e-3)" % policy),
            "operand_policy": policy,
            "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if args.shape == 'C2' else 'product shape (developer flag --shape R)'}: B={B_PER_GPU}/GPU synthetic, L={L}, S={S}, 7-way guidance (denoiser batch {Be}), "
                                   f"DDPM 1000-step schedule, {args.steps} timed iterations of the hipGraph-captured loop",
                       "shape": args.shape,
                       "parallelism": f"batch-shard x{world}, one all_gather of latents"},
            "utterance_steps_per_s": world * B_PER_GPU * args.steps / dt,
            "value_without_zero_weight_chunk": (args.steps / dt_skip) if dt_skip else None,
            "value_split_pairs_everywhere": (args.steps / dt_pairs) if dt_pairs else None,
            "full_loop_s": full_loop_s,
            "full_loop_steps_per_s": (n_sched / full_loop_s) if full_loop_s else None,
            "canonical_tflop_per_step": canon / 1e12,
            "canonical_tflops": world * canon * args.steps / dt / 1e12,
            "roofline": {"bound": "mfma", "kernel": "gemm_sp_kernel (the matrix products launched inside one iteration: token side; the memory-side "
                                                            "projections run once per run since round 2)", "achieved": achieved,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                         "frac_issued": 3.0 * achieved / PEAK_BF16_TFLOPS,
                         "peak_div3": PEAK_BF16_TFLOPS / 3.0, "frac_of_issued_peak_div3": achieved / (PEAK_BF16_TFLOPS / 3.0),
                         "achieved_eager": achieved_eager, "frac_eager": achieved_eager / PEAK_BF16_TFLOPS,
                         "eager_to_replay_scale": scale,
                         "algorithmic_tflop_per_step": sum(ex[k] for k in gk) / 1e12,
                         "traffic": traffic,
                         "traffic_source": "profiles/hbm_traffic_gemm.json (a committed rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE profile of this "
                                           "command, not a quantity of this run)" if traffic else None,
                         "launches_per_step": dom_n, "avg_launch_ms": dom_ms * scale / max(dom_n, 1), "avg_launch_ms_eager": dom_ms / max(dom_n, 1),
                         "all_mfma_kernels_achieved": all_mfma, "all_mfma_kernels_frac": all_mfma / PEAK_BF16_TFLOPS,
                         "note": "achieved / frac = algorithmic (single-product) FLOPs of the launched GEMMs / the class's share of ms_per_step "
                                 "(the timed region of replayed hipGraph iterations); the shares come from ONE EAGER forward with every launch "
                                 "bracketed by a HIP-event pair on the launch stream (cfd_profile_forward) -- its class times sum to more than "
                                 "ms_per_step (the brackets cost that: achieved_eager / frac_eager are the un-scaled figures), the shares are what "
                                 "the committed rocprofv3 kernel stats of this command give (profiles/rNN_roofline.json); "
                                 "each product is issued as 3 f16 MFMAs, so the MFMA pipe sees 3x this and the ceiling of the "
                                 "f16x3 instruction mix is peak / 3 = 833 TFLOP/s (frac_of_issued_peak_div3); traffic = mean HBM bytes "
                                 "per launch from rocprofv3 FETCH_SIZE (x2, gfx950) + WRITE_SIZE (profiles/)"},
            "roofline_xattn": xattn_roofline,
            "roofline_step": step_roofline,
            "kernel_classes": classes,
            "tflop_once_per_run_not_per_step": once_per_run,
            "gpu_state": {"before_timed_region": state0, "after_timed_region": state1,
                          "note": "amdsmi readings outside the timed region: the boxes of the pool differ by +-3 % on one binary"},
        }
        if world == 1 and not args.no_secondary and args.shape == "C2":
            try:                      # (secondary measurements never take the headline line down with them)
                out["other_configs"] = secondary_configs(model, device)
                out["c1_single_utterance"] = out["other_configs"].pop("c1_single_utterance")
            except Exception as e:    # noqa: BLE001
                out["other_configs"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
