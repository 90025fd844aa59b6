#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container (needs /root/reference on disk; it is imported, never copied).
The reference ``Denoiser`` class (convofusion/models/architectures/denoiser.py) is instantiated
with the configs/modules/denoiser.yaml values, loaded (strict) with oracle.weights' seeded
state-dict, and run on oracle.inputs' seeded inputs.  Outputs only are stored (inputs and weights
are regenerated from their seeds by the tests).  The sampler loop
(convofusion/models/modeltype/convofusion.py:391-549) cannot be imported (pytorch_lightning etc.
are missing), so trajectories are produced by oracle.sampler_ref driving the *reference* Denoiser
with oracle.scheduler_ref (restated diffusers 0.14.0) and oracle.philox_ref noise.

Usage:  python tests/golden/make_golden.py            (writes *.npz next to this file)
"""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from convofusion.models.architectures.denoiser import Denoiser  # noqa: E402  (the reference)
from convofusion.models.operator.position_encoding import PositionEmbeddingSine1D  # noqa: E402

from oracle import denoiser_ref, inputs, philox_ref, sampler_ref, scheduler_ref, weights  # noqa: E402

torch.set_grad_enabled(False)


def build_reference(sd_np, mem_len=1024):
    abl = SimpleNamespace(SKIP_CONNECT=True, VAE_TYPE="convofusion", DIFF_PE_TYPE="convofusion", CAUSAL_ATTN=False)
    m = Denoiser(ablation=abl, nfeats=189, condition="text+audio", latent_dim=[1, 128], ff_size=1024,
                 num_layers=9, num_heads=4, dropout=0.1, normalize_before=True, activation="gelu",
                 flip_sin_to_cos=True, return_intermediate_dec=False, position_embedding="sine",
                 arch="trans_dec", freq_shift=0, guidance_scale=7.5, guidance_uncondp=0.1,
                 text_encoded_dim=512, audio_encoded_dim=512, nclasses=10)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    missing = m.load_state_dict(sd, strict=True)
    if mem_len > 1024:  # the reference crashes for S > 1024; extend the closed-form buffer
        m.mem_pos = PositionEmbeddingSine1D(512, max_len=mem_len)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(m.state_dict().keys()) == [k for k, _ in weights.key_shapes()]
    assert all(tuple(v.shape) == dict(weights.key_shapes())[k] for k, v in m.state_dict().items()) or mem_len > 1024
    return m.eval()


def ref_forward(m, sample, t, mems, masks):
    md = {k: (torch.from_numpy(v) if v is not None else None) for k, v in masks.items()}
    out, att = m(sample=torch.from_numpy(sample), timestep=torch.tensor(t),
                 encoder_hidden_states=[torch.from_numpy(x) for x in mems], mem_mask_dict=md)
    return out.numpy(), [a.numpy() for a in att]


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))


def main():
    torch.manual_seed(0)
    sd = weights.make_state_dict(seed=1234)
    sd_sharp = weights.make_state_dict(seed=4321, sharp=4.0)
    ref = build_reference(sd)
    ref_sharp = build_reference(sd_sharp)

    # ---- scheduler tables, computed exactly as diffusers does (torch float32) -----------------
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    np.savez_compressed(os.path.join(HERE, "scheduler_tables.npz"), betas=betas.numpy(), alphas_cumprod=ac.numpy())
    s = scheduler_ref.DDPMSchedulerRef()
    print("tables: betas bit-equal", np.array_equal(s.betas, betas.numpy()),
          "acp bit-equal", np.array_equal(s.alphas_cumprod, ac.numpy()))

    # ---- single forwards -----------------------------------------------------------------------
    cases = {
        # name: (weights, Be, L, S, pad_tail, t, scale)
        "tiny": (sd, ref, 7, 16, (6, 20, 6, 8, 1), (2, 0, 1, 0, 0), 37, 1.0),
        "tiny_sharp": (sd_sharp, ref_sharp, 5, 8, (9, 33, 12, 8, 1), (3, 0, 4, 0, 0), 999, 2.0),
        "real": (sd, ref, 14, 16, (24, 161, 24, 8, 1), (5, 0, 7, 0, 0), 500, 1.0),
        "oddlen": (sd_sharp, ref_sharp, 3, 34, (5, 70, 3, 8, 1), (1, 0, 1, 0, 0), 0, 1.0),
    }
    for name, (w, m, Be, L, S, pad, t, scale) in cases.items():
        inp = inputs.make_plain_batch(seed=100 + len(name), Be=Be, L=L, S=S, pad_tail=pad, scale=scale)
        t0 = time.time()
        out, att = ref_forward(m, inp["sample"], t, inp["memories"], inp["masks"])
        taps = {}
        o2, a2 = denoiser_ref.denoiser_forward(w, inp["sample"], t, inp["memories"], inp["masks"], taps=taps)
        print(f"{name}: ref {time.time()-t0:.2f}s  oracle-vs-ref out rel {rel(o2, out):.2e}",
              " att max abs", max(float(np.abs(x - y).max()) for x, y in zip(a2, att)))
        if name == "real":  # keep the fixture small: audio attention only for the first 2 rows
            att = [att[0], att[1][:2], att[2], att[3], att[4]]
        np.savez_compressed(os.path.join(HERE, f"denoiser_{name}.npz"), out=out,
                            **{f"att{j}": att[j] for j in range(5)},
                            meta=np.array([Be, L, *S, *pad, t], dtype=np.int64), scale=np.float32(scale))

    # synthetic long-memory case (extended PE), small batch
    sd_ext = weights.extend_pe(sd, 1536)
    ref_ext = build_reference(sd, mem_len=1536)
    Be, L, S, pad, t = 2, 32, (32, 1500, 32, 8, 1), (8, 0, 8, 0, 0), 250
    inp = inputs.make_plain_batch(seed=77, Be=Be, L=L, S=S, pad_tail=pad)
    out, att = ref_forward(ref_ext, inp["sample"], t, inp["memories"], inp["masks"])
    o2, a2 = denoiser_ref.denoiser_forward(sd_ext, inp["sample"], t, inp["memories"], inp["masks"])
    print(f"synth: oracle-vs-ref out rel {rel(o2, out):.2e}")
    np.savez_compressed(os.path.join(HERE, "denoiser_synth.npz"), out=out, att1_rowsum=att[1].sum(-1),
                        att1_head=att[1][:, :, :, :64], att0=att[0],
                        meta=np.array([Be, L, *S, *pad, t], dtype=np.int64), scale=np.float32(1.0))

    # ---- trajectories: restated loop + restated scheduler driving the REFERENCE denoiser --------
    def ref_fn(model):
        def fn(x, t, enc, masks):
            return ref_forward(model, x, t, enc, masks)
        return fn

    seed = 2024
    for name, sched, n_steps, eta, B, L, S, pad, keep in [
        ("ddpm1000", scheduler_ref.DDPMSchedulerRef(), 1000, 0.0, 1, 16, (6, 20, 6, 8, 1), (2, 0, 1, 0, 0), (1, 10, 100, 500, 1000)),
        ("ddpm20_b2", scheduler_ref.DDPMSchedulerRef(), 20, 0.0, 2, 16, (24, 161, 24, 8, 1), (4, 0, 6, 0, 0), (1, 5, 20)),
        ("ddim50", scheduler_ref.DDIMSchedulerRef(), 50, 0.0, 2, 16, (6, 20, 6, 8, 1), (2, 0, 1, 0, 0), (1, 10, 50)),
    ]:
        cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=pad)
        init = philox_ref.normal_tensor(seed, 0, range(B), 1, L)
        t0 = time.time()
        lat, snaps, _ = sampler_ref.diffusion_reverse(
            ref_fn(ref), sched, cb["memories"], cb["masks"], init,
            lambda i, t: philox_ref.normal_tensor(seed, i, range(B), 0, L),
            guidance_scale=7.5, num_inference_steps=n_steps, eta=eta, keep_steps=keep)
        print(f"traj {name}: {time.time()-t0:.1f}s  |lat| {np.abs(lat).mean():.3f}")
        np.savez_compressed(os.path.join(HERE, f"traj_{name}.npz"), latents=lat,
                            **{f"step{k}": v for k, v in snaps.items()},
                            meta=np.array([B, L, *S, *pad, n_steps, seed], dtype=np.int64))

    # in-painting rollout window (unbounded_synthesis.py:28-187): 8 preseq tokens, 30 steps
    B, L, S, pad = 2, 16, (6, 20, 6, 8, 1), (2, 0, 1, 0, 0)
    cb = inputs.make_cfg_batch(seed=seed + 1, B=B, L=L, S=S, pad_tail=pad)
    init = philox_ref.normal_tensor(seed + 1, 0, range(B), 1, L)
    preseq = (0.5 * philox_ref.normal_tensor(seed + 1, 7, range(B), 2, 8)).astype(np.float32)
    lat, snaps, _ = sampler_ref.diffusion_reverse(
        ref_fn(ref), scheduler_ref.DDPMSchedulerRef(), cb["memories"], cb["masks"], init,
        lambda i, t: philox_ref.normal_tensor(seed + 1, i, range(B), 0, L),
        guidance_scale=7.5, num_inference_steps=25, preseq=preseq, keep_steps=(1, 2, 25))
    np.savez_compressed(os.path.join(HERE, "traj_inpaint25.npz"), latents=lat, preseq=preseq,
                        **{f"step{k}": v for k, v in snaps.items()},
                        meta=np.array([B, L, *S, *pad, 25, seed + 1], dtype=np.int64))
    print("done")


if __name__ == "__main__":
    main()
