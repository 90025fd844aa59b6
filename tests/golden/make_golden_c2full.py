#!/usr/bin/env python
"""FULL-LENGTH golden trajectories at the HEADLINE shape (BASELINE.json configs[1]: B=32, L=196, S=(32,1500,32,8,1),
7-way guidance), generated from the REFERENCE ``Denoiser`` (imported from /root/reference; build container only).

north_star's acceptance sentence is "B=32, 196-frame latents, 1000-step DDPM, outputs within 1e-3 rel of reference".
make_golden_c2rows.py pins 5 steps of that loop; this script pins all of it for one utterance:

  traj_c2_ddpm1000.npz : utterance 17 of the seeded B=32 batch, 1000 DDPM steps of the restated loop
                         (oracle.sampler_ref + restated diffusers-0.14.0 scheduler) driving the REFERENCE denoiser;
                         snapshots after steps 1, 10, 100, 500 and the final latents
  traj_c2_ddim50.npz   : the same utterance, 50 DDIM steps (eta = 0); snapshots after steps 1, 10, 25

Rows of the effective batch are independent and the Philox streams are keyed by the GLOBAL utterance id, so the B=32
run's row 17 and a one-utterance shard with first_utterance=17 must reproduce these numbers.  ~1.2 s per reference step
on 8 cores: ~20 minutes for the DDPM run, ~1 minute for DDIM.

Usage:  python tests/golden/make_golden_c2full.py [ddim50] [ddpm1000]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import build_reference, ref_forward  # noqa: E402  (imports the reference Denoiser)
from make_golden_c2rows import B, L, S, PAD, SEED, utterance_rows  # noqa: E402
from oracle import inputs, philox_ref, sampler_ref, scheduler_ref, weights  # noqa: E402

torch.set_grad_enabled(False)

U = 17
CASES = {
    "ddim50": (scheduler_ref.DDIMSchedulerRef, 50, (1, 10, 25)),
    "ddpm1000": (scheduler_ref.DDPMSchedulerRef, 1000, (1, 10, 100, 500)),
}


def main():
    which = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    sd = weights.make_state_dict(seed=1234)
    ref = build_reference(sd, mem_len=1536)
    cb = inputs.make_cfg_batch(seed=SEED, B=B, L=L, S=S, pad_tail=PAD, uncond_pad_tail=PAD)
    mems, masks = utterance_rows(cb, U)
    calls = [0]
    t_start = [time.time()]

    def ref_fn(x, t, enc, m):
        calls[0] += 1
        if calls[0] % 50 == 0:
            print(f"  step {calls[0]}  {time.time() - t_start[0]:.0f}s", flush=True)
        return ref_forward(ref, x, t, enc, m)

    for name in which:
        cls, n, keep = CASES[name]
        calls[0], t_start[0] = 0, time.time()
        init = philox_ref.normal_tensor(SEED, 0, [U], 1, L)
        lat, snaps, _ = sampler_ref.diffusion_reverse(
            ref_fn, cls(), mems, masks, init, lambda i, t: philox_ref.normal_tensor(SEED, i, [U], 0, L),
            guidance_scale=7.5, num_inference_steps=n, keep_steps=keep)
        print(f"traj_c2_{name}: {time.time() - t_start[0]:.1f}s |lat| {np.abs(lat).mean():.3f}", flush=True)
        np.savez_compressed(os.path.join(HERE, f"traj_c2_{name}.npz"), latents=lat,
                            **{f"step{k}": v for k, v in snaps.items()},
                            meta=np.array([B, L, *S, *PAD, n, SEED, U], dtype=np.int64))
    print("done")


if __name__ == "__main__":
    main()
