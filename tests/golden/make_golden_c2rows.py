#!/usr/bin/env python
"""Golden fixtures at the HEADLINE shape (BASELINE.json configs[1]: L=196, S=(32,1500,32,8,1), 7-way guidance)
generated from the REFERENCE ``Denoiser`` (imported from /root/reference; build container only).

Rows of the effective batch are independent (no BatchNorm; the guidance combine only mixes the 7 replicas of ONE
utterance), so the reference only has to evaluate the 7 guidance rows of a few utterances of the B=32 batch to pin
those rows of the full-size HIP forward / loop:

  denoiser_c2rows.npz : reference ``Denoiser.forward`` (extended memory PE) on the 7 guidance rows
                        of utterances {0, 17, 31} of the seeded B=32 batch, t = 640
  traj_c2_ddpm5.npz / traj_c2_ddim5.npz : 5 guided steps of the restated loop (oracle.sampler_ref + restated
                        diffusers-0.14.0 scheduler) driving the REFERENCE denoiser for utterance 17 alone; the
                        Philox streams are keyed by the GLOBAL utterance id, so the same numbers must come out of
                        the B=32 run's row 17 and out of a B=1 run with first_utterance=17.

Inputs and weights are regenerated from their seeds by the tests (oracle.inputs / oracle.weights); outputs only
are stored.  Usage:  python tests/golden/make_golden_c2rows.py
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

from make_golden import build_reference, ref_forward, rel  # noqa: E402  (imports the reference Denoiser)
from oracle import denoiser_ref, inputs, philox_ref, sampler_ref, scheduler_ref, weights  # noqa: E402

torch.set_grad_enabled(False)

B, L, S = 32, 196, (32, 1500, 32, 8, 1)
PAD = (8, 0, 8, 0, 0)
SEED = 1234
UTTS = (0, 17, 31)
T_FWD = 640


def utterance_rows(cb, u, B=B):
    """The 7 guidance rows (chunk-major batch: row c*B + u) of utterance u."""
    idx = np.array([c * B + u for c in range(7)])
    mems = [m[idx] for m in cb["memories"]]
    masks = {k: (v[idx] if v is not None else None) for k, v in cb["masks"].items()}
    return mems, masks


def main():
    sd = weights.make_state_dict(seed=1234)
    sd_ext = weights.extend_pe(sd, 1536)
    ref = build_reference(sd, mem_len=1536)
    cb = inputs.make_cfg_batch(seed=SEED, B=B, L=L, S=S, pad_tail=PAD, uncond_pad_tail=PAD)

    outs = {}
    for u in UTTS:
        mems, masks = utterance_rows(cb, u)
        x = np.concatenate([cb["init"][u:u + 1]] * 7)
        t0 = time.time()
        out, att = ref_forward(ref, x, T_FWD, mems, masks)
        o2, _ = denoiser_ref.denoiser_forward(sd_ext, x, T_FWD, mems, masks)
        print(f"utt {u}: reference {time.time() - t0:.1f}s, oracle-vs-reference rel {rel(o2, out):.2e}")
        outs[f"out{u}"] = out
        if u == 17:   # a slice of the audio attention of the audio-only chunk (row 2) and the text maps of the text-only chunk
            outs["att1_u17_row2_head"] = att[1][2, :, :, :96]
            outs["att1_u17_row2_rowsum"] = att[1][2].sum(-1)
            outs["att2_u17_row1"] = att[2][1]
    np.savez_compressed(os.path.join(HERE, "denoiser_c2rows.npz"), **outs,
                        meta=np.array([B, L, *S, *PAD, T_FWD, SEED, *UTTS], dtype=np.int64))

    def ref_fn(x, t, enc, masks):
        return ref_forward(ref, x, t, enc, masks)

    u = 17
    mems, masks = utterance_rows(cb, u)
    for name, sched, n in (("ddpm5", scheduler_ref.DDPMSchedulerRef(), 5), ("ddim5", scheduler_ref.DDIMSchedulerRef(), 5)):
        init = philox_ref.normal_tensor(SEED, 0, [u], 1, L)
        t0 = time.time()
        lat, snaps, _ = sampler_ref.diffusion_reverse(
            ref_fn, sched, mems, masks, init, lambda i, t: philox_ref.normal_tensor(SEED, i, [u], 0, L),
            guidance_scale=7.5, num_inference_steps=n, keep_steps=(1, 3))
        print(f"traj_c2_{name}: {time.time() - t0:.1f}s |lat| {np.abs(lat).mean():.3f}")
        np.savez_compressed(os.path.join(HERE, f"traj_c2_{name}.npz"), latents=lat,
                            **{f"step{k}": v for k, v in snaps.items()},
                            meta=np.array([B, L, *S, *PAD, n, SEED, u], dtype=np.int64))
    print("done")


if __name__ == "__main__":
    main()
