#!/usr/bin/env python
"""Heavy-tailed stress weights at the HEADLINE shape (BASELINE.json configs[1]: B = 32, L = 196, S = (32, 1500, 32, 8, 1), 7-way guidance),
from the REFERENCE itself (build container only): tests/golden/heavy_c2.npz.

  out5        reference ``Denoiser.forward`` (oracle.weights.make_state_dict_heavy, outlier factor 20; extended memory PE) on the 7 guidance
              rows of utterance 5 of the seeded batch whose DISTINCT memories carry outlier tokens (oracle.inputs.add_outlier_tokens), t = 333
  out5_f64    the same forward with the reference module in float64: the input is ill-conditioned for one guidance chunk (one query token
              meets a softmax with very large logits at layer 4), where the float32 reference itself is 3e-4 from the float64 result
  traj_*      5 guided DDIM steps (eta = 0) of the restated loop driving the reference denoiser for utterance 5 alone, outlier factor 8
              (at 20 the guided loop is chaotic: oracle/weights.py)

Rows are independent, so these pin rows 5, 37, ... of the full-size HIP forward / loop (as make_golden_c2rows.py does on uniform weights).
Usage:  python tests/golden/make_golden_heavy_c2.py"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import build_reference, ref_forward, rel  # noqa: E402
from make_golden_c2rows import utterance_rows  # noqa: E402

from oracle import denoiser_ref, inputs, philox_ref, sampler_ref, scheduler_ref, weights  # noqa: E402

torch.set_grad_enabled(False)
B, L, S, PAD, SEED, U, T_FWD = 32, 196, (32, 1500, 32, 8, 1), (8, 0, 8, 0, 0), 4321, 5, 333


def heavy_batch():
    cb = inputs.make_cfg_batch(seed=SEED, B=B, L=L, S=S, pad_tail=PAD, uncond_pad_tail=PAD)
    cb["unique"] = [inputs.add_outlier_tokens(u, SEED + j) for j, u in enumerate(cb["unique"])]
    cb["memories"] = [u[rm] for u, rm in zip(cb["unique"], cb["row_map"])]
    return cb


def main():
    cb = heavy_batch()
    mems, masks = utterance_rows(cb, U)
    out = {}
    sd = weights.extend_pe(weights.make_state_dict_heavy(seed=777), 1536)
    ref = build_reference(weights.make_state_dict_heavy(seed=777), mem_len=1536)
    x = np.concatenate([cb["init"][U:U + 1]] * 7)
    t0 = time.time()
    o, _ = ref_forward(ref, x, T_FWD, mems, masks)
    o2, _ = denoiser_ref.denoiser_forward(sd, x, T_FWD, mems, masks)
    print(f"forward: reference {time.time() - t0:.1f}s, oracle-vs-reference rel {rel(o2, o):.2e}, |out| {np.abs(o).mean():.3f}")
    out["out5"] = o
    # the same forward in float64 (the reference module in double precision): what the float32 reference itself is worth on this input
    ref64 = build_reference(weights.make_state_dict_heavy(seed=777), mem_len=1536).double()
    md = {k: (torch.from_numpy(v) if v is not None else None) for k, v in masks.items()}
    o64, _ = ref64(sample=torch.from_numpy(x).double(), timestep=torch.tensor(T_FWD), encoder_hidden_states=[torch.from_numpy(m).double() for m in mems],
                   mem_mask_dict=md)
    out["out5_f64"] = o64.numpy()
    print("per guidance chunk, float32 reference vs float64:", [f"{rel(o[c], out['out5_f64'][c]):.1e}" for c in range(7)])
    ref8 = build_reference(weights.make_state_dict_heavy(seed=777, gain=8.0), mem_len=1536)
    init = philox_ref.normal_tensor(SEED, 0, [U], 1, L)
    lat, snaps, _ = sampler_ref.diffusion_reverse(
        lambda xx, t, e, mk: ref_forward(ref8, xx, t, e, mk), scheduler_ref.DDIMSchedulerRef(), mems, masks, init,
        lambda i, t: philox_ref.normal_tensor(SEED, i, [U], 0, L), guidance_scale=7.5, num_inference_steps=50, eta=0.0, keep_steps=(1, 3, 5), stop_after=5)
    sd8 = weights.extend_pe(weights.make_state_dict_heavy(seed=777, gain=8.0), 1536)
    _, s_orc, _ = sampler_ref.diffusion_reverse(
        lambda xx, t, e, mk: denoiser_ref.denoiser_forward(sd8, xx, t, e, mk), scheduler_ref.DDIMSchedulerRef(), mems, masks, init,
        lambda i, t: philox_ref.normal_tensor(SEED, i, [U], 0, L), guidance_scale=7.5, num_inference_steps=50, eta=0.0, keep_steps=(1, 3, 5), stop_after=5)
    print(f"traj: {time.time() - t0:.1f}s; numpy oracle vs torch reference (both float32) after 1 / 3 / 5 guided steps:",
          [f"{rel(s_orc[k], snaps[k]):.1e}" for k in (1, 3, 5)])
    out.update({f"traj_step{k}": v for k, v in snaps.items()})
    out["meta"] = np.array([B, L, *S, *PAD, T_FWD, SEED, U], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "heavy_c2.npz"), **out)
    print("wrote heavy_c2.npz")


if __name__ == "__main__":
    main()
