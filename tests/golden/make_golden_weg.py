#!/usr/bin/env python
"""Generate tests/golden/weg.npz from the REFERENCE itself (build container only; /root/reference is imported,
never copied): torch autograd through the reference ``Denoiser`` and the reference
``convofusion.models.tools.word_excitation_guidance`` functions, as the WEG branch of the loop calls them
(convofusion/models/modeltype/convofusion.py:447-495).  Weights / inputs are regenerated from their seeds by the
tests; only losses, max-attention values and gradients are stored.

Usage:  python tests/golden/make_golden_weg.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

import convofusion.models.tools.word_excitation_guidance as weg  # noqa: E402  (the reference)

from make_golden import build_reference, rel  # noqa: E402
from oracle import inputs, weg_ref, weights  # noqa: E402

# name: (weight seed, sharp, B, L, S, pad_tail, t, focus indices per sample, normalize_eot)
CASES = {
    "b1": (1234, None, 1, 16, (6, 20, 12, 8, 1), (2, 0, 3, 0, 0), 981, [[2, 5]], True),
    "b1_sharp": (4321, 4.0, 1, 16, (24, 161, 24, 8, 1), (5, 0, 7, 0, 0), 400, [[3, 9, 14]], True),
    "b3": (1234, None, 3, 8, (5, 33, 10, 8, 1), (1, 0, 2, 0, 0), 37, [[1, 4], [], [6]], False),
    # the heavy-tailed stress weights (outlier factor 8) and memories with outlier tokens at the product shape (round 5)
    "b1_heavy": (777, "heavy8", 1, 16, (24, 161, 24, 8, 1), (5, 0, 7, 0, 0), 600, [[3, 9, 14]], True),
}


def case_inputs(name):
    wseed, sharp, B, L, S, pad, t, focus, neot = CASES[name]
    if sharp == "heavy8":
        sd = weights.make_state_dict_heavy(seed=wseed, gain=8.0)
        inp = inputs.make_outlier_batch(seed=500 + len(name), Be=B, L=L, S=S, pad_tail=pad)
        return sd, inp, t, focus, neot
    sd = weights.make_state_dict(seed=wseed) if sharp is None else weights.make_state_dict(seed=wseed, sharp=sharp)
    inp = inputs.make_plain_batch(seed=500 + len(name), Be=B, L=L, S=S, pad_tail=pad)
    return sd, inp, t, focus, neot


def main():
    out = {}
    for name in CASES:
        sd, inp, t, focus, neot = case_inputs(name)
        m = build_reference(sd)
        lat = torch.from_numpy(inp["sample"]).clone().requires_grad_(True)
        masks = {k: (torch.from_numpy(v) if v is not None else None) for k, v in inp["masks"].items()}
        with torch.enable_grad():
            _, att = m(sample=lat, timestep=torch.tensor(t), encoder_hidden_states=[torch.from_numpy(x) for x in inp["memories"]],
                       mem_mask_dict=masks)
            eot = torch.argmax(masks["tlsn"].int(), dim=1) - 1                      # convofusion.py:460
            a = weg.aggregate_attentions(att[2])
            mx = weg.get_max_attention_at_indices(a, focus, smooth_attentions=True, normalize_eot=neot, eot_indices=eot)
            if any(len(s) == 0 for s in mx):   # the reference's empty-sample branch calls .cuda(); same value on the CPU
                torch.Tensor.cuda = lambda self, *a, **k: self
            loss, losses = weg.compute_attention_focus_loss(mx)
            grad = torch.autograd.grad(loss.requires_grad_(True), [lat], retain_graph=True)[0]
            new = weg.update_latent(lat, loss, 1000 * np.sqrt(0.9))
        out[name + ".loss"] = np.float32(loss.item())
        out[name + ".losses"] = losses.detach().numpy().astype(np.float32)
        out[name + ".max_att"] = np.array([v.item() for s in mx for v in s], dtype=np.float32)
        out[name + ".grad"] = grad.numpy()
        out[name + ".updated"] = new.detach().numpy()
        out[name + ".att_tlsn"] = att[2].detach().numpy()
        # the oracle against the reference
        l2, ls2, mx2, g2 = weg_ref.loss_and_grad(sd, inp["sample"], t, inp["memories"], inp["masks"], focus, neot, eot.numpy())
        print(f"{name}: loss ref {loss.item():.6f} oracle {float(l2):.6f}  grad rel {rel(g2, grad.numpy()):.2e}  "
              f"|grad| {np.abs(grad.numpy()).max():.3e}  max_att {out[name + '.max_att']}")
    np.savez_compressed(os.path.join(HERE, "weg.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
