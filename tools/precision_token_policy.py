"""Developer study (CPU, numpy oracle; not a test): what would an operand policy for the TOKEN-side products cost a DDPM run?

tools/precision_classes.py answered the question per class for DDIM-50 (none qualifies: DDIM amplifies ~100x).  Since round 6 a DDPM
run has an operand policy (fused cross-attention against the long memories on single-fp16 operands), and DDPM's
noise injection damps perturbations -- so the same question again for DDPM, for SETS of classes and for three operand mixes:

  a   activation operand one fp16, weight a pair           2 MFMAs per product, half the activation bytes
  w   weight operand one fp16, activation a pair           2 MFMAs per product, half the weight bytes
  aw  both one fp16: a plain fp16 product, fp32 accumulate 1 MFMA per product, half of all operand bytes

Error = rel. L2 of the final latents against the all-pairs emulation (what the HIP pipeline computes today), product shape, one utterance
x 7 guidance chunks, seeded weights.  Adoption gate of an operand policy: 3e-4 on every DDPM golden (budget 1e-3).
Usage: python tools/precision_token_policy.py [steps ...]      (default: 20 200; about ten minutes per mix on 8 cores)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import denoiser_ref, inputs, philox_ref, sampler_ref, scheduler_ref  # noqa: E402
from precision_classes import Emu, F32, pair, patched  # noqa: E402
from tests.helpers import rel_l2, state_dict  # noqa: E402

TOKEN = ["self_qk", "self_v", "self_wo", "tb", "ffn1", "ffn2"]


class Emu2(Emu):
    """``mix``: {class: "a" | "w" | "aw"}"""

    def __init__(self, sd, mix):
        super().__init__(sd)
        self.mix = dict(mix)

    def linear(self, x, w, b=None):
        cls = self.weight_class(w)
        key = (w.__array_interface__["data"][0], w.shape)
        if key not in self.wcache:
            self.wcache[key] = pair(np.ascontiguousarray(w.T))
        m = self.mix.get(cls, "")
        ah, al = pair(np.asarray(x, dtype=F32))
        wh, wl = self.wcache[key]
        if m == "aw":
            y = np.matmul(ah, wh)
        elif m == "a":
            y = np.matmul(ah, wh) + np.matmul(ah, wl)
        elif m == "w":
            y = np.matmul(ah, wh) + np.matmul(al, wh)
        else:
            y = np.matmul(ah, wh) + (np.matmul(ah, wl) + np.matmul(al, wh))
        y = y.astype(F32)
        return (y + b).astype(F32) if b is not None else y


def main():
    steps = [int(a) for a in sys.argv[1:] if a.isdigit()] or [20, 200]
    sd = state_dict()
    seed = 2024
    B, L, S, pad = 1, 16, (6, 20, 6, 8, 1), (2, 0, 1, 0, 0)
    cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=pad)
    init = philox_ref.normal_tensor(seed, 0, range(B), 1, L)
    x = np.concatenate([init] * 7)

    def run(n):
        lat, _, _ = sampler_ref.diffusion_reverse(
            lambda xx, t, e, m: denoiser_ref.denoiser_forward(sd, xx, t, e, m), scheduler_ref.DDPMSchedulerRef(), cb["memories"], cb["masks"], init,
            lambda i, t: philox_ref.normal_tensor(seed, i, range(B), 0, L), num_inference_steps=n)
        return lat

    t0 = time.time()
    with patched(Emu2(sd, {})):
        base_f, _ = denoiser_ref.denoiser_forward(sd, x, 500, cb["memories"], cb["masks"])
        base = {n: run(n) for n in steps}
    print(f"all pairs: reference built in {time.time() - t0:.0f}s", flush=True)
    plans = []
    for m in ("a", "w", "aw"):
        plans.append((f"all six token classes: {m}", {c: m for c in TOKEN}))
    for group in (["self_qk", "self_v"], ["self_qk", "self_v", "self_wo"], ["ffn1", "ffn2"], ["tb"]):
        plans.append((" + ".join(group) + ": aw", {c: "aw" for c in group}))
    for c in TOKEN:
        plans.append((c + ": w", {c: "w"}))
    print("| mix | one forward | " + " | ".join(f"DDPM-{n}" for n in steps) + " |\n|---|---|" + "---|" * len(steps), flush=True)
    for name, mix in plans:
        with patched(Emu2(sd, mix)):
            f, _ = denoiser_ref.denoiser_forward(sd, x, 500, cb["memories"], cb["masks"])
            errs = [rel_l2(run(n), base[n]) for n in steps]
        print(f"| {name} | {rel_l2(f, base_f):.2e} | " + " | ".join(f"{e:.2e}" for e in errs) + " |", flush=True)


if __name__ == "__main__":
    main()
