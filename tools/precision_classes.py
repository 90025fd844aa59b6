"""Developer study (CPU, numpy oracle; not a test): WHICH matrix products tolerate a single-fp16 activation operand?

Every product of the HIP pipeline is issued as 3 MFMAs on fp16 pairs.  With the ACTIVATION side carried as ONE
fp16 (the weight / memory side stays a pair) a product needs 2 MFMAs and half the activation bytes.  tests/precision_study.py showed
that doing this everywhere costs 3.9e-4 per forward; this tool switches ONE class of products at a time:

  self_qk  self_v  self_wo      self-attention in-projection (q, k), (v), out-projection        activation = LN1(x) / attention output
  tb                            the two TimeBlock output projections                              activation = SiLU(AdaLN(x))
  ffn1  ffn2                    feed-forward                                                      activation = LN3(x) / GELU(.)
  cross_q                       cross-attention query side (in the HIP kernel: LN2(x) x folded keys) activation = LN2(x)
  cross_out                     cross-attention out-projection + att_fuser (HIP: folded into the value path)
  score_self  pv_self           q.k^T with q single / P.v with P single (self-attention)
  score_cross pv_cross          the same for the five cross-attentions

For each class: relative L2 error of one forward, of a DDIM-50 and of a DDPM-200 trajectory against the all-pairs emulation
(the quantity that matters: DDIM amplifies a per-forward perturbation ~100x over 50 steps; budget 1e-3, wiring threshold 3e-4).
Usage: python tools/precision_classes.py [class ...]        (default: all, one at a time)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import denoiser_ref, inputs, philox_ref, sampler_ref, scheduler_ref  # noqa: E402
from tests.helpers import rel_l2, state_dict  # noqa: E402

F32 = np.float32
E = 512
CLASSES = ["self_qk", "self_v", "self_wo", "tb", "ffn1", "ffn2", "cross_q", "cross_out", "score_self", "pv_self", "score_cross", "pv_cross"]


def h16(x):
    return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


def pair(x):
    h = h16(x)
    return h, h16(x - h)


class Emu:
    """linear / matmul of oracle.denoiser_ref with fp16-pair operands; classes in ``single`` use a single-fp16 activation."""

    def __init__(self, sd, single=()):
        self.single = set(single)
        self.ranges = sorted((v.__array_interface__["data"][0], v.nbytes, k) for k, v in sd.items())
        self.wcache = {}
        self.att = "self"      # which attention the next score / P.V product belongs to
        self.n_mm = 0

    def weight_class(self, w):
        p = w.__array_interface__["data"][0]
        for a, n, k in self.ranges:
            if a <= p < a + n:
                off = (p - a) // (E * E * 4)
                if "in_proj_weight" in k:
                    self.att = "self" if "self_attn" in k else "cross"
                    self.n_mm = 0
                    if "self_attn" in k:
                        return "self_qk" if off < 2 else "self_v"
                    return "cross_q" if off == 0 else "mem"     # cross k / v projections act on the memories (memory side: pairs)
                if "out_proj.weight" in k:
                    return "self_wo" if "self_attn" in k else "cross_out"
                if "att_fuser" in k:
                    return "cross_out"
                if "out_layers" in k:
                    return "tb"
                if "linear1" in k:
                    return "ffn1"
                if "linear2" in k:
                    return "ffn2"
                return "other"
        return "other"

    def prod(self, a, b, a_single):
        ah, al = pair(a)
        bh, bl = b if isinstance(b, tuple) else pair(b)
        if a_single:
            return (np.matmul(ah, bh) + np.matmul(ah, bl)).astype(F32)
        return (np.matmul(ah, bh) + (np.matmul(ah, bl) + np.matmul(al, bh))).astype(F32)

    def linear(self, x, w, b=None):
        cls = self.weight_class(w)
        key = (w.__array_interface__["data"][0], w.shape)
        if key not in self.wcache:
            self.wcache[key] = pair(np.ascontiguousarray(w.T))
        y = self.prod(np.asarray(x, dtype=F32), self.wcache[key], cls in self.single)
        return (y + b).astype(F32) if b is not None else y

    def matmul(self, a, b):
        kind = ("score_" if self.n_mm == 0 else "pv_") + self.att
        self.n_mm += 1
        return self.prod(np.asarray(a, dtype=F32), np.asarray(b, dtype=F32), kind in self.single)


class patched:
    def __init__(self, emu):
        self.emu = emu

    def __enter__(self):
        self.keep = (denoiser_ref.linear, denoiser_ref.np)
        denoiser_ref.linear = self.emu.linear
        shadow = type("npx", (), {})()
        for k in dir(np):
            try:
                setattr(shadow, k, getattr(np, k))
            except Exception:
                pass
        shadow.matmul = self.emu.matmul
        denoiser_ref.np = shadow

    def __exit__(self, *a):
        denoiser_ref.linear, denoiser_ref.np = self.keep


def main():
    which = [c for c in sys.argv[1:] if c in CLASSES] or CLASSES
    sd = state_dict()
    seed = 2024
    B, L, S, pad = 1, 16, (6, 20, 6, 8, 1), (2, 0, 1, 0, 0)
    cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=pad)
    init = philox_ref.normal_tensor(seed, 0, range(B), 1, L)
    x = np.concatenate([init] * 7)
    plans = [("ddim", 50)] if os.environ.get("LEAN") else [("ddim", 50), ("ddpm", 200)]

    def run(kind, n):
        sched = scheduler_ref.DDIMSchedulerRef() if kind == "ddim" else scheduler_ref.DDPMSchedulerRef()
        lat, _, _ = sampler_ref.diffusion_reverse(
            lambda xx, t, e, m: denoiser_ref.denoiser_forward(sd, xx, t, e, m), sched, cb["memories"], cb["masks"], init,
            lambda i, t: philox_ref.normal_tensor(seed, i, range(B), 0, L), num_inference_steps=n)
        return lat

    with patched(Emu(sd)):
        t0 = time.time()
        base_f, _ = denoiser_ref.denoiser_forward(sd, x, 500, cb["memories"], cb["masks"])
        base = {p: run(*p) for p in plans}
        print(f"all pairs (3 products everywhere): reference built in {time.time() - t0:.0f}s", flush=True)
    f32_f, _ = denoiser_ref.denoiser_forward(sd, x, 500, cb["memories"], cb["masks"])
    print(f"all pairs vs plain fp32 oracle: forward {rel_l2(base_f, f32_f):.2e}", flush=True)
    print("| class (activation as ONE fp16) | one forward | " + " | ".join(f"{k.upper()}-{n}" for k, n in plans) + " |\n|---|---|" + "---|" * len(plans), flush=True)
    for c in which:
        with patched(Emu(sd, single=[c])):
            f, _ = denoiser_ref.denoiser_forward(sd, x, 500, cb["memories"], cb["masks"])
            errs = [rel_l2(run(*p), base[p]) for p in plans]
        print(f"| {c} | {rel_l2(f, base_f):.2e} | " + " | ".join(f"{e:.2e}" for e in errs) + " |", flush=True)


if __name__ == "__main__":
    main()
