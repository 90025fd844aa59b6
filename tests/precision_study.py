"""Developer study (not a test): how GEMM operand precision propagates to the final latents.

Emulates on CPU, inside the numpy oracle, the split-precision products the HIP kernels issue
(operands split into 2 x bf16 or 2 x fp16, 3 partial products, fp32 accumulate) and reports the
per-forward and per-trajectory relative error against the plain-fp32 oracle.
Usage: python tests/precision_study.py [scheme ...]      schemes: bf16x1 bf16x3 fp16x1 fp16x3
                                                         fp16w2 : weight x activation products with the ACTIVATION as one fp16 and the
                                                                  weight as an fp16 pair (2 partial products); activation x activation
                                                                  products (attention) keep 3
"""
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import denoiser_ref, inputs, philox_ref, sampler_ref, scheduler_ref  # noqa: E402
from tests.helpers import rel_l2, state_dict  # noqa: E402

F32 = np.float32


def to_bf16(x):
    u = np.ascontiguousarray(x, dtype=F32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    return r.view(F32)


def to_fp16(x):
    return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


WEIGHT_IDS = set()


def make_mm(scheme):
    rnd = to_bf16 if scheme.startswith("bf16") else to_fp16
    mixed = scheme == "fp16w2"
    terms = 3 if mixed else int(scheme[-1])

    cache = {}

    def split(x):
        key = (x.__array_interface__["data"][0], x.shape, x.strides)
        r = x
        while r.base is not None:
            r = r.base
        big = id(r) in WEIGHT_IDS  # only views of the persistent weight arrays are cached
        if big and key in cache:
            return (*cache[key], big) if mixed else cache[key]
        h = rnd(x)
        l = rnd(x - h) if terms == 3 else None
        if big:
            cache[key] = (h, l)
        return (h, l, big) if mixed else (h, l)

    def mm(a, b):
        a = np.asarray(a, dtype=F32)
        b = np.asarray(b, dtype=F32)
        if mixed:
            (ah, al, aw), (bh, bl, bw) = split(a), split(b)
            if bw and not aw:      # activation x weight: a_h (w_h + w_l)
                return (np.matmul(ah, bh) + np.matmul(ah, bl)).astype(F32)
            if aw and not bw:
                return (np.matmul(ah, bh) + np.matmul(al, bh)).astype(F32)
            return (np.matmul(ah, bh) + (np.matmul(ah, bl) + np.matmul(al, bh))).astype(F32)
        (ah, al), (bh, bl) = split(a), split(b)
        if terms == 1:
            return np.matmul(ah, bh)
        return (np.matmul(ah, bh) + (np.matmul(ah, bl) + np.matmul(al, bh))).astype(F32)
    return mm


def patched(scheme):
    mm = make_mm(scheme)
    orig_linear, orig_matmul = denoiser_ref.linear, np.matmul

    class P:
        def __enter__(self):
            def linear(x, w, b=None):
                y = mm(x, w.T)
                return (y + b).astype(F32) if b is not None else y
            denoiser_ref.linear = linear
            denoiser_ref.np = type("npx", (), {})()  # shadow module-level np inside denoiser_ref
            for k in dir(np):
                try:
                    setattr(denoiser_ref.np, k, getattr(np, k))
                except Exception:
                    pass
            denoiser_ref.np.matmul = mm

        def __exit__(self, *a):
            denoiser_ref.linear = orig_linear
            denoiser_ref.np = np
    return P()


def main():
    schemes = sys.argv[1:] or ["bf16x3", "fp16x3"]
    sd = state_dict()
    WEIGHT_IDS.update(id(v) for v in sd.values())
    seed = 2024
    B, L, S, pad = 1, 16, (6, 20, 6, 8, 1), (2, 0, 1, 0, 0)
    cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=pad)
    init = philox_ref.normal_tensor(seed, 0, range(B), 1, L)

    def run(kind, n):
        sched = scheduler_ref.DDIMSchedulerRef() if kind == "ddim" else scheduler_ref.DDPMSchedulerRef()
        fn = lambda x, t, e, m: denoiser_ref.denoiser_forward(sd, x, t, e, m)
        keep = tuple(sorted({1, n // 4, n // 2, n}))
        lat, snaps, _ = sampler_ref.diffusion_reverse(
            fn, sched, cb["memories"], cb["masks"], init,
            lambda i, t: philox_ref.normal_tensor(seed, i, range(B), 0, L), num_inference_steps=n, keep_steps=keep)
        return snaps

    plans = [("ddpm", 20), ("ddim", 50), ("ddpm", 200)]
    if os.environ.get("FULL"):
        plans.append(("ddpm", 1000))
    base = {p: run(*p) for p in plans}
    x = np.concatenate([init] * 7)
    o_ref, _ = denoiser_ref.denoiser_forward(sd, x, 500, cb["memories"], cb["masks"])
    for s in schemes:
        with patched(s):
            o, _ = denoiser_ref.denoiser_forward(sd, x, 500, cb["memories"], cb["masks"])
            print(f"{s}: one forward rel {rel_l2(o, o_ref):.2e}")
            for p in plans:
                t0 = time.time()
                sn = run(*p)
                print(f"  {p}: " + "  ".join(f"step{k}: {rel_l2(sn[k], base[p][k]):.2e}" for k in sorted(sn)),
                      f"({time.time()-t0:.0f}s)")


if __name__ == "__main__":
    main()
