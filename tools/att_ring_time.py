"""Developer tool (GPU box): what the reference's per-iteration attention dict costs at the product shape -- a run without maps, with the
ring kept by the captured iteration (row-tile kernels up to 6 utterances, the fused cross-attention kernel's ATT instance beyond), and with
the fall-back an over-budget ring takes (one extra forward + host round trip per iteration).   python tools/att_ring_time.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import sampler, scheduler  # noqa: E402

bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mems, masks = bench.make_inputs(B, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)


def run(mode, budget=None):
    keep = sampler.ATT_RING_MAX_BYTES
    if budget is not None:
        sampler.ATT_RING_MAX_BYTES = budget
    try:
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            sampler.sample(model, sch, mems, masks, B=B, L=16, num_inference_steps=N, seed=0, return_attention=mode)
            torch.cuda.synchronize()
            best = min(best, time.time() - t0)
        return best
    finally:
        sampler.ATT_RING_MAX_BYTES = keep


t_plain, t_ring, t_fwd = run(False), run("all"), run("all", 0)
print(f"B={B} product shape, {N} steps: no maps {t_plain / N * 1e3:.3f} ms/step, ring {t_ring / N * 1e3:.3f} ms/step (x{t_ring / t_plain:.3f}), "
      f"one forward per iteration {t_fwd / N * 1e3:.3f} ms/step (x{t_fwd / t_plain:.3f})")
