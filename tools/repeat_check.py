"""Developer tool (GPU box): the same sampling run several times on one handle and on a fresh handle -- the final latents must be bit-identical
(a race in a kernel's pipeline shows up as a run-to-run difference long before it shows up against a tolerance).
  python tools/repeat_check.py            headline shape (B = 32, L = 196, 1500 audio keys), 1000 DDPM steps, 3 runs
  SHAPE=R python tools/repeat_check.py    product shape (B = 32, L = 16, 161 audio keys), 1000 steps, 4 runs"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler  # noqa: E402
from convofusion_amd.sampler import SamplingRun  # noqa: E402

R = os.environ.get("SHAPE") == "R"
if R:
    bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
mems, masks = bench.make_inputs(32, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)
outs = []
for k in range(4 if R else 3):
    model = bench.make_model(dev) if k in (0, 2) else model      # (runs 0-1 share a handle, run 2 starts a fresh one)
    run = SamplingRun(model, sch, mems, masks, 32, bench.L, 1000, guidance_scale=7.5, seed=7)
    run.steps(1000)
    outs.append(run.read(close=True).clone())
    print("run", k, "finite", bool(torch.isfinite(outs[-1]).all()), "identical to run 0:", bool(torch.equal(outs[-1], outs[0])), flush=True)
assert all(torch.equal(o, outs[0]) for o in outs), "run-to-run difference"
print("bit-identical")
