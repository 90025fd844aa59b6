"""Developer experiment (GPU box): one sampling run of B utterances against two concurrent runs of B/2 on two library handles /
streams (python tools/shard2_experiment.py [R|C2])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler  # noqa: E402
from convofusion_amd.distributed import shard_cfg_batch  # noqa: E402
from convofusion_amd.sampler import SamplingRun  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "R"
if shape == "R":
    bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
B, L = 32, bench.L
mems, masks = bench.make_inputs(B, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)
N = 100 if shape == "R" else 20


def timed(runs):
    for r in runs:
        r.steps(5)
    for r in runs:
        r.read()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        for r in runs:
            r.steps(1)
    outs = [r.read() for r in runs]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for r in runs:
        r.close()
    return N / dt, torch.cat(outs, 0)


one = SamplingRun(model, sch, mems, masks, B, L, 1000, guidance_scale=7.5, seed=0)
v1, lat1 = timed([one])
for K in (2, 3, 4):
    models = [model] + [bench.make_model(dev) for _ in range(K - 1)]
    cuts = [round(B * k / K) for k in range(K + 1)]
    shards = []
    for k in range(K):
        a, b = cuts[k], cuts[k + 1]
        m = [shard_cfg_batch(x, a, b, B) for x in mems]
        mk = {n: shard_cfg_batch(v, a, b, B) for n, v in masks.items()}
        shards.append(SamplingRun(models[k], sch, m, mk, b - a, L, 1000, guidance_scale=7.5, seed=0, first_utterance=a))
    v2, lat2 = timed(shards)
    print(f"{shape}: one run {v1:.1f} steps/s; {K} concurrent shard runs {v2:.1f} steps/s; identical latents: {bool(torch.equal(lat1, lat2))}")
    del models, shards
