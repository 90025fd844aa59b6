"""Developer tool (GPU box): wall time of one utterance at the product shape, 1000-step DDPM, through sample() (C1 on the GPU)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler  # noqa: E402
from convofusion_amd.sampler import sample  # noqa: E402

bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mems, masks = bench.make_inputs(B, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)
sample(model, sch, mems, masks, B=B, L=16, num_inference_steps=4, seed=0)
ts = []
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for _ in range(REPS):
    torch.cuda.synchronize()
    t0 = time.time()
    sample(model, sch, mems, masks, B=B, L=16, num_inference_steps=1000, seed=0)
    torch.cuda.synchronize()
    ts.append(time.time() - t0)
print(f"B={B} product shape, 1000 steps: {min(ts):.3f} s  (CFD_FUSED_XATTN_MIN_WGS={os.environ.get('CFD_FUSED_XATTN_MIN_WGS', 'default')})")
