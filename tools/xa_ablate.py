"""Developer tool (GPU box): per-class HIP-event times of one eager forward of the benchmark problem -- no checks on the
values, so it also runs the XA_ABLATE builds of libcfdenoise (CFD_LIB=<variant .so>) whose results are garbage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler  # noqa: E402
from convofusion_amd.sampler import SamplingRun  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.make_model(dev)
mems, masks = bench.make_inputs(32, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)
run = SamplingRun(model, sch, mems, masks, 32, bench.L, 1000, guidance_scale=7.5, seed=0)
run.steps(3)
for _ in range(2):
    prof = run.profile()
print(os.environ.get("CFD_LIB", "default"), {k: round(v[0], 3) for k, v in prof.items()})
