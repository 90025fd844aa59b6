"""Developer tool (GPU box): where the time of a whole sampling call goes besides the replays -- de-duplication of the replicated
batch, cfd_sample_begin (tables, once-per-run memory projections, warm-up iteration, capture), the replays, the read.
  python tools/setup_time.py [C2|R|C1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler  # noqa: E402
from convofusion_amd.sampler import SamplingRun, dedup_memories  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = 32
if shape in ("R", "C1"):
    bench.L, bench.S = 16, (24, 161, 24, 8, 1)
if shape == "C1":
    B = 1
dev = torch.device("cuda:0")
model = bench.make_model(dev)
mems, masks = bench.make_inputs(B, dev, 1234)
SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=True)
for name, sch, n in (("DDIM-50", scheduler.DDIMScheduler(**SCHED, set_alpha_to_one=True, steps_offset=0), 50),
                     ("DDPM-1000", scheduler.DDPMScheduler(**SCHED, variance_type="fixed_small"), 1000)):
    r = SamplingRun(model, sch, mems, masks, B, bench.L, n)
    r.steps(2)
    r.read(close=True)
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.time()
        u = dedup_memories(mems, masks)
        torch.cuda.synchronize()
        t1 = time.time()
        r = SamplingRun(model, sch, u[0], u[2], B, bench.L, n, dedup=False, row_maps=u[1])
        torch.cuda.synchronize()
        t2 = time.time()
        r.steps(n)
        x = r.read(close=True)
        torch.cuda.synchronize()
        t3 = time.time()
    print(f"{shape} B={B} {name}: dedup {1e3 * (t1 - t0):.1f} ms, begin (tables, once-per-run projections, warm-up, capture) {1e3 * (t2 - t1):.1f} ms, "
          f"{n} steps + read {1e3 * (t3 - t2):.1f} ms")
