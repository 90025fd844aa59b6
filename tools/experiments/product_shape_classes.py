import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from convofusion_amd import scheduler
from convofusion_amd.sampler import SamplingRun
bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", variance_type="fixed_small", clip_sample=True)
for B in (8, 16, 32):
    mems, masks = bench.make_inputs(B, dev, seed=1234)
    run = SamplingRun(model, sch, mems, masks, B, 16, 1000, guidance_scale=7.5, seed=0)
    run.steps(5); run.read()
    torch.cuda.synchronize(); t0 = time.perf_counter(); run.steps(300); run.read(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    prof = run.profile()
    print("B", B, "ms/step", round(dt * 1e3, 4), {k: (round(v[0], 4), v[1]) for k, v in prof.items() if v[1]})
    run.close()
