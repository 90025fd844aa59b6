import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from convofusion_amd import scheduler
from convofusion_amd.sampler import SamplingRun, dedup_memories
dev = torch.device("cuda:0")
model = bench.make_model(dev)
mems, masks = bench.make_inputs(32, dev, 1234)
SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=True)
sch = scheduler.DDIMScheduler(**SCHED, set_alpha_to_one=True, steps_offset=0)
r = SamplingRun(model, sch, mems, masks, 32, bench.L, 50); r.steps(2); r.read(close=True)
torch.cuda.synchronize()
for _ in range(2):
    t0 = time.time(); u = dedup_memories(mems, masks); torch.cuda.synchronize(); t1 = time.time()
    r = SamplingRun(model, sch, u[0], u[2], 32, bench.L, 50, dedup=False, row_maps=u[1]); torch.cuda.synchronize(); t2 = time.time()
    r.steps(50); x = r.read(close=True); torch.cuda.synchronize(); t3 = time.time()
    print(f"dedup {1e3*(t1-t0):.1f} ms, begin (tables, warm-up, capture) {1e3*(t2-t1):.1f} ms, 50 steps + read {1e3*(t3-t2):.1f} ms")
