import os, sys, torch
sys.path.insert(0, os.getcwd())
from oracle import inputs
from convofusion_amd.denoiser import Denoiser
from convofusion_amd.sampler import sample
from convofusion_amd import scheduler
from tests.gpu_helpers import ABL, DENOISER_KW, SCHED_KW, hip_denoiser, to_dev
B, L, S = 3, 16, (24, 161, 24, 8, 1)
def mk(seed):
    cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=(4, 0, 6, 0, 0))
    return [to_dev(x) for x in cb["memories"]], {k: to_dev(v) for k, v in cb["masks"].items()}
sch = lambda: scheduler.DDPMScheduler(**SCHED_KW)
m = hip_denoiser(1234, 1.0)
def fresh():
    m2 = Denoiser(ablation=ABL, **DENOISER_KW)
    m2.load_state_dict(m.state_dict(), strict=True)
    return m2.cuda().eval()
mems8, masks8 = mk(8)
mems6, masks6 = mk(6)
ref = sample(fresh(), sch(), mems8, masks8, B=B, L=L, num_inference_steps=4, seed=11)
for name, pre in [("none", None), ("skip", dict(skip_zero_weight_chunks=True)), ("plain", dict()), ("nodedup", dict(dedup=False))]:
    h = fresh()
    if pre is not None:
        sample(h, sch(), mems6, masks6, B=B, L=L, num_inference_steps=4, seed=11, **pre)
    a = sample(h, sch(), mems8, masks8, B=B, L=L, num_inference_steps=4, seed=11)
    print(name, torch.equal(a, ref), float((a - ref).abs().max()))
