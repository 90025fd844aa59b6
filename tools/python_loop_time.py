"""Developer tool (GPU box): what a caller gets WITHOUT convofusion_amd.install() -- only the yaml edits of INTEGRATION.md sections 1-2: the
reference's own Python loop (convofusion.py:499-544: replicate x7, denoiser, guidance combine, scheduler.step) with the HIP Denoiser.forward
and scheduler underneath -- against the installed (captured) loop, one utterance at the product shape.  ms per iteration."""
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
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mems, masks = bench.make_inputs(B, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)


def reference_style_loop(n):
    sch.set_timesteps(1000)
    latents = torch.randn((B, 16, 128), device=dev)
    for t in sch.timesteps[:n]:
        x = torch.cat([latents] * 7)                                                    # :499-501
        with torch.no_grad():
            noise_pred, att = model(sample=x, timestep=t, encoder_hidden_states=mems, mem_mask_dict=masks)   # :507-513
        u, tx, a, s, p, i, f = noise_pred.chunk(7)                                       # :527-541
        noise_pred = u + 7.5 * (tx - u) + 7.5 * (a - u) + 7.5 * (s - u) + 7.5 * (p - u) + 7.5 * (i - u) + 7.5 * 0 * (f - u)
        latents = sch.step(noise_pred, t, latents).prev_sample                           # :544
    return latents


reference_style_loop(5)
torch.cuda.synchronize()
t0 = time.time()
reference_style_loop(N)
torch.cuda.synchronize()
t_py = (time.time() - t0) / N
sample(model, sch, mems, masks, B=B, L=16, num_inference_steps=4, seed=0)
torch.cuda.synchronize()
t0 = time.time()
sample(model, sch, mems, masks, B=B, L=16, num_inference_steps=1000, seed=0)
torch.cuda.synchronize()
t_inst = (time.time() - t0) / 1000
print(f"B={B} product shape: reference-style Python loop on the HIP denoiser {t_py * 1e3:.3f} ms per iteration; installed (captured) loop {t_inst * 1e3:.3f} ms")
