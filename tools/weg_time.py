"""Developer tool (GPU box): one WEG objective + gradient evaluation at the product shape (B = 1, L = 16), and a guided single-utterance run.
CFD_WEG_ROWTILE=0 selects the float32 launch sequence of weg_eval.hpp for the A/B."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler, weg  # noqa: E402
from convofusion_amd.sampler import sample, sample_with_weg  # noqa: E402

bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
gw = torch.Generator().manual_seed(9)
Sw = (24, 161, 24, 8, 1)
enc_w = [torch.randn(1, s, 512, generator=gw).to(dev) for s in Sw]
mask_w = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].to(dev)}
lat_w = torch.randn(1, 16, 128, generator=gw).to(dev)
focus_w = [[3, 9, 14]]


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


out = {"rowtile": os.environ.get("CFD_WEG_ROWTILE", "1")}
out["eval_ms"] = timeit(lambda: weg.loss_and_grad(model, lat_w, 500, enc_w, mask_w, focus_w))
out["eval_same_conditioning_ms"] = timeit(lambda: weg.loss_and_grad(model, lat_w, 500, enc_w, mask_w, focus_w, same_conditioning=True))
ts = iter(range(999, 0, -1))
out["eval_new_timestep_same_memories_ms"] = timeit(lambda: weg.loss_and_grad(model, lat_w, next(ts), enc_w, mask_w, focus_w, same_conditioning="memories"))
if len(sys.argv) > 1 and sys.argv[1] == "guided":
    g1 = torch.Generator().manual_seed(11)
    cond1 = [torch.randn(1, s, 512, generator=g1) for s in Sw]
    unc1 = [torch.randn(1, s, 512, generator=g1) for s in Sw]
    pat = {0: (3, 6), 1: (2, 6), 2: (1, 6), 3: (4, 6), 4: (5, 6)}
    enc7 = [torch.cat([(cond1[j] if c in pat[j] else unc1[j]) for c in range(7)], 0).to(dev) for j in range(5)]
    mask7 = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].expand(7, 24).contiguous().to(dev)}
    sch1 = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   variance_type="fixed_small", clip_sample=True)
    wp = dict(scale_factor=1000, scale_range=[1.0, 0.5], max_iter_to_alter=800, thresholds={0: 0.05, 200: 0.4, 400: 0.6, 600: 0.8}, max_refinement_steps=300)
    sample(model, sch1, enc7, mask7, B=1, L=16, num_inference_steps=4, seed=1)
    torch.cuda.synchronize()
    t0 = time.time()
    sample_with_weg(model, sch1, enc7, mask7, [[3, 9, 14]], wp, B=1, L=16, num_inference_steps=1000, seed=1)
    torch.cuda.synchronize()
    out["guided_utterance_s"] = time.time() - t0
print(json.dumps(out))
