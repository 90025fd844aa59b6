"""Timings of the BASELINE.json configurations that are not the bench.py line (developer tool; bench.py stays the
contract).  Prints one JSON object:
  C4  50-step DDIM, B=32, synthetic 196-token latents + 1500 audio tokens: end-to-end latency of the captured loop
      (cfd_sample_begin .. cfd_sample_read, i.e. including table construction, warm-up iteration and capture)
  C5  dyadic reactive path, B=16 per side, same shape (speaker memory = partner projection, 196 keys)
  audio encoder: AudioConvEncoder over (B+1) x 1500 Mel frames (the conditioning producer of one batch)
  VAE decode: ConvoFusionVae.decode of one batch (B=32, 128 frames)
  C1 on the GPU: one utterance at the product shape, 1000-step DDPM end to end, without / with WEG
  WEG: one objective + gradient evaluation (convofusion_amd.weg.loss_and_grad) at the product shape, B=1
usage: python tools/bench_configs.py   (on the GPU box)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler  # noqa: E402
from convofusion_amd.conditioning import AudioConvEncoder, default_fuser  # noqa: E402
from convofusion_amd.dyadic import DyadicRun  # noqa: E402
from convofusion_amd.sampler import sample  # noqa: E402

dev = torch.device("cuda:0")
SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=True)
out = {}


def sync():
    torch.cuda.synchronize(dev)


model = bench.make_model(dev)
mems, masks = bench.make_inputs(32, dev, 1234)
L = bench.L

# ---- C4: DDIM-50 latency
sch = scheduler.DDIMScheduler(**SCHED, set_alpha_to_one=True, steps_offset=0)
sample(model, sch, mems, masks, B=32, L=L, num_inference_steps=2, seed=0)   # first use: weight upload, kernel attributes
sync()
lat = []
for r in range(3):
    t0 = time.time()
    x = sample(model, sch, mems, masks, B=32, L=L, num_inference_steps=50, eta=0.0, seed=r)
    sync()
    lat.append(time.time() - t0)
out["C4_ddim50_B32"] = {"latency_s_end_to_end": min(lat), "all": lat, "steps_per_s": 50 / min(lat),
                        "note": "sample(): de-duplication of the replicated batch, tables, warm-up iteration, capture, 50 replays, read"}

# ---- C5: dyadic, B=16 per side
B = 16
g = torch.Generator().manual_seed(5)
S = bench.S
cond = lambda: [torch.randn(B, L if j == 0 else S[j], 512, generator=g).to(dev) for j in range(5)]
uncond = [torch.randn(1, L if j == 0 else S[j], 512, generator=g).to(dev) for j in range(5)]
model_b = bench.make_model(dev)
fuser = default_fuser().to(dev).eval()
run = DyadicRun(model, model_b, scheduler.DDPMScheduler(**SCHED), fuser, cond(), cond(), uncond, B, L, 1000, seed=3)
run.steps(3)
sync()
t0 = time.time()
n = 20
run.steps(n)
run.read()
sync()
dt = (time.time() - t0) / n
out["C5_dyadic_B16x2"] = {"ms_per_lockstep_iteration": dt * 1e3, "iterations_per_s": 1 / dt,
                          "note": "one iteration = both sides' guided denoising step + two partner projections (host-driven: "
                                  "2 reads, 4 cfd_linear_act launches, 2 graph replays)"}
run.read(close=True)

# ---- audio encoder over one batch's Mel frames
enc = AudioConvEncoder(input_size=80, hidden_size=256, latent_dim=512, max_seq_len=128, fps=25, sample_rate=16000, hop_length=160).to(dev).eval()
mel = torch.randn(33, 1500, 80, device=dev)
enc(mel)
sync()
t0 = time.time()
for _ in range(10):
    enc(mel)
sync()
dt = (time.time() - t0) / 10
fl = 2.0 * 33 * 1500 * (80 * 256 + 256 * 512 + 512 * 512)
out["audio_encoder_33x1500"] = {"ms": dt * 1e3, "tflops_fp32": fl / dt / 1e12}
# ---- VAE decode of one batch (the step after the loop): B=32, 8 chunks -> 128 frames
from types import SimpleNamespace  # noqa: E402
from convofusion_amd.vae import ConvoFusionVae  # noqa: E402
vae = ConvoFusionVae(ablation=SimpleNamespace(MLP_DIST=False, PE_TYPE="convofusion"), nfeats=189, latent_dim=[1, 128], ff_size=1024,
                     num_layers=5, num_heads=2, arch="encoder_decoder", normalize_before=True, activation="gelu",
                     position_embedding="sine").to(dev).eval()
z = torch.randn(2, 32, 8, 128, device=dev)
vae.decode(z, [128] * 32)
sync()
t0 = time.time()
for _ in range(5):
    vae.decode(z, [128] * 32)
sync()
out["vae_decode_B32_128frames"] = {"ms": (time.time() - t0) / 5 * 1e3, "note": "~190 small float32 launches driven from Python"}
# ---- word-excitation guidance: one objective + gradient evaluation on the text-only chunk, product shape (B=1, L=16)
from convofusion_amd import weg  # noqa: E402
gw = torch.Generator().manual_seed(9)
Sw = (24, 161, 24, 8, 1)
enc_w = [torch.randn(1, s, 512, generator=gw).to(dev) for s in Sw]
mask_w = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].to(dev)}
lat_w = torch.randn(1, 16, 128, generator=gw).to(dev)
focus_w = [[3, 9, 14]]
def timeit(fn, n=20):
    fn()
    sync()
    t0 = time.time()
    for _ in range(n):
        fn()
    sync()
    return (time.time() - t0) / n * 1e3


out["weg_loss_and_grad_B1_L16"] = {
    "ms": timeit(lambda: weg.loss_and_grad(model, lat_w, 500, enc_w, mask_w, focus_w)),
    "stepwise_from_python_ms": timeit(lambda: weg.loss_and_grad_stepwise(model, lat_w, 500, enc_w, mask_w, focus_w), 5),
    "note": "cfd_weg_eval: ~510 small float32 launches (forward with saved activations, objective, backward sweep) enqueued by the "
            "library, one host sync for the loss; stepwise = the same kernels, one C call per launch from Python"}
# ---- C1 on the GPU: ONE utterance at the product shape (B=1, L=16, 161 audio tokens), the full 1000-step DDPM loop end to end,
#      without and with word-excitation guidance (the reference's defaults: 800 altered iterations, 4 threshold iterations)
from convofusion_amd.sampler import sample_with_weg  # noqa: E402
g1 = torch.Generator().manual_seed(11)
cond1 = [torch.randn(1, s, 512, generator=g1) for s in Sw]
unc1 = [torch.randn(1, s, 512, generator=g1) for s in Sw]
pat = {0: (3, 6), 1: (2, 6), 2: (1, 6), 3: (4, 6), 4: (5, 6)}        # chunks that carry the conditional memory (convofusion.py:909-929)
enc7 = [torch.cat([(cond1[j] if c in pat[j] else unc1[j]) for c in range(7)], 0).to(dev) for j in range(5)]
mask7 = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].expand(7, 24).contiguous().to(dev)}
sch1 = scheduler.DDPMScheduler(**SCHED)
sample(model, sch1, enc7, mask7, B=1, L=16, num_inference_steps=4, seed=1)
sync()
t0 = time.time()
sample(model, sch1, enc7, mask7, B=1, L=16, num_inference_steps=1000, seed=1)
sync()
t_plain = time.time() - t0
wp = dict(scale_factor=1000, scale_range=[1.0, 0.5], max_iter_to_alter=800, thresholds={0: 0.05, 200: 0.4, 400: 0.6, 600: 0.8}, max_refinement_steps=300)
t0 = time.time()
sample_with_weg(model, sch1, enc7, mask7, [[3, 9, 14]], wp, B=1, L=16, num_inference_steps=1000, seed=1)
sync()
t_weg = time.time() - t0
out["C1_single_utterance_ddpm1000"] = {"seconds_end_to_end": t_plain, "steps_per_s": 1000 / t_plain, "with_weg_seconds": t_weg,
                                       "note": "B=1 product shape (L=16, S=(24,161,24,8,1)); WEG with configs/assets.yaml:18-23 parameters on random weights "
                                               "(the threshold iterations run their refinement loops to the 300-evaluation cap or until the objective falls)"}
print(json.dumps(out))
