"""Developer tool (GPU box): per-section cycle breakdown of xattn_fused_kernel from an XA_STAMP build
(python -m convofusion_amd.build -DXA_STAMP=1 -o tools/experiments/lib_xastamp.so;
CFD_LIB=$PWD/tools/experiments/lib_xastamp.so python tools/xa_stamps.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import _lib, scheduler  # noqa: E402
from convofusion_amd.sampler import SamplingRun  # noqa: E402

if os.environ.get("SHAPE") == "R":          # the product shape with 32 utterances instead of the headline shape
    bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
mems, masks = bench.make_inputs(32, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)
run = SamplingRun(model, sch, mems, masks, 32, bench.L, 1000, guidance_scale=7.5, seed=0)
run.steps(3)
prof = run.profile()
nwg, W, NS = int(os.environ.get("NWG", "744")), 8, 16       # workgroups of the full-size launch (744 at the headline shape, 224 at SHAPE=R)
buf = torch.zeros(nwg * W * NS * 2, dtype=torch.float32, device=dev)
_lib.check(_lib.load().cfd_debug_read(model._handle, b"xa_stamps", C.c_void_p(buf.data_ptr()), buf.numel()))
st = buf.cpu().numpy().view(np.int64).reshape(nwg, W, NS).astype(np.float64)
names = ["prologue + pipeline priming", "A0a compute", "wait+barrier mid-A0", "A0b+A1 compute", "wait+barrier end-A1", "B0a compute", "final flush",
         "B0b + B1a compute", "wait+barrier mid-B1", "B1b + loop tail", "fills + softmax", "segment setup"]
if os.environ.get("DB", "1") != "0":       # the shipped single-fp16 instance: the long memories' steps are kt_step_db (two barriers); the short memories' three-barrier steps add to the same slots
    names = ["prologue (+ first tile requested)", "step tail -> B0", "wait+barrier B0 (K landed)", "K reads + score MFMAs", "wait+barrier B1 (V^T landed)", "P.V group 0", "final flush",
             "P.V groups 1-2", "(short memories: mid-B1)", "P.V group 3 + tail", "V^T reads, fills, softmax", "segment setup / drain"]
names[0] = "prologue: wait for the first tile"
names += ["prologue: half rows loaded, statistics", "prologue: barrier, first tile requested", "prologue: fragments", "prologue: c_q"]
tot = st.sum(-1)
print("xattn class ms:", prof["xattn"], " per-wave total cycles: mean %.0f  min %.0f  max %.0f" % (tot.mean(), tot.min(), tot.max()))
for k, n in enumerate(names):
    print(f"  {n:34s} {st[:, :, k].mean():10.0f} cycles  {100 * st[:, :, k].mean() / tot.mean():5.1f} %   (per step {st[:, :, k].mean() / float(os.environ.get('NSTEP', '51')):7.1f})")
