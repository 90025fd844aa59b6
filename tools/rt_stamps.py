"""Developer tool (GPU box): time line of the row-tile launches of a captured step, from a -DRT_STAMP=1 build of the library:
    python -m convofusion_amd.build -DRT_STAMP=1 -o tools/experiments/lib_rtstamp.so
    CFD_LIB=$PWD/tools/experiments/lib_rtstamp.so python tools/rt_stamps.py
Workgroup (0, 0) of every launch records s_memrealtime (100 MHz) at entry, when its operands have arrived and at exit."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import _lib, scheduler  # noqa: E402
from convofusion_amd.sampler import SamplingRun  # noqa: E402

bench.L, bench.S = 16, (24, 161, 24, 8, 1)
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
mems, masks = bench.make_inputs(1, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)
weg_mode = len(sys.argv) > 1 and sys.argv[1] == "weg"
if weg_mode:      # one WEG evaluation (forward with saved activations, objective, reverse sweep): `python tools/rt_stamps.py weg`
    from convofusion_amd import weg
    gw = torch.Generator().manual_seed(9)
    enc_w = [torch.randn(1, s, 512, generator=gw).to(dev) for s in bench.S]
    mask_w = {"spkemb": None, "alsn": None, "apb": None, "lsnemb": None, "tlsn": (torch.arange(24) >= 17)[None].to(dev)}
    lat_w = torch.randn(1, 16, 128, generator=gw).to(dev)
    for k in range(6):
        weg.loss_and_grad(model, lat_w, 500, enc_w, mask_w, [[3, 9, 14]], same_conditioning=k > 0)
    torch.cuda.synchronize()
    handle = model.engine(dev)
else:
    run = SamplingRun(model, sch, mems, masks, 1, 16, 1000, guidance_scale=7.5, seed=0)
    run.steps(40)
    run.read()
    handle = run.handle
buf = torch.empty(4 * 4096 * 2 + 2, dtype=torch.float32, device=dev)
_lib.check(_lib.load().cfd_debug_read(handle, b"rt_ring", C.c_void_p(buf.data_ptr()), buf.numel()))
raw = buf.cpu().numpy().tobytes()
ring = np.frombuffer(raw[:4 * 4096 * 8], dtype=np.uint64).reshape(4096, 4)
seq = int(np.frombuffer(raw[4 * 4096 * 8:4 * 4096 * 8 + 4], dtype=np.uint32)[0])
n = min(seq, 4096)
rows = ring[:n] if seq <= 4096 else np.roll(ring, -(seq % 4096), axis=0)
per = 163 if weg_mode else 101                # ring records per evaluation / per step
rows = rows[-per * 3:]                       # the last three steps (evaluations)
names = {5000: "b:rows16", 5001: "b:rows32", 5002: "b:rows48(dqkv)", 5100: "b:ln16", 5110: "b:ln16+gelu", 5200: "b:tb16", 6000: "b:xdP", 6100: "b:xdy", 6200: "b:selfattn",
         0: "oproj", 1: "ffn2", 30: "embed", 110: "ffn1", 120: "final", 140: "qkv", 200: "timeblock", 1000: "selfattn", 2000: "xscore", 3000: "xpv"}
t0 = int(rows[0, 1])
prev_exit = None
agg = {}
for kid, a, b, c in rows:
    kid, a, b, c = int(kid), int(a), int(b), int(c)
    iss, kid = (kid >> 16) / 100.0, kid & 0xFFFF      # (product kernels: entry -> all loads issued, in the id's upper half)
    if kid == 3002:                                   # xpv: entry -> scores requested, cell statistics requested, V^T slices requested
        agg.setdefault("xpv-issue", []).append((0.0, a / 100.0, b / 100.0, c / 100.0))
        continue
    if kid == 6101:                                   # WEG dy kernel: entry -> loads issued, first barrier passed, cell scales applied
        agg.setdefault("b:xdy-pro", []).append((0.0, a / 100.0, b / 100.0, c / 100.0))
        continue
    if kid == 3001:                                   # xpv's first record: entry, loads issued, softmax done
        agg.setdefault("xpv-pro", []).append((0.0, (b - a) / 100.0, (c - a) / 100.0, 0.0))
        continue
    gap = (a - prev_exit) / 100.0 if prev_exit is not None else 0.0
    agg.setdefault(names.get(kid, str(kid)), []).append((gap, iss, (b - a) / 100.0, (c - b) / 100.0))
    prev_exit = c
print("per kernel type (workgroup 0): mean us  gap-before-entry | (entry->loads issued) | entry->operands | operands->exit | sum")
tot = 0.0
for k, v in agg.items():
    m = np.mean(np.array(v), axis=0)
    print(f"  {k:10s} x{len(v):3d}   {m[0]:6.2f} | ({m[1]:5.2f}) | {m[2]:6.2f} | {m[3]:6.2f} | {m[0] + m[2] + m[3]:6.2f}")
    tot += np.sum(np.array(v)[:, [0, 2, 3]])
print("total over the three steps (us):", tot, " per step:", tot / 3)
