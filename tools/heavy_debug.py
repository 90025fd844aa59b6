"""Developer tool (GPU box): the heavy-tailed stress forwards (tests/golden/heavy.npz) under several code paths, and -- for a failing one --
the residual stream after every sub-block against the oracle's taps (first bad stage)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd import _lib  # noqa: E402
from convofusion_amd.denoiser import Denoiser  # noqa: E402
from oracle import denoiser_ref  # noqa: E402
from tests.gpu_helpers import ABL, DENOISER_KW, dev_inputs, read_debug, to_dev  # noqa: E402
from tests.helpers import heavy_case, heavy_state_dict, rel_l2  # noqa: E402

gain = float(os.environ.get("GAIN", "20"))
sd = heavy_state_dict(gain)
m = Denoiser(ablation=ABL, **DENOISER_KW)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
m = m.cuda().eval()
m.return_attention = os.environ.get("ATT", "0") == "1"
lib = _lib.load()
for name in ("fwd_small", "fwd_tile"):
    inp, t, want, _ = heavy_case(name)
    if gain != 20.0:
        want, _ = denoiser_ref.denoiser_forward(sd, inp["sample"], t, inp["memories"], inp["masks"])
    mems, masks = dev_inputs(inp)
    x = to_dev(inp["sample"])
    with torch.no_grad():
        out, _ = m(x, torch.tensor(t), mems, mem_mask_dict=masks)
    e = rel_l2(out.cpu().numpy(), want)
    print(f"{name}: rel {e:.3e}  finite {bool(torch.isfinite(out).all())}", flush=True)
    if e > 1e-4 or os.environ.get("TAPS"):
        taps = {}
        denoiser_ref.denoiser_forward(sd, inp["sample"], t, inp["memories"], inp["masks"], taps=taps)
        Be, L = inp["sample"].shape[:2]
        stages = [(1, "x0")]
        for l in range(9):
            stages += [(2 + 4 * l, f"l{l}.after_self"), (3 + 4 * l, f"l{l}.after_tb1"), (4 + 4 * l, f"l{l}.after_cross"), (5 + 4 * l, f"l{l}.out")]
        for stage, key in stages:
            _lib.check(lib.cfd_debug_stop_stage(m._handle, stage))
            with torch.no_grad():
                m(x, torch.tensor(t), mems, mem_mask_dict=masks)
            got = read_debug(m, "x", (Be, L, 512))
            w = taps[key].transpose(1, 0, 2)
            err = np.abs(got - w)
            print(f"  {key:16s} rel {rel_l2(got, w):.3e}  |want| max {np.abs(w).max():.3e}  worst element {np.unravel_index(err.argmax(), err.shape)} err {err.max():.3e}  nan {int(np.isnan(got).sum())}", flush=True)
        _lib.check(lib.cfd_debug_stop_stage(m._handle, 0))
