"""Developer tool (GPU box): the heavy-tailed forward at the headline shape (tests/golden/heavy_c2.npz) under the code-path knobs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd.denoiser import Denoiser  # noqa: E402
from oracle import inputs  # noqa: E402
from tests.gpu_helpers import ABL, DENOISER_KW, to_dev  # noqa: E402
from tests.helpers import heavy_state_dict, load_golden, rel_l2  # noqa: E402

g = load_golden("heavy_c2")
meta = [int(v) for v in g["meta"]]
B, L, S, pad, t, seed, u = meta[0], meta[1], tuple(meta[2:7]), tuple(meta[7:12]), meta[12], meta[13], meta[14]
cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=pad, uncond_pad_tail=pad)
idx = np.array([c * B + u for c in range(7)])
rows_only = os.environ.get("ROWS_ONLY", "0") == "1"
uq = [inputs.add_outlier_tokens(q, seed + j) for j, q in enumerate(cb["unique"])]
mems_np = [q[rm] for q, rm in zip(uq, cb["row_map"])]
if rows_only:
    mems_np = [m[idx] for m in mems_np]
    masks_np = {k: (v[idx] if v is not None else None) for k, v in cb["masks"].items()}
    x_np = np.concatenate([cb["init"][u:u + 1]] * 7)
else:
    masks_np = cb["masks"]
    x_np = np.concatenate([cb["init"]] * 7)
m = Denoiser(ablation=ABL, **DENOISER_KW)
m.load_state_dict({k: torch.from_numpy(v) for k, v in heavy_state_dict(float(os.environ.get("GAIN", "20"))).items()}, strict=True)
m = m.cuda().eval()
m.return_attention = os.environ.get("ATT", "0") == "1"
with torch.no_grad():
    out, _ = m(to_dev(x_np), torch.tensor(t), [to_dev(v) for v in mems_np], mem_mask_dict={k: to_dev(v) for k, v in masks_np.items()})
out = out.cpu().numpy()
got = out if rows_only else out[idx]
per_row = [rel_l2(got[c], g["out5"][c]) for c in range(7)]
print({k: os.environ.get(k) for k in ("CFD_HOIST_MEMSIDE", "CFD_FUSED_XATTN", "ATT", "ROWS_ONLY", "CFD_L0_DEDUP", "CFD_ONE_KEY")}, f"rel {rel_l2(got, g['out5']):.2e}", "per chunk", [f"{e:.1e}" for e in per_row])
