"""Developer tool (GPU box): the error table of the fused cross-attention's operand policies (cfd_sample_args.operand_policy:
0 = fp16 split pairs, 1 = folded values single fp16, 2 = folded keys single fp16, 3 = both) on every DDPM golden, through the
parity tests' own code (their printed errors are collected; a failed threshold is recorded, not raised).

  python tools/xa_operands_table.py            headline-shape goldens (B = 32, L = 196, 1500 audio keys): ddpm5 / ddpm1000 x {b32, skip, b1_shard},
                                               and a 50-step DDPM loop on the heavy-tailed weights at that shape, every mode against mode 0
  python tools/xa_operands_table.py --small    the small goldens (ddpm20_b2, inpaint25, ddpm1000, heavy ddpm1000) forced onto the tile kernels
                                               (CFD_ROWTILE=0, CFD_FUSED_XATTN_MIN_WGS=0: by default they take the row-tile path, which has no policy)
"""
import contextlib
import io
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SMALL = "--small" in sys.argv
MODES = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0, 3, 7, 11, 15]
os.environ.setdefault("CFD_FUSED_XATTN_MIN_WGS", "0")
if SMALL:
    os.environ["CFD_ROWTILE"] = "0"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from convofusion_amd import sampler  # noqa: E402
from tests import test_gpu_sampler as T  # noqa: E402


def run(fn, *a):
    buf = io.StringIO()
    status = "ok"
    with contextlib.redirect_stdout(buf):
        try:
            fn(*a)
        except AssertionError as e:
            status = "FAIL " + str(e)[:200]
    errs = [float(x) for x in re.findall(r"'(\d\.\d+e[-+]\d+)'", buf.getvalue())]
    return (max(errs) if errs else float("nan")), (errs[-1] if errs else float("nan")), status


def heavy_c2_ddpm(mode, n=50, ref={}):
    """50 DDPM steps (of a 50-step schedule: stride 20) of the captured B = 32 loop at the headline shape on the heavy-tailed weights (factor 8)
    with outlier-token memories; row u against mode 0's row (mode 0 itself: 0)."""
    from convofusion_amd.denoiser import Denoiser
    from oracle import inputs
    from tests.gpu_helpers import ABL, DENOISER_KW, to_dev
    from tests.helpers import heavy_state_dict, load_golden, rel_l2
    g = load_golden("heavy_c2")
    meta = [int(v) for v in g["meta"]]
    B, L, S, pad, seed, u = meta[0], meta[1], tuple(meta[2:7]), tuple(meta[7:12]), meta[13], meta[14]
    cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=pad, uncond_pad_tail=pad)
    mems = [to_dev(inputs.add_outlier_tokens(uq, seed + j)[rm]) for j, (uq, rm) in enumerate(zip(cb["unique"], cb["row_map"]))]
    masks = {k: to_dev(v) for k, v in cb["masks"].items()}
    def make():
        m = Denoiser(ablation=ABL, **DENOISER_KW)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in heavy_state_dict(8.0).items()}, strict=True)
        return m.cuda().eval()
    if "m" not in ref:
        ref["m"] = make()
    den = ref["m"]
    if mode == "reassoc":     # context: split pairs everywhere, but layer 0's cross-attention summed in another order (CFD_L0_DEDUP=0: an fp32 re-association only)
        os.environ["CFD_L0_DEDUP"] = "0"
        den = make()
        den.engine("cuda")    # (the knobs are read when the handle is created)
        del os.environ["CFD_L0_DEDUP"]
        mode = 0
        label = "pairs, layer-0 sum re-associated"
    else:
        label = f"mode {mode}"
    r = sampler.SamplingRun(den, T._sched("ddpm"), mems, masks, B, L, n, guidance_scale=7.5, seed=seed, operands=mode)
    r.steps(n)
    lat = r.read(close=True).cpu().numpy()
    if mode == 0 and label == "mode 0":
        ref["lat"] = lat
    e_all = rel_l2(lat, ref["lat"])
    e_row = max(rel_l2(lat[b], ref["lat"][b]) for b in range(B))
    print(f"heavy c2 ddpm{n}: {label} vs mode 0: all rows {e_all:.2e}, worst row {e_row:.2e}", file=sys.stderr)
    return e_row, e_all, "ok" if np.isfinite(lat).all() else "FAIL nan"


rows = []
for mode in MODES:
    sampler.OPERAND_POLICY[0] = sampler.OPERAND_POLICY[1] = mode     # (DDIM too: the table says what DDIM would lose)
    if SMALL:
        for name in ("ddpm20_b2", "inpaint25", "ddpm1000"):
            rows.append((mode, "traj_" + name + " (tile kernels)", *run(T.test_sampler_matches_reference_trajectory, name)))
        rows.append((mode, "heavy ddpm1000 (tile kernels)", *run(T.test_heavy_tailed_weights_ddpm1000_trajectory)))
    else:
        for kind in ("ddpm5", "ddpm1000"):
            for variant in ("b32", "b32_skip_zero_weight_chunk", "b1_shard"):
                rows.append((mode, f"traj_c2_{kind} {variant}", *run(T.test_headline_shape_loop_row_matches_reference, kind, variant)))
        rows.append((mode, "traj_c2_ddim50 b32 (DDIM: not a default candidate)", *run(T.test_headline_shape_loop_row_matches_reference, "ddim50", "b32")))
        rows.append((mode, "heavy_c2 ddpm50, worst row vs mode 0 (last column: all rows)", *heavy_c2_ddpm(mode)))
        if mode == 0:
            rows.append(("0*", "heavy_c2 ddpm50, pairs with layer 0's sum re-associated (fp32 rounding order only), worst row vs mode 0 (last column: all rows)",
                         *heavy_c2_ddpm("reassoc")))
    for r in rows[-10:]:
        if r[0] == mode or (mode == 0 and r[0] == "0*"):
            print(f"mode {r[0]} | {r[1]} | max over snapshots {r[2]:.2e} | final {r[3]:.2e} | {r[4]}", flush=True)
print("\n| operand policy | golden | max over snapshots | final | thresholds |\n|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]:.2e} | {r[3]:.2e} | {r[4]} |")
