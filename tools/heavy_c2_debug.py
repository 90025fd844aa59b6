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

if os.environ.get("TAPS"):
    # residual stream of the listener-id chunk's row (chunk 5) after every sub-block against the numpy oracle's taps (needs ROWS_ONLY=1)
    from convofusion_amd import _lib
    from oracle import denoiser_ref, weights
    from tests.gpu_helpers import read_debug
    assert rows_only
    sd = weights.extend_pe(heavy_state_dict(float(os.environ.get("GAIN", "20"))), 1536)
    taps = {}
    denoiser_ref.denoiser_forward(sd, x_np, t, mems_np, masks_np, taps=taps)
    lib = _lib.load()
    stages = [(1, "x0")]
    for l in range(9):
        stages += [(2 + 4 * l, f"l{l}.after_self"), (3 + 4 * l, f"l{l}.after_tb1"), (4 + 4 * l, f"l{l}.after_cross"), (5 + 4 * l, f"l{l}.out")]
    for stage, key in stages:
        _lib.check(lib.cfd_debug_stop_stage(m._handle, stage))
        with torch.no_grad():
            m(to_dev(x_np), torch.tensor(t), [to_dev(v) for v in mems_np], mem_mask_dict={k: to_dev(v) for k, v in masks_np.items()})
        got = read_debug(m, "x", (7, L, 512))
        w = taps[key].transpose(1, 0, 2)
        print(f"  {key:16s} per chunk rel:", [f"{rel_l2(got[c], w[c]):.1e}" for c in range(7)], flush=True)
    _lib.check(lib.cfd_debug_stop_stage(m._handle, 0))

if os.environ.get("DIAG"):
    # structure of the error of chunk 5 behind layer DIAG's cross-attention: the same vector for every token (a per-row constant: the one-key
    # memory's contribution, a bias) or different per token (scores / softmax / P.V)?
    l = int(os.environ["DIAG"])
    for stage, key in ((3 + 4 * l, f"l{l}.after_tb1"), (4 + 4 * l, f"l{l}.after_cross")):
        _lib.check(lib.cfd_debug_stop_stage(m._handle, stage))
        with torch.no_grad():
            m(to_dev(x_np), torch.tensor(t), [to_dev(v) for v in mems_np], mem_mask_dict={k: to_dev(v) for k, v in masks_np.items()})
        got = read_debug(m, "x", (7, L, 512))[5].astype(np.float64)
        w = taps[key].transpose(1, 0, 2)[5].astype(np.float64)
        if key.endswith("tb1"):
            g0, w0 = got, w
        else:
            dg, dw = got - g0, w - w0                        # what the cross-attention block added (HIP / oracle)
            e = dg - dw
            tok = np.linalg.norm(e, axis=1)
            mean_e = e.mean(0)
            print(f"layer {l} cross-attention update of chunk 5: |update| {np.linalg.norm(dw):.3e}, |error| {np.linalg.norm(e):.3e}, "
                  f"error explained by ONE vector common to all tokens: {1 - np.linalg.norm(e - mean_e) ** 2 / np.linalg.norm(e) ** 2:.3f}; "
                  f"per-token |error| min/max {tok.min():.2e}/{tok.max():.2e}; largest features of the common vector {np.argsort(-np.abs(mean_e))[:5]} "
                  f"values {mean_e[np.argsort(-np.abs(mean_e))[:5]]}", flush=True)
    _lib.check(lib.cfd_debug_stop_stage(m._handle, 0))

if os.environ.get("EXACT"):
    # Round-6 question: would an EXACT score product (fp32 or better operands) in the cross-attention repair the one
    # ill-conditioned chunk?  Layer EXACT's cross-attention block of chunk 5 recomputed in FLOAT64, reference formulation, (a) from the HIP
    # path's own input to the block and (b) from the oracle's input: |HIP update - (a)| is what any amount of precision INSIDE the block
    # could remove, |(a) - (b)| is what the block does to the input difference it is handed (upstream rounding, amplified by the
    # ill-conditioned softmax) and no block-local precision can touch.  Needs ROWS_ONLY=1 TAPS=1.
    l = int(os.environ["EXACT"])
    p = f"decoder.layers.{l}."
    sd64 = {k: v.astype(np.float64) for k, v in sd.items() if k.startswith(p)}

    def ln64(x, g_, b_):
        mu = x.mean(-1, keepdims=True)
        xc = x - mu
        return xc / np.sqrt((xc * xc).mean(-1, keepdims=True) + 1e-5) * g_ + b_

    def cross64(x):          # x [L, 512] of chunk 5 -> the block's update (cross_attention.py:578-652)
        q_in = ln64(x, sd64[p + "norm2.weight"], sd64[p + "norm2.bias"])
        outs = []
        for name in ("spkemb", "alsn", "tlsn", "apb", "lsnemb"):
            mem = taps["mem." + name][:, 5].astype(np.float64)
            mn = ln64(mem, sd64[p + name + "_norm.weight"], sd64[p + name + "_norm.bias"])
            a = p + "multihead_attn_" + name
            w_, b_ = sd64[a + ".in_proj_weight"], sd64[a + ".in_proj_bias"]
            q = q_in @ w_[:512].T + b_[:512]
            k = mn @ w_[512:1024].T + b_[512:1024]
            v = mn @ w_[1024:].T + b_[1024:]
            s = (q / np.sqrt(512.0)) @ k.T
            mk = masks_np.get(name)
            if mk is not None:
                s = np.where(np.asarray(mk[5], dtype=bool)[None, :], -np.inf, s)
            s = s - s.max(-1, keepdims=True)
            pr = np.exp(s)
            pr = pr / pr.sum(-1, keepdims=True)
            outs.append((pr @ v) @ sd64[a + ".out_proj.weight"].T + sd64[a + ".out_proj.bias"])
        return np.concatenate(outs, -1) @ sd64[p + "att_fuser.weight"].T + sd64[p + "att_fuser.bias"]

    x_in, x_out = {}, {}
    for stage, store in ((3 + 4 * l, x_in), (4 + 4 * l, x_out)):
        _lib.check(lib.cfd_debug_stop_stage(m._handle, stage))
        with torch.no_grad():
            m(to_dev(x_np), torch.tensor(t), [to_dev(v) for v in mems_np], mem_mask_dict={k: to_dev(v) for k, v in masks_np.items()})
        store["hip"] = read_debug(m, "x", (7, L, 512))[5].astype(np.float64)
    _lib.check(lib.cfd_debug_stop_stage(m._handle, 0))
    x_in["ref"] = taps[f"l{l}.after_tb1"].transpose(1, 0, 2)[5].astype(np.float64)
    x_out["ref"] = taps[f"l{l}.after_cross"].transpose(1, 0, 2)[5].astype(np.float64)
    upd_hip, upd_ref32 = x_out["hip"] - x_in["hip"], x_out["ref"] - x_in["ref"]
    ex_hip, ex_ref = cross64(x_in["hip"]), cross64(x_in["ref"])
    n = np.linalg.norm
    print(f"layer {l} cross-attention block, chunk 5: |update| {n(ex_ref):.3e}; input difference HIP vs float32 oracle {n(x_in['hip'] - x_in['ref']) / n(x_in['ref']):.2e} rel\n"
          f"  HIP block vs float64 block on the HIP INPUT (what block-local precision could remove):      {n(upd_hip - ex_hip):.3e}  = {n(upd_hip - ex_hip) / n(ex_ref):.2e} of the update\n"
          f"  float32 oracle block vs float64 block on the ORACLE input (float32's own error):           {n(upd_ref32 - ex_ref):.3e}  = {n(upd_ref32 - ex_ref) / n(ex_ref):.2e}\n"
          f"  float64 block on the HIP input vs on the oracle input (input difference, amplified):       {n(ex_hip - ex_ref):.3e}  = {n(ex_hip - ex_ref) / n(ex_ref):.2e}\n"
          f"  HIP block output vs float32 oracle block output (what the stage-wise taps report):         {n(upd_hip - upd_ref32):.3e}  = {n(upd_hip - upd_ref32) / n(ex_ref):.2e}", flush=True)
