"""numpy float32 restatement of word-excitation guidance (WEG) -- TEST INFRASTRUCTURE (oracle/__init__.py).

The reference obtains d(loss)/d(latents) with torch autograd through ``Denoiser.forward``; there is no autograd
here, so the backward pass is written out by hand, op for op in the reverse order of ``oracle.denoiser_ref``:

  weg.aggregate_attentions / get_max_attention_at_indices / compute_attention_focus_loss / update_latent
                                      convofusion/models/tools/word_excitation_guidance.py:11-81
  GaussianSmoothing (3x3, sigma 0.5)  convofusion/models/operator/gaussian_smoothing.py:21-72
  the WEG branch of the loop          convofusion/models/modeltype/convofusion.py:437-496
  iterative_refinement_step           convofusion/models/modeltype/convofusion.py:298-388
  the differentiated forward          denoiser.py:173-386, cross_attention.py:204-247,426-439,556-664

Pinned against torch autograd through the IMPORTED reference ``Denoiser`` and the imported reference ``weg``
functions by tests/golden/make_golden_weg.py -> tests/golden/weg.npz.
"""
import math

import numpy as np
from scipy.special import erf

from .denoiser_ref import D, F32, MEM_NAMES, gelu, layer_norm, linear, silu, timestep_embedding

TLSN = 2  # position of the listener-text memory in the tuple (denoiser.py:220, convofusion.py:464)


# ----------------------------------------------------------------------------- the loss on the attention maps
def gaussian_kernel(kernel_size=3, sigma=0.5):
    """gaussian_smoothing.py:28-43 (note the reference's exponent: -((x - mean) / (2 sigma))**2)."""
    ax = np.arange(kernel_size, dtype=F32)
    mean = (kernel_size - 1) / 2
    g = (F32(1 / (sigma * math.sqrt(2 * math.pi))) * np.exp(-(((ax - F32(mean)) / F32(2 * sigma)) ** 2))).astype(F32)
    k = (g[:, None] * g[None, :]).astype(F32)
    return (k / k.sum(dtype=F32)).astype(F32)


def focus_loss(att_tlsn, focus_indices, normalize_eot=False, eot_indices=(), smooth=True, want_grad=True):
    """aggregate_attentions + get_max_attention_at_indices + compute_attention_focus_loss, and the gradient of the
    loss with respect to ``att_tlsn`` [B, layers, L, S].  Returns (loss, losses [B], max_att list of lists, grad)."""
    att = np.asarray(att_tlsn, dtype=F32)
    B, NL, L, S = att.shape
    A = att.mean(axis=1, dtype=F32)                                              # weg.py:11-14
    last = -1
    if normalize_eot:                                                            # :23-26
        assert len(eot_indices) > 0, "Need to provide eot indices for normalization"
        assert B == 1, "EOS/BOS normalization only works for test batch size 1 currently"
        last = int(eot_indices[0])
    X = A[:, :, 1:last]                                                          # :28
    W = X.shape[2]
    e = np.exp(X - X.max(axis=-1, keepdims=True))
    sm = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)              # :30
    K = gaussian_kernel()
    if smooth:                                                                   # :33-36 (reflect pad 1, 3x3 correlation)
        pad = np.pad(sm, ((0, 0), (1, 1), (1, 1)), mode="reflect")
        sg = np.zeros_like(sm)
        for a in range(3):
            for b in range(3):
                sg = sg + K[a, b] * pad[:, a:a + L, b:b + W]
        sg = sg.astype(F32)
    else:
        sg = sm
    max_att, losses = [], []
    dsg = np.zeros_like(sg)
    n_b = len(focus_indices)
    for b in range(n_b):                                                         # :40-50, :65-76
        vals = []
        if len(focus_indices[b]) == 0:
            max_att.append(vals)
            losses.append(F32(0))
            continue
        nt = len(focus_indices[b])
        for i in focus_indices[b]:
            col = sg[b, :, i - 1]
            l_star = int(np.argmax(col))
            vals.append(F32(col[l_star]))
            if F32(1) - col[l_star] > 0:                                         # max(0, 1 - token)
                dsg[b, l_star, i - 1] -= F32(1.0 / (nt * n_b))
        max_att.append(vals)
        losses.append(np.mean([max(F32(0), F32(1) - v) for v in vals], dtype=F32))
    loss = F32(np.mean(losses, dtype=F32)) if losses else F32(0)                 # :78-81
    if not want_grad:
        return loss, np.asarray(losses, dtype=F32), max_att, None
    if smooth:
        dpad = np.zeros((B, L + 2, W + 2), dtype=F32)
        for a in range(3):
            for b in range(3):
                dpad[:, a:a + L, b:b + W] += K[a, b] * dsg
        # reflect padding backwards: padded index 0 mirrors 1 (-> original 1), padded L+1 mirrors original L-2
        dsm = dpad[:, 1:-1, 1:-1].copy()
        dsm[:, 1, :] += dpad[:, 0, 1:-1]
        dsm[:, L - 2, :] += dpad[:, L + 1, 1:-1]
        dsm[:, :, 1] += dpad[:, 1:-1, 0]
        dsm[:, :, W - 2] += dpad[:, 1:-1, W + 1]
        dsm[:, 1, 1] += dpad[:, 0, 0]
        dsm[:, 1, W - 2] += dpad[:, 0, W + 1]
        dsm[:, L - 2, 1] += dpad[:, L + 1, 0]
        dsm[:, L - 2, W - 2] += dpad[:, L + 1, W + 1]
    else:
        dsm = dsg
    dX = (sm * (dsm - (dsm * sm).sum(axis=-1, keepdims=True, dtype=F32))).astype(F32)
    dA = np.zeros_like(A)
    dA[:, :, 1:last] = dX
    grad = np.repeat((dA / F32(NL))[:, None], NL, axis=1).astype(F32)
    return loss, np.asarray(losses, dtype=F32), max_att, grad


# ----------------------------------------------------------------------------- backward building blocks
def layer_norm_bwd(x, g, dy, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    rstd = F32(1) / np.sqrt(var + F32(eps))
    xh = xc * rstd
    dh = dy * g
    return (rstd * (dh - dh.mean(axis=-1, keepdims=True, dtype=F32)
                    - xh * (dh * xh).mean(axis=-1, keepdims=True, dtype=F32))).astype(F32)


def silu_grad(x):
    s = F32(1) / (F32(1) + np.exp(-x))
    return (s * (F32(1) + x * (F32(1) - s))).astype(F32)


def gelu_grad(x):
    return (F32(0.5) * (F32(1) + erf(x * F32(1 / math.sqrt(2)))) + x * np.exp(-F32(0.5) * x * x) * F32(1 / math.sqrt(2 * math.pi))).astype(F32)


def mha_fwd(query, key, value, in_w, in_b, out_w, out_b, nhead, key_padding_mask=None):
    """denoiser_ref.mha keeping what the backward needs."""
    T, B, E = query.shape
    S = key.shape[0]
    hd = E // nhead
    q = linear(query, in_w[:E], in_b[:E]).reshape(T, B * nhead, hd).transpose(1, 0, 2) * F32(math.sqrt(1.0 / hd))
    k = linear(key, in_w[E:2 * E], in_b[E:2 * E]).reshape(S, B * nhead, hd).transpose(1, 0, 2)
    v = linear(value, in_w[2 * E:], in_b[2 * E:]).reshape(S, B * nhead, hd).transpose(1, 0, 2)
    sc = np.matmul(q, k.transpose(0, 2, 1))
    if key_padding_mask is not None:
        m = np.zeros((B, 1, 1, S), dtype=F32)
        m[np.asarray(key_padding_mask, dtype=bool)[:, None, None, :]] = -np.inf
        sc = (sc.reshape(B, nhead, T, S) + m).reshape(B * nhead, T, S)
    sc = sc - sc.max(axis=-1, keepdims=True)
    p = np.exp(sc)
    p = (p / p.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    o = np.matmul(p, v).transpose(1, 0, 2).reshape(T, B, E)
    out = linear(o, out_w, out_b)
    return out, p.reshape(B, nhead, T, S).mean(axis=1, dtype=F32), dict(q=q, k=k, v=v, p=p, T=T, B=B, E=E, S=S, H=nhead)


def mha_bwd(sv, dout, in_w, out_w, d_avg_prob=None, self_attention=False):
    """Gradient with respect to the query input (and, for self-attention where query = key = value, the sum of the
    three paths).  ``d_avg_prob`` [B, T, S]: gradient arriving at the head-averaged probabilities."""
    T, B, E, S, H = sv["T"], sv["B"], sv["E"], sv["S"], sv["H"]
    hd = E // H
    do = np.matmul(dout, out_w).astype(F32)                                        # out-proj
    do = do.reshape(T, B * H, hd).transpose(1, 0, 2)
    dp = np.matmul(do, sv["v"].transpose(0, 2, 1)).astype(F32)
    if d_avg_prob is not None:
        dp = dp + np.repeat(d_avg_prob[:, None] / F32(H), H, axis=1).reshape(B * H, T, S)
    p = sv["p"]
    ds = (p * (dp - (dp * p).sum(axis=-1, keepdims=True, dtype=F32))).astype(F32)
    dq = np.matmul(ds, sv["k"]).astype(F32) * F32(math.sqrt(1.0 / hd))
    dq = dq.transpose(1, 0, 2).reshape(T, B, E)
    dx = np.matmul(dq, in_w[:E]).astype(F32)
    if self_attention:
        dk = np.matmul(ds.transpose(0, 2, 1), sv["q"]).astype(F32).transpose(1, 0, 2).reshape(S, B, E)
        dv = np.matmul(p.transpose(0, 2, 1), do).astype(F32).transpose(1, 0, 2).reshape(S, B, E)
        dx = dx + np.matmul(dk, in_w[E:2 * E]) + np.matmul(dv, in_w[2 * E:])
    return dx.astype(F32)


def time_block_fwd(sd, prefix, x, emb):
    e = linear(silu(emb), sd[prefix + "emb_layers.1.weight"], sd[prefix + "emb_layers.1.bias"])
    scale = e[..., :D]
    shift = e[..., D:]
    h = layer_norm(x, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"]) * (F32(1) + scale) + shift
    return linear(silu(h), sd[prefix + "out_layers.2.weight"], sd[prefix + "out_layers.2.bias"]), dict(x=x, h=h, scale=scale)


def time_block_bwd(sd, prefix, sv, dout):
    dh = np.matmul(dout, sd[prefix + "out_layers.2.weight"]).astype(F32) * silu_grad(sv["h"])
    return layer_norm_bwd(sv["x"], sd[prefix + "norm.weight"], dh * (F32(1) + sv["scale"]))


# ----------------------------------------------------------------------------- forward with saved activations, backward
def forward_saved(sd, sample, timestep, encoder_hidden_states, mem_mask_dict=None, num_layers=9, nhead=4):
    """denoiser_ref.denoiser_forward keeping the activations; returns (att_mats list, saved)."""
    masks = dict(mem_mask_dict or {})
    x = linear(np.asarray(sample, dtype=F32).transpose(1, 0, 2), sd["latent_embd.weight"], sd["latent_embd.bias"]).copy()
    L, Be, _ = x.shape
    temb = timestep_embedding(np.full((Be,), float(timestep)))
    temb = linear(silu(linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                  sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])[None]
    mems = [np.asarray(m, dtype=F32).transpose(1, 0, 2) + temb for m in encoder_hidden_states]
    bh, qpe = sd["bh_embedding.weight"], sd["query_pos.pe"]
    x[0::2] += bh[0] + qpe[: L // 2]
    x[1::2] += bh[1] + qpe[: L // 2]
    for j in range(5):
        mems[j] = (mems[j] + sd["condition_embedding.weight"][j]) + sd["mem_pos.pe"][: mems[j].shape[0]]
    layers, per_layer = [], []
    for i in range(num_layers):
        p = f"decoder.layers.{i}."
        sv = {"x0": x}
        t2 = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        a = p + "self_attn"
        o, _, sv["self"] = mha_fwd(t2, t2, t2, sd[a + ".in_proj_weight"], sd[a + ".in_proj_bias"], sd[a + ".out_proj.weight"],
                                   sd[a + ".out_proj.bias"], nhead)
        x = x + o
        o, sv["tb1"] = time_block_fwd(sd, p + "time_block1.", x, temb)
        x = x + o
        sv["x2"] = x
        t2 = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        outs, atts, sv["cross"] = [], [], []
        for name, mem in zip(MEM_NAMES, mems):
            m = layer_norm(mem, sd[p + name + "_norm.weight"], sd[p + name + "_norm.bias"])
            a = p + "multihead_attn_" + name
            o, att, s = mha_fwd(t2, m, m, sd[a + ".in_proj_weight"], sd[a + ".in_proj_bias"], sd[a + ".out_proj.weight"],
                                sd[a + ".out_proj.bias"], 1, masks.get(name))
            outs.append(o)
            atts.append(att)
            sv["cross"].append(s)
        x = x + linear(np.concatenate(outs, axis=-1), sd[p + "att_fuser.weight"], sd[p + "att_fuser.bias"])
        o, sv["tb2"] = time_block_fwd(sd, p + "time_block2.", x, temb)
        x = x + o
        sv["x4"] = x
        t2 = layer_norm(x, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
        sv["ffn_pre"] = linear(t2, sd[p + "linear1.weight"], sd[p + "linear1.bias"])
        x = x + linear(gelu(sv["ffn_pre"]), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        layers.append(sv)
        per_layer.append(atts)
    att_mats = [np.stack([per_layer[i][j] for i in range(num_layers)], axis=1) for j in range(5)]
    return att_mats, layers


def backward_to_sample(sd, layers, d_att_tlsn, num_layers=9):
    """Gradient of a scalar that depends on the tlsn attention probabilities only (d_att_tlsn [B, layers, L, S])
    with respect to ``sample`` [B, L, 128]."""
    g = None                                                                         # gradient at the layer's output
    for i in reversed(range(num_layers)):
        p = f"decoder.layers.{i}."
        sv = layers[i]
        if g is not None:
            d1 = np.matmul(g, sd[p + "linear2.weight"]).astype(F32) * gelu_grad(sv["ffn_pre"])
            g = g + layer_norm_bwd(sv["x4"], sd[p + "norm3.weight"], np.matmul(d1, sd[p + "linear1.weight"]).astype(F32))
            g = g + time_block_bwd(sd, p + "time_block2.", sv["tb2"], g)
        dt2 = None
        dcat = np.matmul(g, sd[p + "att_fuser.weight"]).astype(F32) if g is not None else None
        for j, name in enumerate(MEM_NAMES):
            if dcat is None and j != TLSN:
                continue
            a = p + "multihead_attn_" + name
            dout = dcat[..., j * D:(j + 1) * D] if dcat is not None else np.zeros((sv["cross"][j]["T"], sv["cross"][j]["B"], D), F32)
            d = mha_bwd(sv["cross"][j], dout, sd[a + ".in_proj_weight"], sd[a + ".out_proj.weight"],
                        d_att_tlsn[:, i] if j == TLSN else None)
            dt2 = d if dt2 is None else dt2 + d
        gq = layer_norm_bwd(sv["x2"], sd[p + "norm2.weight"], dt2)
        g = gq if g is None else g + gq
        g = g + time_block_bwd(sd, p + "time_block1.", sv["tb1"], g)
        a = p + "self_attn"
        dt2 = mha_bwd(sv["self"], g, sd[a + ".in_proj_weight"], sd[a + ".out_proj.weight"], None, self_attention=True)
        g = g + layer_norm_bwd(sv["x0"], sd[p + "norm1.weight"], dt2)
    return np.matmul(g, sd["latent_embd.weight"]).astype(F32).transpose(1, 0, 2).copy()


def loss_and_grad(sd, latents, t, encoder_hidden_states, cond_masks, focus_indices, normalize_eot=True, eot_indices=None):
    """One evaluation of the WEG objective on the text-only chunk (convofusion.py:447-471) and d(loss)/d(latents)."""
    if eot_indices is None:
        eot_indices = np.argmax(np.asarray(cond_masks["tlsn"]).astype(np.int64), axis=1) - 1     # :460
    att, layers = forward_saved(sd, latents, t, encoder_hidden_states, cond_masks)
    loss, losses, max_att, datt = focus_loss(att[TLSN], focus_indices, normalize_eot, eot_indices)
    grad = backward_to_sample(sd, layers, datt)
    return loss, losses, max_att, grad


def scale_range_schedule(weg_parameters, num_steps, carry=None):
    """Step-size factor table of one loop iteration.  ``carry=None``: the rollout's fresh ``np.linspace(lo, hi, N)``
    (unbounded_synthesis.py:82-89).  ``carry=[lo, hi]`` threaded through the loop: ``Convofusion._diffusion_reverse``'s
    re-assignment ``scale_range = np.linspace(scale_range[0], scale_range[1], N)`` (convofusion.py:395,442-444), whose
    interval collapses after iteration 0.  Pinned by tests/golden/weg_scale_schedule.npz (the reference's own statement
    executed by make_golden_weg_schedule.py)."""
    if carry is None:
        return np.linspace(weg_parameters["scale_range"][0], weg_parameters["scale_range"][1], num_steps)
    sr = np.linspace(carry[0], carry[1], num_steps)
    carry[0], carry[1] = sr[0], sr[1] if num_steps > 1 else sr[0]
    return sr


def weg_update(sd, latents, i, t, text_only_states, text_only_masks, focus_indices, weg_parameters, num_steps, scale_carry=None):
    """The WEG branch of one loop iteration (convofusion.py:437-496): returns the altered latents.
    ``text_only_states`` / masks are chunk 1 of the 7-way guidance batch (:447-448)."""
    scale_range = scale_range_schedule(weg_parameters, num_steps, scale_carry)      # :442-444
    eot = np.argmax(np.asarray(text_only_masks["tlsn"]).astype(np.int64), axis=1) - 1
    step_size = weg_parameters["scale_factor"] * np.sqrt(scale_range[i])
    loss, _, _, grad = loss_and_grad(sd, latents, t, text_only_states, text_only_masks, focus_indices, True, eot)
    thresholds = weg_parameters["thresholds"]
    if i in thresholds and loss > 1.0 - thresholds[i]:                              # :474-487, :298-388
        target, it = max(0.0, 1.0 - thresholds[i]), 0
        while loss > target:
            it += 1
            loss, _, _, grad = loss_and_grad(sd, latents, t, text_only_states, text_only_masks, focus_indices, True, eot)
            if loss != 0:
                latents = (latents - F32(step_size) * grad).astype(F32)
            if it >= weg_parameters["max_refinement_steps"]:
                break
        loss, _, _, grad = loss_and_grad(sd, latents, t, text_only_states, text_only_masks, focus_indices, True, eot)
    if i < weg_parameters["max_iter_to_alter"] and loss != 0:                       # :490-495
        latents = (latents - F32(step_size) * grad).astype(F32)
    return latents, float(loss)
