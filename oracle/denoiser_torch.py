"""PyTorch CPU-eager float32 restatement of the reference ``Denoiser.forward`` -- TEST INFRASTRUCTURE
(see oracle/__init__.py).  This is the ``cpu_baseline`` leg of bench.py: the reference runs PyTorch eager
on the host (BASELINE.json configs[0]), so the baseline timed beside the HIP path is
the same torch op sequence -- ``F.linear``, ``F.layer_norm``, ``F.multi_head_attention_forward`` (what
``nn.MultiheadAttention.forward`` calls on its slow path with ``need_weights=True``), ``F.gelu``, ``F.silu`` --
driven from a plain state dict instead of the reference's module tree (which cannot travel to the GPU box).

  Denoiser.forward            convofusion/models/architectures/denoiser.py:173-386
  TransformerDecoder.forward  convofusion/models/operator/cross_attention.py:204-247
  ...Layer2Att.forward_pre    convofusion/models/operator/cross_attention.py:556-664
  TimeBlock.forward           convofusion/models/operator/cross_attention.py:426-439
  get_timestep_embedding      convofusion/models/architectures/tools/embeddings.py:245-285

Pinned against the outputs of the imported reference class (tests/golden/denoiser_*.npz) by
tests/test_oracle_denoiser.py::test_torch_restatement_matches_reference_golden.
"""
import math

import torch
import torch.nn.functional as F

D = 512
MEM_NAMES = ("spkemb", "alsn", "tlsn", "apb", "lsnemb")


def to_torch(sd_np):
    return {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}


def timestep_embedding(timesteps, dim=D):
    """embeddings.py:245-285 with flip_sin_to_cos=True, downscale_freq_shift=0."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    return torch.cat([emb[:, half:], emb[:, :half]], dim=-1)


def _mha(sd, prefix, q, kv, mask, heads):
    out, att = F.multi_head_attention_forward(
        q, kv, kv, D, heads, sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"], None, None, False, 0.0,
        sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"], training=False, key_padding_mask=mask,
        need_weights=True, average_attn_weights=True)
    return out, att


def _time_block(sd, prefix, h, emb):
    emb_out = F.linear(F.silu(emb), sd[prefix + "emb_layers.1.weight"], sd[prefix + "emb_layers.1.bias"])
    scale, shift = torch.chunk(emb_out, 2, dim=-1)
    h = F.layer_norm(h, (D,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"]) * (1 + scale) + shift
    return F.linear(F.silu(h), sd[prefix + "out_layers.2.weight"], sd[prefix + "out_layers.2.bias"])


def _layer(sd, i, tgt, memory, temb, masks, nhead):
    p = f"decoder.layers.{i}."
    tgt2 = F.layer_norm(tgt, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    tgt = tgt + _mha(sd, p + "self_attn", tgt2, tgt2, None, nhead)[0]
    tgt = tgt + _time_block(sd, p + "time_block1.", tgt, temb)
    tgt2 = F.layer_norm(tgt, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    outs, atts = [], []
    for name, mem in zip(MEM_NAMES, memory):
        m = F.layer_norm(mem, (D,), sd[p + name + "_norm.weight"], sd[p + name + "_norm.bias"])
        o, a = _mha(sd, p + "multihead_attn_" + name, tgt2, m, masks.get(name), 1)
        outs.append(o)
        atts.append(a)
    tgt = tgt + F.linear(torch.cat(outs, dim=-1), sd[p + "att_fuser.weight"], sd[p + "att_fuser.bias"])
    tgt = tgt + _time_block(sd, p + "time_block2.", tgt, temb)
    tgt2 = F.layer_norm(tgt, (D,), sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    tgt2 = F.linear(F.gelu(F.linear(tgt2, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                    sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return tgt + tgt2, atts


@torch.no_grad()
def denoiser_forward(sd, sample, timestep, encoder_hidden_states, mem_mask_dict=None, num_layers=9, nhead=4):
    """sd: dict of torch CPU tensors (``to_torch``); sample [Be,L,128]; timestep int; memories 5 x [Be,S_j,512];
    masks name -> bool [Be,S_j] | None.  Returns (out [Be,L,128], 5 x [Be,num_layers,L,S_j])."""
    masks = {k: v for k, v in (mem_mask_dict or {}).items() if v is not None}
    sample = sample.permute(1, 0, 2)
    L, Be, _ = sample.shape
    x = F.linear(sample, sd["latent_embd.weight"], sd["latent_embd.bias"])
    t = torch.as_tensor(timestep).reshape(-1).expand(Be) if torch.as_tensor(timestep).dim() == 0 else torch.as_tensor(timestep).reshape(Be)
    temb = timestep_embedding(t)
    temb = F.linear(F.silu(F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                    sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"]).unsqueeze(0)
    mems = [m.permute(1, 0, 2) + temb for m in encoder_hidden_states]
    bh, qpe = sd["bh_embedding.weight"], sd["query_pos.pe"]
    x = x.clone()
    x[0::2] = x[0::2] + bh[0] + qpe[: L // 2]
    x[1::2] = x[1::2] + bh[1] + qpe[: L // 2]
    ce, mpe = sd["condition_embedding.weight"], sd["mem_pos.pe"]
    mems = [(m + ce[j]) + mpe[: m.shape[0]] for j, m in enumerate(mems)]
    per_layer = []
    for i in range(num_layers):
        x, atts = _layer(sd, i, x, mems, temb, masks, nhead)
        per_layer.append(atts)
    att_mats = [torch.stack([per_layer[i][j] for i in range(num_layers)], dim=1) for j in range(5)]
    x = F.layer_norm(x, (D,), sd["decoder.norm.weight"], sd["decoder.norm.bias"])
    out = F.linear(x, sd["latent_proj.weight"], sd["latent_proj.bias"])
    return out.permute(1, 0, 2).contiguous(), att_mats
