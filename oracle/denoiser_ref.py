"""numpy float32 restatement of the reference ``Denoiser.forward`` -- TEST INFRASTRUCTURE.

Follows the reference op-for-op for the shipped configuration (condition ``text+audio``,
arch ``trans_dec``, ``normalize_before=True``, eval mode, configs/modules/denoiser.yaml):

  Denoiser.forward            convofusion/models/architectures/denoiser.py:173-386
  TransformerDecoder.forward  convofusion/models/operator/cross_attention.py:204-247
  ...Layer2Att.forward_pre    convofusion/models/operator/cross_attention.py:556-664
  TimeBlock.forward           convofusion/models/operator/cross_attention.py:426-439
  get_timestep_embedding      convofusion/models/architectures/tools/embeddings.py:245-285
  TimestepEmbedding.forward   convofusion/models/architectures/tools/embeddings.py:298-305
  PositionEmbeddingSine1D/BH  convofusion/models/operator/position_encoding.py:129-136,154-163
  nn.MultiheadAttention       torch.nn.functional.multi_head_attention_forward (slow path,
                              need_weights=True, average_attn_weights=True)

Tensors use the reference's [T, B, D] layout inside; the public function takes and returns the
reference's batch-first tensors.  Pinned against the imported reference class by
tests/golden/make_golden.py -> tests/golden/denoiser_*.npz.
"""
import math

import numpy as np
from scipy.special import erf

F32 = np.float32
D = 512
MEM_NAMES = ("spkemb", "alsn", "tlsn", "apb", "lsnemb")


def linear(x, w, b=None):
    y = np.matmul(x, w.T)
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def layer_norm(x, g, b, eps=1e-5):
    # torch.nn.LayerNorm: biased variance, eps inside the sqrt
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * g + b).astype(F32, copy=False)


def silu(x):
    return (x / (F32(1.0) + np.exp(-x))).astype(F32, copy=False)


def gelu(x):
    # F.gelu default = exact erf form (cross_attention.py:708-709)
    return (x * F32(0.5) * (F32(1.0) + erf(x * F32(1.0 / math.sqrt(2.0))))).astype(F32, copy=False)


def timestep_embedding(timesteps, dim=D, flip_sin_to_cos=True, downscale_freq_shift=0.0):
    """embeddings.py:245-285.  ``timesteps``: 1-D array."""
    half = dim // 2
    exponent = F32(-math.log(10000)) * np.arange(half, dtype=F32)
    exponent = exponent / F32(half - downscale_freq_shift)
    emb = np.exp(exponent).astype(F32)
    emb = np.asarray(timesteps, dtype=F32)[:, None] * emb[None, :]
    emb = np.concatenate([np.sin(emb), np.cos(emb)], axis=-1)
    if flip_sin_to_cos:
        emb = np.concatenate([emb[:, half:], emb[:, :half]], axis=-1)
    return emb.astype(F32)


def mha(query, key, value, in_w, in_b, out_w, out_b, nhead, key_padding_mask=None):
    """F.multi_head_attention_forward, batch_first=False, need_weights=True.
    query [T,B,E]; key/value [S,B,E]; key_padding_mask bool [B,S] (True = ignore).
    Returns (out [T,B,E], attn averaged over heads [B,T,S])."""
    T, B, E = query.shape
    S = key.shape[0]
    hd = E // nhead
    q = linear(query, in_w[:E], in_b[:E])
    k = linear(key, in_w[E:2 * E], in_b[E:2 * E])
    v = linear(value, in_w[2 * E:], in_b[2 * E:])
    q = q.reshape(T, B * nhead, hd).transpose(1, 0, 2)
    k = k.reshape(S, B * nhead, hd).transpose(1, 0, 2)
    v = v.reshape(S, B * nhead, hd).transpose(1, 0, 2)
    q = q * F32(math.sqrt(1.0 / hd))
    scores = np.matmul(q, k.transpose(0, 2, 1))  # [B*h, T, S]
    if key_padding_mask is not None:
        m = np.zeros((B, 1, 1, S), dtype=F32)
        m[np.asarray(key_padding_mask, dtype=bool)[:, None, None, :]] = -np.inf
        scores = (scores.reshape(B, nhead, T, S) + m).reshape(B * nhead, T, S)
    scores = scores - scores.max(axis=-1, keepdims=True)
    p = np.exp(scores)
    p = (p / p.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    o = np.matmul(p, v)  # [B*h, T, hd]
    o = o.transpose(1, 0, 2).reshape(T, B, E)
    o = linear(o, out_w, out_b)
    return o, p.reshape(B, nhead, T, S).mean(axis=1, dtype=F32)


def time_block(sd, prefix, h, emb):
    """cross_attention.py:426-439; h [T,B,D], emb [1,B,D]."""
    emb_out = linear(silu(emb), sd[prefix + "emb_layers.1.weight"], sd[prefix + "emb_layers.1.bias"])
    scale, shift = emb_out[..., :D], emb_out[..., D:]
    h = layer_norm(h, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"]) * (F32(1.0) + scale) + shift
    return linear(silu(h), sd[prefix + "out_layers.2.weight"], sd[prefix + "out_layers.2.bias"])


def decoder_layer(sd, i, tgt, memory, time_embed, masks, nhead=4, taps=None):
    """TransformerDecoderLayer2Att.forward_pre (cross_attention.py:556-664)."""
    p = f"decoder.layers.{i}."

    def attn(name, q, kv, mask, heads):
        a = p + name
        return mha(q, kv, kv, sd[a + ".in_proj_weight"], sd[a + ".in_proj_bias"],
                   sd[a + ".out_proj.weight"], sd[a + ".out_proj.bias"], heads, mask)

    tgt2 = layer_norm(tgt, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    tgt = tgt + attn("self_attn", tgt2, tgt2, None, nhead)[0]
    if taps is not None:
        taps[f"l{i}.after_self"] = tgt.copy()
    tgt = tgt + time_block(sd, p + "time_block1.", tgt, time_embed)
    if taps is not None:
        taps[f"l{i}.after_tb1"] = tgt.copy()

    tgt2 = layer_norm(tgt, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    outs, atts = [], []
    for name, mem in zip(MEM_NAMES, memory):
        m = layer_norm(mem, sd[p + name + "_norm.weight"], sd[p + name + "_norm.bias"])
        o, a = attn("multihead_attn_" + name, tgt2, m, masks.get(name), 1)
        outs.append(o)
        atts.append(a)
    cat = np.concatenate(outs, axis=-1)
    tgt = tgt + linear(cat, sd[p + "att_fuser.weight"], sd[p + "att_fuser.bias"])
    if taps is not None:
        taps[f"l{i}.after_cross"] = tgt.copy()
    tgt = tgt + time_block(sd, p + "time_block2.", tgt, time_embed)

    tgt2 = layer_norm(tgt, sd[p + "norm3.weight"], sd[p + "norm3.bias"])
    tgt2 = linear(gelu(linear(tgt2, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                  sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    tgt = tgt + tgt2
    if taps is not None:
        taps[f"l{i}.out"] = tgt.copy()
    return tgt, atts


def denoiser_forward(sd, sample, timestep, encoder_hidden_states, mem_mask_dict=None,
                     num_layers=9, nhead=4, taps=None):
    """Denoiser.forward (denoiser.py:173-386).

    sample [Be,L,128]; timestep scalar or [Be]; encoder_hidden_states = (spk, alsn, tlsn, apb,
    lsnemb), each [Be,S_j,512]; mem_mask_dict: name -> bool [Be,S_j] or None.
    Returns (out [Be,L,128], [5 x [Be,num_layers,L,S_j]]).
    """
    masks = dict(mem_mask_dict or {})
    sample = np.asarray(sample, dtype=F32).transpose(1, 0, 2)                       # :183
    L, Be, _ = sample.shape
    x = linear(sample, sd["latent_embd.weight"], sd["latent_embd.bias"])           # :187
    t = np.broadcast_to(np.asarray(timestep, dtype=np.float64).reshape(-1), (Be,)) if np.ndim(timestep) == 0 \
        else np.asarray(timestep).reshape(Be)
    temb = timestep_embedding(t)                                                    # :195-197
    temb = linear(silu(linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                  sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])[None]  # :199
    if taps is not None:
        taps["temb"] = temb.copy()
    mems = [np.asarray(m, dtype=F32).transpose(1, 0, 2) + temb for m in encoder_hidden_states]   # :223-261

    bh = sd["bh_embedding.weight"]
    x = x.copy()
    x[0::2] = x[0::2] + bh[0]                                                       # :316-317
    x[1::2] = x[1::2] + bh[1]
    qpe = sd["query_pos.pe"]
    if L % 2 or L // 2 > qpe.shape[0]:
        raise ValueError("latent length must be even and L/2 <= query PE length")   # position_encoding.py:160-161
    x[0::2] = x[0::2] + qpe[: L // 2]
    x[1::2] = x[1::2] + qpe[: L // 2]
    ce = sd["condition_embedding.weight"]
    mpe = sd["mem_pos.pe"]
    for j in range(5):                                                              # :332-353
        if mems[j].shape[0] > mpe.shape[0]:
            raise ValueError("memory longer than the memory PE buffer")             # position_encoding.py:135
        mems[j] = (mems[j] + ce[j]) + mpe[: mems[j].shape[0]]
    if taps is not None:
        taps["x0"] = x.copy()
        for j, n in enumerate(MEM_NAMES):
            taps["mem." + n] = mems[j].copy()

    per_layer = []
    for i in range(num_layers):                                                     # cross_attention.py:218-234
        x, atts = decoder_layer(sd, i, x, mems, temb, masks, nhead, taps)
        per_layer.append(atts)
    att_mats = [np.stack([per_layer[i][j] for i in range(num_layers)], axis=1) for j in range(5)]
    x = layer_norm(x, sd["decoder.norm.weight"], sd["decoder.norm.bias"])          # :238-239
    out = linear(x, sd["latent_proj.weight"], sd["latent_proj.bias"])              # denoiser.py:382
    return out.transpose(1, 0, 2).copy(), att_mats
