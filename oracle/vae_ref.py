"""numpy restatement of ``ConvoFusionVae.decode`` -- TEST INFRASTRUCTURE (oracle/__init__.py).

Reference: convofusion/models/architectures/vae.py:268-372 (arch 'encoder_decoder', PE_TYPE 'convofusion',
configs/modules/motion_vae.yaml: 5 layers, 2 heads, ff 1024, pre-norm, gelu, sine PE, latent_dim [1, 128]);
SkipTransformerDecoder cross_attention.py:66-125; TransformerDecoderLayer.forward_pre :361-382;
nn.MultiheadAttention (torch): packed in-projection rows [q | k | v], q scaled by 1/sqrt(head_dim) before the
product, key-padding mask as -inf, softmax, value sum, out-projection; PositionEmbeddingSine1D
position_encoding.py:113-136; lengths_to_mask utils/temos_utils.py:11-18.
Pinned against the imported reference class: tests/golden/vae_decode.npz (tests/golden/make_golden_vae.py).
"""
import math

import numpy as np

from .conditioning_ref import gelu, linear

F32 = np.float32


def layer_norm(x, g, b, eps=1e-5):
    x64 = x.astype(np.float64)
    m = x64.mean(-1, keepdims=True)
    v = ((x64 - m) ** 2).mean(-1, keepdims=True)
    return ((x64 - m) / np.sqrt(v + eps) * g.astype(np.float64) + b.astype(np.float64)).astype(F32)


def mha(sd, pre, query, key, value, nhead, key_padding_mask=None):
    """nn.MultiheadAttention.forward(query, key, value, key_padding_mask)[0]; tensors are [L, N, E]."""
    E = query.shape[-1]
    W, B = sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"]
    q = linear(query, W[:E], B[:E])
    k = linear(key, W[E:2 * E], B[E:2 * E])
    v = linear(value, W[2 * E:], B[2 * E:])
    Lq, N, _ = q.shape
    Lk = k.shape[0]
    hd = E // nhead
    qh = (q.astype(np.float64) * math.sqrt(1.0 / hd)).reshape(Lq, N, nhead, hd).transpose(1, 2, 0, 3)   # [N, H, Lq, hd]
    kh = k.astype(np.float64).reshape(Lk, N, nhead, hd).transpose(1, 2, 0, 3)
    vh = v.astype(np.float64).reshape(Lk, N, nhead, hd).transpose(1, 2, 0, 3)
    s = qh @ kh.transpose(0, 1, 3, 2)                                                                      # [N, H, Lq, Lk]
    if key_padding_mask is not None:
        s = np.where(key_padding_mask[:, None, None, :], -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = (p @ vh).transpose(2, 0, 1, 3).reshape(Lq, N, E).astype(F32)
    return linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def decoder_layer(sd, pre, tgt, memory, nhead, tgt_key_padding_mask):
    """TransformerDecoderLayer.forward_pre (cross_attention.py:361-382), eval mode (dropout = identity)."""
    t2 = layer_norm(tgt, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    tgt = tgt + mha(sd, pre + "self_attn.", t2, t2, t2, nhead, tgt_key_padding_mask)
    t2 = layer_norm(tgt, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    tgt = tgt + mha(sd, pre + "multihead_attn.", t2, memory, memory, nhead, None)
    t2 = layer_norm(tgt, sd[pre + "norm3.weight"], sd[pre + "norm3.bias"])
    h = gelu(linear(t2, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"]))
    return (tgt + linear(h, sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])).astype(F32)


def skip_decoder(sd, pre, tgt, memory, num_layers, nhead, tgt_key_padding_mask):
    """SkipTransformerDecoder.forward (cross_attention.py:89-125)."""
    nb = (num_layers - 1) // 2
    x, xs = tgt, []
    for i in range(nb):
        x = decoder_layer(sd, f"{pre}input_blocks.{i}.", x, memory, nhead, tgt_key_padding_mask)
        xs.append(x)
    x = decoder_layer(sd, pre + "middle_block.", x, memory, nhead, tgt_key_padding_mask)
    for i in range(nb):
        x = np.concatenate([x, xs.pop()], axis=-1)
        x = linear(x, sd[f"{pre}linear_blocks.{i}.weight"], sd[f"{pre}linear_blocks.{i}.bias"])
        x = decoder_layer(sd, f"{pre}output_blocks.{i}.", x, memory, nhead, tgt_key_padding_mask)
    return layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])


def decode(sd, z, lengths, num_layers=5, nhead=2):
    """ConvoFusionVae.decode(z [2, bs, n_chunks, D], lengths) -> feats [bs, nframes, 189] (vae.py:268-372)."""
    _, bs, n_chunks, D = z.shape
    lengths = np.asarray(lengths)
    nframes = int(lengths.max())
    mask = np.arange(nframes)[None, :] < lengths[:, None]                                  # temos_utils.py:11-18
    queries = np.zeros((nframes, bs, D), F32) + sd["query_pos_decoder.pe"][:nframes]       # vae.py:277,328
    outs = []
    for part, zi in (("body", z[0]), ("hands", z[1])):
        mem = zi.transpose(1, 0, 2) + sd["mem_pos_decoder.pe"][:n_chunks]                  # :281-286,329,339
        x = skip_decoder(sd, part + "_decoder.", queries, mem.astype(F32), num_layers, nhead, ~mask)
        outs.append(linear(x, sd[part + "_final_layer.weight"], sd[part + "_final_layer.bias"]))   # :359-360
    out = np.concatenate(outs, axis=-1)                                                     # :362
    out[~mask.T] = 0                                                                        # :368
    return out.transpose(1, 0, 2).copy()                                                    # :370
