"""Seeded synthetic ``ConvoFusionVae`` state dict -- TEST INFRASTRUCTURE (oracle/__init__.py).

The checkpoint layout of ``motion_vae.*`` (reference vae.py:33-150 with configs/modules/motion_vae.yaml: latent_dim
[1, 128], 5 layers, 2 heads, ff 1024, arch 'encoder_decoder', MLP_DIST False): 337 entries.  No pretrained checkpoint
is available offline, so parity runs on these seeded weights: tests/golden/make_golden_vae.py loads them (strict)
into the imported reference class, the tests load them into the mirror.
"""
import numpy as np

F32 = np.float32
D, FF, NL, NHEAD = 128, 1024, 5, 2
BODY, HANDS = 23 * 3, 40 * 3   # vae.py:53-54


def sine_pe(max_len=1024, d=D):
    """PositionEmbeddingSine1D buffer [max_len, 1, d] (position_encoding.py:118-125), float32 arithmetic like torch."""
    pe = np.zeros((max_len, d), F32)
    pos = np.arange(max_len, dtype=F32)[:, None]
    div = np.exp(np.arange(0, d, 2).astype(F32) * F32(-np.log(10000.0) / d)).astype(F32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe[:, None, :]


def _attn(pre):
    return [(pre + "in_proj_weight", (3 * D, D)), (pre + "in_proj_bias", (3 * D,)), (pre + "out_proj.weight", (D, D)),
            (pre + "out_proj.bias", (D,))]


def _layer(pre, decoder):
    ks = _attn(pre + "self_attn.")
    if decoder:
        ks += _attn(pre + "multihead_attn.")
    ks += [(pre + "linear1.weight", (FF, D)), (pre + "linear1.bias", (FF,)), (pre + "linear2.weight", (D, FF)), (pre + "linear2.bias", (D,))]
    for n in (("norm1", "norm2", "norm3") if decoder else ("norm1", "norm2")):
        ks += [(pre + n + ".weight", (D,)), (pre + n + ".bias", (D,))]
    return ks


def _skip(pre, decoder):
    nb = (NL - 1) // 2
    ks = [(pre + "norm.weight", (D,)), (pre + "norm.bias", (D,))]
    for i in range(nb):
        ks += _layer(f"{pre}input_blocks.{i}.", decoder)
    ks += _layer(pre + "middle_block.", decoder)
    for i in range(nb):
        ks += _layer(f"{pre}output_blocks.{i}.", decoder)
    for i in range(nb):
        ks += [(f"{pre}linear_blocks.{i}.weight", (D, 2 * D)), (f"{pre}linear_blocks.{i}.bias", (D,))]
    return ks


def key_shapes():
    ks = [("body_global_motion_token", (2, D)), ("hands_global_motion_token", (2, D)),
          ("query_pos_encoder.pe", (1024, 1, D)), ("query_pos_decoder.pe", (1024, 1, D)), ("mem_pos_decoder.pe", (1024, 1, D))]
    ks += _skip("body_encoder.", False) + _skip("hands_encoder.", False)
    ks += _skip("body_decoder.", True) + _skip("hands_decoder.", True)
    ks += [("body_skel_embedding.weight", (D, BODY)), ("body_skel_embedding.bias", (D,)),
           ("hands_skel_embedding.weight", (D, HANDS)), ("hands_skel_embedding.bias", (D,)),
           ("body_final_layer.weight", (BODY, D)), ("body_final_layer.bias", (BODY,)),
           ("hands_final_layer.weight", (HANDS, D)), ("hands_final_layer.bias", (HANDS,))]
    return ks


def make_state_dict(seed=4321):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for k, shp in key_shapes():
        if k.endswith(".pe"):
            sd[k] = sine_pe()
        elif "norm" in k and k.endswith("weight"):
            sd[k] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(F32)
        elif k.endswith("bias"):
            sd[k] = (0.05 * rng.standard_normal(shp)).astype(F32)
        elif len(shp) == 2 and not k.endswith("token"):
            sd[k] = (rng.standard_normal(shp) * (1.5 / np.sqrt(shp[1]))).astype(F32)   # sharp enough for non-uniform attention
        else:
            sd[k] = rng.standard_normal(shp).astype(F32)
    return sd
