"""Deterministic random state-dict with the reference Denoiser's key list and shapes.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference has no shippable checkpoint
(README links to an external download), so every parity test runs on random weights.  The
key list below is the one ``Denoiser.state_dict()`` produces for configs/modules/denoiser.yaml
(537 entries); ``tests/golden/make_golden.py`` proves it by a strict
``load_state_dict`` into the imported reference class.

All 9 layers get *independent* values (the reference deep-copies one layer at init,
cross_attention.py:687-688, which would hide layer-index bugs).
"""
import numpy as np

D = 512
FF = 1024
LAT = 128
MEM_NAMES = ("spkemb", "alsn", "tlsn", "apb", "lsnemb")  # att_fuser concat order, cross_attention.py:629
MHA_DECL_ORDER = ("spkemb", "tlsn", "alsn", "apb", "lsnemb")  # module declaration order, :451-459


def sine_pe(max_len, d_model=D):
    """position_encoding.py:118-125 (same arithmetic in float32)."""
    position = np.arange(0, max_len, dtype=np.float32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2).astype(np.float32) * np.float32(-np.log(10000.0) / d_model))
    pe = np.zeros((max_len, d_model), dtype=np.float32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe[:, None, :]  # [max_len, 1, d]


def key_shapes(num_layers=9):
    ks = [
        ("cond_params", (5,)),
        ("latent_embd.weight", (D, LAT)), ("latent_embd.bias", (D,)),
        ("latent_proj.weight", (LAT, D)), ("latent_proj.bias", (LAT,)),
        ("time_embedding.linear_1.weight", (D, D)), ("time_embedding.linear_1.bias", (D,)),
        ("time_embedding.linear_2.weight", (D, D)), ("time_embedding.linear_2.bias", (D,)),
        ("query_pos.pe", (1024, 1, D)), ("mem_pos.pe", (1024, 1, D)),
        ("bh_embedding.weight", (2, D)), ("condition_embedding.weight", (5, D)),
    ]
    for i in range(num_layers):
        p = f"decoder.layers.{i}."

        def mha(name):
            return [(p + name + ".in_proj_weight", (3 * D, D)), (p + name + ".in_proj_bias", (3 * D,)),
                    (p + name + ".out_proj.weight", (D, D)), (p + name + ".out_proj.bias", (D,))]

        def tb(name):
            return [(p + name + ".emb_layers.1.weight", (2 * D, D)), (p + name + ".emb_layers.1.bias", (2 * D,)),
                    (p + name + ".norm.weight", (D,)), (p + name + ".norm.bias", (D,)),
                    (p + name + ".out_layers.2.weight", (D, D)), (p + name + ".out_layers.2.bias", (D,))]

        ks += mha("self_attn") + tb("time_block1")
        for m in MHA_DECL_ORDER:
            ks += mha("multihead_attn_" + m)
        ks += [(p + "att_fuser.weight", (D, 5 * D)), (p + "att_fuser.bias", (D,))]
        ks += tb("time_block2")
        ks += [(p + "linear1.weight", (FF, D)), (p + "linear1.bias", (FF,)),
               (p + "linear2.weight", (D, FF)), (p + "linear2.bias", (D,))]
        for n in ("norm1", "norm2", "norm3"):
            ks += [(p + n + ".weight", (D,)), (p + n + ".bias", (D,))]
        for m in ("spkemb", "alsn", "tlsn", "apb", "lsnemb"):
            ks += [(p + m + "_norm.weight", (D,)), (p + m + "_norm.bias", (D,))]
    ks += [("decoder.norm.weight", (D,)), ("decoder.norm.bias", (D,))]
    return ks


def make_state_dict(seed=1234, num_layers=9, sharp=1.0):
    """name -> float32 ndarray.  ``sharp`` scales attention q/k projections so that the
    softmaxes are far from uniform (catches scale / mask / max-subtraction bugs)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shape in key_shapes(num_layers):
        if name.endswith(".pe"):
            sd[name] = sine_pe(shape[0])
        elif name == "cond_params":
            sd[name] = np.full(shape, 0.2, dtype=np.float32)
        elif name.endswith("embedding.weight"):
            sd[name] = rng.standard_normal(shape, dtype=np.float32)
        elif "norm" in name and name.endswith(".weight"):
            sd[name] = (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("bias"):
            sd[name] = (0.05 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        else:  # weight matrices: U(-a, a), a = 1.5/sqrt(fan_in)
            a = np.float32(1.5 / np.sqrt(shape[-1]))
            w = (rng.random(shape, dtype=np.float32) * 2 - 1) * a
            if name.endswith("in_proj_weight") and sharp != 1.0:
                w[: 2 * D] *= np.float32(sharp)
            sd[name] = w.astype(np.float32)
    return sd


def extend_pe(sd, max_len):
    """Return a copy of ``sd`` whose memory PE buffer covers ``max_len`` positions.  The
    reference caps memories at 1024 tokens (position_encoding.py:115,135); the synthetic
    1500-token benchmark config needs the closed-form buffer extended."""
    out = dict(sd)
    if max_len > sd["mem_pos.pe"].shape[0]:
        out["mem_pos.pe"] = sine_pe(max_len)
    return out


def make_state_dict_heavy(seed=777, num_layers=9, gain=20.0):
    """A stand-in for what trained checkpoints look like and seeded uniform weights do not (the reference's checkpoint is an external
    download, README.md:52-57): heavy tails.  Starts from ``make_state_dict(seed)`` and, with a generator of its own,
      * scales 1 % of the entries of every LayerNorm weight (decoder norms, memory norms, time-block norms) by 20,
      * scales 1 % of the ROWS of every FFN matrix and of every attention in-projection by 20 (outlier features),
      * shrinks three rows of every FFN / in-projection matrix by 1e-6 (their fp16 ``hi`` halves are subnormal, ``lo`` is 0),
    so that the split-pair operands meet large dynamic range inside one matrix, sharp softmaxes and near-dead features.
    ``gain`` is the outlier factor (20 for single forwards; the guided 50-step loop is tested at 8: at 20 it is CHAOTIC on these
    random weights -- the numpy oracle and the torch reference, both float32, are 9e-3 apart after 3 steps and 0.95 after 5
    (tests/golden/make_golden_heavy.py prints the figures), so no implementation can be pinned there; at 8 they stay 1e-4 apart over
    all 50 steps, like on the uniform weights)."""
    sd = make_state_dict(seed=seed, num_layers=num_layers)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    for name in sorted(sd):
        w = sd[name]
        if "norm" in name and name.endswith(".weight"):
            idx = rng.choice(w.shape[0], size=max(1, w.shape[0] // 100), replace=False)
            w[idx] *= np.float32(gain)
        elif name.endswith(("linear1.weight", "linear2.weight", "in_proj_weight")):
            rows = rng.choice(w.shape[0], size=max(1, w.shape[0] // 100), replace=False)
            w[rows] *= np.float32(gain)
            tiny = rng.choice(np.setdiff1d(np.arange(w.shape[0]), rows), size=3, replace=False)
            w[tiny] *= np.float32(1e-6)
    return sd
