"""GPU tests (through the C ABI): the HIP VAE decoder and its float32 building blocks against the oracle / goldens."""
import os

import numpy as np
import pytest

from oracle import vae_ref, vae_weights
from tests.test_oracle_vae import ABL, G, KW, cases

pytestmark = pytest.mark.gpu


def _model():
    import torch
    from convofusion_amd.vae import ConvoFusionVae
    m = ConvoFusionVae(ablation=ABL, **KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vae_weights.make_state_dict().items()}, strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["ragged", "single", "long"])
def test_decode_matches_reference_golden(name):
    import torch
    z, lengths = cases()[name]
    got = _model().decode(torch.from_numpy(z).cuda(), lengths).cpu().numpy()
    assert got.shape == G[name].shape
    err = float(np.abs(got - G[name]).max())
    print(name, "HIP decode vs reference: max abs", err)
    assert err < 1e-4
    for b, n in enumerate(lengths):
        assert not got[b, n:].any()


@pytest.mark.parametrize("rows,D", [(1, 128), (77, 512), (5, 100), (3, 2048)])
def test_layer_norm(rows, D):
    import torch
    from convofusion_amd.vae import layer_norm
    rng = np.random.Generator(np.random.PCG64(rows + D))
    x = (3.0 * rng.standard_normal((rows, D)) + 1.0).astype(np.float32)
    ln = torch.nn.LayerNorm(D)
    with torch.no_grad():
        ln.weight.copy_(torch.from_numpy(rng.standard_normal(D).astype(np.float32)))
        ln.bias.copy_(torch.from_numpy(rng.standard_normal(D).astype(np.float32)))
    want = vae_ref.layer_norm(x, ln.weight.detach().numpy(), ln.bias.detach().numpy())
    got = layer_norm(torch.from_numpy(x).cuda(), ln.cuda()).cpu().numpy()
    assert float(np.abs(got - want).max()) < 2e-5


@pytest.mark.parametrize("Lq,Lk,bs,E,H,masked", [(5, 7, 2, 128, 2, True), (33, 200, 1, 64, 4, False), (16, 8, 3, 128, 2, False),
                                                 (3, 1000, 1, 32, 1, True)])
def test_mha_core(Lq, Lk, bs, E, H, masked):
    import torch
    from convofusion_amd.vae import mha
    rng = np.random.Generator(np.random.PCG64(Lq * 31 + Lk))
    attn = torch.nn.MultiheadAttention(E, H).eval()
    sd = {"in_proj_weight": (rng.standard_normal((3 * E, E)) / np.sqrt(E)).astype(np.float32),
          "in_proj_bias": (0.1 * rng.standard_normal(3 * E)).astype(np.float32),
          "out_proj.weight": (rng.standard_normal((E, E)) / np.sqrt(E)).astype(np.float32),
          "out_proj.bias": (0.1 * rng.standard_normal(E)).astype(np.float32)}
    attn.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    q = rng.standard_normal((Lq, bs, E)).astype(np.float32)
    kv = rng.standard_normal((Lk, bs, E)).astype(np.float32)
    kpm = None
    if masked:
        kpm = np.zeros((bs, Lk), bool)
        for b in range(bs):
            kpm[b, Lk - 1 - b - Lk // 3:] = True
    want = vae_ref.mha(sd, "", q, kv, kv, H, kpm)
    with torch.no_grad():
        got = mha(attn.cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(kv).cuda(), torch.from_numpy(kv).cuda(),
                  torch.from_numpy(kpm).cuda() if kpm is not None else None).cpu().numpy()
    assert float(np.abs(got - want).max()) < 2e-5


def test_attach_hip_decode_uses_the_modules_own_weights():
    """attach_hip_decode(vae) on a module with the reference's attribute layout (here: the mirror itself stands in for
    the reference class, which cannot travel to the GPU box) reroutes decode and snapshots the weights."""
    import torch
    from convofusion_amd.vae import attach_hip_decode
    host = _model()
    host.mlp_dist, host.pe_type = False, "convofusion"
    for blk in [host.body_decoder.middle_block]:
        blk.normalize_before = True
    mirror = attach_hip_decode(host)
    z, lengths = cases()["single"]
    got = host.decode(torch.from_numpy(z).cuda(), lengths).cpu().numpy()
    assert float(np.abs(got - G["single"]).max()) < 1e-4 and mirror is not host
