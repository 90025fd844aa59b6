#!/usr/bin/env python
"""Generate tests/golden/vae_decode.npz from the REFERENCE ``ConvoFusionVae`` (build container only: needs
/root/reference on disk; the class is imported, never copied).

The reference module is built with the configs/modules/motion_vae.yaml values (arch 'encoder_decoder', 5 layers,
2 heads, ff 1024, pre-norm, gelu, sine PE, latent_dim [1, 128], nfeats 189; ablation MLP_DIST False, PE_TYPE
'convofusion'), loaded STRICTLY with oracle.vae_weights' seeded state dict, and ``decode`` is run on seeded latents.
Only inputs' seeds and the outputs are stored; weights and inputs are regenerated from their seeds by the tests.

Usage:  python tests/golden/make_golden_vae.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from convofusion.models.architectures.vae import ConvoFusionVae  # noqa: E402  (the reference)

from oracle import vae_ref, vae_weights  # noqa: E402

torch.set_grad_enabled(False)


def cases():
    """name -> (z [2, bs, n_chunks, 128], lengths)"""
    rng = np.random.Generator(np.random.PCG64(77))
    return {
        "ragged": (rng.standard_normal((2, 3, 8, 128), dtype=np.float32), [128, 100, 37]),
        "single": (rng.standard_normal((2, 1, 8, 128), dtype=np.float32), [64]),
        "long": (1.5 * rng.standard_normal((2, 2, 16, 128), dtype=np.float32), [200, 256]),
    }


def main():
    sd_np = vae_weights.make_state_dict()
    abl = SimpleNamespace(MLP_DIST=False, PE_TYPE="convofusion")
    m = ConvoFusionVae(ablation=abl, nfeats=189, latent_dim=[1, 128], ff_size=1024, num_layers=5, num_heads=2, dropout=0.1,
                       arch="encoder_decoder", normalize_before=True, activation="gelu", position_embedding="sine").eval()
    assert set(m.state_dict().keys()) == set(sd_np.keys())
    assert all(tuple(v.shape) == sd_np[k].shape for k, v in m.state_dict().items())
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    out = {}
    for name, (z, lengths) in cases().items():
        ref = m.decode(torch.from_numpy(z), lengths).numpy()
        mine = vae_ref.decode(sd_np, z, lengths)
        err = float(np.abs(ref - mine).max())
        print(name, ref.shape, "oracle vs reference max abs", err, "ref rms", float(np.sqrt((ref ** 2).mean())))
        assert err < 1e-4, err
        out[name] = ref
    np.savez_compressed(os.path.join(HERE, "vae_decode.npz"), **out)
    print("wrote vae_decode.npz")


if __name__ == "__main__":
    main()
