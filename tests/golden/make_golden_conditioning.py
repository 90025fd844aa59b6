#!/usr/bin/env python
"""Generate tests/golden/conditioning.npz from the REFERENCE classes (build container only: needs /root/reference
on disk; the reference is imported, never copied).

* ``convofusion.models.architectures.audioenc.AudioConvEncoder`` with the configs/modules/audio_encoder.yaml values
  (input_size 80, hidden_size 256, latent_dim 512).  The module imports ``convofusion.config`` (omegaconf is not
  installed here) only for a helper the class does not use, so an empty placeholder module is registered under
  that name for the import; no reference arithmetic is replaced.
* ``convofusion.models.architectures.condfuser.TextAudioMotionFuser`` (latent_dim [1, 128], out_dim 512): forward
  and its ``latent_proj`` MLP.
Weights are stored (they come from torch's default init under a fixed seed), inputs are stored, outputs are stored.

Usage:  python tests/golden/make_golden_conditioning.py
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
if "convofusion.config" not in sys.modules:
    try:
        import convofusion.config  # noqa: F401
    except Exception:
        ph = types.ModuleType("convofusion.config")
        ph.instantiate_from_config = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("placeholder"))
        sys.modules["convofusion.config"] = ph

from convofusion.models.architectures.audioenc import AudioConvEncoder  # noqa: E402  (the reference)
from convofusion.models.architectures.condfuser import TextAudioMotionFuser  # noqa: E402  (the reference)

torch.set_grad_enabled(False)
torch.manual_seed(1234)
g = torch.Generator().manual_seed(99)

enc = AudioConvEncoder(input_size=80, hidden_size=256, latent_dim=512, max_seq_len=128, fps=25, sample_rate=16000,
                       hop_length=160).eval()
mel = -40.0 + 25.0 * torch.randn((3, 37, 80), generator=g)          # Mel dB-like values, ragged length 37
mel[1, 20:] = -90.0                                                   # the unconditional fill (convofusion.py:911-912)
mel[1, 20:, 40:45] = 0.0
audio_out = enc(mel)

cfg = SimpleNamespace(model=SimpleNamespace(latent_dim=[1, 128], vae_type="convofusion"))
fus = TextAudioMotionFuser(cfg, 512).eval()
lat = 1.5 * torch.randn((2, 16, 128), generator=g)
proj_out = fus.latent_proj(lat)
bits = torch.tensor([0, 1, 2, 1])
lsn_id = [0, 3, 35, 7]
spk = torch.randn((4, 5, 512), generator=g)
_, _, _, apb, lsnemb = fus(spk, spk, spk, bits, lsn_id)

out = {"mel": mel.numpy(), "audio_out": audio_out.numpy(), "lat": lat.numpy(), "proj_out": proj_out.numpy(),
       "bits": bits.numpy(), "lsn_id": np.asarray(lsn_id), "apb": apb.numpy(), "lsnemb": lsnemb.numpy()}
for k, v in enc.state_dict().items():
    out["enc." + k] = v.numpy()
for k, v in fus.state_dict().items():
    out["fus." + k] = v.numpy()
np.savez_compressed(os.path.join(HERE, "conditioning.npz"), **out)
print("wrote conditioning.npz:", {k: v.shape for k, v in out.items()})
