"""numpy restatement of the conditioning producers -- TEST INFRASTRUCTURE (oracle/__init__.py).

* ``audio_conv_encoder``: AudioConvEncoder.forward, convofusion/models/architectures/audioenc.py:12-21,29-34
  (Linear, Dropout, LeakyReLU(0.1), Linear, Dropout, LeakyReLU(0.1), then out_net Linear; dropout is the
  identity in eval mode).
* ``latent_proj``: TextAudioMotionFuser.latent_proj, condfuser.py:22-27 (Linear, GELU, Linear, GELU; nn.GELU()
  is the erf form).
* ``fuser_forward``: TextAudioMotionFuser.forward, condfuser.py:31-50 (embedding look-ups).
Pinned against outputs of the imported reference classes: tests/golden/conditioning.npz
(tests/golden/make_golden_conditioning.py).
"""
import math

import numpy as np

F32 = np.float32
_erf = np.vectorize(math.erf, otypes=[np.float64])


def linear(x, w, b):
    """F.linear in float32 (accumulated in float64 and rounded once: the order-free value every float32
    implementation must match to a few ulp)."""
    y = x.astype(np.float64) @ w.astype(np.float64).T
    if b is not None:
        y = y + b.astype(np.float64)
    return y.astype(F32)


def gelu(x):
    x64 = x.astype(np.float64)
    return (x64 * 0.5 * (1.0 + _erf(x64 / math.sqrt(2.0)))).astype(F32)


def leaky_relu(x, slope=0.1):
    return np.where(x > 0, x, F32(slope) * x).astype(F32)


def audio_conv_encoder(sd, inputs, prefix=""):
    h = leaky_relu(linear(inputs, sd[prefix + "main.0.weight"], sd[prefix + "main.0.bias"]))     # audioenc.py:14-16
    h = leaky_relu(linear(h, sd[prefix + "main.3.weight"], sd[prefix + "main.3.bias"]))          # :17-19
    return linear(h, sd[prefix + "out_net.weight"], sd[prefix + "out_net.bias"])                 # :21,34


def latent_proj(sd, latents, prefix=""):
    h = gelu(linear(latents, sd[prefix + "latent_proj.0.weight"], sd[prefix + "latent_proj.0.bias"]))   # condfuser.py:23-24
    return gelu(linear(h, sd[prefix + "latent_proj.2.weight"], sd[prefix + "latent_proj.2.bias"]))      # :25-26


def fuser_forward(sd, spkemb, alsn, tlsn, active_passive_bit, lsn_id, prefix=""):
    apb = sd[prefix + "active_passive_emb.weight"][np.asarray(active_passive_bit).astype(np.int64)]     # condfuser.py:41-44
    lsnemb = sd[prefix + "lsn_id_emb.weight"][np.asarray(lsn_id).astype(np.int64)][:, None, :]          # :46-48
    return spkemb, alsn, tlsn, apb, lsnemb
