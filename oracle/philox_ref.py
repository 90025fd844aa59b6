"""CPU restatement of the product's on-device RNG -- TEST INFRASTRUCTURE (oracle/__init__.py).

The reference draws its Gaussian noise with ``torch.randn`` on the model device
(convofusion.py:412-416 for the initial latents; diffusers' ``randn_tensor`` inside
``scheduler.step`` for the per-step noise).  A CUDA/CPU torch generator stream cannot be
reproduced on another device, so the product defines its own counter-based stream (documented in
DESIGN.md) and this file restates it so tests can check the HIP kernel draw-for-draw:

  Philox4x32-10 (Salmon et al., SC'11), key = (seed_lo, seed_hi),
  counter = (element_index // 4, step_index, global_utterance_id, stream),
  stream 0 = per-step noise, stream 1 = initial latents;
  the 4 output words w0..w3 give 4 normals via two Box-Muller pairs:
      u = ((w >> 8) + 0.5) * 2^-24          (24-bit uniform in (0,1), exact in float32)
      r = sqrt(-2 ln u_a), th = 2*pi*u_b ;  z = (r cos th, r sin th)
  element e of an utterance's [L*128] block uses word pair (w0,w1) for e%4 in {0,1} and
  (w2,w3) for e%4 in {2,3}.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _uniform24(w):
    return ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def normal_block(seed, step, utt_id, stream, n_elems):
    """float32 [n_elems] normals for one utterance / step / stream."""
    assert n_elems % 4 == 0
    g = np.arange(n_elems // 4, dtype=np.uint32)
    w = philox4x32_10(g, np.uint32(step), np.uint32(utt_id), np.uint32(stream),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.empty((n_elems // 4, 4), dtype=np.float32)
    two_pi = np.float32(6.283185307179586)
    for pair in range(2):
        ua, ub = _uniform24(w[2 * pair]), _uniform24(w[2 * pair + 1])
        r = np.sqrt(np.float32(-2.0) * np.log(ua)).astype(np.float32)
        th = (two_pi * ub).astype(np.float32)
        out[:, 2 * pair] = r * np.cos(th)
        out[:, 2 * pair + 1] = r * np.sin(th)
    return out.reshape(-1)


def normal_tensor(seed, step, utt_ids, stream, L, latent=128):
    """[len(utt_ids), L, latent] normals, one independent block per global utterance id."""
    return np.stack([normal_block(seed, step, u, stream, L * latent).reshape(L, latent) for u in utt_ids])
