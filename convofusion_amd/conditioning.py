"""Conditioning producers either side of the denoising loop, on the HIP path.

* ``AudioConvEncoder`` -- drop-in for ``convofusion.models.architectures.audioenc.AudioConvEncoder``
  (reference audioenc.py:9-34): Mel frames [B, S, input_size] -> audio memory [B, S, latent_dim]; same constructor,
  same state-dict keys (``main.0 / main.3 / out_net``), inference only.
* ``TextAudioMotionFuser`` -- drop-in for ``convofusion.models.architectures.condfuser.TextAudioMotionFuser``
  (condfuser.py:8-50): the activity-bit and listener-id embedding look-ups that make the last two memories, plus the
  ``latent_proj`` MLP (defined by the reference, used by the dyadic path: ``convofusion_amd.dyadic``).
* The 7-way modality-guidance structure (convofusion.py:909-929) over these memories is
  ``convofusion_amd.sampler.build_guidance_batch`` (B + 1 distinct memories and row maps, no 7x batch).

The linear layers run through ``cfd_linear_act`` (libcfdenoise, float32 FMA chains); the look-ups are torch
indexing (plumbing).  There is no CPU fallback.
"""
import ctypes as C
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib

ACT_NONE, ACT_GELU, ACT_LEAKY01 = 0, 1, 2


def _engine_handle(device):
    """A libcfdenoise handle for the stand-alone producers (no denoiser weights needed)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("convofusion_amd.conditioning runs on an MI355X only (tensors must be on 'cuda'); no CPU fallback")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    h = _engine_handle.cache.get(idx)
    if h is None:
        h = _lib.create_handle(idx)
        _engine_handle.cache[idx] = h
    return h


_engine_handle.cache = {}


def linear_act(x, weight, bias, act=ACT_NONE, out=None):
    """``act(F.linear(x, weight, bias))`` on the device; x [..., K] float32 cuda, weight [N, K]."""
    if x.device.type != "cuda":
        raise RuntimeError("linear_act: tensors must live on the MI355X (no CPU fallback)")
    K = x.shape[-1]
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise ValueError(f"weight is {tuple(weight.shape)}, input has {K} features")
    xf = x.detach().to(torch.float32).contiguous().reshape(-1, K)
    w = weight.detach().to(device=x.device, dtype=torch.float32).contiguous()
    b = bias.detach().to(device=x.device, dtype=torch.float32).contiguous() if bias is not None else None
    fresh = out is None
    if out is None:
        out = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
    elif not (out.is_contiguous() and out.dtype == torch.float32 and out.numel() == xf.shape[0] * N):
        raise ValueError("out must be a contiguous float32 tensor of the result's size")
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cfd_linear_act(_engine_handle(x.device), C.c_void_p(xf.data_ptr()), xf.shape[0], K,
                                              C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
                                              N, int(act), C.c_void_p(out.data_ptr()), C.c_void_p(stream)))
    if not fresh:
        _lib.wrote(out)      # a caller's tensor rewritten through its raw pointer: torch must see it (it may be a memory of the next forward)
    return out


class AudioConvEncoder(nn.Module):
    """Reference audioenc.py:9-34.  ``main`` keeps the reference's Sequential indices (0 Linear, 1 Dropout,
    2 LeakyReLU, 3 Linear, 4 Dropout, 5 LeakyReLU) so checkpoints load strictly."""

    def __init__(self, input_size, hidden_size, latent_dim, **kwargs):
        super().__init__()
        output_size = latent_dim
        self.main = nn.Sequential(
            nn.Linear(input_size, hidden_size), nn.Dropout(0.1), nn.LeakyReLU(0.1),
            nn.Linear(hidden_size, output_size), nn.Dropout(0.1), nn.LeakyReLU(0.1))
        self.out_net = nn.Linear(output_size, output_size)
        self.max_seq_len = kwargs.get("max_seq_len")
        self.fps = kwargs.get("fps")
        self.sample_rate = kwargs.get("sample_rate")
        self.hop_length = kwargs.get("hop_length")
        if None not in (self.max_seq_len, self.fps, self.sample_rate, self.hop_length):
            self.audio_max_length = int((self.max_seq_len / self.fps) * self.sample_rate // self.hop_length + 1)

    def forward(self, inputs):
        if self.training:
            raise NotImplementedError("the HIP AudioConvEncoder is inference-only (call .eval())")
        h = linear_act(inputs, self.main[0].weight, self.main[0].bias, ACT_LEAKY01)
        h = linear_act(h, self.main[3].weight, self.main[3].bias, ACT_LEAKY01)
        return linear_act(h, self.out_net.weight, self.out_net.bias, ACT_NONE)


class TextAudioMotionFuser(nn.Module):
    """Reference condfuser.py:8-50: passes the three sequence memories through and looks up the activity-bit
    (3 x out_dim) and listener-id (36 x out_dim) embeddings; ``latent_proj`` (Linear, GELU, Linear, GELU) is
    declared by the reference and left unused by its forward -- ``project_latents`` runs it on the device."""

    def __init__(self, cfg, out_dim):
        super().__init__()
        lat1 = cfg.model.latent_dim[-1] if cfg is not None else 128
        try:
            self.vae_type = cfg.model.vae_type
        except Exception:
            try:
                self.vae_type = cfg.model.motion_vae.target.split(".")[-1].lower().replace("vae", "")
            except Exception:
                self.vae_type = "convofusion"
        self.out_dim = out_dim
        self.active_passive_emb = nn.Embedding(3, out_dim)
        self.lsn_id_emb = nn.Embedding(5 + 1 + 30, out_dim)
        self.latent_proj = nn.Sequential(nn.Linear(lat1 if self.vae_type != "no" else 189, 128), nn.GELU(),
                                         nn.Linear(128, out_dim), nn.GELU())

    def forward(self, spkemb, alsn, tlsn, active_passive_bit, lsn_id):
        apb = self.active_passive_emb(active_passive_bit.to(torch.int))                 # condfuser.py:41-44
        lsnemb = self.lsn_id_emb(torch.IntTensor(lsn_id).to(spkemb.device)).unsqueeze(1)   # :46-48
        return spkemb, alsn, tlsn, apb, lsnemb

    def project_latents(self, latents, out=None):
        """latent_proj(latents): [B, L, 128] -> [B, L, out_dim] (condfuser.py:22-27)."""
        h = linear_act(latents, self.latent_proj[0].weight, self.latent_proj[0].bias, ACT_GELU)
        return linear_act(h, self.latent_proj[2].weight, self.latent_proj[2].bias, ACT_GELU, out=out)


def default_fuser(out_dim=512, latent_dim=(1, 128)):
    """TextAudioMotionFuser with the shipped config values (configs/config_cf_beatdnd.yaml latent_dim [1, 128])."""
    cfg = SimpleNamespace(model=SimpleNamespace(latent_dim=list(latent_dim), vae_type="convofusion"))
    return TextAudioMotionFuser(cfg, out_dim)
