"""Helpers for the -m gpu tests: product objects loaded with the seeded test weights."""
import ctypes as C
import functools
from types import SimpleNamespace

import numpy as np
import torch

from tests.helpers import state_dict

ABL = SimpleNamespace(SKIP_CONNECT=True, VAE_TYPE="convofusion", DIFF_PE_TYPE="convofusion", CAUSAL_ATTN=False)
DENOISER_KW = dict(nfeats=189, condition="text+audio", latent_dim=[1, 128], ff_size=1024, num_layers=9, num_heads=4,
                   dropout=0.1, normalize_before=True, activation="gelu", flip_sin_to_cos=True,
                   return_intermediate_dec=False, position_embedding="sine", arch="trans_dec", freq_shift=0,
                   guidance_scale=7.5, guidance_uncondp=0.1, text_encoded_dim=512, audio_encoded_dim=512, nclasses=10)
SCHED_KW = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                clip_sample=True)


@functools.lru_cache(maxsize=3)
def hip_denoiser(seed=1234, sharp=1.0):
    from convofusion_amd.denoiser import Denoiser
    m = Denoiser(ablation=ABL, **DENOISER_KW)
    sd = {k: torch.from_numpy(v) for k, v in state_dict(seed, sharp).items()}
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def to_dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def dev_inputs(inp):
    mems = [to_dev(m) for m in inp["memories"]]
    masks = {k: to_dev(v) for k, v in inp["masks"].items()}
    return mems, masks


def read_debug(m, what, shape):
    from convofusion_amd import _lib
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    _lib.check(_lib.load().cfd_debug_read(m._handle, what.encode(), C.c_void_p(out.data_ptr()), out.numel()))
    return out.cpu().numpy()
