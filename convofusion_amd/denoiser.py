"""Drop-in for ``convofusion.models.architectures.denoiser.Denoiser`` backed by libcfdenoise.

Same constructor signature (reference convofusion/models/architectures/denoiser.py:18-39), same
``forward(sample, timestep, encoder_hidden_states, lengths=None, mem_mask_dict={}, **kwargs)
-> (sample, att_mats)`` (:173-179,386) and the same 537-entry state-dict layout (SURVEY.md
section 8b), so ``instantiate_from_config`` (convofusion/config.py:24-31) can point at it by
changing ``target`` in configs/modules/denoiser.yaml, and ``model.load_state_dict`` of a reference
checkpoint loads strictly.  The torch modules below are parameter CONTAINERS only: the forward is
the HIP path.  Inference only (no autograd through the kernels).
"""
import copy
import ctypes as C
import math
import os

import torch
from torch import nn

from . import _lib

MEM_NAMES = _lib.MEM_NAMES
# declaration order of the reference layer (cross_attention.py:451-459) -- matters for state-dict order
_MHA_DECL_ORDER = ("spkemb", "tlsn", "alsn", "apb", "lsnemb")


def sinusoid_table(n_rows, dim=512, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000):
    """Rows t = 0..n_rows-1 of get_timestep_embedding (reference tools/embeddings.py:245-285), computed
    with the same torch float32 ops so the table is bit-identical to what the reference feeds its MLP."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = torch.arange(n_rows)[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb.contiguous()


def sine_pe(max_len, d_model=512):
    """PositionEmbeddingSine1D / SineBH buffer (reference operator/position_encoding.py:118-125)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).transpose(0, 1).contiguous()


def _contents_signature(tensors, force_checksum=False):
    """Per tensor (version counter or None, device checksum or None): what ``Denoiser.forward`` compares between two calls that pass the same
    tensor objects.  The version counter catches every in-place torch op for free.  Tensors created under ``torch.inference_mode()`` --
    everything the reference's test loop passes, pytorch_lightning's Trainer runs it in inference mode -- have no version counter
    (reading it raises); for those, and for every tensor with ``force_checksum``, a checksum of the bytes stands in: three strided integer
    sums per tensor, all tensors' sums fetched with ONE host read."""
    vers, todo = [], []
    for i, t in enumerate(tensors):
        v = None
        if t is not None:
            try:
                v = None if t.is_inference() else t._version
            except RuntimeError:      # "Inference tensors do not track version counter"
                v = None
            if v is None or force_checksum:
                todo.append(i)
        vers.append(v)
    sums = [None] * len(tensors)
    if todo:
        parts = []
        for i in todo:
            t = tensors[i].detach()
            flat = (t if t.is_contiguous() else t.contiguous()).reshape(-1)
            w = flat.view(torch.int32) if flat.element_size() == 4 else flat.view(torch.uint8)
            parts.append(torch.stack([w.sum(dtype=torch.int64), w[::7].sum(dtype=torch.int64), w[3::13].sum(dtype=torch.int64)]))
        got = torch.stack(parts).tolist()     # the one synchronising read
        for i, g in zip(todo, got):
            sums[i] = tuple(g)
    return list(zip(vers, sums))


class _PE(nn.Module):
    def __init__(self, d_model, max_len=1024):
        super().__init__()
        self.register_buffer("pe", sine_pe(max_len, d_model))


class _TimestepEmbedding(nn.Module):
    def __init__(self, channel, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(channel, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class _TimeBlock(nn.Module):
    def __init__(self, d, dropout):
        super().__init__()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(d, 2 * d))
        self.norm = nn.LayerNorm(d)
        self.out_layers = nn.Sequential(nn.SiLU(), nn.Dropout(p=dropout), nn.Linear(d, d))


class _Layer(nn.Module):
    """Parameter container mirroring TransformerDecoderLayer2Att.__init__ (cross_attention.py:444-489)."""

    def __init__(self, d, nhead, ff, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead, dropout=dropout)
        self.time_block1 = _TimeBlock(d, dropout)
        for m in _MHA_DECL_ORDER:
            setattr(self, "multihead_attn_" + m, nn.MultiheadAttention(d, 1, dropout=dropout))
        self.att_fuser = nn.Linear(d * 5, d)
        self.time_block2 = _TimeBlock(d, dropout)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.norm3 = nn.LayerNorm(d)
        for m in ("spkemb", "alsn", "tlsn", "apb", "lsnemb"):
            setattr(self, m + "_norm", nn.LayerNorm(d))


class _Decoder(nn.Module):
    def __init__(self, layer, num_layers, d):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])  # _get_clones, :687-688
        self.norm = nn.LayerNorm(d)


class Denoiser(nn.Module):

    def __init__(self,
                 ablation,
                 nfeats: int = 263,
                 condition: str = "text",
                 latent_dim: list = [1, 256],
                 ff_size: int = 1024,
                 num_layers: int = 6,
                 num_heads: int = 4,
                 dropout: float = 0.1,
                 normalize_before: bool = False,
                 activation: str = "gelu",
                 flip_sin_to_cos: bool = True,
                 return_intermediate_dec: bool = False,
                 position_embedding: str = "learned",
                 arch: str = "trans_enc",
                 freq_shift: int = 0,
                 guidance_scale: float = 7.5,
                 guidance_uncondp: float = 0.1,
                 text_encoded_dim: int = 768,
                 audio_encoded_dim: int = 512,
                 nclasses: int = 10,
                 **kwargs) -> None:
        super().__init__()
        self.latent_dim = latent_dim[-1]
        self.text_encoded_dim = text_encoded_dim
        self.audio_encoded_dim = audio_encoded_dim
        self.condition = condition
        self.arch = arch
        self.pe_type = ablation.DIFF_PE_TYPE
        self.causal_attn = ablation.CAUSAL_ATTN
        self.num_layers = num_layers
        # the configurations the HIP path implements (everything configs/modules/denoiser.yaml selects)
        if condition not in ("text+audio", "textaudio_uncond"):
            raise TypeError(f"condition type {condition} not supported")            # denoiser.py:113
        if self.pe_type != "convofusion":
            raise ValueError("Not Support PE type")                                   # :123
        if arch != "trans_dec":
            raise ValueError(f"Not supported architechure{arch}!")                    # :171 (trans_enc: VAE-only)
        if ablation.VAE_TYPE == "no":
            raise ValueError("diffusion-only (no VAE) mode is not implemented by the HIP path")
        if not normalize_before or activation != "gelu" or position_embedding not in ("sine", "v2") \
                or return_intermediate_dec or self.causal_attn or not flip_sin_to_cos or freq_shift != 0:
            raise ValueError("the HIP denoiser implements the shipped configuration only "
                             "(pre-norm, gelu, sine PE, flip_sin_to_cos, freq_shift 0, no causal mask)")
        d = text_encoded_dim
        self.latent_embd = nn.Linear(latent_dim[-1], d)
        self.latent_proj = nn.Linear(d, latent_dim[-1])
        self.time_embedding = _TimestepEmbedding(d, d)
        self.query_pos = _PE(d)
        self.mem_pos = _PE(d)
        self.bh_embedding = nn.Embedding(2, d)
        self.condition_embedding = nn.Embedding(5, d)
        self.cond_params = nn.Parameter(1 / 5 * torch.ones(5))
        self.decoder = _Decoder(_Layer(d, num_heads, ff_size, dropout), num_layers, d)
        self._cfg = dict(num_layers=num_layers, latent_dim=latent_dim[-1], d_model=d, ff_size=ff_size, num_heads=num_heads)
        self.return_attention = True   # set False to skip materialising att_mats (returns [])
        # Consecutive forwards that pass the SAME conditioning tensor objects at the same version counters reuse the memories'
        # timestep-independent projections (cfd_forward_same_memories).  A write that torch's version counter does not see (``t.data.copy_()``,
        # DLPack / raw-pointer writes by other libraries) is invisible to that test: set this to False if the caller does such writes between
        # forwards, or verify_constant_memories = True (env CFD_VERIFY_MEMORIES=1) to have every reuse checked against a device checksum
        # (one small reduction per tensor and one host read per forward) and refused with a RuntimeError when the contents moved.
        self.assume_constant_memories = True
        self.verify_constant_memories = os.environ.get("CFD_VERIFY_MEMORIES", "0") not in ("", "0")
        self._main = dict(handle=None, device=None, version=-1, mem_len=0)
        self._side = dict(handle=None, device=None, version=-1, mem_len=0)
        self._version = 0          # bumped whenever the parameters may have changed: engines re-upload lazily
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    # ---- engine management ---------------------------------------------------------------------
    def _mark_dirty(self):
        self._version += 1

    def _apply(self, fn, *a, **kw):
        self._version = getattr(self, "_version", 0) + 1
        return super()._apply(fn, *a, **kw)

    def __del__(self):
        try:
            for st in (self._main, self._side):
                if st["handle"] is not None:
                    _lib.load().cfd_destroy(st["handle"])
        except Exception:
            pass

    def engine(self, device=None, mem_len=0, side=False):
        """The libcfdenoise handle with the current weights uploaded (re-uploaded after
        load_state_dict / .to()).  ``mem_len``: longest memory the call will pass; the closed-form
        memory PE is extended past the checkpoint's 1024 rows when needed (SURVEY.md fact 4 -- the
        reference itself raises for S > 1024).  ``side=True``: a second handle with the same weights for forwards
        issued while a sampling run is open on the first (the run owns its handle's workspace and captured graph)."""
        device = torch.device(device) if device is not None else self.latent_embd.weight.device
        if device.type != "cuda":
            raise RuntimeError("convofusion_amd.Denoiser runs on an MI355X only (move the module to 'cuda'); "
                               "there is no CPU fallback")
        lib = _lib.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self._side if side else self._main
        if st["handle"] is None or st["device"] != idx:
            if st["handle"] is not None:
                lib.cfd_destroy(st["handle"])
            st["handle"] = _lib.create_handle(idx, **self._cfg)
            st["device"] = idx
            st["version"] = -1
        if st["version"] != self._version or mem_len > st["mem_len"]:
            sd = self.state_dict()
            need = max(mem_len, sd["mem_pos.pe"].shape[0])
            for name, t in sd.items():
                if name == "mem_pos.pe" and need > t.shape[0]:
                    # keep the checkpoint's rows bit for bit (they are what the reference adds; a re-computed
                    # table can differ from a loaded buffer in the last bit) and append the closed form
                    ext = sine_pe(need, self.text_encoded_dim).to(t.device, t.dtype)
                    t = torch.cat([t, ext[t.shape[0]:]], dim=0)
                t = t.detach().to(torch.float32).contiguous()
                _lib.check(lib.cfd_load_tensor(st["handle"], name.encode(), C.c_void_p(t.data_ptr()), t.numel(),
                                               1 if t.is_cuda else 0))
            _lib.check(lib.cfd_finalize_weights(st["handle"]))
            tab = sinusoid_table(1000, self.text_encoded_dim)
            _lib.check(lib.cfd_set_timestep_table(st["handle"], C.c_void_p(tab.data_ptr()), tab.shape[0]))
            st["version"] = self._version
            st["mem_len"] = need
        return st["handle"]

    @property
    def _handle(self):
        return self._main["handle"]

    # ---- conditioning helpers ------------------------------------------------------------------
    @staticmethod
    def pack_memories(encoder_hidden_states, mem_mask_dict, row_maps=None):
        """Build the cfd_memory array.  Returns (array, keepalive list)."""
        keep = []
        arr = (_lib.Memory * _lib.NUM_MEM)()
        if len(encoder_hidden_states) != _lib.NUM_MEM:
            raise ValueError("encoder_hidden_states must be the 5-tuple (spk_emb, alsn, tlsn, apb, lsnemb)")
        for j, name in enumerate(MEM_NAMES):
            m = encoder_hidden_states[j]
            if m.dim() != 3 or m.shape[-1] != 512:
                raise ValueError(f"memory {name}: expected [rows, S, 512], got {tuple(m.shape)}")
            m = m.detach().to(torch.float32).contiguous()
            keep.append(m)
            mask = (mem_mask_dict or {}).get(name)
            mp = None
            if mask is not None:
                mask = mask.to(device=m.device, dtype=torch.uint8).contiguous()
                if tuple(mask.shape) != (m.shape[0], m.shape[1]):
                    raise ValueError(f"key padding mask of {name}: expected {(m.shape[0], m.shape[1])}, got {tuple(mask.shape)}")
                keep.append(mask)
                mp = mask.data_ptr()
            rm = None
            if row_maps is not None and row_maps[j] is not None:
                r = row_maps[j].to(device=m.device, dtype=torch.int32).contiguous()
                keep.append(r)
                rm = r.data_ptr()
            arr[j] = _lib.Memory(m.data_ptr(), rm, mp, m.shape[0], m.shape[1])
        return arr, keep

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, lengths=None, mem_mask_dict=dict(), **kwargs):
        """``kwargs['side_engine']=True`` (extension): run on the second handle, for calls made while a sampling run is
        open on the first; every other keyword (``return_dict`` ...) is ignored like the reference does."""
        if torch.is_grad_enabled() and sample.requires_grad:
            raise NotImplementedError("the HIP denoiser is inference-only: call it under torch.no_grad()")
        if sample.dim() != 3 or sample.shape[-1] != self.latent_dim:
            raise ValueError(f"sample must be [batch, tokens, {self.latent_dim}]")
        lib = _lib.load()
        Be, L, _ = sample.shape
        h = self.engine(sample.device, mem_len=max(int(m.shape[1]) for m in encoder_hidden_states), side=bool(kwargs.get("side_engine", False)))
        x = sample.detach().to(torch.float32).contiguous()
        t = torch.as_tensor(timestep)
        if t.numel() == 1:
            ts = [int(t.reshape(-1)[0].item())]
        elif t.numel() == Be:
            ts = [int(v) for v in t.reshape(-1).tolist()]
        else:
            raise ValueError("timestep must be a scalar or have one entry per batch row")
        ts_arr = (C.c_int32 * len(ts))(*ts)
        mems, keep = self.pack_memories(encoder_hidden_states, mem_mask_dict)
        out = torch.empty_like(x)
        att_ptrs = (C.c_void_p * _lib.NUM_MEM)()
        att = []
        if self.return_attention:
            for j in range(_lib.NUM_MEM):
                a = torch.empty((Be, self.num_layers, L, int(encoder_hidden_states[j].shape[1])), dtype=torch.float32, device=x.device)
                att.append(a)
                att_ptrs[j] = a.data_ptr()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        # The reference's own loop (no convofusion_amd.install) calls this once per iteration with the SAME conditioning tensors
        # (convofusion.py:499-513): the same tensor objects with the same contents as in the previous call on this handle (the previous
        # call's tensors are kept alive here, so an address cannot come back with other data), and the library then reuses their
        # timestep-independent projections (cfd_forward_same_memories; it checks shapes, weights and what ran in between).
        srcs = list(encoder_hidden_states) + [(mem_mask_dict or {}).get(name) for name in MEM_NAMES]
        same, sig = False, None
        if self.assume_constant_memories:
            last = getattr(self, "_last_forward_memories", None)
            sig = _contents_signature(srcs, force_checksum=self.verify_constant_memories)
            same = last is not None and last[0] == h and len(last[1]) == len(srcs) and all(a is b for a, b in zip(last[1], srcs))
            # (a refusal needs checksums on BOTH sides: the call before verify_constant_memories was switched on recorded none -- then the
            #  signatures merely differ and the projections are made again)
            both = same and all((a[1] is None) == (b[1] is None) for a, b in zip(last[2], sig))
            if same and both and self.verify_constant_memories and [v for v, _ in last[2]] == [v for v, _ in sig] and last[2] != sig:
                self._last_forward_memories = None
                raise RuntimeError("convofusion_amd.Denoiser: a conditioning tensor was rewritten between two forwards without moving its version "
                                   "counter (t.data.copy_(), a DLPack / raw-pointer write): the reuse of its projections would be stale; bump it with "
                                   "torch.autograd.graph.increment_version(t) or set denoiser.assume_constant_memories = False")
            same = same and last[2] == sig
        self._last_forward_memories = None
        self.last_forward_reused = bool(same)     # (diagnostic: whether this call promised the library the previous call's memories)
        with torch.cuda.device(x.device):
            if same:
                _lib.check(lib.cfd_forward_same_memories(h))
            _lib.check(lib.cfd_forward(h, C.c_void_p(x.data_ptr()), Be, L, ts_arr, len(ts), mems, C.c_void_p(out.data_ptr()),
                                       att_ptrs if self.return_attention else None, C.c_void_p(stream)))
        if sig is not None:
            self._last_forward_memories = (h, srcs, sig)
        return (out, att)
