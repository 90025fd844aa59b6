"""``ConvoFusionVae`` with the decoder on the HIP path.

Drop-in for ``convofusion.models.architectures.vae.ConvoFusionVae`` (reference vae.py:33-372) as far as the
generation flow needs it: same constructor, same 337-entry state-dict layout (so ``motion_vae.*`` of a reference
checkpoint loads strictly), and ``decode(z, lengths)`` -- the step right after the denoising loop
(test.py / unbounded_synthesis.py: latents -> 189 motion features per frame).  ``encode`` / ``forward`` belong to
training and evaluation and are not provided: they raise.

``decode`` restates vae.py:268-372 for arch 'encoder_decoder' / PE_TYPE 'convofusion' with every arithmetic step in
libcfdenoise float32 kernels (cfd_linear_act, cfd_layer_norm, cfd_mha, cfd_add, cfd_zero_rows); torch only slices,
concatenates and allocates.  Two SkipTransformerDecoders (cross_attention.py:66-125) of pre-norm
TransformerDecoderLayers (:361-382): d_model 128, 2 heads, ff 1024, 5 layers -- tiny next to the loop (a few ms per
batch), so the kernels are plain and exact rather than tuned.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .conditioning import ACT_GELU, ACT_NONE, _engine_handle, linear_act


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def layer_norm(x, norm):
    """nn.LayerNorm over the last dimension on the device."""
    x = x.contiguous()
    out = torch.empty_like(x)
    D = x.shape[-1]
    w = norm.weight.detach().to(x.device, torch.float32).contiguous()
    b = norm.bias.detach().to(x.device, torch.float32).contiguous()
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cfd_layer_norm(_engine_handle(x.device), _ptr(x), x.numel() // D, D, _ptr(w), _ptr(b),
                                              C.c_float(norm.eps), _ptr(out), _stream(x)))
    return out


def add_(x, y):
    """x += y on the device (x, y contiguous float32 of equal size)."""
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().cfd_add(_engine_handle(x.device), _ptr(x), _ptr(y), x.numel(), _stream(x)))
        _lib.wrote(x)
    return x


def mha(attn, query, key, value, key_padding_mask=None):
    """``nn.MultiheadAttention.forward(query, key, value, key_padding_mask=...)[0]`` for [L, N, E] tensors."""
    E, H = attn.embed_dim, attn.num_heads
    W, B = attn.in_proj_weight, attn.in_proj_bias
    q = linear_act(query, W[:E], B[:E])
    k = linear_act(key, W[E:2 * E], B[E:2 * E])
    v = linear_act(value, W[2 * E:], B[2 * E:])
    Lq, N, _ = q.shape
    Lk = k.shape[0]
    out = torch.empty_like(q)
    kpm = None
    if key_padding_mask is not None:
        kpm = key_padding_mask.to(device=q.device, dtype=torch.uint8).contiguous()
    with torch.cuda.device(q.device):
        _lib.check(_lib.load().cfd_mha(_engine_handle(q.device), _ptr(q), _ptr(k), _ptr(v), Lq, Lk, N, E, H,
                                       _ptr(kpm) if kpm is not None else None, _ptr(out), _stream(q)))
    return linear_act(out, attn.out_proj.weight, attn.out_proj.bias)


class _EncoderLayer(nn.Module):
    """Parameter layout of TransformerEncoderLayer (cross_attention.py:250-308); only held for checkpoint loading."""

    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _DecoderLayer(nn.Module):
    """TransformerDecoderLayer, pre-norm, eval mode (cross_attention.py:311-382)."""

    def __init__(self, d, nhead, ff):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, nhead)
        self.multihead_attn = nn.MultiheadAttention(d, nhead)
        self.linear1 = nn.Linear(d, ff)
        self.linear2 = nn.Linear(ff, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)
        self.norm3 = nn.LayerNorm(d)

    def run(self, tgt, memory, tgt_key_padding_mask):
        t2 = layer_norm(tgt, self.norm1)                                                      # :368
        add_(tgt, mha(self.self_attn, t2, t2, t2, tgt_key_padding_mask))                      # :369-372
        t2 = layer_norm(tgt, self.norm2)                                                      # :373
        add_(tgt, mha(self.multihead_attn, t2, memory, memory, None))                         # :374-378
        t2 = layer_norm(tgt, self.norm3)                                                      # :379
        h = linear_act(t2, self.linear1.weight, self.linear1.bias, ACT_GELU)                  # :380
        add_(tgt, linear_act(h, self.linear2.weight, self.linear2.bias, ACT_NONE))            # :380-381
        return tgt


class _Skip(nn.Module):
    """SkipTransformerEncoder / SkipTransformerDecoder parameter layout and the decoder's forward (:66-125)."""

    def __init__(self, layer_cls, d, nhead, ff, num_layers):
        super().__init__()
        assert num_layers % 2 == 1
        nb = (num_layers - 1) // 2
        self.norm = nn.LayerNorm(d)
        self.input_blocks = nn.ModuleList([layer_cls(d, nhead, ff) for _ in range(nb)])
        self.middle_block = layer_cls(d, nhead, ff)
        self.output_blocks = nn.ModuleList([layer_cls(d, nhead, ff) for _ in range(nb)])
        self.linear_blocks = nn.ModuleList([nn.Linear(2 * d, d) for _ in range(nb)])

    def run(self, tgt, memory, tgt_key_padding_mask):
        x, xs = tgt, []
        for blk in self.input_blocks:
            x = blk.run(x, memory, tgt_key_padding_mask)
            xs.append(x.clone())
        x = self.middle_block.run(x, memory, tgt_key_padding_mask)
        for blk, lin in zip(self.output_blocks, self.linear_blocks):
            x = linear_act(torch.cat([x, xs.pop()], dim=-1), lin.weight, lin.bias)
            x = blk.run(x, memory, tgt_key_padding_mask)
        return layer_norm(x, self.norm)


class _SinePE(nn.Module):
    def __init__(self, d, max_len=1024):
        super().__init__()
        from .denoiser import sine_pe
        self.register_buffer("pe", sine_pe(max_len, d))


class ConvoFusionVae(nn.Module):
    def __init__(self, ablation, nfeats, latent_dim=[1, 256], ff_size=1024, num_layers=9, num_heads=4, dropout=0.1,
                 arch="all_encoder", normalize_before=False, activation="gelu", position_embedding="learned", **kwargs):
        super().__init__()
        if arch != "encoder_decoder" or not normalize_before or activation != "gelu" or position_embedding not in ("sine", "v2"):
            raise ValueError("the HIP VAE decoder implements the shipped configuration only (configs/modules/motion_vae.yaml: "
                             "arch 'encoder_decoder', pre-norm, gelu, sine position embedding)")
        if getattr(ablation, "PE_TYPE", "convofusion") != "convofusion":
            raise ValueError("Not support position encoding type!")          # vae.py:350
        if getattr(ablation, "MLP_DIST", False):
            raise ValueError("MLP_DIST=True is not supported by the HIP VAE mirror")
        self.latent_size, self.latent_dim = latent_dim[0], latent_dim[-1]
        self.body_nfeats, self.hands_nfeats = 23 * 3, 40 * 3                  # vae.py:53-54
        self.arch, self.num_heads, self.num_layers = arch, num_heads, num_layers
        d = self.latent_dim
        self.body_global_motion_token = nn.Parameter(torch.randn(self.latent_size * 2, d))
        self.hands_global_motion_token = nn.Parameter(torch.randn(self.latent_size * 2, d))
        self.query_pos_encoder = _SinePE(d)
        self.query_pos_decoder = _SinePE(d)
        self.mem_pos_decoder = _SinePE(d)
        self.body_encoder = _Skip(_EncoderLayer, d, num_heads, ff_size, num_layers)
        self.hands_encoder = _Skip(_EncoderLayer, d, num_heads, ff_size, num_layers)
        self.body_decoder = _Skip(_DecoderLayer, d, num_heads, ff_size, num_layers)
        self.hands_decoder = _Skip(_DecoderLayer, d, num_heads, ff_size, num_layers)
        self.body_skel_embedding = nn.Linear(self.body_nfeats, d)
        self.hands_skel_embedding = nn.Linear(self.hands_nfeats, d)
        self.body_final_layer = nn.Linear(d, self.body_nfeats)
        self.hands_final_layer = nn.Linear(d, self.hands_nfeats)

    def forward(self, features, lengths=None):
        raise NotImplementedError("convofusion_amd.vae.ConvoFusionVae provides decode() only (generation); training and "
                                  "evaluation use the reference module")

    def encode(self, features, lengths=None):
        raise NotImplementedError("convofusion_amd.vae.ConvoFusionVae provides decode() only (generation)")

    @torch.no_grad()
    def decode(self, z, lengths):
        """z [2, bs, n_chunks, latent_dim] (body | hands), lengths: frames per sequence -> feats [bs, nframes, 189]."""
        if z.device.type != "cuda":
            raise RuntimeError("the HIP VAE decoder runs on an MI355X only (move the module and z to 'cuda'); no CPU fallback")
        _, bs, n_chunks, D = z.shape
        dev = z.device
        lens = torch.as_tensor(list(lengths), device=dev)
        nframes = int(max(lengths))
        mask = torch.arange(nframes, device=dev).expand(bs, nframes) < lens.unsqueeze(1)      # lengths_to_mask
        # queries = zeros + PE = the PE rows themselves (vae.py:277,328)
        queries = self.query_pos_decoder.pe[:nframes].to(torch.float32).expand(nframes, bs, D).contiguous()
        pe_mem = self.mem_pos_decoder.pe[:n_chunks].to(torch.float32).expand(n_chunks, bs, D).contiguous()
        kpm = ~mask
        outs = []
        for dec, fin, zi in ((self.body_decoder, self.body_final_layer, z[0]), (self.hands_decoder, self.hands_final_layer, z[1])):
            mem = zi.detach().to(torch.float32).permute(1, 0, 2).contiguous()                 # :281-286
            add_(mem, pe_mem)                                                                  # :329,339
            x = dec.run(queries.clone(), mem, kpm)                                            # :330-347
            outs.append(linear_act(x, fin.weight, fin.bias))                                  # :359-360
        out = torch.cat(outs, dim=-1).contiguous()                                            # :362
        keep = mask.t().contiguous().to(torch.uint8)                                          # rows are (frame, batch)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().cfd_zero_rows(_engine_handle(dev), _ptr(out), _ptr(keep), nframes * bs, out.shape[-1], _stream(out)))
            _lib.wrote(out)
        return out.permute(1, 0, 2)                                                           # :370


def attach_hip_decode(vae):
    """Route ``vae.decode`` of a REFERENCE ``ConvoFusionVae`` instance (kept for encode / training) to the HIP path:
    builds the mirror from the module's own hyper-parameters and weights and replaces the bound method.  The mirror
    takes a snapshot of the weights: call again after loading a different checkpoint.  Returns the mirror."""
    dec = vae.body_decoder
    nl = 2 * len(dec.input_blocks) + 1
    attn = dec.middle_block.self_attn
    from types import SimpleNamespace
    m = ConvoFusionVae(ablation=SimpleNamespace(MLP_DIST=getattr(vae, "mlp_dist", False), PE_TYPE=getattr(vae, "pe_type", "convofusion")),
                       nfeats=vae.body_nfeats + vae.hands_nfeats, latent_dim=[vae.latent_size, vae.latent_dim],
                       ff_size=dec.middle_block.linear1.out_features, num_layers=nl, num_heads=attn.num_heads,
                       arch=vae.arch, normalize_before=dec.middle_block.normalize_before, activation="gelu",
                       position_embedding="sine")
    m.load_state_dict(vae.state_dict(), strict=True)
    m = m.to(next(vae.parameters()).device).eval()
    vae.decode = m.decode
    return m
