"""Word-excitation guidance (WEG) on the HIP path.

The reference steers the first ``max_iter_to_alter`` iterations of the sampling loop with the gradient of an
attend-and-excite objective on the listener-text attention maps of the text-only guidance chunk
(convofusion/models/modeltype/convofusion.py:437-496, ``iterative_refinement_step`` :298-388,
convofusion/models/tools/word_excitation_guidance.py:11-81).  It gets d(loss)/d(latents) from torch autograd over
``Denoiser.forward``; here the backward pass is written out: a float32 forward that keeps its activations and the
reverse sweep, both strings of libcfdenoise launches (``cfd_gemm_f32`` on strided views -- every transpose is a view --
``cfd_softmax(_bwd)``, ``cfd_layer_norm(_bwd)``, ``cfd_ew``, ``cfd_weg_focus``).  The product path is ``loss_and_grad`` =
``cfd_weg_eval``: the library enqueues the whole evaluation itself (csrc/weg_eval.hpp; ~7x faster than one C call per
kernel from Python).  ``forward_saved`` / ``attention_focus_loss`` / ``backward_to_sample`` string the same kernels
together from here, one launch per call -- the inspectable form (``loss_and_grad_stepwise``) the tests compare against.
torch allocates, slices and permutes; it does no arithmetic.  There is no CPU fallback.

The gradient only flows through the query side: memories, time embedding and weights are constants, so the sweep
needs no key / value gradients for the five cross-attentions and no weight gradients at all.

Public names follow the reference module (``aggregate_attentions`` + ``get_max_attention_at_indices`` +
``compute_attention_focus_loss`` are one fused kernel: ``attention_focus_loss``; ``update_latent`` takes the gradient
instead of a graph) and the loop branch is ``weg_update`` / ``iterative_refinement_step``.
"""
import ctypes as C
import functools
import math

import numpy as np
import torch

from . import _lib
from .denoiser import MEM_NAMES, Denoiser, sinusoid_table

TLSN = 2  # the listener-text memory (denoiser.py:220; ``text_only_att_mats[2]``, convofusion.py:464)
EW_SILU, EW_GELU, EW_SILU_BWD, EW_GELU_BWD, EW_AXPY, EW_ADD_BCAST, EW_MODULATE, EW_MODULATE_BWD = range(8)


# ----------------------------------------------------------------------------- launch wrappers (no arithmetic in torch)
class _Ops:
    def __init__(self, handle, device):
        self.lib = _lib.load()
        self.h = handle
        self.dev = device
        self.stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    @staticmethod
    def _mat(t):
        if t.dtype != torch.float32 or not t.is_cuda or t.dim() < 2 or t.dim() > 4:
            raise ValueError("matrix views must be 2-4 dimensional float32 device tensors")
        st, sh = t.stride(), t.shape
        b2 = st[-3] if t.dim() >= 3 else 0
        b1 = st[-4] if t.dim() == 4 else 0
        nb2 = sh[-3] if t.dim() >= 3 else 1
        nb1 = sh[-4] if t.dim() == 4 else 1
        return _lib.Mat(t.data_ptr(), st[-2], st[-1], b1, b2), nb1, nb2

    def gemm(self, a, b, out, bias=None, alpha=1.0, accumulate=False):
        """out[..., m, n] = alpha * sum_k a[..., m, k] b[..., k, n] + bias[n] (+ out); any strides."""
        A, a1, a2 = self._mat(a)
        Bm, b1, b2 = self._mat(b)
        Cm, c1, c2 = self._mat(out)
        M, K = a.shape[-2], a.shape[-1]
        N = b.shape[-1]
        if b.shape[-2] != K or out.shape[-2] != M or out.shape[-1] != N or (a1, a2) != (b1, b2) or (a1, a2) != (c1, c2):
            raise ValueError(f"gemm shapes do not agree: {tuple(a.shape)} x {tuple(b.shape)} -> {tuple(out.shape)}")
        _lib.check(self.lib.cfd_gemm_f32(self.h, M, N, K, a1, a2, C.byref(A), C.byref(Bm), C.byref(Cm),
                                         C.c_void_p(bias.data_ptr()) if bias is not None else None, float(alpha), 1 if accumulate else 0,
                                         self.stream))
        return out

    def linear(self, x, w, b=None):
        """F.linear(x, w, b) for contiguous x [..., K]."""
        K, N = x.shape[-1], w.shape[0]
        out = self.new(*x.shape[:-1], N)
        self.gemm(x.reshape(-1, K), w.t(), out.view(-1, N), b)
        return out

    def linear_bwd(self, dy, w, out=None, accumulate=False):
        """Gradient of F.linear(x, w) with respect to x: dy @ w."""
        N, K = w.shape
        if out is None:
            out = self.new(*dy.shape[:-1], K)
        self.gemm(dy.reshape(-1, N), w, out.view(-1, K), None, 1.0, accumulate)
        return out

    def softmax_(self, scores, mask, rows_per_batch):
        Lk = scores.shape[-1]
        _lib.check(self.lib.cfd_softmax(self.h, C.c_void_p(scores.data_ptr()), scores.numel() // Lk, Lk,
                                        C.c_void_p(mask.data_ptr()) if mask is not None else None, rows_per_batch, self.stream))
        return scores

    def softmax_bwd_(self, p, dp, extra=None):
        Lk = p.shape[-1]
        _lib.check(self.lib.cfd_softmax_bwd(self.h, C.c_void_p(p.data_ptr()), C.c_void_p(dp.data_ptr()),
                                            C.c_void_p(extra.data_ptr()) if extra is not None else None, p.numel() // Lk, Lk, self.stream))
        return dp

    def layer_norm(self, x, g, b, eps=1e-5):
        out = torch.empty_like(x)
        D = x.shape[-1]
        _lib.check(self.lib.cfd_layer_norm(self.h, C.c_void_p(x.data_ptr()), x.numel() // D, D, C.c_void_p(g.data_ptr()), C.c_void_p(b.data_ptr()),
                                           C.c_float(eps), C.c_void_p(out.data_ptr()), self.stream))
        return out

    def layer_norm_bwd(self, x, g, dy, dx, accumulate, eps=1e-5):
        D = x.shape[-1]
        _lib.check(self.lib.cfd_layer_norm_bwd(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(dy.data_ptr()),
                                               C.c_void_p(dx.data_ptr()), x.numel() // D, D, C.c_float(eps), 1 if accumulate else 0, self.stream))
        return dx

    def ew(self, op, a, b=None, out=None, D=0, R1=0, s0=0, s1=0, alpha=0.0):
        if out is None:
            out = torch.empty_like(a)
        _lib.check(self.lib.cfd_ew(self.h, op, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
                                   C.c_void_p(out.data_ptr()), a.numel(), D, R1, s0, s1, float(alpha), self.stream))
        return out

    def add_(self, x, y):
        _lib.check(self.lib.cfd_add(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.numel(), self.stream))
        return x


def _w(t, dev):
    t = t.detach()
    if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(device=dev, dtype=torch.float32).contiguous()
    return t


# ----------------------------------------------------------------------------- attention with saved activations
def _mha_fwd(ops, attn, query, memory, mask, dev):
    """nn.MultiheadAttention(query, memory, memory, key_padding_mask=mask) on [T, B, E] / [S, B, E] tensors.
    Returns (out [T, B, E], probabilities [B, H, T, S], saved)."""
    E, H = attn.embed_dim, attn.num_heads
    hd = E // H
    W, Bi = _w(attn.in_proj_weight, dev), _w(attn.in_proj_bias, dev)
    T, B, _ = query.shape
    S = memory.shape[0]
    q = ops.linear(query, W[:E], Bi[:E])
    k = ops.linear(memory, W[E:2 * E], Bi[E:2 * E])
    v = ops.linear(memory, W[2 * E:], Bi[2 * E:])
    p = ops.new(B, H, T, S)
    scale = math.sqrt(1.0 / hd)                                                            # q * sqrt(1 / head_dim)
    ops.gemm(q.view(T, B, H, hd).permute(1, 2, 0, 3), k.view(S, B, H, hd).permute(1, 2, 3, 0), p, alpha=scale)
    ops.softmax_(p, mask, H * T)
    o = ops.new(T, B, E)
    ops.gemm(p, v.view(S, B, H, hd).permute(1, 2, 0, 3), o.view(T, B, H, hd).permute(1, 2, 0, 3))
    out = ops.linear(o, _w(attn.out_proj.weight, dev), _w(attn.out_proj.bias, dev))
    return out, p, dict(q=q, k=k, v=v, p=p, W=W, Wo=_w(attn.out_proj.weight, dev), T=T, B=B, S=S, E=E, H=H, scale=scale)


def _mha_bwd(ops, sv, dout, d_prob=None, self_attention=False):
    """Gradient with respect to the query input (self-attention: query = key = value, all three paths).
    ``dout`` [T, B, E] or None (nothing arrives through the output); ``d_prob`` [B, H, T, S] arrives at the probabilities."""
    T, B, S, E, H = sv["T"], sv["B"], sv["S"], sv["E"], sv["H"]
    hd = E // H
    p = sv["p"]
    hv = lambda t, n: t.view(n, B, H, hd).permute(1, 2, 0, 3)          # [n, B, E] -> [B, H, n, hd]
    if dout is not None:
        do = ops.linear_bwd(dout, sv["Wo"])
        dp = ops.new(B, H, T, S)
        ops.gemm(hv(do, T), sv["v"].view(S, B, H, hd).permute(1, 2, 3, 0), dp)
        ops.softmax_bwd_(p, dp, d_prob)
    else:
        do = None
        dp = ops.softmax_bwd_(p, d_prob.clone())
    dq = ops.new(T, B, E)
    ops.gemm(dp, hv(sv["k"], S), hv(dq, T), alpha=sv["scale"])
    dx = ops.linear_bwd(dq, sv["W"][:E])
    if self_attention:
        dk = ops.new(S, B, E)
        ops.gemm(dp.transpose(2, 3), hv(sv["q"], T), hv(dk, S), alpha=sv["scale"])
        ops.linear_bwd(dk, sv["W"][E:2 * E], dx, accumulate=True)
        dv = ops.new(S, B, E)
        ops.gemm(p.transpose(2, 3), hv(do, T), hv(dv, S))
        ops.linear_bwd(dv, sv["W"][2 * E:], dx, accumulate=True)
    return dx


def _time_block_fwd(ops, tb, x, silu_temb, dev):
    """TimeBlock.forward (cross_attention.py:426-439) for one shared time embedding row."""
    D = x.shape[-1]
    e = ops.linear(silu_temb, _w(tb.emb_layers[1].weight, dev), _w(tb.emb_layers[1].bias, dev))        # [1, 2 D], scale first
    n = ops.layer_norm(x, _w(tb.norm.weight, dev), _w(tb.norm.bias, dev))
    h = ops.ew(EW_MODULATE, n, e, D=D, R1=1)
    out = ops.linear(ops.ew(EW_SILU, h), _w(tb.out_layers[2].weight, dev), _w(tb.out_layers[2].bias, dev))
    return out, dict(x=x, h=h, e=e)


def _time_block_bwd(ops, tb, sv, g, dev):
    """g += d TimeBlock(x) / dx applied to g (the block sits on a residual branch)."""
    D = g.shape[-1]
    dh = ops.linear_bwd(g, _w(tb.out_layers[2].weight, dev))
    ops.ew(EW_SILU_BWD, dh, sv["h"], out=dh)
    ops.ew(EW_MODULATE_BWD, dh, sv["e"], out=dh, D=D, R1=1)
    ops.layer_norm_bwd(sv["x"], _w(tb.norm.weight, dev), dh, g, accumulate=True)
    return g


# ----------------------------------------------------------------------------- the differentiated forward and its reverse sweep
def forward_saved(denoiser, sample, timestep, encoder_hidden_states, mem_mask_dict=None):
    """Float32 ``Denoiser.forward`` (denoiser.py:173-386) up to the last cross-attention, keeping what the backward
    needs.  Returns (att_tlsn [B, layers, L, S_text], saved)."""
    if not isinstance(denoiser, Denoiser):
        raise TypeError("denoiser must be a convofusion_amd.denoiser.Denoiser")
    dev = sample.device
    if dev.type != "cuda":
        raise RuntimeError("WEG runs on an MI355X only (tensors must be on 'cuda'); no CPU fallback")
    if len(encoder_hidden_states) != 5:
        raise ValueError("encoder_hidden_states must be the 5-tuple (spk_emb, alsn, tlsn, apb, lsnemb)")
    B, L, _ = sample.shape
    D = denoiser.text_encoded_dim
    if L % 2 or L // 2 > denoiser.query_pos.pe.shape[0]:
        raise ValueError("latent length must be even and L/2 <= query PE length")      # position_encoding.py:160-161
    masks = dict(mem_mask_dict or {})
    with torch.cuda.device(dev):
        ops = _Ops(denoiser.engine(dev), dev)
        x = ops.linear(sample.detach().to(torch.float32).permute(1, 0, 2).contiguous(), _w(denoiser.latent_embd.weight, dev),
                       _w(denoiser.latent_embd.bias, dev))                             # [L, B, D]  denoiser.py:183-187
        te = denoiser.time_embedding
        tab = getattr(denoiser, "_weg_time_table", None)                                # get_timestep_embedding rows (load-time table)
        if tab is None or tab.device != dev or tab.shape[0] <= int(timestep):
            tab = sinusoid_table(max(1000, int(timestep) + 1), D).to(dev)
            denoiser._weg_time_table = tab
        trow = tab[int(timestep):int(timestep) + 1]
        temb = ops.linear(ops.ew(EW_SILU, ops.linear(trow, _w(te.linear_1.weight, dev), _w(te.linear_1.bias, dev))),
                          _w(te.linear_2.weight, dev), _w(te.linear_2.bias, dev))      # [1, D]  :195-199
        silu_temb = ops.ew(EW_SILU, temb)
        ar = torch.arange(L, device=dev)
        ops.ew(EW_ADD_BCAST, x, _w(denoiser.bh_embedding.weight, dev)[ar % 2].contiguous(), out=x, D=D, R1=B, s0=D, s1=0)      # :316-317
        ops.ew(EW_ADD_BCAST, x, _w(denoiser.query_pos.pe, dev)[:, 0][ar // 2].contiguous(), out=x, D=D, R1=B, s0=D, s1=0)       # SineBH
        mems, kpm = [], []
        ce = _w(denoiser.condition_embedding.weight, dev)
        for j, name in enumerate(MEM_NAMES):                                            # :223-261, 332-353
            # a fresh [S, B, D] copy (for B = 1 the permuted view is already contiguous and would alias the caller's tensor)
            m = encoder_hidden_states[j].detach().to(device=dev, dtype=torch.float32).permute(1, 0, 2).clone(memory_format=torch.contiguous_format)
            S = m.shape[0]
            if S > denoiser.mem_pos.pe.shape[0]:
                raise ValueError("memory longer than the memory PE buffer")             # position_encoding.py:135
            ops.ew(EW_ADD_BCAST, m, temb, out=m, D=D, R1=B, s0=0, s1=0)
            ops.ew(EW_ADD_BCAST, m, ce[j], out=m, D=D, R1=B, s0=0, s1=0)
            ops.ew(EW_ADD_BCAST, m, _w(denoiser.mem_pos.pe, dev)[:S, 0].contiguous(), out=m, D=D, R1=B, s0=D, s1=0)
            mems.append(m)
            mk = masks.get(name)
            kpm.append(mk.to(device=dev, dtype=torch.uint8).contiguous() if mk is not None else None)
        layers, att = [], []
        for i, layer in enumerate(denoiser.decoder.layers):                             # cross_attention.py:556-664
            sv = {"x0": x}
            t2 = ops.layer_norm(x, _w(layer.norm1.weight, dev), _w(layer.norm1.bias, dev))
            o, _, sv["self"] = _mha_fwd(ops, layer.self_attn, t2, t2, None, dev)
            x = ops.add_(o, x)
            o, sv["tb1"] = _time_block_fwd(ops, layer.time_block1, x, silu_temb, dev)
            x = ops.add_(o, x)
            sv["x2"] = x
            t2 = ops.layer_norm(x, _w(layer.norm2.weight, dev), _w(layer.norm2.bias, dev))
            cat = ops.new(L, B, 5 * D)
            sv["cross"] = []
            for j, name in enumerate(MEM_NAMES):
                nrm = getattr(layer, name + "_norm")
                m = ops.layer_norm(mems[j], _w(nrm.weight, dev), _w(nrm.bias, dev))
                o, p, s = _mha_fwd(ops, getattr(layer, "multihead_attn_" + name), t2, m, kpm[j], dev)
                cat[:, :, j * D:(j + 1) * D].copy_(o)                                   # torch.cat (:629)
                sv["cross"].append(s)
                if j == TLSN:
                    att.append(p[:, 0])
            if i == len(denoiser.decoder.layers) - 1:
                layers.append(sv)                                                       # nothing above the last cross-attention
                break                                                                   # reaches the objective
            x = ops.add_(ops.linear(cat, _w(layer.att_fuser.weight, dev), _w(layer.att_fuser.bias, dev)), x)
            sv["x3"] = x
            o, sv["tb2"] = _time_block_fwd(ops, layer.time_block2, x, silu_temb, dev)
            x = ops.add_(o, x)
            sv["x4"] = x
            t2 = ops.layer_norm(x, _w(layer.norm3.weight, dev), _w(layer.norm3.bias, dev))
            sv["ffn_pre"] = ops.linear(t2, _w(layer.linear1.weight, dev), _w(layer.linear1.bias, dev))
            x = ops.add_(ops.linear(ops.ew(EW_GELU, sv["ffn_pre"]), _w(layer.linear2.weight, dev), _w(layer.linear2.bias, dev)), x)
            layers.append(sv)
        return torch.stack(att, dim=1).contiguous(), dict(layers=layers, ops=ops, B=B, L=L, D=D)


def backward_to_sample(denoiser, saved, d_att_tlsn):
    """Gradient of a scalar that depends on the tlsn attention probabilities only (``d_att_tlsn`` [B, layers, L, S])
    with respect to ``sample`` [B, L, 128]."""
    ops, layers, D = saved["ops"], saved["layers"], saved["D"]
    dev = d_att_tlsn.device
    g = None
    with torch.cuda.device(dev):
        for i in reversed(range(len(layers))):
            layer, sv = denoiser.decoder.layers[i], layers[i]
            dcat = None
            if g is not None:
                d1 = ops.linear_bwd(g, _w(layer.linear2.weight, dev))                    # FFN
                ops.ew(EW_GELU_BWD, d1, sv["ffn_pre"], out=d1)
                ops.layer_norm_bwd(sv["x4"], _w(layer.norm3.weight, dev), ops.linear_bwd(d1, _w(layer.linear1.weight, dev)), g, accumulate=True)
                _time_block_bwd(ops, layer.time_block2, sv["tb2"], g, dev)
                dcat = ops.linear_bwd(g, _w(layer.att_fuser.weight, dev))                # [L, B, 5 D]
            dt2 = None
            for j, name in enumerate(MEM_NAMES):
                if dcat is None and j != TLSN:
                    continue
                dprob = d_att_tlsn[:, i].unsqueeze(1).contiguous() if j == TLSN else None
                dout = dcat[:, :, j * D:(j + 1) * D].contiguous() if dcat is not None else None
                d = _mha_bwd(ops, sv["cross"][j], dout, dprob)
                dt2 = d if dt2 is None else ops.add_(dt2, d)
            if g is None:
                g = torch.zeros_like(dt2)
            ops.layer_norm_bwd(sv["x2"], _w(layer.norm2.weight, dev), dt2, g, accumulate=True)
            _time_block_bwd(ops, layer.time_block1, sv["tb1"], g, dev)
            dt2 = _mha_bwd(ops, sv["self"], g, None, self_attention=True)
            ops.layer_norm_bwd(sv["x0"], _w(layer.norm1.weight, dev), dt2, g, accumulate=True)
        dlat = ops.linear_bwd(g, _w(denoiser.latent_embd.weight, dev))                   # [L, B, 128]
    return dlat.permute(1, 0, 2).contiguous()


# ----------------------------------------------------------------------------- the objective
@functools.lru_cache(maxsize=4)
def gaussian_kernel3(sigma=0.5):
    """GaussianSmoothing(channels=1, kernel_size=3, sigma=0.5, dim=2).weight with the reference's own float32 torch ops
    (gaussian_smoothing.py:28-43; a load-time table).  Returns (corner, edge, centre)."""
    kernel = 1
    grids = torch.meshgrid([torch.arange(3, dtype=torch.float32)] * 2, indexing="ij")
    for mgrid in grids:
        mean = (3 - 1) / 2
        kernel = kernel * (1 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-((mgrid - mean) / (2 * sigma)) ** 2))
    kernel = kernel / torch.sum(kernel)
    return float(kernel[0, 0]), float(kernel[0, 1]), float(kernel[1, 1])


def attention_focus_loss(denoiser, att_tlsn, focus_indices, normalize_eot=False, eot_indices=()):
    """``aggregate_attentions`` + ``get_max_attention_at_indices(smooth_attentions=True)`` +
    ``compute_attention_focus_loss`` (word_excitation_guidance.py:11-81) and the gradient with respect to the maps.
    Returns (loss 0-d tensor, losses [B], max_attention_at_indices list of lists of 0-d tensors, d_att [B, layers, L, S])."""
    dev = att_tlsn.device
    B, NL, L, S = att_tlsn.shape
    last, off, flat = _focus_tables(B, S, focus_indices, normalize_eot, eot_indices)
    W = last - 1
    nt_max = max(1, int(max(len(s) for s in focus_indices)))
    tok_off = torch.from_numpy(off).to(dev)
    tok_idx = torch.from_numpy(flat).to(dev)
    ws = torch.empty(B * (3 * L * W + 3 * nt_max), dtype=torch.float32, device=dev)
    losses = torch.empty(B, dtype=torch.float32, device=dev)
    max_att = torch.empty(max(1, int(off[-1])), dtype=torch.float32, device=dev)
    d_att = torch.empty_like(att_tlsn)
    k3 = (C.c_float * 3)(*gaussian_kernel3())
    with torch.cuda.device(dev):
        ops = _Ops(denoiser.engine(dev), dev)
        _lib.check(ops.lib.cfd_weg_focus(ops.h, C.c_void_p(att_tlsn.data_ptr()), B, NL, L, S, C.c_void_p(tok_off.data_ptr()),
                                         C.c_void_p(tok_idx.data_ptr()), last, nt_max, k3, C.c_void_p(ws.data_ptr()),
                                         C.c_void_p(losses.data_ptr()), C.c_void_p(max_att.data_ptr()), C.c_void_p(d_att.data_ptr()), ops.stream))
    lh = losses.cpu()
    loss = torch.tensor(float(np.mean(lh.numpy(), dtype=np.float32)))                     # torch.mean(losses) over the batch (:80)
    mx = [[max_att[off[b] + k] for k in range(len(focus_indices[b]))] for b in range(B)]
    return loss, losses, mx, d_att


def _focus_tables(B, S, focus_indices, normalize_eot, eot_indices):
    """(last, offsets int32 [B + 1], flat indices int32) of the text slice and the focus tokens (weg.py:19-28,40-49)."""
    if len(focus_indices) != B:
        raise ValueError("focus_indices needs one list per batch row")
    last = S - 1                                                                          # att_mat[:, :, 1:-1]
    if normalize_eot:
        assert len(eot_indices) > 0, "Need to provide eot indices for normalization"     # :24
        assert B == 1, "EOS/BOS normalization only works for test batch size 1 currently"  # :25
        last = int(eot_indices[0])
        if last < 0:
            last += S
    for s in focus_indices:
        for i in s:
            if not 1 <= int(i) <= last - 1:
                raise IndexError(f"focus index {i} is outside the text slice [1, {last})")
    off = np.cumsum([0] + [len(s) for s in focus_indices]).astype(np.int32)
    flat = np.array([int(i) for s in focus_indices for i in s] or [0], dtype=np.int32)
    return last, off, flat


def loss_and_grad(denoiser, latents, timestep, encoder_hidden_states, cond_masks, focus_indices, normalize_eot=True, eot_indices=None,
                  same_conditioning=False):
    """One evaluation of the WEG objective on the text-only chunk (convofusion.py:447-471) and d(loss)/d(latents) --
    ``cfd_weg_eval``: forward with saved activations, objective and backward sweep enqueued by the library.
    ``same_conditioning``: True = timestep, memories and masks are those of the previous call (only the latents moved): the
    memory-side LayerNorms / key-value projections and the time embeddings are reused; ``"memories"`` = memories and masks are
    the previous call's, the timestep may differ (one evaluation per loop iteration: the library keeps per-timestep tables).
    Returns (loss 0-d tensor, losses [B], max_attention_at_indices, grad [B, L, 128])."""
    if not isinstance(denoiser, Denoiser):
        raise TypeError("denoiser must be a convofusion_amd.denoiser.Denoiser")
    dev = latents.device
    if dev.type != "cuda":
        raise RuntimeError("WEG runs on an MI355X only (tensors must be on 'cuda'); no CPU fallback")
    if eot_indices is None:
        eot_indices = torch.argmax(cond_masks["tlsn"].int(), dim=1) - 1                   # :460 (index look-up)
    B, L, _ = latents.shape
    if encoder_hidden_states[0].shape[0] != B:
        raise ValueError("the conditioning tuple must have one row per latent row (the text-only chunk)")
    last, off, flat = _focus_tables(B, int(encoder_hidden_states[TLSN].shape[1]), focus_indices, normalize_eot, eot_indices)
    lat = latents.detach().to(torch.float32).contiguous()
    handle = denoiser.engine(dev, mem_len=max(int(m.shape[1]) for m in encoder_hidden_states))
    marr, keep = Denoiser.pack_memories(encoder_hidden_states, cond_masks)
    a = _lib.WegArgs()
    a.B, a.L, a.timestep, a.latents, a.mem = B, L, int(timestep), lat.data_ptr(), marr
    a.tok_off, a.tok_idx, a.last = off.ctypes.data, flat.ctypes.data, last
    a.kernel3 = (C.c_float * 3)(*gaussian_kernel3())
    a.reuse_memory_side = 2 if same_conditioning == "memories" else (1 if same_conditioning else 0)
    losses = torch.empty(B, dtype=torch.float32, device=dev)
    max_att = torch.empty(max(1, int(off[-1])), dtype=torch.float32, device=dev)
    grad = torch.empty_like(lat)
    loss = C.c_float()
    with torch.cuda.device(dev):
        _lib.check(_lib.load().cfd_weg_eval(handle, C.byref(a), C.c_void_p(losses.data_ptr()), C.c_void_p(max_att.data_ptr()),
                                            C.c_void_p(grad.data_ptr()), C.byref(loss), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    mx = [[max_att[off[b] + k] for k in range(len(focus_indices[b]))] for b in range(B)]
    return torch.tensor(loss.value), losses, mx, grad


def loss_and_grad_stepwise(denoiser, latents, timestep, encoder_hidden_states, cond_masks, focus_indices, normalize_eot=True,
                           eot_indices=None):
    """The same evaluation strung together from this module's launch wrappers (one C call per kernel): the inspectable
    form the tests use to look at intermediate tensors; ``loss_and_grad`` is the product path."""
    if eot_indices is None:
        eot_indices = torch.argmax(cond_masks["tlsn"].int(), dim=1) - 1
    att, saved = forward_saved(denoiser, latents, timestep, encoder_hidden_states, cond_masks)
    loss, losses, mx, d_att = attention_focus_loss(denoiser, att, focus_indices, normalize_eot, eot_indices)
    grad = backward_to_sample(denoiser, saved, d_att)
    return loss, losses, mx, grad


def update_latent(latents, grad, lr, denoiser):
    """``weg.update_latent``: latents - lr * d(loss)/d(latents) (word_excitation_guidance.py:54-61)."""
    dev = latents.device
    with torch.cuda.device(dev):
        ops = _Ops(denoiser.engine(dev), dev)
        return ops.ew(EW_AXPY, latents.contiguous(), grad.contiguous(), alpha=-float(lr))


def iterative_refinement_step(denoiser, latents, indices_to_alter, loss, threshold, encoder_hidden_states, cond_masks, step_size, t,
                              max_refinement_steps=400, normalize_eot=False, eot_indices=(), conditioning_seen=False):
    """``Convofusion.iterative_refinement_step`` (convofusion.py:298-388): repeat the update at one timestep until the
    objective falls below ``1 - threshold``.  ``conditioning_seen``: the caller's previous evaluation was at this timestep with
    these memories (``weg_update``).  Returns (loss, latents, max_attention_at_indices, grad at the returned latents)."""
    iteration = 0
    target_loss = max(0, 1.0 - threshold)
    same = bool(conditioning_seen)   # the loop stays at one timestep with the same memories: from the second evaluation on only the latents differ
    while loss > target_loss:
        iteration += 1
        loss, _, _, grad = loss_and_grad(denoiser, latents, t, encoder_hidden_states, cond_masks, indices_to_alter, normalize_eot, eot_indices,
                                         same_conditioning=same)
        same = True
        if loss != 0:
            latents = update_latent(latents, grad, step_size, denoiser)
        if iteration >= max_refinement_steps:
            break
    loss, _, mx, grad = loss_and_grad(denoiser, latents, t, encoder_hidden_states, cond_masks, indices_to_alter, normalize_eot, eot_indices,
                                      same_conditioning=same)
    return loss, latents, mx, grad


def scale_range_schedule(weg_parameters, num_steps, i, carry=None):
    """The WEG step-size factor table of loop iteration ``i``.

    ``unbounded_synthesis.diffusion_reverse_forecast`` resets ``scale_range = (1., 0.5)`` and takes a fresh
    ``np.linspace(lo, hi, N)`` every iteration (unbounded_synthesis.py:82-89): entry i runs from 1.0 down to 0.5.
    ``Convofusion._diffusion_reverse`` instead RE-ASSIGNS the name inside the loop and never resets it
    (convofusion.py:395,442-444): iteration i takes ``np.linspace(sr[0], sr[1], N)`` of the PREVIOUS iteration's
    table, so from iteration 1 on the interval collapses and entry i stays ~1.0 (0.9999995 at i = 1, N = 1000).
    ``carry``: a 2-element list holding (sr[0], sr[1]) of the previous iteration, updated in place -- pass one list
    through the whole loop to get the `_diffusion_reverse` behaviour; ``None`` gives the rollout's fresh table."""
    if carry is None:
        return np.linspace(weg_parameters["scale_range"][0], weg_parameters["scale_range"][1], num_steps)
    sr = np.linspace(carry[0], carry[1], num_steps)
    carry[0], carry[1] = sr[0], sr[1] if num_steps > 1 else sr[0]
    return sr


def weg_update(denoiser, latents, i, t, text_only_states, text_only_masks, focus_indices, weg_parameters, num_steps, scale_carry=None,
               same_memories=False):
    """The WEG branch of loop iteration ``i`` at timestep ``t`` (convofusion.py:437-496).  ``text_only_*``: chunk 1 of
    the 7-way guidance batch (:447-448).  ``scale_carry``: see ``scale_range_schedule`` (a list threaded through the loop
    for ``_diffusion_reverse``; None for the rollout).  ``same_memories``: the memories and masks are the ones the previous
    ``weg_update`` of this loop was given (every iteration but the first).  Returns (latents, loss)."""
    scale_range = scale_range_schedule(weg_parameters, num_steps, i, scale_carry)                                # :442-444
    eot = torch.argmax(text_only_masks["tlsn"].int(), dim=1) - 1                                                  # :460
    step_size = weg_parameters["scale_factor"] * np.sqrt(scale_range[i])
    loss, _, _, grad = loss_and_grad(denoiser, latents, t, text_only_states, text_only_masks, focus_indices, True, eot,
                                     same_conditioning="memories" if same_memories else False)
    thresholds = weg_parameters["thresholds"]
    if i in thresholds and loss > 1.0 - thresholds[i]:                                                            # :474-487
        loss, latents, _, grad = iterative_refinement_step(denoiser, latents, focus_indices, loss, thresholds[i], text_only_states,
                                                           text_only_masks, step_size, t, weg_parameters["max_refinement_steps"], True, eot,
                                                           conditioning_seen=True)
    if i < weg_parameters["max_iter_to_alter"] and loss != 0:                                                     # :490-495
        latents = update_latent(latents, grad, step_size, denoiser)
    return latents, float(loss)
