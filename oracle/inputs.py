"""Seeded synthetic inputs shared by the golden generator, the tests and bench.py's CPU leg.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Builds the 7-way modality-guidance batch the
reference constructs at convofusion/models/modeltype/convofusion.py:909-929 and consumes at
:527-541: chunk order
    [all_drop, text_only, audio_only, spk_only, apb_only, lsnid_only, full]
and per chunk each memory is either the utterance's conditional tensor or ONE shared
"unconditional" tensor:
    spk  : [u, u, u, S, u, u, S]      alsn : [u, u, A, u, u, u, A]
    tlsn : [u, T, u, u, u, u, T]      apb  : [u, u, u, u, P, u, P]
    lsn  : [u, u, u, u, u, I, I]
Memory tuple order is the denoiser's: (spk, alsn, tlsn, apb, lsnemb) (denoiser.py:220).
"""
import numpy as np

F32 = np.float32
MEM_NAMES = ("spkemb", "alsn", "tlsn", "apb", "lsnemb")
# which CFG chunks carry the conditional version of memory j
COND_CHUNKS = {0: (3, 6), 1: (2, 6), 2: (1, 6), 3: (4, 6), 4: (5, 6)}


def make_cfg_batch(seed, B, L, S, pad_tail=(0, 0, 0, 0, 0), uncond_pad_tail=None, scale=1.0):
    """Returns dict(sample_init [B,L,128], memories 5x[7B,S_j,512], masks name->bool[7B,S_j]|None,
    unique 5x[(B+1),S_j,512] (row 0 = uncond), row_map 5x int32[7B]).

    pad_tail[j]: number of trailing key positions masked (True) in the conditional rows of memory
    j; uncond_pad_tail[j] likewise for the unconditional row (defaults to pad_tail[j] + 1 when the
    memory is masked, so conditional and unconditional rows carry different masks like the
    reference's '-'*10 dummy text does).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if uncond_pad_tail is None:
        uncond_pad_tail = tuple((p + 1 if p and p + 1 < S[j] else p) for j, p in enumerate(pad_tail))
    unique, mems, maps, masks = [], [], [], {}
    for j in range(5):
        cond = (scale * rng.standard_normal((B, S[j], 512), dtype=F32)).astype(F32)
        unc = (scale * rng.standard_normal((1, S[j], 512), dtype=F32)).astype(F32)
        uq = np.concatenate([unc, cond], axis=0)
        rm = np.zeros((7, B), dtype=np.int32)
        for c in COND_CHUNKS[j]:
            rm[c] = 1 + np.arange(B)
        rm = rm.reshape(-1)
        unique.append(uq)
        maps.append(rm)
        mems.append(uq[rm])
        if pad_tail[j] or uncond_pad_tail[j]:
            um = np.zeros((B + 1, S[j]), dtype=bool)
            if uncond_pad_tail[j]:
                um[0, S[j] - uncond_pad_tail[j]:] = True
            for b in range(B):  # ragged: utterance b masks pad_tail[j] + (b % 3) keys (kept < S)
                n = min(pad_tail[j] + (b % 3), S[j] - 1)
                if n:
                    um[1 + b, S[j] - n:] = True
            masks[MEM_NAMES[j]] = um[rm]
        else:
            masks[MEM_NAMES[j]] = None
    init = rng.standard_normal((B, L, 128), dtype=F32)
    return dict(init=init, memories=mems, masks=masks, unique=unique, row_map=maps)


def make_plain_batch(seed, Be, L, S, pad_tail=(0, 0, 0, 0, 0), scale=1.0):
    """Independent random rows (no CFG structure): sample [Be,L,128], memories, masks."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sample = rng.standard_normal((Be, L, 128), dtype=F32)
    mems = [(scale * rng.standard_normal((Be, S[j], 512), dtype=F32)).astype(F32) for j in range(5)]
    masks = {}
    for j in range(5):
        if pad_tail[j]:
            m = np.zeros((Be, S[j]), dtype=bool)
            for b in range(Be):
                n = min(pad_tail[j] + (b % 4), S[j] - 1)
                m[b, S[j] - n:] = True
            masks[MEM_NAMES[j]] = m
        else:
            masks[MEM_NAMES[j]] = None
    return dict(sample=sample, memories=mems, masks=masks)


def add_outlier_tokens(mem, seed, frac=0.03, factor=30.0):
    """A copy of ``mem`` [rows, S, 512] in which ``frac`` of the tokens (at least one per tensor) are ``factor`` times larger and a
    handful of single features are 100 times larger: what encoder outputs look like next to Gaussian test memories."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = mem.copy()
    rows, S, D = out.shape
    n = max(1, int(frac * rows * S))
    r, s = rng.integers(0, rows, n), rng.integers(0, S, n)
    out[r, s] *= F32(factor)
    k = max(1, n // 2)
    out[rng.integers(0, rows, k), rng.integers(0, S, k), rng.integers(0, D, k)] *= F32(100.0)
    return out


def make_outlier_batch(seed, Be, L, S, pad_tail=(0, 0, 0, 0, 0)):
    """``make_plain_batch`` with outlier tokens / features in every memory and a few large latent entries."""
    inp = make_plain_batch(seed, Be, L, S, pad_tail)
    inp["memories"] = [add_outlier_tokens(m, seed + 10 + j) for j, m in enumerate(inp["memories"])]
    rng = np.random.Generator(np.random.PCG64(seed + 99))
    k = max(1, Be * L // 50)
    inp["sample"][rng.integers(0, Be, k), rng.integers(0, L, k)] *= F32(8.0)
    return inp
