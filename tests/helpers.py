"""Shared test helpers: golden loading, error metrics, seeded weights (cached)."""
import functools
import os

import numpy as np

from oracle import inputs, weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (weight seed, sharp, input seed)   (must match tests/golden/make_golden.py)
FORWARD_CASES = {
    "tiny": (1234, 1.0, 100 + 4),
    "tiny_sharp": (4321, 4.0, 100 + 10),
    "real": (1234, 1.0, 100 + 4),
    "oddlen": (4321, 4.0, 100 + 6),
}


@functools.lru_cache(maxsize=4)
def state_dict(seed=1234, sharp=1.0, mem_len=1024):
    if isinstance(sharp, str) and sharp.startswith("heavy"):     # "heavy8": the heavy-tailed stress weights at outlier factor 8 (oracle/weights.py)
        sd = weights.make_state_dict_heavy(seed=seed, gain=float(sharp[5:]))
        return weights.extend_pe(sd, mem_len) if mem_len > 1024 else sd
    sd = weights.make_state_dict(seed=seed, sharp=sharp)
    return weights.extend_pe(sd, mem_len) if mem_len > 1024 else sd


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def forward_case(name):
    """(state_dict, inputs dict, t, golden npz) for a denoiser_<name> fixture."""
    g = load_golden("denoiser_" + name)
    meta = g["meta"]
    Be, L = int(meta[0]), int(meta[1])
    S = tuple(int(x) for x in meta[2:7])
    pad = tuple(int(x) for x in meta[7:12])
    t = int(meta[12])
    if name == "synth":
        sd, iseed = state_dict(1234, 1.0, 1536), 77
    else:
        wseed, sharp, iseed = FORWARD_CASES[name]
        sd = state_dict(wseed, sharp)
    inp = inputs.make_plain_batch(seed=iseed, Be=Be, L=L, S=S, pad_tail=pad, scale=float(g["scale"]))
    return sd, inp, t, g


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def max_abs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


@functools.lru_cache(maxsize=2)
def heavy_state_dict(gain=20.0):
    """gain 20: the single forwards; gain 8: the guided 50-step loop (at 20 it is chaotic: oracle/weights.py)."""
    return weights.make_state_dict_heavy(seed=777, gain=gain)


def heavy_case(name):
    """(inputs dict, t, expected output, expected listener-text maps) of tests/golden/heavy.npz (made by make_golden_heavy.py from the
    imported reference on oracle.weights.make_state_dict_heavy)."""
    g = load_golden("heavy")
    meta = [int(x) for x in g[name + "_meta"]]
    Be, L, S, pad, t = meta[0], meta[1], tuple(meta[2:7]), tuple(meta[7:12]), meta[12]
    inp = inputs.make_outlier_batch(seed=50 + len(name), Be=Be, L=L, S=S, pad_tail=pad)
    return inp, t, g[name], g[name + "_att2"]


def heavy_traj_case(key="traj"):
    """(guidance batch with outlier tokens, B, L, steps, seed, golden npz) of a trajectory in heavy.npz: "traj" = the 50-step DDIM loop,
    "ddpm1000" = the full-length DDPM loop."""
    g = load_golden("heavy")
    meta = [int(x) for x in g[key + "_meta"]]
    B, L, S, pad, n, seed = meta[0], meta[1], tuple(meta[2:7]), tuple(meta[7:12]), meta[12], meta[13]
    cb = inputs.make_cfg_batch(seed=seed, B=B, L=L, S=S, pad_tail=pad)
    cb["memories"] = [inputs.add_outlier_tokens(u, seed + j)[rm] for j, (u, rm) in enumerate(zip(cb["unique"], cb["row_map"]))]
    return cb, B, L, n, seed, g
