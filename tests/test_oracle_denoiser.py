"""The numpy oracle against the golden outputs of the reference Denoiser (CPU, no GPU)."""
import numpy as np
import pytest

from oracle import denoiser_ref, weights
from tests.helpers import forward_case, heavy_case, heavy_state_dict, heavy_traj_case, max_abs, rel_l2, state_dict


def test_state_dict_layout():
    ks = weights.key_shapes()
    assert len(ks) == 537
    sd = state_dict()
    assert sum(v.size for k, v in sd.items() if not k.endswith(".pe")) == 92923013  # parameter count
    # the 9 layers must differ (reference clones one layer at init)
    assert not np.array_equal(sd["decoder.layers.0.linear1.weight"], sd["decoder.layers.1.linear1.weight"])


@pytest.mark.parametrize("name", ["tiny", "tiny_sharp", "real", "oddlen"])
def test_forward_matches_reference(name):
    sd, inp, t, g = forward_case(name)
    out, att = denoiser_ref.denoiser_forward(sd, inp["sample"], t, inp["memories"], inp["masks"])
    assert out.shape == g["out"].shape
    assert rel_l2(out, g["out"]) < 2e-5          # thread-count sensitive at ~1e-6
    for j in range(5):
        ref = g[f"att{j}"]
        assert max_abs(att[j][: ref.shape[0]], ref) < 2e-4
    # masked key columns are exactly zero, rows sum to one
    for j, nm in enumerate(denoiser_ref.MEM_NAMES):
        m = inp["masks"][nm]
        if m is not None:
            assert np.all(att[j][np.broadcast_to(m[:, None, None, :], att[j].shape)] == 0)
        np.testing.assert_allclose(att[j].sum(-1), 1.0, atol=1e-5)


def test_forward_long_memory():
    sd, inp, t, g = forward_case("synth")
    out, att = denoiser_ref.denoiser_forward(sd, inp["sample"], t, inp["memories"], inp["masks"])
    assert rel_l2(out, g["out"]) < 2e-5
    assert max_abs(att[1][..., :64], g["att1_head"]) < 2e-5
    assert max_abs(att[0], g["att0"]) < 2e-5


def test_rejects_what_the_reference_rejects():
    sd, inp, t, _ = forward_case("tiny")
    with pytest.raises(ValueError):  # odd L: broadcasting error at position_encoding.py:160-161
        denoiser_ref.denoiser_forward(sd, inp["sample"][:, :15], t, inp["memories"], inp["masks"])
    mems = list(inp["memories"])
    mems[1] = np.zeros((mems[1].shape[0], 1025, 512), dtype=np.float32)
    with pytest.raises(ValueError):  # S > 1024: position_encoding.py:135
        denoiser_ref.denoiser_forward(sd, inp["sample"], t, mems, {})


def test_per_row_timesteps_equal_scalar():
    sd, inp, t, _ = forward_case("tiny")
    a, _ = denoiser_ref.denoiser_forward(sd, inp["sample"], t, inp["memories"], inp["masks"], num_layers=2)
    b, _ = denoiser_ref.denoiser_forward(sd, inp["sample"], np.full((7,), t), inp["memories"], inp["masks"], num_layers=2)
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["tiny", "tiny_sharp", "real", "oddlen"])
def test_torch_restatement_matches_reference_golden(name):
    """oracle/denoiser_torch.py (bench.py's cpu_baseline leg: the reference's own torch CPU-eager op sequence driven
    from a state dict) against the outputs of the imported reference class."""
    import torch
    from oracle import denoiser_torch
    sd, inp, t, g = forward_case(name)
    tsd = denoiser_torch.to_torch(sd)
    mems = [torch.from_numpy(m) for m in inp["memories"]]
    masks = {k: (torch.from_numpy(v) if v is not None else None) for k, v in inp["masks"].items()}
    out, att = denoiser_torch.denoiser_forward(tsd, torch.from_numpy(inp["sample"]), t, mems, masks)
    assert rel_l2(out.numpy(), g["out"]) < 2e-5
    for j in range(5):
        ref = g[f"att{j}"]
        assert max_abs(att[j].numpy()[: ref.shape[0]], ref) < 2e-4


def test_heavy_tailed_weights_against_the_reference():
    """The oracle on the heavy-tailed stress weights (outlier LayerNorm gains, outlier / near-zero FFN and in-projection rows, memories
    with outlier tokens: the stand-in for the trained checkpoint that cannot be loaded here) against the imported reference's outputs
    (tests/golden/heavy.npz), and the first 10 steps of the guided 50-step DDIM loop."""
    from oracle import philox_ref, sampler_ref, scheduler_ref
    sd = heavy_state_dict()
    for name in ("fwd_small", "fwd_tile"):
        inp, t, want, want_att = heavy_case(name)
        out, att = denoiser_ref.denoiser_forward(sd, inp["sample"], t, inp["memories"], inp["masks"])
        # (the outlier gains make the logits large, so two float32 evaluations of one softmax already differ by ~3e-3 in a probability:
        #  numpy oracle vs torch reference here; the OUTPUT budget is unchanged)
        assert rel_l2(out, want) < 1e-4 and max_abs(att[2], want_att) < 1e-2
    cb, B, L, n, seed, g = heavy_traj_case()
    sd8 = heavy_state_dict(8.0)
    _, snaps, _ = sampler_ref.diffusion_reverse(
        lambda x, t, e, mk: denoiser_ref.denoiser_forward(sd8, x, t, e, mk), scheduler_ref.DDIMSchedulerRef(), cb["memories"], cb["masks"],
        philox_ref.normal_tensor(seed, 0, range(B), 1, L), lambda i, t: philox_ref.normal_tensor(seed, i, range(B), 0, L),
        guidance_scale=7.5, num_inference_steps=n, eta=0.0, keep_steps=(1, 10), stop_after=10)
    assert rel_l2(snaps[1], g["traj_step1"]) < 1e-4 and rel_l2(snaps[10], g["traj_step10"]) < 1e-3
