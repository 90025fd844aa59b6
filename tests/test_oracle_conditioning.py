"""CPU tests: the conditioning oracle against the golden outputs of the imported reference classes
(tests/golden/conditioning.npz, made by tests/golden/make_golden_conditioning.py), and the host-side mirrors."""
import os

import numpy as np
import pytest

from oracle import conditioning_ref, dyadic_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "conditioning.npz"))
ENC = {k[4:]: G[k] for k in G.files if k.startswith("enc.")}
FUS = {k[4:]: G[k] for k in G.files if k.startswith("fus.")}


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))


def test_audio_conv_encoder_matches_reference():
    got = conditioning_ref.audio_conv_encoder(ENC, G["mel"])
    assert got.shape == G["audio_out"].shape and rel(got, G["audio_out"]) < 1e-6


def test_latent_proj_matches_reference():
    got = conditioning_ref.latent_proj(FUS, G["lat"])
    assert rel(got, G["proj_out"]) < 1e-6 and float(np.abs(got - G["proj_out"]).max()) < 1e-5


def test_fuser_lookups_match_reference():
    spk = np.zeros((4, 5, 512), np.float32)
    s, a, t, apb, lsn = conditioning_ref.fuser_forward(FUS, spk, spk, spk, G["bits"], G["lsn_id"])
    assert s is spk and np.array_equal(apb, G["apb"]) and np.array_equal(lsn, G["lsnemb"])


def test_mirrors_keep_reference_state_dict_and_refuse_cpu():
    import torch
    from convofusion_amd.conditioning import AudioConvEncoder, default_fuser
    enc = AudioConvEncoder(input_size=80, hidden_size=256, latent_dim=512, max_seq_len=128, fps=25, sample_rate=16000, hop_length=160)
    assert list(enc.state_dict().keys()) == list(ENC.keys())
    assert all(tuple(v.shape) == ENC[k].shape for k, v in enc.state_dict().items())
    assert enc.audio_max_length == int((128 / 25) * 16000 // 160 + 1)           # audioenc.py:27
    fus = default_fuser()
    assert list(fus.state_dict().keys()) == list(FUS.keys())
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in ENC.items()}, strict=True)
    with pytest.raises(RuntimeError):   # no CPU fallback
        enc.eval()(torch.from_numpy(G["mel"]))
    with pytest.raises(NotImplementedError):
        enc.train()(torch.from_numpy(G["mel"]))


def test_dyadic_guidance_batch_structure():
    """Chunks 3 and 6 carry the conditional speaker memory, 2 and 6 the audio, ... (convofusion.py:909-929)."""
    B = 2
    cond = [np.full((B, 3, 512), 10 + j, np.float32) for j in range(5)]
    unc = [np.full((1, 3, 512), -1 - j, np.float32) for j in range(5)]
    mems = dyadic_ref.guidance_batch(cond, unc)
    from oracle.inputs import COND_CHUNKS
    for j in range(5):
        m = mems[j].reshape(7, B, 3, 512)
        for c in range(7):
            assert float(m[c].mean()) == (10 + j if c in COND_CHUNKS[j] else -1 - j)
