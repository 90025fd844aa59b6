"""-m gpu: kernel-level parity of the HIP building blocks through the C ABI (MI355X only)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import torch
    from convofusion_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _lib.load(), _lib.create_handle(0)


def _gemm(ops, X, Y, cfg):
    import torch
    from convofusion_amd import _lib
    lib, h = ops
    x, y = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    out = torch.full((Y.shape[0], X.shape[0]), float("nan"), dtype=torch.float32, device="cuda")
    _lib.check(lib.cfd_test_gemm(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(out.data_ptr()),
                                 X.shape[0], Y.shape[0], X.shape[1], cfg, None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("cfg", [1, 3, 6, 19, 20, 30])
def test_mfma_layout_identity(ops, cfg):
    """A = I against an ASYMMETRIC B: catches swapped row/col maps, k-permutations and swizzle bugs."""
    I, J = 128, 256
    K = 128
    X = np.zeros((I, K), np.float32)
    X[np.arange(I), np.arange(I) % K] = 1.0
    Y = (np.arange(J)[:, None] * 50 + np.arange(K)[None, :] + 1).astype(np.float32)  # <= 14 bits: exact in hi+lo
    want = Y.astype(np.float64) @ X.astype(np.float64).T
    got = _gemm(ops, X, Y, cfg)
    np.testing.assert_array_equal(got, want.astype(np.float32))


@pytest.mark.parametrize("cfg", [0, 1, 3, 6, 19, 20, 30])
@pytest.mark.parametrize("shape", [(512, 300, 512), (1536, 77, 128), (36, 16, 1504), (128, 1000, 32), (4640, 130, 512)])
def test_gemm_split_bf16_accuracy(ops, cfg, shape):
    """D = Y X^T through the 3-MFMA split path; ragged I/J edges; fp32-class accuracy."""
    I, J, K = shape
    rng = np.random.default_rng(I * 7 + J)
    X = rng.standard_normal((I, K)).astype(np.float32)
    Y = (rng.standard_normal((J, K)) * rng.uniform(0.01, 30, (J, 1))).astype(np.float32)
    want = Y.astype(np.float64) @ X.astype(np.float64).T
    got = _gemm(ops, X, Y, cfg)
    assert np.isfinite(got).all()
    scale = np.sqrt((Y.astype(np.float64) ** 2).sum(1))[:, None] * np.sqrt((X.astype(np.float64) ** 2).sum(1))[None, :]
    err = np.abs(got - want) / scale
    assert err.max() < 1e-5, (err.max(), np.unravel_index(err.argmax(), err.shape))
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 1e-5, rel


def test_philox_matches_oracle(ops):
    import torch
    from convofusion_amd import _lib
    from oracle import philox_ref
    lib, h = ops
    B, per = 3, 16 * 128
    out = torch.empty((B, per), dtype=torch.float32, device="cuda")
    for seed, step, utt0, stream in [(2024, 0, 0, 1), (0xDEADBEEFCAFE, 999, 31, 0)]:
        _lib.check(lib.cfd_philox_normal(h, C.c_void_p(out.data_ptr()), B, per, seed, step, utt0, stream, None))
        torch.cuda.synchronize()
        want = np.stack([philox_ref.normal_block(seed, step, utt0 + b, stream, per) for b in range(B)])
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-5)


@pytest.mark.parametrize("kind", ["ddpm", "ddim"])
def test_scheduler_step_and_add_noise(kind):
    import torch
    from convofusion_amd import scheduler
    from oracle import scheduler_ref
    from tests.gpu_helpers import SCHED_KW
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 16, 128)).astype(np.float32)
    eps = rng.standard_normal(x.shape).astype(np.float32)
    z = rng.standard_normal(x.shape).astype(np.float32)
    if kind == "ddpm":
        s, r = scheduler.DDPMScheduler(variance_type="fixed_small", **SCHED_KW), scheduler_ref.DDPMSchedulerRef()
    else:
        s, r = scheduler.DDIMScheduler(**SCHED_KW), scheduler_ref.DDIMSchedulerRef()
    for n in (1000, 50):
        s.set_timesteps(n)
        r.set_timesteps(n)
        np.testing.assert_array_equal(s.timesteps.numpy(), r.timesteps)
        for t in (int(r.timesteps[0]), int(r.timesteps[n // 2]), int(r.timesteps[-1])):
            if kind == "ddpm":
                out = s.step(torch.from_numpy(eps).cuda(), t, torch.from_numpy(x).cuda(), variance_noise=torch.from_numpy(z).cuda())
                want = r.step(eps, t, x, noise=z)
            else:
                out = s.step(torch.from_numpy(eps).cuda(), t, torch.from_numpy(x).cuda(), eta=0.0)
                want = r.step(eps, t, x)
            np.testing.assert_allclose(out.prev_sample.cpu().numpy(), want, rtol=2e-5, atol=2e-5)
            # the x0 estimate the step forms on the way (the reference's training path reads it: convofusion.py:619)
            np.testing.assert_allclose(out.pred_original_sample.cpu().numpy(), r.pred_original_sample, rtol=2e-5, atol=2e-5)
            assert float(out.pred_original_sample.abs().max()) <= 1.0      # clip_sample
    got = s.add_noise(torch.from_numpy(x).cuda(), torch.from_numpy(z).cuda(), torch.tensor([321]))
    np.testing.assert_allclose(got.cpu().numpy(), r.add_noise(x, z, np.array([321])), rtol=1e-6, atol=1e-6)


def test_mfma_subnormal_operands(ops):
    """Do the MFMA inputs keep fp16 subnormals?  (Decides whether small split-pair 'lo' halves survive.)"""
    I = J = 16
    K = 32
    X = np.zeros((I, K), np.float32)
    X[np.arange(I), np.arange(I)] = 1.0
    Y = np.full((J, K), 3e-6, np.float32)
    got = _gemm(ops, X, Y, 19)
    print("subnormal probe: D =", got[0, 0], "(input 3e-6; fp16 subnormal spacing 5.96e-8)")
    assert abs(got[0, 0] - 3e-6) < 1e-7
