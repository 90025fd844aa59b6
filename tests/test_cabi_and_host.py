"""CPU (-m "not gpu"): the C-ABI library builds, loads and exports every symbol include/cfdenoise.h
declares; host-side mirrors (state-dict layout, scheduler tables, config validation, sharding over gloo)."""
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.helpers import load_golden, state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from convofusion_amd import _lib, build
    build.build()
    lib = _lib.load()
    declared = set()
    for h in ("cfdenoise.h", "cfdenoise_dev.h"):      # the drop-in boundary, and the developer / test hooks beside it
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        found = set(re.findall(r"\b(cfd_[a-z_0-9]+)\s*\(", hdr))
        assert found, "no declarations parsed in " + h
        declared |= found
    public = open(os.path.join(ROOT, "include", "cfdenoise.h")).read()
    assert not re.search(r"cfd_(debug|test|bench)_", re.sub(r"/\*.*?\*/", "", public, flags=re.S)), "developer hooks belong in cfdenoise_dev.h"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header():
    import ctypes as C
    from convofusion_amd import _lib
    # natural C layout of the header's structs on x86-64
    assert C.sizeof(_lib.Config) == 24
    assert C.sizeof(_lib.Memory) == 32
    assert _lib.SampleArgs.mem.offset % 8 == 0 and C.sizeof(_lib.SampleArgs) == _lib.SampleArgs.mem.offset + 5 * 32 + 8 + 16 + 5 * 8 + 8
    assert _lib.SampleArgs.timesteps.offset == _lib.SampleArgs.mem.offset + 5 * 32 + 8   # (8-aligned pointer behind the two ints)
    assert _lib.SampleArgs.att_ring.offset == _lib.SampleArgs.timesteps.offset + 16      # (pointer, int, padding; then the five ring pointers)
    assert _lib.SampleArgs.operand_policy.offset == _lib.SampleArgs.att_ring.offset + 5 * 8   # (int + tail padding to the struct's 8-byte alignment)


def test_weg_host_logic():
    """Struct layout of cfd_weg_args / cfd_mat, the focus-token tables, the smoothing kernel and the CPU refusal."""
    import ctypes as C
    from convofusion_amd import _lib, weg
    from oracle import weg_ref
    assert C.sizeof(_lib.Mat) == 40
    assert _lib.WegArgs.mem.offset == 24 and _lib.WegArgs.tok_off.offset == 24 + 5 * 32 and C.sizeof(_lib.WegArgs) == 24 + 160 + 16 + 4 + 12 + 4 + 4
    last, off, flat = weg._focus_tables(2, 12, [[2, 5], []], False, ())
    assert last == 11 and off.tolist() == [0, 2, 2] and flat.tolist() == [2, 5]                      # att[:, :, 1:-1]
    last, off, flat = weg._focus_tables(1, 12, [[3]], True, torch.tensor([8]))
    assert last == 8
    with pytest.raises(AssertionError):      # "EOS/BOS normalization only works for test batch size 1 currently" (weg.py:25)
        weg._focus_tables(2, 12, [[2], [3]], True, torch.tensor([8, 8]))
    with pytest.raises(IndexError):          # the reference would index outside the sliced map
        weg._focus_tables(1, 12, [[8]], True, torch.tensor([8]))
    with pytest.raises(ValueError):
        weg._focus_tables(2, 12, [[2]], False, ())
    k = weg_ref.gaussian_kernel()
    np.testing.assert_allclose(weg.gaussian_kernel3(), (k[0, 0], k[0, 1], k[1, 1]), rtol=1e-6)
    from tests.gpu_helpers import ABL, DENOISER_KW
    from convofusion_amd.denoiser import Denoiser
    m = Denoiser(ablation=ABL, **DENOISER_KW)
    enc = [torch.zeros(1, s, 512) for s in (4, 6, 12, 8, 1)]
    with pytest.raises(RuntimeError):        # the product never computes on the CPU
        weg.loss_and_grad(m, torch.zeros(1, 16, 128), 5, enc, {"tlsn": torch.zeros(1, 12, dtype=torch.bool)}, [[2]], True, torch.tensor([8]))


def test_denoiser_mirror_state_dict_and_validation():
    from convofusion_amd.denoiser import Denoiser, sine_pe, sinusoid_table
    from oracle import denoiser_ref, weights
    from tests.gpu_helpers import ABL, DENOISER_KW
    m = Denoiser(ablation=ABL, **DENOISER_KW)
    ks = weights.key_shapes()
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == ks          # reference layout, 537 entries
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state_dict().items()}, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 92923013
    np.testing.assert_allclose(sine_pe(1024).numpy(), weights.sine_pe(1024), atol=2e-4)  # torch vs numpy sin/cos/exp ulps
    np.testing.assert_allclose(sinusoid_table(1000).numpy(), denoiser_ref.timestep_embedding(np.arange(1000)), atol=2e-4)
    np.testing.assert_array_equal(sinusoid_table(1000)[0].numpy(), denoiser_ref.timestep_embedding(np.arange(1))[0])
    with pytest.raises(TypeError):
        Denoiser(ablation=ABL, **{**DENOISER_KW, "condition": "action"})
    with pytest.raises(ValueError):
        Denoiser(ablation=ABL, **{**DENOISER_KW, "arch": "trans_enc"})
    with pytest.raises(ValueError):
        Denoiser(ablation=SimpleNamespace(SKIP_CONNECT=True, VAE_TYPE="convofusion", DIFF_PE_TYPE="mld", CAUSAL_ATTN=False), **DENOISER_KW)
    with pytest.raises(RuntimeError):   # the product never computes on the CPU
        m.engine(torch.device("cpu"))


def test_scheduler_mirror_tables_and_timesteps():
    from convofusion_amd import scheduler
    from tests.gpu_helpers import SCHED_KW
    g = load_golden("scheduler_tables")
    s = scheduler.DDPMScheduler(variance_type="fixed_small", **SCHED_KW)
    np.testing.assert_array_equal(s.betas.numpy(), g["betas"])
    np.testing.assert_array_equal(s.alphas_cumprod.numpy(), g["alphas_cumprod"])
    assert s.init_noise_sigma == 1.0 and len(s) == 1000
    s.set_timesteps(1000)
    assert int(s.timesteps[0]) == 999 and int(s.timesteps[-1]) == 0
    s.set_timesteps(5000)            # clamped to the training schedule (diffusers 0.14.0: min(T, N))
    assert s.num_inference_steps == 1000 and len(s.timesteps) == 1000
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    from oracle import scheduler_ref
    for bad in (30, 300):            # T % N != 0: the diffusers releases disagree on the table, nothing here can pin it: refused by default
        with pytest.raises(ValueError):
            s.set_timesteps(bad)
        with pytest.raises(ValueError):
            scheduler_ref.DDPMSchedulerRef().set_timesteps(bad)
    # ... and opt-in: the 0.14.0 table arange(0, T, T // N)[::-1] (MORE than N entries), the same in the mirror and the oracle
    u = scheduler.DDPMScheduler(variance_type="fixed_small", allow_unpinned_timesteps=True, **SCHED_KW)
    o = scheduler_ref.DDPMSchedulerRef(allow_unpinned_timesteps=True)
    for n, count in ((300, 334), (30, 31), (999, 1000), (7, 8)):
        u.set_timesteps(n)
        o.set_timesteps(n)
        assert u.num_inference_steps == n and u.timesteps.tolist() == o.timesteps.tolist() == list(range(0, 1000, 1000 // n))[::-1]
        assert len(u.timesteps) == count
    u.set_timesteps(50)              # a dividing count is the pinned table whatever the switch
    assert u.timesteps.tolist() == list(range(980, -1, -20))
    assert s.timestep_table(50)[1].tolist() == list(range(980, -1, -20)) and s.num_inference_steps == 50   # (no state change)
    d = scheduler.DDIMScheduler(steps_offset=1, **SCHED_KW)
    d.set_timesteps(50)
    assert int(d.timesteps[0]) == 981
    with pytest.raises(NotImplementedError):
        scheduler.DDPMScheduler(variance_type="learned", **SCHED_KW)
    with pytest.raises(RuntimeError):
        s.step(torch.zeros(1, 2, 128), 5, torch.zeros(1, 2, 128))  # CPU tensors: no fallback


def test_dedup_memories_groups_exactly():
    """The device-agnostic grouping logic of the sampler's de-duplication (hash groups + bitwise verification) against the
    known structure of the guidance batch and against the pairwise exact path."""
    from convofusion_amd import sampler
    from oracle import inputs
    for seed, B in [(1, 1), (2, 3), (3, 8)]:
        cb = inputs.make_cfg_batch(seed=seed, B=B, L=16, S=(6, 20, 6, 8, 1), pad_tail=(2, 0, 1, 0, 0))
        enc = [torch.from_numpy(x) for x in cb["memories"]]
        masks = {k: (torch.from_numpy(v) if v is not None else None) for k, v in cb["masks"].items()}
        u, maps, um = sampler.dedup_memories(enc, masks)
        for j, name in enumerate(inputs.MEM_NAMES):
            assert torch.equal(maps[j], torch.from_numpy(cb["row_map"][j])) and torch.equal(u[j], torch.from_numpy(cb["unique"][j]))
            mk = masks[name].to(torch.uint8) if masks[name] is not None else None
            assert torch.equal(sampler._dedup_rows_exact(enc[j], mk)[1], maps[j])
    u, maps, _ = sampler.dedup_memories([torch.randn(5, 3, 512) for _ in range(5)], {})          # nothing shared
    assert all(m.tolist() == [0, 1, 2, 3, 4] for m in maps)
    e = torch.randn(1, 4, 512).expand(3, 4, 512).contiguous()                                     # equal data, different masks
    mk = torch.tensor([[0, 0, 0, 1], [0, 0, 1, 1], [0, 0, 0, 1]], dtype=torch.bool)
    _, maps, um = sampler.dedup_memories([e] * 5, {"tlsn": mk})
    assert maps[2].tolist() == [0, 1, 0] and maps[0].tolist() == [0, 0, 0] and um["tlsn"].shape[0] == 2


def test_shard_helpers():
    from convofusion_amd.distributed import shard_cfg_batch, shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    t = torch.arange(7 * 5).reshape(7 * 5, 1)
    s = shard_cfg_batch(t, 1, 3, 5)
    assert s.reshape(7, 2).tolist() == [[5 * c + 1, 5 * c + 2] for c in range(7)]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from convofusion_amd.distributed import sample_sharded
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    total, L = 5, 4
    enc = [torch.arange(7 * total, dtype=torch.float32).reshape(7 * total, 1, 1).expand(7 * total, 2, 512).contiguous() for _ in range(5)]

    def fake_sample(e, masks, B, first_utterance):
        # latents are a function of the global utterance id and of the full-conditioning chunk's memory
        assert e[0].shape[0] == 7 * B
        ids = torch.arange(first_utterance, first_utterance + B, dtype=torch.float32)
        return (ids[:, None, None] + e[0].reshape(7, B, 2, 512)[6, :, 0, 0][:, None, None] * 100).expand(B, L, 128).contiguous()

    out = sample_sharded(fake_sample, enc, {"tlsn": None}, total)
    q.put((rank, out[:, 0, 0].tolist()))
    dist.destroy_process_group()


def test_sharded_sampling_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [float(u + (6 * 5 + u) * 100) for u in range(5)]
    assert res[0] == want and res[1] == want


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_flag_launches_that_many_ranks(n):
    """`python bench.py --gpus N` must start N ranks by itself (round 1: the flag was parsed and ignored).  The launch
    path -- spawn, rendezvous on 127.0.0.1, the collate all_gather, one JSON line from rank 0 -- runs here over gloo
    with the measurement itself switched off (`--selftest-cpu`), at 2 ranks and at the 8 of BASELINE configs[2] (one node of
    8 MI355X; the pool has no such node, so this is the only place the 8-rank launch runs): every rank is seen by a real
    collective, a RAGGED batch of 3 N + 1 utterances sharded over the ranks reproduces the single-rank rows, and there is
    exactly one JSON line.  A mismatching launcher environment is an error."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"     # (8 ranks on this container's 8 cores)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--selftest-cpu", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["selftest"] is True and d["value"] is None and d["shards_reproduce_single_rank"] is True
    if n != 2:
        return
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--selftest-cpu"], capture_output=True, text=True,
                         timeout=120, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)


def test_stale_library_is_detected_by_content():
    """The library carries a hash of the sources it was built from; the binding compares it with the sources on disk before
    mapping the file (a library left over from other sources would mis-read struct arguments instead of failing)."""
    import ctypes as C
    from convofusion_amd import _lib, build
    lib = _lib.load()
    lib.cfd_source_hash.restype = C.c_char_p
    assert lib.cfd_source_hash().decode() == build.source_hash()
    blob = open(build.LIB, "rb").read()
    k = blob.find(b"cfd-src-hash:")
    assert k >= 0 and blob[k + 13:k + 29].decode() == build.source_hash()
    assert len(build.source_hash()) == 16 and set(build.DEPS) >= {"cfd_core.hip", "cfd_forward.hip", "cfd_internal.hpp", "xattn_fused.hpp", os.path.join("..", "..", "include", "cfdenoise.h")}


def test_no_edit_installer_binds_the_reference_entry_points():
    """convofusion_amd.install(model) / patch_rollout(module): the reference's loop entry points are replaced by binding,
    not by editing its sources; signatures equal the reference's (convofusion.py:391, unbounded_synthesis.py:28)."""
    import inspect
    import types
    import convofusion_amd
    from convofusion_amd import installer as inst
    from convofusion_amd.denoiser import Denoiser
    from convofusion_amd import scheduler
    from tests.gpu_helpers import ABL, DENOISER_KW, SCHED_KW
    assert str(inspect.signature(inst._diffusion_reverse)) == "(self, encoder_hidden_states, lengths=None, cond_masks={}, focus_indices=[])"
    assert str(inspect.signature(inst.diffusion_reverse_forecast)) == \
        "(model, encoder_hidden_states, lengths=None, preseq=None, cond_masks={}, focus_indices=[])"

    class RefLike:                      # stands in for the reference Convofusion module: the class method must stay untouched
        def _diffusion_reverse(self, encoder_hidden_states, lengths=None, cond_masks=dict(), focus_indices=[]):
            return "reference loop"
    model = RefLike()
    with pytest.raises(TypeError):      # yaml still points at the reference denoiser
        model.denoiser, model.scheduler = object(), object()
        convofusion_amd.install(model)
    model.denoiser = Denoiser(ablation=ABL, **DENOISER_KW)
    model.scheduler = scheduler.DDPMScheduler(variance_type="fixed_small", **SCHED_KW)
    assert convofusion_amd.install(model) is model
    assert isinstance(model._diffusion_reverse, types.MethodType) and model._diffusion_reverse.__func__ is inst._diffusion_reverse
    assert RefLike()._diffusion_reverse(None) == "reference loop"           # other instances and the class are untouched
    assert model._cfd_attention_steps == "auto"      # "all" while the ring of maps fits the budget, "last" beyond it
    convofusion_amd.install(model, attention_steps="all")                  # the reference's per-iteration attention dict (base.py:252-259)
    assert model._cfd_attention_steps == "all"
    with pytest.raises(ValueError):
        convofusion_amd.install(model, attention_steps="some")
    convofusion_amd.uninstall(model)
    assert model._diffusion_reverse(None) == "reference loop" and "_cfd_attention_steps" not in vars(model)
    script = types.ModuleType("unbounded_synthesis")
    with pytest.raises(AttributeError):
        convofusion_amd.patch_rollout(script)
    script.diffusion_reverse_forecast = lambda *a, **k: "reference rollout"
    orig = convofusion_amd.patch_rollout(script)
    assert script.diffusion_reverse_forecast is inst.diffusion_reverse_forecast and orig() == "reference rollout"


def test_no_kernel_of_the_library_uses_scratch_memory():
    """Every kernel keeps its state in registers / LDS: private (scratch) memory is slow, and it was the first suspect when two
    captured graphs replaying side by side (tools/experiments/concurrent_runs.py) gave wrong latents in an experiment (the spills
    turned out not to be the cause, but they are gone -- 68 bytes per lane of wave-uniform pointers in xattn_fused_kernel -- and this
    keeps it that way).  hipcc's resource-usage remarks are the check (cross-compiles without a GPU)."""
    import re
    import subprocess
    import tempfile
    from convofusion_amd import build
    from concurrent.futures import ThreadPoolExecutor

    def remarks(unit):
        with tempfile.TemporaryDirectory() as tmp:
            r = subprocess.run(["/opt/rocm/bin/hipcc", *build.FLAGS, "-c", "-Rpass-analysis=kernel-resource-usage", os.path.join(build.CSRC, unit),
                                "-o", os.path.join(tmp, "unit.o")], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stderr
    with ThreadPoolExecutor(max_workers=4) as ex:      # every translation unit of the library
        err = "\n".join(ex.map(remarks, build.SOURCES))
    names = re.findall(r"Function Name: (\S+)", err)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", err)]
    assert len(names) == len(scratch) and len(names) > 40
    bad = {n: s for n, s in zip(names, scratch) if s != 0}
    assert not bad, bad
