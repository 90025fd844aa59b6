"""`python -m convofusion_amd.run <script> ...`: the reference's own scripts with no yaml edit and no source edit.

The reference package cannot travel to the GPU box and most of it does not import here (pytorch_lightning, omegaconf ... are absent),
so the three bindings are proven on a STAND-IN tree written by this test: a `convofusion` package with the same dotted paths and the
same plug mechanism (a dotted `target` string resolved through `importlib.import_module` + `getattr`, as convofusion/config.py:16-31
does), a `get_model` that builds the model from such targets (convofusion/models/get_model.py:4-16 -> modeltype/convofusion.py:100-106),
and two scripts shaped like test.py (:39,151-152) and unbounded_synthesis.py (:28,438,520,578-579).  No GPU: nothing is computed."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TREE = {
    "convofusion/__init__.py": "",
    "convofusion/config.py": """
        import importlib
        def get_obj_from_str(string):
            module, cls = string.rsplit(".", 1)
            return getattr(importlib.import_module(module, package=None), cls)
        def instantiate_from_config(config):
            return get_obj_from_str(config["target"])(**config.get("params", dict()))
        """,
    "convofusion/models/__init__.py": "",
    "convofusion/models/architectures/__init__.py": "",
    "convofusion/models/architectures/denoiser.py": """
        class Denoiser:
            ORIGIN = "stand-in reference denoiser"
            def __init__(self, **kw):
                pass
        """,
    "convofusion/models/get_model.py": """
        from types import SimpleNamespace
        from convofusion.config import instantiate_from_config
        DENOISER = dict(target="convofusion.models.architectures.denoiser.Denoiser", params=dict(
            ablation=SimpleNamespace(SKIP_CONNECT=True, VAE_TYPE="convofusion", DIFF_PE_TYPE="convofusion", CAUSAL_ATTN=False),
            nfeats=189, condition="text+audio", latent_dim=[1, 128], ff_size=1024, num_layers=2, num_heads=4, dropout=0.1,
            normalize_before=True, activation="gelu", flip_sin_to_cos=True, position_embedding="sine", arch="trans_dec", freq_shift=0,
            text_encoded_dim=512, audio_encoded_dim=512))
        SCHEDULER = dict(target="diffusers.DDPMScheduler", params=dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                         beta_schedule="scaled_linear", variance_type="fixed_small", clip_sample=True))
        class Convofusion:
            def __init__(self):
                self.denoiser = instantiate_from_config(DENOISER)
                self.scheduler = instantiate_from_config(SCHEDULER)
                self.noise_scheduler = instantiate_from_config(SCHEDULER)
            def _diffusion_reverse(self, encoder_hidden_states, lengths=None, cond_masks=dict(), focus_indices=[]):
                return "reference loop"
        def get_model(cfg, datamodule, phase="train"):
            return Convofusion()
        """,
    "test.py": """
        import json, sys
        from convofusion.models.get_model import get_model
        def main():
            model = get_model({"cfg": sys.argv[1:]}, None)
            json.dump({"argv": sys.argv[1:], "name": __name__,
                       "denoiser": type(model.denoiser).__module__ + "." + type(model.denoiser).__name__,
                       "scheduler": type(model.scheduler).__module__ + "." + type(model.scheduler).__name__,
                       "noise_scheduler": type(model.noise_scheduler).__module__,
                       "loop": model._diffusion_reverse.__func__.__module__,
                       "loop_is_instance_binding": "_diffusion_reverse" in vars(model)}, open("out_test.json", "w"))
        if __name__ == "__main__":
            main()
        """,
    "unbounded_synthesis.py": """
        import json, sys
        from convofusion.models.get_model import get_model
        def diffusion_reverse_forecast(model, encoder_hidden_states, lengths=None, preseq=None, cond_masks=dict(), focus_indices=[]):
            return "reference rollout loop"
        def process_samples(model):
            return diffusion_reverse_forecast.__module__          # (looked up in this module's globals at call time, like :438)
        def main():
            model = get_model(None, None)
            json.dump({"rollout": process_samples(model), "denoiser": type(model.denoiser).__module__}, open("out_rollout.json", "w"))
        if __name__ == "__main__":
            main()
        """,
    "plain_script.py": """
        import json
        from convofusion.config import get_obj_from_str
        json.dump({"name": __name__, "denoiser": get_obj_from_str("convofusion.models.architectures.denoiser.Denoiser").__module__},
                  open("out_plain.json", "w"))
        """,
}


def _write_tree(tmp_path):
    for rel, body in TREE.items():
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(body))


def _launch(tmp_path, *args):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "convofusion_amd.run", *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r


def test_launcher_binds_targets_installs_the_loop_and_patches_the_rollout(tmp_path):
    _write_tree(tmp_path)
    # without the launcher the stand-in tree resolves to its own classes (and there is no diffusers here: the scheduler target fails)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", "from convofusion.config import get_obj_from_str as g; "
                        "print(g('convofusion.models.architectures.denoiser.Denoiser').ORIGIN)"], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert "stand-in reference denoiser" in r.stdout
    _launch(tmp_path, "test.py", "--cfg", "configs/config_cf_beatdnd.yaml", "--nodebug")
    got = json.load(open(tmp_path / "out_test.json"))
    assert got["argv"] == ["--cfg", "configs/config_cf_beatdnd.yaml", "--nodebug"] and got["name"] == "test"
    assert got["denoiser"] == "convofusion_amd.denoiser.Denoiser"                   # binding 1: yaml target -> HIP mirror, no yaml edit
    assert got["scheduler"] == "convofusion_amd.scheduler.DDPMScheduler" and got["noise_scheduler"] == "convofusion_amd.scheduler"
    assert got["loop"] == "convofusion_amd.installer" and got["loop_is_instance_binding"]   # binding 2: install(model) inside get_model
    _launch(tmp_path, "unbounded_synthesis.py")
    got = json.load(open(tmp_path / "out_rollout.json"))
    assert got["rollout"] == "convofusion_amd.installer" and got["denoiser"] == "convofusion_amd.denoiser"   # binding 3
    _launch(tmp_path, "plain_script.py")                                            # no main(): run as __main__ with the targets redirected
    got = json.load(open(tmp_path / "out_plain.json"))
    assert got == {"name": "__main__", "denoiser": "convofusion_amd.denoiser"}


def test_launcher_functions_are_idempotent_and_documented():
    import importlib
    run = importlib.import_module("convofusion_amd.run")
    assert "no yaml edit" in run.__doc__.lower() or "NO yaml edit" in run.__doc__
    keep = {k: sys.modules.get(k) for k in (run.REF_DENOISER_MODULE, "diffusers")}
    try:
        bound = run.redirect_targets()
        assert bound == [run.REF_DENOISER_MODULE + ".Denoiser", "diffusers.DDPMScheduler", "diffusers.DDIMScheduler"]
        from convofusion_amd.denoiser import Denoiser
        assert sys.modules[run.REF_DENOISER_MODULE].Denoiser is Denoiser
        assert run.redirect_targets() == bound
    finally:
        for k, v in keep.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
