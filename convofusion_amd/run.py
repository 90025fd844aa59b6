"""Zero-edit launcher: run the reference's own scripts on the MI355X engine with NO yaml edit and NO source edit.

    cd <reference checkout>
    python -m convofusion_amd.run test.py --cfg configs/config_cf_beatdnd.yaml --cfg_assets configs/assets.yaml ...
    python -m convofusion_amd.run unbounded_synthesis.py --cfg ...

What it binds, in this process, before the script's first line runs (nothing has touched the GPU yet):

1. The reference instantiates its denoiser and schedulers from dotted class paths in yaml
   (``instantiate_from_config`` -> ``get_obj_from_str`` -> ``importlib.import_module``, convofusion/config.py:16-31;
   configs/modules/denoiser.yaml:2 ``convofusion.models.architectures.denoiser.Denoiser``, configs/modules/scheduler.yaml:2,13
   ``diffusers.DDPMScheduler``).  ``importlib.import_module`` returns what ``sys.modules`` holds, so the launcher seeds
   ``sys.modules["convofusion.models.architectures.denoiser"]`` with a module whose ``Denoiser`` is
   ``convofusion_amd.denoiser.Denoiser`` and makes ``diffusers.DDPMScheduler`` / ``DDIMScheduler`` the HIP-backed schedulers
   (attributes of the real ``diffusers`` package when it is installed, a stand-in module when it is not).
2. ``convofusion.models.get_model.get_model`` (test.py:14,67; unbounded_synthesis.py:16) is wrapped: the model it returns gets
   ``convofusion_amd.install(model)`` -- the fused loop bound as ``model._diffusion_reverse`` (convofusion.py:391, called :251, :1023).
3. The script is loaded as a module (its ``if __name__ == "__main__"`` guard does not fire), ``diffusion_reverse_forecast`` -- which
   unbounded_synthesis.py defines itself (:28) and looks up in its own globals (:438) -- is rebound there
   (``convofusion_amd.patch_rollout``), and then its ``main()`` is called (test.py:39,151-152; unbounded_synthesis.py:520,578-579).
   A script without ``main()`` is run with ``runpy`` as ``__main__`` with bindings 1 and 2 only.

``CFD_RUN_ATTENTION_STEPS`` = auto | last | all selects ``install``'s attention dict (default auto).
"""
import importlib
import importlib.util
import os
import runpy
import sys
import types

REF_DENOISER_MODULE = "convofusion.models.architectures.denoiser"      # configs/modules/denoiser.yaml:2
REF_GET_MODEL_MODULE = "convofusion.models.get_model"                   # test.py:14, unbounded_synthesis.py:16


def redirect_targets():
    """Binding 1: the yaml's dotted targets resolve to the HIP mirrors.  Returns the names it bound."""
    from . import denoiser as amd_denoiser
    from . import scheduler as amd_scheduler
    bound = []
    stub = types.ModuleType(REF_DENOISER_MODULE)
    stub.__doc__ = "convofusion_amd.run: stands in for the reference's denoiser module; Denoiser is the MI355X engine's"
    stub.Denoiser = amd_denoiser.Denoiser
    stub.__cfd_redirect__ = True
    sys.modules[REF_DENOISER_MODULE] = stub
    bound.append(REF_DENOISER_MODULE + ".Denoiser")
    try:
        diffusers = importlib.import_module("diffusers")
    except ImportError:
        diffusers = types.ModuleType("diffusers")
        diffusers.__doc__ = "convofusion_amd.run: stand-in for the absent diffusers package (schedulers only)"
        sys.modules["diffusers"] = diffusers
    diffusers.DDPMScheduler = amd_scheduler.DDPMScheduler
    diffusers.DDIMScheduler = amd_scheduler.DDIMScheduler
    bound += ["diffusers.DDPMScheduler", "diffusers.DDIMScheduler"]
    return bound


def wrap_get_model(attention_steps="auto"):
    """Binding 2: ``get_model`` returns a model with the fused loop installed.  Returns True if the reference module was found."""
    try:
        mod = importlib.import_module(REF_GET_MODEL_MODULE)
    except ImportError:
        return False
    original = mod.get_model
    if getattr(original, "__cfd_wrapped__", False):
        return True

    def get_model(*args, **kwargs):
        from .installer import install
        return install(original(*args, **kwargs), attention_steps=attention_steps)

    get_model.__cfd_wrapped__ = True
    get_model.__wrapped__ = original
    get_model.__doc__ = original.__doc__
    mod.get_model = get_model
    return True


def run_script(path, argv):
    """Bindings 1-3, then the script.  ``argv``: the script's own arguments."""
    path = os.path.abspath(path)
    if not os.path.isfile(path):
        raise SystemExit(f"convofusion_amd.run: no such script: {path}")
    sys.argv = [path] + list(argv)
    script_dir = os.path.dirname(path)
    if script_dir not in sys.path:
        sys.path.insert(0, script_dir)           # what `python script.py` does: the script's directory first
    redirect_targets()
    found = wrap_get_model(os.environ.get("CFD_RUN_ATTENTION_STEPS", "auto"))
    if not found:
        print("convofusion_amd.run: convofusion.models.get_model is not importable from here (run from the reference checkout); "
              "the denoiser / scheduler targets are redirected, the fused loop is not installed", file=sys.stderr)
    name = os.path.splitext(os.path.basename(path))[0]
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    with open(path, "r") as f:
        has_main = "\ndef main(" in "\n" + f.read()
    if not has_main:
        return runpy.run_path(path, run_name="__main__")
    sys.modules[name] = module                   # (functions defined in the script pickle by this name: dataloader workers)
    spec.loader.exec_module(module)
    if hasattr(module, "diffusion_reverse_forecast"):
        from .installer import patch_rollout
        patch_rollout(module)
    return module.main()


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help"):
        print(__doc__)
        raise SystemExit(0 if len(sys.argv) > 1 else 2)
    run_script(sys.argv[1], sys.argv[2:])


if __name__ == "__main__":
    main()
