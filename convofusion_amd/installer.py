"""No-source-edit installation of the fused loop into a reference model object / script module.

The reference's two loop entry points are a method and a module-level function:

    Convofusion._diffusion_reverse(self, encoder_hidden_states, lengths=None, cond_masks=dict(), focus_indices=[])
        convofusion/models/modeltype/convofusion.py:391   (called at :251 and :1023)
    unbounded_synthesis.diffusion_reverse_forecast(model, encoder_hidden_states, lengths=None, preseq=None,
                                                   cond_masks=dict(), focus_indices=[])
        unbounded_synthesis.py:28                          (called at :438)

``install(model)`` binds the first on the model INSTANCE (``types.MethodType``: instance attributes shadow the class
method, the class and every other instance stay untouched); ``patch_rollout(module)`` rebinds the second in the script's
module namespace, which is where its call site looks the name up.  With the two yaml edits of INTEGRATION.md sections 1-2
(denoiser and scheduler targets) ``test.py`` and ``unbounded_synthesis.py`` then run unchanged:

    import convofusion_amd
    model = get_model(cfg, dataset)            # reference code, yaml points at convofusion_amd's Denoiser / scheduler
    convofusion_amd.install(model)
    import unbounded_synthesis                 # only for the rollout script
    convofusion_amd.patch_rollout(unbounded_synthesis)
"""
import types


def _check_model(model):
    from .denoiser import Denoiser
    den = getattr(model, "denoiser", None)
    if not isinstance(den, Denoiser):
        raise TypeError("model.denoiser is %s: point configs/modules/denoiser.yaml at convofusion_amd.denoiser.Denoiser "
                        "(INTEGRATION.md section 1) before installing the fused loop" % type(den).__name__)
    if getattr(getattr(model, "scheduler", None), "KIND", None) is None:
        raise TypeError("model.scheduler is %s: point configs/modules/scheduler.yaml at convofusion_amd.scheduler.DDPMScheduler / "
                        "DDIMScheduler (INTEGRATION.md section 2)" % type(getattr(model, "scheduler", None)).__name__)


def _diffusion_reverse(self, encoder_hidden_states, lengths=None, cond_masks=dict(), focus_indices=[]):
    """Bound replacement of ``Convofusion._diffusion_reverse`` (same signature, same return value: latents [L, B, 128] and
    the attention-matrix dict -- see ``convofusion_amd.sampler.diffusion_reverse`` for which entries it holds)."""
    from .sampler import diffusion_reverse
    return diffusion_reverse(self, encoder_hidden_states, lengths, cond_masks, focus_indices,
                             attention_steps=getattr(self, "_cfd_attention_steps", "auto"))


def diffusion_reverse_forecast(model, encoder_hidden_states, lengths=None, preseq=None, cond_masks=dict(), focus_indices=[]):
    """Replacement of ``unbounded_synthesis.diffusion_reverse_forecast`` (same signature and return value)."""
    from .sampler import diffusion_reverse_forecast as impl
    return impl(model, encoder_hidden_states, lengths, preseq, cond_masks, focus_indices)


def install(model, attention_steps="auto", operands=None):
    """Bind the fused loop as ``model._diffusion_reverse``.  Returns the model.  ``uninstall`` removes the binding.

    ``attention_steps``: which entries the returned attention-matrix dict holds.  The reference keeps the full-conditioning
    chunk's maps of EVERY iteration (convofusion.py:517-523) and its result writer dumps one ``att_<t>.npy`` per entry and
    memory (convofusion/models/modeltype/base.py:252-259).  "all" reproduces the reference's dict, so that an unchanged ``test.py``
    writes the same files: the captured iteration stores the maps itself (+2 - 5 % run time; a ring of iterations x B x layers x L x keys
    floats, 4 GB for 32 utterances x 1000 iterations at the product shape); beyond ``sampler.ATT_RING_MAX_BYTES`` it costs one extra
    forward of the B full-conditioning rows and one host round trip per iteration.  "last" returns the final iteration's entry only --
    the loop then does not evaluate the zero-weight full-conditioning chunk.  "auto" (default): "all" while the ring fits the budget,
    "last" beyond it.

    ``operands``: the run's operand policy for the fused cross-attention (``cfd_sample_args.operand_policy``; None = the default of
    ``convofusion_amd.sampler.OPERAND_POLICY`` for the model's scheduler: DDPM runs carry the long memories' folded keys / values as single
    fp16).  ``operands=0`` keeps fp16 split pairs everywhere -- the precision escape, at ~8 % of the loop's throughput at the headline shape."""
    _check_model(model)
    if attention_steps not in ("auto", "last", "all"):
        raise ValueError("attention_steps must be 'auto', 'last' or 'all'")
    model._cfd_attention_steps = attention_steps
    model._cfd_operands = operands
    model._diffusion_reverse = types.MethodType(_diffusion_reverse, model)
    return model


def uninstall(model):
    """Remove the instance binding: the class's own ``_diffusion_reverse`` is visible again."""
    if "_diffusion_reverse" in vars(model):
        del model._diffusion_reverse
    vars(model).pop("_cfd_attention_steps", None)
    vars(model).pop("_cfd_operands", None)
    return model


def patch_rollout(module):
    """Rebind ``diffusion_reverse_forecast`` in ``module`` (the imported ``unbounded_synthesis`` script).  Returns the original
    function so it can be put back (``module.diffusion_reverse_forecast = original``)."""
    original = getattr(module, "diffusion_reverse_forecast", None)
    if original is None:
        raise AttributeError("%s has no diffusion_reverse_forecast to replace" % getattr(module, "__name__", module))
    module.diffusion_reverse_forecast = diffusion_reverse_forecast
    return original
