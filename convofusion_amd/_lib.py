"""ctypes binding of libcfdenoise.so (the C ABI in include/cfdenoise.h).

There is no CPU fallback: importing this module without the built library, or creating a handle
without an MI355X, raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be imported BEFORE the library: torch bundles its own libamdhip64 and the two
#                               HIP runtimes must not both be loaded (loading ours first breaks device discovery)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFD_LIB", os.path.join(HERE, "libcfdenoise.so"))  # CFD_LIB: developer override

NUM_MEM = 5
MEM_NAMES = ("spkemb", "alsn", "tlsn", "apb", "lsnemb")
PROF_CLASSES = ("gemm_token", "gemm_mem", "gemm_attn", "rows", "other", "xattn")

SYMBOLS = [
    "cfd_create", "cfd_destroy", "cfd_last_error", "cfd_source_hash", "cfd_load_tensor", "cfd_finalize_weights",
    "cfd_set_timestep_table", "cfd_forward", "cfd_forward_same_memories", "cfd_sample_begin", "cfd_sample_steps", "cfd_sample_position",
    "cfd_sample_read", "cfd_scheduler_step", "cfd_add_noise", "cfd_philox_normal", "cfd_profile_forward",
    "cfd_test_gemm", "cfd_debug_stop_stage", "cfd_debug_read", "cfd_bench_gemm", "cfd_linear_act",
    "cfd_layer_norm", "cfd_mha", "cfd_add", "cfd_zero_rows", "cfd_gemm_f32", "cfd_softmax", "cfd_softmax_bwd",
    "cfd_layer_norm_bwd", "cfd_ew", "cfd_weg_focus", "cfd_sample_write", "cfd_sample_inpaint", "cfd_weg_eval", "cfd_dyadic_steps",
]


class CfdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcfdenoise error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("latent_dim", C.c_int), ("text_encoded_dim", C.c_int), ("ff_size", C.c_int),
                ("num_layers", C.c_int), ("num_heads", C.c_int), ("device", C.c_int)]


class Memory(C.Structure):
    _fields_ = [("data", C.c_void_p), ("row_map", C.c_void_p), ("key_padding_mask", C.c_void_p),
                ("U", C.c_int), ("S", C.c_int)]


class Mat(C.Structure):
    """cfd_mat: element (z1, z2, r, c) = p[z1*b1 + z2*b2 + r*rs + c*cs]."""
    _fields_ = [("p", C.c_void_p), ("rs", C.c_longlong), ("cs", C.c_longlong), ("b1", C.c_longlong), ("b2", C.c_longlong)]


class DyadicProj(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("hidden", C.c_int), ("out_dim", C.c_int),
                ("spk_a", C.c_void_p), ("spk_b", C.c_void_p), ("tmp", C.c_void_p)]


class SampleArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("L", C.c_int), ("G", C.c_int), ("guidance_weight", C.c_float * 8),
                ("scheduler", C.c_int), ("num_train_timesteps", C.c_int), ("num_inference_steps", C.c_int),
                ("clip_sample", C.c_int), ("eta", C.c_float), ("set_alpha_to_one", C.c_int),
                ("steps_offset", C.c_int), ("alphas_cumprod", C.c_void_p), ("init_latents", C.c_void_p),
                ("step_noise", C.c_void_p), ("seed", C.c_uint64), ("first_utterance", C.c_uint32),
                ("preseq", C.c_void_p), ("preseq_len", C.c_int), ("mem", Memory * NUM_MEM),
                ("skip_zero_weight_chunks", C.c_int), ("dynamic_memory_mask", C.c_int),
                ("timesteps", C.c_void_p), ("num_timesteps", C.c_int), ("att_ring", C.c_void_p * NUM_MEM),
                ("operand_policy", C.c_int)]


class WegArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("L", C.c_int), ("timestep", C.c_int), ("latents", C.c_void_p), ("mem", Memory * NUM_MEM),
                ("tok_off", C.c_void_p), ("tok_idx", C.c_void_p), ("last", C.c_int), ("kernel3", C.c_float * 3), ("reuse_memory_side", C.c_int)]


_lib = None


def _built_hash():
    """The source hash embedded in the built library file ("missing" if there is none)."""
    try:
        with open(LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return "missing"
    k = blob.find(b"cfd-src-hash:")
    return blob[k + 13:k + 29].decode(errors="replace") if k >= 0 else "missing"


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m convofusion_amd.build` "
            "(hipcc --offload-arch=gfx950).  convofusion_amd has no CPU fallback.")
    if "CFD_LIB" not in os.environ:      # (a developer override is taken as it is)
        from . import build
        want = build.source_hash()
        if _built_hash() != want:
            # built from other sources (an update changed csrc/ or include/cfdenoise.h): a stale library would mis-read
            # struct arguments instead of failing.  hipcc needs no GPU, so rebuild before the file is mapped; without hipcc refuse.
            # One process rebuilds (file lock: the ranks of a multi-GPU launch all get here at once); build.build() writes to a
            # temporary name and renames it into place, and the hash is read again before the file is mapped.
            import fcntl
            try:
                lock = open(LIB_PATH + ".lock", "w")
            except OSError as e:    # read-only install: the package directory cannot hold the lock (nor a rebuilt library)
                raise ImportError(f"{LIB_PATH} was built from other sources (hash {_built_hash()}, sources {want}) and the package "
                                  f"directory is not writable ({e}); run `python -m convofusion_amd.build` where it is") from e
            with lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    have = _built_hash()
                    if have != want:
                        try:
                            build.build(force=True)
                        except Exception as e:
                            raise ImportError(f"{LIB_PATH} was built from other sources (hash {have}, sources {want}) and could not be "
                                              f"rebuilt ({e}); run `python -m convofusion_amd.build`") from e
                        if _built_hash() != want:
                            raise ImportError(f"{LIB_PATH}: rebuilt library still carries hash {_built_hash()}, sources are {want}")
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
    lib = C.CDLL(LIB_PATH)
    lib.cfd_last_error.restype = C.c_char_p
    lib.cfd_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.cfd_destroy.argtypes = [C.c_void_p]
    lib.cfd_destroy.restype = None
    lib.cfd_load_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.cfd_finalize_weights.argtypes = [C.c_void_p]
    lib.cfd_set_timestep_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.cfd_forward_same_memories.argtypes = [C.c_void_p]
    lib.cfd_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                C.POINTER(Memory), C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]
    lib.cfd_sample_begin.argtypes = [C.c_void_p, C.POINTER(SampleArgs), C.c_void_p]
    lib.cfd_sample_steps.argtypes = [C.c_void_p, C.c_int]
    lib.cfd_sample_position.argtypes = [C.c_void_p]
    lib.cfd_sample_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.cfd_scheduler_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.cfd_add_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_size_t, C.c_void_p]
    lib.cfd_philox_normal.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_void_p]
    lib.cfd_profile_forward.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.cfd_test_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p]
    lib.cfd_bench_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.cfd_linear_act.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]
    lib.cfd_layer_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    lib.cfd_mha.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                            C.c_void_p, C.c_void_p]
    lib.cfd_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.cfd_zero_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
    lib.cfd_gemm_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Mat), C.POINTER(Mat), C.POINTER(Mat),
                                 C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    lib.cfd_softmax.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_longlong, C.c_void_p]
    lib.cfd_softmax_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
    lib.cfd_layer_norm_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_float,
                                       C.c_int, C.c_void_p]
    lib.cfd_ew.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_longlong,
                           C.c_longlong, C.c_float, C.c_void_p]
    lib.cfd_weg_focus.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                  C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cfd_sample_write.argtypes = [C.c_void_p, C.c_void_p]
    lib.cfd_sample_inpaint.argtypes = [C.c_void_p]
    lib.cfd_weg_eval.argtypes = [C.c_void_p, C.POINTER(WegArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
    lib.cfd_dyadic_steps.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(DyadicProj), C.c_int]
    lib.cfd_debug_stop_stage.argtypes = [C.c_void_p, C.c_int]
    lib.cfd_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("cfd_last_error", "cfd_destroy", "cfd_source_hash"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def wrote(*tensors):
    """Tell torch that the library wrote these caller-owned tensors through their raw pointers: bumps their version counters
    (``torch.autograd.graph.increment_version``), which is what ``Denoiser.forward`` keys its reuse of the memories' projections on --
    a ctypes write is otherwise invisible to it (the same-memories staleness hole of round 5)."""
    live = [t for t in tensors if t is not None]
    if live:
        torch.autograd.graph.increment_version(live)


def check(code):
    if code != 0:
        raise CfdError(code, load().cfd_last_error().decode())
    return code


def create_handle(device_index=0, num_layers=9, latent_dim=128, d_model=512, ff_size=1024, num_heads=4):
    lib = load()
    cfg = Config(latent_dim, d_model, ff_size, num_layers, num_heads, device_index)
    h = C.c_void_p()
    check(lib.cfd_create(C.byref(cfg), C.byref(h)))
    return h
