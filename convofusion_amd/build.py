"""Build libcfdenoise.so (hipcc, gfx950) in-tree.  No GPU is needed to build."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcfdenoise.so")
# translation units of the library (each includes csrc/cfd_internal.hpp; kernels are templates in the headers, so a unit only compiles the
# kernels it launches)
SOURCES = ["cfd_core.hip", "cfd_problem.hip", "cfd_forward.hip", "cfd_sample.hip", "cfd_blocks.hip", "cfd_weg.hip", "cfd_dev.hip"]
HEADERS = ["cfd_internal.hpp", "cfd_common.hpp", "gemm_sp.hpp", "rows.hpp", "attn_fused.hpp", "xattn_fused.hpp", "rowtile.hpp", "rowtile_bwd.hpp", "grad.hpp",
           "weg_eval.hpp", "weg_rt.hpp", os.path.join("..", "..", "include", "cfdenoise.h"), os.path.join("..", "..", "include", "cfdenoise_dev.h")]
DEPS = SOURCES + HEADERS
OBJ_DIR = os.path.join(CSRC, ".obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_hash():
    """sha256 over the sources the library is built from (first 16 hex digits): compiled into the library
    (``cfd_source_hash``) so ``_lib.load`` can tell a library built from other sources -- a stale one after a pull that
    changed a struct in cfdenoise.h would corrupt arguments instead of failing.  Content-based: immune to copied trees
    whose modification times are meaningless."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        with open(os.path.join(CSRC, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile the HIP library for gfx950: the translation units in parallel (one hipcc per unit), then one link.  Returns the path of
    the shared object.  ``extra_flags`` / ``out``: developer builds (-D switches) under another name; they do not use the object cache."""
    lib = os.path.abspath(out) if out else LIB
    if not force and not extra_flags and not is_stale():
        return lib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    obj_dir = OBJ_DIR if not extra_flags else OBJ_DIR + "." + str(os.getpid())
    os.makedirs(obj_dir, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    define = f'-DCFD_SOURCE_HASH="{source_hash()}"'

    def compile_unit(srcname):
        obj = os.path.join(obj_dir, srcname.replace(".hip", ".o"))
        src = os.path.join(CSRC, srcname)
        # (cfd_core.hip carries the source hash: it is recompiled whenever any source changed)
        fresh = os.path.exists(obj) and os.path.getmtime(obj) > max(newest_header, os.path.getmtime(src)) and srcname != "cfd_core.hip"
        if fresh and not extra_flags and not force:      # (force: every unit is recompiled -- the driver's "does it build" check)
            return obj
        cmd = [hipcc, *FLAGS, *extra_flags, define, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    tmp = f"{lib}.{os.getpid()}.tmp"     # never a half-written file under the final name: another process may be mapping it
    try:
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_unit, SOURCES))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
        if extra_flags:
            shutil.rmtree(obj_dir, ignore_errors=True)
    return lib


if __name__ == "__main__":
    import sys
    # python -m convofusion_amd.build [-DXA_STAMP=1 ...] [-o other.so]: developer builds with extra switches under another name
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    outp = sys.argv[sys.argv.index("-o") + 1] if "-o" in sys.argv else None
    print(build(force=True, verbose=True, extra_flags=flags, out=outp))
