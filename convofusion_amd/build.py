"""Build libcfdenoise.so (hipcc, gfx950) in-tree.  No GPU is needed to build."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcfdenoise.so")
SOURCES = ["cfd_api.hip"]
DEPS = ["cfd_api.hip", "gemm_sp.hpp", "rowtile.hpp", "rowtile_bwd.hpp", "weg_rt.hpp", "rows.hpp", "attn_fused.hpp", "xattn_fused.hpp", "grad.hpp", "weg_eval.hpp", "cfd_common.hpp", os.path.join("..", "..", "include", "cfdenoise.h")]


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


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950.  Returns the path of the shared object."""
    if not force and not is_stale():
        return LIB
    tmp = f"{LIB}.{os.getpid()}.tmp"     # never a half-written file under the final name: another process may be mapping it
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
           f'-DCFD_SOURCE_HASH="{source_hash()}"', *[os.path.join(CSRC, s) for s in SOURCES], "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
