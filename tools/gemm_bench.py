"""Micro-benchmark of the split-pair MFMA GEMM at the token-GEMM shapes (developer tool).
usage: python tools/gemm_bench.py [cfg ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd import _lib  # noqa: E402

lib = _lib.load()
h = _lib.create_handle(0)
cfgs = [int(x) for x in sys.argv[1:]] or [1, 5, 4]
for (J, K) in [(43904, 512), (43904, 1024), (3584, 512)]:
    for cfg in cfgs:
        ms = C.c_float()
        _lib.check(lib.cfd_bench_gemm(h, 512, J, K, cfg, 20, C.byref(ms)))
        fl = 2.0 * 512 * J * K
        print(f"J={J} K={K} cfg={cfg}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF algorithmic  ({3*fl/ms.value/1e9/2500*100:5.1f}% of bf16/f16 MFMA peak issued)")
