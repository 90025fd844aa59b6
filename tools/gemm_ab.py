"""Developer tool (GPU box): the token-side product 43 904 x 512 x K under the tile configurations given on the command line
(default 1 = symmetric 2-stage 128 x 128, 30 = asymmetric ring), with each epilogue kind of cfd_bench_gemm
(0 residual read-modify-write, 1 no stores, 2 plain fp32 store), interleaved and repeated so that clock drift hits all alike.
usage: python tools/gemm_ab.py [cfg ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd import _lib  # noqa: E402

lib = _lib.load()
h = _lib.create_handle(0)
cfgs = [int(x) for x in sys.argv[1:]] or [1, 30]
for (J, K) in [(43904, 512), (43904, 1024), (6272, 512)]:
    for epi in (0, 1, 2):
        os.environ["CFD_BENCH_EPI"] = str(epi)
        best = {c: 1e9 for c in cfgs}
        for rep in range(3):
            for cfg in cfgs:
                ms = C.c_float()
                _lib.check(lib.cfd_bench_gemm(h, 512, J, K, cfg, 30, C.byref(ms)))
                best[cfg] = min(best[cfg], ms.value)
        fl = 2.0 * 512 * J * K
        print(f"J={J} K={K} epi={epi}: " + "  ".join(f"cfg {c}: {best[c] * 1e3:7.1f} us ({3 * fl / best[c] / 1e9 / 2500 * 100:4.1f}% issued)" for c in cfgs), flush=True)
