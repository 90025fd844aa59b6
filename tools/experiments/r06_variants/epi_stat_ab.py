"""Developer tool (GPU box): go / no-go for the algebraic LayerNorm fold.  The fold removes a ln_rows launch
(28 us at the headline shape: 90 MB read + 90 MB written) by having the PRODUCING residual product also store the split-pair copy of
its new rows and per-row partial sums.  It can only pay if that epilogue (CFD_BENCH_EPI=4, EpiResidStat) costs less than the ~14 us the
LayerNorm's read is worth.  Interleaved, one process: the 43 904 x 512 x {512, 1024} and 3 584 x 512 x {512, 1024} residual products."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd import _lib  # noqa: E402

lib = _lib.load()
h = _lib.create_handle(0)
for (J, K, cfg) in [(43904, 512, 1), (43904, 1024, 1), (3584, 512, 19), (3584, 1024, 19)]:
    res = {0: [], 4: []}
    for rnd in range(5):
        for epi in (0, 4):
            os.environ["CFD_BENCH_EPI"] = str(epi)
            ms = C.c_float()
            _lib.check(lib.cfd_bench_gemm(h, 512, J, K, cfg, 30, C.byref(ms)))
            res[epi].append(ms.value * 1e3)
    a, b = sorted(res[0])[len(res[0]) // 2], sorted(res[4])[len(res[4]) // 2]
    print(f"J={J} K={K} cfg={cfg}: residual epilogue {a:7.1f} us   + split-pair copy + row statistics {b:7.1f} us   delta {b - a:+6.1f} us   "
          f"(all rounds: {[round(x, 1) for x in res[0]]} / {[round(x, 1) for x in res[4]]})", flush=True)
