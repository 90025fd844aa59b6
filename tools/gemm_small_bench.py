"""Developer tool (GPU box): back-to-back timing of small split-pair products (3 584 and 112 rows) against K, to separate the fixed cost of a
launch (K = 32: one k-step) from the k-loop.  MI355X, residual epilogue: 64 x 64 tiles 4.9 us + 0.4 us per k-step at 3 584 rows (10.9 us at
K = 512; 22 us inside the loop, where the operands were just written by other XCDs), 3.6 us + 0.28 us at 112 rows.
CFD_BENCH_EPI=n: no stores."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd import _lib
lib = _lib.load(); h = _lib.create_handle(0)
for J in (3584, 112):
    for K in (32, 128, 512, 1024):
        for cfg in (19, 1):
            ms = C.c_float()
            _lib.check(lib.cfd_bench_gemm(h, 512, J, K, cfg, 50, C.byref(ms)))
            print(f"J={J} K={K} cfg={cfg}: {ms.value*1e3:7.2f} us")
