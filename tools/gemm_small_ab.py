"""Developer tool (GPU box): the token-side product at the PRODUCT shapes (J = 3584 rows: 32 utterances x 7 chunks x 16 tokens;
J = 112: one utterance) under every tile configuration, three epilogue kinds (0 residual, 1 no stores, 2 fp32 store)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd import _lib  # noqa: E402

lib = _lib.load()
h = _lib.create_handle(0)
cfgs = [int(x) for x in sys.argv[1:]] or [1, 6, 19, 20, 30]
I = int(os.environ.get("GEMM_I", "512"))
for (J, K) in [(3584, 512), (3584, 1024), (1792, 512), (896, 512), (112, 512)]:
    for epi in ((0, 1) if I == 512 else (3, 1)):
        os.environ["CFD_BENCH_EPI"] = str(epi)
        best = {c: 1e9 for c in cfgs}
        for rep in range(3):
            for cfg in cfgs:
                ms = C.c_float()
                _lib.check(lib.cfd_bench_gemm(h, I, J, K, cfg, 50, C.byref(ms)))
                best[cfg] = min(best[cfg], ms.value)
        print(f"I={I} J={J} K={K} epi={epi}: " + "  ".join(f"cfg {c}: {best[c] * 1e3:6.1f} us" for c in cfgs), flush=True)
