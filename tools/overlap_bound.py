"""Developer tool (GPU box): what a one-queue overlap of half-batches could gain at the benchmark shape.
Running utterance-half A's products next to half B's row kernels in ONE launch needs every product at HALF the rows.  This times the
residual product (43 904 x 512 x 512, read-modify-write epilogue) at the full batch, at half of it, and over a sweep of row counts,
interleaved and repeated (best of 5): the product's time follows the number of ROUNDS of 128 x 128 tiles over the 512 workgroup slots
(256 CUs x 2), so two half-batch launches cost more than the full-batch one, and the difference is what an overlap has to win back
before it gains anything.  usage: python tools/overlap_bound.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from convofusion_amd import _lib  # noqa: E402

lib = _lib.load()
h = _lib.create_handle(0)
os.environ["CFD_BENCH_EPI"] = "0"
rows = [43904, 21952, 10976, 16384, 32768, 49152, 65536]
best = {j: 1e9 for j in rows}
for rep in range(5):
    for j in rows:
        ms = C.c_float()
        _lib.check(lib.cfd_bench_gemm(h, 512, j, 512, 1, 30, C.byref(ms)))
        best[j] = min(best[j], ms.value)
for j in rows:
    wgs = (j + 127) // 128 * 4
    print(f"rows {j:6d}: {wgs:5d} workgroups = {wgs / 512:5.2f} rounds of 512 slots   {best[j] * 1e3:7.1f} us   {best[j] * 1e3 / (wgs / 512):6.1f} us per round-equivalent"
          f"   {best[j] * 1e6 / j:6.3f} ns per row", flush=True)
full, half = best[43904] * 1e3, best[21952] * 1e3
print(f"full batch {full:.1f} us; two half batches {2 * half:.1f} us: +{2 * half - full:.1f} us per product before any overlap "
      f"(ln_rows over the full batch: ~27 us, profiles/r04_bench_c2_kernel_stats.csv)")
