#!/bin/bash
# Collects the evidence profiles/ holds for one round, on the GPU box:  gpurun -- tools/profile_round.sh r02
#   1. the bench line (default flags)                      -> gpurun_out/<tag>_bench_c2.json, <tag>_bench_realshape.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> gpurun_out/<tag>_stats/
#   3. FETCH_SIZE, WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in SEPARATE --pmc passes (kernel-trace only,
#      as the pool requires)                                -> per-kernel aggregates gpurun_out/<tag>_hbm_traffic.csv, <tag>_mfma_busy.csv
# tools/profile_post.py (run in the build container afterwards) condenses them into profiles/.
# The program after `--` is python itself (no env / bash -c hop: the profiler initialises the GPU before it starts).
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > $out/${tag}_bench_c2.json 2> $out/${tag}_bench_c2.err
python bench.py --steps 50 --warmup 5 --shape R --no-cpu-baseline --no-secondary > $out/${tag}_bench_realshape.json 2>> $out/${tag}_bench_c2.err
rm -rf $out/${tag}_stats $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o run -- python bench.py --steps 20 --warmup 3 --headline-only > $out/${tag}_rocprof_stdout.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o run -- python bench.py --steps 3 --warmup 1 --headline-only > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o run -- python bench.py --steps 3 --warmup 1 --headline-only > $out/${tag}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_mfma -o run -- python bench.py --steps 3 --warmup 1 --headline-only > $out/${tag}_pmc_mfma.log 2>&1
# one utterance at the product shape (C1) and WEG evaluations: kernel stats of the row-tile path
bash tools/c1prof.sh 1 ${tag}_c1 > $out/${tag}_c1.txt 2>&1
bash tools/wegprof.sh ${tag}_weg > $out/${tag}_weg.txt 2>&1
# the raw per-dispatch counter files are large: keep only per-kernel aggregates
python tools/profile_post.py $tag --aggregate-only
rm -rf $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_mfma
ls -la $out | tail -20
