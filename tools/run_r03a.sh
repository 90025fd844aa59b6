set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03a/pytest.log
cat gpurun_out/r03a/pytest.log
timeout 600 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
tail -c 3000 gpurun_out/r03a/bench.json
timeout 1500 tools/concurrency_variants.sh run 300 > gpurun_out/r03a/variants.log 2>&1
cat gpurun_out/r03a/variants.log
