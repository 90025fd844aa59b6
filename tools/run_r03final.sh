cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r03_pytest.log
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
tail -3 gpurun_out/r03_profile_round.log
timeout 900 python tools/bench_configs.py > gpurun_out/r03_other_configs.json 2> gpurun_out/r03_other_configs.err
tail -c 600 gpurun_out/r03_other_configs.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
