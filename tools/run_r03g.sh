cd $GRAFT_REPO_ROOT
export CFD_XA_ROLE=1
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -k headline 2>&1 | grep -v "^$" | tail -40
