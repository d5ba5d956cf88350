cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r04g/pytest_all.txt
