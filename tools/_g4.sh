cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r03.py -x -q -m gpu -k "tcn" 2>&1 | tail -15 | tee gpurun_out/r04f/pytest_tcn.txt
bash tools/gather_pmc.sh > gpurun_out/r04f/gather_pmc.log 2>&1; tail -3 gpurun_out/r04f/gather_pmc.log
bash tools/profile_step_hbm.sh > gpurun_out/r04f/step_pmc.log 2>&1; tail -5 gpurun_out/r04f/step_pmc.log
