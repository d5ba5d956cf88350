#!/bin/bash
# Window-50 TCN path (round 4): parity tests, then C5-TCN on the 8-sequence time-resident kernels against round 3's path
# (DOF_TCN_RESIDENT_MAX_T=25), the C4 / C2-TCN lines as the regression check of the shared kernel, and a kernel trace.
#   bash tools/w50_ab.sh            (on the GPU box; writes gpurun_out/w50/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/w50
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r03.py -m gpu -q -k "tcn" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
timeout 600 python tools/bench_configs.py --only c5tcn --steps 6 --warmup 14 > $OUT/c5tcn_new.json 2> $OUT/c5tcn_new.err
DOF_TCN_RESIDENT_MAX_T=25 timeout 600 python tools/bench_configs.py --only c5tcn --steps 6 --warmup 14 > $OUT/c5tcn_r03path.json 2> $OUT/c5tcn_r03path.err
timeout 600 python tools/bench_configs.py --only c4,c2tcn --steps 6 --warmup 25 > $OUT/c4_c2tcn.json 2> $OUT/c4_c2tcn.err
cat $OUT/c5tcn_new.json $OUT/c5tcn_r03path.json $OUT/c4_c2tcn.json
bash tools/prof_c4.sh w50/prof_c5tcn c5tcn
