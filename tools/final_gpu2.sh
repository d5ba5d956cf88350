#!/bin/bash
# Second end-of-round pass: the headline command under rocprofv3 (hipGraph mode) and the data-parallel forms, final build.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final2
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_final
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_final -o r -- python $ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
db=$(find /tmp/rp_final -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$db" 60 > $OUT/step_kernel_stats_graph.md
cd $ROOT
DOF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --no-cpu-baseline --no-secondary > $OUT/bench_gpus2_shared.json 2> $OUT/bench_gpus2_shared.err
tail -c 400 $OUT/bench_gpus2_shared.json
bash tools/bench_dp_world1.sh > $OUT/dp_world1.txt 2>&1
cat $OUT/dp_world1.txt
