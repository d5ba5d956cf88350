#!/bin/bash
# quick per-kernel trace of the C2 step (graph mode): bash tools/quick_prof.sh <tag>
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_q
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_q -o r -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 --gather-iters 2 --sustain-seconds 0 > $OUT/bench.json 2> $OUT/err.txt
db=$(find /tmp/rp_q -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$db" 70 > $OUT/kernel_stats.md
python $ROOT/tools/rocpd_stats.py "$db" --timeline "k_window_gather<true" > $OUT/timeline.md
