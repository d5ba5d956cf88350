#!/bin/bash
# per-kernel trace of the C4 step (contrastive TCN, B=8192): bash tools/prof_c4.sh <tag> [config]
TAG=${1:-c4}
CFG=${2:-c4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_c4
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_c4 -o r -- python $ROOT/tools/bench_configs.py --only $CFG --steps 5 --warmup 25 > $OUT/bench.json 2> $OUT/err.txt
db=$(find /tmp/rp_c4 -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$db" 50 > $OUT/kernel_stats.md
