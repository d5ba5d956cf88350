#!/bin/bash
# per-kernel trace of the C2-shape step at latent 16 (eager launches): bash tools/prof_latent16.sh <tag>
TAG=${1:-l16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_l16
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_l16 -o r -- python $ROOT/bench.py --latent 16 --no-cpu-baseline --no-secondary --no-graph --steps 20 --warmup 5 --gather-iters 1 --sustain-seconds 0 > $OUT/bench.json 2> $OUT/err.txt
db=$(find /tmp/rp_l16 -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$db" 40 > $OUT/kernel_stats.md
