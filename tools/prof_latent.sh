#!/bin/bash
# per-kernel trace of the C2-shape step at another latent size (eager launches): bash tools/prof_latent.sh <latent> <tag>
LAT=${1:-16}
TAG=${2:-l$LAT}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_lat
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_lat -o r -- python $ROOT/bench.py --latent $LAT --no-cpu-baseline --no-secondary --no-graph --steps 20 --warmup 5 --gather-iters 1 --sustain-seconds 0 > $OUT/bench.json 2> $OUT/err.txt
db=$(find /tmp/rp_lat -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$db" 40 > $OUT/kernel_stats.md
python -c "import json;d=json.load(open('$OUT/bench.json'));print('latent $LAT:', d['ms_per_step'], 'ms/step (eager, under rocprofv3)')"
