#!/bin/bash
# PMC counters of one kernel family in the C2-shape step at a latent size (eager launches), one rocprofv3 pass per counter group:
#   bash tools/prof_latent_pmc.sh <latent> <tag> <kernel filter> "<ctrs pass 1>" ["<ctrs pass 2>" ...]
LAT=$1; TAG=$2; FILT=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rm -rf /tmp/rp_lp_$i
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/rp_lp_$i -o r -- python $ROOT/bench.py --latent $LAT --no-cpu-baseline --no-secondary --no-graph --steps 4 --warmup 2 --gather-iters 1 --sustain-seconds 0 > $OUT/pass$i.bench.json 2> $OUT/pass$i.err
  db=$(find /tmp/rp_lp_$i -name '*.db' | head -1)
  python $ROOT/tools/pmc_summary.py "$db" "$FILT" > $OUT/pmc_pass$i.txt 2>> $OUT/pass$i.err
done
cat $OUT/pmc_pass*.txt
