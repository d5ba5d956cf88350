#!/bin/bash
# Per-kernel PMC counters of one bench_configs.py configuration (one rocprofv3 pass per counter group; --pmc with
# --kernel-trace only, as the pool requires): bash tools/prof_pmc.sh <tag> <config> <kernel filter> "<ctrs pass 1>" ["<ctrs pass 2>" ...]
TAG=$1; CFG=$2; FILT=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rm -rf /tmp/rp_pmc_$i
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/rp_pmc_$i -o r -- python $ROOT/tools/bench_configs.py --only $CFG --steps 2 --warmup 1 > $OUT/pass$i.bench.json 2> $OUT/pass$i.err
  db=$(find /tmp/rp_pmc_$i -name '*.db' | head -1)
  python $ROOT/tools/pmc_summary.py "$db" "$FILT" > $OUT/pmc_pass$i.txt 2>> $OUT/pass$i.err
done
