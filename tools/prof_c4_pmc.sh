#!/bin/bash
# HBM bytes per launch of the C4 (contrastive TCN) kernels: FETCH_SIZE / WRITE_SIZE passes (separate, as
# MI355X_MICROARCH.md prescribes) over two eager steps -> gpurun_out/<tag>/pmc_{FETCH,WRITE}_SIZE.txt
TAG=${1:-c4_pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_c4_$ctr
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/rp_c4_$ctr -o r -- python $ROOT/tools/bench_configs.py --only c4 --steps 2 --warmup 1 > $OUT/$ctr.bench.json 2> $OUT/$ctr.err
  db=$(find /tmp/rp_c4_$ctr -name '*.db' | head -1)
  python $ROOT/tools/pmc_summary.py "$db" k_tcn > $OUT/pmc_$ctr.txt 2>> $OUT/$ctr.err
done
