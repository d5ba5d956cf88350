#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the eager C2 step -> gpurun_out/step_pmc.json (copy to profiles/rNN_step_pmc.json)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
STEPS=30; WARM=10
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$ctr
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/rp_$ctr -o r -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-graph --steps $STEPS --warmup $WARM --gather-iters 1 --sustain-seconds 0 > $ROOT/gpurun_out/pmc_$ctr.bench.json 2> $ROOT/gpurun_out/pmc_$ctr.err
done
python $ROOT/tools/step_hbm_bytes.py $(find /tmp/rp_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/rp_WRITE_SIZE -name '*.db' | head -1) $((STEPS + WARM)) $ROOT > $ROOT/gpurun_out/step_pmc.json
cat $ROOT/gpurun_out/step_pmc.json | head -30
