#!/bin/bash
# Per-kernel roofline of the C2 step (durations, HBM bytes, matrix-pipe busy): four rocprofv3 runs of the eager step.
#   bash tools/step_roofline.sh   -> gpurun_out/step_roofline.md   (copy to profiles/rNN_step_roofline.md)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
STEPS=30; WARM=10
run() {  # tag, rocprof flags
  tag=$1; shift
  rm -rf /tmp/sr_$tag
  timeout 900 rocprofv3 "$@" -d /tmp/sr_$tag -o r -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-graph --steps $STEPS --warmup $WARM --gather-iters 1 --sustain-seconds 0 > /dev/null 2>&1
  find /tmp/sr_$tag -name '*.db' | head -1
}
T=$(run trace --kernel-trace)
F=$(run fetch --pmc FETCH_SIZE --kernel-trace)
W=$(run write --pmc WRITE_SIZE --kernel-trace)
S=$(run sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace)
python $ROOT/tools/step_roofline.py "$T" "$F" "$W" "$S" $((STEPS + WARM)) > $ROOT/gpurun_out/step_roofline.md
python $ROOT/tools/step_hbm_bytes.py "$F" "$W" $((STEPS + WARM)) $ROOT > $ROOT/gpurun_out/step_pmc.json
cat $ROOT/gpurun_out/step_roofline.md
