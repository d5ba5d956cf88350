#!/bin/bash
# PMC passes over the transformer-family C2-shape step (tools/bench_configs.py --only c2tfm, eager launches).
# Usage (GPU box, repo root): bash tools/profile_tfm.sh   -> gpurun_out/prof_tfm/*.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tfm
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/bench_configs.py --only c2tfm --steps 10 --warmup 3"
pass() {
  name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 900 rocprofv3 "$@" --kernel-trace -d /tmp/rp_$name -o r -- $CMD > $OUT/$name.bench.json 2> $OUT/$name.err
  db=$(find /tmp/rp_$name -name '*.db' | head -1)
  [ -n "$db" ] && python $ROOT/tools/pmc_summary.py "$db" k_tfm > $OUT/$name.txt 2>> $OUT/$name.err
}
pass fetch --pmc FETCH_SIZE
pass write --pmc WRITE_SIZE
pass sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU
pass sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
ls -la $OUT
