#!/bin/bash
# rocprofv3 passes over the C2 train step (bench.py): kernel-trace stats in hipGraph mode, then separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / SQ counters cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage (on the GPU box, from the repo root):  bash tools/profile_step.sh <tag> [extra bench flags]
# Writes text summaries to gpurun_out/prof_<tag>/*.txt (the databases stay on the box).
set -u
TAG=${1:-step}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 --gather-iters 3 $*"
cd /tmp

run_pass() {  # name, rocprof flags..., then the command
  local name=$1
  shift
  rm -rf /tmp/rp_$name
  local extra=""
  case $name in pmc_*) extra="--no-graph" ;; esac   # counters are collected per eager dispatch
  timeout 600 rocprofv3 "$@" -d /tmp/rp_$name -o r -- $BENCH $extra >"$OUT/$name.bench.json" 2>"$OUT/$name.err"
  local db
  db=$(find /tmp/rp_$name -name '*.db' | head -1)
  echo "$db"
}

db=$(run_pass trace_graph --kernel-trace --stats)
[ -n "$db" ] && python $ROOT/tools/rocpd_stats.py "$db" 60 >"$OUT/kernel_stats_graph.md" 2>>"$OUT/trace_graph.err"

for ctr in FETCH_SIZE WRITE_SIZE; do
  db=$(run_pass pmc_$ctr --pmc $ctr --kernel-trace)
  [ -n "$db" ] && python $ROOT/tools/pmc_summary.py "$db" >"$OUT/pmc_$ctr.txt" 2>>"$OUT/pmc_$ctr.err"
done
db=$(run_pass pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace)
[ -n "$db" ] && python $ROOT/tools/pmc_summary.py "$db" >"$OUT/pmc_sq.txt" 2>>"$OUT/pmc_sq.err"
db=$(run_pass pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace)
[ -n "$db" ] && python $ROOT/tools/pmc_summary.py "$db" >"$OUT/pmc_sq2.txt" 2>>"$OUT/pmc_sq2.err"
ls -la "$OUT"
