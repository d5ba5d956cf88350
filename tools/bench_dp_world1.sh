#!/bin/bash
# The data-parallel step at world 1 (a 1-rank RCCL group): torch.distributed vs the C ABI's dof_flat_allreduce,
# two graphs around the eager collective vs the collective captured into the step graph.  bash tools/bench_dp_world1.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  env "$@" timeout 300 python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 400 --warmup 20 --gather-iters 2 --sustain-seconds 0 2>/dev/null \
    | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); dp=d.get('data_parallel') or {}; print(round(d['value']), round(d['ms_per_step'], 4), '|', dp.get('form'), '| self-check:', dp.get('self_check'), '| eager all-reduce us:', dp.get('allreduce_eager_latency_us'))"
}
for rep in 1 2; do
echo "no DP route:            $(run X=1)"
echo "torch, two graphs (the safe form: DOF_DP_NATIVE=0): $(run DOF_BENCH_FORCE_PG=1 DOF_FORCE_DP=1 DOF_DP_NATIVE=0)"
echo "native, two graphs:     $(run DOF_BENCH_FORCE_PG=1 DOF_FORCE_DP=1 DOF_DP_ONE_GRAPH=0)"
echo "torch, one graph (opt-in): $(run DOF_BENCH_FORCE_PG=1 DOF_FORCE_DP=1 DOF_DP_NATIVE=0 DOF_DP_ONE_GRAPH=1)"
echo "native, one graph (default): $(run DOF_BENCH_FORCE_PG=1 DOF_FORCE_DP=1)"
done
