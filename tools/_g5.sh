cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_t
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_t -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 --gather-iters 3 --sustain-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r04i/bench_under_rocprof.json 2>/dev/null
db=$(find /tmp/rp_t -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 60 > $GRAFT_REPO_ROOT/gpurun_out/r04i/kernel_stats_graph.md
head -45 $GRAFT_REPO_ROOT/gpurun_out/r04i/kernel_stats_graph.md | cut -c1-140
