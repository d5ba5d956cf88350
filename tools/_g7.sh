cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
timeout 900 python bench.py > gpurun_out/r04j/bench_default.json 2> gpurun_out/r04j/bench_default.err
tail -c 600 gpurun_out/r04j/bench_default.json
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/rp_t
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_t -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 --gather-iters 3 --sustain-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r04j/bench_under_rocprof.json 2>/dev/null
db=$(find /tmp/rp_t -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 60 > $GRAFT_REPO_ROOT/gpurun_out/r04j/kernel_stats_graph.md
cd $GRAFT_REPO_ROOT
bash tools/bench_dp_world1.sh > gpurun_out/r04j/dp_world1.txt 2>&1
cat gpurun_out/r04j/dp_world1.txt
