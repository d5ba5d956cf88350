cd $GRAFT_REPO_ROOT
bash tools/prof_secondary.sh > /dev/null 2>&1
bash tools/prof_c4.sh c4_r04 c4 > /dev/null 2>&1
bash tools/prof_c4_pmc.sh c4_pmc_r04 > /dev/null 2>&1
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rp_tfm
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_tfm -o r -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only c2tfm --steps 10 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/c2tfm_r04.json 2>/dev/null
db=$(find /tmp/rp_tfm -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 30 > $GRAFT_REPO_ROOT/gpurun_out/c2tfm_r04_kernel_stats.md
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_secondary gpurun_out/c4_r04 gpurun_out/c4_pmc_r04
tail -c 400 gpurun_out/c4_r04/bench.json
