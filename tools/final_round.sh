#!/bin/bash
# End-of-round measurement pass on one MI355X box: everything MEASUREMENTS.md quotes, into gpurun_out/<tag>/ (copy the
# summaries to profiles/rNN_*).    bash tools/final_round.sh <tag> [rNN]
TAG=${1:-final}
RN=${2:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
# 1. PMC passes on the present sources: the step (per-kernel roofline + HBM bytes per step) and the gather launch
bash tools/step_roofline.sh > /dev/null 2>&1
cp gpurun_out/step_roofline.md $OUT/step_roofline.md
cp gpurun_out/step_pmc.json $OUT/step_pmc.json
cp gpurun_out/step_pmc.json profiles/${RN}_step_pmc.json          # bench.py below quotes them (sha-stamped)
bash tools/gather_pmc.sh > $OUT/gather_pmc.log 2>&1
cp gpurun_out/gather_pmc.json $OUT/gather_pmc.json
cp gpurun_out/gather_pmc.json profiles/${RN}_gather_pmc.json
# 2. the default command (with the CPU baseline and the secondary configurations)
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 300 $OUT/bench_default.json; echo
# 3. the headline command under rocprofv3 (hipGraph mode): kernel stats + one step's timeline
cd /tmp
rm -rf /tmp/rp_final
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_final -o r -- python $ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
db=$(find /tmp/rp_final -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$db" 60 > $OUT/step_kernel_stats_graph.md
python $ROOT/tools/rocpd_stats.py "$db" --timeline "k_window_gather<true" > $OUT/step_timeline.md
cd $ROOT
# 4. the data-parallel forms at world 1, the GRU probes
bash tools/bench_dp_world1.sh > $OUT/dp_world1.txt 2>&1
timeout 200 tools/probe/gru16_probe > $OUT/gru_probe.txt 2>&1
# 5. secondary configurations under the profiler
bash tools/prof_latent.sh 16 $TAG/l16 > /dev/null 2>&1
bash tools/prof_latent.sh 32 $TAG/l32 > /dev/null 2>&1
bash tools/prof_secondary.sh $TAG > /dev/null 2>&1
# 6. smoke + the GPU suite on the same build
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
