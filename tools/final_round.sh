#!/bin/bash
# End-of-round measurement pass on one MI355X box: everything MEASUREMENTS.md quotes, into gpurun_out/<tag>/ (copy the
# summaries to profiles/rNN_*).    bash tools/final_round.sh <tag> [rNN]
TAG=${1:-final}
RN=${2:-r06}
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
for L in 4 6 16 32; do bash tools/prof_latent.sh $L $TAG/l$L > /dev/null 2>&1; done
bash tools/prof_secondary.sh $TAG > /dev/null 2>&1
bash tools/latent_times.sh "4 5 6 7 8 9 10 12 14 16 20 24 32" > $OUT/latent_times.txt 2>&1
# 5b. the TCN family: C4 per kernel (time, then HBM bytes), C2 / C5 with the TCN family, the transformer family, every
#     secondary configuration as one JSON line each, bench.py's own --config lines, the convolution kernels in isolation
bash tools/prof_c4.sh $TAG/c4 > /dev/null 2>&1
bash tools/prof_c4.sh $TAG/c2tfm c2tfm > /dev/null 2>&1
bash tools/prof_c4.sh $TAG/c5tcn c5tcn > /dev/null 2>&1
bash tools/prof_c4_pmc.sh $TAG/c4_pmc > /dev/null 2>&1
python tools/c4_pmc_table.py $OUT/c4_pmc/pmc_FETCH_SIZE.txt $OUT/c4_pmc/pmc_WRITE_SIZE.txt > $OUT/c4_pmc.md 2> $OUT/c4_pmc.err
timeout 1200 python tools/bench_configs.py --steps 30 --warmup 8 > $OUT/configs.jsonl 2> $OUT/configs.err
timeout 600 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
timeout 600 python bench.py --config c5 --steps 20 --warmup 5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
if [ -x tools/probe/tcn_conv_probe ]; then
  (timeout 120 tools/probe/tcn_conv_probe 1; timeout 120 tools/probe/tcn_conv_probe 8) > $OUT/tcn_conv_probe.txt 2>&1
fi
# 6. smoke + the GPU suite on the same build
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
