#!/bin/bash
# copy one final_round.sh pass from gpurun_out/<tag>/ into profiles/<rNN>_*:   bash tools/collect_profiles.sh <tag> [rNN]
TAG=${1:-r6final}
RN=${2:-r06}
S=gpurun_out/$TAG
P=profiles
cp $S/bench_default.json $P/${RN}_bench_default.json
cp $S/bench_under_rocprof.json $P/${RN}_bench_under_rocprof.json
cp $S/step_kernel_stats_graph.md $P/${RN}_step_kernel_stats_graph.md
cp $S/step_timeline.md $P/${RN}_step_timeline.md
cp $S/step_roofline.md $P/${RN}_step_roofline.md
cp $S/step_pmc.json $P/${RN}_step_pmc.json
cp $S/gather_pmc.json $P/${RN}_gather_pmc.json
cp $S/dp_world1.txt $P/${RN}_dp_world1.txt
cp $S/gru_probe.txt $P/${RN}_gru_probe.txt
for L in 4 6 16 32; do cp $S/l$L/kernel_stats.md $P/${RN}_latent${L}_kernel_stats.md; done
cp $S/c3_kernel_stats.md $P/${RN}_c3_kernel_stats.md
cp $S/c5_kernel_stats.md $P/${RN}_c5_kernel_stats.md
cp $S/latent_times.txt $P/${RN}_latent_sizes_times.txt
cp $S/c4/kernel_stats.md $P/${RN}_c4_kernel_stats.md
cp $S/c2tfm/kernel_stats.md $P/${RN}_c2_transformer_kernel_stats.md
cp $S/c5tcn/kernel_stats.md $P/${RN}_c5tcn_kernel_stats.md
cp $S/c4_pmc.md $P/${RN}_c4_pmc_table.md
cp $S/configs.jsonl $P/${RN}_configs.jsonl
cp $S/bench_c4.json $P/${RN}_bench_config_c4.json
cp $S/bench_c5.json $P/${RN}_bench_config_c5.json
[ -f $S/tcn_conv_probe.txt ] && cp $S/tcn_conv_probe.txt $P/${RN}_tcn_conv_probe.txt
tail -3 $S/pytest_gpu.txt > $P/${RN}_pytest_gpu_tail.txt
ls -la $P | grep ${RN}_ | wc -l
