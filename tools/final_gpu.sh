#!/bin/bash
# End-of-round pass on one MI355X box: the step's PMC passes on the final sources, the default bench line, smoke, the GPU suite.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
bash tools/step_roofline.sh > /dev/null 2>&1
cp gpurun_out/step_pmc.json profiles/r04_step_pmc.json   # bench.py below quotes it (sha-stamped)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
bash tools/prof_c4.sh final/prof_c4 c4
bash tools/prof_c4.sh final/prof_c5tcn c5tcn
bash tools/prof_c4_pmc.sh final/c4_pmc
