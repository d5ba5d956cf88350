#!/bin/bash
# rocprofv3 kernel traces of the secondary BASELINE configurations through the product steppers (bench.py's functions):
#   bash tools/prof_secondary.sh   [tag]  ->  gpurun_out/<tag>/{c3,c5}_kernel_stats.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-prof_secondary}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # tag, python expression
  rm -rf /tmp/rp_$1
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_$1 -o r -- python -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print('$1', bench.$2)" > $OUT/$1.log 2>&1
  db=$(find /tmp/rp_$1 -name '*.db' | head -1)
  python $ROOT/tools/rocpd_stats.py "$db" 40 > $OUT/$1_kernel_stats.md
}
run c3 "run_vqvae_product(4096, 512, 30, 8)"
run c5 "run_vade_product(['B', 'W'], 50, 25, 4096, 'recurrent', 30, 8)"
tail -2 $OUT/c3.log $OUT/c5.log
