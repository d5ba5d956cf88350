#!/bin/bash
# same-box A/B of the weight-gradient reduction kernels (k_outer_b3 default, DOF_OUTER_B3=0: the fp32 k_outer)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in 1 0; do
  echo "DOF_OUTER_B3=$v"
  bash $ROOT/tools/latent_times.sh "6 8 12 16 24 32" DOF_OUTER_B3=$v
  DOF_OUTER_B3=$v python $ROOT/tools/bench_configs.py --only c2tfm,c3,c5 --steps 30 --warmup 8 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['workload'][:60], '%.3f ms' % d['ms_per_step'])"
done
