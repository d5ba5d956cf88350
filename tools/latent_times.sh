#!/bin/bash
# C2-shape step time at a list of latent sizes (hipGraph replay): bash tools/latent_times.sh "4 5 6 7" [extra env as VAR=VAL ...]
LATS=${1:-"4 5 6"}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in $LATS; do
  env "$@" python $ROOT/bench.py --latent $L --no-cpu-baseline --no-secondary --steps 30 --warmup 5 --gather-iters 1 --sustain-seconds 0 2>/dev/null \
    | python -c "import sys,json;d=json.loads(sys.stdin.readline());print('latent $L ms_per_step %.3f windows_per_s %.0f' % (d['ms_per_step'], d['value']))"
done
