cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
B="python bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 30 --gather-iters 2 --sustain-seconds 0"
for rep in 1 2 3; do
for cfg in "DOF_GRU8_MFMA=0" "X=1"; do
  echo "$cfg: $(env $cfg timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), d['config']['final_total_loss'])")"
done; done | tee gpurun_out/r04h/ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r03.py -x -q -m gpu -k "not tcn and not tfm and not transformer and not preprocess and not turtle" 2>&1 | tail -4 | tee gpurun_out/r04h/pytest.txt
