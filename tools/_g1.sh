cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "data_parallel or gather or tfm_other_widths or preprocess" 2>&1 | tail -15 > gpurun_out/r04a/pytest.txt
cat gpurun_out/r04a/pytest.txt
DOF_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 --gather-iters 2 --sustain-seconds 0 > gpurun_out/r04a/bench_share2.json 2> gpurun_out/r04a/bench_share2.err
tail -3 gpurun_out/r04a/bench_share2.err; cat gpurun_out/r04a/bench_share2.json | cut -c1-1500
timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-secondary --steps 50 > gpurun_out/r04a/bench_gpus2_onebox.txt 2>&1; tail -2 gpurun_out/r04a/bench_gpus2_onebox.txt
DOF_BENCH_FORCE_PG=1 DOF_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 --gather-iters 5 > gpurun_out/r04a/bench_dp1.json 2> gpurun_out/r04a/bench_dp1.err; cut -c1-1200 gpurun_out/r04a/bench_dp1.json
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 20 --gather-iters 10 > gpurun_out/r04a/bench_plain.json 2> gpurun_out/r04a/bench_plain.err; python -c "
import json; d=json.load(open('gpurun_out/r04a/bench_plain.json')); print(d['ms_per_step'], d['roofline'])"
