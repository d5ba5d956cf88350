cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gather or bf16" 2>&1 | tail -2
for rep in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 100 --warmup 20 --gather-iters 20 --sustain-seconds 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; b=d['roofline_bf16_storage']
print('step ms', round(d['ms_per_step'],4), '| fp32 gather ms', round(r['avg_launch_ms'],4), 'GB/s', round(r['achieved']), 'frac', round(r['frac'],3), 'of fill', round(r['frac_of_measured'],3), round(r['measured_peak']), '| bf16 ms', round(b['avg_launch_ms'],4), 'GB/s', round(b['achieved']), 'frac', round(b['frac'],3), 'of fill', round(b['frac_of_measured'],3))"
done
