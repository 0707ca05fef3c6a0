for w in 5 50 500 5 50 500; do python3 bench.py --gpus 1 --steps 20 --warmup $w --no-cpu-baseline --no-extra-config 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('warmup', d['warmup'], 'value %.1f M' % (d['value']/1e6), 'ms_per_step', d['ms_per_step'], 'gemm', r['kernel_ms_per_launch']['blockdft_gemm'], 'sclk', r['sclk_mhz'], 'frac', r['frac'])"; done
for k in 20 200 2000; do python3 bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline --no-extra-config 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('steps', d['steps'], 'value %.1f M' % (d['value']/1e6), 'ms_per_step', d['ms_per_step'], 'gemm', r['kernel_ms_per_launch']['blockdft_gemm'], 'sclk', r['sclk_mhz'], 'frac', r['frac'])"; done
