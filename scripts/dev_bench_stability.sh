for k in 1 2 3; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config2_single_gpu']
print('cfg1 value %.1f M ms %.4f gemm %.4f sclk %.0f frac %.3f | cold %.1f M sclk %.0f | cfg2 extra %.1f M frac %.3f' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms_per_launch']['blockdft_gemm'], r['sclk_mhz'], r['frac'], d['cold_start']['value']/1e6, d['cold_start']['sclk_mhz'], c['value']/1e6, c['roofline_frac']))"
python3 bench.py --steps 10 --warmup 3 --config 2 --no-cpu-baseline --no-extra-config 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('cfg2 headline value %.1f M ms %.4f gemm %.4f sclk %.0f frac %.3f | cold %.1f M sclk %.0f' % (d['value']/1e6, d['ms_per_step'], r['kernel_ms_per_launch']['blockdft_gemm'], r['sclk_mhz'], r['frac'], d['cold_start']['value']/1e6, d['cold_start']['sclk_mhz']))"
done
