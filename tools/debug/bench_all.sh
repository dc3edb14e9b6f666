# every BASELINE configuration + the "next rows" through bench.py (no CPU baseline): gpurun_out/r2_all_<name>.json
for c in c1 c2 c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/r2_all_$c.json 2> gpurun_out/r2_all_$c.err || echo "$c failed"
done
timeout -k 10 300 python bench.py --genes 60530,52437 --no-cpu-baseline > gpurun_out/r2_all_refgenes.json 2> gpurun_out/r2_all_refgenes.err || echo "refgenes failed"
timeout -k 10 300 python bench.py --mode validate --no-cpu-baseline > gpurun_out/r2_all_validate.json 2>/dev/null || echo "validate failed"
timeout -k 10 300 python bench.py --mode predict --no-cpu-baseline > gpurun_out/r2_all_predict.json 2>/dev/null || echo "predict failed"
timeout -k 10 300 python bench.py --input csr --no-cpu-baseline > gpurun_out/r2_all_csr.json 2>/dev/null || echo "csr failed"
for f in gpurun_out/r2_all_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('r2_all_')[1], round(d['ms_per_step'],3), round(d['value']), round(d.get('step_tflops',0),1), (d.get('roofline') or {}).get('frac'))"; done
