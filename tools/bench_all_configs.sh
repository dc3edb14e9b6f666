# Every row of README's table in one go (one JSON line per configuration): tools/bench_all_configs.sh > gpurun_out/configs.jsonl
b() { python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-parity "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(json.dumps({'args': '$*', 'ms_per_step': round(d['ms_per_step'], 4), 'cells_per_s': round(d['value']), 'step_tflops': round(d.get('step_tflops') or 0, 1), 'workload': d['config'].get('workload')}))"; }
b --config c1
b --config c2
b --config c3
b --config c4
b --config c5
b --genes 60530,52437
b --input csr
b --mode validate
b --mode predict
b --hidden 1000
