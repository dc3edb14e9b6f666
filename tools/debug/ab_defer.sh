set -e
mkdir -p gpurun_out
for r in 1 2; do
for d in "0 0" "1 0" "1 48" "1 64" "1 96" "1 128" "1 192"; do
  set -- $d
  MMVAE_DEFER_ADAM=$1 MMVAE_DEFER_ADAM_WG=$2 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=[json.loads(x) for x in sys.stdin if x.startswith('{')][0]; print('defer=$1 wg=$2 c2', l['ms_per_step'], [(k['name'], round(k['us'])) for k in l['roofline']['kernels']])"
done; done
