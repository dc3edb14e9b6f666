for i in 1 2; do for c in 128 96 144 160 192 256; do
  MMVAE_PREFETCH=$c python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; l=[json.loads(x) for x in sys.stdin if x.startswith('{')][0]; print('prefetch cap=$c', round(l['ms_per_step'],4), [(k['name'], round(k['us'])) for k in l['roofline']['kernels']][:2])"
done; done
