# interleaved A/B of the side-branch cap; prints mean and median ms per step
for v in "$@"; do echo -n "SIDE_DW=$v "; MMVAE_SIDE_DW=$v timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'])"; done
