#!/bin/bash
# Interleaved comparison of bench.py under several settings on one box: tools/abn.sh ROUNDS "ENV1" "ENV2" ... [-- bench flags]
rounds=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in $(seq $rounds); do
  for e in "${envs[@]}"; do
    ms=$(env $e python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-parity "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "[$e] $ms"
  done
done
