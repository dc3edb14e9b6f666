# interleaved bench A/B over environment settings: each argument is a comma-separated VAR=VAL list ("-" = defaults)
for rep in 1 2; do
for cfg in "$@"; do
  ( if [ "$cfg" != "-" ]; then IFS=','; for kv in $cfg; do export "$kv"; done; unset IFS; fi
    echo -n "$cfg  "; timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_median'],4))" )
done; done
