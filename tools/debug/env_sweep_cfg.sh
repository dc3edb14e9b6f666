# tools/debug/env_sweep_cfg.sh CONFIG "VAR=VAL,..." ...   ("-" = defaults)
cfgname=$1; shift
for cfg in "$@"; do
  ( if [ "$cfg" != "-" ]; then IFS=','; for kv in $cfg; do export "$kv"; done; unset IFS; fi
    echo -n "$cfgname $cfg  "; timeout -k 10 300 python bench.py --config $cfgname --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_median'],4))" )
done
