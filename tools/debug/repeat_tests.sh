# repeat a test file under environment settings: tools/debug/repeat_tests.sh N "VAR=VAL ..." [pytest args]
n=$1; envs=$2; shift 2
for i in $(seq 1 $n); do
  env $envs timeout -k 10 300 python -m pytest "$@" -x -q -m gpu > gpurun_out/rep_$i.log 2>&1
  echo "[$envs] run $i rc=$? $(grep -a -o 'Fatal Python error: [A-Za-z ]*\|[0-9]* passed\|[0-9]* failed' gpurun_out/rep_$i.log | head -2 | tr '\n' ' ')"
done
