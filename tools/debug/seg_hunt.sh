# Is this box one where a forked engine program crashes inside hipGraphLaunch?  If so, compare mitigations on the SAME box.
runs() {  # runs N "ENV..." -> echo crash count
  local n=$1 e=$2 c=0
  for i in $(seq 1 $n); do
    env $e timeout -k 10 300 python -m pytest tests/test_step_gpu.py -x -q -m gpu > gpurun_out/hunt.log 2>&1
    if [ $? -eq 139 ]; then c=$((c+1)); fi
  done
  echo "[$e] crashes: $c of $n"
}
runs 6 "MMVAE_TEST_NO_GC=1"
runs 8 "MMVAE_TEST_NO_GC=0"
runs 6 "MMVAE_TEST_NO_GC=1"
runs 8 "MMVAE_TEST_NO_GC=0"
