# crash rate of the stress loop per configuration, interleaved, on ONE box
for rep in 1 2 3; do
for e in "MMVAE_REPLAY_FENCE=0" "MMVAE_REPLAY_FENCE=1" "MMVAE_REPLAY_FENCE=0 MMVAE_SIDE_BRANCHES=0" "MMVAE_SIDE_DW=0"; do
  env $e timeout -k 10 200 python tools/debug/graph_crash_stress.py 100 > gpurun_out/stress.log 2>&1; rc=$?
  echo "[$e] rep $rep rc=$rc $(grep -a -o 'cycles [0-9]*\|done [0-9]*' gpurun_out/stress.log | tail -1)"
done; done
