# tools/debug/cond_up_probe.py under runtime settings that change how a host-to-device copy behind a graph launch is issued
for e in "X=1" "HSA_ENABLE_SDMA=0" "HSA_ENABLE_INTERRUPT=0" "AMD_DIRECT_DISPATCH=0" "GPU_MAX_HW_QUEUES=8" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "== $e"
  env $e timeout -k 10 200 python tools/debug/cond_up_probe.py --parallel 2>&1 | tail -1
done
