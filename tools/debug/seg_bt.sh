for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  LD_PRELOAD=$PWD/tools/debug/segv_bt.so timeout -k 10 300 python -m pytest tests/test_step_gpu.py -x -q -m gpu -p no:faulthandler > gpurun_out/bt_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc"
  if [ $rc -eq 139 ]; then break; fi
done
