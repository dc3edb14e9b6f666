# One step of bench.py --config CFG launch by launch: tools/quick_timeline_cfg.sh TAG CFG
tag=$1; cfg=$2
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
rm -rf gpurun_out/prof_${tag}_db
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_${tag}_db -o b -- python3 bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/${tag}_under_rocprof.json 2> gpurun_out/${tag}_timeline.err || exit 1
db=$(find gpurun_out/prof_${tag}_db -name "*.db" | head -1)
python3 tools/timeline.py $db > gpurun_out/${tag}_step_timeline.txt
python3 tools/kernel_rooflines.py $db 37 > gpurun_out/${tag}_kernel_rooflines.csv 2>/dev/null
rm -rf gpurun_out/prof_${tag}_db
