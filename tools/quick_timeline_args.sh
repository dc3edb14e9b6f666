# One step of bench.py with arbitrary arguments, launch by launch: tools/quick_timeline_args.sh TAG [bench args...]
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
rm -rf gpurun_out/prof_${tag}_db
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_${tag}_db -o b -- python3 bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/${tag}_under_rocprof.json 2> gpurun_out/${tag}_timeline.err || exit 1
db=$(find gpurun_out/prof_${tag}_db -name "*.db" | head -1)
python3 tools/timeline.py $db > gpurun_out/${tag}_step_timeline.txt
rm -rf gpurun_out/prof_${tag}_db
