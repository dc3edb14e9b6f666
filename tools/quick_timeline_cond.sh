# One step of the conditional-layer model (tools/bench_conditional.py, engine only) launch by launch:
#   tools/quick_timeline_cond.sh TAG [--parallel]
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
rm -rf gpurun_out/prof_${tag}_db
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_${tag}_db -o b -- python3 tools/bench_conditional.py --engine-only "$@" > gpurun_out/${tag}_under_rocprof.txt 2> gpurun_out/${tag}_timeline.err || exit 1
db=$(find gpurun_out/prof_${tag}_db -name "*.db" | head -1)
python3 tools/timeline.py $db > gpurun_out/${tag}_step_timeline.txt
rm -rf gpurun_out/prof_${tag}_db
