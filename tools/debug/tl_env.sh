# kernel timeline of one step under the environment given as arguments: tools/debug/tl_env.sh TAG VAR=VAL ...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export "$@"
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_$tag -o tl -- python3 bench.py --no-cpu-baseline --steps 30 > /dev/null 2>gpurun_out/$tag.err
python tools/timeline.py $(find gpurun_out/prof_$tag -name "*.db" | head -1) > gpurun_out/timeline_$tag.txt
rm -rf gpurun_out/prof_$tag
