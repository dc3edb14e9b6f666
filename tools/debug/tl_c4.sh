cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_c4 -o tl -- python3 bench.py --config c4 --no-cpu-baseline --steps 30 > /dev/null 2>gpurun_out/c4.err
python tools/timeline.py $(find gpurun_out/prof_c4 -name "*.db" | head -1) > gpurun_out/timeline_c4.txt
rm -rf gpurun_out/prof_c4
