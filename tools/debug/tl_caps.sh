cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 100 167; do
MMVAE_SIDE_DW=$v rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_tl$v -o tl -- python3 bench.py --no-cpu-baseline --steps 30 > /dev/null 2>gpurun_out/tl$v.err
python tools/timeline.py $(find gpurun_out/prof_tl$v -name "*.db" | head -1) > gpurun_out/r2h_timeline_cap$v.txt
rm -rf gpurun_out/prof_tl$v
done
