# The per-round profile set (run on the GPU box from the repo root): tools/collect_profiles.sh TAG
# -> gpurun_out/TAG_*: kernel stats of bench.py under rocprofv3, step timeline, per-kernel rooflines, PMC passes of the
# roofline kernel (each counter set in its own run), the plain bench line of the same box.
tag=$1
root=$PWD
steps=76   # 16 set-up + 10 warm-up + 50 timed
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
# counters first: bench.py reads roofline.traffic from the tracked summary of THIS run
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS"; do
  name=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d gpurun_out/pmc_${tag}_$name -o p -- python3 tools/roofline_kernel.py > /dev/null 2>> gpurun_out/${tag}_bench.err || exit 1
done
python3 tools/pmc_summary.py $tag > gpurun_out/${tag}_pmc_family.csv
cp gpurun_out/${tag}_pmc_family.csv profiles/${tag}_pmc_family.csv
python3 bench.py --steps 50 --warmup 10 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_csv -o b -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2>> gpurun_out/${tag}_bench.err || exit 1
cp $(find gpurun_out/prof_${tag}_csv -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
rm -rf gpurun_out/prof_${tag}_csv
rocprofv3 --kernel-trace --output-format rocpd -d gpurun_out/prof_${tag}_db -o b -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>> gpurun_out/${tag}_bench.err || exit 1
db=$(find gpurun_out/prof_${tag}_db -name "*.db" | head -1)
python3 tools/timeline.py $db > gpurun_out/${tag}_step_timeline.txt
python3 tools/kernel_rooflines.py $db $steps > gpurun_out/${tag}_kernel_rooflines.csv
rm -rf gpurun_out/prof_${tag}_db
cut -c1-300 gpurun_out/${tag}_bench_line.json
tail -3 gpurun_out/${tag}_step_timeline.txt
cat gpurun_out/${tag}_pmc_family.csv
