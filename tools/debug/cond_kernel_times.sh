# per-kernel averages of the conditional-layer kernels inside the parallel program: tools/debug/cond_kernel_times.sh TAG
tag=$1
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o b -- python3 tools/bench_conditional.py --parallel --categorical > gpurun_out/${tag}_cond.txt 2> gpurun_out/${tag}_cond.err || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("cond_", "layernorm", "jobs_kernel")):
        print(r["Name"].replace("(anonymous namespace)::", "")[:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
tail -1 gpurun_out/${tag}_cond.txt | cut -c1-130
rm -rf gpurun_out/prof_$tag
