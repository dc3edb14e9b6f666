# Bench line + rocprofv3 kernel stats for the configurations beside the headline (run on the GPU box from the repo root):
# tools/collect_config_profiles.sh TAG  ->  gpurun_out/TAG_bench_line_<cfg>.json, TAG_kernel_stats_<cfg>.csv (c3, c4, c5,
# the reference's gene counts, the conditional model), TAG_configs.jsonl (tools/bench_all_configs.sh)
tag=$1
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
for cfg in c3 c4 c5; do
  python3 bench.py --config $cfg --no-cpu-baseline > gpurun_out/${tag}_bench_line_${cfg}.json 2> gpurun_out/${tag}_bench_${cfg}.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${cfg} -o b -- python3 bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline --no-parity > gpurun_out/${tag}_bench_under_rocprof_${cfg}.json 2>> gpurun_out/${tag}_bench_${cfg}.err || exit 1
  cp $(find gpurun_out/prof_${tag}_${cfg} -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats_${cfg}.csv
  rm -rf gpurun_out/prof_${tag}_${cfg}
done
python3 bench.py --genes 60530,52437 --no-cpu-baseline --no-parity > gpurun_out/${tag}_bench_line_refgenes.json 2> gpurun_out/${tag}_bench_refgenes.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_refgenes -o b -- python3 bench.py --genes 60530,52437 --steps 30 --warmup 6 --no-cpu-baseline --no-parity > /dev/null 2>> gpurun_out/${tag}_bench_refgenes.err || exit 1
cp $(find gpurun_out/prof_${tag}_refgenes -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats_refgenes.csv
rm -rf gpurun_out/prof_${tag}_refgenes
# the conditional model: sequential selection order, then the reference's default ("parallel") with str and with categorical metadata
python3 tools/bench_conditional.py --engine-only 2> gpurun_out/${tag}_conditional.err | tail -1 > gpurun_out/${tag}_conditional.txt || exit 1
python3 tools/bench_conditional.py --engine-only --categorical 2>> gpurun_out/${tag}_conditional.err | tail -1 | sed 's/captured engine     :/captured engine, sequential order, categorical metadata:/' >> gpurun_out/${tag}_conditional.txt || exit 1
python3 tools/bench_conditional.py --parallel 2>> gpurun_out/${tag}_conditional.err | tail -1 >> gpurun_out/${tag}_conditional.txt || exit 1
python3 tools/bench_conditional.py --parallel --categorical 2>> gpurun_out/${tag}_conditional.err | tail -1 >> gpurun_out/${tag}_conditional.txt || exit 1
python3 tools/time_conditional_host.py --parallel 2>> gpurun_out/${tag}_conditional.err | tail -2 >> gpurun_out/${tag}_conditional.txt || exit 1
python3 tools/time_conditional_host.py --parallel --categorical 2>> gpurun_out/${tag}_conditional.err | tail -2 >> gpurun_out/${tag}_conditional.txt || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_cond -o b -- python3 tools/bench_conditional.py --parallel > /dev/null 2>> gpurun_out/${tag}_conditional.err || exit 1
cp $(find gpurun_out/prof_${tag}_cond -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats_conditional.csv
rm -rf gpurun_out/prof_${tag}_cond
bash tools/bench_all_configs.sh > gpurun_out/${tag}_configs.jsonl
cat gpurun_out/${tag}_configs.jsonl
