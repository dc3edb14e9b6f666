# Extra --pmc passes over the five G-wide GEMMs (tools/roofline_kernel.py family), one counter group per run:
#   tools/pmc_pass.sh TAG "CTR_A CTR_B" "CTR_C" ...   -> gpurun_out/TAG_pmc_extra.csv (kernel, counter, mean per launch)
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
i=0
for pmc in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcx_${tag}_$i
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d gpurun_out/pmcx_${tag}_$i -o p -- python3 tools/roofline_kernel.py family > /dev/null 2>> gpurun_out/${tag}_pmcx.err || echo "pass $i ($pmc) failed" >> gpurun_out/${tag}_pmcx.err
done
python3 - $tag <<'PY' > gpurun_out/${tag}_pmc_extra.csv
import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
print("kernel,counter,launches,mean_per_launch")
for d in sorted(glob.glob(f"gpurun_out/pmcx_{tag}_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "gemm_x3" in row["Kernel_Name"]:
                k = row["Kernel_Name"]
                per[(k[k.index("gemm_x3"):].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(per.items()):
            print(f'"{k}",{c},{len(v)},{sum(v) / len(v):.1f}')
PY
cat gpurun_out/${tag}_pmc_extra.csv
