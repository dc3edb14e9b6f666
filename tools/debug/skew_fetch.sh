cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sk in 0 2 4 8; do
  export MMVAE_XW_SKEW=$sk
  rm -rf gpurun_out/pmc_skew_$sk
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_skew_$sk -o p -- python3 tools/roofline_kernel.py family > /dev/null 2>&1
  python3 - $sk <<'PY'
import csv, glob, sys
from collections import defaultdict
sk = sys.argv[1]
for f in glob.glob(f"gpurun_out/pmc_skew_{sk}/**/*counter_collection.csv", recursive=True):
    per = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "gemm_x3w" in row["Kernel_Name"]:
            k = row["Kernel_Name"]; per[k[k.index("gemm_x3w"):].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in sorted(per.items()):
        print(f"skew {sk} {k:55s} FETCH_SIZE x2 = {2*sum(v)/len(v)/1024:8.1f} MB")
for f in glob.glob(f"gpurun_out/pmc_skew_{sk}/**/*kernel_trace.csv", recursive=True):
    per = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "gemm_x3w" in row["Kernel_Name"]:
            k = row["Kernel_Name"]; per[k[k.index("gemm_x3w"):].split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in sorted(per.items()):
        v = sorted(v); print(f"skew {sk} {k:55s} median {v[len(v)//2]:7.1f} us")
PY
  rm -rf gpurun_out/pmc_skew_$sk
done
