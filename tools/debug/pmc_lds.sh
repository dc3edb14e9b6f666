cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d gpurun_out/pmc_lds_$i -o p -- python3 tools/roofline_kernel.py > /dev/null 2>gpurun_out/pmc_lds_$i.err || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for d in sorted(glob.glob("gpurun_out/pmc_lds_*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    per = defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "gemm_x3w" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        print(d.split("/")[-1], k, len(v), sum(v) / len(v))
PY
