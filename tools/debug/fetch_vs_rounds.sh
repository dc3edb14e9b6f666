# FETCH_SIZE of the two TN weight-gradient kernels against the number of rounds of their persistent grid (cap):
# is the excess over the operand bytes the SMALL operand (3 MB of planes), fetched once per XCD L2 and per round?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cap in 256 250 167 125 84; do
  export MMVAE_RK_CAP=$cap
  rm -rf gpurun_out/pmc_cap_$cap
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_cap_$cap -o p -- python3 tools/roofline_kernel.py family > /dev/null 2>&1
  python3 - $cap <<'PY'
import csv, glob, sys
from collections import defaultdict
cap = int(sys.argv[1])
for f in glob.glob(f"gpurun_out/pmc_cap_{cap}/**/*counter_collection.csv", recursive=True):
    per = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "gemm_x3w_kernel<1, 1" in row["Kernel_Name"]:
            k = row["Kernel_Name"]; per[k[k.index("gemm_x3w"):].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in sorted(per.items()):
        rounds = -(-500 // cap)
        big = 61.4 if "256, 160" in k else 41.0
        print(f"cap {cap:3d} ({rounds} rounds of 500 items) {k:50s} FETCH_SIZE x2 = {2*sum(v)/len(v)/1024:6.1f} MB   big operand once {big} MB + 8 XCDs x 3.1 MB x rounds = {big + 8*3.1*rounds:6.1f} MB")
PY
  rm -rf gpurun_out/pmc_cap_$cap
done
