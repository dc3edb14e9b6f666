"""Summarise the separate rocprofv3 --pmc passes of tools/roofline_kernel.py collected by tools/collect_profiles.sh:
per counter the value per launch of the roofline kernel (the GEMM with the most launches in the pass).
usage: python tools/pmc_summary.py TAG  (reads gpurun_out/pmc_TAG_<counter>/…/p_counter_collection.csv)"""
import csv
import glob
import sys
from collections import defaultdict

tag = sys.argv[1]
print("pass,kernel,counter,launches,value_per_launch,note")
notes = {
    "FETCH_SIZE": "KiB; gfx950 tallies 128-B read requests at 64 B: x2 for 16-B/lane streaming reads",
    "WRITE_SIZE": "KiB; exact for 16-B/lane stores",
    "SQ_VALU_MFMA_BUSY_CYCLES": "cycles summed over the SIMDs that report; / (GRBM_GUI_ACTIVE x 4 SIMD x 256 CU) = MFMA busy fraction",
    "GRBM_GUI_ACTIVE": "GPU clock cycles of the launch",
    "SQ_INSTS_VALU": "vector ALU instructions (wave-level) of the launch",
    "SQ_INSTS_LDS": "LDS instructions (wave-level) of the launch",
    "SQ_INSTS_VMEM_RD": "vector memory read instructions (wave-level) of the launch",
}
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*")):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    per = defaultdict(lambda: defaultdict(list))  # counter -> kernel -> values
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            per[row["Counter_Name"]][row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for counter, kernels in per.items():
        gemm = {k: v for k, v in kernels.items() if "gemm_x3" in k or "gemm_f32" in k}
        if not gemm:
            continue
        for k in sorted(gemm, key=lambda n: -len(gemm[n])):  # every GEMM kernel of the pass (the family of five)
            v = gemm[k]
            print(f'{d.split("/")[-1]},"{k}",{counter},{len(v)},{sum(v) / len(v):.1f},"{notes.get(counter, "")}"')
