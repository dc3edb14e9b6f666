"""MFMA utilisation of the step's G-wide GEMMs from SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (tools/mfma_duty.sh).
SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the cycles a SIMD's matrix pipe is busy; GRBM_GUI_ACTIVE sums
the launch's active cycles over the 8 XCDs.  duty = busy / 1024 / (active / 8).  `expected` is the count the launch's
MFMA instructions predict: 6 bf16 MFMAs per fp32 product element block, 32 pipe cycles per 32x32x16 MFMA (32 768 flop), 16
per 16x16x32 (16 384 flop): flop x 6 / 1024 cycles either way.
usage: python tools/mfma_duty.py TAG"""
import csv
import glob
import sys
from collections import defaultdict

tag = sys.argv[1]
B, G, H1 = 512, 20000, 1024
print(f"{'pass':9s} {'kernel':58s} {'launches':>8s} {'MFMA busy cycles':>17s} {'expected':>12s} {'active cyc/XCD':>14s} "
      f"{'MFMA duty':>9s} {'pipe cycles/SIMD':>16s}")
for which, genes in (("family", G), ("fwd_half", G // 2)):
    files = glob.glob(f"gpurun_out/duty_{tag}_{which}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    per = defaultdict(lambda: defaultdict(list))
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    expected = 2.0 * B * genes * H1 * 6 / 1024
    for k, c in per.items():
        if "gemm_x3" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        act = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8
        name = k[k.index("gemm_x3"):].split("(")[0]
        print(f"{which:9s} {name:58s} {len(c['GRBM_GUI_ACTIVE']):8d} {busy:17.0f} {expected:12.0f} {act:14.0f} "
              f"{busy / 1024 / act:9.3f} {busy / 1024:16.0f}")
