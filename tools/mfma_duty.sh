# MFMA utilisation from the hardware counters (run on the GPU box from the repo root): tools/mfma_duty.sh TAG
# One --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, kernel trace only) over the five G-wide GEMMs of the step,
# one over the forward GEMM at half the genes (the counter halves: it counts), -> gpurun_out/TAG_mfma_duty.txt
tag=$1
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > gpurun_out/${tag}_mfma_counters_available.txt
for which in family fwd_half; do
  rm -rf gpurun_out/duty_${tag}_$which
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/duty_${tag}_$which -o p -- python3 tools/roofline_kernel.py $which > /dev/null 2>> gpurun_out/${tag}_duty.err || exit 1
done
python3 tools/mfma_duty.py $tag > gpurun_out/${tag}_mfma_duty.txt
cat gpurun_out/${tag}_mfma_duty.txt
