"""Per-kernel roofline table of the C2 training step from a rocprofv3 kernel trace (rocpd database).

For every kernel family of the step: launches per step, average duration, ALGORITHMIC bytes or flops per launch (from the
shapes of BASELINE config C2: B = 512, G = 20000, H1 = 1024, H2 = 512, V = 256, Z = 128 -- SURVEY 8d), achieved rate and
the fraction of the bound (HBM 8 TB/s; fp32-equivalent matrix rate 2500/6 TFLOP/s for the bf16x3 kernels, 157.3 TFLOP/s
for the exact-f32 kernels).  Small kernels are latency-bound: their fraction is reported for completeness, not as a
target.   usage: python tools/kernel_rooflines.py prof/b_results.db <steps> > profiles/rXX_kernel_rooflines.csv"""
import sqlite3
import sys

B, G, H1, H2, V, Z = 512, 20000, 1024, 512, 256, 128
P_EXP = 2 * G * H1 + 2 * H1 * H2 + G + 3 * H1 + 3 * H2  # active expert's parameters (weights, biases, BN affine)
HBM, X3, F32 = 8000.0, 2500.0 / 6.0, 157.3  # GB/s, TFLOP/s, TFLOP/s
GF = 2.0 * B * G * H1  # one G-wide GEMM

# kernel-name substring -> (label, kind, algorithmic work per launch [bytes or flops])
TABLE = [
    # wave-specialised persistent kernel (round 2): 4 multiplier + 4 stager waves per CU
    ("gemm_x3w_kernel<1; 1; 256; 160", "dW enc-L1 (TN 256x160, bf16x3, grid capped to 185: VAE optimiser beside it)", "x3", GF),
    ("gemm_x3w_kernel<1; 1; 160; 256", "dW dec-L2 (TN 160x256, bf16x3, side branch on 125 CUs beside the core backward chain)",
     "x3", GF),
    ("gemm_x3_kernel<0; 0; 128; 160; 4; 1; true; 1", "dec-L2 fwd + recon epilogue (NT 128x160, bf16x3, 2 x 4-wave kernel)", "x3", GF),
    ("gemm_x3w_kernel<0; 0; 256; 128", "enc-L1 fwd split-K 16 (NT 256x128, bf16x3)", "x3", GF),
    ("gemm_x3w_kernel<0; 1; 256; 128", "dX dec-L2 split-K 16 (NN 256x128, bf16x3)", "x3", GF),
    # (MMVAE_X3W=0: the round-1 kernels)
    ("gemm_x3_kernel<1; 1; 128; 160", "dW enc-L1 (TN 128x160, bf16x3)", "x3", GF),
    ("gemm_x3_kernel<1; 1; 160; 128", "dW dec-L2 (TN 160x128, bf16x3)", "x3", GF),
    ("gemm_x3_kernel<0; 0; 128; 128", "enc-L1 fwd split-K 16 (NT 128x128, bf16x3)", "x3", GF),
    ("gemm_x3_kernel<0; 1; 128; 128", "dX dec-L2 split-K 16 (NN 128x128, bf16x3)", "x3", GF),
    ("adam_step_kernel", "clip + Adam over the flat arenas (expert 41 M + VAE 0.36 M params, 28 B each)", "hbm",
     28.0 * (P_EXP + 0.36e6) / 2),  # two launches per step: averaged
    # since r1c the G-wide weight gradients leave their norm partials in the GEMM epilogue: the pass covers the rest
    ("sqnorm_kernel", "gradient sum of squares of the ranges no GEMM epilogue covers (~1.5 M floats, 3 launches)",
     "hbm", None),
    ("gemm_f32_batch_kernel", "grouped 64x64-tile GEMMs (exact f32): 7 core-layer weight gradients in one grid; the two heads forward; the two heads dX", "f32",
     2.0 * B * (2 * H1 * H2 + 2 * H2 * V + 2 * V * Z + 2 * V * Z)),
    ("fc_bwd_stats_kernel<false>", "column sums / bias gradients (largest: dP 512 x 20000)", "hbm", None),
    ("sum_parts_batch_kernel", "batched fixed-order finishes (bias partials)", "hbm", None),
]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2])
    rows = db.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name").fetchall()
    print("kernel,role,launches_per_step,avg_us,min_us,bound,algorithmic_per_launch,achieved,peak,frac_of_peak")
    for sub, label, kind, work in TABLE:
        for name, calls, avg, mn in rows:
            n = name.replace("(anonymous namespace)::", "").replace(",", ";")
            if sub not in n:
                continue
            if work is None:
                print(f'"{n[:60]}","{label}",{calls / steps:.1f},{avg / 1e3:.1f},{mn / 1e3:.1f},latency/hbm,,,,')
                continue
            if kind == "hbm":
                ach, peak, unit = work / avg, HBM, "GB/s"  # bytes per ns = GB/s
            else:
                ach, peak, unit = work / avg / 1e3, (X3 if kind == "x3" else F32), "TFLOP/s"
            print(f'"{n[:60]}","{label}",{calls / steps:.1f},{avg / 1e3:.1f},{mn / 1e3:.1f},{"hbm" if kind == "hbm" else "mfma"},'
                  f'{work:.4g} {"B" if kind == "hbm" else "FLOP"},{ach:.1f} {unit},{peak:.1f},{ach / peak:.3f}')


if __name__ == "__main__":
    main()
