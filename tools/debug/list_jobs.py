"""Debug aid: the reductions / grouped GEMMs the C2 training plan queues (sizes), printed while the plan is built."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmvae_amd import synthetic, engine as E

orig_flush = E._Plan._flush_sums
def flush(self):
    for j in self._sum_jobs:
        print(f"  sum job: parts {j.n_parts:3d} rows {j.rows:6d} cols {j.cols:6d} part_stride {j.part_stride} ld_src {j.ld_src} flags {j.flags}")
    print(f"  -> flush of {len(self._sum_jobs)} sum jobs, {len(self._gemm_jobs)} gemm jobs")
    for g in self._gemm_jobs:
        print(f"  gemm job: layout {g.layout} M {g.M} N {g.N} K {g.K}")
    return orig_flush(self)
E._Plan._flush_sums = flush
cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
model = synthetic.build_model(cfg["experts"], adversarial=cfg["adversarial"], n_samples=cfg["K"]).cuda()
model.train(); model.trainer.set_stage("training")
x = synthetic.synthetic_counts(cfg["batch"], 20000, device="cuda")
model.training_step((x, synthetic.synthetic_metadata(cfg["batch"]), "human"), 0)
torch.cuda.synchronize()
