"""Dictionary keys shared across the model / trainer surface (reference `cmmvae/constants.py:7-62`).
The values are part of the drop-in boundary: they name logged scalars and prediction files."""
from types import SimpleNamespace

REGISTRY_KEYS = SimpleNamespace(
    LOSS="loss",
    RECON_LOSS="recon_loss",
    KL_LOSS="kl_loss",
    KL_WEIGHT="kl_weight",
    LABELS="labels",
    PX="px",
    QZ="qz",
    PZ="pz",
    QZM="qzm",
    QZV="qzv",
    Z="z",
    Z_STAR="z_star",
    X="x",
    xhat="xhat",
    Y="Y",
    METADATA="metadata",
    EXPERT="expert",
    HUMAN="human",
    MOUSE="mouse",
    ELBO="elbo",
    REGISTRY="registry",
    EXPERT_ID="expert_id",
    ADV_LOSS="adversarial_loss",
    ADV_WEIGHT="adverserial_weight",  # (sic) spelling of the reference key
    UMAP_EMBEDDINGS="umap_embeddings",
    PREDICT_SAMPLES="data",
    FILTER_CATEGORIES=["sex", "dev_stage", "tissue", "cell_type", "assay"],
)
