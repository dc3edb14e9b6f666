"""He initialisation of every Linear (reference `cmmvae/modules/base/init.py:4-9`): kaiming_normal_ with
mode="fan_out", nonlinearity="relu" on the weight, zeros on the bias.  Init-time only (host RNG)."""
import torch.nn as nn


def he_init_weights(module: nn.Module) -> None:
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
