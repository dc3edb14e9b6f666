"""He initialisation of every Linear (reference `cmmvae/modules/base/init.py:4-9`): kaiming_normal_ with
mode="fan_out", nonlinearity="relu" on the weight, zeros on the bias.  Init-time only (host RNG)."""
from typing import Iterator

import torch
import torch.nn as nn


def _linears(root: nn.Module) -> Iterator[nn.Linear]:
    """Every nn.Linear below `root`, in `modules()` order (the order fixes which random numbers each weight draws)."""
    return (layer for layer in root.modules() if isinstance(layer, nn.Linear))


@torch.no_grad()
def he_init_weights(module: nn.Module) -> None:
    """In place; the parameters keep their storage (they may already live in an optimiser arena)."""
    for linear in _linears(module):
        nn.init.kaiming_normal_(linear.weight, mode="fan_out", nonlinearity="relu")
        bias = linear.bias
        if bias is not None:
            bias.zero_()
