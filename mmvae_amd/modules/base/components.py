"""Host-side mirror of the reference's MLP building blocks, executing on the HIP kernels.

Same class names, constructor arguments, attribute / state_dict key names and error behaviour as
`cmmvae/modules/base/components.py` of zdebruine/MMVAE (so `configs/model/*.yaml` class paths and checkpoints
drop in); the arithmetic is libmmvae_hip.so's (via mmvae_amd.functional).  The torch.nn containers here
(nn.Linear, nn.BatchNorm1d, ...) are parameter holders: on device tensors their forward is never called.

Reference lines are cited per class.
"""
from __future__ import annotations

import os
import random
from collections import OrderedDict, defaultdict
from typing import Callable, List, Literal, Optional, Type, Union

import numpy as np
import pandas as pd
import torch
import torch.nn as nn
from torch.distributions import Normal

from ... import backend, cond_tables, ops
from ... import functional as HF
from ... import optim

GradientReversalFunction = HF.GradientReversalFunction  # components.py:879-899


def is_iterable(obj) -> bool:
    """components.py:14-28"""
    try:
        iter(obj)
        return True
    except TypeError:
        return False


# option name -> (type check, optional?)
_OPTIONS = OrderedDict(
    dropout_rate=(lambda v: isinstance(v, float), False),
    use_batch_norm=(lambda v: isinstance(v, bool), False),
    use_layer_norm=(lambda v: isinstance(v, bool), False),
    activation_fn=(lambda v: isinstance(v, type) and issubclass(v, nn.Module), True),
    return_hidden=(lambda v: isinstance(v, bool), False),
)


class FCBlockConfig:
    """Per-layer options of an FCBlock, broadcast and validated (components.py:40-174).

    `layers` [n0, n1, ..., nk] describes k Linear layers n0->n1, ..., n(k-1)->nk; a single size [n] means one
    n->n layer.  Every other option is either one value (applied to all layers) or a list with one entry per layer.
    Invalid input raises ValueError, as in the reference (components.py:112-120, 152-168)."""

    def __init__(
        self,
        layers: List[int],
        dropout_rate: Union[float, List[float]] = 0.0,
        use_batch_norm: Union[bool, List[bool]] = False,
        use_layer_norm: Union[bool, List[bool]] = False,
        return_hidden: Union[bool, List[bool]] = False,
        activation_fn: Union[Optional[Type[nn.Module]], List[Optional[Type[nn.Module]]]] = None,
    ):
        if not isinstance(layers, list):
            raise ValueError(f"layers must be a list found type: {type(layers)}")
        if not all(isinstance(n, int) and not isinstance(n, bool) and n > 0 for n in layers) or not layers:
            raise ValueError("layers must be positive integers")
        self.layers = layers * 2 if len(layers) == 1 else layers
        given = dict(dropout_rate=dropout_rate, use_batch_norm=use_batch_norm, use_layer_norm=use_layer_norm,
                     activation_fn=activation_fn, return_hidden=return_hidden)
        for name, value in given.items():
            setattr(self, name, value if is_iterable(value) else [value] * self.n_layers)
        self.validate()

    @property
    def n_layers(self) -> int:
        if not hasattr(self, "layers"):
            raise RuntimeError("n_layers called before layers initialized")
        return max(len(self.layers) - 1, 1)

    def validate(self) -> None:
        for name, (ok, optional) in _OPTIONS.items():
            values = getattr(self, name)
            if values is None:
                raise ValueError(f"{name} is not optional but value is None")
            try:
                n = len(values)
            except TypeError:
                raise ValueError(f"'{name}' must be a list with one entry per layer")
            if n != self.n_layers:
                raise ValueError(f"Length of '{name}' must match the length of 'layers':{n} != {self.n_layers}")
            for v in values:
                if v is None and optional:
                    continue
                if v is None or not ok(v):
                    raise ValueError(f"All elements in '{name}' have the wrong type (got {v!r})")


class ConcatBlockConfig(FCBlockConfig):
    """Options of the extra decoder input layer used by CLVAE's "parallel" mode (components.py:177-190).
    Holds single values; CLVAE prepends them to the decoder config's lists (clvae.py:55-79)."""

    def __init__(self, dropout_rate: float = 0.0, use_batch_norm: bool = False, use_layer_norm: bool = False,
                 return_hidden: bool = False, activation_fn: Optional[Type[nn.Module]] = None):
        self.dropout_rate = dropout_rate
        self.use_batch_norm = use_batch_norm
        self.use_layer_norm = use_layer_norm
        self.return_hidden = return_hidden
        self.activation_fn = activation_fn


def _draw_keep_mask(shape, p: float, device) -> torch.Tensor:
    """Dropout keep mask (P(keep) = 1 - p) from the library's Philox stream (k15); replaces nn.Dropout's draw."""
    from ... import ops, rng

    return ops.philox_keep_mask(tuple(shape), p, rng.state(device), stream_id=rng.STREAM_DROPOUT)


class FCBlock(nn.Module):
    """Stack of Linear -> [BatchNorm1d(momentum=0.01, eps=0.001)] -> [LayerNorm(no affine)] -> [activation] ->
    [Dropout] layers (components.py:193-314).  Sub-module names (`fc_layers.{i}.lin|bn|ln|af|dr`) match the
    reference so checkpoints interchange.

    Device tensors run one fused HIP layer per iteration (GEMM + column kernel; see functional.FCLayerFn)."""

    def __init__(self, config: FCBlockConfig):
        super().__init__()
        config.validate()
        self.config = config
        blocks = []
        for i, (n_in, n_out) in enumerate(zip(config.layers[:-1], config.layers[1:])):
            parts = OrderedDict()
            parts["lin"] = nn.Linear(n_in, n_out)
            if config.use_batch_norm[i]:
                parts["bn"] = nn.BatchNorm1d(n_out, momentum=0.01, eps=0.001)
            if config.use_layer_norm[i]:
                parts["ln"] = nn.LayerNorm(n_out, elementwise_affine=False)
            act = config.activation_fn[i]
            if act is not None:
                parts["af"] = act(dim=1) if issubclass(act, nn.Softmax) else act()
            if config.dropout_rate[i] > 0:
                parts["dr"] = nn.Dropout(p=config.dropout_rate[i])
            blocks.append(nn.Sequential(parts))
        self.fc_layers = nn.Sequential(*blocks)
        # parity mode: {layer index: uint8 keep mask [B, n_out]} consumed by the next forward
        self.explicit_masks: Optional[dict] = None

    @property
    def input_dim(self) -> int:
        return self.config.layers[0]

    @property
    def output_dim(self) -> int:
        return self.config.layers[-1]

    @property
    def can_bypass(self) -> bool:
        return not any(self.config.return_hidden)

    # ---- HIP execution of one layer
    def _hip_layer(self, i: int, layer: nn.Sequential, x: torch.Tensor):
        lin = layer.lin
        bn = getattr(layer, "bn", None)
        af = getattr(layer, "af", None)
        dr = getattr(layer, "dr", None)
        has_ln = hasattr(layer, "ln")
        relu = isinstance(af, nn.ReLU)
        fuse_act = relu and not has_ln
        bn_state = None
        if bn is not None:
            bn_state = dict(running_mean=bn.running_mean, running_var=bn.running_var,
                            num_batches_tracked=bn.num_batches_tracked, momentum=bn.momentum, eps=bn.eps)
        mask, p = None, 0.0
        fuse_drop = dr is not None and (af is None or fuse_act) and not has_ln
        if fuse_drop and self.training:
            p = dr.p
            if self.explicit_masks is not None and i in self.explicit_masks:
                mask = self.explicit_masks[i]
            else:
                mask = _draw_keep_mask((x.shape[0], lin.out_features), p, x.device)
        d, a = HF.fc_layer(x, lin.weight, lin.bias, gamma=bn.weight if bn is not None else None,
                           beta=bn.bias if bn is not None else None, bn_state=bn_state,
                           training=self.training if bn is not None else True, relu=fuse_act, keep_mask=mask,
                           dropout_p=p)
        if has_ln:
            # Linear -> [BN] -> LayerNorm -> [ReLU] -> [Dropout] (components.py:279-288; configs/model/configV3.yaml:25-36):
            # the row normalisation sits between the column kernel and the activation, so ReLU and dropout run as one
            # more pass of the layer-tail kernels behind it
            d = HF.LayerNormFn.apply(d, layer.ln.eps)
            a = d
            tail_mask, tail_p = None, 0.0
            if dr is not None and self.training and (af is None or relu):
                tail_p = dr.p
                if self.explicit_masks is not None and i in self.explicit_masks:
                    tail_mask = self.explicit_masks[i]
                else:
                    tail_mask = _draw_keep_mask(d.shape, dr.p, d.device)
            if relu or tail_mask is not None:
                d, a = HF.layer_tail(d, relu=relu, keep_mask=tail_mask, dropout_p=tail_p)
            if af is None or relu:
                return d, a
        if af is not None and not fuse_act:
            # activations other than ReLU (Softmax / Sigmoid heads of stale configs) are outside the HIP hot path
            d = af(d)
            a = d
        if dr is not None and not fuse_drop:
            if self.training:
                if self.explicit_masks is not None and i in self.explicit_masks:
                    m = self.explicit_masks[i]
                else:
                    m = _draw_keep_mask(d.shape, dr.p, d.device)
                d, _ = HF.layer_tail(d, relu=False, keep_mask=m, dropout_p=dr.p)
        return d, a

    def forward(self, x: torch.Tensor):
        """Returns the output, or (output, hidden) when any layer has return_hidden (components.py:292-314):
        hidden holds the tensor right after the activation ("af"), before dropout."""
        if x.layout == torch.sparse_csr:  # CSR batches (datapipes with return_dense: false) are densified on entry
            x = backend.to_dense(x)
        if not backend.on_hip(x):
            return self._forward_cpu_plumbing(x)
        hidden = []
        for i, layer in enumerate(self.fc_layers):
            x, a = self._hip_layer(i, layer, x)
            if self.config.return_hidden[i] and hasattr(layer, "af"):
                hidden.append(a)
        if self.can_bypass:
            return x
        return x, hidden

    def _forward_cpu_plumbing(self, x: torch.Tensor):
        """Caller-enabled CPU plumbing (backend.cpu_plumbing): plain torch modules, for host-logic tests."""
        if self.can_bypass and self.explicit_masks is None:
            return self.fc_layers(x)
        hidden = []
        for i, layer in enumerate(self.fc_layers):
            for name, sub in layer.named_children():
                if name == "dr" and self.training and self.explicit_masks is not None and i in self.explicit_masks:
                    x = x * self.explicit_masks[i].to(x.dtype) / (1.0 - sub.p)
                else:
                    x = sub(x)
                if name == "af" and self.config.return_hidden[i]:
                    hidden.append(x)
        return x if self.can_bypass else (x, hidden)


class ConditionalLayer(nn.Module):
    """One FCBlock per unique value of a metadata column; each sample goes through the block of its own condition
    (components.py:317-413).  Unique values come from a header-less csv; '.' in keys becomes '_' (:353-363)."""

    def __init__(self, batch_key: str, conditions_path: str, fc_block_config: FCBlockConfig):
        super().__init__()
        self.batch_key = batch_key
        values = pd.read_csv(conditions_path, header=None)[0]
        self.conditions = nn.ModuleDict({self.format_condition_key(v): FCBlock(fc_block_config) for v in values})

    def format_condition_key(self, condition: str) -> str:
        return condition.replace(".", "_")

    # ---- grouped HIP path (one launch per layer instead of a Python loop over the conditions of the batch)
    def _bank(self):
        """Offsets of every condition's weight / bias in the optimiser arena they live in, or None when the blocks are
        not single Linears (+ LayerNorm) or do not (yet) live in one arena (no optimiser configured: eval-only use)."""
        bank = getattr(self, "_bank_cache", None)
        if bank is not None and optim.arena_of(bank["probe"]) is not None and bank["probe"].data_ptr() == bank["probe_ptr"]:
            return bank
        blocks = list(self.conditions.values())
        lins = []
        for blk in blocks:
            if len(blk.fc_layers) != 1 or any(n not in ("lin", "ln") for n, _ in blk.fc_layers[0].named_children()):
                return None
            lins.append(blk.fc_layers[0].lin)
        hits = [(optim.arena_of(l.weight), optim.arena_of(l.bias)) for l in lins]
        if any(h[0] is None or h[1] is None for h in hits):
            return None
        opt = hits[0][0][0]
        if any(h[0][0] is not opt or h[1][0] is not opt for h in hits) or not opt._hip:
            return None
        a = opt.arena
        w_idx = np.array([h[0][1] for h in hits], dtype=np.int64)
        b_idx = np.array([h[1][1] for h in hits], dtype=np.int64)
        dev = a.device
        bank = dict(opt=opt, w_idx=w_idx, b_idx=b_idx, n_in=lins[0].in_features, n_out=lins[0].out_features,
                    w_off=torch.tensor([a.offsets[i] for i in w_idx], dtype=torch.int64, device=dev),
                    b_off=torch.tensor([a.offsets[i] for i in b_idx], dtype=torch.int64, device=dev),
                    index={k: i for i, k in enumerate(self.conditions.keys())}, raw_index={},
                    ln_eps=(blocks[0].fc_layers[0].ln.eps if hasattr(blocks[0].fc_layers[0], "ln") else None),
                    probe=lins[0].weight, probe_ptr=lins[0].weight.data_ptr())
        opt.managed.update(int(i) for i in w_idx)  # gradients of these never arrive as autograd .grad tensors
        opt.managed.update(int(i) for i in b_idx)
        self._bank_cache = bank
        return bank

    def _forward_grouped(self, bank, x: torch.Tensor, column: pd.Series) -> torch.Tensor:
        # raw metadata value -> condition index, through a cache (formatting + ModuleDict lookup once per distinct value)
        raw_index = bank["raw_index"]
        values = column.tolist()
        try:
            cond = np.fromiter((raw_index[v] for v in values), dtype=np.int32, count=len(values))
        except KeyError:
            for v in set(values) - raw_index.keys():
                raw_index[v] = bank["index"][self.format_condition_key(str(v))]  # KeyError: unknown condition, as before
            cond = np.fromiter((raw_index[v] for v in values), dtype=np.int32, count=len(values))
        t = cond_tables.group_tables(cond)
        present = t["present"]
        present_params = np.concatenate([bank["w_idx"][present], bank["b_idx"][present]])
        y = HF.CondLinearFn.apply(x, bank, ops.cond_tables_to_device(t, x.device), present_params)
        if bank["ln_eps"] is not None:
            y = HF.LayerNormFn.apply(y, bank["ln_eps"])
        return y

    def forward(self, x: torch.Tensor, metadata: pd.DataFrame, condition: Optional[str] = None):
        if condition:
            return self.conditions[self.format_condition_key(condition)](x)
        if x.is_cuda and os.environ.get("MMVAE_COND_GROUPED", "1") != "0":
            bank = self._bank()
            if bank is not None:
                return self._forward_grouped(bank, x, metadata[self.batch_key])
        keys = metadata[self.batch_key].astype(str).apply(self.format_condition_key).tolist()
        groups: "OrderedDict[str, list]" = OrderedDict()
        for row, key in enumerate(keys):
            groups.setdefault(key, []).append(row)
        out = torch.empty_like(x)
        for key, rows in groups.items():
            idx = torch.tensor(rows, device=x.device)
            out = out.index_copy(0, idx, self.conditions[key](x.index_select(0, idx)))
        return out


def _is_valid_file(fname: str, batch_key: str) -> bool:
    return fname == f"unique_expression_{batch_key}.csv"


def collect_species_files(directory, batch_keys, species_files=None,
                          is_valid_file: Optional[Callable[[str, str], bool]] = None):
    """Map {"shared": {key: path}, species: {key: path}} from a directory tree (components.py:420-464): keys found
    under `shared/` win; species entries only list keys that are not shared; species with nothing are omitted."""
    is_valid_file = is_valid_file or _is_valid_file
    species_files = {} if species_files is None else species_files

    def scan(folder, skip=()):
        found = {}
        for fname in os.listdir(folder):
            path = os.path.join(folder, fname)
            if not os.path.isfile(path):
                continue
            for key in batch_keys:
                if is_valid_file(fname, key):
                    if key not in skip:
                        found[key] = path
                    break
        return found

    shared_dir = os.path.join(directory, "shared")
    shared = scan(shared_dir) if os.path.isdir(shared_dir) else {}
    species_files["shared"] = shared
    for entry in os.listdir(directory):
        path = os.path.join(directory, entry)
        if entry == "shared" or not os.path.isdir(path):
            continue
        own = scan(path, skip=shared)
        if own:
            species_files[entry] = own
    print(f"Collected species files {species_files}")
    return species_files


class ConditionalLayers(nn.Module):
    """All conditional layers of a CLVAE, applied in a fixed, shuffled or "parallel" (concatenated) order
    (components.py:467-631)."""

    def __init__(self, directory: str, conditionals: list, fc_block_config: FCBlockConfig,
                 selection_order: Optional[list] = None):
        super().__init__()
        if not os.path.exists(directory):
            raise FileNotFoundError(
                f"Could not intialize the conditional layers either due to the directory not existing yet\n{directory}")
        without_species = [c for c in conditionals if c != "species"]
        paths = collect_species_files(directory, without_species)
        if "species" in conditionals:
            # the reference removes "species" from the caller's list and appends it again (components.py:509-511): the
            # list that "parallel" / unordered selection shuffles -- and that fixes the concatenation order -- has it last
            conditionals = without_species + ["species"]
        self.shared_conditionals = list(paths["shared"].keys())
        self.is_parallel = bool(selection_order) and selection_order[0] == "parallel"
        self.shuffle_selection_order = False
        if not selection_order or self.is_parallel:
            selection_order = conditionals
            self.shuffle_selection_order = True
        layers = {key: ConditionalLayer(key, path, fc_block_config) for key, path in paths["shared"].items()}
        per_species: dict = {}
        for species, files in paths.items():
            if species == "shared":
                continue
            for key, path in files.items():
                per_species.setdefault(key, {})[species] = path
        for key, by_species in per_species.items():
            if key in layers:
                raise RuntimeError(f"batch_key '{key}' is shared but attempted to make species specific")
            layers[key] = nn.ModuleDict({s: ConditionalLayer(key, p, fc_block_config) for s, p in by_species.items()})
        if "species" in conditionals:
            layers["species"] = nn.ModuleDict({s: FCBlock(fc_block_config) for s in paths if s != "shared"})
        self.layers = nn.ModuleDict(layers)
        self.selection_order = selection_order

    def forward(self, x: torch.Tensor, metadata: pd.DataFrame, species: Optional[str] = None):
        order = self.selection_order
        if self.shuffle_selection_order:
            order = random.sample(order, len(order))
        outs = []
        for key in order:
            layer = self.layers[key]
            if isinstance(layer, nn.ModuleDict):
                if species is None:
                    raise RuntimeError(
                        f"'species' must be set to access non-shared conditional layer for batch_key '{key}'")
                layer = layer[species]
            y = layer(x, metadata) if isinstance(layer, ConditionalLayer) else layer(x)
            if self.is_parallel:
                outs.append(y)
            else:
                x = y
        return torch.cat(outs, dim=1) if outs else x


class Adversarial(nn.Module):
    """Discriminator: encoder FCBlock + one Linear head per metadata condition (components.py:638-674).
    `Adversarial.labels[condition][value] -> class index` is class-level, shared by all instances (:642,657-659);
    class counts come from `<labels_dir>/human/unique_expression_<condition>.csv`."""

    labels: dict = defaultdict(dict)

    def __init__(self, encoder: FCBlockConfig, heads: FCBlockConfig, conditions: list, labels_dir: str):
        super().__init__()
        self.encoder = FCBlock(encoder)
        head_blocks = {}
        for condition in conditions:
            df = pd.read_csv(os.path.join(labels_dir, f"human/unique_expression_{condition}.csv"), header=None)
            if condition not in Adversarial.labels:
                for idx, value in enumerate(df[0]):
                    Adversarial.labels[condition][value] = idx
            heads.layers = [self.encoder.output_dim, len(df)]
            head_blocks[condition] = FCBlock(heads)
        self.heads = nn.ModuleDict(head_blocks)

    def forward(self, x: torch.Tensor):
        enc = self.encoder(x)
        return {condition: head(enc) for condition, head in self.heads.items()}


class Encoder(nn.Module):
    """VAE encoder: FCBlock, then mean / variance Linear heads, then one reparameterised sample
    (components.py:676-809).  v = exp(var_encoder(q)) + var_eps; dist = Normal(mean, sqrt(v)); z = dist.rsample().

    On device the two heads and the sample are HIP kernels (two GEMMs + the fused reparam/KL kernel); the KL row sums
    and the Mean/Variance statistics computed there are attached to the returned distribution
    (`dist._mmvae = {...}`) so that BaseVAE.elbo does not recompute them."""

    def __init__(self, latent_dim: int, fc_block_config: FCBlockConfig,
                 distribution: Union[Literal["ln"], Literal["normal"]] = "normal", return_dist: bool = False,
                 hidden_z: bool = False, var_eps: float = 1e-4, n_samples: int = 1, elbo_mode: str = "analytic"):
        """`n_samples`, `elbo_mode`: arguments of the K-sample extension (not in the reference; the defaults are the
        reference's single-sample ELBO), so that a YAML file can carry BASELINE configs 3 and 5."""
        super().__init__()
        self.fc = FCBlock(fc_block_config)
        n_hidden = fc_block_config.layers[-1]
        self.mean_encoder = nn.Linear(n_hidden, latent_dim)
        self.var_encoder = nn.Linear(n_hidden, latent_dim)
        self.z_transformation = nn.Softmax(dim=-1) if distribution == "ln" else _identity
        self.var_eps = var_eps
        self.return_dist = return_dist
        self.hidden_z = hidden_z
        self.explicit_eps: Optional[torch.Tensor] = None  # parity mode: noise consumed by the next forward
        self.n_samples: int = int(n_samples)  # K of the K-sample extension (1 = the reference)
        # "analytic": loss = recon + kl_weight * analytic KL (the reference, vae.py:136-152; with K samples the
        # reconstruction term is their log-mean-exp).  "iwae": opt-in full importance-weighted objective:
        # the sampled log q(z) - log p(z) sits inside the log-mean-exp; runs in the captured engine only.
        if elbo_mode not in ("analytic", "iwae"):
            raise ValueError(f"elbo_mode must be 'analytic' or 'iwae', got {elbo_mode!r}")
        self.elbo_mode: str = elbo_mode

    @property
    def n_layers(self) -> int:
        return self.fc.config.n_layers

    def encode(self, x: torch.Tensor):
        return self.fc(x)

    def forward(self, x: torch.Tensor):
        encoded = self.encode(x)
        q, hidden = encoded if isinstance(encoded, tuple) else (encoded, [])
        if backend.on_hip(q):
            q_m, _ = HF.fc_layer(q, self.mean_encoder.weight, self.mean_encoder.bias)
            a_raw, _ = HF.fc_layer(q, self.var_encoder.weight, self.var_encoder.bias)
            eps = self.explicit_eps
            if eps is None:
                shape = (q.shape[0], q_m.shape[1]) if self.n_samples == 1 else (self.n_samples, q.shape[0], q_m.shape[1])
                from ... import ops, rng

                eps = ops.philox_normal(shape, rng.state(q.device), stream_id=rng.STREAM_NORMAL)  # replaces rsample()'s draw
            z, std, kl_sum, stat = HF.ReparamKLFn.apply(q_m, a_raw, eps, self.var_eps)
            dist = Normal(q_m, std, validate_args=False)
            dist._mmvae = {"kl_sum": kl_sum, "stat_row": stat, "batch": q.shape[0]}
            q_v = None
        else:
            q_m = self.mean_encoder(q)
            q_v = torch.exp(self.var_encoder(q)) + self.var_eps
            dist = Normal(q_m, q_v.sqrt())
            if self.explicit_eps is not None:
                z = dist.loc + self.explicit_eps * dist.scale
            else:
                z = dist.rsample()
        latent = self.z_transformation(z)
        if self.hidden_z:
            hidden.append(latent if latent.dim() == 2 else latent[0])
        if self.return_dist:
            return dist, latent, hidden
        if q_v is None:
            q_v = dist.variance
        return q_m, q_v, latent, hidden


def _identity(x):
    return x


class Expert(nn.Module):
    """Modality-specific encoder / decoder FCBlocks (components.py:812-857)."""

    def __init__(self, id: str, encoder_config: FCBlockConfig, decoder_config: FCBlockConfig):
        super().__init__()
        self.id = id
        self.encoder = FCBlock(encoder_config)
        self.decoder = FCBlock(decoder_config)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("an Expert is used through encode() / decode(), not forward()")

    def encode(self, x: torch.Tensor):
        return self.encoder(x)

    def decode(self, x: torch.Tensor):
        return self.decoder(x)


class Experts(nn.ModuleDict):
    """{expert.id: Expert}; `labels` maps ids to their position (components.py:860-876)."""

    def __init__(self, experts: List[Expert]):
        super().__init__({e.id: e for e in experts})
        self.labels = {key: i for i, key in enumerate(self.keys())}
