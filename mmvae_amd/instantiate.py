"""Dependency-free instantiation of the reference's LightningCLI YAML schema (`class_path` / `init_args`).

The reference parses `configs/model/*.yaml` with jsonargparse through LightningCLI (runners/cli.py:36-41); neither is
installed here, and only a small subset is needed: nested `{class_path: a.b.C, init_args: {...}}` objects, lists of
them, and dotted class strings for `activation_fn` (e.g. `torch.nn.ReLU`).  Class paths under `cmmvae.` resolve to
this package's mirror classes (same names, same arguments), so reference YAML files load unchanged.
"""
from __future__ import annotations

import importlib
import os
from typing import Any

import yaml

ALIASES = {"cmmvae": "mmvae_amd"}
_CLASS_VALUED_KEYS = {"activation_fn"}


def resolve(path: str):
    root, _, rest = path.partition(".")
    path = ALIASES.get(root, root) + ("." + rest if rest else "")
    module, _, name = path.rpartition(".")
    if not module:
        raise ValueError(f"not a dotted class path: {path!r}")
    return getattr(importlib.import_module(module), name)


def build(node: Any, key: str = "") -> Any:
    """Recursively turn YAML nodes into objects."""
    if isinstance(node, dict):
        if "class_path" in node:
            cls = resolve(node["class_path"])
            kwargs = {k: build(v, k) for k, v in (node.get("init_args") or {}).items()}
            return cls(**kwargs)
        return {k: build(v, k) for k, v in node.items()}
    if isinstance(node, list):
        return [build(v, key) for v in node]
    if isinstance(node, str) and key in _CLASS_VALUED_KEYS:
        return resolve(node)
    if isinstance(node, str) and "$" in node:  # ${VAR} in path-like values (labels_dir, conditionals_directory)
        return os.path.expandvars(node)
    return node


def load_yaml(path: str) -> Any:
    with open(path) as f:
        return build(yaml.safe_load(f))
