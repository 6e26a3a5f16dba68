"""Minimal Hydra-style config composition for the filter runner.

The reference composes `config/config.yaml` with Hydra/OmegaConf (`filter/filter.py:259`,
`config/config.yaml:8-11`); neither is a dependency here.  `load_config` reads the same tree layout
(root `defaults:` list -> `<group>/<name>.yaml`), applies `a.b.c=value` overrides like the Hydra CLI
(`README.md:103-106`) and returns attribute-accessible dicts, which is all the path reads
(`cfg.expt.params.num_particles`, `cfg.tdn.render.pen.max`, ...).
"""
from __future__ import annotations

import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")


class Cfg(dict):
    """dict with attribute access (OmegaConf DictConfig stand-in)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def load_config(overrides=(), config_dir: str = CONFIG_DIR, config_name: str = "config") -> Cfg:
    root = yaml.safe_load(open(os.path.join(config_dir, config_name + ".yaml"))) or {}
    groups = {}
    for item in root.pop("defaults", []):
        (g, n), = item.items()
        groups[g] = n
    plain = []
    for ov in overrides:  # group selection first: expt=mcmaster
        k, v = ov.split("=", 1)
        if k in groups and "." not in k:
            groups[k] = v
        else:
            plain.append((k, v))
    cfg = dict(root)
    for g, n in groups.items():
        cfg[g] = yaml.safe_load(open(os.path.join(config_dir, g, n + ".yaml"))) or {}
    for k, v in plain:
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(v)
    return _wrap(cfg)
