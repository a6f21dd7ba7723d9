"""A ~150-line stand-in for the slice of Hydra the reference uses (`run.py:15-32`,
`configs/**`): defaults-list composition, `group=name` / `a.b=c` command-line overrides,
`_target_` instantiation and `_locate`.  hydra / omegaconf are not installed in the target image
and there is no network; PyYAML is.
"""
from __future__ import annotations

import copy
import importlib
import os
import time
from typing import Any, Dict, List, Optional, Sequence

import yaml


class Config(dict):
    """dict with attribute access (the part of omegaconf.DictConfig the reference relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return Config({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _merge(dst: Dict, src: Dict) -> Dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg: Dict, dotted: str, value) -> None:
    parts = dotted.split(".")
    for p in parts[:-1]:
        cfg = cfg.setdefault(p, {})
    cfg[parts[-1]] = value


def _load_yaml(path: str) -> Dict:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _compose_file(config_dir: str, rel: str, choices: Dict[str, str], group_overrides: Dict[str, str]) -> Dict:
    """Load `<config_dir>/<rel>.yaml` and resolve its defaults list (relative to its own group)."""
    raw = _load_yaml(os.path.join(config_dir, rel + ".yaml"))
    defaults = raw.pop("defaults", None)
    if defaults is None:
        return raw
    here = os.path.dirname(rel)
    out: Dict = {}
    entries = list(defaults)
    if "_self_" not in entries:
        entries.append("_self_")  # hydra >= 1.1: the file's own keys override its defaults
    for entry in entries:
        if entry == "_self_":
            _merge(out, raw)
        elif isinstance(entry, str):  # sibling file in the same group, merged in place
            _merge(out, _compose_file(config_dir, os.path.join(here, entry), choices, group_overrides))
        elif isinstance(entry, dict):
            for group, name in entry.items():
                gpath = os.path.join(here, group)
                name = group_overrides.get(gpath.replace(os.sep, "/"), name)
                if name is None:
                    continue
                choices[gpath.replace(os.sep, "/")] = name
                sub = _compose_file(config_dir, os.path.join(gpath, name), choices, group_overrides)
                node = out
                for part in group.split("/"):
                    node = node.setdefault(part, {})
                _merge(node, sub)
        else:
            raise ValueError(f"unsupported defaults entry {entry!r} in {rel}.yaml")
    return out


def _interpolate(node, choices: Dict[str, str]):
    if isinstance(node, dict):
        return {k: _interpolate(v, choices) for k, v in node.items()}
    if isinstance(node, list):
        return [_interpolate(v, choices) for v in node]
    if isinstance(node, str) and "${" in node:
        out, rest = "", node
        while "${" in rest:
            a = rest.index("${")
            b = rest.index("}", a)
            expr = rest[a + 2 : b]
            if expr.startswith("now:"):
                val = time.strftime(expr[4:])
            elif expr.startswith("hydra:runtime.choices."):
                val = choices.get(expr[len("hydra:runtime.choices."):], "")
            else:
                val = "${" + expr + "}"
            out += rest[:a] + val
            rest = rest[b + 1 :]
        return out + rest
    return node


def compose(config_dir: str, config_name: str = "default", overrides: Sequence[str] = ()) -> Config:
    """Hydra-style composition.  Overrides: `group=name` (or `group/sub=name`) swaps a defaults-list
    choice, `a.b.c=value` sets a value (YAML-parsed), `+a.b=value` adds one, `~a.b` deletes one."""
    group_overrides, value_overrides, deletes = {}, [], []
    for ov in overrides:
        if ov.startswith("~"):
            deletes.append(ov[1:])
            continue
        key, _, val = ov.lstrip("+").partition("=")
        if os.path.isdir(os.path.join(config_dir, key)) and not ov.startswith("+"):
            group_overrides[key] = None if val in ("null", "") else val
        else:
            value_overrides.append((key, yaml.safe_load(val)))
    choices: Dict[str, str] = {}
    cfg = _compose_file(config_dir, config_name, choices, group_overrides)
    for key, val in value_overrides:
        _set_path(cfg, key, val)
    for key in deletes:
        parts = key.split(".")
        node = cfg
        for p in parts[:-1]:
            node = node.get(p, {})
        node.pop(parts[-1], None)
    return _wrap(_interpolate(cfg, choices))


def locate(path: str):
    """Import `pkg.mod.attr` (hydra.utils._locate)."""
    parts = path.split(".")
    for cut in range(len(parts) - 1, 0, -1):
        try:
            obj = importlib.import_module(".".join(parts[:cut]))
        except ModuleNotFoundError:
            continue
        for attr in parts[cut:]:
            obj = getattr(obj, attr)
        return obj
    raise ImportError(f"cannot locate {path!r}")


_locate = locate


def instantiate(node: Optional[Dict[str, Any]], *args, **kwargs):
    """hydra.utils.instantiate for `_target_` nodes (recursive), kwargs override node keys."""
    if node is None:
        return None
    if not isinstance(node, dict) or "_target_" not in node:
        raise ValueError("instantiate() needs a mapping with a _target_ key")
    params = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in node.items() if k != "_target_"}
    params.update(kwargs)
    return locate(node["_target_"])(*args, **params)
