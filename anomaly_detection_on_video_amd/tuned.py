"""Measured (algorithm, split-K) choices per conv shape, produced by tools/tune_convs.py on an
MI355X and committed as tuned/gfx950.json.  Unknown shapes fall back to the library heuristic."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950.json")
_TABLE: Optional[Dict[str, Tuple[int, int]]] = None


def table() -> Dict[str, Tuple[int, int]]:
    global _TABLE
    if _TABLE is None:
        _TABLE = {}
        if os.environ.get("ADV_NO_TUNED") != "1" and os.path.exists(_PATH):
            with open(_PATH) as f:
                _TABLE = {k: (int(v[0]), int(v[1])) for k, v in json.load(f).items()}
            # ADV_ARITH=mixed (opt-in): shapes where a split-bf16 kernel measured faster override the fp32 choice
            mixed = os.path.join(os.path.dirname(_PATH), "gfx950_mixed.json")
            if os.environ.get("ADV_ARITH") == "mixed" and os.path.exists(mixed):
                with open(mixed) as f:
                    _TABLE.update({k: (int(v[0]), int(v[1])) for k, v in json.load(f).items()})
            # ADV_TUNED_OVERLAY=path.json (experiment knob): entries that replace the committed ones for an A/B on one box
            overlay = os.environ.get("ADV_TUNED_OVERLAY", "")
            if overlay:
                with open(overlay) as f:
                    _TABLE.update({k: (int(v[0]), int(v[1])) for k, v in json.load(f).items()})
    return _TABLE


def lookup(key: str, default: Tuple[int, int]) -> Tuple[int, int]:
    algo, splits = table().get(key, default)
    if algo == 192 and os.environ.get("ADV_NO_TSPAN") == "1":  # experiment knob: the T-spanning tiles back on the plain 128x64 tile
        algo = 162
    cap = int(os.environ.get("ADV_MAX_SPLITS", "0"))  # experiment knob: cap the split-K factor of tuned entries
    if cap > 0 and splits > cap:
        splits = cap
    return algo, splits
