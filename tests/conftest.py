import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


_LAUNCHER = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # started BEFORE the first GPU call of this process (pytest_collection_modifyitems below): see tests/_launcher.py
    global _LAUNCHER
    if _LAUNCHER is None and os.path.exists("/dev/kfd"):
        import subprocess

        _LAUNCHER = subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_launcher.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True, bufsize=1)


def pytest_unconfigure(config):
    global _LAUNCHER
    if _LAUNCHER is not None:
        try:
            _LAUNCHER.stdin.close()
            _LAUNCHER.wait(timeout=30)
        except Exception:
            _LAUNCHER.kill()
        _LAUNCHER = None


def launch_fresh(argv, env=None, unset=(), timeout=600, cwd=None):
    """Run `argv` as a fresh program through the GPU-free helper process; -> {"rc", "stdout", "stderr"}."""
    import json

    if _LAUNCHER is None or _LAUNCHER.poll() is not None:
        raise RuntimeError("the launcher helper is not running (no /dev/kfd at session start?)")
    _LAUNCHER.stdin.write(json.dumps({"argv": list(argv), "env": dict(env or {}), "unset": list(unset), "timeout": timeout, "cwd": cwd or REPO}) + "\n")
    _LAUNCHER.stdin.flush()
    line = _LAUNCHER.stdout.readline()
    if not line:
        raise RuntimeError(f"the launcher helper died (exit code {_LAUNCHER.poll()}) while running {list(argv)[:4]}...")
    return json.loads(line)


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) where no GPU is visible, so `-m "not gpu"` and a bare
    `pytest` both work in the CPU-only build container."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_err(a, b):
    """Norm-wise relative error max|a-b| / max|b| used by every parity test."""
    import torch

    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close_elementwise(a, b, rtol=1e-3, afloor=1e-3):
    """Element-wise parity: |a - b| <= rtol * |b| + afloor * rms(b) for EVERY element (the norm-wise `rel_err` lets
    small elements be arbitrarily wrong; this one does not, with an absolute floor scaled to the tensor's own rms so
    that exact zeros / ReLU-clamped entries do not demand infinite relative precision)."""
    import torch

    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    rms = float(b.pow(2).mean().sqrt())
    excess = (a - b).abs() - (rtol * b.abs() + afloor * rms)
    worst = int(excess.argmax())
    assert float(excess.max()) <= 0, (
        f"element {worst}: got {float(a.reshape(-1)[worst])!r}, want {float(b.reshape(-1)[worst])!r} "
        f"(rtol {rtol}, floor {afloor} * rms {rms:.4g})")
