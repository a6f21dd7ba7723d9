"""A process that starts other programs on behalf of the test session.

pytest's own process initialises the GPU (conftest asks torch whether one is visible), and on the GPU pool a process that
has done so must not be the one that execs new programs.  conftest therefore starts THIS helper -- which never imports
torch and never touches the GPU -- before its first GPU call; tests that need fresh rank processes (one process per GPU,
tests/test_hip_two_rank.py) hand it a command line and get back exit code + output.

Protocol: one JSON object per line on stdin {"argv": [...], "env": {...}, "timeout": seconds, "cwd": path} ->
one JSON object per line on stdout {"rc": int, "stdout": str, "stderr": str}.  EOF on stdin ends the helper.
"""
import json
import os
import subprocess
import sys


def main() -> int:
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env") or {})
        for k in req.get("unset") or []:
            env.pop(k, None)
        try:
            p = subprocess.run(req["argv"], env=env, cwd=req.get("cwd"), capture_output=True, text=True, timeout=req.get("timeout", 600))
            rep = {"rc": p.returncode, "stdout": p.stdout[-20000:], "stderr": p.stderr[-20000:]}
        except subprocess.TimeoutExpired as e:
            rep = {"rc": -9, "stdout": (e.stdout or b"").decode("utf-8", "replace")[-20000:] if isinstance(e.stdout, bytes) else (e.stdout or "")[-20000:],
                   "stderr": "timeout"}
        sys.stdout.write(json.dumps(rep) + "\n")
        sys.stdout.flush()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
