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
import signal
import subprocess
import sys
import tempfile


def run_group(argv, env, cwd, timeout):
    """Run argv as the leader of its OWN session / process group with its output in temporary files (not pipes: a grandchild
    that survives would keep a pipe open and block the reader for ever).  On timeout the WHOLE group is killed -- the rank
    processes of a torch.distributed.run launch are grandchildren, and they are the ones that hang in a collective -- and
    waited for; rc -9 and the captured tail come back."""
    with tempfile.TemporaryFile("w+b") as out, tempfile.TemporaryFile("w+b") as err:
        p = subprocess.Popen(argv, env=env, cwd=cwd, stdout=out, stderr=err, stdin=subprocess.DEVNULL, start_new_session=True)
        timed_out = False
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            timed_out = True
            for sig, grace in ((signal.SIGTERM, 10), (signal.SIGKILL, 30)):
                try:
                    os.killpg(p.pid, sig)  # (start_new_session: pgid == pid of the leader)
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=grace)
                    break
                except subprocess.TimeoutExpired:
                    continue
            try:  # stragglers of the group that outlived its leader
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            rc = -9

        def tail(f):
            f.seek(0, os.SEEK_END)
            n = f.tell()
            f.seek(max(0, n - 20000))
            return f.read().decode("utf-8", "replace")

        so, se = tail(out), tail(err)
    if timed_out:
        se += f"\n[launcher] timeout after {timeout} s: process group {p.pid} killed"
    return {"rc": rc, "stdout": so, "stderr": se}


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
            rep = run_group(req["argv"], env, req.get("cwd"), req.get("timeout", 600))
        except Exception as e:  # (a bad command line must not take the helper down with it)
            rep = {"rc": -1, "stdout": "", "stderr": f"[launcher] {type(e).__name__}: {e}"}
        sys.stdout.write(json.dumps(rep) + "\n")
        sys.stdout.flush()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
