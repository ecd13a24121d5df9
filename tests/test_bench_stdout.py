"""bench.py's contract is ONE JSON line on stdout: whatever a library prints to file descriptor 1 after bench.claim_stdout()
(RCCL prints a banner there when a communicator is created) must end up on stderr."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["DR_ROOT"])
import bench
bench.claim_stdout()
os.write(1, b"banner from a C library\n")      # what RCCL does
print("a python print")                          # sys.stdout is file descriptor 1 as well
bench.emit({"metric": "m", "value": 1.5})
"""


def test_stdout_carries_only_the_json_line():
    r = subprocess.run([sys.executable, "-c", _CHILD], env=dict(os.environ, DR_ROOT=ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.5}, r.stdout
    assert "banner from a C library" in r.stderr and "a python print" in r.stderr
