"""Host-pointer argument fuzz of the C-ABI (include/emplanner.h), in a child process so that a crash is a test failure
with its output instead of a dead pytest: tests/abi_fuzz_child.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hostile_arguments_return_errors_and_leave_the_context_usable():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_fuzz_child.py")], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    tail = (run.stdout[-3000:] + "\n" + run.stderr[-3000:])
    assert run.returncode == 0, f"the fuzz child died with {run.returncode}:\n{tail}"
    assert "ABI-FUZZ-OK" in run.stdout, tail
    last = run.stdout.strip().splitlines()[-1].split()
    assert int(last[2]) >= 60 and int(last[4]) >= 45, last  # probes, of which rejected with an error
