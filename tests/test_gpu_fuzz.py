"""The randomised parity scripts (tests/fuzz_*.py) as short GPU tests: a few seconds each, fixed seeds -- so that they cannot rot and every
run of the suite takes a fresh slice of random shapes through the kernels.  The long runs are recorded in profiles/r05_fuzz_parity.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,args", [("fuzz_parity.py", ["6", "31000", "0.3"]), ("fuzz_slabs.py", ["6", "32000"]),
                                         ("fuzz_terms.py", ["5", "33000"]), ("fuzz_solver.py", ["5", "34000"]), ("fuzz_big.py", ["8", "35000"])])
def test_fuzz_script_runs_clean(script, args):
    # one child process at a time (the GPU box allows few processes on its card)
    r = subprocess.run([sys.executable, os.path.join(HERE, script)] + args, capture_output=True, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-6:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert "cases ok" in tail and "MISMATCH" not in r.stdout
