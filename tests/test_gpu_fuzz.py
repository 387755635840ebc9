"""Short runs of the two by-hand fuzzers (tests/fuzz_parity.py, tests/fuzz_state.py) as part of the GPU suite: a dozen
random scenes against the oracle (plus table mode, view-mode rays, packing, caller-made tiles, shards, point queries) and a
few dozen random state transitions of one context against fresh contexts."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(script, *args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), *map(str, args)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


def test_random_scenes_against_the_oracle():
    out = _run("fuzz_parity.py", 3, 16)
    assert len(re.findall(r"^case \d+:", out, re.M)) == 16
    assert "FAIL" not in out and "mismatch" not in out, out[-3000:]
    worst = float(re.search(r"worst vs oracle ([0-9.e+-]+)", out).group(1))
    assert worst <= 1e-4


def test_random_state_transitions_against_fresh_contexts():
    out = _run("fuzz_state.py", 5, 40)
    assert len(re.findall(r"^step \d+:", out, re.M)) == 40
    assert "mismatches: 0" in out, out[-3000:]
