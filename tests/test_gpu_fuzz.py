"""Randomised parity sweep (tests/fuzz_parity.py): sizes 1..2200 x 1..400, six kinds of content, both modes, every
option (saturating NMS, the three front-path forms, k_front8's half-strip form and dense path forced on / off / automatic,
run lengths, L2 gradient, BGR / per-channel / 3-channel Mode O input),
batches of 1..3 frames, the fast path's own blur and bit planes in a third of the Mode R cases -- product vs oracle, bit for bit.  The longer runs (thousands of cases) are done by hand with the tool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_random_cases_match_oracle(oracle, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "150", str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 mismatches" in out.stdout
