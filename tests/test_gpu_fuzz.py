"""GPU fuzz sweep under pytest (fixed budget): tools/fuzz_gpu.py draws random instance shapes (RF 1-4 incl. RF changes,
1-12 uneven racks, random weights, band overrides, scrambled starts, every 25th topic large enough for the
2-/1-wavefront and the global-memory K-search paths) and checks, through the C ABI, K-search against its scalar replay
(bit-exact restart states), K-eval and the incremental bookkeeping against the independent numpy verifier, and K-bound
against its replay.  Any mismatch fails the suite."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed0", [0, 7])
def test_fuzz_sweep(seed0):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "150", str(seed0), "25", "2", "25"],
                         capture_output=True, text=True, timeout=1200)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "0 problems" in out.stdout, tail
