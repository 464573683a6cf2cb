"""cli/kao-cli: same reassignment JSON in/out as kafka-reassign-partitions (README.md:52-78)."""
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, have_gpu, load_golden

CLI = os.path.join(ROOT, "cli", "kao-cli")
ARGS = ["--current", os.path.join(GOLDEN, "readme_current.json"), "--broker-list", ",".join(str(b) for b in range(19)),
        "--racks", os.path.join(GOLDEN, "readme_racks.json")]


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cli")], stdout=subprocess.DEVNULL)


def test_cli_usage_errors():
    _build()
    assert subprocess.run([CLI], capture_output=True).returncode == 2
    assert subprocess.run([CLI, "--bogus"], capture_output=True).returncode == 2
    r = subprocess.run([CLI, "--current", "/nonexistent.json", "--broker-list", "0,1", "--racks", "0:a,1:b"], capture_output=True)
    assert r.returncode == 1 and b"cannot open" in r.stderr


@pytest.mark.skipif(have_gpu(), reason="checks the no-device failure mode")
def test_cli_fails_loudly_without_gpu():
    _build()
    r = subprocess.run([CLI] + ARGS, capture_output=True)
    assert r.returncode == 1 and b"no usable HIP device" in r.stderr and r.stdout == b""


@pytest.mark.gpu
def test_cli_readme_example():
    """README.md:52-63 in -> only partition 1 changes, to [8,1] (README.md:88)."""
    _build()
    r = subprocess.run([CLI] + ARGS + ["--report"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out == load_golden("kat1.json")["expected_json"]
    assert b"status=OPTIMAL_PROVEN objective=58 bound=58 replica_moves=1 leader_changes=0" in r.stderr
    # CSV racks on the command line give the same answer
    racks = ",".join(f"{b}:{'a' if b % 2 == 0 else 'b'}" for b in range(20))
    r2 = subprocess.run([CLI] + ARGS[:4] + ["--racks", racks], capture_output=True, timeout=120)
    assert json.loads(r2.stdout) == out
