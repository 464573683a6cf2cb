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


def test_cli_emits_lp_solve_text(tmp_path, ko):
    """--emit-lp (host only, no GPU): the generated model as lp_solve LP text (README.md:144-185), byte-identical
    to the oracle's writer -- whose text re-solves to the README answer (test_oracle.test_lp_text_roundtrip)."""
    _build()
    prefix = str(tmp_path / "m")
    r = subprocess.run([CLI] + ARGS + ["--emit-lp", prefix, "--lp-only"], capture_output=True)
    assert r.returncode == 0 and r.stdout == b"", r.stderr
    txt = open(prefix + "1.lp").read()
    assert txt == ko.write_lp(ko.readme_example())
    assert "max: 1 t1b0p3 + 4 t1b0p3_l" in txt and txt.rstrip().endswith("t1b18p9, t1b18p9_l;")
    # two topics, an RF increase (2 -> 3), racks given as CSV
    cur = {"version": 1, "partitions": [
        {"topic": "a", "partition": 0, "replicas": [0, 1]}, {"topic": "a", "partition": 1, "replicas": [2, 9]},
        {"topic": "b", "partition": 7, "replicas": [3, 0]}]}
    cur_path = tmp_path / "cur.json"
    cur_path.write_text(json.dumps(cur))
    racks = {b: "r%d" % (b % 2) for b in range(10)}
    r = subprocess.run([CLI, "--current", str(cur_path), "--broker-list", "0,1,2,3,4", "--racks",
                        ",".join(f"{b}:{v}" for b, v in racks.items()), "--rf", "3", "--emit-lp", prefix, "--lp-only"],
                       capture_output=True)
    assert r.returncode == 0, r.stderr
    topics = ko.topics_from_json(cur, [0, 1, 2, 3, 4], racks, rf=3)
    for i, t in enumerate(topics):
        assert open(f"{prefix}{i + 1}.lp").read() == ko.write_lp(t, t_index=i + 1)


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


@pytest.mark.gpu
def test_python_cli_matches_cpp_cli():
    import sys
    _build()
    r1 = subprocess.run([CLI] + ARGS, capture_output=True, timeout=120)
    r2 = subprocess.run([sys.executable, "-m", "kafka_assignment_optimizer_amd.cli"] + ARGS, capture_output=True, timeout=120, cwd=ROOT)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr, r2.stderr)
    assert json.loads(r1.stdout) == json.loads(r2.stdout) == load_golden("kat1.json")["expected_json"]
