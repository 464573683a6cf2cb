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


def _current_json(ots):
    """The topics' current assignment in the reassignment-JSON shape (README.md:52-63); a replica on a broker outside the
    target list gets an id the list does not hold."""
    parts = []
    for t in ots:
        for i in range(t.n_partitions):
            pid = i if t.partition_ids is None else int(t.partition_ids[i])
            parts.append({"topic": t.name, "partition": pid,
                          "replicas": [int(t.broker_ids[b]) if b < t.n_brokers else 1000000 + k for k, b in enumerate(t.current[i])]})
    return {"version": 1, "partitions": parts}


def _cli_args(ots, cur_path):
    t0 = ots[0]
    return ["--current", str(cur_path), "--broker-list", ",".join(str(int(b)) for b in t0.broker_ids),
            "--racks", ",".join(f"{int(b)}:r{int(r):03d}" for b, r in zip(t0.broker_ids, t0.rack_of))]


def test_cli_reads_synthetic_configs(tmp_path, ko):
    """The JSON the multi-GPU CLI test feeds kao-cli describes the same topics the oracle generated (host only: the LP text
    of every topic equals the oracle writer's)."""
    _build()
    ots = ko.gen_config(3, n_topics=2).topics + ko.gen_config(4, n_topics=2).topics[:0]
    cur_path = tmp_path / "cur.json"
    cur_path.write_text(json.dumps(_current_json(ots)))
    prefix = str(tmp_path / "m")
    r = subprocess.run([CLI] + _cli_args(ots, cur_path) + ["--emit-lp", prefix, "--lp-only"], capture_output=True)
    assert r.returncode == 0, r.stderr
    for i, t in enumerate(sorted(ots, key=lambda t: t.name)):
        assert open(f"{prefix}{i + 1}.lp").read() == ko.write_lp(t, t_index=i + 1)


@pytest.mark.gpu
def test_cli_gpus_on_logical_shards(tmp_path, ko):
    """kao-cli --gpus with an explicit device list: two logical shards on device 0 (kao_solve_multi, topics dealt LPT) give the
    same plan as one device; with ONE topic and three shards every shard searches it and the elites travel through the grouped
    all-reduce / broadcast calls, served by the loop-back table (KAO_RCCL_LOOPBACK=1) since the box has one GPU."""
    _build()
    ots = ko.gen_config(3, n_topics=4).topics
    cur = _current_json(ots)
    cur_path = tmp_path / "cur.json"
    cur_path.write_text(json.dumps(cur))
    t0 = ots[0]
    args = _cli_args(ots, cur_path) + ["--report", "--seed", "3"]
    one = subprocess.run([CLI] + args, capture_output=True, timeout=300)
    two = subprocess.run([CLI] + args + ["--gpus", "0,0"], capture_output=True, timeout=300)
    assert one.returncode == 0 and two.returncode == 0, (one.stderr, two.stderr)
    assert one.stderr.count(b"status=OPTIMAL_PROVEN") == two.stderr.count(b"status=OPTIMAL_PROVEN") == 4
    def objectives(stderr):   # "topic <name>: status=... objective=<o> bound=<b> replica_moves=<m> ..." per topic
        import re
        return sorted(re.findall(rb"topic (\S+): status=(\w+) objective=(\d+) bound=(\d+)", stderr))
    assert objectives(one.stderr) == objectives(two.stderr) and len(objectives(one.stderr)) == 4   # the same proven optima
    # one topic, three "ranks": replicated search + elite exchange through the loop-back collectives
    single = {"version": 1, "partitions": [e for e in cur["partitions"] if e["topic"] == t0.name]}
    cur_path.write_text(json.dumps(single))
    env = dict(os.environ, KAO_RCCL_LOOPBACK="1")
    three = subprocess.run([CLI] + args + ["--gpus", "0,0,0"], capture_output=True, timeout=300, env=env)
    assert three.returncode == 0 and three.stderr.count(b"status=OPTIMAL_PROVEN") == 1, three.stderr
    assert [o for o in objectives(one.stderr) if o[0] == t0.name.encode() + b":" or o[0] == t0.name.encode()] == objectives(three.stderr)
    assert len(json.loads(three.stdout)["partitions"]) == t0.n_partitions


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["solve", "solve-multi"])
def test_jni_shim_runs_the_readme_example(tmp_path, mode):
    """cli/java/kao_jni.c executed (fake JNIEnv, tests/jni_stub/fake_env.c -- no JDK in the image): Kao.init, Kao.solve (one
    device / two logical shards through the `devices` argument), Kao.canonicalize and Kao.evaluate on README.md:52-63 give
    objective 58 = bound, one move, partition 1 -> [8,1] (README.md:88)."""
    from test_host import build_jni_harness
    out = subprocess.run([build_jni_harness(tmp_path), mode], capture_output=True, timeout=120)
    assert out.returncode == 0 and b"jni_harness solve: ok" in out.stdout, (out.stdout, out.stderr)
    assert b"status=0 objective=58 bound=58 eval_objective=58 eval_violation=0 p1=[8,1]" in out.stdout
