"""CPU tests of the host layer: the C-ABI library loads and exports every symbol include/kao.h
declares, the host-only helpers agree with the oracle, JSON I/O matches the reassignment format, and
compute entry points fail loudly (no CPU fallback) when there is no GPU."""
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, have_gpu, load_golden, to_product_topic


def test_library_exports_every_declared_symbol():
    from kafka_assignment_optimizer_amd import _ffi
    header = open(os.path.join(ROOT, "include", "kao.h")).read()
    declared = set(re.findall(r"\b(kao_[a-z_]+)\s*\(", header))
    declared -= {"kao_init(device", "kao_strerror"} - {"kao_strerror"}
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    lib = _ffi.load()  # raises if the .so is missing or lacks a symbol
    assert lib.kao_version() == 103
    assert lib.kao_strerror(-3).decode().startswith("no usable HIP device")


def test_struct_layouts_match_header():
    import ctypes as C
    from kafka_assignment_optimizer_amd import _ffi
    assert C.sizeof(_ffi.KaoTopic) == 5 * 4 + 4 + 2 * 8 + 16 + 8 * 4 + 2 * 8  # 5 ints, pad, 2 pointers, w[2][2], 8 bounds, 2 pointers
    assert C.sizeof(_ffi.KaoOpts) == 8 + 8 + 16 * 4 + 8
    assert C.sizeof(_ffi.KaoResult) == 4 + 4 + 8 + 8 + 32 + 8 + 8
    assert C.sizeof(_ffi.KaoStats) == 3 * 8 + 2 * 8 + 2 * 8 + 6 * 4


def test_header_is_plain_c(tmp_path):
    """include/kao.h compiles as pedantic C99 (the JNI shim of INTEGRATION.md is C), its struct sizes are the ones
    the ctypes binding uses, and a C program can call the host-only entry points."""
    import ctypes as C
    import subprocess
    from kafka_assignment_optimizer_amd import _ffi
    assert (C.sizeof(_ffi.KaoTopic), C.sizeof(_ffi.KaoOpts), C.sizeof(_ffi.KaoResult), C.sizeof(_ffi.KaoStats)) == (104, 88, 72, 80)
    exe = str(tmp_path / "abi_check")
    libdir = os.path.join(ROOT, "kafka_assignment_optimizer_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_check.c"), "-o", exe, "-L", libdir, "-lkao",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    out = subprocess.run([exe], capture_output=True)
    assert out.returncode == 0 and out.stdout.strip() == b"abi ok", (out.returncode, out.stdout, out.stderr)


def test_host_helpers_match_oracle(ko):
    import kafka_assignment_optimizer_amd as kao
    cases = [ko.readme_example(), ko.gen_config(2).topics[0], ko.gen_config(3, n_topics=1).topics[0],
             ko.gen_config(5, n_topics=1).topics[0]] + [ko.random_case(s) for s in range(60)] + \
            [ko.random_case(s, max_b=40, max_p=40) for s in range(1000, 1040)]
    for ot in cases:
        if ot.rf > 8 or ot.rf_cur > 8:
            continue
        pt = to_product_topic(ot)
        assert kao.derive_bounds(pt) == ot.bounds()
        assert kao.upper_bound(pt) == ko.upper_bound_forced(ot) <= ko.upper_bound_simple(ot)


def test_infeasibility_proofs_match_oracle_and_highs(ko):
    """kao_check_infeasible (counting arguments) agrees with the oracle's restatement, never fires on an instance
    HiGHS solved, and catches every golden instance HiGHS proved infeasible (what lp_solve reports as "This problem is
    infeasible")."""
    import kafka_assignment_optimizer_amd as kao
    n_inf = 0
    for name in ("random_small.json", "random_medium.json"):
        for c in load_golden(name)["cases"]:
            ot = ko.topic_from_dict(c["topic"])
            why = kao.check_infeasible(to_product_topic(ot))
            assert bool(why) == bool(ko.provably_infeasible(ot)), c["seed"]
            if c["status"] == "infeasible":
                assert why, c["seed"]
                n_inf += 1
            else:
                assert not why, (c["seed"], why)
    assert n_inf >= 30
    h5 = ko.make_cluster("h5", 100, 4, 1, 64, 3, [3, 17, 42, 77, 99], []).topics[0]
    assert "rack" in kao.check_infeasible(to_product_topic(h5))
    assert not kao.check_infeasible(to_product_topic(ko.readme_example()))


def test_wide_family_host_bound_and_infeasibility(ko):
    """tests/golden/random_wide.json (400 instances: uneven racks, RF <= 4 with RF changes, random weights): the
    generator still produces the fingerprinted instances; the host's closed-form bound equals the oracle's and never
    undercuts the HiGHS optimum; every instance HiGHS found infeasible is proven infeasible by counting."""
    import zlib
    import kafka_assignment_optimizer_amd as kao
    n_opt = n_inf = 0
    for c in load_golden("random_wide.json")["cases"]:
        ot = ko.random_case_wide(c["seed"])
        assert [ot.n_brokers, ot.n_racks, ot.n_partitions, ot.rf_cur, ot.rf] == c["shape"]
        assert zlib.crc32(ot.current.astype("<u2").tobytes()) == c["current_crc"]
        pt = to_product_topic(ot)
        why = kao.check_infeasible(pt)
        assert bool(why) == c["proven_infeasible_by_counting"] == bool(ko.provably_infeasible(ot))
        if c["status"] == "infeasible":
            assert why, c["seed"]
            n_inf += 1
        else:
            assert not why
            assert kao.upper_bound(pt) == c["upper_bound"] >= c["objective"], c["seed"]
            n_opt += 1
    assert n_opt >= 150 and n_inf >= 150


def test_jni_shim_compiles():
    """cli/java/kao_jni.c is real C against include/kao.h: it compiles (no JDK here, so against tests/jni_stub/jni.h, which
    declares the JNIEnv members it uses with their real signatures) with -Wall -Wextra -Werror, and it exports one
    Java_io_sqooba_kao_Kao_<name> per native method io/sqooba/kao/Kao.java declares."""
    import re
    import subprocess
    src = os.path.join(ROOT, "cli", "java", "kao_jni.c")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"), src])
    java = open(os.path.join(ROOT, "cli", "java", "io", "sqooba", "kao", "Kao.java")).read()
    natives = set(re.findall(r"public static native [\w\[\]]+ (\w+)\(", java))
    exported = set(re.findall(r"Java_io_sqooba_kao_Kao_(\w+)\(", open(src).read()))
    assert natives == exported == {"init", "solve", "evaluate", "canonicalize", "checkInfeasible"}


def build_jni_harness(tmp_path):
    """cli/java/kao_jni.c linked with the fake JNIEnv of tests/jni_stub/fake_env.c (no JDK in the image): the shim RUNS."""
    import subprocess
    exe = str(tmp_path / "jni_harness")
    libdir = os.path.join(ROOT, "kafka_assignment_optimizer_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "jni_stub", "fake_env.c"),
                           os.path.join(ROOT, "cli", "java", "kao_jni.c"), "-o", exe, "-L", libdir, "-lkao",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    return exe


def test_jni_shim_validates_its_arguments(tmp_path):
    """the shim sized native buffers from caller-supplied counts.  Run against a fake JNIEnv, every malformed call
    (short `current` / `outAssignment` / `rackOf`, negative counts, null arrays, wrong assignment length, rf > 8) must raise
    IllegalArgumentException before anything reaches the C ABI -- no GPU involved -- and a well-formed host-only call works."""
    import subprocess
    out = subprocess.run([build_jni_harness(tmp_path), "validate"], capture_output=True)
    assert out.returncode == 0 and b"jni_harness validate: ok" in out.stdout, (out.stdout, out.stderr)


def test_java_cli_mirrors_the_cpp_cli_flags():
    """The Java CLI cannot be compiled here (no JDK); at least its flag set is held to the C++ CLI's (minus the host-only
    LP export) and its JSON output statement to the README shape (README.md:67-78)."""
    import re
    cpp = open(os.path.join(ROOT, "cli", "kao_cli.cpp")).read()
    java = open(os.path.join(ROOT, "cli", "java", "io", "sqooba", "kao", "KaoCli.java")).read()
    cpp_flags = set(re.findall(r'a == "(--[a-z-]+)"', cpp))
    java_flags = set(re.findall(r'case "(--[a-z-]+)"', java))
    assert cpp_flags - java_flags == {"--emit-lp", "--lp-only"} and java_flags <= cpp_flags and "--gpus" in java_flags
    assert "devices" in open(os.path.join(ROOT, "cli", "java", "kao_jni.c")).read() and "kao_solve_multi" in open(os.path.join(ROOT, "cli", "java", "kao_jni.c")).read()
    assert '{\\"version\\":1,\\"partitions\\":[' in java and "Kao.solve(" in java and "Kao.canonicalize(" in java


def test_validation_errors(ko):
    import kafka_assignment_optimizer_amd as kao
    pt = to_product_topic(ko.readme_example())
    pt.rf = 9
    with pytest.raises(kao.KaoError) as e:
        kao.derive_bounds(pt)
    assert e.value.code == -2  # KAO_ERR_UNSUPPORTED (more than KAO_MAX_RF = 8 replicas)
    pt = to_product_topic(ko.readme_example())
    pt.rack_of = pt.rack_of.copy()
    pt.rack_of[3] = 9
    with pytest.raises(kao.KaoError) as e:
        kao.upper_bound(pt)
    assert e.value.code == -1
    pt = to_product_topic(ko.readme_example())     # a broker listed twice in one partition of the current assignment
    pt.current = pt.current.copy()
    pt.current[4, 1] = pt.current[4, 0]
    with pytest.raises(kao.KaoError) as e:
        kao.upper_bound(pt)
    assert e.value.code == -1 and "twice" in str(e.value)
    pw = to_product_topic(ko.readme_example())
    pw.weights = ((4, -1), (2, 2))                 # negative / oversized objective weights
    with pytest.raises(kao.KaoError) as e:
        kao.upper_bound(pw)
    assert e.value.code == -1
    pw.weights = ((5000, 1), (2, 2))
    with pytest.raises(kao.KaoError):
        kao.upper_bound(pw)
    pt.current[4, 1] = 0xFFFF                      # two empty slots are fine
    pt.current[4, 0] = 0xFFFF
    kao.upper_bound(pt)


def test_json_roundtrip_readme():
    """README.md:52-63 in, README.md:67-78-shaped JSON out."""
    import kafka_assignment_optimizer_amd as kao
    cur = {"version": 1, "partitions": [
        {"topic": "x.y.z.t", "partition": 1, "replicas": [8, 19]},
        {"topic": "x.y.z.t", "partition": 0, "replicas": [7, 18]},
        {"topic": "a", "partition": 0, "replicas": [1, 2]}]}
    racks = {str(b): ("a" if b % 2 == 0 else "b") for b in range(20)}
    topics = kao.topics_from_json(cur, list(range(19)), racks)
    assert [t.name for t in topics] == ["a", "x.y.z.t"]
    t = topics[1]
    assert t.current.tolist() == [[7, 18], [8, 0xFFFF]]  # broker 19 is not in the target list
    assert t.rack_of.tolist() == [b % 2 for b in range(19)]
    out = kao.assignment_to_json(topics, [np.array([[1, 2]]), np.array([[7, 18], [8, 1]])])
    assert out == {"version": 1, "partitions": [
        {"topic": "a", "partition": 0, "replicas": [1, 2]},
        {"topic": "x.y.z.t", "partition": 0, "replicas": [7, 18]},
        {"topic": "x.y.z.t", "partition": 1, "replicas": [8, 1]}]}
    json.dumps(out)
    with pytest.raises(ValueError):
        kao.topics_from_json(cur, [0, 1, 1], racks)


@pytest.mark.skipif(have_gpu(), reason="checks the no-device failure mode")
def test_compute_fails_loudly_without_gpu(ko):
    """The product has no CPU fallback: without a device every compute call raises."""
    import kafka_assignment_optimizer_amd as kao
    pt = to_product_topic(ko.readme_example())
    with pytest.raises(kao.KaoError) as e:
        kao.evaluate(pt, pt.current)
    assert e.value.code == -3
    with pytest.raises(kao.KaoError):
        kao.solve([pt], max_launches=1)
    with pytest.raises(kao.KaoError):
        kao.Session([pt])


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "kafka_assignment_optimizer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"import\s+kao_(oracle|port)", r"from\s+(oracle|kao_oracle|kao_port)\b", r"libkao_port",
                            r"#include\s+[\"<][^\n]*oracle", r"sys\.path[^\n]*oracle", r"dlopen[^\n]*oracle"):
                    assert not re.search(pat, src), (f, pat)


@pytest.mark.slow
def test_host_code_under_asan_and_ubsan():
    """tools/sanitize_host.sh: the host side (validation, bands, closed-form bound, infeasibility proofs, LP writer, CLI parsing)
    rebuilt with -fsanitize=address,undefined and run over the golden families -- no GPU involved.  Minutes of hipcc: -m slow."""
    import subprocess
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh")], cwd=ROOT, capture_output=True, timeout=1800)
    assert out.returncode == 0 and b"sanitised host code: ok" in out.stdout and b"sanitised kao-cli --emit-lp: ok" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


def test_pair_edges_match_the_oracle_prototype():
    """kao_cycle_pair_edges (host only: the enumeration of the next KAO-CX layer, kao_pairs.cpp) against oracle/kao_cycle_pairs.py:
    the same compound-edge cost matrix and the same counts (half-moves, pairs, edges, closed pairs) on drifted topics (rigid and
    slack bands, RF 2..4), wide-family topics with and without broker weights, and RF 5..8 topics."""
    import numpy as np
    import kao_oracle as ko
    import kao_port as kp
    import kao_cycle as kc
    import kao_cycle_pairs as kcp
    import kafka_assignment_optimizer_amd as kao
    from kafka_assignment_optimizer_amd import synthetic as sy
    from conftest import to_product_topic
    cases = []
    for (B, R, P, rf) in [(60, 3, 200, 3), (45, 5, 130, 2), (50, 2, 120, 3), (48, 6, 160, 4)]:
        pt = sy.drift(sy.make_cluster(B, R, 1, P, rf, [], []), 0.2, 1)[0]
        cases.append(ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
                              n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights))
    rng = np.random.default_rng(5)
    for s in range(12):
        t = ko.random_case_wide(s)
        if 2 <= t.rf <= 4:
            if s % 2:
                t.broker_w = rng.integers(0, 4, t.n_brokers).astype(np.int32)
            cases.append(t)
    cases += [t for t in (ko.random_case_rf(s) for s in range(6)) if t.rf > 4]
    n = n_edges = 0
    for t in cases:
        r = kp.port_search(t, 3, 0, 4, 128)
        if r["best_obj"] < 0:
            continue
        X = r["best"]
        edges, closed = kcp.compound_edges(kc.Round(t, X), verbose=False)
        want = np.full((t.n_brokers, t.n_brokers), np.iinfo(np.int32).max, dtype=np.int32)
        for (x, z), (c, _) in edges.items():
            want[x, z] = c
        got, st = kao.cycle_pair_edges(to_product_topic(t), X, -2)
        assert np.array_equal(got, want), t.name
        assert st["edges"] == len(edges) and st["closed_improving"] == len(closed)
        n += 1
        n_edges += len(edges)
    assert n >= 8 and n_edges > 1000


def test_topic_from_dict_reads_the_golden_layout(ko):
    """The product's Topic.from_dict (what bench.py's capped_cluster leg reads tests/golden/capped_medium.json with) agrees with
    the oracle's topic_from_dict field by field, and the C-side host helpers give the same bounds for both."""
    import kafka_assignment_optimizer_amd as kao
    from conftest import load_golden, to_product_topic
    c = load_golden("capped_medium.json")["cases"][0]
    for d in c["topics"][:4]:
        pt, ot = kao.Topic.from_dict(d), ko.topic_from_dict(d)
        via = to_product_topic(ot)
        assert (pt.n_brokers, pt.n_racks, pt.n_partitions, pt.rf, pt.rf_cur) == (via.n_brokers, via.n_racks, via.n_partitions, via.rf, via.rf_cur)
        assert np.array_equal(pt.current, via.current) and np.array_equal(pt.rack_of, via.rack_of) and np.array_equal(pt.broker_ids, via.broker_ids)
        assert tuple(map(tuple, pt.weights)) == tuple(map(tuple, via.weights)) and pt.bounds_override == via.bounds_override
        assert kao.derive_bounds(pt) == kao.derive_bounds(via) and kao.upper_bound(pt) == kao.upper_bound(via)


def test_pair_enumeration_refuses_large_topics():
    """the compound-edge enumeration (kao_pairs.cpp) is quadratic per leader pair; beyond 4,000,000 partition-broker
    pairs it reports no edges instead of running for minutes (host only, no GPU needed)."""
    import kafka_assignment_optimizer_amd as kao
    from kafka_assignment_optimizer_amd import synthetic as sy
    pt = sy.make_cluster(500, 10, 1, 9000, 3, [], [])[0]          # 4.5e6 pairs, balanced start: a feasible assignment
    X = np.ascontiguousarray(pt.current[:, :3], dtype=np.uint16)
    got, st = kao.cycle_pair_edges(pt, X, -2)
    assert st["half_moves"] == 0 and st["edges"] == 0 and (got == np.iinfo(np.int32).max).all()


def test_bench_result_line_is_short_and_strict_json():
    """The driver parses the LAST stdout line of bench.py and keeps a bounded tail (round 5's 21 KB line was not parsed): the
    compact line built from the fattest record we have committed stays under 4 KB, is strict JSON, and keeps the contract's keys."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "r05_zz_bench.json")) as f:
        fat = json.load(f)
    fat["north_star"] = {"workload": "drift100k", "time_limit_s": 1.0, "seconds": 0.9, "status": "OPTIMAL_PROVEN", "objective": 782512,
                         "certificate": 782512, "note": "x" * 5000}
    fat["roofline_lp"] = {"kernel": "k_lp_*", "bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": 1,
                          "ms_per_iteration": 4.0, "note": "y" * 5000}
    line = json.dumps(bench.compact_line(fat))
    assert len(line) < bench.FINAL_LINE_LIMIT <= 4096
    back = json.loads(line, parse_constant=lambda c: pytest.fail("non-strict JSON constant " + c))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert "workload" in back["config"] and back["config"]["workload"].startswith("cfg4")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind"):
        assert k in back["cpu_baseline"], k
    assert all(not isinstance(v, str) or len(v) <= 200 for d in back.values() if isinstance(d, dict) for v in d.values())
