#!/bin/bash
# Host-side code (validation, closed-form bound, infeasibility proofs, LP writer) under ASan + UBSan, no GPU needed:
# builds a sanitised libkao.so / kao-cli in /tmp/kao_san and runs them over the golden families.  Run from the repo root.
set -eu
REPO=$(pwd)
OUT=/tmp/kao_san
mkdir -p "$OUT"
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
( cd kafka_assignment_optimizer_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -shared $SAN -o "$OUT/libkao.so" kao_kernels.hip kao_bound.hip kao_cycle.hip -x hip kao_model.cpp kao_session.cpp kao_solve.cpp -ldl )
( cd cli && g++ -std=c++17 $SAN -o "$OUT/kao-cli" kao_cli.cpp -L"$OUT" -lkao -Wl,-rpath,"$OUT" -Wl,-rpath-link,/opt/rocm/lib -Wl,--allow-shlib-undefined )
ASAN_LIB=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cat > "$OUT/run.py" <<PY
import json, sys
sys.path[:0] = ["$REPO", "$REPO/oracle", "$REPO/tests"]
from kafka_assignment_optimizer_amd import _ffi
_ffi.LIB_PATH = "$OUT/libkao.so"
import kafka_assignment_optimizer_amd as kao
import kao_oracle as ko
from conftest import to_product_topic
n = 0
for c in json.load(open("$REPO/tests/golden/random_wide.json"))["cases"]:
    pt = to_product_topic(ko.random_case_wide(c["seed"]))
    kao.derive_bounds(pt); kao.check_infeasible(pt)
    ub = kao.upper_bound(pt)
    assert c["status"] != "optimal" or ub == c["upper_bound"]
    n += 1
for cfg in (2, 3, 4, 5):
    for t in ko.gen_config(cfg, n_topics=2).topics:
        pt = to_product_topic(t); kao.upper_bound(pt); kao.check_infeasible(pt); n += 1
for s in range(200):
    t = ko.random_case(s, max_b=40, max_p=40)
    if t.rf <= 4 and t.rf_cur <= 4:
        pt = to_product_topic(t); kao.upper_bound(pt); kao.check_infeasible(pt); n += 1
print("sanitised host code: ok on", n, "instances")
PY
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python "$OUT/run.py"
ASAN_OPTIONS=detect_leaks=0 "$OUT/kao-cli" --current tests/golden/readme_current.json --broker-list "$(seq -s, 0 18)" \
    --racks tests/golden/readme_racks.json --emit-lp "$OUT/lp" --lp-only > /dev/null
echo "sanitised kao-cli --emit-lp: ok ($(wc -c < "$OUT/lp1.lp") bytes)"
