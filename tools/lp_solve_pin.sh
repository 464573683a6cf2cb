#!/usr/bin/env bash
# Pins parity with the reference's own solver the day lp_solve 5.5 is available (README.md:135-136: "lp_solve is used
# behind the scene"; it is not installed in the build image and there is no network).  For every golden instance in
# tests/golden/ this emits the generated model as lp_solve LP text (README.md:144-185; the writer is
# oracle/kao_oracle.py::write_lp, byte-identical to `kao-cli --emit-lp`), runs `lp_solve -S4 -max` semantics as the text
# declares (`max:`), and diffs the objective value against the HiGHS optimum stored in the golden file.
#   tools/lp_solve_pin.sh [golden.json ...]        default: kat1 random_small random_medium cfg2 cfg3 cfg4 + drifted
# Exit status 0 = every objective agrees (parity pinned), 1 = a mismatch, 2 = lp_solve not found.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LP_SOLVE="${LP_SOLVE:-$(command -v lp_solve || true)}"
if [ -z "$LP_SOLVE" ]; then
    echo "lp_solve not found (set LP_SOLVE=/path/to/lp_solve); parity with the reference's solver stays unpinned" >&2
    exit 2
fi
OUT="${TMPDIR:-/tmp}/kao_lp_pin.$$"
mkdir -p "$OUT"
FILES=("$@")
[ ${#FILES[@]} -eq 0 ] && FILES=(kat1.json random_small.json random_medium.json cfg2.json cfg3.json cfg4.json cfg2_drift.json cfg3_drift.json cfg4_drift.json)
python3 - "$ROOT" "$OUT" "${FILES[@]}" <<'PY'
import json, os, sys
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(root, "oracle"))
import kao_oracle as ko
n = 0
with open(os.path.join(out, "index.tsv"), "w") as idx:
    for name in sys.argv[3:]:
        doc = json.load(open(os.path.join(root, "tests", "golden", name)))
        entries = [doc] if "topic" in doc else doc.get("cases", doc.get("topics", []))
        for i, e in enumerate(entries):
            if e.get("status", "optimal") not in ("optimal", "infeasible"):
                continue
            t = ko.topic_from_dict(e["topic"])
            path = os.path.join(out, f"{os.path.splitext(name)[0]}_{i}.lp")
            with open(path, "w") as f:
                f.write(ko.write_lp(t))
            idx.write(f"{path}\t{e.get('status', 'optimal')}\t{e.get('objective', '')}\n")
            n += 1
print(f"emitted {n} models into {out}")
PY
bad=0; n=0
while IFS=$'\t' read -r lp status want; do
    res="$("$LP_SOLVE" -S4 "$lp" 2>&1)"
    n=$((n + 1))
    if [ "$status" = "infeasible" ]; then
        echo "$res" | grep -qi "infeasible" || { echo "MISMATCH $lp: expected infeasible, lp_solve says: $(echo "$res" | head -3)"; bad=$((bad + 1)); }
        continue
    fi
    got="$(echo "$res" | sed -n 's/^Value of objective function: *\([-0-9.]*\).*/\1/p' | head -1)"
    if [ -z "$got" ] || [ "$(printf '%.0f' "$got")" != "$want" ]; then
        echo "MISMATCH $lp: lp_solve objective '$got', golden (HiGHS) $want"; bad=$((bad + 1))
    fi
done < "$OUT/index.tsv"
echo "lp_solve pin: $n models, $bad mismatches"
[ "$bad" -eq 0 ]
