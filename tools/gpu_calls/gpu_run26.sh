#!/bin/bash
# round 3, GPU call 26 (final state): whole GPU suite, smoke, family + scale probe, bench, rocprofv3 kernel trace of the bench,
# north-star profile of config 5 as one 100,000-partition topic
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 700 python -m pytest tests -m gpu -q) > gpurun_out/r26_pytest.log 2>&1
tail -4 gpurun_out/r26_pytest.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r26_smoke.log 2>&1; tail -2 gpurun_out/r26_smoke.log
(time R3_SCHEDS=0 timeout 300 python tools/r3_probe.py family,scale 3.0) > gpurun_out/r26_family.log 2>&1
grep "proven" gpurun_out/r26_family.log
(time timeout 400 python bench.py) > gpurun_out/r26_bench.json 2> gpurun_out/r26_bench.err
tail -c 600 gpurun_out/r26_bench.json
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r03e; mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python $REPO/bench.py --steps 10 --warmup 2 --no-extras > "$OUT/bench_trace.json" 2> "$OUT/trace.err")
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; head -30 "$OUT/summary.txt"
timeout 900 bash tools/profile_big.sh r03e cfg5one > gpurun_out/r26_big_cfg5one.log 2>&1
tail -30 gpurun_out/r26_big_cfg5one.log
