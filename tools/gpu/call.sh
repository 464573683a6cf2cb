#!/bin/bash
# round 5, call 31: at 6 waves per SIMD, does the shorter scan loop pay?  (old loop @6 against new loop @6, alternating) + replay tests on the shipped library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c31
L=kafka_assignment_optimizer_amd/libkao.so
cp $L /tmp/new.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "replay or bit_exact or random_small or varied_shapes" > gpurun_out/${T}_pytest.log 2>&1; tail -2 gpurun_out/${T}_pytest.log
for i in 1 2 3; do
  for v in oldloop6 new; do
    if [ $v = new ]; then cp /tmp/new.so $L; else cp build_ab/libkao_$v.so $L; fi
    timeout 600 python bench.py --steps 20 --warmup 3 --no-extras 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print('$v', b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'])" >> gpurun_out/${T}_ab.log
  done
done
cp /tmp/new.so $L
cat gpurun_out/${T}_ab.log
