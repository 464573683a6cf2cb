#!/bin/bash
# round 3, GPU call 30: bench.py --in-library on logical shards of one GPU (functional check of the in-library multi-GPU bench mode)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 100 python bench.py --gpus 2 --in-library --devices 0,0 --steps 3 --warmup 1 > gpurun_out/r30_inlib.json 2> gpurun_out/r30_inlib.err
tail -c 900 gpurun_out/r30_inlib.json; tail -3 gpurun_out/r30_inlib.err
