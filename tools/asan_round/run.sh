#!/bin/bash
# The host half of KAO-LP's rounding (kao_round.cpp) under AddressSanitizer + UBSan, CPU only: the file is compiled a second time with
# g++ -fsanitize=address,undefined into a scratch library (derive_bounds comes from the shipped libkao.so through kao_derive_bounds) and
# run on the iterates of the parity sweeps -- golden families at a vertex and five iterations in, drifted topics at tolerances
# 1e-8 ... 1e-2 with and without fallback rows, pickled 100,000-partition iterates when /tmp/exp holds them.  Then the band repair alone (mode 2) on 120 perturbed assignments.  Expected: no sanitizer
# report, every assignment identical to the shipped library's.   usage: tools/asan_round/run.sh
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
out=${KAO_ROUND_ASAN_DIR:-/tmp/kao_round_asan}; mkdir -p "$out"
cs=$root/kafka_assignment_optimizer_amd/csrc
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -fPIC -shared -D__HIP_PLATFORM_AMD__ \
    -I/opt/rocm/include -I"$cs" "$cs/kao_round.cpp" "$here/shim.cpp" -o "$out/libround_asan.so" \
    -L"$root/kafka_assignment_optimizer_amd" -lkao -Wl,-rpath,"$root/kafka_assignment_optimizer_amd"
KAO_ROUND_ASAN_LIB=$out/libround_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python "$here/run_asan.py"
KAO_ROUND_ASAN_LIB=$out/libround_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python "$here/run_asan_repair.py"
