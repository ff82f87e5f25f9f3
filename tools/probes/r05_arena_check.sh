#!/bin/bash
# GPU box: the whole -m gpu tier on the arena, then one FheUint64 add / mul through the compiled C++ host (alloc / scratch / drop
# per operation, as the reference's host) with the arena and with hipMalloc / hipFree (TFHE_HIP_MALLOC_ASYNC=sync)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r05_gputest_arena.log 2>&1; tail -8 gpurun_out/r05_gputest_arena.log
g++ -std=c++17 -O2 -o /tmp/int_hip tests/cpp/reference_integer_gpu_tests.cpp tfhe_rs_amd/lib/libtfhe_hip_backend.so oracle/libtfhe_oracle.so -Wl,-rpath,$PWD/tfhe_rs_amd/lib -Wl,-rpath,$PWD/oracle
: > gpurun_out/r05_alloc_latency.jsonl
for round in 1 2; do
  for mode in arena sync; do
    TFHE_HIP_MALLOC_ASYNC=$mode timeout 600 /tmp/int_hip latency >> gpurun_out/r05_alloc_latency.jsonl 2>&1
  done
done
cut -c1-330 gpurun_out/r05_alloc_latency.jsonl
