#!/bin/bash
# GPU box: A/B of variant libraries on chosen measure_all selectors, interleaved rounds
# usage: bash tools/ab.sh "<selectors>" "<variants>" [rounds]   -> gpurun_out/ab.txt
out=gpurun_out/ab.txt; mkdir -p gpurun_out; : > $out
sel=$1; vs=$2; rounds=${3:-2}
for r in $(seq 1 $rounds); do
  for v in default $vs; do
    lib=variants/lib_$v.so; [ "$v" = default ] && lib=tfhe_rs_amd/lib/libtfhe_hip_backend.so
    echo -n "$v: " | tee -a $out
    TFHE_HIP_BACKEND_LIB=$lib python tools/measure_all.py $sel 2>&1 | grep '"batch": 4096' | sed -e 's/.*"params": "\([^"]*\)".*"ms": \([0-9.]*\).*/\1 \2/' | tr '\n' ' ' | tee -a $out
    echo | tee -a $out
  done
done
