#!/bin/bash
# GPU box: A/B of the round-2 binary and of the tuning variants of the headline kernel (same box, interleaved rounds)
# usage: bash tools/ab_headline.sh "<wave variants>" "<split variants>"   -> gpurun_out/ab_headline.txt
out=gpurun_out/ab_headline.txt; mkdir -p gpurun_out; : > $out
wv=${1:-"lit own fuse litfuse all3"}; sv=${2:-"regacc"}
ms() { grep '"batch": 4096' | sed -e 's/.*"ms": \([0-9.]*\).*pbs_per_s": \([0-9.]*\).*/ms \1 pbs \2/'; }
echo "== parity of the variants (full-size 2_2 bit-exact test + decomposer boundaries)" | tee -a $out
for v in $wv; do
  echo -n "$v: " | tee -a $out
  TFHE_HIP_BACKEND_LIB=variants/lib_$v.so timeout 600 python -m pytest tests/test_backend_parity.py -q -m gpu -k "full_size_param_message_2_carry_2 or (decomposer_boundary and hip) or (wave_kernel_equals and hip)" 2>&1 | tail -1 | tee -a $out
done
for round in 1 2 3; do
  echo "== round $round" | tee -a $out
  if [ -d variants/r02_tree ]; then echo -n "r02_binary " | tee -a $out; (cd variants/r02_tree && python tools/measure_all.py wave 2>&1 | ms) | tee -a $out; fi
  for v in default $wv; do
    lib=variants/lib_$v.so; [ "$v" = default ] && lib=tfhe_rs_amd/lib/libtfhe_hip_backend.so
    echo -n "$v " | tee -a $out; TFHE_HIP_BACKEND_LIB=$lib python tools/measure_all.py wave 2>&1 | ms | tee -a $out
  done
done
echo "== split-key exact engine (config 3)" | tee -a $out
for v in $sv; do
  echo -n "$v parity: " | tee -a $out
  TFHE_HIP_BACKEND_LIB=variants/lib_$v.so timeout 900 python -m pytest tests/test_backend_parity.py -q -m gpu -k "full_size_ntt_engine_wide_batch" 2>&1 | tail -1 | tee -a $out
done
for round in 1 2; do
  for v in default $sv; do
    lib=variants/lib_$v.so; [ "$v" = default ] && lib=tfhe_rs_amd/lib/libtfhe_hip_backend.so
    echo -n "$v " | tee -a $out; TFHE_HIP_BACKEND_LIB=$lib python tools/measure_all.py ntt_split 2>&1 | ms | tee -a $out
  done
done
