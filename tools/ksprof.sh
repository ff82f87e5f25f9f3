# keyswitch 2048 -> 918, 4 levels, batch 4096: the three kernel choices (0 digit pass + staged GEMM, 2 one-launch matrix-core kernel, 1 scalar)
for c in 0 2; do
  echo "== choice $c"; TFHE_KS_CHOICE=$c python tools/measure_all.py ks ks32 ks1024 2>&1 | cut -c1-200
done
