for i in 1 2 3; do
  for v in lds2 cur; do
    if [ $v = cur ]; then unset TFHE_HIP_BACKEND_LIB; else export TFHE_HIP_BACKEND_LIB=variants/lib_$v.so; fi
    echo -n "$v "; python bench.py --no-cpu-baseline --no-verify --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms_avg'],3))"
  done
done
