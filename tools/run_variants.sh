for v in e0f0 e0f1 e1f0 e1f1 e2f0; do
  echo -n "$v "; TFHE_HIP_BACKEND_LIB=variants/lib_$v.so python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
done
echo -n "default(e2f1) "; python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
