# usage on the GPU box: bash tools/run_variants.sh name1 name2 ...   (libraries built by tools/build_variants.py)
for v in "$@"; do
  echo -n "$v "; TFHE_HIP_BACKEND_LIB=variants/lib_$v.so python bench.py --no-cpu-baseline --no-verify --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
done
echo -n "default "; python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])"
