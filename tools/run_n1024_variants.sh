# usage on the GPU box: bash tools/run_n1024_variants.sh name1 ...  (N=1024 k=2 set, batch 4096; libraries from tools/build_variants.py)
for v in "$@"; do
  echo -n "$v "; TFHE_HIP_BACKEND_LIB=variants/lib_$v.so python tools/measure_all.py n1024 2>&1 | tail -1 | cut -c100-200
done
echo -n "default "; python tools/measure_all.py n1024 2>&1 | tail -1 | cut -c100-200
